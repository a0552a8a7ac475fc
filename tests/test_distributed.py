"""N>1 path on CPU: world_size-2 gloo processes shard the film tiles, render their shard
(with the CPU oracle standing in for the GPU renderer -- tests may use it), reduce the
film with the package's helper and must reproduce the single-process film exactly."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
import pbrt_v3_spectral_amd as pt
import oracle_binding as ob
import scenes_text as st
import importlib.util
spec = importlib.util.spec_from_file_location("ptdist", os.path.join(%(root)r, "pbrt-v3-spectral_amd", "distributed.py"))
ptdist = importlib.util.module_from_spec(spec); spec.loader.exec_module(ptdist)
rank, world, local = ptdist.init_from_env(backend="gloo")
scene = pt.Scene(text=st.material_zoo(res=48, spp=4))
si, sc = ptdist.shard_of(rank, world)
film, weight, c, _ = ob.render(scene, n_threads=2, shard_index=si, shard_count=sc)
tf, tw = torch.from_numpy(film), torch.from_numpy(weight)
ptdist.barrier()
ptdist.reduce_film(tf, tw, dst=0)
tot = ptdist.sum_over_ranks([c.camera_rays, c.regular_rays + c.shadow_rays])
mx = ptdist.max_over_ranks(float(rank))
if rank == 0:
    np.save(%(out)r, tf.numpy()); np.save(%(outw)r, tw.numpy())
    open(%(outc)r, "w").write("%%d %%d %%g" %% (tot[0], tot[1], mx))
'''


def test_two_rank_tile_sharding_and_film_reduce(tmp_path, pt, ob):
    import scenes_text as st
    out, outw, outc = str(tmp_path / "f.npy"), str(tmp_path / "w.npy"), str(tmp_path / "c.txt")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=out, outw=outw, outc=outc))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    scene = pt.Scene(text=st.material_zoo(res=48, spp=4))
    film, weight, c, _ = ob.render(scene, n_threads=2)
    got, gotw = np.load(out), np.load(outw)
    assert np.allclose(got, film, rtol=1e-6, atol=0) and np.array_equal(gotw, weight)
    cam, rays, mx = open(outc).read().split()
    assert int(cam) == c.camera_rays == 48 * 48 * 4 and int(rays) == c.regular_rays + c.shadow_rays
    assert float(mx) == 1.0


def test_shards_partition_the_tiles(pt, ob):
    import scenes_text as st
    scene = pt.Scene(text=st.furnace_area(res=40, spp=2))
    full, wfull, cfull, _ = ob.render(scene, n_threads=2)
    acc = np.zeros_like(full)
    cams = 0
    for r in range(3):
        f, w, c, _ = ob.render(scene, n_threads=2, shard_index=r, shard_count=3)
        # a pixel is written by two shards only through samples that fall exactly on a pixel
        # border of a tile edge (u == 0: both neighbours get filter weight, film.h:131-136)
        assert ((f != 0).any(axis=2) & (acc != 0).any(axis=2)).sum() <= 0.01 * f.shape[0] * f.shape[1]
        acc += f
        cams += c.camera_rays
    assert np.allclose(acc, full, rtol=1e-6, atol=0) and cams == cfull.camera_rays
