"""The N > 1 product path on the GPU box: two ranks launched the way the driver launches bench.py
(`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`), each rendering its tile shard with the HIP
path into its resident device film, one in-place reduce of the film, rank 0 holding the frame. With two GPUs the ranks
use RCCL ("nccl"); on a one-GPU box they share device 0 and reduce through gloo on the host (RCCL refuses two ranks on
one device) -- sharding, scene cache, step function and JSON line are the same either way."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import KILLEROO, ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_render_and_reduce_the_frame(pt, tmp_path):
    import torch
    n_dev = torch.cuda.device_count()
    out = str(tmp_path / "film.npy")
    env = dict(os.environ, MIPT_DIST_BACKEND="nccl" if n_dev >= 2 else "gloo", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "8",
           "--cpu-samples", "0", "--dump-film", out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert len(line["per_rank"]["render_s"]) == 2 and len(line["per_rank"]["reduce_s"]) == 2 and line["per_rank"]["imbalance"] >= 1.0
    # the single-process frame
    s = pt.Scene(KILLEROO, spp=8)
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    c = integ.counters.as_dict()
    assert line["camera_samples"] == c["camera_rays"] == 700 * 700 * 8
    assert line["rays"] == c["regular_rays"] + c["shadow_rays"]
    got = np.load(out)
    assert got.shape == (700, 700, 32)
    assert np.array_equal(got[..., 31], weight)
    d = got[..., :31].astype(np.float64) - film
    assert np.sqrt((d ** 2).sum() / (film.astype(np.float64) ** 2).sum()) < 1e-6   # the same samples, summed in another order
    assert abs(line["film_mean_per_sample"] - float(film.mean()) / 8) < 1e-5


def test_two_ranks_render_the_configs4_workload(pt, tmp_path):
    """BASELINE configs[4]'s workload through the two-rank launcher: the seeded 10 000 002-triangle scene at 700x700, 2048 spp,
    parsed and built once (rank 0) and handed to rank 1 through the binary scene cache, each rank rendering its tile shard
    with the HIP path, one reduce of the film. The reduced frame against the oracle's tiles of that frame
    (tests/golden/procedural_10M_2048spp_tiles.npz, correctly rounded libm: the device's arithmetic)."""
    import torch
    n_dev = torch.cuda.device_count()
    out = str(tmp_path / "film.npy")
    env = dict(os.environ, MIPT_DIST_BACKEND="nccl" if n_dev >= 2 else "gloo", OMP_NUM_THREADS="1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29548", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--procedural-tris", "10000000", "--spp", "2048", "--cpu-samples", "0", "--dump-film", out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["triangles"] == 10_000_002 and line["config"]["spp"] == 2048
    assert line["camera_samples"] == 700 * 700 * 2048
    z = np.load(os.path.join(ROOT, "tests", "golden", "procedural_10M_2048spp_tiles.npz"))
    got = np.load(out)
    assert got.shape == (700, 700, 32)
    ys, xs = z["ys"].astype(int), z["xs"].astype(int)
    # away from the pixels that a sample lying exactly on a pixel border also reaches (box radius 0.5; at 2048 spp one pixel
    # in seven has such a sample of its own or of a neighbour)
    inner = (z["weight"] == 2048) & (got[ys, xs, 31] == 2048)
    assert inner.mean() > 0.8
    d = got[ys, xs][inner][:, :31].astype(np.float64) - z["film_exact"][inner]
    rel = float(np.sqrt((d ** 2).sum() / (z["film_exact"][inner].astype(np.float64) ** 2).sum()))
    assert rel < 2e-5, rel   # (measured 8.3e-6: the reference's sequential float sums at 2048 spp, tests/test_golden.py)
    assert got[..., 31].min() >= 2048   # every pixel of the frame was rendered by one of the two ranks
