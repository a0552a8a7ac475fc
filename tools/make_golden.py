#!/usr/bin/env python3
"""Write the committed golden fixtures under tests/golden/ with the CPU oracle (oracle/, the restatement
of the reference algorithm pinned by the reference's own statistics, see tests/test_oracle_pins.py).
They let the GPU box check the HIP path at BASELINE config sizes without spending CPU minutes there, and
they pin the oracle itself against regressions.

  killeroo_1024spp_crop.npz   film sum [40,40,31] + weights of the 1024-spp frame (BASELINE configs[1]) on a
                              40x40 crop window of the 700x700 film that straddles the killeroo's silhouette,
                              and the oracle's counters for it
  cornell_256spp_crop.npz     same for scenes/cornell-glass.pbrt (maxdepth 8) at 256 spp on a 32x32 crop over
                              the glass sphere's caustic
  killeroo_rays.npz           4096 camera + 4096 random rays with closest-hit (prim, t, b0, b1) and any-hit results
  textured_zoo_64spp.npz      the image-textured material zoo of tests/scenes_text.py (64x64, 64 spp): image textures of
                              three file formats, checkerboards, bump maps, textured spheres
  cornell_4096spp_crop.npz    BASELINE configs[2] at its full sample count: 512x512, maxdepth 8, 4096 spp, the same crop
  killeroo_1024spp_tiles.npz  BASELINE configs[1] as the FULL 700x700 frame at 1024 spp: the film of every 64th 16x16 tile
  cornell_4096spp_tiles.npz   BASELINE configs[2] as the FULL 512x512 frame at 4096 spp, maxdepth 8: every 64th tile
  procedural_10M_256spp.npz   BASELINE configs[3] stand-in at full size (tools/make_procedural_scene.py, 10 000 002
                              triangles, 700x700, 256 spp): the films of every 64th 16x16 tile (shard 0 of 64) and 8192
                              recorded rays (camera rays + random rays through the scene) with their hits
  procedural_10M_2048spp_tiles.npz  BASELINE configs[4]'s workload (the same scene at 2048 spp, which that config shards over
                              8 GPUs): every 64th tile of the full frame

Every film fixture holds the oracle's result twice: `film` / `counters` with the host's libm as the reference binary calls
it (glibc), and `film_exact` / `counters_exact` with correctly rounded libm calls (oracle/o_math.h mode 1), the arithmetic
the device implements -- the first is held to the BASELINE tolerance, the second (almost) exactly.

Usage: python tools/make_golden.py [--only NAME]  (a few minutes on 8 cores; the procedural fixture needs ~12 GB of RAM)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrt_v3_spectral_amd as pt          # noqa: E402
import oracle_binding as ob                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
KILLEROO = os.path.join(ROOT, "scenes", "killeroo-simple.pbrt")
CORNELL = os.path.join(ROOT, "scenes", "cornell-glass.pbrt")
KILLEROO_CROP = (0.30, 0.30 + 40 / 700, 0.44, 0.44 + 40 / 700)
CORNELL_CROP = (0.25, 0.25 + 32 / 512, 0.70, 0.70 + 32 / 512)


def both_modes(scene, **kw):
    """(film, weight, counters dict) of the oracle in glibc mode, and (film, counters) in correctly rounded mode."""
    film, weight, c, secs = ob.render(scene, **kw)
    with ob.exact_libm():
        film_x, weight_x, cx, secs_x = ob.render(scene, **kw)
    assert np.array_equal(weight, weight_x)
    return film, weight, c.as_dict(), film_x, cx.as_dict(), secs + secs_x


def film_fixture(path, scene_file, spp, crop, **scene_kw):
    s = pt.Scene(scene_file, spp=spp, crop=crop, **scene_kw)
    film, weight, d, film_x, dx, secs = both_modes(s)
    np.savez_compressed(path, film=film, weight=weight, crop=np.array(crop), spp=spp, film_exact=film_x,
                        counters=np.array([d[k] for k in sorted(d)], np.int64), counter_names=np.array(sorted(d)),
                        counters_exact=np.array([dx[k] for k in sorted(d)], np.int64))
    rel = float(np.sqrt(((film.astype(np.float64) - film_x) ** 2).sum() / (film_x.astype(np.float64) ** 2).sum()))
    print("%s: film %s mean/spp %.6f, %d camera rays, %.1f s; glibc vs exact libm: rel L2 %.2e, regular rays %+d" %
          (os.path.basename(path), film.shape, film.mean() / spp, d["camera_rays"], secs, rel, d["regular_rays"] - dx["regular_rays"]))


PROCEDURAL = dict(tris=10_000_000, res=700, spp=256, seed=7, depth=5, shard_count=64)


def procedural_scene(tmpdir):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_procedural_scene as mps
    path = os.path.join(tmpdir, "procedural_%d.pbrt" % PROCEDURAL["tris"])
    with open(path, "w") as fh:
        mps.write_scene(fh, PROCEDURAL["tris"], PROCEDURAL["res"], PROCEDURAL["spp"], PROCEDURAL["seed"], PROCEDURAL["depth"])
    return path


def procedural_rays(scene, n=4096, seed=13):
    """n camera rays + n random rays from inside the scene's bounds."""
    rng = np.random.default_rng(seed)
    res = PROCEDURAL["res"]
    samples = np.stack([rng.integers(0, res, n), rng.integers(0, res, n), np.zeros(n, int)], axis=1)
    cam = ob.camera_rays(scene, samples)
    d = scene.desc
    lo = np.array([d.nodes[0].bmin[i] for i in range(3)], np.float32)
    hi = np.array([d.nodes[0].bmax[i] for i in range(3)], np.float32)
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    dr = rng.normal(size=(n, 3)).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.inf, float(np.linalg.norm(hi - lo)) * rng.uniform(0.01, 0.3, n)).astype(np.float32)
    return np.concatenate([cam, np.concatenate([o, dr, tmax[:, None]], axis=1)]).astype(np.float32)


def tile_fixture(path, s, shard_count, rays=None):
    """Every shard_count-th 16x16 tile of the FULL frame (tile_id % shard_count == 0, the multi-GPU decomposition): the
    same sampler, film and Halton indexing as the whole frame -- a crop window would change the sampler's resolution
    (halton.cpp:75-85) and with it every sample. Stored sparsely: the pixels the shard's samples reach."""
    spp = s.spp
    film, weight, d, film_x, dx, secs = both_modes(s, shard_index=0, shard_count=shard_count)
    ys, xs = np.nonzero(weight)
    extra = {}
    if rays is not None:
        closest, _ = ob.trace(s, rays, any_hit=False)
        anyhit, _ = ob.trace(s, rays, any_hit=True)
        extra = dict(rays=rays, closest=closest.view(np.int32), anyhit=anyhit.view(np.int32)[:, 0])
    np.savez_compressed(path, ys=ys.astype(np.int16), xs=xs.astype(np.int16), film=film[ys, xs], film_exact=film_x[ys, xs],
                        weight=weight[ys, xs], spp=spp, shard_count=shard_count,
                        counters=np.array([d[k] for k in sorted(d)], np.int64), counter_names=np.array(sorted(d)),
                        counters_exact=np.array([dx[k] for k in sorted(d)], np.int64),
                        n_triangles=s.stats["n_triangles"], interior_nodes=s.stats["interior_nodes"], **extra)
    rel = float(np.sqrt(((film.astype(np.float64) - film_x) ** 2).sum() / (film_x.astype(np.float64) ** 2).sum()))
    print("%s: %d pixels (%d camera rays), mean/spp %.6f, %.1f s; glibc vs exact libm: rel L2 %.2e, regular rays %+d" %
          (os.path.basename(path), len(ys), d["camera_rays"], film[ys, xs].mean() / spp, secs, rel, d["regular_rays"] - dx["regular_rays"]))


def procedural_fixture(path, spp=None):
    """spp = None: configs[3] (256 spp, with recorded rays). spp = 2048: configs[4]'s workload -- the same scene file with the
    sample count overridden, the tiles of shard 0 of 64 (which lie inside shards 0 of 8, 2 and 4: 64 = 8 x 8)."""
    import tempfile
    s = pt.Scene(procedural_scene(tempfile.mkdtemp()), **({"spp": spp} if spp else {}))
    assert s.stats["n_triangles"] == PROCEDURAL["tris"] + 2, s.stats
    tile_fixture(path, s, PROCEDURAL["shard_count"], rays=procedural_rays(s) if spp is None else None)


def ray_fixture(path):
    s = pt.Scene(KILLEROO, spp=1)
    rng = np.random.default_rng(11)
    n = 4096
    samples = np.stack([rng.integers(0, 700, n), rng.integers(0, 700, n), np.zeros(n, int)], axis=1)
    cam = ob.camera_rays(s, samples)
    o = rng.uniform(-300, 300, (n, 3)).astype(np.float32) + np.array([0, 60, -100], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(10, 400, n)).astype(np.float32)
    rays = np.concatenate([cam, np.concatenate([o, d, tmax[:, None]], axis=1)]).astype(np.float32)
    closest, _ = ob.trace(s, rays, any_hit=False)
    anyhit, _ = ob.trace(s, rays, any_hit=True)
    np.savez_compressed(path, rays=rays, closest=closest.view(np.int32), anyhit=anyhit.view(np.int32)[:, 0])
    print("%s: %d rays, %d closest hits, %d occluded" % (os.path.basename(path), len(rays),
                                                        (closest.view(np.int32)[:, 0] >= 0).sum(), (anyhit.view(np.int32)[:, 0] >= 0).sum()))


def text_scene_fixture(path, which, res, spp):
    """A fixture of one of the authored test scenes that need generated image files (tests/scenes_text.py)."""
    import tempfile
    import scenes_text as st
    d = tempfile.mkdtemp()
    st.write_texture_files(d)
    st.write_alpha_png(d)
    text = {"textured_zoo": st.textured_zoo, "bump_scene": st.bump_scene, "alpha_scene": st.alpha_scene}[which](res=res, spp=spp)
    s = pt.Scene(text=text, base_dir=d)
    assert s.errors == [], s.errors
    film, weight, dd, film_x, dx, secs = both_modes(s)
    np.savez_compressed(path, film=film, weight=weight, spp=spp, res=res, scene=which, film_exact=film_x,
                        counters=np.array([dd[k] for k in sorted(dd)], np.int64), counter_names=np.array(sorted(dd)),
                        counters_exact=np.array([dx[k] for k in sorted(dd)], np.int64))
    print("%s: film %s mean/spp %.6f, %.1f s" % (os.path.basename(path), film.shape, film.mean() / spp, secs))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    jobs = {
        "textured_zoo_64spp": lambda p: text_scene_fixture(p, "textured_zoo", 64, 64),
        "killeroo_rays": ray_fixture,
        "cornell_256spp_crop": lambda p: film_fixture(p, CORNELL, 256, CORNELL_CROP),
        "killeroo_1024spp_crop": lambda p: film_fixture(p, KILLEROO, 1024, KILLEROO_CROP),
        "cornell_4096spp_crop": lambda p: film_fixture(p, CORNELL, 4096, CORNELL_CROP),
        "killeroo_1024spp_tiles": lambda p: tile_fixture(p, pt.Scene(KILLEROO, spp=1024), 64),
        "cornell_4096spp_tiles": lambda p: tile_fixture(p, pt.Scene(CORNELL, spp=4096), 64),
        "procedural_10M_256spp": procedural_fixture,
        "procedural_10M_2048spp_tiles": lambda p: procedural_fixture(p, spp=2048),
    }
    for name, job in jobs.items():
        if only is None or only == name:
            job(os.path.join(OUT, name + ".npz"))
