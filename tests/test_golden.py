"""Committed golden vectors (tests/golden/, written by tools/make_golden.py with the CPU oracle).

Every film fixture holds the oracle's result twice (see tools/make_golden.py): `film` with the host's libm as the
reference binary calls it (glibc's float functions), `film_exact` with correctly rounded libm calls -- the arithmetic the
device implements. The device is held (almost) exactly to the second and to BASELINE.json's tolerance to the first.

CPU part: the oracle still reproduces them bit for bit (regression pin of the checker itself; its agreement
with the reference is pinned in test_oracle_pins.py by the reference's own statistics and known answers).
GPU part: the HIP path against the same vectors at the BASELINE sizes, rendered as FULL frames -- killeroo 700x700 at
1024 spp (configs[1]), the Cornell glass scene 512x512 at 4096 spp (configs[2]), the 10 000 002-triangle procedural
scene 700x700 at 256 spp (configs[3] stand-in) -- without CPU minutes on the GPU box.
"""
import os
import sys

import numpy as np
import pytest

from conftest import KILLEROO, CORNELL, ROOT

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COUNTER_KEYS = ("regular_rays", "shadow_rays", "total_paths", "zero_radiance_paths", "path_length_sum")


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    names = [str(n) for n in z["counter_names"]] if "counter_names" in z else []
    counters = dict(zip(names, [int(v) for v in z["counters"]])) if "counters" in z else None
    exact = dict(zip(names, [int(v) for v in z["counters_exact"]])) if "counters_exact" in z else None
    return z, counters, exact


def _rel_l2(a, b):
    d = a.astype(np.float64) - b
    return float(np.sqrt((d ** 2).sum() / max((b.astype(np.float64) ** 2).sum(), 1e-30)))


@pytest.mark.parametrize("name,scene", [("killeroo_1024spp_crop.npz", KILLEROO), ("cornell_256spp_crop.npz", CORNELL)])
def test_oracle_reproduces_the_golden_films(pt, ob, name, scene):
    z, counters, exact = _load(name)
    s = pt.Scene(scene, spp=int(z["spp"]), crop=tuple(float(v) for v in z["crop"]))
    film, weight, c, _ = ob.render(s)
    assert np.array_equal(film, z["film"]) and np.array_equal(weight, z["weight"])
    assert c.as_dict() == counters
    with ob.exact_libm():
        film, weight, c, _ = ob.render(s)
    assert np.array_equal(film, z["film_exact"]) and np.array_equal(weight, z["weight"])
    assert c.as_dict() == exact
    # the two libm evaluations of the same algorithm: a different path in O(1e-4) of the samples (see test_gpu_parity.py)
    for k in COUNTER_KEYS:
        assert abs(counters[k] - exact[k]) <= 1e-4 * exact[k] + 3, k
    assert _rel_l2(z["film"], z["film_exact"]) < 1e-3


def test_oracle_reproduces_the_golden_rays(pt, ob):
    z, _, _ = _load("killeroo_rays.npz")
    s = pt.Scene(KILLEROO, spp=1)
    closest, _ = ob.trace(s, z["rays"], any_hit=False)
    anyhit, _ = ob.trace(s, z["rays"], any_hit=True)
    assert np.array_equal(closest.view(np.int32), z["closest"])
    assert np.array_equal(anyhit.view(np.int32)[:, 0], z["anyhit"])


@pytest.mark.gpu
def test_gpu_traversal_matches_the_golden_rays_bit_exactly(pt):
    z, _, _ = _load("killeroo_rays.npz")
    integ = pt.CreatePathIntegrator(pt.Scene(KILLEROO, spp=1))
    assert np.array_equal(integ.trace(z["rays"], any_hit=False).view(np.int32), z["closest"])
    assert np.array_equal(integ.trace(z["rays"], any_hit=True).view(np.int32)[:, 0], z["anyhit"])


def _check_against_fixture(film, weight, z, spp, tol, frac_over, max_over, exact_rel=1e-6, exact_max=2e-4):
    """film / weight: the device's result on the fixture's pixels. Exact-libm oracle: image relative L2 < 1e-6 and every
    pixel within 2e-4 of the mean radiance (float accumulation order only); glibc-libm oracle: BASELINE's target --
    image relative L2 < tol, at most `frac_over` of the pixels above 1e-3 x mean (absolute: 1e-3 x mean radiance per
    sample), none above `max_over` x mean."""
    assert np.array_equal(weight, z["weight"])
    gx = z["film_exact"]
    mean = gx.mean() / spp
    px = np.sqrt(((film.astype(np.float64) - gx) ** 2).mean(axis=-1)) / spp
    assert _rel_l2(film, gx) < exact_rel, _rel_l2(film, gx)
    assert px.max() < exact_max * mean, px.max() / mean
    gold = z["film"]
    pg = np.sqrt(((film.astype(np.float64) - gold) ** 2).mean(axis=-1)) / spp
    assert _rel_l2(film, gold) < tol, _rel_l2(film, gold)
    over = pg > 1e-3 * mean
    assert over.mean() <= frac_over, (int(over.sum()), over.size)
    assert pg.max() < max_over * mean, pg.max() / mean


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,tol,frac_over,max_over", [
    ("killeroo_1024spp_crop.npz", KILLEROO, 1e-4, 2e-4, 1e-2),      # BASELINE configs[1]
    ("cornell_256spp_crop.npz", CORNELL, 1e-3, 5e-2, 1.0),
    ("cornell_4096spp_crop.npz", CORNELL, 3e-3, 5e-2, 1.0),         # BASELINE configs[2]; the crop sits on the caustic, where a
])                                                                  # diverged path carries many times the mean radiance
def test_gpu_full_frames_match_the_golden_films(pt, name, scene, tol, frac_over, max_over):
    """The FULL frame at the BASELINE resolution and sample count on the device; the fixture's window of it against the
    oracle's crop-window render (same Halton indexing: the sampler follows the full sample bounds, halton.cpp:75-85)."""
    z, counters, exact = _load(name)
    spp = int(z["spp"])
    s = pt.Scene(scene, spp=spp)
    w, h = s.film_size
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    c = integ.counters.as_dict()
    assert c["camera_rays"] == w * h * spp and c["bad_samples"] == 0
    sc = pt.Scene(scene, spp=spp, crop=tuple(float(v) for v in z["crop"]))
    x0, y0, x1, y1 = [int(v) for v in sc.desc.film.cropped_bounds]   # Film::croppedPixelBounds, film.cpp:56-62
    assert (y1 - y0, x1 - x0) == z["weight"].shape
    _check_against_fixture(film[y0:y1, x0:x1], weight[y0:y1, x0:x1], z, spp, tol, frac_over, max_over)
    assert (weight == spp).all()   # box filter: every pixel holds exactly its own samples
    # and the crop-window render itself: its counters against the oracle's
    ic = pt.CreatePathIntegrator(sc)
    fc, wc = ic.Render()
    cc = ic.counters.as_dict()
    assert cc["camera_rays"] == exact["camera_rays"]
    for k in COUNTER_KEYS:
        assert abs(cc[k] - exact[k]) <= 2, (k, cc[k], exact[k])                      # correctly rounded libm: the same paths
        assert abs(cc[k] - counters[k]) <= 1e-4 * counters[k] + 3, (k, cc[k], counters[k])   # glibc libm
    _check_against_fixture(fc, wc, z, spp, tol, frac_over, max_over)


def _textured_scene(pt, tmp_path, z):
    import scenes_text as st
    st.write_texture_files(str(tmp_path))
    st.write_alpha_png(str(tmp_path))
    s = pt.Scene(text=st.textured_zoo(res=int(z["res"]), spp=int(z["spp"])), base_dir=str(tmp_path))
    assert s.errors == []
    return s


def test_oracle_reproduces_the_textured_golden(pt, ob, tmp_path):
    z, counters, exact = _load("textured_zoo_64spp.npz")
    film, weight, c, _ = ob.render(_textured_scene(pt, tmp_path, z))
    assert np.array_equal(film, z["film"]) and np.array_equal(weight, z["weight"]) and c.as_dict() == counters
    with ob.exact_libm():
        film, weight, c, _ = ob.render(_textured_scene(pt, tmp_path, z))
    assert np.array_equal(film, z["film_exact"]) and c.as_dict() == exact


@pytest.mark.gpu
def test_gpu_film_matches_the_textured_golden(pt, tmp_path):
    z, counters, exact = _load("textured_zoo_64spp.npz")
    integ = pt.CreatePathIntegrator(_textured_scene(pt, tmp_path, z))
    film, weight = integ.Render()
    c = integ.counters.as_dict()
    assert c["camera_rays"] == counters["camera_rays"]
    for k in COUNTER_KEYS:
        assert abs(c[k] - exact[k]) <= 2, k
        assert abs(c[k] - counters[k]) <= 1e-4 * counters[k] + 3, k
    _check_against_fixture(film, weight, z, int(z["spp"]), 1e-4, 2e-3, 0.05)


# ------------------------------------------------------------------ BASELINE configs[3] stand-in at full size
def _procedural_scene(pt, tmp_path, z):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden as mg
    s = pt.Scene(mg.procedural_scene(str(tmp_path)))
    assert s.stats["n_triangles"] == int(z["n_triangles"]) == 10_000_002
    assert s.stats["interior_nodes"] == int(z["interior_nodes"])   # the host BVH is the one the fixture was traced in
    return s


@pytest.mark.gpu
def test_gpu_procedural_10M_triangles_256spp_full_size(pt, tmp_path):
    """The seeded 10 000 002-triangle scene at 700x700, 256 spp: a 0.6 GB BVH that does not fit the caches, rays that use
    the scratch part of the traversal stack. Recorded rays bit-equal to the oracle's; the full frame with exact camera-ray
    count and no bad samples; every 64th film tile against the oracle's film of it (both libm modes); three tile shards
    adding up to the full frame."""
    z, counters, exact = _load("procedural_10M_256spp.npz")
    s = _procedural_scene(pt, tmp_path, z)
    spp = int(z["spp"])
    assert s.spp == spp and s.film_size == (700, 700)
    integ = pt.CreatePathIntegrator(s)
    # recorded rays: hit primitive, t and barycentrics bitwise, closest-hit and any-hit
    assert np.array_equal(integ.trace(z["rays"], any_hit=False).view(np.int32), z["closest"])
    assert np.array_equal((integ.trace(z["rays"], any_hit=True).view(np.int32)[:, 0] >= 0), z["anyhit"] >= 0)
    # the full frame
    full, wfull = integ.Render()
    c = integ.counters.as_dict()
    assert c["camera_rays"] == 700 * 700 * spp and c["bad_samples"] == 0 and (wfull == spp).all()
    ys, xs = z["ys"].astype(int), z["xs"].astype(int)
    n_sc = int(z["shard_count"])
    # shard 0 of 64 alone: counters and film against the oracle's
    f0, w0 = integ.Render(shard_index=0, shard_count=n_sc)
    c0 = integ.counters.as_dict()
    assert c0["camera_rays"] == exact["camera_rays"] == counters["camera_rays"]
    for k in COUNTER_KEYS:
        assert abs(c0[k] - exact[k]) <= 2, (k, c0[k], exact[k])
        assert abs(c0[k] - counters[k]) <= 1e-4 * counters[k] + 3, (k, c0[k], counters[k])
    mask = np.zeros(w0.shape, bool)
    mask[ys, xs] = True
    assert not w0[~mask].any() and not f0[~mask].any()
    zz = {"film": z["film"], "film_exact": z["film_exact"], "weight": z["weight"]}
    _check_against_fixture(f0[ys, xs], w0[ys, xs], zz, spp, 1e-4, 2e-3, 0.05)
    # ... and the same tiles of the full frame, away from the pixels that a neighbouring tile's border samples also reach
    # (the shard only changes which tiles a pool draws)
    inner = (w0[ys, xs] == spp) & (wfull[ys, xs] == spp)
    assert inner.mean() > 0.9
    assert _rel_l2(full[ys, xs][inner], z["film_exact"][inner]) < 1e-6
    # three shards partition the frame
    acc, accw, cams = np.zeros_like(full), np.zeros_like(wfull), 0
    for r in range(3):
        f, w = integ.Render(shard_index=r, shard_count=3)
        acc += f
        accw += w
        cams += integ.counters.camera_rays
    assert cams == c["camera_rays"] and np.array_equal(accw, wfull)
    assert _rel_l2(acc, full) < 1e-6
