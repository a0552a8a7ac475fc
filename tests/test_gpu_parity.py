"""Parity of the HIP path (through the C ABI) against the CPU oracle. Run on the GPU box
with `pytest -m gpu`.

Two comparisons, because the reference's result depends on the C library: it calls std::sin(float), std::acos(float),
... whose last bit glibc rounds faithfully but not always to nearest, and a path amplifies such a bit (a refraction
direction one ulp off, a Russian-roulette test against a throughput of 1.0000001) into a different path in O(1e-4) of the
samples (tools/diag/path_compare.py shows them vertex by vertex).

 * EXACT: the oracle with correctly rounded libm calls (`ob.exact_libm()`, oracle/o_math.h mode 1) -- the arithmetic the
   device implements (d_math.h). Here device and oracle take the same decisions in every path: counters are compared
   (almost) exactly and films to float accumulation order. A difference in this mode is a defect.
 * REFERENCE LIBM: the oracle as the reference binary runs on this host (glibc's float functions; the mode in which the
   oracle reproduces the reference's own counters exactly, tests/test_oracle_pins.py). The device is held to
   BASELINE.json's target against it: per-pixel L2 < 1e-3 of the mean radiance, image relative L2 < 1e-4, at the
   BASELINE sample counts (tests/test_golden.py, and the killeroo tests below).

Integer results (hit primitive, camera-ray count, filter weights) are compared exactly in both.
The measured numbers of every comparison are written to gpurun_out/parity_metrics.json.
"""
import atexit
import json
import os

import numpy as np
import pytest

from conftest import KILLEROO, CORNELL, ROOT
import scenes_text as st
import trace_check as tc

pytestmark = pytest.mark.gpu

METRICS = []


@atexit.register
def _dump_metrics():
    if METRICS:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "parity_metrics.json"), "w") as fh:
                json.dump(METRICS, fh, indent=1)
        except OSError:
            pass


def _rel_l2(a, b):
    d = a.astype(np.float64) - b
    return float(np.sqrt((d ** 2).sum() / max((b.astype(np.float64) ** 2).sum(), 1e-30)))


def _pixel_l2(a, b, spp):
    d = a.astype(np.float64) - b
    return np.sqrt((d ** 2).mean(axis=2)) / spp


COUNTER_KEYS = ("regular_rays", "shadow_rays", "total_paths", "zero_radiance_paths", "path_length_sum")


def _check_counters(c, o, tol=1e-4, slack=3):
    assert c["camera_rays"] == o["camera_rays"]
    for k in COUNTER_KEYS:
        assert abs(c[k] - o[k]) <= tol * o[k] + slack, (k, c[k], o[k])
    assert c["bad_samples"] == o["bad_samples"] == 0


def _parity(pt, ob, s, name, exact=True, rel_tol=None, counter_tol=None, weights_exact=True):
    """Render `s` on the device and with the oracle, record the measured differences, assert the bar of the mode.
    EXACT mode bar: counters equal (2 counts of slack: the only arithmetic not shared is DivBy's reciprocal form, one ulp in
    2^-22 of the quotients, d_math.h; measured difference: 0 in every test), image relative L2 < 1e-6 (float atomics:
    accumulation order; measured <= 2.5e-7), every pixel within 2e-4 x mean radiance (measured <= 4e-5; the target is 1e-3).
    REFERENCE-LIBM bar: what the caller passes (the BASELINE target at BASELINE sample counts)."""
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    if exact:
        with ob.exact_libm():
            ofilm, oweight, oc, _ = ob.render(s)
    else:
        ofilm, oweight, oc, _ = ob.render(s)
    c, o = integ.counters.as_dict(), oc.as_dict()
    spp = max(1, s.spp)
    l2 = _pixel_l2(film, ofilm, spp)
    mean = max(float(ofilm.mean()) / spp, 1e-12)
    m = {"test": name, "mode": "exact" if exact else "reference-libm", "rel_l2": _rel_l2(film, ofilm),
         "pixels": int(l2.size), "pixels_over_1e-3_mean": int((l2 > 1e-3 * mean).sum()), "max_pixel_l2_over_mean": float(l2.max() / mean),
         "counter_diff": {k: int(c[k] - o[k]) for k in COUNTER_KEYS}, "counters": {k: int(o[k]) for k in COUNTER_KEYS},
         "camera_rays": int(o["camera_rays"])}
    METRICS.append(m)
    assert not np.isnan(film).any()
    if weights_exact:
        assert np.array_equal(weight, oweight)
    else:
        assert np.allclose(weight, oweight, rtol=1e-5, atol=1e-6)
    if exact:
        _check_counters(c, o, tol=counter_tol if counter_tol is not None else 0.0, slack=2)
        assert m["rel_l2"] < (rel_tol if rel_tol is not None else 1e-6), m
        assert m["max_pixel_l2_over_mean"] < 2e-4, m
    else:
        _check_counters(c, o, tol=counter_tol if counter_tol is not None else 1e-4)
        assert m["rel_l2"] < (rel_tol if rel_tol is not None else 1e-4), m
    return film, weight, integ, ofilm, oweight, oc


def test_traversal_kernel_matches_oracle_bit_exactly(pt, ob):
    """BVH2 traversal on recorded rays: same primitive, same t and barycentrics (bitwise),
    closest-hit and any-hit, coherent camera rays and incoherent random rays -- answered by the per-ray routine of
    mi_pt_trace (k_trace) AND by the kernels the render launches (mi_pt_trace_wavefront: k_trav<0>, <1>, <2> + resolve)."""
    s = pt.Scene(KILLEROO, spp=1)
    integ = pt.CreatePathIntegrator(s)
    rng = np.random.default_rng(5)
    samples = np.stack([rng.integers(0, 700, 200000), rng.integers(0, 700, 200000), np.zeros(200000, int)], axis=1)
    cam = ob.camera_rays(s, samples)
    n = 200000
    o = rng.uniform(-300, 300, (n, 3)).astype(np.float32) + np.array([0, 60, -100], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(10, 400, n)).astype(np.float32)
    rnd = np.concatenate([o, d, tmax[:, None]], axis=1)
    for rays in (cam, rnd):
        for any_hit in (False, True):
            want, _ = ob.trace(s, rays, any_hit=any_hit)
            got = integ.trace(rays, any_hit=any_hit)
            assert np.array_equal(got.view(np.int32), want.view(np.int32))
        hits, extra = tc.check_wavefront(integ, rays, *tc.oracle_answers(ob, s))
        # the sphere light is a postponed quadric for the rays that meet it (I_NPEND as k_trav left it)
        assert ((extra[:, 2] & 0xff) > 0).any() and ((extra[:, 2] & 0x100) == 0).all()
    assert (ob.trace(s, cam)[0].view(np.int32)[:, 0] >= 0).mean() > 0.5


def _stacked_sheets_scene(pt, seed=11):
    """Triangles that tie: a 6 x 6 sheet of quads, each present four times in ONE mesh -- twice at z = 5 exactly (coplanar
    duplicates: equal t, the tie goes by leaf order) and once each an ulp or two in front of and behind it (two hits of one
    leaf inside the tMax-dependent rejection of triangle.cpp:262-266). Equal centroids end in one leaf (bvh.cpp:289-300)."""
    rng = np.random.default_rng(seed)
    P, idx = [], []
    ulp = float(np.spacing(np.float32(5)))
    for cy in range(6):
        for cx in range(6):
            x0, y0 = -3 + cx, -3 + cy
            for dz in (0.0, 0.0, ulp * int(rng.integers(1, 3)), -ulp * int(rng.integers(1, 3))):
                b = len(P)
                P += [(x0, y0, 5 + dz), (x0 + 1, y0, 5 + dz), (x0 + 1, y0 + 1, 5 + dz), (x0, y0 + 1, 5 + dz)]
                idx += [b, b + 1, b + 2, b, b + 2, b + 3]
    body = ('Shape "trianglemesh" "integer indices" [%s] "point P" [%s]\n'
            % (" ".join(map(str, idx)), " ".join(repr(float(np.float32(v))) for p in P for v in p)))
    return pt.Scene(text=st._HEAD % dict(res=8, spp=1, depth=1, extra="") + 'LightSource "point" "rgb I" [1 1 1]\n' + body + "WorldEnd\n")


@pytest.mark.parametrize("env", [{}, {"MIPT_BVH_WIDTH": "2"}, {"MIPT_NO_COOP_LEAVES": "1"}])
def test_ties_and_near_ties_inside_a_leaf(pt, ob, monkeypatch, env):
    """BVHAccel::Intersect's sequential leaf loop (bvh.cpp:676-682) decides ties by order and lets a second hit through only
    if Triangle::Intersect's tScaled test passes against the tMax the first one left: k_trav tests a leaf's triangles
    side by side (cooperative test) and replays that sequence. Bit-equal hit records against the oracle for the
    two-level records, the BVH2 records (W = 2) and the one-primitive-per-pass leaf path."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = _stacked_sheets_scene(pt)
    assert s.errors == [] and s.stats["n_triangles"] == 6 * 6 * 4 * 2
    leaves = [int(s.desc.nodes[i].n_prims) for i in range(s.desc.n_nodes) if s.desc.nodes[i].n_prims > 0]
    assert max(leaves) >= 4   # (several copies of a triangle share a leaf)
    integ = pt.CreatePathIntegrator(s)
    rng = np.random.default_rng(17)
    n = 60000
    o = np.stack([rng.uniform(-4, 4, n), rng.uniform(-4, 4, n), np.where(rng.random(n) < .5, rng.uniform(-2, 4.5, n), rng.uniform(5.5, 9, n))], -1).astype(np.float32)
    tgt = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), np.full(n, 5.0)], -1).astype(np.float32)
    d = tgt - o
    norm = rng.random(n) < .5
    d[norm] = (d[norm] / np.linalg.norm(d[norm].astype(np.float64), axis=1)[:, None]).astype(np.float32)
    tmax = np.where(rng.random(n) < .7, np.inf, rng.uniform(.5, 12, n)).astype(np.float32)
    rays = np.concatenate([o, d, tmax[:, None]], axis=1).astype(np.float32)
    closest, occluded = tc.oracle_answers(ob, s)
    want = closest(rays)
    assert np.array_equal(integ.trace(rays).view(np.int32), want.view(np.int32))
    hits, extra = tc.check_wavefront(integ, rays, closest, occluded)
    prim = hits.view(np.int32)[:, 0]
    assert (prim >= 0).mean() > .5 and len(np.unique(prim[prim >= 0])) > 100   # (the winners are spread over the copies)


def test_axis_parallel_rays_inside_slab_planes(pt, ob):
    """Bounds3::IntersectP with a zero direction component and the origin IN a slab plane: (plane - o) * invDir = 0 * inf = NaN,
    and the NaN walks through the reference's comparisons (geometry.h:1420-1447: every test with it is false, so tMin stays
    NaN and the box is missed at `tMin < ray.tMax`). k_trav's box test for finite reciprocals (BoxTestFast: minima and
    maxima) must not see such a ray: a wave that holds one takes the select-and-compare test. Axis-parallel rays from
    integer coordinates through the stacked-sheets scene (quad corners, hence node planes, at integers), shuffled among
    ordinary rays so that waves mix both kinds: hit records bit-equal to the oracle's."""
    s = _stacked_sheets_scene(pt)
    integ = pt.CreatePathIntegrator(s)
    rng = np.random.default_rng(23)
    n = 40000
    o = np.stack([rng.uniform(-4, 4, n), rng.uniform(-4, 4, n), np.where(rng.random(n) < .5, rng.uniform(-2, 4.5, n), rng.uniform(5.5, 9, n))], -1).astype(np.float32)
    tgt = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), np.full(n, 5.0)], -1).astype(np.float32)
    d = tgt - o
    kind = rng.integers(0, 6, n)
    grid = rng.integers(-3, 4, (n, 3)).astype(np.float32)
    for k, zero in ((1, [0]), (2, [1]), (3, [0, 1])):   # d.x = 0 / d.y = 0 / both, the origin on an integer plane of that axis
        m = kind == k
        for a in zero:
            d[m, a] = 0.0
            o[m, a] = grid[m, a]
    m = kind == 4                                        # in the sheets' own plane: d.z = 0 at z = 5
    d[m, 2] = 0.0
    o[m, 2] = 5.0
    assert (d == 0).any(axis=1).mean() > .5
    tmax = np.where(rng.random(n) < .7, np.inf, rng.uniform(.5, 12, n)).astype(np.float32)
    rays = np.concatenate([o, d, tmax[:, None]], axis=1).astype(np.float32)
    closest, occluded = tc.oracle_answers(ob, s)
    want = closest(rays)
    assert np.array_equal(integ.trace(rays).view(np.int32), want.view(np.int32))
    hits, extra = tc.check_wavefront(integ, rays, closest, occluded)
    prim = hits.view(np.int32)[:, 0]
    zero = (d == 0).any(axis=1)
    assert (prim[zero] >= 0).any() and (prim[zero] < 0).any() and (prim[~zero] >= 0).any()
    # (straight up from an integer (x, y): inside two slab planes of every box on the way, missed by the reference as well)
    assert (prim[(d[:, :2] == 0).all(axis=1)] < 0).all()


def test_killeroo_full_size_low_spp_against_oracle(pt, ob):
    """700x700, 4 spp (the size of BASELINE configs 1-2): counters, weights, film."""
    s = pt.Scene(KILLEROO, spp=4)
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "killeroo 700x700 4spp")
    assert integ.counters.camera_rays == 700 * 700 * 4
    # and against the oracle as the reference binary runs here (glibc libm): at 4 spp one diverged path moves a pixel by
    # ~L/4, so the per-pixel target is met by all but a handful of pixels (measured 2e-5 of them; 1024 spp: test_golden.py)
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "killeroo 700x700 4spp", exact=False)
    l2 = _pixel_l2(film, ofilm, 4)
    mean = ofilm.mean() / 4
    assert (l2 > 1e-3 * mean).mean() < 2e-4
    assert np.median(l2) < 1e-6 * mean


def test_killeroo_64spp_crop_meets_the_l2_target(pt, ob):
    """BASELINE target: per-pixel L2 < 1e-3 (relative to the mean radiance). Checked at
    64 spp on a 175x175 crop of the 700x700 frame (same Halton indexing as the full
    frame, see SURVEY 8c travel rule); the error shrinks with spp."""
    crop = (0.375, 0.625, 0.5, 0.75)
    s = pt.Scene(KILLEROO, spp=64, crop=crop)
    assert s.film_size == (175, 175)
    _parity(pt, ob, s, "killeroo 64spp crop")
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "killeroo 64spp crop", exact=False)
    l2 = _pixel_l2(film, ofilm, 64)
    mean = ofilm.mean() / 64
    # pixels above the per-pixel target against the glibc-libm oracle: measured 1.2e-3 of this crop at 64 spp (its mean
    # radiance is 1/16 of the frame's; 1.4e-5 of the full frame), none of them far: a diverged path is one sample of 64
    assert (l2 > 1e-3 * mean).mean() < 3e-3
    assert l2.max() < 0.05 * mean
    assert l2.mean() < 1e-5 * mean


def test_sample_ranges_accumulate_to_the_same_film(pt, ob):
    """Passes over sample ranges [0,8) + [8,16) accumulate to the 16-spp film
    (linearity in the sample index, the property bench.py's steps rely on)."""
    s = pt.Scene(KILLEROO, spp=16, xres=128, yres=128)
    integ = pt.CreatePathIntegrator(s)
    full, wfull = integ.Render()
    integ.Render(spp=8, sample_begin=0, download=False)
    part, wpart = integ.Render(spp=8, sample_begin=8, accumulate=True)
    assert np.array_equal(wfull, wpart)
    assert _rel_l2(part, full) < 1e-6   # only the atomic accumulation order differs


def test_tile_shards_partition_the_render(pt, ob):
    """Film tiles split over 3 shards: camera rays add up, films add up, no pixel is
    written by two shards (the multi-GPU decomposition)."""
    s = pt.Scene(KILLEROO, spp=4, xres=160, yres=160)
    integ = pt.CreatePathIntegrator(s)
    full, wfull = integ.Render()
    cam_full = integ.counters.camera_rays
    acc, accw, cams = np.zeros_like(full), np.zeros_like(wfull), 0
    for r in range(3):
        f, w = integ.Render(shard_index=r, shard_count=3)
        assert ((w != 0) & (accw != 0)).sum() <= 0.01 * w.size   # only pixel-border samples are shared
        acc += f
        accw += w
        cams += integ.counters.camera_rays
    assert cams == cam_full and np.array_equal(accw, wfull)
    assert _rel_l2(acc, full) < 1e-6


@pytest.mark.parametrize("strategy", ["spatial", "power", "uniform"])
def test_material_zoo_glass_uber_disney_all_light_types(pt, ob, strategy):
    """Every material of the hot path (matte/Oren-Nayar, plastic, glass smooth+rough, uber
    with opacity, disney thick+thin, mirror, metal, substrate, translucent, mix), area + point +
    distant + spot lights, smooth normals, and the three light-selection strategies."""
    s = pt.Scene(text=st.material_zoo(res=96, spp=32, depth=6, strategy=strategy))
    assert s.errors == []
    _parity(pt, ob, s, "material zoo " + strategy)
    # the per-voxel light tables behind "spatial" are the oracle's bit for bit
    if strategy == "spatial":
        integ = pt.CreatePathIntegrator(s)
        dfunc, dfint = integ.light_distribution()
        ofunc, ofint = ob.light_table(s)
        assert np.array_equal(dfunc, ofunc, equal_nan=True) and np.array_equal(dfint, ofint, equal_nan=True)
    # against the glibc-libm oracle a 96x96x32 render holds ~30 diverged samples: image L2 stays below 2e-4
    _parity(pt, ob, s, "material zoo " + strategy, exact=False, rel_tol=2e-4, counter_tol=1e-4)


def test_cornell_glass_sphere(pt, ob):
    """BASELINE config 3 scene at test size: dielectric sphere, two-triangle area light
    (spatial light distribution over 2 lights), maxdepth 8."""
    s = pt.Scene(CORNELL, spp=16, xres=128, yres=128)
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "cornell glass 128x128 16spp")
    dfunc, dfint = integ.light_distribution()
    ofunc, ofint = ob.light_table(s)
    assert np.array_equal(dfunc, ofunc, equal_nan=True) and np.array_equal(dfint, ofint, equal_nan=True)
    # glibc-libm oracle: the glass sphere turns a last-bit difference of acosf / sinf in Sphere::Intersect into another
    # path inside the sphere (2.7e-4 of the samples, each one sample of 16 in its pixel)
    _parity(pt, ob, s, "cornell glass 128x128 16spp", exact=False, rel_tol=2e-3, counter_tol=1e-3)


@pytest.mark.parametrize("sampler", ["halton", "sobol", "random"])
@pytest.mark.parametrize("text", [st.furnace_point(), st.furnace_point(n_lights=4), st.furnace_area(), st.furnace_uber()])
def test_furnace_scenes_on_gpu(pt, text, sampler):
    """The reference's known answers on the device itself: the scenes of tests/analytic_scenes.cpp:71-203 under the
    samplers of its matrix (:250-267) that this path builds -- radiance 1 everywhere, to the reference's 2 %."""
    assert 'Sampler "halton"' in text
    s = pt.Scene(text=text.replace('Sampler "halton"', 'Sampler "%s"' % sampler))
    assert s.errors == []
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    assert abs(float((film / weight[..., None]).mean()) - 1.0) < 0.02


def test_edge_cases_empty_scene_no_lights_crop_filter(pt, ob):
    head = st._HEAD % dict(res=16, spp=2, depth=3, extra="")
    # no primitives at all, no lights: black film, weights still accumulate
    s = pt.Scene(text=head + "WorldEnd\n")
    film, weight = pt.CreatePathIntegrator(s).Render()
    assert not film.any() and (weight > 0).all()
    # geometry but no light
    s = pt.Scene(text=head + 'Shape "sphere" "float radius" [1]\nTranslate 0 0 3\nWorldEnd\n')
    film, weight = pt.CreatePathIntegrator(s).Render()
    assert not film.any()
    # wide filter + crop window: contributions cross pixel (and tile) borders
    txt = st.furnace_area(res=40, spp=4).replace('Sampler', 'PixelFilter "gaussian" "float xwidth" [2] "float ywidth" [2]\nSampler')
    txt = txt.replace('[40] "integer yresolution" [40]', '[40] "integer yresolution" [40] "float cropwindow" [.2 .8 .1 .9]')
    s = pt.Scene(text=txt)
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "gaussian filter + crop window", weights_exact=False)
    assert integ.counters.camera_rays == oc.camera_rays
    assert abs(float((film.sum(axis=(0, 1)) / weight.sum()).mean()) - 1.0) < 0.02


def test_create_rejects_malformed_descriptions(pt):
    import ctypes as C
    s = pt.Scene(text=st.furnace_area(res=8, spp=1))
    d = pt.SceneDesc.from_buffer_copy(s.desc)
    d.abi_version = 99
    h = C.c_void_p()
    assert pt.hip_lib().mi_pt_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"ABI" in pt.hip_lib().mi_pt_last_error()
    assert pt.hip_lib().mi_pt_create(s.desc_ptr, 12345, C.byref(h)) == -2
    # ABI v10 fields the shading kernels index with: a lobe rule that is none, a sigma map that is no texture, the glass switch on
    # a material whose lobe 0 is not the FresnelSpecular one -- refused, and the scene still renders once they are put back
    mats = (pt.Material * s.desc.n_materials).from_address(C.addressof(s.desc.materials.contents))
    m = next(x for x in mats if x.n_bxdfs > 0)
    for set_bad, undo, needle in ((lambda: setattr(m.tex[0], "rule", 42), lambda: setattr(m.tex[0], "rule", 0), b"mi_lobe_rule"),
                                  (lambda: setattr(m, "sigma_tex", 7), lambda: setattr(m, "sigma_tex", -1), b"sigma_tex"),
                                  (lambda: setattr(m, "rough_flags", 2), lambda: setattr(m, "rough_flags", 0), b"MI_ROUGH_GLASS")):
        set_bad()
        assert pt.hip_lib().mi_pt_create(s.desc_ptr, 0, C.byref(h)) == -1 and needle in pt.hip_lib().mi_pt_last_error()
        undo()
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    assert np.isfinite(film).all() and weight.sum() > 0


def test_killeroo_64spp_full_frame_reproduces_the_reference_counters(pt):
    """BASELINE config 1 at full size: the reference's own deterministic statistics for
    killeroo-simple at 64 spp (31 360 000 camera rays, 134 988 455 regular and
    49 247 871 shadow ray tests, mean path length 1.640, 15.93 % zero-radiance paths).
    Integer work: exact up to the few paths whose libm rounding flips a decision."""
    s = pt.Scene(KILLEROO, spp=64)
    integ = pt.CreatePathIntegrator(s)
    integ.Render(download=False)
    c = integ.counters.as_dict()
    assert c["camera_rays"] == 31_360_000
    assert abs(c["regular_rays"] - 134_988_455) <= 2e-5 * 134_988_455
    assert abs(c["shadow_rays"] - 49_247_871) <= 2e-5 * 49_247_871
    assert abs(c["path_length_sum"] / c["camera_rays"] - 1.640) < 1e-3
    assert abs(c["zero_radiance_paths"] / c["total_paths"] - 0.1593) < 1e-4
    assert c["bad_samples"] == 0


def test_result_does_not_depend_on_the_number_of_sub_renderers(pt, monkeypatch):
    """MIPT_STREAMS only changes how tiles are spread over concurrent path pools: the
    statistics are identical and the films differ by accumulation order alone."""
    s = pt.Scene(KILLEROO, spp=8, xres=200, yres=200)
    out = []
    for k in ("1", "3", "4"):
        monkeypatch.setenv("MIPT_STREAMS", k)
        integ = pt.CreatePathIntegrator(s)
        film, weight = integ.Render()
        out.append((film, weight, integ.counters.as_dict()))
    for film, weight, c in out[1:]:
        assert np.array_equal(weight, out[0][1])
        assert _rel_l2(film, out[0][0]) < 1e-6
        for k in ("camera_rays", "regular_rays", "shadow_rays", "total_paths", "zero_radiance_paths", "path_length_sum",
                  "bvh_nodes_visited", "tri_tests"):
            assert c[k] == out[0][2][k], k


def test_procedural_many_mesh_scene_against_oracle(pt, ob, tmp_path):
    """The BASELINE config 4/5 stand-in (tools/make_procedural_scene.py) at test size:
    626 small meshes, 8 named materials in 4 shading classes (matte, Oren-Nayar, plastic,
    uber), a distant light and 4 spherical area lights, spatial light sampling."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_procedural_scene as mps
    path = tmp_path / "proc.pbrt"
    with open(path, "w") as fh:
        mps.write_scene(fh, 200_000, 96, 16, 7, 5)
    s = pt.Scene(str(path))
    assert s.stats["n_triangles"] == 200_002 and s.stats["n_lights"] == 5
    _parity(pt, ob, s, "procedural 200k triangles 96x96 16spp")
    _parity(pt, ob, s, "procedural 200k triangles 96x96 16spp", exact=False, rel_tol=1e-3, counter_tol=5e-4)


@pytest.mark.parametrize("sampler,lens", [("sobol", False), ("sobol", True), ("random", False), ("random", True),
                                          ("02sequence", False), ("02sequence", True), ("stratified", False), ("stratified", True)])
def test_sobol_and_random_samplers_against_oracle(pt, ob, sampler, lens):
    """Sampler "sobol" (sobol.cpp, lowdiscrepancy.h:229-274: the sample of (pixel, n) is a pure function of its index, as
    with Halton) and Sampler "random" (one PCG32 stream per camera sample, include/mi_pt.h mi_sampler_type) on the material
    zoo, with and without a lens (dimensions 3 / 4; the random stream draws them whether or not there is a lens)."""
    txt = st.material_zoo(res=64, spp=16, depth=6).replace('Sampler "halton"', 'Sampler "%s"' % sampler)
    if lens:
        txt = txt.replace('Camera "perspective" "float fov" [40]', 'Camera "perspective" "float fov" [40] "float lensradius" [.15] "float focaldistance" [8]')
    if sampler == "stratified":   # (the pixel samplers: tables per pixel from the pixel's own stream, include/mi_pt.h mi_sampler_type)
        txt = txt.replace('"integer pixelsamples" [16]', '"integer xsamples" [8] "integer ysamples" [2] "integer dimensions" [3]')
    s = pt.Scene(text=txt)
    assert s.errors == [] and s.desc.sampler.type == {"sobol": 1, "random": 2, "02sequence": 3, "stratified": 4}[sampler] and s.spp == 16
    _parity(pt, ob, s, "zoo sampler %s lens=%s" % (sampler, lens))


def test_furnace_scenes_under_all_five_samplers_on_the_device(pt, ob):
    """The perspective half of tests/analytic_scenes.cpp:250-267 on the HIP path: the three furnace scenes under halton, sobol,
    random, 02sequence and stratified -- radiance 1 +- 0.02 (the reference's own pass mark), and equal to the oracle."""
    for sampler in ("halton", "sobol", "random", "02sequence", "stratified"):
        for text in (st.furnace_point(), st.furnace_area(), st.furnace_uber()):
            text = text.replace('Sampler "halton"', 'Sampler "%s"' % sampler)
            if sampler == "stratified":
                text = text.replace('"integer pixelsamples" [256]', '"integer xsamples" [16] "integer ysamples" [16]')
            s = pt.Scene(text=text)
            assert s.errors == [] and s.spp == 256
            film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "furnace %s" % sampler)
            assert abs(float((film / weight[..., None]).mean()) - 1.0) < 0.02, sampler


def test_pixel_sampler_passes_stay_inside_the_tables(pt):
    """A pixel sampler has tables for samples_per_pixel samples of every pixel: a pass beyond them is an error code."""
    s = pt.Scene(text=st.furnace_point(res=8, spp=16).replace('Sampler "halton"', 'Sampler "02sequence"'))
    integ = pt.CreatePathIntegrator(s)
    a, wa = integ.Render(spp=8, sample_begin=0)
    b, wb = integ.Render(spp=8, sample_begin=8, accumulate=True)
    full, wf = integ.Render()
    assert np.array_equal(wb, wf) and _rel_l2(b, full) < 1e-6   # two passes = the frame
    with pytest.raises(RuntimeError, match="beyond them"):
        integ.Render(spp=16, sample_begin=8)


def test_sobol_sampler_on_the_killeroo_frame(pt, ob):
    """700x700 (Sobol' resolution 1024, log2 10), 4 spp: exact parity on the BASELINE scene under the second global sampler."""
    import os
    text = open(KILLEROO).read().replace('Sampler "halton"', 'Sampler "sobol"')
    assert 'Sampler "sobol"' in text
    s = pt.Scene(text=text, base_dir=os.path.dirname(KILLEROO), spp=4)
    assert s.desc.sampler.sobol_resolution == 1024 and s.desc.sampler.sobol_log2_resolution == 10
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "killeroo sobol 4spp")
    assert integ.counters.camera_rays == 700 * 700 * 4


def _killeroo_spectralpath(pt, n_bands, **kw):
    import os
    text = open(KILLEROO).read().replace('Integrator "path"',
                                         'Integrator "spectralpath" "integer numCABands" [%d]' % n_bands)
    return pt.Scene(text=text, base_dir=os.path.dirname(KILLEROO), **kw)


@pytest.mark.parametrize("n_bands", [4, 3])
def test_spectralpath_bands_against_oracle(pt, ob, n_bands):
    """Integrator "spectralpath" (spectralpath.cpp:258-318): numCABands paths per camera sample on
    consecutive Halton dimensions, band s supplying bins [round(31/n)*s, min(round(31/n)*(s+1), 31))."""
    s = _killeroo_spectralpath(pt, n_bands, spp=8, xres=96, yres=96)
    assert s.desc.integrator.n_ca_bands == n_bands
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "spectralpath %d bands" % n_bands)
    c = integ.counters.as_dict()
    assert c["camera_rays"] == 96 * 96 * 8 * n_bands      # one camera ray per band; one film sample per camera sample (weights exact)
    if n_bands == 3:                                       # round(31/3) = 10: bin 30 is never assigned
        assert not film[..., 30].any() and film[..., 29].any()


def test_spectralpath_with_one_band_is_the_path_integrator(pt):
    """numCABands = 1 stitches bins [0, 31) of the only path: identical to Integrator "path"."""
    a = pt.Scene(KILLEROO, spp=4, xres=96, yres=96)
    b = _killeroo_spectralpath(pt, 1, spp=4, xres=96, yres=96)
    fa, wa = pt.CreatePathIntegrator(a).Render()
    fb, wb = pt.CreatePathIntegrator(b).Render()
    assert np.array_equal(wa, wb) and _rel_l2(fb, fa) < 1e-6


_ZOO_VARIANTS = {
    "thin_lens": [('Camera "perspective" "float fov" [40]', 'Camera "perspective" "float fov" [40] "float lensradius" [.15] "float focaldistance" [8]')],
    "maxdepth0": [('"integer maxdepth" [6]', '"integer maxdepth" [0]')],
    "maxdepth1": [('"integer maxdepth" [6]', '"integer maxdepth" [1]')],
    "rr_off": [('Integrator "path"', 'Integrator "path" "float rrthreshold" [0]')],
    "pixelbounds": [('Integrator "path"', 'Integrator "path" "integer pixelbounds" [10 40 5 30]')],
    "clamp_scale": [('Film "image"', 'Film "image" "float maxsampleluminance" [1.5] "float scale" [2.5]')],
    "mitchell": [('Sampler "halton"', 'PixelFilter "mitchell" "float xwidth" [1.5] "float ywidth" [2] "float B" [.4] "float C" [.3]\nSampler "halton"')],
    "sinc": [('Sampler "halton"', 'PixelFilter "sinc" "float xwidth" [3] "float ywidth" [3] "float tau" [2]\nSampler "halton"')],
    "triangle": [('Sampler "halton"', 'PixelFilter "triangle" "float xwidth" [1.2] "float ywidth" [.7]\nSampler "halton"')],
    "pixel_center": [('Sampler "halton"', 'Sampler "halton" "bool samplepixelcenter" "true"\n# ')],
}


@pytest.mark.parametrize("variant", sorted(_ZOO_VARIANTS))
def test_camera_film_filter_and_integrator_parameters(pt, ob, variant):
    """The parameters of the path's own plugins (perspective.cpp:148-189, film.cpp:311-352, path.cpp:190-213,
    filters/*.cpp, halton.cpp:133-140), one at a time on the material zoo."""
    txt = st.material_zoo(res=48, spp=8, depth=6)
    for a, b in _ZOO_VARIANTS[variant]:
        assert a in txt, a
        txt = txt.replace(a, b, 1)
    s = pt.Scene(text=txt)
    assert s.errors == []
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "zoo variant " + variant, weights_exact=False)
    if variant == "pixelbounds":
        assert not weight[:5].any() and not weight[:, :10].any() and weight[5:30, 10:40].all()
    if variant == "maxdepth0":
        assert integ.counters.shadow_rays == 0


@pytest.mark.parametrize("kind,strategy", [("const", "power"), ("map", "power"), ("map", "spatial"), ("only_env", "spatial")])
def test_infinite_area_light_against_oracle(pt, ob, tmp_path, kind, strategy):
    """LightSource "infinite" (infinite.cpp:43-141): constant and PFM-mapped (resampled to a power of two, rotated),
    Le for escaped camera / specular rays, light sampling through the Distribution2D, MIS with Pdf_Li, Le for escaped
    MIS rays, and its part in the power / spatial light-selection distributions."""
    st.write_env_pfm(str(tmp_path / "env.pfm"))
    s = pt.Scene(text=st.zoo_with_infinite_light(kind, strategy=strategy), base_dir=str(tmp_path))
    assert s.errors == []
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "infinite light %s %s" % (kind, strategy))
    if strategy == "spatial":   # (the tables include the environment light's estimates: Sample_Li through the Distribution2D)
        dfunc, dfint = integ.light_distribution()
        ofunc, ofint = ob.light_table(s)
        assert np.array_equal(dfunc, ofunc, equal_nan=True) and np.array_equal(dfint, ofint, equal_nan=True)
    assert film[:8].mean() > 0           # the sky is visible above the back wall: escaped camera rays see Le


@pytest.mark.parametrize("lens", [False, True])
def test_image_textures_against_oracle(pt, ob, tmp_path, lens):
    """Texture "imagemap" (imagemap.cpp, mipmap.h) on Kd / Ks / Kr / Kt of matte, plastic, uber, substrate, mirror,
    translucent and glass: PNG / TGA / PFM pyramids built by the host, EWA (default), trilinear and unfiltered lookups from
    the hit's (u, v) and the camera ray's differentials (with and without a lens), wrap modes, uv scale / offset, and
    lobes that leave the BSDF where their texture is black."""
    st.write_texture_files(str(tmp_path))
    s = pt.Scene(text=st.textured_zoo(res=64, spp=16, lens=lens), base_dir=str(tmp_path))
    assert s.errors == []
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "textured zoo lens=%s" % lens)
    # the textures show: the ground's chequer pattern makes neighbouring pixels differ far more than noise would
    ground = film[40:60, 8:56].sum(axis=2)
    assert ground.std() > 0.2 * ground.mean()


def test_alpha_masks_against_oracle(pt, ob, tmp_path):
    """ "alpha" / "shadowalpha" textures of triangle meshes (triangle.cpp:331-338, 531-570): hits where the mask is exactly 0
    do not count -- in the closest-hit traversal ("alpha"), and in the shadow traversal ("alpha" and "shadowalpha")."""
    st.write_alpha_png(str(tmp_path))
    s = pt.Scene(text=st.alpha_scene(), base_dir=str(tmp_path))
    assert s.errors == []
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "alpha masks")
    # recorded rays through the cut-out panel: same hits on both sides, closest-hit and any-hit
    rng = np.random.default_rng(3)
    n = 4000
    o = np.tile(np.array([0, 2.5, -7], np.float32), (n, 1))
    tgt = np.stack([rng.uniform(-3.2, 3.2, n), rng.uniform(0.2, 2.6, n), rng.uniform(-0.5, 1.0, n)], -1).astype(np.float32)
    dirs = tgt - o
    rays = np.concatenate([o, dirs, np.full((n, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    for any_hit in (False, True):
        dh = integ.trace(rays, any_hit=any_hit)
        oh, _ = ob.trace(s, rays, any_hit=any_hit)
        if any_hit:
            assert np.array_equal(dh[:, 0].view(np.int32) >= 0, oh[:, 0].view(np.int32) >= 0)
        else:
            assert np.array_equal(dh[:, 0].view(np.int32), oh[:, 0].view(np.int32))
    tc.check_wavefront(integ, rays, *tc.oracle_answers(ob, s), prim_only=True)   # k_trav<*, ALPHA = true>
    tc.check_wavefront(integ, rays, *tc.device_answers(integ))                    # ... and bit for bit what k_trace answers


def test_bump_mapping_against_oracle(pt, ob, tmp_path):
    """ "texture bumpmap" (Material::Bump, material.cpp:47-84): three displaced lookups per vertex, the bumped shading
    frame from shading.dpdu / dpdv / dndu / dndv (triangle.cpp:347-413), on meshes with and without normals, mirrored, and
    together with a textured Kd."""
    st.write_texture_files(str(tmp_path))
    s = pt.Scene(text=st.bump_scene(), base_dir=str(tmp_path))
    assert s.errors == []
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "bump mapping")
    # and the bump map does something: without it the render differs visibly
    flat = pt.Scene(text=st.bump_scene().replace('"texture bumpmap" "bumps_tri"', "").replace('"texture bumpmap" "bumps"', ""), base_dir=str(tmp_path))
    ff, _, _, _ = ob.render(flat)
    assert _rel_l2(ff, ofilm) > 0.02


def test_texture_lookups_bit_for_bit(pt, ob, tmp_path):
    """MIPMap::Lookup on the device against the oracle, query by query (mi_pt_texture_lookup / oracle_texture_lookup): EWA with
    random anisotropic footprints, trilinear, unfiltered, the three wrap modes, coordinates outside [0, 1]. Equal bits against
    the oracle with a correctly rounded logf; against glibc's logf the level of detail differs in the last bit now and then."""
    st.write_texture_files(str(tmp_path))
    s = pt.Scene(text=st.textured_zoo(res=16, spp=1), base_dir=str(tmp_path))
    assert s.errors == []
    integ = pt.CreatePathIntegrator(s)
    rng = np.random.default_rng(11)
    n = 3000
    for tex in range(s.desc.n_textures):
        if s.desc.textures[tex].type != 0:   # checkerboards have no pyramid
            continue
        q = np.zeros((n, 6), np.float32)
        q[:, 0:2] = rng.uniform(-0.7, 1.8, (n, 2))
        scale = 10.0 ** rng.uniform(-4, -0.3, (n, 1))
        q[:, 2:6] = rng.normal(0, 1, (n, 4)) * scale
        q[: n // 10, 2:6] = 0          # zero footprints
        q[n // 10: n // 5, 4:6] = 0    # degenerate ellipses
        dev = integ.texture_lookup(tex, q)
        with ob.exact_libm():   # (the level of detail rests on logf: correctly rounded on both sides)
            ref = np.array([ob.texture_lookup(s, tex, q[i, 0:2], q[i, 2:4], q[i, 4:6])[0] for i in range(n)], np.float32)
        assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)), (tex, np.abs(dev - ref).max())
        ref = np.array([ob.texture_lookup(s, tex, q[i, 0:2], q[i, 2:4], q[i, 4:6])[0] for i in range(n)], np.float32)
        same = (dev.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
        assert same.mean() > 0.995, (tex, same.mean())   # glibc's logf: a different last bit in < 0.5 % of the lookups
        assert np.allclose(dev, ref, rtol=2e-5, atol=1e-7), (tex, np.abs(dev - ref).max())


def test_a_pool_that_cannot_be_allocated_is_an_error_and_leaves_the_renderer_usable(pt, ob):
    """mi_render_params.path_pool is taken at its word: a pool the device cannot hold fails with an error code (no launch on
    half-built buffers), and the next render with the default pool is the oracle's."""
    big = pt.Scene(CORNELL, spp=1 << 20, xres=64, yres=64)   # 4.3e9 samples: the pool is not clamped by the work
    integ = pt.CreatePathIntegrator(big)
    with pytest.raises(RuntimeError, match="hipMalloc|pool"):
        integ.Render(path_pool=(1 << 31) - 256)   # ~1.7 TB of path state
    film, weight = integ.Render(spp=4)
    s = pt.Scene(CORNELL, spp=4, xres=64, yres=64)
    with ob.exact_libm():
        ofilm, oweight, oc, _ = ob.render(s)
    assert _rel_l2(film, ofilm) < 1e-6 and np.array_equal(weight, oweight)
    c, o = integ.counters.as_dict(), oc.as_dict()
    assert all(c[k] == o[k] for k in ("camera_rays", "regular_rays", "shadow_rays", "total_paths"))


def test_rays_with_more_quadrics_than_the_pending_list_holds(pt, ob):
    """k_trav postpones the quadrics a ray meets (four per ray); a ray that meets more is handed to k_resolve_overflow, which
    re-traverses it in the reference's order with inline quadric tests and commits it like the resolve kernel would have:
    closest-hit, shadow and MIS rays. Exact-mode parity on a row of nine spheres along the view axis, and equal recorded
    rays along that axis."""
    s = pt.Scene(text=st.sphere_row_scene())
    assert s.errors == [] and s.desc.n_spheres == 10
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "sphere row (quadric list overflow)")
    rng = np.random.default_rng(9)
    n = 3000
    o = np.tile(np.array([0, 1, -9], np.float32), (n, 1)) + rng.normal(0, .05, (n, 3)).astype(np.float32)
    tgt = np.stack([rng.normal(.2, .4, n), rng.normal(1, .4, n), np.full(n, 14.0)], -1).astype(np.float32)
    rays = np.concatenate([o, tgt - o, np.full((n, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    dh = integ.trace(rays, any_hit=False)
    oh, _ = ob.trace(s, rays, any_hit=False)
    assert np.array_equal(dh[:, 0].view(np.int32), oh[:, 0].view(np.int32)) and np.array_equal(dh[:, 1], oh[:, 1])
    hits, extra = tc.check_wavefront(integ, rays, *tc.oracle_answers(ob, s), prim_only=True)   # k_trav + k_resolve_overflow
    tc.check_wavefront(integ, rays, *tc.device_answers(integ))
    assert ((extra[:, 2] & 0x100) != 0).sum() > 100   # (rays whose quadric list did overflow)


def test_mis_rays_that_a_visibility_query_cannot_settle(pt, ob):
    """On scenes without instances (and without an alpha mask on an emitter's own mesh) the BSDF-sampled ray of a direct-lighting estimate is traced as a
    visibility query up to the span in which the sampled emitter can be hit (k_trav MODE 3); rays for which that does not
    decide what integrator.cpp:196-203 reads -- something accepted inside the span (an emitter coplanar with the ceiling, a
    sphere light sunk into a wall), a quadric on the way (a light in a glass shell) -- are traced again in the reference's
    order. Exact-mode parity on a scene made of those cases, with one-sided emitters facing away, a two-sided one and a
    partial sphere, under both light distributions."""
    s = pt.Scene(text=st.mis_span_scene())
    assert s.errors == [] and s.desc.n_lights == 9 and s.desc.n_spheres == 4
    _parity(pt, ob, s, "MIS rays: emitters coplanar / sunk / in a shell / partial / one-sided")
    s2 = pt.Scene(text=st.mis_span_scene(depth=3, spp=8).replace('"uniform"', '"power"'))
    _parity(pt, ob, s2, "MIS rays, power light distribution")
    # the same scene 3 000 and 20 000 units from the origin: the computed hit distances of the coplanar emitter and ceiling
    # differ by parts in 10^4 there, more than the span's relative part -- the span k_shade hands over also grows with the
    # coordinates (64 ulps of the distance to the far corner of the world bound). Coverage of that term, not a regression
    # test: a build without it (-DMIPT_EXP_NO_SPAN_SLACK) passes these scenes too; the term follows from the error model of
    # the triangle test (pt_kernels.hip, where the span is formed), not from an observed failure.
    for off in (3000.0, 20000.0):
        s3 = pt.Scene(text=st.mis_span_scene(spp=8, offset=off))
        assert s3.errors == []
        _parity(pt, ob, s3, "MIS rays, scene %g units from the origin" % off)


def test_mis_visibility_kernel_verdicts_on_recorded_rays(pt, ob, tmp_path):
    """k_trav<3>'s own verdicts (mi_pt_trace_wavefront mode 3) on rays whose closest hit the oracle knows: with the span's end
    tMax at twice the hit distance something is accepted in front of the span (a primitive >= 0); at t (1 + 2^-10) the hit lies
    inside the span [tMax (1 - 2^-8), tMax] and nothing in front of it: ambiguous (-2; or culled with its box where t is
    rounding noise -- then the plain any-hit routine with that tMax finds nothing either); at 0.9 t nothing is accepted (-1);
    rays that miss everything: -1 whatever the span. Every ray answered. 200k-triangle scene, camera rays and incoherent
    rays."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_procedural_scene as mps
    path = tmp_path / "proc.pbrt"
    with open(path, "w") as fh:
        mps.write_scene(fh, 200_000, 96, 1, 7, 5)
    s = pt.Scene(str(path))
    assert s.errors == []
    shape_of = np.array([s.desc.prims[i].shape for i in range(s.desc.n_prims)], np.int64)   # (< 0: one of the four sphere lights)
    integ = pt.CreatePathIntegrator(s)
    rng = np.random.default_rng(17)
    m = 40000
    cam = ob.camera_rays(s, np.stack([rng.integers(0, 96, m), rng.integers(0, 96, m), np.zeros(m, int)], axis=1))
    far, _ = ob.trace(s, tc.unbounded(cam), any_hit=False)
    hit_pts = cam[:, :3] + cam[:, 3:6] * far[:, 1:2]
    sel = far[:, 0].view(np.int32) >= 0
    # incoherent rays: from points just off the surfaces the camera sees, in random directions
    o2 = (hit_pts[sel] - 1e-3 * cam[sel, 3:6]).astype(np.float32)
    d2 = rng.normal(size=o2.shape).astype(np.float32)
    rnd = np.concatenate([o2, d2, np.full((len(o2), 1), np.inf, np.float32)], axis=1).astype(np.float32)
    for rays in (tc.unbounded(cam), rnd):
        want, _ = ob.trace(s, rays, any_hit=False)
        prim, t = want[:, 0].view(np.int32), want[:, 1]
        miss = prim < 0
        hit = ~miss & (shape_of[np.maximum(prim, 0)] >= 0)   # closest hit on a triangle (k_trav only lists the quadrics it meets)
        assert hit.mean() > 0.3
        for scale, expect in ((2.0, None), (1.0 + 2.0 ** -10, -2), (0.9, -1)):
            q = rays.copy()
            q[hit, 6] = (t[hit] * np.float32(scale)).astype(np.float32)
            got, extra = integ.trace_wavefront(q, mode=3)
            gp = got[:, 0].view(np.int32)
            assert (extra[:, 3] != -3).all() and (gp[miss] == -1).all()
            if expect is None:
                assert (gp[hit] >= 0).all()
                assert (gp[hit] == prim[hit]).mean() > 0.5      # (any accepted primitive ends the ray; mostly the closest)
            elif expect == -1:
                assert (gp[hit] == -1).all(), (scale, np.unique(gp[hit], return_counts=True))
            else:
                # (a span of a thousandth of t holds the hit only where t is not float noise: the triangle test works on
                # coordinates relative to the ray's origin, and a ray that starts 1e-3 above the 2000-unit ground plane meets
                # it at a t that is 10 % rounding, in front of where the slab test meets the plane's box -- the reference culls
                # that box at such a tMax too: mi_pt_trace's any-hit answer below is the oracle's)
                solid = hit & (t >= 10)
                assert solid.sum() > 2000 and (gp[solid] == -2).all(), (scale, np.unique(gp[solid], return_counts=True))
                assert np.isin(gp[hit], (-2, -1)).all()
                plain = integ.trace(q, any_hit=True)[:, 0].view(np.int32)
                assert ((gp == -1) == (plain == -1))[hit].all()
    with pytest.raises(RuntimeError):
        integ.trace_wavefront(np.array([[0, 0, 0, 0, 0, 1, -1]], np.float32), mode=3)


def test_roughness_textures_against_oracle(pt, ob, tmp_path):
    """Float image textures on the roughness parameters (plastic.cpp:57-62, uber.cpp:88-96, substrate.cpp:55-60,
    metal.cpp:66-73, translucent.cpp:70-72): the value at the hit, through RoughnessToAlpha unless "remaproughness" is off,
    is the alpha of the material's microfacet lobes -- one axis or both, with the camera ray's differentials in the lookup."""
    st.write_texture_files(str(tmp_path))
    for lens in (False, True):
        s = pt.Scene(text=st.roughness_scene(lens=lens), base_dir=str(tmp_path))
        assert s.errors == []
        mats = [s.desc.materials[i] for i in range(s.desc.n_materials)]
        assert sum(1 for m in mats if m.rough_tex[0] >= 0 or m.rough_tex[1] >= 0) == 5
        assert any(m.rough_tex[0] >= 0 and m.rough_tex[1] < 0 and not (m.rough_flags & 1) for m in mats)   # uber: u only, not remapped
        assert any(m.kind == 9 and m.textured and m.n_bxdfs == 3 and m.tex[0].tex_R >= 0 and m.tex[2].tex_R >= 0 for m in mats)   # the textured mix
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "roughness textures lens=%s" % lens)
    # the maps matter: with constant roughness in their place the picture differs visibly
    flat = st.roughness_scene(lens=True)
    for name in ("r_soft", "r_tga", "r_pfm"):
        flat = flat.replace('"texture roughness" "%s"' % name, '"float roughness" [.1]').replace('"texture uroughness" "%s"' % name, '"float uroughness" [.1]') \
                   .replace('"texture vroughness" "%s"' % name, '"float vroughness" [.1]')
    fs = pt.Scene(text=flat, base_dir=str(tmp_path))
    assert fs.errors == [] and all(fs.desc.materials[i].rough_tex[0] < 0 for i in range(fs.desc.n_materials))
    ff, _, _, _ = ob.render(fs)
    assert _rel_l2(ff, ofilm) > 0.02


def test_disney_with_a_textured_colour_against_oracle(pt, ob, tmp_path):
    """DisneyMaterial::ComputeScatteringFunctions with `color` an image texture (disney.cpp:485-587): the lobes are added
    whatever the colour is, their weights stay constants, and the sheen, specular and specular-transmission spectra are
    formed at the hit from the colour and its luminance (Csheen, Cspec0, strans * Sqrt(c): mi_lobe_rule). And with `roughness` a
    float image texture (disney.cpp:491, 538-541, 568-573): the value at the hit is the FakeSS / Retro lobes' roughness and, through
    the anisotropy's aspect (and 0.65 eta - 0.35 on a thin surface's transmission lobe), the microfacet lobes' alphas. Exact mode."""
    st.write_texture_files(str(tmp_path))
    for lens in (False, True):
        s = pt.Scene(text=st.disney_textured_scene(lens=lens), base_dir=str(tmp_path))
        assert s.errors == []
        mats = [s.desc.materials[i] for i in range(s.desc.n_materials)]
        disney = [m for m in mats if m.kind == 4]
        assert len(disney) == 5 and all(m.textured for m in disney)
        rules = sorted({m.tex[i].rule for m in disney for i in range(m.n_bxdfs) if m.tex[i].tex_R >= 0})
        assert rules == [3, 4, 5, 6]   # MI_LOBE_ALWAYS, _DISNEY_SHEEN, _DISNEY_SPEC, _DISNEY_STRANS
        assert sum(1 for m in disney if m.rough_flags & 4) == 3   # MI_ROUGH_DISNEY: thick + anisotropic, thin, and one without a colour map
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "disney textured colour lens=%s" % lens)
    # the maps matter: with a constant colour in their place the picture differs visibly
    flat = st.disney_textured_scene(lens=True)
    for name in ("ewa_png", "tri_tga", "pfm_clamp", "png_black"):
        flat = flat.replace('"texture color" "%s"' % name, '"rgb color" [.5 .5 .5]')
    fs = pt.Scene(text=flat, base_dir=str(tmp_path))
    assert fs.errors == []
    ff, _, _, _ = ob.render(fs)
    assert _rel_l2(ff, ofilm) > 0.02


def test_metal_with_textured_eta_and_k_against_oracle(pt, ob, tmp_path):
    """MetalMaterial with `eta` / `k` image textures (metal.cpp:119-122): FresnelConductor(1, eta->Evaluate(si), k->Evaluate(si))
    at the hit -- both maps, one of them, a map with black texels (k = 0: a dielectric-like Fresnel term). Exact mode."""
    st.write_texture_files(str(tmp_path))
    for lens in (False, True):
        s = pt.Scene(text=st.metal_textured_scene(lens=lens), base_dir=str(tmp_path))
        assert s.errors == []
        metals = [s.desc.materials[i] for i in range(s.desc.n_materials) if s.desc.materials[i].kind == 6]
        assert len(metals) == 4 and all(m.textured and m.n_bxdfs == 1 and m.tex[0].rule == 7 for m in metals)   # MI_LOBE_METAL
        assert sorted((m.tex[0].tex_S >= 0, m.tex[0].tex_R >= 0) for m in metals) == [(False, True), (True, False), (True, True), (True, True)]
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "metal textured eta k lens=%s" % lens)
    flat = st.metal_textured_scene(lens=True)
    for name in ("ewa_png", "tri_tga", "pfm_clamp", "png_black"):
        flat = flat.replace('"texture eta" "%s"' % name, '').replace('"texture k" "%s"' % name, '')
    fs = pt.Scene(text=flat, base_dir=str(tmp_path))
    assert fs.errors == []
    ff, _, _, _ = ob.render(fs)
    assert _rel_l2(ff, ofilm) > 0.02


def test_matte_with_a_sigma_map_against_oracle(pt, ob, tmp_path):
    """MatteMaterial with `sigma` a float image texture (matte.cpp:55-62): `sig = Clamp(sigma->Evaluate(si), 0, 90)` at the hit,
    LambertianReflection where it is 0 and OrenNayar(r, sig) elsewhere (A, B of reflection.h:414-420). Exact mode."""
    st.write_texture_files(str(tmp_path))
    for lens in (False, True):
        s = pt.Scene(text=st.sigma_textured_scene(lens=lens), base_dir=str(tmp_path))
        assert s.errors == []
        mats = [s.desc.materials[i] for i in range(s.desc.n_materials)]
        assert sum(1 for m in mats if m.sigma_tex >= 0) == 4 and all(m.textured and m.bxdf[0].type == 1 for m in mats if m.sigma_tex >= 0)
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "matte sigma map lens=%s" % lens)
    flat = st.sigma_textured_scene(lens=True)
    for name in ("sig_deg", "sig_big", "sig_b40"):
        flat = flat.replace('"texture sigma" "%s"' % name, '"float sigma" [0]')
    fs = pt.Scene(text=flat, base_dir=str(tmp_path))
    assert fs.errors == [] and all(fs.desc.materials[i].sigma_tex < 0 for i in range(fs.desc.n_materials))
    ff, _, _, _ = ob.render(fs)
    assert _rel_l2(ff, ofilm) > 0.01


def test_glass_with_roughness_maps_against_oracle(pt, ob, tmp_path):
    """GlassMaterial with `uroughness` / `vroughness` float image textures (glass.cpp:60-92): `isSpecular = urough == 0 && vrough ==
    0` is decided at the hit on the values before the remap -- FresnelSpecular there, MicrofacetReflection / MicrofacetTransmission
    with the hit's alphas elsewhere; one axis or both, remapped and not, a black Kr or Kt. Exact mode."""
    st.write_texture_files(str(tmp_path))
    for lens in (False, True):
        s = pt.Scene(text=st.glass_rough_scene(lens=lens), base_dir=str(tmp_path))
        assert s.errors == []
        glass = [s.desc.materials[i] for i in range(s.desc.n_materials) if s.desc.materials[i].kind == 2]
        assert len(glass) == 4 and all(m.textured and (m.rough_flags & 2) and m.bxdf[0].type == 4 for m in glass)   # MI_ROUGH_GLASS; lobe 0 = FresnelSpecular
        assert sorted(m.n_bxdfs for m in glass) == [2, 2, 3, 3]
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "glass roughness maps lens=%s" % lens)
    flat = st.glass_rough_scene(lens=True)
    for name in ("r_b3", "r_t2"):
        flat = flat.replace('"texture uroughness" "%s"' % name, '').replace('"texture vroughness" "%s"' % name, '')
    fs = pt.Scene(text=flat, base_dir=str(tmp_path))
    assert fs.errors == []
    ff, _, _, _ = ob.render(fs)
    assert _rel_l2(ff, ofilm) > 0.01


def test_a_mix_of_a_mix_against_oracle(pt, ob, tmp_path):
    """MixMaterial whose sub-material is a MixMaterial (mixmat.cpp:46-64 twice): every lobe of the inner mix is wrapped in a second
    ScaledBxDF, f = outer * (inner * f) -- two scale spectra per lobe (mi_bxdf.scaled == 2). And a mix of a bump-mapped material: as
    m1 its map bumps the interaction the mix's BSDF is built on, as m2 it acts on a copy nobody reads. Exact mode; a third level is
    reported."""
    st.write_texture_files(str(tmp_path))
    s = pt.Scene(text=st.nested_mix_scene(), base_dir=str(tmp_path))
    assert s.errors == []
    mixes = [s.desc.materials[i] for i in range(s.desc.n_materials) if s.desc.materials[i].kind == 9]
    assert sorted(max(m.bxdf[i].scaled for i in range(m.n_bxdfs)) for m in mixes) == [1, 1, 1, 2, 2]
    assert sorted(m.bump_tex >= 0 for m in mixes) == [False, False, False, False, True]   # (a bump map counts as m1's only: mixmat.cpp:52-56)
    e = next(m for m in mixes if m.n_bxdfs == 4 and m.bxdf[0].scaled == 2)   # nmE: plastic (2 lobes) + mirror under two scales, matte under one
    assert [e.bxdf[i].scaled for i in range(4)] == [2, 2, 2, 1] and e.textured
    film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "a mix of a mix")
    deep = st.nested_mix_scene().replace('WorldEnd', 'MakeNamedMaterial "nmG" "string type" "mix" "string namedmaterial1" "nmE" "string namedmaterial2" "nmA"\n'
                                         'NamedMaterial "nmG"\nShape "sphere"\nWorldEnd')
    assert any("three nested" in x for x in pt.Scene(text=deep, base_dir=str(tmp_path)).errors)


def test_object_instances_against_oracle(pt, ob, tmp_path, monkeypatch):
    """ObjectInstance as the reference's TransformedPrimitive (primitive.cpp:78-99): the ray goes to the instance's space,
    walks the object's own tree, and the interaction comes back through InstanceToWorld (transform.cpp:262-297) -- with
    quadrics, textures, bump maps and alpha masks inside the object, under rotated, stretched and mirrored uses."""
    st.write_texture_files(str(tmp_path))
    st.write_alpha_png(str(tmp_path))
    for lens in (False, True):
        s = pt.Scene(text=st.instanced_scene(lens=lens), base_dir=str(tmp_path))
        assert s.errors == [] and s.desc.n_instances == 5
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "object instances lens=%s" % lens)
    # recorded rays: the same primitive of the object, the same t, whichever use was hit; the same occlusion
    rng = np.random.default_rng(5)
    n = 6000
    o = np.tile(np.array([0, 2.2, -8], np.float32), (n, 1)) + rng.normal(0, .3, (n, 3)).astype(np.float32)
    tgt = np.stack([rng.uniform(-3.5, 3.5, n), rng.uniform(0, 3, n), rng.uniform(-2, 3, n)], -1).astype(np.float32)
    rays = np.concatenate([o, tgt - o, np.full((n, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    for any_hit in (False, True):
        dh = integ.trace(rays, any_hit=any_hit)
        oh, _ = ob.trace(s, rays, any_hit=any_hit)
        if any_hit:
            assert np.array_equal(dh[:, 0].view(np.int32) >= 0, oh[:, 0].view(np.int32) >= 0)
        else:
            assert np.array_equal(dh[:, 0].view(np.int32), oh[:, 0].view(np.int32))
            assert np.array_equal(dh[:, 1], oh[:, 1])
            closest = dh[:, 0].view(np.int32).copy()
    hits, extra = tc.check_wavefront(integ, rays, *tc.oracle_answers(ob, s), prim_only=True)   # k_trav<*, true, 4, INST = true>: return entries
    tc.check_wavefront(integ, rays, *tc.device_answers(integ))
    assert (extra[:, 1] >= 0).sum() > 200 and (extra[:, 1] < s.desc.n_instances).all()   # I_HITINST
    members = set()   # the primitives under the objects' roots (a first child follows its parent in the array)
    for k in range(s.desc.n_instances):
        todo = [int(s.desc.instances[k].root)]
        while todo:
            i = todo.pop()
            nd = s.desc.nodes[i]
            if nd.n_prims > 0: members.update(range(nd.offset, nd.offset + nd.n_prims))
            else: todo += [i + 1, nd.offset]
    assert np.isin(closest, sorted(members)).sum() > 200   # (rays did land on the objects' primitives)
    # and the same picture as with the instances expanded into world shapes (a different tree and float path: statistically)
    monkeypatch.setenv("MIPT_INSTANCES", "expand")
    flat = pt.Scene(text=st.instanced_scene(lens=True), base_dir=str(tmp_path))
    assert flat.errors == [] and flat.desc.n_instances == 0
    ff, fw = pt.CreatePathIntegrator(flat).Render()
    assert _rel_l2(ff, film) < 0.05


def test_random_scenes_against_oracle(pt, ob, tmp_path):
    """Fuzz: 24 seeded random scenes over the whole supported feature set (scenes_text.random_scene) through the HIP path and
    the oracle (exact mode: same decisions in every path). Per scene: identical filter weights, ray counters equal, image
    relative L2 at the float-accumulation level, no NaNs."""
    st.write_texture_files(str(tmp_path))
    st.write_alpha_png(str(tmp_path))
    worst = []
    for seed in range(24):
        s = pt.Scene(text=st.random_scene(seed), base_dir=str(tmp_path))
        assert s.errors == [], (seed, s.errors)
        film, weight, integ, ofilm, oweight, oc = _parity(pt, ob, s, "random scene %d" % seed, weights_exact=False)
        worst.append((_rel_l2(film, ofilm), seed))
    assert np.median([r for r, _ in worst]) < 1e-6, sorted(worst)[-5:]
