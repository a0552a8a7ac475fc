"""Recorded rays through the kernels a render launches (`mi_pt_trace_wavefront`: the persistent k_trav<0|1|2> with its batched
state machine, cooperative leaf test, postponed quadrics and instance return entries, then the class's resolve step) --
beside `mi_pt_trace`, whose k_trace runs a plain per-ray routine. Restates the question BVHAccel::Intersect / IntersectP
answer (src/accelerators/bvh.cpp:662-738); every comparison is on the int32 bit patterns of (primitive, t, b0, b1)."""
import numpy as np

SHADOW_TMAX = np.float32(1) - np.float32(0.0001)   # Interaction::SpawnRayTo: Ray(o, d, 1 - ShadowEpsilon)


def shadow_form(rays, far=1000.0):
    """(o, d, tMax) -> an NEE shadow ray along the same line: d' = tMax * d (far * d where unbounded), tMax' = 1 - ShadowEpsilon."""
    r = np.array(rays, np.float32, copy=True)
    t = np.where(np.isfinite(r[:, 6]), r[:, 6], np.float32(far)).astype(np.float32)
    r[:, 3:6] = r[:, 3:6] * t[:, None]
    r[:, 6] = SHADOW_TMAX
    return r


def unbounded(rays):
    r = np.array(rays, np.float32, copy=True)
    r[:, 6] = np.inf
    return r


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def check_wavefront(integ, rays, closest, occluded, prim_only=False):
    """closest(rays) -> [n, 4] float32 records the closest-hit kernels must reproduce bit for bit; occluded(rays) -> [n] bool
    for shadow-form rays. Mode 0 on the rays as given, mode 2 on the same rays unbounded, mode 1 on their shadow form.
    prim_only: compare primitive numbers (and t) only where the oracle keeps other barycentrics (quadric hits carry none).
    Returns the mode-0 (hits, extra) for further assertions."""
    rays = np.ascontiguousarray(rays, np.float32)
    out = None
    for mode, rr in ((0, rays), (2, unbounded(rays))):
        got, extra = integ.trace_wavefront(rr, mode=mode)
        want = closest(rr)
        assert (bits(extra)[:, 3] != -2).all(), "a ray was never answered by k_trav<%d>" % mode
        if prim_only:
            assert np.array_equal(bits(got)[:, 0], bits(want)[:, 0]), mode
            assert np.array_equal(bits(got)[:, 1], bits(want)[:, 1]), mode
        else:
            assert np.array_equal(bits(got), bits(want)), mode
        if mode == 0:
            out = (got, extra)
    sh = shadow_form(rays)
    got, extra = integ.trace_wavefront(sh, mode=1)
    assert (bits(extra)[:, 3] != -2).all(), "a ray was never answered by k_trav<1>"
    assert np.array_equal(bits(got)[:, 0] >= 0, np.asarray(occluded(sh), bool))
    assert (bits(got)[:, 1:] == 0).all()
    return out


def oracle_answers(ob, s):
    return (lambda r: ob.trace(s, r, any_hit=False)[0]), (lambda r: ob.trace(s, r, any_hit=True)[0].view(np.int32)[:, 0] >= 0)


def device_answers(integ):
    """k_trace's answers (tests that hold no oracle on the GPU box: the fixture pins k_trace, k_trace pins the wavefront)."""
    return (lambda r: integ.trace(r, any_hit=False)), (lambda r: integ.trace(r, any_hit=True).view(np.int32)[:, 0] >= 0)
