"""The reference's own shape tests for this path, restated against the oracle -- and, where they are made of ray
queries, against the device through `mi_pt_trace` (GPU-marked twins below):

  Sphere.SolidAngle      src/tests/shapes.cpp:331-348  (the only reference fixture for Sphere::Sample(ref) / cone sampling)
  Triangle.SolidAngle    src/tests/shapes.cpp:279-324  (Triangle::Sample(ref) against the closed-form spherical area)
  Triangle.Sampling      src/tests/shapes.cpp:210-277  (the same estimate against uniform-sphere hit counting)
  Triangle.Reintersect   src/tests/shapes.cpp:154-205  (spawned rays never hit the triangle they leave)
  FullSphere.Reintersect src/tests/shapes.cpp:375-439  (the convex variant on spheres of radius 1e-4 .. 1e4)

The reference draws its triangles from its PCG32 stream; the property does not depend on which triangles, so these
use numpy's seeded generator and smaller counts (stated per test). Tolerances are the reference's.
"""
import ctypes as C
import math

import numpy as np
import pytest

import scenes_text as st

_F = C.POINTER(C.c_float)


def _f(a):
    return a.ctypes.data_as(_F)


def _scene(pt, body):
    return pt.Scene(text=st._HEAD % dict(res=4, spp=1, depth=1, extra="") + body + "WorldEnd\n")


def _tri_scene(pt, v):
    return _scene(pt, 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [%s]\n'
                  % " ".join(repr(float(x)) for x in np.asarray(v, np.float32).ravel()))


def _radical_pairs(ob, s, n):
    lib = ob.lib()
    return np.array([[lib.oracle_radical_inverse(s.desc_ptr, 0, i), lib.oracle_radical_inverse(s.desc_ptr, 1, i)]
                     for i in range(n)], np.float32)


def _uniform_sphere(u):
    """UniformSampleSphere, sampling.cpp:113-118."""
    z = 1 - 2 * u[:, 0]
    r = np.sqrt(np.maximum(0, 1 - z * z))
    phi = 2 * math.pi * u[:, 1]
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)


def _rays(o, d, tmax=np.inf):
    n = len(d)
    o = np.broadcast_to(np.asarray(o, np.float32), (n, 3))
    return np.concatenate([o, d, np.full((n, 1), tmax, np.float32)], axis=1).astype(np.float32)


def _shape_samples(ob, s, shape, ref, u):
    out = np.zeros((len(u), 7), np.float32)
    refp = np.asarray(ref, np.float32)
    u = np.ascontiguousarray(u, np.float32)
    lib = ob.lib()
    lib.oracle_shape_sample.argtypes = [C.POINTER(type(s.desc)), C.c_int, _F, C.c_int, _F, _F]
    lib.oracle_shape_sample(s.desc_ptr, shape, _f(refp), len(u), _f(u), _f(out))
    return out


def _occluded(trace, rays):
    return trace(rays, True).view(np.int32)[:, 0] >= 0


def _mc_solid_angle(trace, p, dirs):
    """mcSolidAngle, tests/shapes.cpp:319-329: hits / (UniformSpherePdf * n)."""
    return _occluded(trace, _rays(p, dirs)).sum() * 4 * math.pi / len(dirs)


def _sample_solid_angle(ob, s, shape, trace, p, u):
    """Shape::SolidAngle(p, nSamples), shape.cpp:89-103: the mean of 1 / pdf over the samples whose point is visible
    (`!IntersectP(Ray(p, pShape.p - p, .999f))`)."""
    sm = _shape_samples(ob, s, shape, p, u)
    rays = _rays(p, sm[:, 0:3] - np.asarray(p, np.float32), 0.999)
    ok = (sm[:, 6] > 0) & ~_occluded(trace, rays)
    return float((1.0 / sm[ok, 6].astype(np.float64)).sum() / len(u))


SPHERE_FIXTURE = 'AttributeBegin\nTranslate 1 .5 -.8\nRotate 30 1 0 0\nShape "sphere" "float radius" [1]\nAttributeEnd\n'


def _sphere_solid_angle(pt, ob, trace_of):
    s = _scene(pt, SPHERE_FIXTURE)
    trace = trace_of(s)
    n = 128 * 1024
    u = _radical_pairs(ob, s, n)
    dirs = _uniform_sphere(u)
    inside = (1, .9, -.8)
    assert abs(_mc_solid_angle(trace, inside, dirs) - 4 * math.pi) < .01
    assert abs(_sample_solid_angle(ob, s, ~0, trace, inside, u) - 4 * math.pi) < .01
    outside = (-.25, -1, .8)
    mc = _mc_solid_angle(trace, outside, dirs)
    sa = _sample_solid_angle(ob, s, ~0, trace, outside, u)
    assert abs(mc - sa) < .001, (mc, sa)
    # and the closed form the cone sampling stands for: 2 pi (1 - cos theta_max), sin theta_max = r / d
    c = np.array([1, .5, -.8])
    d2 = ((np.array(outside) - c) ** 2).sum()
    assert abs(sa - 2 * math.pi * (1 - math.sqrt(1 - 1 / d2))) < 1e-3


def _oracle_trace(ob):
    return lambda s: (lambda rays, any_hit: ob.trace(s, rays, any_hit=any_hit)[0])


def _device_trace(pt):
    def of(s):
        integ = pt.CreatePathIntegrator(s)
        return lambda rays, any_hit: integ.trace(rays, any_hit=any_hit)
    return of


def _device_trace_wavefront(pt):
    """Every ray query answered by the kernels a render launches (mi_pt_trace_wavefront): closest hits by k_trav<0> +
    k_resolve_extend; occlusion by k_trav<1> + k_resolve_shadow when the rays are NEE shadow rays (tMax = 1 - ShadowEpsilon,
    what SpawnRayTo makes), otherwise -- any other tMax -- as "k_trav<0> found a closest hit"."""
    import trace_check as tc

    def of(s):
        integ = pt.CreatePathIntegrator(s)

        def trace(rays, any_hit):
            rays = np.ascontiguousarray(rays, np.float32)
            if any_hit and (rays[:, 6] == tc.SHADOW_TMAX).all():
                return integ.trace_wavefront(rays, mode=1)[0]
            hits = integ.trace_wavefront(rays, mode=0)[0]
            if any_hit:
                out = np.zeros((len(rays), 4), np.int32)
                out[:, 0] = np.where(hits.view(np.int32)[:, 0] >= 0, 0, -1)
                return out.view(np.float32)
            return hits
        return trace
    return of


def test_sphere_solid_angle(pt, ob):
    """Sphere.SolidAngle: 4 pi from inside (area sampling branch of Sphere::Sample), and cone sampling from outside
    agreeing with uniform-sphere hit counting to 1e-3 -- with the reference's transform, points and sample count."""
    _sphere_solid_angle(pt, ob, _oracle_trace(ob))


@pytest.mark.gpu
def test_sphere_solid_angle_with_device_rays(pt, ob):
    """The same, every IntersectP answered by the HIP traversal kernel."""
    _sphere_solid_angle(pt, ob, _device_trace(pt))


def _spherical_area(v, p):
    """Solid angle of triangle v seen from p (Van Oosterom & Strackee), float64."""
    a, b, c = [(np.asarray(x, np.float64) - p) / np.linalg.norm(np.asarray(x, np.float64) - p) for x in v]
    num = abs(np.dot(a, np.cross(b, c)))
    den = 1 + np.dot(a, b) + np.dot(b, c) + np.dot(c, a)
    return 2 * math.atan2(num, den)


def _random_far_point(rng, rng_range=10):
    pc = rng.uniform(-rng_range, rng_range, 3)
    pc[rng.integers(3)] = (-rng_range - 3) if rng.random() > .5 else (rng_range + 3)
    return pc.astype(np.float32)


def _error(a, b):
    return abs(a - b) if (abs(a) < 1e-4 or abs(b) < 1e-4) else abs((a - b) / b)


def test_triangle_solid_angle(pt, ob):
    """Triangle.SolidAngle: the estimate sum 1 / (count * pdf) over Triangle::Sample(ref, u) equals the spherical area
    of the triangle to 1.5 % (50 triangles with vertices in [-10, 10]^3, 64k Halton points each: the reference's counts)."""
    rng = np.random.default_rng(100)
    s0 = _tri_scene(pt, [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    u = _radical_pairs(ob, s0, 64 * 1024)
    done = 0
    for i in range(50):
        v = rng.uniform(-10, 10, (3, 3)).astype(np.float32)
        if (np.cross(v[1] - v[0], v[2] - v[0]) ** 2).sum() < 1e-20:
            continue
        pc = _random_far_point(rng)
        s = _tri_scene(pt, v)
        sm = _shape_samples(ob, s, 0, pc, u)
        assert (sm[:, 6] > 0).all()
        est = float((1.0 / sm[:, 6].astype(np.float64)).sum() / len(u))
        assert _error(_spherical_area(v, pc.astype(np.float64)), est) < .015, (i, est)
        done += 1
    assert done >= 45


def _triangle_sampling(pt, ob, trace_of, n_tris, count):
    rng = np.random.default_rng(0)
    s0 = _tri_scene(pt, [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    u = _radical_pairs(ob, s0, count)
    dirs = _uniform_sphere(u)
    checked = 0
    for i in range(n_tris):
        v = rng.uniform(-10, 10, (3, 3)).astype(np.float32)
        pc = _random_far_point(rng)
        s = _tri_scene(pt, v)
        unif = _mc_solid_angle(trace_of(s), pc, dirs)
        sm = _shape_samples(ob, s, 0, pc, u)
        est = float((1.0 / sm[:, 6].astype(np.float64)).sum() / count)
        if est > 1e-3:   # "Don't compare really small triangles"
            assert _error(est, unif) < .1, (i, est, unif)
            checked += 1
    assert checked >= n_tris // 2


def test_triangle_sampling_against_uniform_hits(pt, ob):
    """Triangle.Sampling: Triangle::Sample's solid-angle estimate against hits of uniformly distributed rays
    (Triangle::IntersectP), within the reference's 10 %. 12 triangles x 128k rays here (reference: 30 x 512k)."""
    _triangle_sampling(pt, ob, _oracle_trace(ob), 12, 128 * 1024)


@pytest.mark.gpu
def test_triangle_sampling_against_uniform_hits_on_device(pt, ob):
    """The reference's full count (30 triangles x 512k rays), the rays traced by the HIP kernel."""
    _triangle_sampling(pt, ob, _device_trace(pt), 30, 512 * 1024)


def _p_exp(rng, e=8.0, size=None):
    return np.float32(10.0 ** rng.uniform(-e, e, size))


def _spawned(ob, s, ray, targets, mode, faceforward):
    lib = ob.lib()
    lib.oracle_spawn_rays.argtypes = [C.POINTER(type(s.desc)), _F, C.c_int, _F, C.c_int, C.c_int, _F]
    t = np.ascontiguousarray(targets, np.float32)
    out = np.zeros((len(t), 7), np.float32)
    r = np.asarray(ray, np.float32)
    hit = lib.oracle_spawn_rays(s.desc_ptr, _f(r), len(t), _f(t), mode, 1 if faceforward else 0, _f(out))
    return bool(hit), out


def _triangle_reintersect(pt, ob, trace_of, n_tris, n_rays):
    rng = np.random.default_rng(1)
    tested = 0
    for i in range(n_tris):
        v = _p_exp(rng, size=(3, 3))
        if (np.cross(v[1].astype(np.float64) - v[0], v[2].astype(np.float64) - v[0]) ** 2).sum() < 1e-20:
            continue
        s = _tri_scene(pt, v)
        if s.stats["n_triangles"] != 1:
            continue
        su0 = math.sqrt(rng.random())
        b0, b1 = 1 - su0, rng.random() * su0
        target = b0 * v[0] + b1 * v[1] + (1 - b0 - b1) * v[2]
        o = _p_exp(rng, size=3)
        ray = np.concatenate([o, target - o, [np.inf]]).astype(np.float32)
        w = _uniform_sphere(rng.random((n_rays, 2)).astype(np.float32))
        hit, out_dir = _spawned(ob, s, ray, w, 0, False)
        if not hit:
            continue   # "We should almost always find an intersection, but rarely miss, due to round-off error"
        _, out_to = _spawned(ob, s, ray, _p_exp(rng, size=(n_rays, 3)), 1, False)
        trace = trace_of(s)
        for rays in (out_dir, out_to):
            assert not _occluded(trace, rays).any(), i                       # EXPECT_FALSE(tri->IntersectP(rOut))
            assert (trace(rays, False).view(np.int32)[:, 0] < 0).all(), i    # EXPECT_FALSE(tri->Intersect(rOut, ...))
        tested += 1
    assert tested >= n_tris // 2, tested


def test_triangle_reintersect(pt, ob):
    """Triangle.Reintersect: rays spawned at a hit (SpawnRay in random directions, SpawnRayTo random points) never hit the
    triangle again, for triangles and origins with coordinates 10^[-8, 8]. 150 triangles x 2 x 1000 rays here."""
    _triangle_reintersect(pt, ob, _oracle_trace(ob), 150, 1000)


@pytest.mark.gpu
def test_triangle_reintersect_on_device(pt, ob):
    """300 triangles x 2 x 10 000 spawned rays (the reference's ray count per triangle) through the HIP kernel."""
    _triangle_reintersect(pt, ob, _device_trace(pt), 300, 10000)


def _full_sphere_reintersect(pt, ob, trace_of, n_spheres, n_rays):
    rng = np.random.default_rng(2)
    tested = 0
    for i in range(n_spheres):
        radius = float(_p_exp(rng, 4.0))
        s = _scene(pt, 'Shape "sphere" "float radius" [%r]\n' % radius)
        o = _p_exp(rng, size=3)
        p2 = (-radius + 2 * radius * rng.random(3)).astype(np.float32)   # bbox.Lerp(t) of the full sphere
        d = p2 - o
        if rng.random() < .5:
            d = (d / np.linalg.norm(d.astype(np.float64))).astype(np.float32)
        ray = np.concatenate([o, d, [np.inf]]).astype(np.float32)
        w = _uniform_sphere(rng.random((n_rays, 2)).astype(np.float32))
        hit, out_dir = _spawned(ob, s, ray, w, 0, True)
        if not hit:
            continue
        _, out_to = _spawned(ob, s, ray, _p_exp(rng, size=(n_rays, 3)), 1, True)
        trace = trace_of(s)
        for rays in (out_dir, out_to):
            assert not _occluded(trace, rays).any(), (i, radius)
            assert (trace(rays, False).view(np.int32)[:, 0] < 0).all(), (i, radius)
        tested += 1
    assert tested >= n_spheres // 3


def test_full_sphere_reintersect(pt, ob):
    """FullSphere.Reintersect (TestReintersectConvex): from a hit on a sphere of radius 10^[-4, 4], rays leaving into the
    normal's hemisphere never hit the sphere again. 100 spheres (the reference's count) x 2 x 1000 rays."""
    _full_sphere_reintersect(pt, ob, _oracle_trace(ob), 100, 1000)


@pytest.mark.gpu
def test_full_sphere_reintersect_on_device(pt, ob):
    _full_sphere_reintersect(pt, ob, _device_trace(pt), 100, 10000)


@pytest.mark.gpu
def test_shape_pins_through_the_wavefront_kernels(pt, ob):
    """Sphere.SolidAngle, Triangle.Reintersect and FullSphere.Reintersect once more, every Intersect / IntersectP answered
    by the persistent traversal kernels and resolve steps of the render (not by k_trace's per-ray routine)."""
    _sphere_solid_angle(pt, ob, _device_trace_wavefront(pt))
    _triangle_reintersect(pt, ob, _device_trace_wavefront(pt), 100, 2000)
    _full_sphere_reintersect(pt, ob, _device_trace_wavefront(pt), 40, 2000)
