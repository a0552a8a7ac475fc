"""ctypes binding of the CPU oracle (oracle/liboracle_pt.so). TEST INFRASTRUCTURE:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

import pbrt_v3_spectral_amd as pt

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(_ROOT, "oracle", "liboracle_pt.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            raise RuntimeError("oracle not built: make -C oracle")
        l = C.CDLL(ORACLE_LIB)
        D = C.POINTER(pt.SceneDesc)
        F = C.POINTER(C.c_float)
        l.oracle_render.argtypes = [D, C.c_int, C.c_int, C.c_int, C.c_int64, F, F, C.POINTER(pt.Counters)]
        l.oracle_render.restype = C.c_double
        l.oracle_li.argtypes = [D, C.POINTER(C.c_int32), C.c_int, F, C.POINTER(pt.Counters)]
        l.oracle_camera_rays.argtypes = [D, C.POINTER(C.c_int32), C.c_int, F]
        l.oracle_trace.argtypes = [D, F, C.c_uint32, C.c_int, F, C.POINTER(pt.Counters)]
        l.oracle_texture_lookup.argtypes = [D, C.c_int, F, F, F]
        l.oracle_radical_inverse.argtypes = [D, C.c_int, C.c_uint64]
        l.oracle_radical_inverse.restype = C.c_float
        l.oracle_scrambled_radical_inverse.argtypes = [D, C.c_int, C.c_uint64]
        l.oracle_scrambled_radical_inverse.restype = C.c_float
        l.oracle_sample_dimension.argtypes = [D, C.c_int, C.c_int, C.c_int64, C.c_int]
        l.oracle_sample_dimension.restype = C.c_float
        l.oracle_tri_test.argtypes = [F, F, F]
        l.oracle_bsdf.argtypes = [D, C.c_int, C.c_int, F, F, F, C.c_int, F]
        l.oracle_light_pmf.argtypes = [D, F, F]
        l.oracle_light_voxel.argtypes = [D, C.POINTER(C.c_int32), F, F]
        l.oracle_light_table.argtypes = [D, C.c_int, F, F]
        l.oracle_path_log.argtypes = [D, C.c_int, C.c_int, C.c_int64, C.c_int, F]
        _lib = l
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def render(scene, n_threads=None, shard_index=0, shard_count=1, max_samples=-1):
    w, h = scene.film_size
    film = np.zeros((h, w, pt.NSPEC), np.float32)
    weight = np.zeros((h, w), np.float32)
    c = pt.Counters()
    nt = n_threads or os.cpu_count() or 1
    secs = lib().oracle_render(scene.desc_ptr, nt, shard_index, shard_count, max_samples, _f(film), _f(weight), C.byref(c))
    return film, weight, c, secs


def li(scene, samples):
    s = np.ascontiguousarray(samples, np.int32)
    out = np.zeros((len(s), pt.NSPEC), np.float32)
    c = pt.Counters()
    lib().oracle_li(scene.desc_ptr, s.ctypes.data_as(C.POINTER(C.c_int32)), len(s), _f(out), C.byref(c))
    return out, c


def camera_rays(scene, samples):
    s = np.ascontiguousarray(samples, np.int32)
    out = np.zeros((len(s), 7), np.float32)
    lib().oracle_camera_rays(scene.desc_ptr, s.ctypes.data_as(C.POINTER(C.c_int32)), len(s), _f(out))
    return out


def trace(scene, rays, any_hit=False):
    rays = np.ascontiguousarray(rays, np.float32)
    hits = np.zeros((len(rays), 4), np.float32)
    c = pt.Counters()
    lib().oracle_trace(scene.desc_ptr, _f(rays), len(rays), 1 if any_hit else 0, _f(hits), C.byref(c))
    return hits, c


def texture_lookup(scene, tex, st, dstdx=(0, 0), dstdy=(0, 0)):
    """(rgb[3], spectrum[31]) of MIPMap::Lookup + FromRGB for image texture `tex` of the scene."""
    st2 = np.asarray(st, np.float32)
    d4 = np.asarray(list(dstdx) + list(dstdy), np.float32)
    out = np.zeros(34, np.float32)
    lib().oracle_texture_lookup(scene.desc_ptr, int(tex), _f(st2), _f(d4), _f(out))
    return out[:3], out[3:]


def light_voxel(scene, pi):
    """(func[n_lights], funcInt) of the spatial light distribution's voxel pi = (x, y, z)."""
    idx = (C.c_int32 * 3)(*[int(v) for v in pi])
    func = np.zeros(scene.desc.n_lights, np.float32)
    fint = np.zeros(1, np.float32)
    lib().oracle_light_voxel(scene.desc_ptr, idx, _f(func), _f(fint))
    return func, float(fint[0])


def light_table(scene, n_threads=None):
    """Every voxel of the spatial light distribution: (func [nz, ny, nx, n_lights], funcInt [nz, ny, nx])."""
    d = scene.desc
    nx, ny, nz = [int(v) for v in d.light_distrib.n_voxels]
    func = np.zeros((nz, ny, nx, d.n_lights), np.float32)
    fint = np.zeros((nz, ny, nx), np.float32)
    lib().oracle_light_table(scene.desc_ptr, n_threads or os.cpu_count() or 1, _f(func), _f(fint))
    return func, fint


def path_log(scene, px, py, sample, max_records=64):
    """Vertex-by-vertex log of one camera sample (same record layout as PathIntegrator.debug_path)."""
    rec = np.zeros((max_records, pt.PATH_RECORD_FLOATS), np.float32)
    n = lib().oracle_path_log(scene.desc_ptr, int(px), int(py), int(sample), max_records, _f(rec))
    return rec[:n]


def set_libm(mode):
    """libm evaluation mode of the oracle (oracle/o_math.h): 0 = the host's float functions, as the reference binary calls
    them (default; every pin against the reference's own numbers); 1 = correctly rounded, which is what the device
    computes (device-vs-oracle parity). Returns the previous mode."""
    return lib().oracle_set_libm(int(mode))


class exact_libm:
    """with ob.exact_libm(): ... -- the oracle evaluates sin / cos / acos / atan2 / log / pow correctly rounded inside."""

    def __enter__(self):
        self._old = set_libm(1)

    def __exit__(self, *a):
        set_libm(self._old)
