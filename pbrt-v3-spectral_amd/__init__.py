"""pbrt-v3-spectral_amd -- MI355X-native spectral PathIntegrator hot path.

Python here is plumbing only (ctypes over the two C-ABI libraries):

* ``libmipt_host.so``  -- .pbrt front end -> flat ``mi_scene_desc`` (include/mi_scene.h)
* ``libmipt_hip.so``   -- hand-written HIP wavefront path tracer (include/mi_pt.h)

The host-side mirror of the reference's integrator surface is
:class:`PathIntegrator` with ``Render(scene)`` (reference:
``SamplerIntegrator::Render``, src/core/integrator.cpp:228-342, created by
``CreatePathIntegrator``, src/integrators/path.cpp:190-213).  There is no CPU
fallback: if the HIP library or a GPU is missing, ``PathIntegrator`` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NSPEC = 31
MAX_BXDFS = 8
PATH_RECORD_FLOATS = 96


# ----------------------------------------------------------------------------
# ctypes mirrors of include/mi_pt.h
class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_int32),
                ("n_prims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


class Prim(C.Structure):
    _fields_ = [("shape", C.c_int32), ("material", C.c_int32), ("area_light", C.c_int32), ("instance", C.c_int32)]


class Instance(C.Structure):
    _fields_ = [("i2w", C.c_float * 16), ("w2i", C.c_float * 16), ("root", C.c_uint32), ("pad", C.c_uint32 * 3)]


class Mesh(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32),
                ("first_tri", C.c_uint32), ("n_tris", C.c_uint32), ("alpha_tex", C.c_int32), ("shadow_alpha_tex", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("o2w", C.c_float * 16), ("w2o", C.c_float * 16), ("radius", C.c_float), ("z_min", C.c_float),
                ("z_max", C.c_float), ("theta_min", C.c_float), ("theta_max", C.c_float), ("phi_max", C.c_float),
                ("reverse_orientation", C.c_int32), ("swaps_handedness", C.c_int32)]


class Bxdf(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_int32), ("fresnel", C.c_int32), ("scaled", C.c_int32),
                ("p", C.c_float * 8), ("R", C.c_float * NSPEC), ("S", C.c_float * NSPEC),
                ("K", C.c_float * NSPEC), ("scale", C.c_float * NSPEC), ("scale2", C.c_float * NSPEC)]


class LobeTex(C.Structure):
    _fields_ = [("tex_R", C.c_int32), ("tex_S", C.c_int32), ("flags", C.c_uint32), ("rule", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("n_bxdfs", C.c_int32), ("eta", C.c_float), ("kind", C.c_int32), ("textured", C.c_int32),
                ("bxdf", Bxdf * MAX_BXDFS), ("tex", LobeTex * MAX_BXDFS), ("bump_tex", C.c_int32), ("rough_tex", C.c_int32 * 2), ("rough_flags", C.c_uint32), ("sigma_tex", C.c_int32)]


MAX_MIP_LEVELS = 16


class MipMap(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("wrap", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("texels", C.POINTER(C.c_float)), ("level_offset", C.c_uint32 * MAX_MIP_LEVELS)]


class Texture(C.Structure):
    _fields_ = [("mipmap", C.c_int32), ("filter", C.c_int32), ("max_aniso", C.c_float),
                ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float), ("post_scale", C.c_float),
                ("type", C.c_int32), ("aa_none", C.c_int32), ("spec1", C.c_float * NSPEC), ("spec2", C.c_float * NSPEC)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("shape", C.c_int32), ("two_sided", C.c_int32), ("area", C.c_float),
                ("L", C.c_float * NSPEC), ("pos", C.c_float * 3), ("dir", C.c_float * 3),
                ("world_radius", C.c_float), ("world_center", C.c_float * 3), ("envmap", C.c_int32),
                ("l2w", C.c_float * 9), ("w2l", C.c_float * 9), ("cos_total_width", C.c_float), ("cos_falloff_start", C.c_float)]


class EnvMap(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_float)), ("nu", C.c_int32), ("nv", C.c_int32),
                ("cond_func", C.POINTER(C.c_float)), ("cond_cdf", C.POINTER(C.c_float)), ("cond_func_int", C.POINTER(C.c_float)),
                ("marg_func", C.POINTER(C.c_float)), ("marg_cdf", C.POINTER(C.c_float)), ("marg_func_int", C.c_float)]


class LightDistrib(C.Structure):
    _fields_ = [("type", C.c_int32), ("n_voxels", C.c_int32 * 3), ("n_distributions", C.c_uint32),
                ("func", C.POINTER(C.c_float)), ("cdf", C.POINTER(C.c_float)), ("func_int", C.POINTER(C.c_float))]


class Camera(C.Structure):
    _fields_ = [("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float), ("shutter_open", C.c_float),
                ("shutter_close", C.c_float)]


class Film(C.Structure):
    _fields_ = [("full_res", C.c_int32 * 2), ("cropped_bounds", C.c_int32 * 4), ("sample_bounds", C.c_int32 * 4),
                ("filter_radius", C.c_float * 2), ("filter_table", C.c_float * 256), ("scale", C.c_float),
                ("max_sample_luminance", C.c_float)]


class Sampler(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_int64), ("base_scales", C.c_int32 * 2), ("base_exponents", C.c_int32 * 2),
                ("sample_stride", C.c_int32), ("mult_inverse", C.c_int32 * 2), ("sample_at_pixel_center", C.c_int32),
                ("n_dims", C.c_int32), ("primes", C.POINTER(C.c_int32)), ("prime_sums", C.POINTER(C.c_int32)),
                ("perms", C.POINTER(C.c_uint16)), ("n_perms", C.c_uint32),
                ("type", C.c_int32), ("sobol_resolution", C.c_int32), ("sobol_log2_resolution", C.c_int32),
                ("n_sobol_dims", C.c_int32), ("sobol_matrices", C.POINTER(C.c_uint32)),
                ("sobol_vdc", C.POINTER(C.c_uint64)), ("sobol_vdc_inv", C.POINTER(C.c_uint64)),
                ("pixel_dims", C.c_int32), ("x_samples", C.c_int32), ("y_samples", C.c_int32), ("jitter", C.c_int32)]


class Integrator(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("rr_threshold", C.c_float), ("pixel_bounds", C.c_int32 * 4),
                ("n_ca_bands", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32),
                ("n_nodes", C.c_uint32), ("nodes", C.POINTER(BvhNode)),
                ("n_prims", C.c_uint32), ("prims", C.POINTER(Prim)),
                ("n_tris", C.c_uint32), ("tri_indices", C.POINTER(C.c_int32)), ("tri_mesh", C.POINTER(C.c_uint32)),
                ("n_verts", C.c_uint32), ("P", C.POINTER(C.c_float)), ("N", C.POINTER(C.c_float)),
                ("UV", C.POINTER(C.c_float)),
                ("n_meshes", C.c_uint32), ("meshes", C.POINTER(Mesh)),
                ("n_spheres", C.c_uint32), ("spheres", C.POINTER(Sphere)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
                ("light_distrib", LightDistrib), ("camera", Camera), ("film", Film), ("sampler", Sampler),
                ("integrator", Integrator), ("cie_y", C.c_float * NSPEC),
                ("n_envmaps", C.c_uint32), ("envmaps", C.POINTER(EnvMap)), ("rgb_illum", (C.c_float * NSPEC) * 7),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("n_mipmaps", C.c_uint32), ("mipmaps", C.POINTER(MipMap)),
                ("n_instances", C.c_uint32), ("instances", C.POINTER(Instance))]


class Counters(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("regular_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("total_paths", C.c_uint64), ("zero_radiance_paths", C.c_uint64), ("path_length_sum", C.c_uint64),
                ("bvh_nodes_visited", C.c_uint64), ("tri_tests", C.c_uint64), ("bad_samples", C.c_uint64),
                ("iterations", C.c_uint64), ("extend_rays", C.c_uint64), ("extend_nodes", C.c_uint64),
                ("extend_tri_tests", C.c_uint64), ("launches", C.c_uint64 * 3)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "launches"}


class RenderParams(C.Structure):
    _fields_ = [("shard_index", C.c_int32), ("shard_count", C.c_int32), ("flags", C.c_uint32),
                ("path_pool", C.c_uint32), ("spp_override", C.c_int64), ("sample_begin", C.c_int64),
                ("stream", C.c_void_p)]


class SceneOverrides(C.Structure):
    _fields_ = [("spp", C.c_int32), ("xres", C.c_int32), ("yres", C.c_int32), ("max_depth", C.c_int32),
                ("crop", C.c_float * 4), ("light_strategy", C.c_char_p)]


class SceneStats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_triangles", "n_spheres", "n_meshes", "interior_nodes", "leaf_nodes",
                                          "n_lights", "n_materials", "n_warnings", "n_errors", "accel_on_device")]


RENDER_FILM_ON_DEVICE = 1
RENDER_ACCUMULATE = 2

HOST_LIB = os.path.join(_HERE, "libmipt_host.so")
HIP_LIB = os.environ.get("MIPT_HIP_LIB") or os.path.join(_HERE, "libmipt_hip.so")   # override: kernel tuning builds

_host = None
_hip = None


def host_lib():
    """The .pbrt front-end library (pure host C++)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % HOST_LIB)
        lib = C.CDLL(HOST_LIB)
        lib.mi_scene_load_file.argtypes = [C.c_char_p, C.POINTER(SceneOverrides), C.POINTER(C.c_void_p)]
        lib.mi_scene_load_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(SceneOverrides), C.POINTER(C.c_void_p)]
        lib.mi_scene_save_cache.argtypes = [C.c_void_p, C.c_char_p]
        lib.mi_scene_load_cache.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.mi_scene_get_desc.argtypes = [C.c_void_p]
        lib.mi_scene_get_desc.restype = C.POINTER(SceneDesc)
        lib.mi_scene_get_stats.argtypes = [C.c_void_p, C.POINTER(SceneStats)]
        lib.mi_scene_message.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.mi_scene_message.restype = C.c_char_p
        lib.mi_scene_film_filename.argtypes = [C.c_void_p]
        lib.mi_scene_film_filename.restype = C.c_char_p
        lib.mi_scene_free.argtypes = [C.c_void_p]
        lib.mi_scene_last_error.restype = C.c_char_p
        lib.mi_film_write_dat.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float]
        lib.mi_film_read_dat.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float),
                                         C.c_uint64]
        lib.mi_integrator_render.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(Counters)]
        _host = lib
    return _host


def hip_lib():
    """The HIP path (hand-written gfx950 kernels). Raises if it is not built / loadable."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError("HIP extension %s is missing; the product path has no CPU fallback" % HIP_LIB)
        lib = C.CDLL(HIP_LIB)
        lib.mi_pt_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        lib.mi_pt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(Counters)]
        lib.mi_pt_device_film.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        lib.mi_pt_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        lib.mi_pt_pool_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.mi_pt_destroy.argtypes = [C.c_void_p]
        lib.mi_pt_last_error.restype = C.c_char_p
        lib.mi_pt_trace.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_int, C.POINTER(C.c_float)]
        lib.mi_pt_math_probe.argtypes = [C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.mi_pt_trace_wavefront.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.mi_pt_texture_lookup.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.mi_pt_light_distribution.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint64]
        lib.mi_pt_debug_path.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        _hip = lib
    return _hip


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """A parsed .pbrt scene flattened to ``mi_scene_desc`` (reference: pbrtParseFile +
    pbrtWorldEnd up to, not including, ``integrator->Render``; src/core/api.cpp:1617-1707)."""

    def __init__(self, path=None, text=None, base_dir=None, spp=-1, xres=-1, yres=-1, max_depth=-1, crop=None,
                 light_strategy=None, cache=None):
        lib = host_lib()
        if cache is not None:   # a scene another process loaded and saved (Scene.save_cache): no parsing, no BVH build
            h = C.c_void_p()
            if lib.mi_scene_load_cache(os.fsencode(cache), C.byref(h)) != 0:
                raise RuntimeError("scene cache load failed: %s" % lib.mi_scene_last_error().decode())
            self._h = h
            self.desc_ptr = lib.mi_scene_get_desc(h)
            self.desc = self.desc_ptr.contents
            return
        ov = SceneOverrides(spp, xres, yres, max_depth, (C.c_float * 4)(*(crop or (-1, -1, -1, -1))),
                            light_strategy.encode() if light_strategy else None)
        h = C.c_void_p()
        if path is not None:
            rc = lib.mi_scene_load_file(os.fsencode(path), C.byref(ov), C.byref(h))
        else:
            rc = lib.mi_scene_load_string(text.encode(), os.fsencode(base_dir or "."), C.byref(ov), C.byref(h))
        if rc != 0:
            raise RuntimeError("scene load failed: %s" % lib.mi_scene_last_error().decode())
        self._h = h
        self.desc_ptr = lib.mi_scene_get_desc(h)
        self.desc = self.desc_ptr.contents

    def save_cache(self, path):
        """Write the loaded scene (arrays, BVH, tables) as one binary file for the other ranks of a job (Scene(cache=path))."""
        if host_lib().mi_scene_save_cache(self._h, os.fsencode(path)) != 0:
            raise RuntimeError("scene cache save failed: %s" % host_lib().mi_scene_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            host_lib().mi_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stats(self):
        st = SceneStats()
        host_lib().mi_scene_get_stats(self._h, C.byref(st))
        return {n: getattr(st, n) for n, _ in SceneStats._fields_}

    def messages(self, kind):
        out, i = [], 0
        while True:
            m = host_lib().mi_scene_message(self._h, kind, i)
            if m is None:
                return out
            out.append(m.decode())
            i += 1

    warnings = property(lambda self: self.messages(0))
    errors = property(lambda self: self.messages(1))

    @property
    def film_filename(self):
        return host_lib().mi_scene_film_filename(self._h).decode()

    @property
    def film_size(self):
        cb = self.desc.film.cropped_bounds
        return cb[2] - cb[0], cb[3] - cb[1]

    @property
    def spp(self):
        return int(self.desc.sampler.samples_per_pixel)


class PathIntegrator:
    """Host mirror of ``PathIntegrator`` for ``Integrator "path"``: ``Render(scene)`` runs
    the HIP wavefront pipeline through the C ABI and returns the spectral film sums."""

    def __init__(self, scene, device=0):
        lib = hip_lib()
        self.scene = scene
        h = C.c_void_p()
        rc = lib.mi_pt_create(scene.desc_ptr, device, C.byref(h))
        if rc != 0:
            raise RuntimeError("mi_pt_create failed (%d): %s" % (rc, lib.mi_pt_last_error().decode()))
        self._h = h
        self.counters = Counters()

    def close(self):
        if getattr(self, "_h", None):
            hip_lib().mi_pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Render(self, shard_index=0, shard_count=1, spp=0, path_pool=0, download=True, accumulate=False,
               sample_begin=0, film_out=None, weight_out=None):
        """Render sample numbers [sample_begin, sample_begin+spp) of this shard's tiles.
        film_out / weight_out: optional device pointers (ints) that receive the film sums
        (e.g. torch CUDA tensors' data_ptr()) instead of a host download."""
        if film_out is not None:
            rp = RenderParams(shard_index, shard_count, (RENDER_ACCUMULATE if accumulate else 0) | RENDER_FILM_ON_DEVICE,
                              path_pool, spp, sample_begin, None)
            rc = hip_lib().mi_pt_render(self._h, C.byref(rp), C.cast(film_out, C.POINTER(C.c_float)),
                                        C.cast(weight_out, C.POINTER(C.c_float)) if weight_out else None,
                                        C.byref(self.counters))
            if rc != 0:
                raise RuntimeError("mi_pt_render failed (%d): %s" % (rc, hip_lib().mi_pt_last_error().decode()))
            return None, None
        w, h = self.scene.film_size
        film = np.zeros((h, w, NSPEC), np.float32) if download else None
        weight = np.zeros((h, w), np.float32) if download else None
        rp = RenderParams(shard_index, shard_count, RENDER_ACCUMULATE if accumulate else 0, path_pool, spp,
                          sample_begin, None)
        rc = hip_lib().mi_pt_render(self._h, C.byref(rp), _fptr(film) if download else None,
                                    _fptr(weight) if download else None, C.byref(self.counters))
        if rc != 0:
            raise RuntimeError("mi_pt_render failed (%d): %s" % (rc, hip_lib().mi_pt_last_error().decode()))
        return film, weight

    def timings(self):
        t = (C.c_double * 8)()
        hip_lib().mi_pt_last_timings(self._h, t, 8)
        return list(t)

    def pool_info(self):
        """(slots, bytes) of the path pool the last render ran on (mi_pt_pool_info)."""
        n, b = C.c_uint64(), C.c_uint64()
        hip_lib().mi_pt_pool_info(self._h, C.byref(n), C.byref(b))
        return int(n.value), int(b.value)

    def device_film(self):
        p = C.c_void_p()
        n = C.c_uint64()
        hip_lib().mi_pt_device_film(self._h, C.byref(p), C.byref(n))
        return p.value, int(n.value)

    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        hits = np.zeros((n, 4), np.float32)
        rc = hip_lib().mi_pt_trace(self._h, _fptr(rays), n, 1 if any_hit else 0, _fptr(hits))
        if rc != 0:
            raise RuntimeError("mi_pt_trace failed: %s" % hip_lib().mi_pt_last_error().decode())
        return hits

    def trace_wavefront(self, rays, mode=0):
        """The rays through the render's own kernels (mi_pt_trace_wavefront): mode 0 path rays (k_trav<0> + resolve),
        1 shadow rays (tMax = 1 - 0.0001f), 2 MIS rays (tMax = inf), 3 MIS rays as visibility queries (tMax = the end of the
        emitter's span; prim = an occluder, -1 or -2 = ambiguous). Returns (hits [n, 4] as trace(), extra [n, 4] int32 view:
        b2 bits, hit instance, raw I_NPEND, raw I_HITPRIM)."""
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        hits = np.zeros((n, 4), np.float32)
        extra = np.zeros((n, 4), np.float32)
        rc = hip_lib().mi_pt_trace_wavefront(self._h, _fptr(rays), n, int(mode), _fptr(hits), _fptr(extra))
        if rc != 0:
            raise RuntimeError("mi_pt_trace_wavefront failed: %s" % hip_lib().mi_pt_last_error().decode())
        return hits, extra.view(np.int32)

    def light_distribution(self):
        """The spatial light-selection tables built at create: (func [nz, ny, nx, n_lights], funcInt [nz, ny, nx])."""
        d = self.scene.desc
        nx, ny, nz = [int(v) for v in d.light_distrib.n_voxels]
        func = np.zeros((nz, ny, nx, d.n_lights), np.float32)
        fint = np.zeros((nz, ny, nx), np.float32)
        rc = hip_lib().mi_pt_light_distribution(self._h, _fptr(func), _fptr(fint), nx * ny * nz)
        if rc != 0:
            raise RuntimeError("mi_pt_light_distribution failed: %s" % hip_lib().mi_pt_last_error().decode())
        return func, fint

    def debug_path(self, px, py, sample, max_records=64):
        """Vertex-by-vertex state of one camera sample (records of PATH_RECORD_FLOATS floats, include/mi_pt.h)."""
        rec = np.zeros((max_records, PATH_RECORD_FLOATS), np.float32)
        n = C.c_int32()
        rc = hip_lib().mi_pt_debug_path(self._h, int(px), int(py), int(sample), max_records, _fptr(rec), C.byref(n))
        if rc != 0:
            raise RuntimeError("mi_pt_debug_path failed: %s" % hip_lib().mi_pt_last_error().decode())
        return rec[: n.value]

    def texture_lookup(self, tex, queries):
        """MIPMap::Lookup on the device: queries [n, 6] = (s, t, dsdx, dtdx, dsdy, dtdy) -> rgb [n, 3]."""
        q = np.ascontiguousarray(queries, np.float32)
        out = np.zeros((q.shape[0], 3), np.float32)
        rc = hip_lib().mi_pt_texture_lookup(self._h, int(tex), q.shape[0], _fptr(q), _fptr(out))
        if rc != 0:
            raise RuntimeError("mi_pt_texture_lookup failed: %s" % hip_lib().mi_pt_last_error().decode())
        return out


def math_probe(op, x, y=None, device=0):
    """mi_pt_math_probe: the device's scalar helpers on arrays. x, y: [n, 2] float32; returns [n, 3]."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(x if y is None else y, np.float32)
    out = np.zeros((x.shape[0], 3), np.float32)
    rc = hip_lib().mi_pt_math_probe(device, int(op), x.shape[0], _fptr(x), _fptr(y), _fptr(out))
    if rc != 0:
        raise RuntimeError("mi_pt_math_probe failed: %s" % hip_lib().mi_pt_last_error().decode())
    return out


def CreatePathIntegrator(scene, device=0):
    """Factory named after the reference's (src/integrators/path.h:69-71)."""
    return PathIntegrator(scene, device)


def write_dat(filename, film, scale=1.0):
    film = np.ascontiguousarray(film, np.float32)
    h, w, _ = film.shape
    rc = host_lib().mi_film_write_dat(os.fsencode(filename), w, h, _fptr(film), scale)
    if rc != 0:
        raise RuntimeError(host_lib().mi_scene_last_error().decode())


def write_rgb(filename, film, weight, scale=1.0):
    """Film::WriteImage with spectralFlag = false: weighted RGB image as .pfm or .tga (mi_film_write_rgb)."""
    film = np.ascontiguousarray(film, np.float32)
    weight = np.ascontiguousarray(weight, np.float32)
    h, w, _ = film.shape
    lib = host_lib()
    lib.mi_film_write_rgb.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
    if lib.mi_film_write_rgb(os.fsencode(filename), w, h, _fptr(film), _fptr(weight), scale) != 0:
        raise RuntimeError(lib.mi_scene_last_error().decode())


def read_dat(filename):
    w, h = C.c_int(), C.c_int()
    lib = host_lib()
    if lib.mi_film_read_dat(os.fsencode(filename), C.byref(w), C.byref(h), None, 0) != 0:
        raise RuntimeError(lib.mi_scene_last_error().decode())
    out = np.zeros((h.value, w.value, NSPEC), np.float32)
    if lib.mi_film_read_dat(os.fsencode(filename), C.byref(w), C.byref(h), _fptr(out), out.size) != 0:
        raise RuntimeError(lib.mi_scene_last_error().decode())
    return out
