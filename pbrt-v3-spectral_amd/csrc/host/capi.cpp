// capi.cpp -- extern "C" surface of the host front end (include/mi_scene.h).
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include "../../../include/mi_scene.h"
#include "integrator.h"
#include "scene.h"

using namespace mipt;

struct mi_scene { HostScene *hs; };
static thread_local std::string g_err;

// Nothing may throw across the C ABI (include/mi_scene.h): a corrupt input that gets as far as an allocation
// (a header that promises 2^40 pixels, ...) comes back as an error code like every other malformed scene.
template <typename F>
static int Guarded(F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        g_err = "out of memory while loading the scene (corrupt size in an input file?)";
        return MI_ERR_NOMEM;
    } catch (const std::length_error &e) {
        g_err = std::string("size out of range while loading the scene: ") + e.what();
        return MI_ERR_NOMEM;
    } catch (const std::exception &e) {
        g_err = std::string("scene front end: ") + e.what();
        return MI_ERR_INVALID;
    } catch (...) {
        g_err = "scene front end: unknown exception";
        return MI_ERR_INVALID;
    }
}

static LoadOverrides ToOv(const mi_scene_overrides *ov) {
    LoadOverrides o;
    if (ov) {
        o.spp = ov->spp; o.xres = ov->xres; o.yres = ov->yres; o.maxDepth = ov->max_depth;
        for (int i = 0; i < 4; ++i) o.crop[i] = ov->crop[i];
        if (ov->light_strategy) o.lightStrategy = ov->light_strategy;
    }
    return o;
}

extern "C" {

int mi_scene_load_file(const char *path, const mi_scene_overrides *ov, mi_scene **out) {
    if (!path || !out) { g_err = "null argument"; return MI_ERR_INVALID; }
    return Guarded([&]() -> int {
        std::string err;
        HostScene *hs = LoadSceneFile(path, ToOv(ov), &err);
        if (!hs) { g_err = err; return MI_ERR_INVALID; }
        *out = new mi_scene{hs};
        return MI_OK;
    });
}

int mi_scene_load_string(const char *text, const char *base_dir, const mi_scene_overrides *ov, mi_scene **out) {
    if (!text || !out) { g_err = "null argument"; return MI_ERR_INVALID; }
    return Guarded([&]() -> int {
        std::string err;
        HostScene *hs = LoadSceneString(text, base_dir ? base_dir : ".", ToOv(ov), &err);
        if (!hs) { g_err = err; return MI_ERR_INVALID; }
        *out = new mi_scene{hs};
        return MI_OK;
    });
}

int mi_scene_save_cache(const mi_scene *s, const char *path) {
    if (!s || !path) { g_err = "null argument"; return MI_ERR_INVALID; }
    return Guarded([&]() -> int {
        std::string err;
        if (!SaveSceneCache(*s->hs, path, &err)) { g_err = err; return MI_ERR_INVALID; }
        return MI_OK;
    });
}

int mi_scene_load_cache(const char *path, mi_scene **out) {
    if (!path || !out) { g_err = "null argument"; return MI_ERR_INVALID; }
    return Guarded([&]() -> int {
        std::string err;
        HostScene *hs = LoadSceneCache(path, &err);
        if (!hs) { g_err = err; return MI_ERR_INVALID; }
        *out = new mi_scene{hs};
        return MI_OK;
    });
}

const mi_scene_desc *mi_scene_get_desc(const mi_scene *s) { return s ? &s->hs->desc : nullptr; }

void mi_scene_get_stats(const mi_scene *s, mi_scene_stats *o) {
    if (!s || !o) return;
    const SceneStats &st = s->hs->stats;
    o->n_triangles = st.nTriangles; o->n_spheres = st.nSpheres; o->n_meshes = st.nMeshes;
    o->interior_nodes = st.interiorNodes; o->leaf_nodes = st.leafNodes;
    o->n_lights = st.nLights; o->n_materials = st.nMaterials;
    o->n_warnings = (int)s->hs->warnings.size(); o->n_errors = (int)s->hs->errors.size();
    o->accel_on_device = s->hs->hlbvhOnDevice ? 1 : 0;
}

const char *mi_scene_message(const mi_scene *s, int kind, int i) {
    if (!s || i < 0) return nullptr;
    const auto &v = kind ? s->hs->errors : s->hs->warnings;
    return i < (int)v.size() ? v[i].c_str() : nullptr;
}

const char *mi_scene_film_filename(const mi_scene *s) { return s ? s->hs->filmFilename.c_str() : nullptr; }

void mi_scene_free(mi_scene *s) {
    if (!s) return;
    delete s->hs;
    delete s;
}

const char *mi_scene_last_error(void) { return g_err.c_str(); }

int mi_film_write_dat(const char *filename, int w, int h, const float *film_sum, float scale) {
    if (!filename || !film_sum || w <= 0 || h <= 0) { g_err = "bad argument"; return MI_ERR_INVALID; }
    std::string err;
    if (!WriteSpectralDat(filename, w, h, film_sum, scale, &err)) { g_err = err; return MI_ERR_INVALID; }
    return MI_OK;
}

int mi_film_write_rgb(const char *filename, int w, int h, const float *film_sum, const float *weight_sum, float scale) {
    if (!filename || !film_sum || !weight_sum || w <= 0 || h <= 0) { g_err = "bad argument"; return MI_ERR_INVALID; }
    std::string err;
    if (!WriteRGBImage(filename, w, h, film_sum, weight_sum, scale, nullptr, &err)) { g_err = err; return MI_ERR_INVALID; }
    return MI_OK;
}

int mi_film_read_dat(const char *filename, int *w, int *h, float *data, uint64_t capacity) {
    if (!filename || !w || !h) { g_err = "bad argument"; return MI_ERR_INVALID; }
    FILE *f = fopen(filename, "rb");
    if (!f) { g_err = std::string("cannot open ") + filename; return MI_ERR_INVALID; }
    int n = 0;
    char tag[8] = {0};
    if (fscanf(f, "%d %d %d\n", w, h, &n) != 3 || n != MI_NSPEC) { fclose(f); g_err = "bad .dat header"; return MI_ERR_INVALID; }
    if (!fgets(tag, sizeof(tag), f) || strncmp(tag, "v3", 2) != 0) { fclose(f); g_err = "missing v3 tag"; return MI_ERR_INVALID; }
    if (data) {
        size_t np = (size_t)(*w) * (*h);
        if (capacity < np * MI_NSPEC) { fclose(f); g_err = "buffer too small"; return MI_ERR_INVALID; }
        std::unique_ptr<double[]> plane(new double[np]);
        for (int c = 0; c < MI_NSPEC; ++c) {
            if (fread(plane.get(), sizeof(double), np, f) != np) { fclose(f); g_err = "short read"; return MI_ERR_INVALID; }
            for (size_t j = 0; j < np; ++j) data[j * MI_NSPEC + c] = (float)plane[j];
        }
    }
    fclose(f);
    return MI_OK;
}

int mi_integrator_render(const mi_scene *s, int device_ordinal, const char *outfile, mi_counters *counters) {
    if (!s) { g_err = "null scene"; return MI_ERR_INVALID; }
    return Guarded([&]() -> int {
        std::unique_ptr<PathIntegrator> integ(CreatePathIntegrator(*s->hs, device_ordinal, outfile ? outfile : ""));
        std::string err;
        int rc = integ->Render(*s->hs, &err);
        if (counters) *counters = integ->counters;
        g_err = err;
        return rc;
    });
}

}  // extern "C"
