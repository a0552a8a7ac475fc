/*
 * mi_scene.h -- C ABI of the host-side .pbrt front end (libmipt_host.so).
 *
 * It stands where the reference's parse -> pbrtWorldEnd() path stands
 * (src/core/parser.cpp:1094 pbrtParseFile, src/core/api.cpp:1617 pbrtWorldEnd):
 * it reads a scene file and produces the flat mi_scene_desc (include/mi_pt.h)
 * that mi_pt_create() consumes, plus the spectral ".dat" film writer
 * (src/core/film.cpp:226-308). Pure host C++; no HIP, no GPU needed.
 * Errors never abort: the reference's Error()/Warning() "report and continue"
 * policy (src/core/error.cpp:62-102) is kept; messages are retrievable below.
 */
#ifndef MI_SCENE_H
#define MI_SCENE_H
#include "mi_pt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_scene mi_scene;

/* Values < 0 (or NULL) keep what the scene file says. crop = {x0,x1,y0,y1}. */
typedef struct mi_scene_overrides {
    int32_t spp, xres, yres, max_depth;
    float crop[4];
    const char *light_strategy;
} mi_scene_overrides;

typedef struct mi_scene_stats {
    int32_t n_triangles, n_spheres, n_meshes, interior_nodes, leaf_nodes, n_lights, n_materials;
    int32_t n_warnings, n_errors;
    int32_t accel_on_device; /* 1: the BVH was built by libmipt_hip.so's kernels (splitmethod "hlbvh" with a GPU present) */
} mi_scene_stats;

int mi_scene_load_file(const char *path, const mi_scene_overrides *ov, mi_scene **out);
int mi_scene_load_string(const char *text, const char *base_dir, const mi_scene_overrides *ov, mi_scene **out);
/* A loaded scene as one binary file, for the ranks of a multi-GPU job (one process per GPU): rank 0 parses the .pbrt text
 * and builds the BVH once (the reference's single process does both once, src/core/api.cpp:1617-1737) and saves; the other
 * ranks load the arrays. Same-build, same-host hand-over, written atomically (rename); a truncated or foreign file is an
 * error code. */
int mi_scene_save_cache(const mi_scene *s, const char *path);
int mi_scene_load_cache(const char *path, mi_scene **out);
const mi_scene_desc *mi_scene_get_desc(const mi_scene *s);
void mi_scene_get_stats(const mi_scene *s, mi_scene_stats *out);
/* i-th warning (kind 0) / error (kind 1) message, NULL past the end. */
const char *mi_scene_message(const mi_scene *s, int kind, int i);
const char *mi_scene_film_filename(const mi_scene *s);
void mi_scene_free(mi_scene *s);
const char *mi_scene_last_error(void);

/* Film::WriteImage, spectral branch: "<w> <h> 31\nv3 \n" then 31 planes of w*h
 * float64 (plane-major), values multiplied by `scale`. film_sum is [h*w*31]. */
int mi_film_write_dat(const char *filename, int w, int h, const float *film_sum, float scale);
/* Film::WriteImage, RGB branch ("bool spectralFlag" false; src/core/film.cpp:182-225 + imageio.cpp:81-119):
 * XYZ -> RGB of the summed spectrum, division by the filter-weight sum, clamp, scale. Writes PFM (.pfm),
 * TGA (.tga) or scan-line OpenEXR (.exr: HALF B, G, R, ZIP -- WriteImageEXR, imageio.cpp:163-189); any other extension
 * (PNG) is written as a .pfm beside it. weight_sum is [h*w]. */
int mi_film_write_rgb(const char *filename, int w, int h, const float *film_sum, const float *weight_sum, float scale);
/* Read one back: returns 0 and fills w,h; data (if non-NULL) receives
 * [h*w*31] floats pixel-major. */
int mi_film_read_dat(const char *filename, int *w, int *h, float *data, uint64_t capacity);

/* Integrator-shaped entry: what `integrator->Render(*scene)` (src/core/api.cpp:1707)
 * does for Integrator "path": dlopen()s libmipt_hip.so, creates the device
 * renderer, renders all tiles and writes the film file. Fails loudly
 * (MI_ERR_NO_DEVICE / MI_ERR_HIP) when the HIP library or a GPU is missing --
 * there is no CPU fallback. */
int mi_integrator_render(const mi_scene *s, int device_ordinal, const char *outfile, mi_counters *counters);

#ifdef __cplusplus
}
#endif
#endif
