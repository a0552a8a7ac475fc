// main.cpp -- thin CLI with the reference's flag names (src/main/pbrt.cpp:83-139):
//   pbrt_amd [--outfile F] [--quiet] [--nthreads N (accepted, unused: the render runs on
//   the GPU)] [--gpu ORDINAL] [--spp N] scene.pbrt
// Parses the scene with the host front end and runs Integrator "path" on the HIP path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../../include/mi_scene.h"

int main(int argc, char **argv) {
    std::string outfile, scenefile;
    bool quiet = false;
    int gpu = 0;
    mi_scene_overrides ov;
    ov.spp = ov.xres = ov.yres = ov.max_depth = -1;
    ov.crop[0] = ov.crop[1] = ov.crop[2] = ov.crop[3] = -1;
    ov.light_strategy = nullptr;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            std::string p = std::string("--") + name;
            if (a == p && i + 1 < argc) return argv[++i];
            if (a.compare(0, p.size() + 1, p + "=") == 0) return argv[i] + p.size() + 1;
            return nullptr;
        };
        const char *v;
        if ((v = val("outfile"))) outfile = v;
        else if ((v = val("nthreads"))) (void)v;
        else if ((v = val("gpu"))) gpu = atoi(v);
        else if ((v = val("spp"))) ov.spp = atoi(v);
        else if (a == "--quiet") quiet = true;
        else if (a == "--help" || a == "-h") {
            printf("usage: pbrt_amd [--outfile F] [--quiet] [--gpu N] [--spp N] <scene.pbrt>\n");
            return 0;
        } else scenefile = a;
    }
    if (scenefile.empty()) { fprintf(stderr, "pbrt_amd: no scene file given\n"); return 1; }
    mi_scene *scene = nullptr;
    if (mi_scene_load_file(scenefile.c_str(), &ov, &scene) != 0) {
        fprintf(stderr, "Error: %s\n", mi_scene_last_error());
        return 1;
    }
    for (int k = 0; k < 2; ++k)
        for (int i = 0;; ++i) {
            const char *m = mi_scene_message(scene, k, i);
            if (!m) break;
            if (!quiet || k == 1) fprintf(stderr, "%s: %s\n", k ? "Error" : "Warning", m);
        }
    mi_counters c;
    int rc = mi_integrator_render(scene, gpu, outfile.empty() ? nullptr : outfile.c_str(), &c);
    if (rc != 0) {
        fprintf(stderr, "Error: render failed (%d): %s\n", rc, mi_scene_last_error());
        mi_scene_free(scene);
        return 1;
    }
    if (!quiet) {
        // the reference's STAT names (src/core/integrator.cpp:48, src/core/scene.cpp:40-42)
        printf("Statistics:\n  Integrator/Camera rays traced %llu\n  Intersections/Regular ray intersection tests %llu\n"
               "  Intersections/Shadow ray intersection tests %llu\n  Integrator/Zero-radiance paths %llu / %llu\n",
               (unsigned long long)c.camera_rays, (unsigned long long)c.regular_rays, (unsigned long long)c.shadow_rays,
               (unsigned long long)c.zero_radiance_paths, (unsigned long long)c.total_paths);
    }
    mi_scene_free(scene);
    return 0;
}
