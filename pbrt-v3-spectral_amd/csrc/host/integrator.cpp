// integrator.cpp -- PathIntegrator::Render(): bind the HIP library at run time and
// drive it. No CPU fallback: if libmipt_hip.so or a GPU is missing this fails with
// an error, it never renders on the host.
#include <dlfcn.h>
#include <cstdlib>
#include <vector>
#include "integrator.h"

namespace mipt {
namespace {
struct HipApi {
    void *lib = nullptr;
    int (*create)(const mi_scene_desc *, int, mi_pt **) = nullptr;
    int (*render)(mi_pt *, const mi_render_params *, float *, float *, mi_counters *) = nullptr;
    void (*destroy)(mi_pt *) = nullptr;
    const char *(*last_error)(void) = nullptr;
    int (*timings)(mi_pt *, double *, int) = nullptr;
};

bool LoadHip(HipApi *api, std::string *err) {
    std::vector<std::string> candidates;
    if (const char *env = getenv("MIPT_HIP_LIB")) candidates.push_back(env);
    Dl_info info;
    if (dladdr((void *)&LoadHip, &info) && info.dli_fname) {
        std::string self = info.dli_fname;
        size_t slash = self.find_last_of('/');
        if (slash != std::string::npos) candidates.push_back(self.substr(0, slash) + "/libmipt_hip.so");
    }
    candidates.push_back("libmipt_hip.so");
    std::string tried;
    for (const auto &c : candidates) {
        api->lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (api->lib) break;
        tried += "\n  " + c + ": " + dlerror();
    }
    if (!api->lib) { *err = "HIP extension libmipt_hip.so not loadable (no CPU fallback exists):" + tried; return false; }
    api->create = (decltype(api->create))dlsym(api->lib, "mi_pt_create");
    api->render = (decltype(api->render))dlsym(api->lib, "mi_pt_render");
    api->destroy = (decltype(api->destroy))dlsym(api->lib, "mi_pt_destroy");
    api->last_error = (decltype(api->last_error))dlsym(api->lib, "mi_pt_last_error");
    api->timings = (decltype(api->timings))dlsym(api->lib, "mi_pt_last_timings");
    if (!api->create || !api->render || !api->destroy || !api->last_error) {
        *err = "libmipt_hip.so does not export the mi_pt_* entry points";
        return false;
    }
    return true;
}
}  // namespace

int PathIntegrator::Render(const HostScene &scene, std::string *err) {
    HipApi api;
    if (!LoadHip(&api, err)) return MI_ERR_NO_DEVICE;
    mi_pt *pt = nullptr;
    int rc = api.create(&scene.desc, device, &pt);
    if (rc != MI_OK) { *err = std::string("mi_pt_create: ") + api.last_error(); return rc; }
    const mi_film &f = scene.desc.film;
    int w = f.cropped_bounds[2] - f.cropped_bounds[0], h = f.cropped_bounds[3] - f.cropped_bounds[1];
    std::vector<float> film((size_t)w * h * MI_NSPEC), weight((size_t)w * h);
    mi_render_params rp{};
    rp.shard_index = 0;
    rp.shard_count = 1;
    rc = api.render(pt, &rp, film.data(), weight.data(), &counters);
    if (rc != MI_OK) { *err = std::string("mi_pt_render: ") + api.last_error(); api.destroy(pt); return rc; }
    if (api.timings) api.timings(pt, &seconds, 1);
    api.destroy(pt);
    std::string out = outfile.empty() ? scene.filmFilename : outfile;
    std::string werr;
    if (!scene.spectralFlag) {  // Film::WriteImage, RGB branch (film.cpp:182-225)
        std::string written;
        if (!WriteRGBImage(out, w, h, film.data(), weight.data(), f.scale, &written, &werr)) { *err = werr; return MI_ERR_INVALID; }
        if (written != out) *err = "image written as \"" + written + "\" (this build writes PFM and TGA; EXR/PNG are not linked)";
        return MI_OK;
    }
    if (!WriteSpectralDat(out, w, h, film.data(), f.scale, &werr)) { *err = werr; return MI_ERR_INVALID; }
    return MI_OK;
}

PathIntegrator *CreatePathIntegrator(const HostScene &, int deviceOrdinal, const std::string &outfile) {
    return new PathIntegrator(deviceOrdinal, outfile);
}

}  // namespace mipt
