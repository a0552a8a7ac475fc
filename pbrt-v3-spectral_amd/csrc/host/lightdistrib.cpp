// lightdistrib.cpp -- light-selection distribution setup on the host.
// CreateLightSampleDistribution (src/core/lightdistrib.cpp:48-66): one light ->
// uniform regardless of the requested strategy; "power" uses Light::Power().y()
// (src/core/integrator.cpp:217-225, diffuse.cpp:64-66, point.cpp:55, distant.cpp:61-63).
// For "spatial" only the voxel resolution is fixed here (lightdistrib.cpp:96-112);
// the per-voxel pmfs are estimated by the consumer of the description with the
// reference's 128-point Halton estimator (lightdistrib.cpp:232-300), because that
// estimator needs Light::Sample_Li, which lives on the render side of the C ABI.
#include <cmath>
#include "scene.h"

namespace mipt {

static void Distribution1D(const std::vector<float> &f, std::vector<float> *func, std::vector<float> *cdf,
                           std::vector<float> *funcInt) {  // src/core/sampling.h:57-70
    int n = (int)f.size();
    std::vector<float> c(n + 1);
    c[0] = 0;
    for (int i = 1; i < n + 1; ++i) c[i] = c[i - 1] + f[i - 1] / n;
    float fi = c[n];
    if (fi == 0) {
        for (int i = 1; i < n + 1; ++i) c[i] = float(i) / float(n);
    } else {
        for (int i = 1; i < n + 1; ++i) c[i] /= fi;
    }
    func->insert(func->end(), f.begin(), f.end());
    cdf->insert(cdf->end(), c.begin(), c.end());
    funcInt->push_back(fi);
}

void BuildLightDistribution(HostScene *scene, const std::string &strategyIn) {
    mi_lightdistrib &ld = scene->desc.light_distrib;
    ld = mi_lightdistrib{};
    scene->ldFunc.clear(); scene->ldCdf.clear(); scene->ldFuncInt.clear();
    size_t nLights = scene->lights.size();
    if (nLights == 0) { ld.type = MI_LD_UNIFORM; ld.n_distributions = 0; return; }
    std::string strategy = strategyIn;
    if (strategy != "uniform" && strategy != "power" && strategy != "spatial") {
        scene->errors.push_back("Light sample distribution type \"" + strategy + "\" unknown. Using \"spatial\".");
        strategy = "spatial";
    }
    if (strategy == "uniform" || nLights == 1) {
        ld.type = MI_LD_UNIFORM;
        std::vector<float> prob(nLights, 1.f);
        Distribution1D(prob, &scene->ldFunc, &scene->ldCdf, &scene->ldFuncInt);
        ld.n_distributions = 1;
    } else if (strategy == "power") {
        ld.type = MI_LD_POWER;
        std::vector<float> power;
        for (const mi_light &l : scene->lights) {
            Spectrum L = Spectrum::FromArray(l.L), P;
            if (l.type == MI_LIGHT_DIFFUSE_AREA) P = (l.two_sided ? 2 : 1) * L * l.area * kPi;
            else if (l.type == MI_LIGHT_POINT) P = 4 * kPi * L;
            else if (l.type == MI_LIGHT_SPOT) P = L * 2 * kPi * (1 - .5f * (l.cos_falloff_start + l.cos_total_width));  // spot.cpp:71-73
            else if (l.type == MI_LIGHT_INFINITE) P = (kPi * l.world_radius * l.world_radius) * L;  // infinite.cpp:85-89
            else P = L * kPi * l.world_radius * l.world_radius;
            power.push_back(P.y());
        }
        Distribution1D(power, &scene->ldFunc, &scene->ldCdf, &scene->ldFuncInt);
        ld.n_distributions = 1;
    } else {
        ld.type = MI_LD_SPATIAL;
        const mi_bvh_node &root = scene->nodes[0];
        float diag[3] = {root.bmax[0] - root.bmin[0], root.bmax[1] - root.bmin[1], root.bmax[2] - root.bmin[2]};
        int me = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);
        float bmax = diag[me];
        const int maxVoxels = 64;  // lightdistrib.h default
        for (int i = 0; i < 3; ++i)
            ld.n_voxels[i] = std::max(1, int(std::round(diag[i] / bmax * maxVoxels)));
        ld.n_distributions = 0;  // computed by the consumer
    }
}

}  // namespace mipt
