// hlbvh_device.cpp -- Accelerator "bvh" "string splitmethod" "hlbvh" through the device build of libmipt_hip.so
// (mi_bvh_build_hlbvh, include/mi_pt.h): bind the library at run time like the integrator does, hand over the primitive
// bounds, supply the SAH tree over the treelet roots (BuildUpperSAH, bvh.cpp) as the callback.
#include <dlfcn.h>
#include <cstdlib>
#include "scene.h"

namespace mipt {
namespace {

int UpperCallback(void *, uint32_t nTreelets, const float *rootBounds, const int32_t *treeletSizes, mi_bvh_node *upperNodes,
                  int32_t *upperIndex, uint32_t *nUpper, uint32_t *nTotal, int32_t *treeletOffset) {
    std::vector<mi_bvh_node> nodes;
    std::vector<int> index;
    const int total = BuildUpperSAH(nTreelets, rootBounds, treeletSizes, &nodes, &index, treeletOffset);
    if (nodes.size() > nTreelets) return MI_ERR_INVALID;   // (a binary tree over k leaves has k - 1 interior nodes)
    for (size_t k = 0; k < nodes.size(); ++k) { upperNodes[k] = nodes[k]; upperIndex[k] = index[k]; }
    *nUpper = (uint32_t)nodes.size();
    *nTotal = (uint32_t)total;
    return MI_OK;
}

}  // namespace

bool BuildHLBVHOnDevice(const std::vector<Bounds3> &primBounds, int maxPrimsInNode, int device, std::vector<mi_bvh_node> *nodes,
                        std::vector<int> *orderedPrims, int *interior, int *leaves, double *seconds, std::string *why) {
    using BuildFn = int (*)(int, const float *, uint32_t, int32_t, mi_bvh_upper_fn, void *, mi_bvh_node *, uint32_t, uint32_t *, int32_t *, double *);
    using ErrFn = const char *(*)(void);
    std::vector<std::string> candidates;
    if (const char *env = getenv("MIPT_HIP_LIB")) candidates.push_back(env);
    Dl_info info;
    if (dladdr((void *)&UpperCallback, &info) && info.dli_fname) {
        std::string self = info.dli_fname;
        const size_t slash = self.find_last_of('/');
        if (slash != std::string::npos) candidates.push_back(self.substr(0, slash) + "/libmipt_hip.so");
    }
    candidates.push_back("libmipt_hip.so");
    void *lib = nullptr;
    for (const auto &c : candidates) if ((lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) { *why = "libmipt_hip.so not loadable"; return false; }
    BuildFn build = (BuildFn)dlsym(lib, "mi_bvh_build_hlbvh");
    ErrFn lastErr = (ErrFn)dlsym(lib, "mi_bvh_last_error");
    if (!build) { *why = "libmipt_hip.so does not export mi_bvh_build_hlbvh"; return false; }
    const uint32_t n = (uint32_t)primBounds.size();
    static_assert(sizeof(Bounds3) == 6 * sizeof(float), "Bounds3 is {min xyz, max xyz}");
    nodes->assign((size_t)2 * n + 1, mi_bvh_node{});
    orderedPrims->assign(n, 0);
    uint32_t nNodes = 0;
    const int rc = build(device, reinterpret_cast<const float *>(primBounds.data()), n, maxPrimsInNode, &UpperCallback, nullptr, nodes->data(),
                         (uint32_t)nodes->size(), &nNodes, orderedPrims->data(), seconds);
    if (rc != MI_OK) { *why = lastErr ? lastErr() : "mi_bvh_build_hlbvh failed"; nodes->clear(); orderedPrims->clear(); return false; }
    nodes->resize(nNodes);
    nodes->shrink_to_fit();
    *interior = *leaves = 0;
    for (const mi_bvh_node &ln : *nodes) (ln.n_prims > 0 ? *leaves : *interior)++;
    return true;
}

}  // namespace mipt
