// hlbvh.hip -- BVHAccel::HLBVHBuild (src/accelerators/bvh.cpp:404-638) on the device, behind mi_bvh_build_hlbvh
// (include/mi_pt.h). The steps are the reference's, one kernel (or a few) each:
//   centroid bounds      block reduction + ordered-integer atomics                     bvh.cpp:409-412
//   Morton codes         10 bits per axis of the centroid's offset in those bounds      bvh.cpp:107-137, 414-422
//   radix sort           stable, 30 bits; here one bit per pass as a scan-based split   bvh.cpp:139-181
//   treelets             runs of equal top 12 bits                                      bvh.cpp:428-447
//   emitLBVH             one lane per treelet walks its run with an explicit stack      bvh.cpp:474-532
//   buildUpperSAH        SAH over <= 4096 treelet roots: host work, supplied by the caller as a callback (bvh.cpp:534-638)
//   flatten              treelet nodes copied to their place in the depth-first array   bvh.cpp:640-658
// The reference hands out leaf offsets with an atomic from parallel treelet builds, so its primitive order depends on the
// schedule; this build produces the order of ONE thread (leaves in Morton order), which is what the host restatement
// (csrc/host/bvh.cpp, BuildHLBVH) produces too -- the two are compared node for node (tests/test_hlbvh.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "../../../include/mi_pt.h"

namespace {

thread_local std::string g_hlbvhErr;

#define HB_CHK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            g_hlbvhErr = std::string(#x) + ": " + hipGetErrorString(e_);           \
            return MI_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

constexpr int HB_BLOCK = 256;
constexpr int HB_ITEMS = 4;                       // elements per thread in the split kernels
constexpr int HB_TILE = HB_BLOCK * HB_ITEMS;      // elements per block

struct PrimBounds { float mn[3], mx[3]; };

// float <-> unsigned whose integer order is the float order (for atomicMin / atomicMax on bounds)
__device__ __forceinline__ unsigned OrderedOf(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float FloatOfOrdered(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

__device__ __forceinline__ void Centroid(const PrimBounds &b, float c[3]) {
    for (int a = 0; a < 3; ++a) c[a] = .5f * b.mn[a] + .5f * b.mx[a];   // BVHPrimitiveInfo, bvh.cpp:56
}

__global__ void __launch_bounds__(HB_BLOCK) k_centroid_bounds(const PrimBounds *__restrict__ pb, uint32_t n, unsigned *bounds6) {
    float mn[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, mx[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    for (uint32_t i = blockIdx.x * HB_BLOCK + threadIdx.x; i < n; i += gridDim.x * HB_BLOCK) {
        float c[3];
        Centroid(pb[i], c);
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], c[a]); mx[a] = fmaxf(mx[a], c[a]); }
    }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64)); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) { atomicMin(&bounds6[a], OrderedOf(mn[a])); atomicMax(&bounds6[3 + a], OrderedOf(mx[a])); }
}

__device__ __forceinline__ uint32_t LeftShift3(uint32_t x) {   // bvh.cpp:107-130
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

__global__ void __launch_bounds__(HB_BLOCK) k_morton(const PrimBounds *__restrict__ pb, uint32_t n, const unsigned *bounds6, uint32_t *codes, int32_t *idx) {
    const uint32_t i = blockIdx.x * HB_BLOCK + threadIdx.x;
    if (i >= n) return;
    float c[3], o[3];
    Centroid(pb[i], c);
    for (int a = 0; a < 3; ++a) {   // Bounds3::Offset, geometry.h:767-773
        const float lo = FloatOfOrdered(bounds6[a]), hi = FloatOfOrdered(bounds6[3 + a]);
        o[a] = c[a] - lo;
        if (hi > lo) o[a] /= hi - lo;
    }
    const float scale = 1024.f;
    codes[i] = (LeftShift3((uint32_t)(o[2] * scale)) << 2) | (LeftShift3((uint32_t)(o[1] * scale)) << 1) | LeftShift3((uint32_t)(o[0] * scale));
    idx[i] = (int32_t)i;
}

// ---- one pass of the stable split by bit `bit`: zeros keep their order in front, ones keep theirs behind
__global__ void __launch_bounds__(HB_BLOCK) k_split_count(const uint32_t *__restrict__ codes, uint32_t n, int bit, uint32_t *blockZeros) {
    const uint32_t base = blockIdx.x * HB_TILE + threadIdx.x * HB_ITEMS;
    unsigned z = 0;
    for (int k = 0; k < HB_ITEMS; ++k) if (base + k < n && !((codes[base + k] >> bit) & 1u)) ++z;
    for (int off = 32; off > 0; off >>= 1) z += __shfl_down(z, off, 64);
    __shared__ unsigned sw[HB_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = z;
    __syncthreads();
    if (threadIdx.x == 0) blockZeros[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

// exclusive scan of v[0..m) in place by ONE block; v[m] receives the total
__global__ void __launch_bounds__(1024) k_scan_single(uint32_t *v, uint32_t m) {
    __shared__ uint32_t sPart[1024];
    __shared__ uint32_t sCarry;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < m; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < m ? v[i] : 0;
        sPart[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
            const uint32_t t = threadIdx.x >= (unsigned)off ? sPart[threadIdx.x - off] : 0;
            __syncthreads();
            sPart[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = sPart[threadIdx.x], carry = sCarry;
        if (i < m) v[i] = carry + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) sCarry = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) v[m] = sCarry;
}

__global__ void __launch_bounds__(HB_BLOCK) k_split_scatter(const uint32_t *__restrict__ codesIn, const int32_t *__restrict__ idxIn, uint32_t *codesOut,
                                                            int32_t *idxOut, uint32_t n, int bit, const uint32_t *__restrict__ blockZeros, uint32_t nBlocks) {
    const uint32_t base = blockIdx.x * HB_TILE + threadIdx.x * HB_ITEMS;
    uint32_t c[HB_ITEMS];
    unsigned z = 0;
    for (int k = 0; k < HB_ITEMS; ++k) {
        c[k] = base + k < n ? codesIn[base + k] : 0xffffffffu;
        if (base + k < n && !((c[k] >> bit) & 1u)) ++z;
    }
    // exclusive prefix of the threads' zero counts within the block (thread order = element order)
    unsigned incl = z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    __shared__ unsigned sw[HB_BLOCK / 64];
    if (lane == 63) sw[wave] = incl;
    __syncthreads();
    unsigned before = incl - z;
    for (int w = 0; w < wave; ++w) before += sw[w];
    const uint32_t zerosBeforeBlock = blockZeros[blockIdx.x], totalZeros = blockZeros[nBlocks];
    uint32_t zerosBefore = zerosBeforeBlock + before;
    for (int k = 0; k < HB_ITEMS; ++k) {
        const uint32_t i = base + k;
        if (i >= n) break;
        const bool isZero = !((c[k] >> bit) & 1u);
        const uint32_t dst = isZero ? zerosBefore : totalZeros + (i - zerosBefore);
        codesOut[dst] = c[k];
        idxOut[dst] = idxIn[i];
        if (isZero) ++zerosBefore;
    }
}

__global__ void __launch_bounds__(HB_BLOCK) k_treelet_starts(const uint32_t *__restrict__ codes, uint32_t n, uint32_t *starts, uint32_t *count, uint32_t capacity) {
    const uint32_t i = blockIdx.x * HB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t mask = 0x3ffc0000u;   // the top 12 of the 30 bits, bvh.cpp:432-436
    if (i == 0 || (codes[i] & mask) != (codes[i - 1] & mask)) {
        const uint32_t p = atomicAdd(count, 1u);
        if (p < capacity) starts[p] = i;
    }
}

// A treelet node in emission (= depth-first) order: 32 bytes like the final node, `second` relative to the treelet.
struct LbvhNode {
    float mn[3], mx[3];
    int32_t second;      // interior: treelet-relative index of the second child; leaf: first primitive (index into the sorted order)
    uint16_t nPrims;
    uint8_t axis, pad;
};

// emitLBVH for treelet t = sorted primitives [start, start + count): nodes into scratch[2 * start ...], in the order the
// recursion creates them (a node before its subtrees, the first subtree before the second). One lane per treelet.
__global__ void __launch_bounds__(64) k_emit_lbvh(const uint32_t *__restrict__ codes, const int32_t *__restrict__ idx, const PrimBounds *__restrict__ pb,
                                                  const uint32_t *__restrict__ starts, uint32_t nTreelets, uint32_t n, int maxPrimsInNode,
                                                  LbvhNode *scratch, int32_t *treeletSize, float *rootBounds) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nTreelets) return;
    const uint32_t start = starts[t], end = (t + 1 < nTreelets) ? starts[t + 1] : n;
    LbvhNode *nodes = scratch + 2 * (size_t)start;
    struct Task { int first, count, bit, parent; };
    Task stack[40];
    int sp = 0, nNodes = 0;
    stack[sp++] = Task{(int)start, (int)(end - start), 29 - 12, -1};
    while (sp > 0) {
        Task tk = stack[--sp];
        // advance to the next bit that splits the run (bvh.cpp:499-504), or to a leaf
        while (!(tk.bit == -1 || tk.count < maxPrimsInNode) && ((codes[tk.first] >> tk.bit) & 1u) == ((codes[tk.first + tk.count - 1] >> tk.bit) & 1u)) --tk.bit;
        const int me = nNodes++;
        if (tk.parent >= 0) nodes[tk.parent].second = me;   // (only the second child arrives with a parent)
        LbvhNode &nd = nodes[me];
        if (tk.bit == -1 || tk.count < maxPrimsInNode) {
            float mn[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, mx[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
            for (int i = 0; i < tk.count; ++i) {
                const PrimBounds b = pb[idx[tk.first + i]];
                for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], b.mn[a]); mx[a] = fmaxf(mx[a], b.mx[a]); }
            }
            for (int a = 0; a < 3; ++a) { nd.mn[a] = mn[a]; nd.mx[a] = mx[a]; }
            nd.second = tk.first; nd.nPrims = (uint16_t)tk.count; nd.axis = 0; nd.pad = 0;
            continue;
        }
        int searchStart = 0, searchEnd = tk.count - 1;   // the split point, bvh.cpp:506-521
        while (searchStart + 1 != searchEnd) {
            const int mid = (searchStart + searchEnd) / 2;
            if (((codes[tk.first + searchStart] >> tk.bit) & 1u) == ((codes[tk.first + mid] >> tk.bit) & 1u)) searchStart = mid;
            else searchEnd = mid;
        }
        nd.second = -1; nd.nPrims = 0; nd.axis = (uint8_t)(tk.bit % 3); nd.pad = 0;
        stack[sp++] = Task{tk.first + searchEnd, tk.count - searchEnd, tk.bit - 1, me};   // second child: after the whole first subtree
        stack[sp++] = Task{tk.first, searchEnd, tk.bit - 1, -1};                          // first child: the next node
    }
    for (int k = nNodes - 1; k >= 0; --k) {   // interior bounds, children before parents
        LbvhNode &nd = nodes[k];
        if (nd.nPrims > 0) continue;
        const LbvhNode &c0 = nodes[k + 1], &c1 = nodes[nd.second];
        for (int a = 0; a < 3; ++a) { nd.mn[a] = fminf(c0.mn[a], c1.mn[a]); nd.mx[a] = fmaxf(c0.mx[a], c1.mx[a]); }
    }
    treeletSize[t] = nNodes;
    for (int a = 0; a < 3; ++a) { rootBounds[6 * t + a] = nodes[0].mn[a]; rootBounds[6 * t + 3 + a] = nodes[0].mx[a]; }
}

// The treelets' nodes into their places of the final depth-first array (flattenBVHTree, bvh.cpp:640-658): one block per treelet.
__global__ void __launch_bounds__(HB_BLOCK) k_flatten(const LbvhNode *__restrict__ scratch, const uint32_t *__restrict__ starts, const int32_t *__restrict__ treeletSize,
                                                      const int32_t *__restrict__ treeletOffset, uint32_t nTreelets, mi_bvh_node *out) {
    const uint32_t t = blockIdx.x;
    if (t >= nTreelets) return;
    const LbvhNode *nodes = scratch + 2 * (size_t)starts[t];
    const int base = treeletOffset[t], m = treeletSize[t];
    for (int k = threadIdx.x; k < m; k += HB_BLOCK) {
        const LbvhNode nd = nodes[k];
        mi_bvh_node o;
        for (int a = 0; a < 3; ++a) { o.bmin[a] = nd.mn[a]; o.bmax[a] = nd.mx[a]; }
        o.n_prims = nd.nPrims; o.pad = 0;
        if (nd.nPrims > 0) { o.offset = nd.second; o.axis = 0; }
        else { o.offset = base + nd.second; o.axis = nd.axis; }
        out[base + k] = o;
    }
}

__global__ void k_put_upper(const mi_bvh_node *__restrict__ upperNodes, const int32_t *__restrict__ upperIndex, uint32_t nUpper, mi_bvh_node *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nUpper) out[upperIndex[i]] = upperNodes[i];
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <typename T> T *as() const { return (T *)p; }
};

}  // namespace

extern "C" {

const char *mi_bvh_last_error(void) { return g_hlbvhErr.c_str(); }

int mi_bvh_build_hlbvh(int device_ordinal, const float *prim_bounds, uint32_t n, int32_t max_prims_in_node, mi_bvh_upper_fn upper, void *user,
                       mi_bvh_node *nodes_out, uint32_t nodes_capacity, uint32_t *n_nodes, int32_t *ordered_out, double *seconds) {
    if (!prim_bounds || !upper || !nodes_out || !n_nodes || !ordered_out) { g_hlbvhErr = "null argument"; return MI_ERR_INVALID; }
    *n_nodes = 0;
    if (n == 0) return MI_OK;
    if (n > (1u << 30)) { g_hlbvhErr = "too many primitives for the device HLBVH build"; return MI_ERR_UNSUPPORTED; }
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0) { g_hlbvhErr = "no HIP device available"; return MI_ERR_NO_DEVICE; }
    if (device_ordinal < 0 || device_ordinal >= nDev) { g_hlbvhErr = "device ordinal out of range"; return MI_ERR_NO_DEVICE; }
    HB_CHK(hipSetDevice(device_ordinal));
    const auto t0 = std::chrono::steady_clock::now();
    const int maxPrims = std::min(255, std::max(1, (int)max_prims_in_node));
    const uint32_t nTiles = (n + HB_TILE - 1) / HB_TILE, nBlk = (n + HB_BLOCK - 1) / HB_BLOCK;
    constexpr uint32_t kMaxTreelets = 4096;
    Buf dPb, dBounds, dCodes[2], dIdx[2], dZeros, dStarts, dCount, dScratch, dSize, dRoot, dOffset, dOut;
    HB_CHK(dPb.alloc((size_t)n * sizeof(PrimBounds)));
    HB_CHK(hipMemcpy(dPb.p, prim_bounds, (size_t)n * sizeof(PrimBounds), hipMemcpyHostToDevice));
    HB_CHK(dBounds.alloc(6 * 4));
    {
        const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
        HB_CHK(hipMemcpy(dBounds.p, init, sizeof(init), hipMemcpyHostToDevice));
    }
    for (int k = 0; k < 2; ++k) { HB_CHK(dCodes[k].alloc((size_t)n * 4)); HB_CHK(dIdx[k].alloc((size_t)n * 4)); }
    HB_CHK(dZeros.alloc(((size_t)nTiles + 1) * 4));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min<uint32_t>(nBlk, 4096)), dim3(HB_BLOCK), 0, 0, dPb.as<PrimBounds>(), n, dBounds.as<unsigned>());
    hipLaunchKernelGGL(k_morton, dim3(nBlk), dim3(HB_BLOCK), 0, 0, dPb.as<PrimBounds>(), n, dBounds.as<unsigned>(), dCodes[0].as<uint32_t>(), dIdx[0].as<int32_t>());
    int cur = 0;
    for (int bit = 0; bit < 30; ++bit) {   // least significant bit first: each pass stable, so the whole sort is
        hipLaunchKernelGGL(k_split_count, dim3(nTiles), dim3(HB_BLOCK), 0, 0, dCodes[cur].as<uint32_t>(), n, bit, dZeros.as<uint32_t>());
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, 0, dZeros.as<uint32_t>(), nTiles);
        hipLaunchKernelGGL(k_split_scatter, dim3(nTiles), dim3(HB_BLOCK), 0, 0, dCodes[cur].as<uint32_t>(), dIdx[cur].as<int32_t>(), dCodes[cur ^ 1].as<uint32_t>(),
                           dIdx[cur ^ 1].as<int32_t>(), n, bit, dZeros.as<uint32_t>(), nTiles);
        cur ^= 1;
    }
    HB_CHK(hipGetLastError());
    HB_CHK(dStarts.alloc(kMaxTreelets * 4));
    HB_CHK(dCount.alloc(4));
    HB_CHK(hipMemset(dCount.p, 0, 4));
    hipLaunchKernelGGL(k_treelet_starts, dim3(nBlk), dim3(HB_BLOCK), 0, 0, dCodes[cur].as<uint32_t>(), n, dStarts.as<uint32_t>(), dCount.as<uint32_t>(), kMaxTreelets);
    uint32_t nTreelets = 0;
    HB_CHK(hipMemcpy(&nTreelets, dCount.p, 4, hipMemcpyDeviceToHost));
    if (nTreelets == 0 || nTreelets > kMaxTreelets) { g_hlbvhErr = "treelet count out of range"; return MI_ERR_HIP; }
    std::vector<uint32_t> starts(nTreelets);
    HB_CHK(hipMemcpy(starts.data(), dStarts.p, (size_t)nTreelets * 4, hipMemcpyDeviceToHost));
    std::sort(starts.begin(), starts.end());   // (found in arrival order)
    HB_CHK(hipMemcpy(dStarts.p, starts.data(), (size_t)nTreelets * 4, hipMemcpyHostToDevice));
    HB_CHK(dScratch.alloc(2 * (size_t)n * sizeof(LbvhNode)));
    HB_CHK(dSize.alloc((size_t)nTreelets * 4));
    HB_CHK(dRoot.alloc((size_t)nTreelets * 24));
    hipLaunchKernelGGL(k_emit_lbvh, dim3((nTreelets + 63) / 64), dim3(64), 0, 0, dCodes[cur].as<uint32_t>(), dIdx[cur].as<int32_t>(), dPb.as<PrimBounds>(),
                       dStarts.as<uint32_t>(), nTreelets, n, maxPrims, dScratch.as<LbvhNode>(), dSize.as<int32_t>(), dRoot.as<float>());
    std::vector<int32_t> sizes(nTreelets), offsets(nTreelets);
    std::vector<float> roots((size_t)nTreelets * 6);
    HB_CHK(hipMemcpy(sizes.data(), dSize.p, (size_t)nTreelets * 4, hipMemcpyDeviceToHost));
    HB_CHK(hipMemcpy(roots.data(), dRoot.p, (size_t)nTreelets * 24, hipMemcpyDeviceToHost));
    // the SAH tree over the treelet roots: the caller's (host) routine returns the upper nodes with their places in the
    // final array and says where each treelet goes
    uint32_t total = 0, nUpper = 0;
    std::vector<mi_bvh_node> upperNodes(nTreelets);
    std::vector<int32_t> upperIndex(nTreelets);
    const int rc = upper(user, nTreelets, roots.data(), sizes.data(), upperNodes.data(), upperIndex.data(), &nUpper, &total, offsets.data());
    if (rc != MI_OK) { g_hlbvhErr = "upper-tree callback failed"; return rc; }
    if (total > nodes_capacity || nUpper >= nTreelets + 1) { g_hlbvhErr = "node capacity too small"; return MI_ERR_INVALID; }
    Buf dUpperNodes, dUpperIndex;
    HB_CHK(dOffset.alloc((size_t)nTreelets * 4));
    HB_CHK(hipMemcpy(dOffset.p, offsets.data(), (size_t)nTreelets * 4, hipMemcpyHostToDevice));
    HB_CHK(dOut.alloc((size_t)total * sizeof(mi_bvh_node)));
    hipLaunchKernelGGL(k_flatten, dim3(nTreelets), dim3(HB_BLOCK), 0, 0, dScratch.as<LbvhNode>(), dStarts.as<uint32_t>(), dSize.as<int32_t>(), dOffset.as<int32_t>(),
                       nTreelets, dOut.as<mi_bvh_node>());
    if (nUpper) {
        HB_CHK(dUpperNodes.alloc((size_t)nUpper * sizeof(mi_bvh_node)));
        HB_CHK(dUpperIndex.alloc((size_t)nUpper * 4));
        HB_CHK(hipMemcpy(dUpperNodes.p, upperNodes.data(), (size_t)nUpper * sizeof(mi_bvh_node), hipMemcpyHostToDevice));
        HB_CHK(hipMemcpy(dUpperIndex.p, upperIndex.data(), (size_t)nUpper * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_put_upper, dim3((nUpper + HB_BLOCK - 1) / HB_BLOCK), dim3(HB_BLOCK), 0, 0, dUpperNodes.as<mi_bvh_node>(), dUpperIndex.as<int32_t>(), nUpper,
                           dOut.as<mi_bvh_node>());
    }
    HB_CHK(hipGetLastError());
    HB_CHK(hipDeviceSynchronize());
    HB_CHK(hipMemcpy(nodes_out, dOut.p, (size_t)total * sizeof(mi_bvh_node), hipMemcpyDeviceToHost));
    HB_CHK(hipMemcpy(ordered_out, dIdx[cur].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    *n_nodes = total;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return MI_OK;
}

}  // extern "C"
