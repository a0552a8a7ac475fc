// bvh.cpp -- host BVH2 build (SAH with 12 buckets / middle / equal counts) and
// depth-first flatten into 32-byte nodes. The partitioning decisions, including the
// use of std::partition / std::nth_element on the primitive-info array, follow
// BVHAccel::recursiveBuild and flattenBVHTree (src/accelerators/bvh.cpp:236-402,
// 640-658) so the tree (node count, leaf contents, primitive order) is the
// reference's tree: killeroo-simple -> 59 188 interior + 59 189 leaf nodes.
// HLBVH (bvh.cpp:404-638) is a "next" row.
// The recursion is the reference's, but large builds run it on several threads: the top of the tree is built
// serially down to ranges of a grain size, those ranges become independent tasks (a range [start, end) of the
// primitive-info array is private to its subtree, and it always yields end - start ordered primitives, so every
// leaf's offset into the ordered list is known without waiting for the subtrees to its left), and the flatten
// pass walks the stitched tree depth first. Same partitions, same node order, same bytes as the serial build.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <thread>
#include "scene.h"

namespace mipt {
namespace {

struct PrimInfo {
    size_t primitiveNumber;
    Bounds3 bounds;
    Vec3 centroid;
};

struct BuildNode {
    Bounds3 bounds;
    BuildNode *children[2] = {nullptr, nullptr};
    int splitAxis = 0, firstPrimOffset = 0, nPrimitives = 0;
};

struct DeferredSubtree { BuildNode **slot; int start, end, orderedBase; };

struct Builder {
    int maxPrimsInNode;
    SplitMethod method;
    std::vector<std::unique_ptr<BuildNode[]>> chunks;   // nodes in blocks: no per-node allocation
    size_t chunkUsed = 0;
    int *ordered;            // [nPrims], written at explicit offsets
    int orderedNext = 0;     // offset of the next leaf in depth-first order
    int interior = 0, leaves = 0, total = 0;
    int grain = 0;           // > 0: ranges of at most this many primitives are deferred to tasks
    std::vector<DeferredSubtree> *deferred = nullptr;

    BuildNode *Alloc() {
        constexpr size_t kChunk = 4096;
        if (chunks.empty() || chunkUsed == kChunk) { chunks.emplace_back(new BuildNode[kChunk]); chunkUsed = 0; }
        return &chunks.back()[chunkUsed++];
    }
    void InitLeaf(BuildNode *node, std::vector<PrimInfo> &info, int start, int end, const Bounds3 &b) {
        int first = orderedNext;
        for (int i = start; i < end; ++i) ordered[orderedNext++] = (int)info[i].primitiveNumber;
        node->firstPrimOffset = first;
        node->nPrimitives = end - start;
        node->bounds = b;
        ++leaves;
    }
    // a child subtree: built here, or left to a task (its ordered primitives keep their place)
    void Child(BuildNode **slot, std::vector<PrimInfo> &info, int start, int end) {
        if (deferred && end - start <= grain && end - start > 1) {
            deferred->push_back(DeferredSubtree{slot, start, end, orderedNext});
            orderedNext += end - start;
            *slot = nullptr;
        } else
            *slot = Build(info, start, end);
    }
    BuildNode *Build(std::vector<PrimInfo> &info, int start, int end) {
        BuildNode *node = Alloc();
        ++total;
        Bounds3 bounds;
        for (int i = start; i < end; ++i) bounds = Union(bounds, info[i].bounds);
        int nPrimitives = end - start;
        if (nPrimitives == 1) {
            InitLeaf(node, info, start, end, bounds);
            return node;
        }
        Bounds3 centroidBounds;
        for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, info[i].centroid);
        int dim = centroidBounds.MaximumExtent();
        int mid = (start + end) / 2;
        if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) {
            InitLeaf(node, info, start, end, bounds);
            return node;
        }
        bool partitioned = false;
        if (method == SplitMethod::Middle) {
            float pmid = (centroidBounds.pMin[dim] + centroidBounds.pMax[dim]) / 2;
            PrimInfo *midPtr = std::partition(&info[start], &info[end - 1] + 1,
                                              [dim, pmid](const PrimInfo &pi) { return pi.centroid[dim] < pmid; });
            mid = (int)(midPtr - &info[0]);
            if (mid != start && mid != end) partitioned = true;
        }
        if (!partitioned && (method == SplitMethod::Middle || method == SplitMethod::EqualCounts)) {
            mid = (start + end) / 2;
            std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                             [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
            partitioned = true;
        }
        if (!partitioned) {  // SAH
            if (nPrimitives <= 2) {
                mid = (start + end) / 2;
                std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                                 [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
            } else {
                constexpr int nBuckets = 12;
                struct BucketInfo { int count = 0; Bounds3 bounds; } buckets[nBuckets];
                for (int i = start; i < end; ++i) {
                    int b = nBuckets * centroidBounds.Offset(info[i].centroid)[dim];
                    if (b == nBuckets) b = nBuckets - 1;
                    buckets[b].count++;
                    buckets[b].bounds = Union(buckets[b].bounds, info[i].bounds);
                }
                float cost[nBuckets - 1];
                for (int i = 0; i < nBuckets - 1; ++i) {
                    Bounds3 b0, b1;
                    int count0 = 0, count1 = 0;
                    for (int j = 0; j <= i; ++j) { b0 = Union(b0, buckets[j].bounds); count0 += buckets[j].count; }
                    for (int j = i + 1; j < nBuckets; ++j) { b1 = Union(b1, buckets[j].bounds); count1 += buckets[j].count; }
                    cost[i] = 1 + (count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea()) / bounds.SurfaceArea();
                }
                float minCost = cost[0];
                int minCostSplitBucket = 0;
                for (int i = 1; i < nBuckets - 1; ++i)
                    if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
                float leafCost = nPrimitives;
                if (nPrimitives > maxPrimsInNode || minCost < leafCost) {
                    PrimInfo *pmid = std::partition(&info[start], &info[end - 1] + 1, [=](const PrimInfo &pi) {
                        int b = nBuckets * centroidBounds.Offset(pi.centroid)[dim];
                        if (b == nBuckets) b = nBuckets - 1;
                        return b <= minCostSplitBucket;
                    });
                    mid = (int)(pmid - &info[0]);
                } else {
                    InitLeaf(node, info, start, end, bounds);
                    return node;
                }
            }
        }
        Child(&node->children[0], info, start, mid);
        Child(&node->children[1], info, mid, end);
        // (= Union(c0->bounds, c1->bounds), InitInterior bvh.cpp:68-69: the union of the children's primitive bounds,
        // which is the `bounds` computed above; taken from there because a deferred child is not built yet)
        node->bounds = bounds;
        node->splitAxis = dim;
        node->nPrimitives = 0;
        ++interior;
        return node;
    }
};

int Flatten(const BuildNode *node, std::vector<mi_bvh_node> &nodes, int *offset) {
    mi_bvh_node &ln = nodes[*offset];
    for (int k = 0; k < 3; ++k) { ln.bmin[k] = node->bounds.pMin[k]; ln.bmax[k] = node->bounds.pMax[k]; }
    int myOffset = (*offset)++;
    if (node->nPrimitives > 0) {
        ln.offset = node->firstPrimOffset;
        ln.n_prims = (uint16_t)node->nPrimitives;
        ln.axis = 0;
        ln.pad = 0;
    } else {
        ln.axis = (uint8_t)node->splitAxis;
        ln.n_prims = 0;
        ln.pad = 0;
        Flatten(node->children[0], nodes, offset);
        int second = Flatten(node->children[1], nodes, offset);
        nodes[myOffset].offset = second;
    }
    return myOffset;
}

}  // namespace

// ------------------------------------------------------------------ HLBVH (BVHAccel::HLBVHBuild, src/accelerators/bvh.cpp:404-638)
// Morton codes of the centroids (10 bits per axis), a stable radix sort, one LBVH treelet per run of equal top 12 bits
// (emitLBVH: split where the next lower bit changes, leaves of fewer than maxPrimsInNode primitives), and a SAH tree over
// the treelet roots (buildUpperSAH). The reference builds its treelets on several threads and hands out leaf offsets with
// an atomic, so its primitive order depends on scheduling; this is the order one thread produces -- leaves in Morton
// order -- which makes the tree a pure function of the input and lets the device build (mi_bvh_build_hlbvh in
// libmipt_hip.so: the same steps as kernels) be checked against this one node for node.
namespace {

uint32_t LeftShift3(uint32_t x) {   // bvh.cpp:107-130
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

struct LbvhNode {   // a treelet node in emission (= depth-first) order
    Bounds3 bounds;
    int second = 0;       // interior: index of the second child within the treelet (the first is the next node)
    int firstPrim = 0, nPrims = 0, axis = 0;
};

// emitLBVH, bvh.cpp:474-532, for primitives [first, first + n) of the sorted array; appends to `out`, returns the node's index.
int EmitLBVH(std::vector<LbvhNode> &out, const std::vector<uint32_t> &codes, const std::vector<int> &order,
             const std::vector<Bounds3> &primBounds, int first, int n, int bitIndex, int maxPrimsInNode) {
    while (true) {
        if (bitIndex == -1 || n < maxPrimsInNode) {
            LbvhNode leaf;
            leaf.firstPrim = first; leaf.nPrims = n;
            for (int i = 0; i < n; ++i) leaf.bounds = Union(leaf.bounds, primBounds[order[first + i]]);
            out.push_back(leaf);
            return (int)out.size() - 1;
        }
        const uint32_t mask = 1u << bitIndex;
        if ((codes[first] & mask) == (codes[first + n - 1] & mask)) { --bitIndex; continue; }   // no split at this bit
        int searchStart = 0, searchEnd = n - 1;
        while (searchStart + 1 != searchEnd) {
            const int mid = (searchStart + searchEnd) / 2;
            if ((codes[first + searchStart] & mask) == (codes[first + mid] & mask)) searchStart = mid;
            else searchEnd = mid;
        }
        const int splitOffset = searchEnd;
        const int me = (int)out.size();
        out.emplace_back();
        const int c0 = EmitLBVH(out, codes, order, primBounds, first, splitOffset, bitIndex - 1, maxPrimsInNode);
        const int c1 = EmitLBVH(out, codes, order, primBounds, first + splitOffset, n - splitOffset, bitIndex - 1, maxPrimsInNode);
        out[me].axis = bitIndex % 3;
        out[me].second = c1;
        out[me].bounds = Union(out[c0].bounds, out[c1].bounds);
        return me;
    }
}

}  // namespace

void MortonCodesAndOrder(const std::vector<Bounds3> &primBounds, std::vector<uint32_t> *codes, std::vector<int> *order) {
    const size_t n = primBounds.size();
    Bounds3 cb;
    std::vector<Vec3> centroid(n);
    for (size_t i = 0; i < n; ++i) { centroid[i] = .5f * primBounds[i].pMin + .5f * primBounds[i].pMax; cb = Union(cb, centroid[i]); }
    std::vector<uint32_t> code(n);
    for (size_t i = 0; i < n; ++i) {
        Vec3 o = centroid[i] - cb.pMin;   // Bounds3::Offset, geometry.h:767-773
        if (cb.pMax.x > cb.pMin.x) o.x /= cb.pMax.x - cb.pMin.x;
        if (cb.pMax.y > cb.pMin.y) o.y /= cb.pMax.y - cb.pMin.y;
        if (cb.pMax.z > cb.pMin.z) o.z /= cb.pMax.z - cb.pMin.z;
        const float scale = 1024.f;
        code[i] = (LeftShift3((uint32_t)(o.z * scale)) << 2) | (LeftShift3((uint32_t)(o.y * scale)) << 1) | LeftShift3((uint32_t)(o.x * scale));
    }
    // RadixSort, bvh.cpp:139-181: stable, 30 bits (equal codes keep the order of the primitive numbers)
    std::vector<int> idx(n), tmpIdx(n);
    std::vector<uint32_t> tmpCode(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (int)i;
    for (int pass = 0; pass < 5; ++pass) {
        const int lowBit = pass * 6;
        size_t count[64] = {0}, start[64];
        for (size_t i = 0; i < n; ++i) ++count[(code[i] >> lowBit) & 63u];
        start[0] = 0;
        for (int b = 1; b < 64; ++b) start[b] = start[b - 1] + count[b - 1];
        for (size_t i = 0; i < n; ++i) { const size_t d = start[(code[i] >> lowBit) & 63u]++; tmpCode[d] = code[i]; tmpIdx[d] = idx[i]; }
        code.swap(tmpCode);
        idx.swap(tmpIdx);
    }
    codes->swap(code);
    order->swap(idx);
}

// buildUpperSAH (bvh.cpp:534-638) over the treelet roots: the upper interior nodes with their indices in the final
// depth-first array, and where each treelet's (already depth-first) nodes go.
namespace {
struct UpperItem { int treelet; Bounds3 bounds; };
struct UpperBuilder {
    const int32_t *treeletSizes;
    std::vector<mi_bvh_node> *upperNodes;
    std::vector<int> *upperIndex;
    int32_t *treeletOffset;
    int next = 0;   // the next free index of the final array

    int Emit(std::vector<UpperItem> &items, int start, int end) {
        const int n = end - start;
        if (n == 1) {
            const int t = items[start].treelet;
            const int base = next;
            treeletOffset[t] = base;
            next += treeletSizes[t];
            return base;
        }
        const int me = next++;
        const size_t slot = upperNodes->size();
        upperNodes->emplace_back();
        upperIndex->push_back(me);
        Bounds3 bounds, centroidBounds;
        for (int i = start; i < end; ++i) bounds = Union(bounds, items[i].bounds);
        for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, (items[i].bounds.pMin + items[i].bounds.pMax) * 0.5f);
        const int dim = centroidBounds.MaximumExtent();
        int mid;
        if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) {
            mid = (start + end) / 2;   // (the reference CHECKs this away: all centroids coincide; split the list in the middle)
        } else {
            constexpr int nBuckets = 12;
            struct Bucket { int count = 0; Bounds3 bounds; } buckets[nBuckets];
            auto bucketOf = [&](const Bounds3 &b) {
                const float centroid = (b.pMin[dim] + b.pMax[dim]) * 0.5f;
                int k = (int)(nBuckets * ((centroid - centroidBounds.pMin[dim]) / (centroidBounds.pMax[dim] - centroidBounds.pMin[dim])));
                if (k == nBuckets) k = nBuckets - 1;
                return k;
            };
            for (int i = start; i < end; ++i) { const int k = bucketOf(items[i].bounds); buckets[k].count++; buckets[k].bounds = Union(buckets[k].bounds, items[i].bounds); }
            float cost[nBuckets - 1];
            for (int i = 0; i < nBuckets - 1; ++i) {
                Bounds3 b0, b1;
                int count0 = 0, count1 = 0;
                for (int j = 0; j <= i; ++j) { b0 = Union(b0, buckets[j].bounds); count0 += buckets[j].count; }
                for (int j = i + 1; j < nBuckets; ++j) { b1 = Union(b1, buckets[j].bounds); count1 += buckets[j].count; }
                cost[i] = .125f + (count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea()) / bounds.SurfaceArea();
            }
            float minCost = cost[0];
            int minCostSplitBucket = 0;
            for (int i = 1; i < nBuckets - 1; ++i) if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
            UpperItem *pmid = std::partition(&items[start], &items[end - 1] + 1, [&](const UpperItem &it) { return bucketOf(it.bounds) <= minCostSplitBucket; });
            mid = (int)(pmid - &items[0]);
            if (mid == start || mid == end) mid = (start + end) / 2;   // (cannot happen with a valid SAH split; never recurse on an empty side)
        }
        Emit(items, start, mid);
        const int second = Emit(items, mid, end);
        mi_bvh_node &ln = (*upperNodes)[slot];
        for (int a = 0; a < 3; ++a) { ln.bmin[a] = bounds.pMin[a]; ln.bmax[a] = bounds.pMax[a]; }
        ln.offset = second; ln.n_prims = 0; ln.axis = (uint8_t)dim; ln.pad = 0;
        return me;
    }
};
}  // namespace

// Root bounds (6 floats per treelet: min xyz, max xyz) and node counts in; the upper nodes, their final indices, each
// treelet's final offset and the total node count out. (The callback the device build receives: mi_bvh_upper_fn.)
int BuildUpperSAH(uint32_t nTreelets, const float *rootBounds, const int32_t *treeletSizes, std::vector<mi_bvh_node> *upperNodes,
                  std::vector<int> *upperIndex, int32_t *treeletOffset) {
    std::vector<UpperItem> items(nTreelets);
    for (uint32_t t = 0; t < nTreelets; ++t) {
        items[t].treelet = (int)t;
        items[t].bounds.pMin = Vec3(rootBounds[6 * t], rootBounds[6 * t + 1], rootBounds[6 * t + 2]);
        items[t].bounds.pMax = Vec3(rootBounds[6 * t + 3], rootBounds[6 * t + 4], rootBounds[6 * t + 5]);
    }
    upperNodes->clear();
    upperIndex->clear();
    UpperBuilder ub{treeletSizes, upperNodes, upperIndex, treeletOffset};
    if (nTreelets) ub.Emit(items, 0, (int)nTreelets);
    return ub.next;
}

void BuildHLBVH(const std::vector<Bounds3> &primBounds, int maxPrimsInNode, std::vector<mi_bvh_node> *nodes,
                std::vector<int> *orderedPrims, int *interior, int *leaves) {
    nodes->clear();
    orderedPrims->clear();
    *interior = *leaves = 0;
    if (primBounds.empty()) return;
    std::vector<uint32_t> codes;
    MortonCodesAndOrder(primBounds, &codes, orderedPrims);
    const int n = (int)primBounds.size();
    std::vector<std::vector<LbvhNode>> treelets;
    std::vector<float> roots;
    std::vector<int32_t> sizes;
    const uint32_t mask = 0x3ffc0000u;   // the top 12 of the 30 bits
    for (int start = 0, end = 1; end <= n; ++end) {
        if (end == n || ((codes[start] & mask) != (codes[end] & mask))) {
            treelets.emplace_back();
            EmitLBVH(treelets.back(), codes, *orderedPrims, primBounds, start, end - start, 29 - 12, std::min(255, maxPrimsInNode));
            const Bounds3 &rb = treelets.back()[0].bounds;
            for (int a = 0; a < 3; ++a) roots.push_back(rb.pMin[a]);
            for (int a = 0; a < 3; ++a) roots.push_back(rb.pMax[a]);
            sizes.push_back((int32_t)treelets.back().size());
            start = end;
        }
    }
    std::vector<mi_bvh_node> upperNodes;
    std::vector<int> upperIndex;
    std::vector<int32_t> treeletOffset(treelets.size(), 0);
    const int total = BuildUpperSAH((uint32_t)treelets.size(), roots.data(), sizes.data(), &upperNodes, &upperIndex, treeletOffset.data());
    nodes->resize(total);
    for (size_t k = 0; k < upperNodes.size(); ++k) (*nodes)[upperIndex[k]] = upperNodes[k];
    for (size_t t = 0; t < treelets.size(); ++t) {
        const std::vector<LbvhNode> &tn = treelets[t];
        const int base = treeletOffset[t];
        for (size_t k = 0; k < tn.size(); ++k) {
            mi_bvh_node &ln = (*nodes)[base + k];
            for (int a = 0; a < 3; ++a) { ln.bmin[a] = tn[k].bounds.pMin[a]; ln.bmax[a] = tn[k].bounds.pMax[a]; }
            ln.pad = 0;
            if (tn[k].nPrims > 0) { ln.offset = tn[k].firstPrim; ln.n_prims = (uint16_t)tn[k].nPrims; ln.axis = 0; }
            else { ln.offset = base + tn[k].second; ln.n_prims = 0; ln.axis = (uint8_t)tn[k].axis; }
        }
    }
    for (const mi_bvh_node &ln : *nodes) (ln.n_prims > 0 ? *leaves : *interior)++;
}

void BuildBVH(const std::vector<Bounds3> &primBounds, int maxPrimsInNode, SplitMethod method,
              std::vector<mi_bvh_node> *nodes, std::vector<int> *orderedPrims, int *interior, int *leaves) {
    nodes->clear();
    orderedPrims->clear();
    *interior = *leaves = 0;
    if (primBounds.empty()) return;
    std::vector<PrimInfo> info(primBounds.size());
    for (size_t i = 0; i < primBounds.size(); ++i) {
        info[i].primitiveNumber = i;
        info[i].bounds = primBounds[i];
        info[i].centroid = .5f * primBounds[i].pMin + .5f * primBounds[i].pMax;  // bvh.cpp:56
    }
    const int n = (int)primBounds.size();
    orderedPrims->assign(n, 0);
    unsigned nThreads = std::max(1u, std::thread::hardware_concurrency());
    if (const char *e = getenv("MIPT_BUILD_THREADS")) nThreads = (unsigned)std::max(1, atoi(e));
    nThreads = std::min(nThreads, 64u);
    if (n < 65536) nThreads = 1;
    Builder top;
    top.maxPrimsInNode = std::min(255, maxPrimsInNode);
    top.method = method;
    top.ordered = orderedPrims->data();
    std::vector<DeferredSubtree> deferred;
    if (nThreads > 1) { top.grain = std::max(4096, n / (int)(nThreads * 8)); top.deferred = &deferred; }
    BuildNode *root = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    top.Child(&root, info, 0, n);
    const auto t1 = std::chrono::steady_clock::now();
    int total = top.total, nInterior = top.interior, nLeaves = top.leaves;
    std::vector<std::unique_ptr<Builder>> workers(nThreads);   // one per thread (its nodes must outlive the flatten pass)
    if (!deferred.empty()) {
        std::atomic<size_t> next{0};
        auto run = [&](unsigned t) {
            workers[t].reset(new Builder());
            Builder &w = *workers[t];
            w.maxPrimsInNode = top.maxPrimsInNode;
            w.method = method;
            w.ordered = orderedPrims->data();
            for (size_t i; (i = next.fetch_add(1)) < deferred.size();) {
                w.orderedNext = deferred[i].orderedBase;
                *deferred[i].slot = w.Build(info, deferred[i].start, deferred[i].end);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(run, t);
        run(0);
        for (std::thread &t : pool) t.join();
        for (const auto &w : workers) if (w) { total += w->total; nInterior += w->interior; nLeaves += w->leaves; }
    }
    const auto t2 = std::chrono::steady_clock::now();
    nodes->resize(total);
    int offset = 0;
    Flatten(root, *nodes, &offset);
    if (getenv("MIPT_TIMING")) {
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        fprintf(stderr, "[mipt] BVH: top %.3f s, %zu subtrees on %u threads %.3f s, flatten %.3f s\n", sec(t0, t1), deferred.size(), nThreads,
                sec(t1, t2), sec(t2, std::chrono::steady_clock::now()));
    }
    *interior = nInterior;
    *leaves = nLeaves;
}

}  // namespace mipt
