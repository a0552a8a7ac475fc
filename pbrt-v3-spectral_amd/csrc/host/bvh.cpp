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
