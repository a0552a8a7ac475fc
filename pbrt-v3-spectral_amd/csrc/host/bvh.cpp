// bvh.cpp -- host BVH2 build (SAH with 12 buckets / middle / equal counts) and
// depth-first flatten into 32-byte nodes. The partitioning decisions, including the
// use of std::partition / std::nth_element on the primitive-info array, follow
// BVHAccel::recursiveBuild and flattenBVHTree (src/accelerators/bvh.cpp:236-402,
// 640-658) so the tree (node count, leaf contents, primitive order) is the
// reference's tree: killeroo-simple -> 59 188 interior + 59 189 leaf nodes.
// HLBVH (bvh.cpp:404-638) is a "next" row.
#include <algorithm>
#include <memory>
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

struct Builder {
    int maxPrimsInNode;
    SplitMethod method;
    std::vector<std::unique_ptr<BuildNode>> pool;
    std::vector<int> *ordered;
    int interior = 0, leaves = 0, total = 0;

    BuildNode *Alloc() {
        pool.emplace_back(new BuildNode());
        return pool.back().get();
    }
    void InitLeaf(BuildNode *node, std::vector<PrimInfo> &info, int start, int end, const Bounds3 &b) {
        int first = (int)ordered->size();
        for (int i = start; i < end; ++i) ordered->push_back((int)info[i].primitiveNumber);
        node->firstPrimOffset = first;
        node->nPrimitives = end - start;
        node->bounds = b;
        ++leaves;
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
        BuildNode *c0 = Build(info, start, mid);
        BuildNode *c1 = Build(info, mid, end);
        node->children[0] = c0;
        node->children[1] = c1;
        node->bounds = Union(c0->bounds, c1->bounds);
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
    Builder b;
    b.maxPrimsInNode = std::min(255, maxPrimsInNode);
    b.method = method;
    b.ordered = orderedPrims;
    orderedPrims->reserve(primBounds.size());
    BuildNode *root = b.Build(info, 0, (int)primBounds.size());
    nodes->resize(b.total);
    int offset = 0;
    Flatten(root, *nodes, &offset);
    *interior = b.interior;
    *leaves = b.leaves;
}

}  // namespace mipt
