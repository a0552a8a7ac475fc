// loopsubdiv.cpp -- Loop subdivision surfaces -> limit-surface triangle mesh with
// limit normals. Index-based half-structure (faces/vertices in std::vector, ints
// instead of pointers); the refinement rules, the visiting order of faces and edges
// and the float expression order follow src/shapes/loopsubdiv.cpp:149-400,426-469 so
// the emitted vertices are the same floats in the same order (killeroo.pbrt is one
// Shape "loopsubdiv" with nlevels 1: 8316 faces -> 33264 triangles).
#include <cmath>
#include <map>
#include <set>
#include "scene.h"

namespace mipt {
namespace {

#define NEXT(i) (((i) + 1) % 3)
#define PREV(i) (((i) + 2) % 3)

struct SDVert {
    Vec3 p;
    int startFace = -1;
    int child = -1;
    bool regular = false, boundary = false;
};
struct SDFace {
    int v[3] = {-1, -1, -1};
    int f[3] = {-1, -1, -1};
    int children[4] = {-1, -1, -1, -1};
};
// Edge keyed by the (min,max) of its end-vertex ids. The reference keys on pointer
// order (loopsubdiv.cpp:97-101); ids are allocation order, and the only
// order-dependent use (v[0]/v[1] in the odd-vertex rule) adds two products, which
// commutes exactly.
struct SDEdge {
    int v[2];
    int f0 = -1, f0edgeNum = -1;
    SDEdge(int v0 = -1, int v1 = -1) { v[0] = std::min(v0, v1); v[1] = std::max(v0, v1); }
    bool operator<(const SDEdge &e2) const {
        if (v[0] == e2.v[0]) return v[1] < e2.v[1];
        return v[0] < e2.v[0];
    }
};

struct Mesh {
    std::vector<SDVert> V;
    std::vector<SDFace> F;
    int vnum(int face, int vert) const {
        for (int i = 0; i < 3; ++i)
            if (F[face].v[i] == vert) return i;
        return -1;
    }
    int nextFace(int face, int vert) const { return F[face].f[vnum(face, vert)]; }
    int prevFace(int face, int vert) const { return F[face].f[PREV(vnum(face, vert))]; }
    int nextVert(int face, int vert) const { return F[face].v[NEXT(vnum(face, vert))]; }
    int prevVert(int face, int vert) const { return F[face].v[PREV(vnum(face, vert))]; }
    int otherVert(int face, int v0, int v1) const {
        for (int i = 0; i < 3; ++i)
            if (F[face].v[i] != v0 && F[face].v[i] != v1) return F[face].v[i];
        return -1;
    }
    int valence(int vi) const {  // loopsubdiv.cpp:120-136
        const SDVert &vert = V[vi];
        int f = vert.startFace;
        if (!vert.boundary) {
            int nf = 1;
            while ((f = nextFace(f, vi)) != vert.startFace) ++nf;
            return nf;
        } else {
            int nf = 1;
            while ((f = nextFace(f, vi)) != -1) ++nf;
            f = vert.startFace;
            while ((f = prevFace(f, vi)) != -1) ++nf;
            return nf + 1;
        }
    }
    void oneRing(int vi, Vec3 *p) const {  // loopsubdiv.cpp:438-457
        const SDVert &vert = V[vi];
        if (!vert.boundary) {
            int face = vert.startFace;
            do {
                *p++ = V[nextVert(face, vi)].p;
                face = nextFace(face, vi);
            } while (face != vert.startFace);
        } else {
            int face = vert.startFace, f2;
            while ((f2 = nextFace(face, vi)) != -1) face = f2;
            *p++ = V[nextVert(face, vi)].p;
            do {
                *p++ = V[prevVert(face, vi)].p;
                face = prevFace(face, vi);
            } while (face != -1);
        }
    }
    Vec3 weightOneRing(int vi, float beta) const {  // loopsubdiv.cpp:426-436
        int val = valence(vi);
        std::vector<Vec3> ring(val);
        oneRing(vi, ring.data());
        Vec3 p = (1 - val * beta) * V[vi].p;
        for (int i = 0; i < val; ++i) p += beta * ring[i];
        return p;
    }
    Vec3 weightBoundary(int vi, float beta) const {  // loopsubdiv.cpp:459-469
        int val = valence(vi);
        std::vector<Vec3> ring(val);
        oneRing(vi, ring.data());
        Vec3 p = (1 - 2 * beta) * V[vi].p;
        p += beta * ring[0];
        p += beta * ring[val - 1];
        return p;
    }
};

inline float betaf(int valence) {  // loopsubdiv.cpp:138-143
    if (valence == 3) return 3.f / 16.f;
    else return 3.f / (8.f * valence);
}
inline float loopGamma(int valence) { return 1.f / (valence + 3.f / (8.f * betaf(valence))); }

}  // namespace

bool LoopSubdivide(int nLevels, const std::vector<int> &indices, const std::vector<Vec3> &P,
                   std::vector<int> *outIndices, std::vector<Vec3> *outP, std::vector<Vec3> *outN,
                   std::string *err) {
    Mesh M;
    const int nVertices = (int)P.size();
    const int nFaces = (int)indices.size() / 3;
    for (int idx : indices)
        if (idx < 0 || idx >= nVertices) { *err = "loopsubdiv: vertex index out of range"; return false; }
    M.V.resize(nVertices);
    for (int i = 0; i < nVertices; ++i) M.V[i].p = P[i];
    M.F.resize(nFaces);
    // Level-0 faces are [0,nFaces), vertices [0,nVertices): lists of the current level
    std::vector<int> f(nFaces), v(nVertices);
    for (int i = 0; i < nFaces; ++i) f[i] = i;
    for (int i = 0; i < nVertices; ++i) v[i] = i;

    // face -> vertex pointers (last referencing face becomes startFace), :170-178
    for (int i = 0; i < nFaces; ++i)
        for (int j = 0; j < 3; ++j) {
            int vi = indices[3 * i + j];
            M.F[i].v[j] = vi;
            M.V[vi].startFace = i;
        }
    // neighbour pointers, :181-200
    {
        std::set<SDEdge> edges;
        for (int i = 0; i < nFaces; ++i) {
            for (int edgeNum = 0; edgeNum < 3; ++edgeNum) {
                int v0 = edgeNum, v1 = NEXT(edgeNum);
                SDEdge e(M.F[i].v[v0], M.F[i].v[v1]);
                auto it = edges.find(e);
                if (it == edges.end()) {
                    e.f0 = i;
                    e.f0edgeNum = edgeNum;
                    edges.insert(e);
                } else {
                    e = *it;
                    M.F[e.f0].f[e.f0edgeNum] = i;
                    M.F[i].f[edgeNum] = e.f0;
                    edges.erase(it);
                }
            }
        }
    }
    // finish vertex initialisation, :203-217
    for (int i = 0; i < nVertices; ++i) {
        SDVert &vert = M.V[i];
        if (vert.startFace < 0) { *err = "loopsubdiv: unreferenced vertex"; return false; }
        int face = vert.startFace;
        do {
            face = M.nextFace(face, i);
        } while (face != -1 && face != vert.startFace);
        vert.boundary = (face == -1);
        if (!vert.boundary && M.valence(i) == 6) vert.regular = true;
        else if (vert.boundary && M.valence(i) == 4) vert.regular = true;
        else vert.regular = false;
    }

    for (int level = 0; level < nLevels; ++level) {
        std::vector<int> newFaces, newVertices;
        // allocate children, :230-241
        for (int vi : v) {
            int c = (int)M.V.size();
            M.V.push_back(SDVert());
            M.V[vi].child = c;
            M.V[c].regular = M.V[vi].regular;
            M.V[c].boundary = M.V[vi].boundary;
            newVertices.push_back(c);
        }
        for (int fi : f)
            for (int k = 0; k < 4; ++k) {
                int c = (int)M.F.size();
                M.F.push_back(SDFace());
                M.F[fi].children[k] = c;
                newFaces.push_back(c);
            }
        // even vertices, :246-259
        for (int vi : v) {
            Vec3 np;
            if (!M.V[vi].boundary) {
                if (M.V[vi].regular) np = M.weightOneRing(vi, 1.f / 16.f);
                else np = M.weightOneRing(vi, betaf(M.valence(vi)));
            } else
                np = M.weightBoundary(vi, 1.f / 8.f);
            M.V[M.V[vi].child].p = np;
        }
        // odd (edge) vertices, :262-293
        std::map<SDEdge, int> edgeVerts;
        for (int fi : f) {
            for (int k = 0; k < 3; ++k) {
                SDEdge edge(M.F[fi].v[k], M.F[fi].v[NEXT(k)]);
                auto it = edgeVerts.find(edge);
                if (it == edgeVerts.end()) {
                    int nv = (int)M.V.size();
                    M.V.push_back(SDVert());
                    newVertices.push_back(nv);
                    SDVert &vert = M.V[nv];
                    vert.regular = true;
                    vert.boundary = (M.F[fi].f[k] == -1);
                    vert.startFace = M.F[fi].children[3];
                    if (vert.boundary) {
                        vert.p = 0.5f * M.V[edge.v[0]].p;
                        vert.p += 0.5f * M.V[edge.v[1]].p;
                    } else {
                        vert.p = 3.f / 8.f * M.V[edge.v[0]].p;
                        vert.p += 3.f / 8.f * M.V[edge.v[1]].p;
                        vert.p += 1.f / 8.f * M.V[M.otherVert(fi, edge.v[0], edge.v[1])].p;
                        vert.p += 1.f / 8.f * M.V[M.otherVert(M.F[fi].f[k], edge.v[0], edge.v[1])].p;
                    }
                    edgeVerts[edge] = nv;
                }
            }
        }
        // topology, :298-335
        for (int vi : v) {
            int vertNum = M.vnum(M.V[vi].startFace, vi);
            M.V[M.V[vi].child].startFace = M.F[M.V[vi].startFace].children[vertNum];
        }
        for (int fi : f) {
            for (int j = 0; j < 3; ++j) {
                const SDFace face = M.F[fi];
                M.F[face.children[3]].f[j] = face.children[NEXT(j)];
                M.F[face.children[j]].f[NEXT(j)] = face.children[3];
                int f2 = face.f[j];
                M.F[face.children[j]].f[j] = f2 != -1 ? M.F[f2].children[M.vnum(f2, face.v[j])] : -1;
                f2 = face.f[PREV(j)];
                M.F[face.children[j]].f[PREV(j)] = f2 != -1 ? M.F[f2].children[M.vnum(f2, face.v[j])] : -1;
            }
        }
        for (int fi : f) {
            for (int j = 0; j < 3; ++j) {
                const SDFace face = M.F[fi];
                M.F[face.children[j]].v[j] = M.V[face.v[j]].child;
                int vert = edgeVerts[SDEdge(face.v[j], face.v[NEXT(j)])];
                M.F[face.children[j]].v[NEXT(j)] = vert;
                M.F[face.children[NEXT(j)]].v[j] = vert;
                M.F[face.children[3]].v[j] = vert;
            }
        }
        f = newFaces;
        v = newVertices;
    }

    // limit surface, :344-352
    std::vector<Vec3> pLimit(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        if (M.V[v[i]].boundary) pLimit[i] = M.weightBoundary(v[i], 1.f / 5.f);
        else pLimit[i] = M.weightOneRing(v[i], loopGamma(M.valence(v[i])));
    }
    for (size_t i = 0; i < v.size(); ++i) M.V[v[i]].p = pLimit[i];

    // limit normals from tangents, :355-392
    std::vector<Vec3> Ns;
    Ns.reserve(v.size());
    std::vector<Vec3> pRing(16);
    for (int vi : v) {
        Vec3 S(0, 0, 0), T(0, 0, 0);
        int valence = M.valence(vi);
        if (valence > (int)pRing.size()) pRing.resize(valence);
        M.oneRing(vi, pRing.data());
        const SDVert &vertex = M.V[vi];
        if (!vertex.boundary) {
            for (int j = 0; j < valence; ++j) {
                S += std::cos(2 * kPi * j / valence) * pRing[j];
                T += std::sin(2 * kPi * j / valence) * pRing[j];
            }
        } else {
            S = pRing[valence - 1] - pRing[0];
            if (valence == 2) T = pRing[0] + pRing[1] - 2 * vertex.p;
            else if (valence == 3) T = pRing[1] - vertex.p;
            else if (valence == 4)
                T = -1 * pRing[0] + 2 * pRing[1] + 2 * pRing[2] + -1 * pRing[3] + -2 * vertex.p;
            else {
                float theta = kPi / float(valence - 1);
                T = std::sin(theta) * (pRing[0] + pRing[valence - 1]);
                for (int k = 1; k < valence - 1; ++k) {
                    float wt = (2 * std::cos(theta) - 2) * std::sin((k)*theta);
                    T += wt * pRing[k];
                }
                T = -T;
            }
        }
        Ns.push_back(Cross(S, T));
    }

    // triangle mesh, :395-412
    std::map<int, int> usedVerts;
    for (size_t i = 0; i < v.size(); ++i) usedVerts[v[i]] = (int)i;
    outIndices->clear();
    outIndices->reserve(3 * f.size());
    for (size_t i = 0; i < f.size(); ++i)
        for (int j = 0; j < 3; ++j) outIndices->push_back(usedVerts[M.F[f[i]].v[j]]);
    *outP = pLimit;
    *outN = Ns;
    return true;
}

}  // namespace mipt
