// scene.h -- the host front end's flat scene: owning storage behind the POD
// mi_scene_desc that crosses the C ABI (include/mi_pt.h). Produced by the .pbrt
// front end (api.cpp) the way pbrtWorldEnd() produces Scene + Integrator in the
// reference (src/core/api.cpp:1617-1737), but as flat arrays because the
// reference's Scene/BVHAccel internals are private.
#pragma once
#include <string>
#include <vector>
#include "../../../include/mi_pt.h"
#include "paramset.h"

namespace mipt {

struct SceneStats {
    int nTriangles = 0, nSpheres = 0, nMeshes = 0;
    int interiorNodes = 0, leafNodes = 0;
    int nLights = 0, nMaterials = 0;
};

// LightSource "infinite" (envmap.cpp)
struct HostEnvMap {
    int width = 0, height = 0, nu = 0, nv = 0;
    std::vector<float> rgb, condFunc, condCdf, condFuncInt, margFunc, margCdf;
    float margFuncInt = 0;
};

// Texture "..." "spectrum" "imagemap": one MIPMap pyramid per distinct TexInfo (imagemap.h:50-76)
struct HostMipMap {
    std::string key;           // filename + filter parameters (the reference's texture cache key)
    int width = 0, height = 0, wrap = 0;
    std::vector<float> texels; // RGB, all levels
    std::vector<uint32_t> levelOffset;
};

struct HostScene {
    // geometry
    std::vector<mi_bvh_node> nodes;
    std::vector<mi_prim> prims;  // BVH leaf order
    std::vector<int32_t> triIndices;
    std::vector<uint32_t> triMesh;
    std::vector<float> P, N, UV;
    std::vector<mi_mesh> meshes;
    std::vector<mi_sphere> spheres;
    std::vector<mi_material> materials;
    std::vector<mi_light> lights;
    std::vector<HostEnvMap> envStore;   // storage behind desc.envmaps
    std::vector<mi_envmap> envmaps;
    std::vector<HostMipMap> mipStore;   // storage behind desc.mipmaps
    std::vector<mi_mipmap> mipmaps;
    std::vector<mi_texture> textures;
    std::vector<mi_instance> instances;   // ObjectInstance as TransformedPrimitive (their BVHs follow the world's in `nodes`)
    // light distribution
    std::vector<float> ldFunc, ldCdf, ldFuncInt;
    // sampler tables
    std::vector<int32_t> primes, primeSums;
    std::vector<uint16_t> perms;
    std::vector<uint32_t> sobolMatrices;          // Sampler "sobol": the first n_sobol_dims generator matrices
    std::vector<uint64_t> sobolVdc, sobolVdcInv;  // ... and the pixel-index matrices of the film's resolution
    // output
    std::string filmFilename = "pbrt.exr";
    bool spectralFlag = true;
    bool hlbvhOnDevice = false;   // splitmethod "hlbvh": the tree came from the device build
    std::string integratorName = "path", samplerName = "halton", lightStrategy = "spatial";
    std::vector<std::string> warnings, errors;
    SceneStats stats;
    mi_scene_desc desc{};

    void Finalize();  // point desc at the vectors
};

// ---- builders (each cites the reference routine it restates)
// Loop subdivision to the limit surface, src/shapes/loopsubdiv.cpp:149-400.
bool LoopSubdivide(int nLevels, const std::vector<int> &indices, const std::vector<Vec3> &P,
                   std::vector<int> *outIndices, std::vector<Vec3> *outP, std::vector<Vec3> *outN,
                   std::string *err);

// SAH BVH2 build + depth-first flatten, src/accelerators/bvh.cpp:183-402,640-658.
struct BuildPrim { Bounds3 bounds; };
enum class SplitMethod { SAH, Middle, EqualCounts, HLBVH };
void BuildBVH(const std::vector<Bounds3> &primBounds, int maxPrimsInNode, SplitMethod method,
              std::vector<mi_bvh_node> *nodes, std::vector<int> *orderedPrims, int *interior,
              int *leaves);

// HLBVH, src/accelerators/bvh.cpp:404-638, in the order one thread builds it (bvh.cpp in this directory). The pieces are
// exposed because the device build (mi_bvh_build_hlbvh) shares the upper SAH tree and is tested against the host tree.
void BuildHLBVH(const std::vector<Bounds3> &primBounds, int maxPrimsInNode, std::vector<mi_bvh_node> *nodes,
                std::vector<int> *orderedPrims, int *interior, int *leaves);
int BuildUpperSAH(uint32_t nTreelets, const float *rootBounds, const int32_t *treeletSizes, std::vector<mi_bvh_node> *upperNodes,
                  std::vector<int> *upperIndex, int32_t *treeletOffset);   // returns the total node count
// The same tree built by libmipt_hip.so's kernels (hlbvh_device.cpp); false (with `why`) when no device / library is there.
bool BuildHLBVHOnDevice(const std::vector<Bounds3> &primBounds, int maxPrimsInNode, int device, std::vector<mi_bvh_node> *nodes,
                        std::vector<int> *orderedPrims, int *interior, int *leaves, double *seconds, std::string *why);

// Halton tables, src/core/lowdiscrepancy.cpp:2490-2504 (+ rng.h PCG32, sampling.h Shuffle).
void ComputeHaltonTables(int nDims, std::vector<int32_t> *primes, std::vector<int32_t> *primeSums,
                         std::vector<uint16_t> *perms);
float RadicalInverseHost(int baseIndex, uint64_t a);  // unscrambled, lowdiscrepancy.cpp:389-424
// Sobol' tables for a film whose sample bounds span `extent` pixels (sobol.h:51-62); false when the resolution is beyond the tables.
bool ComputeSobolTables(int extent, int nDims, HostScene *scene, int *resolution, int *log2Resolution);

// Light-selection distributions, src/core/lightdistrib.cpp:48-300.
void BuildLightDistribution(HostScene *scene, const std::string &strategy);

// Material compilation (constant textures -> fixed BxDF list),
// src/materials/{matte,plastic,glass,uber,disney,mirror}.cpp.
// Returns false (and appends to errs) for materials this path does not cover.
// Shape "plymesh" (plymesh.cpp)
struct PLYMeshData {
    std::vector<int> indices;
    std::vector<Vec3> P, N;
    std::vector<Vec2> UV;
};
bool ReadPLYMesh(const std::string &filename, PLYMeshData *out, std::vector<std::string> *warnings, std::string *err);

// ImageTexture<RGBSpectrum, Spectrum>::GetTexture (imagemap.cpp:58-106): read, flip, convertIn, build the pyramid.
// Returns the index into scene->mipStore (cached by TexInfo).
int BuildTextureMipMap(HostScene *scene, const std::string &filename, bool trilinear, bool noFiltering, float maxAniso,
                       int wrap, float scale, bool gamma, bool isFloat = false);
int ConstantFloatMipMap(HostScene *scene, float value);   // ConstantTexture<Float> as a 1x1 pyramid
bool BuildEnvMap(const Spectrum &L, const std::string &texmap, HostEnvMap *store, Spectrum *centre, std::vector<std::string> *errors);
bool WriteRGBImage(const std::string &filename, int w, int h, const float *filmSum, const float *weightSum, float scale,
                   std::string *written, std::string *err);
bool CompileMixMaterial(const mi_material &m1, const mi_material &m2, const Spectrum &amount, mi_material *out,
                        std::vector<std::string> *errs);
bool CompileMaterial(const std::string &type, const TextureParams &mp, mi_material *out,
                     std::vector<std::string> *warnings, std::vector<std::string> *errs);

// .pbrt front end. Overrides < 0 / empty leave the file's values.
struct LoadOverrides {
    int spp = -1, xres = -1, yres = -1, maxDepth = -1;
    float crop[4] = {-1, -1, -1, -1};
    std::string lightStrategy;
};
HostScene *LoadSceneFile(const std::string &path, const LoadOverrides &ov, std::string *err);
HostScene *LoadSceneString(const std::string &text, const std::string &baseDir, const LoadOverrides &ov,
                           std::string *err);

// One loaded scene as a binary file (scene_cache.cpp): the hand-over between the ranks of a multi-GPU job.
bool SaveSceneCache(const HostScene &scene, const std::string &path, std::string *err);
HostScene *LoadSceneCache(const std::string &path, std::string *err);

// Spectral film writer, src/core/film.cpp:226-308 (".dat": text header + 31 planes of float64).
bool WriteSpectralDat(const std::string &filename, int w, int h, const float *filmSum, float scale,
                      std::string *err);

}  // namespace mipt
