// scene_cache.cpp -- a loaded scene as one binary file, so that the ranks of a multi-GPU job (one process per GPU) do not
// each parse the same .pbrt text and build the same BVH: rank 0 loads and saves, the others read the arrays back.
// (The reference is one process: its pbrtWorldEnd() builds the scene once, src/core/api.cpp:1617-1737.)
// Layout: magic, ABI version, sizeof checks, the POD part of mi_scene_desc (pointers are rebuilt by
// HostScene::Finalize), then every owning array as {uint64 count, bytes}. Host-endian, same-build only: a cache is a
// hand-over between processes of one job, not an interchange format.
#include <cstdio>
#include <cstring>
#include "scene.h"

namespace mipt {
namespace {

const char kMagic[8] = {'M', 'I', 'P', 'T', 'S', 'C', '0', '4'};

struct Out {
    FILE *f;
    bool ok = true;
    void Raw(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    template <typename T> void Pod(const T &v) { Raw(&v, sizeof(T)); }
    template <typename T> void Vec(const std::vector<T> &v) { Pod<uint64_t>(v.size()); Raw(v.data(), v.size() * sizeof(T)); }
    void Str(const std::string &s) { Pod<uint64_t>(s.size()); Raw(s.data(), s.size()); }
    void Strs(const std::vector<std::string> &v) { Pod<uint64_t>(v.size()); for (const auto &s : v) Str(s); }
};
struct In {
    FILE *f;
    long long left;   // bytes of the file not yet read: a count cannot promise more than that
    bool ok = true;
    void Raw(void *p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; left -= (long long)n; }
    template <typename T> void Pod(T &v) { Raw(&v, sizeof(T)); }
    template <typename T> void Vec(std::vector<T> &v) {
        uint64_t n = 0;
        Pod(n);
        if (!ok || (long long)(n * sizeof(T)) > left) { ok = false; return; }
        v.resize((size_t)n);
        Raw(v.data(), (size_t)n * sizeof(T));
    }
    void Str(std::string &s) {
        uint64_t n = 0;
        Pod(n);
        if (!ok || (long long)n > left) { ok = false; return; }
        s.resize((size_t)n);
        Raw(&s[0], (size_t)n);
    }
    void Strs(std::vector<std::string> &v) {
        uint64_t n = 0;
        Pod(n);
        if (!ok || (long long)n > left) { ok = false; return; }
        v.resize((size_t)n);
        for (auto &s : v) Str(s);
    }
};

template <typename IO, typename S>
void Fields(IO &io, S &s) {   // the same walk writes and reads
    io.Vec(s.nodes); io.Vec(s.prims); io.Vec(s.triIndices); io.Vec(s.triMesh);
    io.Vec(s.P); io.Vec(s.N); io.Vec(s.UV);
    io.Vec(s.meshes); io.Vec(s.spheres); io.Vec(s.materials); io.Vec(s.lights); io.Vec(s.textures); io.Vec(s.instances);
    io.Vec(s.ldFunc); io.Vec(s.ldCdf); io.Vec(s.ldFuncInt);
    io.Vec(s.primes); io.Vec(s.primeSums); io.Vec(s.perms);
    io.Vec(s.sobolMatrices); io.Vec(s.sobolVdc); io.Vec(s.sobolVdcInv);
    io.Str(s.filmFilename); io.Str(s.integratorName); io.Str(s.samplerName); io.Str(s.lightStrategy);
    io.Strs(s.warnings); io.Strs(s.errors);
}

}  // namespace

bool SaveSceneCache(const HostScene &scene, const std::string &path, std::string *err) {
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { *err = "cannot open " + tmp; return false; }
    Out o{f};
    o.Raw(kMagic, 8);
    o.Pod<uint32_t>(MI_ABI_VERSION);
    o.Pod<uint32_t>((uint32_t)sizeof(mi_scene_desc));
    o.Pod<uint32_t>((uint32_t)sizeof(mi_material));
    o.Pod<uint32_t>((uint32_t)sizeof(mi_light));
    o.Pod(scene.desc);
    o.Pod(scene.stats);
    o.Pod<uint8_t>(scene.spectralFlag ? 1 : 0);
    Fields(o, const_cast<HostScene &>(scene));
    o.Pod<uint64_t>(scene.envStore.size());
    for (const HostEnvMap &e : scene.envStore) {
        o.Pod(e.width); o.Pod(e.height); o.Pod(e.nu); o.Pod(e.nv); o.Pod(e.margFuncInt);
        o.Vec(e.rgb); o.Vec(e.condFunc); o.Vec(e.condCdf); o.Vec(e.condFuncInt); o.Vec(e.margFunc); o.Vec(e.margCdf);
    }
    o.Pod<uint64_t>(scene.mipStore.size());
    for (const HostMipMap &m : scene.mipStore) {
        o.Str(m.key); o.Pod(m.width); o.Pod(m.height); o.Pod(m.wrap); o.Vec(m.texels); o.Vec(m.levelOffset);
    }
    const bool ok = o.ok && fclose(f) == 0;
    if (!ok) { *err = "short write to " + tmp; remove(tmp.c_str()); return false; }
    if (rename(tmp.c_str(), path.c_str()) != 0) { *err = "cannot rename " + tmp; return false; }   // readers never see half a file
    return true;
}

HostScene *LoadSceneCache(const std::string &path, std::string *err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { *err = "cannot open scene cache \"" + path + "\""; return nullptr; }
    fseek(f, 0, SEEK_END);
    In in{f, (long long)ftell(f)};
    fseek(f, 0, SEEK_SET);
    char magic[8];
    uint32_t abi = 0, sDesc = 0, sMat = 0, sLight = 0;
    in.Raw(magic, 8); in.Pod(abi); in.Pod(sDesc); in.Pod(sMat); in.Pod(sLight);
    if (!in.ok || memcmp(magic, kMagic, 8) != 0 || abi != MI_ABI_VERSION || sDesc != sizeof(mi_scene_desc) ||
        sMat != sizeof(mi_material) || sLight != sizeof(mi_light)) {
        fclose(f);
        *err = "\"" + path + "\" is not a scene cache of this build";
        return nullptr;
    }
    HostScene *s = new HostScene();
    uint8_t spectral = 1;
    in.Pod(s->desc); in.Pod(s->stats); in.Pod(spectral);
    s->spectralFlag = spectral != 0;
    Fields(in, *s);
    uint64_t n = 0;
    in.Pod(n);
    if (in.ok && (long long)n <= in.left) {
        s->envStore.resize((size_t)n);
        for (HostEnvMap &e : s->envStore) {
            in.Pod(e.width); in.Pod(e.height); in.Pod(e.nu); in.Pod(e.nv); in.Pod(e.margFuncInt);
            in.Vec(e.rgb); in.Vec(e.condFunc); in.Vec(e.condCdf); in.Vec(e.condFuncInt); in.Vec(e.margFunc); in.Vec(e.margCdf);
        }
    } else in.ok = false;
    n = 0;
    in.Pod(n);
    if (in.ok && (long long)n <= in.left) {
        s->mipStore.resize((size_t)n);
        for (HostMipMap &m : s->mipStore) { in.Str(m.key); in.Pod(m.width); in.Pod(m.height); in.Pod(m.wrap); in.Vec(m.texels); in.Vec(m.levelOffset); }
    } else in.ok = false;
    fclose(f);
    // the arrays must agree with the counts the description was saved with (a truncated or foreign file does not)
    const mi_scene_desc &d = s->desc;
    const bool consistent = in.ok && s->nodes.size() == d.n_nodes && s->prims.size() == d.n_prims && s->triIndices.size() == 3ull * d.n_tris &&
                            s->triMesh.size() == d.n_tris && s->P.size() == 3ull * d.n_verts && s->N.size() == 3ull * d.n_verts &&
                            s->UV.size() == 2ull * d.n_verts && s->meshes.size() == d.n_meshes && s->spheres.size() == d.n_spheres &&
                            s->materials.size() == d.n_materials && s->lights.size() == d.n_lights && s->textures.size() == d.n_textures && s->instances.size() == d.n_instances &&
                            s->envStore.size() == d.n_envmaps && s->mipStore.size() == d.n_mipmaps && (int)s->primes.size() == d.sampler.n_dims &&
                            s->primeSums.size() == s->primes.size() && s->perms.size() == d.sampler.n_perms;
    if (!consistent) { delete s; *err = "scene cache \"" + path + "\" is truncated or inconsistent"; return nullptr; }
    s->Finalize();
    return s;
}

}  // namespace mipt
