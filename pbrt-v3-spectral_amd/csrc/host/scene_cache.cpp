// scene_cache.cpp -- a loaded scene as one binary file, so that the ranks of a multi-GPU job (one process per GPU) do not
// each parse the same .pbrt text and build the same BVH: rank 0 loads and saves, the others read the arrays back.
// (The reference is one process: its pbrtWorldEnd() builds the scene once, src/core/api.cpp:1617-1737.)
// Layout: magic, ABI version, sizeof checks, the POD part of mi_scene_desc (pointers are rebuilt by
// HostScene::Finalize), then every owning array as {uint64 count, bytes}. Host-endian, same-build only: a cache is a
// hand-over between processes of one job, not an interchange format.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include "scene.h"

namespace mipt {
namespace {

const char kMagic[8] = {'M', 'I', 'P', 'T', 'S', 'C', '0', '5'};

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
        // (by division: n * sizeof(T) wraps for a forged count, and the resize below would throw past every owner)
        if (!ok || left < 0 || n > (uint64_t)left / sizeof(T)) { ok = false; return; }
        v.resize((size_t)n);
        Raw(v.data(), (size_t)n * sizeof(T));
    }
    void Str(std::string &s) {
        uint64_t n = 0;
        Pod(n);
        if (!ok || left < 0 || n > (uint64_t)left) { ok = false; return; }
        s.resize((size_t)n);
        Raw(&s[0], (size_t)n);
    }
    void Strs(std::vector<std::string> &v) {
        uint64_t n = 0;
        Pod(n);
        if (!ok || left < 0 || n > (uint64_t)left / sizeof(uint64_t)) { ok = false; return; }   // (every string carries its 8-byte length)
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
    // a new file of this process's own (O_EXCL: never somebody else's file or link, never followed), readable by the owner only
    const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    FILE *f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) close(fd); *err = "cannot create " + tmp; return false; }
    Out o{f};
    o.Raw(kMagic, 8);
    o.Pod<uint32_t>(MI_ABI_VERSION);
    o.Pod<uint32_t>((uint32_t)sizeof(mi_scene_desc));
    o.Pod<uint32_t>((uint32_t)sizeof(mi_material));
    o.Pod<uint32_t>((uint32_t)sizeof(mi_light));
    o.Pod(scene.desc);
    o.Pod(scene.stats);
    o.Pod<uint8_t>(scene.spectralFlag ? 1 : 0);
    o.Pod<uint8_t>(scene.hlbvhOnDevice ? 1 : 0);
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
    // a regular file of this user (a cache is a hand-over between the processes of one job), opened without following a link
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid()) {
        if (fd >= 0) close(fd);
        *err = "cannot open scene cache \"" + path + "\" (a regular file owned by this user is expected)";
        return nullptr;
    }
    struct Closer { void operator()(FILE *f) const { if (f) fclose(f); } };
    std::unique_ptr<FILE, Closer> file(fdopen(fd, "rb"));
    if (!file) { close(fd); *err = "cannot open scene cache \"" + path + "\""; return nullptr; }
    FILE *f = file.get();
    In in{f, (long long)st.st_size};
    char magic[8];
    uint32_t abi = 0, sDesc = 0, sMat = 0, sLight = 0;
    in.Raw(magic, 8); in.Pod(abi); in.Pod(sDesc); in.Pod(sMat); in.Pod(sLight);
    if (!in.ok || memcmp(magic, kMagic, 8) != 0 || abi != MI_ABI_VERSION || sDesc != sizeof(mi_scene_desc) ||
        sMat != sizeof(mi_material) || sLight != sizeof(mi_light)) {
        *err = "\"" + path + "\" is not a scene cache of this build";
        return nullptr;
    }
    std::unique_ptr<HostScene> s(new HostScene());
    uint8_t spectral = 1, onDevice = 0;
    in.Pod(s->desc); in.Pod(s->stats); in.Pod(spectral); in.Pod(onDevice);
    s->spectralFlag = spectral != 0;
    s->hlbvhOnDevice = onDevice != 0;
    Fields(in, *s);
    uint64_t n = 0;
    in.Pod(n);
    if (in.ok && in.left >= 0 && n <= (uint64_t)in.left / 16) {   // (an environment map is at least its five header words)
        s->envStore.resize((size_t)n);
        for (HostEnvMap &e : s->envStore) {
            in.Pod(e.width); in.Pod(e.height); in.Pod(e.nu); in.Pod(e.nv); in.Pod(e.margFuncInt);
            in.Vec(e.rgb); in.Vec(e.condFunc); in.Vec(e.condCdf); in.Vec(e.condFuncInt); in.Vec(e.margFunc); in.Vec(e.margCdf);
        }
    } else in.ok = false;
    n = 0;
    in.Pod(n);
    if (in.ok && in.left >= 0 && n <= (uint64_t)in.left / 16) {
        s->mipStore.resize((size_t)n);
        for (HostMipMap &m : s->mipStore) { in.Str(m.key); in.Pod(m.width); in.Pod(m.height); in.Pod(m.wrap); in.Vec(m.texels); in.Vec(m.levelOffset); }
    } else in.ok = false;
    // Every array must have the size the description (and the consumers of the description: mi_pt_create copies count-derived
    // sizes from these pointers) assumes -- a truncated, foreign or forged file does not.
    const mi_scene_desc &d = s->desc;
    bool consistent = in.ok && s->nodes.size() == d.n_nodes && s->prims.size() == d.n_prims && s->triIndices.size() == 3ull * d.n_tris &&
                      s->triMesh.size() == d.n_tris && s->P.size() == 3ull * d.n_verts && s->N.size() == 3ull * d.n_verts &&
                      s->UV.size() == 2ull * d.n_verts && s->meshes.size() == d.n_meshes && s->spheres.size() == d.n_spheres &&
                      s->materials.size() == d.n_materials && s->lights.size() == d.n_lights && s->textures.size() == d.n_textures && s->instances.size() == d.n_instances &&
                      s->envStore.size() == d.n_envmaps && s->mipStore.size() == d.n_mipmaps && d.sampler.n_dims >= 0 && (int)s->primes.size() == d.sampler.n_dims &&
                      s->primeSums.size() == s->primes.size() && s->perms.size() == d.sampler.n_perms;
    if (consistent) {   // light-selection tables (mi_lightdistrib): UNIFORM / POWER carry one distribution, SPATIAL none (built on the device)
        if (d.n_lights == 0 || d.light_distrib.type == MI_LD_SPATIAL) consistent = s->ldFunc.empty() && s->ldCdf.empty() && s->ldFuncInt.empty();
        else consistent = s->ldFunc.size() == d.n_lights && s->ldCdf.size() == (size_t)d.n_lights + 1 && s->ldFuncInt.size() == 1;
        if (d.light_distrib.type == MI_LD_SPATIAL)
            for (int a = 0; a < 3; ++a) consistent = consistent && d.light_distrib.n_voxels[a] >= 1 && d.light_distrib.n_voxels[a] <= 4096;
    }
    if (consistent) {   // Sobol' tables (mi_sampler)
        if (d.sampler.type == MI_SAMPLER_SOBOL)
            consistent = d.sampler.n_sobol_dims >= 0 && s->sobolMatrices.size() == (size_t)d.sampler.n_sobol_dims * MI_SOBOL_MATRIX_SIZE &&
                         s->sobolVdc.size() == MI_SOBOL_MATRIX_SIZE && s->sobolVdcInv.size() == MI_SOBOL_MATRIX_SIZE;
        else consistent = s->sobolMatrices.empty() && s->sobolVdc.empty() && s->sobolVdcInv.empty();
    }
    for (size_t i = 0; consistent && i < s->primes.size(); ++i)
        consistent = s->primes[i] >= 2 && s->primeSums[i] >= 0 && (size_t)s->primeSums[i] + (size_t)s->primes[i] <= s->perms.size();
    for (size_t i = 0; consistent && i < s->envStore.size(); ++i) {   // mi_envmap
        const HostEnvMap &e = s->envStore[i];
        consistent = e.width >= 1 && e.height >= 1 && e.nu >= 1 && e.nv >= 1 && e.width <= (1 << 16) && e.height <= (1 << 16) && e.nu <= (1 << 17) && e.nv <= (1 << 17) &&
                     e.rgb.size() == 3ull * e.width * e.height && e.condFunc.size() == (size_t)e.nu * e.nv && e.condCdf.size() == ((size_t)e.nu + 1) * e.nv &&
                     e.condFuncInt.size() == (size_t)e.nv && e.margFunc.size() == (size_t)e.nv && e.margCdf.size() == (size_t)e.nv + 1;
    }
    for (size_t i = 0; consistent && i < s->mipStore.size(); ++i) {   // mi_mipmap: every level inside the texel array
        const HostMipMap &m = s->mipStore[i];
        consistent = m.width >= 1 && m.height >= 1 && !m.levelOffset.empty() && m.levelOffset.size() <= MI_MAX_MIP_LEVELS && m.texels.size() % 3 == 0;
        for (size_t l = 0; consistent && l < m.levelOffset.size(); ++l) {
            const size_t w = (size_t)std::max(1, m.width >> l), h = (size_t)std::max(1, m.height >> l);
            consistent = ((size_t)m.levelOffset[l] + w * h) * 3 <= m.texels.size();
        }
    }
    if (!consistent) { *err = "scene cache \"" + path + "\" is truncated or inconsistent"; return nullptr; }
    s->Finalize();
    return s.release();
}

}  // namespace mipt
