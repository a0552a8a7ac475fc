// api.cpp -- the .pbrt scene-file front end for the PathIntegrator hot path:
// tokenizer, typed parameter lists, CTM / graphics-state stacks and the
// directive set the BASELINE configs use; WorldEnd flattens everything into a
// HostScene (scene.h) instead of building pbrt's pointer graph.
//
// Behaviour restated from the reference (same directive names, parameter names,
// defaults, error/warning policy of "report and continue"):
//   tokenizer / parameter typing      src/core/parser.cpp:252-330,440-790
//   directive dispatch                src/core/parser.cpp:786-1090
//   state machine, CTM, attributes    src/core/api.cpp:895-1140,1142-1260
//   Shape / Material / lights         src/core/api.cpp:1264-1443,441-548,747-786
//   WorldEnd -> camera, film, sampler src/core/api.cpp:1617-1737,1750-1860
// Out of scope here (reported as errors, never silently ignored): object
// instancing, animated transforms, participating media, non-constant textures,
// shapes other than trianglemesh / loopsubdiv / sphere.
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include "scene.h"

namespace mipt {
namespace {

// ------------------------------------------------------------------ tokenizer
struct Tokenizer {
    std::string text, filename;
    size_t pos = 0;
    int line = 1;
    bool Next(std::string *tok, bool *quoted, std::string *err) {
        *quoted = false;
        while (pos < text.size()) {
            char ch = text[pos];
            if (ch == '\n') { ++line; ++pos; }
            else if (ch == ' ' || ch == '\t' || ch == '\r') ++pos;
            else if (ch == '#') { while (pos < text.size() && text[pos] != '\n') ++pos; }
            else break;
        }
        if (pos >= text.size()) return false;
        char ch = text[pos];
        if (ch == '"') {
            size_t start = ++pos;
            std::string out;
            bool closed = false;
            while (pos < text.size()) {
                char c = text[pos];
                if (c == '"') { closed = true; break; }
                if (c == '\n') { *err = "premature EOL in string"; return false; }
                if (c == '\\' && pos + 1 < text.size()) {
                    ++pos;
                    char e = text[pos];
                    switch (e) {
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case '\\': out += '\\'; break;
                    case '\'': out += '\''; break;
                    case '"': out += '"'; break;
                    default: *err = "bad escaped character"; return false;
                    }
                } else
                    out += c;
                ++pos;
            }
            (void)start;
            if (!closed) { *err = "premature EOF in string"; return false; }
            ++pos;
            *tok = out;
            *quoted = true;
            return true;
        }
        if (ch == '[' || ch == ']') { *tok = std::string(1, ch); ++pos; return true; }
        size_t start = pos;
        while (pos < text.size()) {
            char c = text[pos];
            if (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '"' || c == '[' || c == ']') break;
            ++pos;
        }
        *tok = text.substr(start, pos - start);
        return true;
    }
};

double ParseNumber(const std::string &s, bool *ok) {  // parser.cpp:322-368
    *ok = true;
    if (s.size() == 1) {
        if (!(s[0] >= '0' && s[0] <= '9')) { *ok = false; return 0; }
        return s[0] - '0';
    }
    bool isInt = true;
    for (char c : s) if (!(c >= '0' && c <= '9')) isInt = false;
    char *end = nullptr;
    double val;
    if (isInt) val = double(strtol(s.c_str(), &end, 10));
    else val = strtof(s.c_str(), &end);  // Float == float
    if (val == 0 && end == s.c_str()) *ok = false;
    return val;
}

// ------------------------------------------------------------------ state
struct MaterialInstance {
    std::string name;  // material type
    int material = -1; // index into HostScene::materials, -1 = none
    ParamSet params;
};

struct GraphicsState {
    TextureMaps textures;
    std::map<std::string, std::shared_ptr<MaterialInstance>> namedMaterials;
    std::shared_ptr<MaterialInstance> currentMaterial;
    ParamSet areaLightParams;
    std::string areaLight;
    bool reverseOrientation = false;
};

struct PendingPrim {
    int shape;  // >=0 tri, <0 ~sphere
    int material;
    int light;
    Bounds3 bounds;
    int instance = 0;   // k + 1: the TransformedPrimitive of instance k
};

struct Api {
    HostScene *scene;
    LoadOverrides ov;
    std::string baseDir;
    enum { Uninit, Options, World } state = Options;
    Transform ctm;
    std::vector<Transform> transformStack;
    std::vector<GraphicsState> gsStack;
    std::vector<char> pushKinds;
    GraphicsState gs;
    std::map<std::string, Transform> namedCoordSys;
    // render options (api.cpp:168-196 defaults)
    std::string filterName = "box", filmName = "image", samplerName = "halton", accelName = "bvh",
                integratorName = "path", cameraName = "perspective";
    ParamSet filterParams, filmParams, samplerParams, accelParams, integratorParams, cameraParams;
    Transform cameraToWorld;
    std::vector<PendingPrim> pending;
    // Object instancing (api.cpp:1544-1615). As in the reference, an object's primitives are created once, in the space they
    // were declared in, and every ObjectInstance is a TransformedPrimitive in the world's BVH: the world-space box of the
    // object and InstanceToWorld; the object's primitives get a BVH of their own at WorldEnd. (MIPT_INSTANCES=expand
    // re-creates the recorded shapes under InstanceToWorld * (their CTM) instead -- round 1's world-space copies: the same
    // surfaces, hit points that differ from the reference's in rounding.)
    struct RecordedShape { std::string name; ParamSet params; Transform ctm; GraphicsState gs; };
    std::map<std::string, std::vector<RecordedShape>> instances;
    std::vector<RecordedShape> *currentInstance = nullptr;
    struct ObjectDef { std::vector<PendingPrim> prims; Bounds3 bounds; bool created = false; int root = -1; };
    std::map<std::string, ObjectDef> objectDefs;
    std::vector<std::string> objectOrder;                 // objects in the order they were first instanced
    struct InstanceRec { std::string object; Transform i2w; };
    std::vector<InstanceRec> instanceRecs;
    bool expandInstances = false;
    std::map<std::string, std::shared_ptr<PLYMeshData>> plyCache;   // an instanced plymesh is read once
    std::map<std::string, Spectrum> cachedSpectra;                  // paramset.cpp:48, SPD files by name
    bool worldEnded = false;
    bool fatal = false;
    std::string fatalMsg;

    void Warn(const std::string &m) { scene->warnings.push_back(m); }
    void Err(const std::string &m) { scene->errors.push_back(m); }

    Api(HostScene *s, const LoadOverrides &o) : scene(s), ov(o) {
        if (const char *e = getenv("MIPT_INSTANCES")) expandInstances = std::string(e) == "expand";
        // default material: matte with default params (GraphicsState ctor, api.cpp:214-222)
        ParamSet empty;
        gs.currentMaterial = std::make_shared<MaterialInstance>();
        gs.currentMaterial->name = "matte";
        gs.currentMaterial->material = MakeMaterial("matte", empty, empty);
    }

    int MakeMaterial(const std::string &name, const ParamSet &geom, const ParamSet &mat) {
        if (name == "" || name == "none") return -1;
        TextureParams mp(geom, mat, gs.textures, &scene->errors);
        mi_material m;
        std::vector<std::string> errs;
        std::string type = name;
        static const char *known[] = {"matte", "plastic", "glass", "uber", "disney", "mirror", "translucent",
                                      "metal", "substrate", "mix", "hair", "fourier", "subsurface",
                                      "kdsubsurface", "retroreflective"};
        bool isKnown = false;
        for (const char *k : known) if (type == k) isKnown = true;
        if (!isKnown) {  // api.cpp:604-607
            Warn("Material \"" + name + "\" unknown. Using \"matte\".");
            type = "matte";
        }
        bool compiled;
        if (type == "mix") {  // api.cpp:573-593 + CreateMixMaterial (mixmat.cpp:66-72)
            auto lookup = [&](const char *param) -> mi_material {
                const std::string nm = mp.FindString(param, "");
                auto it = gs.namedMaterials.find(nm);
                if (it == gs.namedMaterials.end() || it->second->material < 0) {
                    Err("Named material \"" + nm + "\" undefined.  Using \"matte\"");
                    mi_material mm;
                    std::vector<std::string> e2;
                    CompileMaterial("matte", mp, &mm, &scene->warnings, &e2);
                    return mm;
                }
                return scene->materials[it->second->material];
            };
            const mi_material m1 = lookup("namedmaterial1"), m2 = lookup("namedmaterial2");
            compiled = CompileMixMaterial(m1, m2, mp.GetSpectrum("amount", Spectrum(0.5f)), &m, &errs);
        } else
            compiled = CompileMaterial(type, mp, &m, &scene->warnings, &errs);
        if (compiled && (m.bump_tex >= 0 || m.rough_tex[0] >= 0 || m.rough_tex[1] >= 0)) m.textured = 1;   // bump-mapped / roughness-mapped vertices take the texture-evaluating shading instances
        if (!compiled) {
            for (auto &e : errs) Err(e);
            Err("Material \"" + name + "\" replaced by default matte on this path.");
            ParamSet e1, e2;
            TextureParams mp2(e1, e2, gs.textures, &scene->errors);
            CompileMaterial("matte", mp2, &m, &scene->warnings, &errs);
        }
        // de-duplicate identical records (a loopsubdiv shape re-creates its material, api.cpp:1502-1513)
        for (size_t i = 0; i < scene->materials.size(); ++i)
            if (std::memcmp(&scene->materials[i], &m, sizeof(m)) == 0) return (int)i;
        scene->materials.push_back(m);
        return (int)scene->materials.size() - 1;
    }

    // ---- shapes
    static bool ShapeMaySetMaterialParameters(const ParamSet &ps) {  // api.cpp:1451-1500
        for (const auto &p : ps.textures)
            if (p.name != "alpha" && p.name != "shadowalpha") return true;
        for (const auto &p : ps.floats)
            if (p.values.size() == 1 && p.name != "radius") return true;
        for (const auto &p : ps.strings)
            if (p.values.size() == 1 && p.name != "filename" && p.name != "type" && p.name != "scheme") return true;
        for (const auto &p : ps.bools) if (p.values.size() == 1) return true;
        for (const auto &p : ps.ints) if (p.values.size() == 1) return true;
        for (const auto &p : ps.point2s) if (p.values.size() == 1) return true;
        for (const auto &p : ps.point3s) if (p.values.size() == 1) return true;
        for (const auto &p : ps.vector3s) if (p.values.size() == 1) return true;
        for (const auto &p : ps.normals) if (p.values.size() == 1) return true;
        for (const auto &p : ps.spectra) if (p.values.size() == 1) return true;
        return false;
    }

    // Append a TriangleMesh (src/shapes/triangle.cpp:54-92): world-space P, N.
    // Returns first triangle index.
    // "alpha" / "shadowalpha" of a mesh (triangle.cpp:716-740): a named float texture or the constant 0
    int AlphaTexture(const ParamSet &params, const char *param) {
        const std::string texName = params.FindTexture(param);
        if (texName != "") {
            auto it = gs.textures.floatImageTex.find(texName);
            if (it != gs.textures.floatImageTex.end()) return it->second;
            auto c = gs.textures.floatTex.find(texName);
            if (c == gs.textures.floatTex.end()) { Err("Couldn't find float texture \"" + texName + "\" for \"" + param + "\" parameter"); return -1; }
            if (c->second != 0.f) return -1;   // a constant that is never 0 masks nothing
        } else if (params.FindOneFloat(param, 1.f) != 0.f)
            return -1;
        mi_texture t{};
        t.mipmap = ConstantFloatMipMap(scene, 0.f);
        t.filter = MI_TEX_TRILINEAR; t.max_aniso = 8.f; t.su = t.sv = 1.f; t.post_scale = 1.f;
        scene->textures.push_back(t);
        return (int)scene->textures.size() - 1;
    }

    int AddTriangleMesh(const std::vector<int> &indices, const std::vector<Vec3> &P,
                        const std::vector<Vec3> *N, const std::vector<Vec2> *UV, int alphaTex = -1, int shadowAlphaTex = -1) {
        mi_mesh mesh{};
        mesh.alpha_tex = alphaTex; mesh.shadow_alpha_tex = shadowAlphaTex;
        mesh.first_vertex = (uint32_t)(scene->P.size() / 3);
        mesh.n_vertices = (uint32_t)P.size();
        mesh.first_tri = (uint32_t)(scene->triIndices.size() / 3);
        mesh.n_tris = (uint32_t)(indices.size() / 3);
        mesh.flags = 0;
        if (N) mesh.flags |= MI_MESH_HAS_N;
        if (UV) mesh.flags |= MI_MESH_HAS_UV;
        if (gs.reverseOrientation ^ ctm.SwapsHandedness()) mesh.flags |= MI_MESH_FLIP;
        uint32_t meshId = (uint32_t)scene->meshes.size();
        for (size_t i = 0; i < P.size(); ++i) {
            Vec3 pw = ctm.Point(P[i]);
            scene->P.push_back(pw.x); scene->P.push_back(pw.y); scene->P.push_back(pw.z);
            Vec3 nw(0, 0, 0);
            if (N) nw = ctm.Normal((*N)[i]);
            scene->N.push_back(nw.x); scene->N.push_back(nw.y); scene->N.push_back(nw.z);
            Vec2 uv;
            if (UV) uv = (*UV)[i];
            scene->UV.push_back(uv.x); scene->UV.push_back(uv.y);
        }
        for (size_t i = 0; i < indices.size(); ++i) scene->triIndices.push_back(indices[i] + (int)mesh.first_vertex);
        for (uint32_t i = 0; i < mesh.n_tris; ++i) scene->triMesh.push_back(meshId);
        scene->meshes.push_back(mesh);
        scene->stats.nMeshes++;
        scene->stats.nTriangles += (int)mesh.n_tris;
        return (int)mesh.first_tri;
    }

    Vec3 VertexP(int vi) const { return Vec3(scene->P[3 * vi], scene->P[3 * vi + 1], scene->P[3 * vi + 2]); }

    int MakeAreaLight(int shape, float area) {  // CreateDiffuseAreaLight, src/lights/diffuse.cpp:136-147
        if (gs.areaLight != "area" && gs.areaLight != "diffuse") {
            Warn("Area light \"" + gs.areaLight + "\" unknown.");
            return -1;
        }
        const ParamSet &ps = gs.areaLightParams;
        Spectrum L = ps.FindOneSpectrum("L", Spectrum(1.0));
        Spectrum sc = ps.FindOneSpectrum("scale", Spectrum(1.0));
        (void)ps.FindOneInt("samples", ps.FindOneInt("nsamples", 1));
        bool twoSided = ps.FindOneBool("twosided", false);
        mi_light l{};
        l.type = MI_LIGHT_DIFFUSE_AREA;
        l.shape = shape;
        l.two_sided = twoSided ? 1 : 0;
        l.area = area;
        Spectrum Lemit = L * sc;
        for (int i = 0; i < MI_NSPEC; ++i) l.L[i] = Lemit.c[i];
        scene->lights.push_back(l);
        return (int)scene->lights.size() - 1;
    }

    void AddPrims(int firstTri, int nTris, int material) {
        for (int t = 0; t < nTris; ++t) {
            int tri = firstTri + t;
            const int32_t *v = &scene->triIndices[3 * tri];
            Vec3 p0 = VertexP(v[0]), p1 = VertexP(v[1]), p2 = VertexP(v[2]);
            PendingPrim pp;
            pp.shape = tri;
            pp.material = material;
            pp.light = -1;
            if (gs.areaLight != "") {
                float area = 0.5 * Cross(p1 - p0, p2 - p0).Length();  // Triangle::Area, triangle.cpp:575-581
                pp.light = MakeAreaLight(tri, area);
            }
            pp.bounds = Union(Bounds3(p0, p1), p2);  // Triangle::WorldBound
            pending.push_back(pp);
        }
    }

    void Shape(const std::string &name, const ParamSet &params) {
        if (state != World) { Err("Scene description must be inside world block; \"Shape\" not allowed. Ignoring."); return; }
        if (currentInstance) {   // api.cpp:1431-1435
            if (gs.areaLight != "") Warn("Area lights not supported with object instancing");
            RecordedShape r{name, params, ctm, gs};
            r.gs.areaLight = "";
            currentInstance->push_back(std::move(r));
            return;
        }
        int firstTri = -1, nTris = 0;
        int sphereIdx = -1;
        if (name == "trianglemesh") {  // CreateTriangleMeshShape, triangle.cpp:642-740
            const std::vector<int> *vi = ParamSet::Find(params.ints, "indices");
            const std::vector<Vec3> *P = ParamSet::Find(params.point3s, "P");
            const std::vector<Vec2> *uvs = ParamSet::Find(params.point2s, "uv");
            if (!uvs) uvs = ParamSet::Find(params.point2s, "st");
            std::vector<Vec2> tempUVs;
            if (!uvs) {
                const std::vector<float> *fuv = ParamSet::Find(params.floats, "uv");
                if (!fuv) fuv = ParamSet::Find(params.floats, "st");
                if (fuv) {
                    for (size_t i = 0; i + 1 < fuv->size(); i += 2) { Vec2 q; q.x = (*fuv)[i]; q.y = (*fuv)[i + 1]; tempUVs.push_back(q); }
                    uvs = &tempUVs;
                }
            }
            if (!vi) { Err("Vertex indices \"indices\" not provided with triangle mesh shape"); return; }
            if (!P) { Err("Vertex positions \"P\" not provided with triangle mesh shape"); return; }
            if (uvs) {
                if (uvs->size() < P->size()) {
                    Err("Not enough of \"uv\"s for triangle mesh. Discarding.");
                    uvs = nullptr;
                } else if (uvs->size() > P->size())
                    Warn("More \"uv\"s provided than will be used for triangle mesh.");
            }
            if (ParamSet::Find(params.vector3s, "S")) Warn("trianglemesh \"S\" tangents ignored on this path");
            const std::vector<Vec3> *N = ParamSet::Find(params.normals, "N");
            if (N && N->size() != P->size()) { Err("Number of \"N\"s for triangle mesh must match \"P\"s"); N = nullptr; }
            for (int idx : *vi)
                if (idx >= (int)P->size() || idx < 0) { Err("trianglemesh has out of-bounds vertex index"); return; }
            std::vector<int> idx(vi->begin(), vi->begin() + (vi->size() / 3) * 3);
            nTris = (int)idx.size() / 3;
            const int alphaTex = AlphaTexture(params, "alpha"), shadowAlphaTex = AlphaTexture(params, "shadowalpha");
            firstTri = AddTriangleMesh(idx, *P, N, uvs, alphaTex, shadowAlphaTex);
        } else if (name == "plymesh") {  // CreatePLYMesh, plymesh.cpp:149-283
            std::string fn = params.FindOneString("filename", "");
            if (!fn.empty() && fn[0] != '/') fn = baseDir + "/" + fn;  // FindOneFilename -> AbsolutePath(ResolveFilename())
            std::shared_ptr<PLYMeshData> &cached = plyCache[fn];
            if (!cached) {
                cached = std::make_shared<PLYMeshData>();
                std::vector<std::string> w;
                std::string e;
                const bool okPly = ReadPLYMesh(fn, cached.get(), &w, &e);
                for (const std::string &m : w) Warn(m);
                if (!okPly) { Err(e); cached.reset(); plyCache.erase(fn); return; }
            }
            const PLYMeshData &ply = *cached;
            nTris = (int)ply.indices.size() / 3;
            const int alphaTex = AlphaTexture(params, "alpha"), shadowAlphaTex = AlphaTexture(params, "shadowalpha");
            firstTri = AddTriangleMesh(ply.indices, ply.P, ply.N.empty() ? nullptr : &ply.N, ply.UV.empty() ? nullptr : &ply.UV, alphaTex, shadowAlphaTex);
        } else if (name == "loopsubdiv") {  // CreateLoopSubdiv, loopsubdiv.cpp:402-424
            int nLevels = params.FindOneInt("levels", params.FindOneInt("nlevels", 3));
            const std::vector<int> *vi = ParamSet::Find(params.ints, "indices");
            const std::vector<Vec3> *P = ParamSet::Find(params.point3s, "P");
            if (!vi) { Err("Vertex indices \"indices\" not provided for LoopSubdiv shape."); return; }
            if (!P) { Err("Vertex positions \"P\" not provided for LoopSubdiv shape."); return; }
            (void)params.FindOneString("scheme", "loop");
            std::vector<int> oi;
            std::vector<Vec3> oP, oN;
            std::string e;
            if (!LoopSubdivide(nLevels, *vi, *P, &oi, &oP, &oN, &e)) { Err(e); return; }
            nTris = (int)oi.size() / 3;
            firstTri = AddTriangleMesh(oi, oP, &oN, nullptr);
        } else if (name == "sphere") {  // CreateSphereShape, sphere.cpp:318-328; Sphere ctor sphere.h:52-63
            float radius = params.FindOneFloat("radius", 1.f);
            float zmin = params.FindOneFloat("zmin", -radius);
            float zmax = params.FindOneFloat("zmax", radius);
            float phimax = params.FindOneFloat("phimax", 360.f);
            mi_sphere s{};
            Transform w2o = Inverse(ctm);
            std::memcpy(s.o2w, ctm.m.m, sizeof(s.o2w));
            std::memcpy(s.w2o, w2o.m.m, sizeof(s.w2o));
            s.radius = radius;
            s.z_min = Clamp(std::min(zmin, zmax), -radius, radius);
            s.z_max = Clamp(std::max(zmin, zmax), -radius, radius);
            s.theta_min = std::acos(Clamp(std::min(zmin, zmax) / radius, -1, 1));
            s.theta_max = std::acos(Clamp(std::max(zmin, zmax) / radius, -1, 1));
            s.phi_max = Radians(Clamp(phimax, 0, 360));
            s.reverse_orientation = gs.reverseOrientation ? 1 : 0;
            s.swaps_handedness = ctm.SwapsHandedness() ? 1 : 0;
            sphereIdx = (int)scene->spheres.size();
            scene->spheres.push_back(s);
            scene->stats.nSpheres++;
        } else {
            Err("Shape \"" + name + "\" is outside the PathIntegrator hot-path scope (SURVEY 2 rows 13-14); skipped.");
            return;
        }
        // material (api.cpp:1378, GetMaterialForShape 1502-1513)
        int material;
        if (ShapeMaySetMaterialParameters(params))
            material = MakeMaterial(gs.currentMaterial->name, params, gs.currentMaterial->params);
        else
            material = gs.currentMaterial->material;
        if (sphereIdx >= 0) {
            const mi_sphere &s = scene->spheres[sphereIdx];
            PendingPrim pp;
            pp.shape = ~sphereIdx;
            pp.material = material;
            pp.light = -1;
            if (gs.areaLight != "") {
                float area = s.phi_max * s.radius * (s.z_max - s.z_min);  // Sphere::Area, sphere.cpp:217
                pp.light = MakeAreaLight(~sphereIdx, area);
                if (ctm.HasScale()) Warn("Scaling detected in world to light transformation!");
            }
            Bounds3 ob(Vec3(-s.radius, -s.radius, s.z_min), Vec3(s.radius, s.radius, s.z_max));
            pp.bounds = ctm.Bounds(ob);  // Shape::WorldBound, shape.cpp:54
            pending.push_back(pp);
        } else
            AddPrims(firstTri, nTris, material);
        std::vector<std::string> unused;
        params.ReportUnused(&unused);
        for (auto &u : unused) Warn("Parameter \"" + u + "\" not used");
    }

    void ObjectBegin(const std::string &name) {  // api.cpp:1544-1553
        gsStack.push_back(gs); transformStack.push_back(ctm); pushKinds.push_back('a');
        if (currentInstance) Err("ObjectBegin called inside of instance definition");
        instances[name] = std::vector<RecordedShape>();
        currentInstance = &instances[name];
    }
    void ObjectEnd() {  // api.cpp:1557-1566
        if (!currentInstance) Err("ObjectEnd called outside of instance definition");
        currentInstance = nullptr;
        if (gsStack.empty()) { Err("Unmatched AttributeEnd encountered. Ignoring it."); return; }
        gs = gsStack.back(); gsStack.pop_back();
        ctm = transformStack.back(); transformStack.pop_back();
        pushKinds.pop_back();
    }
    void ObjectInstance(const std::string &name) {  // api.cpp:1570-1615
        if (currentInstance) { Err("ObjectInstance can't be called inside instance definition"); return; }
        auto it = instances.find(name);
        if (it == instances.end()) { Err("Unable to find instance named \"" + name + "\""); return; }
        const Transform instanceToWorld = ctm;
        const GraphicsState saved = gs;
        if (expandInstances) {
            for (const RecordedShape &r : it->second) {
                ctm = instanceToWorld * r.ctm;
                gs = r.gs;
                Shape(r.name, r.params);
            }
            ctm = instanceToWorld;
            gs = saved;
            return;
        }
        ObjectDef &od = objectDefs[name];
        if (!od.created) {   // the object's shapes, once, where they were declared
            od.created = true;
            objectOrder.push_back(name);
            std::vector<PendingPrim> world;
            world.swap(pending);
            for (const RecordedShape &r : it->second) {
                ctm = r.ctm;
                gs = r.gs;
                Shape(r.name, r.params);
            }
            od.prims.swap(pending);
            pending.swap(world);
            ctm = instanceToWorld;
            gs = saved;
            for (const PendingPrim &pp : od.prims) od.bounds = Union(od.bounds, pp.bounds);
        }
        if (od.prims.empty()) return;   // api.cpp:1580
        instanceRecs.push_back(InstanceRec{name, instanceToWorld});
        PendingPrim pp;
        pp.shape = 0;
        pp.material = -1;
        pp.light = -1;
        pp.instance = (int)instanceRecs.size();
        pp.bounds = instanceToWorld.Bounds(od.bounds);   // TransformedPrimitive::WorldBound: MotionBounds of a static transform
        pending.push_back(pp);
    }

    void LightSource(const std::string &name, const ParamSet &ps) {  // MakeLight, api.cpp:747-771
        if (state != World) { Err("\"LightSource\" not allowed outside world block. Ignoring."); return; }
        mi_light l{};
        if (name == "point") {  // CreatePointLight, point.cpp:80-88
            Spectrum I = ps.FindOneSpectrum("I", Spectrum(1.0));
            Spectrum sc = ps.FindOneSpectrum("scale", Spectrum(1.0));
            Vec3 from = ps.FindOnePoint3("from", Vec3(0, 0, 0));
            Transform l2w = Translate(from) * ctm;
            Vec3 p = l2w.Point(Vec3(0, 0, 0));
            l.type = MI_LIGHT_POINT;
            l.shape = 0;
            Spectrum Is = I * sc;
            for (int i = 0; i < MI_NSPEC; ++i) l.L[i] = Is.c[i];
            l.pos[0] = p.x; l.pos[1] = p.y; l.pos[2] = p.z;
        } else if (name == "distant") {  // CreateDistantLight, distant.cpp:94-102
            Spectrum L = ps.FindOneSpectrum("L", Spectrum(1.0));
            Spectrum sc = ps.FindOneSpectrum("scale", Spectrum(1.0));
            Vec3 from = ps.FindOnePoint3("from", Vec3(0, 0, 0));
            Vec3 to = ps.FindOnePoint3("to", Vec3(0, 0, 1));
            Vec3 dir = from - to;
            Vec3 w = Normalize(ctm.Vector(dir));
            l.type = MI_LIGHT_DISTANT;
            Spectrum Ls = L * sc;
            for (int i = 0; i < MI_NSPEC; ++i) l.L[i] = Ls.c[i];
            l.dir[0] = w.x; l.dir[1] = w.y; l.dir[2] = w.z;
        } else if (name == "spot") {  // CreateSpotLight, spot.cpp:102-122
            Spectrum I = ps.FindOneSpectrum("I", Spectrum(1.0));
            Spectrum sc = ps.FindOneSpectrum("scale", Spectrum(1.0));
            float coneangle = ps.FindOneFloat("coneangle", 30.);
            float conedelta = ps.FindOneFloat("conedeltaangle", 5.);
            Vec3 from = ps.FindOnePoint3("from", Vec3(0, 0, 0));
            Vec3 to = ps.FindOnePoint3("to", Vec3(0, 0, 1));
            Vec3 dir = Normalize(to - from);
            Vec3 du, dv;
            CoordinateSystem(dir, &du, &dv);
            Transform dirToZ(Matrix4x4(du.x, du.y, du.z, 0., dv.x, dv.y, dv.z, 0., dir.x, dir.y, dir.z, 0., 0, 0, 0, 1.));
            Transform light2world = ctm * Translate(Vec3(from.x, from.y, from.z)) * Inverse(dirToZ);
            Vec3 p = light2world.Point(Vec3(0, 0, 0));
            l.type = MI_LIGHT_SPOT;
            Spectrum Is = I * sc;
            for (int i = 0; i < MI_NSPEC; ++i) l.L[i] = Is.c[i];
            l.pos[0] = p.x; l.pos[1] = p.y; l.pos[2] = p.z;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) { l.l2w[3 * r + c] = light2world.m.m[r][c]; l.w2l[3 * r + c] = light2world.mInv.m[r][c]; }
            l.cos_total_width = std::cos(Radians(coneangle));
            l.cos_falloff_start = std::cos(Radians(coneangle - conedelta));
        } else if (name == "infinite" || name == "exinfinite") {  // CreateInfiniteLight, infinite.cpp:176-186
            Spectrum L = ps.FindOneSpectrum("L", Spectrum(1.0));
            Spectrum sc = ps.FindOneSpectrum("scale", Spectrum(1.0));
            std::string texmap = ps.FindOneString("mapname", "");
            if (!texmap.empty() && texmap[0] != '/') texmap = baseDir + "/" + texmap;   // FindOneFilename
            (void)ps.FindOneInt("samples", ps.FindOneInt("nsamples", 1));
            HostEnvMap env;
            Spectrum centre;
            std::vector<std::string> errs;
            BuildEnvMap(L * sc, texmap, &env, &centre, &errs);
            for (const std::string &e : errs) Err(e);
            l.type = MI_LIGHT_INFINITE;
            l.envmap = (int)scene->envStore.size();
            scene->envStore.push_back(std::move(env));
            for (int i = 0; i < MI_NSPEC; ++i) l.L[i] = centre.c[i];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) { l.l2w[3 * r + c] = ctm.m.m[r][c]; l.w2l[3 * r + c] = ctm.mInv.m[r][c]; }
        } else {
            Err("LightSource \"" + name + "\" is outside the hot-path scope (SURVEY 2 row 20); skipped.");
            return;
        }
        scene->lights.push_back(l);
        std::vector<std::string> unused;
        ps.ReportUnused(&unused);
        for (auto &u : unused) Warn("Parameter \"" + u + "\" not used");
    }

    void Texture(const std::string &name, const std::string &type, const std::string &texname, const ParamSet &ps) {
        // Every texture on this path evaluates to a constant, so "scale" and "mix" of such textures fold into
        // constants with the reference's arithmetic (src/textures/scale.h:56-58, mix.h:57-61, scale.cpp, mix.cpp).
        if (texname != "constant" && texname != "scale" && texname != "mix" && texname != "imagemap" && texname != "checkerboard") {
            Err("Texture class \"" + texname + "\" is outside the hot-path scope (\"constant\", \"scale\", \"mix\", \"imagemap\", \"checkerboard\")");
            return;
        }
        ParamSet empty;
        TextureParams tp(ps, empty, gs.textures, &scene->errors);
        const bool isFloat = type == "float", isSpec = type == "color" || type == "spectrum";
        if (!isFloat && !isSpec) { Err("Texture type \"" + type + "\" unknown."); return; }
        if (texname == "checkerboard") {  // CreateCheckerboardSpectrumTexture, checkerboard.cpp:99-150
            if (isFloat) { Err("Texture \"" + name + "\": float \"checkerboard\" textures are outside the hot-path scope"); return; }
            if (ps.FindOneInt("dimension", 2) != 2) { Err("Texture \"" + name + "\": 3D \"checkerboard\" textures are outside the hot-path scope"); return; }
            const std::string mapping = ps.FindOneString("mapping", "uv");
            if (mapping != "uv") { Err("Texture \"" + name + "\": 2D texture mapping \"" + mapping + "\" is outside the hot-path scope (\"uv\" only)"); return; }
            const SpectrumParam t1 = tp.GetSpectrumParam("tex1", Spectrum(1.f)), t2 = tp.GetSpectrumParam("tex2", Spectrum(0.f));
            if (t1.tex >= 0 || t2.tex >= 0) { Err("Texture \"" + name + "\": a \"checkerboard\" of image textures is outside the hot-path scope (constant tex1 / tex2)"); return; }
            mi_texture t{};
            t.type = MI_TEX_CHECKERBOARD;
            t.mipmap = -1;
            t.su = ps.FindOneFloat("uscale", 1.f); t.sv = ps.FindOneFloat("vscale", 1.f);
            t.du = ps.FindOneFloat("udelta", 0.f); t.dv = ps.FindOneFloat("vdelta", 0.f);
            t.post_scale = 1.f; t.max_aniso = 8.f;
            const std::string aa = ps.FindOneString("aamode", "closedform");
            if (aa == "none") t.aa_none = 1;
            else if (aa != "closedform") Warn("Antialiasing mode \"" + aa + "\" not understood by Checkerboard2DTexture; using \"closedform\"");
            for (int i = 0; i < MI_NSPEC; ++i) { t.spec1[i] = t1.s.c[i]; t.spec2[i] = t2.s.c[i]; }
            gs.textures.spectrumTex.erase(name);
            gs.textures.imageTex[name] = (int)scene->textures.size();
            scene->textures.push_back(t);
            return;
        }
        if (texname == "imagemap") {  // CreateImageSpectrumTexture, imagemap.cpp:152-197
            const std::string mapping = ps.FindOneString("mapping", "uv");
            if (mapping != "uv") { Err("Texture \"" + name + "\": 2D texture mapping \"" + mapping + "\" is outside the hot-path scope (\"uv\" only)"); return; }
            mi_texture t{};
            t.su = ps.FindOneFloat("uscale", 1.f); t.sv = ps.FindOneFloat("vscale", 1.f);
            t.du = ps.FindOneFloat("udelta", 0.f); t.dv = ps.FindOneFloat("vdelta", 0.f);
            t.post_scale = 1.f;
            t.max_aniso = ps.FindOneFloat("maxanisotropy", 8.f);
            if (!(t.max_aniso <= 64.f)) {   // an EWA footprint is up to 2 x maxanisotropy texels long: keep the per-lane loop bounded
                Warn("Texture \"" + name + "\": \"maxanisotropy\" " + std::to_string(t.max_aniso) + " clamped to 64 on this path");
                t.max_aniso = 64.f;
            }
            const bool trilerp = ps.FindOneBool("trilinear", false), noFilt = ps.FindOneBool("noFiltering", false);
            const std::string wrap = ps.FindOneString("wrap", "repeat");
            const int wrapMode = wrap == "black" ? 1 : (wrap == "clamp" ? 2 : 0);
            const float scale = ps.FindOneFloat("scale", 1.f);
            std::string filename = ps.FindOneString("filename", "");
            std::string ext = filename.size() >= 4 ? filename.substr(filename.size() - 4) : "";
            for (char &c : ext) c = (char)tolower(c);
            const bool gamma = ps.FindOneBool("gamma", ext == ".tga" || ext == ".png");
            if (ps.FindOneBool("useSPD", false)) Warn("Texture \"" + name + "\": \"useSPD\" (display primaries) is not implemented on this path; reflectance conversion used");
            if (!filename.empty() && filename[0] != '/') filename = baseDir + "/" + filename;   // FindOneFilename
            // (MIPMap::Lookup: noFiltering takes the trilinear entry point, mipmap.h:283-288)
            t.filter = noFilt ? MI_TEX_NONE : (trilerp ? MI_TEX_TRILINEAR : MI_TEX_EWA);
            t.mipmap = BuildTextureMipMap(scene, filename, trilerp, noFilt, t.max_aniso, wrapMode, scale, gamma, isFloat);
            if (isFloat) {   // (on this path a float image texture can be an "alpha" / "shadowalpha" mask of a mesh)
                gs.textures.floatTex.erase(name);
                gs.textures.floatImageTex[name] = (int)scene->textures.size();
            } else {
                gs.textures.spectrumTex.erase(name);
                gs.textures.imageTex[name] = (int)scene->textures.size();
            }
            scene->textures.push_back(t);
            std::vector<std::string> unused;
            ps.ReportUnused(&unused);
            for (auto &u : unused) Warn("Parameter \"" + u + "\" not used");
            return;
        }
        if (texname == "constant") {
            if (isFloat) gs.textures.floatTex[name] = tp.GetFloat("value", 1.f);
            else gs.textures.spectrumTex[name] = tp.GetSpectrum("value", Spectrum(1.f));
        } else if (texname == "scale") {
            // ScaleTexture of a float image texture and a constant (the usual way to size a bump map): the image texture
            // with the constant as post-lookup factor -- tex1(si) * tex2(si), scale.h:56-58
            const int img1 = isFloat ? tp.GetFloatImageTexture("tex1") : -1, img2 = isFloat ? tp.GetFloatImageTexture("tex2") : -1;
            if (img1 >= 0 || img2 >= 0) {
                if (img1 >= 0 && img2 >= 0) { Err("Texture \"" + name + "\": a \"scale\" of two image textures is outside the hot-path scope"); return; }
                mi_texture t = scene->textures[img1 >= 0 ? img1 : img2];
                t.post_scale = t.post_scale * tp.GetFloat(img1 >= 0 ? "tex2" : "tex1", 1.f);
                gs.textures.floatTex.erase(name);
                gs.textures.floatImageTex[name] = (int)scene->textures.size();
                scene->textures.push_back(t);
                return;
            }
            if (!isFloat) {   // spectrum: image texture * constant spectrum -> the lobe keeps the constant and multiplies the texture value in
                const SpectrumParam p1 = tp.GetSpectrumParam("tex1", Spectrum(1.f)), p2 = tp.GetSpectrumParam("tex2", Spectrum(1.f));
                if (p1.tex >= 0 || p2.tex >= 0) {
                    if (p1.tex >= 0 && p2.tex >= 0) { Err("Texture \"" + name + "\": a \"scale\" of two image textures is outside the hot-path scope"); return; }
                    const SpectrumParam &img = p1.tex >= 0 ? p1 : p2, &cst = p1.tex >= 0 ? p2 : p1;
                    gs.textures.spectrumTex.erase(name);
                    gs.textures.scaledImageTex[name] = std::make_pair(img.tex, img.scaled ? img.s * cst.s : cst.s);
                    return;
                }
            }
            if (isFloat) gs.textures.floatTex[name] = tp.GetFloat("tex1", 1.f) * tp.GetFloat("tex2", 1.f);
            else gs.textures.spectrumTex[name] = tp.GetSpectrum("tex1", Spectrum(1.f)) * tp.GetSpectrum("tex2", Spectrum(1.f));
        } else {
            const float amt = tp.GetFloat("amount", 0.5f);
            if (isFloat) gs.textures.floatTex[name] = (1 - amt) * tp.GetFloat("tex1", 0.f) + amt * tp.GetFloat("tex2", 1.f);
            else gs.textures.spectrumTex[name] = (1 - amt) * tp.GetSpectrum("tex1", Spectrum(0.f)) + amt * tp.GetSpectrum("tex2", Spectrum(1.f));
        }
    }

    void WorldEnd();
};

float EvalFilter(const std::string &name, const ParamSet &ps, float radius[2], std::vector<std::string> *warn,
                 std::function<float(float, float)> *fn) {
    // src/filters/{box,triangle,gaussian,mitchell,sinc}.cpp
    if (name == "box") {
        radius[0] = ps.FindOneFloat("xwidth", 0.5f); radius[1] = ps.FindOneFloat("ywidth", 0.5f);
        *fn = [](float, float) { return 1.f; };
    } else if (name == "triangle") {
        radius[0] = ps.FindOneFloat("xwidth", 2.f); radius[1] = ps.FindOneFloat("ywidth", 2.f);
        float rx = radius[0], ry = radius[1];
        *fn = [rx, ry](float x, float y) { return std::max((float)0, rx - std::abs(x)) * std::max((float)0, ry - std::abs(y)); };
    } else if (name == "gaussian") {
        radius[0] = ps.FindOneFloat("xwidth", 2.f); radius[1] = ps.FindOneFloat("ywidth", 2.f);
        float alpha = ps.FindOneFloat("alpha", 2.f);
        float expX = std::exp(-alpha * radius[0] * radius[0]), expY = std::exp(-alpha * radius[1] * radius[1]);
        *fn = [alpha, expX, expY](float x, float y) {
            auto g = [alpha](float d, float expv) { return std::max((float)0, float(std::exp(-alpha * d * d) - expv)); };
            return g(x, expX) * g(y, expY);
        };
    } else if (name == "mitchell") {
        radius[0] = ps.FindOneFloat("xwidth", 2.f); radius[1] = ps.FindOneFloat("ywidth", 2.f);
        float B = ps.FindOneFloat("B", 1.f / 3.f), C = ps.FindOneFloat("C", 1.f / 3.f);
        float irx = 1 / radius[0], iry = 1 / radius[1];
        *fn = [B, C, irx, iry](float x, float y) {
            auto m1 = [B, C](float x) {
                x = std::abs(2 * x);
                if (x > 1)
                    return ((-B - 6 * C) * x * x * x + (6 * B + 30 * C) * x * x + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) * (1.f / 6.f);
                else
                    return ((12 - 9 * B - 6 * C) * x * x * x + (-18 + 12 * B + 6 * C) * x * x + (6 - 2 * B)) * (1.f / 6.f);
            };
            return m1(x * irx) * m1(y * iry);
        };
    } else if (name == "sinc") {
        radius[0] = ps.FindOneFloat("xwidth", 4.); radius[1] = ps.FindOneFloat("ywidth", 4.);
        float tau = ps.FindOneFloat("tau", 3.f);
        float rx = radius[0], ry = radius[1];
        *fn = [tau, rx, ry](float x, float y) {
            auto sinc = [](float x) { x = std::abs(x); if (x < 1e-5) return 1.f; return std::sin(kPi * x) / (kPi * x); };
            auto ws = [&](float x, float r) { x = std::abs(x); if (x > r) return 0.f; float l = sinc(x / tau); return sinc(x) * l; };
            return ws(x, rx) * ws(y, ry);
        };
    } else {
        warn->push_back("Filter \"" + name + "\" unknown.");
        return EvalFilter("box", ps, radius, warn, fn);
    }
    return 0;
}

void Api::WorldEnd() {
    worldEnded = true;
    mi_scene_desc &d = scene->desc;
    // ---- film (CreateFilm, film.cpp:313-352; Film ctor film.cpp:50-83)
    {
        float radius[2];
        std::function<float(float, float)> fn;
        EvalFilter(filterName, filterParams, radius, &scene->warnings, &fn);
        if (filmName != "image") Err("Film \"" + filmName + "\" unknown.");
        scene->filmFilename = filmParams.FindOneString("filename", "");
        if (scene->filmFilename == "") scene->filmFilename = "pbrt.exr";
        int xres = filmParams.FindOneInt("xresolution", 1280);
        int yres = filmParams.FindOneInt("yresolution", 720);
        if (ov.xres > 0) xres = ov.xres;
        if (ov.yres > 0) yres = ov.yres;
        float crop[4] = {0, 1, 0, 1};
        const std::vector<float> *cr = ParamSet::Find(filmParams.floats, "cropwindow");
        if (cr && cr->size() == 4) {
            crop[0] = Clamp(std::min((*cr)[0], (*cr)[1]), 0.f, 1.f);
            crop[1] = Clamp(std::max((*cr)[0], (*cr)[1]), 0.f, 1.f);
            crop[2] = Clamp(std::min((*cr)[2], (*cr)[3]), 0.f, 1.f);
            crop[3] = Clamp(std::max((*cr)[2], (*cr)[3]), 0.f, 1.f);
        } else if (cr)
            Err("values supplied for \"cropwindow\". Expected 4.");
        if (ov.crop[0] >= 0) for (int i = 0; i < 4; ++i) crop[i] = ov.crop[i];
        mi_film &f = d.film;
        f.scale = filmParams.FindOneFloat("scale", 1.);
        (void)filmParams.FindOneFloat("diagonal", 35.);
        f.max_sample_luminance = filmParams.FindOneFloat("maxsampleluminance", kInfinity);
        scene->spectralFlag = filmParams.FindOneBool("spectralFlag", true);
        f.full_res[0] = xres; f.full_res[1] = yres;
        f.cropped_bounds[0] = (int)std::ceil(xres * crop[0]);
        f.cropped_bounds[1] = (int)std::ceil(yres * crop[2]);
        f.cropped_bounds[2] = (int)std::ceil(xres * crop[1]);
        f.cropped_bounds[3] = (int)std::ceil(yres * crop[3]);
        f.filter_radius[0] = radius[0]; f.filter_radius[1] = radius[1];
        int offset = 0;
        for (int y = 0; y < 16; ++y)
            for (int x = 0; x < 16; ++x, ++offset) {
                float px = (x + 0.5f) * radius[0] / 16;
                float py = (y + 0.5f) * radius[1] / 16;
                f.filter_table[offset] = fn(px, py);
            }
        // GetSampleBounds, film.cpp:85-92
        f.sample_bounds[0] = (int)std::floor(float(f.cropped_bounds[0]) + 0.5f - radius[0]);
        f.sample_bounds[1] = (int)std::floor(float(f.cropped_bounds[1]) + 0.5f - radius[1]);
        f.sample_bounds[2] = (int)std::ceil(float(f.cropped_bounds[2]) - 0.5f + radius[0]);
        f.sample_bounds[3] = (int)std::ceil(float(f.cropped_bounds[3]) - 0.5f + radius[1]);
    }
    // ---- camera (CreatePerspectiveCamera, perspective.cpp:235-285; ProjectiveCamera ctor camera.h:91-112)
    {
        if (cameraName != "perspective")
            Err("Camera \"" + cameraName + "\" is outside the hot-path scope (SURVEY 2 row 25); using perspective.");
        const ParamSet &ps = cameraParams;
        float shutteropen = ps.FindOneFloat("shutteropen", 0.f);
        float shutterclose = ps.FindOneFloat("shutterclose", 1.f);
        if (shutterclose < shutteropen) { Warn("Shutter close time < shutter open. Swapping them."); std::swap(shutterclose, shutteropen); }
        float lensradius = ps.FindOneFloat("lensradius", 0.f);
        float focaldistance = ps.FindOneFloat("focaldistance", 1e6);
        float frame = ps.FindOneFloat("frameaspectratio", float(d.film.full_res[0]) / float(d.film.full_res[1]));
        float screen[4];  // xmin xmax ymin ymax
        if (frame > 1.f) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.f; screen[3] = 1.f; }
        else { screen[0] = -1.f; screen[1] = 1.f; screen[2] = -1.f / frame; screen[3] = 1.f / frame; }
        const std::vector<float> *sw = ParamSet::Find(ps.floats, "screenwindow");
        if (sw) {
            if (sw->size() == 4) for (int i = 0; i < 4; ++i) screen[i] = (*sw)[i];
            else Err("\"screenwindow\" should have four values");
        }
        float fov = ps.FindOneFloat("fov", 90.);
        float halffov = ps.FindOneFloat("halffov", -1.f);
        if (halffov > 0.f) fov = 2.f * halffov;
        Transform cameraToScreen = Perspective(fov, 1e-2f, 1000.f);
        Transform screenToRaster = Scale((float)d.film.full_res[0], (float)d.film.full_res[1], 1) *
                                   Scale(1 / (screen[1] - screen[0]), 1 / (screen[2] - screen[3]), 1) *
                                   Translate(Vec3(-screen[0], -screen[3], 0));
        Transform rasterToScreen = Inverse(screenToRaster);
        Transform rasterToCamera = Inverse(cameraToScreen) * rasterToScreen;
        std::memcpy(d.camera.raster_to_camera, rasterToCamera.m.m, sizeof(float) * 16);
        std::memcpy(d.camera.camera_to_world, cameraToWorld.m.m, sizeof(float) * 16);
        d.camera.lens_radius = lensradius;
        d.camera.focal_distance = focaldistance;
        d.camera.shutter_open = shutteropen;
        d.camera.shutter_close = shutterclose;
        if (cameraToWorld.HasScale()) Warn("Scaling detected in world-to-camera transformation!");
    }
    // ---- sampler (CreateHaltonSampler + ctor, halton.cpp:65-96,133-140)
    {
        scene->samplerName = samplerName;
        mi_sampler &s = d.sampler;
        s.type = MI_SAMPLER_HALTON;
        if (samplerName == "sobol") s.type = MI_SAMPLER_SOBOL;          // CreateSobolSampler, sobol.cpp:66-71
        else if (samplerName == "random") s.type = MI_SAMPLER_RANDOM;   // CreateRandomSampler, random.cpp:62-65
        else if (samplerName == "02sequence" || samplerName == "lowdiscrepancy") s.type = MI_SAMPLER_ZEROTWO;   // api.cpp:859-860
        else if (samplerName == "stratified") s.type = MI_SAMPLER_STRATIFIED;
        else if (samplerName != "halton")
            Err("Sampler \"" + samplerName + "\" is not built on this path (SURVEY 2: \"maxmindist\" is outside the hot-path scope); using halton.");
        int nsamp = samplerParams.FindOneInt("pixelsamples", s.type == MI_SAMPLER_RANDOM ? 4 : 16);
        if (ov.spp > 0) nsamp = ov.spp;
        if (s.type == MI_SAMPLER_SOBOL) {   // GlobalSampler(RoundUpPow2(samplesPerPixel)), sobol.h:52-57
            int p2 = 1;
            while (p2 < nsamp) p2 *= 2;
            if (p2 != nsamp) Warn("Non power-of-two sample count rounded up to " + std::to_string(p2) + " for SobolSampler.");
            nsamp = p2;
        }
        s.pixel_dims = 0; s.x_samples = s.y_samples = 0; s.jitter = 1;
        if (s.type == MI_SAMPLER_ZEROTWO) {   // ZeroTwoSequenceSampler ctor + CreateZeroTwoSequenceSampler, zerotwosequence.cpp:43-51,77-82
            int p2 = 1;
            while (p2 < nsamp) p2 *= 2;
            if (p2 != nsamp) Warn("Pixel samples being rounded up to power of 2 (from " + std::to_string(nsamp) + " to " + std::to_string(p2) + ").");
            nsamp = p2;
            s.pixel_dims = samplerParams.FindOneInt("dimensions", 4);
        }
        if (s.type == MI_SAMPLER_STRATIFIED) {   // CreateStratifiedSampler, stratified.cpp:79-86
            s.jitter = samplerParams.FindOneBool("jitter", true) ? 1 : 0;
            s.x_samples = samplerParams.FindOneInt("xsamples", 4);
            s.y_samples = samplerParams.FindOneInt("ysamples", 4);
            if (ov.spp > 0) {   // a sample-count override keeps the pixel's grid square-ish: the largest divisor of spp below its root
                int xs = 1;
                for (int k = 1; (long long)k * k <= ov.spp; ++k) if (ov.spp % k == 0) xs = k;
                s.x_samples = ov.spp / xs; s.y_samples = xs;
            }
            if (s.x_samples < 1 || s.y_samples < 1) { Err("Sampler \"stratified\": xsamples and ysamples must be positive; using 1."); s.x_samples = std::max(1, s.x_samples); s.y_samples = std::max(1, s.y_samples); }
            nsamp = s.x_samples * s.y_samples;
            s.pixel_dims = samplerParams.FindOneInt("dimensions", 4);
        }
        if (s.pixel_dims < 0 || s.pixel_dims > 64) { Err("Sampler \"" + samplerName + "\": dimensions must be in [0, 64]; using 4."); s.pixel_dims = 4; }
        s.samples_per_pixel = nsamp;
        s.sample_at_pixel_center = samplerParams.FindOneBool("samplepixelcenter", false) ? 1 : 0;
        int res[2] = {d.film.sample_bounds[2] - d.film.sample_bounds[0], d.film.sample_bounds[3] - d.film.sample_bounds[1]};
        const int kMaxResolution = 128;
        for (int i = 0; i < 2; ++i) {
            int base = (i == 0) ? 2 : 3;
            int scale = 1, exp = 0;
            while (scale < std::min(res[i], kMaxResolution)) { scale *= base; ++exp; }
            s.base_scales[i] = scale;
            s.base_exponents[i] = exp;
        }
        s.sample_stride = s.base_scales[0] * s.base_scales[1];
        auto extendedGCD = [](auto &&self, uint64_t a, uint64_t b, int64_t *x, int64_t *y) -> void {
            if (b == 0) { *x = 1; *y = 0; return; }
            int64_t dd = a / b, xp, yp;
            self(self, b, a % b, &xp, &yp);
            *x = yp;
            *y = xp - (dd * yp);
        };
        auto multInv = [&](int64_t a, int64_t n) {
            int64_t x, y;
            extendedGCD(extendedGCD, a, n, &x, &y);
            int64_t r = x - (x / n) * n;  // Mod(), pbrt.h:317-320
            return (int)((r < 0) ? r + n : r);
        };
        s.mult_inverse[0] = multInv(s.base_scales[1], s.base_scales[0]);
        s.mult_inverse[1] = multInv(s.base_scales[0], s.base_scales[1]);
    }
    // ---- integrator (CreatePathIntegrator, path.cpp:190-213)
    {
        scene->integratorName = integratorName;
        if (integratorName != "path" && integratorName != "spectralpath")
            Err("Integrator \"" + integratorName + "\" is outside the hot-path scope (SURVEY 2 row 7); using path.");
        mi_integrator &it = d.integrator;
        it.n_ca_bands = 1;
        if (integratorName == "spectralpath") {  // CreateSpectralPathIntegrator, spectralpath.cpp:342-376
            it.n_ca_bands = integratorParams.FindOneInt("numCABands", 4);
            if (it.n_ca_bands < 1) { Err("\"numCABands\" must be at least 1."); it.n_ca_bands = 1; }
            if (it.n_ca_bands != 1)
                Warn("Using spectral rendering. For every pixel sample we will trace " + std::to_string(it.n_ca_bands) +
                     "x more rays. Rendering will be " + std::to_string(it.n_ca_bands) + " times slower.");
        }
        it.max_depth = integratorParams.FindOneInt("maxdepth", 5);
        if (ov.maxDepth >= 0) it.max_depth = ov.maxDepth;
        for (int i = 0; i < 4; ++i) it.pixel_bounds[i] = d.film.sample_bounds[i];
        const std::vector<int> *pb = ParamSet::Find(integratorParams.ints, "pixelbounds");
        if (pb) {
            if (pb->size() != 4) Err("Expected four values for \"pixelbounds\" parameter.");
            else {
                it.pixel_bounds[0] = std::max(it.pixel_bounds[0], (*pb)[0]);
                it.pixel_bounds[1] = std::max(it.pixel_bounds[1], (*pb)[2]);
                it.pixel_bounds[2] = std::min(it.pixel_bounds[2], (*pb)[1]);
                it.pixel_bounds[3] = std::min(it.pixel_bounds[3], (*pb)[3]);
                if ((it.pixel_bounds[2] - it.pixel_bounds[0]) * (it.pixel_bounds[3] - it.pixel_bounds[1]) == 0)
                    Err("Degenerate \"pixelbounds\" specified.");
            }
        }
        it.rr_threshold = integratorParams.FindOneFloat("rrthreshold", 1.);
        scene->lightStrategy = integratorParams.FindOneString("lightsamplestrategy", "spatial");
        if (!ov.lightStrategy.empty()) scene->lightStrategy = ov.lightStrategy;
    }
    // Halton tables: 5 camera dims + per path and bounce 7 (+1 RR) for bounces 0..maxDepth-1
    {
        int nDims = 5 + d.integrator.n_ca_bands * 8 * (d.integrator.max_depth + 1) + 8;
        nDims = std::max(64, std::min(nDims, 1000));  // PrimeTableSize = 1000
        ComputeHaltonTables(nDims, &scene->primes, &scene->primeSums, &scene->perms);
        d.sampler.n_dims = nDims;
        if (d.sampler.type == MI_SAMPLER_SOBOL) {
            const int extent = std::max(d.film.sample_bounds[2] - d.film.sample_bounds[0], d.film.sample_bounds[3] - d.film.sample_bounds[1]);
            const int nSobol = std::min(nDims, 1024);   // NumSobolDimensions, sobolmatrices.h:47
            if (!ComputeSobolTables(extent, nSobol, scene, &d.sampler.sobol_resolution, &d.sampler.sobol_log2_resolution)) {
                Err("Sampler \"sobol\": film resolution beyond the tabulated pixel-index matrices; using halton.");
                d.sampler.type = MI_SAMPLER_HALTON;
            } else d.sampler.n_sobol_dims = nSobol;
        }
    }
    // ---- accelerator (CreateBVHAccelerator, bvh.cpp:740-762)
    {
        if (accelName != "bvh") Err("Accelerator \"" + accelName + "\" is outside the hot-path scope (SURVEY 2 row 9); using bvh.");
        std::string sm = accelParams.FindOneString("splitmethod", "sah");
        SplitMethod method = SplitMethod::SAH;
        if (sm == "sah") method = SplitMethod::SAH;
        else if (sm == "middle") method = SplitMethod::Middle;
        else if (sm == "equal") method = SplitMethod::EqualCounts;
        else if (sm == "hlbvh") method = SplitMethod::HLBVH;
        else Warn("BVH split method \"" + sm + "\" unknown.  Using \"sah\".");
        int maxPrims = accelParams.FindOneInt("maxnodeprims", 4);
        std::vector<Bounds3> bounds(pending.size());
        for (size_t i = 0; i < pending.size(); ++i) bounds[i] = pending[i].bounds;
        std::vector<int> order;
        const auto tb0 = std::chrono::steady_clock::now();
        if (method == SplitMethod::HLBVH) {
            // on the device when there is one (mi_bvh_build_hlbvh); the host restatement builds the same tree, node for node
            std::string why;
            double secs = 0;
            const char *where = getenv("MIPT_HLBVH");   // "host": do not try the device
            bool onDevice = !(where && std::string(where) == "host") &&
                            BuildHLBVHOnDevice(bounds, maxPrims, 0, &scene->nodes, &order, &scene->stats.interiorNodes, &scene->stats.leafNodes, &secs, &why);
            if (!onDevice) BuildHLBVH(bounds, maxPrims, &scene->nodes, &order, &scene->stats.interiorNodes, &scene->stats.leafNodes);
            scene->hlbvhOnDevice = onDevice;
            if (getenv("MIPT_TIMING")) fprintf(stderr, "[mipt] HLBVH on the %s%s%s\n", onDevice ? "device" : "host", onDevice ? "" : ": ", onDevice ? "" : why.c_str());
        } else
            BuildBVH(bounds, maxPrims, method, &scene->nodes, &order, &scene->stats.interiorNodes, &scene->stats.leafNodes);
        if (getenv("MIPT_TIMING"))
            fprintf(stderr, "[mipt] BVH build over %zu primitives: %.3f s\n", bounds.size(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count());
        scene->prims.resize(order.size());
        for (size_t i = 0; i < order.size(); ++i) {
            const PendingPrim &pp = pending[order[i]];
            mi_prim p{};
            p.shape = pp.shape;
            p.material = pp.material;
            p.area_light = pp.light;
            p.instance = pp.instance;
            scene->prims[i] = p;
        }
        // the objects' own BVHs (MakeAccelerator over the object's primitives at its first ObjectInstance, api.cpp:1583-1590),
        // appended to the node and primitive arrays with absolute offsets; then the instances that point at them
        for (const std::string &name : objectOrder) {
            ObjectDef &od = objectDefs[name];
            if (od.prims.empty()) continue;
            std::vector<Bounds3> ob(od.prims.size());
            for (size_t i = 0; i < od.prims.size(); ++i) ob[i] = od.prims[i].bounds;
            std::vector<mi_bvh_node> onodes;
            std::vector<int> oorder;
            int oi = 0, ol = 0;
            if (method == SplitMethod::HLBVH) BuildHLBVH(ob, maxPrims, &onodes, &oorder, &oi, &ol);
            else BuildBVH(ob, maxPrims, method, &onodes, &oorder, &oi, &ol);
            const int baseNode = (int)scene->nodes.size(), basePrim = (int)scene->prims.size();
            for (mi_bvh_node &n : onodes) n.offset += (n.n_prims > 0) ? basePrim : baseNode;
            scene->nodes.insert(scene->nodes.end(), onodes.begin(), onodes.end());
            for (int k : oorder) {
                const PendingPrim &pp = od.prims[k];
                mi_prim p{};
                p.shape = pp.shape; p.material = pp.material; p.area_light = pp.light; p.instance = 0;
                scene->prims.push_back(p);
            }
            od.root = baseNode;
            scene->stats.interiorNodes += oi;
            scene->stats.leafNodes += ol;
        }
        for (const InstanceRec &ir : instanceRecs) {
            mi_instance mi{};
            std::memcpy(mi.i2w, ir.i2w.m.m, sizeof(float) * 16);
            std::memcpy(mi.w2i, ir.i2w.mInv.m, sizeof(float) * 16);
            mi.root = (uint32_t)objectDefs[ir.object].root;
            scene->instances.push_back(mi);
        }
    }
    // ---- lights: Preprocess (distant.h:52-54) + selection distribution
    {
        Bounds3 wb;
        if (!scene->nodes.empty()) {
            wb.pMin = Vec3(scene->nodes[0].bmin[0], scene->nodes[0].bmin[1], scene->nodes[0].bmin[2]);
            wb.pMax = Vec3(scene->nodes[0].bmax[0], scene->nodes[0].bmax[1], scene->nodes[0].bmax[2]);
        } else
            wb = Bounds3();
        Vec3 c;
        float r;
        wb.BoundingSphere(&c, &r);
        for (auto &l : scene->lights) {
            l.world_radius = r;
            l.world_center[0] = c.x; l.world_center[1] = c.y; l.world_center[2] = c.z;
        }
        if (scene->lights.empty())
            Warn("No light sources defined in scene; rendering a black image.");
        BuildLightDistribution(scene, scene->lightStrategy);
    }
    scene->stats.nLights = (int)scene->lights.size();
    scene->stats.nMaterials = (int)scene->materials.size();
    const float *Y = Spectrum::CIE_Y();
    for (int i = 0; i < MI_NSPEC; ++i) d.cie_y[i] = Y[i];
    for (int k = 0; k < 7; ++k) {
        const float *basis = Spectrum::RGBIllumBasis(k);
        for (int i = 0; i < MI_NSPEC; ++i) d.rgb_illum[k][i] = basis[i];
    }
    scene->Finalize();
}

// ReadFloatFile, src/core/floatfile.cpp:40-83 (including its habit of dropping a number that ends at end-of-file)
bool ReadFloatFile(const std::string &filename, std::vector<float> *values, Api *api) {
    FILE *f = fopen(filename.c_str(), "r");
    if (!f) { api->Err("Unable to open file \"" + filename + "\""); return false; }
    int c;
    bool inNumber = false;
    char curNumber[32];
    int curNumberPos = 0;
    int lineNumber = 1;
    while ((c = getc(f)) != EOF) {
        if (c == '\n') ++lineNumber;
        if (inNumber) {
            if (curNumberPos >= (int)sizeof(curNumber) - 1) { api->Err("Overflowed buffer for parsing number in file: " + filename); fclose(f); return false; }
            if (isdigit(c) || c == '.' || c == 'e' || c == '-' || c == '+') curNumber[curNumberPos++] = (char)c;
            else {
                curNumber[curNumberPos++] = '\0';
                values->push_back((float)atof(curNumber));
                inNumber = false;
                curNumberPos = 0;
            }
        } else {
            if (isdigit(c) || c == '.' || c == '-' || c == '+') { inNumber = true; curNumber[curNumberPos++] = (char)c; }
            else if (c == '#') {
                while ((c = getc(f)) != '\n' && c != EOF) {}
                ++lineNumber;
            } else if (!isspace(c))
                api->Warn("Unexpected text found at line " + std::to_string(lineNumber) + " of float file \"" + filename + "\"");
        }
    }
    fclose(f);
    return true;
}

// Blackbody / BlackbodyNormalized, src/core/spectrum.cpp:1009-1034
void Blackbody(const float *lambda, int n, float T, float *Le) {
    if (T <= 0) { for (int i = 0; i < n; ++i) Le[i] = 0.f; return; }
    const float c = 299792458;
    const float h = 6.62606957e-34;
    const float kb = 1.3806488e-23;
    for (int i = 0; i < n; ++i) {
        float l = lambda[i] * 1e-9;
        float lambda5 = (l * l) * (l * l) * l;
        Le[i] = (2 * h * c * c) / (lambda5 * (std::exp((h * c) / (l * kb * T)) - 1));
    }
}
void BlackbodyNormalized(const float *lambda, int n, float T, float *Le) {
    Blackbody(lambda, n, T, Le);
    float lambdaMax = 2.8977721e-3 / T * 1e9;
    float maxL;
    Blackbody(&lambdaMax, 1, T, &maxL);
    for (int i = 0; i < n; ++i) Le[i] /= maxL;
}

// ------------------------------------------------------------------ parser
enum ParamType { PT_INT, PT_BOOL, PT_FLOAT, PT_POINT2, PT_VECTOR2, PT_POINT3, PT_VECTOR3, PT_NORMAL, PT_RGB, PT_XYZ,
                 PT_BLACKBODY, PT_SPECTRUM, PT_STRING, PT_TEXTURE, PT_UNKNOWN };

bool LookupType(const std::string &decl, ParamType *type, std::string *name) {  // parser.cpp:440-520
    size_t i = 0;
    auto skip = [&]() { while (i < decl.size() && (decl[i] == ' ' || decl[i] == '\t')) ++i; };
    skip();
    size_t ts = i;
    while (i < decl.size() && decl[i] != ' ' && decl[i] != '\t') ++i;
    std::string t = decl.substr(ts, i - ts);
    skip();
    size_t ns = i;
    while (i < decl.size() && decl[i] != ' ' && decl[i] != '\t') ++i;
    *name = decl.substr(ns, i - ns);
    if (t.empty() || name->empty()) return false;
    if (t == "float") *type = PT_FLOAT;
    else if (t == "integer") *type = PT_INT;
    else if (t == "bool") *type = PT_BOOL;
    else if (t == "point2") *type = PT_POINT2;
    else if (t == "vector2") *type = PT_VECTOR2;
    else if (t == "point3" || t == "point") *type = PT_POINT3;
    else if (t == "vector3" || t == "vector") *type = PT_VECTOR3;
    else if (t == "normal") *type = PT_NORMAL;
    else if (t == "string") *type = PT_STRING;
    else if (t == "texture") *type = PT_TEXTURE;
    else if (t == "color" || t == "rgb") *type = PT_RGB;
    else if (t == "xyz") *type = PT_XYZ;
    else if (t == "blackbody") *type = PT_BLACKBODY;
    else if (t == "spectrum") *type = PT_SPECTRUM;
    else { *type = PT_UNKNOWN; return false; }
    return true;
}

struct Parser {
    Api *api;
    std::vector<Tokenizer> stack;
    bool hasUnget = false;
    std::string ungetTok;
    bool ungetQuoted = false;
    std::string error;

    bool Next(std::string *tok, bool *quoted) {
        if (hasUnget) { hasUnget = false; *tok = ungetTok; *quoted = ungetQuoted; return true; }
        while (!stack.empty()) {
            std::string err;
            if (stack.back().Next(tok, quoted, &err)) return true;
            if (!err.empty()) { error = stack.back().filename + ":" + std::to_string(stack.back().line) + ": " + err; return false; }
            stack.pop_back();
        }
        return false;
    }
    void Unget(const std::string &tok, bool quoted) { hasUnget = true; ungetTok = tok; ungetQuoted = quoted; }
    std::string Loc() { return stack.empty() ? std::string("<eof>") : stack.back().filename + ":" + std::to_string(stack.back().line); }
    bool Fail(const std::string &m) { if (error.empty()) error = Loc() + ": " + m; return false; }

    bool ParseParams(ParamSet *ps) {  // parser.cpp:707-780 + AddParam 520-705
        while (true) {
            std::string decl;
            bool q;
            if (!Next(&decl, &q)) return error.empty();
            if (!q) { Unget(decl, q); return true; }
            ParamType type = PT_UNKNOWN;
            std::string name;
            bool known = LookupType(decl, &type, &name);
            std::vector<double> nums;
            std::vector<std::string> strs;
            std::string val;
            bool vq;
            if (!Next(&val, &vq)) return Fail("premature EOF in parameter list");
            auto addVal = [&](const std::string &v, bool isq) -> bool {
                if (isq) {
                    if (!nums.empty()) return Fail("mixed string and numeric parameters");
                    strs.push_back(v);
                } else {
                    if (!strs.empty()) return Fail("mixed string and numeric parameters");
                    bool ok;
                    double dv = ParseNumber(v, &ok);
                    if (!ok) return Fail("\"" + v + "\": expected a number");
                    nums.push_back(dv);
                }
                return true;
            };
            if (!vq && val == "[") {
                while (true) {
                    if (!Next(&val, &vq)) return Fail("premature EOF in parameter list");
                    if (!vq && val == "]") break;
                    if (!addVal(val, vq)) return false;
                }
            } else if (!addVal(val, vq)) return false;
            if (!known) { api->Warn("Type of parameter \"" + decl + "\" is unknown"); continue; }
            bool wantString = (type == PT_STRING || type == PT_TEXTURE || type == PT_BOOL);
            if (type == PT_SPECTRUM && !strs.empty()) {  // AddSampledSpectrumFiles, paramset.cpp:171-207
                std::vector<Spectrum> v;
                for (const std::string &nm : strs) {
                    const std::string fn = (!nm.empty() && nm[0] != '/') ? api->baseDir + "/" + nm : nm;
                    auto cached = api->cachedSpectra.find(fn);
                    if (cached != api->cachedSpectra.end()) { v.push_back(cached->second); continue; }
                    std::vector<float> vals;
                    Spectrum sp(0.f);
                    if (!ReadFloatFile(fn, &vals, api))
                        api->Warn("Unable to read SPD file \"" + fn + "\".  Using black distribution.");
                    else {
                        if (vals.size() % 2) api->Warn("Extra value found in spectrum file \"" + fn + "\". Ignoring it.");
                        std::vector<float> wls, vv;
                        for (size_t j = 0; j < vals.size() / 2; ++j) { wls.push_back(vals[2 * j]); vv.push_back(vals[2 * j + 1]); }
                        if (!wls.empty()) sp = Spectrum::FromSampled(wls.data(), vv.data(), (int)wls.size());
                    }
                    api->cachedSpectra[fn] = sp;
                    v.push_back(sp);
                }
                ParamSet::Add(ps->spectra, name, v);
                continue;
            }
            if (wantString && strs.empty() && !nums.empty()) { api->Err("Expected string parameter value for \"" + name + "\""); continue; }
            if (!wantString && !strs.empty()) { api->Err("Expected numeric parameter value for \"" + name + "\""); continue; }
            size_t n = nums.size();
            auto f = [&](size_t i) { return (float)nums[i]; };
            switch (type) {
            case PT_INT: { std::vector<int> v(n); for (size_t i = 0; i < n; ++i) v[i] = int(nums[i]); ParamSet::Add(ps->ints, name, v); break; }
            case PT_BOOL: {
                std::vector<bool> v;
                for (auto &s : strs) {
                    if (s == "true") v.push_back(true);
                    else if (s == "false") v.push_back(false);
                    else { api->Warn("Value \"" + s + "\" unknown for Boolean parameter \"" + name + "\". Using \"false\"."); v.push_back(false); }
                }
                ParamSet::Add(ps->bools, name, v);
                break;
            }
            case PT_FLOAT: { std::vector<float> v(n); for (size_t i = 0; i < n; ++i) v[i] = f(i); ParamSet::Add(ps->floats, name, v); break; }
            case PT_POINT2: case PT_VECTOR2: {
                if (n % 2) api->Warn("Excess values given with point2 parameter \"" + name + "\". Ignoring last one of them.");
                std::vector<Vec2> v(n / 2);
                for (size_t i = 0; i < n / 2; ++i) { v[i].x = f(2 * i); v[i].y = f(2 * i + 1); }
                ParamSet::Add(ps->point2s, name, v);
                break;
            }
            case PT_POINT3: case PT_VECTOR3: case PT_NORMAL: {
                if (n % 3) api->Warn("Excess values given with parameter \"" + name + "\". Ignoring extra.");
                std::vector<Vec3> v(n / 3);
                for (size_t i = 0; i < n / 3; ++i) v[i] = Vec3(f(3 * i), f(3 * i + 1), f(3 * i + 2));
                ParamSet::Add(type == PT_POINT3 ? ps->point3s : (type == PT_VECTOR3 ? ps->vector3s : ps->normals), name, v);
                break;
            }
            case PT_RGB: case PT_XYZ: {
                if (n % 3) { api->Warn("Excess RGB values given with parameter \"" + name + "\"."); n -= n % 3; }
                std::vector<Spectrum> v;
                for (size_t i = 0; i < n / 3; ++i) {
                    float c[3] = {f(3 * i), f(3 * i + 1), f(3 * i + 2)};
                    // AddRGBSpectrum / AddXYZSpectrum use FromRGB / FromXYZ defaults (paramset.cpp:110-131)
                    v.push_back(type == PT_RGB ? Spectrum::FromRGB(c) : Spectrum::FromXYZ(c));
                }
                ParamSet::Add(ps->spectra, name, v);
                break;
            }
            case PT_SPECTRUM: {  // AddSampledSpectrum, paramset.cpp:152-169
                if (n % 2) { api->Warn("Non-even number of values given with sampled spectrum parameter \"" + name + "\". Ignoring extra."); n -= n % 2; }
                if (n == 0) { api->Err("empty sampled spectrum \"" + name + "\""); break; }
                std::vector<float> wl(n / 2), vv(n / 2);
                for (size_t i = 0; i < n / 2; ++i) { wl[i] = f(2 * i); vv[i] = f(2 * i + 1); }
                std::vector<Spectrum> v(1, Spectrum::FromSampled(wl.data(), vv.data(), (int)(n / 2)));
                ParamSet::Add(ps->spectra, name, v);
                break;
            }
            case PT_BLACKBODY: {  // AddBlackbodySpectrum, paramset.cpp:133-150: (temperature K, scale) pairs
                if (n % 2) { api->Warn("Excess value given with blackbody parameter \"" + name + "\". Ignoring extra one."); n -= n % 2; }
                std::vector<Spectrum> v;
                const int nCIESamples = 471;   // CIE_lambda = 360 .. 830 nm, spectrum.cpp:543
                std::vector<float> lambda(nCIESamples), le(nCIESamples);
                for (int i = 0; i < nCIESamples; ++i) lambda[i] = (float)(360 + i);
                for (size_t i = 0; i < n / 2; ++i) {
                    BlackbodyNormalized(lambda.data(), nCIESamples, f(2 * i), le.data());
                    v.push_back(f(2 * i + 1) * Spectrum::FromSampled(lambda.data(), le.data(), nCIESamples));
                }
                ParamSet::Add(ps->spectra, name, v);
                break;
            }
            case PT_STRING: ParamSet::Add(ps->strings, name, strs); break;
            case PT_TEXTURE:
                if (strs.size() == 1) ParamSet::Add(ps->textures, name, strs);
                else api->Err("Only one string allowed for \"texture\" parameter \"" + name + "\"");
                break;
            default: break;
            }
        }
    }

    bool ReadFloats(int n, float *out) {
        for (int i = 0; i < n; ++i) {
            std::string t; bool q;
            if (!Next(&t, &q) || q) return Fail("expected a number");
            bool ok;
            out[i] = (float)ParseNumber(t, &ok);
            if (!ok) return Fail("\"" + t + "\": expected a number");
        }
        return true;
    }
    bool ReadString(std::string *s) {
        bool q;
        if (!Next(s, &q) || !q) return Fail("expected quoted string");
        return true;
    }
    bool OpenFile(const std::string &path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        std::stringstream ss;
        ss << in.rdbuf();
        Tokenizer t;
        t.text = ss.str();
        t.filename = path;
        stack.push_back(std::move(t));
        return true;
    }

    bool Run();
};

bool Parser::Run() {
    Api &a = *api;
    std::string tok;
    bool quoted;
    auto needWorld = [&](const char *n) {
        if (a.state != Api::World) { a.Err(std::string("Scene description must be inside world block; \"") + n + "\" not allowed. Ignoring."); return false; }
        return true;
    };
    auto needOptions = [&](const char *n) {
        if (a.state != Api::Options) { a.Err(std::string("Options cannot be set inside world block; \"") + n + "\" not allowed.  Ignoring."); return false; }
        return true;
    };
    while (Next(&tok, &quoted)) {
        if (quoted) return Fail("unexpected string \"" + tok + "\"");
        if (tok == "AttributeBegin") {
            if (!needWorld("AttributeBegin")) continue;
            a.gsStack.push_back(a.gs); a.transformStack.push_back(a.ctm); a.pushKinds.push_back('a');
        } else if (tok == "AttributeEnd") {
            if (!needWorld("AttributeEnd")) continue;
            if (a.gsStack.empty()) { a.Err("Unmatched AttributeEnd encountered. Ignoring it."); continue; }
            a.gs = a.gsStack.back(); a.gsStack.pop_back();
            a.ctm = a.transformStack.back(); a.transformStack.pop_back();
            a.pushKinds.pop_back();
        } else if (tok == "TransformBegin") {
            if (!needWorld("TransformBegin")) continue;
            a.transformStack.push_back(a.ctm); a.pushKinds.push_back('t');
        } else if (tok == "TransformEnd") {
            if (!needWorld("TransformEnd")) continue;
            if (a.transformStack.empty() || a.pushKinds.back() != 't') { a.Err("Unmatched TransformEnd encountered. Ignoring it."); continue; }
            a.ctm = a.transformStack.back(); a.transformStack.pop_back(); a.pushKinds.pop_back();
        } else if (tok == "ActiveTransform") {
            std::string w; bool q;
            if (!Next(&w, &q)) return Fail("premature EOF");
            if (w != "All") a.Err("ActiveTransform " + w + ": animated transforms are outside the hot-path scope; using one transform");
        } else if (tok == "Identity") a.ctm = Transform();
        else if (tok == "Translate") { float v[3]; if (!ReadFloats(3, v)) return false; a.ctm = a.ctm * Translate(Vec3(v[0], v[1], v[2])); }
        else if (tok == "Scale") { float v[3]; if (!ReadFloats(3, v)) return false; a.ctm = a.ctm * Scale(v[0], v[1], v[2]); }
        else if (tok == "Rotate") { float v[4]; if (!ReadFloats(4, v)) return false; a.ctm = a.ctm * Rotate(v[0], Vec3(v[1], v[2], v[3])); }
        else if (tok == "LookAt") {
            float v[9]; if (!ReadFloats(9, v)) return false;
            bool degenerate;
            Transform la = LookAt(Vec3(v[0], v[1], v[2]), Vec3(v[3], v[4], v[5]), Vec3(v[6], v[7], v[8]), &degenerate);
            if (degenerate) a.Err("\"up\" vector and viewing direction passed to LookAt are pointing in the same direction.  Using the identity transformation.");
            a.ctm = a.ctm * la;
        } else if (tok == "Transform" || tok == "ConcatTransform") {
            std::string b; bool q;
            if (!Next(&b, &q) || b != "[") return Fail("expected [");
            float tr[16]; if (!ReadFloats(16, tr)) return false;
            if (!Next(&b, &q) || b != "]") return Fail("expected ]");
            Transform t(Matrix4x4(tr[0], tr[4], tr[8], tr[12], tr[1], tr[5], tr[9], tr[13], tr[2], tr[6], tr[10], tr[14],
                                  tr[3], tr[7], tr[11], tr[15]));
            a.ctm = (tok == "Transform") ? t : a.ctm * t;
        } else if (tok == "CoordinateSystem") { std::string n; if (!ReadString(&n)) return false; a.namedCoordSys[n] = a.ctm; }
        else if (tok == "CoordSysTransform") {
            std::string n; if (!ReadString(&n)) return false;
            if (a.namedCoordSys.count(n)) a.ctm = a.namedCoordSys[n];
            else a.Warn("Couldn't find named coordinate system \"" + n + "\"");
        } else if (tok == "TransformTimes") { float v[2]; if (!ReadFloats(2, v)) return false; }
        else if (tok == "ReverseOrientation") { if (needWorld("ReverseOrientation")) a.gs.reverseOrientation = !a.gs.reverseOrientation; }
        else if (tok == "Include") {
            std::string fn; if (!ReadString(&fn)) return false;
            std::string path = (fn.size() && fn[0] == '/') ? fn : a.baseDir + "/" + fn;  // ResolveFilename
            if (!OpenFile(path)) a.Err("Couldn't open include file \"" + path + "\"");
        } else if (tok == "WorldBegin") {
            if (!needOptions("WorldBegin")) continue;
            a.state = Api::World;
            a.ctm = Transform();
            a.namedCoordSys["world"] = a.ctm;
        } else if (tok == "WorldEnd") {
            if (!needWorld("WorldEnd")) continue;
            while (!a.gsStack.empty()) { a.Warn("Missing end to AttributeBegin"); a.gsStack.pop_back(); a.transformStack.pop_back(); }
            a.WorldEnd();
            a.state = Api::Options;
            return true;  // one render per file on this path
        } else if (tok == "ObjectBegin" || tok == "ObjectInstance") {
            std::string n; if (!ReadString(&n)) return false;
            if (!needWorld(tok.c_str())) continue;
            if (tok == "ObjectBegin") a.ObjectBegin(n); else a.ObjectInstance(n);
        } else if (tok == "ObjectEnd") {
            if (!needWorld("ObjectEnd")) continue;
            a.ObjectEnd();
        }
        else if (tok == "MediumInterface") {
            std::string n; if (!ReadString(&n)) return false;
            std::string t2; bool q2;
            if (Next(&t2, &q2) && !q2) Unget(t2, q2);
            a.Warn("MediumInterface ignored: PathIntegrator does not handle media (path.cpp:122-123)");
        } else {
            // directives of the form:  Name "type" <params>
            static const char *named[] = {"Camera", "Film", "Sampler", "Accelerator", "Integrator", "PixelFilter", "Material",
                                          "MakeNamedMaterial", "NamedMaterial", "AreaLightSource", "LightSource", "Shape",
                                          "Texture", "MakeNamedMedium"};
            bool isNamed = false;
            for (const char *n : named) if (tok == n) isNamed = true;
            if (!isNamed) return Fail("unknown directive \"" + tok + "\"");
            std::string name;
            if (!ReadString(&name)) return false;
            std::string texType, texClass;
            if (tok == "Texture") { if (!ReadString(&texType) || !ReadString(&texClass)) return false; }
            ParamSet ps;
            if (tok != "NamedMaterial") { if (!ParseParams(&ps)) return false; }
            if (tok == "Camera") {
                if (!needOptions("Camera")) continue;
                a.cameraName = name; a.cameraParams = ps;
                a.cameraToWorld = Inverse(a.ctm);
                a.namedCoordSys["camera"] = a.cameraToWorld;
            } else if (tok == "Film") { if (needOptions("Film")) { a.filmName = name; a.filmParams = ps; } }
            else if (tok == "Sampler") { if (needOptions("Sampler")) { a.samplerName = name; a.samplerParams = ps; } }
            else if (tok == "Accelerator") { if (needOptions("Accelerator")) { a.accelName = name; a.accelParams = ps; } }
            else if (tok == "Integrator") { if (needOptions("Integrator")) { a.integratorName = name; a.integratorParams = ps; } }
            else if (tok == "PixelFilter") { if (needOptions("PixelFilter")) { a.filterName = name; a.filterParams = ps; } }
            else if (tok == "Material") {
                if (!needWorld("Material")) continue;
                ParamSet empty;
                auto mi = std::make_shared<MaterialInstance>();
                mi->name = name; mi->params = ps;
                mi->material = a.MakeMaterial(name, ps, empty);
                a.gs.currentMaterial = mi;
            } else if (tok == "MakeNamedMaterial") {
                if (!needWorld("MakeNamedMaterial")) continue;
                ParamSet empty;
                std::string matName = ps.FindOneString("type", "");
                if (matName == "") { a.Err("No parameter string \"type\" found in MakeNamedMaterial"); continue; }
                auto mi = std::make_shared<MaterialInstance>();
                mi->name = matName; mi->params = ps;
                mi->material = a.MakeMaterial(matName, ps, empty);
                if (a.gs.namedMaterials.count(name)) a.Warn("Named material \"" + name + "\" redefined.");
                a.gs.namedMaterials[name] = mi;
            } else if (tok == "NamedMaterial") {
                if (!needWorld("NamedMaterial")) continue;
                auto it = a.gs.namedMaterials.find(name);
                if (it == a.gs.namedMaterials.end()) { a.Err("NamedMaterial \"" + name + "\" unknown."); continue; }
                a.gs.currentMaterial = it->second;
            } else if (tok == "AreaLightSource") { if (needWorld("AreaLightSource")) { a.gs.areaLight = name; a.gs.areaLightParams = ps; } }
            else if (tok == "LightSource") a.LightSource(name, ps);
            else if (tok == "Shape") a.Shape(name, ps);
            else if (tok == "Texture") { if (needWorld("Texture")) a.Texture(name, texType, texClass, ps); }
            else if (tok == "MakeNamedMedium") a.Warn("MakeNamedMedium ignored: media are outside the hot-path scope");
        }
    }
    return error.empty();
}

HostScene *Load(Parser &p, Api &api, HostScene *scene, std::string *err) {
    bool ok = p.Run();
    if (!ok) { *err = p.error.empty() ? "parse error" : p.error; delete scene; return nullptr; }
    if (!api.worldEnded) { *err = "scene file has no WorldEnd"; delete scene; return nullptr; }
    return scene;
}

}  // namespace

void HostScene::Finalize() {
    mi_scene_desc &d = desc;
    d.abi_version = MI_ABI_VERSION;
    mipmaps.clear();
    for (const HostMipMap &h : mipStore) {
        mi_mipmap m{};
        m.n_levels = (int)h.levelOffset.size(); m.wrap = h.wrap; m.width = h.width; m.height = h.height;
        m.texels = h.texels.data();
        for (size_t l = 0; l < h.levelOffset.size() && l < MI_MAX_MIP_LEVELS; ++l) m.level_offset[l] = h.levelOffset[l];
        mipmaps.push_back(m);
    }
    d.n_mipmaps = (uint32_t)mipmaps.size();
    d.mipmaps = mipmaps.empty() ? nullptr : mipmaps.data();
    d.n_textures = (uint32_t)textures.size();
    d.textures = textures.empty() ? nullptr : textures.data();
    envmaps.clear();
    for (const HostEnvMap &e : envStore) {
        mi_envmap m{};
        m.width = e.width; m.height = e.height; m.rgb = e.rgb.data();
        m.nu = e.nu; m.nv = e.nv;
        m.cond_func = e.condFunc.data(); m.cond_cdf = e.condCdf.data(); m.cond_func_int = e.condFuncInt.data();
        m.marg_func = e.margFunc.data(); m.marg_cdf = e.margCdf.data(); m.marg_func_int = e.margFuncInt;
        envmaps.push_back(m);
    }
    d.n_envmaps = (uint32_t)envmaps.size();
    d.envmaps = envmaps.empty() ? nullptr : envmaps.data();
    d.n_nodes = (uint32_t)nodes.size(); d.nodes = nodes.data();
    d.n_prims = (uint32_t)prims.size(); d.prims = prims.data();
    d.n_tris = (uint32_t)(triIndices.size() / 3); d.tri_indices = triIndices.data(); d.tri_mesh = triMesh.data();
    d.n_verts = (uint32_t)(P.size() / 3); d.P = P.data(); d.N = N.data(); d.UV = UV.data();
    d.n_meshes = (uint32_t)meshes.size(); d.meshes = meshes.data();
    d.n_spheres = (uint32_t)spheres.size(); d.spheres = spheres.data();
    d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
    d.n_lights = (uint32_t)lights.size(); d.lights = lights.data();
    d.n_instances = (uint32_t)instances.size(); d.instances = instances.empty() ? nullptr : instances.data();
    d.light_distrib.func = ldFunc.empty() ? nullptr : ldFunc.data();
    d.light_distrib.cdf = ldCdf.empty() ? nullptr : ldCdf.data();
    d.light_distrib.func_int = ldFuncInt.empty() ? nullptr : ldFuncInt.data();
    d.sampler.primes = primes.data();
    d.sampler.prime_sums = primeSums.data();
    d.sampler.perms = perms.data();
    d.sampler.n_perms = (uint32_t)perms.size();
    d.sampler.sobol_matrices = sobolMatrices.empty() ? nullptr : sobolMatrices.data();
    d.sampler.sobol_vdc = sobolVdc.empty() ? nullptr : sobolVdc.data();
    d.sampler.sobol_vdc_inv = sobolVdcInv.empty() ? nullptr : sobolVdcInv.data();
}

HostScene *LoadSceneFile(const std::string &path, const LoadOverrides &ov, std::string *err) {
    HostScene *scene = new HostScene();
    Api api(scene, ov);
    size_t slash = path.find_last_of('/');
    api.baseDir = (slash == std::string::npos) ? "." : path.substr(0, slash);
    Parser p;
    p.api = &api;
    if (!p.OpenFile(path)) { *err = "Couldn't open scene file \"" + path + "\""; delete scene; return nullptr; }
    return Load(p, api, scene, err);
}

HostScene *LoadSceneString(const std::string &text, const std::string &baseDir, const LoadOverrides &ov, std::string *err) {
    HostScene *scene = new HostScene();
    Api api(scene, ov);
    api.baseDir = baseDir.empty() ? "." : baseDir;
    Parser p;
    p.api = &api;
    Tokenizer t;
    t.text = text;
    t.filename = "<string>";
    p.stack.push_back(std::move(t));
    return Load(p, api, scene, err);
}

}  // namespace mipt
