// paramset.h -- typed key/value bag for .pbrt directives. Same lookup names and
// defaults-on-miss behaviour as the reference's ParamSet / TextureParams
// (src/core/paramset.h:95-118, paramset.cpp:110-120,396-470,700-760): FindOne*
// returns the default unless exactly one value is present; "color"/"rgb" values
// are converted with Spectrum::FromRGB at parse time (Illuminant basis in this
// fork, paramset.cpp:116). Only constant textures exist on this path, so a
// texture lookup resolves to a value.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "ptmath.h"
#include "spectrum.h"

namespace mipt {

template <typename T>
struct ParamItem {
    std::string name;
    std::vector<T> values;
    mutable bool lookedUp = false;
};

struct Vec2 { float x = 0, y = 0; };

struct ParamSet {
    std::vector<ParamItem<bool>> bools;
    std::vector<ParamItem<int>> ints;
    std::vector<ParamItem<float>> floats;
    std::vector<ParamItem<Vec2>> point2s;
    std::vector<ParamItem<Vec3>> point3s, vector3s, normals;
    std::vector<ParamItem<Spectrum>> spectra;
    std::vector<ParamItem<std::string>> strings, textures;

    template <typename T>
    static void Add(std::vector<ParamItem<T>> &v, const std::string &name, std::vector<T> vals) {
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].name == name) { v.erase(v.begin() + i); break; }
        ParamItem<T> it;
        it.name = name;
        it.values = std::move(vals);
        v.push_back(std::move(it));
    }
    template <typename T>
    static const std::vector<T> *Find(const std::vector<ParamItem<T>> &v, const std::string &name) {
        for (const auto &it : v)
            if (it.name == name) { it.lookedUp = true; return &it.values; }
        return nullptr;
    }
    template <typename T>
    static T FindOne(const std::vector<ParamItem<T>> &v, const std::string &name, const T &d) {
        for (const auto &it : v)
            if (it.name == name && it.values.size() == 1) { it.lookedUp = true; return it.values[0]; }
        return d;
    }
    bool FindOneBool(const std::string &n, bool d) const { return FindOne(bools, n, d); }
    int FindOneInt(const std::string &n, int d) const { return FindOne(ints, n, d); }
    float FindOneFloat(const std::string &n, float d) const { return FindOne(floats, n, d); }
    Vec3 FindOnePoint3(const std::string &n, const Vec3 &d) const { return FindOne(point3s, n, d); }
    Vec3 FindOneVector3(const std::string &n, const Vec3 &d) const { return FindOne(vector3s, n, d); }
    Spectrum FindOneSpectrum(const std::string &n, const Spectrum &d) const { return FindOne(spectra, n, d); }
    std::string FindOneString(const std::string &n, const std::string &d) const { return FindOne(strings, n, d); }
    std::string FindTexture(const std::string &n) const { return FindOne(textures, n, std::string("")); }
    void ReportUnused(std::vector<std::string> *out) const {
        auto chk = [&](const auto &v) { for (const auto &it : v) if (!it.lookedUp) out->push_back(it.name); };
        chk(bools); chk(ints); chk(floats); chk(point2s); chk(point3s); chk(vector3s); chk(normals);
        chk(spectra); chk(strings); chk(textures);
    }
};

// Named constant textures (Texture "name" "float|spectrum" "constant" "... value").
struct TextureMaps {
    std::map<std::string, float> floatTex;
    std::map<std::string, Spectrum> spectrumTex;
    std::map<std::string, int> imageTex;   // Texture "name" "spectrum" "imagemap": index into HostScene::textures
    std::map<std::string, int> floatImageTex;   // Texture "name" "float" "imagemap" (alpha masks, bump maps)
    std::map<std::string, std::pair<int, Spectrum>> scaledImageTex;   // spectrum "scale" of an image texture and a constant
};
// A spectrum material parameter: a constant, or an image texture evaluated per hit.
struct SpectrumParam {
    Spectrum s;          // the constant; with `scaled`: the constant factor of a "scale" texture over image texture `tex`
    int tex = -1;
    bool scaled = false;
    SpectrumParam() {}
    SpectrumParam(const Spectrum &v) : s(v) {}
};

// TextureParams (paramset.cpp:700-790): geometry params shadow material params.
struct TextureParams {
    const ParamSet &geom, &mat;
    const TextureMaps &tex;
    std::vector<std::string> *errors;
    TextureParams(const ParamSet &g, const ParamSet &m, const TextureMaps &t, std::vector<std::string> *e)
        : geom(g), mat(m), tex(t), errors(e) {}

    bool GetSpectrumOrNull(const std::string &n, Spectrum *out) const {
        std::string name = geom.FindTexture(n);
        if (name == "") {
            const std::vector<Spectrum> *s = ParamSet::Find(geom.spectra, n);
            if (s && s->size() >= 1) { *out = (*s)[0]; return true; }  // count>1 only warns upstream
            name = mat.FindTexture(n);
            if (name == "") {
                const std::vector<Spectrum> *s2 = ParamSet::Find(mat.spectra, n);
                if (s2 && s2->size() >= 1) { *out = (*s2)[0]; return true; }
                return false;
            }
        }
        auto it = tex.spectrumTex.find(name);
        if (it != tex.spectrumTex.end()) { *out = it->second; return true; }
        if (tex.imageTex.count(name) || tex.scaledImageTex.count(name)) {
            if (errors) errors->push_back("Image texture \"" + name + "\" on parameter \"" + n + "\": this path evaluates image textures "
                                          "for Kd / Ks / Kr / Kt of matte, plastic, mirror, glass, uber, substrate and translucent only");
            return false;
        }
        if (errors) errors->push_back("Couldn't find spectrum texture named \"" + name + "\" for parameter \"" + n + "\"");
        return false;
    }
    // The same lookup for a parameter that may be bound to an image texture.
    SpectrumParam GetSpectrumParam(const std::string &n, const Spectrum &def) const {
        std::string name = geom.FindTexture(n);
        if (name == "" && !ParamSet::Find(geom.spectra, n)) name = mat.FindTexture(n);
        if (name != "") {
            auto it = tex.imageTex.find(name);
            if (it != tex.imageTex.end()) { SpectrumParam p; p.s = Spectrum(1.f); p.tex = it->second; return p; }
            auto sc = tex.scaledImageTex.find(name);
            if (sc != tex.scaledImageTex.end()) { SpectrumParam p; p.s = sc->second.second; p.tex = sc->second.first; p.scaled = true; return p; }
        }
        return SpectrumParam(GetSpectrum(n, def));
    }
    Spectrum GetSpectrum(const std::string &n, const Spectrum &def) const {
        Spectrum s;
        return GetSpectrumOrNull(n, &s) ? s : def;
    }
    bool GetFloatOrNull(const std::string &n, float *out) const {
        std::string name = geom.FindTexture(n);
        if (name == "") {
            const std::vector<float> *s = ParamSet::Find(geom.floats, n);
            if (s && s->size() >= 1) { *out = (*s)[0]; return true; }
            name = mat.FindTexture(n);
            if (name == "") {
                const std::vector<float> *s2 = ParamSet::Find(mat.floats, n);
                if (s2 && s2->size() >= 1) { *out = (*s2)[0]; return true; }
                return false;
            }
        }
        auto it = tex.floatTex.find(name);
        if (it != tex.floatTex.end()) { *out = it->second; return true; }
        if (tex.floatImageTex.count(name)) {
            if (errors) errors->push_back("Float image texture \"" + name + "\" on parameter \"" + n + "\": this path evaluates float image "
                                          "textures as \"alpha\" / \"shadowalpha\" masks, bump maps and the roughness of plastic / uber / substrate / metal / translucent only");
            return false;
        }
        if (errors) errors->push_back("Couldn't find float texture named \"" + name + "\" for parameter \"" + n + "\"");
        return false;
    }
    // index of the float image texture bound to parameter n, or -1
    int GetFloatImageTexture(const std::string &n) const {
        std::string name = geom.FindTexture(n);
        if (name == "" && !ParamSet::Find(geom.floats, n)) name = mat.FindTexture(n);
        if (name == "") return -1;
        auto it = tex.floatImageTex.find(name);
        return it == tex.floatImageTex.end() ? -1 : it->second;
    }
    float GetFloat(const std::string &n, float def) const {
        float f;
        return GetFloatOrNull(n, &f) ? f : def;
    }
    float FindFloat(const std::string &n, float d) const { return geom.FindOneFloat(n, mat.FindOneFloat(n, d)); }
    bool FindBool(const std::string &n, bool d) const { return geom.FindOneBool(n, mat.FindOneBool(n, d)); }
    std::string FindString(const std::string &n, const std::string &d = "") const {
        return geom.FindOneString(n, mat.FindOneString(n, d));
    }
};

}  // namespace mipt
