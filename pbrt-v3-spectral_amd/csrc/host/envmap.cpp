// envmap.cpp -- LightSource "infinite": the InfiniteAreaLight constructor (src/lights/infinite.cpp:43-83)
// on the host. Produces level 0 of the light's MIPMap<RGBSpectrum> (after the reference's power-of-two
// Lanczos resampling, mipmap.h:118-196), the Distribution2D over the 2W x 2H luminance image that the
// constructor filters out of the map with Lookup(st, fwidth) (trilinear between pyramid levels,
// mipmap.h:252-281), and the value Power() is built from (infinite.cpp:85-89).
//   image reading     ReadImagePFM, src/core/imageio.cpp:349-435 (EXR / PNG / TGA need libraries or code this
//                     build does not carry; such a map is reported as an error and the light becomes constant)
//   Distribution1D/2D src/core/sampling.h:55-109,123-147, sampling.cpp:41-56
#include <cmath>
#include <cstdio>
#include <cstring>
#include "scene.h"

namespace mipt {
namespace {

struct RGB {
    float c[3];
    RGB(float v = 0.f) { c[0] = c[1] = c[2] = v; }
    RGB operator+(const RGB &o) const { RGB r; for (int i = 0; i < 3; ++i) r.c[i] = c[i] + o.c[i]; return r; }
    RGB &operator+=(const RGB &o) { for (int i = 0; i < 3; ++i) c[i] += o.c[i]; return *this; }
    RGB operator*(float a) const { RGB r; for (int i = 0; i < 3; ++i) r.c[i] = c[i] * a; return r; }
    RGB operator*(const RGB &o) const { RGB r; for (int i = 0; i < 3; ++i) r.c[i] = c[i] * o.c[i]; return r; }
    RGB Clamp() const { RGB r; for (int i = 0; i < 3; ++i) r.c[i] = std::min(std::max(c[i], 0.f), INFINITY); return r; }
    float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }  // spectrum.h:535-538
};
inline RGB operator*(float a, const RGB &s) { return s * a; }

inline int Mod(int a, int b) { int r = a - (a / b) * b; return (r < 0) ? r + b : r; }
inline bool IsPowerOf2(int v) { return v && !(v & (v - 1)); }
inline int RoundUpPow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline int Log2Int(uint32_t v) { return 31 - __builtin_clz(v); }
inline float Log2(float x) { const float invLog2 = 1.442695040888963387004650940071; return std::log(x) * invLog2; }
inline float Lanczos(float x, float tau = 2) {  // texture.cpp:254-262
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= kPi;
    float s = std::sin(x * tau) / (x * tau);
    float lanczos = std::sin(x) / x;
    return s * lanczos;
}

struct MIPMap {  // MIPMap<RGBSpectrum>, wrap mode Repeat, mipmap.h
    struct Level { int w, h; std::vector<RGB> t; };
    std::vector<Level> pyramid;
    int Levels() const { return (int)pyramid.size(); }
    int Width() const { return pyramid[0].w; }
    int Height() const { return pyramid[0].h; }
    const RGB &Texel(int level, int s, int t) const {
        const Level &l = pyramid[level];
        s = Mod(s, l.w);
        t = Mod(t, l.h);
        return l.t[(size_t)t * l.w + s];
    }
    RGB triangle(int level, const float st[2]) const {
        level = std::min(std::max(level, 0), Levels() - 1);
        float s = st[0] * pyramid[level].w - 0.5f;
        float t = st[1] * pyramid[level].h - 0.5f;
        int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        float ds = s - s0, dt = t - t0;
        return (1 - ds) * (1 - dt) * Texel(level, s0, t0) + (1 - ds) * dt * Texel(level, s0, t0 + 1) +
               ds * (1 - dt) * Texel(level, s0 + 1, t0) + ds * dt * Texel(level, s0 + 1, t0 + 1);
    }
    RGB Lookup(const float st[2], float width) const {
        float level = Levels() - 1 + Log2(std::max(width, (float)1e-8));
        if (level < 0) return triangle(0, st);
        else if (level >= Levels() - 1) return Texel(Levels() - 1, 0, 0);
        int iLevel = (int)std::floor(level);
        float delta = level - iLevel;
        return (1 - delta) * triangle(iLevel, st) + delta * triangle(iLevel + 1, st);  // Lerp, pbrt.h:420
    }
    MIPMap(int rx, int ry, const std::vector<RGB> &img) {
        std::vector<RGB> base = img;
        if (!IsPowerOf2(rx) || !IsPowerOf2(ry)) {
            const int px = RoundUpPow2(rx), py = RoundUpPow2(ry);
            struct W { int first; float w[4]; };
            auto weights = [](int oldRes, int newRes) {
                std::vector<W> wt(newRes);
                const float filterwidth = 2.f;
                for (int i = 0; i < newRes; ++i) {
                    float center = (i + .5f) * oldRes / newRes;
                    wt[i].first = (int)std::floor((center - filterwidth) + 0.5f);
                    for (int j = 0; j < 4; ++j) {
                        float pos = wt[i].first + j + .5f;
                        wt[i].w[j] = Lanczos((pos - center) / filterwidth);
                    }
                    float invSumWts = 1 / (wt[i].w[0] + wt[i].w[1] + wt[i].w[2] + wt[i].w[3]);
                    for (int j = 0; j < 4; ++j) wt[i].w[j] *= invSumWts;
                }
                return wt;
            };
            std::vector<RGB> res((size_t)px * py);
            std::vector<W> sW = weights(rx, px);
            for (int t = 0; t < ry; ++t)
                for (int s = 0; s < px; ++s) {
                    RGB &o = res[(size_t)t * px + s];
                    o = RGB(0.f);
                    for (int j = 0; j < 4; ++j) {
                        int origS = Mod(sW[s].first + j, rx);
                        if (origS >= 0 && origS < rx) o += sW[s].w[j] * img[(size_t)t * rx + origS];
                    }
                }
            std::vector<W> tW = weights(ry, py);
            std::vector<RGB> work(py);
            for (int s = 0; s < px; ++s) {
                for (int t = 0; t < py; ++t) {
                    work[t] = RGB(0.f);
                    for (int j = 0; j < 4; ++j) {
                        int offset = Mod(tW[t].first + j, ry);
                        if (offset >= 0 && offset < ry) work[t] += tW[t].w[j] * res[(size_t)offset * px + s];
                    }
                }
                for (int t = 0; t < py; ++t) res[(size_t)t * px + s] = work[t].Clamp();
            }
            base.swap(res);
            rx = px; ry = py;
        }
        int nLevels = 1 + Log2Int((uint32_t)std::max(rx, ry));
        pyramid.resize(nLevels);
        pyramid[0] = Level{rx, ry, base};
        for (int i = 1; i < nLevels; ++i) {
            int sRes = std::max(1, pyramid[i - 1].w / 2), tRes = std::max(1, pyramid[i - 1].h / 2);
            pyramid[i].w = sRes; pyramid[i].h = tRes;
            pyramid[i].t.resize((size_t)sRes * tRes);
            for (int t = 0; t < tRes; ++t)
                for (int s = 0; s < sRes; ++s)
                    pyramid[i].t[(size_t)t * sRes + s] =
                        .25f * (Texel(i - 1, 2 * s, 2 * t) + Texel(i - 1, 2 * s + 1, 2 * t) + Texel(i - 1, 2 * s, 2 * t + 1) +
                                Texel(i - 1, 2 * s + 1, 2 * t + 1));
        }
    }
};

void MakeDistribution1D(const float *f, int n, float *func, float *cdf, float *funcInt) {  // sampling.h:57-70
    for (int i = 0; i < n; ++i) func[i] = f[i];
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n); }
    else { for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt; }
}

bool ReadPFM(const std::string &filename, int *xres, int *yres, std::vector<RGB> *out) {  // imageio.cpp:349-435
    FILE *fp = fopen(filename.c_str(), "rb");
    if (!fp) return false;
    auto readWord = [&](char *buf, int len) {
        int n = 0, c;
        while ((c = fgetc(fp)) != EOF && !isspace(c) && n < len - 1) buf[n++] = (char)c;
        buf[n] = 0;
        return (c == EOF && n == 0) ? -1 : n;
    };
    char buf[80];
    int nChannels = 0;
    bool ok = readWord(buf, 80) != -1;
    if (ok) { if (!strcmp(buf, "Pf")) nChannels = 1; else if (!strcmp(buf, "PF")) nChannels = 3; else ok = false; }
    int width = 0, height = 0;
    float scale = 1;
    if (ok && readWord(buf, 80) != -1) width = atoi(buf); else ok = false;
    if (ok && readWord(buf, 80) != -1) height = atoi(buf); else ok = false;
    if (ok && readWord(buf, 80) != -1) sscanf(buf, "%f", &scale); else ok = false;
    if (!ok || width <= 0 || height <= 0) { fclose(fp); return false; }
    std::vector<float> data((size_t)nChannels * width * height);
    for (int y = height - 1; y >= 0 && ok; --y)   // P*M has its origin at the lower left
        ok = fread(&data[(size_t)y * nChannels * width], sizeof(float), (size_t)nChannels * width, fp) == (size_t)nChannels * width;
    fclose(fp);
    if (!ok) return false;
    if (!(scale < 0.f))   // big-endian file on this little-endian host
        for (float &v : data) { unsigned char b[4]; memcpy(b, &v, 4); std::swap(b[0], b[3]); std::swap(b[1], b[2]); memcpy(&v, b, 4); }
    if (std::abs(scale) != 1.f) for (float &v : data) v *= std::abs(scale);
    out->resize((size_t)width * height);
    for (size_t i = 0; i < out->size(); ++i) {
        if (nChannels == 1) (*out)[i] = RGB(data[i]);
        else { (*out)[i].c[0] = data[3 * i]; (*out)[i].c[1] = data[3 * i + 1]; (*out)[i].c[2] = data[3 * i + 2]; }
    }
    *xres = width; *yres = height;
    return true;
}

}  // namespace

// L = the light's "L" * "scale"; texmap may be empty. Fills *env (storage in *store) and *centre =
// Spectrum(Lmap->Lookup((.5, .5), .5f), Illuminant) -- Power() / (pi * worldRadius^2).
bool BuildEnvMap(const Spectrum &L, const std::string &texmap, HostEnvMap *store, Spectrum *centre,
                 std::vector<std::string> *errors) {
    float lrgb[3];
    {   // L.ToRGBSpectrum(): spectrum.h:422-426
        float xyz[3];
        L.ToXYZ(xyz);
        lrgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
        lrgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
        lrgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
    }
    RGB Lrgb;
    for (int i = 0; i < 3; ++i) Lrgb.c[i] = lrgb[i];
    int rx = 0, ry = 0;
    std::vector<RGB> texels;
    if (!texmap.empty()) {
        const size_t dot = texmap.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : texmap.substr(dot);
        for (char &c : ext) c = (char)tolower(c);
        if (ext != ".pfm")
            errors->push_back("Unable to load image stored in format \"" + (ext.empty() ? std::string("(unknown)") : ext.substr(1)) +
                              "\" for filename \"" + texmap + "\" (this build reads PFM environment maps only).");
        else if (!ReadPFM(texmap, &rx, &ry, &texels)) errors->push_back("Error reading PFM file \"" + texmap + "\"");
        if (!texels.empty()) for (RGB &t : texels) t = t * Lrgb;
    }
    if (texels.empty()) { rx = ry = 1; texels.assign(1, Lrgb); }
    MIPMap mip(rx, ry, texels);
    HostEnvMap &e = *store;
    e.width = mip.Width(); e.height = mip.Height();
    e.rgb.resize((size_t)e.width * e.height * 3);
    for (size_t i = 0; i < (size_t)e.width * e.height; ++i) for (int k = 0; k < 3; ++k) e.rgb[3 * i + k] = mip.pyramid[0].t[i].c[k];
    // sampling distribution, infinite.cpp:64-83
    const int width = 2 * mip.Width(), height = 2 * mip.Height();
    std::vector<float> img((size_t)width * height);
    const float fwidth = 0.5f / std::min(width, height);
    for (int v = 0; v < height; ++v) {
        float vp = (v + .5f) / (float)height;
        float sinTheta = std::sin(kPi * (v + .5f) / height);
        for (int u = 0; u < width; ++u) {
            float up = (u + .5f) / (float)width;
            float st[2] = {up, vp};
            img[u + (size_t)v * width] = mip.Lookup(st, fwidth).y();
            img[u + (size_t)v * width] *= sinTheta;
        }
    }
    e.nu = width; e.nv = height;
    e.condFunc.resize((size_t)width * height);
    e.condCdf.resize((size_t)(width + 1) * height);
    e.condFuncInt.resize(height);
    for (int v = 0; v < height; ++v)
        MakeDistribution1D(&img[(size_t)v * width], width, &e.condFunc[(size_t)v * width], &e.condCdf[(size_t)v * (width + 1)], &e.condFuncInt[v]);
    e.margFunc.resize(height);
    e.margCdf.resize(height + 1);
    MakeDistribution1D(e.condFuncInt.data(), height, e.margFunc.data(), e.margCdf.data(), &e.margFuncInt);
    const float half[2] = {.5f, .5f};
    const RGB c = mip.Lookup(half, .5f);
    *centre = Spectrum::FromRGB(c.c, SpectrumType::Illuminant);
    return true;
}

}  // namespace mipt
