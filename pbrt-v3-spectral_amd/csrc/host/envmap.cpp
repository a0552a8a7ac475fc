// envmap.cpp -- LightSource "infinite": the InfiniteAreaLight constructor (src/lights/infinite.cpp:43-83)
// on the host. Produces level 0 of the light's MIPMap<RGBSpectrum> (after the reference's power-of-two
// Lanczos resampling, mipmap.h:118-196), the Distribution2D over the 2W x 2H luminance image that the
// constructor filters out of the map with Lookup(st, fwidth) (trilinear between pyramid levels,
// mipmap.h:252-281), and the value Power() is built from (infinite.cpp:85-89).
//   image reading     image.cpp (PFM, TGA, PNG; an EXR map is reported as an error and the light becomes constant)
//   Distribution1D/2D src/core/sampling.h:55-109,123-147, sampling.cpp:41-56
#include <cmath>
#include <cstdio>
#include <cstring>
#include "scene.h"
#include "image.h"

namespace mipt {
namespace {

void MakeDistribution1D(const float *f, int n, float *func, float *cdf, float *funcInt) {  // sampling.h:57-70
    for (int i = 0; i < n; ++i) func[i] = f[i];
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n); }
    else { for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt; }
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
        std::string ioErr;
        if (!ReadImage(texmap, &rx, &ry, &texels, &ioErr)) { errors->push_back(ioErr); texels.clear(); }
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

int ConstantFloatMipMap(HostScene *scene, float value) {
    char keyBuf[64];
    snprintf(keyBuf, sizeof keyBuf, "<constant>|%a", value);
    for (size_t i = 0; i < scene->mipStore.size(); ++i) if (scene->mipStore[i].key == keyBuf) return (int)i;
    HostMipMap h;
    h.key = keyBuf;
    h.width = h.height = 1; h.wrap = 0;
    h.levelOffset.push_back(0);
    h.texels.assign(3, value);
    scene->mipStore.push_back(std::move(h));
    return (int)scene->mipStore.size() - 1;
}

int BuildTextureMipMap(HostScene *scene, const std::string &filename, bool trilinear, bool noFiltering, float maxAniso,
                       int wrap, float scale, bool gamma, bool isFloat) {
    char keyBuf[128];
    snprintf(keyBuf, sizeof keyBuf, "|%d|%d|%a|%d|%a|%d|%d", (int)trilinear, (int)noFiltering, maxAniso, wrap, scale, (int)gamma, (int)isFloat);
    const std::string key = filename + keyBuf;
    for (size_t i = 0; i < scene->mipStore.size(); ++i) if (scene->mipStore[i].key == key) return (int)i;
    int rx = 0, ry = 0;
    std::vector<RGB> texels;
    std::string ioErr;
    if (!ReadImage(filename, &rx, &ry, &texels, &ioErr)) {
        scene->errors.push_back(ioErr);
        scene->warnings.push_back("Creating a constant grey texture to replace \"" + filename + "\".");
        rx = ry = 1;
        texels.assign(1, RGB(0.5f));
    }
    // flip in y: texture space has (0,0) at the lower left corner (imagemap.cpp:79-86)
    for (int y = 0; y < ry / 2; ++y)
        for (int x = 0; x < rx; ++x) std::swap(texels[(size_t)y * rx + x], texels[(size_t)(ry - 1 - y) * rx + x]);
    auto inverseGamma = [](float value) {  // pbrt.h:301-304
        if (value <= 0.04045f) return value * 1.f / 12.92f;
        return std::pow((value + 0.055f) * 1.f / 1.055f, (float)2.4f);
    };
    if (isFloat)   // convertIn to Float: scale * (gamma ? InverseGammaCorrect(y) : y), imagemap.h:107-110; kept grey
        for (RGB &t : texels) { const float y = t.y(); t = RGB(scale * (gamma ? inverseGamma(y) : y)); }
    else
        for (RGB &t : texels) for (int k = 0; k < 3; ++k) t.c[k] = scale * (gamma ? inverseGamma(t.c[k]) : t.c[k]);  // convertIn
    MIPMap mip(rx, ry, texels, (ImageWrap)wrap);
    HostMipMap h;
    h.key = key;
    h.width = mip.Width(); h.height = mip.Height(); h.wrap = wrap;
    for (const MIPMap::Level &l : mip.pyramid) {
        h.levelOffset.push_back((uint32_t)(h.texels.size() / 3));
        for (const RGB &t : l.t) for (int k = 0; k < 3; ++k) h.texels.push_back(t.c[k]);
    }
    scene->mipStore.push_back(std::move(h));
    return (int)scene->mipStore.size() - 1;
}

}  // namespace mipt
