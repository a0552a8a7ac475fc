// image.h -- image files and MIPMap<RGBSpectrum> on the host (src/core/imageio.cpp, src/core/mipmap.h).
#pragma once
#include <cmath>
#include <string>
#include <vector>

namespace mipt {

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

enum class ImageWrap { Repeat = 0, Black = 1, Clamp = 2 };  // mipmap.h:50

// ReadImage (imageio.cpp:60-79): .pfm, .tga (uncompressed / RLE, 8/24/32 bit, colour-mapped) and .png (via zlib);
// top row first. EXR is not read by this build. On failure returns false with *err set.
bool ReadImage(const std::string &filename, int *xres, int *yres, std::vector<RGB> *texels, std::string *err);

// WriteImageEXR (imageio.cpp:163-189): RGB as HALF channels B, G, R of a scan-line OpenEXR file (ZIP compression; the
// reference's RgbaOutputFile defaults to PIZ, any reader takes either).
bool WriteEXR(const std::string &filename, int w, int h, const float *rgb, std::string *err);

// MIPMap<RGBSpectrum>: power-of-two Lanczos resampling + box-filtered pyramid (mipmap.h:118-211).
struct MIPMap {
    struct Level { int w, h; std::vector<RGB> t; };
    std::vector<Level> pyramid;
    ImageWrap wrap;
    MIPMap(int rx, int ry, const std::vector<RGB> &img, ImageWrap wrapMode = ImageWrap::Repeat);
    int Levels() const { return (int)pyramid.size(); }
    int Width() const { return pyramid[0].w; }
    int Height() const { return pyramid[0].h; }
    RGB Texel(int level, int s, int t) const;           // mipmap.h:213-235
    RGB triangle(int level, const float st[2]) const;   // mipmap.h:268-279
    RGB Lookup(const float st[2], float width) const;   // trilinear, mipmap.h:238-266
};

}  // namespace mipt
