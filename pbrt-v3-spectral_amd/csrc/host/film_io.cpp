// film_io.cpp -- spectral film file, Film::WriteImage's spectralFlag branch
// (src/core/film.cpp:226-308): text header "<w> <h> 31\nv3 \n" then 31 planes
// (wavelength-major) of w*h float64 values, un-normalised sum(L*weight)*scale.
#include <cmath>
#include <cstdio>
#include <cstring>
#include "scene.h"
#include "image.h"

namespace mipt {

bool WriteSpectralDat(const std::string &filename, int w, int h, const float *filmSum, float scale,
                      std::string *err) {
    size_t ext = filename.find_last_of('.');
    std::string dat = filename.substr(0, ext) + ".dat";
    FILE *f = fopen(dat.c_str(), "wb");
    if (!f) { *err = "cannot open " + dat; return false; }
    fprintf(f, "%d %d %d\n", w, h, kNSpec);
    fprintf(f, "v3 \n");
    std::vector<double> plane((size_t)w * h);
    for (int c = 0; c < kNSpec; ++c) {
        for (size_t j = 0; j < (size_t)w * h; ++j) {
            float v = filmSum[j * kNSpec + c];
            v += 1.f * 0.f;  // splatScale * splatSpectrum (no splats on this path)
            v *= scale;
            plane[j] = (double)v;
        }
        if (fwrite(plane.data(), sizeof(double), plane.size(), f) != plane.size()) {
            fclose(f);
            *err = "short write to " + dat;
            return false;
        }
    }
    fclose(f);
    return true;
}

// Film::WriteImage, RGB branch (film.cpp:182-225) followed by pbrt::WriteImage (imageio.cpp:81-119,
// 437-482): per pixel XYZ -> RGB, division by the filter-weight sum, clamp at 0, scale; then the
// file format picked by the extension. Pixel::xyz is the sum over film tiles of
// contribSum.ToXYZ() (film.cpp:133-137); ToXYZ is linear, so it is taken here from the summed
// spectrum. PFM (float RGB, rows bottom to top) and TGA (8 bit, gamma) are written natively; EXR and
// PNG need libraries this build does not link, so those names get a .pfm beside them.
bool WriteRGBImage(const std::string &filename, int w, int h, const float *filmSum, const float *weightSum,
                   float scale, std::string *written, std::string *err) {
    std::vector<float> rgb((size_t)w * h * 3);
    for (size_t j = 0; j < (size_t)w * h; ++j) {
        Spectrum L;
        for (int c = 0; c < kNSpec; ++c) L.c[c] = filmSum[j * kNSpec + c];
        float xyz[3];
        L.ToXYZ(xyz);
        float *o = &rgb[3 * j];
        o[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];   // XYZToRGB, spectrum.h:56-60
        o[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
        o[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
        const float filterWeightSum = weightSum[j];
        if (filterWeightSum != 0) {
            const float invWt = (float)1 / filterWeightSum;
            for (int k = 0; k < 3; ++k) o[k] = std::max((float)0, o[k] * invWt);
        }
        for (int k = 0; k < 3; ++k) { o[k] += 1.f * 0.f; o[k] *= scale; }   // no splats on this path
    }
    auto hasExt = [&](const char *e) {
        const size_t n = strlen(e);
        if (filename.size() < n) return false;
        for (size_t i = 0; i < n; ++i) if (tolower(filename[filename.size() - n + i]) != e[i]) return false;
        return true;
    };
    std::string out = filename;
    const bool tga = hasExt(".tga");
    if (hasExt(".exr")) {   // WriteImageEXR, imageio.cpp:163-189
        if (written) *written = out;
        return WriteEXR(out, w, h, rgb.data(), err);
    }
    if (!tga && !hasExt(".pfm")) out = filename.substr(0, filename.find_last_of('.')) + ".pfm";
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) { *err = "Unable to open output file \"" + out + "\""; return false; }
    bool ok = true;
    if (tga) {  // 24-bit uncompressed, bottom-left origin, BGR, gamma-corrected (imageio.cpp:90-117, pbrt.h:435-438)
        unsigned char hd[18] = {0};
        hd[2] = 2; hd[12] = w & 0xff; hd[13] = (w >> 8) & 0xff; hd[14] = h & 0xff; hd[15] = (h >> 8) & 0xff; hd[16] = 24;
        ok = fwrite(hd, 1, 18, f) == 18;
        auto toByte = [](float v) {
            float g = (v <= 0.0031308f) ? 12.92f * v : 1.055f * std::pow(v, (float)(1.f / 2.4f)) - 0.055f;
            float b = 255.f * g + 0.5f;
            return (unsigned char)(b < 0.f ? 0.f : (b > 255.f ? 255.f : b));
        };
        std::vector<unsigned char> row((size_t)w * 3);
        for (int y = h - 1; y >= 0 && ok; --y) {
            for (int x = 0; x < w; ++x) {
                const float *p = &rgb[3 * ((size_t)y * w + x)];
                row[3 * x] = toByte(p[2]); row[3 * x + 1] = toByte(p[1]); row[3 * x + 2] = toByte(p[0]);
            }
            ok = fwrite(row.data(), 1, row.size(), f) == row.size();
        }
    } else {
        ok = fprintf(f, "PF\n%d %d\n%f\n", w, h, -1.f) > 0;
        for (int y = h - 1; y >= 0 && ok; --y)
            ok = fwrite(&rgb[3 * (size_t)y * w], sizeof(float), (size_t)w * 3, f) == (size_t)w * 3;
    }
    fclose(f);
    if (!ok) { *err = "Error writing image file \"" + out + "\""; return false; }
    if (written) *written = out;
    return true;
}

}  // namespace mipt
