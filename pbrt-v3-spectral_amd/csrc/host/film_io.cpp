// film_io.cpp -- spectral film file, Film::WriteImage's spectralFlag branch
// (src/core/film.cpp:226-308): text header "<w> <h> 31\nv3 \n" then 31 planes
// (wavelength-major) of w*h float64 values, un-normalised sum(L*weight)*scale.
#include <cstdio>
#include <cstring>
#include "scene.h"

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

}  // namespace mipt
