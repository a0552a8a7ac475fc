// spectrum.h -- host-side 31-bin SampledSpectrum (395..705 nm) used while the
// scene is being flattened. Device spectra are plain float[31]; everything here
// runs once at load time. Follows CoefficientSpectrum / SampledSpectrum in
// src/core/spectrum.h:106-293,295-500 and FromRGB in src/core/spectrum.cpp:98-180.
#pragma once
#include <cmath>
#include <algorithm>

namespace mipt {

static constexpr int kNSpec = 31;          // spectrum.h:50
static constexpr int kLambdaStart = 395;   // spectrum.h:48
static constexpr int kLambdaEnd = 705;     // spectrum.h:49
static constexpr float kCIE_Y_integral = 106.856895f;

enum class SpectrumType { Reflectance, Illuminant };

struct Spectrum {
    float c[kNSpec];
    Spectrum(float v = 0.f) { for (int i = 0; i < kNSpec; ++i) c[i] = v; }
    static Spectrum FromArray(const float *v) { Spectrum s; for (int i = 0; i < kNSpec; ++i) s.c[i] = v[i]; return s; }
    Spectrum &operator+=(const Spectrum &s) { for (int i = 0; i < kNSpec; ++i) c[i] += s.c[i]; return *this; }
    Spectrum operator+(const Spectrum &s) const { Spectrum r = *this; r += s; return r; }
    Spectrum operator-(const Spectrum &s) const { Spectrum r = *this; for (int i = 0; i < kNSpec; ++i) r.c[i] -= s.c[i]; return r; }
    Spectrum operator*(const Spectrum &s) const { Spectrum r = *this; for (int i = 0; i < kNSpec; ++i) r.c[i] *= s.c[i]; return r; }
    Spectrum operator*(float a) const { Spectrum r = *this; for (int i = 0; i < kNSpec; ++i) r.c[i] *= a; return r; }
    Spectrum &operator*=(float a) { for (int i = 0; i < kNSpec; ++i) c[i] *= a; return *this; }
    Spectrum operator/(float a) const { Spectrum r = *this; for (int i = 0; i < kNSpec; ++i) r.c[i] /= a; return r; }
    Spectrum operator-() const { Spectrum r; for (int i = 0; i < kNSpec; ++i) r.c[i] = -c[i]; return r; }
    bool IsBlack() const { for (int i = 0; i < kNSpec; ++i) if (c[i] != 0.) return false; return true; }
    Spectrum Clamp(float low = 0, float high = INFINITY) const {
        Spectrum r;
        for (int i = 0; i < kNSpec; ++i) r.c[i] = c[i] < low ? low : (c[i] > high ? high : c[i]);
        return r;
    }
    void ToXYZ(float xyz[3]) const;  // spectrum.h:402-414
    float y() const;  // spectrum.h:415-421 (fork clamps negative sums to 0)
    static Spectrum FromRGB(const float rgb[3], SpectrumType type = SpectrumType::Illuminant);
    static Spectrum FromXYZ(const float xyz[3], SpectrumType type = SpectrumType::Reflectance);
    // "spectrum" parameters given as (lambda, value) pairs, spectrum.h:302-321
    static Spectrum FromSampled(const float *lambda, const float *v, int n);
    static const float *CIE_Y();  // the 31-bin Y matching function
    static const float *RGBIllumBasis(int k);  // rgbIllum2Spect White, Cyan, Magenta, Yellow, Red, Green, Blue
};
inline Spectrum operator*(float a, const Spectrum &s) { return s * a; }
inline Spectrum Sqrt(const Spectrum &s) { Spectrum r; for (int i = 0; i < kNSpec; ++i) r.c[i] = std::sqrt(s.c[i]); return r; }
inline Spectrum Lerp(float t, const Spectrum &s1, const Spectrum &s2) {  // spectrum.h:536-538
    return (1 - t) * s1 + t * s2;
}

}  // namespace mipt
