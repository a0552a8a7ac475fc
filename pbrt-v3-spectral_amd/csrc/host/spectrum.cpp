// spectrum.cpp -- RGB -> 31-bin SPD conversion (Smits-style basis) and y().
// Basis data: spectral_basis_31.inc (derived, see tools/make_spectral_basis.py).
// Branch structure follows SampledSpectrum::FromRGB, src/core/spectrum.cpp:98-180.
#include "spectrum.h"
#include <utility>
#include <vector>

namespace mipt {
namespace {
#include "spectral_basis_31.inc"
inline Spectrum S(const float *v) { return Spectrum::FromArray(v); }
inline void XYZToRGB(const float xyz[3], float rgb[3]) {  // spectrum.h:56-60
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
}  // namespace

const float *Spectrum::CIE_Y() { return kCIE_Y; }
const float *Spectrum::RGBIllumBasis(int k) {
    static const float *const b[7] = {kRGBIllum2SpectWhite, kRGBIllum2SpectCyan, kRGBIllum2SpectMagenta, kRGBIllum2SpectYellow,
                                      kRGBIllum2SpectRed, kRGBIllum2SpectGreen, kRGBIllum2SpectBlue};
    return b[k];
}

void Spectrum::ToXYZ(float xyz[3]) const {
    xyz[0] = xyz[1] = xyz[2] = 0.f;
    for (int i = 0; i < kNSpec; ++i) {
        xyz[0] += kCIE_X[i] * c[i];
        xyz[1] += kCIE_Y[i] * c[i];
        xyz[2] += kCIE_Z[i] * c[i];
    }
    float scale = float(kLambdaEnd - kLambdaStart) / float(kCIE_Y_integral * kNSpec);
    xyz[0] *= scale;
    xyz[1] *= scale;
    xyz[2] *= scale;
}

float Spectrum::y() const {
    float yy = 0.f;
    for (int i = 0; i < kNSpec; ++i) yy += kCIE_Y[i] * c[i];
    yy = (yy < 0) ? 0 : yy;
    return yy * float(kLambdaEnd - kLambdaStart) / float(kCIE_Y_integral * kNSpec);
}

Spectrum Spectrum::FromRGB(const float rgb[3], SpectrumType type) {
    Spectrum r;
    const bool refl = (type == SpectrumType::Reflectance);
    const Spectrum white = S(refl ? kRGBRefl2SpectWhite : kRGBIllum2SpectWhite);
    const Spectrum cyan = S(refl ? kRGBRefl2SpectCyan : kRGBIllum2SpectCyan);
    const Spectrum magenta = S(refl ? kRGBRefl2SpectMagenta : kRGBIllum2SpectMagenta);
    const Spectrum yellow = S(refl ? kRGBRefl2SpectYellow : kRGBIllum2SpectYellow);
    const Spectrum red = S(refl ? kRGBRefl2SpectRed : kRGBIllum2SpectRed);
    const Spectrum green = S(refl ? kRGBRefl2SpectGreen : kRGBIllum2SpectGreen);
    const Spectrum blue = S(refl ? kRGBRefl2SpectBlue : kRGBIllum2SpectBlue);
    if (rgb[0] <= rgb[1] && rgb[0] <= rgb[2]) {
        r += rgb[0] * white;
        if (rgb[1] <= rgb[2]) {
            r += (rgb[1] - rgb[0]) * cyan;
            r += (rgb[2] - rgb[1]) * blue;
        } else {
            r += (rgb[2] - rgb[0]) * cyan;
            r += (rgb[1] - rgb[2]) * green;
        }
    } else if (rgb[1] <= rgb[0] && rgb[1] <= rgb[2]) {
        r += rgb[1] * white;
        if (rgb[0] <= rgb[2]) {
            r += (rgb[0] - rgb[1]) * magenta;
            r += (rgb[2] - rgb[0]) * blue;
        } else {
            r += (rgb[2] - rgb[1]) * magenta;
            r += (rgb[0] - rgb[2]) * red;
        }
    } else {
        r += rgb[2] * white;
        if (rgb[0] <= rgb[1]) {
            r += (rgb[0] - rgb[2]) * yellow;
            r += (rgb[1] - rgb[0]) * green;
        } else {
            r += (rgb[1] - rgb[2]) * yellow;
            r += (rgb[0] - rgb[1]) * red;
        }
    }
    // "r *= .94" converts the double literal to Float first (operator*=(Float))
    r *= refl ? (float).94 : .86445f;
    return r.Clamp();
}

// AverageSpectrumSamples, src/core/spectrum.cpp:59-90 (note the double-precision
// accumulation of each trapezoid through the 0.5 literal).
static float AverageSpectrumSamples(const float *lambda, const float *vals, int n, float lambdaStart, float lambdaEnd) {
    if (lambdaEnd <= lambda[0]) return vals[0];
    if (lambdaStart >= lambda[n - 1]) return vals[n - 1];
    if (n == 1) return vals[0];
    float sum = 0;
    if (lambdaStart < lambda[0]) sum += vals[0] * (lambda[0] - lambdaStart);
    if (lambdaEnd > lambda[n - 1]) sum += vals[n - 1] * (lambdaEnd - lambda[n - 1]);
    int i = 0;
    while (lambdaStart > lambda[i + 1]) ++i;
    auto lerp = [](float t, float v1, float v2) { return (1 - t) * v1 + t * v2; };
    auto interp = [&](float w, int i) { return lerp((w - lambda[i]) / (lambda[i + 1] - lambda[i]), vals[i], vals[i + 1]); };
    for (; i + 1 < n && lambdaEnd >= lambda[i]; ++i) {
        float segLambdaStart = std::max(lambdaStart, lambda[i]);
        float segLambdaEnd = std::min(lambdaEnd, lambda[i + 1]);
        sum += 0.5 * (interp(segLambdaStart, i) + interp(segLambdaEnd, i)) * (segLambdaEnd - segLambdaStart);
    }
    return sum / (lambdaEnd - lambdaStart);
}

Spectrum Spectrum::FromSampled(const float *lambdaIn, const float *vIn, int n) {
    std::vector<std::pair<float, float>> sv;
    for (int i = 0; i < n; ++i) sv.push_back(std::make_pair(lambdaIn[i], vIn[i]));
    bool sorted = true;
    for (int i = 0; i + 1 < n; ++i) if (lambdaIn[i] > lambdaIn[i + 1]) sorted = false;
    if (!sorted) std::sort(sv.begin(), sv.end());
    std::vector<float> lambda(n), v(n);
    for (int i = 0; i < n; ++i) { lambda[i] = sv[i].first; v[i] = sv[i].second; }
    Spectrum r;
    auto lerp = [](float t, float v1, float v2) { return (1 - t) * v1 + t * v2; };
    for (int i = 0; i < kNSpec; ++i) {
        float lambda0 = lerp(float(i) / float(kNSpec), (float)kLambdaStart, (float)kLambdaEnd);
        float lambda1 = lerp(float(i + 1) / float(kNSpec), (float)kLambdaStart, (float)kLambdaEnd);
        r.c[i] = AverageSpectrumSamples(lambda.data(), v.data(), n, lambda0, lambda1);
    }
    return r;
}

Spectrum Spectrum::FromXYZ(const float xyz[3], SpectrumType type) {
    float rgb[3];
    XYZToRGB(xyz, rgb);
    return FromRGB(rgb, type);
}

}  // namespace mipt
