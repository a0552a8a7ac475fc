// image.cpp -- ReadImage for .pfm / .tga / .png / .exr and MIPMap<RGBSpectrum> construction.
//   ReadImagePFM   src/core/imageio.cpp:349-435
//   ReadImageTGA   src/core/imageio.cpp:216-255 (pixels / 255, BGR order; the reference decodes through ext/targa,
//                  here the TGA 2.0 layout is read directly: types 1, 2, 3 and their RLE forms 9, 10, 11)
//   ReadImagePNG   src/core/imageio.cpp:258-287 (lodepng_decode24 in the reference: 8-bit RGB, alpha dropped, 16-bit
//                  samples reduced to their high byte, palette expanded; here inflate comes from zlib)
//   MIPMap         src/core/mipmap.h:118-279
#include <cstdio>
#include <cstring>
#include <zlib.h>
#include "image.h"
#include "ptmath.h"

namespace mipt {
namespace {

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

bool ReadFile(const std::string &filename, std::vector<unsigned char> *out) {
    FILE *fp = fopen(filename.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    bool ok = n >= 0 && fread(out->data(), 1, out->size(), fp) == out->size();
    fclose(fp);
    return ok;
}

bool ReadPFM(const std::string &filename, int *xres, int *yres, std::vector<RGB> *out) {
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
    {   // the header's size against the bytes that follow it: a corrupt width / height is an error, not an allocation
        const long here = ftell(fp);
        fseek(fp, 0, SEEK_END);
        const long long left = (long long)ftell(fp) - here;
        fseek(fp, here, SEEK_SET);
        if ((long long)nChannels * width * height * 4 > left) { fclose(fp); return false; }
    }
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

bool ReadTGA(const std::string &filename, int *xres, int *yres, std::vector<RGB> *out, std::string *err) {
    std::vector<unsigned char> f;
    if (!ReadFile(filename, &f) || f.size() < 18) { *err = "Unable to read from TGA file \"" + filename + "\""; return false; }
    const int idLen = f[0], cmapType = f[1], type = f[2];
    const int cmapFirst = f[3] | (f[4] << 8), cmapLen = f[5] | (f[6] << 8), cmapBits = f[7];
    const int w = f[12] | (f[13] << 8), h = f[14] | (f[15] << 8), bpp = f[16], desc = f[17];
    const bool rle = type >= 9;
    const int base = rle ? type - 8 : type;
    if (w <= 0 || h <= 0 || (base != 1 && base != 2 && base != 3) || (bpp != 8 && bpp != 16 && bpp != 24 && bpp != 32)) {
        *err = "Unable to read from TGA file \"" + filename + "\" (unsupported image type)";
        return false;
    }
    size_t pos = 18 + (size_t)idLen;
    const int cmapBytes = (cmapBits + 7) / 8;
    const unsigned char *cmap = nullptr;
    if (cmapType == 1) { cmap = f.data() + pos; pos += (size_t)cmapLen * cmapBytes; }
    if (base == 1 && (!cmap || (cmapBytes != 3 && cmapBytes != 4))) { *err = "Unable to read from TGA file \"" + filename + "\" (colour map)"; return false; }
    const int pb = bpp / 8;
    std::vector<unsigned char> pix((size_t)w * h * pb);
    if (!rle) {
        if (pos + pix.size() > f.size()) { *err = "Unable to read from TGA file \"" + filename + "\" (truncated)"; return false; }
        memcpy(pix.data(), f.data() + pos, pix.size());
    } else {
        size_t o = 0;
        while (o < pix.size()) {
            if (pos >= f.size()) { *err = "Unable to read from TGA file \"" + filename + "\" (truncated)"; return false; }
            const int hd = f[pos++], n = (hd & 0x7f) + 1;
            if (hd & 0x80) {
                if (pos + pb > f.size()) { *err = "Unable to read from TGA file \"" + filename + "\" (truncated)"; return false; }
                for (int i = 0; i < n && o < pix.size(); ++i, o += pb) memcpy(&pix[o], &f[pos], pb);
                pos += pb;
            } else {
                const size_t nb = (size_t)n * pb;
                if (pos + nb > f.size() || o + nb > pix.size()) { *err = "Unable to read from TGA file \"" + filename + "\" (truncated)"; return false; }
                memcpy(&pix[o], &f[pos], nb);
                pos += nb; o += nb;
            }
        }
    }
    const bool rightToLeft = (desc & 0x10) != 0, topToBottom = (desc & 0x20) != 0;
    out->resize((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sx = rightToLeft ? w - 1 - x : x, sy = topToBottom ? y : h - 1 - y;
            const unsigned char *src = &pix[((size_t)sy * w + sx) * pb];
            RGB v;
            if (base == 3) v = RGB(src[0] / 255.f);
            else {
                const unsigned char *bgr = src;
                unsigned char tmp[3];
                if (base == 1) {
                    int idx = (int)src[0] - cmapFirst;
                    if (idx < 0 || idx >= cmapLen) idx = 0;
                    bgr = cmap + (size_t)idx * cmapBytes;
                } else if (pb == 2) {   // 5-5-5
                    const int p = src[0] | (src[1] << 8);
                    tmp[0] = (unsigned char)(((p)&31) * 255 / 31); tmp[1] = (unsigned char)(((p >> 5) & 31) * 255 / 31); tmp[2] = (unsigned char)(((p >> 10) & 31) * 255 / 31);
                    bgr = tmp;
                }
                v.c[2] = bgr[0] / 255.f; v.c[1] = bgr[1] / 255.f; v.c[0] = bgr[2] / 255.f;
            }
            (*out)[(size_t)y * w + x] = v;
        }
    *xres = w; *yres = h;
    return true;
}

bool ReadPNG(const std::string &filename, int *xres, int *yres, std::vector<RGB> *out, std::string *err) {
    std::vector<unsigned char> f;
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    auto fail = [&](const char *why) { *err = "Error reading PNG \"" + filename + "\": " + why; return false; };
    if (!ReadFile(filename, &f) || f.size() < 8 + 25 || memcmp(f.data(), sig, 8)) return fail("not a PNG file");
    auto be32 = [&](size_t o) { return ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]; };
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    for (size_t pos = 8; pos + 12 <= f.size();) {
        const uint32_t len = be32(pos);
        if (pos + 12 + (size_t)len > f.size()) return fail("truncated chunk");
        const char *tag = (const char *)&f[pos + 4];
        const unsigned char *data = &f[pos + 8];
        if (!memcmp(tag, "IHDR", 4) && len >= 13) { w = be32(pos + 8); h = be32(pos + 12); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!memcmp(tag, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 65536 || h > 65536) return fail("bad header");
    if (interlace != 0) return fail("interlaced images are not read by this build");
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: return fail("bad colour type"); }
    if (depth != 8 && depth != 16 && !(depth < 8 && (ctype == 0 || ctype == 3))) return fail("unsupported bit depth");
    const size_t bitsPerPixel = (size_t)channels * depth, rowBytes = (w * bitsPerPixel + 7) / 8, bpp = std::max<size_t>(1, bitsPerPixel / 8);
    std::vector<unsigned char> raw((rowBytes + 1) * h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return fail("inflate failed");
    // undo the scanline filters (PNG specification, section 9)
    std::vector<unsigned char> img(rowBytes * h);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char *in = &raw[(rowBytes + 1) * y];
        const int ft = in[0];
        ++in;
        unsigned char *cur = &img[rowBytes * y];
        const unsigned char *up = y ? &img[rowBytes * (y - 1)] : nullptr;
        for (size_t i = 0; i < rowBytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: { int p = a + b - c, pa = std::abs(p - a), pbb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pbb && pa <= pc) ? a : (pbb <= pc ? b : c); break; }
            default: return fail("bad filter type");
            }
            cur[i] = (unsigned char)v;
        }
    }
    out->resize((size_t)w * h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const unsigned char *row = &img[rowBytes * y];
            auto sample = [&](int ch) -> int {   // 8-bit value of channel ch of pixel x
                if (depth == 8) return row[(size_t)x * channels + ch];
                if (depth == 16) return row[((size_t)x * channels + ch) * 2];   // high byte
                const size_t bit = (size_t)x * depth;
                const int v = (row[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
                return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
            };
            unsigned char rgb[3];
            if (ctype == 3) {
                const size_t idx = (size_t)sample(0);
                for (int k = 0; k < 3; ++k) rgb[k] = (3 * idx + k < plte.size()) ? plte[3 * idx + k] : 0;
            } else if (ctype == 0 || ctype == 4) rgb[0] = rgb[1] = rgb[2] = (unsigned char)sample(0);
            else for (int k = 0; k < 3; ++k) rgb[k] = (unsigned char)sample(k);
            RGB v;
            for (int k = 0; k < 3; ++k) v.c[k] = rgb[k] / 255.f;
            (*out)[(size_t)y * w + x] = v;
        }
    *xres = (int)w; *yres = (int)h;
    return true;
}

// ---- OpenEXR, the subset the reference's scenes use: single-part scan-line files, channels R / G / B (HALF or
// FLOAT, sampling 1), compression NONE, RLE, ZIPS, ZIP or PIZ. The reference reads through Imf::RgbaInputFile
// (imageio.cpp:121-160), whose frame buffer is HALF: FLOAT channels are rounded to half on the way in, missing colour
// channels read as 0. Layout: OpenEXR file layout document ("Structure of a scan-line file", "Predictor and
// reordering" for the zip / rle codecs).
inline float HalfToFloat(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {   // subnormal half
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
inline uint16_t FloatToHalf(float f) {   // round to nearest even, as half(float) does
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000);
    const uint32_t ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00);    // NaN
    if (ax >= 0x47800000u) return (uint16_t)(sign | 0x7c00);   // >= 65536 (and infinity)
    if (ax < 0x38800000u) {                                     // below 2^-14: a subnormal half, units of 2^-24
        if (ax < 0x33000000u) return sign;                      // below 2^-25
        const int shift = 126 - (int)(ax >> 23);                // 14 .. 24
        const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        uint32_t q = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) ++q;
        return (uint16_t)(sign | q);                            // (q == 0x400 is the smallest normal half: the encodings join)
    }
    uint32_t h = (((ax >> 23) - 112u) << 10) | ((ax & 0x7fffffu) >> 13);
    const uint32_t rem = ax & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;    // a carry runs into the exponent, up to infinity
    return (uint16_t)(sign | h);
}

// ---- PIZ (compression 4): a 16-bit wavelet transform of each channel of a 32-line block, the values first mapped onto a dense range
// through a bitmap of the values that occur, then Huffman-coded. Written from the published description of the format (OpenEXR
// technical introduction, "PIZ"; the reference reads it through the OpenEXR library, which this image does not hold -- there is no
// PIZ file from another writer here, so this reader is pinned by a round trip with the test suite's own writer only).
namespace piz {
constexpr int kEncSize = (1 << 16) + 1;   // symbols 0..65535 and the run-length marker
struct BitReader {
    const unsigned char *p, *end;
    uint64_t c = 0;
    int lc = 0;
    bool ok = true;
    unsigned Get(int n) {   // most significant bit first
        while (lc < n) { if (p >= end) { ok = false; return 0; } c = (c << 8) | *p++; lc += 8; }
        lc -= n;
        return (unsigned)((c >> lc) & ((1ull << n) - 1));
    }
};
// code lengths -> canonical codes: the longest codes are numbered from 0, each shorter length starts at half of where the longer
// one ended; codes of one length go to the symbols in increasing order
inline void CanonicalCodes(const std::vector<unsigned char> &len, uint64_t base[59], unsigned count[59]) {
    for (int i = 0; i < 59; ++i) { base[i] = 0; count[i] = 0; }
    for (unsigned char l : len) ++count[l];
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) { const uint64_t nc = (c + count[i]) >> 1; base[i] = c; c = nc; }
}
// Huffman-coded 16-bit values. Layout: im, iM (first and last symbol with a code), table length, number of data bits, a reserved
// word (five little-endian 32-bit words), the code lengths of symbols im..iM in 6 bits each (59..62: a run of 2..5 zero lengths;
// 63: a run of 6 + the next 8 bits), then the data. The symbol iM is the run-length marker: the 8 bits behind it repeat the
// previous value that many times.
inline bool HufDecode(const unsigned char *in, size_t nIn, uint16_t *out, size_t nOut) {
    if (nIn == 0) return nOut == 0;
    if (nIn < 20) return false;
    auto u32 = [&](size_t o) { return (uint32_t)in[o] | ((uint32_t)in[o + 1] << 8) | ((uint32_t)in[o + 2] << 16) | ((uint32_t)in[o + 3] << 24); };
    const uint32_t im = u32(0), iM = u32(4), nBits = u32(12);
    if (im >= (uint32_t)kEncSize || iM >= (uint32_t)kEncSize || im > iM) return false;
    std::vector<unsigned char> len(kEncSize, 0);
    BitReader tr{in + 20, in + nIn};
    for (uint32_t i = im; i <= iM; ++i) {
        const unsigned l = tr.Get(6);
        if (!tr.ok) return false;
        if (l == 63 || l >= 59) {
            const unsigned run = l == 63 ? tr.Get(8) + 6 : l - 59 + 2;
            if (!tr.ok || i + run > iM + 1) return false;
            i += run - 1;   // (the lengths stay 0)
        } else len[i] = (unsigned char)l;
    }
    const unsigned char *data = tr.p;   // (the table ends on a byte boundary: what is left in the reader's last byte is padding)
    if ((uint64_t)nBits > 8ull * (uint64_t)(in + nIn - data)) return false;
    uint64_t base[59];
    unsigned count[59];
    CanonicalCodes(len, base, count);
    std::vector<uint32_t> first(60, 0);            // symbols by length, in increasing order
    for (int l = 1; l < 59; ++l) first[l + 1] = first[l] + count[l];
    std::vector<uint32_t> syms(first[59]), fill(first.begin(), first.end());
    for (uint32_t i = im; i <= iM; ++i) if (len[i]) syms[fill[len[i]]++] = i;
    BitReader br{data, in + nIn};
    uint64_t left = nBits;
    size_t o = 0;
    while (left > 0) {
        uint64_t code = 0;
        int l = 0;
        uint32_t sym = 0;
        bool found = false;
        while (l < 58 && left > 0) {
            code = (code << 1) | br.Get(1);
            --left; ++l;
            if (!br.ok) return false;
            if (count[l] && code >= base[l] && code - base[l] < count[l]) { sym = syms[first[l] + (uint32_t)(code - base[l])]; found = true; break; }
        }
        if (!found) return false;
        if (sym == iM) {   // run-length marker
            if (left < 8 || o == 0) return false;
            const unsigned run = br.Get(8);
            left -= 8;
            if (!br.ok || o + run > nOut) return false;
            for (unsigned k = 0; k < run; ++k, ++o) out[o] = out[o - 1];
        } else {
            if (o >= nOut) return false;
            out[o++] = (uint16_t)sym;
        }
    }
    return o == nOut;
}
// the inverse of the two-value lifting step, for data below 2^14 (wdec14) and for the full 16-bit range (wdec16, modulo arithmetic)
inline void Wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int16_t ls = (int16_t)l, hs = (int16_t)h;
    const int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hi);
}
inline void Wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb; a = (uint16_t)aa;
}
// 2D wavelet decoding of nx x ny values at strides ox, oy, from the coarsest level down
inline void Wav2Decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) { if (w14) Wdec14(l, h, a, b); else Wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t *py = in, *ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py, *ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) { uint16_t *p10 = px + oy1; dec(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            uint16_t *px = py, *ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) { uint16_t *p01 = px + ox1; dec(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}
// One block: `raw` receives the scan lines as an uncompressed file stores them (per line, per channel, its pixels).
// chSize[c]: 16-bit words per pixel of channel c (HALF 1, FLOAT / UINT 2).
inline bool Decompress(const unsigned char *in, size_t nIn, const std::vector<int> &chSize, int w, int nLines, unsigned char *raw) {
    size_t total = 0;
    for (int sz : chSize) total += (size_t)w * nLines * sz;
    if (nIn < 4) return false;
    const unsigned minNonZero = in[0] | (in[1] << 8), maxNonZero = in[2] | (in[3] << 8);
    if (maxNonZero >= 8192) return false;
    std::vector<unsigned char> bitmap(8192, 0);
    size_t pos = 4;
    if (minNonZero <= maxNonZero) {
        const size_t nb = maxNonZero - minNonZero + 1;
        if (pos + nb > nIn) return false;
        memcpy(&bitmap[minNonZero], in + pos, nb);
        pos += nb;
    }
    std::vector<uint16_t> lut(65536, 0);
    unsigned k = 0;
    for (unsigned i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t maxValue = (uint16_t)(k - 1);
    if (pos + 4 > nIn) return false;
    const uint32_t length = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
    pos += 4;
    if ((size_t)length > nIn - pos) return false;
    std::vector<uint16_t> tmp(total);
    if (!HufDecode(in + pos, length, tmp.data(), total)) return false;
    std::vector<size_t> start(chSize.size());
    size_t o = 0;
    for (size_t c = 0; c < chSize.size(); ++c) {
        start[c] = o;
        for (int j = 0; j < chSize[c]; ++j) Wav2Decode(&tmp[o + j], w, chSize[c], nLines, w * chSize[c], maxValue);
        o += (size_t)w * nLines * chSize[c];
    }
    for (uint16_t &v : tmp) v = lut[v];
    unsigned char *dst = raw;
    for (int y = 0; y < nLines; ++y)
        for (size_t c = 0; c < chSize.size(); ++c) {
            const size_t n = (size_t)w * chSize[c];
            const uint16_t *src = &tmp[start[c] + (size_t)y * n];
            for (size_t i = 0; i < n; ++i) { dst[0] = (unsigned char)(src[i] & 0xff); dst[1] = (unsigned char)(src[i] >> 8); dst += 2; }
        }
    return true;
}
}  // namespace piz

bool ReadEXR(const std::string &filename, int *xres, int *yres, std::vector<RGB> *out, std::string *err) {
    std::vector<unsigned char> f;
    auto fail = [&](const std::string &why) { *err = "Unable to read image file \"" + filename + "\": " + why; return false; };
    if (!ReadFile(filename, &f) || f.size() < 16) return fail("cannot open, or not an OpenEXR file");
    auto i32 = [&](size_t o) { int32_t v; memcpy(&v, &f[o], 4); return v; };
    if ((uint32_t)i32(0) != 20000630u) return fail("not an OpenEXR file");
    const uint32_t version = (uint32_t)i32(4);
    if ((version & 0xff) != 2 || (version & 0x1a00)) return fail("tiled, deep or multi-part OpenEXR files are not read by this build");
    size_t pos = 8;
    struct Channel { std::string name; int type; };
    std::vector<Channel> channels;
    int compression = -1, lineOrder = 0, dw[4] = {0, 0, -1, -1};
    // a NUL-terminated string inside [p, end)
    auto str = [&](size_t &p, size_t end, std::string *out) {
        size_t e = p;
        while (e < end && f[e] != 0) ++e;
        if (e >= end) return false;
        out->assign((const char *)&f[p], e - p);
        p = e + 1;
        return true;
    };
    while (pos < f.size() && f[pos] != 0) {
        std::string name, type;
        if (!str(pos, f.size(), &name) || !str(pos, f.size(), &type) || pos + 4 > f.size()) return fail("truncated header");
        const int size = i32(pos); pos += 4;
        if (size < 0 || pos + (size_t)size > f.size()) return fail("truncated header");
        const size_t end = pos + (size_t)size;
        if (name == "channels") {
            size_t p = pos;
            while (p < end && f[p] != 0) {
                Channel c;
                if (!str(p, end, &c.name) || p + 16 > end) return fail("bad channel list");
                c.type = i32(p);
                const int xs = i32(p + 8), ys = i32(p + 12);
                p += 16;
                if (xs != 1 || ys != 1) return fail("subsampled channels are not read by this build");
                channels.push_back(c);
            }
        } else if (name == "compression" && size >= 1) compression = f[pos];
        else if (name == "dataWindow" && size >= 16) for (int k = 0; k < 4; ++k) dw[k] = i32(pos + 4 * k);
        else if (name == "lineOrder" && size >= 1) lineOrder = f[pos];
        pos = end;
    }
    ++pos;   // end of header
    (void)lineOrder;   // every chunk carries its y
    const long long wl = (long long)dw[2] - dw[0] + 1, hl = (long long)dw[3] - dw[1] + 1;
    if (wl <= 0 || hl <= 0 || wl > 65536 || hl > 65536 || channels.empty() || channels.size() > 64) return fail("bad header");
    const int w = (int)wl, h = (int)hl;
    int linesPerBlock;
    switch (compression) { case 0: case 1: case 2: linesPerBlock = 1; break; case 3: linesPerBlock = 16; break; case 4: linesPerBlock = 32; break;
                           default: return fail("compression method " + std::to_string(compression) + " (PXR24, B44, DWA) is not read by this build"); }
    size_t lineBytes = 0;
    std::vector<size_t> chOffset(channels.size());
    for (size_t c = 0; c < channels.size(); ++c) {
        if (channels[c].type < 0 || channels[c].type > 2) return fail("bad channel type");
        chOffset[c] = lineBytes;
        lineBytes += (size_t)w * (channels[c].type == 1 ? 2 : 4);
    }
    const int nChunks = (h + linesPerBlock - 1) / linesPerBlock;
    if (pos + (size_t)nChunks * 8 > f.size()) return fail("truncated offset table");
    std::vector<float> plane[3];
    int src[3] = {-1, -1, -1};
    for (size_t c = 0; c < channels.size(); ++c) {
        if (channels[c].name == "R") src[0] = (int)c; else if (channels[c].name == "G") src[1] = (int)c; else if (channels[c].name == "B") src[2] = (int)c;
        else if (channels[c].name == "Y" && src[0] < 0) { src[0] = src[1] = src[2] = (int)c; }   // luminance-only file
    }
    out->assign((size_t)w * h, RGB(0.f));
    std::vector<unsigned char> raw, tmp;
    for (int chunk = 0; chunk < nChunks; ++chunk) {
        uint64_t off;
        memcpy(&off, &f[pos + (size_t)chunk * 8], 8);
        if (off > f.size() || off + 8 > f.size()) return fail("truncated chunk");
        const int y = i32(off), dataSize = i32(off + 4);
        if (dataSize < 0 || off + 8 + (size_t)dataSize > f.size()) return fail("truncated chunk");
        const long long y0l = (long long)y - dw[1];
        if (y0l < 0 || y0l >= h) return fail("chunk outside the data window");
        const int y0 = (int)y0l, nLines = std::min(linesPerBlock, h - y0);
        const size_t rawSize = lineBytes * nLines;
        raw.resize(rawSize);
        const unsigned char *data = &f[off + 8];
        if (compression == 0 && (size_t)dataSize != rawSize) return fail("truncated chunk");
        if (compression == 0 || (size_t)dataSize == rawSize) memcpy(raw.data(), data, rawSize);
        else if (compression == 4) {
            std::vector<int> chSize(channels.size());
            for (size_t c = 0; c < channels.size(); ++c) chSize[c] = channels[c].type == 1 ? 1 : 2;
            if (!piz::Decompress(data, (size_t)dataSize, chSize, w, nLines, raw.data())) return fail("bad PIZ data");
        } else {
            tmp.resize(rawSize);
            if (compression == 1) {   // run-length
                size_t o = 0, i = 0;
                while (i < (size_t)dataSize && o < rawSize) {
                    const int count = (signed char)data[i++];
                    if (count < 0) { const size_t n = (size_t)(-count); if (i + n > (size_t)dataSize || o + n > rawSize) return fail("bad RLE data"); memcpy(&tmp[o], &data[i], n); i += n; o += n; }
                    else { const size_t n = (size_t)count + 1; if (i >= (size_t)dataSize || o + n > rawSize) return fail("bad RLE data"); memset(&tmp[o], data[i++], n); o += n; }
                }
                if (o != rawSize) return fail("bad RLE data");
            } else {
                uLongf n = (uLongf)rawSize;
                if (uncompress(tmp.data(), &n, data, (uLong)dataSize) != Z_OK || n != rawSize) return fail("inflate failed");
            }
            for (size_t i = 1; i < rawSize; ++i) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);   // predictor
            const size_t halfN = (rawSize + 1) / 2;                                                  // reordering
            for (size_t i = 0; i < rawSize; ++i) raw[i] = (i & 1) ? tmp[halfN + i / 2] : tmp[i / 2];
        }
        for (int l = 0; l < nLines; ++l) {
            const unsigned char *line = &raw[lineBytes * l];
            for (int k = 0; k < 3; ++k) {
                if (src[k] < 0) continue;
                const Channel &c = channels[src[k]];
                const unsigned char *p = line + chOffset[src[k]];
                for (int x = 0; x < w; ++x) {
                    float v;
                    if (c.type == 1) { uint16_t hbits; memcpy(&hbits, p + 2 * x, 2); v = HalfToFloat(hbits); }
                    else if (c.type == 2) { float fv; memcpy(&fv, p + 4 * x, 4); v = HalfToFloat(FloatToHalf(fv)); }
                    else { uint32_t u; memcpy(&u, p + 4 * x, 4); v = HalfToFloat(FloatToHalf((float)u)); }
                    (*out)[(size_t)(y0 + l) * w + x].c[k] = v;
                }
            }
        }
    }
    *xres = w; *yres = h;
    return true;
}

}  // namespace

bool WriteEXR(const std::string &filename, int w, int h, const float *rgb, std::string *err) {
    std::vector<unsigned char> head;
    auto put = [&](const void *p, size_t n) { head.insert(head.end(), (const unsigned char *)p, (const unsigned char *)p + n); };
    auto putStr = [&](const char *s) { put(s, strlen(s) + 1); };
    auto attr = [&](const char *name, const char *type, const void *data, int32_t size) { putStr(name); putStr(type); put(&size, 4); put(data, size); };
    const uint32_t magic = 20000630u, version = 2;
    put(&magic, 4); put(&version, 4);
    {
        std::vector<unsigned char> ch;
        for (const char *n : {"B", "G", "R"}) {
            ch.insert(ch.end(), n, n + 2);
            const int32_t v[4] = {1 /* HALF */, 0, 1, 1};
            ch.insert(ch.end(), (const unsigned char *)v, (const unsigned char *)v + 16);
        }
        ch.push_back(0);
        attr("channels", "chlist", ch.data(), (int32_t)ch.size());
    }
    const unsigned char zip = 3, incY = 0;
    attr("compression", "compression", &zip, 1);
    const int32_t box[4] = {0, 0, w - 1, h - 1};
    attr("dataWindow", "box2i", box, 16);
    attr("displayWindow", "box2i", box, 16);
    attr("lineOrder", "lineOrder", &incY, 1);
    const float one = 1.f, centre[2] = {0.f, 0.f};
    attr("pixelAspectRatio", "float", &one, 4);
    attr("screenWindowCenter", "v2f", centre, 8);
    attr("screenWindowWidth", "float", &one, 4);
    head.push_back(0);
    const int linesPerBlock = 16, nChunks = (h + linesPerBlock - 1) / linesPerBlock;
    std::vector<std::vector<unsigned char>> chunks(nChunks);
    const size_t lineBytes = (size_t)w * 3 * 2;
    std::vector<unsigned char> raw, tmp;
    for (int c = 0; c < nChunks; ++c) {
        const int y0 = c * linesPerBlock, nLines = std::min(linesPerBlock, h - y0);
        raw.resize(lineBytes * nLines);
        for (int l = 0; l < nLines; ++l)
            for (int k = 0; k < 3; ++k)   // channel order B, G, R
                for (int x = 0; x < w; ++x) {
                    const uint16_t hv = FloatToHalf(rgb[3 * ((size_t)(y0 + l) * w + x) + (2 - k)]);
                    memcpy(&raw[lineBytes * l + ((size_t)k * w + x) * 2], &hv, 2);
                }
        const size_t n = raw.size(), halfN = (n + 1) / 2;
        tmp.resize(n);
        for (size_t i = 0; i < n; ++i) tmp[(i & 1) ? halfN + i / 2 : i / 2] = raw[i];            // reordering
        for (size_t i = n - 1; i >= 1; --i) tmp[i] = (unsigned char)(tmp[i] - tmp[i - 1] + 128);   // predictor
        uLongf zn = compressBound((uLong)n);
        std::vector<unsigned char> z(zn);
        const bool packed = compress2(z.data(), &zn, tmp.data(), (uLong)n, 6) == Z_OK && zn < n;
        const int32_t y = y0, size = (int32_t)(packed ? zn : n);
        chunks[c].insert(chunks[c].end(), (const unsigned char *)&y, (const unsigned char *)&y + 4);
        chunks[c].insert(chunks[c].end(), (const unsigned char *)&size, (const unsigned char *)&size + 4);
        if (packed) chunks[c].insert(chunks[c].end(), z.begin(), z.begin() + zn);
        else chunks[c].insert(chunks[c].end(), raw.begin(), raw.end());
    }
    FILE *f = fopen(filename.c_str(), "wb");
    if (!f) { *err = "Error writing \"" + filename + "\""; return false; }
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
    uint64_t off = head.size() + (uint64_t)nChunks * 8;
    for (int c = 0; c < nChunks && ok; ++c) { ok = fwrite(&off, 8, 1, f) == 1; off += chunks[c].size(); }
    for (int c = 0; c < nChunks && ok; ++c) ok = fwrite(chunks[c].data(), 1, chunks[c].size(), f) == chunks[c].size();
    fclose(f);
    if (!ok) *err = "Error writing \"" + filename + "\"";
    return ok;
}

bool ReadImage(const std::string &filename, int *xres, int *yres, std::vector<RGB> *texels, std::string *err) {
    const size_t dot = filename.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : filename.substr(dot);
    for (char &c : ext) c = (char)tolower(c);
    if (ext == ".pfm") {
        if (ReadPFM(filename, xres, yres, texels)) return true;
        *err = "Error reading PFM file \"" + filename + "\"";
        return false;
    }
    if (ext == ".tga") return ReadTGA(filename, xres, yres, texels, err);
    if (ext == ".png") return ReadPNG(filename, xres, yres, texels, err);
    if (ext == ".exr") return ReadEXR(filename, xres, yres, texels, err);
    *err = "Unable to load image stored in format \"" + (ext.empty() ? std::string("(unknown)") : ext.substr(1)) + "\" for filename \"" +
           filename + "\" (this build reads PFM, TGA, PNG and scan-line EXR).";
    return false;
}

MIPMap::MIPMap(int rx, int ry, const std::vector<RGB> &img, ImageWrap wrapMode) : wrap(wrapMode) {
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
        auto wrapIndex = [&](int i, int res) {
            if (wrap == ImageWrap::Repeat) return Mod(i, res);
            if (wrap == ImageWrap::Clamp) return std::min(std::max(i, 0), res - 1);
            return i;
        };
        std::vector<RGB> res((size_t)px * py);
        std::vector<W> sW = weights(rx, px);
        for (int t = 0; t < ry; ++t)
            for (int s = 0; s < px; ++s) {
                RGB &o = res[(size_t)t * px + s];
                o = RGB(0.f);
                for (int j = 0; j < 4; ++j) {
                    int origS = wrapIndex(sW[s].first + j, rx);
                    if (origS >= 0 && origS < rx) o += sW[s].w[j] * img[(size_t)t * rx + origS];
                }
            }
        std::vector<W> tW = weights(ry, py);
        std::vector<RGB> work(py);
        for (int s = 0; s < px; ++s) {
            for (int t = 0; t < py; ++t) {
                work[t] = RGB(0.f);
                for (int j = 0; j < 4; ++j) {
                    int offset = wrapIndex(tW[t].first + j, ry);
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

RGB MIPMap::Texel(int level, int s, int t) const {
    const Level &l = pyramid[level];
    switch (wrap) {
    case ImageWrap::Repeat: s = Mod(s, l.w); t = Mod(t, l.h); break;
    case ImageWrap::Clamp: s = std::min(std::max(s, 0), l.w - 1); t = std::min(std::max(t, 0), l.h - 1); break;
    case ImageWrap::Black: if (s < 0 || s >= l.w || t < 0 || t >= l.h) return RGB(0.f); break;
    }
    return l.t[(size_t)t * l.w + s];
}
RGB MIPMap::triangle(int level, const float st[2]) const {
    level = std::min(std::max(level, 0), Levels() - 1);
    float s = st[0] * pyramid[level].w - 0.5f;
    float t = st[1] * pyramid[level].h - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * Texel(level, s0, t0) + (1 - ds) * dt * Texel(level, s0, t0 + 1) +
           ds * (1 - dt) * Texel(level, s0 + 1, t0) + ds * dt * Texel(level, s0 + 1, t0 + 1);
}
RGB MIPMap::Lookup(const float st[2], float width) const {
    float level = Levels() - 1 + Log2(std::max(width, (float)1e-8));
    if (level < 0) return triangle(0, st);
    else if (level >= Levels() - 1) return Texel(Levels() - 1, 0, 0);
    int iLevel = (int)std::floor(level);
    float delta = level - iLevel;
    return (1 - delta) * triangle(iLevel, st) + delta * triangle(iLevel + 1, st);  // Lerp, pbrt.h:420
}

}  // namespace mipt
