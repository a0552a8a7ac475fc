// image.cpp -- ReadImage for .pfm / .tga / .png and MIPMap<RGBSpectrum> construction.
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

}  // namespace

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
    *err = "Unable to load image stored in format \"" + (ext.empty() ? std::string("(unknown)") : ext.substr(1)) + "\" for filename \"" +
           filename + "\" (this build reads PFM, TGA and PNG).";
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
