// rt_png.hpp -- PNG codec of the host mirror over zlib (the reference vendors lodepng for writing, src/lodepng.cpp, and
// stb_image for reading, src/stb_image.h): RGBA8 writer for the screenshot path, reader for Texture(path).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rth {
namespace png {
inline void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
inline void chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data)
{
    put32(out, (uint32_t)data.size());
    std::vector<unsigned char> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, (uint32_t)crc32(0L, body.data(), (uInt)body.size()));
}
// RGBA8, row 0 = top
inline bool encode(const std::string& path, const unsigned char* rgba, unsigned w, unsigned h)
{
    std::vector<unsigned char> raw;
    raw.reserve((size_t)h * (w * 4 + 1));
    for (unsigned y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4); }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    z.resize(zlen);
    std::vector<unsigned char> out = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    std::vector<unsigned char> ihdr; put32(ihdr, w); put32(ihdr, h); ihdr.insert(ihdr.end(), { 8, 6, 0, 0, 0 });
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}
inline unsigned char paeth(int a, int b, int c) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (unsigned char)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)); }
// Every PNG that stb_image v2.28 (the reference's decoder, src/stb_image.h, called as stbi_load(path, ..., 0) at Material.cu:179)
// reads, with its conversions, so that the bytes are the ones the reference gets -- PNG is lossless, so a conforming decode
// plus these rules is all there is to it:
//   bit depths 1, 2, 4, 8, 16; gray, gray + alpha, RGB, RGBA, palette; Adam7 interlacing;
//   16-bit samples reduced with >> 8 (stb_image.h:1198); 1-, 2-, 4-bit gray scaled by 255, 85, 17 (stbi__depth_scale_table);
//   a tRNS chunk adds an alpha channel: palette -> RGBA with the listed alphas, gray -> gray + alpha, RGB -> RGBA with alpha 0
//   where the pixel equals the colour key (compared at 16 bits for 16-bit images, after scaling for the low depths) and 255
//   elsewhere; a tRNS chunk on an image that already has alpha is an error there and a refusal here.
// Returns the image with stb_image's channel count (1, 2, 3 or 4).
inline bool unfilter(const unsigned char* raw, size_t rawLen, int w, int h, int ch, int depth, std::vector<unsigned char>& img)
{
    const size_t stride = ((size_t)w * ch * depth + 7) / 8;
    const int bpp = depth < 8 ? 1 : ch * depth / 8;
    if ((stride + 1) * (size_t)h > rawLen) return false;
    img.assign(stride * (size_t)h, 0);
    for (int y = 0; y < h; y++) {
        const unsigned char ft = raw[(stride + 1) * y];
        if (ft > 4) return false;
        const unsigned char* s = &raw[(stride + 1) * y + 1];
        unsigned char* o = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? o[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int v = s[i];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
            o[i] = (unsigned char)v;
        }
    }
    return true;
}
// sample k of a row of unfiltered bytes, as a 16-bit value for depth 16 and as the raw field otherwise
inline unsigned sample_at(const unsigned char* row, size_t k, int depth)
{
    if (depth == 16) return ((unsigned)row[2 * k] << 8) | row[2 * k + 1];
    if (depth == 8) return row[k];
    const size_t bit = k * depth;
    return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
}
inline bool decode(const std::string& path, std::vector<unsigned char>& pix, int& w, int& h, int& nCh)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> d;
    unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) return false;
    auto rd32 = [&](size_t p) { return ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | d[p + 3]; };
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    bool hasTrns = false;
    w = h = 0;
    for (size_t p = 8; p + 12 <= d.size();) {
        const uint32_t len = rd32(p);
        if (p + 12 + (size_t)len > d.size()) return false;
        const std::string t((const char*)&d[p + 4], 4);
        const unsigned char* body = &d[p + 8];
        if (t == "IHDR") { if (len < 13) return false; w = (int)rd32(p + 8); h = (int)rd32(p + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (t == "PLTE") plte.assign(body, body + len);
        else if (t == "tRNS") { trns.assign(body, body + len); hasTrns = true; }
        else if (t == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (t == "IEND") break;
        p += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || interlace > 1) return false;
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    if (depth < 8 && ctype != 0 && ctype != 3) return false;
    if (ctype == 3 && (depth == 16 || plte.empty())) return false;
    if (hasTrns && (ctype == 4 || ctype == 6)) return false;            // stb_image: "tRNS with alpha"
    if (hasTrns && ctype != 3 && trns.size() < (size_t)ch * 2) return false;
    // stb_image's limit on either extent (STBI_MAX_DIMENSIONS); and deflate cannot expand by more than 1032 : 1, so a header
    // that asks for more than the IDAT data can hold is damaged: refuse it before sizing buffers by it
    if (w > (1 << 24) || h > (1 << 24)) return false;
    const size_t stride = ((size_t)w * ch * depth + 7) / 8;
    const unsigned long long need = (unsigned long long)(stride + 1) * (unsigned long long)h;
    if (need > 1032ull * idat.size() + 64ull) return false;
    // interlaced data is at most 7 filter bytes per 8 rows + a byte of padding per pass row larger
    std::vector<unsigned char> raw((size_t)need + (interlace ? (size_t)h * 8 + 64 : 0));
    uLongf rlen = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &rlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || (!interlace && rlen != (uLongf)need)) return false;

    // samples of the whole image, one unsigned per sample (16-bit kept whole for the colour key)
    std::vector<unsigned short> smp((size_t)w * h * ch);
    if (!interlace) {
        std::vector<unsigned char> img;
        if (!unfilter(raw.data(), rlen, w, h, ch, depth, img)) return false;
        for (int y = 0; y < h; y++)
            for (size_t k = 0; k < (size_t)w * ch; k++) smp[(size_t)y * w * ch + k] = (unsigned short)sample_at(&img[stride * y], k, depth);
    } else {
        static const int xo[7] = { 0, 4, 0, 2, 0, 1, 0 }, yo[7] = { 0, 0, 4, 0, 2, 0, 1 }, xs[7] = { 8, 8, 4, 4, 2, 2, 1 }, ys[7] = { 8, 8, 8, 4, 4, 2, 2 };
        size_t off = 0;
        for (int p = 0; p < 7; p++) {
            const int pw = (w - xo[p] + xs[p] - 1) / xs[p], ph = (h - yo[p] + ys[p] - 1) / ys[p];
            if (pw <= 0 || ph <= 0) continue;
            const size_t ps = ((size_t)pw * ch * depth + 7) / 8;
            std::vector<unsigned char> img;
            if (off > rlen || !unfilter(raw.data() + off, rlen - off, pw, ph, ch, depth, img)) return false;
            off += (ps + 1) * (size_t)ph;
            for (int j = 0; j < ph; j++)
                for (int i = 0; i < pw; i++)
                    for (int c = 0; c < ch; c++)
                        smp[((size_t)(yo[p] + j * ys[p]) * w + (size_t)(xo[p] + i * xs[p])) * ch + c] = (unsigned short)sample_at(&img[ps * j], (size_t)i * ch + c, depth);
        }
    }
    const size_t npx = (size_t)w * h;
    if (ctype == 3) {
        nCh = hasTrns ? 4 : 3;
        pix.resize(npx * nCh);
        for (size_t i = 0; i < npx; i++) {
            const size_t k = smp[i];
            for (int c = 0; c < 3; c++) pix[i * nCh + c] = k * 3 + c < plte.size() ? plte[k * 3 + c] : 0;
            if (hasTrns) pix[i * 4 + 3] = k < trns.size() ? trns[k] : 255;
        }
        return true;
    }
    static const unsigned scaleTab[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    const unsigned scale = depth < 8 ? scaleTab[depth] : 1u;
    unsigned key[3] = { 0, 0, 0 };
    if (hasTrns)
        for (int c = 0; c < ch; c++) {
            const unsigned v = ((unsigned)trns[2 * c] << 8) | trns[2 * c + 1];
            key[c] = depth == 16 ? v : (v & 255u) * scale;   // stb_image keeps the low byte of the key and scales it like the samples
        }
    nCh = ch + (hasTrns ? 1 : 0);
    pix.resize(npx * nCh);
    for (size_t i = 0; i < npx; i++) {
        bool isKey = hasTrns;
        for (int c = 0; c < ch; c++) {
            const unsigned v16 = smp[i * ch + c];
            const unsigned v = depth == 16 ? v16 : (v16 * scale) & 255u;
            if (hasTrns && v != key[c]) isKey = false;
            pix[i * nCh + c] = (unsigned char)(depth == 16 ? v16 >> 8 : v);
        }
        if (hasTrns) pix[i * nCh + ch] = isKey ? 0 : 255;
    }
    return true;
}
} // namespace png

} // namespace rth
