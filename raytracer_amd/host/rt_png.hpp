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
// Non-interlaced 8/16-bit gray, gray+alpha, RGB, RGBA and 8-bit palette PNGs.  Returns the decoded
// image with its native channel count; 16-bit samples are reduced with >> 8 as stb_image does.
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
    std::vector<unsigned char> idat, plte;
    for (size_t p = 8; p + 12 <= d.size();) {
        const uint32_t len = rd32(p);
        if (p + 12 + len > d.size()) return false;
        const std::string t((const char*)&d[p + 4], 4);
        const unsigned char* body = &d[p + 8];
        if (t == "IHDR") { if (len < 13) return false; w = (int)rd32(p + 8); h = (int)rd32(p + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (t == "PLTE") plte.assign(body, body + len);
        else if (t == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (t == "IEND") break;
        p += 12 + len;
    }
    if (w <= 0 || h <= 0 || interlace != 0 || (depth != 8 && depth != 16) || (ctype == 3 && depth != 8)) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const int bpp = ch * depth / 8;
    // stb_image's limit on either extent (STBI_MAX_DIMENSIONS); and deflate cannot expand by more than 1032 : 1, so a header
    // that asks for more than the IDAT data can hold is damaged: refuse it before sizing buffers by it
    if (w > (1 << 24) || h > (1 << 24)) return false;
    const size_t stride = (size_t)w * bpp;
    if ((unsigned long long)(stride + 1) * (unsigned long long)h > 1032ull * idat.size() + 64ull) return false;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf rlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rlen, idat.data(), (uLong)idat.size()) != Z_OK || rlen != raw.size()) return false;
    std::vector<unsigned char> img(stride * (size_t)h);
    for (int y = 0; y < h; y++) {
        const unsigned char ft = raw[(stride + 1) * y];
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
    nCh = ctype == 3 ? 3 : ch;
    pix.resize((size_t)w * h * nCh);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        if (ctype == 3) { const int k = img[i]; for (int c = 0; c < 3; c++) pix[i * 3 + c] = (size_t)(k * 3 + c) < plte.size() ? plte[k * 3 + c] : 0; }
        else for (int c = 0; c < ch; c++) pix[i * ch + c] = depth == 8 ? img[i * ch + c] : img[(i * ch + c) * 2]; // big-endian high byte = >> 8
    }
    return true;
}
} // namespace png

} // namespace rth
