// rt_jpeg.hpp -- baseline JPEG decoder of the host mirror (the Texture(path) leg of the upload path, Material.cu:162-206).
//
// The reference decodes its texture files with the stb_image v2.28 it vendors (src/stb_image.h), and JPEG is the one format
// where the decoder shows in the texel bytes: the inverse DCT, the chroma upsampling filter and the YCbCr -> RGB conversion
// are all fixed-point choices.  This decoder makes stb_image's choices, restated from its published definitions:
//   - inverse DCT: the Loeffler-Ligtenberg-Moshovitz integer transform of libjpeg's jidctint ("ISLOW") with 12-bit constants,
//     columns first (2 extra bits kept, rounded with +512 >> 10), then rows (+65536 + (128 << 17) >> 17), clamped;
//     coefficients are dequantised into 16-bit integers (stb_image.h:2414-2512, 2208-2262);
//   - upsampling: the "JFIF-centred" triangle filters for 2x horizontally ((3a + b + 2) >> 2), 2x vertically, both
//     ((3 * (3a + b) + (3c + d) + 8) >> 4), nearest for any other factor (:3452-3526, 3644-3655);
//   - colour: y + 1.402 cr etc. in 20-bit fixed point with constants of 12 significant bits and the green chroma term masked
//     to its upper 16 bits (:3658-3684); three components without a JFIF / Adobe marker, or named 'R','G','B', are RGB (:3878);
//   - output: 1 channel for one component, 3 for three (stbi_load(..., 0), Material.cu:179).
// tests/test_texture_decode.py compares it with that stb_image itself (oracle/_ref, compiled from the reference checkout) on
// every JPEG the reference ships and on files written with other sampling factors and restart intervals: identical bytes.
// Refused (decode returns false, Texture::getLoadSuccess() is then false): progressive and arithmetic-coded files, 12-bit
// samples, four components.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rth {
namespace jpeg {

struct HuffTable {            // canonical Huffman code, ITU T.81 annex C / F.2.2.3
    bool present = false;
    uint8_t vals[256];
    int maxcode[18], valptr[17], mincode[17];
    bool build(const uint8_t counts[16], const uint8_t* symbols, int n)
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        if (k != n || n > 256) return false;
        memcpy(vals, symbols, (size_t)n);
        present = true;
        return true;
    }
};

struct Bits {                 // entropy-coded segment reader: FF00 -> FF, any other marker ends the data (zeros follow)
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0; int marker = 0;
    void fill()
    {
        while (n <= 24) {
            uint32_t b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    uint32_t c = p < end ? *p++ : 0xD9u;
                    while (c == 0xFF && p < end) c = *p++;   // fill bytes
                    if (c != 0) { marker = (int)c; b = 0; }
                }
            }
            acc |= b << (24 - n);
            n += 8;
        }
    }
    int get(int count) { if (n < count) fill(); const uint32_t v = count ? acc >> (32 - count) : 0u; acc <<= count; n -= count; return (int)v; }
    int symbol(const HuffTable& h)
    {
        if (n < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | (int)(acc >> 31);
            acc <<= 1; n--;
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
    int extend(int s) { const int v = get(s); return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; } // T.81 F.2.2.1
    void reset() { acc = 0; n = 0; marker = 0; }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0, pred = 0;
    std::vector<uint8_t> data;
};

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// Two's-complement int that wraps: a corrupt stream can hold coefficients that overflow the transform's 32-bit intermediates.
// stb_image computes in plain int there (wraps on every target it runs on); this type does the same without the undefined
// behaviour, so a damaged file decodes to the same garbage instead of whatever the optimiser makes of an overflow.
struct wint {
    uint32_t u;
    wint() : u(0) {}
    wint(int v) : u((uint32_t)v) {}
    int s() const { int v; memcpy(&v, &u, 4); return v; }
    friend wint operator+(wint a, wint b) { wint r; r.u = a.u + b.u; return r; }
    friend wint operator-(wint a, wint b) { wint r; r.u = a.u - b.u; return r; }
    friend wint operator*(wint a, wint b) { wint r; r.u = a.u * b.u; return r; }
    int operator>>(int k) const { return s() >> k; }   // arithmetic shift
};

// One pass of the LLM transform on eight values; C(x) = round(x * 4096).  even part in x0..x3, odd part in t0..t3.
struct Llm { wint x0, x1, x2, x3, t0, t1, t2, t3; };
inline Llm llm_1d(wint s0, wint s1, wint s2, wint s3, wint s4, wint s5, wint s6, wint s7)
{
    const wint C0_541 = 2217, Cm1_847 = -7567, C0_765 = 3135, C1_175 = 4816, C0_298 = 1223, C2_053 = 8410, C3_072 = 12586,
               C1_501 = 6149, Cm0_899 = -3685, Cm2_562 = -10497, Cm1_961 = -8034, Cm0_390 = -1597, k4096 = 4096;
    Llm r;
    const wint z = (s2 + s6) * C0_541;
    const wint e2 = z + s6 * Cm1_847, e3 = z + s2 * C0_765;
    const wint e0 = (s0 + s4) * k4096, e1 = (s0 - s4) * k4096;
    r.x0 = e0 + e3; r.x3 = e0 - e3; r.x1 = e1 + e2; r.x2 = e1 - e2;
    const wint a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
    const wint z5 = (a + b) * C1_175;
    const wint pc = z5 + c * Cm0_899, pd = z5 + d * Cm2_562, pa = a * Cm1_961, pb = b * Cm0_390;
    r.t3 = s1 * C1_501 + (pc + pb);
    r.t2 = s3 * C3_072 + (pd + pa);
    r.t1 = s5 * C2_053 + (pd + pb);
    r.t0 = s7 * C0_298 + (pc + pa);
    return r;
}

inline void idct8x8(uint8_t* out, int stride, const short d[64])
{
    int v[64];
    for (int i = 0; i < 8; i++) {
        if (!d[8 + i] && !d[16 + i] && !d[24 + i] && !d[32 + i] && !d[40 + i] && !d[48 + i] && !d[56 + i]) {
            const int dc = d[i] * 4;   // a column of one DC term
            for (int k = 0; k < 8; k++) v[k * 8 + i] = dc;
            continue;
        }
        Llm r = llm_1d(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
        const wint bias = 512;
        r.x0 = r.x0 + bias; r.x1 = r.x1 + bias; r.x2 = r.x2 + bias; r.x3 = r.x3 + bias;
        v[i] = (r.x0 + r.t3) >> 10; v[56 + i] = (r.x0 - r.t3) >> 10;
        v[8 + i] = (r.x1 + r.t2) >> 10; v[48 + i] = (r.x1 - r.t2) >> 10;
        v[16 + i] = (r.x2 + r.t1) >> 10; v[40 + i] = (r.x2 - r.t1) >> 10;
        v[24 + i] = (r.x3 + r.t0) >> 10; v[32 + i] = (r.x3 - r.t0) >> 10;
    }
    for (int i = 0; i < 8; i++) {
        const int* w = v + i * 8;
        uint8_t* o = out + (size_t)i * stride;
        Llm r = llm_1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        const wint bias = 65536 + (128 << 17);
        r.x0 = r.x0 + bias; r.x1 = r.x1 + bias; r.x2 = r.x2 + bias; r.x3 = r.x3 + bias;
        o[0] = clamp8((r.x0 + r.t3) >> 17); o[7] = clamp8((r.x0 - r.t3) >> 17);
        o[1] = clamp8((r.x1 + r.t2) >> 17); o[6] = clamp8((r.x1 - r.t2) >> 17);
        o[2] = clamp8((r.x2 + r.t1) >> 17); o[5] = clamp8((r.x2 - r.t1) >> 17);
        o[3] = clamp8((r.x3 + r.t0) >> 17); o[4] = clamp8((r.x3 - r.t0) >> 17);
    }
}

static const uint8_t kZigzag[64 + 16] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };   // a corrupt run past 63 lands on the last coefficient

// one row of a component stretched to the image width: near = the closer source row, far = the other one
inline const uint8_t* upsample_row(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2); return out; }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near[0]; return out; }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        int i = 1;
        for (; i < w - 1; i++) {
            const int n3 = 3 * near[i] + 2;
            out[i * 2] = (uint8_t)((n3 + near[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n3 + near[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[i * 2 + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return out; }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}

inline void ycbcr_row(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count)
{
    auto fx = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
    const int kR = fx(1.40200f), kGr = fx(0.71414f), kGb = fx(0.34414f), kB = fx(1.77200f);
    for (int i = 0; i < count; i++) {
        const int yf = (y[i] << 20) + (1 << 19), cr = pcr[i] - 128, cb = pcb[i] - 128;
        const int r = (yf + cr * kR) >> 20;
        const int g = (yf + (cr * -kGr) + (int)((uint32_t)(cb * -kGb) & 0xffff0000u)) >> 20;
        const int b = (yf + cb * kB) >> 20;
        out[i * 3] = clamp8(r); out[i * 3 + 1] = clamp8(g); out[i * 3 + 2] = clamp8(b);
    }
}

inline bool decode_memory(const uint8_t* d, size_t size, std::vector<unsigned char>& pix, int& width, int& height, int& nCh)
{
    if (size < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
    uint16_t quant[4][64]; bool haveQ[4] = { false, false, false, false };
    HuffTable dc[4], ac[4];
    std::vector<Component> comp;
    int restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, adobe = -1, rgbIds = 0;
    bool jfif = false, frame = false, scanned = false;
    size_t pos = 2;
    auto u16 = [&](size_t p) { return (int)((d[p] << 8) | d[p + 1]); };
    for (;;) {
        while (pos < size && d[pos] != 0xFF) pos++;          // stray bytes between segments
        while (pos < size && d[pos] == 0xFF) pos++;
        if (pos >= size) break;
        const int m = d[pos++];
        if (m == 0xD9) break;                                  // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;   // no payload
        if (pos + 2 > size) return false;
        const int len = u16(pos);
        if (len < 2 || pos + (size_t)len > size) return false;
        const uint8_t* s = d + pos + 2; const int n = len - 2;
        if (m == 0xDB) {                                       // DQT
            for (int k = 0; k < n;) {
                const int pq = s[k] >> 4, tq = s[k] & 15; k++;
                if (pq > 1 || tq > 3 || k + (pq ? 128 : 64) > n) return false;
                for (int i = 0; i < 64; i++) { quant[tq][kZigzag[i]] = (uint16_t)(pq ? ((s[k] << 8) | s[k + 1]) : s[k]); k += pq ? 2 : 1; }
                haveQ[tq] = true;
            }
        } else if (m == 0xC4) {                                // DHT
            for (int k = 0; k < n;) {
                if (k + 17 > n) return false;
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) return false;
                int total = 0;
                for (int i = 0; i < 16; i++) total += s[k + 1 + i];
                if (total > 256 || k + 17 + total > n) return false;
                if (!(tc ? ac[th] : dc[th]).build(s + k + 1, s + k + 17, total)) return false;
                k += 17 + total;
            }
        } else if (m == 0xDD) { if (n < 2) return false; restart = (s[0] << 8) | s[1]; }
        else if (m == 0xE0) { jfif = jfif || (n >= 5 && memcmp(s, "JFIF\0", 5) == 0); }
        else if (m == 0xEE) { if (n >= 12 && memcmp(s, "Adobe\0", 6) == 0) adobe = s[11]; }
        else if (m == 0xC0 || m == 0xC1) {                     // SOF0 / SOF1: sequential, Huffman
            if (frame || n < 6 || s[0] != 8) return false;
            height = u16(pos + 3); width = u16(pos + 5);
            const int nc = s[5];
            if (width <= 0 || height <= 0 || (nc != 1 && nc != 3) || n < 6 + 3 * nc) return false;
            comp.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                Component& c = comp[i];
                c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
                if (nc == 3 && c.id == "RGB"[i]) rgbIds++;
                hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
            }
            for (auto& c : comp) if (hmax % c.h || vmax % c.v) return false;
            mcux = (width + hmax * 8 - 1) / (hmax * 8); mcuy = (height + vmax * 8 - 1) / (vmax * 8);
            // A block costs at least two bits of entropy-coded data (an empty DC difference and an end-of-block), so a frame
            // header that asks for more than four blocks per byte of file is damaged: refuse it before sizing buffers by it
            // (a corrupt SOF can claim 65535 x 65535).  stb_image would allocate and decode such a file as far as it goes.
            unsigned long long blocks = 0;
            for (auto& c : comp) blocks += (unsigned long long)mcux * mcuy * c.h * c.v;
            if (blocks > 4ull * size) return false;
            for (auto& c : comp) {
                c.x = (width * c.h + hmax - 1) / hmax; c.y = (height * c.v + vmax - 1) / vmax;
                c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;
                c.data.assign((size_t)c.w2 * c.h2 + 15, 0);
            }
            frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) return false; // progressive, lossless, arithmetic ...
        else if (m == 0xDA) {                                  // SOS + entropy-coded data
            if (!frame || n < 1) return false;
            const int ns = s[0];
            if (ns < 1 || ns > (int)comp.size() || n < 1 + 2 * ns + 3) return false;
            int order[3];
            for (int i = 0; i < ns; i++) {
                int which = -1;
                for (size_t c = 0; c < comp.size(); c++) if (comp[c].id == s[1 + 2 * i]) which = (int)c;
                if (which < 0) return false;
                comp[which].td = s[2 + 2 * i] >> 4; comp[which].ta = s[2 + 2 * i] & 15;
                if (comp[which].td > 3 || comp[which].ta > 3) return false;
                order[i] = which;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return false; // spectral selection of a baseline scan
            Bits br; br.p = d + pos + len; br.end = d + size;
            for (auto& c : comp) c.pred = 0;
            int todo = restart ? restart : 0x7FFFFFFF;
            short blk[64];
            auto block = [&](Component& c, int bx, int by) -> bool {
                if (!dc[c.td].present || !ac[c.ta].present || !haveQ[c.tq]) return false;
                const uint16_t* q = quant[c.tq];
                memset(blk, 0, sizeof blk);
                const int t = br.symbol(dc[c.td]);
                if (t < 0 || t > 15) return false;
                c.pred = (wint(c.pred) + wint(t ? br.extend(t) : 0)).s();   // wraps on a corrupt stream, like the transform
                blk[0] = (short)(wint(c.pred) * wint(q[0])).s();
                for (int k = 1; k < 64;) {
                    const int rs = br.symbol(ac[c.ta]);
                    if (rs < 0) return false;
                    const int run = rs >> 4, sz = rs & 15;
                    if (sz == 0) { if (rs != 0xF0) break; k += 16; continue; }
                    k += run;
                    const int zig = kZigzag[k++];
                    blk[zig] = (short)(br.extend(sz) * q[zig]);
                }
                idct8x8(c.data.data() + (size_t)c.w2 * by * 8 + (size_t)bx * 8, c.w2, blk);
                return true;
            };
            auto mcu_done = [&]() -> bool {   // false: stop decoding this scan (no restart marker where one is due)
                if (--todo > 0) return true;
                br.fill();
                if (br.marker < 0xD0 || br.marker > 0xD7) return false;
                br.reset();
                for (auto& c : comp) c.pred = 0;
                todo = restart ? restart : 0x7FFFFFFF;
                return true;
            };
            bool go = true;
            if (ns == 1) {                                     // one component: its blocks in raster order
                Component& c = comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
                for (int j = 0; j < bh && go; j++) for (int i = 0; i < bw && go; i++) { if (!block(c, i, j)) return false; go = mcu_done(); }
            } else {
                for (int j = 0; j < mcuy && go; j++) for (int i = 0; i < mcux && go; i++) {
                    for (int k = 0; k < ns; k++) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) if (!block(c, i * c.h + x, j * c.v + y)) return false;
                    }
                    go = mcu_done();
                }
            }
            scanned = true;
            // continue after the entropy-coded data: at the marker the reader stopped on, or search for the next one
            pos = (size_t)(br.p - d);
            if (br.marker) { pos -= 2; }
            continue;
        }
        pos += (size_t)len;
    }
    if (!frame || !scanned) return false;
    const int nc = (int)comp.size();
    nCh = nc >= 3 ? 3 : 1;
    const bool isRgb = nc == 3 && (rgbIds == 3 || (adobe == 0 && !jfif));
    pix.assign((size_t)width * height * nCh, 0);
    struct Up { int hs, vs, ystep, ypos, wlo; const uint8_t *l0, *l1; std::vector<uint8_t> buf; };
    std::vector<Up> up((size_t)nc);
    for (int k = 0; k < nc; k++) {
        Up& u = up[k];
        u.hs = hmax / comp[k].h; u.vs = vmax / comp[k].v; u.ystep = u.vs >> 1; u.ypos = 0; u.wlo = (width + u.hs - 1) / u.hs;
        u.l0 = u.l1 = comp[k].data.data();
        u.buf.assign((size_t)width + 16, 0);
    }
    const uint8_t* rows[3] = { nullptr, nullptr, nullptr };
    for (int j = 0; j < height; j++) {
        for (int k = 0; k < nc; k++) {
            Up& u = up[k];
            const bool bottom = u.ystep >= (u.vs >> 1);
            rows[k] = upsample_row(u.buf.data(), bottom ? u.l1 : u.l0, bottom ? u.l0 : u.l1, u.wlo, u.hs, u.vs);
            if (++u.ystep >= u.vs) {
                u.ystep = 0; u.l0 = u.l1;
                if (++u.ypos < comp[k].y) u.l1 += comp[k].w2;
            }
        }
        uint8_t* out = pix.data() + (size_t)j * width * nCh;
        if (nc == 1) memcpy(out, rows[0], (size_t)width);
        else if (isRgb) for (int i = 0; i < width; i++) { out[i * 3] = rows[0][i]; out[i * 3 + 1] = rows[1][i]; out[i * 3 + 2] = rows[2][i]; }
        else ycbcr_row(out, rows[0], rows[1], rows[2], width);
    }
    return true;
}

inline bool decode(const std::string& path, std::vector<unsigned char>& pix, int& w, int& h, int& nCh)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> d;
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    return decode_memory(d.data(), d.size(), pix, w, h, nCh);
}

} // namespace jpeg
} // namespace rth
