// Host check of the launch-constant divisions of the wavefront pipeline (csrc/rt_device.h: make_fast_div, fast_div):
// the constants make_fast_div produces, with the device formula restated on 64-bit host integers (__umulhi = high word of
// the 32 x 32 product), against n / d.  Built and run by tests/test_host.py.
#include "../../raytracer_amd/csrc/rt_device.h"
#include <cstdio>
#include <cstdint>
#include <cstdlib>

static uint32_t host_fast_div(uint32_t n, const rt::FastDiv& f)
{
    const uint32_t t = (uint32_t)(((unsigned long long)f.m * n) >> 32);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

int main(int argc, char** argv)
{
    const uint32_t maxD = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 10) : 200000u;
    unsigned long long checked = 0, bad = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); };
    auto check = [&](uint32_t n, uint32_t d, const rt::FastDiv& f) {
        checked++;
        if (host_fast_div(n, f) != n / d) { if (bad++ < 5) printf("mismatch n=%u d=%u\n", n, d); }
    };
    for (uint32_t d = 1; d <= maxD; d++) {
        const rt::FastDiv f = rt::make_fast_div(d);
        const uint32_t edge[] = { 0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu,
                                  0xFFFFFFFFu / d * d, 0xFFFFFFFFu / d * d - 1u };
        for (uint32_t n : edge) check(n, d, f);
        for (int k = 0; k < 24; k++) check(next(), d, f);
    }
    // divisors up to 2^32 - 1, among them every power of two and its neighbours
    for (int b = 1; b < 32; b++)
        for (int o = -1; o <= 1; o++) {
            const uint32_t d = (1u << b) + (uint32_t)o;
            const rt::FastDiv f = rt::make_fast_div(d);
            for (int k = 0; k < 4096; k++) check(next(), d, f);
            check(0xFFFFFFFFu, d, f); check(d, d, f); check(d - 1u, d, f);
        }
    for (int k = 0; k < 2000000; k++) {
        uint32_t d = next() >> (next() & 31u); if (d == 0u) d = 1u;
        check(next(), d, rt::make_fast_div(d));
    }
    printf("checked %llu bad %llu\n", checked, bad);
    return bad ? 1 : 0;
}
