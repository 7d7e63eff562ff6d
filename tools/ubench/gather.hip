// Microbenchmark: cost model of divergent per-lane record fetches on gfx950 (decides the BVH node
// format).  Every lane walks a dependent chain of random records of R bytes inside a table of T
// bytes; variants differ in how the record is fetched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// V=0: own record, R/16 dwordx4 loads per lane.  V=1: 4 lanes share one 64-B record per round, 4 rounds
// (16 distinct lines per instruction), data handed over through LDS.
template <int R, int V>
__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ table, uint32_t nRec, int iters, int activeLanes,
                                            uint32_t* __restrict__ out)
{
    __shared__ uint4 stage[256 * 4];
    const int lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u % nRec;
    uint32_t acc = 0;
    if (lane >= activeLanes) { out[blockIdx.x * 256 + threadIdx.x] = 0; return; }
    for (int it = 0; it < iters; it++) {
        if (V == 0) {
            const uint4* p = table + (size_t)idx * (R / 16);
            uint4 v[R / 16];
#pragma unroll
            for (int k = 0; k < R / 16; k++) v[k] = p[k];
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < R / 16; k++) h += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
            acc += h;
            idx = (h ^ (idx * 747796405u + 2891336453u)) % nRec;
        } else {
            // round r: lanes 4g..4g+3 fetch the record wanted by lane (g + 16 r) of the wave
            const int g = lane >> 2, c = lane & 3;
            uint4* st = stage + (threadIdx.x & ~63) * 4;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t want = __shfl(idx, g + 16 * r);
                const uint4 v = table[(size_t)want * 4 + c];
                st[(g + 16 * r) * 4 + c] = v;
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint4 v = st[lane * 4 + k]; h += v.x ^ v.y ^ v.z ^ v.w; }
            __builtin_amdgcn_wave_barrier();
            acc += h;
            idx = (h ^ (idx * 747796405u + 2891336453u)) % nRec;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int R, int V>
void run(const char* name, const uint4* table, size_t tableBytes, int wgPerCU, int activeLanes, uint32_t* out)
{
    const uint32_t nRec = (uint32_t)(tableBytes / R);
    const int iters = 2000;
    const int grid = 256 * wgPerCU;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((walk<R, V>), dim3(grid), dim3(256), 0, 0, table, nRec, 200, activeLanes, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk<R, V>), dim3(grid), dim3(256), 0, 0, table, nRec, iters, activeLanes, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)grid * 4 * activeLanes * iters;
    const double waveIters = (double)grid * 4 * iters;
    // cycles per wave-iteration per CU at 2.4 GHz: time * f / (wave-iterations per CU)
    printf("%-28s table %6.0f MB  wg/CU %d lanes %2d : %7.2f ms  %8.1f Grec/s  %6.1f GB/s  %6.0f cyc/wave-iter/CU\n", name,
           tableBytes / 1e6, wgPerCU, activeLanes, ms, recs / ms / 1e6, recs * R / ms / 1e6, ms * 1e-3 * 2.4e9 / (waveIters / 256));
}

int main()
{
    const size_t maxBytes = 512ull << 20;
    uint4* table; CK(hipMalloc(&table, maxBytes));
    std::vector<uint32_t> h(maxBytes / 4);
    uint32_t s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s; }
    CK(hipMemcpy(table, h.data(), maxBytes, hipMemcpyHostToDevice));
    uint32_t* out; CK(hipMalloc(&out, 256 * 16 * 256 * 4));
    for (size_t tb : { (size_t)2 << 20, (size_t)64 << 20, (size_t)512 << 20 }) {
        for (int wg : { 3, 6 }) {
            run<16, 0>("own 16B (1 x b128)", table, tb, wg, 64, out);
            run<32, 0>("own 32B (2 x b128)", table, tb, wg, 64, out);
            run<64, 0>("own 64B (4 x b128)", table, tb, wg, 64, out);
            run<128, 0>("own 128B (8 x b128)", table, tb, wg, 64, out);
            run<64, 0>("own 64B, 24 lanes", table, tb, wg, 24, out);
            run<128, 0>("own 128B, 24 lanes", table, tb, wg, 24, out);
            run<64, 1>("shared 64B via LDS", table, tb, wg, 64, out);
        }
    }
    return 0;
}
