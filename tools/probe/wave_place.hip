// Probe: where do the wavefronts of a 4-wavefront workgroup land?  (SIMD of wavefront 0 of the workgroups sharing a CU)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/wave_place tools/probe/wave_place.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t spin)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident long enough for the whole round to be placed
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < spin; i++) { x = x * 1664525u + 1013904223u; lds[threadIdx.x] = x; __syncthreads(); x += lds[(threadIdx.x + 1) & 255]; }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wave) * 2] = hw; out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc | (x == 12345u ? 1u << 31 : 0u); }
}
int main(int argc, char **argv)
{
    const uint32_t nwg = argc > 1 ? atoi(argv[1]) : 1024, ldsBytes = argc > 2 ? atoi(argv[2]) : 38 * 1024;
    uint32_t *d; hipMalloc(&d, nwg * 4 * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), ldsBytes, 0, d, 20000u);
    std::vector<uint32_t> h(nwg * 8); hipMemcpy(h.data(), d, nwg * 32, hipMemcpyDeviceToHost);
    // per CU: list of (workgroup, simd of wavefront 0..3)
    std::map<uint32_t, std::vector<std::pair<uint32_t, uint32_t>>> cu;
    for (uint32_t b = 0; b < nwg; b++) {
        uint32_t simds = 0;
        for (int w = 0; w < 4; w++) simds |= ((h[(b * 4 + w) * 2] >> 4) & 3u) << (4 * w);
        const uint32_t hw = h[b * 8], xcc = h[b * 8 + 1] & 0xF;
        const uint32_t key = (xcc << 16) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 0xFu);
        cu[key].push_back({ b, simds });
    }
    printf("%zu CUs hold the %u workgroups\n", cu.size(), nwg);
    int shown = 0; uint32_t hist[5] = {0,0,0,0,0}, sameOrder = 0;
    for (auto &e : cu) {
        uint32_t seen = 0;
        for (auto &p : e.second) { seen |= 1u << (p.second & 3u); if (p.second == 0x3210) sameOrder++; }
        hist[__builtin_popcount(seen)]++;
        if (shown++ < 6) { printf("xcc %u se %u sh %u cu %2u:", e.first >> 16, (e.first >> 8) & 7, (e.first >> 4) & 1, e.first & 15); for (auto &p : e.second) printf("  wg %4u simd(w0..w3)=%u%u%u%u", p.first, p.second & 3, (p.second >> 4) & 3, (p.second >> 8) & 3, (p.second >> 12) & 3); printf("\n"); }
    }
    printf("distinct SIMDs holding a wavefront 0 per CU: 1:%u 2:%u 3:%u 4:%u CUs; workgroups with wavefront w on SIMD w: %u of %u\n", hist[1], hist[2], hist[3], hist[4], sameOrder, nwg);
    return 0;
}
