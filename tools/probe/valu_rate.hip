// Hardware probe (development aid): how many cycles a SIMD spends per wave64 vector instruction with 1, 2, 4, 8 wavefronts on it
// (independent integer adds / xors, no memory): the issue floor behind "vector instructions x cycles" estimates in DESIGN.md.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/valu_rate tools/probe/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(uint32_t *out, unsigned long long *ticks, int iters)
{
    uint32_t a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 7u + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            #pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 5) & 15]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t s = 0; for (int i = 0; i < 16; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main()
{
    uint32_t *out; unsigned long long *ticks;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&ticks, 256 * 32 * 8);
    const int iters = 2000;                                   // 2000 x 4 x 16 x 2 = 256000 vector instructions per wavefront
    for (int wavesPerSimd : { 1, 2, 4, 8 }) {
        const int threads = wavesPerSimd * 4 * 64;            // one workgroup per CU, its wavefronts spread over the 4 SIMDs
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, ticks, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        const double instr = (double)iters * 4 * 16 * 2;
        printf("%d wavefront(s) per SIMD: %.2f s_memtime ticks per instruction per wavefront, %.2f ticks of SIMD time per instruction (wall %.3f ms: %.2f ns per instruction and SIMD)\n",
               wavesPerSimd, avg / instr, avg / instr / wavesPerSimd, ms, ms * 1e6 / (instr * wavesPerSimd));
    }
    return 0;
}
