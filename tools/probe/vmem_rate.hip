// Hardware probe (development aid): what a vector-memory instruction costs a CU when the data sits in L2 / L1 (the decode
// execute kernel's regime: short copies at byte-granular addresses inside a few KiB of recent output).  24 wavefronts a CU,
// each issuing loads (or load + store pairs) of 1 / 4 / 8 / 16 bytes a lane at unaligned, lane-scattered addresses inside its
// own 2 KiB window, with 64 / 16 / 4 / 1 active lanes.  Reports ns of CU time per wave64 instruction.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/vmem_rate tools/probe/vmem_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int WIDTH, bool STORE>
__global__ void __launch_bounds__(256) k(uint8_t *buf, uint32_t *out, int iters, uint32_t activeLanes, uint32_t aligned)
{
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    uint8_t *w = buf + wave * 4096;                                      // 2 KiB read window + 2 KiB written: a CU's 24 wavefronts stay inside L2
    uint32_t pos = (lane * 97u + 5u) & 0x7FFu;                         // a lane's own spot: 97 bytes apart, any alignment
    if (aligned) pos &= ~15u;
    uint64_t acc = 0;
    if (lane < activeLanes) {
        for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint8_t *p = w + ((pos + r * 1031u + it * 13u * (aligned ? 16u : 1u)) & 0x7FFu);
                if (WIDTH == 1) { acc += *p; if (STORE) w[2048 + ((p - w) & 0x7FF)] = (uint8_t)acc; }
                else if (WIDTH == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); acc += v; if (STORE) __builtin_memcpy(w + 2048 + ((p - w) & 0x7FF), &v, 4); }
                else if (WIDTH == 8) { uint64_t v; __builtin_memcpy(&v, p, 8); acc += v; if (STORE) __builtin_memcpy(w + 2048 + ((p - w) & 0x7FF), &v, 8); }
                else { uint64_t v[2]; __builtin_memcpy(v, p, 16); acc += v[0] ^ v[1]; if (STORE) __builtin_memcpy(w + 2048 + ((p - w) & 0x7FF), v, 16); }
            }
        }
    }
    out[wave * 64 + lane] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

template <int WIDTH, bool STORE>
static void run(uint8_t *buf, uint32_t *out, const char *what)
{
    const int iters = 400, wgs = 256 * 6;                                 // 6 workgroups of 4 wavefronts a CU
    for (uint32_t aligned : { 0u, 1u }) {
        printf("%s %s:", what, aligned ? "16-byte aligned" : "any alignment  ");
        for (uint32_t act : { 64u, 16u, 4u, 1u }) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL((k<WIDTH, STORE>), dim3(wgs), dim3(256), 0, 0, buf, out, 5, act, aligned);
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<WIDTH, STORE>), dim3(wgs), dim3(256), 0, 0, buf, out, iters, act, aligned);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)iters * 8 * (STORE ? 2 : 1) * 24;    // wave instructions a CU
            printf("  %2u lanes: %7.2f ns", act, ms * 1e6 / instr);
        }
        printf("   (per wave64 instruction and CU)\n");
    }
}

int main()
{
    uint8_t *buf; uint32_t *out;
    const size_t waves = 256 * 6 * 4;
    hipMalloc(&buf, waves * 4096 + 64); hipMemset(buf, 1, waves * 4096 + 64);
    hipMalloc(&out, waves * 64 * 4);
    run<1, false>(buf, out, "load  1 B ");
    run<4, false>(buf, out, "load  4 B ");
    run<8, false>(buf, out, "load  8 B ");
    run<16, false>(buf, out, "load 16 B ");
    run<1, true>(buf, out, "copy  1 B ");
    run<8, true>(buf, out, "copy  8 B ");
    run<16, true>(buf, out, "copy 16 B ");
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "no error" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
