// Hardware probe (development aid): what an unaligned LDS access costs.  A CU's wavefronts read / write 8 bytes a lane at
// byte address 8 * slot + mis (mis = 0: aligned; 1..7: every lane straddles two dwords / banks), slots a lane apart or spread;
// reports ns of CU time per wave64 LDS instruction.  (Behind the choice of an LDS tile buffer with byte-placed copies in k_dec_execute.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/lds_unaligned_rate tools/probe/lds_unaligned_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int WIDTH, bool STORE>
__global__ void k(uint32_t *out, int iters, uint32_t mis, uint32_t stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[16 * 1024 + 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < sizeof(buf) / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(buf)[i] = i * 2654435761u;
    __syncthreads();
    typedef __attribute__((address_space(3))) uint8_t *L8;
    uint32_t addr = (uint32_t)(uintptr_t)(L8)buf + ((lane * stride + wave * 520u) & 0x3FF8u) + mis;
    uint64_t acc = lane;
    uint32_t acc4 = lane;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            if (WIDTH == 8) {
                if (STORE) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(acc) : "memory");
                else { uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); acc ^= v; }
            } else {
                if (STORE) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(acc4) : "memory");
                else { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); acc4 ^= v; }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ acc4;
}

template <int WIDTH, bool STORE>
static void run(uint32_t *out, const char *what)
{
    const int iters = 2000, threads = 1024;                     // 16 wavefronts a CU, one workgroup a CU
    for (uint32_t stride : { 8u, 40u }) {
        printf("%s, slots %2u bytes apart:", what, stride);
        for (uint32_t mis : { 0u, 1u, 3u, 4u, 6u }) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL((k<WIDTH, STORE>), dim3(256), dim3(threads), 0, 0, out, 10, mis, stride);
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<WIDTH, STORE>), dim3(256), dim3(threads), 0, 0, out, iters, mis, stride);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  mis %u: %6.2f ns", mis, ms * 1e6 / ((double)iters * 16 * (threads / 64)));
        }
        printf("   (per wave64 instruction and CU)\n");
    }
}

int main()
{
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4);
    run<8, false>(out, "ds_read_b64 ");
    run<8, true>(out, "ds_write_b64");
    run<4, false>(out, "ds_read_b32 ");
    run<4, true>(out, "ds_write_b32");
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "no error" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
