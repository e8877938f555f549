// Hardware probe (development aid): does an LDS-DMA load (global_load_lds_dwordx4, M0 = LDS destination of lane 0) land in the issuing
// workgroup's own LDS allocation when two workgroups share a CU, and does "s_waitcnt vmcnt(N)" cover it in issue order with a younger store?
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/lds_dma tools/probe/lds_dma.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int WAITN>
__global__ void __launch_bounds__(512) k(const uint4 *__restrict__ g, uint32_t *__restrict__ bad, uint32_t *__restrict__ sink, uint32_t ldsBytes, uint32_t xoff, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // signature over the whole allocation
    for (uint32_t i = tid; i < ldsBytes / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = 0xA5000000u ^ (blockIdx.x << 12) ^ i;
    __syncthreads();
    uint4 *xb = reinterpret_cast<uint4 *>(lds + xoff) + wave * 64;
    typedef __attribute__((address_space(3))) uint4 *L4;
    const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(L4)xb);
    uint32_t nbad = 0;
    for (int it = 0; it < iters; it++) {
        const uint4 *src = g + ((size_t)blockIdx.x * 131 + (size_t)it * 977 + wave * 64 + lane) % (1u << 20);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        sink[(size_t)blockIdx.x * 512 + tid] = it;                                   // a younger store
        if (WAITN == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 got = xb[lane];
        const uint32_t idx = (uint32_t)(((size_t)blockIdx.x * 131 + (size_t)it * 977 + wave * 64 + lane) % (1u << 20));
        if (got.x != idx * 4u || got.y != idx * 4u + 1 || got.z != idx * 4u + 2 || got.w != idx * 4u + 3) nbad++;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // everything outside the exchange area must still carry the signature
    uint32_t nsig = 0;
    for (uint32_t i = tid; i < ldsBytes / 4; i += 512) {
        if (i * 4 >= xoff && i * 4 < xoff + 8192) continue;
        if (reinterpret_cast<uint32_t *>(lds)[i] != (0xA5000000u ^ (blockIdx.x << 12) ^ i)) nsig++;
    }
    if (nbad) atomicAdd(&bad[0], nbad);
    if (nsig) atomicAdd(&bad[1], nsig);
}

int main()
{
    const size_t N = 1u << 20;
    std::vector<uint32_t> h(N * 4);
    for (size_t i = 0; i < N * 4; i++) h[i] = (uint32_t)i;
    uint4 *g; uint32_t *bad, *sink;
    hipMalloc(&g, N * 16); hipMalloc(&bad, 8); hipMalloc(&sink, 4096ull * 512 * 4);
    hipMemcpy(g, h.data(), N * 16, hipMemcpyHostToDevice);
    const uint32_t ldsBytes = 79040;
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    for (int waitn = 0; waitn < 2; waitn++)
        for (uint32_t xoff : { 0u, 70016u }) {
            for (int grid : { 128, 2048 }) {
                uint32_t z[2] = { 0, 0 };
                hipMemcpy(bad, z, 8, hipMemcpyHostToDevice);
                if (waitn) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), ldsBytes, 0, g, bad, sink, ldsBytes, xoff, 200);
                else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), ldsBytes, 0, g, bad, sink, ldsBytes, xoff, 200);
                hipError_t e = hipDeviceSynchronize();
                hipMemcpy(z, bad, 8, hipMemcpyDeviceToHost);
                printf("wait vmcnt(%d) xoff %6u grid %5d : %s  wrong pieces %u  foreign LDS words changed %u\n", waitn, xoff, grid, hipGetErrorString(e), z[0], z[1]);
            }
        }
    return 0;
}
