// Hardware probe (not part of the product): LDS atomic exchange with return (ds_wrxchg_rtn_b32).
//  test 1: lanes of one wave instruction that hit the same address -- is the exchange resolved in ascending lane
//          order (lane l gets back what the nearest lower lane with the same address wrote)?
//  test 2: cost per wave instruction of {xchg with return} vs {read + write} on random addresses (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe_order(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t t[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *T = t + wave * 2048;                 // each wavefront its own 2048 slots
    for (int i = lane; i < 2048; i += 64) T[i] = 0xFFFFFFFFu;
    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
    uint32_t bad = 0, chains = 0;
    uint32_t x = seed + blockIdx.x * 977 + wave * 131;
    for (int it = 0; it < iters; it++) {
        // address pattern: a wave-uniform modulus picks how crowded the addresses are
        x = x * 1664525u + 1013904223u;
        const uint32_t mod = 1u + ((x >> 8) % 96u);                    // 1 .. 96 distinct addresses
        const uint32_t a = ((uint32_t)lane * 2654435761u + x) >> 7;    // scrambled
        const uint32_t addr = (a % mod) * 17u % 2048u;
        const uint32_t val = ((uint32_t)it << 6) | (uint32_t)lane;
        const uint32_t old = atomicExch(&T[addr], val);
        // expected: nearest lower lane with the same address in this instruction; else "some earlier iteration's / initial" value
        int prev = -1;
        for (int l = lane - 1; l >= 0; l--) {
            const uint32_t al = ((uint32_t)l * 2654435761u + x) >> 7;
            if ((al % mod) * 17u % 2048u == addr) { prev = l; break; }
        }
        if (prev >= 0) { chains++; if (old != (((uint32_t)it << 6) | (uint32_t)prev)) bad++; }
        else if (old != 0xFFFFFFFFu && (old >> 6) >= (uint32_t)it) bad++;      // must come from an earlier iteration
        __builtin_amdgcn_wave_barrier();
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = bad;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = chains;
}

template <int MODE>
__global__ void probe_rate(uint64_t *cyc, uint32_t *sink, int iters, int slotsLog)
{
    extern __shared__ uint32_t tt[];
    const int lane = threadIdx.x & 63;
    const uint32_t slots = 1u << slotsLog;
    for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) tt[i] = i;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t h[8], o[8];
        #pragma unroll
        for (int u = 0; u < 8; u++) { x = x * 1664525u + 1013904223u; h[u] = x >> (32 - slotsLog); }
        #pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) o[u] = atomicExch(&tt[h[u]], x + u);
            else { o[u] = tt[h[u]]; tt[h[u]] = x + u; }
        }
        #pragma unroll
        for (int u = 0; u < 8; u++) acc += o[u];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    uint32_t *d; const int blocks = 512, threads = 256;
    hipMalloc(&d, blocks * threads * 2 * 4);
    uint64_t badTot = 0, chainTot = 0;
    for (int rep = 0; rep < 8; rep++) {
        probe_order<<<blocks, threads>>>(d, 4000, 12345u + rep * 7919u);
        std::vector<uint32_t> h(blocks * threads * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h.size(); i += 2) { badTot += h[i]; chainTot += h[i + 1]; }
    }
    printf("xchg order: %llu violations in %llu same-address successor checks\n", (unsigned long long)badTot, (unsigned long long)chainTot);
    uint64_t *c; uint32_t *s; hipMalloc(&c, 4096 * 8); hipMalloc(&s, 512 * 1024 * 4);
    for (int waves = 1; waves <= 8; waves *= 2) for (int mode = 0; mode < 2; mode++) {
        const int iters = 2000, slotsLog = 13;
        hipMemset(c, 0, 4096 * 8);
        if (mode == 0) probe_rate<0><<<256, waves * 64, 4u << slotsLog>>>(c, s, iters, slotsLog);
        else probe_rate<1><<<256, waves * 64, 4u << slotsLog>>>(c, s, iters, slotsLog);
        std::vector<uint64_t> h(256 * waves);
        hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        // s_memtime ticks at 100 MHz on MI300-class parts: report ticks and wall-clock separately
        printf("%s, %d wavefronts per workgroup (1 per CU): %.2f memtime ticks per wave instruction slot (8 per iteration)\n", mode == 0 ? "xchg_rtn" : "read+write", waves, sum / h.size() / iters / 8);
    }
    // wall-clock: 256 workgroups x 4 waves x iters x 8 instructions
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        hipEventRecord(e0);
        if (mode == 0) probe_rate<0><<<256, 256, 4u << 13>>>(c, s, iters, 13); else probe_rate<1><<<256, 256, 4u << 13>>>(c, s, iters, 13);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s wall: %.3f ms for %d x 8 wave instructions per wavefront, 4 wavefronts per CU -> %.1f ns per instruction per wavefront\n", mode == 0 ? "xchg_rtn" : "read+write", ms, iters, ms * 1e6 / iters / 8);
    }
    return 0;
}
