// Hardware probe (not part of the product): does LDS resolve same-address accesses of one
// wave instruction in ascending lane order?
//  test 1: ds_write_b16, all 64 lanes to the same address (and to a few shared addresses): which lane's value stays?
//  test 2: ds_max_rtn_u32 same address, values ascending with lane: does lane l get lane l-1's value back?
//  test 3: unaligned 4-byte LDS reads and unaligned global 4/8-byte loads return the right bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(uint32_t *out, const uint8_t *gsrc, int groups)
{
    __shared__ uint16_t t16[256];
    __shared__ uint32_t t32[256];
    __shared__ __attribute__((aligned(16))) uint8_t bytes[512];
    int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { t16[i] = 0; t32[i] = 0; }
    for (int i = threadIdx.x; i < 512; i += blockDim.x) bytes[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t bad1 = 0, bad2 = 0, bad3 = 0;
    for (int iter = 0; iter < 1000; iter++) {
        // lanes share addresses in groups: addr = lane % groups  -> highest lane with that residue should win
        int addr = (lane * 37 + iter) % groups;           // scrambled sharing pattern
        t16[addr] = (uint16_t)(iter * 64 + lane + 1);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        uint16_t got = t16[addr];
        // expected winner: highest lane l' with same addr
        int win = -1;
        for (int l = 63; l >= 0; l--) if (((l * 37 + iter) % groups) == addr) { win = l; break; }
        if (got != (uint16_t)(iter * 64 + win + 1)) bad1++;
        __builtin_amdgcn_wave_barrier();
        // atomic max with return
        uint32_t val = (uint32_t)(iter * 64 + lane + 1);
        uint32_t old = atomicMax(&t32[addr], val);
        int prev = -1;
        for (int l = lane - 1; l >= 0; l--) if (((l * 37 + iter) % groups) == addr) { prev = l; break; }
        uint32_t expect_old;
        if (prev >= 0) expect_old = (uint32_t)(iter * 64 + prev + 1);
        else {   // last value of previous iteration at this address (or 0)
            expect_old = 0;
            for (int it2 = iter - 1; it2 >= 0 && !expect_old; it2--)
                for (int l = 63; l >= 0; l--) if (((l * 37 + it2) % groups) == addr) { expect_old = (uint32_t)(it2 * 64 + l + 1); break; }
        }
        if (old != expect_old) bad2++;
        __builtin_amdgcn_wave_barrier();
    }
    // unaligned reads
    for (int off = 0; off < 8; off++) {
        uint32_t v; __builtin_memcpy(&v, &bytes[lane * 5 + off], 4);
        uint32_t e = 0; for (int k = 0; k < 4; k++) e |= (uint32_t)bytes[lane * 5 + off + k] << (8 * k);
        if (v != e) bad3++;
        uint32_t g; __builtin_memcpy(&g, gsrc + lane * 3 + off, 4);
        uint32_t ge = 0; for (int k = 0; k < 4; k++) ge |= (uint32_t)gsrc[lane * 3 + off + k] << (8 * k);
        if (g != ge) bad3 += 100;
        uint64_t g8; __builtin_memcpy(&g8, gsrc + lane * 9 + off, 8);
        uint64_t g8e = 0; for (int k = 0; k < 8; k++) g8e |= (uint64_t)gsrc[lane * 9 + off + k] << (8 * k);
        if (g8 != g8e) bad3 += 10000;
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 3 + 0] = bad1;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 3 + 1] = bad2;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 3 + 2] = bad3;
}

int main()
{
    const int blocks = 512, threads = 64;
    uint32_t *d; uint8_t *g;
    hipMalloc(&d, blocks * threads * 3 * 4); hipMalloc(&g, 4096);
    std::vector<uint8_t> h(4096); for (int i = 0; i < 4096; i++) h[i] = (uint8_t)(i * 13 + 5);
    hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
    for (int groups : {1, 2, 7, 16, 33, 64}) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, g, groups);
        std::vector<uint32_t> r(blocks * threads * 3);
        hipMemcpy(r.data(), d, r.size() * 4, hipMemcpyDeviceToHost);
        uint64_t b1 = 0, b2 = 0, b3 = 0;
        for (size_t i = 0; i < r.size(); i += 3) { b1 += r[i]; b2 += r[i + 1]; b3 += r[i + 2]; }
        printf("groups=%2d  write16_highest_lane_wins_violations=%llu  atomicmax_rtn_laneorder_violations=%llu  unaligned_bad=%llu\n",
               groups, (unsigned long long)b1, (unsigned long long)b2, (unsigned long long)b3);
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  sharedMemPerBlock %zu  maxSharedOptin %zu  l2 %d\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock, (size_t)p.sharedMemPerBlockOptin, p.l2CacheSize);
    return 0;
}
