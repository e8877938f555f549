// Hardware probe (not part of the product): are unaligned 2/4/8-byte LDS STORES correct on gfx950 (as the loads are)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(uint32_t *bad, uint32_t *detail)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[64 * 40 + 64];
    const int lane = threadIdx.x;
    uint32_t nbad = 0;
    for (int off = 0; off < 8; off++) for (int width = 2; width <= 8; width *= 2) {
        for (int i = lane; i < 64 * 40 + 64; i += 64) buf[i] = 0xEE;
        __syncthreads();
        uint8_t *p = buf + lane * 40 + off + (lane & 3);           // every alignment
        const uint64_t v = 0x0807060504030201ull * (uint64_t)(lane + 1);
        if (width == 2) { uint16_t x = (uint16_t)v; __builtin_memcpy(p, &x, 2); }
        else if (width == 4) { uint32_t x = (uint32_t)v; __builtin_memcpy(p, &x, 4); }
        else { __builtin_memcpy(p, &v, 8); }
        __syncthreads();
        for (int k = -2; k < width + 2; k++) {
            const uint8_t want = (k >= 0 && k < width) ? (uint8_t)(v >> (8 * k)) : 0xEE;
            if (p[k] != want && (p + k) >= buf + 0) { if (nbad < 4) detail[lane * 4 + nbad] = (uint32_t)width | ((uint32_t)((p - buf) & 15) << 8) | ((uint32_t)(k & 0xFF) << 16) | ((uint32_t)p[k] << 24); nbad++; }
        }
        __syncthreads();
    }
    bad[lane] = nbad;
}
int main()
{
    uint32_t *d, *dd; hipMalloc(&d, 64 * 4); hipMalloc(&dd, 64 * 16); hipMemset(dd, 0, 64 * 16);
    probe<<<1, 64>>>(d, dd);
    std::vector<uint32_t> hd(256); hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; i++) if (hd[i]) printf("lane %d: width %u addr%%16=%u k=%d got %02x\n", i / 4, hd[i] & 0xFF, (hd[i] >> 8) & 0xFF, (int)(int8_t)((hd[i] >> 16) & 0xFF), hd[i] >> 24);
    std::vector<uint32_t> h(64); hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
    uint32_t tot = 0; for (auto v : h) tot += v;
    printf("unaligned LDS stores of 2/4/8 bytes at every alignment: %u wrong bytes\n", tot);
    return 0;
}
