// Entropy stage of the block encoder: the sequences left by k_lz_walk become one Zstandard compressed block per 64 KiB block.
//   k_encode_sequences  (4 blocks = 4 wavefronts per workgroup)  -> sequences section: recent-offset codes, histograms, FSE tables, bitstream
//   k_encode_literals   (1 block, 4 wavefronts per workgroup)    -> literals section: gather, Huffman lengths / codes / description / streams;
//                                                                    writes the frame of a one-block chunk as its workgroup finishes
//   k_assemble_frames   (1 chunk per workgroup)                  -> frames of chunks of several blocks
// Scalar statement of the same algorithm: oracle/zso_encoder.c (compressBlock and below); the two
// must agree bit for bit.  Every piece is the format-inverse of a function of the reference decoder:
//   literals section        <-> DecodeLiteralsBlock            csharp/src/ZStdDecompress.cs:683-821
//   Huffman table / streams <-> ReadStats, HUF_readDTableX2,   EntropyCommon.cs:198-269, HufDecompress.cs:117-358
//   sequences header/tables <-> DecodeSeqHeaders, BuildFSETable ZStdDecompress.cs:958-1180, EntropyCommon.cs:79-188
//   sequences bitstream     <-> DecodeSequence, decompressSequences_body  ZStdDecompress.cs:1473-1608
#include "zsmi_device.h"
// timing aids of the development tools (tools/time_kernels.py): end a kernel after a stage.  Compiled in only with
// -DZSMI_DEBUG_HOOKS (the library the product ships ignores the stopAt argument).
#ifdef ZSMI_DEBUG_HOOKS
#define ZS_STOP_AT(v) (stopAt == (v))
#define ZS_STOPPED (stopAt != 0)     // a stopped kernel writes no frame (its stage time is what is asked for)
#else
#define ZS_STOP_AT(v) false
#define ZS_STOPPED false
#endif

#define MaxLL 35
#define MaxML 52
#define MaxOff 31

// ---- constant tables (ZStdInternal.cs:158-192, ZStdDecompress.cs:1081-1105) ----
__constant__ uint8_t c_LL_bits[36] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
__constant__ uint8_t c_ML_bits[53] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
                                      1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
__constant__ uint32_t c_LL_base[36] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,18,20,22,24,28,32,40,
                                       48,64,0x80,0x100,0x200,0x400,0x800,0x1000, 0x2000,0x4000,0x8000,0x10000 };
__constant__ uint32_t c_ML_base[53] = { 3,4,5,6,7,8,9,10, 11,12,13,14,15,16,17,18, 19,20,21,22,23,24,25,26,
                                       27,28,29,30,31,32,33,34, 35,37,39,41,43,47,51,59, 67,83,99,0x83,0x103,0x203,0x403,0x803,
                                       0x1003,0x2003,0x4003,0x8003,0x10003 };
__constant__ int16_t c_LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
__constant__ int16_t c_ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                             1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
__constant__ int16_t c_OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
__constant__ uint8_t c_LL_Code[64] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,16,17,17,18,18,19,19, 20,20,20,20,21,21,21,21,
                                      22,22,22,22,22,22,22,22, 23,23,23,23,23,23,23,23, 24,24,24,24,24,24,24,24, 24,24,24,24,24,24,24,24 };
__constant__ uint8_t c_ML_Code[128] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,17,18,19,20,21,22,23, 24,25,26,27,28,29,30,31,
                                       32,32,33,33,34,34,35,35, 36,36,36,36,37,37,37,37, 38,38,38,38,38,38,38,38, 39,39,39,39,39,39,39,39,
                                       40,40,40,40,40,40,40,40, 40,40,40,40,40,40,40,40, 41,41,41,41,41,41,41,41, 41,41,41,41,41,41,41,41,
                                       42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42 };

// Codes and extra bits by arithmetic: the same values as the tables above (LL_Code / ML_Code, LL_base / LL_bits, ML_base /
// ML_bits; checked value by value for every length up to 2^17), without a gather from constant memory in the middle of a
// tile (a memory round trip each, and its wait takes the loop's prefetch along).
__device__ __forceinline__ uint32_t llCodeOf(uint32_t ll)
{
    uint32_t c = zs_highbit(ll | 1u) + 19;                        // ll >= 64
    c = (ll < 64) ? 24u : c;
    c = (ll < 48) ? 22u + ((ll - 32) >> 3) : c;
    c = (ll < 32) ? 20u + ((ll - 24) >> 2) : c;
    c = (ll < 24) ? 16u + ((ll - 16) >> 1) : c;
    return (ll < 16) ? ll : c;
}
__device__ __forceinline__ uint32_t mlCodeOf(uint32_t x)         // x = match length - 3
{
    uint32_t c = zs_highbit(x | 1u) + 36;                         // x >= 128
    c = (x < 128) ? 42u : c;
    c = (x < 96) ? 40u + ((x - 64) >> 4) : c;
    c = (x < 64) ? 38u + ((x - 48) >> 3) : c;
    c = (x < 48) ? 36u + ((x - 40) >> 2) : c;
    c = (x < 40) ? 32u + ((x - 32) >> 1) : c;
    return (x < 32) ? x : c;
}
// extra bits of a literal length / of a match length (x = length - 3): value and count
__device__ __forceinline__ uint32_t llExtraOf(uint32_t ll, uint32_t &bits)
{
    const uint32_t h = zs_highbit(ll | 1u);
    uint32_t v = ll - (1u << h), b = h;                            // ll >= 64
    if (ll < 64) { v = ll - 48; b = 4; }
    if (ll < 48) { v = ll & 7u; b = 3; }
    if (ll < 32) { v = ll & 3u; b = 2; }
    if (ll < 24) { v = ll & 1u; b = 1; }
    if (ll < 16) { v = 0; b = 0; }
    bits = b; return v;
}
__device__ __forceinline__ uint32_t mlExtraOf(uint32_t x, uint32_t &bits)
{
    const uint32_t h = zs_highbit(x | 1u);
    uint32_t v = x - (1u << h), b = h;                             // x >= 128
    if (x < 128) { v = x - 96; b = 5; }
    if (x < 96) { v = x & 15u; b = 4; }
    if (x < 64) { v = x & 7u; b = 3; }
    if (x < 48) { v = x & 3u; b = 2; }
    if (x < 40) { v = x & 1u; b = 1; }
    if (x < 32) { v = 0; b = 0; }
    bits = b; return v;
}

// ---------------------------------------------------------------------------------------------
// per-wavefront LDS workspace
// ---------------------------------------------------------------------------------------------
struct FseCT {                       // encoding table of one symbol type
    uint16_t stateTable[512];
    int32_t  deltaFindState[64];
    uint32_t deltaNbBits[64];
    uint32_t tableLog;
    uint32_t rle;
};
// workgroups per CU the literals kernel is compiled for: 8 caps it at 64 VGPRs, so the 16 blocks a CU gets per 4096-block
// batch run in two full rounds (measured alone: 0.89 ms uncapped (88 VGPRs, 5 per CU), 0.74 at 6, 0.57 at 8 despite spills)
#ifndef ZS_LIT_MINWG
#define ZS_LIT_MINWG 8
#endif
#ifndef ZS_LIT_BITMAP
#define ZS_LIT_BITMAP 1             // literal gather as a stream compaction of the block (0: sequence by sequence)
#endif
#ifndef ZS_LIT_TILES
#define ZS_LIT_TILES 2               // tiles of 64 sequences the literal gather keeps in flight per wavefront
#endif
struct K3Lds {                       // literals kernel
    uint32_t count[256];             // literal histogram
    uint8_t  nbBits[256];
    uint32_t codeNb[256];            // code | nbBits << 16
    uint32_t leafW[256];
    uint16_t leafSym[256];
    uint8_t  lenOfRank[256];
    uint8_t  weights[256];
    union {
        struct { uint32_t pkg[10][256]; uint32_t S[512]; uint32_t npk[12]; } pm;      // package-merge (levels 2..11)
        struct { FseCT ct[1]; int16_t norm[64]; uint8_t tableSymbol[512]; uint32_t cumul[66]; uint16_t stepOut[256]; uint32_t bits[64]; } fse;   // weights table; per step: state bits out | count << 8; the description's bitstream
        uint32_t tile[4][208];       // bit-packing tiles, one per wavefront (streams are written after the tables are done)
        uint32_t hist[8][257];       // literal gather: eight private histograms (lane & 7), rows one word apart in the banks
        struct { uint32_t T[2052]; uint32_t wpar[4], wcnt[4]; uint32_t sel[16]; } gm;   // literal gather: a bit per block byte (toggles at match ends -> inside a match -> literal), per-wavefront parities / literal counts, byte-compaction selectors
    } u;
    uint32_t misc[16];
    uint32_t rngN[ZS_WALK_RANGES], rngCarry[ZS_WALK_RANGES], litBase[ZS_WALK_RANGES];
    uint8_t  rngFirst[ZS_WALK_RANGES];   // index of a range's first record that counts (after the walk kernel's stitch)
    uint32_t wcount[16]; int16_t wnorm[16]; uint32_t rankStart[16], rankCount[16];   // small tables kept out of scratch memory
};
struct SeqLds {                      // sequences kernel.  Kept under 10 KiB: 16 workgroups per CU = one round for 4096 blocks
    uint32_t count[192];             // code counts: [0..63] LL, [64..127] OF, [128..191] ML
    FseCT ct[3];                     // LL, OF, ML
    int16_t norm[64];
    union {
        struct { uint8_t tableSymbol[512]; uint32_t cumul[66]; uint32_t symCount[64]; uint32_t symMask[128]; } build;      // while a table is built
        struct { uint32_t op[3][64]; } chain;    // while the state chains run: per table and code, deltaNbBits | (deltaFindState + 1024) << 20 (one read a step)
    } u;
    uint32_t tile[208];              // bit-packing tile
    uint32_t misc[16];
    uint32_t rngN[ZS_WALK_RANGES], rngCarry[ZS_WALK_RANGES], rngStart[ZS_WALK_RANGES + 1];
    uint8_t  rngFirst[ZS_WALK_RANGES];   // index of a range's first record that counts (after the walk kernel's stitch)
};

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
// Cross-lane moves by data-parallel primitives (DPP): vector-ALU operand modifiers, no LDS round trip (a __shfl is a
// ds_bpermute: ~100 cycles of latency each, six in a row for a scan).  Control codes (gfx9): row_shr:n = 0x110 + n
// (zero fill with bound_ctrl), row_bcast15 = 0x142 (lane 15 of a row to the next row), row_bcast31 = 0x143.
#define ZS_DPP(old, v, ctrl, rowMask, boundCtrl) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rowMask), 0xF, (boundCtrl)))
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += ZS_DPP(0, v, 0x111, 0xF, true);
    v += ZS_DPP(0, v, 0x112, 0xF, true);
    v += ZS_DPP(0, v, 0x114, 0xF, true);
    v += ZS_DPP(0, v, 0x118, 0xF, true);            // inclusive within each row of 16
    v += ZS_DPP(0, v, 0x142, 0xA, false);           // rows 1, 3 += total of the row before
    v += ZS_DPP(0, v, 0x143, 0xC, false);           // rows 2, 3 += total of rows 0..1
    return v;
}
// value of lane l, l the same for the whole wavefront
__device__ __forceinline__ uint32_t wave_get(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return wave_last(wave_incl_scan(v)); }
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
    v = max(v, ZS_DPP(0, v, 0x111, 0xF, true));
    v = max(v, ZS_DPP(0, v, 0x112, 0xF, true));
    v = max(v, ZS_DPP(0, v, 0x114, 0xF, true));
    v = max(v, ZS_DPP(0, v, 0x118, 0xF, true));
    v = max(v, ZS_DPP(0, v, 0x142, 0xA, false));
    v = max(v, ZS_DPP(0, v, 0x143, 0xC, false));
    return wave_last(v);
}

// ordering point between LDS accesses of different lanes of ONE wavefront: LDS instructions of a wave execute in issue
// order, so only the compiler has to be kept from moving them across
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Wave-cooperative forward bit writer.  Each put() appends, lane 0 first, up to 96 bits per lane.
// out32 must be 4-byte aligned; bits are packed little-endian (bit k of the stream = bit k%8 of byte k/8).
// a workgroup copies n bytes: 16-byte pieces (two unaligned 8-byte accesses), four pieces a thread in flight, then the tail.
// (A byte a thread and iteration was 100 dependent load -> store rounds per section.)
__device__ __forceinline__ void zs_block_copy(uint8_t *__restrict__ d, const uint8_t *__restrict__ s, uint32_t n, uint32_t tid, uint32_t nthreads)
{
    const uint32_t n16 = n >> 4;
    for (uint32_t i = tid; i < n16; i += 4 * nthreads) {
        uint64_t a[4], b[4];
        #pragma unroll
        for (uint32_t k = 0; k < 4; k++) { const uint32_t idx = min(i + k * nthreads, n16 - 1); a[k] = zs_load64(s + 16 * idx); b[k] = zs_load64(s + 16 * idx + 8); }
        #pragma unroll
        for (uint32_t k = 0; k < 4; k++) { const uint32_t idx = i + k * nthreads; if (idx < n16) { zs_store64(d + 16 * idx, a[k]); zs_store64(d + 16 * idx + 8, b[k]); } }
    }
    for (uint32_t j = (n16 << 4) + tid; j < n; j += nthreads) d[j] = s[j];
}
// the same for one wavefront inside a kernel short of registers: one 16-byte piece a lane and round
__device__ __forceinline__ void zs_wave_copy(uint8_t *__restrict__ d, const uint8_t *__restrict__ s, uint32_t n, uint32_t lane)
{
    const uint32_t n16 = n >> 4;
    for (uint32_t i = lane; i < n16; i += 64) { const uint64_t a = zs_load64(s + 16 * i), b = zs_load64(s + 16 * i + 8); zs_store64(d + 16 * i, a); zs_store64(d + 16 * i + 8, b); }
    for (uint32_t j = (n16 << 4) + lane; j < n; j += 64) d[j] = s[j];
}

struct BitSink {
    uint32_t *out32;
    uint32_t *tile;       // LDS, >= 200 words
    uint32_t bitpos;
};
__device__ __forceinline__ void sink_init(BitSink &b, void *out, uint32_t *tile)
{
    b.out32 = (uint32_t *)out; b.tile = tile; b.bitpos = 0;
    if (zs_lane() == 0) tile[0] = 0;
    wave_sync();
}
__device__ __forceinline__ void sink_put(BitSink &b, uint64_t lo, uint32_t hi, uint32_t nb)
{
    const uint32_t lane = (uint32_t)zs_lane();
    const uint32_t incl = wave_incl_scan(nb);
    const uint32_t total = wave_last(incl);
    const uint32_t base = b.bitpos & 31u;
    const uint32_t nwords = (base + total + 31u) >> 5;
    for (uint32_t i = 1 + lane; i <= nwords; i += 64) b.tile[i] = 0;
    wave_sync();
    if (nb) {
        const uint32_t bit = base + incl - nb;
        const uint32_t w = bit >> 5, sh = bit & 31u;
        // 96-bit value shifted left by sh (< 32) -> up to 4 words
        const uint32_t v0 = (uint32_t)lo, v1 = (uint32_t)(lo >> 32), v2 = hi;
        const uint32_t w0 = v0 << sh;
        const uint32_t w1 = sh ? ((v1 << sh) | (v0 >> (32 - sh))) : v1;
        const uint32_t w2 = sh ? ((v2 << sh) | (v1 >> (32 - sh))) : v2;
        const uint32_t w3 = sh ? (v2 >> (32 - sh)) : 0u;
        if (w0) atomicOr(&b.tile[w], w0);
        if (w1) atomicOr(&b.tile[w + 1], w1);
        if (w2) atomicOr(&b.tile[w + 2], w2);
        if (w3) atomicOr(&b.tile[w + 3], w3);
    }
    wave_sync();
    const uint32_t nfull = (base + total) >> 5;
    uint32_t *dst = b.out32 + (b.bitpos >> 5);
    for (uint32_t i = lane; i < nfull; i += 64) dst[i] = b.tile[i];
    const uint32_t carry = b.tile[nfull];
    wave_sync();
    if (lane == 0) b.tile[0] = ((base + total) & 31u) ? carry : 0u;
    wave_sync();
    b.bitpos += total;
}
// adds the end mark (one 1 bit), flushes; returns the stream size in bytes
__device__ __forceinline__ uint32_t sink_close(BitSink &b)
{
    const uint32_t lane = (uint32_t)zs_lane();
    sink_put(b, lane == 0 ? 1ull : 0ull, 0u, lane == 0 ? 1u : 0u);
    if (lane == 0 && (b.bitpos & 31u)) b.out32[b.bitpos >> 5] = b.tile[0];
    wave_sync();
    return (b.bitpos + 7u) >> 3;
}

// ---------------------------------------------------------------------------------------------
// lane-0 sequential pieces (small tables; scalar statement in oracle/zso_encoder.c)
// ---------------------------------------------------------------------------------------------
struct BitW { uint64_t acc; uint32_t nbits; uint8_t *ptr, *start, *end; int overflow; };
__device__ static void bw_init(BitW &b, uint8_t *dst, uint32_t cap) { b.acc = 0; b.nbits = 0; b.ptr = b.start = dst; b.end = dst + cap; b.overflow = 0; }
__device__ static void bw_add(BitW &b, uint32_t value, uint32_t nb)
{
    if (!nb) return;
    b.acc |= (uint64_t)(value & ((nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1))) << b.nbits;
    b.nbits += nb;
    while (b.nbits >= 8) { if (b.ptr < b.end) *b.ptr++ = (uint8_t)b.acc; else b.overflow = 1; b.acc >>= 8; b.nbits -= 8; }
}
__device__ static uint32_t bw_close(BitW &b)
{
    bw_add(b, 1, 1);
    if (b.nbits) { if (b.ptr < b.end) *b.ptr++ = (uint8_t)b.acc; else b.overflow = 1; b.nbits = 0; }
    return b.overflow ? 0u : (uint32_t)(b.ptr - b.start);
}

__device__ static void normalizeCounts(int16_t *norm, uint32_t tableLog, const uint32_t *count, uint32_t total, uint32_t maxSym)
{
    const uint32_t tableSize = 1u << tableLog;
    int still = (int)tableSize;
    uint32_t largest = 0;
    for (uint32_t s = 0; s <= maxSym; s++) {
        if (!count[s]) { norm[s] = 0; continue; }
        const uint64_t scaled = (uint64_t)count[s] * tableSize;
        uint32_t p = (uint32_t)(scaled / total);
        const uint32_t rem = (uint32_t)(scaled % total);
        if (2 * (uint64_t)rem >= total) p++;
        if (p == 0) p = 1;
        norm[s] = (int16_t)p;
        still -= (int)p;
        if (norm[s] > norm[largest] || !count[largest]) largest = s;
    }
    if (still > 0) norm[largest] = (int16_t)(norm[largest] + still);
    while (still < 0) {
        uint32_t best = 0; int found = 0;
        for (uint32_t s = 0; s <= maxSym; s++) if (norm[s] > 1 && (!found || norm[s] > norm[best])) { best = s; found = 1; }
        norm[best]--; still++;
    }
}

__device__ static uint32_t writeNCount(uint8_t *dst, uint32_t cap, const int16_t *norm, uint32_t maxSym, uint32_t tableLog)
{
    uint8_t *out = dst; uint8_t *const oend = dst + cap;
    const int tableSize = 1 << tableLog;
    int remaining = tableSize + 1, threshold = tableSize, nbBits = (int)tableLog + 1;
    uint32_t bitStream = 0; int bitCount = 0; uint32_t charnum = 0; int previous0 = 0;
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    while (remaining > 1) {
        if (previous0) {
            uint32_t start = charnum;
            while (charnum <= maxSym && !norm[charnum]) charnum++;
            while (charnum >= start + 24) {
                start += 24; bitStream += 0xFFFFu << bitCount;
                if (out + 2 > oend) return 0;
                out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (charnum >= start + 3) { start += 3; bitStream += 3u << bitCount; bitCount += 2; }
            bitStream += (charnum - start) << bitCount; bitCount += 2;
            if (bitCount > 16) {
                if (out + 2 > oend) return 0;
                out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
            }
        }
        {
            int count = norm[charnum++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (uint32_t)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previous0 = (count == 1);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) {
            if (out + 2 > oend) return 0;
            out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
        }
    }
    if (out + 2 > oend) return 0;
    out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (uint32_t)(out - dst);
}

// cell order is the decoder's (ZStdDecompress.cs:993-1013): spread with the same step and low-probability area
__device__ static void buildCTable(FseCT &ct, uint8_t *tableSymbol, uint32_t *cumul, const int16_t *norm, uint32_t maxSym, uint32_t tableLog)
{
    const uint32_t tableSize = 1u << tableLog, tableMask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    uint32_t highThreshold = tableSize - 1, position = 0;
    ct.tableLog = tableLog; ct.rle = 0;
    cumul[0] = 0;
    for (uint32_t s = 1; s <= maxSym + 1; s++) {
        if (norm[s - 1] == -1) { cumul[s] = cumul[s - 1] + 1; tableSymbol[highThreshold--] = (uint8_t)(s - 1); }
        else cumul[s] = cumul[s - 1] + (uint32_t)norm[s - 1];
    }
    for (uint32_t s = 0; s <= maxSym; s++)
        for (int i = 0; i < norm[s]; i++) {
            tableSymbol[position] = (uint8_t)s;
            position = (position + step) & tableMask;
            while (position > highThreshold) position = (position + step) & tableMask;
        }
    for (uint32_t u = 0; u < tableSize; u++) { const uint8_t sym = tableSymbol[u]; ct.stateTable[cumul[sym]++] = (uint16_t)(tableSize + u); }
    uint32_t total = 0;
    for (uint32_t s = 0; s <= maxSym; s++) {
        const int nv = norm[s];
        if (nv == 0) { ct.deltaNbBits[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct.deltaFindState[s] = 0; }
        else if (nv == 1 || nv == -1) { ct.deltaNbBits[s] = (tableLog << 16) - (1u << tableLog); ct.deltaFindState[s] = (int)total - 1; total++; }
        else {
            const uint32_t maxBitsOut = tableLog - zs_highbit((uint32_t)nv - 1);
            const uint32_t minStatePlus = (uint32_t)nv << maxBitsOut;
            ct.deltaNbBits[s] = (maxBitsOut << 16) - minStatePlus;
            ct.deltaFindState[s] = (int)total - nv;
            total += (uint32_t)nv;
        }
    }
}
__device__ __forceinline__ uint32_t cstate_init(const FseCT &ct, uint32_t symbol)
{
    const uint32_t dnb = ct.deltaNbBits[symbol];
    const uint32_t nbBitsOut = (dnb + (1u << 15)) >> 16;
    const uint32_t v = (nbBitsOut << 16) - dnb;
    return ct.stateTable[(v >> nbBitsOut) + ct.deltaFindState[symbol]];
}

// weights -> FSE (inverse of FSE_decompress_wksp as used by ReadStats, EntropyCommon.cs:226-231), by one wavefront; the weight
// histogram L.wcount[] (all 16 entries) is already filled.  Lane 0 makes the table (<= 13 symbols, <= 64 cells).  The encoder's two
// interleaved states run side by side on lanes 0 and 1 (state "first" takes the symbols nw-3, nw-5, ..., "second" nw-4, nw-6, ...;
// per step: value | bit count << 8 into LDS), then all lanes place the steps' bits by a prefix sum of the counts - the scalar form
// (one lane, a byte-wise bit writer, ~250 dependent steps) was a third of the kernel's time between the code lengths and the streams.
// Same bytes as fseCompressWeights in oracle/zso_encoder.c.
__device__ __forceinline__ uint32_t fseCompressWeightsWave(K3Lds &L, uint8_t *dst, uint32_t cap, const uint8_t *weights, uint32_t nw)
{
    const uint32_t lane = (uint32_t)zs_lane();
    if (lane == 0) {
        uint32_t hs = 0, tl = 0;
        do {
            uint32_t *count = L.wcount; int16_t *norm = L.wnorm;
            uint32_t maxSym = 0, tableLog;
            if (nw <= 1) break;
            for (uint32_t i = 0; i < 16; i++) if (count[i]) maxSym = i;
            bool one = false;
            for (uint32_t i = 0; i <= maxSym; i++) if (count[i] == nw) one = true;
            if (one) break;
            tableLog = 6;
            while (tableLog > 5 && (1u << (tableLog - 1)) >= nw) tableLog--;
            { uint32_t present = 0; for (uint32_t i = 0; i <= maxSym; i++) present += count[i] != 0; if (present > (1u << tableLog)) break; }
            normalizeCounts(norm, tableLog, count, nw, maxSym);
            hs = writeNCount(dst, cap, norm, maxSym, tableLog);
            if (!hs) break;
            buildCTable(L.u.fse.ct[0], L.u.fse.tableSymbol, L.u.fse.cumul, norm, maxSym, tableLog);
            tl = tableLog;
        } while (0);
        L.misc[12] = hs; L.misc[13] = tl;
    }
    L.u.fse.bits[lane] = 0;
    wave_sync();
    const uint32_t hsize = L.misc[12], tableLog = L.misc[13];
    if (!hsize) return 0;
    const FseCT &ct = L.u.fse.ct[0];
    const uint32_t K = nw - 2;                                            // encode steps; step k takes symbol nw - 3 - k
    uint16_t *stepOut = L.u.fse.stepOut;
    uint32_t fin = 0;
    if (lane < 2) {
        uint32_t st = cstate_init(ct, weights[nw - 1 - lane]);
        for (uint32_t k = lane; k < K; k += 2) {
            const uint32_t sym = weights[nw - 3 - k];
            const uint32_t nbo = (st + ct.deltaNbBits[sym]) >> 16;
            stepOut[k] = (uint16_t)((st & ((1u << nbo) - 1u)) | (nbo << 8));
            st = ct.stateTable[(st >> nbo) + ct.deltaFindState[sym]];
        }
        fin = st;
    }
    wave_sync();
    // final states go out st2 first, then st1: with an odd count the first state is st1, with an even count st2
    const uint32_t stFirst = wave_get(fin, 0), stSecond = wave_get(fin, 1);
    const uint32_t st2 = (nw & 1u) ? stSecond : stFirst, st1 = (nw & 1u) ? stFirst : stSecond;
    uint32_t *bits = L.u.fse.bits;
    uint32_t total = 0;
    for (uint32_t k0 = 0; k0 < K + 3; k0 += 64) {
        const uint32_t k = k0 + lane;
        uint32_t val = 0, nb = 0;
        if (k < K) { const uint32_t e = stepOut[k]; val = e & 0xFFu; nb = e >> 8; }
        else if (k == K) { val = st2 & ((1u << tableLog) - 1u); nb = tableLog; }
        else if (k == K + 1) { val = st1 & ((1u << tableLog) - 1u); nb = tableLog; }
        else if (k == K + 2) { val = 1; nb = 1; }                         // the end mark
        const uint32_t incl = wave_incl_scan(nb);
        const uint32_t off = total + incl - nb;
        if (nb) {
            const uint32_t w = off >> 5, sh = off & 31u;
            atomicOr(&bits[w], val << sh);
            if (sh + nb > 32u) atomicOr(&bits[w + 1], val >> (32u - sh));
        }
        total += wave_last(incl);
    }
    wave_sync();
    const uint32_t bytes = (total + 7u) >> 3;
    if (bytes > cap - hsize) return 0;
    const uint8_t *bb = reinterpret_cast<const uint8_t *>(bits);
    for (uint32_t j = lane; j < bytes; j += 64) dst[hsize + j] = bb[j];
    return hsize + bytes;
}

// Huffman table description (inverse of ReadStats, EntropyCommon.cs:198-269), by one wavefront; L.weights[0..maxSym) and the
// histogram of those weights are already filled.
__device__ __forceinline__ uint32_t writeHuffHeaderWave(K3Lds &L, uint8_t *dst, uint32_t cap, uint32_t maxSym, uint32_t tableLog)
{
    const uint32_t lane = (uint32_t)zs_lane();
    uint8_t *weights = L.weights;
    if (maxSym >= 2 && cap > 1) {
        const uint32_t h = fseCompressWeightsWave(L, dst + 1, cap - 1 < 127 ? cap - 1 : 127, weights, maxSym);
        if (h > 1 && h < maxSym / 2 && h < 128) { if (lane == 0) dst[0] = (uint8_t)h; return h + 1; }
    }
    if (maxSym > 128) return 0;
    if ((maxSym + 1) / 2 + 1 > cap) return 0;
    if (lane == 0) { dst[0] = (uint8_t)(128 + (maxSym - 1)); weights[maxSym] = 0; }
    wave_sync();
    for (uint32_t s = 2 * lane; s < maxSym; s += 128) dst[s / 2 + 1] = (uint8_t)((weights[s] << 4) + weights[s + 1]);
    return (maxSym + 1) / 2 + 1;
}

// ---------------------------------------------------------------------------------------------
// package-merge code lengths, wave-parallel (scalar statement: huffLengths in oracle/zso_encoder.c)
// returns the longest code length (= Huffman tableLog)
// ---------------------------------------------------------------------------------------------
__device__ static uint32_t lowerBound(const uint32_t *a, uint32_t n, uint32_t key)   // #elements < key
{ uint32_t lo = 0, hi = n; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; } return lo; }
__device__ static uint32_t upperBound(const uint32_t *a, uint32_t n, uint32_t key)   // #elements <= key
{ uint32_t lo = 0, hi = n; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; } return lo; }

__device__ static uint32_t huffLengths(K3Lds &L, uint32_t maxSym, uint32_t maxBits)
{
    // 256 threads: thread s owns symbol s, later leaf rank s / package index s
    const uint32_t tid = threadIdx.x;
    const uint32_t c = (tid <= maxSym) ? L.count[tid] : 0u;
    L.nbBits[tid] = 0;
    // leaves in ascending (count, symbol) order.  The symbols that occur are packed first (key = count << 8 | symbol, in S), so a
    // symbol's rank is a count over the ~90 keys of a text block, not a test of all 256 counters
    uint32_t *keys = L.u.pm.S;
    const uint32_t lane_ = tid & 63u, wave_ = tid >> 6;
    const uint64_t pres = __ballot(c != 0);
    if (lane_ == 0) L.u.pm.npk[4 + wave_] = (uint32_t)__popcll(pres);      // npk[0..11] is free until the levels start
    __syncthreads();
    uint32_t before = 0, n = 0;
    #pragma unroll
    for (uint32_t v = 0; v < 4; v++) { const uint32_t k = L.u.pm.npk[4 + v]; if (v < wave_) before += k; n += k; }
    const uint32_t key = (c << 8) | tid;
    if (c) keys[before + (uint32_t)__popcll(pres & ((1ull << lane_) - 1ull))] = key;
    __syncthreads();
    if (c) {
        uint32_t rank = 0;
        for (uint32_t t = 0; t < n; t++) rank += (keys[t] < key) ? 1u : 0u;
        L.leafW[rank] = c; L.leafSym[rank] = (uint16_t)tid;
    }
    __syncthreads();
    if (n == 1) { if (tid == 0) L.nbBits[L.leafSym[0]] = 1; __syncthreads(); return 1; }
    uint32_t (*pkg)[256] = L.u.pm.pkg;         // pkg[level - 2]
    uint32_t *S = L.u.pm.S;
    uint32_t *npk = L.u.pm.npk;                // npk[level]
    if (tid == 0) npk[1] = 0;
    __syncthreads();
    for (uint32_t level = 2; level <= maxBits; level++) {
        const uint32_t np = npk[level - 1];
        const uint32_t *prev = (level >= 3) ? pkg[level - 3] : nullptr;
        if (tid < n) { const uint32_t w = L.leafW[tid]; S[tid + (np ? lowerBound(prev, np, w) : 0u)] = w; }
        if (tid < np) { const uint32_t w = prev[tid]; S[tid + upperBound(L.leafW, n, w)] = w; }
        __syncthreads();
        const uint32_t have = (n + np) >> 1;
        if (tid < have) pkg[level - 2][tid] = S[2 * tid] + S[2 * tid + 1];
        if (tid == 0) npk[level] = have;
        __syncthreads();
    }
    L.lenOfRank[tid] = 0;
    __syncthreads();
    uint32_t m = 2 * n - 2;
    for (uint32_t level = maxBits; level >= 1; level--) {
        const uint32_t np = npk[level];
        const uint32_t *cur = (level >= 2) ? pkg[level - 2] : nullptr;
        if (m > n + np) m = n + np;
        bool in = false;
        if (tid < n) { const uint32_t pos = tid + (np ? lowerBound(cur, np, L.leafW[tid]) : 0u); in = pos < m; }
        const uint32_t li = (uint32_t)__syncthreads_count(in);
        if (in) L.lenOfRank[tid]++;
        const uint32_t pi = m - li;
        m = 2 * pi;
        if (!m) break;
    }
    __syncthreads();
    if (tid < n) L.nbBits[L.leafSym[tid]] = L.lenOfRank[tid];
    __syncthreads();
    return L.lenOfRank[0];
}

// code values in the decoder's table order (HufDecompress.cs:148-176), weights and their histogram; 256 threads
// (thread s owns symbol s).  code[s] = rankStart[w] / 2^(w-1) + (number of lower symbols with the same weight).
__device__ static void huffCodesAndWeights(K3Lds &L, uint32_t maxSym, uint32_t tableLog)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nb = (tid <= maxSym) ? L.nbBits[tid] : 0u;
    const uint32_t w = nb ? tableLog + 1 - nb : 0u;
    if (tid < 16) { L.rankCount[tid] = 0; L.wcount[tid] = 0; }
    __syncthreads();
    if (nb) atomicAdd(&L.rankCount[w], 1u);
    if (tid < maxSym) { L.weights[tid] = (uint8_t)w; atomicAdd(&L.wcount[w], 1u); }       // the last symbol's weight is implied
    __syncthreads();
    if (tid == 0) { uint32_t next = 0; for (uint32_t k = 1; k <= tableLog; k++) { L.rankStart[k] = next; next += L.rankCount[k] << (k - 1); } }
    // index among the symbols of equal weight: lower lanes of my wavefront + whole lower wavefronts (counted per weight)
    uint32_t idx = 0;
    for (uint32_t k = 1; k <= tableLog; k++) {
        const uint64_t m = __ballot(w == k);
        if (w == k) idx = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (lane == 0) L.u.pm.S[wave * 16 + k] = (uint32_t)__popcll(m);                    // S is free again after the code lengths
    }
    __syncthreads();
    if (nb) {
        for (uint32_t v = 0; v < wave; v++) idx += L.u.pm.S[v * 16 + w];
        L.codeNb[tid] = ((L.rankStart[w] >> (w - 1)) + idx) | (nb << 16);
    }
    __syncthreads();
}

// one Huffman stream of lits[from .. from+len) into tmp (4-byte aligned); last symbol first. returns bytes.
__device__ __forceinline__ uint32_t huffEncodeStream(K3Lds &L, uint32_t *tile, uint8_t *tmp, const uint8_t *lits, uint32_t from, uint32_t len)
{
    const uint32_t lane = (uint32_t)zs_lane();
    BitSink sink; sink_init(sink, tmp, tile);
    uint32_t remaining = len;
    // lane l takes symbols k = 8l .. 8l+7 of the tile, i.e. the 8 literals ending at position from + remaining - 1 - 8l, last first
    const uint32_t k0 = lane * 8;
    auto fetch = [&](uint32_t rem) -> uint64_t {
        const uint32_t T = min(512u, rem);
        uint64_t e = 0;
        if (k0 + 8 <= T) e = zs_load64(lits + from + rem - 8 - k0);
        else for (uint32_t j = 0; j < 8; j++) if (k0 + j < T) e |= (uint64_t)lits[from + rem - 1 - k0 - j] << (8 * (7 - j));
        return e;
    };
    uint64_t next = remaining ? fetch(remaining) : 0ull;                   // the next tile's bytes travel while this one is packed
    while (remaining) {
        const uint32_t T = min(512u, remaining);
        uint64_t lo = 0; uint32_t hi = 0, nb = 0;
        const uint64_t eight = next;
        if (remaining > T) next = fetch(remaining - T);
        #pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (k0 + j < T) {
                const uint32_t sym = (uint32_t)(eight >> (8 * (7 - j))) & 0xFFu;
                const uint32_t cn = L.codeNb[sym], c = cn & 0xFFFFu, b = cn >> 16;
                if (nb < 64) { lo |= (uint64_t)c << nb; if (nb + b > 64) hi |= c >> (64 - nb); }
                else hi |= c << (nb - 64);
                nb += b;
            }
        }
        sink_put(sink, lo, hi, nb);
        remaining -= T;
    }
    return sink_close(sink);
}

// ---------------------------------------------------------------------------------------------
// per-block result of the two encode kernels, consumed by k_assemble_frames
// ---------------------------------------------------------------------------------------------
struct ZsBlockMeta { uint32_t type;       // 0 raw, 1 rle, 2 literal + sequence sections present
                     uint32_t rleByte; uint32_t litSecSize; uint32_t seqSecSize;      // seqSecSize 0xFFFFFFFF: section failed / overflowed
                     uint32_t seqHdrSize, seqGap;     // the sequence section lies in its buffer as seqHdrSize bytes, seqGap (0..3) unused bytes,
                     uint32_t pad[2]; };              // then the bitstream (built 4-byte aligned); k_assemble_frames closes the gap as it copies
#define ZS_LITSEC_STRIDE  (ZS_BLOCK_MAX + 1024u)
#define ZS_SEQSEC_STRIDE  (ZS_BLOCK_MAX + 4096u)
#define ZS_STREAM_STRIDE  (24u * 1024u)          // per Huffman stream scratch: 16384 symbols * 11 bits = 22528 B max

// exclusive "last lane below me with flag" : returns lane index or -1
__device__ __forceinline__ int lastFlagBelow(bool flag)
{
    const uint64_t m = __ballot(flag);
    const uint32_t lane = (uint32_t)zs_lane();
    const uint64_t below = lane ? (m & ((1ull << lane) - 1)) : 0ull;
    return below ? 63 - __builtin_clzll(below) : -1;
}

// Range bookkeeping by one wavefront, lane r = walk range r (64 ranges):
//   rngN[r] sequences, rngStart[r] index of its first sequence in block order, rngCarry[r] literals carried into its first
//   sequence (trailing literals of the ranges since the last one that had a sequence), litBase[r] (optional) index of its
//   first own literal; returns the literals left after the last sequence of the block and the total literal count.
__device__ __forceinline__ void loadRangesWave(const ZsRangeHdr *hdr, uint32_t *rngN, uint32_t *rngCarry, uint32_t *rngStart, uint32_t *litBase, uint8_t *rngFirst,
                                               uint32_t *lastLits, uint32_t *allLits)
{
    const uint32_t lane = (uint32_t)zs_lane();
    const ZsRangeHdr h = hdr[lane];
    const uint32_t ns = h.nseq, tr = h.trailing, lsum = h.litSum + h.trailing;
    const uint32_t nsIncl = wave_incl_scan(ns), trIncl = wave_incl_scan(tr), lsIncl = wave_incl_scan(lsum);
    const uint32_t P = trIncl - tr;                                     // trailing literals of the ranges before me
    const int j = lastFlagBelow(ns != 0);
    const uint32_t Pj = (uint32_t)__shfl((int)P, max(j, 0));
    rngN[lane] = ns; rngFirst[lane] = (uint8_t)h.first; if (rngStart) rngStart[lane] = nsIncl - ns; rngCarry[lane] = (j >= 0) ? P - Pj : P;
    if (litBase) litBase[lane] = lsIncl - lsum;
    const uint64_t has = __ballot(ns != 0);
    const uint32_t total = wave_last(trIncl);
    const uint32_t Plast = has ? wave_get(P, 63 - __builtin_clzll(has)) : 0u;
    if (lane == 63) { if (rngStart) rngStart[ZS_WALK_RANGES] = nsIncl; *lastLits = has ? total - Plast : total; *allLits = lsIncl; }
}

struct ZsChunkDesc { uint64_t srcOff; uint64_t dstOff; uint32_t size; uint32_t firstBlock; uint32_t nBlocks; uint32_t pad; };

// frame header of a chunk (magic + FHD + FCS, single segment); returns its size.  One thread writes it.
__device__ __forceinline__ uint32_t zs_frame_header(uint8_t *out, uint32_t size, bool writer)
{
    if (writer) {
        out[0] = 0x28; out[1] = 0xB5; out[2] = 0x2F; out[3] = 0xFD;
        if (size < 256) { out[4] = 0x20; out[5] = (uint8_t)size; }
        else if (size < 65536 + 256) { out[4] = 0x60; const uint32_t v = size - 256; out[5] = (uint8_t)v; out[6] = (uint8_t)(v >> 8); }
        else { out[4] = 0xA0; out[5] = (uint8_t)size; out[6] = (uint8_t)(size >> 8); out[7] = (uint8_t)(size >> 16); out[8] = (uint8_t)(size >> 24); }
    }
    return (size < 256) ? 6 : (size < 65536 + 256 ? 7 : 9);
}
// one block into its frame at out + pos (all threads of the workgroup); returns the bytes written.  A block is emitted compressed
// iff both sections exist and literal section + sequence section < block size (else raw; RLE if flagged).
__device__ __forceinline__ uint32_t zs_emit_block(uint8_t *out, uint32_t pos, const uint8_t *blockSrc, uint32_t n, uint32_t last, const ZsBlockMeta m,
                                                 const uint8_t *p1, const uint8_t *p2, uint32_t tid, uint32_t nthreads)
{
    uint32_t type = m.type;
    uint32_t total = 0;
    if (type == 2) {
        // same decisions as the scalar statement: room left after the literals, section failures, final size test
        if (m.seqSecSize == 0xFFFFFFFFu || m.litSecSize + 4 > n + 512) type = 0;
        else { total = m.litSecSize + m.seqSecSize; if (total > n + 512 || total >= n) type = 0; }
    }
    if (type == 1) {
        if (tid == 0) { const uint32_t h = last + (1u << 1) + (n << 3); out[pos] = (uint8_t)h; out[pos + 1] = (uint8_t)(h >> 8); out[pos + 2] = (uint8_t)(h >> 16); out[pos + 3] = (uint8_t)m.rleByte; }
        return 4;
    }
    if (type == 2) {
        if (tid == 0) { const uint32_t h = last + (2u << 1) + (total << 3); out[pos] = (uint8_t)h; out[pos + 1] = (uint8_t)(h >> 8); out[pos + 2] = (uint8_t)(h >> 16); }
        if (p1 != out + pos + 3) zs_block_copy(out + pos + 3, p1, m.litSecSize, tid, nthreads);       // (the literals kernel builds a one-block chunk's section in place)
        zs_block_copy(out + pos + 3 + m.litSecSize, p2, m.seqHdrSize, tid, nthreads);
        zs_block_copy(out + pos + 3 + m.litSecSize + m.seqHdrSize, p2 + m.seqHdrSize + m.seqGap, m.seqSecSize - m.seqHdrSize, tid, nthreads);
        return 3 + total;
    }
    if (tid == 0) { const uint32_t h = last + (0u << 1) + (n << 3); out[pos] = (uint8_t)h; out[pos + 1] = (uint8_t)(h >> 8); out[pos + 2] = (uint8_t)(h >> 16); }
    zs_block_copy(out + pos + 3, blockSrc, n, tid, nthreads);
    return 3 + n;
}

// ---------------------------------------------------------------------------------------------
// k_encode_literals : one workgroup of 4 wavefronts per block.  Block type (raw for tiny blocks, RLE block),
// literal gather + histogram (wavefront w takes ranges w and w+4), Huffman lengths by package-merge (256 threads),
// table description (one lane), the 4 Huffman streams (one wavefront each)  -> litSec[], meta.{type, rleByte, litSecSize}
// ---------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256, ZS_LIT_MINWG)
k_encode_literals(const uint8_t *__restrict__ src, const ZsBlockDesc *__restrict__ blocks,
                  const ZsSeqRec *__restrict__ seqAll, const ZsRangeHdr *__restrict__ hdrAll,
                  uint8_t *__restrict__ litsAll, uint8_t *__restrict__ streamAll, uint8_t *__restrict__ litSecAll,
                  ZsBlockMeta *__restrict__ metas, int stopAt,
                  const ZsChunkDesc *__restrict__ chunks, const uint8_t *__restrict__ seqSecAll, uint8_t *__restrict__ dst, uint32_t *__restrict__ dstSizes)
{
    __shared__ K3Lds L;
    const uint32_t blk = blockIdx.x;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const ZsBlockDesc bd = blocks[blk];
    const uint8_t *s = src + bd.srcOff;
    const uint32_t n = bd.size;
    const ZsSeqRec *seqBase = seqAll + (size_t)blk * ZS_WALK_RANGES * ZS_SEQ_PER_RANGE;
    const ZsRangeHdr *hdr = hdrAll + (size_t)blk * ZS_WALK_RANGES;
    uint8_t *lits = litsAll + (size_t)blk * (ZS_BLOCK_MAX + 64);
    uint8_t *streams = streamAll + (size_t)blk * 4 * ZS_STREAM_STRIDE;
    // the literal section of a one-block chunk is built where its frame wants it (behind the frame header and the 3-byte block header: the
    // slot holds zsmi_compressBound(n) >= n + 27 bytes, the section never more than n + 3), every other block's in the section buffer
    const bool solo = bd.firstInChunk && bd.lastInChunk && !ZS_STOPPED;
    uint8_t *payload = litSecAll + (size_t)blk * ZS_LITSEC_STRIDE;
    if (solo) { const ZsChunkDesc cd0 = chunks[bd.chunk]; payload = dst + cd0.dstOff + zs_frame_header(nullptr, cd0.size, false) + 3; }
    const uint32_t cap = n + 512;

    // The block's literal side is done: its meta goes out; a chunk of ONE block is assembled right here (the sequences kernel ran before this
    // one on the stream, its section and meta fields are there) - k_assemble_frames is launched only for batches with longer chunks
    #define FINISH(tp, lsz, rb) do { \
        if (tid == 0) { metas[blk].type = (tp); metas[blk].rleByte = (rb); metas[blk].litSecSize = (lsz); } \
        if (bd.firstInChunk && bd.lastInChunk && !ZS_STOPPED) { \
            __syncthreads();                         /* the literal section was written by all wavefronts */ \
            ZsBlockMeta m_ = metas[blk]; m_.type = (tp); m_.rleByte = (rb); m_.litSecSize = (lsz); \
            const ZsChunkDesc cd_ = chunks[bd.chunk]; \
            uint8_t *out_ = dst + cd_.dstOff; \
            uint32_t pos_ = zs_frame_header(out_, cd_.size, tid == 0); \
            pos_ += zs_emit_block(out_, pos_, s, n, 1u, m_, payload, seqSecAll + (size_t)blk * ZS_SEQSEC_STRIDE, tid, 256); \
            if (tid == 0) dstSizes[bd.chunk] = pos_; \
        } \
        return; } while (0)

    if (n == 0) FINISH(0, 0, 0);
    {   // RLE block: every byte equal (ZStdDecompress.cs:1945-1950 on the decode side)
        // 16 bytes per thread and load; most blocks fail within the first 4 KiB
        const uint32_t b0 = s[0];
        const uint64_t rep = 0x0101010101010101ull * b0;
        bool diff = false;
        for (uint32_t base = 0; base < n; base += 256 * 16) {
            const uint32_t i = base + tid * 16;
            if (i + 16 <= n) { const uint64_t a = zs_load64(s + i), b = zs_load64(s + i + 8); diff |= (a != rep) | (b != rep); }
            else for (uint32_t j = i; j < n; j++) diff |= (s[j] != b0);
            if (__syncthreads_or(diff)) { diff = true; break; }
        }
        if (!diff) FINISH(1, 0, b0);
    }
    if (n < 16) FINISH(0, 0, 0);

    if (wave == 0) loadRangesWave(hdr, L.rngN, L.rngCarry, nullptr, L.litBase, L.rngFirst, &L.misc[1], &L.misc[2]);
#if !ZS_LIT_BITMAP
    #pragma unroll
    for (uint32_t k = 0; k < 8; k++) L.u.hist[k][tid] = 0;
#endif
    __syncthreads();
    const uint32_t lastLits = L.misc[1];
    const uint32_t nlit = L.misc[2];

    // ---- literals: gather into lits[], histogram.  Range r's own literals start at litBase[r]; the first sequence of a
    //      range also takes the literals carried over from the ranges before it (they sit right in front). ----
    // The wavefront's tiles (64 sequences of one of its ranges) are taken ZS_LIT_TILES at a time: the record loads of all of
    // them are issued together, then their source loads, then the stores.  A tile alone is two dependent memory round
    // trips, and the loop was bound by exactly that latency.
    uint32_t *hist = L.u.hist[lane & 7u];                                  // this lane's private histogram
    // a block without a match (a matchless unit: noise, packed data) has its literals where they are: no gather, the source is read in its place
    const bool ungathered = nlit == n;
    const uint8_t *litp = ungathered ? s : lits;
#if ZS_LIT_BITMAP
    // The literals of a block are its bytes outside every match, in order.  So: a bit per byte, toggled at every match start and
    // end (LDS atomics, lane = sequence); a prefix xor turns the toggles into "inside a match"; the rest is a stream compaction of
    // the source, 16 bytes a lane and round, every lane busy with contiguous, coalesced bytes.  (Taking the literal runs sequence by
    // sequence - a run of ~3 bytes per lane, its own load and store pieces - was ~300 instructions per 64 sequences, issue-bound.)
    if (!ungathered) {
        uint32_t *T = L.u.gm.T;
        for (uint32_t i = tid; i < 2052; i += 256) T[i] = 0;
        if (tid < 16) {                                                   // byte selectors of v_perm for a 4-bit mask: the set bytes in order, then zeros
            uint32_t sel = 0, k = 0;
            for (uint32_t bsel = 0; bsel < 4; bsel++) if (tid & (1u << bsel)) { sel |= bsel << (8 * k); k++; }
            for (; k < 4; k++) sel |= 0x0Cu << (8 * k);
            L.u.gm.sel[tid] = sel;
        }
        __syncthreads();
        // 1. toggles.  Wavefront w takes ranges w, w + 4, ...; four ranges at a time, their record loads (<= 4 per lane and range) issued together
        for (uint32_t r0 = wave; r0 < ZS_WALK_RANGES; r0 += 16) {
            uint2 rec[4][4]; uint32_t ns[4];
            #pragma unroll
            for (uint32_t g = 0; g < 4; g++) {
                const uint32_t r = r0 + 4 * g;
                ns[g] = (r < ZS_WALK_RANGES) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.rngN[r]) : 0u;
                const ZsSeqRec *rb = seqBase + (size_t)min(r, (uint32_t)ZS_WALK_RANGES - 1) * ZS_SEQ_PER_RANGE + L.rngFirst[min(r, (uint32_t)ZS_WALK_RANGES - 1)];
                #pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    rec[g][q] = make_uint2(0, 0);
                    if (64 * q < ns[g] && lane + 64 * q < ns[g]) rec[g][q] = *reinterpret_cast<const uint2 *>(rb + lane + 64 * q);
                }
            }
            #pragma unroll
            for (uint32_t g = 0; g < 4; g++) {
                #pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    if (64 * q < ns[g] && lane + 64 * q < ns[g]) {
                        const uint32_t p0 = zs_rec_pos(rec[g][q].y), p1 = p0 + zs_rec_ml(rec[g][q].x);
                        atomicXor(&T[p0 >> 5], 1u << (p0 & 31u));
                        atomicXor(&T[p1 >> 5], 1u << (p1 & 31u));
                    }
                }
            }
        }
        __syncthreads();
        // 2. prefix xor over the 65536 bits (thread t: dwords 8 t .. 8 t + 7), literal bits = its complement below n, literals per wavefront region
        uint32_t x[8];
        {
            uint32_t carry = 0;
            #pragma unroll
            for (uint32_t d = 0; d < 8; d++) {
                uint32_t y = T[8 * tid + d];
                y ^= y << 1; y ^= y << 2; y ^= y << 4; y ^= y << 8; y ^= y << 16;
                y ^= 0u - carry;                                          // the dwords before end inside a match: all flipped
                carry = y >> 31;
                x[d] = y;
            }
            const uint64_t pm = __ballot(carry != 0);
            const uint32_t pin = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull)) & 1u;
            if (lane == 0) L.u.gm.wpar[wave] = (uint32_t)__popcll(pm) & 1u;
            __syncthreads();
            uint32_t flip = pin;
            for (uint32_t v = 0; v < wave; v++) flip ^= L.u.gm.wpar[v];
            uint32_t cnt = 0;
            #pragma unroll
            for (uint32_t d = 0; d < 8; d++) {
                const uint32_t bitBase = (8 * tid + d) * 32;
                uint32_t lm = ~(x[d] ^ (0u - flip));
                if (bitBase + 32 > n) lm = (bitBase < n) ? (lm & ((1u << (n - bitBase)) - 1u)) : 0u;
                T[8 * tid + d] = lm;
                cnt += (uint32_t)__popc(lm);
            }
            const uint32_t tot = wave_sum(cnt);
            if (lane == 0) L.u.gm.wcnt[wave] = tot;
        }
        __syncthreads();
        // 3. compaction: wavefront w takes the bytes [16384 w, 16384 (w + 1)), 1 KiB a round (two rounds' loads in flight)
        uint32_t running = 0;
        for (uint32_t v = 0; v < wave; v++) running += L.u.gm.wcnt[v];
        const uint32_t regionEnd = min(n, 16384u * (wave + 1));
        for (uint32_t p0 = 16384u * wave; p0 < regionEnd; p0 += 2048) {
            uint64_t a[2], b[2]; uint32_t m16[2];
            #pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t p = p0 + 1024 * h + 16 * lane;
                a[h] = 0; b[h] = 0; m16[h] = 0;
                if (p < n) {
                    m16[h] = (T[p >> 5] >> (p & 31u)) & 0xFFFFu;
                    if (p + 16 <= n) { a[h] = zs_load64(s + p); b[h] = zs_load64(s + p + 8); }
                    else for (uint32_t j = 0; p + j < n; j++) { const uint64_t c = s[p + j]; if (j < 8) a[h] |= c << (8 * j); else b[h] |= c << (8 * (j - 8)); }
                }
            }
            #pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                if (p0 + 1024 * h >= regionEnd) continue;
                const uint32_t m = m16[h];
                const uint32_t c0 = (uint32_t)__popc(m & 0xFu), c1 = (uint32_t)__popc(m & 0xF0u), c2 = (uint32_t)__popc(m & 0xF00u), c3 = (uint32_t)__popc(m & 0xF000u);
                const uint32_t q0 = __builtin_amdgcn_perm(0u, (uint32_t)a[h], L.u.gm.sel[m & 15u]);
                const uint32_t q1 = __builtin_amdgcn_perm(0u, (uint32_t)(a[h] >> 32), L.u.gm.sel[(m >> 4) & 15u]);
                const uint32_t q2 = __builtin_amdgcn_perm(0u, (uint32_t)b[h], L.u.gm.sel[(m >> 8) & 15u]);
                const uint32_t q3 = __builtin_amdgcn_perm(0u, (uint32_t)(b[h] >> 32), L.u.gm.sel[(m >> 12) & 15u]);
                const uint64_t lo = (uint64_t)q0 | ((uint64_t)q1 << (8 * c0)), hi = (uint64_t)q2 | ((uint64_t)q3 << (8 * c2));   // <= 8 bytes each
                const uint32_t cl = c0 + c1, cnt = cl + c2 + c3;
                // 16 compacted bytes (w0 low, w1 high): lo, then hi from byte cl
                const uint64_t w0 = (cl < 8) ? (lo | (hi << (8 * cl))) : lo;
                const uint64_t w1 = (cl == 0) ? 0ull : ((cl < 8) ? (hi >> (64 - 8 * cl)) : hi);
                const uint32_t incl = wave_incl_scan(cnt);
                uint8_t *dp = lits + running + incl - cnt;
                if (cnt == 16) { zs_store64(dp, w0); zs_store64(dp + 8, w1); }
                else {
                    uint64_t t = w0; uint32_t at = 0;
                    if (cnt & 8) { zs_store64(dp, t); t = w1; at = 8; }
                    if (cnt & 4) { zs_store32(dp + at, (uint32_t)t); t >>= 32; at += 4; }
                    if (cnt & 2) { zs_store16(dp + at, (uint16_t)t); t >>= 16; at += 2; }
                    if (cnt & 1) dp[at] = (uint8_t)t;
                }
                running += wave_last(incl);
            }
        }
        __syncthreads();                                                  // the bit plane is done with: the histograms take its place
    }
    #pragma unroll
    for (uint32_t k = 0; k < 8; k++) L.u.hist[k][tid] = 0;
    __syncthreads();
    // 4. histogram of the gathered literals, four bytes a thread and round
    for (uint32_t j = 4 * tid; j < nlit; j += 1024) {
        const uint32_t k = min(4u, nlit - j);
        uint32_t w;
        if (k == 4 || !ungathered) w = zs_load32(litp + j);               // (the literal buffer has 64 bytes of slack; the source has none)
        else { w = 0; for (uint32_t q = 0; q < k; q++) w |= (uint32_t)litp[j + q] << (8 * q); }
        atomicAdd(&hist[w & 0xFFu], 1u);
        if (k > 1) atomicAdd(&hist[(w >> 8) & 0xFFu], 1u);
        if (k > 2) atomicAdd(&hist[(w >> 16) & 0xFFu], 1u);
        if (k > 3) atomicAdd(&hist[w >> 24], 1u);
    }
    (void)lastLits;
    __syncthreads();
#else
    {
        constexpr uint32_t GT = ZS_LIT_TILES;
        auto rangeN = [&](uint32_t r) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)L.rngN[r]); };
        uint32_t r = wave, base = 0, done = 0;                               // next tile: range, first sequence (wavefront-uniform)
        while (r < ZS_WALK_RANGES && rangeN(r) == 0) r += 4;
        while (r < ZS_WALK_RANGES) {
            uint32_t tr[GT], tb[GT];                                          // the tiles of this round; tr == ZS_WALK_RANGES: none
            uint2 rec[GT];                                                    // records as raw words: ll, ml | off, flags
            #pragma unroll
            for (uint32_t g = 0; g < GT; g++) {                               // stage 1: the records
                tr[g] = r; tb[g] = base;
                rec[g] = make_uint2(0, 0);
                if (r < ZS_WALK_RANGES) {
                    const uint32_t ns = rangeN(r), k = base + lane;
                    if (k < ns) rec[g] = *reinterpret_cast<const uint2 *>(seqBase + (size_t)r * ZS_SEQ_PER_RANGE + L.rngFirst[r] + k);
                    base += 64;
                    if (base >= ns) { base = 0; r += 4; while (r < ZS_WALK_RANGES && rangeN(r) == 0) r += 4; }
                }
            }
            uint32_t ll[GT], dstOff[GT], srcPos[GT];
            uint64_t w0[GT], w1[GT];
            #pragma unroll
            for (uint32_t g = 0; g < GT; g++) {                               // stage 2: places in the literal buffer, source loads
                ll[g] = 0; dstOff[g] = 0; srcPos[g] = 0; w0[g] = 0; w1[g] = 0;
                if (tr[g] < ZS_WALK_RANGES) {
                    const uint32_t k = tb[g] + lane;
                    uint32_t l = (k < rangeN(tr[g])) ? zs_rec_ll(rec[g].x) : 0u;
                    if (k == 0) l += L.rngCarry[tr[g]];
                    if (tb[g] == 0) done = L.litBase[tr[g]] - L.rngCarry[tr[g]];
                    const uint32_t incl = wave_incl_scan(l);
                    ll[g] = l; dstOff[g] = done + incl - l; srcPos[g] = zs_rec_pos(rec[g].y) - l;
                    done += wave_last(incl);
                    // short runs by their own lane (one round of loads), long runs by the whole wavefront
                    if (l && l <= 16) {
                        const uint32_t endPos = srcPos[g] + l, skip = 16 - l;
                        if (endPos >= 16) { w0[g] = zs_load64(s + endPos - 16); w1[g] = zs_load64(s + endPos - 8); }
                        else for (uint32_t j = 0; j < l; j++) { const uint64_t c = s[srcPos[g] + j]; const uint32_t bi = skip + j; if (bi < 8) w0[g] |= c << (8 * bi); else w1[g] |= c << (8 * (bi - 8)); }
                    }
                }
            }
            #pragma unroll
            for (uint32_t g = 0; g < GT; g++) {                               // stage 3: the stores and the histogram
                if (tr[g] >= ZS_WALK_RANGES) continue;
                const uint32_t l = ll[g];
                if (l && l <= 16) {
                    // the run sits right-aligned in the 16-byte window (w0, w1): it leaves in at most five unaligned stores, biggest
                    // piece first from its end, and is counted a window word at a time (a loop over its bytes cost the vector ALU
                    // ~200 instructions a tile at a third of the lanes).  Counting in a pass of its own over the gathered buffer
                    // (every lane busy) was slower: 0.50 vs 0.48 ms for the kernel
                    uint8_t *dp = lits + dstOff[g];
                    uint64_t t1 = w1[g];
                    if (l == 16) { zs_store64(dp, w0[g]); zs_store64(dp + 8, t1); }
                    else {
                        if (l & 8) { zs_store64(dp + l - 8, t1); t1 = w0[g]; }                   // what is left ends at the top of t1
                        if (l & 4) { zs_store32(dp + (l & 7) - 4, (uint32_t)(t1 >> 32)); t1 <<= 32; }
                        if (l & 2) { zs_store16(dp + (l & 3) - 2, (uint16_t)(t1 >> 48)); t1 <<= 16; }
                        if (l & 1) dp[0] = (uint8_t)(t1 >> 56);
                    }
                }
                {
                    const uint32_t first = (l && l <= 16) ? 16 - l : 16;        // window bytes [first, 16) are literals
                    const uint32_t wd[4] = { (uint32_t)w0[g], (uint32_t)(w0[g] >> 32), (uint32_t)w1[g], (uint32_t)(w1[g] >> 32) };
                    #pragma unroll
                    for (uint32_t i = 0; i < 4; i++) {
                        if (!__any(first < 4 * i + 4)) continue;
                        #pragma unroll
                        for (uint32_t bb = 0; bb < 4; bb++) if (4 * i + bb >= first) atomicAdd(&hist[(wd[i] >> (8 * bb)) & 0xFFu], 1u);
                    }
                }
                uint64_t longm = __ballot(l > 16);
                while (longm) {
                    const int t = __builtin_ctzll(longm); longm &= longm - 1;
                    const uint32_t l2 = wave_get(l, t), d2 = wave_get(dstOff[g], t), s2 = wave_get(srcPos[g], t);
                    for (uint32_t j = lane; j < l2; j += 64) { const uint8_t c = s[s2 + j]; lits[d2 + j] = c; atomicAdd(&hist[c], 1u); }
                }
            }
        }
    }
    for (uint32_t j = tid; j < lastLits; j += 256) { const uint8_t c = s[n - lastLits + j]; lits[nlit - lastLits + j] = c; atomicAdd(&hist[c], 1u); }
    __syncthreads();
#endif
    {
        uint32_t c = 0;
        #pragma unroll
        for (uint32_t k = 0; k < 8; k++) c += L.u.hist[k][tid];
        L.count[tid] = c;
    }
    __syncthreads();
    if (ZS_STOP_AT(1)) FINISH(0, 0, 0);            // timing aid (ZSMI_STOP_LIT): stop after the literal gather

    // ---- literals section (inverse of DecodeLiteralsBlock, ZStdDecompress.cs:683-821) ----
    if (wave == 0) {
        uint32_t maxSym = 0, largest = 0, smallest = 0xFFFFFFFFu;
        for (uint32_t b = 0; b < 256; b += 64) {
            const uint32_t c = L.count[b + lane];
            const uint64_t present = __ballot(c != 0);
            if (present) maxSym = b + 63u - (uint32_t)__builtin_clzll(present);
            largest = max(largest, wave_max(c));
            smallest = min(smallest, ~wave_max(~c));
        }
        if (lane == 0) { L.misc[3] = maxSym; L.misc[4] = largest; L.misc[5] = smallest; }
    }
    __syncthreads();
    const uint32_t maxSym = L.misc[3], largest = L.misc[4];
    // all 256 byte values occur and the most frequent less than twice as often as the rarest (noise, packed data in blocks of 64 KiB): every
    // Huffman merge then pairs two nodes of one level before any node of the next - the code is the complete tree, 8 bits a symbol, and
    // the section cannot be smaller than the literals: no code is built
    const bool flatCounts = L.misc[5] != 0 && largest < 2u * L.misc[5];
    uint32_t litSecSize = 0;
    bool done = false;
    if (nlit > 0 && largest == nlit && nlit > 4) {
        if (tid == 0) {
            const uint8_t b0 = litp[0];
            if (nlit < 32) { payload[0] = (uint8_t)(1 + (nlit << 3)); payload[1] = b0; }
            else if (nlit < 4096) { const uint32_t h = 1 + (1 << 2) + (nlit << 4); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); payload[2] = b0; }
            else { const uint32_t h = 1 + (3 << 2) + (nlit << 4); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); payload[2] = (uint8_t)(h >> 16); payload[3] = b0; }
        }
        litSecSize = (nlit < 32) ? 2 : (nlit < 4096 ? 3 : 4);
        done = true;
    }
    if (!done && nlit >= 64 && !flatCounts) {
        const uint32_t tableLog = huffLengths(L, maxSym, ZS_HUF_MAXBITS);
        if (ZS_STOP_AT(2)) FINISH(0, 0, 0);    // stop after the code lengths
        const uint32_t lhSize = 3 + (nlit >= 1024) + (nlit >= 16384);
        const bool single = nlit < 256;
        // The code lengths say how many bits the streams will hold: when the section cannot come out smaller than the literals themselves
        // (noise, packed data: 8 bits a symbol) the codes, the table description and the four streams are not made at all - the same
        // outcome as the size test behind them (the streams are at least their bits, the description at least a byte), 0.37 -> 0.15 ms a
        // launch of 4096 blocks of noise
        {
            const uint32_t mine = (tid <= maxSym) ? L.count[tid] * (uint32_t)L.nbBits[tid] : 0u;
            const uint32_t ws = wave_sum(mine);
            if (lane == 0) L.misc[12 + wave] = ws;
            __syncthreads();
        }
        const uint32_t codeBits = L.misc[12] + L.misc[13] + L.misc[14] + L.misc[15];
        const bool hopeless = lhSize + 1u + (single ? 0u : 6u) + (codeBits >> 3) >= nlit + (3 - (nlit < 32) - (nlit < 4096));
        if (!hopeless) {
        huffCodesAndWeights(L, maxSym, tableLog);
        if (wave == 0) { const uint32_t hs_ = writeHuffHeaderWave(L, payload + lhSize, cap - lhSize, maxSym, tableLog); if (lane == 0) L.misc[0] = hs_; }
        __syncthreads();
        const uint32_t hsz = L.misc[0];
        if (ZS_STOP_AT(3)) FINISH(0, 0, 0);    // stop after codes + table description
        if (hsz) {
            const uint32_t seg = (nlit + 3) / 4;
            if (single) { if (wave == 0) { const uint32_t z = huffEncodeStream(L, L.u.tile[0], streams, litp, 0, nlit); if (lane == 0) L.misc[8] = z; } }
            else {
                const uint32_t len = (wave < 3) ? seg : nlit - 3 * seg;
                const uint32_t z = huffEncodeStream(L, L.u.tile[wave], streams + wave * ZS_STREAM_STRIDE, litp, wave * seg, len);
                if (lane == 0) L.misc[8 + wave] = z;
            }
            __syncthreads();
            const uint32_t ssz0 = L.misc[8], ssz1 = single ? 0 : L.misc[9], ssz2 = single ? 0 : L.misc[10], ssz3 = single ? 0 : L.misc[11];
            const bool ok = single || (ssz0 <= 65535 && ssz1 <= 65535 && ssz2 <= 65535 && ssz3 <= 65535);
            const uint32_t csz = hsz + (single ? ssz0 : 6 + ssz0 + ssz1 + ssz2 + ssz3);
            if (ok && lhSize + csz <= cap && csz + lhSize < nlit + (3 - (nlit < 32) - (nlit < 4096)) && (single || csz >= 10)) {
                uint8_t *op = payload + lhSize + hsz;
                if (!single) {
                    if (tid == 0) { op[0] = (uint8_t)ssz0; op[1] = (uint8_t)(ssz0 >> 8); op[2] = (uint8_t)ssz1; op[3] = (uint8_t)(ssz1 >> 8); op[4] = (uint8_t)ssz2; op[5] = (uint8_t)(ssz2 >> 8); }
                    op += 6;
                }
                const uint32_t ssz[4] = { ssz0, ssz1, ssz2, ssz3 };
                for (uint32_t k = 0; k < (single ? 1u : 4u); k++) {
                    const uint8_t *from = streams + k * ZS_STREAM_STRIDE;
                    zs_block_copy(op, from, ssz[k], tid, 256);
                    op += ssz[k];
                }
                if (tid == 0) {
                    if (lhSize == 3) { const uint32_t h = 2 + ((single ? 0u : 1u) << 2) + (nlit << 4) + (csz << 14); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); payload[2] = (uint8_t)(h >> 16); }
                    else if (lhSize == 4) { const uint32_t h = 2 + (2 << 2) + (nlit << 4) + (csz << 18); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); payload[2] = (uint8_t)(h >> 16); payload[3] = (uint8_t)(h >> 24); }
                    else { const uint32_t h = 2 + (3 << 2) + (nlit << 4) + (csz << 22); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); payload[2] = (uint8_t)(h >> 16); payload[3] = (uint8_t)(h >> 24); payload[4] = (uint8_t)(csz >> 10); }
                }
                litSecSize = lhSize + csz;
                done = true;
            }
        }
        }
    }
    if (!done) {
        const uint32_t lh = 1 + (nlit > 31) + (nlit > 4095);
        if (tid == 0) {
            if (lh == 1) payload[0] = (uint8_t)(nlit << 3);
            else if (lh == 2) { const uint32_t h = (1 << 2) + (nlit << 4); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); }
            else { const uint32_t h = (3 << 2) + (nlit << 4); payload[0] = (uint8_t)h; payload[1] = (uint8_t)(h >> 8); payload[2] = (uint8_t)(h >> 16); }
        }
        zs_block_copy(payload + lh, litp, nlit, tid, 256);
        litSecSize = lh + nlit;
    }
    FINISH(2, litSecSize, 0);
    #undef FINISH
}

// ---------------------------------------------------------------------------------------------
// k_encode_sequences : one wavefront per block.  Repcodes (parallel: two last-index scans), code
// histograms, tables (normalisation and encoding tables built by all lanes), bitstream
// -> seqSec[], meta.seqSecSize
// ---------------------------------------------------------------------------------------------
// normalise counts to 2^tableLog, all lanes (lane s owns symbol s); scalar statement: normalizeCounts in the oracle
__device__ static void normalizeCountsWave(int16_t *norm, uint32_t tableLog, const uint32_t *count, uint32_t total, uint32_t maxSym)
{
    const uint32_t lane = (uint32_t)zs_lane();
    const uint32_t tableSize = 1u << tableLog;
    const uint32_t c = (lane <= maxSym) ? count[lane] : 0u;
    uint32_t p = 0;
    if (c) {
        const uint64_t scaled = (uint64_t)c * tableSize;
        p = (uint32_t)(scaled / total);
        const uint32_t rem = (uint32_t)(scaled % total);
        if (2 * (uint64_t)rem >= total) p++;
        if (p == 0) p = 1;
    }
    int still = (int)tableSize - (int)wave_sum(p);
    // largest: first symbol holding the maximum
    { const uint32_t key = (p << 6) | (63u - lane); const uint32_t best = wave_max(key); const uint32_t li = 63u - (best & 63u);
      if (still > 0 && lane == li) p += (uint32_t)still; }
    while (still < 0) {
        const uint32_t key = (p > 1) ? ((p << 6) | (63u - lane)) : 0u;
        const uint32_t best = wave_max(key);
        const uint32_t li = 63u - (best & 63u);
        if (lane == li) p--;
        still++;
    }
    if (lane <= maxSym) norm[lane] = (int16_t)p;
    wave_sync();
}

// encoding table from a distribution without -1 entries, all lanes.  Cell order is the decoder's
// (ZStdDecompress.cs:993-1013): with no low-probability area the j-th laid cell is (j * step) mod size.
__device__ static void buildCTableWave(SeqLds &L, FseCT &ct, const int16_t *norm, uint32_t maxSym, uint32_t tableLog)
{
    const uint32_t lane = (uint32_t)zs_lane();
    const uint32_t tableSize = 1u << tableLog, tableMask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    const uint32_t nv = (lane <= maxSym) ? (uint32_t)norm[lane] : 0u;
    const uint32_t incl = wave_incl_scan(nv);
    const uint32_t excl = incl - nv;
    L.u.build.cumul[lane] = excl; if (lane == 63) L.u.build.cumul[64] = incl;
    L.u.build.symCount[lane] = excl;
    if (lane <= maxSym) {
        if (nv == 0) { ct.deltaNbBits[lane] = ((tableLog + 1) << 16) - (1u << tableLog); ct.deltaFindState[lane] = 0; }
        else if (nv == 1) { ct.deltaNbBits[lane] = (tableLog << 16) - (1u << tableLog); ct.deltaFindState[lane] = (int)excl - 1; }
        else { const uint32_t maxBitsOut = tableLog - zs_highbit(nv - 1); ct.deltaNbBits[lane] = (maxBitsOut << 16) - (nv << maxBitsOut); ct.deltaFindState[lane] = (int)excl - (int)nv; }
    }
    if (lane == 0) { ct.tableLog = tableLog; ct.rle = 0; }
    wave_sync();
    for (uint32_t j = lane; j < tableSize; j += 64) {
        // symbol owning slot j : last s with cumul[s] <= j
        uint32_t lo = 0, hi = maxSym + 1;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.u.build.cumul[mid] <= j) lo = mid; else hi = mid; }
        L.u.build.tableSymbol[(j * step) & tableMask] = (uint8_t)lo;
    }
    wave_sync();
    // stateTable[cumul[sym] + (rank of cell u among the cells of sym)] = tableSize + u, cells taken in ascending u.
    // 64 cells at a time: every lane ors its bit into its symbol's 64-bit lane mask (LDS); the mask read back gives the lane its rank
    // among the chunk's cells of that symbol (bits below it) and the symbol's count in the chunk, which the symbol's first lane adds to
    // the running count.  (A loop over the chunk's distinct symbols, a ballot each, was ~25 rounds of three LDS round trips per chunk:
    // most of the kernel's table stage.)
    uint32_t *symMask = L.u.build.symMask;
    for (uint32_t base = 0; base < tableSize; base += 64) {
        const uint32_t u = base + lane;
        const bool in = u < tableSize;
        const uint32_t sym = in ? L.u.build.tableSymbol[u] : 0u;
        symMask[2 * lane] = 0; symMask[2 * lane + 1] = 0;
        wave_sync();
        if (in) atomicOr(&symMask[2 * sym + (lane >> 5)], 1u << (lane & 31u));
        wave_sync();
        uint32_t start = 0, rank = 1, cnt = 0;
        if (in) {
            const uint32_t lo = symMask[2 * sym], hi = symMask[2 * sym + 1];
            const uint32_t belowLo = (lane < 32u) ? ((1u << lane) - 1u) : 0xFFFFFFFFu, belowHi = (lane < 32u) ? 0u : ((1u << (lane - 32u)) - 1u);
            rank = (uint32_t)__popc(lo & belowLo) + (uint32_t)__popc(hi & belowHi);
            cnt = (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
            start = L.u.build.symCount[sym];
            ct.stateTable[start + rank] = (uint16_t)(tableSize + u);
        }
        wave_sync();
        if (in && rank == 0) L.u.build.symCount[sym] = start + cnt;
        wave_sync();
    }
    wave_sync();
}

#ifndef ZS_SEQ_GROUP
#define ZS_SEQ_GROUP 4             // blocks (= wavefronts) per workgroup of the sequences kernel
#endif
#ifndef ZS_SEQ_TB
#define ZS_SEQ_TB 4                // tiles of 64 sequences whose records are requested together (pass 1, packing)
#endif
#define ZS_CHAIN_CODES 16384u      // most sequences a block can hold (64 output ranges of 256 record slots): elements per table in the chain scratch
#ifndef ZS_CHAIN_WARM
#define ZS_CHAIN_WARM 512u         // steps a chain segment starts ahead of its first output (a multiple of 16)
#endif
#ifndef ZS_CHAIN_MAXSEG
#define ZS_CHAIN_MAXSEG 21u        // most segments a chain is cut in (3 chains a wavefront: <= 21)
#endif
#ifndef ZS_CHAIN_MINSEG
#define ZS_CHAIN_MINSEG 4u         // shortest segment, in blocks of 16 steps
#endif
static_assert(3u * ZS_CHAIN_CODES <= ZS_BLOCK_MAX + 64u && 3u * 2u * ZS_CHAIN_CODES <= 4u * ZS_STREAM_STRIDE && ZS_CHAIN_CODES == ZS_WALK_RANGES * ZS_SEQ_PER_RANGE, "the chain scratch fits the buffers it borrows");
template <int G>
__global__ void __launch_bounds__(64 * G)
k_encode_sequences(const ZsBlockDesc *__restrict__ blocks, uint32_t nBlocks, const ZsSeqRec *__restrict__ seqAll, const ZsRangeHdr *__restrict__ hdrAll,
                   uint8_t *__restrict__ seqSecAll, ZsBlockMeta *__restrict__ metas, int stopAt, uint8_t *__restrict__ litsAll, uint8_t *__restrict__ streamAll,
                   uint2 *__restrict__ packRecAll)
{
    __shared__ SeqLds LS[G];
    const uint32_t wave = threadIdx.x >> 6;
    SeqLds &L = LS[wave];
    const uint32_t blk = blockIdx.x * G + wave;
    const uint32_t lane = (uint32_t)zs_lane();
    const bool exists = blk < nBlocks;
    const ZsBlockDesc bd = blocks[exists ? blk : 0];
    const uint32_t n = bd.size;
    const ZsSeqRec *seqBase = seqAll + (size_t)blk * ZS_WALK_RANGES * ZS_SEQ_PER_RANGE;
    const ZsRangeHdr *hdr = hdrAll + (size_t)blk * ZS_WALK_RANGES;
    uint8_t *out = seqSecAll + (size_t)blk * ZS_SEQSEC_STRIDE;         // 4-byte aligned
    const uint32_t cap = n + 512;
    // scratch of part 2's state chains, borrowed from the literals kernel (which runs after this one and writes both before it reads them):
    // a code byte a sequence and table in the block's literal buffer, 16 bits of chain output a sequence and table in its Huffman stream buffers
    uint8_t *chainCodes = litsAll + (size_t)blk * (ZS_BLOCK_MAX + 64);
    uint16_t *chainOuts = reinterpret_cast<uint16_t *>(streamAll + (size_t)blk * 4 * ZS_STREAM_STRIDE);
    uint2 *packRec = packRecAll + (size_t)blk * ZS_CHAIN_CODES;       // 8 bytes a sequence: its extra bits; the block's slot of the stage-1 distances (128 KiB), dead since the walk

    // ======== part 1, each wavefront on its own block (no workgroup barrier inside): header, recent-offset codes,
    //          histograms, tables.  result: section size so far / 0xFFFFFFFF = no compressed sequences section ========
    uint32_t result = 0xFFFFFFFFu, nseq = 0, bitstreamOff = 0;
    uint32_t secHdr = 0xFFFFFFFFu, secGap = 0;        // where the bitstream lies behind the headers (0xFFFFFFFF: no bitstream, the section is all headers)
    bool live = false;                                  // this wavefront has a bitstream to write in part 2
    do {
        if (!exists || n < 16) break;
        loadRangesWave(hdr, L.rngN, L.rngCarry, L.rngStart, nullptr, L.rngFirst, &L.misc[1], &L.misc[2]);
        for (uint32_t i = lane; i < 192; i += 64) L.count[i] = 0;
        wave_sync();
        nseq = L.rngStart[ZS_WALK_RANGES];
        const uint32_t *rngN = L.rngN, *rngCarry = L.rngCarry;

        // ---- sequences section header (inverse of DecodeSeqHeaders, ZStdDecompress.cs:1110-1180) ----
        uint32_t hdrBytes = (nseq < 128) ? 1 : (nseq < 0x7F00 ? 2 : 3);
        if (lane == 0) {
            if (nseq < 128) out[0] = (uint8_t)nseq;
            else if (nseq < 0x7F00) { out[0] = (uint8_t)((nseq >> 8) + 0x80); out[1] = (uint8_t)nseq; }
            else { out[0] = 0xFF; out[1] = (uint8_t)(nseq - 0x7F00); out[2] = (uint8_t)((nseq - 0x7F00) >> 8); }
        }
        if (nseq == 0) { result = hdrBytes; break; }

        // ---- pass 1: recent-offset codes + code histograms.  Rules (inverse of ZStdDecompress.cs:1509-1530):
        //   the state changes unless (ll > 0 and off == rep0); a change gives [off, rep0, off == rep1 ? rep2 : rep1].
        //   rep0 before a sequence is always the previous sequence's offset, so
        //   rep1 = previous offset of the last changing sequence, rep2 = rep1 as seen by the last changing sequence
        //   whose offset differed from its rep1: two "last index below me" scans per 64 sequences. ----
        {
            uint32_t cPrev, cA, cB;                       // carried: previous offset (= rep0), rep1, rep2
            if (bd.firstInChunk) { cPrev = 1; cA = 4; cB = 8; } else { cPrev = 0xFFFFFFF1u; cA = 0xFFFFFFF2u; cB = 0xFFFFFFF3u; }
            // sequences are taken 64 at a time in block order, whatever walk range they belong to; the records of the next 64 are
            // loaded while these are worked on (their addresses depend on nothing that is carried)
            auto locate = [&](uint32_t g, uint32_t &kOut, uint32_t &rrOut) -> const ZsSeqRec * {
                uint32_t rr = 0;
                #pragma unroll
                for (uint32_t stepb = ZS_WALK_RANGES / 2; stepb >= 1; stepb >>= 1) if (g >= L.rngStart[rr + stepb]) rr += stepb;
                kOut = g - L.rngStart[rr]; rrOut = rr;
                return seqBase + (size_t)rr * ZS_SEQ_PER_RANGE + L.rngFirst[rr] + kOut;
            };
            // (a record travels as its two raw words and is taken apart only where it is used: unpacked next to the load, the
            // compiler waits for the load on the spot).  ZS_SEQ_TB tiles of 64 are requested together and worked through one after the other:
            // a request a tile was a memory round trip a tile (the kernel's wavefronts spent half their time in s_waitcnt).
            constexpr uint32_t TB = ZS_SEQ_TB;
            uint2 rawN[TB]; uint32_t kN[TB], rrN[TB];
            auto request = [&](uint32_t base) {
                #pragma unroll
                for (uint32_t t = 0; t < TB; t++) {
                    const uint32_t g = base + 64u * t + lane;
                    rawN[t] = make_uint2(0, 0); kN[t] = 1; rrN[t] = 0;
                    if (g < nseq) { const ZsSeqRec *rp = locate(g, kN[t], rrN[t]); rawN[t] = *reinterpret_cast<const uint2 *>(rp); }
                }
            };
            request(0);
            #pragma unroll
            for (uint32_t t = 0; t < TB; t++) asm volatile("" : "+v"(rawN[t].x), "+v"(rawN[t].y));     // the first records are waited for here, not inside the loop
            for (uint32_t base0 = 0; base0 < nseq; base0 += 64u * TB) {
                uint2 rawC[TB]; uint32_t kC[TB], rrC[TB];
                #pragma unroll
                for (uint32_t t = 0; t < TB; t++) { rawC[t] = rawN[t]; kC[t] = kN[t]; rrC[t] = rrN[t]; }
                if (base0 + 64u * TB < nseq) request(base0 + 64u * TB);
                #pragma unroll
                for (uint32_t t = 0; t < TB; t++) {
                const uint32_t base = base0 + 64u * t;
                if (base >= nseq) break;
                const uint32_t g = base + lane;
                const bool in = g < nseq;
                const uint2 raw = rawC[t]; const uint32_t k = kC[t], rr = rrC[t];
                uint32_t off = 0, ll = 0, ml = 0;
                if (in) { off = zs_rec_off(raw.x, raw.y); ll = zs_rec_ll(raw.x); ml = zs_rec_ml(raw.x); if (k == 0) ll += rngCarry[rr]; }
                uint32_t prev = ZS_DPP(0, off, 0x138, 0xF, true); if (lane == 0) prev = cPrev;      // wave_shr:1        // rep0 before me
                const bool change = in && !(ll > 0 && off == prev);
                const int j = lastFlagBelow(change);                                                  // last changing sequence before me
                const uint32_t aSh = (uint32_t)__shfl((int)prev, max(j, 0));                          // every lane takes part: a source lane must be active
                const uint32_t a = (j >= 0) ? aSh : cA;                                               // rep1 before me
                const bool reset = change && (off != a);                                              // sequences after which rep2 = their rep1
                const int kk = lastFlagBelow(reset);
                const uint32_t bSh = (uint32_t)__shfl((int)a, max(kk, 0));
                const uint32_t b = (kk >= 0) ? bSh : cB;                                              // rep2 before me
                if (in) {
                    uint32_t val;
                    if (ll) { val = (off == prev) ? 1u : (off == a) ? 2u : (off == b) ? 3u : 0u; }
                    else    { val = (off == a) ? 1u : (off == b) ? 2u : 0u; }
                    const uint32_t v = val ? val : off + 3;
                    const uint32_t llc = llCodeOf(ll), ofc = zs_highbit(v), mlc = mlCodeOf(ml - 3);
                    atomicAdd(&L.count[llc], 1u);
                    atomicAdd(&L.count[64 + ofc], 1u);
                    atomicAdd(&L.count[128 + mlc], 1u);
                    // for part 2, in encoding order (last sequence first): the codes the state chains walk along, and the sequence's extra bits
                    // as they follow its three state outputs in the stream (literal length, match length, offset: <= 16 + 16 + 17 bits) with their count
                    const uint32_t je = nseq - 1u - g;
                    chainCodes[je] = (uint8_t)llc; chainCodes[ZS_CHAIN_CODES + je] = (uint8_t)ofc; chainCodes[2u * ZS_CHAIN_CODES + je] = (uint8_t)mlc;
                    uint32_t xbL, xbM;
                    const uint32_t xvL = llExtraOf(ll, xbL), xvM = mlExtraOf(ml - 3, xbM);
                    const uint64_t E = (uint64_t)xvL | ((uint64_t)xvM << xbL) | ((uint64_t)(v - (1u << ofc)) << (xbL + xbM));
                    packRec[je] = make_uint2((uint32_t)E, (uint32_t)(E >> 32) | ((xbL + xbM + ofc) << 24));
                }
                // carries for the next 64: state after the last sequence of this batch
                const uint32_t cnt = min(64u, nseq - base);
                const uint64_t chm = __ballot(change), rsm = __ballot(reset);
                const uint32_t lastOff = wave_get(off, (int)(cnt - 1));
                if (chm) { const int jl = 63 - __builtin_clzll(chm); cA = wave_get(prev, jl); }
                if (rsm) { const int kl = 63 - __builtin_clzll(rsm); cB = wave_get(a, kl); }
                cPrev = lastOff;
                }
            }
        }
        wave_sync();
        if (ZS_STOP_AT(1)) break;                         // timing aid (ZSMI_STOP_SEQ): stop after repcodes + histograms

        // ---- modes and tables ----
        uint32_t pos = hdrBytes + 1;                        // after nbSeq and the modes byte
        uint32_t modeByte = 0;
        bool fail = false;
        #pragma unroll 1
        for (uint32_t t = 0; t < 3; t++) {
            const uint32_t *count = L.count + 64 * t;
            const uint32_t maxCode = t == 0 ? MaxLL : (t == 1 ? MaxOff : MaxML);
            const uint32_t maxLog = t == 1 ? 8 : 9;
            const uint32_t defLog = t == 1 ? 5 : 6, defMax = t == 0 ? MaxLL : (t == 1 ? 28 : MaxML);
            FseCT &ct = L.ct[t];
            const uint32_t c = (lane <= maxCode) ? count[lane] : 0u;
            const uint64_t present = __ballot(c != 0);
            const uint32_t maxSym = 63u - (uint32_t)__builtin_clzll(present);       // nseq > 0: at least one symbol
            const uint32_t largest = wave_max(c);
            uint32_t mode;
            if (largest == nseq) {
                mode = 1;
                if (pos >= cap) { fail = true; break; }
                if (lane == 0) { out[pos] = (uint8_t)maxSym; ct.rle = 1; ct.tableLog = 0; }
                pos += 1;
            } else if (nseq < 64 && maxSym <= defMax) {
                mode = 0;
                if (lane == 0) {
                    const int16_t *defNorm = t == 0 ? c_LL_defaultNorm : (t == 1 ? c_OF_defaultNorm : c_ML_defaultNorm);
                    for (uint32_t i = 0; i <= defMax; i++) L.norm[i] = defNorm[i];
                    buildCTable(ct, L.u.build.tableSymbol, L.u.build.cumul, L.norm, defMax, defLog);
                }
            } else {
                uint32_t tableLog = maxLog;
                { const uint32_t hb = zs_highbit(nseq - 1); const uint32_t want = hb > 2 ? hb - 2 : 5; if (want < tableLog) tableLog = want; }
                { const uint32_t minBits = zs_highbit(maxSym) + 2; const uint32_t npresent = (uint32_t)__popcll(present);
                  if (tableLog < minBits) tableLog = minBits; while ((1u << tableLog) < npresent) tableLog++; }
                if (tableLog < 5) tableLog = 5;
                if (tableLog > maxLog) tableLog = maxLog;
                normalizeCountsWave(L.norm, tableLog, count, nseq, maxSym);
                if (lane == 0) L.misc[0] = writeNCount(out + pos, cap - pos, L.norm, maxSym, tableLog);
                wave_sync();
                const uint32_t h = L.misc[0];
                if (!h) { fail = true; break; }
                pos += h;
                mode = 2;
                buildCTableWave(L, ct, L.norm, maxSym, tableLog);
            }
            modeByte |= mode << (6 - 2 * t);
            wave_sync();
        }
        if (fail) break;
        if (ZS_STOP_AT(2)) break;                         // stop after the tables
        if (lane == 0) out[hdrBytes] = (uint8_t)modeByte;
        bitstreamOff = pos;
        live = true;
    } while (0);

    // ======== part 2: sequences bitstream (inverse of ZStdDecompress.cs:1473-1608), last sequence first; each wavefront on its own block,
    //   no workgroup barrier.  An FSE state chain is a dependent table look-up per sequence (~3600 a block and table), and the format fixes
    //   ONE chain per table.  But chains over the same symbols COUPLE: started from any state they agree with the true chain after some
    //   steps for good (a step keeps state >> nbBits: tools/lab/coupling.c measures half of them coupled after 60 steps, 99.5 % after 512 on
    //   the Zipf log, 96.5 % on the mixed corpus).  So a table's chain is cut in K <= 21 SEGMENTS of whole blocks of 16 steps; lane c K + k runs
    //   segment k of table c, started ZS_CHAIN_WARM steps early from the state its first symbol alone gives (the rule of the block's true
    //   first step), silent until its own segment begins; then the seams are checked in order - entry state of segment k == exit
    //   state of segment k - 1 - and a segment that fails is run again from the right state (its outputs overwritten) before the
    //   next seam is looked at, so the result is exactly the serial chain's.  63 lanes run ~(nseq / K + 512) steps instead of 3 lanes nseq.
    //   Codes come from pass 1 (a byte a step and table, in encoding order), outputs go to scratch as 16 bits a step
    //   (state bits | count << 12); both live in the literals kernel's scratch (streams, literals), which that kernel writes after this one.
    //   Then the tiles of 64 sequences are packed, all lanes. ========
    uint8_t *bsTmp = out + ((bitstreamOff + 3u) & ~3u);                // aligned start inside the section buffer
    BitSink sink; sink_init(sink, bsTmp, L.tile);
    const uint32_t bsCap = ZS_SEQSEC_STRIDE - ((bitstreamOff + 3u) & ~3u) - 1024u;
    bool overflow = false;
    if (live && !ZS_STOP_AT(3)) {                                      // stopAt 3: timing aid, no chains
        const uint32_t nFull = nseq >> 4, tail = nseq & 15u;           // whole blocks of 16 steps; the steps behind them belong to the last segment
        uint32_t Sb = max((nFull + ZS_CHAIN_MAXSEG - 1u) / ZS_CHAIN_MAXSEG, (uint32_t)ZS_CHAIN_MINSEG);   // blocks a segment
        const uint32_t K = max((nFull + Sb - 1u) / Sb, 1u);            // segments a chain (<= 21)
        const uint32_t c = (lane >= K ? 1u : 0u) + (lane >= 2u * K ? 1u : 0u), k = lane - c * K;
        const bool chainLane = lane < 3u * K && !L.ct[min(c, 2u)].rle;
        const FseCT &ct = L.ct[min(c, 2u)];
        const uint8_t *codes = chainCodes + (size_t)min(c, 2u) * ZS_CHAIN_CODES;
        uint16_t *outs = chainOuts + (size_t)min(c, 2u) * ZS_CHAIN_CODES;
        typedef const __attribute__((address_space(3))) uint16_t *LdsU16;
        typedef const __attribute__((address_space(3))) uint32_t *LdsU32;
        // a code's two operands in one word: deltaNbBits < 2^20 (a state added to it stays below), deltaFindState + 1024 in the bits above
        #pragma unroll
        for (uint32_t t = 0; t < 3; t++) L.u.chain.op[t][lane] = L.ct[t].deltaNbBits[lane] | ((uint32_t)(L.ct[t].deltaFindState[lane] + 1024) << 20);
        wave_sync();
        const uint32_t tabAddr = (uint32_t)(uintptr_t)(LdsU16)ct.stateTable - 2048u, opAddr = (uint32_t)(uintptr_t)(LdsU32)L.u.chain.op[min(c, 2u)];
        const uint32_t firstOut = k * Sb, bEnd = min(firstOut + Sb, nFull);
        const bool lastSeg = k + 1u == K;
        const uint32_t bWarm = firstOut > ZS_CHAIN_WARM / 16u ? firstOut - ZS_CHAIN_WARM / 16u : 0u;
        // one step: the state's low bits leave (their count from the sum's high half), the rest picks the next state from the symbol's stretch of the table
        #define CHAIN_STEP(sym, o) { \
            const uint32_t op_ = *(LdsU32)(uintptr_t)(opAddr + ((sym) << 2)); \
            const uint32_t nbo_ = __builtin_amdgcn_ubfe(state + op_, 16, 4); \
            (o) = __builtin_amdgcn_ubfe(state, 0, nbo_) | (nbo_ << 12); \
            state = *(LdsU16)(uintptr_t)(tabAddr + (((state >> nbo_) + (op_ >> 20)) << 1)); }
        // the blocks [from, to) with the state carried; outputs stored for blocks >= outFrom; entry: the state in front of block `mark`.
        // Loads and stores share the wavefront's vmcnt, and the compiler waits vmcnt(0) for a load it sees - behind a round's stores that is their
        // whole trip to L2, every round.  So the codes of round r + 1 are requested BEFORE round r's two stores, by a load the compiler does not
        // see, and waited for by count: vmcnt(2) leaves exactly those stores in flight (in-order return: DESIGN section 6).  For the count to
        // hold every lane stores every round - where it has nothing to store, into a junk slot of the section buffer (written by the packing later).
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint8_t *junk = out + 32768u + 32u * lane;
        auto runBlocks = [&](uint32_t from, uint32_t to, uint32_t outFrom, uint32_t mark, uint32_t &state, uint32_t &entry, uint32_t rounds) {
            u32x4 cw;
            {
                const uint8_t *a0 = codes + 16u * min(from, nFull ? nFull - 1u : 0u);
                asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(cw) : "v"(a0) : "memory");
            }
            for (uint32_t it = 0; it < rounds; it++) {
                const uint32_t blkI = from + it;
                const bool act = blkI < to;
                const u32x4 cur = cw;
                {
                    const uint8_t *a1 = codes + 16u * min(blkI + 1u, nFull ? nFull - 1u : 0u);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cw) : "v"(a1) : "memory");
                }
                if (act && blkI == mark) entry = state;
                uint32_t o[16] = {};
                if (act) {
                    const uint32_t w4[4] = { cur.x, cur.y, cur.z, cur.w };
                    #pragma unroll
                    for (uint32_t t = 0; t < 16; t++) { const uint32_t sym = (w4[t >> 2] >> (8u * (t & 3u))) & 0xFFu; CHAIN_STEP(sym, o[t]) }
                }
                uint4 *op_ = reinterpret_cast<uint4 *>((act && blkI >= outFrom) ? reinterpret_cast<uint8_t *>(outs + 16u * blkI) : junk);
                op_[0] = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
                op_[1] = make_uint4(o[8] | (o[9] << 16), o[10] | (o[11] << 16), o[12] | (o[13] << 16), o[14] | (o[15] << 16));
                asm volatile("s_waitcnt vmcnt(2)" : "+v"(cw) :: "memory");
            }
        };
        // the same without outputs (a segment's warm-up: most of a lane's steps): lanes run the LAST `n` of `rounds` rounds, blocks [to - n, to)
        auto warmBlocks = [&](uint32_t to, uint32_t n, uint32_t &state, uint32_t rounds) {
            u32x4 cw;
            {
                const uint8_t *a0 = codes + 16u * min(to - n, nFull ? nFull - 1u : 0u);
                asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(cw) : "v"(a0) : "memory");
            }
            for (uint32_t it = 0; it < rounds; it++) {
                const bool act = it + n >= rounds;                      // my first round is rounds - n
                const uint32_t blkI = to - rounds + it;                 // (meaningful where act)
                const u32x4 cur = cw;
                {
                    const uint8_t *a1 = codes + 16u * (act ? min(blkI + 1u, nFull ? nFull - 1u : 0u) : min(to - n, nFull ? nFull - 1u : 0u));
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cw) : "v"(a1) : "memory");
                }
                if (act) {
                    const uint32_t w4[4] = { cur.x, cur.y, cur.z, cur.w };
                    #pragma unroll
                    for (uint32_t t = 0; t < 16; t++) {
                        const uint32_t sym = (w4[t >> 2] >> (8u * (t & 3u))) & 0xFFu;
                        const uint32_t op_ = *(LdsU32)(uintptr_t)(opAddr + (sym << 2));
                        const uint32_t nbo_ = __builtin_amdgcn_ubfe(state + op_, 16, 4);
                        state = *(LdsU16)(uintptr_t)(tabAddr + (((state >> nbo_) + (op_ >> 20)) << 1));
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(cw) :: "memory");
            }
        };
        auto runTail = [&](uint32_t &state) {                           // the steps behind the whole blocks, one by one
            for (uint32_t t = 0; t < tail; t++) { const uint32_t sym = codes[16u * nFull + t]; uint32_t o; CHAIN_STEP(sym, o) outs[16u * nFull + t] = (uint16_t)o; }
        };
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");          // pass 1's code bytes (other lanes' stores) are read below
#ifdef ZS_CHAIN_COUNT
        if (lane == 0) { metas[blk].pad[0] = 0; metas[blk].pad[1] = K | (Sb << 8) | (nseq << 16); }
#endif
        uint32_t state = 0, entry = 0;
        const bool runs = chainLane && (firstOut < nFull || k == 0);
        if (runs) {
            // the first step of a chain takes its state from the symbol alone (ZStdCompress' first-symbol rule; the decoder's final state); as a
            // step from a made-up state: nbBits = (deltaNbBits + 2^15) >> 16, state = (nbBits << 16) - deltaNbBits
            const uint32_t sym0 = codes[16u * bWarm];
            const uint32_t dnb0 = ct.deltaNbBits[sym0], nb0 = (dnb0 + (1u << 15)) >> 16;
            state = (nb0 << 16) - dnb0;
        }
        const uint32_t warmN = runs ? firstOut - bWarm : 0u;            // silent blocks in front of my segment
        const uint32_t warmRounds = wave_max(warmN), rounds = wave_max(runs ? bEnd - firstOut : 0u);
        if (warmRounds) warmBlocks(firstOut, warmN, state, warmRounds);
        entry = state;
        if (runs) {
            uint32_t dummy = 0;
            runBlocks(firstOut, bEnd, firstOut, 0xFFFFFFFFu, state, dummy, rounds);
            if (lastSeg && tail) runTail(state);
        }
        // the seams, in order
        for (uint32_t kk = 1; kk < K; kk++) {
            const uint32_t prevExit = ZS_DPP(0, state, 0x138, 0xF, true);           // wave_shr:1: lane c K + kk - 1
            const bool bad = runs && k == kk && entry != prevExit;
            if (__any(bad)) {
#ifdef ZS_CHAIN_COUNT
                if (lane == 0) metas[blk].pad[0] += (uint32_t)__popcll(__ballot(bad)) | (1u << 16);      // development aid: seams that failed / rounds of repair
#endif
                // (a repair is the chain run serially with one to three lanes: its wavefront issues ahead of the others on its SIMD, as the shared chain
                //  wavefront of rounds 1 - 3 did)
                __builtin_amdgcn_s_setprio(3);
                if (bad) {
                    uint32_t dummy = 0;
                    state = prevExit;
                    runBlocks(firstOut, bEnd, firstOut, 0xFFFFFFFFu, state, dummy, bEnd - firstOut);
                    if (lastSeg && tail) runTail(state);
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        #undef CHAIN_STEP
        if (chainLane && k == 0) outs[0] = 0;                           // the first step of all writes no bits
        // final states of the three tables -> misc[4..6] (an RLE table has none)
        if (lane < 3u * K && k + 1u == K) L.misc[4 + c] = state;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");          // the outputs (other lanes' stores) are read below
        wave_sync();
    }
    if (live && !ZS_STOP_AT(3) && !ZS_STOP_AT(4)) {                     // stopAt 4: timing aid, no packing
        // per sequence: the three chains' outputs of its step and its extra bits (pass 1), all in encoding order: consecutive lanes, consecutive
        // addresses.  ZS_SEQ_TB tiles are requested together (clamped, branch-free) and packed one after the other.
        constexpr uint32_t TB = ZS_SEQ_TB;
        const bool rle0 = L.ct[0].rle != 0, rle1 = L.ct[1].rle != 0, rle2 = L.ct[2].rle != 0;
        uint2 prN[TB]; uint32_t sN[TB][3];
        auto request = [&](uint32_t j0) {
            #pragma unroll
            for (uint32_t t = 0; t < TB; t++) {
                const uint32_t j = min(j0 + 64u * t + lane, nseq - 1u);
                prN[t] = packRec[j];
                sN[t][0] = chainOuts[j]; sN[t][1] = chainOuts[ZS_CHAIN_CODES + j]; sN[t][2] = chainOuts[2u * ZS_CHAIN_CODES + j];
            }
        };
        request(0);
        for (uint32_t j0 = 0; j0 < nseq && !overflow; j0 += 64u * TB) {
            uint2 pr[TB]; uint32_t sC[TB][3];
            #pragma unroll
            for (uint32_t t = 0; t < TB; t++) { pr[t] = prN[t]; sC[t][0] = sN[t][0]; sC[t][1] = sN[t][1]; sC[t][2] = sN[t][2]; }
            if (j0 + 64u * TB < nseq) request(j0 + 64u * TB);
            #pragma unroll
            for (uint32_t t = 0; t < TB; t++) {
                const uint32_t jt = j0 + 64u * t;
                if (jt >= nseq || overflow) break;
                uint64_t lo = 0; uint32_t hi = 0, nb = 0;
                if (jt + lane < nseq) {
                    const uint32_t sLL = rle0 ? 0u : sC[t][0], sOF = rle1 ? 0u : sC[t][1], sML = rle2 ? 0u : sC[t][2];
                    #define PUTB(v, b) { const uint32_t b_ = (b); if (b_) { const uint64_t v_ = (uint64_t)(v); if (nb < 64) { lo |= v_ << nb; if (nb + b_ > 64) hi |= (uint32_t)(v_ >> (64 - nb)); } else hi |= (uint32_t)(v_ << (nb - 64)); nb += b_; } }
                    PUTB(sOF & 0xFFFu, sOF >> 12);
                    PUTB(sML & 0xFFFu, sML >> 12);
                    PUTB(sLL & 0xFFFu, sLL >> 12);
                    PUTB((uint64_t)pr[t].x | ((uint64_t)(pr[t].y & 0xFFFFFFu) << 32), pr[t].y >> 24);
                    #undef PUTB
                }
                if (sink.bitpos / 8 + 1024 > bsCap) overflow = true;
                else sink_put(sink, lo, hi, nb);
            }
        }
    }
    if (live && !overflow) {
        {
            // final states: ML, OF, LL (ZStdDecompress.cs:1578-1580 reads LL, OF, ML)
            const uint32_t tlLL = L.ct[0].rle ? 0 : L.ct[0].tableLog;
            const uint32_t tlOF = L.ct[1].rle ? 0 : L.ct[1].tableLog;
            const uint32_t tlML = L.ct[2].rle ? 0 : L.ct[2].tableLog;
            const uint32_t stLL = L.misc[4], stOF = L.misc[5], stML = L.misc[6];
            uint64_t lo = 0; uint32_t nb = 0;
            if (lane == 0) {
                if (tlML) { lo |= (uint64_t)(stML & ((1u << tlML) - 1)) << nb; nb += tlML; }
                if (tlOF) { lo |= (uint64_t)(stOF & ((1u << tlOF) - 1)) << nb; nb += tlOF; }
                if (tlLL) { lo |= (uint64_t)(stLL & ((1u << tlLL) - 1)) << nb; nb += tlLL; }
            }
            sink_put(sink, lo, 0u, nb);
        }
        const uint32_t bsSize = sink_close(sink);
        // the stream stays where it was built (4-byte aligned, up to 3 bytes behind the headers): the frame assembly copies the two
        // pieces next to each other (moving it down here was ~160 dependent load -> store rounds per block)
        const uint32_t gap = (uint32_t)(bsTmp - (out + bitstreamOff));
        secHdr = bitstreamOff; secGap = gap;
        const uint32_t total = bitstreamOff + bsSize;
        result = total > cap ? 0xFFFFFFFFu : total;
    }
    if (lane == 0 && exists) { metas[blk].seqSecSize = result; metas[blk].seqHdrSize = (secHdr == 0xFFFFFFFFu) ? result : secHdr; metas[blk].seqGap = secGap; }
}

// ---------------------------------------------------------------------------------------------
// k_assemble_frames : one workgroup per chunk.  frame = magic + FHD + FCS (single segment)
// + blocks  (inverse of ZStdDecompress.cs:421-499, 646-659, 2008-2091).  A block is emitted compressed
// iff both sections exist and literal section + sequence section < block size (else raw; RLE if flagged).
// ---------------------------------------------------------------------------------------------

extern "C" __global__ void __launch_bounds__(256)
k_assemble_frames(const uint8_t *__restrict__ src, const ZsChunkDesc *__restrict__ chunks, const ZsBlockDesc *__restrict__ blocks,
                  const ZsBlockMeta *__restrict__ metas, const uint8_t *__restrict__ litSecAll, const uint8_t *__restrict__ seqSecAll,
                  uint32_t blockBase, uint8_t *__restrict__ dst, uint32_t *__restrict__ dstSizes, uint32_t chunkBase)
{
    const ZsChunkDesc cd = chunks[chunkBase + blockIdx.x];
    if (cd.nBlocks <= 1) return;                               // one-block chunks were assembled by the literals kernel
    uint8_t *out = dst + cd.dstOff;
    const uint32_t tid = threadIdx.x;
    uint32_t pos = zs_frame_header(out, cd.size, tid == 0);
    for (uint32_t b = 0; b < cd.nBlocks; b++) {
        const uint32_t gb = cd.firstBlock + b;             // global block index
        const uint32_t lb = gb - blockBase;                // index inside this sub-batch's scratch
        const ZsBlockDesc bd = blocks[gb];
        pos += zs_emit_block(out, pos, src + bd.srcOff, bd.size, (b + 1 == cd.nBlocks) ? 1u : 0u, metas[lb],
                             litSecAll + (size_t)lb * ZS_LITSEC_STRIDE, seqSecAll + (size_t)lb * ZS_SEQSEC_STRIDE, tid, blockDim.x);
    }
    if (tid == 0) dstSizes[chunkBase + blockIdx.x] = pos;
}
