// LZ stage of the block encoder: candidate search (k_lz_candidates) and greedy walk with
// look-ahead (k_lz_walk).  The scalar statement of the same algorithm is oracle/zso_encoder.c
// (findCandidates / walkRange); the two must agree bit for bit.
//
// There is no reference code for this stage (the reference has no encoder);
// what it emits is consumed by entropy_kernels.hip, whose output the reference decoder must accept.
#include "zsmi_device.h"

// ---------------------------------------------------------------------------------------------
// k_lz_candidates : one workgroup (8 wavefronts) per block, wavefront w owns range w (8 KiB).
// LDS: 8 hash tables of 2^hashLog 16-bit entries (position + 1).  Positions are taken 64 at a
// time: all lanes read the table, then all lanes write it (same-bucket writes of one instruction:
// the highest lane stays -- probed on MI355X by tools/probe/lds_order.hip).
// Phase A fills the tables and leaves each position's own-range predecessor in dist[];
// phase B (after a barrier: earlier ranges' tables are final) falls back to the nearest earlier
// range that has the hash, checks the 4 bytes and writes the match distance.
// HBM/L2 traffic per block: reads n (twice, second time from cache) + n gathers; writes 2n (dist twice).
// ---------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(512)
k_lz_candidates(const uint8_t *__restrict__ src, const ZsBlockDesc *__restrict__ blocks,
                uint16_t *__restrict__ distAll, int hashLog)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t tables[];
    const ZsBlockDesc bd = blocks[blockIdx.x];
    const uint8_t *s = src + bd.srcOff;
    const uint32_t n = bd.size;
    uint16_t *dist = distAll + (size_t)blockIdx.x * ZS_BLOCK_MAX;

    {   // clear the tables
        const uint32_t words = (ZS_MAX_RANGES << hashLog) >> 1;      // 32-bit words
        uint32_t *t32 = reinterpret_cast<uint32_t *>(tables);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) t32[i] = 0;
    }
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t start = wave << ZS_RANGE_LOG;
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;                   // positions [0, hashable) have 4 bytes
    const uint32_t end = min(start + ZS_RANGE_SIZE, hashable);
    uint16_t *T = tables + ((size_t)wave << hashLog);

    // 8 steps per trip: the 8 loads of a trip are issued together, then the table is visited in step order
    constexpr uint32_t U = 8;
    for (uint32_t base = start; base < end; base += 64 * U) {
        uint32_t v[U];
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t p = base + u * 64 + lane; v[u] = (p < end) ? zs_load32(s + p) : 0u; }
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t p = base + u * 64 + lane;
            if (p < end) {
                const uint32_t h = zs_hash4(v[u], hashLog);
                const uint16_t own = T[h];
                T[h] = (uint16_t)(p + 1);
                dist[p] = own;
            }
        }
    }
    __syncthreads();

    for (uint32_t base = start; base < end; base += 64 * U) {
        uint32_t v[U], cand[U], cv[U];
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t p = base + u * 64 + lane; const bool in = p < end; v[u] = in ? zs_load32(s + p) : 0u; cand[u] = in ? (uint32_t)dist[p] : 0u; }
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t p = base + u * 64 + lane;
            if (p < end && !cand[u]) {
                // all earlier ranges are read at once (independent LDS reads); the nearest one that has the hash wins
                const uint32_t h = zs_hash4(v[u], hashLog);
                uint32_t c[ZS_MAX_RANGES - 1];
                #pragma unroll
                for (uint32_t q = 0; q < ZS_MAX_RANGES - 1; q++) c[q] = (q < wave) ? (uint32_t)tables[((size_t)q << hashLog) + h] : 0u;
                #pragma unroll
                for (uint32_t q = 0; q < ZS_MAX_RANGES - 1; q++) if (c[q]) cand[u] = c[q];
            }
        }
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) cv[u] = cand[u] ? zs_load32(s + cand[u] - 1) : 0u;
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t p = base + u * 64 + lane;
            if (p < end) dist[p] = (cand[u] && cv[u] == v[u]) ? (uint16_t)(p - (cand[u] - 1)) : (uint16_t)0;
        }
    }
    // positions without 4 bytes left: no candidate
    {
        const uint32_t rend = min(start + ZS_RANGE_SIZE, n);
        for (uint32_t p = max(end, start) + lane; p < rend; p += 64) dist[p] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_lz_walk : one wavefront per range.  Each step the 64 lanes look at dist[ip .. ip+64), the
// first LOOK positions holding a candidate are handed to groups of 8 lanes; a group compares
// 64 bytes forward and 32 bytes backward (into the pending literals) in one round of loads,
// scores its candidate, the best one becomes a sequence.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load64_clamped(const uint8_t *s, int32_t pos, uint32_t n)
{
    // bytes s[pos .. pos+8) with positions >= n read as 0 (pos >= 0, n >= 8)
    if ((uint32_t)pos + 8 <= n) return zs_load64(s + pos);
    if ((uint32_t)pos >= n) return 0;
    const uint32_t sh = (uint32_t)pos + 8 - n;            // 1..7 bytes beyond the end
    return zs_load64(s + n - 8) >> (8 * sh);
}
__device__ __forceinline__ uint32_t load32_back(const uint8_t *s, int32_t pos)
{
    // bytes s[pos .. pos+4) with positions < 0 read as 0 (pos + 4 > 0 is not required)
    if (pos >= 0) return zs_load32(s + pos);
    if (pos <= -4) return 0;
    return zs_load32(s) << (8 * (uint32_t)(-pos));
}

extern "C" __global__ void __launch_bounds__(64)
k_lz_walk(const uint8_t *__restrict__ src, const ZsBlockDesc *__restrict__ blocks,
          const uint16_t *__restrict__ distAll, ZsSeqRec *__restrict__ seqAll, ZsRangeHdr *__restrict__ hdrAll,
          int look)
{
    const uint32_t blk = blockIdx.x >> 3, range = blockIdx.x & 7u;
    const ZsBlockDesc bd = blocks[blk];
    const uint8_t *s = src + bd.srcOff;
    const uint32_t n = bd.size;
    const uint16_t *dist = distAll + (size_t)blk * ZS_BLOCK_MAX;
    ZsSeqRec *seqs = seqAll + ((size_t)blk * ZS_MAX_RANGES + range) * ZS_SEQ_PER_RANGE;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t start = range << ZS_RANGE_LOG;
    if (start >= n) { if (lane == 0) { hdrAll[blockIdx.x].nseq = 0; hdrAll[blockIdx.x].trailing = 0; } return; }
    const uint32_t end = min(start + ZS_RANGE_SIZE, n);
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;
    const uint32_t scanEnd = min(end, hashable);
    const uint32_t grp = lane >> 3, sub = lane & 7u;

    uint32_t ip = start, anchor = start, nseq = 0;
    while (ip < scanEnd) {
        const uint32_t wend = min(ip + ZS_WINDOW, scanEnd);
        const uint32_t q0 = ip + lane;
        const uint32_t d = (q0 < wend) ? dist[q0] : 0;
        uint64_t m = __ballot(d != 0);
        if (m == 0) { ip = wend; continue; }
        // lane index of the grp-th candidate
        uint32_t ncand = (uint32_t)__popcll(m);
        if (ncand > (uint32_t)look) ncand = (uint32_t)look;
        uint32_t selLane = 0;
        {
            uint64_t mm = m;
            #pragma unroll
            for (uint32_t g = 0; g < 8; g++) {
                const uint32_t l = mm ? (uint32_t)__builtin_ctzll(mm) : 0;
                if (g == grp) selLane = l;
                mm &= mm - 1;
            }
        }
        const bool active = grp < ncand;
        const uint32_t off = (uint32_t)__shfl((int)d, (int)selLane);
        const uint32_t q = ip + selLane;
        // one round of loads: 8 bytes forward per lane, 4 bytes backward per lane
        uint32_t nbF = 0, nbB = 0;
        if (active) {
            const uint32_t cap = min(end - q, ZS_FCAP);
            const int32_t fo = (int32_t)(8 * sub);
            if ((uint32_t)fo < cap) {
                const uint64_t a = load64_clamped(s, (int32_t)q + fo, n);
                const uint64_t b = load64_clamped(s, (int32_t)(q - off) + fo, n);
                const uint64_t x = a ^ b;
                nbF = x ? ((uint32_t)__builtin_ctzll(x) >> 3) : 8u;
                nbF = min(nbF, cap - (uint32_t)fo);
            }
            const uint32_t maxBack = min(min(q - anchor, q - off), ZS_BCAP);
            const uint32_t bo = 4 * sub;                                   // this lane covers back distances bo+1 .. bo+4
            if (bo < maxBack) {
                const uint32_t a = load32_back(s, (int32_t)q - (int32_t)bo - 4);
                const uint32_t b = load32_back(s, (int32_t)(q - off) - (int32_t)bo - 4);
                const uint32_t x = a ^ b;
                nbB = x ? ((uint32_t)__builtin_clz(x) >> 3) : 4u;
                nbB = min(nbB, maxBack - bo);
            }
        }
        // group reductions through ballots: first lane of the group that stopped early
        const uint64_t stopF = __ballot(nbF < 8u);
        const uint64_t stopB = __ballot(nbB < 4u);
        const uint32_t gF = (uint32_t)((stopF >> (8 * grp)) & 0xFFu);
        const uint32_t gB = (uint32_t)((stopB >> (8 * grp)) & 0xFFu);
        const uint32_t fF = gF ? (uint32_t)__builtin_ctz(gF) : 8u;
        const uint32_t fB = gB ? (uint32_t)__builtin_ctz(gB) : 8u;
        const uint32_t partF = (uint32_t)__shfl((int)nbF, (int)(8 * grp + (fF & 7u)));
        const uint32_t partB = (uint32_t)__shfl((int)nbB, (int)(8 * grp + (fB & 7u)));
        const uint32_t fwd = (fF < 8u) ? 8u * fF + partF : ZS_FCAP;
        const uint32_t back = (fB < 8u) ? 4u * fB + partB : ZS_BCAP;
        // note: lanes beyond cap / maxBack report nb == 0 < 8/4, so they stop the count where the limit is
        int key = 0;
        if (active && fwd >= ZS_MINMATCH) {
            const int gain = (int)(fwd + back) * 4 - (int)zs_highbit(off + 1) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
            key = ((gain + 2048) << 3) | (int)(7u - grp);
        }
        // wave max over groups (all lanes of a group hold the same key)
        int best = key;
        best = max(best, __shfl_xor(best, 8));
        best = max(best, __shfl_xor(best, 16));
        best = max(best, __shfl_xor(best, 32));
        if (best == 0) { ip = wend; continue; }
        const uint32_t bg = 7u - (uint32_t)(best & 7);
        const uint32_t bq = (uint32_t)__shfl((int)q, (int)(8 * bg));
        const uint32_t boff = (uint32_t)__shfl((int)off, (int)(8 * bg));
        uint32_t bfwd = (uint32_t)__shfl((int)fwd, (int)(8 * bg));
        const uint32_t bback = (uint32_t)__shfl((int)back, (int)(8 * bg));
        if (bfwd == ZS_FCAP) {
            // long match: the whole wavefront extends it, 512 bytes per round
            uint32_t pos = bq + ZS_FCAP;
            while (pos < end) {
                const uint32_t cap = end - pos;
                const uint32_t fo = 8 * lane;
                uint32_t nb = 0;
                if (fo < cap) {
                    const uint64_t a = load64_clamped(s, (int32_t)(pos + fo), n);
                    const uint64_t b = load64_clamped(s, (int32_t)(pos - boff + fo), n);
                    const uint64_t x = a ^ b;
                    nb = x ? ((uint32_t)__builtin_ctzll(x) >> 3) : 8u;
                    nb = min(nb, cap - fo);
                }
                const uint64_t stop = __ballot(nb < 8u);
                if (stop) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(stop);
                    pos += 8 * f + (uint32_t)__shfl((int)nb, (int)f);
                    break;
                }
                pos += 512;
            }
            bfwd = pos - bq;
        }
        if (lane == 0) {
            ZsSeqRec r;
            r.ll = (uint16_t)(bq - bback - anchor); r.ml = (uint16_t)(bback + bfwd); r.off = (uint16_t)boff; r.flags = (uint16_t)(bq - bback);   // flags: position of the match start
            seqs[nseq] = r;
        }
        nseq++;
        ip = bq + bfwd; anchor = ip;
    }
    if (lane == 0) { hdrAll[blockIdx.x].nseq = nseq; hdrAll[blockIdx.x].trailing = end - anchor; }
}
