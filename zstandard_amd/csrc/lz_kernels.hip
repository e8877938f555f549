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

    // 8 steps per trip.  The loads of trip t+1 are issued before trip t is worked on (registers double-buffered),
    // so the table walk of a trip runs under the memory latency of the next one.
    constexpr uint32_t U = 8;
    {
        uint32_t v[U], vn[U];
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t p = start + u * 64 + lane; v[u] = (p < end) ? zs_load32(s + p) : 0u; }
        for (uint32_t base = start; base < end; base += 64 * U) {
            const uint32_t nbase = base + 64 * U;
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t p = nbase + u * 64 + lane; vn[u] = (p < end) ? zs_load32(s + p) : 0u; }
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
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) v[u] = vn[u];
        }
    }
    __syncthreads();

    {
        // trip t: (v, own candidate) loaded one trip ahead; its verification gathers are issued, then trip t-1's
        // gathers (issued one trip earlier) are compared and stored.
        uint32_t v[U], cand[U], vn[U], candn[U], pv[U], pcand[U], pcv[U];
        uint32_t pbase = 0; bool havePrev = false;
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t p = start + u * 64 + lane; const bool in = p < end; v[u] = in ? zs_load32(s + p) : 0u; cand[u] = in ? (uint32_t)dist[p] : 0u; }
        for (uint32_t base = start; base < end; base += 64 * U) {
            const uint32_t nbase = base + 64 * U;
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t p = nbase + u * 64 + lane; const bool in = p < end; vn[u] = in ? zs_load32(s + p) : 0u; candn[u] = in ? (uint32_t)dist[p] : 0u; }
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
            uint32_t cv[U];
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) cv[u] = cand[u] ? zs_load32(s + cand[u] - 1) : 0u;
            if (havePrev) {
                #pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t p = pbase + u * 64 + lane;
                    if (p < end) dist[p] = (pcand[u] && pcv[u] == pv[u]) ? (uint16_t)(p - (pcand[u] - 1)) : (uint16_t)0;
                }
            }
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) { pv[u] = v[u]; pcand[u] = cand[u]; pcv[u] = cv[u]; v[u] = vn[u]; cand[u] = candn[u]; }
            pbase = base; havePrev = true;
        }
        if (havePrev) {
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = pbase + u * 64 + lane;
                if (p < end) dist[p] = (pcand[u] && pcv[u] == pv[u]) ? (uint16_t)(p - (pcand[u] - 1)) : (uint16_t)0;
            }
        }
    }
    // positions without 4 bytes left: no candidate
    {
        const uint32_t rend = min(start + ZS_RANGE_SIZE, n);
        for (uint32_t p = max(end, start) + lane; p < rend; p += 64) dist[p] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_lz_walk : one wavefront per block; the 64 lanes are 8 independent walkers of 8 lanes, walker g
// walks range g.  A walker step: its 8 lanes read dist[ip .. ip+64) (8 positions each), the first
// LOOK positions holding a candidate go one per lane; a lane compares 64 bytes forward and 32 bytes
// backward (into the pending literals) for its candidate with one round of loads (ZS_FCAP / ZS_BCAP bytes), scores it, the
// best one of the walker becomes a sequence (extended by the walker's 8 lanes if it hit the 64-byte
// cap).  Eight dependent chains per wavefront hide each other's memory latency.
// Scalar statement: walkRange in oracle/zso_encoder.c.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load64_fwd(const uint8_t *s, uint32_t pos, uint32_t n)
{
    // bytes s[pos .. pos+8), positions >= n read as 0   (n >= 8)
    const uint32_t a = min(pos, n - 8u);
    const uint32_t sh = pos - a;
    const uint64_t w = zs_load64(s + a);
    return sh >= 8u ? 0ull : (w >> (8u * sh));
}
__device__ __forceinline__ uint64_t load64_bwd(const uint8_t *s, int32_t pos)
{
    // bytes s[pos .. pos+8), positions < 0 read as 0   (pos + 8 > 0 not required)
    const int32_t a = max(pos, 0);
    const uint32_t sh = (uint32_t)(a - pos);
    const uint64_t w = zs_load64(s + a);
    return sh >= 8u ? 0ull : (w << (8u * sh));
}

#define ZS_WALK_WAVES 4
extern "C" __global__ void __launch_bounds__(64 * ZS_WALK_WAVES)
k_lz_walk(const uint8_t *__restrict__ src, const ZsBlockDesc *__restrict__ blocks,
          const uint16_t *__restrict__ distAll, ZsSeqRec *__restrict__ seqAll, ZsRangeHdr *__restrict__ hdrAll,
          int look, uint32_t nBlocks)
{
    __shared__ __attribute__((aligned(16))) uint16_t winDist[ZS_WALK_WAVES][8][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t blk = blockIdx.x * ZS_WALK_WAVES + wave;
    if (blk >= nBlocks) return;                                   // no workgroup barrier is used below
    const ZsBlockDesc bd = blocks[blk];
    const uint8_t *s = src + bd.srcOff;
    const uint32_t n = bd.size;
    const uint16_t *dist = distAll + (size_t)blk * ZS_BLOCK_MAX;
    const uint32_t grp = lane >> 3, sub = lane & 7u;
    ZsSeqRec *seqs = seqAll + ((size_t)blk * ZS_MAX_RANGES + grp) * ZS_SEQ_PER_RANGE;
    uint16_t *win = winDist[wave][grp];
    const uint32_t start = grp << ZS_RANGE_LOG;
    const bool alive = (start < n) && (n >= 16);
    const uint32_t end = min(start + ZS_RANGE_SIZE, n);
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;
    const uint32_t scanEnd = alive ? min(end, hashable) : 0;

    uint32_t ip = start, anchor = start, nseq = 0, litSum = 0;
    for (;;) {
        const bool run = ip < scanEnd;
        if (!__any(run)) break;
        const uint32_t wend = min(ip + ZS_WINDOW, scanEnd);
        // ---- window: 8 positions per lane ----
        uint64_t w0 = 0, w1 = 0;
        if (run) { const uint8_t *dp = reinterpret_cast<const uint8_t *>(dist + ip + 8 * sub); w0 = zs_load64(dp); w1 = zs_load64(dp + 8); }
        uint32_t mask8 = 0;
        #pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t dk = (uint32_t)((k < 4 ? (w0 >> (16 * k)) : (w1 >> (16 * (k - 4)))) & 0xFFFFu);
            if (dk && (ip + 8 * sub + k) < wend) mask8 |= 1u << k;
        }
        *reinterpret_cast<uint64_t *>(win + 8 * sub) = w0;
        *reinterpret_cast<uint64_t *>(win + 8 * sub + 4) = w1;
        uint64_t m64 = run ? ((uint64_t)mask8 << (8 * sub)) : 0ull;
        m64 |= (uint64_t)__shfl_xor((long long)m64, 1);
        m64 |= (uint64_t)__shfl_xor((long long)m64, 2);
        m64 |= (uint64_t)__shfl_xor((long long)m64, 4);
        const uint32_t ncand = min((uint32_t)__popcll(m64), (uint32_t)look);
        const bool active = run && sub < ncand;
        uint32_t idx = 0;
        { uint64_t mm = m64; for (uint32_t t = 0; t < sub; t++) mm &= mm - 1; idx = mm ? (uint32_t)__builtin_ctzll(mm) : 0u; }
        const uint32_t off = active ? (uint32_t)win[idx] : 0u;
        const uint32_t q = ip + idx;
        // ---- one round of loads: 64 bytes forward, 32 bytes backward, both sides ----
        uint32_t fwd = 0, back = 0;
        int key = 0;
        if (active) {
            // 16 bytes forward (the score only counts ZS_FCAP of them; the rest saves most extension rounds),
            // 8 bytes backward, both sides: 6 loads
            const uint64_t fa0 = load64_fwd(s, q, n), fa1 = load64_fwd(s, q + 8, n);
            const uint64_t fb0 = load64_fwd(s, q - off, n), fb1 = load64_fwd(s, q - off + 8, n);
            const uint64_t ba = load64_bwd(s, (int32_t)q - 8), bb = load64_bwd(s, (int32_t)(q - off) - 8);
            const uint32_t cap = min(end - q, ZS_LCAP);
            {
                const uint64_t x0 = fa0 ^ fb0, x1 = fa1 ^ fb1;
                const uint32_t n0 = x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : 8u;
                const uint32_t n1 = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8u;
                fwd = min((n0 < 8u) ? n0 : 8u + n1, cap);
            }
            {
                const uint32_t maxBack = min(min(q - anchor, q - off), ZS_BCAP);
                const uint64_t x = ba ^ bb;
                back = min(x ? ((uint32_t)__builtin_clzll(x) >> 3) : 8u, maxBack);
            }
            if (fwd >= ZS_MINMATCH) {
                const int gain = (int)(min(fwd, ZS_FCAP) + back) * 4 - (int)zs_highbit(off + 1) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
                key = ((gain + 2048) << 3) | (int)(7u - sub);
            }
        }
        int best = key;
        best = max(best, __shfl_xor(best, 1));
        best = max(best, __shfl_xor(best, 2));
        best = max(best, __shfl_xor(best, 4));
        const uint32_t bl = (lane & ~7u) + (7u - (uint32_t)(best & 7));      // lane holding the best candidate
        const uint32_t bq = (uint32_t)__shfl((int)q, (int)bl);
        const uint32_t boff = (uint32_t)__shfl((int)off, (int)bl);
        uint32_t bfwd = (uint32_t)__shfl((int)fwd, (int)bl);
        const uint32_t bback = (uint32_t)__shfl((int)back, (int)bl);
        const bool took = run && best != 0;
        // ---- long match: the walker's 8 lanes extend it, 128 bytes per round ----
        bool need = took && bfwd == ZS_LCAP && (end - bq) > ZS_LCAP;
        const bool extended = need;
        uint32_t pos = bq + ZS_LCAP;
        while (__any(need)) {
            uint32_t nb = 0;
            if (need) {
                const uint32_t cap = end - pos;            // pos < end while need
                #pragma unroll
                for (uint32_t h = 0; h < 2; h++) {
                    const uint32_t fo = 16 * sub + 8 * h;
                    uint32_t m = 0;
                    if (fo < cap) {
                        const uint64_t x = load64_fwd(s, pos + fo, n) ^ load64_fwd(s, pos - boff + fo, n);
                        m = x ? ((uint32_t)__builtin_ctzll(x) >> 3) : 8u;
                        m = min(m, cap - fo);
                    }
                    if (h == 0) nb = m; else if (nb == 8u) nb += m;
                }
            }
            const uint64_t stopm = __ballot(nb < 16u);
            const uint32_t g8 = (uint32_t)((stopm >> (8 * grp)) & 0xFFu);
            const uint32_t f = g8 ? (uint32_t)__builtin_ctz(g8) : 0u;
            const uint32_t part = (uint32_t)__shfl((int)nb, (int)((lane & ~7u) + f));
            if (need) {
                if (g8) { pos += 16 * f + part; need = false; }
                else { pos += 128; if (pos >= end) { pos = end; need = false; } }
            }
        }
        if (extended) bfwd = pos - bq;
        if (took) {
            if (sub == 0) {
                ZsSeqRec r;
                r.ll = (uint16_t)(bq - bback - anchor); r.ml = (uint16_t)(bback + bfwd); r.off = (uint16_t)boff; r.flags = (uint16_t)(bq - bback);   // flags: position of the match start
                seqs[nseq] = r;
            }
            nseq++; litSum += bq - bback - anchor;
            ip = bq + bfwd; anchor = ip;
        } else if (run) ip = wend;
    }
    if (sub == 0) { ZsRangeHdr h; h.nseq = nseq; h.trailing = alive ? end - anchor : ((start < n) ? end - start : 0u); h.litSum = litSum; h.pad = 0; hdrAll[(size_t)blk * ZS_MAX_RANGES + grp] = h; }
}
