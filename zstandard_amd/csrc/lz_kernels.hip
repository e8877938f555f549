// LZ stage of the block encoder: candidate search (k_lz_candidates) and greedy walk with
// look-ahead (k_lz_walk).  The scalar statement of the same algorithm is oracle/zso_encoder.c
// (findCandidates / walkRange); the two must agree bit for bit.
//
// There is no reference code for this stage (the reference has no encoder);
// what it emits is consumed by entropy_kernels.hip, whose output the reference decoder must accept.
#include "zsmi_device.h"

// ---------------------------------------------------------------------------------------------
// k_lz_candidates<NR, WPR> : one workgroup per LZ unit (<= NR ranges of 8 KiB), NR * WPR wavefronts.
//   <8, *>  units of <= 64 KiB (one block),  LDS  64 KiB at 2^12 slots
//   <16, 1> units of <= 128 KiB (two blocks), LDS 128 KiB
// LDS: NR hash tables of 2^hashLog 16-bit slots: tag (3 hash bits) << 13 | position in the range; 0xFFFF = empty.
// Positions are taken 64 at a time: all lanes read the table, then all lanes write it (same-slot writes of one
// instruction: the highest lane stays -- probed on MI355X by tools/probe/lds_order.hip).
// Phase A (one wavefront per range: table order matters) fills the tables and leaves each position's distance to its
// own-range predecessor (same slot, same tag) in dist[]; phase B (after a barrier: earlier ranges' tables are final;
// WPR wavefronts per range, alternating trips) falls back to the nearest earlier range holding the slot with the
// same tag, checks the 4 bytes and writes the match distance: low 16 bits to dist[], bit 16 to distHi (NR == 16 only).
// Scalar statement: findCandidates in oracle/zso_encoder.c.
// HBM/L2 traffic per unit: reads n (twice, second time from cache) + n gathers; writes 2n (dist twice).
// ---------------------------------------------------------------------------------------------
#define ZS_TAG_BITS 3
#ifndef ZS_CAND_WPR
#define ZS_CAND_WPR 2              // wavefronts per range of the small-unit candidates kernel (8 ranges): 1 -> 512 threads, 2 -> 1024
#endif
#define ZS_SLOT_EMPTY 0xFFFFu
__device__ __forceinline__ uint32_t zs_slot_entry(uint32_t hh, int hashLog, uint32_t p)
{ return (((hh >> (32 - hashLog - ZS_TAG_BITS)) & ((1u << ZS_TAG_BITS) - 1)) << ZS_RANGE_LOG) | (p & (ZS_RANGE_SIZE - 1)); }

// measured on MI355X (4096 x 64 KiB, ms per launch): WPR 1: U 8 1.13, U 4 0.96, U 2 1.39; WPR 2 with 2 workgroups per CU
// (<= 64 VGPRs): U 8 1.02, U 4 0.91, U 3 0.90, U 2 1.12, U 1 1.51
#ifndef ZS_CAND_U
#define ZS_CAND_U 4                // steps of 64 positions per trip (loads in flight per lane)
#endif
#ifndef ZS_CAND_MINWG
#define ZS_CAND_MINWG 2            // small-unit kernel: 64 KiB of LDS, so two workgroups share a CU if the registers allow
#endif
template <int NR, int WPR>
__global__ void __launch_bounds__(NR * WPR * 64, (NR == 8 ? ZS_CAND_MINWG : 1))
k_lz_candidates(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
                uint16_t *__restrict__ distAll, uint8_t *__restrict__ distHiAll, int hashLog)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t tables[];
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    uint16_t *dist = distAll + (size_t)(ud.firstBlock - block0) * ZS_BLOCK_MAX;
    uint8_t *distHi = distHiAll + (size_t)(ud.firstBlock - block0) * (ZS_BLOCK_MAX / 8);

    {   // clear the tables
        const uint32_t words = ((uint32_t)NR << hashLog) >> 1;      // 32-bit words
        uint32_t *t32 = reinterpret_cast<uint32_t *>(tables);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) t32[i] = 0xFFFFFFFFu;
    }
    __syncthreads();

    const uint32_t waveAll = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave = waveAll % NR, half = waveAll / NR;          // range, and which of the range's WPR wavefronts
    const uint32_t start = wave << ZS_RANGE_LOG;
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;                   // positions [0, hashable) have 4 bytes
    const uint32_t end = min(start + ZS_RANGE_SIZE, hashable);
    uint16_t *T = tables + ((size_t)wave << hashLog);

    // U steps per trip.  The loads of trip t+1 are issued before trip t is worked on (registers double-buffered),
    // so the table walk of a trip runs under the memory latency of the next one.
    constexpr uint32_t U = ZS_CAND_U;
    if (half == 0) {
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
                    const uint32_t hh = v[u] * 2654435761u;
                    const uint32_t h = hh >> (32 - hashLog);
                    const uint32_t mine = zs_slot_entry(hh, hashLog, p);
                    const uint32_t own = T[h];
                    T[h] = (uint16_t)mine;
                    // same tag, filled: the predecessor is earlier in this range, so the distance is 1..8191
                    dist[p] = (own != ZS_SLOT_EMPTY && ((own ^ mine) >> ZS_RANGE_LOG) == 0) ? (uint16_t)(mine - own) : (uint16_t)0;
                }
            }
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) v[u] = vn[u];
        }
    }
    __syncthreads();

    {
        // trip t: (v, own candidate) loaded one trip ahead; its verification gathers are issued, then trip t-1's
        // gathers (issued one trip earlier) are compared and stored.  cand = candidate position + 1, 0 = none.
        uint32_t v[U], cand[U], vn[U], candn[U], pv[U], pcand[U], pcv[U];
        uint32_t pbase = 0; bool havePrev = false;
        const uint32_t first = start + half * 64 * U;
        auto finish = [&](uint32_t fbase) {
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = fbase + u * 64 + lane;
                uint32_t d = (p < end && pcand[u] && pcv[u] == pv[u]) ? p - (pcand[u] - 1) : 0u;
                if (NR > 8) {
                    if (d == 65536u) d = 0;
                    const uint64_t hi = __ballot((d >> 16) != 0);
                    if (lane == 0 && fbase + u * 64 < end) *reinterpret_cast<uint64_t *>(distHi + ((fbase + u * 64) >> 3)) = hi;
                }
                if (p < end) dist[p] = (uint16_t)d;
            }
        };
        #pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t p = first + u * 64 + lane; const bool in = p < end;
            v[u] = in ? zs_load32(s + p) : 0u; const uint32_t d = in ? (uint32_t)dist[p] : 0u; cand[u] = d ? p - d + 1 : 0u;
        }
        for (uint32_t base = first; base < end; base += WPR * 64 * U) {
            const uint32_t nbase = base + WPR * 64 * U;
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = nbase + u * 64 + lane; const bool in = p < end;
                vn[u] = in ? zs_load32(s + p) : 0u; const uint32_t d = in ? (uint32_t)dist[p] : 0u; candn[u] = d ? p - d + 1 : 0u;
            }
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = base + u * 64 + lane;
                if (p < end && !cand[u]) {
                    // all earlier ranges are read at once (independent LDS reads); the nearest one holding the slot with this tag wins
                    const uint32_t hh = v[u] * 2654435761u;
                    const uint32_t h = hh >> (32 - hashLog);
                    const uint32_t tagv = zs_slot_entry(hh, hashLog, 0);
                    uint32_t c[NR - 1];
                    #pragma unroll
                    for (uint32_t q = 0; q < NR - 1; q++) c[q] = (q < wave) ? (uint32_t)tables[((size_t)q << hashLog) + h] : ZS_SLOT_EMPTY;
                    #pragma unroll
                    for (uint32_t q = 0; q < NR - 1; q++)
                        if (c[q] != ZS_SLOT_EMPTY && ((c[q] ^ tagv) >> ZS_RANGE_LOG) == 0) cand[u] = (q << ZS_RANGE_LOG) + (c[q] & (ZS_RANGE_SIZE - 1)) + 1;
                }
            }
            uint32_t cv[U];
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) cv[u] = cand[u] ? zs_load32(s + cand[u] - 1) : 0u;
            if (havePrev) finish(pbase);
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) { pv[u] = v[u]; pcand[u] = cand[u]; pcv[u] = cv[u]; v[u] = vn[u]; cand[u] = candn[u]; }
            pbase = base; havePrev = true;
        }
        if (havePrev) finish(pbase);
    }
    // positions without 4 bytes left: no candidate
    {
        const uint32_t rend = min(start + ZS_RANGE_SIZE, n);
        if (half == 0) for (uint32_t p = max(end, start) + lane; p < rend; p += 64) dist[p] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_lz_walk<NW> : one workgroup of NW / 8 wavefronts per LZ unit; NW = 64 (unit <= 64 KiB) or 128 (<= 128 KiB).
// The unit's source bytes are staged in LDS once (NW KiB + pads), so every compare of the walk is an LDS read and
// the unit is fetched from HBM once.  The lanes are NW independent walkers of 8 lanes; walker g walks the walk
// range g (1 KiB).  A walker step: its 8 lanes read dist[ip .. ip+64) (8 positions each, global, coalesced); the
// first LOOK positions holding a candidate go one per lane; a lane compares 16 bytes forward (the score counts
// ZS_FCAP of them) and 8 bytes backward (into the pending literals) for its candidate, scores it; the
// best one of the walker becomes a sequence (extended by the walker's 8 lanes if it hit the 16-byte cap).
// Scalar statement: walkRange in oracle/zso_encoder.c.
// ---------------------------------------------------------------------------------------------
#define ZS_WALK_FRONT 16u          // LDS bytes in front of the unit (backward reads near position 0)
#define ZS_WALK_TAIL  144u         // zero bytes behind the unit (forward reads near the end)
#define ZS_WALK_LDS(NW) (ZS_WALK_FRONT + (NW) * ZS_WALK_SIZE + ZS_WALK_TAIL + (NW) * 64u * 2u)

__device__ __forceinline__ uint64_t lds64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

template <int NW>
__global__ void __launch_bounds__(NW * 8)
k_lz_walk(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
          const uint16_t *__restrict__ distAll, const uint8_t *__restrict__ distHiAll,
          ZsSeqRec *__restrict__ seqAll, ZsRangeHdr *__restrict__ hdrAll, int look)
{
    constexpr uint32_t CAP = NW * ZS_WALK_SIZE;                                   // unit capacity in bytes
    constexpr bool BIG = NW > 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t walkLds[];
    uint8_t *ls = walkLds + ZS_WALK_FRONT;                                       // ls[p] = source byte p
    uint16_t *winAll = reinterpret_cast<uint16_t *>(walkLds + ZS_WALK_FRONT + CAP + ZS_WALK_TAIL);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint32_t slot = ud.firstBlock - block0;                                // scratch slot of the unit's first block
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    const uint16_t *dist = distAll + (size_t)slot * ZS_BLOCK_MAX;
    const uint8_t *distHi = distHiAll + (size_t)slot * (ZS_BLOCK_MAX / 8);
    const uint32_t grp = lane >> 3, sub = lane & 7u;
    const uint32_t walker = wave * 8 + grp;
    ZsSeqRec *seqs = seqAll + ((size_t)slot * ZS_WALK_RANGES + walker) * ZS_SEQ_PER_RANGE;
    uint16_t *win = winAll + walker * 64;

    // ---- stage the unit ----
    if (tid < ZS_WALK_FRONT / 4) reinterpret_cast<uint32_t *>(walkLds)[tid] = 0;
    for (uint32_t i = tid * 16; i < n + ZS_WALK_TAIL; i += NW * 8 * 16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i + 16 <= n) __builtin_memcpy(&v, s + i, 16);
        else if (i < n) {
            uint64_t lo = 0, hi = 0;
            for (uint32_t k = 0; k < 16 && i + k < n; k++) { const uint64_t c = s[i + k]; if (k < 8) lo |= c << (8 * k); else hi |= c << (8 * (k - 8)); }
            v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        }
        if (i + 16 <= CAP + ZS_WALK_TAIL) *reinterpret_cast<uint4 *>(ls + i) = v;
    }
    __syncthreads();

    const uint32_t start = walker << ZS_WALK_LOG;
    const uint32_t blockStart = start & ~(ZS_BLOCK_MAX - 1);                     // the walker's block inside the unit
    const uint32_t blockN = (blockStart < n) ? min(n - blockStart, ZS_BLOCK_MAX) : 0u;
    const bool alive = (start < n) && (blockN >= 16);
    const uint32_t end = min(start + ZS_WALK_SIZE, n);
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;
    const uint32_t scanEnd = alive ? min(end, hashable) : 0;

    uint32_t ip = start, anchor = start, nseq = 0, litSum = 0;
    for (;;) {
        const bool run = ip < scanEnd;
        if (!__any(run)) break;
        const uint32_t wend = min(ip + ZS_WINDOW, scanEnd);
        // ---- window: 8 positions per lane ----
        uint64_t w0 = 0, w1 = 0; uint32_t hi8 = 0;
        if (run) {
            const uint8_t *dp = reinterpret_cast<const uint8_t *>(dist + ip + 8 * sub); w0 = zs_load64(dp); w1 = zs_load64(dp + 8);
            if (BIG) { const uint8_t *hp = distHi + ((ip + 8 * sub) >> 3); hi8 = (((uint32_t)hp[0] | ((uint32_t)hp[1] << 8)) >> (ip & 7u)) & 0xFFu; }
        }
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
        uint32_t off = active ? (uint32_t)win[idx] : 0u;
        if (BIG) { const uint32_t hsrc = (uint32_t)__shfl((int)hi8, (int)((lane & ~7u) + (idx >> 3))); if (active) off |= ((hsrc >> (idx & 7u)) & 1u) << 16; }
        const uint32_t q = ip + idx;
        // ---- compare from LDS: 16 bytes forward, 8 bytes backward, both sides ----
        uint32_t fwd = 0, back = 0;
        int key = 0;
        if (active) {
            const uint8_t *pa = ls + q, *pb = ls + q - off;
            const uint64_t x0 = lds64(pa) ^ lds64(pb), x1 = lds64(pa + 8) ^ lds64(pb + 8);
            const uint64_t xb = lds64(pa - 8) ^ lds64(pb - 8);
            const uint32_t cap = min(end - q, ZS_LCAP);
            const uint32_t n0 = x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : 8u;
            const uint32_t n1 = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8u;
            fwd = min((n0 < 8u) ? n0 : 8u + n1, cap);
            const uint32_t maxBack = min(min(q - anchor, q - off), ZS_BCAP);
            back = min(xb ? ((uint32_t)__builtin_clzll(xb) >> 3) : 8u, maxBack);
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
        // ---- long match: the walker's 8 lanes extend it, 128 bytes per round (LDS) ----
        bool need = took && bfwd == ZS_LCAP && (end - bq) > ZS_LCAP;
        const bool extended = need;
        uint32_t pos = bq + ZS_LCAP;
        while (__any(need)) {
            uint32_t nb = 0;
            if (need) {
                const uint32_t cap = end - pos;            // pos < end while need
                const uint32_t fo = 16 * sub;
                if (fo < cap) {
                    const uint8_t *pa = ls + pos + fo, *pb = ls + pos - boff + fo;
                    const uint64_t x0 = lds64(pa) ^ lds64(pb), x1 = lds64(pa + 8) ^ lds64(pb + 8);
                    const uint32_t n0 = x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : 8u;
                    const uint32_t n1 = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8u;
                    nb = min((n0 < 8u) ? n0 : 8u + n1, cap - fo);
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
                // ml bit 13: bit 16 of the offset; flags: position of the match start in its block
                r.ll = (uint16_t)(bq - bback - anchor); r.ml = (uint16_t)((bback + bfwd) | ((boff >> 16) << 13)); r.off = (uint16_t)boff; r.flags = (uint16_t)(bq - bback);
                seqs[nseq] = r;
            }
            nseq++; litSum += bq - bback - anchor;
            ip = bq + bfwd; anchor = ip;
        } else if (run) ip = wend;
    }
    if (sub == 0) { ZsRangeHdr h; h.nseq = nseq; h.trailing = alive ? end - anchor : ((start < n) ? end - start : 0u); h.litSum = litSum; h.pad = 0; hdrAll[(size_t)slot * ZS_WALK_RANGES + walker] = h; }
}
