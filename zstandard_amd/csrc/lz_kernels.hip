// LZ stage of the block encoder: candidate search (k_lz_candidates) and greedy walk with
// look-ahead (k_lz_walk).  The scalar statement of the same algorithm is oracle/zso_encoder.c
// (findCandidates / walkRange); the two must agree bit for bit.
//
// There is no reference code for this stage (the reference has no encoder);
// what it emits is consumed by entropy_kernels.hip, whose output the reference decoder must accept.
#include "zsmi_device.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// k_lz_candidates<NR, WPR> : one workgroup per LZ unit (<= NR ranges of 8 KiB), NR * WPR wavefronts.
//   <8, 2>  units of <= 64 KiB (one block),  LDS  64 KiB at 2^12 slots, two workgroups per CU
//   <16, 1> units of <= 128 KiB (two blocks), LDS 128 KiB
// LDS: NR hash tables of 2^hashLog 16-bit slots: tag (3 hash bits) << 13 | position in the range; 0xFFFF = empty.
// Positions are taken 64 at a time: all lanes read the table, then all lanes write it (same-slot writes of one
// instruction: the highest lane stays -- probed on MI355X by tools/probe/lds_order.hip).
// Phase A (one wavefront per range: table order matters) fills the tables and leaves each position's distance to its
// own-range predecessor (same slot, same tag) in dist[].  Phase B (after a barrier: every table is final) takes the
// positions without one, falls back to the nearest earlier range holding the slot with the same tag, checks the 4 bytes
// of every candidate and writes the match distance: low 16 bits to dist[], bit 16 to distHi (NR == 16 only), and the
// "has a candidate" bit plane.  Phase B work is dealt in trips of 64 * U positions from a queue in LDS, last trip first:
// a trip of range r probes r tables, so fixed shares would leave the wavefronts of the low ranges waiting at the end.
// Trips that lie wholly inside the hashable positions run without bound checks; the one trip that may not (the last
// of the unit) has its own guarded code.
// Scalar statement: findCandidates in oracle/zso_encoder.c.
// HBM/L2 traffic per unit: reads n (twice, second time from cache) + n gathers; writes 2n (dist twice).
// What bounds it on MI355X (profiles/r1_h_*): the vector-memory path.  TA_TA_BUSY 95 % of the kernel's cycles, three
// quarters of it the verification gathers (64 lanes = 64 cache lines an instruction); vector ALU issue 82 %.
// ---------------------------------------------------------------------------------------------
#define ZS_TAG_BITS 3
#ifndef ZS_CAND_WPR
#define ZS_CAND_WPR 2              // wavefronts per range of the small-unit candidates kernel (8 ranges): 1 -> 512 threads, 2 -> 1024
#endif
#define ZS_SLOT_EMPTY 0xFFFFu
#define ZS_CAND_LDS(NR) (((size_t)(NR) << ZS_HASH_LOG) * 2 + 16)      // the tables + the trip queue
__device__ __forceinline__ uint32_t zs_slot_entry(uint32_t hh, int hashLog, uint32_t p)
{ return (((hh >> (32 - hashLog - ZS_TAG_BITS)) & ((1u << ZS_TAG_BITS) - 1)) << ZS_RANGE_LOG) | (p & (ZS_RANGE_SIZE - 1)); }

// measured on MI355X (4096 x 64 KiB, ms per launch): WPR 1: U 8 1.13, U 4 0.96, U 2 1.39; WPR 2 with 2 workgroups per CU
// (<= 64 VGPRs): U 8 1.02, U 4 0.91, U 3 0.90, U 2 1.12, U 1 1.51
#ifndef ZS_CAND_U
#define ZS_CAND_U 4                // steps of 64 positions per trip (loads in flight per lane)
#endif
#ifndef ZS_CAND_U_BIG
#define ZS_CAND_U_BIG 8
#endif
#ifndef ZS_CAND_MINWG
#define ZS_CAND_MINWG 2            // small-unit kernel: 64 KiB of LDS, so two workgroups share a CU if the registers allow
#endif
template <int NR, int WPR>
__global__ void __launch_bounds__(NR * WPR * 64, (NR == 8 ? ZS_CAND_MINWG : 1))
k_lz_candidates(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
                uint16_t *__restrict__ distAll, uint8_t *__restrict__ distHiAll, uint8_t *__restrict__ distMaskAll, int hashLogArg)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t tables[];
    constexpr int hashLog = ZS_HASH_LOG;                              // fixed: table strides become instruction immediates
    if (hashLogArg != hashLog) return;                                // the host passes ZS_HASH_LOG (zsmi_api.hip)
    uint32_t *queue = reinterpret_cast<uint32_t *>(tables + ((size_t)NR << hashLog));      // next phase-B trip
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    uint16_t *dist = distAll + (size_t)(ud.firstBlock - block0) * ZS_BLOCK_MAX;
    uint8_t *distHi = distHiAll + (size_t)(ud.firstBlock - block0) * (ZS_BLOCK_MAX / 8);
    uint8_t *distMask = distMaskAll + (size_t)(ud.firstBlock - block0) * (ZS_BLOCK_MAX / 8);     // bit p: position p has a candidate

#ifdef ZS_K1_PROFILE          // development aid (tools/k1_profile.py): s_memtime at the phase boundaries, left in the unit's distHi plane
    uint64_t profT[4]; profT[0] = __builtin_amdgcn_s_memtime();
#endif
    {   // clear the tables
        const uint32_t words = ((uint32_t)NR << hashLog) >> 1;      // 32-bit words
        uint32_t *t32 = reinterpret_cast<uint32_t *>(tables);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) t32[i] = 0xFFFFFFFFu;
        if (threadIdx.x == 0) *queue = 0;
    }
    __syncthreads();

    // wavefront-uniform values are made scalars (readfirstlane): range bounds, trip bases and the loop control live in SGPRs
    const uint32_t waveAll = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;                   // positions [0, hashable) have 4 bytes
    // U steps of 64 positions per trip.  The loads of trip t+1 are issued before trip t is worked on (registers
    // double-buffered), so the table work of a trip runs under the memory latency of the next one.
    constexpr uint32_t U = (NR > 8) ? ZS_CAND_U_BIG : ZS_CAND_U;      // the 128 KiB shape runs one workgroup per CU: more loads in flight per wavefront
    constexpr uint32_t TRIP = 64 * U;
    using Whole = std::true_type; using Guarded = std::false_type;

#ifdef ZS_K1_PROFILE
    profT[1] = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase A: wavefront r fills table r in position order
    if (waveAll < NR) {
        const uint32_t start = waveAll << ZS_RANGE_LOG, end = min(start + ZS_RANGE_SIZE, hashable);
        uint16_t *T = tables + ((size_t)waveAll << hashLog);
        auto load = [&](auto tag, uint32_t base, uint32_t (&v)[U]) {
            constexpr bool WHOLE = decltype(tag)::value;
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t p = base + u * 64 + lane; v[u] = (WHOLE || p < end) ? zs_load32(s + p) : 0u; }
        };
        auto insert = [&](auto tag, uint32_t base, const uint32_t (&v)[U]) {
            constexpr bool WHOLE = decltype(tag)::value;
            // all U reads and writes of the trip go to the LDS back to back (it keeps their order); the distances are worked
            // out afterwards, so the trip pays the LDS latency once
            uint32_t own[U], mine[U];
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = base + u * 64 + lane;
                if (WHOLE || p < end) {
                    const uint32_t hh = v[u] * 2654435761u;
                    const uint32_t h = __builtin_amdgcn_ubfe(hh, 32 - hashLog, hashLog);
                    mine[u] = zs_slot_entry(hh, hashLog, p);
                    own[u] = T[h];
                    T[h] = (uint16_t)mine[u];
                }
            }
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = base + u * 64 + lane;
                if (WHOLE || p < end) {
                    // same tag: the distance to the slot's position; an empty slot reads as (tag 7, position 8191), which no
                    // position of the range lies behind: its "distance" is <= 0 and becomes 0 = no predecessor
                    const int d = (int)mine[u] - (int)own[u];
                    dist[p] = (uint16_t)(((own[u] ^ mine[u]) < ZS_RANGE_SIZE) ? max(d, 0) : 0);
                }
            }
        };
        if (start < end) {
            uint32_t v[U], vn[U];
            if (start + TRIP <= end) load(Whole{}, start, v); else load(Guarded{}, start, v);
            // the first trip's values are waited for here, in front of the loop: left to the loop body, the wait lands behind the
            // next trip's loads and takes them along (every trip would then sit out a full memory round trip)
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) asm volatile("" : "+v"(v[u]));
            uint32_t base = start;
            for (; base + TRIP <= end; base += TRIP) {
                const uint32_t nbase = base + TRIP;
                if (nbase + TRIP <= end) load(Whole{}, nbase, vn); else if (nbase < end) load(Guarded{}, nbase, vn);
                insert(Whole{}, base, v);
                #pragma unroll
                for (uint32_t u = 0; u < U; u++) v[u] = vn[u];
            }
            if (base < end) insert(Guarded{}, base, v);
        }
    }
#ifdef ZS_K1_PROFILE
    profT[2] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef ZS_K1_PROFILE
    profT[3] = __builtin_amdgcn_s_memtime();
#endif

    // ---- phase B
    {
        // (v, own-range distance) of a trip are loaded one trip ahead; the trip's verification gathers are issued, then the
        // previous trip's gathers are compared and stored.  d = distance to the candidate, 0 = none.
        auto load = [&](auto tag, uint32_t base, uint32_t (&v)[U], uint32_t (&d)[U]) {
            constexpr bool WHOLE = decltype(tag)::value;
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = base + u * 64 + lane; const bool in = WHOLE || p < hashable;
                v[u] = in ? zs_load32(s + p) : 0u; d[u] = in ? (uint32_t)dist[p] : 0u;
            }
        };
        auto probe = [&](auto tag, uint32_t base, const uint32_t (&v)[U], uint32_t (&d)[U]) {
            constexpr bool WHOLE = decltype(tag)::value;
            uint32_t rangeV;                                          // the trip's range, as a per-lane value: the table reads below
            asm volatile("v_mov_b32 %0, %1" : "=v"(rangeV) : "s"(base >> ZS_RANGE_LOG));      // are masked, not branched around
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = base + u * 64 + lane;
                if ((WHOLE || p < hashable) && !d[u]) {
                    // the earlier ranges are read at once (independent LDS reads, range q at byte offset q << (hashLog + 1): an
                    // immediate of the instruction); the nearest one holding the slot with this tag wins
                    const uint32_t hh = v[u] * 2654435761u;
                    const uint32_t h = __builtin_amdgcn_ubfe(hh, 32 - hashLog, hashLog);
                    const uint32_t tagv = zs_slot_entry(hh, hashLog, 0);
                    // slot ^ tagv < 8192 <=> same tag, and then it is the position in the range.  The empty slot 0xFFFF would pass
                    // as (tag 7, position 8191): with tag 7 the limit drops to 8191.  Ascending ranges, each hit replacing the last;
                    // "none" is the position itself (distance 0).  (Ending the chain at the trip's own range count through scalar
                    // branches halves its instructions and buys nothing: the kernel is bound by the vector-memory path, TA busy 95 %.)
                    const uint32_t limit = ZS_RANGE_SIZE - (tagv == (7u << ZS_RANGE_LOG) ? 1u : 0u);
                    const uint16_t *Th = tables + h;
                    uint32_t c[NR - 1];
                    #pragma unroll
                    for (uint32_t q = 0; q < NR - 1; q++) c[q] = (q < rangeV) ? (uint32_t)Th[(size_t)q << hashLog] : ZS_SLOT_EMPTY;
                    uint32_t best = p;
                    #pragma unroll
                    for (uint32_t q = 0; q < NR - 1; q++) {
                        const uint32_t x = c[q] ^ tagv;
                        best = (x < limit) ? x + (q << ZS_RANGE_LOG) : best;
                    }
                    d[u] = p - best;
                }
            }
        };
        auto gather = [&](uint32_t base, const uint32_t (&d)[U], uint32_t (&cv)[U]) {
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) cv[u] = d[u] ? zs_load32(s + (base + u * 64 + lane - d[u])) : 0u;
        };
        auto finish = [&](auto tag, uint32_t fbase, const uint32_t (&pv)[U], const uint32_t (&pd)[U], const uint32_t (&pcv)[U]) {
            constexpr bool WHOLE = decltype(tag)::value;
            // candidate bits (and bit 16 of the distances) of the trip's U groups of 64 positions: lane u keeps group u's word,
            // so each plane takes one store of U * 8 contiguous bytes.  Without a candidate the gathered word is 0 and so is d.
            uint64_t pmMine = 0, hiMine = 0;
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t p = fbase + u * 64 + lane;
                uint32_t d = (pcv[u] == pv[u]) ? pd[u] : 0u;
                if (NR > 8) {
                    if (d == 65536u) d = 0;
                    const uint64_t hi = __ballot((d >> 16) != 0);
                    if (lane == u) hiMine = hi;
                }
                const uint64_t pm = __ballot(d != 0);
                if (lane == u) pmMine = pm;
                if (WHOLE || p < hashable) dist[p] = (uint16_t)d;
            }
            if (lane < U && (WHOLE || fbase + lane * 64 < hashable)) {
                *reinterpret_cast<uint64_t *>(distMask + ((fbase + lane * 64) >> 3)) = pmMine;
                if (NR > 8) *reinterpret_cast<uint64_t *>(distHi + ((fbase + lane * 64) >> 3)) = hiMine;
            }
        };
        auto grab = [&]() -> uint32_t {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(queue, 1u);
            return __builtin_amdgcn_readfirstlane(i);
        };

        const uint32_t nWhole = hashable / TRIP;                      // trips [t * TRIP, (t + 1) * TRIP) wholly hashable
        uint32_t v[U], d[U], vn[U], dn[U], pv[U], pd[U], pcv[U], cv[U];
        if (waveAll == NR * WPR - 1 && nWhole * TRIP < hashable) {    // the unit's last, partial trip: on its own, not pipelined
            const uint32_t base = nWhole * TRIP;
            load(Guarded{}, base, v, d);
            probe(Guarded{}, base, v, d);
            gather(base, d, cv);
            finish(Guarded{}, base, v, d, cv);
        }
        uint32_t cur = grab(), pbase = 0; bool havePrev = false;
        if (cur < nWhole) {
            load(Whole{}, (nWhole - 1 - cur) * TRIP, v, d);
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) asm volatile("" : "+v"(v[u]), "+v"(d[u]));      // as in phase A: wait in front of the loop
        }
        while (cur < nWhole) {
            const uint32_t base = (nWhole - 1 - cur) * TRIP;
            const uint32_t nxt = grab();
            if (nxt < nWhole) load(Whole{}, (nWhole - 1 - nxt) * TRIP, vn, dn);
            probe(Whole{}, base, v, d);
            gather(base, d, cv);
            if (havePrev) finish(Whole{}, pbase, pv, pd, pcv);
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) { pv[u] = v[u]; pd[u] = d[u]; pcv[u] = cv[u]; v[u] = vn[u]; d[u] = dn[u]; }
            pbase = base; havePrev = true; cur = nxt;
        }
        if (havePrev) finish(Whole{}, pbase, pv, pd, pcv);
    }
#ifdef ZS_K1_PROFILE
    if (NR == 8 && lane == 0) {           // per wavefront: start, phase A begin / end, phase B begin / end
        uint64_t *o = reinterpret_cast<uint64_t *>(distHi) + waveAll * 8;
        o[0] = profT[0]; o[1] = profT[1]; o[2] = profT[2]; o[3] = profT[3]; o[4] = __builtin_amdgcn_s_memtime();
    }
#endif
    // positions without 4 bytes left: no candidate
    if (waveAll == 0) for (uint32_t p = hashable + lane; p < n; p += 64) dist[p] = 0;
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
#define ZS_WALK_LDS(CAPB) (ZS_WALK_FRONT + (CAPB) + ZS_WALK_TAIL + (CAPB) / 8 + 16)   // source + candidate bit plane

// K dwords of the LDS copy starting at any byte offset, fetched as K + 1 aligned dwords and shifted into place
// (an unaligned ds_read_b64 / b128 costs the LDS several passes: SQ_LDS_UNALIGNED_STALL was 80 % of its busy time)
template <int K>
__device__ __forceinline__ void lds_span(const uint8_t *ldsBase, uint32_t byteOff, uint32_t (&out)[K])
{
    const uint32_t *d = reinterpret_cast<const uint32_t *>(ldsBase + (byteOff & ~3u));
    uint32_t w[K + 1];
    #pragma unroll
    for (int k = 0; k <= K; k++) w[k] = d[k];
    const uint32_t sh = byteOff & 3u;
    #pragma unroll
    for (int k = 0; k < K; k++) out[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
}
__device__ __forceinline__ uint64_t zs_u64(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

template <int NW, int WLOG>
__global__ void __launch_bounds__(NW * 8)
k_lz_walk(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
          const uint16_t *__restrict__ distAll, const uint8_t *__restrict__ distHiAll, const uint8_t *__restrict__ distMaskAll,
          ZsSeqRec *__restrict__ seqAll, ZsRangeHdr *__restrict__ hdrAll, int look)
{
    constexpr uint32_t WSIZE = 1u << WLOG;                                        // bytes per walk range
    constexpr uint32_t CAP = NW * WSIZE;                                          // unit capacity in bytes
    constexpr bool BIG = CAP > ZS_BLOCK_MAX;
    extern __shared__ __attribute__((aligned(16))) uint8_t walkLds[];
    uint8_t *ls = walkLds + ZS_WALK_FRONT;                                       // ls[p] = source byte p
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint32_t slot = ud.firstBlock - block0;                                // scratch slot of the unit's first block
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    const uint16_t *dist = distAll + (size_t)slot * ZS_BLOCK_MAX;
    const uint8_t *distHi = distHiAll + (size_t)slot * (ZS_BLOCK_MAX / 8);
    const uint64_t *distMask = reinterpret_cast<const uint64_t *>(distMaskAll + (size_t)slot * (ZS_BLOCK_MAX / 8));
    uint64_t *lm = reinterpret_cast<uint64_t *>(walkLds + ZS_WALK_FRONT + CAP + ZS_WALK_TAIL);     // lm: bit p set = position p has a candidate
    const uint32_t grp = lane >> 3, sub = lane & 7u;
    const uint32_t walker = wave * 8 + grp;
    ZsSeqRec *seqs = seqAll + (size_t)slot * (ZS_BLOCK_MAX / 4) + (size_t)walker * (WSIZE / 4);

    // ---- stage the unit ----
    if (tid < ZS_WALK_FRONT / 4) reinterpret_cast<uint32_t *>(walkLds)[tid] = 0;
    {
        // whole 16-byte pieces: the (up to 8) loads of a thread are issued together (a load behind a branch, followed by its LDS
        // store, would wait out one memory round trip per piece); then the partial piece and the zero tail
        constexpr uint32_t T = NW * 8, PER = CAP / 16 / T;                       // threads, pieces per thread (8)
        const uint32_t nFull = n & ~15u;
        uint4 v[PER];
        #pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = (tid + k * T) * 16;
            const uint32_t ii = (i + 16 <= nFull) ? i : 0u;                      // clamped: always a valid address when nFull >= 16
            v[k] = make_uint4(0, 0, 0, 0);
            if (nFull >= 16) __builtin_memcpy(&v[k], s + ii, 16);
        }
        #pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = (tid + k * T) * 16;
            if (i + 16 <= nFull) *reinterpret_cast<uint4 *>(ls + i) = v[k];
        }
        for (uint32_t i = nFull + tid * 16; i < n + ZS_WALK_TAIL; i += T * 16) {
            uint4 w = make_uint4(0, 0, 0, 0);
            if (i < n) {
                uint64_t lo = 0, hi = 0;
                for (uint32_t k = 0; k < 16 && i + k < n; k++) { const uint64_t c = s[i + k]; if (k < 8) lo |= c << (8 * k); else hi |= c << (8 * (k - 8)); }
                w = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
            }
            if (i + 16 <= CAP + ZS_WALK_TAIL) *reinterpret_cast<uint4 *>(ls + i) = w;
        }
    }
    // candidate bits of the positions that can start a match; the word behind them reads as zero
    {
        const uint32_t hashableAll = (n >= 4) ? n - 3 : 0, words = (hashableAll + 63) >> 6;
        for (uint32_t i = tid; i <= words; i += NW * 8) lm[i] = (i < words) ? distMask[i] : 0ull;
    }
    __syncthreads();

    const uint32_t start = walker << WLOG;
    const uint32_t blockStart = start & ~(ZS_BLOCK_MAX - 1);                     // the walker's block inside the unit
    const uint32_t blockN = (blockStart < n) ? min(n - blockStart, ZS_BLOCK_MAX) : 0u;
    const bool alive = (start < n) && (blockN >= 16);
    const uint32_t end = min(start + WSIZE, n);
    const uint32_t hashable = (n >= 4) ? n - 3 : 0;
    const uint32_t scanEnd = alive ? min(end, hashable) : 0;

    uint32_t ip = start, anchor = start, nseq = 0, litSum = 0;
    for (;;) {
        const bool run = ip < scanEnd;
        if (!__any(run)) break;
        const uint32_t wend = min(ip + ZS_WINDOW, scanEnd);
        // ---- window: candidate bits of [ip, ip + 64) from LDS (the same for the walker's 8 lanes); lane sub takes the
        //      sub-th candidate and fetches its distance ----
        uint64_t m64 = 0;
        if (run) {
            const uint32_t wi = ip >> 6, sh = ip & 63u, wlen = wend - ip;
            const uint64_t lo = lm[wi], hi = lm[wi + 1];
            m64 = (lo >> sh) | ((hi << 1) << (63u - sh));
            if (wlen < 64u) m64 &= (1ull << wlen) - 1ull;
        }
        const uint32_t mlo = (uint32_t)m64, mhi = (uint32_t)(m64 >> 32);
        const uint32_t clo = (uint32_t)__popc(mlo);
        const uint32_t ncand = min(clo + (uint32_t)__popc(mhi), (uint32_t)look);
        const bool active = run && sub < ncand;
        uint32_t idx = 0;
        {   // position of the sub-th set bit
            const bool upper = sub >= clo;
            uint32_t mm = upper ? mhi : mlo; const uint32_t skip = upper ? sub - clo : sub;     // skip <= 7
            // clear the lowest skip set bits, branch free: 4, 2, 1 of them by the bits of skip
            { uint32_t t = mm & (mm - 1); t &= t - 1; t &= t - 1; t &= t - 1; mm = (skip & 4u) ? t : mm; }
            { uint32_t t = mm & (mm - 1); t &= t - 1; mm = (skip & 2u) ? t : mm; }
            { const uint32_t t = mm & (mm - 1); mm = (skip & 1u) ? t : mm; }
            idx = mm ? (uint32_t)__builtin_ctz(mm) + (upper ? 32u : 0u) : 0u;
        }
        uint32_t off = active ? (uint32_t)dist[ip + idx] : 0u;
        if (BIG) { if (active) off |= (((uint32_t)distHi[(ip + idx) >> 3] >> ((ip + idx) & 7u)) & 1u) << 16; }
        const uint32_t q = ip + idx;
        // ---- compare from LDS: 16 bytes forward, 8 bytes backward, both sides ----
        uint32_t fwd = 0, back = 0;
        int key = 0;
        if (active) {
            uint32_t a[6], b[6];                                                 // bytes [q - 8, q + 16) of both sides
            lds_span<6>(walkLds, ZS_WALK_FRONT + q - 8, a);
            lds_span<6>(walkLds, ZS_WALK_FRONT + q - off - 8, b);
            const uint64_t xb = zs_u64(a[0] ^ b[0], a[1] ^ b[1]), x0 = zs_u64(a[2] ^ b[2], a[3] ^ b[3]), x1 = zs_u64(a[4] ^ b[4], a[5] ^ b[5]);
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
        // best candidate of the walker's 8 lanes, by data-parallel-primitive moves (xor 1, xor 2 inside a quad, then the mirrored
        // lane of the other quad).  64 KiB units: a max over 64-bit words -- high word key << 17 | distance (keys differ between
        // lanes, so the key decides), low word (q - ip, fwd, back) -- leaves every lane with the winner's words and no LDS
        // round trip (walk 0.857 -> 0.833 ms).  128 KiB units (one workgroup per CU, issue-bound): a max over the keys and two
        // ds_bpermute reads of the winner's words is the shorter instruction sequence (1.00 vs 1.04 ms).
        uint32_t whi = ((uint32_t)key << 17) | off, wlo = idx | (fwd << 8) | (back << 16);
        int best; uint32_t packed, boff;
        if (BIG) {
            best = key;
            best = max(best, __builtin_amdgcn_update_dpp(0, best, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
            best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
            best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x141, 0xF, 0xF, false));    // row_half_mirror: lane i <- lane 7 - i
            const uint32_t bl = (lane & ~7u) + (7u - (uint32_t)(best & 7));      // lane holding the best candidate
            packed = (uint32_t)__shfl((int)wlo, (int)bl);
            boff = (uint32_t)__shfl((int)off, (int)bl);
        } else {
            #define ZS_WMAX(ctrl) { const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)whi, ctrl, 0xF, 0xF, false), \
                                                   olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wlo, ctrl, 0xF, 0xF, false); \
                                    const bool t_ = ohi > whi; whi = t_ ? ohi : whi; wlo = t_ ? olo : wlo; }
            ZS_WMAX(0xB1) ZS_WMAX(0x4E) ZS_WMAX(0x141)
            #undef ZS_WMAX
            best = (int)(whi >> 17); packed = wlo; boff = whi & 0x1FFFFu;
        }
        const uint32_t bq = ip + (packed & 0xFFu);
        uint32_t bfwd = (packed >> 8) & 0xFFu;
        const uint32_t bback = packed >> 16;
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
                    uint32_t a[4], b[4];
                    lds_span<4>(walkLds, ZS_WALK_FRONT + pos + fo, a);
                    lds_span<4>(walkLds, ZS_WALK_FRONT + pos - boff + fo, b);
                    const uint64_t x0 = zs_u64(a[0] ^ b[0], a[1] ^ b[1]), x1 = zs_u64(a[2] ^ b[2], a[3] ^ b[3]);
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
    if (sub == 0) { ZsRangeHdr h; h.nseq = nseq; h.trailing = alive ? end - anchor : ((start < n) ? end - start : 0u); h.litSum = litSum; h.pad = 0; hdrAll[(size_t)slot * (ZS_BLOCK_MAX >> WLOG) + walker] = h; }
}
