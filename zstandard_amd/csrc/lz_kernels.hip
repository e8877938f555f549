// LZ stage of the block encoder: candidate search (k_lz_candidates) and greedy walk with
// look-ahead, recent-offset candidates and the stitch of the walk ranges (k_lz_walk).  The scalar statement
// of the same algorithm is oracle/zso_encoder.c (findCandidates / walkRange / the stitch in compressBlock);
// the two must agree bit for bit.
//
// There is no reference code for this stage (the reference has no encoder);
// what it emits is consumed by entropy_kernels.hip, whose output the reference decoder must accept.
#include "zsmi_device.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// k_lz_candidates<TLOG, NT> : one workgroup of 3 NT wavefronts per LZ unit: an OWNER, a HASHER and a MERGER wavefront per table.
//   table 0 "short": hash of the 5 bytes at p, table 1 "long" (NT == 2: level >= 3): hash of the 8 bytes at p;
//   2^TLOG 32-bit slots each (TLOG 13: units <= 64 KiB, 64 KiB of LDS for both; TLOG 14: units <= 128 KiB, 128 KiB).
//   slot = tag (the 15 hash bits below the index bits) << 17 | position; 0xFFFFFFFF = empty.
// The owner takes 64 consecutive positions per LDS exchange (ds_wrxchg_rtn_b32): every lane leaves its entry and gets the
// slot's previous content back.  Lanes of one instruction that hit the same slot are served in ascending lane order
// (tools/probe/lds_xchg.hip: 0 violations in 2e9 same-slot pairs on MI355X), and a wavefront's LDS instructions execute in
// order: so position p receives exactly the last earlier position inserted with the same slot -- the sequential loop of
// findCandidates in oracle/zso_encoder.c.  Same tag -> distance p - that position; the long table's distance wins.
// Candidates are not compared with the source bytes here (index + tag = 28 hash bits agree; the walk measures every match).
//
// Only the exchanges have to be made in order by one wavefront; everything around them is data parallel, and a wavefront that did
// all of it was alone on its SIMD waiting out every latency (round 2, first shape: 1875 cycles per group of 512 positions, a third of
// them issuing).  So the work of a table is cut in three roles that run side by side, a group (G = 8 steps of 64 positions) apart:
//   hasher, interval i: hashes group i -- lane l takes the 8 positions 8 l .. 8 l + 7 of the group from two coalesced 8-byte loads
//           (prefetched 2 .. 4 groups ahead) -- and leaves an operand word per position (tag << 17 | slot index) in an LDS ring;
//   owner,  interval i: group i - 1: operand -> entry, exchange, distance (same tag and an earlier position) written over the operand;
//   merger, interval i: group i - 2, half of its steps (the other table's merger takes the other half): long distance, else short
//           -> dist[] (low 16 bits), distHi (bit 16, big units), distMask.  (ZS_CAND_MERGERS 0: the hashers merge.)
// One barrier per interval.  The operand ring holds 3 groups per table; a step's 64 words are ZS_CAND_ROW = 65 words apart, which
// spreads both the hashers' writes (8 consecutive positions a lane) and the owners' reads (64 consecutive) over all banks and keeps
// every address of the form base + constant.
// HBM traffic per unit: reads n (twice through L2: both hashers), writes 2 n + n / 8 (+ n / 8).
// ---------------------------------------------------------------------------------------------
#ifndef ZS_CAND_G
#define ZS_CAND_G 8                // steps of 64 positions per group
#endif
#ifndef ZS_CAND_DEPTH
#define ZS_CAND_DEPTH 2            // a register set holds the source loads of this many groups (two sets: 2 .. 4 groups in flight; 4: 0.64 vs 0.52 ms, the unrolled body grows)
#endif
#ifndef ZS_CAND_MERGERS
#define ZS_CAND_MERGERS 1            // 1: a third wavefront per table merges and stores (0: the hashers do)
#endif
#define ZS_CAND_WAVES(NT) ((2 + ZS_CAND_MERGERS) * (NT))
#define ZS_CAND_ROW 65u            // words per step in the operand ring (64 + 1 of padding)
#define ZS_CAND_LDS(TLOG, NT) ((size_t)(NT) * (4u << (TLOG)) + 3u * (NT) * ZS_CAND_G * ZS_CAND_ROW * 4u + (NT) * 256u)   // tables, operand ring, a dummy word per owner lane
#define ZS_SLOT_EMPTY 0xFFFFFFFFu
// hashes made of 24 x 24 -> 32 bit multiplies (v_mul_u32_u24 / v_mad_u32_u24: full rate; a v_mul_lo_u32 is quarter rate and the first
// version's two per long hash were a quarter of the kernel's issue slots).  short: bytes 0-2 and 2-4; long: bytes 0-2, 3-5, 6-7
// (the multiplies read only the low 24 bits of their operands).  Same functions in oracle/zso_encoder.c.
__device__ __forceinline__ uint32_t zs_hash_short(uint32_t lo, uint32_t hi) { return __umul24(lo, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbit(hi, lo, 16), 0x85EBCBu); }
__device__ __forceinline__ uint32_t zs_hash_long(uint32_t lo, uint32_t hi)
{ return __umul24(lo, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbit(hi, lo, 24), 0x85EBCBu) + __umul24(hi >> 16, 0xC2B2AFu); }

template <int TLOG, int NT>
__global__ void __launch_bounds__(64 * ZS_CAND_WAVES(NT))
k_lz_candidates(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
                uint16_t *__restrict__ distAll, uint8_t *__restrict__ distHiAll, uint8_t *__restrict__ distMaskAll)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t candLds[];
    constexpr bool BIG = TLOG > ZS_TABLE_LOG_SMALL;
    constexpr uint32_t G = ZS_CAND_G, GP = G * 64u, H = G / NT;         // H: steps of a group a hasher merges and stores
    constexpr uint32_t M = ZS_CAND_DEPTH;
    static_assert(G == 8, "a hasher lane takes 8 consecutive positions of a group");
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t tab = (wave < NT) ? wave : ((wave < 2 * NT) ? wave - NT : wave - 2 * NT);   // wavefronts [0, NT): owners, [NT, 2 NT): hashers, [2 NT, 3 NT): mergers
    uint32_t *T = candLds + ((size_t)tab << TLOG);
    uint32_t *opnd = candLds + ((size_t)NT << TLOG);                     // [3][NT][GP]
    constexpr uint32_t ROW = ZS_CAND_ROW, GR = G * ROW;                  // words of a group in the ring, per table
    constexpr uint32_t RING = 3u * NT * GR;
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    const size_t slot = (size_t)(ud.firstBlock - block0);
    uint16_t *dist = distAll + slot * ZS_BLOCK_MAX;
    uint8_t *distHi = distHiAll + slot * (ZS_BLOCK_MAX / 8);
    uint8_t *distMask = distMaskAll + slot * (ZS_BLOCK_MAX / 8);     // bit p: position p has a candidate
    const uint32_t hashable = (n >= 8) ? n - 7 : 0;                   // positions [0, hashable) have 8 bytes
    const uint32_t nGroups = (hashable + GP - 1) / GP;
    if (nGroups == 0) return;
    // a position that is not inserted (behind the hashable ones) exchanges with a word of its owner lane's own behind the ring;
    // index relative to the owner's table, < 2^16
    const uint32_t dummyBase = ((uint32_t)(NT - tab) << TLOG) + RING + tab * 64u;
    // merge of a group (both tables' distances -> dist / distMask / distHi): this wavefront's H steps of it
    auto mergeRead = [&](uint32_t g, uint32_t (&mS)[H], uint32_t (&mL)[H]) {
        const uint32_t *xb = opnd + (size_t)(g % 3u) * NT * GR + (NT > 1 ? tab * H : 0u) * ROW + lane;
        #pragma unroll
        for (uint32_t uu = 0; uu < H; uu++) { mS[uu] = xb[uu * ROW]; mL[uu] = (NT > 1) ? xb[GR + uu * ROW] : 0u; }
    };
    auto mergeStore = [&](uint32_t g, const uint32_t (&mS)[H], const uint32_t (&mL)[H]) {
        // candidate bits (and bit 16 of the distances): lane uu keeps step uu's word, so each plane takes one store of H * 8
        // contiguous bytes.  (What is stored for positions behind the hashable ones is never read: the walk stops at them.)
        const uint32_t sbase = g * GP + ((NT > 1) ? tab * H * 64u : 0u);
        uint64_t pmMine = 0, hiMine = 0;
        #pragma unroll
        for (uint32_t uu = 0; uu < H; uu++) {
            const uint32_t dm = (NT > 1) ? (mL[uu] ? mL[uu] : mS[uu]) : mS[uu];
            if (BIG) { const uint64_t hi = __ballot((dm >> 16) != 0); if (lane == uu) hiMine = hi; }
            const uint64_t pm = __ballot(dm != 0);
            if (lane == uu) pmMine = pm;
            dist[sbase + uu * 64 + lane] = (uint16_t)dm;
        }
        if (lane < H) {
            *reinterpret_cast<uint64_t *>(distMask + ((sbase + lane * 64) >> 3)) = pmMine;
            if (BIG) *reinterpret_cast<uint64_t *>(distHi + ((sbase + lane * 64) >> 3)) = hiMine;
        }
    };
    if (ZS_CAND_MERGERS && wave >= 2 * NT) {
        // ---------------- merger: interval i merges group i - 2 ----------------
        __syncthreads(); __syncthreads();
        for (uint32_t i = 2; i <= nGroups + 1; i++) {
            uint32_t mS[H], mL[H];
            mergeRead(i - 2, mS, mL);
            mergeStore(i - 2, mS, mL);
            __syncthreads();
        }
        return;
    }

    if (wave < NT) {
        // ---------------- owner ----------------
        {   // clears its own table (its LDS instructions execute in order: no barrier needed before it uses it)
            uint4 *t4 = reinterpret_cast<uint4 *>(T);
            const uint4 e = make_uint4(ZS_SLOT_EMPTY, ZS_SLOT_EMPTY, ZS_SLOT_EMPTY, ZS_SLOT_EMPTY);
            #pragma unroll 8
            for (uint32_t i = lane; i < (1u << TLOG) / 4; i += 64) t4[i] = e;
        }
        uint32_t ringSlot = 0;                                           // (i - 1) % 3
        __syncthreads();                                                 // interval 0: the hashers fill group 0
        for (uint32_t i = 1; i <= nGroups; i++) {
            uint32_t *ob = opnd + ((size_t)ringSlot * NT + tab) * GR + lane;
            const uint32_t base = (i - 1) * GP + lane;
            uint32_t op[G], entry[G], old[G];
            #pragma unroll
            for (uint32_t u = 0; u < G; u++) op[u] = ob[u * ROW];
            #pragma unroll
            for (uint32_t u = 0; u < G; u++) {                           // the G exchanges go to the LDS back to back
                uint32_t pu = base + u * 64u;
                asm volatile("" : "+v"(pu));                             // one add, then v_and_or (split into a scalar and a lane part it takes two ors)
                entry[u] = (op[u] & 0xFFFE0000u) | pu;
                old[u] = atomicExch(&T[op[u] & 0xFFFFu], entry[u]);
            }
            #pragma unroll
            for (uint32_t u = 0; u < G; u++) {
                // same tag: entry - old is the distance (the tags cancel).  It counts if it is positive: an empty slot reads as
                // position 131071, behind every position; both conditions in one compare: max(old ^ entry, entry - old) < 2^17.
                const uint32_t x = old[u] ^ entry[u], dd = entry[u] - old[u];
                ob[u * ROW] = (max(x, dd) < (1u << 17)) ? dd : 0u;
            }
            ringSlot = (ringSlot == 2u) ? 0u : ringSlot + 1u;
            __syncthreads();
        }
        __syncthreads();                                                 // interval nGroups + 1: the hashers merge the last group
        return;
    }

    // ---------------- hasher ----------------
    // Loads in flight live in two register sets A and B of M groups each, addressed statically; the loop body is
    // [load B | M intervals on A | load A | M intervals on B].  The compiler puts a full s_waitcnt vmcnt(0) at the loop head (it
    // cannot count loads across the back edge): there it only meets the loads of A issued M intervals earlier.
    auto run = [&](auto roleTag) {
        constexpr bool LONG = decltype(roleTag)::value;
        const uint32_t stepOfLane = lane >> 3, l0 = (lane & 7u) * 8u;    // the lane's positions 8 lane + u = step (lane >> 3), owner lane l0 + u
        const uint32_t wbase = stepOfLane * ROW + l0;
        const uint32_t last8 = n - 8;                                    // loads are clamped, never branched around
        auto loadGroup = [&](uint32_t g, uint32_t (&w)[4]) {
            // bytes [o, o + 16) of the unit; a piece that would pass the unit's end is read at the last whole 8 bytes and shifted down
            // (the bytes below the end stay right, nothing behind the end is touched)
            const uint32_t o = g * GP + lane * 8u;
            const uint32_t oa = min(o, last8), ob = min(o + 8u, last8);
            const uint64_t a = zs_load64(s + oa) >> (8u * min(o - oa, 7u));
            const uint64_t b = zs_load64(s + ob) >> (8u * min(o + 8u - ob, 7u));
            w[0] = (uint32_t)a; w[1] = (uint32_t)(a >> 32); w[2] = (uint32_t)b; w[3] = (uint32_t)(b >> 32);
        };
        uint32_t bufA[M][4], bufB[M][4];
        auto loadM = [&](uint32_t g0, uint32_t (&buf)[M][4]) {
            #pragma unroll
            for (uint32_t k = 0; k < M; k++) loadGroup(min(g0 + k, nGroups), buf[k]);
        };
        auto iter = [&](auto wholeTag, uint32_t i, const uint32_t (&w)[4]) {
            constexpr bool WHOLE = decltype(wholeTag)::value;            // group i lies wholly inside the hashable positions
            const uint32_t ringSlot = i % 3u;
            // distances of group i - 2 (both tables): reads issued first, they travel under the hashing
            uint32_t mS[H], mL[H];
            if (!ZS_CAND_MERGERS && i >= 2) mergeRead(i - 2, mS, mL);
            if (i < nGroups) {
                uint32_t *ob = opnd + ((size_t)ringSlot * NT + tab) * GR + wbase;
                const uint32_t pbase = i * GP + lane * 8u;
                #pragma unroll
                for (uint32_t u = 0; u < G; u++) {
                    const uint32_t k = u >> 2, sh = u & 3u;
                    const uint32_t lo = sh ? __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh) : w[k];
                    const uint32_t hi = sh ? __builtin_amdgcn_alignbyte(w[(k + 2) & 3], w[k + 1], sh) : w[k + 1];   // u == 4 needs no fourth dword beyond w[3]
                    uint32_t h = LONG ? zs_hash_long(lo, hi) : zs_hash_short(lo, hi);
                    asm volatile("" : "+v"(h));                          // (keeps the compiler from folding the shifts below into two more multiplies)
                    // tag << 17 | index: the hash rotated left by TLOG holds both (bits 31..17 and TLOG-1..0)
                    const uint32_t r = __builtin_amdgcn_alignbit(h, h, 32 - TLOG);
                    ob[u] = (WHOLE || pbase + u < hashable) ? (r & (0xFFFE0000u | ((1u << TLOG) - 1u))) : ((r & 0xFFFE0000u) | (dummyBase + l0 + u));
                }
            }
            if (!ZS_CAND_MERGERS && i >= 2) mergeStore(i - 2, mS, mL);
            __syncthreads();
        };
        auto step = [&](uint32_t i, const uint32_t (&w)[4]) {
            if (i > nGroups + 1) return;
#ifdef ZS_CAND_NOWHOLE
            iter(std::false_type{}, i, w);
#else
            if ((i + 1) * GP + 8u <= n) iter(std::true_type{}, i, w); else iter(std::false_type{}, i, w);
#endif
        };
        loadM(0, bufA);
        for (uint32_t g0 = 0; g0 <= nGroups + 1; g0 += 2 * M) {
            loadM(g0 + M, bufB);
            #pragma unroll
            for (uint32_t k = 0; k < M; k++) step(g0 + k, bufA[k]);
            loadM(g0 + 2 * M, bufA);
            #pragma unroll
            for (uint32_t k = 0; k < M; k++) step(g0 + M + k, bufB[k]);
        }
    };
    if (tab == 0) run(std::false_type{}); else run(std::true_type{});
}

// ---------------------------------------------------------------------------------------------
// k_lz_walk<NW> : one workgroup of NW / 8 wavefronts per LZ unit; NW = 64 (unit <= 64 KiB) or 128 (<= 128 KiB).
// The unit's source bytes are staged in LDS once (NW KiB + pads), so every compare of the walk is an LDS read and
// the unit is fetched from HBM once.  The lanes are NW independent walkers of 8 lanes; walker g walks the walk
// range g (1 KiB).  A walker step looks at the 64 positions from ip: the candidate bits of stage 1 (LDS copy of the
// bit plane), plus, for the first ZS_REPWIN = 8 positions, the positions where one of the walker's two recent offsets repeats
// 4 bytes (lane sub tries ip + sub).  The first LOOK positions holding a candidate go one per lane; a lane
// takes the recent offset or fetches the stage-1 distance (global), compares 16 bytes forward (the score counts ZS_FCAP of
// them) and 8 bytes backward (into the pending literals) and scores; the best one of the walker becomes a sequence
// (extended by the walker's 8 lanes if it hit the 16-byte cap).  A match may run past the range end, ZS_CROSS_MAX bytes at most.
// After a barrier one lane per range stitches (compressBlock in oracle/zso_encoder.c): reach = running maximum of the
// ranges' last match ends; a range drops the records an earlier range's match covers and cuts the front of one that
// straddles; its header gets first / nseq / litSum / trailing for its territory [max(start, reach before), max(end, reach)).
// Scalar statement: walkRange + the stitch in oracle/zso_encoder.c.
// ---------------------------------------------------------------------------------------------
#define ZS_WALK_FRONT 16u          // LDS bytes in front of the unit (backward reads near position 0)
#define ZS_WALK_TAIL  144u         // zero bytes behind the unit (forward reads near the end)
#define ZS_WALK_LDS(CAPB) (ZS_WALK_FRONT + (CAPB) + ZS_WALK_TAIL + (CAPB) / 8 + 16 + ((CAPB) >> ZS_WALK_LOG) * 16)   // source + candidate bit plane + per-range results

// K dwords of the LDS copy starting at any byte offset, fetched as K + 1 aligned dwords and shifted into place
// (an unaligned ds_read_b64 / b128 costs the LDS several passes: SQ_LDS_UNALIGNED_STALL was 80 % of its busy time)
// (byteOff is an LDS address: the kernel's dynamic LDS starts at address 0 - there is no static LDS in it, which zsmi_createCtx checks
// through hipFuncGetAttributes - so the compiler has no symbol to add to every address)
typedef const __attribute__((address_space(3))) uint32_t *ZsLdsU32;
template <int K>
__device__ __forceinline__ void lds_span(uint32_t byteOff, uint32_t (&out)[K])
{
    ZsLdsU32 d = (ZsLdsU32)(uintptr_t)(byteOff & ~3u);
    uint32_t w[K + 1];
    #pragma unroll
    for (int k = 0; k <= K; k++) w[k] = d[k];
    const uint32_t sh = byteOff & 3u;
    #pragma unroll
    for (int k = 0; k < K; k++) out[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
}
__device__ __forceinline__ uint64_t zs_u64(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }
template <int NW>
__global__ void __launch_bounds__(NW * 8)
k_lz_walk(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
          const uint16_t *__restrict__ distAll, const uint8_t *__restrict__ distHiAll, const uint8_t *__restrict__ distMaskAll,
          ZsSeqRec *__restrict__ seqAll, ZsRangeHdr *__restrict__ hdrAll, int look)
{
    constexpr uint32_t WLOG = ZS_WALK_LOG, WSIZE = 1u << WLOG;                     // bytes per walk range
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
    uint4 *res = reinterpret_cast<uint4 *>(walkLds + ZS_WALK_FRONT + CAP + ZS_WALK_TAIL + CAP / 8 + 16);   // per range: nseq, last match end, sum of match lengths
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
    const uint32_t hashable = (n >= 8) ? n - 7 : 0;
    // candidate bits of the positions that can start a match; the word behind them reads as zero
    {
        const uint32_t words = (hashable + 63) >> 6;
        for (uint32_t i = tid; i <= words; i += NW * 8) lm[i] = (i < words) ? distMask[i] : 0ull;
    }
    __syncthreads();

    const uint32_t start = walker << WLOG;
    const uint32_t blockStart = start & ~(ZS_BLOCK_MAX - 1);                     // the walker's block inside the unit
    const uint32_t blockN = (blockStart < n) ? min(n - blockStart, ZS_BLOCK_MAX) : 0u;
    const uint32_t blockEnd = blockStart + blockN;
    const bool alive = (start < n) && (blockN >= 16);
    const uint32_t end = min(start + WSIZE, n);
    const uint32_t limit = min(end + ZS_CROSS_MAX, blockEnd);                     // matches end at or before this
    const uint32_t scanEnd = alive ? min(end, hashable) : 0;

    uint32_t ip = start, anchor = start, nseq = 0, mlSum = 0, rep0 = 0, rep1 = 0;
    const uint32_t bit0 = 1u << sub, bit1 = 256u << sub;
    for (;;) {
        const bool run = ip < scanEnd;
        if (!__any(run)) break;
        const uint32_t wend = min(ip + ZS_WINDOW, scanEnd);
        // ---- recent offsets: lane sub tries position ip + sub (ZS_REPWIN = 8 = the walker's lanes) ----
        uint32_t rm0 = 0, rm1 = 0;                                               // bit i: rep0 / rep1 repeats 4 bytes at ip + i
        {
            const uint32_t q = ip + sub;
            const bool in0 = run && q < wend && q + 4 <= limit;
            const bool t0 = in0 && rep0 != 0 && q >= rep0, t1 = in0 && rep1 != 0 && q >= rep1;
            uint32_t a[1], b[1], c[1];
            lds_span<1>(ZS_WALK_FRONT + q, a);
            lds_span<1>(ZS_WALK_FRONT + (t0 ? q - rep0 : q), b);
            lds_span<1>(ZS_WALK_FRONT + (t1 ? q - rep1 : q), c);
            // the walker's two bytes of flags: each lane's bit (1 << sub, 256 << sub) or-ed over its 8 lanes by three data-parallel moves
            // (xor 1, xor 2 inside a quad, the mirrored lane of the other quad) -- a ballot cost two more extracts per flag
            uint32_t v = ((t0 && a[0] == b[0]) ? bit0 : 0u) | ((t1 && a[0] == c[0]) ? bit1 : 0u);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);     // row_half_mirror: lane i <- lane 7 - i
            rm0 = v & 0xFFu; rm1 = v >> 8;
        }
        // ---- window: candidate bits of [ip, ip + 64) from LDS (the same for the walker's 8 lanes) or'ed with the recent-offset
        //      bits; lane sub takes the sub-th candidate position.  32-bit words: two funnel shifts give the window ----
        uint32_t mlo = 0, mhi = 0;
        if (run) {
            const uint32_t *lw = reinterpret_cast<const uint32_t *>(lm) + (ip >> 5);
            const uint32_t w0 = lw[0], w1 = lw[1], w2 = lw[2], sh = ip & 31u, wlen = wend - ip;
            mlo = __builtin_amdgcn_alignbit(w1, w0, sh); mhi = __builtin_amdgcn_alignbit(w2, w1, sh);
            const uint64_t keep = ~0ull >> (64u - wlen);                                    // 1 <= wlen <= 64 while run
            mlo &= (uint32_t)keep; mhi &= (uint32_t)(keep >> 32);
            mlo |= rm0 | rm1;
        }
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
        const uint32_t q = ip + idx;
        const bool isR0 = active && idx < 8u && ((rm0 >> (idx & 7u)) & 1u) != 0, isR1 = active && !isR0 && idx < 8u && ((rm1 >> (idx & 7u)) & 1u) != 0;
        const bool isRep = isR0 || isR1;
        uint32_t off = isR0 ? rep0 : (isR1 ? rep1 : 0u);
        if (active && !isRep) {
            off = (uint32_t)dist[q];
            if (BIG) off |= (((uint32_t)distHi[q >> 3] >> (q & 7u)) & 1u) << 16;
        }
        // ---- compare from LDS: 16 bytes forward, 8 bytes backward, both sides ----
        uint32_t fwd = 0, back = 0;
        int key = 0;
        if (active) {
            uint32_t a[6], b[6];                                                 // bytes [q - 8, q + 16) of both sides
            lds_span<6>(ZS_WALK_FRONT + q - 8, a);
            lds_span<6>(ZS_WALK_FRONT + q - off - 8, b);
            const uint64_t xb = zs_u64(a[0] ^ b[0], a[1] ^ b[1]), x0 = zs_u64(a[2] ^ b[2], a[3] ^ b[3]), x1 = zs_u64(a[4] ^ b[4], a[5] ^ b[5]);
            const uint32_t cap = min(limit - q, ZS_LCAP);
            const uint32_t n0 = x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : 8u;
            const uint32_t n1 = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8u;
            fwd = min((n0 < 8u) ? n0 : 8u + n1, cap);
            const uint32_t maxBack = min(min(q - anchor, q - off), ZS_BCAP);
            back = min(xb ? ((uint32_t)__builtin_clzll(xb) >> 3) : 8u, maxBack);
            if (fwd >= (isRep ? ZS_REPMIN : ZS_MINMATCH)) {
                const int gain = (int)(min(fwd, ZS_FCAP) + back) * 4 - (isRep ? 0 : (int)zs_highbit(off + 1)) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
                key = ((gain + 2048) << 3) | (int)(7u - sub);
            }
        }
        // best candidate of the walker's 8 lanes, by data-parallel-primitive moves (xor 1, xor 2 inside a quad, then the mirrored
        // lane of the other quad): a max over the keys, then two ds_bpermute reads of the winner's words
        int best = key;
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x141, 0xF, 0xF, false));    // row_half_mirror: lane i <- lane 7 - i
        const uint32_t bl = (lane & ~7u) + (7u - (uint32_t)(best & 7));          // lane holding the best candidate
        const uint32_t packed = (uint32_t)__shfl((int)(idx | (fwd << 8) | (back << 16)), (int)bl);
        const uint32_t boff = (uint32_t)__shfl((int)off, (int)bl);
        const uint32_t bq = ip + (packed & 0xFFu);
        uint32_t bfwd = (packed >> 8) & 0xFFu;
        const uint32_t bback = packed >> 16;
        const bool took = run && best != 0;
        // ---- long match: the walker's 8 lanes extend it, 128 bytes per round (LDS) ----
        bool need = took && bfwd == ZS_LCAP && (limit - bq) > ZS_LCAP;
        const bool extended = need;
        uint32_t pos = bq + ZS_LCAP;
        while (__any(need)) {
            uint32_t nb = 0;
            if (need) {
                const uint32_t cap = limit - pos;          // pos < limit while need
                const uint32_t fo = 16 * sub;
                if (fo < cap) {
                    uint32_t a[4], b[4];
                    lds_span<4>(ZS_WALK_FRONT + pos + fo, a);
                    lds_span<4>(ZS_WALK_FRONT + pos - boff + fo, b);
                    const uint64_t x0 = zs_u64(a[0] ^ b[0], a[1] ^ b[1]), x1 = zs_u64(a[2] ^ b[2], a[3] ^ b[3]);
                    const uint32_t n0 = x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : 8u;
                    const uint32_t n1 = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8u;
                    nb = min((n0 < 8u) ? n0 : 8u + n1, cap - fo);
                }
            }
            const uint64_t stopm = __ballot(nb < 16u);
            const uint32_t g8 = __builtin_amdgcn_ubfe((grp & 4u) ? (uint32_t)(stopm >> 32) : (uint32_t)stopm, 8u * (grp & 3u), 8u);
            const uint32_t f = g8 ? (uint32_t)__builtin_ctz(g8) : 0u;
            const uint32_t part = (uint32_t)__shfl((int)nb, (int)((lane & ~7u) + f));
            if (need) {
                if (g8) { pos += 16 * f + part; need = false; }
                else { pos += 128; if (pos >= limit) { pos = limit; need = false; } }
            }
        }
        if (extended) bfwd = pos - bq;
        if (took) {
            const uint32_t mstart = bq - bback, ml = bback + bfwd;
            if (sub == 0) { ZsSeqRec r; r.x = zs_rec_x(mstart - anchor, ml, boff); r.y = zs_rec_y(boff, mstart - blockStart); seqs[nseq] = r; }
            nseq++; mlSum += ml;
            ip = bq + bfwd; anchor = ip;
            if (boff == rep1) { rep1 = rep0; rep0 = boff; }
            else if (boff != rep0) { rep1 = rep0; rep0 = boff; }
        } else if (run) ip = wend;
    }
    if (sub == 0) res[walker] = make_uint4(nseq, nseq ? anchor : 0u, mlSum, 0u);
    __syncthreads();

    // ---- the stitch: wavefront b takes block b of the unit, lane r its walk range r ----
    if (wave < (BIG ? 2u : 1u)) {
        const uint32_t bStart = wave * ZS_BLOCK_MAX;
        const uint32_t bN = (bStart < n) ? min(n - bStart, ZS_BLOCK_MAX) : 0u;
        const uint32_t bEnd = bStart + bN;
        const uint32_t rg = wave * 64 + lane;                                     // range index in the unit
        const uint4 rr = res[rg];
        const uint32_t ns = rr.x, le = rr.y;
        uint32_t sumMl = rr.z;
        // reach before me: maximum of the earlier ranges' last match ends (and the block start)
        uint32_t incl = max(le, bStart);
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl = max(incl, o); }
        uint32_t own = (uint32_t)__shfl_up((int)incl, 1); if (lane == 0) own = bStart;
        const uint32_t reach = incl;
        const uint32_t rs = min(rg << WLOG, bEnd), re = min((rg << WLOG) + WSIZE, bEnd);   // the range, cut at the block end
        const uint32_t es = max(rs, own), te = max(re, reach);
        ZsSeqRec *recs = seqAll + (size_t)slot * (ZS_BLOCK_MAX / 4) + (size_t)rg * (WSIZE / 4);
        uint32_t f = 0;
        if (ns && own > rs) {
            while (f < ns) {
                const ZsSeqRec r = recs[f];
                const uint32_t st = zs_rec_pos(r.y) + bStart, ml = zs_rec_ml(r.x), off = zs_rec_off(r.x, r.y);
                if (st + ml <= own) { sumMl -= ml; f++; continue; }              // covered by an earlier range's match
                if (st < own) {                                                  // straddles: the front goes
                    const uint32_t left = st + ml - own;
                    if (left < ZS_MINMATCH) { sumMl -= ml; f++; continue; }
                    ZsSeqRec w; w.x = zs_rec_x(0, left, off); w.y = zs_rec_y(off, own - bStart); recs[f] = w;
                    sumMl -= ml - left;
                } else { ZsSeqRec w; w.x = zs_rec_x(st - es, ml, off); w.y = r.y; recs[f] = w; }   // first kept one: its literals count from es
                break;
            }
        }
        const uint32_t kept = ns - f;
        const uint32_t lastEnd = kept ? le : es;
        ZsRangeHdr h; h.nseq = kept; h.first = f; h.trailing = te - lastEnd; h.litSum = (lastEnd - es) - sumMl;
        hdrAll[(size_t)slot * ZS_WALK_RANGES + rg] = h;
    }
}
