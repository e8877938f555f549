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
// (undocumented; tools/probe/lds_xchg.hip: 0 violations in 2e9 same-slot pairs on MI355X, run as tests/test_gpu_probes.py: byte-identity with oracle E is
// conditional on it, validity is not - a slot owner served out of order gives a non-positive distance and the candidate is dropped), and a wavefront's LDS instructions execute in
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
//           -> dist[] (low 16 bits), distHi (bit 16, big units).
// One barrier per interval.  The operand ring holds 3 groups per table; a step's 64 words are ZS_CAND_ROW = 65 words apart, which
// spreads both the hashers' writes (8 consecutive positions a lane) and the owners' reads (64 consecutive) over all banks and keeps
// every address of the form base + constant.
// Runs of one byte (round 4, ZS_CAND_RUNS): zeros in a binary or a line of blanks give consecutive positions the SAME operand - up to 64 lanes
// of one exchange on one slot, which the LDS serves one after the other (repetitive class: 1.33 ms against 0.46).  What such a run's lanes
// get is known without asking: every lane but the run's first the entry of the lane below, distance 1, and the slot keeps the run's last
// entry.  The hasher sees the runs in its bytes (a lane whose G positions + 7 bytes are one byte, and the lanes next to it): it MARKS the
// run's positions but the first (a bit a position, next to the ring, and a word a group that says whether any is marked), and sends the
// marked positions that have a marked one above them to the owner lane's dummy word instead of the slot.  So only a run's first lane (the
// slot's old content) and last lane (leaves its entry; served after the first: ascending lane order, as everywhere here) meet at the slot,
// and the merger gives a marked position the distance 1.  The owner - the wavefront the other two wait for - runs the same instructions as
// without any of this (finding the runs there cost 0.13 ms of 0.46 on data without runs; a flag word a group read there 0.06; marks carried
// through its result words 0.10: each role's instructions count), a group without a flat lane costs the hasher the test and the merger a
// word's read.  Runs are cut at a step's border: its lane 0 asks the slot, which holds the step before's last entry.
// HBM traffic per unit: reads n (twice through L2: both hashers), writes 2 n + n / 8 (+ n / 8).
// ---------------------------------------------------------------------------------------------
#ifndef ZS_CAND_G
#define ZS_CAND_G 8                // steps of 64 positions per group, units <= 64 KiB (16: 85 KiB of LDS, one workgroup a CU: 1.54 vs 0.48 ms)
#endif
#ifndef ZS_CAND_G_BIG
#define ZS_CAND_G_BIG 16           // the same for units > 64 KiB: their 2^14-slot tables allow one workgroup a CU anyway (128 + 25 KiB), and an interval's three LDS round trips and its barrier
#endif                             // then cover 1024 positions instead of 512
#define ZS_CAND_GOF(TLOG) ((TLOG) > ZS_TABLE_LOG_SMALL ? ZS_CAND_G_BIG : ZS_CAND_G)
#ifndef ZS_CAND_DEPTH
#define ZS_CAND_DEPTH 2            // a register set holds the source loads of this many groups (two sets: 2 .. 4 groups in flight; 4: 0.64 vs 0.52 ms, the unrolled body grows)
#endif
#ifndef ZS_CAND_RUNS
#define ZS_CAND_RUNS 1             // runs of one byte: the inner positions' exchanges are kept off the run's slot (below); 0: every position exchanges there
#endif
#define ZS_CAND_WAVES(NT) (3 * (NT))
#define ZS_CAND_ROW 65u            // words per step in the operand ring (64 + 1 of padding)
#define ZS_CAND_LDS(TLOG, NT) ((size_t)(NT) * (4u << (TLOG)) + 3u * (NT) * ZS_CAND_GOF(TLOG) * ZS_CAND_ROW * 4u + (NT) * 256u + 3u * 16u * 8u + 32u)   // tables, operand ring, a dummy word per owner lane, the run marks of the groups in the ring (a bit a position) and a flag a group
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
                uint16_t *__restrict__ distAll, uint8_t *__restrict__ distHiAll, uint32_t *__restrict__ candCount)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t candLds[];
    constexpr bool BIG = TLOG > ZS_TABLE_LOG_SMALL;
    constexpr uint32_t G = ZS_CAND_GOF(TLOG), GP = G * 64u, H = G / NT;  // H: steps of a group a hasher merges and stores
    constexpr uint32_t M = ZS_CAND_DEPTH;
    constexpr uint32_t LPS = 64u / G;                                   // hasher lanes per step of 64 positions: a lane takes G consecutive positions of the group
    constexpr uint32_t WD = G / 4u + 2u;                                // dwords that hold a lane's G positions and the 7 bytes behind them
    static_assert(G == 8 || G == 16, "a hasher lane takes G consecutive positions of a group: 8 or 16");
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t tab = (wave < NT) ? wave : ((wave < 2 * NT) ? wave - NT : wave - 2 * NT);   // wavefronts [0, NT): owners, [NT, 2 NT): hashers, [2 NT, 3 NT): mergers
    uint32_t *T = candLds + ((size_t)tab << TLOG);
    uint32_t *opnd = candLds + ((size_t)NT << TLOG);                     // [3][NT][GP]
    constexpr uint32_t ROW = ZS_CAND_ROW, GR = G * ROW;                  // words of a group in the ring, per table
    constexpr uint32_t RING = 3u * NT * GR;
    uint64_t *runMarks = reinterpret_cast<uint64_t *>(opnd + RING + NT * 64u);   // [3][16]: per group in the ring and step, the marked positions (ZS_CAND_RUNS; both tables' hashers see the same: table 0's writes)
    uint32_t *runFlag = opnd + RING + NT * 64u + 3u * 16u * 2u;           // [3][2]: the group has marked positions, a word a merger (set by table 0's hasher, cleared by the merger)
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    const size_t slot = (size_t)(ud.firstBlock - block0);
    uint16_t *dist = distAll + slot * ZS_BLOCK_MAX;
    uint8_t *distHi = distHiAll + slot * (ZS_BLOCK_MAX / 8);
    const uint32_t hashable = (n >= 8) ? n - 7 : 0;                   // positions [0, hashable) have 8 bytes
    const uint32_t nGroups = (hashable + GP - 1) / GP;
    // candidate positions of the unit, counted by the mergers (a word each: candCount[2 slot + table]): k_lz_walk leaves a unit with fewer than
    // n >> ZS_MATCHLESS_SHIFT of them alone (findCandidates in oracle/zso_encoder.c: matchless units)
    if (threadIdx.x < 2) candCount[2 * slot + threadIdx.x] = 0;          // (written again by the mergers at their end: same wavefront order does not matter, they add nothing before)
    if (nGroups == 0) return;
    // A unit of one repeated byte needs no candidates: the walk kernel tests for exactly this (same condition) and skips its walk, the
    // blocks become RLE blocks.  Ordinary data fails the test on its first 16 bytes.
    if (n >= 16 && (n & 15u) == 0) {
        uint4 f;
        __builtin_memcpy(&f, s, 16);
        const uint32_t splat = (f.x & 0xFFu) * 0x01010101u;
        if (((f.x ^ splat) | (f.y ^ splat) | (f.z ^ splat) | (f.w ^ splat)) == 0) {     // (the same for every thread)
            if (threadIdx.x == 0) candLds[0] = 0;
            __syncthreads();
            uint32_t mixed = 0;
            for (uint32_t i = threadIdx.x * 16u; i < n && !mixed; i += ZS_CAND_WAVES(NT) * 64u * 16u) {
                uint4 v;
                __builtin_memcpy(&v, s + i, 16);
                mixed = (v.x ^ splat) | (v.y ^ splat) | (v.z ^ splat) | (v.w ^ splat);
            }
            if (mixed) candLds[0] = 1u;
            __syncthreads();
            const bool uniform = candLds[0] == 0;
            __syncthreads();                                             // (the word belongs to a table the owners clear next)
            if (uniform) return;
        }
    }
    // a position that is not inserted (behind the hashable ones) exchanges with a word of its owner lane's own behind the ring;
    // index relative to the owner's table, < 2^16
    const uint32_t dummyBase = ((uint32_t)(NT - tab) << TLOG) + RING + tab * 64u;
    // merge of a group (both tables' distances -> dist / distHi): this wavefront's H steps of it
    auto mergeRead = [&](uint32_t g, uint32_t (&mS)[H], uint32_t (&mL)[H]) {
        const uint32_t *xb = opnd + (size_t)(g % 3u) * NT * GR + (NT > 1 ? tab * H : 0u) * ROW + lane;
        #pragma unroll
        for (uint32_t uu = 0; uu < H; uu++) { mS[uu] = xb[uu * ROW]; mL[uu] = (NT > 1) ? xb[GR + uu * ROW] : 0u; }
    };
    uint32_t found = 0;                                                 // mergers: candidate positions among this wavefront's steps (exact below matchlessBelow, which is all k_lz_walk asks)
    const uint32_t matchlessBelow = n >> ZS_MATCHLESS_SHIFT;
    auto mergeStore = [&](uint32_t g, const uint32_t (&mS)[H], const uint32_t (&mL)[H]) {
        // bit 16 of the distances (big units): lane uu keeps step uu's word, so the plane takes one store of H * 8 contiguous bytes.
        // (What is stored for positions behind the hashable ones is never used: the walk cuts them off its window.)
        const uint32_t sbase = g * GP + ((NT > 1) ? tab * H * 64u : 0u);
        uint64_t hiMine = 0;
        #pragma unroll
        for (uint32_t uu = 0; uu < H; uu++) {
            const uint32_t dm = (NT > 1) ? (mL[uu] ? mL[uu] : mS[uu]) : mS[uu];
            if (BIG) { const uint64_t hi = __ballot((dm >> 16) != 0); if (lane == uu) hiMine = hi; }
            // (counted only until the unit cannot be matchless any more - on ordinary data that is its first group -; only a unit's last group reaches behind
            //  the hashable positions: the test is uniform for every other)
            if (found < matchlessBelow) {
                if ((g + 1u) * GP <= hashable) found += (uint32_t)__popcll(__ballot(dm != 0));
                else found += (uint32_t)__popcll(__ballot(dm != 0 && sbase + uu * 64 + lane < hashable));
            }
            dist[sbase + uu * 64 + lane] = (uint16_t)dm;
        }
        if (BIG && lane < H) *reinterpret_cast<uint64_t *>(distHi + ((sbase + lane * 64) >> 3)) = hiMine;
    };
    if (wave >= 2 * NT) {
        // ---------------- merger: interval i merges group i - 2 ----------------
        __syncthreads(); __syncthreads();
        for (uint32_t i = 2; i <= nGroups + 1; i++) {
            uint32_t mS[H], mL[H];
            mergeRead(i - 2, mS, mL);
#if ZS_CAND_RUNS
            if (__builtin_amdgcn_readfirstlane((int)runFlag[((i - 2) % 3u) * 2u + tab])) {
                if (lane == 0) runFlag[((i - 2) % 3u) * 2u + tab] = 0;
                #pragma unroll
                for (uint32_t uu = 0; uu < H; uu++) {
                    const uint64_t marks = runMarks[((i - 2) % 3u) * 16u + ((NT > 1) ? tab * H : 0u) + uu];
                    if ((marks >> lane) & 1ull) { if (NT > 1) mL[uu] = 1u; else mS[uu] = 1u; }
                }
            }
#endif
            mergeStore(i - 2, mS, mL);
            __syncthreads();
        }
        if (lane == 0) candCount[2 * slot + tab] = found;
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
        const uint32_t stepOfLane = lane / LPS, l0 = (lane % LPS) * G;    // the lane's positions G lane + u = step (lane / LPS), owner lane l0 + u
        const uint32_t wbase = stepOfLane * ROW + l0;
        const uint32_t last8 = n - 8;                                    // loads are clamped, never branched around
        auto loadGroup = [&](uint32_t g, uint32_t (&w)[WD]) {
            // bytes [o, o + 4 WD) of the unit; a piece that would pass the unit's end is read at the last whole 8 bytes and shifted down
            // (the bytes below the end stay right, nothing behind the end is touched)
            const uint32_t o = g * GP + lane * G;
            #pragma unroll
            for (uint32_t k = 0; k < WD / 2; k++) {
                const uint32_t ok = o + 8u * k, oc = min(ok, last8);
                const uint64_t v = zs_load64(s + oc) >> (8u * min(ok - oc, 7u));
                w[2 * k] = (uint32_t)v; w[2 * k + 1] = (uint32_t)(v >> 32);
            }
        };
        uint32_t bufA[M][WD], bufB[M][WD];
        auto loadM = [&](uint32_t g0, uint32_t (&buf)[M][WD]) {
            #pragma unroll
            for (uint32_t k = 0; k < M; k++) loadGroup(min(g0 + k, nGroups), buf[k]);
        };
        auto iter = [&](auto wholeTag, uint32_t i, const uint32_t (&w)[WD]) {
            constexpr bool WHOLE = decltype(wholeTag)::value;            // group i lies wholly inside the hashable positions
            const uint32_t ringSlot = i % 3u;
            if (i < nGroups) {
                uint32_t *ob = opnd + ((size_t)ringSlot * NT + tab) * GR + wbase;
                const uint32_t pbase = i * GP + lane * G;
                #pragma unroll
                for (uint32_t u = 0; u < G; u++) {
                    const uint32_t k = u >> 2, sh = u & 3u;
                    const uint32_t lo = sh ? __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh) : w[k];
                    const uint32_t hi = sh ? __builtin_amdgcn_alignbyte(w[min(k + 2, WD - 1)], w[k + 1], sh) : w[k + 1];   // (sh == 0 at the last k needs no dword beyond w[WD - 1])
                    uint32_t h = LONG ? zs_hash_long(lo, hi) : zs_hash_short(lo, hi);
                    asm volatile("" : "+v"(h));                          // (keeps the compiler from folding the shifts below into two more multiplies)
                    // tag << 17 | index: the hash rotated left by TLOG holds both (bits 31..17 and TLOG-1..0)
                    const uint32_t r = __builtin_amdgcn_alignbit(h, h, 32 - TLOG);
                    ob[u] = (WHOLE || pbase + u < hashable) ? (r & (0xFFFE0000u | ((1u << TLOG) - 1u))) : ((r & 0xFFFE0000u) | (dummyBase + l0 + u));
                }
#if ZS_CAND_RUNS
                // flat: the lane's bytes (its G positions and the 7 behind) are one byte - its positions but the first hash what the position
                // below them hashes; the first too if the lane below is flat with the same byte (and belongs to the same step of 64 positions).
                // Groups with such lanes (rare on ordinary data: this wavefront's instructions are the kernel's time as much as the owner's)
                // get the operands of their inner positions written again, with the dummy word's index.
                // (first a one-instruction test that every flat lane passes: the full test only in groups where some lane does)
                bool flat = false;
                uint64_t flatLanes = 0;
                if (WHOLE && __ballot(w[0] == w[1])) {
                    uint32_t diff = w[0] ^ __builtin_amdgcn_alignbyte(w[0], w[0], 1);
                    #pragma unroll
                    for (uint32_t k = 1; k < WD; k++) diff |= w[k] ^ w[k - 1];
                    flat = diff == 0;
                    flatLanes = __ballot(flat);
                }
                if (flatLanes) {
                    const uint32_t key = flat ? ((w[0] & 0xFFu) | 0x100u) : 0u;       // (never 0 for a flat lane)
                    const uint32_t keyBelow = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x138, 0xF, 0xF, true);   // wave_shr:1 (lane 0 reads 0)
                    const uint32_t keyAbove = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x130, 0xF, 0xF, true);   // wave_shl:1 (lane 63 reads 0)
                    const bool markFirst = flat && keyBelow == key && (lane % LPS) != 0u;
                    const bool innerLast = flat && keyAbove == key && (lane % LPS) != LPS - 1u;   // the position above the lane's last one is marked too
                    if (flat) {
                        uint32_t h = LONG ? zs_hash_long(w[0], w[1]) : zs_hash_short(w[0], w[1]);     // every position of the lane hashes the same bytes
                        asm volatile("" : "+v"(h));
                        const uint32_t away = (__builtin_amdgcn_alignbit(h, h, 32 - TLOG) & 0xFFFE0000u) | (dummyBase + l0);
                        if (markFirst) ob[0] = away;
                        #pragma unroll
                        for (uint32_t u = 1; u + 1 < G; u++) ob[u] = away + u;
                        if (innerLast) ob[G - 1] = away + (G - 1);
                    }
                    if (!LONG) {                                             // the marks: G bits a lane, LPS lanes a step; a flag a merger
                        if (lane < NT) runFlag[ringSlot * 2u + lane] = 1u;
                        const uint32_t bits = flat ? (((1u << G) - 2u) | (markFirst ? 1u : 0u)) : 0u;
                        if (G == 8) reinterpret_cast<uint8_t *>(runMarks + ringSlot * 16u)[lane] = (uint8_t)bits;
                        else reinterpret_cast<uint16_t *>(runMarks + ringSlot * 16u)[lane] = (uint16_t)bits;
                    }
                }
#endif
            }
            __syncthreads();
        };
        auto step = [&](uint32_t i, const uint32_t (&w)[WD]) {
            if (i > nGroups + 1) return;
            if ((i + 1) * GP + 8u <= n) iter(std::true_type{}, i, w); else iter(std::false_type{}, i, w);
        };
#if ZS_CAND_RUNS
        if (!LONG && lane < 6u) runFlag[lane] = 0;                         // (this wavefront's LDS instructions execute in order; the mergers read after two barriers)
#endif
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
// k_lz_walk<LPW, LOOK, REPWIN, BIG, NT> : one workgroup of NT threads per LZ unit (BIG: 64 KiB < unit <= 128 KiB).
// The unit's source bytes are staged in LDS once (+ pads): every compare of the walk is an LDS read and the unit is fetched from HBM
// once.  The unit is cut in walk ranges of R = 1 << rangeLog bytes (256; 512 at levels <= 2); a WALKER = 4 adjacent lanes takes ranges
// from a queue (an LDS counter) and walks each on its own (walkRange in oracle/zso_encoder.c).  A step looks at the positions from ip to
// the end of the fourth aligned group of 8: lane j of the walker has group j's stage-1 distances brought into its slot of the wavefront's
// exchange buffer in LDS (16 coalesced bytes by LDS-DMA: the walker's 64 bytes are one cache line; requested a step ahead, right after ip
// is known) and turns them into 8 candidate bits; plus, for the first REPWIN positions, the positions where one of the walker's two recent offsets
// repeats 4 bytes.  The first LOOK such positions are scored, LOOK / 4 per lane: recent offset or the distance from the exchange buffer,
// 8 bytes forward and 4 backward compared in LDS; the walker's best becomes a record (start, length, offset) in the range's slots of
// recAll, after its lanes measured its whole length (16 bytes a lane and round).  A match may run past the range end, ZS_CROSS_MAX bytes
// at most, never past the block end.
// (Round 2 gave a walker 8 lanes with one candidate each and ranges of 1 KiB: of its ~215 vector instructions a step ~100 were the
// same in all 8 lanes, and 8 candidates a step were measured of which the next step measured most again.  The first form of this
// kernel fetched each candidate's distance with its own 2-byte load: 4 requests to L2 a step, none of them hitting L1, 14 K a unit,
// and the kernel sat at the rate the vector cache takes misses: profiles/r3_walk_first_pmc.csv.)
// Then the stitch, one lane per range and block (parseBlock in oracle/zso_encoder.c): reach = running maximum of the ranges' last match
// ends (a scan); a range drops the records that end at or below the reach before it and cuts the front of one that straddles; its first
// record left is JOINED to the match that defines the reach if it starts there with the same offset (long matches are found piecewise:
// the walkers cannot see each other); a second, backward scan gives every range's last record the end of what was joined to it.
// Last, the records that count are packed, 1 KiB of source (an OUTPUT RANGE = 1024 / R walk ranges) at a time, into the layout the
// entropy kernels read: seqAll[block][64 output ranges][256 records] + hdrAll (nseq, trailing, litSum, first = 0).
// ---------------------------------------------------------------------------------------------
#ifndef ZS_WALK_MASK
#define ZS_WALK_MASK 2             // lanes without a candidate, walkers at rest: 0 read along, 1 masked out of the LDS reads (branches), 2 read their own exchange slot (no branch, no conflict)
#endif
#ifndef ZS_WALK_MINW
#define ZS_WALK_MINW 1             // waves per SIMD the small-unit kernel is compiled for (register budget)
#endif
// aligned groups of 8 positions a step looks at (oracle: windowGroups): 2 for walk ranges of 256 bytes, 4 for ranges of 512 bytes (levels <= 2);
// a walker = LPW lanes that share them (LPW divides the groups); either way a walker per walk range
#define ZS_WALK_WGRP(WLOG) ((WLOG) >= 9 ? 4 : 2)
#ifndef ZS_WALK_LPW_256
#define ZS_WALK_LPW_256 2
#endif
#ifndef ZS_WALK_LPW_512
#define ZS_WALK_LPW_512 4
#endif
#define ZS_WALK_LPW(WLOG) ((WLOG) >= 9 ? ZS_WALK_LPW_512 : ZS_WALK_LPW_256)
#define ZS_WALK_THREADS(BIG, WLOG) ((((BIG) ? ZS_UNIT_MAX : ZS_BLOCK_MAX) >> (WLOG)) * ZS_WALK_LPW(WLOG))
#define ZS_WALK_KERNEL(LOOK, REPW, BIG, WLOG) k_lz_walk<ZS_WALK_LPW(WLOG), ZS_WALK_WGRP(WLOG), LOOK, REPW, BIG, ZS_WALK_THREADS(BIG, WLOG)>
// The unit's source in LDS is SKEWED: rows of 256 source bytes lie 256 + ZS_WALK_SKEW bytes apart, and the ZS_WALK_SKEW bytes behind a row
// repeat the head of the next row, so a read of up to 24 bytes that STARTS in a row is contiguous in that row's storage:
// LDS address of position p = SRC + p + ZS_WALK_SKEW * (p >> 8) (p signed: row -1 is the front pad).  Why: the walkers of a wavefront sit at
// nearly the same offset inside ranges that are 256 bytes apart - bank (a / 4) mod 32 made every "own side" read of a step a 4 - 16-way
// conflict (tools/lab/lds_sim.c prices the kernel's reads with the bank rules: 309 LDS cycles a wave step, measured 293; with the skew
// and idle lanes masked 171).  24 bytes = 6 banks a row: 16 walkers (32 lanes = a dword-read group) land on 16 different bank pairs.
#ifndef ZS_WALK_SKEW
#define ZS_WALK_SKEW  24u
#endif
#define ZS_WALK_FRONT 48u          // row -1 of the staged source: 24 zero bytes (positions -24 .. -1: backward reads near position 0) + the copy of row 0's head
#define ZS_WALK_TAIL  144u         // zero bytes behind the unit (forward reads near the end)
#define ZS_WALK_SRCBYTES(CAPB) ((((CAPB) + ZS_WALK_TAIL + ZS_WALK_SKEW * (((CAPB) + ZS_WALK_TAIL) / 256u + 1u)) + 15u) & ~15u)
// exchange buffer (16 bytes a lane) + front + skewed source + queue head + a byte a lane (BIG: bit 16 of the distances)
#define ZS_WALK_LDS(CAPB) ((CAPB) / 8 + 16 + ZS_WALK_FRONT + ZS_WALK_SRCBYTES(CAPB) + 16 + 1024)

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
// the 4 bytes at byte offset I of a span (I a constant)
template <int I, int K>
__device__ __forceinline__ uint32_t span_at(const uint32_t (&v)[K])
{
    if constexpr ((I & 3) == 0) return v[I >> 2];
    else return __builtin_amdgcn_alignbyte(v[(I >> 2) + 1], v[I >> 2], I & 3);
}
// bytes two 16-byte pieces agree on from their start (16: all)
__device__ __forceinline__ uint32_t zs_agree16(const uint32_t (&a)[4], const uint32_t (&b)[4])
{
    const uint64_t x0 = (uint64_t)(a[0] ^ b[0]) | ((uint64_t)(a[1] ^ b[1]) << 32), x1 = (uint64_t)(a[2] ^ b[2]) | ((uint64_t)(a[3] ^ b[3]) << 32);
    const uint32_t n0 = x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : 8u;
    const uint32_t n1 = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8u;
    return (n0 < 8u) ? n0 : 8u + n1;
}
// or / max / min over the LPW (2 or 4) adjacent lanes of a walker (data-parallel-primitive moves inside a quad: no LDS round trip)
template <int LPW> __device__ __forceinline__ uint32_t walker_or(uint32_t v)
{
    if constexpr (LPW >= 2) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    if constexpr (LPW == 4) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
    return v;
}
template <int LPW> __device__ __forceinline__ int walker_max(int v)
{
    if constexpr (LPW >= 2) v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));
    if constexpr (LPW == 4) v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
    return v;
}
template <int LPW> __device__ __forceinline__ uint32_t walker_min(uint32_t v)
{
    if constexpr (LPW >= 2) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    if constexpr (LPW == 4) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    return v;
}
// inclusive maximum over the threads 0 .. t of the workgroup (DIR = 1) or t .. NT - 1 (DIR = -1); wv: NT / 64 words of LDS.  Two barriers.
template <int NT, int DIR>
__device__ __forceinline__ uint32_t block_scan_max(uint32_t v, uint32_t *wv, uint32_t tid, uint32_t *total)
{
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)(DIR > 0 ? __shfl_up((int)v, d) : __shfl_down((int)v, d));
        if (DIR > 0 ? (int)lane >= d : (int)lane + d < 64) v = max(v, o);
    }
    __syncthreads();                                                     // (the words may still be read from the scan before)
    if (lane == (DIR > 0 ? 63u : 0u)) wv[wave] = v;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    #pragma unroll
    for (uint32_t k = 0; k < NT / 64; k++) { const uint32_t x = wv[k]; all = max(all, x); if (DIR > 0 ? k < wave : k > wave) pre = max(pre, x); }
    *total = all;
    return max(v, pre);
}
// 8 distances (16 bits each) -> bit i: distance i is not zero
__device__ __forceinline__ uint32_t zs_nonzero8(const uint4 d)
{
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 one = { 1, 1 };
    (void)one;
    auto nz = [&](uint32_t w) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(0x00010001u)); return r; };   // 1 per half that is not zero (asked for by name: the generic form became compares, selects and v_perm, ~25 instructions a group of 8)
    const uint32_t v = nz(d.x) | (nz(d.y) << 2) | (nz(d.z) << 4) | (nz(d.w) << 6);       // low halves at bits 0, 2, 4, 6; high halves at bits 16, 18, 20, 22
    return (v | (v >> 15)) & 0xFFu;
}

#ifdef ZS_WALK_PROFILE
// diagnostic build (tools/walk_profile.py): s_memtime stamps per wavefront and step part, every outstanding memory operation waited for at a stamp
// (so the parts do not overlap as they do in the product); the sums go behind all blocks' range results (the output stays valid)
#define WPROF_STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); wprof[k] += t_ - wlast; wlast = t_; }
#else
#define WPROF_STAMP(k)
#endif
template <int LPW, int WGRP, int LOOK, int REPWIN, bool BIG, int NT>
__global__ void __launch_bounds__(NT, BIG ? 1 : ZS_WALK_MINW)
k_lz_walk(const uint8_t *__restrict__ src, const ZsUnitDesc *__restrict__ units, uint32_t block0,
          const uint16_t *__restrict__ distAll, const uint8_t *__restrict__ distHiAll,
          uint2 *__restrict__ recAll, uint32_t junkSlot, uint4 *__restrict__ resAll, int rangeLogArg, const uint32_t *__restrict__ candCount)
{
    constexpr uint32_t CAP = BIG ? ZS_UNIT_MAX : ZS_BLOCK_MAX;                  // unit capacity in bytes
    constexpr uint32_t CPL = LOOK / LPW, RPL = REPWIN / LPW, GPL = WGRP / LPW;   // candidates / recent-offset positions / groups of distances per lane and step
    static_assert((LPW == 1 || LPW == 2 || LPW == 4) && WGRP % LPW == 0 && LOOK % LPW == 0 && REPWIN % LPW == 0 && CPL >= 1 && RPL >= 1 && LOOK <= 8 && REPWIN <= 8, "a walker's lanes share the groups and candidates evenly");
    static_assert(NT >= 64 && NT % 64 == 0 && NT * GPL * 16 <= CAP / 8, "the exchange buffer holds 16 bytes a lane and group");
    extern __shared__ __attribute__((aligned(16))) uint8_t walkLds[];
    // LDS: exchange buffer (the low addresses: the LDS-DMA's base register is not known to reach beyond 64 KiB), front pad, source, tail pad, results, queue, xhi
    constexpr uint32_t SRC = CAP / 8 + 16 + ZS_WALK_FRONT;                       // LDS address of source byte 0
    // LDS address of source position p (p may be a little negative: row -1 = the front pad)
    auto lpos = [](uint32_t p) { return SRC + p + (uint32_t)__mul24((int)p >> 8, (int)ZS_WALK_SKEW); };
    // 16 source bytes at position i (a multiple of 16) into the skewed copy, and into the row before's tail where they are a row's head
    auto stage16 = [&](uint32_t i, const uint4 v) {
        uint8_t *d = walkLds + lpos(i);
        *reinterpret_cast<uint2 *>(d) = make_uint2(v.x, v.y); *reinterpret_cast<uint2 *>(d + 8) = make_uint2(v.z, v.w);
        const uint32_t o = i & 255u;
        if (o == 0) { *reinterpret_cast<uint2 *>(d - ZS_WALK_SKEW) = make_uint2(v.x, v.y); *reinterpret_cast<uint2 *>(d - ZS_WALK_SKEW + 8) = make_uint2(v.z, v.w); }
        else if (o == 16) *reinterpret_cast<uint2 *>(d - ZS_WALK_SKEW) = make_uint2(v.x, v.y);
    };
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t rangeLog = (uint32_t)rangeLogArg, R = 1u << rangeLog;
    const ZsUnitDesc ud = units[blockIdx.x];
    const uint32_t slot = ud.firstBlock - block0;                                // scratch slot of the unit's first block
    const uint8_t *s = src + ud.srcOff;
    const uint32_t n = ud.size;
    const uint16_t *dist = distAll + (size_t)slot * ZS_BLOCK_MAX;
    const uint8_t *distHi = distHiAll + (size_t)slot * (ZS_BLOCK_MAX / 8);
    uint4 *xbuf = reinterpret_cast<uint4 *>(walkLds);                              // exchange buffer: wavefront w's region holds GPL x 64 slots of 8 distances: lane t's k-th group at [(64 w) GPL + 64 k + t]
    const uint32_t xw = (tid & ~63u) * GPL;                                      // my wavefront's first slot
    const uint32_t safeAddr = (xw + lane) * 16u;                                 // LDS address of my own first slot: where a lane with nothing to read reads (its own banks, no conflict)
    uint32_t *queue = reinterpret_cast<uint32_t *>(walkLds + SRC + ZS_WALK_SRCBYTES(CAP));
    uint4 *res = resAll + (size_t)slot * ZS_RES_PER_BLOCK;                       // per walk range (R >= 256: at most 256 a block): records, last match end in its block, last offset
    uint8_t *xhi = reinterpret_cast<uint8_t *>(queue + 4);                        // BIG: lane t's byte of bit 16 of its 8 distances
    uint2 *recs = recAll + (size_t)slot * (ZS_BLOCK_MAX / 4);                     // range at unit position p: slots from p / 4 (its matches start inside it, >= 4 bytes each)
    const uint32_t sub = lane & (LPW - 1u);
#ifdef ZS_WALK_PROFILE
    unsigned long long wprof[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, wlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wlast) :: "memory");
#endif

    // a MATCHLESS unit (fewer than n >> ZS_MATCHLESS_SHIFT candidate positions: incompressible input's chance candidates) is not walked: its ranges
    // report no records, its blocks go on without sequences (findCandidates / parseBlock in oracle/zso_encoder.c)
    if (candCount[2 * slot] + candCount[2 * slot + 1] < (n >> ZS_MATCHLESS_SHIFT)) {
        const uint32_t nRanges = (n + R - 1) >> rangeLog, perBlockLog = 16u - rangeLog;
        for (uint32_t r = tid; r < nRanges; r += NT) res[(r >> perBlockLog) * ZS_RES_PER_BLOCK + (r & ((1u << perBlockLog) - 1u))] = make_uint4(0, 0, 0, 0);
        return;
    }
    // ---- stage the unit ----
    uint32_t mixed = 0;                                                          // some byte of the unit differs from its first byte
    if (tid < (ZS_WALK_FRONT - ZS_WALK_SKEW) / 4) reinterpret_cast<uint32_t *>(walkLds + SRC - ZS_WALK_FRONT)[tid] = 0;
    if (tid == 0) { queue[0] = 0; queue[1] = 0; }
    __syncthreads();
    {
        // whole 16-byte pieces: the loads of a thread are issued together (a load behind a branch, followed by its LDS
        // store, would wait out one memory round trip per piece); then the partial piece and the zero tail
        constexpr uint32_t PER = (CAP / 16 + NT - 1) / NT;                       // pieces per thread
        const uint32_t nFull = n & ~15u;
        uint4 v[PER];
        const uint32_t splat = (n ? (uint32_t)s[0] : 0u) * 0x01010101u;
        #pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = (tid + k * NT) * 16;
            const uint32_t ii = (i + 16 <= nFull) ? i : 0u;                      // clamped: always a valid address when nFull >= 16
            v[k] = make_uint4(0, 0, 0, 0);
            if (nFull >= 16) __builtin_memcpy(&v[k], s + ii, 16);
        }
        #pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = (tid + k * NT) * 16;
            if (i + 16 <= nFull) stage16(i, v[k]);
            // (clamped pieces hold piece 0: comparing them too is harmless)
            mixed |= (v[k].x ^ splat) | (v[k].y ^ splat) | (v[k].z ^ splat) | (v[k].w ^ splat);
        }
        if (n != nFull || n < 16) mixed = 1;                                     // (a ragged unit is walked as any other: the check is a short cut, not a decision)
        for (uint32_t i = nFull + tid * 16; i < n + ZS_WALK_TAIL; i += NT * 16) {
            uint4 w = make_uint4(0, 0, 0, 0);
            if (i < n) {
                uint64_t lo = 0, hi = 0;
                for (uint32_t k = 0; k < 16 && i + k < n; k++) { const uint64_t c = s[i + k]; if (k < 8) lo |= c << (8 * k); else hi |= c << (8 * (k - 8)); }
                w = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
            }
            if (i + 16 <= CAP + ZS_WALK_TAIL) stage16(i, w);
        }
    }
#if defined(ZS_WALK_STOP) && ZS_WALK_STOP == 1
    if (n) return;                                                               // timing aid: staging only
#endif
    const uint32_t hashable = (n >= 8) ? n - 7 : 0;
    // A unit of one repeated byte becomes RLE blocks whatever the parse says (the literals kernel decides that from the bytes, as
    // zso_encoder.c:833-836 does before parsing): its walk -- the slowest there is, every match running to the crossing limit -- is skipped.
    if (mixed) queue[1] = 1u;                                                    // (no __syncthreads_or: hipcc gives it static LDS of its own, which moves the dynamic part this kernel addresses by hand)
    __syncthreads();
    if (queue[1] == 0) {
        const uint32_t nRanges = (n + R - 1) >> rangeLog, perBlockLog = 16u - rangeLog;
        for (uint32_t r = tid; r < nRanges; r += NT) res[(r >> perBlockLog) * ZS_RES_PER_BLOCK + (r & ((1u << perBlockLog) - 1u))] = make_uint4(0, 0, 0, 0);
        return;
    }
    WPROF_STAMP(0)

    // ---- the walk ----
    {
        const uint32_t nRanges = (n + R - 1) >> rangeLog;
        bool active = false, more = true;                                        // walking a range / the queue may hold more
        uint32_t r = 0, ip = 0, anchor = 0, scanEnd = 0, limit = 0, rep0 = 0, rep1 = 0, nseq = 0, lastOff = 0, recBase = 0, blockBase = 0;
        uint2 heldRec = make_uint2(0, 0);                                        // the odd record waiting for its pair
        const uint32_t perBlockLog = 16u - rangeLog;                             // walk ranges per block
        auto resIndex = [&](uint32_t rr) { return (rr >> perBlockLog) * ZS_RES_PER_BLOCK + (rr & ((1u << perBlockLog) - 1u)); };
        // distances of the group (ip >> 3) + sub, requested as soon as ip is known: 16 bytes a lane straight into the lane's slot of the
        // exchange buffer (LDS-DMA: no registers carried around the loop, no wait the compiler places for me; behind the hashable positions
        // whatever the scratch holds: those bits are cut off the window).  BIG: bit 16 of the distances, a byte per group, by an ordinary load.
        // (The load is an asm statement: hipcc counts a __builtin_amdgcn_global_load_lds and then waits vmcnt(0) in front of every LDS access it
        // cannot tell apart from the destination; this one is waited for by hand: "s_waitcnt vmcnt(0)" at the top of the step and where a
        // walker takes a new range.  M0 = LDS destination of lane 0, set inside the statement that uses it.)
        typedef __attribute__((address_space(3))) uint4 *ZsLdsU4;
        const uint32_t xwaveLds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(ZsLdsU4)(xbuf + xw));
        uint32_t gh[GPL] = {};
        auto loadGroup = [&](uint32_t p) {
            #pragma unroll
            for (uint32_t k = 0; k < GPL; k++) {                                 // the lane's groups: sub, sub + LPW, ..
                const uint32_t g = (p >> 3) + sub + LPW * k;
                const uint16_t *gsrc = dist + (size_t)g * 8u;
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(xwaveLds + k * 1024u) : "memory");
                if (BIG) gh[k] = (uint32_t)distHi[g];
            }
        };
        for (;;) {
            if (__any(!active && more)) {
                // idle walkers take the next ranges of the queue: one LDS atomic per wavefront
                const uint64_t idle = __ballot(!active && more && sub == 0);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(queue, (uint32_t)__popcll(idle));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (!active && more) {
                    const uint32_t leader = lane & ~(LPW - 1u);
                    r = base + (uint32_t)__popcll(idle & ((1ull << leader) - 1ull));
                    if (r >= nRanges) more = false;
                    else {
                        const uint32_t start = r << rangeLog;
                        const uint32_t blockStart = start & ~(ZS_BLOCK_MAX - 1);     // the range's block inside the unit
                        const uint32_t blockEnd = blockStart + min(n - blockStart, ZS_BLOCK_MAX);
                        const uint32_t end = min(start + R, blockEnd);
                        limit = min(end + ZS_CROSS_MAX, blockEnd);               // matches end at or before this
                        scanEnd = (blockEnd - blockStart >= 16) ? min(end, hashable) : 0;
                        ip = start; anchor = start; rep0 = 0; rep1 = 0; nseq = 0; lastOff = 0; recBase = start >> 2; blockBase = blockStart;
                        active = ip < scanEnd;
                        if (!active && sub == 0) res[resIndex(r)] = make_uint4(0, 0, 0, 0);    // nothing to scan
                        if (active) { loadGroup(ip); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }    // (its first group: nothing to overlap it with)
                    }
                }
            }
            if (!__any(active)) { if (!__any(more)) break; continue; }
            WPROF_STAMP(1)
#ifdef ZS_WALK_PROFILE
            wprof[8] += 1; wprof[9] += (unsigned long long)__popcll(__ballot(active));
#endif
            const uint32_t wend = min((ip & ~7u) + 8u * WGRP, scanEnd);           // the walker's groups
            // ---- recent offsets: the walker's lanes try RPL positions each ----
            uint32_t rm0 = 0, rm1 = 0;                                           // bit i: rep0 / rep1 repeats 4 bytes at ip + i
            {
                const bool t0 = active && rep0 != 0 && ip >= rep0, t1 = active && rep1 != 0 && ip >= rep1;
                const uint32_t p0 = ip + sub * RPL;
                constexpr int SP = (RPL + 3 + 3) / 4;                            // dwords that hold RPL + 3 bytes
                uint32_t a[SP] = {}, b[SP] = {}, c[SP] = {};
                // (lanes with nothing to try stay out of the source's banks - an LDS instruction costs what its busiest bank takes, and walkers at rest
                // all sit at a range end, the same bank - : ZS_WALK_MASK 1 by the exec mask (branches around the reads), 2 by reading their own
                // slot of the exchange buffer instead (no branch: the reads of a step stay one batch))
                if (ZS_WALK_MASK != 1 || t0 || t1) {
                    const bool tr = ZS_WALK_MASK != 2 || t0 || t1;
                    lds_span<SP>(tr ? lpos(p0) : safeAddr, a);
                    lds_span<SP>(tr ? lpos(t0 ? p0 - rep0 : p0) : safeAddr, b);
                    lds_span<SP>(tr ? lpos(t1 ? p0 - rep1 : p0) : safeAddr, c);
                }
                auto tryAt = [&](auto iTag) {
                    constexpr int I = decltype(iTag)::value;
                    const uint32_t q = p0 + I;
                    const bool ok = q < wend && q + 4 <= limit;
                    const uint32_t av = span_at<I>(a);
                    if (t0 && ok && av == span_at<I>(b)) rm0 |= 1u << (sub * RPL + I);
                    if (t1 && ok && av == span_at<I>(c)) rm1 |= 1u << (sub * RPL + I);
                };
                tryAt(std::integral_constant<int, 0>{});
                if constexpr (RPL >= 2) tryAt(std::integral_constant<int, 1>{});
                if constexpr (RPL >= 4) { tryAt(std::integral_constant<int, 2>{}); tryAt(std::integral_constant<int, 3>{}); }
                if constexpr (RPL >= 8) { tryAt(std::integral_constant<int, 4>{}); tryAt(std::integral_constant<int, 5>{}); tryAt(std::integral_constant<int, 6>{}); tryAt(std::integral_constant<int, 7>{}); }
                rm0 = walker_or<LPW>(rm0); rm1 = walker_or<LPW>(rm1);
                rm1 &= ~rm0;                                                     // rep0 is tried first
            }
            WPROF_STAMP(2)
            // ---- the window: the walker's four groups of distances are in the exchange buffer (requested a step ago); my group's candidate bits, the four groups' side by side ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t m = 0;
            #pragma unroll
            for (uint32_t k = 0; k < GPL; k++) {
                const uint4 gd = xbuf[xw + 64u * k + lane];
                if (BIG) xhi[xw + 64u * k + lane] = (uint8_t)gh[k];
                m |= (zs_nonzero8(gd) | gh[k]) << (8u * (sub + LPW * k));
            }
            m = walker_or<LPW>(m) >> (ip & 7u);                                  // bit i: position ip + i has a candidate
            if (active) m = (m & (0xFFFFFFFFu >> (32u - (wend - ip)))) | rm0 | rm1; else m = 0;   // 1 <= wend - ip <= 8 LPW while active
            {                                                                    // lane sub takes the candidates sub * CPL ..: clear the lower ones
                const uint32_t skip = sub * CPL;
                if constexpr ((LPW - 1) * CPL >= 4) { uint32_t t = m & (m - 1); t &= t - 1; t &= t - 1; t &= t - 1; m = (skip & 4u) ? t : m; }
                if constexpr ((((LPW - 1) * CPL) & 2) != 0 || (LPW == 4 && CPL == 2)) { uint32_t t = m & (m - 1); t &= t - 1; m = (skip & 2u) ? t : m; }
                if constexpr ((CPL & 1) != 0) { const uint32_t t = m & (m - 1); m = (skip & 1u) ? t : m; }
            }
            uint32_t idx[CPL], off[CPL], dv[CPL]; bool have[CPL], isRep[CPL];
            #pragma unroll
            for (uint32_t c = 0; c < CPL; c++) {
                have[c] = m != 0;
                idx[c] = have[c] ? (uint32_t)__builtin_ctz(m) : 0u;
                m &= m - 1;
                const uint32_t q = ip + idx[c];
                // the distance: group (q >> 3) - (ip >> 3) of the walker, that lane's slot of the exchange buffer (same wavefront: in order behind the stores above)
                const uint32_t j = (q >> 3) - (ip >> 3);                         // its group: lane j % LPW of the walker, that lane's group j / LPW
                const uint32_t owner = xw + 64u * (j / LPW) + (lane & ~(LPW - 1u)) + (j % LPW);
                dv[c] = reinterpret_cast<const uint16_t *>(xbuf + owner)[q & 7u];
                if (BIG) dv[c] |= (((uint32_t)xhi[owner] >> (q & 7u)) & 1u) << 16;
            }
            #pragma unroll
            for (uint32_t c = 0; c < CPL; c++) asm volatile("" : "+v"(dv[c]));      // every candidate's distance is read before the first is used: one LDS round trip, not one a candidate
            #pragma unroll
            for (uint32_t c = 0; c < CPL; c++) {
                const bool r0 = ((rm0 >> idx[c]) & 1u) != 0, r1 = ((rm1 >> idx[c]) & 1u) != 0;
                isRep[c] = have[c] && (r0 || r1);
                off[c] = r0 ? rep0 : (r1 ? rep1 : dv[c]);
                if (!have[c]) off[c] = 0;
            }
            WPROF_STAMP(3)
            // ---- score: 8 bytes forward, 4 bytes backward, from LDS ----
            int bestKey = 0; uint32_t bestPack = 0, bestOff = 0;
            uint32_t sa[CPL][3], sb[CPL][3];                                     // bytes [q - 4, q + 8) of both sides, every candidate's requested before the first is looked at
            #pragma unroll
            for (uint32_t c = 0; c < CPL; c++) {
                const uint32_t q = ip + idx[c];
                #pragma unroll
                for (int k = 0; k < 3; k++) { sa[c][k] = 0; sb[c][k] = 0; }
                if (ZS_WALK_MASK != 1 || have[c]) {
                    const bool tr = ZS_WALK_MASK != 2 || have[c];
                    lds_span<3>(tr ? lpos(q - 4) : safeAddr, sa[c]);
                    lds_span<3>(tr ? lpos(q - off[c] - 4) : safeAddr, sb[c]);
                }
            }
            if (ZS_WALK_MASK != 1) {
                #pragma unroll
                for (uint32_t c = 0; c < CPL; c++) asm volatile("" : "+v"(sa[c][0]), "+v"(sa[c][1]), "+v"(sa[c][2]), "+v"(sb[c][0]), "+v"(sb[c][1]), "+v"(sb[c][2]));
            }
            #pragma unroll
            for (uint32_t c = 0; c < CPL; c++) {
                const uint32_t q = ip + idx[c];
                const uint32_t (&a)[3] = sa[c], (&b)[3] = sb[c];
                const uint32_t xb = a[0] ^ b[0], x0 = a[1] ^ b[1], x1 = a[2] ^ b[2];
                const uint32_t f0 = x0 ? (uint32_t)__builtin_ctz(x0) : 32u, f1 = x1 ? (uint32_t)__builtin_ctz(x1) : 32u;
                const uint32_t fwd = min(((x0 ? f0 : 32u + f1)) >> 3, limit - q);
                const uint32_t maxBack = min(min(q - anchor, q - off[c]), ZS_BCAP);
                const uint32_t back = min(xb ? ((uint32_t)__builtin_clz(xb) >> 3) : 4u, maxBack);
                if (have[c] && fwd >= (isRep[c] ? ZS_REPMIN : ZS_MINMATCH)) {
                    const int gain = (int)(4u * fwd + 8u * back) - (isRep[c] ? 0 : (int)zs_highbit(off[c] + 1)) - 5 * (int)idx[c];
                    const int key = ((gain + 2048) << 3) | (int)(7u - (sub * CPL + c));
                    if (key > bestKey) { bestKey = key; bestPack = idx[c] | (fwd << 8) | (back << 16); bestOff = off[c]; }
                }
            }
            const int best = walker_max<LPW>(bestKey);
            const bool took = active && best != 0;
            const uint32_t packed = walker_or<LPW>(bestKey == best ? bestPack : 0u), boff = walker_or<LPW>(bestKey == best ? bestOff : 0u);
            const uint32_t bq = ip + (packed & 0xFFu), bback = packed >> 16;
            uint32_t bfwd = (packed >> 8) & 0xFFu;
            WPROF_STAMP(4)
            // ---- the match's whole length: the walker's lanes compare on, 16 bytes a lane and round ----
            const bool extend = took && bfwd == ZS_FCAP && (limit - bq) > ZS_FCAP;
            bool need = extend;
            uint32_t pos = bq + ZS_FCAP;
            while (__any(need)) {
                uint32_t e = 0xFFFFFFFFu;                                        // bytes gained if the match ends in this lane's piece
                if (need) {
                    const uint32_t cap = limit - pos, fo = 16u * sub;            // cap >= 1 while need
                    uint32_t nb = 0;
                    if (fo < cap) {
                        uint32_t a[4], b[4];
                        lds_span<4>(lpos(pos + fo), a);
                        lds_span<4>(lpos(pos - boff + fo), b);
                        nb = zs_agree16(a, b);
                    }
                    if (nb < 16u || fo + 16u >= cap) e = min(fo + nb, cap);
                }
                e = walker_min<LPW>(e);
                if (need) {
                    if (e != 0xFFFFFFFFu) { pos += e; need = false; }
                    else pos += 16u * LPW;
                }
            }
            if (extend) bfwd = pos - bq;
            WPROF_STAMP(5)
            uint32_t recLl = 0;
            if (took) {
                recLl = bq - bback - anchor;                                     // literals in front of the match, inside the range (< R <= 512)
                nseq++; lastOff = boff;
                ip = bq + bfwd; anchor = ip;
                if (boff == rep1) { rep1 = rep0; rep0 = boff; }
                else if (boff != rep0) { rep1 = rep0; rep0 = boff; }
            } else if (active) ip = wend;
            // the record leaves (the walker's first lane, a step that took a match); the next step's distances are requested (walkers with a
            // step to come): the memory system takes a wavefront's small requests one by one, and the kernel's time follows their number
            // (every lane loading and storing every step, walkers at rest included: 4 requests a walker step, 0.95 ms; profiles/r3_walk_*)
            // (records leave in pairs, 16 aligned bytes: half the write requests; a range's odd last one alone)
            if (took) {
                const uint2 rec = make_uint2((bq - bback - blockBase) | ((bback + bfwd) << 17), boff | (recLl << 17));    // start: position in its block; offset, literals
                if (nseq & 1u) heldRec = rec;
                else if (sub == 0) *reinterpret_cast<uint4 *>(recs + recBase + nseq - 2) = make_uint4(heldRec.x, heldRec.y, rec.x, rec.y);
            }
            if (active && ip >= scanEnd) {
                if (sub == 0) {
                    if (nseq & 1u) recs[recBase + nseq - 1] = heldRec;
                    res[resIndex(r)] = make_uint4(nseq, nseq ? anchor - blockBase : 0u, lastOff, 0u);
                }
                active = false;
            }
            if (active) loadGroup(ip);
            WPROF_STAMP(6)
        }
    }
    WPROF_STAMP(1)
#ifdef ZS_WALK_PROFILE
    __syncthreads();
    if (lane == 0) { unsigned long long *o = reinterpret_cast<unsigned long long *>(resAll + (size_t)(junkSlot / 64)) + ((size_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 10;   /* behind all blocks' results */ for (int k = 0; k < 10; k++) o[k] = wprof[k]; }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_lz_stitch : one workgroup of 256 threads per block, thread t = walk range t of the block (parseBlock in oracle/zso_encoder.c):
// the stitch of the ranges' records and their packing into the entropy kernels' layout (see k_lz_walk).  A kernel of its own: these
// are chains of dependent loads with little arithmetic, which many small workgroups per CU overlap, while inside k_lz_walk they held
// 64 KiB of LDS and 8 wavefronts for a sixth of that kernel's time.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_lz_stitch(const ZsBlockDesc *__restrict__ blocks, const uint2 *__restrict__ recAll, const uint4 *__restrict__ resAll,
            ZsSeqRec *__restrict__ seqAll, ZsRangeHdr *__restrict__ hdrAll, int rangeLogArg)
{
    constexpr int NT = 256;
    __shared__ uint32_t sa[9 * 256 + 16];
    uint32_t *sKept = sa, *sCnt = sa + 256, *sTrail = sa + 512, *sFOut = sa + 768, *sNs = sa + 1024, *sOwn = sa + 1280, *sEs = sa + 1536, *sPend = sa + 1792, *sChain = sa + 2048, *sWave = sa + 2304;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t rangeLog = (uint32_t)rangeLogArg, R = 1u << rangeLog;
    const uint32_t slot = blockIdx.x;
    const uint32_t NRB = ZS_BLOCK_MAX >> rangeLog;                               // walk ranges per block
    const uint32_t gLog = ZS_OUT_LOG - rangeLog, G = 1u << gLog;                 // walk ranges per output range
    const uint2 *recs = recAll + (size_t)slot * (ZS_BLOCK_MAX / 4);
    const uint4 *res = resAll + (size_t)slot * ZS_RES_PER_BLOCK;
    {
        const uint32_t bStart = 0, bi = 0;
        const uint32_t bN = blocks[slot].size;
        const uint32_t bEnd = bN;
        const uint32_t t = tid;
        const bool mine = t < NRB;
        const bool walked = mine && bN >= 16 && (t << rangeLog) < bN;            // (ranges of a block too small to compress, or behind its end, leave no result)
        const uint4 rr = walked ? res[t] : make_uint4(0, 0, 0, 0);
        const uint32_t ns = rr.x, le = rr.y;
        const uint32_t rs = min(bStart + (t << rangeLog), bEnd), re = min(bStart + (t << rangeLog) + R, bEnd);   // the range, cut at the block end
        const uint2 *rp = recs + ((bStart + (t << rangeLog)) >> 2);
        // reach before me / with me: maximum of the ranges' last match ends; the low bits name the range that holds it (the earliest of equals)
        uint32_t total;
        const uint32_t key = ns ? ((le << 9) | (511u - t)) : 0u;
        const uint32_t incl = block_scan_max<NT, 1>(key, sWave, tid, &total);
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) { excl = 0; for (uint32_t k = 0; k < (tid >> 6); k++) excl = max(excl, sWave[k]); }
        const uint32_t own = max(excl >> 9, bStart), reach = max(incl >> 9, bStart), reachAll = max(total >> 9, bStart);
        const uint32_t definer = 511u - (excl & 511u);                           // valid if excl != 0
#if defined(ZS_STITCH_STOP) && ZS_STITCH_STOP == 2
        if (bN) return;                                                          // timing aid: the first scan only
#endif
        // first record that counts: records end in ascending order
        uint32_t f = 0, fStart = 0, fEnd = 0, fOff = 0;
        if (ns && own >= rs) {
            if (le <= own) f = ns;
            else {
                // the first record that ends above own.  The match that crosses into a range mostly covers none or one of its records: the
                // first two come with one 16-byte load (the slots are 16-byte aligned), a search over the rest only if both end at or below own
                const uint4 r01 = *reinterpret_cast<const uint4 *>(rp);          // (a second record that is not there reads as whatever: only looked at if ns > 1)
                auto endOf = [](uint32_t x) { return (x & 0x1FFFFu) + (x >> 17); };
                uint2 rec = make_uint2(r01.x, r01.y), nxt = make_uint2(r01.z, r01.w);
                bool haveNxt = ns > 1;
                if (endOf(r01.x) > own) f = 0;
                else if (ns > 1 && endOf(r01.z) > own) { f = 1; rec = nxt; haveNxt = false; }
                else {
                    uint32_t lo = 2, hi = ns - 1;                                // (le > own: the last record ends above own, ns > 2 here)
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (endOf(rp[mid].x) > own) hi = mid; else lo = mid + 1; }
                    f = lo; rec = rp[f]; haveNxt = false;
                }
                uint32_t st = rec.x & 0x1FFFFu, en = endOf(rec.x);
                if (st < own && en - own < ZS_MINMATCH) {                        // too little left of a straddling one (the next starts above own)
                    f++;
                    if (f < ns) { rec = haveNxt ? nxt : rp[f]; st = rec.x & 0x1FFFFu; en = endOf(rec.x); }
                }
                fStart = max(st, own); fEnd = en; fOff = rec.y & 0x1FFFFu;
            }
        }
        const bool keptAny = f < ns;
        if (mine) sKept[t] = keptAny;
        __syncthreads();
        // joined to the match that defines the reach: starts there, same offset, and that match (its range's last record) counts itself
        const bool joined = keptAny && excl != 0 && own >= rs && fStart == own && fOff == res[definer].z && sKept[definer] != 0;
        const uint32_t fOut = f + (joined ? 1u : 0u), cnt = ns - fOut;
        const uint32_t es = joined ? fEnd : max(rs, own), te = max(re, reach), lastEnd = cnt ? le : es;
        // what was joined to my last record: the nearest range behind me that ends a chain says where (one that only passes a chain on does not)
        const bool stopper = ns && le > own && !(joined && cnt == 0);
        uint32_t total2;
        const uint32_t key2 = stopper ? (((512u - t) << 18) | (joined ? fEnd : own)) : 0u;
        const uint32_t sfx = block_scan_max<NT, -1>(key2, sWave, tid, &total2);
        uint32_t sfxEx = (uint32_t)__shfl_down((int)sfx, 1);
        if (lane == 63) { sfxEx = 0; for (uint32_t k = (tid >> 6) + 1; k < NT / 64; k++) sfxEx = max(sfxEx, sWave[k]); }
        const uint32_t chainEnd = sfxEx ? (sfxEx & 0x3FFFFu) : reachAll;
        if (mine) { sCnt[t] = cnt; sTrail[t] = te - lastEnd; sFOut[t] = fOut; sNs[t] = ns; sOwn[t] = own; sEs[t] = es; sChain[t] = chainEnd; }
        __syncthreads();
        const uint32_t gi = t & (G - 1u);
        if (mine) {                                                              // literals left over in front of my territory, inside my output range
            uint32_t pend = 0;
            for (uint32_t i = gi; i > 0; i--) { pend += sTrail[t - gi + i - 1]; if (sCnt[t - gi + i - 1]) break; }
            sPend[t] = pend;
        }
        __syncthreads();
#if defined(ZS_STITCH_STOP) && ZS_STITCH_STOP == 1
        if (bN) return;                                                          // timing aid: no packing
#endif
        // pack, a lane a record: wavefront w takes the output ranges 16 w .. 16 w + 15 in turn; lane d of a round the d-th record of the output
        // range = record fOut + (d - records of the walk ranges before) of its walk range: loads of consecutive lanes run along a walk range's
        // slots, stores along the output range's.  (A thread packing its own range's records one by one - 8 scattered bytes a lane and
        // store - took 0.26 ms of this kernel's 0.32: the stores queued at issue.)
        const uint32_t wave = tid >> 6;
        for (uint32_t go = 0; go < ZS_WALK_RANGES / (NT / 64); go++) {
            const uint32_t g = wave * (ZS_WALK_RANGES / (NT / 64)) + go, j0 = g << gLog;
            uint32_t c[4] = { 0, 0, 0, 0 }, total = 0;
            for (uint32_t i = 0; i < G; i++) { c[i] = sCnt[j0 + i]; total += c[i]; }
            ZsSeqRec *out = seqAll + ((size_t)slot * ZS_WALK_RANGES + g) * ZS_SEQ_PER_RANGE;
            uint32_t lits = 0;
            for (uint32_t d0 = 0; d0 < total; d0 += 64) {
                const uint32_t d = d0 + lane;
                if (d < total) {
                    uint32_t j = 0, pre = 0;
                    if (d >= c[0]) { j = 1; pre = c[0]; if (G > 2) { if (d >= pre + c[1]) { j = 2; pre += c[1]; if (d >= pre + c[2]) { j = 3; pre += c[2]; } } } }
                    const uint32_t jr = j0 + j, k = sFOut[jr] + (d - pre);
                    const uint2 rec = recs[(jr << (rangeLog - 2)) + k];
                    uint32_t st = rec.x & 0x1FFFFu, ml = rec.x >> 17, ll = rec.y >> 17;
                    const uint32_t off = rec.y & 0x1FFFFu, en = st + ml;
                    if (d == pre) {                                              // the walk range's first record that counts: a straddling one is cut; its literals start at the territory
                        const uint32_t ownj = sOwn[jr];
                        if (st < ownj) { st = ownj; ml = en - st; }
                        ll = st - sEs[jr] + sPend[jr];
                    }
                    if (k + 1 == sNs[jr]) ml = sChain[jr] - st;                  // its last: with what was joined to it
                    ZsSeqRec w; w.x = zs_rec_x(ll, ml, off); w.y = zs_rec_y(off, st);
                    out[d] = w;
                    lits += ll;
                }
            }
            #pragma unroll
            for (int o = 32; o >= 1; o >>= 1) lits += (uint32_t)__shfl_xor((int)lits, o);
            if (lane == 0) {
                ZsRangeHdr h; h.nseq = total; h.litSum = lits; h.trailing = 0; h.first = 0;
                for (uint32_t i = G; i > 0; i--) { h.trailing += sTrail[j0 + i - 1]; if (sCnt[j0 + i - 1]) break; }
                hdrAll[(size_t)slot * ZS_WALK_RANGES + g] = h;
            }
        }
    }
}
