// Batch decoder: k_decode_frames, one wavefront per input item (one or more concatenated frames).
// Follows the reference decoder function by function (csharp/src/ZStdDecompress.cs and friends);
// the C restatement used as the checker is oracle/zso_decoder.c.  Same error codes
// (csharp/src/ZStdErrors.cs:61-90).
//
//   frame / block layer   DecompressMultiFrame :2096-2160, DecompressFrame :2008-2091, GetcBlockSize :646-659
//   frame header          ZSTD_getFrameHeader_advanced :421-499
//   literals              DecodeLiteralsBlock :683-821, HufDecompress.cs:117-358, EntropyCommon.cs:198-269
//   sequence tables       DecodeSeqHeaders :1110-1180, BuildFSETable :958-1034, ReadNCount EntropyCommon.cs:79-188
//   sequences             DecodeSequence :1473-1553, ExecSequence :1265-1352, body :1555-1608
//   checksum              XxHash.cs (XXH64 seed 0, low 32 bits) :2076-2085
//
// Work split inside the wavefront: headers and FSE/Huffman table parsing on lane 0 (small, serial),
// Huffman streams on lanes 0..3 (one stream each), sequence decoding on lane 0 in tiles of 64,
// literal and match copies by all 64 lanes.
#include "zsmi_device.h"

// -DZS_DEC_PROFILE: cycles per phase of each item, left in the 64 spare bytes behind its literal scratch
// (0 literals incl. Huffman table, 1 sequence tables, 2 sequence decoding, 3 sequence execution, 4 checksum, 5 whole item)
#ifdef ZS_DEC_PROFILE
#define PROF_T0() uint64_t prof_t_ = __builtin_readcyclecounter()
#define PROF_ADD(k) do { const uint64_t now_ = __builtin_readcyclecounter(); if (g_prof) g_prof[k] += now_ - prof_t_; prof_t_ = now_; } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ADD(k) do {} while (0)
#endif
// Synchronisation inside one item is wavefront-local.  wave_sync() orders LDS traffic between the lanes (LDS instructions of
// a wavefront execute in issue order).  Bytes handed from lane to lane through GLOBAL memory additionally need the stores to
// have completed before the loads are issued (loads and stores of a wavefront may complete out of order with respect to
// each other): wave_mem_sync() waits for the outstanding vector-memory operations of the wavefront.
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
#ifdef ZS_DEC_ERRLINE                     // debugging aid: an error result carries the source line that raised it
#define ZE(code) (0xFF000000u | (uint32_t)__LINE__)
#else
#define ZE(code) (0u - (uint32_t)(code))
#endif
#define E_GENERIC 1
#define E_prefix_unknown 10
#define E_frameParameter_unsupported 14
#define E_frameParameter_windowTooLarge 16
#define E_corruption_detected 20
#define E_checksum_wrong 22
#define E_dictionary_corrupted 30
#define E_dictionary_wrong 32
#define E_tableLog_tooLarge 44
#define E_dstSize_tooSmall 70
#define E_srcSize_wrong 72
#ifdef ZS_DEC_ERRLINE
__device__ __forceinline__ bool isErr(uint32_t v) { return v >= 0xFF000000u; }
#else
__device__ __forceinline__ bool isErr(uint32_t v) { return v > ZE(120); }
#endif

struct ZsDecItem { uint64_t srcOff; uint64_t dstOff; uint32_t srcSize; uint32_t dstCap; };

__constant__ uint8_t d_LL_bits[36] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
__constant__ uint8_t d_ML_bits[53] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
                                      1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
__constant__ uint32_t d_LL_base[36] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,18,20,22,24,28,32,40,
                                       48,64,0x80,0x100,0x200,0x400,0x800,0x1000, 0x2000,0x4000,0x8000,0x10000 };
__constant__ uint32_t d_ML_base[53] = { 3,4,5,6,7,8,9,10, 11,12,13,14,15,16,17,18, 19,20,21,22,23,24,25,26,
                                       27,28,29,30,31,32,33,34, 35,37,39,41,43,47,51,59, 67,83,99,0x83,0x103,0x203,0x403,0x803,
                                       0x1003,0x2003,0x4003,0x8003,0x10003 };
__constant__ int16_t d_LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
__constant__ int16_t d_ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                             1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
__constant__ int16_t d_OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };

// one cell of a sequence decoding table (ZStdDecompress.cs:132-146).  The reference keeps baseValue and nbAdditionalBits in
// the cell; here the cell keeps the symbol (4 bytes instead of 8: the three tables take 5 KiB of LDS, not 10) and the
// decoder takes base / extra bits of LL and ML codes from two small LDS tables, those of an offset code from the code itself.
struct SeqSym { uint16_t nextState; uint8_t nbBits; uint8_t sym; };
struct SeqTab { uint32_t tableLog; SeqSym cells[512]; };
#define ZS_LITWIN 512u      // bytes of each Huffman stream staged in LDS at a time
#define ZS_SEQWIN 1024u     // bytes of the sequence bitstream staged in LDS at a time (a tile of 64 sequences reads < 720)
struct DLds {
    uint32_t hufLog;
    SeqTab LL;
    union {                             // one phase at a time
        struct { uint8_t weights[256]; uint16_t symStart[256]; uint32_t rank[16]; int16_t norm[64]; uint16_t symbolNext[64];
                 struct { uint16_t newState; uint8_t symbol; uint8_t nbBits; } wfse[64];   // weight FSE table (tableLog <= 6)
                 uint32_t hdrWin[66];                                                        // the header being parsed: 256 bytes + zero pad
                 uint32_t symMask[128];                                                      // table build: a 64-bit lane mask per symbol
               } tb;                                                                        // a table is being parsed / built
        uint32_t litWin[4][(ZS_LITWIN + 8) / 4 + 2];                                        // the Huffman streams run
        struct { uint32_t tileLL[64], tileML[64], tileOff[64]; uint32_t win[(ZS_SEQWIN + 8) / 4 + 2]; } sq;   // sequences run
    } u;
    uint32_t misc[16];
    uint32_t llTab[36], mlTab[53];      // base | extra bits << 24 of each LL / ML code (LL_base, LL_bits, ML_base, ML_bits)
    uint32_t pad16[1];
#ifdef ZS_PREP_PROFILE
    unsigned long long pp[12], ppMark;  // development aid (tools/prep_profile.py): s_memtime ticks per phase of k_dec_prep, this wavefront
#endif
    // k_dec_prep allocates the struct up to here (ZS_DLDS_PREP bytes): it builds its three sequence tables one after the other
    // in LL and its Huffman table in global memory, and what it holds per item decides how many items a CU prepares at once
    SeqTab ML;
    struct { uint32_t tableLog; SeqSym cells[256]; } OF;
    uint16_t huf[4096];                 // byte | nbBits << 8   (HufDecompress.cs:109-113)
};
#define ZS_DLDS_PREP (offsetof(DLds, ML))
#ifdef ZS_PREP_PROFILE
#define PPROF(L, k) do { if (zs_lane() == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (L).pp[k] += now_ - (L).ppMark; (L).ppMark = now_; } } while (0)
#else
#define PPROF(L, k) do { } while (0)
#endif
__device__ __forceinline__ uint32_t ofBaseOf(uint32_t sym) { return sym == 0 ? 0u : (sym == 1 ? 1u : ((1u << sym) - 3u)); }   // OF_base :1088

__device__ __forceinline__ uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t rd24(const uint8_t *p) { return rd16(p) | ((uint32_t)p[2] << 16); }
__device__ __forceinline__ uint32_t rd32(const uint8_t *p) { return zs_load32(p); }

// ---- backward bit reader (BitStream.cs:322-494).  64-bit container here; the reference's 32-bit reload
//      points only decide where a damaged stream is rejected, the bits read are the same. ----
struct BitR { const uint8_t *start; uint32_t size; int64_t bitPos; /* number of unread bits below the cursor */ uint32_t err;
              uint64_t cont; int64_t contLo; /* cont = stream bits [contLo, contLo + 64) */ };
__device__ __forceinline__ void br_fill(BitR &b)
{
    // 8 bytes whose top byte holds the bit just below the cursor (or the first 8 bytes of the stream)
    int64_t b0 = 0;
    if (b.size >= 8) {
        b0 = ((b.bitPos - 1) >> 3) - 7; if (b0 < 0) b0 = 0;
        if (b0 > (int64_t)b.size - 8) b0 = (int64_t)b.size - 8;
        b.cont = zs_load64(b.start + b0);
    } else {
        uint64_t w = 0;
        for (uint32_t k = 0; k < b.size; k++) w |= (uint64_t)b.start[k] << (8 * k);
        b.cont = w;
    }
    b.contLo = 8 * b0;
}
__device__ __forceinline__ void br_init(BitR &b, const uint8_t *src, uint32_t size)
{
    b.start = src; b.size = size; b.err = 0; b.bitPos = 0; b.cont = 0; b.contLo = 0;
    if (size == 0) { b.err = 1; return; }
    const uint32_t last = src[size - 1];
    if (last == 0) { b.err = 1; return; }
    b.bitPos = (int64_t)size * 8 - (int64_t)(8 - zs_highbit(last));      // bits below the end mark
    br_fill(b);
}
// next n bits (n <= 32) below the cursor, most significant first; bits below the stream start read as 0
__device__ __forceinline__ uint32_t br_look(BitR &b, uint32_t n)
{
    if (n == 0 || b.bitPos <= 0) return 0;
    int64_t rel = b.bitPos - (int64_t)n - b.contLo;       // position of the lowest wanted bit inside cont
    if (rel < 0 && b.contLo > 0) { br_fill(b); rel = b.bitPos - (int64_t)n - b.contLo; }
    const uint64_t v = (rel >= 0) ? (b.cont >> (uint32_t)rel) : ((rel <= -64) ? 0ull : (b.cont << (uint32_t)(-rel)));
    return (uint32_t)(v & ((n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1)));
}
__device__ __forceinline__ uint32_t br_read(BitR &b, uint32_t n) { const uint32_t v = br_look(b, n); b.bitPos -= n; return v; }

// ---- a header region staged in LDS (all lanes load 4 bytes each: 256 bytes, zero beyond the region), read by the serial
//      parsers below, which run on one lane: a global load inside them is a memory round trip per few bits.
//      (Keeping the window in registers and reading it with v_readlane inside the one-lane branch is NOT safe: register
//      copies made under the one-lane exec mask drop the other lanes' values.) ----
__device__ __forceinline__ void hw_stage(uint32_t *win, const uint8_t *p, uint32_t size)
{
    const uint32_t lane = (uint32_t)zs_lane(), o = 4u * lane;
    uint32_t v = 0;
    if (o + 4 <= size) v = zs_load32(p + o);
    else for (uint32_t k = 0; k < 4; k++) if (o + k < size) v |= (uint32_t)p[o + k] << (8 * k);
    win[lane] = v;
    if (lane < 2) win[64 + lane] = 0;
    wave_sync();
}
// unaligned dword at byte offset off of the staged region (off <= 256)
__device__ __forceinline__ uint32_t rw_rd32(const uint32_t *win, uint32_t off)
{
    const uint32_t i = min(off >> 2, 64u);
    return __builtin_amdgcn_alignbyte(win[i + 1], win[i], off & 3u);
}

// ---- ReadNCount (EntropyCommon.cs:79-188), one lane; the header sits in the staged window w (a description of <= 53
//      symbols is < 100 bytes).  Offsets instead of pointers; hbSize may exceed the window (the rest of the block). ----
__device__ __forceinline__ uint32_t readNCount(int16_t *norm, uint32_t *maxSVPtr, uint32_t *tableLogPtr, const uint32_t *w, uint32_t hbSize, uint32_t wOff)
{
    const int32_t iend = (int32_t)hbSize; int32_t ip = 0;
    int nbBits, remaining, threshold, bitCount; uint32_t bitStream, charnum = 0; int previous0 = 0;
    if (hbSize < 4) return ZE(E_srcSize_wrong);
    bitStream = rw_rd32(w, wOff + (uint32_t)ip);
    nbBits = (int)(bitStream & 0xF) + 5;
    if (nbBits > 15) return ZE(E_tableLog_tooLarge);
    bitStream >>= 4; bitCount = 4;
    *tableLogPtr = (uint32_t)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    while ((remaining > 1) & (charnum <= *maxSVPtr)) {
        if (previous0) {
            uint32_t n0 = charnum;
            while ((bitStream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (ip < iend - 5) { ip += 2; bitStream = rw_rd32(w, wOff + (uint32_t)ip) >> bitCount; }
                else { bitStream >>= 16; bitCount += 16; }
            }
            while ((bitStream & 3) == 3) { n0 += 3; bitStream >>= 2; bitCount += 2; }
            n0 += bitStream & 3; bitCount += 2;
            if (n0 > *maxSVPtr) return ZE(48);
            while (charnum < n0) norm[charnum++] = 0;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; bitStream = rw_rd32(w, wOff + (uint32_t)ip) >> bitCount; }
            else bitStream >>= 2;
        }
        {
            const int max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (uint32_t)(threshold - 1)) < (uint32_t)max) { count = (int)(bitStream & (uint32_t)(threshold - 1)); bitCount += nbBits - 1; }
            else { count = (int)(bitStream & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= max; bitCount += nbBits; }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[charnum++] = (int16_t)count;
            previous0 = !count;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= (int)(8 * (iend - 4 - ip)); ip = iend - 4; }
            bitStream = rw_rd32(w, wOff + (uint32_t)ip) >> (bitCount & 31);
        }
    }
    if (remaining != 1) return ZE(E_corruption_detected);
    if (bitCount > 32) return ZE(E_corruption_detected);
    *maxSVPtr = charnum - 1;
    ip += (bitCount + 7) >> 3;
    return (uint32_t)ip;
}

// ---- BuildFSETable (ZStdDecompress.cs:958-1034) by all 64 lanes; lane s owns symbol s (maxSym <= 52).
// The reference walks the cells in the order p_k = (k * step) & mask, skipping the low-probability area at the top, and
// hands them to the symbols in turn; step is odd, so p_k is a permutation: the k-th visit is valid iff p_k <= highThreshold
// and takes the j-th entry of the expanded symbol list, j = valid visits before k.  Then nextState numbers go to the cells
// of a symbol in ascending cell order. ----
__device__ __forceinline__ void buildSeqTableWave(DLds &L, SeqSym *cells, uint32_t *tableLogOut, uint32_t maxSym, uint32_t tableLog)
{
    const uint32_t lane = (uint32_t)zs_lane();
    const uint32_t tableSize = 1u << tableLog, tableMask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    const uint64_t below = (1ull << lane) - 1ull;
    uint16_t *cumul = L.u.tb.symStart, *symbolNext = L.u.tb.symbolNext;              // scratch of this phase
    const int n = (lane <= maxSym) ? (int)L.u.tb.norm[lane] : 0;
    const bool low = n == -1;
    const uint64_t lowMask = __ballot(low);
    const uint32_t highThreshold = tableSize - 1 - (uint32_t)__popcll(lowMask);
    if (low) cells[tableSize - 1 - (uint32_t)__popcll(lowMask & below)].sym = (uint8_t)lane;   // :975-978, symbols ascending take cells descending
    const uint32_t cnt = n > 0 ? (uint32_t)n : 0u;
    const uint32_t incl = wave_incl_scan(cnt);
    cumul[lane] = (uint16_t)(incl - cnt);
    symbolNext[lane] = (uint16_t)(low ? 1 : cnt);
    if (lane == 0) *tableLogOut = tableLog;
    wave_sync();
    uint32_t validBefore = 0;
    for (uint32_t base = 0; base < tableSize; base += 64) {
        const uint32_t k = base + lane, p = (k * step) & tableMask;
        const bool valid = k < tableSize && p <= highThreshold;
        const uint64_t vm = __ballot(valid);
        if (valid) {
            const uint32_t j = validBefore + (uint32_t)__popcll(vm & below);
            uint32_t lo = 0, hi = maxSym + 1;                                        // last symbol whose first entry index is <= j
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cumul[mid] <= j) lo = mid; else hi = mid; }
            cells[p].sym = (uint8_t)lo;
        }
        validBefore += (uint32_t)__popcll(vm);
    }
    wave_sync();
    // nextState numbers in ascending cell order (:1017-1027), 64 cells at a time: every lane ors its bit into its symbol's 64-bit lane
    // mask (LDS); the mask read back gives its rank among the chunk's cells of that symbol and the symbol's count in the chunk, which
    // the symbol's first lane adds to symbolNext.  (A loop over the chunk's distinct symbols cost three LDS round trips per symbol.)
    uint32_t *symMask = L.u.tb.symMask;
    for (uint32_t base = 0; base < tableSize; base += 64) {
        const uint32_t u = base + lane;
        const bool in = u < tableSize;
        const uint32_t sym = in ? cells[u].sym : 0u;
        symMask[2 * lane] = 0; symMask[2 * lane + 1] = 0;
        wave_sync();
        if (in) atomicOr(&symMask[2 * sym + (lane >> 5)], 1u << (lane & 31u));
        wave_sync();
        uint32_t first = 0, rank = 1, cnt = 0;
        if (in) {
            const uint32_t lo = symMask[2 * sym], hi = symMask[2 * sym + 1];
            rank = (uint32_t)__popc(lo & (uint32_t)below) + (uint32_t)__popc(hi & (uint32_t)(below >> 32));
            cnt = (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
            first = symbolNext[sym];
            const uint32_t nextState = first + rank;                                                       // :1021-1023
            const uint32_t nb = tableLog - zs_highbit(nextState);
            cells[u].nbBits = (uint8_t)nb;
            cells[u].nextState = (uint16_t)((nextState << nb) - tableSize);
        }
        wave_sync();
        if (in && rank == 0) symbolNext[sym] = (uint16_t)(first + cnt);
        wave_sync();
    }
    wave_sync();
}

// ---- ReadStats + table fill (EntropyCommon.cs:198-269, HufDecompress.cs:117-180) ----
// The weight description (<= 128 bytes) is held in a register window.  One lane parses FSE-compressed weights (serial by
// nature: two interleaved FSE states); everything after the weights -- checks, rank counts, cell ranges, table fill -- runs
// on all lanes, lane s of chunk c owning symbol 64c + s.  returns header size or error.
// TO_GLOBAL false: the flat table goes to L.huf (LDS).  TO_GLOBAL true (k_dec_prep): a two-level table goes to hufGlobal
// (ZS_HUF2_ENTRIES 16-bit entries); a table of more than maxLog bits or one that does not fit is reported as
// E_tableLog_tooLarge before anything the fast path would use is complete (the item is then left to the general kernel)
#define ZS_HUF2_SUBS    32u                         // sub-tables (9-bit prefixes holding longer codes) the fast path has room for
#define ZS_HUF2_ENTRIES (512u + 4u * ZS_HUF2_SUBS)
template <bool TO_GLOBAL>
__device__ __forceinline__ uint32_t readHufTableT(DLds &L, const uint8_t *src, uint32_t srcSize, uint16_t *hufGlobal, uint32_t maxLog)
{
    const uint32_t lane = (uint32_t)zs_lane();
    if (!srcSize) return ZE(E_srcSize_wrong);
    PPROF(L, 0);
    hw_stage(L.u.tb.hdrWin, src, min(srcSize, 256u));
    const uint32_t *hw = L.u.tb.hdrWin;
    uint32_t iSize = src[0], oSize = 0;
    PPROF(L, 1);
    if (iSize >= 128) {                                                     // direct: 4 bits per weight (EntropyCommon.cs:215-225)
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return ZE(E_srcSize_wrong);
        if (oSize >= 256) return ZE(E_corruption_detected);
        for (uint32_t n = lane; n < 2 * iSize; n += 64) { const uint32_t byte = src[1 + n / 2]; L.u.tb.weights[n] = (uint8_t)((n & 1) ? (byte & 15) : (byte >> 4)); }
        wave_sync();
    } else {
        if (iSize + 1 > srcSize) return ZE(E_srcSize_wrong);
        // FSE-compressed weights (FseDecompress.cs:233-332), table log <= 6, read from the staged window (offset 1): the counts are parsed by one
        // lane, the decoding table (FseDecompress.cs:111-181: the same construction as a sequence table's) is built by all of them, the weights
        // are decoded by one lane again - two interleaved states: their cells are read together, the stream bits come from a 64-bit container.
        if (lane == 0) {
            uint32_t tableLog = 0, maxSV = 63;
            const uint32_t nc = readNCount(L.u.tb.norm, &maxSV, &tableLog, hw, iSize, 1);       // the description starts at byte 1
            uint32_t err = 0;
            if (isErr(nc)) err = nc;
            else if (tableLog > 6 || maxSV > 63) err = ZE(E_tableLog_tooLarge);
            else if (iSize <= nc) err = ZE(E_corruption_detected);
            L.misc[0] = err; L.misc[1] = nc; L.misc[2] = tableLog; L.misc[3] = maxSV;
        }
        wave_sync();
        if (L.misc[0]) return L.misc[0];
        PPROF(L, 2);
        SeqSym *wc = reinterpret_cast<SeqSym *>(L.u.tb.wfse);
        buildSeqTableWave(L, wc, &L.misc[5], L.misc[3], L.misc[2]);
        PPROF(L, 3);
        if (lane == 0) {
            uint32_t result = 0;
            do {
                const uint32_t nc = L.misc[1], tableLog = L.misc[2];
                {   // two interleaved states, FseDecompress.cs:233-295; the stream is bytes [1 + nc, 1 + iSize) of the window
                    const uint32_t s0 = 1 + nc, ssz = iSize - nc;
                    const uint32_t lastByte = rw_rd32(hw, s0 + ssz - 1) & 0xFFu;
                    if (lastByte == 0) { result = ZE(E_corruption_detected); break; }
                    int32_t bitPos = (int32_t)(ssz * 8 - (8 - zs_highbit(lastByte)));
                    // container: the stream bits below the cursor from its top bit down, `avail` of them valid (bits below the stream's start are 0, BitStream.cs:412)
                    uint64_t cont = 0; int32_t avail = 0;
                    auto refill = [&]() {
                        if (bitPos <= 0) { cont = 0; avail = 64; return; }
                        const int32_t bh = (bitPos - 1) >> 3;
                        uint64_t raw;
                        if (bh >= 7) raw = (uint64_t)rw_rd32(hw, s0 + (uint32_t)bh - 7) | ((uint64_t)rw_rd32(hw, s0 + (uint32_t)bh - 3) << 32);
                        else raw = ((uint64_t)rw_rd32(hw, s0) | ((uint64_t)rw_rd32(hw, s0 + 4) << 32)) << (8 * (7 - bh));
                        const uint32_t sh = 7u - (uint32_t)((bitPos - 1) & 7);
                        cont = raw << sh; avail = 64 - (int32_t)sh;
                    };
                    auto readBits = [&](uint32_t n) -> uint32_t {            // n <= 6
                        if (avail < (int32_t)n) refill();
                        const uint32_t v = n ? (uint32_t)(cont >> (64 - n)) : 0u;
                        cont <<= n; avail -= (int32_t)n; bitPos -= (int32_t)n;
                        return v;
                    };
                    uint32_t s1 = readBits(tableLog), s2 = readBits(tableLog);
                    uint32_t op = 0; bool bad = false;
                    for (;;) {
                        if (op > 253) { bad = true; break; }
                        const SeqSym c1 = wc[s1], c2 = wc[s2];
                        L.u.tb.weights[op++] = c1.sym; s1 = c1.nextState + readBits(c1.nbBits);
                        if (bitPos < 0) { L.u.tb.weights[op++] = c2.sym; break; }
                        if (op > 253) { bad = true; break; }
                        L.u.tb.weights[op++] = c2.sym; s2 = c2.nextState + readBits(c2.nbBits);
                        if (bitPos < 0) { L.u.tb.weights[op++] = wc[s1].sym; break; }
                    }
                    if (bad) { result = ZE(E_corruption_detected); break; }
                    result = op;                                             // number of weights
                }
            } while (0);
            L.misc[0] = result;
            PPROF(L, 4);
        }
        wave_sync();
        const uint32_t r = L.misc[0];
        if (isErr(r)) return r;
        oSize = r;
    }
    // ---- from here on all lanes.  weights[0 .. oSize) are known; the last symbol's weight is implied ----
    uint32_t wgt[4]; uint32_t weightTotal = 0; bool bad = false;
    #pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint32_t n = 64 * c + lane;
        wgt[c] = (n < oSize) ? (uint32_t)L.u.tb.weights[n] : 0u;
        bad |= wgt[c] >= 12;
        weightTotal += (1u << min(wgt[c], 12u)) >> 1;
    }
    if (__ballot(bad)) return ZE(E_corruption_detected);
    weightTotal = wave_sum(weightTotal);
    if (weightTotal == 0) return ZE(E_corruption_detected);
    const uint32_t tableLog = zs_highbit(weightTotal) + 1;
    if (tableLog > 12) return ZE(E_corruption_detected);
    if (TO_GLOBAL && tableLog > maxLog) return ZE(E_tableLog_tooLarge);
    const uint32_t rest = (1u << tableLog) - weightTotal, lastWeight = zs_highbit(rest) + 1;
    if ((1u << zs_highbit(rest)) != rest) return ZE(E_corruption_detected);
    #pragma unroll
    for (uint32_t c = 0; c < 4; c++) if (64 * c + lane == oSize) wgt[c] = lastWeight;
    const uint32_t nbSymbols = oSize + 1;
    // symbols per weight, then the first cell of each weight (cells sorted by weight, then by symbol: HufDecompress.cs:148-176)
    uint32_t rankStart[13], rankIdx[13], nSorted = 0;                         // uniform values; rankIdx: symbols (of a weight) in front, in cell order
    {
        uint32_t next = 0;
        #pragma unroll
        for (uint32_t wv = 1; wv <= 12; wv++) {
            uint32_t cnt = 0;
            #pragma unroll
            for (uint32_t c = 0; c < 4; c++) cnt += (uint32_t)__popcll(__ballot(wgt[c] == wv));
            if (wv == 1 && ((cnt < 2) || (cnt & 1))) bad = true;             // EntropyCommon.cs:262
            rankStart[wv] = next; next += cnt << (wv - 1);
            rankIdx[wv] = nSorted; nSorted += cnt;
        }
        rankStart[0] = 0; rankIdx[0] = 0;
    }
    if (bad) return ZE(E_corruption_detected);
    if (lane == 0) L.hufLog = tableLog;
    // every symbol with a weight fills its cells: start = first cell of its weight + (lower symbols of that weight) * cells per symbol
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t before[13];
    #pragma unroll
    for (uint32_t wv = 0; wv <= 12; wv++) before[wv] = 0;
    // TO_GLOBAL: the symbols in cell order (first cell, table entry) for the entry-by-entry fill below; scratch: the LL table's room
    uint16_t *sStart = reinterpret_cast<uint16_t *>(L.LL.cells), *sEntry = sStart + 264;
    if (TO_GLOBAL) { for (uint32_t i = lane; i < 264; i += 64) sStart[i] = 0xFFFFu; wave_sync(); }
    #pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        uint32_t myStart = 0, myRank = 0;
        #pragma unroll
        for (uint32_t wv = 1; wv <= 12; wv++) {
            const uint64_t m = __ballot(wgt[c] == wv);
            if (wgt[c] == wv) { const uint32_t k = before[wv] + (uint32_t)__popcll(m & below); myStart = rankStart[wv] + (k << (wv - 1)); myRank = rankIdx[wv] + k; }
            before[wv] += (uint32_t)__popcll(m);
        }
        if (wgt[c]) L.u.tb.symStart[64 * c + lane] = (uint16_t)myStart;
        L.u.tb.weights[64 * c + lane] = (uint8_t)wgt[c];
        if (TO_GLOBAL && wgt[c]) { sStart[myRank] = (uint16_t)myStart; sEntry[myRank] = (uint16_t)((64 * c + lane) | ((tableLog + 1 - wgt[c]) << 8)); }
    }
    wave_sync();
#ifdef ZS_DEC_ERRLINE
    {   // debugging aid: the serial statement (old code) must agree on every derived value
        uint32_t mism = 0;
        if (lane == 0) {
            uint32_t rk[16]; for (int i = 0; i < 16; i++) rk[i] = 0;
            uint32_t wt = 0;
            for (uint32_t n = 0; n < oSize; n++) { rk[L.u.tb.weights[n]]++; wt += (1u << L.u.tb.weights[n]) >> 1; }
            const uint32_t tl = zs_highbit(wt) + 1;
            const uint32_t rs = (1u << tl) - wt, lw = zs_highbit(rs) + 1;
            if (tl != tableLog) mism = 0x810000u | tl;
            else if (lw != lastWeight || L.u.tb.weights[oSize] != lw) mism = 0x820000u | (lastWeight << 12) | ((uint32_t)L.u.tb.weights[oSize] << 8) | (oSize & 0xFFu);
            else {
                rk[lw]++;
                uint32_t next = 0; for (uint32_t n = 1; n < tl + 1; n++) { const uint32_t cur = next; next += rk[n] << (n - 1); rk[n] = cur; }
                for (uint32_t n = 0; n < nbSymbols && !mism; n++) { const uint32_t w = L.u.tb.weights[n]; if (w) { if (rk[w] != L.u.tb.symStart[n]) mism = 0x830000u | n; rk[w] += (1u << w) >> 1; } }
            }
        }
        mism = wave_get(mism, 0);
        if (mism) return 0xFF000000u | mism;
    }
#endif
    PPROF(L, 5);
    if (!TO_GLOBAL)
    for (uint32_t n = 0; n < nbSymbols; n++) {            // uniform loop; lanes fill one symbol's cells together
        const uint32_t w = L.u.tb.weights[n];
        if (!w) continue;
        const uint32_t length = (1u << w) >> 1, startAt = L.u.tb.symStart[n];
        const uint16_t e = (uint16_t)(n | ((tableLog + 1 - w) << 8));
        if (!TO_GLOBAL) { for (uint32_t u = lane; u < length; u += 64) L.huf[startAt + u] = e; }
    }
    if (TO_GLOBAL) {
        // The fast path's two-level table (ZS_HUF2_*): P[512] is indexed by the next 9 stream bits; a code of <= 9 bits owns whole
        // entries of P (tables of < 9 bits are spread out), a 9-bit prefix under which 10- and 11-bit codes sit points (bit 15)
        // to a 4-entry sub-table indexed by the next 2 bits.  1.25 KiB an item instead of the 4 KiB of a flat 2^11 table; more
        // than ZS_HUF2_SUBS such prefixes: the item is left to the general kernel.
        // Filled entry by entry: lane l owns P[8 l .. 8 l + 7]; an entry finds its symbol in the cell-ordered list by binary search (its
        // 8 searches side by side) and the lane stores its entries as one 16-byte piece.  (Symbol by symbol - the lanes filling one
        // symbol's run - was ~150 rounds of an LDS read and a 2-byte scattered store: 44 % of the prep kernel's time.)
        const uint32_t extra = tableLog > 9 ? tableLog - 9 : 0u, rep = tableLog < 9 ? 9 - tableLog : 0u;
        uint32_t r[8], flat[8];
        #pragma unroll
        for (uint32_t q = 0; q < 8; q++) { const uint32_t i = 8 * lane + q; flat[q] = tableLog >= 9 ? i << extra : i >> rep; r[q] = 0; }
        #pragma unroll
        for (uint32_t step = 128; step >= 1; step >>= 1) {
            uint32_t probe[8];
            #pragma unroll
            for (uint32_t q = 0; q < 8; q++) probe[q] = sStart[r[q] + step];
            #pragma unroll
            for (uint32_t q = 0; q < 8; q++) if (probe[q] <= flat[q]) r[q] += step;
        }
        uint32_t ent[8], cnt = 0;
        #pragma unroll
        for (uint32_t q = 0; q < 8; q++) { ent[q] = sEntry[r[q]]; cnt += (ent[q] >> 8) > 9u; }
        const uint32_t incl = wave_incl_scan(cnt);
        if (wave_last(incl) > ZS_HUF2_SUBS) {
            // more long-code prefixes than sub-tables (wide alphabets: binaries): the FLAT table of 2^11 entries instead (4 KiB, the slot's whole
            // room; k_dec_huffman's flat class).  Lane l owns entries 32 l .. 32 l + 31: a search for the first, then along the cell-ordered symbols.
            const uint32_t sh = 11u - tableLog;                                  // a table of fewer bits is spread out
            const uint32_t j0 = (32u * lane) >> sh;
            uint32_t rr = 0;
            #pragma unroll
            for (uint32_t step = 128; step >= 1; step >>= 1) if (sStart[rr + step] <= j0) rr += step;
            uint32_t pk[16];
            #pragma unroll
            for (uint32_t q = 0; q < 32; q++) {
                const uint32_t j = (32u * lane + q) >> sh;
                while (sStart[rr + 1] <= j) rr++;
                const uint32_t e = sEntry[rr];
                if (q & 1u) pk[q >> 1] |= e << 16; else pk[q >> 1] = e;
            }
            #pragma unroll
            for (uint32_t q = 0; q < 4; q++) *reinterpret_cast<uint4 *>(hufGlobal + 32 * lane + 8 * q) = make_uint4(pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]);
            wave_sync();
            PPROF(L, 6);
            return (iSize + 1) | 0x40000000u;                                    // (bit 30: the flat table; header sizes are < 2^17)
        }
        uint32_t id = incl - cnt;
        #pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            if ((ent[q] >> 8) > 9u) {
                // the sub-table: slot t stands for flat cell (prefix << extra) + (t >> (2 - extra)); the symbols under the prefix follow r[q]
                uint32_t rr = r[q]; uint32_t se[4];
                #pragma unroll
                for (uint32_t t = 0; t < 4; t++) {
                    const uint32_t j = flat[q] + (t >> (2 - extra));
                    while (sStart[rr + 1] <= j) rr++;
                    se[t] = sEntry[rr];
                }
                const uint64_t pack = (uint64_t)(se[0] | (se[1] << 16)) | ((uint64_t)(se[2] | (se[3] << 16)) << 32);
                __builtin_memcpy(hufGlobal + 512 + id * 4, &pack, 8);
                ent[q] = 0x8000u | id;
                id++;
            }
        }
        const uint4 pk = make_uint4(ent[0] | (ent[1] << 16), ent[2] | (ent[3] << 16), ent[4] | (ent[5] << 16), ent[6] | (ent[7] << 16));
        *reinterpret_cast<uint4 *>(hufGlobal + 8 * lane) = pk;
    }
    wave_sync();
    PPROF(L, 6);
    return iSize + 1;
}
__device__ __forceinline__ uint32_t readHufTable(DLds &L, const uint8_t *src, uint32_t srcSize) { return readHufTableT<false>(L, src, srcSize, nullptr, 12); }

// ---- stream windows in LDS.  The serial decoders (Huffman: one lane per stream; sequences: lane 0) read their backward
//      bitstreams from LDS: a global load inside such a dependent chain costs a full memory round trip per ~64 bits
//      (and on gfx9 a wait for it also waits for the stores in flight).  All 64 lanes stage the bytes. ----
// win[] <- stream bytes [base - 8, base + W); bytes outside [0, size) read as 0 (bits below the stream start are 0, BitStream.cs:412)
__device__ __forceinline__ void stageWindow(uint32_t *win, const uint8_t *src, uint32_t size, int32_t base, uint32_t W)
{
    const uint32_t lane = (uint32_t)zs_lane();
    for (uint32_t j = lane; j < (W + 8) / 4 + 2; j += 64) {
        const int32_t p = base - 8 + 4 * (int32_t)j;
        uint32_t v = 0;
        if (p >= 0 && p + 4 <= (int32_t)size) v = zs_load32(src + p);
        else for (int k = 0; k < 4; k++) { const int32_t q = p + k; if (q >= 0 && q < (int32_t)size) v |= (uint32_t)src[q] << (8 * k); }
        win[j] = v;
    }
}
__device__ __forceinline__ uint64_t wave_get64(uint64_t v, int l) { return (uint64_t)wave_get((uint32_t)v, l) | ((uint64_t)wave_get((uint32_t)(v >> 32), l) << 32); }
// 8 window bytes from byte offset o (aligned dword reads, shifted into place)
__device__ __forceinline__ uint64_t win64(const uint32_t *win, uint32_t o)
{
    const uint32_t *d = win + (o >> 2);
    const uint32_t w0 = d[0], w1 = d[1], w2 = d[2], sh = o & 3u;
    return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32);
}
// the bit container of a serial decoder: c holds the next bits at its top, avail of them are valid, bitPos = unread bits
struct BitC { uint64_t c; uint32_t avail; int32_t bitPos; };      // streams are < 2^28 bytes (a block is < 128 KiB)
__device__ __forceinline__ void bc_refill(BitC &b, const uint32_t *win, int32_t base)
{
    if (b.bitPos <= 0) { b.c = 0; b.avail = 64; return; }               // past the start: zeros (the caller rejects the stream)
    const int32_t bh = (b.bitPos - 1) >> 3;                             // byte holding the next bit
    const uint64_t raw = win64(win, (uint32_t)(bh - base + 1));         // stream bytes [bh - 7, bh]
    const uint32_t sh = 7u - (uint32_t)((b.bitPos - 1) & 7);
    b.c = raw << sh; b.avail = 64 - sh;
}
__device__ __forceinline__ bool bc_init(BitC &b, const uint8_t *src, uint32_t size)   // BitStream.cs:322-378
{
    b.c = 0; b.avail = 0; b.bitPos = 0;
    if (size == 0) return false;
    const uint32_t last = src[size - 1];
    if (last == 0) return false;
    b.bitPos = (int32_t)(size * 8 - (8 - zs_highbit(last)));           // bits below the end mark
    return true;
}
__device__ __forceinline__ int32_t bc_windowBase(const BitC &b, uint32_t W)
{
    const int32_t bh = (b.bitPos > 0) ? ((b.bitPos - 1) >> 3) : 0;
    return (bh + 1 > (int32_t)W) ? bh + 1 - (int32_t)W : 0;
}
// n bits (n <= 32) from the top of the container, no availability check; n = 0 gives 0 (bit-field extract of width 0)
__device__ __forceinline__ uint32_t bc_take(BitC &b, uint32_t n)
{
    const uint32_t v = __builtin_amdgcn_ubfe((uint32_t)(b.c >> 32), 32u - n, n);
    b.c <<= n; b.avail -= n; b.bitPos -= (int32_t)n;
    return v;
}

// the Huffman streams of one block (HufDecompress.cs:222-358): stream k on lane k (nStreams = 1 or 4), in rounds of
// "stage ZS_LITWIN bytes of every stream, decode until a stream needs more".  8 decoded bytes go out per store.
// SelectDecoder (HufDecompress.cs:1056-1095): 1 = the reference takes its double-symbol decoder for a 4-stream section with a new table
__device__ __forceinline__ uint32_t hufSelectDecoder(uint32_t dstSize, uint32_t cSrcSize)
{
    // algoTime[Q][single, double] = {tableTime, decode256Time}
    const uint32_t t0[16] = { 0, 0, 38, 448, 556, 714, 883, 897, 926, 947, 1107, 1177, 1242, 1349, 1455, 722 };
    const uint32_t d0[16] = { 0, 0, 130, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128 };
    const uint32_t t1[16] = { 1, 1, 1313, 1353, 1353, 1418, 1437, 1515, 1613, 1729, 2083, 2379, 2415, 2644, 2422, 1891 };
    const uint32_t d1[16] = { 1, 1, 74, 74, 74, 74, 74, 75, 75, 77, 81, 87, 93, 106, 124, 145 };
    const uint32_t Q = (cSrcSize >= dstSize) ? 15u : (uint32_t)((uint64_t)cSrcSize * 16 / dstSize);
    const uint32_t D256 = dstSize >> 8;
    uint32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
    #pragma unroll
    for (uint32_t q = 0; q < 16; q++) if (q == Q) { a0 = t0[q]; b0 = d0[q]; a1 = t1[q]; b1 = d1[q]; }
    const uint32_t DTime0 = a0 + b0 * D256;
    uint32_t DTime1 = a1 + b1 * D256;
    DTime1 += DTime1 >> 3;
    return DTime1 < DTime0 ? 1u : 0u;
}

// The reference decodes a literal section with its single-symbol (X2) or its double-symbol (X4) Huffman decoder (SelectDecoder,
// HufDecompress.cs:1082-1095; a repeated table: the decoder that built it; a dictionary's table: X4).  Both read the same symbols
// from the same bits; they differ in one place: a stream's LAST symbol, when the X4 table entry found for it holds two symbols
// (HUF_decodeLastSymbolX4, HufDecompress.cs:369-385): the entry's bits (both symbols') are skipped only if bits were left, and the
// count is clamped at the stream's end -- so up to l1 + l2 unread bits pass where the X2 decoder asks for exactly l1; with no bit
// left the lookup wraps to the top of the stream's first word and nothing is skipped.  The streams are decoded the X2 way here;
// a stream of an X4 section that fails the strict end check is walked again (rare: damaged input only) with the X4 decoder's
// grouping (a lookup takes two symbols when their lengths sum to <= 12 = its table log) to see whether its last symbol is taken
// alone and what the entry holds.  Oracle: HUF_decodeStreamX4 / HUF_decodeLastSymbolX4 in oracle/zso_decoder.c.
__device__ static bool hufX4TailAccepts(const uint16_t *huf, uint32_t dtLog, const uint8_t *src, uint32_t size, uint32_t n, uint8_t *out)
{
    if (size == 0 || src[size - 1] == 0 || n == 0) return false;
    auto peek12 = [&](int32_t bp) -> uint32_t {                        // the 12 bits below unread-bit count bp; bits below the stream start read 0
        const int32_t bh = (bp - 1) >> 3;
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) { const int32_t q = bh - k; w |= (q >= 0 ? (uint32_t)src[q] : 0u) << (24 - 8 * k); }
        return (w << (7u - (uint32_t)((bp - 1) & 7))) >> 20;
    };
    auto info = [&](uint32_t v, uint32_t &l1, uint32_t &both, uint32_t &sym) -> bool {   // the X4 entry at 12-bit index v
        const uint32_t e1 = huf[v >> (12u - dtLog)]; l1 = e1 >> 8; sym = e1 & 0xFFu;
        const uint32_t rest = (v << l1) & 0xFFFu;
        const uint32_t e2 = huf[rest >> (12u - dtLog)];
        both = l1 + (e2 >> 8);
        return both <= 12u;
    };
    int32_t bp = (int32_t)(size * 8 - (8 - zs_highbit(src[size - 1])));
    uint32_t i = 0, l1, both, sym;
    while (i + 2 <= n) {
        if (bp <= 0) return false;                                      // more than one symbol to go and no bit left: the count passes the end
        if (info(peek12(bp), l1, both, sym)) { bp -= (int32_t)both; i += 2; } else { bp -= (int32_t)l1; i += 1; }
    }
    if (i == n) return bp == 0;
    if (bp < 0) return false;
    if (bp == 0) {                                                      // the lookup wraps: the top 12 bits of the stream's first word (BitStream.cs:412, shift & 31)
        uint32_t c = 0;
        for (uint32_t k = 0; k < 4 && k < size; k++) c |= (uint32_t)src[k] << (8 * k);
        if (!info(c >> 20, l1, both, sym)) return false;                // a one-symbol entry is skipped: past the end
        out[n - 1] = (uint8_t)sym;
        return true;
    }
    if (info(peek12(bp), l1, both, sym)) return (uint32_t)bp <= both;
    return (uint32_t)bp == l1;
}

__device__ __forceinline__ bool hufDecodeStreams(DLds &L, uint32_t nStreams, uint8_t *out, uint32_t n, const uint8_t *src, uint32_t size, bool useX4, uint64_t *g_prof)
{
    (void)g_prof;
    const uint32_t lane = (uint32_t)zs_lane();
    const bool mine = lane < nStreams;
    BitC b; b.c = 0; b.avail = 0; b.bitPos = 0;
    const bool okInit = !mine || bc_init(b, src, size);
    if (__ballot(!okInit)) return false;
    const uint32_t dtLog = L.hufLog;
    uint32_t i = 0;
    bool done = !mine || n == 0;
    for (;;) {
        const int32_t base = bc_windowBase(b, ZS_LITWIN);
        for (uint32_t k = 0; k < nStreams; k++) {
            const uint64_t sp = wave_get64((uint64_t)(uintptr_t)src, (int)k);
            stageWindow(L.u.litWin[k], reinterpret_cast<const uint8_t *>(sp), wave_get(size, (int)k), (int32_t)wave_get((uint32_t)base, (int)k), ZS_LITWIN);
        }
        wave_sync();
#ifdef ZS_DEC_PROFILE
        const uint64_t t7_ = __builtin_readcyclecounter();
#endif
        if (!done) {
            const uint32_t *win = L.u.litWin[lane];
            const uint32_t sh = 32u - dtLog;
            // bulk: a refill leaves >= 57 bits and 4 symbols take <= 4 * 12: four symbols per refill without a check between
            // them, one 4-byte store.  Not near the head of the stream (bits below it read as 0) nor at the window edge.
            while (i + 4 <= n) {
                const int32_t bh = (b.bitPos - 1) >> 3;
                if (b.bitPos < 64 || (base > 0 && bh < base + 16)) break;
                uint64_t c = win64(win, (uint32_t)(bh - base + 1)) << (7u - (uint32_t)((b.bitPos - 1) & 7));
                uint32_t used = 0, pack = 0;
                #pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t e = L.huf[(uint32_t)(c >> 32) >> sh];
                    const uint32_t nb = e >> 8;
                    c <<= nb; used += nb; pack |= (e & 0xFFu) << (8 * k);
                }
                b.bitPos -= (int32_t)used;
                __builtin_memcpy(out + i, &pack, 4);
                i += 4;
            }
            b.avail = 0;                                                 // the careful loop below refills first
            while (i < n) {
                if (b.avail < dtLog) {
                    if (b.bitPos > 0 && base > 0 && ((b.bitPos - 1) >> 3) < base + 8) break;      // window used up: next round
                    bc_refill(b, win, base);
                }
                const uint32_t e = L.huf[(uint32_t)(b.c >> 32) >> sh];
                const uint32_t nb = e >> 8;
                b.c <<= nb; b.avail -= nb; b.bitPos -= (int32_t)nb;
                out[i++] = (uint8_t)e;
            }
            if (i == n) done = true;
        }
        wave_sync();
#ifdef ZS_DEC_PROFILE
        if (g_prof) g_prof[7] += __builtin_readcyclecounter() - t7_;
#endif
        if (!__ballot(!done)) break;
    }
    bool bad = mine && b.bitPos != 0;                  // EndOfDStream (BitStream.cs:494): every bit consumed, none over-read
    if (useX4 && __ballot(bad)) { if (bad) bad = !hufX4TailAccepts(L.huf, dtLog, src, size, n, out); }
    return !__ballot(bad);
}

// XXH64 seed 0 (XxHash.cs:896-1161), single lane
__device__ static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ static uint64_t xxh64(const uint8_t *p, uint64_t len)
{
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    const uint8_t *const bEnd = p + len; uint64_t h64;
    #define XXR(acc, in) { acc += (in) * P2; acc = rotl64(acc, 31); acc *= P1; }
    if (len >= 32) {
        const uint8_t *const limit = bEnd - 32;
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        do { XXR(v1, zs_load64(p)); p += 8; XXR(v2, zs_load64(p)); p += 8; XXR(v3, zs_load64(p)); p += 8; XXR(v4, zs_load64(p)); p += 8; } while (p <= limit);
        h64 = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        #define XXM(v) { uint64_t t_ = 0; XXR(t_, v); h64 ^= t_; h64 = h64 * P1 + P4; }
        XXM(v1); XXM(v2); XXM(v3); XXM(v4);
        #undef XXM
    } else h64 = P5;
    h64 += len;
    while (p + 8 <= bEnd) { uint64_t k1 = 0; XXR(k1, zs_load64(p)); h64 ^= k1; h64 = rotl64(h64, 27) * P1 + P4; p += 8; }
    if (p + 4 <= bEnd) { h64 ^= (uint64_t)zs_load32(p) * P1; h64 = rotl64(h64, 23) * P2 + P3; p += 4; }
    while (p < bEnd) { h64 ^= (*p) * P5; h64 = rotl64(h64, 11) * P1; p++; }
    #undef XXR
    h64 ^= h64 >> 33; h64 *= P2; h64 ^= h64 >> 29; h64 *= P3; h64 ^= h64 >> 32;
    return h64;
}

// The same by FOUR lanes (a quad of the wavefront, r = lane & 3): the stripe loop is four independent accumulators, lane r runs the r-th (the 8 bytes at 8 r of
// every 32-byte stripe), the quad's first lane merges them and finishes the tail.  Every lane of the quad must call; the result is valid in its first lane.
// (k_dec_checksum hashed an item's whole output on one lane: 1 MiB frames are 32768 dependent rounds there.)
__device__ __forceinline__ uint64_t xxh64_quad(const uint8_t *p, uint64_t len)
{
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    const uint32_t r = (uint32_t)zs_lane() & 3u;
    const uint8_t *const bEnd = p + len; uint64_t h64 = P5;
    #define XXR(acc, in) { acc += (in) * P2; acc = rotl64(acc, 31); acc *= P1; }
    const uint64_t stripes = len >> 5;
    if (stripes) {
        uint64_t v = r == 0 ? P1 + P2 : (r == 1 ? P2 : (r == 2 ? 0ull : 0ull - P1));
        const uint8_t *q = p + 8u * r;
        for (uint64_t i = 0; i < stripes; i++) { XXR(v, zs_load64(q)); q += 32; }
        const int base = zs_lane() & ~3;
        uint64_t vv[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) vv[k] = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), base + k) << 32) | (uint32_t)__shfl((int)(uint32_t)v, base + k);
        h64 = rotl64(vv[0], 1) + rotl64(vv[1], 7) + rotl64(vv[2], 12) + rotl64(vv[3], 18);
        #pragma unroll
        for (int k = 0; k < 4; k++) { uint64_t t_ = 0; XXR(t_, vv[k]); h64 ^= t_; h64 = h64 * P1 + P4; }
        p += stripes << 5;
    }
    h64 += len;
    while (p + 8 <= bEnd) { uint64_t k1 = 0; XXR(k1, zs_load64(p)); h64 ^= k1; h64 = rotl64(h64, 27) * P1 + P4; p += 8; }
    if (p + 4 <= bEnd) { h64 ^= (uint64_t)zs_load32(p) * P1; h64 = rotl64(h64, 23) * P2 + P3; p += 4; }
    while (p < bEnd) { h64 ^= (*p) * P5; h64 = rotl64(h64, 11) * P1; p++; }
    #undef XXR
    h64 ^= h64 >> 33; h64 *= P2; h64 ^= h64 >> 29; h64 *= P3; h64 ^= h64 >> 32;
    return h64;
}

struct DState { uint32_t rep[3]; uint32_t litEntropy, fseEntropy; uint32_t llRepeatOk; uint32_t hufX4; };   // hufX4: the reference built the current Huffman table for its double-symbol decoder

// ---- one tile of <= 64 decoded sequences (L.u.sq.tile*) -> output bytes.  Returns 0 or an error; advances op / litPos. ----
// DICT: a dictionary's content is the segment in front of the frame (RefDictContent :2366, CheckContinuity :1911): offsets may reach
// dictSize bytes beyond the frame's start (:1290-1315); dictEnd = one past the content's last byte.
template <bool DICT>
__device__ __forceinline__ uint32_t execTileT(const uint32_t *tileLL, const uint32_t *tileML, const uint32_t *tileOff, uint32_t T, uint8_t *dstBase,
                                              uint64_t frameStart, uint64_t oend, const uint8_t *litPtr, uint32_t litSize, uint64_t &op, uint32_t &litPos,
                                              const uint8_t *dictEnd, uint32_t dictSize)
{
    const uint32_t lane = (uint32_t)zs_lane();
    // execute the tile (ExecSequence :1265-1352).  Lane t owns sequence t: output positions by prefix sums, the checks
    // of the reference in its order (the first failing sequence decides), then all literal runs at once, then all
    // matches whose source lies before this tile's output at once, then the matches that read this tile's own output
    // one after the other.
    {
        const uint32_t ll = (lane < T) ? tileLL[lane] : 0u, ml = (lane < T) ? tileML[lane] : 0u, off = (lane < T) ? tileOff[lane] : 0u;
        const uint32_t incl = wave_incl_scan(ll + ml), inclL = wave_incl_scan(ll);
        const uint64_t outStart64 = op + (incl - ll - ml);            // where my literals go
        const uint32_t litStart = litPos + (inclL - ll);
        uint32_t err = 0;
        if (lane < T) {
            if ((uint64_t)ll + ml > oend - outStart64 || outStart64 > oend) err = E_dstSize_tooSmall;
            else if (ll > litSize - litStart || litStart > litSize) err = E_corruption_detected;
            else if (off > outStart64 + ll - frameStart + (DICT ? dictSize : 0u)) err = E_corruption_detected;
        }
        const uint64_t em = __ballot(err != 0);
        if (em) return ZE(wave_get(err, __builtin_ctzll(em)));
        // past the checks every position of the tile lies inside the item's output, whose capacity is a 32-bit count
        // (ZStdDecompress.cs:2182, size_t = UInt32): positions are 32-bit offsets from dstBase from here on
        const uint32_t outStart = (uint32_t)outStart64, mdst = outStart + ll, tileStart = (uint32_t)op, oend32 = (uint32_t)oend;
        // literals.  Runs of <= 16 bytes by their own lane (two 8-byte loads), longer runs by the whole wavefront, eight runs at a
        // time; all the loads of a round are issued before its stores (a run at a time is a memory round trip per run; ~6 long
        // runs a tile on the log data), and the first round of long runs shares its round trip with the short ones.
        {
            uint64_t lm = __ballot(ll > 16);
#ifndef ZS_EXEC_LR
#define ZS_EXEC_LR 8
#endif
            constexpr uint32_t LR = ZS_EXEC_LR;
            uint32_t l2[LR], s2[LR], d2[LR]; uint8_t v[LR];
            auto take = [&]() {
                #pragma unroll
                for (uint32_t k = 0; k < LR; k++) {
                    l2[k] = 0; s2[k] = 0; d2[k] = 0;
                    if (lm) { const int t = __builtin_ctzll(lm); lm &= lm - 1; l2[k] = wave_get(ll, t); s2[k] = wave_get(litStart, t); d2[k] = wave_get(outStart, t); }
                }
            };
            auto loads = [&]() {
                #pragma unroll
                for (uint32_t k = 0; k < LR; k++) v[k] = (lane < l2[k]) ? litPtr[s2[k] + lane] : (uint8_t)0;
            };
            auto stores = [&]() {
                #pragma unroll
                for (uint32_t k = 0; k < LR; k++) if (lane < l2[k]) dstBase[d2[k] + lane] = v[k];
                #pragma unroll
                for (uint32_t k = 0; k < LR; k++) for (uint32_t j = 64 + lane; j < l2[k]; j += 64) dstBase[d2[k] + j] = litPtr[s2[k] + j];
            };
            take();
            const bool shortRun = ll && ll <= 16, wide = shortRun && (litStart + 16 <= litSize);
            uint64_t a = 0, c = 0;
            if (wide) { a = zs_load64(litPtr + litStart); c = zs_load64(litPtr + litStart + 8); }
            loads();
            if (shortRun) {
                uint8_t *dp = dstBase + outStart;
                if (wide) {
                    if ((ll + ml >= 16 && oend32 - outStart >= 16) || ll == 16) {
                        // 16 bytes at once: what runs past the literals lands in this sequence's own match bytes, written later
                        __builtin_memcpy(dp, &a, 8); __builtin_memcpy(dp + 8, &c, 8);
                    } else {
                        // exactly ll bytes in at most four pieces (8, 4, 2, 1 by the bits of ll), not a store per byte
                        uint64_t w = a; uint32_t at = 0;
                        if (ll & 8) { __builtin_memcpy(dp, &w, 8); w = c; at = 8; }
                        if (ll & 4) { const uint32_t x = (uint32_t)w; __builtin_memcpy(dp + at, &x, 4); w >>= 32; at += 4; }
                        if (ll & 2) { const uint16_t x = (uint16_t)w; __builtin_memcpy(dp + at, &x, 2); w >>= 16; at += 2; }
                        if (ll & 1) dp[at] = (uint8_t)w;
                    }
                } else for (uint32_t j = 0; j < ll; j++) dp[j] = litPtr[litStart + j];
            }
            stores();
            while (lm) { take(); loads(); stores(); }
        }
        // matches reading only output that existed before this tile
        const uint32_t msrc = mdst - off;                             // (wraps for a match that starts in the dictionary: not used then)
        const bool inDict = DICT && ml && off > mdst - (uint32_t)frameStart;
        const bool indep = ml && !inDict && (msrc + ml <= tileStart);
        if (indep && ml <= 32) {
            if (ml >= 8) {
                // whole 8-byte pieces, the last one moved back so that it ends with the match (source and destination do not overlap)
                uint64_t v[4]; const uint32_t lastAt = ml - 8;
                #pragma unroll
                for (uint32_t k = 0; k < 4; k++) v[k] = zs_load64(dstBase + msrc + min(8 * k, lastAt));
                #pragma unroll
                for (uint32_t k = 0; k < 4; k++) if (8 * k < ml) __builtin_memcpy(dstBase + mdst + min(8 * k, lastAt), &v[k], 8);
            } else if (ml >= 4) {
                // 4 .. 7 bytes: two 4-byte pieces, the second ending with the match (a byte loop is a memory round trip per byte)
                const uint32_t x0 = zs_load32(dstBase + msrc), x1 = zs_load32(dstBase + msrc + ml - 4);
                __builtin_memcpy(dstBase + mdst, &x0, 4); __builtin_memcpy(dstBase + mdst + ml - 4, &x1, 4);
            } else for (uint32_t j = 0; j < ml; j++) dstBase[mdst + j] = dstBase[msrc + j];
        }
        for (uint64_t lm = __ballot(indep && ml > 32); lm; lm &= lm - 1) {
            const int t = __builtin_ctzll(lm);
            const uint32_t m2 = wave_get(ml, t);
            const uint32_t s2 = wave_get(msrc, t), d2 = wave_get(mdst, t);
            for (uint32_t j = lane; j < m2; j += 64) dstBase[d2 + j] = dstBase[s2 + j];
        }
        wave_mem_sync();
        // matches reading this tile's own output (earlier sequences are complete by then): in order, a group at a time.  A group
        // is a run of such matches none of which reads what the group writes -- destinations ascend, so that is: every source ends
        // at or before the FIRST member's destination.  Its members copy side by side (a lane each, like the matches above); only
        // the first member may overlap its own destination (offset < length: the periodic copy).  One memory round trip a group
        // instead of one a match (26 such matches in ~8 groups per tile on the log data).
        for (uint64_t rem = __ballot(ml && !indep); rem; ) {
            const int g0 = __builtin_ctzll(rem);
            const uint32_t lo = wave_get(mdst, g0);
            const uint64_t viol = __ballot(((rem >> lane) & 1ull) && (int)lane > g0 && (inDict || msrc + ml > lo));     // a dictionary match only ever leads a group
            const uint64_t grp = viol ? (rem & ((1ull << __builtin_ctzll(viol)) - 1ull)) : rem;
            const bool in = (grp >> lane) & 1ull;
            const bool self = in && (inDict || off < ml);            // only lane g0 can be
            if (in && !self && ml <= 32) {
                if (ml >= 8) {
                    uint64_t v[4]; const uint32_t lastAt = ml - 8;
                    #pragma unroll
                    for (uint32_t k = 0; k < 4; k++) v[k] = zs_load64(dstBase + msrc + min(8 * k, lastAt));
                    #pragma unroll
                    for (uint32_t k = 0; k < 4; k++) if (8 * k < ml) __builtin_memcpy(dstBase + mdst + min(8 * k, lastAt), &v[k], 8);
                } else if (ml >= 4) {
                    const uint32_t x0 = zs_load32(dstBase + msrc), x1 = zs_load32(dstBase + msrc + ml - 4);
                    __builtin_memcpy(dstBase + mdst, &x0, 4); __builtin_memcpy(dstBase + mdst + ml - 4, &x1, 4);
                } else for (uint32_t j = 0; j < ml; j++) dstBase[mdst + j] = dstBase[msrc + j];
            }
            for (uint64_t lm = __ballot(in && (self || ml > 32)); lm; lm &= lm - 1) {
                const int t = __builtin_ctzll(lm);
                const uint32_t m2 = wave_get(ml, t), o2 = wave_get(off, t);
                const uint32_t s2 = wave_get(msrc, t), d2 = wave_get(mdst, t);
                if (DICT && wave_get(inDict ? 1u : 0u, t)) {
                    // the first `beyond` bytes come from the end of the dictionary, the rest from the start of the frame (:1295-1315)
                    const uint32_t fs = (uint32_t)frameStart, beyond = o2 - (d2 - fs);
                    if (m2 <= o2) { for (uint32_t j = lane; j < m2; j += 64) dstBase[d2 + j] = (j < beyond) ? dictEnd[(int32_t)j - (int32_t)beyond] : dstBase[fs + (j - beyond)]; }
                    else if (lane == 0) { for (uint32_t j = 0; j < m2; j++) dstBase[d2 + j] = (j < beyond) ? dictEnd[(int32_t)j - (int32_t)beyond] : dstBase[fs + (j - beyond)]; }
                }
                else if (o2 >= m2) { for (uint32_t j = lane; j < m2; j += 64) dstBase[d2 + j] = dstBase[s2 + j]; }
                else { for (uint32_t j = lane; j < m2; j += 64) dstBase[d2 + j] = dstBase[s2 + (j % o2)]; }      // period = offset
            }
            wave_mem_sync();
            rem &= ~grp;
        }
        op += wave_last(incl);
        litPos += wave_last(inclL);
    }
    return 0;
}

// A fast-path table cell in 16 bits (half the LDS of a 4-byte cell = twice the items a CU decodes at once): the symbol in
// bits 0-5; above it 1 << (9 - nbBits) | (nextState >> nbBits).  nextState is a multiple of 2^nbBits, one of 2^(tableLog -
// nbBits) (FseDecompress.cs:111-181), and tableLog <= 9: the marker is the highest bit set and gives nbBits back.
__device__ __forceinline__ uint32_t zs_fastcell(uint32_t next, uint32_t nb, uint32_t sym) { return ((((1u << (9u - nb)) | (next >> nb)) << 6) | sym); }

// ---- sequence headers (DecodeSeqHeaders :1110-1180): number of sequences, the three tables.  ip / remaining move past them.
//      Returns 0 or an error code. ----
//      EMIT (k_dec_prep): each table is built in L.LL, turned into 16-bit fast-path cells and stored to stab (LL at cell 0, OF at
//      512, ML at 768) before the next one takes its place; logsOut[t] = its tableLog.
template <bool EMIT>
__device__ __forceinline__ uint32_t seqHeadersT(DLds &L, const DState &st, const uint8_t *&ip, uint32_t &remaining, uint32_t &nbSeq, uint16_t *stab, uint32_t *logsOut, const uint16_t *stabPrev = nullptr)
{
    const uint32_t lane = (uint32_t)zs_lane();
    {
        if (remaining < 1) return ZE(E_srcSize_wrong);
        const uint8_t *const iend = ip + remaining;
        nbSeq = *ip++;
        if (nbSeq) {
            if (nbSeq > 0x7F) {
                if (nbSeq == 0xFF) { if (ip + 2 > iend) return ZE(E_srcSize_wrong); nbSeq = rd16(ip) + 0x7F00; ip += 2; }
                else { if (ip >= iend) return ZE(E_srcSize_wrong); nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
            }
            if (ip + 4 > iend) return ZE(E_srcSize_wrong);
            const uint32_t modes = *ip++;
            uint32_t consumed = 0;
            for (int t = 0; t < 3; t++) {
                const uint32_t type = (modes >> (6 - 2 * t)) & 3;
                const uint32_t maxS = t == 0 ? 35 : (t == 1 ? 31 : 52), maxLog = t == 1 ? 8 : 9;
                SeqSym *cells = (EMIT || t == 0) ? L.LL.cells : (t == 1 ? L.OF.cells : L.ML.cells);
                uint32_t *tl = (EMIT || t == 0) ? &L.LL.tableLog : (t == 1 ? &L.OF.tableLog : &L.ML.tableLog);
                const int16_t *dn = t == 0 ? d_LL_defaultNorm : (t == 1 ? d_OF_defaultNorm : d_ML_defaultNorm);
                const uint32_t dmax = t == 0 ? 35 : (t == 1 ? 28 : 52);
                PPROF(L, 7);
                hw_stage(L.u.tb.hdrWin, ip + consumed, (uint32_t)(iend - (ip + consumed)));           // this table's description, staged
                PPROF(L, 8);
                if (lane == 0) {                                   // parse (serial, small): what to build and how many bytes it took
                    uint32_t err = 0, adv = 0, bmax = 0, blog = 0;
                    const uint8_t *p = ip + consumed;
                    const uint32_t left = (uint32_t)(iend - p);
                    if (type == 1) {
                        if (!left) err = ZE(E_srcSize_wrong);
                        else {
                            const uint32_t symbol = *p;
                            if (symbol > maxS) err = ZE(E_corruption_detected);
                            else { *tl = 0; cells[0].nbBits = 0; cells[0].nextState = 0; cells[0].sym = (uint8_t)symbol; adv = 1; }
                        }
                    } else if (type == 0) { bmax = dmax; blog = t == 1 ? 5 : 6; }
                    else if (type == 3) { if (!st.fseEntropy) err = ZE(E_corruption_detected); }
                    else {
                        uint32_t tableLog = 0, max = maxS;
                        const uint32_t h = readNCount(L.u.tb.norm, &max, &tableLog, L.u.tb.hdrWin, left, 0);
                        if (isErr(h) || tableLog > maxLog) err = ZE(E_corruption_detected);
                        else { bmax = max; blog = tableLog; adv = h; }
                    }
                    L.misc[0] = err; L.misc[1] = adv; L.misc[3] = bmax; L.misc[4] = blog;
                }
                wave_sync();
                PPROF(L, 9);
                if (L.misc[0]) return ZE(E_corruption_detected);
                consumed += L.misc[1];
                const uint32_t bmax = L.misc[3], blog = L.misc[4];
                if (type == 0) { if (lane <= dmax) L.u.tb.norm[lane] = dn[lane]; wave_sync(); }
                if (type == 0 || type == 2) buildSeqTableWave(L, cells, tl, bmax, blog);
                wave_sync();
                PPROF(L, 10);
                if (EMIT && type == 3) {
                    // a table repeated from the last block that had sequences: its 16-bit cells copied from that block's slot (logsOut[t] holds its log)
                    const uint32_t log = logsOut[t], at = t == 0 ? 0u : (t == 1 ? 512u : 768u);
                    if (!stabPrev) return ZE(E_corruption_detected);
                    wave_mem_sync();                                   // (the cells were stored by this wavefront, a block earlier)
                    for (uint32_t i = lane; i < (1u << log); i += 64) stab[at + i] = stabPrev[at + i];
                    wave_sync();
                } else if (EMIT) {
                    const uint32_t log = *tl, at = t == 0 ? 0u : (t == 1 ? 512u : 768u);
                    for (uint32_t i = lane; i < (1u << log); i += 64) { const SeqSym c = cells[i]; stab[at + i] = (uint16_t)zs_fastcell(c.nextState, c.nbBits, c.sym); }
                    if (lane == 0) logsOut[t] = log;
                    wave_sync();
                    PPROF(L, 11);
                }
            }
            ip += consumed;
        }
        remaining = (uint32_t)(iend - ip);                             // (also for a block without sequences: round 3 left it at the section's size there, and k_dec_prep, which
                                                                       //  wants nothing behind a zero count, sent every literals-only block to the general kernel: 1 % of ELF-class frames)
    }
    return 0;
}
__device__ __forceinline__ uint32_t seqHeaders(DLds &L, const DState &st, const uint8_t *&ip, uint32_t &remaining, uint32_t &nbSeq)
{ return seqHeadersT<false>(L, st, ip, remaining, nbSeq, nullptr, nullptr); }

// ---- one compressed block (ZSTD_decompressBlock_internal :1868-1909). returns decoded size or error ----

template <bool DICT>
__device__ __forceinline__ uint32_t decodeBlock(DLds &L, DState &st, uint8_t *dstBase, uint64_t frameStart, uint64_t op, uint64_t oend,
                                       const uint8_t *src, uint32_t srcSize, uint8_t *litBuf, uint64_t windowSize, uint64_t *g_prof,
                                       const uint8_t *dictEnd, uint32_t dictSize)
{
    const uint32_t lane = (uint32_t)zs_lane();
    PROF_T0(); (void)g_prof;
    if (srcSize >= (1u << 17)) return ZE(E_srcSize_wrong);
    if (srcSize < 3) return ZE(E_corruption_detected);
    // ---- literals (DecodeLiteralsBlock :683-821) ----
    const uint8_t *litPtr; uint32_t litSize, litCSizeTot;
    {
        const uint32_t type = src[0] & 3, lhl = (src[0] >> 2) & 3;
        if (type >= 2) {
            if (type == 3 && st.litEntropy == 0) return ZE(E_dictionary_corrupted);
            if (srcSize < 5) return ZE(E_corruption_detected);
            const uint32_t lhc = rd32(src);
            uint32_t lhSize, litCSize; bool single = false;
            if (lhl < 2) { single = !lhl; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; }
            else if (lhl == 2) { lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; }
            else { lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((uint32_t)src[4] << 10); }
            if (litSize > (1u << 17)) return ZE(E_corruption_detected);
            if (litCSize + lhSize > srcSize) return ZE(E_corruption_detected);
            const uint8_t *cs = src + lhSize; uint32_t csz = litCSize;
            if (type == 2) {
                if (!single && litSize == 0) return ZE(E_corruption_detected);
                if (!single && litCSize == 0) return ZE(E_corruption_detected);
                st.hufX4 = single ? 0u : hufSelectDecoder(litSize, litCSize);      // ZStdDecompress.cs:737 (1 stream: X2) / HufDecompress.cs:1208-1220
                const uint32_t h = readHufTable(L, cs, csz);
#ifdef ZS_DEC_PROFILE
                { const uint64_t now_ = __builtin_readcyclecounter(); if (g_prof) g_prof[6] += now_ - prof_t_; }
#endif
#ifdef ZS_DEC_ERRLINE
                if (isErr(h)) return h;
#endif
                if (isErr(h)) return ZE(E_corruption_detected);
                if (h >= csz) return ZE(E_corruption_detected);
                cs += h; csz -= h;
            }
            // one call site (the function is inlined: a call would turn its stores into flat stores, whose completion
            // every LDS read of the symbol loop would then wait for)
            uint32_t nStreams = 1, sOff = 0, sLen = csz, sCnt = litSize, sOut = 0;
            if (!single) {
                if (csz < 10) return ZE(E_corruption_detected);
                const uint32_t l1 = rd16(cs), l2 = rd16(cs + 2), l3 = rd16(cs + 4);
                if (l1 + l2 + l3 + 6 > csz) return ZE(E_corruption_detected);
                const uint32_t l4 = csz - (l1 + l2 + l3 + 6);
                const uint32_t seg = (litSize + 3) / 4;
                if (3 * seg > litSize) return ZE(E_corruption_detected);
                const uint32_t sl = min(lane, 3u);
                nStreams = 4;
                sOff = 6 + (sl > 0 ? l1 : 0) + (sl > 1 ? l2 : 0) + (sl > 2 ? l3 : 0);
                sLen = sl == 0 ? l1 : (sl == 1 ? l2 : (sl == 2 ? l3 : l4));
                sCnt = sl < 3 ? seg : litSize - 3 * seg;
                sOut = sl * seg;
            }
            const bool ok = hufDecodeStreams(L, nStreams, litBuf + sOut, sCnt, cs + sOff, sLen, st.hufX4 != 0, g_prof);
            if (__ballot(!ok)) return ZE(E_corruption_detected);
            wave_mem_sync();
            litPtr = litBuf; st.litEntropy = 1; litCSizeTot = litCSize + lhSize;
        } else {
            uint32_t lhSize;
            if (lhl == 1) { lhSize = 2; litSize = rd16(src) >> 4; }
            else if (lhl == 3) { lhSize = 3; litSize = rd24(src) >> 4; }
            else { lhSize = 1; litSize = src[0] >> 3; }
            if (type == 0) {
                if (litSize + lhSize > srcSize) return ZE(E_corruption_detected);
                litPtr = src + lhSize; litCSizeTot = lhSize + litSize;
            } else {
                if (lhl == 3 && srcSize < 4) return ZE(E_corruption_detected);
                if (litSize > (1u << 17)) return ZE(E_corruption_detected);
                const uint8_t v = src[lhSize];
                for (uint32_t j = lane; j < litSize; j += 64) litBuf[j] = v;
                wave_mem_sync();
                litPtr = litBuf; litCSizeTot = lhSize + 1;
            }
        }
    }
    PROF_ADD(0);
    const uint8_t *ip = src + litCSizeTot;
    uint32_t remaining = srcSize - litCSizeTot;
    uint32_t nbSeq;
    { const uint32_t e = seqHeaders(L, st, ip, remaining, nbSeq); if (e) return e; }
    PROF_ADD(1);
    // ---- sequences (decompressSequences_body :1555-1608) ----
    const uint64_t ostart = op;
    uint32_t litPos = 0;
    if (nbSeq) {
        st.fseEntropy = 1;
        // the bitstream is read by lane 0 from an LDS window (stageWindow); one tile of 64 sequences reads < 720 bytes of it
        BitC b; b.c = 0; b.avail = 0; b.bitPos = 0;
        uint32_t sLL = 0, sOF = 0, sML = 0;
        uint32_t rep0 = st.rep[0], rep1 = st.rep[1], rep2 = st.rep[2];
        if (!bc_init(b, ip, remaining)) return ZE(E_corruption_detected);             // same bytes for every lane: uniform
        int32_t base = bc_windowBase(b, ZS_SEQWIN);
        stageWindow(L.u.sq.win, ip, remaining, base, ZS_SEQWIN);
        wave_sync();
        // one availability check per group of reads (the container holds >= 57 bits after a refill)
        #define SEQ_NEED(nbits) do { if (b.avail < (nbits)) bc_refill(b, L.u.sq.win, base); } while (0)
        if (lane == 0) { SEQ_NEED(L.LL.tableLog + L.OF.tableLog + L.ML.tableLog); sLL = bc_take(b, L.LL.tableLog); sOF = bc_take(b, L.OF.tableLog); sML = bc_take(b, L.ML.tableLog); }
        uint32_t left = nbSeq;
        while (left) {
            const uint32_t T = min(64u, left);
            {   // keep a tile's worth of stream below the cursor inside the window
                const int32_t bp0 = (int32_t)wave_get((uint32_t)b.bitPos, 0);
                const int32_t bh = (bp0 > 0) ? ((bp0 - 1) >> 3) : 0;
                if (base > 0 && bh < base + 736) {
                    wave_sync();
                    base = (bh + 1 > (int32_t)ZS_SEQWIN) ? bh + 1 - (int32_t)ZS_SEQWIN : 0;
                    stageWindow(L.u.sq.win, ip, remaining, base, ZS_SEQWIN);
                    wave_sync();
                }
            }
            if (lane == 0) {
                uint32_t bad = 0;
                for (uint32_t t = 0; t < T; t++) {
                    if (b.bitPos < 0) { bad = 1; break; }            // stream exhausted before all sequences (:1582, :1594)
                    const SeqSym eLL = L.LL.cells[sLL], eOF = L.OF.cells[sOF], eML = L.ML.cells[sML];
                    const uint32_t tLL = L.llTab[eLL.sym], tML = L.mlTab[eML.sym];
                    const uint32_t llBase = tLL & 0xFFFFFFu, llAdd = tLL >> 24, mlBase = tML & 0xFFFFFFu, mlAdd = tML >> 24, ofAdd = eOF.sym;
                    uint32_t offset, ml, ll;
                    if (ofAdd + mlAdd + llAdd <= 57u) {               // the usual case: all extra bits from one container
                        SEQ_NEED(ofAdd + mlAdd + llAdd);
                        offset = ofBaseOf(ofAdd) + bc_take(b, ofAdd);      // ofBaseOf(0) = 0 and 0 bits read: offset code 0
                        ml = mlBase + bc_take(b, mlAdd);
                        ll = llBase + bc_take(b, llAdd);
                    } else {
                        SEQ_NEED(ofAdd); offset = ofBaseOf(ofAdd) + bc_take(b, ofAdd);
                        SEQ_NEED(mlAdd + llAdd); ml = mlBase + bc_take(b, mlAdd); ll = llBase + bc_take(b, llAdd);
                    }
                    if (ofAdd <= 1) {                                  // recent offsets (:1509-1530)
                        offset += (llBase == 0);
                        if (offset) {
                            uint32_t temp = (offset == 3) ? rep0 - 1 : (offset == 1 ? rep1 : rep2);
                            temp += !temp;
                            if (offset != 1) rep2 = rep1;
                            rep1 = rep0; rep0 = offset = temp;
                        } else offset = rep0;
                    } else { rep2 = rep1; rep1 = rep0; rep0 = offset; }
                    SEQ_NEED((uint32_t)eLL.nbBits + eML.nbBits + eOF.nbBits);
                    sLL = eLL.nextState + bc_take(b, eLL.nbBits);
                    sML = eML.nextState + bc_take(b, eML.nbBits);
                    sOF = eOF.nextState + bc_take(b, eOF.nbBits);
                    L.u.sq.tileLL[t] = ll; L.u.sq.tileML[t] = ml; L.u.sq.tileOff[t] = offset;
                }
                L.misc[0] = bad;
            }
            wave_sync();
            PROF_ADD(2);
            if (L.misc[0]) return ZE(E_corruption_detected);
            { const uint32_t e = execTileT<DICT>(L.u.sq.tileLL, L.u.sq.tileML, L.u.sq.tileOff, T, dstBase, frameStart, oend, litPtr, litSize, op, litPos, dictEnd, dictSize); if (e) return e; }
            PROF_ADD(3);
            left -= T;
        }
        #undef SEQ_NEED
        st.rep[0] = wave_get(rep0, 0); st.rep[1] = wave_get(rep1, 0); st.rep[2] = wave_get(rep2, 0);
    }
    {
        const uint32_t lastLL = litSize - litPos;
        if (lastLL > oend - op) return ZE(E_dstSize_tooSmall);
        for (uint32_t j = lane; j < lastLL; j += 64) dstBase[op + j] = litPtr[litPos + j];
        op += lastLL;
    }
    wave_mem_sync();
    return (uint32_t)(op - ostart);
}

#ifndef ZS_DEC_GROUP
#define ZS_DEC_GROUP 2             // items (= wavefronts) per workgroup
#endif
// every synchronisation inside an item is wavefront-local (wave_sync): the wavefronts of a workgroup run independently.
// Every function that touches the LDS workspace is force-inlined: through a call the workspace reference becomes a generic
// pointer and its accesses flat_* instructions, which complete out of order with the ds_* accesses of the inlined code.
// DICT: every frame of every item is decoded with the dictionary dict[0 .. dictBytes) (ZSTD_decompress_usingDict :2162): raw
// content, or a formatted dictionary (magic 0xEC30A437) whose entropy tables and recent offsets are loaded in front of each frame
// (ZSTD_decompressBegin_usingDict :2501, LoadEntropy :2378-2450) -- by every wavefront for itself: a dictionary is a few KiB.
#define ZS_DEC_LITBUF ((1u << 17) + 64u)                 // a wavefront's literal buffer: the largest block + slack
template <bool DICT>
__device__ __forceinline__ void zs_decode_item(DLds &L, const uint32_t item, const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint8_t *dstAll,
                                               uint32_t *__restrict__ dstSizes, uint8_t *litBuf, const uint8_t *__restrict__ dict, uint32_t dictBytes)
{
    const ZsDecItem it = items[item];
    const uint32_t lane = (uint32_t)zs_lane();
    const uint8_t *src = srcAll + it.srcOff;
    uint8_t *dstBase = dstAll + it.dstOff;
    uint64_t *g_prof = nullptr;
#ifdef ZS_DEC_PROFILE
    if (lane == 0) { g_prof = reinterpret_cast<uint64_t *>(litBuf + (1u << 17)); for (int k = 0; k < 8; k++) g_prof[k] = 0; }
    const uint64_t prof_start_ = __builtin_readcyclecounter();
#endif
    uint32_t srcSize = it.srcSize;
    uint64_t ipos = 0, op = 0;
    const uint64_t oend = it.dstCap;
    uint32_t result = 0;

    #define DONE(v) do { result = (v); goto finish; } while (0)
    while (srcSize - ipos >= 5) {                                   // DecompressMultiFrame :2111-2160
        const uint8_t *ip = src + ipos;
        uint64_t rem = srcSize - ipos;
        const uint32_t magic = rd32(ip);
        if (magic != 0xFD2FB528u) {
            if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
                if (rem < 8) DONE(ZE(E_srcSize_wrong));
                const uint64_t skip = (uint64_t)rd32(ip + 4) + 8;
                if (rem < skip) DONE(ZE(E_srcSize_wrong));
                ipos += skip; continue;
            }
            DONE(ZE(E_prefix_unknown));
        }
        // ---- frame header (:389-499, :2008-2031) ----
        if (rem < 6 + 3) DONE(ZE(E_srcSize_wrong));
        const uint32_t fhd = ip[4];
        const uint32_t dictIDCode = fhd & 3, checksumFlag = (fhd >> 2) & 1, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
        const uint32_t didSize = dictIDCode == 3 ? 4 : dictIDCode, fcsSize = fcsID == 0 ? 0 : (fcsID == 1 ? 2 : (fcsID == 2 ? 4 : 8));
        const uint32_t fhs = 5 + !singleSegment + didSize + fcsSize + (singleSegment && !fcsID);
        if (rem < fhs + 3) DONE(ZE(E_srcSize_wrong));
        if (fhd & 0x08) DONE(ZE(E_frameParameter_unsupported));
        uint32_t pos = 5; uint64_t windowSize = 0, fcs = ~0ull; uint32_t dictID = 0;
        if (!singleSegment) {
            const uint32_t wl = ip[pos++]; const uint32_t windowLog = (wl >> 3) + 10;
            if (windowLog > 30) DONE(ZE(E_frameParameter_windowTooLarge));
            windowSize = 1ull << windowLog; windowSize += (windowSize >> 3) * (wl & 7);
        }
        if (dictIDCode == 1) { dictID = ip[pos]; pos += 1; } else if (dictIDCode == 2) { dictID = rd16(ip + pos); pos += 2; } else if (dictIDCode == 3) { dictID = rd32(ip + pos); pos += 4; }
        if (fcsID == 0) { if (singleSegment) fcs = ip[pos]; } else if (fcsID == 1) fcs = rd16(ip + pos) + 256; else if (fcsID == 2) fcs = rd32(ip + pos); else fcs = zs_load64(ip + pos);
        if (singleSegment) windowSize = fcs;
        ipos += fhs;
        DState st; st.rep[0] = 1; st.rep[1] = 4; st.rep[2] = 8; st.litEntropy = 0; st.fseEntropy = 0; st.llRepeatOk = 0; st.hufX4 = 0;   // DecompressBegin :2478-2499
        const uint8_t *dictEnd = nullptr; uint32_t dictSize = 0, dictIDLoaded = 0;
        if (DICT && dict && dictBytes) {                             // ZSTD_decompress_insertDictionary :2452-2475
            const uint8_t *content = dict; uint32_t contentSize = dictBytes;
            if (dictBytes >= 8 && rd32(dict) == 0xEC30A437u) {
                dictIDLoaded = rd32(dict + 4);
                const uint8_t *p = dict + 8; const uint8_t *const pend = dict + dictBytes;
                if (dictBytes <= 8) DONE(ZE(E_dictionary_corrupted));
                { const uint32_t h = readHufTable(L, p, (uint32_t)(pend - p)); if (isErr(h)) DONE(ZE(E_dictionary_corrupted)); p += h; st.hufX4 = 1; }   // HUF_readDTableX4_wksp (:2391)
                for (int t = 0; t < 3; t++) {                         // offset codes, match lengths, literal lengths (:2395-2435)
                    const uint32_t maxS = t == 0 ? 31 : (t == 1 ? 52 : 35), maxLog = t == 0 ? 8 : 9;
                    SeqSym *cells = t == 0 ? L.OF.cells : (t == 1 ? L.ML.cells : L.LL.cells);
                    uint32_t *tl = t == 0 ? &L.OF.tableLog : (t == 1 ? &L.ML.tableLog : &L.LL.tableLog);
                    const uint32_t left = (uint32_t)(pend - p);
                    hw_stage(L.u.tb.hdrWin, p, left);
                    if (lane == 0) {
                        uint32_t tableLog = 0, max = maxS;
                        const uint32_t h = readNCount(L.u.tb.norm, &max, &tableLog, L.u.tb.hdrWin, left, 0);
                        L.misc[0] = (isErr(h) || max > maxS || tableLog > maxLog) ? 1u : 0u; L.misc[1] = h; L.misc[3] = max; L.misc[4] = tableLog;
                    }
                    wave_sync();
                    if (L.misc[0]) DONE(ZE(E_dictionary_corrupted));
                    const uint32_t adv = L.misc[1], bmax = L.misc[3], blog = L.misc[4];
                    buildSeqTableWave(L, cells, tl, bmax, blog);
                    wave_sync();
                    p += adv;
                }
                if (p + 12 > pend) DONE(ZE(E_dictionary_corrupted));
                contentSize = (uint32_t)(pend - (p + 12));
                for (int i = 0; i < 3; i++) { const uint32_t rep = rd32(p); p += 4; if (rep == 0 || rep >= contentSize) DONE(ZE(E_dictionary_corrupted)); st.rep[i] = rep; }
                st.litEntropy = 1; st.fseEntropy = 1;
                content = p;
            }
            dictEnd = content + contentSize; dictSize = contentSize;
        }
        if (dictID != 0 && dictID != dictIDLoaded) DONE(ZE(E_dictionary_wrong));      // :632-634
        const uint64_t frameStart = op;
        for (;;) {                                                   // block loop :2033-2067
            if (srcSize - ipos < 3) DONE(ZE(E_srcSize_wrong));
            const uint32_t bh = rd24(src + ipos);
            const uint32_t lastBlock = bh & 1, btype = (bh >> 1) & 3, cSize = bh >> 3;
            if (btype == 3) DONE(ZE(E_corruption_detected));
            const uint32_t cBlockSize = btype == 1 ? 1 : cSize;
            ipos += 3;
            if (cBlockSize > srcSize - ipos) DONE(ZE(E_srcSize_wrong));
            uint32_t decoded;
            if (btype == 2) {
                decoded = decodeBlock<DICT>(L, st, dstBase, frameStart, op, oend, src + ipos, cBlockSize, litBuf, windowSize, g_prof, dictEnd, dictSize);
                if (isErr(decoded)) DONE(decoded);
            } else if (btype == 0) {
                if (cBlockSize > oend - op) DONE(ZE(E_dstSize_tooSmall));
                zs_block_copy(dstBase + op, src + ipos, cBlockSize, lane, 64);    // raw block: 16 bytes a lane (a byte a lane was 2048 dependent rounds for 128 KiB)
                decoded = cBlockSize;
            } else {
                if (cSize > oend - op) DONE(ZE(E_dstSize_tooSmall));
                const uint8_t v = src[ipos];
                for (uint32_t j = lane; j < cSize; j += 64) dstBase[op + j] = v;
                decoded = cSize;
            }
            op += decoded; ipos += cBlockSize;
            wave_mem_sync();
            if (lastBlock) break;
        }
        if (fcs != ~0ull && (op - frameStart) != fcs) DONE(ZE(E_corruption_detected));
        if (checksumFlag) {
            if (srcSize - ipos < 4) DONE(ZE(E_checksum_wrong));
            { const uint64_t hq = xxh64_quad(dstBase + frameStart, op - frameStart); if (lane == 0) L.misc[2] = (uint32_t)hq; }     // (every lane calls: the quads hash the same bytes, lane 0's counts)
            wave_sync();
            if (rd32(src + ipos) != L.misc[2]) DONE(ZE(E_checksum_wrong));
            ipos += 4;
        }
    }
    if (srcSize != ipos) DONE(ZE(E_srcSize_wrong));
    result = (uint32_t)op;
finish:
#ifdef ZS_DEC_PROFILE
    if (g_prof) g_prof[5] = __builtin_readcyclecounter() - prof_start_;
#endif
    if (lane == 0) dstSizes[item] = result;
    #undef DONE
}

// The kernel: a POOL of wavefronts, each with its own literal buffer, takes the items one after the other from a queue (a counter in global
// memory, zeroed before the launch): what the buffers cost is set by the wavefronts the chip holds, not by the items of a call (round 3 reserved
// 128 KiB per ITEM: 7 GiB for 57344 frames of 32 KiB that the fast path had already decoded), and wavefronts that finish early take more.
// After the fast path the queue runs over the LIST of the items it left (k_dec_collect), so an atomic is spent per item to decode, not per item of the call.
__global__ void __launch_bounds__(256)
k_dec_collect(const uint32_t *__restrict__ doneFlags, uint32_t flagStride, uint32_t nItems, uint32_t *__restrict__ list, uint32_t *__restrict__ listCount)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    const bool todo = i < nItems && doneFlags[(size_t)i * flagStride] == 0;
    const uint64_t m = __ballot(todo);
    if (!m) return;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(listCount, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, __builtin_ctzll(m));
    if (todo) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}
template <int F, bool DICT>
__global__ void __launch_bounds__(64 * F)
k_decode_frames(const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, uint8_t *dstAll,
                uint32_t *__restrict__ dstSizes, uint8_t *__restrict__ litScratchAll, const uint32_t *__restrict__ list, const uint32_t *__restrict__ listCount,
                const uint8_t *__restrict__ dict, uint32_t dictBytes, uint32_t *__restrict__ queue)
{
    __shared__ DLds LS[F];
    const uint32_t total = list ? *listCount : nItems;                          // (list == nullptr: every item of the call)
    if (blockIdx.x * F >= total) return;                                        // nothing left for this workgroup (behind the fast path usually for all of them: the launch is then a few microseconds, not the 0.04 ms of 3072 wavefronts setting up)
    DLds &L = LS[threadIdx.x >> 6];
    const uint32_t lane = (uint32_t)zs_lane();
    uint8_t *litBuf = litScratchAll + (size_t)(blockIdx.x * F + (threadIdx.x >> 6)) * ZS_DEC_LITBUF;
    if (lane < 36) L.llTab[lane] = d_LL_base[lane] | ((uint32_t)d_LL_bits[lane] << 24);
    if (lane < 53) L.mlTab[lane] = d_ML_base[lane] | ((uint32_t)d_ML_bits[lane] << 24);
    wave_sync();
    for (;;) {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(queue, 1u);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at >= total) break;                                                 // (every wavefront gets here: the queue only grows)
        const uint32_t item = list ? list[at] : at;
        zs_decode_item<DICT>(L, item, srcAll, items, dstAll, dstSizes, litBuf, dict, dictBytes);
        wave_mem_sync();                                                        // the buffer and the LDS image are the next item's
    }
}
