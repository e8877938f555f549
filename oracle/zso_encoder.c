/*
 * ORACLE E -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * Scalar, single-threaded C statement of THIS REPO's Zstandard block encoder: the
 * algorithm that the HIP kernels in zstandard_amd/csrc run, written as plain loops so
 * that (a) the GPU output can be compared byte for byte with it and (b) bench.py can
 * time it on host cores as the "port" CPU baseline.
 *
 * The reference (epam/Zstandard) has NO encoder (Compress() is a
 * comment at csharp/src/ZStd.cs:89-96; FSE/bitstream writers are comments at
 * Fse.cs:548-592, BitStream.cs:103-115,234-309).  So nothing here restates reference
 * code; every stage is the format-inverse of a decoder function of the reference and
 * cites it.  Correctness of what it emits is pinned by oracle D (zso_decoder.c, the
 * restated reference decoder) decoding it back bit-exactly, and cross-checked with
 * upstream libzstd 1.4.8 in tests.  Ratio parity "vs the reference" is unpinned by the
 * reference (no encoder exists); the stated yardstick is libzstd level 3.
 *
 * Algorithm (one frame per chunk, blocks of <= 64 KiB; match search works on LZ UNITS of <= 128 KiB =
 * two consecutive blocks of a chunk, so the second block of a unit may copy from the first):
 *  1. candidates: two unit-wide hash tables (short: 5 bytes hashed, long: 8 bytes hashed, level >= 3 only),
 *     every position inserted in order, last writer keeps the slot, slots carry a 15-bit tag.  A position's
 *     candidate = previous owner of its long slot, else of its short slot (tag must agree).  -> dist[p]
 *  2. parse: each block is cut in walk ranges of 256 bytes (512 at levels <= 2); each is walked greedily and
 *     independently: the first LOOK candidate positions of a 9 .. 16-position window (stage-1 candidates, and repeats of
 *     the walker's two recent offsets) are scored, the best one becomes a sequence; matches may pass the range end
 *     by up to 1 KiB.
 *  3. stitch: ranges give up what an earlier range's match already covers; a range's first match that continues the
 *     match it meets (same offset, no gap) is joined to it, so a long match found piecewise is one sequence; the ranges
 *     are concatenated, offsets become repcodes through the decoder's 3-entry recent-offset list
 *     (inverse of ZStdDecompress.cs:1509-1530).
 *  4. literals: histogram, length-limited (11 bit) Huffman by package-merge, weights
 *     written direct or FSE-compressed (inverse of EntropyCommon.cs:198-269 and
 *     HufDecompress.cs:117-180), 1 or 4 streams (inverse of HufDecompress.cs:247-358).
 *  5. sequences: LL/ML/OF codes, normalised counts (inverse of EntropyCommon.cs:79-188),
 *     encoding tables (mirror of ZStdDecompress.cs:958-1034), one backward bitstream
 *     (inverse of ZStdDecompress.cs:1473-1608).
 *  6. block / frame assembly (inverse of ZStdDecompress.cs:421-499,646-659,683-821),
 *     raw / RLE block fallbacks.
 */
#include "zso_oracle.h"
#include <string.h>
#include <stdlib.h>

typedef uint8_t BYTE;
typedef uint16_t U16;
typedef int16_t S16;
typedef uint32_t U32;
typedef uint64_t U64;

#define ERR(code) ((size_t)0 - (size_t)(code))

/* ---- tunables (the HIP kernels are built with the same values) ---- */
#define BLOCK_MAX   65536u          /* bytes per block */
#define UNIT_MAX    131072u         /* bytes per LZ unit (match window): two blocks */
#define WALK_LOG_MIN 8              /* the walk cuts a block in ranges of 256 bytes (level >= 3) or 512 bytes (level <= 2) */
#define OUT_LOG     10              /* the ranges are handed on in groups of 1 KiB (the HIP walk kernel's output layout) */
#ifndef CROSS_MAX
#define CROSS_MAX   1024u           /* a match may pass its range's end by this much (and never the block's end) */
#endif
#define MINMATCH    5               /* shortest match kept; a recent-offset match may be 4 */
#define MAX_TABLE_LOG 14
#define HUF_MAXBITS 11
#define MaxLL 35
#define MaxML 52
#define MaxOff 31
#define LLFSELog 9
#define MLFSELog 9
#define OffFSELog 8

/* level <= 2 ("fast"): the short table only, walk ranges of 512 bytes, recent offsets tried on 8 positions;
 * level 3 ("double"): short + long table, ranges of 256 bytes, recent offsets tried on 4 positions, a step's window is 2 aligned groups
 * of 8 positions (levels <= 2: 4 groups);
 * level >= 4: as 3 with LOOK 8 and recent offsets on 8 positions. */
typedef struct { int useLong; int look; int walkLog; int repWin; int windowGroups; } EParams;
static EParams paramsForLevel(int level)
{
    EParams p;
    p.useLong = level >= 3; p.look = (level <= 3) ? 4 : 8; p.walkLog = (level <= 2) ? 9 : 8; p.repWin = (level == 3) ? 4 : 8;
    p.windowGroups = (level <= 2) ? 4 : 2;      /* = lanes of a GPU walker: a range of 512 bytes gets 4 lanes, one of 256 bytes 2 (the same threads per byte) */
    return p;
}

static U32 rd32(const BYTE *p) { U32 v; memcpy(&v, p, 4); return v; }
static void wr16(BYTE *p, U32 v) { p[0] = (BYTE)v; p[1] = (BYTE)(v >> 8); }
static void wr24(BYTE *p, U32 v) { wr16(p, v); p[2] = (BYTE)(v >> 16); }
static void wr32(BYTE *p, U32 v) { wr16(p, v); wr16(p + 2, v >> 16); }
static U32 highbit32(U32 v) { return 31 - (U32)__builtin_clz(v); }

/* ---- code tables (ZStdInternal.cs:158,173 ; ZStdDecompress.cs:1081,1100) ---- */
static const BYTE LL_bits[MaxLL + 1] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
static const BYTE ML_bits[MaxML + 1] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
                                         1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
static const U32 LL_base[MaxLL + 1] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,18,20,22,24,28,32,40,
                                        48,64,0x80,0x100,0x200,0x400,0x800,0x1000, 0x2000,0x4000,0x8000,0x10000 };
static const U32 ML_base[MaxML + 1] = { 3,4,5,6,7,8,9,10, 11,12,13,14,15,16,17,18, 19,20,21,22,23,24,25,26,
                                        27,28,29,30,31,32,33,34, 35,37,39,41,43,47,51,59, 67,83,99,0x83,0x103,0x203,0x403,0x803,
                                        0x1003,0x2003,0x4003,0x8003,0x10003 };
static const S16 LL_defaultNorm[MaxLL + 1] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
static const S16 ML_defaultNorm[MaxML + 1] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                               1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
static const S16 OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };

/* litLength / matchLength -> code : closed forms of the tables above */
static U32 llCodeOf(U32 ll)
{
    static const BYTE LL_Code[64] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,16,17,17,18,18,19,19, 20,20,20,20,21,21,21,21,
                                      22,22,22,22,22,22,22,22, 23,23,23,23,23,23,23,23, 24,24,24,24,24,24,24,24, 24,24,24,24,24,24,24,24 };
    return (ll > 63) ? highbit32(ll) + 19 : LL_Code[ll];
}
static U32 mlCodeOf(U32 mlBase)   /* mlBase = matchLength - 3 */
{
    static const BYTE ML_Code[128] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,17,18,19,20,21,22,23, 24,25,26,27,28,29,30,31,
                                       32,32,33,33,34,34,35,35, 36,36,36,36,37,37,37,37, 38,38,38,38,38,38,38,38, 39,39,39,39,39,39,39,39,
                                       40,40,40,40,40,40,40,40, 40,40,40,40,40,40,40,40, 41,41,41,41,41,41,41,41, 41,41,41,41,41,41,41,41,
                                       42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42 };
    return (mlBase > 127) ? highbit32(mlBase) + 36 : ML_Code[mlBase];
}

/* ======================================================================= *
 *  forward bit writer (read backward by BitStream.cs:322-494)
 * ======================================================================= */
typedef struct { U64 acc; U32 nbits; BYTE *ptr; BYTE *start; BYTE *end; int overflow; } BitW;
static void bw_init(BitW *b, BYTE *dst, size_t cap) { b->acc = 0; b->nbits = 0; b->ptr = b->start = dst; b->end = dst + cap; b->overflow = 0; }
static void bw_flush(BitW *b)
{
    while (b->nbits >= 8) {
        if (b->ptr < b->end) *b->ptr++ = (BYTE)b->acc; else b->overflow = 1;
        b->acc >>= 8; b->nbits -= 8;
    }
}
static void bw_add(BitW *b, U32 value, U32 nb)
{
    if (!nb) return;
    b->acc |= (U64)(value & ((nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1))) << b->nbits;
    b->nbits += nb;
    bw_flush(b);
}
/* end mark = a single 1 bit, then zero padding (BitStream.cs:337-339) */
static size_t bw_close(BitW *b)
{
    bw_add(b, 1, 1);
    if (b->nbits) { if (b->ptr < b->end) *b->ptr++ = (BYTE)b->acc; else b->overflow = 1; b->nbits = 0; }
    return b->overflow ? 0 : (size_t)(b->ptr - b->start);
}

/* ======================================================================= *
 *  FSE encoding side
 * ======================================================================= */
static U32 FSE_TABLESTEP(U32 tableSize) { return (tableSize >> 1) + (tableSize >> 3) + 3; }   /* Fse.cs:714 */

typedef struct { int deltaFindState; U32 deltaNbBits; } SymTT;
typedef struct { U32 tableLog; U16 stateTable[512]; SymTT tt[256]; } CTable;

/* Encoding table for a normalised distribution.  Cell order is the decoder's
 * (spread with the same step and low-probability area, ZStdDecompress.cs:993-1013 /
 * FseDecompress.cs:144-160): the encoder's state for (symbol, k-th occurrence) is the
 * decoder cell holding it. */
static void buildCTable(CTable *ct, const S16 *norm, U32 maxSymbolValue, U32 tableLog)
{
    U32 const tableSize = 1u << tableLog, tableMask = tableSize - 1, step = FSE_TABLESTEP(tableSize);
    BYTE tableSymbol[512];
    U32 cumul[258];
    U32 highThreshold = tableSize - 1, s, position = 0, u;
    ct->tableLog = tableLog;
    cumul[0] = 0;
    for (s = 1; s <= maxSymbolValue + 1; s++) {
        if (norm[s - 1] == -1) { cumul[s] = cumul[s - 1] + 1; tableSymbol[highThreshold--] = (BYTE)(s - 1); }
        else cumul[s] = cumul[s - 1] + (U32)norm[s - 1];
    }
    for (s = 0; s <= maxSymbolValue; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            tableSymbol[position] = (BYTE)s;
            position = (position + step) & tableMask;
            while (position > highThreshold) position = (position + step) & tableMask;
        }
    }
    for (u = 0; u < tableSize; u++) { BYTE const sym = tableSymbol[u]; ct->stateTable[cumul[sym]++] = (U16)(tableSize + u); }
    {
        U32 total = 0;
        for (s = 0; s <= maxSymbolValue; s++) {
            switch (norm[s]) {
            case 0: ct->tt[s].deltaNbBits = ((tableLog + 1) << 16) - (1u << tableLog); ct->tt[s].deltaFindState = 0; break;
            case -1:
            case 1: ct->tt[s].deltaNbBits = (tableLog << 16) - (1u << tableLog); ct->tt[s].deltaFindState = (int)total - 1; total++; break;
            default: {
                U32 const maxBitsOut = tableLog - highbit32((U32)norm[s] - 1);
                U32 const minStatePlus = (U32)norm[s] << maxBitsOut;
                ct->tt[s].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                ct->tt[s].deltaFindState = (int)total - norm[s];
                total += (U32)norm[s];
            } }
        }
    }
}
typedef struct { U32 value; const CTable *ct; int rle; } CState;
static void cstate_init(CState *st, const CTable *ct, U32 symbol, int rle)   /* first symbol costs no bits */
{
    st->ct = ct; st->rle = rle; st->value = 0;
    if (rle) return;
    {
        SymTT const tt = ct->tt[symbol];
        U32 const nbBitsOut = (tt.deltaNbBits + (1u << 15)) >> 16;
        U32 const v = (nbBitsOut << 16) - tt.deltaNbBits;
        st->value = ct->stateTable[(v >> nbBitsOut) + tt.deltaFindState];
    }
}
static void cstate_encode(BitW *b, CState *st, U32 symbol)
{
    if (st->rle) return;
    {
        SymTT const tt = st->ct->tt[symbol];
        U32 const nbBitsOut = (st->value + tt.deltaNbBits) >> 16;
        bw_add(b, st->value, nbBitsOut);
        st->value = st->ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
    }
}
static void cstate_flush(BitW *b, const CState *st) { if (!st->rle) bw_add(b, st->value, st->ct->tableLog); }

/* Normalise counts to sum 2^tableLog.  Every present symbol gets >= 1 (no -1 entries are
 * produced: a 1 costs the same bits and the decoder treats both as one cell).  Rounding
 * surplus/deficit goes to / comes from the largest entries, one unit at a time for a deficit. */
static void normalizeCounts(S16 *norm, U32 tableLog, const U32 *count, U32 total, U32 maxSymbolValue)
{
    U32 const tableSize = 1u << tableLog;
    int still = (int)tableSize;
    U32 s, largest = 0;
    for (s = 0; s <= maxSymbolValue; s++) {
        if (!count[s]) { norm[s] = 0; continue; }
        {
            U64 const scaled = (U64)count[s] * tableSize;
            U32 p = (U32)(scaled / total);
            U32 const rem = (U32)(scaled % total);
            if (2 * (U64)rem >= total) p++;          /* round to nearest, ties up */
            if (p == 0) p = 1;
            norm[s] = (S16)p;
            still -= (int)p;
            if (norm[s] > norm[largest] || !count[largest]) largest = s;
        }
    }
    if (still > 0) norm[largest] = (S16)(norm[largest] + still);
    while (still < 0) {
        U32 best = 0; int found = 0;
        for (s = 0; s <= maxSymbolValue; s++) if (norm[s] > 1 && (!found || norm[s] > norm[best])) { best = s; found = 1; }
        norm[best]--; still++;
    }
}

/* inverse of ReadNCount (EntropyCommon.cs:79-188) */
static size_t writeNCount(BYTE *dst, size_t cap, const S16 *norm, U32 maxSymbolValue, U32 tableLog)
{
    BYTE *out = dst;
    BYTE *const oend = dst + cap;
    int const tableSize = 1 << tableLog;
    int remaining = tableSize + 1, threshold = tableSize, nbBits = (int)tableLog + 1;
    U32 bitStream = 0;
    int bitCount = 0;
    U32 charnum = 0;
    int previous0 = 0;
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    while (remaining > 1) {
        if (previous0) {
            U32 start = charnum;
            while (charnum <= maxSymbolValue && !norm[charnum]) charnum++;
            while (charnum >= start + 24) {
                start += 24;
                bitStream += 0xFFFFu << bitCount;
                if (out + 2 > oend) return 0;
                out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (charnum >= start + 3) { start += 3; bitStream += 3u << bitCount; bitCount += 2; }
            bitStream += (charnum - start) << bitCount; bitCount += 2;
            if (bitCount > 16) {
                if (out + 2 > oend) return 0;
                out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
            }
        }
        {
            int count = norm[charnum++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (U32)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previous0 = (count == 1);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) {
            if (out + 2 > oend) return 0;
            out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
        }
    }
    if (out + 2 > oend) return 0;
    out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (size_t)(out - dst);
}

/* ======================================================================= *
 *  Huffman: length-limited code lengths by package-merge
 * ======================================================================= */
/* Leaves are the present symbols sorted by (count, symbol) ascending.  Level lists are
 * merged "leaf first on equal weight".  Only per-level package weights are kept; the number
 * of leaves inside the chosen prefix of each level gives the code lengths. */
static U32 huffLengths(BYTE *nbBits /*256*/, const U32 *count, U32 maxSymbolValue, U32 maxBits)
{
    U32 leafW[256], leafSym[256];
    static __thread U32 pkg[HUF_MAXBITS + 1][256];
    U32 npk[HUF_MAXBITS + 1];
    U32 n = 0, s, level;
    memset(nbBits, 0, 256);
    for (s = 0; s <= maxSymbolValue; s++) if (count[s]) { leafW[n] = count[s]; leafSym[n] = s; n++; }
    if (n == 0) return 0;
    if (n == 1) { nbBits[leafSym[0]] = 1; return 1; }
    /* sort by (count, symbol): insertion sort keeps symbol order for ties */
    { U32 i; for (i = 1; i < n; i++) { U32 w = leafW[i], sy = leafSym[i], j = i; while (j && leafW[j - 1] > w) { leafW[j] = leafW[j - 1]; leafSym[j] = leafSym[j - 1]; j--; } leafW[j] = w; leafSym[j] = sy; } }
    npk[1] = 0;
    for (level = 2; level <= maxBits; level++) {
        /* merged list of level-1 = leaves + pkg[level-1]; pair its items */
        U32 const np = npk[level - 1];
        U32 li = 0, pi = 0, k = 0, total = n + np, have = 0, prev = 0;
        for (; k < total; k++) {
            U32 w;
            if (pi >= np || (li < n && leafW[li] <= pkg[level - 1][pi])) w = leafW[li++]; else w = pkg[level - 1][pi++];
            if (k & 1) pkg[level][have++] = prev + w; else prev = w;
        }
        npk[level] = have;
    }
    {
        U32 m = 2 * n - 2;
        U32 lenOfRank[256];
        memset(lenOfRank, 0, sizeof lenOfRank);
        for (level = maxBits; level >= 1; level--) {
            /* among the first m items of merge(leaves, pkg[level]) count the leaves */
            U32 const np = npk[level];
            U32 li = 0, pi = 0, k;
            if (m > n + np) m = n + np;
            for (k = 0; k < m; k++) { if (pi >= np || (li < n && leafW[li] <= pkg[level][pi])) li++; else pi++; }
            { U32 r; for (r = 0; r < li; r++) lenOfRank[r]++; }
            m = 2 * pi;
            if (!m) break;
        }
        { U32 r, maxLen = 0; for (r = 0; r < n; r++) { nbBits[leafSym[r]] = (BYTE)lenOfRank[r]; if (lenOfRank[r] > maxLen) maxLen = lenOfRank[r]; } return maxLen; }
    }
}

/* code values in the order the decoder lays its table out (HufDecompress.cs:148-176):
 * weight w = tableLog+1-nbBits; lower weights first, symbols ascending inside a weight. */
static void huffCodes(U16 *code, const BYTE *nbBits, U32 maxSymbolValue, U32 tableLog)
{
    U32 rankStart[HUF_MAXBITS + 2];
    U32 rankCount[HUF_MAXBITS + 2];
    U32 s, w, next = 0;
    memset(rankCount, 0, sizeof rankCount);
    for (s = 0; s <= maxSymbolValue; s++) if (nbBits[s]) rankCount[tableLog + 1 - nbBits[s]]++;
    for (w = 1; w <= tableLog; w++) { rankStart[w] = next; next += rankCount[w] << (w - 1); }
    for (s = 0; s <= maxSymbolValue; s++) if (nbBits[s]) {
        w = tableLog + 1 - nbBits[s];
        code[s] = (U16)(rankStart[w] >> (w - 1));
        rankStart[w] += 1u << (w - 1);
    }
}

/* FSE-compress the weights (inverse of FSE_decompress_wksp, FseDecompress.cs:310-332, as used by
 * ReadStats EntropyCommon.cs:226-231).  Two interleaved states, decoder order: state1 first. */
static size_t fseCompressWeights(BYTE *dst, size_t cap, const BYTE *weights, U32 nw)
{
    U32 count[16];
    S16 norm[16];
    U32 maxSym = 0, i, tableLog;
    CTable ct;
    size_t hsize;
    BitW b;
    memset(count, 0, sizeof count);
    if (nw <= 1) return 0;
    for (i = 0; i < nw; i++) { count[weights[i]]++; if (weights[i] > maxSym) maxSym = weights[i]; }
    for (i = 0; i <= maxSym; i++) if (count[i] == nw) return 0;      /* single symbol: not representable */
    tableLog = 6;
    while (tableLog > 5 && (1u << (tableLog - 1)) >= nw) tableLog--;  /* 5 or 6 (FSE_MIN_TABLELOG = 5) */
    { U32 present = 0; for (i = 0; i <= maxSym; i++) present += count[i] != 0; if (present > (1u << tableLog)) return 0; }
    normalizeCounts(norm, tableLog, count, nw, maxSym);
    hsize = writeNCount(dst, cap, norm, maxSym, tableLog);
    if (!hsize) return 0;
    buildCTable(&ct, norm, maxSym, tableLog);
    bw_init(&b, dst + hsize, cap - hsize);
    {
        /* Decoder (FseDecompress.cs:233-295) emits s1,s2,s1,s2,... and ends when the stream
         * overflows after the final symbol.  Encode from the last weight backwards, alternating
         * states so that weight index i is carried by state (i & 1 ? 2 : 1). */
        CState st1, st2;
        int n = (int)nw;
        const BYTE *ip = weights + nw;
        if (n & 1) { cstate_init(&st1, &ct, *--ip, 0); cstate_init(&st2, &ct, *--ip, 0); cstate_encode(&b, &st1, *--ip); n -= 3; }
        else { cstate_init(&st2, &ct, *--ip, 0); cstate_init(&st1, &ct, *--ip, 0); n -= 2; }
        while (n > 0) { cstate_encode(&b, &st2, *--ip); cstate_encode(&b, &st1, *--ip); n -= 2; }
        cstate_flush(&b, &st2);
        cstate_flush(&b, &st1);
    }
    { size_t const s = bw_close(&b); if (!s) return 0; return hsize + s; }
}

/* Huffman table description (inverse of ReadStats, EntropyCommon.cs:198-269) */
static size_t writeHuffHeader(BYTE *dst, size_t cap, const BYTE *nbBits, U32 maxSymbolValue, U32 tableLog)
{
    BYTE weights[256];
    U32 s;
    /* last present symbol's weight is implied */
    for (s = 0; s < maxSymbolValue; s++) weights[s] = nbBits[s] ? (BYTE)(tableLog + 1 - nbBits[s]) : 0;
    if (maxSymbolValue >= 2 && cap > 1) {
        size_t const h = fseCompressWeights(dst + 1, cap - 1 < 127 ? cap - 1 : 127, weights, maxSymbolValue);
        if (h > 1 && h < maxSymbolValue / 2 && h < 128) { dst[0] = (BYTE)h; return h + 1; }
    }
    if (maxSymbolValue > 128) return 0;                       /* direct form holds at most 128 weights */
    if ((maxSymbolValue + 1) / 2 + 1 > cap) return 0;
    dst[0] = (BYTE)(128 + (maxSymbolValue - 1));
    weights[maxSymbolValue] = 0;
    for (s = 0; s < maxSymbolValue; s += 2) dst[s / 2 + 1] = (BYTE)((weights[s] << 4) + weights[s + 1]);
    return (maxSymbolValue + 1) / 2 + 1;
}

/* one Huffman stream: last symbol first, so the backward reader meets symbol 0 first */
static size_t huffEncodeStream(BYTE *dst, size_t cap, const BYTE *src, size_t n, const U16 *code, const BYTE *nbBits)
{
    BitW b;
    size_t i;
    bw_init(&b, dst, cap);
    for (i = n; i > 0; i--) bw_add(&b, code[src[i - 1]], nbBits[src[i - 1]]);
    return bw_close(&b);
}

/* literals section (inverse of DecodeLiteralsBlock, ZStdDecompress.cs:683-821) */
static size_t writeLiterals(BYTE *dst, size_t cap, const BYTE *lit, U32 nlit)
{
    U32 count[256];
    U32 i, maxSym = 0, largest = 0;
    if (cap < 8) return 0;
    memset(count, 0, sizeof count);
    for (i = 0; i < nlit; i++) count[lit[i]]++;
    for (i = 0; i < 256; i++) if (count[i]) { maxSym = i; if (count[i] > largest) largest = count[i]; }
    if (nlit > 0 && largest == nlit && nlit > 4) {
        /* RLE literals */
        if (nlit < 32) { dst[0] = (BYTE)(1 + (nlit << 3)); dst[1] = lit[0]; return 2; }
        if (nlit < 4096) { wr16(dst, 1 + (1 << 2) + (nlit << 4)); dst[2] = lit[0]; return 3; }
        wr24(dst, 1 + (3 << 2) + (nlit << 4)); dst[3] = lit[0]; return 4;
    }
    if (nlit >= 64) {
        BYTE nbBits[256];
        U16 code[256];
        U32 const tableLog = huffLengths(nbBits, count, maxSym, HUF_MAXBITS);
        U32 const lhSize = 3 + (nlit >= 1024) + (nlit >= 16384);
        int const single = nlit < 256;
        BYTE *op = dst + lhSize;
        BYTE *const oend = dst + cap;
        size_t hsz, csz;
        huffCodes(code, nbBits, maxSym, tableLog);
        hsz = writeHuffHeader(op, (size_t)(oend - op), nbBits, maxSym, tableLog);
        if (hsz) {
            int ok = 1;
            op += hsz;
            if (single) {
                size_t const s = huffEncodeStream(op, (size_t)(oend - op), lit, nlit, code, nbBits);
                if (!s) ok = 0; else op += s;
            } else {
                U32 const seg = (nlit + 3) / 4;
                BYTE *const jump = op;
                U32 k;
                if ((size_t)(oend - op) < 6) ok = 0; else op += 6;
                for (k = 0; ok && k < 4; k++) {
                    U32 const from = k * seg;
                    U32 const len = (k < 3) ? seg : nlit - 3 * seg;
                    size_t const s = huffEncodeStream(op, (size_t)(oend - op), lit + from, len, code, nbBits);
                    if (!s || s > 65535) { ok = 0; break; }
                    if (k < 3) wr16(jump + 2 * k, (U32)s);
                    op += s;
                }
            }
            csz = (size_t)(op - (dst + lhSize));
            if (ok && csz + lhSize < nlit + (3 - (nlit < 32) - (nlit < 4096)) && (single || csz >= 10)) {
                U32 const hType = 2;   /* set_compressed */
                switch (lhSize) {
                case 3: wr24(dst, hType + ((U32)(!single) << 2) + (nlit << 4) + ((U32)csz << 14)); break;
                case 4: wr32(dst, hType + (2 << 2) + (nlit << 4) + ((U32)csz << 18)); break;
                default: wr32(dst, hType + (3 << 2) + (nlit << 4) + ((U32)csz << 22)); dst[4] = (BYTE)(csz >> 10); break;
                }
                return lhSize + csz;
            }
        }
    }
    /* raw literals */
    {
        U32 const lh = 1 + (nlit > 31) + (nlit > 4095);
        if (lh + nlit > cap) return 0;
        switch (lh) {
        case 1: dst[0] = (BYTE)(nlit << 3); break;
        case 2: wr16(dst, (1 << 2) + (nlit << 4)); break;
        default: wr24(dst, (3 << 2) + (nlit << 4)); break;
        }
        memcpy(dst + lh, lit, nlit);
        return lh + nlit;
    }
}

/* ======================================================================= *
 *  LZ stage
 * ======================================================================= */
typedef struct { U32 litLength, matchLength, offset; } Seq;
typedef struct { U32 start, ml, off; } ASeq;   /* match start (unit position, after backward extension), length, distance */

typedef struct {
    U32 dist[UNIT_MAX];                       /* candidate distance per unit position, 0 = none */
    U32 tabS[1u << MAX_TABLE_LOG], tabL[1u << MAX_TABLE_LOG];
    ASeq rangeSeq[BLOCK_MAX / 4];             /* a range's records start at (range start - block start) / 4: its matches start inside it and are >= 4 bytes long */
    U32 rangeN[BLOCK_MAX >> WALK_LOG_MIN];
    Seq seqs[BLOCK_MAX / 3 + 8];
    BYTE lits[BLOCK_MAX + 8];
    BYTE llCode[BLOCK_MAX / 3 + 8], mlCode[BLOCK_MAX / 3 + 8], ofCode[BLOCK_MAX / 3 + 8];
    U32 ofValue[BLOCK_MAX / 3 + 8];           /* offset field value: 1..3 repcode, else offset+3 */
    BYTE tmp[BLOCK_MAX + 1024];
    int matchless;                            /* the unit's candidates are too few to be worth a parse: its blocks get no sequences */
} Work;

/* stage 1, once per unit: two unit-wide hash tables, every position inserted in position order, the last writer
 * keeps a slot.
 *   short: hash of 5 bytes, long: hash of 8 bytes (level >= 3 only); 2^13 slots for units <= 64 KiB, 2^14 above.
 *   slot = tag (15 hash bits below the index bits) << 17 | position; a lookup whose tag differs is a miss.
 * A position's candidate is the previous owner of its long slot if the tag agrees, else that of its short slot.
 * Candidates are NOT compared with the bytes here (28 hash bits agree; the walk measures every match it uses).
 * On the GPU one wavefront per table takes 64 positions per LDS exchange instruction (ds_wrxchg_rtn_b32);
 * the LDS resolves lanes that hit the same slot in ascending lane order (probed: tools/probe/lds_xchg.hip), which
 * is exactly this loop.
 * MATCHLESS units (round 4): a unit with fewer than n / 2048 candidate positions (32 per 64 KiB; units below 2 KiB never) is not parsed at
 * all: its blocks get no sequences and go straight to the literal / raw decision.  Incompressible input leaves a handful of chance
 * candidates (28 hash bits agree by accident: ~4 per 64 KiB of noise) and used to pay the whole walk for them; what the rule can
 * cost is those < 32 matches of an otherwise matchless unit. */
#define MATCHLESS_SHIFT 11
#define SLOT_EMPTY 0xFFFFFFFFu
/* hashes made of 24 x 24 -> 32 bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 run at full rate on CDNA, v_mul_lo_u32 at a quarter):
 * short: bytes 0-2 and 2-4; long: bytes 0-2, 3-5, 6-7 */
static U32 mul24(U32 a, U32 b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static U32 hashShort(const BYTE *p) { U32 const lo = rd32(p), hi = rd32(p + 4); return mul24(lo, 0x9E3779u) + mul24((lo >> 16) | (hi << 16), 0x85EBCBu); }
static U32 hashLong(const BYTE *p) { U32 const lo = rd32(p), hi = rd32(p + 4); return mul24(lo, 0x9E3779u) + mul24((lo >> 24) | (hi << 8), 0x85EBCBu) + mul24(hi >> 16, 0xC2B2AFu); }
static U32 tableLogFor(U32 unitN) { return unitN > BLOCK_MAX ? MAX_TABLE_LOG : MAX_TABLE_LOG - 1; }
static void findCandidates(Work *w, const BYTE *src, U32 n, const EParams *prm)
{
    U32 const tlog = tableLogFor(n);
    U32 p, found = 0;
    memset(w->dist, 0, n * sizeof(U32));
    w->matchless = 0;
    if (n < 8) return;
    memset(w->tabS, 0xFF, sizeof(U32) << tlog);
    if (prm->useLong) memset(w->tabL, 0xFF, sizeof(U32) << tlog);
    for (p = 0; p + 8 <= n; p++) {
        U32 const hs = hashShort(src + p);
        U32 const es = (((hs >> (32 - tlog - 15)) & 0x7FFFu) << 17) | p;
        U32 const os = w->tabS[hs >> (32 - tlog)];
        U32 d = 0;
        w->tabS[hs >> (32 - tlog)] = es;
        if (os != SLOT_EMPTY && ((os ^ es) >> 17) == 0) d = p - (os & 0x1FFFFu);
        if (prm->useLong) {
            U32 const hl = hashLong(src + p);
            U32 const el = (((hl >> (32 - tlog - 15)) & 0x7FFFu) << 17) | p;
            U32 const ol = w->tabL[hl >> (32 - tlog)];
            w->tabL[hl >> (32 - tlog)] = el;
            if (ol != SLOT_EMPTY && ((ol ^ el) >> 17) == 0) d = p - (ol & 0x1FFFFu);
        }
        w->dist[p] = d;
        found += d != 0;
    }
    w->matchless = found < (n >> MATCHLESS_SHIFT);
}

static U32 matchLen(const BYTE *src, U32 a, U32 b, U32 limit)   /* common prefix of src[a..] and src[b..], a > b, up to limit */
{
    U32 l = 0;
    while (a + l < limit && src[a + l] == src[b + l]) l++;
    return l;
}

/* stage 2 : one walk range [start, end), walked by one walker on the GPU; matches may run on to `limit` (> end: the
 * next ranges' territory, given back by the stitch below).
 * Each step looks at the positions from ip to the end of the windowGroups-th aligned group of 8 (2 groups: 9 .. 16 positions; 4 groups:
 * 25 .. 32: what the GPU walker's lanes load of the candidate distances, 16 bytes each).  A position holds a candidate if (in this order of preference)
 * one of the walker's two recent offsets repeats 4 bytes there (only the first repWin positions of the window are
 * tried, and only while ip lies at least that offset inside the unit) or stage 1 left a distance.  The first LOOK such
 * positions are scored: forward match length (the score counts at most FCAP bytes), backward extension into the
 * pending literals (at most BCAP bytes), offset cost (none for a recent offset), literals skipped.  The best one (the
 * earliest among equals) becomes a sequence with its full forward length. */
#define FCAP 8u
#define BCAP 4u
#define REPMIN 4u
static U32 walkRange(Work *w, const BYTE *src, U32 n, U32 start, U32 end, U32 limit, const EParams *prm, ASeq *out)
{
    U32 ip = start, anchor = start, nseq = 0, rep0 = 0, rep1 = 0;
    U32 const hashable = (n >= 8) ? n - 7 : 0;
    U32 const look = (U32)prm->look, repWin = (U32)prm->repWin, window = 8u * (U32)prm->windowGroups;
    U32 const scanEnd = (end < hashable) ? end : hashable;     /* candidates start below this */
    while (ip < scanEnd) {
        int bestGain = 0, have = 0; U32 bestQ = 0, bestFwd = 0, bestBack = 0, bestOff = 0, q, seen = 0;
        U32 const wend = ((ip & ~7u) + window < scanEnd) ? (ip & ~7u) + window : scanEnd;   /* windowGroups aligned groups of 8 positions */
        int const try0 = rep0 && ip >= rep0, try1 = rep1 && ip >= rep1;
        for (q = ip; q < wend && seen < look; q++) {
            U32 off = 0, fwd, back = 0; int isRep = 0, gain;
            if (q < ip + repWin && q + 4 <= limit) {
                if (try0 && rd32(src + q) == rd32(src + q - rep0)) { off = rep0; isRep = 1; }
                else if (try1 && rd32(src + q) == rd32(src + q - rep1)) { off = rep1; isRep = 1; }
            }
            if (!off) off = w->dist[q];
            if (!off) continue;
            seen++;
            fwd = matchLen(src, q, q - off, limit);
            if (fwd < (isRep ? REPMIN : MINMATCH)) continue;
            while (back < BCAP && q - back > anchor && q - off - back > 0 && src[q - back - 1] == src[q - off - back - 1]) back++;
            gain = (int)((fwd > FCAP ? FCAP : fwd) + back) * 4 - (isRep ? 0 : (int)highbit32(off + 1)) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
            if (!have || gain > bestGain) { have = 1; bestGain = gain; bestQ = q; bestFwd = fwd; bestBack = back; bestOff = off; }
        }
        if (!have) { ip = wend; continue; }
        out[nseq].start = bestQ - bestBack; out[nseq].ml = bestBack + bestFwd; out[nseq].off = bestOff; nseq++;
        ip = bestQ + bestFwd; anchor = ip;
        if (bestOff == rep1) { rep1 = rep0; rep0 = bestOff; }
        else if (bestOff != rep0) { rep1 = rep0; rep0 = bestOff; }
    }
    return nseq;
}

/* stages 2 + 3a for one block: every range walked on its own, then the stitch.  reach = farthest match end of the ranges
 * so far (as walked).  A range's records that end at or before the reach are dropped; one that straddles it loses its front
 * (and goes if fewer than MINMATCH bytes are left).  The first record a range keeps is JOINED to the match that defines the
 * reach -- instead of becoming a sequence -- when it starts exactly at the reach, has that match's offset, and that match
 * (the last record of its range) was itself kept: a long match found piecewise by consecutive ranges is one sequence.
 * Every decision follows from the ranges' last match ends and last offsets alone, so the GPU decides all ranges at once
 * (k_lz_walk: a scan of the ends, one lane per range, a second scan for the joined lengths).
 * Leaves w->seqs / w->lits; returns the number of sequences. */
static U32 parseBlock(Work *w, const BYTE *src, U32 unitN, U32 blockOff, U32 n, const EParams *prm, U32 *nlitOut)
{
    U32 nseq = 0, nlit = 0;
    U32 const WS = 1u << prm->walkLog;
    U32 const nRanges = (n + WS - 1) >> prm->walkLog;
    U32 const blockEnd = blockOff + n;
    U32 r, reach = blockOff, pos = blockOff;       /* pos: end of the last sequence's match (<= reach) */
    U32 reachOff = 0; int reachKept = 0;            /* the match that defines the reach: its offset; whether it is the last sequence */
    for (r = 0; r < nRanges; r++) {
        U32 const start = blockOff + (r << prm->walkLog);
        U32 const end = (start + WS < blockEnd) ? start + WS : blockEnd;
        U32 const limit = (end + CROSS_MAX < blockEnd) ? end + CROSS_MAX : blockEnd;
        w->rangeN[r] = w->matchless ? 0 : walkRange(w, src, unitN, start, end, limit, prm, w->rangeSeq + ((start - blockOff) >> 2));
    }
    for (r = 0; r < nRanges; r++) {
        ASeq *const rs = w->rangeSeq + (r << (prm->walkLog - 2));
        U32 const own = reach;                         /* reach of the ranges before r */
        U32 const ns = w->rangeN[r];
        U32 const le = ns ? rs[ns - 1].start + rs[ns - 1].ml : 0;
        U32 f = 0, k;
        while (f < ns) {
            ASeq *s = &rs[f];
            if (s->start + s->ml <= own) { f++; continue; }
            if (s->start < own) { U32 const cut = own - s->start; if (s->ml - cut < MINMATCH) { f++; continue; } s->start += cut; s->ml -= cut; }
            break;
        }
        for (k = f; k < ns; k++) {
            ASeq const s = rs[k];
            if (k == f && reachKept && s.start == own && s.off == reachOff) { w->seqs[nseq - 1].matchLength += s.ml; pos = s.start + s.ml; continue; }
            w->seqs[nseq].litLength = s.start - pos; w->seqs[nseq].matchLength = s.ml; w->seqs[nseq].offset = s.off; nseq++;
            memcpy(w->lits + nlit, src + pos, s.start - pos); nlit += s.start - pos;
            pos = s.start + s.ml;
        }
        if (le > reach) { reach = le; reachOff = rs[ns - 1].off; reachKept = (f < ns); }
    }
    memcpy(w->lits + nlit, src + pos, blockEnd - pos); nlit += blockEnd - pos;    /* last literals */
    *nlitOut = nlit;
    return nseq;
}

/* ======================================================================= *
 *  one block -> compressed block payload (without the 3-byte block header)
 *  returns payload size, or 0 if the block should be stored raw
 * ======================================================================= */
/* src / unitN: the LZ unit (findCandidates has run on it); the block is src[blockOff .. blockOff + n) */
static size_t encodeParsed(Work *w, BYTE *dst, size_t cap, U32 nseq, U32 nlit, int firstBlock);
static size_t compressBlock(Work *w, BYTE *dst, size_t cap, const BYTE *src, U32 unitN, U32 blockOff, U32 n, const EParams *prm, int firstBlock)
{
    U32 nseq, nlit = 0;
    if (n < 16) return 0;
    nseq = parseBlock(w, src, unitN, blockOff, n, prm, &nlit);
    return encodeParsed(w, dst, cap, nseq, nlit, firstBlock);
}

/* stages 3b-6 for one block whose sequences (w->seqs) and literals (w->lits) are in place */
U32 g_lastNseq;   /* development aid (tools/lab): sequences of the last block encoded */
static size_t encodeParsed(Work *w, BYTE *dst, size_t cap, U32 nseq, U32 nlit, int firstBlock)
{
    g_lastNseq = nseq;
    {
        /* stage 3b: offsets -> offset field values through the 3-entry recent-offset list
         * (inverse of ZStdDecompress.cs:1509-1530).  Blocks after the first start from an unknown
         * history: sentinels that never equal a real offset (< 131072). */
        U32 rep[3];
        U32 i;
        if (firstBlock) { rep[0] = 1; rep[1] = 4; rep[2] = 8; } else { rep[0] = 0xFFFFFFF1u; rep[1] = 0xFFFFFFF2u; rep[2] = 0xFFFFFFF3u; }
        for (i = 0; i < nseq; i++) {
            U32 const off = w->seqs[i].offset, ll = w->seqs[i].litLength;
            U32 val;
            if (ll) {
                if (off == rep[0]) val = 1;
                else if (off == rep[1]) { val = 2; rep[1] = rep[0]; rep[0] = off; }
                else if (off == rep[2]) { val = 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
                else { val = off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
            } else {
                if (off == rep[1]) { val = 1; rep[1] = rep[0]; rep[0] = off; }
                else if (off == rep[2]) { val = 2; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
                else { val = off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }   /* includes off == rep[0] */
            }
            w->ofValue[i] = val;
            w->ofCode[i] = (BYTE)highbit32(val);
            w->llCode[i] = (BYTE)llCodeOf(ll);
            w->mlCode[i] = (BYTE)mlCodeOf(w->seqs[i].matchLength - 3);
        }
    }
    {
        BYTE *op = dst;
        BYTE *const oend = dst + cap;
        size_t const litSize = writeLiterals(op, cap, w->lits, nlit);
        if (!litSize) return 0;
        op += litSize;
        /* sequences section header (inverse of DecodeSeqHeaders, ZStdDecompress.cs:1110-1180) */
        if ((size_t)(oend - op) < 4) return 0;
        if (nseq < 128) *op++ = (BYTE)nseq;
        else if (nseq < 0x7F00) { op[0] = (BYTE)((nseq >> 8) + 0x80); op[1] = (BYTE)nseq; op += 2; }
        else { op[0] = 0xFF; wr16(op + 1, nseq - 0x7F00); op += 3; }
        if (nseq == 0) return (size_t)(op - dst);
        {
            BYTE *const modes = op++;
            CTable ctLL, ctOF, ctML;
            int rleLL = 0, rleOF = 0, rleML = 0;
            U32 t;
            for (t = 0; t < 3; t++) {
                const BYTE *codes = t == 0 ? w->llCode : (t == 1 ? w->ofCode : w->mlCode);
                U32 const maxCode = t == 0 ? MaxLL : (t == 1 ? MaxOff : MaxML);
                U32 const maxLog = t == 1 ? OffFSELog : LLFSELog;
                const S16 *defNorm = t == 0 ? LL_defaultNorm : (t == 1 ? OF_defaultNorm : ML_defaultNorm);
                U32 const defLog = t == 1 ? 5 : 6, defMax = t == 0 ? MaxLL : (t == 1 ? 28 : MaxML);
                CTable *ct = t == 0 ? &ctLL : (t == 1 ? &ctOF : &ctML);
                int *rle = t == 0 ? &rleLL : (t == 1 ? &rleOF : &rleML);
                U32 count[64];
                U32 i, maxSym = 0, largest = 0, mode;
                memset(count, 0, sizeof count);
                for (i = 0; i < nseq; i++) count[codes[i]]++;
                for (i = 0; i <= maxCode; i++) if (count[i]) { maxSym = i; if (count[i] > largest) largest = count[i]; }
                if (largest == nseq) {
                    mode = 1;                                       /* set_rle (ZStdDecompress.cs:937-955) */
                    if (op >= oend) return 0;
                    *op++ = (BYTE)maxSym; *rle = 1; ct->tableLog = 0;
                } else if (nseq < 64 && maxSym <= defMax) {
                    mode = 0;                                       /* set_basic: predefined distribution */
                    buildCTable(ct, defNorm, defMax, defLog);
                } else {
                    S16 norm[64];
                    U32 tableLog = maxLog;
                    size_t h;
                    { U32 const hb = highbit32(nseq - 1); U32 const want = hb > 2 ? hb - 2 : 5; if (want < tableLog) tableLog = want; }
                    { U32 const minBits = highbit32(maxSym) + 2; U32 present = 0; for (i = 0; i <= maxSym; i++) present += count[i] != 0;
                      if (tableLog < minBits) tableLog = minBits; while ((1u << tableLog) < present) tableLog++; }
                    if (tableLog < 5) tableLog = 5;
                    if (tableLog > maxLog) tableLog = maxLog;
                    normalizeCounts(norm, tableLog, count, nseq, maxSym);
                    h = writeNCount(op, (size_t)(oend - op), norm, maxSym, tableLog);
                    if (!h) return 0;
                    op += h;
                    mode = 2;                                       /* set_compressed */
                    buildCTable(ct, norm, maxSym, tableLog);
                }
                if (t == 0) *modes = (BYTE)(mode << 6); else if (t == 1) *modes |= (BYTE)(mode << 4); else *modes |= (BYTE)(mode << 2);
            }
            {
                /* bitstream (inverse of ZStdDecompress.cs:1473-1608): last sequence first */
                BitW b;
                CState sLL, sOF, sML;
                U32 i = nseq - 1;
                size_t s;
                bw_init(&b, op, (size_t)(oend - op));
                cstate_init(&sML, &ctML, w->mlCode[i], rleML);
                cstate_init(&sOF, &ctOF, w->ofCode[i], rleOF);
                cstate_init(&sLL, &ctLL, w->llCode[i], rleLL);
                bw_add(&b, w->seqs[i].litLength - LL_base[w->llCode[i]], LL_bits[w->llCode[i]]);
                bw_add(&b, w->seqs[i].matchLength - ML_base[w->mlCode[i]], ML_bits[w->mlCode[i]]);
                bw_add(&b, w->ofValue[i] - (1u << w->ofCode[i]), w->ofCode[i]);
                while (i-- > 0) {
                    cstate_encode(&b, &sOF, w->ofCode[i]);
                    cstate_encode(&b, &sML, w->mlCode[i]);
                    cstate_encode(&b, &sLL, w->llCode[i]);
                    bw_add(&b, w->seqs[i].litLength - LL_base[w->llCode[i]], LL_bits[w->llCode[i]]);
                    bw_add(&b, w->seqs[i].matchLength - ML_base[w->mlCode[i]], ML_bits[w->mlCode[i]]);
                    bw_add(&b, w->ofValue[i] - (1u << w->ofCode[i]), w->ofCode[i]);
                }
                cstate_flush(&b, &sML);
                cstate_flush(&b, &sOF);
                cstate_flush(&b, &sLL);
                s = bw_close(&b);
                if (!s) return 0;
                op += s;
            }
        }
        return (size_t)(op - dst);
    }
}

/* upstream bound formula (commented macro at ZStd.cs:144-145), plus room for the extra block headers
 * this encoder's 64 KiB blocks add */
size_t zso_compressBound(size_t srcSize)
{
    return srcSize + (srcSize >> 8) + ((srcSize < (128u << 10)) ? (((128u << 10) - srcSize) >> 11) : 0) + 3 * (srcSize / BLOCK_MAX + 1) + 18;
}

/* frame = magic + FHD + FCS (single segment) + blocks   (inverse of ZStdDecompress.cs:421-499, 2008-2091) */
size_t zso_compress(void *dstv, size_t dstCapacity, const void *srcv, size_t srcSize, int level)
{
    BYTE *const dst = (BYTE *)dstv;
    const BYTE *const src = (const BYTE *)srcv;
    BYTE *op = dst;
    BYTE *const oend = dst + dstCapacity;
    EParams const prm = paramsForLevel(level);
    Work *w;
    size_t pos = 0;
    if (srcSize > 0xFFFFFFFFu) return ERR(ZSO_srcSize_wrong);
    if (dstCapacity < 4 + 1 + 4 + 3) return ERR(ZSO_dstSize_tooSmall);
    wr32(op, 0xFD2FB528u); op += 4;
    if (srcSize < 256) { *op++ = 0x20; *op++ = (BYTE)srcSize; }
    else if (srcSize < 65536 + 256) { *op++ = 0x60; wr16(op, (U32)srcSize - 256); op += 2; }
    else { *op++ = 0xA0; wr32(op, (U32)srcSize); op += 4; }
    {
        static __thread Work *tls_work;          /* one workspace per thread, kept for the thread's life */
        if (!tls_work) tls_work = (Work *)malloc(sizeof(Work));
        w = tls_work;
    }
    if (!w) return ERR(ZSO_memory_allocation);
    do {
        U32 const n = (U32)((srcSize - pos < BLOCK_MAX) ? srcSize - pos : BLOCK_MAX);
        int const last = (pos + n == srcSize);
        size_t const unitPos = pos & ~(size_t)(UNIT_MAX - 1);             /* units are cut every 128 KiB of the chunk */
        U32 const unitN = (U32)((srcSize - unitPos < UNIT_MAX) ? srcSize - unitPos : UNIT_MAX);
        if (pos == unitPos && n) findCandidates(w, src + unitPos, unitN, &prm);
        size_t csize = 0;
        U32 i, same = n > 0;
        if ((size_t)(oend - op) < 3 + 1) return ERR(ZSO_dstSize_tooSmall);
        for (i = 1; i < n && same; i++) same = src[pos + i] == src[pos];
        if (same && n > 0) {                                        /* RLE block (ZStdDecompress.cs:1945-1950) */
            wr24(op, (U32)last + (1u << 1) + (n << 3)); op[3] = src[pos]; op += 4;
        } else {
            /* the payload is built in scratch of n + 512 bytes; it is used iff it is smaller than n
             * (anything that would not fit the scratch is larger than n anyway) */
            if (n) csize = compressBlock(w, w->tmp, n + 512, src + unitPos, unitN, (U32)(pos - unitPos), n, &prm, pos == 0);
            if (csize && csize < n && (size_t)(oend - op) >= 3 + csize) { wr24(op, (U32)last + (2u << 1) + ((U32)csize << 3)); memcpy(op + 3, w->tmp, csize); op += 3 + csize; }
            else {                                                   /* raw block (ZStdDecompress.cs:662-667) */
                if ((size_t)(oend - op) < 3 + (size_t)n) return ERR(ZSO_dstSize_tooSmall);
                wr24(op, (U32)last + (0u << 1) + (n << 3)); memcpy(op + 3, src + pos, n); op += 3 + n;
            }
        }
        pos += n;
    } while (pos < srcSize);
    return (size_t)(op - dst);
}

/* ---- test hooks: intermediate results of stages 1 and 2 for one LZ unit (n <= 131072), so the HIP
 *      kernels can be checked stage by stage ---- */
int zso_debugCandidates(uint32_t *distOut, const void *src, uint32_t n, int level)
{
    EParams const prm = paramsForLevel(level);
    Work *w = (Work *)malloc(sizeof(Work));
    if (!w || n > UNIT_MAX) { free(w); return -1; }
    findCandidates(w, (const BYTE *)src, n, &prm);
    memcpy(distOut, w->dist, n * sizeof(U32));
    free(w);
    return 0;
}
/* after the stitch: the sequences of the unit's blocks in turn.  seqOut[3 k ..] = (match start in its block, matchLength, offset) of the
 * block's k-th sequence, nseqOut[b] = sequences of block b (what the GPU walk kernel leaves, its output ranges read in order) */
int zso_debugWalk(uint32_t *seqOut, uint32_t *nseqOut, const void *src, uint32_t n, int level)
{
    EParams const prm = paramsForLevel(level);
    Work *w = (Work *)malloc(sizeof(Work));
    U32 blockOff, b = 0, o = 0;
    if (!w || n > UNIT_MAX) { free(w); return -1; }
    findCandidates(w, (const BYTE *)src, n, &prm);
    for (blockOff = 0; blockOff < n; blockOff += BLOCK_MAX, b++) {
        U32 const bn = (n - blockOff < BLOCK_MAX) ? n - blockOff : BLOCK_MAX;
        U32 nlit = 0, k, pos = 0;
        U32 const ns = (bn < 16) ? 0 : parseBlock(w, (const BYTE *)src, n, blockOff, bn, &prm, &nlit);
        for (k = 0; k < ns; k++) {
            pos += w->seqs[k].litLength;
            seqOut[o++] = pos; seqOut[o++] = w->seqs[k].matchLength; seqOut[o++] = w->seqs[k].offset;
            pos += w->seqs[k].matchLength;
        }
        nseqOut[b] = ns;
    }
    free(w);
    return 0;
}
