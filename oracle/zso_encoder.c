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
 *  1. candidates: the unit is cut in RANGES of 8 KiB; each range owns a hash table of 2^HASH_LOG
 *     16-bit slots.  A slot holds the position within the range (13 bits) and a 3-bit TAG (the
 *     hash bits below the slot index).  Position p hashes its 4 bytes and takes the slot's
 *     previous owner in its own range if the tag agrees, else the entry of the nearest earlier
 *     range whose slot is filled with the same tag.  A candidate is kept when its 4 bytes equal
 *     those at p.  -> dist[p]
 *  2. parse: the unit is cut again, in walk ranges of 1 KiB; each is walked greedily and
 *     independently: the first LOOK candidates of a 64-position window are scored, the best one
 *     becomes a sequence, extended forward to the range end at most.
 *  3. ranges are concatenated (a range's trailing literals go to the next range's first
 *     sequence), offsets become repcodes through the decoder's 3-entry recent-offset list
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
#define RANGE_LOG   13
#define RANGE_SIZE  (1u << RANGE_LOG)
#define MAX_RANGES  (UNIT_MAX / RANGE_SIZE)
#define WALK_LOG    10              /* the walk cuts the unit in ranges of 1 KiB (the hash tables keep their 8 KiB ranges) */
#define WALK_SIZE   (1u << WALK_LOG)
#define WALK_RANGES (UNIT_MAX / WALK_SIZE)
#define MINMATCH    5               /* shortest match kept (candidates are still found by their first 4 bytes) */
#define MAX_HASH_LOG 13
#define TAG_BITS    3               /* RANGE_LOG + TAG_BITS = 16: one slot is a uint16 */
#define SLOT_EMPTY  0xFFFFu
#define HUF_MAXBITS 11
#define MaxLL 35
#define MaxML 52
#define MaxOff 31
#define LLFSELog 9
#define MLFSELog 9
#define OffFSELog 8

/* level <= 2 : LOOK 4 ("fast") ; level >= 3 : LOOK 8.  Both: 2^12 slots per 8 KiB range. */
typedef struct { int hashLog; int look; } EParams;
static EParams g_override = { 0, 0 };
/* test hook: lets the ratio-tuning script try parameters without recompiling (0 = keep level default) */
void zso_encoderOverride(int hashLog, int look)
{ g_override.hashLog = hashLog > MAX_HASH_LOG ? MAX_HASH_LOG : hashLog; g_override.look = look; }

static EParams paramsForLevel(int level)
{
    EParams p;
    p.hashLog = 12; p.look = (level <= 2) ? 4 : 8;
    if (g_override.hashLog) p.hashLog = g_override.hashLog;
    if (g_override.look) p.look = g_override.look;
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

typedef struct {
    U32 dist[UNIT_MAX];                       /* verified candidate distance per unit position, 0 = none */
    U16 tables[MAX_RANGES][1 << MAX_HASH_LOG];          /* per-range hash tables (slot = tag << 13 | position in range) */
    Seq seqs[BLOCK_MAX / 3 + 8];
    BYTE lits[BLOCK_MAX + 8];
    BYTE llCode[BLOCK_MAX / 3 + 8], mlCode[BLOCK_MAX / 3 + 8], ofCode[BLOCK_MAX / 3 + 8];
    U32 ofValue[BLOCK_MAX / 3 + 8];           /* offset field value: 1..3 repcode, else offset+3 */
    BYTE tmp[BLOCK_MAX + 1024];
} Work;

/* stage 1, once per unit.  Positions are taken in STEPS of 64 (one wavefront): all 64 read the tables
 * first, then all 64 write their own range's table, the highest position winning a shared slot; so a
 * position never sees a candidate from its own step.  A distance of exactly 65536 is not kept (the GPU
 * keeps the low 16 bits of the distance in one array and bit 16 in another; low bits 0 = no candidate). */
#define STEP 64u
static U32 slotOf(U32 v, int hashLog) { return (v * 2654435761u) >> (32 - hashLog); }
static U32 tagOf(U32 v, int hashLog) { return ((v * 2654435761u) >> (32 - hashLog - TAG_BITS)) & ((1u << TAG_BITS) - 1); }
static void findCandidates(Work *w, const BYTE *src, U32 n, const EParams *prm)
{
    U32 const nRanges = (n + RANGE_SIZE - 1) >> RANGE_LOG;
    U32 const last = (n >= 4) ? n - 4 : 0;          /* last position whose 4 bytes exist */
    U32 r, p, base;
    memset(w->dist, 0, n * sizeof(U32));
    if (n < 4) return;
    for (r = 0; r < nRanges; r++) memset(w->tables[r], 0xFF, sizeof(U16) << prm->hashLog);
    for (r = 0; r < nRanges; r++) {
        U32 const start = r << RANGE_LOG;
        U32 end = start + RANGE_SIZE; if (end > last + 1) end = last + 1;
        for (base = start; base < end; base += STEP) {
            U32 const stop = base + STEP < end ? base + STEP : end;
            for (p = base; p < stop; p++) {
                U32 const v = rd32(src + p);
                U32 const h = slotOf(v, prm->hashLog), tag = tagOf(v, prm->hashLog);
                int q;
                for (q = (int)r; q >= 0; q--) {
                    U32 const e = w->tables[q][h];
                    if (e != SLOT_EMPTY && (e >> RANGE_LOG) == tag) {
                        U32 const cand = ((U32)q << RANGE_LOG) + (e & (RANGE_SIZE - 1));
                        if (rd32(src + cand) == v && p - cand != 65536u) w->dist[p] = p - cand;
                        break;
                    }
                }
            }
            for (p = base; p < stop; p++) {
                U32 const v = rd32(src + p);
                w->tables[r][slotOf(v, prm->hashLog)] = (U16)((tagOf(v, prm->hashLog) << RANGE_LOG) | (p & (RANGE_SIZE - 1)));
            }
        }
    }
}

static U32 matchLen(const BYTE *src, U32 a, U32 b, U32 limit)   /* common prefix of src[a..] and src[b..], a > b, up to limit */
{
    U32 l = 0;
    while (a + l < limit && src[a + l] == src[b + l]) l++;
    return l;
}

/* stage 2 : one range, walked by one wavefront on the GPU.
 * Each step looks at the WINDOW = 64 positions from ip, takes the first LOOK (<= 8) of them that hold a
 * candidate, and scores each: forward match length (compared over at most FCAP bytes for the score),
 * backward extension into the pending literals (at most BCAP bytes), offset cost, literals skipped.
 * The best one becomes a sequence; if its forward compare hit FCAP it is then extended in full.
 * Matches stop at the range end. */
#define WINDOW 64u
#define FCAP 8u
#define BCAP 8u
static U32 walkRange(Work *w, const BYTE *src, U32 n, U32 start, U32 end, const EParams *prm, Seq *out, U32 *trailingLits)
{
    U32 ip = start, anchor = start, nseq = 0;
    U32 const lastStart = (n >= 4) ? n - 4 : 0;
    U32 const look = (U32)prm->look;
    U32 const scanEnd = (end < lastStart + 1) ? end : lastStart + 1;     /* candidates start below this */
    while (ip < scanEnd) {
        int bestGain = 0, have = 0; U32 bestQ = 0, bestFwd = 0, bestBack = 0, bestOff = 0, q, seen = 0;
        U32 const wend = (ip + WINDOW < scanEnd) ? ip + WINDOW : scanEnd;
        for (q = ip; q < wend && seen < look; q++) {
            U32 const off = w->dist[q];
            U32 fwd, back = 0, cap;
            int gain;
            if (!off) continue;
            seen++;
            cap = end - q; if (cap > FCAP) cap = FCAP;
            fwd = matchLen(src, q, q - off, q + cap);
            if (fwd < MINMATCH) continue;
            while (back < BCAP && q - back > anchor && q - off - back > 0 && src[q - back - 1] == src[q - off - back - 1]) back++;
            gain = (int)(fwd + back) * 4 - (int)highbit32(off + 1) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
            if (!have || gain > bestGain) { have = 1; bestGain = gain; bestQ = q; bestFwd = fwd; bestBack = back; bestOff = off; }
        }
        if (!have) { ip = wend; continue; }
        if (bestFwd == FCAP) bestFwd = matchLen(src, bestQ, bestQ - bestOff, end);
        out[nseq].litLength = bestQ - bestBack - anchor; out[nseq].matchLength = bestBack + bestFwd; out[nseq].offset = bestOff; nseq++;
        ip = bestQ + bestFwd; anchor = ip;
    }
    *trailingLits = end - anchor;
    return nseq;
}

/* ======================================================================= *
 *  one block -> compressed block payload (without the 3-byte block header)
 *  returns payload size, or 0 if the block should be stored raw
 * ======================================================================= */
/* src / unitN: the LZ unit (findCandidates has run on it); the block is src[blockOff .. blockOff + n) */
static size_t compressBlock(Work *w, BYTE *dst, size_t cap, const BYTE *src, U32 unitN, U32 blockOff, U32 n, const EParams *prm, int firstBlock)
{
    U32 nseq = 0, nlit = 0;
    U32 const nRanges = (n + WALK_SIZE - 1) >> WALK_LOG;
    U32 const blockEnd = blockOff + n;
    U32 r;
    if (n < 16) return 0;
    {
        /* stage 2 + 3a: walk ranges, concatenate */
        U32 carry = 0, pos = blockOff;
        static __thread Seq rangeSeq[WALK_SIZE / 3 + 8];
        for (r = 0; r < nRanges; r++) {
            U32 const start = blockOff + (r << WALK_LOG);
            U32 const end = (start + WALK_SIZE < blockEnd) ? start + WALK_SIZE : blockEnd;
            U32 trailing, k;
            U32 const ns = walkRange(w, src, unitN, start, end, prm, rangeSeq, &trailing);
            for (k = 0; k < ns; k++) {
                Seq s = rangeSeq[k];
                if (k == 0) s.litLength += carry;
                memcpy(w->lits + nlit, src + pos, s.litLength); nlit += s.litLength; w->seqs[nseq++] = s;
                pos += s.litLength + s.matchLength;
            }
            carry = ns ? trailing : carry + trailing;
        }
        memcpy(w->lits + nlit, src + pos, blockEnd - pos); nlit += blockEnd - pos;    /* last literals */
    }
    if (nseq == 0 && nlit == n) {
        /* no match at all: only worth a compressed block if Huffman alone wins; handled below with nbSeq = 0 */
    }
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
/* seqOut: per walk range r (1 KiB), up to 256 triples (litLength, matchLength, offset) at seqOut[(r*256 + k)*3];
 * hdrOut[r*2] = number of sequences, hdrOut[r*2+1] = trailing literals of the range */
int zso_debugWalk(uint32_t *seqOut, uint32_t *hdrOut, const void *src, uint32_t n, int level)
{
    EParams const prm = paramsForLevel(level);
    Work *w = (Work *)malloc(sizeof(Work));
    Seq *tmp = (Seq *)malloc(sizeof(Seq) * (WALK_SIZE / 3 + 8));
    U32 r, nRanges = (n + WALK_SIZE - 1) >> WALK_LOG;
    if (!w || !tmp || n > UNIT_MAX) { free(w); free(tmp); return -1; }
    findCandidates(w, (const BYTE *)src, n, &prm);
    for (r = 0; r < nRanges; r++) {
        U32 const start = r << WALK_LOG;
        U32 const end = (start + WALK_SIZE < n) ? start + WALK_SIZE : n;
        U32 trailing, k;
        U32 const ns = walkRange(w, (const BYTE *)src, n, start, end, &prm, tmp, &trailing);
        hdrOut[r * 2] = ns; hdrOut[r * 2 + 1] = trailing;
        for (k = 0; k < ns; k++) { seqOut[(r * 256 + k) * 3] = tmp[k].litLength; seqOut[(r * 256 + k) * 3 + 1] = tmp[k].matchLength; seqOut[(r * 256 + k) * 3 + 2] = tmp[k].offset; }
    }
    free(w); free(tmp);
    return 0;
}
