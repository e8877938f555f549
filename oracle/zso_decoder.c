/*
 * ORACLE D -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * Plain-C, single-threaded restatement of the reference decoder
 * (epam/Zstandard, C# port of zstd v1.3.4, csharp/src/*.cs).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it.
 *
 * Parity status: PINNED for the constructs exercised by the reference's own two
 * golden vectors (csharp/test/TestDecompress.cs:58-90,
 * java/src/test/java/com/epam/deltix/zstd/TestDecompress.java:8-10; both kept
 * as byte fixtures under tests/golden/).  Huffman literals, RLE/repeat modes,
 * raw/RLE blocks, multi-block and multi-frame input are NOT exercised by any
 * reference test ("parity unpinned" by the reference); those are pinned here
 * against frames produced by upstream libzstd 1.4.8 (tests/golden/gen_fixtures.py).
 * The reference itself (C#/Java) cannot be built or run in this image (no
 * dotnet/mono/java), so there is no oracle/_ref.
 *
 * The reference compiles with size_t = System.UInt32 (ZStdDecompress.cs:14), so
 * MEM_32bits() is always true (Mem.cs:86-89) and the bit container is 32 bits
 * wide (BitStream.cs:311).  This file keeps that model: the bit reader below is
 * a 32-bit container with the reference's reload rules, because they decide
 * where a damaged stream is rejected.
 *
 * Each function cites the reference file:line it follows.
 */
#include "zso_oracle.h"
#include <string.h>
#include <stdlib.h>

typedef uint8_t BYTE;
typedef uint16_t U16;
typedef int16_t S16;
typedef uint32_t U32;
typedef uint64_t U64;

/* ---- coverage counters (which format constructs a decode call exercised) ---- */
/* [0..3] literals type basic/rle/compressed/repeat ; [4] 1-stream huf ; [5] 4-stream huf ; [6] direct weights ; [7] FSE weights
 * [8..11] LL mode basic/rle/compressed/repeat ; [12..15] OF ; [16..19] ML ; [20..22] block raw/rle/compressed
 * [23] frames ; [24] skippable ; [25] checksum ; [26] blocks with nbSeq==0 ; [27] long nbSeq (>=0x7F00) ; [28] repcode used ; [29] multi-block frame */
static uint32_t g_stats[40];   /* [30] sequences [31] literals [32] literal-section bytes [33] sequence-section bytes [34] literal sections decoded by the double-symbol (X4) Huffman decoder */
void zso_statsReset(void) { memset(g_stats, 0, sizeof g_stats); }
void zso_statsGet(uint32_t *out) { memcpy(out, g_stats, 32 * sizeof(uint32_t)); }
void zso_statsGet40(uint32_t *out) { memcpy(out, g_stats, sizeof g_stats); }
#define STAT(i) (g_stats[i]++)

/* ---- error ABI : ZStdErrors.cs:61-100 ---------------------------------- */
#define ERR(code) ((size_t)0 - (size_t)(code))
unsigned zso_isError(size_t code) { return code > ERR(ZSO_maxCode); }      /* ZStdErrors.cs:95-98 */
unsigned zso_errorCode(size_t code) { return zso_isError(code) ? (unsigned)(0 - code) : 0; }

/* ---- Mem.cs:312-365 : little-endian loads -------------------------------- */
static U32 rdLE16(const void *p) { const BYTE *b = p; return (U32)b[0] | ((U32)b[1] << 8); }
static U32 rdLE24(const void *p) { const BYTE *b = p; return rdLE16(b) | ((U32)b[2] << 16); }
static U32 rdLE32(const void *p) { const BYTE *b = p; return rdLE16(b) | (rdLE16(b + 2) << 16); }
static U64 rdLE64(const void *p) { const BYTE *b = p; return (U64)rdLE32(b) | ((U64)rdLE32(b + 4) << 32); }

static U32 highbit32(U32 v) { return 31 - (U32)__builtin_clz(v); }          /* BitStream.cs:205 */

/* ---- format constants : ZStd.cs:386-416,1387-1389 ; ZStdInternal.cs:109-198 */
#define ZSTD_MAGICNUMBER 0xFD2FB528u
#define ZSTD_MAGIC_SKIPPABLE_START 0x184D2A50u
#define ZSTD_WINDOWLOG_ABSOLUTEMIN 10
#define ZSTD_WINDOWLOG_MAX 30                 /* 32-bit build: ZStd.cs:390-392 */
#define ZSTD_BLOCKSIZE_MAX (1u << 17)
#define ZSTD_frameHeaderSize_prefix 5
#define ZSTD_frameHeaderSize_min 6
#define ZSTD_skippableHeaderSize 8
#define ZSTD_blockHeaderSize 3
#define MIN_CBLOCK_SIZE 3
#define WILDCOPY_OVERLENGTH 8
#define LONGNBSEQ 0x7F00
#define MaxML 52
#define MaxLL 35
#define ZSTD_MAGIC_DICTIONARY 0xEC30A437u      /* ZStd.cs */
#define MaxOff 31
#define MaxSeq 52
#define MLFSELog 9
#define LLFSELog 9
#define OffFSELog 8
#define HUF_TABLELOG_MAX 12
#define HUF_SYMBOLVALUE_MAX 255
#define FSE_MIN_TABLELOG 5
#define FSE_TABLELOG_ABSOLUTE_MAX 15
#define FSE_MAX_TABLELOG 12
#define FSE_MAX_SYMBOL_VALUE 255
#define STREAM_ACCUMULATOR_MIN_32 25
#define CONTENTSIZE_UNKNOWN ((U64)0 - 1)
#define CONTENTSIZE_ERROR ((U64)0 - 2)

static const U32 LL_bits[MaxLL + 1] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3,
                                        4,6,7,8,9,10,11,12, 13,14,15,16 };          /* ZStdInternal.cs:158 */
static const U32 ML_bits[MaxML + 1] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
                                        0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11,
                                        12,13,14,15,16 };                              /* ZStdInternal.cs:173 */
static const S16 LL_defaultNorm[MaxLL + 1] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2,
                                               2,3,2,1,1,1,1,1, -1,-1,-1,-1 };        /* ZStdInternal.cs:164 */
static const S16 ML_defaultNorm[MaxML + 1] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                               1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1,
                                               -1,-1,-1,-1,-1 };                      /* ZStdInternal.cs:181 */
static const S16 OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                        -1,-1,-1,-1,-1 };                             /* ZStdInternal.cs:192 */
static const U32 LL_base[MaxLL + 1] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,18,20,22,24,28,32,40,
                                        48,64,0x80,0x100,0x200,0x400,0x800,0x1000, 0x2000,0x4000,0x8000,0x10000 }; /* ZStdDecompress.cs:1081 */
static const U32 ML_base[MaxML + 1] = { 3,4,5,6,7,8,9,10, 11,12,13,14,15,16,17,18, 19,20,21,22,23,24,25,26,
                                        27,28,29,30,31,32,33,34, 35,37,39,41,43,47,51,59, 67,83,99,0x83,0x103,0x203,0x403,0x803,
                                        0x1003,0x2003,0x4003,0x8003,0x10003 };       /* ZStdDecompress.cs:1100 */
static U32 OF_base(U32 c) { return c == 0 ? 0 : (c == 1 ? 1 : (c == 2 ? 1 : ((1u << c) - 3))); }   /* ZStdDecompress.cs:1088 */
/* OF_bits[c] == c : ZStdDecompress.cs:1094 */

/* ======================================================================= *
 *  BitStream.cs : backward bit reader, 32-bit container
 * ======================================================================= */
typedef enum { BIT_unfinished = 0, BIT_endOfBuffer = 1, BIT_completed = 2, BIT_overflow = 3 } BIT_status;
typedef struct {
    U32 bitContainer;
    U32 bitsConsumed;
    const BYTE *ptr, *start, *limitPtr;
} BIT_D;
#define CONTAINER_BYTES 4u                                                    /* BitStream.cs:311 */

/* BitStream.cs:322-378 */
static size_t BIT_init(BIT_D *b, const void *src, size_t srcSize)
{
    if (srcSize < 1) { memset(b, 0, sizeof *b); return ERR(ZSO_srcSize_wrong); }
    b->start = (const BYTE *)src;
    b->limitPtr = b->start + CONTAINER_BYTES;
    if (srcSize >= CONTAINER_BYTES) {
        BYTE last = b->start[srcSize - 1];
        b->ptr = b->start + srcSize - CONTAINER_BYTES;
        b->bitContainer = rdLE32(b->ptr);
        b->bitsConsumed = last ? 8 - highbit32(last) : 0;
        if (last == 0) return ERR(ZSO_GENERIC);
    } else {
        BYTE last = b->start[srcSize - 1];
        b->ptr = b->start;
        b->bitContainer = b->start[0];
        if (srcSize >= 3) b->bitContainer += (U32)b->start[2] << 16;
        if (srcSize >= 2) b->bitContainer += (U32)b->start[1] << 8;
        b->bitsConsumed = last ? 8 - highbit32(last) : 0;
        if (last == 0) return ERR(ZSO_corruption_detected);
        b->bitsConsumed += (U32)(CONTAINER_BYTES - srcSize) * 8;
    }
    return srcSize;
}
/* BitStream.cs:412-416 */
static U32 BIT_look(const BIT_D *b, U32 n) { return ((b->bitContainer << (b->bitsConsumed & 31)) >> 1) >> ((31 - n) & 31); }
/* BitStream.cs:420-425 */
static U32 BIT_lookFast(const BIT_D *b, U32 n) { return (b->bitContainer << (b->bitsConsumed & 31)) >> ((32 - n) & 31); }
static void BIT_skip(BIT_D *b, U32 n) { b->bitsConsumed += n; }
static U32 BIT_read(BIT_D *b, U32 n) { U32 v = BIT_look(b, n); BIT_skip(b, n); return v; }          /* :437 */
static U32 BIT_readFast(BIT_D *b, U32 n) { U32 v = BIT_lookFast(b, n); BIT_skip(b, n); return v; }  /* :445 */
/* BitStream.cs:458-489 */
static BIT_status BIT_reload(BIT_D *b)
{
    if (b->bitsConsumed > CONTAINER_BYTES * 8) return BIT_overflow;
    if (b->ptr >= b->limitPtr) {
        b->ptr -= b->bitsConsumed >> 3;
        b->bitsConsumed &= 7;
        b->bitContainer = rdLE32(b->ptr);
        return BIT_unfinished;
    }
    if (b->ptr == b->start) {
        if (b->bitsConsumed < CONTAINER_BYTES * 8) return BIT_endOfBuffer;
        return BIT_completed;
    }
    {
        U32 nbBytes = b->bitsConsumed >> 3;
        BIT_status r = BIT_unfinished;
        if (b->ptr - nbBytes < b->start) { nbBytes = (U32)(b->ptr - b->start); r = BIT_endOfBuffer; }
        b->ptr -= nbBytes;
        b->bitsConsumed -= nbBytes * 8;
        b->bitContainer = rdLE32(b->ptr);
        return r;
    }
}
/* BitStream.cs:494 */
static unsigned BIT_end(const BIT_D *b) { return (b->ptr == b->start) && (b->bitsConsumed == CONTAINER_BYTES * 8); }

/* ======================================================================= *
 *  EntropyCommon.cs:79-188 : ReadNCount
 * ======================================================================= */
static size_t readNCount(S16 *norm, U32 *maxSVPtr, U32 *tableLogPtr, const void *hdr, size_t hbSize)
{
    const BYTE *const istart = (const BYTE *)hdr;
    const BYTE *const iend = istart + hbSize;
    const BYTE *ip = istart;
    int nbBits, remaining, threshold, bitCount;
    U32 bitStream, charnum = 0;
    int previous0 = 0;

    if (hbSize < 4) return ERR(ZSO_srcSize_wrong);
    bitStream = rdLE32(ip);
    nbBits = (int)(bitStream & 0xF) + FSE_MIN_TABLELOG;
    if (nbBits > FSE_TABLELOG_ABSOLUTE_MAX) return ERR(ZSO_tableLog_tooLarge);
    bitStream >>= 4;
    bitCount = 4;
    *tableLogPtr = (U32)nbBits;
    remaining = (1 << nbBits) + 1;
    threshold = 1 << nbBits;
    nbBits++;

    while ((remaining > 1) & (charnum <= *maxSVPtr)) {
        if (previous0) {
            U32 n0 = charnum;
            while ((bitStream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (ip < iend - 5) { ip += 2; bitStream = rdLE32(ip) >> bitCount; }
                else { bitStream >>= 16; bitCount += 16; }
            }
            while ((bitStream & 3) == 3) { n0 += 3; bitStream >>= 2; bitCount += 2; }
            n0 += bitStream & 3;
            bitCount += 2;
            if (n0 > *maxSVPtr) return ERR(ZSO_maxSymbolValue_tooSmall);
            while (charnum < n0) norm[charnum++] = 0;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) {
                ip += bitCount >> 3; bitCount &= 7; bitStream = rdLE32(ip) >> bitCount;
            } else bitStream >>= 2;
        }
        {
            int const max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (U32)(threshold - 1)) < (U32)max) {
                count = (int)(bitStream & (U32)(threshold - 1));
                bitCount += nbBits - 1;
            } else {
                count = (int)(bitStream & (U32)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bitCount += nbBits;
            }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[charnum++] = (S16)count;
            previous0 = !count;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) {
                ip += bitCount >> 3; bitCount &= 7;
            } else {
                bitCount -= (int)(8 * (iend - 4 - ip));
                ip = iend - 4;
            }
            bitStream = rdLE32(ip) >> (bitCount & 31);
        }
    }
    if (remaining != 1) return ERR(ZSO_corruption_detected);
    if (bitCount > 32) return ERR(ZSO_corruption_detected);
    *maxSVPtr = charnum - 1;
    ip += (bitCount + 7) >> 3;
    return (size_t)(ip - istart);
}

/* ======================================================================= *
 *  FseDecompress.cs : generic FSE (Huffman weight headers only)
 * ======================================================================= */
typedef struct { U16 newState; BYTE symbol; BYTE nbBits; } FSE_decode_t;       /* Fse.cs:604 */
typedef struct { U32 tableLog; U32 fastMode; FSE_decode_t cells[1 << 6]; } FSE_DTable6;

static U32 FSE_TABLESTEP(U32 tableSize) { return (tableSize >> 1) + (tableSize >> 3) + 3; }   /* Fse.cs:714 */

/* FseDecompress.cs:111-181 */
static size_t FSE_buildDTable(FSE_DTable6 *dt, const S16 *norm, U32 maxSymbolValue, U32 tableLog)
{
    U16 symbolNext[FSE_MAX_SYMBOL_VALUE + 1];
    U32 const maxSV1 = maxSymbolValue + 1;
    U32 const tableSize = 1u << tableLog;
    U32 highThreshold = tableSize - 1;
    U32 s;
    if (maxSymbolValue > FSE_MAX_SYMBOL_VALUE) return ERR(ZSO_maxSymbolValue_tooLarge);
    if (tableLog > FSE_MAX_TABLELOG) return ERR(ZSO_tableLog_tooLarge);
    dt->tableLog = tableLog;
    dt->fastMode = 1;
    {
        S16 const largeLimit = (S16)(1 << (tableLog - 1));
        for (s = 0; s < maxSV1; s++) {
            if (norm[s] == -1) { dt->cells[highThreshold--].symbol = (BYTE)s; symbolNext[s] = 1; }
            else { if (norm[s] >= largeLimit) dt->fastMode = 0; symbolNext[s] = (U16)norm[s]; }
        }
    }
    {
        U32 const tableMask = tableSize - 1, step = FSE_TABLESTEP(tableSize);
        U32 position = 0;
        for (s = 0; s < maxSV1; s++) {
            int i;
            for (i = 0; i < norm[s]; i++) {
                dt->cells[position].symbol = (BYTE)s;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
        if (position != 0) return ERR(ZSO_GENERIC);
    }
    {
        U32 u;
        for (u = 0; u < tableSize; u++) {
            BYTE const symbol = dt->cells[u].symbol;
            U32 const nextState = symbolNext[symbol]++;
            dt->cells[u].nbBits = (BYTE)(tableLog - highbit32(nextState));
            dt->cells[u].newState = (U16)((nextState << dt->cells[u].nbBits) - tableSize);
        }
    }
    return 0;
}

typedef struct { U32 state; const FSE_decode_t *table; } FSE_DState;
static void FSE_initDState(FSE_DState *s, BIT_D *b, const FSE_DTable6 *dt)         /* Fse.cs:611 */
{ s->state = BIT_read(b, dt->tableLog); BIT_reload(b); s->table = dt->cells; }
static BYTE FSE_decodeSymbol(FSE_DState *s, BIT_D *b, int fast)                    /* Fse.cs:634-656 */
{
    FSE_decode_t const d = s->table[s->state];
    U32 const low = fast ? BIT_readFast(b, d.nbBits) : BIT_read(b, d.nbBits);
    s->state = d.newState + low;
    return d.symbol;
}

/* FseDecompress.cs:233-295 */
static size_t FSE_decompress_usingDTable(BYTE *dst, size_t maxDstSize, const void *cSrc, size_t cSrcSize, const FSE_DTable6 *dt)
{
    BYTE *const ostart = dst;
    BYTE *op = ostart;
    BYTE *const omax = op + maxDstSize;
    BYTE *const olimit = omax - 3;
    int const fast = (int)dt->fastMode;
    BIT_D bitD;
    FSE_DState s1, s2;
    { size_t const e = BIT_init(&bitD, cSrc, cSrcSize); if (zso_isError(e)) return e; }
    FSE_initDState(&s1, &bitD, dt);
    FSE_initDState(&s2, &bitD, dt);
    for (; (BIT_reload(&bitD) == BIT_unfinished) & (op < olimit); op += 4) {
        op[0] = FSE_decodeSymbol(&s1, &bitD, fast);
        op[1] = FSE_decodeSymbol(&s2, &bitD, fast);
        if (BIT_reload(&bitD) > BIT_unfinished) { op += 2; break; }   /* 12*4+7 > 32 : static test true */
        op[2] = FSE_decodeSymbol(&s1, &bitD, fast);
        op[3] = FSE_decodeSymbol(&s2, &bitD, fast);
    }
    for (;;) {
        if (op > (omax - 2)) return ERR(ZSO_dstSize_tooSmall);
        *op++ = FSE_decodeSymbol(&s1, &bitD, fast);
        if (BIT_reload(&bitD) == BIT_overflow) { *op++ = FSE_decodeSymbol(&s2, &bitD, fast); break; }
        if (op > (omax - 2)) return ERR(ZSO_dstSize_tooSmall);
        *op++ = FSE_decodeSymbol(&s2, &bitD, fast);
        if (BIT_reload(&bitD) == BIT_overflow) { *op++ = FSE_decodeSymbol(&s1, &bitD, fast); break; }
    }
    return (size_t)(op - ostart);
}

/* FseDecompress.cs:310-332 */
static size_t FSE_decompress_wksp(BYTE *dst, size_t dstCapacity, const void *cSrc, size_t cSrcSize, FSE_DTable6 *wksp, U32 maxLog)
{
    const BYTE *ip = (const BYTE *)cSrc;
    S16 counting[FSE_MAX_SYMBOL_VALUE + 1];
    U32 tableLog, maxSymbolValue = FSE_MAX_SYMBOL_VALUE;
    size_t const nc = readNCount(counting, &maxSymbolValue, &tableLog, ip, cSrcSize);
    if (zso_isError(nc)) return nc;
    if (tableLog > maxLog) return ERR(ZSO_tableLog_tooLarge);
    ip += nc; cSrcSize -= nc;
    { size_t const e = FSE_buildDTable(wksp, counting, maxSymbolValue, tableLog); if (zso_isError(e)) return e; }
    return FSE_decompress_usingDTable(dst, dstCapacity, ip, cSrcSize, wksp);
}

/* ======================================================================= *
 *  EntropyCommon.cs:198-269 : ReadStats (Huffman weights)
 * ======================================================================= */
static size_t HUF_readStats(BYTE *huffWeight, size_t hwSize, U32 *rankStats, U32 *nbSymbolsPtr, U32 *tableLogPtr,
                            const void *src, size_t srcSize)
{
    const BYTE *ip = (const BYTE *)src;
    size_t iSize, oSize;
    U32 weightTotal;
    if (!srcSize) return ERR(ZSO_srcSize_wrong);
    iSize = ip[0];
    if (iSize >= 128) {
        STAT(6);
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return ERR(ZSO_srcSize_wrong);
        if (oSize >= hwSize) return ERR(ZSO_corruption_detected);
        ip += 1;
        { U32 n; for (n = 0; n < oSize; n += 2) { huffWeight[n] = ip[n / 2] >> 4; huffWeight[n + 1] = ip[n / 2] & 15; } }
    } else {
        FSE_DTable6 fseWorkspace;
        STAT(7);
        if (iSize + 1 > srcSize) return ERR(ZSO_srcSize_wrong);
        oSize = FSE_decompress_wksp(huffWeight, hwSize - 1, ip + 1, iSize, &fseWorkspace, 6);
        if (zso_isError(oSize)) return oSize;
    }
    memset(rankStats, 0, (HUF_TABLELOG_MAX + 1) * sizeof(U32));
    weightTotal = 0;
    { U32 n; for (n = 0; n < oSize; n++) {
        if (huffWeight[n] >= HUF_TABLELOG_MAX) return ERR(ZSO_corruption_detected);
        rankStats[huffWeight[n]]++;
        weightTotal += (1u << huffWeight[n]) >> 1;
    } }
    if (weightTotal == 0) return ERR(ZSO_corruption_detected);
    {
        U32 const tableLog = highbit32(weightTotal) + 1;
        if (tableLog > HUF_TABLELOG_MAX) return ERR(ZSO_corruption_detected);
        *tableLogPtr = tableLog;
        {
            U32 const total = 1u << tableLog;
            U32 const rest = total - weightTotal;
            U32 const verif = 1u << highbit32(rest);
            U32 const lastWeight = highbit32(rest) + 1;
            if (verif != rest) return ERR(ZSO_corruption_detected);
            huffWeight[oSize] = (BYTE)lastWeight;
            rankStats[lastWeight]++;
        }
    }
    if ((rankStats[1] < 2) || (rankStats[1] & 1)) return ERR(ZSO_corruption_detected);
    *nbSymbolsPtr = (U32)(oSize + 1);
    return iSize + 1;
}

/* ======================================================================= *
 *  HufDecompress.cs : single-symbol ("X2") and double-symbol ("X4") tables, 1/4-stream decoders, and the choice between
 *  them (SelectDecoder, HufDecompress.cs:1082-1095).  Both decoders produce the same bytes from a well-formed stream;
 *  they differ in where a damaged stream is caught, so the reference's dispatch is kept: a new table with 4 streams goes
 *  through SelectDecoder, a new table with 1 stream is always X2 (ZStdDecompress.cs:737), a repeated table is decoded
 *  by the decoder that built it (tableType, HufDecompress.cs:1179-1205), a dictionary's table is X4 (LoadEntropy :2391).
 * ======================================================================= */
typedef struct { BYTE byte; BYTE nbBits; } HUF_DElt;                          /* HufDecompress.cs:109-113 */
typedef struct { U16 sequence; BYTE nbBits; BYTE length; } HUF_DEltX4;        /* HufDecompress.cs:190-195 */
typedef struct { U32 tableLog; int valid; int tableType; HUF_DElt dt[1 << HUF_TABLELOG_MAX]; HUF_DEltX4 dx[1 << HUF_TABLELOG_MAX]; } HUF_DTable;

/* HufDecompress.cs:117-180 */
static size_t HUF_readDTable(HUF_DTable *D, const void *src, size_t srcSize)
{
    U32 rankVal[HUF_TABLELOG_MAX + 4];
    BYTE huffWeight[HUF_SYMBOLVALUE_MAX + 1];
    U32 tableLog = 0, nbSymbols = 0;
    size_t const iSize = HUF_readStats(huffWeight, HUF_SYMBOLVALUE_MAX + 1, rankVal, &nbSymbols, &tableLog, src, srcSize);
    if (zso_isError(iSize)) return iSize;
    if (tableLog > HUF_TABLELOG_MAX + 1) return ERR(ZSO_tableLog_tooLarge);
    D->tableLog = tableLog; D->tableType = 0;
    { U32 n, nextRankStart = 0;
      for (n = 1; n < tableLog + 1; n++) { U32 const cur = nextRankStart; nextRankStart += rankVal[n] << (n - 1); rankVal[n] = cur; } }
    { U32 n;
      for (n = 0; n < nbSymbols; n++) {
          U32 const w = huffWeight[n];
          U32 const length = (1u << w) >> 1;
          U32 u;
          HUF_DElt e; e.byte = (BYTE)n; e.nbBits = (BYTE)(tableLog + 1 - w);
          for (u = rankVal[w]; u < rankVal[w] + length; u++) D->dt[u] = e;
          rankVal[w] += length;
      } }
    return iSize;
}

static BYTE HUF_decodeSymbol(BIT_D *b, const HUF_DElt *dt, U32 dtLog)          /* HufDecompress.cs:193-199 */
{ U32 const val = BIT_lookFast(b, dtLog); BYTE const c = dt[val].byte; BIT_skip(b, dt[val].nbBits); return c; }

/* HufDecompress.cs:222-245 ; 32-bit mode => SYMBOLX2_2 is a no-op, SYMBOLX2_1 active (tableLog max 12) */
static void HUF_decodeStream(BYTE *p, BIT_D *b, BYTE *pEnd, const HUF_DElt *dt, U32 dtLog)
{
    while ((BIT_reload(b) == BIT_unfinished) & (p < pEnd - 3)) {
        *p++ = HUF_decodeSymbol(b, dt, dtLog);   /* _1 */
        *p++ = HUF_decodeSymbol(b, dt, dtLog);   /* _0 */
    }
    while ((BIT_reload(b) == BIT_unfinished) & (p < pEnd)) *p++ = HUF_decodeSymbol(b, dt, dtLog);
    while (p < pEnd) *p++ = HUF_decodeSymbol(b, dt, dtLog);
}

/* HufDecompress.cs:247-264 */
static size_t HUF_decompress1X(BYTE *dst, size_t dstSize, const void *cSrc, size_t cSrcSize, const HUF_DTable *D)
{
    BIT_D bitD;
    { size_t const e = BIT_init(&bitD, cSrc, cSrcSize); if (zso_isError(e)) return e; }
    HUF_decodeStream(dst, &bitD, dst + dstSize, D->dt, D->tableLog);
    if (!BIT_end(&bitD)) return ERR(ZSO_corruption_detected);
    return dstSize;
}

/* HufDecompress.cs:266-358 */
static size_t HUF_decompress4X(BYTE *dst, size_t dstSize, const void *cSrc, size_t cSrcSize, const HUF_DTable *D)
{
    if (cSrcSize < 10) return ERR(ZSO_corruption_detected);
    {
        const BYTE *const istart = (const BYTE *)cSrc;
        BYTE *const ostart = dst;
        BYTE *const oend = ostart + dstSize;
        const HUF_DElt *const dt = D->dt;
        U32 const dtLog = D->tableLog;
        BIT_D b1, b2, b3, b4;
        size_t const length1 = rdLE16(istart), length2 = rdLE16(istart + 2), length3 = rdLE16(istart + 4);
        size_t const length4 = cSrcSize - (length1 + length2 + length3 + 6);
        const BYTE *const istart1 = istart + 6;
        const BYTE *const istart2 = istart1 + length1;
        const BYTE *const istart3 = istart2 + length2;
        const BYTE *const istart4 = istart3 + length3;
        size_t const segmentSize = (dstSize + 3) / 4;
        BYTE *const opStart2 = ostart + segmentSize;
        BYTE *const opStart3 = opStart2 + segmentSize;
        BYTE *const opStart4 = opStart3 + segmentSize;
        BYTE *op1 = ostart, *op2 = opStart2, *op3 = opStart3, *op4 = opStart4;
        U32 endSignal;
        if (length4 > cSrcSize) return ERR(ZSO_corruption_detected);
        { size_t const e = BIT_init(&b1, istart1, length1); if (zso_isError(e)) return e; }
        { size_t const e = BIT_init(&b2, istart2, length2); if (zso_isError(e)) return e; }
        { size_t const e = BIT_init(&b3, istart3, length3); if (zso_isError(e)) return e; }
        { size_t const e = BIT_init(&b4, istart4, length4); if (zso_isError(e)) return e; }
        /* The reference computes endSignal once and does not refresh it inside the loop
         * (HufDecompress.cs:316-340): the loop is bounded by op4 alone. Kept as is. */
        endSignal = (U32)BIT_reload(&b1) | (U32)BIT_reload(&b2) | (U32)BIT_reload(&b3) | (U32)BIT_reload(&b4);
        /* guard (oend - 3) against pointer underflow for tiny dstSize: the comparison is on addresses in the
         * reference; segment pointers beyond oend make it false at once. */
        while ((endSignal == BIT_unfinished) && ((size_t)(op4 - ostart) + 3 < dstSize)) {
            *op1++ = HUF_decodeSymbol(&b1, dt, dtLog); *op2++ = HUF_decodeSymbol(&b2, dt, dtLog);
            *op3++ = HUF_decodeSymbol(&b3, dt, dtLog); *op4++ = HUF_decodeSymbol(&b4, dt, dtLog);
            *op1++ = HUF_decodeSymbol(&b1, dt, dtLog); *op2++ = HUF_decodeSymbol(&b2, dt, dtLog);
            *op3++ = HUF_decodeSymbol(&b3, dt, dtLog); *op4++ = HUF_decodeSymbol(&b4, dt, dtLog);
            BIT_reload(&b1); BIT_reload(&b2); BIT_reload(&b3); BIT_reload(&b4);
        }
        if (op1 > opStart2) return ERR(ZSO_corruption_detected);
        if (op2 > opStart3) return ERR(ZSO_corruption_detected);
        if (op3 > opStart4) return ERR(ZSO_corruption_detected);
        if (opStart4 > oend) return ERR(ZSO_corruption_detected);   /* oracle safety: reference would write out of bounds */
        HUF_decodeStream(op1, &b1, opStart2, dt, dtLog);
        HUF_decodeStream(op2, &b2, opStart3, dt, dtLog);
        HUF_decodeStream(op3, &b3, opStart4, dt, dtLog);
        HUF_decodeStream(op4, &b4, oend, dt, dtLog);
        if (!(BIT_end(&b1) & BIT_end(&b2) & BIT_end(&b3) & BIT_end(&b4))) return ERR(ZSO_corruption_detected);
        return dstSize;
    }
}

/* ---- double-symbol decoder ---- */
typedef struct { BYTE symbol; BYTE weight; } sortedSymbol_t;                   /* HufDecompress.cs:687-691 */

/* HufDecompress.cs:695-741 */
static void HUF_fillDTableX4Level2(HUF_DEltX4 *DTable, U32 sizeLog, U32 consumed, const U32 *rankValOrigin, int minWeight,
                                   const sortedSymbol_t *sortedSymbols, U32 sortedListSize, U32 nbBitsBaseline, U16 baseSeq)
{
    HUF_DEltX4 DElt;
    U32 rankVal[HUF_TABLELOG_MAX + 1];
    memcpy(rankVal, rankValOrigin, sizeof rankVal);
    if (minWeight > 1) {
        U32 i, skipSize = rankVal[minWeight];
        DElt.sequence = baseSeq; DElt.nbBits = (BYTE)consumed; DElt.length = 1;
        for (i = 0; i < skipSize; i++) DTable[i] = DElt;
    }
    {
        U32 s2;
        for (s2 = 0; s2 < sortedListSize; s2++) {
            U32 const symbol = sortedSymbols[s2].symbol, weight = sortedSymbols[s2].weight;
            U32 const nbBits = nbBitsBaseline - weight;
            U32 const length = 1u << (sizeLog - nbBits);
            U32 const start = rankVal[weight];
            U32 i = start;
            U32 const end = start + length;
            DElt.sequence = (U16)(baseSeq + (symbol << 8)); DElt.nbBits = (BYTE)(nbBits + consumed); DElt.length = 2;
            do { DTable[i++] = DElt; } while (i < end);
            rankVal[weight] += length;
        }
    }
}

/* HufDecompress.cs:761-808 */
static void HUF_fillDTableX4(HUF_DEltX4 *DTable, U32 targetLog, const sortedSymbol_t *sortedList, U32 sortedListSize,
                             const U32 *rankStart, const U32 *rankValOrigin, U32 maxWeight, U32 nbBitsBaseline)
{
    U32 rankVal[HUF_TABLELOG_MAX + 1];
    int const scaleLog = (int)(nbBitsBaseline - targetLog);
    U32 const minBits = nbBitsBaseline - maxWeight;
    U32 s2;
    memcpy(rankVal, rankValOrigin, sizeof rankVal);
    for (s2 = 0; s2 < sortedListSize; s2++) {
        U16 const symbol = sortedList[s2].symbol;
        U32 const weight = sortedList[s2].weight;
        U32 const nbBits = nbBitsBaseline - weight;
        U32 const start = rankVal[weight];
        U32 const length = 1u << (targetLog - nbBits);
        if (targetLog - nbBits >= minBits) {
            U32 sortedRank;
            int minWeight = (int)nbBits + scaleLog;
            if (minWeight < 1) minWeight = 1;
            sortedRank = rankStart[minWeight];
            HUF_fillDTableX4Level2(DTable + start, targetLog - nbBits, nbBits, rankValOrigin + nbBits * (HUF_TABLELOG_MAX + 1), minWeight,
                                   sortedList + sortedRank, sortedListSize - sortedRank, nbBitsBaseline, symbol);
        } else {
            HUF_DEltX4 DElt; U32 u; U32 const end = start + length;
            DElt.sequence = symbol; DElt.nbBits = (BYTE)nbBits; DElt.length = 1;
            for (u = start; u < end; u++) DTable[u] = DElt;
        }
        rankVal[weight] += length;
    }
}

/* HufDecompress.cs:813-925 ; the table is always laid out for maxTableLog = HUF_TABLELOG_MAX (hufTable[0] = 12 * 0x1000001, ZStdDecompress.cs:2488) */
static size_t HUF_readDTableX4(HUF_DTable *D, const void *src, size_t srcSize)
{
    U32 const maxTableLog = HUF_TABLELOG_MAX;
    U32 rankVal[HUF_TABLELOG_MAX][HUF_TABLELOG_MAX + 1];
    U32 rankStats[HUF_TABLELOG_MAX + 1 + 3];
    U32 rankStart0[HUF_TABLELOG_MAX + 2];
    U32 *const rankStart = rankStart0 + 1;
    sortedSymbol_t sortedSymbol[HUF_SYMBOLVALUE_MAX + 1];
    BYTE weightList[HUF_SYMBOLVALUE_MAX + 1];
    U32 tableLog = 0, maxW, sizeOfSort, nbSymbols = 0;
    size_t iSize;
    memset(rankVal, 0, sizeof rankVal); memset(rankStats, 0, sizeof rankStats); memset(rankStart0, 0, sizeof rankStart0);
    iSize = HUF_readStats(weightList, HUF_SYMBOLVALUE_MAX + 1, rankStats, &nbSymbols, &tableLog, src, srcSize);
    if (zso_isError(iSize)) return iSize;
    if (tableLog > maxTableLog) return ERR(ZSO_tableLog_tooLarge);
    for (maxW = tableLog; rankStats[maxW] == 0; maxW--) {}
    {
        U32 w, nextRankStart = 0;
        for (w = 1; w < maxW + 1; w++) { U32 const cur = nextRankStart; nextRankStart += rankStats[w]; rankStart[w] = cur; }
        rankStart[0] = nextRankStart;
        sizeOfSort = nextRankStart;
    }
    {
        U32 s2;
        for (s2 = 0; s2 < nbSymbols; s2++) {
            U32 const w = weightList[s2];
            U32 const r = rankStart[w]++;
            sortedSymbol[r].symbol = (BYTE)s2; sortedSymbol[r].weight = (BYTE)w;
        }
        rankStart[0] = 0;
    }
    {
        U32 *const rankVal0 = rankVal[0];
        {
            int const rescale = (int)(maxTableLog - tableLog) - 1;
            U32 nextRankVal = 0, w;
            for (w = 1; w < maxW + 1; w++) { U32 const cur = nextRankVal; nextRankVal += rankStats[w] << (w + rescale); rankVal0[w] = cur; }
        }
        {
            U32 const minBits = tableLog + 1 - maxW;
            U32 consumed;
            for (consumed = minBits; consumed < maxTableLog - minBits + 1; consumed++) {
                U32 *const rankValPtr = rankVal[consumed];
                U32 w;
                for (w = 1; w < maxW + 1; w++) rankValPtr[w] = rankVal0[w] >> consumed;
            }
        }
    }
    HUF_fillDTableX4(D->dx, maxTableLog, sortedSymbol, sizeOfSort, rankStart0, &rankVal[0][0], maxW, tableLog + 1);
    D->tableLog = maxTableLog; D->tableType = 1;
    return iSize;
}

/* HufDecompress.cs:361-367 ; the reference copies two bytes whatever the length */
static U32 HUF_decodeSymbolX4(BYTE *op, BIT_D *b, const HUF_DEltX4 *dt, U32 dtLog)
{
    U32 const val = BIT_lookFast(b, dtLog);
    op[0] = (BYTE)dt[val].sequence; op[1] = (BYTE)(dt[val].sequence >> 8);
    BIT_skip(b, dt[val].nbBits);
    return dt[val].length;
}
/* HufDecompress.cs:369-385 */
static U32 HUF_decodeLastSymbolX4(BYTE *op, BIT_D *b, const HUF_DEltX4 *dt, U32 dtLog)
{
    U32 const val = BIT_lookFast(b, dtLog);
    op[0] = (BYTE)dt[val].sequence;
    if (dt[val].length == 1) BIT_skip(b, dt[val].nbBits);
    else if (b->bitsConsumed < CONTAINER_BYTES * 8) {
        BIT_skip(b, dt[val].nbBits);
        if (b->bitsConsumed > CONTAINER_BYTES * 8) b->bitsConsumed = CONTAINER_BYTES * 8;
    }
    return 1;
}
/* HufDecompress.cs:405-428 ; 32-bit mode: SYMBOLX4_2 is a no-op, SYMBOLX4_1 active (HUF_TABLELOG_MAX <= 12) */
static void HUF_decodeStreamX4(BYTE *p, BIT_D *b, BYTE *pEnd, const HUF_DEltX4 *dt, U32 dtLog)
{
    while ((BIT_reload(b) == BIT_unfinished) & (p + (CONTAINER_BYTES - 1) < pEnd)) {
        p += HUF_decodeSymbolX4(p, b, dt, dtLog);   /* _1 */
        p += HUF_decodeSymbolX4(p, b, dt, dtLog);   /* _0 */
    }
    while ((BIT_reload(b) == BIT_unfinished) & (p + 2 <= pEnd)) p += HUF_decodeSymbolX4(p, b, dt, dtLog);
    while (p + 2 <= pEnd) p += HUF_decodeSymbolX4(p, b, dt, dtLog);
    if (p < pEnd) p += HUF_decodeLastSymbolX4(p, b, dt, dtLog);
}
/* HufDecompress.cs:430-453 */
static size_t HUF_decompress1X4(BYTE *dst, size_t dstSize, const void *cSrc, size_t cSrcSize, const HUF_DTable *D)
{
    BIT_D bitD;
    { size_t const e = BIT_init(&bitD, cSrc, cSrcSize); if (zso_isError(e)) return e; }
    HUF_decodeStreamX4(dst, &bitD, dst + dstSize, D->dx, D->tableLog);
    if (!BIT_end(&bitD)) return ERR(ZSO_corruption_detected);
    return dstSize;
}
/* HufDecompress.cs:455-542.  Unlike the X2 body, this loop refreshes endSignal (:513). */
static size_t HUF_decompress4X4(BYTE *dst, size_t dstSize, const void *cSrc, size_t cSrcSize, const HUF_DTable *D)
{
    if (cSrcSize < 10) return ERR(ZSO_corruption_detected);
    {
        const BYTE *const istart = (const BYTE *)cSrc;
        BYTE *const ostart = dst;
        BYTE *const oend = ostart + dstSize;
        const HUF_DEltX4 *const dt = D->dx;
        U32 const dtLog = D->tableLog;
        BIT_D b1, b2, b3, b4;
        size_t const length1 = rdLE16(istart), length2 = rdLE16(istart + 2), length3 = rdLE16(istart + 4);
        size_t const length4 = cSrcSize - (length1 + length2 + length3 + 6);
        const BYTE *const istart1 = istart + 6;
        const BYTE *const istart2 = istart1 + length1;
        const BYTE *const istart3 = istart2 + length2;
        const BYTE *const istart4 = istart3 + length3;
        size_t const segmentSize = (dstSize + 3) / 4;
        BYTE *const opStart2 = ostart + segmentSize;
        BYTE *const opStart3 = opStart2 + segmentSize;
        BYTE *const opStart4 = opStart3 + segmentSize;
        BYTE *op1 = ostart, *op2 = opStart2, *op3 = opStart3, *op4 = opStart4;
        U32 endSignal;
        if (length4 > cSrcSize) return ERR(ZSO_corruption_detected);
        { size_t const e = BIT_init(&b1, istart1, length1); if (zso_isError(e)) return e; }
        { size_t const e = BIT_init(&b2, istart2, length2); if (zso_isError(e)) return e; }
        { size_t const e = BIT_init(&b3, istart3, length3); if (zso_isError(e)) return e; }
        { size_t const e = BIT_init(&b4, istart4, length4); if (zso_isError(e)) return e; }
        if (opStart4 > oend) return ERR(ZSO_corruption_detected);   /* oracle safety: the reference would write out of bounds */
        endSignal = (U32)BIT_reload(&b1) | (U32)BIT_reload(&b2) | (U32)BIT_reload(&b3) | (U32)BIT_reload(&b4);
        /* streams 1-3 may run past their segment here (caught right after the loop): the literal buffer has the room */
        while ((endSignal == BIT_unfinished) && ((size_t)(op4 - ostart) + (CONTAINER_BYTES - 1) < dstSize)) {
            op1 += HUF_decodeSymbolX4(op1, &b1, dt, dtLog); op2 += HUF_decodeSymbolX4(op2, &b2, dt, dtLog);
            op3 += HUF_decodeSymbolX4(op3, &b3, dt, dtLog); op4 += HUF_decodeSymbolX4(op4, &b4, dt, dtLog);
            op1 += HUF_decodeSymbolX4(op1, &b1, dt, dtLog); op2 += HUF_decodeSymbolX4(op2, &b2, dt, dtLog);
            op3 += HUF_decodeSymbolX4(op3, &b3, dt, dtLog); op4 += HUF_decodeSymbolX4(op4, &b4, dt, dtLog);
            endSignal = (U32)BIT_reload(&b1) | (U32)BIT_reload(&b2) | (U32)BIT_reload(&b3) | (U32)BIT_reload(&b4);
        }
        if (op1 > opStart2) return ERR(ZSO_corruption_detected);
        if (op2 > opStart3) return ERR(ZSO_corruption_detected);
        if (op3 > opStart4) return ERR(ZSO_corruption_detected);
        HUF_decodeStreamX4(op1, &b1, opStart2, dt, dtLog);
        HUF_decodeStreamX4(op2, &b2, opStart3, dt, dtLog);
        HUF_decodeStreamX4(op3, &b3, opStart4, dt, dtLog);
        HUF_decodeStreamX4(op4, &b4, oend, dt, dtLog);
        if (!(BIT_end(&b1) & BIT_end(&b2) & BIT_end(&b3) & BIT_end(&b4))) return ERR(ZSO_corruption_detected);
        return dstSize;
    }
}

/* HufDecompress.cs:1056-1095 : algoTime[Q][single, double] and SelectDecoder (0 = X2, 1 = X4) */
static U32 HUF_selectDecoder(size_t dstSize, size_t cSrcSize)
{
    static const U32 algoTime[16][2][2] = {
        {{0,0},{1,1}}, {{0,0},{1,1}}, {{38,130},{1313,74}}, {{448,128},{1353,74}}, {{556,128},{1353,74}}, {{714,128},{1418,74}},
        {{883,128},{1437,74}}, {{897,128},{1515,75}}, {{926,128},{1613,75}}, {{947,128},{1729,77}}, {{1107,128},{2083,81}},
        {{1177,128},{2379,87}}, {{1242,128},{2415,93}}, {{1349,128},{2644,106}}, {{1455,128},{2422,124}}, {{722,128},{1891,145}} };
    U32 const Q = (cSrcSize >= dstSize) ? 15 : (U32)(cSrcSize * 16 / dstSize);
    U32 const D256 = (U32)(dstSize >> 8);
    U32 const DTime0 = algoTime[Q][0][0] + algoTime[Q][0][1] * D256;
    U32 DTime1 = algoTime[Q][1][0] + algoTime[Q][1][1] * D256;
    DTime1 += DTime1 >> 3;
    return DTime1 < DTime0 ? 1u : 0u;
}

/* ======================================================================= *
 *  ZStdDecompress.cs : sequence tables
 * ======================================================================= */
typedef struct { U16 nextState; BYTE nbAdditionalBits; BYTE nbBits; U32 baseValue; } SeqSymbol;   /* :132-146 */
typedef struct { U32 tableLog; SeqSymbol cells[1 << 9]; } SeqTable;

/* ZStdDecompress.cs:958-1034 */
static void buildFSETable(SeqTable *dt, const S16 *norm, U32 maxSymbolValue, const U32 *baseValue, int baseIsOF,
                          const U32 *nbAdditionalBits, U32 tableLog)
{
    U16 symbolNext[MaxSeq + 1];
    U32 const maxSV1 = maxSymbolValue + 1, tableSize = 1u << tableLog;
    U32 highThreshold = tableSize - 1, s;
    dt->tableLog = tableLog;
    for (s = 0; s < maxSV1; s++) {
        if (norm[s] == -1) { dt->cells[highThreshold--].baseValue = s; symbolNext[s] = 1; }
        else symbolNext[s] = (U16)norm[s];
    }
    {
        U32 const tableMask = tableSize - 1, step = FSE_TABLESTEP(tableSize);
        U32 position = 0;
        for (s = 0; s < maxSV1; s++) {
            int i;
            for (i = 0; i < norm[s]; i++) {
                dt->cells[position].baseValue = s;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    {
        U32 u;
        for (u = 0; u < tableSize; u++) {
            U32 const symbol = dt->cells[u].baseValue;
            U32 const nextState = symbolNext[symbol]++;
            dt->cells[u].nbBits = (BYTE)(tableLog - highbit32(nextState));
            dt->cells[u].nextState = (U16)((nextState << dt->cells[u].nbBits) - tableSize);
            dt->cells[u].nbAdditionalBits = (BYTE)(baseIsOF ? symbol : nbAdditionalBits[symbol]);
            dt->cells[u].baseValue = baseIsOF ? OF_base(symbol) : baseValue[symbol];
        }
    }
}

typedef struct {
    HUF_DTable huf;                 /* entropy.hufTable */
    SeqTable LL, OF, ML;            /* entropy.LLTable / OFTable / MLTable */
    SeqTable LLdef, OFdef, MLdef;   /* LL/OF/ML_defaultDTable (:833,873,897), rebuilt from the default norms */
    const SeqTable *LLptr, *OFptr, *MLptr;
    U32 rep[3];
    U32 litEntropy, fseEntropy;
    const BYTE *litPtr; size_t litSize;
    BYTE *litBuffer;                /* ZSTD_BLOCKSIZE_MAX + WILDCOPY_OVERLENGTH, :261 */
    const BYTE *base;               /* start of this frame's output */
    const BYTE *vBase, *dictEnd;    /* dictionary content as the segment in front of it: [dictEnd - (base - vBase), dictEnd)  (:2366-2372, :1911-1920) */
    const BYTE *dictContent; size_t dictContentSize; U32 dictIDLoaded;   /* what decompress_insertDictionary left for the frame */
    /* frame params */
    U64 frameContentSize, windowSize; U32 checksumFlag, dictID, headerSize;
} DCtx;

/* ZStdDecompress.cs:1040-1079 */
static size_t buildSeqTable(SeqTable *space, const SeqTable **ptr, U32 type, U32 max, U32 maxLog,
                            const void *src, size_t srcSize, const U32 *baseValue, int baseIsOF, const U32 *nbAddBits,
                            const SeqTable *defaultTable, U32 flagRepeatTable)
{
    STAT((baseIsOF ? 12 : (max == MaxLL ? 8 : 16)) + type);
    switch (type) {
    case 1: /* set_rle : :937-955 */
        if (!srcSize) return ERR(ZSO_srcSize_wrong);
        if (*(const BYTE *)src > max) return ERR(ZSO_corruption_detected);
        {
            U32 const symbol = *(const BYTE *)src;
            space->tableLog = 0;
            space->cells[0].nbBits = 0; space->cells[0].nextState = 0;
            space->cells[0].nbAdditionalBits = (BYTE)(baseIsOF ? symbol : nbAddBits[symbol]);
            space->cells[0].baseValue = baseIsOF ? OF_base(symbol) : baseValue[symbol];
        }
        *ptr = space;
        return 1;
    case 0: /* set_basic */
        *ptr = defaultTable;
        return 0;
    case 3: /* set_repeat */
        if (!flagRepeatTable) return ERR(ZSO_corruption_detected);
        return 0;
    case 2: /* set_compressed */
        {
            U32 tableLog;
            S16 norm[MaxSeq + 1];
            size_t const headerSize = readNCount(norm, &max, &tableLog, src, srcSize);
            if (zso_isError(headerSize)) return ERR(ZSO_corruption_detected);
            if (tableLog > maxLog) return ERR(ZSO_corruption_detected);
            buildFSETable(space, norm, max, baseValue, baseIsOF, nbAddBits, tableLog);
            *ptr = space;
            return headerSize;
        }
    default:
        return ERR(ZSO_GENERIC);
    }
}

/* ZStdDecompress.cs:1110-1180 */
static size_t decodeSeqHeaders(DCtx *d, int *nbSeqPtr, const void *src, size_t srcSize)
{
    const BYTE *const istart = (const BYTE *)src;
    const BYTE *const iend = istart + srcSize;
    const BYTE *ip = istart;
    if (srcSize < 1) return ERR(ZSO_srcSize_wrong);
    {
        int nbSeq = *ip++;
        if (!nbSeq) { *nbSeqPtr = 0; STAT(26); return 1; }
        if (nbSeq > 0x7F) {
            if (nbSeq == 0xFF) { if (ip + 2 > iend) return ERR(ZSO_srcSize_wrong); nbSeq = (int)rdLE16(ip) + LONGNBSEQ; ip += 2; STAT(27); }
            else { if (ip >= iend) return ERR(ZSO_srcSize_wrong); nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
        }
        *nbSeqPtr = nbSeq;
    }
    if (ip + 4 > iend) return ERR(ZSO_srcSize_wrong);
    {
        U32 const LLtype = *ip >> 6, OFtype = (*ip >> 4) & 3, MLtype = (*ip >> 2) & 3;
        ip++;
        { size_t const h = buildSeqTable(&d->LL, &d->LLptr, LLtype, MaxLL, LLFSELog, ip, (size_t)(iend - ip), LL_base, 0, LL_bits, &d->LLdef, d->fseEntropy);
          if (zso_isError(h)) return ERR(ZSO_corruption_detected); ip += h; }
        { size_t const h = buildSeqTable(&d->OF, &d->OFptr, OFtype, MaxOff, OffFSELog, ip, (size_t)(iend - ip), NULL, 1, NULL, &d->OFdef, d->fseEntropy);
          if (zso_isError(h)) return ERR(ZSO_corruption_detected); ip += h; }
        { size_t const h = buildSeqTable(&d->ML, &d->MLptr, MLtype, MaxML, MLFSELog, ip, (size_t)(iend - ip), ML_base, 0, ML_bits, &d->MLdef, d->fseEntropy);
          if (zso_isError(h)) return ERR(ZSO_corruption_detected); ip += h; }
    }
    return (size_t)(ip - istart);
}

/* ZStdDecompress.cs:683-821 */
static size_t decodeLiteralsBlock(DCtx *d, const void *src, size_t srcSize)
{
    const BYTE *const istart = (const BYTE *)src;
    if (srcSize < MIN_CBLOCK_SIZE) return ERR(ZSO_corruption_detected);
    {
        U32 const litEncType = istart[0] & 3;
        STAT(litEncType);
        switch (litEncType) {
        case 3: /* set_repeat */
            if (d->litEntropy == 0) return ERR(ZSO_dictionary_corrupted);
            /* fall-through */
        case 2: /* set_compressed */
            if (srcSize < 5) return ERR(ZSO_corruption_detected);
            {
                size_t lhSize, litSize, litCSize;
                int singleStream = 0;
                U32 const lhlCode = (istart[0] >> 2) & 3;
                U32 const lhc = rdLE32(istart);
                size_t r;
                switch (lhlCode) {
                default: singleStream = !lhlCode; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; break;
                case 2: lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; break;
                case 3: lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((U32)istart[4] << 10); break;
                }
                if (litSize > ZSTD_BLOCKSIZE_MAX) return ERR(ZSO_corruption_detected);
                if (litCSize + lhSize > srcSize) return ERR(ZSO_corruption_detected);
                STAT(singleStream ? 4 : 5);
                if (litEncType == 3) {               /* HufDecompress.cs:1179-1205: the decoder that built the table */
                    if (d->huf.tableType) r = singleStream ? HUF_decompress1X4(d->litBuffer, litSize, istart + lhSize, litCSize, &d->huf)
                                                           : HUF_decompress4X4(d->litBuffer, litSize, istart + lhSize, litCSize, &d->huf);
                    else r = singleStream ? HUF_decompress1X(d->litBuffer, litSize, istart + lhSize, litCSize, &d->huf)
                                          : HUF_decompress4X(d->litBuffer, litSize, istart + lhSize, litCSize, &d->huf);
                } else if (singleStream) {           /* HufDecompress.cs:1186-1197 */
                    size_t const hSize = HUF_readDTable(&d->huf, istart + lhSize, litCSize);
                    if (zso_isError(hSize)) r = hSize;
                    else if (hSize >= litCSize) r = ERR(ZSO_srcSize_wrong);
                    else r = HUF_decompress1X(d->litBuffer, litSize, istart + lhSize + hSize, litCSize - hSize, &d->huf);
                } else {                             /* HufDecompress.cs:1208-1220 (SelectDecoder) + :647-660 / :984-997 */
                    if (litSize == 0) r = ERR(ZSO_dstSize_tooSmall);
                    else if (litCSize == 0) r = ERR(ZSO_corruption_detected);
                    else if (HUF_selectDecoder(litSize, litCSize)) {
                        size_t const hSize = HUF_readDTableX4(&d->huf, istart + lhSize, litCSize);
                        STAT(34);
                        if (zso_isError(hSize)) r = hSize;
                        else if (hSize >= litCSize) r = ERR(ZSO_srcSize_wrong);
                        else r = HUF_decompress4X4(d->litBuffer, litSize, istart + lhSize + hSize, litCSize - hSize, &d->huf);
                    } else {
                        size_t const hSize = HUF_readDTable(&d->huf, istart + lhSize, litCSize);
                        if (zso_isError(hSize)) r = hSize;
                        else if (hSize >= litCSize) r = ERR(ZSO_srcSize_wrong);
                        else r = HUF_decompress4X(d->litBuffer, litSize, istart + lhSize + hSize, litCSize - hSize, &d->huf);
                    }
                }
                if (zso_isError(r)) return ERR(ZSO_corruption_detected);
                d->litPtr = d->litBuffer; d->litSize = litSize; d->litEntropy = 1;
                memset(d->litBuffer + d->litSize, 0, WILDCOPY_OVERLENGTH);
                return litCSize + lhSize;
            }
        case 0: /* set_basic */
            {
                size_t litSize, lhSize;
                U32 const lhlCode = (istart[0] >> 2) & 3;
                switch (lhlCode) {
                default: lhSize = 1; litSize = istart[0] >> 3; break;
                case 1: lhSize = 2; litSize = rdLE16(istart) >> 4; break;
                case 3: lhSize = 3; litSize = rdLE24(istart) >> 4; break;
                }
                if (lhSize + litSize + WILDCOPY_OVERLENGTH > srcSize) {
                    if (litSize + lhSize > srcSize) return ERR(ZSO_corruption_detected);
                    memcpy(d->litBuffer, istart + lhSize, litSize);
                    d->litPtr = d->litBuffer; d->litSize = litSize;
                    memset(d->litBuffer + d->litSize, 0, WILDCOPY_OVERLENGTH);
                    return lhSize + litSize;
                }
                d->litPtr = istart + lhSize; d->litSize = litSize;
                return lhSize + litSize;
            }
        case 1: /* set_rle */
            {
                U32 const lhlCode = (istart[0] >> 2) & 3;
                size_t litSize, lhSize;
                switch (lhlCode) {
                default: lhSize = 1; litSize = istart[0] >> 3; break;
                case 1: lhSize = 2; litSize = rdLE16(istart) >> 4; break;
                case 3: lhSize = 3; litSize = rdLE24(istart) >> 4; if (srcSize < 4) return ERR(ZSO_corruption_detected); break;
                }
                if (litSize > ZSTD_BLOCKSIZE_MAX) return ERR(ZSO_corruption_detected);
                memset(d->litBuffer, istart[lhSize], litSize + WILDCOPY_OVERLENGTH);
                d->litPtr = d->litBuffer; d->litSize = litSize;
                return lhSize + 1;
            }
        }
    }
    return ERR(ZSO_corruption_detected);
}

/* ======================================================================= *
 *  ZStdDecompress.cs:1443-1553 : sequence decode (regular offsets, 32-bit mode)
 * ======================================================================= */
typedef struct { U32 state; const SeqSymbol *table; } FseState;
typedef struct { BIT_D DStream; FseState stateLL, stateOffb, stateML; U32 prevOffset[3]; } SeqState;
typedef struct { U32 litLength, matchLength, offset; } Seq;

static void initFseState(FseState *s, BIT_D *b, const SeqTable *dt)              /* :1443-1452 */
{ s->state = BIT_read(b, dt->tableLog); BIT_reload(b); s->table = dt->cells; }
static void updateFseState(FseState *s, BIT_D *b)                                /* :1454-1460 */
{ SeqSymbol const d = s->table[s->state]; U32 const low = BIT_read(b, d.nbBits); s->state = d.nextState + low; }

/* :1473-1553.  longOffsets (32-bit mode) is set when windowSize > 2^25 (:1878). */
static Seq decodeSequence(SeqState *st, int longOffsets)
{
    Seq seq;
    U32 const llBits = st->stateLL.table[st->stateLL.state].nbAdditionalBits;
    U32 const mlBits = st->stateML.table[st->stateML.state].nbAdditionalBits;
    U32 const ofBits = st->stateOffb.table[st->stateOffb.state].nbAdditionalBits;
    U32 const llBase = st->stateLL.table[st->stateLL.state].baseValue;
    U32 const mlBase = st->stateML.table[st->stateML.state].baseValue;
    U32 const ofBase = st->stateOffb.table[st->stateOffb.state].baseValue;
    {
        U32 offset;
        if (!ofBits) offset = 0;
        else if (longOffsets && ofBits >= STREAM_ACCUMULATOR_MIN_32) {
            U32 const avail = 32 - st->DStream.bitsConsumed;
            U32 const extraBits = ofBits - (ofBits < avail ? ofBits : avail);
            offset = ofBase + (BIT_readFast(&st->DStream, ofBits - extraBits) << extraBits);
            BIT_reload(&st->DStream);
            if (extraBits) offset += BIT_readFast(&st->DStream, extraBits);
        }
        else { offset = ofBase + BIT_readFast(&st->DStream, ofBits); BIT_reload(&st->DStream); }
        if (ofBits <= 1) {
            STAT(28);
            offset += (llBase == 0);
            if (offset) {
                U32 temp = (offset == 3) ? st->prevOffset[0] - 1 : st->prevOffset[offset];
                temp += !temp;
                if (offset != 1) st->prevOffset[2] = st->prevOffset[1];
                st->prevOffset[1] = st->prevOffset[0];
                st->prevOffset[0] = offset = temp;
            } else offset = st->prevOffset[0];
        } else {
            st->prevOffset[2] = st->prevOffset[1];
            st->prevOffset[1] = st->prevOffset[0];
            st->prevOffset[0] = offset;
        }
        seq.offset = offset;
    }
    seq.matchLength = mlBase + ((mlBits > 0) ? BIT_readFast(&st->DStream, mlBits) : 0);
    if (mlBits + llBits >= STREAM_ACCUMULATOR_MIN_32 - 5) BIT_reload(&st->DStream);
    seq.litLength = llBase + ((llBits > 0) ? BIT_readFast(&st->DStream, llBits) : 0);
    BIT_reload(&st->DStream);
    updateFseState(&st->stateLL, &st->DStream);
    updateFseState(&st->stateML, &st->DStream);
    BIT_reload(&st->DStream);
    updateFseState(&st->stateOffb, &st->DStream);
    return seq;
}

/* :1265-1352 (+ :1212-1260).  The reference copies with 8-byte wild copies; the bytes it
 * leaves in [op, oMatchEnd) are those of a plain overlap-safe forward byte copy, which is
 * what is restated here together with every check the reference makes. */
static size_t execSequence(BYTE *op, BYTE *const oend, Seq seq, const BYTE **litPtr, const BYTE *const litLimit, const BYTE *const base,
                           const BYTE *const vBase, const BYTE *const dictEnd)
{
    BYTE *const oLitEnd = op + seq.litLength;
    size_t const sequenceLength = (size_t)seq.litLength + seq.matchLength;
    const BYTE *const iLitEnd = *litPtr + seq.litLength;
    const BYTE *match;
    if (sequenceLength > (size_t)(oend - op)) return ERR(ZSO_dstSize_tooSmall);
    if (seq.litLength > (size_t)(litLimit - *litPtr)) return ERR(ZSO_corruption_detected);
    memcpy(op, *litPtr, seq.litLength);
    *litPtr = iLitEnd;
    if (seq.offset > (size_t)(oLitEnd - base)) {              /* offset beyond the prefix: into the dictionary segment, :1290-1315 */
        size_t const beyond = seq.offset - (size_t)(oLitEnd - base);
        if (seq.offset > (size_t)(oLitEnd - vBase)) return ERR(ZSO_corruption_detected);   /* no dictionary: vBase == base, :1293 */
        match = dictEnd - beyond;
        if (seq.matchLength <= beyond) { memmove(oLitEnd, match, seq.matchLength); return sequenceLength; }
        memmove(oLitEnd, match, beyond);                      /* spans the dictionary end and the start of the prefix */
        { BYTE *o = oLitEnd + beyond; const BYTE *m = base; U32 i, rest = seq.matchLength - (U32)beyond; for (i = 0; i < rest; i++) o[i] = m[i]; }
        return sequenceLength;
    }
    match = oLitEnd - seq.offset;
    { U32 i; for (i = 0; i < seq.matchLength; i++) oLitEnd[i] = match[i]; }
    return sequenceLength;
}

/* :1555-1608 */
static size_t decompressSequences(DCtx *d, void *dst, size_t maxDstSize, const void *seqStart, size_t seqSize, int nbSeq)
{
    const BYTE *ip = (const BYTE *)seqStart;
    BYTE *const ostart = (BYTE *)dst;
    BYTE *const oend = ostart + maxDstSize;
    BYTE *op = ostart;
    const BYTE *litPtr = d->litPtr;
    const BYTE *const litEnd = litPtr + d->litSize;
    if (nbSeq) {
        SeqState st;
        U32 i;
        d->fseEntropy = 1;
        for (i = 0; i < 3; i++) st.prevOffset[i] = d->rep[i];
        { size_t const e = BIT_init(&st.DStream, ip, seqSize); if (zso_isError(e)) return ERR(ZSO_corruption_detected); }
        initFseState(&st.stateLL, &st.DStream, d->LLptr);
        initFseState(&st.stateOffb, &st.DStream, d->OFptr);
        initFseState(&st.stateML, &st.DStream, d->MLptr);
        for (; (BIT_reload(&st.DStream) <= BIT_completed) && nbSeq;) {
            nbSeq--;
            {
                Seq const sequence = decodeSequence(&st, d->windowSize > (1ULL << STREAM_ACCUMULATOR_MIN_32));
                size_t const one = execSequence(op, oend, sequence, &litPtr, litEnd, d->base, d->vBase, d->dictEnd);
                if (zso_isError(one)) return one;
                op += one;
            }
        }
        if (nbSeq) return ERR(ZSO_corruption_detected);
        for (i = 0; i < 3; i++) d->rep[i] = st.prevOffset[i];
    }
    {
        size_t const lastLLSize = (size_t)(litEnd - litPtr);
        if (lastLLSize > (size_t)(oend - op)) return ERR(ZSO_dstSize_tooSmall);
        memcpy(op, litPtr, lastLLSize);
        op += lastLLSize;
    }
    return (size_t)(op - ostart);
}

/* :1868-1909.  The prefetching variant (:1709, chosen by a heuristic when windowSize > 2^24)
 * regenerates the same bytes as the plain loop and is not restated. */
static size_t decompressBlock_internal(DCtx *d, void *dst, size_t dstCapacity, const void *src, size_t srcSize)
{
    const BYTE *ip = (const BYTE *)src;
    if (srcSize >= ZSTD_BLOCKSIZE_MAX) return ERR(ZSO_srcSize_wrong);
    { size_t const litCSize = decodeLiteralsBlock(d, src, srcSize); if (zso_isError(litCSize)) return litCSize; ip += litCSize; srcSize -= litCSize;
      g_stats[32] += (U32)litCSize; g_stats[31] += (U32)d->litSize; }
    {
        int nbSeq = 0;
        size_t const seqHSize = decodeSeqHeaders(d, &nbSeq, ip, srcSize);
        if (zso_isError(seqHSize)) return seqHSize;
        ip += seqHSize; srcSize -= seqHSize;
        g_stats[30] += (U32)nbSeq; g_stats[33] += (U32)(seqHSize + srcSize);
        return decompressSequences(d, dst, dstCapacity, ip, srcSize, nbSeq);
    }
}

/* ======================================================================= *
 *  XxHash.cs : XXH64, seed 0, one-shot over the frame's content.
 *  (The reference streams it block by block, :1029-1093; the digest is the same.)
 * ======================================================================= */
#define P1 11400714785074694791ULL
#define P2 14029467366897019727ULL
#define P3 1609587929392839161ULL
#define P4 9650029242287828579ULL
#define P5 2870177450012600261ULL
static U64 rotl64(U64 x, int r) { return (x << r) | (x >> (64 - r)); }
static U64 xxround(U64 acc, U64 in) { acc += in * P2; acc = rotl64(acc, 31); acc *= P1; return acc; }    /* :744-750 */
static U64 xxmerge(U64 acc, U64 v) { v = xxround(0, v); acc ^= v; acc = acc * P1 + P4; return acc; }     /* :752-758 */
uint64_t zso_xxh64(const void *input, size_t len, uint64_t seed)
{
    const BYTE *p = (const BYTE *)input;
    const BYTE *const bEnd = p + len;
    U64 h64;
    if (len >= 32) {
        const BYTE *const limit = bEnd - 32;
        U64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed + 0, v4 = seed - P1;                          /* :896-905 */
        do { v1 = xxround(v1, rdLE64(p)); p += 8; v2 = xxround(v2, rdLE64(p)); p += 8;
             v3 = xxround(v3, rdLE64(p)); p += 8; v4 = xxround(v4, rdLE64(p)); p += 8; } while (p <= limit);
        h64 = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);                          /* :1119 */
        h64 = xxmerge(h64, v1); h64 = xxmerge(h64, v2); h64 = xxmerge(h64, v3); h64 = xxmerge(h64, v4);
    } else h64 = seed + P5;
    h64 += (U64)len;
    while (p + 8 <= bEnd) { U64 const k1 = xxround(0, rdLE64(p)); h64 ^= k1; h64 = rotl64(h64, 27) * P1 + P4; p += 8; }
    if (p + 4 <= bEnd) { h64 ^= (U64)rdLE32(p) * P1; h64 = rotl64(h64, 23) * P2 + P3; p += 4; }
    while (p < bEnd) { h64 ^= (*p) * P5; h64 = rotl64(h64, 11) * P1; p++; }
    h64 ^= h64 >> 33; h64 *= P2; h64 ^= h64 >> 29; h64 *= P3; h64 ^= h64 >> 32;
    return h64;
}

/* ======================================================================= *
 *  ZStdDecompress.cs : frame layer
 * ======================================================================= */
static const size_t ZSTD_fcs_fieldSize[4] = { 0, 2, 4, 8 };
static const size_t ZSTD_did_fieldSize[4] = { 0, 1, 2, 4 };

/* :389-403 */
static size_t frameHeaderSize_internal(const void *src, size_t srcSize)
{
    size_t const minInputSize = ZSTD_frameHeaderSize_prefix;
    if (srcSize < minInputSize) return ERR(ZSO_srcSize_wrong);
    {
        U32 const fhd = ((const BYTE *)src)[minInputSize - 1];
        U32 const dictID = fhd & 3;
        int const singleSegment = (fhd >> 5) & 1;
        U32 const fcsId = fhd >> 6;
        return minInputSize + !singleSegment + ZSTD_did_fieldSize[dictID] + ZSTD_fcs_fieldSize[fcsId] + (singleSegment && !fcsId);
    }
}

typedef struct { U64 frameContentSize, windowSize; U32 blockSizeMax, frameType, headerSize, dictID, checksumFlag; } FrameHeader;  /* FrameHeader.cs:24-47 */

/* :421-499 */
static size_t getFrameHeader(FrameHeader *zfh, const void *src, size_t srcSize)
{
    const BYTE *ip = (const BYTE *)src;
    size_t const minInputSize = ZSTD_frameHeaderSize_prefix;
    if (srcSize < minInputSize) return minInputSize;
    if (rdLE32(src) != ZSTD_MAGICNUMBER) {
        if ((rdLE32(src) & 0xFFFFFFF0u) == ZSTD_MAGIC_SKIPPABLE_START) {
            if (srcSize < ZSTD_skippableHeaderSize) return ZSTD_skippableHeaderSize;
            memset(zfh, 0, sizeof *zfh);
            zfh->frameContentSize = rdLE32((const BYTE *)src + 4);
            zfh->frameType = 1;
            return 0;
        }
        return ERR(ZSO_prefix_unknown);
    }
    { size_t const fhsize = frameHeaderSize_internal(src, srcSize); if (srcSize < fhsize) return fhsize; zfh->headerSize = (U32)fhsize; }
    {
        U32 const fhdByte = ip[minInputSize - 1];
        size_t pos = minInputSize;
        U32 const dictIDSizeCode = fhdByte & 3, checksumFlag = (fhdByte >> 2) & 1, singleSegment = (fhdByte >> 5) & 1, fcsID = fhdByte >> 6;
        U64 windowSize = 0, frameContentSize = CONTENTSIZE_UNKNOWN;
        U32 dictID = 0;
        if (fhdByte & 0x08) return ERR(ZSO_frameParameter_unsupported);
        if (!singleSegment) {
            U32 const wlByte = ip[pos++];
            U32 const windowLog = (wlByte >> 3) + ZSTD_WINDOWLOG_ABSOLUTEMIN;
            if (windowLog > ZSTD_WINDOWLOG_MAX) return ERR(ZSO_frameParameter_windowTooLarge);
            windowSize = 1ULL << windowLog;
            windowSize += (windowSize >> 3) * (wlByte & 7);
        }
        switch (dictIDSizeCode) {
        default: break;
        case 1: dictID = ip[pos]; pos++; break;
        case 2: dictID = rdLE16(ip + pos); pos += 2; break;
        case 3: dictID = rdLE32(ip + pos); pos += 4; break;
        }
        switch (fcsID) {
        default: if (singleSegment) frameContentSize = ip[pos]; break;
        case 1: frameContentSize = rdLE16(ip + pos) + 256; break;
        case 2: frameContentSize = rdLE32(ip + pos); break;
        case 3: frameContentSize = rdLE64(ip + pos); break;
        }
        if (singleSegment) windowSize = frameContentSize;
        zfh->frameType = 0;
        zfh->frameContentSize = frameContentSize;
        zfh->windowSize = windowSize;
        zfh->blockSizeMax = (U32)(windowSize < ZSTD_BLOCKSIZE_MAX ? windowSize : ZSTD_BLOCKSIZE_MAX);
        zfh->dictID = dictID;
        zfh->checksumFlag = checksumFlag;
    }
    return 0;
}

/* :518-532, :617-622 */
unsigned long long zso_getDecompressedSize(const void *src, size_t srcSize)
{
    FrameHeader zfh;
    U64 ret;
    if (getFrameHeader(&zfh, src, srcSize) != 0) ret = CONTENTSIZE_ERROR;
    else if (zfh.frameType == 1) ret = 0;
    else ret = zfh.frameContentSize;
    return (ret >= CONTENTSIZE_ERROR) ? 0 : ret;
}

/* :2478-2499 */
static void decompressBegin(DCtx *d)
{
    d->base = NULL; d->vBase = NULL; d->dictEnd = NULL;
    d->dictContent = NULL; d->dictContentSize = 0; d->dictIDLoaded = 0;
    d->litEntropy = d->fseEntropy = 0;
    d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;          /* repStartValue, ZStdInternal.cs:111 */
    d->LLptr = &d->LL; d->MLptr = &d->ML; d->OFptr = &d->OF;
    d->huf.tableLog = 0;
}

/* :2008-2091 */
static size_t decompressFrame(DCtx *d, void *dst, size_t dstCapacity, const void **srcPtr, size_t *srcSizePtr)
{
    const BYTE *ip = (const BYTE *)(*srcPtr);
    BYTE *const ostart = (BYTE *)dst;
    BYTE *const oend = ostart + dstCapacity;
    BYTE *op = ostart;
    size_t remainingSize = *srcSizePtr;
    if (remainingSize < ZSTD_frameHeaderSize_min + ZSTD_blockHeaderSize) return ERR(ZSO_srcSize_wrong);
    {
        size_t const fhs = frameHeaderSize_internal(ip, ZSTD_frameHeaderSize_prefix);
        FrameHeader zfh;
        size_t r;
        if (zso_isError(fhs)) return fhs;
        if (remainingSize < fhs + ZSTD_blockHeaderSize) return ERR(ZSO_srcSize_wrong);
        r = getFrameHeader(&zfh, ip, fhs);                   /* DecodeFrameHeader :626-637 */
        if (zso_isError(r)) return r;
        if (r > 0) return ERR(ZSO_srcSize_wrong);
        if (zfh.dictID != 0 && d->dictIDLoaded != zfh.dictID) return ERR(ZSO_dictionary_wrong);   /* :632-634 */
        d->frameContentSize = zfh.frameContentSize; d->windowSize = zfh.windowSize; d->checksumFlag = zfh.checksumFlag;
        ip += fhs; remainingSize -= fhs;
    }
    /* RefDictContent + CheckContinuity (:2366-2372, :1911-1920): the dictionary content is the segment right in front of this
     * frame's output; frames of one call do not see each other (DecompressBegin_usingDict runs per frame, :2143) */
    d->base = ostart; d->vBase = ostart - d->dictContentSize; d->dictEnd = d->dictContent + d->dictContentSize;
    if (!d->dictContent) { d->vBase = ostart; d->dictEnd = NULL; }
    for (;;) {
        size_t decodedSize;
        U32 lastBlock, blockType, origSize;
        size_t cBlockSize;
        if (remainingSize < ZSTD_blockHeaderSize) return ERR(ZSO_srcSize_wrong);      /* GetcBlockSize :646-659 */
        {
            U32 const h = rdLE24(ip);
            U32 const cSize = h >> 3;
            lastBlock = h & 1; blockType = (h >> 1) & 3; origSize = cSize;
            if (blockType == 1) cBlockSize = 1;
            else if (blockType == 3) return ERR(ZSO_corruption_detected);
            else cBlockSize = cSize;
        }
        ip += ZSTD_blockHeaderSize; remainingSize -= ZSTD_blockHeaderSize;
        if (cBlockSize > remainingSize) return ERR(ZSO_srcSize_wrong);
        STAT(20 + blockType);
        if (!lastBlock) STAT(29);
        switch (blockType) {
        case 2: decodedSize = decompressBlock_internal(d, op, (size_t)(oend - op), ip, cBlockSize); break;
        case 0: if (cBlockSize > (size_t)(oend - op)) decodedSize = ERR(ZSO_dstSize_tooSmall);          /* :662-667 */
                else { memcpy(op, ip, cBlockSize); decodedSize = cBlockSize; } break;
        case 1: if (origSize > (size_t)(oend - op)) decodedSize = ERR(ZSO_dstSize_tooSmall);            /* :1945-1950 */
                else { memset(op, *ip, origSize); decodedSize = origSize; } break;
        default: return ERR(ZSO_corruption_detected);
        }
        if (zso_isError(decodedSize)) return decodedSize;
        op += decodedSize; ip += cBlockSize; remainingSize -= cBlockSize;
        if (lastBlock) break;
    }
    if (d->frameContentSize != CONTENTSIZE_UNKNOWN) {
        if ((U64)(op - ostart) != d->frameContentSize) return ERR(ZSO_corruption_detected);
    }
    if (d->checksumFlag) {
        U32 const checkCalc = (U32)zso_xxh64(ostart, (size_t)(op - ostart), 0);
        STAT(25);
        if (remainingSize < 4) return ERR(ZSO_checksum_wrong);
        if (rdLE32(ip) != checkCalc) return ERR(ZSO_checksum_wrong);
        ip += 4; remainingSize -= 4;
    }
    *srcPtr = ip; *srcSizePtr = remainingSize;
    return (size_t)(op - ostart);
}

/* LoadEntropy :2378-2450.  The dictionary's Huffman table is read in the double-symbol layout (HUF_readDTableX4_wksp, :2391): a block
 * that repeats it is decoded by the X4 decoder.  @return bytes read (magic and dictID included) or an error */
static size_t loadEntropy(DCtx *d, const void *dict, size_t dictSize)
{
    const BYTE *dictPtr = (const BYTE *)dict;
    const BYTE *const dictEnd = dictPtr + dictSize;
    S16 norm[MaxSeq + 1];
    if (dictSize <= 8) return ERR(ZSO_dictionary_corrupted);
    dictPtr += 8;
    { size_t const hSize = HUF_readDTableX4(&d->huf, dictPtr, (size_t)(dictEnd - dictPtr));
      if (zso_isError(hSize)) return ERR(ZSO_dictionary_corrupted);
      d->huf.valid = 1; dictPtr += hSize; }
    { U32 max = MaxOff, log; size_t const h = readNCount(norm, &max, &log, dictPtr, (size_t)(dictEnd - dictPtr));
      if (zso_isError(h) || max > MaxOff || log > OffFSELog) return ERR(ZSO_dictionary_corrupted);
      buildFSETable(&d->OF, norm, max, NULL, 1, NULL, log); dictPtr += h; }
    { U32 max = MaxML, log; size_t const h = readNCount(norm, &max, &log, dictPtr, (size_t)(dictEnd - dictPtr));
      if (zso_isError(h) || max > MaxML || log > MLFSELog) return ERR(ZSO_dictionary_corrupted);
      buildFSETable(&d->ML, norm, max, ML_base, 0, ML_bits, log); dictPtr += h; }
    { U32 max = MaxLL, log; size_t const h = readNCount(norm, &max, &log, dictPtr, (size_t)(dictEnd - dictPtr));
      if (zso_isError(h) || max > MaxLL || log > LLFSELog) return ERR(ZSO_dictionary_corrupted);
      buildFSETable(&d->LL, norm, max, LL_base, 0, LL_bits, log); dictPtr += h; }
    if (dictPtr + 12 > dictEnd) return ERR(ZSO_dictionary_corrupted);
    { int i; size_t const dictContentSize = (size_t)(dictEnd - (dictPtr + 12));
      for (i = 0; i < 3; i++) { U32 const rep = rdLE32(dictPtr); dictPtr += 4;
          if (rep == 0 || rep >= dictContentSize) return ERR(ZSO_dictionary_corrupted);
          d->rep[i] = rep; } }
    return (size_t)(dictPtr - (const BYTE *)dict);
}

/* ZSTD_decompress_insertDictionary :2452-2475 */
static size_t insertDictionary(DCtx *d, const void *dict, size_t dictSize)
{
    if (dictSize >= 8 && rdLE32(dict) == ZSTD_MAGIC_DICTIONARY) {
        size_t eSize;
        d->dictIDLoaded = rdLE32((const BYTE *)dict + 4);
        eSize = loadEntropy(d, dict, dictSize);
        if (zso_isError(eSize)) return ERR(ZSO_dictionary_corrupted);
        dict = (const BYTE *)dict + eSize; dictSize -= eSize;
        d->litEntropy = d->fseEntropy = 1;
    }
    d->dictContent = (const BYTE *)dict; d->dictContentSize = dictSize;      /* RefDictContent: pure content (mode) or what follows the tables */
    return 0;
}

/* ZSTD_decompress_usingDict :2162-2167 -> DecompressMultiFrame :2096-2160, with a fresh context per call (:2174-2180).
 * dict == NULL / dictSize == 0: the reference's public Decompress */
size_t zso_decompress_usingDict(void *dst, size_t dstCapacity, const void *src, size_t srcSize, const void *dict, size_t dictSize)
{
    DCtx *d = (DCtx *)malloc(sizeof(DCtx));
    BYTE *const dststart = (BYTE *)dst;
    BYTE *op = dststart;
    size_t result;
    if (!d) return ERR(ZSO_memory_allocation);
    d->litBuffer = (BYTE *)malloc(ZSTD_BLOCKSIZE_MAX + 2 * WILDCOPY_OVERLENGTH);
    if (!d->litBuffer) { free(d); return ERR(ZSO_memory_allocation); }
    buildFSETable(&d->LLdef, LL_defaultNorm, MaxLL, LL_base, 0, LL_bits, 6);
    buildFSETable(&d->OFdef, OF_defaultNorm, 28, NULL, 1, NULL, 5);
    buildFSETable(&d->MLdef, ML_defaultNorm, MaxML, ML_base, 0, ML_bits, 6);
    result = 0;
    while (srcSize >= ZSTD_frameHeaderSize_prefix) {
        U32 const magicNumber = rdLE32(src);
        if (magicNumber != ZSTD_MAGICNUMBER) {
            if ((magicNumber & 0xFFFFFFF0u) == ZSTD_MAGIC_SKIPPABLE_START) {
                size_t skippableSize;
                if (srcSize < ZSTD_skippableHeaderSize) { result = ERR(ZSO_srcSize_wrong); goto done; }
                skippableSize = (size_t)rdLE32((const BYTE *)src + 4) + ZSTD_skippableHeaderSize;
                if (srcSize < skippableSize) { result = ERR(ZSO_srcSize_wrong); goto done; }
                STAT(24);
                src = (const BYTE *)src + skippableSize; srcSize -= skippableSize;
                continue;
            }
            result = ERR(ZSO_prefix_unknown); goto done;
        }
        decompressBegin(d);
        if (dict && dictSize) {                                  /* ZSTD_decompressBegin_usingDict :2501-2508 */
            size_t const e = insertDictionary(d, dict, dictSize);
            if (zso_isError(e)) { result = ERR(ZSO_dictionary_corrupted); goto done; }
        }
        STAT(23);
        {
            size_t const res = decompressFrame(d, op, dstCapacity, &src, &srcSize);
            if (zso_isError(res)) { result = res; goto done; }
            op += res; dstCapacity -= res;
        }
    }
    if (srcSize) { result = ERR(ZSO_srcSize_wrong); goto done; }
    result = (size_t)(op - dststart);
done:
    free(d->litBuffer); free(d);
    return result;
}

size_t zso_decompress(void *dst, size_t dstCapacity, const void *src, size_t srcSize)
{
    return zso_decompress_usingDict(dst, dstCapacity, src, srcSize, NULL, 0);
}
