/* Match-finder laboratory (development aid, not product, not oracle): includes the oracle encoder's
 * source to reuse its entropy stages, and swaps the LZ stage for parametrised variants so that ratio
 * effects can be measured on the mixed corpus before an algorithm is frozen into oracle E + HIP. */
#include "../../oracle/zso_encoder.c"

typedef struct {
    int shortLen;    /* bytes hashed for the short candidate (4..6) */
    int longLen;     /* bytes hashed for the long candidate (0 = none, 6..8) */
    int walkLog;     /* log2 of walk-range size */
    int look;        /* candidates scored per step */
    int fcap, bcap;  /* forward score cap, backward extension cap */
    int rep;         /* 1: rep-offset candidates */
    int cross;       /* 1: matches may pass the walk range end (up to the block end), stitched afterwards */
    int minmatch;
    int ideal;       /* 1: collision-free most-recent k-gram candidates; 0: ranged tagged tables of the oracle (short only) */
    int window;      /* positions looked at per step */
    int stepModel;   /* 1: candidates inside the same 64-position step are invisible (GPU insertion model) */
    int repcost;     /* offset cost of a rep candidate */
    int longMin;     /* verified length required of a long candidate */
    int crossCap;    /* max bytes past range end (0 = unlimited) */
    int perPos;      /* 1: one candidate per position (priority rep > long > short) ; 0: all */
    int skipMul;     /* literal skip penalty multiplier (x4 default) */
    int nreps;       /* rep offsets tracked by a walker (1..3) */
    int repWin;      /* positions of the window checked for rep matches */
    int sLog, lLog;  /* table model (ideal=2): unit-wide tables of 2^sLog / 2^lLog slots, last writer wins */
    int tagBits;     /* table model: tag bits compared (rest of the hash); a tag mismatch = no candidate */
    int verify;      /* table model: 1 = candidate bytes verified (shortLen / longMin bytes) */
    int repMin;      /* minimum length of a rep match */
} LabP;
static LabP P = { 4, 0, 10, 8, 8, 8, 0, 0, 5, 0, 64, 1, 0, 8, 0, 0, 4, 1, 64, 13, 13, 16, 1, 4 };
void lab_set(const char *k, int v)
{
#define K(name) if (!strcmp(k, #name)) { P.name = v; return; }
    K(shortLen) K(longLen) K(walkLog) K(look) K(fcap) K(bcap) K(rep) K(cross) K(minmatch) K(ideal) K(window) K(stepModel) K(repcost) K(longMin) K(crossCap) K(perPos) K(skipMul) K(nreps) K(repWin) K(sLog) K(lLog) K(tagBits) K(verify) K(repMin)
#undef K
}

static U64 rd64(const BYTE *p) { U64 v; memcpy(&v, p, 8); return v; }
static U32 hashK(const BYTE *p, int k, int bits)
{
    U64 v = rd64(p);
    if (k < 8) v <<= (64 - 8 * k);
    return (U32)((v * 0x9E3779B185EBCA87ULL) >> (64 - bits));
}

#define LBITS 20
static U32 *distS, *distL;   /* per unit position */
static U32 *tabS, *tabL;

static void labCandidates(Work *w, const BYTE *src, U32 n, const EParams *prm)
{
    U32 p, base;
    if (!distS) { distS = malloc(4 * UNIT_MAX); distL = malloc(4 * UNIT_MAX); tabS = malloc(4u << LBITS); tabL = malloc(4u << LBITS); }
    memset(distS, 0, 4 * n); memset(distL, 0, 4 * n);
    if (!P.ideal) { findCandidates(w, src, n, prm); memcpy(distS, w->dist, 4 * n); }
    if (n < 16) return;
    memset(tabS, 0xFF, 4u << LBITS); memset(tabL, 0xFF, 4u << LBITS);
    for (base = 0; base + 8 <= n; base += 64) {
        U32 stop = base + 64; if (stop + 8 > n) stop = n - 8;
        U32 step = P.stepModel ? 64 : 1, b2;
        for (b2 = base; b2 < stop; b2 += step) {
            U32 s2 = P.stepModel ? stop : b2 + 1;
            for (p = b2; p < s2; p++) {
                if (P.ideal == 1) {
                    U32 c = tabS[hashK(src + p, P.shortLen, LBITS)];
                    if (c != 0xFFFFFFFFu && matchLen(src, p, c, p + P.shortLen > n ? n : p + P.shortLen) >= (U32)P.shortLen) distS[p] = p - c;
                    if (P.longLen) {
                        c = tabL[hashK(src + p, P.longLen, LBITS)];
                        if (c != 0xFFFFFFFFu && matchLen(src, p, c, p + P.longMin > n ? n : p + P.longMin) >= (U32)P.longMin) distL[p] = p - c;
                    }
                } else if (P.ideal == 2) {
                    U32 hh = hashK(src + p, P.shortLen, 32), e = tabS[hh >> (32 - P.sLog)];
                    U32 tag = (hh >> (32 - P.sLog - P.tagBits)) & ((1u << P.tagBits) - 1);
                    if (e != 0xFFFFFFFFu && (e >> 17) == tag) { U32 c = e & 0x1FFFF; if (!P.verify || matchLen(src, p, c, p + P.shortLen > n ? n : p + P.shortLen) >= (U32)P.shortLen) distS[p] = p - c; }
                    if (P.longLen) {
                        hh = hashK(src + p, P.longLen, 32); e = tabL[hh >> (32 - P.lLog)];
                        tag = (hh >> (32 - P.lLog - P.tagBits)) & ((1u << P.tagBits) - 1);
                        if (e != 0xFFFFFFFFu && (e >> 17) == tag) { U32 c = e & 0x1FFFF; if (!P.verify || matchLen(src, p, c, p + P.longMin > n ? n : p + P.longMin) >= (U32)P.longMin) distL[p] = p - c; }
                    }
                } else if (P.longLen) {
                    U32 c = tabL[hashK(src + p, P.longLen, LBITS)];
                    if (c != 0xFFFFFFFFu && matchLen(src, p, c, p + P.longMin > n ? n : p + P.longMin) >= (U32)P.longMin) distL[p] = p - c;
                }
            }
            for (p = b2; p < s2; p++) {
                if (P.ideal == 2) {
                    U32 hh = hashK(src + p, P.shortLen, 32);
                    tabS[hh >> (32 - P.sLog)] = (((hh >> (32 - P.sLog - P.tagBits)) & ((1u << P.tagBits) - 1)) << 17) | p;
                    if (P.longLen) { hh = hashK(src + p, P.longLen, 32); tabL[hh >> (32 - P.lLog)] = (((hh >> (32 - P.lLog - P.tagBits)) & ((1u << P.tagBits) - 1)) << 17) | p; }
                } else {
                    tabS[hashK(src + p, P.shortLen, LBITS)] = p;
                    if (P.longLen) tabL[hashK(src + p, P.longLen, LBITS)] = p;
                }
            }
        }
    }
}


/* one walk range [start, end); matches may extend to limit (= end without cross, else block end) */
static U32 labWalk(const BYTE *src, U32 n, U32 start, U32 end, U32 limit, ASeq *out, U32 *lastAnchor)
{
    U32 ip = start, anchor = start, nseq = 0, rep[3] = { 0, 0, 0 };
    U32 const lastStart = (n >= 8) ? n - 8 : 0;
    U32 const scanEnd = (end < lastStart + 1) ? end : lastStart + 1;
    while (ip < scanEnd) {
        int bestGain = 0, have = 0; U32 bestQ = 0, bestFwd = 0, bestBack = 0, bestOff = 0, q, seen = 0;
        U32 const wend = (ip + (U32)P.window < scanEnd) ? ip + (U32)P.window : scanEnd;
        for (q = ip; q < wend && seen < (U32)P.look; q++) {
            U32 offs[5]; int isrep[5]; int nc = 0, c, k, j;
            if (P.rep && q < ip + (U32)P.repWin) for (k = 0; k < P.nreps; k++) if (rep[k] && q >= rep[k] && rd32(src + q) == rd32(src + q - rep[k])) { offs[nc] = rep[k]; isrep[nc++] = 1; }
            if (distL[q]) { int dup = 0; for (j = 0; j < nc; j++) dup |= offs[j] == distL[q]; if (!dup) { offs[nc] = distL[q]; isrep[nc++] = 0; } }
            if (distS[q]) { int dup = 0; for (j = 0; j < nc; j++) dup |= offs[j] == distS[q]; if (!dup) { offs[nc] = distS[q]; isrep[nc++] = 0; } }
            if (!nc) continue;
            if (P.perPos) nc = 1;
            for (c = 0; c < nc && seen < (U32)P.look; c++) {
                U32 const off = offs[c];
                U32 fwd, back = 0, cap;
                int gain;
                seen++;
                cap = limit - q;
                fwd = matchLen(src, q, q - off, q + cap);
                if (fwd < (U32)(isrep[c] ? P.repMin : P.minmatch)) continue;
                while (back < (U32)P.bcap && q - back > anchor && q - off - back > 0 && src[q - back - 1] == src[q - off - back - 1]) back++;
                gain = (int)((fwd > (U32)P.fcap ? (U32)P.fcap : fwd) + back) * 4 - (isrep[c] ? P.repcost : (int)highbit32(off + 1)) - P.skipMul * ((int)(q - back) - (int)ip) - (int)(q - ip);
                if (!have || gain > bestGain) { have = 1; bestGain = gain; bestQ = q; bestFwd = fwd; bestBack = back; bestOff = off; }
            }
        }
        if (!have) { ip = wend; continue; }
        out[nseq].start = bestQ - bestBack; out[nseq].ml = bestBack + bestFwd; out[nseq].off = bestOff; nseq++;
        ip = bestQ + bestFwd; anchor = ip;
        if (bestOff == rep[0]) {}
        else if (bestOff == rep[1]) { rep[1] = rep[0]; rep[0] = bestOff; }
        else { rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = bestOff; }
    }
    *lastAnchor = anchor;
    return nseq;
}

static size_t labBlock(Work *w, BYTE *dst, size_t cap, const BYTE *src, U32 unitN, U32 blockOff, U32 n, int firstBlock)
{
    U32 nseq = 0, nlit = 0;
    U32 const WS = 1u << P.walkLog;
    U32 const nRanges = (n + WS - 1) >> P.walkLog;
    U32 const blockEnd = blockOff + n;
    U32 r, reach = blockOff;     /* end of the last emitted match */
    static ASeq rs[8192];
    if (n < 16) return 0;
    for (r = 0; r < nRanges; r++) {
        U32 const start = blockOff + (r << P.walkLog);
        U32 const end = (start + WS < blockEnd) ? start + WS : blockEnd;
        U32 limit = P.cross ? blockEnd : end, la, k, ns;
        if (P.cross && P.crossCap && end + (U32)P.crossCap < limit) limit = end + (U32)P.crossCap;
        ns = labWalk(src, unitN, start, end, limit, rs, &la);
        for (k = 0; k < ns; k++) {
            ASeq s = rs[k];
            if (s.start + s.ml <= reach) continue;                 /* covered by an earlier range's match */
            if (s.start < reach) { U32 cut = reach - s.start; s.start += cut; s.ml -= cut; if (s.ml < 4) continue; }
            w->seqs[nseq].litLength = s.start - reach; w->seqs[nseq].matchLength = s.ml; w->seqs[nseq].offset = s.off; nseq++;
            memcpy(w->lits + nlit, src + reach, s.start - reach); nlit += s.start - reach;
            reach = s.start + s.ml;
        }
    }
    memcpy(w->lits + nlit, src + reach, blockEnd - reach); nlit += blockEnd - reach;
    return encodeParsed(w, dst, cap, nseq, nlit, firstBlock);
}

size_t lab_compress(void *dstv, size_t dstCapacity, const void *srcv, size_t srcSize, int level)
{
    BYTE *const dst = (BYTE *)dstv; const BYTE *const src = (const BYTE *)srcv;
    BYTE *op = dst;
    EParams const prm = paramsForLevel(level);
    static Work *w; size_t pos = 0;
    if (!w) w = malloc(sizeof(Work));
    wr32(op, 0xFD2FB528u); op += 4;
    if (srcSize < 256) { *op++ = 0x20; *op++ = (BYTE)srcSize; }
    else if (srcSize < 65536 + 256) { *op++ = 0x60; wr16(op, (U32)srcSize - 256); op += 2; }
    else { *op++ = 0xA0; wr32(op, (U32)srcSize); op += 4; }
    do {
        U32 const n = (U32)((srcSize - pos < BLOCK_MAX) ? srcSize - pos : BLOCK_MAX);
        int const last = (pos + n == srcSize);
        size_t const unitPos = pos & ~(size_t)(UNIT_MAX - 1);
        U32 const unitN = (U32)((srcSize - unitPos < UNIT_MAX) ? srcSize - unitPos : UNIT_MAX);
        size_t csize = 0; U32 i, same = n > 0;
        if (pos == unitPos && n) labCandidates(w, src + unitPos, unitN, &prm);
        for (i = 1; i < n && same; i++) same = src[pos + i] == src[pos];
        if (same && n > 0) { wr24(op, (U32)last + (1u << 1) + (n << 3)); op[3] = src[pos]; op += 4; }
        else {
            if (n) csize = labBlock(w, w->tmp, n + 512, src + unitPos, unitN, (U32)(pos - unitPos), n, pos == 0);
            if (csize && csize < n) { wr24(op, (U32)last + (2u << 1) + ((U32)csize << 3)); memcpy(op + 3, w->tmp, csize); op += 3 + csize; }
            else { wr24(op, (U32)last + (0u << 1) + (n << 3)); memcpy(op + 3, src + pos, n); op += 3 + n; }
        }
        pos += n;
    } while (pos < srcSize);
    return (size_t)(op - dst);
}
