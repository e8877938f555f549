/* Walk laboratory (development aid, not product, not oracle): includes the oracle encoder's source to reuse its
 * candidate and entropy stages, and swaps the walk + stitch for a parametrised variant, so that ratio and work
 * (steps, candidates scored) can be measured on the mixed corpus before a change is frozen into oracle E + HIP. */
#include <stdio.h>
#include "../../oracle/zso_encoder.c"

typedef struct {
    int walkLog;      /* log2 of the walk range */
    int crossMax;     /* a match may pass its range end by this much */
    int look;         /* 0: the level's; else candidates scored per step */
    int merge;        /* 1: the stitch joins a range's first kept match to the reach-defining match when it starts at the reach with the same offset */
    int window;       /* positions looked at per step */
    int repwin;       /* positions tried for recent offsets */
    int longEven;     /* 1: the long table takes even positions only (insert and look-up) */
    int carryRep;     /* 1 (not parallel: what the resets cost): a range starts with the previous range's recent offsets */
    int skipFirst;    /* 1: first candidate wins if its gain >= this (0 = off) */
    int lazyMax;      /* reserved */
    int approx;       /* K > 0: the candidates of a step are ranked by an estimate (no compares), the best K are measured */
    int estLong, estShort, estRep;   /* estimated forward lengths */
    int estSkip;      /* estimated cost per skipped position */
    int useBack;      /* 0: no backward extension */
    int estOff;       /* 1: the estimate counts the offset's cost */
    int initRep;      /* 1: a range starts with rep0 = the distance of the nearest candidate position below its start */
    int bcap, fcap;   /* backward cap, forward score cap */
    int repMode;      /* 0: recent offsets tried on the first repwin positions; 1: only where stage 1 left a candidate; 2: no probe: a candidate whose distance IS a recent offset scores as one */
    int walign;       /* 1: the window ends at an aligned group of 8: (ip & ~7) + window */
    int estRun;       /* 1: the estimate adds the run of candidate positions that follows (cap FCAP) */
} WP;
static WP P = { 10, 16384, 0, 0, 64, 8, 0, 0, 0, 0, 0, 8, 5, 4, 5, 1, 1, 0, 8, 8, 0, 0, 0 };
static BYTE wlLong[UNIT_MAX];
static struct { unsigned long long steps, emptySteps, scored, emitted, kept, extBytes, ranges, merged, waveMax32, waveMax16, waveMax64; } S;
static U32 wlSteps[1024];
void wl_set(const char *k, int v)
{
#define K(name) if (!strcmp(k, #name)) { P.name = v; return; }
    K(walkLog) K(crossMax) K(look) K(merge) K(window) K(repwin) K(longEven) K(carryRep) K(skipFirst) K(lazyMax) K(approx) K(estLong) K(estShort) K(estRep) K(estSkip) K(useBack) K(estOff) K(initRep) K(bcap) K(fcap) K(repMode) K(walign) K(estRun)
#undef K
    fprintf(stderr, "unknown %s\n", k); abort();
}
void wl_stats(unsigned long long *o) { memcpy(o, &S, sizeof S); memset(&S, 0, sizeof S); }

static void wlCandidates(Work *w, const BYTE *src, U32 n, const EParams *prm)
{
    U32 const tlog = tableLogFor(n);
    U32 p, found = 0;
    memset(wlLong, 0, n);
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
        if (prm->useLong && (!P.longEven || !(p & 1))) {
            U32 const hl = hashLong(src + p);
            U32 const el = (((hl >> (32 - tlog - 15)) & 0x7FFFu) << 17) | p;
            U32 const ol = w->tabL[hl >> (32 - tlog)];
            w->tabL[hl >> (32 - tlog)] = el;
            if (ol != SLOT_EMPTY && ((ol ^ el) >> 17) == 0) { d = p - (ol & 0x1FFFFu); wlLong[p] = 1; }
        }
        w->dist[p] = d;
        found += d != 0;
    }
    w->matchless = found < (n >> MATCHLESS_SHIFT);
}

static U32 wlLastSteps;
static U32 wlWalk(Work *w, const BYTE *src, U32 n, U32 start, U32 end, U32 limit, const EParams *prm, ASeq *out, U32 *reps)
{
    U32 ip = start, anchor = start, nseq = 0, rep0 = reps[0], rep1 = reps[1];
    U32 const hashable = (n >= 8) ? n - 7 : 0;
    U32 const look = P.look ? (U32)P.look : (U32)prm->look;
    U32 const scanEnd = (end < hashable) ? end : hashable;
    U32 mySteps = 0;
    while (ip < scanEnd) {
        int bestGain = 0, have = 0; U32 bestQ = 0, bestFwd = 0, bestBack = 0, bestOff = 0, q, seen = 0;
        U32 const wbase = P.walign ? (ip & ~7u) : ip;
        U32 const wend = (wbase + (U32)P.window < scanEnd) ? wbase + (U32)P.window : scanEnd;
        S.steps++; mySteps++;
        if (P.approx) {
            U32 cq[16], coff[16]; int crep[16], cest[16], nc = 0, k, tries;
            for (q = ip; q < wend && (U32)nc < look; q++) {
                U32 off = 0; int isRep = 0;
                if (q < ip + (U32)P.repwin && q + 4 <= limit) {
                    if (rep0 && ip >= rep0 && rd32(src + q) == rd32(src + q - rep0)) { off = rep0; isRep = 1; }
                    else if (rep1 && ip >= rep1 && rd32(src + q) == rd32(src + q - rep1)) { off = rep1; isRep = 1; }
                }
                if (!off) off = w->dist[q];
                if (!off) continue;
                cq[nc] = q; coff[nc] = off; crep[nc] = isRep;
                { int el = isRep ? P.estRep : (wlLong[q] ? P.estLong : P.estShort);
                  if (P.estRun == 1) { U32 r = 1; while (q + r < wend && r < 8 && w->dist[q + r]) r++; el += (int)r - 1; if (el > (int)FCAP) el = FCAP; }
                  cest[nc] = 4 * el; }
                if (P.estRun >= 2 && q > ip && !w->dist[q - 1]) cest[nc] += P.estRun;   /* run start */
                cest[nc] = cest[nc] - ((isRep || !P.estOff) ? 0 : (int)highbit32(off + 1)) - P.estSkip * (int)(q - ip);
                nc++;
            }
            for (tries = 0; tries < P.approx && nc; tries++) {
                int b = 0; U32 off, fwd, back = 0; int gain;
                for (k = 1; k < nc; k++) if (cest[k] > cest[b]) b = k;
                q = cq[b]; off = coff[b]; S.scored++;
                fwd = matchLen(src, q, q - off, limit);
                if (fwd >= (crep[b] ? REPMIN : MINMATCH)) {
                    if (P.useBack) while (back < (U32)P.bcap && q - back > anchor && q - off - back > 0 && src[q - back - 1] == src[q - off - back - 1]) back++;
                    gain = (int)((fwd > (U32)P.fcap ? (U32)P.fcap : fwd) + back) * 4 - (crep[b] ? 0 : (int)highbit32(off + 1)) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
                    if (!have || gain > bestGain) { have = 1; bestGain = gain; bestQ = q; bestFwd = fwd; bestBack = back; bestOff = off; }
                }
                cq[b] = cq[nc - 1]; coff[b] = coff[nc - 1]; crep[b] = crep[nc - 1]; cest[b] = cest[nc - 1]; nc--;
                if (have && P.skipFirst) break;
            }
        } else
        for (q = ip; q < wend && seen < look; q++) {
            U32 off = 0, fwd, back = 0; int isRep = 0, gain;
            if (P.repMode == 0 || (P.repMode == 1 && w->dist[q])) if (q < ip + (U32)P.repwin && q + 4 <= limit) {
                if (rep0 && ip >= rep0 && rd32(src + q) == rd32(src + q - rep0)) { off = rep0; isRep = 1; }
                else if (rep1 && ip >= rep1 && rd32(src + q) == rd32(src + q - rep1)) { off = rep1; isRep = 1; }
            }
            if (!off) off = w->dist[q];
            if (!off) continue;
            if (P.repMode == 2 && (off == rep0 || off == rep1)) isRep = 1;
            seen++; S.scored++;
            fwd = matchLen(src, q, q - off, limit);
            if (fwd < (isRep ? REPMIN : MINMATCH)) continue;
            if (P.useBack) while (back < (U32)P.bcap && q - back > anchor && q - off - back > 0 && src[q - back - 1] == src[q - off - back - 1]) back++;
            gain = (int)((fwd > (U32)P.fcap ? (U32)P.fcap : fwd) + back) * 4 - (isRep ? 0 : (int)highbit32(off + 1)) - 4 * ((int)(q - back) - (int)ip) - (int)(q - ip);
            if (!have || gain > bestGain) { have = 1; bestGain = gain; bestQ = q; bestFwd = fwd; bestBack = back; bestOff = off; }
            if (P.skipFirst && have && bestGain >= P.skipFirst) break;
        }
        if (!have) { ip = wend; S.emptySteps++; continue; }
        out[nseq].start = bestQ - bestBack; out[nseq].ml = bestBack + bestFwd; out[nseq].off = bestOff; nseq++;
        if (bestFwd > 16) S.extBytes += bestFwd - 16;
        ip = bestQ + bestFwd; anchor = ip;
        if (bestOff == rep1) { rep1 = rep0; rep0 = bestOff; }
        else if (bestOff != rep0) { rep1 = rep0; rep0 = bestOff; }
    }
    reps[0] = rep0; reps[1] = rep1;
    S.emitted += nseq; S.ranges++;
    wlLastSteps = mySteps;
    return nseq;
}

static ASeq wlSeq[1024][512];
static U32 wlN[1024];
static size_t wlBlock(Work *w, BYTE *dst, size_t cap, const BYTE *src, U32 unitN, U32 blockOff, U32 n, const EParams *prm, int firstBlock)
{
    U32 nseq = 0, nlit = 0;
    U32 const WS = 1u << P.walkLog;
    U32 const nRanges = (n + WS - 1) >> P.walkLog;
    U32 const blockEnd = blockOff + n;
    U32 r, reach = blockOff;
    U32 reps[2] = { 0, 0 };
    if (n < 16) return 0;
    for (r = 0; r < nRanges; r++) {
        U32 const start = blockOff + (r << P.walkLog);
        U32 const end = (start + WS < blockEnd) ? start + WS : blockEnd;
        U32 const limit = (end + (U32)P.crossMax < blockEnd) ? end + (U32)P.crossMax : blockEnd;
        if (!P.carryRep) reps[0] = reps[1] = 0;
        if (P.initRep && start > blockOff) { U32 b; for (b = 1; b <= 64 && start >= b + (P.initRep == 2 ? 0 : blockOff); b++) if (w->dist[start - b]) { reps[0] = w->dist[start - b]; break; } }
        wlLastSteps = 0;
        wlN[r] = w->matchless ? 0 : wlWalk(w, src, unitN, start, end, limit, prm, wlSeq[r], reps);
        wlSteps[r] = wlLastSteps;
    }
    {   /* wave imbalance: consecutive ranges side by side in a wavefront, which runs as long as its slowest walker */
        U32 g, k;
        for (g = 0; g < nRanges; g += 16) { U32 m = 0; for (k = g; k < g + 16 && k < nRanges; k++) if (wlSteps[k] > m) m = wlSteps[k]; S.waveMax16 += m; }
        for (g = 0; g < nRanges; g += 32) { U32 m = 0; for (k = g; k < g + 32 && k < nRanges; k++) if (wlSteps[k] > m) m = wlSteps[k]; S.waveMax32 += m; }
        for (g = 0; g < nRanges; g += 64) { U32 m = 0; for (k = g; k < g + 64 && k < nRanges; k++) if (wlSteps[k] > m) m = wlSteps[k]; S.waveMax64 += m; }
    }
    {   /* the stitch, sequential statement (same result as oracle's stitch + concatenation when merge == 0) */
        U32 pos = blockOff;                 /* == reach throughout */
        for (r = 0; r < nRanges; r++) {
            U32 const own = reach;
            U32 const ns = wlN[r];
            U32 f = 0, k;
            U32 const le = ns ? wlSeq[r][ns - 1].start + wlSeq[r][ns - 1].ml : 0;
            while (f < ns) {
                ASeq *s = &wlSeq[r][f];
                if (s->start + s->ml <= own) { f++; continue; }
                if (s->start < own) {
                    U32 const cut = own - s->start;
                    if (P.merge == 1 && nseq && s->off == w->seqs[nseq - 1].offset) { s->start += cut; s->ml -= cut; break; }   /* joins the match it straddles into */
                    if (s->ml - cut < MINMATCH) { f++; continue; }
                    s->start += cut; s->ml -= cut;
                }
                break;
            }
            for (k = f; k < ns; k++) {
                ASeq const s = wlSeq[r][k];
                if (P.merge && k == f && nseq && s.start == pos && s.off == w->seqs[nseq - 1].offset) { w->seqs[nseq - 1].matchLength += s.ml; pos = s.start + s.ml; S.merged++; continue; }
                w->seqs[nseq].litLength = s.start - pos; w->seqs[nseq].matchLength = s.ml; w->seqs[nseq].offset = s.off; nseq++;
                memcpy(w->lits + nlit, src + pos, s.start - pos); nlit += s.start - pos;
                pos = s.start + s.ml;
            }
            if (le > reach) reach = le;
        }
        memcpy(w->lits + nlit, src + pos, blockEnd - pos); nlit += blockEnd - pos;
    }
    S.kept += nseq;
    return encodeParsed(w, dst, cap, nseq, nlit, firstBlock);
}

size_t wl_compress(void *dstv, size_t dstCapacity, const void *srcv, size_t srcSize, int level)
{
    BYTE *const dst = (BYTE *)dstv; const BYTE *const src = (const BYTE *)srcv;
    BYTE *op = dst;
    EParams const prm = paramsForLevel(level);
    static __thread Work *w; size_t pos = 0;
    (void)dstCapacity;
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
        if (pos == unitPos && n) wlCandidates(w, src + unitPos, unitN, &prm);
        for (i = 1; i < n && same; i++) same = src[pos + i] == src[pos];
        if (same && n > 0) { wr24(op, (U32)last + (1u << 1) + (n << 3)); op[3] = src[pos]; op += 4; }
        else {
            if (n) csize = wlBlock(w, w->tmp, n + 512, src + unitPos, unitN, (U32)(pos - unitPos), n, &prm, pos == 0);
            if (csize && csize < n) { wr24(op, (U32)last + (2u << 1) + ((U32)csize << 3)); memcpy(op + 3, w->tmp, csize); op += 3 + csize; }
            else { wr24(op, (U32)last + (0u << 1) + (n << 3)); memcpy(op + 3, src + pos, n); op += 3 + n; }
        }
        pos += n;
    } while (pos < srcSize);
    return (size_t)(op - dst);
}
