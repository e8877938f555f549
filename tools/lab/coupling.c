/* development aid: how many steps until two FSE encoder state chains that start from different states over the same symbols agree? */
#include "../../oracle/zso_encoder.c"
#include <stdio.h>
static U32 stepState(const CTable *ct, U32 st, U32 sym) { SymTT tt = ct->tt[sym]; U32 nb = (st + tt.deltaNbBits) >> 16; return ct->stateTable[(st >> nb) + tt.deltaFindState]; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); static BYTE buf[1 << 24]; size_t n = fread(buf, 1, sizeof buf, f); fclose(f);
    Work *w = malloc(sizeof(Work)); EParams prm = paramsForLevel(3);
    U32 hist[64] = {0}; double sum = 0; U32 cnt = 0, worst = 0, nover = 0;
    for (size_t pos = 0; pos + 65536 <= n && pos < (8u << 20); pos += 65536) {
        findCandidates(w, buf + pos, 65536, &prm);
        compressBlock(w, w->tmp, 65536 + 512, buf + pos, 65536, 0, 65536, &prm, 1);
        /* recover nseq: count until seqs matchLength 0? use codes arrays: recompute nseq by re-parsing is heavy; use a global */
        extern U32 g_lastNseq; U32 nseq = g_lastNseq;
        for (int t = 0; t < 3; t++) {
            const BYTE *codes = t == 0 ? w->llCode : (t == 1 ? w->ofCode : w->mlCode);
            U32 count[64] = {0}, maxSym = 0, i; S16 norm[64]; CTable ct;
            for (i = 0; i < nseq; i++) { count[codes[i]]++; if (codes[i] > maxSym) maxSym = codes[i]; }
            U32 tableLog = t == 1 ? 8 : 9; { U32 hb = highbit32(nseq - 1); U32 want = hb > 2 ? hb - 2 : 5; if (want < tableLog) tableLog = want; }
            { U32 minBits = highbit32(maxSym) + 2, present = 0; for (i = 0; i <= maxSym; i++) present += count[i] != 0; if (tableLog < minBits) tableLog = minBits; while ((1u << tableLog) < present) tableLog++; }
            if (tableLog < 5) tableLog = 5; if (tableLog > (t == 1 ? 8u : 9u)) tableLog = t == 1 ? 8 : 9;
            normalizeCounts(norm, tableLog, count, nseq, maxSym); buildCTable(&ct, norm, maxSym, tableLog);
            /* true chain from the last sequence */
            static U16 truth[70000]; CState cs; cstate_init(&cs, &ct, codes[nseq - 1], 0); U32 st = cs.value; truth[nseq - 1] = (U16)st;
            for (i = nseq - 1; i-- > 0;) { st = stepState(&ct, st, codes[i]); truth[i] = (U16)st; }
            /* 16 segments: chain of segment j starts at its last sequence with a guessed state (the first-symbol rule) */
            for (U32 j = 0; j < 15; j++) {
                U32 e = (j + 1) * nseq / 16;            /* sequences [.., e) ; start encoding at e-1 with a guess */
                if (e < 2) continue;
                cstate_init(&cs, &ct, codes[e - 1], 0); U32 g = cs.value; U32 k = e - 1, steps = 0;
                while (g != truth[k] && k > 0) { k--; g = stepState(&ct, g, codes[k]); steps++; }
                sum += steps; cnt++; if (steps > worst) worst = steps; hist[steps > 63 ? 63 : steps]++; if (steps > 200) nover++;
            }
        }
    }
    printf("segments %u: mean coupling %.1f steps, worst %u, >200: %u\n", cnt, sum / cnt, worst, nover);
    for (int i = 0; i < 64; i += 8) { for (int k = 0; k < 8; k++) printf("%6u", hist[i + k]); printf("\n"); }
    return 0;
}
