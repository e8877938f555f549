/* A plain-C caller of include/zsmi.h, the way a P/Invoke or JNI shim reaches it: one-shot host calls with the reference's
 * conventions (ZStdDecompress.cs:2182-2191: sizes in, size or (size_t)-code out, nothing retained) and the batch calls.
 * Built with gcc against libzsmi.so by tests/test_c_abi.py; exits 0 when every check holds. */
#include "zsmi.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

/* ---- the one-shot calls from several threads at once (the reference's static API is re-entrant: ZStdDecompress.cs:2174-2191) ---- */
typedef struct {
    int id; const unsigned char *src; size_t n; const unsigned char *frame; size_t csize;      /* shared, read only */
    const unsigned char *golden; size_t goldenSize; const unsigned char *goldenWant; size_t goldenWantSize;
    int failed;
} Job;
static void *worker(void *arg)
{
    Job *j = (Job *)arg;
    size_t const piece = 20000 + 1000 * (size_t)j->id, off = 30000 * (size_t)j->id;
    unsigned char *buf = (unsigned char *)malloc(zsmi_compressBound(j->n)), *back = (unsigned char *)malloc(j->n);
    int it;
    j->failed = 1;
    if (!buf || !back) return NULL;
    for (it = 0; it < 6; it++) {
        size_t r;
        /* my own slice: compress, decompress, compare */
        size_t const c = zsmi_compress(buf, zsmi_compressBound(piece), j->src + off, piece, 1 + (j->id + it) % 4);
        if (zsmi_isError(c)) return NULL;
        r = zsmi_decompress(back, piece, buf, c);
        if (r != piece || memcmp(back, j->src + off, piece) != 0) return NULL;
        /* the shared frame, made on one thread */
        r = zsmi_decompress(back, j->n, j->frame, j->csize);
        if (r != j->n || memcmp(back, j->src, j->n) != 0) return NULL;
        /* the reference's golden vector, when the caller gave it */
        if (j->golden) {
            r = zsmi_decompress(back, j->goldenWantSize, j->golden, j->goldenSize);
            if (r != j->goldenWantSize || memcmp(back, j->goldenWant, r) != 0) return NULL;
        }
        /* an error on one thread is that call's result alone */
        r = zsmi_decompress(back, 10, j->frame, j->csize);
        if (!zsmi_isError(r) || zsmi_getErrorCode(r) != ZSMI_error_dstSize_tooSmall) return NULL;
    }
    free(buf); free(back);
    j->failed = 0;
    return NULL;
}
static unsigned char *slurp(const char *path, size_t *size)
{
    FILE *f = fopen(path, "rb"); unsigned char *p; long n;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    p = (unsigned char *)malloc((size_t)n + 1);
    if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f); *size = (size_t)n;
    return p;
}
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int main(int argc, char **argv)
{
    /* text-like input: words from a small vocabulary */
    size_t n = 300000, i = 0;
    unsigned char *src = (unsigned char *)malloc(n), *back = (unsigned char *)malloc(n);
    static const char *words[] = { "frame", "block", "literal", "sequence", "offset", "match", "table", "state", "stream", "window" };
    while (i < n) { const char *w = words[rnd() % 10]; size_t l = strlen(w); if (i + l + 1 > n) l = n - i - 1; memcpy(src + i, w, l); i += l; src[i++] = (rnd() % 7) ? ' ' : '\n'; }

    /* one-shot compress / decompress (host buffers) */
    size_t cap = zsmi_compressBound(n);
    unsigned char *frame = (unsigned char *)malloc(cap);
    size_t csize = zsmi_compress(frame, cap, src, n, 3);
    CHECK(!zsmi_isError(csize));
    CHECK(csize < n / 2);
    CHECK(zsmi_getDecompressedSize(frame, csize) == n);
    size_t dsize = zsmi_decompress(back, n, frame, csize);
    CHECK(dsize == n && memcmp(back, src, n) == 0);

    /* error ABI: same codes as ZStdErrors.cs:61-90 */
    size_t r = zsmi_decompress(back, 10, frame, csize);
    CHECK(zsmi_isError(r) && zsmi_getErrorCode(r) == ZSMI_error_dstSize_tooSmall);
    r = zsmi_decompress(back, n, frame, csize - 1);
    CHECK(zsmi_isError(r));
    r = zsmi_decompress(back, n, "garbage!garbage!", 16);
    CHECK(zsmi_isError(r) && zsmi_getErrorCode(r) == ZSMI_error_prefix_unknown);
    CHECK(strlen(zsmi_getErrorName(r)) > 0);
    CHECK(zsmi_compress(frame, 4, src, n, 3) > (size_t)-ZSMI_error_maxCode);

    /* batch calls on host buffers: 4 ragged chunks <-> 4 frames */
    zsmi_ctx *ctx = zsmi_createCtx(-1, NULL);
    CHECK(ctx != NULL);
    uint64_t so[4] = { 0, 70000, 70001, 200000 }; uint32_t ss[4] = { 70000, 1, 129999, 100000 };
    uint64_t dofs[4]; uint32_t dsz[4]; size_t tot = 0;
    for (i = 0; i < 4; i++) { dofs[i] = tot; tot += zsmi_compressBound(ss[i]); }
    unsigned char *arena = (unsigned char *)malloc(tot);
    CHECK(zsmi_compressBatchHost(ctx, src, so, ss, 4, arena, dofs, dsz, 3) == 0);
    for (i = 0; i < 4; i++) CHECK(dsz[i] < 0xFFFFFF88u);
    uint32_t osz[4]; uint32_t caps[4] = { 70000, 1, 129999, 100000 };
    memset(back, 0, n);
    CHECK(zsmi_decompressBatchHost(ctx, arena, dofs, dsz, 4, back, so, caps, osz) == 0);
    for (i = 0; i < 4; i++) CHECK(osz[i] == ss[i]);
    CHECK(memcmp(back, src, 70001) == 0 && memcmp(back + 70001, src + 70001, 129999) == 0 && memcmp(back + 200000, src + 200000, 100000) == 0);
    caps[2] = 1000;                                            /* one chunk's room too small: its status carries the code */
    CHECK(zsmi_decompressBatchHost(ctx, arena, dofs, dsz, 4, back, so, caps, osz) == 0);
    CHECK(osz[2] == (uint32_t)-ZSMI_error_dstSize_tooSmall && osz[0] == 70000 && osz[3] == 100000);
    zsmi_freeCtx(ctx);

    /* 8 threads, mixed one-shot calls; argv[1] / argv[2]: the reference's golden frame and its content (tests/golden/csharp_alphabet.*) */
    {
        enum { NTHREADS = 8 };
        pthread_t th[NTHREADS]; Job jobs[NTHREADS];
        size_t gsize = 0, wsize = 0; int t;
        unsigned char *golden = argc > 2 ? slurp(argv[1], &gsize) : NULL, *want = argc > 2 ? slurp(argv[2], &wsize) : NULL;
        CHECK(argc <= 2 || (golden && want));
        for (t = 0; t < NTHREADS; t++) {
            Job jb = { t, src, n, frame, csize, golden, gsize, want, wsize, 1 };
            jobs[t] = jb;
            CHECK(pthread_create(&th[t], NULL, worker, &jobs[t]) == 0);
        }
        for (t = 0; t < NTHREADS; t++) pthread_join(th[t], NULL);
        for (t = 0; t < NTHREADS; t++) CHECK(!jobs[t].failed);
        if (golden) {                                          /* latency of one small one-shot call (a 484-byte frame), warm */
            double t0, best = 1e9; int k;
            for (k = 0; k < 20; k++) { t0 = now_ms(); CHECK(zsmi_decompress(back, wsize, golden, gsize) == wsize); if (now_ms() - t0 < best) best = now_ms() - t0; }
            printf("one-shot zsmi_decompress of the %zu-byte golden frame: %.3f ms (best of 20, host buffers, everything included)\n", gsize, best);
        }
    }
    printf("abi_consumer ok: %zu -> %zu bytes, 8 threads of mixed one-shot calls, %s\n", n, csize, zsmi_versionString());
    return 0;
}
