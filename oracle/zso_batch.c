/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Multi-threaded drivers over the scalar oracle
 * functions: n independent chunks, one frame each, static partition over nThreads.
 * Used by bench.py's cpu_baseline leg (timed on host cores) and by tests.
 */
#include "zso_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <dlfcn.h>

/* upstream libzstd through dlopen, if the box has it: an independent implementation used as a labelled yardstick, never as the
 * reference (which has no encoder and no runtime here) */
typedef size_t (*zstd_compress_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
static zstd_compress_fn z_compress; static zstd_decompress_fn z_decompress; static zstd_iserror_fn z_iserror;
static int loadLibzstd(void)
{
    static int state;          /* 0 untried, 1 loaded, -1 absent */
    if (!state) {
        void *h = dlopen("libzstd.so.1", RTLD_NOW);
        state = -1;
        if (h) {
            z_compress = (zstd_compress_fn)dlsym(h, "ZSTD_compress"); z_decompress = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
            z_iserror = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
            if (z_compress && z_decompress && z_iserror) state = 1;
        }
    }
    return state == 1;
}

typedef struct {
    int compress, level, useLibzstd;
    uint8_t *dst; const uint64_t *dstOffsets; const uint32_t *dstCaps; uint32_t *dstSizes;
    const uint8_t *src; const uint64_t *srcOffsets; const uint32_t *srcSizes;
    uint32_t begin, end; int failed;
} Job;

static void *worker(void *arg)
{
    Job *j = (Job *)arg;
    uint32_t i;
    for (i = j->begin; i < j->end; i++) {
        size_t r;
        if (j->useLibzstd) {
            if (j->compress) r = z_compress(j->dst + j->dstOffsets[i], zso_compressBound(j->srcSizes[i]), j->src + j->srcOffsets[i], j->srcSizes[i], j->level);
            else r = z_decompress(j->dst + j->dstOffsets[i], j->dstCaps[i], j->src + j->srcOffsets[i], j->srcSizes[i]);
            if (z_iserror(r)) { j->failed = 1; j->dstSizes[i] = 0xFFFFFFFFu; } else j->dstSizes[i] = (uint32_t)r;
            continue;
        }
        if (j->compress) {
            size_t const cap = zso_compressBound(j->srcSizes[i]);
            r = zso_compress(j->dst + j->dstOffsets[i], cap, j->src + j->srcOffsets[i], j->srcSizes[i], j->level);
        } else {
            r = zso_decompress(j->dst + j->dstOffsets[i], j->dstCaps[i], j->src + j->srcOffsets[i], j->srcSizes[i]);
        }
        if (zso_isError(r)) { j->failed = 1; j->dstSizes[i] = 0xFFFFFFFFu - (uint32_t)(zso_errorCode(r) - 1); }
        else j->dstSizes[i] = (uint32_t)r;
    }
    return NULL;
}

static int run(Job proto, uint32_t n, int nThreads)
{
    pthread_t th[256];
    Job jobs[256];
    int t, failed = 0;
    if (nThreads < 1) nThreads = 1;
    if (nThreads > 256) nThreads = 256;
    if ((uint32_t)nThreads > n) nThreads = n ? (int)n : 1;
    for (t = 0; t < nThreads; t++) {
        jobs[t] = proto;
        jobs[t].begin = (uint32_t)((uint64_t)n * t / nThreads);
        jobs[t].end = (uint32_t)((uint64_t)n * (t + 1) / nThreads);
        jobs[t].failed = 0;
        if (pthread_create(&th[t], NULL, worker, &jobs[t])) return -1;
    }
    for (t = 0; t < nThreads; t++) { pthread_join(th[t], NULL); failed |= jobs[t].failed; }
    return failed;
}

int zso_compressBatch(void *dst, const uint64_t *dstOffsets, uint32_t *dstSizes,
                      const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                      uint32_t n, int level, int nThreads)
{
    Job j = { 1, level, 0, (uint8_t *)dst, dstOffsets, NULL, dstSizes, (const uint8_t *)src, srcOffsets, srcSizes, 0, 0, 0 };
    return run(j, n, nThreads);
}

int zso_decompressBatch(void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                        const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                        uint32_t n, int nThreads)
{
    Job j = { 0, 0, 0, (uint8_t *)dst, dstOffsets, dstCaps, dstSizes, (const uint8_t *)src, srcOffsets, srcSizes, 0, 0, 0 };
    return run(j, n, nThreads);
}

/* the same two drivers over upstream libzstd (dlopen): -2 if the box has no libzstd.so.1.  dst slots must hold zso_compressBound bytes. */
int zso_libzstdCompressBatch(void *dst, const uint64_t *dstOffsets, uint32_t *dstSizes,
                             const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                             uint32_t n, int level, int nThreads)
{
    Job j = { 1, level, 1, (uint8_t *)dst, dstOffsets, NULL, dstSizes, (const uint8_t *)src, srcOffsets, srcSizes, 0, 0, 0 };
    if (!loadLibzstd()) return -2;
    return run(j, n, nThreads);
}
int zso_libzstdDecompressBatch(void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                               const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                               uint32_t n, int nThreads)
{
    Job j = { 0, 0, 1, (uint8_t *)dst, dstOffsets, dstCaps, dstSizes, (const uint8_t *)src, srcOffsets, srcSizes, 0, 0, 0 };
    if (!loadLibzstd()) return -2;
    return run(j, n, nThreads);
}
