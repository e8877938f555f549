/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Multi-threaded drivers over the scalar oracle
 * functions: n independent chunks, one frame each, static partition over nThreads.
 * Used by bench.py's cpu_baseline leg (timed on host cores) and by tests.
 */
#include "zso_oracle.h"
#include <pthread.h>
#include <stdlib.h>

typedef struct {
    int compress, level;
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
    Job j = { 1, level, (uint8_t *)dst, dstOffsets, NULL, dstSizes, (const uint8_t *)src, srcOffsets, srcSizes, 0, 0, 0 };
    return run(j, n, nThreads);
}

int zso_decompressBatch(void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                        const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                        uint32_t n, int nThreads)
{
    Job j = { 0, 0, (uint8_t *)dst, dstOffsets, dstCaps, dstSizes, (const uint8_t *)src, srcOffsets, srcSizes, 0, 0, 0 };
    return run(j, n, nThreads);
}
