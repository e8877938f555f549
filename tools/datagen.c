/*
 * Synthetic workload generators for bench.py and the tests (integer-only, reproducible).
 * Bench/test infrastructure: not part of the codec.
 *
 * zipf_log: the "Zipf-token log stream" of SURVEY.md §8(d).  PRNG is xorshift128+ with the
 * update used by the reference's test helper (csharp/test/XorShift128Plus.cs:45-53).
 *   vocabulary: 50000 words; word i = 3 + (h % 10) lowercase letters, letter k = 'a' + ((h >> 5k) % 26),
 *               h = splitmix64(i)
 *   rank:       Zipf s=1 by integer CDF cum[i] = sum_{j<=i} floor(2^32 / (j+1)); sample = upper_bound(next() % total)
 *   line:       "<ts> <LEVEL> <8 + next()%17 words>\n", ts = 1700000000000 + 17*line + next()%13,
 *               LEVEL = table[next() % 16] with INFO x10, DEBUG x3, WARN x2, ERROR x1
 * The stream is generated in independent segments of SEG bytes (segment k is seeded (seedLo + k, seedHi) and its
 * line counter starts at k * 4096) so that it can be produced by several threads; a segment's last line is cut.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <pthread.h>

#define VOCAB 50000
#define SEG (1u << 20)

typedef struct { uint64_t s0, s1; } Rng;
static uint64_t rng_next(Rng *r)
{
    uint64_t x = r->s0, y = r->s1;
    r->s0 = y;
    x ^= x << 23;
    r->s1 = x ^ y ^ (x >> 17) ^ (y >> 26);
    return r->s1 + y;
}
static uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static char g_words[VOCAB][13];
static uint8_t g_wlen[VOCAB];
static uint64_t g_cum[VOCAB];
static uint64_t g_total;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void init_tables(void)
{
    uint64_t acc = 0;
    for (uint32_t i = 0; i < VOCAB; i++) {
        uint64_t h = splitmix64(i);
        uint32_t len = 3 + (uint32_t)(h % 10);
        g_wlen[i] = (uint8_t)len;
        for (uint32_t k = 0; k < len; k++) g_words[i][k] = (char)('a' + ((h >> (5 * k)) % 26));
        acc += (1ULL << 32) / (i + 1);
        g_cum[i] = acc;
    }
    g_total = acc;
}

static uint32_t sample_rank(Rng *r)
{
    uint64_t v = rng_next(r) % g_total;
    uint32_t lo = 0, hi = VOCAB;           /* first index with cum > v */
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (g_cum[mid] > v) hi = mid; else lo = mid + 1; }
    return lo;
}

static void gen_segment(uint8_t *dst, size_t n, uint64_t seedLo, uint64_t seedHi, uint64_t line)
{
    static const char *levels[16] = { "INFO","INFO","INFO","INFO","INFO","INFO","INFO","INFO","INFO","INFO",
                                      "DEBUG","DEBUG","DEBUG","WARN","WARN","ERROR" };
    Rng r = { seedLo, seedHi };
    size_t pos = 0;
    char buf[512];
    while (pos < n) {
        uint64_t ts = 1700000000000ULL + 17 * line + rng_next(&r) % 13;
        const char *lv = levels[rng_next(&r) % 16];
        uint32_t nt = 8 + (uint32_t)(rng_next(&r) % 17);
        int l = snprintf(buf, sizeof buf, "%llu %s", (unsigned long long)ts, lv);
        for (uint32_t t = 0; t < nt; t++) {
            uint32_t w = sample_rank(&r);
            buf[l++] = ' ';
            memcpy(buf + l, g_words[w], g_wlen[w]); l += g_wlen[w];
        }
        buf[l++] = '\n';
        { size_t c = (size_t)l < n - pos ? (size_t)l : n - pos; memcpy(dst + pos, buf, c); pos += c; }
        line++;
    }
}

typedef struct { uint8_t *dst; size_t n; uint64_t seedLo, seedHi; uint32_t first, step, nseg; } GJob;
static void *gworker(void *a)
{
    GJob *j = (GJob *)a;
    for (uint32_t k = j->first; k < j->nseg; k += j->step) {
        size_t off = (size_t)k * SEG;
        size_t len = j->n - off < SEG ? j->n - off : SEG;
        gen_segment(j->dst + off, len, j->seedLo + k, j->seedHi, (uint64_t)k * 4096);
    }
    return NULL;
}

/* fills dst[0..n) ; nThreads >= 1 */
void datagen_zipf_log(uint8_t *dst, size_t n, uint64_t seedLo, uint64_t seedHi, int nThreads)
{
    pthread_once(&g_once, init_tables);
    uint32_t nseg = (uint32_t)((n + SEG - 1) / SEG);
    if (nThreads < 1) nThreads = 1;
    if (nThreads > 64) nThreads = 64;
    pthread_t th[64]; GJob jobs[64];
    for (int t = 0; t < nThreads; t++) {
        jobs[t] = (GJob){ dst, n, seedLo, seedHi, (uint32_t)t, (uint32_t)nThreads, nseg };
        pthread_create(&th[t], NULL, gworker, &jobs[t]);
    }
    for (int t = 0; t < nThreads; t++) pthread_join(th[t], NULL);
}

/* one un-segmented stream (the SURVEY prototype's form), for the sha256 anchor */
void datagen_zipf_log_single(uint8_t *dst, size_t n, uint64_t seedLo, uint64_t seedHi)
{
    pthread_once(&g_once, init_tables);
    gen_segment(dst, n, seedLo, seedHi, 0);
}
