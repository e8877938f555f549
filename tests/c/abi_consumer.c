/* A plain-C caller of include/zsmi.h, the way a P/Invoke or JNI shim reaches it: one-shot host calls with the reference's
 * conventions (ZStdDecompress.cs:2182-2191: sizes in, size or (size_t)-code out, nothing retained) and the batch calls.
 * Built with gcc against libzsmi.so by tests/test_c_abi.py; exits 0 when every check holds. */
#include "zsmi.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

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
    printf("abi_consumer ok: %zu -> %zu bytes, %s\n", n, csize, zsmi_versionString());
    (void)argc; (void)argv;
    return 0;
}
