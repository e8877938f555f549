/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see zso_decoder.c / zso_encoder.c headers).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef ZSO_ORACLE_H
#define ZSO_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* error codes: reference csharp/src/ZStdErrors.cs:61-90 */
enum {
    ZSO_no_error = 0, ZSO_GENERIC = 1, ZSO_prefix_unknown = 10, ZSO_version_unsupported = 12,
    ZSO_frameParameter_unsupported = 14, ZSO_frameParameter_windowTooLarge = 16,
    ZSO_corruption_detected = 20, ZSO_checksum_wrong = 22, ZSO_dictionary_corrupted = 30,
    ZSO_dictionary_wrong = 32, ZSO_dictionaryCreation_failed = 34, ZSO_parameter_unsupported = 40,
    ZSO_parameter_outOfBound = 42, ZSO_tableLog_tooLarge = 44, ZSO_maxSymbolValue_tooLarge = 46,
    ZSO_maxSymbolValue_tooSmall = 48, ZSO_stage_wrong = 60, ZSO_init_missing = 62,
    ZSO_memory_allocation = 64, ZSO_workSpace_tooSmall = 66, ZSO_dstSize_tooSmall = 70,
    ZSO_srcSize_wrong = 72, ZSO_frameIndex_tooLarge = 100, ZSO_seekableIO = 102, ZSO_maxCode = 120
};

/* oracle D : restatement of ZStdDecompress.Decompress / GetDecompressedSize */
size_t zso_decompress(void *dst, size_t dstCapacity, const void *src, size_t srcSize);
/* ZSTD_decompress_usingDict (ZStdDecompress.cs:2162): raw-content or formatted (magic 0xEC30A437) dictionary; NULL / 0 = none */
size_t zso_decompress_usingDict(void *dst, size_t dstCapacity, const void *src, size_t srcSize, const void *dict, size_t dictSize);
unsigned long long zso_getDecompressedSize(const void *src, size_t srcSize);
unsigned zso_isError(size_t code);
unsigned zso_errorCode(size_t code);
uint64_t zso_xxh64(const void *input, size_t len, uint64_t seed);
void zso_statsReset(void);
void zso_statsGet(uint32_t *out32);   /* 32 counters, see zso_decoder.c */

/* oracle E : scalar CPU statement of this repo's block encoder (the algorithm the HIP kernels run) */
size_t zso_compressBound(size_t srcSize);
size_t zso_compress(void *dst, size_t dstCapacity, const void *src, size_t srcSize, int level);
/* multi-threaded drivers used only by bench.py's cpu_baseline leg: n independent chunks */
int zso_compressBatch(void *dst, const uint64_t *dstOffsets, uint32_t *dstSizes,
                      const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                      uint32_t n, int level, int nThreads);
int zso_decompressBatch(void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                        const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                        uint32_t n, int nThreads);
/* the same drivers over upstream libzstd loaded with dlopen (a labelled yardstick, NOT the reference); -2: no libzstd.so.1 */
int zso_libzstdCompressBatch(void *dst, const uint64_t *dstOffsets, uint32_t *dstSizes,
                             const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                             uint32_t n, int level, int nThreads);
int zso_libzstdDecompressBatch(void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                               const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                               uint32_t n, int nThreads);
#ifdef __cplusplus
}
#endif
#endif
