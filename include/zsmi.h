/*
 * zsmi.h -- C ABI of the MI355X-native Zstandard block codec (libzsmi.so).
 *
 * This is the drop-in boundary for the reference's public managed surface.  The reference
 * (epam/Zstandard) has no FFI of its own (pure C#/Java), so each entry point
 * below names the reference signature it replaces; a maintainer binds them with DllImport /
 * JNI as shown in INTEGRATION.md.  Plain pointers and sizes only; no torch types.
 *
 * Conventions kept from the reference:
 *   - results are sizes; errors are (size_t)-code with the codes of csharp/src/ZStdErrors.cs:61-90,
 *     tested with zsmi_isError() (ZStdErrors.cs:95-98: code > (size_t)-120);
 *   - nothing is retained past return for the one-shot calls (ZStdDecompress.cs:2174-2180);
 *   - one-shot calls are re-entrant; a zsmi_ctx must not be used from two threads at once.
 *
 * Every compute call runs on the GPU (HIP kernels for gfx950).  There is no CPU fallback:
 * if no device is usable the calls return ZSMI_error_GENERIC / a NULL context.
 */
#ifndef ZSMI_H
#define ZSMI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes : csharp/src/ZStdErrors.cs:61-90 ---- */
enum {
    ZSMI_error_no_error = 0, ZSMI_error_GENERIC = 1, ZSMI_error_prefix_unknown = 10,
    ZSMI_error_version_unsupported = 12, ZSMI_error_frameParameter_unsupported = 14,
    ZSMI_error_frameParameter_windowTooLarge = 16, ZSMI_error_corruption_detected = 20,
    ZSMI_error_checksum_wrong = 22, ZSMI_error_dictionary_corrupted = 30, ZSMI_error_dictionary_wrong = 32,
    ZSMI_error_parameter_unsupported = 40, ZSMI_error_parameter_outOfBound = 42,
    ZSMI_error_tableLog_tooLarge = 44, ZSMI_error_maxSymbolValue_tooLarge = 46,
    ZSMI_error_maxSymbolValue_tooSmall = 48, ZSMI_error_stage_wrong = 60, ZSMI_error_init_missing = 62,
    ZSMI_error_memory_allocation = 64, ZSMI_error_workSpace_tooSmall = 66,
    ZSMI_error_dstSize_tooSmall = 70, ZSMI_error_srcSize_wrong = 72, ZSMI_error_maxCode = 120
};

/* replaces: internal ZStdErrors.IsError (ZStdErrors.cs:95-98) */
unsigned zsmi_isError(size_t code);
/* replaces: commented upstream ZSTD_getErrorName (ZStd.cs:146-148) */
const char *zsmi_getErrorName(size_t code);
/* error code (0 if not an error) */
unsigned zsmi_getErrorCode(size_t code);

/* ------------------------------------------------------------------------------------------
 * One-shot calls on HOST buffers (the reference's public API shape).
 * ------------------------------------------------------------------------------------------ */

/* replaces: EPAM.Deltix.ZStd.ZStdDecompress.Decompress(byte[] dst, uint dstCapacity, byte[] src, uint srcSize)
 *           csharp/src/ZStdDecompress.cs:2182-2191  (and Java ZstdDecompressor.decompress, ZstdDecompressor.java:22)
 * Decodes every frame in src (concatenated and skippable frames included, ZStdDecompress.cs:2096-2160).
 * Returns the number of bytes written, or an error code. */
size_t zsmi_decompress(void *dst, size_t dstCapacity, const void *src, size_t srcSize);

/* replaces: ZSTD_decompress_usingDict(dctx, dst, dstCapacity, src, srcSize, dict, dictSize)  csharp/src/ZStdDecompress.cs:2162-2167
 * (internal in the reference: its public Decompress passes no dictionary, :2171).  dict: raw content, or a formatted dictionary
 * (magic 0xEC30A437: entropy tables + recent offsets + content, LoadEntropy :2378-2450); NULL / 0 = zsmi_decompress.
 * Errors as the reference: dictionary_corrupted (30), dictionary_wrong (32: the frame names another dictionary ID, :632-634). */
size_t zsmi_decompress_usingDict(void *dst, size_t dstCapacity, const void *src, size_t srcSize, const void *dict, size_t dictSize);

/* replaces: ZStdDecompress.GetDecompressedSize(byte[] src, uint srcSize)  csharp/src/ZStdDecompress.cs:590-622
 *           (Java ZstdDecompressor.getDecompressedSize, ZstdDecompressor.java:31)
 * Content size of the first frame; 0 if unknown, on error, or for a skippable frame. Host-only header parse. */
unsigned long long zsmi_getDecompressedSize(const void *src, size_t srcSize);

/* replaces: commented upstream declaration  size_t Compress(void* dst, size_t dstCapacity, void* src, size_t srcSize,
 *           int compressionLevel)  csharp/src/ZStd.cs:89-96   (the reference has no live compressor)
 * One frame for the whole input.  level <= 2: fast parameters, level >= 3: default parameters. */
size_t zsmi_compress(void *dst, size_t dstCapacity, const void *src, size_t srcSize, int level);

/* replaces: commented macro ZSTD_COMPRESSBOUND  csharp/src/ZStd.cs:144-145 (plus this codec's per-64 KiB block headers) */
size_t zsmi_compressBound(size_t srcSize);

/* ------------------------------------------------------------------------------------------
 * Batch calls: n independent chunks <-> n frames, the data-parallel hot path (no analogue in the
 * reference).  Chunk i is src[srcOffsets[i] .. +srcSizes[i]); its result goes to
 * dst[dstOffsets[i] ..).  Offsets/sizes arrays are HOST memory; src/dst/dstSizes are DEVICE memory
 * in the *Device calls and host memory in the *Host calls.
 * Per-chunk status: dstSizes[i] = bytes produced, or (uint32_t)-code on error (same codes as above).
 * ------------------------------------------------------------------------------------------ */
typedef struct zsmi_ctx zsmi_ctx;

/* device < 0: current HIP device.  stream: a hipStream_t (e.g. torch's current stream handle) or NULL
 * for a private stream.  Returns NULL if the device or its HIP runtime is unusable. */
zsmi_ctx *zsmi_createCtx(int device, void *hipStream);
void zsmi_freeCtx(zsmi_ctx *ctx);
/* block until everything queued on the context's stream has finished; returns 0 or an error code value */
int zsmi_sync(zsmi_ctx *ctx);

/* Asynchronous on the context's stream.  dstOffsets[i] must leave zsmi_compressBound(srcSizes[i]) bytes.
 * Chunks may be any size >= 0; the codec cuts them in 64 KiB blocks inside one frame. */
int zsmi_compressBatchDevice(zsmi_ctx *ctx, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                             uint32_t n, void *dDst, const uint64_t *dstOffsets, uint32_t *dDstSizes, int level);

/* Asynchronous on the context's stream.  Each frame i = src[srcOffsets[i] .. +srcSizes[i]) may hold several
 * concatenated / skippable frames (same rules as zsmi_decompress); dstCaps[i] is the room at dstOffsets[i]. */
int zsmi_decompressBatchDevice(zsmi_ctx *ctx, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                               uint32_t n, void *dDst, const uint64_t *dstOffsets, const uint32_t *dstCaps,
                               uint32_t *dDstSizes);

/* The same with one dictionary for every frame of the call (dDict: device memory).  Frames decoded with a dictionary take the
 * general kernel. */
int zsmi_decompressBatchDevice_usingDict(zsmi_ctx *ctx, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                         uint32_t n, void *dDst, const uint64_t *dstOffsets, const uint32_t *dstCaps,
                                         uint32_t *dDstSizes, const void *dDict, size_t dictSize);

/* Host-buffer forms: stage through device memory, run the device form, copy back, synchronise. */
int zsmi_compressBatchHost(zsmi_ctx *ctx, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                           uint32_t n, void *dst, const uint64_t *dstOffsets, uint32_t *dstSizes, int level);
int zsmi_decompressBatchHost(zsmi_ctx *ctx, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                             uint32_t n, void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps,
                             uint32_t *dstSizes);
int zsmi_decompressBatchHost_usingDict(zsmi_ctx *ctx, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                       uint32_t n, void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps,
                                       uint32_t *dstSizes, const void *dict, size_t dictSize);

/* Pack frames that sit at dstOffsets[] (sizes dDstSizes[], device) into one contiguous run at dPacked;
 * dPackedOffsets[n+1] (device, uint64) receives the running offsets.  Asynchronous. */
int zsmi_packFramesDevice(zsmi_ctx *ctx, const void *dFrames, const uint64_t *dstOffsets, const uint32_t *dSizes,
                          uint32_t n, void *dPacked, uint64_t *dPackedOffsets);

/* ---- measurement hooks (bench.py): HIP-event timing of the kernels launched on the context's stream by the
 *      last batch call; one entry per kernel name, seconds are summed over launches.  Returns entries written.
 *      on = 1: events around every launch; on = 2: only around the dominant kernel of each direction (k_lz_walk*, k_dec_execute):
 *      ten event records a step between five short kernels are not free, the timed region of bench.py carries two. ---- */
typedef struct { char name[48]; double seconds; uint32_t launches; } zsmi_kernel_time;
int zsmi_enableKernelTiming(zsmi_ctx *ctx, int on);
int zsmi_getKernelTimes(zsmi_ctx *ctx, zsmi_kernel_time *out, int maxEntries);

/* device scratch the context holds for decoding after the last zsmi_decompressBatch* call (bytes): sized by that call's items in flight and
 * its largest capacity; a later, smaller call gives most of it back */
size_t zsmi_decodeScratchBytes(zsmi_ctx *ctx);

/* Releases the per-device contexts the one-shot calls (zsmi_compress / zsmi_decompress*) keep.  For embedders that unload the library: nothing
 * is released from an exit-time destructor (the HIP runtime may be gone by then); call this before dlclose.  No one-shot call may be running. */
void zsmi_shutdown(void);

/* library / device description, for logs */
const char *zsmi_versionString(void);

#ifdef __cplusplus
}
#endif
#endif
