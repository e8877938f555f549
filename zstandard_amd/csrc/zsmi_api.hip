// libzsmi.so : C ABI (include/zsmi.h) over the HIP kernels.  Host side of the codec: contexts, device
// workspaces, block planning, kernel launches, staging for host-buffer calls.  No CPU codec path exists
// in this library: every compress/decompress call launches the gfx950 kernels or fails.
//
// Single translation unit: the kernel sources are included so that launches and kernels share one code object.
#include "lz_kernels.hip"
#include "entropy_kernels.hip"
#include "decode_kernels.hip"
#include "decode_fast.hip"
#include "../../include/zsmi.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define ZSMI_ERR(code) ((size_t)0 - (size_t)(code))

extern "C" unsigned zsmi_isError(size_t code) { return code > ZSMI_ERR(ZSMI_error_maxCode); }     // ZStdErrors.cs:95-98
extern "C" unsigned zsmi_getErrorCode(size_t code) { return zsmi_isError(code) ? (unsigned)(0 - code) : 0; }
extern "C" const char *zsmi_getErrorName(size_t code)
{
    switch (zsmi_getErrorCode(code)) {
    case 0: return "No error detected";
    case ZSMI_error_GENERIC: return "Error (generic)";
    case ZSMI_error_prefix_unknown: return "Unknown frame descriptor";
    case ZSMI_error_version_unsupported: return "Version not supported";
    case ZSMI_error_frameParameter_unsupported: return "Unsupported frame parameter";
    case ZSMI_error_frameParameter_windowTooLarge: return "Frame requires too much memory for decoding";
    case ZSMI_error_corruption_detected: return "Corrupted block detected";
    case ZSMI_error_checksum_wrong: return "Restored data doesn't match checksum";
    case ZSMI_error_dictionary_corrupted: return "Dictionary is corrupted";
    case ZSMI_error_dictionary_wrong: return "Dictionary mismatch";
    case ZSMI_error_parameter_unsupported: return "Unsupported parameter";
    case ZSMI_error_parameter_outOfBound: return "Parameter is out of bound";
    case ZSMI_error_tableLog_tooLarge: return "tableLog requires too much memory : unsupported";
    case ZSMI_error_maxSymbolValue_tooLarge: return "Unsupported max Symbol Value : too large";
    case ZSMI_error_maxSymbolValue_tooSmall: return "Specified maxSymbolValue is too small";
    case ZSMI_error_stage_wrong: return "Operation not authorized at current processing stage";
    case ZSMI_error_init_missing: return "Context should be init first";
    case ZSMI_error_memory_allocation: return "Allocation error : not enough memory";
    case ZSMI_error_workSpace_tooSmall: return "workSpace buffer is not large enough";
    case ZSMI_error_dstSize_tooSmall: return "Destination buffer is too small";
    case ZSMI_error_srcSize_wrong: return "Src size is incorrect";
    default: return "Unspecified error code";
    }
}
// ZSMI_SOURCE_FP: fingerprint of zstandard_amd/csrc the library was built from (zstandard_amd/_lib.py passes it; a library that ships with the
// sources is rebuilt when they differ, so a measurement can name the code it ran)
#ifndef ZSMI_SOURCE_FP
#define ZSMI_SOURCE_FP "unknown"
#endif
extern "C" const char *zsmi_versionString(void) { return "zsmi 0.3 (gfx950 HIP kernels; zstd frame format, decoder semantics of epam/Zstandard = zstd v1.3.4; sources " ZSMI_SOURCE_FP ")"; }

extern "C" size_t zsmi_compressBound(size_t srcSize)
{
    return srcSize + (srcSize >> 8) + ((srcSize < (128u << 10)) ? (((128u << 10) - srcSize) >> 11) : 0) + 3 * (srcSize / ZS_BLOCK_MAX + 1) + 18;
}

// ---- host-only frame header parse: ZStdDecompress.cs:421-499, 518-532, 617-622 ----
static uint32_t h_rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
extern "C" unsigned long long zsmi_getDecompressedSize(const void *srcv, size_t srcSize)
{
    const uint8_t *src = (const uint8_t *)srcv;
    if (srcSize < 5) return 0;
    const uint32_t magic = h_rd32(src);
    if (magic != 0xFD2FB528u) return 0;                    // skippable -> 0, unknown -> 0
    const uint32_t fhd = src[4];
    const uint32_t dictIDCode = fhd & 3, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
    const size_t didSize = dictIDCode == 3 ? 4 : dictIDCode, fcsSize = fcsID == 0 ? 0 : (fcsID == 1 ? 2 : (fcsID == 2 ? 4 : 8));
    const size_t fhs = 5 + !singleSegment + didSize + fcsSize + (singleSegment && !fcsID);
    if (srcSize < fhs) return 0;
    if (fhd & 0x08) return 0;
    size_t pos = 5;
    if (!singleSegment) { const uint32_t wl = src[pos++]; if ((wl >> 3) + 10 > 30) return 0; }
    pos += didSize;
    unsigned long long fcs;
    switch (fcsID) {
    case 0: if (!singleSegment) return 0; fcs = src[pos]; break;
    case 1: fcs = ((unsigned long long)src[pos] | ((unsigned long long)src[pos + 1] << 8)) + 256; break;
    case 2: fcs = h_rd32(src + pos); break;
    default: fcs = (unsigned long long)h_rd32(src + pos) | ((unsigned long long)h_rd32(src + pos + 4) << 32); break;
    }
    return (fcs >= 0xFFFFFFFFFFFFFFFEull) ? 0 : fcs;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + (n >> 3) + 4096;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
        cap = want; return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf {
    void *p = nullptr; size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + (n >> 3) + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        cap = want; return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct TimedLaunch { const char *name; hipEvent_t a, b; };

struct zsmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    uint32_t maxBlocksInFlight = 16384;   // ZSMI_BLOCKS_IN_FLIGHT: 64 KiB blocks per sub-batch (scratch ~0.6 MiB a block, reserved for what a call needs); 2 GiB of 128 KiB chunks: 8192: 86.5, 16384: 88.2, 32768: 89.4 GiB/s
    // compress workspace: plan (shared) + one scratch set per internal stream ("lane")
    DevBuf dBlocks, dChunks, dUnits;     // dUnits: small units (<= 64 KiB) first, then big ones, each in chunk order
    struct Scratch { DevBuf dDist, dDistHi, dCand, dRecs, dRes, dSeqs, dHdrs, dLits, dStreams, dLitSec, dSeqSec, dMetas; hipStream_t stream = nullptr; hipEvent_t done = nullptr; };
    static const int kMaxLanes = 8;
    Scratch lanes[kMaxLanes];
    int nLanes = 1;
    int stopAfterWalk = 0;                 // ZSMI_STOP_AFTER_WALK (debug-hooks build, tools/walk_check.py): the entropy kernels are not launched
    int stopLit = 0, stopSeq = 0;          // timing aids of a -DZSMI_DEBUG_HOOKS build (ZSMI_STOP_LIT / ZSMI_STOP_SEQ): end a kernel after a stage; always 0 in the product
    hipEvent_t evStart = nullptr;
    PinBuf hBlocks, hChunks, hUnits;
    std::vector<uint32_t> smallBefore, bigBefore;   // per chunk (n + 1 entries): small / big units in front of it
    uint32_t planSmall = 0, planBig = 0;
    std::vector<uint64_t> planKey;       // copy of (srcOffsets, srcSizes, dstOffsets) the device-side plan was built from
    uint64_t planBlocks = 0; uint32_t planMaxChunkBlocks = 1;
    // decompress workspace
    DevBuf dItems, dLitScratch, dFastDesc, dHufTabs, dSeqTabs, dSeqOut, dSeqLists;     // decode: items, literal scratch, fast-path tables and sequences, the blocks of each table class
    bool decodeFast = true;              // ZSMI_DEC_FAST=0: general kernel only
    uint32_t maxItemsInFlight = 65536;   // ZSMI_ITEMS_IN_FLIGHT: items per decode launch (scratch: ~263 KiB per item and block slot, one slot unless an item can hold two 64 KiB blocks; cut down to half the free device memory).
                                         // Every decode kernel is a long dependent chain per item: a launch is one to three rounds of workgroups and its
                                         // last round is mostly tail, so big launches pay (16384 frames of 32 KiB: 82 GiB/s, 57344: 104 GiB/s)
    PinBuf hItems2[2]; hipEvent_t hItemsEv[2] = { nullptr, nullptr }; bool hItemsBusy[2] = { false, false }; uint32_t decodeCalls = 0;    // the decode item list: two pinned buffers taken in turn
    DevBuf dPoolLit;                         // the general decode kernel's literal buffers: one per wavefront of its pool
    uint32_t seqLog9Group = 0;               // experiment: force the 2.5 KiB sequence-table class to 16 or 4 items a wavefront (ZSMI_SEQ_LOG9_G; 0: the heuristic)
    uint32_t decodeFuseBelow = 0;            // ZSMI_DEC_FUSE_BELOW given: (item, block) pairs of a call up to which the entropy kernels are one launch (0: never) - instead of the rule below
    bool decodeFuseSet = false;
    int execWaves = 0;                       // ZSMI_EXEC_WAVES=7 / 8: force the execute kernel's form for one-block items (0: by the call's size, below)
    uint32_t cus = 256;                      // compute units of the device (rounds of workgroups a launch takes)
    uint32_t decodePool = 3072;              // wavefronts of that pool (ZSMI_DEC_POOL): the chip holds 10 a CU x 256
    size_t lastDecodeScratch = 0;            // bytes of scratch the last decode call needed (INTEGRATION.md states them)
    // staging for host-buffer calls
    DevBuf sSrc, sDst, sSizes, sDict, sPack, sPackOff;
    PinBuf hPack;
    // timing
    int timing = 0;                      // 1: events around every launch; 2: only around the dominant kernels (k_lz_walk*, k_dec_execute)
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> eventPool;
};

static hipEvent_t getEvent(zsmi_ctx *c)
{
    if (!c->eventPool.empty()) { hipEvent_t e = c->eventPool.back(); c->eventPool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
static inline bool dominantKernel(const char *name) { return strncmp(name, "k_lz_walk", 9) == 0 || strncmp(name, "k_dec_execute", 13) == 0; }
#define LAUNCH_ON(ctx, strm, name, kernel, grid, block, lds, ...) do { \
        TimedLaunch tl_{name, nullptr, nullptr}; \
        const bool timed_ = (ctx)->timing == 1 || ((ctx)->timing == 2 && dominantKernel(name)); \
        if (timed_) { tl_.a = getEvent(ctx); tl_.b = getEvent(ctx); (void)hipEventRecord(tl_.a, (strm)); } \
        hipLaunchKernelGGL(kernel, grid, block, lds, (strm), __VA_ARGS__); \
        if (timed_) { (void)hipEventRecord(tl_.b, (strm)); (ctx)->launches.push_back(tl_); } \
    } while (0)
#define LAUNCH(ctx, name, kernel, grid, block, lds, ...) LAUNCH_ON(ctx, (ctx)->stream, name, kernel, grid, block, lds, __VA_ARGS__)

extern "C" zsmi_ctx *zsmi_createCtx(int device, void *hipStream)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return nullptr;
    zsmi_ctx *c = new zsmi_ctx();
    if (device < 0) { if (hipGetDevice(&c->device) != hipSuccess) { delete c; return nullptr; } }
    else { c->device = device; if (hipSetDevice(device) != hipSuccess) { delete c; return nullptr; } }
    if (hipStream) { c->stream = (hipStream_t)hipStream; c->ownStream = false; }
    else { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; } c->ownStream = true; }
    // dynamic LDS beyond the 64 KiB default
    {   // dynamic LDS beyond the 64 KiB default: refused requests fail here, not at the first launch
        bool ok = true;
        ok &= hipFuncSetAttribute((const void *)k_lz_candidates<ZS_TABLE_LOG_SMALL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZS_CAND_LDS(ZS_TABLE_LOG_SMALL, 1)) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *)k_lz_candidates<ZS_TABLE_LOG_SMALL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZS_CAND_LDS(ZS_TABLE_LOG_SMALL, 2)) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *)k_lz_candidates<ZS_TABLE_LOG_BIG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZS_CAND_LDS(ZS_TABLE_LOG_BIG, 1)) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *)k_lz_candidates<ZS_TABLE_LOG_BIG, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZS_CAND_LDS(ZS_TABLE_LOG_BIG, 2)) == hipSuccess;
        // k_lz_walk addresses its (dynamic) LDS from 0: that holds as long as the kernel has no static LDS in front of it
        auto walkOk = [](const void *fn, size_t lds) {
            hipFuncAttributes fa;
            return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess && hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.sharedSizeBytes == 0;
        };
        ok &= walkOk((const void *)ZS_WALK_KERNEL(4, 8, false, 9), ZS_WALK_LDS(ZS_BLOCK_MAX)) && walkOk((const void *)ZS_WALK_KERNEL(4, 8, true, 9), ZS_WALK_LDS(ZS_UNIT_MAX));
        ok &= walkOk((const void *)ZS_WALK_KERNEL(4, 4, false, 8), ZS_WALK_LDS(ZS_BLOCK_MAX)) && walkOk((const void *)ZS_WALK_KERNEL(4, 4, true, 8), ZS_WALK_LDS(ZS_UNIT_MAX));
        ok &= walkOk((const void *)ZS_WALK_KERNEL(8, 8, false, 8), ZS_WALK_LDS(ZS_BLOCK_MAX)) && walkOk((const void *)ZS_WALK_KERNEL(8, 8, true, 8), ZS_WALK_LDS(ZS_UNIT_MAX));
        if (!ok) { (void)hipGetLastError(); if (c->ownStream) (void)hipStreamDestroy(c->stream); delete c; return nullptr; }
    }
    if (const char *e = getenv("ZSMI_BLOCKS_IN_FLIGHT")) { long v = atol(e); if (v >= 64) c->maxBlocksInFlight = (uint32_t)v; }
    if (const char *e = getenv("ZSMI_DEC_FAST")) c->decodeFast = atoi(e) != 0;
    if (const char *e = getenv("ZSMI_SEQ_LOG9_G")) c->seqLog9Group = (uint32_t)atol(e);
    if (const char *e = getenv("ZSMI_DEC_FUSE_BELOW")) { c->decodeFuseBelow = (uint32_t)atol(e); c->decodeFuseSet = true; }
    if (const char *e = getenv("ZSMI_EXEC_WAVES")) c->execWaves = atoi(e);
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0) c->cus = (uint32_t)v; }
    if (const char *e = getenv("ZSMI_DEC_POOL")) { long v = atol(e); if (v >= 2 && v <= (1 << 20)) c->decodePool = (uint32_t)v; }
    if (const char *e = getenv("ZSMI_ITEMS_IN_FLIGHT")) { long v = atol(e); if (v >= 64 && v <= (1 << 20)) c->maxItemsInFlight = (uint32_t)v; }
#ifdef ZSMI_DEBUG_HOOKS
    if (const char *e = getenv("ZSMI_STOP_LIT")) c->stopLit = atoi(e);
    if (const char *e = getenv("ZSMI_STOP_AFTER_WALK")) c->stopAfterWalk = atoi(e);
    if (const char *e = getenv("ZSMI_STOP_SEQ")) c->stopSeq = atoi(e);
#endif
    if (const char *e = getenv("ZSMI_LANES")) { long v = atol(e); if (v >= 1 && v <= zsmi_ctx::kMaxLanes) c->nLanes = (int)v; }
    // sub-batches of one call run on internal streams so that the latency-bound kernels of different sub-batches overlap
    for (int i = 0; i < c->nLanes; i++) {
        if (hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->lanes[i].done, hipEventDisableTiming) != hipSuccess) { c->nLanes = i; break; }
    }
    if (c->nLanes == 0 || hipEventCreateWithFlags(&c->evStart, hipEventDisableTiming) != hipSuccess) { zsmi_freeCtx(c); return nullptr; }
    return c;
}
extern "C" void zsmi_freeCtx(zsmi_ctx *c)
{
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : { &c->dBlocks, &c->dChunks, &c->dUnits, &c->dItems, &c->dPoolLit, &c->dLitScratch, &c->dFastDesc, &c->dHufTabs, &c->dSeqTabs, &c->dSeqOut, &c->dSeqLists, &c->sSrc, &c->sDst, &c->sSizes, &c->sDict, &c->sPack, &c->sPackOff }) b->release();
    for (int i = 0; i < zsmi_ctx::kMaxLanes; i++) {
        zsmi_ctx::Scratch &L = c->lanes[i];
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        for (DevBuf *b : { &L.dDist, &L.dDistHi, &L.dCand, &L.dRecs, &L.dRes, &L.dSeqs, &L.dHdrs, &L.dLits, &L.dStreams, &L.dLitSec, &L.dSeqSec, &L.dMetas }) b->release();
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    for (int i = 0; i < 2; i++) if (c->hItemsEv[i]) (void)hipEventDestroy(c->hItemsEv[i]);
    for (PinBuf *b : { &c->hBlocks, &c->hChunks, &c->hUnits, &c->hItems2[0], &c->hItems2[1], &c->hPack }) b->release();
    for (auto &tl : c->launches) { (void)hipEventDestroy(tl.a); (void)hipEventDestroy(tl.b); }
    for (auto e : c->eventPool) (void)hipEventDestroy(e);
    if (c->ownStream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" void zsmi_freeCtx(zsmi_ctx *c);
extern "C" int zsmi_sync(zsmi_ctx *c)
{
    if (!c) return ZSMI_error_init_missing;
    return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : ZSMI_error_GENERIC;
}
extern "C" int zsmi_enableKernelTiming(zsmi_ctx *c, int on)
{
    if (!c) return ZSMI_error_init_missing;
    c->timing = (on == 2) ? 2 : (on != 0);
    for (auto &tl : c->launches) { c->eventPool.push_back(tl.a); c->eventPool.push_back(tl.b); }
    c->launches.clear();
    return 0;
}
extern "C" int zsmi_getKernelTimes(zsmi_ctx *c, zsmi_kernel_time *out, int maxEntries)
{
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    int n = 0;
    for (auto &tl : c->launches) {
        float ms = 0; (void)hipEventElapsedTime(&ms, tl.a, tl.b);
        int k = 0;
        for (; k < n; k++) if (!strcmp(out[k].name, tl.name)) break;
        if (k == n) { if (n >= maxEntries) continue; memset(&out[n], 0, sizeof out[n]); strncpy(out[n].name, tl.name, sizeof(out[n].name) - 1); n++; }
        out[k].seconds += ms * 1e-3; out[k].launches++;
        c->eventPool.push_back(tl.a); c->eventPool.push_back(tl.b);
    }
    c->launches.clear();
    return n;
}

// ---------------------------------------------------------------------------------------------
// compress
// ---------------------------------------------------------------------------------------------
extern "C" int zsmi_compressBatchDevice(zsmi_ctx *c, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                        uint32_t n, void *dDst, const uint64_t *dstOffsets, uint32_t *dDstSizes, int level)
{
    if (!c) return ZSMI_error_init_missing;
    if (n == 0) return 0;
    if (hipSetDevice(c->device) != hipSuccess) return ZSMI_error_GENERIC;
    // level <= 2: short table only ("fast"), walk ranges of 512 bytes; level >= 3: short + long table ("double"), ranges of 256 bytes;
    // level >= 4 scores 8 candidates a step instead of 4 (paramsForLevel in oracle/zso_encoder.c)
    const bool useLong = level >= 3;
    const int walkLog = level <= 2 ? 9 : 8;
    // plan: chunks -> blocks.  The device-side plan is reused when the chunk layout repeats (steady-state batches).
    // (compared in place: a call of a repeating layout allocates and copies nothing)
    bool samePlan = c->planKey.size() == (size_t)n * 3 + 1 && c->planKey[0] == n;
    for (uint32_t i = 0; samePlan && i < n; i++) samePlan = c->planKey[1 + i] == srcOffsets[i] && c->planKey[1 + n + i] == dstOffsets[i] && c->planKey[1 + 2 * (size_t)n + i] == srcSizes[i];
    uint64_t nBlocks; uint32_t maxChunkBlocks;
    if (samePlan) { nBlocks = c->planBlocks; maxChunkBlocks = c->planMaxChunkBlocks; }
    else {
        std::vector<uint64_t> key((size_t)n * 3 + 1);
        key[0] = n;
        for (uint32_t i = 0; i < n; i++) { key[1 + i] = srcOffsets[i]; key[1 + n + i] = dstOffsets[i]; key[1 + 2 * (size_t)n + i] = srcSizes[i]; }
        nBlocks = 0;
        for (uint32_t i = 0; i < n; i++) nBlocks += srcSizes[i] ? (srcSizes[i] + ZS_BLOCK_MAX - 1) / ZS_BLOCK_MAX : 1;
        if (nBlocks > 0x7FFFFFFFull) return ZSMI_error_srcSize_wrong;
        if (!c->hChunks.reserve(sizeof(ZsChunkDesc) * n) || !c->hBlocks.reserve(sizeof(ZsBlockDesc) * nBlocks)) return ZSMI_error_memory_allocation;
        if (!c->dChunks.reserve(sizeof(ZsChunkDesc) * n) || !c->dBlocks.reserve(sizeof(ZsBlockDesc) * nBlocks)) return ZSMI_error_memory_allocation;
        if (!c->hUnits.reserve(sizeof(ZsUnitDesc) * nBlocks) || !c->dUnits.reserve(sizeof(ZsUnitDesc) * nBlocks)) return ZSMI_error_memory_allocation;
        // the pinned plan buffers may still feed a previous asynchronous copy
        if (hipStreamSynchronize(c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        ZsChunkDesc *hc0 = (ZsChunkDesc *)c->hChunks.p; ZsBlockDesc *hb = (ZsBlockDesc *)c->hBlocks.p;
        uint32_t b = 0; maxChunkBlocks = 1;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t nb = srcSizes[i] ? (srcSizes[i] + ZS_BLOCK_MAX - 1) / ZS_BLOCK_MAX : 1;
            hc0[i].srcOff = srcOffsets[i]; hc0[i].dstOff = dstOffsets[i]; hc0[i].size = srcSizes[i]; hc0[i].firstBlock = b; hc0[i].nBlocks = nb; hc0[i].pad = 0;
            for (uint32_t k = 0; k < nb; k++, b++) {
                hb[b].srcOff = srcOffsets[i] + (uint64_t)k * ZS_BLOCK_MAX;
                const uint64_t left = (uint64_t)srcSizes[i] - (uint64_t)k * ZS_BLOCK_MAX;
                hb[b].size = (uint32_t)(left < ZS_BLOCK_MAX ? left : ZS_BLOCK_MAX);
                hb[b].chunk = i; hb[b].firstInChunk = (k == 0); hb[b].lastInChunk = (k + 1 == nb);
            }
            if (nb > maxChunkBlocks) maxChunkBlocks = nb;
        }
        // LZ units: every 128 KiB of a chunk (two blocks); a unit of <= 64 KiB goes to the small-unit kernels
        c->smallBefore.assign((size_t)n + 1, 0); c->bigBefore.assign((size_t)n + 1, 0);
        uint32_t nSmall = 0, nBig = 0;
        for (uint32_t i = 0; i < n; i++) {
            c->smallBefore[i] = nSmall; c->bigBefore[i] = nBig;
            for (uint64_t o = 0; o < srcSizes[i]; o += ZS_UNIT_MAX) { if ((uint64_t)srcSizes[i] - o > ZS_BLOCK_MAX) nBig++; else nSmall++; }
        }
        c->smallBefore[n] = nSmall; c->bigBefore[n] = nBig;
        {
            ZsUnitDesc *hu = (ZsUnitDesc *)c->hUnits.p;
            uint32_t is = 0, ib = nSmall;
            for (uint32_t i = 0; i < n; i++)
                for (uint64_t o = 0; o < srcSizes[i]; o += ZS_UNIT_MAX) {
                    const uint64_t left = (uint64_t)srcSizes[i] - o;
                    ZsUnitDesc &u = hu[left > ZS_BLOCK_MAX ? ib++ : is++];
                    u.srcOff = srcOffsets[i] + o; u.size = (uint32_t)(left < ZS_UNIT_MAX ? left : ZS_UNIT_MAX); u.firstBlock = hc0[i].firstBlock + (uint32_t)(o / ZS_BLOCK_MAX);
                }
            if (nSmall + nBig && hipMemcpyAsync(c->dUnits.p, hu, sizeof(ZsUnitDesc) * (nSmall + nBig), hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        }
        c->planSmall = nSmall; c->planBig = nBig;
        if (hipMemcpyAsync(c->dChunks.p, hc0, sizeof(ZsChunkDesc) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        if (hipMemcpyAsync(c->dBlocks.p, hb, sizeof(ZsBlockDesc) * nBlocks, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        c->planKey.swap(key); c->planBlocks = nBlocks; c->planMaxChunkBlocks = maxChunkBlocks;
    }
    const ZsChunkDesc *hc = (const ZsChunkDesc *)c->hChunks.p;
    // sub-batches of whole chunks, dealt round-robin to the internal streams; each stream owns a scratch set
    const int nLanes = (int)std::min<uint64_t>((uint64_t)c->nLanes, std::max<uint64_t>(1, nBlocks / 256));
    uint32_t cap = (uint32_t)std::min<uint64_t>((nBlocks + nLanes - 1) / nLanes, std::max<uint32_t>(64, c->maxBlocksInFlight / (uint32_t)nLanes));
    if (cap < maxChunkBlocks) cap = maxChunkBlocks;
    for (int i = 0; i < nLanes; i++) {
        zsmi_ctx::Scratch &L = c->lanes[i];
        if (!L.dDist.reserve((size_t)cap * ZS_BLOCK_MAX * 2 + 256) || !L.dDistHi.reserve((size_t)cap * (ZS_BLOCK_MAX / 8) + 256) || !L.dCand.reserve((size_t)cap * 2 * sizeof(uint32_t) + 64) || !L.dRecs.reserve(((size_t)cap * (ZS_BLOCK_MAX / 4) + 64) * sizeof(uint2)) || !L.dRes.reserve((size_t)cap * ZS_RES_PER_BLOCK * sizeof(uint4) + ((size_t)8 << 20)) || !L.dSeqs.reserve((size_t)cap * ZS_WALK_RANGES * ZS_SEQ_PER_RANGE * sizeof(ZsSeqRec)) ||
            !L.dHdrs.reserve((size_t)cap * ZS_WALK_RANGES * sizeof(ZsRangeHdr)) || !L.dLits.reserve((size_t)cap * (ZS_BLOCK_MAX + 64)) ||
            !L.dStreams.reserve((size_t)cap * 4 * ZS_STREAM_STRIDE) || !L.dLitSec.reserve((size_t)cap * ZS_LITSEC_STRIDE) ||
            !L.dSeqSec.reserve((size_t)cap * ZS_SEQSEC_STRIDE) || !L.dMetas.reserve((size_t)cap * sizeof(ZsBlockMeta))) return ZSMI_error_memory_allocation;
    }
    // One lane (the default): its kernels go to the caller's stream itself.  (Through a stream of the lane's own - an event from the caller's
    // stream in front, one back behind - every call crossed from one hardware queue to another twice: ~0.01 ms a crossing on most boxes of
    // the pool, ~0.09 on some - a bench line of 126 GiB/s where the same binary's kernels added up to 137.)
    const bool direct = nLanes == 1;
    if (!direct) {
        if (hipEventRecord(c->evStart, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        for (int i = 0; i < nLanes; i++) if (hipStreamWaitEvent(c->lanes[i].stream, c->evStart, 0) != hipSuccess) return ZSMI_error_GENERIC;
    }
    uint32_t chunk0 = 0; int turn = 0;
    while (chunk0 < n) {
        uint32_t chunk1 = chunk0, nb = 0;
        while (chunk1 < n && (nb == 0 || nb + hc[chunk1].nBlocks <= cap)) { nb += hc[chunk1].nBlocks; chunk1++; }
        const uint32_t block0 = hc[chunk0].firstBlock;
        const ZsBlockDesc *dB = (const ZsBlockDesc *)c->dBlocks.p + block0;
        zsmi_ctx::Scratch &L = c->lanes[turn % nLanes]; turn++;
        hipStream_t st = direct ? c->stream : L.stream;
        // match search per LZ unit: small units (one block) and big units (two blocks) have their own kernel shapes
        const uint32_t s0 = c->smallBefore[chunk0], ns = c->smallBefore[chunk1] - s0, b0 = c->bigBefore[chunk0], nbig = c->bigBefore[chunk1] - b0;
        const ZsUnitDesc *dUS = (const ZsUnitDesc *)c->dUnits.p + s0, *dUB = (const ZsUnitDesc *)c->dUnits.p + c->planSmall + b0;
        #define CAND_LAUNCH(name, TL, NT, cnt, du) LAUNCH_ON(c, st, name, (k_lz_candidates<TL, NT>), dim3(cnt), dim3(64 * ZS_CAND_WAVES(NT)), ZS_CAND_LDS(TL, NT), (const uint8_t *)dSrc, du, block0, \
                          (uint16_t *)L.dDist.p, (uint8_t *)L.dDistHi.p, (uint32_t *)L.dCand.p)
        if (ns) { if (useLong) CAND_LAUNCH("k_lz_candidates", ZS_TABLE_LOG_SMALL, 2, ns, dUS); else CAND_LAUNCH("k_lz_candidates", ZS_TABLE_LOG_SMALL, 1, ns, dUS); }
        if (nbig) { if (useLong) CAND_LAUNCH("k_lz_candidates_big", ZS_TABLE_LOG_BIG, 2, nbig, dUB); else CAND_LAUNCH("k_lz_candidates_big", ZS_TABLE_LOG_BIG, 1, nbig, dUB); }
        #undef CAND_LAUNCH
        #define WALK_LAUNCH(name, LOOK, REPW, BIG, WLOG, cnt, du) LAUNCH_ON(c, st, name, (ZS_WALK_KERNEL(LOOK, REPW, BIG, WLOG)), dim3(cnt), dim3(ZS_WALK_THREADS(BIG, WLOG)), ZS_WALK_LDS((BIG) ? ZS_UNIT_MAX : ZS_BLOCK_MAX), \
                          (const uint8_t *)dSrc, du, block0, (const uint16_t *)L.dDist.p, (const uint8_t *)L.dDistHi.p, (uint2 *)L.dRecs.p, cap * (ZS_BLOCK_MAX / 4), (uint4 *)L.dRes.p, WLOG, (const uint32_t *)L.dCand.p)
        if (ns) { if (level <= 2) WALK_LAUNCH("k_lz_walk", 4, 8, false, 9, ns, dUS); else if (level == 3) WALK_LAUNCH("k_lz_walk", 4, 4, false, 8, ns, dUS); else WALK_LAUNCH("k_lz_walk", 8, 8, false, 8, ns, dUS); }
        if (nbig) { if (level <= 2) WALK_LAUNCH("k_lz_walk_big", 4, 8, true, 9, nbig, dUB); else if (level == 3) WALK_LAUNCH("k_lz_walk_big", 4, 4, true, 8, nbig, dUB); else WALK_LAUNCH("k_lz_walk_big", 8, 8, true, 8, nbig, dUB); }
        #undef WALK_LAUNCH
        LAUNCH_ON(c, st, "k_lz_stitch", k_lz_stitch, dim3(nb), dim3(256), 0, dB, (const uint2 *)L.dRecs.p, (const uint4 *)L.dRes.p, (ZsSeqRec *)L.dSeqs.p, (ZsRangeHdr *)L.dHdrs.p, walkLog);
        if (c->stopAfterWalk) { chunk0 = chunk1; continue; }
        // sequences first: the literals kernel assembles the frames of one-block chunks as its workgroups finish, and reads the sequence
        // sections then.  (The two side by side on two streams was measured slower: both want the whole LDS.)
        LAUNCH_ON(c, st, "k_encode_sequences", (k_encode_sequences<ZS_SEQ_GROUP>), dim3((nb + ZS_SEQ_GROUP - 1) / ZS_SEQ_GROUP), dim3(64 * ZS_SEQ_GROUP), 0, dB, nb, (const ZsSeqRec *)L.dSeqs.p, (const ZsRangeHdr *)L.dHdrs.p,
                  (uint8_t *)L.dSeqSec.p, (ZsBlockMeta *)L.dMetas.p, c->stopSeq, (uint8_t *)L.dLits.p, (uint8_t *)L.dStreams.p, (uint2 *)L.dDist.p);
        LAUNCH_ON(c, st, "k_encode_literals", k_encode_literals, dim3(nb), dim3(256), 0, (const uint8_t *)dSrc, dB, (const ZsSeqRec *)L.dSeqs.p, (const ZsRangeHdr *)L.dHdrs.p,
                  (uint8_t *)L.dLits.p, (uint8_t *)L.dStreams.p, (uint8_t *)L.dLitSec.p, (ZsBlockMeta *)L.dMetas.p, c->stopLit,
                  (const ZsChunkDesc *)c->dChunks.p, (const uint8_t *)L.dSeqSec.p, (uint8_t *)dDst, dDstSizes);
        if (maxChunkBlocks > 1)                    // chunks of several blocks
            LAUNCH_ON(c, st, "k_assemble_frames", k_assemble_frames, dim3(chunk1 - chunk0), dim3(256), 0, (const uint8_t *)dSrc, (const ZsChunkDesc *)c->dChunks.p,
                      (const ZsBlockDesc *)c->dBlocks.p, (const ZsBlockMeta *)L.dMetas.p, (const uint8_t *)L.dLitSec.p, (const uint8_t *)L.dSeqSec.p, block0, (uint8_t *)dDst, dDstSizes, chunk0);
        chunk0 = chunk1;
    }
    for (int i = 0; !direct && i < nLanes; i++) {
        if (hipEventRecord(c->lanes[i].done, c->lanes[i].stream) != hipSuccess) return ZSMI_error_GENERIC;
        if (hipStreamWaitEvent(c->stream, c->lanes[i].done, 0) != hipSuccess) return ZSMI_error_GENERIC;
    }
    return hipGetLastError() == hipSuccess ? 0 : ZSMI_error_GENERIC;
}

// ---------------------------------------------------------------------------------------------
// decompress
// ---------------------------------------------------------------------------------------------
static int decompressBatchDeviceImpl(zsmi_ctx *c, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                     uint32_t n, void *dDst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dDstSizes,
                                     const void *dDict, uint32_t dictSize)
{
    if (!c) return ZSMI_error_init_missing;
    if (n == 0) return 0;
    if (hipSetDevice(c->device) != hipSuccess) return ZSMI_error_GENERIC;
    // the item list travels through one of two pinned buffers: a call waits only for the copy that last read the buffer it is about to fill
    // (two calls back), not for the device to finish the call before it (round 3 began every call with hipStreamSynchronize)
    const int hb = (int)(c->decodeCalls++ & 1u);
    PinBuf &hItems = c->hItems2[hb];
    if (c->hItemsBusy[hb]) { if (hipEventSynchronize(c->hItemsEv[hb]) != hipSuccess) return ZSMI_error_GENERIC; c->hItemsBusy[hb] = false; }
    if (!hItems.reserve(sizeof(ZsDecItem) * n) || !c->dItems.reserve(sizeof(ZsDecItem) * n)) return ZSMI_error_memory_allocation;
    ZsDecItem *hi = (ZsDecItem *)hItems.p;
    for (uint32_t i = 0; i < n; i++) { hi[i].srcOff = srcOffsets[i]; hi[i].dstOff = dstOffsets[i]; hi[i].srcSize = srcSizes[i]; hi[i].dstCap = dstCaps[i]; }
    if (hipMemcpyAsync(c->dItems.p, hi, sizeof(ZsDecItem) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    if (!c->hItemsEv[hb] && hipEventCreateWithFlags(&c->hItemsEv[hb], hipEventDisableTiming) != hipSuccess) return ZSMI_error_GENERIC;
    if (hipEventRecord(c->hItemsEv[hb], c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    c->hItemsBusy[hb] = true;
    const bool useDict = dDict != nullptr && dictSize != 0;
    const bool fast = c->decodeFast && !useDict;                  // frames that name a dictionary go to the general kernel
    // Scratch.  The general kernel's wavefronts form a POOL with a literal buffer each (ZS_DEC_LITBUF = 128 KiB + 64; as many as the chip holds at once),
    // whatever the call's size.  The fast path keeps per-block scratch in slot = block index * cap + item, sized by the CALL'S LARGEST CAPACITY:
    //   literal bytes a slot:  min(capacity, 128 KiB) + 64        (a block regenerates no more than its item may hold)
    //   sequences a slot:      min(capacity, 128 KiB) / 3 + 64    (a sequence copies >= 3 bytes), 8 bytes each
    //   + Huffman table 4 KiB + sequence tables 2.5 KiB + a descriptor                     -> 32 KiB items: ~125 KiB an item (round 3: 263 KiB whatever the capacity)
    // A block that wants more than its slot holds cannot fit its item's capacity: k_dec_prep leaves it to the general kernel, which reports it.
    // Block slots per item: 1; 2 when some item can hold more than one 64 KiB block; up to ZS_FAST_MAXBLOCKS when at least a quarter of the
    // call's items can hold more than two (a call of large frames; a few large frames among many small ones go to the general kernel, so the
    // small ones do not pay for slots and launches they do not use).  If the scratch does not fit half of the free device memory, first the
    // slots per item go back to 2 and 1, then the items in flight are cut down (never below one sub-batch of 64, never above n).
    uint32_t maxBlocks = 1, bigItems = 0, needBlocks = 1, maxCapBytes = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t nb = (uint32_t)(((uint64_t)dstCaps[i] + ZS_BLOCK_MAX - 1) / ZS_BLOCK_MAX);
        maxCapBytes = std::max(maxCapBytes, dstCaps[i]);
        if (nb > 1) maxBlocks = 2;
        if (nb > 2) { bigItems++; needBlocks = std::max(needBlocks, std::min<uint32_t>(nb, ZS_FAST_MAXBLOCKS)); }
    }
    if (bigItems && (uint64_t)bigItems * 4 >= n) maxBlocks = needBlocks;
    const uint32_t blockCap = std::min<uint32_t>(std::max<uint32_t>(maxCapBytes, 64u), 1u << 17);
    const uint32_t litStride = ((blockCap + 63u) & ~63u) + 64u, litCap = litStride - 64u;
    const uint32_t seqCap = std::min<uint32_t>(ZS_FAST_MAXSEQ, ((blockCap / 3u + 64u) & ~63u));
    uint32_t pool = std::min<uint32_t>(std::max<uint32_t>(n, 1u), c->decodePool);
    pool = ((pool + ZS_DEC_GROUP - 1) / ZS_DEC_GROUP) * ZS_DEC_GROUP;
    uint32_t cap = std::min<uint32_t>(n, c->maxItemsInFlight);
    uint32_t descSlots = std::max(2u, maxBlocks);
    auto perItemBytes = [&](uint32_t mb) { return (size_t)std::max(2u, mb) * sizeof(ZsFastDesc) + (size_t)mb * ((size_t)litStride + ZS_FAST_HUFTAB_BYTES + ZS_FAST_SEQTAB_BYTES + (size_t)seqCap * sizeof(ZsFastSeq) + 2 * sizeof(uint32_t)) + sizeof(uint32_t); };
    size_t perItem = fast ? perItemBytes(maxBlocks) : 0;
    const size_t poolBytes = (size_t)pool * ZS_DEC_LITBUF;
    {
        const size_t have = c->dLitScratch.cap + c->dFastDesc.cap + c->dHufTabs.cap + c->dSeqTabs.cap + c->dSeqOut.cap + c->dPoolLit.cap;     // what the context holds already counts as available
        const size_t need = (size_t)cap * perItem + poolBytes;
        if (need > have) {                                           // (a call the buffers already hold asks the runtime nothing: the one-shot path)
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                const size_t budget = (freeB + have) / 2;
                while (fast && maxBlocks > 1 && (size_t)std::min<uint32_t>(cap, 64u) * perItemBytes(maxBlocks) + poolBytes > budget) { maxBlocks = maxBlocks > 2 ? 2 : 1; }
                descSlots = std::max(2u, maxBlocks); perItem = fast ? perItemBytes(maxBlocks) : 0;
                if (perItem && (size_t)cap * perItem + poolBytes > budget) {
                    const size_t fit = budget > poolBytes ? (budget - poolBytes) / perItem : 0;
                    cap = (uint32_t)std::min<size_t>(cap, std::max<size_t>(std::min<uint32_t>(64u, n), fit));
                }
            }
        }
        // give back what an earlier, larger call left behind: a buffer more than twice (and 256 MiB) beyond this call's need is released first
        auto fitBuf = [&](DevBuf &b, size_t want) { if (b.cap > 2 * want + ((size_t)256 << 20)) b.release(); return b.reserve(want); };
        if (!fitBuf(c->dPoolLit, poolBytes)) return ZSMI_error_memory_allocation;
        if (fast && (!fitBuf(c->dLitScratch, (size_t)cap * maxBlocks * litStride) || !fitBuf(c->dFastDesc, (size_t)cap * descSlots * sizeof(ZsFastDesc)) ||
                     !fitBuf(c->dHufTabs, (size_t)cap * maxBlocks * ZS_FAST_HUFTAB_BYTES) || !fitBuf(c->dSeqTabs, (size_t)cap * maxBlocks * ZS_FAST_SEQTAB_BYTES) ||
                     !fitBuf(c->dSeqOut, (size_t)cap * maxBlocks * seqCap * sizeof(ZsFastSeq)) || !fitBuf(c->dSeqLists, (8 + 2 * (size_t)cap * maxBlocks + cap) * sizeof(uint32_t)))) return ZSMI_error_memory_allocation;
        if (!fast && !c->dSeqLists.reserve(8 * sizeof(uint32_t))) return ZSMI_error_memory_allocation;
        c->lastDecodeScratch = (fast ? (size_t)cap * perItem : 0) + poolBytes;
    }
    for (uint32_t i0 = 0; i0 < n; i0 += cap) {
        const uint32_t cnt = std::min(cap, n - i0);
        const ZsDecItem *dI = (const ZsDecItem *)c->dItems.p + i0;
        const uint32_t *doneFlags = nullptr;
        if (fast) {
            // items that are one frame with one compressed block: entropy decoding lane-parallel across 16 items per wavefront
            // (decode_fast.hip); whatever those kernels do not take or reject is left to the general kernel below
            ZsFastDesc *dD = (ZsFastDesc *)c->dFastDesc.p;
            const uint32_t groups = (cnt + ZS_FAST_GROUP - 1) / ZS_FAST_GROUP;
            uint32_t *dLists = (uint32_t *)c->dSeqLists.p + 4;            // [0], [1]: blocks listed per sequence-table class; then the two lists.  (In front of them: the general kernel's queue)
            if (hipMemsetAsync(c->dSeqLists.p, 0, 6 * sizeof(uint32_t), c->stream) != hipSuccess) return ZSMI_error_GENERIC;
            LAUNCH(c, "k_dec_prep", (k_dec_prep<ZS_DEC_GROUP>), dim3((cnt + ZS_DEC_GROUP - 1) / ZS_DEC_GROUP), dim3(64 * ZS_DEC_GROUP), 0, (const uint8_t *)dSrc, dI, cnt, dD,
                   (uint8_t *)c->dHufTabs.p, (uint8_t *)c->dSeqTabs.p, cap, maxBlocks, dLists, litCap, seqCap);
            {   // every block index of the items in one launch per kernel class (the grid: maxBlocks runs of the items' groups; a wavefront whose items
                // have no such block leaves at once)
                const uint32_t mb = maxBlocks, vcnt = cnt * mb;                 // (item, block) pairs: what a launch's rounds of workgroups count
                // One launch for the four entropy kernels, or one each?  Both are rounds of workgroups of an item's chain each: fused, a CU holds 4 workgroups of
                // the 37 KiB image and a round is 2 CUs' worth of Huffman AND sequence groups (8192 items on 256 CUs, ~0.48 ms); apart, 5 workgroups of 30 KiB and
                // a round of each kind is 20480 items (~0.55 ms, twice).  The fewer round-milliseconds win - measured over 8192 .. 65536 frames of 32 KiB: fused
                // below 20480 items except right at it, at 22528 .. 32768 (4.42 against 4.64 ms at 28672) and 49152; apart at 20480, 36864 .. 45056, 53248 .. 61440.
                bool fuse;
                if (c->decodeFuseSet) fuse = vcnt <= c->decodeFuseBelow;
                else { const uint32_t perF = 32u * c->cus, perS = 80u * c->cus; fuse = ((vcnt + perF - 1) / perF) * 48u < 2u * ((vcnt + perS - 1) / perS) * 55u; }
                if (fuse && mb == 1) {
                    // a round of workgroups or less of one-block items: the four entropy launches as one (k_dec_entropy), the 2.5 KiB sequence class at 4 items a wavefront as below
                    // (items of several blocks - 128 KiB frames: 64 KiB blocks, the 2.5 KiB table class at 16 a wavefront - keep the separate launches: 8192 two-block frames of text
                    //  decoded at 100 GiB/s fused against 135 apart)
                    const uint32_t gH0 = groups * mb, gH1 = ((cnt + 7) / 8) * mb, gS0 = ((cnt + ZS_FAST_SEQGROUP_SMALL - 1) / ZS_FAST_SEQGROUP_SMALL) * mb, gS1 = ((cnt + 3) / 4) * mb;
                    LAUNCH(c, "k_dec_entropy", k_dec_entropy, dim3(gH0 + gH1 + gS0 + gS1), dim3(64), 0, (const uint8_t *)dSrc, dI, cnt, dD, (const uint8_t *)c->dHufTabs.p, (uint8_t *)c->dLitScratch.p,
                           (const uint8_t *)c->dSeqTabs.p, (ZsFastSeq *)c->dSeqOut.p, mb, cap, (const uint32_t *)dLists, litStride, seqCap, gH0, gH1, gS0);
                } else {
                    LAUNCH(c, "k_dec_huffman", (k_dec_huffman<false, ZS_FAST_GROUP>), dim3(groups * mb), dim3(64), 0, (const uint8_t *)dSrc, dI, cnt, dD, (const uint8_t *)c->dHufTabs.p, (uint8_t *)c->dLitScratch.p, mb, cap, litStride);
                    LAUNCH(c, "k_dec_huffman", (k_dec_huffman<true, 8u>), dim3(((cnt + 7) / 8) * mb), dim3(64), 0, (const uint8_t *)dSrc, dI, cnt, dD, (const uint8_t *)c->dHufTabs.p, (uint8_t *)c->dLitScratch.p, mb, cap, litStride);
                    LAUNCH(c, "k_dec_sequences", (k_dec_sequences<false, ZS_FAST_SEQGROUP_SMALL>), dim3(((cnt + ZS_FAST_SEQGROUP_SMALL - 1) / ZS_FAST_SEQGROUP_SMALL) * mb), dim3(64), 0, (const uint8_t *)dSrc, dI, cnt, dD, (const uint8_t *)c->dSeqTabs.p, (ZsFastSeq *)c->dSeqOut.p, mb, cap, (const uint32_t *)dLists, seqCap, 0u, 0xFFFFFFFFu);
                    // the 2.5 KiB table class (blocks of > 2048 sequences: sources, tables, binaries at 32 KiB; the 64 KiB blocks of 128 KiB frames).  How many blocks of a call
                    // are in it only the device knows (k_dec_prep's list), and it decides the shape: 16 items a wavefront when the class holds most of a large call (the
                    // wavefront's instructions are what the kernel costs: 57344 frames of Python sources 4.00 -> 3.78 ms, of a binary table 4.99 -> 4.00), 4 a wavefront
                    // when it is a fraction of it (libzstd's 32 KiB frames: 9 % of the blocks; fewer, emptier wavefronts finish sooner: 3.8 vs 5.2 ms) or the call is small.
                    // Both shapes are launched; each looks at the list's length and leaves at once when the other one serves it.
                    const uint32_t many = c->seqLog9Group == 16 ? 0u : (c->seqLog9Group == 4 ? 0xFFFFFFFFu : ZS_FAST_SEQGROUP_MANY);
                    LAUNCH(c, "k_dec_sequences", (k_dec_sequences<true, ZS_FAST_SEQGROUP>), dim3(((cnt + ZS_FAST_SEQGROUP - 1) / ZS_FAST_SEQGROUP) * mb), dim3(64), 0, (const uint8_t *)dSrc, dI, cnt, dD, (const uint8_t *)c->dSeqTabs.p, (ZsFastSeq *)c->dSeqOut.p, mb, cap, (const uint32_t *)dLists, seqCap, many, 0xFFFFFFFFu);
                    LAUNCH(c, "k_dec_sequences", (k_dec_sequences<true, 4u>), dim3(((cnt + 3) / 4) * mb), dim3(64), 0, (const uint8_t *)dSrc, dI, cnt, dD, (const uint8_t *)c->dSeqTabs.p, (ZsFastSeq *)c->dSeqOut.p, mb, cap, (const uint32_t *)dLists, seqCap, 0u, many);
                }
            }
            // one-block items: 7 wavefronts a SIMD (decode_fast.hip) - but a call that fits ONE round of wavefronts at 8 a SIMD and not at 7 (7169 .. 8192 items on 256 CUs)
            // takes the 8 form: a round of it is ~12 % longer (64 VGPRs: more spills), one round instead of two is not (8192 frames: execute 0.67 -> 0.59 ms; at every
            // other size measured, 4096 .. 57344, the 7 form is as fast or faster)
            const bool oneRoundAt8 = cnt > 7u * 4u * c->cus && cnt <= 8u * 4u * c->cus;
            if (maxBlocks == 1 && (c->execWaves == 8 || (c->execWaves == 0 && oneRoundAt8)))
                LAUNCH(c, "k_dec_execute", (k_dec_execute<4, 8>), dim3((cnt + 3) / 4), dim3(256), 0, (const uint8_t *)dSrc, dI, cnt, dD, (ZsFastSeq *)c->dSeqOut.p,
                       (uint8_t *)c->dLitScratch.p, (uint8_t *)dDst, dDstSizes + i0, cap, descSlots, litStride, seqCap);
            else if (maxBlocks == 1)
                LAUNCH(c, "k_dec_execute", (k_dec_execute<4, 7>), dim3((cnt + 3) / 4), dim3(256), 0, (const uint8_t *)dSrc, dI, cnt, dD, (ZsFastSeq *)c->dSeqOut.p,
                       (uint8_t *)c->dLitScratch.p, (uint8_t *)dDst, dDstSizes + i0, cap, descSlots, litStride, seqCap);
            else
                LAUNCH(c, "k_dec_execute", (k_dec_execute<4, 6>), dim3((cnt + 3) / 4), dim3(256), 0, (const uint8_t *)dSrc, dI, cnt, dD, (ZsFastSeq *)c->dSeqOut.p,
                       (uint8_t *)c->dLitScratch.p, (uint8_t *)dDst, dDstSizes + i0, cap, descSlots, litStride, seqCap);
            LAUNCH(c, "k_dec_checksum", k_dec_checksum, dim3((cnt + 15) / 16), dim3(64), 0, dI, cnt, (const ZsFastDesc *)dD, (const uint8_t *)dDst, dDstSizes + i0);
            doneFlags = &dD->fast;
        }
        // the general kernel: a pool of wavefronts over a queue - of every item, or (behind the fast path) of the list of the items it left
        uint32_t *dQueue = (uint32_t *)c->dSeqLists.p, *dLeftCount = dQueue + 1, *dLeft = nullptr;
        if (!fast && hipMemsetAsync(dQueue, 0, 2 * sizeof(uint32_t), c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        if (doneFlags) {
            dLeft = dQueue + 8 + 2 * (size_t)cap * maxBlocks;
            LAUNCH(c, "k_dec_collect", k_dec_collect, dim3((cnt + 255) / 256), dim3(256), 0, doneFlags, (uint32_t)(sizeof(ZsFastDesc) / sizeof(uint32_t)), cnt, dLeft, dLeftCount);
        }
        const uint32_t poolWgs = std::min<uint32_t>((cnt + ZS_DEC_GROUP - 1) / ZS_DEC_GROUP, pool / ZS_DEC_GROUP);
        if (useDict)
            LAUNCH(c, "k_decode_frames_dict", (k_decode_frames<ZS_DEC_GROUP, true>), dim3(poolWgs), dim3(64 * ZS_DEC_GROUP), 0, (const uint8_t *)dSrc,
                   dI, cnt, (uint8_t *)dDst, dDstSizes + i0, (uint8_t *)c->dPoolLit.p, (const uint32_t *)dLeft, (const uint32_t *)dLeftCount, (const uint8_t *)dDict, dictSize, dQueue);
        else
            LAUNCH(c, "k_decode_frames", (k_decode_frames<ZS_DEC_GROUP, false>), dim3(poolWgs), dim3(64 * ZS_DEC_GROUP), 0, (const uint8_t *)dSrc,
                   dI, cnt, (uint8_t *)dDst, dDstSizes + i0, (uint8_t *)c->dPoolLit.p, (const uint32_t *)dLeft, (const uint32_t *)dLeftCount, (const uint8_t *)nullptr, 0u, dQueue);
    }
    return hipGetLastError() == hipSuccess ? 0 : ZSMI_error_GENERIC;
}

extern "C" int zsmi_decompressBatchDevice(zsmi_ctx *c, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                          uint32_t n, void *dDst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dDstSizes)
{
    return decompressBatchDeviceImpl(c, dSrc, srcOffsets, srcSizes, n, dDst, dstOffsets, dstCaps, dDstSizes, nullptr, 0);
}
// every frame of every item is decoded with the dictionary dDict[0 .. dictSize) (device memory; ZSTD_decompress_usingDict,
// ZStdDecompress.cs:2162): raw content or a formatted dictionary (magic 0xEC30A437)
extern "C" int zsmi_decompressBatchDevice_usingDict(zsmi_ctx *c, const void *dSrc, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                                    uint32_t n, void *dDst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dDstSizes,
                                                    const void *dDict, size_t dictSize)
{
    if (dictSize > 0xFFFFFFFFull) return ZSMI_error_dictionary_corrupted;
    return decompressBatchDeviceImpl(c, dSrc, srcOffsets, srcSizes, n, dDst, dstOffsets, dstCaps, dDstSizes, dDict, (uint32_t)dictSize);
}

// ---------------------------------------------------------------------------------------------
// pack frames
// ---------------------------------------------------------------------------------------------
__global__ void k_pack_offsets(const uint32_t *sizes, uint32_t n, uint64_t *offsets)
{
    // single workgroup exclusive scan over n sizes (errors count as 0 bytes)
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x, per = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(n, tid * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = sizes[i]; s += (v > 0xFFFFFF88u) ? 0 : v; }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { uint64_t run = 0; for (uint32_t t = 0; t < blockDim.x; t++) { const uint64_t v = part[t]; part[t] = run; run += v; } offsets[n] = run; }
    __syncthreads();
    uint64_t run = part[tid];
    for (uint32_t i = lo; i < hi; i++) { offsets[i] = run; const uint32_t v = sizes[i]; run += (v > 0xFFFFFF88u) ? 0 : v; }
}
__global__ void k_pack_copy(const uint8_t *frames, const uint64_t *srcOffsets, const uint32_t *sizes, const uint64_t *packedOffsets, uint8_t *packed)
{
    const uint32_t i = blockIdx.x;
    const uint32_t sz = sizes[i] > 0xFFFFFF88u ? 0 : sizes[i];
    const uint8_t *s = frames + srcOffsets[i]; uint8_t *d = packed + packedOffsets[i];
    zs_block_copy(d, s, sz, threadIdx.x, blockDim.x);
}
extern "C" int zsmi_packFramesDevice(zsmi_ctx *c, const void *dFrames, const uint64_t *dstOffsets, const uint32_t *dSizes,
                                     uint32_t n, void *dPacked, uint64_t *dPackedOffsets)
{
    if (!c) return ZSMI_error_init_missing;
    if (n == 0) return 0;
    if (!c->sSizes.reserve(sizeof(uint64_t) * n)) return ZSMI_error_memory_allocation;
    if (hipMemcpyAsync(c->sSizes.p, dstOffsets, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    LAUNCH(c, "k_pack_offsets", k_pack_offsets, dim3(1), dim3(1024), 0, dSizes, n, dPackedOffsets);
    LAUNCH(c, "k_pack_copy", k_pack_copy, dim3(n), dim3(256), 0, (const uint8_t *)dFrames, (const uint64_t *)c->sSizes.p, dSizes, (const uint64_t *)dPackedOffsets, (uint8_t *)dPacked);
    return hipGetLastError() == hipSuccess ? 0 : ZSMI_error_GENERIC;
}

// ---------------------------------------------------------------------------------------------
// host-buffer forms
// ---------------------------------------------------------------------------------------------
static bool spanOf(const uint64_t *off, const uint32_t *sz, const uint32_t *caps, uint32_t n, uint64_t &lo, uint64_t &hi)
{
    lo = ~0ull; hi = 0;
    for (uint32_t i = 0; i < n; i++) { const uint64_t a = off[i], b = off[i] + (caps ? caps[i] : sz[i]); if (a < lo) lo = a; if (b > hi) hi = b; }
    if (lo == ~0ull) { lo = 0; hi = 0; }
    return true;
}
// Results of a host-buffer call go back to the caller's buffer: items that sit back to back are one transfer; a scattered batch (compressed
// frames in bound-sized slots) is packed on the device, crosses PCIe once into pinned memory and is placed from there (a transfer per item costs
// ~10 us each: 4096 frames took longer to return than to compress)
__global__ void k_pack_items(const uint8_t *base, const uint64_t *offs /* [0..n): where, [n..2n): packed position */, const uint32_t *sizes, uint32_t n, uint8_t *packed)
{
    const uint32_t i = blockIdx.x;
    const uint32_t sz = sizes[i] > 0xFFFFFF88u ? 0 : sizes[i];
    zs_block_copy(packed + offs[n + i], base + offs[i], sz, threadIdx.x, blockDim.x);
}
static int copyBack(zsmi_ctx *c, const uint8_t *dBase, const uint64_t *dof, uint8_t *dst, const uint64_t *dstOffsets, const uint32_t *sizes /* host, synchronised */, uint32_t n)
{
    struct Run { uint64_t host, dev, len; };
    std::vector<Run> runs;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (sizes[i] > 0xFFFFFF88u || sizes[i] == 0) continue;
        total += sizes[i];
        if (!runs.empty() && runs.back().host + runs.back().len == dstOffsets[i] && runs.back().dev + runs.back().len == dof[i]) runs.back().len += sizes[i];
        else runs.push_back({ dstOffsets[i], dof[i], sizes[i] });
    }
    if (runs.size() <= 16) {
        for (const Run &r : runs)
            if (hipMemcpyAsync(dst + r.host, dBase + r.dev, r.len, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
        return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : ZSMI_error_GENERIC;
    }
    if (!c->sPack.reserve(total + 64) || !c->sPackOff.reserve(sizeof(uint64_t) * 2 * n + sizeof(uint32_t) * n) || !c->hPack.reserve(std::max<uint64_t>(total, sizeof(uint64_t) * 2 * n + sizeof(uint32_t) * n))) return ZSMI_error_memory_allocation;
    uint64_t *ho = (uint64_t *)c->hPack.p; uint32_t *hs = (uint32_t *)(ho + 2 * n);
    uint64_t run = 0;
    for (uint32_t i = 0; i < n; i++) { ho[i] = dof[i]; ho[n + i] = run; hs[i] = sizes[i]; run += (sizes[i] > 0xFFFFFF88u) ? 0 : sizes[i]; }
    if (hipMemcpyAsync(c->sPackOff.p, ho, sizeof(uint64_t) * 2 * n + sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return ZSMI_error_GENERIC;      // hPack is reused for the payload below
    LAUNCH(c, "k_pack_items", k_pack_items, dim3(n), dim3(256), 0, dBase, (const uint64_t *)c->sPackOff.p, (const uint32_t *)((const uint64_t *)c->sPackOff.p + 2 * n), n, (uint8_t *)c->sPack.p);
    if (hipMemcpyAsync(c->hPack.p, c->sPack.p, total, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    const uint8_t *hp = (const uint8_t *)c->hPack.p;
    run = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (sizes[i] > 0xFFFFFF88u) continue;
        memcpy(dst + dstOffsets[i], hp + run, sizes[i]); run += sizes[i];
    }
    return 0;
}
extern "C" int zsmi_compressBatchHost(zsmi_ctx *c, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                      uint32_t n, void *dst, const uint64_t *dstOffsets, uint32_t *dstSizes, int level)
{
    if (!c) return ZSMI_error_init_missing;
    if (n == 0) return 0;
    uint64_t slo, shi, dlo, dhi;
    spanOf(srcOffsets, srcSizes, nullptr, n, slo, shi);
    std::vector<uint32_t> bounds(n);
    for (uint32_t i = 0; i < n; i++) bounds[i] = (uint32_t)zsmi_compressBound(srcSizes[i]);
    spanOf(dstOffsets, nullptr, bounds.data(), n, dlo, dhi);
    if (!c->sSrc.reserve(shi - slo + 64) || !c->sDst.reserve(dhi - dlo + 64) || !c->sSizes.reserve(sizeof(uint32_t) * n + sizeof(uint64_t) * n)) return ZSMI_error_memory_allocation;
    std::vector<uint64_t> so(n), dof(n);
    for (uint32_t i = 0; i < n; i++) { so[i] = srcOffsets[i] - slo; dof[i] = dstOffsets[i] - dlo; }
    if (shi > slo && hipMemcpyAsync(c->sSrc.p, (const uint8_t *)src + slo, shi - slo, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    uint32_t *dSizes = (uint32_t *)((uint8_t *)c->sSizes.p + sizeof(uint64_t) * n);
    const int rc = zsmi_compressBatchDevice(c, c->sSrc.p, so.data(), srcSizes, n, c->sDst.p, dof.data(), dSizes, level);
    if (rc) return rc;
    if (hipMemcpyAsync(dstSizes, dSizes, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    return copyBack(c, (const uint8_t *)c->sDst.p, dof.data(), (uint8_t *)dst, dstOffsets, dstSizes, n);
}
static int decompressBatchHostImpl(zsmi_ctx *c, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                   uint32_t n, void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                                   const void *dict, size_t dictSize)
{
    if (!c) return ZSMI_error_init_missing;
    if (n == 0) return 0;
    if (dictSize > 0xFFFFFFFFull) return ZSMI_error_dictionary_corrupted;
    const bool useDict = dict != nullptr && dictSize != 0;
    if (useDict) {
        if (!c->sDict.reserve(dictSize + 64)) return ZSMI_error_memory_allocation;
        if (hipMemcpyAsync(c->sDict.p, dict, dictSize, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    }
    uint64_t slo, shi, dlo, dhi;
    spanOf(srcOffsets, srcSizes, nullptr, n, slo, shi);
    spanOf(dstOffsets, nullptr, dstCaps, n, dlo, dhi);
    if (!c->sSrc.reserve(shi - slo + 64) || !c->sDst.reserve(dhi - dlo + 64) || !c->sSizes.reserve(sizeof(uint32_t) * n)) return ZSMI_error_memory_allocation;
    std::vector<uint64_t> so(n), dof(n);
    for (uint32_t i = 0; i < n; i++) { so[i] = srcOffsets[i] - slo; dof[i] = dstOffsets[i] - dlo; }
    if (shi > slo && hipMemcpyAsync(c->sSrc.p, (const uint8_t *)src + slo, shi - slo, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    const int rc = decompressBatchDeviceImpl(c, c->sSrc.p, so.data(), srcSizes, n, c->sDst.p, dof.data(), dstCaps, (uint32_t *)c->sSizes.p,
                                             useDict ? c->sDict.p : nullptr, useDict ? (uint32_t)dictSize : 0u);
    if (rc) return rc;
    if (hipMemcpyAsync(dstSizes, c->sSizes.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return ZSMI_error_GENERIC;
    return copyBack(c, (const uint8_t *)c->sDst.p, dof.data(), (uint8_t *)dst, dstOffsets, dstSizes, n);
}

extern "C" int zsmi_decompressBatchHost(zsmi_ctx *c, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                        uint32_t n, void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes)
{
    return decompressBatchHostImpl(c, src, srcOffsets, srcSizes, n, dst, dstOffsets, dstCaps, dstSizes, nullptr, 0);
}
extern "C" int zsmi_decompressBatchHost_usingDict(zsmi_ctx *c, const void *src, const uint64_t *srcOffsets, const uint32_t *srcSizes,
                                                  uint32_t n, void *dst, const uint64_t *dstOffsets, const uint32_t *dstCaps, uint32_t *dstSizes,
                                                  const void *dict, size_t dictSize)
{
    return decompressBatchHostImpl(c, src, srcOffsets, srcSizes, n, dst, dstOffsets, dstCaps, dstSizes, dict, dictSize);
}

// ---------------------------------------------------------------------------------------------
// one-shot calls (the reference's public API shape).  The reference's static calls are re-entrant: a fresh DCtx per call
// (ZStdDecompress.cs:2174-2180).  Here a call borrows a context from a per-device pool (created on demand, at most
// ZSMI_ONESHOT_CONTEXTS = 8 per device; further callers wait), so concurrent callers run side by side instead of queueing behind
// one mutex, and a context's device buffers are reused from call to call.  The pool is emptied when the library is unloaded.
// ---------------------------------------------------------------------------------------------
namespace {
struct OneShotPool {
    static const int kMax = 8;
    std::mutex mu;
    std::condition_variable cv;
    struct PerDevice { std::vector<zsmi_ctx *> idle; int created = 0; };
    std::map<int, PerDevice> dev;
    zsmi_ctx *acquire()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return nullptr;
        std::unique_lock<std::mutex> lk(mu);
        PerDevice &p = dev[d];
        for (;;) {
            if (!p.idle.empty()) { zsmi_ctx *c = p.idle.back(); p.idle.pop_back(); return c; }
            if (p.created < kMax) {
                p.created++;
                lk.unlock();
                zsmi_ctx *c = zsmi_createCtx(d, nullptr);
                if (!c) { lk.lock(); p.created--; cv.notify_one(); }
                return c;
            }
            cv.wait(lk);
        }
    }
    void release(zsmi_ctx *c)
    {
        { std::lock_guard<std::mutex> lk(mu); dev[c->device].idle.push_back(c); }
        cv.notify_one();
    }
    // No destructor work: at process exit or dlclose the order against the HIP runtime's own teardown is not defined, and HIP calls from an
    // exit-time destructor are a known source of aborts.  The contexts are left to the process; an embedder that unloads the library
    // calls zsmi_shutdown() first.
    void drain()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : dev) { for (zsmi_ctx *c : kv.second.idle) { zsmi_freeCtx(c); kv.second.created--; } kv.second.idle.clear(); }
    }
};
OneShotPool &g_pool = *new OneShotPool();                  // (never destroyed: see above)
struct Borrowed {
    zsmi_ctx *c;
    Borrowed() : c(g_pool.acquire()) {}
    ~Borrowed() { if (c) g_pool.release(c); }
};
}

extern "C" void zsmi_shutdown(void) { g_pool.drain(); }
extern "C" size_t zsmi_decodeScratchBytes(zsmi_ctx *c)
{
    if (!c) return 0;
    return c->dPoolLit.cap + c->dLitScratch.cap + c->dFastDesc.cap + c->dHufTabs.cap + c->dSeqTabs.cap + c->dSeqOut.cap + c->dSeqLists.cap;
}

extern "C" size_t zsmi_compress(void *dst, size_t dstCapacity, const void *src, size_t srcSize, int level)
{
    if (srcSize > 0xFFFFFFFFull) return ZSMI_ERR(ZSMI_error_srcSize_wrong);
    Borrowed b; zsmi_ctx *c = b.c;
    if (!c) return ZSMI_ERR(ZSMI_error_GENERIC);
    const size_t bound = zsmi_compressBound(srcSize);
    std::vector<uint8_t> tmp;
    uint8_t *out = (uint8_t *)dst;
    if (dstCapacity < bound) { tmp.resize(bound); out = tmp.data(); }     // compress into a bound-sized buffer, then check the fit
    const uint64_t so = 0, dof = 0; const uint32_t ss = (uint32_t)srcSize; uint32_t ds = 0;
    const int rc = zsmi_compressBatchHost(c, src, &so, &ss, 1, out, &dof, &ds, level);
    if (rc) return ZSMI_ERR(rc);
    if (ds > 0xFFFFFF88u) return ZSMI_ERR(0u - ds);
    if (ds > dstCapacity) return ZSMI_ERR(ZSMI_error_dstSize_tooSmall);
    if (out != dst) memcpy(dst, out, ds);
    return ds;
}
extern "C" size_t zsmi_decompress(void *dst, size_t dstCapacity, const void *src, size_t srcSize)
{
    if (srcSize > 0xFFFFFFFFull) return ZSMI_ERR(ZSMI_error_srcSize_wrong);
    Borrowed b; zsmi_ctx *c = b.c;
    if (!c) return ZSMI_ERR(ZSMI_error_GENERIC);
    const uint64_t so = 0, dof = 0; const uint32_t ss = (uint32_t)srcSize; uint32_t ds = 0;
    const uint32_t cap = (uint32_t)std::min<size_t>(dstCapacity, 0xFFFFFF00u);
    const int rc = zsmi_decompressBatchHost(c, src, &so, &ss, 1, dst, &dof, &cap, &ds);
    if (rc) return ZSMI_ERR(rc);
    if (ds > 0xFFFFFF88u) return ZSMI_ERR(0u - ds);
    return ds;
}

extern "C" size_t zsmi_decompress_usingDict(void *dst, size_t dstCapacity, const void *src, size_t srcSize, const void *dict, size_t dictSize)
{
    if (srcSize > 0xFFFFFFFFull) return ZSMI_ERR(ZSMI_error_srcSize_wrong);
    Borrowed b; zsmi_ctx *c = b.c;
    if (!c) return ZSMI_ERR(ZSMI_error_GENERIC);
    const uint64_t so = 0, dof = 0; const uint32_t ss = (uint32_t)srcSize; uint32_t ds = 0;
    const uint32_t cap = (uint32_t)std::min<size_t>(dstCapacity, 0xFFFFFF00u);
    const int rc = zsmi_decompressBatchHost_usingDict(c, src, &so, &ss, 1, dst, &dof, &cap, &ds, dict, dictSize);
    if (rc) return ZSMI_ERR(rc);
    if (ds > 0xFFFFFF88u) return ZSMI_ERR(0u - ds);
    return ds;
}

#ifdef ZSMI_DEBUG_HOOKS
// ---- test hook (not in include/zsmi.h): copy a scratch buffer of the last compress sub-batch to the host.
//      which: 0 dist (u16 x 65536 per block), 1 sequences (ZsSeqRec x 8 x 2048 per block), 2 range headers, 3 block results ----
// words of a ZsFastDesc, and the word index of its fields `fast`, `why`, `nbSeq`, `litType`, `hufLog` (tools/fastpath_check.py, tools/dec_why.py)
extern "C" void zsmi_dbg_descLayout(uint32_t out[6])
{
    out[0] = (uint32_t)(sizeof(ZsFastDesc) / 4); out[1] = (uint32_t)(offsetof(ZsFastDesc, fast) / 4); out[2] = (uint32_t)(offsetof(ZsFastDesc, why) / 4);
    out[3] = (uint32_t)(offsetof(ZsFastDesc, nbSeq) / 4); out[4] = (uint32_t)(offsetof(ZsFastDesc, litType) / 4); out[5] = (uint32_t)(offsetof(ZsFastDesc, hufLog) / 4);
}
extern "C" int zsmi_dbg_copyScratch(zsmi_ctx *c, int which, void *hostDst, size_t bytes)
{
    if (!c) return -1;
    (void)hipStreamSynchronize(c->stream);
    zsmi_ctx::Scratch &L0 = c->lanes[0];
    DevBuf *b = which == 0 ? &L0.dDist : which == 1 ? &L0.dSeqs : which == 2 ? &L0.dHdrs : which == 4 ? &L0.dDistHi : which == 7 ? &L0.dRecs : which == 8 ? &L0.dRes : which == 5 ? &c->dLitScratch : which == 9 ? &c->dHufTabs : which == 10 ? &c->dFastDesc : &L0.dMetas;
    if (bytes > b->cap) return -2;
    return hipMemcpy(hostDst, b->p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
