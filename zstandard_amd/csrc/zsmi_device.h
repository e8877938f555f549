// Shared definitions for the HIP kernels of the MI355X Zstandard block codec (gfx950 only).
// Format constants follow the reference decoder: csharp/src/ZStdInternal.cs:109-198, ZStd.cs:386-416.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZS_BLOCK_MAX   65536u     // bytes per block; block positions fit 16 bits
#define ZS_UNIT_MAX    131072u    // bytes per LZ unit (match window): two consecutive blocks of a chunk
#define ZS_RANGE_LOG   13
#define ZS_RANGE_SIZE  (1u << ZS_RANGE_LOG)   // hash-table ranges (8 KiB): 8 per block, 16 per full unit
#define ZS_HASH_LOG    12         // slots per range table
#define ZS_WALK_LOG    10
#define ZS_WALK_SIZE   (1u << ZS_WALK_LOG)   // the walk cuts the block in ranges of 1 KiB
#define ZS_WALK_RANGES 64u        // per block
#define ZS_MINMATCH    5u         // shortest match kept (candidates are found by their first 4 bytes)
#define ZS_WINDOW      64u        // positions looked at per walk step (one wavefront)
#define ZS_FCAP        8u         // forward bytes compared when scoring a candidate
#define ZS_BCAP        8u         // backward bytes compared when scoring a candidate
#define ZS_LCAP        16u        // forward bytes a walker lane compares in its one round of loads; longer matches are extended
#define ZS_SEQ_PER_RANGE 256u     // record slots per walk range (1024 / 4)
#define ZS_HUF_MAXBITS 11u

// one 64 KiB block of one chunk
struct ZsBlockDesc {
    uint64_t srcOff;      // byte offset of the block in the source arena
    uint32_t size;        // 0..65536
    uint32_t chunk;       // owning chunk (frame)
    uint32_t firstInChunk;
    uint32_t lastInChunk;
};

// one LZ unit: blocks firstBlock, firstBlock + 1 (if size > 64 KiB) of one chunk
struct ZsUnitDesc {
    uint64_t srcOff;      // byte offset of the unit in the source arena
    uint32_t size;        // 1..131072
    uint32_t firstBlock;  // index of its first block in the call's block list
};

// one sequence as the walk kernel leaves it (per range) / as the encode kernel consumes it
struct ZsSeqRec { uint16_t ll, ml, off, flags; };   // off: low 16 bits of the offset, ml bit 13: its bit 16; flags: block position of the match start; ml bits 14-15: repcode (encode kernel)

struct ZsRangeHdr { uint32_t nseq, trailing, litSum, pad; };   // litSum: literal bytes in front of the range's matches

// per-block result of the encode kernel
struct ZsBlockResult { uint32_t payloadSize; uint32_t type; /* 0 raw, 1 rle, 2 compressed */ uint32_t rleByte; uint32_t pad; };

// unaligned loads.  memcpy keeps the alignment-1 fact visible to the compiler: a cast to an over-aligned
// pointer lets it turn a wave-uniform address into a scalar load, which drops the low address bits.
__device__ __forceinline__ uint32_t zs_load32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t zs_load64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
// unaligned stores (global memory takes them at any byte address)
__device__ __forceinline__ void zs_store16(uint8_t *p, uint16_t v) { __builtin_memcpy(p, &v, 2); }
__device__ __forceinline__ void zs_store32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void zs_store64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ uint32_t zs_hash4(uint32_t v, int hashLog) { return (v * 2654435761u) >> (32 - hashLog); }
__device__ __forceinline__ uint32_t zs_highbit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
__device__ __forceinline__ int zs_lane() { return (int)(threadIdx.x & 63u); }
