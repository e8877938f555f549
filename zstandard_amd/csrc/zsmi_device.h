// Shared definitions for the HIP kernels of the MI355X Zstandard block codec (gfx950 only).
// Format constants follow the reference decoder: csharp/src/ZStdInternal.cs:109-198, ZStd.cs:386-416.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZS_BLOCK_MAX   65536u     // bytes per block; block positions fit 16 bits
#define ZS_UNIT_MAX    131072u    // bytes per LZ unit (match window): two consecutive blocks of a chunk
#define ZS_TABLE_LOG_SMALL 13     // slots of each candidate table, units <= 64 KiB
#define ZS_TABLE_LOG_BIG   14     // units <= 128 KiB
#define ZS_WALK_LOG_MIN 8         // the walk cuts a block in ranges of 256 bytes (512 at levels <= 2: the launch says which)
#define ZS_OUT_LOG     10         // the walk kernel hands the sequences on in output ranges of 1 KiB
#define ZS_WALK_RANGES 64u        // output ranges per block
#define ZS_RES_PER_BLOCK 256u     // walk-range results per block (ranges of >= 256 bytes)
#ifndef ZS_CROSS_MAX
#define ZS_CROSS_MAX   1024u      // a match may pass its walk range's end by this much (never the block's end)
#endif
#define ZS_MATCHLESS_SHIFT 11      // a unit with fewer than n >> 11 candidate positions is not parsed (oracle: MATCHLESS_SHIFT)
#define ZS_MINMATCH    5u         // shortest match kept
#define ZS_REPMIN      4u         // shortest match at one of the walker's two recent offsets
#define ZS_WINDOW      32u        // positions looked at per walk step
#define ZS_FCAP        8u         // forward bytes compared when scoring a candidate (the match taken is measured to its end)
#define ZS_BCAP        4u         // backward bytes compared when scoring a candidate
#define ZS_SEQ_PER_RANGE 256u     // record slots per output range: its matches start inside it and are >= 4 bytes long
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

// one sequence as the walk kernel leaves it (per range) / as the entropy kernels consume it: two 32-bit words
//   x: bits 0-10 literals in front of it inside its output range's territory (<= 1024), bits 11-27 match length (<= 65536: matches found
//      piecewise are joined), bit 28 bit 16 of the offset, bits 29-30 recent-offset code (written by the sequences kernel)
//   y: bits 0-15 low 16 bits of the offset, bits 16-31 block position of the match start
struct ZsSeqRec { uint32_t x, y; };
__device__ __forceinline__ uint32_t zs_rec_ll(uint32_t x) { return x & 0x7FFu; }
__device__ __forceinline__ uint32_t zs_rec_ml(uint32_t x) { return (x >> 11) & 0x1FFFFu; }
__device__ __forceinline__ uint32_t zs_rec_rep(uint32_t x) { return (x >> 29) & 3u; }
__device__ __forceinline__ uint32_t zs_rec_with_rep(uint32_t x, uint32_t rep) { return (x & 0x1FFFFFFFu) | (rep << 29); }
__device__ __forceinline__ uint32_t zs_rec_off(uint32_t x, uint32_t y) { return (y & 0xFFFFu) | (((x >> 28) & 1u) << 16); }
__device__ __forceinline__ uint32_t zs_rec_pos(uint32_t y) { return y >> 16; }
__device__ __forceinline__ uint32_t zs_rec_x(uint32_t ll, uint32_t ml, uint32_t off) { return ll | (ml << 11) | ((off >> 16) << 28); }
__device__ __forceinline__ uint32_t zs_rec_y(uint32_t off, uint32_t pos) { return (off & 0xFFFFu) | (pos << 16); }

// a walk range after the stitch: its records [first, first + nseq) count; litSum: literal bytes in front of those matches inside the
// range's territory; trailing: literal bytes behind the last of them (the whole territory if there is none)
struct ZsRangeHdr { uint32_t nseq, trailing, litSum, first; };

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
__device__ __forceinline__ uint32_t zs_highbit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
__device__ __forceinline__ int zs_lane() { return (int)(threadIdx.x & 63u); }
