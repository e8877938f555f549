// Fast decode path for the shapes the batch codec itself produces and BASELINE config 4 names: an item that is exactly
// one frame with one or two compressed blocks (chunks of <= 64 KiB / <= 128 KiB) - up to ZS_FAST_MAXBLOCKS blocks in a call
// that is mostly large frames -, with or without a content checksum.
// Everything per block (descriptor, tables, literals, decoded sequences) lives in slot blk * cap + item: the Huffman and
// sequences kernels are launched once per block index; the execute kernel walks an item's blocks in order, carrying the
// output position and the recent offsets (ZStdDecompress.cs:1596).  A block that repeats tables (literals type 3, sequence mode 3:
// ZStdDecompress.cs:696-697, 1062-1064) gets a copy of them in its own slot from k_dec_prep.  The serial entropy decoders run lane-parallel ACROSS frames:
//
//   k_dec_prep      one wavefront per item : headers, Huffman table (two-level; flat 2^11 for wide alphabets), sequence tables (all lanes build,
//                                            one lane parses the counts) -> tables + a descriptor in global memory; decides fast / general
//   k_dec_huffman   one wavefront per 16 items: lane 4g + k decodes Huffman stream k of item g (64 streams at once); flat-table class: 8 items
//   k_dec_sequences one wavefront per 16 items: four lanes an item, one per FSE state (the three states of a sequence side by side)
//   k_dec_execute   one wavefront per item : extra bits, recent offsets, positions, checks per tile of 64 sequences; the block's literals spread into
//                                            the output; then its matches, in-tile sources redirected, the rest level by level
//
// Anything unusual -- another frame shape, a table the fast kernels do not hold (Huffman log 12, > 16384 sequences),
// a stream that does not end exactly, a failed check while executing -- clears the item's fast flag, and the general
// kernel (k_decode_frames, same reference semantics and error codes) decodes it afterwards from scratch.  So the fast
// path only ever has to be right about VALID frames; the error behaviour stays that of decode_kernels.hip.
//
// Reference functions restated: see decode_kernels.hip (same helpers: readHufTable, seqHeaders, execTile, BitC readers).
#include "zsmi_device.h"

#define ZS_FAST_HUFLOG   11u                      // Huffman tables the fast kernel holds: 2^11 entries per item
#define ZS_FAST_MAXSEQ   16384u                   // sequences per block the fast path buffers (8 bytes each)
#define ZS_FAST_MAXBLOCKS 16u                      // block slots per item a call may reserve (frames of up to 1 MiB; 1 or 2 unless the call is mostly large frames)
#ifndef ZS_FAST_HUFWIN
#define ZS_FAST_HUFWIN   128u                     // bytes of each Huffman stream staged in LDS at a time
#endif
#ifndef ZS_FAST_SEQWIN
#define ZS_FAST_SEQWIN   256u
#endif                     // bytes of each sequence bitstream staged in LDS at a time
#define ZS_FAST_GROUP    16u                      // items per wavefront of the Huffman kernel (4 lanes each)
#ifndef ZS_FAST_SEQGROUP
#define ZS_FAST_SEQGROUP 16u                      // items per wavefront of the sequences kernel (2.5 KiB of tables each); 16384 two-block frames of 128 KiB: 2: 10.4 ms, 4: 8.5, 8: 8.2, 16: 7.0 (round 1, with the carried bit container: 4 was best)
#endif
#define ZS_FAST_SEQGROUP_MANY 12288u                // items in a launch from which the 2.5 KiB class takes ZS_FAST_SEQGROUP items a wavefront (below: 4)
#ifndef ZS_FAST_SEQGROUP_SMALL
#define ZS_FAST_SEQGROUP_SMALL 16u                // the same for items with tables of <= 2^8 cells (1.5 KiB): measured 4: 4.36 ms, 8: 3.70, 16: 3.16 per 57344 items
#endif

struct ZsFastDesc {                               // per item, global memory, written by k_dec_prep
    uint32_t fast;                                // 1: the fast kernels own this item; any of them may clear it
    uint32_t litType;                             // 0 raw (bytes in the source), 1 RLE, 2 Huffman
    uint32_t litSize;
    uint32_t litSrc;                              // raw: offset of the literals in the item's source; RLE: the byte
    uint32_t hufLog, nStreams;
    uint32_t sOff[4], sLen[4], sCnt[4], sOut[4];  // Huffman streams: source offset, bytes, symbols, offset in the literal scratch
    uint32_t nbSeq, seqOff, seqSize;              // sequence bitstream inside the item's source
    uint32_t llLog, ofLog, mlLog;
    uint32_t contentSize, hasContentSize;
    uint32_t hasChecksum, checksum;               // content checksum (low 32 bits of XXH64, ZStdDecompress.cs:2078-2082): checked by k_dec_checksum
    uint32_t hufFlat;                             // 1: the item's Huffman table is the flat one of 2^11 entries (more long-code prefixes than the two-level table has sub-tables)
    uint32_t why;                                 // which fast kernel handed the item to the general one (1 Huffman stream, 2 sequence stream, 3.. execute: tools/dec_why.py)
};
#define ZS_FAST_HUFTAB_BYTES (2u << ZS_FAST_HUFLOG)                       // uint16 entries
#define ZS_FAST_SEQTAB_BYTES ((512u + 256u + 512u) * 2u)                  // LL, OF, ML cells, 2 bytes each
// what the sequences kernel leaves per sequence, 8 bytes: where its extra bits start in the bitstream (bit position, 20 bits)
// and its three codes (LL 6 bits at 20, ML 6 bits at 26, OF 5 bits at 32).  The execute kernel turns that into lengths and
// offsets, 64 sequences at a time on 64 lanes; only the FSE state chain stays serial.
typedef uint64_t ZsFastSeq;
__device__ __forceinline__ ZsFastSeq zs_fastseq(uint32_t bitPos, uint32_t symLL, uint32_t symML, uint32_t symOF)
{ return (uint64_t)(bitPos | (symLL << 20) | (symML << 26)) | ((uint64_t)symOF << 32); }
// (the 16-bit cell format: zs_fastcell in decode_kernels.hip.)  The extra bits of a code come from the code by arithmetic
// (LL_bits / ML_bits, ZStdInternal.cs:158,173).
__device__ __forceinline__ void zs_fastcell_open(uint32_t c, uint32_t &next, uint32_t &nb, uint32_t &sym)
{
    const uint32_t p = c >> 6, hb = 31u - (uint32_t)__builtin_clz(p | 1u);
    nb = 9u - hb; next = (p ^ (1u << hb)) << nb; sym = c & 63u;
}
__device__ __forceinline__ uint32_t zs_llExtraBits(uint32_t s) { return s < 16 ? 0u : (s <= 19 ? 1u : (s <= 21 ? 2u : (s <= 23 ? 3u : (s == 24 ? 4u : s - 19)))); }
__device__ __forceinline__ uint32_t zs_mlExtraBits(uint32_t s) { return s < 32 ? 0u : (s <= 35 ? 1u : (s <= 37 ? 2u : (s <= 39 ? 3u : (s <= 41 ? 4u : (s == 42 ? 5u : s - 36))))); }

// ---------------------------------------------------------------------------------------------------------------------
// k_dec_prep
// ---------------------------------------------------------------------------------------------------------------------
#ifndef ZS_PREP_MINWG
#define ZS_PREP_MINWG 4                 // wavefronts per SIMD the prep kernel is compiled for (128 VGPRs; per 57344 frames: 2: 0.96 ms, 3: 0.97, 4: 0.97, 5: 1.02, 6: 1.04, 8: 1.10 with ~450 spilled registers; the kernel is bound by the instructions it issues)
#endif
template <int F>
__global__ void __launch_bounds__(64 * F, ZS_PREP_MINWG)
k_dec_prep(const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems,
           ZsFastDesc *__restrict__ descs, uint8_t *__restrict__ hufTabs, uint8_t *__restrict__ seqTabs, uint32_t cap, uint32_t maxBlocks, uint32_t *__restrict__ seqLists,
           uint32_t litCap, uint32_t seqCap)
{
    // (litCap, seqCap: literal bytes / sequences a block slot of this call holds - sized by the call's largest capacity, zsmi_api.hip; a block that
    //  wants more cannot fit its item's capacity and is left to the general kernel, which says why)
    // (maxBlocks: block slots the call reserved per item, 1, 2 or up to ZS_FAST_MAXBLOCKS - frames of more compressed blocks are left to the general kernel;
    //  descriptors always have both)
    // the general decoder's LDS image without its Huffman table and with one sequence table instead of three (4.4 of 15.5 KiB)
    __shared__ __attribute__((aligned(16))) unsigned char LSraw[F][(ZS_DLDS_PREP + 15) & ~15u];
    const uint32_t item = blockIdx.x * F + (threadIdx.x >> 6);
    if (item >= nItems) return;
    DLds &L = *reinterpret_cast<DLds *>(LSraw[threadIdx.x >> 6]);
    const ZsDecItem it = items[item];
    const uint32_t lane = (uint32_t)zs_lane();
    const uint8_t *src = srcAll + it.srcOff;
    const uint32_t srcSize = it.srcSize;
    // descriptor fields go to global memory as they become known (lane 0), the fast flags last: nothing of a descriptor is kept in
    // registers (27 of them otherwise: the kernel's occupancy)
    #define DSET(field, value) do { if (lane == 0) dp->field = (value); } while (0)
    uint32_t prepWhy = 0;                                      // source line of the test that left the fast path (ZsFastDesc.why = 1000 + line: tools/dec_why.py)
    #define ZS_PREP_WHY(line) do { prepWhy = 1000u + (uint32_t)(line); } while (0)
    if (lane < 36) L.llTab[lane] = d_LL_base[lane] | ((uint32_t)d_LL_bits[lane] << 24);
    if (lane < 53) L.mlTab[lane] = d_ML_base[lane] | ((uint32_t)d_ML_bits[lane] << 24);
    wave_sync();
#ifdef ZS_PREP_PROFILE
    if (lane == 0) { for (int k = 0; k < 12; k++) L.pp[k] = 0; L.ppMark = __builtin_amdgcn_s_memtime(); }
    const unsigned long long ppStart = __builtin_amdgcn_s_memtime();
#endif
    bool ok = false; uint32_t nBlocks = 0;
    do {
        // ---- frame header (:389-499): one frame, no dictionary ----
        if (srcSize < 5 + 1 + 3 || rd32(src) != 0xFD2FB528u) { ZS_PREP_WHY(__LINE__); break; }
        const uint32_t fhd = src[4];
        const uint32_t dictIDCode = fhd & 3, checksumFlag = (fhd >> 2) & 1, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
        if (dictIDCode || (fhd & 0x08)) { ZS_PREP_WHY(__LINE__); break; }
        const uint32_t tail = checksumFlag ? 4u : 0u;              // the checksum behind the last block
        const uint32_t fcsSize = fcsID == 0 ? 0 : (fcsID == 1 ? 2 : (fcsID == 2 ? 4 : 8));
        const uint32_t fhs = 5 + !singleSegment + fcsSize + (singleSegment && !fcsID);
        if (srcSize < fhs + 3 + tail) { ZS_PREP_WHY(__LINE__); break; }
        uint32_t pos = 5;
        if (!singleSegment) { const uint32_t wl = src[pos++]; if ((wl >> 3) + 10 > 30) { ZS_PREP_WHY(__LINE__); break; } }
        uint64_t fcs = ~0ull;
        if (fcsID == 0) { if (singleSegment) fcs = src[pos]; } else if (fcsID == 1) fcs = rd16(src + pos) + 256; else if (fcsID == 2) fcs = rd32(src + pos); else fcs = zs_load64(src + pos);
        if (fcs != ~0ull && fcs > 0xFFFFFFFFull) { ZS_PREP_WHY(__LINE__); break; }
        const uint32_t hasContentSize = fcs != ~0ull, contentSize = (uint32_t)fcs;
        uint32_t b0 = fhs;                                          // offset of the next block header in the item
        bool fail = false;
        // tables a later block may repeat (literals type 3, sequence mode 3: ZStdDecompress.cs:696-697, 1062-1064): the slot that holds the frame's
        // current Huffman table, the slot of the last block that had sequences (it holds all three sequence tables, built or copied) and their logs.
        // A repeating block gets a COPY in its own slot, so the kernels behind this one never look at another slot.
        // (kept in LDS words, not registers: the loop body is the whole table code inlined, and eight more values alive across it brought the spills back.
        //  misc[11]: block + 1 of the Huffman table's slot; misc[12]: block + 1 of the last block with sequences; misc[13]: Huffman log | flat << 8;
        //  misc[14]: raw / RLE blocks so far; misc[8..10]: the sequence tables' logs, left alone by a block without sequences)
        if (lane < 8) L.misc[8 + lane] = 0;
        wave_sync();
        #pragma unroll 1
        for (uint32_t blk = 0; blk < maxBlocks && !fail; blk++) {
            fail = true;
            const size_t slot = (size_t)blk * cap + item;
            ZsFastDesc *dp = descs + slot;
            if (blk == 0) { DSET(why, 0u); DSET(hasContentSize, hasContentSize); DSET(contentSize, contentSize); DSET(hasChecksum, checksumFlag); DSET(checksum, checksumFlag ? rd32(src + srcSize - 4) : 0u); }
            // ---- a block (:646-659): compressed; the last one fills the rest of the item ----
            if ((uint64_t)b0 + 3 + tail > srcSize) { ZS_PREP_WHY(__LINE__); break; }
            const uint32_t bh = rd24(src + b0);
            const uint32_t lastBlock = bh & 1, btype = (bh >> 1) & 3, cSize = bh >> 3;
            if (btype == 3) { ZS_PREP_WHY(__LINE__); break; }
            if (btype != 2) {
                // a raw or RLE block among the compressed ones (:2043-2056): to the kernels behind this one a block of nothing but literals - raw
                // literals at the block's bytes, or RLE literals of its byte - and no sequences; the entropy tables a later block may repeat stay
                const uint32_t csz = btype == 1 ? 1u : cSize;
                if (cSize > (1u << 17) || (btype == 1 && cSize > litCap)) { ZS_PREP_WHY(__LINE__); break; }       // (an RLE block is spread through the slot's literal buffer)
                if ((uint64_t)b0 + 3 + csz + tail > srcSize) { ZS_PREP_WHY(__LINE__); break; }
                if (lastBlock && (uint64_t)b0 + 3 + csz + tail != srcSize) { ZS_PREP_WHY(__LINE__); break; }
                if (!lastBlock && blk + 1 == maxBlocks) { ZS_PREP_WHY(__LINE__); break; }
                b0 += 3;
                DSET(litType, btype == 0 ? 0u : 1u); DSET(litSrc, btype == 0 ? b0 : (uint32_t)src[b0]); DSET(litSize, cSize); DSET(nStreams, 0u);
                DSET(nbSeq, 0u); DSET(seqOff, 0u); DSET(seqSize, 0u); DSET(llLog, 0u); DSET(ofLog, 0u); DSET(mlLog, 0u); DSET(hufFlat, 0u);
                DSET(fast, 1u);
                nBlocks = blk + 1; if (lane == 0) L.misc[14] = L.misc[14] + 1;
                b0 += csz;
                fail = false;
                if (lastBlock) break;
                continue;
            }
            if (cSize >= (1u << 17) || cSize < 3) { ZS_PREP_WHY(__LINE__); break; }
            if ((uint64_t)b0 + 3 + cSize + tail > srcSize) { ZS_PREP_WHY(__LINE__); break; }
            if (lastBlock && (uint64_t)b0 + 3 + cSize + tail != srcSize) { ZS_PREP_WHY(__LINE__); break; }
            if (!lastBlock && blk + 1 == maxBlocks) { ZS_PREP_WHY(__LINE__); break; }           // more blocks than slots: general kernel
            b0 += 3;                                                // block payload offset in the item
            const uint8_t *bs = src + b0;
            // ---- literals section header (:683-821) ----
            uint32_t litCSizeTot;
            {
                const uint32_t type = bs[0] & 3, lhl = (bs[0] >> 2) & 3;
                if (type == 3 && L.misc[11] == 0) { ZS_PREP_WHY(__LINE__); break; }            // a repeated Huffman table without one before it: the general kernel says what is wrong
                if (type >= 2) {
                    if (cSize < 5) { ZS_PREP_WHY(__LINE__); break; }
                    const uint32_t lhc = rd32(bs);
                    uint32_t lhSize, litSize, litCSize; bool single = false;
                    if (lhl < 2) { single = !lhl; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; }
                    else if (lhl == 2) { lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; }
                    else { lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((uint32_t)bs[4] << 10); }
                    if (litSize > litCap || litCSize + lhSize > cSize) { ZS_PREP_WHY(__LINE__); break; }
                    if (!single && (litSize == 0 || litCSize == 0)) { ZS_PREP_WHY(__LINE__); break; }
                    uint16_t *ht = reinterpret_cast<uint16_t *>(hufTabs + slot * ZS_FAST_HUFTAB_BYTES);
                    uint32_t h = 0;
                    if (type == 2) {
                        h = readHufTableT<true>(L, bs + lhSize, litCSize, ht, ZS_FAST_HUFLOG);
                        if (isErr(h)) { ZS_PREP_WHY(__LINE__); break; }
                        const uint32_t flat = h >> 30; h &= 0x3FFFFFFFu;       // (readHufTableT<true> marks a flat table in bit 30)
                        if (h >= litCSize || L.hufLog > ZS_FAST_HUFLOG) { ZS_PREP_WHY(__LINE__); break; }
                        if (lane == 0) L.misc[13] = L.hufLog | (flat << 8);
                    } else {
                        // the table of the block that built it, into this block's slot (the whole 4 KiB: 16 bytes a lane, 4 rounds)
                        const uint4 *from = reinterpret_cast<const uint4 *>(hufTabs + ((size_t)(L.misc[11] - 1u) * cap + item) * ZS_FAST_HUFTAB_BYTES);
                        uint4 *to = reinterpret_cast<uint4 *>(ht);
                        uint4 v[ZS_FAST_HUFTAB_BYTES / 16 / 64];
                        wave_mem_sync();                               // (stored by this wavefront, some blocks earlier)
                        #pragma unroll
                        for (uint32_t u = 0; u < ZS_FAST_HUFTAB_BYTES / 16 / 64; u++) v[u] = from[lane + 64 * u];
                        #pragma unroll
                        for (uint32_t u = 0; u < ZS_FAST_HUFTAB_BYTES / 16 / 64; u++) to[lane + 64 * u] = v[u];
                    }
                    wave_sync();
                    if (lane == 0) L.misc[11] = blk + 1;
                    wave_sync();
                    DSET(hufFlat, L.misc[13] >> 8);
                    const uint32_t cs0 = b0 + lhSize + h, csz = litCSize - h;
                    DSET(litType, 2u); DSET(litSize, litSize); DSET(hufLog, L.misc[13] & 0xFFu);
                    if (single) { DSET(nStreams, 1u); DSET(sOff[0], cs0); DSET(sLen[0], csz); DSET(sCnt[0], litSize); DSET(sOut[0], 0u); }
                    else {
                        if (csz < 10) { ZS_PREP_WHY(__LINE__); break; }
                        const uint8_t *cs = src + cs0;
                        const uint32_t l1 = rd16(cs), l2 = rd16(cs + 2), l3 = rd16(cs + 4);
                        if (l1 + l2 + l3 + 6 > csz) { ZS_PREP_WHY(__LINE__); break; }
                        const uint32_t seg = (litSize + 3) / 4;
                        if (3 * seg > litSize) { ZS_PREP_WHY(__LINE__); break; }
                        if (lane == 0) {
                            dp->nStreams = 4;
                            dp->sOff[0] = cs0 + 6; dp->sOff[1] = cs0 + 6 + l1; dp->sOff[2] = cs0 + 6 + l1 + l2; dp->sOff[3] = cs0 + 6 + l1 + l2 + l3;
                            dp->sLen[0] = l1; dp->sLen[1] = l2; dp->sLen[2] = l3; dp->sLen[3] = csz - (l1 + l2 + l3 + 6);
                            for (uint32_t k = 0; k < 4; k++) { dp->sCnt[k] = k < 3 ? seg : litSize - 3 * seg; dp->sOut[k] = k * seg; }
                        }
                    }
                    litCSizeTot = litCSize + lhSize;
                } else {
                    uint32_t lhSize, litSize;
                    if (lhl == 1) { lhSize = 2; litSize = rd16(bs) >> 4; }
                    else if (lhl == 3) { lhSize = 3; litSize = rd24(bs) >> 4; }
                    else { lhSize = 1; litSize = bs[0] >> 3; }
                    if (type == 0) { if (litSize + lhSize > cSize) { ZS_PREP_WHY(__LINE__); break; } DSET(litType, 0u); DSET(litSrc, b0 + lhSize); litCSizeTot = lhSize + litSize; }
                    else { if (lhSize + 1 > cSize || litSize > litCap) { ZS_PREP_WHY(__LINE__); break; } DSET(litType, 1u); DSET(litSrc, (uint32_t)bs[lhSize]); litCSizeTot = lhSize + 1; }
                    DSET(litSize, litSize); DSET(nStreams, 0u);
                }
            }
            if (litCSizeTot > cSize) { ZS_PREP_WHY(__LINE__); break; }
            // ---- sequence headers + tables (:1110-1180), then the tables leave for the sequences kernel.  st.fseEntropy stays 0: a
            //      table repeated from the block before (mode 3) is an error here and sends the item to the general kernel ----
            const uint8_t *ip = bs + litCSizeTot; uint32_t remaining = cSize - litCSizeTot, nbSeq = 0;
            const uint32_t seqPrev = L.misc[12];                    // block + 1 of the last block that had sequences (its slot holds all three tables)
            DState st; st.rep[0] = 1; st.rep[1] = 4; st.rep[2] = 8; st.litEntropy = 0; st.fseEntropy = seqPrev != 0; st.llRepeatOk = 0; st.hufX4 = 0;
            uint16_t *stab = reinterpret_cast<uint16_t *>(seqTabs + slot * ZS_FAST_SEQTAB_BYTES);
            const uint16_t *stabPrev = seqPrev ? reinterpret_cast<const uint16_t *>(seqTabs + ((size_t)(seqPrev - 1u) * cap + item) * ZS_FAST_SEQTAB_BYTES) : nullptr;
            if (seqHeadersT<true>(L, st, ip, remaining, nbSeq, stab, &L.misc[8], stabPrev)) { ZS_PREP_WHY(__LINE__); break; }      // (misc[8..10]: a repeated table keeps the log it had)
            if (nbSeq > seqCap) { ZS_PREP_WHY(__LINE__); break; }
            if (nbSeq == 0 && remaining != 0) { ZS_PREP_WHY(__LINE__); break; }
            DSET(nbSeq, nbSeq); DSET(seqOff, (uint32_t)(ip - src)); DSET(seqSize, remaining);
            DSET(llLog, nbSeq ? L.misc[8] : 0u); DSET(ofLog, nbSeq ? L.misc[9] : 0u); DSET(mlLog, nbSeq ? L.misc[10] : 0u);
            if (nbSeq && lane == 0) {
                L.misc[12] = blk + 1;
                // the block joins the list of its table class (k_dec_sequences takes its items from the lists: a launch over every slot with most lanes
                // idle cost a whole chain's time for a class that holds a tenth of the blocks - libzstd's 32 KiB frames: 9 % have 2^9-cell tables)
                const uint32_t cls = (L.misc[8] > 8u || L.misc[10] > 8u) ? 1u : 0u;
                const uint32_t at = atomicAdd(&seqLists[cls], 1u);
                seqLists[2 + (size_t)cls * cap * maxBlocks + at] = (uint32_t)slot;
            }
            wave_sync();
            DSET(fast, 1u);
            nBlocks = blk + 1;
            b0 += cSize;
            fail = false;
            if (lastBlock) break;
        }
        wave_sync();
        ok = !fail;                                                     // (round 4: also a frame of raw / RLE blocks only - BASELINE config 1's 1 MiB of zeros is 16 RLE blocks - : a lane-parallel copy / fill in k_dec_execute instead of a wavefront an item in the general kernel)
    } while (0);
#ifdef ZS_PREP_PROFILE
    if (lane == 0 && !descs[item].hufFlat) {          // phases 0-11, then the wavefront's whole time: behind the item's Huffman table (the two-level table ends at 1280 bytes; a flat one fills the slot)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(hufTabs + (size_t)item * ZS_FAST_HUFTAB_BYTES + 2048);
        for (int k = 0; k < 12; k++) o[k] = L.pp[k];
        o[12] = __builtin_amdgcn_s_memtime() - ppStart;
    }
#endif
    // an item is fast only as a whole; a block index it does not use reads as absent
    if (lane == 0) {
        const uint32_t slots = max(2u, maxBlocks);                               // (a descriptor slot 1 exists even when the call reserved one block slot)
        for (uint32_t b = ok ? nBlocks : 0u; b < slots; b++) descs[(size_t)b * cap + item].fast = 0;
        if (!ok) descs[item].why = prepWhy;
    }
    #undef ZS_PREP_WHY
    #undef DSET
}


// A lane stages the window of ITS OWN stream: win[] <- stream bytes [base - 8, base + W), zero outside [0, size).
// All the loads of a window are independent, so they overlap (a loop over streams with one load each serialises a memory
// round trip per stream: that was 90 % of these kernels' time).
template <uint32_t W>
__device__ __forceinline__ void stageOwnWindow(uint32_t *win, const uint8_t *src, uint32_t size, int32_t base)
{
    constexpr uint32_t N = (W + 8) / 4 + 2;
    static_assert(N % 4 == 0, "the window is staged in 16-byte pieces");
    if (base >= 8 && (uint32_t)base - 8u + 4u * N <= size) {
        // the whole window lies inside the stream (every window but a stream's first and last): 16-byte pieces, no fix-up per dword
        // (a quarter of the vector-memory instructions, ~20 instead of ~430 vector-ALU instructions a window)
        const uint8_t *p = src + base - 8;
        #pragma unroll
        for (uint32_t k0 = 0; k0 < N / 4; k0 += 8) {
            uint4 v[8];
            #pragma unroll
            for (uint32_t u = 0; u < 8; u++) if (k0 + u < N / 4) __builtin_memcpy(&v[u], p + 16 * (k0 + u), 16);
            #pragma unroll
            for (uint32_t u = 0; u < 8; u++) if (k0 + u < N / 4) *reinterpret_cast<uint4 *>(win + 4 * (k0 + u)) = v[u];
        }
    } else if (size >= 4) {
        // branch free, so that the loads of a batch are issued together: every dword comes from a clamped address and is
        // shifted / zeroed where it sticks out of the stream
        const int32_t last = (int32_t)size - 4;
        #pragma unroll
        for (uint32_t j0 = 0; j0 < N; j0 += 16) {
            uint32_t v[16];
            #pragma unroll
            for (uint32_t u = 0; u < 16; u++) {
                const int32_t p = base - 8 + 4 * (int32_t)(j0 + u);
                const int32_t q = min(max(p, 0), last);
                v[u] = (j0 + u < N) ? zs_load32(src + q) : 0u;
            }
            #pragma unroll
            for (uint32_t u = 0; u < 16; u++) {
                if (j0 + u < N) {
                    const int32_t p = base - 8 + 4 * (int32_t)(j0 + u);
                    const int32_t q = min(max(p, 0), last);
                    const int32_t dlt = p - q;                                         // < 0: sticks out below, > 0: above
                    uint32_t x = v[u];
                    x = (dlt < 0) ? ((dlt > -4) ? x << (8 * (uint32_t)(-dlt)) : 0u) : ((dlt > 0) ? ((dlt < 4) ? x >> (8 * (uint32_t)dlt) : 0u) : x);
                    win[j0 + u] = x;
                }
            }
        }
    } else {
        for (uint32_t j = 0; j < N; j++) {
            const int32_t p = base - 8 + 4 * (int32_t)j;
            uint32_t v = 0;
            for (int q = 0; q < 4; q++) { const int32_t r = p + q; if (r >= 0 && r < (int32_t)size) v |= (uint32_t)src[r] << (8 * q); }
            win[j] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dec_huffman : lane 4g + k = stream k of item g.  Tables in LDS (16 x 4 KiB), stream windows in LDS, refilled in rounds
// by all lanes; symbol loop of the general decoder (four symbols per refill, then the careful tail).
// ---------------------------------------------------------------------------------------------------------------------
// Two table classes (as the sequences kernel's): FLAT = false, the two-level table (1.25 KiB an item, 16 items a wavefront); FLAT = true, items
// whose codes need more sub-tables than it has - wide alphabets, binaries - with the flat table of 2^11 entries (4 KiB an item, 8 items a
// wavefront = 32 lanes).  Both are launched over all groups; a lane takes its stream only in the kernel of its item's class.
template <bool FLAT, uint32_t G>
struct HufLds { uint16_t huf[G][FLAT ? (1u << ZS_FAST_HUFLOG) : ZS_HUF2_ENTRIES]; uint32_t win[4 * G][(ZS_FAST_HUFWIN + 8) / 4 + 2]; };

template <bool FLAT, uint32_t G>
__device__ __forceinline__ void zs_dec_huffman_body(HufLds<FLAT, G> &H, const uint32_t bid, const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, ZsFastDesc *__restrict__ descs,
              const uint8_t *__restrict__ hufTabs, uint8_t *__restrict__ litScratchAll, uint32_t nBlk, uint32_t cap, uint32_t litStride)
{
    const uint32_t lane = (uint32_t)zs_lane();
    const uint32_t g = lane >> 2, k = lane & 3u;
    // one launch for every block index of the call (nBlk of them): the grid is nBlk runs of the items' groups - blocks decode independently of
    // each other, and a launch per block index was a launch of few wavefronts each when the items are large frames (r3: 16 launches -> 1)
    const uint32_t groupsPerBlk = (nItems + G - 1) / G;
    const uint32_t blk = bid / groupsPerBlk, bx = bid - blk * groupsPerBlk;
    (void)nBlk;
    const uint32_t item = bx * G + g;
    bool mine = false; uint32_t dtLog = 1, n = 0, size = 0;
    const uint8_t *src = srcAll; uint8_t *out = litScratchAll;
    const size_t slot0 = (size_t)blk * cap;                          // this block index's descriptors, tables, literal scratch
    if (g < G && item < nItems) {
        const ZsFastDesc *d = descs + slot0 + item;
        if (descs[item].fast && d->fast && d->litType == 2 && k < d->nStreams && (d->hufFlat != 0) == FLAT) {
            mine = true; dtLog = d->hufLog; n = d->sCnt[k]; size = d->sLen[k];
            src = srcAll + items[item].srcOff + d->sOff[k];
            out = litScratchAll + (slot0 + item) * litStride + d->sOut[k];
        }
    }
    if (!__ballot(mine)) return;
    // tables of the items that need them
    for (uint32_t gg = 0; gg < G; gg++) {
        const uint32_t it2 = bx * G + gg;
        const uint32_t log2 = wave_get(mine ? dtLog : 0u, (int)(gg * 4));        // stream 0 of the item exists whenever any does
        if (!log2) continue;
        const uint32_t *ht = reinterpret_cast<const uint32_t *>(hufTabs + (slot0 + it2) * ZS_FAST_HUFTAB_BYTES);
        uint32_t *dstw = reinterpret_cast<uint32_t *>(H.huf[gg]);
        {   // 5 (flat: 16) dwords per lane, the loads issued together
            constexpr uint32_t words = (FLAT ? (1u << ZS_FAST_HUFLOG) : ZS_HUF2_ENTRIES) / 2, per = (words + 63) / 64;
            uint32_t v[per];
            #pragma unroll
            for (uint32_t u = 0; u < per; u++) v[u] = (lane + 64 * u < words) ? ht[lane + 64 * u] : 0u;
            #pragma unroll
            for (uint32_t u = 0; u < per; u++) if (lane + 64 * u < words) dstw[lane + 64 * u] = v[u];
        }
    }
    BitC b; b.c = 0; b.avail = 0; b.bitPos = 0;
    bool ok = !mine || bc_init(b, src, size);
    uint32_t i = 0;
    bool done = !mine || !ok || n == 0;
    const uint16_t *huf = H.huf[min(g, G - 1u)];
    const uint32_t *win = H.win[min(lane, 4u * G - 1u)];
    for (;;) {
        const int32_t base = bc_windowBase(b, ZS_FAST_HUFWIN);
        wave_sync();
        if (!done) stageOwnWindow<ZS_FAST_HUFWIN>(H.win[min(lane, 4u * G - 1u)], src, size, base);
        wave_sync();
        if (!done) {
            while (i + 4 <= n) {
                const int32_t bh = (b.bitPos - 1) >> 3;
                if (b.bitPos < 64 || (base > 0 && bh < base + 16)) break;
                uint64_t c = win64(win, (uint32_t)(bh - base + 1)) << (7u - (uint32_t)((b.bitPos - 1) & 7));
                uint32_t used = 0, pack = 0;
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t top = (uint32_t)(c >> 32);
                    uint32_t e = FLAT ? huf[top >> 21] : huf[top >> 23];             // 9 bits; codes of 10 / 11 bits: 2 more in a sub-table (flat: 11 bits at once)
                    if (!FLAT && (e & 0x8000u)) e = huf[512u + ((e & 0x7FFFu) << 2) + ((top >> 21) & 3u)];
                    const uint32_t nb = e >> 8;
                    c <<= nb; used += nb; pack |= (e & 0xFFu) << (8 * q);
                }
                b.bitPos -= (int32_t)used;
                __builtin_memcpy(out + i, &pack, 4);
                i += 4;
            }
            b.avail = 0;
            while (i < n) {
                if (b.avail < dtLog) {
                    if (b.bitPos > 0 && base > 0 && ((b.bitPos - 1) >> 3) < base + 8) break;
                    bc_refill(b, win, base);
                }
                const uint32_t top = (uint32_t)(b.c >> 32);
                uint32_t e = FLAT ? huf[top >> 21] : huf[top >> 23];
                if (!FLAT && (e & 0x8000u)) e = huf[512u + ((e & 0x7FFFu) << 2) + ((top >> 21) & 3u)];
                const uint32_t nb = e >> 8;
                b.c <<= nb; b.avail -= nb; b.bitPos -= (int32_t)nb;
                out[i++] = (uint8_t)e;
            }
            if (i == n) done = true;
        }
        if (!__ballot(!done)) break;
    }
    // a stream must end exactly (BitStream.cs:494); otherwise the general decoder takes the item
    if (mine && (!ok || b.bitPos != 0)) { descs[item].fast = 0; descs[item].why = 1; }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS of k_dec_sequences: per item its three tables (16-bit cells) and a window of its bitstream.
// The per-sequence code is decodeBlock's (:1473-1553); results go to global memory, 8 bytes a sequence (ZsFastSeq).
// ---------------------------------------------------------------------------------------------------------------------
// The kernel comes in two table sizes: LOG9 = false holds items whose LL and ML tables have <= 2^8 cells (blocks of <= 2048
// sequences get such tables: FSE_optimalTableLog) in 1.5 KiB, LOG9 = true the rest in 2.5 KiB.  How many items a CU decodes
// at once is set by that LDS share, and the kernel's time by how many it decodes at once.  Both are launched over all
// groups; a lane takes its item only in the kernel of the item's class.
template <bool LOG9, uint32_t G>
struct SeqDecLds { uint16_t cells[G][(LOG9 ? 1280 : 768)]; uint32_t win[G][(ZS_FAST_SEQWIN + 8) / 4 + 2]; };

// ---------------------------------------------------------------------------------------------------------------------
// k_dec_sequences : four lanes an item (lane 4 g + r: item g of the group; r = 0 the literal-length state, 1 the match-length state, 2 the
// offset state, 3 idle).  Every lane opens ONE cell a step and the quad exchanges bit counts and symbols by data-parallel moves (quad_perm):
// ~65 instructions a step where one lane an item (round 3's first form: three cells opened one after the other) took ~85 and round 2's
// branchy form ~140 - and the kernel's time is the instructions it issues: a wavefront is alone on its SIMD (~4.4 cycles an instruction),
// 16 items a wavefront either way, because the LDS holds no more.  The four lanes of an item carry the same stream position and sequence
// count, so every branch is uniform inside a quad.  A step is written without branches and with the cell format taken apart by hand:
//   cell = (1 << (9 - nb) | next >> nb) << 6 | symbol, so with p = cell >> 6:  nb = clz32(p) - 22  and  next = (p << nb) - 512;
//   the states are kept + 512 (the table pointer - 512 cells), so a new state is (p << nb) + the nb stream bits;
//   extra bits of a length code (LL_bits / ML_bits, ZStdInternal.cs:158,173) = max((code - c0) >> 1, code >= c0, code >= c1 ? code - c2 : 0)
//   with (c0, c1, c2) = (16, 25, 19) for literal lengths and (32, 43, 36) for match lengths (arithmetic shift: below c0 all three are <= 0);
//   an offset code is its own count.
// One exit test a step: the position against a limit - with a window that does not reach the stream's start, where the next step could read
// below the window (a sequence reads < 12 bytes; 24 are kept); with one that does, below 0 = the stream ended before the sequences (:1582, :1594).
// One LDS round trip a step: the cell and the 8 stream bytes below the position are read together (no bit container carried from sequence
// to sequence).  Bytes below the stream start read as zeros, as in bc_refill.
// ---------------------------------------------------------------------------------------------------------------------
#define ZS_QUAD(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true))      // (bound_ctrl: no "old" value to set up; a quad_perm reads valid lanes only)
template <bool LOG9, uint32_t G>
__device__ __forceinline__ void zs_dec_sequences_body(SeqDecLds<LOG9, G> &S, const uint32_t bid, const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, ZsFastDesc *__restrict__ descs,
                  const uint8_t *__restrict__ seqTabs, ZsFastSeq *__restrict__ seqOutAll, uint32_t nBlk, uint32_t cap, const uint32_t *__restrict__ seqLists, uint32_t seqCap,
                  uint32_t listedMin = 0, uint32_t listedMax = 0xFFFFFFFFu)
{
    // the blocks of this table class, listed by k_dec_prep (every block index of the call in one launch: blocks decode independently)
    const uint32_t listed = seqLists[LOG9 ? 1 : 0];
    if (bid * G >= listed || listed < listedMin || listed >= listedMax) return;      // (listedMin / Max: the launch serves the class only when it holds that many blocks - the host cannot know)
    const uint32_t *list = seqLists + 2 + (LOG9 ? (size_t)cap * nBlk : 0);
    static_assert(G <= 16, "four lanes an item");
    constexpr uint32_t LLC = LOG9 ? 512 : 256, OFB = LLC, MLB = LLC + 256;        // cells of the LL table; where OF and ML start
    const uint32_t lane = (uint32_t)zs_lane();
    const uint32_t g = lane >> 2, r = lane & 3u;
    const uint32_t v = bid * G + g;
    const uint32_t slot = (g < G && v < listed) ? list[v] : 0u;                 // block index * cap + item
    const uint32_t item = slot % cap;
    bool mine = false; uint32_t nbSeq = 0, size = 0, llLog = 0, ofLog = 0, mlLog = 0;
    const uint8_t *src = srcAll;
    if (g < G && v < listed && item < nItems) {
        const ZsFastDesc *d = descs + slot;
        if (descs[item].fast && d->fast && d->nbSeq && ((d->llLog > 8 || d->mlLog > 8) == LOG9)) { mine = true; nbSeq = d->nbSeq; size = d->seqSize; llLog = d->llLog; ofLog = d->ofLog; mlLog = d->mlLog; src = srcAll + items[item].srcOff + d->seqOff; }
    }
    if (!__ballot(mine)) return;
    for (uint32_t gg = 0; gg < G; gg++) {
        if (!wave_get(mine ? 1u : 0u, (int)(4 * gg))) continue;
        const uint32_t *st = reinterpret_cast<const uint32_t *>(seqTabs + (size_t)wave_get(slot, (int)(4 * gg)) * ZS_FAST_SEQTAB_BYTES);
        const uint32_t a = 1u << wave_get(llLog, (int)(4 * gg)), o = 1u << wave_get(ofLog, (int)(4 * gg)), m = 1u << wave_get(mlLog, (int)(4 * gg));
        {   // the three tables, two cells a dword: every load issued before the first LDS store (up to 4 + 2 + 4 dwords per lane)
            uint32_t va[4], vo[2], vm[4];
            uint32_t *cw = reinterpret_cast<uint32_t *>(S.cells[gg]);
            #pragma unroll
            for (uint32_t u = 0; u < 4; u++) va[u] = (2 * (lane + 64 * u) < a) ? st[lane + 64 * u] : 0u;
            #pragma unroll
            for (uint32_t u = 0; u < 2; u++) vo[u] = (2 * (lane + 64 * u) < o) ? st[256 + lane + 64 * u] : 0u;
            #pragma unroll
            for (uint32_t u = 0; u < 4; u++) vm[u] = (2 * (lane + 64 * u) < m) ? st[384 + lane + 64 * u] : 0u;
            #pragma unroll
            for (uint32_t u = 0; u < 4; u++) if (2 * (lane + 64 * u) < a) cw[lane + 64 * u] = va[u];
            #pragma unroll
            for (uint32_t u = 0; u < 2; u++) if (2 * (lane + 64 * u) < o) cw[OFB / 2 + lane + 64 * u] = vo[u];
            #pragma unroll
            for (uint32_t u = 0; u < 4; u++) if (2 * (lane + 64 * u) < m) cw[MLB / 2 + lane + 64 * u] = vm[u];
        }
    }
    BitC b; b.c = 0; b.avail = 0; b.bitPos = 0;
    bool ok = !mine || bc_init(b, src, size);
    const uint32_t gi = min(g, G - 1u);
    uint32_t *winW = S.win[gi];
    const uint32_t *win = winW;
    ZsFastSeq *outp = seqOutAll + (size_t)slot * seqCap;
    // what a lane's role fixes: its table, the constants of its code's extra-bit count (see k_dec_sequences; an offset code IS its count),
    // where its state bits sit below the other states' (LL on top, then ML, then OF, :1547-1550)
    const uint16_t *cellsB = S.cells[gi] + (r == 1 ? MLB : (r == 2 ? OFB : 0u)) - 512;
    const int32_t c0 = (r == 1) ? 32 : 16, c1 = (r == 1) ? 43 : 25, c2 = (r == 1) ? 36 : 19;
    const uint32_t lenMask = (r < 2) ? 0xFFFFFFFFu : 0u, ofMask = (r == 2) ? 63u : 0u, nbMask = (r < 3) ? 0xFu : 0u;
    const uint32_t aboveM = (r == 0) ? 0xFFu : 0u, aboveO = (r < 2) ? 0xFFu : 0u;       // counts of the states whose bits lie below mine
    uint32_t st1 = 512, t = 0;                                                           // my state + 512
    bool started = false, done = !mine || !ok;
    #define FSEQ_NEED(nbits) do { if (b.avail < (nbits)) bc_refill(b, win, base); } while (0)
    for (;;) {
        const int32_t base = bc_windowBase(b, ZS_FAST_SEQWIN);
        wave_sync();
        if (!done) {
            // the item's window, its 16-byte pieces dealt to the four lanes (a first or last window, which sticks out of the stream: lane 0 alone)
            constexpr uint32_t N = (ZS_FAST_SEQWIN + 8) / 4 + 2;
            if (base >= 8 && (uint32_t)base - 8u + 4u * N <= size) {
                const uint8_t *p = src + base - 8;
                uint4 v[(N / 4 + 3) / 4];
                #pragma unroll
                for (uint32_t u = 0; u < (N / 4 + 3) / 4; u++) if (4 * u + r < N / 4) __builtin_memcpy(&v[u], p + 16 * (4 * u + r), 16);
                #pragma unroll
                for (uint32_t u = 0; u < (N / 4 + 3) / 4; u++) if (4 * u + r < N / 4) *reinterpret_cast<uint4 *>(winW + 4 * (4 * u + r)) = v[u];
            } else if (r == 0) stageOwnWindow<ZS_FAST_SEQWIN>(winW, src, size, base);
        }
        wave_sync();
        if (!done) {
            if (!started) {
                FSEQ_NEED(llLog + ofLog + mlLog);
                const uint32_t iLL = bc_take(b, llLog), iOF = bc_take(b, ofLog), iML = bc_take(b, mlLog);
                st1 = 512 + (r == 1 ? iML : (r == 2 ? iOF : iLL)); started = true;
            }
            const int32_t lim = (base > 0) ? 8 * (base + 24) : -1;
            int32_t bp = b.bitPos;
            while (t < nbSeq && bp > lim) {
                const int32_t bh = (bp - 1) >> 3;                        // bp == 0: -1, the 8 zero bytes in front of the stream
                const uint32_t cell = cellsB[st1];
                uint64_t raw = win64(win, (uint32_t)(bh - base + 1));
                asm volatile("" : "+v"(raw));                            // (read here, beside the cell)
                const uint32_t y = cell & 63u, p = cell >> 6;
                const uint32_t nb = ((uint32_t)__builtin_clz(p) - 22u) & nbMask;
                const int32_t iy = (int32_t)y;
                const uint32_t xLen = (uint32_t)max(max((iy - c0) >> 1, (int32_t)(iy >= c0)), (iy >= c1) ? iy - c2 : 0);
                const uint32_t xOwn = (xLen & lenMask) | (y & ofMask);
                // the quad's sums (every lane gets both) and the other states' counts and symbols
                const uint32_t x2 = xOwn + ZS_QUAD(xOwn, 0xB1), xbits = x2 + ZS_QUAD(x2, 0x4E);
                const uint32_t n2 = nb + ZS_QUAD(nb, 0xB1), sbits = n2 + ZS_QUAD(n2, 0x4E);
                const uint32_t nM = ZS_QUAD(nb, 0x55), nO = ZS_QUAD(nb, 0xAA), yML = ZS_QUAD(y, 0x55), yOF = ZS_QUAD(y, 0xAA);
                if (r == 0) outp[t] = zs_fastseq((uint32_t)bp, y, yML, yOF);
                t++;
                uint64_t c = (raw << (7u - (uint32_t)((bp - 1) & 7))) << xbits;     // >= 57 valid bits from the top, the extra bits skipped
                const int32_t p2 = bp - (int32_t)xbits;
                if (xbits + sbits > 57u)                                 // rare (a very long offset + long lengths): read again at the state bits
                    c = (p2 <= 0) ? 0ull : win64(win, (uint32_t)(((p2 - 1) >> 3) - base + 1)) << (7u - (uint32_t)((p2 - 1) & 7));
                const uint32_t x = __builtin_amdgcn_ubfe((uint32_t)(c >> 32), 32u - sbits, sbits);
                bp = p2 - (int32_t)sbits;
                st1 = (p << nb) + __builtin_amdgcn_ubfe(x, (nM & aboveM) + (nO & aboveO), nb);
            }
            b.bitPos = bp;
            if (t < nbSeq && lim < 0) ok = false;                        // stream exhausted before all sequences
            if (t == nbSeq || !ok) done = true;
        }
        if (!__ballot(!done)) break;
    }
    #undef FSEQ_NEED
    if (mine && !ok && r == 0) { descs[item].fast = 0; descs[item].why = 2; }
}

// ---------------------------------------------------------------------------------------------------------------------
// The matches of one tile of <= 64 sequences, lane t = sequence t: destination mdst, length ml, offset off (ml == 0: no sequence).
// The literals of the whole block are in place already (k_dec_execute expands them before any match).  Order: matches whose source ends
// before the tile's first output byte side by side, then the matches that read this tile's own output - their sources first redirected
// through the matches they read, what is left level by level of its dependence (below).  Between a round's stores and the next round's loads
// of the same bytes the wavefront waits for its stores (wave_mem_sync: a workgroup-scope fence, s_waitcnt vmcnt(0)); dropping the wait changed
// nothing (the wait for a round's loads covers the stores before them: loads and stores share the counter).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef ZS_EXEC_ROUNDS
#define ZS_EXEC_ROUNDS 3                // rounds of pointer jumping a tile's matches get (what is left after them stays a dependent match)
#endif
__device__ __forceinline__ void execTileMatchesFast(uint32_t mdst, uint32_t ml, uint32_t off, uint8_t *dstBase, uint32_t tileStart, uint32_t safeEnd, uint32_t *lds3)
{
    // Every piece below is an instruction only if some lane needs it, and short matches share the first load (a vector-memory instruction
    // costs a CU ~7 ns with a few lanes active, ~19 ns with all 64: tools/probe/vmem_rate.hip; the rounds themselves are latency).
    const uint32_t lane = (uint32_t)zs_lane();
    uint32_t msrc = mdst - off;
    lds3[lane] = ml ? mdst : 0xFFFFFFFFu;                                // the tile's match destinations, ascending over the lanes; lanes without a sequence behind every position
    // A match that reads this tile's own output waits for the matches that write it: a memory round trip per level of that dependence
    // (measured on the bench frames: 24 such matches a tile, 8.7 rounds).  Most of those sources lie INSIDE the destination of one earlier
    // match of the tile, whose bytes are a copy themselves: the reader takes them from that match's source instead (and so on: every lane
    // follows the current sources of the others, a few rounds of pointer jumping over the 64 sequences in LDS).  What is left reads across
    // a sequence border (12 matches a tile, 2 levels).  Matches with overlapping source and destination neither move nor are read through.
    {
        uint32_t *dstArr = lds3, *endArr = lds3 + 64, *srcArr = lds3 + 128;
        const bool periodic = off < ml;
        endArr[lane] = (ml && !periodic) ? mdst + ml : 0u;               // 0: not to be read through
        bool act = ml && !periodic && msrc >= tileStart;
        #pragma unroll 1
        for (uint32_t round = 0; round < ZS_EXEC_ROUNDS && __ballot(act); round++) {
            srcArr[lane] = msrc;
            wave_sync();
            if (act) {
                uint32_t pos = 0;
                #pragma unroll
                for (uint32_t step = 32; step >= 1; step >>= 1) if (dstArr[pos + step] <= msrc) pos += step;
                const uint32_t d0 = dstArr[pos];
                act = d0 <= msrc && msrc + ml <= endArr[pos];            // inside that match (pos < lane: its destination starts in front of this source)
                if (act) { msrc = srcArr[pos] + (msrc - d0); act = msrc >= tileStart; }
            }
            wave_sync();
        }
    }
    wave_sync();
    const uint32_t effOff = mdst - msrc;                                 // (>= off: the same bytes from further back)
    const bool indep = ml && (msrc + ml <= tileStart);
    // matches of <= 32 bytes whose source and destination do not overlap, a lane each: 8-byte pieces k = 0..3 at min(8 k, ml - 8) (the last one
    // ends with the match); below 8 bytes one 8-byte load (if it stays inside the item's buffer) and two overlapping 4-byte stores
    auto copyShort = [&](bool mine) {
        const bool wide = mine && (ml >= 8 || msrc + 8 <= safeEnd);
        const uint32_t lastAt = (ml >= 8) ? ml - 8 : 0u;
        uint64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (wide) v0 = zs_load64(dstBase + msrc);
        if (mine && ml > 8) v1 = zs_load64(dstBase + msrc + min(8u, lastAt));
        if (mine && ml > 16) v2 = zs_load64(dstBase + msrc + min(16u, lastAt));
        if (mine && ml > 24) v3 = zs_load64(dstBase + msrc + lastAt);
        if (mine && ml >= 8) __builtin_memcpy(dstBase + mdst, &v0, 8);
        if (mine && ml > 8) __builtin_memcpy(dstBase + mdst + min(8u, lastAt), &v1, 8);
        if (mine && ml > 16) __builtin_memcpy(dstBase + mdst + min(16u, lastAt), &v2, 8);
        if (mine && ml > 24) __builtin_memcpy(dstBase + mdst + lastAt, &v3, 8);
        if (wide && ml < 8) {
            if (ml >= 4) {
                const uint32_t a = (uint32_t)v0, b = (uint32_t)(v0 >> (8u * (ml - 4u)));
                __builtin_memcpy(dstBase + mdst, &a, 4); __builtin_memcpy(dstBase + mdst + ml - 4, &b, 4);
            } else for (uint32_t j = 0; j < ml; j++) dstBase[mdst + j] = (uint8_t)(v0 >> (8u * j));
        }
        if (mine && !wide) for (uint32_t j = 0; j < ml; j++) dstBase[mdst + j] = dstBase[msrc + j];    // (a short match at the very end of the buffer)
    };
    // one match of > 32 bytes by the whole wavefront, 8 bytes a lane.  Without overlap: pieces at min(8 j, ml - 8).  With overlap (offset < length:
    // the source is the period of `off` bytes in front of the destination, ZStdDecompress.cs:1335-1352 copies it byte by byte): a piece is 8 bytes
    // of that period from phase (8 j) % off, wrapped once - two loads - when the period is >= 8 bytes; a shorter period goes byte by byte.
    auto copyLong = [&](uint32_t m2, uint32_t o2, uint32_t s2, uint32_t d2) {
        if (o2 >= m2) {
            for (uint32_t j = lane * 8u; j < m2; j += 512u) { const uint32_t jj = min(j, m2 - 8u); const uint64_t v = zs_load64(dstBase + s2 + jj); __builtin_memcpy(dstBase + d2 + jj, &v, 8); }
        } else if (o2 >= 8u) {
            for (uint32_t j = lane * 8u; j < m2; j += 512u) {
                const uint32_t jj = min(j, m2 - 8u), ph = jj % o2, room = o2 - ph;     // room: bytes of the period from the phase on
                uint64_t v = zs_load64(dstBase + s2 + ph);                              // (reads into the destination when room < 8: those bytes are replaced)
                if (room < 8u) { const uint64_t h = zs_load64(dstBase + s2); v = (v & ((1ull << (8u * room)) - 1ull)) | (h << (8u * room)); }
                __builtin_memcpy(dstBase + d2 + jj, &v, 8);
            }
        } else { for (uint32_t j = lane; j < m2; j += 64) dstBase[d2 + j] = dstBase[s2 + (j % o2)]; }
    };
#if defined(ZS_EXEC_STOP) && ZS_EXEC_STOP == 5
    return;                                                              // timing aid: records and scans of the match pass only
#endif
    if (__ballot(indep && ml <= 32)) copyShort(indep && ml <= 32);
#if defined(ZS_EXEC_STOP) && ZS_EXEC_STOP == 6
    return;                                                              // timing aid: + short matches from before the tile
#endif
    for (uint64_t lm = __ballot(indep && ml > 32); lm; lm &= lm - 1) {
        const int t = __builtin_ctzll(lm);
        copyLong(wave_get(ml, t), 0xFFFFFFFFu, wave_get(msrc, t), wave_get(mdst, t));
    }
    wave_mem_sync();
#if defined(ZS_EXEC_STOP) && ZS_EXEC_STOP == 3
    return;                                                              // timing aid: no matches that read their own tile
#endif
    // The matches left read this tile's output across a sequence border.  Each waits for exactly the earlier matches of the tile whose
    // destination its source touches - a run of lanes [a, b], found in the sorted destinations - and only while those are pending: a round
    // copies every pending match none of whose run is pending (the first pending one always is such), so the rounds are the levels of the
    // dependence (2.1 a tile on the bench frames; taking the pending matches in order until one reads behind the first: 4.2).
    {
        uint32_t *dstArr = lds3, *fullEnd = lds3 + 128;
        uint64_t pend = __ballot(ml && !indep);
        if (pend) {
            fullEnd[lane] = mdst + ml;                                       // (the redirection's sources are no longer needed)
            wave_sync();
            uint64_t need = 0;
            if ((pend >> lane) & 1ull) {
                const uint32_t e = min(msrc + ml, mdst);                     // (an overlapping match reads its period only)
                uint32_t c = 0;
                #pragma unroll
                for (uint32_t step = 32; step >= 1; step >>= 1) if (dstArr[c + step] <= msrc) c += step;
                const uint32_t a = (msrc < dstArr[0]) ? 0u : ((msrc < fullEnd[c]) ? c : c + 1u);      // first lane whose match ends behind my source's start
                if (a < lane && dstArr[a] < e) {
                    const uint32_t d1 = dstArr[min(a + 1u, 63u)], d2 = dstArr[min(a + 2u, 63u)], d3 = dstArr[min(a + 3u, 63u)];
                    uint32_t b = (d1 >= e) ? a : ((d2 >= e) ? a + 1u : ((d3 >= e) ? a + 2u : lane - 1u));   // last lane whose match starts in front of my source's end
                    b = min(b, lane - 1u);
                    need = ((2ull << b) - 1ull) & ~((1ull << a) - 1ull);
                }
            }
            while (pend) {
                const bool in = ((pend >> lane) & 1ull) && !(need & pend);
                const uint64_t grp = __ballot(in);
                const bool self = in && (off < ml);
                if (__ballot(in && !self && ml <= 32)) copyShort(in && !self && ml <= 32);
#if defined(ZS_EXEC_STOP) && ZS_EXEC_STOP == 4
                if (0)                                                   // timing aid: rounds without their long / overlapping matches
#endif
                for (uint64_t lm = __ballot(in && (self || ml > 32)); lm; lm &= lm - 1) {
                    const int t = __builtin_ctzll(lm);
                    copyLong(wave_get(ml, t), wave_get(effOff, t), wave_get(msrc, t), wave_get(mdst, t));
                }
                wave_mem_sync();
                pend &= ~grp;
            }
        }
    }
}

// the kernels over the two bodies: each class of the Huffman and of the sequences decoding as a launch of its own ...
template <bool FLAT, uint32_t G>
__global__ void __launch_bounds__(64)
k_dec_huffman(const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, ZsFastDesc *__restrict__ descs,
              const uint8_t *__restrict__ hufTabs, uint8_t *__restrict__ litScratchAll, uint32_t nBlk, uint32_t cap, uint32_t litStride)
{
    __shared__ __attribute__((aligned(16))) HufLds<FLAT, G> H;
    zs_dec_huffman_body<FLAT, G>(H, blockIdx.x, srcAll, items, nItems, descs, hufTabs, litScratchAll, nBlk, cap, litStride);
}
template <bool LOG9, uint32_t G>
__global__ void __launch_bounds__(64)
k_dec_sequences(const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, ZsFastDesc *__restrict__ descs,
                const uint8_t *__restrict__ seqTabs, ZsFastSeq *__restrict__ seqOutAll, uint32_t nBlk, uint32_t cap, const uint32_t *__restrict__ seqLists, uint32_t seqCap,
                uint32_t listedMin, uint32_t listedMax)
{
    __shared__ __attribute__((aligned(16))) SeqDecLds<LOG9, G> S;
    zs_dec_sequences_body<LOG9, G>(S, blockIdx.x, srcAll, items, nItems, descs, seqTabs, seqOutAll, nBlk, cap, seqLists, seqCap, listedMin, listedMax);
}
// ... and ALL FOUR as one launch, for calls of a round of workgroups or less (<= ~16 K blocks): there each of the four is as long as one item's
// chain whatever the call's size (0.46 + 0.49 ms at 8192 frames of 32 KiB, the CUs at two wavefronts each), and the Huffman and the sequence
// streams do not depend on each other - side by side they take the longer chain's time.  (Two streams for the same: 0.4 ms SLOWER, the
// cross-stream waits cost more than the overlap brought.)  The grid is the four grids one behind the other; the LDS is the largest of the four
// images (45 KiB: fewer workgroups a CU than the Huffman kernel's 30 KiB allows, which is why large calls keep the separate launches).
__global__ void __launch_bounds__(64)
k_dec_entropy(const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, ZsFastDesc *__restrict__ descs,
              const uint8_t *__restrict__ hufTabs, uint8_t *__restrict__ litScratchAll, const uint8_t *__restrict__ seqTabs, ZsFastSeq *__restrict__ seqOutAll,
              uint32_t nBlk, uint32_t cap, const uint32_t *__restrict__ seqLists, uint32_t litStride, uint32_t seqCap, uint32_t gridH0, uint32_t gridH1, uint32_t gridS0)
{
    __shared__ __attribute__((aligned(16))) union U_ { HufLds<false, ZS_FAST_GROUP> h0; HufLds<true, 8u> h1; SeqDecLds<false, ZS_FAST_SEQGROUP_SMALL> s0; SeqDecLds<true, 4u> s1; } U;
    const uint32_t b = blockIdx.x;
    if (b < gridH0) zs_dec_huffman_body<false, ZS_FAST_GROUP>(U.h0, b, srcAll, items, nItems, descs, hufTabs, litScratchAll, nBlk, cap, litStride);
    else if (b < gridH0 + gridH1) zs_dec_huffman_body<true, 8u>(U.h1, b - gridH0, srcAll, items, nItems, descs, hufTabs, litScratchAll, nBlk, cap, litStride);
    else if (b < gridH0 + gridH1 + gridS0) zs_dec_sequences_body<false, ZS_FAST_SEQGROUP_SMALL>(U.s0, b - gridH0 - gridH1, srcAll, items, nItems, descs, seqTabs, seqOutAll, nBlk, cap, seqLists, seqCap);
    else zs_dec_sequences_body<true, 4u>(U.s1, b - gridH0 - gridH1 - gridS0, srcAll, items, nItems, descs, seqTabs, seqOutAll, nBlk, cap, seqLists, seqCap);
}


// ---------------------------------------------------------------------------------------------------------------------
// k_dec_execute : one wavefront per item: pass A over the decoded sequences (64 at a time), the block's literals, its matches; size check.
// ---------------------------------------------------------------------------------------------------------------------
#define ZS_EXEC_WINDOW 32768u           // output bytes whose literal bits a wavefront holds in LDS at a time
// wavefronts per SIMD the kernel is compiled for: 6 = 80 VGPRs (9 spilled), 7 = 72 (more spills).  Per 57344 frames of 32 KiB: 5: 3.79 ms, 6: 3.57,
// 7: 3.41, 8: 3.65; per 16384 frames of 128 KiB: 6: 5.42, 7: 5.57 - so a call of one-block items takes the 7 form, any other the 6 form (MW); the 8 form (64 VGPRs) only
// where it turns two rounds of wavefronts into one (zsmi_api.hip)
template <int F, int MW>
__global__ void __launch_bounds__(64 * F, MW)
k_dec_execute(const uint8_t *__restrict__ srcAll, const ZsDecItem *__restrict__ items, uint32_t nItems, ZsFastDesc *__restrict__ descs,
              ZsFastSeq *seqAll, uint8_t *__restrict__ litScratchAll, uint8_t *dstAll, uint32_t *__restrict__ dstSizes, uint32_t cap, uint32_t slots,
              uint32_t litStride, uint32_t seqCap)
{
    __shared__ uint32_t tiles[F][3][64];
    __shared__ uint32_t codeTabs[36 + 53];                                      // base | extra bits << 24 of the LL / ML codes
    __shared__ uint32_t litBits[F][ZS_EXEC_WINDOW / 32 + 4];                   // a bit per output byte of the window: toggles at match ends -> inside a match -> literal
    __shared__ uint32_t expSel[16];                                             // v_perm selectors that spread the next literals over a 4-bit mask's set bytes
    const uint32_t w = threadIdx.x >> 6, lane = (uint32_t)zs_lane();
    const uint32_t item = blockIdx.x * F + w;
    if (threadIdx.x < 36) codeTabs[threadIdx.x] = d_LL_base[threadIdx.x] | ((uint32_t)d_LL_bits[threadIdx.x] << 24);
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + 53) codeTabs[36 + threadIdx.x - 64] = d_ML_base[threadIdx.x - 64] | ((uint32_t)d_ML_bits[threadIdx.x - 64] << 24);
    if (threadIdx.x >= 128 && threadIdx.x < 144) {
        const uint32_t m = threadIdx.x - 128; uint32_t sel = 0, k = 0;
        for (uint32_t j = 0; j < 4; j++) { if (m & (1u << j)) { sel |= k << (8 * j); k++; } else sel |= 0x0Cu << (8 * j); }
        expSel[m] = sel;
    }
    __syncthreads();                                                            // the only workgroup barrier: before any wavefront leaves
    if (item >= nItems) return;
    // the item index is the same in every lane: said so, its descriptor and item record are scalar loads (27 + 6 registers that
    // would otherwise sit in every lane; the kernel lives at its 80-VGPR cap)
    const uint32_t itemU = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
    if (!descs[itemU].fast) return;
    const uint32_t hasContentSize = descs[itemU].hasContentSize, contentSize = descs[itemU].contentSize;
    const ZsDecItem it = items[itemU];
    uint8_t *dstBase = dstAll + it.dstOff;
    const uint64_t oend = it.dstCap;
    uint64_t op = 0; bool bad = false; uint32_t why = 0;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;                                      // the list carried from tile to tile and block to block (lane 0 holds it)
    #pragma unroll 1
    for (uint32_t blk = 0; blk < slots && !bad; blk++) {
    const size_t slot = (size_t)blk * cap + itemU;
    const ZsFastDesc *dp = descs + slot;
    if (blk && !dp->fast) break;                                                // a one-block frame
    struct { uint32_t litType, litSize, litSrc, nbSeq, seqOff; } d;            // what this kernel needs of the block's descriptor (scalars)
    d.litType = dp->litType; d.litSize = dp->litSize; d.litSrc = dp->litSrc; d.nbSeq = dp->nbSeq; d.seqOff = dp->seqOff;
    uint8_t *litBuf = litScratchAll + slot * litStride;
    const uint8_t *litPtr = litBuf;
    if (d.litType == 0) litPtr = srcAll + it.srcOff + d.litSrc;
    else if (d.litType == 1 && d.nbSeq != 0) { for (uint32_t j = lane; j < d.litSize; j += 64) litBuf[j] = (uint8_t)d.litSrc; wave_mem_sync(); }
    ZsFastSeq *seqs = seqAll + slot * seqCap;                           // (pass A writes each sequence back in place)
    const uint8_t *bits = srcAll + it.srcOff + d.seqOff;                        // the sequence bitstream, d.seqSize bytes
    // the 64 stream bits below bit position p, top aligned (bit p - 1 at bit 63); bits below the stream start read as 0
    auto bitsBelow = [&](int32_t p) -> uint64_t {
        if (p <= 0) return 0ull;
        const int32_t bh = (p - 1) >> 3;
        uint64_t wv;
        if (bh >= 7) wv = zs_load64(bits + bh - 7);
        else { wv = 0; for (int32_t k = 0; k <= bh; k++) wv |= (uint64_t)bits[k] << (8 * (k + 7 - bh)); }
        return wv << (7u - (uint32_t)((p - 1) & 7));
    };
    uint32_t litPos = 0;
    const uint32_t blockStart32 = (uint32_t)op; (void)blockStart32;       // where this block's output starts (a 32-bit position: the item's capacity is a 32-bit count)
    for (uint32_t i = lane; i < ZS_EXEC_WINDOW / 32 + 4; i += 64) litBits[w][i] = 0;      // pass A toggles the first window's bits as it goes
    wave_sync();
    for (uint32_t t0 = 0; t0 < d.nbSeq; t0 += 64) {
        const uint32_t T = min(64u, d.nbSeq - t0);
        // lane t: the extra bits of sequence t0 + t (:1487-1545) -> lengths, offset value and its recent-offset class
        if (lane < T) {
            const ZsFastSeq r = seqs[t0 + lane];
            const int32_t bp = (int32_t)((uint32_t)r & 0xFFFFFu);
            const uint32_t tLL = codeTabs[((uint32_t)r >> 20) & 63u], tML = codeTabs[36 + (((uint32_t)r >> 26) & 63u)], ofAdd = (uint32_t)(r >> 32) & 31u;
            const uint32_t llBase = tLL & 0xFFFFFFu, llAdd = tLL >> 24, mlBase = tML & 0xFFFFFFu, mlAdd = tML >> 24;
            uint32_t ofBits, mlBits, llBits;
            if (ofAdd + mlAdd + llAdd <= 57u) {
                const uint64_t c = bitsBelow(bp);
                ofBits = ofAdd ? (uint32_t)(c >> (64 - ofAdd)) : 0u;
                const uint64_t c2 = c << ofAdd;
                mlBits = mlAdd ? (uint32_t)(c2 >> (64 - mlAdd)) : 0u;
                const uint64_t c3 = c2 << mlAdd;
                llBits = llAdd ? (uint32_t)(c3 >> (64 - llAdd)) : 0u;
            } else {
                ofBits = (uint32_t)(bitsBelow(bp) >> (64 - ofAdd));
                const uint64_t c2 = bitsBelow(bp - (int32_t)ofAdd);
                mlBits = mlAdd ? (uint32_t)(c2 >> (64 - mlAdd)) : 0u;
                const uint64_t c3 = c2 << mlAdd;
                llBits = llAdd ? (uint32_t)(c3 >> (64 - llAdd)) : 0u;
            }
            const uint32_t ofVal = ofBaseOf(ofAdd) + ofBits;
            tiles[w][0][lane] = llBase + llBits;
            tiles[w][1][lane] = mlBase + mlBits;
            // class in the top bits: 0..3 = recent-offset code (with the "literal length 0" shift applied), 4 = a new offset
            tiles[w][2][lane] = (ofAdd <= 1) ? ((ofVal + (llBase == 0)) << 29) : ((4u << 29) | ofVal);
            if (ofAdd > 28) { bad = true; why = 3; }                                         // an offset that does not fit beside the class: general decoder
        }
        if (__ballot(bad)) { bad = true; break; }
        wave_sync();
        // recent offsets (:1509-1530).  A sequence maps the list (r0, r1, r2) to a new list whose entries are each a constant
        // (a new offset) or one of the old entries: code 0 keeps it, 1 -> (r1, r0, r2), 2 -> (r2, r0, r1), new v -> (v, r0, r1).
        // Such maps compose, so the list in front of every sequence comes from one wavefront scan.  (Code 3, "r0 - 1", is not
        // of that form: a tile that has one takes the serial loop.)
        {
            const uint32_t v = (lane < T) ? tiles[w][2][lane] : 0u, cls = v >> 29, val = v & 0x1FFFFFFFu;
            constexpr uint32_t R0 = 0xFFFFFFF0u, R1 = 0xFFFFFFF1u, R2 = 0xFFFFFFF2u;          // "old entry k"
            if (!__ballot(lane < T && cls == 3u)) {
                uint32_t m0 = (cls == 4u) ? val : ((cls == 1u) ? R1 : ((cls == 2u) ? R2 : R0));
                uint32_t m1 = (cls == 0u) ? R1 : R0;
                uint32_t m2 = (cls == 0u || cls == 1u) ? R2 : R1;
                if (lane >= T) { m0 = R0; m1 = R1; m2 = R2; }
                // inclusive scan of "mine after the other": the other covers earlier sequences
                #define REP_STEP(ctrl, rowMask) { \
                    const uint32_t o0 = (uint32_t)__builtin_amdgcn_update_dpp((int)R0, (int)m0, ctrl, rowMask, 0xF, false); \
                    const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp((int)R1, (int)m1, ctrl, rowMask, 0xF, false); \
                    const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp((int)R2, (int)m2, ctrl, rowMask, 0xF, false); \
                    if (m0 >= R0) m0 = (m0 == R0) ? o0 : ((m0 == R1) ? o1 : o2); \
                    if (m1 >= R0) m1 = (m1 == R0) ? o0 : ((m1 == R1) ? o1 : o2); \
                    if (m2 >= R0) m2 = (m2 == R0) ? o0 : ((m2 == R1) ? o1 : o2); }
                REP_STEP(0x111, 0xF) REP_STEP(0x112, 0xF) REP_STEP(0x114, 0xF) REP_STEP(0x118, 0xF) REP_STEP(0x142, 0xA) REP_STEP(0x143, 0xC)
                #undef REP_STEP
                // the map in front of me = the inclusive result of the lane below (identity for lane 0), applied to the carried list
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_update_dpp((int)R0, (int)m0, 0x138, 0xF, 0xF, false);
                const uint32_t e1 = (uint32_t)__builtin_amdgcn_update_dpp((int)R1, (int)m1, 0x138, 0xF, 0xF, false);
                const uint32_t e2 = (uint32_t)__builtin_amdgcn_update_dpp((int)R2, (int)m2, 0x138, 0xF, 0xF, false);
                const uint32_t c0 = wave_get(rep0, 0), c1 = wave_get(rep1, 0), c2 = wave_get(rep2, 0);
                #define REP_APPLY(x) (((x) >= R0) ? (((x) == R0) ? c0 : (((x) == R1) ? c1 : c2)) : (x))
                const uint32_t b0 = REP_APPLY(e0), b1 = REP_APPLY(e1), b2 = REP_APPLY(e2);
                if (lane < T) tiles[w][2][lane] = (cls == 4u) ? val : ((cls == 0u) ? b0 : ((cls == 1u) ? b1 : b2));
                // carried list after the tile: the last sequence's inclusive map applied to the carry
                const uint32_t l0 = wave_get(m0, (int)T - 1), l1 = wave_get(m1, (int)T - 1), l2 = wave_get(m2, (int)T - 1);
                rep0 = REP_APPLY(l0); rep1 = REP_APPLY(l1); rep2 = REP_APPLY(l2);
                #undef REP_APPLY
            } else if (lane == 0) {
                for (uint32_t t = 0; t < T; t++) {
                    const uint32_t vv = tiles[w][2][t], cc = vv >> 29;
                    uint32_t offset = vv & 0x1FFFFFFFu;
                    if (cc == 4u) { rep2 = rep1; rep1 = rep0; rep0 = offset; }
                    else if (cc) {
                        uint32_t temp = (cc == 3) ? rep0 - 1 : (cc == 1 ? rep1 : rep2);
                        temp += !temp;
                        if (cc != 1) rep2 = rep1;
                        rep1 = rep0; rep0 = offset = temp;
                    } else offset = rep0;
                    tiles[w][2][t] = offset;
                }
            }
        }
        wave_sync();
        {   // pass A: positions and the reference's checks (ExecSequence :1265-1352; any failure hands the item to the general decoder);
            // the sequence goes back to its 8-byte slot as (ll + ml) | ml << 18 | offset << 35 for the two passes below
            const uint32_t ll = (lane < T) ? tiles[w][0][lane] : 0u, ml = (lane < T) ? tiles[w][1][lane] : 0u, off = (lane < T) ? tiles[w][2][lane] : 0u;
            const uint32_t incl = wave_incl_scan(ll + ml), inclL = wave_incl_scan(ll);
            const uint64_t outStart64 = op + (incl - ll - ml);
            const uint32_t litStart = litPos + (inclL - ll);
            bool e = false;
            if (lane < T) {
                if ((uint64_t)ll + ml > oend - outStart64 || outStart64 > oend) e = true;
                else if (ll > d.litSize - litStart || litStart > d.litSize) e = true;
                else if (off > outStart64 + ll || off >= (1u << 29) || ml >= (1u << 17) || ll + ml >= (1u << 18)) e = true;
                else {
                    seqs[t0 + lane] = (uint64_t)(ll + ml) | ((uint64_t)ml << 18) | ((uint64_t)off << 35);
                    const uint32_t ms = (uint32_t)outStart64 + ll - blockStart32, me = ms + ml;            // the match, relative to the block's output
                    if (ml && ms < ZS_EXEC_WINDOW) {
                        const uint32_t b = min(me, ZS_EXEC_WINDOW);
                        atomicXor(&litBits[w][ms >> 5], 1u << (ms & 31u)); atomicXor(&litBits[w][b >> 5], 1u << (b & 31u));
                    }
                }
            }
            if (__ballot(e)) { bad = true; why = 4; break; }
            op += wave_last(incl); litPos += wave_last(inclL);
        }
        wave_sync();
    }
    if (!bad) {
        const uint32_t lastLL = d.litSize - litPos;
        if (lastLL > oend - op) { bad = true; why = 5; }
        else {
            const uint32_t blockStart = blockStart32, blockEnd = (uint32_t)op + lastLL;
            uint32_t *bm = litBits[w];
            wave_mem_sync();                                              // pass A's records are read back below
            // ---- the literals: window by window a bit per output byte, toggled at the ends of every match (clipped to the window), prefix xor =
            //      inside a match; then 16 output bytes a lane and round: the next popcount(literal bits) literals, spread by v_perm ----
            if (d.nbSeq == 0) {
                // a block of literals only (a raw or RLE block, a compressed block without sequences): a straight copy / fill, 16 bytes a lane and round
                // (round 4: frames of raw / RLE blocks only are on this path - BASELINE config 1's 1 MiB of zeros is 16 RLE blocks)
                uint8_t *o = dstBase + blockStart;
                const uint64_t fill = 0x0101010101010101ull * (uint64_t)(d.litSrc & 0xFFu);
                for (uint32_t j = 16 * lane; j < lastLL; j += 1024) {
                    if (j + 16 <= lastLL) {
                        uint64_t A = fill, B = fill;
                        if (d.litType != 1) { A = zs_load64(litPtr + j); B = zs_load64(litPtr + j + 8); }
                        __builtin_memcpy(o + j, &A, 8); __builtin_memcpy(o + j + 8, &B, 8);
                    } else for (uint32_t q = j; q < lastLL; q++) o[q] = (d.litType == 1) ? (uint8_t)d.litSrc : litPtr[q];
                }
            } else
#if defined(ZS_EXEC_STOP) && ZS_EXEC_STOP == 1
            if (0)                                                        // timing aid: pass A only
#endif
            for (uint32_t w0 = blockStart; w0 < blockEnd; w0 += ZS_EXEC_WINDOW) {
                const uint32_t w1 = min(w0 + ZS_EXEC_WINDOW, blockEnd);
                uint32_t pos = blockStart, matchBefore = 0;
                if (w0 != blockStart) { for (uint32_t i = lane; i < ZS_EXEC_WINDOW / 32 + 4; i += 64) bm[i] = 0; wave_sync(); }
                for (uint32_t t0 = 0; w0 != blockStart && t0 < d.nbSeq; t0 += 64) {      // (the first window's toggles were made by pass A)
                    const uint64_t r = (t0 + lane < d.nbSeq) ? seqs[t0 + lane] : 0ull;
                    const uint32_t tot = (uint32_t)r & 0x3FFFFu, ml = (uint32_t)(r >> 18) & 0x1FFFFu;
                    const uint32_t incl = wave_incl_scan(tot);
                    const uint32_t mend = pos + incl, mst = mend - ml;
                    const uint32_t a = max(mst, w0), b = min(mend, w1);
                    if (ml && a < b) { atomicXor(&bm[(a - w0) >> 5], 1u << ((a - w0) & 31u)); atomicXor(&bm[(b - w0) >> 5], 1u << ((b - w0) & 31u)); }
                    matchBefore += wave_sum(min(mend, w0) - min(mst, w0));
                    pos += wave_last(incl);
                }
                wave_sync();
                {   // prefix xor over the window (lane l: dwords 16 l .. 16 l + 15), literal bit = not inside a match
                    uint32_t x[16]; uint32_t carry = 0;
                    #pragma unroll
                    for (uint32_t k = 0; k < 16; k++) {
                        uint32_t y = bm[16 * lane + k];
                        y ^= y << 1; y ^= y << 2; y ^= y << 4; y ^= y << 8; y ^= y << 16;
                        y ^= 0u - carry; carry = y >> 31; x[k] = y;
                    }
                    const uint64_t pm = __ballot(carry != 0);
                    const uint32_t flip = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull)) & 1u;
                    #pragma unroll
                    for (uint32_t k = 0; k < 16; k++) bm[16 * lane + k] = ~(x[k] ^ (0u - flip));
                }
                wave_sync();
                uint32_t running = (w0 - blockStart) - matchBefore;        // literals of the block in front of the window
                for (uint32_t r0 = w0; r0 < w1; r0 += 1024) {
                    const uint32_t p = r0 + 16 * lane;
                    uint32_t m16 = 0;
                    if (p < w1) {
                        m16 = (bm[(p - w0) >> 5] >> ((p - w0) & 31u)) & 0xFFFFu;
                        if (w1 - p < 16) m16 &= (1u << (w1 - p)) - 1u;
                    }
                    const uint32_t cnt = (uint32_t)__popc(m16);
                    const uint32_t incl = wave_incl_scan(cnt);
                    const uint32_t so = running + incl - cnt;
                    if (cnt) {
                        uint64_t A = 0, B = 0;                            // the 16 literal bytes from so (what lies behind the literals is not read)
                        if (so + 16 <= d.litSize) { A = zs_load64(litPtr + so); B = zs_load64(litPtr + so + 8); }
                        else for (uint32_t j = 0; j < cnt; j++) { const uint64_t c = litPtr[so + j]; if (j < 8) A |= c << (8 * j); else B |= c << (8 * (j - 8)); }
                        const uint32_t c0 = (uint32_t)__popc(m16 & 0xFu), c01 = (uint32_t)__popc(m16 & 0xFFu), c2 = (uint32_t)__popc(m16 & 0xF00u);
                        const uint64_t S1 = (c01 == 0) ? A : ((c01 >= 8) ? B : ((A >> (8 * c01)) | (B << (64 - 8 * c01))));   // literal bytes from c01 on
                        const uint32_t o0 = __builtin_amdgcn_perm(0u, (uint32_t)A, expSel[m16 & 15u]);
                        const uint32_t o1 = __builtin_amdgcn_perm(0u, (uint32_t)(A >> (8 * c0)), expSel[(m16 >> 4) & 15u]);
                        const uint32_t o2 = __builtin_amdgcn_perm(0u, (uint32_t)S1, expSel[(m16 >> 8) & 15u]);
                        const uint32_t o3 = __builtin_amdgcn_perm(0u, (uint32_t)(S1 >> (8 * c2)), expSel[(m16 >> 12) & 15u]);
                        uint8_t *dp = dstBase + p;
                        if (w1 - p >= 16) { const uint64_t lo = (uint64_t)o0 | ((uint64_t)o1 << 32), hi = (uint64_t)o2 | ((uint64_t)o3 << 32); __builtin_memcpy(dp, &lo, 8); __builtin_memcpy(dp + 8, &hi, 8); }
                        else { const uint32_t ow[4] = { o0, o1, o2, o3 }; for (uint32_t j = 0; j < w1 - p; j++) if ((m16 >> j) & 1u) dp[j] = (uint8_t)(ow[j >> 2] >> (8 * (j & 3u))); }
                    }
                    running += wave_last(incl);
                }
                wave_sync();
            }
            wave_mem_sync();                                              // the matches read literals
            // ---- pass B: the matches, tile by tile ----
            uint32_t pos = blockStart;
#if defined(ZS_EXEC_STOP) && (ZS_EXEC_STOP == 1 || ZS_EXEC_STOP == 2)
            if (0)                                                        // timing aid: no matches
#endif
            for (uint32_t t0 = 0; t0 < d.nbSeq; t0 += 64) {
                const uint64_t r = (t0 + lane < d.nbSeq) ? seqs[t0 + lane] : 0ull;
                const uint32_t tot = (uint32_t)r & 0x3FFFFu, ml = (uint32_t)(r >> 18) & 0x1FFFFu, off = (uint32_t)(r >> 35);
                const uint32_t incl = wave_incl_scan(tot);
                execTileMatchesFast(pos + incl - ml, ml, off, dstBase, pos, (uint32_t)oend, &tiles[w][0][0]);
                pos += wave_last(incl);
            }
            op = blockEnd;
            wave_mem_sync();                                              // the next block's matches read these bytes
        }
    }
    }   // blocks of the item
    if (!bad && hasContentSize && op != contentSize) { bad = true; why = 6; }
    if (__ballot(bad)) { const uint32_t wmax = wave_max(why); if (lane == 0) { descs[item].fast = 0; descs[item].why = wmax ? wmax : 7u; } }
    else if (lane == 0) dstSizes[item] = (uint32_t)op;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dec_checksum : four lanes an item (round 3: one).  Items the fast kernels decoded and whose frame carries a content checksum: XXH64 of the
// output, low 32 bits against the stored value (ZStdDecompress.cs:2076-2083); a mismatch is the item's result (checksum_wrong).
// Launched only when a call's frames may carry checksums; items without one cost their lane a flag read.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_dec_checksum(const ZsDecItem *__restrict__ items, uint32_t nItems, const ZsFastDesc *__restrict__ descs, const uint8_t *__restrict__ dstAll,
               uint32_t *__restrict__ dstSizes)
{
    // four lanes an item (round 4): XXH64's four stripe accumulators side by side
    const uint32_t item = min(blockIdx.x * 16 + (threadIdx.x >> 2), nItems - 1u);
    const bool real = blockIdx.x * 16 + (threadIdx.x >> 2) < nItems;
    const ZsFastDesc *d = descs + item;
    uint32_t size = (real && d->fast && d->hasChecksum) ? dstSizes[item] : 0xFFFFFFFFu;
    const bool work = size <= 0xFFFFFF88u;
    if (!__ballot(work)) return;
    const uint64_t h = xxh64_quad(dstAll + items[item].dstOff, work ? size : 0u);     // (whole quads take part: a quad without work hashes nothing)
    if (work && (threadIdx.x & 3u) == 0 && (uint32_t)h != d->checksum) dstSizes[item] = ZE(E_checksum_wrong);
}
