// lz2.h — data structures of the LDS-resident match finder (lz2_partition.hip, lz2_find.hip).
#pragma once
#include "lz_common.h"

#define LZ2_NG_BITS   14
#define LZ2_NG        (1u << LZ2_NG_BITS)   // bucket groups per block for the overflow certificate
// tuning points (overridable at build time for A/B runs: make OUT=... EXTRA="-DLZ2_THREADS=..."; MI_CODEC_LIB selects the
// library at run time).  Same-box A/B on MI355X, 1 GB deflate, when parts still had a target size: target 2304 / capacity
// 3584 / 512 threads 11.29 GB/s; 2048 / 3072 / 768 threads 11.03; 2048 / 3072 / 1024 threads 11.07 — the kernel is not
// short of waves.  Parts are now cut greedily up to the capacity (lz2_partition.hip).
#ifndef LZ2_CAP
#define LZ2_CAP       4096u                 // largest part stage 2 takes (entries): k_lz2_find_wide, 76 KiB of LDS, two workgroups per CU
#endif
#ifndef LZ2_CAP_S
#define LZ2_CAP_S     2560u                 // what the partition aims for: k_lz2_find, 51 KiB, THREE workgroups per CU (same-box A/B against
                                            // parts of 4096 everywhere: 19.7 -> 20.4 GB/s).  A part between the two exists only where no
                                            // certified cut lies within 2560 entries (one cluster of 2561..4096 entries: runs of one byte).
#endif
#ifndef LZ2_THREADS
#define LZ2_THREADS   512                   // k_lz2_find workgroup
#endif
#define LZ2_NWAVES    (LZ2_THREADS / 64)
#ifndef LZ2_PARTBITS
#define LZ2_PARTBITS  7                     // parts per block <= 2^LZ2_PARTBITS (one radix digit of stage 1)
#endif
#define LZ2_MAXPARTS  (1u << LZ2_PARTBITS)
static_assert(LZ2_CAP % LZ2_THREADS == 0 && LZ2_CAP_S % LZ2_THREADS == 0 && LZ2_CAP_S <= LZ2_CAP, "entries per thread must be whole");
// The greedy cuts of k_lz2_partition can never run out of part numbers.  A part ends at the LAST certified group that keeps it
// within LZ2_CAP_S entries and (2^20-bucket tables) within 2^16 homes; so for two consecutive parts either their entries
// exceed LZ2_CAP_S together or their homes exceed 2^16 together — otherwise the first would have been cut where the second
// ends.  Disjoint pairs of the first kind: at most LZ_MAX_BLOCK / LZ2_CAP_S; of the second kind: at most T / 2^16 = 16 (none
// for larger tables).  Round 3 had 32 part numbers for ~27 parts of text and a silent exit to the fallback pipeline (a tenth
// of the speed) at the 33rd; an A/B build with parts of 2048 entries took that exit on every block.
static_assert(2u * (LZ_MAX_BLOCK / LZ2_CAP_S + 1u + 16u) + 2u <= LZ2_MAXPARTS, "a block could need more parts than the partition can number");
#define LZ2_GRID_PARTS 32u                  // k_lz2_find's grid covers this many parts per block (text: 26-27); parts listed beyond
                                            // the grid (never seen on text) are taken by the looping k_lz2_find_wide
#ifndef LZ2_BIG
#define LZ2_BIG       8u                    // clusters of at least this many entries leave k_lz2_find (8 or 16: the register replay holds < 16)
#endif
#define LZ2_WAVE      128u                  // ... and from this size on a whole wave replays one cluster
#define LZ2_MAXBIG    (LZ2_CAP / LZ2_BIG)   // exported clusters per part, at most
// Exported clusters start on 8-entry boundaries of the block's big arrays (the lane replay loads and stores 16 bytes =
// 8 entries at a time).  <= 65536 entries in <= 8192 clusters, <= 7 pad entries each: 2 x 65536 slots always suffice.
// A pad entry carries position 0 and result 0; consumers skip pairs whose result equals their position (lz2.h below).
#define LZ2_BIG_STRIDE (2u * LZ_MAX_BLOCK)
#define LZ2_ALIGN8(x)  (((x) + 7u) & ~7u)
#define LZ2_DESC_SMALL (LZ_MAX_BLOCK / LZ2_BIG + 32u)   // exported clusters per block, at most
// export classes: 7: 8..15, 0: 16..31, 1: 32..63, 2: 64..127 entries (a lane per cluster, 64 clusters per wave);
//                 5: 128..1024, 6: > 1024 (a wave per cluster); 3 and 4 (lane replay of 128..511) exist but are not fed
#define LZ2_NCLASS    8u                    // class 7: 8..15 entries (a lane per cluster)
#define LZ2_BIG_SMALL 1024u                 // boundary between the two wave-replay classes (1 vs 4 bitmap dwords per lane)
// wave_min: clusters of at least this many entries are replayed by a whole wave (128, 256 or 512)
__host__ __device__ __forceinline__ uint32_t lz2_class_of(uint32_t cnt, uint32_t wave_min)
{
    // wave replay: 4 = 512..1024 entries and 5 = wave_min..511 share one launch that starts the long chains first
    // (three wave-replay classes by LDS need: 3 = 128..255 entries (1.5 KiB per wave), 5 = 256..511 (3 KiB), 4 = 512..1024 (6 KiB))
    // wave_min bits 16..: the row replay is on (lz2_find.hip k_lz2_rows): 128..255 entries are a class of their own (3: rows of 8 lanes)
    const bool three = (wave_min >> 16) != 0u;
    wave_min &= 0xFFFFu;
    if (cnt >= wave_min) return cnt > LZ2_BIG_SMALL ? 6u : cnt >= 512u ? 4u : (three && cnt < 256u) ? 3u : 5u;   // (class 3 on the WAVE replay with 1.5 KiB was measured: slower)
    return cnt < 16 ? 7u : cnt < 32 ? 0u : cnt < 64 ? 1u : cnt < 128 ? 2u : cnt < 256 ? 3u : 4u;
}
// cand placeholder of an entry whose cluster was exported: the entry's OWN position (a candidate is always smaller
// than its position, so no real result collides).  Position 0xFFFF pending == LZ_NONE16: consumers preset that one
// position to "none" and let the exported list overwrite it.

struct Lz2BlockMeta {
    uint32_t n, nparts, base, fallback;
    uint32_t nbig, nbig_entries;            // filled with atomics by k_lz2_find
    uint32_t pad[2];
    uint32_t part_start[LZ2_MAXPARTS];      // offset of the part's list inside the block's plist
    uint32_t part_count[LZ2_MAXPARTS];
    uint32_t part_lo[LZ2_MAXPARTS];         // first home' of the part
};

struct Lz2BigDesc {
    uint32_t block;                          // local block index in the batch
    uint32_t start;                          // first entry in the block's big-entry arrays
    uint32_t count;
    uint32_t anom;                           // slot (relative to the cluster) of bucket 0, or ~0u
    uint32_t limit;                          // slot of bucket T for deflate's non-wrapping find, or ~0u
    uint32_t pad[3];
};

struct Lz2Scratch {
    uint8_t      *partmap;      // [nb][65536] part of every position (0xFF past the block's end): written by the partition, scanned by stage 2
    uint16_t     *plist;        // [nb][65536] positions, grouped by part, time order inside a part: written by stage 2 for the parse
    uint16_t     *cand;         // [nb][65536] find() result aligned with plist (own position = pending: see bigcand)
    Lz2BlockMeta *meta;         // [nb]
    uint32_t     *fallback_count, *fallback_list;    // blocks the first pipeline has to do
    uint32_t     *work_count;                        // [0] parts listed from the front, [1] parts above LZ2_CAP_S entries listed from the end
    uint32_t      work_slots;                        // entries of `work`
    uint64_t     *work;                              // [nb * LZ2_MAXPARTS] parts of the batch, appended by the partition (parts above LZ2_CAP_S entries from the END
                                                     // of the array backwards, counted in work_count[1]: k_lz2_find_wide takes those): block | part << 16 | entries << 24 | list start << 40
                                                     // (k_lz2_find starts its list loads from the item alone: one dependent round trip less)
    uint16_t     *bigpos, *bigrs, *bigpid, *bigcand; // [nb][LZ2_BIG_STRIDE] entries of exported clusters, (cluster, time) order
    Lz2BigDesc   *desc[LZ2_NCLASS];                  // per class: [nb * capacity of the class]
    uint32_t     *big_count;                         // [LZ2_NCLASS]
    uint64_t     *dbg;                               // phase cycle counters (MI_LZ_DEBUG=1), else NULL
    uint32_t      wave_min;                          // see lz2_class_of
    uint32_t      stop_phase;                        // measurement only (MI_LZ_STOP_PHASE=k): k_lz2_find leaves after phase k; 0 = run everything
};
