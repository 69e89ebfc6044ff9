/*
 * mi_codec.h — C ABI of the MI355X-native block-parallel compressor core
 * (libmi_codec.so: hand-written HIP kernels for gfx950 + this thin C layer).
 *
 * This is the drop-in boundary for the hot path of jdm365/Compression_Algorithms
 * (SURVEY.md section 8b).  Plain pointers and sizes only; no torch / C++ types.
 * Every entry point returns an mi_status instead of the reference's printf+exit(1).
 * The reference-NAMED wrappers (lz77_compress, huffman_compress, compress, ...) that a
 * maintainer links instead of the sources under algorithms/<dir>/ are declared in mi_lz77.h,
 * mi_huffman.h, mi_deflate.h and mi_fse.h; each is a few lines over the functions here.
 *
 * Pointer conventions
 *   d_*   device (HBM) pointers, caller-owned (hipMalloc / torch tensor.data_ptr()).
 *   h_*   host pointers.
 *   stream: a hipStream_t passed as void* (NULL = HIP's default stream, as everywhere in HIP).
 *           The host-buffer convenience calls use a private stream of the context.
 * Encoders (*_encode_dev, mi_huffman_hist/build/encode_with_tree_dev, mi_fse_normalise_dev) are asynchronous on `stream`.
 * Their scratch is the context workspace: it grows — hipDeviceSynchronize + hipFree + hipMalloc — only when a call needs
 * more than any earlier call of the context did; after a first call of the largest size a context will see, the encoders
 * neither allocate nor synchronise.  One encode per context may be in flight at a time (the workspace is shared; use one
 * context per concurrent stream).  The LZ encoders fork onto three internal streams of the context and join back into
 * `stream` with events before they return control of it.  One exception: the lz77 flavour on blocks above 64 KiB
 * synchronises `stream` once per batch of <= 256 MiB (it reads back whether a block needs the whole-block finder).
 * Decoders (*_decode_dev) synchronise `stream` before returning: MI_ERR_CORRUPT is decided on the device.  They take the
 * readable length of the stream and never read outside it, whatever an (untrusted) offset table says; every decode call
 * uses its own device status word, so decodes on different streams of one context do not interfere.
 */
#ifndef MI_CODEC_H
#define MI_CODEC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MI_OK = 0,
    MI_ERR_ARG = 1,            /* bad argument (NULL, size, params out of range)           */
    MI_ERR_HIP = 2,            /* a HIP call failed; mi_last_hip_error() has the code        */
    MI_ERR_NOMEM = 3,
    MI_ERR_CAPACITY = 4,       /* output buffer too small                                   */
    MI_ERR_EMPTY_INPUT = 5,    /* reference: "ERROR: Queue is empty" exit(1)  huffman.c:149-152 */
    MI_ERR_SINGLE_SYMBOL = 6,  /* reference: "ERROR: No code for character" exit(1) huffman.c:278-281 */
    MI_ERR_CODE_TOO_LONG = 7,  /* a Huffman code > 32 bits: the reference silently emits garbage (u32 code) */
    MI_ERR_CORRUPT = 8,        /* decoder: malformed stream                                  */
    MI_ERR_NO_DEVICE = 9,      /* no gfx950 device / HIP runtime: there is NO CPU fallback   */
    MI_ERR_UNSTABLE = 10,      /* a kernel found one of its sorts out of (key, time) order: see mi_order_violations */
    MI_ERR_TRANSPORT = 11      /* multi-device gather: RCCL missing or an RCCL call failed (mi_multi_last_transport_error) */
} mi_status;

typedef struct mi_ctx mi_ctx;

/* One context per process-and-GPU (one process per GPU is the deployment model). */
mi_status   mi_ctx_create(mi_ctx **out, int device_ordinal);
void        mi_ctx_destroy(mi_ctx *ctx);
const char *mi_status_str(mi_status s);
int         mi_last_hip_error(const mi_ctx *ctx);
const char *mi_version(void);
/* blocks until everything queued on `stream` has finished.  Returns MI_ERR_UNSTABLE once if an encoder kernel reported a
 * sort out of order since the last call (below): the stream it was building is valid but may not be the reference's. */
mi_status   mi_sync(mi_ctx *ctx, void *stream);
/* The match finders sort positions by bucket with LDS radix passes whose ranks come from returning LDS atomics — stable
 * only if the hardware serves the lanes of one such instruction in lane order.  gfx950 does (probed when the context is
 * created; MI_LZ_NO_ARANK=1 forces the ballot ranking), the ISA does not promise it, and an unstable sort would still
 * round-trip.  So every consumer of a sort checks its order, and a violation is never silent: it is counted here, the
 * context ranks with ballots from its next call on, mi_sync() returns MI_ERR_UNSTABLE once, and the host-buffer entry points
 * (mi_lz_encode, mi_deflate_h_encode — what the drop-ins call) encode again before they return.  A caller of the
 * asynchronous *_dev encoders re-encodes when mi_sync says so.  Returns the number of violations seen by this context. */
uint32_t    mi_order_violations(mi_ctx *ctx);
/* Host-side check of a block table (exclusive prefix of per-block stream lengths in BITS, nblocks+1 entries) that came
 * from a file or a peer: non-decreasing, every entry a multiple of align_bits (1: bit-packed lz77; 8: deflate tokens;
 * 32: mode-H and FSE records), last entry <= 8 * stream_bytes.  MI_OK or MI_ERR_CORRUPT.  The host-buffer decoders call
 * it themselves; callers of the *_dev decoders that hold the table on the host should. */
mi_status   mi_validate_block_table(const uint64_t *h_block_bits, uint64_t nblocks, uint64_t stream_bytes, uint32_t align_bits);

/* ------------------------------------------------------------------------------------
 * Huffman, whole buffer, one tree        replaces algorithms/huffman/huffman.c:288-328
 *   histogram (huffman.c:184-187) -> heap-exact tree (:189-211) -> tree-path codes
 *   (:217-250) -> MSB-first u32 words (:18-48)
 * ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t total_bits;
    uint64_t word_idx;          /* BitWriter.word_idx  = total_bits / 32                    */
    uint64_t bit_idx;           /* BitWriter.bit_idx   = total_bits % 32                    */
    uint64_t buffer_size;       /* BitWriter.buffer_size per huffman.c:318-320              */
    uint32_t n_symbols;
    uint32_t max_code_len;
    uint32_t status;            /* mi_status decided on the device (empty / single / too long) */
    uint32_t n_nodes;           /* tree nodes written to the tree arrays (<= 511)           */
} mi_huffman_info;

/* the tree in array form, node ids in creation order (leaves in symbol order, then merges);
 * root = n_nodes-1.  Mirrors the reference's Node{value,frequency,left,right}. */
typedef struct {
    uint32_t frequency[511];
    int16_t  left[511];         /* -1 for a leaf */
    int16_t  right[511];
    uint8_t  value[511];
    uint8_t  pad;
    uint32_t code[256];
    uint8_t  length[256];
} mi_huffman_tree;

/* words needed for n input bytes in the worst case the ABI accepts (codes <= 32 bits) */
static inline uint64_t mi_huffman_bound_words(uint64_t n) { return n + 2; }

/* d_words must hold cap_words u32 (>= ceil(bits/32)+1).  Words [0, ceil(bits/32)) are fully defined
 * (unused low bits of the last one are 0, as after init_bitwriter's memset); words past that are not touched.
 * d_info / d_tree are device buffers of sizeof(mi_huffman_info) / sizeof(mi_huffman_tree). */
#define MI_HUFFMAN_TILE 32768u   /* input bytes per encoder tile (one sync point each) */
/* d_tile_off (optional, may be NULL): u64[ceil(n/MI_HUFFMAN_TILE)+1], bit offset at which each
 * tile's codes start — the only sync points a variable-length code has; the parallel decoder
 * needs them, the reference format has no place for them. */
mi_status mi_huffman_encode_dev(mi_ctx *ctx, const uint8_t *d_in, uint64_t n,
                                uint32_t *d_words, uint64_t cap_words,
                                mi_huffman_info *d_info, mi_huffman_tree *d_tree,
                                uint64_t *d_tile_off, void *stream);
/* host-buffer convenience: copies in, encodes, copies out, synchronises. h_words: cap_words u32. */
mi_status mi_huffman_encode(mi_ctx *ctx, const uint8_t *h_in, uint64_t n,
                            uint32_t *h_words, uint64_t cap_words,
                            mi_huffman_info *h_info, mi_huffman_tree *h_tree);
/* decode exactly n symbols with the tree arrays; replaces huffman.c:330-364.  With d_tile_off
 * (from the encoder) one lane per tile; with NULL a single lane walks the whole stream.
 * d_words needs one readable word past the stream.  Synchronises (returns MI_ERR_CORRUPT). */
mi_status mi_huffman_decode_dev(mi_ctx *ctx, const uint32_t *d_words, uint64_t total_bits,
                                const mi_huffman_tree *d_tree, uint32_t n_nodes,
                                const uint64_t *d_tile_off, uint8_t *d_out, uint64_t n, void *stream);

/* host-buffer decode (copies in/out, synchronises).  h_tile_off may be NULL (single-lane decode). */
mi_status mi_huffman_decode(mi_ctx *ctx, const uint32_t *h_words, uint64_t total_bits,
                            const mi_huffman_tree *h_tree, uint32_t n_nodes,
                            const uint64_t *h_tile_off, uint8_t *h_out, uint64_t n);
/* like mi_huffman_encode, also returning the tile offsets (h_tile_off: u64[ceil(n/MI_HUFFMAN_TILE)+1] or NULL) */
mi_status mi_huffman_encode2(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, uint32_t *h_words, uint64_t cap_words,
                             mi_huffman_info *h_info, mi_huffman_tree *h_tree, uint64_t *h_tile_off);

/* The same encoder in three steps, for ONE tree over a buffer spread over several GPUs (whole-buffer parity across
 * ranks: huffman.c:179-215 builds one tree over the whole buffer, :267-328 packs with it).  Per rank:
 *   mi_huffman_hist_dev   shard -> d_hist u64[256] (+ d_tile_hist u32[mi_huffman_num_tiles(n)][256], kept by the caller)
 *   -- all-reduce (sum) of d_hist over the ranks: 2 KiB --
 *   mi_huffman_build_dev  summed histogram (taken modulo 2^32 like the reference's u32 counters) -> tree, codes, status
 *   -- bits of a shard = sum(hist[s] * length[s]); an all-gather of those gives every shard its global bit offset --
 *   mi_huffman_encode_with_tree_dev  shard -> words; the stream starts bit_offset (= global offset mod 32) bits into
 *                         d_words[0]; d_info->total_bits = bit_offset + the shard's bits; d_tile_off (optional)
 *                         are offsets relative to d_words[0].  MI_ERR_ARG in d_info->status if the shard holds a
 *                         byte the tree has no code for.
 * Shard word ranges overlap by one word at a seam; OR-ing them there yields the single-GPU stream bit for bit
 * (compression_algorithms_amd/sharded.py does the exchange over torch.distributed / RCCL). */
uint64_t  mi_huffman_num_tiles(uint64_t n);
mi_status mi_huffman_hist_dev(mi_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist, uint32_t *d_tile_hist, void *stream);
mi_status mi_huffman_build_dev(mi_ctx *ctx, const uint64_t *d_hist, mi_huffman_info *d_info, mi_huffman_tree *d_tree, void *stream);
mi_status mi_huffman_encode_with_tree_dev(mi_ctx *ctx, const uint8_t *d_in, uint64_t n, const mi_huffman_tree *d_tree,
                                          const uint32_t *d_tile_hist, uint32_t bit_offset, uint32_t *d_words, uint64_t cap_words,
                                          mi_huffman_info *d_info, uint64_t *d_tile_off, void *stream);

/* host-buffer forms of the steps (the drop-in's build_huffman_tree and _huffman_compress, huffman.c:179-215, :267-285) */
mi_status mi_huffman_build(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, mi_huffman_info *h_info, mi_huffman_tree *h_tree);
mi_status mi_huffman_encode_with_codes(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, const uint32_t *h_codes,
                                       const uint8_t *h_lengths, uint32_t bit_offset, uint32_t *h_words, uint64_t cap_words,
                                       mi_huffman_info *h_info);

/* ------------------------------------------------------------------------------------
 * LZ77 greedy tokenisers, block-parallel.
 *   deflate flavour: algorithms/deflate/lz77.c:199-280 per block of `block` bytes with a
 *     FRESH table per block (the sharded parity definition); byte tokens
 *     {0,c} / {1,dlo,dhi,len} (lz77.c:176-197).
 *   lz77 flavour: algorithms/lz77/lz77.c:264-345 per block; LSB-first bit tokens
 *     1+8 / 1+wbits+lbits (lz77.c:290-330).
 * A block is encoded as if followed by zero bytes (the reference reads past `size`).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t wbits;       /* window bits: lz77 14 (shipped) or 16; deflate 15             */
    uint32_t lbits;       /* length bits: lz77 4; deflate 5                               */
    uint32_t tbits;       /* log2 table size: lz77 wbits+6; deflate 20                    */
    uint32_t deflate;     /* 1: deflate rules (insert probe wraps, literal iff p-m >= W-1, byte tokens) */
    uint32_t block;       /* block size in bytes: 1..65536; lz77 flavour also 65792..1048576 in steps of 256 —
                           * the HBM-resident finder of lzw.hip, where a 64 KiB window really slides (exact, slower) */
} mi_lz_params;

static inline mi_lz_params mi_lz_params_deflate(void) { mi_lz_params p = {15, 5, 20, 1, 65536}; return p; }
static inline mi_lz_params mi_lz_params_lz77(uint32_t wbits) { mi_lz_params p = {wbits, 4, wbits + 6, 0, 65536}; return p; }

static inline uint64_t mi_lz_num_blocks(uint64_t n, const mi_lz_params *p) { return (n + p->block - 1) / p->block; }
/* bound on the concatenated stream, in bytes.  Per block: every byte a literal (2 bytes / 9 bits), except that the
 * block's LAST token may be a match that covers a single real byte and runs on into the zero tail the reference
 * reads past `size`: that match costs 4 bytes (deflate) or 1+wbits+lbits bits (lz77) instead of
 * one literal — +2 bytes / +(wbits+lbits-8) bits per block. */
static inline uint64_t mi_lz_bound_bytes(uint64_t n, const mi_lz_params *p)
{
    const uint64_t nblocks = p->block ? (n + p->block - 1) / p->block : 0;
    if (p->deflate) return 2 * n + 2 * nblocks + 8;
    return (9 * n + nblocks * (uint64_t)(p->wbits + p->lbits - 8) + 7) / 8 + 16;
}

/*
 * Encode n bytes at d_in as ceil(n/block) independent blocks.
 *   d_out        concatenated stream: deflate flavour = byte tokens of block 0,1,2,...;
 *                lz77 flavour = the blocks' bit streams concatenated bit-contiguously
 *                (block b starts at bit d_block_bits_excl[b]); zero-filled by the call.
 *   d_block_bits u64[nblocks+1]: EXCLUSIVE prefix sum of per-block stream lengths in BITS
 *                (deflate: 8 * bytes); entry nblocks = total.
 */
mi_status mi_lz_encode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                           uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_block_bits, void *stream);
mi_status mi_lz_encode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                       uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits);
/* decode; every block is truncated at its original length (an overshooting last match, A.3.4).
 * stream_bytes = readable bytes at d_stream: the kernel never reads outside [d_stream, d_stream + stream_bytes) and
 * never outside a block's own bit range, whatever the table says.  Like every decoder of this ABI it synchronises
 * `stream` before returning (MI_ERR_CORRUPT is decided on the device). */
mi_status mi_lz_decode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                           const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, void *stream);

/* host buffers: the table is validated (mi_validate_block_table) before anything is copied; above one chunk (4 096
 * blocks) the stream goes up and the bytes come down chunk by chunk around the decoder.  On MI_ERR_CORRUPT h_out may hold
 * the chunks decoded before the bad block. */
mi_status mi_lz_decode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_stream, uint64_t stream_bytes,
                       const uint64_t *h_block_bits, uint8_t *h_out, uint64_t n);

/* debugging / parity hooks used by the tests: find() at every position of every block
 * (0xFFFF = none), i.e. the output of the match-finder stage alone. */
mi_status mi_lz_find_all_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                             uint16_t *d_cand, void *stream);
/* the same for blocks above 64 KiB (lz77 flavour only): 32-bit positions, 0xFFFFFFFF = none */
mi_status mi_lz_find_all32_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                               uint32_t *d_cand, void *stream);

/* The reference's first, brute-force parser — lz77_compress_old, algorithms/lz77/lz77.h:51-54, lz77.c:185-262 (its call is
 * commented out at lz77/main.c:26): the whole window of 2^wbits - 1 bytes is searched at every token start, first-longest
 * match wins, ONE stream over the whole buffer in lz77_compress's token format.  O(n * 2^wbits) by definition; on the GPU
 * the best match of every position is found independently (lz_old.hip).  d_out: mi_lz77_old_bound_bytes(n) bytes, 4-byte
 * aligned, zeroed by the call; *total_bits = the reference's bit_index (the stream is total_bits / 8 + 1 bytes, lz77.c:258).
 * wbits 8..16, lbits 3..5 (the reference: 14, 4).  mi_lz77_whole_decode* decodes a whole-buffer stream (this parser's, or
 * lz77_compress's for a buffer of one block): lz77.c:347-377 on one wave; n < 2^32. */
uint64_t  mi_lz77_old_bound_bytes(uint64_t n);
mi_status mi_lz77_old_encode_dev(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *d_in, uint64_t n,
                                 uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_total_bits, void *stream);
mi_status mi_lz77_old_encode(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *h_in, uint64_t n,
                             uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_total_bits);
mi_status mi_lz77_whole_decode_dev(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *d_stream, uint64_t stream_bytes,
                                   uint64_t total_bits, uint8_t *d_out, uint64_t n, void *stream);
mi_status mi_lz77_whole_decode(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *h_stream, uint64_t stream_bytes,
                               uint64_t total_bits, uint8_t *h_out, uint64_t n);

/* ------------------------------------------------------------------------------------
 * Deflate "mode H": the entropy stage algorithms/deflate/lz77.c:279 leaves as a TODO
 * ("Build huffman tree and encode compressed buffer").  The token sequence is the
 * reference's (same finder and parse as mi_lz_encode_dev with deflate = 1); each block's
 * tokens are then coded with a dynamic Huffman code over the reference's 286-symbol
 * alphabet (deflate/huffman.h:6, huffman.c:49-62), lengths from the reference's heap
 * procedure (algorithms/huffman/huffman.c:100-163), canonical code assignment, MSB-first
 * u32 packing (deflate/huffman.c:16-46).  The reference has no such encoder: the bit
 * stream is defined by this build (oracle/orc_defh.c restates it).
 *
 * Block record (4-byte aligned): u32 n_tokens | u8 len[286] + 2 pad | u32 words[]
 *   literal b -> code[b];  match (d, l) -> code[256 + clz16(d)], the 15 - clz16(d) offset
 *   bits below d's leading one, the 5-bit length.
 * d_block_bits u64[nblocks+1]: exclusive prefix of record lengths in BITS (multiples of 32).
 * p must be a deflate-flavour parameter set with lbits <= 5 and wbits <= 16.
 * ------------------------------------------------------------------------------------ */
/* worst case of the concatenated records: per block the 292-byte header plus 9 bits per byte (a Huffman code is never
 * longer than the fixed 9-bit code over 286 symbols) plus one overshooting last match (<= 20 extra bits), word aligned */
uint64_t  mi_deflate_h_bound_bytes(uint64_t n, const mi_lz_params *p);
mi_status mi_deflate_h_encode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                                  uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_block_bits, void *stream);
mi_status mi_deflate_h_decode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                                  const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, void *stream);
mi_status mi_deflate_h_encode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                              uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits);
mi_status mi_deflate_h_decode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_stream, uint64_t stream_bytes,
                              const uint64_t *h_block_bits, uint8_t *h_out, uint64_t n);

/* ------------------------------------------------------------------------------------
 * FSE / tANS, block-parallel (fse/src/main.zig — an unfinished sketch; the stream format is
 * defined by this build).  Record layout in include/mi_fse.h.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t table_log;   /* 8 (reference TABLE_LOG) .. 12                                 */
    uint32_t streams;     /* sub-streams per block, 1..64 (one GPU lane each)               */
    uint32_t spread;      /* 0: contiguous symbol ranges (main.zig:159-177), 1: stride spread */
    uint32_t block;       /* block size in bytes, 4..65536                                  */
} mi_fse_params;

static inline mi_fse_params mi_fse_params_default(void) { mi_fse_params p = {8, 64, 1, 65536}; return p; }
uint64_t  mi_fse_block_bound(const mi_fse_params *p);        /* bytes per block record, worst case */
/* d_out: nblocks records at stride mi_fse_block_bound(); d_sizes u32[nblocks] = used bytes.
 * d_packed (optional, may be NULL): the records concatenated; d_offsets u64[nblocks+1]. */
mi_status mi_fse_encode_dev(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *d_in, uint64_t n,
                            uint8_t *d_packed, uint64_t cap_bytes, uint64_t *d_offsets, void *stream);
mi_status mi_fse_decode_dev(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *d_packed, uint64_t packed_bytes,
                            const uint64_t *d_offsets, uint8_t *d_out, uint64_t n, void *stream);
/* host-buffer versions.  h_packed needs mi_fse_block_bound() * nblocks bytes; h_offsets u64[nblocks+1] (bits). */
mi_status mi_fse_encode(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *h_in, uint64_t n,
                        uint8_t *h_packed, uint64_t cap_bytes, uint64_t *h_offsets);
mi_status mi_fse_decode(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *h_packed, uint64_t packed_bytes,
                        const uint64_t *h_offsets, uint8_t *h_out, uint64_t n);
/* the normalisation step alone (main.zig:106-149), for parity tests: d_freq u64[256] -> d_cnt u32[256] */
mi_status mi_fse_normalise_dev(mi_ctx *ctx, const uint64_t *d_freq, uint32_t table_log, uint32_t *d_cnt, void *stream);

/* ------------------------------------------------------------------------------------
 * Several GPUs of one node from ONE process (BASELINE config 5 behind the reference's own API).
 *   The reference's block loop (algorithms/deflate/deflate.c:47-63) walks the file block by block; blocks are independent
 *   here (fresh table per block), so device g of `ndev` encodes the contiguous block range
 *   mi_multi_shard(nblocks, g, ndev) with its own context and the single-GPU pipeline, concurrently, with no data-path
 *   collective.  The only exchange is the assembly of ONE stream + ONE block table on device 0 ("RCCL gather of per-block
 *   compressed streams over xGMI"): the per-device sizes meet on the host (8 bytes each), then ONE group of point-to-point
 *   transfers — ncclGroupStart .. ncclSend / ncclRecv x (ndev - 1) .. ncclGroupEnd — moves streams and tables into device 0,
 *   every peer over its own xGMI link.  A shard that starts on a 32-bit boundary of the final stream (mode H always) is
 *   received in place; otherwise it lands in a staging area and one kernel shifts it in (the bit-packed lz77 flavour:
 *   result bit-contiguous, exactly what one GPU writes for the whole buffer).
 *   Transport: RCCL (librccl.so, loaded on first use — libmi_codec.so does not link it) when the listed devices are
 *   distinct; hipMemcpyPeerAsync when a device is listed more than once (two contexts on one GPU: how the path is tested
 *   on a one-GPU box) or when MI_MULTI_TRANSPORT=peer.  MI_MULTI_TRANSPORT=rccl insists on RCCL (MI_ERR_TRANSPORT if the
 *   device list has a duplicate or the library is missing).
 *   The result is byte-identical to the single-context entry points' (tests/test_multi_gpu.py).
 *   The drop-in compress() takes this path when MI_CODEC_DEVICES=0,1,... is set (include/mi_deflate.h).
 * ------------------------------------------------------------------------------------ */
typedef struct mi_multi mi_multi;
mi_status   mi_multi_create(mi_multi **out, const int *devices, int ndev);       /* 1 <= ndev <= 64 */
void        mi_multi_destroy(mi_multi *m);
int         mi_multi_ndev(const mi_multi *m);
mi_ctx     *mi_multi_ctx(mi_multi *m, int g);                                    /* device g's context (owned by m) */
const char *mi_multi_transport(const mi_multi *m);                               /* "rccl" | "peer-copy" */
const char *mi_multi_last_transport_error(const mi_multi *m);                    /* text of the last MI_ERR_TRANSPORT, or "" */
/* contiguous block range [*lo, *hi) of device g: ceil(nblocks / ndev) blocks each, the last ones possibly fewer or none
 * (the rule of compression_algorithms_amd/sharded.py shard_blocks) */
void        mi_multi_shard(uint64_t nblocks, int g, int ndev, uint64_t *lo, uint64_t *hi);
/* Shards resident: d_in[g] points at device g's shard (its block range of the n input bytes, on device g; NULL for an
 * empty range).  mode_h = 0: mi_lz_encode_dev's stream (either flavour, any block size it takes), 1: mi_deflate_h_encode_dev's.
 * d_out0 (cap_bytes >= the single-device bound for n, 4-byte aligned) and d_block_bits0 (u64[nblocks + 1]) live on
 * devices[0].  Returns when the assembled stream is complete (it synchronises every device's stream). */
mi_status   mi_lz_encode_multi_dev(mi_multi *m, const mi_lz_params *p, int mode_h, const uint8_t *const *d_in, uint64_t n,
                                   uint8_t *d_out0, uint64_t cap_bytes, uint64_t *d_block_bits0);
/* host buffers: shards go up to their devices side by side, the assembled stream comes down from device 0 */
mi_status   mi_lz_encode_multi(mi_multi *m, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                               uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits);
mi_status   mi_deflate_h_encode_multi(mi_multi *m, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                      uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits);
/* moves `bytes` pattern bytes from the LAST device to device 0 through the gather's transport and checks them there:
 * the one way to drive the RCCL entry points on a one-GPU box (ndev = 1: a send to self inside the group). */
mi_status   mi_multi_selftest_transport(mi_multi *m, uint64_t bytes);
/* what the LZ encoders of a context met since it was created: blocks the LDS-resident finder handed to the fallback
 * pipeline (a giant cluster), parts above 2 560 entries (k_lz2_find_wide).  Both 0 on text; a corpus that lives there
 * runs at the fallback's rate and would otherwise only show as a slow number.  Synchronises the device. */
mi_status   mi_lz_path_stats(mi_ctx *ctx, uint64_t *fallback_blocks, uint64_t *wide_parts);

/* ------------------------------------------------------------------------------------
 * timing of the last *_dev call's dominant kernel, measured with hipEvents on the stream
 * the kernels ran on (bench.py's roofline leg).  Enabled by mi_set_profiling(ctx, 1).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const char *name;
    double      ms;       /* average duration per launch                                   */
    uint64_t    launches;
    uint64_t    bytes;    /* algorithmic bytes the launches moved               */
} mi_kernel_time;
mi_status mi_set_profiling(mi_ctx *ctx, int on);
/* returns the number of entries written (<= cap); resets the accumulators */
int       mi_get_kernel_times(mi_ctx *ctx, mi_kernel_time *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
