/*
 * mi_frame.h — self-describing container for the streams of libmi_codec.so.
 *
 * The reference's C codecs hand their streams over in memory only: huffman_decompress needs the in-memory Node tree
 * (algorithms/huffman/huffman.h:102-107), lz77_decompress the original size (lz77.h:59-63), and `<name>.deflate` is a
 * bare token concatenation (deflate/deflate.c:56).  Its second, Zig-only Huffman program shows the intended file form:
 * the tree serialised in pre-order, then chunks of {last_block:1, size:31} + payload
 * (algorithms/huffman/zig_huffman/src/main.zig:11-18 CompressedSize, :155-176 serializeHuffmanTree, :513-530 flushChunk).
 * This container follows that shape for every codec here, so a stream can be written to disk and decoded without any
 * side information.  Pure host code (plain C): it moves bytes that the encoders produced, it does not encode.
 *
 * Layout (little endian):
 *   header, 32 bytes:  "MIFR" | u8 version = 1 | u8 codec | u16 0 | u32 block | u32 p0 | u32 p1 | u64 original_size | u32 0
 *   MI_FRAME_HUFFMAN   u64 total_bits | tree in pre-order: per node u8 value, u32 frequency; i32 -1 for an absent child
 *                      (main.zig:155-176) | u32 ntiles | u64 tile_off[ntiles+1] (0 tiles: none) |
 *                      chunks of u32 {last_block:1 (LSB), size:31} + size payload bytes (the u32 words as stored)
 *   block codecs       per block: u32 {last_block:1, size:31} | [MI_FRAME_LZ77 only: u32 bits in this block] | payload
 *                      (block b's stream, padded to a byte; an empty input has one empty last chunk)
 * Parameters: lz77 p0 = wbits, p1 = lbits; deflate T/H p0 = 15, p1 = 5; FSE p0 = table_log, p1 = streams | spread << 16;
 * Huffman p0 = n_nodes.
 */
#ifndef MI_FRAME_H
#define MI_FRAME_H
#include "mi_codec.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MI_FRAME_MAGIC      "MIFR"
#define MI_FRAME_HEADER     32u
#define MI_FRAME_HUFFMAN    1u     /* whole-buffer Huffman (mi_huffman_encode*)                  */
#define MI_FRAME_DEFLATE_T  2u     /* deflate byte tokens, the reference's stream (mi_lz_encode*) */
#define MI_FRAME_DEFLATE_H  3u     /* deflate mode H records (mi_deflate_h_encode*)               */
#define MI_FRAME_LZ77       4u     /* bit-packed lz77 blocks (mi_lz_encode*, deflate = 0)         */
#define MI_FRAME_FSE        5u     /* FSE block records (mi_fse_encode*)                          */

typedef struct {
    uint32_t codec, block, p0, p1;
    uint64_t original_size;
    uint64_t nblocks;           /* block codecs: chunks in the frame; Huffman: tiles in the table             */
    uint64_t stream_bytes;      /* payload bytes: size of the buffer mi_frame_unpack_* needs for the stream   */
    uint64_t total_bits;        /* Huffman: bits of the stream; block codecs: bits of the re-concatenated one */
} mi_frame_info;

/* ---- block codecs: stream + block table (exclusive prefix in bits, nblocks+1 entries) <-> frame ---------------- */
uint64_t  mi_frame_bound_blocks(uint64_t nblocks, uint64_t stream_bytes);
mi_status mi_frame_pack_blocks(uint32_t codec, uint32_t block, uint32_t p0, uint32_t p1, uint64_t original_size,
                               const uint8_t *h_stream, const uint64_t *h_block_bits, uint64_t nblocks,
                               uint8_t *out, uint64_t cap, uint64_t *out_bytes);
/* walks and VALIDATES the whole frame (every size against frame_bytes); MI_ERR_CORRUPT on any inconsistency */
mi_status mi_frame_parse(const uint8_t *frame, uint64_t frame_bytes, mi_frame_info *info);
/* rebuilds the contiguous stream (bit-contiguous for MI_FRAME_LZ77) and its table: what mi_*_decode takes.
 * h_stream needs info.stream_bytes + 8 bytes, h_block_bits info.nblocks + 1 entries. */
mi_status mi_frame_unpack_blocks(const uint8_t *frame, uint64_t frame_bytes, uint8_t *h_stream, uint64_t cap_bytes,
                                 uint64_t *h_block_bits, uint64_t cap_blocks);

/* ---- whole-buffer Huffman: words + tree + tile offsets <-> frame -------------------------------------------------- */
uint64_t  mi_frame_bound_huffman(uint64_t total_bits, uint64_t ntiles);
mi_status mi_frame_pack_huffman(uint64_t original_size, const mi_huffman_tree *tree, uint32_t n_nodes,
                                const uint32_t *h_words, uint64_t total_bits, const uint64_t *h_tile_off, uint64_t ntiles,
                                uint8_t *out, uint64_t cap, uint64_t *out_bytes);
/* tree arrays come back in the ABI's form (root = *n_nodes - 1; codes and lengths re-derived from the paths, left = 0,
 * right = 1 as huffman.c:217-250); h_words needs (total_bits + 31) / 32 + 1 entries (the last one zero), h_tile_off
 * info.nblocks + 1 (pass NULL / 0 to skip the table). */
mi_status mi_frame_unpack_huffman(const uint8_t *frame, uint64_t frame_bytes, mi_huffman_tree *tree, uint32_t *n_nodes,
                                  uint32_t *h_words, uint64_t cap_words, uint64_t *total_bits,
                                  uint64_t *h_tile_off, uint64_t cap_tiles);

#ifdef __cplusplus
}
#endif
#endif
