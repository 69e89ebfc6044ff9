/*
 * mi_lz77.h — drop-in for algorithms/lz77/lz77.h (libmi_lz77.so).
 *
 * Same names, struct layout and ownership rules as the reference header:
 *   BitStream                       algorithms/lz77/lz77.h:14-17
 *   lz77_compress                   algorithms/lz77/lz77.h:55-58   (lz77.c:264-345)
 *   lz77_decompress                 algorithms/lz77/lz77.h:59-63   (lz77.c:347-377)
 *   check_buffer_equivalence        algorithms/lz77/lz77.h:45-49   (lz77.c:379-392)
 *   read_input_buffer, min          algorithms/lz77/lz77.h:11,37   (used by lz77/main.c)
 *
 * Behaviour.  The buffer is cut into MI_LZ77_BLOCK-byte blocks (64 KiB by default; up to 1 MiB through the environment
 * variable of the same name — the window then slides inside a block as in the reference, on a slower, HBM-resident
 * finder) that are encoded
 * independently on the GPU, each exactly as the reference encodes a buffer of that size
 * (WINDOW_BITS / LENGTH_BITS / TABLE_SIZE as in lz77.h:6-8), and the block streams are
 * concatenated bit-contiguously.  A buffer of at most one block therefore yields the
 * reference's stream bit for bit; larger buffers yield a stream the reference's own
 * lz77_decompress still decodes whenever no block's last match overshoots (it never does on
 * text).  The per-block bit offsets that a parallel decoder needs are kept
 * out of band by the library (keyed by the `data` pointer); mi_frame.h serialises both.
 */
#ifndef MI_LZ77_H
#define MI_LZ77_H
#include <stdint.h>
#include <stdbool.h>

#ifndef LENGTH_BITS
#define LENGTH_BITS 4
#endif
#ifndef WINDOW_BITS
#define WINDOW_BITS 14          /* override with -DWINDOW_BITS=16 for the 64 KiB window; or MI_LZ77_WINDOW_BITS at run time */
#endif
#define TABLE_SIZE (1 << (WINDOW_BITS + 6))
#define MI_LZ77_BLOCK 65536u      /* default; MI_LZ77_BLOCK=<bytes> at run time: up to 1048576 (multiples of 256 above 65536) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint8_t *data;
    uint64_t bit_index;
} BitStream;

/* algorithms/lz77/lz77.h:19-30, same layout: reference-style callers (`HashTableArray t; init_hash_table(&t);`) compile
 * and run.  The GPU path never materialises this table: insert_hash_table / find below work on it on the host. */
typedef struct ArrayNode {
    uint32_t pattern;
    uint64_t index;
    bool     is_set;
} ArrayNode;

typedef struct {
    ArrayNode *buckets;
    uint32_t   bucket_indices[1 << WINDOW_BITS];
    uint32_t   current_idx;
    bool       is_full;
} HashTableArray;

uint64_t min(uint64_t a, uint64_t b);
uint64_t max(uint64_t a, uint64_t b);
/* lz77.h:32, lz77.c:13-41: the host twin of the device constant every kernel hashes with (lz_common.h lz_mix32) */
uint32_t hash(uint32_t pattern);
/* lz77.h:33, lz77.c:43-53: allocates and zeroes the table exactly like the reference */
void init_hash_table(HashTableArray *table);
/* lz77.h:34-35, lz77.c:55-108: per-entry operations on the host table above — host helpers (the encoder's table is never
 * materialised; mi_lz_find_all_dev gives find() for every position of a buffer on the GPU).  Unlike the
 * reference they stop at the last bucket and exit(1) where the reference's unbounded probe would write past the array. */
void     insert_hash_table(HashTableArray *table, uint32_t pattern, uint64_t index);
uint64_t find(HashTableArray *table, uint32_t pattern);
/* lz77.h:51-54, lz77.c:185-262: the reference's first, brute-force parser (its call is commented out at lz77/main.c:26) —
 * a different stream from lz77_compress: longest match of the whole window at every token, first-longest wins, one stream
 * over the whole buffer.  On the GPU: lz_old.hip (mi_lz77_old_encode).  Same ownership as lz77_compress; lz77_decompress
 * decodes it (the stream is registered as a whole-buffer stream). */
BitStream *lz77_compress_old(const char *buffer, uint64_t size);
void  print_bit_string(const char *buffer, uint64_t size);
char *read_input_buffer(const char *filename, uint64_t *size);
bool  check_buffer_equivalence(const char *buffer1, const char *buffer2, uint64_t size);
/* lz77.h:39-44, lz77.c:139-184: LSB-first bit I/O on a caller's buffer (host helpers, not on the hot path) */
void     init_bitstream(BitStream *stream, uint8_t *buffer);
void     write_bit(BitStream *stream, bool bit);
bool     read_bit(BitStream *stream);
void     write_bits(BitStream *stream, uint64_t value, uint64_t num_bits);
uint64_t read_bits(BitStream *stream, uint64_t num_bits);

/* returns a malloc'd BitStream whose data is malloc'd; caller frees ->data then the struct (lz77/main.c:66-67) */
BitStream *lz77_compress(const char *buffer, uint64_t size);
/* `size` is the ORIGINAL length (the stream has no header); returns malloc(size) */
char *lz77_decompress(BitStream *compressed_stream, uint64_t size, uint64_t *decompressed_size);

/* Extensions.  A stream of more than one block keeps its per-block bit offsets in a registry of this library (keyed by the
 * `data` pointer, in-process only, grows as needed, thread-safe).  mi_lz77_release frees the stream AND its entry; callers
 * that free ->data and the struct by hand (lz77/main.c:66-67) leave the entry until the address is registered again.
 * lz77_decompress on a multi-block stream without an entry prints why and exit(1)s (the reference's error convention). */
void     mi_lz77_release(BitStream *stream);
uint64_t mi_lz77_registered_streams(void);
/* run-time override of the window (14 or 16): lets one binary serve both reference builds */
void mi_lz77_set_window_bits(uint32_t wbits);

#ifdef __cplusplus
}
#endif
#endif
