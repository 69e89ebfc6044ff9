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
 * Behaviour.  The buffer is cut into MI_LZ77_BLOCK-byte blocks that are encoded
 * independently on the GPU, each exactly as the reference encodes a buffer of that size
 * (WINDOW_BITS / LENGTH_BITS / TABLE_SIZE as in lz77.h:6-8), and the block streams are
 * concatenated bit-contiguously.  A buffer of at most one block therefore yields the
 * reference's stream bit for bit; larger buffers yield a stream the reference's own
 * lz77_decompress still decodes whenever no block's last match overshoots (it never does on
 * text).  The per-block bit offsets that a parallel decoder needs ride
 * behind the stream bytes inside the same malloc'd `data` block.
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
#define MI_LZ77_BLOCK 65536u

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint8_t *data;
    uint64_t bit_index;
} BitStream;

uint64_t min(uint64_t a, uint64_t b);
uint64_t max(uint64_t a, uint64_t b);
char *read_input_buffer(const char *filename, uint64_t *size);
bool  check_buffer_equivalence(const char *buffer1, const char *buffer2, uint64_t size);

/* returns a malloc'd BitStream whose data is malloc'd; caller frees ->data then the struct (lz77/main.c:66-67) */
BitStream *lz77_compress(const char *buffer, uint64_t size);
/* `size` is the ORIGINAL length (the stream has no header); returns malloc(size) */
char *lz77_decompress(BitStream *compressed_stream, uint64_t size, uint64_t *decompressed_size);

/* run-time override of the window (14 or 16): lets one binary serve both reference builds */
void mi_lz77_set_window_bits(uint32_t wbits);

#ifdef __cplusplus
}
#endif
#endif
