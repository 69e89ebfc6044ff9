/*
 * mi_deflate.h — drop-in for algorithms/deflate/deflate.h + lz77.h (libmi_deflate.so).
 *
 *   StateData, compress, decompress   algorithms/deflate/deflate.h:23-30 (deflate.c:10-79)
 *   per-block lz77_compress           algorithms/deflate/lz77.h:47-53    (lz77.c:199-280)
 *   BUFFER_SIZE                       algorithms/deflate/deflate.h:8
 *
 * compress() reads the file, tokenises its 65 536-byte blocks on the GPU — each block with a
 * FRESH table, the block-independent definition (the shipped loop keeps one
 * table across blocks, which only makes its output larger) — and writes "<basename>.deflate"
 * in the current directory: the token bytes, exactly what the reference's per-block
 * lz77_compress emits.  A side-car "<basename>.deflate.idx" (original size, block size,
 * per-block byte offsets) makes the stream decodable; the reference's decompress() is empty.
 *
 * MI_DEFLATE_MODE=H in the environment makes compress() finish the stage the reference leaves as
 * "// TODO: Build huffman tree and encode compressed buffer" (deflate/lz77.c:279): the same
 * tokens, Huffman coded per block over the reference's 286-symbol alphabet (mi_codec.h, "mode
 * H").  The side-car records the mode, so decompress() needs no switch.
 */
#ifndef MI_DEFLATE_H
#define MI_DEFLATE_H
#include <stdint.h>
#include <stdbool.h>

#define BUFFER_SIZE 65536
#define MAX_WINDOW_BITS 15
#define MAX_LENGTH_BITS 5

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HashTableArray HashTableArray;     /* opaque: the GPU path never materialises the table */
typedef struct HuffmanNode HuffmanNode;

typedef struct StateData {
    HashTableArray *table;              /* always NULL (the reference returns a dangling pointer here) */
    HuffmanNode    *huffman_root;       /* NULL, as in the reference */
    char           *compressed_filename;/* malloc'd; caller frees */
} StateData;

StateData compress(const char *input_filename);
void      decompress(StateData *state_data, const char *input_filename);

/* one block (<= 65 536 bytes), fresh table; `table` is accepted for source compatibility and ignored */
void lz77_compress(const char *input_buffer, uint64_t input_buffer_size, char *compressed_buffer,
                   uint64_t *compressed_buffer_size, HashTableArray *table);

#ifdef __cplusplus
}
#endif
#endif
