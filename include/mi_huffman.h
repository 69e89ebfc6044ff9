/*
 * mi_huffman.h — drop-in for algorithms/huffman/huffman.h (libmi_huffman.so).
 *
 *   BitWriter            algorithms/huffman/huffman.h:42-47
 *   Node                 algorithms/huffman/huffman.h:54-60
 *   huffman_compress     algorithms/huffman/huffman.h:97-101  (huffman.c:288-328)
 *   huffman_decompress   algorithms/huffman/huffman.h:102-107 (huffman.c:330-364)
 *   read_input_buffer    algorithms/huffman/huffman.h:77      (used by huffman/main.c)
 *
 * huffman_compress returns the root by value with malloc'd children, exactly the shape the
 * reference builds (huffman_decompress walks it), and fills the BitWriter like the reference:
 * buffer (malloc'd), word_idx, bit_idx, buffer_size per huffman.c:318-320.  Unlike the
 * reference's realloc (which can cut live bytes of the last partial word) the
 * buffer keeps every word, followed by the encoder's tile offsets for the parallel decoder.
 * Errors the reference reports with printf + exit(1) are reported the same way.
 */
#ifndef MI_HUFFMAN_H
#define MI_HUFFMAN_H
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t *buffer;
    uint64_t  bit_idx;
    uint64_t  word_idx;
    uint64_t  buffer_size;
} BitWriter;

typedef struct Node Node;
struct Node {
    uint8_t  value;
    uint32_t frequency;
    Node    *left;
    Node    *right;
};

char *read_input_buffer(const char *filename, uint64_t *size);
Node  huffman_compress(char *buffer, uint64_t size, BitWriter *writer);
/* *output_size carries the ORIGINAL length on entry (huffman/main.c:69) and the decoded count on return */
void  huffman_decompress(BitWriter *writer, Node *root, char *output, uint64_t *output_size);
void  gather_codes(Node *root, uint32_t code, uint32_t length, uint32_t *codes, uint8_t *code_lengths);

#ifdef __cplusplus
}
#endif
#endif
