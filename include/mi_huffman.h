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
 * buffer keeps every word; the encoder's tile offsets for the parallel decoder are kept out of band
 * by the library (keyed by the buffer pointer); mi_frame.h serialises stream + tree + offsets.
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

/* huffman.h:62-73 (huffman.c:80-163): the reference's array min-heap of Node pointers, host helpers with the reference's
 * semantics — strict `<` on frequency in both sifts (ties keep the earlier arrangement, the left child wins a tie against
 * the right), enqueue on a full queue and dequeue on an empty one print the reference's message and exit(1).  Building a
 * tree with them (leaves in symbol order, two dequeues per merge, first = left: huffman.c:189-211) yields the codes of
 * build_huffman_tree / the GPU's k_huff_build, whose tie-breaking this heap defines (tests/test_dropin.py). */
typedef struct PriorityQueue {
    Node   **nodes;
    uint64_t size;
    uint64_t capacity;
} PriorityQueue;
PriorityQueue *init_priority_queue(uint64_t capacity);
void  swap_nodes(Node **a, Node **b);
void  heapify_up(PriorityQueue *queue, uint64_t idx);
void  heapify_down(PriorityQueue *queue, uint64_t idx);
void  enqueue(PriorityQueue *queue, Node *node);
Node *dequeue(PriorityQueue *queue);
bool  is_empty(PriorityQueue *queue);

char *read_input_buffer(const char *filename, uint64_t *size);
/* huffman.c:9-15 / :18-48: host-side bit writer helpers (MSB-first into u32 words), same semantics */
void  init_bitwriter(BitWriter *writer, uint64_t buffer_size);
void  write_bits(BitWriter *writer, uint32_t bits, uint8_t length);
Node *init_node(uint8_t value, uint32_t frequency);                                     /* huffman.c:165-177 */
void  print_bit_string(uint8_t *buffer, uint64_t size);                                 /* huffman.c:50-59 */
void  print_codes(uint32_t *codes, uint8_t *code_lengths);                              /* huffman.c:252-265 */
/* huffman.c:179-215 on the GPU: histogram + heap-exact tree; *root receives a malloc'd tree of the reference's shape */
void  build_huffman_tree(char *buffer, uint64_t size, Node **root);
/* huffman.c:267-285 on the GPU: packs `buffer` with the GIVEN codes, appending to `writer` wherever it stands */
void  _huffman_compress(char *buffer, uint64_t size, uint32_t *codes, uint8_t *code_lengths, BitWriter *writer);
Node  huffman_compress(char *buffer, uint64_t size, BitWriter *writer);
/* *output_size carries the ORIGINAL length on entry (huffman/main.c:69) and the decoded count on return */
void  huffman_decompress(BitWriter *writer, Node *root, char *output, uint64_t *output_size);
void  gather_codes(Node *root, uint32_t code, uint32_t length, uint32_t *codes, uint8_t *code_lengths);
/* huffman.h:108-113 (huffman.c:366-401): the lookup-table decoder the reference left unfinished; the GPU decoder is one,
 * so this is huffman_decompress under its second name */
void  huffman_decompress_lookup_table(BitWriter *writer, Node *root, char *output, uint64_t *output_size);
/* extension: frees writer->buffer and the tile-offset table registered for it (in-process registry, as in mi_lz77.h) */
void  mi_huffman_release(BitWriter *writer);

/* Extensions (no counterpart in huffman.h): the same codec to and from a self-describing FILE (mi_frame.h: serialised
 * tree + {last_block, size} chunks, after zig_huffman/src/main.zig:155-176,513-530), so that decoding needs neither the
 * in-memory Node tree nor the original size.  Return 0; errors follow the reference's printf + exit(1) convention. */
int huffman_compress_file(const char *input_filename, const char *output_filename);
int huffman_decompress_file(const char *input_filename, const char *output_filename);

#ifdef __cplusplus
}
#endif
#endif
