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

#define WINDOW_SIZE (1 << MAX_WINDOW_BITS)
#define TABLE_SIZE  (1 << (MAX_WINDOW_BITS + 5))

/* algorithms/deflate/lz77.h:10-28, same layouts: a caller written like deflate.c:13-14 (`HashTableArray table;
 * init_hash_table(&table);`) compiles and runs.  The GPU path never materialises the table: lz77_compress
 * accepts it and neither reads nor updates it — every call encodes its block against a FRESH table. */
typedef struct ArrayNode {
    uint32_t pattern;
    uint64_t index;
    bool     is_set;
} ArrayNode;

typedef struct Buckets {
    uint32_t *patterns;
    uint64_t *indices;
    bool     *is_set;
} Buckets;

typedef struct {
    Buckets  buckets;
    uint32_t bucket_indices[1 << MAX_WINDOW_BITS];
    uint32_t current_idx;
    bool     is_full;
} HashTableArray;

/* algorithms/deflate/deflate.h:12-17 */
typedef struct HuffmanNode {
    struct HuffmanNode *left;
    struct HuffmanNode *right;
    uint16_t value;
    uint64_t frequency;
} HuffmanNode;

typedef struct StateData {
    HashTableArray *table;              /* always NULL (the reference returns a dangling pointer here) */
    HuffmanNode    *huffman_root;       /* NULL, as in the reference */
    char           *compressed_filename;/* malloc'd; caller frees */
} StateData;

StateData compress(const char *input_filename);
void      decompress(StateData *state_data, const char *input_filename);

uint64_t min(uint64_t a, uint64_t b);
uint64_t max(uint64_t a, uint64_t b);
uint32_t hash(uint32_t pattern);                                  /* deflate/lz77.c:14-42 */
void     init_hash_table(HashTableArray *table);                  /* deflate/lz77.c:44-67: allocates and zeroes, as the reference */
/* deflate/lz77.h:32-33, deflate/lz77.c:77-174: per-entry operations on the host table above — host helpers (the GPU path
 * has no table; mi_lz_find_all_dev gives find() for every position of a buffer).  insert wraps modulo TABLE_SIZE, find
 * stops at the last bucket (the reference reads one element past the array there). */
void     insert_hash_table(HashTableArray *table, uint32_t pattern, uint64_t index);
uint64_t find(HashTableArray *table, uint32_t pattern);
/* NOT exported: lz77_decompress (deflate/lz77.h:54-59): the reference's body discards its output
 * (deflate/lz77.c:282-311) — use decompress() or mi_lz_decode. */
void write_literal(char *buffer, char c, uint64_t *buffer_index);                                   /* deflate/lz77.c:176-184 */
void write_length_distance(char *buffer, uint8_t length, uint16_t distance, uint64_t *buffer_index);   /* deflate/lz77.c:186-197 */

/* algorithms/deflate/huffman.h (the entropy stage the reference sketches and never calls; deflate/huffman.c:16-97) and
 * deflate.h:12-21.  Host helpers with the reference's semantics: the 286-bin tally is what k_lz_parse_emit counts on the GPU
 * in mode H (tests/test_oracle_defh.py: equal to these), write_bits is the MSB-first u32 packer k_defh_encode follows.
 * push_heap / pop_heap / build_huffman_tree / new_node are declared by the reference header but have no body anywhere in
 * the reference: nothing to mirror. */
#define NUM_CODES 286
typedef struct MinHeapNode {
    uint8_t  data;                      /* (the reference's type: symbols 256..285 do not fit — one reason the stage is unfinished) */
    uint32_t frequency;
    struct MinHeapNode *left;
    struct MinHeapNode *right;
} MinHeapNode;
typedef struct {
    uint32_t *buffer;
    uint64_t  bit_idx;
    uint64_t  word_idx;
    uint64_t  buffer_size;
} BitWriter;
void init_bitwriter(BitWriter *writer, uint64_t buffer_size);                                  /* deflate/huffman.c:7-13 */
void write_bits(BitWriter *writer, uint32_t bits, uint8_t length);                             /* deflate/huffman.c:16-46 */
void append_huffman_tree_literal(uint32_t *frequencies, char literal);                         /* deflate/huffman.c:49-54 */
void append_huffman_tree_pair(uint32_t *frequencies, uint16_t offset);                         /* deflate/huffman.c:56-62: bin 256 + clz16(offset) */
void gather_codes(MinHeapNode *root, uint16_t code, uint8_t length, uint16_t *codes, uint8_t *code_lengths);   /* deflate/huffman.c:64-97 */
void init_huffman_node(HuffmanNode *node);                                                     /* deflate.c:81-86 */
void destroy_huffman_node(HuffmanNode *node);                                                  /* deflate.c:88-98: frees the subtrees, not the node */
bool compare_huffman_node(const HuffmanNode *a, const HuffmanNode *b);                         /* deflate.c:100-102 */

/* one block (<= 65 536 bytes), fresh table; `table` is accepted for source compatibility and ignored */
void lz77_compress(const char *input_buffer, uint64_t input_buffer_size, char *compressed_buffer,
                   uint64_t *compressed_buffer_size, HashTableArray *table);

#ifdef __cplusplus
}
#endif
#endif
