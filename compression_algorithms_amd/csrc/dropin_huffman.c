/* dropin_huffman.c — algorithms/huffman entry points over the HIP path (see include/mi_huffman.h). */
#include <string.h>
#include "../../include/mi_huffman.h"
#include "../../include/mi_frame.h"
#include "dropin_common.h"

char *read_input_buffer(const char *filename, uint64_t *size)      /* huffman.c:61-78 */
{
    FILE *f = fopen(filename, "rb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", filename); exit(1); }
    fseek(f, 0, SEEK_END); *size = (uint64_t)ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc(*size + 1);
    if (fread(buf, 1, *size, f) != *size) { fprintf(stderr, "Error: short read on %s\n", filename); exit(1); }
    fclose(f);
    return buf;
}

void gather_codes(Node *root, uint32_t code, uint32_t length, uint32_t *codes, uint8_t *lens)   /* huffman.c:217-250 */
{
    if (!root->left && !root->right) { codes[root->value] = code; lens[root->value] = (uint8_t)length; return; }
    code <<= 1;
    if (root->left) gather_codes(root->left, code, length + 1, codes, lens);
    if (root->right) gather_codes(root->right, code + 1, length + 1, codes, lens);
}

void init_bitwriter(BitWriter *writer, uint64_t buffer_size)       /* huffman.c:9-15 */
{
    writer->buffer = (uint32_t *)calloc(buffer_size ? buffer_size : 1, 1);
    writer->word_idx = 0; writer->bit_idx = 0; writer->buffer_size = buffer_size;
}

void write_bits(BitWriter *writer, uint32_t bits, uint8_t length)   /* huffman.c:18-48: MSB first, spill into the next word */
{
    if (!length) return;
    const uint32_t room = 32u - (uint32_t)writer->bit_idx;
    if (length <= room) {
        writer->buffer[writer->word_idx] |= (length == 32 ? bits : (bits & ((1u << length) - 1u))) << (room - length);
        writer->bit_idx += length;
        if (writer->bit_idx == 32) { writer->bit_idx = 0; ++writer->word_idx; }
    } else {
        const uint32_t spill = length - room;
        writer->buffer[writer->word_idx] |= (bits >> spill) & (room == 32 ? 0xFFFFFFFFu : ((1u << room) - 1u));
        ++writer->word_idx;
        writer->buffer[writer->word_idx] |= bits << (32u - spill);
        writer->bit_idx = spill;
    }
}

/* ---- huffman.h:62-73: the reference's priority queue (huffman.c:80-163), host side ---------------------------------------
 * An implicit binary min-heap over Node pointers keyed by frequency.  Only the comparisons matter for parity: both sifts
 * move an element only past a STRICTLY smaller / larger key, and sifting down looks at the left child first, so on a tie
 * between the children the left one moves up.  The GPU's k_huff_build performs these same operations in this order. */
PriorityQueue *init_priority_queue(uint64_t capacity)
{
    PriorityQueue *q = (PriorityQueue *)malloc(sizeof *q);
    if (!q) { printf("ERROR: out of memory\n"); exit(1); }
    q->nodes = (Node **)malloc((capacity ? capacity : 1) * sizeof(Node *));
    q->size = 0;
    q->capacity = capacity;
    return q;
}

void swap_nodes(Node **a, Node **b) { Node *t = *a; *a = *b; *b = t; }

void heapify_up(PriorityQueue *q, uint64_t idx)
{
    for (; idx > 0; idx = (idx - 1) / 2) {
        Node **child = &q->nodes[idx], **parent = &q->nodes[(idx - 1) / 2];
        if (!((*child)->frequency < (*parent)->frequency)) return;
        swap_nodes(child, parent);
    }
}

void heapify_down(PriorityQueue *q, uint64_t idx)
{
    for (;;) {
        uint64_t pick = idx;
        const uint64_t l = 2 * idx + 1, r = l + 1;
        if (l < q->size && q->nodes[l]->frequency < q->nodes[pick]->frequency) pick = l;
        if (r < q->size && q->nodes[r]->frequency < q->nodes[pick]->frequency) pick = r;
        if (pick == idx) return;
        swap_nodes(&q->nodes[idx], &q->nodes[pick]);
        idx = pick;
    }
}

void enqueue(PriorityQueue *q, Node *node)
{
    if (q->size == q->capacity) { printf("ERROR: Queue is full\n"); exit(1); }       /* huffman.c:138-141 */
    q->nodes[q->size] = node;
    heapify_up(q, q->size++);
}

Node *dequeue(PriorityQueue *q)
{
    if (q->size == 0) { printf("ERROR: Queue is empty\n"); exit(1); }               /* huffman.c:149-152 */
    Node *top = q->nodes[0];
    q->nodes[0] = q->nodes[--q->size];
    heapify_down(q, 0);
    return top;
}

bool is_empty(PriorityQueue *q) { return q->size == 0; }

Node *init_node(uint8_t value, uint32_t frequency)                  /* huffman.c:165-177 */
{
    Node *n = (Node *)malloc(sizeof *n);
    n->value = value; n->frequency = frequency; n->left = NULL; n->right = NULL;
    return n;
}

void print_bit_string(uint8_t *buffer, uint64_t size)               /* huffman.c:50-59 */
{
    for (uint64_t i = 0; i < size; ++i) {
        for (int bit = 7; bit >= 0; --bit) printf("%d", (buffer[i] >> bit) & 1);
        printf(" ");
    }
    printf("\n");
}

void print_codes(uint32_t *codes, uint8_t *lens)                    /* huffman.c:252-265 */
{
    for (uint32_t s = 0; s < 256; ++s) {
        if (!lens[s]) continue;
        printf("%c: ", (char)s);
        for (uint32_t b = 0; b < lens[s]; ++b) printf("%d", (codes[s] >> (lens[s] - b - 1)) & 1);
        printf("\n");
    }
}

static Node *build_node(const mi_huffman_tree *t, int id)
{
    Node *n = (Node *)malloc(sizeof *n);
    n->value = t->value[id]; n->frequency = t->frequency[id];
    n->left = t->left[id] >= 0 ? build_node(t, t->left[id]) : NULL;
    n->right = t->right[id] >= 0 ? build_node(t, t->right[id]) : NULL;
    return n;
}

/* writer->buffer holds the words alone (every word kept, plus one zero word); the encoder's tile offsets — the sync
 * points the parallel decoder needs — are registered out of band (dropin_common.h) under the buffer pointer. */
Node huffman_compress(char *buffer, uint64_t size, BitWriter *writer)
{
    mi_ctx *ctx = dropin_ctx();
    const uint64_t cap = mi_huffman_bound_words(size);
    const uint64_t ntiles = (size + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE;
    uint32_t *words = (uint32_t *)calloc(cap + 4, 4);
    uint64_t *toff = (uint64_t *)malloc(8 * (ntiles + 1));
    mi_huffman_info info; mi_huffman_tree tree;
    if (!words || !toff) { fprintf(stderr, "huffman_compress: out of memory\n"); exit(1); }
    mi_status st = mi_huffman_encode2(ctx, (const uint8_t *)buffer, size, words, cap, &info, &tree, toff);
    if (st == MI_ERR_EMPTY_INPUT) { printf("ERROR: Queue is empty\n"); exit(1); }                      /* huffman.c:149-152 */
    if (st == MI_ERR_SINGLE_SYMBOL) { printf("ERROR: No code for character %c\n", buffer[0]); exit(1); }   /* huffman.c:278-281 */
    if (st != MI_OK) { fprintf(stderr, "huffman_compress: %s\n", mi_status_str(st)); exit(1); }
    writer->word_idx = info.word_idx; writer->bit_idx = info.bit_idx; writer->buffer_size = info.buffer_size;
    const uint64_t nw = info.word_idx + (info.bit_idx > 0);
    writer->buffer = (uint32_t *)realloc(words, (nw + 1) * 4);
    if (!writer->buffer) { fprintf(stderr, "huffman_compress: out of memory\n"); exit(1); }
    dropin_side_put(writer->buffer, info.total_bits, size, ntiles, toff, ntiles + 1);
    free(toff);
    Node *root = build_node(&tree, (int)info.n_nodes - 1);
    Node r = *root;
    free(root);                         /* the reference leaks it; the copy carries the children */
    return r;
}

void build_huffman_tree(char *buffer, uint64_t size, Node **root)  /* huffman.c:179-215: histogram + heap merge, on the GPU */
{
    mi_huffman_info info; mi_huffman_tree tree;
    mi_status st = mi_huffman_build(dropin_ctx(), (const uint8_t *)buffer, size, &info, &tree);
    if (st == MI_ERR_EMPTY_INPUT) { printf("ERROR: Queue is empty\n"); exit(1); }                      /* huffman.c:149-152 */
    if (st != MI_OK && st != MI_ERR_SINGLE_SYMBOL && st != MI_ERR_CODE_TOO_LONG) { fprintf(stderr, "build_huffman_tree: %s\n", mi_status_str(st)); exit(1); }
    *root = build_node(&tree, (int)info.n_nodes - 1);            /* a single-symbol input yields a lone leaf, as in the reference */
}

void _huffman_compress(char *buffer, uint64_t size, uint32_t *codes, uint8_t *code_lengths, BitWriter *writer)   /* huffman.c:267-285 */
{
    const uint64_t cap = mi_huffman_bound_words(size) + 2;
    uint32_t *tmp = (uint32_t *)calloc(cap, 4);
    mi_huffman_info info;
    if (!tmp) { fprintf(stderr, "_huffman_compress: out of memory\n"); exit(1); }
    mi_status st = mi_huffman_encode_with_codes(dropin_ctx(), (const uint8_t *)buffer, size, codes, code_lengths,
                                                (uint32_t)writer->bit_idx, tmp, cap, &info);
    if (st == MI_ERR_ARG && size) { printf("ERROR: No code for character\n"); exit(1); }              /* huffman.c:278-281 */
    if (st != MI_OK) { fprintf(stderr, "_huffman_compress: %s\n", mi_status_str(st)); exit(1); }
    const uint64_t nw = (info.total_bits + 31) / 32;
    for (uint64_t i = 0; i < nw; ++i) writer->buffer[writer->word_idx + i] |= tmp[i];    /* word 0 may already hold bits */
    writer->word_idx += info.total_bits / 32;
    writer->bit_idx = info.total_bits % 32;
    free(tmp);
}

static int flatten(const Node *n, mi_huffman_tree *t, int *next)
{
    int l = -1, r = -1;
    if (n->left) l = flatten(n->left, t, next);
    if (n->right) r = flatten(n->right, t, next);
    int id = (*next)++;
    t->value[id] = n->value; t->frequency[id] = n->frequency; t->left[id] = (int16_t)l; t->right[id] = (int16_t)r;
    return id;                          /* post-order: the root gets the last id */
}

void huffman_decompress(BitWriter *writer, Node *root, char *output, uint64_t *output_size)
{
    mi_ctx *ctx = dropin_ctx();
    mi_huffman_tree tree;
    memset(&tree, 0, sizeof tree);
    for (int i = 0; i < 511; ++i) tree.left[i] = tree.right[i] = -1;
    int next = 0;
    flatten(root, &tree, &next);
    uint32_t codes[256] = {0}; uint8_t lens[256] = {0};
    gather_codes(root, 0, 0, codes, lens);
    memcpy(tree.code, codes, sizeof codes); memcpy(tree.length, lens, sizeof lens);
    const uint64_t n = *output_size;                            /* the original length (huffman/main.c:69) */
    const uint64_t bits = writer->word_idx * 32 + writer->bit_idx;
    dropin_side e;
    const int have = dropin_side_get(writer->buffer, bits, n, &e);
    const uint64_t *toff = (have && e.count == (n + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE + 1) ? e.table : NULL;   /* foreign stream: single-lane decode */
    mi_status st = mi_huffman_decode(ctx, writer->buffer, bits, &tree, (uint32_t)next, toff, (uint8_t *)output, n);
    if (st != MI_OK) { fprintf(stderr, "huffman_decompress: %s\n", mi_status_str(st)); exit(1); }
    *output_size = n;
}

/* huffman.h:108-113 (huffman.c:366-401): the reference's second decoder, a lookup-table walk it left unfinished.  The GPU
 * decoder IS a lookup-table decoder (k_huff_decode: 12-bit LUT + tree walk for longer codes), so both names run it. */
void huffman_decompress_lookup_table(BitWriter *writer, Node *root, char *output, uint64_t *output_size)
{
    huffman_decompress(writer, root, output, output_size);
}

/* frees writer->buffer and the tile table huffman_compress registered for it (the struct itself is the caller's) */
void mi_huffman_release(BitWriter *writer)
{
    if (!writer) return;
    dropin_side_drop(writer->buffer);
    free(writer->buffer);
    writer->buffer = NULL;
}

/* ---- framed files (mi_frame.h): self-describing, decodable without the Node tree ------------------------------------ */
static void write_file(const char *name, const uint8_t *p, uint64_t n)
{
    FILE *f = fopen(name, "wb");
    if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "Error: could not write file %s\n", name); exit(1); }
    fclose(f);
}

int huffman_compress_file(const char *input_filename, const char *output_filename)
{
    uint64_t n; char *in = read_input_buffer(input_filename, &n);
    const uint64_t cap = mi_huffman_bound_words(n), ntiles = (n + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE;
    uint32_t *words = (uint32_t *)calloc(cap + 4, 4);
    uint64_t *toff = (uint64_t *)malloc(8 * (ntiles + 1));
    mi_huffman_info info; mi_huffman_tree tree;
    if (!words || !toff) { fprintf(stderr, "huffman_compress_file: out of memory\n"); exit(1); }
    mi_status st = mi_huffman_encode2(dropin_ctx(), (const uint8_t *)in, n, words, cap, &info, &tree, toff);
    if (st == MI_ERR_EMPTY_INPUT) { printf("ERROR: Queue is empty\n"); exit(1); }
    if (st == MI_ERR_SINGLE_SYMBOL) { printf("ERROR: No code for character %c\n", in[0]); exit(1); }
    if (st != MI_OK) { fprintf(stderr, "huffman_compress_file: %s\n", mi_status_str(st)); exit(1); }
    const uint64_t fcap = mi_frame_bound_huffman(info.total_bits, ntiles);
    uint8_t *frame = (uint8_t *)malloc(fcap); uint64_t fn = 0;
    if (!frame) { fprintf(stderr, "huffman_compress_file: out of memory\n"); exit(1); }
    st = mi_frame_pack_huffman(n, &tree, info.n_nodes, words, info.total_bits, toff, ntiles, frame, fcap, &fn);
    if (st != MI_OK) { fprintf(stderr, "huffman_compress_file: %s\n", mi_status_str(st)); exit(1); }
    write_file(output_filename, frame, fn);
    free(frame); free(toff); free(words); free(in);
    return 0;
}

int huffman_decompress_file(const char *input_filename, const char *output_filename)
{
    uint64_t fn; uint8_t *frame = (uint8_t *)read_input_buffer(input_filename, &fn);
    mi_frame_info fi;
    mi_status st = mi_frame_parse(frame, fn, &fi);
    if (st != MI_OK || fi.codec != MI_FRAME_HUFFMAN) { fprintf(stderr, "huffman_decompress_file: %s is not a Huffman frame\n", input_filename); exit(1); }
    const uint64_t n = fi.original_size, ntiles = (n + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE;
    mi_huffman_tree tree; uint32_t n_nodes = 0; uint64_t bits = 0;
    uint32_t *words = (uint32_t *)malloc(fi.stream_bytes + 8);
    uint64_t *toff = (uint64_t *)malloc(8 * (fi.nblocks + 1));
    uint8_t *out = (uint8_t *)malloc(n ? n : 1);
    if (!words || !toff || !out) { fprintf(stderr, "huffman_decompress_file: out of memory\n"); exit(1); }
    st = mi_frame_unpack_huffman(frame, fn, &tree, &n_nodes, words, fi.stream_bytes / 4 + 2, &bits, toff, fi.nblocks + 1);
    if (st == MI_OK) st = mi_huffman_decode(dropin_ctx(), words, bits, &tree, n_nodes, fi.nblocks == ntiles && ntiles ? toff : NULL, out, n);
    if (st != MI_OK) { fprintf(stderr, "huffman_decompress_file: %s\n", mi_status_str(st)); exit(1); }
    write_file(output_filename, out, n);
    free(out); free(toff); free(words); free(frame);
    return 0;
}
