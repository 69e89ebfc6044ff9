/* dropin_huffman.c — algorithms/huffman entry points over the HIP path (see include/mi_huffman.h). */
#include <string.h>
#include "../../include/mi_huffman.h"
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
    const dropin_side *e = dropin_side_get(writer->buffer, bits, n);
    const uint64_t *toff = (e && e->count == (n + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE + 1) ? e->table : NULL;   /* foreign stream: single-lane decode */
    mi_status st = mi_huffman_decode(ctx, writer->buffer, bits, &tree, (uint32_t)next, toff, (uint8_t *)output, n);
    if (st != MI_OK) { fprintf(stderr, "huffman_decompress: %s\n", mi_status_str(st)); exit(1); }
    *output_size = n;
}
