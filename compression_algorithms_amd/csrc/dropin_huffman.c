/* dropin_huffman.c — algorithms/huffman entry points over the HIP path (see include/mi_huffman.h). */
#include <string.h>
#include "../../include/mi_huffman.h"
#include "dropin_common.h"

#define TRAILER_MAGIC 0x4846464D4954494Cull

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

/* buffer layout: [words: 4*(word_idx + (bit_idx>0))][one zero word][pad to 8][magic][n][ntiles][tile_off[ntiles+1]] */
Node huffman_compress(char *buffer, uint64_t size, BitWriter *writer)
{
    mi_ctx *ctx = dropin_ctx();
    const uint64_t cap = mi_huffman_bound_words(size);
    const uint64_t ntiles = (size + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE;
    uint32_t *words = (uint32_t *)calloc(cap + 4 + 2 * (4 + ntiles + 1), 4);
    uint64_t *toff = (uint64_t *)malloc(8 * (ntiles + 1));
    mi_huffman_info info; mi_huffman_tree tree;
    if (!words || !toff) { fprintf(stderr, "huffman_compress: out of memory\n"); exit(1); }
    mi_status st = mi_huffman_encode2(ctx, (const uint8_t *)buffer, size, words, cap, &info, &tree, toff);
    if (st == MI_ERR_EMPTY_INPUT) { printf("ERROR: Queue is empty\n"); exit(1); }                      /* huffman.c:149-152 */
    if (st == MI_ERR_SINGLE_SYMBOL) { printf("ERROR: No code for character %c\n", buffer[0]); exit(1); }   /* huffman.c:278-281 */
    if (st != MI_OK) { fprintf(stderr, "huffman_compress: %s\n", mi_status_str(st)); exit(1); }
    writer->word_idx = info.word_idx; writer->bit_idx = info.bit_idx; writer->buffer_size = info.buffer_size;
    const uint64_t nw = info.word_idx + (info.bit_idx > 0);
    uint64_t at = ((nw + 1) * 4 + 7) & ~7ull;
    uint64_t *t = (uint64_t *)((uint8_t *)words + at);
    t[0] = TRAILER_MAGIC; t[1] = size; t[2] = ntiles;
    memcpy(t + 3, toff, 8 * (ntiles + 1));
    writer->buffer = (uint32_t *)realloc(words, at + 8 * (3 + ntiles + 1));
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
    const uint64_t nw = writer->word_idx + (writer->bit_idx > 0);
    const uint64_t at = ((nw + 1) * 4 + 7) & ~7ull;
    const uint64_t *t = (const uint64_t *)((const uint8_t *)writer->buffer + at);
    const uint64_t *toff = (t[0] == TRAILER_MAGIC && t[1] == n) ? t + 3 : NULL;   /* foreign stream: single-lane decode */
    mi_status st = mi_huffman_decode(ctx, writer->buffer, bits, &tree, (uint32_t)next, toff, (uint8_t *)output, n);
    if (st != MI_OK) { fprintf(stderr, "huffman_decompress: %s\n", mi_status_str(st)); exit(1); }
    *output_size = n;
}
