/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_table.h).
 *
 * orc_huff.c — CPU restatement of the reference's whole-buffer order-0 Huffman
 * coder (algorithms/huffman/huffman.c).
 *
 *   histogram            huffman.c:184-187   u32 counters, wrapping
 *   min-heap + merge     huffman.c:100-163, 189-211  (tie-breaking defines the codes)
 *   tree-path codes      huffman.c:217-250   left = 0, right = 1, not canonical
 *   MSB-first u32 pack   huffman.c:18-48
 *   sizes                huffman.c:318-320
 *   decoder              huffman.c:330-364   (here: stops at the true length)
 *
 * The heap is restated over integer node ids instead of malloc'd nodes; the
 * comparison sequence is the reference's: sift-up while child < parent
 * (strict), sift-down picks left when left < cur, then right when
 * right < best (strict), recursing only on a swap.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_HUFF_MAXNODES 511

typedef struct {
    uint32_t freq[ORC_HUFF_MAXNODES];
    int16_t  left[ORC_HUFF_MAXNODES], right[ORC_HUFF_MAXNODES];
    uint8_t  value[ORC_HUFF_MAXNODES];
    int      nnodes, root;
} orc_huff_tree;

typedef struct { int16_t a[256]; int n; } orc_heap;

static void heap_up(orc_heap *h, const orc_huff_tree *t, int i)
{
    while (i > 0) {
        int par = (i - 1) / 2;
        if (!(t->freq[h->a[i]] < t->freq[h->a[par]])) break;
        int16_t s = h->a[i]; h->a[i] = h->a[par]; h->a[par] = s;
        i = par;
    }
}
static void heap_down(orc_heap *h, const orc_huff_tree *t, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = 2 * i + 2, best = i;
        if (l < h->n && t->freq[h->a[l]] < t->freq[h->a[best]]) best = l;
        if (r < h->n && t->freq[h->a[r]] < t->freq[h->a[best]]) best = r;
        if (best == i) return;
        int16_t s = h->a[i]; h->a[i] = h->a[best]; h->a[best] = s;
        i = best;
    }
}
static void heap_push(orc_heap *h, const orc_huff_tree *t, int id) { h->a[h->n++] = (int16_t)id; heap_up(h, t, h->n - 1); }
static int  heap_pop(orc_heap *h, const orc_huff_tree *t)
{
    int id = h->a[0];
    h->a[0] = h->a[--h->n];
    heap_down(h, t, 0);
    return id;
}

void orc_huff_histogram(const uint8_t *in, uint64_t n, uint32_t freq[256])
{
    memset(freq, 0, 256 * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) ++freq[in[i]];
}

/* returns number of distinct symbols; 0 => reference would exit(1) "Queue is empty" */
int orc_huff_build(const uint32_t freq[256], orc_huff_tree *t)
{
    orc_heap h; h.n = 0;
    t->nnodes = 0; t->root = -1;
    for (int s = 0; s < 256; ++s) {             /* 0..254 then 255: plain symbol order */
        if (!freq[s]) continue;
        int id = t->nnodes++;
        t->freq[id] = freq[s]; t->left[id] = t->right[id] = -1; t->value[id] = (uint8_t)s;
        heap_push(&h, t, id);
    }
    int nsym = t->nnodes;
    if (!nsym) return 0;
    while (h.n > 1) {
        int l = heap_pop(&h, t), r = heap_pop(&h, t);
        int id = t->nnodes++;
        t->freq[id] = t->freq[l] + t->freq[r];   /* u32, wraps like the reference */
        t->left[id] = (int16_t)l; t->right[id] = (int16_t)r; t->value[id] = 0;
        heap_push(&h, t, id);
    }
    t->root = heap_pop(&h, t);
    return nsym;
}

/* codes as 64-bit so that depths > 32 are representable and can be reported; returns max length */
int orc_huff_codes(const orc_huff_tree *t, uint64_t code[256], uint8_t len[256])
{
    memset(code, 0, 256 * sizeof(uint64_t)); memset(len, 0, 256);
    if (t->root < 0) return 0;
    int stack[ORC_HUFF_MAXNODES]; uint64_t cstack[ORC_HUFF_MAXNODES]; int dstack[ORC_HUFF_MAXNODES];
    int sp = 0, maxlen = 0;
    stack[0] = t->root; cstack[0] = 0; dstack[0] = 0; sp = 1;
    while (sp) {
        --sp;
        int id = stack[sp]; uint64_t c = cstack[sp]; int d = dstack[sp];
        if (t->left[id] < 0 && t->right[id] < 0) {
            code[t->value[id]] = c; len[t->value[id]] = (uint8_t)d;
            if (d > maxlen) maxlen = d;
            continue;
        }
        stack[sp] = t->right[id]; cstack[sp] = (c << 1) | 1; dstack[sp] = d + 1; ++sp;
        stack[sp] = t->left[id];  cstack[sp] = (c << 1);     dstack[sp] = d + 1; ++sp;
    }
    return maxlen;
}

/*
 * Encode.  words must hold ceil(bits/32)+1 zeroed u32.  Returns total bits,
 * or UINT64_MAX where the reference would have exit(1)'d (empty input, a
 * single distinct symbol => code length 0) or silently produced garbage
 * (a code longer than 32 bits).
 * Stream bit j is bit (31 - j%32) of word j/32 (huffman.c:18-48).
 */
uint64_t orc_huff_encode(const uint8_t *in, uint64_t n, uint32_t *words, uint64_t nwords,
                         uint32_t codes_out[256], uint8_t lens_out[256])
{
    uint32_t freq[256];
    orc_huff_tree t;
    uint64_t code[256]; uint8_t len[256];
    orc_huff_histogram(in, n, freq);
    int nsym = orc_huff_build(freq, &t);
    int maxlen = orc_huff_codes(&t, code, len);
    if (codes_out) for (int s = 0; s < 256; ++s) codes_out[s] = (uint32_t)code[s];
    if (lens_out) memcpy(lens_out, len, 256);
    if (nsym < 2 || maxlen > 32) return UINT64_MAX;
    memset(words, 0, nwords * sizeof(uint32_t));
    uint64_t bit = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t c = code[in[i]]; unsigned l = len[in[i]];
        for (unsigned k = 0; k < l; ++k, ++bit)
            if ((c >> (l - 1 - k)) & 1) words[bit >> 5] |= 1u << (31 - (bit & 31));
    }
    return bit;
}

/* huffman.c:318-320 */
uint64_t orc_huff_buffer_size(uint64_t bits)
{
    uint64_t w = bits >> 5, b = bits & 31;
    return w * 4 + b / 8 + ((b % 8) > 0);
}

/* tree-walk decode of exactly n symbols (huffman.c:330-364 minus its pad-bit overrun) */
uint64_t orc_huff_decode(const uint32_t *words, uint64_t bits, const uint32_t freq[256],
                         uint8_t *out, uint64_t n)
{
    orc_huff_tree t;
    if (orc_huff_build(freq, &t) < 2) return UINT64_MAX;
    uint64_t bit = 0, o = 0;
    while (o < n) {
        int id = t.root;
        while (t.left[id] >= 0) {
            if (bit >= bits) return UINT64_MAX;
            int b = (words[bit >> 5] >> (31 - (bit & 31))) & 1; ++bit;
            id = b ? t.right[id] : t.left[id];
        }
        out[o++] = t.value[id];
    }
    return o;
}

/* serialise the tree shape for comparison with the reference's Node tree:
 * pre-order, leaf = {1,value}, inner = {0,0}. Returns count of records. */
int orc_huff_preorder(const uint32_t freq[256], uint8_t *kinds, uint8_t *values, uint32_t *freqs)
{
    orc_huff_tree t;
    if (orc_huff_build(freq, &t) < 1) return 0;
    int stack[ORC_HUFF_MAXNODES], sp = 0, k = 0;
    stack[sp++] = t.root;
    while (sp) {
        int id = stack[--sp];
        int leaf = t.left[id] < 0;
        kinds[k] = (uint8_t)leaf; values[k] = t.value[id]; freqs[k] = t.freq[id]; ++k;
        if (!leaf) { stack[sp++] = t.right[id]; stack[sp++] = t.left[id]; }
    }
    return k;
}

/* ---- pieces of the same coder, for the sharded (multi-GPU) parity tests: codes from a given histogram
 * (huffman.c:189-250 on the SUMMED counts) and packing a shard with given codes from a bit offset
 * (huffman.c:18-48 continued mid-word). */
int orc_huff_codes_from_freq(const uint32_t freq[256], uint32_t codes[256], uint8_t lens[256])
{
    orc_huff_tree t;
    uint64_t code[256];
    int nsym = orc_huff_build(freq, &t);
    int maxlen = orc_huff_codes(&t, code, lens);
    for (int s = 0; s < 256; ++s) codes[s] = (uint32_t)code[s];
    return (nsym < 2 || maxlen > 32) ? -1 : maxlen;
}

/* words: zeroed, >= ceil((bit_offset + bits)/32) + 1 entries; returns bit_offset + bits */
uint64_t orc_huff_pack(const uint8_t *in, uint64_t n, const uint32_t codes[256], const uint8_t lens[256],
                       uint64_t bit_offset, uint32_t *words)
{
    uint64_t bit = bit_offset;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = codes[in[i]]; unsigned l = lens[in[i]];
        for (unsigned k = 0; k < l; ++k, ++bit)
            if ((c >> (l - 1 - k)) & 1) words[bit >> 5] |= 1u << (31 - (bit & 31));
    }
    return bit;
}
