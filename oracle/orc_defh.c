/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_table.h).
 *
 * orc_defh.c — "mode H": the entropy stage the reference leaves as a TODO
 * (algorithms/deflate/lz77.c:279 "// TODO: Build huffman tree and encode compressed buffer").
 *
 * PARITY UNPINNED for the bit stream: the reference has no such encoder.  What is taken
 * from the reference:
 *   the token sequence          algorithms/deflate/lz77.c:199-280 (byte-exact, see orc_lz.c)
 *   the 286-symbol alphabet     algorithms/deflate/huffman.h:6 (NUM_CODES) and
 *                               huffman.c:49-62: literal byte b -> symbol b; a match with
 *                               offset d -> symbol 256 + (clz32(d) - 16), i.e. 257..271
 *   the tally                   lz77.c:206,231,273 (the `frequencies` array it throws away)
 *   MSB-first u32 bit packing   algorithms/deflate/huffman.c:16-46 (write_bits)
 *   the heap / merge procedure  algorithms/huffman/huffman.c:100-163,189-211 (deflate's own
 *                               push_heap/pop_heap are declared, huffman.h:16-32, never defined)
 *
 * Defined here (and in DESIGN.md): code LENGTHS come from that heap procedure over the
 * block's tally; codes are then assigned canonically (by length, then symbol) so that 286
 * length bytes describe the tree; a match symbol is followed by the offset's bits below its
 * leading one (15 - clz16(d) of them) and the 5-bit length.
 *
 * Block record, 4-byte aligned:
 *   u32 n_tokens | u8 len[286] + 2 pad | u32 words[] (MSB-first)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DEFH_NSYM 286

typedef struct { uint32_t freq[2 * DEFH_NSYM]; int16_t left[2 * DEFH_NSYM], right[2 * DEFH_NSYM]; int n; } defh_tree;
typedef struct { int16_t a[DEFH_NSYM]; int n; } defh_heap;

static void dh_up(defh_heap *h, const defh_tree *t, int i)
{
    while (i > 0) {
        int par = (i - 1) / 2;
        if (!(t->freq[h->a[i]] < t->freq[h->a[par]])) break;
        int16_t s = h->a[i]; h->a[i] = h->a[par]; h->a[par] = s;
        i = par;
    }
}
static void dh_down(defh_heap *h, const defh_tree *t, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, best = i;
        if (l < h->n && t->freq[h->a[l]] < t->freq[h->a[best]]) best = l;
        if (r < h->n && t->freq[h->a[r]] < t->freq[h->a[best]]) best = r;
        if (best == i) return;
        int16_t s = h->a[i]; h->a[i] = h->a[best]; h->a[best] = s;
        i = best;
    }
}
static int dh_pop(defh_heap *h, const defh_tree *t)
{
    int id = h->a[0];
    h->a[0] = h->a[--h->n];
    dh_down(h, t, 0);
    return id;
}

/* code lengths from the reference heap procedure (leaves enqueued in symbol order) */
void orc_defh_lengths(const uint32_t freq[DEFH_NSYM], uint8_t len[DEFH_NSYM])
{
    defh_tree t; defh_heap h; int leaf_of[DEFH_NSYM];
    t.n = 0; h.n = 0;
    memset(len, 0, DEFH_NSYM);
    for (int s = 0; s < DEFH_NSYM; ++s) {
        leaf_of[s] = -1;
        if (!freq[s]) continue;
        int id = t.n++;
        t.freq[id] = freq[s]; t.left[id] = t.right[id] = -1; leaf_of[s] = id;
        h.a[h.n++] = (int16_t)id; dh_up(&h, &t, h.n - 1);
    }
    if (t.n == 0) return;
    if (t.n == 1) { for (int s = 0; s < DEFH_NSYM; ++s) if (freq[s]) len[s] = 1; return; }
    int parent[2 * DEFH_NSYM];
    while (h.n > 1) {
        int l = dh_pop(&h, &t), r = dh_pop(&h, &t), id = t.n++;
        t.freq[id] = t.freq[l] + t.freq[r]; t.left[id] = (int16_t)l; t.right[id] = (int16_t)r;
        parent[l] = id; parent[r] = id;
        h.a[h.n++] = (int16_t)id; dh_up(&h, &t, h.n - 1);
    }
    int root = dh_pop(&h, &t);
    for (int s = 0; s < DEFH_NSYM; ++s) {
        if (leaf_of[s] < 0) continue;
        int d = 0, node = leaf_of[s];
        while (node != root) { node = parent[node]; ++d; }
        len[s] = (uint8_t)d;
    }
}

/* canonical codes: by (length, symbol) */
void orc_defh_codes(const uint8_t len[DEFH_NSYM], uint32_t code[DEFH_NSYM])
{
    uint32_t count[34] = {0}, next[34] = {0};
    for (int s = 0; s < DEFH_NSYM; ++s) if (len[s]) ++count[len[s]];
    uint32_t c = 0;
    for (int l = 1; l <= 32; ++l) { c = (c + count[l - 1]) << 1; next[l] = c; }
    for (int s = 0; s < DEFH_NSYM; ++s) code[s] = len[s] ? next[len[s]]++ : 0;
}

static inline uint32_t clz16(uint32_t d) { return (uint32_t)__builtin_clz(d & 0xFFFFu) - 16u; }

typedef struct { uint32_t *w; uint64_t bit; } dh_bits;
static inline void dh_put(dh_bits *b, uint32_t v, uint32_t k)       /* MSB-first, deflate/huffman.c:16-46 */
{
    for (uint32_t i = 0; i < k; ++i, ++b->bit)
        if ((v >> (k - 1 - i)) & 1u) b->w[b->bit >> 5] |= 1u << (31 - (b->bit & 31));
}
static inline uint32_t dh_get(const uint32_t *w, uint64_t *bit, uint32_t k)
{
    uint32_t v = 0;
    for (uint32_t i = 0; i < k; ++i, ++*bit) v = (v << 1) | ((w[*bit >> 5] >> (31 - (*bit & 31))) & 1u);
    return v;
}

/* tok: the reference's byte tokens of ONE block ({0,c} / {1,dlo,dhi,len}); out must hold
 * 4 + 288 + ntok_bytes*4 + 8 bytes.  Returns the record size in bytes (multiple of 4). */
uint64_t orc_defh_encode_block(const uint8_t *tok, uint64_t ntok_bytes, uint8_t *out)
{
    uint32_t freq[DEFH_NSYM] = {0}, code[DEFH_NSYM], ntok = 0;
    uint8_t len[DEFH_NSYM];
    for (uint64_t i = 0; i < ntok_bytes;) {
        if (tok[i] == 0) { ++freq[tok[i + 1]]; i += 2; }
        else { uint32_t d = tok[i + 1] | ((uint32_t)tok[i + 2] << 8); ++freq[256 + clz16(d)]; i += 4; }
        ++ntok;
    }
    orc_defh_lengths(freq, len);
    orc_defh_codes(len, code);
    memcpy(out, &ntok, 4);
    memcpy(out + 4, len, DEFH_NSYM); out[4 + 286] = out[4 + 287] = 0;
    dh_bits b = { (uint32_t *)(out + 292), 0 };
    memset(out + 292, 0, ntok_bytes * 4 + 8);
    for (uint64_t i = 0; i < ntok_bytes;) {
        if (tok[i] == 0) { dh_put(&b, code[tok[i + 1]], len[tok[i + 1]]); i += 2; }
        else {
            uint32_t d = tok[i + 1] | ((uint32_t)tok[i + 2] << 8), l = tok[i + 3], cz = clz16(d), nx = 15u - cz;
            dh_put(&b, code[256 + cz], len[256 + cz]);
            dh_put(&b, d - (1u << nx), nx);
            dh_put(&b, l, 5);
            i += 4;
        }
    }
    return 292 + ((b.bit + 31) >> 5) * 4;
}

/* record -> byte tokens; returns token bytes written, UINT64_MAX on a malformed record */
uint64_t orc_defh_decode_block(const uint8_t *rec, uint64_t rec_bytes, uint8_t *tok_out, uint64_t cap)
{
    if (rec_bytes < 292) return UINT64_MAX;
    uint32_t ntok; memcpy(&ntok, rec, 4);
    const uint8_t *len = rec + 4;
    uint32_t code[DEFH_NSYM];
    orc_defh_codes(len, code);
    const uint32_t *w = (const uint32_t *)(rec + 292);
    const uint64_t nbits = (rec_bytes - 292) * 8;
    uint64_t bit = 0, o = 0;
    for (uint32_t t = 0; t < ntok; ++t) {
        uint32_t acc = 0, l = 0; int sym = -1;
        while (l < 32 && sym < 0) {
            if (bit >= nbits) return UINT64_MAX;
            acc = (acc << 1) | dh_get(w, &bit, 1); ++l;
            for (int s = 0; s < DEFH_NSYM; ++s) if (len[s] == l && code[s] == acc) { sym = s; break; }
        }
        if (sym < 0) return UINT64_MAX;
        if (sym < 256) { if (o + 2 > cap) return UINT64_MAX; tok_out[o++] = 0; tok_out[o++] = (uint8_t)sym; }
        else {
            uint32_t cz = (uint32_t)sym - 256u; if (cz < 1 || cz > 15) return UINT64_MAX;
            uint32_t nx = 15u - cz, d = (1u << nx) + dh_get(w, &bit, nx), ln = dh_get(w, &bit, 5);
            if (o + 4 > cap) return UINT64_MAX;
            tok_out[o++] = 1; tok_out[o++] = d & 0xFF; tok_out[o++] = (uint8_t)(d >> 8); tok_out[o++] = (uint8_t)ln;
        }
    }
    return o;
}
