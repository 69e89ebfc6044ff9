/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_table.h).
 *
 * orc_lz.c — CPU restatement of the reference's two greedy LZ77 tokenisers.
 *
 *   orc_lz77_*     algorithms/lz77/lz77.c:264-345 (compress), :347-377 (decompress),
 *                  :139-184 (LSB-first bit I/O)
 *   orc_deflate_*  algorithms/deflate/lz77.c:199-280 (per-block tokeniser),
 *                  :176-197 (byte tokens), algorithms/deflate/deflate.c:47-63 (block loop),
 *                  algorithms/deflate/huffman.c:49-62 (the 286-bin tally that the
 *                  reference computes and throws away)
 *
 * Tail rule: the reference reads a 4-byte word at every
 * position and extends matches without a `p < size` bound, i.e. it reads past
 * the end of its input.  The restatement DEFINES those bytes: callers hand in
 * `in` with at least ORC_TAIL readable bytes after `n`; the *_padded helpers
 * copy into such a buffer with a zero tail.
 */
#include <stdio.h>
#include "orc_table.h"

#define ORC_TAIL 64

typedef struct {
    uint32_t wbits;     /* window bits            lz77: 14 (or 16), deflate: 15 */
    uint32_t lbits;     /* length bits            lz77: 4,          deflate: 5  */
    uint32_t tbits;     /* table bits             lz77: wbits+6,    deflate: 20 */
    int      deflate;   /* 0: lz77 rules, 1: deflate rules (wrap insert, literal test >= W-1) */
} orc_lz_params;

/* one step of the shared parse.  Returns match length (0 = literal) and the
 * candidate position through *cand.  Inserts every covered position.
 * lz77.c:285-337 / deflate lz77.c:219-274 */
static inline uint32_t orc_lz_step(orc_table *t, const orc_lz_params *P,
                                   const uint8_t *in, uint64_t p, uint64_t *cand)
{
    const uint64_t W = 1ull << P->wbits;
    const uint32_t max_len = (1u << P->lbits) - 1u;
    uint32_t w = orc_word_at(in + p);
    uint64_t m = orc_table_find(t, w);
    *cand = m;
    int literal;
    if (m == ORC_NONE) literal = 1;
    else if (P->deflate) literal = (uint64_t)(p - m) >= W - 1;     /* deflate lz77.c:223 */
    else literal = (p - m) == W;                                    /* lz77.c:290        */
    if (literal) {
        orc_table_insert(t, w, p);
        return 0;
    }
    uint64_t a = m + 4, b = p + 4;
    while (in[a] == in[b] && a - m < max_len) { ++a; ++b; }        /* no `b < n` bound  */
    uint32_t len = (uint32_t)(a - m);
    for (uint32_t i = 0; i < len; ++i)
        orc_table_insert(t, orc_word_at(in + p + i), p + i);
    return len;
}

/* ---------- LSB-first bit writer (lz77.c:144-174) ---------- */
typedef struct { uint8_t *data; uint64_t nbits; } orc_bits;

static inline void orc_put(orc_bits *s, uint64_t v, unsigned k)
{
    for (unsigned i = 0; i < k; ++i) {
        uint64_t byte = s->nbits >> 3; unsigned off = s->nbits & 7;
        if ((v >> i) & 1) s->data[byte] |= (uint8_t)(1u << off);
        else              s->data[byte] &= (uint8_t)~(1u << off);
        s->nbits++;
    }
}
static inline uint64_t orc_get(const uint8_t *d, uint64_t *pos, unsigned k)
{
    uint64_t v = 0;
    for (unsigned i = 0; i < k; ++i, ++*pos)
        if ((d[*pos >> 3] >> (*pos & 7)) & 1) v |= 1ull << i;
    return v;
}

/*
 * Whole-buffer lz77 encode.  `in` must have ORC_TAIL defined bytes after n.
 * `out` must hold 2*n+8 bytes and is fully zeroed first, so pad bits are 0
 * (the reference leaves them as uninitialised heap: SURVEY.md A.1.6 ii).
 * Returns the bit count (= BitStream.bit_index).  cand_trace (optional, n
 * entries) receives, for token-start positions, the find() result
 * (0xFFFFFFFF = none) and 0xFFFFFFFE for positions covered by a match.
 */
uint64_t orc_lz77_encode(const uint8_t *in, uint64_t n, uint32_t wbits, uint32_t lbits,
                         uint32_t tbits, uint8_t *out, uint32_t *cand_trace)
{
    orc_lz_params P = { wbits, lbits, tbits, 0 };
    orc_table t;
    if (orc_table_init(&t, tbits, wbits, 0)) return 0;
    memset(out, 0, 2 * n + 8);
    orc_bits s = { out, 0 };
    uint64_t p = 0;
    while (p < n) {
        uint64_t cand;
        uint32_t len = orc_lz_step(&t, &P, in, p, &cand);
        if (cand_trace) cand_trace[p] = cand == ORC_NONE ? 0xFFFFFFFFu : (uint32_t)cand;
        if (!len) {
            orc_put(&s, 0, 1); orc_put(&s, in[p], 8);
            p += 1;
        } else {
            if (cand_trace) for (uint32_t i = 1; i < len && p + i < n; ++i) cand_trace[p + i] = 0xFFFFFFFEu;
            uint64_t off = p - cand;
            orc_put(&s, 1, 1); orc_put(&s, off, wbits); orc_put(&s, len, lbits);
            p += len;
        }
    }
    orc_table_free(&t);
    return s.nbits;
}

/* lz77.c:347-377, with the overshoot of the last match truncated at n */
uint64_t orc_lz77_decode(const uint8_t *stream, uint64_t nbits, uint32_t wbits, uint32_t lbits,
                         uint8_t *out, uint64_t n)
{
    uint64_t pos = 0, o = 0;
    while (o < n && pos < nbits) {
        if (orc_get(stream, &pos, 1)) {
            uint64_t off = orc_get(stream, &pos, wbits);
            uint64_t len = orc_get(stream, &pos, lbits);
            if (off == 0 || off > o) return UINT64_MAX;
            for (uint64_t i = 0; i < len && o < n; ++i, ++o) out[o] = out[o - off];
        } else {
            out[o++] = (uint8_t)orc_get(stream, &pos, 8);
        }
    }
    return o;
}

/* The reference's first, brute-force parser: lz77_compress_old, algorithms/lz77/lz77.c:185-262.  At every token start the
 * window of 2^wbits - 1 bytes is searched from its far end; a candidate counts when its 4-byte word equals the word at the
 * position (lz77.c:215-221: both words may reach up to 3 bytes past `size`; the zero tail behind `in` defines them), its
 * length is the common prefix, stopped at the maximum and at `size` (lz77.c:223-233), and it replaces the best so far only when
 * strictly longer (lz77.c:236-239).  For the first 2^wbits - 1 positions `buffer_index - window_size` wraps around (uint64_t;
 * max() on unsigned operands, lz77.c:208): the loop over the window never runs and the token is a literal.  Returns the
 * stream's length in bits; `out` (2 n + 8 bytes) is zeroed first.  `in` needs >= 34 readable zero bytes after n. */
uint64_t orc_lz77_old_encode(const uint8_t *in, uint64_t n, uint32_t wbits, uint32_t lbits, uint8_t *out)
{
    const uint64_t ws = (1ull << wbits) - 1, max_len = (1ull << lbits) - 1;
    memset(out, 0, 2 * n + 8);
    orc_bits s = { out, 0 };
    uint64_t p = 0;
    while (p < n) {
        uint64_t best = 0, off = 0;
        if (p >= ws) {
            for (uint64_t q = p - ws; q < p; ++q) {
                if (memcmp(in + q, in + p, 4) != 0) continue;
                uint64_t len = 0;
                while (p + len < n && in[q + len] == in[p + len]) { ++len; if (len >= max_len) break; }
                if (len > best) { best = len; off = p - q; }
            }
        }
        if (best) {
            orc_put(&s, 1, 1); orc_put(&s, off, wbits); orc_put(&s, best, lbits);
            p += best;
        } else {
            orc_put(&s, 0, 1); orc_put(&s, in[p], 8);
            p += 1;
        }
    }
    return s.nbits;
}

/* ---------- deflate per-block tokeniser ---------- */

typedef struct {
    orc_table table;
    orc_lz_params P;
    uint8_t *buf;          /* the reference's reused 65536-byte fread buffer (+ tail) */
    uint32_t block;
} orc_deflate;

orc_deflate *orc_deflate_new(uint32_t block)
{
    orc_deflate *d = (orc_deflate *)calloc(1, sizeof *d);
    if (!d) return NULL;
    d->P.wbits = 15; d->P.lbits = 5; d->P.tbits = 20; d->P.deflate = 1;
    d->block = block;
    d->buf = (uint8_t *)calloc(block + ORC_TAIL, 1);
    if (!d->buf || orc_table_init(&d->table, 20, 15, 1)) { free(d->buf); free(d); return NULL; }
    return d;
}
void orc_deflate_free(orc_deflate *d) { if (d) { orc_table_free(&d->table); free(d->buf); free(d); } }
void orc_deflate_reset_table(orc_deflate *d) { orc_table_reset(&d->table); }
void orc_deflate_zero_buffer(orc_deflate *d) { memset(d->buf, 0, d->block + ORC_TAIL); }

/*
 * One call of the reference's per-block lz77_compress (deflate/lz77.c:199-280).
 * `in` is used in place (must have ORC_TAIL defined bytes after n).
 * out needs 2*n+4 bytes.  freq286 (optional) gets the tally of
 * deflate/huffman.c:49-62.  Returns bytes written.
 */
uint64_t orc_deflate_block_raw(orc_deflate *d, const uint8_t *in, uint64_t n, uint8_t *out,
                               uint32_t *freq286, uint32_t *cand_trace)
{
    uint64_t p = 0, o = 0;
    if (freq286) memset(freq286, 0, 286 * sizeof(uint32_t));
    while (p < n) {
        uint64_t cand;
        uint32_t len = orc_lz_step(&d->table, &d->P, in, p, &cand);
        if (cand_trace) cand_trace[p] = cand == ORC_NONE ? 0xFFFFFFFFu : (uint32_t)cand;
        if (!len) {
            out[o++] = 0; out[o++] = in[p];                         /* lz77.c:176-184 */
            if (freq286) ++freq286[in[p]];
            p += 1;
        } else {
            if (cand_trace) for (uint32_t i = 1; i < len && p + i < n; ++i) cand_trace[p + i] = 0xFFFFFFFEu;
            uint32_t off = (uint32_t)(p - cand);
            out[o++] = 1; out[o++] = off & 0xFF; out[o++] = (off >> 8) & 0xFF; out[o++] = (uint8_t)len;
            if (freq286) ++freq286[256 + (__builtin_clz(off & 0xFFFFu) - 16)];
            p += len;
        }
    }
    return o;
}

/* Block copied into the persistent buffer first (what fread does in deflate.c:47). */
uint64_t orc_deflate_block(orc_deflate *d, const uint8_t *in, uint64_t n, uint8_t *out,
                           uint32_t *freq286, uint32_t *cand_trace)
{
    if (n > d->block) return UINT64_MAX;
    memcpy(d->buf, in, n);
    return orc_deflate_block_raw(d, d->buf, n, out, freq286, cand_trace);
}

/*
 * Whole-input drivers.
 *  mode 0 "shipped":     deflate.c:10-76 — one table for all blocks, one reused buffer
 *                        (a short last block is followed by stale bytes of the previous one).
 *  mode 1 "independent": fresh table and zero tail for every block (the sharded /
 *                        block-parallel parity definition).
 * out needs 2*n + 4*nblocks bytes.  sizes (optional) gets per-block byte counts.
 */
uint64_t orc_deflate_stream(const uint8_t *in, uint64_t n, uint32_t block, int mode,
                            uint8_t *out, uint64_t *sizes, uint64_t *max_bucket)
{
    orc_deflate *d = orc_deflate_new(block);
    if (!d) return UINT64_MAX;
    uint64_t o = 0, b = 0;
    for (uint64_t at = 0; at < n; at += block, ++b) {
        uint64_t len = n - at < block ? n - at : block;
        if (mode == 1) { orc_deflate_reset_table(d); orc_deflate_zero_buffer(d); }
        uint64_t w = orc_deflate_block(d, in + at, len, out + o, NULL, NULL);
        if (sizes) sizes[b] = w;
        o += w;
    }
    if (max_bucket) *max_bucket = d->table.max_bucket;   /* >= 2^20: reference UB on this input */
    orc_deflate_free(d);
    return o;
}

/* Decoder for the byte-token stream of ONE block; truncates an overshooting last match. */
uint64_t orc_deflate_block_decode(const uint8_t *tok, uint64_t ntok, uint8_t *out, uint64_t n)
{
    uint64_t i = 0, o = 0;
    while (i < ntok && o < n) {
        if (tok[i] == 0) { if (i + 2 > ntok) return UINT64_MAX; out[o++] = tok[i + 1]; i += 2; }
        else if (tok[i] == 1) {
            if (i + 4 > ntok) return UINT64_MAX;
            uint32_t off = tok[i + 1] | ((uint32_t)tok[i + 2] << 8), len = tok[i + 3];
            if (off == 0 || off > o) return UINT64_MAX;
            for (uint32_t k = 0; k < len && o < n; ++k, ++o) out[o] = out[o - off];
            i += 4;
        } else return UINT64_MAX;
    }
    return (i == ntok) ? o : UINT64_MAX;
}

/*
 * find() at EVERY position: cand[p] = result of find(word_p) in the table state
 * after positions 0..p-1 were inserted — well defined because every position is
 * inserted exactly once, in order, whatever the parse does.
 * Used to check the GPU match finder independently of the parse.
 */
void orc_find_all(const uint8_t *in, uint64_t n, uint32_t wbits, uint32_t tbits, int deflate,
                  uint32_t *cand)
{
    orc_table t;
    if (orc_table_init(&t, tbits, wbits, deflate)) return;
    for (uint64_t p = 0; p < n; ++p) {
        uint32_t w = orc_word_at(in + p);
        uint64_t m = orc_table_find(&t, w);
        cand[p] = m == ORC_NONE ? 0xFFFFFFFFu : (uint32_t)m;
        orc_table_insert(&t, w, p);
    }
    orc_table_free(&t);
}

/* the reference's hash(): home bucket of a 4-byte word */
uint32_t orc_home_of(uint32_t w, uint32_t tbits) { return orc_home(w, tbits); }

/*
 * Highest bucket number any probe of a find-at-every-position run looks at.  A value
 * >= 2^tbits means the REFERENCE would read (deflate find) or write (lz77 insert) past the
 * end of its bucket array on this input — undefined behaviour there (it corrupts the heap),
 * defined here as "bucket ids are unbounded, buckets past T start empty".  The golden
 * generator never runs the reference on such inputs.
 */
uint64_t orc_max_bucket(const uint8_t *in, uint64_t n, uint32_t wbits, uint32_t tbits, int deflate)
{
    orc_table t;
    if (orc_table_init(&t, tbits, wbits, deflate)) return UINT64_MAX;
    for (uint64_t p = 0; p < n; ++p) {
        uint32_t w = orc_word_at(in + p);
        (void)orc_table_find(&t, w);
        orc_table_insert(&t, w, p);
    }
    uint64_t m = t.max_bucket;
    orc_table_free(&t);
    return m;
}

/* probe statistics of the last full run, for DESIGN.md */
void orc_probe_stats(const uint8_t *in, uint64_t n, uint32_t wbits, uint32_t tbits, int deflate,
                     uint64_t stats[4])
{
    orc_table t;
    if (orc_table_init(&t, tbits, wbits, deflate)) return;
    for (uint64_t p = 0; p < n; ++p) {
        uint32_t w = orc_word_at(in + p);
        (void)orc_table_find(&t, w);
        orc_table_insert(&t, w, p);
    }
    stats[0] = t.n_insert; stats[1] = t.n_probe_insert; stats[2] = t.n_find; stats[3] = t.n_probe_find;
    orc_table_free(&t);
}
