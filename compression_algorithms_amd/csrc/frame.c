/* frame.c — the self-describing container of include/mi_frame.h.  Plain C, host only: it moves the bytes the encoders
 * produced into and out of a framed file image; it never encodes or decodes.  Shape after the reference's Zig Huffman
 * program (algorithms/huffman/zig_huffman/src/main.zig:11-18 CompressedSize {last_block:1, size:31}; :155-176 tree in
 * pre-order with -1 for an absent child; :513-530 chunk = size word + payload). */
#include <string.h>
#include "../../include/mi_frame.h"

#define CHUNK_MAX 0x7FFFFFFFull            /* 31-bit size field */

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t get64(const uint8_t *p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }

static int is_block_codec(uint32_t c) { return c == MI_FRAME_DEFLATE_T || c == MI_FRAME_DEFLATE_H || c == MI_FRAME_LZ77 || c == MI_FRAME_FSE; }
static uint32_t align_of(uint32_t c) { return c == MI_FRAME_LZ77 ? 1u : c == MI_FRAME_DEFLATE_T ? 8u : 32u; }

static void put_header(uint8_t *out, uint32_t codec, uint32_t block, uint32_t p0, uint32_t p1, uint64_t original_size)
{
    memcpy(out, MI_FRAME_MAGIC, 4);
    out[4] = 1; out[5] = (uint8_t)codec; out[6] = 0; out[7] = 0;
    put32(out + 8, block); put32(out + 12, p0); put32(out + 16, p1); put64(out + 20, original_size); put32(out + 28, 0);
}

/* nbits of LSB-first packed bits from src (starting at bit sbit) to dst (starting at bit 0); dst bytes fully written */
static void take_bits(uint8_t *dst, const uint8_t *src, uint64_t sbit, uint64_t nbits)
{
    const uint64_t nbytes = (nbits + 7) / 8, sb = sbit >> 3;
    const uint32_t sh = (uint32_t)(sbit & 7u);
    if (!sh) memcpy(dst, src + sb, nbytes);
    else {
        const uint64_t last_src = (sbit + nbits - 1) >> 3;            /* last source byte that holds a wanted bit */
        for (uint64_t i = 0; i < nbytes; ++i) {
            uint32_t v = src[sb + i] >> sh;
            if (sb + i + 1 <= last_src) v |= (uint32_t)src[sb + i + 1] << (8u - sh);
            dst[i] = (uint8_t)v;
        }
    }
    if (nbits & 7u) dst[nbytes - 1] &= (uint8_t)((1u << (nbits & 7u)) - 1u);
}

/* nbits from src (bit 0 on) OR-ed into dst at bit dbit; dst must be zero where the bits land */
static void place_bits(uint8_t *dst, uint64_t dbit, const uint8_t *src, uint64_t nbits)
{
    const uint64_t nbytes = (nbits + 7) / 8, db = dbit >> 3;
    const uint32_t sh = (uint32_t)(dbit & 7u);
    for (uint64_t i = 0; i < nbytes; ++i) {
        uint32_t v = src[i];
        if (i == nbytes - 1 && (nbits & 7u)) v &= (1u << (nbits & 7u)) - 1u;
        dst[db + i] |= (uint8_t)(v << sh);
        if (sh && (v >> (8u - sh))) dst[db + i + 1] |= (uint8_t)(v >> (8u - sh));
    }
}

uint64_t mi_frame_bound_blocks(uint64_t nblocks, uint64_t stream_bytes)
{
    return MI_FRAME_HEADER + stream_bytes + 9 * (nblocks + 1) + 16;     /* size word (+ bit count) + <= 1 pad byte per block */
}

mi_status mi_frame_pack_blocks(uint32_t codec, uint32_t block, uint32_t p0, uint32_t p1, uint64_t original_size,
                               const uint8_t *h_stream, const uint64_t *t, uint64_t nblocks,
                               uint8_t *out, uint64_t cap, uint64_t *out_bytes)
{
    if (!is_block_codec(codec) || !t || !out || !out_bytes || (nblocks && !h_stream) || !block) return MI_ERR_ARG;
    if (nblocks != (original_size + block - 1) / block) return MI_ERR_ARG;
    const uint32_t al = align_of(codec);
    if (t[0] % al) return MI_ERR_ARG;
    uint64_t at = MI_FRAME_HEADER;
    if (cap < at) return MI_ERR_CAPACITY;
    put_header(out, codec, block, p0, p1, original_size);
    if (nblocks == 0) {                                                 /* one empty last chunk */
        if (cap < at + 4 + (codec == MI_FRAME_LZ77 ? 4u : 0u)) return MI_ERR_CAPACITY;
        put32(out + at, 1u); at += 4;
        if (codec == MI_FRAME_LZ77) { put32(out + at, 0); at += 4; }
        *out_bytes = at;
        return MI_OK;
    }
    for (uint64_t b = 0; b < nblocks; ++b) {
        if (t[b + 1] < t[b] || (t[b + 1] % al)) return MI_ERR_ARG;
        const uint64_t nbits = t[b + 1] - t[b], nbytes = (nbits + 7) / 8;
        if (nbytes > CHUNK_MAX) return MI_ERR_ARG;
        const uint64_t need = 4 + (codec == MI_FRAME_LZ77 ? 4u : 0u) + nbytes;
        if (cap - at < need) return MI_ERR_CAPACITY;
        put32(out + at, (uint32_t)(b + 1 == nblocks) | ((uint32_t)nbytes << 1)); at += 4;
        if (codec == MI_FRAME_LZ77) { put32(out + at, (uint32_t)nbits); at += 4; }
        if (nbytes) take_bits(out + at, h_stream, t[b], nbits);
        at += nbytes;
    }
    *out_bytes = at;
    return MI_OK;
}

/* the tree part of a Huffman frame: counts nodes, checks the shape; returns the offset behind it or 0 */
static uint64_t walk_tree(const uint8_t *f, uint64_t n, uint64_t at, uint32_t *nodes, uint32_t depth)
{
    if (depth > 600 || at + 4 > n) return 0;
    if (get32(f + at) == 0xFFFFFFFFu) return at + 4;                    /* absent child (main.zig:168-173) */
    if (at + 5 > n || ++*nodes > 511) return 0;
    at += 5;                                                            /* u8 value, u32 frequency */
    at = walk_tree(f, n, at, nodes, depth + 1);
    if (!at) return 0;
    return walk_tree(f, n, at, nodes, depth + 1);
}

mi_status mi_frame_parse(const uint8_t *f, uint64_t n, mi_frame_info *info)
{
    if (!f || !info) return MI_ERR_ARG;
    memset(info, 0, sizeof *info);
    if (n < MI_FRAME_HEADER || memcmp(f, MI_FRAME_MAGIC, 4) != 0 || f[4] != 1) return MI_ERR_CORRUPT;
    info->codec = f[5]; info->block = get32(f + 8); info->p0 = get32(f + 12); info->p1 = get32(f + 16);
    info->original_size = get64(f + 20);
    uint64_t at = MI_FRAME_HEADER;
    if (info->codec == MI_FRAME_HUFFMAN) {
        if (at + 8 > n) return MI_ERR_CORRUPT;
        info->total_bits = get64(f + at); at += 8;
        uint32_t nodes = 0;
        at = walk_tree(f, n, at, &nodes, 0);
        if (!at || nodes != info->p0 || nodes < 3 || !(nodes & 1u)) return MI_ERR_CORRUPT;      /* a full binary tree: odd, >= 3 */
        if (at + 4 > n) return MI_ERR_CORRUPT;
        const uint64_t ntiles = get32(f + at); at += 4;
        if (ntiles && (ntiles + 1 > (n - at) / 8)) return MI_ERR_CORRUPT;
        if (ntiles) {
            uint64_t prev = get64(f + at);
            for (uint64_t i = 1; i <= ntiles; ++i) { const uint64_t v = get64(f + at + 8 * i); if (v < prev) return MI_ERR_CORRUPT; prev = v; }
            if (prev != info->total_bits) return MI_ERR_CORRUPT;
            at += 8 * (ntiles + 1);
        }
        info->nblocks = ntiles;
        for (;;) {                                                      /* payload chunks */
            if (at + 4 > n) return MI_ERR_CORRUPT;
            const uint32_t w = get32(f + at); at += 4;
            const uint64_t sz = w >> 1;
            if (sz > n - at) return MI_ERR_CORRUPT;
            info->stream_bytes += sz; at += sz;
            if (w & 1u) break;
        }
        if (info->stream_bytes != ((info->total_bits + 31) / 32) * 4) return MI_ERR_CORRUPT;
        return at == n ? MI_OK : MI_ERR_CORRUPT;
    }
    if (!is_block_codec(info->codec) || !info->block) return MI_ERR_CORRUPT;
    const uint64_t want = (info->original_size + info->block - 1) / info->block;
    const uint32_t al = align_of(info->codec);
    for (;;) {
        if (at + 4 > n) return MI_ERR_CORRUPT;
        const uint32_t w = get32(f + at); at += 4;
        const uint64_t sz = w >> 1;
        uint64_t nbits = sz * 8;
        if (info->codec == MI_FRAME_LZ77) {
            if (at + 4 > n) return MI_ERR_CORRUPT;
            nbits = get32(f + at); at += 4;
            if ((nbits + 7) / 8 != sz) return MI_ERR_CORRUPT;
        }
        if (sz > n - at || (nbits % al)) return MI_ERR_CORRUPT;
        at += sz;
        info->total_bits += nbits;
        if (want || sz) ++info->nblocks;                                /* an empty input carries one empty chunk, no block */
        if (w & 1u) break;
        if (info->nblocks > want) return MI_ERR_CORRUPT;
    }
    info->stream_bytes = (info->total_bits + 7) / 8;
    return (at == n && info->nblocks == want) ? MI_OK : MI_ERR_CORRUPT;
}

mi_status mi_frame_unpack_blocks(const uint8_t *f, uint64_t n, uint8_t *h_stream, uint64_t cap_bytes,
                                 uint64_t *t, uint64_t cap_blocks)
{
    mi_frame_info info;
    mi_status st = mi_frame_parse(f, n, &info);
    if (st) return st;
    if (!is_block_codec(info.codec) || !h_stream || !t) return MI_ERR_ARG;
    if (cap_blocks < info.nblocks + 1 || cap_bytes < info.stream_bytes + 8) return MI_ERR_CAPACITY;
    memset(h_stream, 0, info.stream_bytes + 8);
    uint64_t at = MI_FRAME_HEADER, bit = 0;
    t[0] = 0;
    for (uint64_t b = 0; b < info.nblocks; ++b) {
        const uint32_t w = get32(f + at); at += 4;
        const uint64_t sz = w >> 1;
        uint64_t nbits = sz * 8;
        if (info.codec == MI_FRAME_LZ77) { nbits = get32(f + at); at += 4; }
        if (nbits) place_bits(h_stream, bit, f + at, nbits);
        at += sz; bit += nbits;
        t[b + 1] = bit;
    }
    return MI_OK;
}

/* ---- Huffman ---------------------------------------------------------------------------------------------------- */
uint64_t mi_frame_bound_huffman(uint64_t total_bits, uint64_t ntiles)
{
    const uint64_t bytes = ((total_bits + 31) / 32) * 4;
    return MI_FRAME_HEADER + 8 + 511 * 5 + 512 * 4 + 4 + 8 * (ntiles + 1) + bytes + 4 * (bytes / (1u << 30) + 2) + 16;
}

static int ser_tree(const mi_huffman_tree *t, int id, uint8_t *out, uint64_t cap, uint64_t *at, uint32_t depth)
{
    if (depth > 600) return 0;
    if (id < 0) { if (cap - *at < 4) return 0; put32(out + *at, 0xFFFFFFFFu); *at += 4; return 1; }
    if (id > 510 || cap - *at < 5) return 0;
    out[*at] = t->value[id]; put32(out + *at + 1, t->frequency[id]); *at += 5;
    return ser_tree(t, t->left[id], out, cap, at, depth + 1) && ser_tree(t, t->right[id], out, cap, at, depth + 1);
}

mi_status mi_frame_pack_huffman(uint64_t original_size, const mi_huffman_tree *tree, uint32_t n_nodes,
                                const uint32_t *h_words, uint64_t total_bits, const uint64_t *h_tile_off, uint64_t ntiles,
                                uint8_t *out, uint64_t cap, uint64_t *out_bytes)
{
    if (!tree || !out || !out_bytes || n_nodes < 3 || n_nodes > 511 || (total_bits && !h_words) || (ntiles && !h_tile_off)) return MI_ERR_ARG;
    if (cap < mi_frame_bound_huffman(total_bits, ntiles)) return MI_ERR_CAPACITY;
    put_header(out, MI_FRAME_HUFFMAN, 0, n_nodes, 0, original_size);
    uint64_t at = MI_FRAME_HEADER;
    put64(out + at, total_bits); at += 8;
    if (!ser_tree(tree, (int)n_nodes - 1, out, cap, &at, 0)) return MI_ERR_ARG;
    put32(out + at, (uint32_t)ntiles); at += 4;
    if (ntiles) for (uint64_t i = 0; i <= ntiles; ++i) { put64(out + at, h_tile_off[i]); at += 8; }
    const uint64_t bytes = ((total_bits + 31) / 32) * 4;
    uint64_t done = 0;
    do {                                                                /* chunks of < 2^30 bytes: {last_block, size} + payload */
        const uint64_t sz = bytes - done < (1u << 30) ? bytes - done : (1u << 30);
        put32(out + at, (uint32_t)(done + sz == bytes) | ((uint32_t)sz << 1)); at += 4;
        if (sz) memcpy(out + at, (const uint8_t *)h_words + done, sz);
        at += sz; done += sz;
    } while (done < bytes);
    *out_bytes = at;
    return MI_OK;
}

/* pre-order image -> arrays; ids are handed out in POST-order so that the root receives the last one (the ABI's
 * convention: root = n_nodes - 1); codes are the paths (left 0, right 1: huffman.c:217-250) */
static int de_tree(const uint8_t *f, uint64_t *at, mi_huffman_tree *t, int *next, uint32_t code, uint32_t len)
{
    if (get32(f + *at) == 0xFFFFFFFFu) { *at += 4; return -1; }
    const uint8_t value = f[*at]; const uint32_t freq = get32(f + *at + 1);
    *at += 5;
    const int l = de_tree(f, at, t, next, code << 1, len + 1);
    const int r = de_tree(f, at, t, next, (code << 1) | 1u, len + 1);
    const int id = (*next)++;
    t->value[id] = value; t->frequency[id] = freq; t->left[id] = (int16_t)l; t->right[id] = (int16_t)r;
    if (l < 0 && r < 0) { t->code[value] = len <= 32 ? code : 0; t->length[value] = (uint8_t)(len > 255 ? 255 : len); }
    return id;
}

mi_status mi_frame_unpack_huffman(const uint8_t *f, uint64_t n, mi_huffman_tree *tree, uint32_t *n_nodes,
                                  uint32_t *h_words, uint64_t cap_words, uint64_t *total_bits,
                                  uint64_t *h_tile_off, uint64_t cap_tiles)
{
    mi_frame_info info;
    mi_status st = mi_frame_parse(f, n, &info);                         /* validates every offset used below */
    if (st) return st;
    if (info.codec != MI_FRAME_HUFFMAN || !tree || !n_nodes || !h_words || !total_bits) return MI_ERR_ARG;
    if (cap_words < info.stream_bytes / 4 + 1) return MI_ERR_CAPACITY;
    if (h_tile_off && cap_tiles < info.nblocks + 1) return MI_ERR_CAPACITY;
    memset(tree, 0, sizeof *tree);
    for (int i = 0; i < 511; ++i) tree->left[i] = tree->right[i] = -1;
    uint64_t at = MI_FRAME_HEADER + 8;
    int next = 0;
    const int root = de_tree(f, &at, tree, &next, 0, 0);
    if (root != (int)info.p0 - 1) return MI_ERR_CORRUPT;
    /* a node with exactly one child cannot come out of the reference's merge; the decoder would walk off it */
    for (int i = 0; i < next; ++i) if ((tree->left[i] < 0) != (tree->right[i] < 0)) return MI_ERR_CORRUPT;
    *n_nodes = (uint32_t)next;
    *total_bits = info.total_bits;
    at += 4;
    if (info.nblocks) {
        if (h_tile_off) for (uint64_t i = 0; i <= info.nblocks; ++i) h_tile_off[i] = get64(f + at + 8 * i);
        at += 8 * (info.nblocks + 1);
    }
    uint64_t done = 0;
    for (;;) {
        const uint32_t w = get32(f + at); at += 4;
        const uint64_t sz = w >> 1;
        memcpy((uint8_t *)h_words + done, f + at, sz);
        at += sz; done += sz;
        if (w & 1u) break;
    }
    h_words[done / 4] = 0;                                              /* the word the decoder peeks at */
    return MI_OK;
}
