/* dropin_deflate.c — algorithms/deflate entry points over the HIP path (see include/mi_deflate.h). */
#include <string.h>
#include <time.h>
#include "../../include/mi_deflate.h"
#include "../../include/mi_frame.h"
#include "dropin_common.h"

static const char *extension = ".deflate";                      /* deflate.h:10 */

static char *slurp(const char *name, uint64_t *size)
{
    FILE *f = fopen(name, "rb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", name); exit(1); }   /* deflate.c:25-28 */
    fseek(f, 0, SEEK_END); *size = (uint64_t)ftell(f); fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc(*size + 1);
    if (fread(b, 1, *size, f) != *size) { fprintf(stderr, "Error: short read on %s\n", name); exit(1); }
    fclose(f);
    return b;
}

uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
uint64_t max(uint64_t a, uint64_t b) { return a > b ? a : b; }

uint32_t hash(uint32_t pattern)                                     /* deflate/lz77.c:14-42 */
{
    uint32_t k = pattern * 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    uint32_t h = (k << 13) | (k >> 19);
    h = h * 5u + 0xe6546b64u;
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13; h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h % TABLE_SIZE;
}

void init_hash_table(HashTableArray *table)                         /* deflate/lz77.c:44-67 */
{
    table->buckets.patterns = (uint32_t *)calloc(TABLE_SIZE, sizeof(uint32_t));
    table->buckets.indices = (uint64_t *)calloc(TABLE_SIZE, sizeof(uint64_t));
    table->buckets.is_set = (bool *)calloc(TABLE_SIZE, sizeof(bool));
    memset(table->bucket_indices, 0, sizeof table->bucket_indices);
    table->current_idx = 0;
    table->is_full = false;
}

/* deflate/lz77.h:32-33 (deflate/lz77.c:77-174): the per-entry table operations on the host table above; host helpers, the
 * encoder never calls them.  insert wraps modulo TABLE_SIZE as the reference's does (lz77.c:99-101); find does not wrap
 * (lz77.c:163-169) and — where the reference reads patterns[TABLE_SIZE] — stops at the last bucket: "not found", the
 * definition the oracle and the kernels share. */
void insert_hash_table(HashTableArray *table, uint32_t pattern, uint64_t index)
{
    uint32_t b = hash(pattern);
    for (uint32_t probes = 0; table->buckets.is_set[b]; ++probes) {
        if (probes >= TABLE_SIZE) { fprintf(stderr, "insert_hash_table: table full\n"); exit(1); }   /* the reference spins forever */
        b = (b + 1u) % TABLE_SIZE;
    }
    table->buckets.patterns[b] = pattern; table->buckets.indices[b] = index; table->buckets.is_set[b] = true;
    if (table->is_full) {                                         /* FIFO retirement of the bucket recorded W insertions ago */
        const uint32_t old = table->bucket_indices[table->current_idx];
        table->buckets.patterns[old] = 0; table->buckets.indices[old] = 0; table->buckets.is_set[old] = false;
    }
    table->bucket_indices[table->current_idx] = b;
    if (++table->current_idx >= WINDOW_SIZE - 1) table->is_full = true;
    table->current_idx %= WINDOW_SIZE;
}

uint64_t find(HashTableArray *table, uint32_t pattern)
{
    for (uint32_t b = hash(pattern); b < TABLE_SIZE; ++b) {
        if (!table->buckets.is_set[b]) return UINT64_MAX;         /* a hole ends the probe: no tombstones */
        if (table->buckets.patterns[b] == pattern) return table->buckets.indices[b];
    }
    return UINT64_MAX;
}

void write_literal(char *buffer, char c, uint64_t *at) { buffer[(*at)++] = 0; buffer[(*at)++] = c; }          /* deflate/lz77.c:176-184 */
void write_length_distance(char *buffer, uint8_t length, uint16_t distance, uint64_t *at)              /* deflate/lz77.c:186-197 */
{
    buffer[(*at)++] = 1; buffer[(*at)++] = (char)(distance & 0xFF); buffer[(*at)++] = (char)(distance >> 8); buffer[(*at)++] = (char)length;
}

/* ---- deflate/huffman.h + deflate.h:19-21: host helpers of the entropy stage the reference sketches -------------------------- */
void init_bitwriter(BitWriter *w, uint64_t buffer_size)                                     /* deflate/huffman.c:7-13: size in BYTES, zeroed */
{
    w->buffer = (uint32_t *)calloc(buffer_size ? buffer_size : 4, 1);
    w->word_idx = 0; w->bit_idx = 0; w->buffer_size = buffer_size;
}

void write_bits(BitWriter *w, uint32_t bits, uint8_t length)                                  /* deflate/huffman.c:16-46 */
{
    /* MSB first: the next free bit of the current word is bit 31 - bit_idx.  `bits` beyond `length` are the caller's to
     * keep zero (the reference ORs them in unmasked when the code fits the word exactly, masked otherwise: same result
     * for well-formed input) */
    if (!length) return;
    const uint32_t room = 32u - (uint32_t)w->bit_idx;
    const uint32_t v = length >= 32 ? bits : (bits & ((1u << length) - 1u));
    if (length <= room) {
        w->buffer[w->word_idx] |= v << (room - length);
        w->bit_idx += length;
        if (w->bit_idx == 32) { w->bit_idx = 0; ++w->word_idx; }
    } else {
        const uint32_t spill = length - room;
        w->buffer[w->word_idx++] |= v >> spill;
        w->buffer[w->word_idx] |= v << (32u - spill);
        w->bit_idx = spill;
    }
}

void append_huffman_tree_literal(uint32_t *frequencies, char literal) { ++frequencies[(uint8_t)literal]; }
void append_huffman_tree_pair(uint32_t *frequencies, uint16_t offset)
{
    /* bin 256 + (leading zeros of the 16-bit offset): deflate/huffman.c:60-61; offset 0 is the caller's bug there (clz(0)) and
     * lands in the last bin here */
    const unsigned lz16 = offset ? (unsigned)__builtin_clz((unsigned)offset) - 16u : 16u;
    ++frequencies[256 + (lz16 > 29u ? 29u : lz16)];
}

void gather_codes(MinHeapNode *root, uint16_t code, uint8_t length, uint16_t *codes, uint8_t *code_lengths)
{
    if (!root) return;
    if (!root->left && !root->right) { codes[root->data] = code; code_lengths[root->data] = length; return; }   /* a leaf keeps the path */
    gather_codes(root->left, (uint16_t)(code << 1), (uint8_t)(length + 1), codes, code_lengths);           /* left = 0 */
    gather_codes(root->right, (uint16_t)((code << 1) | 1u), (uint8_t)(length + 1), codes, code_lengths);    /* right = 1 */
}

void init_huffman_node(HuffmanNode *node) { node->left = NULL; node->right = NULL; node->value = 0; node->frequency = 0; }
void destroy_huffman_node(HuffmanNode *node)
{
    if (node->left) { destroy_huffman_node(node->left); free(node->left); node->left = NULL; }
    if (node->right) { destroy_huffman_node(node->right); free(node->right); node->right = NULL; }
}
bool compare_huffman_node(const HuffmanNode *a, const HuffmanNode *b) { return a->frequency < b->frequency; }

StateData compress(const char *input_filename)
{
    const char *slash = strrchr(input_filename, '/');
    const char *filename = slash ? slash + 1 : input_filename;  /* the reference requires a '/' (deflate.c:11) */
    StateData sd = { NULL, NULL, (char *)malloc(strlen(filename) + strlen(extension) + 1) };
    strcpy(sd.compressed_filename, filename); strcat(sd.compressed_filename, extension);
    uint64_t n; char *in = slurp(input_filename, &n);
    mi_multi *mm = dropin_multi();                              /* MI_CODEC_DEVICES=0,1,...: the blocks spread over several GPUs */
    mi_ctx *ctx = mm ? NULL : dropin_ctx();
    mi_lz_params p = mi_lz_params_deflate();
    /* MI_DEFLATE_MODE=H: finish what lz77.c:279 leaves as a TODO — the same tokens, Huffman coded per block
     * (include/mi_codec.h "mode H").  Default: the reference's raw token bytes. */
    const char *mode = getenv("MI_DEFLATE_MODE");
    const int mode_h = mode && (mode[0] == 'H' || mode[0] == 'h');
    const uint64_t nblocks = mi_lz_num_blocks(n, &p);
    const uint64_t cap = (mode_h ? mi_deflate_h_bound_bytes(n, &p) : mi_lz_bound_bytes(n, &p)) + 64;
    uint8_t *out = (uint8_t *)malloc(cap); uint64_t *bits = (uint64_t *)malloc(8 * (nblocks + 1));
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    /* the reference's block loop (deflate.c:47-63), all blocks at once; with MI_CODEC_DEVICES each device takes a contiguous
     * range of them and the streams meet on the first device — the same bytes either way */
    mi_status st = mm ? (mode_h ? mi_deflate_h_encode_multi(mm, &p, (const uint8_t *)in, n, out, cap, bits)
                                : mi_lz_encode_multi(mm, &p, (const uint8_t *)in, n, out, cap, bits))
                      : (mode_h ? mi_deflate_h_encode(ctx, &p, (const uint8_t *)in, n, out, cap, bits)
                                : mi_lz_encode(ctx, &p, (const uint8_t *)in, n, out, cap, bits));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (st != MI_OK) { fprintf(stderr, "compress: %s\n", mi_status_str(st)); exit(1); }
    /* Mode T keeps the reference's file byte for byte (the bare token concatenation of deflate.c:56) and puts what a
     * decoder needs into a side-car.  Mode H is this build's own stream, so it is written self-describing (mi_frame.h:
     * {last_block, size} + record per block) with no side-car; MI_DEFLATE_FRAMED=1 frames mode T the same way. */
    const char *fr = getenv("MI_DEFLATE_FRAMED");
    const int framed = mode_h || (fr && fr[0] == '1');
    FILE *f = fopen(sd.compressed_filename, "wb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", sd.compressed_filename); exit(1); }   /* deflate.c:30-34 */
    char *idx = (char *)malloc(strlen(sd.compressed_filename) + 5);
    strcpy(idx, sd.compressed_filename); strcat(idx, ".idx");
    if (framed) {
        const uint64_t fcap = mi_frame_bound_blocks(nblocks, bits[nblocks] / 8);
        uint8_t *frame = (uint8_t *)malloc(fcap); uint64_t fn = 0;
        if (!frame) { fprintf(stderr, "compress: out of memory\n"); exit(1); }
        st = mi_frame_pack_blocks(mode_h ? MI_FRAME_DEFLATE_H : MI_FRAME_DEFLATE_T, p.block, p.wbits, p.lbits, n, out, bits, nblocks, frame, fcap, &fn);
        if (st != MI_OK) { fprintf(stderr, "compress: %s\n", mi_status_str(st)); exit(1); }
        fwrite(frame, 1, fn, f); fclose(f);
        free(frame);
        remove(idx);                                            /* a stale side-car of an earlier mode-T run must not shadow the frame */
    } else {
        fwrite(out, 1, bits[nblocks] / 8, f); fclose(f);
        f = fopen(idx, "wb");
        /* side-car: original size, block size | mode H flag << 32, block count, then the per-block bit offsets */
        if (f) { uint64_t hdr[3] = { n, p.block | ((uint64_t)mode_h << 32), nblocks }; fwrite(hdr, 8, 3, f); fwrite(bits, 8, nblocks + 1, f); fclose(f); }
    }
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("MB/s: %f\n", (double)n / (1024 * 1024) / sec);     /* deflate.c:65 */
    free(idx); free(in); free(out); free(bits);
    return sd;
}

void decompress(StateData *sd, const char *input_filename)
{
    const char *name = (sd && sd->compressed_filename) ? sd->compressed_filename : input_filename;
    uint64_t csz; char *cb = slurp(name, &csz);
    uint8_t *out = NULL; uint64_t n = 0;
    mi_status st;
    if (csz >= 4 && memcmp(cb, MI_FRAME_MAGIC, 4) == 0) {
        /* a framed, self-describing file (mode H, or MI_DEFLATE_FRAMED=1): no side-car */
        mi_frame_info fi;
        st = mi_frame_parse((const uint8_t *)cb, csz, &fi);
        if (st != MI_OK || (fi.codec != MI_FRAME_DEFLATE_H && fi.codec != MI_FRAME_DEFLATE_T)) { fprintf(stderr, "decompress: %s is not a deflate frame\n", name); exit(1); }
        uint8_t *stream = (uint8_t *)malloc(fi.stream_bytes + 8); uint64_t *t = (uint64_t *)malloc(8 * (fi.nblocks + 1));
        n = fi.original_size; out = (uint8_t *)malloc(n ? n : 1);
        if (!stream || !t || !out) { fprintf(stderr, "decompress: out of memory\n"); exit(1); }
        st = mi_frame_unpack_blocks((const uint8_t *)cb, csz, stream, fi.stream_bytes + 8, t, fi.nblocks + 1);
        mi_lz_params p = mi_lz_params_deflate(); p.block = fi.block; p.wbits = fi.p0; p.lbits = fi.p1;
        if (st == MI_OK) st = fi.codec == MI_FRAME_DEFLATE_H ? mi_deflate_h_decode(dropin_ctx(), &p, stream, fi.stream_bytes, t, out, n)
                                                             : mi_lz_decode(dropin_ctx(), &p, stream, fi.stream_bytes, t, out, n);
        free(stream); free(t);
    } else {
    char *idx = (char *)malloc(strlen(name) + 5); strcpy(idx, name); strcat(idx, ".idx");
    uint64_t isz; char *ib = slurp(idx, &isz);
    /* the side-car is a file: nothing in it is trusted before it is checked against its own size and the stream's */
    if (isz < 24) { fprintf(stderr, "decompress: %s is truncated\n", idx); exit(1); }
    const uint64_t *h = (const uint64_t *)ib;
    const uint64_t nblocks = h[2];
    n = h[0];
    const uint32_t block = (uint32_t)h[1];
    const int mode_h = (int)((h[1] >> 32) & 1u);
    if (block < 1 || block > BUFFER_SIZE || (h[1] >> 33) || nblocks != (n + block - 1) / block ||
        nblocks > (isz - 24) / 8 || isz < 24 + 8 * (nblocks + 1)) {
        fprintf(stderr, "decompress: %s is corrupt or truncated\n", idx); exit(1);
    }
    mi_lz_params p = mi_lz_params_deflate(); p.block = block;
    if (mi_validate_block_table(h + 3, nblocks, csz, mode_h ? 32u : 8u) != MI_OK) {
        fprintf(stderr, "decompress: the block table of %s does not fit %s\n", idx, name); exit(1);
    }
    out = (uint8_t *)malloc(n ? n : 1);
    st = mode_h ? mi_deflate_h_decode(dropin_ctx(), &p, (const uint8_t *)cb, csz, h + 3, out, n)
                : mi_lz_decode(dropin_ctx(), &p, (const uint8_t *)cb, csz, h + 3, out, n);
    free(ib); free(idx);
    }
    if (st != MI_OK) { fprintf(stderr, "decompress: %s\n", mi_status_str(st)); exit(1); }
    char *on = (char *)malloc(strlen(name) + 6); strcpy(on, name); strcat(on, ".orig");
    FILE *f = fopen(on, "wb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", on); exit(1); }
    fwrite(out, 1, n, f); fclose(f);
    free(on); free(out); free(cb);
}

void lz77_compress(const char *in, uint64_t n, char *out, uint64_t *out_n, HashTableArray *table)
{
    (void)table;
    mi_lz_params p = mi_lz_params_deflate();
    if (n > p.block) { fprintf(stderr, "lz77_compress: a block is at most %u bytes\n", p.block); exit(1); }
    uint64_t bits[2];
    uint8_t *tmp = (uint8_t *)malloc(2 * n + 128);
    if (!tmp) { fprintf(stderr, "lz77_compress: out of memory\n"); exit(1); }
    mi_status st = mi_lz_encode(dropin_ctx(), &p, (const uint8_t *)in, n, tmp, 2 * n + 128, bits);
    if (st != MI_OK) { fprintf(stderr, "lz77_compress: %s\n", mi_status_str(st)); exit(1); }
    *out_n = bits[n ? 1 : 0] / 8;
    memcpy(out, tmp, *out_n);
    free(tmp);
}
