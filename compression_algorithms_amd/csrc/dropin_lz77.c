/* dropin_lz77.c — algorithms/lz77 entry points over the HIP path (see include/mi_lz77.h). */
#include <string.h>
#include "../../include/mi_lz77.h"
#include "dropin_common.h"

static uint32_t g_wbits = WINDOW_BITS;
void mi_lz77_set_window_bits(uint32_t w) { g_wbits = w; }
static uint32_t cur_wbits(void)
{
    const char *e = getenv("MI_LZ77_WINDOW_BITS");
    return e ? (uint32_t)atoi(e) : g_wbits;
}

/* block size: MI_LZ77_BLOCK (default 65 536).  Up to 1 MiB (multiples of 256 above 64 KiB): the window then slides inside a
 * block as in the reference, and a buffer of at most one block yields the reference's whole-buffer stream bit for bit. */
static uint32_t cur_block(void)
{
    const char *e = getenv("MI_LZ77_BLOCK");
    long v = e ? atol(e) : (long)MI_LZ77_BLOCK;
    if (v < 1 || v > (1 << 20) || (v > 65536 && (v & 255))) { fprintf(stderr, "MI_LZ77_BLOCK: 1..65536, or a multiple of 256 up to 1048576\n"); exit(1); }
    return (uint32_t)v;
}

uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
uint64_t max(uint64_t a, uint64_t b) { return a > b ? a : b; }

uint32_t hash(uint32_t pattern)                                     /* lz77.c:13-41: murmur3-style mix, % TABLE_SIZE */
{
    uint32_t k = pattern * 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    uint32_t h = (k << 13) | (k >> 19);
    h = h * 5u + 0xe6546b64u;
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13; h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h % (1u << (cur_wbits() + 6));
}

void init_hash_table(HashTableArray *table)                         /* lz77.c:43-53 */
{
    table->buckets = (ArrayNode *)calloc((size_t)TABLE_SIZE, sizeof(ArrayNode));
    if (!table->buckets) { fprintf(stderr, "Error: could not allocate memory for hash table\n"); exit(1); }
    memset(table->bucket_indices, 0, sizeof table->bucket_indices);
    table->current_idx = 0;
    table->is_full = false;
}

/* lz77.h:34-35 (lz77.c:55-108): the reference's per-entry table operations, on the host table init_hash_table allocates.
 * Host helpers like write_bit: the encoder never calls them (its table is never materialised) — they are here
 * so that a caller written against lz77.h links completely, and they are what tests/test_dropin.py checks the oracle's
 * literal table against.  One deliberate difference: the reference probes without wrapping and without a bound (lz77.c:61,
 * :97-105), so a run of occupied buckets that reaches the end of the array makes it write past it; here the probe stops at
 * the last bucket and the call fails loudly instead. */
static void table_overrun(const char *who)
{
    fprintf(stderr, "%s: probe ran past bucket TABLE_SIZE-1 (the reference writes out of bounds here, lz77.c:61)\n", who);
    exit(1);
}

void insert_hash_table(HashTableArray *table, uint32_t pattern, uint64_t index)
{
    const uint32_t ring = 1u << WINDOW_BITS;
    uint64_t b = hash(pattern) % (uint32_t)TABLE_SIZE;
    while (b < (uint64_t)TABLE_SIZE && table->buckets[b].is_set) ++b;          /* first free bucket at or after the home */
    if (b >= (uint64_t)TABLE_SIZE) table_overrun("insert_hash_table");
    table->buckets[b].pattern = pattern; table->buckets[b].index = index; table->buckets[b].is_set = true;
    if (table->is_full) {                                                      /* FIFO retirement: whatever sits in the bucket */
        ArrayNode *old = &table->buckets[table->bucket_indices[table->current_idx]];   /* recorded W insertions ago goes */
        old->pattern = 0; old->index = 0; old->is_set = false;
    }
    table->bucket_indices[table->current_idx] = (uint32_t)b;
    if (++table->current_idx >= ring - 1u) table->is_full = true;              /* trips after insertion #(W - 1): SURVEY A.1.2 */
    table->current_idx %= ring;
}

uint64_t find(HashTableArray *table, uint32_t pattern)
{
    for (uint64_t b = hash(pattern) % (uint32_t)TABLE_SIZE; b < (uint64_t)TABLE_SIZE; ++b) {
        if (!table->buckets[b].is_set) return UINT64_MAX;                      /* a hole ends the probe: no tombstones */
        if (table->buckets[b].pattern == pattern) return table->buckets[b].index;
    }
    table_overrun("find");
    return UINT64_MAX;
}

void init_bitstream(BitStream *stream, uint8_t *buffer) { stream->data = buffer; stream->bit_index = 0; }   /* lz77.c:139-142 */

void write_bit(BitStream *stream, bool bit)                         /* lz77.c:144-156: sets OR clears (the buffer need not be zeroed) */
{
    const uint64_t byte = stream->bit_index / 8, off = stream->bit_index % 8;
    if (bit) stream->data[byte] |= (uint8_t)(1u << off); else stream->data[byte] &= (uint8_t)~(1u << off);
    ++stream->bit_index;
}

bool read_bit(BitStream *stream)                                    /* lz77.c:159-167 */
{
    const bool b = (stream->data[stream->bit_index / 8] >> (stream->bit_index % 8)) & 1u;
    ++stream->bit_index;
    return b;
}

void write_bits(BitStream *stream, uint64_t value, uint64_t num_bits)   /* lz77.c:169-174: LSB first */
{
    for (uint64_t b = 0; b < num_bits; ++b) write_bit(stream, b < 64 && ((value >> b) & 1u));
}

uint64_t read_bits(BitStream *stream, uint64_t num_bits)            /* lz77.c:176-184 */
{
    uint64_t v = 0;
    for (uint64_t b = 0; b < num_bits; ++b) if (read_bit(stream) && b < 64) v |= 1ull << b;
    return v;
}

void print_bit_string(const char *buffer, uint64_t size)            /* lz77.c:110-119 */
{
    for (uint64_t i = 0; i < size; ++i) {
        for (int bit = 7; bit >= 0; --bit) printf("%d", (buffer[i] >> bit) & 1);
    }
    printf("\n");
}

char *read_input_buffer(const char *filename, uint64_t *size)      /* lz77.c:121-137 */
{
    FILE *f = fopen(filename, "rb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", filename); exit(1); }
    fseek(f, 0, SEEK_END); *size = (uint64_t)ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc(*size + 1);
    if (fread(buf, 1, *size, f) != *size) { fprintf(stderr, "Error: short read on %s\n", filename); exit(1); }
    fclose(f);
    return buf;
}

bool check_buffer_equivalence(const char *a, const char *b, uint64_t size)     /* lz77.c:379-392 */
{
    uint64_t diff = 0;
    for (uint64_t i = 0; i < size; ++i) diff += a[i] != b[i];
    printf("Number of differences: %lu\n", (unsigned long)diff);
    return diff == 0;
}

/* `data` holds the stream alone (bit_index/8 + 1 bytes, like lz77.c:341-342); the per-block bit offsets a parallel
 * decoder needs are registered out of band (dropin_common.h) under the data pointer. */
BitStream *lz77_compress(const char *buffer, uint64_t size)
{
    mi_ctx *ctx = dropin_ctx();
    mi_lz_params p = mi_lz_params_lz77(cur_wbits());
    p.block = cur_block();
    const uint64_t nblocks = mi_lz_num_blocks(size, &p);
    const uint64_t cap = mi_lz_bound_bytes(size, &p) + 64;
    uint8_t *data = (uint8_t *)calloc(1, cap + 8);
    uint64_t *bits = (uint64_t *)malloc(8 * (nblocks + 1));
    BitStream *s = (BitStream *)malloc(sizeof *s);
    if (!data || !bits || !s) { fprintf(stderr, "lz77_compress: out of memory\n"); exit(1); }
    mi_status st = mi_lz_encode(ctx, &p, (const uint8_t *)buffer, size, data, cap, bits);
    if (st != MI_OK) { fprintf(stderr, "lz77_compress: %s\n", mi_status_str(st)); exit(1); }
    s->bit_index = bits[nblocks];
    s->data = (uint8_t *)realloc(data, s->bit_index / 8 + 1);  /* lz77.c:341-342 */
    if (!s->data) { fprintf(stderr, "lz77_compress: out of memory\n"); exit(1); }
    dropin_side_put(s->data, s->bit_index, size, ((uint64_t)p.wbits << 32) | p.block, bits, nblocks + 1);
    free(bits);
    return s;
}

/* lz77.c:185-262: the brute-force parser; a whole-buffer stream, registered with block 0 so that lz77_decompress knows */
BitStream *lz77_compress_old(const char *buffer, uint64_t size)
{
    mi_ctx *ctx = dropin_ctx();
    const uint32_t wbits = cur_wbits();
    const uint64_t cap = mi_lz77_old_bound_bytes(size);
    uint8_t *data = (uint8_t *)calloc(1, cap + 8);
    BitStream *s = (BitStream *)malloc(sizeof *s);
    if (!data || !s) { fprintf(stderr, "lz77_compress_old: out of memory\n"); exit(1); }
    uint64_t bits = 0;
    mi_status st = mi_lz77_old_encode(ctx, wbits, 4, (const uint8_t *)buffer, size, data, cap, &bits);
    if (st != MI_OK) { fprintf(stderr, "lz77_compress_old: %s\n", mi_status_str(st)); exit(1); }
    s->bit_index = bits;
    s->data = (uint8_t *)realloc(data, s->bit_index / 8 + 1);  /* lz77.c:258-259 */
    if (!s->data) { fprintf(stderr, "lz77_compress_old: out of memory\n"); exit(1); }
    const uint64_t table[2] = {0, bits};
    dropin_side_put(s->data, s->bit_index, size, (uint64_t)wbits << 32, table, 2);
    return s;
}

char *lz77_decompress(BitStream *cs, uint64_t size, uint64_t *decompressed_size)
{
    mi_ctx *ctx = dropin_ctx();
    const uint64_t total = cs->bit_index;
    cs->bit_index = 0;                                          /* lz77.c:355-356 */
    char *out = (char *)malloc(size ? size : 1);
    mi_lz_params p = mi_lz_params_lz77(cur_wbits());
    p.block = cur_block();
    uint64_t one[2] = {0, total};
    const uint64_t *bits = one;
    if (size > p.block) {
        /* more than one block: the offsets are the ones lz77_compress registered for this buffer */
        dropin_side e;
        if (!dropin_side_get(cs->data, total, size, &e)) {
            fprintf(stderr, "lz77_decompress: no block table is registered for this stream: it was not produced by this process's "
                            "lz77_compress, or it was released (mi_lz77_release).  Streams that cross processes are framed: mi_frame.h\n");
            exit(1);
        }
        if ((uint32_t)e.aux == 0) {                             /* a whole-buffer stream (lz77_compress_old) */
            mi_status sw = mi_lz77_whole_decode(ctx, (uint32_t)(e.aux >> 32), 4, cs->data, total / 8 + 1, total, (uint8_t *)out, size);
            if (sw != MI_OK) { fprintf(stderr, "lz77_decompress: %s\n", mi_status_str(sw)); exit(1); }
            cs->bit_index = total;
            *decompressed_size = size;
            return out;
        }
        p.wbits = (uint32_t)(e.aux >> 32); p.tbits = p.wbits + 6; p.block = (uint32_t)e.aux;
        if (e.count != mi_lz_num_blocks(size, &p) + 1) { fprintf(stderr, "lz77_decompress: block table does not match the size\n"); exit(1); }
        bits = e.table;
    }
    mi_status st = mi_lz_decode(ctx, &p, cs->data, total / 8 + 1, bits, (uint8_t *)out, size);
    if (st != MI_OK) { fprintf(stderr, "lz77_decompress: %s\n", mi_status_str(st)); exit(1); }
    cs->bit_index = total;
    *decompressed_size = size;
    return out;
}

/* frees a stream lz77_compress returned (data, struct) and its out-of-band block table.  Reference-style callers that
 * free ->data and the struct themselves (lz77/main.c:66-67) leave the table registered until the address is reused. */
void mi_lz77_release(BitStream *stream)
{
    if (!stream) return;
    dropin_side_drop(stream->data);
    free(stream->data);
    free(stream);
}

uint64_t mi_lz77_registered_streams(void) { return (uint64_t)dropin_side_count(); }
