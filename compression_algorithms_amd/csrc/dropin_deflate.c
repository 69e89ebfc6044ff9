/* dropin_deflate.c — algorithms/deflate entry points over the HIP path (see include/mi_deflate.h). */
#include <string.h>
#include <time.h>
#include "../../include/mi_deflate.h"
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

StateData compress(const char *input_filename)
{
    const char *slash = strrchr(input_filename, '/');
    const char *filename = slash ? slash + 1 : input_filename;  /* the reference requires a '/' (deflate.c:11) */
    StateData sd = { NULL, NULL, (char *)malloc(strlen(filename) + strlen(extension) + 1) };
    strcpy(sd.compressed_filename, filename); strcat(sd.compressed_filename, extension);
    uint64_t n; char *in = slurp(input_filename, &n);
    mi_ctx *ctx = dropin_ctx();
    mi_lz_params p = mi_lz_params_deflate();
    /* MI_DEFLATE_MODE=H: finish what lz77.c:279 leaves as a TODO — the same tokens, Huffman coded per block
     * (include/mi_codec.h "mode H").  Default: the reference's raw token bytes. */
    const char *mode = getenv("MI_DEFLATE_MODE");
    const int mode_h = mode && (mode[0] == 'H' || mode[0] == 'h');
    const uint64_t nblocks = mi_lz_num_blocks(n, &p);
    const uint64_t cap = (mode_h ? mi_deflate_h_bound_bytes(n, &p) : mi_lz_bound_bytes(n, &p)) + 64;
    uint8_t *out = (uint8_t *)malloc(cap); uint64_t *bits = (uint64_t *)malloc(8 * (nblocks + 1));
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    mi_status st = mode_h ? mi_deflate_h_encode(ctx, &p, (const uint8_t *)in, n, out, cap, bits)
                          : mi_lz_encode(ctx, &p, (const uint8_t *)in, n, out, cap, bits);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (st != MI_OK) { fprintf(stderr, "compress: %s\n", mi_status_str(st)); exit(1); }
    FILE *f = fopen(sd.compressed_filename, "wb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", sd.compressed_filename); exit(1); }   /* deflate.c:30-34 */
    fwrite(out, 1, bits[nblocks] / 8, f); fclose(f);
    char *idx = (char *)malloc(strlen(sd.compressed_filename) + 5);
    strcpy(idx, sd.compressed_filename); strcat(idx, ".idx");
    f = fopen(idx, "wb");
    /* side-car: original size, block size | mode H flag << 32, block count, then the per-block bit offsets */
    if (f) { uint64_t hdr[3] = { n, p.block | ((uint64_t)mode_h << 32), nblocks }; fwrite(hdr, 8, 3, f); fwrite(bits, 8, nblocks + 1, f); fclose(f); }
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("MB/s: %f\n", (double)n / (1024 * 1024) / sec);     /* deflate.c:65 */
    free(idx); free(in); free(out); free(bits);
    return sd;
}

void decompress(StateData *sd, const char *input_filename)
{
    const char *name = (sd && sd->compressed_filename) ? sd->compressed_filename : input_filename;
    char *idx = (char *)malloc(strlen(name) + 5); strcpy(idx, name); strcat(idx, ".idx");
    uint64_t isz; char *ib = slurp(idx, &isz);
    /* the side-car is a file: nothing in it is trusted before it is checked against its own size and the stream's */
    if (isz < 24) { fprintf(stderr, "decompress: %s is truncated\n", idx); exit(1); }
    const uint64_t *h = (const uint64_t *)ib;
    const uint64_t n = h[0], nblocks = h[2];
    const uint32_t block = (uint32_t)h[1];
    const int mode_h = (int)((h[1] >> 32) & 1u);
    if (block < 1 || block > BUFFER_SIZE || (h[1] >> 33) || nblocks != (n + block - 1) / block ||
        nblocks > (isz - 24) / 8 || isz < 24 + 8 * (nblocks + 1)) {
        fprintf(stderr, "decompress: %s is corrupt or truncated\n", idx); exit(1);
    }
    uint64_t csz; char *cb = slurp(name, &csz);
    mi_lz_params p = mi_lz_params_deflate(); p.block = block;
    if (mi_validate_block_table(h + 3, nblocks, csz, mode_h ? 32u : 8u) != MI_OK) {
        fprintf(stderr, "decompress: the block table of %s does not fit %s\n", idx, name); exit(1);
    }
    uint8_t *out = (uint8_t *)malloc(n ? n : 1);
    mi_status st = mode_h ? mi_deflate_h_decode(dropin_ctx(), &p, (const uint8_t *)cb, csz, h + 3, out, n)
                          : mi_lz_decode(dropin_ctx(), &p, (const uint8_t *)cb, csz, h + 3, out, n);
    if (st != MI_OK) { fprintf(stderr, "decompress: %s\n", mi_status_str(st)); exit(1); }
    char *on = (char *)malloc(strlen(name) + 6); strcpy(on, name); strcat(on, ".orig");
    FILE *f = fopen(on, "wb");
    if (!f) { fprintf(stderr, "Error: could not open file %s\n", on); exit(1); }
    fwrite(out, 1, n, f); fclose(f);
    (void)nblocks;
    free(on); free(out); free(cb); free(ib); free(idx);
}

void lz77_compress(const char *in, uint64_t n, char *out, uint64_t *out_n, HashTableArray *table)
{
    (void)table;
    mi_lz_params p = mi_lz_params_deflate();
    if (n > p.block) { fprintf(stderr, "lz77_compress: a block is at most %u bytes\n", p.block); exit(1); }
    uint64_t bits[2];
    uint8_t *tmp = (uint8_t *)malloc(2 * n + 128);
    mi_status st = mi_lz_encode(dropin_ctx(), &p, (const uint8_t *)in, n, tmp, 2 * n + 128, bits);
    if (st != MI_OK) { fprintf(stderr, "lz77_compress: %s\n", mi_status_str(st)); exit(1); }
    *out_n = bits[n ? 1 : 0] / 8;
    memcpy(out, tmp, *out_n);
    free(tmp);
}
