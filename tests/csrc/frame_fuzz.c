/* Sanitizer harness for the host-only C of the boundary (compression_algorithms_amd/csrc/frame.c): built with
 * -fsanitize=address,undefined by tests/test_frame_sanitize.py — the analogue of the reference's `make sanitize`
 * (algorithms/{lz77,huffman,deflate}/Makefile) for the code that runs on the CPU here.  No GPU, no libmi_codec.so.
 *   1. random block tables and streams are packed, parsed and unpacked: the stream and the table must come back;
 *   2. every frame is then corrupted (random byte flips, truncation, appended bytes) and parsed / unpacked again:
 *      the calls must return (MI_OK or an error) without the sanitizers firing.
 * Exit code 0 = everything held; output on stdout names the first violation otherwise. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi_codec.h"
#include "mi_frame.h"

static uint64_t rs = 88172645463325252ull;
static uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

static int one_case(uint32_t codec, uint64_t nblocks, uint32_t block)
{
    uint64_t *bits = (uint64_t *)malloc((nblocks + 1) * 8);
    uint64_t pos = 0;
    const uint32_t align = codec == MI_FRAME_LZ77 ? 1u : codec == MI_FRAME_DEFLATE_H ? 32u : codec == MI_FRAME_FSE ? 32u : 16u;
    for (uint64_t b = 0; b < nblocks; ++b) {
        bits[b] = pos;
        uint64_t len = rnd() % (9ull * block + 64);              /* bits */
        len -= len % align;
        if (len == 0) len = align;
        pos += len;
    }
    bits[nblocks] = pos;
    const uint64_t sbytes = (pos + 7) / 8;
    uint8_t *stream = (uint8_t *)malloc(sbytes + 16);
    for (uint64_t i = 0; i < sbytes; ++i) stream[i] = (uint8_t)rnd();
    if (pos & 7) stream[sbytes - 1] &= (uint8_t)((1u << (pos & 7)) - 1u);         /* pad bits are zero */
    const uint64_t cap = mi_frame_bound_blocks(nblocks, sbytes);
    uint8_t *frame = (uint8_t *)malloc(cap + 64);
    uint64_t fbytes = 0;
    const uint64_t original = nblocks ? (nblocks - 1) * (uint64_t)block + 1 + rnd() % block : 0;
    mi_status st = mi_frame_pack_blocks(codec, block, 15, 5, original, stream, bits, nblocks, frame, cap, &fbytes);
    if (st != MI_OK) { printf("pack failed: codec %u nblocks %llu status %d\n", codec, (unsigned long long)nblocks, (int)st); return 1; }
    mi_frame_info info;
    st = mi_frame_parse(frame, fbytes, &info);
    if (st != MI_OK || info.nblocks != nblocks || info.original_size != original) { printf("parse of a good frame failed (%d)\n", (int)st); return 1; }
    uint8_t *s2 = (uint8_t *)malloc(info.stream_bytes + 16);
    uint64_t *b2 = (uint64_t *)malloc((info.nblocks + 1) * 8);
    st = mi_frame_unpack_blocks(frame, fbytes, s2, info.stream_bytes + 8, b2, info.nblocks + 1);
    if (st != MI_OK || memcmp(b2, bits, (nblocks + 1) * 8) != 0 || memcmp(s2, stream, sbytes) != 0) { printf("round trip differs: codec %u nblocks %llu\n", codec, (unsigned long long)nblocks); return 1; }
    /* corruptions: whatever comes back, nothing may be read or written out of bounds */
    for (int k = 0; k < 24; ++k) {
        uint64_t fb = fbytes;
        uint8_t *bad = (uint8_t *)malloc(fbytes + 40);
        memcpy(bad, frame, fbytes);
        const int what = (int)(rnd() % 4);
        if (what == 0) { for (int j = 0; j < 1 + (int)(rnd() % 4); ++j) bad[rnd() % fbytes] ^= (uint8_t)(1u << (rnd() % 8)); }
        else if (what == 1) fb = rnd() % (fbytes + 1);                                  /* truncated */
        else if (what == 2) { for (int j = 0; j < 32; ++j) bad[fbytes + j] = (uint8_t)rnd(); fb = fbytes + 1 + rnd() % 32; }
        else { const uint64_t at = rnd() % (fbytes < 64 ? fbytes : 64); for (uint64_t j = at; j < at + 8 && j < fbytes; ++j) bad[j] = (uint8_t)rnd(); }
        uint8_t *exact = (uint8_t *)malloc(fb ? fb : 1);                                /* exact-size copy: ASan sees any over-read */
        memcpy(exact, bad, fb);
        mi_frame_info i2;
        if (mi_frame_parse(exact, fb, &i2) == MI_OK) {
            if (i2.stream_bytes < (1ull << 28) && i2.nblocks < (1ull << 24)) {
                uint8_t *s3 = (uint8_t *)malloc(i2.stream_bytes + 8);
                uint64_t *b3 = (uint64_t *)malloc((i2.nblocks + 1) * 8);
                (void)mi_frame_unpack_blocks(exact, fb, s3, i2.stream_bytes + 8, b3, i2.nblocks + 1);
                free(s3); free(b3);
            }
        }
        /* too small destination buffers must be refused, not overrun */
        { uint8_t tiny[8]; uint64_t tb[2]; (void)mi_frame_unpack_blocks(exact, fb, tiny, 8, tb, 2); }
        free(exact); free(bad);
    }
    free(bits); free(stream); free(frame); free(s2); free(b2);
    return 0;
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 200;
    const uint32_t codecs[4] = {MI_FRAME_DEFLATE_T, MI_FRAME_DEFLATE_H, MI_FRAME_LZ77, MI_FRAME_FSE};
    for (int c = 0; c < cases; ++c) {
        const uint64_t nb = c < 4 ? (uint64_t)c : 1 + rnd() % 40;
        if (one_case(codecs[c % 4], nb, (c % 3) ? 65536u : 4096u)) return 1;
    }
    printf("frame harness: %d cases held\n", cases);
    return 0;
}
