/*
 * mi_fse.h — C entry points for the FSE path (libmi_fse.so).  The reference's
 * algorithms/fse/src/main.zig:50-68 `compress(input, output) usize` is Zig and does not
 * compile; this is the C signature a maintainer would bind instead.
 *
 * Stream = concatenation of block records (one per 65 536 input bytes), each 4-byte aligned:
 *   u8  present[32]            bitmap of symbols with a non-zero normalised count
 *   u16 count[nsym] (+pad)     normalised counts (main.zig:106-149), symbol order
 *   u16 final_state[S] (+pad)  per sub-stream: x - 2^table_log after the last (= first) symbol
 *   u32 nbits[S]               per sub-stream bit count
 *   u32 payload[...]           sub-stream i: ceil(nbits[i]/32) words, bits LSB first (main.zig:28-39)
 * Sub-stream i covers bytes [i*m, min((i+1)*m, n)) of the block, m = ceil(n/S) rounded up to 4;
 * it is encoded last symbol first (main.zig:58-62) from state 2^table_log.
 */
#ifndef MI_FSE_H
#define MI_FSE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#define MI_FSE_TABLE_LOG 8      /* main.zig:80 */
#define MI_FSE_STREAMS   64
#define MI_FSE_BLOCK     65536

size_t fse_compress_bound(size_t len);
/* returns compressed bytes written to output (capacity fse_compress_bound(len)), 0 on error.
 * Layout: u64 original length, u64 nblocks, u64 bit offsets[nblocks+1], then the records. */
size_t fse_compress(const uint8_t *input, size_t len, uint8_t *output);
/* returns the decoded length, 0 on error */
size_t fse_decompress(const uint8_t *input, size_t len, uint8_t *output, size_t capacity);
#ifdef __cplusplus
}
#endif
#endif
