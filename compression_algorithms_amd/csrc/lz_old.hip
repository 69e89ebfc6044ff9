// lz_old.hip — the reference's FIRST lz77 parser, the brute-force one it keeps beside the hash-table parser
// (lz77_compress_old, algorithms/lz77/lz77.h:51-54, algorithms/lz77/lz77.c:185-262; the call is commented out at
// algorithms/lz77/main.c:26).  A different stream from lz77_compress: at every token start the WHOLE window of
// 2^WINDOW_BITS - 1 bytes is searched for the longest match (first-longest wins: candidates are visited from the far end of
// the window, a later one must be strictly longer), the 4-byte words must be equal, the length stops at LENGTH_BITS' maximum
// and at the end of the buffer.  One stream over the whole buffer, same token format as lz77_compress (lz77.c:243-252).
//
// What the reference does per TOKEN (O(W) word compares, serial over the buffer) is parse independent: the best match of
// position p is a function of the bytes alone.  So:
//   k_lzold_match    best (length, offset) of EVERY position: a workgroup stages its tile's window in LDS, a thread walks the
//                    window of its position with a rolling word (one LDS byte per candidate), first-longest, early exit at
//                    the maximum — brute force like the reference, n x W compares spread over the chip
//   k_lzold_exit     greedy chain p -> p + max(1, length) resolved per 64-position chunk: where does a walk that enters the
//                    chunk at offset e (< 32: a match is at most 31 long) leave it — one thread per (chunk, e)
//   k_lzold_super / k_lzold_chain / k_lzold_entries   the same composed over 64 chunks, one serial walk over the
//                    super-chunks (n / 4096 steps), entries of every chunk
//   k_lzold_count, k_lzold_scan, k_lzold_emit   bits per chunk, their prefix, tokens OR-ed LSB first into the
//                    zeroed output (lz77.c:144-174: bit i of the stream is bit i % 8 of byte i / 8)
// Defined where the reference is not: bytes past the buffer read as zero (its word compare reads up to 3 bytes past `size`,
// lz77.c:216-220; the golden vectors were made with a zero tail behind the input).  Kept as written: for the first
// 2^WINDOW_BITS - 1 positions `buffer_index - window_size` wraps (uint64_t, lz77.c:208 with max() on unsigned operands), the
// window is empty and every token is a literal.
#include "common.h"
#include "lz_common.h"
#include <stdlib.h>

#define LZOLD_TILE   1024u       // positions per workgroup: they share one staged window (2^16: 66 KiB, two workgroups per CU)
#define LZOLD_ENT    32u          // entry offsets per chunk (max length 31: LENGTH_BITS <= 5)

__global__ __launch_bounds__(LZOLD_TILE)
void k_lzold_match(const uint8_t *__restrict__ in, uint64_t n, uint32_t wbits, uint32_t max_len,
                   uint8_t *__restrict__ L, uint16_t *__restrict__ O)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_win[];      // bytes [t0 - WS, t0 + TILE + 40)
    const uint32_t WS = (1u << wbits) - 1u;                               // window_size, lz77.c:193
    const uint64_t t0 = (uint64_t)blockIdx.x * LZOLD_TILE;
    const uint64_t lo = t0 >= WS ? t0 - WS : 0;                           // first byte staged
    const uint32_t span = (uint32_t)(t0 - lo) + LZOLD_TILE + 40u;
    for (uint32_t i = threadIdx.x; i < span; i += LZOLD_TILE) s_win[i] = (lo + i < n) ? in[lo + i] : (uint8_t)0;
    __syncthreads();
    const uint64_t p = t0 + threadIdx.x;
    if (p >= n) return;
    uint32_t best = 0, off = 0;
    if (p >= WS) {                                                        // (below: the unsigned wrap of lz77.c:208 — no window)
        const uint32_t sp = (uint32_t)(p - lo);                           // this position in the staged bytes
        const uint32_t lim = (n - p) < (uint64_t)max_len ? (uint32_t)(n - p) : max_len;      // lz77.c:223-233: stops at size and at MAX_LEN
        const uint32_t wp = (uint32_t)s_win[sp] | ((uint32_t)s_win[sp + 1] << 8) | ((uint32_t)s_win[sp + 2] << 16) | ((uint32_t)s_win[sp + 3] << 24);
        uint32_t q = sp - WS;                                             // window_start = buffer_index - window_size
        uint32_t w = (uint32_t)s_win[q] | ((uint32_t)s_win[q + 1] << 8) | ((uint32_t)s_win[q + 2] << 16) | ((uint32_t)s_win[q + 3] << 24);
        for (; q < sp; ++q) {
            if (w == wp) {                                                // lz77.c:215-221
                uint32_t len = lim < 4u ? lim : 4u;                       // the words are equal (bytes past `size` do not count)
                while (len < lim && s_win[q + len] == s_win[sp + len]) ++len;
                if (len > best) { best = len; off = sp - q; if (best == lim) break; }      // strictly longer: the first longest wins
            }
            w = (w >> 8) | ((uint32_t)s_win[q + 4] << 24);
        }
    }
    L[p] = (uint8_t)best;
    O[p] = (uint16_t)off;
}

// one thread per (chunk, entry offset): where the greedy walk leaves the chunk
__global__ __launch_bounds__(256)
void k_lzold_exit(const uint8_t *__restrict__ L, uint64_t n, uint64_t nchunks, uint8_t *__restrict__ ex)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t c = t / LZOLD_ENT;
    if (c >= nchunks) return;
    uint32_t o = (uint32_t)(t % LZOLD_ENT);
    const uint64_t base = c * 64u;
    while (o < 64u && base + o < n) { const uint32_t l = L[base + o]; o += l ? l : 1u; }
    ex[t] = (uint8_t)(o >= 64u ? o - 64u : 0u);
}

__global__ __launch_bounds__(256)
void k_lzold_super(const uint8_t *__restrict__ ex, uint64_t nchunks, uint64_t nsuper, uint8_t *__restrict__ sx)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t s = t / LZOLD_ENT;
    if (s >= nsuper) return;
    uint32_t x = (uint32_t)(t % LZOLD_ENT);
    const uint64_t c1 = (s + 1) * 64u < nchunks ? (s + 1) * 64u : nchunks;
    for (uint64_t c = s * 64u; c < c1; ++c) x = ex[c * LZOLD_ENT + x];
    sx[t] = (uint8_t)x;
}

// the one serial walk: n / 4096 dependent steps
__global__ void k_lzold_chain(const uint8_t *__restrict__ sx, uint64_t nsuper, uint8_t *__restrict__ sentry)
{
    if (threadIdx.x || blockIdx.x) return;
    uint32_t e = 0;
    for (uint64_t s = 0; s < nsuper; ++s) { sentry[s] = (uint8_t)e; e = sx[s * LZOLD_ENT + e]; }
}

__global__ __launch_bounds__(256)
void k_lzold_entries(const uint8_t *__restrict__ ex, const uint8_t *__restrict__ sentry, uint64_t nchunks, uint64_t nsuper, uint8_t *__restrict__ entry)
{
    const uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (s >= nsuper) return;
    uint32_t x = sentry[s];
    const uint64_t c1 = (s + 1) * 64u < nchunks ? (s + 1) * 64u : nchunks;
    for (uint64_t c = s * 64u; c < c1; ++c) { entry[c] = (uint8_t)x; x = ex[c * LZOLD_ENT + x]; }
}

__global__ __launch_bounds__(256)
void k_lzold_count(const uint8_t *__restrict__ L, const uint8_t *__restrict__ entry, uint64_t n, uint64_t nchunks,
                   uint32_t lit_bits, uint32_t mat_bits, uint64_t *__restrict__ chunk_bits)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t o = entry[c], bits = 0;
    const uint64_t base = c * 64u;
    while (o < 64u && base + o < n) { const uint32_t l = L[base + o]; bits += l ? mat_bits : lit_bits; o += l ? l : 1u; }
    chunk_bits[c] = bits;
}

__global__ __launch_bounds__(256)
void k_lzold_emit(const uint8_t *__restrict__ in, const uint8_t *__restrict__ L, const uint16_t *__restrict__ O,
                  const uint8_t *__restrict__ entry, const uint64_t *__restrict__ chunk_off, uint64_t n, uint64_t nchunks,
                  uint32_t wbits, uint32_t lbits, uint32_t *__restrict__ out, uint64_t *__restrict__ total_bits)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (c == 0 && total_bits) *total_bits = chunk_off[nchunks];
    if (c >= nchunks) return;
    uint32_t o = entry[c];
    uint64_t q = chunk_off[c];
    const uint64_t base = c * 64u;
    while (o < 64u && base + o < n) {
        const uint32_t l = L[base + o];
        uint32_t v, nb;
        if (l) { v = 1u | ((uint32_t)O[base + o] << 1) | (l << (1u + wbits)); nb = 1u + wbits + lbits; }   // lz77.c:243-246
        else   { v = (uint32_t)in[base + o] << 1; nb = 9u; }                                                // lz77.c:249-251
        const uint64_t wi = q >> 5; const uint32_t sh = (uint32_t)q & 31u;
        atomicOr(&out[wi], v << sh);
        if (sh + nb > 32u) atomicOr(&out[wi + 1], v >> (32u - sh));
        q += nb;
        o += l ? l : 1u;
    }
}

// exclusive prefix of the chunks' bit counts, out[nchunks] = the stream's length in bits (one workgroup)
struct OpAddU64 { __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a + b; } };
__global__ __launch_bounds__(1024)
void k_lzold_scan(const uint64_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out)
{
    __shared__ uint64_t s_tmp[18];
    const uint64_t per = (n + 1023) / 1024, a = (uint64_t)threadIdx.x * per, b = a + per < n ? a + per : n;
    uint64_t v = 0;
    for (uint64_t i = a; i < b; ++i) v += in[i];
    uint64_t tot;
    uint64_t run = block_exclusive_scan<uint64_t>(v, OpAddU64(), 0ull, s_tmp, &tot);
    for (uint64_t i = a; i < b; ++i) { out[i] = run; run += in[i]; }
    if (threadIdx.x == 0) out[n] = tot;
}

extern "C" uint64_t mi_lz77_old_bound_bytes(uint64_t n) { return (9 * n + 7) / 8 + 16; }     // all literals, plus the word the last token may touch

// d_out: at least mi_lz77_old_bound_bytes(n) bytes, 4-byte aligned; zeroed here.  *d_total_bits = the reference's bit_index.
extern "C" mi_status mi_lz77_old_encode_dev(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *d_in, uint64_t n,
                                            uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_total_bits, void *stream)
{
    if (!ctx || !d_out || !d_total_bits || (n && !d_in) || ((uintptr_t)d_out & 3u)) return MI_ERR_ARG;
    if (wbits < 8 || wbits > 16 || lbits < 3 || lbits > 5) return MI_ERR_ARG;
    if (cap_bytes < mi_lz77_old_bound_bytes(n)) return MI_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    MI_HIP(ctx, hipMemsetAsync(d_out, 0, mi_lz77_old_bound_bytes(n), s));
    if (n == 0) { MI_HIP(ctx, hipMemsetAsync(d_total_bits, 0, 8, s)); return MI_OK; }
    const uint64_t nchunks = (n + 63) / 64, nsuper = (nchunks + 63) / 64;
    // workspace: L n | O 2n | ex 32 per chunk | sx 32 per super | sentry | entry | chunk_bits | chunk_off
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = mi_align_up(off + bytes, 256); return o; };
    const size_t oL = take(n), oO = take(2 * n), oex = take(nchunks * LZOLD_ENT), osx = take(nsuper * LZOLD_ENT), ose = take(nsuper),
                 oen = take(nchunks), ocb = take(8 * nchunks), oco = take(8 * (nchunks + 1));
    mi_status st = mi_ws_reserve(ctx, off);
    if (st) return st;
    uint8_t *ws = (uint8_t *)ctx->ws;
    uint8_t *L = ws + oL; uint16_t *O = (uint16_t *)(ws + oO); uint8_t *ex = ws + oex, *sx = ws + osx, *sentry = ws + ose, *entry = ws + oen;
    uint64_t *cb = (uint64_t *)(ws + ocb), *co = (uint64_t *)(ws + oco);
    const uint32_t max_len = (1u << lbits) - 1u, WS = (1u << wbits) - 1u;
    const size_t lds = (size_t)WS + LZOLD_TILE + 48;
    {
        mi_prof_scope p(ctx, "k_lzold_match", s, n);
        hipLaunchKernelGGL(k_lzold_match, dim3((unsigned)((n + LZOLD_TILE - 1) / LZOLD_TILE)), dim3(LZOLD_TILE), lds, s, d_in, n, wbits, max_len, L, O);
    }
    auto grid = [](uint64_t threads) { return dim3((unsigned)((threads + 255) / 256)); };
    hipLaunchKernelGGL(k_lzold_exit, grid(nchunks * LZOLD_ENT), dim3(256), 0, s, L, n, nchunks, ex);
    hipLaunchKernelGGL(k_lzold_super, grid(nsuper * LZOLD_ENT), dim3(256), 0, s, ex, nchunks, nsuper, sx);
    hipLaunchKernelGGL(k_lzold_chain, dim3(1), dim3(64), 0, s, sx, nsuper, sentry);
    hipLaunchKernelGGL(k_lzold_entries, grid(nsuper), dim3(256), 0, s, ex, sentry, nchunks, nsuper, entry);
    hipLaunchKernelGGL(k_lzold_count, grid(nchunks), dim3(256), 0, s, L, entry, n, nchunks, 9u, 1u + wbits + lbits, cb);
    hipLaunchKernelGGL(k_lzold_scan, dim3(1), dim3(1024), 0, s, cb, nchunks, co);
    hipLaunchKernelGGL(k_lzold_emit, grid(nchunks), dim3(256), 0, s, d_in, L, O, entry, co, n, nchunks, wbits, lbits,
                       reinterpret_cast<uint32_t *>(d_out), d_total_bits);
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// host buffers (what the drop-in's lz77_compress_old calls): h_out holds mi_lz77_old_bound_bytes(n) bytes
extern "C" mi_status mi_lz77_old_encode(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *h_in, uint64_t n,
                                        uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_total_bits)
{
    if (!ctx || !h_out || !h_total_bits || (n && !h_in)) return MI_ERR_ARG;
    if (cap_bytes < mi_lz77_old_bound_bytes(n)) return MI_ERR_CAPACITY;
    const uint64_t cap = mi_lz77_old_bound_bytes(n);
    uint8_t *d_in = nullptr, *d_out = nullptr; uint64_t *d_bits = nullptr;
    mi_status st = MI_OK;
    if (hipMalloc(&d_in, n ? n : 1) != hipSuccess || hipMalloc(&d_out, cap) != hipSuccess || hipMalloc(&d_bits, 8) != hipSuccess) st = MI_ERR_NOMEM;
    if (!st && n && hipMemcpy(d_in, h_in, n, hipMemcpyHostToDevice) != hipSuccess) st = MI_ERR_HIP;
    if (!st) st = mi_lz77_old_encode_dev(ctx, wbits, lbits, d_in, n, d_out, cap, d_bits, nullptr);
    if (!st && hipDeviceSynchronize() != hipSuccess) st = MI_ERR_HIP;
    if (!st && hipMemcpy(h_total_bits, d_bits, 8, hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    if (!st && hipMemcpy(h_out, d_out, (size_t)(*h_total_bits / 8 + 1), hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_bits);
    return st;
}

// Decoder of a WHOLE-BUFFER lz77 stream (one stream, no block table: what lz77_compress_old writes, and lz77_compress for a
// buffer of one block): one wave, the block decoder of lz_decode.hip with the buffer as its only block (lz77.c:347-377).
void lz_launch_decode(const uint8_t *d_stream, uint64_t stream_bytes, const uint64_t *d_block_bits, const LzP &P, uint8_t *d_out,
                      uint64_t n, uint64_t nblocks, uint32_t *err, hipStream_t s);      // lz_decode.hip

extern "C" mi_status mi_lz77_whole_decode_dev(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *d_stream, uint64_t stream_bytes,
                                              uint64_t total_bits, uint8_t *d_out, uint64_t n, void *stream)
{
    if (!ctx || !d_stream || (n && !d_out) || n >= (1ull << 32) || wbits < 8 || wbits > 16 || lbits < 3 || lbits > 5) return MI_ERR_ARG;
    if (total_bits > stream_bytes * 8) return MI_ERR_CORRUPT;
    if (n == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    mi_status st = mi_ws_reserve(ctx, 64);
    if (st) return st;
    mi_lz_params p = mi_lz_params_lz77(wbits);
    p.lbits = lbits;
    LzP P = lz_params_of(ctx, &p);
    P.block = (uint32_t)n;                                       // the buffer is the block
    uint64_t *d_bits = reinterpret_cast<uint64_t *>(ctx->ws);
    const uint64_t h_bits[2] = {0, total_bits};
    MI_HIP(ctx, hipMemcpyAsync(d_bits, h_bits, 16, hipMemcpyHostToDevice, s));
    uint32_t *err = mi_err_slot(ctx, s);
    if (!err) return MI_ERR_HIP;
    lz_launch_decode(d_stream, stream_bytes, d_bits, P, d_out, n, 1, err, s);
    uint32_t h_err = 0;
    MI_HIP(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    return h_err ? MI_ERR_CORRUPT : MI_OK;
}

// host buffers (the drop-in's lz77_decompress for a stream lz77_compress_old wrote)
extern "C" mi_status mi_lz77_whole_decode(mi_ctx *ctx, uint32_t wbits, uint32_t lbits, const uint8_t *h_stream, uint64_t stream_bytes,
                                          uint64_t total_bits, uint8_t *h_out, uint64_t n)
{
    if (!ctx || !h_stream || (n && !h_out)) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    uint8_t *d_s = nullptr, *d_o = nullptr;
    const uint64_t sb = (stream_bytes + 3) & ~3ull;
    mi_status st = MI_OK;
    if (hipMalloc(&d_s, sb + 8) != hipSuccess || hipMalloc(&d_o, n) != hipSuccess) st = MI_ERR_NOMEM;
    if (!st && (hipMemset(d_s, 0, sb + 8) != hipSuccess || hipMemcpy(d_s, h_stream, stream_bytes, hipMemcpyHostToDevice) != hipSuccess)) st = MI_ERR_HIP;
    if (!st) st = mi_lz77_whole_decode_dev(ctx, wbits, lbits, d_s, sb, total_bits, d_o, n, nullptr);
    if (!st && hipMemcpy(h_out, d_o, n, hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    (void)hipFree(d_s); (void)hipFree(d_o);
    return st;
}
