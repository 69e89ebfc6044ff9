// huffman.hip — whole-buffer order-0 Huffman encode on gfx950.
//
// Replaces algorithms/huffman/huffman.c:288-328 (huffman_compress) and what it calls:
//   k_huff_hist       histogram                      huffman.c:184-187
//   k_huff_build      min-heap merge + path codes    huffman.c:100-163, 189-211, 217-250
//   k_huff_tile_bits  sum(len) per tile              (the serial bit cursor of write_bits, :18-48,
//   k_scan_u64        exclusive scan of tile bits     turned into a prefix sum)
//   k_huff_encode     MSB-first u32 packing          huffman.c:18-48, 267-285
//
// Data layout: the input stays where the caller put it in HBM and is read twice (histogram,
// encode) with 16 B/lane coalesced loads: 2n + c algorithmic bytes.  A tile is
// HUFF_TILE input bytes = one workgroup; per-tile histograms (1 KiB each, 3 % of n) turn the
// bit-offset computation into a dot product with the code lengths instead of a third pass.
#include "common.h"
#include "heap_cells.h"
#include <stddef.h>

#define HUFF_TILE      32768u          // input bytes per workgroup
#define HUFF_THREADS   256
#define HUFF_SUB       (HUFF_THREADS * 16)   // bytes staged per inner step (16 B per lane)

// ---------------------------------------------------------------------------------------------
// histogram: per-wave LDS sub-histograms, one tile per workgroup, tile histogram kept for the
// bit-offset pass, global histogram by atomics (u32, wrapping like the reference's counters).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HUFF_THREADS)
void k_huff_hist(const uint8_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ tile_hist,
                 uint32_t *__restrict__ hist, unsigned long long *__restrict__ hist64)
{
    __shared__ uint32_t sh[HUFF_THREADS / MI_WAVE][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < (HUFF_THREADS / MI_WAVE) * 256; i += HUFF_THREADS) (&sh[0][0])[i] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * HUFF_TILE;
    const uint64_t end = base + HUFF_TILE < n ? base + HUFF_TILE : n;
    uint32_t *my = sh[wave];
    for (uint64_t at = base + (uint64_t)tid * 16; at < end; at += HUFF_SUB) {
        if (at + 16 <= end) {
            const uint4 v = *reinterpret_cast<const uint4 *>(in + at);   // tiles are 16-B aligned
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&my[w[k] & 0xFF], 1u);
                atomicAdd(&my[(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&my[(w[k] >> 16) & 0xFF], 1u);
                atomicAdd(&my[w[k] >> 24], 1u);
            }
        } else {
            for (uint64_t q = at; q < end; ++q) atomicAdd(&my[in[q]], 1u);
        }
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < HUFF_THREADS / MI_WAVE; ++w) c += sh[w][tid];
    tile_hist[(uint64_t)blockIdx.x * 256 + tid] = c;
    if (c && hist) atomicAdd(&hist[tid], c);
    if (c && hist64) atomicAdd(&hist64[tid], (unsigned long long)c);      // shard histograms are summed across GPUs in 64 bits
}

// u64 histogram (the all-reduced one of a sharded job) -> the reference's u32 counters, wrapping like huffman.c:184-187
__global__ void k_huff_hist_narrow(const uint64_t *__restrict__ h64, uint32_t *__restrict__ h32)
{
    h32[threadIdx.x] = (uint32_t)h64[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// tree + codes.  One workgroup; lane 0 replays the reference's heap operation by operation —
// the codes depend on its tie-breaking (strict '<' in both sifts, leaves enqueued in symbol
// order, left = first dequeued), so the heap is emulated, not replaced.  255 merges on <= 256
// leaves: negligible next to the two data passes.
// ---------------------------------------------------------------------------------------------
struct HeapLds {
    uint32_t freq[511];
    int16_t  left[511], right[511], parent[511];
    uint8_t  value[511], is_right[511];
    uint64_t heap[256];                 // frequency << 16 | node id: a comparison is ONE LDS read per node (the loop runs on one lane)
    int16_t  leaf_of[256];
    uint32_t hist[256];                 // the counters, so that lane 0 does not fetch them from global memory one by one
    int      nnodes, root;
};
typedef HeapCells<uint64_t, 16> HuffCells;   // sifts that read ahead of their decisions (heap_cells.h); strict '<' on the
                                             // frequency alone, ties keep their places (huffman.c:111-119, 121-140)
#define HH_F(c) ((uint32_t)((c) >> 16))
#define HH_ID(c) ((int)((c) & 0xFFFFu))

__global__ __launch_bounds__(256)
void k_huff_build(const uint32_t *__restrict__ hist, mi_huffman_tree *__restrict__ tree,
                  mi_huffman_info *__restrict__ info, uint32_t *__restrict__ code_out,
                  uint8_t *__restrict__ len_out)
{
    __shared__ HeapLds h;
    __shared__ uint32_t s_maxlen;
    const int tid = threadIdx.x;
    const uint32_t f = hist[tid];
    if (tid == 0) s_maxlen = 0;
    h.leaf_of[tid] = -1;
    h.hist[tid] = f;
    __syncthreads();
    if (tid == 0) {
        int nheap = 0, nnodes = 0, root = -1;
        for (int s0 = 0; s0 < 256; s0 += 8) {
            uint32_t f8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f8[k] = h.hist[s0 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = s0 + k;
                const uint32_t fs = f8[k];
                if (!fs) continue;
                const int id = nnodes++;
                h.freq[id] = fs; h.left[id] = -1; h.right[id] = -1; h.parent[id] = -1;
                h.value[id] = (uint8_t)s; h.is_right[id] = 0; h.leaf_of[s] = (int16_t)id;
                HuffCells::push(h.heap, nheap, ((uint64_t)fs << 16) | (uint64_t)id);
            }
        }
        if (nheap > 0) {
            while (nheap > 1) {
                const uint64_t lc = HuffCells::pop(h.heap, nheap), rc = HuffCells::pop(h.heap, nheap);
                const int l = HH_ID(lc), r = HH_ID(rc);
                const int id = nnodes++;
                const uint32_t fm = HH_F(lc) + HH_F(rc);           // u32, wraps like the reference
                h.freq[id] = fm;
                h.left[id] = (int16_t)l; h.right[id] = (int16_t)r; h.parent[id] = -1;
                h.value[id] = 0; h.is_right[id] = 0;
                h.parent[l] = (int16_t)id; h.parent[r] = (int16_t)id; h.is_right[r] = 1;
                HuffCells::push(h.heap, nheap, ((uint64_t)fm << 16) | (uint64_t)id);
            }
            root = HH_ID(HuffCells::pop(h.heap, nheap));
        }
        h.nnodes = nnodes; h.root = root;
    }
    __syncthreads();
    // path codes: every present symbol walks leaf -> root; the edge next to the leaf is the LSB
    uint32_t code = 0, len = 0;
    if (f) {
        int node = h.leaf_of[tid];
        while (node != h.root) {
            if (len < 32) code |= (uint32_t)h.is_right[node] << len;
            ++len;
            node = h.parent[node];
        }
        atomicMax(&s_maxlen, len);
    }
    code_out[tid] = code;
    len_out[tid] = (uint8_t)(len > 255 ? 255 : len);
    tree->code[tid] = code;
    tree->length[tid] = (uint8_t)(len > 255 ? 255 : len);
    __syncthreads();
    for (int i = tid; i < 511; i += 256) {
        bool live = i < h.nnodes;
        tree->frequency[i] = live ? h.freq[i] : 0;
        tree->left[i] = live ? h.left[i] : (int16_t)-1;
        tree->right[i] = live ? h.right[i] : (int16_t)-1;
        tree->value[i] = live ? h.value[i] : 0;
    }
    if (tid == 0) {
        uint32_t nsym = 0;
        for (int s = 0; s < 256; ++s) nsym += h.leaf_of[s] >= 0;
        info->n_symbols = nsym;
        info->max_code_len = s_maxlen;
        info->n_nodes = (uint32_t)h.nnodes;
        info->status = nsym == 0 ? MI_ERR_EMPTY_INPUT : nsym == 1 ? MI_ERR_SINGLE_SYMBOL
                       : s_maxlen > 32 ? MI_ERR_CODE_TOO_LONG : MI_OK;
    }
}

// bits of one tile = dot(tile histogram, code lengths); one wave per tile
__global__ __launch_bounds__(256)
void k_huff_tile_bits(const uint32_t *__restrict__ tile_hist, const uint8_t *__restrict__ len,
                      uint64_t ntiles, uint64_t *__restrict__ tile_bits, uint32_t *__restrict__ uncovered)
{
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const uint4 c = *reinterpret_cast<const uint4 *>(tile_hist + tile * 256 + lane * 4);
    const uint32_t l4 = *reinterpret_cast<const uint32_t *>(len + lane * 4);
    // a byte that occurs but has no code: the tree was built for other data (only possible with a caller-supplied tree)
    if (uncovered && ((c.x && !(l4 & 0xFF)) || (c.y && !((l4 >> 8) & 0xFF)) || (c.z && !((l4 >> 16) & 0xFF)) || (c.w && !(l4 >> 24))))
        atomicOr(uncovered, 1u);
    uint64_t s = (uint64_t)c.x * (l4 & 0xFF) + (uint64_t)c.y * ((l4 >> 8) & 0xFF) +
                 (uint64_t)c.z * ((l4 >> 16) & 0xFF) + (uint64_t)c.w * (l4 >> 24);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) tile_bits[tile] = s;
}

// ---------------------------------------------------------------------------------------------
// single-workgroup exclusive scan, u64, in place capable; out[n] = total
// ---------------------------------------------------------------------------------------------
__device__ inline uint64_t block_exclusive_scan_u64(uint64_t v, uint64_t *total, uint64_t *s_tmp /* [nwaves+1] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int w = 0; w < nw; ++w) { uint64_t t = s_tmp[w]; s_tmp[w] = run; run += t; }
        s_tmp[nw] = run;
    }
    __syncthreads();
    uint64_t res = s_tmp[wave] + inc - v;
    *total = s_tmp[nw];
    __syncthreads();
    return res;
}

__global__ __launch_bounds__(1024)
void k_scan_u64(const uint64_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out, uint64_t base)
{
    __shared__ uint64_t s_tmp[17];
    const uint64_t per = (n + 1023) / 1024;
    const uint64_t a = (uint64_t)threadIdx.x * per, b = a + per < n ? a + per : n;
    uint64_t s = 0;
    for (uint64_t i = a; i < b; ++i) s += in[i];
    uint64_t total;
    uint64_t run = base + block_exclusive_scan_u64(s, &total, s_tmp);
    for (uint64_t i = a; i < b; ++i) { uint64_t v = in[i]; out[i] = run; run += v; }
    if (threadIdx.x == 0) out[n] = base + total;
}

// finish the info struct and zero the words that neighbouring tiles share (they are OR-merged)
__global__ __launch_bounds__(256)
void k_huff_finish_info(const uint64_t *__restrict__ tile_off, uint64_t ntiles, uint32_t *__restrict__ words,
                        uint64_t cap_words, mi_huffman_info *__restrict__ info, const uint32_t *__restrict__ uncovered)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = tile_off[ntiles];
    const bool fits = ((total + 31) >> 5) + 1 <= cap_words;
    if (i == 0) {
        info->total_bits = total;
        info->word_idx = total >> 5;
        info->bit_idx = total & 31;
        info->buffer_size = (total >> 5) * 4 + (total & 31) / 8 + (((total & 31) % 8) > 0);   // huffman.c:318-320
        if (info->status == MI_OK && uncovered && *uncovered) info->status = MI_ERR_ARG;
        if (info->status == MI_OK && !fits) info->status = MI_ERR_CAPACITY;
    }
    if (i <= ntiles && fits) words[tile_off[i] >> 5] = 0;
}

// ---------------------------------------------------------------------------------------------
// encode: 16 input bytes per lane per step, wave+block prefix sum of code lengths, bits ORed
// into an LDS staging window (ds_or_b32), window written out as whole dwords, coalesced.
// Stream bit j is bit (31 - j%32) of word j/32 (huffman.c:18-48).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HUFF_THREADS)
void k_huff_encode(const uint8_t *__restrict__ in, uint64_t n, const uint32_t *__restrict__ code,
                   const uint8_t *__restrict__ len, const uint64_t *__restrict__ tile_off,
                   uint32_t *__restrict__ words, const mi_huffman_info *__restrict__ info)
{
    __shared__ uint32_t s_code[256];
    __shared__ uint8_t  s_len[256];
    __shared__ uint32_t s_stage[HUFF_SUB + 2];          // worst case 32 bits per input byte
    __shared__ uint32_t s_wave[HUFF_THREADS / MI_WAVE + 1];
    if (info->status != MI_OK) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_code[tid] = code[tid];
    s_len[tid] = len[tid];
    const uint64_t base = (uint64_t)blockIdx.x * HUFF_TILE;
    const uint64_t end = base + HUFF_TILE < n ? base + HUFF_TILE : n;
    uint64_t bitpos = tile_off[blockIdx.x];
    const uint64_t first_word = bitpos >> 5;
    uint32_t carry = 0;                                   // partial word handed from step to step
    __syncthreads();
    for (uint64_t sub = base; sub < end; sub += HUFF_SUB) {
        const uint64_t at = sub + (uint64_t)tid * 16;
        uint8_t b[16];
        int nvalid = 0;
        if (at + 16 <= end) {
            const uint4 v = *reinterpret_cast<const uint4 *>(in + at);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = (uint8_t)(w[k >> 2] >> ((k & 3) * 8));
            nvalid = 16;
        } else if (at < end) {
            nvalid = (int)(end - at);
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = k < nvalid ? in[at + k] : 0;
        }
        uint32_t mybits = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < nvalid) mybits += s_len[b[k]];
        // block exclusive scan (u32: <= 4096 * 32 bits)
        uint32_t inc = mybits;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < HUFF_THREADS / MI_WAVE; ++w) { uint32_t t = s_wave[w]; if (w < wave) wbase += t; total += t; }
        const uint32_t lead = (uint32_t)(bitpos & 31);
        const uint32_t nwords_touched = (lead + total + 31) >> 5;
        for (uint32_t i = tid; i < nwords_touched + 1; i += HUFF_THREADS) s_stage[i] = (i == 0) ? carry : 0u;
        __syncthreads();
        uint32_t q = lead + wbase + inc - mybits;       // bit offset inside the staging window
        uint32_t widx = q >> 5, nacc = q & 31;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < nvalid) {
                const uint32_t c = s_code[b[k]], l = s_len[b[k]];
                acc |= (uint64_t)c << (64 - nacc - l);
                nacc += l;
                if (nacc >= 32) {
                    atomicOr(&s_stage[widx++], (uint32_t)(acc >> 32));
                    acc <<= 32; nacc -= 32;
                }
            }
        }
        if (nacc && acc) atomicOr(&s_stage[widx], (uint32_t)(acc >> 32));
        __syncthreads();
        const uint32_t ncomplete = (lead + total) >> 5;
        const uint64_t gw0 = bitpos >> 5;
        for (uint32_t i = tid; i < ncomplete; i += HUFF_THREADS) {
            const uint64_t gw = gw0 + i;
            if (gw == first_word) atomicOr(&words[gw], s_stage[i]);   // shared with the previous tile
            else words[gw] = s_stage[i];
        }
        carry = s_stage[ncomplete];                     // 0 when the step ended on a word boundary
        bitpos += total;
        __syncthreads();
    }
    if (tid == 0 && (bitpos & 31)) atomicOr(&words[bitpos >> 5], carry);   // shared with the next tile
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline uint64_t huff_ntiles(uint64_t n) { return (n + HUFF_TILE - 1) / HUFF_TILE; }

extern "C" mi_status mi_huffman_encode_dev(mi_ctx *ctx, const uint8_t *d_in, uint64_t n, uint32_t *d_words,
                                           uint64_t cap_words, mi_huffman_info *d_info, mi_huffman_tree *d_tree,
                                           uint64_t *d_tile_off, void *stream)
{
    if (!ctx || !d_words || !d_info || !d_tree || (n && !d_in) || cap_words < 2) return MI_ERR_ARG;
    if (((uintptr_t)d_in & 15) != 0) return MI_ERR_ARG;        // 16-B loads
    hipStream_t s = (hipStream_t)stream;   // NULL = HIP's default stream
    const uint64_t ntiles = huff_ntiles(n);
    size_t need = 256 * 4 + 256 * 4 + 256 + ntiles * 1024 + (ntiles + 1) * 8 * 2 + 4096;
    if (need > ctx->ws_bytes) { mi_status st = mi_ws_reserve(ctx, need); if (st) return st; }
    mi_carver cv(ctx->ws);
    uint32_t *hist = cv.take<uint32_t>(256);
    uint32_t *code = cv.take<uint32_t>(256);
    uint8_t  *len = cv.take<uint8_t>(256);
    uint32_t *tile_hist = cv.take<uint32_t>((ntiles ? ntiles : 1) * 256);
    uint64_t *tile_bits = cv.take<uint64_t>(ntiles + 1);
    uint64_t *tile_off = cv.take<uint64_t>(ntiles + 2);
    MI_HIP(ctx, hipMemsetAsync(hist, 0, 256 * 4, s));
    if (ntiles) {
        mi_prof_scope p(ctx, "k_huff_hist", s, n);
        hipLaunchKernelGGL(k_huff_hist, dim3((unsigned)ntiles), dim3(HUFF_THREADS), 0, s, d_in, n, tile_hist, hist, (unsigned long long *)nullptr);
    }
    {
        mi_prof_scope p(ctx, "k_huff_build", s, 1024);
        hipLaunchKernelGGL(k_huff_build, dim3(1), dim3(256), 0, s, hist, d_tree, d_info, code, len);
    }
    if (ntiles) {
        mi_prof_scope p(ctx, "k_huff_tile_bits", s, ntiles * 1024);
        hipLaunchKernelGGL(k_huff_tile_bits, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, s, tile_hist, len, ntiles, tile_bits, (uint32_t *)nullptr);
    }
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, tile_bits, ntiles, tile_off, (uint64_t)0);
    if (d_tile_off) MI_HIP(ctx, hipMemcpyAsync(d_tile_off, tile_off, (ntiles + 1) * 8, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_huff_finish_info, dim3((unsigned)((ntiles + 1 + 255) / 256)), dim3(256), 0, s,
                       tile_off, ntiles, d_words, cap_words, d_info, (const uint32_t *)nullptr);
    if (ntiles) {
        mi_prof_scope p(ctx, "k_huff_encode", s, n + (n * 5) / 8);
        hipLaunchKernelGGL(k_huff_encode, dim3((unsigned)ntiles), dim3(HUFF_THREADS), 0, s, d_in, n, code, len, tile_off,
                           d_words, d_info);
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------
// The same encoder in three steps, for ONE tree over a buffer that is spread over several GPUs (SURVEY.md 8e,
// "whole-buffer Huffman": huffman.c:179-215 builds one tree over the whole buffer, :267-328 packs with it):
//   mi_huffman_hist_dev              shard -> u64[256] histogram (+ per-tile histograms the caller keeps)
//   [the caller sums the histograms of all shards: one all-reduce of 2 KiB]
//   mi_huffman_build_dev             summed histogram -> the reference's tree and codes (identical on every GPU)
//   [bits of a shard = dot(its histogram, lengths); an all-gather of those gives every shard its global bit offset]
//   mi_huffman_encode_with_tree_dev  shard -> words, the stream starting `bit_offset` (0..31) bits into d_words[0]
// The shards' word ranges overlap by at most one word at each seam; OR-ing them there gives the single-GPU stream.
// ---------------------------------------------------------------------------------------------
extern "C" uint64_t mi_huffman_num_tiles(uint64_t n) { return huff_ntiles(n); }

extern "C" mi_status mi_huffman_hist_dev(mi_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist,
                                         uint32_t *d_tile_hist, void *stream)
{
    if (!ctx || !d_hist || (n && (!d_in || !d_tile_hist))) return MI_ERR_ARG;
    if (((uintptr_t)d_in & 15) != 0 || ((uintptr_t)d_tile_hist & 15) != 0) return MI_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    MI_HIP(ctx, hipMemsetAsync(d_hist, 0, 256 * 8, s));
    const uint64_t ntiles = huff_ntiles(n);
    if (ntiles) {
        mi_prof_scope p(ctx, "k_huff_hist", s, n);
        hipLaunchKernelGGL(k_huff_hist, dim3((unsigned)ntiles), dim3(HUFF_THREADS), 0, s, d_in, n, d_tile_hist,
                           (uint32_t *)nullptr, reinterpret_cast<unsigned long long *>(d_hist));
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

extern "C" mi_status mi_huffman_build_dev(mi_ctx *ctx, const uint64_t *d_hist, mi_huffman_info *d_info,
                                          mi_huffman_tree *d_tree, void *stream)
{
    if (!ctx || !d_hist || !d_info || !d_tree) return MI_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    size_t need = 256 * 4 + 256 * 4 + 256 + 4096;
    if (need > ctx->ws_bytes) { mi_status st = mi_ws_reserve(ctx, need); if (st) return st; }
    mi_carver cv(ctx->ws);
    uint32_t *hist = cv.take<uint32_t>(256);
    uint32_t *code = cv.take<uint32_t>(256);
    uint8_t  *len = cv.take<uint8_t>(256);
    hipLaunchKernelGGL(k_huff_hist_narrow, dim3(1), dim3(256), 0, s, d_hist, hist);
    {
        mi_prof_scope p(ctx, "k_huff_build", s, 1024);
        hipLaunchKernelGGL(k_huff_build, dim3(1), dim3(256), 0, s, hist, d_tree, d_info, code, len);
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

extern "C" mi_status mi_huffman_encode_with_tree_dev(mi_ctx *ctx, const uint8_t *d_in, uint64_t n, const mi_huffman_tree *d_tree,
                                                     const uint32_t *d_tile_hist, uint32_t bit_offset, uint32_t *d_words,
                                                     uint64_t cap_words, mi_huffman_info *d_info, uint64_t *d_tile_off, void *stream)
{
    if (!ctx || !d_words || !d_info || !d_tree || (n && (!d_in || !d_tile_hist)) || cap_words < 2 || bit_offset > 31) return MI_ERR_ARG;
    if (((uintptr_t)d_in & 15) != 0 || ((uintptr_t)d_tile_hist & 15) != 0) return MI_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t ntiles = huff_ntiles(n);
    size_t need = (ntiles + 1) * 8 * 2 + 4096;
    if (need > ctx->ws_bytes) { mi_status st = mi_ws_reserve(ctx, need); if (st) return st; }
    mi_carver cv(ctx->ws);
    uint64_t *tile_bits = cv.take<uint64_t>(ntiles + 1);
    uint64_t *tile_off = cv.take<uint64_t>(ntiles + 2);
    uint32_t *uncovered = cv.take<uint32_t>(4);
    const uint32_t *code = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(d_tree) + offsetof(mi_huffman_tree, code));
    const uint8_t *len = reinterpret_cast<const uint8_t *>(d_tree) + offsetof(mi_huffman_tree, length);
    // the caller's info is rewritten for THIS shard: status OK, sizes of bit_offset + shard bits
    MI_HIP(ctx, hipMemsetAsync(d_info, 0, sizeof(mi_huffman_info), s));
    MI_HIP(ctx, hipMemsetAsync(uncovered, 0, 4, s));
    if (ntiles) {
        mi_prof_scope p(ctx, "k_huff_tile_bits", s, ntiles * 1024);
        hipLaunchKernelGGL(k_huff_tile_bits, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, s, d_tile_hist, len, ntiles, tile_bits, uncovered);
    }
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, tile_bits, ntiles, tile_off, (uint64_t)bit_offset);
    if (d_tile_off) MI_HIP(ctx, hipMemcpyAsync(d_tile_off, tile_off, (ntiles + 1) * 8, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_huff_finish_info, dim3((unsigned)((ntiles + 1 + 255) / 256)), dim3(256), 0, s,
                       tile_off, ntiles, d_words, cap_words, d_info, uncovered);
    if (ntiles) {
        mi_prof_scope p(ctx, "k_huff_encode", s, n + (n * 5) / 8);
        hipLaunchKernelGGL(k_huff_encode, dim3((unsigned)ntiles), dim3(HUFF_THREADS), 0, s, d_in, n, code, len, tile_off,
                           d_words, d_info);
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

extern "C" mi_status mi_huffman_encode(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, uint32_t *h_words,
                                       uint64_t cap_words, mi_huffman_info *h_info, mi_huffman_tree *h_tree)
{
    if (!ctx || !h_words || !h_info || (n && !h_in)) return MI_ERR_ARG;
    hipStream_t s = mi_host_stream(ctx);
    uint8_t *d_in = nullptr; uint32_t *d_words = nullptr; mi_huffman_info *d_info = nullptr; mi_huffman_tree *d_tree = nullptr;
    mi_status st = MI_OK;
    // reserve the kernels' workspace first: growing it later would synchronise mid-sequence
    {
        const uint64_t ntiles = huff_ntiles(n);
        st = mi_ws_reserve(ctx, 256 * 4 + 256 * 4 + 256 + ntiles * 1024 + (ntiles + 1) * 8 * 2 + 4096);
        if (st) return st;
    }
    if (hipMalloc(&d_in, n + 16) != hipSuccess || hipMalloc(&d_words, cap_words * 4) != hipSuccess ||
        hipMalloc(&d_info, sizeof(mi_huffman_info)) != hipSuccess || hipMalloc(&d_tree, sizeof(mi_huffman_tree)) != hipSuccess) {
        st = MI_ERR_NOMEM;
    }
    if (st == MI_OK && n && hipMemcpyAsync(d_in, h_in, n, hipMemcpyHostToDevice, s) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK) st = mi_huffman_encode_dev(ctx, d_in, n, d_words, cap_words, d_info, d_tree, nullptr, s);
    if (st == MI_OK && hipMemcpyAsync(h_info, d_info, sizeof(*h_info), hipMemcpyDeviceToHost, s) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK && hipStreamSynchronize(s) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK && h_tree && hipMemcpy(h_tree, d_tree, sizeof(*h_tree), hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK && h_info->status != MI_OK) st = (mi_status)h_info->status;
    if (st == MI_OK) {
        uint64_t nw = (h_info->total_bits + 31) >> 5;
        if (nw > cap_words) st = MI_ERR_CAPACITY;
        else if (nw && hipMemcpy(h_words, d_words, nw * 4, hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    }
    (void)hipFree(d_in); (void)hipFree(d_words); (void)hipFree(d_info); (void)hipFree(d_tree);
    return st;
}


// ---------------------------------------------------------------------------------------------
// decode (replaces huffman_decompress, huffman.c:330-364; the 12-bit lookup table is what the
// reference's unfinished huffman_decompress_lookup_table, :366-401, was reaching for).
// One lane per encoder tile: the encoder's tile bit offsets are the only sync points a
// variable-length code offers.  Without them (d_tile_off == NULL) a single lane walks the stream.
// ---------------------------------------------------------------------------------------------
#define HUFF_LUT_BITS 12
__global__ __launch_bounds__(64)
void k_huff_decode(const uint32_t *__restrict__ words, uint64_t total_bits, const mi_huffman_tree *__restrict__ tree,
                   uint32_t n_nodes, const uint64_t *__restrict__ tile_off, uint64_t ntiles, uint32_t tile_bytes,
                   uint8_t *__restrict__ out, uint64_t n, uint32_t *__restrict__ err)
{
    __shared__ uint16_t s_lut[1 << HUFF_LUT_BITS];      // len << 8 | symbol; 0 = longer than the table: walk
    __shared__ int16_t  s_left[511], s_right[511];
    __shared__ uint8_t  s_val[511];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < (1u << HUFF_LUT_BITS); i += 64) s_lut[i] = 0;
    // the tree arrays come from the caller (a file, a peer): children must be node ids or -1, codes must fit their length
    bool bad = false;
    for (uint32_t i = lane; i < 511; i += 64) {
        const int l = tree->left[i], r = tree->right[i];
        if (l < -1 || l > 510 || r < -1 || r > 510 || ((l < 0) != (r < 0))) bad = true;
        s_left[i] = (int16_t)l; s_right[i] = (int16_t)r; s_val[i] = tree->value[i];
    }
    __syncthreads();
    for (uint32_t s = lane; s < 256; s += 64) {
        const uint32_t l = tree->length[s], c = tree->code[s];
        if (l && l <= HUFF_LUT_BITS && (c >> l) != 0) bad = true;
        else if (l && l <= HUFF_LUT_BITS) {
            const uint32_t lo = c << (HUFF_LUT_BITS - l), cnt = 1u << (HUFF_LUT_BITS - l);
            for (uint32_t k = 0; k < cnt; ++k) s_lut[lo + k] = (uint16_t)((l << 8) | s);
        }
    }
    __syncthreads();
    if (__ballot(bad) != 0ull) { if (lane == 0) atomicOr(err, 1u); return; }
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * 64 + lane;
    if (t >= ntiles) return;
    const uint64_t o0 = t * tile_bytes, o1 = (o0 + tile_bytes < n) ? o0 + tile_bytes : n;
    uint64_t bit = tile_off ? tile_off[t] : 0;
    const int root = (int)n_nodes - 1;
    const uint64_t nwords = (total_bits + 31) >> 5;              // words past the stream read as zero
    uint32_t pack = 0;
    if (bit > total_bits) { atomicOr(err, 1u); return; }
    for (uint64_t o = o0; o < o1; ++o) {
        // next 32 stream bits, MSB first: bit j of the stream is bit 31 - j%32 of word j/32
        const uint64_t wi = bit >> 5; const uint32_t sh = (uint32_t)(bit & 31u);
        const uint64_t two = ((uint64_t)(wi < nwords ? words[wi] : 0u) << 32) | (wi + 1 < nwords ? words[wi + 1] : 0u);
        const uint32_t peek = (uint32_t)((two << sh) >> 32);
        const uint32_t e = s_lut[peek >> (32 - HUFF_LUT_BITS)];
        uint32_t sym, len;
        if (e) { sym = e & 0xFFu; len = e >> 8; }
        else {
            int node = root; len = 0;
            while (s_left[node] >= 0 && len < 32) { node = ((peek >> (31 - len)) & 1u) ? s_right[node] : s_left[node]; ++len; }
            if (s_left[node] >= 0) { bad = true; break; }
            sym = s_val[node];
        }
        bit += len;
        if (bit > total_bits) { bad = true; break; }
        pack |= sym << (8 * (uint32_t)(o & 3u));                 // tiles start 4-byte aligned
        if ((o & 3u) == 3u) { *reinterpret_cast<uint32_t *>(out + o - 3) = pack; pack = 0; }
    }
    // a tile must end exactly where the table says the next one starts (the last entry is the end of the stream): a
    // table that does not belong to this stream or to this tile grid (a shard that did not start on a tile
    // boundary) otherwise decodes to garbage with MI_OK
    if (!bad && tile_off && bit != tile_off[t + 1]) bad = true;
    if (!bad) for (uint64_t q = o1 & ~3ull; q < o1; ++q) out[q] = (uint8_t)(pack >> (8 * (uint32_t)(q & 3u)));
    if (bad) atomicOr(err, 1u);
}

extern "C" mi_status mi_huffman_decode_dev(mi_ctx *ctx, const uint32_t *d_words, uint64_t total_bits,
                                           const mi_huffman_tree *d_tree, uint32_t n_nodes, const uint64_t *d_tile_off,
                                           uint8_t *d_out, uint64_t n, void *stream)
{
    if (!ctx || !d_words || !d_tree || (n && !d_out) || n_nodes < 3 || n_nodes > 511) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    if (((uintptr_t)d_out & 3u) != 0) return MI_ERR_ARG;
    // without tile offsets one lane walks the whole stream with a 32-bit tile length: refuse what it cannot cover
    // instead of decoding a prefix
    if (!d_tile_off && n > 0xFFFFFFFFull) return MI_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *err = mi_err_slot(ctx, s);
    if (!err) return MI_ERR_HIP;
    const uint64_t ntiles = d_tile_off ? huff_ntiles(n) : 1;
    const uint32_t tile_bytes = d_tile_off ? HUFF_TILE : 0xFFFFFFFFu;
    {
        mi_prof_scope p(ctx, "k_huff_decode", s, n);
        hipLaunchKernelGGL(k_huff_decode, dim3((unsigned)((ntiles + 63) / 64)), dim3(64), 0, s, d_words, total_bits, d_tree,
                           n_nodes, d_tile_off, ntiles, tile_bytes, d_out, n, err);
    }
    uint32_t h_err = 0;
    MI_HIP(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    return h_err ? MI_ERR_CORRUPT : MI_OK;
}
