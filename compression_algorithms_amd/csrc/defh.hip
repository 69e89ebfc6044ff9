// defh.hip — "mode H": the entropy stage the reference's deflate leaves as a TODO
// (algorithms/deflate/lz77.c:279 "// TODO: Build huffman tree and encode compressed buffer").
//
// Taken from the reference: the token sequence (algorithms/deflate/lz77.c:199-280, produced by the
// same finder / parse kernels as the byte-token stream), the 286-symbol alphabet
// (deflate/huffman.h:6 NUM_CODES; huffman.c:49-62: literal b -> b, match with offset d ->
// 256 + clz16(d)), the per-block tally (lz77.c:206,231,273), MSB-first u32 packing
// (deflate/huffman.c:16-46) and the heap merge procedure with its tie-breaking
// (algorithms/huffman/huffman.c:100-163,189-211).  Defined by this project (the bit
// stream is PARITY UNPINNED — the reference has no such encoder; the oracle restatement is
// oracle/orc_defh.c): lengths from that heap, canonical codes by (length, symbol), a match symbol is
// followed by the offset bits below its leading one and the 5-bit length.
//
// Block record, 4-byte aligned:   u32 n_tokens | u8 len[286] + 2 pad | u32 words[] (MSB-first)
//
//   k_defh_encode   one workgroup per block over the token records k_lz_parse_emit left (mode H)
//   k_defh_decode   one wave per block: record -> tokens -> bytes (LZ copy in LDS), no token stream in HBM
#include "lz_common.h"
#include "lz_decode.h"
#include "heap_cells.h"
#include <stdlib.h>

#define DEFH_NSYM     286
#define DEFH_HDR      292u          // bytes before the packed words
// 256 threads per block: the kernel spends its time behind ONE lane (the heap), so what counts is how many blocks a CU holds —
// wave slots, not LDS, are the limit (32 per CU: 8 blocks of 4 waves instead of 4 of 8).  Same box: 512 / 256 / 128 threads
// 17.30 / 17.72 / 17.68 GB/s for the whole mode-H step.
#ifndef DEFH_THREADS
#define DEFH_THREADS  256
#endif
#define DEFH_PER      4             // tokens per thread per round
#define DEFH_MAXBITS  44u           // code <= 24 (65536 tokens: Fibonacci bound) + 15 offset bits + 5 length bits
// direct LUT of the decoder: 9 bits = 1 KiB of LDS (more waves per CU beat fewer slow-path symbols: 11 / 10 / 9 / 8 / 7 bits
// decode 17.8 / 19.2 / 19.5 / 18.6 / 18.6 GB/s); decode-side only, the format does not depend on it
#ifndef DEFH_LUT_BITS
#define DEFH_LUT_BITS 9
#endif

// The reference's array heap (algorithms/huffman/huffman.c:100-163: strict '<' on the frequency in both sifts, leaves
// enqueued in symbol order, first pop = left).  A heap cell holds frequency << 10 | node id (frequencies are <= 65 536
// tokens, ids < 572), so a comparison is ONE LDS read per node instead of two dependent ones (heap[i], then freq[heap[i]]) and
// a merged node's frequency comes out of the two cells it pops: no frequency array.  The merge loop runs on one lane and is the
// critical path of the entropy stage; its sifts read ahead of their decisions (heap_cells.h).  Ties are still decided by
// position alone: only the frequency field is compared.
struct DefhHeap {
    int16_t  parent[2 * DEFH_NSYM];
    uint32_t heap[DEFH_NSYM + 2];        // frequency << 10 | node id
    int16_t  leaf_of[DEFH_NSYM];
    int      nnodes, root;
};
typedef HeapCells<uint32_t, 10> DefhCells;
#define DH_F(c) ((c) >> 10)
#define DH_ID(c) ((int)((c) & 1023u))

__device__ __forceinline__ uint32_t clz16(uint32_t d) { return (uint32_t)__builtin_clz(d & 0xFFFFu) - 16u; }   // d in 1..65535

// canonical codes from lengths: code[s] = first code of its length + rank among equal lengths (symbol order)
__device__ __forceinline__ void defh_canonical(const uint8_t *s_len, uint32_t *s_code, uint32_t *s_count /*[34]*/, uint32_t *s_next /*[34]*/)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < 34; i += nt) s_count[i] = 0;
    __syncthreads();
    for (int s = tid; s < DEFH_NSYM; s += nt) if (s_len[s]) atomicAdd(&s_count[s_len[s] > 33 ? 33 : s_len[s]], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0;
        s_next[0] = 0;
        for (int l = 1; l <= 32; ++l) { c = (c + s_count[l - 1]) << 1; s_next[l] = c; }
        s_next[33] = 0;
    }
    __syncthreads();
    for (int s = tid; s < DEFH_NSYM; s += nt) {
        const uint32_t l = s_len[s];
        uint32_t rank = 0;
        for (int k = 0; k < s; ++k) rank += (s_len[k] == l);
        s_code[s] = (l && l <= 32) ? s_next[l] + rank : 0u;
    }
    __syncthreads();
}

// Two kernels since round 3.  The code LENGTHS wait ~1.4 M cycles per block behind one lane (the reference heap): that kernel is
// ONE wave per block with 8 KiB of LDS — twenty blocks in flight per CU where the fused kernel (256 threads, the pack stage's
// 5.6 KiB window) held eight.  It leaves the lengths in the record's header and the canonical codes behind the tally in the
// block's slot; the pack kernel reads both.
#define DEFH_CODE_AT (LZ_DEFH_HIST_AT + 288u)          // slot words [.., +288): the canonical codes, from k_defh_lengths to k_defh_pack
static_assert(DEFH_CODE_AT + 288u <= LZ_SLOT_WORDS, "tally and codes live behind the record in the slot");

__global__ __launch_bounds__(64)
void k_defh_lengths(uint32_t *__restrict__ slots)
{
    // 4.5 KiB of LDS (8 before): the tally's array becomes the codes' once the leaves are enqueued, the heap cells carry the
    // frequencies — the wave holds its LDS for the whole serial merge, and LDS-seconds are what the pipeline's stages compete for
    __shared__ DefhHeap h;
    __shared__ uint32_t s_hist[DEFH_NSYM + 2];
    __shared__ __attribute__((aligned(16))) uint8_t s_len[DEFH_NSYM + 2];
    __shared__ uint32_t s_count[34], s_next[34];
    uint32_t *const s_code = s_hist;
    const int tid = threadIdx.x;
    uint32_t *out = slots + (size_t)blockIdx.x * LZ_SLOT_WORDS;
    // ---- tally (lz77.c:206,231,273): taken by k_lz_parse_emit while it wrote the token records, left at the end of the slot
    for (int i = tid; i < DEFH_NSYM + 2; i += 64) { s_hist[i] = out[LZ_DEFH_HIST_AT + i]; s_len[i] = 0; }
    for (int i = tid; i < DEFH_NSYM; i += 64) h.leaf_of[i] = -1;
    __syncthreads();
    // ---- code lengths: the reference heap, leaves enqueued in symbol order (one lane; <= 285 merges)
    if (tid == 0) {
        int nheap = 0, nnodes = 0, root = -1;
        for (int s0 = 0; s0 < DEFH_NSYM; s0 += 8) {
            uint32_t f8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f8[k] = s_hist[s0 + k < DEFH_NSYM ? s0 + k : 0];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = s0 + k;
                const uint32_t f = f8[k];
                if (s >= DEFH_NSYM || !f) continue;
                const int id = nnodes++;
                h.parent[id] = -1; h.leaf_of[s] = (int16_t)id;
                DefhCells::push(h.heap, nheap, (f << 10) | (uint32_t)id);
            }
        }
        if (nnodes > 1) {
            while (nheap > 1) {
                const uint32_t lc = DefhCells::pop(h.heap, nheap), rc = DefhCells::pop(h.heap, nheap);
                const int id = nnodes++;
                h.parent[id] = -1;
                h.parent[DH_ID(lc)] = (int16_t)id; h.parent[DH_ID(rc)] = (int16_t)id;
                DefhCells::push(h.heap, nheap, ((DH_F(lc) + DH_F(rc)) << 10) | (uint32_t)id);
            }
            root = DH_ID(DefhCells::pop(h.heap, nheap));
        }
        h.nnodes = nnodes; h.root = root;
    }
    __syncthreads();
    for (int sy = tid; sy < DEFH_NSYM; sy += 64) {
        const int leaf = h.leaf_of[sy];
        if (leaf < 0) continue;
        uint32_t len = 0;
        if (h.nnodes == 1) len = 1;
        else for (int node = leaf; node != h.root; node = h.parent[node]) ++len;
        s_len[sy] = (uint8_t)len;
    }
    __syncthreads();
    defh_canonical(s_len, s_code, s_count, s_next);
    for (int i = tid; i < (DEFH_NSYM + 2) / 4; i += 64) out[1 + i] = reinterpret_cast<const uint32_t *>(s_len)[i];     // the record's header
    for (int i = tid; i < DEFH_NSYM + 2; i += 64) out[DEFH_CODE_AT + i] = s_code[i];
}

__global__ __launch_bounds__(DEFH_THREADS)
void k_defh_encode(const uint32_t *__restrict__ trec_all, uint32_t *__restrict__ slots, uint64_t *__restrict__ block_bits)
{
    __shared__ uint32_t s_code[DEFH_NSYM + 2];
    __shared__ __attribute__((aligned(16))) uint8_t s_len[DEFH_NSYM + 2];
    __shared__ uint32_t s_scan[DEFH_THREADS / 64 + 2];
    __shared__ uint32_t s_stage[DEFH_THREADS * DEFH_PER * DEFH_MAXBITS / 32 + 8];

    const int tid = threadIdx.x;
    const uint32_t lb = blockIdx.x;
    const uint32_t ntok = (uint32_t)block_bits[lb];                   // k_lz_parse_emit (mode H) left the token count here
    const uint32_t *trec = trec_all + (size_t)lb * LZ_MAX_BLOCK;
    uint32_t *out = slots + (size_t)lb * LZ_SLOT_WORDS;
    auto symbol_of = [](uint32_t r) -> uint32_t { return (r >> 31) ? 256u + clz16(r) : (r & 0xFFu); };

    // ---- lengths and codes: k_defh_lengths left them in the header and behind the tally
    for (int i = tid; i < DEFH_NSYM + 2; i += DEFH_THREADS) s_code[i] = out[DEFH_CODE_AT + i];
    for (int i = tid; i < (DEFH_NSYM + 2) / 4; i += DEFH_THREADS) reinterpret_cast<uint32_t *>(s_len)[i] = out[1 + i];
    __syncthreads();

    // ---- header
    if (tid == 0) out[0] = ntok;
    uint32_t *words = out + DEFH_HDR / 4;

    // ---- pack, DEFH_THREADS * DEFH_PER tokens per round; a thread owns DEFH_PER consecutive tokens
    uint64_t qbase = 0;
    uint32_t carry = 0;
    // (the next round's records are in flight while this round is packed: a round is three barriers and one HBM round trip)
    uint4 nrv = make_uint4(0, 0, 0, 0);
    if ((uint32_t)tid * DEFH_PER < ntok) nrv = *reinterpret_cast<const uint4 *>(trec + (uint32_t)tid * DEFH_PER);
    for (uint32_t t0 = 0; t0 < ntok; t0 += DEFH_THREADS * DEFH_PER) {
        const uint32_t t = t0 + (uint32_t)tid * DEFH_PER;
        const uint4 rv = nrv;                                              // the token array is 65536 words: in bounds
        if (t + DEFH_THREADS * DEFH_PER < ntok) nrv = *reinterpret_cast<const uint4 *>(trec + t + DEFH_THREADS * DEFH_PER);
        const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w};
        uint32_t c_v[4], c_k[4], x_v[4], x_k[4], mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c_k[k] = 0; x_k[k] = 0; c_v[k] = 0; x_v[k] = 0;
            if (t + k < ntok) {
                const uint32_t sym = symbol_of(r[k]);
                c_v[k] = s_code[sym]; c_k[k] = s_len[sym];
                if (r[k] >> 31) {
                    const uint32_t d = r[k] & 0xFFFFu, nx = 15u - clz16(d);
                    x_v[k] = ((d - (1u << nx)) << 5) | ((r[k] >> 16) & 31u);
                    x_k[k] = nx + 5u;
                }
                mine += c_k[k] + x_k[k];
            }
        }
        uint32_t total;
        uint32_t rel = block_exclusive_scan<uint32_t>(mine, OpAddU32(), 0u, s_scan, &total);
        const uint64_t w0 = qbase >> 5;
        const uint32_t sh0 = (uint32_t)(qbase & 31u);
        const uint32_t nwords = (sh0 + total + 31u) >> 5;
        for (uint32_t i = tid; i < nwords + 1; i += DEFH_THREADS) s_stage[i] = (i == 0) ? carry : 0u;
        __syncthreads();
        rel += sh0;
        auto put = [&](uint32_t v, uint32_t k) {                        // MSB first: k <= 32 bits at stage bit `rel`
            if (!k) return;
            const uint32_t wi = rel >> 5, sh = rel & 31u;
            const uint64_t x = (uint64_t)v << (64u - k - sh);
            atomicOr(&s_stage[wi], (uint32_t)(x >> 32));
            if ((uint32_t)x) atomicOr(&s_stage[wi + 1], (uint32_t)x);
            rel += k;
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) { put(c_v[k], c_k[k]); put(x_v[k], x_k[k]); }
        __syncthreads();
        const uint32_t ncomplete = (sh0 + total) >> 5;
        for (uint32_t i = tid; i < ncomplete; i += DEFH_THREADS) words[w0 + i] = s_stage[i];
        carry = s_stage[ncomplete];
        qbase += total;
        __syncthreads();
    }
    if (tid == 0) {
        if (qbase & 31u) words[qbase >> 5] = carry;
        block_bits[lb] = ((uint64_t)DEFH_HDR + ((qbase + 31) >> 5) * 4) * 8;
    }
}

void defh_launch_encode(const uint32_t *trec, uint32_t *slots, uint64_t *block_bits, uint32_t nb, hipStream_t s)
{
    hipLaunchKernelGGL(k_defh_lengths, dim3(nb), dim3(64), 0, s, slots);
    hipLaunchKernelGGL(k_defh_encode, dim3(nb), dim3(DEFH_THREADS), 0, s, trec, slots, block_bits);
}

// ---------------------------------------------------------------------------------------------
// decode: one wave per block.  Every lane reads the same bits (uniform control flow); the LZ copy of a
// match is spread over the lanes, the output block is staged in LDS like k_lz_decode.
// ---------------------------------------------------------------------------------------------
template <uint32_t RING>
__global__ __launch_bounds__(64)
void k_defh_decode(const uint8_t *__restrict__ stream, uint64_t stream_bytes, const uint64_t *__restrict__ block_bits, LzP P,
                   uint8_t *__restrict__ out, uint64_t n_total, uint32_t *__restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[RING];
    __shared__ uint16_t s_lut[1 << DEFH_LUT_BITS];                     // symbol | length << 9; 0xFFFF = longer than the table
    // lengths and codes are only needed while the tables are built: they lie in the (still unused) ring
    static_assert(RING >= 4u * (DEFH_NSYM + 2) + ((DEFH_NSYM + 2 + 3) & ~3u), "the ring holds the build-time tables");
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_ring);
    uint8_t  *s_len = s_ring + 4u * (DEFH_NSYM + 2);
    __shared__ uint32_t s_count[34], s_next[34];
    __shared__ uint16_t s_sorted[DEFH_NSYM];                           // symbols by (length, symbol)
    __shared__ uint32_t s_first[34];                                   // index into s_sorted of the first symbol of a length
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const uint64_t off = b * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint64_t rb = block_bits[b], re = block_bits[b + 1];
    // the record must lie inside the stream: every later read is bounded by [rb, re)
    bool bad = (rb & 31u) || (re & 31u) || re < rb + DEFH_HDR * 8ull || re > stream_bytes * 8ull;
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    const uint32_t *rec = reinterpret_cast<const uint32_t *>(stream + (rb >> 3));
    const uint32_t ntok = rec[0];
    // n bytes of output need at most n tokens of at most 64 bits
    const uint64_t words64 = ((re - rb) >> 5) - DEFH_HDR / 4;
    const uint32_t nwords = words64 > 2ull * n + 2u ? 2u * n + 2u : (uint32_t)words64;
    const uint32_t *w = rec + DEFH_HDR / 4;
    for (uint32_t i = lane; i < (DEFH_NSYM + 2) / 4; i += 64) reinterpret_cast<uint32_t *>(s_len)[i] = rec[1 + i];
    __syncthreads();
    defh_canonical(s_len, s_code, s_count, s_next);
    // tables: first-code / sorted symbols for the general walk, a direct LUT for codes <= DEFH_LUT_BITS
    for (uint32_t i = lane; i < (1u << DEFH_LUT_BITS); i += 64) s_lut[i] = 0xFFFFu;
    if (lane == 0) { uint32_t run = 0; for (int l = 0; l < 34; ++l) { s_first[l] = run; run += (l >= 1 && l <= 32) ? s_count[l] : 0u; } }
    __syncthreads();
    for (uint32_t s = lane; s < DEFH_NSYM; s += 64) {
        const uint32_t l = s_len[s];
        if (l > 32) bad = true;
        if (l && l <= 32) {
            const uint32_t c = s_code[s];
            s_sorted[s_first[l] + (c - s_next[l])] = (uint16_t)s;
            if (l <= DEFH_LUT_BITS) {
                const uint32_t lo = c << (DEFH_LUT_BITS - l);
                if (lo + (1u << (DEFH_LUT_BITS - l)) > (1u << DEFH_LUT_BITS)) bad = true;       // over-subscribed lengths
                else for (uint32_t k = 0; k < (1u << (DEFH_LUT_BITS - l)); ++k) s_lut[lo + k] = (uint16_t)(s | (l << 9));
            }
        }
    }
    if (__ballot(bad) != 0ull) { if (lane == 0) atomicOr(err, 1u); return; }
    __syncthreads();

    // every control value below is wave-uniform: the bit buffer sits in SGPRs and is refilled from 64 record words held
    // one per lane (lz_decode.h); the only memory on a token's critical path is its LUT cell
    const uint32_t nbits = nwords * 32u;                               // nwords <= 2 n + 2
    BitsMsb br;
    br.init(w, nwords, lane);
    OutRing<RING> ring;
    ring.init(s_ring, out + off, lane);
    uint32_t pos = 0;
    uint32_t o = 0;
    for (uint32_t t = 0; t < ntok && o < n; ++t) {
        br.refill();
        const uint32_t v = br.top32();
        uint32_t sym, l;
        const uint32_t e = s_lut[v >> (32 - DEFH_LUT_BITS)];
        if (e != 0xFFFFu) { sym = e & 511u; l = e >> 9; }
        else {
            sym = DEFH_NSYM; l = DEFH_LUT_BITS + 1;
            for (; l <= 32; ++l) {
                const uint32_t c = v >> (32u - l), rel = c - s_next[l];
                if (c >= s_next[l] && rel < s_count[l]) { sym = s_sorted[s_first[l] + rel]; break; }
            }
            if (sym == DEFH_NSYM) { bad = true; break; }
        }
        br.skip(l); pos += l;
        if (sym < 256u) {
            ring.put_literal(o, sym);
            o += 1;
        } else {
            const uint32_t cz = sym - 256u;
            if (cz < 1u || cz > 15u) { bad = true; break; }
            br.refill();
            const uint32_t nx = 15u - cz, x = br.top32();
            const uint32_t d = (1u << nx) + (nx ? x >> (32u - nx) : 0u);
            const uint32_t len = (x << nx) >> 27;
            br.skip(nx + 5u); pos += nx + 5u;
            if (d > o) { bad = true; break; }
            const uint32_t take = (o + len <= n) ? len : n - o;
            ring.copy(o, d, take);
            o += take;
        }
        if (pos > nbits) { bad = true; break; }
        __builtin_amdgcn_wave_barrier();
        ring.advance(o);
    }
    if (o != n) bad = true;
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    ring.finish(n);
}

mi_status lz_check_params(const mi_lz_params *p);

extern "C" uint64_t mi_deflate_h_bound_bytes(uint64_t n, const mi_lz_params *p)
{
    // Per block of b bytes: a Huffman code cannot be longer than a fixed 9-bit code over 286 symbols, and a match
    // (code + <= 15 offset bits + 5 length bits <= 29 bits) replaces >= 4 literals (36 bits) — except the block's last
    // token, which may be a match that covers ONE real byte and runs into the zero tail: 9 b + 20 bits at most.
    // Plus the 292-byte header; records are whole words.  Every block pays the header, so the bound depends on p->block.
    const uint64_t block = (p && p->block) ? p->block : LZ_MAX_BLOCK;
    const uint64_t nblocks = (n + block - 1) / block;
    const uint64_t last = n - (nblocks ? (nblocks - 1) * block : 0);
    auto rec = [](uint64_t b) -> uint64_t { return DEFH_HDR + 4 * ((9 * b + 20 + 31) / 32); };
    return (nblocks ? (nblocks - 1) * rec(block) + rec(last) : 0) + 64;
}

// launch alone: errors accumulate in *err (an mi_err_slot the caller reads once everything it launched has run)
mi_status mi_deflate_h_decode_launch(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                                     const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, uint32_t *err, hipStream_t s)
{
    if (!ctx || !d_stream || !d_block_bits || (n && !d_out) || !err) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (!p->deflate || p->lbits > 5 || p->wbits > 16 || ((uintptr_t)d_stream & 3u)) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    const LzP P = lz_params_of(ctx, p);
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    {
        mi_prof_scope pr(ctx, "k_defh_decode", s, n);
        // a 4 KiB ring whatever the window (lz_decode.h): far matches read the output buffer
        const uint32_t W = 1u << P.wbits, need = W < P.block ? W : P.block;
        const char *e = getenv("MI_LZ_DECODE_RING");
        const uint32_t want = e ? (uint32_t)atoi(e) : (nblocks < 1024u ? need : 4096u);       // (few blocks: lz_decode.hip; with the 11-bit LUT 4 KiB: 17 GB/s, 8 KiB: 15, 16 KiB: 12.4, 2 KiB: 16.2; 9-bit LUT: 19.5 at 4 KiB, 19.7 at 8)
        if (want <= 4096u) hipLaunchKernelGGL(k_defh_decode<4096u>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
        else if (want <= 8192u) hipLaunchKernelGGL(k_defh_decode<8192u>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
        else if (need <= 16384u || want <= 16384u) hipLaunchKernelGGL(k_defh_decode<16384u>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
        else if (need <= 32768u || want <= 32768u) hipLaunchKernelGGL(k_defh_decode<32768u>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
        else hipLaunchKernelGGL(k_defh_decode<65536u>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
    }
    return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_HIP;
}

extern "C" mi_status mi_deflate_h_decode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                                             const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, void *stream)
{
    if (!ctx || !d_stream || !d_block_bits || (n && !d_out)) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (!p->deflate || p->lbits > 5 || p->wbits > 16 || ((uintptr_t)d_stream & 3u)) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *err = mi_err_slot(ctx, s);
    if (!err) return MI_ERR_HIP;
    st = mi_deflate_h_decode_launch(ctx, p, d_stream, stream_bytes, d_block_bits, d_out, n, err, s);
    if (st) return st;
    uint32_t h_err = 0;
    MI_HIP(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    return h_err ? MI_ERR_CORRUPT : MI_OK;
}
