// host_api.hip — host-buffer convenience entry points: copy in, run the *_dev path on the
// context's private stream, copy out, synchronise.  The PCIe-inclusive path of the drop-in
// libraries (dropin_*.c); throughput numbers are quoted on the *_dev entry points.
#include "common.h"
#include <stdlib.h>

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess; }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};
}


// ---------------------------------------------------------------------------------------------------------------------
// Host buffers larger than one chunk, byte-aligned formats (deflate tokens, mode H): the input goes up and the stream comes
// down in chunks while the GPU encodes the chunk between them.  One-shot (copy everything, encode, copy everything) was
// 93-98 ms per 10^9 bytes of which only 61 are the encoder (scripts/time_host_api.py).  Chunks are whole blocks, their
// streams are whole bytes, so they are laid end to end on the host and the per-chunk block tables shifted by the running
// total.  Two device buffers of each kind, two short-lived copy streams (created per call: an idle extra stream costs the
// *_dev pipeline 4-7 %, ctx.hip), events only — the host blocks once per chunk to learn the chunk's size.
// MI_HOST_CHUNK_BLOCKS sets the chunk (default 1024 blocks = one batch of the encoder: 14.7 GB/s against 11.6 at 2048 and 11.4 at 4000; tests use small ones).
// ---------------------------------------------------------------------------------------------------------------------
static uint64_t host_chunk_blocks()
{
    const char *e = getenv("MI_HOST_CHUNK_BLOCKS");
    long v = e ? atol(e) : 1024;
    if (v < 1) v = 1;
    if (v > 4000) v = 4000;                               // one chunk's block table must fit half of the pinned staging area
    return (uint64_t)v;
}

mi_status mi_encode_host_pipelined(mi_ctx *ctx, const mi_lz_params *p, int mode_h, const uint8_t *h_in, uint64_t n,
                                   uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits, bool *done)
{
    *done = false;
    const uint64_t cb = host_chunk_blocks(), C = cb * (uint64_t)p->block;
    if (!p->deflate || n <= C || !ctx->h_pinned || ctx->h_pinned_bytes < 2 * (cb + 1) * 8) return MI_OK;      // the caller's one-shot path
    const uint64_t nchunks = (n + C - 1) / C, nblocks = (n + p->block - 1) / p->block;
    const uint64_t cbound = (mode_h ? mi_deflate_h_bound_bytes(C, p) : mi_lz_bound_bytes(C, p)) + 64;
    hipStream_t s = mi_host_stream(ctx), cin = nullptr, cout = nullptr;
    hipEvent_t ev_in[2] = {}, ev_enc[2] = {}, ev_out[2] = {};
    DevBuf in[2], out[2], bits[2];
    mi_status st = MI_OK;
    bool ok = hipStreamCreateWithFlags(&cin, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&cout, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b)
        ok = in[b].alloc(C + 64) && out[b].alloc(cbound) && bits[b].alloc((cb + 1) * 8) &&
             hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ev_enc[b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ev_out[b], hipEventDisableTiming) == hipSuccess;
    auto chunk_len = [&](uint64_t c) -> uint64_t { return (c + 1 < nchunks) ? C : n - c * C; };
    auto issue_h2d = [&](uint64_t c) -> bool {
        const int b = (int)(c & 1);
        if (c >= 2 && hipStreamWaitEvent(cin, ev_enc[b], 0) != hipSuccess) return false;          // encode(c - 2) has read in[b]
        if (hipMemcpyAsync(in[b].p, h_in + c * C, chunk_len(c), hipMemcpyHostToDevice, cin) != hipSuccess) return false;
        return hipEventRecord(ev_in[b], cin) == hipSuccess;
    };
    uint64_t *pinned = reinterpret_cast<uint64_t *>(ctx->h_pinned);
    auto issue_encode = [&](uint64_t c) -> mi_status {
        const int b = (int)(c & 1);
        if (hipStreamWaitEvent(s, ev_in[b], 0) != hipSuccess) return MI_ERR_HIP;
        if (c >= 2 && hipStreamWaitEvent(s, ev_out[b], 0) != hipSuccess) return MI_ERR_HIP;       // the stream of chunk c - 2 has left out[b]
        const mi_status e = mode_h ? mi_deflate_h_encode_dev(ctx, p, in[b].as<uint8_t>(), chunk_len(c), out[b].as<uint8_t>(), cbound, bits[b].as<uint64_t>(), s)
                                   : mi_lz_encode_dev(ctx, p, in[b].as<uint8_t>(), chunk_len(c), out[b].as<uint8_t>(), cbound, bits[b].as<uint64_t>(), s);
        if (e) return e;
        const uint64_t nb = (chunk_len(c) + p->block - 1) / p->block;
        if (hipMemcpyAsync(pinned + (size_t)b * (cb + 1), bits[b].p, (nb + 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return MI_ERR_HIP;
        return hipEventRecord(ev_enc[b], s) == hipSuccess ? MI_OK : MI_ERR_HIP;
    };
    uint64_t base_bits = 0;
    if (!ok) st = MI_ERR_NOMEM;
    if (st == MI_OK && !issue_h2d(0)) st = MI_ERR_HIP;
    if (st == MI_OK) st = issue_encode(0);
    for (uint64_t c = 0; c < nchunks && st == MI_OK; ++c) {
        const int b = (int)(c & 1);
        if (c + 1 < nchunks) {
            if (!issue_h2d(c + 1)) { st = MI_ERR_HIP; break; }                                     // (may block the host: the GPU is encoding chunk c)
            st = issue_encode(c + 1);                                                              // queued behind chunk c: no idle GPU while the host reads c's sizes
            if (st) break;
        }
        if (hipEventSynchronize(ev_enc[b]) != hipSuccess) { st = MI_ERR_HIP; break; }
        const uint64_t nb = (chunk_len(c) + p->block - 1) / p->block, blk0 = c * cb;
        const uint64_t *cbits = pinned + (size_t)b * (cb + 1);
        for (uint64_t k = 0; k <= nb; ++k) h_block_bits[blk0 + k] = base_bits + cbits[k];
        const uint64_t cbytes = cbits[nb] / 8;                                                     // whole bytes: byte tokens / word-aligned records
        if (base_bits / 8 + cbytes > cap_bytes) { st = MI_ERR_CAPACITY; break; }
        if (cbytes && hipMemcpyAsync(h_out + base_bits / 8, out[b].p, cbytes, hipMemcpyDeviceToHost, cout) != hipSuccess) { st = MI_ERR_HIP; break; }
        if (hipEventRecord(ev_out[b], cout) != hipSuccess) { st = MI_ERR_HIP; break; }
        base_bits += cbits[nb];
    }
    (void)nblocks;
    // nothing may outlive the buffers: drain all three streams whatever happened
    if (cin) (void)hipStreamSynchronize(cin);
    (void)hipStreamSynchronize(s);
    if (cout) (void)hipStreamSynchronize(cout);
    for (int b = 0; b < 2; ++b) { if (ev_in[b]) (void)hipEventDestroy(ev_in[b]); if (ev_enc[b]) (void)hipEventDestroy(ev_enc[b]); if (ev_out[b]) (void)hipEventDestroy(ev_out[b]); }
    if (cin) (void)hipStreamDestroy(cin);
    if (cout) (void)hipStreamDestroy(cout);
    if (st == MI_OK) *done = true;
    return st;
}

// ---------------------------------------------------------------------------------------------------------------------
// The way back, every block format: the stream goes up and the bytes come down in chunks of whole blocks while the GPU decodes
// the chunks between them.  The device buffers are whole (stream, table, output) — a chunk's kernel gets the table from its own
// first block on and the output from its own first byte on; the table's bit offsets stay absolute — so there is nothing to
// recycle.  A decoder wave is one block and the chip wants ~8 000 of them in flight, so chunks are launched AHEAD chunks before
// their bytes are fetched, on AHEAD streams (kernels of neighbouring chunks overlap: one stream alone ran them one after the
// other and mode H gained nothing).  Host order per chunk c: upload(c + AHEAD), launch(c + AHEAD), download(c) — pageable
// copies block the calling thread, and the GPU has AHEAD chunks to decode while they do.
// MI_HOST_DECODE_CHUNK_BLOCKS sets the chunk, MI_HOST_DECODE_AHEAD the look-ahead (1..4); tests use small chunks.  10^9 bytes of
// output, ms one shot -> chunks of 4096 ahead 2: lz77 tokens 65 -> 52, deflate tokens 64 -> 41 (2 GB over the link: it is the
// copies now), mode H 85 -> 78 (2048 x 3: 57 / 44 / 80; 1024 x 4: 71 / 56 / 100; scripts/ab_hostdec.sh).
// ---------------------------------------------------------------------------------------------------------------------
static uint64_t host_decode_chunk_blocks()
{
    const char *e = getenv("MI_HOST_DECODE_CHUNK_BLOCKS");
    long v = e ? atol(e) : 4096;
    if (v < 1) v = 1;
    return (uint64_t)v;
}
static uint32_t host_decode_ahead()
{
    const char *e = getenv("MI_HOST_DECODE_AHEAD");
    long v = e ? atol(e) : 2;
    return (uint32_t)(v < 1 ? 1 : v > 4 ? 4 : v);
}

static mi_status mi_decode_host_pipelined(mi_ctx *ctx, const mi_lz_params *p, int mode_h, const uint8_t *h_stream, uint64_t stream_bytes,
                                          const uint64_t *h_block_bits, uint8_t *h_out, uint64_t n, bool *done)
{
    *done = false;
    constexpr uint32_t NE = 8;                                                                    // events in rotation (> AHEAD)
    const uint64_t cb = host_decode_chunk_blocks(), C = cb * (uint64_t)p->block;
    const uint32_t AHEAD = host_decode_ahead();
    const uint64_t nblocks = (n + p->block - 1) / p->block;
    if (nblocks <= cb) return MI_OK;                                                              // the caller's one-shot path
    const uint64_t nchunks = (nblocks + cb - 1) / cb;
    hipStream_t s = mi_host_stream(ctx), cin = nullptr, cout = nullptr, sx[4] = {s, nullptr, nullptr, nullptr};
    hipEvent_t ev_in[NE] = {}, ev_dec[NE] = {}, ev_setup = nullptr;
    DevBuf st_, bits, out;
    if (!st_.alloc(stream_bytes + 64) || !bits.alloc((nblocks + 1) * 8) || !out.alloc(n + 16)) return MI_ERR_NOMEM;
    bool ok = hipStreamCreateWithFlags(&cin, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&cout, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&ev_setup, hipEventDisableTiming) == hipSuccess;
    for (uint32_t k = 1; k < AHEAD && ok; ++k) ok = hipStreamCreateWithFlags(&sx[k], hipStreamNonBlocking) == hipSuccess;
    for (uint32_t b = 0; b < NE && ok; ++b)
        ok = hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_dec[b], hipEventDisableTiming) == hipSuccess;
    mi_status st = ok ? MI_OK : MI_ERR_HIP;
    uint32_t *err = nullptr;
    if (st == MI_OK) {
        ok = hipMemsetAsync(st_.as<uint8_t>() + stream_bytes, 0, 64, s) == hipSuccess &&
             hipMemcpyAsync(bits.p, h_block_bits, (nblocks + 1) * 8, hipMemcpyHostToDevice, s) == hipSuccess;
        err = ok ? mi_err_slot(ctx, s) : nullptr;
        if (!err || hipEventRecord(ev_setup, s) != hipSuccess) st = MI_ERR_HIP;
        for (uint32_t k = 1; k < AHEAD && st == MI_OK; ++k)                                       // table and status word first
            if (hipStreamWaitEvent(sx[k], ev_setup, 0) != hipSuccess) st = MI_ERR_HIP;
    }
    // stream bytes [lo, hi) of chunk c: from where the chunk before stopped to the dword that holds the chunk's last bit
    // (the readers fetch aligned dwords, lz_decode.h)
    auto stream_end = [&](uint64_t c) -> uint64_t {
        if (c + 1 >= nchunks) return stream_bytes;
        const uint64_t e = (((h_block_bits[(c + 1) * cb] + 7) >> 3) + 3) & ~3ull;
        return e < stream_bytes ? e : stream_bytes;
    };
    auto chunk_len = [&](uint64_t c) -> uint64_t { return (c + 1 < nchunks) ? C : n - c * C; };
    auto upload_launch = [&](uint64_t c) -> mi_status {
        const uint64_t lo = c ? stream_end(c - 1) : 0, hi = stream_end(c);
        hipStream_t sc = sx[c % AHEAD];
        if (hi > lo && hipMemcpyAsync(st_.as<uint8_t>() + lo, h_stream + lo, hi - lo, hipMemcpyHostToDevice, cin) != hipSuccess) return MI_ERR_HIP;
        if (hipEventRecord(ev_in[c % NE], cin) != hipSuccess || hipStreamWaitEvent(sc, ev_in[c % NE], 0) != hipSuccess) return MI_ERR_HIP;
        const mi_status e = mode_h ? mi_deflate_h_decode_launch(ctx, p, st_.as<uint8_t>(), stream_bytes, bits.as<uint64_t>() + c * cb, out.as<uint8_t>() + c * C, chunk_len(c), err, sc)
                                   : mi_lz_decode_launch(ctx, p, st_.as<uint8_t>(), stream_bytes, bits.as<uint64_t>() + c * cb, out.as<uint8_t>() + c * C, chunk_len(c), err, sc);
        if (e) return e;
        return hipEventRecord(ev_dec[c % NE], sc) == hipSuccess ? MI_OK : MI_ERR_HIP;
    };
    for (uint64_t c = 0; c < AHEAD && c < nchunks && st == MI_OK; ++c) st = upload_launch(c);
    for (uint64_t c = 0; c < nchunks && st == MI_OK; ++c) {
        if (c + AHEAD < nchunks) { st = upload_launch(c + AHEAD); if (st) break; }
        if (hipStreamWaitEvent(cout, ev_dec[c % NE], 0) != hipSuccess ||
            hipMemcpyAsync(h_out + c * C, out.as<uint8_t>() + c * C, chunk_len(c), hipMemcpyDeviceToHost, cout) != hipSuccess) { st = MI_ERR_HIP; break; }
    }
    // nothing may outlive the buffers: drain every stream whatever happened
    if (cin) (void)hipStreamSynchronize(cin);
    for (uint32_t k = 0; k < 4; ++k) if (k == 0 || sx[k]) (void)hipStreamSynchronize(sx[k]);
    if (cout) (void)hipStreamSynchronize(cout);
    uint32_t h_err = 0;
    if (st == MI_OK && hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    for (uint32_t b = 0; b < NE; ++b) { if (ev_in[b]) (void)hipEventDestroy(ev_in[b]); if (ev_dec[b]) (void)hipEventDestroy(ev_dec[b]); }
    if (ev_setup) (void)hipEventDestroy(ev_setup);
    if (cin) (void)hipStreamDestroy(cin);
    if (cout) (void)hipStreamDestroy(cout);
    for (uint32_t k = 1; k < 4; ++k) if (sx[k]) (void)hipStreamDestroy(sx[k]);
    if (st == MI_OK && h_err) st = MI_ERR_CORRUPT;       // (the caller's buffer may hold the chunks that came down before the bad one)
    if (st == MI_OK) *done = true;
    return st;
}

extern "C" mi_status mi_huffman_encode2(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, uint32_t *h_words, uint64_t cap_words,
                                        mi_huffman_info *h_info, mi_huffman_tree *h_tree, uint64_t *h_tile_off)
{
    if (!ctx || !h_words || !h_info || (n && !h_in)) return MI_ERR_ARG;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t ntiles = (n + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE;
    DevBuf in, words, info, tree, toff;
    if (!in.alloc(n + 16) || !words.alloc(cap_words * 4) || !info.alloc(sizeof(mi_huffman_info)) ||
        !tree.alloc(sizeof(mi_huffman_tree)) || !toff.alloc((ntiles + 1) * 8)) return MI_ERR_NOMEM;
    if (n) MI_HIP(ctx, hipMemcpyAsync(in.p, h_in, n, hipMemcpyHostToDevice, s));
    mi_status st = mi_huffman_encode_dev(ctx, in.as<uint8_t>(), n, words.as<uint32_t>(), cap_words, info.as<mi_huffman_info>(),
                                         tree.as<mi_huffman_tree>(), toff.as<uint64_t>(), s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpyAsync(h_info, info.p, sizeof(*h_info), hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    if (h_tree) MI_HIP(ctx, hipMemcpy(h_tree, tree.p, sizeof(*h_tree), hipMemcpyDeviceToHost));
    if (h_info->status != MI_OK) return (mi_status)h_info->status;
    const uint64_t nw = (h_info->total_bits + 31) >> 5;
    if (nw > cap_words) return MI_ERR_CAPACITY;
    if (nw) MI_HIP(ctx, hipMemcpy(h_words, words.p, nw * 4, hipMemcpyDeviceToHost));
    if (h_tile_off) MI_HIP(ctx, hipMemcpy(h_tile_off, toff.p, (ntiles + 1) * 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" mi_status mi_huffman_decode(mi_ctx *ctx, const uint32_t *h_words, uint64_t total_bits, const mi_huffman_tree *h_tree,
                                       uint32_t n_nodes, const uint64_t *h_tile_off, uint8_t *h_out, uint64_t n)
{
    if (!ctx || !h_words || !h_tree || (n && !h_out)) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t nw = (total_bits + 31) >> 5, ntiles = (n + MI_HUFFMAN_TILE - 1) / MI_HUFFMAN_TILE;
    if (h_tile_off) {                                   // tile offsets index the words: monotonic and inside the stream
        mi_status vt = mi_validate_block_table(h_tile_off, ntiles, nw * 4, 1u);
        if (vt) return vt;
        if (h_tile_off[ntiles] > total_bits) return MI_ERR_CORRUPT;
    }
    DevBuf words, tree, toff, out;
    if (!words.alloc((nw + 2) * 4) || !tree.alloc(sizeof(mi_huffman_tree)) || !toff.alloc((ntiles + 1) * 8) || !out.alloc(n + 16))
        return MI_ERR_NOMEM;
    MI_HIP(ctx, hipMemsetAsync(words.as<uint32_t>() + nw, 0, 8, s));
    if (nw) MI_HIP(ctx, hipMemcpyAsync(words.p, h_words, nw * 4, hipMemcpyHostToDevice, s));
    MI_HIP(ctx, hipMemcpyAsync(tree.p, h_tree, sizeof(*h_tree), hipMemcpyHostToDevice, s));
    if (h_tile_off) MI_HIP(ctx, hipMemcpyAsync(toff.p, h_tile_off, (ntiles + 1) * 8, hipMemcpyHostToDevice, s));
    mi_status st = mi_huffman_decode_dev(ctx, words.as<uint32_t>(), total_bits, tree.as<mi_huffman_tree>(), n_nodes,
                                         h_tile_off ? toff.as<uint64_t>() : nullptr, out.as<uint8_t>(), n, s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpy(h_out, out.p, n, hipMemcpyDeviceToHost));
    return MI_OK;
}

// histogram + tree build alone, host buffer in (the drop-in's build_huffman_tree, huffman.c:179-215)
extern "C" mi_status mi_huffman_build(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, mi_huffman_info *h_info, mi_huffman_tree *h_tree)
{
    if (!ctx || !h_info || !h_tree || (n && !h_in)) return MI_ERR_ARG;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t ntiles = mi_huffman_num_tiles(n);
    DevBuf in, hist, thist, info, tree;
    if (!in.alloc(n + 16) || !hist.alloc(256 * 8) || !thist.alloc((ntiles + 1) * 1024) || !info.alloc(sizeof(mi_huffman_info)) ||
        !tree.alloc(sizeof(mi_huffman_tree))) return MI_ERR_NOMEM;
    if (n) MI_HIP(ctx, hipMemcpyAsync(in.p, h_in, n, hipMemcpyHostToDevice, s));
    mi_status st = mi_huffman_hist_dev(ctx, in.as<uint8_t>(), n, hist.as<uint64_t>(), thist.as<uint32_t>(), s);
    if (st) return st;
    st = mi_huffman_build_dev(ctx, hist.as<uint64_t>(), info.as<mi_huffman_info>(), tree.as<mi_huffman_tree>(), s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpyAsync(h_info, info.p, sizeof(*h_info), hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipMemcpyAsync(h_tree, tree.p, sizeof(*h_tree), hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    return (mi_status)h_info->status;
}

// pack with GIVEN codes from a bit offset (the drop-in's _huffman_compress, huffman.c:267-285, which appends to a
// BitWriter wherever it stands).  h_words[0] holds the stream from bit `bit_offset` on (its first bit_offset bits are 0).
extern "C" mi_status mi_huffman_encode_with_codes(mi_ctx *ctx, const uint8_t *h_in, uint64_t n, const uint32_t *h_codes,
                                                  const uint8_t *h_lengths, uint32_t bit_offset, uint32_t *h_words,
                                                  uint64_t cap_words, mi_huffman_info *h_info)
{
    if (!ctx || !h_codes || !h_lengths || !h_words || !h_info || (n && !h_in) || bit_offset > 31 || cap_words < 2) return MI_ERR_ARG;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t ntiles = mi_huffman_num_tiles(n);
    mi_huffman_tree t;
    memset(&t, 0, sizeof t);
    memcpy(t.code, h_codes, sizeof t.code); memcpy(t.length, h_lengths, sizeof t.length);
    DevBuf in, hist, thist, info, tree, words;
    if (!in.alloc(n + 16) || !hist.alloc(256 * 8) || !thist.alloc((ntiles + 1) * 1024) || !info.alloc(sizeof(mi_huffman_info)) ||
        !tree.alloc(sizeof(mi_huffman_tree)) || !words.alloc(cap_words * 4)) return MI_ERR_NOMEM;
    if (n) MI_HIP(ctx, hipMemcpyAsync(in.p, h_in, n, hipMemcpyHostToDevice, s));
    MI_HIP(ctx, hipMemcpyAsync(tree.p, &t, sizeof t, hipMemcpyHostToDevice, s));
    mi_status st = mi_huffman_hist_dev(ctx, in.as<uint8_t>(), n, hist.as<uint64_t>(), thist.as<uint32_t>(), s);
    if (st) return st;
    st = mi_huffman_encode_with_tree_dev(ctx, in.as<uint8_t>(), n, tree.as<mi_huffman_tree>(), thist.as<uint32_t>(), bit_offset,
                                         words.as<uint32_t>(), cap_words, info.as<mi_huffman_info>(), nullptr, s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpyAsync(h_info, info.p, sizeof(*h_info), hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    if (h_info->status != MI_OK) return (mi_status)h_info->status;
    const uint64_t nw = (h_info->total_bits + 31) >> 5;
    if (nw > cap_words) return MI_ERR_CAPACITY;
    if (nw) MI_HIP(ctx, hipMemcpy(h_words, words.p, nw * 4, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" mi_status mi_lz_decode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_stream, uint64_t stream_bytes,
                                  const uint64_t *h_block_bits, uint8_t *h_out, uint64_t n)
{
    if (!ctx || !p || !h_stream || !h_block_bits || (n && !h_out) || !p->block) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t nblocks = (n + p->block - 1) / p->block;
    // the table indexes the stream: check it before anything is copied or launched
    mi_status st = mi_validate_block_table(h_block_bits, nblocks, stream_bytes, p->deflate ? 8u : 1u);
    if (st) return st;
    {
        bool done = false;
        st = mi_decode_host_pipelined(ctx, p, 0, h_stream, stream_bytes, h_block_bits, h_out, n, &done);
        if (st || done) return st;
    }
    DevBuf st_, bits, out;
    if (!st_.alloc(stream_bytes + 64) || !bits.alloc((nblocks + 1) * 8) || !out.alloc(n + 16)) return MI_ERR_NOMEM;
    MI_HIP(ctx, hipMemsetAsync(st_.as<uint8_t>() + stream_bytes, 0, 64, s));
    MI_HIP(ctx, hipMemcpyAsync(st_.p, h_stream, stream_bytes, hipMemcpyHostToDevice, s));
    MI_HIP(ctx, hipMemcpyAsync(bits.p, h_block_bits, (nblocks + 1) * 8, hipMemcpyHostToDevice, s));
    st = mi_lz_decode_dev(ctx, p, st_.as<uint8_t>(), stream_bytes, bits.as<uint64_t>(), out.as<uint8_t>(), n, s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpy(h_out, out.p, n, hipMemcpyDeviceToHost));
    return MI_OK;
}

static mi_status deflate_h_encode_once(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                       uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits);

// A synchronous entry point has its result in hand when it returns, so an order violation reported while it ran (lz_common.h
// lz_order_violation) is repaired here: the context has switched to ballot ranking, the call encodes once more.
mi_status mi_encode_again_if_unstable(mi_ctx *ctx, uint32_t seen_before, mi_status st, mi_status (*again)(void *), void *arg)
{
    mi_order_poll(ctx);                                                     // (chunked paths poll between their chunks too)
    if (st != MI_OK || ctx->order_violations == seen_before) return st;
    const uint32_t seen = ctx->order_violations;
    st = again(arg);
    mi_order_poll(ctx);
    if (st == MI_OK && ctx->order_violations != seen) st = MI_ERR_UNSTABLE; // ballots cannot mis-rank: this does not happen
    ctx->order_reported = ctx->order_violations;                            // handled here: mi_sync need not repeat it
    return st;
}

struct HostEncArgs { mi_ctx *ctx; const mi_lz_params *p; const uint8_t *h_in; uint64_t n; uint8_t *h_out; uint64_t cap; uint64_t *bits; };

extern "C" mi_status mi_deflate_h_encode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                         uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    if (!ctx || !p || !h_out || !h_block_bits || (n && !h_in) || !p->block) return MI_ERR_ARG;
    mi_order_poll(ctx);
    const uint32_t seen = ctx->order_violations;
    HostEncArgs a{ctx, p, h_in, n, h_out, cap_bytes, h_block_bits};
    return mi_encode_again_if_unstable(ctx, seen, deflate_h_encode_once(ctx, p, h_in, n, h_out, cap_bytes, h_block_bits),
        [](void *v) { HostEncArgs *q = (HostEncArgs *)v; return deflate_h_encode_once(q->ctx, q->p, q->h_in, q->n, q->h_out, q->cap, q->bits); }, &a);
}

static mi_status deflate_h_encode_once(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                       uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    {
        bool done = false;
        const mi_status ps = mi_encode_host_pipelined(ctx, p, 1, h_in, n, h_out, cap_bytes, h_block_bits, &done);
        if (ps || done) return ps;
    }
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t nblocks = (n + p->block - 1) / p->block, bound = mi_deflate_h_bound_bytes(n, p);
    DevBuf in, out, bits;
    if (!in.alloc(n + 64) || !out.alloc(bound + 64) || !bits.alloc((nblocks + 1) * 8)) return MI_ERR_NOMEM;
    if (n) MI_HIP(ctx, hipMemcpyAsync(in.p, h_in, n, hipMemcpyHostToDevice, s));
    mi_status st = mi_deflate_h_encode_dev(ctx, p, in.as<uint8_t>(), n, out.as<uint8_t>(), bound + 64, bits.as<uint64_t>(), s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpyAsync(h_block_bits, bits.p, (nblocks + 1) * 8, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    const uint64_t bytes = h_block_bits[nblocks] / 8;
    if (bytes > cap_bytes) return MI_ERR_CAPACITY;
    if (bytes) MI_HIP(ctx, hipMemcpy(h_out, out.p, bytes, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" mi_status mi_deflate_h_decode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_stream, uint64_t stream_bytes,
                                         const uint64_t *h_block_bits, uint8_t *h_out, uint64_t n)
{
    if (!ctx || !p || !h_stream || !h_block_bits || (n && !h_out) || !p->block) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t nblocks = (n + p->block - 1) / p->block;
    mi_status st = mi_validate_block_table(h_block_bits, nblocks, stream_bytes, 32u);
    if (st) return st;
    {
        bool done = false;
        st = mi_decode_host_pipelined(ctx, p, 1, h_stream, stream_bytes, h_block_bits, h_out, n, &done);
        if (st || done) return st;
    }
    DevBuf st_, bits, out;
    if (!st_.alloc(stream_bytes + 64) || !bits.alloc((nblocks + 1) * 8) || !out.alloc(n + 16)) return MI_ERR_NOMEM;
    MI_HIP(ctx, hipMemsetAsync(st_.as<uint8_t>() + stream_bytes, 0, 64, s));
    MI_HIP(ctx, hipMemcpyAsync(st_.p, h_stream, stream_bytes, hipMemcpyHostToDevice, s));
    MI_HIP(ctx, hipMemcpyAsync(bits.p, h_block_bits, (nblocks + 1) * 8, hipMemcpyHostToDevice, s));
    st = mi_deflate_h_decode_dev(ctx, p, st_.as<uint8_t>(), stream_bytes, bits.as<uint64_t>(), out.as<uint8_t>(), n, s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpy(h_out, out.p, n, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" mi_status mi_fse_encode(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *h_in, uint64_t n, uint8_t *h_packed,
                                   uint64_t cap_bytes, uint64_t *h_offsets)
{
    if (!ctx || !p || !h_packed || !h_offsets || (n && !h_in) || !p->block) return MI_ERR_ARG;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t nblocks = (n + p->block - 1) / p->block, need = nblocks * mi_fse_block_bound(p);
    if (cap_bytes < need) return MI_ERR_CAPACITY;
    DevBuf in, out, offs;
    if (!in.alloc(n + 16) || !out.alloc(need + 16) || !offs.alloc((nblocks + 1) * 8)) return MI_ERR_NOMEM;
    if (n) MI_HIP(ctx, hipMemcpyAsync(in.p, h_in, n, hipMemcpyHostToDevice, s));
    mi_status st = mi_fse_encode_dev(ctx, p, in.as<uint8_t>(), n, out.as<uint8_t>(), need + 16, offs.as<uint64_t>(), s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpyAsync(h_offsets, offs.p, (nblocks + 1) * 8, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    const uint64_t bytes = h_offsets[nblocks] / 8;
    if (bytes) MI_HIP(ctx, hipMemcpy(h_packed, out.p, bytes, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" mi_status mi_fse_decode(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *h_packed, uint64_t packed_bytes,
                                   const uint64_t *h_offsets, uint8_t *h_out, uint64_t n)
{
    if (!ctx || !p || !h_packed || !h_offsets || (n && !h_out) || !p->block) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    hipStream_t s = mi_host_stream(ctx);
    const uint64_t nblocks = (n + p->block - 1) / p->block;
    mi_status st = mi_validate_block_table(h_offsets, nblocks, packed_bytes, 32u);
    if (st) return st;
    const uint64_t bytes = h_offsets[nblocks] / 8;
    DevBuf in, offs, out;
    if (!in.alloc(bytes + 16) || !offs.alloc((nblocks + 1) * 8) || !out.alloc(n + 16)) return MI_ERR_NOMEM;
    MI_HIP(ctx, hipMemcpyAsync(in.p, h_packed, bytes, hipMemcpyHostToDevice, s));
    MI_HIP(ctx, hipMemcpyAsync(offs.p, h_offsets, (nblocks + 1) * 8, hipMemcpyHostToDevice, s));
    st = mi_fse_decode_dev(ctx, p, in.as<uint8_t>(), bytes, offs.as<uint64_t>(), out.as<uint8_t>(), n, s);
    if (st) return st;
    MI_HIP(ctx, hipMemcpy(h_out, out.p, n, hipMemcpyDeviceToHost));
    return MI_OK;
}
