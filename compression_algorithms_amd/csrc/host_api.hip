// host_api.hip — host-buffer convenience entry points: copy in, run the *_dev path on the
// context's private stream, copy out, synchronise.  The PCIe-inclusive path of the drop-in
// libraries (dropin_*.c); throughput numbers are quoted on the *_dev entry points.
#include "common.h"

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess; }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};
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

extern "C" mi_status mi_deflate_h_encode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                         uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    if (!ctx || !p || !h_out || !h_block_bits || (n && !h_in) || !p->block) return MI_ERR_ARG;
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
