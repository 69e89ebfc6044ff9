"""Host-side mirror of the reference's Huffman interface (algorithms/huffman/huffman.h:90-113)
over the HIP path.  `huffman_compress` keeps the reference's meaning — one tree over the
whole buffer, tree-path codes, MSB-first u32 words — and returns what the reference returns
through its BitWriter and root Node: the words, (word_idx, bit_idx, buffer_size) and the tree.
Errors the reference reports with printf+exit(1) are raised as MiError with the same cause."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .context import as_device_bytes, default_context

TILE = 32768     # MI_HUFFMAN_TILE


class HuffmanResult:
    def __init__(self, words, info, tree):
        self.words = words                      # uint32 tensor on the device, ceil(bits/32) long
        self.total_bits = int(info.total_bits)
        self.word_idx = int(info.word_idx)      # BitWriter.word_idx
        self.bit_idx = int(info.bit_idx)        # BitWriter.bit_idx
        self.buffer_size = int(info.buffer_size)  # BitWriter.buffer_size (huffman.c:318-320)
        self.n_symbols = int(info.n_symbols)
        self.max_code_len = int(info.max_code_len)
        self.n_nodes = int(info.n_nodes)
        self.codes = np.ctypeslib.as_array(tree.code).copy()
        self.lengths = np.ctypeslib.as_array(tree.length).copy()
        self.tree = tree

    def preorder(self):
        """(is_leaf, value, frequency) in pre-order, like walking the reference's Node tree"""
        t = self.tree
        out, stack = [], [self.n_nodes - 1]
        while stack:
            i = stack.pop()
            leaf = t.left[i] < 0
            out.append((1 if leaf else 0, int(t.value[i]), int(t.frequency[i])))
            if not leaf:
                stack.append(t.right[i])
                stack.append(t.left[i])
        return out


def huffman_compress(data, ctx=None):
    """data: bytes / numpy uint8 / uint8 tensor (device tensors are used in place)."""
    ctx = ctx or default_context()
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    # the ABI's worst case is 32 bits per byte; in practice a Huffman stream never exceeds 8 bits per byte of input
    # (the fixed 8-bit code is a valid prefix code), so n/4 + slack words are enough
    cap = n // 4 + 66
    words = torch.empty(cap, dtype=torch.int32, device=ctx.device)
    d_info = torch.zeros(C.sizeof(_lib.HuffmanInfo), dtype=torch.uint8, device=ctx.device)
    d_tree = torch.zeros(C.sizeof(_lib.HuffmanTree), dtype=torch.uint8, device=ctx.device)
    ntiles = (n + TILE - 1) // TILE
    tile_off = torch.zeros(ntiles + 1, dtype=torch.int64, device=ctx.device)
    st = ctx.L.mi_huffman_encode_dev(ctx.h, C.c_void_p(d_in.data_ptr() if n else 0), n, C.c_void_p(words.data_ptr()), cap,
                                     C.c_void_p(d_info.data_ptr()), C.c_void_p(d_tree.data_ptr()),
                                     C.c_void_p(tile_off.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_huffman_encode_dev")
    info = _lib.HuffmanInfo.from_buffer_copy(d_info.cpu().numpy().tobytes())
    if info.status != _lib.MI_OK:
        raise _lib.MiError(info.status, "huffman_compress")
    tree = _lib.HuffmanTree.from_buffer_copy(d_tree.cpu().numpy().tobytes())
    nw = (info.total_bits + 31) // 32
    words[nw:nw + 1] = 0            # the decoder peeks one word past the stream
    r = HuffmanResult(words[:nw + 1][:nw], info, tree)
    r._words_padded = words[:nw + 1]
    r.tile_off = tile_off
    r.n = n
    r._d_tree = d_tree
    return r


def huffman_decompress(result, use_tiles=True, ctx=None):
    """decode exactly result.n symbols (the reference decoder, huffman.c:330-364, overruns into the
    pad bits; the original length is what `*output_size` carries on entry there)"""
    ctx = ctx or default_context()
    out = torch.empty(max(result.n, 1), dtype=torch.uint8, device=ctx.device)
    st = ctx.L.mi_huffman_decode_dev(ctx.h, C.c_void_p(result._words_padded.data_ptr()), result.total_bits,
                                     C.c_void_p(result._d_tree.data_ptr()), result.n_nodes,
                                     C.c_void_p(result.tile_off.data_ptr() if use_tiles else 0),
                                     C.c_void_p(out.data_ptr()), result.n, ctx.stream_ptr())
    _lib.check(st, "mi_huffman_decode_dev")
    return out[: result.n]


class HipShardEngine:
    """The three-step encoder of include/mi_codec.h (mi_huffman_hist_dev / _build_dev / _encode_with_tree_dev) as the
    per-rank engine of sharded.huffman_compress: one tree over a buffer spread over several GPUs
    (algorithms/huffman/huffman.c:179-215 builds it over the WHOLE buffer, :267-328 packs with it)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def hist(self, shard):
        """-> (int64[256] device tensor, state for encode())"""
        ctx = self.ctx
        d_in = as_device_bytes(shard, ctx.device)
        n = d_in.numel()
        ntiles = int(ctx.L.mi_huffman_num_tiles(n))
        h = torch.zeros(256, dtype=torch.int64, device=ctx.device)
        tile_hist = torch.empty(max(ntiles, 1) * 256, dtype=torch.int32, device=ctx.device)
        st = ctx.L.mi_huffman_hist_dev(ctx.h, C.c_void_p(d_in.data_ptr() if n else 0), n, C.c_void_p(h.data_ptr()),
                                       C.c_void_p(tile_hist.data_ptr()), ctx.stream_ptr())
        _lib.check(st, "mi_huffman_hist_dev")
        return h, dict(d_in=d_in, n=n, ntiles=ntiles, tile_hist=tile_hist)

    def build(self, hist):
        """summed histogram -> dict(d_tree, info, tree, lengths int64 device tensor); raises MiError like huffman_compress"""
        ctx = self.ctx
        d_info = torch.zeros(C.sizeof(_lib.HuffmanInfo), dtype=torch.uint8, device=ctx.device)
        d_tree = torch.zeros(C.sizeof(_lib.HuffmanTree), dtype=torch.uint8, device=ctx.device)
        st = ctx.L.mi_huffman_build_dev(ctx.h, C.c_void_p(hist.data_ptr()), C.c_void_p(d_info.data_ptr()),
                                        C.c_void_p(d_tree.data_ptr()), ctx.stream_ptr())
        _lib.check(st, "mi_huffman_build_dev")
        info = _lib.HuffmanInfo.from_buffer_copy(d_info.cpu().numpy().tobytes())
        if info.status != _lib.MI_OK:
            raise _lib.MiError(info.status, "huffman_compress (sharded)")
        tree = _lib.HuffmanTree.from_buffer_copy(d_tree.cpu().numpy().tobytes())
        off = _lib.HuffmanTree.length.offset
        lengths = d_tree[off:off + 256].to(torch.int64)
        return dict(d_tree=d_tree, info=info, tree=tree, lengths=lengths)

    def shard_bits(self, hist, tree):
        return int((hist * tree["lengths"]).sum().item())

    def encode(self, state, tree, bit_offset, nbits):
        """-> (int32 device tensor of ceil((bit_offset + nbits) / 32) words, int64 tile offsets relative to word 0)"""
        ctx = self.ctx
        nw = (bit_offset + nbits + 31) // 32
        words = torch.zeros(nw + 2, dtype=torch.int32, device=ctx.device)
        d_info = torch.zeros(C.sizeof(_lib.HuffmanInfo), dtype=torch.uint8, device=ctx.device)
        tile_off = torch.zeros(state["ntiles"] + 1, dtype=torch.int64, device=ctx.device)
        st = ctx.L.mi_huffman_encode_with_tree_dev(ctx.h, C.c_void_p(state["d_in"].data_ptr() if state["n"] else 0), state["n"],
                                                   C.c_void_p(tree["d_tree"].data_ptr()), C.c_void_p(state["tile_hist"].data_ptr()),
                                                   bit_offset, C.c_void_p(words.data_ptr()), nw + 2, C.c_void_p(d_info.data_ptr()),
                                                   C.c_void_p(tile_off.data_ptr()), ctx.stream_ptr())
        _lib.check(st, "mi_huffman_encode_with_tree_dev")
        info = _lib.HuffmanInfo.from_buffer_copy(d_info.cpu().numpy().tobytes())
        if info.status != _lib.MI_OK:
            raise _lib.MiError(info.status, "mi_huffman_encode_with_tree_dev")
        assert info.total_bits == bit_offset + nbits, (info.total_bits, bit_offset, nbits)
        return words[:nw], tile_off
