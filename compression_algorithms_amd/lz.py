"""Host-side mirror of the reference's LZ77 interfaces over the HIP path.

  lz77 flavour     algorithms/lz77/lz77.h:55-63    lz77_compress / lz77_decompress
  deflate flavour  algorithms/deflate/lz77.h:47-53 per-block lz77_compress (fresh table per block)

Both are block-parallel: the input is cut into `block`-byte blocks that are encoded
independently exactly as the reference encodes a buffer of that size.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .context import as_device_bytes, default_context


def params(flavour="deflate", wbits=None, block=65536):
    if flavour == "deflate":
        return _lib.LzParams(15, 5, 20, 1, block)
    if flavour == "lz77":
        wb = 14 if wbits is None else wbits
        return _lib.LzParams(wb, 4, wb + 6, 0, block)
    raise ValueError(flavour)


def find_all(data, p, ctx=None):
    """find() at every position of every block -> uint16 tensor (0xFFFF = none)."""
    ctx = ctx or default_context()
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    cand = torch.empty(max(n, 1), dtype=torch.int16, device=ctx.device)
    st = ctx.L.mi_lz_find_all_dev(ctx.h, C.byref(p), C.c_void_p(d_in.data_ptr()), n, C.c_void_p(cand.data_ptr()),
                                  ctx.stream_ptr())
    _lib.check(st, "mi_lz_find_all_dev")
    return cand[:n]


def find_all32(data, p, ctx=None):
    """find() at every position of every block, for blocks above 64 KiB -> int32 tensor viewed as uint32 (0xFFFFFFFF = none)."""
    ctx = ctx or default_context()
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    cand = torch.empty(max(n, 1), dtype=torch.int32, device=ctx.device)
    st = ctx.L.mi_lz_find_all32_dev(ctx.h, C.byref(p), C.c_void_p(d_in.data_ptr()), n, C.c_void_p(cand.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_lz_find_all32_dev")
    return cand[:n]


class LzStream:
    """A block-parallel LZ77 stream on the device.

    data        uint8 tensor, the concatenated block streams (lz77 flavour: bit-contiguous)
    block_bits  int64 tensor [nblocks+1], exclusive prefix of per-block lengths in BITS
    """

    def __init__(self, data, block_bits, n, p, ctx=None, violations_before=0):
        self.data, self.block_bits, self.n, self.p = data, block_bits, n, p
        self._ctx, self._v0 = ctx, violations_before

    def check_order(self):
        """mi_codec.h promises that a sort found out of (key, time) order is never silent: the asynchronous encoder cannot
        say so when it returns, so the first read of its result does — MiError(MI_ERR_UNSTABLE); the context ranks
        with ballots from then on, encode again."""
        if self._ctx is not None and self._ctx.order_violations() != self._v0:
            self._v0 = self._ctx.order_violations()
            raise _lib.MiError(10, "the encoder that wrote this stream reported a sort out of order")

    @property
    def total_bits(self):
        t = int(self.block_bits[-1])                       # (synchronises: the encoder has joined torch's stream)
        self.check_order()
        return t

    @property
    def nbytes(self):
        return (self.total_bits + 7) // 8

    def tobytes(self):
        return self.data[: self.nbytes].cpu().numpy().tobytes()


def bound_bytes(n, p):
    """mi_lz_bound_bytes of include/mi_codec.h (a static inline there)"""
    nblocks = (n + p.block - 1) // p.block
    if p.deflate:
        return 2 * n + 2 * nblocks + 8
    return (9 * n + nblocks * (p.wbits + p.lbits - 8) + 7) // 8 + 16


def compress(data, p, ctx=None):
    """Encode `data` as independent `p.block`-byte blocks (reference semantics per block)."""
    ctx = ctx or default_context()
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    nblocks = (n + p.block - 1) // p.block
    cap = bound_bytes(n, p) + 64
    out = torch.empty(cap, dtype=torch.uint8, device=ctx.device)
    bits = torch.zeros(nblocks + 1, dtype=torch.int64, device=ctx.device)
    v0 = ctx.order_violations()
    st = ctx.L.mi_lz_encode_dev(ctx.h, C.byref(p), C.c_void_p(d_in.data_ptr() if n else 0), n, C.c_void_p(out.data_ptr()), cap,
                                C.c_void_p(bits.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_lz_encode_dev")
    return LzStream(out, bits, n, p, ctx, v0)


def decompress(stream, ctx=None):
    ctx = ctx or default_context()
    out = torch.empty(max(stream.n, 1), dtype=torch.uint8, device=ctx.device)
    st = ctx.L.mi_lz_decode_dev(ctx.h, C.byref(stream.p), C.c_void_p(stream.data.data_ptr()), stream.data.numel(),
                                C.c_void_p(stream.block_bits.data_ptr()), C.c_void_p(out.data_ptr()), stream.n, ctx.stream_ptr())
    _lib.check(st, "mi_lz_decode_dev")
    return out[: stream.n]


# ---- mode H: the reference's deflate token sequence, entropy coded per block (include/mi_codec.h) ----------------
def compress_h(data, p=None, ctx=None):
    """Deflate tokens (algorithms/deflate/lz77.c:199-280) + per-block dynamic Huffman over the 286-symbol
    alphabet the reference sketches (deflate/huffman.c:49-62).  Returns an LzStream whose `data` holds the
    concatenated block records and whose block_bits are record offsets in bits."""
    ctx = ctx or default_context()
    p = p or params("deflate")
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    nblocks = (n + p.block - 1) // p.block
    cap = int(ctx.L.mi_deflate_h_bound_bytes(n, C.byref(p))) + 64
    out = torch.empty(cap, dtype=torch.uint8, device=ctx.device)
    bits = torch.zeros(nblocks + 1, dtype=torch.int64, device=ctx.device)
    v0 = ctx.order_violations()
    st = ctx.L.mi_deflate_h_encode_dev(ctx.h, C.byref(p), C.c_void_p(d_in.data_ptr() if n else 0), n, C.c_void_p(out.data_ptr()),
                                       cap, C.c_void_p(bits.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_deflate_h_encode_dev")
    return LzStream(out, bits, n, p, ctx, v0)


def decompress_h(stream, ctx=None):
    ctx = ctx or default_context()
    out = torch.empty(max(stream.n, 1), dtype=torch.uint8, device=ctx.device)
    st = ctx.L.mi_deflate_h_decode_dev(ctx.h, C.byref(stream.p), C.c_void_p(stream.data.data_ptr()), stream.data.numel(),
                                       C.c_void_p(stream.block_bits.data_ptr()), C.c_void_p(out.data_ptr()), stream.n,
                                       ctx.stream_ptr())
    _lib.check(st, "mi_deflate_h_decode_dev")
    return out[: stream.n]


# ---------------------------------------------------------------------------------------------------------------------
# the reference's first parser: lz77_compress_old (algorithms/lz77/lz77.h:51-54, lz77.c:185-262)
class WholeStream:
    """ONE lz77 stream over a whole buffer (no blocks): `data` uint8 tensor, `total_bits` = the reference's bit_index"""

    def __init__(self, data, total_bits, n, wbits, lbits):
        self.data, self.total_bits, self.n, self.wbits, self.lbits = data, total_bits, n, wbits, lbits

    @property
    def nbytes(self):
        return self.total_bits // 8 + 1                   # lz77.c:258

    def tobytes(self):
        return self.data[: self.nbytes].cpu().numpy().tobytes()


def compress_old(data, wbits=14, lbits=4, ctx=None):
    """lz77_compress_old on the GPU: longest match of the whole 2^wbits - 1 window at every token, first-longest wins, one
    stream over the whole buffer (lz_old.hip).  O(n * 2^wbits) like the reference."""
    ctx = ctx or default_context()
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    cap = int(ctx.L.mi_lz77_old_bound_bytes(n))
    out = torch.empty(cap, dtype=torch.uint8, device=ctx.device)
    bits = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    st = ctx.L.mi_lz77_old_encode_dev(ctx.h, wbits, lbits, C.c_void_p(d_in.data_ptr() if n else 0), n, C.c_void_p(out.data_ptr()), cap,
                                      C.c_void_p(bits.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_lz77_old_encode_dev")
    return WholeStream(out, int(bits.item()), n, wbits, lbits)


def decompress_whole(stream, ctx=None):
    """decode a whole-buffer lz77 stream (lz77.c:347-377) on one wave"""
    ctx = ctx or default_context()
    out = torch.empty(max(stream.n, 1), dtype=torch.uint8, device=ctx.device)
    st = ctx.L.mi_lz77_whole_decode_dev(ctx.h, stream.wbits, stream.lbits, C.c_void_p(stream.data.data_ptr()), stream.data.numel(),
                                        stream.total_bits, C.c_void_p(out.data_ptr()), stream.n, ctx.stream_ptr())
    _lib.check(st, "mi_lz77_whole_decode_dev")
    return out[: stream.n]
