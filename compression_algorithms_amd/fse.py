"""Host-side mirror of the reference's FSE program (algorithms/fse/src/main.zig: histogram ->
normalise -> table build -> reverse-order encode) over the HIP path.  The reference file is
an unfinished sketch; the stream format is this build's (include/mi_fse.h)."""
import ctypes as C

import torch

from . import _lib
from .context import as_device_bytes, default_context


def params(table_log=8, streams=64, spread=1, block=65536):
    return _lib.FseParams(table_log, streams, spread, block)


class FseStream:
    def __init__(self, data, offsets, n, p):
        self.data, self.offsets, self.n, self.p = data, offsets, n, p

    @property
    def nbytes(self):
        return int(self.offsets[-1]) // 8

    def record(self, b):
        a, e = int(self.offsets[b]) // 8, int(self.offsets[b + 1]) // 8
        return self.data[a:e]

    def tobytes(self):
        return self.data[: self.nbytes].cpu().numpy().tobytes()


def block_bound(p, ctx=None):
    ctx = ctx or default_context()
    return int(ctx.L.mi_fse_block_bound(C.byref(p)))


def compress(data, p=None, ctx=None):
    ctx = ctx or default_context()
    p = p or params()
    d_in = as_device_bytes(data, ctx.device)
    n = d_in.numel()
    nblocks = (n + p.block - 1) // p.block
    cap = max(nblocks, 1) * block_bound(p, ctx)
    out = torch.empty(cap, dtype=torch.uint8, device=ctx.device)
    offs = torch.zeros(nblocks + 1, dtype=torch.int64, device=ctx.device)
    st = ctx.L.mi_fse_encode_dev(ctx.h, C.byref(p), C.c_void_p(d_in.data_ptr() if n else 0), n, C.c_void_p(out.data_ptr()), cap,
                                 C.c_void_p(offs.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_fse_encode_dev")
    return FseStream(out, offs, n, p)


def decompress(stream, ctx=None):
    ctx = ctx or default_context()
    out = torch.empty(max(stream.n, 1), dtype=torch.uint8, device=ctx.device)
    st = ctx.L.mi_fse_decode_dev(ctx.h, C.byref(stream.p), C.c_void_p(stream.data.data_ptr()), stream.data.numel(),
                                 C.c_void_p(stream.offsets.data_ptr()), C.c_void_p(out.data_ptr()), stream.n, ctx.stream_ptr())
    _lib.check(st, "mi_fse_decode_dev")
    return out[: stream.n]


def normalise(freq, table_log=8, ctx=None):
    """the normalisation rule of main.zig:106-149 alone: int64 counts[256] -> int32 normalised[256]"""
    ctx = ctx or default_context()
    f = torch.as_tensor(freq, dtype=torch.int64).to(ctx.device).contiguous()
    cnt = torch.zeros(256, dtype=torch.int32, device=ctx.device)
    st = ctx.L.mi_fse_normalise_dev(ctx.h, C.c_void_p(f.data_ptr()), table_log, C.c_void_p(cnt.data_ptr()), ctx.stream_ptr())
    _lib.check(st, "mi_fse_normalise_dev")
    return cnt
