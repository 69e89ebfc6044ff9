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
