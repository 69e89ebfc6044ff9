"""Several GPUs of one node from ONE process: the mi_multi entry points of include/mi_codec.h (csrc/multi.hip).

BASELINE config 5 behind the C boundary — the reference's block loop (algorithms/deflate/deflate.c:47-63) spread over the
listed devices, one context each, the streams gathered into the first device over RCCL (or peer copies when a device is
listed twice: the one-GPU test shape).  bench.py --gpus N (one process per GPU over torch.distributed) is the other way in.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .lz import LzStream, bound_bytes, params


class Multi:
    def __init__(self, devices):
        self.L = _lib.lib()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _lib.check(self.L.mi_multi_create(C.byref(h), arr, len(self.devices)), "mi_multi_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def transport(self):
        return self.L.mi_multi_transport(self.h).decode()

    def transport_error(self):
        return self.L.mi_multi_last_transport_error(self.h).decode()

    def shard(self, nblocks, g):
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        self.L.mi_multi_shard(nblocks, g, len(self.devices), C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    def selftest_transport(self, nbytes=1 << 20):
        _lib.check(self.L.mi_multi_selftest_transport(self.h, nbytes), "mi_multi_selftest_transport: " + self.transport_error())

    def compress_dev(self, shards, n, p, mode_h=False):
        """shards[g]: uint8 tensor on devices[g] holding device g's block range (None for an empty range).  Returns an
        LzStream on devices[0] — byte-identical to lz.compress / lz.compress_h of the whole buffer on one GPU."""
        nd = len(self.devices)
        dev0 = torch.device("cuda", self.devices[0])
        nblocks = (n + p.block - 1) // p.block
        cap = (int(self.L.mi_deflate_h_bound_bytes(n, C.byref(p))) if mode_h else bound_bytes(n, p)) + 64
        out = torch.empty(cap, dtype=torch.uint8, device=dev0)
        bits = torch.zeros(nblocks + 1, dtype=torch.int64, device=dev0)
        ptrs = (C.c_void_p * nd)(*[C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None for t in shards])
        for d in set(self.devices):
            torch.cuda.synchronize(d)                      # the shards were written on torch's streams
        st = self.L.mi_lz_encode_multi_dev(self.h, C.byref(p), 1 if mode_h else 0, ptrs, n, C.c_void_p(out.data_ptr()), cap,
                                           C.c_void_p(bits.data_ptr()))
        _lib.check(st, "mi_lz_encode_multi_dev " + self.transport_error())
        return LzStream(out, bits, n, p)

    def compress_host(self, data, p=None, mode_h=False):
        """host bytes in, (stream bytes, block table) out through mi_lz_encode_multi / mi_deflate_h_encode_multi"""
        p = p or params("deflate")
        a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        n = a.size
        nblocks = (n + p.block - 1) // p.block
        cap = (int(self.L.mi_deflate_h_bound_bytes(n, C.byref(p))) if mode_h else bound_bytes(n, p)) + 64
        out = np.empty(cap, dtype=np.uint8)
        bits = np.zeros(nblocks + 1, dtype=np.uint64)
        f = self.L.mi_deflate_h_encode_multi if mode_h else self.L.mi_lz_encode_multi
        st = f(self.h, C.byref(p), C.c_void_p(a.ctypes.data if n else 0), n, C.c_void_p(out.ctypes.data), cap, C.c_void_p(bits.ctypes.data))
        _lib.check(st, "mi_lz_encode_multi " + self.transport_error())
        return out[: (int(bits[-1]) + 7) // 8].copy(), bits
