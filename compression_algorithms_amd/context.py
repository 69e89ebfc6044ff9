"""One codec context per process-and-GPU (mi_ctx of include/mi_codec.h), plus the small
torch plumbing the Python layer needs: device buffers are torch tensors, only their
data_ptr() crosses the C ABI."""
import ctypes as C

import torch

from . import _lib


class Context:
    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise _lib.MiError(9, "no HIP device visible (the codec has no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self.L.mi_ctx_create(C.byref(h), device), "mi_ctx_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self):
        """mi_sync on the current torch stream: raises MiError(MI_ERR_UNSTABLE) if an encoder reported a sort out of order"""
        _lib.check(self.L.mi_sync(self.h, self.stream_ptr()), "mi_sync")

    def order_violations(self):
        """sorts found out of (key, time) order by the kernels of this context so far (include/mi_codec.h); 0 on gfx950"""
        return int(self.L.mi_order_violations(self.h))

    def path_stats(self):
        """{fallback_blocks, wide_parts} the LZ encoders of this context met since it was created (mi_lz_path_stats)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.L.mi_lz_path_stats(self.h, C.byref(a), C.byref(b)), "mi_lz_path_stats")
        return dict(fallback_blocks=int(a.value), wide_parts=int(b.value))

    def set_profiling(self, on=True):
        _lib.check(self.L.mi_set_profiling(self.h, 1 if on else 0), "mi_set_profiling")

    def kernel_times(self):
        arr = (_lib.KernelTime * 64)()
        k = self.L.mi_get_kernel_times(self.h, arr, 64)
        return [dict(name=arr[i].name.decode(), ms=arr[i].ms, launches=int(arr[i].launches), bytes=int(arr[i].bytes))
                for i in range(k)]


_default = {}


def default_context(device=None):
    if device is None:
        device = torch.cuda.current_device()
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


def as_device_bytes(data, device):
    """bytes / numpy / tensor -> contiguous uint8 tensor on `device`, 16-byte aligned"""
    if isinstance(data, torch.Tensor):
        t = data.to(device=device, dtype=torch.uint8).contiguous()
    else:
        import numpy as np
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        t = torch.from_numpy(np.ascontiguousarray(a).copy()).to(device)
    if t.data_ptr() % 16:
        t = t.clone()
    return t
