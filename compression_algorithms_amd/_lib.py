"""ctypes loader for libmi_codec.so (the C ABI of include/mi_codec.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing or no gfx950 device is present, everything here raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MI_CODEC_LIB") or os.path.join(LIB_DIR, "libmi_codec.so")   # override: A/B builds of the same ABI

MI_OK = 0
STATUS = {0: "MI_OK", 1: "MI_ERR_ARG", 2: "MI_ERR_HIP", 3: "MI_ERR_NOMEM", 4: "MI_ERR_CAPACITY",
          5: "MI_ERR_EMPTY_INPUT", 6: "MI_ERR_SINGLE_SYMBOL", 7: "MI_ERR_CODE_TOO_LONG", 8: "MI_ERR_CORRUPT",
          9: "MI_ERR_NO_DEVICE", 10: "MI_ERR_UNSTABLE", 11: "MI_ERR_TRANSPORT"}


class MiError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = int(status)
        super().__init__(f"{what}: {STATUS.get(self.status, self.status)}")


class HuffmanInfo(C.Structure):
    _fields_ = [("total_bits", C.c_uint64), ("word_idx", C.c_uint64), ("bit_idx", C.c_uint64),
                ("buffer_size", C.c_uint64), ("n_symbols", C.c_uint32), ("max_code_len", C.c_uint32),
                ("status", C.c_uint32), ("n_nodes", C.c_uint32)]


class HuffmanTree(C.Structure):
    _fields_ = [("frequency", C.c_uint32 * 511), ("left", C.c_int16 * 511), ("right", C.c_int16 * 511),
                ("value", C.c_uint8 * 511), ("pad", C.c_uint8), ("code", C.c_uint32 * 256), ("length", C.c_uint8 * 256)]


class LzParams(C.Structure):
    _fields_ = [("wbits", C.c_uint32), ("lbits", C.c_uint32), ("tbits", C.c_uint32), ("deflate", C.c_uint32),
                ("block", C.c_uint32)]


class FseParams(C.Structure):
    _fields_ = [("table_log", C.c_uint32), ("streams", C.c_uint32), ("spread", C.c_uint32), ("block", C.c_uint32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_uint64), ("bytes", C.c_uint64)]


# every symbol include/mi_codec.h declares; tests check that the library exports them all
EXPORTS = [
    "mi_ctx_create", "mi_ctx_destroy", "mi_status_str", "mi_last_hip_error", "mi_version", "mi_sync", "mi_order_violations", "mi_validate_block_table",
    "mi_huffman_encode_dev", "mi_huffman_encode", "mi_huffman_encode2", "mi_huffman_decode_dev", "mi_huffman_decode",
    "mi_huffman_num_tiles", "mi_huffman_hist_dev", "mi_huffman_build_dev", "mi_huffman_encode_with_tree_dev",
    "mi_huffman_build", "mi_huffman_encode_with_codes",
    "mi_lz_encode_dev", "mi_lz_encode", "mi_lz_decode_dev", "mi_lz_decode", "mi_lz_find_all_dev", "mi_lz_find_all32_dev",
    "mi_lz77_old_bound_bytes", "mi_lz77_old_encode_dev", "mi_lz77_old_encode", "mi_lz77_whole_decode_dev", "mi_lz77_whole_decode",
    "mi_deflate_h_bound_bytes", "mi_deflate_h_encode_dev", "mi_deflate_h_decode_dev", "mi_deflate_h_encode", "mi_deflate_h_decode",
    "mi_fse_block_bound", "mi_fse_encode_dev", "mi_fse_decode_dev", "mi_fse_encode", "mi_fse_decode", "mi_fse_normalise_dev",
    "mi_set_profiling", "mi_get_kernel_times",
    "mi_multi_create", "mi_multi_destroy", "mi_multi_ndev", "mi_multi_ctx", "mi_multi_transport", "mi_multi_last_transport_error",
    "mi_multi_shard", "mi_lz_encode_multi_dev", "mi_lz_encode_multi", "mi_deflate_h_encode_multi", "mi_multi_selftest_transport",
    "mi_lz_path_stats",
]

_lib = None


def build(verbose=False):
    """compile libmi_codec.so for gfx950 (hipcc cross-compiles without a GPU)"""
    cmd = ["make", "-C", CSRC_DIR, "-j4"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, u64 = C.c_void_p, C.c_uint64
        L.mi_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
        L.mi_ctx_destroy.argtypes = [vp]
        L.mi_ctx_destroy.restype = None
        L.mi_status_str.restype = C.c_char_p
        L.mi_status_str.argtypes = [C.c_int]
        L.mi_version.restype = C.c_char_p
        L.mi_last_hip_error.argtypes = [vp]
        L.mi_sync.argtypes = [vp, vp]
        if hasattr(L, "mi_order_violations"):                 # (absent from older builds used in A/B runs through MI_CODEC_LIB)
            L.mi_order_violations.restype = C.c_uint32
            L.mi_order_violations.argtypes = [vp]
        L.mi_validate_block_table.argtypes = [vp, u64, u64, C.c_uint32]
        L.mi_set_profiling.argtypes = [vp, C.c_int]
        L.mi_get_kernel_times.argtypes = [vp, C.POINTER(KernelTime), C.c_int]
        if hasattr(L, "mi_huffman_encode_dev"):
            L.mi_huffman_encode_dev.argtypes = [vp, vp, u64, vp, u64, vp, vp, vp, vp]
            L.mi_huffman_encode.argtypes = [vp, vp, u64, vp, u64, C.POINTER(HuffmanInfo), C.POINTER(HuffmanTree)]
        if hasattr(L, "mi_huffman_hist_dev"):
            L.mi_huffman_num_tiles.restype = u64
            L.mi_huffman_num_tiles.argtypes = [u64]
            L.mi_huffman_hist_dev.argtypes = [vp, vp, u64, vp, vp, vp]
            L.mi_huffman_build_dev.argtypes = [vp, vp, vp, vp, vp]
            L.mi_huffman_encode_with_tree_dev.argtypes = [vp, vp, u64, vp, vp, C.c_uint32, vp, u64, vp, vp, vp]
        if hasattr(L, "mi_huffman_decode_dev"):
            L.mi_huffman_decode_dev.argtypes = [vp, vp, u64, vp, C.c_uint32, vp, vp, u64, vp]
        if hasattr(L, "mi_lz_encode_dev"):
            L.mi_lz_encode_dev.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, u64, vp, vp]
            L.mi_lz_encode.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, u64, vp]
            L.mi_lz_decode_dev.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, vp, u64, vp]
            L.mi_lz_decode.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, vp, u64]
            L.mi_lz_find_all_dev.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, vp]
            L.mi_lz_find_all32_dev.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, vp]
        if hasattr(L, "mi_lz77_old_encode_dev"):
            u32 = C.c_uint32
            L.mi_lz77_old_bound_bytes.restype = u64
            L.mi_lz77_old_bound_bytes.argtypes = [u64]
            L.mi_lz77_old_encode_dev.argtypes = [vp, u32, u32, vp, u64, vp, u64, vp, vp]
            L.mi_lz77_old_encode.argtypes = [vp, u32, u32, vp, u64, vp, u64, vp]
            L.mi_lz77_whole_decode_dev.argtypes = [vp, u32, u32, vp, u64, u64, vp, u64, vp]
            L.mi_lz77_whole_decode.argtypes = [vp, u32, u32, vp, u64, u64, vp, u64]
        if hasattr(L, "mi_deflate_h_encode_dev"):
            L.mi_deflate_h_bound_bytes.restype = u64
            L.mi_deflate_h_bound_bytes.argtypes = [u64, C.POINTER(LzParams)]
            L.mi_deflate_h_encode_dev.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, u64, vp, vp]
            L.mi_deflate_h_decode_dev.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, vp, u64, vp]
            L.mi_deflate_h_encode.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, u64, vp]
            L.mi_deflate_h_decode.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, vp, u64]
        if hasattr(L, "mi_fse_encode_dev"):
            L.mi_fse_block_bound.restype = u64
            L.mi_fse_block_bound.argtypes = [C.POINTER(FseParams)]
            L.mi_fse_encode_dev.argtypes = [vp, C.POINTER(FseParams), vp, u64, vp, u64, vp, vp]
            L.mi_fse_decode_dev.argtypes = [vp, C.POINTER(FseParams), vp, u64, vp, vp, u64, vp]
            L.mi_fse_decode.argtypes = [vp, C.POINTER(FseParams), vp, u64, vp, vp, u64]
            L.mi_fse_normalise_dev.argtypes = [vp, vp, C.c_uint32, vp, vp]
        if hasattr(L, "mi_multi_create"):
            L.mi_multi_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
            L.mi_multi_destroy.argtypes = [vp]
            L.mi_multi_destroy.restype = None
            L.mi_multi_ndev.argtypes = [vp]
            L.mi_multi_ctx.argtypes = [vp, C.c_int]
            L.mi_multi_ctx.restype = vp
            L.mi_multi_transport.argtypes = [vp]
            L.mi_multi_transport.restype = C.c_char_p
            L.mi_multi_last_transport_error.argtypes = [vp]
            L.mi_multi_last_transport_error.restype = C.c_char_p
            L.mi_multi_shard.argtypes = [u64, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64)]
            L.mi_multi_shard.restype = None
            L.mi_lz_encode_multi_dev.argtypes = [vp, C.POINTER(LzParams), C.c_int, C.POINTER(vp), u64, vp, u64, vp]
            L.mi_lz_encode_multi.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, u64, vp]
            L.mi_deflate_h_encode_multi.argtypes = [vp, C.POINTER(LzParams), vp, u64, vp, u64, vp]
            L.mi_multi_selftest_transport.argtypes = [vp, u64]
        if hasattr(L, "mi_lz_path_stats"):
            L.mi_lz_path_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        _lib = L
    return _lib


def check(status, what):
    if status != MI_OK:
        raise MiError(status, what)
