"""ctypes mirror of include/mi_frame.h: the self-describing container (serialised tree / {last_block, size} chunks, after
algorithms/huffman/zig_huffman/src/main.zig:11-18,155-176,513-530).  Host-only code of libmi_codec.so: works without a GPU."""
import ctypes as C

import numpy as np

from . import _lib

HUFFMAN, DEFLATE_T, DEFLATE_H, LZ77, FSE = 1, 2, 3, 4, 5


class FrameInfo(C.Structure):
    _fields_ = [("codec", C.c_uint32), ("block", C.c_uint32), ("p0", C.c_uint32), ("p1", C.c_uint32),
                ("original_size", C.c_uint64), ("nblocks", C.c_uint64), ("stream_bytes", C.c_uint64), ("total_bits", C.c_uint64)]


def _L():
    L = _lib.lib()
    if not getattr(L, "_frame_ready", False):
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.mi_frame_bound_blocks.restype = u64
        L.mi_frame_bound_blocks.argtypes = [u64, u64]
        L.mi_frame_pack_blocks.argtypes = [u32, u32, u32, u32, u64, vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.mi_frame_parse.argtypes = [vp, u64, C.POINTER(FrameInfo)]
        L.mi_frame_unpack_blocks.argtypes = [vp, u64, vp, u64, vp, u64]
        L.mi_frame_bound_huffman.restype = u64
        L.mi_frame_bound_huffman.argtypes = [u64, u64]
        L.mi_frame_pack_huffman.argtypes = [u64, C.POINTER(_lib.HuffmanTree), u32, vp, u64, vp, u64, vp, u64, C.POINTER(u64)]
        L.mi_frame_unpack_huffman.argtypes = [vp, u64, C.POINTER(_lib.HuffmanTree), C.POINTER(u32), vp, u64, C.POINTER(u64), vp, u64]
        L._frame_ready = True
    return L


def pack_blocks(codec, block, p0, p1, original_size, stream, block_bits):
    """stream: uint8 array, block_bits: uint64 array [nblocks+1] -> frame bytes"""
    L = _L()
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    t = np.ascontiguousarray(block_bits, dtype=np.uint64)
    nblocks = len(t) - 1
    cap = int(L.mi_frame_bound_blocks(nblocks, len(stream)))
    out = np.zeros(cap, np.uint8)
    got = C.c_uint64(0)
    src = np.concatenate([stream, np.zeros(8, np.uint8)])
    _lib.check(L.mi_frame_pack_blocks(codec, block, p0, p1, original_size, src.ctypes.data, t.ctypes.data, nblocks,
                                      out.ctypes.data, cap, C.byref(got)), "mi_frame_pack_blocks")
    return out[: got.value].tobytes()


def parse(frame):
    L = _L()
    buf = np.frombuffer(bytes(frame), dtype=np.uint8)
    info = FrameInfo()
    _lib.check(L.mi_frame_parse(buf.ctypes.data if len(buf) else None, len(buf), C.byref(info)), "mi_frame_parse")
    return info


def unpack_blocks(frame):
    """-> (info, stream uint8 array, block_bits uint64 array)"""
    L = _L()
    buf = np.frombuffer(bytes(frame), dtype=np.uint8)
    info = parse(frame)
    stream = np.zeros(info.stream_bytes + 8, np.uint8)
    t = np.zeros(info.nblocks + 1, np.uint64)
    _lib.check(L.mi_frame_unpack_blocks(buf.ctypes.data, len(buf), stream.ctypes.data, len(stream), t.ctypes.data, len(t)),
               "mi_frame_unpack_blocks")
    return info, stream[: info.stream_bytes], t


def pack_huffman(original_size, tree, n_nodes, words, total_bits, tile_off=None):
    L = _L()
    w = np.ascontiguousarray(words, dtype=np.uint32)
    to = np.ascontiguousarray(tile_off, dtype=np.uint64) if tile_off is not None else np.zeros(0, np.uint64)
    ntiles = max(len(to) - 1, 0)
    cap = int(L.mi_frame_bound_huffman(total_bits, ntiles))
    out = np.zeros(cap, np.uint8)
    got = C.c_uint64(0)
    _lib.check(L.mi_frame_pack_huffman(original_size, C.byref(tree), n_nodes, w.ctypes.data, total_bits,
                                       to.ctypes.data if ntiles else None, ntiles, out.ctypes.data, cap, C.byref(got)),
               "mi_frame_pack_huffman")
    return out[: got.value].tobytes()


def unpack_huffman(frame):
    """-> (info, tree, n_nodes, words uint32 array, total_bits, tile_off uint64 array)"""
    L = _L()
    buf = np.frombuffer(bytes(frame), dtype=np.uint8)
    info = parse(frame)
    tree = _lib.HuffmanTree()
    nn, bits = C.c_uint32(0), C.c_uint64(0)
    words = np.zeros(info.stream_bytes // 4 + 2, np.uint32)
    to = np.zeros(info.nblocks + 1, np.uint64)
    _lib.check(L.mi_frame_unpack_huffman(buf.ctypes.data, len(buf), C.byref(tree), C.byref(nn), words.ctypes.data, len(words),
                                         C.byref(bits), to.ctypes.data, len(to)), "mi_frame_unpack_huffman")
    return info, tree, nn.value, words[: info.stream_bytes // 4], bits.value, to
