"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes bindings for oracle/_ref/*.so: the REAL reference, compiled by oracle/Makefile
(`make ref`) from the sources where they lie under /root/reference.  Exists only in the
build container; used to (1) validate the restatement in liborc.so and (2) generate the
golden vectors committed under tests/golden/ (oracle/gen_golden.py).

Every input is handed to the reference in a buffer with >= 64 zero bytes after its end,
so the reference's reads past `size` are defined.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
TAIL = 64


def available():
    return all(os.path.exists(os.path.join(_REF, f)) for f in
               ("liblz77_w14.so", "liblz77_w16.so", "libhuffman.so", "libdeflate.so"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _padded(data):
    a = np.zeros(len(data) + TAIL, dtype=np.uint8)
    a[: len(data)] = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return a


# ----------------------------------------------------------------------------- lz77
class _BitStream(C.Structure):                      # lz77/lz77.h:14-17
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("bit_index", C.c_uint64)]


_lz = {}


def _lz77(wbits):
    if wbits not in _lz:
        L = C.CDLL(os.path.join(_REF, f"liblz77_w{wbits}.so"))
        L.lz77_compress.restype = C.POINTER(_BitStream)
        L.lz77_compress.argtypes = [C.c_void_p, C.c_uint64]
        L.lz77_decompress.restype = C.c_void_p
        L.lz77_decompress.argtypes = [C.POINTER(_BitStream), C.c_uint64, C.POINTER(C.c_uint64)]
        L.hash.restype = C.c_uint32
        L.hash.argtypes = [C.c_uint32]
        _lz[wbits] = L
    return _lz[wbits]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def lz77_compress(data, wbits=14):
    """-> (stream bytes [bit_index//8+1 of them, pad bits masked to 0], bit_index)"""
    L = _lz77(wbits)
    src = _padded(data)
    bs = L.lz77_compress(_p(src), len(data))
    nbits = int(bs.contents.bit_index)
    nbytes = nbits // 8 + 1
    raw = np.ctypeslib.as_array(bs.contents.data, shape=(nbytes,)).copy()
    if nbits % 8:
        raw[nbits // 8] &= (1 << (nbits % 8)) - 1
    else:
        raw[nbits // 8] = 0
    _libc.free(C.cast(bs.contents.data, C.c_void_p))
    _libc.free(C.cast(bs, C.c_void_p))
    return raw, nbits


def lz77_compress_old(data, wbits=14):
    """the reference's lz77_compress_old (lz77.c:185-262) -> (stream bytes, pad bits masked to 0, bit_index)"""
    L = _lz77(wbits)
    L.lz77_compress_old.restype = C.POINTER(_BitStream)
    L.lz77_compress_old.argtypes = [C.c_void_p, C.c_uint64]
    src = _padded(data)
    bs = L.lz77_compress_old(_p(src), len(data))
    nbits = int(bs.contents.bit_index)
    nbytes = nbits // 8 + 1
    raw = np.ctypeslib.as_array(bs.contents.data, shape=(nbytes,)).copy()
    if nbits % 8:
        raw[nbits // 8] &= (1 << (nbits % 8)) - 1
    else:
        raw[nbits // 8] = 0
    _libc.free(C.cast(bs.contents.data, C.c_void_p))
    _libc.free(C.cast(bs, C.c_void_p))
    return raw, nbits


def lz77_hash(word, wbits=14):
    return int(_lz77(wbits).hash(word))


# ----------------------------------------------------------------------------- huffman
class _BitWriter(C.Structure):                      # huffman/huffman.h:42-47
    _fields_ = [("buffer", C.POINTER(C.c_uint32)), ("bit_idx", C.c_uint64),
                ("word_idx", C.c_uint64), ("buffer_size", C.c_uint64)]


class _Node(C.Structure):                           # huffman/huffman.h:54-60
    pass


_Node._fields_ = [("value", C.c_uint8), ("frequency", C.c_uint32),
                  ("left", C.POINTER(_Node)), ("right", C.POINTER(_Node))]

_hf = None


def _huff():
    global _hf
    if _hf is None:
        L = C.CDLL(os.path.join(_REF, "libhuffman.so"))
        L.init_bitwriter.argtypes = [C.POINTER(_BitWriter), C.c_uint64]
        L.build_huffman_tree.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(_Node))]
        L.gather_codes.argtypes = [C.POINTER(_Node), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L._huffman_compress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(_BitWriter)]
        L.huffman_compress.restype = _Node
        L.huffman_compress.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_BitWriter)]
        L.huffman_decompress.argtypes = [C.POINTER(_BitWriter), C.POINTER(_Node), C.c_void_p, C.POINTER(C.c_uint64)]
        _hf = L
    return _hf


def huffman_compress(data):
    """Runs the reference on `data` (needs >= 2 distinct symbols, else it exit(1)s).

    The full 32-bit words are obtained by running huffman_compress's own steps
    (huffman/huffman.c:293-316) without its final realloc, which can drop live bytes of the
    last partial word; (word_idx, bit_idx, buffer_size) come from a second,
    unmodified call of huffman_compress itself.
    """
    L = _huff()
    arr = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    n = len(arr)
    if len(np.unique(arr)) < 2:
        raise ValueError("reference exit(1)s on < 2 distinct symbols")
    src = _padded(arr)
    # pieces
    w = _BitWriter()
    L.init_bitwriter(C.byref(w), n + 8)
    root = C.POINTER(_Node)()
    L.build_huffman_tree(_p(src), n, C.byref(root))
    codes = np.zeros(256, dtype=np.uint32)
    lens = np.zeros(256, dtype=np.uint8)
    L.gather_codes(root, 0, 0, _p(codes), _p(lens))
    bits_total = int((np.bincount(arr, minlength=256).astype(np.uint64) * lens.astype(np.uint64)).sum())
    if bits_total > 8 * n:
        raise ValueError("reference would overflow its size-byte output buffer")
    L._huffman_compress(_p(src), n, _p(codes), _p(lens), C.byref(w))
    nw = int(w.word_idx) + (1 if w.bit_idx else 0)
    words = np.ctypeslib.as_array(w.buffer, shape=(max(nw, 1),)).copy()[:nw]
    # the real entry point, for the size triple
    w2 = _BitWriter()
    L.huffman_compress(_p(src), n, C.byref(w2))
    assert (w2.word_idx, w2.bit_idx) == (w.word_idx, w.bit_idx)
    pre = []

    def walk(nd):
        leaf = not bool(nd.contents.left)
        pre.append((1 if leaf else 0, int(nd.contents.value), int(nd.contents.frequency)))
        if not leaf:
            walk(nd.contents.left)
            walk(nd.contents.right)

    walk(root)
    return dict(words=words, bits=int(w.word_idx) * 32 + int(w.bit_idx), word_idx=int(w2.word_idx),
                bit_idx=int(w2.bit_idx), buffer_size=int(w2.buffer_size), codes=codes, lens=lens, preorder=pre)


# ----------------------------------------------------------------------------- deflate
class _Buckets(C.Structure):                        # deflate/lz77.h:16-20
    _fields_ = [("patterns", C.POINTER(C.c_uint32)), ("indices", C.POINTER(C.c_uint64)),
                ("is_set", C.POINTER(C.c_bool))]


class _HashTableArray(C.Structure):                 # deflate/lz77.h:22-28
    _fields_ = [("buckets", _Buckets), ("bucket_indices", C.c_uint32 * 32768),
                ("current_idx", C.c_uint32), ("is_full", C.c_bool)]


class RefDeflate:
    """deflate/lz77.c per-block tokeniser with a table allocated ONCE (re-init after free hangs:
    SURVEY.md A.3.2) and one persistent 65536+64 byte input buffer (what fread reuses)."""
    T = 1 << 20

    def __init__(self):
        self.L = C.CDLL(os.path.join(_REF, "libdeflate.so"))
        self.L.init_hash_table.argtypes = [C.POINTER(_HashTableArray)]
        self.L.lz77_compress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(_HashTableArray)]
        self.t = _HashTableArray()
        self.L.init_hash_table(C.byref(self.t))
        self.buf = np.zeros(65536 + TAIL, dtype=np.uint8)
        self.fresh()

    def fresh(self):
        C.memset(self.t.buckets.patterns, 0, 4 * self.T)
        C.memset(self.t.buckets.indices, 0, 8 * self.T)
        C.memset(self.t.buckets.is_set, 0, self.T)
        C.memset(self.t.bucket_indices, 0, 4 * 32768)
        self.t.current_idx = 0
        self.t.is_full = False
        self.buf[:] = 0

    def block(self, data):
        n = len(data)
        assert n <= 65536
        self.buf[:n] = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = np.zeros(2 * n + 8, dtype=np.uint8)
        m = C.c_uint64(0)
        self.L.lz77_compress(_p(self.buf), n, _p(out), C.byref(m), C.byref(self.t))
        return out[: m.value].copy()

    def stream(self, data, independent):
        arr = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        self.fresh()
        outs, sizes = [], []
        for at in range(0, len(arr), 65536):
            if independent:
                self.fresh()
            o = self.block(arr[at: at + 65536])
            outs.append(o)
            sizes.append(len(o))
        return (np.concatenate(outs) if outs else np.zeros(0, np.uint8)), np.array(sizes, dtype=np.uint64)
