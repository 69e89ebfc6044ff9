"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes bindings for oracle/liborc.so, the CPU restatement of the reference's
algorithms (orc_lz.c, orc_huff.c, orc_fse.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module, and only as the checker.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
TAIL = 64
NONE32 = 0xFFFFFFFF
COVERED32 = 0xFFFFFFFE


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_lz.c", "orc_huff.c", "orc_fse.c", "orc_defh.c", "orc_table.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.orc_lz77_encode.restype = C.c_uint64
        L.orc_lz77_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_lz77_decode.restype = C.c_uint64
        L.orc_lz77_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
        L.orc_deflate_new.restype = C.c_void_p
        L.orc_deflate_new.argtypes = [C.c_uint32]
        L.orc_deflate_free.argtypes = [C.c_void_p]
        L.orc_deflate_reset_table.argtypes = [C.c_void_p]
        L.orc_deflate_zero_buffer.argtypes = [C.c_void_p]
        L.orc_deflate_block.restype = C.c_uint64
        L.orc_deflate_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_deflate_stream.restype = C.c_uint64
        L.orc_deflate_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_deflate_block_decode.restype = C.c_uint64
        L.orc_deflate_block_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_find_all.restype = None
        L.orc_find_all.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_home_of.restype = C.c_uint32
        L.orc_home_of.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_max_bucket.restype = C.c_uint64
        L.orc_max_bucket.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_probe_stats.restype = None
        L.orc_probe_stats.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_huff_histogram.restype = None
        L.orc_huff_histogram.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_huff_encode.restype = C.c_uint64
        L.orc_huff_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_huff_buffer_size.restype = C.c_uint64
        L.orc_huff_buffer_size.argtypes = [C.c_uint64]
        L.orc_huff_decode.restype = C.c_uint64
        L.orc_huff_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_huff_codes_from_freq.restype = C.c_int
        L.orc_huff_codes_from_freq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_huff_pack.restype = C.c_uint64
        L.orc_huff_pack.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_huff_preorder.restype = C.c_int
        L.orc_huff_preorder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_fse_normalise.restype = C.c_int
        L.orc_fse_normalise.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_fse_block_bound.restype = C.c_uint64
        L.orc_fse_block_bound.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_fse_encode_block.restype = C.c_uint64
        L.orc_fse_encode_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_fse_decode_block.restype = C.c_int
        L.orc_fse_decode_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64]
        L.orc_fse_ideal_bits.restype = C.c_double
        L.orc_fse_ideal_bits.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.orc_defh_encode_block.restype = C.c_uint64
        L.orc_defh_encode_block.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_defh_decode_block.restype = C.c_uint64
        L.orc_defh_decode_block.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_defh_lengths.restype = None
        L.orc_defh_lengths.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _padded(data, tail=TAIL):
    a = np.zeros(len(data) + tail, dtype=np.uint8)
    a[: len(data)] = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- lz77 (bit-packed)
def lz77_encode(data, wbits=14, lbits=4, tbits=None, trace=False):
    """-> (stream bytes of length nbits//8+1 with zero pad bits, nbits[, cand trace])"""
    tbits = wbits + 6 if tbits is None else tbits
    n = len(data)
    src = _padded(data)
    out = np.zeros(2 * n + 8, dtype=np.uint8)
    tr = np.zeros(max(n, 1), dtype=np.uint32) if trace else None
    nbits = lib().orc_lz77_encode(_p(src), n, wbits, lbits, tbits, _p(out), _p(tr) if trace else None)
    stream = out[: nbits // 8 + 1].copy()
    return (stream, nbits, tr[:n]) if trace else (stream, nbits)


def lz77_old_encode(data, wbits=14, lbits=4):
    """the reference's brute-force parser (lz77_compress_old, lz77.c:185-262) -> (stream bytes, nbits); O(n * 2^wbits)"""
    n = len(data)
    src = _padded(data)
    out = np.zeros(2 * n + 8, dtype=np.uint8)
    L = lib()
    L.orc_lz77_old_encode.restype = C.c_uint64
    L.orc_lz77_old_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    nbits = L.orc_lz77_old_encode(_p(src), n, wbits, lbits, _p(out))
    return out[: nbits // 8 + 1].copy(), nbits


def lz77_decode(stream, nbits, n, wbits=14, lbits=4):
    s = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    s = np.concatenate([s, np.zeros(8, np.uint8)])
    out = np.zeros(max(n, 1), dtype=np.uint8)
    got = lib().orc_lz77_decode(_p(s), nbits, wbits, lbits, _p(out), n)
    if got != n:
        raise ValueError(f"lz77 decode produced {got} of {n} bytes")
    return out[:n]


# ---------------------------------------------------------------- deflate byte tokens
class Deflate:
    """Per-block tokeniser with the reference's persistent table + reused buffer."""

    def __init__(self, block=65536):
        self.block = block
        self.h = lib().orc_deflate_new(block)
        if not self.h:
            raise MemoryError

    def close(self):
        if self.h:
            lib().orc_deflate_free(self.h)
            self.h = None

    __del__ = close

    def fresh(self):
        lib().orc_deflate_reset_table(self.h)
        lib().orc_deflate_zero_buffer(self.h)

    def block_encode(self, data, want_freq=False, trace=False):
        n = len(data)
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        out = np.zeros(2 * n + 8, dtype=np.uint8)
        fr = np.zeros(286, dtype=np.uint32) if want_freq else None
        tr = np.zeros(max(n, 1), dtype=np.uint32) if trace else None
        m = lib().orc_deflate_block(self.h, _p(src), n, _p(out), _p(fr) if want_freq else None, _p(tr) if trace else None)
        if m == 0xFFFFFFFFFFFFFFFF:
            raise ValueError("block too large")
        res = [out[:m].copy()]
        if want_freq:
            res.append(fr)
        if trace:
            res.append(tr[:n])
        return res[0] if len(res) == 1 else tuple(res)


def deflate_stream(data, block=65536, independent=True, want_ub=False):
    """-> (token bytes, per-block sizes u64[, reference-UB flag]).  independent=False is the shipped compress()."""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    n = len(src)
    nb = (n + block - 1) // block
    out = np.zeros(2 * n + 4 * nb + 8, dtype=np.uint8)
    sizes = np.zeros(max(nb, 1), dtype=np.uint64)
    mb = np.zeros(1, dtype=np.uint64)
    m = lib().orc_deflate_stream(_p(src), n, block, 1 if independent else 0, _p(out), _p(sizes), _p(mb))
    if want_ub:
        return out[:m].copy(), sizes[:nb], bool(int(mb[0]) >= (1 << 20))
    return out[:m].copy(), sizes[:nb]


def deflate_block_decode(tokens, n):
    t = np.ascontiguousarray(tokens, dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    got = lib().orc_deflate_block_decode(_p(t), len(t), _p(out), n)
    if got != n:
        raise ValueError(f"deflate block decode produced {got} of {n}")
    return out[:n]


def find_all(data, wbits, tbits, deflate):
    n = len(data)
    src = _padded(data)
    cand = np.zeros(max(n, 1), dtype=np.uint32)
    lib().orc_find_all(_p(src), n, wbits, tbits, 1 if deflate else 0, _p(cand))
    return cand[:n]


def home(word, tbits):
    return int(lib().orc_home_of(word, tbits))


def past_table_end(data, wbits, tbits, deflate):
    """True when the reference would touch memory past its bucket array on this input (UB there)."""
    src = _padded(data)
    return int(lib().orc_max_bucket(_p(src), len(data), wbits, tbits, 1 if deflate else 0)) >= (1 << tbits)


def probe_stats(data, wbits, tbits, deflate):
    src = _padded(data)
    st = np.zeros(4, dtype=np.uint64)
    lib().orc_probe_stats(_p(src), len(data), wbits, tbits, 1 if deflate else 0, _p(st))
    return dict(inserts=int(st[0]), insert_probes=int(st[1]), finds=int(st[2]), find_probes=int(st[3]))


# ---------------------------------------------------------------- huffman
def huff_histogram(data):
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    f = np.zeros(256, dtype=np.uint32)
    lib().orc_huff_histogram(_p(src), len(src), _p(f))
    return f


def huff_encode(data):
    """-> dict(words u32[], bits, word_idx, bit_idx, buffer_size, codes u32[256], lens u8[256]) or None
    where the reference would exit(1) / overflow."""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    n = len(src)
    nwords = n + 2  # codes are <= 32 bits: n words always suffice
    words = np.zeros(nwords, dtype=np.uint32)
    codes = np.zeros(256, dtype=np.uint32)
    lens = np.zeros(256, dtype=np.uint8)
    bits = lib().orc_huff_encode(_p(src), n, _p(words), nwords, _p(codes), _p(lens))
    if bits == 0xFFFFFFFFFFFFFFFF:
        return None
    nw = (bits + 31) // 32
    return dict(words=words[:nw].copy(), bits=int(bits), word_idx=int(bits // 32), bit_idx=int(bits % 32),
                buffer_size=int(lib().orc_huff_buffer_size(bits)), codes=codes, lens=lens)


def huff_codes_from_freq(freq):
    """codes/lengths the reference derives from a histogram (u32, wrapping) -> (codes u32[256], lens u8[256]) or None"""
    f = np.ascontiguousarray(np.asarray(freq, dtype=np.uint64) & 0xFFFFFFFF, dtype=np.uint32)
    codes = np.zeros(256, dtype=np.uint32)
    lens = np.zeros(256, dtype=np.uint8)
    if lib().orc_huff_codes_from_freq(_p(f), _p(codes), _p(lens)) < 0:
        return None
    return codes, lens


def huff_pack(data, codes, lens, bit_offset=0):
    """pack `data` with given codes, the stream starting bit_offset bits into word 0 -> (words u32[], end bit)"""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    bits = int(np.asarray(lens, np.uint64)[src].sum())
    words = np.zeros((bit_offset + bits + 31) // 32 + 1, dtype=np.uint32)
    end = lib().orc_huff_pack(_p(src), len(src), _p(np.ascontiguousarray(codes, np.uint32)),
                              _p(np.ascontiguousarray(lens, np.uint8)), bit_offset, _p(words))
    return words[: (end + 31) // 32].copy(), int(end)


def huff_decode(words, bits, freq, n):
    w = np.concatenate([np.ascontiguousarray(words, dtype=np.uint32), np.zeros(2, np.uint32)])
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    got = lib().orc_huff_decode(_p(w), bits, _p(f), _p(out), n)
    if got != n:
        raise ValueError("huffman decode failed")
    return out[:n]


def huff_preorder(freq):
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    kinds = np.zeros(511, np.uint8); vals = np.zeros(511, np.uint8); fr = np.zeros(511, np.uint32)
    k = lib().orc_huff_preorder(_p(f), _p(kinds), _p(vals), _p(fr))
    return kinds[:k], vals[:k], fr[:k]


# ---------------------------------------------------------------- fse
def fse_normalise(freq, L=8):
    f = np.ascontiguousarray(freq, dtype=np.uint64)
    cnt = np.zeros(256, dtype=np.uint32)
    rc = lib().orc_fse_normalise(_p(f), L, _p(cnt))
    if rc < 0:
        raise ValueError("more symbols than table slots")
    return cnt


def fse_encode_block(data, L=8, S=64, spread=1):
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    n = len(src)
    out = np.zeros(lib().orc_fse_block_bound(n, L, S), dtype=np.uint8)
    m = lib().orc_fse_encode_block(_p(src), n, L, S, spread, _p(out))
    if m == 0xFFFFFFFFFFFFFFFF:
        raise ValueError("fse encode failed")
    return out[:m].copy()


def fse_decode_block(rec, n, L=8, S=64, spread=1):
    r = np.ascontiguousarray(rec, dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    rc = lib().orc_fse_decode_block(_p(r), len(r), L, S, spread, _p(out), n)
    if rc:
        raise ValueError(f"fse decode failed rc={rc}")
    return out[:n]


def fse_ideal_bits(data, L=8):
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    return float(lib().orc_fse_ideal_bits(_p(src), len(src), L))


# ---------------------------------------------------------------- mode H (tokens + per-block dynamic Huffman)
def defh_encode_block(tokens):
    """reference byte tokens of one block -> mode-H record (oracle/orc_defh.c)"""
    t = np.ascontiguousarray(tokens, dtype=np.uint8)
    out = np.zeros(4 + 288 + 4 * len(t) + 16, dtype=np.uint8)
    m = lib().orc_defh_encode_block(_p(t), len(t), _p(out))
    return out[:m].copy()


def defh_decode_block(rec, cap):
    r = np.ascontiguousarray(rec, dtype=np.uint8)
    out = np.zeros(cap, dtype=np.uint8)
    m = lib().orc_defh_decode_block(_p(r), len(r), _p(out), cap)
    if m == 0xFFFFFFFFFFFFFFFF:
        raise ValueError("malformed mode-H record")
    return out[:m].copy()


def defh_lengths(freq):
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    ln = np.zeros(286, dtype=np.uint8)
    lib().orc_defh_lengths(_p(f), _p(ln))
    return ln
