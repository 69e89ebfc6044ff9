"""The reference's first, brute-force parser — lz77_compress_old, algorithms/lz77/lz77.h:51-54, lz77.c:185-262.
CPU part: the oracle's restatement (orc_lz77_old_encode) against tests/golden/lz77_old.json, which oracle/gen_golden_old.py
produced by running the compiled reference.  GPU part: lz_old.hip through the C ABI against the golden vectors, against the
oracle on further seeds, the round trip, and the drop-in's lz77_compress_old called the way lz77/main.c:26 would call it."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from compression_algorithms_amd import _lib, synth
from oracle import orc
from oracle.gen_golden_old import CASES, make_input


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "lz77_old.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name,kind,n,seed,wbits", CASES)
def test_oracle_is_the_reference(golden, name, kind, n, seed, wbits):
    g = golden[name]
    data = make_input(kind, n, seed)
    assert _sha(data) == g["input_sha256"]
    s, nb = orc.lz77_old_encode(data, wbits, 4)
    assert nb == g["bits"] and _sha(s) == g["sha256"]


def test_first_window_is_all_literals():
    """for the first 2^14 - 1 positions `buffer_index - window_size` wraps (lz77.c:208): 9 bits per byte, whatever the data"""
    data = np.zeros(16383, np.uint8)
    s, nb = orc.lz77_old_encode(data, 14, 4)
    assert nb == 9 * 16383
    s, nb = orc.lz77_old_encode(np.zeros(16384, np.uint8), 14, 4)     # position 16383 sees the window: a match of 1 byte (the end cuts it)
    assert nb == 9 * 16383 + 19


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,n,seed,wbits", CASES)
def test_hip_stream_is_the_reference_stream(golden, name, kind, n, seed, wbits):
    from compression_algorithms_amd import lz
    g = golden[name]
    data = make_input(kind, n, seed)
    st = lz.compress_old(data, wbits, 4)
    assert st.total_bits == g["bits"]
    got = np.frombuffer(st.tobytes(), dtype=np.uint8)
    assert _sha(got) == g["sha256"]
    if n:
        assert np.array_equal(lz.decompress_whole(st).cpu().numpy(), data)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,wbits,lbits", [(71, 70_001, 14, 4), (72, 33_333, 12, 4), (73, 20_000, 10, 5), (74, 150_000, 16, 4), (75, 40_000, 14, 3)])
def test_hip_against_the_oracle(seed, n, wbits, lbits):
    from compression_algorithms_amd import lz
    data = synth.enwik_like(n, seed=seed).numpy()
    want, nb = orc.lz77_old_encode(data, wbits, lbits)
    st = lz.compress_old(data, wbits, lbits)
    assert st.total_bits == nb
    assert np.array_equal(np.frombuffer(st.tobytes(), dtype=np.uint8), want)
    assert np.array_equal(lz.decompress_whole(st).cpu().numpy(), data)
    assert np.array_equal(orc.lz77_decode(want, nb, n, wbits, lbits), data)


class BitStream(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("bit_index", C.c_uint64)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 50_000, 100_000])
def test_dropin_like_reference_main(golden, n):
    """BitStream* compressed_stream = lz77_compress_old(buffer, filesize);   (lz77/main.c:26, commented out there)"""
    L = C.CDLL(os.path.join(_lib.LIB_DIR, "libmi_lz77.so"))
    L.lz77_compress_old.restype = C.POINTER(BitStream)
    L.lz77_compress_old.argtypes = [C.c_void_p, C.c_uint64]
    L.lz77_decompress.restype = C.c_void_p
    L.lz77_decompress.argtypes = [C.POINTER(BitStream), C.c_uint64, C.POINTER(C.c_uint64)]
    L.check_buffer_equivalence.restype = C.c_bool
    L.check_buffer_equivalence.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    data = synth.enwik_like(100_000, seed=61).numpy()[:n].copy()
    bs = L.lz77_compress_old(data.ctypes.data_as(C.c_void_p), n)
    nbits = int(bs.contents.bit_index)
    got = np.ctypeslib.as_array(bs.contents.data, shape=(nbits // 8 + 1,)).copy()
    want, nb = orc.lz77_old_encode(data, 14, 4)
    assert nbits == nb and np.array_equal(got, want)
    dsz = C.c_uint64(0)
    out = L.lz77_decompress(bs, n, C.byref(dsz))                         # lz77/main.c:33-37
    assert dsz.value == n
    assert L.check_buffer_equivalence(data.ctypes.data_as(C.c_void_p), out, n)
    L.mi_lz77_release(bs)
