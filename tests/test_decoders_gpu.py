"""The block decoders of lz_decode.hip / defh.hip on the shapes their rewrite introduced (round 2): the LDS ring of every
size (16 / 32 / 64 KiB: window or block, whichever is smaller), ring wrap with the farthest distances the window allows,
output and stream pointers that are not 16-byte aligned, blocks that are not multiples of 16, token windows whose last unit
opens a match (its second half comes out of the next 64 units), long literal runs (one parallel store per run)."""
import ctypes as C

import numpy as np
import pytest
import torch

from compression_algorithms_amd import synth

pytestmark = pytest.mark.gpu


def _roundtrip(data, flavour, wbits, block, lbits=None, mode_h=False):
    from compression_algorithms_amd import lz
    p = lz.params(flavour, wbits, block)
    if lbits is not None:
        p.lbits = lbits
    st = lz.compress_h(data, p) if mode_h else lz.compress(data, p)
    back = lz.decompress_h(st) if mode_h else lz.decompress(st)
    assert np.array_equal(back.cpu().numpy(), data), (flavour, wbits, block)
    return st


@pytest.mark.parametrize("flavour,wbits,block", [
    ("deflate", None, 65536),      # ring 32 KiB (W), block wraps it once
    ("deflate", None, 20000),      # ring 32 KiB > block: never wraps, block not a multiple of 16
    ("deflate", None, 4096),       # ring 16 KiB class
    ("lz77", 14, 65536),           # ring 16 KiB = W, wraps three times
    ("lz77", 16, 65536),           # ring 64 KiB = block
    ("lz77", 12, 65536),           # W = 4 KiB inside a 16 KiB ring
    ("lz77", 15, 50000),           # ring 32 KiB
    ("lz77", 16, 262144),          # blocks above 64 KiB: ring 64 KiB = W, wraps three times
    ("lz77", 14, 131072),
])
def test_roundtrip_every_ring(flavour, wbits, block):
    data = synth.enwik_like(3 * block + 777, seed=block % 97).numpy()
    _roundtrip(data, flavour, wbits, block)


@pytest.mark.parametrize("mode_h", [False, True])
def test_farthest_distances_across_the_ring_seam(mode_h):
    """every match reaches back almost a whole window: sources sit right behind the cells the copy overwrites"""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, 32760, dtype=np.uint8)
    data = np.concatenate([base, base, base[:100]])            # the second copy matches the first at distance 32760 (< W - 1)
    _roundtrip(data, "deflate", None, 65536, mode_h=mode_h)
    base = rng.integers(0, 256, 16380, dtype=np.uint8)
    _roundtrip(np.concatenate([base, base, base, base]), "lz77", 14, 65536)
    base = rng.integers(0, 256, 65530, dtype=np.uint8)
    _roundtrip(np.concatenate([base, base, base, base[:5000]]), "lz77", 16, 262144)


def test_long_matches_and_short_periods():
    """length fields up to 255 (lbits 8): copies longer than one 64-lane step, periods shorter than the copy"""
    data = np.concatenate([np.zeros(5000, np.uint8), np.tile(np.arange(3, dtype=np.uint8), 3000),
                           np.tile(np.arange(70, dtype=np.uint8), 200), synth.enwik_like(30000, seed=5).numpy()])
    _roundtrip(data, "deflate", None, 65536, lbits=8)
    _roundtrip(data, "lz77", 14, 65536, lbits=8)


def test_literal_runs_and_matches_on_unit_63():
    """random bytes (literal runs of thousands of units: one store per 64) with matches planted so that the flag unit of a match
    falls on every position of the 64-unit window, including the last"""
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, 60000, dtype=np.uint8)
    phrase = rng.integers(0, 256, 12, dtype=np.uint8)
    for k in range(200):
        at = 300 + k * 257 + (k % 64)
        data[at:at + 12] = phrase
    _roundtrip(data, "deflate", None, 65536)
    _roundtrip(data, "deflate", None, 65536, mode_h=True)
    _roundtrip(data, "lz77", 16, 65536)


@pytest.mark.parametrize("flavour,wbits,mode_h", [("deflate", None, False), ("deflate", None, True), ("lz77", 14, False)])
@pytest.mark.parametrize("out_off,in_off", [(1, 0), (3, 4), (0, 4), (8, 8)])
def test_unaligned_output_and_stream_pointers(flavour, wbits, mode_h, out_off, in_off):
    """d_out + off is not 16-byte aligned (the ring is flushed with byte stores then); the stream may start at any 4-byte
    boundary for mode H and anywhere for the token formats"""
    from compression_algorithms_amd import lz, _lib
    from compression_algorithms_amd.context import default_context
    ctx = default_context()
    data = synth.enwik_like(150_001, seed=4).numpy()
    p = lz.params(flavour, wbits, 65536)
    st = lz.compress_h(data, p) if mode_h else lz.compress(data, p)
    nbytes = (int(st.block_bits[-1].item()) + 7) // 8
    if not mode_h and flavour == "lz77":
        in_off = in_off + 1                                   # the bit format may start at any byte
    shifted = torch.zeros(nbytes + in_off + 64, dtype=torch.uint8, device=ctx.device)
    shifted[in_off:in_off + nbytes] = st.data[:nbytes]
    out = torch.zeros(len(data) + out_off + 16, dtype=torch.uint8, device=ctx.device)
    fn = ctx.L.mi_deflate_h_decode_dev if mode_h else ctx.L.mi_lz_decode_dev
    rc = fn(ctx.h, C.byref(st.p), C.c_void_p(shifted.data_ptr() + in_off), nbytes, C.c_void_p(st.block_bits.data_ptr()),
            C.c_void_p(out.data_ptr() + out_off), len(data), ctx.stream_ptr())
    _lib.check(rc, "decode")
    got = out.cpu().numpy()
    assert np.array_equal(got[out_off:out_off + len(data)], data)
    assert not got[:out_off].any() and not got[out_off + len(data):].any()      # nothing written outside [out, out + n)


@pytest.mark.parametrize("ring", ["4096", "8192", "16384", "65536"])
@pytest.mark.parametrize("flavour,wbits,mode_h", [("deflate", None, False), ("deflate", None, True), ("lz77", 16, False)])
def test_far_matches_read_the_output_buffer(ring, flavour, wbits, mode_h, monkeypatch):
    """a ring smaller than the window: matches whose source has left the ring read the bytes the wave flushed earlier (past
    the L1); forced ring sizes, many blocks (the default picks the small ring only then), distances right at the ring edge"""
    from compression_algorithms_amd import lz
    monkeypatch.setenv("MI_LZ_DECODE_RING", ring)
    rng = np.random.default_rng(21)
    base = rng.integers(0, 256, 16384 - 40, dtype=np.uint8)
    blk = np.concatenate([base, base[:9000], rng.integers(0, 256, 3000, dtype=np.uint8), base[100:16000], base[5000:12000], base[4090:4100], base[8185:8200]])
    blk = np.concatenate([blk, synth.enwik_like(65536 - len(blk), seed=3).numpy()])[:65536]
    data = np.concatenate([blk, synth.enwik_like(3 * 65536 + 123, seed=9).numpy()])
    p = lz.params(flavour, wbits, 65536)
    st = lz.compress_h(data, p) if mode_h else lz.compress(data, p)
    back = lz.decompress_h(st) if mode_h else lz.decompress(st)
    assert np.array_equal(back.cpu().numpy(), data)
