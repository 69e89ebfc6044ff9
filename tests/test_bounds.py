"""Output bounds of the block-parallel encoders against the oracle's worst cases (CPU only: the bound functions
are host code of the C ABI; no kernel runs).  ADVICE r1 (high): the bounds must account for p.block — every block
pays its own header in mode H, and every block's last token may be a match that covers one real byte and runs on
into the zero tail."""
import ctypes as C

import numpy as np
import pytest

from compression_algorithms_amd import _lib, lz
from oracle import orc

UNIT = np.array([0x41, 0, 0, 0, 0x61, 0x62, 0x63, 0x41], np.uint8)      # 7 literals, then a match into the tail


def _h_bound(n, p):
    return int(_lib.lib().mi_deflate_h_bound_bytes(n, C.byref(p)))


def test_deflate_tokens_block8_worst_case():
    data = np.tile(UNIT, 1000)
    stream, sizes = orc.deflate_stream(data, 8, True)
    assert len(stream) == 18 * 1000                                    # 2n + 2 per block
    assert len(stream) <= lz.bound_bytes(len(data), lz.params("deflate", None, 8))


@pytest.mark.parametrize("wbits", [14, 16])
def test_lz77_bits_block8_worst_case(wbits):
    data = np.tile(UNIT, 200)
    total = 0
    for at in range(0, len(data), 8):
        _, nb = orc.lz77_encode(data[at:at + 8].tobytes(), wbits, 4)
        assert nb == 7 * 9 + 1 + wbits + 4
        total += nb
    assert (total + 7) // 8 <= lz.bound_bytes(len(data), lz.params("lz77", wbits, 8))


@pytest.mark.parametrize("block", [8, 256, 1024, 2048, 65536])
def test_mode_h_bound_covers_random_and_crafted(block):
    rng = np.random.default_rng(1)
    for data in (rng.integers(0, 256, 65536, dtype=np.uint8), np.tile(UNIT, 2048)):
        d = orc.Deflate(block)
        total = 0
        for at in range(0, len(data), block):
            d.fresh()
            total += len(orc.defh_encode_block(d.block_encode(data[at:at + block])))
        assert total <= _h_bound(len(data), lz.params("deflate", None, block)), block


def test_bounds_of_empty_input():
    assert lz.bound_bytes(0, lz.params("deflate")) >= 8
    assert _h_bound(0, lz.params("deflate")) >= 0
