"""Untrusted-input checks that run on the host, before any kernel: the block-table validator of
the C ABI and the side-car checks of the deflate drop-in's decompress().  No GPU needed: the checks come first."""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from compression_algorithms_amd import _lib

OK, CORRUPT, ARG = 0, 8, 1


def _v(table, stream_bytes, align):
    t = np.asarray(table, dtype=np.uint64)
    return _lib.lib().mi_validate_block_table(t.ctypes.data, len(t) - 1, stream_bytes, align)


def test_block_table_validator():
    assert _v([0, 16, 48, 64], 8, 8) == OK
    assert _v([0, 16, 48, 64], 7, 8) == CORRUPT          # last entry past the stream
    assert _v([0, 48, 16, 64], 8, 8) == CORRUPT          # not monotonic
    assert _v([0, 12, 48, 64], 8, 8) == CORRUPT          # token streams are byte aligned
    assert _v([0, 12, 48, 64], 8, 1) == OK               # the bit-packed lz77 flavour is not
    assert _v([0, 32, 96], 12, 32) == OK
    assert _v([0, 40, 96], 12, 32) == CORRUPT            # mode-H / FSE records are whole words
    assert _v([0], 0, 8) == OK                           # no blocks
    assert _v([2 ** 63, 2 ** 63 + 8], 2 ** 62, 8) == ARG  # 8 * stream_bytes would overflow
    assert _lib.lib().mi_validate_block_table(None, 0, 0, 8) == ARG


def _run_decompress(tmp_path, idx_bytes, stream=b"\x00" * 64):
    (tmp_path / "x.deflate").write_bytes(stream)
    (tmp_path / "x.deflate.idx").write_bytes(idx_bytes)
    lib = os.path.join(_lib.LIB_DIR, "libmi_deflate.so")
    code = ("import ctypes as C; L = C.CDLL(%r); L.decompress.argtypes = [C.c_void_p, C.c_char_p]; "
            "L.decompress(None, b'x.deflate')" % lib)
    return subprocess.run([sys.executable, "-c", code], cwd=tmp_path, capture_output=True, text=True, timeout=120)


@pytest.mark.parametrize("case", ["short", "truncated_table", "nblocks_lie", "block_zero", "offsets_past_stream", "not_monotonic"])
def test_deflate_sidecar_is_checked_before_use(tmp_path, case):
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    n, block, nblocks = 100, 65536, 1
    good = struct.pack("<3Q", n, block, nblocks) + struct.pack("<2Q", 0, 64 * 8)
    idx = {
        "short": good[:16],
        "truncated_table": good[:-8],
        "nblocks_lie": struct.pack("<3Q", n, block, 1 << 40) + struct.pack("<2Q", 0, 8),
        "block_zero": struct.pack("<3Q", n, 0, nblocks) + struct.pack("<2Q", 0, 8),
        "offsets_past_stream": struct.pack("<3Q", n, block, nblocks) + struct.pack("<2Q", 0, 65 * 8),
        "not_monotonic": struct.pack("<3Q", n, block, nblocks) + struct.pack("<2Q", 64, 8),
    }[case]
    r = _run_decompress(tmp_path, idx)
    assert r.returncode == 1, (case, r.returncode, r.stderr)
    assert "decompress:" in r.stderr and not os.path.exists(tmp_path / "x.deflate.orig")
