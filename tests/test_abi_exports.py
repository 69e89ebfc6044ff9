"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/mi_codec.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

from compression_algorithms_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "mi_codec.h")).read()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.startswith("mi_lz_params_") and d not in (
        "mi_huffman_bound_words", "mi_lz_num_blocks", "mi_lz_bound_bytes", "mi_fse_params_default")}  # static inline
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    missing = [s for s in sorted(declared) if not hasattr(built, s)]
    assert not missing, f"libmi_codec.so does not export: {missing}"


def test_frame_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "mi_frame.h")).read()
    declared = set(re.findall(r"\b(mi_frame_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) == 7, declared
    missing = [s for s in sorted(declared) if not hasattr(built, s)]
    assert not missing, f"libmi_codec.so does not export: {missing}"


def test_no_device_is_an_error_not_a_fallback(built):
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert built.mi_ctx_create(C.byref(h), 0) == 9      # MI_ERR_NO_DEVICE
    assert b"no CPU fallback" in built.mi_status_str(9)
