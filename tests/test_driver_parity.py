"""Driver parity: the reference's own drivers — algorithms/lz77/main.c, huffman/main.c, deflate/main.c —
compiled UNCHANGED against this repo's drop-in headers and linked to the drop-in libraries.

CPU part (needs /root/reference, i.e. the build container): they compile and link (oracle/Makefile `drivers`).
GPU part (needs the binaries that build() left under oracle/_ref/drivers/, which travel to the GPU box; the reference's
SOURCES do not): they run on corpora prepared by scripts/prep_data.py (get_data.sh:6-8's `head -c` derivations, from
synthetic enwik-shaped data) and print the reference's report — lz77/main.c:54-62, huffman/main.c:89-97,
deflate/deflate.c:65 + deflate/main.c:13 — with SUCCESS from the reference's own round-trip check."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "oracle", "_ref", "drivers")
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/algorithms"), reason="the reference sources exist only in the build container")
def test_reference_drivers_compile_and_link_unchanged():
    from compression_algorithms_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    for b in ("lz77_main", "lz77_w16_main", "huffman_main", "deflate_main"):
        p = os.path.join(DRV, b)
        if os.path.exists(p):
            os.remove(p)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "drivers"])
    for b in ("lz77_main", "lz77_w16_main", "huffman_main", "deflate_main"):
        assert os.access(os.path.join(DRV, b), os.X_OK), b
    # the unresolved symbols of each driver are exactly reference names its drop-in library exports
    want = {"lz77_main": {"read_input_buffer", "min", "lz77_compress", "lz77_decompress", "check_buffer_equivalence"},
            "huffman_main": {"read_input_buffer", "huffman_compress", "huffman_decompress"},
            "deflate_main": {"compress"}}
    for b, syms in want.items():
        nm = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(DRV, b)], text=True)
        got = {l.split()[-1].split("@")[0] for l in nm.splitlines() if l.strip()}
        assert syms <= got, (b, syms - got)


def test_prep_data_derivations(tmp_path):
    """get_data.sh:6-8: enwik8/7/6 are prefixes of enwik9 of 10^8 / 10^7 / 10^6 bytes (here from a short stand-in)"""
    import prep_data
    src = tmp_path / "src9"
    src.write_bytes(bytes(range(256)) * 5000)                  # 1.28 MB "enwik9"
    kind, made = prep_data.prepare(str(tmp_path / "data"), source=str(src))
    assert made == {"enwik9": 1_280_000, "enwik8": 1_280_000, "enwik7": 1_280_000, "enwik6": 1_000_000}
    assert (tmp_path / "data" / "enwik6").read_bytes() == src.read_bytes()[:1_000_000]


def _tree(tmp_path, size):
    import prep_data
    kind, made = prep_data.prepare(str(tmp_path / "data"), size=size, device="cuda:0")
    for d in ("lz77", "huffman", "deflate"):
        os.makedirs(tmp_path / "algorithms" / d, exist_ok=True)
    return made


def _field(out, name):
    m = re.search(re.escape(name) + r"\s*([0-9.]+)", out)
    assert m, (name, out[-600:])
    return float(m.group(1))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(DRV, "lz77_main")), reason="oracle/_ref/drivers not built (run build() in the build container)")
def test_reference_drivers_run_on_the_drop_ins(tmp_path):
    made = _tree(tmp_path, 150_000_000)                        # "enwik9" = 150 MB synthetic; enwik8 = its first 10^8 bytes
    assert made["enwik8"] == 100_000_000
    env = dict(os.environ)
    # ---- lz77/main.c (reads ../../data/enwik8), shipped window and the 64 KiB-window build
    for exe, wb in (("lz77_main", "14"), ("lz77_w16_main", "16")):
        env["MI_LZ77_WINDOW_BITS"] = wb
        r = subprocess.run([os.path.join(DRV, exe)], cwd=tmp_path / "algorithms" / "lz77", env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-400:]
        assert "SUCCESS" in r.stdout and "Number of differences: 0" in r.stdout
        assert _field(r.stdout, "Uncompressed size:") == 100_000_000 and _field(r.stdout, "Reconstructed size:") == 100_000_000
        assert 1.2 < _field(r.stdout, "Compression ratio:") < 4.0
        assert _field(r.stdout, "Compression MB/s:") > 0
    # ---- huffman/main.c (reads ../../data/enwik9; ends with the reference's own exit(1), main.c:98)
    r = subprocess.run([os.path.join(DRV, "huffman_main")], cwd=tmp_path / "algorithms" / "huffman", env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, r.stderr[-400:]
    assert "SUCCESS" in r.stdout and "Number of mismatches: 0" in r.stdout
    assert _field(r.stdout, "Uncompressed size:") == 150_000_000 and _field(r.stdout, "Reconstructed size:") == 150_000_000
    assert 1.3 < _field(r.stdout, "Compression ratio:") < 2.5
    # ---- deflate/main.c: compress("../../data/enwik8") writes enwik8.deflate in the cwd (deflate.c:19-22)
    for mode in ("T", "H"):
        env["MI_DEFLATE_MODE"] = mode
        d = tmp_path / "algorithms" / "deflate"
        r = subprocess.run([os.path.join(DRV, "deflate_main")], cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-400:]
        assert "MB/s:" in r.stdout and "Compression took" in r.stdout
        size = os.path.getsize(d / "enwik8.deflate")
        assert (size > 100_000_000) if mode == "T" else (size < 60_000_000)     # raw tokens expand text; mode H halves it
