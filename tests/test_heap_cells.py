"""csrc/heap_cells.h on the host: the look-ahead sifts of the two tree builders (k_huff_build, k_defh_lengths) leave the SAME heap
array after every operation as the plain restatement of the reference heap (algorithms/huffman/huffman.c:100-163: strict '<' in
both sifts, ties keep their places) — random and tie-heavy frequency sets, 1..286 symbols, both cell widths."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lookahead_sifts_equal_the_reference_heap(tmp_path):
    exe = tmp_path / "heap_cells_harness"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(HERE, "heap_cells_harness.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
