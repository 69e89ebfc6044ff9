"""GPU parity of the match-finder stage alone: find() at every position (parse-independent,
SURVEY.md section 0) from the HIP cluster replay against the oracle's literal table."""
import numpy as np
import pytest
import torch

from compression_algorithms_amd import synth

pytestmark = pytest.mark.gpu

CONFIGS = [("deflate", None), ("lz77", 14), ("lz77", 16)]


def _oracle_find(data, p):
    from oracle import orc
    out = np.empty(len(data), dtype=np.uint32)
    for at in range(0, len(data), p.block):
        out[at:at + p.block] = orc.find_all(data[at:at + p.block], p.wbits, p.tbits, bool(p.deflate))
    return np.where(out == 0xFFFFFFFF, 0xFFFF, out).astype(np.uint16)


def _check(data, flavour, wbits, block=65536):
    from compression_algorithms_amd import lz
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    p = lz.params(flavour, wbits, block)
    got = lz.find_all(data, p).cpu().numpy().view(np.uint16)
    want = _oracle_find(data, p)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{flavour} w{wbits}: {bad.size} mismatches, first at {bad[:5]}: got {got[bad[:5]]} want {want[bad[:5]]}"


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
def test_enwik_like(flavour, wbits):
    _check(synth.enwik_like(400_000, seed=7).numpy(), flavour, wbits)


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
def test_seed_with_table_end_overflow(flavour, wbits):
    # seed 12345 has a frequent word whose home is 5 buckets before the end of a 2^20 table: the
    # lz77 insert probes past T-1 (unbounded bucket ids), the deflate insert wraps to bucket 0
    _check(synth.enwik_like(300_000, seed=12345).numpy(), flavour, wbits)


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
@pytest.mark.parametrize("kind,n", [("zeros", 65536), ("zeros", 70000), ("single", 40000), ("two", 65536),
                                    ("random", 65536), ("period3", 65536), ("period4", 65536), ("period16384", 49152),
                                    ("period32767", 65536), ("period32768", 65536), ("zero_tail", 1000),
                                    ("random", 5), ("random", 3), ("random", 1), ("skewed", 65536)])
def test_adversarial(flavour, wbits, kind, n):
    _check(synth.adversarial(kind, n), flavour, wbits)


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
def test_source_code_like(flavour, wbits):
    # indentation runs: clusters of many thousand entries (the giant-cluster kernel)
    rng = np.random.default_rng(5)
    lines = []
    words = [b"self", b"return", b"if", b"else", b"for", b"in", b"range", b"value", b"def", b"None", b"x", b"y"]
    for _ in range(6000):
        ind = b"    " * int(rng.integers(0, 6))
        lines.append(ind + b" ".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 7)))) + b"\n")
    _check(b"".join(lines)[:200_000], flavour, wbits)


@pytest.mark.parametrize("block", [4096, 10000, 65536])
def test_block_sizes(block):
    _check(synth.enwik_like(150_000, seed=9).numpy(), "deflate", None, block)


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
@pytest.mark.parametrize("run", [5000, 10000, 20000, 40000])
def test_zero_run_inside_random(flavour, wbits, run):
    """a long single-byte run inside random data: one giant cluster that foreign entries join (the general fallback
    replay, in LDS up to 18432 entries and through global scratch beyond); the pure-run closed form must NOT fire"""
    rng = np.random.default_rng(run)
    data = rng.integers(0, 256, 65536 + 3000, dtype=np.uint8)
    data[9000:9000 + run] = 0
    _check(data, flavour, wbits)


@pytest.mark.parametrize("flavour,wbits", [("deflate", None), ("lz77", 14)])
@pytest.mark.parametrize("shape", ["pages", "sparse_foreign", "several_runs", "crowded", "half"])
def test_dominated_giant_clusters(flavour, wbits, shape):
    """giant clusters that one word dominates take k_lz_emulate_dom (occupancy bits + the slots of live entries + a FIFO of
    live foreign entries, lz_find.hip); "crowded" has more live foreign entries than that FIFO holds and "half" is not
    dominated: both must fall through to the general replay.  find() at every position against the oracle's literal table."""
    rng = np.random.default_rng(len(shape) * 7 + (wbits or 15))
    n = 2 * 65536 + 12345
    if shape == "pages":
        data = synth.family("pages", 99, n)
    elif shape == "sparse_foreign":                 # zeros with a few hundred stray bytes: every one makes <= 4 foreign words
        data = np.zeros(n, np.uint8)
        at = rng.integers(0, n, 300)
        data[at] = rng.integers(1, 256, 300, dtype=np.uint8)
    elif shape == "several_runs":                   # runs of different byte values, 6 000 .. 30 000 long, separated by text
        data = synth.enwik_like(n, seed=77).numpy().copy()
        at = 1000
        for ln in (6000, 30000, 9000, 14000, 22000):
            data[at:at + ln] = int(rng.integers(0, 256))
            at += ln + int(rng.integers(500, 4000))
    elif shape == "crowded":                        # 85 % zeros, 15 % random bytes in short bursts: thousands of live foreign entries
        data = np.zeros(n, np.uint8)
        for at in rng.integers(0, n - 8, n // 40):
            data[at:at + 6] = rng.integers(1, 256, 6, dtype=np.uint8)
    else:                                           # half zeros, half random: a giant cluster nobody dominates
        data = rng.integers(0, 256, n, dtype=np.uint8)
        data[::2] = 0
        data[20000:52000] = 0
    _check(data, flavour, wbits)


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
def test_pure_runs_closed_form(flavour, wbits):
    """blocks that are one byte value throughout (closed form: anchors) next to a block that is not"""
    data = np.concatenate([np.zeros(65536, np.uint8), np.full(65536, 0x41, np.uint8),
                           synth.enwik_like(65536, seed=3).numpy(), np.full(30000, 0xFF, np.uint8)])
    _check(data, flavour, wbits)


def test_block_scan_composes_earlier_maps_first():
    """The partition's overflow certificate is a prefix scan of non-commutative maps x -> max(x + a, b): the device scan must
    equal the serial left-to-right composition (round 1's wave-level step composed them backwards and under-estimated
    the carry behind a long run — parity held only because such cuts rarely change find()'s answer)."""
    import ctypes as C
    from compression_algorithms_amd.context import default_context
    ctx = default_context()
    rng = np.random.default_rng(5)
    NEG = -(1 << 28)
    for trial in range(6):
        c = rng.integers(0, 9, 1024)
        c[rng.integers(0, 1024)] += int(rng.integers(800, 3000))            # one long run somewhere
        ab = np.stack([c - 64, np.where(c > 0, c - 1, 0)], axis=1).astype(np.int32)
        out = np.zeros(1025, np.uint64)
        ctx.L.mi_selftest_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert ctx.L.mi_selftest_scan(ctx.h, ab.ctypes.data, out.ctypes.data) == 1
        a_run, b_run = 0, NEG
        for t in range(1024):
            got_a, got_b = np.int32(np.uint32(int(out[t]) >> 32)), np.int32(np.uint32(int(out[t]) & 0xFFFFFFFF))
            assert (int(got_a), int(got_b)) == (a_run, b_run), (trial, t)
            a2, b2 = int(ab[t, 0]), int(ab[t, 1])
            a_run, b_run = max(a_run + a2, NEG), max(max(b_run + a2, NEG), b2)


@pytest.mark.parametrize("flavour,wbits", CONFIGS)
@pytest.mark.parametrize("shape", ["runs_1k_4k", "runs_with_strays", "two_runs_one_home_range"])
def test_wide_clusters_take_the_dominated_replay(flavour, wbits, shape):
    """round 4: exported clusters of 1 025 .. 4 096 entries (what the wide finder hands over: a run of one byte value with foreign
    words inside its bucket range) that one word dominates are replayed by k_lz2_dom (lz_dom.h) — flagged in their descriptor —
    and what it leaves (not dominated, given up, covering bucket 0 / T) by the general wave replay with one or two bitmap
    registers.  find() at every position against the oracle's literal table; the counters must show the path was taken."""
    from compression_algorithms_amd import lz
    from compression_algorithms_amd.context import Context
    rng = np.random.default_rng(len(shape) * 13 + (wbits or 15))
    n = 3 * 65536 + 777
    data = synth.enwik_like(n, seed=123).numpy().copy()
    at = 500
    while at + 5000 < n:
        ln = int(rng.integers(1100, 3900))
        b = int(rng.integers(0, 256))
        data[at:at + ln] = b
        if shape == "runs_with_strays":                 # foreign words inside the run's bucket range AND inside the run itself
            for q in rng.integers(at, at + ln, 12):
                data[q] = (b + 1 + int(rng.integers(0, 200))) & 0xFF
        if shape == "two_runs_one_home_range":          # the same byte again a little later: one word, two bursts, retirements between
            data[at + ln + 700: at + ln + 700 + ln // 2] = b
            at += ln // 2 + 700
        at += ln + int(rng.integers(3000, 9000))
    _check(data, flavour, wbits)
    ctx = Context(0)
    p = lz.params(flavour, wbits)
    lz.compress(data, p, ctx).nbytes
    st = ctx.path_stats()
    # parts above 2 560 entries existed: the wide finder and its clusters ran (two bursts of one byte can exceed 4 096 entries
    # together: such a block takes the fallback pipeline instead, whose dominated replay is the same code)
    assert st["wide_parts"] + st["fallback_blocks"] > 0, st
    if shape != "two_runs_one_home_range":
        assert st["wide_parts"] > 0, st


def test_path_counters_are_zero_on_text_and_count_the_fallback():
    """mi_lz_path_stats: text takes neither the fallback pipeline nor wide parts; a block of one byte value is a
    giant cluster and takes the fallback, and is counted"""
    from compression_algorithms_amd import lz
    from compression_algorithms_amd.context import Context
    ctx = Context(0)
    p = lz.params("deflate")
    lz.compress_h(synth.enwik_like(40 * 65536, seed=5), p, ctx).nbytes
    assert ctx.path_stats() == {"fallback_blocks": 0, "wide_parts": 0}
    data = np.concatenate([np.zeros(3 * 65536, np.uint8), synth.enwik_like(65536, seed=6).numpy()])
    lz.compress_h(data, p, ctx).nbytes
    assert ctx.path_stats()["fallback_blocks"] == 3
