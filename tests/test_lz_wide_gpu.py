"""The lz77 flavour on blocks ABOVE 64 KiB (lzw.hip): WINDOW_BITS 16 with a window that really slides (SURVEY.md 8d
config 2 "64 KiB ... 1 MiB").  Parity against the oracle's literal table (find() at every position),
against the oracle's encoder per block, against tests/golden/lz77_wide.json — vectors the REAL reference produced
(oracle/gen_golden_wide.py) — and against the committed 300 kB whole-buffer fixture of round 1 (one 1 MiB block = the
reference's whole-buffer stream); every stream is decoded back on the GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

from compression_algorithms_amd import synth

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_find(data, wbits, block):
    from compression_algorithms_amd import lz
    from oracle import orc
    p = lz.params("lz77", wbits, block)
    got = lz.find_all32(data, p).cpu().numpy().view(np.uint32)
    for at in range(0, len(data), block):
        want = orc.find_all(data[at:at + block], wbits, wbits + 6, False)
        g = got[at:at + block]
        bad = np.flatnonzero(g != want)
        assert bad.size == 0, f"block at {at}: {bad.size} mismatches, first {bad[:5]}: got {g[bad[:5]]} want {want[bad[:5]]}"


def _check_stream(data, wbits, block):
    from compression_algorithms_amd import lz
    from oracle import orc
    p = lz.params("lz77", wbits, block)
    st = lz.compress(data, p)
    bb = st.block_bits.cpu().numpy()
    raw = np.frombuffer(st.tobytes(), dtype=np.uint8)
    allbits = np.unpackbits(raw, bitorder="little")
    out = []
    for k, at in enumerate(range(0, len(data), block)):
        s, nb = orc.lz77_encode(data[at:at + block].tobytes(), wbits, 4)
        assert int(bb[k + 1] - bb[k]) == nb, (k, int(bb[k + 1] - bb[k]), nb)
        got = np.packbits(allbits[int(bb[k]):int(bb[k + 1])], bitorder="little")
        want = s[: (nb + 7) // 8].copy()
        assert np.array_equal(got, want), f"block {k}: stream differs at byte {np.flatnonzero(got != want)[:5]}"
        out.append((got, nb))
    assert np.array_equal(lz.decompress(st).cpu().numpy(), data)
    return out


@pytest.mark.parametrize("wbits,block", [(16, 131072), (16, 262144), (14, 131072)])
def test_find_all_equals_the_literal_table(wbits, block):
    _check_find(synth.enwik_like(300_000, seed=51).numpy(), wbits, block)


def test_window_really_slides():
    """sanity of the premise: with a 256 KiB block and W = 64 KiB some candidates of the 64 KiB-block encoding are gone and
    the stream differs from the concatenation of 64 KiB blocks"""
    from oracle import orc
    data = synth.enwik_like(262144, seed=52).numpy()
    whole = orc.find_all(data, 16, 22, False)
    first = orc.find_all(data[:65536], 16, 22, False)
    assert np.array_equal(whole[:65532], first[:65532])               # nothing retired in the first 64 KiB (the last 3 words see the zero tail)
    assert (whole[65536:] != 0xFFFFFFFF).any()
    _check_stream(data, 16, 262144)


@pytest.mark.parametrize("kind,n", [("zeros", 70000), ("two", 100_000), ("random", 140_000), ("period3", 90_000), ("period16384", 150_000),
                                    ("zero_tail", 66_000), ("skewed", 131072)])
def test_adversarial(kind, n):
    data = np.frombuffer(synth.adversarial(kind, n), dtype=np.uint8)
    _check_find(data, 16, 131072)
    _check_stream(data, 16, 131072)


def test_bucket_zero_cluster_and_short_last_block():
    """a word whose home is bucket 0 (the ring's zero-fill clears bucket 0 at insertion W-1) repeated across
    the window boundary, and a last block shorter than the others"""
    from oracle import orc
    rng = np.random.default_rng(9)
    # find a 4-byte word with home 0 at T = 2^22
    w = None
    for cand in range(1, 1 << 26):
        if orc.home(cand, 22) == 0:
            w = cand
            break
    assert w is not None
    word = np.frombuffer(int(w).to_bytes(4, "little"), dtype=np.uint8)
    data = rng.integers(1, 255, 200_000, dtype=np.uint8)
    for at in (100, 5000, 65000, 65530, 66000, 70000, 131000, 140000, 199000):
        data[at:at + 4] = word
    _check_find(data, 16, 131072)
    _check_stream(data, 16, 131072)


def test_golden_vectors_from_the_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "lz77_wide.json")))
    for name, e in g.items():
        data = synth.enwik_like(e["n"], seed=e["seed"]).numpy()
        assert _sha(data) == e["input_sha256"], name
        blocks = _check_stream(data, 16, e["block"])
        for k, (got, nb) in enumerate(blocks):
            ref = e["blocks"][k]
            if ref is None:
                continue                                              # the reference's own UB on this block: not pinned
            s = np.zeros(nb // 8 + 1, np.uint8)
            s[: len(got)] = got
            assert nb == ref["bits"] and _sha(s) == ref["sha256"], (name, k)


def test_whole_buffer_fixture_of_round_1(golden_dir):
    """tests/golden/enwik_like_300k: ONE 1 MiB block holds the whole 300 kB sample, so the HIP stream must be the reference's
    whole-buffer lz77_compress stream (lz77_w16_whole / lz77_w14_whole), which round 1 could only check against the oracle"""
    from compression_algorithms_amd import lz
    e = json.load(open(os.path.join(golden_dir, "enwik_like_300k.json")))
    sample = np.fromfile(os.path.join(golden_dir, "enwik_like_300k.bin"), dtype=np.uint8)
    for wb in (16, 14):
        st = lz.compress(sample, lz.params("lz77", wb, 1 << 20))
        nb = st.total_bits
        s = np.zeros(nb // 8 + 1, np.uint8)
        raw = np.frombuffer(st.tobytes(), dtype=np.uint8)
        s[: len(raw)] = raw
        assert nb == e[f"lz77_w{wb}_whole"]["bits"] and _sha(s) == e[f"lz77_w{wb}_whole"]["sha256"], wb
        assert np.array_equal(lz.decompress(st).cpu().numpy(), sample)


# ---- the time-sliced finder (lzs.hip) against the oracle on every input family, and against the whole-block finder
@pytest.mark.parametrize("kind", ["text", "lowent", "phrases", "runs", "pages"])
@pytest.mark.parametrize("wbits,block", [(16, 131072), (16, 262144), (14, 131072)])
def test_sliced_finder_families(kind, wbits, block):
    import test_fuzz_gpu as F
    seed = 7000 + block // 1000 + wbits
    data = F._family(kind, seed, int(2.3 * block))
    _check_find(data, wbits, block)


@pytest.mark.parametrize("wbits", [16, 14])
def test_sliced_equals_whole_block_finder(wbits, monkeypatch):
    """1 MiB blocks (16 / 64 steps): lzs.hip (sub-blocks of W positions, old entries pre-placed) and lzw.hip (one cluster
    structure for the whole block) must agree position by position; the first block also against the oracle"""
    from compression_algorithms_amd import lz
    from oracle import orc
    data = synth.enwik_like(1_500_000, seed=77).numpy()
    p = lz.params("lz77", wbits, 1 << 20)
    monkeypatch.setenv("MI_LZW_SLICED", "0")
    whole = lz.find_all32(data, p).cpu().numpy().view(np.uint32)
    monkeypatch.setenv("MI_LZW_SLICED", "1")
    sliced = lz.find_all32(data, p).cpu().numpy().view(np.uint32)
    bad = np.flatnonzero(whole != sliced)
    assert bad.size == 0, (bad.size, bad[:5], whole[bad[:5]], sliced[bad[:5]])
    want = orc.find_all(data[: 1 << 20], wbits, wbits + 6, False)
    assert np.array_equal(sliced[: 1 << 20], want)


def test_sliced_finder_dead_entries_cross_steps():
    """a word whose home is bucket 0, repeated every few thousand positions over five windows: the clear of bucket 0 at
    insertion W-1 removes one copy early, whose own retirement then removes a LATER copy early, and so on through every
    step (an entry removed early is carried into the next step as dead)"""
    from oracle import orc
    w = next(c for c in range(1, 1 << 26) if orc.home(c, 22) == 0)
    word = np.frombuffer(int(w).to_bytes(4, "little"), dtype=np.uint8)
    rng = np.random.default_rng(31)
    data = rng.integers(1, 255, 330_000, dtype=np.uint8)
    for at in range(50, 329_000, 3001):
        data[at:at + 4] = word
    _check_find(data, 16, 1 << 20)
    _check_stream(data, 16, 1 << 20)


def test_ballot_ranking_fallback_of_the_finders(monkeypatch):
    """a context whose self-check of lane-ordered LDS atomics failed (forced: MI_LZ_NO_ARANK=1 at creation) ranks with ballots
    in every radix scatter and sorts by cluster with two radix passes instead of the cursor placement: same results"""
    from compression_algorithms_amd import lz
    from compression_algorithms_amd.context import Context
    from oracle import orc
    monkeypatch.setenv("MI_LZ_NO_ARANK", "1")
    ctx = Context(0)
    try:
        data = synth.enwik_like(300_000, seed=51).numpy()
        p = lz.params("lz77", 16, 131072)
        got = lz.find_all32(data, p, ctx).cpu().numpy().view(np.uint32)
        for at in range(0, len(data), 131072):
            assert np.array_equal(got[at:at + 131072], orc.find_all(data[at:at + 131072], 16, 22, False))
        p = lz.params("deflate")
        got = lz.find_all(data, p, ctx).cpu().numpy().view(np.uint16)
        for at in range(0, len(data), 65536):
            want = orc.find_all(data[at:at + 65536], 15, 20, True)
            assert np.array_equal(got[at:at + 65536], np.where(want == 0xFFFFFFFF, 0xFFFF, want).astype(np.uint16))
    finally:
        ctx.close()


@pytest.mark.parametrize("wbits,tbits", [(16, 20), (16, 24), (13, 19), (15, 18)])
def test_other_table_sizes(wbits, tbits):
    """the reference ties TABLE_SIZE to the window (lz77.h:6-8); the ABI does not: a table of 2^20 buckets under a 64 KiB window
    (16 x the load: long clusters, parts cut much finer), 2^24 buckets (whole-block finder: the sliced one sorts 24-bit keys),
    small windows with many steps"""
    from compression_algorithms_amd import lz, _lib
    from oracle import orc
    data = synth.enwik_like(400_000, seed=61).numpy()
    p = _lib.LzParams(wbits, 4, tbits, 0, 262144)
    got = lz.find_all32(data, p).cpu().numpy().view(np.uint32)
    for at in range(0, len(data), 262144):
        want = orc.find_all(data[at:at + 262144], wbits, tbits, False)
        g = got[at:at + 262144]
        bad = np.flatnonzero(g != want)
        assert bad.size == 0, f"w{wbits} T{tbits} block at {at}: {bad.size} mismatches, first {bad[:5]}"
    st = lz.compress(data, p)
    assert np.array_equal(lz.decompress(st).cpu().numpy(), data)


def test_one_flagged_block_is_redone_alone():
    """six blocks of text, ONE of them with a 40 000-byte run of one value (a cluster above the sliced finder's capacity):
    that block alone goes through the whole-block finder, in its own rows of the workspace; every block must match the oracle"""
    data = synth.enwik_like(6 * 131072, seed=71).numpy().copy()
    data[3 * 131072 + 5000: 3 * 131072 + 45000] = 0
    data[5 * 131072 + 100: 5 * 131072 + 130] = 7                      # (a short run: no flag)
    _check_find(data, 16, 131072)
    _check_stream(data, 16, 131072)
    data2 = synth.enwik_like(5 * 262144, seed=72).numpy().copy()      # the FIRST block flagged, 256 KiB blocks
    data2[1000:60000] = 0x41
    _check_find(data2, 16, 262144)
