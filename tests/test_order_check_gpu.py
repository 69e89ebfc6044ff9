"""The stable radix scatters rank by the order in which one returning LDS atomic serves its lanes (lz_common.h radix_pass,
LZP_ARANK) — lane order on gfx950, not an ISA promise.  Every consumer of a sort checks the
final (key, time) order; a violation must never be silent.  MI_LZ_TEST_BREAK_RANK=1 makes the scatter mis-rank on purpose
(neighbouring lanes with one digit trade places: what an out-of-order atomic would do):

  * the asynchronous device encoders report it (mi_order_violations, mi_sync -> MI_ERR_UNSTABLE) and the context ranks
    with ballots from its next call on — the second encode is the oracle's stream;
  * the host-buffer entry points (what the drop-ins call) encode again by themselves and return the oracle's stream;
  * an undisturbed context counts no violation on any flavour."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from compression_algorithms_amd import _lib, lz, synth
from compression_algorithms_amd.context import Context

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_LIB = os.path.join(_lib.LIB_DIR, "..", "lib_test", "libmi_codec.so")     # the core library built with -DMI_TEST_HOOKS (csrc/Makefile)


def _run_with_broken_ranking(body):
    """the hook exists only in the TEST build of the library, and a library is chosen when the process starts: run `body`
    in a child process with MI_CODEC_LIB=lib_test and MI_LZ_TEST_BREAK_RANK=1"""
    if not os.path.exists(TEST_LIB):
        _lib.build()
    env = dict(os.environ, MI_CODEC_LIB=os.path.abspath(TEST_LIB), MI_LZ_TEST_BREAK_RANK="1", PYTHONPATH=ROOT)
    prelude = """
        import ctypes as C
        import numpy as np, torch
        from compression_algorithms_amd import _lib, lz, synth
        from compression_algorithms_amd.context import Context
        from oracle import orc
    """
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(prelude) + textwrap.dedent(body)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("shape", ["deflate", "lz77w14"])
def test_broken_ranking_is_noticed_and_the_context_recovers(shape):
    # (the hook acts on the 64 KiB pipeline; the time-sliced finder of larger blocks runs the same checks — see
    #  test_an_undisturbed_context_counts_nothing — but its replay is not fed mis-ordered events on purpose)
    _run_with_broken_ranking(f"""
        shape = {shape!r}
        ctx = Context(0)
        p = {{"deflate": lz.params("deflate"), "lz77w14": lz.params("lz77", 14)}}[shape]
        data = synth.enwik_like(3 * p.block + 1234, seed=91).numpy()
        first = lz.compress(data, p, ctx)
        torch.cuda.synchronize()
        assert ctx.order_violations() >= 1, "the mis-ranked scatter went unnoticed"
        try:
            first.nbytes                                      # the Python device path: the first read of the result says so (ADVICE r3)
            raise SystemExit("LzStream did not report the violation")
        except _lib.MiError as e:
            assert e.status == 10
        try:
            ctx.sync()
            raise SystemExit("mi_sync did not report the violation")
        except _lib.MiError as e:
            assert e.status == 10                             # MI_ERR_UNSTABLE, once
        ctx.sync()
        del first                                             # a valid-looking stream that is not the reference's: why it must be checked
        seen = ctx.order_violations()
        second = lz.compress(data, p, ctx)                    # ballots now: the reference's stream
        ctx.sync()
        assert ctx.order_violations() == seen
        if p.deflate:
            want = orc.deflate_stream(data, 65536, True)[0]
            assert np.array_equal(second.data[: second.nbytes].cpu().numpy(), want)
        else:
            blocks = [orc.lz77_encode(data[a:a + p.block], p.wbits, 4) for a in range(0, len(data), p.block)]
            tb = np.concatenate([[0], np.cumsum([nb for _, nb in blocks])])
            assert np.array_equal(second.block_bits.cpu().numpy(), tb)
            bits = np.unpackbits(second.data[: second.nbytes].cpu().numpy(), bitorder="little")
            for b, (s, nb) in enumerate(blocks):
                assert np.array_equal(bits[tb[b]:tb[b + 1]], np.unpackbits(s, bitorder="little")[:nb]), b
    """)


@pytest.mark.parametrize("mode_h", [False, True])
def test_host_entry_points_encode_again_by_themselves(mode_h):
    _run_with_broken_ranking(f"""
        mode_h = {mode_h!r}
        ctx = Context(0)
        p = lz.params("deflate")
        data = synth.enwik_like(5 * 65536 + 99, seed=92).numpy()
        n = len(data)
        nblocks = (n + 65535) // 65536
        cap = (int(ctx.L.mi_deflate_h_bound_bytes(n, C.byref(p))) if mode_h else lz.bound_bytes(n, p)) + 64
        out = np.zeros(cap, np.uint8)
        bits = np.zeros(nblocks + 1, np.uint64)
        fn = ctx.L.mi_deflate_h_encode if mode_h else ctx.L.mi_lz_encode
        rc = fn(ctx.h, C.byref(p), C.c_void_p(data.ctypes.data), C.c_uint64(n), C.c_void_p(out.ctypes.data), C.c_uint64(cap), C.c_void_p(bits.ctypes.data))
        assert rc == 0
        assert ctx.order_violations() >= 1
        ctx.sync()                                            # handled inside the call: nothing left to report
        tok, sizes = orc.deflate_stream(data, 65536, True)
        if not mode_h:
            assert np.array_equal(out[: int(bits[-1]) // 8], tok)
        else:
            at = 0
            for b, s in enumerate(sizes):
                want = orc.defh_encode_block(tok[at:at + int(s)])
                at += int(s)
                assert np.array_equal(out[int(bits[b]) // 8:int(bits[b + 1]) // 8], want), b
    """)


def test_the_shipped_library_has_no_test_hook(monkeypatch):
    """MI_LZ_TEST_BREAK_RANK means nothing to the product library: same stream, no violation"""
    monkeypatch.setenv("MI_LZ_TEST_BREAK_RANK", "1")
    ctx = Context(0)
    data = synth.enwik_like(2 * 65536 + 7, seed=94).numpy()
    a = lz.compress(data, lz.params("deflate"), ctx)
    ctx.sync()
    from oracle import orc
    import numpy as np
    assert np.array_equal(a.data[: a.nbytes].cpu().numpy(), orc.deflate_stream(data, 65536, True)[0])
    assert ctx.order_violations() == 0


def test_an_undisturbed_context_counts_nothing():
    ctx = Context(0)
    x = synth.enwik_like(40 * 65536 + 5, seed=93, device="cuda")
    for p in (lz.params("deflate"), lz.params("lz77", 14), lz.params("lz77", 16), lz.params("lz77", 16, 262144)):
        lz.compress(x, p, ctx)
    lz.compress_h(x, lz.params("deflate"), ctx)
    ctx.sync()
    assert ctx.order_violations() == 0
