"""The fallback pipeline (lz_find.hip) under load it never met in rounds 1-3: EVERY block of whole batches of ordinary text.

Round 3 recorded a GPU memory fault from an A/B build with parts of 2 048 entries (gpurun_out/r3/part_c2048_1.err): with 32 part
numbers such a build sends every full block through the partition's "out of part numbers" exit, i.e. a 3 072-block batch of
text onto the fallback chain — a state no shipped test reached (the fuzz families fall back through giant clusters, a few
blocks at a time; MI_LZ_V2=0 was in no test).  Since then the exit is unreachable by construction (lz2.h: 128 part numbers and
a static_assert with the argument), and this file pins the chain itself:

  * lib_test (-DMI_TEST_HOOKS) + MI_LZ_TEST_FORCE_FALLBACK=1: the partition hands every block to the fallback list; the
    streams must be the oracle's (small multi-batch input) and, at the production batch size (three sets of 3 072 blocks in
    rotation, > 600 MB of text), equal to the unforced context's stream, which test_full_size_gpu.py pins to the oracle;
    mi_lz_path_stats must count every block;
  * MI_LZ_V2=0 (first pipeline for everything, no list) on a multi-batch input against the oracle.
Reference behaviour emulated: algorithms/deflate/lz77.c:77-174, algorithms/lz77/lz77.c:55-108."""
import os
import subprocess
import sys
import textwrap

import pytest

from compression_algorithms_amd import _lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_LIB = os.path.join(_lib.LIB_DIR, "..", "lib_test", "libmi_codec.so")

PRELUDE = """
    import os
    import numpy as np, torch
    from compression_algorithms_amd import _lib, lz, synth
    from compression_algorithms_amd.context import Context
    from oracle import orc

    def oracle_equal(st, data, p):
        if p.deflate:
            want = orc.deflate_stream(data, p.block, True)[0]
            return np.array_equal(st.data[: st.nbytes].cpu().numpy(), want)
        blocks = [orc.lz77_encode(data[a:a + p.block], p.wbits, 4) for a in range(0, len(data), p.block)]
        tb = np.concatenate([[0], np.cumsum([nb for _, nb in blocks])])
        if not np.array_equal(st.block_bits.cpu().numpy(), tb):
            return False
        bits = np.unpackbits(st.data[: st.nbytes].cpu().numpy(), bitorder="little")
        return all(np.array_equal(bits[tb[b]:tb[b + 1]], np.unpackbits(s, bitorder="little")[:nb]) for b, (s, nb) in enumerate(blocks))
"""


def _child(body, **env_extra):
    if not os.path.exists(TEST_LIB):
        _lib.build()
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(PRELUDE) + textwrap.dedent(body)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    return r.stdout


@pytest.mark.parametrize("shape", ["deflate", "lz77w14", "lz77w16"])
def test_every_block_of_several_batches_on_the_fallback_chain(shape):
    out = _child(f"""
        shape = {shape!r}
        p = {{"deflate": lz.params("deflate"), "lz77w14": lz.params("lz77", 14), "lz77w16": lz.params("lz77", 16)}}[shape]
        ctx = Context(0)
        data = synth.enwik_like(200 * 65536 - 4321, seed=77).numpy()        # 200 blocks in batches of 64: three scratch sets in rotation
        st = lz.compress(data, p, ctx)
        ctx.sync()
        stats = ctx.path_stats()
        assert stats["fallback_blocks"] == 200, stats                       # the hook really sent every block there
        assert oracle_equal(st, data, p), "forced-fallback stream differs from the oracle"
        assert np.array_equal(lz.decompress(st, ctx).cpu().numpy(), data)
        if p.deflate:
            h = lz.compress_h(data, p, ctx)
            assert np.array_equal(lz.decompress_h(h, ctx).cpu().numpy(), data)
        print("ok", stats)
    """, MI_CODEC_LIB=os.path.abspath(TEST_LIB), MI_LZ_TEST_FORCE_FALLBACK="1", MI_LZ_BATCH="64")
    assert "ok" in out


def test_production_batches_of_text_on_the_fallback_chain():
    """9 300 blocks (> 3 x 3 072: the batch size and set rotation of the 10^9-byte run), mode H and mode T: the forced context's
    stream equals the unforced context's"""
    out = _child("""
        n = 9300 * 65536 - 99
        data = synth.enwik_like(n, seed=12345, device="cuda")
        p = lz.params("deflate")
        forced = Context(0)                                   # reads MI_LZ_TEST_FORCE_FALLBACK when it is created
        del os.environ["MI_LZ_TEST_FORCE_FALLBACK"]
        plain = Context(0)
        for mode in ("T", "H"):
            a = (lz.compress_h if mode == "H" else lz.compress)(data, p, forced)
            forced.sync()
            b = (lz.compress_h if mode == "H" else lz.compress)(data, p, plain)
            plain.sync()
            assert torch.equal(a.block_bits, b.block_bits), mode
            assert torch.equal(a.data[: a.nbytes], b.data[: b.nbytes]), mode
        fs, ps = forced.path_stats(), plain.path_stats()
        assert fs["fallback_blocks"] == 2 * 9300 and ps["fallback_blocks"] == 0, (fs, ps)
        print("ok", fs, ps)
    """, MI_CODEC_LIB=os.path.abspath(TEST_LIB), MI_LZ_TEST_FORCE_FALLBACK="1")
    assert "ok" in out


@pytest.mark.parametrize("shape", ["deflate", "lz77w14"])
def test_first_pipeline_alone_on_several_batches(shape):
    out = _child(f"""
        shape = {shape!r}
        p = {{"deflate": lz.params("deflate"), "lz77w14": lz.params("lz77", 14)}}[shape]
        ctx = Context(0)
        data = synth.enwik_like(150 * 65536 + 17, seed=78).numpy()
        st = lz.compress(data, p, ctx)
        ctx.sync()
        assert oracle_equal(st, data, p), "MI_LZ_V2=0 stream differs from the oracle"
        print("ok")
    """, MI_LZ_V2="0", MI_LZ_BATCH="64")
    assert "ok" in out


@pytest.mark.parametrize("family", ["pages", "runs"])
def test_warm_calls_on_two_fallback_streams_equal_the_oracle(family):
    """A context that has seen fallback-heavy input sizes the fallback grids by the earlier call's count and runs the chains of
    consecutive batches on two streams (lz_emit.hip: the hint is what an EARLIER call left).  The first call of a context runs
    without, the later ones with: every call's stream must be the oracle's, token for token, and all of them the same bytes."""
    out = _child(f"""
        data = synth.family({family!r}, 77, 160 * 65536 + 4099)
        x = torch.from_numpy(data).cuda()
        p = lz.params("deflate")
        ctx = Context(0)
        os.environ["MI_LZ_BATCH"] = "64"
        outs = []
        for call in range(3):
            st = lz.compress(x, p, ctx)
            assert oracle_equal(st, data, p), call
            outs.append(st.data[: st.nbytes].cpu().numpy().copy())
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
        ps = ctx.path_stats()
        assert ps["fallback_blocks"] > 0, ps
        print("path", ps)
    """, MI_LZ_BATCH="64")
    assert "path" in out
