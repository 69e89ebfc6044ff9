"""The row replay of the exported clusters (k_lz2_rows, lz2_find.hip: four clusters of 128..511 entries per wave, one per 16-lane
row) against the oracle, and its switches: MI_LZ_ROWS=0 (the wave replay for every class: two wave classes instead of three) and
MI_LZ_BIG_SPLIT=0 (one launch for the wave classes) must give the same streams.  Inputs: text (its frequent words are the
128..511-entry clusters) and a low-entropy alphabet with long-lived clusters; deflate (W = 32 KiB: retirements in every cluster)
and the shipped lz77 window (W = 16 KiB).  Reference behaviour emulated: algorithms/lz77/lz77.c:55-108,
algorithms/deflate/lz77.c:77-174."""
import pytest

from test_fallback_chain_gpu import _child

pytestmark = pytest.mark.gpu

BODY = """
    ctx = Context(0)
    rng = np.random.default_rng(5)
    text = synth.enwik_like(40 * 65536 - 777, seed=91).numpy()
    low = rng.choice(np.frombuffer(b"abcdefgh \\n", np.uint8), size=12 * 65536 + 5, p=[.3, .2, .1, .1, .05, .05, .05, .05, .05, .05]).astype(np.uint8)
    for name, data in (("text", text), ("low", low)):
        for p in (lz.params("deflate"), lz.params("lz77", 14)):
            st = lz.compress(data, p, ctx)
            ctx.sync()
            assert oracle_equal(st, data, p), (name, "stream differs from the oracle")
            assert np.array_equal(lz.decompress(st, ctx).cpu().numpy(), data)
    assert ctx.order_violations() == 0
    print("ok")
"""


@pytest.mark.parametrize("env", [{}, {"MI_LZ_ROWS": "0"}, {"MI_LZ_BIG_SPLIT": "0"}, {"MI_LZ_ROWS": "0", "MI_LZ_BIG_SPLIT": "0"},
                                 {"MI_LZ_ROWS_WAVES": "1"}])
def test_row_replay_and_its_switches_equal_the_oracle(env):
    assert "ok" in _child(BODY, **env)
