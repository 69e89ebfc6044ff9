"""Decoders on malformed input: a corrupt offset table, record field or tree must come back as
MI_ERR_CORRUPT — never an out-of-bounds read.  Every case keeps the buffers the kernel may legally touch small and
exact, so a missing check would show as a fault rather than pass by luck."""
import ctypes as C

import numpy as np
import pytest
import torch

from compression_algorithms_amd import _lib, synth

pytestmark = pytest.mark.gpu
CORRUPT = 8


def _status(fn):
    try:
        fn()
    except _lib.MiError as e:
        return e.status
    return 0


@pytest.mark.parametrize("flavour,wbits", [("deflate", None), ("lz77", 14)])
def test_lz_decode_rejects_bad_tables_and_truncation(flavour, wbits):
    from compression_algorithms_amd import lz
    data = synth.enwik_like(150_000, seed=11).numpy()
    st = lz.compress(data, lz.params(flavour, wbits))
    assert np.array_equal(lz.decompress(st).cpu().numpy(), data)
    good = st.block_bits.clone()
    # (a) an offset far outside the stream
    st.block_bits = good.clone(); st.block_bits[1] = 1 << 40
    assert _status(lambda: lz.decompress(st)) == CORRUPT
    # (b) not monotonic
    st.block_bits = good.clone(); st.block_bits[1], st.block_bits[2] = good[2].item(), good[1].item()
    assert _status(lambda: lz.decompress(st)) == CORRUPT
    # (c) the stream buffer is shorter than the table says
    st.block_bits = good.clone()
    full = st.data
    st.data = full[: st.nbytes // 2].clone()
    assert _status(lambda: lz.decompress(st)) == CORRUPT
    # (d) a block that ends in the middle of a token
    st.data = full
    st.block_bits = good.clone(); st.block_bits[1] -= 8; st.block_bits[2:] -= 8
    assert _status(lambda: lz.decompress(st)) == CORRUPT
    st.block_bits = good
    assert np.array_equal(lz.decompress(st).cpu().numpy(), data)


def test_mode_h_decode_rejects_bad_records():
    from compression_algorithms_amd import lz
    data = synth.enwik_like(150_000, seed=12).numpy()
    st = lz.compress_h(data)
    good_bits, good_data = st.block_bits.clone(), st.data.clone()
    st.block_bits = good_bits.clone(); st.block_bits[-1] += 1 << 30
    assert _status(lambda: lz.decompress_h(st)) == CORRUPT
    st.block_bits = good_bits.clone(); st.block_bits[1] += 8               # records are whole words
    assert _status(lambda: lz.decompress_h(st)) == CORRUPT
    st.block_bits = good_bits
    st.data = good_data.clone(); st.data[4:4 + 286] = 1                    # every symbol a 1-bit code: over-subscribed
    assert _status(lambda: lz.decompress_h(st)) == CORRUPT
    st.data = good_data.clone(); st.data[0:4] = 255                        # token count 2^32-1: decoding must still stop
    assert _status(lambda: lz.decompress_h(st)) in (0, CORRUPT)
    st.data = good_data
    assert np.array_equal(lz.decompress_h(st).cpu().numpy(), data)


def test_fse_decode_rejects_bad_records():
    from compression_algorithms_amd import fse
    data = synth.enwik_like(150_000, seed=13).numpy()
    st = fse.compress(data)
    good_off, good_data = st.offsets.clone(), st.data.clone()
    assert np.array_equal(fse.decompress(st).cpu().numpy(), data)
    st.offsets = good_off.clone(); st.offsets[1] = 1 << 45
    assert _status(lambda: fse.decompress(st)) == CORRUPT
    st.offsets = good_off.clone(); st.offsets[1], st.offsets[2] = good_off[2].item(), good_off[1].item()
    assert _status(lambda: fse.decompress(st)) == CORRUPT
    st.offsets = good_off
    # per-lane bit counts set to 2^32-1: they used to index the payload unchecked
    rec = good_data.clone().cpu().numpy()
    nsym = int(np.unpackbits(rec[:32]).sum())
    hdr = 32 + 2 * (nsym + (nsym & 1))
    lens_at = hdr + 2 * 64
    rec[lens_at:lens_at + 4 * 64] = 255
    st.data = torch.from_numpy(rec).to(good_data.device)
    assert _status(lambda: fse.decompress(st)) == CORRUPT
    # counts that do not sum to the table size
    rec = good_data.clone().cpu().numpy(); rec[32] ^= 1
    st.data = torch.from_numpy(rec).to(good_data.device)
    assert _status(lambda: fse.decompress(st)) == CORRUPT
    st.data = good_data
    assert np.array_equal(fse.decompress(st).cpu().numpy(), data)


def test_huffman_decode_rejects_bad_trees_and_offsets():
    from compression_algorithms_amd import huffman
    from compression_algorithms_amd.context import default_context
    ctx = default_context()
    data = synth.enwik_like(100_000, seed=14).numpy()
    r = huffman.huffman_compress(data)
    assert np.array_equal(huffman.huffman_decompress(r).cpu().numpy(), data)
    tree = r._d_tree.clone()

    def decode(tree_t, tile_off, total_bits):
        out = torch.empty(len(data), dtype=torch.uint8, device=ctx.device)
        return ctx.L.mi_huffman_decode_dev(ctx.h, C.c_void_p(r.words.data_ptr()), total_bits, C.c_void_p(tree_t.data_ptr()),
                                           r.n_nodes, C.c_void_p(tile_off.data_ptr()), C.c_void_p(out.data_ptr()), len(data),
                                           ctx.stream_ptr())

    assert decode(tree, r.tile_off, r.total_bits) == 0
    bad = tree.clone()
    off_left = _lib.HuffmanTree.left.offset                    # left[] of mi_huffman_tree
    bad[off_left + 2 * (r.n_nodes - 1): off_left + 2 * r.n_nodes] = torch.tensor([0x30, 0x75], dtype=torch.uint8)   # root.left = 30000
    assert decode(bad, r.tile_off, r.total_bits) == CORRUPT
    off = r.tile_off.clone(); off[1] = 1 << 50
    assert decode(tree, off, r.total_bits) == CORRUPT
    assert decode(tree, r.tile_off, r.total_bits // 2) == CORRUPT          # stream shorter than the offsets say
