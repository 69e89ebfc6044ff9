"""Config 5 from the C boundary: mi_multi_* of include/mi_codec.h — one process, one context per listed
device, contiguous block ranges, the streams gathered into the first device.  This box has ONE GPU, so the device list is
{0, 0} (two contexts on one GPU, the peer-copy transport) or {0, 0, 0}; the assembled stream and block table must be the
single-context entry point's byte for byte, for every stream format:

  deflate tokens  byte aligned: a shard lands in place only when it starts on a dword, otherwise k_bits_append shifts it in
  mode H          records are dword aligned: always in place
  lz77 bit stream shards start mid-byte: the seam dword is OR-ed (what sharded.gather_streams does in torch)

The RCCL entry points cannot meet a second GPU here; mi_multi_selftest_transport drives them on a one-rank communicator
(a send to self inside ncclGroupStart/End), which checks the dlopen'ed table, the group call pattern and the stream order.
The reference loop this replaces: algorithms/deflate/deflate.c:47-63."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from compression_algorithms_amd import _lib, lz, synth
from compression_algorithms_amd.multi import Multi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rule_is_shared_with_the_torch_layer():
    """CPU: mi_multi_shard == sharded.shard_blocks (the gloo tests pin that one)"""
    from compression_algorithms_amd import sharded
    L = _lib.lib()
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    for nblocks in (0, 1, 7, 8, 9, 37, 15259):
        for nd in (1, 2, 3, 8):
            cover = []
            for g in range(nd):
                L.mi_multi_shard(nblocks, g, nd, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == sharded.shard_blocks(nblocks, g, nd)
                cover += list(range(lo.value, hi.value))
            assert cover == list(range(nblocks))


def test_no_device_is_an_error():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    arr = (C.c_int * 2)(0, 0)
    assert _lib.lib().mi_multi_create(C.byref(h), arr, 2) == 9          # MI_ERR_NO_DEVICE: no CPU fallback


def _shards(mm, d, p):
    nblocks = (d.numel() + p.block - 1) // p.block
    out = []
    for g in range(len(mm.devices)):
        lo, hi = mm.shard(nblocks, g)
        out.append(d[lo * p.block: min(hi * p.block, d.numel())].clone() if hi > lo else None)
    return out


CASES = [("deflate-T", 0, 37 * 65536 - 4321), ("deflate-H", 1, 37 * 65536 - 4321), ("lz77w14", 0, 37 * 65536 - 77),
         ("lz77w16", 0, 11 * 65536 + 5), ("deflate-T-small-blocks", 0, 200_003), ("deflate-H-fewer-blocks-than-devices", 1, 70_000),
         ("lz77w16-256KiB-blocks", 0, 7 * 262144 + 4097)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode_h,n", CASES)
@pytest.mark.parametrize("ndev", [2, 3])
def test_two_contexts_on_one_gpu_equal_the_single_context_stream(name, mode_h, n, ndev):
    p = {"deflate-T": lz.params("deflate"), "deflate-H": lz.params("deflate"), "lz77w14": lz.params("lz77", 14),
         "lz77w16": lz.params("lz77", 16), "deflate-T-small-blocks": lz.params("deflate", block=4099),
         "deflate-H-fewer-blocks-than-devices": lz.params("deflate"), "lz77w16-256KiB-blocks": lz.params("lz77", 16, 262144)}[name]
    data = synth.enwik_like(n, seed=404)
    d = data.cuda()
    one = lz.compress_h(d, p) if mode_h else lz.compress(d, p)
    mm = Multi([0] * ndev)
    assert mm.transport == "peer-copy"
    got = mm.compress_dev(_shards(mm, d, p), n, p, mode_h=bool(mode_h))
    assert torch.equal(got.block_bits, one.block_bits)
    assert got.nbytes == one.nbytes
    assert torch.equal(got.data[: got.nbytes], one.data[: one.nbytes])
    back = lz.decompress_h(got) if mode_h else lz.decompress(got)
    assert torch.equal(back.cpu(), data)
    # a second call on the same object (buffers cached, staging reused) with another size
    n2 = n // 2 + 13
    d2 = d[:n2].clone()
    one2 = lz.compress_h(d2, p) if mode_h else lz.compress(d2, p)
    got2 = mm.compress_dev(_shards(mm, d2, p), n2, p, mode_h=bool(mode_h))
    assert torch.equal(got2.block_bits, one2.block_bits) and torch.equal(got2.data[: got2.nbytes], one2.data[: one2.nbytes])
    mm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode_h", [0, 1])
def test_host_buffer_entry_points(mode_h):
    p = lz.params("deflate")
    data = synth.enwik_like(23 * 65536 + 999, seed=405)
    one = lz.compress_h(data, p) if mode_h else lz.compress(data, p)
    mm = Multi([0, 0])
    stream, bits = mm.compress_host(data.numpy(), p, mode_h=bool(mode_h))
    assert np.array_equal(bits.astype(np.int64), one.block_bits.cpu().numpy())
    assert np.array_equal(stream, one.data[: one.nbytes].cpu().numpy())
    e_stream, e_bits = mm.compress_host(np.zeros(0, dtype=np.uint8), p, mode_h=bool(mode_h))     # empty input: one table entry
    assert e_stream.size == 0 and list(e_bits) == [0]
    mm.close()


@pytest.mark.gpu
def test_rccl_entry_points_on_a_one_rank_communicator():
    """librccl.so loaded with dlopen, ncclCommInitAll over {0}, one ncclSend/ncclRecv pair to self inside a group, the bytes
    checked on the device — in a child process (MI_MULTI_TRANSPORT is read when the object is created)"""
    body = """
        from compression_algorithms_amd.multi import Multi
        mm = Multi([0])
        assert mm.transport == "rccl", mm.transport
        mm.selftest_transport(3 * 1024 * 1024 + 17)
        mm.close()
        print("rccl selftest ok")
    """
    env = dict(os.environ, MI_MULTI_TRANSPORT="rccl", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(body)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_peer_copy_selftest_and_duplicate_devices_refuse_rccl():
    mm = Multi([0, 0])
    mm.selftest_transport(1 << 20)
    mm.close()
    body = """
        import ctypes as C
        from compression_algorithms_amd import _lib
        h = C.c_void_p()
        assert _lib.lib().mi_multi_create(C.byref(h), (C.c_int * 2)(0, 0), 2) == 11      # MI_ERR_TRANSPORT: one rank per device
        print("refused")
    """
    env = dict(os.environ, MI_MULTI_TRANSPORT="rccl", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(body)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "refused" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_dropin_compress_spreads_over_the_listed_devices(tmp_path):
    """the reference's own entry point: compress("dir/name") of algorithms/deflate/deflate.c:9-69 with MI_CODEC_DEVICES=0,0
    writes the same .deflate file (and side-car) as on one device"""
    data = synth.enwik_like(9 * 65536 + 321, seed=406).numpy()
    src = tmp_path / "in" / "sample"
    src.parent.mkdir()
    src.write_bytes(data.tobytes())
    outs = {}
    for tag, env_extra in (("one", {}), ("multi", {"MI_CODEC_DEVICES": "0,0"}), ("multi_h", {"MI_CODEC_DEVICES": "0,0,0", "MI_DEFLATE_MODE": "H"}),
                           ("one_h", {"MI_DEFLATE_MODE": "H"})):
        wd = tmp_path / tag
        wd.mkdir()
        body = f"""
            import ctypes as C
            L = C.CDLL({os.path.join(_lib.LIB_DIR, 'libmi_deflate.so')!r})
            class SD(C.Structure):
                _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("name", C.c_char_p)]
            L.compress.restype = SD
            L.compress.argtypes = [C.c_char_p]
            sd = L.compress({str(src).encode()!r})
            print(sd.name.decode())
        """
        env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
        r = subprocess.run([sys.executable, "-c", textwrap.dedent(body)], env=env, cwd=str(wd), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[tag] = (wd / "sample.deflate").read_bytes()
    assert outs["multi"] == outs["one"]
    assert outs["multi_h"] == outs["one_h"]
