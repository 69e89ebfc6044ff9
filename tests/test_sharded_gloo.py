"""N > 1 host logic on CPU: two gloo ranks shard the blocks, "encode" their range (the oracle
stands in for the GPU encoder here — this test is about sharding and assembly), gather to rank 0,
and rank 0 must hold exactly the single-process stream and block table."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from compression_algorithms_amd import sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synth.enwik_like(n, seed=2).numpy()
    lo, hi = sharded.shard_bytes(n, 65536, rank, world)
    tok, sizes = orc.deflate_stream(data[lo:hi], 65536, True)
    bits = torch.tensor(np.concatenate([[0], np.cumsum(sizes.astype(np.int64) * 8)]), dtype=torch.int64)
    stream, table = sharded.gather_streams(torch.from_numpy(tok.copy()), bits, dst=0)
    if rank == 0:
        want, wsizes = orc.deflate_stream(data, 65536, True)
        ok = np.array_equal(stream.numpy(), want) and \
            np.array_equal(table.numpy(), np.concatenate([[0], np.cumsum(wsizes.astype(np.int64) * 8)]))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_assemble_the_single_process_stream():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n = 5 * 65536 + 777          # odd number of blocks: ranks get 3 and 3(last short)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
    assert ok
    assert all(p.exitcode == 0 for p in procs)


def test_shard_ranges_cover_everything():
    for nblocks in (0, 1, 7, 8, 9, 15259):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = sharded.shard_blocks(nblocks, r, world)
                assert 0 <= lo <= hi <= nblocks
                seen.extend(range(lo, hi))
            assert seen == list(range(nblocks))
