"""Multi-GPU layout of the block-parallel codecs: one process per GPU, independent blocks.

Blocks never depend on each other (fresh table per block), so the path shards with NO data-path
collective: rank r encodes the contiguous block range shard_blocks(nblocks, r, world).  The
only exchange is the optional assembly of ONE output stream on rank 0, the north star's "RCCL
gather of per-block compressed streams": an all_gather of the per-rank byte counts and block
tables (a few KB) followed by point-to-point sends of the variable-length streams — on an xGMI
node every peer has its own direct link into rank 0, so the gather is not ring-bound
(SURVEY.md 8e).  Works on any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" in
the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_blocks(nblocks, rank, world):
    """contiguous block range [lo, hi) of rank `rank`"""
    per = (nblocks + world - 1) // world
    lo = min(rank * per, nblocks)
    return lo, min(lo + per, nblocks)


def shard_bytes(n, block, rank, world):
    nblocks = (n + block - 1) // block
    lo, hi = shard_blocks(nblocks, rank, world)
    return lo * block, min(hi * block, n)


def gather_streams(data, block_bits, dst=0, group=None):
    """data: uint8 tensor (this rank's stream, byte aligned); block_bits: int64 [nb+1] exclusive
    prefix in bits.  Returns on `dst`: (stream uint8 tensor, global int64 block table), else (None, None).
    Streams are byte-concatenated: use with byte-aligned flavours (deflate tokens, mode-H / FSE records);
    for the bit-packed lz77 flavour keep the per-rank streams separate or pad each to a byte.

    Exchange: one all_gather of {bytes, blocks} per rank (16 B each), then ONE group of point-to-point
    transfers (ncclGroupStart .. ncclSend/ncclRecv x (world-1) .. ncclGroupEnd under RCCL): every peer owns a direct
    xGMI link into `dst`, so the variable-length gather is not ring-bound.  Two host reads: this rank's stream length
    and the gathered sizes (the receive buffers have to be sized on the host)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = data.device
    nbytes = (int(block_bits[-1]) + 7) // 8
    meta = torch.tensor([nbytes, block_bits.numel() - 1], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu().tolist()
    sizes = [int(m[0]) for m in metas]
    nbs = [int(m[1]) for m in metas]
    if rank == dst:
        streams = [None] * world
        tables = [None] * world
        ops = []
        for r in range(world):
            if r == rank:
                streams[r], tables[r] = data[:nbytes], block_bits
            else:
                streams[r] = torch.empty(sizes[r], dtype=torch.uint8, device=dev)
                tables[r] = torch.empty(nbs[r] + 1, dtype=torch.int64, device=dev)
                if sizes[r]:
                    ops.append(dist.P2POp(dist.irecv, streams[r], r, group))
                ops.append(dist.P2POp(dist.irecv, tables[r], r, group))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        out = torch.cat(streams)
        base, parts = 0, []
        for r in range(world):
            parts.append(tables[r][:-1] + base)
            base += sizes[r] * 8
        parts.append(torch.tensor([base], dtype=torch.int64, device=dev))
        return out, torch.cat(parts)
    ops = []
    if nbytes:
        ops.append(dist.P2POp(dist.isend, data[:nbytes].contiguous(), dst, group))
    ops.append(dist.P2POp(dist.isend, block_bits.contiguous(), dst, group))
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    return None, None
