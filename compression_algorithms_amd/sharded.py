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


class ShardedHuffman:
    """what rank `dst` holds after huffman_compress: the stream of the WHOLE buffer, as one GPU would have written it"""

    def __init__(self, words, total_bits, tree, n, tile_off):
        self.words, self.total_bits, self.tree, self.n, self.tile_off = words, total_bits, tree, n, tile_off

    @property
    def word_idx(self):
        return self.total_bits // 32

    @property
    def bit_idx(self):
        return self.total_bits % 32


def huffman_compress(shard, engine, dst=0, group=None):
    """Whole-buffer Huffman over a buffer whose contiguous byte ranges live on the ranks of `group` (rank order = byte
    order; ranges should start on multiples of 32 768 bytes — shard_bytes() with 64 KiB blocks does — so that the tile
    table of the parallel decoder stays valid).  The reference semantics are kept: ONE tree over the whole buffer
    (algorithms/huffman/huffman.c:179-215), tree-path codes, MSB-first u32 words (:18-48, :267-328).

    Exchange steps, everything else is per-rank work through `engine`
    (compression_algorithms_amd.huffman.HipShardEngine on GPUs):
      1. all_reduce(sum) of the u64[256] shard histograms (2 KiB)      -> every rank builds the identical tree
      2. all_gather of the shard bit counts (8 B per rank)             -> global bit offset of every shard
      3. each rank packs its shard starting (offset mod 32) bits into its first word
      4. point-to-point gather of the word ranges to `dst`; the one word two neighbours share is OR-merged there
    Returns a ShardedHuffman on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    h, state = engine.hist(shard)
    dev = h.device
    hg = h.clone()
    dist.all_reduce(hg, op=dist.ReduceOp.SUM, group=group)
    tree = engine.build(hg)                                  # same input on every rank -> same tree, or the same error
    my_bits = engine.shard_bits(h, tree)
    meta = torch.tensor([my_bits, state["n"], state["ntiles"]], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu().tolist()
    bits = [int(m[0]) for m in metas]
    starts = [0] * (world + 1)
    for r in range(world):
        starts[r + 1] = starts[r] + bits[r]
    total = starts[world]
    words, tile_off = engine.encode(state, tree, starts[rank] % 32, my_bits)
    # global tile offsets: local ones are relative to this rank's first word
    tile_glob = tile_off + (starts[rank] // 32) * 32

    def nwords(r):
        return ((starts[r] % 32) + bits[r] + 31) // 32 if bits[r] else 0

    if rank != dst:
        ops = []
        if nwords(rank):
            ops.append(dist.P2POp(dist.isend, words.contiguous(), dst, group))
        ops.append(dist.P2POp(dist.isend, tile_glob.contiguous(), dst, group))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        return None
    parts, tiles, ops = [None] * world, [None] * world, []
    for r in range(world):
        if r == rank:
            parts[r], tiles[r] = words, tile_glob
            continue
        parts[r] = torch.empty(nwords(r), dtype=torch.int32, device=dev)
        tiles[r] = torch.empty(int(metas[r][2]) + 1, dtype=torch.int64, device=dev)
        if nwords(r):
            ops.append(dist.P2POp(dist.irecv, parts[r], r, group))
        ops.append(dist.P2POp(dist.irecv, tiles[r], r, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    out = torch.zeros((total + 31) // 32 + 1, dtype=torch.int32, device=dev)       # + the word the decoder peeks at
    for r in range(world):
        if nwords(r):
            w0 = starts[r] // 32
            out[w0:w0 + nwords(r)] |= parts[r]                                     # seam words are shared: OR-merge
    tile_table = torch.cat([t[:-1] for t in tiles] + [torch.tensor([total], dtype=torch.int64, device=dev)])
    return ShardedHuffman(out[: (total + 31) // 32], total, tree, sum(int(m[1]) for m in metas), tile_table)
