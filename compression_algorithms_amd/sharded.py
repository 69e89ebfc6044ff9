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

HUFF_TILE = 32768          # MI_HUFFMAN_TILE of include/mi_codec.h: the output tile of the parallel Huffman decoder


def shard_blocks(nblocks, rank, world):
    """contiguous block range [lo, hi) of rank `rank`"""
    per = (nblocks + world - 1) // world
    lo = min(rank * per, nblocks)
    return lo, min(lo + per, nblocks)


def shard_bytes(n, block, rank, world):
    nblocks = (n + block - 1) // block
    lo, hi = shard_blocks(nblocks, rank, world)
    return lo * block, min(hi * block, n)


def _shift_left_bits(stream, nbits, sh):
    """uint8 tensor holding `nbits` stream bits (LSB first, pad bits zero) -> the same bits starting `sh` (0..7) bits into
    byte 0: ceil((sh + nbits) / 8) bytes, pad bits zero"""
    nb_in = (nbits + 7) // 8
    nb_out = (sh + nbits + 7) // 8
    if sh == 0:
        return stream[:nb_in]
    s16 = stream[:nb_in].to(torch.int32)
    out = torch.zeros(nb_out, dtype=torch.int32, device=stream.device)
    out[:nb_in] = (s16 << sh) & 0xFF
    k = min(nb_in, nb_out - 1)
    out[1:1 + k] |= s16[:k] >> (8 - sh)
    return out.to(torch.uint8)


def gather_streams(data, block_bits, dst=0, group=None):
    """data: uint8 tensor holding this rank's stream from bit 0 (LSB-first bits, pad bits zero); block_bits: int64 [nb+1]
    exclusive prefix in bits.  Returns on `dst`: (stream uint8 tensor, global int64 block table in bits), else (None, None).

    The result is BIT-contiguous, exactly the stream one GPU writes for the whole buffer: rank r's bits start where rank
    r-1's end.  For the byte-aligned flavours (deflate tokens, mode-H and FSE records) that is a byte concatenation and the
    peers' streams are received straight into their place; for the bit-packed lz77 flavour (config 2) every rank first
    shifts its stream by its global bit offset modulo 8 — on its own GPU — and `dst` OR-merges the byte two neighbours share
    (the same seam rule as the whole-buffer Huffman below).

    Exchange: one all_gather of {bits, blocks} per rank (16 B each), then ONE group of point-to-point
    transfers (ncclGroupStart .. ncclSend/ncclRecv x (world-1) .. ncclGroupEnd under RCCL): every peer owns a direct
    xGMI link into `dst`, so the variable-length gather is not ring-bound.  ONE host read: the gathered sizes (receive
    buffers are sized on the host); this rank's own length travels in the same tensor."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = data.device
    meta = torch.stack([block_bits[-1].to(torch.int64), torch.tensor(block_bits.numel() - 1, dtype=torch.int64, device=dev)])
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu().tolist()
    bits = [int(m[0]) for m in metas]
    nbs = [int(m[1]) for m in metas]
    start = [0] * (world + 1)
    for r in range(world):
        start[r + 1] = start[r] + bits[r]
    total = start[world]
    aligned = all(start[r] % 8 == 0 for r in range(world))

    def part_bytes(r):                                   # bytes rank r sends: its bits placed (start % 8) bits into byte 0
        return ((start[r] % 8) + bits[r] + 7) // 8 if bits[r] else 0

    mine = _shift_left_bits(data, bits[rank], start[rank] % 8) if bits[rank] else data[:0]
    if rank != dst:
        ops = []
        if part_bytes(rank):
            ops.append(dist.P2POp(dist.isend, mine.contiguous(), dst, group))
        ops.append(dist.P2POp(dist.isend, block_bits.contiguous(), dst, group))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        return None, None
    out = torch.zeros((total + 7) // 8, dtype=torch.uint8, device=dev) if not aligned else \
        torch.empty((total + 7) // 8, dtype=torch.uint8, device=dev)
    parts, tables, ops = [None] * world, [None] * world, []
    for r in range(world):
        b0 = start[r] // 8
        if r == rank:
            tables[r] = block_bits
            continue
        # byte-aligned streams land in place; shifted ones share a seam byte with their neighbour and are OR-ed in below
        parts[r] = out[b0:b0 + part_bytes(r)] if aligned else torch.empty(part_bytes(r), dtype=torch.uint8, device=dev)
        tables[r] = torch.empty(nbs[r] + 1, dtype=torch.int64, device=dev)
        if part_bytes(r):
            ops.append(dist.P2POp(dist.irecv, parts[r], r, group))
        ops.append(dist.P2POp(dist.irecv, tables[r], r, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    b0 = start[rank] // 8
    if aligned:
        out[b0:b0 + part_bytes(rank)] = mine
    else:
        parts[rank] = mine
        for r in range(world):
            if part_bytes(r):
                out[start[r] // 8:start[r] // 8 + part_bytes(r)] |= parts[r]
    table = torch.cat([tables[r][:-1] + start[r] for r in range(world)] + [torch.tensor([total], dtype=torch.int64, device=dev)])
    return out, table


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
    order; when every range but the last is a multiple of 32 768 bytes — shard_bytes() with 64 KiB blocks gives that — the
    result carries the tile table of the parallel decoder, otherwise its `tile_off` is None).  The reference semantics are kept: ONE tree over the whole buffer
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
    # the parallel decoder cuts the OUTPUT into fixed 32 768-byte tiles: the per-rank tables only line up with that grid if
    # every shard but the last is a whole number of tiles.  Otherwise no table is handed on (the decoder then walks the
    # stream with one lane, slowly but correctly; k_huff_decode refuses a table that does not fit its grid).
    aligned = all(int(metas[r][1]) % HUFF_TILE == 0 for r in range(world - 1))
    tile_table = torch.cat([t[:-1] for t in tiles] + [torch.tensor([total], dtype=torch.int64, device=dev)]) if aligned else None
    return ShardedHuffman(out[: (total + 31) // 32], total, tree, sum(int(m[1]) for m in metas), tile_table)
