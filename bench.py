#!/usr/bin/env python3
"""bench.py — encode throughput of the MI355X-native block-parallel compressor.

    python bench.py --gpus N --steps K --warmup W [--workload deflate|deflate-h|lz77w16|lz77w14|huffman|fse]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU (RCCL).  A "step" is one pass of the hot path over one batch of synthetic
enwik-shaped input that is already resident in HBM.  Rank 0 prints ONE JSON line.

Workload at N=1 (BASELINE.json metric "encode GB/s on enwik9", config "deflate (LZ77 ...) on
enwik9, 32 KiB window, 1 MI355X"): 10^9 enwik-shaped bytes cut into 15 259 independent 64 KiB
blocks, reference tokeniser algorithms/deflate/lz77.c:199-280 per block (fresh table per block),
byte-exact token stream.  N > 1: every rank encodes its own 10^9-byte shard of independent
blocks (weak scaling); --gather adds the north star's RCCL gather of the compressed streams
to rank 0 inside the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="deflate", choices=["deflate", "deflate-h", "lz77w16", "lz77w14", "huffman", "fse"])
    ap.add_argument("--bytes", type=int, default=1_000_000_000, help="input bytes per GPU")
    ap.add_argument("--gather", action="store_true", help="N>1: gather the compressed streams to rank 0 (timed)")
    ap.add_argument("--cpu-sample-mb", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices)")
    return ap.parse_args()


def make_step(workload, x, ctx):
    """returns (callable doing one encode pass, callable -> compressed bytes of the last pass, dtype, description)"""
    from compression_algorithms_amd import lz, huffman
    if workload in ("deflate", "lz77w16", "lz77w14"):
        p = {"deflate": lz.params("deflate"), "lz77w16": lz.params("lz77", 16), "lz77w14": lz.params("lz77", 14)}[workload]
        holder = {}

        def step():
            holder["st"] = None          # release the previous output first: the caching allocator then reuses its block
            holder["st"] = lz.compress(x, p, ctx)      # (otherwise the first timed step pays a 2 GB hipMalloc)

        desc = {"deflate": "deflate tokeniser (LZ77, W=32 KiB, len<=31, byte tokens), independent 64 KiB blocks",
                "lz77w16": "lz77 (W=64 KiB, len<=15, bit-packed), independent 64 KiB blocks",
                "lz77w14": "lz77 (W=16 KiB, len<=15, bit-packed), independent 64 KiB blocks"}[workload]
        return step, (lambda: holder["st"].nbytes), (lambda: holder["st"]), "u8", desc
    if workload == "deflate-h":
        holder = {}
        p = lz.params("deflate")

        def step():
            holder["st"] = None
            holder["st"] = lz.compress_h(x, p, ctx)

        return step, (lambda: holder["st"].nbytes), (lambda: holder["st"]), "u8", \
            "deflate tokeniser + per-block dynamic Huffman over the 286-symbol alphabet (mode H), independent 64 KiB blocks"
    if workload == "huffman":
        holder = {}

        def step():
            holder["r"] = None
            holder["r"] = huffman.huffman_compress(x, ctx)

        return step, (lambda: (holder["r"].total_bits + 7) // 8), (lambda: holder["r"]), "u8", "whole-buffer Huffman, one tree"
    if workload == "fse":
        from compression_algorithms_amd import fse
        holder = {}
        p = fse.params()

        def step():
            holder["r"] = None
            holder["r"] = fse.compress(x, p, ctx)

        return step, (lambda: holder["r"].nbytes), (lambda: holder["r"]), "u8", "FSE/tANS table_log 8, independent 64 KiB blocks x 64 sub-streams"
    raise ValueError(workload)


def cpu_baseline(workload, sample, sample_desc):
    """the reference (oracle/_ref, compiled from /root/reference in the build container) or, where
    that is absent, the oracle's CPU restatement, timed single-threaded on the host."""
    import numpy as np
    from oracle import orc, ref
    n = len(sample)
    kind = "port"
    t0 = time.perf_counter()
    if workload == "deflate":
        if ref.available():
            kind = "reference"
            rd = ref.RefDeflate()
            t0 = time.perf_counter()
            rd.stream(sample, independent=True)
        else:
            t0 = time.perf_counter()
            orc.deflate_stream(sample, 65536, True)
    elif workload == "deflate-h":
        # the reference has no entropy stage for this path: the oracle's tokeniser + its mode-H coder
        d = orc.Deflate(65536)
        t0 = time.perf_counter()
        for at in range(0, n, 65536):
            d.fresh()
            orc.defh_encode_block(d.block_encode(sample[at:at + 65536]))
    elif workload in ("lz77w16", "lz77w14"):
        wb = 16 if workload == "lz77w16" else 14
        t0 = time.perf_counter()
        for at in range(0, n, 65536):
            orc.lz77_encode(sample[at:at + 65536].tobytes(), wb, 4)
    elif workload == "huffman":
        if ref.available():
            kind = "reference"
            import ctypes as C
            L = ref._huff()
            src = np.concatenate([sample, np.zeros(64, np.uint8)])
            w = ref._BitWriter()
            t0 = time.perf_counter()
            L.huffman_compress(src.ctypes.data_as(C.c_void_p), n, C.byref(w))      # the reference entry point alone
        else:
            t0 = time.perf_counter()
            orc.huff_encode(sample)
    elif workload == "fse":
        t0 = time.perf_counter()
        for at in range(0, n, 65536):
            orc.fse_encode_block(sample[at:at + 65536], 8, 64, 1)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": kind,
            "sample": f"{sample_desc}; {n} bytes in {dt:.2f} s on 1 of {os.cpu_count()} host cores"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
            local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from compression_algorithms_amd import synth
    from compression_algorithms_amd.context import Context
    ctx = Context(local)

    n = args.bytes
    # every rank owns a different shard of the same enwik-shaped corpus (independent blocks)
    x = synth.enwik_like(n, seed=args.seed + 1000 * rank, device=dev)
    torch.cuda.synchronize()
    step, out_bytes, last, dtype, desc = make_step(args.workload, x, ctx)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def gather_streams():
        """north star: RCCL gather of the per-block compressed streams to rank 0 (sizes and block
        tables first, then the variable-length streams point to point: sharded.gather_streams)"""
        from compression_algorithms_amd import sharded
        st = last()
        table = st.block_bits if hasattr(st, "block_bits") else st.offsets
        sharded.gather_streams(st.data, table, dst=0)

    for _ in range(args.warmup):
        step()
        if dist is not None and args.gather:
            gather_streams()
    barrier()
    ctx.set_profiling(True)
    ctx.kernel_times()              # reset
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if dist is not None and args.gather:
            gather_streams()
    barrier()
    dt = time.perf_counter() - t0
    ctx.set_profiling(False)
    ktimes = ctx.kernel_times()
    c = out_bytes()

    t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        total_bytes = n * world * args.steps
        value = total_bytes / dt / 1e9
        # dominant kernel: largest share of the timed region
        roof = None
        if ktimes:
            dom = max(ktimes, key=lambda k: k["ms"] * k["launches"])
            launches_per_step = dom["launches"] / args.steps
            passes = 2 if args.workload == "huffman" else 1
            alg_bytes_per_launch = (passes * n + c) / launches_per_step       # (passes*n + c) per job, split over the launches
            achieved = alg_bytes_per_launch / (dom["ms"] * 1e-3) / 1e9
            traffic = None
            tj = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
            if os.path.exists(tj):
                try:
                    traffic = json.load(open(tj)).get(dom["name"])
                except Exception:
                    traffic = None
            # the dominant kernel's OWN share of the algorithmic bytes, for orientation (Huffman: the histogram reads n,
            # the encoder reads n and writes c; the LZ kernels each see the block once)
            own = {"k_huff_hist": n, "k_huff_encode": n + c}.get(dom["name"], (passes * n + c)) / launches_per_step
            roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "achieved_own_bytes": round(own / (dom["ms"] * 1e-3) / 1e9, 2),
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(dom["ms"], 4), "launches_per_step": launches_per_step,
                    "kernel_share": round(dom["ms"] * dom["launches"] / (dt * 1e3), 3),
                    "all_kernels_ms_per_step": {k["name"]: round(k["ms"] * k["launches"] / args.steps, 3) for k in ktimes}}
        cpu = None
        if not args.no_cpu_baseline:
            rate = {"deflate": 0.016, "deflate-h": 0.012, "lz77w16": 0.012, "lz77w14": 0.014, "huffman": 0.23, "fse": 0.15}[args.workload]
            mb = args.cpu_sample_mb if args.cpu_sample_mb else min(n / 1e6, max(4.0, 15.0 * rate * 1e3))
            nsamp = int(mb * 1e6) // 65536 * 65536 or min(n, 65536)
            sample = x[:nsamp].cpu().numpy()
            cpu = cpu_baseline(args.workload, sample, f"first {nsamp} bytes of the rank-0 buffer")
        line = {
            "metric": "encode GB/s on enwik9 at 1/2/4/8 MI355X; ratio vs ref; round-trip bit-exact",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{desc}; {n} enwik-shaped bytes per GPU",
                       "input_bytes_per_gpu": n, "compressed_bytes_rank0": int(c), "ratio": round(n / max(c, 1), 4),
                       "gather_to_rank0": bool(args.gather and world > 1), "parallelism": f"blocks sharded over {world} GPU(s)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
