#!/usr/bin/env python3
"""bench.py — encode throughput of the MI355X-native block-parallel compressor.

    python bench.py --gpus N --steps K --warmup W
                    [--workload deflate-h|deflate|lz77w16|lz77w14|huffman|fse] [--scaling weak|strong] [--bytes B]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU (RCCL).  A "step" is one pass of the hot path over one batch of synthetic
enwik-shaped input that is already resident in HBM.  Rank 0 prints ONE JSON line.

Default workload = BASELINE.json config 4 as it is worded, "deflate (LZ77 + dynamic Huffman) on enwik9, 32 KiB
window, 1 MI355X": 10^9 enwik-shaped bytes cut into 15 259 independent 64 KiB blocks; per block the reference
tokeniser (algorithms/deflate/lz77.c:199-280, fresh table per block, token sequence byte-exact) followed by the
per-block dynamic Huffman stage the reference leaves as a TODO (lz77.c:279; "mode H", include/mi_codec.h).  The same
line also carries
    mode_T        the token-only mode (the reference's own byte-token stream) timed the same way,
    shard_125MB   config 5's per-GPU shape (1/8 of enwik9 = 1 908 blocks) on this one GPU,
    ratio_vs_ref  own output size against the reference's shipped compress() output (persistent table,
                  deflate/deflate.c:47-63) on a sample,
    roundtrip     decode(encode(x)) == x, checked once outside the timed region,
    decode_gbps   this rank's decoder on its last stream (one call incl. its synchronisation, outside the timed region),
    end_to_end    the same bytes through the host-buffer entry point of the drop-in (PCIe inclusive; never `value`);
                  `decode_gbps` inside it = the way back through the host-buffer decoder, bytes of output per second.

N > 1, --scaling weak (default): every rank encodes its own --bytes shard of independent blocks.
N > 1, --scaling strong: ONE --bytes buffer (enwik9's size by default); rank r encodes the contiguous block range
sharded.shard_bytes gives it (config 5: 1 908 blocks / 125 MB per GPU at N = 8).
The north star's RCCL gather of the compressed streams to rank 0 (sharded.gather_streams: one all_gather of sizes, one
group of point-to-point transfers over xGMI) is INSIDE the timed region for --scaling strong (config 5 as worded: one
enwik9, 65 MB of mode-H stream per peer) and OUTSIDE it for the default weak scaling, where every rank owns a whole
enwik9-sized shard and the path itself has no data-path collective (each rank's stream is a finished product; gathering
8 x 0.5 GB into one GPU is not part of any BASELINE config): there it is run and timed once after the timed region and
reported as `gather`.  --gather / --no-gather force it in or out.
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
BLOCK = 65536
WORKLOADS = ["deflate-h", "deflate", "lz77w16", "lz77w14", "lz77w16-256k", "lz77w16-1m", "lz77old", "huffman", "fse"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="deflate-h", choices=WORKLOADS)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--bytes", type=int, default=1_000_000_000,
                    help="input bytes per GPU (weak) or of the whole job (strong)")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="N>1: gather the compressed streams to rank 0 inside the timed region (default for --scaling strong)")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="target duration of each leg of the headline's CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip mode_T / shard_125MB / end_to_end (A/B runs)")
    ap.add_argument("--multi-devices", default=None,
                    help="e.g. 0,1,2,3: ONE process, the C boundary's mi_lz_encode_multi_dev over these devices (config 5 through "
                         "csrc/multi.hip: contiguous block ranges, RCCL gather into the first device; a device may repeat). "
                         "Prints its own JSON line; --gpus N (one process per GPU) is what the driver launches")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices)")
    return ap.parse_args()


DESC = {
    "deflate-h": "deflate (LZ77 W=32 KiB len<=31 + per-block dynamic Huffman over the 286-symbol alphabet, mode H), independent 64 KiB blocks",
    "deflate": "deflate tokeniser alone (LZ77, W=32 KiB, len<=31, the reference's byte tokens, mode T), independent 64 KiB blocks",
    "lz77w16": "lz77 (W=64 KiB, len<=15, bit-packed), independent 64 KiB blocks",
    "lz77w14": "lz77 (W=16 KiB, len<=15, bit-packed), independent 64 KiB blocks",
    "lz77w16-256k": "lz77 (W=64 KiB, len<=15, bit-packed), independent 256 KiB blocks: the window slides (time-sliced finder, lzs.hip)",
    "lz77w16-1m": "lz77 (W=64 KiB, len<=15, bit-packed), independent 1 MiB blocks: the window slides (time-sliced finder, lzs.hip)",
    "lz77old": "lz77_compress_old (lz77.c:185-262): brute-force longest match of the whole 16 KiB window at every token, ONE stream over the buffer",
    "huffman": "whole-buffer Huffman, one tree",
    "fse": "FSE/tANS table_log 8, independent 64 KiB blocks x 64 sub-streams",
}


class Codec:
    """one workload over the C ABI: encode(x) -> handle, nbytes(handle), decode(handle) -> tensor"""

    def __init__(self, workload, ctx):
        from compression_algorithms_amd import fse, huffman, lz
        self.w, self.ctx = workload, ctx
        self.lz, self.huffman, self.fse = lz, huffman, fse
        self.p = {"deflate": lz.params("deflate"), "deflate-h": lz.params("deflate"), "lz77w16": lz.params("lz77", 16),
                  "lz77w14": lz.params("lz77", 14), "lz77w16-256k": lz.params("lz77", 16, 262144), "lz77w16-1m": lz.params("lz77", 16, 1 << 20),
                  "lz77old": None,
                  "fse": fse.params(), "huffman": None}[workload]

    def encode(self, x):
        if self.w == "deflate-h":
            return self.lz.compress_h(x, self.p, self.ctx)
        if self.w in ("deflate", "lz77w16", "lz77w14", "lz77w16-256k", "lz77w16-1m"):
            return self.lz.compress(x, self.p, self.ctx)
        if self.w == "huffman":
            return self.huffman.huffman_compress(x, self.ctx)
        if self.w == "lz77old":
            return self.lz.compress_old(x, 14, 4, self.ctx)
        return self.fse.compress(x, self.p, self.ctx)

    def nbytes(self, h):
        return (h.total_bits + 7) // 8 if self.w == "huffman" else h.nbytes

    def decode(self, h):
        if self.w == "deflate-h":
            return self.lz.decompress_h(h, self.ctx)
        if self.w in ("deflate", "lz77w16", "lz77w14", "lz77w16-256k", "lz77w16-1m"):
            return self.lz.decompress(h, self.ctx)
        if self.w == "huffman":
            return self.huffman.huffman_decompress(h, ctx=self.ctx)
        if self.w == "lz77old":
            return self.lz.decompress_whole(h, self.ctx)
        return self.fse.decompress(h, self.ctx)

    def stream_and_table(self, h):
        """(uint8 stream tensor, int64 block table in bits) for the gather; None for the whole-buffer Huffman"""
        if self.w in ("huffman", "lz77old"):
            return None
        return h.data, (h.block_bits if hasattr(h, "block_bits") else h.offsets)


def timed_steps(codec, x, steps, warmup, sync, after_step=None):
    """W untimed + K timed encode passes; the previous output is released first so that the caching allocator reuses
    its block (otherwise the first timed step pays a multi-GB hipMalloc).  Returns (seconds, last handle)."""
    holder = {"h": None}

    def step():
        holder["h"] = None
        holder["h"] = codec.encode(x)
        if after_step is not None:
            after_step(holder["h"])

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return time.perf_counter() - t0, holder["h"]


def cpu_encode(workload, sample):
    """one pass of the CPU path over `sample` (numpy uint8): the REFERENCE (oracle/_ref, compiled from /root/reference in
    the build container) where it has code for the workload, the oracle's restatement otherwise.  Checker code, never the
    product.  -> (kind, seconds, extra)"""
    import numpy as np
    from oracle import orc, ref
    n = len(sample)
    kind, extra = "port", {}
    have_ref = ref.available()
    if workload == "deflate":
        if have_ref:
            kind = "reference"
            rd = ref.RefDeflate()
            t0 = time.perf_counter()
            rd.stream(sample, independent=True)
        else:
            t0 = time.perf_counter()
            orc.deflate_stream(sample, BLOCK, True)
    elif workload == "deflate-h":
        # the reference stops at a TODO where the entropy stage would be (deflate/lz77.c:279): its tokeniser (the real
        # reference when oracle/_ref is there) + the oracle's restatement of the mode-H coder
        rd = ref.RefDeflate() if have_ref else None
        d = None if rd else orc.Deflate(BLOCK)
        kind = "reference" if rd else "port"
        t_tok = 0.0
        t0 = time.perf_counter()
        for at in range(0, n, BLOCK):
            ta = time.perf_counter()
            if rd:
                rd.fresh()
                tok = rd.block(sample[at:at + BLOCK])
            else:
                d.fresh()
                tok = d.block_encode(sample[at:at + BLOCK])
            t_tok += time.perf_counter() - ta
            orc.defh_encode_block(tok)
        extra = {"tokeniser_only_gbs": round(n / max(t_tok, 1e-9) / 1e9, 5),
                 "note": "tokeniser = " + ("the compiled reference" if rd else "oracle port") +
                         "; entropy stage = oracle port (the reference has none)"}
    elif workload == "lz77old":
        # O(n * 2^14) by definition (lz77.c:205-241): the sample is small
        if have_ref:
            kind = "reference"
            t0 = time.perf_counter()
            ref.lz77_compress_old(sample, 14)
        else:
            t0 = time.perf_counter()
            orc.lz77_old_encode(sample, 14, 4)
    elif workload.startswith("lz77w"):
        wb = 14 if workload == "lz77w14" else 16
        blk = {"lz77w16-256k": 262144, "lz77w16-1m": 1 << 20}.get(workload, BLOCK)
        # the reference's insert probes without wrapping (lz77.c:61): on an input where a word with a home just below the
        # table's end repeats, the compiled reference writes past its bucket array (heap corruption).
        # The oracle tells beforehand (untimed); such a sample is timed with the oracle port instead, and says so.
        safe = have_ref and not orc.past_table_end(sample, wb, wb + 6, False)
        if have_ref and not safe:
            extra = {"note": "the compiled reference would write past its table on this sample (lz77.c:61 UB): oracle port timed instead"}
        if safe:
            # lz77_compress (lz77/lz77.c:264-345) as the reference runs it: ONE call over the sample, one stream — its
            # 2^(W+6)-bucket table is allocated and cleared once per call (lz77.c:43-53), so per-block calls would mostly
            # time that fill; the block-parallel streams this build emits are those of per-block calls (parity tests)
            kind = "reference"
            extra = {"note": "reference lz77_compress over the sample as one stream (one table fill per call)"}
            t0 = time.perf_counter()
            ref.lz77_compress(sample, wb)
        else:
            t0 = time.perf_counter()
            for at in range(0, n, blk):
                orc.lz77_encode(sample[at:at + blk], wb, 4)
    elif workload == "huffman":
        if have_ref:
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
        extra = {"note": "the reference's FSE does not compile (fse/src/main.zig:47): oracle port of this build's format"}
        t0 = time.perf_counter()
        for at in range(0, n, BLOCK):
            orc.fse_encode_block(sample[at:at + BLOCK], 8, 64, 1)
    else:
        raise ValueError(workload)
    return kind, time.perf_counter() - t0, extra


def _cpu_worker(job):
    """all-cores leg: one independent-block worker per core; runs in a process forked BEFORE the parent
    touched the GPU.  The sample comes through a file in /dev/shm."""
    import numpy as np
    workload, path, lo, hi, reps = job
    a = np.array(np.memmap(path, dtype=np.uint8, mode="r")[lo:hi])
    dt = 0.0
    for _ in range(reps):
        dt += cpu_encode(workload, a)[1]
    return (hi - lo) * reps, dt


def host_cores():
    """cores this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(int(t[0]) / int(t[1]))))
            elif int(t[0]) > 0:
                n = min(n, max(1, int(int(t[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except Exception:
            continue
    env = os.environ.get("MI_BENCH_CPU_WORKERS")
    return max(1, min(n, int(env))) if env else n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


class CpuPool:
    """worker processes for the all-cores CPU baseline, forked at the top of main() before any torch.cuda call (a forked
    child of a process that has initialised HIP is not usable, and a late fork is what the box's exec guard forbids)"""

    def __init__(self):
        import multiprocessing as mp
        self.cores = host_cores()
        self.pool = mp.get_context("fork").Pool(self.cores) if self.cores > 1 else None
        self.model = cpu_model()

    def run(self, workload, sample, per_core_bytes, grain, want_bytes):
        """every core encodes its own `per_core_bytes` slice of `sample` (whole blocks), as often as it takes to have done
        about `want_bytes`; a first tiny round loads the libraries in every worker; -> dict or None"""
        import numpy as np
        if self.pool is None:
            return None
        per = max(grain, per_core_bytes // grain * grain)
        k = min(self.cores, len(sample) // per)
        if k < 2:
            return None
        path = f"/dev/shm/mi_bench_{os.getpid()}.bin"
        try:
            np.asarray(sample[: k * per]).tofile(path)
            reps = max(1, -(-want_bytes // per))
            self.pool.map(_cpu_worker, [(workload, path, 0, min(grain, per), 1)] * k, chunksize=1)      # warm-up
            jobs = [(workload, path, i * per, (i + 1) * per, reps) for i in range(k)]
            t0 = time.perf_counter()
            res = self.pool.map(_cpu_worker, jobs, chunksize=1)
            wall = time.perf_counter() - t0
        finally:
            try:
                os.unlink(path)
            except OSError:
                pass
        nb = sum(r[0] for r in res)
        return {"value": round(nb / wall / 1e9, 4), "unit": "GB/s", "cores": k, "host_cores": os.cpu_count(), "cpu_model": self.model,
                "sample": f"{k} workers x {reps} pass(es) over {per} bytes each (independent blocks, one worker per core), wall {wall:.2f} s, "
                          f"slowest worker {max(r[1] for r in res):.2f} s"}

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None


# single-core CPU rates (GB/s) used only to size the samples so that each leg takes ~2-3 s
CPU_RATE = {"lz77old": 0.0004, "deflate": 0.045, "deflate-h": 0.029, "lz77w16": 0.012, "lz77w14": 0.014, "lz77w16-256k": 0.012, "lz77w16-1m": 0.012,
            "huffman": 0.23, "fse": 0.04}


def cpu_baseline(workload, x, pool, seconds=2.5):
    """cpu_baseline object of the JSON line for one workload: single core + all cores, on a bounded prefix of `x`"""
    n = x.numel()
    grain = {"lz77w16-256k": 262144, "lz77w16-1m": 1 << 20}.get(workload, BLOCK)
    want = int(CPU_RATE[workload] * 1e9 * seconds)
    nsamp = max(grain, min(n, want) // grain * grain) if n >= grain else n
    cores = pool.cores if pool is not None else 1
    per_core = nsamp if workload != "huffman" else max(grain, nsamp // 4 // grain * grain)
    nall = min(n, per_core * cores) // grain * grain
    sample = x[: max(nsamp, nall)].cpu().numpy()
    kind, dt, extra = cpu_encode(workload, sample[:nsamp])
    out = {"value": round(nsamp / dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": kind,
           "sample": f"first {nsamp} bytes of the rank-0 buffer in {dt:.2f} s on 1 of {os.cpu_count()} host cores"}
    out.update(extra)
    if pool is not None:
        try:
            out["all_cores"] = pool.run(workload, sample, min(per_core, max(grain, nall // max(cores, 1))), grain, per_core)
        except Exception as e:
            out["all_cores"] = {"value": None, "error": repr(e)[:200]}
    return out, sample[:nsamp]


def reference_output_bytes(workload, sample):
    """size of what the REFERENCE writes for `sample` (the denominator of ratio_vs_ref), and what it is"""
    from oracle import orc, ref
    if workload in ("deflate", "deflate-h"):
        # the shipped compress(): 64 KiB chunks, ONE table kept across them (deflate/deflate.c:13-14,47-63), raw byte tokens
        if ref.available():
            s, _ = ref.RefDeflate().stream(sample, independent=False)
            return len(s), "reference compress(): persistent-table byte-token stream (deflate/deflate.c:47-63), compiled reference"
        s, _ = orc.deflate_stream(sample, BLOCK, False)
        return len(s), "reference compress(): persistent-table byte-token stream (deflate/deflate.c:47-63), oracle port"
    if workload.startswith("lz77w"):
        wb = 14 if workload == "lz77w14" else 16
        if ref.available():
            s, nb = ref.lz77_compress(sample, wb)
        else:
            s, nb = orc.lz77_encode(sample.tobytes(), wb, 4)
        return (nb + 7) // 8, f"reference lz77_compress over the whole sample as ONE stream, W=2^{wb} (lz77/lz77.c:264-345)"
    if workload == "huffman":
        return (orc.huff_encode(sample)["bits"] + 7) // 8, "reference huffman_compress, one tree (identical stream by construction)"
    return None, "the reference's FSE does not compile: no reference output exists"


def roofline_of(workload, n, c, ktimes, steps, dt):
    """roofline object for one measured workload: the dominant kernel (largest share of the timed region, HIP events on
    its launch stream) is credited the job's ALGORITHMIC bytes — (passes*n + c), DESIGN.md section 3 — split over its
    launches.  `traffic` is HBM bytes per launch from a committed rocprofv3 --pmc run of the same command
    (profiles/pmc_traffic_<workload>.json: `traffic_source`), not a measurement of this run."""
    if not ktimes:
        return None
    dom = max(ktimes, key=lambda k: k["ms"] * k["launches"])
    latency_bound = None
    if workload == "huffman" and dom["name"] == "k_huff_build":
        # whole-buffer Huffman at 10^8 B: the longest kernel is ONE lane's heap over 256 cells (17 KB of traffic) — crediting it
        # the job's bytes says nothing.  The roofline names the longest kernel that STREAMS the buffer and
        # credits it its own bytes; the heap kernel is reported beside it as what it is, a latency-bound serial section.
        latency_bound = {"kernel": dom["name"], "ms_per_step": round(dom["ms"] * dom["launches"] / steps, 4),
                         "share": round(dom["ms"] * dom["launches"] / (dt * 1e3), 3),
                         "what": "one lane's heap over the 256-bin histogram (the reference's tie order defines the codes): latency, not bandwidth"}
        streaming = [k for k in ktimes if k["name"] in ("k_huff_hist", "k_huff_encode")]
        if streaming:
            dom = max(streaming, key=lambda k: k["ms"] * k["launches"])
    lps = dom["launches"] / steps
    passes = 2 if workload == "huffman" else 1
    alg = (passes * n + c) / lps
    if latency_bound is not None:
        alg = {"k_huff_hist": n, "k_huff_encode": n + c}[dom["name"]] / lps
    achieved = alg / (dom["ms"] * 1e-3) / 1e9
    traffic, src = None, None
    tj = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    if os.path.exists(tj):
        try:
            traffic = json.load(open(tj)).get(dom["name"])
            src = f"profiles/pmc_traffic_{workload}.json (committed rocprofv3 --pmc run, not this run)" if traffic is not None else None
        except Exception:
            traffic = None
    own = {"k_huff_hist": n, "k_huff_encode": n + c}.get(dom["name"], (passes * n + c)) / lps
    return {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "achieved_own_bytes": round(own / (dom["ms"] * 1e-3) / 1e9, 2),
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes_per_launch": int(alg),
            "whole_step_frac": round((passes * n + c) / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5),
            "avg_launch_ms": round(dom["ms"], 4), "launches_per_step": lps,
            "kernel_share": round(dom["ms"] * dom["launches"] / (dt * 1e3), 3),
            "all_kernels_ms_per_step": {k["name"]: round(k["ms"] * k["launches"] / steps, 3) for k in ktimes},
            **({"latency_bound": latency_bound} if latency_bound else {})}


def measure(ctx, workload, x, steps, warmup, pool=None, cpu=True, cold=False):
    """one sub-record: K timed encode steps of `workload` over the resident buffer x (after `warmup` untimed ones), kernel
    times from HIP events, one decode for the round trip, roofline, CPU baseline.  cold=True: no warm-up at all — the timed
    step is this input's first contact with the context (the fallback grid hint is whatever the call before left)"""
    co = Codec(workload, ctx)
    n = x.numel()
    if not cold:
        timed_steps(co, x, 0, max(warmup, 1), torch.cuda.synchronize)
    ctx.set_profiling(True)
    ctx.kernel_times()
    dt, h = timed_steps(co, x, steps, 0, torch.cuda.synchronize)
    ctx.set_profiling(False)
    kt = ctx.kernel_times()
    c = co.nbytes(h)
    ok = bool(torch.equal(co.decode(h), x))
    del h
    rec = {"value": round(n * steps / dt / 1e9, 3), "unit": "GB/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "input_bytes": n, "compressed_bytes": int(c), "ratio": round(n / max(c, 1), 4), "roundtrip": ok, "what": DESC[workload],
           "roofline": roofline_of(workload, n, c, kt, steps, dt)}
    if cpu:
        try:
            rec["cpu_baseline"], _ = cpu_baseline(workload, x, pool)
        except Exception as e:
            rec["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
    return rec


def main_multi_devices(args):
    """--multi-devices: config 5 from the C boundary in ONE process (include/mi_codec.h mi_multi_*).  Strong scaling by
    construction: ONE buffer of --bytes, device g holds and encodes its contiguous block range, the streams are gathered into the
    first device inside the timed region.  Not the driver's line (that is --gpus N)."""
    from compression_algorithms_amd import synth, lz
    from compression_algorithms_amd.multi import Multi
    devs = [int(d) for d in args.multi_devices.split(",")]
    assert args.workload in ("deflate-h", "deflate"), "--multi-devices: the deflate workloads"
    mm = Multi(devs)
    p = lz.params("deflate")
    n = args.bytes
    nblk = (n + BLOCK - 1) // BLOCK
    whole = synth.enwik_like(n, seed=args.seed, device=torch.device("cuda", devs[0]))
    shards = []
    for g, d in enumerate(devs):
        lo, hi = mm.shard(nblk, g)
        shards.append(whole[lo * BLOCK: min(hi * BLOCK, n)].to(torch.device("cuda", d)).clone() if hi > lo else None)
    mode_h = args.workload == "deflate-h"
    for _ in range(max(args.warmup, 1)):
        h = mm.compress_dev(shards, n, p, mode_h=mode_h)
    for d in set(devs):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h = None
        h = mm.compress_dev(shards, n, p, mode_h=mode_h)           # returns synchronised (the gather included)
    dt = time.perf_counter() - t0
    torch.cuda.set_device(devs[0])
    back = lz.decompress_h(h) if mode_h else lz.decompress(h)
    ok = bool(torch.equal(back, whole))
    print(json.dumps({"metric": "encode GB/s on enwik9 at 1/2/4/8 MI355X; ratio vs ref; round-trip bit-exact", "value": round(n * args.steps / dt / 1e9, 3),
                      "unit": "GB/s", "n_gpus": len(set(devs)), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                      "config": {"workload": f"{DESC[args.workload]}; {n} enwik-shaped bytes, ONE process over devices {devs} "
                                             f"(mi_lz_encode_multi_dev, transport {mm.transport}), gather to device {devs[0]} inside the timed region",
                                 "input_sha256": synth.digest(whole), "compressed_bytes_job": int(h.nbytes), "ratio": round(n / max(h.nbytes, 1), 4)},
                      "roundtrip": ok}), flush=True)
    mm.close()
    if not ok:
        sys.exit(3)


def main():
    args = parse_args()
    if args.multi_devices:
        return main_multi_devices(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # the all-cores CPU baseline needs worker processes: fork them NOW, before anything touches the GPU
    pool = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            pool = CpuPool()
        except Exception:
            pool = None
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

    from compression_algorithms_amd import sharded, synth
    from compression_algorithms_amd.context import Context
    ctx = Context(local)
    codec = Codec(args.workload, ctx)
    can_gather = world > 1 and args.workload != "huffman"
    gather = (can_gather and args.scaling == "strong") if args.gather is None else (args.gather and can_gather)

    if args.scaling == "strong":
        # ONE corpus of --bytes; rank r owns the contiguous block range of sharded.shard_bytes (config 5's shape)
        n_job = args.bytes
        lo, hi = sharded.shard_bytes(n_job, BLOCK, rank, world)
        whole = synth.enwik_like(n_job, seed=args.seed, device=dev)
        x = whole[lo:hi].clone()
        del whole
        torch.cuda.empty_cache()
    else:
        # every rank owns a different shard of the same kind of corpus (independent blocks)
        n_job = args.bytes * world
        x = synth.enwik_like(args.bytes, seed=args.seed + 1000 * rank, device=dev)
    n = x.numel()
    torch.cuda.synchronize()
    # digest of the rank-0 buffer: the generator is byte-reproducible (synth.py), so this field is too
    input_digest = synth.digest(x) if rank == 0 else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def gather_streams(h):
        """north star: RCCL gather of the per-block compressed streams to rank 0 (sizes and block tables first, then the
        variable-length streams point to point in one group: sharded.gather_streams)"""
        st = codec.stream_and_table(h)
        if st is None:
            return
        sharded.gather_streams(st[0], st[1], dst=0)

    after = gather_streams if gather else None
    for _ in range(args.warmup):
        h = codec.encode(x)
        if after:
            after(h)
        h = None
    barrier()
    ctx.set_profiling(True)
    ctx.kernel_times()              # reset
    dt, last = timed_steps(codec, x, args.steps, 0, barrier, after)
    ctx.set_profiling(False)
    ktimes = ctx.kernel_times()
    c = codec.nbytes(last)
    # what the timed steps met (mi_lz_path_stats / mi_order_violations): blocks on the fallback pipeline, parts above 2 560
    # entries, sorts found out of order — all 0 on text; a corpus that lives in the fallback would otherwise only be a slow number
    path = ctx.path_stats()
    path["order_violations"] = ctx.order_violations()

    t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- outside the timed region (weak scaling): the RCCL gather of the streams to rank 0, timed on its own
    gather_report = None
    if can_gather and not gather and args.gather is None:
        try:
            gather_streams(last)                      # warm-up: communicators, receive buffers
            barrier()
            tg0 = time.perf_counter()
            gather_streams(last)
            barrier()
            tg = time.perf_counter() - tg0
            tt = torch.tensor([tg], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            gather_report = {"ms": round(float(tt.item()) * 1e3, 3), "what": "one gather of every rank's last stream + block table to rank 0 "
                             "(all_gather of sizes, one group of point-to-point transfers), outside the timed region"}
        except Exception as e:                        # reported, never fatal for the throughput line
            gather_report = {"ms": None, "error": repr(e)[:200]}
    # ---- outside the timed region: round trip of the last output on every rank
    back = codec.decode(last)
    rt_ok = bool(torch.equal(back, x))
    del back
    torch.cuda.synchronize()
    td0 = time.perf_counter()
    back = codec.decode(last)                         # the decoders synchronise their stream themselves (status word)
    torch.cuda.synchronize()
    decode_gbps = round(n / (time.perf_counter() - td0) / 1e9, 2)
    del back
    if dist is not None:
        f = torch.tensor([1 if rt_ok else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        rt_ok = bool(int(f.item()))
    csum = torch.tensor([c], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(csum, op=dist.ReduceOp.SUM)
    c_job = int(csum.item())

    if rank == 0:
        value = n_job * args.steps / dt / 1e9
        roof = roofline_of(args.workload, n, c, ktimes, args.steps, dt)

        extras = {}
        single = world == 1 and not args.no_extras
        cpu_on = not args.no_cpu_baseline
        if single and args.workload == "deflate-h":
            # the token-only mode, timed the same way (what BENCH_r01 carried as its headline; a launch is credited
            # n + c = 2.0008 n here where mode H is credited n + 0.52 n)
            extras["mode_T"] = measure(ctx, "deflate", x, args.steps, 1, pool, cpu_on)
        if single and args.workload in ("deflate-h", "deflate") and n >= 125_000_000 + BLOCK:
            # config 5's per-GPU shape on this one GPU: 1/8 of enwik9 = 125 MB = 1 908 blocks
            xs = x[:125_000_000]
            k = max(args.steps, 10)
            dts, hs = timed_steps(codec, xs, k, 2, torch.cuda.synchronize)
            extras["shard_125MB"] = {"value": round(xs.numel() * k / dts / 1e9, 3), "unit": "GB/s", "ms_per_step": round(dts / k * 1e3, 3),
                                     "blocks": (xs.numel() + BLOCK - 1) // BLOCK, "steps": k,
                                     "what": "one GPU's share of config 5 (enwik9 / 8), same workload, single-GPU proxy"}
            del hs
        if single and args.workload == "deflate-h" and n >= 1_000_000_000:
            # the other BASELINE.json configs on the same box, same method (inputs resident, K steps after a warm-up), each
            # with its own roofline and CPU baseline, so that the driver's record backs every figure the documents quote:
            # config 1's codec (whole-buffer Huffman, 10^8 B), config 2 (lz77, 10^8 B: W = 64 KiB on 64 KiB / 256 KiB /
            # 1 MiB blocks and the shipped W = 16 KiB), config 3 (FSE table_log 8, 10^9 B)
            others = {}
            for key, wl, nb_ in (("config1_huffman_1e8", "huffman", 100_000_000), ("config2_lz77_w16_1e8", "lz77w16", 100_000_000),
                                 ("config2_lz77_w14_1e8", "lz77w14", 100_000_000),
                                 ("config2_lz77_w16_256KiB_blocks_1e8", "lz77w16-256k", 100_000_000),
                                 ("config2_lz77_w16_1MiB_blocks_1e8", "lz77w16-1m", 100_000_000),
                                 ("config3_fse_1e9", "fse", 1_000_000_000), ("lz77_compress_old_w14_2e7", "lz77old", 20_000_000)):
                try:
                    others[key] = measure(ctx, wl, x[:nb_], max(args.steps, 5), 1, pool, cpu_on)
                except Exception as e:
                    others[key] = {"value": None, "error": repr(e)[:200]}
            extras["other_configs"] = others
            # inputs that are not text (SURVEY.md 8d "adversarial"; the fuzz families of tests/test_fuzz_gpu.py): 10^8 bytes
            # each = a seeded 2^24-byte sample repeated (blocks are independent and the window is 32 KiB, so the repeat
            # changes nothing per block), through the headline workload; value + round trip
            adv = {}
            import numpy as np
            reps = 6
            kinds = [(k, lambda k=k: synth.family(k, 4242, 1 << 24)) for k in synth.FAMILIES if k != "text"]
            kinds += [("zeros", lambda: np.zeros(1 << 24, np.uint8)), ("single_symbol", lambda: np.full(1 << 24, 0x61, np.uint8)),
                      ("random", lambda: np.random.default_rng(1).integers(0, 256, 1 << 24, dtype=np.uint8))]
            for kind, gen in kinds:
                try:
                    xa = torch.from_numpy(gen()).to(dev).repeat(reps)[:100_000_000].contiguous()
                    # cold first: one text call resets the fallback hint (and sheds the second fallback stream), then ONE timed
                    # step with no warm-up — what a caller sees who switches from text to this family; then warm
                    codec.encode(x[:100_000_000]); torch.cuda.synchronize()
                    ps0 = ctx.path_stats()
                    rc_ = measure(ctx, args.workload, xa, 1, 0, None, False, cold=True)
                    ps1 = ctx.path_stats()
                    r = measure(ctx, args.workload, xa, 2, 1, None, False)
                    adv[kind] = {k_: r[k_] for k_ in ("value", "unit", "ms_per_step", "ratio", "roundtrip", "input_bytes")}
                    adv[kind]["value_cold"] = rc_["value"]
                    nblk_ = (xa.numel() + BLOCK - 1) // BLOCK
                    adv[kind]["fallback_blocks_per_pass"] = ps1["fallback_blocks"] - ps0["fallback_blocks"]
                    adv[kind]["wide_parts_per_pass"] = ps1["wide_parts"] - ps0["wide_parts"]
                    adv[kind]["blocks"] = nblk_
                    # the same family at the headline's size (15 259 blocks: five batches in the three-stage pipeline, the fallback
                    # grids sized by the batch before; 10^8 bytes are two batches, the first of them on the small fallback grid)
                    xb = xa.repeat(10)
                    rb = measure(ctx, args.workload, xb, 2, 1, None, False)
                    adv[kind]["at_1e9_bytes"] = {k_: rb[k_] for k_ in ("value", "ms_per_step", "roundtrip", "input_bytes")}
                    del xa, xb
                except Exception as e:
                    adv[kind] = {"value": None, "error": repr(e)[:200]}
            extras["adversarial"] = adv
        if single and args.workload in ("deflate-h", "deflate"):
            # config 5 from the C boundary (mi_lz_encode_multi_dev, csrc/multi.hip): ONE process, one context per listed device,
            # contiguous block ranges, streams gathered into the first device.  This box has one GPU, so the list is {0, 0}: two
            # contexts share the GPU and the gather is a peer copy — a proof that the path runs and what it costs, not a scaling
            # figure (the driver's --gpus N run is one process per GPU over torch.distributed / RCCL)
            try:
                from compression_algorithms_amd.multi import Multi
                mm = Multi([local, local])
                p_ = codec.p
                nblk = (n + BLOCK - 1) // BLOCK
                sh = []
                for g in range(2):
                    lo_, hi_ = mm.shard(nblk, g)
                    sh.append(x[lo_ * BLOCK: min(hi_ * BLOCK, n)])
                mode_h = args.workload == "deflate-h"
                hm = mm.compress_dev(sh, n, p_, mode_h=mode_h)          # warm-up: workspaces, buffers
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    hm = None
                    hm = mm.compress_dev(sh, n, p_, mode_h=mode_h)
                tm = (time.perf_counter() - t0) / 3
                same = bool(hm.nbytes == c and torch.equal(hm.data[:c], last.data[:c]) and torch.equal(hm.block_bits, last.block_bits))
                extras["multi_one_process"] = {"value": round(n / tm / 1e9, 3), "unit": "GB/s", "devices": [local, local], "transport": mm.transport,
                                               "equal_to_single_context": same,
                                               "what": "mi_lz_encode_multi_dev: two contexts on this one GPU, halves encoded side by side, gathered "
                                                       "by peer copy; call returns synchronised"}
                del hm
                mm.close()
            except Exception as e:
                extras["multi_one_process"] = {"value": None, "error": repr(e)[:300]}
        if single:
            # PCIe-inclusive (never `value`).  The deflate workloads go through the HOST-buffer entry points the drop-in
            # compress() calls (pageable memory in and out, transfers chunked beside the encoder: host_api.hip); the others
            # through a pinned buffer and one copy each way.
            try:
                if args.workload in ("deflate-h", "deflate"):
                    import ctypes as C
                    import numpy as np
                    from compression_algorithms_amd import lz as lzmod
                    xh = x.cpu().numpy()
                    p_ = codec.p
                    ctx.L.mi_deflate_h_bound_bytes.restype = C.c_uint64
                    mode_h = args.workload == "deflate-h"
                    cap = (int(ctx.L.mi_deflate_h_bound_bytes(C.c_uint64(n), C.byref(p_))) if mode_h else lzmod.bound_bytes(n, p_)) + 64
                    outh = np.empty(cap, np.uint8)
                    bits = np.zeros((n + BLOCK - 1) // BLOCK + 1, np.uint64)
                    fn = ctx.L.mi_deflate_h_encode if mode_h else ctx.L.mi_lz_encode
                    t_e2e = []
                    for it in range(3):
                        t0 = time.perf_counter()
                        rc = fn(ctx.h, C.byref(p_), C.c_void_p(xh.ctypes.data), C.c_uint64(n), C.c_void_p(outh.ctypes.data), C.c_uint64(cap), C.c_void_p(bits.ctypes.data))
                        t_e2e.append(time.perf_counter() - t0)
                        if rc != 0:
                            raise RuntimeError(f"host entry point returned {rc}")
                    if int(bits[-1]) // 8 != c:
                        raise RuntimeError("host entry point produced a different stream size")
                    extras["end_to_end"] = {"value": round(n / min(t_e2e[1:]) / 1e9, 3), "unit": "GB/s", "bytes": n,
                                            "what": "pageable host buffer -> mi_deflate_h_encode / mi_lz_encode (the drop-in's entry point: chunked H2D, encode, "
                                                    "chunked D2H of stream and block table) -> host buffer; best of 2 after 1 warm-up"}
                    # and the way back through the drop-in's decompress() entry point: host stream -> host bytes
                    dfn = ctx.L.mi_deflate_h_decode if mode_h else ctx.L.mi_lz_decode
                    back = np.empty(n, np.uint8)
                    t_d = []
                    for it in range(3):
                        t0 = time.perf_counter()
                        rc = dfn(ctx.h, C.byref(p_), C.c_void_p(outh.ctypes.data), C.c_uint64(int(bits[-1]) // 8), C.c_void_p(bits.ctypes.data),
                                 C.c_void_p(back.ctypes.data), C.c_uint64(n))
                        t_d.append(time.perf_counter() - t0)
                        if rc != 0:
                            raise RuntimeError(f"host decode entry point returned {rc}")
                    if not np.array_equal(back, xh):
                        raise RuntimeError("host decode entry point returned different bytes")
                    extras["end_to_end"]["decode_gbps"] = round(n / min(t_d[1:]) / 1e9, 3)
                    del xh, outh, back
                else:
                    ne = min(n, 256_000_000)
                    xh = x[:ne].cpu().pin_memory()
                    xd = torch.empty(ne, dtype=torch.uint8, device=dev)
                    outh = None
                    t_e2e = []
                    for it in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        xd.copy_(xh, non_blocking=True)
                        he = codec.encode(xd)
                        nb = codec.nbytes(he)
                        st = codec.stream_and_table(he)
                        if st is not None:
                            if outh is None or outh.numel() < nb:
                                outh = torch.empty(int(nb * 1.1) + 64, dtype=torch.uint8).pin_memory()
                            outh[:nb].copy_(st[0][:nb], non_blocking=True)
                            st[1].cpu()
                        else:
                            he.words.cpu()
                        torch.cuda.synchronize()
                        t_e2e.append(time.perf_counter() - t0)
                        he = None
                    extras["end_to_end"] = {"value": round(ne / min(t_e2e[1:]) / 1e9, 3), "unit": "GB/s", "bytes": ne,
                                            "what": "pinned host input -> H2D -> encode -> D2H of the stream and block table; best of 2 after 1 warm-up"}
            except Exception as e:       # pinned allocation refused etc.: the column is optional
                extras["end_to_end"] = {"value": None, "error": repr(e)[:200]}

        cpu = None
        ratio_vs_ref = None
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is a rank-0, N = 1 measurement
            cpu, sample = cpu_baseline(args.workload, x, pool, seconds=args.cpu_seconds)
            # ratio vs the reference's own output on (a prefix of) the same sample
            nr = min(len(sample), 64 * 1024 * 1024)
            try:
                ref_bytes, ref_what = reference_output_bytes(args.workload, sample[:nr])
                own = codec.nbytes(codec.encode(x[:nr]))
                if ref_bytes:
                    ratio_vs_ref = {"value": round(ref_bytes / max(own, 1), 4), "own_bytes": int(own), "ref_bytes": int(ref_bytes),
                                    "sample_bytes": int(nr), "ref": ref_what,
                                    "meaning": "(input/own) / (input/ref) = ref_bytes / own_bytes; > 1: smaller than the reference's output"}
                else:
                    ratio_vs_ref = {"value": None, "ref": ref_what}
            except Exception as e:
                ratio_vs_ref = {"value": None, "error": repr(e)[:200]}
        if pool is not None:
            pool.close()
        line = {
            "metric": "encode GB/s on enwik9 at 1/2/4/8 MI355X; ratio vs ref; round-trip bit-exact",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{DESC[args.workload]}; {n_job} enwik-shaped bytes in the job, {n} on rank 0",
                       "input_sha256": input_digest, "input_seed": args.seed,
                       "mode": {"deflate-h": "H", "deflate": "T"}.get(args.workload),
                       "input_bytes_job": n_job, "input_bytes_rank0": n, "compressed_bytes_job": c_job, "compressed_bytes_rank0": int(c),
                       "ratio": round(n_job / max(c_job, 1), 4), "gather_to_rank0": bool(gather),
                       "parallelism": f"blocks sharded over {world} GPU(s), {args.scaling} scaling"},
            "ratio": round(n_job / max(c_job, 1), 4), "ratio_vs_ref": ratio_vs_ref, "roundtrip": rt_ok, "decode_gbps": decode_gbps, "gather": gather_report,
            "roofline": roof, "cpu_baseline": cpu,
            "fallback_blocks": path["fallback_blocks"], "wide_parts": path["wide_parts"], "order_violations": path["order_violations"],
        }
        line.update(extras)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not rt_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
