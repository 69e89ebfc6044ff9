"""Synthetic enwik-shaped byte buffers.

The reference's data step (get_data.sh:1-9) downloads enwik9 and derives enwik8/7/6 with
`head -c`; there is no network here, so benches and tests run on a seeded stand-in with the
same gross structure: a Zipf-distributed vocabulary of pseudo-words
wrapped in the MediaWiki XML page/revision skeleton with [[links]], '''bold''', &quot;
entities, == headings == and punctuation.  Like the reference's derived files, a shorter
buffer is a prefix of a longer one (same seed).

Everything is tensor ops, so a 10^9-byte buffer is produced directly in HBM in well under a
second; on CPU the same code serves the small test inputs.

Reproducible to the byte: the vocabulary is built once on the CPU; every per-slot draw is a
counter-based integer hash of (seed, stream, global slot number) compared against INTEGER
thresholds (the Zipf CDF is summed in float64 on the CPU, scaled to 53-bit integers and looked up
with searchsorted) — no device floating point, no device random generator, no reduction whose
order could differ.  One seed therefore gives the same bytes on every run, on the CPU and on the
GPU alike, whatever `slots_per_chunk` is (tests/test_synth.py, tests/test_determinism_gpu.py).
"""
import math

import numpy as np
import torch

_U53 = 1 << 53

_LETTERS = "etaoinshrdlcumwfgypbvkjxqz"
_V = 20000

_HDR_A = b"</text>\n    </revision>\n  </page>\n  <page>\n    <title>"
_HDR_B = b"</title>\n    <id>"
_HDR_C = b"</id>\n    <revision>\n      <id>"
_HDR_D = b"</id>\n      <timestamp>2006-03-"
_HDR_E = b"</timestamp>\n      <contributor>\n        <username>"
_HDR_F = b"</username>\n        <id>"
_HDR_G = b"</id>\n      </contributor>\n      <text xml:space=\"preserve\">"


class _Tables:
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        w = torch.tensor([3.0 ** (-i * 0.12) for i in range(26)])
        lens = torch.clamp((torch.randn(_V, generator=g) * 2.5 + 5.5).floor().long(), min=1, max=18)
        letters = torch.multinomial(w, int(lens.sum()), replacement=True, generator=g)
        flat = [(_LETTERS[i]) for i in letters.tolist()]
        words, at = [], 0
        for n in lens.tolist():
            words.append("".join(flat[at:at + n]).encode())
            at += n
        toks = []
        toks += [x + b" " for x in words]                                  # 0: plain
        toks += [b"[[" + x + b"]] " for x in words]                        # 1: link
        toks += [b"'''" + x + b"''' " for x in words]                      # 2: bold
        toks += [x + b".\n\n" for x in words]                              # 3: paragraph end
        toks += [x + b", " for x in words]                                 # 4: comma
        toks += [x.capitalize() + b" " for x in words]                     # 5: capitalised
        toks += [b"&quot;" + x + b"&quot; " for x in words]                # 6: quoted
        toks += [b"\n== " + x.title() + b" ==\n" for x in words]           # 7: heading
        toks += [x.title() for x in words]                                 # 8: bare title word
        self.n_variants = 9
        self.num_base = len(toks)
        nums = torch.randint(1, 40_000_000, (4096,), generator=g).tolist()
        toks += [str(v).encode() for v in nums]
        self.ts_base = len(toks)
        ts = torch.randint(0, 28 * 86400, (4096,), generator=g).tolist()
        toks += [("%02dT%02d:%02d:%02dZ" % (v // 86400 + 1, v // 3600 % 24, v // 60 % 60, v % 60)).encode() for v in ts]
        self.fixed_base = len(toks)
        toks += [_HDR_A, _HDR_B, _HDR_C, _HDR_D, _HDR_E, _HDR_F, _HDR_G]
        self.tok_len = torch.tensor([len(t) for t in toks], dtype=torch.int64)
        self.tok_start = torch.cumsum(self.tok_len, 0) - self.tok_len
        self.flat = torch.frombuffer(bytearray(b"".join(toks)), dtype=torch.uint8).clone()
        # Zipf (1/rank) CDF, summed in float64 on the CPU, as 53-bit integer thresholds
        cdf = np.cumsum(1.0 / np.arange(1, _V + 1, dtype=np.float64))
        self.zipf_thr = torch.from_numpy(np.floor(cdf / cdf[-1] * float(_U53)).astype(np.int64))
        # cumulative thresholds of the decoration variants (SURVEY.md Appendix C proportions)
        self.var_edges = torch.tensor([int(e * _U53) for e in (0.04, 0.05, 0.06, 0.12, 0.17, 0.175, 0.18)], dtype=torch.int64)
        self.var_ids = torch.tensor([1, 2, 3, 4, 5, 6, 7, 0])
        self.hdr_thr = _U53 // 1500


_TAB = {}


def _tables(seed, device):
    key = (seed, str(device))
    if key not in _TAB:
        t = _Tables(seed)
        for name in ("tok_len", "tok_start", "flat", "zipf_thr", "var_edges", "var_ids"):
            setattr(t, name, getattr(t, name).to(device))
        _TAB[key] = t
    return _TAB[key]


def _s64(v):
    """python int -> the same 64 bits as a signed value (torch has no uint64 arithmetic)"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def _lsr(x, k):
    """logical right shift of an int64 tensor"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _u53(seed, stream, idx):
    """counter-based draw: 53 uniform bits per element of the int64 tensor `idx` (splitmix64's finaliser over
    seed / stream / counter; int64 products wrap identically on the CPU and on the device)"""
    x = idx * _s64(0x9E3779B97F4A7C15) + _s64((seed * 0xD1342543DE82EF95 + (stream + 1) * 0xA0761D6478BD642F))
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    x = x ^ _lsr(x, 31)
    return _lsr(x, 11)


def _zipf(t, u):
    return torch.clamp(torch.searchsorted(t.zipf_thr, u, right=True), max=_V - 1)


def _chunk(t, seed, first_slot, n_slots, device):
    """word slots [first_slot, first_slot + n_slots) of the corpus of `seed` -> uint8 tensor"""
    slot = torch.arange(first_slot, first_slot + n_slots, dtype=torch.int64, device=device)
    word = _zipf(t, _u53(seed, 0, slot))
    variant = t.var_ids[torch.bucketize(_u53(seed, 1, slot), t.var_edges, right=True)]
    tok = variant * _V + word
    hdr = _u53(seed, 2, slot) < t.hdr_thr
    if first_slot == 0:
        hdr[0] = True
    H = 13
    count = torch.where(hdr, torch.full_like(word, H), torch.ones_like(word))
    off = torch.cumsum(count, 0) - count
    total = int(count.sum())
    out = torch.empty(total, dtype=torch.int64, device=device)
    out[off[~hdr]] = tok[~hdr]
    ho = off[hdr]
    hs = slot[hdr]
    rnd = lambda stream, hi: _u53(seed, stream, hs) % hi
    fb = t.fixed_base
    seq = [fb + 0, 8 * _V + _zipf(t, _u53(seed, 3, hs)),
           fb + 1, t.num_base + rnd(4, 4096), fb + 2, t.num_base + rnd(5, 4096), fb + 3, t.ts_base + rnd(6, 4096),
           fb + 4, 8 * _V + rnd(7, _V), fb + 5, t.num_base + rnd(8, 4096), fb + 6]
    for k, v in enumerate(seq):
        out[ho + k] = v
    lens = t.tok_len[out]
    ends = torch.cumsum(lens, 0)
    nbytes = int(ends[-1])
    tok_of = torch.repeat_interleave(torch.arange(total, device=device), lens, output_size=nbytes)
    pos = torch.arange(nbytes, device=device) - (ends - lens)[tok_of]
    return t.flat[t.tok_start[out][tok_of] + pos]


def enwik_like(nbytes, seed=12345, device="cpu", slots_per_chunk=1 << 21):
    """uint8 tensor of exactly `nbytes` enwik-shaped bytes on `device`: the same bytes for one seed on every run and
    every device (module docstring)."""
    device = torch.device(device)
    t = _tables(seed, device)
    parts, have, slot = [], 0, 0
    while have < nbytes:
        c = _chunk(t, seed, slot, slots_per_chunk, device)
        slot += slots_per_chunk
        parts.append(c)
        have += c.numel()
    out = torch.cat(parts)[:nbytes].contiguous()
    return out


def digest(x):
    """sha256 (first 16 hex digits) of a buffer's bytes — the `input_sha256` field of bench.py's line"""
    import hashlib
    a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    return hashlib.sha256(memoryview(np.ascontiguousarray(a))).hexdigest()[:16]


def adversarial(kind, n, seed=1):
    """Small edge-case inputs: returns bytes."""
    import random
    rng = random.Random(seed)
    if kind == "zeros":
        return bytes(n)
    if kind == "single":
        return b"a" * n
    if kind == "two":
        return bytes(rng.choice(b"ab") for _ in range(n))
    if kind == "random":
        return bytes(rng.randrange(256) for _ in range(n))
    if kind == "random_nonzero":
        return bytes(rng.randrange(1, 256) for _ in range(n))
    if kind.startswith("period"):
        p = int(kind[6:])
        base = bytes(rng.randrange(1, 256) for _ in range(p))
        return (base * (n // p + 1))[:n]
    if kind == "zero_tail":
        body = bytes(rng.randrange(1, 256) for _ in range(max(n - 40, 0)))
        return (body + bytes(40))[:n]
    if kind == "skewed":
        # geometric byte distribution: deep Huffman trees
        return bytes(min(int(-math.log(1.0 - rng.random()) * 3.0), 255) for _ in range(n))
    raise ValueError(kind)


FAMILIES = ("text", "lowent", "phrases", "runs", "pages")


def family(kind, seed, n):
    """seeded input families that stress different parts of the cluster machinery (tests/test_fuzz_gpu.py, scripts/fuzz_campaign.py,
    bench.py `adversarial`): numpy uint8 array of n bytes"""
    rng = np.random.default_rng(seed)
    if kind == "text":
        return enwik_like(n, seed=seed).numpy()
    if kind == "lowent":                      # tiny alphabet: a few hundred distinct words, clusters of hundreds of entries
        k = int(rng.integers(2, 7))
        return (rng.integers(0, k, n) + 97).astype(np.uint8)
    if kind == "phrases":                     # random bytes with a handful of phrases pasted at random places
        data = rng.integers(0, 256, n, dtype=np.uint8)
        phrases = [rng.integers(0, 256, int(rng.integers(4, 40)), dtype=np.uint8) for _ in range(6)]
        for _ in range(n // 60):
            ph = phrases[int(rng.integers(0, len(phrases)))]
            at = int(rng.integers(0, n - len(ph)))
            data[at:at + len(ph)] = ph
        return data
    if kind == "runs":                        # text interrupted by runs of one byte, 10 .. 3000 long
        data = enwik_like(n, seed=seed).numpy().copy()
        at = 0
        while at < n:
            at += int(rng.integers(200, 6000))
            ln = int(rng.integers(10, 3000))
            data[at:at + ln] = int(rng.integers(0, 256))
            at += ln
        return data
    if kind == "pages":                       # binary-like: zero pages, counters, a repeated record
        data = np.zeros(n, dtype=np.uint8)
        rec = rng.integers(0, 256, 24, dtype=np.uint8)
        at = 0
        while at + 64 < n:
            what = int(rng.integers(0, 4))
            ln = int(rng.integers(64, 5000))
            ln = min(ln, n - at)
            if what == 0:
                pass                          # zeros
            elif what == 1:
                data[at:at + ln] = np.arange(ln, dtype=np.uint32).astype(np.uint8)
            elif what == 2:
                data[at:at + ln] = np.resize(rec, ln)
            else:
                data[at:at + ln] = rng.integers(0, 256, ln, dtype=np.uint8)
            at += ln
        return data
    raise ValueError(kind)
