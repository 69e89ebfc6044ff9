"""Synthetic enwik-shaped byte buffers.

The reference's data step (get_data.sh:1-9) downloads enwik9 and derives enwik8/7/6 with
`head -c`; there is no network here, so benches and tests run on a seeded stand-in with the
same gross structure: a Zipf-distributed vocabulary of pseudo-words
wrapped in the MediaWiki XML page/revision skeleton with [[links]], '''bold''', &quot;
entities, == headings == and punctuation.  Like the reference's derived files, a shorter
buffer is a prefix of a longer one (same seed, same device type).

Everything is tensor ops, so a 10^9-byte buffer is produced directly in HBM in well under a
second; on CPU the same code serves the small test inputs.  CPU and GPU generators give
different (equally shaped) bytes for one seed — comparisons always use one buffer.
"""
import math

import torch

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
        self.zipf = 1.0 / torch.arange(1, _V + 1, dtype=torch.float32)
        # cumulative thresholds of the decoration variants (SURVEY.md Appendix C proportions)
        self.var_edges = torch.tensor([0.04, 0.05, 0.06, 0.12, 0.17, 0.175, 0.18])
        self.var_ids = torch.tensor([1, 2, 3, 4, 5, 6, 7, 0])


_TAB = {}


def _tables(seed, device):
    key = (seed, str(device))
    if key not in _TAB:
        t = _Tables(seed)
        for name in ("tok_len", "tok_start", "flat", "zipf", "var_edges", "var_ids"):
            setattr(t, name, getattr(t, name).to(device))
        _TAB[key] = t
    return _TAB[key]


def _chunk(t, n_slots, gen, device):
    """one chunk of `n_slots` word slots -> uint8 tensor"""
    word = torch.multinomial(t.zipf, n_slots, replacement=True, generator=gen)
    r = torch.rand(n_slots, generator=gen, device=device)
    variant = t.var_ids[torch.bucketize(r, t.var_edges, right=True)]
    tok = variant * _V + word
    hdr = torch.rand(n_slots, generator=gen, device=device) < (1.0 / 1500.0)
    hdr[0] = True
    H = 13
    count = torch.where(hdr, torch.full_like(word, H), torch.ones_like(word))
    off = torch.cumsum(count, 0) - count
    total = int(count.sum())
    out = torch.empty(total, dtype=torch.int64, device=device)
    out[off[~hdr]] = tok[~hdr]
    ho = off[hdr]
    nh = ho.numel()
    rnd = lambda hi: torch.randint(0, hi, (nh,), generator=gen, device=device)
    fb = t.fixed_base
    seq = [fb + 0, 8 * _V + torch.multinomial(t.zipf, nh, replacement=True, generator=gen),
           fb + 1, t.num_base + rnd(4096), fb + 2, t.num_base + rnd(4096), fb + 3, t.ts_base + rnd(4096),
           fb + 4, 8 * _V + rnd(_V), fb + 5, t.num_base + rnd(4096), fb + 6]
    for k, v in enumerate(seq):
        out[ho + k] = v
    lens = t.tok_len[out]
    ends = torch.cumsum(lens, 0)
    nbytes = int(ends[-1])
    tok_of = torch.repeat_interleave(torch.arange(total, device=device), lens, output_size=nbytes)
    pos = torch.arange(nbytes, device=device) - (ends - lens)[tok_of]
    return t.flat[t.tok_start[out][tok_of] + pos]


def enwik_like(nbytes, seed=12345, device="cpu", slots_per_chunk=1 << 21):
    """uint8 tensor of exactly `nbytes` enwik-shaped bytes on `device`."""
    device = torch.device(device)
    t = _tables(seed, device)
    gen = torch.Generator(device=device).manual_seed(seed + 1)
    parts, have = [], 0
    while have < nbytes:
        c = _chunk(t, slots_per_chunk, gen, device)
        parts.append(c)
        have += c.numel()
    out = torch.cat(parts)[:nbytes].contiguous()
    return out


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
