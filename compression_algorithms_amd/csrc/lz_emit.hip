// lz_emit.hip — greedy parse, token emission and stream concatenation for the LZ77 paths.
//
// Replaces the loop body of lz77_compress (algorithms/lz77/lz77.c:281-338; deflate variant
// algorithms/deflate/lz77.c:215-275) given the match finder's per-position candidates:
//
//   k_lz_parse_emit  per block: match length at every position (lz77.c:302-313), the greedy
//                    chain p -> p+1 | p+len resolved with 64-position chunk exit tables
//                    composed in two levels, token ranks by popcount prefix sums, tokens
//                    bit-packed LSB first through an LDS window (lz77.c:144-174: bit i of the
//                    stream is bit i%8 of byte i/8; deflate byte tokens lz77.c:176-197 are the
//                    same thing with 16/32-bit tokens)
//   k_lz_concat      block streams -> one stream, bit-contiguous (a funnel-shift gather, one
//                    thread per output dword)
//   (decoders: lz_decode.hip)
#include "lz_common.h"
#include "lz2.h"
#include <stdlib.h>


__global__ __launch_bounds__(1024)
void k_lz_parse_emit(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzScratch sc, Lz2Scratch s2, int use_v2,
                     uint64_t block0, uint32_t *__restrict__ trec_all)
{
    // region0: input bytes -> exit tables [64][1024] -> {token base, match base, staging window}
    __shared__ __attribute__((aligned(16))) uint8_t s_r0[LZ_MAX_BLOCK + LZ_TAIL + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_L[LZ_MAX_BLOCK];
    __shared__ uint64_t s_tok[1024], s_mat[1024];
    __shared__ uint8_t  s_entry[1024];
    __shared__ uint8_t  s_sexit[32][32];
    __shared__ uint8_t  s_sentry[32];
    __shared__ uint32_t s_scan[18];

    const int tid = threadIdx.x;
    const uint32_t lb = blockIdx.x;
    long long tk = clock64();
#define PE_TICK(k) do { if (s2.dbg && tid == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&s2.dbg[16 + (k)], (unsigned long long)(t2 - tk)); tk = t2; } } while (0)
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    const uint16_t *cand = sc.cand + (size_t)lb * LZ_MAX_BLOCK;
    const uint32_t W = 1u << P.wbits, max_len = (1u << P.lbits) - 1u;

    // ---- block -> LDS with the zero tail
    lz_block_to_lds(s_r0, src, n, (uint32_t)tid);
    if (tid == 0) s_L[LZ_MAX_BLOCK - 1] = 0;     // position 0xFFFF: its "pending" marker equals "none" (lz2.h); none unless a list says otherwise
    __syncthreads();

    PE_TICK(0);
    // ---- A: token length at every position, were a token to start there.  The LDS-resident finder hands
    //      over (position, candidate) LISTS (coalesced); the first pipeline an array indexed by position.
    const bool lists = use_v2 && !s2.meta[lb].fallback;
    auto token_len = [&](uint32_t p, uint32_t c) -> uint32_t {
        uint32_t len = 0;
        if (c != LZ_NONE16) {
            const uint32_t dist = p - c;
            const bool literal = P.deflate ? (dist >= W - 1u) : (dist == W);      // deflate lz77.c:223 / lz77.c:290
            if (!literal) {
                len = 4;                                                        // the words are equal
                // extend 4 bytes at a time (no `p < size` bound: the zero tail is there); first differing byte by ctz
                while (len < max_len) {
                    const uint32_t x = lds_word(s_r0, c + len) ^ lds_word(s_r0, p + len);
                    if (x) { len += (uint32_t)__builtin_ctz(x) >> 3; break; }
                    len += 4;
                }
                if (len > max_len) len = max_len;
            }
        }
        return len;
    };
    const uint16_t *l_pos = s2.plist + (size_t)lb * LZ_MAX_BLOCK, *l_cand = s2.cand + (size_t)lb * LZ_MAX_BLOCK;
    const uint16_t *b_pos = s2.bigpos + (size_t)lb * LZ2_BIG_STRIDE, *b_cand = s2.bigcand + (size_t)lb * LZ2_BIG_STRIDE;
    const uint32_t nbig = lists ? s2.meta[lb].nbig_entries : 0u;
    // (position, candidate) pairs of a list, 8 per lane per step: two 16-byte loads instead of sixteen 2-byte ones
    auto for_each_pair = [&](const uint16_t *lp, const uint16_t *lc, uint32_t cnt, auto &&fn) {
        // the next 8 pairs are in flight while these are worked on (the body is LDS work the compiler will not hoist
        // the loads over: measured, this loop was waiting on HBM, not on LDS)
        uint4 np = make_uint4(0, 0, 0, 0), nc = make_uint4(0, 0, 0, 0);
        uint32_t j0 = tid * 8u;
        if (j0 + 8u <= cnt) { np = *reinterpret_cast<const uint4 *>(lp + j0); nc = *reinterpret_cast<const uint4 *>(lc + j0); }
        for (; j0 < cnt; j0 += 1024u * 8u) {
            if (j0 + 8u <= cnt) {
                const uint4 vp = np, vc = nc;
                const uint32_t j1 = j0 + 1024u * 8u;
                if (j1 + 8u <= cnt) { np = *reinterpret_cast<const uint4 *>(lp + j1); nc = *reinterpret_cast<const uint4 *>(lc + j1); }
                const uint32_t wp[4] = {vp.x, vp.y, vp.z, vp.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) fn((wp[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu, (wc[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
            } else {
                for (uint32_t j = j0; j < cnt; ++j) fn((uint32_t)lp[j], (uint32_t)lc[j]);
            }
        }
    };
    // token lengths of a list, eight pairs at a time: the first compare step (bytes 4..7 — where most matches of a text
    // end) of all eight is issued together; a per-pair loop would walk its LDS round trips one pair after the other.
    // Matches that go on past byte 7 are QUEUED and extended 64 at a time: the extension is a data-dependent loop, and run in
    // place every wave paid its longest match (up to seven rounds) for each of its eight pair slots with a handful of lanes
    // alive — the phase was VALU-bound on those rounds (~1000 instructions per thread and step, 4 waves per SIMD: half of
    // this kernel).  The queue is one register per lane: a ballot ranks the lanes that have a pair to extend, ds_permute (the
    // LDS crossbar, no LDS memory) sends pair number r to lane (fill + r) mod 64 — the other lanes send to the slots that are
    // left, so the whole thing is a permutation — and whenever 64 pairs are together they are extended by a full wave.
    uint32_t q_cur = 0, q_nxt = 0, q_fill = 0;                   // queued (position | candidate << 16) of this lane; pairs queued (uniform)
    const uint32_t lane = (uint32_t)tid & 63u;
    auto extend = [&](uint32_t item, bool on) {
        if (on) {
            const uint32_t p = item & 0xFFFFu, c = item >> 16;
            uint32_t len = 8;
            while (len < max_len) {
                const uint32_t x = lds_word(s_r0, c + len) ^ lds_word(s_r0, p + len);
                if (x) { len += (uint32_t)__builtin_ctz(x) >> 3; break; }
                len += 4;
            }
            s_L[p] = (uint8_t)(len > max_len ? max_len : len);
        }
    };
    auto enqueue = [&](bool ext, uint32_t item) {
        const uint64_t mk = __ballot(ext);
        if (mk == 0ull) return;                                    // (uniform)
        const uint32_t cnt = (uint32_t)__popcll(mk);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));   // queued lanes below this one
        const uint32_t slot = ext ? below : cnt + (lane - below);  // a stable partition of the 64 lanes: a permutation
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(((slot + q_fill) & 63u) << 2), (int)item);
        const uint32_t rel = (lane - q_fill) & 63u;                // lane q_fill + r received queued pair r
        if (rel < cnt) { if (lane >= q_fill) q_cur = got; else q_nxt = got; }
        q_fill += cnt;
        if (q_fill >= 64u) { extend(q_cur, true); q_cur = q_nxt; q_fill -= 64u; }
    };
    auto lengths_of_list = [&](const uint16_t *lp, const uint16_t *lc, uint32_t cnt) {
        uint4 np = make_uint4(0, 0, 0, 0), nc = make_uint4(0, 0, 0, 0);
        uint32_t j0 = tid * 8u;
        if (j0 + 8u <= cnt) { np = *reinterpret_cast<const uint4 *>(lp + j0); nc = *reinterpret_cast<const uint4 *>(lc + j0); }
        // (the trip count is the same for every lane of a wave — the queue's ballots need the whole wave: 512 pairs per wave and step)
        for (uint32_t w0 = ((uint32_t)tid & ~63u) * 8u; w0 < cnt; w0 += 1024u * 8u, j0 += 1024u * 8u) {
            const bool full = j0 + 8u <= cnt;                       // (every lane runs the body: the queue works on whole waves)
            {
                const uint4 vp = np, vc = nc;
                const uint32_t j1 = j0 + 1024u * 8u;
                if (j1 + 8u <= cnt) { np = *reinterpret_cast<const uint4 *>(lp + j1); nc = *reinterpret_cast<const uint4 *>(lc + j1); }
                const uint32_t wp[4] = {vp.x, vp.y, vp.z, vp.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
                uint32_t pp[8], cc[8], xx[8];
                bool act[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    pp[k] = (wp[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu; cc[k] = (wc[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                    const uint32_t dist = pp[k] - cc[k];
                    // a candidate that the literal rule does not reject (deflate lz77.c:223 / lz77.c:290); c == p: pending or pad
                    act[k] = full && cc[k] != LZ_NONE16 && cc[k] != pp[k] && !(P.deflate ? (dist >= W - 1u) : (dist == W));
                    xx[k] = 0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) if (act[k]) xx[k] = lds_word(s_r0, cc[k] + 4u) ^ lds_word(s_r0, pp[k] + 4u);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool ext = act[k] && xx[k] == 0u && max_len > 8u;
                    if (full && cc[k] != pp[k] && !ext) {              // (pending / pad: the other list, or nobody, writes it)
                        uint32_t len = 0;
                        if (act[k]) { len = xx[k] ? 4u + ((uint32_t)__builtin_ctz(xx[k]) >> 3) : 8u; if (len > max_len) len = max_len; }
                        s_L[pp[k]] = (uint8_t)len;
                    }
                    enqueue(ext, pp[k] | (cc[k] << 16));
                }
            }
            if (!full)
                for (uint32_t j = j0; j < cnt && j < j0 + 8u; ++j) { const uint32_t p = lp[j], c = lc[j]; if (c != p) s_L[p] = (uint8_t)token_len(p, c); }
        }
    };
    auto flush_queue = [&]() { extend(q_cur, lane < q_fill); q_fill = 0; };
    if (lists) {
        lengths_of_list(l_pos, l_cand, n);
        lengths_of_list(b_pos, b_cand, nbig);
        flush_queue();
    } else {
        for (uint32_t p = tid; p < n; p += 1024u) s_L[p] = (uint8_t)token_len(p, cand[p]);
    }
    __syncthreads();

    PE_TICK(1);
    // ---- B: exit offset of every position of a 64-position chunk into the next chunk
    uint8_t *ex = s_r0;                       // ex[o * 1024 + chunk]
    {
        // the chunk's 64 lengths come in as four 16-byte LDS reads up front: the backward walk then has ONE dependent LDS
        // read per step (the exit of the position it jumps to) instead of two
        const uint32_t c = tid;
        uint32_t lw[16];
        {
            const uint4 *lp = reinterpret_cast<const uint4 *>(s_L + c * 64u);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const uint4 v = lp[q]; lw[4 * q] = v.x; lw[4 * q + 1] = v.y; lw[4 * q + 2] = v.z; lw[4 * q + 3] = v.w; }
        }
#pragma unroll
        for (int o = 63; o >= 0; --o) {
            const uint32_t p = c * 64u + (uint32_t)o;
            uint32_t e = 0;
            if (p < n) {
                const uint32_t l = (lw[o >> 2] >> (8 * (o & 3))) & 0xFFu;
                const uint32_t nx = (uint32_t)o + (l ? l : 1u);
                e = nx >= 64u ? nx - 64u : ex[nx * 1024u + c];
            }
            ex[(uint32_t)o * 1024u + c] = (uint8_t)e;
        }
    }
    __syncthreads();
    PE_TICK(2);
    // compose over super-chunks of 32 chunks.  A chunk is entered at offset <= max_len - 1 <= 30 for the
    // reference's length fields (lbits 4 and 5); longer fields take the serial walk below.
    const bool wide = max_len > 31u;
    if (!wide) {
        const uint32_t scn = tid >> 5, o = tid & 31u;
        uint32_t x = o;
        for (uint32_t c = scn * 32u; c < scn * 32u + 32u; ++c) x = ex[x * 1024u + c];
        s_sexit[scn][o] = (uint8_t)x;
        __syncthreads();
        if (tid == 0) {
            uint32_t e = 0;
            for (uint32_t s = 0; s < 32; ++s) { s_sentry[s] = (uint8_t)e; e = s_sexit[s][e]; }
        }
        __syncthreads();
        if (tid < 32) {
            uint32_t xx = s_sentry[tid];
            for (uint32_t c = tid * 32u; c < tid * 32u + 32u; ++c) { s_entry[c] = (uint8_t)xx; xx = ex[xx * 1024u + c]; }
        }
        __syncthreads();
    }

    PE_TICK(3);
    // ---- C: token starts of each chunk
    {
        const uint32_t c = tid;
        uint64_t tok = 0, mat = 0;
        if (!wide) {
            if (c * 64u < n) {
                uint32_t o = s_entry[c];
                while (o < 64u && c * 64u + o < n) {
                    const uint32_t l = s_L[c * 64u + o];
                    tok |= 1ull << o;
                    if (l) mat |= 1ull << o;
                    o += l ? l : 1u;
                }
            }
            s_tok[c] = tok; s_mat[c] = mat;
        } else {
            s_tok[c] = 0; s_mat[c] = 0;
        }
    }
    __syncthreads();
    if (wide && tid == 0) {
        // lbits > 5: a plain serial walk (not a reference configuration; kept for completeness)
        uint32_t p = 0;
        while (p < n) {
            const uint32_t l = s_L[p];
            s_tok[p >> 6] |= 1ull << (p & 63u);
            if (l) s_mat[p >> 6] |= 1ull << (p & 63u);
            p += l ? l : 1u;
        }
    }
    __syncthreads();
    PE_TICK(4);
    const uint64_t my_tok = s_tok[tid], my_mat = s_mat[tid];
    uint32_t ntok = 0, nmat = 0;
    const uint32_t tbase = block_exclusive_scan<uint32_t>((uint32_t)__popcll(my_tok), OpAddU32(), 0u, s_scan, &ntok);
    const uint32_t mbase = block_exclusive_scan<uint32_t>((uint32_t)__popcll(my_mat), OpAddU32(), 0u, s_scan, &nmat);
    uint32_t *tb = reinterpret_cast<uint32_t *>(s_r0);            // [1025]
    uint32_t *mb = tb + 1026;                                      // [1025]
    uint32_t *stage = mb + 1026;                                   // [TPR * 1024 + 16]
    constexpr uint32_t TPR = 4;                                   // tokens per thread and emit round
    uint16_t *md = reinterpret_cast<uint16_t *>(stage + TPR * 1024 + 16);   // [<= 16384] distance of the k-th match token
    tb[tid] = tbase; mb[tid] = mbase;
    if (tid == 0) { tb[1024] = ntok; mb[1024] = nmat; }
    __syncthreads();
    if (lists) {
        auto put = [&](uint32_t p, uint32_t c) {
            const uint32_t ch = p >> 6, o = p & 63u;
            if ((s_mat[ch] >> o) & 1ull) md[mb[ch] + (uint32_t)__popcll(s_mat[ch] & ((1ull << o) - 1ull))] = (uint16_t)(p - c);
        };
        for_each_pair(l_pos, l_cand, n, [&](uint32_t p, uint32_t c) { if (c != p && c != LZ_NONE16) put(p, c); });
        for_each_pair(b_pos, b_cand, nbig, [&](uint32_t p, uint32_t c) { if (c != LZ_NONE16 && c != p) put(p, c); });
        __syncthreads();
    }

    PE_TICK(5);
    // ---- D: emit, 2048 tokens per barrier round (two per lane: their latencies overlap)
    const uint32_t LB = P.deflate ? 16u : 9u, MB = P.deflate ? 32u : (1u + P.wbits + P.lbits);
    uint32_t *slot = sc.slot + (size_t)lb * LZ_SLOT_WORDS;
    uint32_t carry = 0;
    if (trec_all) {
        // mode H (defh.hip): no packed tokens; one 32-bit record per token, in token order, for the entropy stage:
        //   literal  byte                       match  1<<31 | length << 16 | distance
        // The 286-bin tally of the block (deflate/lz77.c:206,231,273; deflate/huffman.c:49-62) is taken here, where every
        // token passes by anyway, and handed to k_defh_encode at the end of the block's slot: it no longer reads the
        // token records twice.
        uint32_t *trec = trec_all + (size_t)lb * LZ_MAX_BLOCK;
        // eight copies of the tally (a pair of waves each): the frequent symbols of a text serialise on one LDS address
        constexpr uint32_t NH = 8u, HS = 288u;
        static_assert(NH * HS <= TPR * 1024u, "the tallies live in the staging window");
        for (uint32_t i = tid; i < NH * HS; i += 1024u) stage[i] = 0;
        // a literal token's byte goes where its length (0) was: the chunk's 64 bytes come in as four coalesced 16-byte loads
        // instead of one dependent byte load per literal inside the walk
        {
            const uint64_t lit = my_tok & ~my_mat;
            const uint32_t p0 = (uint32_t)tid * 64u;
            if (lit) {
                const bool v16 = (((uintptr_t)src) & 15u) == 0 && p0 + 64u <= n;
                // (the chunk's four 16-byte loads first, then their uses: one round trip, not four)
                uint4 v4[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                if (v16) {
#pragma unroll
                    for (uint32_t k4 = 0; k4 < 4u; ++k4) v4[k4] = *reinterpret_cast<const uint4 *>(src + p0 + 16u * k4);
                }
#pragma unroll
                for (uint32_t k4 = 0; k4 < 4u; ++k4) {
                    uint32_t w[4] = {v4[k4].x, v4[k4].y, v4[k4].z, v4[k4].w};
                    if (!v16) { for (uint32_t j = 0; j < 16u; ++j) if (p0 + 16u * k4 + j < n) w[j >> 2] |= (uint32_t)src[p0 + 16u * k4 + j] << (8u * (j & 3u)); }
#pragma unroll
                    for (uint32_t j = 0; j < 16u; ++j)
                        if ((lit >> (16u * k4 + j)) & 1ull) s_L[p0 + 16u * k4 + j] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
                }
            }
        }
        __syncthreads();
        // POSITION-driven: thread c walks the token starts of its own chunk (it knows how many tokens and matches lie before
        // it) — no token -> chunk search, no bit selection, nothing but LDS on the walk
        {
            uint64_t tk = my_tok;
            uint32_t t = tbase, mi = mbase;
            const uint32_t p0 = (uint32_t)tid * 64u;
            uint32_t *hist = stage + ((uint32_t)tid >> 7) * HS;
            while (tk) {
                const uint32_t o = (uint32_t)__builtin_ctzll(tk), p = p0 + o;
                tk &= tk - 1ull;
                uint32_t rec = s_L[p], sym = rec;                         // a literal's byte, or a match's length
                if ((my_mat >> o) & 1ull) {
                    const uint32_t d = lists ? (uint32_t)md[mi] : p - cand[p];
                    ++mi;
                    rec = 0x80000000u | (rec << 16) | d;
                    sym = 256u + ((uint32_t)__builtin_clz(d & 0xFFFFu) - 16u);
                }
                atomicAdd(&hist[sym], 1u);
                trec[t++] = rec;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < 288u; i += 1024u) {
            uint32_t v = 0;
            for (uint32_t h = 0; h < NH; ++h) v += stage[h * HS + i];
            slot[LZ_DEFH_HIST_AT + i] = v;
        }
        if (tid == 0) sc.block_bits[lb] = ntok;
        PE_TICK(6);
        if (s2.dbg && tid == 0) atomicAdd((unsigned long long *)&s2.dbg[31], 1ull);
        return;
    }
    // The packed formats, POSITION-driven like the records above: thread c walks the token starts of its own chunk; the bit
    // position of its first token follows from the token and match counts before it.  The output goes through an LDS window
    // of 4096 words per round (tokens are OR-ed in, LSB first: bit i of the stream is bit i%8 of byte i/8, lz77.c:144-174);
    // bit positions grow with the chunk number, so a thread takes part in the one or two rounds its chunk's range meets and
    // resumes where it stopped.  A literal's byte is parked where its length (0) was: four coalesced 16-byte loads per chunk
    // instead of a dependent byte load per literal.  (The token-driven loop this replaces — token -> chunk search, r-th set
    // bit, scattered byte loads, four tokens per thread and round — cost 117 k cycles per block.)
    {
        const uint64_t lit = my_tok & ~my_mat;
        const uint32_t p0 = (uint32_t)tid * 64u;
        if (lit) {
            const bool v16 = (((uintptr_t)src) & 15u) == 0 && p0 + 64u <= n;
            uint4 v4[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            if (v16) {
#pragma unroll
                for (uint32_t k4 = 0; k4 < 4u; ++k4) v4[k4] = *reinterpret_cast<const uint4 *>(src + p0 + 16u * k4);
            }
#pragma unroll
            for (uint32_t k4 = 0; k4 < 4u; ++k4) {
                uint32_t w[4] = {v4[k4].x, v4[k4].y, v4[k4].z, v4[k4].w};
                if (!v16) { for (uint32_t j = 0; j < 16u; ++j) if (p0 + 16u * k4 + j < n) w[j >> 2] |= (uint32_t)src[p0 + 16u * k4 + j] << (8u * (j & 3u)); }
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j)
                    if ((lit >> (16u * k4 + j)) & 1ull) s_L[p0 + 16u * k4 + j] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
            }
        }
    }
    {
        constexpr uint32_t WW = TPR * 1024u;                              // window words per round (stage holds WW + 16)
        const uint64_t total = (uint64_t)(ntok - nmat) * LB + (uint64_t)nmat * MB;
        uint64_t tk = my_tok;
        uint64_t q = (uint64_t)(tbase - mbase) * LB + (uint64_t)mbase * MB;   // bit position of this chunk's first token
        uint32_t mi = mbase;
        const uint32_t p0 = (uint32_t)tid * 64u;
        for (uint64_t w0 = 0; (w0 << 5) < total; w0 += WW) {
            for (uint32_t i = tid; i < WW + 2u; i += 1024u) stage[i] = (i == 0) ? carry : 0u;
            __syncthreads();
            const uint64_t hi = (w0 + WW) << 5;
            while (tk && q < hi) {
                const uint32_t o = (uint32_t)__builtin_ctzll(tk), p = p0 + o;
                tk &= tk - 1ull;
                uint32_t v = s_L[p], nbits = LB;                           // a literal's byte, or a match's length
                if ((my_mat >> o) & 1ull) {
                    const uint32_t d = lists ? (uint32_t)md[mi] : p - cand[p];
                    ++mi;
                    v = P.deflate ? (1u | (d << 8) | (v << 24)) : (1u | (d << 1) | (v << (1u + P.wbits)));
                    nbits = MB;
                } else v = P.deflate ? (v << 8) : (v << 1);
                const uint32_t rel = (uint32_t)(q - (w0 << 5)), wi = rel >> 5, sh = rel & 31u;
                atomicOr(&stage[wi], v << sh);
                if (sh + nbits > 32u) atomicOr(&stage[wi + 1], v >> (32u - sh));
                q += nbits;
            }
            __syncthreads();
            // complete words of this window: all of them, or up to the stream's last complete word
            const uint64_t endw = (total >> 5) < w0 + WW ? (total >> 5) : w0 + WW;
            const uint32_t ncomplete = (uint32_t)(endw - w0);
            for (uint32_t i = tid; i < ncomplete; i += 1024u) slot[w0 + i] = stage[i];
            carry = stage[ncomplete];
            __syncthreads();
        }
    }
    PE_TICK(6);
    if (s2.dbg && tid == 0) atomicAdd((unsigned long long *)&s2.dbg[31], 1ull);
    const uint64_t total_bits = (uint64_t)(ntok - nmat) * LB + (uint64_t)nmat * MB;
    if (tid == 0) {
        slot[total_bits >> 5] = (total_bits & 31u) ? carry : 0u;
        slot[(total_bits >> 5) + 1] = 0;
        sc.block_bits[lb] = total_bits;
    }
}

// ---------------------------------------------------------------------------------------------
// single-workgroup exclusive scan of the batch's block bit counts; also publishes the global
// exclusive offsets (base + local) to the caller's array
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_lz_scan_blocks(const uint64_t *bits, uint32_t nb, const uint64_t *__restrict__ base_bits,
                      uint64_t *excl_local /* may alias `bits`: scanned in place */, uint64_t *__restrict__ excl_global)
{
    // one small workgroup (it sits between two LDS-hungry kernels of its stream and must find a free slot quickly):
    // PER consecutive blocks per thread, a batch holds up to 4096 blocks
    __shared__ uint64_t s_tmp[6];
    const uint32_t PER = (nb + 255u) / 256u, a = threadIdx.x * PER, b = a + PER < nb ? a + PER : nb;
    uint64_t v = 0;
    for (uint32_t i = a; i < b; ++i) v += bits[i];
    uint64_t inc = v;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint64_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t run = 0; for (int w = 0; w < 4; ++w) { uint64_t t = s_tmp[w]; s_tmp[w] = run; run += t; } s_tmp[4] = run; }
    __syncthreads();
    uint64_t run = s_tmp[wave] + inc - v;
    const uint64_t base = *base_bits;
    for (uint32_t i = a; i < b; ++i) { const uint64_t x = bits[i]; excl_local[i] = run; excl_global[i] = base + run; run += x; }
    __syncthreads();                                    // (in place: every thread has read its own range before anyone writes [nb])
    if (threadIdx.x == 0) { excl_local[nb] = s_tmp[4]; excl_global[nb] = base + s_tmp[4]; }
}

__global__ void k_lz_advance(uint64_t *base_bits, const uint64_t *excl_local, uint32_t nb) { *base_bits += excl_local[nb]; }

__device__ __forceinline__ uint32_t extract_bits(const uint32_t *w, uint64_t lo, uint32_t k)   // k in 1..32
{
    const uint64_t wi = lo >> 5; const uint32_t sh = (uint32_t)(lo & 31u);
    const uint64_t two = (uint64_t)w[wi] | ((uint64_t)w[wi + 1] << 32);
    const uint64_t v = two >> sh;
    return k >= 32 ? (uint32_t)v : ((uint32_t)v & ((1u << k) - 1u));
}

// one thread per output dword of the batch's bit range [base, base + total)
__global__ __launch_bounds__(256)
void k_lz_concat(const uint32_t *__restrict__ slots, const uint64_t *__restrict__ excl_local, uint32_t nb,
                 const uint64_t *__restrict__ base_bits, uint32_t *__restrict__ out, uint64_t cap_words, uint32_t slot_words)
{
    const uint64_t base = *base_bits, total = excl_local[nb];
    const uint64_t wfirst = base >> 5;
    uint64_t wlast = (base + total + 31) >> 5;                                 // [wfirst, wlast)
    if (wlast > cap_words) wlast = cap_words;       // never past the caller's buffer (the host checked cap against the bound)
    // grid-stride: the grid is sized for a typical stream, the loop covers the worst case
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + wfirst; j < wlast; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g0 = j << 5, g1 = g0 + 32;
        uint64_t pos = g0 > base ? g0 : base;
        const uint64_t end = g1 < base + total ? g1 : base + total;
        // largest i with base + excl[i] <= pos
        uint32_t lo = 0, hi = nb - 1;
        const uint64_t rel = pos - base;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (excl_local[mid] <= rel) lo = mid; else hi = mid - 1; }
        uint32_t i = lo, word = 0;
        while (pos < end) {
            const uint64_t bs = base + excl_local[i], be = base + excl_local[i + 1];
            const uint64_t se = be < end ? be : end;
            if (se > pos) {
                const uint32_t k = (uint32_t)(se - pos);
                word |= extract_bits(slots + (size_t)i * slot_words, pos - bs, k) << (uint32_t)(pos - g0);
                pos = se;
            }
            if (pos == be) ++i;
        }
        if (j == wfirst && (base & 31u)) atomicOr(&out[j], word);      // shares a dword with the previous batch
        else out[j] = word;
    }
}

// ---------------------------------------------------------------------------------------------
// decode: one wave per block, output staged in LDS.  A block stops at its original length: the
// last match may overshoot (the encoder compares into the zero tail), SURVEY.md A.3.4.
// ---------------------------------------------------------------------------------------------
// k <= 25 stream bits from bit `pos`; bytes at or past `nbytes` read as zero (the table and the stream come from a file
// or a peer: nothing is read outside the buffer the caller described)
__device__ __forceinline__ uint32_t stream_bits(const uint8_t *s, uint64_t nbytes, uint64_t pos, uint32_t k)
{
    const uint64_t byte = pos >> 3;
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) if (byte + i < nbytes) v |= (uint64_t)s[byte + i] << (8 * i);
    return (uint32_t)(v >> (pos & 7u)) & ((1u << k) - 1u);
}

// =============================================================================================
// host side
// =============================================================================================
size_t   lz_scratch_bytes(uint32_t nb);
void     lz_carve(mi_ctx *ctx, uint32_t nb, LzScratch *sc, Lz2Scratch *sc2, int set);
mi_status lz_find_stage_a(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                          const LzScratch &sc, const Lz2Scratch &sc2, hipStream_t s, hipStream_t sf, hipEvent_t ev_part, hipEvent_t ev_fb,
                          hipEvent_t ev_wide);
mi_status lz_find_stage_b(mi_ctx *ctx, const LzP &P, uint32_t nb, const Lz2Scratch &sc2, hipStream_t s, int which);
bool     lz_use_v2();
mi_status lz_run_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                      const LzScratch &sc, const Lz2Scratch &sc2, hipStream_t s);
mi_status lz_check_params(const mi_lz_params *p);
mi_status lz_find_batch(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                        const LzScratch &sc, hipStream_t s);
uint32_t lz_batch_blocks(mi_ctx *ctx, uint64_t nblocks);

void defh_launch_encode(const uint32_t *trec, uint32_t *slots, uint64_t *block_bits, uint32_t nb, hipStream_t s);

// lzw.hip: the lz77 flavour on blocks above 64 KiB
size_t    lzw_scratch_bytes(uint32_t nb, uint32_t block);
void      lzw_carve(mi_ctx *ctx, uint32_t nb, uint32_t block, LzwScratch *sc);
uint32_t  lzw_batch_blocks(mi_ctx *ctx, uint64_t nblocks, uint32_t block);
mi_status lzw_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb, const LzwScratch &sc, hipStream_t s);
mi_status lzw_or_lzs_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb, const LzwScratch &sc, hipStream_t s);
void      lzw_launch_parse_emit(const uint8_t *d_in, uint64_t n, const LzP &P, const LzwScratch &sc, uint64_t block0, uint32_t nb, hipStream_t s);
void      lz_launch_decode(const uint8_t *d_stream, uint64_t stream_bytes, const uint64_t *d_block_bits, const LzP &P, uint8_t *d_out,
                           uint64_t n, uint64_t nblocks, uint32_t *err, hipStream_t s);      // lz_decode.hip
extern "C" uint64_t mi_deflate_h_bound_bytes(uint64_t n, const mi_lz_params *p);
mi_status mi_encode_host_pipelined(mi_ctx *ctx, const mi_lz_params *p, int mode_h, const uint8_t *h_in, uint64_t n,
                                   uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits, bool *done);

// mode_h = 0: the reference's token stream.  mode_h = 1: the same tokens, entropy coded per block (defh.hip); the
// per-block records are word aligned, so the same scan / concatenate kernels place them.
static mi_status lz_encode_impl(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                                uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_block_bits, void *stream, int mode_h)
{
    if (!ctx || !d_out || !d_block_bits || (n && !d_in)) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (((uintptr_t)d_out & 3u) != 0) return MI_ERR_ARG;
    if (mode_h && (!p->deflate || p->lbits > 5 || p->wbits > 16)) return MI_ERR_ARG;
    if (cap_bytes < (mode_h ? mi_deflate_h_bound_bytes(n, p) : mi_lz_bound_bytes(n, p))) return MI_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    const LzP P = lz_params_of(ctx, p);
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    if (P.block > LZ_MAX_BLOCK) {
        // blocks above 64 KiB (lz77 flavour): the HBM-resident finder of lzw.hip, one stream, batches sized by workspace
        if (mode_h) return MI_ERR_ARG;
        uint32_t nbw = lzw_batch_blocks(ctx, nblocks, P.block);
        // (a workspace the context already holds is reused whatever its size; a batch that does not fit is halved)
        while ((st = mi_ws_reserve(ctx, lzw_scratch_bytes(nbw, P.block) + 4096)) == MI_ERR_NOMEM && nbw > 1) nbw = (nbw + 1) / 2;
        if (st) return st;
        LzwScratch ws;
        lzw_carve(ctx, nbw, P.block, &ws);
        uint64_t *base_bits_w = reinterpret_cast<uint64_t *>((uint8_t *)ctx->ws + lzw_scratch_bytes(nbw, P.block));
        MI_HIP(ctx, hipMemsetAsync(base_bits_w, 0, 8, s));
        if (nblocks == 0) { MI_HIP(ctx, hipMemsetAsync(d_block_bits, 0, 8, s)); return MI_OK; }
        for (uint64_t b0 = 0; b0 < nblocks; b0 += nbw) {
            const uint32_t nb = (uint32_t)((nblocks - b0) < nbw ? (nblocks - b0) : nbw);
            st = lzw_or_lzs_find(ctx, P, d_in, n, b0, nb, ws, s);
            if (st) return st;
            { mi_prof_scope pr(ctx, "k_lzw_parse_emit", s, (uint64_t)nb * P.block);
              lzw_launch_parse_emit(d_in, n, P, ws, b0, nb, s); }
            hipLaunchKernelGGL(k_lz_scan_blocks, dim3(1), dim3(256), 0, s, ws.block_bits, nb, base_bits_w, ws.block_bits, d_block_bits + b0);
            { mi_prof_scope pr(ctx, "k_lz_concat", s, (uint64_t)nb * P.block);
              const uint64_t typw = (uint64_t)nb * (P.block / 4 + 64);
              hipLaunchKernelGGL(k_lz_concat, dim3((unsigned)((typw + 255) / 256 < 65535 ? (typw + 255) / 256 : 65535)), dim3(256), 0, s, ws.slot, ws.block_bits, nb,
                                 base_bits_w, reinterpret_cast<uint32_t *>(d_out), cap_bytes / 4, ws.slot_words); }
            hipLaunchKernelGGL(k_lz_advance, dim3(1), dim3(1), 0, s, base_bits_w, ws.block_bits, nb);
        }
        MI_HIP(ctx, hipGetLastError());
        return MI_OK;
    }
    const uint32_t nbmax = lz_batch_blocks(ctx, nblocks);
    // three stages on three streams, MI_SETS scratch sets in rotation:
    //   `s`          partition + find of batch i+2          (LDS heavy: one / three workgroups per CU)
    //   ctx->side    replay of the exported clusters of i+1 (almost no LDS: runs beside the find)
    //   ctx->parse   parse / emit / concatenate of batch i  (one 150 KiB workgroup per CU; the stream has raised priority)
    // plus ctx->fb for the normally empty fallback chain.  Fork/join with events only: no host synchronisation.
    const bool overlap = nblocks > nbmax && !getenv("MI_LZ_NO_OVERLAP");
    const int nsets = overlap ? MI_SETS : 1;
    const size_t set_bytes = mi_align_up(lz_scratch_bytes(nbmax), 4096);
    const size_t trec_bytes = mode_h ? (size_t)nbmax * LZ_MAX_BLOCK * 4 : 0;       // token records, one array per set
    st = mi_ws_reserve(ctx, set_bytes * nsets + 8192 + trec_bytes * nsets);
    if (st) return st;
    LzScratch sc[MI_SETS]; Lz2Scratch sc2[MI_SETS];
    for (int k = 0; k < nsets; ++k) lz_carve(ctx, nbmax, &sc[k], &sc2[k], k);
    uint64_t *base_bits = reinterpret_cast<uint64_t *>((uint8_t *)ctx->ws + set_bytes * nsets);
    uint32_t *trec_base = reinterpret_cast<uint32_t *>((uint8_t *)ctx->ws + set_bytes * nsets + 8192);
    MI_HIP(ctx, hipMemsetAsync(base_bits, 0, 8, s));
    if (nblocks == 0) { MI_HIP(ctx, hipMemsetAsync(d_block_bits, 0, 8, s)); return MI_OK; }
    hipStream_t sb = overlap ? ctx->side : s, sp = overlap ? ctx->parse : s;
    // stage C of one batch (set k): parse / emit (+ the entropy stage in mode H) / scan / concatenate
    auto stage_c = [&](int k, uint64_t b0, uint32_t nb) -> mi_status {
        uint64_t *excl_local = sc[k].block_bits;                   // reused in place by the scan
        uint32_t *trec = mode_h ? trec_base + (size_t)k * nbmax * LZ_MAX_BLOCK : nullptr;
        {
            mi_prof_scope pr(ctx, "k_lz_parse_emit", sp, (uint64_t)nb * P.block);
            hipLaunchKernelGGL(k_lz_parse_emit, dim3(nb), dim3(1024), 0, sp, d_in, n, P, sc[k], sc2[k], lz_use_v2() ? 1 : 0, b0, trec);
        }
        if (mode_h) {
            mi_prof_scope ph(ctx, "k_defh_encode", sp, (uint64_t)nb * P.block);
            defh_launch_encode(trec, sc[k].slot, sc[k].block_bits, nb, sp);
        }
        hipLaunchKernelGGL(k_lz_scan_blocks, dim3(1), dim3(256), 0, sp, sc[k].block_bits, nb, base_bits, excl_local, d_block_bits + b0);
        {
            mi_prof_scope pr(ctx, "k_lz_concat", sp, (uint64_t)nb * P.block);
            const uint64_t typw = (uint64_t)nb * (P.block / 4 + 64);   // about one output byte per input byte; the kernel strides
            hipLaunchKernelGGL(k_lz_concat, dim3((unsigned)((typw + 255) / 256)), dim3(256), 0, sp, sc[k].slot, excl_local, nb,
                               base_bits, reinterpret_cast<uint32_t *>(d_out), cap_bytes / 4, (uint32_t)LZ_SLOT_WORDS);
        }
        hipLaunchKernelGGL(k_lz_advance, dim3(1), dim3(1), 0, sp, base_bits, excl_local, nb);
        if (overlap) MI_HIP(ctx, hipEventRecord(ctx->ev_done[k], sp));
        return MI_OK;
    };
    // MI_LZ_SCHED=1 holds the parse of batch i-1 until the partition of batch i is through (both want a whole CU's LDS for
    // one workgroup and the partition is on the chain the pipeline waits for).  Measured: no difference (12.66 vs 12.65
    // GB/s, partition 22.2 vs 22.6 ms) — the pipeline is bound by the total work, not by one chain.  Left as a switch.
    const bool hold_parse = overlap && lz_use_v2() && getenv("MI_LZ_SCHED") != nullptr;
#ifdef MI_MEASURE
    // measurement builds only (make EXTRA=-DMI_MEASURE): what would the step cost if a stage were free?  The stream is WRONG.
    const int skip = getenv("MI_LZ_SKIP") ? atoi(getenv("MI_LZ_SKIP")) : 0;       // 1: no stage B, 2: no stage C, 3: neither
#else
    const int skip = 0;
#endif
    // The fallback chains of consecutive batches go to two streams WHEN the input lives in the fallback: their kernels are long
    // serial chains on few workgroups and overlap well — pages 1.72 -> 2.16, runs 1.65 -> 2.77 GB/s on 10^8 B.  On text a second
    // stream costs 7 % even idle (20.4 -> 19.1 GB/s, whatever GPU_MAX_HW_QUEUES says), so it exists only while the hint says so.
    // The hint (ctx->h_order[1]) is the fallback count of the last batch the GPU has FINISHED — every batch of a call is queued
    // before the first one runs, so in practice it is what an EARLIER call left: a first call on non-text input runs on one
    // fallback stream and small grids, a text call right after such input still carries the second stream (bench.py reports the
    // adversarial families warm AND cold).  Decided once per call; the stream is created here and released here — when a later
    // call finds the hint low and the stream idle (hipStreamQuery: no host wait) — or with the context, never inside the loop
    // (stream create/destroy per batch, skipped on early returns, may block in hipStreamDestroy).
    const uint32_t fb_hint = (overlap && ctx->h_order) ? __atomic_load_n(ctx->h_order + 1, __ATOMIC_RELAXED) : 0u;
    const bool fb_busy = fb_hint > 8u;                                                                             // (text: 0)
    // When MOST blocks of a batch fall back (zeros, binary pages) stage B has next to nothing to do: the chains of the odd batches
    // then go to ITS stream and no fifth stream is made at all — every stream beyond the four of the pipeline costs all of them
    // (the process has four hardware queues: pages 33.2 -> 29.5 ms, zeros 12.2 -> 8.4 ms per 10^8 B against fb2).  Where stage B has
    // real work beside the fallback (the "runs" family: 17 % of the blocks fall back, the rest export wide clusters) a chain in front
    // of it costs more than the fifth stream (22.5 -> 27.2 ms): there fb2 stays.  MI_LZ_FB2_SIDE=0 / 1 forces either (A/B).
    static const char *fb2_env = getenv("MI_LZ_FB2_SIDE");
    const uint32_t fb_of = (overlap && ctx->h_order) ? __atomic_load_n(ctx->h_order + 2, __ATOMIC_RELAXED) : 0u;    // blocks of the batch the hint is from
    const bool fb2_side = fb_busy && (fb2_env ? fb2_env[0] == '1' : fb_hint * 2u >= fb_of);
    if (fb_busy && !ctx->fb2 && !fb2_side) {
        int lo_ = 0, hi_ = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_, &hi_);
        if (hipStreamCreateWithPriority(&ctx->fb2, hipStreamNonBlocking, lo_) != hipSuccess) { (void)hipGetLastError(); ctx->fb2 = nullptr; }
    } else if ((!fb_busy || fb2_side) && ctx->fb2) {
        if (hipStreamQuery(ctx->fb2) == hipSuccess) { (void)hipStreamDestroy(ctx->fb2); ctx->fb2 = nullptr; }
        else (void)hipGetLastError();                     // still draining an earlier call's chains: try again next time
    }
    uint64_t batch = 0, prev_b0 = 0; uint32_t prev_nb = 0; int prev_k = -1;
    // MI_LZ_TAPER=1: the last batches taper (full batches while two or more are left, then halves down to a quarter batch).  When
    // the last find has finished only that batch's replay and parse are left, one small kernel after the other on an emptying GPU:
    // ~6 ms of a 47.8 ms step with five equal batches (kernel timeline, round 4).  Measured: 21.17 -> 20.89 GB/s — the shorter
    // drain is worth less than what seven batches instead of five cost; off by default.
    static const bool taper = getenv("MI_LZ_TAPER") && getenv("MI_LZ_TAPER")[0] == '1';
    uint32_t nb_next = 0;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += nb_next, ++batch) {
        const uint64_t rem = nblocks - b0;
        uint32_t nb = (uint32_t)(rem < nbmax ? rem : nbmax);
        if (taper && overlap && rem < 2ull * nbmax && rem > nbmax / 4u) {
            const uint64_t half = (rem + 1) / 2;
            nb = (uint32_t)(half > nbmax / 4u ? half : nbmax / 4u);
            if (nb > nbmax) nb = nbmax;
        }
        nb_next = nb;
        const int k = (int)(batch % (uint64_t)nsets);
        if (overlap && batch >= (uint64_t)nsets) MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_done[k], 0));   // set k is free again
        static const bool b_split = !(getenv("MI_LZ_B_SPLIT") && getenv("MI_LZ_B_SPLIT")[0] == '0');
        const bool solo = !overlap && b_split && lz_use_v2() && !skip;
        st = lz_find_stage_a(ctx, P, d_in, n, b0, nb, sc[k], sc2[k], s,
                             overlap ? ((batch & 1u) && fb_busy && (ctx->fb2 || fb2_side) ? (fb2_side ? sb : ctx->fb2) : ctx->fb) : (solo ? ctx->fb : s), ctx->ev_part[k], ctx->ev_fb[k], ctx->ev_wide[k]);
        if (st) return st;
        if (hold_parse && prev_k >= 0) {
            MI_HIP(ctx, hipStreamWaitEvent(sp, ctx->ev_part[k], 0));
            st = stage_c(prev_k, prev_b0, prev_nb);
            if (st) return st;
        }
        if (overlap) {
            MI_HIP(ctx, hipEventRecord(ctx->ev_find[k], s)); MI_HIP(ctx, hipStreamWaitEvent(sb, ctx->ev_find[k], 0));
            if (lz_use_v2() && !getenv("MI_LZ_WIDE_INLINE")) MI_HIP(ctx, hipStreamWaitEvent(sb, ctx->ev_wide[k], 0));    // the wide parts' exports (lz_find.hip)
        }
        // ONE batch (no pipeline: the side, parse and fallback streams are idle): the lane replays run BESIDE the wave / row replays, the
        // long size classes on the side stream, the short ones on the parse stream — the groups are independent, each is a chain of
        // launches with tails, and what follows needs all of them; the (normally empty) fallback chain and the wide finder leave the
        // partition -> find chain for the fallback stream as in the pipeline (MI_LZ_B_SPLIT=0: one chain on one stream, A/B)
        if (solo && !(skip & 1)) {
            if (!getenv("MI_LZ_WIDE_INLINE")) MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_wide[0], 0));        // the wide parts' exports
            MI_HIP(ctx, hipEventRecord(ctx->ev_find[0], s));
            MI_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_find[0], 0)); MI_HIP(ctx, hipStreamWaitEvent(ctx->parse, ctx->ev_find[0], 0));
            st = lz_find_stage_b(ctx, P, nb, sc2[k], ctx->side, 4);
            if (st) return st;
            MI_HIP(ctx, hipEventRecord(ctx->ev_replay[0], ctx->side));
            st = lz_find_stage_b(ctx, P, nb, sc2[k], ctx->parse, 1);
            if (st) return st;
            MI_HIP(ctx, hipEventRecord(ctx->ev_done[0], ctx->parse));
            st = lz_find_stage_b(ctx, P, nb, sc2[k], s, 2);
            MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_replay[0], 0)); MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_done[0], 0));
            MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_fb[0], 0));                                           // the fallback blocks' candidates
        } else if (!(skip & 1)) st = lz_find_stage_b(ctx, P, nb, sc2[k], sb, 7);
        if (st) return st;
        if (overlap) {
            MI_HIP(ctx, hipEventRecord(ctx->ev_replay[k], sb));
            MI_HIP(ctx, hipStreamWaitEvent(sp, ctx->ev_replay[k], 0));
            if (lz_use_v2()) MI_HIP(ctx, hipStreamWaitEvent(sp, ctx->ev_fb[k], 0));      // the fallback blocks' candidates
        }
        if (hold_parse) { prev_k = k; prev_b0 = b0; prev_nb = nb; }
        else if (!(skip & 2)) { st = stage_c(k, b0, nb); if (st) return st; }
        else if (overlap) MI_HIP(ctx, hipEventRecord(ctx->ev_done[k], sp));
    }
    if (hold_parse && prev_k >= 0) { st = stage_c(prev_k, prev_b0, prev_nb); if (st) return st; }
    if (overlap) {                                                 // join: the last stage finishes everything
        MI_HIP(ctx, hipEventRecord(ctx->ev_fork, sp));
        MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_fork, 0));
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

extern "C" mi_status mi_lz_encode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                                      uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_block_bits, void *stream)
{
    return lz_encode_impl(ctx, p, d_in, n, d_out, cap_bytes, d_block_bits, stream, 0);
}

extern "C" mi_status mi_deflate_h_encode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                                             uint8_t *d_out, uint64_t cap_bytes, uint64_t *d_block_bits, void *stream)
{
    return lz_encode_impl(ctx, p, d_in, n, d_out, cap_bytes, d_block_bits, stream, 1);
}

mi_status mi_encode_again_if_unstable(mi_ctx *ctx, uint32_t seen_before, mi_status st, mi_status (*again)(void *), void *arg);     // host_api.hip
static mi_status lz_encode_host_once(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                     uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits);
struct LzHostEncArgs { mi_ctx *ctx; const mi_lz_params *p; const uint8_t *h_in; uint64_t n; uint8_t *h_out; uint64_t cap; uint64_t *bits; };

extern "C" mi_status mi_lz_encode(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                  uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    if (!ctx || !h_out || !h_block_bits || (n && !h_in) || !p) return MI_ERR_ARG;
    mi_order_poll(ctx);
    const uint32_t seen = ctx->order_violations;
    LzHostEncArgs a{ctx, p, h_in, n, h_out, cap_bytes, h_block_bits};
    return mi_encode_again_if_unstable(ctx, seen, lz_encode_host_once(ctx, p, h_in, n, h_out, cap_bytes, h_block_bits),
        [](void *v) { LzHostEncArgs *q = (LzHostEncArgs *)v; return lz_encode_host_once(q->ctx, q->p, q->h_in, q->n, q->h_out, q->cap, q->bits); }, &a);
}

static mi_status lz_encode_host_once(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                     uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    const uint64_t nblocks = p->block ? (n + p->block - 1) / p->block : 0;
    const uint64_t bound = mi_lz_bound_bytes(n, p);
    if (cap_bytes < bound) return MI_ERR_CAPACITY;
    if (lz_check_params(p) == MI_OK && p->deflate) {       // byte tokens: chunks overlap their transfers with the encoder (host_api.hip)
        bool done = false;
        const mi_status ps = mi_encode_host_pipelined(ctx, p, 0, h_in, n, h_out, cap_bytes, h_block_bits, &done);
        if (ps || done) return ps;
    }
    uint8_t *d_in = nullptr, *d_out = nullptr; uint64_t *d_bits = nullptr;
    mi_status st = MI_OK;
    hipStream_t s = mi_host_stream(ctx);
    if (hipMalloc(&d_in, n + 64) != hipSuccess || hipMalloc(&d_out, bound + 64) != hipSuccess ||
        hipMalloc(&d_bits, (nblocks + 1) * 8) != hipSuccess) st = MI_ERR_NOMEM;
    if (st == MI_OK && n && hipMemcpyAsync(d_in, h_in, n, hipMemcpyHostToDevice, s) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK) st = mi_lz_encode_dev(ctx, p, d_in, n, d_out, bound + 64, d_bits, s);
    if (st == MI_OK && hipMemcpyAsync(h_block_bits, d_bits, (nblocks + 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK && hipStreamSynchronize(s) != hipSuccess) st = MI_ERR_HIP;
    if (st == MI_OK) {
        const uint64_t bytes = (h_block_bits[nblocks] + 7) / 8;
        if (bytes && hipMemcpy(h_out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) st = MI_ERR_HIP;
    }
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_bits);
    return st;
}

// launch alone: errors accumulate in *err (an mi_err_slot the caller reads once everything it launched has run)
mi_status mi_lz_decode_launch(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                              const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, uint32_t *err, hipStream_t s)
{
    if (!ctx || !d_stream || !d_block_bits || (n && !d_out) || !err) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (n == 0) return MI_OK;
    const LzP P = lz_params_of(ctx, p);
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    mi_prof_scope pr(ctx, "k_lz_decode", s, n);
    lz_launch_decode(d_stream, stream_bytes, d_block_bits, P, d_out, n, nblocks, err, s);       // any block size
    return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_HIP;
}

extern "C" mi_status mi_lz_decode_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                                      const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, void *stream)
{
    if (!ctx || !d_stream || !d_block_bits || (n && !d_out)) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (n == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *err = mi_err_slot(ctx, s);
    if (!err) return MI_ERR_HIP;
    st = mi_lz_decode_launch(ctx, p, d_stream, stream_bytes, d_block_bits, d_out, n, err, s);
    if (st) return st;
    uint32_t h_err = 0;
    MI_HIP(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    return h_err ? MI_ERR_CORRUPT : MI_OK;
}
