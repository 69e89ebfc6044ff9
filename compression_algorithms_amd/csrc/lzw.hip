// lzw.hip — the lz77 flavour on blocks LARGER than 64 KiB (up to 1 MiB), so that a 64 KiB window really slides
// (SURVEY.md 8d config 2 "64 KiB ... 1 MiB").  Replaces the same reference functions as the LDS-resident
// finder — hash / insert_hash_table / find and the loop of lz77_compress, algorithms/lz77/lz77.c:13-108,264-345 — for
// blocks the u16 / 65 536-entry kernels of lz2_*.hip cannot hold.
//
// Same idea, different residence.  find() at p is a function of the block prefix (every position is inserted once, in
// order), and a probe cluster — a maximal run of the parking sweep over the homes of ALL entries of the block — owns
// exactly as many consecutive buckets as it has entries and never meets another cluster.  So:
//   k_lzw_keys          mix32(word) | position for every position              (the hash, lz77.c:13-41)
//   k_lzw_sort          stable LSD radix sort by home bucket, through HBM      (3 x 8 bits)
//   k_lzw_sweep         parking sweep -> cluster number and dense home slot of every position
//   k_lzw_sort (again)  positions by (cluster, time)
//   k_lzw_heads         first sorted index of every cluster; singletons answered ("none")
//   k_lzw_replay        one WAVE per cluster replays its entries in time order against the cluster's own slice of a LITERAL
//                       table in HBM (live flag, position, mixed word per dense bucket): 64 buckets per probe step,
//                       FIFO retirement by position (lz77.c:70-76), the one-time spurious clear of bucket 0
//   k_lzw_parse_emit    one workgroup per block walks its 64 KiB segments in order: match lengths from the bytes in HBM/L2,
//                       the greedy chain by composed 64-position exit tables (as k_lz_parse_emit) entered at the offset the
//                       previous segment left, tokens LSB-first (lz77.c:290-330) appended to the block's slot
//   (decoder: k_lz_decode_bits of lz_decode.hip, whose LDS ring is the window)
// The table state lives in HBM and is touched through agent-scope atomics (the wave re-reads what it wrote a moment ago;
// plain loads could hit stale lines of the CU's L1).  This path is exact, not fast: a cluster is a serial chain of HBM/L2
// round trips.  It exists so that WINDOW_BITS 16 means what it says; DESIGN.md 6 says what would make it quick.
#include "lz_common.h"
#include "lz_replay.h"
#include <stdlib.h>
#include <stdio.h>

#define LZW_MAX_BLOCK   (1u << 20)
#define LZW_SEG         65536u                    // parse/emit segment
#define LZW_NONE        0xFFFFFFFFu
#define LZW_SLOT_WORDS(block) ((uint32_t)(((uint64_t)(block) * 9u + 7u) / 8u / 4u + 24u))     // 9 bits per byte worst case + one overshoot

// 4 bytes at an arbitrary offset of the block; bytes past the block end read as zero (the parity definition of the
// reference's over-read)
__device__ __forceinline__ uint32_t lzw_word(const uint8_t *src, uint32_t p, uint32_t n)
{
    if (p + 8u <= n && ((((uintptr_t)src) & 3u) == 0)) {
        const uint32_t *a = reinterpret_cast<const uint32_t *>(src + (p & ~3u));
        const uint64_t v = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
        return (uint32_t)(v >> ((p & 3u) * 8u));
    }
    uint32_t w = 0;
    for (uint32_t k = 0; k < 4 && p + k < n; ++k) w |= (uint32_t)src[p + k] << (8 * k);
    return w;
}

__global__ __launch_bounds__(256)
void k_lzw_keys(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0)
{
    const uint32_t lb = blockIdx.y;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    uint64_t *e = sc.eA + (size_t)lb * sc.S;
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u)
        e[p] = ((uint64_t)lz_mix32(lzw_word(src, p, n)) << 32) | p;
}

// stable LSD radix sort of the block's u64 elements by `bits` key bits starting at `shift` of (element >> 32) & mask,
// 8 bits per pass, ping-pong through HBM; one workgroup per block.  Result in eA when the number of passes is even.
__global__ __launch_bounds__(1024)
void k_lzw_sort(uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0, uint32_t key_mask, uint32_t npass, int from_gid)
{
    __shared__ uint32_t s_cnt[17][256];
    const uint32_t lb = blockIdx.x;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    uint64_t *a = sc.eA + (size_t)lb * sc.S, *b = sc.eB + (size_t)lb * sc.S;
    if (from_gid) {                                   // second sort: (cluster << 32 | position) in time order
        const uint32_t *g = sc.gid + (size_t)lb * sc.S;
        for (uint32_t p = threadIdx.x; p < n; p += 1024) a[p] = ((uint64_t)g[p] << 32) | p;
        __syncthreads();
    }
    // ballots always: this finder is the exact fallback behind lzs.hip (and what MI_LZW_SLICED=0 selects) — it must not
    // lean on the lane order of LDS atomics that the fast paths use and check (lz_common.h lz_order_violation)
    const uint32_t arank = 0u;
    for (uint32_t pass = 0; pass < npass; ++pass) {
        const uint64_t *src = (pass & 1u) ? b : a;
        uint64_t *dst = (pass & 1u) ? a : b;
        const uint32_t sh = 8u * pass;
        radix_pass_1024<8, uint64_t>(n, s_cnt,
            [&](uint32_t i) { return __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },   // this CU's L1 may hold lines of an earlier pass
            [&](uint64_t e) { return (((uint32_t)(e >> 32) & key_mask) >> sh) & 255u; },
            [&](uint32_t j, uint64_t e) { dst[j] = e; }, arank);
        __threadfence();
        __syncthreads();
    }
}

// parking sweep over the home-sorted order (in `srt`): cluster heads are the weak prefix maxima of home_k - k
// (DESIGN.md 2); dense home slot = index of the cluster's head + (home - the head's home).
__global__ __launch_bounds__(1024)
void k_lzw_sweep(uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0, int sorted_in_b)
{
    __shared__ int32_t  s_i32[18];
    __shared__ uint64_t s_u64[18];
    __shared__ int32_t  s_run;
    __shared__ uint32_t s_gs, s_base, s_count, s_hs;
    constexpr uint32_t CH = 8;
    const uint32_t lb = blockIdx.x, tid = threadIdx.x;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint64_t *srt = (sorted_in_b ? sc.eB : sc.eA) + (size_t)lb * sc.S;
    uint32_t *gid = sc.gid + (size_t)lb * sc.S, *rd = sc.rd + (size_t)lb * sc.S, *relw = sc.relw + (size_t)lb * sc.S;
    const uint32_t Tmask = (1u << P.tbits) - 1u;
    if (tid == 0) { s_run = INT32_MIN; s_gs = 0; s_base = 0; s_count = 0; s_hs = 0; }
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n; t0 += 1024u * CH) {
        const uint32_t k0 = t0 + tid * CH, k1 = (k0 + CH < n) ? k0 + CH : n;
        uint32_t rp[CH], rm[CH]; int32_t rh[CH];
        int32_t mx = INT32_MIN;
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t k = k0 + c;
            rp[c] = 0; rh[c] = 0; rm[c] = 0;
            if (k < k1) {
                const uint64_t e = srt[k];
                rp[c] = (uint32_t)e; rm[c] = (uint32_t)(e >> 32); rh[c] = (int32_t)(rm[c] & Tmask);
                const int32_t g = rh[c] - (int32_t)k;
                mx = g > mx ? g : mx;
            }
        }
        int32_t tile_max;
        int32_t premax = block_exclusive_scan<int32_t>(mx, OpMaxI32(), INT32_MIN, s_i32, &tile_max);
        const int32_t carry_run = s_run;
        premax = premax > carry_run ? premax : carry_run;
        // heads in my chunk, the last head's index + 1 (for the carry of the open cluster)
        uint32_t nheads = 0; int32_t lasthead = -1;
        {
            int32_t run = premax;
#pragma unroll
            for (uint32_t c = 0; c < CH; ++c) {
                const uint32_t k = k0 + c;
                if (k < k1) {
                    const int32_t g = rh[c] - (int32_t)k;
                    const bool head = (k == 0) || (g >= run);
                    run = g > run ? g : run;
                    if (head) { ++nheads; lasthead = (int32_t)k; }
                }
            }
        }
        // last start of a home run in my chunk (the word id of an entry is the sorted index of the first entry of its word)
        int32_t lastrun = -1;
        {
            uint32_t prev_h = (k0 > 0 && k0 < n) ? (uint32_t)((uint32_t)(srt[k0 - 1] >> 32) & Tmask) : 0u;
#pragma unroll
            for (uint32_t c = 0; c < CH; ++c) {
                const uint32_t k = k0 + c;
                if (k < k1) { if (k == 0 || (uint32_t)rh[c] != prev_h) lastrun = (int32_t)k; prev_h = (uint32_t)rh[c]; }
            }
        }
        struct OpHL {           // heads: sum (bits 0..20); last head index + 1: max (21..41); last run start + 1: max (42..62)
            __device__ uint64_t operator()(uint64_t a, uint64_t b) const {
                const uint64_t M = 0x1FFFFFull;
                const uint64_t s0 = ((a & M) + (b & M)) & M, a1 = (a >> 21) & M, b1 = (b >> 21) & M, a2 = (a >> 42) & M, b2 = (b >> 42) & M;
                return s0 | ((a1 > b1 ? a1 : b1) << 21) | ((a2 > b2 ? a2 : b2) << 42);
            }
        };
        uint64_t tot2;
        const uint64_t pre2 = block_exclusive_scan<uint64_t>((uint64_t)nheads | ((uint64_t)(uint32_t)(lasthead >= 0 ? lasthead + 1 - (int32_t)t0 : 0) << 21) |
                                                             ((uint64_t)(uint32_t)(lastrun >= 0 ? lastrun + 1 - (int32_t)t0 : 0) << 42), OpHL(), 0ull, s_u64, &tot2);
        const uint32_t count0 = s_count;
        uint32_t cur_gs = s_gs, cur_base = s_base, cur_gid = count0 + (uint32_t)(pre2 & 0x1FFFFFull) - 1u;   // the cluster open at my first entry
        const int32_t gs_carry = (int32_t)((pre2 >> 21) & 0x1FFFFFull) - 1;      // last head before my chunk inside this tile (tile-relative)
        if (gs_carry >= 0) { cur_gs = t0 + (uint32_t)gs_carry; cur_base = (uint32_t)((uint32_t)(srt[cur_gs] >> 32) & Tmask); }
        uint32_t cur_hs = s_hs;                                                // start of the home run open at my first entry
        const int32_t hs_carry = (int32_t)((pre2 >> 42) & 0x1FFFFFull) - 1;
        if (hs_carry >= 0) cur_hs = t0 + (uint32_t)hs_carry;
        uint32_t hs_mix = (k0 < n && (k0 > 0)) ? (uint32_t)(srt[cur_hs] >> 32) : 0u;
        {
            int32_t run = premax;
            uint32_t seen = 0;
            uint32_t prev_h2 = (k0 > 0 && k0 < n) ? (uint32_t)((uint32_t)(srt[k0 - 1] >> 32) & Tmask) : 0u;
#pragma unroll
            for (uint32_t c = 0; c < CH; ++c) {
                const uint32_t k = k0 + c;
                if (k < k1) {
                    const int32_t h = rh[c], g = h - (int32_t)k;
                    const bool head = (k == 0) || (g >= run);
                    run = g > run ? g : run;
                    if (head) { cur_gs = k; cur_base = (uint32_t)h; cur_gid = count0 + (uint32_t)(pre2 & 0x1FFFFFull) + seen; ++seen; }
                    // word id: sorted index of the first entry of this word (entries of one home are in time order; two
                    // different words on one home are rare: a short scan of the run)
                    const uint32_t mixk = rm[c];
                    uint32_t id;
                    if (k == 0 || (uint32_t)h != prev_h2) { cur_hs = k; hs_mix = mixk; id = k; }
                    else if (mixk == hs_mix) id = cur_hs;
                    else {
                        id = k;
                        for (uint32_t kk = cur_hs + 1; kk < k; ++kk) if ((uint32_t)(srt[kk] >> 32) == mixk) { id = kk; break; }
                    }
                    prev_h2 = (uint32_t)h;
                    gid[rp[c]] = cur_gid;
                    rd[rp[c]] = cur_gs + ((uint32_t)h - cur_base);
                    // relative to the cluster (both < its size): what the LDS-resident replay reads, if the cluster is <= 65535
                    const uint32_t rrel = (uint32_t)h - cur_base, irel = id - cur_gs;
                    relw[rp[c]] = (rrel < 0xFFFFu ? rrel : 0xFFFFu) | ((irel < 0xFFFFu ? irel : 0xFFFFu) << 16);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            s_run = tile_max > carry_run ? tile_max : carry_run;
            s_count = count0 + (uint32_t)(tot2 & 0x1FFFFFull);
            const int32_t lh = (int32_t)((tot2 >> 21) & 0x1FFFFFull) - 1;   // last head of the tile, if any (tile-relative)
            if (lh >= 0) { s_gs = t0 + (uint32_t)lh; s_base = (uint32_t)((uint32_t)(srt[t0 + (uint32_t)lh] >> 32) & Tmask); }
            const int32_t lr = (int32_t)((tot2 >> 42) & 0x1FFFFFull) - 1;
            if (lr >= 0) s_hs = t0 + (uint32_t)lr;
        }
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t *nc = sc.ncl + (size_t)lb * 4;
        nc[0] = s_count;
        nc[1] = (n && ((uint32_t)(srt[0] >> 32) & Tmask) == 0u) ? 1u : 0u;   // cluster 0 starts at bucket 0: the spurious clear is its
    }
}

// first sorted index of every cluster (clusters are contiguous in the (cluster, time) order, and in the same index range
// as in the home order); a cluster of one entry is answered here: its only find() precedes its only insert.  Larger ones
// join the list of their size class; every entry's {relative home slot, word id} moves into (cluster, time) order.
#define LZW_CAP_S 1024u
#define LZW_CAP_M 4096u
#define LZW_CAP_L 24576u
#define LZW_CAP_K 8192u
#define LZW_NCLS  6u
#define LZW_TINY  7u                        // clusters of 2..7 entries: one LANE each, the table in registers
__global__ __launch_bounds__(256)
void k_lzw_heads(uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0, int sorted_in_b)
{
    const uint32_t lb = blockIdx.y;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint64_t *srt = (sorted_in_b ? sc.eB : sc.eA) + (size_t)lb * sc.S;
    uint32_t *cs = sc.cstart + (size_t)lb * (sc.S + 2);
    uint32_t *cand = sc.cand + (size_t)lb * sc.S;
    const uint32_t *relw = sc.relw + (size_t)lb * sc.S;
    uint32_t *ent = sc.ent + (size_t)lb * sc.S;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint64_t e = srt[k];
        const uint32_t g = (uint32_t)(e >> 32), p = (uint32_t)e;
        ent[k] = relw[p];
        const bool head = k == 0 || (uint32_t)(srt[k - 1] >> 32) != g;
        if (head) {
            cs[g] = k;
            if (k + 1 == n || (uint32_t)(srt[k + 1] >> 32) != g) cand[p] = LZW_NONE;
        }
        if (k + 1 == n) cs[g + 1] = n;
    }
}

// after k_lzw_heads: one thread per cluster appends it to the list of its size class.  Ranks are taken in LDS and ONE
// global atomic per workgroup, class and round reserves the list space (four counters shared by every cluster of the
// batch: a global atomic per cluster serialised ~4 M of them, 50-100 ms per 100 MB).
__global__ __launch_bounds__(256)
void k_lzw_classify(LzwScratch sc, uint32_t nb)
{
    __shared__ uint32_t s_n[LZW_NCLS], s_base[LZW_NCLS];
    for (uint32_t lb = blockIdx.y; lb < nb; lb += gridDim.y) {
        const uint32_t ncl = sc.ncl[(size_t)lb * 4];
        const uint32_t *cs = sc.cstart + (size_t)lb * (sc.S + 2);
        for (uint32_t c0 = blockIdx.x * 256u; c0 < ncl; c0 += gridDim.x * 256u) {     // uniform trip count per workgroup
            if (threadIdx.x < LZW_NCLS) s_n[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t c = c0 + threadIdx.x;
            uint32_t cls = LZW_NCLS, rank = 0;
            if (c < ncl) {
                const uint32_t m = cs[c + 1] - cs[c];
                if (m >= 2) { cls = m <= LZW_TINY ? 5u : m <= LZW_CAP_S ? 0u : m <= LZW_CAP_M ? 1u : m <= LZW_CAP_K ? 2u : m <= LZW_CAP_L ? 3u : 4u; rank = atomicAdd(&s_n[cls], 1u); }
            }
            __syncthreads();
            if (threadIdx.x < LZW_NCLS && s_n[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&sc.ccount[threadIdx.x], s_n[threadIdx.x]);
            __syncthreads();
            if (cls < LZW_NCLS) sc.clist[cls][s_base[cls] + rank] = ((uint64_t)lb << 32) | c;
            __syncthreads();
        }
    }
}

// LDS-resident replay of one cluster by one wave (the shape of big_replay in lz_replay.h, with 32-bit positions kept out
// of the loop): occupancy bitmap in registers, per bucket the occupant as {word id, entry index}, per entry its bucket.
// find() returns an ENTRY INDEX; k_lzw_resolve turns it into a position afterwards, off the serial chain.
template <int CAP, int NW>
__global__ __launch_bounds__(64)
void k_lzw_replay_lds(LzP P, LzwScratch sc, uint32_t cls)
{
    __shared__ uint32_t s_occ[CAP];                       // bucket -> word id | entry index << 16
    __shared__ uint16_t s_slot[CAP];                      // entry -> bucket
    const uint32_t lane = threadIdx.x;
    const uint32_t W = 1u << P.wbits;
    const uint32_t count = sc.ccount[cls];
    for (uint32_t ci = blockIdx.x; ci < count; ci += gridDim.x) {
        const uint64_t item = sc.clist[cls][ci];
        const uint32_t lb = (uint32_t)(item >> 32), c = (uint32_t)item;
        const uint32_t *cs = sc.cstart + (size_t)lb * (sc.S + 2);
        const uint32_t s = cs[c], m = cs[c + 1] - s;
        const uint64_t *srt = sc.eA + (size_t)lb * sc.S + s;              // the cluster's entries in time order
        const uint32_t *ent = sc.ent + (size_t)lb * sc.S + s;
        uint16_t *ce = sc.cand_e + (size_t)lb * sc.S + s;
        bool anom_pending = (c == 0u) && sc.ncl[(size_t)lb * 4 + 1];
        WaveBitmap<NW> bm;
        bm.clear();
        uint32_t ev = 0, out_acc = 0xFFFFu;
        uint32_t ev_pos = lane < m ? (uint32_t)srt[lane] : 0u;            // positions of entries [ev & ~63, +64): the next to retire
        uint32_t pe = RLANE(ev_pos, 0);
        uint32_t n_pos = ev_pos, n_ent = lane < m ? ent[lane] : 0u;       // the next 64 entries are in flight while these are replayed
        for (uint32_t i0 = 0; i0 < m; i0 += 64) {
            const uint32_t ii = i0 + lane;
            const uint32_t c_pos = n_pos, c_ent = n_ent;
            if (ii + 64 < m) { n_pos = (uint32_t)srt[ii + 64]; n_ent = ent[ii + 64]; }
            const uint32_t lim = (m - i0) < 64u ? (m - i0) : 64u;
            for (uint32_t t = 0; t < lim; ++t) {
                const uint32_t i = i0 + t;
                const uint32_t p = RLANE(c_pos, t), re = RLANE(c_ent, t), r = re & 0xFFFFu, id = re >> 16;
                while (ev < i && (uint64_t)pe + W < (uint64_t)p) {           // FIFO retirement (lz77.c:70-76)
                    const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_slot[ev]);
                    bm.clear_bit(sl, lane);                                 // clears the bucket, whoever sits there
                    ++ev;
                    if ((ev & 63u) == 0) { const uint32_t q = ev + lane; ev_pos = q < m ? (uint32_t)srt[q] : 0u; }
                    pe = RLANE(ev_pos, ev & 63u);
                }
                if (anom_pending && p > W - 1u) { bm.clear_bit(0u, lane); anom_pending = false; }   // bucket 0, once
                uint32_t res = 0xFFFFu;
                if (bm.test(r)) {
                    const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_occ[r]);
                    if ((h & 0xFFFFu) == id) res = h >> 16;
                    else {
                        uint32_t e = bm.first_zero_from(r + 1u, lane);
                        if (e > (uint32_t)CAP) e = (uint32_t)CAP;
                        for (uint32_t b0 = r + 1u; b0 < e; b0 += 64u) {
                            const uint32_t b = b0 + lane;
                            const uint32_t o = b < e ? s_occ[b] : 0u;
                            const uint64_t hit = __ballot(b < e && (o & 0xFFFFu) == id);
                            if (hit) { res = RLANE(o, (uint32_t)__builtin_ctzll(hit)) >> 16; break; }
                        }
                    }
                }
                const uint32_t b = bm.first_zero_from(r, lane);             // insert: first fit (inside the cluster by the parking bound)
                bm.flip(b, lane);
                s_occ[b] = id | (i << 16); s_slot[i] = (uint16_t)b;
                if (lane == t) out_acc = res;
                __builtin_amdgcn_wave_barrier();
            }
            if (ii < m) ce[ii] = (uint16_t)out_acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// clusters of 2..7 entries (most clusters of a text): one LANE per cluster, the whole table in registers — 7 occupancy bits and
// 3-bit fields for the occupant's word id, the occupant's entry index and every entry's bucket (the shape of
// replay_small_reg in lz2_find.hip, with 32-bit positions)
__global__ __launch_bounds__(256)
void k_lzw_replay_tiny(LzP P, LzwScratch sc)
{
    const uint32_t W = 1u << P.wbits;
    const uint32_t count = sc.ccount[5];
    for (uint32_t ci = blockIdx.x * 256u + threadIdx.x; ci < count; ci += gridDim.x * 256u) {
        const uint64_t item = sc.clist[5][ci];
        const uint32_t lb = (uint32_t)(item >> 32), c = (uint32_t)item;
        const uint32_t *cs = sc.cstart + (size_t)lb * (sc.S + 2);
        const uint32_t s = cs[c], m = cs[c + 1] - s;
        const uint64_t *srt = sc.eA + (size_t)lb * sc.S + s;
        const uint32_t *ent = sc.ent + (size_t)lb * sc.S + s;
        uint16_t *ce = sc.cand_e + (size_t)lb * sc.S + s;
        uint32_t pos[LZW_TINY], en[LZW_TINY];
#pragma unroll
        for (uint32_t k = 0; k < LZW_TINY; ++k) { pos[k] = 0; en[k] = 0; if (k < m) { pos[k] = (uint32_t)srt[k]; en[k] = ent[k]; } }
        bool anom_pending = (c == 0u) && sc.ncl[(size_t)lb * 4 + 1];
        uint32_t mask = 0, occ_id = 0, occ_en = 0, slots = 0, ev = 0;
        auto pos_of = [&](uint32_t k) -> uint32_t {
            uint32_t v = pos[0];
#pragma unroll
            for (uint32_t q = 1; q < LZW_TINY; ++q) v = (k == q) ? pos[q] : v;
            return v;
        };
#pragma unroll
        for (uint32_t k = 0; k < LZW_TINY; ++k) {
            if (k < m) {
                const uint32_t p = pos[k], r = en[k] & 0xFFFFu, id = en[k] >> 16;
                while (ev < k && (uint64_t)pos_of(ev) + W < (uint64_t)p) { mask &= ~(1u << ((slots >> (3u * ev)) & 7u)); ++ev; }   // FIFO retirement
                if (anom_pending && p > W - 1u) { mask &= ~1u; anom_pending = false; }
                uint32_t res = 0xFFFFu;
                for (uint32_t b = r; (mask >> b) & 1u; ++b)                     // bits >= m are never set: the walk ends inside the cluster
                    if (((occ_id >> (3u * b)) & 7u) == id) { res = (occ_en >> (3u * b)) & 7u; break; }
                const uint32_t b = r + (uint32_t)__builtin_ctz(~(mask >> r));   // first fit
                mask |= 1u << b;
                occ_id = (occ_id & ~(7u << (3u * b))) | (id << (3u * b));
                occ_en = (occ_en & ~(7u << (3u * b))) | (k << (3u * b));
                slots |= b << (3u * k);
                ce[k] = (uint16_t)res;
            }
        }
    }
}

// entry index -> position for every entry of a cluster the LDS replay handled (one thread per sorted entry, coalesced;
// singletons were answered by k_lzw_heads, clusters above the LDS classes by k_lzw_replay itself)
__global__ __launch_bounds__(256)
void k_lzw_resolve(uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0)
{
    const uint32_t lb = blockIdx.y;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint64_t *srt = sc.eA + (size_t)lb * sc.S;
    const uint32_t *cs = sc.cstart + (size_t)lb * (sc.S + 2);
    const uint16_t *ce = sc.cand_e + (size_t)lb * sc.S;
    uint32_t *cand = sc.cand + (size_t)lb * sc.S;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint64_t e = srt[k];
        const uint32_t g = (uint32_t)(e >> 32), s = cs[g], m = cs[g + 1] - s;
        if (m < 2 || m > LZW_CAP_L) continue;
        const uint32_t r = ce[k];
        cand[(uint32_t)e] = r == 0xFFFFu ? LZW_NONE : (uint32_t)srt[s + r];
    }
}

// coherent (L2) accesses to the table state this wave keeps rewriting
template <typename T> __device__ __forceinline__ T lzw_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void lzw_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wave per cluster of >= 2 entries: the literal table of lz77.c:55-108 restricted to the cluster's own buckets
__global__ __launch_bounds__(64)
void k_lzw_replay(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0, uint32_t nb)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t W = 1u << P.wbits;
    (void)nb;
    const uint32_t count = sc.ccount[4];                  // only clusters too large for the LDS replay (> 24 576 entries)
    for (uint32_t ci = blockIdx.x; ci < count; ci += gridDim.x) {
        const uint32_t lb = (uint32_t)(sc.clist[4][ci] >> 32), c_only = (uint32_t)sc.clist[4][ci];
        const uint64_t off = (block0 + lb) * (uint64_t)P.block;
        const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
        const uint8_t *src = in + off;
        const uint32_t ncl = sc.ncl[(size_t)lb * 4], zero_in_0 = sc.ncl[(size_t)lb * 4 + 1];
        const uint64_t *srt = sc.eA + (size_t)lb * sc.S;            // (cluster, time) order: three passes end in eB, see host
        const uint32_t *cs = sc.cstart + (size_t)lb * (sc.S + 2);
        const uint32_t *rd = sc.rd + (size_t)lb * sc.S;
        uint8_t  *live = sc.t_live + (size_t)lb * (sc.S + 64);
        uint32_t *tpos = sc.t_pos + (size_t)lb * sc.S, *tmix = sc.t_mix + (size_t)lb * sc.S, *slot_of = sc.slot_of + (size_t)lb * sc.S;
        uint32_t *cand = sc.cand + (size_t)lb * sc.S;
        (void)ncl;
        for (uint32_t c = c_only; c == c_only; ++c) {
            const uint32_t s = cs[c], e = cs[c + 1], m = e - s;
            if (m < 2) continue;
            // the cluster's dense buckets are [s, e): its head's dense slot is its own sorted index in the HOME order, which
            // the sweep used as the base — and the (cluster, time) order permutes entries only inside [s, e)
            const uint32_t lo = s, hi = e;
            uint32_t ev = s;
            bool anom_pending = (c == 0u) && zero_in_0;
            for (uint32_t k = s; k < e; ++k) {
                const uint32_t p = (uint32_t)srt[k];
                const uint32_t r = rd[p];
                const uint32_t x = lz_mix32(lzw_word(src, p, n));
                // FIFO retirement: insertion q + W clears the bucket insertion q wrote, whoever sits there now (lz77.c:70-76)
                while (ev < k) {
                    const uint32_t q = (uint32_t)srt[ev];
                    if ((uint64_t)q + W >= (uint64_t)p) break;
                    if (lane == 0) lzw_st<uint8_t>(&live[lzw_ld(&slot_of[ev])], (uint8_t)0);
                    ++ev;
                }
                // the ring starts zero-filled: insertion W-1 clears bucket 0 once
                if (anom_pending && p > W - 1u) { if (lane == 0) lzw_st<uint8_t>(&live[lo], (uint8_t)0); anom_pending = false; }
                __builtin_amdgcn_wave_barrier();
                // one walk serves find() and the insert: the first bucket that is empty or holds this word ends find();
                // the first empty bucket takes the entry.  Bucket `hi` is never occupied by this cluster (parking bound) and a
                // probe that reached it would find nothing of this word beyond: it reads as empty.
                uint32_t res = LZW_NONE, free_b = LZW_NONE;
                bool found_done = false;
                for (uint32_t b0 = r; free_b == LZW_NONE; b0 += 64u) {
                    const uint32_t b = b0 + lane;
                    const bool in = b < hi;
                    const uint32_t lv = in ? (uint32_t)lzw_ld(&live[b]) : 0u;
                    const uint32_t mx = (in && lv) ? lzw_ld(&tmix[b]) : 0u;
                    const uint64_t empty = __ballot(!lv);
                    const uint64_t hit = __ballot(lv && mx == x);
                    if (!found_done) {
                        const uint64_t stop = empty | hit;
                        if (stop) {
                            const uint32_t f = (uint32_t)__builtin_ctzll(stop);
                            if ((hit >> f) & 1ull) res = lzw_ld(&tpos[b0 + f]);
                            found_done = true;
                        }
                    }
                    if (empty) free_b = b0 + (uint32_t)__builtin_ctzll(empty);
                }
                if (lane == 0) {
                    cand[p] = res;
                    lzw_st<uint32_t>(&tpos[free_b], p);
                    lzw_st<uint32_t>(&tmix[free_b], x);
                    lzw_st<uint32_t>(&slot_of[k], free_b);
                    lzw_st<uint8_t>(&live[free_b], (uint8_t)1);
                }
                __threadfence();                          // the next entry's probes must see these stores
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// =============================================================================================
// parse + emit: one workgroup per block, 64 KiB segments in order
// =============================================================================================
__device__ __forceinline__ uint32_t lzw_select_bit(uint64_t m, uint32_t r)
{
    for (uint32_t k = 0; k < r; ++k) m &= m - 1;
    return (uint32_t)__builtin_ctzll(m);
}

__global__ __launch_bounds__(1024)
void k_lzw_parse_emit(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzwScratch sc, uint64_t block0)
{
    __shared__ uint8_t  s_ex[64 * 1024];             // exit tables [64][1024]; later {token base, match base, staging}
    __shared__ uint8_t  s_L[LZW_SEG];
    __shared__ uint64_t s_tok[1024], s_mat[1024];
    __shared__ uint8_t  s_entry[1024];
    __shared__ uint8_t  s_sexit[32][32];
    __shared__ uint8_t  s_sentry[33];
    __shared__ uint32_t s_scan[18];
    __shared__ uint64_t s_q0, s_q1;

    const int tid = threadIdx.x;
    const uint32_t lb = blockIdx.x;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    const uint32_t *cand = sc.cand + (size_t)lb * sc.S;
    uint32_t *slot = sc.slot + (size_t)lb * sc.slot_words;
    const uint32_t W = 1u << P.wbits, max_len = (1u << P.lbits) - 1u;      // max_len <= 31 (checked on the host)
    const uint32_t LB = 9u, MB = 1u + P.wbits + P.lbits;
    uint32_t e_in = 0;                                // offset at which the greedy chain enters the segment
    uint64_t qbase = 0;                               // bits emitted so far
    uint32_t carry = 0;

    for (uint32_t seg0 = 0; seg0 < n; seg0 += LZW_SEG) {
        const uint32_t ns = (n - seg0) < LZW_SEG ? (n - seg0) : LZW_SEG;
        // ---- A: token length at every position of the segment (bytes from HBM / L2; zero past the block end)
        for (uint32_t q = tid; q < ns; q += 1024u) {
            const uint32_t p = seg0 + q, c = cand[p];
            uint32_t len = 0;
            if (c != LZW_NONE && (p - c) != W) {                            // lz77.c:290: distance == W is a literal
                len = 4;
                while (len < max_len) {
                    const uint32_t x = lzw_word(src, c + len, n) ^ lzw_word(src, p + len, n);
                    if (x) { len += (uint32_t)__builtin_ctz(x) >> 3; break; }
                    len += 4;
                }
                if (len > max_len) len = max_len;
            }
            s_L[q] = (uint8_t)len;
        }
        __syncthreads();
        // ---- B: exit offset of every position of a 64-position chunk into the next chunk
        {
            const uint32_t c = tid;
            for (int o = 63; o >= 0; --o) {
                const uint32_t q = c * 64u + (uint32_t)o;
                uint32_t e = 0;
                if (q < ns) {
                    const uint32_t l = s_L[q];
                    const uint32_t nx = (uint32_t)o + (l ? l : 1u);
                    e = nx >= 64u ? nx - 64u : s_ex[nx * 1024u + c];
                }
                s_ex[(uint32_t)o * 1024u + c] = (uint8_t)e;
            }
        }
        __syncthreads();
        {
            const uint32_t scn = tid >> 5, o = tid & 31u;
            uint32_t x = o;
            for (uint32_t c = scn * 32u; c < scn * 32u + 32u; ++c) x = s_ex[x * 1024u + c];
            s_sexit[scn][o] = (uint8_t)x;
            __syncthreads();
            if (tid == 0) {
                uint32_t e = e_in;
                for (uint32_t s = 0; s < 32; ++s) { s_sentry[s] = (uint8_t)e; e = s_sexit[s][e]; }
                s_sentry[32] = (uint8_t)e;                                  // where the chain enters the NEXT segment
            }
            __syncthreads();
            if (tid < 32) {
                uint32_t xx = s_sentry[tid];
                for (uint32_t c = tid * 32u; c < tid * 32u + 32u; ++c) { s_entry[c] = (uint8_t)xx; xx = s_ex[xx * 1024u + c]; }
            }
            __syncthreads();
        }
        // ---- C: token starts of each chunk
        {
            const uint32_t c = tid;
            uint64_t tok = 0, mat = 0;
            if (c * 64u < ns) {
                uint32_t o = s_entry[c];
                while (o < 64u && c * 64u + o < ns) {
                    const uint32_t l = s_L[c * 64u + o];
                    tok |= 1ull << o;
                    if (l) mat |= 1ull << o;
                    o += l ? l : 1u;
                }
            }
            s_tok[c] = tok; s_mat[c] = mat;
        }
        __syncthreads();
        const uint32_t e_next = s_sentry[32];
        const uint64_t my_tok = s_tok[tid], my_mat = s_mat[tid];
        uint32_t ntok = 0, nmat = 0;
        const uint32_t tbase = block_exclusive_scan<uint32_t>((uint32_t)__popcll(my_tok), OpAddU32(), 0u, s_scan, &ntok);
        const uint32_t mbase = block_exclusive_scan<uint32_t>((uint32_t)__popcll(my_mat), OpAddU32(), 0u, s_scan, &nmat);
        uint32_t *tb = reinterpret_cast<uint32_t *>(s_ex);             // [1025]
        uint32_t *mb = tb + 1026;                                       // [1025]
        uint32_t *stage = mb + 1026;                                    // [4 * 1024 + 16]
        constexpr uint32_t TPR = 4;
        uint16_t *tch = reinterpret_cast<uint16_t *>(stage + TPR * 1024 + 16);   // [<= 1024] chunk that holds token 64 * k
        tb[tid] = tbase; mb[tid] = mbase;
        if (tid == 0) { tb[1024] = ntok; mb[1024] = nmat; }
        for (uint32_t mlt = (tbase + 63u) & ~63u; mlt < tbase + (uint32_t)__popcll(my_tok); mlt += 64u) tch[mlt >> 6] = (uint16_t)tid;
        __syncthreads();
        auto chunk_of = [&](uint32_t t) -> uint32_t { uint32_t c = tch[t >> 6]; while (tb[c + 1] <= t) ++c; return c; };
        // ---- D: emit, 4096 tokens per barrier round, appended behind the bits of the earlier segments
        for (uint32_t t0 = 0; t0 < ntok; t0 += TPR * 1024u) {
            uint64_t q[TPR]; uint32_t v[TPR], nbits[TPR]; bool valid[TPR];
#pragma unroll
            for (uint32_t u = 0; u < TPR; ++u) {
                const uint32_t t = t0 + u * 1024u + (uint32_t)tid;
                valid[u] = t < ntok; q[u] = 0; v[u] = 0; nbits[u] = 0;
                if (valid[u]) {
                    const uint32_t c = chunk_of(t), o = lzw_select_bit(s_tok[c], t - tb[c]);
                    const uint32_t p = seg0 + c * 64u + o;
                    const uint32_t mbefore = mb[c] + (uint32_t)__popcll(s_mat[c] & ((1ull << o) - 1ull));
                    q[u] = qbase + (uint64_t)(t - mbefore) * LB + (uint64_t)mbefore * MB;
                    if ((s_mat[c] >> o) & 1ull) {
                        const uint32_t d = p - cand[p], l = s_L[c * 64u + o];
                        v[u] = 1u | (d << 1) | (l << (1u + P.wbits)); nbits[u] = MB;     // flag, offset (wbits), length (lbits)
                    } else { v[u] = (uint32_t)src[p] << 1; nbits[u] = LB; }
                    if (u == 0 && tid == 0) s_q0 = q[0];
                    if (t == ntok - 1 || (u == TPR - 1 && tid == 1023)) s_q1 = q[u] + nbits[u];
                }
            }
            __syncthreads();
            const uint64_t q0 = s_q0, q1 = s_q1;
            const uint64_t w0 = q0 >> 5;
            const uint32_t nwords = (uint32_t)(((q1 + 31) >> 5) - w0);
            for (uint32_t i = tid; i < nwords + 1; i += 1024u) stage[i] = (i == 0) ? carry : 0u;
            __syncthreads();
#pragma unroll
            for (uint32_t u = 0; u < TPR; ++u) {
                if (valid[u]) {
                    const uint32_t rel = (uint32_t)(q[u] - (w0 << 5)), wi = rel >> 5, sh = rel & 31u;
                    atomicOr(&stage[wi], v[u] << sh);
                    if (sh + nbits[u] > 32u) atomicOr(&stage[wi + 1], v[u] >> (32u - sh));
                }
            }
            __syncthreads();
            const uint32_t ncomplete = (uint32_t)((q1 >> 5) - w0);
            for (uint32_t i = tid; i < ncomplete; i += 1024u) slot[w0 + i] = stage[i];
            carry = stage[ncomplete];
            __syncthreads();
        }
        qbase += (uint64_t)(ntok - nmat) * LB + (uint64_t)nmat * MB;
        e_in = e_next;
        __syncthreads();
    }
    if (tid == 0) {
        slot[qbase >> 5] = (qbase & 31u) ? carry : 0u;
        slot[(qbase >> 5) + 1] = 0;
        sc.block_bits[lb] = qbase;
    }
}

// =============================================================================================
// host side
// =============================================================================================
size_t lzw_scratch_bytes(uint32_t nb, uint32_t block)
{
    const size_t S = mi_align_up(block, 256);
    return (size_t)nb * (S * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 4 + 4 + 4 + 4 + 2 + 24) + (S + 64) + 8 * 4 + 16 + (size_t)LZW_SLOT_WORDS(block) * 4 + 8 + 4096) + 65536;
}

void lzw_carve(mi_ctx *ctx, uint32_t nb, uint32_t block, LzwScratch *sc)
{
    mi_carver cv(ctx->ws);
    const size_t S = mi_align_up(block, 256);
    sc->S = (uint32_t)S; sc->slot_words = LZW_SLOT_WORDS(block);
    sc->eA = cv.take<uint64_t>(nb * S); sc->eB = cv.take<uint64_t>(nb * S);
    sc->gid = cv.take<uint32_t>(nb * S); sc->rd = cv.take<uint32_t>(nb * S);
    sc->cstart = cv.take<uint32_t>(nb * (S + 2)); sc->ncl = cv.take<uint32_t>((size_t)nb * 4);
    sc->t_live = cv.take<uint8_t>(nb * (S + 64));
    sc->t_pos = cv.take<uint32_t>(nb * S); sc->t_mix = cv.take<uint32_t>(nb * S); sc->slot_of = cv.take<uint32_t>(nb * S);
    sc->cand = cv.take<uint32_t>(nb * S);
    sc->ent = cv.take<uint32_t>(nb * S); sc->relw = cv.take<uint32_t>(nb * S); sc->cand_e = cv.take<uint16_t>(nb * S);
    for (int c = 0; c < 6; ++c) sc->clist[c] = cv.take<uint64_t>(nb * S / 2 + 64);
    sc->ccount = cv.take<uint32_t>(64);
    sc->slot = cv.take<uint32_t>((size_t)nb * sc->slot_words);
    sc->block_bits = cv.take<uint64_t>(nb + 1);
}

// blocks per batch.  The workspace is ~80 bytes per input byte (lzw_scratch_bytes: S * 78 + the table bytes + the output
// slot), so 256 MiB of input per batch — what keeps the latency chains of the time-sliced finder long enough to fill the
// chip — is ~21 GiB of workspace: nothing beside 288 GB of HBM, too much for a card that is nearly full.  The batch is
// therefore also held to half of the memory that is free right now, and the caller halves it again on MI_ERR_NOMEM.
uint32_t lzw_batch_blocks(mi_ctx *ctx, uint64_t nblocks, uint32_t block)
{
    uint64_t cap = (256ull << 20) / block;                                 // 256 MiB of input per batch
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t fit = (uint64_t)(free_b / 2 + (ctx ? ctx->ws_bytes : 0)) / (lzw_scratch_bytes(1, block) - 65536);   // (the workspace it already holds is reused)
        if (fit < cap) cap = fit;
    } else (void)hipGetLastError();
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    return (uint32_t)(nblocks < cap ? (nblocks ? nblocks : 1) : cap);
}

// find() at every position of blocks [block0, block0 + nb): results in sc.cand (by position, LZW_NONE = none)
mi_status lzw_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                   const LzwScratch &sc, hipStream_t s)
{
    const uint32_t chunks = (P.block + 256u * 16u - 1u) / (256u * 16u);
    const uint32_t Tmask = (1u << P.tbits) - 1u;
    const uint32_t np_home = (P.tbits + 7u) / 8u;                          // passes over the home bits
    MI_HIP(ctx, hipMemsetAsync(sc.t_live, 0, (size_t)nb * (sc.S + 64), s));
    { mi_prof_scope p(ctx, "k_lzw_keys", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_keys, dim3(chunks, nb), dim3(256), 0, s, d_in, n, P, sc, block0); }
    { mi_prof_scope p(ctx, "k_lzw_sort(home)", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_sort, dim3(nb), dim3(1024), 0, s, n, P, sc, block0, Tmask, np_home, 0); }
    { mi_prof_scope p(ctx, "k_lzw_sweep", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_sweep, dim3(nb), dim3(1024), 0, s, n, P, sc, block0, (int)(np_home & 1u)); }
    // cluster numbers are < 2^20 (<= one per position): three passes, starting again from eA: result in eB
    { mi_prof_scope p(ctx, "k_lzw_sort(cluster)", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_sort, dim3(nb), dim3(1024), 0, s, n, P, sc, block0, 0xFFFFFFFFu, 3u, 1); }
    // the replay reads the (cluster, time) order from eA: copy it there (three passes end in eB)
    MI_HIP(ctx, hipMemcpyAsync(sc.eA, sc.eB, (size_t)nb * sc.S * 8, hipMemcpyDeviceToDevice, s));
    MI_HIP(ctx, hipMemsetAsync(sc.ccount, 0, 64 * 4, s));
    { mi_prof_scope p(ctx, "k_lzw_heads", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_heads, dim3(chunks, nb), dim3(256), 0, s, n, P, sc, block0, 0);
      hipLaunchKernelGGL(k_lzw_classify, dim3(chunks, nb < 256 ? nb : 256), dim3(256), 0, s, sc, nb); }
    // The size classes are independent: their replays run side by side on the context's other streams (a launch lasts as
    // long as its longest chain — one wave — so running them one after the other adds the tails up), joined before the
    // resolve pass.
    hipStream_t s1 = ctx->side ? ctx->side : s, s2 = ctx->parse ? ctx->parse : s, s3 = ctx->fb ? ctx->fb : s;
    MI_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
    if (s1 != s) MI_HIP(ctx, hipStreamWaitEvent(s1, ctx->ev_fork, 0));
    if (s2 != s) MI_HIP(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
    if (s3 != s) MI_HIP(ctx, hipStreamWaitEvent(s3, ctx->ev_fork, 0));
    { mi_prof_scope p(ctx, "k_lzw_replay(global)", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_replay, dim3((unsigned)ctx->num_cu), dim3(64), 0, s, d_in, n, P, sc, block0, nb); }
    { mi_prof_scope p(ctx, "k_lzw_replay<24576>", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lzw_replay_lds<LZW_CAP_L, 12>), dim3((unsigned)ctx->num_cu), dim3(64), 0, s, P, sc, 3u); }
    { mi_prof_scope p(ctx, "k_lzw_replay<8192>", s1, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lzw_replay_lds<LZW_CAP_K, 4>), dim3((unsigned)ctx->num_cu * 3u), dim3(64), 0, s1, P, sc, 2u); }
    { mi_prof_scope p(ctx, "k_lzw_replay<4096>", s2, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lzw_replay_lds<LZW_CAP_M, 2>), dim3((unsigned)ctx->num_cu * 6u), dim3(64), 0, s2, P, sc, 1u); }
    { mi_prof_scope p(ctx, "k_lzw_replay_tiny", s3, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_replay_tiny, dim3((unsigned)ctx->num_cu * 16u), dim3(256), 0, s3, P, sc); }
    { mi_prof_scope p(ctx, "k_lzw_replay<1024>", s3, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lzw_replay_lds<LZW_CAP_S, 1>), dim3((unsigned)ctx->num_cu * 64u), dim3(64), 0, s3, P, sc, 0u); }
    if (s1 != s) { MI_HIP(ctx, hipEventRecord(ctx->ev_replay[0], s1)); MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_replay[0], 0)); }
    if (s2 != s) { MI_HIP(ctx, hipEventRecord(ctx->ev_replay[1], s2)); MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_replay[1], 0)); }
    if (s3 != s) { MI_HIP(ctx, hipEventRecord(ctx->ev_replay[2], s3)); MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_replay[2], 0)); }
    { mi_prof_scope p(ctx, "k_lzw_resolve", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL(k_lzw_resolve, dim3(chunks, nb), dim3(256), 0, s, n, P, sc, block0); }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// The time-sliced LDS-resident finder (lzs.hip) first; a batch with a block it had to flag (a cluster above its capacity: long
// runs of one byte value) is redone here, whole-block clusters of any size.
bool      lzs_applicable(const LzP &P);
mi_status lzs_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                   const LzwScratch &ws, hipStream_t s, uint32_t *flagged, const uint32_t **flag_list);

// the workspace as block `lb` of the batch sees it: a one-block run of the whole-block finder then works in that block's own rows
static LzwScratch lzw_view_at(const LzwScratch &ws, uint32_t lb)
{
    LzwScratch v = ws;
    const size_t S = ws.S, o = (size_t)lb * S;
    v.eA += o; v.eB += o; v.gid += o; v.rd += o; v.cstart += (size_t)lb * (S + 2); v.ncl += (size_t)lb * 4;
    v.t_live += (size_t)lb * (S + 64); v.t_pos += o; v.t_mix += o; v.slot_of += o; v.cand += o; v.ent += o; v.relw += o; v.cand_e += o;
    v.slot += (size_t)lb * ws.slot_words; v.block_bits += lb;
    return v;                                            // (the class lists and their counters are shared: lzw_find resets them)
}

mi_status lzw_or_lzs_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb, const LzwScratch &sc, hipStream_t s)
{
    if (lzs_applicable(P)) {
        uint32_t flagged = 0;
        const uint32_t *list = nullptr;
        const mi_status st = lzs_find(ctx, P, d_in, n, block0, nb, sc, s, &flagged, &list);
        if (st) return st;
        if (!flagged) return MI_OK;
        if (list && flagged <= nb / 4u + 1u) {           // a few blocks with a giant cluster: only they are redone
            uint32_t todo[64];
            const uint32_t k = flagged < 64u ? flagged : 64u;
            if (flagged <= 64u) {
                for (uint32_t i = 0; i < k; ++i) todo[i] = list[i];          // (the pinned list is reused by later calls)
                if (getenv("MI_LZ_DEBUG")) fprintf(stderr, "lzs: %u of %u blocks flagged, redone alone by the whole-block finder (first: %u)\n", flagged, nb, todo[0]);
                for (uint32_t i = 0; i < k; ++i) {
                    if (todo[i] >= nb) return MI_ERR_HIP;
                    const mi_status s2 = lzw_find(ctx, P, d_in, n, block0 + todo[i], 1u, lzw_view_at(sc, todo[i]), s);
                    if (s2) return s2;
                }
                return MI_OK;
            }
        }
    }
    return lzw_find(ctx, P, d_in, n, block0, nb, sc, s);
}

void lzw_launch_parse_emit(const uint8_t *d_in, uint64_t n, const LzP &P, const LzwScratch &sc, uint64_t block0, uint32_t nb, hipStream_t s)
{
    hipLaunchKernelGGL(k_lzw_parse_emit, dim3(nb), dim3(1024), 0, s, d_in, n, P, sc, block0);
}

