// lz_find.hip — the reference's hash-table match finder, replayed without the table.
//
// Replaces, per block: hash (algorithms/lz77/lz77.c:13-41), insert_hash_table (:55-86,
// deflate variant algorithms/deflate/lz77.c:77-145) and find (:94-108 / deflate :147-174)
// as driven by lz77_compress (lz77.c:281-338 / deflate lz77.c:215-275).
//
//   k_lz_sort_home       block -> LDS; positions sorted by (home bucket, time) with three
//                        stable 8-bit radix passes (a pass that would not move anything stores
//                        nothing); probe clusters by a prefix-max ("parking") sweep; per position
//                        {cluster, dense home index, word id}
//   k_lz_sort_cluster    entries sorted by (cluster, time): two stable 8-bit radix passes
//   k_lz_emulate         a tile of clusters in LDS; clusters of one word by a closed form, mixed
//                        clusters of < 16 entries one per LANE off a dense list, of 16..127
//                        entries one per WAVE of the workgroup (lz_replay.h), in time order: FIFO
//                        eviction, find, first-fit insert on an occupancy bitmap
//   k_lz_emulate_dom     giant clusters that one word dominates, off a cursor (lz_dom.h)
//   k_lz_emulate_giant   what is left of the giant list: one workgroup per cluster, a 24 KiB
//                        instance for <= 4096 entries in front of the 128 KiB one
//
// Output: cand[p] = what find(word at p) returns in the table state after positions 0..p-1
// were inserted (0xFFFF = none).  The parse consumes it in lz_emit.hip.
#include "lz_common.h"
#include "lz_replay.h"
#include "lz2.h"
#include <stdlib.h>

// =============================================================================================
// k_lz_sort_home
// =============================================================================================
// As the fallback of the LDS-resident finder these kernels normally have nothing to do, but every workgroup of a launch
// must still be given its 82..155 KiB of LDS before it can find that out, and it takes that LDS from k_lz2_find
// (rocprofv3, round 1: 1024-workgroup empty launches spent 4.8 ms each queueing).  So the bodies loop over the listed
// blocks and the fallback launches use a small grid (LZ_FB_GRID workgroups).
#define LZ_FB_GRID 128u
#define DOM_DONE 0x80000000u                          // giant_list entry handled by k_lz_emulate_dom: k_lz_emulate_giant skips it
__device__ __forceinline__ void lz_sort_home_block(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzScratch sc, uint64_t block0,
                                                   uint32_t lb)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[LZ_MAX_BLOCK + LZ_TAIL + 16];
    __shared__ uint32_t s_cnt[17][256];
    __shared__ int32_t  s_i32[18];
    __shared__ uint32_t s_u32[18];

    const int tid = threadIdx.x;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    const uint32_t T = 1u << P.tbits, Tmask = T - 1u;

    // ---- block -> LDS, zero tail (the reference reads past `size`; the parity definition is zeros)
    lz_block_to_lds(s_in, src, n, (uint32_t)tid);
    if (tid < 16) s_in[LZ_MAX_BLOCK + LZ_TAIL + tid] = 0;
    __syncthreads();

    uint16_t *A = sc.posA + (size_t)lb * LZ_MAX_BLOCK;
    uint16_t *B = sc.posB + (size_t)lb * LZ_MAX_BLOCK;
    auto home_of = [&](uint32_t p) -> uint32_t { return lz_mix32(lds_word(s_in, p)) & Tmask; };

    // ---- sort positions by home, stable in time.  A pass in which every position has the same digit (a block of one byte value:
    //      all three) stores nothing and the next one reads what this one would have read: `cur` = the current order (nullptr: the
    //      identity), a pass writes to the array `cur` is not
    const uint16_t *cur = nullptr;
    for (uint32_t pass = 0; pass < 3; ++pass) {
        uint16_t *dst = (cur == A) ? B : A;
        const uint32_t sh = 8u * pass;
        const bool same = radix_pass_1024_or_skip<8, uint32_t>(n, s_cnt,
            [&](uint32_t i) { return cur ? (uint32_t)cur[i] : i; },
            [&](uint32_t e) { return (home_of(e) >> sh) & 255u; },
            [&](uint32_t j, uint32_t e) { dst[j] = (uint16_t)e; });
        if (!same) cur = dst;
    }

    // ---- cluster sweep over the sorted order (logical index k; `rot` rotates the order when a
    //      deflate-style cluster wraps past bucket T-1 into bucket 0)
    const uint32_t k0 = tid * 64u, k1 = (k0 + 64u < n) ? k0 + 64u : n;
    uint32_t rot = 0;
    uint32_t gid_base = 0, ngroups = 0;
    int32_t  premax = INT32_MIN, gs_carry = -1, hs_carry = -1;
    int32_t  gmax_total = INT32_MIN;
    for (int iter = 0; iter < 2; ++iter) {
        auto phys = [&](uint32_t k) { uint32_t p = k + rot; return p >= n ? p - n : p; };
        auto hk_at = [&](uint32_t k, uint32_t &pos) -> int32_t {
            const uint32_t ph = phys(k);
            pos = cur ? (uint32_t)cur[ph] : ph;
            int32_t h = (int32_t)home_of(pos);
            if (rot && ph >= rot) h -= (int32_t)T;
            return h;
        };
        // the chunk's sorted positions, in order: fn(k, position).  A whole chunk of an unrotated order comes in as 16-byte
        // pieces, the next one in flight while eight entries are worked on (element by element every entry of the three loops
        // below was its own HBM round trip — the stores of loop (c) keep the compiler from moving a load over them —, 192 in a
        // row per thread at one workgroup per CU: most of the 1.2 ms a block spent in this kernel, round 4)
        auto h_of = [&](uint32_t k, uint32_t pos) -> int32_t {
            int32_t h = (int32_t)home_of(pos);
            if (rot && phys(k) >= rot) h -= (int32_t)T;
            return h;
        };
        auto for_chunk = [&](auto &&fn) {
            if (k0 >= k1) return;
            if (rot == 0 && k1 - k0 == 64u && cur) {
                uint4 nv = *reinterpret_cast<const uint4 *>(cur + k0);
#pragma unroll 1
                for (uint32_t k = k0; k < k1; k += 8u) {
                    const uint4 v = nv;
                    nv = *reinterpret_cast<const uint4 *>(cur + ((k + 8u < k1) ? k + 8u : k));        // unconditional (clamped)
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) fn(k + u, (w[u >> 1] >> ((u & 1u) * 16u)) & 0xFFFFu);
                }
            } else {
                for (uint32_t k = k0; k < k1; ++k) { const uint32_t ph = phys(k); fn(k, cur ? (uint32_t)cur[ph] : ph); }
            }
        };
        // (a) chunk maximum of g = home - k
        int32_t m = INT32_MIN;
        for_chunk([&](uint32_t k, uint32_t pos) { const int32_t g = h_of(k, pos) - (int32_t)k; m = g > m ? g : m; });
        premax = block_exclusive_scan<int32_t>(m, OpMaxI32(), INT32_MIN, s_i32, &gmax_total);
        // (b) heads and home-run starts of the chunk
        uint32_t nheads = 0; int32_t lasthead = -1, lastrun = -1;
        {
            int32_t run = premax, prev_h = 0;
            if (k0 > 0 && k0 < n) { uint32_t pp; prev_h = hk_at(k0 - 1, pp); }
            for_chunk([&](uint32_t k, uint32_t pos) {
                const int32_t h = h_of(k, pos), g = h - (int32_t)k;
                const bool head = (k == 0) || (g >= run);
                run = g > run ? g : run;
                if (head) { ++nheads; lasthead = (int32_t)k; }
                if (k == 0 || h != prev_h) lastrun = (int32_t)k;
                prev_h = h;
            });
        }
        uint32_t total_heads; int32_t last_group_start, dummy;
        gid_base = block_exclusive_scan<uint32_t>(nheads, OpAddU32(), 0u, s_u32, &total_heads);
        gs_carry = block_exclusive_scan<int32_t>(lasthead, OpMaxI32(), -1, s_i32, &last_group_start);
        hs_carry = block_exclusive_scan<int32_t>(lastrun, OpMaxI32(), -1, s_i32, &dummy);
        ngroups = total_heads;
        const int64_t e_last = (int64_t)gmax_total + (int64_t)n;
        if (iter == 0 && P.deflate && e_last > (int64_t)T && last_group_start > 0) {
            rot = (uint32_t)last_group_start;           // uniform: every thread computes the same value
            continue;
        }
        // (c) emit one record per position
        {
            uint64_t *E = sc.eA + (size_t)lb * LZ_MAX_BLOCK;
            uint16_t *cand = sc.cand + (size_t)lb * LZ_MAX_BLOCK;
            int32_t run = premax, prev_h = 0;
            uint32_t cur_gs = 0, cur_gid = gid_base; int32_t cur_base = 0;
            uint32_t cur_hs = 0, hs_word = 0, hs_pos = 0;
            if (k0 < n) {
                uint32_t pp;
                if (k0 > 0) prev_h = hk_at(k0 - 1, pp);
                if (gs_carry >= 0) { cur_gs = (uint32_t)gs_carry; cur_base = hk_at(cur_gs, pp); cur_gid = gid_base - 1u; }
                if (hs_carry >= 0) { cur_hs = (uint32_t)hs_carry; (void)hk_at(cur_hs, hs_pos); hs_word = lds_word(s_in, hs_pos); }
            }
            uint32_t seen = 0;
            for_chunk([&](uint32_t k, uint32_t pos) {
                const int32_t h = h_of(k, pos), g = h - (int32_t)k;
                const bool head = (k == 0) || (g >= run);
                run = g > run ? g : run;
                const uint32_t w = lds_word(s_in, pos);
                if (head) { cur_gs = k; cur_base = h; cur_gid = gid_base + seen; ++seen; }
                uint32_t pid;
                if (k == 0 || h != prev_h) { cur_hs = k; hs_word = w; hs_pos = pos; pid = pos; }
                else if (w == hs_word) pid = hs_pos;
                else {                                     // two different words share a home bucket: rare
                    pid = pos;
                    for (uint32_t kk = cur_hs + 1; kk < k; ++kk) {
                        uint32_t p2; (void)hk_at(kk, p2);
                        if (lds_word(s_in, p2) == w) { pid = p2; break; }
                    }
                }
                prev_h = h;
                const uint32_t rloc = cur_gs + (uint32_t)(h - cur_base);
                E[pos] = lz_pack(cur_gid, pos, rloc, pid);
                cand[pos] = LZ_NONE16;
            });
        }
        if (tid == 0) {
            LzBlockMeta mt;
            mt.n = n; mt.ngroups = ngroups; mt.rot = rot;
            mt.anom_idx = ~0u; mt.limit_idx = ~0u;
            uint32_t p0; const int32_t h0 = hk_at(0, p0);
            const bool cross = P.deflate && (rot > 0 || (e_last > (int64_t)T && ngroups == 1));
            if (cross) {
                const uint32_t raw = home_of(p0);
                mt.anom_idx = T - raw; mt.limit_idx = T - raw;
            } else if (h0 == 0) {
                mt.anom_idx = 0;
            }
            mt.pad[0] = mt.pad[1] = mt.pad[2] = 0;
            sc.meta[lb] = mt;
        }
        break;
    }
}

__global__ __launch_bounds__(1024)
void k_lz_sort_home(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzScratch sc, uint64_t block0, uint32_t nb,
                    const uint32_t *__restrict__ blist, const uint32_t *__restrict__ bcount)
{
    const uint32_t count = bcount ? *bcount : nb;          // only the listed blocks when there is a list
    // how many blocks fell back, left where the host sees it (pinned word behind the order flag): LATER launches size their
    // fallback grids by it — no synchronisation; the batches of one call are all queued before the first runs, so the hint a
    // call reads is what an earlier call left (lz_emit.hip)
    if (bcount && blockIdx.x == 0 && threadIdx.x == 0 && P.order_flag) { P.order_flag[1] = count; P.order_flag[2] = nb; }
    for (uint32_t bi = blockIdx.x; bi < count; bi += gridDim.x) {
        lz_sort_home_block(in, n_total, P, sc, block0, blist ? blist[bi] : bi);
        __syncthreads();
    }
}

// =============================================================================================
// k_lz_sort_cluster: (cluster, time) order.  Input records sit at index = position, i.e. in
// time order, so two stable passes over the 16-bit cluster number are enough.
// =============================================================================================
__global__ __launch_bounds__(1024)
void k_lz_sort_cluster(LzScratch sc, uint32_t nb, const uint32_t *__restrict__ blist, const uint32_t *__restrict__ bcount)
{
    __shared__ uint32_t s_cnt[17][256];
    const uint32_t count = bcount ? *bcount : nb;
    for (uint32_t bi = blockIdx.x; bi < count; bi += gridDim.x) {
    const uint32_t lb = blist ? blist[bi] : bi;
    const uint32_t n = sc.meta[lb].n;
    if (sc.meta[lb].ngroups <= 1u) continue;              // ONE cluster (a block of one byte value): position order is (cluster, time) order already
    uint64_t *A = sc.eA + (size_t)lb * LZ_MAX_BLOCK;
    uint64_t *B = sc.eB + (size_t)lb * LZ_MAX_BLOCK;
    radix_pass_1024<8, uint64_t>(n, s_cnt,
        [&](uint32_t i) { return A[i]; },
        [&](uint64_t e) { return (uint32_t)e & 255u; },
        [&](uint32_t j, uint64_t e) { B[j] = e; });
    radix_pass_1024<8, uint64_t>(n, s_cnt,
        [&](uint32_t i) { return B[i]; },
        [&](uint64_t e) { return ((uint32_t)e >> 8) & 255u; },
        [&](uint32_t j, uint64_t e) { A[j] = e; });
    }
}

// =============================================================================================
// cluster replay
// =============================================================================================
// LDS image of a set of clusters, indices are dense "slots": entry i's home bucket is slot
// rs[i] (before its insertion; afterwards rs[i] holds the slot it occupies).  A cluster of m
// entries owns exactly m consecutive slots and can never need more (parking bound).
struct TileView {
    uint16_t *pos, *rs, *pid, *occ;
    uint32_t *bm;           // occupancy bits, one per slot
    uint32_t *bm1;          // one bit per bm word: "word is full"; kept only for words that lie entirely
                            // inside one cluster (their owner lane is the only writer), so it is exact
    uint32_t  n;            // entries (= slots) in the view
};
#define RS_HEAD 0x8000u
#define RS_MASK 0x7FFFu

__device__ __forceinline__ bool bm_test(const uint32_t *bm, uint32_t b) { return (bm[b >> 5] >> (b & 31u)) & 1u; }

// first zero bit at or after r (exists inside the cluster by the parking bound).  Full words
// that the summary level knows about are skipped 32 at a time: a pile of several hundred
// copies of one word costs a couple of reads per insert instead of a walk over the pile.
__device__ __forceinline__ uint32_t bm_next_zero(const uint32_t *bm, const uint32_t *bm1, uint32_t r)
{
    uint32_t wi = r >> 5;
    uint32_t w = bm[wi] | ((1u << (r & 31u)) - 1u);
    if (w != 0xFFFFFFFFu) return (wi << 5) + (uint32_t)__builtin_ctz(~w);
    ++wi;
    for (;;) {
        const uint32_t sh = wi & 31u;
        const uint32_t notfull = ~(bm1[wi >> 5] >> sh) & (sh ? ((1u << (32u - sh)) - 1u) : 0xFFFFFFFFu);
        if (!notfull) { wi = (wi | 31u) + 1u; continue; }
        wi += (uint32_t)__builtin_ctz(notfull);
        w = bm[wi];
        if (w != 0xFFFFFFFFu) return (wi << 5) + (uint32_t)__builtin_ctz(~w);
        ++wi;                                    // a word shared with a neighbour cluster: full but unmarked
    }
}

// replay entries [s, e) of the view (one cluster) in time order.  anom / limit: slot of bucket 0 /
// bucket T for the cluster that contains them (else ~0u).
__device__ void replay_cluster(const TileView &v, uint32_t s, uint32_t e, uint32_t W, uint32_t anom, uint32_t limit,
                               uint16_t *__restrict__ cand)
{
    // bm words [wlo, whi) belong to this cluster alone
    const uint32_t wlo = (s + 31u) >> 5, whi = e >> 5;
    auto clear_slot = [&](uint32_t b) {
        const uint32_t wi = b >> 5, bit = 1u << (b & 31u);
        if (wi >= wlo && wi < whi) {
            const uint32_t w = v.bm[wi];
            v.bm[wi] = w & ~bit;
            if (w == 0xFFFFFFFFu) atomicAnd(&v.bm1[wi >> 5], ~(1u << (wi & 31u)));
        } else {
            atomicAnd(&v.bm[wi], ~bit);
        }
    };
    auto set_slot = [&](uint32_t b) {
        const uint32_t wi = b >> 5, bit = 1u << (b & 31u);
        if (wi >= wlo && wi < whi) {
            const uint32_t w = v.bm[wi] | bit;
            v.bm[wi] = w;
            if (w == 0xFFFFFFFFu) atomicOr(&v.bm1[wi >> 5], 1u << (wi & 31u));
        } else {
            atomicOr(&v.bm[wi], bit);
        }
    };
    uint32_t ev = s;
    bool anom_pending = anom != ~0u;
    for (uint32_t i = s; i < e; ++i) {
        const uint32_t p = v.pos[i];
        const uint32_t rsv = v.rs[i];
        const uint32_t r = rsv & RS_MASK;
        const uint32_t pid = v.pid[i];
        // FIFO eviction: insertion k retires insertion k-W *after* writing, so what find/insert at
        // p see is everything inserted at or after p-W  (lz77.c:70-76).  It clears the BUCKET,
        // whoever sits there.
        while (ev < i && (uint32_t)v.pos[ev] + W < p) { clear_slot(v.rs[ev] & RS_MASK); ++ev; }
        // the ring starts zero-filled, so insertion W-1 clears bucket 0 once
        if (anom_pending && p > W - 1u) { clear_slot(anom); anom_pending = false; }
        // find: first slot from the home that is empty (-> none) or holds the same word
        uint32_t res = LZ_NONE16;
        for (uint32_t b = r;; ++b) {
            if (b == limit && r < limit) break;                 // deflate find() does not wrap past T-1
            if (!bm_test(v.bm, b)) break;
            const uint32_t o = v.occ[b];
            if (v.pid[o] == pid) { res = v.pos[o]; break; }
        }
        if (res != LZ_NONE16) cand[p] = (uint16_t)res;
        // insert: first free slot from the home
        const uint32_t b = bm_next_zero(v.bm, v.bm1, r);
        set_slot(b);
        v.occ[b] = (uint16_t)i;
        v.rs[i] = (uint16_t)((rsv & RS_HEAD) | b);
    }
}

__device__ __forceinline__ void lz_emulate_tile(LzP P, LzScratch sc, uint32_t lb, uint32_t t, uint64_t *dbg)
{
    long long tkt = dbg ? clock64() : 0;
#define TL_TICK(k) do { if (dbg && threadIdx.x == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&dbg[40 + (k)], (unsigned long long)(t2 - tkt)); if ((k) >= 2) atomicMax((unsigned long long *)&dbg[43 + (k)], (unsigned long long)(t2 - tkt)); tkt = t2; } } while (0)
    __shared__ uint16_t s_pos[LZ_TILE_CAP], s_rs[LZ_TILE_CAP], s_pid[LZ_TILE_CAP], s_occ[LZ_TILE_CAP];
    __shared__ uint32_t s_bm[LZ_TILE_CAP / 32 + 2];
    __shared__ uint32_t s_bm1[LZ_TILE_CAP / 1024 + 2];
    __shared__ uint32_t s_a, s_b, s_lasthead;
    // clusters of LZ_TILE_WAVE_MIN .. LZ_WAVE_MIN - 1 entries that are not one word: a list for the workgroup's eight waves, and a
    // wave's replay tables (slot -> occupant, entry -> slot, relative home slots, results): 1.25 KiB per wave
    __shared__ uint32_t s_wl[LZ_TILE_CAP / LZ_TILE_WAVE_MIN + 1], s_nwl;
    __shared__ uint32_t s_wv[8 * LZ_WAVE_MIN * 5 / 2];                       // per wave: occupants (u32), slots, homes, results (u16)
    uint32_t (*s_wocc)[LZ_WAVE_MIN] = reinterpret_cast<uint32_t (*)[LZ_WAVE_MIN]>(s_wv);
    uint16_t (*s_wslot)[LZ_WAVE_MIN] = reinterpret_cast<uint16_t (*)[LZ_WAVE_MIN]>(s_wv + 8 * LZ_WAVE_MIN);
    uint16_t (*s_wrs)[LZ_WAVE_MIN] = s_wslot + 8, (*s_wc)[LZ_WAVE_MIN] = s_wslot + 16;
    // the tile's cluster heads, one bit per entry, and — before the waves need their tables — the heads of the clusters the lanes
    // work on, densely: taken by entry index a lane found a head at one in ten of its entries and a wave waited eight times for
    // its longest cluster (round 4: 216 k of a tile's 380 k cycles on "pages")
    __shared__ uint32_t s_hb[LZ_TILE_CAP / 32 + 2];
    uint16_t *s_item = reinterpret_cast<uint16_t *>(s_wv);
    __shared__ uint32_t s_nitem;
    static_assert(sizeof(s_wv) >= (LZ_TILE_CAP / 2) * sizeof(uint16_t), "a cluster on the lanes' list has two entries at least");

    const int tid = threadIdx.x;
    const LzBlockMeta mt = sc.meta[lb];
    const uint32_t n = mt.n;
    const uint32_t lo = t * LZ_TILE_NOM, hi = (lo + LZ_TILE_NOM < n) ? lo + LZ_TILE_NOM : n;
    if (lo >= n) return;
    const uint64_t *E = sc.eA + (size_t)lb * LZ_MAX_BLOCK;

    if (tid == 0) { s_a = ~0u; s_b = ~0u; s_lasthead = 0; s_nwl = 0; s_nitem = 0; }
    __syncthreads();
    // Cluster heads of a stretch of up to 4 096 + LZ_GIANT_MIN sorted entries, eight per thread: the cluster numbers of entry i and
    // of i - 1 come in as UNCONDITIONAL loads (indices clamped), all sixteen of a thread before the first compare.  (Written as
    // "for (i ...) if (is_head(i)) ..." every iteration was a dependent HBM round trip — a load under a condition waits for its value
    // in its own basic block — and a tile did ~40 of them in a row before its replay began: round 4, on the way to "pages".)
    auto heads_of = [&](uint32_t first, uint32_t end, auto &&fn) {            // fn(i) for every head i in [first, end), end - first <= 8 * 512
        uint32_t g[8], gp[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            uint32_t i = first + tid + 512u * u;
            if (i >= n) i = n - 1u;
            g[u] = (uint32_t)E[i] & 0xFFFFu;
            gp[u] = (uint32_t)E[i ? i - 1u : 0u] & 0xFFFFu;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = first + tid + 512u * u;
            if (i < end && (i == 0 || g[u] != gp[u])) fn(i);
        }
    };
    // a = first head in [lo, hi)
    // (one LDS atomic per wave: a tile of a text block has ~4 000 heads, and as many atomics on ONE address are served one by one)
    auto wave_min = [&](uint32_t v) { for (int o = 32; o; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t < v ? t : v; } return v; };
    auto wave_max = [&](uint32_t v) { for (int o = 32; o; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; } return v; };
    {
        uint32_t mn = ~0u;
        heads_of(lo, hi, [&](uint32_t i) { mn = i < mn ? i : mn; });
        mn = wave_min(mn);
        if ((tid & 63) == 0 && mn != ~0u) atomicMin(&s_a, mn);
    }
    __syncthreads();
    const uint32_t a = s_a;
    if (a == ~0u) return;                       // a cluster that started earlier covers this whole range
    // b = first head at or after hi (or n); give up after LZ_GIANT_MIN entries: the last cluster is a giant
    static_assert(LZ_TILE_NOM <= 8 * 512 && LZ_GIANT_MIN + 1 <= 8 * 512 + 1, "heads_of covers eight entries per thread");
    {
        const uint32_t end2 = (n < hi + LZ_GIANT_MIN + 1u) ? n : hi + LZ_GIANT_MIN + 1u;
        uint32_t mn = ~0u;
        heads_of(hi, end2 < hi + 8u * 512u ? end2 : hi + 8u * 512u, [&](uint32_t i) { mn = i < mn ? i : mn; });
        mn = wave_min(mn);
        if ((tid & 63) == 0 && mn != ~0u) atomicMin(&s_b, mn);
        if (end2 > hi + 8u * 512u && tid == 0) {                              // (the one index beyond eight per thread)
            const uint32_t i = hi + 8u * 512u;
            if (((uint32_t)E[i] & 0xFFFFu) != ((uint32_t)E[i - 1u] & 0xFFFFu)) atomicMin(&s_b, i);
        }
    }
    {
        uint32_t mx = 0;
        heads_of(a, hi, [&](uint32_t i) { mx = i > mx ? i : mx; });
        mx = wave_max(mx);
        if ((tid & 63) == 0) atomicMax(&s_lasthead, mx);
    }
    __syncthreads();
    uint32_t b = s_b;
    if (b == ~0u) b = (n <= hi + LZ_GIANT_MIN) ? n : ~0u;
    const uint32_t lasthead = s_lasthead;
    if (b == ~0u || b - lasthead > LZ_GIANT_MIN || b - a > LZ_TILE_CAP) {
        // the cluster that starts at `lasthead` is too large for this tile: hand it to the giant kernel
        if (tid == 0) {
            const uint32_t k = sc.giant_cap - 1u - atomicAdd(&sc.giant_count[2], 1u);      // from the end: handed out first (lz_giant_slot)
            sc.giant_list[2 * k] = lb; sc.giant_list[2 * k + 1] = lasthead;
        }
        b = lasthead;
    }
    const uint32_t m = b - a;
    if (m == 0) return;
    TL_TICK(0);
    // ---- load the tile: sixteen entries per thread at most, eight at a time, loads first (unconditional, clamped)
    for (uint32_t i0 = 0; i0 < m; i0 += 8u * 512u) {
        uint64_t ev[8]; uint32_t pg[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            uint32_t i = i0 + tid + 512u * u;
            if (i >= m) i = m - 1u;
            ev[u] = E[a + i];
            pg[u] = (uint32_t)E[a + (i ? i - 1u : 0u)] & 0xFFFFu;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = i0 + tid + 512u * u;
            bool head = false;
            if (i < m) {
                const uint64_t e = ev[u];
                const uint32_t gid = (uint32_t)e & 0xFFFFu;
                head = (i == 0) || (pg[u] != gid);
                s_pos[i] = (uint16_t)(e >> 16);
                s_rs[i] = (uint16_t)((((uint32_t)(e >> 32) & 0xFFFFu) - a) | (head ? RS_HEAD : 0u));
                s_pid[i] = (uint16_t)(e >> 48);
            }
            const uint64_t hm = __ballot(head);                       // the wave's 64 consecutive entries (i - lane is a multiple of 64)
            if ((tid & 63) == 0 && i < ((m + 63u) & ~63u)) { s_hb[i >> 5] = (uint32_t)hm; s_hb[(i >> 5) + 1u] = (uint32_t)(hm >> 32); }
        }
    }
    for (uint32_t i = tid; i < LZ_TILE_CAP / 32 + 2; i += 512) s_bm[i] = 0;
    if (tid < LZ_TILE_CAP / 1024 + 2) s_bm1[tid] = 0;
    __syncthreads();
    TL_TICK(1);
    // pid is a POSITION (first occurrence of the word); compare through it directly
    TileView v{s_pos, s_rs, s_pid, s_occ, s_bm, s_bm1, m};
    const uint32_t W = 1u << P.wbits;
    uint16_t *cand = sc.cand + (size_t)lb * LZ_MAX_BLOCK;
    // end of the cluster that starts at head s: the next head bit, or m
    auto cluster_end = [&](uint32_t s) -> uint32_t {
        uint32_t wi = (s + 1u) >> 5;
        uint32_t w = s_hb[wi] & ~((1u << ((s + 1u) & 31u)) - 1u);
        const uint32_t wend = (m + 31u) >> 5;
        while (!w && ++wi < wend) w = s_hb[wi];
        const uint32_t e = w ? (wi << 5) + (uint32_t)__builtin_ctz(w) : m;
        return e < m ? e : m;
    };
    for (uint32_t s = tid; s < m; s += 512) {
        if (!(s_rs[s] & RS_HEAD)) continue;
        const uint32_t e = cluster_end(s);
        if (e - s < 2) continue;                 // a lone entry finds nothing and blocks nobody
        if (e - s >= LZ_WAVE_MIN) {              // a wave replays it (k_lz_emulate_giant): a lane would walk its probe chains bucket by bucket
            const uint32_t k = atomicAdd(sc.giant_count, 1u);
            sc.giant_list[2 * k] = lb; sc.giant_list[2 * k + 1] = a + s;
            continue;
        }
        s_item[atomicAdd(&s_nitem, 1u)] = (uint16_t)s;
    }
    __syncthreads();
    const uint32_t nitem = s_nitem;
    for (uint32_t q = tid; q < nitem; q += 512) {
        const uint32_t s = s_item[q], e = cluster_end(s);
        const bool first = (a + s) == 0;
        // a cluster of ONE word (most of them): find() only ever looks at the home slot, whose occupant — the anchor — is the
        // first entry, then the first entry after anchor + W, which itself finds the slot empty: one pass over
        // the positions, no table.  Not for the cluster that covers bucket 0 / T.
        if (!(first && (mt.anom_idx != ~0u || mt.limit_idx != ~0u))) {
            const uint32_t id0 = s_pid[s];
            bool one = true;
            for (uint32_t i = s + 1; i < e && one; ++i) one = s_pid[i] == id0;
            if (one) {
                uint32_t an = s_pos[s];
                for (uint32_t i = s + 1; i < e; ++i) {
                    const uint32_t p = s_pos[i];
                    if (an + W < p) an = p; else cand[p] = (uint16_t)an;
                }
                continue;
            }
        }
        // A mixed cluster of LZ_TILE_WAVE_MIN .. LZ_WAVE_MIN - 1 entries waits for a WAVE of this workgroup (below): on a lane its
        // probe walks are three dependent LDS reads per bucket (occupancy bit -> occupant -> its word id), ~4.5 k cycles per entry
        // on the "pages" family (phase counters, round 4: the longest such lane was 458 k of a tile's 525 k cycles); the wave
        // replay of lz_replay.h looks at 64 buckets per step with the occupancy in registers.
        if (e - s >= LZ_TILE_WAVE_MIN) { s_wl[atomicAdd(&s_nwl, 1u)] = s | ((e - s) << 16); continue; }
        replay_cluster(v, s, e, W, first ? mt.anom_idx : ~0u, first ? mt.limit_idx : ~0u, cand);
    }
    __syncthreads();
    TL_TICK(2);
    {
        // ---- the listed clusters, one wave each (eight waves per workgroup): big_replay on the tile's LDS arrays — the same
        //      entry format k_lz_emulate_giant feeds it (position, home slot relative to the cluster, word id)
        const uint32_t wv = (uint32_t)tid >> 6, ln = (uint32_t)tid & 63u, nwl = s_nwl;
        uint32_t *w_occ = s_wocc[wv];
        uint16_t *w_slot = s_wslot[wv], *w_rs = s_wrs[wv], *w_c = s_wc[wv];
        for (uint32_t q = wv; q < nwl; q += 8u) {
            const uint32_t cs = s_wl[q] & 0xFFFFu, cm = s_wl[q] >> 16;
            for (uint32_t i = ln; i < cm; i += 64u) w_rs[i] = (uint16_t)((s_rs[cs + i] & RS_MASK) - cs);
            __builtin_amdgcn_wave_barrier();
            const bool first = (a + cs) == 0;
            const uint32_t anom = first ? mt.anom_idx : ~0u, limit = first ? mt.limit_idx : ~0u;
            if (anom == ~0u && limit == ~0u) big_replay<LZ_WAVE_MIN, 1, true>(w_occ, w_slot, ln, W, cm, anom, limit, s_pos + cs, w_rs, s_pid + cs, w_c);
            else big_replay<LZ_WAVE_MIN, 1, false>(w_occ, w_slot, ln, W, cm, anom, limit, s_pos + cs, w_rs, s_pid + cs, w_c);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = ln; i < cm; i += 64u) { const uint32_t c = w_c[i]; if (c != LZ_NONE16) cand[s_pos[cs + i]] = (uint16_t)c; }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    TL_TICK(3);
    if (dbg && threadIdx.x == 0) { atomicAdd((unsigned long long *)&dbg[44], 1ull); atomicMax((unsigned long long *)&dbg[47], (unsigned long long)m); }
}

__global__ __launch_bounds__(512)
void k_lz_emulate(LzP P, LzScratch sc, uint32_t nb, const uint32_t *__restrict__ blist, const uint32_t *__restrict__ bcount, uint64_t *dbg)
{
    const uint32_t count = bcount ? *bcount : nb;
    for (uint32_t bi = blockIdx.y; bi < count; bi += gridDim.y) {
        lz_emulate_tile(P, sc, blist ? blist[bi] : bi, blockIdx.x, dbg);
        __syncthreads();
    }
}

// one workgroup per giant cluster (and per cluster of >= LZ_WAVE_MIN entries the tile kernel hands over).
// One word only: closed form.  <= LZ_GIANT_CAP entries: wave replay (lz_replay.h).  Larger: one lane, slots/occupants in
// global scratch, bitmap in LDS.
// CAPE = LZ_GIANT_CAP: every cluster left on the list (128 KiB of LDS: one workgroup per CU).  CAPE = LZ_GIANT_SMALL: only the
// clusters of at most that many entries — most of a list: the "runs" family lists ~2 000 clusters of 128 .. 7 500 entries per batch,
// and at one workgroup per CU, one wave of it replaying, they took 11 of that family's 23 ms — on 24 KiB, six workgroups per CU,
// launched first with a cursor of its own; a cluster it finishes is flagged like those of k_lz_emulate_dom and the full instance
// behind it skips it.
#define LZ_GIANT_SMALL 4096u
template <uint32_t CAPE>
__global__ __launch_bounds__(256)
void k_lz_emulate_giant(LzP P, LzScratch sc)
{
    constexpr bool SMALL = CAPE < LZ_GIANT_CAP;
    // one LDS buffer, two uses: wave replay of <= CAPE entries (slot -> word id | position << 16, entry -> slot) or — full instance
    // only — of up to 65 536 entries (slot -> word id only; lz_replay.h huge_replay)
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[SMALL ? 6 * CAPE : 2 * LZ_MAX_BLOCK];
    uint32_t *s_occ32 = reinterpret_cast<uint32_t *>(s_raw);                              // [CAPE]
    uint16_t *s_slot16 = reinterpret_cast<uint16_t *>(s_raw + 4 * CAPE);                  // [CAPE]
    uint16_t *s_oid16 = reinterpret_cast<uint16_t *>(s_raw);                              // [65536] (full instance)
    static_assert(6 * LZ_GIANT_CAP <= 2 * LZ_MAX_BLOCK, "wave-replay tables must fit the buffer");
    __shared__ uint32_t s_end;
    const int tid = threadIdx.x;
    const uint32_t n_back = sc.giant_count[2], count = sc.giant_count[0] + n_back;
    // clusters come off a cursor, the tile-spanning ones first (lz_giant_slot); every workgroup reaches the exit: the cursor only grows
    __shared__ uint32_t s_gnext;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_gnext = atomicAdd(&sc.giant_count[SMALL ? 4 : 3], 1u);
        __syncthreads();
        if (s_gnext >= count) break;
        const uint32_t g = lz_giant_slot(sc, s_gnext, n_back);
        const uint32_t lb = sc.giant_list[2 * g], a = sc.giant_list[2 * g + 1];
        if (a & DOM_DONE) continue;                      // k_lz_emulate_dom (or the small instance of this kernel) has replayed it
        const LzBlockMeta mt = sc.meta[lb];
        const uint32_t n = mt.n;
        const uint64_t *E = sc.eA + (size_t)lb * LZ_MAX_BLOCK;
        const uint32_t gid = (uint32_t)E[a] & 0xFFFFu;
        __syncthreads();
        if (tid == 0) s_end = n;
        __syncthreads();
        // (the small instance looks no further than one entry past its capacity: a longer cluster is not its business)
        const uint32_t scan_end = SMALL ? (n < a + CAPE + 2u ? n : a + CAPE + 2u) : n;
        for (uint32_t i = a + 1 + tid; i < scan_end; i += 256) if (((uint32_t)E[i] & 0xFFFFu) != gid) { atomicMin(&s_end, i); break; }
        __syncthreads();
        const uint32_t b = s_end, m = b - a;
        if (SMALL && (m > CAPE || (s_end == n && scan_end < n))) continue;
        const uint32_t W = 1u << P.wbits;
        uint16_t *cand = sc.cand + (size_t)lb * LZ_MAX_BLOCK;
        const bool first = a == 0;
        const uint32_t anom = first ? mt.anom_idx : ~0u, limit = first ? mt.limit_idx : ~0u;
        // A cluster of ONE word (a run of one byte value: zero pages, padding) needs no table: every entry has the same
        // home, find() only ever looks at that slot, and its occupant — the "anchor" — changes exactly when it is
        // retired: the step that retires it finds the slot empty and then takes it.  anchors: a(0) = first entry,
        // a(k+1) = first entry more than W positions after a(k); result = position of the current anchor, none at an
        // anchor.  (Not for the cluster that covers bucket 0 / T: spurious clear, non-wrapping find.)
        __shared__ uint32_t s_pure, s_nanch, s_anch[LZ_MAX_BLOCK / 256 + 4];
        if (tid == 0) s_pure = (anom == ~0u && limit == ~0u) ? 1u : 0u;
        __syncthreads();
        {
            const uint32_t pid0 = (uint32_t)(E[a] >> 48);
            bool same = true;
            for (uint32_t i = tid; i < m && same; i += 256) same = ((uint32_t)(E[a + i] >> 48) == pid0);
            if (!same) atomicAnd(&s_pure, 0u);
        }
        __syncthreads();
        if (s_pure) {
            if (tid == 0) {
                uint32_t k = 0, cur = 0;
                s_anch[0] = 0;
                for (;;) {
                    const uint32_t lim = ((uint32_t)(E[a + cur] >> 16) & 0xFFFFu) + W;      // retired by the first position beyond this
                    uint32_t lo = cur + 1, hi = m;                                          // first i in (cur, m) with pos_i > lim
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (((uint32_t)(E[a + mid] >> 16) & 0xFFFFu) > lim) hi = mid; else lo = mid + 1; }
                    if (lo >= m) break;
                    cur = lo; s_anch[++k] = cur;
                }
                s_nanch = k + 1;
            }
            __syncthreads();
            const uint32_t na = s_nanch;
            for (uint32_t i = tid; i < m; i += 256) {
                uint32_t lo = 0, hi = na - 1;                                               // last anchor <= i
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_anch[mid] <= i) lo = mid; else hi = mid - 1; }
                const uint32_t an = s_anch[lo];
                if (an != i) cand[(uint32_t)(E[a + i] >> 16) & 0xFFFFu] = (uint16_t)((uint32_t)(E[a + an] >> 16) & 0xFFFFu);
            }
            if (SMALL && tid == 0) sc.giant_list[2 * g + 1] = a | DOM_DONE;
            __syncthreads();
            continue;
        }
        {
            // One WAVE replays the cluster (lz_replay.h: bitmap in registers, the probe walk 64 buckets per step) — a
            // cluster this large is a long run of one word with foreign words inside its bucket range, and every find() of
            // a foreign word walks the run: lane-serial, that was seconds per block.  The entries are first laid out as
            // four u16 arrays in this block's eB (free after the sort; clusters use disjoint ranges [a, a + m)).
            uint16_t *g16 = reinterpret_cast<uint16_t *>(sc.eB + (size_t)lb * LZ_MAX_BLOCK);
            uint16_t *gp = g16 + a, *grs = g16 + LZ_MAX_BLOCK + a, *gid = g16 + 2 * LZ_MAX_BLOCK + a, *gc = g16 + 3 * LZ_MAX_BLOCK + a;
            for (uint32_t i = tid; i < m; i += 256) {
                const uint64_t e = E[a + i];
                gp[i] = (uint16_t)(e >> 16);
                grs[i] = (uint16_t)(((uint32_t)(e >> 32) & 0xFFFFu) - a);
                gid[i] = (uint16_t)(e >> 48);
            }
            __threadfence();
            __syncthreads();
            if (tid < 64) {
                const bool plain = anom == ~0u && limit == ~0u;
                if constexpr (SMALL) {
                    if (m <= 2048u) {
                        if (plain) big_replay<CAPE, 1, true>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                        else big_replay<CAPE, 1, false>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                    } else {
                        if (plain) big_replay<CAPE, 2, true>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                        else big_replay<CAPE, 2, false>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                    }
                } else if (m <= LZ_GIANT_CAP) {
                    // as few bitmap registers as the cluster needs (2048 slots each): every first-fit and every clear walks them all
                    if (m <= 4096u) {
                        if (plain) big_replay<LZ_GIANT_CAP, 2, true>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                        else big_replay<LZ_GIANT_CAP, 2, false>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                    } else if (m <= 8192u) {
                        if (plain) big_replay<LZ_GIANT_CAP, 4, true>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                        else big_replay<LZ_GIANT_CAP, 4, false>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                    } else {
                        if (plain) big_replay<LZ_GIANT_CAP, (LZ_GIANT_CAP + 2047) / 2048, true>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                        else big_replay<LZ_GIANT_CAP, (LZ_GIANT_CAP + 2047) / 2048, false>(s_occ32, s_slot16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc);
                    }
                } else {
                    // posA / posB are free after k_lz_sort_home; every cluster uses its own range [a, a + m)
                    uint16_t *gslot = sc.posA + (size_t)lb * LZ_MAX_BLOCK + a, *gopos = sc.posB + (size_t)lb * LZ_MAX_BLOCK + a;
                    if (plain) huge_replay<true>(s_oid16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc, gslot, gopos);
                    else huge_replay<false>(s_oid16, (uint32_t)tid, W, m, anom, limit, gp, grs, gid, gc, gslot, gopos);
                }
                __threadfence();
            }
            __syncthreads();
            for (uint32_t i = tid; i < m; i += 256) { const uint32_t c = gc[i]; if (c != LZ_NONE16) cand[gp[i]] = (uint16_t)c; }
            if (SMALL && tid == 0) sc.giant_list[2 * g + 1] = a | DOM_DONE;
        }
        __syncthreads();
    }
}

// =============================================================================================
// k_lz_emulate_dom — giant clusters that ONE word dominates (the "pages" cliff)
// =============================================================================================
// A block of binary data — zero pages, padding, a repeated record — puts 13 000 .. 60 000 copies of one 4-byte word X
// into one cluster, with a few hundred entries of other words whose homes fall inside X's run of buckets (measured on
// the "pages" family: 97.5 % of the largest cluster are X).  The general wave replay (lz_replay.h) needs per slot the
// occupant's word id and position and per entry its slot: 6 bytes x the cluster, LDS for 18 432 entries, HBM round
// trips on the serial chain beyond that (huge_replay: ~3 us per entry, 0.14 GB/s on the family).
//
// With a dominant word nearly all of that state is redundant.  find(X) only ever looks at X's home slot rX, whose
// occupant — while it is a copy of X — is "the anchor": one register.  A slot that holds some other copy of X needs no
// record at all: a foreign word's find() walks past it (different word), and nobody asks for its position.  What is
// left:  one occupancy BIT per slot (8 KiB),  the slot of every LIVE entry for its retirement (a ring of W u16: entries
// are retired in insertion order, at most W are alive),  and {word id, position, slot} of the live FOREIGN entries (a
// FIFO of a few hundred).  78 KiB for any cluster size: two workgroups per CU, nothing on the serial chain but LDS.
//   find(Y), Y foreign, home r: nothing if slot r is free; else the occupied run [r, e) ends at the first free slot e
//            (one wave-wide bitmap scan) and the answer is the live copy of Y with the smallest slot inside it (one
//            wave-wide pass over the foreign FIFO).
//   insert: first free slot from the home (the same scan; for X it starts at a hint: every slot between rX and the
//            hint word is known to be full, retirements pull the hint back).
// Two things this representation cannot express end the attempt: a foreign entry landing ON rX (find(X) would have to
// walk to the next copy of X and return its position), and more live foreign entries than the FIFO holds.  The cluster is
// then left — its partial results wiped — to k_lz_emulate_giant, as is every cluster that is not dominated or that covers
// bucket 0 / T (spurious clear, non-wrapping find).  Exactness: tests/test_fuzz_gpu.py ("pages", "runs" at every position
// against the oracle's literal table), tests/test_lz_find_gpu.py::test_dominated_giant_clusters.
#include "lz_dom.h"

__global__ __launch_bounds__(256)
void k_lz_emulate_dom(LzP P, LzScratch sc, uint64_t *dbg)
{
    __shared__ uint32_t s_occ[LZ_MAX_BLOCK / 32 + 72];
    __shared__ uint16_t s_ring[DOM_RING];
    __shared__ uint16_t s_fid[DOM_FCAP], s_fpos[DOM_FCAP], s_fslot[DOM_FCAP];
    __shared__ uint32_t s_end, s_votes[3], s_result;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = 1u << P.wbits;
    if (W > DOM_RING) return;
    const uint32_t n_back = sc.giant_count[2], count = sc.giant_count[0] + n_back;
    const long long tk_wg = dbg ? clock64() : 0;
    // clusters are handed out by a cursor (they differ in size by a factor of 70: a static stride left the longest workgroup
    // 3.4 x the average), the tile-spanning ones first; every workgroup reaches the exit: the cursor only grows
    __shared__ uint32_t s_next;
    struct Src {                                        // the fallback pipeline's entry records; results by position
        const uint64_t *E; uint32_t a; uint16_t *cand;
        __device__ __forceinline__ uint64_t ent(uint32_t i) const { const uint64_t e = E[a + i]; return e - ((uint64_t)a << 32); }   // (the slot field is >= a)
        __device__ __forceinline__ void put(uint32_t, uint32_t pos, uint32_t res) const { cand[pos] = (uint16_t)res; }
    };
    for (;;) {
        __syncthreads();
        if (tid == 0) s_next = atomicAdd(&sc.giant_count[1], 1u);
        __syncthreads();
        if (s_next >= count) break;
        const uint32_t g = lz_giant_slot(sc, s_next, n_back);
        const uint32_t lb = sc.giant_list[2 * g], a = sc.giant_list[2 * g + 1];
        const LzBlockMeta mt = sc.meta[lb];
        const uint32_t n = mt.n;
        const uint64_t *E = sc.eA + (size_t)lb * LZ_MAX_BLOCK;
        if (a == 0 && (mt.anom_idx != ~0u || mt.limit_idx != ~0u)) continue;      // the cluster that covers bucket 0 / T
        const uint32_t gid = (uint32_t)E[a] & 0xFFFFu;
        __syncthreads();
        if (tid == 0) { s_end = n; s_votes[0] = s_votes[1] = s_votes[2] = 0; s_result = 0; }
        __syncthreads();
        for (uint32_t i = a + 1 + tid; i < n; i += 256) if (((uint32_t)E[i] & 0xFFFFu) != gid) { atomicMin(&s_end, i); break; }
        __syncthreads();
        const uint32_t m = s_end - a;
        Src src{E, a, sc.cand + (size_t)lb * LZ_MAX_BLOCK};
        const bool done = dom_cluster(src, m, W, W - 1u, s_occ, s_ring, s_fid, s_fpos, s_fslot, s_votes, s_result, dbg);
        if (done && tid == 0) sc.giant_list[2 * g + 1] = a | DOM_DONE;
        if (dbg && tid == 0) atomicMax((unsigned long long *)&dbg[58], (unsigned long long)m);
    }
    if (dbg && tid == 0) { atomicMax((unsigned long long *)&dbg[59], (unsigned long long)(clock64() - tk_wg)); atomicMax((unsigned long long *)&dbg[60], (unsigned long long)count); }
}

// =============================================================================================
// host side of the match finder
// =============================================================================================
size_t lz2_scratch_bytes(uint32_t nb);
size_t lz_scratch_bytes(uint32_t nb)
{
    size_t per = (size_t)LZ_MAX_BLOCK * (2 + 2 + 8 + 8 + 2) + sizeof(LzBlockMeta) + LZ_MAX_GIANTS_PER_BLOCK * 8 +
                 (size_t)LZ_SLOT_WORDS * 4 + 8;
    return per * nb + 16 * 256 + 4096 + lz2_scratch_bytes(nb) + 64 * 256;
}

mi_status lz_find_batch(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                        const LzScratch &sc, hipStream_t s, const uint32_t *blist, const uint32_t *bcount);
size_t lz2_scratch_bytes(uint32_t nb);
void   lz2_carve(mi_carver &cv, uint32_t nb, Lz2Scratch *sc);
mi_status lz2_stage_partition(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                              const Lz2Scratch &sc, hipStream_t s);
mi_status lz2_stage_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                         const Lz2Scratch &sc, hipStream_t s);
mi_status lz2_stage_find_wide(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                              const Lz2Scratch &sc, hipStream_t s, bool aside);
mi_status lz2_stage_b(mi_ctx *ctx, const LzP &P, uint32_t nb, const Lz2Scratch &sc, hipStream_t s, int which);
void   lz2_launch_scatter(const Lz2Scratch &sc, uint16_t *cand_by_pos, uint32_t nb, hipStream_t s);

bool lz_use_v2()
{
    const char *e = getenv("MI_LZ_V2");
    return !(e && e[0] == '0');
}

void lz_carve(mi_ctx *ctx, uint32_t nb, LzScratch *sc, Lz2Scratch *sc2, int set)
{
    mi_carver cv((uint8_t *)ctx->ws + (size_t)set * mi_align_up(lz_scratch_bytes(nb), 4096));
    sc->posA = cv.take<uint16_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->posB = cv.take<uint16_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->eA = cv.take<uint64_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->eB = cv.take<uint64_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->cand = cv.take<uint16_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->meta = cv.take<LzBlockMeta>(nb);
    sc->giant_count = cv.take<uint32_t>(64);
    sc->giant_list = cv.take<uint32_t>((size_t)nb * LZ_MAX_GIANTS_PER_BLOCK * 2);
    sc->giant_cap = nb * LZ_MAX_GIANTS_PER_BLOCK;
    sc->slot = cv.take<uint32_t>((size_t)nb * LZ_SLOT_WORDS);
    sc->block_bits = cv.take<uint64_t>(nb + 1);
    if (sc2) lz2_carve(cv, nb, sc2);
}

// match finder for blocks [block0, block0+nb), in two stages so that a caller can overlap them across
// batches: A = partition + find (+ the first pipeline for the blocks the LDS-resident path hands back, or
// for everything when MI_LZ_V2=0); B = replay of the exported clusters.
// `sf` = stream of the fallback chain (may equal `s`); when it differs the caller joins it before the parse:
// the chain is normally empty, but its launches ask for 82..155 KiB of LDS per workgroup and would otherwise
// sit in front of the real work waiting for that LDS.
mi_status lz_find_stage_a(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                          const LzScratch &sc, const Lz2Scratch &sc2, hipStream_t s, hipStream_t sf, hipEvent_t ev_part, hipEvent_t ev_fb,
                          hipEvent_t ev_wide)
{
    if (!lz_use_v2()) return lz_find_batch(ctx, P, d_in, n, block0, nb, sc, s, nullptr, nullptr);
    mi_status st = lz2_stage_partition(ctx, P, d_in, n, block0, nb, sc2, s);
    if (st) return st;
    // MI_LZ_UNSAFE_NO_FALLBACK=1 (measurement only: wrong output for any block the partition hands back) leaves the
    // fallback chain out, to price its normally empty launches
    static const bool no_fb = getenv("MI_LZ_UNSAFE_NO_FALLBACK") != nullptr;
    static const bool wide_inline = getenv("MI_LZ_WIDE_INLINE") != nullptr;      // A/B: the wide finder on the main stream as in round 3
    if (sf != s) { MI_HIP(ctx, hipEventRecord(ev_part, s)); MI_HIP(ctx, hipStreamWaitEvent(sf, ev_part, 0)); }
    const bool wide_aside = sf != s && ev_wide && !wide_inline;
    if (wide_aside) {
        // first thing on the side chain: stage B waits for it (exported clusters of wide parts), the fallback chain behind it does not matter
        st = lz2_stage_find_wide(ctx, P, d_in, n, block0, nb, sc2, sf, true);
        if (st) return st;
        MI_HIP(ctx, hipEventRecord(ev_wide, sf));
    }
    if (!no_fb) {
        st = lz_find_batch(ctx, P, d_in, n, block0, nb, sc, sf, sc2.fallback_list, sc2.fallback_count);
        if (st) return st;
    }
    if (sf != s) MI_HIP(ctx, hipEventRecord(ev_fb, sf));
    st = lz2_stage_find(ctx, P, d_in, n, block0, nb, sc2, s);
    if (st || wide_aside) return st;
    return lz2_stage_find_wide(ctx, P, d_in, n, block0, nb, sc2, s, false);
}
mi_status lz_find_stage_b(mi_ctx *ctx, const LzP &P, uint32_t nb, const Lz2Scratch &sc2, hipStream_t s, int which)
{
    return lz_use_v2() ? lz2_stage_b(ctx, P, nb, sc2, s, which) : MI_OK;
}
mi_status lz_run_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                      const LzScratch &sc, const Lz2Scratch &sc2, hipStream_t s)
{
    mi_status st = lz_find_stage_a(ctx, P, d_in, n, block0, nb, sc, sc2, s, s, nullptr, nullptr, nullptr);
    return st ? st : lz_find_stage_b(ctx, P, nb, sc2, s, 7);
}
mi_status lz_check_params(const mi_lz_params *p)
{
    if (!p) return MI_ERR_ARG;
    if (p->wbits < 8 || p->wbits > 16) return MI_ERR_ARG;
    if (p->lbits < 3 || p->lbits > 8) return MI_ERR_ARG;
    if (p->tbits < 17 || p->tbits > 24) return MI_ERR_ARG;
    // blocks above 64 KiB: the lz77 flavour only (lzw.hip), multiples of 256 bytes up to 1 MiB, so that WINDOW_BITS 16 slides
    if (p->block < 1) return MI_ERR_ARG;
    if (p->block > LZ_MAX_BLOCK && (p->deflate || p->block > (1u << 20) || (p->block & 255u) || p->lbits > 5)) return MI_ERR_ARG;
    if (1u + p->wbits + p->lbits > 32) return MI_ERR_ARG;
    return MI_OK;
}

mi_status lz_find_batch(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                        const LzScratch &sc, hipStream_t s, const uint32_t *blist, const uint32_t *bcount);
mi_status lz_find_batch(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                        const LzScratch &sc, hipStream_t s, const uint32_t *blist, const uint32_t *bcount)
{
    // as the fallback of the LDS-resident finder these launches are normally empty and merely wait for LDS behind
    // k_lz2_find: timing them would report that wait as kernel time
    const int saved_prof = ctx->profiling;
    static const bool prof_fb = getenv("MI_LZ_PROF_FALLBACK") != nullptr;      // inputs that live in the fallback (scripts/adv_profile.py)
    if (blist && !prof_fb) ctx->profiling = 0;
    MI_HIP(ctx, hipMemsetAsync(sc.giant_count, 0, 32, s));       // [0] / [2] clusters listed from the front / the end, [1] / [3] / [4] cursors (lz_common.h)
    // fallback: few looping workgroups (see LZ_FB_GRID) — unless the last finished batch (in practice: of an earlier call) had
    // many blocks here (non-text input): the count k_lz_sort_home left in pinned memory sizes the grids (pages family: half the
    // chip sat idle behind 128 workgroups)
    const uint32_t hint = (blist && ctx->h_order) ? __atomic_load_n(ctx->h_order + 1, __ATOMIC_RELAXED) : 0u;
    const uint32_t fb_want = hint > LZ_FB_GRID ? hint : LZ_FB_GRID;
    const uint32_t fgrid = (blist && nb > fb_want) ? fb_want : nb;
    {
        mi_prof_scope p(ctx, "k_lz_sort_home", s, (uint64_t)nb * P.block);
        hipLaunchKernelGGL(k_lz_sort_home, dim3(fgrid), dim3(1024), 0, s, d_in, n, P, sc, block0, nb, blist, bcount);
    }
    {
        mi_prof_scope p(ctx, "k_lz_sort_cluster", s, (uint64_t)nb * P.block);
        hipLaunchKernelGGL(k_lz_sort_cluster, dim3(fgrid), dim3(1024), 0, s, sc, nb, blist, bcount);
    }
    {
        mi_prof_scope p(ctx, "k_lz_emulate", s, (uint64_t)nb * P.block);
        const uint32_t tiles = (P.block + LZ_TILE_NOM - 1) / LZ_TILE_NOM;
        hipLaunchKernelGGL(k_lz_emulate, dim3(tiles, fgrid), dim3(512), 0, s, P, sc, nb, blist, bcount, ctx->lz_dbg);
    }
    {
        // dominated giant clusters first (78 KiB of LDS: two workgroups per CU); what it leaves goes to the general kernel
        mi_prof_scope p(ctx, "k_lz_emulate_dom", s, (uint64_t)nb * P.block);
        const uint32_t grid = nb < 512 ? nb : 512;          // two per CU; a workgroup that finds the list empty leaves at once
        hipLaunchKernelGGL(k_lz_emulate_dom, dim3(grid), dim3(256), 0, s, P, sc, ctx->lz_dbg);
    }
    {
        mi_prof_scope p(ctx, "k_lz_emulate_giant", s, (uint64_t)nb * P.block);
        const uint32_t cap = blist ? (hint > LZ_FB_GRID ? 512u : LZ_FB_GRID) : 1024u;
        const uint32_t grid = nb < cap ? nb : cap;
        if (blist && hint > 8u) {                           // blocks do fall back (lz_emit.hip's fb_busy): the clusters of <= LZ_GIANT_SMALL entries six workgroups per CU
            const uint32_t gs = nb * 2u < (uint32_t)ctx->num_cu * 6u ? nb * 2u : (uint32_t)ctx->num_cu * 6u;
            hipLaunchKernelGGL(k_lz_emulate_giant<LZ_GIANT_SMALL>, dim3(gs), dim3(256), 0, s, P, sc);
        }
        hipLaunchKernelGGL(k_lz_emulate_giant<LZ_GIANT_CAP>, dim3(grid), dim3(256), 0, s, P, sc);
    }
    ctx->profiling = saved_prof;
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

uint32_t lz_batch_blocks(mi_ctx *ctx, uint64_t nblocks)
{
    // As few, as large and as EQUAL batches as fit 3840 blocks (round 4).  The three-stream overlap buys 8 % over running every kernel
    // alone, while every batch pays every kernel's tail: 10^8 B (1 526 blocks) in one batch 17.9 GB/s against 16.5 in two (1 024 + 502),
    // 125 MB 18.4 / 17.6, 2 x 10^8 B 19.9 / 18.6, 3 x 10^8 B 20.5 (2 x 2 289) / 18.8 (1 024 x 4 + 482), 5 x 10^8 B 21.5 (2 x 3 815) / 20.8
    // (2 048 x 3 + 1 486), 10^9 B 21.8-21.9 (4 x 3 815) / 21.7 (3 072 x 4 + 2 971) — same box, mode H.  (Until then: 1 024 below 6 144
    // blocks, 2 048 below 9 216, else 3 072 — "as long as three stages have three batches to overlap".)  ~1.6 GB of workspace per 1 024
    // blocks and set, three sets: 288 GB of HBM make that free; k_lz_scan_blocks holds a batch to 4 096 blocks.
    const uint64_t nbat = (nblocks + 3839u) / 3840u;
    uint64_t cap = nbat ? (nblocks + nbat - 1u) / nbat : 1u;
    // ... unless the input lives in the fallback pipeline (the hint an earlier call left, lz_emit.hip: at least a sixteenth of a batch's
    // blocks): its chains are long serial kernels on few workgroups, and what helps THEM is two batches' chains side by side on two
    // streams — "runs" 4.7 GB/s in two batches against 3.8 in one, "pages" 3.5 / 3.4 (10^8 B): the old rule stays for such input
    if (ctx && ctx->h_order) {
        const uint32_t fb = __atomic_load_n(ctx->h_order + 1, __ATOMIC_RELAXED), of = __atomic_load_n(ctx->h_order + 2, __ATOMIC_RELAXED);
        if (fb > 8u && (uint64_t)fb * 16u >= of) cap = nblocks >= 3u * 3072u ? 3072u : nblocks >= 3u * 2048u ? 2048u : 1024u;
    }
    if (const char *e = getenv("MI_LZ_BATCH")) { long v = atol(e); if (v >= 1 && v <= 4096) cap = (uint64_t)v; }
    return (uint32_t)(nblocks < cap ? (nblocks ? nblocks : 1) : cap);
}

extern "C" mi_status mi_lz_find_all_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                                        uint16_t *d_cand, void *stream)
{
    if (!ctx || !d_in || !d_cand) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (n == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    const LzP P = lz_params_of(ctx, p);
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    const uint32_t nbmax = lz_batch_blocks(ctx, nblocks);
    st = mi_ws_reserve(ctx, lz_scratch_bytes(nbmax));
    if (st) return st;
    LzScratch sc; Lz2Scratch sc2;
    lz_carve(ctx, nbmax, &sc, &sc2, 0);
    for (uint64_t b0 = 0; b0 < nblocks; b0 += nbmax) {
        const uint32_t nb = (uint32_t)((nblocks - b0) < nbmax ? (nblocks - b0) : nbmax);
        st = lz_run_find(ctx, P, d_in, n, b0, nb, sc, sc2, s);
        if (st) return st;
        if (lz_use_v2()) lz2_launch_scatter(sc2, sc.cand, nb, s);
        // cand rows are LZ_MAX_BLOCK apart in scratch; the caller's array is block-size apart
        for (uint32_t i = 0; i < nb; ++i) {
            const uint64_t off = (b0 + i) * (uint64_t)P.block;
            const uint64_t len = (n - off) < P.block ? (n - off) : P.block;
            MI_HIP(ctx, hipMemcpyAsync(d_cand + off, sc.cand + (size_t)i * LZ_MAX_BLOCK, len * 2, hipMemcpyDeviceToDevice, s));
        }
    }
    return MI_OK;
}


// lzw.hip
size_t    lzw_scratch_bytes(uint32_t nb, uint32_t block);
void      lzw_carve(mi_ctx *ctx, uint32_t nb, uint32_t block, LzwScratch *sc);
uint32_t  lzw_batch_blocks(mi_ctx *ctx, uint64_t nblocks, uint32_t block);
mi_status lzw_or_lzs_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb, const LzwScratch &sc, hipStream_t s);

// the same hook for blocks above 64 KiB (lz77 flavour, lzw.hip): 32-bit positions, 0xFFFFFFFF = none
extern "C" mi_status mi_lz_find_all32_dev(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_in, uint64_t n,
                                          uint32_t *d_cand, void *stream)
{
    if (!ctx || !d_in || !d_cand) return MI_ERR_ARG;
    mi_status st = lz_check_params(p);
    if (st) return st;
    if (p->block <= LZ_MAX_BLOCK) return MI_ERR_ARG;
    if (n == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    const LzP P = lz_params_of(ctx, p);
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    uint32_t nbw = lzw_batch_blocks(ctx, nblocks, P.block);
    while ((st = mi_ws_reserve(ctx, lzw_scratch_bytes(nbw, P.block) + 4096)) == MI_ERR_NOMEM && nbw > 1) nbw = (nbw + 1) / 2;
    if (st) return st;
    LzwScratch ws;
    lzw_carve(ctx, nbw, P.block, &ws);
    for (uint64_t b0 = 0; b0 < nblocks; b0 += nbw) {
        const uint32_t nb = (uint32_t)((nblocks - b0) < nbw ? (nblocks - b0) : nbw);
        st = lzw_or_lzs_find(ctx, P, d_in, n, b0, nb, ws, s);
        if (st) return st;
        for (uint32_t i = 0; i < nb; ++i) {
            const uint64_t off = (b0 + i) * (uint64_t)P.block;
            const uint64_t len = (n - off) < P.block ? (n - off) : P.block;
            MI_HIP(ctx, hipMemcpyAsync(d_cand + off, ws.cand + (size_t)i * ws.S, len * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    return MI_OK;
}
