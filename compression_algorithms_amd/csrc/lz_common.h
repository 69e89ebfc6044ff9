// lz_common.h — shared pieces of the block-parallel LZ77 pipeline (lz_find.hip, lz_emit.hip).
//
// The reference's match finder (algorithms/lz77/lz77.c:55-108, algorithms/deflate/lz77.c:77-174)
// is one candidate lookup in a 2^20..2^22 bucket linear-probing table with FIFO eviction and
// no tombstones.  Because every position is inserted exactly once, in order, whatever the
// parse does, find() at position p is a pure function of the block prefix, and the table
// never has to exist: positions are sorted by home bucket, split into probe clusters that
// cannot interact ("parking" bound), and each cluster is replayed in time order against a
// dense occupancy bitmap held in LDS.  DESIGN.md has the argument and the measurements.
#pragma once
#include "common.h"

#define LZ_MAX_BLOCK   65536u
#define LZ_TAIL        64u            // zero bytes the reference would read past the block end
#define LZ_NONE16      0xFFFFu

// emulation tiles
#define LZ_TILE_NOM    4096u          // nominal entries per tile (tile t = clusters that start in [t*NOM,(t+1)*NOM))
#define LZ_TILE_CAP    8192u          // LDS capacity of a normal tile
#define LZ_GIANT_MIN   (LZ_TILE_CAP - LZ_TILE_NOM)   // a cluster larger than this is replayed by k_lz_emulate_giant
#define LZ_GIANT_CAP   18432u         // LDS capacity of the giant kernel; above: global-memory path
#define LZ_MAX_TILES   (LZ_MAX_BLOCK / LZ_TILE_NOM)
#define LZ_WAVE_MIN    128u           // fallback pipeline: clusters from this size on are replayed by a wave, not a lane
#define LZ_MAX_GIANTS_PER_BLOCK (LZ_MAX_BLOCK / LZ_WAVE_MIN + 32u)
#ifndef LZ_TILE_WAVE_MIN
#define LZ_TILE_WAVE_MIN 16u          // ... and mixed clusters from this size on by a wave of the tile's own workgroup (lz_find.hip)
#endif

struct LzP {
    uint32_t wbits, lbits, tbits, deflate, block;
    uint32_t flags;        // LZP_ARANK: the context's self-check found LDS returning atomics lane-ordered (ctx.hip)
    uint32_t *order_flag;  // host-visible word of the context: a consumer that finds a sort out of order sets it (below)
    uint64_t *stats;       // device counters of the context (mi_lz_path_stats): [0] blocks sent to the fallback pipeline, [1] wide parts
};
#define LZP_ARANK 1u
#define LZP_BREAK 2u       // TEST BUILD ONLY (-DMI_TEST_HOOKS, MI_LZ_TEST_BREAK_RANK=1): the scatter swaps the ranks of neighbouring lanes with equal digits
#define LZP_FORCE_FB 4u    // TEST BUILD ONLY (-DMI_TEST_HOOKS, MI_LZ_TEST_FORCE_FALLBACK=1): the partition hands EVERY block to the fallback pipeline

// The stable radix scatter under LZP_ARANK ranks by the order in which ONE returning LDS add serves the lanes that hit one
// address — lane order on gfx950, measured (scripts/micro/lds_atomic_order.hip) and probed once per context (ctx.hip), but
// not an ISA promise.  An unstable sort would still round-trip (find() would pick a later copy of a word: a valid,
// different stream), so stability is CHECKED where every sort is consumed, for a few compares per entry: the final order
// must be ascending in (key, time).  A violation sets this word in pinned host memory; the context then ranks with
// ballots from its next call on, mi_sync() / mi_order_violations() report it (MI_ERR_UNSTABLE), and the host-buffer
// entry points — which wait for their result anyway — encode again with ballots before they return.
__device__ __forceinline__ void lz_order_violation(const LzP &P)
{
    if (P.order_flag) __hip_atomic_fetch_or(P.order_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// kernel parameters of one call: the caller's mi_lz_params + what the context knows (ranking mode, order-violation word)
static inline LzP lz_params_of(mi_ctx *ctx, const mi_lz_params *p)
{
    mi_order_poll(ctx);                                             // a violation reported by an earlier call switches to ballots
    uint32_t fl = ctx->lds_rank_ok ? LZP_ARANK : 0u;
    if (fl && ctx->test_break_rank) fl |= LZP_BREAK;
    if (ctx->test_force_fb) fl |= LZP_FORCE_FB;
    return LzP{p->wbits, p->lbits, p->tbits, p->deflate, p->block, fl, ctx->d_order, ctx->d_stats};
}

// per-block record written by k_lz_sort_home
struct LzBlockMeta {
    uint32_t n;            // bytes in this block
    uint32_t ngroups;
    uint32_t anom_idx;     // dense index of bucket 0 inside cluster 0 (spurious one-time clear), or ~0u
    uint32_t limit_idx;    // dense index of bucket T inside cluster 0 (deflate find() must stop there), or ~0u
    uint32_t rot;          // rotation applied to the home order (deflate insert wraps modulo T)
    uint32_t pad[3];
};

// entry record, 64 bits: gid | pos << 16 | rloc << 32 | pid << 48
//   gid   cluster number in home order        pos   position in the block (= time)
//   rloc  dense bucket index of the home      pid   position of the first occurrence of the same word
__device__ __forceinline__ uint64_t lz_pack(uint32_t gid, uint32_t pos, uint32_t rloc, uint32_t pid)
{
    return (uint64_t)gid | ((uint64_t)pos << 16) | ((uint64_t)rloc << 32) | ((uint64_t)pid << 48);
}

// per-batch scratch (device pointers into the context workspace); all per-block strides are fixed
struct LzScratch {
    uint16_t    *posA, *posB;      // [nb][65536]  sort by home, ping-pong
    uint64_t    *eA, *eB;          // [nb][65536]  entry records, ping-pong of the sort by cluster
    uint16_t    *cand;             // [nb][65536]  find() result per position
    LzBlockMeta *meta;             // [nb]
    uint32_t    *giant_count;      // [0] clusters listed from the front, [1] cursor of k_lz_emulate_dom, [2] clusters listed from the END
                                   // (the ones a tile could not hold: the largest — handed out first), [3] cursor of k_lz_emulate_giant
    uint32_t    *giant_list;       // [giant_cap][2] = {block, start index}
    uint32_t     giant_cap;        // nb * LZ_MAX_GIANTS_PER_BLOCK
    uint32_t    *slot;             // [nb][SLOT_WORDS] block-local token stream
    uint64_t    *block_bits;       // [nb] bits produced per block (this batch)
};
#define LZ_SLOT_WORDS  (LZ_MAX_BLOCK / 2 + 16)     // 2 bytes per input byte worst case (+ slack), in u32
#define LZ_DEFH_HIST_AT 32000u                    // mode H: k_lz_parse_emit leaves the block's 286-bin tally here (a record is < 18 508 words)

// scratch of the HBM-resident finder for blocks above 64 KiB (lzw.hip)
struct LzwScratch {
    uint64_t *eA, *eB;          // [nb][S] sort ping-pong: key << 32 | position
    uint32_t *gid;              // [nb][S] cluster number by position
    uint32_t *rd;               // [nb][S] dense home slot by position
    uint32_t *cstart;           // [nb][S + 2] first (cluster, time)-sorted index of every cluster; [ncl] = n
    uint32_t *ncl;              // [nb][4] clusters, flag: bucket 0 is inside cluster 0
    uint8_t  *t_live;           // [nb][S + 64] literal table, by dense bucket
    uint32_t *t_pos, *t_mix;    // [nb][S]
    uint32_t *slot_of;          // [nb][S] bucket of the k-th (cluster, time)-sorted entry (for its retirement)
    uint32_t *cand;             // [nb][S] find() by position (LZW_NONE = none)
    uint32_t *ent;              // [nb][S] per (cluster, time)-sorted entry: home slot relative to the cluster | word id << 16
    uint32_t *relw;             // [nb][S] by position: the same pair, written by the sweep
    uint16_t *cand_e;           // [nb][S] per sorted entry: index (inside its cluster) of the entry find() returned, 0xFFFF = none
    uint64_t *clist[6];         // per size class: block << 32 | cluster number (classes: <= 1024, <= 4096, <= 8192, <= 24576 entries, larger; 5: 2..7 entries)
    uint32_t *ccount;           // [6] entries of the class lists (zeroed per batch)
    uint32_t *slot;             // [nb][slot_words] block-local token stream
    uint64_t *block_bits;       // [nb + 1]
    uint32_t  S, slot_words;
};

// The g-th cluster of the fallback's work list, longest processing time first: the clusters that span tiles (listed from the
// end of the array) come before the ones found inside a tile.  A cursor that hands out a 35 000-entry cluster last leaves one
// workgroup replaying it alone: measured on the "pages" family, the longest workgroup of k_lz_emulate_dom ran 3.3 x the average.
__device__ __forceinline__ uint32_t lz_giant_slot(const LzScratch &sc, uint32_t g, uint32_t n_back)
{
    return g < n_back ? sc.giant_cap - 1u - g : g - n_back;
}

// reference hash(): algorithms/lz77/lz77.c:13-41 == algorithms/deflate/lz77.c:14-42
__device__ __forceinline__ uint32_t lz_mix32(uint32_t w)
{
    uint32_t k = w * 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    uint32_t h = (k << 13) | (k >> 19);
    h = h * 5u + 0xe6546b64u;
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13; h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// little-endian 32-bit word at an arbitrary byte offset of an LDS byte array (4-byte aligned base,
// >= 8 readable bytes after the last valid offset)
__device__ __forceinline__ uint32_t lds_word(const uint8_t *s, uint32_t p)
{
    const uint32_t *a = reinterpret_cast<const uint32_t *>(s + (p & ~3u));
    const uint64_t v = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
    return (uint32_t)(v >> ((p & 3u) * 8u));
}

// A block's bytes -> LDS with the zero tail the reference reads past `size`, by a workgroup of 1024 threads:
// s_dst holds LZ_MAX_BLOCK + LZ_TAIL bytes (+ slack), 16-byte aligned.  The 16-byte loads of a thread (four or five) are
// issued together and UNCONDITIONALLY (offsets clamped to the last whole vector) before the first LDS store: written as
// "if (in range) s_dst[i] = src[i]" per iteration every load waited for its value in its own basic block — five HBM round
// trips one after the other at the head of three kernels (round 4).
__device__ __forceinline__ void lz_block_to_lds(uint8_t *s_dst, const uint8_t *__restrict__ src, uint32_t n, uint32_t tid)
{
    constexpr uint32_t NIT = (LZ_MAX_BLOCK + LZ_TAIL + 1024u * 16u - 1u) / (1024u * 16u);
    const bool vec_ok = ((((uintptr_t)src) & 15u) == 0) && n >= 16u;
    if (vec_ok) {
        const uint32_t last = (n - 16u) & ~15u;                   // offset of the last whole 16-byte piece
        uint4 v[NIT];
#pragma unroll
        for (uint32_t k = 0; k < NIT; ++k) {
            const uint32_t i = tid * 16u + k * 1024u * 16u;
            v[k] = *reinterpret_cast<const uint4 *>(src + (i <= last ? i : last));
        }
#pragma unroll
        for (uint32_t k = 0; k < NIT; ++k) {
            const uint32_t i = tid * 16u + k * 1024u * 16u;
            if (i >= LZ_MAX_BLOCK + LZ_TAIL) continue;
            if (i + 16u <= n) *reinterpret_cast<uint4 *>(s_dst + i) = v[k];
            else if (i >= n) *reinterpret_cast<uint4 *>(s_dst + i) = make_uint4(0u, 0u, 0u, 0u);
            else {                                              // the one ragged piece of a block (rolled: unrolled byte loads of five
#pragma unroll 1                                                 // pieces cost k_lz_sort_home 244 bytes of scratch per lane)
                for (uint32_t b = 0; b < 16; ++b) s_dst[i + b] = (i + b < n) ? src[i + b] : (uint8_t)0;
            }
        }
    } else {
#pragma unroll 1
        for (uint32_t i = tid * 16u; i < LZ_MAX_BLOCK + LZ_TAIL; i += 1024u * 16u) {
#pragma unroll 1
            for (uint32_t b = 0; b < 16; ++b) s_dst[i + b] = (i + b < n) ? src[i + b] : (uint8_t)0;
        }
    }
}

// ---- block-wide scans over 1024 threads (16 waves) -------------------------------------------
template <typename T, typename Op>
__device__ __forceinline__ T wave_inclusive_scan(T v, Op op)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t = __shfl_up(v, o);
        if (lane >= o) v = op(t, v);          // (earlier, later): the order matters for non-commutative operators —
    }                                         // round 1 had op(v, t), which composed the partition's overflow maps backwards
    return v;                                 // inside a wave and UNDER-estimated the carry behind a long run (mi_selftest_scan)
}

// exclusive scan across the block; `ident` is the identity; s_tmp needs nwaves+1 entries.
// returns the exclusive prefix for this thread and the block total through *total.
template <typename T, typename Op>
__device__ __forceinline__ T block_exclusive_scan(T v, Op op, T ident, T *s_tmp, T *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    T inc = wave_inclusive_scan(v, op);
    T exc = __shfl_up(inc, 1);
    if (lane == 0) exc = ident;
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (one thread walks the <= 16 wave totals; scanning them with the first wave's lanes instead measured 1 % slower
        //  in k_lz2_find: six 64-bit shuffle steps against eight short LDS round trips)
        T run = ident;
        for (int w = 0; w < nw; ++w) { T t = s_tmp[w]; s_tmp[w] = run; run = op(run, t); }
        s_tmp[nw] = run;
    }
    __syncthreads();
    T res = op(s_tmp[wave], exc);
    *total = s_tmp[nw];
    __syncthreads();
    return res;
}

struct OpAddU32 { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct OpMaxI32 { __device__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; } };

// ---- the overflow certificate's maps x -> max(x + a, b) (lz2_partition.hip, lzs.hip): composition is associative, not commutative
struct AffMax { int32_t a, b; };                     // x -> max(x + a, b)
__device__ __forceinline__ uint64_t am_pack(AffMax f) { return ((uint64_t)(uint32_t)f.a << 32) | (uint32_t)f.b; }
__device__ __forceinline__ AffMax am_unpack(uint64_t v) { AffMax f; f.a = (int32_t)(v >> 32); f.b = (int32_t)(uint32_t)v; return f; }
#define AM_NEG (-(1 << 28))
// (second after first)
__device__ __forceinline__ AffMax am_then(AffMax first, AffMax second)
{
    AffMax r;
    r.a = first.a + second.a;
    if (r.a < AM_NEG) r.a = AM_NEG;
    int32_t t = first.b + second.a;
    if (t < AM_NEG) t = AM_NEG;
    r.b = t > second.b ? t : second.b;
    return r;
}
struct OpAm { __device__ uint64_t operator()(uint64_t earlier, uint64_t later) const { return am_pack(am_then(am_unpack(earlier), am_unpack(later))); } };

// ---- one stable LSD radix pass over n elements held by a workgroup of NWAVES waves ----------------
// Wave w owns the contiguous segment [w*seg, (w+1)*seg): per-wave digit counts -> offsets by a
// (digit-major, wave-minor) scan -> each wave scatters its segment in order, ranking the 64
// elements of a step with ballots; no workgroup barrier inside the scatter.
//   load(i)  -> element (any trivially copyable type E) at input index i
//   digit(e) -> 0 .. (1<<NBITS)-1
//   store(j, e) writes element e to output index j
//   s_cnt    counters, used as a flat array [digit][NWAVES + 1]: the caller provides at least (NWAVES + 1) << NBITS
//            words (declare [NWAVES + 1][ROW], ROW >= 1 << NBITS).  Digit-major with an odd stride: the thread that
//            owns a digit reads its NWAVES counters at once (no read-modify-write chain through LDS) and the lanes of a
//            wave, which hit different digits, still spread over the banks.
//   counted  the caller has zeroed the counters and counted already: cnt[digit * (NWAVES + 1) + i / radix_seg<NWAVES>(n)] for every
//            input index i (it had the digits in hand: one pass over the input less)
//   hook(j, e)  called for every element with the output index it was stored at (e.g. to count the NEXT pass's digits)
struct RadixNoHook { __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {} };
template <int NWAVES> __device__ __forceinline__ uint32_t radix_seg(uint32_t n) { return ((n + (uint32_t)(NWAVES * 64) - 1u) / (uint32_t)(NWAVES * 64)) * 64u; }
template <int D, bool SKIP = false> struct RadixDepth { static constexpr int value = D; static constexpr bool skip_identity = SKIP; };
template <int NWAVES, int NBITS, typename E, typename CntRow, typename Load, typename Digit, typename Store, typename Hook = RadixNoHook, typename Depth = RadixDepth<1>>
__device__ __forceinline__ bool radix_pass(uint32_t n, CntRow *s_cnt, Load load, Digit digit, Store store, uint32_t arank = 0u /* LZP_ARANK | LZP_BREAK */, uint64_t *dbg = nullptr,
                                           bool counted = false, Hook hook = Hook(), Depth = Depth())
{
    long long tk_ = dbg ? clock64() : 0;
#define RP_TICK(k) do { if (dbg && threadIdx.x == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&dbg[k], (unsigned long long)(t2 - tk_)); tk_ = t2; } } while (0)
    static_assert(sizeof(CntRow) / sizeof(uint32_t) >= (1u << NBITS), "counter row too narrow for the digit");
    constexpr int ND = 1 << NBITS;
    constexpr int NT = NWAVES * 64;
    constexpr int ST = NWAVES + 1;
    uint32_t *cnt = &s_cnt[0][0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t seg = ((n + (uint32_t)NT - 1u) / (uint32_t)NT) * 64u;
    const uint32_t a = wave * seg, b = (a + seg < n) ? a + seg : n;
    if (!counted) {
        for (int i = tid; i < ND * ST; i += NT) cnt[i] = 0;
        __syncthreads();
        // four elements per lane and step: their (dependent) key lookups are in flight together — at 4 waves per SIMD a pass is
        // a chain of LDS round trips, not a stream of instructions
        if constexpr (Depth::value > 1) {
            // keys in HBM (the fallback's sorts): the loads of D steps UNCONDITIONALLY (indices clamped) before the first use — a
            // load under a condition waits for its value in its own basic block, and at one workgroup per CU nothing else hides it
            constexpr int D = Depth::value;
            for (uint32_t i = a + lane; i < b; i += 64u * D) {
                E ee[D]; uint32_t dg[D];
#pragma unroll
                for (int u = 0; u < D; ++u) ee[u] = load(i + 64u * u < b ? i + 64u * u : b - 1u);
#pragma unroll
                for (int u = 0; u < D; ++u) dg[u] = digit(ee[u]);
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    // 64 elements with ONE digit (a run of one byte value: equal words, equal homes) are one add, not 64 on one address
                    const bool v = i + 64u * u < b;
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg[u]);
                    const uint64_t live = __ballot(v);
                    if (__ballot(v && dg[u] != d0) == 0ull && (live & 1ull)) {               // (lane 0 is live: d0 is a live lane's digit)
                        if (lane == 0) atomicAdd(&cnt[d0 * ST + wave], (uint32_t)__popcll(live));
                    } else if (v) atomicAdd(&cnt[dg[u] * ST + wave], 1u);
                }
            }
        } else
        for (uint32_t i = a + lane; i < b; i += 256) {
            uint32_t dg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) dg[u] = (i + 64u * u < b) ? digit(load(i + 64u * u)) : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + 64u * u < b) atomicAdd(&cnt[dg[u] * ST + wave], 1u);
        }
    }
    __syncthreads();
    RP_TICK(8);
    // offsets: thread d < ND owns digit d: exclusive prefix over its waves in registers, then an exclusive scan over digits
    uint32_t pre[NWAVES];
    uint32_t tot = 0;
    if (tid < ND) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) pre[w] = cnt[tid * ST + w];
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { const uint32_t t = pre[w]; pre[w] = tot; tot += t; }
    }
    if constexpr (Depth::skip_identity) {
        // one digit holds every element (a block of one byte value on the fallback's sorts): the pass would copy its input — the
        // caller keeps the input instead (returns true; nothing has been stored)
        if (__syncthreads_or((tid < ND && tot == n) ? 1 : 0)) return true;
    }
    __shared__ uint32_t s_scan[18];
    uint32_t total;
    const uint32_t base = block_exclusive_scan<uint32_t>(tid < ND ? tot : 0u, OpAddU32(), 0u, s_scan, &total);
    if (tid < ND) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) cnt[tid * ST + w] = pre[w] + base;
    }
    __syncthreads();
    RP_TICK(9);
    if (arank & LZP_ARANK) {
        // Rank by the LDS itself: a returning add on the (digit, wave) cursor hands every lane of the instruction its slot,
        // and lanes that hit one address are served in LANE order (measured on gfx950, scripts/micro/lds_atomic_order.hip;
        // the context re-checks it when it is created and clears LZP_ARANK otherwise) — so the pass stays stable with ~10
        // instructions per 64 elements instead of the ~60 of the ballot ranking below (NBITS ballots build each lane's
        // peer mask).  The scatter was bound by VALU issue (SQ counters, profiles/r02a): this is where the instructions were.
        // Four 64-element steps at a time: loads and key lookups of all four first, then the four returning adds IN STEP
        // ORDER (one wave's LDS instructions execute in issue order, so step k's elements still rank before step k+1's),
        // then the stores.
        for (uint32_t i = a + lane; i < b; i += 256) {
            E ee[4]; uint32_t dg[4], sl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ee[u] = E{}; dg[u] = 0; if (i + 64u * u < b) { ee[u] = load(i + 64u * u); dg[u] = digit(ee[u]); } }
#pragma unroll
            for (int u = 0; u < 4; ++u) sl[u] = (i + 64u * u < b) ? atomicAdd(&cnt[dg[u] * ST + wave], 1u) : 0u;
#ifdef MI_TEST_HOOKS                                       /* compiled only into lib_test/ (csrc/Makefile): the hook cost the shipped scatter 2.5 % */
            if (arank & LZP_BREAK) {
                // test hook: what an out-of-order atomic would do — neighbours with one digit trade places (still a permutation)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool v = i + 64u * u < b;
                    const uint32_t dn = __shfl_xor(dg[u], 1), sn = __shfl_xor(sl[u], 1);
                    const bool vn = __shfl_xor(v ? 1 : 0, 1) != 0;
                    if (v && vn && dn == dg[u]) sl[u] = sn;
                }
            }
#endif
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + 64u * u < b) { store(sl[u], ee[u]); hook(sl[u], (uint32_t)ee[u]); }
        }
        __syncthreads();
        RP_TICK(10);
        return false;
    }
    if constexpr (Depth::value > 1) {
        // ballot ranking, D steps of 64 elements per round: all D loads, then all D key lookups, then the D rankings in step order
        constexpr int D = Depth::value;
        for (uint32_t i0 = a; i0 < b; i0 += 64u * D) {
            E ee[D]; uint32_t dg[D];
#pragma unroll
            for (int u = 0; u < D; ++u) { const uint32_t i = i0 + 64u * u + lane; ee[u] = load(i < b ? i : b - 1u); }
#pragma unroll
            for (int u = 0; u < D; ++u) dg[u] = digit(ee[u]);
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const bool valid = i0 + 64u * u + lane < b;
                const uint32_t d = dg[u];
                uint64_t mask = __ballot(valid);
                if (mask == 0ull) continue;                          // (uniform)
#pragma unroll
                for (int k = 0; k < NBITS; ++k) {
                    const bool bit = (d >> k) & 1u;
                    const uint64_t bal = __ballot(valid && bit);
                    mask &= bit ? bal : ~bal;
                }
                const uint64_t below = mask & ((1ull << lane) - 1ull);
                const uint32_t rank = __popcll(below), num = __popcll(mask);
                const int leader = __ffsll((unsigned long long)mask) - 1;
                uint32_t old = 0;
                if (valid && lane == leader) old = atomicAdd(&cnt[d * ST + wave], num);
                old = __shfl(old, leader < 0 ? 0 : leader);
                if (valid) { store(old + rank, ee[u]); hook(old + rank, (uint32_t)ee[u]); }
            }
        }
        __syncthreads();
        RP_TICK(10);
        return false;
    }
    // the element and its digit of the NEXT step are fetched while this step ranks and stores (two dependent LDS reads
    // off the chain: measured, the scatter is a chain of LDS round trips, 58 % of a pass)
    E en{};
    uint32_t dn = 0;
    if (a + lane < b) { en = load(a + lane); dn = digit(en); }
    for (uint32_t i0 = a; i0 < b; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < b;
        const E e = en;
        const uint32_t d = dn;
        if (i + 64 < b) { en = load(i + 64); dn = digit(en); }
        uint64_t mask = __ballot(valid);
#pragma unroll
        for (int k = 0; k < NBITS; ++k) {
            const bool bit = (d >> k) & 1u;
            const uint64_t bal = __ballot(valid && bit);
            mask &= bit ? bal : ~bal;
        }
        const uint64_t below = mask & ((1ull << lane) - 1ull);
        const uint32_t rank = __popcll(below), num = __popcll(mask);
        const int leader = __ffsll((unsigned long long)mask) - 1;
        uint32_t old = 0;
        if (valid && lane == leader) old = atomicAdd(&cnt[d * ST + wave], num);
        old = __shfl(old, leader < 0 ? 0 : leader);
        if (valid) { store(old + rank, e); hook(old + rank, (uint32_t)e); }
    }
    __syncthreads();
    RP_TICK(10);
    return false;
}

// Measured in k_lz2_find (MI_LZ_DEBUG counters, 3.5k keys, 8 waves): count 2.8k, offsets 2.0k, scatter 6.6k cycles per 8-bit
// pass; the scatter is bound by VALU issue — ~60 instructions per 64-element step, of which 5 per key bit build the
// 64-bit peer mask — not by its LDS round trips (prefetching the next element changed nothing).  Ranking through
// per-wave LDS masks (atomicOr of the lane bit, read back, popcount) costs the same at 6 bits and loses at 3 x 6 against
// 2 x 8 bits (12.35 vs 13.1 GB/s); it was removed.

template <int NBITS, typename E, typename CntRow, typename Load, typename Digit, typename Store>
__device__ __forceinline__ void radix_pass_1024(uint32_t n, CntRow *s_cnt, Load load, Digit digit, Store store, uint32_t arank = 0u)
{
    (void)radix_pass<16, NBITS, E>(n, s_cnt, load, digit, store, arank, nullptr, false, RadixNoHook(), RadixDepth<4>());
}
// ... the same, but a pass in which ONE digit holds every element stores nothing and returns true: the caller goes on from its input
template <int NBITS, typename E, typename CntRow, typename Load, typename Digit, typename Store>
__device__ __forceinline__ bool radix_pass_1024_or_skip(uint32_t n, CntRow *s_cnt, Load load, Digit digit, Store store)
{
    return radix_pass<16, NBITS, E>(n, s_cnt, load, digit, store, 0u, nullptr, false, RadixNoHook(), RadixDepth<4, true>());
}
