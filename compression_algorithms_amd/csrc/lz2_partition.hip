// lz2_partition.hip — stage 1 of the LDS-resident match finder: cut a block's positions into
// home-bucket ranges ("parts") that no probe cluster can span, and write each part's positions,
// in time order, to one contiguous list.
//
// Why: the first pipeline (lz_find.hip, kept as the fallback) sorts a whole 64 KiB block through
// global scratch; rocprofv3 shows 170x HBM traffic amplification from its scattered 2-byte
// stores (profiles/r01a).  A part of <= LZ2_CAP entries fits in LDS with everything the replay
// needs, so the sorts of stage 2 never leave the CU.
//
// A part boundary must not be crossed by a cluster.  Cluster extents are only known after the
// sort, so the boundary is certified conservatively from per-group counts (group = T/16384
// consecutive buckets, c_g entries):  overflow out of a group is at most
//     out_g = max(in_g + c_g - Gw, c_g - 1, 0)            (all entries on the last bucket)
// a composition of x -> max(x + a, b) maps, i.e. a parallel prefix scan.  Wherever out_g == 0
// no cluster crosses the end of group g.  The deflate flavour's insert wraps modulo T, so its
// bucket space is a ring: the ring is cut at the first certified point and every home is
// re-expressed relative to it (home' = home - base mod T); the lz77 flavour never wraps and
// keeps base = 0.
#include "lz_common.h"
#include "lz2.h"

__global__ __launch_bounds__(1024)
void k_lz2_partition(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, Lz2Scratch sc, uint64_t block0)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[LZ_MAX_BLOCK + LZ_TAIL + 16];
    __shared__ uint32_t s_grp[LZ2_NG];               // counts -> inclusive prefix; later the staging area
    __shared__ uint32_t s_safe[LZ2_NG / 32];
    __shared__ uint64_t s_scan64[18];
    __shared__ uint32_t s_scan32[18];
    __shared__ uint32_t s_thr[LZ2_MAXPARTS + 1];     // part k = home' in [s_thr[k], s_thr[k+1])
    __shared__ uint32_t s_pstart[LZ2_MAXPARTS + 1];
    __shared__ uint32_t s_s0, s_flag;

    const int tid = threadIdx.x;
    const uint32_t lb = blockIdx.x;
    long long tk = clock64();
#define PT_TICK(k) do { if (sc.dbg && tid == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&sc.dbg[32 + (k)], (unsigned long long)(t2 - tk)); tk = t2; } } while (0)
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    const uint32_t T = 1u << P.tbits, Tmask = T - 1u;
    const uint32_t gshift = P.tbits - LZ2_NG_BITS, Gw = 1u << gshift;

    lz_block_to_lds(s_in, src, n, (uint32_t)tid);
    for (uint32_t i = tid; i < LZ2_NG; i += 1024) s_grp[i] = 0;
    for (uint32_t i = tid; i < LZ2_NG / 32; i += 1024) s_safe[i] = 0;
    if (tid == 0) { s_s0 = ~0u; s_flag = 0; }
    __syncthreads();
    auto home_of = [&](uint32_t p) -> uint32_t { return lz_mix32(lds_word(s_in, p)) & Tmask; };
    PT_TICK(0);
    // the group of every position stays in registers (two 14-bit numbers per register, 64 positions per thread): the hash is
    // computed once — the pass that tabulates the part of every position further down used to compute it again
    uint32_t gcache[32];
#pragma unroll
    for (uint32_t i = 0; i < 64u; ++i) {
        const uint32_t p = (uint32_t)tid + 1024u * i;
        uint32_t g = 0;
        if (p < n) { g = home_of(p) >> gshift; atomicAdd(&s_grp[g], 1u); }
        if (i & 1u) gcache[i >> 1] |= g << 16; else gcache[i >> 1] = g;
    }
    __syncthreads();
    PT_TICK(1);

    // ---- overflow certificate: 16 consecutive groups per thread
    const uint32_t g0 = tid * (LZ2_NG / 1024);
    AffMax mine{0, AM_NEG};
    for (uint32_t k = 0; k < LZ2_NG / 1024; ++k) {
        const int32_t c = (int32_t)s_grp[g0 + k];
        mine = am_then(mine, AffMax{c - (int32_t)Gw, c > 0 ? c - 1 : 0});
    }
    uint64_t tot64;
    const uint64_t pre64 = block_exclusive_scan<uint64_t>(am_pack(mine), OpAm(), am_pack(AffMax{0, AM_NEG}), s_scan64, &tot64);
    const AffMax pre = am_unpack(pre64), total = am_unpack(tot64);
    // carry into group 0: the ring's fixed point for deflate (F(0) = total.b since total.a = n - T < 0), 0 for lz77
    const int32_t x0 = P.deflate ? (total.b > 0 ? total.b : 0) : 0;
    {
        int32_t x = x0 + pre.a > pre.b ? x0 + pre.a : pre.b;
        if (x < 0) x = 0;
        uint32_t safe_bits = 0;
        for (uint32_t k = 0; k < LZ2_NG / 1024; ++k) {
            const int32_t c = (int32_t)s_grp[g0 + k];
            int32_t o = x + c - (int32_t)Gw;
            const int32_t o2 = c > 0 ? c - 1 : 0;
            o = o > o2 ? o : o2;
            if (o < 0) o = 0;
            if (o == 0) safe_bits |= 1u << k;
            x = o;
        }
        // 16 groups per thread: two threads share a 32-bit word
        atomicOr(&s_safe[g0 >> 5], safe_bits << (g0 & 31u));
        if (safe_bits) atomicMin(&s_s0, g0 + (uint32_t)__builtin_ctz(safe_bits));
    }
    // inclusive prefix of counts, in place
    {
        uint32_t sum = 0;
        for (uint32_t k = 0; k < LZ2_NG / 1024; ++k) sum += s_grp[g0 + k];
        uint32_t tot;
        uint32_t run = block_exclusive_scan<uint32_t>(sum, OpAddU32(), 0u, s_scan32, &tot);
        for (uint32_t k = 0; k < LZ2_NG / 1024; ++k) { run += s_grp[g0 + k]; s_grp[g0 + k] = run; }
    }
    __syncthreads();

    PT_TICK(2);
    // ---- cut the ring / choose the parts
    Lz2BlockMeta *mt = sc.meta + lb;
    const uint32_t s0 = s_s0;
    const bool no_safe = (s0 == ~0u);
    // rotated group order starts after s0 (deflate) or at group 0 (lz77: nothing wraps into bucket 0)
    const uint32_t gstart = P.deflate ? (no_safe ? 0u : ((s0 + 1u) & (LZ2_NG - 1u))) : 0u;
    const uint32_t base = gstart << gshift;
    auto cum_incl = [&](uint32_t gr) -> uint32_t {      // entries in rotated groups 0..gr
        const uint32_t g = (gstart + gr) & (LZ2_NG - 1u);
        const uint32_t before = gstart ? s_grp[gstart - 1] : 0u;
        return g >= gstart ? s_grp[g] - before : s_grp[g] + (n - before);
    };
    auto is_safe = [&](uint32_t gr) -> bool {
        const uint32_t g = (gstart + gr) & (LZ2_NG - 1u);
        return (s_safe[g >> 5] >> (g & 31u)) & 1u;
    };
    // Greedy cuts: every part takes as many entries as stage 2 can hold (LZ2_CAP), ending at the last certified group
    // that fits.  A fixed target with a fixed margin (parts of TS + "how far back the cut had to move") fails as soon as
    // a block holds one cluster larger than the margin — measured: target 3072 gave +5 % on one corpus and a 6x collapse
    // (every block in the fallback) on another seed whose text has a 1000-entry cluster — while greedy parts only fail when
    // a single cluster exceeds LZ2_CAP.  The cuts depend on each other, so one wave walks them (<= 32 64-way searches).
    __shared__ uint32_t s_K;
    if (tid < 64) {                                     // wave 0, every lane with the same cur / glo / k
        const uint32_t lane = (uint32_t)tid;
        uint32_t k = 0, cur = 0, glo = 0;               // parts so far, entries before the current part, its first rotated group
        bool bad = false;
        if (lane == 0) s_thr[0] = 0;
        // last certified group g in [glo, .) with at most `limit` entries in rotated groups 0..g, or -1
        auto cut_below = [&](uint32_t limit) -> int32_t {
            // (a part of a 2^20-bucket table is also kept below 2^16 homes: stage 2 then sorts 16-bit keys in two radix
            //  passes instead of three; a 2^22-bucket table spreads 4096 entries over ~2^18 homes whatever the cut)
            const uint32_t span = (gshift <= 6u) ? (65536u >> gshift) : LZ2_NG;
            // first rotated group in [glo, hi) whose inclusive count exceeds the limit, else hi: a 64-way search
            uint32_t lo = glo, hi = (glo + span < LZ2_NG) ? glo + span : LZ2_NG - 1;
            while (lo < hi) {
                const uint32_t len = hi - lo, step = (len + 63u) / 64u;
                const uint32_t pr = lo + lane * step;
                const bool over = (pr < hi) && cum_incl(pr) > limit;
                const uint64_t mk = __ballot(over);
                if (mk == 0ull) lo = lo + ((len - 1u) / step) * step + 1u;         // every probe is below: continue behind the last one
                else {
                    const uint32_t j = (uint32_t)__builtin_ctzll(mk);
                    hi = lo + j * step;
                    if (j) lo = lo + (j - 1u) * step + 1u;
                }
            }
            // boundary after the last certified group before it, inside this part: 64 groups per look
            for (int32_t g = (int32_t)lo - 1; g >= (int32_t)glo; g -= 64) {
                const int32_t c = g - (int32_t)lane;
                const uint64_t mk = __ballot(c >= (int32_t)glo && is_safe((uint32_t)c));
                if (mk) return g - (int32_t)__builtin_ctzll(mk);
            }
            return -1;
        };
        // parts of at most LZ2_CAP_S entries (three workgroups of stage 2 per CU); where no certified cut lies that close — one
        // cluster of more entries — a part of up to LZ2_CAP entries (k_lz2_find_wide: two per CU)
        while (n - cur > LZ2_CAP_S) {
            int32_t gr = cut_below(cur + LZ2_CAP_S);
            if (gr < (int32_t)glo) {
                if (n - cur <= LZ2_CAP) break;                                     // the rest is one wide part
                gr = cut_below(cur + LZ2_CAP);
            }
            // no certified cut within LZ2_CAP entries: one cluster larger than stage 2 can hold.  (k + 2 > LZ2_MAXPARTS cannot
            // happen — the static_assert in lz2.h has the argument — and stays as a guard.)
            if (gr < (int32_t)glo || k + 2 > LZ2_MAXPARTS) { bad = true; break; }
            cur = cum_incl((uint32_t)gr);
            glo = (uint32_t)gr + 1u;
            ++k;
            if (lane == 0) s_thr[k] = glo << gshift;
        }
#ifdef MI_TEST_HOOKS                                       /* lib_test only (MI_LZ_TEST_FORCE_FALLBACK=1): every block takes the exit above, so a whole */
        if (P.flags & LZP_FORCE_FB) bad = true;                /* batch of ordinary text runs through the fallback chain (tests/test_fallback_chain_gpu.py) */
#endif
        if (lane == 0) {
            s_thr[k + 1] = T;                           // the last part takes everything up to the cut
            s_K = k + 1;
            if (bad) s_flag = 1;
        }
    }
    __syncthreads();
    const uint32_t K = s_K;                             // <= LZ2_MAXPARTS
    PT_TICK(3);
    // part sizes; refuse the block if any part exceeds the LDS capacity of stage 2
    if (tid < K) {
        const uint32_t a = s_thr[tid], b = s_thr[tid + 1];
        auto cnt_below = [&](uint32_t h) -> uint32_t {  // entries with home' < h (h is a multiple of Gw or T)
            if (h == 0) return 0u;
            return cum_incl((h >> gshift) - 1u);
        };
        const uint32_t cnt = b > a ? cnt_below(b) - cnt_below(a) : 0u;
        s_pstart[tid] = cnt_below(a);
        if (cnt > LZ2_CAP) atomicOr(&s_flag, 1u);
        mt->part_start[tid] = cnt_below(a);
        mt->part_count[tid] = cnt;
        mt->part_lo[tid] = a;
    }
    if (tid == 0) {
        s_pstart[K] = n;
        mt->n = n; mt->nparts = K; mt->base = base;
        mt->nbig = 0; mt->nbig_entries = 0;
    }
    __syncthreads();
    if (tid == 0) {
        const bool fb = (s_flag != 0) || (P.deflate && no_safe);
        mt->fallback = fb ? 1u : 0u;
        if (fb) { const uint32_t k = atomicAdd(sc.fallback_count, 1u); sc.fallback_list[k] = lb; }
    }
    if (s_flag != 0 || (P.deflate && no_safe)) {
        if (tid == 0 && P.stats) atomicAdd((unsigned long long *)&P.stats[0], 1ull);      // mi_lz_path_stats: blocks sent to the fallback
        return;
    }
    // this block's parts join the batch's work list: stage 2 runs one workgroup per LISTED part.  (A grid of
    // blocks x "most parts a block can have" interleaved a third of empty workgroups with the real ones; each still had to
    // be given stage 2's 67 KiB of LDS before it could leave: 14.65 -> 11.4 GB/s.)
    {
        __shared__ uint32_t s_wbase, s_wbase2;
        __shared__ uint8_t s_wrank[LZ2_MAXPARTS];
        if (tid == 0) {
            uint32_t ns = 0, nw = 0;
            for (uint32_t k = 0; k < K; ++k) s_wrank[k] = (uint8_t)(mt->part_count[k] <= LZ2_CAP_S ? ns++ : nw++);
            s_wbase = atomicAdd(sc.work_count, ns);
            s_wbase2 = nw ? atomicAdd(sc.work_count + 1, nw) : 0u;
            if (nw && P.stats) atomicAdd((unsigned long long *)&P.stats[1], (unsigned long long)nw);       // mi_lz_path_stats: wide parts
        }
        __syncthreads();
        if (tid < (int)K) {
            const uint64_t item = (uint64_t)lb | ((uint64_t)tid << 16) | ((uint64_t)mt->part_count[tid] << 24) | ((uint64_t)mt->part_start[tid] << 40);
            if (mt->part_count[tid] <= LZ2_CAP_S) sc.work[s_wbase + s_wrank[tid]] = item;
            else sc.work[sc.work_slots - 1u - (s_wbase2 + s_wrank[tid])] = item;          // the wide list grows from the end (lz2.h)
        }
    }

    // ---- positions -> part lists, time order kept: ONE stable radix pass over the whole block by part number.  The part of
    //      every position is tabulated first (one hash per position), the pass then stores straight to the block's list in
    //      HBM: consecutive positions of one part land on consecutive addresses, and the exclusive scan over the parts is
    //      exactly the list layout (part k starts at the number of entries in parts < k).  (Staging 8192 positions at a
    //      time through LDS paid the pass's fixed costs eight times: 7.5 -> see DESIGN.md for the measurement.)
    uint8_t  *part_in = reinterpret_cast<uint8_t *>(s_grp);             // [65536] part of every position; the group array is dead now
    __shared__ __attribute__((aligned(16))) uint8_t s_gpart[LZ2_NG];   // part of every rotated group (part boundaries are group boundaries)
    {
        // sixteen consecutive groups per thread: one binary search for the first, then a walk along the thresholds
        constexpr uint32_t GPT = LZ2_NG / 1024;
        const uint32_t gr0 = (uint32_t)tid * GPT;
        uint32_t lo = 0, hi = K - 1;                    // last k with thr[k] <= h
        { const uint32_t h = gr0 << gshift; while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_thr[mid] <= h) lo = mid; else hi = mid - 1; } }
        uint32_t nxt = lo + 1 < K ? s_thr[lo + 1] : 0xFFFFFFFFu;
        for (uint32_t k = 0; k < GPT; ++k) {
            const uint32_t h = (gr0 + k) << gshift;
            while (h >= nxt) { ++lo; nxt = lo + 1 < K ? s_thr[lo + 1] : 0xFFFFFFFFu; }
            s_gpart[gr0 + k] = (uint8_t)lo;
        }
    }
    __syncthreads();
    PT_TICK(4);
    {
        const uint32_t gstart_ = base >> gshift;        // the ring is cut on a group boundary
#pragma unroll
        for (uint32_t i = 0; i < 64u; ++i) {
            const uint32_t p = (uint32_t)tid + 1024u * i;
            const uint32_t g = (gcache[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu;
            part_in[p] = (p < n) ? s_gpart[(g - gstart_) & (LZ2_NG - 1u)] : (uint8_t)0xFF;      // 0xFF: no position (parts are numbered < 128)
        }
    }
    __syncthreads();
    PT_TICK(5);
    // The part of every position goes out as it stands, 64 KiB of bytes, coalesced.  Round 3 sorted the positions into
    // per-part lists here (one stable radix pass over the block: 87 k of this kernel's 205 k cycles — 27 distinct digits, so the
    // lanes of every counting and ranking atomic pile up on a few LDS addresses — and 128 KiB of scattered 2-byte stores);
    // now every workgroup of stage 2 picks its own positions out of this map with byte compares and a prefix sum, in time
    // order by construction (lz2_find.hip).
    {
        uint4 *dst = reinterpret_cast<uint4 *>(sc.partmap + (size_t)lb * LZ_MAX_BLOCK);
        const uint4 *srcv = reinterpret_cast<const uint4 *>(part_in);
        for (uint32_t i = tid; i < LZ_MAX_BLOCK / 16u; i += 1024u) dst[i] = srcv[i];
    }
    PT_TICK(6);
    if (sc.dbg && tid == 0) atomicAdd((unsigned long long *)&sc.dbg[47], 1ull);
}

void lz2_launch_partition(const uint8_t *d_in, uint64_t n, const LzP &P, const Lz2Scratch &sc, uint64_t block0, uint32_t nb, hipStream_t s)
{
    hipLaunchKernelGGL(k_lz2_partition, dim3(nb), dim3(1024), 0, s, d_in, n, P, sc, block0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-test of block_exclusive_scan with a NON-commutative operator (development / test hook, not in the public header):
// composition of the partition's own x -> max(x + a, b) maps over 1024 threads must equal the serial left-to-right
// composition.  The overflow certificate is only a certificate if earlier maps are applied first.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void k_selftest_scan(const int32_t *__restrict__ ab, uint64_t *__restrict__ out)
{
    __shared__ uint64_t s_tmp[18];
    const int tid = threadIdx.x;
    uint64_t tot;
    const uint64_t pre = block_exclusive_scan<uint64_t>(am_pack(AffMax{ab[2 * tid], ab[2 * tid + 1]}), OpAm(), am_pack(AffMax{0, AM_NEG}), s_tmp, &tot);
    out[tid] = pre;
    if (tid == 0) out[1024] = tot;
}

extern "C" int mi_selftest_scan(mi_ctx *ctx, const int32_t *h_ab /* [1024][2] */, uint64_t *h_out /* [1025] packed a << 32 | b */)
{
    if (!ctx || !h_ab || !h_out) return 0;
    int32_t *d_ab = nullptr; uint64_t *d_out = nullptr;
    if (hipMalloc(&d_ab, 2048 * 4) != hipSuccess || hipMalloc(&d_out, 1025 * 8) != hipSuccess) return 0;
    int ok = hipMemcpy(d_ab, h_ab, 2048 * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) { hipLaunchKernelGGL(k_selftest_scan, dim3(1), dim3(1024), 0, 0, d_ab, d_out); ok = hipDeviceSynchronize() == hipSuccess; }
    if (ok) ok = hipMemcpy(h_out, d_out, 1025 * 8, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_ab); (void)hipFree(d_out);
    return ok;
}
