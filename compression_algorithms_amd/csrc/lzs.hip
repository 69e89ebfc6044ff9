// lzs.hip — the lz77 flavour on blocks above 64 KiB, sliced in TIME so that everything stays in LDS.
//
// lzw.hip clusters a whole block at once: 2^20 entries on 2^22 buckets are a 25 % load "of all time", clusters of 20 000
// entries, serial chains of that length.  But at any moment the table only holds the last W = 2^wbits insertions — 1.5 %.
// So a block is walked in sub-blocks of W positions.  Step k knows the table it starts from exactly: the W entries of
// sub-block k-1, each ON THE SLOT step k-1 gave it.  During step k every one of them retires — entry t of the old
// sub-block right after insertion t of the new one (lz77.c:70-76: insertion p clears the bucket insertion p-W recorded)
// — and every new entry is inserted, none of them retired.  The events of a step are therefore
//     e = 2 t      find() + insert of position kW + t
//     e = 2 t + 1  clear the bucket recorded by position (k-1)W + t          (whoever sits there now)
// in the order of e, and their table coordinates are known up front: an old entry's slot, a new entry's home bucket.
// Treating an old entry as "inserted first, home = its slot" the isolation argument of DESIGN.md 2.1 holds unchanged: sort
// the 2W events by coordinate, parking sweep -> clusters that never touch each other.  With 131 072 events on 2^22
// buckets the clusters are those of a 64 KiB block (a frequent word's alive copies), not of a megabyte.
//
//   k_lzs_keys   mix32(word) of every position, once per block                      (the hash, lz77.c:13-41)
//   k_lzs_part   per block and step: events -> parts of <= 4096 events that no cluster crosses (the overflow certificate
//                of lz2_partition.hip over 32 768 groups), lists in event order
//   k_lzs_find   per part, all in LDS: stable radix sort by coordinate, parking sweep, (cluster, time) order by cursor
//                placement, replay — a LANE per cluster of <= 16 events (occupancy and occupants in three registers),
//                a WAVE per larger one (64 buckets per probe step) — then find() results and the new entries' slots
//                go out by position
// Zero-filled ring: in step 0 one pseudo event clears bucket 0 after insertion W-1; an entry that a
// clear removes early is carried into the next step as DEAD (it occupies nothing, its own clear still happens).
// A step with a cluster above 4096 events (a run of one byte value) or more than 63 parts flags the block; flagged blocks
// are then redone by lzw.hip (whole-block clusters, any size): one by one when they are few, the whole batch otherwise.
#include "lz_common.h"
#include <stdlib.h>
#include <stdio.h>

#define LZS_NG_BITS   15
#define LZS_NG        (1u << LZS_NG_BITS)
#ifndef LZS_CAP
#define LZS_CAP       2560u                       // events per part: 50 KiB of LDS, THREE workgroups of k_lzs_find per CU (4096: 75 KiB, two; same-box A/B 6.44 -> 6.81 / 4.06 -> 4.18 GB/s)
#endif
#define LZS_THREADS   512
#define LZS_NWAVES    (LZS_THREADS / 64)
#define LZS_MAXPARTS  128u                        // room in the meta records; part numbers are bytes of the step's part map (0xFF: no event)
#define LZS_KMAX      (LZS_CAP >= 4096u ? 64u : 96u)   // parts a step may cut per block (k_lzs_find's grid covers that many): 2 W / LZS_CAP is 32 / 51 on text
static_assert(LZS_KMAX <= LZS_MAXPARTS && LZS_KMAX < 255u, "part numbers are bytes");
#ifndef LZS_LANE_MAX
#define LZS_LANE_MAX  16u                         // clusters of up to this many events are replayed inside k_lzs_find, one LANE each (16: registers only).
                                                  // 64 (occupants in LDS, lzs_replay_lane64; no k_lzs_mid launches) was built and measured in round 4:
                                                  // parity green, but the workgroup waits for its longest 64-event lane — k_lzs_find 10.8 -> 13.8 ms,
                                                  // 6.81 -> 6.36 GB/s at 256 KiB blocks, 4.18 -> 3.48 at 1 MiB (make EXTRA=-DLZS_LANE_MAX=64u)
#endif
#define LZS_DEAD      0x80000000u
#define LZS_NONE      0xFFFFFFFFu
// entry flags in LDS
#define CF_OLD        (1u << 30)
#define CF_DEAD       (1u << 31)
#define CF_MASK       0x3FFFFFFFu
#define RF_SLOT       0x0FFFu
#define RF_OLD        0x1000u
#define RF_DEAD       0x2000u
#define ES_KILLED     0x8000u
#define RF_BIG        0x4000u                     // the event's cluster is replayed by k_lzs_big
#define LZS_BIG_SMALL 512u                        // class 0: 17..512 events (4 KiB of LDS per wave), class 1: up to LZS_CAP (32 KiB)
#define LZS_CTR_WORDS 1024u                       // counters per group of blocks: [0..63] parts of step k, [128 + 8k ..] exports, [768 + 2k ..] replay cursors
#define LZS_NCLS      4u                          // export classes: 0: 17..32 and 1: 33..64 events (a lane per cluster), 2: 65..512 and 3: up to LZS_CAP (a wave per cluster)
#define LZS_MID_MAX   64u
#define BI_EID        0x1FFFFu                    // big_info: event id | slot relative to the cluster << 17 | dead << 29

struct LzsMeta {
    uint32_t nparts, fallback, pad[2];
    uint32_t part_start[LZS_MAXPARTS], part_count[LZS_MAXPARTS], part_lo[LZS_MAXPARTS];
};

struct LzsScratch {
    uint32_t *key;        // [nb][S] mix32(word) by position
    uint32_t *slot;       // [nb][S] bucket an entry was inserted on | LZS_DEAD
    uint32_t *cand;       // [nb][S] find() by position (LZS_NONE = none)
    uint32_t *plist;      // [nb][S] event ids of the current step, grouped by part, event order inside a part (written by k_lzs_find)
    uint8_t  *pmap;       // [nb][2 W] the part of every event id of the current step (0xFF: not an event), written by k_lzs_part
    LzsMeta  *meta;       // [nb]
    uint64_t *work;       // [nb * LZS_MAXPARTS] parts of the current step: block | part << 16 | events << 24 | list start << 40
    uint32_t *counters;   // [0..63] parts listed in step k, [64] flagged blocks, [128 + 8 k ..] step k: exported events, clusters of class 0..3
    uint32_t *big_key, *big_info;   // [nb * S] events of the exported clusters of the current step, (cluster, event) order
    uint64_t *big_desc[LZS_NCLS];   // per class: first event | count << 32 | block << 48
    uint32_t *flag_count; // flagged blocks of the whole batch
    uint32_t *flag_list;  // their indices in the batch
    uint32_t  lb0;        // batch index of this group's first block
    uint32_t  S;
    uint64_t *dbg;        // phase cycle counters of k_lzs_find (MI_LZ_DEBUG=1), else NULL
};

__device__ __forceinline__ uint32_t lzs_word(const uint8_t *src, uint32_t p, uint32_t n)
{
    if (p + 8u <= n && ((((uintptr_t)src) & 3u) == 0)) {
        const uint32_t *a = reinterpret_cast<const uint32_t *>(src + (p & ~3u));
        const uint64_t v = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
        return (uint32_t)(v >> ((p & 3u) * 8u));
    }
    uint32_t w = 0;                                   // bytes past the block end read as zero
    for (uint32_t k = 0; k < 4 && p + k < n; ++k) w |= (uint32_t)src[p + k] << (8 * k);
    return w;
}

__global__ __launch_bounds__(256)
void k_lzs_keys(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, LzsScratch sc, uint64_t block0)
{
    const uint32_t lb = blockIdx.y;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    uint32_t *key = sc.key + (size_t)lb * sc.S;
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) key[p] = lz_mix32(lzs_word(src, p, n));
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc.meta[lb].fallback = 0; sc.meta[lb].nparts = 0; }
}

// =============================================================================================
// stage 1: the events of one step of one block -> parts
// =============================================================================================
__global__ __launch_bounds__(1024)
void k_lzs_part(uint64_t n_total, LzP P, LzsScratch sc, uint64_t block0, uint32_t step)
{
    __shared__ uint32_t s_grp[LZS_NG];                // counts -> inclusive prefix; later the part of every event id (bytes)
    __shared__ uint32_t s_safe[LZS_NG / 32];
    __shared__ uint64_t s_scan64[18];
    __shared__ uint32_t s_scan32[18];
    __shared__ uint32_t s_thr[LZS_MAXPARTS + 1];      // part k = coordinates in [s_thr[k], s_thr[k+1])
    __shared__ uint32_t s_flag, s_K, s_wbase;

    const int tid = threadIdx.x;
    const uint32_t lb = blockIdx.x;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint32_t W = 1u << P.wbits, t0 = step * W;
    long long tkp = clock64();
#define LZP_TICK(k) do { if (sc.dbg && tid == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&sc.dbg[16 + (k)], (unsigned long long)(t2 - tkp)); tkp = t2; } } while (0)
    LzsMeta *mt = sc.meta + lb;
    if (t0 >= n || mt->fallback) { if (tid == 0) mt->nparts = 0; return; }
    const uint32_t nnew = (n - t0) < W ? (n - t0) : W;
    const uint32_t *key = sc.key + (size_t)lb * sc.S + t0;
    const uint32_t *slot_old = sc.slot + (size_t)lb * sc.S + (t0 - (step ? W : 0u));
    const uint32_t T = 1u << P.tbits, Tmask = T - 1u;
    const uint32_t gshift = P.tbits - LZS_NG_BITS, Gw = 1u << gshift;
    const uint32_t NE = 2u * W;                        // event id space
    auto valid = [&](uint32_t e) -> bool {
        const uint32_t t = e >> 1;
        return (e & 1u) ? (step ? true : (t == W - 1u && nnew == W)) : (t < nnew);
    };
    auto grp = [&](uint32_t c) -> uint32_t { const uint32_t g = c >> gshift; return g < LZS_NG ? g : LZS_NG - 1u; };   // slots past T: the last group

    for (uint32_t i = tid; i < LZS_NG; i += 1024) s_grp[i] = 0;
    for (uint32_t i = tid; i < LZS_NG / 32; i += 1024) s_safe[i] = 0;
    if (tid == 0) s_flag = 0;
    __syncthreads();
    // coordinates of a thread's events, EB at a time: the (dependent-free) global loads of a batch are issued together.  One
    // load per iteration — as this loop was written until round 4 — is a chain of 128 HBM round trips on the ONE workgroup per
    // block that every step of the block waits for: most of this kernel's 0.26 ms at 95 blocks (kernel timeline, round 4)
    constexpr uint32_t EB = 16u;
    auto load_batch = [&](uint32_t e0, uint32_t (&raw)[EB]) {
        // every load of the batch is UNCONDITIONAL (the address is chosen with a select, out-of-range ids read a clamped
        // element): a load under an `if` sits in a basic block of its own together with the wait for its value, and sixteen
        // such blocks are sixteen round trips one after the other — what the first "batched" form of this code still did
#pragma unroll
        for (uint32_t u = 0; u < EB; ++u) {
            const uint32_t e = e0 + u * 1024u;
            uint32_t t = e >> 1;
            const uint32_t tmax = (e & 1u) ? W - 1u : nnew - 1u;      // (a new position past the block's end, an id past 2 W: clamped, then ignored)
            if (t > tmax) t = tmax;
            const uint32_t *ptr = (e & 1u) ? slot_old + t : key + t;
            raw[u] = *ptr;
        }
    };
    auto coord_of = [&](uint32_t e, uint32_t raw, bool &ok) -> uint32_t {
        ok = e < NE && valid(e);
        return !ok ? 0u : (e & 1u) ? (step ? (raw & ~LZS_DEAD) : 0u) : (raw & Tmask);
    };
    // (the next batch's loads are in flight while this batch's events are worked on)
    {
        uint32_t raw[EB], nraw[EB];
        load_batch((uint32_t)tid, raw);
        for (uint32_t e0 = tid; e0 < NE; e0 += EB * 1024u) {
            const bool more = e0 + EB * 1024u < NE;
            if (more) load_batch(e0 + EB * 1024u, nraw);
#pragma unroll
            for (uint32_t u = 0; u < EB; ++u) {
                bool ok;
                const uint32_t c = coord_of(e0 + u * 1024u, raw[u], ok);
                if (ok) atomicAdd(&s_grp[grp(c)], 1u);
            }
            if (more) {
#pragma unroll
                for (uint32_t u = 0; u < EB; ++u) raw[u] = nraw[u];
            }
        }
    }
    __syncthreads();

    LZP_TICK(0);
    // ---- overflow certificate (lz2_partition.hip): out_g = max(in_g + c_g - Gw, c_g - 1, 0), a scan of x -> max(x + a, b) maps
    constexpr uint32_t GPT = LZS_NG / 1024;
    const uint32_t g0 = tid * GPT;
    AffMax mine{0, AM_NEG};
    for (uint32_t k = 0; k < GPT; ++k) {
        const int32_t c = (int32_t)s_grp[g0 + k];
        mine = am_then(mine, AffMax{c - (int32_t)Gw, c > 0 ? c - 1 : 0});
    }
    uint64_t tot64;
    const uint64_t pre64 = block_exclusive_scan<uint64_t>(am_pack(mine), OpAm(), am_pack(AffMax{0, AM_NEG}), s_scan64, &tot64);
    const AffMax pre = am_unpack(pre64);
    {
        int32_t x = pre.a > pre.b ? pre.a : pre.b;      // carry into group 0 is 0: nothing wraps (lz77.c:61)
        if (x < 0) x = 0;
        uint32_t safe_bits = 0;
        for (uint32_t k = 0; k < GPT; ++k) {
            const int32_t c = (int32_t)s_grp[g0 + k];
            int32_t o = x + c - (int32_t)Gw;
            const int32_t o2 = c > 0 ? c - 1 : 0;
            o = o > o2 ? o : o2;
            if (o < 0) o = 0;
            if (o == 0) safe_bits |= 1u << k;
            x = o;
        }
        s_safe[g0 >> 5] = safe_bits;                    // 32 groups per thread: one word each
    }
    {
        uint32_t sum = 0;
        for (uint32_t k = 0; k < GPT; ++k) sum += s_grp[g0 + k];
        uint32_t tot;
        uint32_t run = block_exclusive_scan<uint32_t>(sum, OpAddU32(), 0u, s_scan32, &tot);
        for (uint32_t k = 0; k < GPT; ++k) { run += s_grp[g0 + k]; s_grp[g0 + k] = run; }
    }
    __syncthreads();
    const uint32_t nev = s_grp[LZS_NG - 1];
    auto is_safe = [&](uint32_t g) -> bool { return (s_safe[g >> 5] >> (g & 31u)) & 1u; };

    LZP_TICK(1);
    // ---- greedy cuts: every part takes as many events as stage 2 holds, ending at the last certified group that fits
    if (tid < 64) {
        const uint32_t lane = (uint32_t)tid;
        uint32_t k = 0, cur = 0, glo = 0;
        bool bad = false;
        if (lane == 0) s_thr[0] = 0;
        while (nev - cur > LZS_CAP) {
            const uint32_t limit = cur + LZS_CAP;
            uint32_t lo = glo, hi = LZS_NG - 1;         // first group whose inclusive count exceeds the limit: a 64-way search
            while (lo < hi) {
                const uint32_t len = hi - lo, st = (len + 63u) / 64u;
                const uint32_t pr = lo + lane * st;
                const bool over = (pr < hi) && s_grp[pr] > limit;
                const uint64_t mk = __ballot(over);
                if (mk == 0ull) lo = lo + ((len - 1u) / st) * st + 1u;
                else {
                    const uint32_t j = (uint32_t)__builtin_ctzll(mk);
                    hi = lo + j * st;
                    if (j) lo = lo + (j - 1u) * st + 1u;
                }
            }
            int32_t gr = -1;                            // last certified group before it, inside this part
            for (int32_t g = (int32_t)lo - 1; g >= (int32_t)glo; g -= 64) {
                const int32_t c = g - (int32_t)lane;
                const uint64_t mk = __ballot(c >= (int32_t)glo && is_safe((uint32_t)c));
                if (mk) { gr = g - (int32_t)__builtin_ctzll(mk); break; }
            }
            if (gr < (int32_t)glo || k + 3 > LZS_KMAX) { bad = true; break; }          // one cluster above the capacity / too many parts
            cur = s_grp[gr];
            glo = (uint32_t)gr + 1u;
            ++k;
            if (lane == 0) s_thr[k] = glo << gshift;
        }
        if (lane == 0) {
            s_thr[k + 1] = 0xFFFFFFFFu;                 // the last part takes everything above
            s_K = k + 1;
            if (bad) s_flag = 1;
        }
    }
    __syncthreads();
    const uint32_t K = s_K;                             // <= LZS_KMAX - 1
    LZP_TICK(2);
    if (tid < (int)K) {
        auto cnt_below = [&](uint32_t h) -> uint32_t { return h == 0 ? 0u : (h == 0xFFFFFFFFu ? nev : s_grp[(h >> gshift) - 1u]); };
        const uint32_t a = s_thr[tid], b = s_thr[tid + 1];
        const uint32_t cnt = cnt_below(b) - cnt_below(a);
        if (cnt > LZS_CAP) atomicOr(&s_flag, 1u);
        mt->part_start[tid] = cnt_below(a);
        mt->part_count[tid] = cnt;
        mt->part_lo[tid] = a;
    }
    __syncthreads();
    if (s_flag) {
        if (tid == 0) { mt->fallback = 1; mt->nparts = 0; sc.flag_list[atomicAdd(sc.flag_count, 1u)] = sc.lb0 + lb; }
        return;
    }
    if (tid == 0) { mt->nparts = K; s_wbase = atomicAdd(&sc.counters[step], K); }
    __syncthreads();
    if (tid < (int)K) sc.work[s_wbase + tid] = (uint64_t)lb | ((uint64_t)tid << 16) | ((uint64_t)mt->part_count[tid] << 24) | ((uint64_t)mt->part_start[tid] << 40);

    // ---- the part of every event id (a binary search over the thresholds), one byte each, goes out as it stands: 2 W bytes,
    //      coalesced.  Every workgroup of stage 2 picks its own events out of this map (byte compares + a prefix sum: event
    //      order by construction).  Round 3 sorted the ids into per-part lists here with one stable radix pass over 2 W ids —
    //      half of this kernel, and it runs on ONE workgroup per block and step, on the chain every step waits for.
    uint8_t *part_in = reinterpret_cast<uint8_t *>(s_grp);          // [NE <= 131072] (the group array is dead: barrier above)
    // the part of every PAIR of groups (of its first coordinate): one binary search per pair, 16 pairs per thread
    __shared__ uint8_t s_gp2[LZS_NG / 2u];
    for (uint32_t g2 = tid; g2 < LZS_NG / 2u; g2 += 1024u) {
        const uint32_t c = g2 << (gshift + 1u);
        uint32_t lo = 0, hi = K - 1;                    // last k with thr[k] <= c
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_thr[mid] <= c) lo = mid; else hi = mid - 1; }
        s_gp2[g2] = (uint8_t)lo;
    }
    __syncthreads();
    LZP_TICK(3);
    {
        uint32_t raw[EB], nraw[EB];
        load_batch((uint32_t)tid, raw);
        for (uint32_t e0 = tid; e0 < NE; e0 += EB * 1024u) {
            const bool more = e0 + EB * 1024u < NE;
            if (more) load_batch(e0 + EB * 1024u, nraw);
#pragma unroll
            for (uint32_t u = 0; u < EB; ++u) {
                const uint32_t e = e0 + u * 1024u;
                if (e >= NE) continue;
                bool ok;
                const uint32_t c = coord_of(e, raw[u], ok);
                // the part of the coordinate's group PAIR from the table, plus one where a part boundary lies between the pair's two
                // groups (boundaries are group boundaries: at most one inside a pair): two dependent LDS reads, no loop, where the
                // binary search over ~53 thresholds took six
                uint32_t g2 = c >> (gshift + 1u);
                if (g2 >= LZS_NG / 2u) g2 = LZS_NG / 2u - 1u;
                uint32_t pk = s_gp2[g2];
                pk += (pk + 1u < K && s_thr[pk + 1u] <= c) ? 1u : 0u;
                part_in[e] = ok ? (uint8_t)pk : (uint8_t)0xFFu;
            }
            if (more) {
#pragma unroll
                for (uint32_t u = 0; u < EB; ++u) raw[u] = nraw[u];
            }
        }
    }
    __syncthreads();
    LZP_TICK(4);
    {
        uint4 *dst = reinterpret_cast<uint4 *>(sc.pmap + (size_t)lb * NE);
        const uint4 *srcv = reinterpret_cast<const uint4 *>(part_in);
        for (uint32_t i = tid; i < NE / 16u; i += 1024u) dst[i] = srcv[i];
    }
    LZP_TICK(5);
    if (sc.dbg && tid == 0) atomicAdd((unsigned long long *)&sc.dbg[23], 1ull);
}

// =============================================================================================
// stage 2: one part, in LDS
// =============================================================================================
// a cluster of <= 16 events [s, s + m) on one lane: 16 occupancy bits and sixteen 4-bit occupants in three registers
__device__ __forceinline__ void lzs_replay_lane(const uint32_t *e_key, const uint16_t *e_rf, uint16_t *e_slot, uint16_t *cand_i, uint32_t s, uint32_t m)
{
    uint32_t mask = 0;
    uint64_t tab = 0;
    bool any_new = false;
    for (uint32_t li = 0; li < m; ++li) {               // the table the step starts from: old entries that are still there
        const uint32_t rf = e_rf[s + li];
        if ((rf & (RF_OLD | RF_DEAD)) == RF_OLD) { const uint32_t b = (rf & RF_SLOT) - s; mask |= 1u << b; tab |= (uint64_t)li << (4u * b); }
        any_new |= !(rf & RF_OLD);
    }
    if (!any_new) return;                               // nobody asks
    for (uint32_t li = 0; li < m; ++li) {
        const uint32_t i = s + li, rf = e_rf[i], b0 = (rf & RF_SLOT) - s;
        if (rf & RF_OLD) {                              // clear the recorded bucket, whoever sits there (lz77.c:70-76)
            if ((mask >> b0) & 1u) {
                const uint32_t o = (uint32_t)(tab >> (4u * b0)) & 15u;
                if (o != li && !(e_rf[s + o] & RF_OLD)) e_slot[s + o] |= ES_KILLED;
                mask &= ~(1u << b0);
            }
        } else {
            const uint32_t key = e_key[i];
            uint32_t res = 0xFFFFu;
            for (uint32_t b = b0; (mask >> b) & 1u; ++b) {                 // find(): lz77.c:94-108
                const uint32_t o = (uint32_t)(tab >> (4u * b)) & 15u;
                if (e_key[s + o] == key) { res = s + o; break; }
            }
            cand_i[i] = (uint16_t)res;
            const uint32_t fb = b0 + (uint32_t)__builtin_ctz(~(mask >> b0));   // first fit, inside the cluster by the parking bound
            mask |= 1u << fb;
            tab = (tab & ~(15ull << (4u * fb))) | ((uint64_t)li << (4u * fb));
            e_slot[i] = (uint16_t)(s + fb);
        }
    }
}

// a cluster of 17..64 events [s, s + m) on one lane: occupancy in a 64-bit register, the occupant of every slot (its index inside the
// cluster, a byte) in the part's own LDS — `occ8` is indexed by slot like e_slot, and a cluster owns its slots [s, s + m).  Round 3
// exported these clusters (8 bytes per event through HBM) to lane kernels of their own (k_lzs_mid: 16 / 32 KiB of LDS per wave, two
// more launches on every step's chain) or, where those did not pay (few blocks), to the wave replay at ~75 CU-cycles per event; in
// here they cost the workgroup its longest such lane: <= 64 events of a few LDS round trips each.
__device__ __forceinline__ void lzs_replay_lane64(const uint32_t *e_key, const uint16_t *e_rf, uint16_t *e_slot, uint16_t *cand_i, uint8_t *occ8,
                                                  uint32_t s, uint32_t m)
{
    uint64_t mask = 0;
    bool any_new = false;
    for (uint32_t li = 0; li < m; ++li) {               // the table the step starts from: old entries that are still there
        const uint32_t rf = e_rf[s + li];
        if ((rf & (RF_OLD | RF_DEAD)) == RF_OLD) { const uint32_t b = (rf & RF_SLOT) - s; mask |= 1ull << b; occ8[s + b] = (uint8_t)li; }
        any_new |= !(rf & RF_OLD);
    }
    if (!any_new) return;                               // nobody asks
    for (uint32_t li = 0; li < m; ++li) {
        const uint32_t i = s + li, rf = e_rf[i], b0 = (rf & RF_SLOT) - s;
        if (rf & RF_OLD) {                              // clear the recorded bucket, whoever sits there (lz77.c:70-76)
            if ((mask >> b0) & 1ull) {
                const uint32_t o = occ8[s + b0];
                if (o != li && !(e_rf[s + o] & RF_OLD)) e_slot[s + o] |= ES_KILLED;
                mask &= ~(1ull << b0);
            }
        } else {
            const uint32_t key = e_key[i];
            const uint32_t fb = b0 + (uint32_t)__builtin_ctzll(~(mask >> b0));   // first fit, inside the cluster by the parking bound
            uint32_t res = 0xFFFFu;
            for (uint32_t b = b0; b < fb; ++b) {                             // find(): lz77.c:94-108 — every slot of [b0, fb) is occupied
                const uint32_t o = occ8[s + b];
                if (e_key[s + o] == key) { res = s + o; break; }
            }
            cand_i[i] = (uint16_t)res;
            mask |= 1ull << fb;
            occ8[s + fb] = (uint8_t)li;
            e_slot[i] = (uint16_t)(s + fb);
        }
    }
}

// use_mid = 0: everything up to 512 events goes to the wave replay (few blocks per step: two more launches on the step's
// latency chain cost more than they save)
__host__ __device__ __forceinline__ uint32_t lzs_export_class(uint32_t cm, uint32_t use_mid)
{
    return (use_mid && cm <= LZS_MID_MAX) ? (cm <= 32u ? 0u : 1u) : cm <= LZS_BIG_SMALL ? 2u : 3u;
}

__global__ __launch_bounds__(LZS_THREADS)
void k_lzs_find(uint64_t n_total, LzP P, LzsScratch sc, uint64_t block0, uint32_t step, uint32_t use_mid)
{
    __shared__ uint32_t s_key[LZS_CAP];                 // mix32(word) by event index j; later cand_i (u16, replay order)
    __shared__ uint32_t s_c[LZS_CAP];                   // coordinate - part_lo | CF_OLD | CF_DEAD by j; later e_key (replay order)
    __shared__ uint16_t s_j0[LZS_CAP], s_j1[LZS_CAP];   // sort ping-pong; later e_rf / the replay order -> j
    __shared__ __attribute__((aligned(16))) uint16_t s_gr[2 * LZS_CAP];   // s_g | s_r; during the sort: the counters of every other pass
    uint16_t *const s_g = s_gr;                          // cluster number by j; later occ
    uint16_t *const s_r = s_gr + LZS_CAP;                // slot index by j; later e_slot (replay order)
    __shared__ uint32_t s_cnt[LZS_NWAVES + 1][256];     // radix counters; later cluster cursors (u16, rows 0..7) and the wave list (row 8)
    __shared__ int32_t  s_i32[18];
    __shared__ uint64_t s_u64[18];
    __shared__ uint32_t s_ngroups, s_nbig;
    __shared__ uint16_t s_bigm[LZS_CAP / (LZS_LANE_MAX + 1u) + 2];   // sizes of the clusters this part exports

    const int tid = threadIdx.x;
    const uint32_t nwork = sc.counters[step];
    // XCD-aware order (lz2_find.hip): every XCD takes a contiguous eighth of the list, a block's parts share an L2
    const uint32_t cpx = (nwork + 7u) >> 3, item_idx = (blockIdx.x & 7u) * cpx + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= cpx || item_idx >= nwork) return;
    const uint64_t item = sc.work[item_idx];             // list position and length ride in the item: the list loads do not wait for the meta record
    const uint32_t lb = (uint32_t)item & 0xFFFFu, part = (uint32_t)(item >> 16) & 0xFFu;
    const LzsMeta *mt = sc.meta + lb;
    const uint32_t m = (uint32_t)(item >> 24) & 0xFFFFu;
    if (m == 0) return;
    const uint32_t pstart = (uint32_t)(item >> 40), plo = mt->part_lo[part];
    const uint32_t phi = (part + 1 < mt->nparts) ? mt->part_lo[part + 1] : 0xFFFFFFFFu;
    const uint32_t W = 1u << P.wbits, t0 = step * W, Tmask = (1u << P.tbits) - 1u;
    (void)n_total; (void)block0;
    long long tk = clock64();
#define LZS_TICK(k) do { if (sc.dbg && tid == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&sc.dbg[k], (unsigned long long)(t2 - tk)); tk = t2; } } while (0)
    uint32_t *plist = sc.plist + (size_t)lb * sc.S + pstart;
    const uint32_t *key_new = sc.key + (size_t)lb * sc.S + t0;
    const uint32_t *key_old = sc.key + (size_t)lb * sc.S + (t0 - (step ? W : 0u));
    const uint32_t *slot_old = sc.slot + (size_t)lb * sc.S + (t0 - (step ? W : 0u));
    uint32_t *slot_new = sc.slot + (size_t)lb * sc.S + t0;
    uint32_t *cand = sc.cand + (size_t)lb * sc.S + t0;
    const uint32_t arank = P.flags & LZP_ARANK;          // (the test hook LZP_BREAK is for the 64 KiB pipeline only: this replay trusts event order for its indices)
    constexpr uint32_t CH = LZS_CAP / LZS_THREADS;
    bool viol = false;                                   // order checks (lz_common.h lz_order_violation)

    // the sort's digits are counted where they are in hand (k_lz2_find): the first pass's here, while the coordinates are put
    // together, each later pass's while the one before it scatters; the second counter array sits in the idle s_g / s_r
    uint32_t *const cntA = &s_cnt[0][0];
    uint32_t (*const s_cntB)[256] = reinterpret_cast<uint32_t (*)[256]>(s_gr);
    uint32_t *const cntB = &s_cntB[0][0];
    constexpr uint32_t RST = LZS_NWAVES + 1;
    const uint32_t seg = radix_seg<LZS_NWAVES>(m), seg_inv = (uint32_t)((0x100000000ull + seg - 1u) / seg);
    for (uint32_t i = tid; i < 256u * RST; i += LZS_THREADS) { cntA[i] = 0; cntB[i] = 0; }
    // ---- this part's events, in event order: picked out of the step's part map (k_lzs_part: one byte per event id).  Thread t
    //      looks at ids [BPT t, BPT t + BPT) — 16-byte loads — keeps one match bit per id, a prefix sum over the threads gives its
    //      place, and it writes its ids there in ascending order (into s_key, which the keys only take over further down).
    {
        const uint32_t NE = 2u * W;
        constexpr uint32_t BPTMAX = 256u;                                   // W <= 65536: at most 131072 / 512 bytes of the map per thread
        const uint32_t BPT = NE / LZS_THREADS;                              // (a multiple of 16 for W >= 4096; smaller windows take the byte loop)
        const uint8_t *pm = sc.pmap + (size_t)lb * NE + (size_t)tid * BPT;
        const uint32_t pat = part * 0x01010101u;
        auto nib = [&](uint32_t w) -> uint32_t {                            // bit k set: byte k of w equals `part`
            const uint32_t x = w ^ pat;
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
            return (((z >> 7) * 0x00204081u) >> 21) & 15u;
        };
        uint32_t bm[BPTMAX / 32u];
#pragma unroll
        for (uint32_t k = 0; k < BPTMAX / 32u; ++k) bm[k] = 0u;
        if ((BPT & 15u) == 0u) {
#pragma unroll
            for (uint32_t q = 0; q < BPTMAX / 16u; ++q) {
                if (q * 16u < BPT) {
                    const uint4 v = reinterpret_cast<const uint4 *>(pm)[q];
                    bm[q >> 1] |= (nib(v.x) | (nib(v.y) << 4) | (nib(v.z) << 8) | (nib(v.w) << 12)) << (16u * (q & 1u));
                }
            }
        } else {
            for (uint32_t i = 0; i < BPT; ++i) if (pm[i] == (uint8_t)part) bm[i >> 5] |= 1u << (i & 31u);
        }
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < BPTMAX / 32u; ++k) mine += (uint32_t)__popc(bm[k]);
        __shared__ uint32_t s_scanp[LZS_NWAVES + 2];
        uint32_t total_m;
        uint32_t at = block_exclusive_scan<uint32_t>(mine, OpAddU32(), 0u, s_scanp, &total_m);
        (void)total_m;                                                      // == m: k_lzs_part counted the same bytes
        const uint32_t e0 = (uint32_t)tid * BPT;
#pragma unroll
        for (uint32_t k = 0; k < BPTMAX / 32u; ++k) {
            uint32_t z = bm[k];
            while (z) {
                const uint32_t bpos = (uint32_t)__builtin_ctz(z);
                z &= z - 1u;
                if (at < LZS_CAP) s_key[at] = e0 + 32u * k + bpos;
                ++at;
            }
        }
    }
    __syncthreads();
    // ---- gather: keys and slots of all of a thread's events together; the ids also go to the part's list in HBM (coalesced): the
    //      permutation further down needs every event's time again and LDS has no room to keep it until then
    {
        uint32_t ge[CH], gk[CH], gs[CH];
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t j = tid + c * LZS_THREADS;
            ge[c] = j < m ? s_key[j] : 0u;
            if (j < m) plist[j] = ge[c];
        }
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t j = tid + c * LZS_THREADS, t = ge[c] >> 1;
            gk[c] = 0; gs[c] = LZS_DEAD;                 // step 0's only "old" event: clear bucket 0, nobody's entry
            if (j < m) {
                if (ge[c] & 1u) { if (step) { gk[c] = key_old[t]; gs[c] = slot_old[t]; } }
                else gk[c] = key_new[t];
            }
        }
        __syncthreads();                                    // the counters are zero (the loads above are in flight across it)
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t j = tid + c * LZS_THREADS;
            if (j < m) {
                s_key[j] = gk[c];
                const uint32_t cv = (ge[c] & 1u) ? (((gs[c] & ~LZS_DEAD) - plo) | CF_OLD | ((gs[c] & LZS_DEAD) ? CF_DEAD : 0u)) : ((gk[c] & Tmask) - plo);
                s_c[j] = cv;
                atomicAdd(&cntA[(cv & 255u) * RST + __umulhi(j, seg_inv)], 1u);
            }
        }
    }
    if (tid == 0) s_nbig = 0;
    __syncthreads();
    auto keyp = [&](uint32_t j) -> uint32_t { return s_c[j] & CF_MASK; };
    LZS_TICK(0);

    // ---- stable sort of the event indices by coordinate: 8-bit passes over 16 or 24 bits -> s_j0
    if (phi - plo > 65536u) {
        radix_pass<LZS_NWAVES, 8, uint32_t>(m, s_cnt, [&](uint32_t i) { return i; },
            [&](uint32_t e) { return keyp(e) & 255u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank, nullptr, true,
            [&](uint32_t d, uint32_t e) { atomicAdd(&cntB[((keyp(e) >> 8) & 255u) * RST + __umulhi(d, seg_inv)], 1u); });
        for (uint32_t i = tid; i < 256u * RST; i += LZS_THREADS) cntA[i] = 0;         // (the next pass opens with a barrier)
        radix_pass<LZS_NWAVES, 8, uint32_t>(m, s_cntB, [&](uint32_t i) { return (uint32_t)s_j0[i]; },
            [&](uint32_t e) { return (keyp(e) >> 8) & 255u; }, [&](uint32_t d, uint32_t e) { s_j1[d] = (uint16_t)e; }, arank, nullptr, true,
            [&](uint32_t d, uint32_t e) { atomicAdd(&cntA[((keyp(e) >> 16) & 255u) * RST + __umulhi(d, seg_inv)], 1u); });
        radix_pass<LZS_NWAVES, 8, uint32_t>(m, s_cnt, [&](uint32_t i) { return (uint32_t)s_j1[i]; },
            [&](uint32_t e) { return (keyp(e) >> 16) & 255u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank, nullptr, true);
    } else {
        radix_pass<LZS_NWAVES, 8, uint32_t>(m, s_cnt, [&](uint32_t i) { return i; },
            [&](uint32_t e) { return keyp(e) & 255u; }, [&](uint32_t d, uint32_t e) { s_j1[d] = (uint16_t)e; }, arank, nullptr, true,
            [&](uint32_t d, uint32_t e) { atomicAdd(&cntB[((keyp(e) >> 8) & 255u) * RST + __umulhi(d, seg_inv)], 1u); });
        radix_pass<LZS_NWAVES, 8, uint32_t>(m, s_cntB, [&](uint32_t i) { return (uint32_t)s_j1[i]; },
            [&](uint32_t e) { return (keyp(e) >> 8) & 255u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank, nullptr, true);
    }

    LZS_TICK(1);
    // ---- parking sweep over the sorted order: with g_k = coordinate_k - k, cluster heads are the weak prefix
    //      maxima of g; a cluster of m' events owns exactly the m' consecutive slots [first index, first index + m')
    const uint32_t k0 = tid * CH, k1 = (k0 + CH < m) ? k0 + CH : m;
    uint16_t *cur16 = reinterpret_cast<uint16_t *>(&s_cnt[0][0]);     // per cluster: its first replay index, post-incremented by the placement
    {
        uint32_t rj[CH]; int32_t rh[CH];
        int32_t mx = INT32_MIN;
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t k = k0 + c;
            rj[c] = 0; rh[c] = 0;
            if (k < k1) { rj[c] = s_j0[k]; rh[c] = (int32_t)keyp(rj[c]); const int32_t g = rh[c] - (int32_t)k; mx = g > mx ? g : mx; }
        }
        int32_t gmax_total;
        const int32_t premax = block_exclusive_scan<int32_t>(mx, OpMaxI32(), INT32_MIN, s_i32, &gmax_total);
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
        struct OpHeads {                                 // heads so far (sum) | last head index + 1 (max)
            __device__ uint64_t operator()(uint64_t a, uint64_t b) const {
                const uint64_t s0 = (a & 0xFFFFu) + (b & 0xFFFFu), a1 = a >> 16, b1 = b >> 16;
                return s0 | ((a1 > b1 ? a1 : b1) << 16);
            }
        };
        uint64_t tot2;
        const uint64_t pre2 = block_exclusive_scan<uint64_t>((uint64_t)nheads | ((uint64_t)(lasthead + 1) << 16), OpHeads(), 0ull, s_u64, &tot2);
        const uint32_t gid_base = (uint32_t)(pre2 & 0xFFFFu);
        const int32_t gs_carry = (int32_t)(pre2 >> 16) - 1;
        if (tid == 0) s_ngroups = (uint32_t)(tot2 & 0xFFFFu);
        int32_t run = premax;
        uint32_t cur_gs = 0, cur_gid = gid_base; int32_t cur_base = 0;
        if (k0 < m && gs_carry >= 0) { cur_gs = (uint32_t)gs_carry; cur_base = (int32_t)keyp(s_j0[cur_gs]); cur_gid = gid_base - 1u; }
        uint32_t seen = 0;
        int32_t prev_h = 0; uint32_t prev_j = 0;
        if (k0 > 0 && k0 < m) { prev_j = s_j0[k0 - 1]; prev_h = (int32_t)keyp(prev_j); }
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t k = k0 + c;
            if (k < k1) {
                const uint32_t j = rj[c];
                const int32_t h = rh[c], g = h - (int32_t)k;
                const bool head = (k == 0) || (g >= run);
                run = g > run ? g : run;
                if (k > 0 && (h < prev_h || (h == prev_h && j < prev_j))) viol = true;      // (coordinate, event) ascending
                prev_h = h; prev_j = j;
                if (head) { cur_gs = k; cur_base = h; cur_gid = gid_base + seen; ++seen; cur16[cur_gid] = (uint16_t)k; }
                s_g[j] = (uint16_t)cur_gid;
                s_r[j] = (uint16_t)(cur_gs + (uint32_t)(h - cur_base));
            }
        }
    }
    __syncthreads();
    const uint32_t ngroups = s_ngroups;
    LZS_TICK(2);

    // ---- (cluster, event) order: an entry's place is its cluster's cursor, post-incremented in event order (lz2_find.hip:
    //      every cursor pair is advanced by one wave only; LDS is in order, returning adds are served in lane order)
    if (arank & LZP_ARANK) {
        uint32_t *cur32 = &s_cnt[0][0];
        const uint32_t wv = (uint32_t)tid >> 6, ln = (uint32_t)tid & 63u;
        for (uint32_t j0 = 0; j0 < m; j0 += 256u) {
            uint32_t gg[4], old[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) { const uint32_t j = j0 + 64u * u + ln; gg[u] = j < m ? (uint32_t)s_g[j] : 0xFFFFFFFFu; }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                old[u] = 0;
                if (gg[u] != 0xFFFFFFFFu && ((gg[u] >> 1) % (uint32_t)LZS_NWAVES) == wv) old[u] = atomicAdd(&cur32[gg[u] >> 1], 1u << (16u * (gg[u] & 1u)));
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (gg[u] != 0xFFFFFFFFu && ((gg[u] >> 1) % (uint32_t)LZS_NWAVES) == wv) s_j1[(old[u] >> (16u * (gg[u] & 1u))) & 0xFFFFu] = (uint16_t)(j0 + 64u * u + ln);
        }
        __syncthreads();
    } else {
        // without the lane-order guarantee: two stable 6-bit passes by cluster number, then the cluster ends for the replay
        __syncthreads();
        radix_pass<LZS_NWAVES, 6, uint32_t>(m, s_cnt, [&](uint32_t i) { return i; },
            [&](uint32_t e) { return (uint32_t)s_g[e] & 63u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, false);
        radix_pass<LZS_NWAVES, 6, uint32_t>(m, s_cnt, [&](uint32_t i) { return (uint32_t)s_j0[i]; },
            [&](uint32_t e) { return (uint32_t)s_g[e] >> 6; }, [&](uint32_t d, uint32_t e) { s_j1[d] = (uint16_t)e; }, false);
        for (uint32_t i = tid; i < m; i += LZS_THREADS) {
            const uint32_t g = s_g[s_j1[i]];
            if (i + 1 == m || s_g[s_j1[i + 1]] != g) cur16[g] = (uint16_t)(i + 1u);
        }
        __syncthreads();
    }
    // cur16[g] is now the END of cluster g; it starts where g - 1 ends
    LZS_TICK(3);

    // ---- permute into replay order; the event ids come back from the list here (all of a thread's loads in flight together)
    //      so that nothing after this point waits for a dependent global load
    uint32_t *e_key = s_c;
    uint16_t *e_rf = s_j0;
    uint16_t *cand_i = reinterpret_cast<uint16_t *>(s_key);              // s_key is dead after the permutation: results ...
    uint16_t *e_t = cand_i + LZS_CAP;                                    // ... and the time of every event, replay order
    {
        uint32_t rk[CH], rr[CH], rt[CH];
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t i = tid + c * LZS_THREADS;
            // (written by this workgroup in the gather: an agent-scope load goes past the L1, which may hold nothing newer but is not
            //  coherent.  UNCONDITIONAL, index clamped, value used further down: a load inside a conditional expression waits for its
            //  value in its own basic block, and the thread's loads would be one round trip after the other)
            rt[c] = __hip_atomic_load(&plist[s_j1[i < m ? i : m - 1u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t i = tid + c * LZS_THREADS;
            rk[c] = 0; rr[c] = 0;
            if (i < m) {
                const uint32_t j = s_j1[i], cf = s_c[j];
                if (i > 0) {                                 // the (cluster, event) order, however it was made
                    const uint32_t jp = s_j1[i - 1], gp = s_g[jp], gj = s_g[j];
                    if (gp > gj || (gp == gj && jp > j)) viol = true;
                }
                rk[c] = s_key[j];
                rr[c] = (uint32_t)s_r[j] | ((cf & CF_OLD) ? RF_OLD : 0u) | ((cf & CF_DEAD) ? RF_DEAD : 0u);
            }
        }
        if (viol) lz_order_violation(P);
        __syncthreads();
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t i = tid + c * LZS_THREADS;
            if (i < m) { e_key[i] = rk[c]; e_rf[i] = (uint16_t)rr[c]; e_t[i] = (uint16_t)(rt[c] >> 1); }
        }
    }
    uint16_t *e_slot = s_r;
    uint16_t *lane_list = s_g;                            // cluster numbers are dead: clusters of 2..16 events, largest class first
    uint32_t *s_big = &s_cnt[LZS_NWAVES][0];              // <= 4096 / 17 clusters above the lane size: 240 words of the last row
    constexpr uint32_t NLC = 6u;                          // lane classes: 33..64, 17..32, 9..16, 5..8, 3..4, 2 events — the longest chains first
    __shared__ uint32_t s_ccnt[NLC], s_cbase[NLC], s_ebase, s_dbase[LZS_NCLS];
    uint8_t *occ8 = reinterpret_cast<uint8_t *>(s_j1);   // occupant of every slot, for the lanes that replay 17..64 events (s_j1 is dead after the permutation)
    if (tid < NLC) s_ccnt[tid] = 0;
    __syncthreads();
    LZS_TICK(4);

    // ---- classify the clusters: singletons are answered on the spot, 2..16 events go to four size classes (a wave whose
    //      lanes hold clusters of one class does not wait for one long lane), larger ones are listed for export
    constexpr uint32_t GPT = LZS_CAP / LZS_THREADS;      // clusters per thread, at most
    uint32_t my_cls[GPT], my_rank[GPT];
#pragma unroll
    for (uint32_t c = 0; c < GPT; ++c) {
        const uint32_t g = tid + c * LZS_THREADS;
        my_cls[c] = NLC; my_rank[c] = 0;
        if (g < ngroups) {
            const uint32_t cs = g ? (uint32_t)cur16[g - 1] : 0u, cm = (uint32_t)cur16[g] - cs;
            if (cm == 1u) { if (!(e_rf[cs] & RF_OLD)) { cand_i[cs] = 0xFFFFu; e_slot[cs] = (uint16_t)cs; } }     // nobody to find, its home is free
            else if (cm <= LZS_LANE_MAX) { my_cls[c] = cm > 32u ? 0u : cm > 16u ? 1u : cm > 8u ? 2u : cm > 4u ? 3u : cm > 2u ? 4u : 5u; my_rank[c] = atomicAdd(&s_ccnt[my_cls[c]], 1u); }
            else { const uint32_t q = atomicAdd(&s_nbig, 1u); s_big[q] = cs | (cm << 16); s_bigm[q] = (uint16_t)cm; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t c = 0; c < NLC; ++c) { s_cbase[c] = run; run += s_ccnt[c]; }
        // one reservation per part for what it exports (events, clusters per class): the returns are not needed before the
        // lane replay is over
        const uint32_t nbig = s_nbig;
        uint32_t ev = 0, nc[LZS_NCLS] = {0, 0, 0, 0};
        for (uint32_t c = 0; c < nbig; ++c) { const uint32_t cm = s_bigm[c]; ev += cm; ++nc[lzs_export_class(cm, use_mid)]; }
        uint32_t *ctr = sc.counters + 128 + 8 * step;
        s_ebase = nbig ? atomicAdd(&ctr[0], ev) : 0u;
        for (uint32_t q = 0; q < LZS_NCLS; ++q) s_dbase[q] = nc[q] ? atomicAdd(&ctr[1 + q], nc[q]) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < GPT; ++c) if (my_cls[c] < NLC) lane_list[s_cbase[my_cls[c]] + my_rank[c]] = (uint16_t)(tid + c * LZS_THREADS);
    __syncthreads();
    const uint32_t nlane = s_cbase[NLC - 1u] + s_ccnt[NLC - 1u];
    for (uint32_t q = tid; q < nlane; q += LZS_THREADS) {
        const uint32_t g = lane_list[q];
        const uint32_t cs = g ? (uint32_t)cur16[g - 1] : 0u, cm = (uint32_t)cur16[g] - cs;
        if (cm <= 16u) lzs_replay_lane(e_key, e_rf, e_slot, cand_i, cs, cm);
        else lzs_replay_lane64(e_key, e_rf, e_slot, cand_i, occ8, cs, cm);
    }
    __syncthreads();
    LZS_TICK(5);
    if (sc.dbg && tid == 0) { atomicAdd((unsigned long long *)&sc.dbg[8], 1ull); atomicAdd((unsigned long long *)&sc.dbg[9], (unsigned long long)m);
                              atomicAdd((unsigned long long *)&sc.dbg[10], (unsigned long long)ngroups); atomicAdd((unsigned long long *)&sc.dbg[11], (unsigned long long)s_nbig);
                              uint32_t be = 0, h[3] = {0, 0, 0};
                              for (uint32_t c = 0; c < s_nbig; ++c) { const uint32_t cm = s_bigm[c]; be += cm; h[0] += cm > 512 ? cm : 0; h[1] += cm > 1024 ? cm : 0; h[2] += cm > 2048 ? cm : 0; }
                              atomicAdd((unsigned long long *)&sc.dbg[12], (unsigned long long)be);
                              for (int q = 0; q < 3; ++q) atomicAdd((unsigned long long *)&sc.dbg[13 + q], (unsigned long long)h[q]); }
    // ---- larger clusters leave for k_lzs_big (a wave each, little LDS, many per CU): in here the workgroup would wait for
    //      its longest chain — measured: 74 % of this kernel
    {
        const uint32_t nbig = s_nbig;
        if (nbig) {
            if (tid == 0) {
                uint32_t run = s_ebase, dn[LZS_NCLS] = {s_dbase[0], s_dbase[1], s_dbase[2], s_dbase[3]};
                for (uint32_t c = 0; c < nbig; ++c) {
                    const uint32_t cm = s_bigm[c], q = lzs_export_class(cm, use_mid);
                    sc.big_desc[q][dn[q]++] = (uint64_t)run | ((uint64_t)cm << 32) | ((uint64_t)lb << 48);
                    s_big[c] = (s_big[c] & 0xFFFFu) | ((run - s_ebase) << 16);        // start in replay order | offset in the reservation (< 4096)
                    run += cm;
                }
            }
            __syncthreads();
            const uint32_t ebase = s_ebase;
            const uint32_t wv = (uint32_t)tid >> 6, ln = (uint32_t)tid & 63u;
            for (uint32_t c = wv; c < nbig; c += LZS_NWAVES) {                         // a wave per cluster: coalesced records
                const uint32_t cs = s_big[c] & 0xFFFFu, ob = s_big[c] >> 16, cm = s_bigm[c];
                for (uint32_t idx = ln; idx < cm; idx += 64u) {
                    const uint32_t i = cs + idx, rf = e_rf[i];
                    sc.big_key[ebase + ob + idx] = e_key[i];
                    sc.big_info[ebase + ob + idx] = ((uint32_t)e_t[i] << 1) | ((rf & RF_OLD) ? 1u : 0u) | (((rf & RF_SLOT) - cs) << 17) | ((rf & RF_DEAD) ? (1u << 29) : 0u);
                    e_rf[i] = (uint16_t)(rf | RF_BIG);
                }
            }
        }
    }
    __syncthreads();
    LZS_TICK(6);
    // ---- results out, by position: find() as a block position, the new entry's bucket for the next step
    const uint32_t base_new = t0, base_old = t0 - (step ? W : 0u);
    for (uint32_t i = tid; i < m; i += LZS_THREADS) {
        const uint32_t rf = e_rf[i];
        if (rf & (RF_OLD | RF_BIG)) continue;
        const uint32_t t = e_t[i];
        const uint32_t sl = e_slot[i];
        slot_new[t] = ((e_key[i] & Tmask) + ((sl & RF_SLOT) - (rf & RF_SLOT))) | ((sl & ES_KILLED) ? LZS_DEAD : 0u);
        const uint32_t ci = cand_i[i];
        uint32_t res = LZS_NONE;
        if (ci != 0xFFFFu) res = ((e_rf[ci] & RF_OLD) ? base_old : base_new) + (uint32_t)e_t[ci];
        cand[t] = res;
    }
    LZS_TICK(7);
}

#define LZS_EMPTY 0xFFFFFFFFu
// =============================================================================================
// stage 3a: a LANE per exported cluster of 17..64 events (64 clusters per wave).  The wave replay below spends ~75 CU-cycles
// of scalar issue per event; here an event is a few dozen vector instructions shared by 64 clusters.  Per lane: occupancy
// in one 64-bit register, per slot {occupant's mixed word, event id} in LDS (word-interleaved over the lanes: no bank
// conflicts), records four at a time (16-byte loads, the next four in flight).
// =============================================================================================
template <uint32_t CAPM>
__global__ __launch_bounds__(64)
void k_lzs_mid(LzP P, LzsScratch sc, uint32_t step, uint32_t cls)
{
    __shared__ uint32_t s_okey[CAPM * 64u], s_oeid[CAPM * 64u];
    const uint32_t lane = threadIdx.x;
    const uint32_t W = 1u << P.wbits, t0 = step * W, Tmask = (1u << P.tbits) - 1u;
    const uint32_t base_new = t0, base_old = t0 - (step ? W : 0u);
    const uint32_t count = sc.counters[128 + 8 * step + 1 + cls];
    for (uint32_t wg = blockIdx.x; wg * 64u < count; wg += gridDim.x) {
        const uint32_t ci = wg * 64u + lane;
        uint32_t first = 0, m = 0, lb = 0;
        if (ci < count) { const uint64_t d = sc.big_desc[cls][ci]; first = (uint32_t)d; m = (uint32_t)(d >> 32) & 0xFFFFu; lb = (uint32_t)(d >> 48); }
        uint32_t *slot_new = sc.slot + (size_t)lb * sc.S + t0;
        uint32_t *cand = sc.cand + (size_t)lb * sc.S + t0;
        // aligned groups of four records covering [first, first + m)
        const uint32_t g0 = first >> 2, skip = first & 3u, ng = m ? (skip + m + 3u) >> 2 : 0u;
        const uint4 *vk = reinterpret_cast<const uint4 *>(sc.big_key) + g0, *vi = reinterpret_cast<const uint4 *>(sc.big_info) + g0;
        uint64_t mask = 0;
        auto word_of = [](const uint4 &v, uint32_t k) -> uint32_t { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; };
        // ---- the table the step starts from: old entries that are still there
        {
            uint4 nk = make_uint4(0, 0, 0, 0), ni = nk;
            if (ng) { nk = vk[0]; ni = vi[0]; }
            for (uint32_t g = 0; __ballot(g < ng) != 0ull; ++g) {
                const uint4 ck = nk, cinf = ni;
                if (g + 1 < ng) { nk = vk[g + 1]; ni = vi[g + 1]; }
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t idx = g * 4u + k;
                    if (g < ng && idx >= skip && idx < skip + m) {
                        const uint32_t inf = word_of(cinf, k);
                        if ((inf & 1u) && !((inf >> 29) & 1u)) {
                            const uint32_t r = (inf >> 17) & 0xFFFu;
                            mask |= 1ull << r;
                            s_okey[r * 64u + lane] = word_of(ck, k); s_oeid[r * 64u + lane] = inf & BI_EID;
                        }
                    }
                }
            }
        }
        // ---- the events in order
        {
            uint4 nk = make_uint4(0, 0, 0, 0), ni = nk;
            if (ng) { nk = vk[0]; ni = vi[0]; }
            for (uint32_t g = 0; __ballot(g < ng) != 0ull; ++g) {
                const uint4 ck = nk, cinf = ni;
                if (g + 1 < ng) { nk = vk[g + 1]; ni = vi[g + 1]; }
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t idx = g * 4u + k;
                    if (g < ng && idx >= skip && idx < skip + m) {
                        const uint32_t inf = word_of(cinf, k), kw = word_of(ck, k);
                        const uint32_t r = (inf >> 17) & 0xFFFu, eid = inf & BI_EID;
                        if (inf & 1u) {                                    // clear the recorded bucket, whoever sits there (lz77.c:70-76)
                            if ((mask >> r) & 1ull) {
                                const uint32_t o = s_oeid[r * 64u + lane];
                                if (o != eid && !(o & 1u)) atomicOr(&slot_new[o >> 1], LZS_DEAD);     // a new entry removed early
                                mask &= ~(1ull << r);
                            }
                        } else {
                            const uint64_t above = ~(mask >> r);           // first fit, inside the cluster by the parking bound
                            const uint32_t fe = r + (uint32_t)__builtin_ctzll(above);
                            uint32_t found = LZS_EMPTY;
                            for (uint32_t b = r; b < fe; ++b)              // find(): the first occupant of [r, fe) with this word
                                if (s_okey[b * 64u + lane] == kw) { found = s_oeid[b * 64u + lane]; break; }
                            const uint32_t t = eid >> 1;
                            cand[t] = found == LZS_EMPTY ? LZS_NONE : ((found & 1u) ? base_old : base_new) + (found >> 1);
                            if (fe < CAPM) { mask |= 1ull << fe; s_okey[fe * 64u + lane] = kw; s_oeid[fe * 64u + lane] = eid; }
                            __hip_atomic_store(&slot_new[t], (kw & Tmask) + (fe - r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =============================================================================================
// stage 3: a wave per exported cluster (65 .. LZS_CAP events)
// =============================================================================================
template <uint32_t CAPB>
__global__ __launch_bounds__(64)
void k_lzs_big(LzP P, LzsScratch sc, uint32_t step, uint32_t cls)
{
    constexpr uint32_t BW = CAPB / 32u;
    __shared__ uint32_t s_okey[CAPB], s_oeid[CAPB];      // per slot of the cluster: the occupant's mixed word / event id
    __shared__ uint32_t s_bm[BW];                        // occupancy, one bit per slot
    const uint32_t lane = threadIdx.x;
    const uint32_t W = 1u << P.wbits, t0 = step * W, Tmask = (1u << P.tbits) - 1u;
    const uint32_t base_new = t0, base_old = t0 - (step ? W : 0u);
    const uint32_t count = sc.counters[128 + 8 * step + 1 + cls];
    // the long chains are the step's critical path: their waves go first whenever they can issue (the short-chain class
    // runs beside them with 32 waves per CU and saturates the scalar unit)
    if (CAPB > 1024u) __builtin_amdgcn_s_setprio(3);
    // clusters are handed out by a cursor (a static stride left waves with twice the average chain), FETCH at a time: one
    // returning atomic per cluster on one address cost more than the replays (91 k clusters per step: 34 -> 45 ms)
    uint32_t *cursor = &sc.counters[768 + 2 * step + (cls & 1u)];
    constexpr uint32_t FETCH = CAPB > 1024u ? 1u : 8u;
    uint32_t have = 0, next = 0;
    for (;;) {
        if (have == 0) {
            uint32_t c0 = 0;
            if (lane == 0) c0 = atomicAdd(cursor, FETCH);
            next = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
            have = FETCH;
        }
        const uint32_t ci = next++;
        --have;
        if (ci >= count) break;
        const uint64_t d = sc.big_desc[cls][ci];
        const uint32_t first = (uint32_t)d, m = (uint32_t)(d >> 32) & 0xFFFFu, lb = (uint32_t)(d >> 48);
        const uint32_t *key = sc.big_key + first, *info = sc.big_info + first;
        uint32_t *slot_new = sc.slot + (size_t)lb * sc.S + t0;
        uint32_t *cand = sc.cand + (size_t)lb * sc.S + t0;
        // A launch lasts as long as its longest chain, and a chain shares its SIMD's issue slots with up to seven other waves
        // (~75 wave-uniform instructions per event: 512 events took ~0.5 ms, the whole launch, where they need ~0.07 alone):
        // the longer the cluster, the higher the wave's priority while it replays it (s_setprio takes an immediate)
        if (CAPB <= 1024u) {
            if (m >= 384u) __builtin_amdgcn_s_setprio(3);
            else if (m >= 256u) __builtin_amdgcn_s_setprio(2);
            else if (m >= 128u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // the first 64 records serve the set-up pass and the first round of the replay: one load round for most clusters
        uint32_t n_key = lane < m ? key[lane] : 0u, n_inf = lane < m ? info[lane] : 0u;
        for (uint32_t idx = lane; idx < BW; idx += 64u) s_bm[idx] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t idx = lane; idx < m; idx += 64u) {                   // the table the step starts from
            uint32_t inf = n_inf, kw = n_key;
            if (idx >= 64u) { inf = info[idx]; kw = key[idx]; }
            if ((inf & 1u) && !((inf >> 29) & 1u)) {
                const uint32_t r = (inf >> 17) & 0xFFFu;
                s_oeid[r] = inf & BI_EID; s_okey[r] = kw; atomicOr(&s_bm[r >> 5], 1u << (r & 31u));
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i0 = 0; i0 < m; i0 += 64u) {
            const uint32_t my_key = n_key, my_inf = n_inf;
            if (i0 + 64u + lane < m) { n_key = key[i0 + 64u + lane]; n_inf = info[i0 + 64u + lane]; }    // in flight during these 64
            const uint32_t cnt = (m - i0) < 64u ? (m - i0) : 64u;
            // bit l set: event l is a find + insert (or lies past the end): runs of clears between them are one parallel step
            const uint64_t m_new = __ballot(lane >= cnt || !(my_inf & 1u));
            uint32_t l = 0;
            while (l < cnt) {
                const uint64_t rest = m_new >> l;
                const uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64u - l;
                if (run) {                                                 // clear the recorded buckets, whoever sits there (lz77.c:70-76)
                    if (lane >= l && lane < l + run) {
                        const uint32_t r = (my_inf >> 17) & 0xFFFu, bit = 1u << (r & 31u);
                        if (s_bm[r >> 5] & bit) {
                            const uint32_t o = s_oeid[r];
                            if (o != (my_inf & BI_EID) && !(o & 1u)) atomicOr(&slot_new[o >> 1], LZS_DEAD);    // a new entry removed early
                            atomicAnd(&s_bm[r >> 5], ~bit);
                        }
                    }
                    l += run;
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                const uint32_t inf = (uint32_t)__builtin_amdgcn_readlane((int)my_inf, (int)l), kw = (uint32_t)__builtin_amdgcn_readlane((int)my_key, (int)l);
                const uint32_t r = (inf >> 17) & 0xFFFu, eid = inf & BI_EID;
                // first fit: the occupancy words from r's on, one per lane (2048 buckets per look) — a frequent word's live
                // copies fill hundreds of buckets above its home, and walking them 64 at a time was the chain's cost.
                // The first 64 buckets' occupants are read in the same breath: the word's own copy usually sits right there.
                const uint32_t wbase = r >> 5;
                uint32_t wv = (wbase + lane < BW) ? s_bm[wbase + lane] : 0u;                  // past the array: nobody's buckets
                const uint32_t idx0 = r + lane;
                uint32_t v = idx0 < CAPB ? s_oeid[idx0] : 0u, k = idx0 < CAPB ? s_okey[idx0] : 0u;
                if (lane == 0) wv |= (1u << (r & 31u)) - 1u;                                   // buckets below the home do not count
                uint32_t fe;
                {
                    uint64_t bf = __ballot(wv != 0xFFFFFFFFu);
                    uint32_t wb = wbase;
                    while (!bf) {                                                              // (more than 2048 occupied buckets in a row)
                        wb += 64u;
                        wv = (wb + lane < BW) ? s_bm[wb + lane] : 0u;
                        bf = __ballot(wv != 0xFFFFFFFFu);
                    }
                    const uint32_t fl = (uint32_t)__builtin_ctzll(bf);
                    const uint32_t fw = ~(uint32_t)__builtin_amdgcn_readlane((int)wv, (int)fl);
                    fe = ((wb + fl) << 5) + (uint32_t)__builtin_ctz(fw);
                }
                // find(): the first occupant of [r, fe) that holds this word (every bucket in there is occupied)
                uint32_t found = LZS_EMPTY;
                for (uint32_t base = r;;) {
                    const uint64_t bmm = __ballot(base + lane < fe && k == kw);
                    if (bmm) { found = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_ctzll(bmm)); break; }
                    base += 64u;
                    if (base >= fe) break;
                    const uint32_t idx = base + lane;
                    v = idx < CAPB ? s_oeid[idx] : 0u; k = idx < CAPB ? s_okey[idx] : 0u;
                }
                if (lane == 0) {
                    const uint32_t t = eid >> 1;
                    cand[t] = found == LZS_EMPTY ? LZS_NONE : ((found & 1u) ? base_old : base_new) + (found >> 1);
                    if (fe < m) { s_okey[fe] = kw; s_oeid[fe] = eid; atomicOr(&s_bm[fe >> 5], 1u << (fe & 31u)); }     // fe < m by the parking bound
                    __hip_atomic_store(&slot_new[t], (kw & Tmask) + (fe - r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an early clear ORs LZS_DEAD into it later)
                }
                l += 1u;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =============================================================================================
// host side
// =============================================================================================
void lzs_view(const LzwScratch &ws, uint32_t nb, LzsScratch *sc)
{
    // the sliced finder lives in the arrays of lzw.hip's workspace (which it never uses at the same time)
    sc->key = ws.gid; sc->slot = ws.rd; sc->cand = ws.cand; sc->plist = ws.t_pos; sc->S = ws.S;
    sc->meta = reinterpret_cast<LzsMeta *>(ws.eA);                       // nb records at the start of the nb x S x 8 B of eA ...
    sc->pmap = reinterpret_cast<uint8_t *>(ws.eA) + mi_align_up((size_t)nb * sizeof(LzsMeta), 256);   // ... then nb part maps of 2 W <= 128 KiB (S >= 65 792: 8 S bytes per block)
    sc->work = reinterpret_cast<uint64_t *>(ws.eB);                      // nb x 64 items
    sc->counters = reinterpret_cast<uint32_t *>(sc->work + (size_t)nb * LZS_MAXPARTS);                 // four groups x LZS_CTR_WORDS ([64] of the first: flagged blocks)
    sc->flag_count = sc->counters + 64;
    sc->flag_list = sc->counters + 4 * LZS_CTR_WORDS + 64;               // behind the debug counters: one word per block
    sc->lb0 = 0;
    sc->dbg = getenv("MI_LZ_DEBUG") ? reinterpret_cast<uint64_t *>(sc->counters + 4 * LZS_CTR_WORDS) : nullptr;
    sc->big_key = ws.t_mix; sc->big_info = ws.slot_of;                   // nb x S words each: a step has at most nb x S events
    for (uint32_t q = 0; q < LZS_NCLS; ++q) sc->big_desc[q] = ws.clist[q];  // nb x S / 2 + 64 each (a cluster has >= 17 events)
}

// at most 64 steps per block (their work counters), event ids below 2^17; MI_LZW_SLICED=0 keeps lzw.hip's whole-block path (A/B, tests)
bool lzs_applicable(const LzP &P)
{
    const char *e = getenv("MI_LZW_SLICED");
    if (e && e[0] == '0') return false;
    const uint32_t W = 1u << P.wbits;
    // (coordinates relative to a part are sorted as 24-bit keys: a 2^24-bucket table plus the slots past its end would not fit)
    return !P.deflate && P.tbits <= 23u && (P.block + W - 1u) / W <= 64u;
}

// Every block of the batch, step by step.  A step is a chain part -> find -> wave replays whose last link waits for the longest
// cluster, and the next step needs every slot of this one: so the batch is cut into two groups of blocks that walk
// their steps independently on the context's streams — one group's serial chains run beside another's sorts.
// *flagged = blocks the sliced finder could not do (read back: one stream synchronisation per batch).
mi_status lzs_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                   const LzwScratch &ws, hipStream_t s, uint32_t *flagged, const uint32_t **flag_list)
{
    static_assert(sizeof(LzsMeta) + 2u * 65536u + 512u <= 65792u * 8u, "meta records and part maps live in eA: 8 S bytes per block, S >= 65 792");
    if (!lzs_applicable(P)) return MI_ERR_ARG;
    LzsScratch all;
    lzs_view(ws, nb, &all);
    const uint32_t W = 1u << P.wbits;
    const uint32_t nsteps = (P.block + W - 1u) / W;
    const uint32_t chunks = (P.block + 256u * 16u - 1u) / (256u * 16u);
    // group g walks its steps on st[g]; the few long chains of a step (clusters above 512 events) run beside the many short
    // ones on the group's second stream ax[g] (a launch lasts as long as its longest chain)
    const bool multi = ctx->side && ctx->parse && ctx->fb && !getenv("MI_LZS_SERIAL");       // MI_LZS_SERIAL=1: one stream (per-kernel times)
    hipStream_t st[2] = {s, multi ? ctx->side : s}, ax[2] = {multi ? ctx->parse : s, multi ? ctx->fb : s};
    const char *eg = getenv("MI_LZS_GROUPS");
    uint32_t G = eg ? (uint32_t)atoi(eg) : 2u;
    if (G < 1u || G > 2u) G = 2u;
    if (G > nb || !multi) G = 1;
    MI_HIP(ctx, hipMemsetAsync(all.counters, 0, (size_t)LZS_CTR_WORDS * 4 * 4 + 32 * 8, s));      // counters of four groups + the 32 debug counters (the flag list lies behind them)
    MI_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
    LzsScratch sg[2]; uint32_t lo[3];
    for (uint32_t g = 0; g <= G; ++g) lo[g] = (uint32_t)(((uint64_t)nb * g) / G);
    for (uint32_t g = 0; g < G; ++g) {
        if (g) MI_HIP(ctx, hipStreamWaitEvent(st[g], ctx->ev_fork, 0));
        LzsScratch &q = sg[g];
        q = all;
        const size_t o = (size_t)lo[g] * all.S;
        q.key += o; q.slot += o; q.cand += o; q.plist += o; q.big_key += o; q.big_info += o;
        q.pmap += (size_t)lo[g] * 2u * (1u << P.wbits);
        q.meta += lo[g]; q.work += (size_t)lo[g] * LZS_MAXPARTS; q.counters += (size_t)g * LZS_CTR_WORDS;
        for (uint32_t c = 0; c < LZS_NCLS; ++c) q.big_desc[c] += o / 2;
        q.flag_count = all.counters + 64;                 // one count and one list of flagged blocks for the whole batch
        q.flag_list = all.flag_list; q.lb0 = lo[g];
        const uint32_t nbg = lo[g + 1] - lo[g];
        mi_prof_scope p(ctx, "k_lzs_keys", st[g], (uint64_t)nbg * P.block);
        hipLaunchKernelGGL(k_lzs_keys, dim3(chunks, nbg), dim3(256), 0, st[g], d_in, n, P, q, block0 + lo[g]);
    }
    for (uint32_t k = 0; k < nsteps; ++k) {
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t nbg = lo[g + 1] - lo[g];
            const LzsScratch &q = sg[g];
            // the lane-per-cluster kernels pay when a step has enough clusters to fill them (MI_LZS_MID=0/1 forces it: A/B)
            const char *em = getenv("MI_LZS_MID");
            const uint32_t use_mid = LZS_LANE_MAX >= LZS_MID_MAX ? 0u      // (k_lzs_find replays clusters of up to LZS_LANE_MAX events itself)
                                     : em ? (uint32_t)(em[0] == '1') : (uint32_t)((uint64_t)nbg * W >= (6u << 20));
            // MI_LZS_PHASE=1: the groups walk their steps out of phase (group 1 starts its first partition when group 0 has finished
            // its first find).  The kernel timeline of round 4 (95 blocks of 1 MiB) shows the two groups in lockstep — both
            // partitions side by side, 2 x 47 workgroups on 256 CUs for 0.26 of a step's 1.23 ms, then both finds, then both
            // replays — but half a step apart they are no faster (4.18 -> 4.06 GB/s at 1 MiB, 6.79 -> 6.7 at 256 KiB): a group's
            // step is a chain of three kernel LATENCIES (partition 0.26, find 0.41, longest replay 0.53 ms) whoever runs beside it.
            static const bool phase = getenv("MI_LZS_PHASE") && getenv("MI_LZS_PHASE")[0] == '1';
            if (phase && G > 1 && k == 0 && g == 1) MI_HIP(ctx, hipStreamWaitEvent(st[1], ctx->ev_part[0], 0));
            { mi_prof_scope p(ctx, "k_lzs_part", st[g], (uint64_t)nbg * W);
              hipLaunchKernelGGL(k_lzs_part, dim3(nbg), dim3(1024), 0, st[g], n, P, q, block0 + lo[g], k); }
            { mi_prof_scope p(ctx, "k_lzs_find", st[g], (uint64_t)nbg * W);
              hipLaunchKernelGGL(k_lzs_find, dim3(nbg * LZS_KMAX), dim3(LZS_THREADS), 0, st[g], n, P, q, block0 + lo[g], k, use_mid); }
            if (phase && G > 1 && k == 0 && g == 0) MI_HIP(ctx, hipEventRecord(ctx->ev_part[0], st[0]));
            if (ax[g] != st[g]) { MI_HIP(ctx, hipEventRecord(ctx->ev_find[g], st[g])); MI_HIP(ctx, hipStreamWaitEvent(ax[g], ctx->ev_find[g], 0)); }
            { mi_prof_scope p(ctx, "k_lzs_big<4096>", ax[g], (uint64_t)nbg * W);
              hipLaunchKernelGGL(k_lzs_big<LZS_CAP>, dim3((unsigned)ctx->num_cu * 4u), dim3(64), 0, ax[g], P, q, k, 3u); }
            if (use_mid) {
                { mi_prof_scope p(ctx, "k_lzs_mid<64>", ax[g], (uint64_t)nbg * W);
                  hipLaunchKernelGGL(k_lzs_mid<64u>, dim3((unsigned)ctx->num_cu * 4u), dim3(64), 0, ax[g], P, q, k, 1u); }
                { mi_prof_scope p(ctx, "k_lzs_mid<32>", ax[g], (uint64_t)nbg * W);
                  hipLaunchKernelGGL(k_lzs_mid<32u>, dim3((unsigned)ctx->num_cu * 8u), dim3(64), 0, ax[g], P, q, k, 0u); }
            }
            { mi_prof_scope p(ctx, "k_lzs_big<512>", st[g], (uint64_t)nbg * W);
              hipLaunchKernelGGL(k_lzs_big<LZS_BIG_SMALL>, dim3((unsigned)ctx->num_cu * 32u), dim3(64), 0, st[g], P, q, k, 2u); }
            if (ax[g] != st[g]) { MI_HIP(ctx, hipEventRecord(ctx->ev_done[g], ax[g])); MI_HIP(ctx, hipStreamWaitEvent(st[g], ctx->ev_done[g], 0)); }
        }
    }
    for (uint32_t g = 1; g < G; ++g) { MI_HIP(ctx, hipEventRecord(ctx->ev_replay[g - 1], st[g])); MI_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_replay[g - 1], 0)); }
    const LzsScratch &sc = all;
    MI_HIP(ctx, hipGetLastError());
    uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
    MI_HIP(ctx, hipMemcpyAsync(h, sc.flag_count, 4, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    *flagged = *h;
    if (*flagged && flag_list) {                         // which blocks: the caller redoes only those
        const uint32_t k = *flagged < nb ? *flagged : nb;
        if ((size_t)(k + 1) * 4 <= ctx->h_pinned_bytes) {
            MI_HIP(ctx, hipMemcpyAsync(h + 1, sc.flag_list, (size_t)k * 4, hipMemcpyDeviceToHost, s));
            MI_HIP(ctx, hipStreamSynchronize(s));
            *flag_list = h + 1;
        } else *flag_list = nullptr;
    }
    if (sc.dbg) {                                        // development aid: phase shares of k_lzs_find on stderr
        uint64_t v[24];
        if (hipMemcpy(v, sc.dbg, sizeof v, hipMemcpyDeviceToHost) == hipSuccess && v[8]) {
            if (v[23]) {
                static const char *pn[6] = {"zero + count", "certificate + prefix", "cuts", "part sizes + list", "part of every event", "map out"};
                double pt = 0; for (int k = 0; k < 6; ++k) pt += (double)v[16 + k];
                fprintf(stderr, "k_lzs_part: %llu workgroups, %.0f cycles each\n", (unsigned long long)v[23], pt / v[23]);
                for (int k = 0; k < 6; ++k) fprintf(stderr, "   %-22s %5.1f %%  %8.0f cycles\n", pn[k], 100.0 * v[16 + k] / pt, (double)v[16 + k] / v[23]);
            }
            static const char *nm[8] = {"gather", "sort", "sweep", "place", "permute", "lane replay", "export", "out"};
            double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)v[k];
            fprintf(stderr, "k_lzs_find: %llu parts, %.0f events, %.1f clusters, %.2f wave clusters with %.0f events per part; %.0f cycles per part; flagged %u\n",
                    (unsigned long long)v[8], (double)v[9] / v[8], (double)v[10] / v[8], (double)v[11] / v[8], (double)v[12] / v[8], tot / v[8], *flagged);
            fprintf(stderr, "   exported events per part in clusters > 512: %.0f, > 1024: %.0f, > 2048: %.0f\n", (double)v[13] / v[8], (double)v[14] / v[8], (double)v[15] / v[8]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, "   %-12s %5.1f %%  %8.0f cycles/part\n", nm[k], 100.0 * v[k] / tot, (double)v[k] / v[8]);
        }
    }
    return MI_OK;
}
