// lz2_find.hip — stage 2 and 3 of the LDS-resident match finder.
//
//   k_lz2_find        one workgroup per part (<= LZ2_CAP_S = 2560 positions whose clusters lie inside the part, certified by
//                     stage 1; 51 KiB of LDS, three per CU.  k_lz2_find_wide: the same body for the rare parts of up to
//                     LZ2_CAP = 4096).  Everything happens in LDS: mixed words (home = mask, equality = equality), positions
//                     sorted by home (2-3 stable 8-bit radix passes), clusters / dense home slots / word ids / "mixed"
//                     bits from the parking sweep, (cluster, time) order by cursor placement.  Then
//                       - a cluster of ONE word is answered by a closed form (anchor chain): most clusters;
//                       - while a cluster has retired nothing, find() is the first occurrence of the word = the word
//                         id: quiet clusters are answered without any replay, and lz77 with a window
//                         that covers the block stops right after the sweep;
//                       - mixed clusters of 2..7 entries are replayed one lane each, their table in two registers;
//                       - larger mixed ones are exported on 16-byte boundaries, by size class.
//   k_lz2_mid_direct  lane per exported cluster of 8..127 entries (64 clusters per wave, LDS regions per lane),
//                     eight entries per load.
//   k_lz2_big         one WAVE per exported cluster of 512..1024 entries (and of what the row replay leaves): occupancy bitmap in
//                     registers, first fit by ballot + v_readlane, 6 bytes of LDS per entry; bound by scalar-instruction issue.
//   k_lz2_rows        FOUR exported clusters of 128..511 entries per wave, one per 16-lane row, 256..511 first (round 4).
//   k_lz2_scatter     lists -> by-position array (test hook and fallback boundary).
//
// Replaces the same reference functions as lz_find.hip (hash / insert_hash_table / find,
// algorithms/lz77/lz77.c:13-108 and algorithms/deflate/lz77.c:14-174).
#include "lz_common.h"
#include "lz2.h"
#include "lz_dom.h"
#include "lz_replay.h"
#include <stdlib.h>

#define RS_HEAD 0x8000u
#define RS_MASK 0x7FFFu

__device__ __forceinline__ bool bm2_test(const uint32_t *bm, uint32_t b) { return (bm[b >> 5] >> (b & 31u)) & 1u; }

// lane-per-cluster replay of a SMALL cluster [s, e) (< LZ2_BIG entries): plain word scan is enough
__device__ void replay_small(const uint16_t *pos, uint16_t *rs, const uint16_t *pid, uint16_t *occ, uint32_t *bm,
                             uint32_t s, uint32_t e, uint32_t W, uint32_t anom, uint32_t limit, uint16_t *cand_i)
{
    uint32_t ev = s;
    bool anom_pending = anom != ~0u;
    const bool plain = (anom == ~0u) && (limit == ~0u);                 // not the cluster that covers bucket 0 / T
    if (plain && (uint32_t)pos[e - 1] <= (uint32_t)pos[s] + W) {
        // a cluster whose entries all lie within one window never evicts: no replay at all, find() = first occurrence
        for (uint32_t i = s; i < e; ++i) { const uint32_t id = pid[i]; cand_i[i] = (id != pos[i]) ? (uint16_t)id : (uint16_t)LZ_NONE16; }
        return;
    }
    for (uint32_t i = s; i < e; ++i) {
        const uint32_t p = pos[i], rsv = rs[i], r = rsv & RS_MASK, id = pid[i];
        while (ev < i && (uint32_t)pos[ev] + W < p) {                   // FIFO retirement, lz77.c:70-76
            const uint32_t b = rs[ev] & RS_MASK;
            atomicAnd(&bm[b >> 5], ~(1u << (b & 31u)));
            ++ev;
        }
        if (anom_pending && p > W - 1u) { atomicAnd(&bm[anom >> 5], ~(1u << (anom & 31u))); anom_pending = false; }
        // the bitmap word and the occupant of the home slot are fetched together, then its id and position together
        uint32_t wi = r >> 5;
        const uint32_t w0 = bm[wi];
        const uint32_t o0 = occ[r];
        uint32_t res = LZ_NONE16;
        if (plain && ev == s) {                                         // nothing evicted yet: the first occurrence (see the sweep)
            if (id != p) res = id;
        } else if ((w0 >> (r & 31u)) & 1u) {
            const uint32_t id0 = pid[o0], pos0 = pos[o0];
            if (id0 == id) res = pos0;
            else {
                for (uint32_t b = r + 1;; ++b) {                         // rare: the home holds another word
                    if (b == limit && r < limit) break;
                    if (!bm2_test(bm, b)) break;
                    const uint32_t o = occ[b];
                    if (pid[o] == id) { res = pos[o]; break; }
                }
            }
        }
        cand_i[i] = (uint16_t)res;
        uint32_t wv = w0 | ((1u << (r & 31u)) - 1u);                     // first fit: word scan (inside the cluster by the parking bound)
        while (wv == 0xFFFFFFFFu) wv = bm[++wi];
        const uint32_t b = (wi << 5) + (uint32_t)__builtin_ctz(~wv);
        atomicOr(&bm[b >> 5], 1u << (b & 31u));
        occ[b] = (uint16_t)i;
        rs[i] = (uint16_t)((rsv & RS_HEAD) | b);
    }
}

// The same replay for a cluster of < 8 entries that does not cover bucket 0 / T, with its whole table in registers:
// 7 occupancy bits, the occupant of every slot and the slot of every entry as 4-bit fields.  LDS is only read for the
// entry records (and for an occupant's word id when a probe has to compare) and written for the result.
// F = uint32_t: up to 8 four-bit fields (clusters below 8 entries); F = uint64_t: up to 16 (below 16).
template <typename F>
__device__ __forceinline__ void replay_small_reg(const uint16_t *pos, const uint16_t *rs, const uint16_t *pid,
                                                 uint32_t s, uint32_t e, uint32_t W, uint16_t *cand_i)
{
    if ((uint32_t)pos[e - 1] <= (uint32_t)pos[s] + W) {
        // a cluster whose entries all lie within one window never evicts: no replay at all, find() = first occurrence
        for (uint32_t i = s; i < e; ++i) { const uint32_t id = pid[i]; cand_i[i] = (id != pos[i]) ? (uint16_t)id : (uint16_t)LZ_NONE16; }
        return;
    }
    const uint32_t n = e - s;
    uint32_t mask = 0;
    F ent = 0, slots = 0;
    uint32_t ev = 0, ev_pos = pos[s];
    for (uint32_t li = 0; li < n; ++li) {
        const uint32_t i = s + li;
        const uint32_t p = pos[i], rr = ((uint32_t)rs[i] & RS_MASK) - s, id = pid[i];
        while (ev < li && ev_pos + W < p) {                              // FIFO retirement, lz77.c:70-76
            mask &= ~(1u << ((uint32_t)(slots >> (4u * ev)) & 15u));
            ++ev; ev_pos = pos[s + ev];
        }
        uint32_t res = LZ_NONE16;
        if (ev == 0) {                                                  // nothing evicted yet: the first occurrence (see the sweep)
            if (id != p) res = id;
        } else if (id != p) {                                           // (a word's first occurrence in the block finds nothing, ever)
            for (uint32_t b = rr; (mask >> b) & 1u; ++b) {              // bits >= n are never set: the probe ends inside the cluster
                const uint32_t o = s + ((uint32_t)(ent >> (4u * b)) & 15u);
                if (pid[o] == id) { res = pos[o]; break; }
            }
        }
        cand_i[i] = (uint16_t)res;
        const uint32_t b = rr + (uint32_t)__builtin_ctz(~(mask >> rr));  // first fit (inside the cluster by the parking bound)
        mask |= 1u << b;
        ent = (ent & ~((F)15u << (4u * b))) | ((F)li << (4u * b));
        slots |= (F)b << (4u * li);
    }
}

// one part: `item` = block | part << 16 | entries << 24 | list start << 40 (lz2.h).  CAP = LDS capacity in entries.
template <uint32_t CAP>
__device__ __forceinline__ void lz2_find_part(const uint8_t *__restrict__ in, uint64_t n_total, const LzP &P, const Lz2Scratch &sc, uint64_t block0, const uint64_t item)
{
    // 16 bytes per entry + radix counters
    __shared__ uint16_t s_pos[CAP];                 // position by time index j
    __shared__ uint32_t s_word[CAP];                // mix32(word) by j; later e_pos / e_rs (replay order)
    __shared__ uint16_t s_j0[CAP], s_j1[CAP];   // sort ping-pong; later e_pid / (free)
    __shared__ __attribute__((aligned(16))) uint16_t s_gr[2 * CAP];   // s_g | s_r; during the home sort: the second pass's counters
    uint16_t *const s_g = s_gr;                          // cluster number by j; later occ
    uint16_t *const s_r = s_gr + CAP;                // dense home slot by j; later cand by replay index
    __shared__ uint16_t s_pid[CAP + 2];             // word id by j (position of the first occurrence); then s_gstart; last cand by j
    __shared__ uint32_t s_cnt[LZ2_NWAVES + 1][256];   // radix counters, [digit][wave + pad] (lz_common.h)
    __shared__ uint32_t s_bm[CAP / 32 + 2];
    __shared__ uint32_t s_mix[CAP / 32 + 2];        // bit g: cluster g holds more than one home or more than one word ("mixed")
    __shared__ int32_t  s_i32[18];
    __shared__ uint32_t s_zslot, s_zgid, s_nbigl, s_ngroups;
    // LDS diet: two of these workgroups share a CU, and whatever they leave (160 KiB - 2 x this kernel) is all that the
    // replay kernels of the previous batch can use beside them.  Arrays that are dead by then are reused:
    //   s_gstart (replay index of the head of every cluster) lives in s_pid, dead between the permutation and the final
    //            reordering of the results;
    //   s_big    ({s, e, global dst} of the clusters this part exports) and s_quiet (quiet clusters of >= LZ2_BIG
    //            entries) live in the radix counters, dead after the second sort.
    uint16_t *const s_gstart = s_pid;
    uint32_t *const s_big = &s_cnt[0][0];
    uint16_t *const s_quiet = reinterpret_cast<uint16_t *>(&s_cnt[0][0] + 3 * (CAP / LZ2_BIG));
    static_assert(3 * (CAP / LZ2_BIG) * 4 + 2 * (CAP / LZ2_BIG) * 2 <= sizeof(uint32_t) * LZ2_NWAVES * 256, "export lists must fit the radix counters");

    const int tid = threadIdx.x;
    const uint32_t lb = (uint32_t)item & 0xFFFFu, part = (uint32_t)(item >> 16) & 0xFFu;
    Lz2BlockMeta *mt = sc.meta + lb;
    long long tk = clock64();
    // stop_phase exists only in a measurement build (make EXTRA=-DMI_MEASURE, scripts/phase_pmc.sh): the shipped library never
    // leaves early, whatever the environment says (a stray MI_LZ_STOP_PHASE produced wrong streams with MI_OK)
#ifdef MI_MEASURE
#define LZ2_STOP(k) if ((k) < 7 && sc.stop_phase == (uint32_t)(k) + 1u) return
#else
#define LZ2_STOP(k) do { } while (0)
#endif
#define LZ2_TICK(k) do { if (sc.dbg && tid == 0) { long long t2 = clock64(); atomicAdd((unsigned long long *)&sc.dbg[k], (unsigned long long)(t2 - tk)); tk = t2; } \
                         LZ2_STOP(k); } while (0)
    // (only parts of blocks that did not fall back are listed; the list position and length ride in the item, so the list
    //  loads below do not wait for the block's meta record)
    const uint32_t m = (uint32_t)(item >> 24) & 0xFFFFu;
    if (m == 0) return;
    const uint32_t pstart = (uint32_t)(item >> 40), base = mt->base;
    const uint64_t off = (block0 + lb) * (uint64_t)P.block;
    const uint32_t nblk = mt->n;
    const uint8_t *src = in + off;
    const uint32_t T = 1u << P.tbits, Tmask = T - 1u, W = 1u << P.wbits;
    uint16_t *plist = sc.plist + (size_t)lb * LZ_MAX_BLOCK + pstart;       // written here (the parse reads position / candidate lists)
    (void)n_total;
    // keys of the home sort are relative to the part's first home; a part of a 2^20-bucket table spans < 2^16 homes, so two
    // 8-bit passes are enough (three otherwise).  For two passes the digits are COUNTED where they are in hand anyway: the first
    // pass's while the words are hashed, the second pass's while the first one scatters — no counting loops of their own.
    const uint32_t plo_ = mt->part_lo[part];
    const uint32_t phi_ = (part + 1 < mt->nparts) ? mt->part_lo[part + 1] : T;
    const bool three = (phi_ - plo_) > 65536u;
    uint32_t *const cntA = &s_cnt[0][0];
    uint32_t (*const s_cntB)[256] = reinterpret_cast<uint32_t (*)[256]>(s_gr);       // 9 KiB of the 16 (s_g / s_r are idle until the sweep)
    uint32_t *const cntB = &s_cntB[0][0];
    constexpr uint32_t RST = LZ2_NWAVES + 1;                                        // counter stride of radix_pass
    const uint32_t seg = radix_seg<LZ2_NWAVES>(m), seg_inv = (uint32_t)((0x100000000ull + seg - 1u) / seg);   // i / seg = umulhi(i, seg_inv) for i < 2^16
    for (uint32_t i = tid; i < 256u * RST; i += LZ2_THREADS) { cntA[i] = 0; cntB[i] = 0; }

    bool viol = false;
    // ---- this part's positions, in time order: picked out of the block's part map (one byte per position, written by the
    //      partition).  Thread t looks at positions [128 t, 128 t + 128) — eight 16-byte loads — counts the bytes that equal this
    //      part's number (an exact SWAR zero-byte test per dword), a prefix sum over the threads gives its place, and it writes
    //      its positions there in ascending order: a stable compaction without a single atomic.  (Round 3 read a list the
    //      partition had sorted: item -> list -> words was a chain of three dependent global round trips, 26 % of this kernel.)
    {
        constexpr uint32_t BPT = LZ_MAX_BLOCK / LZ2_THREADS;                // bytes of the map per thread
        static_assert(BPT % 16u == 0, "whole 16-byte loads");
        const uint4 *pm = reinterpret_cast<const uint4 *>(sc.partmap + (size_t)lb * LZ_MAX_BLOCK + (size_t)tid * BPT);
        static_assert(BPT == 128u, "the match bitmap below is four dwords per thread");
        const uint32_t pat = part * 0x01010101u;
        auto nib = [&](uint32_t w) -> uint32_t {                            // bit k set: byte k of w equals `part`
            const uint32_t x = w ^ pat;
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;      // bit 7 of every zero byte of x, exactly
            return (((z >> 7) * 0x00204081u) >> 21) & 15u;                  // bits 0, 8, 16, 24 -> 0..3 (the partial products never overlap)
        };
        // the loads first, all eight; what stays alive across the prefix sum is one match bit per position: four registers
        uint32_t bm[4] = {0u, 0u, 0u, 0u};
        {
            uint4 pv[BPT / 16u];
#pragma unroll
            for (uint32_t q = 0; q < BPT / 16u; ++q) pv[q] = pm[q];
#pragma unroll
            for (uint32_t q = 0; q < BPT / 16u; ++q)
                bm[q >> 1] |= (nib(pv[q].x) | (nib(pv[q].y) << 4) | (nib(pv[q].z) << 8) | (nib(pv[q].w) << 12)) << (16u * (q & 1u));
        }
        const uint32_t mine = (uint32_t)(__popc(bm[0]) + __popc(bm[1]) + __popc(bm[2]) + __popc(bm[3]));
        __shared__ uint32_t s_scanp[LZ2_NWAVES + 2];
        uint32_t total_m;
        uint32_t at = block_exclusive_scan<uint32_t>(mine, OpAddU32(), 0u, s_scanp, &total_m);
        (void)total_m;                                                      // == m: the partition counted the same bytes
        const uint32_t p0 = (uint32_t)tid * BPT;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t z = bm[k];
            while (z) {
                const uint32_t bpos = (uint32_t)__builtin_ctz(z);
                z &= z - 1u;
                if (at < CAP) s_pos[at] = (uint16_t)(p0 + 32u * k + bpos);
                ++at;
            }
        }
    }
    __syncthreads();
    // ---- gather: the words (4 unaligned bytes each from the block; bytes past the block end read as zero, the parity
    //      definition of the reference's over-read)
    {
        // all of a thread's words first, then the hashing: the loads are in flight together instead of one entry at a time
        constexpr uint32_t GCH = CAP / LZ2_THREADS;
        uint32_t gp[GCH], glo[GCH], ghi[GCH];
        const bool aligned = (((uintptr_t)src) & 3u) == 0;
#pragma unroll
        for (uint32_t c = 0; c < GCH; ++c) { const uint32_t j = tid + c * LZ2_THREADS; gp[c] = j < m ? (uint32_t)s_pos[j] : 0u; if (j < m) plist[j] = (uint16_t)gp[c]; }
#pragma unroll
        for (uint32_t c = 0; c < GCH; ++c) {
            const uint32_t j = tid + c * LZ2_THREADS, p = gp[c];
            glo[c] = 0; ghi[c] = 0;
            if (j < m) {
                if (p + 8 <= nblk && aligned) {
                    const uintptr_t a = (uintptr_t)(src + p) & ~(uintptr_t)3;
                    glo[c] = *reinterpret_cast<const uint32_t *>(a);
                    ghi[c] = *reinterpret_cast<const uint32_t *>(a + 4);
                } else {
                    // bytes past the block end read as zero (the parity definition of the reference's over-read)
                    uint32_t w = 0;
                    for (uint32_t k = 0; k < 4 && p + k < nblk; ++k) w |= (uint32_t)src[p + k] << (8 * k);
                    glo[c] = w;
                }
            }
        }
        __syncthreads();                                    // the counters are zero (the loads above are in flight across it)
#pragma unroll
        for (uint32_t c = 0; c < GCH; ++c) {
            const uint32_t j = tid + c * LZ2_THREADS, p = gp[c];
            if (j < m) {
                uint32_t w = glo[c];
                if (p + 8 <= nblk && aligned) {
                    const uint32_t sh = ((uint32_t)((uintptr_t)(src + p) & 3u)) * 8u;
                    if (sh) w = (glo[c] >> sh) | (ghi[c] << (32u - sh));
                }
                // every step of the reference hash is invertible (odd multipliers, rotations, xor-shifts), so the mixed
                // value identifies the word: keep it instead of the word — the home is a mask away, equality is equality
                const uint32_t mx = lz_mix32(w);
                s_word[j] = mx;
                atomicAdd(&cntA[(((((mx & Tmask) - base) & Tmask) - plo_) & 255u) * RST + __umulhi(j, seg_inv)], 1u);
            }
        }
    }
    for (uint32_t i = tid; i < CAP / 32 + 2; i += LZ2_THREADS) { s_bm[i] = 0; s_mix[i] = 0; }
    if (tid == 0) { s_zslot = ~0u; s_zgid = ~0u; s_nbigl = 0; }
    __syncthreads();
    auto homep = [&](uint32_t j) -> uint32_t { return ((s_word[j] & Tmask) - base) & Tmask; };
    // (a part's positions are in time order by construction — a compaction, no sort: the order checks start with the home sort)
    LZ2_TICK(0);

    // ---- sort time indices by home', stable
    const uint32_t arank = P.flags & (LZP_ARANK | LZP_BREAK);
    auto keyp = [&](uint32_t j) -> uint32_t { return homep(j) - plo_; };
    uint16_t *srt = s_j0;                                   // where the home order ends up
    if (three) {
        radix_pass<LZ2_NWAVES, 8, uint32_t>(m, s_cnt, [&](uint32_t i) { return i; },
            [&](uint32_t e) { return keyp(e) & 255u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank, nullptr, true,
            [&](uint32_t d, uint32_t e) { atomicAdd(&cntB[((keyp(e) >> 8) & 255u) * RST + __umulhi(d, seg_inv)], 1u); });
        for (uint32_t i = tid; i < 256u * RST; i += LZ2_THREADS) cntA[i] = 0;         // (the next pass opens with a barrier)
        radix_pass<LZ2_NWAVES, 8, uint32_t>(m, s_cntB, [&](uint32_t i) { return (uint32_t)s_j0[i]; },
            [&](uint32_t e) { return (keyp(e) >> 8) & 255u; }, [&](uint32_t d, uint32_t e) { s_j1[d] = (uint16_t)e; }, arank, nullptr, true,
            [&](uint32_t d, uint32_t e) { atomicAdd(&cntA[((keyp(e) >> 16) & 255u) * RST + __umulhi(d, seg_inv)], 1u); });
        radix_pass<LZ2_NWAVES, 8, uint32_t>(m, s_cnt, [&](uint32_t i) { return (uint32_t)s_j1[i]; },
            [&](uint32_t e) { return (keyp(e) >> 16) & 255u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank, nullptr, true);
    } else {
        radix_pass<LZ2_NWAVES, 8, uint32_t>(m, s_cnt, [&](uint32_t i) { return i; },
            [&](uint32_t e) { return keyp(e) & 255u; }, [&](uint32_t d, uint32_t e) { s_j1[d] = (uint16_t)e; }, arank, sc.dbg, true,
            [&](uint32_t d, uint32_t e) { atomicAdd(&cntB[((keyp(e) >> 8) & 255u) * RST + __umulhi(d, seg_inv)], 1u); });
        radix_pass<LZ2_NWAVES, 8, uint32_t>(m, s_cntB, [&](uint32_t i) { return (uint32_t)s_j1[i]; },
            [&](uint32_t e) { return (keyp(e) >> 8) & 255u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank, sc.dbg, true);
    }
    (void)srt;

    LZ2_TICK(1);
    // ---- parking sweep over the sorted order: CH consecutive entries per thread
    constexpr uint32_t CH = CAP / LZ2_THREADS;
    const uint32_t k0 = tid * CH, k1 = (k0 + CH < m) ? k0 + CH : m;
    {
        // the thread's CH entries live in registers for the three passes below (time index, mixed word, home')
        uint32_t rj[CH], rw[CH]; int32_t rh[CH];
        int32_t mx = INT32_MIN;
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t k = k0 + c;
            rj[c] = 0; rw[c] = 0; rh[c] = 0;
            if (k < k1) {
                rj[c] = s_j0[k]; rw[c] = s_word[rj[c]]; rh[c] = (int32_t)(((rw[c] & Tmask) - base) & Tmask);
                const int32_t g = rh[c] - (int32_t)k;
                mx = g > mx ? g : mx;
            }
        }
        int32_t gmax_total;
        const int32_t premax = block_exclusive_scan<int32_t>(mx, OpMaxI32(), INT32_MIN, s_i32, &gmax_total);
        int32_t prev_h0 = 0; uint32_t prev_j = 0;
        if (k0 > 0 && k0 < m) { prev_j = s_j0[k0 - 1]; prev_h0 = (int32_t)homep(prev_j); }
        uint32_t nheads = 0; int32_t lasthead = -1, lastrun = -1;
        {
            int32_t run = premax, prev_h = prev_h0;
#pragma unroll
            for (uint32_t c = 0; c < CH; ++c) {
                const uint32_t k = k0 + c;
                if (k < k1) {
                    const int32_t h = rh[c], g = h - (int32_t)k;
                    const bool head = (k == 0) || (g >= run);
                    run = g > run ? g : run;
                    if (head) { ++nheads; lasthead = (int32_t)k; }
                    if (k == 0 || h != prev_h) lastrun = (int32_t)k;
                    prev_h = h;
                }
            }
        }
        // one scan for three carries: heads so far (sum), last head index + 1 (max), last home-run start + 1 (max)
        struct OpHeads {
            __device__ uint64_t operator()(uint64_t a, uint64_t b) const {
                const uint64_t s0 = (a & 0xFFFFu) + (b & 0xFFFFu);
                const uint64_t a1 = (a >> 16) & 0xFFFFu, b1 = (b >> 16) & 0xFFFFu, a2 = (a >> 32) & 0xFFFFu, b2 = (b >> 32) & 0xFFFFu;
                return s0 | ((a1 > b1 ? a1 : b1) << 16) | ((a2 > b2 ? a2 : b2) << 32);
            }
        };
        __shared__ uint64_t s_u64[18];
        uint64_t tot3;
        const uint64_t pre3 = block_exclusive_scan<uint64_t>((uint64_t)nheads | ((uint64_t)(lasthead + 1) << 16) | ((uint64_t)(lastrun + 1) << 32),
                                                            OpHeads(), 0ull, s_u64, &tot3);
        const uint32_t gid_base = (uint32_t)(pre3 & 0xFFFFu), total_heads = (uint32_t)(tot3 & 0xFFFFu);
        const int32_t gs_carry = (int32_t)((pre3 >> 16) & 0xFFFFu) - 1, hs_carry = (int32_t)((pre3 >> 32) & 0xFFFFu) - 1;
        if (tid == 0) s_ngroups = total_heads;
        int32_t run = premax, prev_h = prev_h0;
        uint32_t cur_gs = 0, cur_gid = gid_base; int32_t cur_base = 0;
        uint32_t cur_hs = 0, hs_word = 0, hs_j = 0;
        if (k0 < m) {
            if (gs_carry >= 0) { cur_gs = (uint32_t)gs_carry; cur_base = (int32_t)homep(s_j0[cur_gs]); cur_gid = gid_base - 1u; }
            if (hs_carry >= 0) { cur_hs = (uint32_t)hs_carry; hs_j = s_j0[cur_hs]; hs_word = s_word[hs_j]; }
        }
        uint32_t seen = 0;
#pragma unroll
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t k = k0 + c;
            if (k < k1) {
                const uint32_t j = rj[c];
                const int32_t h = rh[c], g = h - (int32_t)k;
                const bool head = (k == 0) || (g >= run);
                run = g > run ? g : run;
                const uint32_t w = rw[c];
                // ... and the home sort must have left (home, time) ascending: the check of every pass at once
                if (k > 0 && (h < prev_h || (h == prev_h && j < prev_j))) viol = true;
                prev_j = j;
                if (head) {
                    cur_gs = k; cur_base = h; cur_gid = gid_base + seen; ++seen;
                    // cursor of the cluster for the one-pass placement below: its first index in replay order (a cluster
                    // keeps its index range in the home order and in the (cluster, time) order)
                    reinterpret_cast<uint16_t *>(&s_cnt[0][0])[cur_gid] = (uint16_t)k;
                }
                uint32_t id;
                if (k == 0 || h != prev_h) { cur_hs = k; hs_word = w; hs_j = j; id = j; }
                else if (w == hs_word) id = hs_j;
                else {                                      // two different words share a home bucket: rare
                    id = j;
                    for (uint32_t kk = cur_hs + 1; kk < k; ++kk) { const uint32_t j2 = s_j0[kk]; if (s_word[j2] == w) { id = j2; break; } }
                    atomicOr(&s_mix[cur_gid >> 5], 1u << (cur_gid & 31u));
                }
                prev_h = h;
                s_g[j] = (uint16_t)cur_gid;
                s_r[j] = (uint16_t)(cur_gs + (uint32_t)(h - cur_base));
                // a cluster whose entries do not all share one home needs its (home, time) order re-sorted by time below;
                // the others (nearly all) are in time order as they stand.  s_bm is free until the replay.
                if (h != cur_base) {
                    if (arank & LZP_ARANK) atomicOr(&s_bm[cur_gid >> 5], 1u << (cur_gid & 31u));
                    atomicOr(&s_mix[cur_gid >> 5], 1u << (cur_gid & 31u));
                }
                // the word's identity is the POSITION of its first occurrence in the block (the home order is stable in
                // time).  While nothing of a cluster has been evicted that position is also what find() returns: the first
                // copy sits at the lowest slot of the word and every slot between the home and it stays occupied.
                s_pid[j] = s_pos[id];
            }
        }
    }
    if (viol) { lz_order_violation(P); viol = false; }
    __syncthreads();
    if (!P.deflate && W >= nblk) {
        // lz77 flavour with a window that covers the block (the 64 KiB-window build): nothing is ever retired, find() never
        // stops at the table end, and the spurious clear of bucket 0 happens at the last insertion — so find() is the
        // first occurrence of the word for EVERY position (2.3 above): no clusters, no second sort, no replay.
        uint16_t *cout = sc.cand + (size_t)lb * LZ_MAX_BLOCK + pstart;
        for (uint32_t j = tid; j < m; j += LZ2_THREADS) { const uint32_t id = s_pid[j]; cout[j] = (id != s_pos[j]) ? (uint16_t)id : (uint16_t)LZ_NONE16; }
        return;
    }
    // bucket 0 (= bucket T on deflate's ring) sits at home' Z: the cluster that covers it gets the one-time
    // spurious clear and, for deflate, the point where find() stops instead of wrapping
    if (tid == 0) {
        const uint32_t Z = (0u - base) & Tmask;
        const uint32_t plo = mt->part_lo[part];
        const uint32_t phi = (part + 1 < mt->nparts) ? mt->part_lo[part + 1] : T;
        if (Z >= plo && (Z < phi || phi == T)) {
            // last sorted entry with home' <= Z
            int32_t lo = -1, hi = (int32_t)m - 1;
            while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (homep(s_j0[mid]) <= Z) lo = mid; else hi = mid - 1; }
            if (lo >= 0) {
                const uint32_t j = s_j0[lo], gid = s_g[j];
                const uint32_t slot = (uint32_t)s_r[j] + (Z - homep(j));
                // last sorted index of that cluster (cluster numbers are non-decreasing along the order)
                int32_t a = lo, b = (int32_t)m - 1;
                while (a < b) { const int32_t mid = (a + b + 1) >> 1; if (s_g[s_j0[mid]] == gid) a = mid; else b = mid - 1; }
                if (slot <= (uint32_t)a) { s_zslot = slot; s_zgid = gid; }
            }
        }
    }
    __syncthreads();

    LZ2_TICK(2);
    // ---- sort time indices by cluster number (they start in time order): identity -> j0 -> j1
    //      (cluster numbers are < CAP <= 4096: two 6-bit passes, fewer ballots and a shorter offset scan than 8 + 8)
    static_assert(CAP <= 8192, "cluster numbers must fit 13 bits");
    constexpr int GB2 = CAP > 4096 ? 7 : 6;          // bits of the second cluster-number pass
    static_assert(2 * CAP <= sizeof(uint32_t) * (LZ2_NWAVES + 1) * 256, "the cluster cursors live in the radix counters");
    if (arank & LZP_ARANK) {
        // ONE pass, no counting: the sweep left every cluster's first replay index as a 16-bit cursor; an entry's place is its
        // cluster's cursor, post-incremented — in TIME order.  Time order needs every cursor to be advanced by one wave only
        // (a wave's LDS instructions execute in order, and the lanes of one returning add are served in lane order: the
        // property radix_pass uses), so wave w takes the clusters whose cursor PAIR (two 16-bit cursors share a dword, the
        // unit of an LDS atomic) has number = w modulo the wave count, and every wave walks the whole time-ordered list.
        // Replaces two 6-bit radix passes (count, scan, scatter each): 20 k -> ~8 k cycles per part.
        // Only clusters with MORE THAN ONE HOME need it (a cluster of one home is in time order already: the home sort is stable);
        // their entries are compacted first, time order kept, so the eight waves walk a short list instead of the whole part.
        uint32_t *cur32 = &s_cnt[0][0];
        const uint32_t wv = (uint32_t)tid >> 6, ln = (uint32_t)tid & 63u;
        auto multi = [&](uint32_t g) -> bool { return (s_bm[g >> 5] >> (g & 31u)) & 1u; };
        for (uint32_t k = k0; k < k1; ++k) { const uint32_t j = s_j0[k]; if (!multi(s_g[j])) s_j1[k] = (uint16_t)j; }
        __syncthreads();                                     // (s_j0 is dead now: it takes the compacted list)
        uint32_t nm;
        {
            uint32_t mine[CH], cnt = 0;
#pragma unroll
            for (uint32_t c = 0; c < CH; ++c) { const uint32_t j = k0 + c; mine[c] = (j < m && multi(s_g[j])) ? 1u : 0u; cnt += mine[c]; }
            uint32_t at = block_exclusive_scan<uint32_t>(cnt, OpAddU32(), 0u, reinterpret_cast<uint32_t *>(s_i32), &nm);
#pragma unroll
            for (uint32_t c = 0; c < CH; ++c) if (mine[c]) s_j0[at++] = (uint16_t)(k0 + c);
        }
        __syncthreads();
        for (uint32_t x0 = 0; x0 < nm; x0 += 256u) {
            uint32_t jj[4], gg[4], old[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) { const uint32_t x = x0 + 64u * u + ln; jj[u] = x < nm ? (uint32_t)s_j0[x] : 0u; gg[u] = x < nm ? (uint32_t)s_g[jj[u]] : 0xFFFFFFFFu; }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                old[u] = 0;
                if (gg[u] != 0xFFFFFFFFu && ((gg[u] >> 1) % (uint32_t)LZ2_NWAVES) == wv) old[u] = atomicAdd(&cur32[gg[u] >> 1], 1u << (16u * (gg[u] & 1u)));
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (gg[u] != 0xFFFFFFFFu && ((gg[u] >> 1) % (uint32_t)LZ2_NWAVES) == wv) s_j1[(old[u] >> (16u * (gg[u] & 1u))) & 0xFFFFu] = (uint16_t)jj[u];
        }
        __syncthreads();
        for (uint32_t i = tid; i < CAP / 32 + 2; i += LZ2_THREADS) s_bm[i] = 0;     // back to the replay's empty bitmap (barriers follow)
    } else {
        radix_pass<LZ2_NWAVES, 6, uint32_t>(m, s_cnt, [&](uint32_t i) { return i; },
            [&](uint32_t e) { return (uint32_t)s_g[e] & 63u; }, [&](uint32_t d, uint32_t e) { s_j0[d] = (uint16_t)e; }, arank);
        radix_pass<LZ2_NWAVES, GB2, uint32_t>(m, s_cnt, [&](uint32_t i) { return (uint32_t)s_j0[i]; },
            [&](uint32_t e) { return (uint32_t)s_g[e] >> 6; }, [&](uint32_t d, uint32_t e) { s_j1[d] = (uint16_t)e; }, arank);
    }

    LZ2_TICK(3);
    // ---- permute into replay order (cluster, time).  e_pos / e_rs overlay the dead word array.
    uint16_t *e_pos = reinterpret_cast<uint16_t *>(s_word);
    uint16_t *e_rs = e_pos + CAP;
    uint16_t *e_pid = s_j0;
    uint32_t regs[CH][3];
    for (uint32_t c = 0; c < CH; ++c) {
        const uint32_t i = tid + c * (uint32_t)LZ2_THREADS;
        if (i < m) {
            const uint32_t j = s_j1[i];
            const bool head = (i == 0) || (s_g[j] != s_g[s_j1[i - 1]]);
            if (i > 0) {                                       // the (cluster, time) order, however it was made
                const uint32_t jp = s_j1[i - 1], gp = s_g[jp], gj = s_g[j];
                if (gp > gj || (gp == gj && jp > j)) viol = true;
            }
            regs[c][0] = s_pos[j];
            regs[c][1] = (uint32_t)s_r[j] | (head ? RS_HEAD : 0u);
            regs[c][2] = (uint32_t)s_pid[j] | ((uint32_t)s_g[j] << 16);
        }
    }
    if (viol) lz_order_violation(P);
    __syncthreads();
    uint16_t *occ = s_g;                    // cluster numbers are dead from here on, except the one remembered in s_zgid
    uint16_t *cand_i = s_r;
    __shared__ uint16_t s_zhead;            // replay index of the head of the cluster that covers bucket 0
    if (tid == 0) { s_zhead = 0xFFFF; s_gstart[s_ngroups] = (uint16_t)m; }
    __syncthreads();
    for (uint32_t c = 0; c < CH; ++c) {
        const uint32_t i = tid + c * (uint32_t)LZ2_THREADS;
        if (i < m) {
            e_pos[i] = (uint16_t)regs[c][0]; e_rs[i] = (uint16_t)regs[c][1]; e_pid[i] = (uint16_t)regs[c][2];
            if (regs[c][1] & RS_HEAD) {
                s_gstart[regs[c][2] >> 16] = (uint16_t)i;
                if ((regs[c][2] >> 16) == s_zgid) s_zhead = (uint16_t)i;
            }
        }
    }
    __syncthreads();

    LZ2_TICK(4);
    __shared__ uint32_t s_cls[LZ2_NCLASS], s_clsbase[LZ2_NCLASS], s_ent, s_entbase;
    uint32_t my_rank[((CAP / LZ2_BIG) + LZ2_THREADS - 1) / LZ2_THREADS], my_dst[((CAP / LZ2_BIG) + LZ2_THREADS - 1) / LZ2_THREADS];
    uint32_t nbig = 0;
    // ---- replay.  Clusters below LZ2_BIG entries: one lane each, lanes sorted by cluster size so that the
    //      64 lanes of a wave run the same number of steps.  Larger clusters are exported by size class.
    {
        // cluster heads -> compact list (order irrelevant), sizes from the next head
        __shared__ uint32_t s_ncl, s_bin[LZ2_BIG + 1], s_nquiet;
        if (tid == 0) { s_ncl = 0; s_nquiet = 0; s_ent = 0; }
        if (tid < (int)LZ2_NCLASS) s_cls[tid] = 0;
        if (tid <= (int)LZ2_BIG) s_bin[tid] = 0;
        __syncthreads();
        uint16_t *c_start = s_j1 + 0;                        // (s_j1 is still needed: cand back to time order) -> use s_pos, dead now
        c_start = s_pos;                                     // [<= m/2] start of each lane-replayed cluster, bucketed by size
        uint32_t my_s[CH], my_n[CH];
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t i = tid + c * (uint32_t)LZ2_THREADS;
            my_n[c] = 0; my_s[c] = i;
            if (i < m && (e_rs[i] & RS_HEAD)) {
                const uint32_t g_ = regs[c][2] >> 16;
                const uint32_t e = s_gstart[g_ + 1];                      // clusters are contiguous and in cluster-number order
                const uint32_t size = e - i;
                my_n[c] = size;
                // a cluster whose entries all lie within one window never evicts: no replay, find() = first occurrence
                if (size == 1) cand_i[i] = LZ_NONE16;
                else if (!((s_mix[g_ >> 5] >> (g_ & 31u)) & 1u) && i != s_zhead) my_n[c] = 0;     // ONE word: the closed form below, no replay
                else if (size < LZ2_BIG) atomicAdd(&s_bin[size], 1u);          // (replay_small notices a quiet cluster itself)
                else if ((uint32_t)e_pos[e - 1] <= (uint32_t)e_pos[i] + W && i != s_zhead) { const uint32_t q = atomicAdd(&s_nquiet, 1u); s_quiet[2 * q] = (uint16_t)i; s_quiet[2 * q + 1] = (uint16_t)e; }
                else {
                    const uint32_t q = atomicAdd(&s_nbigl, 1u);
                    if (q < (CAP / LZ2_BIG)) { s_big[3 * q] = i; s_big[3 * q + 1] = e; }
                }
            }
        }
        __syncthreads();
        LZ2_TICK(11);
        if (tid == 0) {                                      // bucket offsets, largest size first
            uint32_t run = 0;
            for (int sz = (int)LZ2_BIG - 1; sz >= 2; --sz) { const uint32_t t = s_bin[sz]; s_bin[sz] = run; run += t; }
            s_ncl = run;
        }
        __syncthreads();
        for (uint32_t c = 0; c < CH; ++c)
            if (my_n[c] >= 2 && my_n[c] < LZ2_BIG) c_start[atomicAdd(&s_bin[my_n[c]], 1u)] = (uint16_t)my_s[c];
        __syncthreads();
        LZ2_TICK(12);
        // reserve the output space of the clusters this part exports NOW — ONE global atomic per part and class (the class
        // counters are shared by every workgroup of the batch), local ranks first in LDS — so that the round trips of
        // those atomics pass while the lanes replay
        nbig = s_nbigl < (CAP / LZ2_BIG) ? s_nbigl : (CAP / LZ2_BIG);          // CAP / LZ2_BIG clusters at most
        {
            uint32_t it = 0;
            for (uint32_t q = tid; q < nbig; q += LZ2_THREADS, ++it) {
                const uint32_t cnt = s_big[3 * q + 1] - s_big[3 * q];
                const uint32_t cls = lz2_class_of(cnt, sc.wave_min);
                my_rank[it] = atomicAdd(&s_cls[cls], 1u);
                my_dst[it] = atomicAdd(&s_ent, LZ2_ALIGN8(cnt));        // every cluster starts on an 8-entry boundary
            }
        }
        __syncthreads();
        // The returns of these global atomics (every part of the batch adds to the same eight class counters: ~12 k cycles
        // under that contention, measured) are only needed by the export below: they stay in a register until the lane
        // replay is through, so the wave that issues them does not stall in front of its share of the replay.
        uint32_t pending_base = 0;
        const bool want_cls = tid < (int)LZ2_NCLASS && s_cls[tid], want_ent = tid == 32 && s_ent;
        // (ONE returning atomic instruction with a per-lane address: on a uniform address the compiler's wave-aggregated
        // form reads the result back with v_readfirstlane at once — wave 0 then stood ~13 k cycles in front of its replay)
        if (want_cls || want_ent) {
            uint32_t *ap = want_ent ? &mt->nbig_entries : &sc.big_count[tid & (int)(LZ2_NCLASS - 1u)];
            pending_base = atomicAdd(ap, want_ent ? s_ent : s_cls[tid & (int)(LZ2_NCLASS - 1u)]);
        }
        if (want_ent) atomicAdd(&mt->nbig, nbig);
        for (uint32_t q = tid >> 6; q < s_nquiet; q += LZ2_NWAVES)
            for (uint32_t k = (uint32_t)s_quiet[2 * q] + (tid & 63u); k < s_quiet[2 * q + 1]; k += 64) {
                const uint32_t id = e_pid[k];
                cand_i[k] = (id != e_pos[k]) ? (uint16_t)id : (uint16_t)LZ_NONE16;
            }
        const uint32_t ncl = s_ncl;
        LZ2_TICK(13);
        if (sc.dbg && tid == 0) { atomicAdd((unsigned long long *)&sc.dbg[23], (unsigned long long)ncl); atomicAdd((unsigned long long *)&sc.dbg[24], (unsigned long long)nbig);
                                  atomicAdd((unsigned long long *)&sc.dbg[25], (unsigned long long)s_ent); atomicAdd((unsigned long long *)&sc.dbg[26], (unsigned long long)s_nquiet);
                                  atomicAdd((unsigned long long *)&sc.dbg[27], (unsigned long long)m); atomicAdd((unsigned long long *)&sc.dbg[28], (unsigned long long)s_ngroups); }
        // ---- clusters of ONE word (one home, every entry the same word: most clusters of a text) have a closed form.  find()
        //      for that word only ever looks at the home slot: it holds the "anchor", the copy that was inserted when the slot
        //      was free — the first entry, then, once the anchor has been retired (lz77.c:70-76: its position + W < p), the
        //      next entry, which itself finds the slot empty (nothing) and takes it.  a(0) = first entry, a(k+1) = first entry
        //      after a(k) + W: at most block / W anchors, each a binary search in the cluster's time-ordered positions.  Every
        //      entry on its own, no table, no order: neither the lane replay nor an export (k_lz_emulate_giant uses the same
        //      form for giant clusters).  The cluster that covers bucket 0 / T keeps the general path.
        for (uint32_t c = 0; c < CH; ++c) {
            const uint32_t i = tid + c * (uint32_t)LZ2_THREADS;
            if (i >= m) continue;
            const uint32_t g_ = regs[c][2] >> 16;
            if ((s_mix[g_ >> 5] >> (g_ & 31u)) & 1u) continue;
            const uint32_t s = s_gstart[g_];
            if (s == s_zhead || s_gstart[g_ + 1] - s < 2u) continue;
            const uint32_t p = regs[c][0];
            uint32_t a = e_pos[s], lo = s + 1u;
            while (a + W < p) {                                           // (e_pos[i] = p is beyond a + W: the search ends at i at the latest)
                uint32_t hi = i;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)e_pos[mid] > a + W) hi = mid; else lo = mid + 1u; }
                a = e_pos[lo]; ++lo;
            }
            cand_i[i] = (a == p) ? (uint16_t)LZ_NONE16 : (uint16_t)a;
        }
        for (uint32_t q = tid; q < ncl; q += LZ2_THREADS) {
            const uint32_t s = c_start[q];
            uint32_t e = s + 1;
            while (e < m && !(e_rs[e] & RS_HEAD)) ++e;
            const bool zc = (s == s_zhead);
            if (zc) replay_small(e_pos, e_rs, e_pid, occ, s_bm, s, e, W, s_zslot, P.deflate ? s_zslot : ~0u, cand_i);
            else if (LZ2_BIG > 8u && e - s >= 8u) replay_small_reg<uint64_t>(e_pos, e_rs, e_pid, s, e, W, cand_i);
            else replay_small_reg<uint32_t>(e_pos, e_rs, e_pid, s, e, W, cand_i);
        }
        if (want_cls) s_clsbase[tid] = pending_base;
        if (want_ent) s_entbase = pending_base;
    }
    __syncthreads();
    LZ2_TICK(5);
    // ---- export the larger clusters (cooperatively, coalesced); their space was reserved before the lane replay
    {
        uint32_t it = 0;
        for (uint32_t q = tid; q < nbig; q += LZ2_THREADS, ++it) {
            const uint32_t s = s_big[3 * q], e = s_big[3 * q + 1], cnt = e - s;
            const uint32_t cls = lz2_class_of(cnt, sc.wave_min);
            const uint32_t dst = s_entbase + my_dst[it];
            s_big[3 * q + 2] = dst;
            Lz2BigDesc d;
            d.block = lb; d.start = dst; d.count = cnt;
            const bool zc = (s == s_zhead);
            d.anom = zc ? s_zslot - s : ~0u;
            d.limit = (zc && P.deflate) ? s_zslot - s : ~0u;
            d.pad[0] = pstart; d.pad[1] = d.pad[2] = 0;
            sc.desc[cls][s_clsbase[cls] + my_rank[it]] = d;
        }
    }
    __syncthreads();
    for (uint32_t b = tid >> 6; b < nbig; b += LZ2_NWAVES) {          // a wave per exported cluster
        const uint32_t s = s_big[3 * b], e = s_big[3 * b + 1], dst = s_big[3 * b + 2];
        uint16_t *bp = sc.bigpos + (size_t)lb * LZ2_BIG_STRIDE + dst;
        uint16_t *br = sc.bigrs + (size_t)lb * LZ2_BIG_STRIDE + dst;
        uint16_t *bi = sc.bigpid + (size_t)lb * LZ2_BIG_STRIDE + dst;
        for (uint32_t i = s + (tid & 63); i < e; i += 64) {
            cand_i[i] = e_pos[i];                                // pending (lz2.h)
            bp[i - s] = e_pos[i];
            br[i - s] = (uint16_t)((e_rs[i] & RS_MASK) - s);    // home slot relative to the cluster
            bi[i - s] = e_pid[i];
        }
        uint16_t *bc = sc.bigcand + (size_t)lb * LZ2_BIG_STRIDE + dst;
        for (uint32_t k = (e - s) + (tid & 63); k < LZ2_ALIGN8(e - s); k += 64) { bp[k] = 0; bc[k] = 0; }    // pads: skipped by every consumer
    }
    // ---- cand back to time order and out (coalesced)
    uint16_t *cand_j = s_pid;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += LZ2_THREADS) cand_j[s_j1[i]] = cand_i[i];
    __syncthreads();
    uint16_t *cout = sc.cand + (size_t)lb * LZ_MAX_BLOCK + pstart;
    for (uint32_t j = tid; j < m; j += LZ2_THREADS) cout[j] = cand_j[j];
    LZ2_TICK(6);
    if (sc.dbg && tid == 0) atomicAdd((unsigned long long *)&sc.dbg[15], 1ull);
}


// The grid covers the worst case (32 parts per block) and the listed parts come first; a surplus workgroup reads the
// count and leaves.  Measured (round 2, same box, 10^9 bytes): a persistent grid of 2 workgroups per CU pulling parts
// off a cursor made this kernel 30.3 -> 51 ms per GB on its own — the loop keeps every scratch pointer live across
// the whole body: 84 -> 176 VGPRs, one workgroup per CU instead of two (forced back to 128 VGPRs it spills) — and the
// hardware dispatcher already IS a dynamic scheduler: ~13 k empty workgroups per batch cost nothing measurable.
// XCD-aware order: workgroup ids go round the 8 XCDs (each with its own L2), the work list holds a block's parts one
// after the other — so give every XCD a CONTIGUOUS eighth of the list: the ~26 parts of a block then gather their
// words from one L2 instead of pulling the block's 64 KiB into all eight (HBM fetch of this kernel: profiles/).
__global__ __launch_bounds__(LZ2_THREADS)
void k_lz2_find(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, Lz2Scratch sc, uint64_t block0)
{
    // the grid is a multiple of 8 (lz2_stage_find) and takes the first gridDim.x listed parts; what a batch lists beyond it
    // (more than LZ2_GRID_PARTS parts per block on average: not seen on any corpus) is k_lz2_find_wide's
    const uint32_t listed = *sc.work_count, nwork = listed < gridDim.x ? listed : gridDim.x;
    const uint32_t cpx = (nwork + 7u) >> 3, item_idx = (blockIdx.x & 7u) * cpx + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= cpx || item_idx >= nwork) return;
    lz2_find_part<LZ2_CAP_S>(in, n_total, P, sc, block0, sc.work[item_idx]);
}

// the parts above LZ2_CAP_S entries (rare: listed from the end of the work array backwards): a small grid that loops
__global__ __launch_bounds__(LZ2_THREADS)
void k_lz2_find_wide(const uint8_t *__restrict__ in, uint64_t n_total, LzP P, Lz2Scratch sc, uint64_t block0, uint32_t covered)
{
    // `covered` = what k_lz2_find's grid took from the front list; the rest of that list is done here too (a part of <= LZ2_CAP_S
    // entries fits the wide instance)
    const uint32_t nwide = sc.work_count[1], listed = sc.work_count[0], rest = listed > covered ? listed - covered : 0u;
    for (uint32_t k = blockIdx.x; k < nwide + rest; k += gridDim.x) {
        lz2_find_part<LZ2_CAP>(in, n_total, P, sc, block0, k < nwide ? sc.work[sc.work_slots - 1u - k] : sc.work[covered + (k - nwide)]);
        __syncthreads();
    }
}

// =============================================================================================
// wave-per-cluster replay
// =============================================================================================
template <int LDS_ENTRIES, int NW>
__global__ __launch_bounds__(64)
void k_lz2_big(LzP P, Lz2Scratch sc, int large)
{
    __shared__ uint32_t s_occ[LDS_ENTRIES];               // slot -> word id | position << 16 of its occupant
    __shared__ uint16_t s_slot[LDS_ENTRIES];              // entry -> slot (for its eviction)
    const uint32_t lane = threadIdx.x;
    // a launch lasts as long as its longest cluster: the 512..1024-entry class is dispatched before the 128..511 one
    // large: 0 = classes 4 (512..1024 entries) and 5 (128..511) in one launch, 1 = class 6, 2 = class 5 alone, 3 = class 4 alone
    const uint32_t nhi = large == 1 ? sc.big_count[6] : (large == 2 || large == 4) ? 0u : sc.big_count[4],
                   ncl = nhi + ((large == 0 || large == 2) ? sc.big_count[5] : large == 4 ? sc.big_count[3] : 0u);
    const uint32_t W = 1u << P.wbits;
    for (uint32_t ci = blockIdx.x; ci < ncl; ci += gridDim.x) {
        const Lz2BigDesc *dp = large == 1 ? &sc.desc[6][ci] : large == 4 ? &sc.desc[3][ci] : (ci < nhi ? &sc.desc[4][ci] : &sc.desc[5][ci - nhi]);
        const uint32_t d_block = dp->block, d_start = dp->start, n = dp->count;
        if (n & 0x80000000u) continue;                       // k_lz2_dom has replayed it (class 6 only: counts are <= LZ2_CAP otherwise)
        const uint32_t d_anom = dp->anom, d_limit = dp->limit;
        const uint16_t *bp = sc.bigpos + (size_t)d_block * LZ2_BIG_STRIDE + d_start;
        const uint16_t *br = sc.bigrs + (size_t)d_block * LZ2_BIG_STRIDE + d_start;
        const uint16_t *bi = sc.bigpid + (size_t)d_block * LZ2_BIG_STRIDE + d_start;
        uint16_t *bc = sc.bigcand + (size_t)d_block * LZ2_BIG_STRIDE + d_start;
        // (in the non-PLAIN version the first-occurrence shortcut is off: that one cluster per block keeps the literal replay)
        // the occupancy bitmap takes one register per lane and 2048 slots: a cluster is replayed with as few as it needs (every
        // first-fit and every clear walks all of them; the wide class was replayed with four whatever its size)
        const bool plain = d_anom == ~0u && d_limit == ~0u;
        if constexpr (NW > 1) {
            if (n <= 2048u) {
                if (plain) big_replay<LDS_ENTRIES, 1, true>(s_occ, s_slot, lane, W, n, d_anom, d_limit, bp, br, bi, bc);
                else big_replay<LDS_ENTRIES, 1, false>(s_occ, s_slot, lane, W, n, d_anom, d_limit, bp, br, bi, bc);
                continue;
            }
            if (n <= 4096u) {
                if (plain) big_replay<LDS_ENTRIES, 2, true>(s_occ, s_slot, lane, W, n, d_anom, d_limit, bp, br, bi, bc);
                else big_replay<LDS_ENTRIES, 2, false>(s_occ, s_slot, lane, W, n, d_anom, d_limit, bp, br, bi, bc);
                continue;
            }
        }
        if (plain) big_replay<LDS_ENTRIES, NW, true>(s_occ, s_slot, lane, W, n, d_anom, d_limit, bp, br, bi, bc);
        else big_replay<LDS_ENTRIES, NW, false>(s_occ, s_slot, lane, W, n, d_anom, d_limit, bp, br, bi, bc);
    }
}

// =============================================================================================
// Exported clusters above 1024 entries (class 6: what the wide finder hands over — a run of one byte value with foreign words inside
// its bucket range) that ONE word dominates: the dominated replay of the fallback pipeline (lz_dom.h: occupancy bits, a ring of the
// live entries' slots, a FIFO of the live foreign entries; runs of the dominant word placed up to 64 entries per wave step) instead
// of the general wave replay with a four-dword bitmap per lane, which took 18 of the "runs" family's 34 ms per 10^8 bytes.  A
// cluster it finishes is flagged in its descriptor; k_lz2_big<LZ2_CAP> skips those and replays the rest (not dominated, given up,
// or covering bucket 0 / T).  Reference behaviour emulated: algorithms/lz77/lz77.c:55-108.
// =============================================================================================
#define LZ2_DOM_DONE 0x80000000u
__global__ __launch_bounds__(256)
void k_lz2_dom(LzP P, Lz2Scratch sc)
{
    constexpr uint32_t RING = LZ2_CAP;                       // a cluster has at most LZ2_CAP entries: never more alive than that
    static_assert((RING & (RING - 1u)) == 0, "the ring is indexed with a mask");
    __shared__ uint32_t s_occ[LZ2_CAP / 32 + 72 + 264];     // occupancy bits (+ 64 zero words); the closed-form branch keeps <= block / W + 1 anchors here
    __shared__ uint16_t s_ring[RING];
    __shared__ uint16_t s_fid[DOM_FCAP], s_fpos[DOM_FCAP], s_fslot[DOM_FCAP];
    __shared__ uint32_t s_votes[3], s_result, s_next;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = 1u << P.wbits;
    const uint32_t count = sc.big_count[6];
    uint32_t *cursor = sc.big_count + 8;                     // (zeroed with the other counters by stage 1)
    struct Src {                                             // the exported arrays of lz2.h; results aligned with the entries
        const uint16_t *bp, *br, *bi; uint16_t *bc;
        __device__ __forceinline__ uint64_t ent(uint32_t i) const { return ((uint64_t)bp[i] << 16) | ((uint64_t)br[i] << 32) | ((uint64_t)bi[i] << 48); }
        __device__ __forceinline__ void put(uint32_t i, uint32_t, uint32_t res) const { bc[i] = (uint16_t)res; }
    };
    for (;;) {
        __syncthreads();
        if (tid == 0) s_next = atomicAdd(cursor, 1u);
        __syncthreads();
        const uint32_t ci = s_next;
        if (ci >= count) break;
        Lz2BigDesc *dp = &sc.desc[6][ci];
        const uint32_t m = dp->count;
        if (dp->anom != ~0u || dp->limit != ~0u || m > LZ2_CAP) continue;       // the cluster that covers bucket 0 / T: the general replay
        const size_t at = (size_t)dp->block * LZ2_BIG_STRIDE + dp->start;
        Src src{sc.bigpos + at, sc.bigrs + at, sc.bigpid + at, sc.bigcand + at};
        // the replay writes the entries that find something; everything else reads "none" (the general replay writes all of them)
        for (uint32_t i = tid; i < m; i += 256) src.bc[i] = (uint16_t)LZ_NONE16;
        if (tid == 0) { s_votes[0] = s_votes[1] = s_votes[2] = 0; s_result = 0; }
        __syncthreads();
        const bool done = dom_cluster(src, m, W, (W < RING ? W : RING) - 1u, s_occ, s_ring, s_fid, s_fpos, s_fslot, s_votes, s_result, nullptr);
        if (done && tid == 0) dp->count = m | LZ2_DOM_DONE;
    }
}

// =============================================================================================
// Row replay (round 4): SEVERAL exported clusters per wave, one per row of RL = 8 / 16 / 32 lanes (clusters of up to 256 / 512 /
// 1024 entries: eight, four or two per wave).
// The wave replay above spends ~65 wave-uniform instructions per entry and a CU issues one scalar instruction per cycle for all
// its waves: k_lz2_big alone takes 10 of the step's 51 serial milliseconds (10^9 B, one stream).  Here nothing is wave-uniform
// except the step counter: a row owns one cluster — its occupancy bitmap is RL dwords of LDS, one per lane; first fit = every
// lane masks its dword, a ballot, the row's bits of it, ds_bpermute of the winning dword; a step's retirements (FIFO,
// lz77.c:70-76: up to RL at once, a prefix of the live entries because positions ascend) are one LDS `and` per retiring lane;
// the occupant of a slot and the slot + position of an entry are one LDS dword each — so a vector instruction serves all the
// rows.  All global memory traffic happens at BOUNDARIES between passes of RL steps, uses before issues (a wait inside the
// steps, or for a load some other row has just issued, would stall every row): the results of the last RL entries go out, the
// next RL entries of every row were loaded a pass ago, and a three-stage prefetch (cursor -> descriptor -> first entries) has
// the next cluster ready when a row finishes; rows start clusters on boundaries only.  16.25 KiB of LDS per wave for any RL.
// Clusters come off a cursor (descriptor order); the one cluster per block that covers bucket 0 / T (anom / limit) is left to
// k_lz2_big, which skips the descriptors flagged here.  Reference behaviour emulated: algorithms/lz77/lz77.c:55-108.
// =============================================================================================
#define ROW_BPERM(src_lane, v) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)((src_lane) << 2), (int)(v)))

// first free slot at or above `from` of the row's bitmap (lane l of the row holds dword l in v), or ~0u
template <int RL>
__device__ __forceinline__ uint32_t row_first_zero(uint32_t v, uint32_t from, uint32_t l, uint32_t rb)
{
    constexpr uint32_t RMASK = RL == 32 ? 0xFFFFFFFFu : ((1u << (RL & 31)) - 1u);
    const uint32_t rw = from >> 5, lowmask = (1u << (from & 31u)) - 1u;
    if (l < rw) v = 0xFFFFFFFFu; else if (l == rw) v |= lowmask;
    const uint64_t mk = __ballot(v != 0xFFFFFFFFu);
    const uint32_t my = (uint32_t)(mk >> rb) & RMASK;                       // this row's lanes that still have a free slot
    const uint32_t ln = my ? (uint32_t)__builtin_ctz(my) : 0u;
    const uint32_t mv = ROW_BPERM(rb + ln, v);
    const uint32_t fz = (mv != 0xFFFFFFFFu) ? (uint32_t)__builtin_ctz(~mv) : 0u;
    return my ? ((ln << 5) + fz) : ~0u;
}

template <int RL>                                         // lanes per row: 8, 16 or 32 (clusters of up to 256, 512, 1024 entries)
__global__ __launch_bounds__(64)
void k_lz2_rows(LzP P, Lz2Scratch sc, int cls_a, int cls_b, int cursor_slot)
{
    static_assert(RL == 8 || RL == 16 || RL == 32, "rows of 8, 16 or 32 lanes");
    constexpr int NR = 64 / RL, CAPE = 32 * RL;            // rows per wave; slots of a row's bitmap (one dword per lane): 16.25 KiB per wave
    constexpr uint32_t RMASK = RL == 32 ? 0xFFFFFFFFu : ((1u << (RL & 31)) - 1u);
    __shared__ uint32_t s_occ[NR][CAPE];                   // slot -> word id | position << 16 of its occupant
    __shared__ uint32_t s_ent[NR][CAPE];                   // entry -> its slot | its position << 16 (for its eviction)
    __shared__ uint32_t s_bm[NR][RL];
    const uint32_t lane = threadIdx.x, g = lane / (uint32_t)RL, l = lane & (uint32_t)(RL - 1), rb = lane & ~(uint32_t)(RL - 1);
    const uint32_t W = 1u << P.wbits;
    // two descriptor lists behind one cursor, the class of the LONGER clusters first: a launch lasts as long as its last chain
    const uint32_t count_a = sc.big_count[cls_a], count = count_a + sc.big_count[cls_b];
    uint32_t *cursor = sc.big_count + cursor_slot;         // (zeroed with the other counters by stage 1)
    Lz2BigDesc *const descs_a = sc.desc[cls_a], *const descs_b = sc.desc[cls_b];
    auto desc_of = [&](uint32_t ticket) -> Lz2BigDesc * { return ticket < count_a ? descs_a + ticket : descs_b + (ticket - count_a); };
    uint32_t *occ = s_occ[g], *ent = s_ent[g], *bmw = s_bm[g];

    bool active = false, exhausted = false;
    uint32_t n = 0, i0 = 0, ev = 0;
    size_t at = 0;
    uint32_t cur_a = 0, cur_b = 0, out_acc = LZ_NONE16;
    // the next cluster of this row: 0 = nothing, 1 = cursor value in flight, 2 = descriptor in flight, 3 = first entries in flight / ready
    uint32_t pf_stage = 0, pf_ci = 0, pf_count = 0;
    size_t pf_at = 0;
    // Registers that ONLY loads write (no arithmetic on them before the next boundary, no conditional assignment: either would put
    // a wait for the load right behind its issue).  Every boundary reloads all of them, from a harmless address when a row has
    // nothing to ask for: r_n* = entries i0 + RL .. of the running cluster, r_f* = first RL entries of the prefetched cluster,
    // r_d* = its descriptor, tk = lane 0's ticket from the cursor.
    uint32_t r_np = 0, r_ni = 0, r_nr = 0, r_fp = 0, r_fi = 0, r_fr = 0, tk = 0;
    uint4 r_d4 = make_uint4(0, 0, 0, 0); uint32_t r_d1 = 0;
    // a zero the compiler cannot see through: on a provably uniform address it turns the cursor's atomic into its wave-aggregated
    // form and reads the result back on the spot
    uint32_t vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

    for (;;) {
        // ================= boundary: USES (everything read here was loaded at least one pass ago) =================
        const bool flush = active && i0 + l < n;
        uint16_t *const flush_to = sc.bigcand + at + i0 + l;
        if (active) {
            i0 += (uint32_t)RL;
            if (i0 >= n) active = false;
            else { cur_a = r_np | (r_ni << 16); cur_b = r_nr; }
        }
        bool take = false, want_cursor = false;
        uint32_t done_ci = 0, done_n = 0;
        if (!active && pf_stage == 3u) {
            take = true;
            n = pf_count; at = pf_at;
            cur_a = r_fp | (r_fi << 16); cur_b = r_fr; i0 = 0; ev = 0; active = true; pf_stage = 0;
            bmw[l] = 0;
            done_ci = pf_ci; done_n = n;
        }
        if (pf_stage == 1u) {
            pf_ci = ROW_BPERM(rb, tk);                                      // lane 0 of the row asked the cursor
            if (pf_ci >= count) { exhausted = true; pf_stage = 0; } else pf_stage = 2;
        } else if (pf_stage == 2u) {
            // the cluster that covers bucket 0 / T (and anything that does not fit): not ours, another one
            if (r_d4.w != ~0u || r_d1 != ~0u || r_d4.z > (uint32_t)CAPE || r_d4.z < 2u * RL) pf_stage = 0;
            else { pf_count = r_d4.z; pf_at = (size_t)r_d4.x * LZ2_BIG_STRIDE + r_d4.y; pf_stage = 3; }
        }
        if (pf_stage == 0u && !exhausted) { want_cursor = true; pf_stage = 1; }
        // ================= boundary: ISSUES (all unconditional loads; nothing below waits for them) =================
        {
            const uint32_t k = i0 + (uint32_t)RL + l;
            const size_t na = (active && k < n) ? at + k : 0;
            r_np = sc.bigpos[na]; r_ni = sc.bigpid[na]; r_nr = sc.bigrs[na];
            const size_t fa = (pf_stage == 3u) ? pf_at + l : 0;              // (>= 2 RL entries: the first RL exist)
            r_fp = sc.bigpos[fa]; r_fi = sc.bigpid[fa]; r_fr = sc.bigrs[fa];
            const uint32_t *dp = reinterpret_cast<const uint32_t *>(pf_stage == 2u ? desc_of(pf_ci) : descs_a);
            r_d4 = *reinterpret_cast<const uint4 *>(dp); r_d1 = dp[4];
        }
        if (flush) *flush_to = (uint16_t)out_acc;
        if (take && l == 0) desc_of(done_ci)->count = done_n | LZ2_DOM_DONE;     // k_lz2_big skips it
        if (want_cursor && l == 0) tk = atomicAdd(cursor + vzero, 1u);
        if (__ballot(active || pf_stage != 0u) == 0ull) break;
        if (__ballot(active) == 0ull) continue;
        // ================= RL steps: LDS, cross-lane and vector work only =================
        // Software-pipelined: an entry's broadcast, the live entries a step may retire and the bitmap dwords the first fit needs
        // are asked for as early as their addresses are known, and every loaded value is used unconditionally (a load whose
        // only use sits under an `if` is sunk into that block together with the wait for it).
        uint32_t a = ROW_BPERM(rb, cur_a), r = ROW_BPERM(rb, cur_b);
        uint32_t e = ent[(active && ev + l < i0) ? ev + l : 0u];
        for (uint32_t t = 0; t < (uint32_t)RL; ++t) {
            const uint32_t i = i0 + t;
            const bool on = active && i < n;
            const uint32_t p = a & 0xFFFFu, id = a >> 16;
            for (;;) {                                                      // FIFO retirement (lz77.c:70-76): normally one pass
                const bool ret = on && ev + l < i && (e >> 16) + W < p;
                const uint64_t mk = __ballot(ret);
                if (ret) atomicAnd(&bmw[(e & 0xFFFFu) >> 5], ~(1u << (e & 31u)));       // clears the bucket
                const uint32_t cnt = (uint32_t)__popc((uint32_t)(mk >> rb) & RMASK);
                ev += cnt;
                if (__ballot(cnt == (uint32_t)RL) == 0ull) break;           // a row whose every lane retired one may have more
                e = ent[(on && ev + l < i) ? ev + l : 0u];
            }
            // (behind the clears, in the LDS queue's order) the home's dword and occupant, this lane's bitmap dwords
            const uint32_t wr = bmw[r >> 5], h = occ[r], v = bmw[l];
            const uint32_t tn = (t + 1u) & (uint32_t)(RL - 1);
            const uint32_t na = ROW_BPERM(rb + tn, cur_a), nr = ROW_BPERM(rb + tn, cur_b);
            const bool bit = (wr >> (r & 31u)) & 1u, same = (h & 0xFFFFu) == id;
            const bool seek = on && id != p;                                // (the first occurrence of a word in the block finds nothing, ever)
            // nothing evicted yet: the word's first occurrence = the word id; else the occupant of the home, if it is a copy
            uint32_t res = (seek && ev == 0u) ? id : (seek && bit && same) ? (h >> 16) : (uint32_t)LZ_NONE16;
            bool walk = seek && ev != 0u && bit && !same;
            if (__ballot(walk) != 0ull) {
                // the home holds another word: on to the first copy of this word or the first empty bucket, RL occupants per step
                uint32_t e_end = row_first_zero<RL>(v, r + 1u, l, rb);   // (a walk changes no bit)
                if (e_end > (uint32_t)CAPE) e_end = (uint32_t)CAPE;
                uint32_t b0 = r + 1u;
                while (__ballot(walk && b0 < e_end) != 0ull) {
                    const uint32_t bb = b0 + l;
                    const bool in = walk && bb < e_end;
                    const uint32_t o = occ[in ? bb : 0u];
                    const uint64_t mk = __ballot(in && (o & 0xFFFFu) == id);
                    const uint32_t my = (uint32_t)(mk >> rb) & RMASK;
                    const uint32_t src = ROW_BPERM(rb + (my ? (uint32_t)__builtin_ctz(my) : 0u), o);
                    if (walk && my) { res = src >> 16; walk = false; }
                    b0 += (uint32_t)RL;
                }
            }
            // insert: first fit from the home on (inside the cluster by the parking bound)
            const uint32_t b = row_first_zero<RL>(v, r, l, rb);
            const bool put = on && b < (uint32_t)CAPE;
            if (put && l == 0) atomicOr(&bmw[b >> 5], 1u << (b & 31u));
            if (put) { occ[b] = id | (p << 16); ent[i] = b | (p << 16); }
            if (on && l == t) out_acc = res;
            e = ent[(active && i + 1u < n && ev + l < i + 1u) ? ev + l : 0u];     // (behind this step's own record)
            a = na; r = nr;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// =============================================================================================
// lane-per-cluster replay of exported clusters of one size class [16,32) / [32,64) / [64,128):
// 64 clusters per wave, every lane owns a private LDS region (slot -> word id, slot -> position,
// entry -> slot, occupancy bits).  All lanes of a wave step through clusters of similar size.
// =============================================================================================
template <bool SMALL> struct SlotType { typedef uint8_t type; };
template <> struct SlotType<false> { typedef uint16_t type; };

// Measured alternatives (round 1, same box): staging 16 entries of all 64 clusters through LDS by coalesced loads ran
// 14 % faster alone (7.16 vs 8.35 ms / 400 MB) but its 3x LDS footprint cost the overlapped pipeline 4 % (10.46 vs 10.88
// GB/s) — k_lz2_find needs the whole LDS of a CU for its two workgroups; packing an entry into one 8-byte record
// fetched four steps ahead ran 19 % faster alone (6.76 ms) and cost the pipeline 2 % (11.24 vs 11.46: the export in
// k_lz2_find writes twice the bytes); lane replay of 128..511-entry clusters was 4x slower than the wave replay (one
// wave per CU).  All three were removed.
__device__ __forceinline__ uint32_t u16_of(const uint4 &v, uint32_t k)      // k-th 16-bit element of a 16-byte vector
{
    // two 64-bit halves and a variable shift: a select chain on the four dwords is turned into an indexed scratch access
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    return (uint32_t)(((k & 4u) ? hi : lo) >> ((k & 3u) * 16u)) & 0xFFFFu;
}

template <int CMAX, int LANES>
__global__ __launch_bounds__(64)
void k_lz2_mid_direct(LzP P, Lz2Scratch sc, int cls)
{
    constexpr int STRIDE = CMAX + 1;                      // odd stride: spreads the lanes over the banks
    typedef typename SlotType<(CMAX <= 128)>::type slot_t;     // a slot number fits a byte up to 255
    __shared__ uint32_t s_occ[LANES * STRIDE];            // slot -> word id | position << 16 of its occupant
    __shared__ slot_t   s_slot[LANES * STRIDE];           // entry -> slot (for its eviction)
    __shared__ uint32_t s_bits[LANES * (CMAX / 32 + 1)];
    const uint32_t lane = threadIdx.x;
    const uint32_t ncl = sc.big_count[cls];
    // persistent grid: a wave takes LANES clusters at a time until the class is done (a worst-case grid was 131 k
    // workgroups for the 8..15 class, nearly all of them empty)
    for (uint32_t wg = blockIdx.x; wg * (uint32_t)LANES < ncl; wg += gridDim.x) {
    const uint32_t ci = wg * (uint32_t)LANES + lane;
    const bool active = ci < ncl && lane < (uint32_t)LANES;
    struct { uint32_t block, start, count, anom, limit; } d = {0u, 0u, 0u, ~0u, ~0u};
    if (active) { const Lz2BigDesc *dp = &sc.desc[cls][ci]; d.block = dp->block; d.start = dp->start; d.count = dp->count; d.anom = dp->anom; d.limit = dp->limit; }
    const uint32_t n = d.count, W = 1u << P.wbits;
    // Every lane walks its own cluster, so nothing coalesces, and 2-byte loads cost a cache line each: rocprofv3 showed
    // 1.5 GB of HBM traffic per launch against 60 MB of entries (the lines do not survive in L1 between steps).  Clusters
    // start on 16-byte boundaries (lz2.h), so a lane fetches EIGHT entries per load and stores eight results at once.
    const uint4 *vp = reinterpret_cast<const uint4 *>(sc.bigpos + (size_t)d.block * LZ2_BIG_STRIDE + d.start);
    const uint4 *vr = reinterpret_cast<const uint4 *>(sc.bigrs + (size_t)d.block * LZ2_BIG_STRIDE + d.start);
    const uint4 *vi = reinterpret_cast<const uint4 *>(sc.bigpid + (size_t)d.block * LZ2_BIG_STRIDE + d.start);
    uint4 *vc = reinterpret_cast<uint4 *>(sc.bigcand + (size_t)d.block * LZ2_BIG_STRIDE + d.start);
    const uint32_t lr = lane < (uint32_t)LANES ? lane : 0u;  // surplus lanes (LANES < 64) idle on region 0: n = 0
    uint32_t *occ = s_occ + lr * STRIDE;
    slot_t *slot = s_slot + lr * STRIDE;
    uint32_t *bits = s_bits + lr * (CMAX / 32 + 1);
    if (lane < (uint32_t)LANES) for (int k = 0; k < CMAX / 32 + 1; ++k) bits[k] = 0;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 evw = zero4;                                     // positions of entries [ev & ~7, +8): the next to retire
    if (n) evw = vp[0];
    uint32_t ev = 0, ev_p = evw.x & 0xFFFFu;
    bool anom_pending = d.anom != ~0u;
    const bool plain = d.anom == ~0u && d.limit == ~0u;    // not the cluster that covers bucket 0 / T
    uint4 np = evw, nr = zero4, ni = zero4;                // the group after the current one is in flight
    if (n) { nr = vr[0]; ni = vi[0]; }
    for (uint32_t i0 = 0; i0 < (uint32_t)CMAX; i0 += 8) {
        if (__ballot(i0 < n) == 0ull) break;                // every cluster of this wave is done
        const uint4 cp = np, cr = nr, cid = ni;
        if (i0 + 8 < n) { np = vp[(i0 >> 3) + 1]; nr = vr[(i0 >> 3) + 1]; ni = vi[(i0 >> 3) + 1]; }
        uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t i = i0 + k;
            if (i < n) {                                   // lanes with shorter clusters idle (same size class: < 2x)
                const uint32_t p = u16_of(cp, k), r = u16_of(cr, k), id = u16_of(cid, k);
                while (ev < i && ev_p + W < p) {           // FIFO retirement
                    const uint32_t b = slot[ev];
                    bits[b >> 5] &= ~(1u << (b & 31u));
                    ++ev;
                    if ((ev & 7u) == 0) evw = vp[ev >> 3];
                    ev_p = u16_of(evw, ev & 7u);
                }
                if (anom_pending && p > W - 1u) { bits[d.anom >> 5] &= ~(1u << (d.anom & 31u)); anom_pending = false; }
                uint32_t wi = r >> 5;
                const uint32_t w0 = bits[wi];
                uint32_t res = LZ_NONE16;
                if (plain && ev == 0) {                    // nothing evicted yet: find() = the word's first occurrence,
                    if (id != p) res = id;                 // which is what the word id is (k_lz2_find, the sweep)
                } else if (id != p && ((w0 >> (r & 31u)) & 1u)) {       // (a word's first occurrence in the block finds nothing, ever)
                    for (uint32_t b = r;; ++b) {
                        if (b != r) {
                            if (b == d.limit && r < d.limit) break;
                            if (!((bits[b >> 5] >> (b & 31u)) & 1u)) break;
                        }
                        const uint32_t o = occ[b];
                        if ((o & 0xFFFFu) == id) { res = o >> 16; break; }
                    }
                }
                { const uint32_t v = res << ((k & 1u) * 16u); if (k < 2) o0 |= v; else if (k < 4) o1 |= v; else if (k < 6) o2 |= v; else o3 |= v; }
                uint32_t wv = w0 | ((1u << (r & 31u)) - 1u);
                while (wv == 0xFFFFFFFFu) wv = bits[++wi];
                const uint32_t b = (wi << 5) + (uint32_t)__builtin_ctz(~wv);
                bits[b >> 5] |= 1u << (b & 31u);
                occ[b] = id | (p << 16); slot[i] = (slot_t)b;
            }
        }
        if (i0 < n) vc[i0 >> 3] = make_uint4(o0, o1, o2, o3);      // pads of the last group: 0 (= skipped, lz2.h)
    }
    __builtin_amdgcn_wave_barrier();
    }
}

// lists -> by-position array (test hook mi_lz_find_all_dev and the fallback boundary)
__global__ __launch_bounds__(1024)
void k_lz2_scatter(Lz2Scratch sc, uint16_t *__restrict__ cand_by_pos /* [nb][65536] */)
{
    const uint32_t lb = blockIdx.x;
    const Lz2BlockMeta *mt = sc.meta + lb;
    if (mt->fallback) return;
    const uint32_t n = mt->n;
    const uint16_t *pl = sc.plist + (size_t)lb * LZ_MAX_BLOCK, *cd = sc.cand + (size_t)lb * LZ_MAX_BLOCK;
    uint16_t *out = cand_by_pos + (size_t)lb * LZ_MAX_BLOCK;
    for (uint32_t j = threadIdx.x; j < n; j += 1024) { const uint32_t c = cd[j], p = pl[j]; if (c != p || p == LZ_NONE16) out[p] = (uint16_t)c; }
    __syncthreads();                                                     // position 0xFFFF: "none" first, the exported result over it
    const uint32_t nb = mt->nbig_entries;
    const uint16_t *bp = sc.bigpos + (size_t)lb * LZ2_BIG_STRIDE, *bc = sc.bigcand + (size_t)lb * LZ2_BIG_STRIDE;
    for (uint32_t j = threadIdx.x; j < nb; j += 1024) { const uint32_t p = bp[j], c = bc[j]; if (c != p) out[p] = (uint16_t)c; }    // c == p: a pad
}

template __global__ void k_lz2_mid_direct<16, 64>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_mid_direct<32, 64>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_mid_direct<64, 64>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_mid_direct<128, 48>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_big<LZ2_BIG_SMALL, 1>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_big<512, 1>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_big<256, 1>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_big<LZ2_CAP, 4>(LzP, Lz2Scratch, int);
template __global__ void k_lz2_rows<16>(LzP, Lz2Scratch, int, int, int);


// =============================================================================================
// host side
// =============================================================================================
// clusters of class c per block, at most (a block has 65536 entries)
static uint32_t lz2_class_cap(uint32_t c)
{
    static const uint32_t lo[LZ2_NCLASS] = {16, 32, 64, 128, 256, 128, 1025, 8};    // smallest cluster a class can hold (either mode)
    return LZ_MAX_BLOCK / lo[c] + 8;
}

size_t lz2_scratch_bytes(uint32_t nb)
{
    size_t descs = 0;                                    // what lz2_carve takes for the class descriptor arrays, exactly
    for (uint32_t c = 0; c < LZ2_NCLASS; ++c) descs += lz2_class_cap(c);
    return (size_t)nb * (LZ_MAX_BLOCK * (2 * 2 + 1) + LZ2_BIG_STRIDE * 2 * 4 + sizeof(Lz2BlockMeta) + 4 + 8 * LZ2_MAXPARTS + descs * sizeof(Lz2BigDesc)) + 16 * 256 + 4096 + 64 * 256;
}

void lz2_carve(mi_carver &cv, uint32_t nb, Lz2Scratch *sc)
{
    sc->partmap = cv.take<uint8_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->plist = cv.take<uint16_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->cand = cv.take<uint16_t>((size_t)nb * LZ_MAX_BLOCK);
    sc->meta = cv.take<Lz2BlockMeta>(nb);
    sc->fallback_count = cv.take<uint32_t>(64);
    sc->fallback_list = cv.take<uint32_t>(nb);
    sc->bigpos = cv.take<uint16_t>((size_t)nb * LZ2_BIG_STRIDE);
    sc->bigrs = cv.take<uint16_t>((size_t)nb * LZ2_BIG_STRIDE);
    sc->bigpid = cv.take<uint16_t>((size_t)nb * LZ2_BIG_STRIDE);
    sc->bigcand = cv.take<uint16_t>((size_t)nb * LZ2_BIG_STRIDE);
    for (uint32_t c = 0; c < LZ2_NCLASS; ++c) sc->desc[c] = cv.take<Lz2BigDesc>((size_t)nb * lz2_class_cap(c));
    sc->big_count = sc->fallback_count + 16;
    sc->work_count = sc->fallback_count + 32;            // zeroed with the other counters by stage 1
    sc->work = cv.take<uint64_t>((size_t)nb * LZ2_MAXPARTS);
    sc->work_slots = nb * LZ2_MAXPARTS;
    sc->dbg = getenv("MI_LZ_DEBUG") ? cv.take<uint64_t>(64) : nullptr;
    // bits 16..: the row replay's switch (MI_LZ_ROWS, lz2_stage_b): with it the wave classes are three (128..255 apart)
    { const char *e = getenv("MI_LZ_ROWS"); const uint32_t rows = e ? (uint32_t)atoi(e) : 1u; sc->wave_min = LZ2_WAVE | ((rows ? 1u : 0u) << 16); }
#ifdef MI_MEASURE
    sc->stop_phase = getenv("MI_LZ_STOP_PHASE") ? (uint32_t)atoi(getenv("MI_LZ_STOP_PHASE")) : 0u;
#else
    sc->stop_phase = 0u;
#endif
}

static uint32_t lz2_env_u32(const char *name, uint32_t dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    const long v = atol(e);
    return v >= 0 ? (uint32_t)v : dflt;
}

// phase cycle counters of the last LZ call of this context (MI_LZ_DEBUG=1; a development hook, not in the public header)
extern "C" int mi_lz_debug_counters(mi_ctx *ctx, uint64_t *out32)
{
    if (!ctx || !ctx->lz_dbg) return 0;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out32, ctx->lz_dbg, 64 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 1 : 0;      // 64 counters: [0..31] find / parse, [32..47] partition
}

void lz2_launch_partition(const uint8_t *d_in, uint64_t n, const LzP &P, const Lz2Scratch &sc, uint64_t block0, uint32_t nb, hipStream_t s);

// stage A1: partition (also decides which blocks go to the fallback pipeline)
mi_status lz2_stage_partition(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                              const Lz2Scratch &sc, hipStream_t s)
{
    MI_HIP(ctx, hipMemsetAsync(sc.fallback_count, 0, 256, s));      // fallback_count and big_count[]
    if (sc.dbg && ctx->lz_dbg != sc.dbg) { ctx->lz_dbg = sc.dbg; MI_HIP(ctx, hipMemsetAsync(sc.dbg, 0, 512, s)); }
    mi_prof_scope p(ctx, "k_lz2_partition", s, (uint64_t)nb * P.block);
    lz2_launch_partition(d_in, n, P, sc, block0, nb, s);
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// stage A2: per-part find (LDS heavy: three workgroups per CU)
static uint32_t lz2_find_grid(const LzP &P, uint32_t nb)
{
    // the partition lists a block's parts (greedy, data dependent: 26-27 for a full block of text); the grid covers
    // LZ2_GRID_PARTS per block, rounded up to the 8 XCD slices of the kernel's workgroup mapping (a grid that is not a multiple
    // of 8 would leave items of the last row unassigned: ADVICE r3), the looping wide kernel takes whatever lies beyond
    const uint32_t parts = P.block / 64u + 1u < LZ2_GRID_PARTS ? P.block / 64u + 1u : LZ2_GRID_PARTS;
    return (parts * nb + 7u) & ~7u;
}

mi_status lz2_stage_find(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                         const Lz2Scratch &sc, hipStream_t s)
{
    mi_prof_scope p(ctx, "k_lz2_find", s, (uint64_t)nb * P.block);
    hipLaunchKernelGGL(k_lz2_find, dim3(lz2_find_grid(P, nb)), dim3(LZ2_THREADS), 0, s, d_in, n, P, sc, block0);
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// parts of 2561..4096 entries (none in text) and whatever a batch lists beyond k_lz2_find's grid: 256 looping workgroups, gone at
// once when there is nothing to do.  Its own launch and — in the pipelined encoder — NOT on the stream of partition and find: an
// empty launch still has to be given 76 KiB of LDS per workgroup before it can leave, and it waited about a millisecond per batch
// for that behind the parse kernel of the batch before, on the one chain the pipeline is bound by (kernel timeline, round 4:
// partition 3.7 + find 4.45 + this 1.0 ms of a 9.3 ms batch period)
mi_status lz2_stage_find_wide(mi_ctx *ctx, const LzP &P, const uint8_t *d_in, uint64_t n, uint64_t block0, uint32_t nb,
                              const Lz2Scratch &sc, hipStream_t s, bool aside)
{
    // on the side stream this launch is normally empty and merely waits for its LDS behind the other stages (6 ms of "duration"
    // per batch in rocprofv3): timing it would report that wait as the pipeline's dominant kernel (as for the fallback chain,
    // lz_find_batch); MI_LZ_PROF_FALLBACK=1 times it (inputs with wide parts: scripts/adv_profile.py)
    static const bool prof_fb = getenv("MI_LZ_PROF_FALLBACK") != nullptr;
    const int saved_prof = ctx->profiling;
    if (aside && !prof_fb) ctx->profiling = 0;
    {
        mi_prof_scope p(ctx, "k_lz2_find_wide", s, (uint64_t)nb * P.block);
        hipLaunchKernelGGL(k_lz2_find_wide, dim3(256), dim3(LZ2_THREADS), 0, s, d_in, n, P, sc, block0, lz2_find_grid(P, nb));
    }
    ctx->profiling = saved_prof;
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// stage B: replay of the exported clusters (almost no LDS: runs beside the next batch's stage A)
// which: 1 = the lane replays of 8..31-entry clusters, 4 = of 32..127-entry clusters, 2 = the wave / row replays; 7 = all on `s`
mi_status lz2_stage_b(mi_ctx *ctx, const LzP &P, uint32_t nb, const Lz2Scratch &sc, hipStream_t s, int which)
{
    // Replay grids cover the worst case and the kernels stride, so any grid is correct.  MI_LZ_REPLAY_WAVES=k caps them at
    // k workgroups per CU (0 / unset = worst case).  Measured (round 2, same box): persistent grids sized by LDS (16 / 12 /
    // 7 / 5 / 16 waves per CU) left the lane classes unchanged and made the wave replay 10.9 -> 15.8 ms per GB — a static
    // stride deals a wave whatever chain lengths it draws, the hardware dispatcher hands the next cluster to the first
    // wave that is free.
    const uint32_t ncu = (uint32_t)ctx->num_cu;
    auto grid_of = [&](uint64_t worst) -> uint32_t {
        const uint64_t g = (uint64_t)ncu * lz2_env_u32("MI_LZ_REPLAY_WAVES", 0);             // 0 = the worst-case grid
        return (uint32_t)((g == 0 || worst < g) ? (worst ? worst : 1) : g);
    };
    if (which & 1) {
    { mi_prof_scope p(ctx, "k_lz2_mid<16>", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lz2_mid_direct<16, 64>), dim3(grid_of((uint64_t)nb * lz2_class_cap(7) / 64 + 1)), dim3(64), 0, s, P, sc, 7); }
    { mi_prof_scope p(ctx, "k_lz2_mid<32>", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lz2_mid_direct<32, 64>), dim3(grid_of((uint64_t)nb * lz2_class_cap(0) / 64 + 1)), dim3(64), 0, s, P, sc, 0); }
    }
    if (which & 4) {
    { mi_prof_scope p(ctx, "k_lz2_mid<64>", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lz2_mid_direct<64, 64>), dim3(grid_of((uint64_t)nb * lz2_class_cap(1) / 64 + 1)), dim3(64), 0, s, P, sc, 1); }
    { mi_prof_scope p(ctx, "k_lz2_mid<128>", s, (uint64_t)nb * P.block);
      // 48 clusters per wave: 31 KiB of LDS instead of 42, five waves per CU instead of three (240 replaying lanes, not 192)
      hipLaunchKernelGGL((k_lz2_mid_direct<128, 48>), dim3(grid_of((uint64_t)nb * lz2_class_cap(2) / 48 + 1)), dim3(64), 0, s, P, sc, 2); }
    }
    if (which & 2) {
    { mi_prof_scope p(ctx, "k_lz2_big", s, (uint64_t)nb * P.block);
      // the 512..1024-entry class first and alone on the wave replay (6 KiB of LDS per wave), then the 128..511-entry clusters: on the
      // row replay (below), or — MI_LZ_ROWS=0 — on the wave replay with 3 KiB per wave (MI_LZ_BIG_SPLIT=0: one launch for both wave
      // classes, A/B)
      static const bool split = !(getenv("MI_LZ_BIG_SPLIT") && getenv("MI_LZ_BIG_SPLIT")[0] == '0');
      // row replay (four clusters per wave, off a cursor) of the 128..511-entry clusters, the 256..511 ones first (MI_LZ_ROWS=0: the
      // wave replay for everything, A/B); k_lz2_big then takes what the rows leave (the cluster that covers bucket 0 / T) with
      // small grids that stride over the descriptors
      const uint32_t rows = sc.wave_min >> 16;
      static const uint32_t rows_waves = lz2_env_u32("MI_LZ_ROWS_WAVES", 9);
      if (rows >= 1 && split && (sc.wave_min & 0xFFFFu) == LZ2_WAVE) {
          hipLaunchKernelGGL((k_lz2_big<LZ2_BIG_SMALL, 1>), dim3(grid_of((uint64_t)nb * lz2_class_cap(4))), dim3(64), 0, s, P, sc, 3);
          hipLaunchKernelGGL((k_lz2_rows<16>), dim3(ncu * rows_waves), dim3(64), 0, s, P, sc, 5, 3, 9);
          hipLaunchKernelGGL((k_lz2_big<512, 1>), dim3(ncu * 4u), dim3(64), 0, s, P, sc, 2);
          hipLaunchKernelGGL((k_lz2_big<256, 1>), dim3(ncu * 4u), dim3(64), 0, s, P, sc, 4);
      } else if (split && (sc.wave_min & 0xFFFFu) == LZ2_WAVE) {
          hipLaunchKernelGGL((k_lz2_big<LZ2_BIG_SMALL, 1>), dim3(grid_of((uint64_t)nb * lz2_class_cap(4))), dim3(64), 0, s, P, sc, 3);
          hipLaunchKernelGGL((k_lz2_big<512, 1>), dim3(grid_of((uint64_t)nb * lz2_class_cap(5))), dim3(64), 0, s, P, sc, 2);
      } else {
          hipLaunchKernelGGL((k_lz2_big<LZ2_BIG_SMALL, 1>), dim3(grid_of((uint64_t)nb * (lz2_class_cap(4) + lz2_class_cap(5)))), dim3(64), 0, s, P, sc, 0);
          // (with the row replay's three classes but MI_LZ_BIG_SPLIT=0: the 128..255 class has its own list)
          if (rows) hipLaunchKernelGGL((k_lz2_big<256, 1>), dim3(grid_of((uint64_t)nb * lz2_class_cap(3))), dim3(64), 0, s, P, sc, 4);
      } }
    { mi_prof_scope p(ctx, "k_lz2_dom", s, (uint64_t)nb * P.block);
      // clusters above 1024 entries that one word dominates (none in text: an empty launch of 13 KiB workgroups); what it leaves
      // goes to the general replay below
      const uint64_t worst = (uint64_t)nb * lz2_class_cap(6);
      hipLaunchKernelGGL(k_lz2_dom, dim3((unsigned)(worst < (uint64_t)ncu * 4u ? worst : (uint64_t)ncu * 4u)), dim3(256), 0, s, P, sc); }
    { mi_prof_scope p(ctx, "k_lz2_big<4096>", s, (uint64_t)nb * P.block);
      hipLaunchKernelGGL((k_lz2_big<LZ2_CAP, 4>), dim3(grid_of((uint64_t)nb * lz2_class_cap(6) < 4096 ? (uint64_t)nb * lz2_class_cap(6) : 4096)), dim3(64), 0, s, P, sc, 1); }   // 24 KiB each, normally none: a small striding grid
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

void lz2_launch_scatter(const Lz2Scratch &sc, uint16_t *cand_by_pos, uint32_t nb, hipStream_t s)
{
    hipLaunchKernelGGL(k_lz2_scatter, dim3(nb), dim3(1024), 0, s, sc, cand_by_pos);
}
