// lz_dom.h — the dominated replay of ONE giant cluster (k_lz_emulate_dom's core, lz_find.hip has the description), shared by
// the fallback pipeline (lz_find.hip) and by the LDS-resident finder's wide clusters (k_lz2_dom, lz2_find.hip).
#pragma once
#include "lz_common.h"
#include "lz_replay.h"

#define DOM_FCAP 768u
#define DOM_RING 32768u

__device__ __forceinline__ uint32_t dom_first_zero(const volatile uint32_t *occ, uint32_t from, uint32_t lane)
{
    const uint32_t w0 = from >> 5;
    for (uint32_t base = w0;; base += 64u) {                                   // (the array ends in 64 zero words)
        const uint32_t wi = base + lane;
        uint32_t v = occ[wi];
        if (wi == w0) v |= (1u << (from & 31u)) - 1u;
        const uint64_t nz = __ballot(v != 0xFFFFFFFFu);
        if (nz) {
            const uint32_t ln = (uint32_t)__builtin_ctzll(nz);
            return ((base + ln) << 5) + (uint32_t)__builtin_ctz(~RLANE(v, ln));
        }
    }
}

// One giant cluster of m entries on one workgroup of 256 threads (the serial part on its first wave).  SRC hands out the entries
// — ent(i): pos << 16 | home slot relative to the cluster << 32 | word id << 48 — and takes the results — put(i, pos, res) —, so
// the same replay serves the fallback pipeline's records (k_lz_emulate_dom) and the clusters the LDS-resident finder exports
// (k_lz2_dom, lz2_find.hip).  rmask: ring index mask, min(W, ring entries) - 1 (at most that many entries are alive at once).
// Returns true when every result has been written; false: not dominated, or given up (whatever was written has been wiped to
// "none") — the caller's general replay redoes the cluster.
template <typename SRC>
__device__ __forceinline__ bool dom_cluster(SRC &src, const uint32_t m, const uint32_t W, const uint32_t rmask, uint32_t *s_occ, uint16_t *s_ring,
                                            uint16_t *s_fid, uint16_t *s_fpos, uint16_t *s_fslot, uint32_t *s_votes, uint32_t &s_result, uint64_t *dbg)
{
        const uint32_t tid = threadIdx.x, lane = tid & 63u;
        long long tk0 = dbg ? clock64() : 0;
        if (dbg && tid == 0) { atomicAdd((unsigned long long *)&dbg[48], 1ull); atomicAdd((unsigned long long *)&dbg[49], (unsigned long long)m); }
        // the dominant word: three entries vote (one of them may be foreign), every entry is counted against them
        const uint32_t cidx[3] = {m / 2, m / 4, (3 * m) / 4};
        const uint32_t cpid[3] = {(uint32_t)(src.ent(cidx[0]) >> 48), (uint32_t)(src.ent(cidx[1]) >> 48), (uint32_t)(src.ent(cidx[2]) >> 48)};
        {
            uint32_t v0 = 0, v1 = 0, v2 = 0;
            for (uint32_t i = tid; i < m; i += 256) { const uint32_t q = (uint32_t)(src.ent(i) >> 48); v0 += q == cpid[0]; v1 += q == cpid[1]; v2 += q == cpid[2]; }
            if (v0) atomicAdd(&s_votes[0], v0);
            if (v1) atomicAdd(&s_votes[1], v1);
            if (v2) atomicAdd(&s_votes[2], v2);
        }
        for (uint32_t i = tid; i < (m + 31u) / 32u + 70u; i += 256) s_occ[i] = 0;
        __syncthreads();
        const uint32_t best = s_votes[0] >= s_votes[1] ? (s_votes[0] >= s_votes[2] ? 0u : 2u) : (s_votes[1] >= s_votes[2] ? 1u : 2u);
        if ((uint64_t)s_votes[best] * 5u < (uint64_t)m * 3u) return false;          // below 60 %: the general replay
        if (s_votes[best] == m) {
            // ONE word only: the closed form (k_lz_emulate_giant has the derivation) — here, where two workgroups share a CU and the
            // clusters come off a cursor, instead of in the 128 KiB kernel behind this one.  anchors: a(0) = first entry,
            // a(k+1) = first entry more than W positions after a(k); an entry finds its anchor, an anchor finds nothing.
            uint32_t *s_anch = s_occ;                                              // (<= block / W + 1 anchors)
            __shared__ uint32_t s_nanch;
            __syncthreads();
            if (tid == 0) {
                uint32_t k = 0, cur = 0;
                s_anch[0] = 0;
                for (;;) {
                    const uint32_t lim = ((uint32_t)(src.ent(cur) >> 16) & 0xFFFFu) + W;
                    uint32_t lo = cur + 1, hi = m;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (((uint32_t)(src.ent(mid) >> 16) & 0xFFFFu) > lim) hi = mid; else lo = mid + 1; }
                    if (lo >= m) break;
                    cur = lo; s_anch[++k] = cur;
                }
                s_nanch = k + 1;
            }
            __syncthreads();
            const uint32_t na = s_nanch;
            for (uint32_t i = tid; i < m; i += 256) {
                uint32_t lo = 0, hi = na - 1;                                       // last anchor <= i
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_anch[mid] <= i) lo = mid; else hi = mid - 1; }
                const uint32_t an = s_anch[lo];
                if (an != i) src.put(i, (uint32_t)(src.ent(i) >> 16) & 0xFFFFu, (uint32_t)(src.ent(an) >> 16) & 0xFFFFu);
            }
            return true;
        }
        const uint32_t X = cpid[best];
        const uint32_t rX = (uint32_t)(src.ent(cidx[best]) >> 32) & 0xFFFFu;
        uint32_t done_upto = m;                                                     // entries whose results were written
        if (dbg && tid == 0) { atomicAdd((unsigned long long *)&dbg[50], 1ull); atomicAdd((unsigned long long *)&dbg[51], (unsigned long long)m); atomicAdd((unsigned long long *)&dbg[52], (unsigned long long)(clock64() - tk0)); tk0 = clock64(); }
        uint32_t n_fast = 0, n_scan = 0, n_fgn = 0, n_bulk = 0, n_bulk1 = 0, n_ret = 0;
        if (tid < 64) {
            uint32_t ev = 0, f_ev = 0, f_n = 0, anchor_pos = 0, stash = 0, stash_of = ~0u;
            bool anchor_ok = false, bailed = false;
            // X always takes the FIRST FREE SLOT at or above rX: that slot is tracked (`nx`, exact) together with a bound `lf`
            // up to which everything above it is known to be free, so that a copy of X is placed without looking at the
            // bitmap: b = nx++.  Only when the free run is used up is the bitmap scanned (next free slot, then the next taken
            // one).  A retirement below nx opens a one-slot run in front of the current one, which is kept aside (`sv_*`) and
            // comes back when that slot has been re-taken — the steady state of a long run (retire one copy, insert one)
            // never scans.  `sv_dirty`: something else was freed in between; then the scan decides.
            constexpr uint32_t BIG = 0x7FFFFFFFu;
            const uint32_t nw_all = (m + 31u) / 32u + 64u;
            uint32_t nx = rX, lf = BIG, sv_nx = 0, sv_lf = 0;
            bool sv = false, sv_dirty = false;
            // bitmap updates are single LDS atomics issued by one lane (no read on the chain; LDS executes a wave's
            // instructions in order, so the scans that follow see them)
            auto bit_clear = [&](uint32_t sl) { if (lane == 0) atomicAnd(&s_occ[sl >> 5], ~(1u << (sl & 31u))); };
            auto bit_set = [&](uint32_t sl) { if (lane == 0) atomicOr(&s_occ[sl >> 5], 1u << (sl & 31u)); };
            auto first_one = [&](uint32_t from) -> uint32_t {
                const volatile uint32_t *occ = s_occ;
                const uint32_t w0 = from >> 5;
                for (uint32_t base = w0; base < nw_all; base += 64u) {
                    const uint32_t wi = base + lane;
                    uint32_t v = wi < nw_all ? occ[wi] : 0u;
                    if (wi == w0) v &= ~((1u << (from & 31u)) - 1u);
                    const uint64_t nz = __ballot(v != 0u);
                    if (nz) { const uint32_t ln = (uint32_t)__builtin_ctzll(nz); return ((base + ln) << 5) + (uint32_t)__builtin_ctz(RLANE(v, ln)); }
                }
                return BIG;
            };
            auto refill = [&]() {                                                     // the free run [nx, lf) is used up
                if (sv && !sv_dirty) { nx = sv_nx; lf = sv_lf; sv = false; }
                else { sv = false; nx = dom_first_zero(s_occ, nx, lane); lf = first_one(nx + 1u); ++n_scan; }
            };
            uint64_t ne = lane < m ? src.ent(lane) : 0ull;
            uint32_t e_pos = (uint32_t)(ne >> 16) & 0xFFFFu, e_pid = (uint32_t)(ne >> 48);   // the retirement stream: entries [ev & ~63, +64)
            uint32_t pe = RLANE(e_pos, 0);
            for (uint32_t i0 = 0; i0 < m && !bailed; i0 += 64) {
                const uint64_t ce = ne;
                if (i0 + 64u + lane < m) ne = src.ent(i0 + 64u + lane);
                const uint32_t c_pos = (uint32_t)(ce >> 16) & 0xFFFFu, c_r = (uint32_t)(ce >> 32) & 0xFFFFu, c_pid = (uint32_t)(ce >> 48);
                const uint32_t lim = (m - i0) < 64u ? (m - i0) : 64u;
                uint32_t out_acc = LZ_NONE16;
                // 64 copies of X in a row, nothing to retire before the last of them, 64 free slots in a row: they take them in
                // order and all find the same thing — one wave-wide step instead of 64 serial ones (every run in the first W
                // positions of a block, where nothing retires at all)
                if (lim == 64u && nx + 64u <= lf && __ballot(c_pid == X) == ~0ull && !(ev < i0 + 63u && pe + W < RLANE(c_pos, 63))
                    && i0 + 64u - ev <= W) {
                    const uint32_t b = nx + lane;
                    uint32_t res = LZ_NONE16;
                    if (c_pid != c_pos) { if (ev == 0) res = c_pid; else if (anchor_ok) res = anchor_pos; }
                    if (nx == rX) {                                                 // the first of them takes X's home
                        if (ev != 0 && !anchor_ok && lane > 0 && c_pid != c_pos) res = RLANE(c_pos, 0);
                        anchor_ok = true; anchor_pos = RLANE(c_pos, 0);
                    }
                    atomicOr(&s_occ[b >> 5], 1u << (b & 31u));
                    ((volatile uint16_t *)s_ring)[(i0 + lane) & rmask] = (uint16_t)b;
                    nx += 64u;
                    if (nx >= lf) refill();
                    n_fast += 64u;
                    if (res != LZ_NONE16) src.put(i0 + lane, c_pos, res);
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                for (uint32_t t = 0; t < lim; ) {
                    const uint32_t i = i0 + t;
                    const uint32_t p = RLANE(c_pos, t), r = RLANE(c_r, t), id = RLANE(c_pid, t);
                    // The STEADY STATE of a long run of X — consecutive positions, each retiring exactly the oldest live entry, which
                    // is itself a copy of X from a run of consecutive positions — goes k entries at a time: the slot a retirement
                    // opens lies below the first free slot, so the new copy takes exactly that slot (the serial code below: save the
                    // free run, take the slot, restore the free run) — the bitmap and the free run do not change, the slots move
                    // down the ring k places, and find() changes only where the retired slot is X's home (the new copy becomes
                    // the anchor).  One wave-wide step for up to 64 entries instead of ~700 cycles each.
                    // ... and the stretches in which nothing retires (the window's far edge lies in other data): k copies of X take the
                    // next k free slots and all find the same thing — the partial form of the 64-entry step above
                    if (id == X && !(ev < i && pe + W < p)) {
                        const uint32_t u = lane - t;
                        const uint32_t room = lf - nx;
                        const bool ok = lane >= t && lane < lim && c_pid == X && !(ev < i0 + lane && pe + W < c_pos) && u < room && (i0 + lane - ev) < W;
                        const uint64_t okm = __ballot(ok) >> t;
                        const uint32_t k = (~okm) ? (uint32_t)__builtin_ctzll(~okm) : 64u;
                        if (k >= 2u) {
                            uint32_t res = LZ_NONE16;
                            if (c_pid != c_pos) { if (ev == 0) res = c_pid; else if (anchor_ok) res = anchor_pos; }
                            if (nx == rX) {                                         // the first of them takes X's home
                                if (ev != 0 && !anchor_ok && u > 0 && c_pid != c_pos) res = p;
                                anchor_ok = true; anchor_pos = p;
                            }
                            if (lane >= t && u < k) {
                                const uint32_t b = nx + u;
                                atomicOr(&s_occ[b >> 5], 1u << (b & 31u));
                                ((volatile uint16_t *)s_ring)[(i0 + lane) & rmask] = (uint16_t)b;
                                out_acc = res;
                            }
                            nx += k;
                            if (nx >= lf) refill();
                            n_fast += k; n_bulk += k; t += k;
                            __builtin_amdgcn_wave_barrier();
                            continue;
                        }
                    }
                    if (id == X && ev > 0 && ev < i && pe + W + 1u == p && !sv) {
                        const uint32_t u = lane - t, eoff = ev & 63u;
                        const uint32_t maxk = (lim - t) < (64u - eoff) ? (lim - t) : (64u - eoff);
                        const bool in = lane >= t && u < maxk;
                        const uint32_t rl = (eoff + u) & 63u;
                        const uint32_t rp = (uint32_t)__shfl((int)e_pos, (int)rl), rid = (uint32_t)__shfl((int)e_pid, (int)rl);
                        const bool alias = (i - ev - 1u) == W;                      // the ring is full: entry i + u takes the place of entry ev + u + 1
                        uint32_t su = in ? (uint32_t)((const volatile uint16_t *)s_ring)[(ev + u) & rmask] : 0u;
                        if (alias && lane == t) su = stash;
                        // two regimes: the retired slot lies BELOW the first free one (the new copy takes it) or ABOVE the free run
                        // (occupied, so not inside it: the new copy takes the free run's next slot) — a stretch is one or the other
                        const bool below = RLANE(su, t) < nx;
                        const uint32_t room = lf - nx;
                        const bool ok = in && c_pid == X && c_pos == p + u && rid == X && rp == pe + u && su >= rX &&
                                        (below ? su < nx : (su >= nx && u < room));
                        const uint64_t okm = __ballot(ok) >> t;
                        const uint32_t k = (~okm) ? (uint32_t)__builtin_ctzll(~okm) : 64u;
                        if (k >= 2u && (!alias || stash_of == ev)) {
                            uint32_t new_stash = 0;
                            if (alias) new_stash = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const volatile uint16_t *)s_ring)[(ev + k) & rmask]);
                            const bool mine = lane >= t && u < k;
                            if (below) {
                                const uint64_t hit = __ballot(mine && su == rX);
                                const uint32_t ts = hit ? (uint32_t)__builtin_ctzll(hit) : 64u;     // the entry that retires the anchor and takes its place
                                if (mine) {
                                    uint32_t res = LZ_NONE16;
                                    if (lane < ts) { if (anchor_ok) res = anchor_pos; }
                                    else if (lane > ts) res = p + (ts - t);
                                    if (c_pid == c_pos) res = LZ_NONE16;                // (a word's first occurrence finds nothing, ever)
                                    ((volatile uint16_t *)s_ring)[(i0 + lane) & rmask] = (uint16_t)su;
                                    out_acc = res;
                                }
                                if (hit) { anchor_ok = true; anchor_pos = p + (ts - t); }
                            } else {
                                uint32_t res = LZ_NONE16;
                                if (c_pid != c_pos && anchor_ok) res = anchor_pos;
                                if (nx == rX) {                                     // the first of them takes X's (free) home
                                    if (!anchor_ok && u > 0 && c_pid != c_pos) res = p;
                                    anchor_ok = true; anchor_pos = p;
                                }
                                const uint32_t b = nx + u;
                                uint32_t *const wc = s_occ + (mine ? su >> 5 : 0u), *const ws = s_occ + (mine ? b >> 5 : 0u);
                                __hip_atomic_fetch_and(wc, mine ? ~(1u << (su & 31u)) : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_or(ws, mine ? 1u << (b & 31u) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (mine) {
                                    ((volatile uint16_t *)s_ring)[(i0 + lane) & rmask] = (uint16_t)b;
                                    out_acc = res;
                                }
                                nx += k;
                            }
                            if (alias) { stash = new_stash; stash_of = ev + k; }
                            n_bulk1 += k;
                            ev += k;
                            if ((ev & 63u) == 0) { const uint32_t q = ev + lane; const uint64_t x = q < m ? src.ent(q) : 0ull; e_pos = (uint32_t)(x >> 16) & 0xFFFFu; e_pid = (uint32_t)(x >> 48); }
                            pe = RLANE(e_pos, ev & 63u);
                            n_fast += k; n_bulk += k; t += k;
                            __builtin_amdgcn_wave_barrier();
                            if (nx >= lf) refill();                                 // (here, not inside the branch above: the backend trips over it there)
                            continue;
                        }
                    }
                    // catch-up after a gap (the window's far edge ran through a whole old run while the cluster had no entry): up to 64
                    // retirements in one step — their slots' bits cleared by the lanes, then ONE scan for the first free slot
                    if (ev + 4u <= i && pe + W < p) {
                        const uint32_t eoff = ev & 63u, u = lane;
                        const uint32_t rl = (eoff + u) & 63u;
                        const uint32_t rp = (uint32_t)__shfl((int)e_pos, (int)rl), rid = (uint32_t)__shfl((int)e_pid, (int)rl);
                        const bool ok = u < 64u - eoff && ev + u < i && rp + W < p;
                        const uint64_t okm = __ballot(ok);
                        const uint32_t kr = (~okm) ? (uint32_t)__builtin_ctzll(~okm) : 64u;      // positions ascend: a prefix
                        if (kr >= 4u) {
                            const bool mine = u < kr;
                            uint32_t su = mine ? (uint32_t)((const volatile uint16_t *)s_ring)[(ev + u) & rmask] : 0u;
                            if (u == 0 && ev == stash_of) su = stash;
                            if (mine) atomicAnd(&s_occ[su >> 5], ~(1u << (su & 31u)));
                            f_ev += (uint32_t)__popcll(__ballot(mine && rid != X));
                            if (__ballot(mine && rid == X && su == rX)) anchor_ok = false;
                            if (__ballot(mine && su >= rX)) { sv = false; nx = dom_first_zero(s_occ, rX, lane); lf = first_one(nx + 1u); ++n_scan; }
                            ev += kr; n_ret += kr;
                            if ((ev & 63u) == 0) { const uint32_t q = ev + lane; const uint64_t x = q < m ? src.ent(q) : 0ull; e_pos = (uint32_t)(x >> 16) & 0xFFFFu; e_pid = (uint32_t)(x >> 48); }
                            pe = RLANE(e_pos, ev & 63u);
                            __builtin_amdgcn_wave_barrier();
                            continue;                                               // (the same entry again: more to retire, or its turn)
                        }
                    }
                    while (ev < i && pe + W < p) {                                  // FIFO retirement (lz77.c:70-76): clears the bucket
                        const uint32_t sl = ev == stash_of ? stash : (uint32_t)__builtin_amdgcn_readfirstlane((int)((const volatile uint16_t *)s_ring)[ev & rmask]);
                        bit_clear(sl); ++n_ret;
                        if (RLANE(e_pid, ev & 63u) != X) ++f_ev;                    // foreign entries leave their FIFO in order
                        else if (sl == rX) anchor_ok = false;
                        if (sl >= rX) {
                            if (sl < nx) {
                                if (sl + 1u == nx) nx = sl;                         // the free run grows downwards
                                else {
                                    if (!sv) { sv = true; sv_nx = nx; sv_lf = lf; sv_dirty = false; } else sv_dirty = true;
                                    nx = sl; lf = sl + 1u;
                                }
                            } else if (sv && sl < sv_nx) sv_dirty = true;
                        }
                        ++ev;
                        if ((ev & 63u) == 0) { const uint32_t q = ev + lane; const uint64_t x = q < m ? src.ent(q) : 0ull; e_pos = (uint32_t)(x >> 16) & 0xFFFFu; e_pid = (uint32_t)(x >> 48); }
                        pe = RLANE(e_pos, ev & 63u);
                        __builtin_amdgcn_wave_barrier();
                    }
                    const bool isx = id == X;
                    uint32_t res = LZ_NONE16;
                    if (ev == 0) {
                        if (id != p) res = id;                                      // nothing retired yet: the first occurrence
                    } else if (id != p) {                                           // (a word's first occurrence finds nothing, ever)
                        if (isx) { if (anchor_ok) res = anchor_pos; }               // rX holds a copy of X, or nothing
                        else if ((((const volatile uint32_t *)s_occ)[r >> 5] >> (r & 31u)) & 1u) {
                            const uint32_t e = dom_first_zero(s_occ, r, lane);
                            uint32_t key = ~0u;
                            for (uint32_t k0 = f_ev; k0 < f_n; k0 += 64u) {
                                const uint32_t k = k0 + lane;
                                if (k < f_n) {
                                    const uint32_t x = k % DOM_FCAP;
                                    const uint32_t fs = ((const volatile uint16_t *)s_fslot)[x];
                                    if (((const volatile uint16_t *)s_fid)[x] == id && fs >= r && fs < e) { const uint32_t c = (fs << 16) | ((const volatile uint16_t *)s_fpos)[x]; key = c < key ? c : key; }
                                }
                            }
#pragma unroll
                            for (int o = 32; o >= 1; o >>= 1) { const uint32_t c = __shfl_xor(key, o); key = c < key ? c : key; }
                            if (key != ~0u) res = key & 0xFFFFu;
                        }
                    }
                    // insert: first fit from the home
                    uint32_t b;
                    if (isx) {
                        b = nx; bit_set(b); ++nx; ++n_fast;
                        if (nx >= lf) refill();
                        if (b == rX) { anchor_ok = true; anchor_pos = p; }
                    } else {
                        b = dom_first_zero(s_occ, r, lane);
                        bit_set(b);
                        if (b == rX || f_n - f_ev >= DOM_FCAP) { bailed = true; done_upto = i0; break; }
                        if (b >= rX) {                                              // (then b >= nx: nx is the first free slot from rX)
                            if (b == nx) { ++nx; if (nx >= lf) refill(); }
                            else if (b < lf) lf = b;
                            else if (sv) {
                                if (b == sv_nx) { ++sv_nx; if (sv_nx >= sv_lf) sv_dirty = true; }
                                else if (b > sv_nx && b < sv_lf) sv_lf = b;
                            }
                        }
                        const uint32_t x = f_n % DOM_FCAP;
                        if (lane == 0) { ((volatile uint16_t *)s_fid)[x] = (uint16_t)id; ((volatile uint16_t *)s_fpos)[x] = (uint16_t)p; ((volatile uint16_t *)s_fslot)[x] = (uint16_t)b; }
                        ++f_n; ++n_fgn;
                    }
                    // the ring holds W slots, but W + 1 entries are alive for a moment (insertion k retires k - W AFTER it
                    // has written, lz77.c:70-76): when this entry takes the place of the oldest one, that one's slot moves
                    // to a register (there can only be one such entry: positions are distinct)
                    if (i - ev == W) { stash = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const volatile uint16_t *)s_ring)[ev & rmask]); stash_of = ev; }
                    if (lane == 0) ((volatile uint16_t *)s_ring)[i & rmask] = (uint16_t)b;
                    if (lane == t) out_acc = res;
                    __builtin_amdgcn_wave_barrier();
                    ++t;
                }
                if (!bailed && i0 + lane < m && out_acc != LZ_NONE16) src.put(i0 + lane, c_pos, out_acc);
            }
            if (lane == 0) s_result = bailed ? (1u | (done_upto << 1)) : 0u;
            if (dbg && lane == 0) { atomicAdd((unsigned long long *)&dbg[53], (unsigned long long)(clock64() - tk0)); atomicAdd((unsigned long long *)&dbg[54], (unsigned long long)n_fast);
                                    atomicAdd((unsigned long long *)&dbg[55], (unsigned long long)n_scan); atomicAdd((unsigned long long *)&dbg[56], (unsigned long long)n_fgn);
                                    if (bailed) atomicAdd((unsigned long long *)&dbg[57], 1ull);
                                    atomicAdd((unsigned long long *)&dbg[61], (unsigned long long)n_bulk);
                                    atomicAdd((unsigned long long *)&dbg[62], (unsigned long long)n_bulk1); atomicAdd((unsigned long long *)&dbg[63], (unsigned long long)n_ret); }
        }
        __syncthreads();
        const uint32_t rs_ = s_result;
        if (rs_ & 1u) {
            // given up: wipe what was written (k_lz_emulate_giant only writes the positions that find something)
            const uint32_t upto = rs_ >> 1;
            for (uint32_t i = tid; i < upto; i += 256) src.put(i, (uint32_t)(src.ent(i) >> 16) & 0xFFFFu, LZ_NONE16);
            return false;
        }
        return true;
}

