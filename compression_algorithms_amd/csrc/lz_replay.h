// lz_replay.h — wave-per-cluster replay shared by k_lz2_big (lz2_find.hip) and the fallback's giant clusters
// (lz_find.hip): occupancy bitmap in registers, first fit by ballot + v_readlane, slot -> occupant and entry -> slot in
// LDS.  Reference functions replaced: insert_hash_table / find, algorithms/lz77/lz77.c:55-108 and
// algorithms/deflate/lz77.c:77-174.
#pragma once
#include "lz_common.h"

#define RLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))

// occupancy bitmap of one cluster held in registers: slot s = bit (s & 31) of word (s >> 5); word w lives in
// lane (w & 63), register (w >> 6).  Every index into w[] is a compile-time constant (unrolled), so the
// array stays in VGPRs; all cross-lane traffic is v_readlane / ballots on wave-uniform indices.
// a dense switch over a wave-uniform register index: every case names its register by a compile-time constant, so the
// array stays in VGPRs and only one register is touched (large bitmaps: a select over all 32 registers per operation
// made every step of huge_replay ~500 instructions)
#define WB_SWITCH(q, OP) switch (q) { case 0: OP(0); break; case 1: OP(1); break; case 2: OP(2); break; case 3: OP(3); break; case 4: OP(4); break; case 5: OP(5); break; case 6: OP(6); break; case 7: OP(7); break; case 8: OP(8); break; case 9: OP(9); break; case 10: OP(10); break; case 11: OP(11); break; case 12: OP(12); break; case 13: OP(13); break; case 14: OP(14); break; case 15: OP(15); break; case 16: OP(16); break; case 17: OP(17); break; case 18: OP(18); break; case 19: OP(19); break; case 20: OP(20); break; case 21: OP(21); break; case 22: OP(22); break; case 23: OP(23); break; case 24: OP(24); break; case 25: OP(25); break; case 26: OP(26); break; case 27: OP(27); break; case 28: OP(28); break; case 29: OP(29); break; case 30: OP(30); break; case 31: OP(31); break; default: break; }

template <int NW>
struct WaveBitmap {
    uint32_t w[NW];
    uint32_t full;                // NW > 4 only: bit q = register q has no zero bit in any lane (wave-uniform)
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int q = 0; q < NW; ++q) w[q] = 0;
        full = 0;
    }
    template <int K> __device__ __forceinline__ uint32_t reg_lane(uint32_t ln) const { if constexpr (K < NW) return RLANE(w[K], ln); else return 0u; }
    template <int K> __device__ __forceinline__ void reg_and(uint32_t keep) { if constexpr (K < NW) w[K] &= keep; }
    template <int K> __device__ __forceinline__ void reg_set(uint32_t bit) {
        if constexpr (K < NW) { w[K] |= bit; if (__ballot(w[K] != 0xFFFFFFFFu) == 0ull) full |= 1u << K; }
    }
    template <int K> __device__ __forceinline__ uint32_t reg_first_zero(uint32_t rw, uint32_t lowmask, uint32_t lane) const {
        if constexpr (K < NW) {
            const uint32_t wi = (uint32_t)K * 64u + lane;
            uint32_t v = w[K];
            if (wi < rw) v = 0xFFFFFFFFu; else if (wi == rw) v |= lowmask;
            const uint64_t nz = __ballot(v != 0xFFFFFFFFu);
            if (!nz) return ~0u;
            const uint32_t ln = (uint32_t)__builtin_ctzll(nz);
            const uint32_t mv = RLANE(v, ln);
            return (((uint32_t)K * 64u + ln) << 5) + (uint32_t)__builtin_ctz(~mv);
        } else return ~0u;
    }
    __device__ __forceinline__ uint32_t word(uint32_t wi) const {
        const uint32_t wq = wi >> 6, ln = wi & 63u;
        uint32_t r = 0;
        if constexpr (NW > 4) {
#define OP(K) r = reg_lane<K>(ln)
            WB_SWITCH(wq, OP)
#undef OP
        } else {
#pragma unroll
            for (int q = 0; q < NW; ++q) { const uint32_t t = RLANE(w[q], ln); if (NW == 1 || (uint32_t)q == wq) r = t; }
        }
        return r;
    }
    __device__ __forceinline__ bool test(uint32_t slot) const { return (word(slot >> 5) >> (slot & 31u)) & 1u; }
    __device__ __forceinline__ void clear_bit(uint32_t slot, uint32_t lane) {      // whoever sits there, or nobody
        const uint32_t wi = slot >> 5, wq = wi >> 6;
        const uint32_t keep = (lane == (wi & 63u)) ? ~(1u << (slot & 31u)) : 0xFFFFFFFFu;
        if constexpr (NW > 4) {
#define OP(K) reg_and<K>(keep)
            WB_SWITCH(wq, OP)
#undef OP
            full &= ~(1u << wq);
        } else {
#pragma unroll
            for (int q = 0; q < NW; ++q) w[q] &= (NW == 1 || (uint32_t)q == wq) ? keep : 0xFFFFFFFFu;
        }
    }
    __device__ __forceinline__ void flip(uint32_t slot, uint32_t lane) {           // SETS the bit of a free slot
        const uint32_t wi = slot >> 5, wq = wi >> 6;
        const uint32_t bit = (lane == (wi & 63u)) ? (1u << (slot & 31u)) : 0u;
        if constexpr (NW > 4) {
#define OP(K) reg_set<K>(bit)
            WB_SWITCH(wq, OP)
#undef OP
        } else {
#pragma unroll
            for (int q = 0; q < NW; ++q) w[q] ^= (NW == 1 || (uint32_t)q == wq) ? bit : 0u;
        }
    }
    __device__ __forceinline__ uint32_t first_zero_from(uint32_t r, uint32_t lane) const {
        const uint32_t rw = r >> 5, lowmask = (1u << (r & 31u)) - 1u;
        uint32_t res = ~0u;
        if constexpr (NW > 4) {
            uint32_t q = rw >> 6;
            while (q < (uint32_t)NW) {
#define OP(K) res = reg_first_zero<K>(rw, lowmask, lane)
                WB_SWITCH(q, OP)
#undef OP
                if (res != ~0u) break;
                const uint32_t rest = (q + 1u < 32u) ? (~full >> (q + 1u)) : 0u;       // next register that is not full
                if (!rest) break;
                q = q + 1u + (uint32_t)__builtin_ctz(rest);
            }
            return res;
        }
        bool found = false;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const uint32_t wi = (uint32_t)q * 64u + lane;
            uint32_t v = w[q];
            if (wi < rw) v = 0xFFFFFFFFu; else if (wi == rw) v |= lowmask;
            const uint64_t nz = __ballot(v != 0xFFFFFFFFu);
            if (!found && nz) {
                const uint32_t ln = (uint32_t)__builtin_ctzll(nz);
                const uint32_t mv = RLANE(v, ln);
                res = (((uint32_t)q * 64u + ln) << 5) + (uint32_t)__builtin_ctz(~mv);
                found = true;
            }
        }
        return res;
    }
};


// one cluster, one wave.  PLAIN: the cluster does not cover bucket 0 / T (all but one per block): the spurious clear and
// the non-wrapping find() drop out of the loop, which is bound by the number of scalar instructions per step.
template <int LDS_ENTRIES, int NW, bool PLAIN>
__device__ __forceinline__ void big_replay(uint32_t *s_occ, uint16_t *s_slot, uint32_t lane, uint32_t W, uint32_t n,
                                           uint32_t d_anom, uint32_t d_limit, const uint16_t *bp, const uint16_t *br,
                                           const uint16_t *bi, uint16_t *bc)
{
    WaveBitmap<NW> bm;
    bm.clear();
    uint32_t ev = 0;
    bool anom_pending = !PLAIN && d_anom != ~0u;
    uint32_t c_pos = 0, c_rs = 0, c_pid = 0, out_acc = 0;
    // position of the oldest entry still in the table, kept in a scalar: "nothing to retire" is one compare per step
    uint32_t ev_pos = lane < n ? bp[lane] : 0u;
    uint32_t pe = RLANE(ev_pos, 0);
    uint32_t n_pos = ev_pos, n_rs = 0, n_pid = 0;                   // the next 64 entries are in flight while these are replayed
    if (lane < n) { n_rs = br[lane]; n_pid = bi[lane]; }
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t ii = i0 + lane;
        c_pos = n_pos; c_rs = n_rs; c_pid = n_pid;
        if (ii + 64 < n) { n_pos = bp[ii + 64]; n_rs = br[ii + 64]; n_pid = bi[ii + 64]; }
        const uint32_t lim = (n - i0) < 64u ? (n - i0) : 64u;
        for (uint32_t t = 0; t < lim; ++t) {
            const uint32_t i = i0 + t;
            const uint32_t p = RLANE(c_pos, t), r = RLANE(c_rs, t), id = RLANE(c_pid, t);
            while (ev < i && pe + W < p) {                          // FIFO retirement (lz77.c:70-76)
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_slot[ev]);
                bm.clear_bit(sl, lane);                             // clears the bucket, whoever sits there
                ++ev;
                if ((ev & 63u) == 0) { const uint32_t q = ev + lane; ev_pos = q < n ? bp[q] : 0u; }
                pe = RLANE(ev_pos, ev & 63u);
            }
            if (!PLAIN && anom_pending && p > W - 1u) { bm.clear_bit(d_anom, lane); anom_pending = false; }
            uint32_t res = LZ_NONE16;
            if (PLAIN && ev == 0) {
                // nothing evicted yet: find() = the word's first occurrence = the word id (k_lz2_find, the sweep)
                if (id != p) res = id;
            } else if (id != p) {                                       // (the first occurrence of a word in the block finds nothing, ever)
                const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_occ[r]);
                if (bm.test(r)) {
                    if ((h & 0xFFFFu) == id) res = h >> 16;
                    else {
                        // The home holds another word: find() walks on to the first copy of this word or the first empty
                        // bucket.  Inside a long single-byte run that walk is thousands of buckets for every foreign word whose
                        // home lies in the run (the reference pays it too: up to 670 probes per call on indented source);
                        // here the end of the walk is one bitmap query and 64 occupants are compared per step.
                        uint32_t e = bm.first_zero_from(r + 1u, lane);
                        if (e > (uint32_t)LDS_ENTRIES) e = (uint32_t)LDS_ENTRIES;
                        if (!PLAIN && r < d_limit && d_limit < e) e = d_limit;          // deflate's find() does not wrap past bucket T
                        for (uint32_t b0 = r + 1u; b0 < e; b0 += 64u) {
                            const uint32_t b = b0 + lane;
                            const uint32_t o = b < e ? s_occ[b] : 0u;
                            const uint64_t hit = __ballot(b < e && (o & 0xFFFFu) == id);
                            if (hit) { res = RLANE(o, (uint32_t)__builtin_ctzll(hit)) >> 16; break; }
                        }
                    }
                }
            }
            const uint32_t b = bm.first_zero_from(r, lane);         // insert: first fit
            bm.flip(b, lane);
            // every lane stores the same value to the same address: no exec-mask juggling (scalar instructions) for a
            // one-lane store, and identical-address stores of a wave do not conflict
            s_occ[b] = id | (p << 16); s_slot[i] = (uint16_t)b;
            if (lane == t) out_acc = res;
            __builtin_amdgcn_wave_barrier();
        }
        if (ii < n) bc[ii] = (uint16_t)out_acc;
    }
    __builtin_amdgcn_wave_barrier();
}


// ---------------------------------------------------------------------------------------------------------------------
// The same replay for a cluster of up to 65 536 entries (a block that is mostly one byte value, with other data mixed in):
// LDS only holds the word id of every slot's occupant (2 B x 65 536); the occupants' positions and the entries' slots
// live in HBM (read past the L1: this wave wrote them), the bitmap in 32 registers per lane.  A find() that hits reads one
// position from HBM — except on the slot that was last taken as its entry's own home slot (the dominant word's anchor),
// which is kept in a register pair.
// ---------------------------------------------------------------------------------------------------------------------
template <bool PLAIN>
__device__ __forceinline__ void huge_replay(uint16_t *s_oid, uint32_t lane, uint32_t W, uint32_t n, uint32_t d_anom, uint32_t d_limit,
                                            const uint16_t *bp, const uint16_t *br, const uint16_t *bi, uint16_t *bc,
                                            uint16_t *gslot, uint16_t *gopos)
{
    constexpr uint32_t CAPS = 65536u;
    WaveBitmap<32> bm;
    bm.clear();
    uint32_t ev = 0;
    bool anom_pending = !PLAIN && d_anom != ~0u;
    uint32_t c_pos = 0, c_rs = 0, c_pid = 0, out_acc = 0;
    uint32_t ev_pos = lane < n ? bp[lane] : 0u;
    uint32_t pe = RLANE(ev_pos, 0);
    uint32_t ev_slot = 0, ev_slot_base = ~0u, ev_slot_upto = 0;      // slots of entries [base, base + 64), valid for entries < upto
    // two cached (slot, position of its occupant) pairs: the slot last taken as its entry's own home slot, and the slot
    // last read from HBM — the dominant word's anchor must survive the foreign words' inserts and hits between its uses
    uint32_t an_slot = ~0u, an_pos = 0, cb_slot = ~0u, cb_pos = 0;
    uint32_t n_pos = ev_pos, n_rs = 0, n_pid = 0;
    if (lane < n) { n_rs = br[lane]; n_pid = bi[lane]; }
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t ii = i0 + lane;
        c_pos = n_pos; c_rs = n_rs; c_pid = n_pid;
        if (ii + 64 < n) { n_pos = bp[ii + 64]; n_rs = br[ii + 64]; n_pid = bi[ii + 64]; }
        const uint32_t lim = (n - i0) < 64u ? (n - i0) : 64u;
        for (uint32_t t = 0; t < lim; ++t) {
            const uint32_t i = i0 + t;
            const uint32_t p = RLANE(c_pos, t), r = RLANE(c_rs, t), id = RLANE(c_pid, t);
            while (ev < i && pe + W < p) {                          // FIFO retirement (lz77.c:70-76)
                if ((ev & ~63u) != ev_slot_base || ev >= ev_slot_upto) {
                    __threadfence();                                // every entry < i has stored its slot
                    ev_slot_base = ev & ~63u;
                    const uint32_t q = ev_slot_base + lane;
                    ev_slot = q < i ? (uint32_t)__builtin_nontemporal_load(gslot + q) : 0u;
                    ev_slot_upto = i;
                }
                const uint32_t sl = RLANE(ev_slot, ev & 63u);
                bm.clear_bit(sl, lane);                             // clears the bucket, whoever sits there
                ++ev;
                if ((ev & 63u) == 0) { const uint32_t q = ev + lane; ev_pos = q < n ? bp[q] : 0u; }
                pe = RLANE(ev_pos, ev & 63u);
            }
            if (!PLAIN && anom_pending && p > W - 1u) { bm.clear_bit(d_anom, lane); anom_pending = false; }
            uint32_t res = LZ_NONE16;
            if (PLAIN && ev == 0) {
                if (id != p) res = id;                              // nothing evicted yet: the first occurrence
            } else if (id != p) {                                       // (the first occurrence of a word in the block finds nothing, ever)
                const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_oid[r]);
                if (bm.test(r)) {
                    uint32_t hit = ~0u;
                    if (h == id) hit = r;
                    else {
                        uint32_t e = bm.first_zero_from(r + 1u, lane);
                        if (e > CAPS) e = CAPS;
                        if (!PLAIN && r < d_limit && d_limit < e) e = d_limit;
                        for (uint32_t b0 = r + 1u; b0 < e; b0 += 64u) {
                            const uint32_t b = b0 + lane;
                            const uint64_t mk = __ballot(b < e && (uint32_t)s_oid[b < e ? b : r] == id);
                            if (mk) { hit = b0 + (uint32_t)__builtin_ctzll(mk); break; }
                        }
                    }
                    if (hit != ~0u) {
                        if (hit == an_slot) res = an_pos;
                        else if (hit == cb_slot) res = cb_pos;
                        else {
                            __threadfence();
                            res = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_nontemporal_load(gopos + hit));
                            cb_slot = hit; cb_pos = res;
                        }
                    }
                }
            }
            const uint32_t b = bm.first_zero_from(r, lane);         // insert: first fit
            bm.flip(b, lane);
            if (b < CAPS) {
                s_oid[b] = (uint16_t)id;
                if (lane == 0) { __builtin_nontemporal_store((uint16_t)p, gopos + b); __builtin_nontemporal_store((uint16_t)b, gslot + i); }
            }
            if (b == cb_slot) cb_pos = p;                           // the cached slots follow their new occupants
            if (b == r) { if (an_slot != ~0u && an_slot != b && cb_slot == ~0u) { cb_slot = an_slot; cb_pos = an_pos; } an_slot = b; an_pos = p; }
            else if (b == an_slot) an_pos = p;
            if (lane == t) out_acc = res;
            __builtin_amdgcn_wave_barrier();
        }
        if (ii < n) bc[ii] = (uint16_t)out_acc;
    }
    __builtin_amdgcn_wave_barrier();
}
