// fse.hip — block-parallel FSE / tANS on gfx950.
//
// The reference's algorithms/fse/src/main.zig is an unfinished sketch (it does not compile);
// what it specifies is kept, what it leaves open is defined in DESIGN.md and oracle/orc_fse.c:
//   histogram            main.zig:88-96
//   normalisation        main.zig:106-149   f64 scale, trunc, min 1, remainder to the first maximum
//   cumulative offsets   main.zig:159-166
//   transition           main.zig:177       next = (state >> bits) + offset   (spread 0: exactly this)
//   reverse encode       main.zig:58-62     last symbol first
//   state flush          main.zig:65
//   LSB-first bit append main.zig:28-39
//
// One wave per block.  The block is read twice from HBM/L2 (histogram pass, coalesced 16 B per
// lane; encode pass, each lane streaming its own contiguous sub-stream with 16-byte loads, the next
// one in flight), the tables (3N bytes: 768 B at table_log 8, 12 KiB at 12) live in dynamic LDS, so
// ~20 waves per CU hide the table-lookup latency of the serial state chain.  The encode loop runs
// ONCE: the chain is serial, so a dry run that only counts bits costs as much as the coding.  Every
// lane writes its words at a fixed stride inside the block's record (the worst case, which is what
// the record is sized for) and the wave then closes the gaps in place, left to right.
//
// Why the in-place compaction needs no fence: sub-stream l moves DOWN, from l * lstr to its prefix
// offset woff[l], and woff[l] + cnt[l] = woff[l+1] <= (l+1) * lstr, so a store for sub-stream l never
// lands on a word that a LATER load (of l itself or of any l' > l) still has to read; the only loads
// a store can overlap are ones of the same or an earlier copy step, and the store carries their data
// (v = payload[src]; payload[dst] = v), so it cannot issue before they have returned.
//
// Staging the sub-stream words through LDS instead was not done: the worst case
// is 64 lanes x 257 words = 64 KiB per wave, i.e. two waves per CU instead of twenty — the state
// chain needs the occupancy more than the ~36 KiB per block of L2 write traffic it would save.
#include "common.h"

#define FSE_MAX_LOG   12
#define FSE_MAX_N     (1 << FSE_MAX_LOG)
#define FSE_MAX_S     64

struct FseP { uint32_t L, S, spread, block; };

__device__ __forceinline__ uint32_t fse_sub_len(uint32_t n, uint32_t S) { uint32_t m = (n + S - 1) / S; return (m + 3u) & ~3u; }


// normalisation on one wave: lane l owns symbols 4l..4l+3
__device__ __forceinline__ void fse_normalise_wave(const uint32_t *s_freq, uint32_t L, uint32_t *s_cnt)
{
    const uint32_t lane = threadIdx.x & 63u, N = 1u << L;
    uint64_t total = 0; uint32_t nsym = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t f = s_freq[lane * 4 + k]; total += f; nsym += f != 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { total += __shfl_xor(total, o); nsym += __shfl_xor(nsym, o); }
    uint32_t g[4] = {0, 0, 0, 0};
    uint32_t sum = 0;
    if (total) {
        const double scale = (double)(N - nsym) / (double)total;          // main.zig:120-121
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t f = s_freq[lane * 4 + k];
            if (f) {
                uint64_t v = (uint64_t)((double)f * scale);                // trunc, main.zig:127-129
                if (v == 0) v = 1;
                g[k] = (uint32_t)v; sum += g[k];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    // remainder -> the first index holding the maximum (main.zig:135-148 re-finds the same one every time)
    uint32_t best = 0, besti = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (g[k] > best) { best = g[k]; besti = lane * 4 + k; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ob = __shfl_xor(best, o), oi = __shfl_xor(besti, o);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const uint32_t rem = total ? N - sum : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_cnt[lane * 4 + k] = g[k] + ((total && lane * 4 + k == besti) ? rem : 0u);
}

// tables shared by encoder and decoder.  s_symat[u] = symbol at table position u;
// s_next[cum[s] + (y - cnt[s])] = N + position of sub-state y of symbol s (positions ascending).
__device__ __forceinline__ void fse_build_tables(const uint32_t *s_cnt, uint32_t L, uint32_t spread, uint32_t *s_cum,
                                                 uint8_t *s_symat, uint16_t *s_next, uint32_t *s_fill)
{
    const uint32_t lane = threadIdx.x & 63u, N = 1u << L, step = (N >> 1) + (N >> 3) + 3u;
    // exclusive prefix of counts in symbol order
    uint32_t c[4], t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = s_cnt[lane * 4 + k]; t += c[k]; }
    uint32_t inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += v; }
    uint32_t run = inc - t;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_cum[lane * 4 + k] = run; s_fill[lane * 4 + k] = run; run += c[k]; }
    __builtin_amdgcn_wave_barrier();
    // slots -> positions: slot k (k-th normalised occurrence in symbol order) sits at (k*step) mod N
    for (uint32_t k = lane; k < N; k += 64) {
        // symbol of slot k: the s with cum[s] <= k < cum[s]+cnt[s] (binary search over 256)
        uint32_t lo = 0, hi = 255;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_cum[mid] <= k) lo = mid; else hi = mid - 1; }
        while (s_cnt[lo] == 0) --lo;                                    // skip absent symbols sharing the offset
        s_symat[spread ? ((k * step) & (N - 1)) : k] = (uint8_t)lo;
    }
    __builtin_amdgcn_wave_barrier();
    // positions in ascending order, stable per symbol: rank the 64 lanes of a step with ballots
    for (uint32_t u0 = 0; u0 < N; u0 += 64) {
        const uint32_t u = u0 + lane;
        const uint32_t s = s_symat[u];
        uint64_t mask = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (s >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull)), cnt = __popcll(mask);
        const int leader = __ffsll((unsigned long long)mask) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) { base = s_fill[s]; s_fill[s] = base + cnt; }
        base = __shfl(base, leader);
        s_next[base + rank] = (uint16_t)(N + u);
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(64)
void k_fse_encode(const uint8_t *__restrict__ in, uint64_t n_total, FseP P, uint8_t *__restrict__ rec_out,
                  uint64_t rec_stride, uint64_t *__restrict__ rec_bits)
{
    // LDS: 6 KiB static + 3N dynamic (N = table size), so table_log 8 runs ~20 waves per CU
    __shared__ uint32_t s_work[4 * 256];                  // 4 sub-histograms, then fill/thresh/delta/nb_hi
    __shared__ uint32_t s_cnt[256], s_cum[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint32_t (*s_hist)[256] = reinterpret_cast<uint32_t (*)[256]>(s_work);
    uint32_t *s_fill = s_work + 256;                      // (the histogram is dead once s_cnt exists)
    uint32_t *s_thresh = s_work + 512;
    int32_t  *s_delta = reinterpret_cast<int32_t *>(s_work + 768);
    uint8_t  *s_nbhi = reinterpret_cast<uint8_t *>(s_work);          // first 256 bytes
    uint16_t *s_next = reinterpret_cast<uint16_t *>(s_dyn);          // [N]
    uint8_t  *s_symat = s_dyn + 2 * (1u << P.L);                     // [N]

    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const uint64_t off = b * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint8_t *src = in + off;
    const uint32_t L = P.L, N = 1u << L, S = P.S;

    for (uint32_t i = lane; i < 4 * 256; i += 64) s_work[i] = 0;
    __builtin_amdgcn_wave_barrier();
    // ---- histogram (main.zig:88-96): 16 B per lane, four interleaved sub-histograms
    const bool vec_ok = (((uintptr_t)src) & 15u) == 0;
    for (uint32_t i = lane * 16u; i < n; i += 64u * 16u) {
        if (vec_ok && i + 16u <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&s_hist[0][w[k] & 0xFF], 1u);
                atomicAdd(&s_hist[1][(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&s_hist[2][(w[k] >> 16) & 0xFF], 1u);
                atomicAdd(&s_hist[3][w[k] >> 24], 1u);
            }
        } else {
            for (uint32_t q = i; q < n && q < i + 16u; ++q) atomicAdd(&s_hist[q & 3u][src[q]], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t s = lane; s < 256; s += 64) s_hist[0][s] += s_hist[1][s] + s_hist[2][s] + s_hist[3][s];
    __builtin_amdgcn_wave_barrier();
    fse_normalise_wave(s_hist[0], L, s_cnt);
    __builtin_amdgcn_wave_barrier();
    fse_build_tables(s_cnt, L, P.spread, s_cum, s_symat, s_next, s_fill);
    for (uint32_t s = lane; s < 256; s += 64) {
        const uint32_t c = s_cnt[s];
        uint32_t nb = 0, th = 0; int32_t dl = 0;
        if (c) { nb = L - (31u - (uint32_t)__builtin_clz(c)); th = c << nb; dl = (int32_t)s_cum[s] - (int32_t)c; }
        s_nbhi[s] = (uint8_t)nb; s_thresh[s] = th; s_delta[s] = dl;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- record header
    uint8_t *rec = rec_out + b * rec_stride;
    uint32_t nsym = 0;
    {
        // bitmap: lane l < 32 writes byte l
        if (lane < 32) {
            uint32_t v = 0;
            for (int k = 0; k < 8; ++k) v |= (uint32_t)(s_cnt[lane * 8 + k] != 0) << k;
            rec[lane] = (uint8_t)v;
        }
        // counts of present symbols in symbol order: rank = number of present symbols before s
        uint32_t pres[4], t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { pres[k] = s_cnt[lane * 4 + k] != 0; t += pres[k]; }
        uint32_t inc = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += v; }
        nsym = __shfl(inc, 63);
        uint32_t r = inc - t;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (pres[k]) {
            const uint32_t c = s_cnt[lane * 4 + k];
            rec[32 + 2 * r] = (uint8_t)c; rec[32 + 2 * r + 1] = (uint8_t)(c >> 8); ++r;
        }
        if ((nsym & 1u) && lane == 0) { rec[32 + 2 * nsym] = 0; rec[32 + 2 * nsym + 1] = 0; }
    }
    const uint32_t hdr = 32u + 2u * (nsym + (nsym & 1u));
    uint8_t *states = rec + hdr;
    const uint32_t states_bytes = 2u * (S + (S & 1u));
    uint8_t *lens = states + states_bytes;
    uint32_t *payload = reinterpret_cast<uint32_t *>(lens + 4u * S);       // records are 4-byte aligned
    if ((S & 1u) && lane == 0) { states[2 * S] = 0; states[2 * S + 1] = 0; }

    // ---- sub-stream of this lane: bytes [a, a+len), encoded last symbol first
    const uint32_t m = fse_sub_len(n, S);
    const uint32_t a = lane * m;
    const uint32_t len = (lane < S && a < n) ? ((n - a < m) ? n - a : m) : 0u;
    // ONE pass over the symbols: the state chain is serial, so a dry run to learn the sub-stream sizes costs as much as the
    // coding itself.  Every lane writes its words at a fixed stride (the worst case, which is what the record is sized
    // for) and the wave then closes the gaps in place, left to right — a coalesced copy of at most 64 KiB.
    const uint32_t lstr = (m * L + 31u) / 32u + 1u;            // words per lane before compaction
    uint32_t x = N;
    uint64_t acc = 0; uint32_t nacc = 0, widx = lane * lstr;
    uint32_t bits = 0;
    auto step = [&](uint32_t s) {
        const uint32_t nb = s_nbhi[s] - (x < s_thresh[s] ? 1u : 0u);
        acc |= (uint64_t)(x & ((1u << nb) - 1u)) << nacc;
        nacc += nb;
        if (nacc >= 32) { payload[widx++] = (uint32_t)acc; acc >>= 32; nacc -= 32; }
        bits += nb;
        x = s_next[(int32_t)(x >> nb) + s_delta[s]];
    };
    if ((len & 15u) == 0 && ((((uintptr_t)(src + a)) & 15u) == 0)) {
        // Every lane walks its own 1 KiB sub-stream, so its loads never coalesce and each costs a trip to HBM that the
        // serial state chain cannot hide: 16 bytes per load, and the next 16 are in flight while these are coded.
        const uint4 *v16 = reinterpret_cast<const uint4 *>(src + a);
        int32_t g = (int32_t)(len >> 4) - 1;
        uint4 cur = make_uint4(0, 0, 0, 0);
        if (g >= 0) cur = v16[g];
        for (; g >= 0; --g) {
            uint4 nxt = make_uint4(0, 0, 0, 0);
            if (g > 0) nxt = v16[g - 1];
            const uint32_t w4[4] = {cur.w, cur.z, cur.y, cur.x};       // backwards: last byte first
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int k = 3; k >= 0; --k) step((w4[q] >> (8 * k)) & 0xFFu);
            }
            cur = nxt;
        }
    } else {
        // walk backwards in 4-byte groups (sub-streams start 4-byte aligned relative to the block)
        for (uint32_t i = len; i > 0;) {
            const uint32_t take = ((i & 3u) ? (i & 3u) : 4u);
            const uint32_t base = i - take;
            uint32_t w = 0;
            if (take == 4 && ((((uintptr_t)(src + a + base)) & 3u) == 0)) w = *reinterpret_cast<const uint32_t *>(src + a + base);
            else for (uint32_t k = 0; k < take; ++k) w |= (uint32_t)src[a + base + k] << (8 * k);
            for (int k = (int)take - 1; k >= 0; --k) step((w >> (8 * k)) & 0xFFu);
            i = base;
        }
    }
    if (nacc) payload[widx] = (uint32_t)acc;
    const uint32_t mybits = bits, final_t = x - N;
    const uint32_t nwords = (bits + 31u) >> 5;
    uint32_t inc = nwords;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += v; }
    const uint32_t word_off = inc - nwords;
    const uint32_t total_words = __shfl(inc, 63);
    if (lane < S) {
        states[2 * lane] = (uint8_t)final_t; states[2 * lane + 1] = (uint8_t)(final_t >> 8);
        lens[4 * lane] = (uint8_t)mybits; lens[4 * lane + 1] = (uint8_t)(mybits >> 8);
        lens[4 * lane + 2] = (uint8_t)(mybits >> 16); lens[4 * lane + 3] = (uint8_t)(mybits >> 24);
    }
    if (lane == 0) rec_bits[b] = 8ull * ((uint64_t)hdr + states_bytes + 4ull * S + 4ull * total_words);
    // close the gaps: sub-stream l moves from l * lstr down to its prefix offset (never up, never past the next source)
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (uint32_t l = 1; l < S; ++l) {
        const uint32_t cnt = __shfl(nwords, l), dsto = __shfl(word_off, l), srco = l * lstr;
        if (dsto == srco) continue;
        for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
            const uint32_t k = k0 + lane;
            uint32_t v = 0;
            if (k < cnt) v = payload[srco + k];
            if (k < cnt) payload[dsto + k] = v;
        }
    }
}

// one wave per block: lane i decodes sub-stream i forwards, reading its bits backwards
__global__ __launch_bounds__(64)
void k_fse_decode(const uint8_t *__restrict__ packed, uint64_t packed_bytes, const uint64_t *__restrict__ offsets, FseP P,
                  uint8_t *__restrict__ out, uint64_t n_total, uint32_t *__restrict__ err)
{
    __shared__ uint32_t s_cnt[256], s_cum[256], s_fill[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // next[N] u16, sub[N] u16, symat[N] u8
    uint16_t *s_next = reinterpret_cast<uint16_t *>(s_dyn);
    uint16_t *s_sub = s_next + (1u << P.L);
    uint8_t  *s_symat = s_dyn + 4 * (1u << P.L);
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const uint64_t off = b * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint32_t L = P.L, N = 1u << L, S = P.S;
    // the record [rb, re) must lie inside the packed buffer and hold at least the fixed part of a header; every field
    // read below is checked against re before it is used (records come from files / peers)
    const uint64_t rb = offsets[b], re = offsets[b + 1];
    if ((rb & 31u) || (re & 31u) || re < rb || re > packed_bytes * 8ull || (re - rb) < 8ull * (32u + 4u)) {
        if (n && lane == 0) atomicOr(err, 1u);
        return;
    }
    const uint32_t rec_bytes = (uint32_t)(((re - rb) >> 3) > 0xFFFFFFFFull ? 0xFFFFFFFFull : ((re - rb) >> 3));
    const uint8_t *rec = packed + (rb >> 3);
    // header: bitmap + counts
    uint32_t pres[4], t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t s = lane * 4 + k; pres[k] = (rec[s >> 3] >> (s & 7u)) & 1u; t += pres[k]; }
    uint32_t inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += v; }
    const uint32_t nsym = __shfl(inc, 63);
    uint32_t r = inc - t, sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t c = 0;
        if (pres[k]) { if (32u + 2u * r + 2u <= rec_bytes) c = rec[32 + 2 * r] | ((uint32_t)rec[32 + 2 * r + 1] << 8); ++r; }
        s_cnt[lane * 4 + k] = c; sum += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    bool bad = (n > 0 && sum != N) || nsym == 0;
    if (n == 0) return;
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    __builtin_amdgcn_wave_barrier();
    fse_build_tables(s_cnt, L, P.spread, s_cum, s_symat, s_next, s_fill);
    // sub-state of every position: y = cnt[s] + rank  <=>  next[cum[s] + rank] = N + u
    for (uint32_t idx = lane; idx < N; idx += 64) {
        // which symbol owns slot idx of `next`: the s with cum[s] <= idx < cum[s] + cnt[s]
        uint32_t lo = 0, hi = 255;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_cum[mid] <= idx) lo = mid; else hi = mid - 1; }
        while (s_cnt[lo] == 0) --lo;
        const uint32_t u = (uint32_t)s_next[idx] - N;
        s_sub[u] = (uint16_t)(s_cnt[lo] + (idx - s_cum[lo]));
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t hdr = 32u + 2u * (nsym + (nsym & 1u));
    const uint32_t fixed = hdr + 2u * (S + (S & 1u)) + 4u * S;                 // bytes before the payload words
    if (fixed > rec_bytes) { if (lane == 0) atomicOr(err, 1u); return; }
    const uint32_t payload_words = (rec_bytes - fixed) >> 2;
    const uint8_t *states = rec + hdr;
    const uint8_t *lens = states + 2u * (S + (S & 1u));
    const uint32_t *payload = reinterpret_cast<const uint32_t *>(lens + 4u * S);
    const uint32_t m = fse_sub_len(n, S);
    const uint32_t a = lane * m;
    const uint32_t len = (lane < S && a < n) ? ((n - a < m) ? n - a : m) : 0u;
    uint32_t nbits = 0, tstate = 0;
    if (lane < S) {
        nbits = lens[4 * lane] | ((uint32_t)lens[4 * lane + 1] << 8) | ((uint32_t)lens[4 * lane + 2] << 16) | ((uint32_t)lens[4 * lane + 3] << 24);
        tstate = states[2 * lane] | ((uint32_t)states[2 * lane + 1] << 8);
    }
    // a sub-stream of m symbols holds at most m * L bits: a length above that is corrupt (and would index past the record)
    if (nbits > m * L) { bad = true; nbits = 0; }
    const uint32_t nwords = (nbits + 31u) >> 5;
    uint32_t winc = nwords;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(winc, o); if (lane >= (uint32_t)o) winc += v; }
    if (__shfl(winc, 63) > payload_words) bad = true;                          // the sub-streams must fit the record
    if (__ballot(bad) != 0ull) { if (lane == 0) atomicOr(err, 1u); return; }
    const uint32_t *w = payload + (winc - nwords);
    uint32_t pos = nbits;
    uint8_t *dst = out + off + a;
    uint32_t pack = 0;
    for (uint32_t i = 0; i < len; ++i) {
        if (tstate >= N) { bad = true; break; }
        const uint32_t y = s_sub[tstate];
        const uint32_t nb = L - (31u - (uint32_t)__builtin_clz(y));
        if (pos < nb) { bad = true; break; }
        pos -= nb;
        uint32_t v = 0;
        if (nb) {
            const uint32_t wi = pos >> 5, sh = pos & 31u;
            uint64_t two = w[wi];
            if (sh + nb > 32) two |= (uint64_t)w[wi + 1] << 32;
            v = (uint32_t)(two >> sh) & ((1u << nb) - 1u);
        }
        pack |= (uint32_t)s_symat[tstate] << (8 * (i & 3u));
        if ((i & 3u) == 3u) {
            if ((((uintptr_t)(dst + i - 3)) & 3u) == 0) *reinterpret_cast<uint32_t *>(dst + i - 3) = pack;
            else { dst[i - 3] = (uint8_t)pack; dst[i - 2] = (uint8_t)(pack >> 8); dst[i - 1] = (uint8_t)(pack >> 16); dst[i] = (uint8_t)(pack >> 24); }
            pack = 0;
        }
        tstate = (y << nb) + v - N;
    }
    if (!bad && len) {
        for (uint32_t k = len & ~3u; k < len; ++k) dst[k] = (uint8_t)(pack >> (8 * (k & 3u)));
        if (tstate != 0 || pos != 0) bad = true;              // must land on the encoder's start state
    }
    if (bad) atomicOr(err, 1u);
}

__global__ __launch_bounds__(64)
void k_fse_normalise(const uint64_t *__restrict__ freq, uint32_t L, uint32_t *__restrict__ cnt)
{
    __shared__ uint32_t s_f[256], s_c[256];
    // the block kernels count in u32 (a block is <= 65536 bytes); this entry takes u64 counts and
    // refuses nothing below 2^32 per symbol — enough for the parity tests of the rule itself
    for (uint32_t s = threadIdx.x; s < 256; s += 64) s_f[s] = (uint32_t)freq[s];
    __builtin_amdgcn_wave_barrier();
    fse_normalise_wave(s_f, L, s_c);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t s = threadIdx.x; s < 256; s += 64) cnt[s] = s_c[s];
}

// ---------------------------------------------------------------------------------------------
// concatenation of the per-block records (byte granular = bit offsets that are multiples of 32)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void k_fse_scan(const uint64_t *__restrict__ bits, uint64_t nblocks, uint64_t *__restrict__ offsets)
{
    __shared__ uint64_t s_tmp[18];
    const uint64_t per = (nblocks + 1023) / 1024;
    const uint64_t a = (uint64_t)threadIdx.x * per, b = a + per < nblocks ? a + per : nblocks;
    uint64_t s = 0;
    for (uint64_t i = a; i < b; ++i) s += bits[i];
    uint64_t inc = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint64_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t run = 0; for (int w = 0; w < 16; ++w) { uint64_t t = s_tmp[w]; s_tmp[w] = run; run += t; } s_tmp[16] = run; }
    __syncthreads();
    uint64_t run = s_tmp[wave] + inc - s;
    for (uint64_t i = a; i < b; ++i) { const uint64_t v = bits[i]; offsets[i] = run; run += v; }
    if (threadIdx.x == 0) offsets[nblocks] = s_tmp[16];
}

__global__ __launch_bounds__(256)
void k_fse_pack(const uint8_t *__restrict__ recs, uint64_t rec_stride, const uint64_t *__restrict__ offsets,
                uint64_t nblocks, uint32_t *__restrict__ out)
{
    // one workgroup per block record: straight dword copy (records and offsets are 4-byte aligned)
    const uint64_t b = blockIdx.x;
    const uint64_t o0 = offsets[b] >> 5, o1 = offsets[b + 1] >> 5;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(recs + b * rec_stride);
    for (uint64_t i = threadIdx.x; i < o1 - o0; i += 256) out[o0 + i] = src[i];
}

// =============================================================================================
static mi_status fse_check(const mi_fse_params *p)
{
    if (!p) return MI_ERR_ARG;
    if (p->table_log < 8 || p->table_log > FSE_MAX_LOG) return MI_ERR_ARG;
    if (p->streams < 1 || p->streams > FSE_MAX_S) return MI_ERR_ARG;
    if (p->spread > 1) return MI_ERR_ARG;
    if (p->block < 4 || p->block > 65536 || (p->block & 3u)) return MI_ERR_ARG;
    return MI_OK;
}

extern "C" uint64_t mi_fse_block_bound(const mi_fse_params *p)
{
    if (!p) return 0;
    const uint64_t S = p->streams, n = p->block;
    const uint64_t m = ((n + S - 1) / S + 3) & ~3ull;
    const uint64_t words = S * ((m * p->table_log + 31) / 32 + 1);
    return (32 + 512 + 2 * (S + 1) + 4 * S + 4 * words + 15) & ~15ull;
}

extern "C" mi_status mi_fse_encode_dev(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *d_in, uint64_t n,
                                       uint8_t *d_packed, uint64_t cap_bytes, uint64_t *d_offsets, void *stream)
{
    if (!ctx || !d_packed || !d_offsets || (n && !d_in)) return MI_ERR_ARG;
    mi_status st = fse_check(p);
    if (st) return st;
    if (((uintptr_t)d_packed & 3u) != 0) return MI_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const FseP P{p->table_log, p->streams, p->spread, p->block};
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    const uint64_t stride = mi_fse_block_bound(p);
    if (cap_bytes < nblocks * stride) return MI_ERR_CAPACITY;
    if (nblocks == 0) { MI_HIP(ctx, hipMemsetAsync(d_offsets, 0, 8, s)); return MI_OK; }
    st = mi_ws_reserve(ctx, nblocks * stride + (nblocks + 2) * 8 + 4096);
    if (st) return st;
    mi_carver cv(ctx->ws);
    uint8_t *recs = cv.take<uint8_t>(nblocks * stride);
    uint64_t *bits = cv.take<uint64_t>(nblocks + 1);
    {
        mi_prof_scope pr(ctx, "k_fse_encode", s, n);
        hipLaunchKernelGGL(k_fse_encode, dim3((unsigned)nblocks), dim3(64), 3u << P.L, s, d_in, n, P, recs, stride, bits);
    }
    hipLaunchKernelGGL(k_fse_scan, dim3(1), dim3(1024), 0, s, bits, nblocks, d_offsets);
    {
        mi_prof_scope pr(ctx, "k_fse_pack", s, n);
        hipLaunchKernelGGL(k_fse_pack, dim3((unsigned)nblocks), dim3(256), 0, s, recs, stride, d_offsets, nblocks,
                           reinterpret_cast<uint32_t *>(d_packed));
    }
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}

extern "C" mi_status mi_fse_decode_dev(mi_ctx *ctx, const mi_fse_params *p, const uint8_t *d_packed, uint64_t packed_bytes,
                                       const uint64_t *d_offsets, uint8_t *d_out, uint64_t n, void *stream)
{
    if (!ctx || !d_packed || !d_offsets || (n && !d_out) || ((uintptr_t)d_packed & 3u)) return MI_ERR_ARG;
    mi_status st = fse_check(p);
    if (st) return st;
    if (n == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    const FseP P{p->table_log, p->streams, p->spread, p->block};
    const uint64_t nblocks = (n + P.block - 1) / P.block;
    uint32_t *err = mi_err_slot(ctx, s);
    if (!err) return MI_ERR_HIP;
    {
        mi_prof_scope pr(ctx, "k_fse_decode", s, n);
        hipLaunchKernelGGL(k_fse_decode, dim3((unsigned)nblocks), dim3(64), 5u << P.L, s, d_packed, packed_bytes, d_offsets, P, d_out, n, err);
    }
    uint32_t h_err = 0;
    MI_HIP(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    MI_HIP(ctx, hipStreamSynchronize(s));
    return h_err ? MI_ERR_CORRUPT : MI_OK;
}

extern "C" mi_status mi_fse_normalise_dev(mi_ctx *ctx, const uint64_t *d_freq, uint32_t table_log, uint32_t *d_cnt, void *stream)
{
    if (!ctx || !d_freq || !d_cnt || table_log < 8 || table_log > FSE_MAX_LOG) return MI_ERR_ARG;
    hipLaunchKernelGGL(k_fse_normalise, dim3(1), dim3(64), 0, (hipStream_t)stream, d_freq, table_log, d_cnt);
    MI_HIP(ctx, hipGetLastError());
    return MI_OK;
}
