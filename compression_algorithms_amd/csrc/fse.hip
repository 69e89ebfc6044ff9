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
// One wave per block, lane l codes sub-stream l (a contiguous 1 KiB of a full block) serially: the state chain is the work.
// Round 3 moved 9 x the algorithmic bytes for it (profiles/pmc_traffic_fse.json: 14.5 GB per 10^9 B at 4.1 TB/s — the kernel was
// bound by its own waste): every lane loaded 16 B at a time from its own cache line, eight visits per line spread over ~2 000
// cycles each while twenty waves per CU shared a 32 KiB L1 (every line fetched up to eight times), and stored its output one
// dword at a time at a fixed stride — 64 partial 32-byte sectors per wave store, 8 x write amplification — after which the wave
// copied the record down in place and k_fse_pack copied it again.  Round 4:
//   input   a lane takes a whole 128-byte line into registers at once (eight 16-byte loads back to back: the line is fetched
//           ONCE), the next line is in flight while these 128 symbols are coded;
//   tables  per symbol ONE packed dword {nb_hi, threshold, delta} instead of three LDS reads, all 16 of a load issued together,
//           off the serial chain (they do not depend on the state);
//   output  words go to a lane-private 64-byte LDS window (interleaved: no bank conflicts) and leave as whole 64-byte pieces
//           (two full sectors) to the lane's fixed-stride region of a SCRATCH record; nothing is compacted in place;
//   pack    k_fse_pack places header and sub-streams at their final offsets: the payload moves once.
// HBM traffic per 10^9 B: 2 n read (histogram pass + coding pass) + c written + c read + c written by the pack (c = 0.61 n),
// against 14.5 + 1.2 GB before.  The record format is unchanged (include/mi_fse.h, oracle/orc_fse.c: byte-equal).
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

// scratch record of one block (k_fse_encode -> k_fse_pack): the final record's fixed part (bitmap, counts, states, lengths:
// FSE_SCR_FIXED bytes at most) followed by the 64 lanes' sub-streams at a fixed stride of `lstr` words (a multiple of 16: a
// lane's 64-byte pieces are 64-byte aligned)
#define FSE_SCR_FIXED 1024u                       // >= 32 + 512 + 2 * (64 + 1) + 4 * 64 = 930
__host__ __device__ __forceinline__ uint32_t fse_lane_stride_words(uint32_t block, uint32_t S, uint32_t L)
{
    uint32_t m = (block + S - 1) / S; m = (m + 3u) & ~3u;
    return (((m * L + 31u) / 32u + 1u) + 15u) & ~15u;
}

__global__ __launch_bounds__(64)
void k_fse_encode(const uint8_t *__restrict__ in, uint64_t n_total, FseP P, uint8_t *__restrict__ scr, uint64_t scr_stride,
                  uint64_t *__restrict__ rec_bits, uint32_t *__restrict__ rec_fixed)
{
    // LDS: 6 KiB static + 3N dynamic (N = table size): table_log 8 could run 23 waves per CU
    __shared__ uint32_t s_work[4 * 256];                  // 4 sub-histograms; then s_fill (table build); then the output windows
    __shared__ uint32_t s_cnt[256], s_cum[256];           // s_cum becomes the packed per-symbol table
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint32_t (*s_hist)[256] = reinterpret_cast<uint32_t (*)[256]>(s_work);
    uint32_t *s_fill = s_work + 256;                      // (the histogram is dead once s_cnt exists)
    uint32_t *s_tab = s_cum;                              // nb_hi | threshold << 4 | (delta + 4096) << 17
    uint32_t *s_win = s_work;                             // [16][64]: word j of lane l's 64-byte window at j * 64 + l
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
    __builtin_amdgcn_wave_barrier();
    for (uint32_t s = lane; s < 256; s += 64) {           // (in place over s_cum: every lane reads and writes its own entries)
        const uint32_t c = s_cnt[s];
        uint32_t e = 0;
        if (c) {
            const uint32_t nb = L - (31u - (uint32_t)__builtin_clz(c));
            e = nb | ((c << nb) << 4) | ((uint32_t)((int32_t)s_cum[s] - (int32_t)c + 4096) << 17);
        }
        s_tab[s] = e;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- record header, into the scratch record
    uint8_t *rec = scr + b * scr_stride;
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
    if ((S & 1u) && lane == 0) { states[2 * S] = 0; states[2 * S + 1] = 0; }
    const uint32_t lstr = fse_lane_stride_words(P.block, S, L);
    uint32_t *mine = reinterpret_cast<uint32_t *>(rec + FSE_SCR_FIXED) + (size_t)lane * lstr;     // 64-byte aligned (scr, stride, FSE_SCR_FIXED are)
    __builtin_amdgcn_wave_barrier();                           // s_fill (in s_work) is dead: the output windows take its place

    // ---- sub-stream of this lane: bytes [a, a+len), encoded last symbol first (main.zig:58-62)
    const uint32_t m = fse_sub_len(n, S);
    const uint32_t a = lane * m;
    const uint32_t len = (lane < S && a < n) ? ((n - a < m) ? n - a : m) : 0u;
    uint32_t x = N;
    uint64_t acc = 0; uint32_t nacc = 0, wcnt = 0;
    // a finished word: into the lane's LDS window; every 16th one sends the window off as one aligned 64-byte piece
    auto put = [&](uint32_t w) {
        s_win[(wcnt & 15u) * 64u + lane] = w;
        ++wcnt;
        if ((wcnt & 15u) == 0) {
            uint4 *dst = reinterpret_cast<uint4 *>(mine + (wcnt - 16u));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[q] = make_uint4(s_win[(4 * q) * 64u + lane], s_win[(4 * q + 1) * 64u + lane], s_win[(4 * q + 2) * 64u + lane], s_win[(4 * q + 3) * 64u + lane]);
        }
    };
    auto step = [&](uint32_t e) {                              // e = s_tab[symbol]
        const uint32_t nb = (e & 15u) - (x < ((e >> 4) & 0x1FFFu) ? 1u : 0u);
        acc |= (uint64_t)(x & ((1u << nb) - 1u)) << nacc;
        nacc += nb;
        x = s_next[(int32_t)(x >> nb) + (int32_t)(e >> 17) - 4096];
    };
    auto drain = [&]() { if (nacc >= 32) { put((uint32_t)acc); acc >>= 32; nacc -= 32; } };     // (after two steps nacc <= 31 + 2 * 12)
    if ((len & 127u) == 0 && ((((uintptr_t)(src + a)) & 15u) == 0)) {
        const uint4 *v16 = reinterpret_cast<const uint4 *>(src + a);
        int32_t g = (int32_t)(len >> 7) - 1;                   // 128-byte lines, last one first
        uint4 cur[8], nxt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { cur[q] = make_uint4(0, 0, 0, 0); nxt[q] = make_uint4(0, 0, 0, 0); }
        if (g >= 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cur[q] = v16[g * 8 + q];
        }
        for (; g >= 0; --g) {
            if (g > 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) nxt[q] = v16[(g - 1) * 8 + q];
            }
#pragma unroll
            for (int q = 7; q >= 0; --q) {
                const uint32_t w4[4] = {cur[q].w, cur[q].z, cur[q].y, cur[q].x};       // backwards: last byte first
                uint32_t e[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int k = 3; k >= 0; --k) e[4 * j + (3 - k)] = s_tab[(w4[j] >> (8 * k)) & 0xFFu];
                }
#pragma unroll
                for (int j = 0; j < 16; j += 2) { step(e[j]); step(e[j + 1]); drain(); }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
        }
    } else {
        // the ragged last block: walk backwards in 4-byte groups (sub-streams start 4-byte aligned relative to the block)
        for (uint32_t i = len; i > 0;) {
            const uint32_t take = ((i & 3u) ? (i & 3u) : 4u);
            const uint32_t base = i - take;
            uint32_t w = 0;
            if (take == 4 && ((((uintptr_t)(src + a + base)) & 3u) == 0)) w = *reinterpret_cast<const uint32_t *>(src + a + base);
            else for (uint32_t k = 0; k < take; ++k) w |= (uint32_t)src[a + base + k] << (8 * k);
            for (int k = (int)take - 1; k >= 0; --k) { step(s_tab[(w >> (8 * k)) & 0xFFu]); drain(); }
            i = base;
        }
    }
    const uint32_t mybits = wcnt * 32u + nacc, final_t = x - N;
    if (nacc) put((uint32_t)acc);                              // (pad bits zero: the accumulator only ever holds real bits)
    const uint32_t nwords = wcnt;
    for (uint32_t k = nwords & ~15u; k < nwords; ++k) mine[k] = s_win[(k & 15u) * 64u + lane];       // the last, partial window
    uint32_t inc = nwords;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += v; }
    const uint32_t total_words = __shfl(inc, 63);
    if (lane < S) {
        states[2 * lane] = (uint8_t)final_t; states[2 * lane + 1] = (uint8_t)(final_t >> 8);
        lens[4 * lane] = (uint8_t)mybits; lens[4 * lane + 1] = (uint8_t)(mybits >> 8);
        lens[4 * lane + 2] = (uint8_t)(mybits >> 16); lens[4 * lane + 3] = (uint8_t)(mybits >> 24);
    }
    if (lane == 0) {
        rec_fixed[b] = hdr + states_bytes + 4u * S;
        rec_bits[b] = 8ull * ((uint64_t)hdr + states_bytes + 4ull * S + 4ull * total_words);
    }
}

// one wave per block: lane i decodes sub-stream i forwards, reading its bits backwards.
// Round 3's form fetched 62 GB per 10^9 decoded bytes (profiles/pmc_traffic_fse.json): two dependent dword loads per SYMBOL from
// a lane-private address (64 cache lines per wave instruction, no reuse left in a thrashed L1) and 4-byte stores at a 1 KiB
// stride.  Now a lane takes a whole 128-byte line of its sub-stream into an LDS window at once (the line is fetched once; the
// bit reader is a 64-bit register refilled a dword at a time from the window), one packed table entry per state
// {bits, base of the next state, symbol} replaces two LDS reads on the chain, and 64 decoded bytes leave as four back-to-back
// 16-byte stores (two full sectors).
__global__ __launch_bounds__(64)
void k_fse_decode(const uint8_t *__restrict__ packed, uint64_t packed_bytes, const uint64_t *__restrict__ offsets, FseP P,
                  uint8_t *__restrict__ out, uint64_t n_total, uint32_t *__restrict__ err)
{
    __shared__ uint32_t s_cnt[256], s_cum[256], s_fill[256];
    __shared__ uint32_t s_in[32 * 64];                               // word j of lane l's 128-byte line at j * 64 + l
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // dec[N] u32, next[N] u16, symat[N] u8
    uint32_t *s_dec = reinterpret_cast<uint32_t *>(s_dyn);            // nb | (y << nb) - N  << 4 | symbol << 20
    uint16_t *s_next = reinterpret_cast<uint16_t *>(s_dyn + 4 * (1u << P.L));
    uint8_t  *s_symat = s_dyn + 6 * (1u << P.L);
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
    // one entry per state u: sub-state y = cnt[s] + rank  <=>  next[cum[s] + rank] = N + u;  nb = L - floor(log2 y);
    // the next state is (y << nb) - N + the nb bits read
    for (uint32_t idx = lane; idx < N; idx += 64) {
        // which symbol owns slot idx of `next`: the s with cum[s] <= idx < cum[s] + cnt[s]
        uint32_t lo = 0, hi = 255;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_cum[mid] <= idx) lo = mid; else hi = mid - 1; }
        while (s_cnt[lo] == 0) --lo;
        const uint32_t u = (uint32_t)s_next[idx] - N;
        const uint32_t y = s_cnt[lo] + (idx - s_cum[lo]);
        const uint32_t nb = L - (31u - (uint32_t)__builtin_clz(y));
        s_dec[u] = nb | (((y << nb) - N) << 4) | (lo << 20);
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
    const uint32_t *w = payload + (winc - nwords);                             // this lane's words [0, nwords): nothing else is ever read
    // ---- the bit reader: stream words come through the lane's LDS line, whole 128-byte lines (absolute alignment) at a time
    const uint32_t wbase = (uint32_t)((reinterpret_cast<uintptr_t>(w) >> 2) & 31u);      // w[k] sits in line (wbase + k) >> 5 of the lane
    uint32_t cur_line = 0xFFFFFFFFu;
    auto word_at = [&](uint32_t k) -> uint32_t {                               // k < nwords
        const uint32_t q = wbase + k, ln = q >> 5;
        if (ln != cur_line) {
            cur_line = ln;
            // words of this line that belong to the lane: q' in [ln * 32, ln * 32 + 32) with 0 <= q' - wbase < nwords
#pragma unroll
            for (uint32_t j = 0; j < 32; ++j) {
                const uint32_t qq = ln * 32u + j;
                uint32_t v = 0;
                if (qq >= wbase && qq - wbase < nwords) v = w[qq - wbase];
                s_in[j * 64u + lane] = v;
            }
        }
        return s_in[(q & 31u) * 64u + lane];
    };
    uint64_t bw = 0; uint32_t have = 0; int32_t nxt = (int32_t)nwords - 1;      // bw: the `have` stream bits just below the read position
    if (nwords) {
        const uint32_t top = nbits - 32u * (nwords - 1u);                     // 1..32 bits in the last word
        const uint32_t v = word_at(nwords - 1u);
        bw = top >= 32 ? v : (v & ((1u << top) - 1u));
        have = top; nxt = (int32_t)nwords - 2;
    }
    uint8_t *dst = out + off + a;
    auto sym_step = [&]() -> uint32_t {                                        // returns the symbol; sets bad on a malformed stream
        if (tstate >= N) { bad = true; return 0u; }
        if (have < 12u && nxt >= 0) { bw = (bw << 32) | word_at((uint32_t)nxt); have += 32; --nxt; }
        const uint32_t e = s_dec[tstate];
        const uint32_t nb = e & 15u;
        if (have < nb) { bad = true; return 0u; }
        have -= nb;
        const uint32_t v = (uint32_t)(bw >> have) & ((1u << nb) - 1u);
        tstate = ((e >> 4) & 0xFFFFu) + v;
        return e >> 20;
    };
    uint32_t i = 0;
    if ((((uintptr_t)dst) & 15u) == 0) {
        for (; i + 64u <= len && !bad; i += 64u) {
            uint32_t o16[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                uint32_t pk = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) pk |= sym_step() << (8 * k);
                o16[q] = pk;
            }
            if (bad) break;
            uint4 *d4 = reinterpret_cast<uint4 *>(dst + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) d4[q] = make_uint4(o16[4 * q], o16[4 * q + 1], o16[4 * q + 2], o16[4 * q + 3]);
        }
    }
    for (; i < len && !bad; ++i) dst[i] = (uint8_t)sym_step();                 // ragged blocks, unaligned output
    if (!bad && len && (tstate != 0 || have != 0 || nxt >= 0)) bad = true;     // must land on the encoder's start state with every bit used
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
void k_fse_pack(const uint8_t *__restrict__ scr, uint64_t scr_stride, const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ rec_fixed,
                FseP P, uint32_t *__restrict__ out)
{
    // one workgroup per block: the fixed part of the record is copied, then every sub-stream goes from its fixed-stride region of
    // the scratch record to its place behind the ones before it — the payload's only move (4-byte aligned: records and offsets are)
    __shared__ uint32_t s_woff[65];
    const uint64_t b = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t fixed = rec_fixed[b], S = P.S;
    const uint8_t *rec = scr + b * scr_stride;
    uint32_t *dst = out + (offsets[b] >> 5);
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(rec);
    for (uint32_t i = tid; i < fixed / 4u; i += 256) dst[i] = fsrc[i];
    if (wave == 0) {
        const uint8_t *lens = rec + fixed - 4u * S;
        uint32_t nw = 0;
        if (lane < S) nw = ((lens[4 * lane] | ((uint32_t)lens[4 * lane + 1] << 8) | ((uint32_t)lens[4 * lane + 2] << 16) | ((uint32_t)lens[4 * lane + 3] << 24)) + 31u) >> 5;
        uint32_t inc = nw;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += v; }
        s_woff[lane] = inc - nw;
        if (lane == 63) s_woff[64] = inc;
    }
    __syncthreads();
    const uint32_t lstr = fse_lane_stride_words(P.block, S, P.L);
    const uint32_t *pay = reinterpret_cast<const uint32_t *>(rec + FSE_SCR_FIXED);
    uint32_t *pdst = dst + fixed / 4u;
    for (uint32_t l = wave; l < S; l += 4) {
        const uint32_t w0 = s_woff[l], cnt = s_woff[l + 1] - w0;
        const uint32_t *ps = pay + (size_t)l * lstr;
        for (uint32_t k = lane; k < cnt; k += 64) pdst[w0 + k] = ps[k];
    }
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
    // scratch records: fixed part + 64 lanes at a fixed stride (fse.hip header); 256-byte aligned stride
    const uint64_t scr_stride = mi_align_up((size_t)FSE_SCR_FIXED + (size_t)FSE_MAX_S * fse_lane_stride_words(P.block, P.S, P.L) * 4u, 256);
    st = mi_ws_reserve(ctx, nblocks * scr_stride + (nblocks + 2) * 12 + 8192);
    if (st) return st;
    mi_carver cv(ctx->ws);
    uint8_t *recs = cv.take<uint8_t>(nblocks * scr_stride);
    uint64_t *bits = cv.take<uint64_t>(nblocks + 1);
    uint32_t *fixed = cv.take<uint32_t>(nblocks + 1);
    {
        mi_prof_scope pr(ctx, "k_fse_encode", s, n);
        hipLaunchKernelGGL(k_fse_encode, dim3((unsigned)nblocks), dim3(64), 3u << P.L, s, d_in, n, P, recs, scr_stride, bits, fixed);
    }
    hipLaunchKernelGGL(k_fse_scan, dim3(1), dim3(1024), 0, s, bits, nblocks, d_offsets);
    {
        mi_prof_scope pr(ctx, "k_fse_pack", s, n);
        hipLaunchKernelGGL(k_fse_pack, dim3((unsigned)nblocks), dim3(256), 0, s, recs, scr_stride, d_offsets, fixed, P,
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
        hipLaunchKernelGGL(k_fse_decode, dim3((unsigned)nblocks), dim3(64), 7u << P.L, s, d_packed, packed_bytes, d_offsets, P, d_out, n, err);
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
