/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_table.h).
 *
 * orc_fse.c — CPU statement of the FSE / tANS path.
 *
 * PARITY UNPINNED: algorithms/fse/src/main.zig does not compile (syntax error at
 * :47, undefined fields at :29-38), there is no zig toolchain here, and the
 * reference holds no test vector for it.  What IS taken from the reference:
 *
 *   histogram                 fse/src/main.zig:88-96
 *   normalisation to 2^L      fse/src/main.zig:106-149  (f64 scale, trunc, min 1,
 *                             remainder to the first maximal symbol)   -> orc_fse_normalise
 *   cumulative start offsets  fse/src/main.zig:159-166  (symbol order, contiguous ranges)
 *   transition rule           fse/src/main.zig:177      next = (state >> bits) + offset
 *   reverse-order encode      fse/src/main.zig:58-62
 *   final state flush         fse/src/main.zig:65
 *   LSB-first bit append      fse/src/main.zig:28-39
 *
 * Everything the sketch leaves unfinished is DEFINED here (and in DESIGN.md):
 * a table-driven tANS.  With N = 2^L, cnt[s] the normalised counts and
 * cum[s] their exclusive prefix sum in symbol order, a state is x in [N, 2N).
 *
 *   encode symbol s:  k = floor(log2(cnt[s])); nb = L - k;
 *                     if (x < (cnt[s] << nb)) nb -= 1;        (state-dependent bit count:
 *                                                              the sketch's fixed `bits` is
 *                                                              not decodable in general)
 *                     emit the low nb bits of x, LSB first;
 *                     y = x >> nb   (in [cnt[s], 2 cnt[s]));
 *                     x = N + position of sub-state y of s    (":177": (state >> bits) + offset,
 *                                                              exactly that when spread = 0)
 *   decode state t = x - N:  s = symbol at position t; y = its sub-state;
 *                     nb = L - floor(log2(y)); x = (y << nb) + readbits(nb)
 *
 * A block is cut into S contiguous sub-streams (one per GPU lane); each is
 * encoded LAST symbol first starting from x = N, so that the decoder, starting
 * from the flushed final state and reading the bits backwards, produces the
 * symbols first to last and must end on x = N (integrity check).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* fse/src/main.zig:106-149.  counts in, normalised counts out (sum = 1 << L).
 * returns number of present symbols (0 = empty input: table left all zero). */
int orc_fse_normalise(const uint64_t freq[256], uint32_t L, uint32_t cnt[256])
{
    const uint64_t N = 1ull << L;
    uint64_t total = 0, nsym = 0;
    memset(cnt, 0, 256 * sizeof(uint32_t));
    for (int s = 0; s < 256; ++s) if (freq[s]) { total += freq[s]; ++nsym; }
    if (!total) return 0;
    if (nsym > N) return -1;
    const double scale = (double)(N - nsym) / (double)total;
    uint64_t remaining = N;
    for (int s = 0; s < 256; ++s) {
        if (!freq[s]) continue;
        uint64_t g = (uint64_t)((double)freq[s] * scale);
        if (g == 0) g = 1;
        cnt[s] = (uint32_t)g;
        remaining -= g;
    }
    if (remaining) {                       /* the loop at :135-148 re-finds the same (first) maximum */
        int best = 0; uint32_t bestv = 0;
        for (int s = 0; s < 256; ++s) if (cnt[s] > bestv) { bestv = cnt[s]; best = s; }
        cnt[best] += (uint32_t)remaining;
    }
    return (int)nsym;
}

static inline uint32_t hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

typedef struct { uint32_t nb_hi, thresh; int32_t delta; } orc_fse_sym;

/*
 * Symbol placement.  Slot k (k counts the normalised occurrences in symbol order:
 * slots cum[s] .. cum[s]+cnt[s]-1 belong to s) sits at table position
 *     spread = 0:  k                          contiguous ranges, fse/src/main.zig:159-166
 *     spread = 1:  (k * step) mod N, step = N/2 + N/8 + 3   (odd, so a bijection)
 * spread 0 is the sketch's layout and costs 2-4 % over the table's ideal size on text;
 * spread 1 is the usual FSE stride and is within a fraction of a percent (measured in
 * tests/test_oracle_golden.py).  Either way a symbol's sub-states y = cnt[s] .. 2cnt[s]-1
 * are assigned to ITS positions in ascending position order.
 */
void orc_fse_spread(const uint32_t cnt[256], uint32_t L, int spread, uint8_t *sym_at)
{
    const uint32_t N = 1u << L, step = (N >> 1) + (N >> 3) + 3;
    uint32_t k = 0;
    for (int s = 0; s < 256; ++s)
        for (uint32_t j = 0; j < cnt[s]; ++j, ++k)
            sym_at[spread ? ((k * step) & (N - 1)) : k] = (uint8_t)s;
}

/* next[cum[s] + (y - cnt[s])] = N + position of sub-state y of symbol s */
void orc_fse_build_encode(const uint32_t cnt[256], uint32_t L, int spread, orc_fse_sym tab[256], uint16_t *next)
{
    const uint32_t N = 1u << L;
    uint8_t *sym_at = (uint8_t *)malloc(N);
    uint32_t fill[256], c = 0;
    for (int s = 0; s < 256; ++s) {
        fill[s] = c;
        if (!cnt[s]) { tab[s].nb_hi = 0; tab[s].thresh = 0; tab[s].delta = 0; continue; }
        uint32_t nb = L - hb32(cnt[s]);
        tab[s].nb_hi = nb;
        tab[s].thresh = cnt[s] << nb;
        tab[s].delta = (int32_t)c - (int32_t)cnt[s];
        c += cnt[s];
    }
    orc_fse_spread(cnt, L, spread, sym_at);
    for (uint32_t u = 0; u < N; ++u) next[fill[sym_at[u]]++] = (uint16_t)(N + u);
    free(sym_at);
}

/* encode one sub-stream; words must be zeroed and hold len*L/32+2 words.
 * returns bit count; *final_t = x - N */
uint64_t orc_fse_encode_sub(const uint8_t *in, uint64_t len, const orc_fse_sym tab[256], const uint16_t *next,
                            uint32_t L, uint32_t *words, uint32_t *final_t)
{
    const uint32_t N = 1u << L;
    uint32_t x = N;
    uint64_t bit = 0;
    for (uint64_t i = len; i-- > 0;) {
        const orc_fse_sym *e = &tab[in[i]];
        uint32_t nb = e->nb_hi - (x < e->thresh);
        uint32_t v = x & ((1u << nb) - 1u);
        for (uint32_t k = 0; k < nb; ++k, ++bit)
            if ((v >> k) & 1) words[bit >> 5] |= 1u << (bit & 31);
        x = next[(int32_t)(x >> nb) + e->delta];
    }
    *final_t = x - N;
    return bit;
}

/* decode one sub-stream; returns 0 on success */
int orc_fse_decode_sub(const uint32_t *words, uint64_t nbits, uint32_t final_t, const uint32_t cnt[256],
                       uint32_t L, int spread, uint8_t *out, uint64_t len)
{
    const uint32_t N = 1u << L;
    uint8_t  *sym_at = (uint8_t *)malloc(N);
    uint16_t *sub = (uint16_t *)malloc(N * sizeof(uint16_t));   /* y of each position */
    uint32_t seen[256];
    if (!sym_at || !sub) { free(sym_at); free(sub); return -1; }
    orc_fse_spread(cnt, L, spread, sym_at);
    for (int s = 0; s < 256; ++s) seen[s] = cnt[s];
    for (uint32_t u = 0; u < N; ++u) sub[u] = (uint16_t)seen[sym_at[u]]++;
    uint32_t t = final_t;
    uint64_t pos = nbits;
    int rc = 0;
    for (uint64_t i = 0; i < len; ++i) {
        if (t >= N) { rc = -2; break; }
        uint32_t y = sub[t];
        uint32_t nb = L - hb32(y);
        if (pos < nb) { rc = -3; break; }
        pos -= nb;
        uint32_t v = 0;
        for (uint32_t k = 0; k < nb; ++k)
            v |= ((words[(pos + k) >> 5] >> ((pos + k) & 31)) & 1u) << k;
        out[i] = sym_at[t];
        t = (y << nb) + v - N;
    }
    if (!rc && (t != 0 || pos != 0)) rc = -4;
    free(sym_at); free(sub);
    return rc;
}

/*
 * Block record (all little-endian, 4-byte aligned), S sub-streams:
 *   u8  present[32]                 bitmap of symbols with cnt > 0
 *   u16 cnt[nsym] (+pad to 4)       normalised counts in symbol order
 *   u16 final_t[S]
 *   u32 nbits[S]
 *   u32 payload[...]                sub-stream i occupies ceil(nbits[i]/32) words
 * Sub-stream i covers bytes [i*m, min((i+1)*m, n)) of the block, m = ceil(n/S) rounded up to 4.
 */
uint64_t orc_fse_sub_len(uint64_t n, uint32_t S) { uint64_t m = (n + S - 1) / S; return (m + 3) & ~3ull; }

uint64_t orc_fse_block_bound(uint64_t n, uint32_t L, uint32_t S)
{
    return 32 + 512 + 2ull * S + 4ull * S + ((n * L + 31) / 32) * 4 + 4ull * S + 16;
}

uint64_t orc_fse_encode_block(const uint8_t *in, uint64_t n, uint32_t L, uint32_t S, int spread, uint8_t *out)
{
    uint64_t freq[256] = {0};
    uint32_t cnt[256];
    orc_fse_sym tab[256];
    uint16_t *next = (uint16_t *)malloc(sizeof(uint16_t) << L);
    for (uint64_t i = 0; i < n; ++i) ++freq[in[i]];
    int nsym = orc_fse_normalise(freq, L, cnt);
    if (nsym < 0 || !next) { free(next); return UINT64_MAX; }
    orc_fse_build_encode(cnt, L, spread, tab, next);
    uint8_t *o = out;
    memset(o, 0, 32);
    for (int s = 0; s < 256; ++s) if (cnt[s]) o[s >> 3] |= (uint8_t)(1u << (s & 7));
    o += 32;
    for (int s = 0; s < 256; ++s) if (cnt[s]) { o[0] = cnt[s] & 0xFF; o[1] = (uint8_t)(cnt[s] >> 8); o += 2; }
    if (nsym & 1) { o[0] = o[1] = 0; o += 2; }
    uint8_t *states = o; o += 2ull * S;
    if (S & 1) { o[0] = o[1] = 0; o += 2; }
    uint8_t *lens = o; o += 4ull * S;
    const uint64_t m = orc_fse_sub_len(n, S);
    uint32_t *scratch = (uint32_t *)calloc(m * L / 32 + 4, 4);
    if (!scratch) { free(next); return UINT64_MAX; }
    for (uint32_t i = 0; i < S; ++i) {
        uint64_t a = (uint64_t)i * m, len = a >= n ? 0 : (n - a < m ? n - a : m);
        uint32_t ft;
        memset(scratch, 0, (m * L / 32 + 4) * 4);
        uint64_t nb = orc_fse_encode_sub(in + (a < n ? a : 0), len, tab, next, L, scratch, &ft);
        states[2 * i] = ft & 0xFF; states[2 * i + 1] = (uint8_t)(ft >> 8);
        uint32_t nb32 = (uint32_t)nb;
        memcpy(lens + 4 * i, &nb32, 4);
        uint64_t nw = (nb + 31) / 32;
        memcpy(o, scratch, nw * 4); o += nw * 4;
    }
    free(scratch); free(next);
    return (uint64_t)(o - out);
}

/* returns 0 on success */
int orc_fse_decode_block(const uint8_t *rec, uint64_t rec_len, uint32_t L, uint32_t S, int spread, uint8_t *out, uint64_t n)
{
    uint32_t cnt[256] = {0};
    const uint8_t *p = rec;
    if (rec_len < 32) return -1;
    int nsym = 0;
    const uint8_t *bm = p; p += 32;
    for (int s = 0; s < 256; ++s) if ((bm[s >> 3] >> (s & 7)) & 1) { cnt[s] = p[0] | ((uint32_t)p[1] << 8); p += 2; ++nsym; }
    if (nsym & 1) p += 2;
    const uint8_t *states = p; p += 2ull * S;
    if (S & 1) p += 2;
    const uint8_t *lens = p; p += 4ull * S;
    const uint64_t m = orc_fse_sub_len(n, S);
    for (uint32_t i = 0; i < S; ++i) {
        uint64_t a = (uint64_t)i * m, len = a >= n ? 0 : (n - a < m ? n - a : m);
        uint32_t ft = states[2 * i] | ((uint32_t)states[2 * i + 1] << 8), nb;
        memcpy(&nb, lens + 4 * i, 4);
        uint64_t nw = ((uint64_t)nb + 31) / 32;
        if ((uint64_t)(p - rec) + nw * 4 > rec_len) return -5;
        uint32_t *w = (uint32_t *)malloc(nw * 4 + 4);
        memcpy(w, p, nw * 4);
        int rc = orc_fse_decode_sub(w, nb, ft, cnt, L, spread, out + (a < n ? a : 0), len);
        free(w);
        if (rc) return rc;
        p += nw * 4;
    }
    return 0;
}

/* ideal cost in bits of coding `in` with the normalised table: sum -log2(cnt/N); for the
 * "within 1 % of ideal" pin of SURVEY.md 8c.  Returned as bits * 2^-0 (double). */
#include <math.h>
double orc_fse_ideal_bits(const uint8_t *in, uint64_t n, uint32_t L)
{
    uint64_t freq[256] = {0}; uint32_t cnt[256];
    for (uint64_t i = 0; i < n; ++i) ++freq[in[i]];
    if (orc_fse_normalise(freq, L, cnt) <= 0) return 0.0;
    double bits = 0.0, N = (double)(1u << L);
    for (int s = 0; s < 256; ++s) if (freq[s]) bits += (double)freq[s] * -log2((double)cnt[s] / N);
    return bits;
}
