/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, load or call it, and only as the checker.
 *
 * orc_table.h — CPU restatement of the reference's "sliding window" map
 * word -> position: an open-addressing table with linear probing whose
 * entries are retired FIFO through a ring of bucket numbers, with NO
 * tombstones (a retired bucket simply becomes empty again).
 *
 * Follows:
 *   algorithms/lz77/lz77.c:13-41     hash (murmur3-style mix, % TABLE_SIZE)
 *   algorithms/lz77/lz77.c:43-53     init   (all clear, ring zero, cur 0)
 *   algorithms/lz77/lz77.c:55-86     insert (probe ++ without wrap)
 *   algorithms/lz77/lz77.c:94-108    find   (probe ++ without wrap)
 *   algorithms/deflate/lz77.c:77-145 insert (probe wraps modulo TABLE_SIZE)
 *   algorithms/deflate/lz77.c:147-174 find  (no wrap)
 *
 * Both reference variants are the same machine with different constants, so
 * the restatement takes them at run time:
 *   tbits  log2(TABLE_SIZE)         lz77: WINDOW_BITS+6, deflate: 15+5
 *   wbits  log2(ring length)        lz77: WINDOW_BITS,   deflate: 15
 *   wrap   insert probe wraps?      lz77: 0,             deflate: 1
 *
 * Reads past bucket T-1 (find never wraps; lz77's insert neither) are
 * undefined behaviour in the reference.  The restatement defines them: the
 * bucket array continues past T with `slack` always-empty-at-start buckets, i.e.
 * "bucket ids are unbounded integers" (SURVEY.md A.1.6 iii).
 */
#ifndef ORC_TABLE_H
#define ORC_TABLE_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NONE UINT64_MAX

typedef struct {
    uint32_t tbits, wbits;
    int      wrap;
    uint64_t nbuckets;        /* T + slack */
    uint32_t *pattern;
    uint64_t *where;
    uint8_t  *live;
    uint32_t *ring;           /* 1 << wbits bucket numbers */
    uint32_t  cur;
    int       full;
    /* instrumentation (not part of the reference state) */
    uint64_t  n_insert, n_probe_insert, n_find, n_probe_find;
    uint64_t  max_bucket;     /* highest bucket number any probe looked at */
} orc_table;

static inline uint32_t orc_rotl32(uint32_t v, unsigned r) { return (v << r) | (v >> (32u - r)); }

/* lz77/lz77.c:13-41 == deflate/lz77.c:14-42, before the final modulo */
static inline uint32_t orc_mix32(uint32_t w)
{
    uint32_t k = w * 0xcc9e2d51u;
    k = orc_rotl32(k, 15) * 0x1b873593u;
    uint32_t h = orc_rotl32(k, 13) * 5u + 0xe6546b64u;   /* 0 ^ k == k */
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13; h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

static inline uint32_t orc_home(uint32_t w, uint32_t tbits) { return orc_mix32(w) & ((1u << tbits) - 1u); }

static inline int orc_table_init(orc_table *t, uint32_t tbits, uint32_t wbits, int wrap)
{
    memset(t, 0, sizeof *t);
    t->tbits = tbits; t->wbits = wbits; t->wrap = wrap;
    t->nbuckets = (1ull << tbits) + (1ull << wbits) + 64;
    t->pattern = (uint32_t *)calloc(t->nbuckets, sizeof(uint32_t));
    t->where   = (uint64_t *)calloc(t->nbuckets, sizeof(uint64_t));
    t->live    = (uint8_t  *)calloc(t->nbuckets, 1);
    t->ring    = (uint32_t *)calloc(1ull << wbits, sizeof(uint32_t));
    return (t->pattern && t->where && t->live && t->ring) ? 0 : -1;
}

/* "fresh table" without reallocating: every field back to its initial value */
static inline void orc_table_reset(orc_table *t)
{
    memset(t->pattern, 0, t->nbuckets * sizeof(uint32_t));
    memset(t->where,   0, t->nbuckets * sizeof(uint64_t));
    memset(t->live,    0, t->nbuckets);
    memset(t->ring,    0, sizeof(uint32_t) << t->wbits);
    t->cur = 0; t->full = 0;
}

static inline void orc_table_free(orc_table *t)
{
    free(t->pattern); free(t->where); free(t->live); free(t->ring);
    memset(t, 0, sizeof *t);
}

static inline void orc_table_insert(orc_table *t, uint32_t w, uint64_t pos)
{
    const uint32_t T = 1u << t->tbits, W = 1u << t->wbits;
    uint64_t b = orc_home(w, t->tbits);
    t->n_insert++;
    while (t->live[b]) {                       /* lz77.c:61 / deflate lz77.c:99-101 */
        b = t->wrap ? ((b + 1) & (T - 1)) : (b + 1);
        t->n_probe_insert++;
    }
    if (b > t->max_bucket) t->max_bucket = b;
    t->pattern[b] = w; t->where[b] = pos; t->live[b] = 1;
    if (t->full) {                             /* retire AFTER the write: lz77.c:70-76 */
        uint32_t old = t->ring[t->cur];
        t->pattern[old] = 0; t->where[old] = 0; t->live[old] = 0;
    }
    t->ring[t->cur++] = (uint32_t)b;
    if (t->cur >= W - 1 && !t->full) t->full = 1;   /* lz77.c:81-83 */
    t->cur %= W;
}

static inline uint64_t orc_table_find(orc_table *t, uint32_t w)
{
    uint64_t b = orc_home(w, t->tbits);
    t->n_find++;
    while (t->pattern[b] != w && t->live[b]) { ++b; t->n_probe_find++; }   /* lz77.c:97-103 */
    if (b > t->max_bucket) t->max_bucket = b;
    return t->live[b] ? t->where[b] : ORC_NONE;
}

static inline uint32_t orc_word_at(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

#endif
