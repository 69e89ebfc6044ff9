/* shared by the drop-in libraries: one lazily created codec context per process */
#ifndef DROPIN_COMMON_H
#define DROPIN_COMMON_H
#include <stdio.h>
#include <stdlib.h>
#include "../../include/mi_codec.h"

static mi_ctx *g_ctx;
static mi_ctx *dropin_ctx(void)
{
    if (!g_ctx) {
        const char *e = getenv("MI_CODEC_DEVICE");
        mi_status st = mi_ctx_create(&g_ctx, e ? atoi(e) : 0);
        if (st != MI_OK) {
            fprintf(stderr, "mi_codec: %s\n", mi_status_str(st));
            exit(1);                    /* the reference's error convention; there is no CPU path to fall back to */
        }
    }
    return g_ctx;
}

/* MI_CODEC_DEVICES=0,1,2,...: the block pipelines (deflate's compress()) spread their blocks over these GPUs from this one
 * process — one context per listed device, contiguous block ranges, the streams gathered into the first device over RCCL / xGMI
 * (include/mi_codec.h "Several GPUs of one node").  Unset: one GPU (MI_CODEC_DEVICE).  A device may be listed twice (two contexts
 * on one GPU, peer copies instead of RCCL): the shape the path is tested in on a one-GPU box. */
static mi_multi *g_multi;
__attribute__((unused)) static mi_multi *dropin_multi(void)
{
    const char *e = getenv("MI_CODEC_DEVICES");
    if (g_multi || !e || !*e) return g_multi;
    int dev[64], nd = 0;
    for (const char *q = e; *q && nd < 64; ) {
        char *end;
        const long v = strtol(q, &end, 10);
        if (end == q || v < 0) { fprintf(stderr, "mi_codec: MI_CODEC_DEVICES=%s is not a list of device ordinals\n", e); exit(1); }
        dev[nd++] = (int)v;
        q = (*end == ',') ? end + 1 : end;
        if (*end && *end != ',') { fprintf(stderr, "mi_codec: MI_CODEC_DEVICES=%s is not a list of device ordinals\n", e); exit(1); }
    }
    const mi_status st = mi_multi_create(&g_multi, dev, nd);
    if (st != MI_OK) { fprintf(stderr, "mi_codec: MI_CODEC_DEVICES=%s: %s\n", e, mi_status_str(st)); exit(1); }
    return g_multi;
}

/* Side tables of the drop-ins, kept OUT of band.  The reference's BitStream / BitWriter have no room for the per-block
 * (per-tile) offsets a parallel decoder needs, and probing for a trailer behind a caller's buffer reads past a buffer
 * the reference produced.  So compress registers {buffer pointer, stream length in bits} -> table here and
 * decompress looks it up; a buffer this library did not produce simply has no entry.  The registry grows (no eviction: every
 * live stream keeps its table), is guarded by a mutex, and an entry goes when its stream is released
 * (mi_lz77_release / mi_huffman_release) or when the same address is registered again (the caller freed and malloc reused it).
 * In-process only: a stream that must outlive the process is framed (mi_frame.h). */
#include <pthread.h>
typedef struct { const void *key; uint64_t bits, n, aux, count; uint64_t *table; } dropin_side;
static dropin_side *g_side;
static size_t g_side_len, g_side_cap;
static pthread_mutex_t g_side_mu = PTHREAD_MUTEX_INITIALIZER;

__attribute__((unused)) static void dropin_side_put(const void *key, uint64_t bits, uint64_t n, uint64_t aux, const uint64_t *table, uint64_t count)
{
    uint64_t *copy = (uint64_t *)malloc(8 * (count ? count : 1));
    if (!copy) { fprintf(stderr, "mi_codec: out of memory\n"); exit(1); }
    for (uint64_t i = 0; i < count; ++i) copy[i] = table[i];
    pthread_mutex_lock(&g_side_mu);
    dropin_side *e = NULL;
    for (size_t i = 0; i < g_side_len; ++i) if (g_side[i].key == key) { e = &g_side[i]; break; }
    if (!e) {
        if (g_side_len == g_side_cap) {
            const size_t cap = g_side_cap ? 2 * g_side_cap : 16;
            dropin_side *g = (dropin_side *)realloc(g_side, cap * sizeof *g);
            if (!g) { fprintf(stderr, "mi_codec: out of memory\n"); exit(1); }
            g_side = g; g_side_cap = cap;
        }
        e = &g_side[g_side_len++];
        e->table = NULL;
    }
    free(e->table);
    e->table = copy; e->key = key; e->bits = bits; e->n = n; e->aux = aux; e->count = count;
    pthread_mutex_unlock(&g_side_mu);
}

/* copies the entry out (the registry may grow under another thread); the table pointer stays valid until the stream is
 * released or its address registered again */
__attribute__((unused)) static int dropin_side_get(const void *key, uint64_t bits, uint64_t n, dropin_side *out)
{
    int found = 0;
    pthread_mutex_lock(&g_side_mu);
    for (size_t i = 0; i < g_side_len; ++i)
        if (g_side[i].key == key && g_side[i].table && g_side[i].bits == bits && g_side[i].n == n) { *out = g_side[i]; found = 1; break; }
    pthread_mutex_unlock(&g_side_mu);
    return found;
}

__attribute__((unused)) static void dropin_side_drop(const void *key)
{
    pthread_mutex_lock(&g_side_mu);
    for (size_t i = 0; i < g_side_len; ++i)
        if (g_side[i].key == key) { free(g_side[i].table); g_side[i] = g_side[--g_side_len]; break; }
    pthread_mutex_unlock(&g_side_mu);
}

__attribute__((unused)) static size_t dropin_side_count(void)
{
    pthread_mutex_lock(&g_side_mu);
    const size_t k = g_side_len;
    pthread_mutex_unlock(&g_side_mu);
    return k;
}
#endif
