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

/* Side tables of the drop-ins, kept OUT of band.  The reference's BitStream / BitWriter have no room for the per-block
 * (per-tile) offsets a parallel decoder needs, and probing for a trailer behind a caller's buffer reads past a buffer
 * the reference produced.  So compress registers {buffer pointer, stream length in bits} -> table here and
 * decompress looks it up; a buffer this library did not produce simply has no entry. */
typedef struct { const void *key; uint64_t bits, n, aux, count; uint64_t *table; } dropin_side;
#define DROPIN_SIDE_SLOTS 64
static dropin_side g_side[DROPIN_SIDE_SLOTS];
static unsigned g_side_next;

__attribute__((unused)) static void dropin_side_put(const void *key, uint64_t bits, uint64_t n, uint64_t aux, const uint64_t *table, uint64_t count)
{
    dropin_side *e = NULL;
    for (unsigned i = 0; i < DROPIN_SIDE_SLOTS; ++i) if (g_side[i].key == key) { e = &g_side[i]; break; }
    if (!e) e = &g_side[g_side_next++ % DROPIN_SIDE_SLOTS];
    free(e->table);
    e->table = (uint64_t *)malloc(8 * (count ? count : 1));
    if (!e->table) { fprintf(stderr, "mi_codec: out of memory\n"); exit(1); }
    for (uint64_t i = 0; i < count; ++i) e->table[i] = table[i];
    e->key = key; e->bits = bits; e->n = n; e->aux = aux; e->count = count;
}

__attribute__((unused)) static const dropin_side *dropin_side_get(const void *key, uint64_t bits, uint64_t n)
{
    for (unsigned i = 0; i < DROPIN_SIDE_SLOTS; ++i)
        if (g_side[i].key == key && g_side[i].table && g_side[i].bits == bits && g_side[i].n == n) return &g_side[i];
    return NULL;
}
#endif
