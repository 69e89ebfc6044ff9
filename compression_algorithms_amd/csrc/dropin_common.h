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
#endif
