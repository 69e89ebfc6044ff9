/* dropin_fse.c — C entry points of the FSE path (see include/mi_fse.h). */
#include <string.h>
#include "../../include/mi_fse.h"
#include "dropin_common.h"

static mi_fse_params fse_p(void) { mi_fse_params p = { MI_FSE_TABLE_LOG, MI_FSE_STREAMS, 1, MI_FSE_BLOCK }; return p; }

size_t fse_compress_bound(size_t len)
{
    mi_fse_params p = fse_p();
    const uint64_t nb = (len + p.block - 1) / p.block;
    return 16 + 8 * (nb + 1) + nb * mi_fse_block_bound(&p);
}

size_t fse_compress(const uint8_t *input, size_t len, uint8_t *output)
{
    mi_fse_params p = fse_p();
    const uint64_t nb = (len + p.block - 1) / p.block;
    uint64_t *h = (uint64_t *)output;
    h[0] = len; h[1] = nb;
    uint8_t *recs = output + 16 + 8 * (nb + 1);
    mi_status st = mi_fse_encode(dropin_ctx(), &p, input, len, recs, nb * mi_fse_block_bound(&p), h + 2);
    if (st != MI_OK) { fprintf(stderr, "fse_compress: %s\n", mi_status_str(st)); return 0; }
    return 16 + 8 * (nb + 1) + h[2 + nb] / 8;
}

size_t fse_decompress(const uint8_t *input, size_t len, uint8_t *output, size_t capacity)
{
    if (len < 16) return 0;
    const uint64_t *h = (const uint64_t *)input;
    const uint64_t n = h[0], nb = h[1];
    if (n > capacity || nb > (len - 16) / 8 || len < 16 + 8 * (nb + 1)) return 0;
    mi_fse_params p = fse_p();
    if (nb != (n + p.block - 1) / p.block) return 0;
    mi_status st = mi_fse_decode(dropin_ctx(), &p, input + 16 + 8 * (nb + 1), len - (16 + 8 * (nb + 1)), h + 2, output, n);
    if (st != MI_OK) { fprintf(stderr, "fse_decompress: %s\n", mi_status_str(st)); return 0; }
    return n;
}
