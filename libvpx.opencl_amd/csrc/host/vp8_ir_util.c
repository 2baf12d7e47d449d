/* Whole-frame forms of include/vp8_ir.h's per-macroblock converters between the dense view of the IR (what the oracle and the
 * tests speak) and the device form (what an IR slot holds): for callers that have one and need the other.  The hot paths never
 * come by here -- the feeder and the device's entropy decoder write the device form themselves. */
#include "vp8_ir.h"
#include <stddef.h>

/* mbs[nmb], coef[nmb * 400] -> mbx[nmb], blocks[up to nmb * 24 * 16]; returns the number of blocks written */
size_t vp8ir_compact_frame(const vp8ir_mb *mbs, const int16_t *coef, int nmb, vp8ir_mbx *mbx, int16_t *blocks)
{
    size_t nb = 0;
    int i;
    for (i = 0; i < nmb; i++) nb += vp8ir_compact_mb(&mbs[i], coef + (size_t)i * VP8IR_COEF_PER_MB, (uint32_t)nb, &mbx[i], blocks);
    return nb;
}

/* mbx[nmb], blocks -> mbs[nmb] (sparse_first cleared; may be NULL), coef[nmb * 400] */
void vp8ir_expand_frame(const vp8ir_mbx *mbx, const int16_t *blocks, int nmb, vp8ir_mb *mbs, int16_t *coef)
{
    int i;
    for (i = 0; i < nmb; i++) vp8ir_expand_mb(&mbx[i], blocks, mbs ? &mbs[i] : NULL, coef + (size_t)i * VP8IR_COEF_PER_MB);
}
