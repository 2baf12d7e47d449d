/* include/vp8_rtcd.h -- run-time dispatch table of the VP8 pixel path.
 *
 * The reference binds its pixel kernels through RTCD: build/make/rtcd.sh turns
 * vp8/common/rtcd_defs.sh:20-204 into vpx_rtcd.h, where each name is either #defined to its only
 * specialisation or is an `RTCD_EXTERN` function pointer that vpx_rtcd() sets once
 * (vp8/common/generic/systemdependent.c:84; rtcd.sh:144-185).  Two more pointers live outside the
 * generator: vp8_yv12_extend_frame_borders_ptr (vpx_scale/generic/scalesystemdependent.c:16-17).
 *
 * This library keeps the mechanism -- a table of function pointers filled in once by vpx_rtcd(),
 * callers go through the pointers -- but binds it at FRAME granularity: the reference's entries
 * take one 4x4 / 16x16 block per call, and a PCIe round trip per block is exactly what made the
 * reference's own OpenCL port unusable.  Entry <-> reference mapping:
 *
 *   vp8_decode_mb_rows                 <- decode_mb_row x rows (vp8/decoder/decodframe.c:334-436) and,
 *                                         through it, vp8_dequant_idct_add*, vp8_short_inv_walsh4x4*,
 *                                         vp8_build_intra_predictors_mb*_s, vp8_intra4x4_predict,
 *                                         vp8_sixtap_predict*, vp8_bilinear_predict*, vp8_copy_mem*
 *   vp8_loop_filter_frame              <- vp8_loop_filter_frame (vp8/common/loopfilter.c:203) and
 *                                         vp8_loop_filter_{mbv,bv,mbh,bh}, ..._simple_*
 *   vp8_yv12_extend_frame_borders_ptr  <- same name (scalesystemdependent.c:16,67)
 *   vp8_decode_frame_pixels            <- the three above fused into one submission
 *
 * The only specialisation is `_hip` (gfx950).  There is deliberately no `_c` fallback in the
 * product; the per-block `_c` functions with the reference's exact names and signatures live in
 * oracle/vp8_oracle.h as test infrastructure.
 */
#ifndef VP8_RTCD_H
#define VP8_RTCD_H
#include "vp8hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#ifdef RTCD_C
#define RTCD_EXTERN
#else
#define RTCD_EXTERN extern
#endif

int vp8_decode_mb_rows_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_decode_mb_rows)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

int vp8_loop_filter_frame_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_loop_filter_frame)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

int vp8_yv12_extend_frame_borders_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_yv12_extend_frame_borders_ptr)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

int vp8_decode_frame_pixels_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_decode_frame_pixels)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

void vpx_rtcd(void);

#ifdef __cplusplus
}
#endif
#endif
