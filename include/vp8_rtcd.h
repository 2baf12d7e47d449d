/* include/vp8_rtcd.h -- run-time dispatch table of the VP8 pixel path.
 *
 * The reference binds its pixel kernels through RTCD: build/make/rtcd.sh turns
 * vp8/common/rtcd_defs.sh:20-204 into vpx_rtcd.h, where each name is either #defined to its only
 * specialisation or is an `RTCD_EXTERN` function pointer that vpx_rtcd() sets once
 * (vp8/common/generic/systemdependent.c:84; rtcd.sh:144-185).
 *
 * This library keeps the mechanism -- a table of function pointers filled in once by vpx_rtcd(),
 * callers go through the pointers -- in two parts.  The only specialisation of every entry is
 * `_hip` (gfx950); there is deliberately no `_c` fallback in the product (the `_c` restatements live in
 * oracle/vp8_oracle.h as test infrastructure).
 *
 * PART 1 -- frame-granular entries (what the decoder uses; vp8_dx_iface.c goes through
 * vp8_decode_frame_pixels).  The reference's entries take one 4x4 / 16x16 block per call, and a PCIe
 * round trip per block is exactly what made the reference's own OpenCL port unusable (SURVEY.md
 * section 2.1), so the table the decoder runs on takes batches of whole frames.  None of these names
 * exists in the reference:
 *
 *   vp8_decode_mb_rows        <- decode_mb_row x rows (vp8/decoder/decodframe.c:334-436) and, through it,
 *                                every transform / prediction entry of part 2
 *   vp8_loop_filter_batch     <- vp8_loop_filter_frame (vp8/common/loopfilter.c:203) and the loop-filter
 *                                entries of part 2
 *   vp8_extend_borders_batch  <- vp8_yv12_extend_frame_borders_ptr (vpx_scale/generic/scalesystemdependent.c:16,67)
 *   vp8_decode_frame_pixels   <- the three above fused into one submission
 *
 * PART 2 -- the reference's 32 per-block decoder entries (every decoder-side `prototype` of rtcd_defs.sh:20-204), same names, same
 * arguments, same results, each executed on the GPU (libvpx.opencl_amd/csrc/hip/vp8_rtcd_blocks.hip: the
 * caller's blocks are staged through pinned memory, one wavefront runs the same block arithmetic the frame
 * kernels use -- vp8_block_prims.hip.h -- and the call returns when the result is back).  They exist so that
 * a maintainer can swap one entry of the reference's table at a time and compare, and so that the block
 * arithmetic of the product is testable in the reference's own terms (tests/test_gpu_rtcd_blocks.py).  They
 * are NOT a fast path: a call costs a kernel launch and a synchronisation.  A failed launch prints to stderr
 * and aborts -- the prototypes have no error channel and a silent no-op would be a wrong picture.
 *   27 entries have the reference's exact prototype.  The five that take the reference's decoder
 * structures cannot (MACROBLOCKD's layout depends on the reference's build configuration, blockd.h:232-330):
 *   - vp8_dequantize_b keeps its prototype over `struct blockd`, declared here with the leading members the
 *     entry reads (blockd.h:186-192), so a reference BLOCKD* can be passed as it is;
 *   - the four vp8_build_intra_predictors_mb* entries are exported under `_px` names taking the fields of
 *     MACROBLOCKD the reference's functions read (reconintra.c:26-137,139-241,243-401,403-521).
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

/* ---------------------------------------------------------------- part 1: frame-granular */
int vp8_decode_mb_rows_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_decode_mb_rows)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

int vp8_loop_filter_batch_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_loop_filter_batch)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

int vp8_extend_borders_batch_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_extend_borders_batch)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

int vp8_decode_frame_pixels_hip(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);
RTCD_EXTERN int (*vp8_decode_frame_pixels)(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs);

/* ---------------------------------------------------------------- part 2: per-block (rtcd_defs.sh) */
/* leading members of the reference's BLOCKD (vp8/common/blockd.h:186-192) */
struct blockd {
    short *qcoeff_base;
    int qcoeff_offset;
    short *dqcoeff_base;
    int dqcoeff_offset;
};
/* vp8/common/loopfilter.h:51-57: each member points at 16 copies of the limit */
struct loop_filter_info {
    const unsigned char *mblim;
    const unsigned char *blim;
    const unsigned char *lim;
    const unsigned char *hev_thr;
};

#define VP8_RTCD_ENTRY(ret, name, args) \
    ret name##_hip args;                \
    RTCD_EXTERN ret(*name) args;

/* rtcd_defs.sh:20-34 */
VP8_RTCD_ENTRY(void, vp8_dequantize_b, (struct blockd *, short *dqc))
VP8_RTCD_ENTRY(void, vp8_dequant_idct_add, (short *input, short *dq, unsigned char *output, int stride))
VP8_RTCD_ENTRY(void, vp8_dequant_idct_add_y_block, (short *q, short *dq, unsigned char *dst, int stride, char *eobs))
VP8_RTCD_ENTRY(void, vp8_dequant_idct_add_uv_block,
               (short *q, short *dq, unsigned char *dst_u, unsigned char *dst_v, int stride, char *eobs))
/* rtcd_defs.sh:39-86 */
VP8_RTCD_ENTRY(void, vp8_loop_filter_mbv,
               (unsigned char *y, unsigned char *u, unsigned char *v, int ystride, int uv_stride, struct loop_filter_info *lfi))
VP8_RTCD_ENTRY(void, vp8_loop_filter_bv,
               (unsigned char *y, unsigned char *u, unsigned char *v, int ystride, int uv_stride, struct loop_filter_info *lfi))
VP8_RTCD_ENTRY(void, vp8_loop_filter_mbh,
               (unsigned char *y, unsigned char *u, unsigned char *v, int ystride, int uv_stride, struct loop_filter_info *lfi))
VP8_RTCD_ENTRY(void, vp8_loop_filter_bh,
               (unsigned char *y, unsigned char *u, unsigned char *v, int ystride, int uv_stride, struct loop_filter_info *lfi))
VP8_RTCD_ENTRY(void, vp8_loop_filter_simple_mbv, (unsigned char *y, int ystride, const unsigned char *blimit))
VP8_RTCD_ENTRY(void, vp8_loop_filter_simple_mbh, (unsigned char *y, int ystride, const unsigned char *blimit))
VP8_RTCD_ENTRY(void, vp8_loop_filter_simple_bv, (unsigned char *y, int ystride, const unsigned char *blimit))
VP8_RTCD_ENTRY(void, vp8_loop_filter_simple_bh, (unsigned char *y, int ystride, const unsigned char *blimit))
/* rtcd_defs.sh:92-108 */
VP8_RTCD_ENTRY(void, vp8_short_idct4x4llm, (short *input, unsigned char *pred, int pitch, unsigned char *dst, int dst_stride))
VP8_RTCD_ENTRY(void, vp8_short_inv_walsh4x4_1, (short *input, short *output))
VP8_RTCD_ENTRY(void, vp8_short_inv_walsh4x4, (short *input, short *output))
VP8_RTCD_ENTRY(void, vp8_dc_only_idct_add,
               (short input, unsigned char *pred, int pred_stride, unsigned char *dst, int dst_stride))
/* rtcd_defs.sh:113-139 */
VP8_RTCD_ENTRY(void, vp8_copy_mem16x16, (unsigned char *src, int src_pitch, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_copy_mem8x8, (unsigned char *src, int src_pitch, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_copy_mem8x4, (unsigned char *src, int src_pitch, unsigned char *dst, int dst_pitch))
/* the four `struct macroblockd *x` entries, by the fields they read: y/u/v = x->dst.{y,u,v}_buffer of the macroblock
   (the row above and the column to the left must be readable, as in a frame with borders), mode =
   x->mode_info_context->mbmi.{mode,uv_mode}, up/left = x->{up,left}_available.  `_s` writes the prediction in place,
   the others into x->predictor: ypred = 16 rows of 16, upred / vpred = 8 rows of 8. */
VP8_RTCD_ENTRY(void, vp8_build_intra_predictors_mby_px,
               (const unsigned char *y, int y_stride, int mode, int up_available, int left_available, unsigned char *ypred))
VP8_RTCD_ENTRY(void, vp8_build_intra_predictors_mby_s_px,
               (unsigned char *y, int y_stride, int mode, int up_available, int left_available))
VP8_RTCD_ENTRY(void, vp8_build_intra_predictors_mbuv_px,
               (const unsigned char *u, const unsigned char *v, int uv_stride, int uv_mode, int up_available,
                int left_available, unsigned char *upred, unsigned char *vpred))
VP8_RTCD_ENTRY(void, vp8_build_intra_predictors_mbuv_s_px,
               (unsigned char *u, unsigned char *v, int uv_stride, int uv_mode, int up_available, int left_available))
VP8_RTCD_ENTRY(void, vp8_intra4x4_predict, (unsigned char *src, int src_stride, int b_mode, unsigned char *dst, int dst_stride))
/* rtcd_defs.sh:174-204 */
VP8_RTCD_ENTRY(void, vp8_sixtap_predict16x16,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_sixtap_predict8x8,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_sixtap_predict8x4,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_sixtap_predict4x4,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_bilinear_predict16x16,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_bilinear_predict8x8,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_bilinear_predict8x4,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))
VP8_RTCD_ENTRY(void, vp8_bilinear_predict4x4,
               (unsigned char *src, int src_pitch, int xofst, int yofst, unsigned char *dst, int dst_pitch))

/* Device the per-block entries run on (default 0); call before the first per-block entry.  Returns 0, or -1 once
   the staging buffers exist on another device.  (A per-block call makes that device the calling thread's current HIP
   device and leaves it so.) */
int vp8_rtcd_blocks_set_device(int device);

void vpx_rtcd(void);

#ifdef __cplusplus
}
#endif
#endif
