/* RTCD table instance + vpx_rtcd() (see include/vp8_rtcd.h).  Reference: vp8/common/rtcd.c, and the setter
 * build/make/rtcd.sh:144-185 generates. */
#define RTCD_C
#include "vp8_rtcd.h"

int vp8_decode_mb_rows_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_RECON); }
int vp8_loop_filter_batch_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_LF); }
int vp8_extend_borders_batch_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_EXTEND); }
int vp8_decode_frame_pixels_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_ALL); }

void vpx_rtcd(void)
{
    /* one specialisation: gfx950 HIP.  (The reference's generated setter picks by CPU flags.) */
    vp8_decode_mb_rows = vp8_decode_mb_rows_hip;
    vp8_loop_filter_batch = vp8_loop_filter_batch_hip;
    vp8_extend_borders_batch = vp8_extend_borders_batch_hip;
    vp8_decode_frame_pixels = vp8_decode_frame_pixels_hip;
    /* the reference's per-block entries (vp8_rtcd_blocks.hip) */
    vp8_dequantize_b = vp8_dequantize_b_hip;
    vp8_dequant_idct_add = vp8_dequant_idct_add_hip;
    vp8_dequant_idct_add_y_block = vp8_dequant_idct_add_y_block_hip;
    vp8_dequant_idct_add_uv_block = vp8_dequant_idct_add_uv_block_hip;
    vp8_loop_filter_mbv = vp8_loop_filter_mbv_hip;
    vp8_loop_filter_bv = vp8_loop_filter_bv_hip;
    vp8_loop_filter_mbh = vp8_loop_filter_mbh_hip;
    vp8_loop_filter_bh = vp8_loop_filter_bh_hip;
    vp8_loop_filter_simple_mbv = vp8_loop_filter_simple_mbv_hip;
    vp8_loop_filter_simple_mbh = vp8_loop_filter_simple_mbh_hip;
    vp8_loop_filter_simple_bv = vp8_loop_filter_simple_bv_hip;
    vp8_loop_filter_simple_bh = vp8_loop_filter_simple_bh_hip;
    vp8_short_idct4x4llm = vp8_short_idct4x4llm_hip;
    vp8_short_inv_walsh4x4_1 = vp8_short_inv_walsh4x4_1_hip;
    vp8_short_inv_walsh4x4 = vp8_short_inv_walsh4x4_hip;
    vp8_dc_only_idct_add = vp8_dc_only_idct_add_hip;
    vp8_copy_mem16x16 = vp8_copy_mem16x16_hip;
    vp8_copy_mem8x8 = vp8_copy_mem8x8_hip;
    vp8_copy_mem8x4 = vp8_copy_mem8x4_hip;
    vp8_build_intra_predictors_mby_px = vp8_build_intra_predictors_mby_px_hip;
    vp8_build_intra_predictors_mby_s_px = vp8_build_intra_predictors_mby_s_px_hip;
    vp8_build_intra_predictors_mbuv_px = vp8_build_intra_predictors_mbuv_px_hip;
    vp8_build_intra_predictors_mbuv_s_px = vp8_build_intra_predictors_mbuv_s_px_hip;
    vp8_intra4x4_predict = vp8_intra4x4_predict_hip;
    vp8_sixtap_predict16x16 = vp8_sixtap_predict16x16_hip;
    vp8_sixtap_predict8x8 = vp8_sixtap_predict8x8_hip;
    vp8_sixtap_predict8x4 = vp8_sixtap_predict8x4_hip;
    vp8_sixtap_predict4x4 = vp8_sixtap_predict4x4_hip;
    vp8_bilinear_predict16x16 = vp8_bilinear_predict16x16_hip;
    vp8_bilinear_predict8x8 = vp8_bilinear_predict8x8_hip;
    vp8_bilinear_predict8x4 = vp8_bilinear_predict8x4_hip;
    vp8_bilinear_predict4x4 = vp8_bilinear_predict4x4_hip;
}
