/* oracle/vp8_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's VP8 pixel path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * oracle/libvp8oracle.so; the product (libvpx.opencl_amd/) never links or calls it.
 *
 * Parity pinning: (host feeder -> IR -> this oracle) reproduces, on every fixture under
 * tests/golden/, the per-frame MD5s the REAL reference decoder printed for them
 * (tests/golden/*.md5, made by tests/golden/make_fixtures.py with oracle/_ref built from
 * /root/reference); and the per-block functions below are compared on random inputs with the
 * reference's own `_c` functions loaded from oracle/_ref/libvpxref.so (tests/test_oracle_vs_ref.py).
 *
 * Two layers:
 *  (1) per-block functions with the reference's RTCD names and argument meaning
 *      (vp8/common/rtcd_defs.sh:20-204; generic-gnu vpx_rtcd.h maps each name to its `_c`),
 *      prefixed vp8o_ so both libraries can be loaded into one process.  Coefficients here are
 *      in the reference's RASTER 4x4 order.
 *  (2) vp8o_decode_frame: the whole-frame driver running from the IR (include/vp8_ir.h),
 *      i.e. decode_mb_row/decode_macroblock (vp8/decoder/decodframe.c:112-436),
 *      vp8_loop_filter_frame (vp8/common/loopfilter.c:203-316) and
 *      vp8_yv12_extend_frame_borders (vpx_scale/generic/yv12extend.c:24-145).
 */
#ifndef VP8_ORACLE_H
#define VP8_ORACLE_H

#include <stdint.h>
#include "../include/vp8_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) RTCD-level restatements ------------------------------------------------------- */
/* vp8_dequantize_b_c            vp8/common/dequantize.c:17  (BLOCKD flattened to pointers) */
void vp8o_dequantize_b(const short *q, const short *dqc, short *dq);
/* vp8_dequant_idct_add_c        vp8/common/dequantize.c:29 */
void vp8o_dequant_idct_add(short *input, const short *dq, unsigned char *dest, int stride);
/* vp8_short_idct4x4llm_c        vp8/common/idctllm.c:28 */
void vp8o_short_idct4x4llm(const short *input, const unsigned char *pred, int pred_stride,
                           unsigned char *dst, int dst_stride);
/* vp8_dc_only_idct_add_c        vp8/common/idctllm.c:112 */
void vp8o_dc_only_idct_add(short input_dc, const unsigned char *pred, int pred_stride,
                           unsigned char *dst, int dst_stride);
/* vp8_short_inv_walsh4x4_c / _1_c   vp8/common/idctllm.c:140 / :194 */
void vp8o_short_inv_walsh4x4(const short *input, short *mb_dqcoeff);
void vp8o_short_inv_walsh4x4_1(const short *input, short *mb_dqcoeff);
/* vp8_dequant_idct_add_y_block_c / _uv_block_c   vp8/common/idct_blk.c:20 / :46 */
void vp8o_dequant_idct_add_y_block(short *q, const short *dq, unsigned char *dst, int stride, const char *eobs);
void vp8o_dequant_idct_add_uv_block(short *q, const short *dq, unsigned char *dstu, unsigned char *dstv,
                                    int stride, const char *eobs);
/* vp8_sixtap_predict{4x4,8x8,8x4,16x16}_c   vp8/common/filter.c:152-240 */
void vp8o_sixtap_predict(const unsigned char *src, int src_stride, int xoffset, int yoffset,
                         unsigned char *dst, int dst_pitch, int w, int h);
/* vp8_bilinear_predict{4x4,8x8,8x4,16x16}_c vp8/common/filter.c:399-494 */
void vp8o_bilinear_predict(const unsigned char *src, int src_stride, int xoffset, int yoffset,
                           unsigned char *dst, int dst_pitch, int w, int h);
/* vp8_intra4x4_predict_c        vp8/common/reconintra4x4.c:16.  `above`/`left`/`top_left` are
 * passed explicitly so callers can substitute the frame-edge rules. */
void vp8o_intra4x4_predict(const unsigned char above[8], const unsigned char left[4], unsigned char top_left,
                           int b_mode, unsigned char *dst, int dst_stride);
/* same entry point with the reference's pointer convention (reads src[-stride-1 ..]) */
void vp8o_build_intra_predictors_plane_s(unsigned char *p, int stride, int n, int mode, int up_available, int left_available);
void vp8o_intra4x4_predict_ptr(unsigned char *src, int src_stride, int b_mode, unsigned char *dst, int dst_stride);

/* loop_filter_info (vp8/common/loopfilter.h:51-57) with scalar members */
typedef struct vp8o_lf_info { unsigned char mblim, blim, lim, hev_thr; } vp8o_lf_info;
/* vp8_loop_filter_{mbv,bv,mbh,bh}_c   vp8/common/loopfilter_filters.c:357-422 */
void vp8o_loop_filter_mbv(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *lfi);
void vp8o_loop_filter_bv(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *lfi);
void vp8o_loop_filter_mbh(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *lfi);
void vp8o_loop_filter_bh(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *lfi);
/* vp8_loop_filter_simple_{vertical,horizontal}_edge_c, bvs/bhs  loopfilter_filters.c:317-430 */
void vp8o_loop_filter_simple_mbv(unsigned char *y, int ys, unsigned char blimit);
void vp8o_loop_filter_simple_mbh(unsigned char *y, int ys, unsigned char blimit);
void vp8o_loop_filter_simple_bv(unsigned char *y, int ys, unsigned char blimit);
void vp8o_loop_filter_simple_bh(unsigned char *y, int ys, unsigned char blimit);

/* per-frame tables: vp8cx_init_de_quantizer + mb_init_dequantizer (decodframe.c:50-109),
 * vp8_loop_filter_frame_init (loopfilter.c:117-201), vp8_loop_filter_update_sharpness (:66-96),
 * lf_init_lut (:24-64) */
typedef struct vp8o_dequant { short y1[2], y2[2], uv[2]; } vp8o_dequant;   /* [0]=DC, [1]=AC */
/* output-side post-processing (vp8/common/postproc.c), oracle/vp8_postproc_oracle.c */
void vp8o_post_proc_down_and_across(const unsigned char *src, unsigned char *dst, int src_stride, int dst_stride,
                                    int rows, int cols, int flimit);
void vp8o_mbpost_proc_across(const unsigned char *src, unsigned char *dst, int stride, int rows, int cols, int flimit);
void vp8o_mbpost_proc_down(const unsigned char *src, unsigned char *dst, int stride, int rows, int cols, int flimit, int rv_offset);
void vp8o_plane_add_noise(unsigned char *plane, const signed char *noise, int clamp, int width, int height, int stride,
                          const unsigned char *row_offset);
void vp8o_pp_strengths(int filter_level, int deblocking_level, int *q, int *ppl, int *ppl_demacro, int *mb_flimit);
void vp8o_pp_noise_table(int q, int a, const unsigned char *r, signed char noise[3072], int *clamp);
void vp8o_mfqe(const vp8ir_frame_hdr *hdr, const vp8ir_geom *g, const vp8ir_mb *mbs, const vp8ir_mv *mvs,
               const unsigned char *show, unsigned char *dest, int qcurr, int qprev);

void vp8o_mb_dequant(const vp8ir_frame_hdr *h, int segment_id, vp8o_dequant *dq);
void vp8o_lf_levels(const vp8ir_frame_hdr *h, unsigned char lvl[4][4][4]);
void vp8o_lf_limits(int sharpness, int filter_level, int frame_type, vp8o_lf_info *lfi);

/* ---- (2) whole-frame driver --------------------------------------------------------------- */
#define VP8O_STAGE_RECON   1   /* border seeding + predict + residual + per-row 4-px extend */
#define VP8O_STAGE_LF      2   /* in-loop deblocking (skipped when hdr->filter_level == 0)  */
#define VP8O_STAGE_EXTEND  4   /* 32/16-pixel border replication                            */
#define VP8O_STAGE_ALL     7

/* dst and refs are whole frame buffers of vp8ir_geom.frame_size bytes (layout: vp8ir_geom_init).
 * refs[VP8IR_LAST_FRAME..VP8IR_ALTREF_FRAME] may be NULL for key frames; refs[0] is unused. */
void vp8o_decode_frame(const vp8ir_frame_hdr *hdr, const vp8ir_mb *mbs, const int16_t *coef,
                       const vp8ir_mv *mvs, uint8_t *dst, const uint8_t *const refs[4], int stages);

#ifdef __cplusplus
}
#endif
#endif
