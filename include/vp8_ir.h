/* include/vp8_ir.h -- frame-batched intermediate representation (IR).
 *
 * The data contract between the host feeder (bitstream + entropy decode, CPU only) and the
 * VP8 pixel path (dequant/IDCT, intra/inter prediction, loop filter, border extension).
 * The HIP backend (include/vp8hip.h) and the CPU oracle (oracle/vp8_oracle.h) consume exactly
 * these structures, so both run from identical inputs.
 *
 * Where the reference keeps this state (all paths relative to the reference tree):
 *   vp8ir_frame_hdr  <- VP8_COMMON / MACROBLOCKD frame-level fields
 *                       (vp8/common/onyxc_int.h:72-194, vp8/common/blockd.h:232-318),
 *                       as parsed by vp8_decode_frame (vp8/decoder/decodframe.c:690-1181)
 *   vp8ir_mb         <- MODE_INFO (vp8/common/blockd.h:168-184) + xd->eobs[25] (blockd.h:240)
 *   vp8ir_mv[16]     <- MODE_INFO.bmi[16].mv, resp. mbmi.mv replicated for non-SPLITMV MBs
 *   coef[400]        <- xd->qcoeff[400] right after vp8_decode_mb_tokens
 *                       (vp8/decoder/detokenize.c:183; decodframe.c:126)
 *
 * Layout decisions (ours, chosen for the GPU):
 *   - everything is a flat array indexed by MB in raster order (mb_row * mb_cols + mb_col);
 *     no border entries (the reference's MODE_INFO array has stride mb_cols+1).
 *   - coefficients are dense int16, 25 blocks x 16 per MB (0-15 Y raster, 16-19 U, 20-23 V,
 *     24 Y2) = 800 B/MB.  INSIDE each 4x4 block they are stored COLUMN-MAJOR:
 *         coef[blk*16 + col*4 + row]  ==  reference qcoeff[blk*16 + row*4 + col]
 *     so one wavefront lane can fetch the 4 coefficients of a block column with a single 8-byte
 *     load and run the (vertical-first) IDCT pass of vp8_short_idct4x4llm_c
 *     (vp8/common/idctllm.c:28-60) without a transpose.
 *   - MBs with VP8IR_MB_SKIP set have UNDEFINED coefficient contents (never read).
 *   - the arrays above are the DENSE form: the host-side view.  What lies in HBM is the device form (vp8ir_mbx, below).
 */
#ifndef VP8_IR_H
#define VP8_IR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MB prediction modes: numeric values of MB_PREDICTION_MODE (vp8/common/blockd.h:73-92) */
enum {
    VP8IR_DC_PRED = 0, VP8IR_V_PRED, VP8IR_H_PRED, VP8IR_TM_PRED, VP8IR_B_PRED,
    VP8IR_NEARESTMV, VP8IR_NEARMV, VP8IR_ZEROMV, VP8IR_NEWMV, VP8IR_SPLITMV
};
/* 4x4 sub-block intra modes: B_PREDICTION_MODE (vp8/common/blockd.h:107-127) */
enum {
    VP8IR_B_DC_PRED = 0, VP8IR_B_TM_PRED, VP8IR_B_VE_PRED, VP8IR_B_HE_PRED, VP8IR_B_LD_PRED,
    VP8IR_B_RD_PRED, VP8IR_B_VR_PRED, VP8IR_B_VL_PRED, VP8IR_B_HD_PRED, VP8IR_B_HU_PRED
};
/* reference frames: MV_REFERENCE_FRAME (vp8/common/blockd.h:129-136) */
enum { VP8IR_INTRA_FRAME = 0, VP8IR_LAST_FRAME, VP8IR_GOLDEN_FRAME, VP8IR_ALTREF_FRAME };

#define VP8IR_MB_SKIP   0x01u   /* mb_skip_coeff AFTER decodframe.c:129 (skip flag, or eobtotal==0) */
#define VP8IR_MB_CLAMP  0x02u   /* need_to_clamp_mvs (decodemv.c:293,505) */

#define VP8IR_COEF_PER_MB 400

/* The DEVICE FORM of the coefficients -- what an IR slot holds in HBM -- is not the dense array but the compact form described
 * at vp8ir_mbx below: every producer writes it, every pixel kernel reads it, nothing converts it in between.  The dense array is
 * the host-side view (oracle, tests, vp8hip_ir_map / vp8hip_ir_fetch). */
#define VP8IR_BORDER 32         /* VP8BORDERINPIXELS (vpx_scale/yv12config.h:20) */

typedef struct vp8ir_mb {       /* 64 bytes */
    uint8_t y_mode;             /* VP8IR_DC_PRED .. VP8IR_SPLITMV */
    uint8_t uv_mode;            /* VP8IR_DC_PRED .. VP8IR_TM_PRED (intra MBs) */
    uint8_t ref_frame;          /* VP8IR_INTRA_FRAME .. VP8IR_ALTREF_FRAME */
    uint8_t flags;              /* VP8IR_MB_* */
    uint8_t segment_id;         /* 0..3 */
    uint8_t partitioning;       /* SPLITMV: 0 = 16x8, 1 = 8x16, 2 = 8x8, 3 = 4x4 */
    uint8_t rsv0[2];
    uint8_t eobs[25];           /* detokenize.c:363 "eobs[i] = c": position after the last token;
                                   Y blocks of an MB that has a Y2 block start at 1 */
    uint8_t rsv1[7];
    uint8_t b_modes[16];        /* B_PRED only */
    uint32_t sparse_first;      /* device form only (vp8ir_mbx): index of this MB's first block in the slot's block stream */
    uint32_t dc_first;          /* reserved (0) */
} vp8ir_mb;

typedef struct vp8ir_mv {       /* MV (vp8/common/mv.h:16-26): 1/8-pel units as stored by decodemv.c:112 */
    int16_t row;
    int16_t col;
} vp8ir_mv;

typedef struct vp8ir_frame_hdr { /* 64 bytes */
    uint16_t width, height;      /* display size (pc->Width/Height) */
    uint16_t mb_cols, mb_rows;   /* coded size / 16 */
    uint8_t  frame_type;         /* 0 = key, 1 = inter */
    uint8_t  version;            /* 0..3: 0 six-tap; 1,2 bilinear; 3 bilinear + full-pixel chroma
                                    (vp8_setup_version, vp8/common/alloccommon.c:153-189) */
    uint8_t  show_frame;
    uint8_t  filter_type;        /* 0 normal, 1 simple (frame-header bit, decodframe.c:878) */
    uint8_t  filter_level;       /* 0..63; 0 = loop filter skipped for the frame (onyxd_if.c:576) */
    uint8_t  sharpness_level;    /* 0..7 */
    uint8_t  segmentation_enabled;
    uint8_t  mb_segment_abs_delta;   /* 1 = SEGMENT_ABSDATA */
    int8_t   segment_quant[4];   /* segment_feature_data[MB_LVL_ALT_Q][] */
    int8_t   segment_lf[4];      /* segment_feature_data[MB_LVL_ALT_LF][] */
    uint8_t  mode_ref_lf_delta_enabled;
    int8_t   ref_lf_deltas[4];
    int8_t   mode_lf_deltas[4];
    uint8_t  base_qindex;        /* 0..127 */
    int8_t   y1dc_delta_q, y2dc_delta_q, y2ac_delta_q, uvdc_delta_q, uvac_delta_q;
    /* reference-buffer bookkeeping (decodframe.c:949-1018); consumed by the frame-pool logic
       (swap_frame_buffers, vp8/decoder/onyxd_if.c:261-316), not by the pixel kernels */
    uint8_t  refresh_last, refresh_golden, refresh_alt;
    uint8_t  copy_buffer_to_gf, copy_buffer_to_arf;   /* 0 none, 1 last, 2 the other one */
    uint8_t  sign_bias_golden, sign_bias_alt;
    uint8_t  color_space, clamping_type;
    uint8_t  num_token_partitions;
    uint8_t  lf_key_frame;       /* 1: a KEY frame some of whose macroblocks error concealment turned into inter macroblocks
                                    (decodframe.c:365-392): frame_type says inter -- the frame reads references --, the loop filter's
                                    hev thresholds stay a key frame's (cm->frame_type, loopfilter.c:24-50).  See vp8ir_lf_frame_type */
    uint8_t  rsv[14];
} vp8ir_frame_hdr;

#if defined(__HIPCC__)
#define VP8IR_INLINE __host__ __device__ static inline       /* the one helper the device side shares with the host */
#else
#define VP8IR_INLINE static inline
#endif
/* the frame type the loop filter's hev threshold goes by */
VP8IR_INLINE int vp8ir_lf_frame_type(const vp8ir_frame_hdr *h) { return h->frame_type && !h->lf_key_frame; }

/* 0: block k of the macroblock has no coefficients, 1: only its first (a lone DC), 2: more than that.
 * A luma block of a macroblock with Y2 starts its token decode at position 1 (detokenize.c:361): it has coefficients only if eob > 1;
 * any other block is a lone DC with eob == 1 and a full block with eob > 1; nothing in a skipped macroblock. */
VP8IR_INLINE int vp8ir_block_kind(const vp8ir_mb *m, int k)
{
    const int has_y2 = m->y_mode != VP8IR_B_PRED && m->y_mode != VP8IR_SPLITMV;
    if (m->flags & VP8IR_MB_SKIP) return 0;
    if (k == 24 && !has_y2) return 0;
    if (m->eobs[k] > 1) return 2;
    return (m->eobs[k] == 1 && !(has_y2 && k < 16)) ? 1 : 0;
}

typedef char vp8ir_static_assert_mb[(sizeof(vp8ir_mb) == 64) ? 1 : -1];
typedef char vp8ir_static_assert_hdr[(sizeof(vp8ir_frame_hdr) == 64) ? 1 : -1];

/* ---------------------------------------------------------------------------------------------------------------------
 * The DEVICE FORM of a frame's macroblock data (an IR slot in HBM; the pinned staging a feeder fills for it has the same layout
 * and is copied as it is):
 *
 *   mbx[nmb]   one vp8ir_mbx per macroblock, raster order: 128 bytes = one cache line -- the descriptor and the coefficients
 *              a kernel needs with it whatever the blocks hold (the Y2 block; the blocks' lone first coefficients)
 *   blocks[]   32 bytes (16 int16, column-major as in the dense form) for every block k in 0..23 with eobs[k] > 1, and for no
 *              other: in block order inside a macroblock, the macroblocks of a MACROBLOCK ROW one after the other from
 *              mbx[row start].d.sparse_first on.  (Rows may stand in any order and with gaps between them: a feeder that decodes
 *              token partitions on several threads, or lanes, gives each a region of its own.  d.sparse_first is valid for every
 *              macroblock; a consumer may follow it per macroblock or count blocks along a row.)
 *   mvs[nmb*16] as in the dense form (inter frames only)
 *
 * This is what the reference's decode_macroblock consumes as vp8_decode_mb_tokens left it (vp8/decoder/decodframe.c:126 ->
 * :224-296, vp8/common/idct_blk.c:20-86: qcoeff + eobs), minus the zeros: a block with eob <= 1 is either nothing or a single
 * first coefficient, and only blocks with more are worth 32 bytes and a memory request.  On the 1080p benchmark stream the
 * form is 0.4 of the dense form's bytes; it is what crosses PCIe, what lies in HBM and what the kernels fetch.
 * A skipped macroblock (VP8IR_MB_SKIP) has no blocks; its y2 / cdc are zero.
 */
typedef struct vp8ir_mbx {      /* 128 bytes */
    vp8ir_mb d;
    int16_t y2[16];             /* macroblock with a Y2 block (y_mode neither B_PRED nor SPLITMV): that block's coefficients, zeros
                                   past its eob.  Without: y2[k] = the first coefficient of luma block k where eobs[k] == 1, else 0 */
    int16_t cdc[8];             /* cdc[k] = the first coefficient of chroma block 16 + k where eobs[16 + k] == 1, else 0 */
    int16_t rsv[8];             /* zero */
} vp8ir_mbx;
typedef char vp8ir_static_assert_mbx[(sizeof(vp8ir_mbx) == 128) ? 1 : -1];
#define VP8IR_MBX_WORDS 32      /* dwords per vp8ir_mbx */
#define VP8IR_MAX_BLOCKS_PER_MB 24

/* dense -> device form of one macroblock: *x and up to 24 blocks at blocks + 16 * first; returns the number of blocks written */
static inline unsigned vp8ir_compact_mb(const vp8ir_mb *m, const int16_t *coef /* 400, dense */, uint32_t first, vp8ir_mbx *x,
                                        int16_t *blocks)
{
    const int has_y2 = m->y_mode != VP8IR_B_PRED && m->y_mode != VP8IR_SPLITMV;
    unsigned n = 0;
    int k, i;
    x->d = *m;
    x->d.sparse_first = first;
    x->d.dc_first = 0;
    for (i = 0; i < 16; i++) x->y2[i] = 0;
    for (i = 0; i < 8; i++) x->cdc[i] = x->rsv[i] = 0;
    if (m->flags & VP8IR_MB_SKIP) return 0;
    if (has_y2)
        for (i = 0; i < 16; i++) x->y2[i] = m->eobs[24] ? coef[384 + i] : (int16_t)0;
    for (k = 0; k < 24; k++) {
        const int kind = vp8ir_block_kind(m, k);
        if (kind == 2) {
            for (i = 0; i < 16; i++) blocks[((size_t)first + n) * 16 + i] = coef[k * 16 + i];
            n++;
        } else if (kind == 1) {
            if (k < 16) x->y2[k] = coef[k * 16];
            else x->cdc[k - 16] = coef[k * 16];
        }
    }
    return n;
}

/* device form -> dense form of one macroblock (zeros where a block has no coefficients; a skipped macroblock: all zeros) */
static inline void vp8ir_expand_mb(const vp8ir_mbx *x, const int16_t *blocks /* the slot's stream */, vp8ir_mb *m, int16_t *coef /* 400 */)
{
    const int has_y2 = x->d.y_mode != VP8IR_B_PRED && x->d.y_mode != VP8IR_SPLITMV;
    size_t at = x->d.sparse_first;
    int k, i;
    if (m) { *m = x->d; m->sparse_first = 0; m->dc_first = 0; }
    for (i = 0; i < VP8IR_COEF_PER_MB; i++) coef[i] = 0;
    if (x->d.flags & VP8IR_MB_SKIP) return;
    if (has_y2 && x->d.eobs[24])
        for (i = 0; i < 16; i++) coef[384 + i] = x->y2[i];
    for (k = 0; k < 24; k++) {
        const int kind = vp8ir_block_kind(&x->d, k);
        if (kind == 2) {
            for (i = 0; i < 16; i++) coef[k * 16 + i] = blocks[at * 16 + i];
            at++;
        } else if (kind == 1)
            coef[k * 16] = k < 16 ? x->y2[k] : x->cdc[k - 16];
    }
}

/* Frame-buffer geometry shared by every implementation (vp8_yv12_alloc_frame_buffer,
 * vpx_scale/generic/yv12config.c:55-112): 32-pixel luma border, 16-pixel chroma border,
 * y_stride = align32(aligned_w + 64), uv_stride = y_stride/2, planes Y,U,V contiguous. */
typedef struct vp8ir_geom {
    int aligned_w, aligned_h;    /* coded size, multiples of 16 */
    int y_stride, uv_stride;
    int y_plane_size, uv_plane_size;
    int frame_size;              /* bytes of one frame buffer */
    int y_off, u_off, v_off;     /* byte offsets of pixel (0,0) of each plane */
} vp8ir_geom;

static inline void vp8ir_geom_init(vp8ir_geom *g, int width, int height)
{
    int aw = (width + 15) & ~15, ah = (height + 15) & ~15;
    g->aligned_w = aw;
    g->aligned_h = ah;
    g->y_stride = (aw + 2 * VP8IR_BORDER + 31) & ~31;
    g->uv_stride = g->y_stride >> 1;
    g->y_plane_size = (ah + 2 * VP8IR_BORDER) * g->y_stride;
    g->uv_plane_size = (ah / 2 + VP8IR_BORDER) * g->uv_stride;
    g->frame_size = g->y_plane_size + 2 * g->uv_plane_size;
    g->y_off = VP8IR_BORDER * g->y_stride + VP8IR_BORDER;
    g->u_off = g->y_plane_size + (VP8IR_BORDER / 2) * g->uv_stride + VP8IR_BORDER / 2;
    g->v_off = g->y_plane_size + g->uv_plane_size + (VP8IR_BORDER / 2) * g->uv_stride + VP8IR_BORDER / 2;
}

#ifdef __cplusplus
}
#endif
#endif /* VP8_IR_H */
