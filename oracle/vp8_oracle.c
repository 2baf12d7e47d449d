/* oracle/vp8_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See vp8_oracle.h.
 *
 * CPU restatement of the reference's VP8 pixel path.  Integer-only; every rounding, truncation
 * and clamp below is the reference's (file:line cited per function).  Written for clarity, not
 * speed: this is the checker the HIP kernels are compared against bit for bit.
 */
#include "vp8_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* ========================================================================================
 * a1-a5: dequantisation, inverse DCT / WHT, add
 * ====================================================================================== */
void vp8o_dequantize_b(const short *q, const short *dqc, short *dq)      /* dequantize.c:17-27 */
{
    for (int i = 0; i < 16; i++) dq[i] = (short)(q[i] * dqc[i]);
}

/* idctllm.c:28-110.  Vertical pass first, 16-bit intermediates, constants 20091 / 35468. */
void vp8o_short_idct4x4llm(const short *in, const unsigned char *pred, int pred_stride,
                           unsigned char *dst, int dst_stride)
{
    short tmp[16];
    for (int c = 0; c < 4; c++) {                /* columns: elements c, c+4, c+8, c+12 */
        int i0 = in[c], i1 = in[4 + c], i2 = in[8 + c], i3 = in[12 + c];
        int a = i0 + i2, b = i0 - i2;
        int cc = ((i1 * 35468) >> 16) - (i3 + ((i3 * 20091) >> 16));
        int d = (i1 + ((i1 * 20091) >> 16)) + ((i3 * 35468) >> 16);
        tmp[c] = (short)(a + d);
        tmp[12 + c] = (short)(a - d);
        tmp[4 + c] = (short)(b + cc);
        tmp[8 + c] = (short)(b - cc);
    }
    for (int r = 0; r < 4; r++) {                /* rows, with the (x+4)>>3 rounding */
        const short *t = tmp + 4 * r;
        int a = t[0] + t[2], b = t[0] - t[2];
        int cc = ((t[1] * 35468) >> 16) - (t[3] + ((t[3] * 20091) >> 16));
        int d = (t[1] + ((t[1] * 20091) >> 16)) + ((t[3] * 35468) >> 16);
        short o[4];
        o[0] = (short)((a + d + 4) >> 3);
        o[3] = (short)((a - d + 4) >> 3);
        o[1] = (short)((b + cc + 4) >> 3);
        o[2] = (short)((b - cc + 4) >> 3);
        for (int c = 0; c < 4; c++)
            dst[r * dst_stride + c] = (unsigned char)clamp255(o[c] + pred[r * pred_stride + c]);
    }
}

void vp8o_dequant_idct_add(short *input, const short *dq, unsigned char *dest, int stride)  /* dequantize.c:29-44 */
{
    for (int i = 0; i < 16; i++) input[i] = (short)(dq[i] * input[i]);
    vp8o_short_idct4x4llm(input, dest, stride, dest, stride);
    memset(input, 0, 32);
}

void vp8o_dc_only_idct_add(short input_dc, const unsigned char *pred, int pred_stride,
                           unsigned char *dst, int dst_stride)           /* idctllm.c:112-138 */
{
    int a1 = (input_dc + 4) >> 3;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            dst[r * dst_stride + c] = (unsigned char)clamp255(a1 + pred[r * pred_stride + c]);
}

void vp8o_short_inv_walsh4x4(const short *in, short *mb_dqcoeff)        /* idctllm.c:140-192 */
{
    short t[16];
    for (int c = 0; c < 4; c++) {
        int a = in[c] + in[12 + c], b = in[4 + c] + in[8 + c];
        int cc = in[4 + c] - in[8 + c], d = in[c] - in[12 + c];
        t[c] = (short)(a + b);
        t[4 + c] = (short)(cc + d);
        t[8 + c] = (short)(a - b);
        t[12 + c] = (short)(d - cc);
    }
    for (int r = 0; r < 4; r++) {
        const short *p = t + 4 * r;
        int a = p[0] + p[3], b = p[1] + p[2], cc = p[1] - p[2], d = p[0] - p[3];
        mb_dqcoeff[(4 * r + 0) * 16] = (short)((a + b + 3) >> 3);
        mb_dqcoeff[(4 * r + 1) * 16] = (short)((cc + d + 3) >> 3);
        mb_dqcoeff[(4 * r + 2) * 16] = (short)((a - b + 3) >> 3);
        mb_dqcoeff[(4 * r + 3) * 16] = (short)((d - cc + 3) >> 3);
    }
}

void vp8o_short_inv_walsh4x4_1(const short *in, short *mb_dqcoeff)      /* idctllm.c:194-204 */
{
    short a1 = (short)((in[0] + 3) >> 3);
    for (int i = 0; i < 16; i++) mb_dqcoeff[i * 16] = a1;
}

static void idct_add_block(short *q, const short *dq, unsigned char *dst, int stride, int eob)
{
    if (eob > 1)
        vp8o_dequant_idct_add(q, dq, dst, stride);
    else {                                       /* idct_blk.c:31-36: also zeroes q[0] and q[1] */
        vp8o_dc_only_idct_add((short)(q[0] * dq[0]), dst, stride, dst, stride);
        q[0] = 0;
        q[1] = 0;
    }
}

void vp8o_dequant_idct_add_y_block(short *q, const short *dq, unsigned char *dst, int stride, const char *eobs)
{                                                /* idct_blk.c:20-44 */
    for (int b = 0; b < 16; b++)
        idct_add_block(q + 16 * b, dq, dst + (b >> 2) * 4 * stride + (b & 3) * 4, stride, eobs[b]);
}

void vp8o_dequant_idct_add_uv_block(short *q, const short *dq, unsigned char *dstu, unsigned char *dstv,
                                    int stride, const char *eobs)        /* idct_blk.c:46-86 */
{
    for (int b = 0; b < 4; b++)
        idct_add_block(q + 16 * b, dq, dstu + (b >> 1) * 4 * stride + (b & 1) * 4, stride, eobs[b]);
    for (int b = 0; b < 4; b++)
        idct_add_block(q + 64 + 16 * b, dq, dstv + (b >> 1) * 4 * stride + (b & 1) * 4, stride, eobs[4 + b]);
}

/* ========================================================================================
 * a6: quantiser tables (quant_common.c:14-132, decodframe.c:50-109)
 * ====================================================================================== */
static const unsigned short dc_q[128] = {
    4, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13, 14, 15, 16, 17, 17, 18, 19, 20, 20, 21, 21, 22, 22, 23, 23, 24, 25, 25, 26,
    27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 46, 47, 48, 49, 50, 51, 52,
    53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 76, 77, 78, 79,
    80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 91, 93, 95, 96, 98, 100, 101, 102, 104, 106, 108, 110, 112, 114, 116,
    118, 122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157
};
static const unsigned short ac_q[128] = {
    4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
    34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 60, 62, 64,
    66, 68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88, 90, 92, 94, 96, 98, 100, 102, 104, 106, 108, 110, 112, 114, 116,
    119, 122, 125, 128, 131, 134, 137, 140, 143, 146, 149, 152, 155, 158, 161, 164, 167, 170, 173, 177, 181, 185,
    189, 193, 197, 201, 205, 209, 213, 217, 221, 225, 229, 234, 239, 245, 249, 254, 259, 264, 269, 274, 279, 284
};
static inline int qclamp(int q) { return q < 0 ? 0 : (q > 127 ? 127 : q); }

void vp8o_mb_dequant(const vp8ir_frame_hdr *h, int segment_id, vp8o_dequant *dq)
{
    int q = h->base_qindex;
    if (h->segmentation_enabled) {
        if (h->mb_segment_abs_delta) q = h->segment_quant[segment_id];
        else q = qclamp(h->base_qindex + h->segment_quant[segment_id]);
    }
    /* NB (decodframe.c:80): the absolute-value branch is used unclamped as a table index in the
       reference; a conforming stream keeps it in 0..127.  Clamp for memory safety only. */
    q = qclamp(q);
    dq->y1[0] = (short)dc_q[qclamp(q + h->y1dc_delta_q)];
    dq->y1[1] = (short)ac_q[q];
    dq->y2[0] = (short)(dc_q[qclamp(q + h->y2dc_delta_q)] * 2);
    {
        int v = (ac_q[qclamp(q + h->y2ac_delta_q)] * 155) / 100;
        dq->y2[1] = (short)(v < 8 ? 8 : v);
    }
    {
        int v = dc_q[qclamp(q + h->uvdc_delta_q)];
        dq->uv[0] = (short)(v > 132 ? 132 : v);
    }
    dq->uv[1] = (short)ac_q[qclamp(q + h->uvac_delta_q)];
}

/* ========================================================================================
 * a7: 16x16 luma / 8x8 chroma intra prediction, written in place (reconintra.c:139-241, 403-521)
 * ====================================================================================== */
static void intra_pred_plane(unsigned char *p, int stride, int n, int mode, int up, int left)
{
    const unsigned char *above = p - stride;
    int tl = above[-1];
    int shift_base = n == 16 ? 3 : 2;
    switch (mode) {
    case VP8IR_DC_PRED: {
        int dc = 128;
        if (up || left) {
            int sum = 0, shift = shift_base + up + left;
            if (up) for (int i = 0; i < n; i++) sum += above[i];
            if (left) for (int i = 0; i < n; i++) sum += p[i * stride - 1];
            dc = (sum + (1 << (shift - 1))) >> shift;
        }
        for (int r = 0; r < n; r++) memset(p + r * stride, dc, (size_t)n);
        break;
    }
    case VP8IR_V_PRED:
        for (int r = 0; r < n; r++) memcpy(p + r * stride, above, (size_t)n);
        break;
    case VP8IR_H_PRED:
        for (int r = 0; r < n; r++) memset(p + r * stride, p[r * stride - 1], (size_t)n);
        break;
    case VP8IR_TM_PRED:
        for (int r = 0; r < n; r++) {
            int l = p[r * stride - 1];
            for (int c = 0; c < n; c++) p[r * stride + c] = (unsigned char)clamp255(l + above[c] - tl);
        }
        break;
    default:
        break;
    }
}

/* The in-place predictor of one plane under an exported name (vp8_build_intra_predictors_mby_s / mbuv_s by the fields
 * of MACROBLOCKD they read; n = 16 for luma, 8 for a chroma plane). */
void vp8o_build_intra_predictors_plane_s(unsigned char *p, int stride, int n, int mode, int up, int left)
{
    intra_pred_plane(p, stride, n, mode, up, left);
}

/* ========================================================================================
 * a8: 4x4 sub-block intra prediction (reconintra4x4.c:16-303)
 *
 * Table form.  Edge vector P[15]: P[0]=L3 (dup), P[1..4]=L3,L2,L1,L0, P[5]=top-left,
 * P[6..13]=A0..A7, P[14]=A7 (dup).  Every directional predictor pixel is either
 *   T3(k) = (P[k-1] + 2*P[k] + P[k+1] + 2) >> 2,  T2(k) = (P[k] + P[k+1] + 1) >> 1,  or  P[k].
 * Entry encoding: kind << 4 | k  (kind 0 = copy, 1 = T2, 2 = T3).
 * ====================================================================================== */
#define C_(k) (0x00 | (k))
#define A_(k) (0x10 | (k))
#define F_(k) (0x20 | (k))
static const unsigned char bpred_tab[10][16] = {
    /* B_DC_PRED, B_TM_PRED: computed, not table driven */
    { 0 }, { 0 },
    /* B_VE_PRED: column c -> T3 centred on A[c] */
    { F_(6), F_(7), F_(8), F_(9), F_(6), F_(7), F_(8), F_(9), F_(6), F_(7), F_(8), F_(9), F_(6), F_(7), F_(8), F_(9) },
    /* B_HE_PRED: row r -> T3 centred on L[r]; last row uses the duplicated L3 */
    { F_(4), F_(4), F_(4), F_(4), F_(3), F_(3), F_(3), F_(3), F_(2), F_(2), F_(2), F_(2), F_(1), F_(1), F_(1), F_(1) },
    /* B_LD_PRED: T3 centred on A[r+c+1]; bottom-right uses the duplicated A7 */
    { F_(7), F_(8), F_(9), F_(10), F_(8), F_(9), F_(10), F_(11), F_(9), F_(10), F_(11), F_(12), F_(10), F_(11), F_(12), F_(13) },
    /* B_RD_PRED: T3 centred on P[5 - r + c] */
    { F_(5), F_(6), F_(7), F_(8), F_(4), F_(5), F_(6), F_(7), F_(3), F_(4), F_(5), F_(6), F_(2), F_(3), F_(4), F_(5) },
    /* B_VR_PRED */
    { A_(5), A_(6), A_(7), A_(8), F_(5), F_(6), F_(7), F_(8), F_(4), A_(5), A_(6), A_(7), F_(3), F_(5), F_(6), F_(7) },
    /* B_VL_PRED */
    { A_(6), A_(7), A_(8), A_(9), F_(7), F_(8), F_(9), F_(10), A_(7), A_(8), A_(9), F_(11), F_(8), F_(9), F_(10), F_(12) },
    /* B_HD_PRED */
    { A_(4), F_(5), F_(6), F_(7), A_(3), F_(4), A_(4), F_(5), A_(2), F_(3), A_(3), F_(4), A_(1), F_(2), A_(2), F_(3) },
    /* B_HU_PRED */
    { A_(3), F_(3), A_(2), F_(2), A_(2), F_(2), A_(1), F_(1), A_(1), F_(1), C_(1), C_(1), C_(1), C_(1), C_(1), C_(1) },
};
#undef C_
#undef A_
#undef F_

void vp8o_intra4x4_predict(const unsigned char above[8], const unsigned char left[4], unsigned char top_left,
                           int b_mode, unsigned char *dst, int dst_stride)
{
    if (b_mode == VP8IR_B_DC_PRED) {
        int s = 4;
        for (int i = 0; i < 4; i++) s += above[i] + left[i];
        s >>= 3;
        for (int r = 0; r < 4; r++) memset(dst + r * dst_stride, s, 4);
        return;
    }
    if (b_mode == VP8IR_B_TM_PRED) {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++)
                dst[r * dst_stride + c] = (unsigned char)clamp255(above[c] - top_left + left[r]);
        return;
    }
    {
        int P[15];
        P[0] = left[3];
        for (int i = 0; i < 4; i++) P[1 + i] = left[3 - i];
        P[5] = top_left;
        for (int i = 0; i < 8; i++) P[6 + i] = above[i];
        P[14] = above[7];
        for (int i = 0; i < 16; i++) {
            int e = bpred_tab[b_mode][i], k = e & 15, v;
            if ((e >> 4) == 2) v = (P[k - 1] + 2 * P[k] + P[k + 1] + 2) >> 2;
            else if ((e >> 4) == 1) v = (P[k] + P[k + 1] + 1) >> 1;
            else v = P[k];
            dst[(i >> 2) * dst_stride + (i & 3)] = (unsigned char)v;
        }
    }
}

void vp8o_intra4x4_predict_ptr(unsigned char *src, int src_stride, int b_mode, unsigned char *dst, int dst_stride)
{
    unsigned char above[8], left[4], tl = src[-src_stride - 1];
    memcpy(above, src - src_stride, 8);
    for (int i = 0; i < 4; i++) left[i] = src[i * src_stride - 1];
    vp8o_intra4x4_predict(above, left, tl, b_mode, dst, dst_stride);
}

/* ========================================================================================
 * a11-a13: sub-pixel interpolation (filter.c:16-494) and full-pel copies (reconinter.c:22-130)
 * ====================================================================================== */
static const short sixtap[8][6] = {
    { 0, 0, 128, 0, 0, 0 }, { 0, -6, 123, 12, -1, 0 }, { 2, -11, 108, 36, -8, 1 }, { 0, -9, 93, 50, -6, 0 },
    { 3, -16, 77, 77, -16, 3 }, { 0, -6, 50, 93, -9, 0 }, { 1, -8, 36, 108, -11, 2 }, { 0, -1, 12, 123, -6, 0 }
};
static const short bilin[8][2] = {
    { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 }, { 64, 64 }, { 48, 80 }, { 32, 96 }, { 16, 112 }
};

/* Two passes, ALWAYS both (the reference never shortcuts a zero offset): horizontal over rows
 * -2..h+2 with rounding, >>7 and a clamp to 0..255, then vertical, same rounding and clamp. */
void vp8o_sixtap_predict(const unsigned char *src, int ss, int xo, int yo, unsigned char *dst, int dp, int w, int h)
{
    int tmp[21 * 16];
    const short *hf = sixtap[xo], *vf = sixtap[yo];
    for (int r = 0; r < h + 5; r++) {
        const unsigned char *s = src + (r - 2) * ss;
        for (int c = 0; c < w; c++) {
            int t = s[c - 2] * hf[0] + s[c - 1] * hf[1] + s[c] * hf[2] + s[c + 1] * hf[3] + s[c + 2] * hf[4]
                    + s[c + 3] * hf[5] + 64;
            tmp[r * w + c] = clamp255(t >> 7);
        }
    }
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            const int *t = tmp + (r + 2) * w + c;
            int v = t[-2 * w] * vf[0] + t[-w] * vf[1] + t[0] * vf[2] + t[w] * vf[3] + t[2 * w] * vf[4]
                    + t[3 * w] * vf[5] + 64;
            dst[r * dp + c] = (unsigned char)clamp255(v >> 7);
        }
}

void vp8o_bilinear_predict(const unsigned char *src, int ss, int xo, int yo, unsigned char *dst, int dp, int w, int h)
{
    unsigned short tmp[17 * 16];
    const short *hf = bilin[xo], *vf = bilin[yo];
    for (int r = 0; r < h + 1; r++)
        for (int c = 0; c < w; c++)
            tmp[r * w + c] = (unsigned short)((src[r * ss + c] * hf[0] + src[r * ss + c + 1] * hf[1] + 64) >> 7);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++)
            dst[r * dp + c] = (unsigned char)((tmp[r * w + c] * vf[0] + tmp[(r + 1) * w + c] * vf[1] + 64) >> 7);
}

/* One prediction block at an (already final) MV: full-pel part mv>>3, fraction mv&7
 * (reconinter.c:161-227: sub-pel predictor iff (row|col)&7, else plain copy). */
static void predict_block(const unsigned char *ref, int stride, int x, int y, int mvrow, int mvcol, int w, int h,
                          int bilinear, unsigned char *dst)
{
    const unsigned char *src = ref + (y + (mvrow >> 3)) * stride + x + (mvcol >> 3);
    unsigned char *d = dst + y * stride + x;
    if ((mvrow | mvcol) & 7) {
        if (bilinear) vp8o_bilinear_predict(src, stride, mvcol & 7, mvrow & 7, d, stride, w, h);
        else vp8o_sixtap_predict(src, stride, mvcol & 7, mvrow & 7, d, stride, w, h);
    } else
        for (int r = 0; r < h; r++) memcpy(d + r * stride, src + r * stride, (size_t)w);
}

/* ========================================================================================
 * a9/a10: inter prediction of one MB (reconinter.c:348-606)
 * ====================================================================================== */
typedef struct mbedges { int left, right, top, bottom; } mbedges;     /* mb_to_*_edge, 1/8 pel */

static void clamp_luma_mv(int *row, int *col, const mbedges *e)       /* clamp_mv_to_umv_border :348-368 */
{
    if (*col < e->left - (19 << 3)) *col = e->left - (16 << 3);
    else if (*col > e->right + (18 << 3)) *col = e->right + (16 << 3);
    if (*row < e->top - (19 << 3)) *row = e->top - (16 << 3);
    else if (*row > e->bottom + (18 << 3)) *row = e->bottom + (16 << 3);
}

static void clamp_chroma_mv(int *row, int *col, const mbedges *e)     /* clamp_uvmv_to_umv_border :371-382 */
{
    if (2 * *col < e->left - (19 << 3)) *col = (e->left - (16 << 3)) >> 1;
    if (2 * *col > e->right + (18 << 3)) *col = (e->right + (16 << 3)) >> 1;
    if (2 * *row < e->top - (19 << 3)) *row = (e->top - (16 << 3)) >> 1;
    if (2 * *row > e->bottom + (18 << 3)) *row = (e->bottom + (16 << 3)) >> 1;
}

static void inter_predict_mb(const vp8ir_frame_hdr *h, const vp8ir_geom *g, const vp8ir_mb *mb, const vp8ir_mv *mv,
                             const uint8_t *ref, uint8_t *dst, int mb_row, int mb_col)
{
    int bil = h->version != 0;
    int fullpix = h->version == 3;               /* fullpixel_mask 0xfffffff8, decodframe.c:683-685 */
    int clampmv = mb->flags & VP8IR_MB_CLAMP;
    mbedges e;
    int x = mb_col * 16, y = mb_row * 16;
    const uint8_t *ry = ref + g->y_off, *ru = ref + g->u_off, *rv = ref + g->v_off;
    uint8_t *dy = dst + g->y_off, *du = dst + g->u_off, *dv = dst + g->v_off;
    e.left = -((mb_col * 16) << 3);
    e.right = ((h->mb_cols - 1 - mb_col) * 16) << 3;
    e.top = -((mb_row * 16) << 3);
    e.bottom = ((h->mb_rows - 1 - mb_row) * 16) << 3;

    if (mb->y_mode != VP8IR_SPLITMV) {           /* vp8_build_inter16x16_predictors_mb :384-441 */
        int row = mv[0].row, col = mv[0].col;
        if (clampmv) clamp_luma_mv(&row, &col, &e);
        predict_block(ry, g->y_stride, x, y, row, col, 16, 16, bil, dy);
        /* chroma MV from the (clamped) luma MV: round half away from zero, C division */
        row = (short)(row + (1 | (row >> 31)));
        col = (short)(col + (1 | (col >> 31)));
        row /= 2;
        col /= 2;
        if (fullpix) { row &= ~7; col &= ~7; }
        predict_block(ru, g->uv_stride, x / 2, y / 2, row, col, 8, 8, bil, du);
        predict_block(rv, g->uv_stride, x / 2, y / 2, row, col, 8, 8, bil, dv);
        return;
    }
    /* SPLITMV: build_inter4x4_predictors_mb :443-518.  Per-pixel results do not depend on how the
       reference groups equal-MV blocks into 8x8 / 8x4 calls, so predict 4x4 by 4x4. */
    for (int b = 0; b < 16; b++) {
        int row = mv[b].row, col = mv[b].col;
        if (clampmv) clamp_luma_mv(&row, &col, &e);
        predict_block(ry, g->y_stride, x + (b & 3) * 4, y + (b >> 2) * 4, row, col, 4, 4, bil, dy);
    }
    for (int i = 0; i < 2; i++)                  /* build_4x4uvmvs :520-558: from the UNclamped luma MVs */
        for (int j = 0; j < 2; j++) {
            int k = i * 8 + j * 2;
            int row = mv[k].row + mv[k + 1].row + mv[k + 4].row + mv[k + 5].row;
            int col = mv[k].col + mv[k + 1].col + mv[k + 4].col + mv[k + 5].col;
            row += 4 + ((row >> 31) << 3);
            col += 4 + ((col >> 31) << 3);
            row /= 8;
            col /= 8;
            if (fullpix) { row &= ~7; col &= ~7; }
            if (clampmv) clamp_chroma_mv(&row, &col, &e);
            predict_block(ru, g->uv_stride, x / 2 + j * 4, y / 2 + i * 4, row, col, 4, 4, bil, du);
            predict_block(rv, g->uv_stride, x / 2 + j * 4, y / 2 + i * 4, row, col, 4, 4, bil, dv);
        }
}

/* ========================================================================================
 * a14-a16: loop filter (loopfilter.c, loopfilter_filters.c)
 * ====================================================================================== */
static inline signed char sclamp(int t) { return (signed char)(t < -128 ? -128 : (t > 127 ? 127 : t)); }

static inline int lf_mask(int limit, int blimit, const unsigned char *s, int st)   /* vp8_filter_mask :27-40 */
{
    int p3 = s[-4 * st], p2 = s[-3 * st], p1 = s[-2 * st], p0 = s[-st];
    int q0 = s[0], q1 = s[st], q2 = s[2 * st], q3 = s[3 * st];
    int m = 0;
    m |= abs(p3 - p2) > limit;
    m |= abs(p2 - p1) > limit;
    m |= abs(p1 - p0) > limit;
    m |= abs(q1 - q0) > limit;
    m |= abs(q2 - q1) > limit;
    m |= abs(q3 - q2) > limit;
    m |= abs(p0 - q0) * 2 + abs(p1 - q1) / 2 > blimit;
    return m ? 0 : -1;                           /* 0xFF..: filter */
}

static inline int lf_hev(int thresh, const unsigned char *s, int st)              /* vp8_hevmask :43-49 */
{
    return (abs(s[-2 * st] - s[-st]) > thresh || abs(s[st] - s[0]) > thresh) ? -1 : 0;
}

static void lf_inner(unsigned char *s, int st, int mask, int hev)                  /* vp8_filter :51-95 */
{
    signed char ps1 = (signed char)(s[-2 * st] ^ 0x80), ps0 = (signed char)(s[-st] ^ 0x80);
    signed char qs0 = (signed char)(s[0] ^ 0x80), qs1 = (signed char)(s[st] ^ 0x80);
    signed char f = sclamp(ps1 - qs1), f1, f2;
    f = (signed char)(f & hev);
    f = sclamp(f + 3 * (qs0 - ps0));
    f = (signed char)(f & mask);
    f1 = sclamp(f + 4);
    f2 = sclamp(f + 3);
    f1 = (signed char)(f1 >> 3);
    f2 = (signed char)(f2 >> 3);
    s[0] = (unsigned char)(sclamp(qs0 - f1) ^ 0x80);
    s[-st] = (unsigned char)(sclamp(ps0 + f2) ^ 0x80);
    f = f1;
    f = (signed char)(f + 1);
    f = (signed char)(f >> 1);
    f = (signed char)(f & ~hev);
    s[st] = (unsigned char)(sclamp(qs1 - f) ^ 0x80);
    s[-2 * st] = (unsigned char)(sclamp(ps1 + f) ^ 0x80);
}

static void lf_mbedge(unsigned char *s, int st, int mask, int hev)                 /* vp8_mbfilter :161-214 */
{
    signed char ps2 = (signed char)(s[-3 * st] ^ 0x80), ps1 = (signed char)(s[-2 * st] ^ 0x80);
    signed char ps0 = (signed char)(s[-st] ^ 0x80), qs0 = (signed char)(s[0] ^ 0x80);
    signed char qs1 = (signed char)(s[st] ^ 0x80), qs2 = (signed char)(s[2 * st] ^ 0x80);
    signed char f = sclamp(ps1 - qs1), f1, f2, u;
    f = sclamp(f + 3 * (qs0 - ps0));
    f = (signed char)(f & mask);
    f2 = (signed char)(f & hev);
    f1 = sclamp(f2 + 4);
    f2 = sclamp(f2 + 3);
    f1 = (signed char)(f1 >> 3);
    f2 = (signed char)(f2 >> 3);
    qs0 = sclamp(qs0 - f1);
    ps0 = sclamp(ps0 + f2);
    f = (signed char)(f & ~hev);
    u = sclamp((63 + f * 27) >> 7);
    s[0] = (unsigned char)(sclamp(qs0 - u) ^ 0x80);
    s[-st] = (unsigned char)(sclamp(ps0 + u) ^ 0x80);
    u = sclamp((63 + f * 18) >> 7);
    s[st] = (unsigned char)(sclamp(qs1 - u) ^ 0x80);
    s[-2 * st] = (unsigned char)(sclamp(ps1 + u) ^ 0x80);
    u = sclamp((63 + f * 9) >> 7);
    s[2 * st] = (unsigned char)(sclamp(qs2 - u) ^ 0x80);
    s[-3 * st] = (unsigned char)(sclamp(ps2 + u) ^ 0x80);
}

/* n positions along an edge; `across` = step over the edge, `along` = step to the next position */
static void edge_normal(unsigned char *s, int across, int along, int n, int blimit, int limit, int thr, int mbedge)
{
    for (int i = 0; i < n; i++, s += along) {
        int m = lf_mask(limit, blimit, s, across), hv = lf_hev(thr, s, across);
        if (mbedge) lf_mbedge(s, across, m, hv);
        else lf_inner(s, across, m, hv);
    }
}

static void edge_simple(unsigned char *s, int across, int along, int blimit)       /* :292-355 */
{
    for (int i = 0; i < 16; i++, s += along) {
        int p1 = s[-2 * across], p0 = s[-across], q0 = s[0], q1 = s[across];
        int mask = (abs(p0 - q0) * 2 + abs(p1 - q1) / 2 <= blimit) ? -1 : 0;
        signed char sp1 = (signed char)(p1 ^ 0x80), sp0 = (signed char)(p0 ^ 0x80);
        signed char sq0 = (signed char)(q0 ^ 0x80), sq1 = (signed char)(q1 ^ 0x80);
        signed char f = sclamp(sp1 - sq1), f1, f2;
        f = sclamp(f + 3 * (sq0 - sp0));
        f = (signed char)(f & mask);
        f1 = sclamp(f + 4);
        f1 = (signed char)(f1 >> 3);
        s[0] = (unsigned char)(sclamp(sq0 - f1) ^ 0x80);
        f2 = sclamp(f + 3);
        f2 = (signed char)(f2 >> 3);
        s[-across] = (unsigned char)(sclamp(sp0 + f2) ^ 0x80);
    }
}

void vp8o_loop_filter_mbv(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *l)
{
    edge_normal(y, 1, ys, 16, l->mblim, l->lim, l->hev_thr, 1);
    if (u) edge_normal(u, 1, uvs, 8, l->mblim, l->lim, l->hev_thr, 1);
    if (v) edge_normal(v, 1, uvs, 8, l->mblim, l->lim, l->hev_thr, 1);
}
void vp8o_loop_filter_mbh(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *l)
{
    edge_normal(y, ys, 1, 16, l->mblim, l->lim, l->hev_thr, 1);
    if (u) edge_normal(u, uvs, 1, 8, l->mblim, l->lim, l->hev_thr, 1);
    if (v) edge_normal(v, uvs, 1, 8, l->mblim, l->lim, l->hev_thr, 1);
}
void vp8o_loop_filter_bv(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *l)
{
    for (int k = 4; k < 16; k += 4) edge_normal(y + k, 1, ys, 16, l->blim, l->lim, l->hev_thr, 0);
    if (u) edge_normal(u + 4, 1, uvs, 8, l->blim, l->lim, l->hev_thr, 0);
    if (v) edge_normal(v + 4, 1, uvs, 8, l->blim, l->lim, l->hev_thr, 0);
}
void vp8o_loop_filter_bh(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const vp8o_lf_info *l)
{
    for (int k = 4; k < 16; k += 4) edge_normal(y + k * ys, ys, 1, 16, l->blim, l->lim, l->hev_thr, 0);
    if (u) edge_normal(u + 4 * uvs, uvs, 1, 8, l->blim, l->lim, l->hev_thr, 0);
    if (v) edge_normal(v + 4 * uvs, uvs, 1, 8, l->blim, l->lim, l->hev_thr, 0);
}
void vp8o_loop_filter_simple_mbv(unsigned char *y, int ys, unsigned char b) { edge_simple(y, 1, ys, b); }
void vp8o_loop_filter_simple_mbh(unsigned char *y, int ys, unsigned char b) { edge_simple(y, ys, 1, b); }
void vp8o_loop_filter_simple_bv(unsigned char *y, int ys, unsigned char b)
{
    for (int k = 4; k < 16; k += 4) edge_simple(y + k, 1, ys, b);
}
void vp8o_loop_filter_simple_bh(unsigned char *y, int ys, unsigned char b)
{
    for (int k = 4; k < 16; k += 4) edge_simple(y + k * ys, ys, 1, b);
}

static inline int lvl_clamp(int v) { return v < 0 ? 0 : (v > 63 ? 63 : v); }

void vp8o_lf_levels(const vp8ir_frame_hdr *h, unsigned char lvl[4][4][4])           /* loopfilter.c:117-201 */
{
    for (int seg = 0; seg < 4; seg++) {
        int base = h->filter_level;
        if (h->segmentation_enabled) {
            if (h->mb_segment_abs_delta) base = h->segment_lf[seg];
            else base = lvl_clamp(base + h->segment_lf[seg]);
        }
        if (!h->mode_ref_lf_delta_enabled) {
            /* reference memsets with the unclamped value (a negative absolute level would wrap);
               conforming encoders keep it in 0..63 */
            memset(lvl[seg], base & 0xff, 16);
            continue;
        }
        {
            int r = base + h->ref_lf_deltas[VP8IR_INTRA_FRAME];
            lvl[seg][0][0] = (unsigned char)lvl_clamp(r + h->mode_lf_deltas[0]);   /* B_PRED */
            lvl[seg][0][1] = (unsigned char)lvl_clamp(r);                          /* other intra */
            lvl[seg][0][2] = lvl[seg][0][3] = 0;                                   /* never indexed */
        }
        for (int ref = 1; ref < 4; ref++) {
            int r = base + h->ref_lf_deltas[ref];
            lvl[seg][ref][0] = 0;                                                  /* never indexed */
            for (int m = 1; m < 4; m++) lvl[seg][ref][m] = (unsigned char)lvl_clamp(r + h->mode_lf_deltas[m]);
        }
    }
}

void vp8o_lf_limits(int sharp, int level, int frame_type, vp8o_lf_info *l)          /* loopfilter.c:24-96 */
{
    int ilimit = level >> (sharp > 0);
    ilimit >>= (sharp > 4);
    if (sharp > 0 && ilimit > 9 - sharp) ilimit = 9 - sharp;
    if (ilimit < 1) ilimit = 1;
    l->lim = (unsigned char)ilimit;
    l->blim = (unsigned char)(2 * level + ilimit);
    l->mblim = (unsigned char)(2 * (level + 2) + ilimit);
    if (level >= 40) l->hev_thr = frame_type == 0 ? 2 : 3;
    else if (level >= 20) l->hev_thr = frame_type == 0 ? 1 : 2;
    else if (level >= 15) l->hev_thr = 1;
    else l->hev_thr = 0;
}

static const unsigned char mode_lf_index[10] = { 1, 1, 1, 1, 0, 2, 2, 1, 2, 3 };   /* lf_init_lut: mode_lf_lut */

static void loop_filter_frame(const vp8ir_frame_hdr *h, const vp8ir_geom *g, const vp8ir_mb *mbs, uint8_t *frame)
{
    unsigned char lvl[4][4][4];
    vp8o_lf_levels(h, lvl);
    for (int r = 0; r < h->mb_rows; r++)
        for (int c = 0; c < h->mb_cols; c++) {
            const vp8ir_mb *mb = &mbs[r * h->mb_cols + c];
            int skip_lf = mb->y_mode != VP8IR_B_PRED && mb->y_mode != VP8IR_SPLITMV && (mb->flags & VP8IR_MB_SKIP);
            int level = lvl[mb->segment_id][mb->ref_frame][mode_lf_index[mb->y_mode]];
            unsigned char *y = frame + g->y_off + r * 16 * g->y_stride + c * 16;
            unsigned char *u = frame + g->u_off + r * 8 * g->uv_stride + c * 8;
            unsigned char *v = frame + g->v_off + r * 8 * g->uv_stride + c * 8;
            vp8o_lf_info l;
            if (!level) continue;
            vp8o_lf_limits(h->sharpness_level, level, vp8ir_lf_frame_type(h), &l);
            if (h->filter_type == 0) {
                if (c > 0) vp8o_loop_filter_mbv(y, u, v, g->y_stride, g->uv_stride, &l);
                if (!skip_lf) vp8o_loop_filter_bv(y, u, v, g->y_stride, g->uv_stride, &l);
                if (r > 0) vp8o_loop_filter_mbh(y, u, v, g->y_stride, g->uv_stride, &l);
                if (!skip_lf) vp8o_loop_filter_bh(y, u, v, g->y_stride, g->uv_stride, &l);
            } else {
                if (c > 0) vp8o_loop_filter_simple_mbv(y, g->y_stride, l.mblim);
                if (!skip_lf) vp8o_loop_filter_simple_bv(y, g->y_stride, l.blim);
                if (r > 0) vp8o_loop_filter_simple_mbh(y, g->y_stride, l.mblim);
                if (!skip_lf) vp8o_loop_filter_simple_bh(y, g->y_stride, l.blim);
            }
        }
}

/* ========================================================================================
 * a17: border handling
 * ====================================================================================== */
static void seed_intra_borders(const vp8ir_geom *g, uint8_t *f)           /* setupintrarecon.c:15-32 */
{
    uint8_t *p[3] = { f + g->y_off, f + g->u_off, f + g->v_off };
    int st[3] = { g->y_stride, g->uv_stride, g->uv_stride };
    int w[3] = { g->aligned_w, g->aligned_w / 2, g->aligned_w / 2 };
    int hh[3] = { g->aligned_h, g->aligned_h / 2, g->aligned_h / 2 };
    for (int k = 0; k < 3; k++) {
        memset(p[k] - 1 - st[k], 127, (size_t)w[k] + 5);
        for (int i = 0; i < hh[k]; i++) p[k][st[k] * i - 1] = 129;
    }
}

static void extend_plane(uint8_t *p, int stride, int w, int h, int border)          /* yv12extend.c:24-145 */
{
    for (int r = 0; r < h; r++) {
        memset(p + r * stride - border, p[r * stride], (size_t)border);
        memset(p + r * stride + w, p[r * stride + w - 1], (size_t)border);
    }
    for (int i = 1; i <= border; i++) {
        memcpy(p - border - i * stride, p - border, (size_t)w + 2 * border);
        memcpy(p - border + (h - 1 + i) * stride, p - border + (h - 1) * stride, (size_t)w + 2 * border);
    }
}

/* ========================================================================================
 * the whole-frame driver
 * ====================================================================================== */
void vp8o_decode_frame(const vp8ir_frame_hdr *h, const vp8ir_mb *mbs, const int16_t *coef, const vp8ir_mv *mvs,
                       uint8_t *dst, const uint8_t *const refs[4], int stages)
{
    vp8ir_geom g;
    vp8ir_geom_init(&g, h->width, h->height);
    uint8_t *Y = dst + g.y_off, *U = dst + g.u_off, *V = dst + g.v_off;

    if (stages & VP8O_STAGE_RECON) {
        seed_intra_borders(&g, dst);
        for (int r = 0; r < h->mb_rows; r++) {
            for (int c = 0; c < h->mb_cols; c++) {
                int n = r * h->mb_cols + c;
                const vp8ir_mb *mb = &mbs[n];
                uint8_t *y = Y + r * 16 * g.y_stride + c * 16;
                uint8_t *u = U + r * 8 * g.uv_stride + c * 8, *v = V + r * 8 * g.uv_stride + c * 8;
                short q[400], dqy[16], dqy_dc1[16], dquv[16], dqy2[16];
                char eobs[25];
                int skip = mb->flags & VP8IR_MB_SKIP;
                vp8o_dequant dq;
                vp8o_mb_dequant(h, mb->segment_id, &dq);
                for (int i = 0; i < 16; i++) {
                    dqy[i] = dq.y1[i != 0];
                    dqy_dc1[i] = i ? dq.y1[1] : 1;     /* dequant_y1_dc: DC factor 1 (decodframe.c:92) */
                    dqy2[i] = dq.y2[i != 0];
                    dquv[i] = dq.uv[i != 0];
                }
                memset(q, 0, sizeof q);
                memset(eobs, 0, sizeof eobs);
                if (!skip) {                     /* IR blocks are column-major; the RTCD layer is raster */
                    const int16_t *cq = coef + (size_t)n * VP8IR_COEF_PER_MB;
                    for (int b = 0; b < 25; b++)
                        for (int i = 0; i < 16; i++) q[b * 16 + (i & 3) * 4 + (i >> 2)] = cq[b * 16 + i];
                    memcpy(eobs, mb->eobs, 25);
                }

                if (mb->ref_frame == VP8IR_INTRA_FRAME) {
                    intra_pred_plane(u, g.uv_stride, 8, mb->uv_mode, r > 0, c > 0);
                    intra_pred_plane(v, g.uv_stride, 8, mb->uv_mode, r > 0, c > 0);
                    if (mb->y_mode != VP8IR_B_PRED)
                        intra_pred_plane(y, g.y_stride, 16, mb->y_mode, r > 0, c > 0);
                    else {
                        /* decodframe.c:200-236.  Above-right of the right-hand block column is the MB's
                           own above-right row for all four block rows (the reference implements this
                           by the down-copy of reconintra4x4.c:305-317; read-only here). */
                        const uint8_t *mb_above_right = y - g.y_stride + 16;
                        for (int b = 0; b < 16; b++) {
                            uint8_t *d = y + (b >> 2) * 4 * g.y_stride + (b & 3) * 4;
                            unsigned char above[8], left[4];
                            memcpy(above, d - g.y_stride, 4);
                            memcpy(above + 4, (b & 3) == 3 ? mb_above_right : d - g.y_stride + 4, 4);
                            for (int i = 0; i < 4; i++) left[i] = d[i * g.y_stride - 1];
                            vp8o_intra4x4_predict(above, left, d[-g.y_stride - 1], mb->b_modes[b], d, g.y_stride);
                            if (eobs[b]) {
                                if (eobs[b] > 1) vp8o_dequant_idct_add(q + 16 * b, dqy, d, g.y_stride);
                                else vp8o_dc_only_idct_add((short)(q[16 * b] * dqy[0]), d, g.y_stride, d, g.y_stride);
                            }
                        }
                    }
                } else
                    inter_predict_mb(h, &g, mb, mvs + (size_t)n * 16, refs[mb->ref_frame], dst, r, c);

                if (!skip) {                     /* decodframe.c:252-304 */
                    if (mb->y_mode != VP8IR_B_PRED) {
                        const short *dqc = dqy;
                        if (mb->y_mode != VP8IR_SPLITMV) {
                            short y2[16];
                            if (eobs[24] > 1) {
                                vp8o_dequantize_b(q + 384, dqy2, y2);
                                vp8o_short_inv_walsh4x4(y2, q);
                            } else {
                                y2[0] = (short)(q[384] * dqy2[0]);
                                vp8o_short_inv_walsh4x4_1(y2, q);
                            }
                            dqc = dqy_dc1;
                        }
                        vp8o_dequant_idct_add_y_block(q, dqc, y, g.y_stride, eobs);
                    }
                    vp8o_dequant_idct_add_uv_block(q + 256, dquv, u, v, g.uv_stride, eobs + 16);
                }
            }
            /* vp8_extend_mb_row (extend.c:160-185): 4 pixels right of the last two rows of the MB row */
            for (int k = 14; k < 16; k++) {
                uint8_t *p = Y + (r * 16 + k) * g.y_stride + g.aligned_w;
                memset(p, p[-1], 4);
            }
            for (int k = 6; k < 8; k++) {
                uint8_t *pu = U + (r * 8 + k) * g.uv_stride + g.aligned_w / 2;
                uint8_t *pv = V + (r * 8 + k) * g.uv_stride + g.aligned_w / 2;
                memset(pu, pu[-1], 4);
                memset(pv, pv[-1], 4);
            }
        }
    }
    if ((stages & VP8O_STAGE_LF) && h->filter_level)
        loop_filter_frame(h, &g, mbs, dst);
    if (stages & VP8O_STAGE_EXTEND) {
        extend_plane(Y, g.y_stride, g.aligned_w, g.aligned_h, VP8IR_BORDER);
        extend_plane(U, g.uv_stride, g.aligned_w / 2, g.aligned_h / 2, VP8IR_BORDER / 2);
        extend_plane(V, g.uv_stride, g.aligned_w / 2, g.aligned_h / 2, VP8IR_BORDER / 2);
    }
}
