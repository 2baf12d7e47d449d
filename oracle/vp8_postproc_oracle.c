/* oracle/vp8_postproc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See vp8_oracle.h.
 *
 * CPU restatement of the reference's output-side post-processing filters (vp8/common/postproc.c): the deblocking and
 * demacroblocking filters, the noise adder, and vp8_post_proc_frame's choice of their strengths.  The reference filters in
 * place through small ring buffers that delay every write until the pixel can no longer be read; each filter is therefore a
 * pure function of its input plane, and is written that way here (input plane -> output plane).  Pinned against the
 * reference's own functions in tests/test_oracle_vs_ref.py and against its vpxdec in tests/test_oracle_golden.py.
 */
#include "vp8_oracle.h"
#include "vp8o_pp_rv.h"

#include <stdlib.h>

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One tap set of vp8_post_proc_down_and_across_c (postproc.c:132-221): weights 1 1 4 1 1, rounded by 4, >> 3, and the
 * centre pixel is kept when any of the five differs from it by more than flimit. */
static int pp_five(int v, const int p[5], int flimit)
{
    int k = 4;
    for (int i = 0; i < 5; i++) {
        if (iabs(v - p[i]) > flimit) return v;
        k += (i == 2 ? 4 : 1) * p[i];
    }
    return k >> 3;
}

/* postproc.c:132-221.  Vertical pass over the source (rows -2..+2 must be readable: the decoder's frames have borders),
 * then the horizontal pass over the vertically filtered row, whose ends are replicated (:176-180). */
void vp8o_post_proc_down_and_across(const unsigned char *src, unsigned char *dst, int src_stride, int dst_stride,
                                    int rows, int cols, int flimit)
{
    int *down = (int *)malloc(sizeof(int) * (size_t)cols);
    for (int r = 0; r < rows; r++) {
        const unsigned char *s = src + (long)r * src_stride;
        for (int c = 0; c < cols; c++) {
            int p[5];
            for (int i = 0; i < 5; i++) p[i] = s[c + (i - 2) * src_stride];
            down[c] = pp_five(s[c], p, flimit);
        }
        for (int c = 0; c < cols; c++) {
            int p[5];
            for (int i = 0; i < 5; i++) p[i] = down[clampi(c + i - 2, 0, cols - 1)];
            dst[(long)r * dst_stride + c] = (unsigned char)pp_five(down[c], p, flimit);
        }
    }
    free(down);
}

/* vp8_mbpost_proc_across_ip_c (postproc.c:230-277): a pixel becomes the rounded mean of itself and the 15 pixels
 * centred on it (ends replicated) where that window is flat: 15 * sum of squares - square of sum < flimit. */
void vp8o_mbpost_proc_across(const unsigned char *src, unsigned char *dst, int stride, int rows, int cols, int flimit)
{
    for (int r = 0; r < rows; r++) {
        const unsigned char *s = src + (long)r * stride;
        for (int c = 0; c < cols; c++) {
            int sum = 0, sumsq = 0;
            for (int i = -7; i <= 7; i++) {
                int v = s[clampi(c + i, 0, cols - 1)];
                sum += v;
                sumsq += v * v;
            }
            dst[(long)r * stride + c] = (unsigned char)(sumsq * 15 - sum * sum < flimit ? (8 + sum + s[c]) >> 4 : s[c]);
        }
    }
}

/* vp8_mbpost_proc_down_c (postproc.c:283-325): the same along columns, rounded by the dither table instead of 8.
 * rv_offset = 63 & rand(), drawn once per call by the reference (:286). */
void vp8o_mbpost_proc_down(const unsigned char *src, unsigned char *dst, int stride, int rows, int cols, int flimit, int rv_offset)
{
    for (int c = 0; c < cols; c++) {
        const short *rv2 = vp8o_pp_rv + rv_offset + ((c * 17) & 127);
        for (int r = 0; r < rows; r++) {
            int sum = 0, sumsq = 0;
            for (int i = -7; i <= 7; i++) {
                int v = src[(long)clampi(r + i, 0, rows - 1) * stride + c];
                sum += v;
                sumsq += v * v;
            }
            int v = src[(long)r * stride + c];
            dst[(long)r * stride + c] = (unsigned char)(sumsq * 15 - sum * sum < flimit ? (rv2[r & 127] + sum + v) >> 4 : v);
        }
    }
}

/* vp8_plane_add_noise_c (postproc.c:489-513): clamp away from black and white by `clamp` (blackclamp[0] == whiteclamp[0],
 * :471-476), then add the noise row that starts row_offset[r] = rand() & 0xff into the 3072-entry table; the sum wraps. */
void vp8o_plane_add_noise(unsigned char *plane, const signed char *noise, int clamp, int width, int height, int stride,
                          const unsigned char *row_offset)
{
    for (int r = 0; r < height; r++) {
        unsigned char *p = plane + (long)r * stride;
        const signed char *ref = noise + row_offset[r];
        for (int c = 0; c < width; c++) {
            int v = p[c];
            if (v < clamp) v = clamp;
            if (v > 255 + (signed char)clamp) v = 255 + (signed char)clamp;
            p[c] = (unsigned char)(v + ref[c]);
        }
    }
}

/* The strengths vp8_post_proc_frame hands to the filters (postproc.c:903-1000, 328-362, 223-228) for a frame with
 * loop-filter level `filter_level`: *q (:905,913), *ppl = the flimit of down_and_across for DEBLOCK, *ppl_demacro and
 * *mb_flimit for DEMACROBLOCK at `deblocking_level`. */
void vp8o_pp_strengths(int filter_level, int deblocking_level, int *q, int *ppl, int *ppl_demacro, int *mb_flimit)
{
    int qq = filter_level * 10 / 6;
    if (qq > 63) qq = 63;
    *q = qq;
    {
        double level = 6.0e-05 * qq * qq * qq - .0067 * qq * qq + .306 * qq + .0065;
        *ppl = (int)(level + .5);
    }
    {
        int qd = qq + (deblocking_level - 5) * 10, x = qd;
        double level = 6.0e-05 * qd * qd * qd - .0067 * qd * qd + .306 * qd + .0065;
        *ppl_demacro = (int)(level + .5);
        if (x < 20) x = 20;
        x = 50 + (x - 50) * 10 / 8;
        *mb_flimit = x * x / 3;
    }
}

#include <math.h>
/* fillrd (postproc.c:410-465): a 256-entry table distributed like a gaussian of width sigma(q, a), sampled 3072 times
 * through r[i] = rand() & 0xff.  q is the function's argument (vp8_post_proc_frame passes 63 - its own q, :993). */
void vp8o_pp_noise_table(int q, int a, const unsigned char *r, signed char noise[3072], int *clamp)
{
    signed char dist[300];
    const double sigma = a + .5 + .6 * (63 - q) / 63.0;
    int next = 0;
    for (int i = -32; i < 32; i++) {
        const double x = i;
        const int n = (int)(.5 + 256 * (1 / (sigma * sqrt(2.0 * 3.14159265)) * exp(-x * x / (2 * sigma * sigma))));
        for (int j = 0; j < n; j++) dist[next + j] = (signed char)i;
        next += n > 0 ? n : 0;
    }
    for (; next < 256; next++) dist[next] = 0;
    for (int i = 0; i < 3072; i++) noise[i] = dist[r[i]];
    *clamp = -dist[0];
}

/* ---- multiframe quality enhancement (VP8_MFQE) ------------------------------------------------------------------------------
 * postproc.c:696-800 (multiframe_quality_enhance_block) for one luma block of `bs` x `bs` pixels (16 or 8) and the chroma
 * blocks of half that size under it: `y`, `u`, `v` the frame about to be shown, `yd`, `ud`, `vd` the post-processing buffer,
 * which still holds what was shown before.  Where the two differ little for the activity of the old picture and the step in
 * quantiser, the old picture is kept or blended with the new one; elsewhere the new one is copied.
 * The activity is vp8_variance16x16 / 8x8 against a row of zeros (encoder/variance_c.c:34-79,115-127): sum of squares minus
 * sum * sum >> 8 (>> 6), and the reference forms sum * sum in a signed int.  A 16x16 block brighter than 181 on average
 * overflows it; the reference build wraps (two's complement imul, arithmetic shift), which is restated here and what the
 * listings printed by that build pin. */
static void mfqe_block(int bs, int qcurr, int qprev, const unsigned char *y, const unsigned char *u, const unsigned char *v,
                       int y_stride, int uv_stride, unsigned char *yd, unsigned char *ud, unsigned char *vd, int yd_stride,
                       int uvd_stride)
{
    const int half = bs >> 1, qdiff = qcurr - qprev;
    const int sh = bs == 16 ? 8 : 6, rnd = 1 << (sh - 1);
    unsigned int sse = 0, sad = 0, act, thr;
    int sum = 0;
    for (int i = 0; i < bs; i++)
        for (int j = 0; j < bs; j++) {
            const int d = yd[i * yd_stride + j];
            sum += d;
            sse += (unsigned)(d * d);
            sad += (unsigned)iabs(y[i * y_stride + j] - d);
        }
    {
        const int sq = (int)((unsigned)sum * (unsigned)sum);         /* avg * avg in an int: wraps for sum >= 46341 */
        const int sq_sh = sq >= 0 ? sq >> sh : -((-(long)sq + (1L << sh) - 1) >> sh);     /* arithmetic shift, spelled out */
        act = (sse - (unsigned)sq_sh + (unsigned)rnd) >> sh;
    }
    sad = (sad + (unsigned)rnd) >> sh;
    thr = (unsigned)(qdiff >> 3);                                    /* thr = qdiff / 8 + log2(act) + log4(qprev) */
    while (act >>= 1) thr++;
    while (qprev >>= 2) thr++;
    if (sad < thr) {
        int ifactor = (int)((sad << 4) / thr);
        ifactor >>= (qdiff >> 5);
        if (ifactor) {
            const int ic = 16 - ifactor;
            for (int i = 0; i < bs; i++)
                for (int j = 0; j < bs; j++)
                    yd[i * yd_stride + j] = (unsigned char)((y[i * y_stride + j] * ifactor + yd[i * yd_stride + j] * ic + 8) >> 4);
            for (int i = 0; i < half; i++)
                for (int j = 0; j < half; j++) {
                    ud[i * uvd_stride + j] = (unsigned char)((u[i * uv_stride + j] * ifactor + ud[i * uvd_stride + j] * ic + 8) >> 4);
                    vd[i * uvd_stride + j] = (unsigned char)((v[i * uv_stride + j] * ifactor + vd[i * uvd_stride + j] * ic + 8) >> 4);
                }
        }                                                            /* ifactor 0: the old picture stays */
    } else {
        for (int i = 0; i < bs; i++)
            for (int j = 0; j < bs; j++) yd[i * yd_stride + j] = y[i * y_stride + j];
        for (int i = 0; i < half; i++)
            for (int j = 0; j < half; j++) {
                ud[i * uvd_stride + j] = u[i * uv_stride + j];
                vd[i * uvd_stride + j] = v[i * uv_stride + j];
            }
    }
}

/* vp8_multiframe_quality_enhance (postproc.c:802-900) over frame buffers laid out by vp8ir_geom: `show` the decoded frame,
 * `dest` the post-processing buffer (in place).  Per macroblock: key frames and macroblocks that moved by at most 10 (in the
 * units the vectors are stored in) in both directions are enhanced -- B_PRED and SPLITMV macroblocks as four 8x8 blocks, the
 * others whole -- the rest is copied.  mbs / mvs: the frame's IR (mvs may be NULL on key frames; the macroblock's vector is
 * that of its last block: decodemv.c:490, and zero for intra macroblocks, :563). */
void vp8o_mfqe(const vp8ir_frame_hdr *hdr, const vp8ir_geom *g, const vp8ir_mb *mbs, const vp8ir_mv *mvs,
               const unsigned char *show, unsigned char *dest, int qcurr, int qprev)
{
    for (int r = 0; r < hdr->mb_rows; r++)
        for (int c = 0; c < hdr->mb_cols; c++) {
            const int n = r * hdr->mb_cols + c;
            const long yo = g->y_off + (long)16 * r * g->y_stride + 16 * c, uo = (long)8 * r * g->uv_stride + 8 * c;
            const unsigned char *y = show + yo, *u = show + g->u_off + uo, *v = show + g->v_off + uo;
            unsigned char *yd = dest + yo, *ud = dest + g->u_off + uo, *vd = dest + g->v_off + uo;
            int still = hdr->frame_type == 0;
            if (!still) {
                const int mr = mbs[n].ref_frame == VP8IR_INTRA_FRAME ? 0 : mvs[n * 16 + 15].row;
                const int mc = mbs[n].ref_frame == VP8IR_INTRA_FRAME ? 0 : mvs[n * 16 + 15].col;
                still = iabs(mr) <= 10 && iabs(mc) <= 10;
            }
            if (!still) {
                for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) yd[i * g->y_stride + j] = y[i * g->y_stride + j];
                for (int i = 0; i < 8; i++)
                    for (int j = 0; j < 8; j++) {
                        ud[i * g->uv_stride + j] = u[i * g->uv_stride + j];
                        vd[i * g->uv_stride + j] = v[i * g->uv_stride + j];
                    }
            } else if (mbs[n].y_mode == VP8IR_B_PRED || mbs[n].y_mode == VP8IR_SPLITMV) {
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++)
                        mfqe_block(8, qcurr, qprev, y + 8 * (i * g->y_stride + j), u + 4 * (i * g->uv_stride + j),
                                   v + 4 * (i * g->uv_stride + j), g->y_stride, g->uv_stride, yd + 8 * (i * g->y_stride + j),
                                   ud + 4 * (i * g->uv_stride + j), vd + 4 * (i * g->uv_stride + j), g->y_stride, g->uv_stride);
            } else
                mfqe_block(16, qcurr, qprev, y, u, v, g->y_stride, g->uv_stride, yd, ud, vd, g->y_stride, g->uv_stride);
        }
}
