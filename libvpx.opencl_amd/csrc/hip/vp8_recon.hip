// VP8 reconstruction kernel for gfx950: dequant + inverse DCT/WHT, intra and inter prediction,
// residual add.  Replaces the per-macroblock loop of the reference decoder,
//   decode_mb_row / decode_macroblock            vp8/decoder/decodframe.c:112-436
// and everything it calls through RTCD:
//   vp8_dequantize_b, vp8_dequant_idct_add, vp8_dc_only_idct_add      vp8/common/dequantize.c, idctllm.c
//   vp8_short_inv_walsh4x4(_1), vp8_dequant_idct_add_{y,uv}_block     vp8/common/idctllm.c, idct_blk.c
//   vp8_build_intra_predictors_mb{y,uv}_s, vp8_intra4x4_predict       vp8/common/reconintra.c, reconintra4x4.c
//   vp8_build_inter_predictors_mb, sixtap / bilinear / copy_mem       vp8/common/reconinter.c, filter.c
//   vp8_setup_intra_recon, vp8_extend_mb_row (edge rules only)        vp8/common/setupintrarecon.c, extend.c
//
// Mapping (MI355X-first, not the reference's per-block launches):
//   * one workgroup = one frame at a time, persistent over the jobs of a launch
//     (job = blockIdx.x, blockIdx.x + gridDim.x, ...): intra prediction chains every MB to its
//     left / above / above-right neighbours, so a frame is a wavefront-parallel problem, and the
//     chip is filled with FRAMES (256 CUs -> hundreds of frames in flight), not with MBs.
//   * one wave = one MB row, marching left to right; wave w owns rows w, w+NW, w+2NW, ...  The
//     row above must be two MBs ahead (above-right pixels of B_PRED); progress is exchanged
//     through LDS flags, never through global memory, never across CUs.
//   * unfiltered neighbour pixels travel through LDS: the bottom pixel line of every MB row sits in
//     a per-wave LDS line buffer, the left column stays in the wave's LDS tile.  The frame in HBM
//     is written exactly once per pixel and never read back by this kernel; coefficients are read
//     exactly once (coalesced 8-byte-per-lane loads, lane = 4x4 block column).
//   * integer only (u8 pixels, i16 coefficients, i32 accumulators); no MFMA by design.
#include "vp8_common.hip.h"

// ---- 4x4 intra predictor table (same encoding as the oracle's, derived from
// vp8/common/reconintra4x4.c:16-303): edge vector P[15] = {L3,L3,L2,L1,L0,TL,A0..A7,A7};
// entry = kind<<4 | k with kind 0 copy, 1 (P[k]+P[k+1]+1)>>1, 2 (P[k-1]+2P[k]+P[k+1]+2)>>2.
#define C_(k) (0x00 | (k))
#define A_(k) (0x10 | (k))
#define F_(k) (0x20 | (k))
__constant__ static const unsigned char k_bpred_tab[10 * 16] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    F_(6), F_(7), F_(8), F_(9), F_(6), F_(7), F_(8), F_(9), F_(6), F_(7), F_(8), F_(9), F_(6), F_(7), F_(8), F_(9),
    F_(4), F_(4), F_(4), F_(4), F_(3), F_(3), F_(3), F_(3), F_(2), F_(2), F_(2), F_(2), F_(1), F_(1), F_(1), F_(1),
    F_(7), F_(8), F_(9), F_(10), F_(8), F_(9), F_(10), F_(11), F_(9), F_(10), F_(11), F_(12), F_(10), F_(11), F_(12), F_(13),
    F_(5), F_(6), F_(7), F_(8), F_(4), F_(5), F_(6), F_(7), F_(3), F_(4), F_(5), F_(6), F_(2), F_(3), F_(4), F_(5),
    A_(5), A_(6), A_(7), A_(8), F_(5), F_(6), F_(7), F_(8), F_(4), A_(5), A_(6), A_(7), F_(3), F_(5), F_(6), F_(7),
    A_(6), A_(7), A_(8), A_(9), F_(7), F_(8), F_(9), F_(10), A_(7), A_(8), A_(9), F_(11), F_(8), F_(9), F_(10), F_(12),
    A_(4), F_(5), F_(6), F_(7), A_(3), F_(4), A_(4), F_(5), A_(2), F_(3), A_(3), F_(4), A_(1), F_(2), A_(2), F_(3),
    A_(3), F_(3), A_(2), F_(2), A_(2), F_(2), A_(1), F_(1), A_(1), F_(1), C_(1), C_(1), C_(1), C_(1), C_(1), C_(1),
};
#undef C_
#undef A_
#undef F_

// sub-pixel filter taps (vp8/common/filter.c:16-39)
__constant__ static const short k_sixtap[8][6] = {
    { 0, 0, 128, 0, 0, 0 }, { 0, -6, 123, 12, -1, 0 }, { 2, -11, 108, 36, -8, 1 }, { 0, -9, 93, 50, -6, 0 },
    { 3, -16, 77, 77, -16, 3 }, { 0, -6, 50, 93, -9, 0 }, { 1, -8, 36, 108, -11, 2 }, { 0, -1, 12, 123, -6, 0 }
};

// ---- per-wave LDS working set ----------------------------------------------------------------
// Tiles hold the MB being reconstructed plus its prediction edges:
//   tY: 17 rows (y = -1..15) x 24 cols (x = -4..19); x = -1 is the left column, x = 16..19 of
//       row -1 the above-right pixels.  tU/tV: 9 rows x 12 cols (x = -4..7).
#define TY_STRIDE 24
#define TC_STRIDE 12
#define TY_AT(y, x) (((y) + 1) * TY_STRIDE + (x) + 4)
#define TC_AT(y, x) (((y) + 1) * TC_STRIDE + (x) + 4)
struct __attribute__((aligned(16))) WaveLds {
    unsigned char tY[17 * TY_STRIDE];   // 408
    unsigned char tU[9 * TC_STRIDE];    // 108
    unsigned char tV[9 * TC_STRIDE];    // 108  -> 624
    short res[384];                     // residual, pixel order: [blk][row][col]      -> 1392
    short tr[400];                      // IDCT transpose scratch: [blk][row][col]      -> 2192
    short wht_dc[16];                   // Y2 -> per-block DC                           -> 2224
    short dq[4][6];                     // per segment: y1dc,y1ac,y2dc,y2ac,uvdc,uvac   -> 2272
    unsigned char pad[16];              //                                              -> 2288
};
static_assert(sizeof(WaveLds) % 16 == 0, "WaveLds alignment");

#define LINE_PAD 16   // line[LINE_PAD + x]; x = -1 valid (left border), x up to W+3 valid

__device__ __forceinline__ int line_bytes(int aligned_w) { return 2 * aligned_w + 6 * LINE_PAD; }

// vp8cx_init_de_quantizer + mb_init_dequantizer (vp8/decoder/decodframe.c:50-109,
// vp8/common/quant_common.c:39-132): six factors per segment.
__device__ __forceinline__ void build_dequant(const vp8ir_frame_hdr &h, short (*dq)[6], int lane)
{
    if (lane < 24) {
        int seg = lane / 6, k = lane % 6;
        int q = h.base_qindex;
        if (h.segmentation_enabled) {
            if (h.mb_segment_abs_delta) q = h.segment_quant[seg];
            else q = q + h.segment_quant[seg];
        }
        q = q < 0 ? 0 : (q > 127 ? 127 : q);
        int delta = k == 0 ? h.y1dc_delta_q : k == 2 ? h.y2dc_delta_q : k == 3 ? h.y2ac_delta_q
                  : k == 4 ? h.uvdc_delta_q : k == 5 ? h.uvac_delta_q : 0;
        int qi = q + delta;
        qi = qi < 0 ? 0 : (qi > 127 ? 127 : qi);
        int v;
        if (k == 0) v = k_dc_q[qi];
        else if (k == 1) v = k_ac_q[qi];
        else if (k == 2) v = k_dc_q[qi] * 2;
        else if (k == 3) { v = (k_ac_q[qi] * 155) / 100; if (v < 8) v = 8; }
        else if (k == 4) { v = k_dc_q[qi]; if (v > 132) v = 132; }
        else v = k_ac_q[qi];
        dq[seg][k] = (short)v;
    }
}

// One column (vertical) pass of vp8_short_idct4x4llm_c (idctllm.c:39-60); outputs truncated to i16.
__device__ __forceinline__ void idct_col(int i0, int i1, int i2, int i3, short o[4])
{
    int a1 = i0 + i2, b1 = i0 - i2;
    int c1 = ((i1 * 35468) >> 16) - (i3 + ((i3 * 20091) >> 16));
    int d1 = (i1 + ((i1 * 20091) >> 16)) + ((i3 * 35468) >> 16);
    o[0] = (short)(a1 + d1);
    o[3] = (short)(a1 - d1);
    o[1] = (short)(b1 + c1);
    o[2] = (short)(b1 - c1);
}
// Row (horizontal) pass with the (x+4)>>3 rounding (idctllm.c:65-88).
__device__ __forceinline__ void idct_row(int t0, int t1, int t2, int t3, short o[4])
{
    int a1 = t0 + t2, b1 = t0 - t2;
    int c1 = ((t1 * 35468) >> 16) - (t3 + ((t3 * 20091) >> 16));
    int d1 = (t1 + ((t1 * 20091) >> 16)) + ((t3 * 35468) >> 16);
    o[0] = (short)((a1 + d1 + 4) >> 3);
    o[3] = (short)((a1 - d1 + 4) >> 3);
    o[1] = (short)((b1 + c1 + 4) >> 3);
    o[2] = (short)((b1 - c1 + 4) >> 3);
}

struct short4v { short x, y, z, w; };

// Residual of one MB -> wl->res (all 24 blocks, zero-filled for skipped MBs).
// qY: lane = Y block*4 + column; qC: lanes 0..31 = U/V block*4 + column, lanes 32..35 = Y2 columns.
__device__ __forceinline__ void compute_residual(WaveLds *wl, int lane, bool skip, bool has_y2, int seg,
                                                 short4v qY, short4v qC)
{
    short *res = wl->res, *tr = wl->tr;
    if (skip) {
        // 768 bytes of zeros: 64 lanes x 12 bytes
        int *z = (int *)res;
        z[lane] = 0; z[64 + lane] = 0; z[128 + lane] = 0;
        wave_lds_sync();
        return;
    }
    const short *dq = wl->dq[seg];
    const int col = lane & 3;
    // ---- chroma blocks + Y2: column pass
    if (lane < 36) {
        short o[4];
        if (lane < 32) {
            int f0 = col == 0 ? dq[4] : dq[5], fa = dq[5];
            idct_col((short)(qC.x * f0), (short)(qC.y * fa), (short)(qC.z * fa), (short)(qC.w * fa), o);
        } else {   // vp8_dequantize_b + first loop of vp8_short_inv_walsh4x4_c (idctllm.c:150-163)
            int f0 = col == 0 ? dq[2] : dq[3], fa = dq[3];
            int i0 = (short)(qC.x * f0), i1 = (short)(qC.y * fa), i2 = (short)(qC.z * fa), i3 = (short)(qC.w * fa);
            int a1 = i0 + i3, b1 = i1 + i2, c1 = i1 - i2, d1 = i0 - i3;
            o[0] = (short)(a1 + b1); o[1] = (short)(c1 + d1); o[2] = (short)(a1 - b1); o[3] = (short)(d1 - c1);
        }
        short *t = tr + 256 + (lane >> 2) * 16 + col;      // [blk][row][col]
        t[0] = o[0]; t[4] = o[1]; t[8] = o[2]; t[12] = o[3];
    }
    wave_lds_sync();
    if (lane < 36) {
        const short *t = tr + 256 + lane * 4;               // row (lane&3) of block (lane>>2)
        int t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
        if (lane < 32) {
            short o[4];
            idct_row(t0, t1, t2, t3, o);
            short *r = res + 256 + lane * 4;
            r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = o[3];
        } else {   // second loop of vp8_short_inv_walsh4x4_c (idctllm.c:168-186): row -> 4 block DCs
            int a1 = t0 + t3, b1 = t1 + t2, c1 = t1 - t2, d1 = t0 - t3;
            short *w = wl->wht_dc + (lane - 32) * 4;
            w[0] = (short)((a1 + b1 + 3) >> 3);
            w[1] = (short)((c1 + d1 + 3) >> 3);
            w[2] = (short)((a1 - b1 + 3) >> 3);
            w[3] = (short)((d1 - c1 + 3) >> 3);
        }
    }
    wave_lds_sync();
    // ---- luma blocks
    {
        short o[4];
        int fa = dq[1];
        int i0;
        if (col == 0) i0 = has_y2 ? (int)wl->wht_dc[lane >> 2] : (int)(short)(qY.x * dq[0]);
        else i0 = (short)(qY.x * fa);
        idct_col(i0, (short)(qY.y * fa), (short)(qY.z * fa), (short)(qY.w * fa), o);
        short *t = tr + (lane >> 2) * 16 + col;
        t[0] = o[0]; t[4] = o[1]; t[8] = o[2]; t[12] = o[3];
    }
    wave_lds_sync();
    {
        const short *t = tr + lane * 4;
        short o[4];
        idct_row(t[0], t[1], t[2], t[3], o);
        short *r = res + lane * 4;
        r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = o[3];
    }
    wave_lds_sync();
}

// 16x16 / 8x8 whole-block intra predictors (reconintra.c:139-241, 403-521): value of pixel (y, x)
// from the tile edges.  `n` = 16 or 8, dc precomputed by the caller.
__device__ __forceinline__ int intra_pixel(const unsigned char *tile, int stride, int mode, int y, int x, int dc)
{
    // tile index of (yy, xx) = (yy+1)*stride + xx + 4
    if (mode == VP8IR_DC_PRED) return dc;
    int above = tile[x + 4];
    if (mode == VP8IR_V_PRED) return above;
    int left = tile[(y + 1) * stride + 3];
    if (mode == VP8IR_H_PRED) return left;
    return clamp255(left + above - tile[3]);   // TM_PRED
}

// ---- inter prediction of a 4-pixel row segment (reconinter.c:161-227 + filter.c) --------------
// ref points at pixel (0,0) of the plane; (px,py) = integer position of the first output pixel in
// the current frame; mv in 1/8 pel.  border = 32 (luma) / 16 (chroma); plane w x h (coded size).
__device__ __forceinline__ void inter_row4(const uint8_t *ref, int stride, int px, int py, int mvrow, int mvcol,
                                           bool bilinear, int w, int h, int border, int out[4])
{
    int sx = px + (mvcol >> 3), sy = py + (mvrow >> 3);
    const int fx = mvcol & 7, fy = mvrow & 7;
    // memory safety only (a conforming stream never triggers these): keep every tap inside the
    // allocated plane incl. its border.
    sx = max(-border + 2, min(sx, w + border - 10));
    sy = max(-border + 2, min(sy, h + border - 4));
    const uint8_t *s = ref + (long)sy * stride + sx;
    if ((fx | fy) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = s[i];
        return;
    }
    if (bilinear) {   // filter_block2d_bil (filter.c:376-397): H pass on rows y, y+1; then V
        const int h0 = 128 - fx * 16, h1 = fx * 16, v0 = 128 - fy * 16, v1 = fy * 16;
        int a[5], b[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { a[i] = s[i]; b[i] = s[stride + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int t0 = (a[i] * h0 + a[i + 1] * h1 + 64) >> 7;
            int t1 = (b[i] * h0 + b[i + 1] * h1 + 64) >> 7;
            out[i] = (t0 * v0 + t1 * v1 + 64) >> 7;
        }
        return;
    }
    // six-tap, both passes always (filter.c:41-128): H over rows -2..+3 with clamp, then V with clamp
    int acc[4] = { 64, 64, 64, 64 };
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const uint8_t *row = s + (long)(r - 2) * stride;
        int p[9];
#pragma unroll
        for (int i = 0; i < 9; i++) p[i] = row[i - 2];
        const int vt = k_sixtap[fy][r];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int t = p[i] * k_sixtap[fx][0] + p[i + 1] * k_sixtap[fx][1] + p[i + 2] * k_sixtap[fx][2]
                  + p[i + 3] * k_sixtap[fx][3] + p[i + 4] * k_sixtap[fx][4] + p[i + 5] * k_sixtap[fx][5] + 64;
            acc[i] += clamp255(t >> 7) * vt;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = clamp255(acc[i] >> 7);
}

// clamp_mv_to_umv_border (reconinter.c:348-368)
__device__ __forceinline__ void clamp_luma_mv(int &row, int &col, int e_left, int e_right, int e_top, int e_bottom)
{
    if (col < e_left - (19 << 3)) col = e_left - (16 << 3);
    else if (col > e_right + (18 << 3)) col = e_right + (16 << 3);
    if (row < e_top - (19 << 3)) row = e_top - (16 << 3);
    else if (row > e_bottom + (18 << 3)) row = e_bottom + (16 << 3);
}
// clamp_uvmv_to_umv_border (reconinter.c:371-382)
__device__ __forceinline__ void clamp_chroma_mv(int &row, int &col, int e_left, int e_right, int e_top, int e_bottom)
{
    if (2 * col < e_left - (19 << 3)) col = (e_left - (16 << 3)) >> 1;
    if (2 * col > e_right + (18 << 3)) col = (e_right + (16 << 3)) >> 1;
    if (2 * row < e_top - (19 << 3)) row = (e_top - (16 << 3)) >> 1;
    if (2 * row > e_bottom + (18 << 3)) row = (e_bottom + (16 << 3)) >> 1;
}

extern "C" __global__ void __launch_bounds__(1024)
vp8_recon_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = blockDim.x >> 6;
    const int cols = g.mb_cols, rows = g.mb_rows;

    // ---- LDS carve: [progress flags 256 B][bpred table 160 B][NW x WaveLds][NW x line slot]
    int *prog = (int *)smem;
    unsigned char *bptab = smem + 256;
    WaveLds *wl = (WaveLds *)(smem + 512) + wave;
    const int lbytes = line_bytes(g.aligned_w);
    unsigned char *lines = smem + 512 + NW * sizeof(WaveLds);
    unsigned char *my_line = lines + wave * lbytes;
    // within a slot: Y at +0 (LINE_PAD + W + LINE_PAD), U, V each (LINE_PAD + W/2 + LINE_PAD)
    const int lU = 2 * LINE_PAD + g.aligned_w, lV = lU + 2 * LINE_PAD + g.aligned_w / 2;

    if (threadIdx.x < 64) prog[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 160; i += blockDim.x) bptab[i] = k_bpred_tab[i];   // blockDim may be 128
    if (lane < 3) {   // x = -1 of every line is the constant 129 left border (setupintrarecon.c:23-30)
        const int off = lane == 0 ? 0 : (lane == 1 ? lU : lV);
        my_line[off + LINE_PAD - 1] = 129;
    }
    __syncthreads();

    const int myjobs = (njobs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_rows = myjobs * rows;
    const int dep_wave = (wave + NW - 1) % NW;

    for (int R = wave, k = 0; R < total_rows; R += NW, ++k) {
        const int jj = R / rows, r = R - jj * rows;
        const DevJob &job = jobs[blockIdx.x + jj * gridDim.x];
        const vp8ir_frame_hdr &hdr = job.hdr;
        const bool key = hdr.frame_type == 0;
        const bool bilinear = hdr.version != 0;
        const bool fullpix = hdr.version == 3;

        // Line-slot reuse guard.  Inside one frame the wavefront dependency chain already orders
        // "row R-NW+1 finished reading my previous line" before my first write; the chain is cut at
        // a frame boundary, so check explicitly when row R-NW+1 belongs to another job.
        if (k > 0 && (R - NW + 1) / rows != jj) {
            const int rd = (wave + 1) % NW;
            const int kr = (wave + 1 < NW) ? k - 1 : k;
            wg_wait_ge(&prog[rd], (kr + 1) << 16);
        }
        build_dequant(hdr, wl->dq, lane);
        wave_lds_sync();

        const unsigned char *dep_line = lines + dep_wave * lbytes;
        const int dep_seq = (R - 1) / NW;
        const vp8ir_mb *mbrow = job.mbs + (long)r * cols;
        const int16_t *coefrow = job.coef + (long)r * cols * VP8IR_COEF_PER_MB;
        uint8_t *dY = job.dst + g.y_off + (long)r * 16 * g.y_stride;
        uint8_t *dU = job.dst + g.u_off + (long)r * 8 * g.uv_stride;
        uint8_t *dV = job.dst + g.v_off + (long)r * 8 * g.uv_stride;

        // software pipeline: coefficients of MB c+1 are in flight while MB c is processed
        short4v qY = { 0, 0, 0, 0 }, qC = { 0, 0, 0, 0 };
        {
            const bool sk = mbrow[0].flags & VP8IR_MB_SKIP;
            if (!sk) {
                qY = *(const short4v *)(coefrow + lane * 4);
                if (lane < 36) qC = *(const short4v *)(coefrow + 256 + lane * 4);
            }
        }

        for (int c = 0; c < cols; ++c) {
            const vp8ir_mb &mb = mbrow[c];
            const int y_mode = mb.y_mode, uv_mode = mb.uv_mode, ref_frame = mb.ref_frame;
            const bool skip = mb.flags & VP8IR_MB_SKIP;
            const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
            const short4v cY = qY, cC = qC;
            if (c + 1 < cols) {
                const bool sk = mbrow[c + 1].flags & VP8IR_MB_SKIP;
                if (!sk) {
                    const int16_t *q = coefrow + (long)(c + 1) * VP8IR_COEF_PER_MB;
                    qY = *(const short4v *)(q + lane * 4);
                    if (lane < 36) qC = *(const short4v *)(q + 256 + lane * 4);
                }
            }

            // ---- residual: independent of every neighbour, done BEFORE waiting on the row above
            compute_residual(wl, lane, skip, has_y2, mb.segment_id & 3, cY, cC);

            // ---- wait for the row above to be two MBs ahead (or finished)
            if (r > 0) {
                const int need = min(c + 2, cols);
                wg_wait_ge(&prog[dep_wave], (dep_seq << 16) + need);
            }

            // ---- prediction edges into the tile
            unsigned char *tY = wl->tY, *tU = wl->tU, *tV = wl->tV;
            wave_lds_sync();
            if (lane < 16) {         // left column := previous MB's right column, or the 129 border
                tY[TY_AT(lane, -1)] = c == 0 ? 129 : tY[TY_AT(lane, 15)];
            } else if (lane < 24) {
                int y = lane - 16;
                tU[TC_AT(y, -1)] = c == 0 ? 129 : tU[TC_AT(y, 7)];
            } else if (lane < 32) {
                int y = lane - 24;
                tV[TC_AT(y, -1)] = c == 0 ? 129 : tV[TC_AT(y, 7)];
            }
            if (lane < 21) {         // above row x = -1..19 (row 0: the 127 border incl. top-left)
                int x = lane - 1;
                tY[TY_AT(-1, x)] = r == 0 ? 127 : dep_line[LINE_PAD + c * 16 + x];
            } else if (lane >= 32 && lane < 41) {
                int x = lane - 33;
                tU[TC_AT(-1, x)] = r == 0 ? 127 : dep_line[lU + LINE_PAD + c * 8 + x];
            } else if (lane >= 48 && lane < 57) {
                int x = lane - 49;
                tV[TC_AT(-1, x)] = r == 0 ? 127 : dep_line[lV + LINE_PAD + c * 8 + x];
            }

            wave_lds_sync();
            const short *res = wl->res;
            if (ref_frame == VP8IR_INTRA_FRAME) {
                // ---- chroma (lanes 0..31: plane = lane>>4, block = (lane>>2)&3, row = lane&3)
                {
                    // DC needs sums over the above row and left column of each plane
                    int v = 0;
                    if (lane < 8) v = tU[TC_AT(-1, lane)];
                    else if (lane < 16) v = tU[TC_AT(lane - 8, -1)];
                    else if (lane < 24) v = tV[TC_AT(-1, lane - 16)];
                    else if (lane < 32) v = tV[TC_AT(lane - 24, -1)];
                    // sums per group of 8 lanes
                    int s = v;
                    s += __shfl_xor(s, 1, WAVE); s += __shfl_xor(s, 2, WAVE); s += __shfl_xor(s, 4, WAVE);
                    const int up = r > 0, lf = c > 0;
                    int sUa = __shfl(s, 0, WAVE), sUl = __shfl(s, 8, WAVE), sVa = __shfl(s, 16, WAVE), sVl = __shfl(s, 24, WAVE);
                    int dcU = 128, dcV = 128;
                    if (up | lf) {
                        int shift = 2 + up + lf;
                        dcU = ((up ? sUa : 0) + (lf ? sUl : 0) + (1 << (shift - 1))) >> shift;
                        dcV = ((up ? sVa : 0) + (lf ? sVl : 0) + (1 << (shift - 1))) >> shift;
                    }
                    if (lane < 32) {
                        const int plane = lane >> 4, blk = (lane >> 2) & 3, row = lane & 3;
                        const int y = (blk >> 1) * 4 + row, x0 = (blk & 1) * 4;
                        unsigned char *t = plane ? tV : tU;
                        const short *rr = res + 256 + lane * 4;
                        int dc = plane ? dcV : dcU;
                        unsigned int packed = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            int p = intra_pixel(t, TC_STRIDE, uv_mode, y, x0 + i, dc);
                            packed |= (unsigned)clamp255(p + rr[i]) << (8 * i);
                        }
                        *(unsigned int *)(t + TC_AT(y, x0)) = packed;
                    }
                }
                // ---- luma
                if (y_mode != VP8IR_B_PRED) {
                    int v = 0;
                    if (lane < 16) v = tY[TY_AT(-1, lane)];
                    else if (lane < 32) v = tY[TY_AT(lane - 16, -1)];
                    int s = v;
                    s += __shfl_xor(s, 1, WAVE); s += __shfl_xor(s, 2, WAVE);
                    s += __shfl_xor(s, 4, WAVE); s += __shfl_xor(s, 8, WAVE);
                    const int up = r > 0, lf = c > 0;
                    int sa = __shfl(s, 0, WAVE), sl = __shfl(s, 16, WAVE);
                    int dc = 128;
                    if (up | lf) {
                        int shift = 3 + up + lf;
                        dc = ((up ? sa : 0) + (lf ? sl : 0) + (1 << (shift - 1))) >> shift;
                    }
                    const int blk = lane >> 2, row = lane & 3;
                    const int y = (blk >> 2) * 4 + row, x0 = (blk & 3) * 4;
                    const short *rr = res + lane * 4;
                    unsigned int packed = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        int p = intra_pixel(tY, TY_STRIDE, y_mode, y, x0 + i, dc);
                        packed |= (unsigned)clamp255(p + rr[i]) << (8 * i);
                    }
                    *(unsigned int *)(tY + TY_AT(y, x0)) = packed;
                } else {
                    // B_PRED: 16 sub-blocks in raster order, each predicted from already reconstructed
                    // pixels (decodframe.c:200-236).  16 lanes, one pixel each.
                    const int pr = (lane >> 2) & 3, pc = lane & 3;
                    for (int b = 0; b < 16; ++b) {
                        const int by = b >> 2, bx = b & 3;
                        const int mode = mb.b_modes[b];
                        if (lane < 16) {
                            const int oy = by * 4, ox = bx * 4;
                            int pred;
                            // edge fetch: P[k], k = 0..14 (see table comment).  Above-right of the
                            // right-hand block column is the MB's own above-right row for every block row.
                            auto P = [&](int kk) -> int {
                                if (kk <= 4) { int j = kk == 0 ? 3 : 4 - kk; return tY[TY_AT(oy + j, ox - 1)]; }
                                if (kk == 5) return tY[TY_AT(oy - 1, ox - 1)];
                                int a = kk == 14 ? 7 : kk - 6;
                                if (a >= 4 && bx == 3) return tY[TY_AT(-1, 12 + a)];
                                return tY[TY_AT(oy - 1, ox + a)];
                            };
                            if (mode == VP8IR_B_DC_PRED) {
                                int s = 4;
#pragma unroll
                                for (int i = 0; i < 4; i++) s += P(6 + i) + P(1 + i);
                                pred = s >> 3;
                            } else if (mode == VP8IR_B_TM_PRED) {
                                pred = clamp255(P(6 + pc) - P(5) + P(4 - pr));
                            } else {
                                int e = bptab[mode * 16 + lane], kk = e & 15, kind = e >> 4;
                                if (kind == 2) pred = (P(kk - 1) + 2 * P(kk) + P(kk + 1) + 2) >> 2;
                                else if (kind == 1) pred = (P(kk) + P(kk + 1) + 1) >> 1;
                                else pred = P(kk);
                            }
                            int v = clamp255(pred + res[b * 16 + pr * 4 + pc]);
                            tY[TY_AT(oy + pr, ox + pc)] = (unsigned char)v;
                        }
                        wave_lds_sync();
                    }
                }
            } else {
                // ---- inter MB (vp8_build_inter_predictors_mb, reconinter.c:560-606)
                const vp8ir_mv *mv = job.mvs + ((long)r * cols + c) * 16;
                const uint8_t *rf = job.ref[ref_frame];
                const bool clampmv = mb.flags & VP8IR_MB_CLAMP;
                const int e_left = -((c * 16) << 3), e_right = ((cols - 1 - c) * 16) << 3;
                const int e_top = -((r * 16) << 3), e_bottom = ((rows - 1 - r) * 16) << 3;
                {   // luma: lane = block*4 + row
                    const int blk = lane >> 2, row = lane & 3;
                    const int y = (blk >> 2) * 4 + row, x0 = (blk & 3) * 4;
                    int mrow = mv[blk].row, mcol = mv[blk].col;
                    if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
                    int o[4];
                    inter_row4(rf + g.y_off, g.y_stride, c * 16 + x0, r * 16 + y, mrow, mcol, bilinear,
                               g.aligned_w, g.aligned_h, 32, o);
                    const short *rr = res + lane * 4;
                    unsigned int packed = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) packed |= (unsigned)clamp255(o[i] + rr[i]) << (8 * i);
                    *(unsigned int *)(tY + TY_AT(y, x0)) = packed;
                }
                if (lane < 32) {   // chroma
                    const int plane = lane >> 4, blk = (lane >> 2) & 3, row = lane & 3;
                    const int y = (blk >> 1) * 4 + row, x0 = (blk & 1) * 4;
                    int mrow, mcol;
                    if (y_mode != VP8IR_SPLITMV) {   // reconinter.c:419-424: from the CLAMPED luma MV
                        mrow = mv[0].row; mcol = mv[0].col;
                        if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
                        mrow = (short)(mrow + (1 | (mrow >> 31)));
                        mcol = (short)(mcol + (1 | (mcol >> 31)));
                        mrow /= 2; mcol /= 2;
                        if (fullpix) { mrow &= ~7; mcol &= ~7; }
                    } else {                          // build_4x4uvmvs (reconinter.c:520-558): UNclamped MVs
                        const int kq = (blk >> 1) * 8 + (blk & 1) * 2;
                        mrow = mv[kq].row + mv[kq + 1].row + mv[kq + 4].row + mv[kq + 5].row;
                        mcol = mv[kq].col + mv[kq + 1].col + mv[kq + 4].col + mv[kq + 5].col;
                        mrow += 4 + ((mrow >> 31) << 3);
                        mcol += 4 + ((mcol >> 31) << 3);
                        mrow /= 8; mcol /= 8;
                        if (fullpix) { mrow &= ~7; mcol &= ~7; }
                        if (clampmv) clamp_chroma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
                    }
                    int o[4];
                    inter_row4(rf + (plane ? g.v_off : g.u_off), g.uv_stride, c * 8 + x0, r * 8 + y, mrow, mcol,
                               bilinear, g.aligned_w / 2, g.aligned_h / 2, 16, o);
                    unsigned char *t = plane ? tV : tU;
                    const short *rr = res + 256 + lane * 4;
                    unsigned int packed = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) packed |= (unsigned)clamp255(o[i] + rr[i]) << (8 * i);
                    *(unsigned int *)(t + TC_AT(y, x0)) = packed;
                }
            }

            // ---- write the finished MB: frame (HBM, once) + my line buffer (bottom rows)
            wave_lds_sync();
            {
                const int y = lane >> 2, xd = (lane & 3) * 4;
                unsigned int v = *(const unsigned int *)(tY + TY_AT(y, xd));
                *(unsigned int *)(dY + (long)y * g.y_stride + c * 16 + xd) = v;
                if (y == 15) *(unsigned int *)(my_line + LINE_PAD + c * 16 + xd) = v;
                if (lane < 32) {
                    const int plane = lane >> 4, yy = (lane >> 1) & 7, xx = (lane & 1) * 4;
                    const unsigned char *t = plane ? tV : tU;
                    unsigned int cv = *(const unsigned int *)(t + TC_AT(yy, xx));
                    uint8_t *dp = (plane ? dV : dU) + (long)yy * g.uv_stride + c * 8 + xx;
                    *(unsigned int *)dp = cv;
                    if (yy == 7) *(unsigned int *)(my_line + (plane ? lV : lU) + LINE_PAD + c * 8 + xx) = cv;
                }
                if (c == cols - 1) {
                    // vp8_extend_mb_row (extend.c:160-185): what the next row's last MB sees as
                    // above-right is the last pixel of this line replicated.
                    if (lane == 0) {
                        unsigned int e = tY[TY_AT(15, 15)] * 0x01010101u;
                        *(unsigned int *)(my_line + LINE_PAD + cols * 16) = e;
                    }
                }
            }
            wg_publish(&prog[wave], c + 1 == cols ? (k + 1) << 16 : (k << 16) + c + 1, lane);
        }
        (void)key;
    }
}
