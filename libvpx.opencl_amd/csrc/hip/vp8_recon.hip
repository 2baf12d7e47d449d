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
//     a per-wave LDS line buffer, the left column stays in the wave's LDS working set.  The frame in
//     HBM is written exactly once per pixel and never read back by this kernel; coefficients are
//     read exactly once (coalesced 8-byte-per-lane loads, lane = 4x4 block column).
//   * the per-MB instruction stream is latency-bound (dependent steps), so dependent memory round
//     trips are designed out: the 4x4 transposes of the IDCT/WHT are DPP quad permutes (no LDS), DC
//     sums are v_sad_u8 over uniformly-read LDS dwords (no cross-lane reduction), MB descriptors are
//     prefetched two MBs ahead with one vector load and read with v_readlane, and a B_PRED sub-block
//     step costs exactly one LDS write->read turnaround (edge vector assembled and indexed in
//     registers with v_perm / v_alignbyte, predictor table and residuals preloaded).
//   * integer only (u8 pixels, i16 coefficients, i32 accumulators); no MFMA by design.
#include "vp8_common.hip.h"
#include "vp8_block_prims.hip.h"
#include <stddef.h>

// ---- per-wave LDS working set ----------------------------------------------------------------
//   tY: 17 rows (y = -1..15), 40-byte rows: x = -4..-1 at 12..15, x = 0..15 at 16..31 (16-byte
//       aligned), x = 16..19 at 32..35.  Row -1 holds the above line incl. top-left and above-right.
//   tU/tV: 9 rows, 24-byte rows: x = -4..-1 at 4..7, x = 0..7 at 8..15.
//   lcol: left column of the current MB, contiguous (Y 16, U 8, V 8) for v_sad_u8 sums.
#define TY_STRIDE 40
#define TC_STRIDE 24
#define TY_AT(y, x) (((y) + 1) * TY_STRIDE + 16 + (x))
#define TC_AT(y, x) (((y) + 1) * TC_STRIDE + 8 + (x))
struct __attribute__((aligned(16))) WaveLds {
    unsigned char lcol[32];             //    0
    unsigned char tY[17 * TY_STRIDE];   //   32 .. 712
    unsigned char padA[8];              //  712 .. 720
    unsigned char tU[9 * TC_STRIDE];    //  720 .. 936
    unsigned char tV[9 * TC_STRIDE];    //  936 .. 1152
    short res[384];                     // 1152 .. 1920  residual, pixel order [blk][row][col] (B_PRED only)
    short dq[4][8];                     // 1920 .. 1984  per segment: y1dc,y1ac,y2dc,y2ac,uvdc,uvac
    unsigned int colbuf[16];            // 1984 .. 2048  right column of each finished 4x4 block
    short wht_dc[16];                   // 2048 .. 2080  Y2 -> per-block DC
};
static_assert(sizeof(WaveLds) == 2080, "WaveLds layout");
static_assert(offsetof(WaveLds, tY) % 16 == 0 && offsetof(WaveLds, tU) % 16 == 0 && offsetof(WaveLds, tV) % 8 == 0
              && offsetof(WaveLds, res) % 16 == 0, "WaveLds alignment");

#define LINE_PAD 16   // line[LINE_PAD + x]; x = -4..-1 readable (x = -1 is the 129 left border), x up to W+3 valid

// 4x4 transpose of 16-bit values across the four lanes of a quad.  In: lane j holds column j as
// (o0,o1,o2,o3) = rows 0..3 (truncated to 16 bits here, as the reference's `short output[16]`).
// Out: lane i holds row i as t[0..3] = columns 0..3.  Two DPP stages, no LDS.
__device__ __forceinline__ void quad_transpose16(int o0, int o1, int o2, int o3, int lane, int t[4])
{
    u32 p01 = ((u32)o0 & 0xffff) | ((u32)o1 << 16), p23 = ((u32)o2 & 0xffff) | ((u32)o3 << 16);
    const u32 selA = (lane & 1) ? 0x03020706u : 0x05040100u;
    u32 a01 = perm(dpp_xor1(p01), p01, selA), a23 = perm(dpp_xor1(p23), p23, selA);
    const bool up = lane & 2;
    u32 recv = dpp_xor2(up ? a01 : a23);
    u32 lo = up ? recv : a01, hi = up ? a23 : recv;
    t[0] = sext16(lo); t[1] = hi16(lo); t[2] = sext16(hi); t[3] = hi16(hi);
}

// vp8cx_init_de_quantizer + mb_init_dequantizer (vp8/decoder/decodframe.c:50-109,
// vp8/common/quant_common.c:39-132): six factors per segment.
__device__ __forceinline__ void build_dequant(const vp8ir_frame_hdr &h, short (*dq)[8], int lane)
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

// four int16 coefficients (one 4x4 block column) as loaded: two dwords
typedef unsigned int coef4 __attribute__((ext_vector_type(2)));
typedef GLOBAL_AS const coef4 *g_cs4p;
typedef GLOBAL_AS const unsigned int *g_cmvp;       // vp8ir_mv {int16 row, col} read as one dword
__device__ __forceinline__ int c4x(coef4 v) { return (int)(short)(v.x & 0xffff); }
__device__ __forceinline__ int c4y(coef4 v) { return (int)v.x >> 16; }
__device__ __forceinline__ int c4z(coef4 v) { return (int)(short)(v.y & 0xffff); }
__device__ __forceinline__ int c4w(coef4 v) { return (int)v.y >> 16; }

// ---- inter prediction of a 4-pixel row segment (reconinter.c:161-227 + filter.c) --------------
// ref points at pixel (0,0) of the plane; (px,py) = integer position of the first output pixel in
// the current frame; mv in 1/8 pel.  border = 32 (luma) / 16 (chroma); plane w x h (coded size).
__device__ __forceinline__ u32 inter_row4(g_cu8p ref, int stride, int px, int py, int mvrow, int mvcol,
                                          bool bilinear, int w, int h, int border, int j)
{
    int sx = px + (mvcol >> 3), sy = py + (mvrow >> 3);
    const int fx = mvcol & 7, fy = mvrow & 7;
    // memory safety only (a conforming stream never triggers these): keep every tap inside the
    // allocated plane incl. its border.
    sx = max(-border + 2, min(sx, w + border - 10));
    sy = max(-border + 2, min(sy, h + border - 7));
    g_cu8p s = ref + (long)sy * stride + sx;
    int out[4];
    if ((fx | fy) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = s[i];
    } else if (bilinear) {   // filter_block2d_bil (filter.c:376-397): H pass on rows y, y+1; then V
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
    } else {
        // six-tap, both passes always (filter.c:41-128): H over rows -2..+3 with clamp, then V with clamp.
        // The four lanes of a quad are the four rows of one 4x4 block (j = row) with one MV: together they need the
        // horizontally filtered source rows -2..6 of the block.  Lane j filters rows j-2 and j+2 (lane 0 also row 6)
        // instead of its own six, and the quad exchanges the results (four clamped pixels = one dword) by DPP.
        const SixTaps tx = sixtap_taps(fx), ty = sixtap_taps(fy);
        auto hrow = [&](int r) -> u32 { return sixtap_hrow(s - 2 + (long)r * stride, tx); };
        const u32 Ha = hrow(-2), Hb = hrow(2), Hc = hrow(j == 0 ? 6 : 2);
        // quad rotations: lane i takes the value of lane (i + k) & 3
        auto rot1 = [](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x39, 0xf, 0xf, false); };
        auto rot2 = [](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); };
        auto rot3 = [](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x93, 0xf, 0xf, false); };
        // every rotation is executed by the whole quad (a DPP read of a lane that sits in the other arm of a
        // branch returns nothing), the selection happens afterwards
        const u32 a1 = rot1(Ha), a2 = rot2(Ha), a3 = rot3(Ha);
        const u32 b1 = rot1(Hb), b2 = rot2(Hb), b3 = rot3(Hb), c1 = rot1(Hc);
        u32 H[6];      // filtered source rows j-2 .. j+3 of the block = rows -2 .. +3 of this lane's output row
        H[0] = Ha;
        H[1] = j + 1 < 4 ? a1 : b1;
        H[2] = j + 2 < 4 ? a2 : b2;
        H[3] = j + 3 < 4 ? a3 : b3;
        H[4] = Hb;
        H[5] = j == 3 ? c1 : b1;
        return sixtap_vcol(H, ty);
    }
    return (u32)out[0] | ((u32)out[1] << 8) | ((u32)out[2] << 16) | ((u32)out[3] << 24);
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

// Roles of the 32 lanes that work on one macroblock: luma segments (block hl>>2 and the one eight further, row hl&3 of it),
// one chroma segment (plane cpl, row cy, first column cx0), `col` = the block column a lane transforms.
struct MbLanes { int hl, ly0, lx0, cpl, cy, cx0, col; };
__device__ __forceinline__ MbLanes mb_lanes(int hl)
{
    MbLanes m;
    m.hl = hl;
    m.ly0 = ((hl >> 4) << 2) + (hl & 3); m.lx0 = ((hl >> 2) & 3) << 2;
    m.cpl = hl >> 4; m.cy = (((hl >> 2) & 3) >> 1) * 4 + (hl & 3); m.cx0 = ((hl >> 2) & 1) << 2;
    m.col = hl & 3;
    return m;
}
struct Coefs { coef4 y0, y1, c, y2; };     // luma blocks 0..7, 8..15, chroma, Y2 (hl < 4)

// A lane's coefficients out of the DEVICE FORM of the IR (include/vp8_ir.h: vp8ir_mbx + the slot's block stream): block column
// hl & 3 of the luma blocks hl >> 2 and 8 + (hl >> 2), of chroma block 16 + (hl >> 2), and (hl < 4) of the Y2 block.  A block with
// eob > 1 stands in the stream, behind the macroblock's such blocks before it; one with a lone first coefficient has it in the
// record (y2[k] in a macroblock without a Y2 block, cdc[k]); any other is zeros.  w: the descriptor's eob words (dwords 2..7)
// and sparse_first (dword 14); rec: the record; blocks: the slot's stream.
struct MbWords { u32 e[6]; u32 first; };
__device__ __forceinline__ Coefs load_coefs_dev(g_cu32p rec, g_cs16p blocks, const MbWords &w, bool has_y2, int hl)
{
    u32 ge2 = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) ge2 |= ((((((w.e[q] + 0x7e7e7e7eu) & 0x80808080u) >> 7) * 0x00204081u) >> 21) & 0xfu) << (4 * q);
    const int b = hl >> 2, col = hl & 3;
    const GLOBAL_AS unsigned short *rec16 = (const GLOBAL_AS unsigned short *)rec;
    g_cs16p base = blocks + (long)w.first * 16 + col * 4;
    u32 l0 = 0, l1 = 0, lc = 0;
    if (col == 0) {
        if (!has_y2) { l0 = rec16[32 + b]; l1 = rec16[40 + b]; }
        lc = rec16[48 + b];
    }
    auto blk = [&](int k, u32 lone) -> coef4 {
        coef4 r = { lone, 0 };
        if ((ge2 >> k) & 1) r = *(g_cs4p)(base + __builtin_popcount(ge2 & ((1u << k) - 1)) * 16);
        return r;
    };
    Coefs v;
    v.y0 = blk(b, l0); v.y1 = blk(b + 8, l1); v.c = blk(16 + b, lc);
    v.y2 = (coef4){ 0, 0 };
    if (hl < 4) v.y2 = *(g_cs4p)((g_cs16p)rec + 32 + hl * 4);
    return v;
}
struct Mv2 { u32 a, b; };                  // the MVs of a lane's two luma segments

// The residual of a macroblock (dequantisation, Y2 WHT, inverse DCT: decodframe.c:239-296, idctllm.c), by the 32 lanes
// of its half wave.  rY0/rY1: the 4 residuals of luma segment (block hl>>2 [+8], row hl&3); rC: chroma likewise.
__device__ __forceinline__ void mb_residual(WaveLds *wl, const Coefs &q, bool skip, bool has_y2, int seg, const MbLanes &R, int lane,
                                            int rY0[4], int rY1[4], int rC[4])
{
    const int hl = R.hl, col = R.col;
    // ---- residual (independent of every neighbour: done BEFORE waiting on the row above).
    // rY0/rY1: the 4 residuals of luma segment (block hl>>2 [+8], row hl&3); rC: chroma likewise.
    #pragma unroll
    for (int i = 0; i < 4; i++) rY0[i] = rY1[i] = rC[i] = 0;
    if (!skip) {
        const short *dq = wl->dq[seg];
        const int dq_y1dc = dq[0], dq_y1ac = dq[1], dq_y2dc = dq[2], dq_y2ac = dq[3], dq_uvdc = dq[4], dq_uvac = dq[5];
        {   // Y2: vp8_dequantize_b + vp8_short_inv_walsh4x4_c (idctllm.c:140-192); lanes hl < 4
            int t[4];
            const int f0 = col == 0 ? dq_y2dc : dq_y2ac;
            const int i0 = (short)(c4x(q.y2) * f0), i1 = (short)(c4y(q.y2) * dq_y2ac);
            const int i2 = (short)(c4z(q.y2) * dq_y2ac), i3 = (short)(c4w(q.y2) * dq_y2ac);
            const int a1 = i0 + i3, b1 = i1 + i2, c1 = i1 - i2, d1 = i0 - i3;
            quad_transpose16(a1 + b1, c1 + d1, a1 - b1, d1 - c1, lane, t);
            if (has_y2 && hl < 4) {
                const int a2 = t[0] + t[3], b2 = t[1] + t[2], c2 = t[1] - t[2], d2 = t[0] - t[3];
                short *w = wl->wht_dc + hl * 4;
                w[0] = (short)((a2 + b2 + 3) >> 3);
                w[1] = (short)((c2 + d2 + 3) >> 3);
                w[2] = (short)((a2 - b2 + 3) >> 3);
                w[3] = (short)((d2 - c2 + 3) >> 3);
            }
        }
        wave_lds_sync();
        // chroma, luma blocks 0..7 and luma blocks 8..15: three independent transforms, written pass by pass so
        // that their dependent instruction chains interleave (one wave alone pays every instruction's latency)
        int oC[4], o0[4], o1[4], tC[4], t0[4], t1[4];
        int iA, iB;
        if (col == 0) {
            iA = has_y2 ? (int)wl->wht_dc[hl >> 2] : (int)(short)(c4x(q.y0) * dq_y1dc);
            iB = has_y2 ? (int)wl->wht_dc[8 + (hl >> 2)] : (int)(short)(c4x(q.y1) * dq_y1dc);
        } else {
            iA = (short)(c4x(q.y0) * dq_y1ac);
            iB = (short)(c4x(q.y1) * dq_y1ac);
        }
        idct_col((short)(c4x(q.c) * (col == 0 ? dq_uvdc : dq_uvac)), (short)(c4y(q.c) * dq_uvac), (short)(c4z(q.c) * dq_uvac),
                 (short)(c4w(q.c) * dq_uvac), oC);
        idct_col(iA, (short)(c4y(q.y0) * dq_y1ac), (short)(c4z(q.y0) * dq_y1ac), (short)(c4w(q.y0) * dq_y1ac), o0);
        idct_col(iB, (short)(c4y(q.y1) * dq_y1ac), (short)(c4z(q.y1) * dq_y1ac), (short)(c4w(q.y1) * dq_y1ac), o1);
        quad_transpose16(oC[0], oC[1], oC[2], oC[3], lane, tC);
        quad_transpose16(o0[0], o0[1], o0[2], o0[3], lane, t0);
        quad_transpose16(o1[0], o1[1], o1[2], o1[3], lane, t1);
        idct_row(tC, rC);
        idct_row(t0, rY0);
        idct_row(t1, rY1);
    }
}

// The prediction of an inter macroblock plus its residual (vp8_build_inter_predictors_mb, reconinter.c:560-606), by the 32
// lanes of its half wave: mv = the macroblock's 16 MVs, mv2 = those of this lane's segments, rf = the reference frame
// buffer.  Uses wl->res as scratch (the first six-tap pass of a macroblock with one MV is shared through it).
__device__ __forceinline__ void mb_inter(const DevGeom &g, WaveLds *wl, g_cu8p rf, g_cmvp mv, const Mv2 &mv2, u32 flags, int y_mode,
                                         int r, int c, bool bilinear, bool fullpix, const MbLanes &R,
                                         const int rY0[4], const int rY1[4], const int rC[4], u32 &outY0, u32 &outY1, u32 &outC)
{
    const int hl = R.hl, ly0 = R.ly0, lx0 = R.lx0, cpl = R.cpl, cy = R.cy, cx0 = R.cx0;
    const int cols = g.mb_cols, rows = g.mb_rows;
    // ---- inter MB (vp8_build_inter_predictors_mb, reconinter.c:560-606)
        const bool clampmv = flags & VP8IR_MB_CLAMP;
    const int e_left = -((c * 16) << 3), e_right = ((cols - 1 - c) * 16) << 3;
    const int e_top = -((r * 16) << 3), e_bottom = ((rows - 1 - r) * 16) << 3;
    if (y_mode != VP8IR_SPLITMV && !bilinear) {
        // One MV for the whole macroblock (vp8_build_inter16x16_predictors_mb, reconinter.c:384-441):
        // the first six-tap pass is shared through LDS -- 21 source rows x 4 segments for luma, 13 x 2
        // for each chroma plane, 136 row segments for 32 lanes instead of nine per lane.
        u32 *hb = (u32 *)wl->res;                 // B_PRED's residual buffer is idle in an inter MB
        int mrow = sext16(mv2.a), mcol = hi16(mv2.a);
        if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
        // chroma MV from the CLAMPED luma MV (reconinter.c:419-424); version 0: no full-pixel mask
        int crow = (short)(mrow + (1 | (mrow >> 31))), ccol = (short)(mcol + (1 | (mcol >> 31)));
        crow /= 2; ccol /= 2;
        const bool fracY = ((mrow | mcol) & 7) != 0, fracC = ((crow | ccol) & 7) != 0;
        // memory safety only, as in inter_row4: every tap stays inside the plane and its border
        const int sx = max(-32 + 2, min(c * 16 + (mcol >> 3), g.aligned_w + 32 - 22));
        const int sy = max(-32 + 2, min(r * 16 + (mrow >> 3), g.aligned_h + 32 - 19));
        const int sxc = max(-16 + 2, min(c * 8 + (ccol >> 3), g.aligned_w / 2 + 16 - 14));
        const int syc = max(-16 + 2, min(r * 8 + (crow >> 3), g.aligned_h / 2 + 16 - 11));
        if (fracY) {
            const SixTaps tx = sixtap_taps(mcol & 7);
            g_cu8p base = rf + g.y_off + (long)(sy - 2) * g.y_stride + (sx - 2);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int t = hl + 32 * i;            // source row t>>2 (0 = two above), segment t&3
                if (t < 84) hb[t] = sixtap_hrow(base + (long)(t >> 2) * g.y_stride + (t & 3) * 4, tx);
            }
        }
        if (fracC) {
            const SixTaps tx = sixtap_taps(ccol & 7);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int t = hl + 32 * i, pl = t >= 26, rem = t - 26 * pl;
                if (t < 52) {
                    g_cu8p base = rf + (pl ? g.v_off : g.u_off) + (long)(syc - 2 + (rem >> 1)) * g.uv_stride + (sxc - 2);
                    hb[84 + t] = sixtap_hrow(base + (rem & 1) * 4, tx);
                }
            }
        }
        wave_lds_sync();
        u32 ppY0, ppY1, ppC;
        if (fracY) {
            const SixTaps ty = sixtap_taps(mrow & 7);
            u32 H[6];
#pragma unroll
            for (int k = 0; k < 6; k++) H[k] = hb[(ly0 + k) * 4 + (lx0 >> 2)];
            ppY0 = sixtap_vcol(H, ty);
#pragma unroll
            for (int k = 0; k < 6; k++) H[k] = hb[(ly0 + 8 + k) * 4 + (lx0 >> 2)];
            ppY1 = sixtap_vcol(H, ty);
        } else {                                      // vp8_copy_mem16x16 (reconinter.c:22-63)
            g_cu8p s0 = rf + g.y_off + (long)(sy + ly0) * g.y_stride + sx + lx0;
            g_cu8p s1 = s0 + 8 * (long)g.y_stride;
            ppY0 = (u32)s0[0] | ((u32)s0[1] << 8) | ((u32)s0[2] << 16) | ((u32)s0[3] << 24);
            ppY1 = (u32)s1[0] | ((u32)s1[1] << 8) | ((u32)s1[2] << 16) | ((u32)s1[3] << 24);
        }
        if (fracC) {
            const SixTaps ty = sixtap_taps(crow & 7);
            u32 H[6];
#pragma unroll
            for (int k = 0; k < 6; k++) H[k] = hb[84 + cpl * 26 + (cy + k) * 2 + (cx0 >> 2)];
            ppC = sixtap_vcol(H, ty);
        } else {
            g_cu8p s0 = rf + (cpl ? g.v_off : g.u_off) + (long)(syc + cy) * g.uv_stride + sxc + cx0;
            ppC = (u32)s0[0] | ((u32)s0[1] << 8) | ((u32)s0[2] << 16) | ((u32)s0[3] << 24);
        }
        outY0 = add_clamp_pack(ppY0, rY0);
        outY1 = add_clamp_pack(ppY1, rY1);
        outC = add_clamp_pack(ppC, rC);
    } else {
#pragma unroll
    for (int p = 0; p < 2; p++) {   // luma: segment of block p*8 + hl>>2, row hl&3
        const int y = ly0 + 8 * p;
        const u32 mvw = p ? mv2.b : mv2.a;
        int mrow = sext16(mvw), mcol = hi16(mvw);
        if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
        const u32 pp = inter_row4(rf + g.y_off, g.y_stride, c * 16 + lx0, r * 16 + y, mrow, mcol, bilinear,
                                  g.aligned_w, g.aligned_h, 32, hl & 3);
        if (p) outY1 = add_clamp_pack(pp, rY1); else outY0 = add_clamp_pack(pp, rY0);
    }
    {   // chroma
        const int blk = (hl >> 2) & 3;
        int mrow, mcol;
        if (y_mode != VP8IR_SPLITMV) {   // reconinter.c:419-424: from the CLAMPED luma MV
            const u32 mvw = mv2.a;
            mrow = sext16(mvw); mcol = hi16(mvw);
            if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
            mrow = (short)(mrow + (1 | (mrow >> 31)));
            mcol = (short)(mcol + (1 | (mcol >> 31)));
            mrow /= 2; mcol /= 2;
            if (fullpix) { mrow &= ~7; mcol &= ~7; }
        } else {                          // build_4x4uvmvs (reconinter.c:520-558): UNclamped MVs
            const int kq = (blk >> 1) * 8 + (blk & 1) * 2;
            const u32 m0 = mv[kq], m1 = mv[kq + 1], m4 = mv[kq + 4], m5 = mv[kq + 5];
            mrow = sext16(m0) + sext16(m1) + sext16(m4) + sext16(m5);
            mcol = hi16(m0) + hi16(m1) + hi16(m4) + hi16(m5);
            mrow += 4 + ((mrow >> 31) << 3);
            mcol += 4 + ((mcol >> 31) << 3);
            mrow /= 8; mcol /= 8;
            if (fullpix) { mrow &= ~7; mcol &= ~7; }
            if (clampmv) clamp_chroma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
        }
        const u32 pp = inter_row4(rf + (cpl ? g.v_off : g.u_off), g.uv_stride, c * 8 + cx0, r * 8 + cy, mrow,
                                  mcol, bilinear, g.aligned_w / 2, g.aligned_h / 2, 16, hl & 3);
        outC = add_clamp_pack(pp, rC);
    }
    }
}

// Two frames per wave: lanes 0..31 reconstruct frame A, lanes 32..63 frame B of a job pair, at the
// same MB position.  The kernel is bound by VALU issue (one wave instruction costs 4 SIMD cycles
// whatever the number of active lanes) and the B_PRED chain keeps only 16 lanes busy per frame, so
// sharing the instruction stream between two independent frames nearly halves the cost of the
// B_PRED-heavy key frames (and of chroma prediction); the 64-lane-wide steps take two passes of 32
// lanes and cost the same per MB as before.
//
// Lane roles inside a half (hl = lane & 31):
//   residual   : hl = block*4 + column; four passes: chroma (8 blocks), Y2 (hl < 4), luma blocks 0..7, 8..15
//   prediction : hl = block*4 + row -> a 4-pixel row segment; luma in two passes, chroma in one
//   B_PRED     : hl 0..15 = the 16 pixels of the current 4x4 sub-block
//
// XCU = true: the rows of ONE frame pair are spread over the waves of S workgroups on different CUs (small launches:
// a frame pair no longer has to live on one CU, vp8hip.hip picks S).  Nothing is shared through LDS then; the
// unfiltered bottom pixel line of a row travels to the wave of the row below as 8-byte granules {4 pixels, tag} in
// a per-row buffer in global memory, written and polled with agent-scope (sc1) accesses: the tag (the launch's
// epoch) doubles as the progress flag, so there is no separate flag, no store-acknowledge wait and no fence.
// A bounded poll turns a broken hand-over into an error status instead of a hang.

//
// INTER_DONE = true: the inter macroblocks of the launch have been reconstructed already by vp8_inter_mb_kernel (they need
// nothing from their neighbours); this pass does the intra macroblocks, in dependency order as ever, and for an inter
// macroblock only reads back the pixels it would have produced, to keep the line below it and the column right of it
// supplied.  Frames without a single intra macroblock (intra_flags[job] == 0) are skipped altogether.
template <bool XCU, bool INTER_DONE>
__device__ __forceinline__ void recon_body(const DevJob *__restrict__ jobs, int njobs, DevGeom g, u64 *gran_base,
                                           u32 epoch, int S, int *err, const unsigned int *__restrict__ intra_flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = blockDim.x >> 6;
    const int cols = g.mb_cols, rows = g.mb_rows;
    const int half = lane >> 5, hl = lane & 31;
    // XCU: workgroups with equal blockIdx.x % 8 sit on one XCD under round-robin placement (speed only, never
    // correctness); S consecutive ones of them form the group of one frame pair
    const int xq = blockIdx.x >> 3;
    const int group = XCU ? (xq / S) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
    const int gw = XCU ? (xq % S) * NW + wave : wave;      // this wave among the TW waves sharing the pair(s)
    const int TW = XCU ? S * NW : NW;
    const int GS = cols * 8 + 2;                           // granules per row: Y cols*4 + 1, U cols*2, V cols*2

    // ---- LDS carve: [progress flags 256 B][B_PRED gather table 640 B, pad to 1024]
    //                 [2*NW x WaveLds][2*NW x line slot]      (one working set and one line per half)
    int *prog = (int *)smem;
    u32 *gtab = (u32 *)(smem + 256);
    WaveLds *wl = (WaveLds *)(smem + 1024) + wave * 2 + half;
    const int lbytes = 2 * g.aligned_w + 6 * LINE_PAD;
    unsigned char *lines = smem + 1024 + 2 * NW * sizeof(WaveLds);
    unsigned char *my_line = lines + (wave * 2 + half) * lbytes;
    // within a slot: Y at +0 (LINE_PAD + W + LINE_PAD), U, V each (LINE_PAD + W/2 + LINE_PAD)
    const int lU = 2 * LINE_PAD + g.aligned_w, lV = lU + 2 * LINE_PAD + g.aligned_w / 2;

    if (threadIdx.x < 64) prog[threadIdx.x] = 0;
    if (!XCU && hl < 3) {   // x = -4..-1 of every line: only x = -1 is ever used, the constant 129 left border
        const int off = hl == 0 ? 0 : (hl == 1 ? lU : lV);
        *(u32 *)(my_line + off + LINE_PAD - 4) = 0x81818181u;
    }
    // B_PRED gather table gtab[mode][pixel] (built once per workgroup): one dword per entry,
    //   byte0..2 = tile byte offsets (relative to the block's top-left pixel, biased by +64) of the three
    //   edge pixels p0,p1,p2 the predictor reads, byte3 = kind:
    //   0: p1   1: (p1+p2+1)>>1   2: (p0+2*p1+p2+2)>>2   3: clamp(p0+p1-p2) (B_TM_PRED)
    // derived from k_bpred_tab with P[k] -> tile offset: L_j = j*stride-1, TL = -stride-1, A_i = -stride+i.
    for (int t = threadIdx.x; t < 160; t += blockDim.x) {
        const int m = t >> 4, i = t & 15, pr = i >> 2, pc = i & 3;
        auto poff = [](int k) -> int {      // tile offset of edge-vector element P[k], k = 0..14
            if (k <= 4) return (k == 0 ? 3 : 4 - k) * TY_STRIDE - 1;
            if (k == 5) return -TY_STRIDE - 1;
            return -TY_STRIDE + (k == 14 ? 7 : k - 6);
        };
        int o0, o1, o2, kind;
        if (m == VP8IR_B_TM_PRED) { o0 = poff(6 + pc); o1 = poff(4 - pr); o2 = poff(5); kind = 3; }
        else {
            const int e = k_bpred_tab[t], kk = e & 15;
            kind = e >> 4;
            o0 = poff(kk > 0 ? kk - 1 : 0); o1 = poff(kk); o2 = poff(kk < 14 ? kk + 1 : 14);
        }
        gtab[t] = (u32)(o0 + 64) | ((u32)(o1 + 64) << 8) | ((u32)(o2 + 64) << 16) | ((u32)kind << 24);
    }
    __syncthreads();

    // prediction-stage roles: segment row / first column for luma pass 0 (blocks 0..7), pass 1 adds 8 rows;
    // chroma: plane = hl>>4, block = (hl>>2)&3
    const int ly0 = ((hl >> 4) << 2) + (hl & 3), lx0 = ((hl >> 2) & 3) << 2;
    const int cpl = hl >> 4, cy = (((hl >> 2) & 3) >> 1) * 4 + (hl & 3), cx0 = ((hl >> 2) & 1) << 2;
    const int col = hl & 3;
    const MbLanes ML = mb_lanes(hl);

    const int npairs = (njobs + 1) >> 1;
    // XCU: one pair per group and launch (the per-row granule buffers are not reused inside a launch)
    const int mypairs = XCU ? (group < npairs ? 1 : 0) : (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_rows = mypairs * rows;
    const int dep_wave = (wave + NW - 1) % NW;

    for (int R = gw, k = 0; R < total_rows; R += TW, ++k) {
        const int jj = R / rows, r = R - jj * rows;
        const int pair = XCU ? group : (int)blockIdx.x + jj * (int)gridDim.x;
        const bool haveB = 2 * pair + 1 < njobs;
        if (INTER_DONE && !intra_flags[2 * pair] && !(haveB && intra_flags[2 * pair + 1])) {
            // nothing to do for this pair of frames; the row still counts in the hand-over protocol of the workgroup
            if (!XCU) wg_publish_lds(&prog[wave], (k + 1) << 16, lane);
            continue;
        }
        const DevJob &jobA = jobs[2 * pair];
        const DevJob &jobB = jobs[haveB ? 2 * pair + 1 : 2 * pair];
        const bool valid = half == 0 || haveB;       // an odd job count leaves the last B half idle
        const vp8ir_frame_hdr &hA = jobA.hdr, &hB = jobB.hdr;
        const int version = half ? hB.version : hA.version;
        const bool bilinear = version != 0, fullpix = version == 3;

        // Line-slot reuse guard.  Inside one frame the wavefront dependency chain already orders
        // "row R-NW+1 finished reading my previous line" before my first write; the chain is cut at
        // a frame boundary, so check explicitly when row R-NW+1 belongs to another job pair.
        if (!XCU && k > 0 && (R - NW + 1) / rows != jj) {
            const int rd = (wave + 1) % NW;
            const int kr = (wave + 1 < NW) ? k - 1 : k;
            wg_wait_ge(&prog[rd], (kr + 1) << 16);
        }
        build_dequant(half ? hB : hA, wl->dq, hl);
        if (hl < 8) ((u32 *)wl->lcol)[hl] = 0x81818181u;        // column 0: the left border is 129
        if (hl >= 16) wl->tY[TY_AT(hl - 16, -1)] = 129;
        wave_lds_sync();

        const unsigned char *dep_line = lines + (dep_wave * 2 + half) * lbytes;
        const int dep_seq = (R - 1) / NW;
        // XCU: granule rows of this frame: mine (row r) and the one above
        g_u64p gran_mine = (g_u64p)(gran_base + ((long)(pair * 2 + half) * rows + r) * GS);
        g_u64p gran_above = gran_mine - GS;
        const vp8ir_mbx *mbs = half ? jobB.mbx : jobA.mbx;
        g_cs16p blocks = (g_cs16p)(half ? jobB.blocks : jobA.blocks);
        const vp8ir_mv *mvs = half ? jobB.mvs : jobA.mvs;
        uint8_t *dst = half ? jobB.dst : jobA.dst;
        g_u8p tile_out = (g_u8p)(half ? jobB.tile : jobA.tile);
        g_cu32p mbrow = (g_cu32p)(mbs + (long)r * cols);          // VP8IR_MBX_WORDS dwords per MB, the descriptor in the first 16
        g_u8p dY = (g_u8p)(dst + g.y_off + (long)r * 16 * g.y_stride);
        g_u8p dU = (g_u8p)(dst + g.u_off + (long)r * 8 * g.uv_stride);
        g_u8p dV = (g_u8p)(dst + g.v_off + (long)r * 8 * g.uv_stride);

        auto load_desc = [&](int c) -> u32 { return hl < 16 ? mbrow[c * VP8IR_MBX_WORDS + hl] : 0u; };
        // the two luma MVs of this lane's segments (blocks hl>>2 and 8 + hl>>2), fetched one macroblock ahead so that
        // the reference fetch does not wait for them (inter frames only; a non-split MB has its MV in all 16 entries)
        g_cu32p mvrow = (g_cu32p)(mvs + (long)r * cols * 16);
        const bool inter_frame = (half ? hB.frame_type : hA.frame_type) != 0;
        auto load_mv = [&](int c) -> Mv2 {
            Mv2 m = { 0u, 0u };
            if (inter_frame && !INTER_DONE) { m.a = mvrow[c * 16 + (hl >> 2)]; m.b = mvrow[c * 16 + 8 + (hl >> 2)]; }
            return m;
        };
        auto half_sel = [&](u32 d, int idx) -> u32 {     // dword `idx` of this half's MB descriptor
            const u32 a = (u32)__builtin_amdgcn_readlane((int)d, idx), b = (u32)__builtin_amdgcn_readlane((int)d, 32 + idx);
            return half ? b : a;
        };
        // the coefficients of macroblock c, whose descriptor is d (a skipped macroblock has none)
        auto load_coefs = [&](int c, u32 d, bool skipped) __attribute__((always_inline)) -> Coefs {
            Coefs v;
            v.y0 = v.y1 = v.c = v.y2 = (coef4){ 0, 0 };
            MbWords w;
#pragma unroll
            for (int q = 0; q < 6; q++) w.e[q] = half_sel(d, 2 + q);
            w.first = half_sel(d, 14);
            const int ym = half_sel(d, 0) & 0xff;
            if (!skipped) v = load_coefs_dev(mbrow + (long)c * VP8IR_MBX_WORDS, blocks, w, ym != VP8IR_B_PRED && ym != VP8IR_SPLITMV, hl);
            return v;
        };

        // ------------------------------------------------------------------------------------------
        // one macroblock (of each frame of the pair)
        // ------------------------------------------------------------------------------------------
        // XCU: lane hl < 12 copies one dword of the above line per MB (Y x = -4..19, U and V x = -4..7); its granule is
        // requested one macroblock ahead, so that the round trip to the other CU's data hides behind this MB's work
        const int apl = hl < 6 ? 0 : (hl < 9 ? 1 : 2), ai = hl - (apl == 0 ? 0 : (apl == 1 ? 6 : 9));
        auto above_gran = [&](int c) -> g_u64p {
            const int x = (apl == 0 ? c * 16 : c * 8) - 4 + ai * 4;
            return gran_above + (apl == 0 ? 0 : (apl == 1 ? cols * 4 + 1 : cols * 6 + 1)) + (x >> 2);
        };
        u64 gpre = 0;
        // (INTER_DONE) the pixels of macroblock c this lane would have stored, where vp8_inter_mb_kernel has left them: requested a
        // macroblock AHEAD (round 6) -- a row of a P frame is mostly such macroblocks, and a load, a granule store and an LDS
        // turnaround in a chain per macroblock made this kernel half a millisecond for a 1080p frame with sixteen intra macroblocks
        struct Done { u32 y0, y1, c; };
        auto load_done = [&](int c, u32 d) -> Done {
            Done v = { 0u, 0u, 0u };
            if (INTER_DONE && ((half_sel(d, 0) >> 16) & 0xff) != VP8IR_INTRA_FRAME) {
                if (tile_out) {
                    g_cu8p t = (g_cu8p)tile_out + ((long)r * cols + c) * VP8_TILE_BYTES;
                    v.y0 = *(g_cu32p)(t + ly0 * 16 + lx0);
                    v.y1 = *(g_cu32p)(t + (ly0 + 8) * 16 + lx0);
                    v.c = *(g_cu32p)(t + 256 + cpl * 64 + cy * 8 + cx0);
                } else {
                    v.y0 = *(g_cu32p)(dY + (long)ly0 * g.y_stride + c * 16 + lx0);
                    v.y1 = *(g_cu32p)(dY + (long)(ly0 + 8) * g.y_stride + c * 16 + lx0);
                    v.c = *(g_cu32p)((cpl ? dV : dU) + (long)cy * g.uv_stride + c * 8 + cx0);
                }
            }
            return v;
        };
        auto process = [&](const int c, const u32 mbw, const Coefs &q, const Mv2 &mv2, const Done &done) __attribute__((always_inline)) {
            unsigned char *const tY = wl->tY, *const tU = wl->tU, *const tV = wl->tV;   // lambda locals: selects between them stay in registers
            const u64 gcur = gpre;
            if (XCU && r > 0 && hl < 12 && c + 1 < cols) gpre = gran_load(above_gran(c + 1));
            const u32 w0 = half_sel(mbw, 0), w1 = half_sel(mbw, 1);
            const int y_mode = w0 & 0xff, uv_mode = (w0 >> 8) & 0xff, ref_frame = (w0 >> 16) & 0xff;
            const u32 flags = w0 >> 24;
            const bool skip = flags & VP8IR_MB_SKIP;
            const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
            const int seg = w1 & 3;

            // ---- residual (independent of every neighbour: done BEFORE waiting on the row above)
            int rY0[4], rY1[4], rC[4];
            const bool elsewhere = INTER_DONE && ref_frame != VP8IR_INTRA_FRAME;
            mb_residual(wl, q, skip || elsewhere, has_y2, seg, ML, lane, rY0, rY1, rC);

            // ---- wait for the row above to be two MBs ahead (or finished)
            if (!XCU && r > 0) wg_wait_ge(&prog[dep_wave], (dep_seq << 16) + min(c + 2, cols));

            // ---- above line -> tile row -1 (dword copies: Y x=-4..19, U/V x=-4..7).  A macroblock that is finished already needs
            // neither the line nor -- spread over several CUs -- the wait for it (an intra macroblock further on waits for its own)
            if (hl < 12 && !(XCU && elsewhere)) {
                const int pl = hl < 6 ? 0 : (hl < 9 ? 1 : 2), i = hl - (pl == 0 ? 0 : (pl == 1 ? 6 : 9));
                u32 v = 0x7f7f7f7fu;                 // frame row 0: everything above is 127
                if (r > 0) {
                    if (XCU) {
                        // the granule of x = 16..19 is written when the row above has finished MB c+1 (or its last
                        // MB): polling it IS the "two MBs ahead" rule
                        const int x = (pl == 0 ? c * 16 : c * 8) - 4 + i * 4;
                        v = 0x81818181u;             // x < 0: the constant 129 left border
                        if (x >= 0) v = gran_wait(above_gran(c), gcur, epoch, err, 1);
                    } else {
                        const unsigned char *src = dep_line + (pl == 0 ? 0 : (pl == 1 ? lU : lV)) + LINE_PAD
                                                 + (pl == 0 ? c * 16 : c * 8) - 4 + i * 4;
                        v = *(const u32 *)src;
                    }
                }
                unsigned char *dstp = (pl == 0 ? tY + TY_AT(-1, -4) : (pl == 1 ? tU : tV) + TC_AT(-1, -4)) + i * 4;
                *(u32 *)dstp = v;
            }
            wave_lds_sync();

            u32 outY0 = 0, outY1 = 0, outC = 0;
            if (ref_frame == VP8IR_INTRA_FRAME) {
                const int up = r > 0, lf = c > 0;
                {   // ---- chroma: DC sums by v_sad_u8 over uniformly read dwords
                    const uint2 aU = *(const uint2 *)(tU + TC_AT(-1, 0)), aV = *(const uint2 *)(tV + TC_AT(-1, 0));
                    const uint2 lU2 = *(const uint2 *)(wl->lcol + 16), lV2 = *(const uint2 *)(wl->lcol + 24);
                    int dcU = 128, dcV = 128;
                    if (up | lf) {
                        const int shift = 2 + up + lf;
                        const int sU = (up ? sad4(aU.x) + sad4(aU.y) : 0) + (lf ? sad4(lU2.x) + sad4(lU2.y) : 0);
                        const int sV = (up ? sad4(aV.x) + sad4(aV.y) : 0) + (lf ? sad4(lV2.x) + sad4(lV2.y) : 0);
                        dcU = (sU + (1 << (shift - 1))) >> shift;
                        dcV = (sV + (1 << (shift - 1))) >> shift;
                    }
                    const unsigned char *t = cpl ? tV : tU;
                    const u32 above = *(const u32 *)(t + TC_AT(-1, cx0));
                    const int left = wl->lcol[16 + cpl * 8 + cy];
                    const int tl = t[TC_AT(-1, -1)];
                    outC = add_clamp_pack(intra_pred4(uv_mode, above, left, tl, cpl ? dcV : dcU), rC);
                }
                if (y_mode != VP8IR_B_PRED) {
                    const uint4 aY = *(const uint4 *)(tY + TY_AT(-1, 0));
                    const uint4 lY = *(const uint4 *)(wl->lcol);
                    int dc = 128;
                    if (up | lf) {
                        const int shift = 3 + up + lf;
                        const int s = (up ? sad4(aY.x) + sad4(aY.y) + sad4(aY.z) + sad4(aY.w) : 0)
                                    + (lf ? sad4(lY.x) + sad4(lY.y) + sad4(lY.z) + sad4(lY.w) : 0);
                        dc = (s + (1 << (shift - 1))) >> shift;
                    }
                    const u32 above = *(const u32 *)(tY + TY_AT(-1, lx0));
                    const int tl = tY[TY_AT(-1, -1)];
                    outY0 = add_clamp_pack(intra_pred4(y_mode, above, wl->lcol[ly0], tl, dc), rY0);
                    outY1 = add_clamp_pack(intra_pred4(y_mode, above, wl->lcol[ly0 + 8], tl, dc), rY1);
                } else {
                    // ---- B_PRED (decodframe.c:200-236): 16 sub-blocks in raster order, each predicted
                    // from already reconstructed pixels; lanes hl 0..15 = the block's 16 pixels.  Every
                    // predictor pixel is a function of at most three edge pixels, gathered straight from
                    // the tile with per-lane byte offsets (one LDS turnaround per sub-block).
                    *(uint2 *)(wl->res + hl * 4) = make_uint2(((u32)rY0[0] & 0xffff) | ((u32)rY0[1] << 16),
                                                              ((u32)rY0[2] & 0xffff) | ((u32)rY0[3] << 16));
                    *(uint2 *)(wl->res + 128 + hl * 4) = make_uint2(((u32)rY1[0] & 0xffff) | ((u32)rY1[1] << 16),
                                                                    ((u32)rY1[2] & 0xffff) | ((u32)rY1[3] << 16));
                    // the reference's "down copy" (reconintra4x4.c:305-317): the MB's above-right pixels also
                    // serve as above-right of the right-hand block column of block rows 1..3
                    if (hl < 3) *(u32 *)(tY + TY_AT(4 * hl + 3, 16)) = *(const u32 *)(tY + TY_AT(-1, 16));
                    const u32 bm0 = half_sel(mbw, 10), bm1 = half_sel(mbw, 11), bm2 = half_sel(mbw, 12), bm3 = half_sel(mbw, 13);
                    wave_lds_sync();
                    const int pr = (hl >> 2) & 3, pc = hl & 3;
                    // Block (bx, by) needs its left, above and above-right neighbours: all blocks with the same
                    // bx + 2*by are independent.  Ten diagonals instead of sixteen blocks in a row: lanes 0..15 take
                    // the diagonal's lower block, lanes 16..31 the upper one (bx + 2, by - 1), where there is one.
                    const int grp = (hl >> 4) & 1;
                    // residual and gather entry of a block: independent of the prediction chain, fetched one diagonal ahead
                    auto block_of = [&](int d, int &by, int &bx, bool &active) {
                        const int byA = d >> 1 < 3 ? d >> 1 : 3, bxA = d - 2 * byA;
                        const bool hasB = byA >= 1 && bxA + 2 <= 3;
                        by = (grp && hasB) ? byA - 1 : byA; bx = (grp && hasB) ? bxA + 2 : bxA;
                        active = grp == 0 || hasB;
                    };
                    auto mode_of = [&](int by, int bx) -> int {
                        const u32 bmw = by == 0 ? bm0 : (by == 1 ? bm1 : (by == 2 ? bm2 : bm3));
                        return (bmw >> (8 * bx)) & 0xff;
                    };
                    int res_nx = wl->res[hl & 15];
                    u32 ent_nx = gtab[mode_of(0, 0) * 16 + (hl & 15)];
#pragma unroll
                    for (int d = 0; d < 10; ++d) {
                        int by, bx; bool active;
                        block_of(d, by, bx, active);
                        const int mode = mode_of(by, bx);
                        const int res_cur = res_nx;
                        const u32 ent_cur = ent_nx;
                        if (d + 1 < 10) {
                            int by1, bx1; bool a1;
                            block_of(d + 1, by1, bx1, a1);
                            res_nx = wl->res[(by1 * 4 + bx1) * 16 + (hl & 15)];
                            ent_nx = gtab[mode_of(by1, bx1) * 16 + (hl & 15)];
                        }
                        const unsigned char *org = tY + TY_AT(by * 4, bx * 4) - 64;
                        int pred;
                        if (mode == VP8IR_B_DC_PRED) {
                            const u32 W1 = *(const u32 *)(org + 64 - TY_STRIDE);
                            pred = (sad4(W1) + org[63] + org[63 + TY_STRIDE] + org[63 + 2 * TY_STRIDE]
                                    + org[63 + 3 * TY_STRIDE] + 4) >> 3;
                        } else {
                            const u32 e = ent_cur;
                            const int p0 = org[e & 0xff], p1 = org[(e >> 8) & 0xff], p2 = org[(e >> 16) & 0xff];
                            const int kind = e >> 24;
                            const int t3 = (p0 + 2 * p1 + p2 + 2) >> 2, t2 = (p1 + p2 + 1) >> 1, tm = clamp255(p0 + p1 - p2);
                            pred = kind == 2 ? t3 : (kind == 1 ? t2 : (kind == 0 ? p1 : tm));
                        }
                        const int v = clamp255(pred + res_cur);
                        if (active) tY[TY_AT(by * 4 + pr, bx * 4 + pc)] = (unsigned char)v;
                        wave_lds_sync();
                    }
                    outY0 = *(const u32 *)(tY + TY_AT(ly0, lx0));
                    outY1 = *(const u32 *)(tY + TY_AT(ly0 + 8, lx0));
                }
            } else if (INTER_DONE) {
                // ---- inter MB, finished by vp8_inter_mb_kernel: what this lane would have stored (load_done, a macroblock ago)
                outY0 = done.y0; outY1 = done.y1; outC = done.c;
            } else {
                // ---- inter MB
                mb_inter(g, wl, (g_cu8p)(half ? jobB.ref[ref_frame & 3] : jobA.ref[ref_frame & 3]),
                         (g_cmvp)(mvs + ((long)r * cols + c) * 16), mv2, flags, y_mode, r, c, bilinear, fullpix, ML, rY0, rY1, rC,
                         outY0, outY1, outC);
            }

            // ---- finished MB: frame (HBM, once), my line buffer (bottom rows), left column for MB c+1
            if (valid && !elsewhere) {
                if (tile_out) {
                    // large launches with inter frames: into the job's macroblock-tiled scratch frame (three 128-byte lines per
                    // macroblock, VP8_TILE_BYTES), which the lane-per-row loop filter takes from there
                    g_u8p t = tile_out + ((long)r * cols + c) * VP8_TILE_BYTES;
                    *(g_u32p)(t + ly0 * 16 + lx0) = outY0;
                    *(g_u32p)(t + (ly0 + 8) * 16 + lx0) = outY1;
                    *(g_u32p)(t + 256 + cpl * 64 + cy * 8 + cx0) = outC;
                } else {
                *(g_u32p)(dY + (long)ly0 * g.y_stride + c * 16 + lx0) = outY0;
                *(g_u32p)(dY + (long)(ly0 + 8) * g.y_stride + c * 16 + lx0) = outY1;
                *(g_u32p)((cpl ? dV : dU) + (long)cy * g.uv_stride + c * 8 + cx0) = outC;
                }
            }
            if (XCU) {
                if (ly0 == 7) gran_store(gran_mine + c * 4 + (lx0 >> 2), outY1, epoch);                 // pixel row 15
                if (cy == 7) gran_store(gran_mine + (cpl ? cols * 6 + 1 : cols * 4 + 1) + c * 2 + (cx0 >> 2), outC, epoch);
                if (c == cols - 1 && hl == 31) gran_store(gran_mine + cols * 4, (outY1 >> 24) * 0x01010101u, epoch);
            }
            if (!XCU && ly0 == 7) *(u32 *)(my_line + LINE_PAD + c * 16 + lx0) = outY1;          // pixel row 15
            if (lx0 == 12) {
                wl->lcol[ly0] = (unsigned char)(outY0 >> 24);
                wl->lcol[ly0 + 8] = (unsigned char)(outY1 >> 24);
                tY[TY_AT(ly0, -1)] = (unsigned char)(outY0 >> 24);
                tY[TY_AT(ly0 + 8, -1)] = (unsigned char)(outY1 >> 24);
            }
            if (!XCU && cy == 7) *(u32 *)(my_line + (cpl ? lV : lU) + LINE_PAD + c * 8 + cx0) = outC;
            if (cx0 == 4) wl->lcol[16 + cpl * 8 + cy] = (unsigned char)(outC >> 24);
            if (!XCU && c == cols - 1 && hl == 31) {
                // vp8_extend_mb_row (extend.c:160-185): what the next row's last MB sees as above-right
                // is the last pixel of this line replicated (hl 31 holds pixel row 15, x = 12..15).
                *(u32 *)(my_line + LINE_PAD + cols * 16) = (outY1 >> 24) * 0x01010101u;
            }
            if (!XCU) wg_publish_lds(&prog[wave], c + 1 == cols ? (k + 1) << 16 : (k << 16) + c + 1, lane);
        };

        // ---- software pipeline, unrolled by two: MB descriptors two ahead (issued before the current MB is worked
        // on), coefficients and MVs one ahead -- by then the descriptor says whether the MB has coefficients at all
        auto skipped = [&](u32 d) -> bool {          // ... or is none of this pass's business
            const u32 w0 = half_sel(d, 0);
            return ((w0 >> 24) & VP8IR_MB_SKIP) || (INTER_DONE && ((w0 >> 16) & 0xff) != VP8IR_INTRA_FRAME);
        };
        u32 dA = load_desc(0), dB = cols > 1 ? load_desc(1) : 0u;
        Coefs qA = load_coefs(0, dA, skipped(dA)), qB = qA;
        Mv2 mA = load_mv(0), mB = mA;
        Done pA = load_done(0, dA), pB = pA;
        for (int c0 = 0; c0 < cols; c0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = c0 + u;
                if (c < cols) {
                    const u32 d = u ? dB : dA;
                    const u32 dn = u ? dA : dB;          // descriptor of MB c+1
                    if (c + 2 < cols) { if (u) dB = load_desc(c + 2); else dA = load_desc(c + 2); }
                    if (c + 1 < cols) {
                        const bool sk = skipped(dn);
                        if (u) { qA = load_coefs(c + 1, dn, sk); mA = load_mv(c + 1); pA = load_done(c + 1, dn); }
                        else { qB = load_coefs(c + 1, dn, sk); mB = load_mv(c + 1); pB = load_done(c + 1, dn); }
                    }
                    process(c, d, u ? qB : qA, u ? mB : mA, u ? pB : pA);
                }
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(768)
vp8_recon_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    recon_body<false, false>(jobs, njobs, g, nullptr, 0u, 1, nullptr, nullptr);
}
extern "C" __global__ void __launch_bounds__(768)
vp8_recon_intra_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, const unsigned int *intra_flags)
{
    recon_body<false, true>(jobs, njobs, g, nullptr, 0u, 1, nullptr, intra_flags);
}

// grid = 8 * S * ceil(npairs / 8) workgroups of four or eight waves; gran: npairs * 2 * rows * (cols * 8 + 2) granules
extern "C" __global__ void __launch_bounds__(512)
vp8_recon_xcu_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                     int S, int *err)
{
    recon_body<true, false>(jobs, njobs, g, gran, epoch, S, err, nullptr);
}
extern "C" __global__ void __launch_bounds__(512)
vp8_recon_intra_xcu_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                           int S, int *err, const unsigned int *intra_flags)
{
    recon_body<true, true>(jobs, njobs, g, gran, epoch, S, err, intra_flags);
}

// ---- inter macroblocks, every one on its own -------------------------------------------------------------------
// An inter macroblock is a function of the reference frames, its motion vectors and its coefficients only
// (vp8_build_inter_predictors_mb + the residual, decodframe.c:112-296): nothing of the frame being decoded goes in.
// So they are not walked row by row behind their neighbours: a half wave (32 lanes, the roles of MbLanes) takes any
// macroblock, two macroblocks per wave, workgroups stride over the (job, macroblock pair) space; with a few KB of LDS and
// no ordering there are eight waves per SIMD to hide the reference fetches behind.  Intra macroblocks of the same frames
// are left to vp8_recon_intra_kernel, which finds out from intra_flags whether a frame has any.
extern "C" __global__ void __launch_bounds__(256)
vp8_inter_mb_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, unsigned int *__restrict__ intra_flags)
{
    __shared__ WaveLds s_wl[8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, hl = lane & 31;
    const MbLanes ML = mb_lanes(hl);
    WaveLds *wl = &s_wl[wave * 2 + half];
    const int cols = g.mb_cols, nmb = cols * g.mb_rows, upj = (nmb + 1) >> 1;
    const long total = (long)njobs * upj;
    int cur_job = -1;
    for (long U = (long)blockIdx.x * 4 + wave; U < total; U += (long)gridDim.x * 4) {
        const int j = (int)(U / upj), u = (int)(U - (long)j * upj);
        const DevJob &job = jobs[j];
        const vp8ir_frame_hdr &h = job.hdr;
        if (h.frame_type == 0) {                 // a key frame in the launch: all of it is the other kernel's
            if (u == 0 && lane == 0) atomicOr(&intra_flags[j], 1u);
            continue;
        }
        if (j != cur_job) {
            wave_lds_sync();
            build_dequant(h, wl->dq, hl);
            cur_job = j;
            wave_lds_sync();
        }
        const int n = 2 * u + half;
        const bool valid = n < nmb;
        const int nn = valid ? n : nmb - 1;
        const int r = nn / cols, c = nn - r * cols;
        const u32 d = hl < 16 ? ((g_cu32p)(job.mbx + nn))[hl] : 0u;
        auto half_sel = [&](int idx) -> u32 {
            const u32 a = (u32)__builtin_amdgcn_readlane((int)d, idx), b = (u32)__builtin_amdgcn_readlane((int)d, 32 + idx);
            return half ? b : a;
        };
        const u32 w0 = half_sel(0), w1 = half_sel(1);
        const int y_mode = w0 & 0xff, ref_frame = (w0 >> 16) & 0xff;
        const u32 flags = w0 >> 24;
        const bool skip = flags & VP8IR_MB_SKIP;
        const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
        const bool inter = valid && ref_frame != VP8IR_INTRA_FRAME;
        if (valid && !inter && hl == 0) atomicOr(&intra_flags[j], 1u);
        Coefs q;
        q.y0 = q.y1 = q.c = q.y2 = (coef4){ 0, 0 };
        Mv2 mv2 = { 0u, 0u };
        g_cmvp mv = (g_cmvp)(job.mvs + (long)nn * 16);
        if (inter) {
            mv2.a = mv[hl >> 2]; mv2.b = mv[8 + (hl >> 2)];
            if (!skip) {
                MbWords w;
#pragma unroll
                for (int qq = 0; qq < 6; qq++) w.e[qq] = half_sel(2 + qq);
                w.first = half_sel(14);
                q = load_coefs_dev((g_cu32p)(job.mbx + nn), (g_cs16p)job.blocks, w, has_y2, hl);
            }
        }
        int rY0[4], rY1[4], rC[4];
        mb_residual(wl, q, skip || !inter, has_y2, w1 & 3, ML, lane, rY0, rY1, rC);
        if (inter) {
            u32 outY0, outY1, outC;
            mb_inter(g, wl, (g_cu8p)job.ref[ref_frame & 3], mv, mv2, flags, y_mode, r, c, h.version != 0, h.version == 3, ML,
                     rY0, rY1, rC, outY0, outY1, outC);
            if (job.tile) {
                g_u8p t = (g_u8p)job.tile + (long)nn * VP8_TILE_BYTES;
                *(g_u32p)(t + ML.ly0 * 16 + ML.lx0) = outY0;
                *(g_u32p)(t + (ML.ly0 + 8) * 16 + ML.lx0) = outY1;
                *(g_u32p)(t + 256 + ML.cpl * 64 + ML.cy * 8 + ML.cx0) = outC;
            } else {
            g_u8p dY = (g_u8p)(job.dst + g.y_off + (long)r * 16 * g.y_stride);
            g_u8p dC = (g_u8p)(job.dst + (ML.cpl ? g.v_off : g.u_off) + (long)r * 8 * g.uv_stride);
            *(g_u32p)(dY + (long)ML.ly0 * g.y_stride + c * 16 + ML.lx0) = outY0;
            *(g_u32p)(dY + (long)(ML.ly0 + 8) * g.y_stride + c * 16 + ML.lx0) = outY1;
            *(g_u32p)(dC + (long)ML.cy * g.uv_stride + c * 8 + ML.cx0) = outC;
            }
        }
        wave_lds_sync();                         // wl->res / wl->wht_dc are reused by the next macroblock
    }
}
