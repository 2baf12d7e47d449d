// VP8 reconstruction, "one macroblock row per LANE" formulation for gfx950.
//
// Same job as vp8_recon.hip (decode_mb_row / decode_macroblock, vp8/decoder/decodframe.c:112-436, and
// what it reaches through RTCD: dequantize.c, idctllm.c, idct_blk.c, reconintra.c, reconintra4x4.c,
// reconinter.c, filter.c), organised around what the first kernel taught us on MI355X: the path is
// bound by VALU *issue* (one wave instruction costs four SIMD cycles however many lanes are live),
// and a macroblock offers at most 16..32 lanes of parallel work once the 4x4 intra chain is
// respected.  So the lanes of a wave are not spent inside a macroblock at all:
//
//   * lane p of a wave owns macroblock ROWS p, p+G, p+2G, ... of a strand of frames (G = lanes per
//     strand, a power of two <= 64; a wave carries 64/G strands) and walks each row left to right, one
//     whole macroblock per step, as straight per-lane code (the reference's C, restated per lane);
//   * lane p runs two macroblocks behind lane p-1 -- the intra dependency (left, above, above-right)
//     is then satisfied by construction, with no flags, no polling and no barriers: a step is one
//     pass of all 64 lanes over 64 different macroblocks of the classic 2-D wavefront;
//   * the unfiltered pixels above a macroblock are the bottom line of the macroblock the lane above
//     finished two steps ago: they travel by DPP wave shift (v_mov_b32 wave_shr:1), not through memory.
//     The first lane of a strand has its predecessor row on the LAST lane of the strand, G rows of
//     work earlier; it reads that line back from the frame in HBM (L2-coherent loads), which the
//     step period P >= 2G+2 guarantees was written at least three steps before;
//   * the 4x4 intra chain of B_PRED macroblocks runs inside the lane on packed bytes: the edge
//     vector's 3-tap and 2-tap smoothings are v_lerp_u8 on four pixels at a time, the ten predictors
//     are byte shuffles (v_perm_b32 / v_alignbyte_b32) of those; divergence between lanes costs the
//     union of the modes present, not a serial chain per macroblock;
//   * no LDS for intra frames; inter prediction stages 16 predicted 4x4 rows per group in LDS
//     (lane-interleaved dwords: conflict-free) so the filter code exists once, in a rolled loop.
//
// Integer only (u8 pixels, i16 coefficients, i32 accumulators); no MFMA by design.
#include "vp8_common.hip.h"
#include <stddef.h>

namespace {

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef GLOBAL_AS const u32x4 *g_cu32x4p;
typedef GLOBAL_AS const u32x2 *g_cu32x2p;
typedef GLOBAL_AS u32x4 *g_u32x4p;
typedef GLOBAL_AS u32x2 *g_u32x2p;

typedef short v2s __attribute__((ext_vector_type(2)));      // two 16-bit lanes: v_pk_* arithmetic
__device__ __forceinline__ v2s as_v2s(u32 v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ u32 as_u32(v2s v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ v2s pk(int lo, int hi) { return (v2s){ (short)lo, (short)hi }; }
__device__ __forceinline__ v2s clamp255_2(v2s v)
{
    return __builtin_elementwise_min(__builtin_elementwise_max(v, pk(0, 0)), pk(255, 255));
}
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ u32 alignb(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ u32 lerp(u32 a, u32 b, u32 c) { return __builtin_amdgcn_lerp(a, b, c); }
__device__ __forceinline__ int sad4(u32 v) { return (int)__builtin_amdgcn_sad_u8(v, 0u, 0u); }
__device__ __forceinline__ u32 splat(int v) { return (u32)v * 0x01010101u; }
__device__ __forceinline__ int sext16(u32 v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(u32 v) { return (int)v >> 16; }
// value held by the lane above (lane l-1); lane 0 keeps its own
__device__ __forceinline__ u32 from_lane_above(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// load that must observe another lane's earlier store to the frame: served by L2, never by the CU's L1
__device__ __forceinline__ u32 load_l2(const unsigned char *p)
{
    return __hip_atomic_load((const u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// clamp255(v >> 7) for the filter passes.  The empty asm keeps LLVM (ROCm 7.2) from fusing shift, clamp
// and byte packing into v_ashr_pk_u8_i32: on gfx950 that instruction leaves the upper half of its
// destination register untouched while the compiler assumes it is zeroed, which ORs stale bytes into
// pixels 2 and 3 of the packed row (found by the one-MB inter fuzz cases).
__device__ __forceinline__ int shr7_clamp255(int v)
{
    int t = v >> 7;
    asm volatile("" : "+v"(t));
    return clamp255(t);
}

// sub-pixel filter taps (vp8/common/filter.c:16-39), padded to 8 shorts per phase
__constant__ __attribute__((aligned(16))) const short k_sixtap8[8][8] = {
    { 0, 0, 128, 0, 0, 0, 0, 0 }, { 0, -6, 123, 12, -1, 0, 0, 0 }, { 2, -11, 108, 36, -8, 1, 0, 0 },
    { 0, -9, 93, 50, -6, 0, 0, 0 }, { 3, -16, 77, 77, -16, 3, 0, 0 }, { 0, -6, 50, 93, -9, 0, 0, 0 },
    { 1, -8, 36, 108, -11, 2, 0, 0 }, { 0, -1, 12, 123, -6, 0, 0, 0 }
};

// one 1-D pass of vp8_short_idct4x4llm_c (idctllm.c:39-60 / 65-88) without the final rounding
__device__ __forceinline__ void idct1d(int i0, int i1, int i2, int i3, int &o0, int &o1, int &o2, int &o3)
{
    const int a1 = i0 + i2, b1 = i0 - i2;
    const int c1 = ((i1 * 35468) >> 16) - (i3 + ((i3 * 20091) >> 16));
    const int d1 = (i1 + ((i1 * 20091) >> 16)) + ((i3 * 35468) >> 16);
    o0 = a1 + d1; o3 = a1 - d1; o1 = b1 + c1; o2 = b1 - c1;
}

// vp8_dequant_idct_add_c (dequantize.c:29-44) on one block held by one lane.
// cq: the block's 16 coefficients as loaded (IR order: column-major, two per dword);
// dc_in: the already dequantised DC when the MB has a Y2 block (dequant factor 1, decodframe.c:92).
// res[row*4+col] = the residual the reference adds to the predictor.
__device__ __forceinline__ void dequant_idct(const u32x4 ca, const u32x4 cb, int dqdc, int dqac, bool dc_given, int dc_in,
                                             int res[16])
{
    const u32 q[8] = { ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w };
    int t[16];                                   // t[row*4+col], i16 like the reference's `short output[16]`
#pragma unroll
    for (int col = 0; col < 4; col++) {
        // DQ = (short)(Q * DQC) (dequantize.c:17-27): the low half of a 16x16 product, two coefficients per v_pk_mul_lo_u16
        const v2s p01 = as_v2s(q[2 * col]) * (col == 0 ? pk(dqdc, dqac) : pk(dqac, dqac));
        const v2s p23 = as_v2s(q[2 * col + 1]) * pk(dqac, dqac);
        int i0 = p01.x;
        if (col == 0 && dc_given) i0 = dc_in;
        const int i1 = p01.y, i2 = p23.x, i3 = p23.y;
        int o0, o1, o2, o3;
        idct1d(i0, i1, i2, i3, o0, o1, o2, o3);  // vertical pass: column `col`, rows 0..3
        t[0 + col] = (short)o0; t[4 + col] = (short)o1; t[8 + col] = (short)o2; t[12 + col] = (short)o3;
    }
#pragma unroll
    for (int row = 0; row < 4; row++) {
        int o0, o1, o2, o3;
        idct1d(t[row * 4], t[row * 4 + 1], t[row * 4 + 2], t[row * 4 + 3], o0, o1, o2, o3);
        res[row * 4 + 0] = (o0 + 4) >> 3; res[row * 4 + 1] = (o1 + 4) >> 3;
        res[row * 4 + 2] = (o2 + 4) >> 3; res[row * 4 + 3] = (o3 + 4) >> 3;
    }
}

// clamp(pred + residual) for a row of four pixels (the tail of vp8_short_idct4x4llm / vp8_dc_only_idct_add), two
// pixels per instruction: predictor bytes widened to 16-bit lanes, residuals (|r| < 2^12) packed beside them
__device__ __forceinline__ u32 add_clamp_pack(u32 pred, const int *r)
{
    const v2s lo = clamp255_2(as_v2s(perm(pred, pred, 0x0c010c00u)) + pk(r[0], r[1]));
    const v2s hi = clamp255_2(as_v2s(perm(pred, pred, 0x0c030c02u)) + pk(r[2], r[3]));
    return perm(as_u32(hi), as_u32(lo), 0x06040200u);
}

// vp8_dc_only_idct_add_c (idctllm.c:112-137): a block with eob <= 1 (idct_blk.c:28-37) adds (dc + 4) >> 3 to all 16
// predictor pixels.  Same result as the full transform of a DC-only block, at a sixth of its cost.
__device__ __forceinline__ u32 add_dc_clamp_pack(u32 pred, v2s d)
{
    const v2s lo = clamp255_2(as_v2s(perm(pred, pred, 0x0c010c00u)) + d);
    const v2s hi = clamp255_2(as_v2s(perm(pred, pred, 0x0c030c02u)) + d);
    return perm(as_u32(hi), as_u32(lo), 0x06040200u);
}

// TM prediction of a row of four pixels: clamp(above[i] + left - top_left), above given as two packed pairs
__device__ __forceinline__ u32 tm_row(v2s a01, v2s a23, int l_minus_tl)
{
    const v2s d = pk(l_minus_tl, l_minus_tl);
    return perm(as_u32(clamp255_2(a23 + d)), as_u32(clamp255_2(a01 + d)), 0x06040200u);
}

// right-hand pixel column of a 4x4 block given as four row dwords -> one dword, top pixel in byte 0
__device__ __forceinline__ u32 right_column(const u32 o[4])
{
    return perm(perm(o[3], o[2], 0x0c0c0703u), perm(o[1], o[0], 0x0c0c0703u), 0x05040100u);
}

// Whole-block predictors DC / V / H / TM (reconintra.c:139-241, 403-521) for one 4x4 block:
// above = the 4 pixels above the block's columns, left = the 4 pixels left of its rows (top in byte 0).
__device__ __forceinline__ void mb_mode_pred(int mode, u32 above, u32 left, int tl, int dc, u32 p[4])
{
    if (mode == VP8IR_DC_PRED) { p[0] = p[1] = p[2] = p[3] = splat(dc); }
    else if (mode == VP8IR_V_PRED) { p[0] = p[1] = p[2] = p[3] = above; }
    else if (mode == VP8IR_H_PRED) {
        p[0] = perm(left, left, 0x00000000u); p[1] = perm(left, left, 0x01010101u);
        p[2] = perm(left, left, 0x02020202u); p[3] = perm(left, left, 0x03030303u);
    } else {
        const v2s a01 = as_v2s(perm(above, above, 0x0c010c00u)), a23 = as_v2s(perm(above, above, 0x0c030c02u));
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = tm_row(a01, a23, (int)((left >> (8 * j)) & 0xff) - tl);
    }
}

// vp8_intra4x4_predict (reconintra4x4.c:16-303) for one block.  a0 = above 4 pixels, a1 = the next 4
// (above-right), left = left 4 pixels (top in byte 0), tl = top-left.  Edge vector as in the oracle:
// P[0..14] = { L3, L3, L2, L1, L0, TL, A0..A7, A7 }; F[k] = (P[k-1]+2P[k]+P[k+1]+2)>>2,
// G[k] = (P[k]+P[k+1]+1)>>1, both computed four pixels per instruction with v_lerp_u8:
// (a+2b+c+2)>>2 == (((a+c)>>1) + b + 1)>>1 exactly.
__device__ __forceinline__ void bpred4x4(int mode, u32 a0, u32 a1, u32 left, int tl, u32 p[4])
{
    if (mode == VP8IR_B_DC_PRED) {
        p[0] = p[1] = p[2] = p[3] = splat((sad4(a0) + sad4(left) + 4) >> 3);
        return;
    }
    if (mode == VP8IR_B_TM_PRED) {
        const v2s a01 = as_v2s(perm(a0, a0, 0x0c010c00u)), a23 = as_v2s(perm(a0, a0, 0x0c030c02u));
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = tm_row(a01, a23, (int)((left >> (8 * j)) & 0xff) - tl);
        return;
    }
    const u32 E0 = perm(left, left, 0x01020303u);                       // L3 L3 L2 L1
    const u32 E1 = perm(a0, left, 0x05040c00u) | ((u32)tl << 8);        // L0 TL A0 A1
    const u32 E2 = alignb(a1, a0, 2);                                   // A2 A3 A4 A5
    const u32 E3 = perm(a1, a1, 0x03030302u);                           // A6 A7 A7 A7
    // neighbours: M_w[j] = P[4w+j-1], N_w[j] = P[4w+j+1]
    const u32 N0 = alignb(E1, E0, 1), N1 = alignb(E2, E1, 1), N2 = alignb(E3, E2, 1), N3 = E3 >> 8;
    const u32 M0 = E0 << 8, M1 = alignb(E1, E0, 3), M2 = alignb(E2, E1, 3), M3 = alignb(E3, E2, 3);
    const u32 one = 0x01010101u;
    const u32 F0 = lerp(lerp(M0, N0, 0), E0, one), F1 = lerp(lerp(M1, N1, 0), E1, one);
    const u32 F2 = lerp(lerp(M2, N2, 0), E2, one), F3 = lerp(lerp(M3, N3, 0), E3, one);
    const u32 G0 = lerp(E0, N0, one), G1 = lerp(E1, N1, one), G2 = lerp(E2, N2, one);
    switch (mode) {
    case VP8IR_B_VE_PRED: p[0] = p[1] = p[2] = p[3] = alignb(F2, F1, 2); break;          // F6..F9
    case VP8IR_B_HE_PRED:                                                                  // F4, F3, F2, F1
        p[0] = perm(F1, F0, 0x04040404u); p[1] = perm(F1, F0, 0x03030303u);
        p[2] = perm(F1, F0, 0x02020202u); p[3] = perm(F1, F0, 0x01010101u);
        break;
    case VP8IR_B_LD_PRED:                                                                  // F[7+r ..]
        p[0] = alignb(F2, F1, 3); p[1] = F2; p[2] = alignb(F3, F2, 1); p[3] = alignb(F3, F2, 2);
        break;
    case VP8IR_B_RD_PRED:                                                                  // F[5-r ..]
        p[0] = alignb(F2, F1, 1); p[1] = F1; p[2] = alignb(F1, F0, 3); p[3] = alignb(F1, F0, 2);
        break;
    case VP8IR_B_VR_PRED:
        p[0] = alignb(G2, G1, 1);                 // G5 G6 G7 G8
        p[1] = alignb(F2, F1, 1);                 // F5 F6 F7 F8
        p[2] = perm(G1, F1, 0x07060500u);         // F4 G5 G6 G7
        p[3] = perm(F1, F0, 0x07060503u);         // F3 F5 F6 F7
        break;
    case VP8IR_B_VL_PRED:
        p[0] = alignb(G2, G1, 2);                 // G6 G7 G8 G9
        p[1] = alignb(F2, F1, 3);                 // F7 F8 F9 F10
        p[2] = perm(F2, alignb(G2, G1, 3), 0x07020100u);   // G7 G8 G9 F11
        p[3] = perm(F3, F2, 0x04020100u);         // F8 F9 F10 F12
        break;
    case VP8IR_B_HD_PRED:
        p[0] = perm(F1, G1, 0x07060500u);                             // G4 F5 F6 F7
        p[1] = perm(perm(F1, G1, 0x0500040cu), G0, 0x07060503u);      // G3 F4 G4 F5
        p[2] = perm(F1, perm(F0, G0, 0x0c030702u), 0x04020100u);      // G2 F3 G3 F4
        p[3] = perm(F0, G0, 0x07020601u);                             // G1 F2 G2 F3
        break;
    default: /* VP8IR_B_HU_PRED */
        p[0] = perm(F0, G0, 0x06020703u);                             // G3 F3 G2 F2
        p[1] = perm(F0, G0, 0x05010602u);                             // G2 F2 G1 F1
        p[2] = perm(E0, perm(F0, G0, 0x0c0c0501u), 0x05050100u);      // G1 F1 L3 L3
        p[3] = perm(E0, E0, 0x01010101u);                             // L3 x4
        break;
    }
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

// Sub-pixel prediction of one 4x4 block (reconinter.c:161-227 + filter.c:41-128, 376-494) at a final
// MV, into four row dwords.  One code path: the six-tap filter with both passes always, as the
// reference runs it; bilinear (profiles 1..3) is the same arithmetic with taps {0,0,128-16f,16f,0,0}
// (its first pass needs no clamp and the rounding is identical), and a whole-pixel MV -- a plain copy
// in the reference -- is the identity taps {0,0,128,0,0,0}.  Rows are fetched as aligned dwords and
// shifted into place, 27 loads per block.
__device__ __forceinline__ void inter4x4(g_cu8p plane, int stride, int x, int y, int mvrow, int mvcol, bool bilinear,
                                         int w, int h, int border, u32 out[4])
{
    int sx = x + (mvcol >> 3), sy = y + (mvrow >> 3);
    const int fx = mvcol & 7, fy = mvrow & 7;
    // memory safety only (a conforming stream never triggers these): keep every tap inside the
    // allocated plane incl. its border
    sx = max(-border + 2, min(sx, w + border - 10));
    sy = max(-border + 2, min(sy, h + border - 7));
    int hx[6], vy[6];
    if (bilinear) {
        hx[0] = hx[1] = hx[4] = hx[5] = 0; hx[2] = 128 - 16 * fx; hx[3] = 16 * fx;
        vy[0] = vy[1] = vy[4] = vy[5] = 0; vy[2] = 128 - 16 * fy; vy[3] = 16 * fy;
    } else {
        const u32x4 tx = *(const u32x4 *)k_sixtap8[fx], ty = *(const u32x4 *)k_sixtap8[fy];
        hx[0] = sext16(tx.x); hx[1] = hi16(tx.x); hx[2] = sext16(tx.y); hx[3] = hi16(tx.y); hx[4] = sext16(tx.z); hx[5] = hi16(tx.z);
        vy[0] = sext16(ty.x); vy[1] = hi16(ty.x); vy[2] = sext16(ty.y); vy[3] = hi16(ty.y); vy[4] = sext16(ty.z); vy[5] = hi16(ty.z);
    }
    g_cu8p s = plane + (long)(sy - 2) * stride + (sx - 2);
    const u32 sh = (u32)(unsigned long)s & 3u;
    g_cu32p sa = (g_cu32p)(s - sh);
    int acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 64;
#pragma unroll
    for (int rr = 0; rr < 9; rr++) {
        g_cu32p rowp = (g_cu32p)((g_cu8p)sa + (long)rr * stride);
        const u32 d0 = rowp[0], d1 = rowp[1], d2 = rowp[2];
        const u32 w0 = alignb(d1, d0, sh), w1 = alignb(d2, d1, sh), w2 = alignb(0u, d2, sh);
        int px[9];
#pragma unroll
        for (int i = 0; i < 4; i++) { px[i] = (w0 >> (8 * i)) & 0xff; px[4 + i] = (w1 >> (8 * i)) & 0xff; }
        px[8] = w2 & 0xff;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int t = px[i] * hx[0] + px[i + 1] * hx[1] + px[i + 2] * hx[2] + px[i + 3] * hx[3] + px[i + 4] * hx[4]
                        + px[i + 5] * hx[5] + 64;
            const int f = shr7_clamp255(t);      // first-pass output row rr (= source row rr-2)
#pragma unroll
            for (int j = 0; j < 4; j++) {        // feeds output row j with vertical tap rr-j
                const int k = rr - j;
                if (k >= 0 && k < 6) acc[j * 4 + i] += f * vy[k];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) o |= (u32)shr7_clamp255(acc[j * 4 + i]) << (8 * i);
        out[j] = o;
    }
}

// six dequantisation factors of one segment (vp8cx_init_de_quantizer + mb_init_dequantizer,
// decodframe.c:50-109, quant_common.c:39-132): packed as (y1dc | y1ac<<16, y2dc | y2ac<<16, uvdc | uvac<<16)
__device__ __forceinline__ void segment_dequant(const vp8ir_frame_hdr &h, int seg, u32 dq[3])
{
    int q = h.base_qindex;
    if (h.segmentation_enabled) q = h.mb_segment_abs_delta ? h.segment_quant[seg] : q + h.segment_quant[seg];
    q = q < 0 ? 0 : (q > 127 ? 127 : q);
    auto qi = [&](int delta) { const int v = q + delta; return v < 0 ? 0 : (v > 127 ? 127 : v); };
    const int y1dc = k_dc_q[qi(h.y1dc_delta_q)], y1ac = k_ac_q[q];
    const int y2dc = k_dc_q[qi(h.y2dc_delta_q)] * 2;
    int y2ac = (k_ac_q[qi(h.y2ac_delta_q)] * 155) / 100; if (y2ac < 8) y2ac = 8;
    int uvdc = k_dc_q[qi(h.uvdc_delta_q)]; if (uvdc > 132) uvdc = 132;
    const int uvac = k_ac_q[qi(h.uvac_delta_q)];
    dq[0] = (u32)y1dc | ((u32)y1ac << 16); dq[1] = (u32)y2dc | ((u32)y2ac << 16); dq[2] = (u32)uvdc | ((u32)uvac << 16);
}

} // namespace

// grid = waves (one wave per block); lgG = log2(lanes per strand); P = steps per row period, >= max(cols, 2G+2);
// nstrands = total strands of the launch: strand q reconstructs jobs q, q+nstrands, ...
// tiled != 0: the frame goes to the job's macroblock-tiled scratch (DevJob::ref[0], VP8_TILE_BYTES per MB:
// 16 luma rows of 16 B, 8 U rows of 8 B, 8 V rows of 8 B) instead of the raster frame buffer, so that every
// lane writes whole 128-byte lines; vp8_detile_kernel converts after the loop filter.
#ifndef VP8_RECON_SIMT_WAVES_PER_SIMD
#define VP8_RECON_SIMT_WAVES_PER_SIMD 1      // measurement knob: 2 caps the kernel at 256 VGPRs so two waves share a SIMD
#endif
extern "C" __global__ void __launch_bounds__(64, VP8_RECON_SIMT_WAVES_PER_SIMD)
vp8_recon_simt_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int tiled)
{
    __shared__ u32 predlds[16 * 64];            // inter prediction of the current group: [block*4+row][lane]
    const int lane = threadIdx.x;
    const int G = 1 << lgG;
    const int pos = lane & (G - 1);
    const int spw = 64 >> lgG;
    const int strand = blockIdx.x * spw + (lane >> lgG);
    const int cols = g.mb_cols, rows = g.mb_rows;
    // byte steps of the destination: between pixel rows, and between horizontally adjacent macroblocks
    const long ysY = tiled ? 16 : g.y_stride, ysC = tiled ? 8 : g.uv_stride;
    const long mbY = tiled ? VP8_TILE_BYTES : 16, mbC = tiled ? VP8_TILE_BYTES : 8;
    const long upY = tiled ? (long)cols * VP8_TILE_BYTES - 15 * 16 : g.y_stride;     // from row 0 of an MB back to row 15 of the MB above
    const long upC = tiled ? (long)cols * VP8_TILE_BYTES - 7 * 8 : g.uv_stride;
    const int myjobs = strand < njobs ? (njobs - strand + nstrands - 1) / nstrands : 0;
    const int Vmax = myjobs * rows;
    const int wavejobs = (njobs - (int)blockIdx.x * spw + nstrands - 1) / nstrands;     // first strand: the most jobs
    const int T = ((wavejobs * rows + G - 1) >> lgG) * P + 2 * (G - 1);

    // ---- per-lane row state
    g_cu32p mbp = nullptr; g_cs16p cfp = nullptr; g_cu32p mvp = nullptr;
    g_u8p dY = nullptr, dU = nullptr, dV = nullptr;
    const DevJob *job = jobs;
    int r = 0;
    bool bilinear = false, fullpix = false;
    u32 dqs[4][3];
#pragma unroll
    for (int s = 0; s < 4; s++) dqs[s][0] = dqs[s][1] = dqs[s][2] = 0;
    // unfiltered context: left columns (top pixel in byte 0), last pixels of the previous step's above lines
    u32 lY[4] = { 0, 0, 0, 0 }, lU[2] = { 0, 0 }, lV[2] = { 0, 0 };
    int prevLastY = 0, prevLastU = 0, prevLastV = 0;
    // bottom lines of the macroblocks finished one and two steps ago (what the lane below asks for)
    u32 h1Y[4] = { 0, 0, 0, 0 }, h1U[2] = { 0, 0 }, h1V[2] = { 0, 0 };
    u32 h2Y[4] = { 0, 0, 0, 0 }, h2U[2] = { 0, 0 }, h2V[2] = { 0, 0 };

    // software pipeline: the descriptor, the Y2 block and the first coefficient group of the NEXT macroblock
    // of the row are fetched while the current one is being finished; inside a macroblock every group of
    // four blocks is fetched one group ahead (`nxt`) and becomes `cur` when its turn comes.
    u32x4 pf_m0 = { 0, 0, 0, 0 }, pf_m1 = { 0, 0, 0, 0 };   // descriptor words 0..7: modes, segment, eobs of blocks 0..23
    u32x4 pf_bm = { 0, 0, 0, 0 }, pf_y2a = { 0, 0, 0, 0 }, pf_y2b = { 0, 0, 0, 0 };
    u32x4 cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cur[i] = nxt[i] = (u32x4){ 0, 0, 0, 0 };

    int c = -2 * pos, V = pos;
#pragma unroll 1
    for (int t = 0; t < T; ++t, ++c) {
        if (c == P) { c = 0; V += G; }
        // what the lane above finished: two steps ago (straight above) and last step (above-right)
        u32 nY[4], nU[2], nV[2];
#pragma unroll
        for (int i = 0; i < 4; i++) nY[i] = from_lane_above(h2Y[i]);
        const u32 nAR = from_lane_above(h1Y[0]);
#pragma unroll
        for (int i = 0; i < 2; i++) { nU[i] = from_lane_above(h2U[i]); nV[i] = from_lane_above(h2V[i]); }
        u32 bY[4] = { h1Y[0], h1Y[1], h1Y[2], h1Y[3] }, bU[2] = { h1U[0], h1U[1] }, bV[2] = { h1V[0], h1V[1] };

        const bool act = c >= 0 && c < cols && V < Vmax;
        if (act) {
            if (c == 0) {
                // ---- new macroblock row: which frame, which row; pointers and quantisers
                const int j = V / rows;
                r = V - j * rows;
                job = jobs + (strand + j * nstrands);
                const vp8ir_frame_hdr &h = job->hdr;
                bilinear = h.version != 0; fullpix = h.version == 3;
                const int nseg = h.segmentation_enabled ? 4 : 1;
                for (int s = 0; s < 4; s++) {
                    u32 d[3];
                    if (s < nseg) segment_dequant(h, s, d);
                    else { d[0] = dqs[0][0]; d[1] = dqs[0][1]; d[2] = dqs[0][2]; }
                    dqs[s][0] = d[0]; dqs[s][1] = d[1]; dqs[s][2] = d[2];
                }
                mbp = (g_cu32p)(job->mbs + (long)r * cols);
                cfp = (g_cs16p)(job->coef + (long)r * cols * VP8IR_COEF_PER_MB);
                mvp = (g_cu32p)(job->mvs + (long)r * cols * 16);
                if (tiled) {
                    dY = (g_u8p)(const_cast<uint8_t *>(job->ref[0]) + (long)r * cols * VP8_TILE_BYTES);   // tile (r, 0)
                    dU = dY + 256; dV = dY + 320;
                } else {
                    uint8_t *dst = job->dst;
                    dY = (g_u8p)(dst + g.y_off + (long)r * 16 * g.y_stride);
                    dU = (g_u8p)(dst + g.u_off + (long)r * 8 * g.uv_stride);
                    dV = (g_u8p)(dst + g.v_off + (long)r * 8 * g.uv_stride);
                }
                lY[0] = lY[1] = lY[2] = lY[3] = 0x81818181u;    // left border 129 (setupintrarecon.c:15-32)
                lU[0] = lU[1] = lV[0] = lV[1] = 0x81818181u;
                // nothing was prefetched for the first macroblock of a row
                pf_m0 = *(g_cu32x4p)mbp; pf_m1 = *(g_cu32x4p)(mbp + 4); pf_bm = *(g_cu32x4p)(mbp + 10);
                pf_y2a = *(g_cu32x4p)(cfp + 384); pf_y2b = *(g_cu32x4p)(cfp + 392);
#pragma unroll
                for (int i = 0; i < 8; i++) cur[i] = *(g_cu32x4p)(cfp + i * 8);
            }
            const bool top = r == 0;
            // ---- macroblock descriptor
            const u32 w0 = pf_m0.x, w1 = pf_m0.y;
            // eobs (detokenize.c:363), a byte per block: luma block rows 0..3, then U | V
            const u32 eobY[4] = { pf_m0.z, pf_m0.w, pf_m1.x, pf_m1.y }, eobU = pf_m1.z, eobV = pf_m1.w;
            const u32x4 bm = pf_bm;
            const int y_mode = w0 & 0xff, uv_mode = (w0 >> 8) & 0xff, ref_frame = (w0 >> 16) & 0xff;
            const u32 flags = w0 >> 24;
            const bool skip = flags & VP8IR_MB_SKIP;
            const bool intra = ref_frame == VP8IR_INTRA_FRAME;
            const bool bpred = intra && y_mode == VP8IR_B_PRED;
            const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
            const int seg = w1 & 3;
            const u32 dq0 = seg == 0 ? dqs[0][0] : seg == 1 ? dqs[1][0] : seg == 2 ? dqs[2][0] : dqs[3][0];
            const u32 dq1 = seg == 0 ? dqs[0][1] : seg == 1 ? dqs[1][1] : seg == 2 ? dqs[2][1] : dqs[3][1];
            const u32 dq2 = seg == 0 ? dqs[0][2] : seg == 1 ? dqs[1][2] : seg == 2 ? dqs[2][2] : dqs[3][2];

            // ---- unfiltered line above (127 above the frame; vp8_setup_intra_recon)
            u32 aY[4], arY, aU[2], aV[2];
            if (top) {
                aY[0] = aY[1] = aY[2] = aY[3] = arY = 0x7f7f7f7fu;
                aU[0] = aU[1] = aV[0] = aV[1] = 0x7f7f7f7fu;
            } else if (pos == 0) {
                const unsigned char *pa = (const unsigned char *)dY - upY + c * mbY;
#pragma unroll
                for (int i = 0; i < 4; i++) aY[i] = load_l2(pa + 4 * i);
                arY = load_l2(pa + mbY);
                const unsigned char *pu = (const unsigned char *)dU - upC + c * mbC, *pv = (const unsigned char *)dV - upC + c * mbC;
                aU[0] = load_l2(pu); aU[1] = load_l2(pu + 4);
                aV[0] = load_l2(pv); aV[1] = load_l2(pv + 4);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) aY[i] = nY[i];
                arY = nAR;
                aU[0] = nU[0]; aU[1] = nU[1]; aV[0] = nV[0]; aV[1] = nV[1];
            }
            // vp8_extend_mb_row (extend.c:160-185): right of the frame the line repeats its last pixel
            if (!top && c == cols - 1) arY = splat(aY[3] >> 24);
            const int tlY = top ? 127 : (c == 0 ? 129 : prevLastY);
            const int tlU = top ? 127 : (c == 0 ? 129 : prevLastU);
            const int tlV = top ? 127 : (c == 0 ? 129 : prevLastV);
            const int up = !top, lf = c > 0;

            // ---- inter MBs: reference plane, MV clamp window
            g_cu8p rf = nullptr;
            if (!intra) rf = (g_cu8p)job->ref[ref_frame & 3];
            const bool clampmv = flags & VP8IR_MB_CLAMP;
            const int e_left = -((c * 16) << 3), e_right = ((cols - 1 - c) * 16) << 3;
            const int e_top = -((r * 16) << 3), e_bottom = ((rows - 1 - r) * 16) << 3;
            const bool any_inter = __builtin_amdgcn_ballot_w64(!intra) != 0;

            // ---- Y2: vp8_dequantize_b + vp8_short_inv_walsh4x4_c (idctllm.c:140-192) -> the 16 luma DCs
            int dc[16];
#pragma unroll
            for (int i = 0; i < 16; i++) dc[i] = 0;
            if (has_y2 && !skip) {
                const u32x4 ca = pf_y2a, cb = pf_y2b;
                const u32 q[8] = { ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w };
                const int fdc = dq1 & 0xffff, fac = dq1 >> 16;
                int tt[16];
#pragma unroll
                for (int col = 0; col < 4; col++) {
                    const int i0 = (short)(sext16(q[2 * col]) * (col == 0 ? fdc : fac));
                    const int i1 = (short)(hi16(q[2 * col]) * fac);
                    const int i2 = (short)(sext16(q[2 * col + 1]) * fac);
                    const int i3 = (short)(hi16(q[2 * col + 1]) * fac);
                    const int a1 = i0 + i3, b1 = i1 + i2, c1 = i1 - i2, d1 = i0 - i3;
                    tt[0 + col] = (short)(a1 + b1); tt[4 + col] = (short)(c1 + d1);
                    tt[8 + col] = (short)(a1 - b1); tt[12 + col] = (short)(d1 - c1);
                }
#pragma unroll
                for (int row = 0; row < 4; row++) {
                    const int a1 = tt[row * 4] + tt[row * 4 + 3], b1 = tt[row * 4 + 1] + tt[row * 4 + 2];
                    const int c1 = tt[row * 4 + 1] - tt[row * 4 + 2], d1 = tt[row * 4] - tt[row * 4 + 3];
                    dc[row * 4 + 0] = (short)((a1 + b1 + 3) >> 3); dc[row * 4 + 1] = (short)((c1 + d1 + 3) >> 3);
                    dc[row * 4 + 2] = (short)((a1 - b1 + 3) >> 3); dc[row * 4 + 3] = (short)((d1 - c1 + 3) >> 3);
                }
            }

            // ======================= luma: four groups of four 4x4 blocks =======================
            int dcY = 128;
            if (up | lf) {
                const int shift = 3 + up + lf;
                const int s = (up ? sad4(aY[0]) + sad4(aY[1]) + sad4(aY[2]) + sad4(aY[3]) : 0)
                            + (lf ? sad4(lY[0]) + sad4(lY[1]) + sad4(lY[2]) + sad4(lY[3]) : 0);
                dcY = (s + (1 << (shift - 1))) >> shift;
            }
            u32 abv[4] = { aY[0], aY[1], aY[2], aY[3] };     // line above the current block row (B_PRED chain)
            int tlrow = tlY;                                  // top-left of the block row's first block
            u32 nl[4] = { 0, 0, 0, 0 };                       // right column of this MB = left of the next
            g_u8p prow = dY + c * mbY;
#pragma unroll 1
            for (int by = 0; by < 4; by++) {
                const u32 lcur = lY[0];
                if (!skip) {                                  // the group after this one (by = 3: the U blocks)
#pragma unroll
                    for (int i = 0; i < 8; i++) nxt[i] = *(g_cu32x4p)(cfp + (by + 1) * 64 + i * 8);
                }
                if (any_inter) {
                    if (!intra) {
                        const u32x4 mv4 = *(g_cu32x4p)(mvp + by * 4);
#pragma unroll 1
                        for (int b = 0; b < 4; b++) {
                            const u32 mvw = b == 0 ? mv4.x : b == 1 ? mv4.y : b == 2 ? mv4.z : mv4.w;
                            int mrow = sext16(mvw), mcol = hi16(mvw);
                            if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
                            u32 o[4];
                            inter4x4(rf + g.y_off, g.y_stride, c * 16 + b * 4, r * 16 + by * 4, mrow, mcol, bilinear,
                                     g.aligned_w, g.aligned_h, 32, o);
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) predlds[(b * 4 + jj) * 64 + lane] = o[jj];
                        }
                    }
                }
                const u32 bmw = by == 0 ? bm.x : by == 1 ? bm.y : by == 2 ? bm.z : bm.w;
                const u32 eobw = by == 0 ? eobY[0] : by == 1 ? eobY[1] : by == 2 ? eobY[2] : eobY[3];
                u32 left = lcur;
                int tl = tlrow;
                u32 orow[4][4];                               // [row][block]: 16-byte rows for the write-out
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 p[4];
                    if (bpred) {
                        // decodframe.c:200-236; above-right of the right-hand block column is the MB's own
                        // above-right for every block row (reconintra4x4.c:305-317)
                        bpred4x4((bmw >> (8 * k)) & 0xff, abv[k], k < 3 ? abv[k + 1] : arY, left, tl, p);
                    } else if (intra) {
                        mb_mode_pred(y_mode, aY[k], lcur, tlY, dcY, p);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) p[jj] = predlds[(k * 4 + jj) * 64 + lane];
                    }
                    u32 o[4] = { p[0], p[1], p[2], p[3] };
                    // idct_blk.c:28-37: eob > 1 -> the full transform, else DC only.  The branch is taken per wave: the
                    // DC-only path is only worth having when no lane needs the transform.
                    const bool full = !skip && ((eobw >> (8 * k)) & 0xff) > 1;
                    if (__builtin_amdgcn_ballot_w64(full) != 0) {
                        if (!skip) {
                            int res[16];
                            dequant_idct(cur[2 * k], cur[2 * k + 1], dq0 & 0xffff, dq0 >> 16, has_y2, dc[k], res);
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) o[jj] = add_clamp_pack(p[jj], res + 4 * jj);
                        }
                    } else if (!skip) {
                        const int d0 = has_y2 ? dc[k] : (short)(sext16(cur[2 * k].x) * (int)(dq0 & 0xffff));
                        const int d = (d0 + 4) >> 3;
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) o[jj] = add_dc_clamp_pack(p[jj], pk(d, d));
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) orow[jj][k] = o[jj];
                    tl = abv[k] >> 24;
                    abv[k] = o[3];
                    left = right_column(o);
                }
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    *(g_u32x4p)(prow + jj * ysY) = (u32x4){ orow[jj][0], orow[jj][1], orow[jj][2], orow[jj][3] };
                // rotate the per-row shift registers
                tlrow = lcur >> 24;
                lY[0] = lY[1]; lY[1] = lY[2]; lY[2] = lY[3];
                nl[0] = nl[1]; nl[1] = nl[2]; nl[2] = nl[3]; nl[3] = left;
#pragma unroll
                for (int i = 0; i < 12; i++) dc[i] = dc[i + 4];
                prow += 4 * ysY;
#pragma unroll
                for (int i = 0; i < 8; i++) cur[i] = nxt[i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { bY[i] = abv[i]; lY[i] = nl[i]; }

            // ======================= chroma: U then V, four 4x4 blocks each =======================
            u32 cmv[4] = { 0, 0, 0, 0 };         // chroma MVs of the four 4x4 chroma blocks (row | col << 16)
            if (any_inter) {
                if (!intra) {
                    if (y_mode != VP8IR_SPLITMV) {   // reconinter.c:419-424: from the CLAMPED luma MV
                        const u32 mvw = mvp[0];
                        int mrow = sext16(mvw), mcol = hi16(mvw);
                        if (clampmv) clamp_luma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
                        mrow = (short)(mrow + (1 | (mrow >> 31)));
                        mcol = (short)(mcol + (1 | (mcol >> 31)));
                        mrow /= 2; mcol /= 2;
                        if (fullpix) { mrow &= ~7; mcol &= ~7; }
                        cmv[0] = cmv[1] = cmv[2] = cmv[3] = ((u32)mrow & 0xffff) | ((u32)mcol << 16);
                    } else {                          // build_4x4uvmvs (reconinter.c:520-558): UNclamped MVs
                        const u32x4 m0 = *(g_cu32x4p)(mvp), m1 = *(g_cu32x4p)(mvp + 4);
                        const u32x4 m2 = *(g_cu32x4p)(mvp + 8), m3 = *(g_cu32x4p)(mvp + 12);
                        const u32 quad[4][4] = { { m0.x, m0.y, m1.x, m1.y }, { m0.z, m0.w, m1.z, m1.w },
                                                 { m2.x, m2.y, m3.x, m3.y }, { m2.z, m2.w, m3.z, m3.w } };
#pragma unroll
                        for (int kq = 0; kq < 4; kq++) {
                            int mrow = sext16(quad[kq][0]) + sext16(quad[kq][1]) + sext16(quad[kq][2]) + sext16(quad[kq][3]);
                            int mcol = hi16(quad[kq][0]) + hi16(quad[kq][1]) + hi16(quad[kq][2]) + hi16(quad[kq][3]);
                            mrow += 4 + ((mrow >> 31) << 3);
                            mcol += 4 + ((mcol >> 31) << 3);
                            mrow /= 8; mcol /= 8;
                            if (fullpix) { mrow &= ~7; mcol &= ~7; }
                            if (clampmv) clamp_chroma_mv(mrow, mcol, e_left, e_right, e_top, e_bottom);
                            cmv[kq] = ((u32)mrow & 0xffff) | ((u32)mcol << 16);
                        }
                    }
                }
            }
            const bool more = c + 1 < cols;      // the row goes on: prefetch its next macroblock
#pragma unroll 1
            for (int pl = 0; pl < 2; pl++) {
                const u32 aC0 = pl ? aV[0] : aU[0], aC1 = pl ? aV[1] : aU[1];
                const u32 lC0 = pl ? lV[0] : lU[0], lC1 = pl ? lV[1] : lU[1];
                const int tlC = pl ? tlV : tlU;
                g_u8p dC = (pl ? dV : dU) + c * mbC;
                int dcC = 128;
                if (up | lf) {
                    const int shift = 2 + up + lf;
                    const int s = (up ? sad4(aC0) + sad4(aC1) : 0) + (lf ? sad4(lC0) + sad4(lC1) : 0);
                    dcC = (s + (1 << (shift - 1))) >> shift;
                }
                if (pl == 0) {
                    if (!skip) {                              // the V blocks
#pragma unroll
                        for (int i = 0; i < 8; i++) nxt[i] = *(g_cu32x4p)(cfp + 320 + i * 8);
                    }
                } else if (more) {                            // the next macroblock of the row
                    pf_m0 = *(g_cu32x4p)(mbp + 16); pf_m1 = *(g_cu32x4p)(mbp + 20); pf_bm = *(g_cu32x4p)(mbp + 26);
                    pf_y2a = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + 384); pf_y2b = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + 392);
#pragma unroll
                    for (int i = 0; i < 8; i++) nxt[i] = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + i * 8);
                }
                if (any_inter) {
                    if (!intra) {
#pragma unroll 1
                        for (int b = 0; b < 4; b++) {
                            const u32 mvw = b == 0 ? cmv[0] : b == 1 ? cmv[1] : b == 2 ? cmv[2] : cmv[3];
                            u32 o[4];
                            inter4x4(rf + (pl ? g.v_off : g.u_off), g.uv_stride, c * 8 + (b & 1) * 4, r * 8 + (b >> 1) * 4,
                                     sext16(mvw), hi16(mvw), bilinear, g.aligned_w / 2, g.aligned_h / 2, 16, o);
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) predlds[(b * 4 + jj) * 64 + lane] = o[jj];
                        }
                    }
                }
                u32 bot[2] = { 0, 0 }, rc[2] = { 0, 0 };
                u32 orow[8][2];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int bx = k & 1, byc = k >> 1;
                    u32 p[4];
                    if (intra) mb_mode_pred(uv_mode, bx ? aC1 : aC0, byc ? lC1 : lC0, tlC, dcC, p);
                    else {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) p[jj] = predlds[(k * 4 + jj) * 64 + lane];
                    }
                    u32 o[4] = { p[0], p[1], p[2], p[3] };
                    const bool full = !skip && (((pl ? eobV : eobU) >> (8 * k)) & 0xff) > 1;
                    if (__builtin_amdgcn_ballot_w64(full) != 0) {
                        if (!skip) {
                            int res[16];
                            dequant_idct(cur[2 * k], cur[2 * k + 1], dq2 & 0xffff, dq2 >> 16, false, 0, res);
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) o[jj] = add_clamp_pack(p[jj], res + 4 * jj);
                        }
                    } else if (!skip) {
                        const int d = ((short)(sext16(cur[2 * k].x) * (int)(dq2 & 0xffff)) + 4) >> 3;
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) o[jj] = add_dc_clamp_pack(p[jj], pk(d, d));
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) orow[byc * 4 + jj][bx] = o[jj];
                    if (byc) bot[bx] = o[3];
                    if (bx) rc[byc] = right_column(o);
                }
#pragma unroll
                for (int jj = 0; jj < 8; jj++) *(g_u32x2p)(dC + jj * ysC) = (u32x2){ orow[jj][0], orow[jj][1] };
                if (pl) { bV[0] = bot[0]; bV[1] = bot[1]; lV[0] = rc[0]; lV[1] = rc[1]; }
                else { bU[0] = bot[0]; bU[1] = bot[1]; lU[0] = rc[0]; lU[1] = rc[1]; }
#pragma unroll
                for (int i = 0; i < 8; i++) cur[i] = nxt[i];
            }

            prevLastY = aY[3] >> 24; prevLastU = aU[1] >> 24; prevLastV = aV[1] >> 24;
            mbp += 16; cfp += VP8IR_COEF_PER_MB; mvp += 16;
        }
        // ---- history: what the lane below will ask for in one and in two steps
#pragma unroll
        for (int i = 0; i < 4; i++) { h2Y[i] = h1Y[i]; h1Y[i] = bY[i]; }
#pragma unroll
        for (int i = 0; i < 2; i++) { h2U[i] = h1U[i]; h1U[i] = bU[i]; h2V[i] = h1V[i]; h1V[i] = bV[i]; }
    }
}
