// Block-level arithmetic of the VP8 pixel path, shared by the wave-per-row frame kernels (vp8_recon.hip,
// vp8_loopfilter.hip) and by the per-block RTCD entry points (vp8_rtcd_blocks.hip): inverse DCT passes,
// whole-block and 4x4 intra predictors, six-tap building blocks, and the loop-filter edge operators.
// Every function cites the reference lines whose arithmetic it reproduces.
#pragma once
#include "vp8_common.hip.h"

// ---- 4x4 intra predictor table (derived from vp8/common/reconintra4x4.c:16-303, same as the
// oracle's): edge vector P[0..14] = {L3,L3,L2,L1,L0,TL,A0..A7,A7}; entry = kind<<4 | k with kind
// 0: P[k], 1: (P[k]+P[k+1]+1)>>1, 2: (P[k-1]+2P[k]+P[k+1]+2)>>2.  Rows = modes; B_DC / B_TM (rows
// 0,1) are computed directly.
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

typedef unsigned int u32;

__device__ __forceinline__ u32 dpp_xor1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ u32 dpp_xor2(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ int sad4(u32 v) { return (int)__builtin_amdgcn_sad_u8(v, 0u, 0u); }
__device__ __forceinline__ int sext16(u32 v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(u32 v) { return (int)v >> 16; }

// Column (vertical) pass of vp8_short_idct4x4llm_c (idctllm.c:39-60); results are truncated to
// i16 by the packing in quad_transpose16.
__device__ __forceinline__ void idct_col(int i0, int i1, int i2, int i3, int o[4])
{
    int a1 = i0 + i2, b1 = i0 - i2;
    int c1 = ((i1 * 35468) >> 16) - (i3 + ((i3 * 20091) >> 16));
    int d1 = (i1 + ((i1 * 20091) >> 16)) + ((i3 * 35468) >> 16);
    o[0] = a1 + d1; o[3] = a1 - d1; o[1] = b1 + c1; o[2] = b1 - c1;
}
// Row (horizontal) pass with the (x+4)>>3 rounding (idctllm.c:65-88); outputs are i16 values.
__device__ __forceinline__ void idct_row(const int t[4], int o[4])
{
    int a1 = t[0] + t[2], b1 = t[0] - t[2];
    int c1 = ((t[1] * 35468) >> 16) - (t[3] + ((t[3] * 20091) >> 16));
    int d1 = (t[1] + ((t[1] * 20091) >> 16)) + ((t[3] * 35468) >> 16);
    o[0] = (short)((a1 + d1 + 4) >> 3);
    o[3] = (short)((a1 - d1 + 4) >> 3);
    o[1] = (short)((b1 + c1 + 4) >> 3);
    o[2] = (short)((b1 - c1 + 4) >> 3);
}

__device__ __forceinline__ u32 add_clamp_pack(u32 pred, const int r[4])
{
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) out |= (u32)clamp255((int)((pred >> (8 * i)) & 0xff) + r[i]) << (8 * i);
    return out;
}

// Whole-block intra predictors (reconintra.c:139-241, 403-521) for a 4-pixel row segment:
// above = the 4 pixels above the segment's columns, left = pixel left of the segment's row.
__device__ __forceinline__ u32 intra_pred4(int mode, u32 above, int left, int tl, int dc)
{
    if (mode == VP8IR_DC_PRED) return (u32)dc * 0x01010101u;
    if (mode == VP8IR_V_PRED) return above;
    if (mode == VP8IR_H_PRED) return (u32)left * 0x01010101u;
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) out |= (u32)clamp255(left + (int)((above >> (8 * i)) & 0xff) - tl) << (8 * i);
    return out;
}

// ---- six-tap building blocks (filter.c:41-128), two pixels per instruction on 16-bit lanes --------------
// A first-pass sum lies in -8160 .. 40864, so biased by 8192 it is an unsigned 16-bit number and wrap-around
// arithmetic (v_pk_mad_u16, negative taps as their two's complement) is exact; (t + 8192) >> 7 == (t >> 7) + 64, and
// a saturating subtraction of 64 plus a minimum with 255 are the clamp.  The second pass has the same range.
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
struct SixTaps { v2u16 t[6]; };
__device__ __forceinline__ SixTaps sixtap_taps(int f)
{
    SixTaps r;
#pragma unroll
    for (int k = 0; k < 6; k++) { const unsigned short t = (unsigned short)k_sixtap[f][k]; r.t[k] = (v2u16){ t, t }; }
    return r;
}
__device__ __forceinline__ u32 sixtap_finish(v2u16 a01, v2u16 a23)     // two biased sums of two pixels -> four clamped bytes
{
    const v2u16 c64 = { 64, 64 }, c255 = { 255, 255 };
    const v2u16 r01 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(a01 >> 7, c64), c255);
    const v2u16 r23 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(a23 >> 7, c64), c255);
    return __builtin_amdgcn_perm(__builtin_bit_cast(u32, r23), __builtin_bit_cast(u32, r01), 0x06040200u);
}
// first pass for four output pixels: s points at the pixel two left of the first one (nine pixels are read, as three
// aligned dwords shifted into place)
__device__ __forceinline__ u32 sixtap_hrow(g_cu8p s, const SixTaps &tx)
{
    auto asv = [](u32 v) { return __builtin_bit_cast(v2u16, v); };
    auto perm = [](u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); };
    const u32 sh = (u32)(unsigned long)s & 3u;
    g_cu32p rp = (g_cu32p)(s - sh);
    const u32 d0 = rp[0], d1 = rp[1], d2 = rp[2];
    const u32 w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const u32 w2 = __builtin_amdgcn_alignbyte(0u, d2, sh);
    // P[k] = pixels (k, k+1) of the row, one per 16-bit lane
    const v2u16 P[8] = { asv(perm(w0, w0, 0x0c010c00u)), asv(perm(w0, w0, 0x0c020c01u)), asv(perm(w0, w0, 0x0c030c02u)),
                         asv(perm(w1, w0, 0x0c040c03u)), asv(perm(w1, w1, 0x0c010c00u)), asv(perm(w1, w1, 0x0c020c01u)),
                         asv(perm(w1, w1, 0x0c030c02u)), asv(perm(w2, w1, 0x0c040c03u)) };
    const v2u16 bias = { 64 + 8192, 64 + 8192 };
    v2u16 a01 = bias, a23 = bias;
#pragma unroll
    for (int k = 0; k < 6; k++) { a01 += P[k] * tx.t[k]; a23 += P[k + 2] * tx.t[k]; }
    return sixtap_finish(a01, a23);
}
// second pass: H[k] = four first-pass pixels (bytes) of source row k - 2
__device__ __forceinline__ u32 sixtap_vcol(const u32 H[6], const SixTaps &ty)
{
    const v2u16 bias = { 64 + 8192, 64 + 8192 };
    v2u16 a01 = bias, a23 = bias;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        a01 += __builtin_bit_cast(v2u16, __builtin_amdgcn_perm(H[k], H[k], 0x0c010c00u)) * ty.t[k];
        a23 += __builtin_bit_cast(v2u16, __builtin_amdgcn_perm(H[k], H[k], 0x0c030c02u)) * ty.t[k];
    }
    return sixtap_finish(a01, a23);
}

// ---- loop-filter edge operators (vp8/common/loopfilter_filters.c) ----
__device__ __forceinline__ int sc8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// vp8_filter_mask (loopfilter_filters.c:27-40): true = filter this position
__device__ __forceinline__ bool lf_mask(int limit, int blimit, const int p[8])
{
    bool m = iabs(p[0] - p[1]) > limit;
    m |= iabs(p[1] - p[2]) > limit;
    m |= iabs(p[2] - p[3]) > limit;
    m |= iabs(p[5] - p[4]) > limit;
    m |= iabs(p[6] - p[5]) > limit;
    m |= iabs(p[7] - p[6]) > limit;
    m |= iabs(p[3] - p[4]) * 2 + iabs(p[2] - p[5]) / 2 > blimit;
    return !m;
}
// vp8_hevmask (:43-49)
__device__ __forceinline__ bool lf_hev(int thr, const int p[8])
{
    return iabs(p[2] - p[3]) > thr || iabs(p[5] - p[4]) > thr;
}
// vp8_filter (:51-95): p[2..5] = p1 p0 q0 q1
__device__ __forceinline__ void lf_inner(int p[8], bool mask, bool hev)
{
    int ps1 = p[2] - 128, ps0 = p[3] - 128, qs0 = p[4] - 128, qs1 = p[5] - 128;
    int f = sc8(ps1 - qs1);
    f = hev ? f : 0;
    f = sc8(f + 3 * (qs0 - ps0));
    f = mask ? f : 0;
    int f1 = sc8(f + 4) >> 3, f2 = sc8(f + 3) >> 3;
    p[4] = sc8(qs0 - f1) + 128;
    p[3] = sc8(ps0 + f2) + 128;
    f = (f1 + 1) >> 1;
    f = hev ? 0 : f;
    p[5] = sc8(qs1 - f) + 128;
    p[2] = sc8(ps1 + f) + 128;
}
// vp8_mbfilter (:161-214): p[1..6] = p2 p1 p0 q0 q1 q2
__device__ __forceinline__ void lf_mbedge(int p[8], bool mask, bool hev)
{
    int ps2 = p[1] - 128, ps1 = p[2] - 128, ps0 = p[3] - 128;
    int qs0 = p[4] - 128, qs1 = p[5] - 128, qs2 = p[6] - 128;
    int f = sc8(ps1 - qs1);
    f = sc8(f + 3 * (qs0 - ps0));
    f = mask ? f : 0;
    int f2 = hev ? f : 0;
    int f1 = sc8(f2 + 4) >> 3;
    f2 = sc8(f2 + 3) >> 3;
    qs0 = sc8(qs0 - f1);
    ps0 = sc8(ps0 + f2);
    f = hev ? 0 : f;
    int u = sc8((63 + f * 27) >> 7);
    p[4] = sc8(qs0 - u) + 128;
    p[3] = sc8(ps0 + u) + 128;
    u = sc8((63 + f * 18) >> 7);
    p[5] = sc8(qs1 - u) + 128;
    p[2] = sc8(ps1 + u) + 128;
    u = sc8((63 + f * 9) >> 7);
    p[6] = sc8(qs2 - u) + 128;
    p[1] = sc8(ps2 + u) + 128;
}
// vp8_simple_filter_mask + vp8_simple_filter (:292-315): p[2..5] = p1 p0 q0 q1
__device__ __forceinline__ void lf_simple(int p[8], int blimit)
{
    bool mask = iabs(p[3] - p[4]) * 2 + iabs(p[2] - p[5]) / 2 <= blimit;
    int p1 = p[2] - 128, p0 = p[3] - 128, q0 = p[4] - 128, q1 = p[5] - 128;
    int f = sc8(p1 - q1);
    f = sc8(f + 3 * (q0 - p0));
    f = mask ? f : 0;
    int f1 = sc8(f + 4) >> 3;
    p[4] = sc8(q0 - f1) + 128;
    int f2 = sc8(f + 3) >> 3;
    p[3] = sc8(p0 + f2) + 128;
}

struct LfParams { int mblim, blim, lim, hev_thr; };

__device__ __forceinline__ void filter_edge(int *q0, int kind, const LfParams &lp, int edge_limit)
{
    int p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = q0[i - 4];
    if (kind == 2) {
        lf_simple(p, edge_limit);
        q0[-1] = p[3]; q0[0] = p[4];
        return;
    }
    const bool m = lf_mask(lp.lim, edge_limit, p), hv = lf_hev(lp.hev_thr, p);
    if (kind == 1) {
        lf_mbedge(p, m, hv);
        q0[-3] = p[1]; q0[2] = p[6];
    } else
        lf_inner(p, m, hv);
    q0[-2] = p[2]; q0[-1] = p[3]; q0[0] = p[4]; q0[1] = p[5];
}

