// Per-lane building blocks of the lane-per-row kernels (vp8_recon_simt.hip, vp8_loopfilter_simt.hip, vp8_keyframe_simt.hip):
// the intra predictors and the inverse transform on packed bytes / packed 16-bit lanes, and the loop-filter arithmetic on
// signed 8.8 pairs.  Everything here is per-lane code -- one lane, one macroblock -- and shared by value: each kernel file
// gets its own copy in an anonymous namespace.
#pragma once
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

typedef __attribute__((address_space(3))) void *lds_vp;
typedef GLOBAL_AS const void *g_cvp;
typedef short v2s __attribute__((ext_vector_type(2)));      // two 16-bit lanes: v_pk_* arithmetic
__device__ __forceinline__ v2s as_v2s(u32 v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ u32 as_u32(v2s v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ v2s pk(int lo, int hi) { return (v2s){ (short)lo, (short)hi }; }
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ u32 alignb(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ u32 lerp(u32 a, u32 b, u32 c) { return __builtin_amdgcn_lerp(a, b, c); }
__device__ __forceinline__ int sad4(u32 v) { return (int)__builtin_amdgcn_sad_u8(v, 0u, 0u); }
__device__ __forceinline__ u32 splat(int v) { return (u32)v * 0x01010101u; }
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

// two signed 16-bit values -> two bytes clamped to 0..255, in bits 15:0 (bits 31:16 zero): the saturation of this
// instruction IS the clamp of vp8_dequant_idct_add_c / vp8_dc_only_idct_add_c / the TM predictor
__device__ __forceinline__ u32 sat_pk_u8(v2s v)
{
    u32 d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(as_u32(v)));
    return d;
}
// four pixels = clamp(a + b) for two pairs of 16-bit lanes: (x0, x1) and (x2, x3)
__device__ __forceinline__ u32 clamp_pack4(v2s s01, v2s s23)
{
    return perm(sat_pk_u8(s23), sat_pk_u8(s01), 0x05040100u);
}

// clamp(pred + residual) for a row of four pixels (the tail of vp8_short_idct4x4llm / vp8_dc_only_idct_add): predictor
// bytes widened to 16-bit lanes, residuals as vp8_residual_kernel left them: (r0, r1), (r2, r3)
__device__ __forceinline__ u32 add_clamp_pack(u32 pred, u32 r01, u32 r23)
{
    return clamp_pack4(as_v2s(perm(pred, pred, 0x0c010c00u)) + as_v2s(r01), as_v2s(perm(pred, pred, 0x0c030c02u)) + as_v2s(r23));
}

// ... for the four rows of a block at once, stage by stage: the rows are independent, and gfx950 wants a wait state between a
// packed operation and its consumer -- row after row the compiler fills it with s_nop
__device__ __forceinline__ void add_clamp_rows(const u32 (&p)[4], const u32x4 ra, const u32x4 rb, u32 (&o)[4])
{
    const u32 r[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
    u32 w[8], t[8];
#pragma unroll
    for (int j = 0; j < 4; j++) { w[2 * j] = perm(p[j], p[j], 0x0c010c00u); w[2 * j + 1] = perm(p[j], p[j], 0x0c030c02u); }
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = as_u32(as_v2s(w[i]) + as_v2s(r[i]));
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = sat_pk_u8(as_v2s(w[i]));
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = perm(t[2 * j + 1], t[2 * j], 0x05040100u);
}

// TM prediction of a row of four pixels: clamp(above[i] + left - top_left), above given as two packed pairs
__device__ __forceinline__ u32 tm_row(v2s a01, v2s a23, int l_minus_tl)
{
    const v2s d = pk(l_minus_tl, l_minus_tl);
    return clamp_pack4(a01 + d, a23 + d);
}

// right-hand pixel column of a 4x4 block given as four row dwords -> one dword, top pixel in byte 0
__device__ __forceinline__ u32 right_column(const u32 o[4])
{
    return perm(perm(o[3], o[2], 0x0c0c0703u), perm(o[1], o[0], 0x0c0c0703u), 0x05040100u);
}

// Whole-block predictors DC / V / H / TM (reconintra.c:139-241, 403-521) for one 4x4 block of a chroma plane (vp8_keyframe_simt.hip):
// above = the 4 pixels above the block's columns, left = the 4 pixels left of its rows (top in byte 0), dcs = the DC value on all
// four bytes.  No branch per mode (see pred4x4_net below for why; through round 4 this was an if-chain): selects, and
// TM behind one wave-uniform test.
__device__ __forceinline__ void mb_mode_pred_sel(int mode, u32 above, u32 left, int tl, u32 dcs, u32 p[4])
{
    const bool v = mode == VP8IR_V_PRED, h = mode == VP8IR_H_PRED, tm = mode == VP8IR_TM_PRED;
    const u32 c = v ? above : dcs;
    p[0] = h ? perm(left, left, 0x00000000u) : c; p[1] = h ? perm(left, left, 0x01010101u) : c;
    p[2] = h ? perm(left, left, 0x02020202u) : c; p[3] = h ? perm(left, left, 0x03030303u) : c;
    if (__builtin_amdgcn_ballot_w64(tm) != 0) {
        const v2s a01 = as_v2s(perm(above, above, 0x0c010c00u)), a23 = as_v2s(perm(above, above, 0x0c030c02u));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 t = tm_row(a01, a23, (int)((left >> (8 * j)) & 0xff) - tl);
            p[j] = tm ? t : p[j];
        }
    }
}

// vp8_intra4x4_predict (reconintra4x4.c:16-303) for one block.  a0 = above 4 pixels, a1 = the next 4
// (above-right), left = left 4 pixels (top in byte 0), tl = top-left.  Edge vector as in the oracle:
// P[0..15] = { L3, L3, L2, L1, L0, TL, A0..A7, A7, A7 }; F[k] = (P[k-1]+2P[k]+P[k+1]+2)>>2,
// G[k] = (P[k]+P[k+1]+1)>>1, both computed four pixels per instruction with v_lerp_u8:
// (a+2b+c+2)>>2 == (((a+c)>>1) + b + 1)>>1 exactly.  Which F / G a pixel of a mode is: tools/gen_pred_sel.py.
// Through round 4 this was a switch over the ten modes.  A wave's lanes hold macroblocks of every kind, so the switch ran ALL
// its cases, each behind an exec-mask save / branch / restore: 145 of the 339 instructions a block's
// prediction + add took were scalar bookkeeping, and to a wave that is alone in keeping its SIMD's issue port (the luma wave:
// the kernel's time is its time) a scalar instruction costs what a vector one does.  Here a predicted row is three v_perm_b32
// over the lane's POOL of filtered edge values, OR-ed, the selectors looked up by mode in an LDS table (tools/gen_pred_sel.py
// has the layout and writes vp8_pred_sel.inc; k_pred_sel is copied to LDS at kernel entry).  Entry 0 serves B_DC_PRED and the
// 16x16 modes DC_PRED / V_PRED (the row is the dword C), entry 10 the 16x16 H_PRED (rows = bytes of `hcol`); TM is arithmetic
// and stays with the caller.
enum { PSEL_MB_H = 10, PSEL_MODES = 11, PSEL_WORDS = 12 };
__device__ const u32 k_pred_sel[PSEL_MODES * PSEL_WORDS] = {
#include "vp8_pred_sel.inc"
};
// sel: the mode's entry in LDS (three 16-byte reads); use_c: the mode is 0 (take C for F[0..3]); mb: a 16x16 mode (take hcol for G[0..3])
__device__ __forceinline__ void pred4x4_net(const u32x4 *sel, u32 a0, u32 a1, u32 left, int tl, bool use_c, u32 C, bool mb, u32 hcol, u32 p[4])
{
    const u32x4 sa = sel[0], sb = sel[1], sc = sel[2];
    const u32 E0 = perm(left, left, 0x01020303u);                       // L3 L3 L2 L1
    const u32 E1 = perm(a0, left, 0x05040c00u) | ((u32)tl << 8);        // L0 TL A0 A1
    const u32 E2 = alignb(a1, a0, 2);                                   // A2 A3 A4 A5
    const u32 E3 = perm(a1, a1, 0x03030302u);                           // A6 A7 A7 A7
    const u32 N0 = alignb(E1, E0, 1), N1 = alignb(E2, E1, 1), N2 = alignb(E3, E2, 1), N3 = E3 >> 8;
    const u32 M0 = E0 << 8, M1 = alignb(E1, E0, 3), M2 = alignb(E2, E1, 3), M3 = alignb(E3, E2, 3);
    const u32 one = 0x01010101u;
    const u32 F0 = lerp(lerp(M0, N0, 0), E0, one), F1 = lerp(lerp(M1, N1, 0), E1, one);
    const u32 F2 = lerp(lerp(M2, N2, 0), E2, one), F3 = lerp(lerp(M3, N3, 0), E3, one);
    const u32 G0 = lerp(E0, N0, one), G1 = lerp(E1, N1, one), G2 = lerp(E2, N2, one);
    const u32 lo0 = use_c ? C : F0, hi1 = perm(G2, F3, 0x05040100u), lo2 = mb ? hcol : G0;      // hi1 = F12 F13 G8 G9
    p[0] = perm(F1, lo0, sa.x) | perm(hi1, F2, sa.y) | perm(G1, lo2, sa.z);
    p[1] = perm(F1, lo0, sa.w) | perm(hi1, F2, sb.x) | perm(G1, lo2, sb.y);
    p[2] = perm(F1, lo0, sb.z) | perm(hi1, F2, sb.w) | perm(G1, lo2, sc.x);
    p[3] = perm(F1, lo0, sc.y) | perm(hi1, F2, sc.z) | perm(G1, lo2, sc.w);
}

__device__ __forceinline__ int sext16(u32 v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(u32 v) { return (int)v >> 16; }

// one 1-D pass of vp8_short_idct4x4llm_c (idctllm.c:39-60 / 65-88) without the final rounding.  The odd inputs come SHIFTED:
// x1 = i1 << 16, x3 = i3 << 16 (a 16-bit value in the upper half of its dword -- where a packed pair holds its second element
// anyway).  Then (i * 35468) >> 16 is the high half of the 64-bit product x * 35468, ONE v_mul_hi_i32 (floor division by 2^16, as
// the arithmetic shift is), and i + ((i * 20091) >> 16) == (i * (65536 + 20091)) >> 16 is one more: four instructions for the two
// rotated terms where multiply, shift, multiply, shift, add took ten (round 6: the transform is a fifth of the luma wave's time on
// dense inter frames).
__device__ __forceinline__ void idct1d(int i0, int x1, int i2, int x3, int &o0, int &o1, int &o2, int &o3)
{
    const int a1 = i0 + i2, b1 = i0 - i2;
    const int c1 = __mulhi(x1, 35468) - __mulhi(x3, 65536 + 20091);
    const int d1 = __mulhi(x1, 65536 + 20091) + __mulhi(x3, 35468);
    o0 = a1 + d1; o3 = a1 - d1; o1 = b1 + c1; o2 = b1 - c1;
}

// The residual of vp8_dequant_idct_add_c (dequantize.c:29-44) for one block held by one thread.
// ca, cb: the block's 16 coefficients as stored (IR order: column-major, two per dword); dc_in: the already
// dequantised DC when the macroblock has a Y2 block (dequant factor 1, decodframe.c:92).  res[row*4+col].
__device__ __forceinline__ void dequant_idct(const u32x4 ca, const u32x4 cb, int dqdc, int dqac, bool dc_given, int dc_in, int res[16])
{
    const u32 q[8] = { ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w };
    int t[16];                                   // t[row*4+col], i16 like the reference's `short output[16]`
#pragma unroll
    for (int col = 0; col < 4; col++) {
        // DQ = (short)(Q * DQC) (dequantize.c:17-27): the low half of a 16x16 product
        const v2s p01 = as_v2s(q[2 * col]) * (col == 0 ? pk(dqdc, dqac) : pk(dqac, dqac));
        const v2s p23 = as_v2s(q[2 * col + 1]) * pk(dqac, dqac);
        int i0 = p01.x;
        if (col == 0 && dc_given) i0 = dc_in;
        const int i2 = p23.x;
        const int x1 = (int)(as_u32(p01) & 0xffff0000u), x3 = (int)(as_u32(p23) & 0xffff0000u);      // (the pairs' upper halves, where they are)
        int o0, o1, o2, o3;
        idct1d(i0, x1, i2, x3, o0, o1, o2, o3);  // vertical pass: column `col`, rows 0..3
        // (`short output[16]`: columns 0, 2 as sign-extended values, columns 1, 3 -- the next pass's odd inputs -- shifted up, which truncates too)
        if (col & 1) { t[0 + col] = (int)((u32)o0 << 16); t[4 + col] = (int)((u32)o1 << 16); t[8 + col] = (int)((u32)o2 << 16); t[12 + col] = (int)((u32)o3 << 16); }
        else { t[0 + col] = (short)o0; t[4 + col] = (short)o1; t[8 + col] = (short)o2; t[12 + col] = (short)o3; }
    }
#pragma unroll
    for (int row = 0; row < 4; row++) {
        int o0, o1, o2, o3;
        // (the rounding 4 of all four outputs rides in on the first input: a1 = i0 + i2 and b1 = i0 - i2 both carry it)
        idct1d(t[row * 4] + 4, t[row * 4 + 1], t[row * 4 + 2], t[row * 4 + 3], o0, o1, o2, o3);
        res[row * 4 + 0] = o0 >> 3; res[row * 4 + 1] = o1 >> 3;
        res[row * 4 + 2] = o2 >> 3; res[row * 4 + 3] = o3 >> 3;
    }
}

// six dequantisation factors of one segment (vp8cx_init_de_quantizer + mb_init_dequantizer, decodframe.c:50-109,
// quant_common.c:39-132): packed as (y1dc | y1ac<<16, y2dc | y2ac<<16, uvdc | uvac<<16)
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


// ---------------- loop filter ----------------


__device__ __forceinline__ unsigned long long load_l2_64(const unsigned char *p)
{
    return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Pixels travel through the filters as SIGNED 8.8 fixed point -- pixel ^ 0x80 (the reference's own bias,
// loopfilter_filters.c:57-60) in the HIGH half of each 16-bit lane -- from the moment they are staged in the LDS tile
// until they are read back for output (one XOR per dword of four pixels each way, not two per value and edge):
//   * the filter arithmetic wants them that way: the 16-bit saturation of `v_pk_add_i16 ... clamp` IS the reference's
//     vp8_signed_char_clamp (every operand is a multiple of 256), so a saturating add costs one instruction instead
//     of add + min + max;
//   * the masks only need |a-b|, which is max-min in any order-preserving representation: signed max / min, and the
//     difference taken modulo 2^16 is the unsigned 8.8 distance; comparisons by unsigned saturating subtraction.
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2u as_v2u(u32 v) { return __builtin_bit_cast(v2u, v); }
__device__ __forceinline__ u32 as_u32(v2u v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ v2u mku(int v) { return (v2u){ (unsigned short)v, (unsigned short)v }; }
__device__ __forceinline__ v2s mks(int v) { return (v2s){ (short)v, (short)v }; }
__device__ __forceinline__ v2u umax(v2u a, v2u b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2u umin(v2u a, v2u b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2u adu(v2u a, v2u b)                                                     // |a - b| of two biased pixels
{
    const v2s x = __builtin_bit_cast(v2s, a), y = __builtin_bit_cast(v2s, b);
    return __builtin_bit_cast(v2u, (v2s)(__builtin_elementwise_max(x, y) - __builtin_elementwise_min(x, y)));
}
__device__ __forceinline__ v2u usubs(v2u a, v2u b) { return __builtin_elementwise_sub_sat(a, b); }    // max(a - b, 0)
__device__ __forceinline__ v2u uadds(v2u a, v2u b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ v2s adds(v2s a, v2s b) { return __builtin_elementwise_add_sat(a, b); }     // signed-char clamp
__device__ __forceinline__ v2s subs(v2s a, v2s b) { return __builtin_elementwise_sub_sat(a, b); }
// (Through round 4: x != 0 ? 0 : 0xffff (nz_clear) and x != 0 ? 0xffff : 0 (nz_set) per half, from max(1 - x, 0) by saturating
// subtraction; masks() now keeps the 1 / 0 form and multiplies.)
// `one` is the constant 1 | 1 << 16 made opaque to LLVM (one empty asm at kernel entry, see Lim::one): with a visible
// constant the expression is canonicalised into a compare-and-select, which gfx950 can only do one half at a time.
__device__ __forceinline__ v2u nz_clear(v2u x, v2u one) { return mku(0) - usubs(one, x); }
__device__ __forceinline__ v2u nz_set(v2u x, v2u one) { return usubs(one, x) - one; }
__device__ __forceinline__ v2s sgn(v2u p) { return as_v2s(as_u32(p)); }                              // (already biased: see above)
__device__ __forceinline__ v2u pix(v2s s) { return as_v2u(as_u32(s)); }
#define VP8_LF_BIAS 0x80808080u     // four pixels <-> four biased pixels, on the way into and out of the LDS tile
__device__ __forceinline__ v2s hib(v2s v) { return as_v2s(as_u32(v) & 0xff00ff00u); }                // floor to a whole byte

struct Lim { v2u mblim, blim, lim, thr, one; };     // the limits, << 8; the opaque constant 1 of nz_clear / nz_set

// The filters are branch-free: `gate` (lf_gate: 0 / 0xffff per lane) switches an edge off by clearing its filter mask,
// which makes every update the identity.  Straight-line code lets the scheduler interleave the independent
// pixel-line pairs, which is what hides the wait state gfx950 wants between dependent packed-math ops.

// |w| of a signed 16-bit pair as an unsigned one (-32768 -> 32768)
__device__ __forceinline__ v2u sabs(v2s w)
{
    return pix(__builtin_elementwise_max(w, sgn(mku(0) - pix(w))));
}

// vp8_filter_mask + vp8_hevmask (loopfilter_filters.c:27-49) for p[0..7] = p3 p2 p1 p0 q0 q1 q2 q3:
// mask = 1 where the edge is filtered (see below), hev = 0xffff where the high-edge-variance rule applies
__device__ __forceinline__ void masks(const v2u p[8], v2u lim, v2u elim, v2u thr, v2u one, v2u gate, v2u &mask, v2u &hev)
{
    const v2u d10 = adu(p[2], p[3]), dq = adu(p[5], p[4]);
    const v2u dh = umax(d10, dq);
    // The four outer differences are only ever compared with the limit, so their sign need not be taken off: with
    // u = clamp(a - b) as a signed 16-bit number, |a - b| <= lim  <=>  (u + lim) mod 2^16 <= 2 lim as UNSIGNED numbers (lim <= 63 << 8:
    // a negative u + lim wraps to >= 32768 + 256, a saturated difference is beyond the limit on either side) -- subtract, add,
    // max instead of max, min, subtract, max.
    const v2u t0 = pix(subs(sgn(p[0]), sgn(p[1]))) + lim, t1 = pix(subs(sgn(p[1]), sgn(p[2]))) + lim;
    const v2u t2 = pix(subs(sgn(p[6]), sgn(p[5]))) + lim, t3 = pix(subs(sgn(p[7]), sgn(p[6]))) + lim;
    const v2u tm = umax(umax(t0, t1), umax(t2, t3));
    // |p0 - q0| from the SATURATED difference the filters need anyway (max(w, -w): two instructions, not three): beyond +-32767 twice
    // of it is larger than any limit (elim <= 193 << 8) either way.  |p1 - q1| is halved and has to be exact.
    const v2u a = sabs(subs(sgn(p[4]), sgn(p[3]))), b = adu(p[2], p[5]);
    const v2u e = uadds(uadds(a, a), (b >> 1) & mku(0xff00));                     // 2|p0-q0| + |p1-q1|/2, saturating
    const v2u over = usubs(tm, lim + lim) | usubs(dh, lim) | usubs(e, elim);      // non-zero: leave the edge alone
    // `mask` is 1 / 0 per half here and is applied by a multiplication (lf_keep): one saturating subtraction makes it, where the
    // all-ones form takes two; the gate (lf_gate: 0 = open) is one more reason to leave the edge alone.
    // hev only matters where the edge is filtered, and there dh <= lim < 2^15: (thr - dh) >> 15, sign and all, is the mask.
    mask = usubs(one, over | gate);
    hev = as_v2u(as_u32(as_v2s(as_u32(thr - dh)) >> 15));
}
// a lane's switch for an edge: see masks
__device__ __forceinline__ v2u lf_gate(bool on)
{
    return mku(on ? 0 : 0xffff);
}
// the filter value of the lines `mask` lets through, 0 for the others
__device__ __forceinline__ v2s lf_keep(v2s f, v2u mask)
{
    return as_v2s(as_u32(as_v2u(as_u32(f)) * mask));
}

// filter_value = clamp(filter_value + 3 * (qs0 - ps0)) (loopfilter_filters.c:66, 176): three saturating adds of
// the saturated difference give the same result as one clamp of the exact sum (same-signed increments)
__device__ __forceinline__ v2s add3w(v2s f, v2s qs0, v2s ps0)
{
    const v2s w = subs(qs0, ps0);
#ifdef LF_ADD3_STEPWISE      // (through round 5: three saturating adds)
    return adds(adds(adds(f, w), w), w);
#else
    // ONE saturation of the exact sum f + 3 w (v_pk_mad_i16 ... clamp: the multiply-add is exact, the clamp works on its result): where
    // w itself was saturated 3 w lies beyond the range by more than any f brings back, so the result is the clamp of the exact sum
    // either way -- which is what the reference computes (loopfilter_filters.c:66, 176)
    u32 d;
    asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(d) : "v"(as_u32(w)), "s"(3u), "v"(as_u32(f)));
    return as_v2s(d);
#endif
}

// vp8_loop_filter_c (loopfilter_filters.c:51-95): inner edges, modifies p1 p0 q0 q1
__device__ __forceinline__ void lf_inner(v2u p[8], const Lim &L, v2u gate)
{
    v2u mask, hev;
    masks(p, L.lim, L.blim, L.thr, L.one, gate, mask, hev);
    v2s ps1 = sgn(p[2]), ps0 = sgn(p[3]), qs0 = sgn(p[4]), qs1 = sgn(p[5]);
    v2s f = as_v2s(as_u32(subs(ps1, qs1)) & as_u32(hev));
    f = lf_keep(add3w(f, qs0, ps0), mask);
    const v2s f1 = hib(adds(f, mks(0x0400)) >> 3), f2 = hib(adds(f, mks(0x0300)) >> 3);
    qs0 = subs(qs0, f1); ps0 = adds(ps0, f2);
    f = as_v2s(as_u32((f1 + mks(0x0100)) >> 1) & (~as_u32(hev) & 0xff00ff00u));
    qs1 = subs(qs1, f); ps1 = adds(ps1, f);
    p[2] = pix(ps1); p[3] = pix(ps0); p[4] = pix(qs0); p[5] = pix(qs1);
}

// vp8_mbloop_filter_c (loopfilter_filters.c:161-214): macroblock edges, modifies p2 p1 p0 q0 q1 q2
__device__ __forceinline__ void lf_mbedge(v2u p[8], const Lim &L, v2u gate)
{
    v2u mask, hev;
    masks(p, L.lim, L.mblim, L.thr, L.one, gate, mask, hev);
    v2s ps2 = sgn(p[1]), ps1 = sgn(p[2]), ps0 = sgn(p[3]), qs0 = sgn(p[4]), qs1 = sgn(p[5]), qs2 = sgn(p[6]);
    v2s f = lf_keep(add3w(subs(ps1, qs1), qs0, ps0), mask);
    v2s f2 = as_v2s(as_u32(f) & as_u32(hev));
    const v2s f1 = hib(adds(f2, mks(0x0400)) >> 3);
    f2 = hib(adds(f2, mks(0x0300)) >> 3);
    const v2s F = as_v2s(as_u32(f) & ~as_u32(hev)) >> 8;             // plain signed value, -128 .. 127
    // ((F * 27 + 63) >> 7) << 8 == (F * 54 + 126) with the low byte cleared (|F * 54 + 126| < 2^15): one multiply-add
    // and one AND instead of multiply-add, shift, shift
    v2s u = hib(F * 54 + 126);
    // (a line takes the hev branch's f1 / f2 or the 27-tap u, never both -- the other is zero: one update of p0 / q0, not two)
    qs0 = subs(qs0, as_v2s(as_u32(f1) | as_u32(u))); ps0 = adds(ps0, as_v2s(as_u32(f2) | as_u32(u)));
    u = hib(F * 36 + 126);
    qs1 = subs(qs1, u); ps1 = adds(ps1, u);
    u = hib(F * 18 + 126);
    qs2 = subs(qs2, u); ps2 = adds(ps2, u);
    p[1] = pix(ps2); p[2] = pix(ps1); p[3] = pix(ps0); p[4] = pix(qs0); p[5] = pix(qs1); p[6] = pix(qs2);
}

// vp8_loop_filter_simple_horizontal/vertical_edge_c (loopfilter_filters.c:292-355): modifies p0 q0
__device__ __forceinline__ void lf_simple(v2u p[8], v2u elim, v2u one, v2u gate)
{
    const v2u a = sabs(subs(sgn(p[4]), sgn(p[3]))), b = adu(p[2], p[5]);
    const v2u e = uadds(uadds(a, a), (b >> 1) & mku(0xff00));
    const v2u mask = usubs(one, usubs(e, elim) | gate);
    v2s ps1 = sgn(p[2]), ps0 = sgn(p[3]), qs0 = sgn(p[4]), qs1 = sgn(p[5]);
    const v2s f = lf_keep(add3w(subs(ps1, qs1), qs0, ps0), mask);
    const v2s f1 = hib(adds(f, mks(0x0400)) >> 3), f2 = hib(adds(f, mks(0x0300)) >> 3);
    p[4] = pix(subs(qs0, f1)); p[3] = pix(adds(ps0, f2));
}

// which filters the lanes of the wave need (wave-uniform) and each lane's gates
struct Gates { v2u mb, inner, mb_s, inner_s; bool any_normal, any_simple; };

// All edges of two pixel lines: a[0..4*W4+3] = positions -4 .. 4*W4-1 across the macroblock.  Order and
// gating as vp8_loop_filter_frame (loopfilter.c:265-299): the MB edge at 0 (if there is a neighbour),
// then the inner edges at 4, 8, 12 (if !skip_lf).
template <int W4>
__device__ __forceinline__ void filter_lines(v2u *a, const Gates &G, const Lim &L)
{
    if (G.any_normal) {
        lf_mbedge(a, L, G.mb);
#pragma unroll
        for (int e = 1; e < W4; e++) lf_inner(a + 4 * e, L, G.inner);
    }
    if (G.any_simple) {
        lf_simple(a, L.mblim, L.one, G.mb_s);
#pragma unroll
        for (int e = 1; e < W4; e++) lf_simple(a + 4 * e, L.blim, L.one, G.inner_s);
    }
}
// the same for two independent sets of lines at once (more instruction-level parallelism)
template <int W4>
__device__ __forceinline__ void filter_lines2(v2u *a, v2u *b, const Gates &G, const Lim &L)
{
    if (G.any_normal) {
        lf_mbedge(a, L, G.mb); lf_mbedge(b, L, G.mb);
#pragma unroll
        for (int e = 1; e < W4; e++) { lf_inner(a + 4 * e, L, G.inner); lf_inner(b + 4 * e, L, G.inner); }
    }
    if (G.any_simple) {
        lf_simple(a, L.mblim, L.one, G.mb_s); lf_simple(b, L.mblim, L.one, G.mb_s);
#pragma unroll
        for (int e = 1; e < W4; e++) { lf_simple(a + 4 * e, L.blim, L.one, G.inner_s); lf_simple(b + 4 * e, L.blim, L.one, G.inner_s); }
    }
}

template <int NX>
__device__ __forceinline__ void unpack_rows(const u32 *ra, const u32 *rb, v2u *a)
{
#pragma unroll
    for (int x = 0; x < NX; x++) {
        const u32 A = ra[x * 64], B = rb[x * 64];
        a[4 * x + 0] = as_v2u(perm(B, A, 0x040c000cu)); a[4 * x + 1] = as_v2u(perm(B, A, 0x050c010cu));
        a[4 * x + 2] = as_v2u(perm(B, A, 0x060c020cu)); a[4 * x + 3] = as_v2u(perm(B, A, 0x070c030cu));
    }
}
template <int NX>
__device__ __forceinline__ void pack_rows(u32 *ra, u32 *rb, const v2u *a)
{
#pragma unroll
    for (int x = 0; x < NX; x++) {
        const u32 p01 = as_u32(a[4 * x]), p11 = as_u32(a[4 * x + 1]), p21 = as_u32(a[4 * x + 2]), p31 = as_u32(a[4 * x + 3]);
        const u32 t01 = perm(p11, p01, 0x07030501u), t23 = perm(p31, p21, 0x07030501u);     // A0 A1 B0 B1 | A2 A3 B2 B3
        ra[x * 64] = perm(t23, t01, 0x05040100u);
        rb[x * 64] = perm(t23, t01, 0x07060302u);
    }
}

// One plane of one macroblock in the lane's LDS tile T[row * NX + xd][lane], NX = W4 + 1 dwords per row:
// row = y + 4 (y = -4 .. H-1), xd = 0 the four pixels left of the macroblock, xd = 1 .. W4 its own.
// gv / gh: gates of the vertical-edge and of the horizontal-edge pass.
template <int W4, int H>
__device__ __forceinline__ void filter_plane(u32 *T, const Gates &gv, const Gates &gh, const Lim &L)
{
    constexpr int NX = W4 + 1;
    // ---- vertical edges: rows (y, y+1) packed, all positions x = -4 .. 4*W4-1 in registers; two row pairs a time
#pragma unroll 1
    for (int rp = 0; rp < H / 4; rp++) {
        u32 *r0 = T + (4 + 4 * rp) * NX * 64, *r1 = r0 + NX * 64, *r2 = r1 + NX * 64, *r3 = r2 + NX * 64;
        v2u a[4 * NX], b[4 * NX];
        unpack_rows<NX>(r0, r1, a);
        unpack_rows<NX>(r2, r3, b);
        filter_lines2<W4>(a, b, gv, L);
        pack_rows<NX>(r0, r1, a);
        pack_rows<NX>(r2, r3, b);
    }
    // ---- horizontal edges: columns (x, x+1) packed, rows y = -4 .. H-1 of the two column pairs in registers
#pragma unroll 1
    for (int xd = 1; xd <= W4; xd++) {
        u32 *col = T + xd * 64;
        v2u lo[H + 4], hi[H + 4];
#pragma unroll
        for (int y = 0; y < H + 4; y++) {
            const u32 D = col[y * NX * 64];
            lo[y] = as_v2u(perm(D, D, 0x010c000cu));
            hi[y] = as_v2u(perm(D, D, 0x030c020cu));
        }
        filter_lines2<H / 4>(lo, hi, gh, L);
#pragma unroll
        for (int y = 1; y < H + 4; y++) col[y * NX * 64] = perm(as_u32(hi[y]), as_u32(lo[y]), 0x07050301u);
    }
}

// vp8_loop_filter_frame_init (loopfilter.c:117-201) for one macroblock
__device__ __forceinline__ int mb_level(const vp8ir_frame_hdr &h, int seg, int ref, int y_mode)
{
    int base = h.filter_level;
    if (h.segmentation_enabled) {
        if (h.mb_segment_abs_delta) base = h.segment_lf[seg];
        else { base += h.segment_lf[seg]; base = base < 0 ? 0 : (base > 63 ? 63 : base); }
    }
    if (!h.mode_ref_lf_delta_enabled) return base & 0xff;
    int v = base + h.ref_lf_deltas[ref];
    if (ref == VP8IR_INTRA_FRAME) {
        if (y_mode == VP8IR_B_PRED) v += h.mode_lf_deltas[0];
    } else {
        // mode_lf_lut (loopfilter.c:52-63): NEAREST, NEAR, NEW -> 2, ZERO -> 1, SPLIT -> 3
        const int m = y_mode == VP8IR_ZEROMV ? 1 : (y_mode == VP8IR_SPLITMV ? 3 : 2);
        v += h.mode_lf_deltas[m];
    }
    return v < 0 ? 0 : (v > 63 ? 63 : v);
}

// vp8_loop_filter_update_sharpness + hev threshold LUT (loopfilter.c:24-96)
__device__ __forceinline__ Lim mb_limits(int sharp, int level, int frame_type, v2u one)
{
    int ilimit = level >> (sharp > 0);
    ilimit >>= (sharp > 4);
    if (sharp > 0 && ilimit > 9 - sharp) ilimit = 9 - sharp;
    if (ilimit < 1) ilimit = 1;
    int thr;
    if (level >= 40) thr = frame_type == 0 ? 2 : 3;
    else if (level >= 20) thr = frame_type == 0 ? 1 : 2;
    else if (level >= 15) thr = 1;
    else thr = 0;
    Lim L;
    L.lim = mku(ilimit << 8); L.blim = mku(((2 * level + ilimit) & 0xff) << 8); L.mblim = mku(((2 * (level + 2) + ilimit) & 0xff) << 8);
    L.thr = mku(thr << 8);
    L.one = one;
    return L;
}


// ---------------- the loop filter streamed through registers (vp8_keyframe_simt.hip; per-block entry: vp8_lane_blocks.hip) ----------------
// one biased row dword (four pixels) <-> two column pairs: (x, x+1) and (x+2, x+3) in the two 16-bit halves
__device__ __forceinline__ v2u col_lo(u32 D) { return as_v2u(perm(D, D, 0x010c000cu)); }
__device__ __forceinline__ v2u col_hi(u32 D) { return as_v2u(perm(D, D, 0x030c020cu)); }
__device__ __forceinline__ u32 col_pack(v2u lo, v2u hi) { return perm(as_u32(hi), as_u32(lo), 0x07050301u); }

// One block row (four pixel rows) of one plane through the loop filter.  W4: dwords per row (4 luma, 2 chroma).
//   o[j][x]   in:  rows j = 0..3 of the block row as reconstructed (plain pixels)
//   s[j]      in:  the last four pixels of the macroblock to the left in these rows (biased), as its own filtering left
//                  them; out: after this macroblock's left edge
//   P[j][x]   in:  the four rows above (biased dwords), vertical edges done; out: these four rows, vertical edges and the
//                  edge above them done
//   top_mb         the edge above is the macroblock's top edge (first block row)
//   d[j][x]   out: the four rows above, final (biased dwords) -- but for their last dword, which the macroblock to the
//                  right may still change
// gv / gh: gates of the vertical-edge and of the horizontal-edge pass (loopfilter.c:265-299).
// The rows travel as packed dwords between the stages and are widened to 16-bit pairs one dword column at a time: the
// widened form of a whole block row and of the rows above it (2 x 32 registers for luma) is what decided whether a luma and
// a chroma wave fit one SIMD together.
template <int W4>
__device__ __forceinline__ void lf_block_row(const u32 (&o)[4][W4], u32 (&s)[4], u32 (&P)[4][W4], const bool top_mb,
                                             const Gates &gv, const Gates &gh, const Lim &L, u32 (&d)[4][W4])
{
    constexpr int NX = W4 + 1;
    v2u a[4 * NX], b[4 * NX];             // rows (0, 1) and (2, 3): positions -4 .. 4*W4-1
#pragma unroll
    for (int x = 0; x < NX; x++) {
        const u32 A0 = x ? o[0][x - 1] ^ VP8_LF_BIAS : s[0], A1 = x ? o[1][x - 1] ^ VP8_LF_BIAS : s[1];
        const u32 A2 = x ? o[2][x - 1] ^ VP8_LF_BIAS : s[2], A3 = x ? o[3][x - 1] ^ VP8_LF_BIAS : s[3];
        a[4 * x + 0] = as_v2u(perm(A1, A0, 0x040c000cu)); a[4 * x + 1] = as_v2u(perm(A1, A0, 0x050c010cu));
        a[4 * x + 2] = as_v2u(perm(A1, A0, 0x060c020cu)); a[4 * x + 3] = as_v2u(perm(A1, A0, 0x070c030cu));
        b[4 * x + 0] = as_v2u(perm(A3, A2, 0x040c000cu)); b[4 * x + 1] = as_v2u(perm(A3, A2, 0x050c010cu));
        b[4 * x + 2] = as_v2u(perm(A3, A2, 0x060c020cu)); b[4 * x + 3] = as_v2u(perm(A3, A2, 0x070c030cu));
    }
    filter_lines2<W4>(a, b, gv, L);
    {   // the left neighbour's last dword, back as rows
        const u32 t01 = perm(as_u32(a[1]), as_u32(a[0]), 0x07030501u), t23 = perm(as_u32(a[3]), as_u32(a[2]), 0x07030501u);
        const u32 u01 = perm(as_u32(b[1]), as_u32(b[0]), 0x07030501u), u23 = perm(as_u32(b[3]), as_u32(b[2]), 0x07030501u);
        s[0] = perm(t23, t01, 0x05040100u); s[1] = perm(t23, t01, 0x07060302u);
        s[2] = perm(u23, u01, 0x05040100u); s[3] = perm(u23, u01, 0x07060302u);
    }
    // the horizontal edge between the rows above (p3..p0) and this block row (q0..q3), one dword column -- two column pairs --
    // at a time.  (Not skipped when no lane of the wave wants the normal filter: the gates switch it off lane by lane.)
    const v2u elim_s = top_mb ? L.mblim : L.blim, gate_s = top_mb ? gh.mb_s : gh.inner_s;
#pragma unroll
    for (int x = 0; x < W4; x++) {
        const u32 a0 = as_u32(a[4 * x + 4]), a1 = as_u32(a[4 * x + 5]), a2 = as_u32(a[4 * x + 6]), a3 = as_u32(a[4 * x + 7]);
        const u32 b0 = as_u32(b[4 * x + 4]), b1 = as_u32(b[4 * x + 5]), b2 = as_u32(b[4 * x + 6]), b3 = as_u32(b[4 * x + 7]);
        v2u p[8] = { col_lo(P[0][x]), col_lo(P[1][x]), col_lo(P[2][x]), col_lo(P[3][x]),
                     as_v2u(perm(a1, a0, 0x050c010cu)), as_v2u(perm(a1, a0, 0x070c030cu)), as_v2u(perm(b1, b0, 0x050c010cu)), as_v2u(perm(b1, b0, 0x070c030cu)) };
        v2u q[8] = { col_hi(P[0][x]), col_hi(P[1][x]), col_hi(P[2][x]), col_hi(P[3][x]),
                     as_v2u(perm(a3, a2, 0x050c010cu)), as_v2u(perm(a3, a2, 0x070c030cu)), as_v2u(perm(b3, b2, 0x050c010cu)), as_v2u(perm(b3, b2, 0x070c030cu)) };
        if (top_mb) { lf_mbedge(p, L, gh.mb); lf_mbedge(q, L, gh.mb); }
        else { lf_inner(p, L, gh.inner); lf_inner(q, L, gh.inner); }
        if (gh.any_simple) { lf_simple(p, elim_s, L.one, gate_s); lf_simple(q, elim_s, L.one, gate_s); }
        // the rows above are done; this block row takes their place
#pragma unroll
        for (int j = 0; j < 4; j++) { d[j][x] = col_pack(p[j], q[j]); P[j][x] = col_pack(p[4 + j], q[4 + j]); }
        __builtin_amdgcn_sched_barrier(0);      // two lines' worth of temporaries at a time, not 2 * W4
    }
}


} // namespace
