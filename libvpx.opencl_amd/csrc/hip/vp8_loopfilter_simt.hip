// VP8 in-loop deblocking filter, "one macroblock row per LANE" formulation for gfx950.
//
// Same job as vp8_loopfilter.hip: vp8_loop_filter_frame (vp8/common/loopfilter.c:203-316) with the
// filters of vp8/common/loopfilter_filters.c (normal: vp8_loop_filter_c / vp8_mbloop_filter_c behind
// vp8_loop_filter_{mbv,bv,mbh,bh}_c; simple: vp8_loop_filter_simple_*), level / limit derivation of
// loopfilter.c:24-201.  Organised like vp8_recon_simt.hip, for the same reason (the path is bound by
// VALU issue, not by memory):
//
//   * lane p of a wave owns macroblock rows p, p+G, p+2G, ... of a strand of frames and filters one
//     whole macroblock per step, two macroblocks behind lane p-1; the raster order of the reference
//     (MB (r-1,c+1) has touched the three pixel columns left of it before MB (r,c) filters its top edge)
//     holds by construction;
//   * the filter arithmetic runs on TWO pixel lines per instruction as packed 16-bit lanes (v_pk_*):
//     rows (y, y+1) for the vertical edges, columns (x, x+1) for the horizontal ones; the signed-char
//     saturations of the reference become packed min/max;
//   * the macroblock (plus the four columns left of it and the four rows above it) sits in a per-lane,
//     lane-interleaved LDS tile, so both passes are short rolled loops over conflict-free ds_read_b32;
//   * pixels another macroblock will still modify are not written early: the four right-hand columns
//     wait in registers for the next macroblock's left edge, the four bottom rows travel to the lane
//     below by DPP wave shift and are written by it.  The first lane of a strand reads them back from
//     the frame (L2-coherent loads); the last lane of a strand and the last row of a frame write them.
//     Every frame byte is written once, as aligned 16-byte (luma) / 8-byte (chroma) row pieces.
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
typedef short v2s __attribute__((ext_vector_type(2)));     // the same pixel position of two lines

__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ u32 from_lane_above(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 load_l2(const unsigned char *p)
{
    return __hip_atomic_load((const u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
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
__device__ __forceinline__ u32 as_u32(v2s v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ v2s as_v2s(u32 v) { return __builtin_bit_cast(v2s, v); }
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
// x != 0 ? 0 : 0xffff (nz_clear) and x != 0 ? 0xffff : 0 (nz_set) per half, from max(1 - x, 0) by saturating subtraction.
// `one` is the constant 1 | 1 << 16 made opaque to LLVM (one empty asm at kernel entry, see Lim::one): with a visible
// constant the expression is canonicalised into a compare-and-select, which gfx950 can only do one half at a time.
__device__ __forceinline__ v2u nz_clear(v2u x, v2u one) { return mku(0) - usubs(one, x); }
__device__ __forceinline__ v2u nz_set(v2u x, v2u one) { return usubs(one, x) - one; }
__device__ __forceinline__ v2s sgn(v2u p) { return as_v2s(as_u32(p)); }                              // (already biased: see above)
__device__ __forceinline__ v2u pix(v2s s) { return as_v2u(as_u32(s)); }
#define VP8_LF_BIAS 0x80808080u     // four pixels <-> four biased pixels, on the way into and out of the LDS tile
__device__ __forceinline__ v2s hib(v2s v) { return as_v2s(as_u32(v) & 0xff00ff00u); }                // floor to a whole byte

struct Lim { v2u mblim, blim, lim, thr, one; };     // the limits, << 8; the opaque constant 1 of nz_clear / nz_set

// The filters are branch-free: `gate` (0xffff / 0 per lane) switches an edge off by clearing its filter mask,
// which makes every update the identity.  Straight-line code lets the scheduler interleave the independent
// pixel-line pairs, which is what hides the wait state gfx950 wants between dependent packed-math ops.

// vp8_filter_mask + vp8_hevmask (loopfilter_filters.c:27-49) for p[0..7] = p3 p2 p1 p0 q0 q1 q2 q3:
// mask = 0xffff where the edge is filtered, hev = 0xffff where the high-edge-variance rule applies
__device__ __forceinline__ void masks(const v2u p[8], v2u lim, v2u elim, v2u thr, v2u one, v2u gate, v2u &mask, v2u &hev)
{
    const v2u d10 = adu(p[2], p[3]), dq = adu(p[5], p[4]);
    const v2u dh = umax(d10, dq);
    v2u m = umax(umax(adu(p[0], p[1]), adu(p[1], p[2])), dh);
    m = umax(m, umax(adu(p[6], p[5]), adu(p[7], p[6])));
    const v2u a = adu(p[3], p[4]);
    const v2u e = uadds(uadds(a, a), (adu(p[2], p[5]) >> 1) & mku(0xff00));      // 2|p0-q0| + |p1-q1|/2, saturating
    const v2u over = usubs(m, lim) | usubs(e, elim);                              // non-zero: leave the edge alone
    mask = nz_clear(over, one) & gate;
    hev = nz_set(usubs(dh, thr), one);
}

// filter_value = clamp(filter_value + 3 * (qs0 - ps0)) (loopfilter_filters.c:66, 176): three saturating adds of
// the saturated difference give the same result as one clamp of the exact sum (same-signed increments)
__device__ __forceinline__ v2s add3w(v2s f, v2s qs0, v2s ps0)
{
    const v2s w = subs(qs0, ps0);
    return adds(adds(adds(f, w), w), w);
}

// vp8_loop_filter_c (loopfilter_filters.c:51-95): inner edges, modifies p1 p0 q0 q1
__device__ __forceinline__ void lf_inner(v2u p[8], const Lim &L, v2u gate)
{
    v2u mask, hev;
    masks(p, L.lim, L.blim, L.thr, L.one, gate, mask, hev);
    v2s ps1 = sgn(p[2]), ps0 = sgn(p[3]), qs0 = sgn(p[4]), qs1 = sgn(p[5]);
    v2s f = as_v2s(as_u32(subs(ps1, qs1)) & as_u32(hev));
    f = as_v2s(as_u32(add3w(f, qs0, ps0)) & as_u32(mask));
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
    v2s f = as_v2s(as_u32(add3w(subs(ps1, qs1), qs0, ps0)) & as_u32(mask));
    v2s f2 = as_v2s(as_u32(f) & as_u32(hev));
    const v2s f1 = hib(adds(f2, mks(0x0400)) >> 3);
    f2 = hib(adds(f2, mks(0x0300)) >> 3);
    qs0 = subs(qs0, f1); ps0 = adds(ps0, f2);
    const v2s F = as_v2s(as_u32(f) & ~as_u32(hev)) >> 8;             // plain signed value, -128 .. 127
    // ((F * 27 + 63) >> 7) << 8 == (F * 54 + 126) with the low byte cleared (|F * 54 + 126| < 2^15): one multiply-add
    // and one AND instead of multiply-add, shift, shift
    v2s u = hib(F * 54 + 126);
    qs0 = subs(qs0, u); ps0 = adds(ps0, u);
    u = hib(F * 36 + 126);
    qs1 = subs(qs1, u); ps1 = adds(ps1, u);
    u = hib(F * 18 + 126);
    qs2 = subs(qs2, u); ps2 = adds(ps2, u);
    p[1] = pix(ps2); p[2] = pix(ps1); p[3] = pix(ps0); p[4] = pix(qs0); p[5] = pix(qs1); p[6] = pix(qs2);
}

// vp8_loop_filter_simple_horizontal/vertical_edge_c (loopfilter_filters.c:292-355): modifies p0 q0
__device__ __forceinline__ void lf_simple(v2u p[8], v2u elim, v2u one, v2u gate)
{
    const v2u a = adu(p[3], p[4]);
    const v2u e = uadds(uadds(a, a), (adu(p[2], p[5]) >> 1) & mku(0xff00));
    const v2u mask = nz_clear(usubs(e, elim), one) & gate;
    v2s ps1 = sgn(p[2]), ps0 = sgn(p[3]), qs0 = sgn(p[4]), qs1 = sgn(p[5]);
    const v2s f = as_v2s(as_u32(add3w(subs(ps1, qs1), qs0, ps0)) & as_u32(mask));
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

} // namespace

// grid = waves (one wave per block); lgG, P, nstrands as in vp8_recon_simt_kernel.  Works in place on the
// jobs' macroblock-tiled scratch frames (DevJob::tile, see VP8_TILE_BYTES): a macroblock is three 128-byte
// lines -- luma rows 0..7, luma rows 8..15, U+V -- and every line is written exactly once, whole, by the lane
// that knows its final content:
//   line 0 of MB (r,c)    by its own lane, one step later (after MB (r,c+1) revisited its last 4 columns);
//   line 1 and the chroma line by the lane below (which filters their last three rows), or by the own lane
//   when nobody is below (last row of the frame) or the lane below reads them back from memory (the first
//   lane of a strand follows the last one).
//
// PLANES selects what a wave filters: LF_BOTH (one wave does a macroblock's luma, then U, then V), or LF_LUMA / LF_CHROMA
// for the split launch, where a luma kernel and a chroma kernel run side by side.  The two halves share nothing but the
// macroblock descriptors -- separate lines of the tiled scratch frame, separate planes of the frame buffer -- and a luma
// wave and a chroma wave together fit one SIMD (registers and LDS), which a pair of whole-macroblock waves does not: the
// SIMD then has two instruction streams to issue from instead of one that stalls on every LDS and memory round trip.
enum { LF_BOTH = 0, LF_LUMA = 1, LF_CHROMA = 2 };
template <int PLANES>
__device__ __forceinline__ void lf_simt_body(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    constexpr bool DO_Y = PLANES != LF_CHROMA, DO_C = PLANES != LF_LUMA;
    __shared__ u32 tile[(DO_Y ? 100 : 36) * 64];   // luma: 20 rows x 5 dwords; chroma (reuses it): 12 rows x 3 dwords
    const int lane = threadIdx.x;
    const int G = 1 << lgG;
    const int pos = lane & (G - 1);
    const int spw = 64 >> lgG;
    const int strand = blockIdx.x * spw + (lane >> lgG);
    const int cols = g.mb_cols, rows = g.mb_rows;
    const int myjobs = strand < njobs ? (njobs - strand + nstrands - 1) / nstrands : 0;
    const int Vmax = myjobs * rows;
    const int wavejobs = (njobs - (int)blockIdx.x * spw + nstrands - 1) / nstrands;
    const int T = ((wavejobs * rows + G - 1) >> lgG) * P + 2 * (G - 1);
    const long rowbytes = (long)cols * VP8_TILE_BYTES;
    u32 *const TL = tile + lane;
    v2u one = mku(1);
    asm volatile("" : "+v"(one));            // see nz_clear

    // ---- per-lane row state
    g_cu32p mbp = nullptr;
    u32 nx_w0 = 0, nx_w1 = 0;                   // descriptor words 0, 1 of the macroblock after the current one, fetched a step ahead
    g_u8p trow = nullptr;                        // tile (r, 0)
    g_u8p rasY = nullptr, rasU = nullptr, rasV = nullptr;   // raster == 1: pixel (0,0) of MB row r in the frame buffer
    const DevJob *job = jobs;
    int r = 0;
    bool lf_on = false, simple = false;
    // the previous macroblock of the row: its first 12 (chroma: 4) pixel columns, final, and its last 4,
    // which the current macroblock's left edge may still change
    u32 pbY[16][3], sY[16], pbU[8], sU[8], pbV[8], sV[8];
    // what the lane below asks for: luma rows 8..15 and the chroma rows of the macroblock finished two steps ago
    u32 hY[8][4], hU[8][2], hV[8][2];
    // raster output: rows of an even macroblock wait one step for their right-hand neighbour, so that a frame row is
    // written in 32-byte pieces (two 16-byte stores back to back) instead of 16-byte ones a step apart
    u32x4 holdA[8], holdB[8];
    u32x2 holdU[8], holdV[8];
    if constexpr (DO_Y) {
#pragma unroll
        for (int y = 0; y < 16; y++) { pbY[y][0] = pbY[y][1] = pbY[y][2] = sY[y] = 0; }
#pragma unroll
        for (int y = 0; y < 8; y++) hY[y][0] = hY[y][1] = hY[y][2] = hY[y][3] = 0;
    }
    if constexpr (DO_C) {
#pragma unroll
        for (int y = 0; y < 8; y++) { pbU[y] = sU[y] = pbV[y] = sV[y] = 0; hU[y][0] = hU[y][1] = hV[y][0] = hV[y][1] = 0; }
    }

    int c = -2 * pos, V = pos;
    STAMP_DECL
#pragma unroll 1
    for (int t = 0; t < T; ++t, ++c) {
        STAMP(0)
        if (c == P) { c = 0; V += G; }
        // rows 8..15 (luma) / all rows (chroma) of the macroblock above, from the lane above
        u32 tY[8][4], tU[8][2], tV[8][2];
#pragma unroll
        for (int y = 0; y < 8; y++) {
            if constexpr (DO_Y) {
#pragma unroll
                for (int i = 0; i < 4; i++) tY[y][i] = from_lane_above(hY[y][i]);
            }
            if constexpr (DO_C) {
                tU[y][0] = from_lane_above(hU[y][0]); tU[y][1] = from_lane_above(hU[y][1]);
                tV[y][0] = from_lane_above(hV[y][0]); tV[y][1] = from_lane_above(hV[y][1]);
            }
        }

        // What the lane below will fetch at the start of the next step: the macroblock held from the previous
        // step.  Its last four columns are still provisional if this step filters a left edge against them
        // (they are fixed up below, after the vertical-edge pass); otherwise -- end of a row, idle step --
        // they are final as they stand.
#pragma unroll
        for (int y = 0; y < 8; y++) {
            if constexpr (DO_Y) { hY[y][0] = pbY[8 + y][0]; hY[y][1] = pbY[8 + y][1]; hY[y][2] = pbY[8 + y][2]; hY[y][3] = sY[8 + y]; }
            if constexpr (DO_C) { hU[y][0] = pbU[y]; hU[y][1] = sU[y]; hV[y][0] = pbV[y]; hV[y][1] = sV[y]; }
        }

        STAMP(1)
        const bool act = c >= 0 && c < cols && V < Vmax;
        if (act) {
            if (c == 0) {
                const int j = V / rows;
                r = V - j * rows;
                job = jobs + (strand + j * nstrands);
                const vp8ir_frame_hdr &h = job->hdr;
                lf_on = h.filter_level != 0;
                simple = h.filter_type == 1;
                mbp = (g_cu32p)(job->mbs + (long)r * cols);
                nx_w0 = mbp[0]; nx_w1 = mbp[1];
                trow = (g_u8p)(job->tile + (long)r * rowbytes);
                rasY = (g_u8p)(job->dst + g.y_off + (long)r * 16 * g.y_stride);
                rasU = (g_u8p)(job->dst + g.u_off + (long)r * 8 * g.uv_stride);
                rasV = (g_u8p)(job->dst + g.v_off + (long)r * 8 * g.uv_stride);
            }
            // (the descriptor was fetched a step ago: the tile loads below go out at once instead of behind a memory round trip)
            const u32 w0 = nx_w0, w1 = nx_w1;
            nx_w0 = mbp[16]; nx_w1 = mbp[17];     // (past the end of a row: the next row's first macroblock, or padding -- unused)
            if (lf_on || raster) {      // raster output: an unfiltered frame is still carried from the scratch to its frame buffer
            const vp8ir_frame_hdr &h = job->hdr;
            const int y_mode = w0 & 0xff, ref_frame = (w0 >> 16) & 0xff;
            const u32 flags = w0 >> 24;
            const int level = mb_level(h, w1 & 3, ref_frame & 3, y_mode);
            const Lim L = mb_limits(h.sharpness_level, level, h.frame_type, one);
            const bool on = lf_on && level != 0;
            const bool skip_lf = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV && (flags & VP8IR_MB_SKIP);
            const bool mbv = on && c > 0, inner = on && !skip_lf, mbh = on && r > 0;
#ifdef VP8_LF_NOFILTER    // measurement aid: data movement only
            const bool any_normal = false, any_simple = false;
#else
            const bool any_normal = __builtin_amdgcn_ballot_w64(on && !simple) != 0;
            const bool any_simple = __builtin_amdgcn_ballot_w64(on && simple) != 0;
#endif
            auto gate = [](bool b) { return mku(b ? 0xffff : 0); };
            const Gates gvY = { gate(mbv && !simple), gate(inner && !simple), gate(mbv && simple), gate(inner && simple), any_normal, any_simple };
            const Gates ghY = { gate(mbh && !simple), gate(inner && !simple), gate(mbh && simple), gate(inner && simple), any_normal, any_simple };
            // the simple filter leaves chroma alone (loopfilter.c:283-299)
            const Gates gvC = { gvY.mb, gvY.inner, mku(0), mku(0), any_normal, false };
            const Gates ghC = { ghY.mb, ghY.inner, mku(0), mku(0), any_normal, false };
            const bool last_col = c == cols - 1;
            // lines another lane would otherwise finish are written here when nobody below takes them over
            const bool write_bottom = pos == G - 1 || r == rows - 1;
            const bool readback = r > 0 && pos == 0;

            g_u8p tp = trow + (long)c * VP8_TILE_BYTES;           // this macroblock's tile
            // Where finished lines go.  raster == 0: back into the tiled scratch frame (vp8_detile_kernel converts later).
            // raster == 1: FINAL lines straight into the raster frame buffer; lines that are only handed to the first
            // lane of the strand (write_bottom on a row that is not the frame's last) still travel through the scratch.
            const bool ras = raster != 0, ras_bottom = ras && r == rows - 1;
            const int ysY = ras ? g.y_stride : 16, ysC = ras ? g.uv_stride : 8;
            const int ybY = ras_bottom ? g.y_stride : 16, ybC = ras_bottom ? g.uv_stride : 8;
            g_u8p o_left_lo = ras ? rasY + (c - 1) * 16 : tp - VP8_TILE_BYTES;                                   // MB c-1 rows 0..7
            g_u8p o_left_hi = ras_bottom ? rasY + (c - 1) * 16 + 8 * g.y_stride : tp - VP8_TILE_BYTES + 128;      // MB c-1 rows 8..15
            g_u8p o_own_lo = ras ? rasY + c * 16 : tp, o_own_hi = ras_bottom ? rasY + c * 16 + 8 * g.y_stride : tp + 128;
            g_u8p o_above = ras ? rasY - 8 * g.y_stride + c * 16 : tp - rowbytes + 128;                           // MB (r-1, c) rows 8..15
            u32x4 inY[16], inU[4], inV[4];
            if constexpr (DO_Y) {
#pragma unroll
                for (int y = 0; y < 16; y++) inY[y] = *(g_cu32x4p)(tp + 16 * y);
            }
            if constexpr (DO_C) {
#pragma unroll
                for (int y = 0; y < 4; y++) { inU[y] = *(g_cu32x4p)(tp + 256 + 16 * y); inV[y] = *(g_cu32x4p)(tp + 320 + 16 * y); }
            }
            if (readback) {
                const unsigned char *ta = (const unsigned char *)tp - rowbytes;
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    if constexpr (DO_Y) {
                        const unsigned long long a = load_l2_64(ta + 128 + 16 * y), b = load_l2_64(ta + 128 + 16 * y + 8);
                        tY[y][0] = (u32)a; tY[y][1] = (u32)(a >> 32); tY[y][2] = (u32)b; tY[y][3] = (u32)(b >> 32);
                    }
                    if constexpr (DO_C) {
                        const unsigned long long u = load_l2_64(ta + 256 + 8 * y), v = load_l2_64(ta + 320 + 8 * y);
                        tU[y][0] = (u32)u; tU[y][1] = (u32)(u >> 32); tV[y][0] = (u32)v; tV[y][1] = (u32)(v >> 32);
                    }
                }
            }

            STAMP(2)
            // =============================== luma ===============================
            if constexpr (DO_Y) {
#pragma unroll
            for (int y = 0; y < 16; y++) {
                u32 *row = TL + (4 + y) * 5 * 64;
                row[0] = sY[y] ^ VP8_LF_BIAS; row[64] = inY[y].x ^ VP8_LF_BIAS; row[128] = inY[y].y ^ VP8_LF_BIAS; row[192] = inY[y].z ^ VP8_LF_BIAS; row[256] = inY[y].w ^ VP8_LF_BIAS;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32 *row = TL + j * 5 * 64;
#pragma unroll
                for (int i = 0; i < 4; i++) row[(1 + i) * 64] = tY[4 + j][i] ^ VP8_LF_BIAS;
            }
            STAMP(3)
            filter_plane<4, 16>(TL, gvY, ghY, L);
            STAMP(4)
            // ---- the macroblock to the left is final now: its 12 held columns + the 4 just revisited
            if (c > 0) {
#pragma unroll
                for (int y = 0; y < 16; y++) {
                    const u32 s = TL[(4 + y) * 5 * 64] ^ VP8_LF_BIAS;
                    if (y < 8) {
                        const u32x4 v = { pbY[y][0], pbY[y][1], pbY[y][2], s };
                        if (!ras) *(g_u32x4p)(o_left_lo + ysY * y) = v;
                        else if ((c - 1) & 1) { *(g_u32x4p)(o_left_lo + ysY * y - 16) = holdA[y]; *(g_u32x4p)(o_left_lo + ysY * y) = v; }
                        else holdA[y] = v;
                    }
                    else if (write_bottom) *(g_u32x4p)(o_left_hi + ybY * (y - 8)) = (u32x4){ pbY[y][0], pbY[y][1], pbY[y][2], s };
                    if (y >= 8) hY[y - 8][3] = s;
                }
            }
            // ---- line 1 of the macroblock above: rows 8..12 as received, rows 13..15 filtered
            if (r > 0) {
                const bool pair_hold = ras && !(c & 1) && !last_col, pair_flush = ras && (c & 1);
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    u32x4 v;
                    if (y < 5) v = (u32x4){ tY[y][0], tY[y][1], tY[y][2], tY[y][3] };
                    else { const u32 *row = TL + (y - 4) * 5 * 64; v = (u32x4){ row[64] ^ VP8_LF_BIAS, row[128] ^ VP8_LF_BIAS, row[192] ^ VP8_LF_BIAS, row[256] ^ VP8_LF_BIAS }; }
                    if (pair_hold) holdB[y] = v;
                    else {
                        if (pair_flush) *(g_u32x4p)(o_above + ysY * y - 16) = holdB[y];
                        *(g_u32x4p)(o_above + ysY * y) = v;
                    }
                }
            }
            // ---- this macroblock: hold it, or finish it at the end of the row
#pragma unroll
            for (int y = 0; y < 16; y++) {
                const u32 *row = TL + (4 + y) * 5 * 64;
                const u32 d0 = row[64] ^ VP8_LF_BIAS, d1 = row[128] ^ VP8_LF_BIAS, d2 = row[192] ^ VP8_LF_BIAS, d3 = row[256] ^ VP8_LF_BIAS;
                pbY[y][0] = d0; pbY[y][1] = d1; pbY[y][2] = d2; sY[y] = d3;
                if (last_col && y < 8) {
                    if (ras && (c & 1)) *(g_u32x4p)(o_own_lo + ysY * y - 16) = holdA[y];      // its even left neighbour was waiting
                    *(g_u32x4p)(o_own_lo + ysY * y) = (u32x4){ d0, d1, d2, d3 };
                }
                if (last_col && y >= 8 && write_bottom) *(g_u32x4p)(o_own_hi + ybY * (y - 8)) = (u32x4){ d0, d1, d2, d3 };
            }
            }
            STAMP(5)
            // =============================== chroma ===============================
            if constexpr (DO_C) {
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                g_u8p tc = tp + (pl ? 320 : 256);
                g_u8p rasC = pl ? rasV : rasU;
                g_u8p oc_left = ras_bottom ? rasC + (c - 1) * 8 : tc - VP8_TILE_BYTES, oc_own = ras_bottom ? rasC + c * 8 : tc;
                g_u8p oc_above = ras ? rasC - 8 * g.uv_stride + c * 8 : tc - rowbytes;
                u32 (&pb)[8] = pl ? pbV : pbU;
                u32 (&sC)[8] = pl ? sV : sU;
                u32 (&tC)[8][2] = pl ? tV : tU;
                u32 (&hC)[8][2] = pl ? hV : hU;
                const u32x4 (&in)[4] = pl ? inV : inU;
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    u32 *row = TL + (4 + y) * 3 * 64;
                    row[0] = sC[y] ^ VP8_LF_BIAS;
                    row[64] = ((y & 1) ? in[y >> 1].z : in[y >> 1].x) ^ VP8_LF_BIAS; row[128] = ((y & 1) ? in[y >> 1].w : in[y >> 1].y) ^ VP8_LF_BIAS;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    u32 *row = TL + j * 3 * 64;
                    row[64] = tC[4 + j][0] ^ VP8_LF_BIAS; row[128] = tC[4 + j][1] ^ VP8_LF_BIAS;
                }
                STAMP(6)
                filter_plane<2, 8>(TL, gvC, ghC, L);
                STAMP(7)
                if (c > 0) {
#pragma unroll
                    for (int y = 0; y < 8; y++) {
                        const u32 s = TL[(4 + y) * 3 * 64] ^ VP8_LF_BIAS;
                        if (write_bottom) *(g_u32x2p)(oc_left + ybC * y) = (u32x2){ pb[y], s };
                        hC[y][1] = s;
                    }
                }
                if (r > 0) {
                    const bool pair_hold = ras && !(c & 1) && !last_col, pair_flush = ras && (c & 1);
                    u32x2 (&hold)[8] = pl ? holdV : holdU;
#pragma unroll
                    for (int y = 0; y < 8; y++) {
                        u32x2 v;
                        if (y < 5) v = (u32x2){ tC[y][0], tC[y][1] };
                        else { const u32 *row = TL + (y - 4) * 3 * 64; v = (u32x2){ row[64] ^ VP8_LF_BIAS, row[128] ^ VP8_LF_BIAS }; }
                        if (pair_hold) hold[y] = v;
                        else {
                            if (pair_flush) *(g_u32x2p)(oc_above + ysC * y - 8) = hold[y];
                            *(g_u32x2p)(oc_above + ysC * y) = v;
                        }
                    }
                }
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const u32 *row = TL + (4 + y) * 3 * 64;
                    const u32 d0 = row[64] ^ VP8_LF_BIAS, d1 = row[128] ^ VP8_LF_BIAS;
                    pb[y] = d0; sC[y] = d1;
                    if (last_col && write_bottom) *(g_u32x2p)(oc_own + ybC * y) = (u32x2){ d0, d1 };
                }
                STAMP(8)
            }
            }
            }
            mbp += 16;
        }
    }
    STAMP_FLUSH(vp8_stamps_lf)
}

extern "C" __global__ void __launch_bounds__(64)
vp8_loopfilter_simt_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    lf_simt_body<LF_BOTH>(jobs, njobs, g, lgG, P, nstrands, raster);
}
extern "C" __global__ void __launch_bounds__(64)
vp8_loopfilter_simt_luma_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    lf_simt_body<LF_LUMA>(jobs, njobs, g, lgG, P, nstrands, raster);
}
extern "C" __global__ void __launch_bounds__(64)
vp8_loopfilter_simt_chroma_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    lf_simt_body<LF_CHROMA>(jobs, njobs, g, lgG, P, nstrands, raster);
}
