// VP8 key frames in ONE pass: reconstruction and in-loop deblocking fused, "one macroblock row per LANE", for gfx950.
//
// What it replaces (all paths relative to the reference tree): the macroblock loop of vp8_decode_frame
// (vp8/decoder/decodframe.c:1116-1129 -> decode_mb_row :334 -> decode_macroblock :112, with reconintra.c, reconintra4x4.c,
// dequantize.c, idctllm.c, idct_blk.c behind it) AND vp8_loop_filter_frame (vp8/common/loopfilter.c:203-316, filters of
// loopfilter_filters.c) for frames whose macroblocks are all intra.  vp8_recon_simt.hip and vp8_loopfilter_simt.hip do the
// same work as two kernels with a macroblock-tiled scratch frame between them; here a lane reconstructs a macroblock and
// filters it in the same step, the pixels never leave its registers in between, and the finished rows go straight to the
// raster frame buffer:
//
//   * the schedule is that of the two kernels (they already shared it): lane p owns macroblock rows p, p+G, ... of a
//     strand of frames and runs two macroblocks behind lane p-1, so that prediction (left, above, above-right, all
//     UNFILTERED) and filtering (left and above neighbours FILTERED, the above one also by its right-hand neighbour)
//     find their inputs finished by construction.  A lane therefore carries two histories for the lane below: the
//     unfiltered bottom pixel line of its last two macroblocks (prediction) and the filtered four bottom rows of the
//     macroblock it finished two steps ago (the filter's context rows p3..p0), both fetched by DPP wave shift;
//   * the loop filter is STREAMED through registers, one block row (four pixel rows) at a time, right behind the
//     prediction of that block row: vertical edges on rows packed in pairs (rows y, y+1 in the two 16-bit halves),
//     then the horizontal edge ABOVE the block row on columns packed in pairs, against the four rows kept from the
//     previous block row (or, for the first one, the four bottom rows of the macroblock above).  That is the
//     reference's order -- all vertical edges of a macroblock, then its horizontal ones, macroblocks in raster order --
//     for every pixel, because an edge only ever reads pixels the edges before it in that order have finished.  No LDS
//     tile, no round trip between the two passes;
//   * rows are written when they are final: the four rows above a block row after its top edge, shifted four pixels to
//     the left -- the last four pixels of a row of the macroblock to the left are only final once this macroblock's left
//     edge has been filtered, so a 16-byte piece is { its last dword, this macroblock's first three }; the four
//     bottom rows of a macroblock are written by the lane below, after its top edge.  Every frame byte is written once;
//   * only the first lane of a strand, whose predecessor row sits on the strand's LAST lane, reads its context back
//     from memory: the last lane leaves the unfiltered bottom line and the filtered bottom rows of its macroblocks in a
//     hand-over tile (DevJob::tile, VP8_TILE_BYTES per macroblock, written and read by lanes of one wave).
//
// The residual transform is the cooperative one of vp8_recon_simt.hip (blocks with coefficients queued by all lanes,
// coefficients by LDS-DMA, one block per lane and round).  Integer only; no MFMA by design.
#include "vp8_simt_prims.hip.h"

namespace {

typedef u32x4 u32x4_u __attribute__((aligned(4)));      // 16-byte pieces at 4-byte alignment (the shifted row pieces)
typedef u32x2 u32x2_u __attribute__((aligned(4)));
typedef GLOBAL_AS u32x4_u *g_u32x4up;
typedef GLOBAL_AS u32x2_u *g_u32x2up;

// one biased row dword (four pixels) <-> two column pairs: (x, x+1) and (x+2, x+3) in the two 16-bit halves
__device__ __forceinline__ v2u col_lo(u32 D) { return as_v2u(perm(D, D, 0x010c000cu)); }
__device__ __forceinline__ v2u col_hi(u32 D) { return as_v2u(perm(D, D, 0x030c020cu)); }
__device__ __forceinline__ u32 col_pack(v2u lo, v2u hi) { return perm(as_u32(hi), as_u32(lo), 0x07050301u); }

// hand-over tile of a macroblock (VP8_TILE_BYTES): what the first lane of a strand reads back
enum {
    HO_Y_ROWS = 0,        // luma rows 12..15, filtered as far as this macroblock row goes: 4 x 16 B
    HO_Y_LINE = 64,       // unfiltered luma row 15: 16 B
    HO_U_ROWS = 128,      // U rows 4..7: 4 x 8 B
    HO_V_ROWS = 160,
    HO_U_LINE = 192,      // unfiltered U row 7: 8 B
    HO_V_LINE = 200
};

// One block row (four pixel rows) of one plane through the loop filter.  W4: dwords per row (4 luma, 2 chroma).
//   o[j][x]     in:  rows j = 0..3 of the block row as reconstructed (plain pixels)
//   s[j]        in:  the last four pixels of the macroblock to the left in these rows (biased), as its own filtering left
//                    them; out: after this macroblock's left edge
//   PL/PH[x][j] in:  the four rows above (biased, column pairs), vertical edges done; out: these four rows, vertical edges
//                    and the edge above them done
//   top_mb           the edge above is the macroblock's top edge (first block row)
//   d[j][x]     out: the four rows above, final (biased dwords) -- but for their last dword, which the macroblock to the
//                    right may still change
// gv / gh: gates of the vertical-edge and of the horizontal-edge pass (loopfilter.c:265-299)
template <int W4>
__device__ __forceinline__ void lf_block_row(const u32 (&o)[4][W4], u32 (&s)[4], v2u (&PL)[W4][4], v2u (&PH)[W4][4], const bool top_mb,
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
    // the block row as column pairs
    v2u CL[W4][4], CH[W4][4];
#pragma unroll
    for (int x = 0; x < W4; x++) {
        const u32 a0 = as_u32(a[4 * x + 4]), a1 = as_u32(a[4 * x + 5]), a2 = as_u32(a[4 * x + 6]), a3 = as_u32(a[4 * x + 7]);
        const u32 b0 = as_u32(b[4 * x + 4]), b1 = as_u32(b[4 * x + 5]), b2 = as_u32(b[4 * x + 6]), b3 = as_u32(b[4 * x + 7]);
        CL[x][0] = as_v2u(perm(a1, a0, 0x050c010cu)); CL[x][1] = as_v2u(perm(a1, a0, 0x070c030cu));
        CH[x][0] = as_v2u(perm(a3, a2, 0x050c010cu)); CH[x][1] = as_v2u(perm(a3, a2, 0x070c030cu));
        CL[x][2] = as_v2u(perm(b1, b0, 0x050c010cu)); CL[x][3] = as_v2u(perm(b1, b0, 0x070c030cu));
        CH[x][2] = as_v2u(perm(b3, b2, 0x050c010cu)); CH[x][3] = as_v2u(perm(b3, b2, 0x070c030cu));
    }
    // the horizontal edge between the rows above (p3..p0) and this block row (q0..q3)
    if (gh.any_normal) {
        if (top_mb) {
#pragma unroll
            for (int x = 0; x < W4; x++) {
                v2u p[8] = { PL[x][0], PL[x][1], PL[x][2], PL[x][3], CL[x][0], CL[x][1], CL[x][2], CL[x][3] };
                v2u q[8] = { PH[x][0], PH[x][1], PH[x][2], PH[x][3], CH[x][0], CH[x][1], CH[x][2], CH[x][3] };
                lf_mbedge(p, L, gh.mb); lf_mbedge(q, L, gh.mb);
#pragma unroll
                for (int j = 0; j < 4; j++) { PL[x][j] = p[j]; CL[x][j] = p[4 + j]; PH[x][j] = q[j]; CH[x][j] = q[4 + j]; }
            }
        } else {
#pragma unroll
            for (int x = 0; x < W4; x++) {
                v2u p[8] = { PL[x][0], PL[x][1], PL[x][2], PL[x][3], CL[x][0], CL[x][1], CL[x][2], CL[x][3] };
                v2u q[8] = { PH[x][0], PH[x][1], PH[x][2], PH[x][3], CH[x][0], CH[x][1], CH[x][2], CH[x][3] };
                lf_inner(p, L, gh.inner); lf_inner(q, L, gh.inner);
#pragma unroll
                for (int j = 0; j < 4; j++) { PL[x][j] = p[j]; CL[x][j] = p[4 + j]; PH[x][j] = q[j]; CH[x][j] = q[4 + j]; }
            }
        }
    }
    if (gh.any_simple) {
        const v2u elim = top_mb ? L.mblim : L.blim, gate = top_mb ? gh.mb_s : gh.inner_s;
#pragma unroll
        for (int x = 0; x < W4; x++) {
            v2u p[8] = { PL[x][0], PL[x][1], PL[x][2], PL[x][3], CL[x][0], CL[x][1], CL[x][2], CL[x][3] };
            v2u q[8] = { PH[x][0], PH[x][1], PH[x][2], PH[x][3], CH[x][0], CH[x][1], CH[x][2], CH[x][3] };
            lf_simple(p, elim, L.one, gate); lf_simple(q, elim, L.one, gate);
            PL[x][3] = p[3]; CL[x][0] = p[4]; PH[x][3] = q[3]; CH[x][0] = q[4];
        }
    }
    // the rows above are done; this block row takes their place
#pragma unroll
    for (int x = 0; x < W4; x++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d[j][x] = col_pack(PL[x][j], PH[x][j]);
            PL[x][j] = CL[x][j]; PH[x][j] = CH[x][j];
        }
    }
}

// loop-filter levels of a frame for its four segments, a byte each: macroblocks with a 16x16 mode / B_PRED macroblocks
// (vp8_loop_filter_frame_init, loopfilter.c:117-201, for intra frames); all zero when the frame is not filtered (onyxd_if.c:576)
__device__ __forceinline__ void frame_levels(const vp8ir_frame_hdr &h, u32 &plain, u32 &bpred)
{
    plain = bpred = 0;
    if (!h.filter_level) return;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        plain |= (u32)mb_level(h, s, VP8IR_INTRA_FRAME, VP8IR_DC_PRED) << (8 * s);
        bpred |= (u32)mb_level(h, s, VP8IR_INTRA_FRAME, VP8IR_B_PRED) << (8 * s);
    }
}

#define SWAP_U32(a, b) { const u32 t_ = (a); (a) = (b); (b) = t_; }

} // namespace

// grid = waves (one wave per block); lgG, P, nstrands as in vp8_recon_simt_kernel.  Every job must be a key frame.  The frames
// go to the jobs' raster frame buffers (DevJob::dst, borders not included: vp8_extend_kernel); DevJob::tile is the hand-over
// scratch of the strands' last lanes.  `dummy`: 512 bytes of scratch nobody reads (idle lanes store there: every memory
// instruction of the step loop is unconditional, see vp8_recon_simt.hip on s_waitcnt).
extern "C" __global__ void __launch_bounds__(64)
vp8_keyframe_simt_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy)
{
    __shared__ __attribute__((aligned(16))) u32 s_res[64 * 68];
    __shared__ __attribute__((aligned(16))) u32 s_stage[8 * 2 * 64 * 4];
    __shared__ u32 s_queue[512];
    __shared__ __attribute__((aligned(16))) u32x4 s_tab[64];
    __shared__ __attribute__((aligned(16))) u32 s_y2dc[64 * 8];
    const int lane = threadIdx.x;
    const int G = 1 << lgG;
    const int pos = lane & (G - 1);
    const int spw = 64 >> lgG;
    const int strand = blockIdx.x * spw + (lane >> lgG);
    const int cols = g.mb_cols, rows = g.mb_rows;
    const long rowbytes = (long)cols * VP8_TILE_BYTES;
    const int myjobs = strand < njobs ? (njobs - strand + nstrands - 1) / nstrands : 0;
    const int Vmax = myjobs * rows;
    const int wavejobs = (njobs - (int)blockIdx.x * spw + nstrands - 1) / nstrands;
    const int T = ((wavejobs * rows + G - 1) >> lgG) * P + 2 * (G - 1);
    u32 *const my_res = s_res + lane * 68;
    const u32 stage_lane = (u32)(unsigned long)(lds_vp)s_stage + lane * 16;
    const int ysY = g.y_stride, ysC = g.uv_stride;
    v2u one = mku(1);
    asm volatile("" : "+v"(one));            // see nz_clear

    // ---- per-lane row state; the pointers are valid addresses at all times
    g_cu32p mbp = (g_cu32p)jobs[0].mbs;
    g_cs16p cfp = (g_cs16p)jobs[0].coef;
    g_u8p tp = (g_u8p)dummy;                    // hand-over tile of the current macroblock
    g_cu8p abp = (g_cu8p)dummy;                 // ... of the macroblock above it
    g_u8p rasY = (g_u8p)dummy, rasU = (g_u8p)dummy, rasV = (g_u8p)dummy;     // pixel (0, 0) of macroblock row r in the frame buffer
    int r = 0;
    u32 dqs[4][3];
#pragma unroll
    for (int s = 0; s < 4; s++) dqs[s][0] = dqs[s][1] = dqs[s][2] = 0;
    u32 lv_plain = 0, lv_bpred = 0;             // loop-filter levels of the frame, per segment
    int sharp = 0; bool simple = false;
    s_tab[lane] = (u32x4){ (u32)(unsigned long)cfp, (u32)((unsigned long)cfp >> 32), 0u, 0u };
    // ---- prediction context (unfiltered): left columns, last pixels of the previous step's above lines, bottom lines of the
    // macroblocks finished one and two steps ago
    u32 lY[4] = { 0, 0, 0, 0 }, lU[2] = { 0, 0 }, lV[2] = { 0, 0 };
    int prevLastY = 0, prevLastU = 0, prevLastV = 0;
    u32 h1Y[4] = { 0, 0, 0, 0 }, h1U[2] = { 0, 0 }, h1V[2] = { 0, 0 };
    u32 h2Y[4] = { 0, 0, 0, 0 }, h2U[2] = { 0, 0 }, h2V[2] = { 0, 0 };
    // ---- loop-filter context (filtered, biased): the last four pixels of every row of the macroblock to the left; the first
    // twelve (chroma: four) of its bottom four rows; and what the lane below asks for: the bottom four rows of the
    // macroblock finished two steps ago
    u32 sY[16], pbY[4][3], hY[4][4], sU[8], sV[8], pbU[4], pbV[4], hU[4][2], hV[4][2];
#pragma unroll
    for (int i = 0; i < 16; i++) sY[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sU[i] = sV[i] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        pbY[j][0] = pbY[j][1] = pbY[j][2] = 0; hY[j][0] = hY[j][1] = hY[j][2] = hY[j][3] = 0;
        pbU[j] = pbV[j] = 0; hU[j][0] = hU[j][1] = hV[j][0] = hV[j][1] = 0;
    }
    // one step ahead, in plain registers (see vp8_recon_simt.hip)
    u32x4 nx_m0 = { 0, 0, 0, 0 }, nx_m1 = { 0, 0, 0, 0 }, nx_bm = { 0, 0, 0, 0 }, nx_y2a = { 0, 0, 0, 0 }, nx_y2b = { 0, 0, 0, 0 };
    u32 nx_aY[4] = { 0, 0, 0, 0 }, nx_ar = 0, nx_aU[2] = { 0, 0 }, nx_aV[2] = { 0, 0 };
    // the chroma half of the previous step's macroblock, finished at the top of the next iteration
    bool p_act = false, p_more = false, p_top = false, p_first = false, p_last = false, p_wb = false, p_hand = false, p_rb = false;
    g_u8p p_tp = (g_u8p)dummy, p_rasU = (g_u8p)dummy, p_rasV = (g_u8p)dummy;
    int p_uv_mode = 0, p_tlU = 0, p_tlV = 0, p_up = 0, p_lf = 0, p_c = 0;
    u32 p_aU[2] = { 0, 0 }, p_aV[2] = { 0, 0 }, p_jmc = 0, p_bY[4] = { 0, 0, 0, 0 };
    int p_lastU = 0, p_lastV = 0;
    Lim p_L = { mku(0), mku(0), mku(0), mku(0), one };
    Gates p_gvC = { mku(0), mku(0), mku(0), mku(0), false, false }, p_ghC = p_gvC;

    int q_n = 0;
    auto queue_phase = [&](const int ph, const u32 m8, const u32 dc_given) {
        wave_lds_sync();
        int n = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool b = (m8 >> i) & 1;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(b);
            const u32 at = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, (u32)n));
            if (b) s_queue[at] = (u32)lane | ((u32)i << 6) | (dc_given << 9);
            n += __builtin_popcountll(bal);
        }
        q_n = n;
        wave_lds_sync();
        const int R = (n + 63) >> 6;
#pragma unroll 1
        for (int rr = 0; rr < R; rr++) {
            const int idx = rr * 64 + lane;
            if (idx < n) {
                const u32 ent = s_queue[idx];
                const u32x4 tb = s_tab[ent & 63];
                g_cs16p cf = (g_cs16p)(((unsigned long)tb.y << 32) | tb.x) + (ph * 8 + (int)((ent >> 6) & 7)) * 16;
                __builtin_amdgcn_global_load_lds((g_cvp)cf, (lds_vp)(s_stage + rr * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((g_cvp)(cf + 8), (lds_vp)(s_stage + rr * 512 + 256), 16, 0, 0);
            }
        }
    };
    auto drain_phase = [&](const int ph, const int younger) {
        if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
        const int R = (q_n + 63) >> 6;
#pragma unroll 1
        for (int rr = 0; rr < R; rr++) {
            if (rr * 64 + lane < q_n) {
                const u32 ent = s_queue[rr * 64 + lane];
                const int owner = ent & 63, i = (ent >> 6) & 7;
                const bool given = (ent >> 9) & 1;
                const u32x4 tb = s_tab[owner];
                const u32 dq = ph < 2 ? tb.z : tb.w;
                u32x4 ca, cb;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ca), "=&v"(cb) : "v"(stage_lane + rr * 2048) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                int dc_in = 0;
                if (given) { const int blk = ph * 8 + i; dc_in = (short)(s_y2dc[owner * 8 + (blk >> 1)] >> (16 * (blk & 1))); }
                int res[16];
                dequant_idct(ca, cb, dq & 0xffff, dq >> 16, given, dc_in, res);
                u32 o[8];
#pragma unroll
                for (int q = 0; q < 8; q++) o[q] = ((u32)res[2 * q] & 0xffff) | ((u32)res[2 * q + 1] << 16);
                u32x4 *dst = (u32x4 *)(s_res + owner * 68 + i * 8);
                dst[0] = (u32x4){ o[0], o[1], o[2], o[3] };
                dst[1] = (u32x4){ o[4], o[5], o[6], o[7] };
            }
        }
        wave_lds_sync();
    };

    auto prepare_mb = [&](const u32x4 m0, const u32x4 m1, const u32x4 y2a, const u32x4 y2b, g_cs16p cf, u32 &jm, u32 &dcg) {
        const u32 w0 = m0.x, w1 = m0.y;
        const int y_mode = w0 & 0xff;
        const bool skip = (w0 >> 24) & VP8IR_MB_SKIP;
        const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
        const int seg = w1 & 3;
        const u32 s00 = dqs[0][0], s01 = dqs[0][1], s02 = dqs[0][2], s10 = dqs[1][0], s11 = dqs[1][1], s12 = dqs[1][2];
        const u32 s20 = dqs[2][0], s21 = dqs[2][1], s22 = dqs[2][2], s30 = dqs[3][0], s31 = dqs[3][1], s32 = dqs[3][2];
        const u32 dq0 = seg == 0 ? s00 : seg == 1 ? s10 : seg == 2 ? s20 : s30;
        const u32 dq1 = seg == 0 ? s01 : seg == 1 ? s11 : seg == 2 ? s21 : s31;
        const u32 dq2 = seg == 0 ? s02 : seg == 1 ? s12 : seg == 2 ? s22 : s32;
        const u32 e[6] = { m0.z, m0.w, m1.x, m1.y, m1.z, m1.w };
        u32 m = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const u32 ge1 = ((e[q] + 0x7f7f7f7fu) & 0x80808080u) >> 7;
            m |= (((ge1 * 0x00204081u) >> 21) & 0xfu) << (4 * q);
        }
        if (has_y2) m |= 0xffffu;
        if (skip) m = 0;
        jm = m;
        dcg = has_y2 && !skip;
        s_tab[lane] = (u32x4){ (u32)(unsigned long)cf, (u32)((unsigned long)cf >> 32), dq0, dq2 };
        if (__builtin_amdgcn_ballot_w64(dcg != 0) != 0) {
            if (dcg) {
                const u32 q[8] = { y2a.x, y2a.y, y2a.z, y2a.w, y2b.x, y2b.y, y2b.z, y2b.w };
                const int fdc = dq1 & 0xffff, fac = dq1 >> 16;
                int tt[16], dc[16];
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
                    dc[row * 4 + 0] = (a1 + b1 + 3) >> 3; dc[row * 4 + 1] = (c1 + d1 + 3) >> 3;
                    dc[row * 4 + 2] = (a1 - b1 + 3) >> 3; dc[row * 4 + 3] = (d1 - c1 + 3) >> 3;
                }
                u32 o[8];
#pragma unroll
                for (int q2 = 0; q2 < 8; q2++) o[q2] = ((u32)dc[2 * q2] & 0xffff) | ((u32)dc[2 * q2 + 1] << 16);
                u32x4 *dst = (u32x4 *)(s_y2dc + lane * 8);
                dst[0] = (u32x4){ o[0], o[1], o[2], o[3] };
                dst[1] = (u32x4){ o[4], o[5], o[6], o[7] };
            }
        }
    };

    int c = -2 * pos - 1, V = pos;
    STAMP_DECL
#pragma unroll 1
    for (int t = 0; t <= T; ++t) {
        STAMP(0)
        // ======================= tail of the previous step =======================
        // prefetches of the macroblock two ahead of the pointers (descriptor, Y2 block) and of the unfiltered line above it
        u32x4 pf_m0 = *(g_cu32x4p)(mbp + 32), pf_m1 = *(g_cu32x4p)(mbp + 36), pf_bm = *(g_cu32x4p)(mbp + 42);
        u32x4 pf_y2a = *(g_cu32x4p)(cfp + 2 * VP8IR_COEF_PER_MB + 384), pf_y2b = *(g_cu32x4p)(cfp + 2 * VP8IR_COEF_PER_MB + 392);
        u32 pf_aY[4], pf_ar, pf_aU[2], pf_aV[2];
        {
            const unsigned char *pa = (const unsigned char *)abp + 2 * VP8_TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; i++) pf_aY[i] = load_l2(pa + HO_Y_LINE + 4 * i);
            pf_ar = load_l2(pa + VP8_TILE_BYTES + HO_Y_LINE);
            pf_aU[0] = load_l2(pa + HO_U_LINE); pf_aU[1] = load_l2(pa + HO_U_LINE + 4);
            pf_aV[0] = load_l2(pa + HO_V_LINE); pf_aV[1] = load_l2(pa + HO_V_LINE + 4);
        }
        u32 n_jm = 0, n_dcg = 0;
        if (p_more) prepare_mb(nx_m0, nx_m1, nx_y2a, nx_y2b, cfp + VP8IR_COEF_PER_MB, n_jm, n_dcg);
        queue_phase(0, n_jm & 0xff, n_dcg);
        STAMP(1)
        // ---- chroma of the previous macroblock: U then V (an idle lane: garbage, into the dummy scratch).  The loop body
        // works on "U"; the two planes' state changes places at its end.
        u32 tU[4][2], tV[4][2];                 // filtered rows 4..7 of the macroblock above, from the lane above
#pragma unroll
        for (int j = 0; j < 4; j++) {
            tU[j][0] = from_lane_above(hU[j][0]); tU[j][1] = from_lane_above(hU[j][1]);
            tV[j][0] = from_lane_above(hV[j][0]); tV[j][1] = from_lane_above(hV[j][1]);
            // what the lane below will fetch at the start of the next step: the macroblock held from the step before (its last
            // four columns are fixed up below, once this step's left edge has revisited them)
            hU[j][0] = pbU[j]; hU[j][1] = sU[4 + j]; hV[j][0] = pbV[j]; hV[j][1] = sV[4 + j];
        }
        if (p_rb) {      // first lane of a strand: the rows come from the hand-over tile (wanted a plane's prediction from here)
            const unsigned char *pa = (const unsigned char *)p_tp - rowbytes;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                tU[j][0] = load_l2(pa + HO_U_ROWS + 8 * j); tU[j][1] = load_l2(pa + HO_U_ROWS + 8 * j + 4);
                tV[j][0] = load_l2(pa + HO_V_ROWS + 8 * j); tV[j][1] = load_l2(pa + HO_V_ROWS + 8 * j + 4);
            }
        }
        u32 bU[2] = { 0, 0 }, bV[2] = { 0, 0 };
        g_u8p q_ras = p_rasU, q_ras2 = p_rasV;
        u32 q_a0 = p_aU[0], q_a1 = p_aU[1], q_b0 = p_aV[0], q_b1 = p_aV[1];
        int q_tl = p_tlU, q_tl2 = p_tlV;
#pragma unroll 1
        for (int pl = 0; pl < 2; pl++) {
            const u32 aC0 = q_a0, aC1 = q_a1, lC0 = lU[0], lC1 = lU[1];
            int dcC = 128;
            if (p_up | p_lf) {
                const int shift = 2 + p_up + p_lf;
                const int s = (p_up ? sad4(aC0) + sad4(aC1) : 0) + (p_lf ? sad4(lC0) + sad4(lC1) : 0);
                dcC = (s + (1 << (shift - 1))) >> shift;
            }
            const u32 rmg = p_jmc >> (4 * pl);
            const u32 *rs = my_res + pl * 32;
            u32 bot[2] = { 0, 0 }, rc[2] = { 0, 0 };
            u32 o0[4][2], o1[4][2];
            u32x4 rr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) rr[k] = *(const u32x4 *)(rs + k * 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int bx = k & 1, byc = k >> 1;
                u32 p[4];
                mb_mode_pred(p_uv_mode, bx ? aC1 : aC0, byc ? lC1 : lC0, q_tl, dcC, p);
                u32 o[4] = { p[0], p[1], p[2], p[3] };
                const bool hasr = (rmg >> k) & 1;
                if (__builtin_amdgcn_ballot_w64(hasr) != 0) {
                    if (hasr) {
                        const u32x4 ra = rr[2 * k], rb = rr[2 * k + 1];
                        o[0] = add_clamp_pack(p[0], ra.x, ra.y); o[1] = add_clamp_pack(p[1], ra.z, ra.w);
                        o[2] = add_clamp_pack(p[2], rb.x, rb.y); o[3] = add_clamp_pack(p[3], rb.z, rb.w);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 4; jj++) { if (byc) o1[jj][bx] = o[jj]; else o0[jj][bx] = o[jj]; }
                if (byc) bot[bx] = o[3];
                if (bx) rc[byc] = right_column(o);
            }
            bU[0] = bot[0]; bU[1] = bot[1]; lU[0] = rc[0]; lU[1] = rc[1];
            STAMP(8)
            // ---- loop filter of the plane, block row by block row
            const int hoff = pl ? HO_V_ROWS : HO_U_ROWS, loff = pl ? HO_V_LINE : HO_U_LINE;
            if (p_hand) *(g_u32x2p)(p_tp + loff) = (u32x2){ bot[0], bot[1] };        // unfiltered bottom line: hand-over
            v2u PL[2][4], PH[2][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                PL[0][j] = col_lo(tU[j][0]); PH[0][j] = col_hi(tU[j][0]);
                PL[1][j] = col_lo(tU[j][1]); PH[1][j] = col_hi(tU[j][1]);
            }
            u32 s0[4] = { sU[0], sU[1], sU[2], sU[3] }, s1[4] = { sU[4], sU[5], sU[6], sU[7] }, d0[4][2], d1[4][2];
            lf_block_row<2>(o0, s0, PL, PH, true, p_gvC, p_ghC, p_L, d0);
            {   // rows 4..7 of the macroblock above: final
                g_u8p pa = (p_act && !p_top) ? q_ras - 4 * ysC + p_c * 8 : (g_u8p)dummy;
                const int st = (p_act && !p_top) ? ysC : 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *(g_u32x2p)(pa + j * st) = (u32x2){ d0[j][0] ^ VP8_LF_BIAS, d0[j][1] ^ VP8_LF_BIAS };
            }
            lf_block_row<2>(o1, s1, PL, PH, false, p_gvC, p_ghC, p_L, d1);
            {   // rows 0..3: the left neighbour's last dword and this macroblock's first
                g_u8p po = p_act ? q_ras + p_c * 8 - 4 : (g_u8p)dummy + 16;
                const int st = p_act ? ysC : 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *(g_u32x2up)(po + j * st) = (u32x2){ s0[j] ^ VP8_LF_BIAS, d1[j][0] ^ VP8_LF_BIAS };
            }
            u32 e[4][2];
#pragma unroll
            for (int j = 0; j < 4; j++) { e[j][0] = col_pack(PL[0][j], PH[0][j]); e[j][1] = col_pack(PL[1][j], PH[1][j]); }
            // the left neighbour's bottom rows are complete now (a lane that is idle, or first in its row, changed nothing)
#pragma unroll
            for (int j = 0; j < 4; j++) hU[j][1] = s1[j];
            if (p_act) {
                if (p_wb && !p_first) {       // nobody below takes them over: to the frame (last row) or the hand-over tile
                    g_u8p pb = p_last ? q_ras + 4 * ysC + (p_c - 1) * 8 : p_tp - VP8_TILE_BYTES + hoff;
                    const int st = p_last ? ysC : 8;
                    const u32 x = p_last ? VP8_LF_BIAS : 0;         // (the hand-over tile keeps the filter's biased form)
#pragma unroll
                    for (int j = 0; j < 4; j++) *(g_u32x2p)(pb + j * st) = (u32x2){ hU[j][0] ^ x, hU[j][1] ^ x };
                }
#pragma unroll
                for (int j = 0; j < 4; j++) { sU[j] = d1[j][1]; sU[4 + j] = e[j][1]; pbU[j] = e[j][0]; }
                if (p_c == cols - 1) {        // end of the row: nobody revisits the last dword
#pragma unroll
                    for (int j = 0; j < 4; j++) *(GLOBAL_AS u32 *)(q_ras + j * ysC + p_c * 8 + 4) = d1[j][1] ^ VP8_LF_BIAS;
                    if (p_wb) {
                        g_u8p pb = p_last ? q_ras + 4 * ysC + p_c * 8 : p_tp + hoff;
                        const int st = p_last ? ysC : 8;
                        const u32 x = p_last ? VP8_LF_BIAS : 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) *(g_u32x2p)(pb + j * st) = (u32x2){ e[j][0] ^ x, e[j][1] ^ x };
                    }
                }
            }
            STAMP(9)
            // ---- the planes change places
            SWAP_U32(bU[0], bV[0]) SWAP_U32(bU[1], bV[1]) SWAP_U32(lU[0], lV[0]) SWAP_U32(lU[1], lV[1])
            SWAP_U32(q_a0, q_b0) SWAP_U32(q_a1, q_b1)
            { const int t_ = q_tl; q_tl = q_tl2; q_tl2 = t_; }
            { g_u8p t_ = q_ras; q_ras = q_ras2; q_ras2 = t_; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                SWAP_U32(tU[j][0], tV[j][0]) SWAP_U32(tU[j][1], tV[j][1]) SWAP_U32(hU[j][0], hV[j][0]) SWAP_U32(hU[j][1], hV[j][1])
                SWAP_U32(pbU[j], pbV[j]) SWAP_U32(sU[j], sV[j]) SWAP_U32(sU[4 + j], sV[4 + j])
            }
        }
        prevLastU = p_lastU; prevLastV = p_lastV;
        if (p_act) { mbp += 16; cfp += VP8IR_COEF_PER_MB; tp += VP8_TILE_BYTES; abp += VP8_TILE_BYTES; }
        // ---- prediction history: what the lane below will ask for in one and in two steps
#pragma unroll
        for (int i = 0; i < 4; i++) { h2Y[i] = h1Y[i]; h1Y[i] = p_act ? p_bY[i] : h1Y[i]; }
#pragma unroll
        for (int i = 0; i < 2; i++) { h2U[i] = h1U[i]; h1U[i] = p_act ? bU[i] : h1U[i]; h2V[i] = h1V[i]; h1V[i] = p_act ? bV[i] : h1V[i]; }
        STAMP(2)

        // ======================= this step =======================
        if (++c == P) { c = 0; V += G; }
        u32 nY[4], nU[2], nV[2];
#pragma unroll
        for (int i = 0; i < 4; i++) nY[i] = from_lane_above(h2Y[i]);
        const u32 nAR = from_lane_above(h1Y[0]);
#pragma unroll
        for (int i = 0; i < 2; i++) { nU[i] = from_lane_above(h2U[i]); nV[i] = from_lane_above(h2V[i]); }
        // filtered rows 12..15 of the macroblock above; then this lane's own offer: the macroblock held from the step before
        u32 tY[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < 4; i++) tY[j][i] = from_lane_above(hY[j][i]);
            hY[j][0] = pbY[j][0]; hY[j][1] = pbY[j][1]; hY[j][2] = pbY[j][2]; hY[j][3] = sY[12 + j];
        }

        const bool act = t < T && c >= 0 && c < cols && V < Vmax;
        const bool late = act && !p_more;
        u32 jm = n_jm, dc_given = n_dcg;
        u32 cur_w0 = nx_m0.x, cur_w1 = nx_m0.y;
        u32x4 bm = nx_bm;
        u32 rbaY[4] = { nx_aY[0], nx_aY[1], nx_aY[2], nx_aY[3] }, rbAR = nx_ar, rbaU[2] = { nx_aU[0], nx_aU[1] }, rbaV[2] = { nx_aV[0], nx_aV[1] };
        if (late) {
            // ---- new macroblock row (c == 0): which frame, which row; pointers, quantisers, filter levels; its first macroblock
            const int j = V / rows;
            r = V - j * rows;
            const DevJob *job = jobs + (strand + j * nstrands);
            const vp8ir_frame_hdr &h = job->hdr;
            const int nseg = h.segmentation_enabled ? 4 : 1;
            for (int s = 0; s < 4; s++) {
                u32 d[3];
                if (s < nseg) segment_dequant(h, s, d);
                else { d[0] = dqs[0][0]; d[1] = dqs[0][1]; d[2] = dqs[0][2]; }
                dqs[s][0] = d[0]; dqs[s][1] = d[1]; dqs[s][2] = d[2];
            }
            frame_levels(h, lv_plain, lv_bpred);
            sharp = h.sharpness_level; simple = h.filter_type == 1;
            mbp = (g_cu32p)(job->mbs + (long)r * cols);
            cfp = (g_cs16p)(job->coef + (long)r * cols * VP8IR_COEF_PER_MB);
            tp = (g_u8p)(job->tile + (long)r * rowbytes);
            abp = r == 0 ? (g_cu8p)tp : (g_cu8p)tp - rowbytes;
            rasY = (g_u8p)(job->dst + g.y_off + (long)r * 16 * ysY);
            rasU = (g_u8p)(job->dst + g.u_off + (long)r * 8 * ysC);
            rasV = (g_u8p)(job->dst + g.v_off + (long)r * 8 * ysC);
            lY[0] = lY[1] = lY[2] = lY[3] = 0x81818181u;    // left border 129 (setupintrarecon.c:15-32)
            lU[0] = lU[1] = lV[0] = lV[1] = 0x81818181u;
            const u32x4 m0 = *(g_cu32x4p)mbp, m1 = *(g_cu32x4p)(mbp + 4), b0 = *(g_cu32x4p)(mbp + 10);
            const u32x4 y2a = *(g_cu32x4p)(cfp + 384), y2b = *(g_cu32x4p)(cfp + 392);
            if (pos == 0) {
                const unsigned char *pa = (const unsigned char *)abp;
#pragma unroll
                for (int i = 0; i < 4; i++) rbaY[i] = load_l2(pa + HO_Y_LINE + 4 * i);
                rbAR = load_l2(pa + VP8_TILE_BYTES + HO_Y_LINE);
                rbaU[0] = load_l2(pa + HO_U_LINE); rbaU[1] = load_l2(pa + HO_U_LINE + 4);
                rbaV[0] = load_l2(pa + HO_V_LINE); rbaV[1] = load_l2(pa + HO_V_LINE + 4);
            }
            pf_m0 = *(g_cu32x4p)(mbp + 16); pf_m1 = *(g_cu32x4p)(mbp + 20); pf_bm = *(g_cu32x4p)(mbp + 26);
            pf_y2a = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + 384); pf_y2b = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + 392);
            if (pos == 0) {
                const unsigned char *pa = (const unsigned char *)abp + VP8_TILE_BYTES;
#pragma unroll
                for (int i = 0; i < 4; i++) pf_aY[i] = load_l2(pa + HO_Y_LINE + 4 * i);
                pf_ar = load_l2(pa + VP8_TILE_BYTES + HO_Y_LINE);
                pf_aU[0] = load_l2(pa + HO_U_LINE); pf_aU[1] = load_l2(pa + HO_U_LINE + 4);
                pf_aV[0] = load_l2(pa + HO_V_LINE); pf_aV[1] = load_l2(pa + HO_V_LINE + 4);
            }
            // consumed here, so that no pending load leaves the branch
            u32x4 m0s = m0, m1s = m1, b0s = b0, y2as = y2a, y2bs = y2b;
            asm volatile("" : "+v"(m0s), "+v"(m1s), "+v"(b0s), "+v"(y2as), "+v"(y2bs));
            asm volatile("" : "+v"(rbaY[0]), "+v"(rbaY[1]), "+v"(rbaY[2]), "+v"(rbaY[3]), "+v"(rbAR), "+v"(rbaU[0]), "+v"(rbaU[1]), "+v"(rbaV[0]), "+v"(rbaV[1]));
            asm volatile("" : "+v"(pf_m0), "+v"(pf_m1), "+v"(pf_bm), "+v"(pf_y2a), "+v"(pf_y2b));
            asm volatile("" : "+v"(pf_aY[0]), "+v"(pf_aY[1]), "+v"(pf_aY[2]), "+v"(pf_aY[3]), "+v"(pf_ar), "+v"(pf_aU[0]), "+v"(pf_aU[1]), "+v"(pf_aV[0]), "+v"(pf_aV[1]));
            cur_w0 = m0s.x; cur_w1 = m0s.y; bm = b0s;
            prepare_mb(m0s, m1s, y2as, y2bs, cfp, jm, dc_given);
        }
        if (!act) { jm = 0; dc_given = 0; }
        const bool top = r == 0;
        const bool more = act && c + 1 < cols;
        const bool readback = act && pos == 0 && !top;
        if (readback) {      // first lane of a strand: filtered rows 12..15 of the macroblock above from the hand-over tile (wanted
                             // by the first block row's top edge, a block row of prediction from here)
            const unsigned char *pa = (const unsigned char *)tp - rowbytes;
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) tY[j][i] = load_l2(pa + HO_Y_ROWS + 16 * j + 4 * i);
            }
        }
        // ---- macroblock descriptor
        const int y_mode = cur_w0 & 0xff, uv_mode = (cur_w0 >> 8) & 0xff;
        const bool bpred = y_mode == VP8IR_B_PRED;
        // ---- its loop-filter parameters (vp8_loop_filter_frame, loopfilter.c:245-299)
        const int level = act ? (int)(((bpred ? lv_bpred : lv_plain) >> (8 * (cur_w1 & 3))) & 0xff) : 0;
        const Lim L = mb_limits(sharp, level, 0, one);
        const bool on = level != 0;
        const bool skip_lf = !bpred && y_mode != VP8IR_SPLITMV && ((cur_w0 >> 24) & VP8IR_MB_SKIP);
        const bool mbv = on && c > 0, inner = on && !skip_lf, mbh = on && !top;
        const bool any_normal = __builtin_amdgcn_ballot_w64(on && !simple) != 0;
        const bool any_simple = __builtin_amdgcn_ballot_w64(on && simple) != 0;
        auto gate = [](bool b) { return mku(b ? 0xffff : 0); };
        const Gates gvY = { gate(mbv && !simple), gate(inner && !simple), gate(mbv && simple), gate(inner && simple), any_normal, any_simple };
        const Gates ghY = { gate(mbh && !simple), gate(inner && !simple), gate(mbh && simple), gate(inner && simple), any_normal, any_simple };
        // the simple filter leaves chroma alone (loopfilter.c:283-299)
        const Gates gvC = { gvY.mb, gvY.inner, mku(0), mku(0), any_normal, false };
        const Gates ghC = { ghY.mb, ghY.inner, mku(0), mku(0), any_normal, false };
        const bool last_col = act && c == cols - 1, last_row = r == rows - 1;
        // bottom rows that no lane below takes over are written here: to the frame (last row), or to the hand-over tile
        const bool write_bottom = pos == G - 1 || last_row;
        const bool hand = act && pos == G - 1 && !last_row;
        // ---- unfiltered line above (127 above the frame; vp8_setup_intra_recon)
        u32 aY[4], arY, aU[2], aV[2];
        if (top) {
            aY[0] = aY[1] = aY[2] = aY[3] = arY = 0x7f7f7f7fu;
            aU[0] = aU[1] = aV[0] = aV[1] = 0x7f7f7f7fu;
        } else if (pos == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) aY[i] = rbaY[i];
            arY = rbAR;
            aU[0] = rbaU[0]; aU[1] = rbaU[1]; aV[0] = rbaV[0]; aV[1] = rbaV[1];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) aY[i] = nY[i];
            arY = nAR;
            aU[0] = nU[0]; aU[1] = nU[1]; aV[0] = nV[0]; aV[1] = nV[1];
        }
        if (!top && c == cols - 1) arY = splat(aY[3] >> 24);
        const int tlY = top ? 127 : (c == 0 ? 129 : prevLastY);
        const int tlU = top ? 127 : (c == 0 ? 129 : prevLastU);
        const int tlV = top ? 127 : (c == 0 ? 129 : prevLastV);
        const int up = !top, lf = c > 0;
        int dcY = 128;
        if (up | lf) {
            const int shift = 3 + up + lf;
            const int s = (up ? sad4(aY[0]) + sad4(aY[1]) + sad4(aY[2]) + sad4(aY[3]) : 0)
                        + (lf ? sad4(lY[0]) + sad4(lY[1]) + sad4(lY[2]) + sad4(lY[3]) : 0);
            dcY = (s + (1 << (shift - 1))) >> shift;
        }
        u32 abv[4] = { aY[0], aY[1], aY[2], aY[3] };
        int tlrow = tlY;
        u32 nl[4] = { 0, 0, 0, 0 };
        const g_u8p tpe = act ? tp : (g_u8p)dummy;
        // ---- the filter's rows above the first block row: rows 12..15 of the macroblock above, as column pairs
        v2u PL[4][4], PH[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const u32 D = tY[j][x];
                PL[x][j] = col_lo(D); PH[x][j] = col_hi(D);
            }
        }
        u32 sfix[4] = { 0, 0, 0, 0 };             // the left neighbour's last dword in the rows above the current block row, fixed up
        // where the rows above the current block row go: first the macroblock above (aligned), then this one (shifted left by 4)
        g_u8p prow = (act && !top) ? rasY - 4 * ysY + c * 16 : (g_u8p)dummy + 16;
        int pstride = (act && !top) ? ysY : 0;
        STAMP(3)

        // ======================= luma: two transform phases of two block rows =======================
        drain_phase(0, 8);
        if (__builtin_amdgcn_ballot_w64(late) != 0) {
            queue_phase(0, late ? jm & 0xff : 0, dc_given);
            drain_phase(0, 0);
        }
        STAMP(4)
#pragma unroll 1
        for (int ph = 0; ph < 2; ph++) {
            queue_phase(ph + 1, (jm >> (8 * (ph + 1))) & 0xff, ph == 0 ? dc_given : 0);
#pragma unroll 1
            for (int by = 2 * ph; by < 2 * ph + 2; by++) {
                const u32 lcur = lY[0];
                const u32 bmw = by == 0 ? bm.x : by == 1 ? bm.y : by == 2 ? bm.z : bm.w;
                const u32 rmg = jm >> (by * 4);
                const u32 *rs = my_res + (by & 1) * 32;
                u32 left = lcur;
                int tl = tlrow;
                u32 orow[4][4];
                u32x4 rr[8];
#pragma unroll
                for (int k = 0; k < 8; k++) rr[k] = *(const u32x4 *)(rs + k * 4);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 p[4];
                    if (bpred) bpred4x4((bmw >> (8 * k)) & 0xff, abv[k], k < 3 ? abv[k + 1] : arY, left, tl, p);
                    else mb_mode_pred(y_mode, aY[k], lcur, tlY, dcY, p);
                    u32 o[4] = { p[0], p[1], p[2], p[3] };
                    const bool hasr = (rmg >> k) & 1;
                    if (__builtin_amdgcn_ballot_w64(hasr) != 0) {
                        if (hasr) {
                            const u32x4 ra = rr[2 * k], rb = rr[2 * k + 1];
                            o[0] = add_clamp_pack(p[0], ra.x, ra.y); o[1] = add_clamp_pack(p[1], ra.z, ra.w);
                            o[2] = add_clamp_pack(p[2], rb.x, rb.y); o[3] = add_clamp_pack(p[3], rb.z, rb.w);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) orow[jj][k] = o[jj];
                    tl = abv[k] >> 24;
                    abv[k] = o[3];
                    left = right_column(o);
                }
                tlrow = lcur >> 24;
                lY[0] = lY[1]; lY[1] = lY[2]; lY[2] = lY[3];
                nl[0] = nl[1]; nl[1] = nl[2]; nl[2] = nl[3]; nl[3] = left;
                STAMP(10)
                // ---- loop filter: the vertical edges of this block row, the horizontal edge above it; the rows above are final
                u32 sb[4] = { sY[0], sY[1], sY[2], sY[3] }, d[4][4];
                lf_block_row<4>(orow, sb, PL, PH, by == 0, gvY, ghY, L, d);
                STAMP(11)
                const bool first = by == 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32x4 v = first ? (u32x4){ d[j][0], d[j][1], d[j][2], d[j][3] } : (u32x4){ sfix[j], d[j][0], d[j][1], d[j][2] };
                    *(g_u32x4up)(prow + j * pstride) = v ^ VP8_LF_BIAS;
                }
                // rotate: the next block row's left context comes to the front, this macroblock's last dwords queue up behind
#pragma unroll
                for (int i = 0; i < 12; i++) sY[i] = sY[i + 4];
#pragma unroll
                for (int j = 0; j < 4; j++) { sY[12 + j] = d[j][3]; sfix[j] = sb[j]; }
                prow = act ? (first ? rasY + c * 16 - 4 : prow + 4 * ysY) : (g_u8p)dummy + 16;
                pstride = act ? ysY : 0;
                STAMP(12)
            }
            STAMP(5)
            drain_phase(ph + 1, 8);
            STAMP(6)
        }
        // ---- the bottom four rows stay (the lane below finishes them); the left neighbour's are complete now
        {
            u32 e[4][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int x = 0; x < 4; x++) e[j][x] = col_pack(PL[x][j], PH[x][j]);
                hY[j][3] = sfix[j];
            }
            // (sY: [junk of the first rotation, rows 0..3, 4..7, 8..11] -> rows 0..15)
            u32 ns[16];
#pragma unroll
            for (int i = 0; i < 12; i++) ns[i] = sY[i + 4];
#pragma unroll
            for (int j = 0; j < 4; j++) ns[12 + j] = e[j][3];
            if (act) {
                if (hand) *(g_u32x4p)(tp + HO_Y_LINE) = (u32x4){ abv[0], abv[1], abv[2], abv[3] };     // unfiltered bottom line
                if (write_bottom && c > 0) {
                    g_u8p pb = last_row ? rasY + 12 * ysY + (c - 1) * 16 : tp - VP8_TILE_BYTES + HO_Y_ROWS;
                    const int st = last_row ? ysY : 16;
                    const u32 x = last_row ? VP8_LF_BIAS : 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) *(g_u32x4p)(pb + j * st) = (u32x4){ hY[j][0], hY[j][1], hY[j][2], hY[j][3] } ^ x;
                }
#pragma unroll
                for (int i = 0; i < 16; i++) sY[i] = ns[i];
#pragma unroll
                for (int j = 0; j < 4; j++) { pbY[j][0] = e[j][0]; pbY[j][1] = e[j][1]; pbY[j][2] = e[j][2]; }
                if (last_col) {
#pragma unroll
                    for (int y = 0; y < 12; y++) *(GLOBAL_AS u32 *)(rasY + y * ysY + c * 16 + 12) = ns[y] ^ VP8_LF_BIAS;
                    if (write_bottom) {
                        g_u8p pb = last_row ? rasY + 12 * ysY + c * 16 : tp + HO_Y_ROWS;
                        const int st = last_row ? ysY : 16;
                        const u32 x = last_row ? VP8_LF_BIAS : 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) *(g_u32x4p)(pb + j * st) = (u32x4){ e[j][0], e[j][1], e[j][2], e[j][3] } ^ x;
                    }
                }
            }
        }
        STAMP(13)
        // ---- hand the chroma half over to the next iteration; the prefetches become plain registers
#pragma unroll
        for (int i = 0; i < 4; i++) { lY[i] = nl[i]; p_bY[i] = abv[i]; }
        prevLastY = aY[3] >> 24; p_lastU = aU[1] >> 24; p_lastV = aV[1] >> 24;
        p_act = act; p_more = more; p_tp = tpe; p_uv_mode = uv_mode; p_tlU = tlU; p_tlV = tlV; p_up = up; p_lf = lf;
        p_aU[0] = aU[0]; p_aU[1] = aU[1]; p_aV[0] = aV[0]; p_aV[1] = aV[1]; p_jmc = jm >> 16;
        p_top = top; p_first = c == 0; p_last = last_row; p_wb = write_bottom; p_hand = hand; p_rb = readback; p_c = c;
        p_rasU = rasU; p_rasV = rasV; p_L = L; p_gvC = gvC; p_ghC = ghC;
        nx_m0 = pf_m0; nx_m1 = pf_m1; nx_bm = pf_bm; nx_y2a = pf_y2a; nx_y2b = pf_y2b;
        asm volatile("" : "+v"(nx_m0), "+v"(nx_m1), "+v"(nx_bm), "+v"(nx_y2a), "+v"(nx_y2b));
#pragma unroll
        for (int i = 0; i < 4; i++) nx_aY[i] = pf_aY[i];
        nx_ar = pf_ar; nx_aU[0] = pf_aU[0]; nx_aU[1] = pf_aU[1]; nx_aV[0] = pf_aV[0]; nx_aV[1] = pf_aV[1];
        asm volatile("" : "+v"(nx_aY[0]), "+v"(nx_aY[1]), "+v"(nx_aY[2]), "+v"(nx_aY[3]), "+v"(nx_ar), "+v"(nx_aU[0]), "+v"(nx_aU[1]), "+v"(nx_aV[0]), "+v"(nx_aV[1]));
        STAMP(7)
    }
    STAMP_FLUSH(vp8_stamps_recon)
}
