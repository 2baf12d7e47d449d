// VP8 key-frame reconstruction, "one macroblock row per LANE" formulation for gfx950.
//
// Same job as vp8_recon.hip restricted to intra frames (decode_mb_row / decode_macroblock,
// vp8/decoder/decodframe.c:112-436, and what it reaches through RTCD: reconintra.c, reconintra4x4.c, and the add
// of dequantize.c / idctllm.c / idct_blk.c), organised around what was measured on MI355X: the path is bound by
// VALU issue (a wave instruction costs a SIMD about four cycles however many lanes are live, and a second wave on
// the SIMD does not make the packed / byte-permute instructions this code is made of any faster), and a macroblock
// offers at most 16..32 lanes of parallel work once the 4x4 intra chain is respected.  So the lanes of a wave are
// not spent inside a macroblock at all:
//
//   * lane p of a wave owns macroblock ROWS p, p+G, p+2G, ... of a strand of frames (G = lanes per strand, a power
//     of two <= 64; a wave carries 64/G strands) and walks each row left to right, one whole macroblock per step,
//     as straight per-lane code (the reference's C, restated per lane);
//   * lane p runs two macroblocks behind lane p-1 -- the intra dependency (left, above, above-right) is then
//     satisfied by construction, with no flags, no polling and no barriers: a step is one pass of all 64 lanes
//     over 64 different macroblocks of the classic 2-D wavefront;
//   * the unfiltered pixels above a macroblock are the bottom line of the macroblock the lane above finished two
//     steps ago: they travel by DPP wave shift (v_mov_b32 wave_shr:1), not through memory.  The first lane of a
//     strand has its predecessor row on the LAST lane of the strand, G rows of work earlier; it reads that line
//     back from the frame (L2-coherent loads), which the step period P >= 2G+2 guarantees was written at least
//     three steps before;
//   * the 4x4 intra chain of B_PRED macroblocks runs inside the lane on packed bytes: the edge vector's 3-tap and
//     2-tap smoothings are v_lerp_u8 on four pixels at a time, the ten predictors are byte shuffles (v_perm_b32 /
//     v_alignbyte_b32) of those; divergence between lanes costs the union of the modes present, not a serial
//     chain per macroblock;
//   * the residual transform is the one part that is NOT done lane-per-macroblock: a block's dequantisation + IDCT costs
//     the wave the same 220 instructions whether one lane needs it or all 64 do, and on real streams only a third of
//     the blocks have coefficients at all.  So three times per step (luma blocks 0-7, 8-15, chroma) the lanes queue
//     the blocks of THEIR macroblocks that have coefficients in LDS (v_mbcnt over the ballot of each block position),
//     then all 64 lanes drain the queue, one block per lane and round, whoever it belongs to: the coefficients come
//     straight from the owner's IR into LDS (global_load_lds_dwordx4: no registers held while they are in flight,
//     requested a whole prediction phase before they are needed), the residual goes into the owner's LDS slot.
//     The owner then adds it to its prediction: v_pk_add_i16 + v_sat_pk_u8_i16, 7 instructions per pixel row (the
//     clamp of vp8_dequant_idct_add_c is the pack instruction's saturation).  A DC-only block goes the same way (the
//     full transform of a DC-only block is bit-identical to vp8_dc_only_idct_add_c's shortcut);
//   * inter frames never come here (the wave-per-row kernels of vp8_recon.hip decode them).
//
// Integer only (u8 pixels, i16 residuals); no MFMA by design.
#include "vp8_common.hip.h"
#include <stddef.h>

#include "vp8_simt_prims.hip.h"


// grid = waves (one wave per block); lgG = log2(lanes per strand); P = steps per row period, >= max(cols, 2G+2);
// nstrands = total strands of the launch: strand q reconstructs jobs q, q+nstrands, ...
// The frame goes to the job's macroblock-tiled scratch (DevJob::tile, VP8_TILE_BYTES per macroblock: 16 luma rows of
// 16 B, 8 U rows of 8 B, 8 V rows of 8 B), so that every lane writes whole 128-byte lines; the loop filter (or
// vp8_detile_kernel) takes it from there.  Every job must be a key frame.  `dummy`: VP8_TILE_BYTES of scratch nobody reads.
//
// Memory instructions and s_waitcnt.  hipcc places the waits for register loads, and it can only COUNT
// (s_waitcnt vmcnt(N): "all but the N youngest") where loads and stores are issued unconditionally, in a fixed order, and
// are consumed inside the loop iteration that issued them; anything else degrades to vmcnt(0), a drain of the wave's whole
// memory queue, stores included -- thousands of cycles per step with one wave per SIMD.  Hence:
//   * every global load and store of the step loop is issued by all 64 lanes every step (an idle lane reads its stale --
//     valid -- addresses and writes the dummy tile); the one exception, the first macroblock of a row, drains its own
//     loads inside its (rare) branch;
//   * the loop is rotated: an iteration starts with the chroma half of the PREVIOUS step's macroblock, so that the
//     coefficient fetches of the next macroblock's first transform phase, issued just before it, are consumed in the same
//     iteration; prefetched descriptor words are copied to plain registers at the end of the iteration that loaded them;
//   * the transform's coefficients travel by LDS-DMA and are read back with inline-asm ds_read behind an explicit counted
//     wait (hipcc would put vmcnt(0) in front of any read of an LDS-DMA target it can see).
extern "C" __global__ void __launch_bounds__(64)
vp8_recon_simt_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy)
{
    // residuals of the eight blocks of the current phase, per owner lane: 8 x 32 B + 16 B of padding (a 272-byte
    // stride keeps the owners' ds_read_b128 of one block position conflict-free)
    __shared__ __attribute__((aligned(16))) u32 s_res[64 * 68];
    // coefficients of the queued blocks: [round][half][lane] 16 B each, written by LDS-DMA
    __shared__ __attribute__((aligned(16))) u32 s_stage[8 * 2 * 64 * 4];
    __shared__ u32 s_queue[512];                                   // owner lane | block in phase << 6 | DC given << 9
    __shared__ __attribute__((aligned(16))) u32x4 s_tab[64];       // per owner: coefficient pointer (lo, hi), quantisers y1, uv
    __shared__ __attribute__((aligned(16))) u32 s_y2dc[64 * 8];    // per owner: the sixteen luma DCs out of the Y2 block's WHT
    const int lane = threadIdx.x;
    const int G = 1 << lgG;
    const int pos = lane & (G - 1);
    const int spw = 64 >> lgG;
    const int strand = blockIdx.x * spw + (lane >> lgG);
    const int cols = g.mb_cols, rows = g.mb_rows;
    const long rowbytes = (long)cols * VP8_TILE_BYTES;
    const int myjobs = strand < njobs ? (njobs - strand + nstrands - 1) / nstrands : 0;
    const int Vmax = myjobs * rows;
    const int wavejobs = (njobs - (int)blockIdx.x * spw + nstrands - 1) / nstrands;     // first strand: the most jobs
    const int T = ((wavejobs * rows + G - 1) >> lgG) * P + 2 * (G - 1);
    u32 *const my_res = s_res + lane * 68;
    const u32 stage_lane = (u32)(unsigned long)(lds_vp)s_stage + lane * 16;             // LDS byte address of this lane's 16 B of round 0, half 0

    // ---- per-lane row state; the pointers are valid addresses at all times (see above)
    g_cu32p mbp = (g_cu32p)jobs[0].mbs;         // descriptor of the current macroblock
    g_cs16p cfp = (g_cs16p)jobs[0].coef;        // its coefficients
    g_u8p tp = (g_u8p)dummy;                    // its tile
    g_cu8p abp = (g_cu8p)dummy;                 // the tile above it (read back by the first lane of a strand)
    int r = 0;
    u32 dqs[4][3];
#pragma unroll
    for (int s = 0; s < 4; s++) dqs[s][0] = dqs[s][1] = dqs[s][2] = 0;
    s_tab[lane] = (u32x4){ (u32)(unsigned long)cfp, (u32)((unsigned long)cfp >> 32), 0u, 0u };
    // unfiltered context: left columns (top pixel in byte 0), last pixels of the previous step's above lines
    u32 lY[4] = { 0, 0, 0, 0 }, lU[2] = { 0, 0 }, lV[2] = { 0, 0 };
    int prevLastY = 0, prevLastU = 0, prevLastV = 0;
    // bottom lines of the macroblocks finished one and two steps ago (what the lane below asks for)
    u32 h1Y[4] = { 0, 0, 0, 0 }, h1U[2] = { 0, 0 }, h1V[2] = { 0, 0 };
    u32 h2Y[4] = { 0, 0, 0, 0 }, h2U[2] = { 0, 0 }, h2V[2] = { 0, 0 };
    // one step ahead, in plain registers: descriptor words 0..7 (modes, segment, eobs of blocks 0..23), sub-block modes and Y2
    // block of the macroblock after the current one, and the line above it as the first lane of a strand reads it back
    u32x4 nx_m0 = { 0, 0, 0, 0 }, nx_m1 = { 0, 0, 0, 0 }, nx_bm = { 0, 0, 0, 0 }, nx_y2a = { 0, 0, 0, 0 }, nx_y2b = { 0, 0, 0, 0 };
    u32 nx_aY[4] = { 0, 0, 0, 0 }, nx_ar = 0, nx_aU[2] = { 0, 0 }, nx_aV[2] = { 0, 0 };
    // the chroma half of the previous step's macroblock, finished at the top of the next iteration
    bool p_act = false, p_more = false;
    g_u8p p_tpe = (g_u8p)dummy;
    int p_uv_mode = 0, p_tlU = 0, p_tlV = 0, p_up = 0, p_lf = 0;
    u32 p_aU[2] = { 0, 0 }, p_aV[2] = { 0, 0 }, p_jmc = 0, p_bY[4] = { 0, 0, 0, 0 };
    int p_lastU = 0, p_lastV = 0;

    // ---- the cooperative part: residuals of the eight blocks of phase `ph` (0, 1: luma blocks 0-7, 8-15; 2: chroma) of
    // every lane's macroblock.  Two halves, so that a phase's coefficients are on their way while the owners still
    // predict the phase before:
    //   queue_phase: the lanes queue the blocks of their macroblock that have coefficients (`m8`; `dc_given`: the DC comes
    //                out of the Y2 block); every queued block's coefficients are requested (LDS-DMA into s_stage);
    //   drain_phase: all lanes transform queued blocks, one per lane and round, into the owners' slots of s_res.
    //                `younger`: a LOWER bound of the memory instructions issued since queue_phase (they may stay in flight).
    int q_n = 0;                                                     // blocks queued (wave-uniform)
    auto queue_phase = [&](const int ph, const u32 m8, const u32 dc_given) {
        wave_lds_sync();                                             // the previous phase's queue has been drained
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
        // the coefficients have landed once at most `younger` memory instructions are outstanding (in-order return)
        if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();                                             // ... and the owners are done reading the previous phase's residuals
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

    // ---- what the transform needs to know about a macroblock, from its descriptor words 0..7 (modes, segment, eobs of blocks
    // 0..23) and its Y2 block: which blocks have a residual (`jm`), whether the luma DCs come out of the Y2 block (`dcg`);
    // the lane's entry of s_tab (coefficient pointer, quantisers) and, with a Y2 block, of s_y2dc.
    auto prepare_mb = [&](const u32x4 m0, const u32x4 m1, const u32x4 y2a, const u32x4 y2b, g_cs16p cf, u32 &jm, u32 &dcg) {
        const u32 w0 = m0.x, w1 = m0.y;
        const int y_mode = w0 & 0xff;
        const bool skip = (w0 >> 24) & VP8IR_MB_SKIP;
        const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
        const int seg = w1 & 3;
        // (copies first: a select between by-reference captures would become a dynamic index into the closure, in scratch)
        const u32 s00 = dqs[0][0], s01 = dqs[0][1], s02 = dqs[0][2], s10 = dqs[1][0], s11 = dqs[1][1], s12 = dqs[1][2];
        const u32 s20 = dqs[2][0], s21 = dqs[2][1], s22 = dqs[2][2], s30 = dqs[3][0], s31 = dqs[3][1], s32 = dqs[3][2];
        const u32 dq0 = seg == 0 ? s00 : seg == 1 ? s10 : seg == 2 ? s20 : s30;
        const u32 dq1 = seg == 0 ? s01 : seg == 1 ? s11 : seg == 2 ? s21 : s31;
        const u32 dq2 = seg == 0 ? s02 : seg == 1 ? s12 : seg == 2 ? s22 : s32;
        // eobs (detokenize.c:363), a byte per block, 0..16: which blocks have a token at all.  A luma block of a
        // macroblock with Y2 always has its DC (idct_blk.c:20-44, decodframe.c:262-296).
        const u32 e[6] = { m0.z, m0.w, m1.x, m1.y, m1.z, m1.w };
        u32 m = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const u32 ge1 = ((e[q] + 0x7f7f7f7fu) & 0x80808080u) >> 7;        // bit 0 of each byte: eob >= 1
            m |= (((ge1 * 0x00204081u) >> 21) & 0xfu) << (4 * q);
        }
        if (has_y2) m |= 0xffffu;
        if (skip) m = 0;
        jm = m;
        dcg = has_y2 && !skip;
        s_tab[lane] = (u32x4){ (u32)(unsigned long)cf, (u32)((unsigned long)cf >> 32), dq0, dq2 };
        // Y2: vp8_dequantize_b + vp8_short_inv_walsh4x4_c (idctllm.c:140-192) -> the 16 luma DCs.  With nothing but a DC
        // coefficient the full transform gives what vp8_short_inv_walsh4x4_1_c gives (decodframe.c:282-285).
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
        // the macroblock of THIS step, where the lane stays in its row: tables, and its luma blocks 0-7 queued (coefficients
        // requested) before the previous macroblock's chroma is predicted
        // First of all the prefetches of the macroblock after THAT one (two ahead of the pointers, which still stand on the previous
        // step's macroblock) and of the line above it; a lane at the end of its row, or idle, fetches whatever follows: never used.
        // (Issued here, a whole chroma prediction before the loops of the luma part: hipcc drains the memory queue in front of
        // a loop in which it sees a register with a load pending, and it sees one -- a copy of undefined tuple halves.)
        u32x4 pf_m0 = *(g_cu32x4p)(mbp + 32), pf_m1 = *(g_cu32x4p)(mbp + 36), pf_bm = *(g_cu32x4p)(mbp + 42);
        u32x4 pf_y2a = *(g_cu32x4p)(cfp + 2 * VP8IR_COEF_PER_MB + 384), pf_y2b = *(g_cu32x4p)(cfp + 2 * VP8IR_COEF_PER_MB + 392);
        u32 pf_aY[4], pf_ar, pf_aU[2], pf_aV[2];
        {
            const unsigned char *pa = (const unsigned char *)abp + 2 * VP8_TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; i++) pf_aY[i] = load_l2(pa + 15 * 16 + 4 * i);
            pf_ar = load_l2(pa + VP8_TILE_BYTES + 15 * 16);
            pf_aU[0] = load_l2(pa + 256 + 56); pf_aU[1] = load_l2(pa + 256 + 60);
            pf_aV[0] = load_l2(pa + 320 + 56); pf_aV[1] = load_l2(pa + 320 + 60);
        }
        u32 n_jm = 0, n_dcg = 0;
        if (p_more) prepare_mb(nx_m0, nx_m1, nx_y2a, nx_y2b, cfp + VP8IR_COEF_PER_MB, n_jm, n_dcg);
        queue_phase(0, n_jm & 0xff, n_dcg);
        STAMP(1)
        // ---- chroma of the previous macroblock: U then V, four 4x4 blocks each (an idle lane: garbage, into the dummy tile)
        u32 bU[2], bV[2];
#pragma unroll 1
        for (int pl = 0; pl < 2; pl++) {
            const u32 aC0 = pl ? p_aV[0] : p_aU[0], aC1 = pl ? p_aV[1] : p_aU[1];
            const u32 lC0 = pl ? lV[0] : lU[0], lC1 = pl ? lV[1] : lU[1];
            const int tlC = pl ? p_tlV : p_tlU;
            g_u8p dC = p_tpe + (pl ? 320 : 256);
            int dcC = 128;
            if (p_up | p_lf) {
                const int shift = 2 + p_up + p_lf;
                const int s = (p_up ? sad4(aC0) + sad4(aC1) : 0) + (p_lf ? sad4(lC0) + sad4(lC1) : 0);
                dcC = (s + (1 << (shift - 1))) >> shift;
            }
            const u32 rmg = p_jmc >> (4 * pl);
            const u32 *rs = my_res + pl * 32;
            u32 bot[2] = { 0, 0 }, rc[2] = { 0, 0 };
            u32 orow[8][2];
            u32x4 rr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) rr[k] = *(const u32x4 *)(rs + k * 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int bx = k & 1, byc = k >> 1;
                u32 p[4];
                mb_mode_pred(p_uv_mode, bx ? aC1 : aC0, byc ? lC1 : lC0, tlC, dcC, p);
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
                for (int jj = 0; jj < 4; jj++) orow[byc * 4 + jj][bx] = o[jj];
                if (byc) bot[bx] = o[3];
                if (bx) rc[byc] = right_column(o);
            }
#pragma unroll
            for (int jj = 0; jj < 8; jj += 2)
                *(g_u32x4p)(dC + jj * 8) = (u32x4){ orow[jj][0], orow[jj][1], orow[jj + 1][0], orow[jj + 1][1] };
            if (pl) { bV[0] = bot[0]; bV[1] = bot[1]; lV[0] = rc[0]; lV[1] = rc[1]; }
            else { bU[0] = bot[0]; bU[1] = bot[1]; lU[0] = rc[0]; lU[1] = rc[1]; }
        }
        prevLastU = p_lastU; prevLastV = p_lastV;
        if (p_act) { mbp += 16; cfp += VP8IR_COEF_PER_MB; tp += VP8_TILE_BYTES; abp += VP8_TILE_BYTES; }
        // ---- history: what the lane below will ask for in one and in two steps (an idle lane repeats its last line)
#pragma unroll
        for (int i = 0; i < 4; i++) { h2Y[i] = h1Y[i]; h1Y[i] = p_act ? p_bY[i] : h1Y[i]; }
#pragma unroll
        for (int i = 0; i < 2; i++) { h2U[i] = h1U[i]; h1U[i] = p_act ? bU[i] : h1U[i]; h2V[i] = h1V[i]; h1V[i] = p_act ? bV[i] : h1V[i]; }
        STAMP(2)

        // ======================= this step =======================
        if (++c == P) { c = 0; V += G; }
        // what the lane above finished: two steps ago (straight above) and last step (above-right)
        u32 nY[4], nU[2], nV[2];
#pragma unroll
        for (int i = 0; i < 4; i++) nY[i] = from_lane_above(h2Y[i]);
        const u32 nAR = from_lane_above(h1Y[0]);
#pragma unroll
        for (int i = 0; i < 2; i++) { nU[i] = from_lane_above(h2U[i]); nV[i] = from_lane_above(h2V[i]); }

        const bool act = t < T && c >= 0 && c < cols && V < Vmax;
        const bool late = act && !p_more;        // first macroblock of a row: nothing was prepared (or prefetched) a step ahead
        u32 jm = n_jm, dc_given = n_dcg;         // blocks 0..23 that have a residual; 1: the luma DCs come out of the Y2 block
        u32 cur_w0 = nx_m0.x;
        u32x4 bm = nx_bm;
        u32 rbY[4] = { nx_aY[0], nx_aY[1], nx_aY[2], nx_aY[3] }, rbAR = nx_ar, rbU[2] = { nx_aU[0], nx_aU[1] }, rbV[2] = { nx_aV[0], nx_aV[1] };
        if (late) {
            // ---- new macroblock row (c == 0): which frame, which row; pointers and quantisers; its first macroblock
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
            mbp = (g_cu32p)(job->mbs + (long)r * cols);
            cfp = (g_cs16p)(job->coef + (long)r * cols * VP8IR_COEF_PER_MB);
            tp = (g_u8p)(job->tile + (long)r * rowbytes);
            abp = r == 0 ? (g_cu8p)tp : (g_cu8p)tp - rowbytes;     // (no row above the first: any valid address)
            lY[0] = lY[1] = lY[2] = lY[3] = 0x81818181u;    // left border 129 (setupintrarecon.c:15-32)
            lU[0] = lU[1] = lV[0] = lV[1] = 0x81818181u;
            const u32x4 m0 = *(g_cu32x4p)mbp, m1 = *(g_cu32x4p)(mbp + 4), b0 = *(g_cu32x4p)(mbp + 10);
            const u32x4 y2a = *(g_cu32x4p)(cfp + 384), y2b = *(g_cu32x4p)(cfp + 392);
            if (pos == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) rbY[i] = load_l2((const unsigned char *)abp + 15 * 16 + 4 * i);
                rbAR = load_l2((const unsigned char *)abp + VP8_TILE_BYTES + 15 * 16);
                rbU[0] = load_l2((const unsigned char *)abp + 256 + 56); rbU[1] = load_l2((const unsigned char *)abp + 256 + 60);
                rbV[0] = load_l2((const unsigned char *)abp + 320 + 56); rbV[1] = load_l2((const unsigned char *)abp + 320 + 60);
            }
            // ... and its second one, which the prefetches at the top of the iteration (old pointers) missed
            pf_m0 = *(g_cu32x4p)(mbp + 16); pf_m1 = *(g_cu32x4p)(mbp + 20); pf_bm = *(g_cu32x4p)(mbp + 26);
            pf_y2a = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + 384); pf_y2b = *(g_cu32x4p)(cfp + VP8IR_COEF_PER_MB + 392);
            if (pos == 0) {
                const unsigned char *pa = (const unsigned char *)abp + VP8_TILE_BYTES;
#pragma unroll
                for (int i = 0; i < 4; i++) pf_aY[i] = load_l2(pa + 15 * 16 + 4 * i);
                pf_ar = load_l2(pa + VP8_TILE_BYTES + 15 * 16);
                pf_aU[0] = load_l2(pa + 256 + 56); pf_aU[1] = load_l2(pa + 256 + 60);
                pf_aV[0] = load_l2(pa + 320 + 56); pf_aV[1] = load_l2(pa + 320 + 60);
            }
            // these loads are this branch's own affair: consumed here (hipcc waits in front of the asm that names them), so that
            // no pending load leaves the branch and degrades the waits of the common path
            u32x4 m0s = m0, m1s = m1, b0s = b0, y2as = y2a, y2bs = y2b;
            asm volatile("" : "+v"(m0s), "+v"(m1s), "+v"(b0s), "+v"(y2as), "+v"(y2bs));
            asm volatile("" : "+v"(rbY[0]), "+v"(rbY[1]), "+v"(rbY[2]), "+v"(rbY[3]), "+v"(rbAR), "+v"(rbU[0]), "+v"(rbU[1]), "+v"(rbV[0]), "+v"(rbV[1]));
            asm volatile("" : "+v"(pf_m0), "+v"(pf_m1), "+v"(pf_bm), "+v"(pf_y2a), "+v"(pf_y2b));
            asm volatile("" : "+v"(pf_aY[0]), "+v"(pf_aY[1]), "+v"(pf_aY[2]), "+v"(pf_aY[3]), "+v"(pf_ar), "+v"(pf_aU[0]), "+v"(pf_aU[1]), "+v"(pf_aV[0]), "+v"(pf_aV[1]));
            cur_w0 = m0s.x; bm = b0s;
            prepare_mb(m0s, m1s, y2as, y2bs, cfp, jm, dc_given);
        }
        if (!act) { jm = 0; dc_given = 0; }
        const bool top = r == 0;
        const bool more = act && c + 1 < cols;
        // ---- macroblock descriptor
        const int y_mode = cur_w0 & 0xff, uv_mode = (cur_w0 >> 8) & 0xff;
        const bool bpred = y_mode == VP8IR_B_PRED;
        // ---- unfiltered line above (127 above the frame; vp8_setup_intra_recon)
        u32 aY[4], arY, aU[2], aV[2];
        if (top) {
            aY[0] = aY[1] = aY[2] = aY[3] = arY = 0x7f7f7f7fu;
            aU[0] = aU[1] = aV[0] = aV[1] = 0x7f7f7f7fu;
        } else if (pos == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) aY[i] = rbY[i];
            arY = rbAR;
            aU[0] = rbU[0]; aU[1] = rbU[1]; aV[0] = rbV[0]; aV[1] = rbV[1];
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
        int dcY = 128;
        if (up | lf) {
            const int shift = 3 + up + lf;
            const int s = (up ? sad4(aY[0]) + sad4(aY[1]) + sad4(aY[2]) + sad4(aY[3]) : 0)
                        + (lf ? sad4(lY[0]) + sad4(lY[1]) + sad4(lY[2]) + sad4(lY[3]) : 0);
            dcY = (s + (1 << (shift - 1))) >> shift;
        }
        u32 abv[4] = { aY[0], aY[1], aY[2], aY[3] };      // line above the current block row (B_PRED chain)
        int tlrow = tlY;                                   // top-left of the block row's first block
        u32 nl[4] = { 0, 0, 0, 0 };                        // right column of this MB = left of the next
        const g_u8p tpe = act ? tp : (g_u8p)dummy;         // where this lane's pixels go
        STAMP(3)

        // ======================= luma: two phases of two block rows =======================
        // Blocks 0-7 of the lanes that stay in their row were queued at the top of the iteration (their coefficients are
        // here by now: the eight chroma row stores were issued since); a lane that starts a row queues them now.  Then the
        // transform of phase ph+1 is queued before the owners predict phase ph.
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
                u32 orow[4][4];                               // [row][block]: 16-byte rows for the write-out
                u32x4 rr[8];                                  // the four blocks' residual slots (whatever they hold)
#pragma unroll
                for (int k = 0; k < 8; k++) rr[k] = *(const u32x4 *)(rs + k * 4);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 p[4];
                    if (bpred) {
                        // decodframe.c:200-236; above-right of the right-hand block column is the MB's own
                        // above-right for every block row (reconintra4x4.c:305-317)
                        bpred4x4((bmw >> (8 * k)) & 0xff, abv[k], k < 3 ? abv[k + 1] : arY, left, tl, p);
                    } else {
                        mb_mode_pred(y_mode, aY[k], lcur, tlY, dcY, p);
                    }
                    u32 o[4] = { p[0], p[1], p[2], p[3] };
                    // the residual, where the block has one (the branch is taken per wave)
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
                g_u8p prow = tpe + by * 64;
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    *(g_u32x4p)(prow + jj * 16) = (u32x4){ orow[jj][0], orow[jj][1], orow[jj][2], orow[jj][3] };
                // rotate the per-row shift registers
                tlrow = lcur >> 24;
                lY[0] = lY[1]; lY[1] = lY[2]; lY[2] = lY[3];
                nl[0] = nl[1]; nl[1] = nl[2]; nl[2] = nl[3]; nl[3] = left;
            }
            STAMP(5)
            drain_phase(ph + 1, 8);                           // the eight row stores of the two block rows just written
            STAMP(6)
        }
        // ---- hand the chroma half over to the next iteration; the prefetches become plain registers
#pragma unroll
        for (int i = 0; i < 4; i++) { lY[i] = nl[i]; p_bY[i] = abv[i]; }
        prevLastY = aY[3] >> 24; p_lastU = aU[1] >> 24; p_lastV = aV[1] >> 24;
        p_act = act; p_more = more; p_tpe = tpe; p_uv_mode = uv_mode; p_tlU = tlU; p_tlV = tlV; p_up = up; p_lf = lf;
        p_aU[0] = aU[0]; p_aU[1] = aU[1]; p_aV[0] = aV[0]; p_aV[1] = aV[1]; p_jmc = jm >> 16;
        // (through volatile asm: the copies stay HERE -- sunk into the loops above they would count as uses of registers with a
        // load pending and make hipcc drain the memory queue in front of those loops)
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
