// VP8 frames in ONE pass: reconstruction and in-loop deblocking fused, "one macroblock row per LANE", for gfx950 -- key frames
// (vp8_keyframe_kernel) and, with the inter macroblocks' prediction taken from where vp8_inter_pred_kernel left it, launches with
// inter frames (vp8_interframe_kernel).
//
// What it replaces (all paths relative to the reference tree): the macroblock loop of vp8_decode_frame
// (vp8/decoder/decodframe.c:1116-1129 -> decode_mb_row :334 -> decode_macroblock :112, with reconintra.c, reconintra4x4.c,
// dequantize.c, idctllm.c, idct_blk.c behind it) AND vp8_loop_filter_frame (vp8/common/loopfilter.c:203-316, filters of
// loopfilter_filters.c) for frames whose macroblocks are all intra.  vp8_recon_simt.hip and vp8_loopfilter_simt.hip do the
// same work as two kernels with a macroblock-tiled scratch frame between them; here a lane reconstructs a macroblock and
// filters it in the same step, the pixels never leave its registers in between, and the finished rows go out once, into
// macroblock-window tiles (below) that a small pass turns into the raster frame buffer:
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

// The output: one 384-byte tile per macroblock in the job's scratch (DevJob::tile), rows x (cols + 1) of them, laid out by who
// knows which pixels when, so that every 64-byte half of a tile's three 128-byte lines is written whole, by one lane, within
// one step (the L2 merges the 16- and 8-byte stores of a step; what leaves it are full sectors -- 16-byte pieces of raster
// rows written a step apart cost this kernel a third of its time and made it 46 or 60 ms depending on where the frame
// buffers happened to lie):
//   the WINDOW of macroblock (r, c) is the pixel columns 16c-4 .. 16c+11 (chroma 8c-4 .. 8c+3): the last four pixels of a row
//   of the macroblock to the left are only final once this macroblock's left edge has been filtered;
//   the bottom four rows (chroma: rows 4..7) of a macroblock are finished by the lane below, two steps later, and are kept
//   macroblock-aligned.  vp8_detile_kf_kernel turns the tiles into the raster frame buffer.
enum {
    KT_Y_WIN = 0,         // luma rows 0..11 of the window, 16 B each: written by the macroblock's own lane
    KT_Y_BOT = 192,       // luma rows 12..15 of the macroblock, 16 B each: by the lane below (or its own at the frame's bottom)
    KT_U_WIN = 256,       // U rows 0..3 of the window, 8 B each; V at + 32
    KT_U_BOT = 320,       // U rows 4..7 of the macroblock; V at + 32
    // behind the tiles, 32 bytes per macroblock: the UNFILTERED bottom pixel line (prediction context), which only the last lane
    // of a strand leaves for the first (the filtered bottom rows it reads back from KT_Y_BOT / KT_U_BOT, where the last lane
    // left them unfinished)
    KH_Y = 0, KH_U = 16, KH_V = 24, KH_BYTES = 32
};

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

// ... for launches with inter frames: the segments' base levels, and the deltas a macroblock's reference frame and mode add
// (mb_level, vp8_simt_prims.hip.h, is the same arithmetic per macroblock)
__device__ __forceinline__ void frame_levels_inter(const vp8ir_frame_hdr &h, u32 &base, u32 &refd, u32 &moded)
{
    base = refd = moded = 0;
    if (!h.filter_level) return;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        int b = h.filter_level;
        if (h.segmentation_enabled) {
            if (h.mb_segment_abs_delta) b = h.segment_lf[s];
            else { b += h.segment_lf[s]; b = b < 0 ? 0 : (b > 63 ? 63 : b); }
        }
        base |= (u32)(b & 0xff) << (8 * s);
    }
    if (!h.mode_ref_lf_delta_enabled) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        refd |= (u32)(unsigned char)h.ref_lf_deltas[k] << (8 * k);
        moded |= (u32)(unsigned char)h.mode_lf_deltas[k] << (8 * k);
    }
}

#ifdef KF_NOSTORE          // (timing experiments only: the row stores of active lanes go to the dummy scratch too)
#define KF_ACT(a) false
#else
#define KF_ACT(a) (a)
#endif
#define SWAP_U32(a, b) { const u32 t_ = (a); (a) = (b); (b) = t_; }

} // namespace


// ---------------------------------------------------------------------------------------------------------------------
// The two ROLES of a wave of vp8_keyframe_kernel / vp8_interframe_kernel (kf_body<LUMA, INTER>; the kernels at the end of the file
// deal them out, a luma and a chroma wave to every SIMD).  Luma and chroma of a frame share nothing but the macroblock descriptors
// (separate rows of the macroblock's tile, separate lines of the hand-over tile), and a luma wave and a chroma wave together fit one
// SIMD -- registers (256 each) and LDS (19 KB each) --, so every SIMD has two instruction streams to issue from: what one stalls
// on (LDS and memory round trips, scalar / branch bubbles; a lone wave of this kind of code keeps the vector ALU busy less than
// half the time) the other fills.
//
// lgG, P, nstrands as in vp8_recon_simt_kernel.  The frames go to the jobs' macroblock-window tiles (DevJob::tile, layout above;
// vp8_detile_kf_kernel + vp8_extend_kernel make the raster frame buffers of them), which are the hand-over scratch of the strands'
// last lanes too.  `dummy`: scratch nobody reads (idle lanes store there: the stores of the step loop are unconditional, see
// vp8_recon_simt.hip on s_waitcnt).  INTER: launches with inter frames -- a macroblock with a reference frame takes its prediction
// out of its own tile, where vp8_inter_pred_kernel left it (see vp8_interframe_kernel below).
//
// The residual transform is the cooperative one of vp8_recon_simt.hip on a diet, so that two waves' worth fits: a PHASE is
// the four blocks the owner consumes next (luma: a block row; chroma: a plane); the lanes queue their blocks with
// coefficients (at most 4 x 64 = 256: four rounds), the coefficients come by LDS-DMA into ONE 8 KB staging buffer, all lanes
// transform one block each per round IN PLACE, and the owners fetch their residuals into registers before the next phase's
// coefficients are requested into the same buffer (they are then on their way during the owner's prediction + filtering of
// the current phase).  The macroblock descriptors of the next step come by LDS-DMA too (a 16-byte slot per lane and piece),
// not through registers held for a step.
// LDS of a wave, whichever role it plays (19.3 KB: eight waves per CU, two per SIMD):
//   s_stage  [block of the phase][half][lane] 16 B: the owner's coefficients in, residuals out     8192 B
//   s_queue  owner lane | block in phase << 6 | DC given << 8                 512 B
//   s_tab    per owner: quantiser (dc | ac << 16)                             256 B
//   s_y2dc   per owner: the sixteen luma DCs out of the Y2 block, or -- no Y2 block -- the luma blocks' lone first coefficients
//            (vp8ir_mbx::y2); chroma: the eight chroma blocks' lone first coefficients (vp8ir_mbx::cdc)                  2048 B
//   s_desc   [piece][lane] 16 B: the next macroblock's record (vp8ir_mbx: luma 5 pieces, chroma 3)   5120 B
//   s_sf     [row][lane]: the last four pixels (filtered, biased) of pixel rows of the macroblock to the left -- luma rows 0..11,
//            chroma U rows 0..3, V rows 0..3 (the bottom four rows' are in registers: they double as the lane below's context).
//            Per-lane state read and written once per step and indexed by the block row: in LDS it costs no registers    3072 B
//   s_psel   (round 5) the v_perm_b32 selectors of the branch-free 4x4 predictor, 48 bytes per mode (pred4x4_net,
//            vp8_simt_prims.hip.h; the table is k_pred_sel, copied in at kernel entry)                                        528 B
template <bool LUMA, bool INTER>
__device__ __forceinline__ void kf_body(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy,
                                        const int wave, u32 *s_stage, unsigned short *s_queue, u32 *s_tab,
                                        u32 *s_y2dc, u32 *s_desc, u32 *s_sf, const u32 *s_psel)
{
    const int lane = threadIdx.x;
    const int G = 1 << lgG;
    const int pos = lane & (G - 1);
    // (a strand's first lane reads its context back from the hand-over tile: issued at the top of the step where the registers are
    // there for it -- key-frame launches: -1.4 % --, right in front of the wait in launches with inter frames, whose kernel spills with it)
#ifdef KF_READBACK_LATE
    constexpr bool RB_EARLY = false;
#else
    constexpr bool RB_EARLY = !INTER;
#endif
    const int spw = 64 >> lgG;
    const int strand = wave * spw + (lane >> lgG);
    const int cols = g.mb_cols, rows = g.mb_rows;
    const long rowbytes = (long)(cols + 1) * VP8_TILE_BYTES, hrow = (long)(cols + 1) * KH_BYTES;
    const int myjobs = strand < njobs ? (njobs - strand + nstrands - 1) / nstrands : 0;
    const int Vmax = myjobs * rows;
    const int wavejobs = (njobs - wave * spw + nstrands - 1) / nstrands;
    const int T = ((wavejobs * rows + G - 1) >> lgG) * P + 2 * (G - 1);
    const u32 stage_lane = (u32)(unsigned long)(lds_vp)s_stage + lane * 16;     // LDS byte address of this lane's slot of round 0, half 0
    const u32 stage_base = (u32)(unsigned long)(lds_vp)s_stage;
    const u32 desc_lane = (u32)(unsigned long)(lds_vp)s_desc + lane * 16;
    v2u one = mku(1);
    asm volatile("" : "+v"(one));            // see nz_clear

    // ---- per-lane row state; the pointers are valid addresses at all times
    g_cu32p mbp = (g_cu32p)jobs[0].mbx;         // record of the current macroblock (vp8ir_mbx: VP8IR_MBX_WORDS dwords)
    g_cs16p bp = (g_cs16p)jobs[0].blocks;                 // its first block in the slot's block stream (the blocks of a row follow each other)
    g_u8p tp = (g_u8p)dummy, hp = (g_u8p)dummy; // first tile / first unfiltered line of its macroblock row
    int r = 0;
    u32 dqs[4][2];                              // luma: y1, y2 quantisers per segment (dc | ac << 16); chroma: uv in [s][0]
#pragma unroll
    for (int s = 0; s < 4; s++) dqs[s][0] = dqs[s][1] = 0;
    // loop-filter levels.  Key-frame launches: lv_plain / lv_bpred = the level per segment, a byte each, for 16x16 modes / B_PRED.
    // Launches with inter frames: lv_plain = the segments' base levels, lv_bpred = ref_lf_deltas[0..3], lv_mode = mode_lf_deltas[0..3]
    // (signed bytes; both zero without mode_ref_lf_delta_enabled), ftype = the frame's type (the hev threshold depends on it)
    u32 lv_plain = 0, lv_bpred = 0, lv_mode = 0;
    int sharp = 0, ftype = 0; bool simple = false, lv_delta = false;
    s_tab[lane] = 0;
    // ---- prediction context (unfiltered).  Luma: l0[0..3] left column, h1/h2[0..3] bottom lines of the macroblocks finished one
    // and two steps ago.  Chroma: U in [0..1], V in [2..3].
    u32 l0[4] = { 0, 0, 0, 0 }, h1[4] = { 0, 0, 0, 0 }, h2[4] = { 0, 0, 0, 0 };
    int prevLast = 0, prevLast2 = 0;            // last pixel of the previous step's line above: Y, or U and V
    // ---- loop-filter context (filtered, biased), besides s_sf.  Luma: pb[j][0..2] the first twelve pixels of the bottom four rows of
    // the macroblock to the left, hF[j][0..3] the bottom four rows of the macroblock finished two steps ago (what the lane below
    // asks for).  Chroma: U in pb[j][0], hF[j][0..1]; V in pb[j][1], hF[j][2..3].
    u32 pb[4][3], hF[4][4];
    u32 *const sF = s_sf + lane;
    u32 sB[4] = { 0, 0, 0, 0 }, sB2[4] = { 0, 0, 0, 0 };      // the left neighbour's last dword in the bottom four rows (chroma: U, V)
#pragma unroll
    for (int i = 0; i < 12; i++) sF[i * 64] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { pb[j][0] = pb[j][1] = pb[j][2] = 0; hF[j][0] = hF[j][1] = hF[j][2] = hF[j][3] = 0; }
    // ---- the macroblock after the current one, prepared at the end of the step before: descriptor words 0, 1, sub-block modes;
    // which of its blocks have a residual, whether the luma DCs come out of the Y2 block
    u32 nx_w0 = 0, nx_w1 = 0, nx_jm = 0, nx_dcg = 0, nx_dq = 0;
    u32x4 nx_bm = { 0, 0, 0, 0 };
    bool p_more = false;

    // inter macroblocks: the part of the prediction (vp8_inter_pred_kernel left it in the macroblock's tile) the lane consumes next
    // -- luma: a block row, rows at 16 y; chroma: a plane, rows (0,1) (2,3) (4,5) (6,7) --, requested behind the row stores of the
    // part before it (for a macroblock's first part: of the macroblock before it), so that it arrives during the transform
    u32x4 pr[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    STAMP_DECL
    int q_n = 0;                                                     // blocks queued (wave-uniform)
    // The owner requests the coefficients of the blocks cfb[0..3] of its macroblock that have any (`m4`) into ITS OWN four staging
    // slots (s_stage[block][half][lane]: no look-up stands between knowing the macroblock and the request), and the lanes
    // queue these blocks, behind the n0 already queued, for the transform.
    // (The slot's block stream -- include/vp8_ir.h, the device form -- holds the blocks with eob > 1 and no others, a macroblock's
    // in block order at cf_mb: what is fetched is what is used.  rank0: such blocks of the macroblock in front of this phase's.)
    // f4: the phase's blocks with more than a DC: fetched, and queued for the transform.  The owner adds a lone DC itself
    // (vp8_dc_only_idct_add_c, idctllm.c:112-137) from the first coefficient that came with the macroblock's record, or, with a
    // Y2 block, from the Walsh transform's output.
    auto queue = [&](g_cs16p cf_mb, const int rank0, const u32 f4, const u32 dcg, const int n0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if ((f4 >> i) & 1) {
                g_cs16p cfb = cf_mb + (rank0 + __builtin_popcount(f4 & ((1u << i) - 1))) * 16;
                __builtin_amdgcn_global_load_lds((g_cvp)cfb, (lds_vp)(s_stage + i * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((g_cvp)(cfb + 8), (lds_vp)(s_stage + i * 512 + 256), 16, 0, 0);
            }
        }
        wave_lds_sync();                                             // the queue's last reader is done
        int n = n0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool b = (f4 >> i) & 1;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(b);
            const u32 at = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, (u32)n));
            if (b) s_queue[at] = (unsigned short)((u32)lane | ((u32)i << 6) | ((dcg & 1u) << 8));
            n += __builtin_popcountll(bal);
        }
        q_n = n;
    };
    // all lanes transform the queued blocks n0 .. q_n-1, one per lane and round, whoever they belong to, in place (in the owner's
    // slot).  `younger`: a LOWER bound of the memory instructions issued since `queue` (they may stay in flight)
    auto drain = [&](const int blk0, const int n0, const int younger) {
        if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(14)
        wave_lds_sync();
        const int R = (q_n + 63) >> 6;
#pragma unroll 1
        for (int rr = n0 >> 6; rr < R; rr++) {
            const int idx = rr * 64 + lane;
            if (idx >= n0 && idx < q_n) {
                const u32 ent = s_queue[idx];
                const int owner = ent & 63, i = (ent >> 6) & 3;
                const bool given = (ent >> 8) & 1;
                // (the quantiser and the Y2 DC are requested before the coefficients: one LDS round trip for all of them)
                const u32 dq = s_tab[owner];
                const int blk = blk0 + i;
                const u32 y2w = LUMA ? s_y2dc[owner * 8 + (blk >> 1)] : 0u;
                const u32 slot = stage_base + i * 2048 + owner * 16;
                u32x4 ca, cb;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ca), "=&v"(cb) : "v"(slot) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int dc_in = (short)(y2w >> (16 * (blk & 1)));
                int res[16];
                dequant_idct(ca, cb, dq & 0xffff, dq >> 16, given, dc_in, res);
                u32x4 oa, ob;
                oa.x = ((u32)res[0] & 0xffff) | ((u32)res[1] << 16); oa.y = ((u32)res[2] & 0xffff) | ((u32)res[3] << 16);
                oa.z = ((u32)res[4] & 0xffff) | ((u32)res[5] << 16); oa.w = ((u32)res[6] & 0xffff) | ((u32)res[7] << 16);
                ob.x = ((u32)res[8] & 0xffff) | ((u32)res[9] << 16); ob.y = ((u32)res[10] & 0xffff) | ((u32)res[11] << 16);
                ob.z = ((u32)res[12] & 0xffff) | ((u32)res[13] << 16); ob.w = ((u32)res[14] & 0xffff) | ((u32)res[15] << 16);
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"(slot), "v"(oa), "v"(ob) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_lds_sync();
    };
    // the owner's residuals of the phase just drained, out of its four slots: block k's two halves in rr[2k], rr[2k+1] (garbage
    // where it has none)
    auto fetch = [&](u32x4 (&rr)[8]) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(rr[2 * k]), "=&v"(rr[2 * k + 1]) : "v"(stage_lane + k * 2048) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]), "+v"(rr[4]), "+v"(rr[5]), "+v"(rr[6]), "+v"(rr[7]) :: "memory");
    };

    // ---- what the transform needs to know about a macroblock, from its record (m0: words 0-3, m1: words 4-7; luma: y2a, y2b =
    // vp8ir_mbx::y2; chroma: y2a = vp8ir_mbx::cdc): which of its blocks (luma: 16 bits, chroma: 8) have a residual (`jm`), whether
    // the luma DCs come out of the Y2 block (bit 0 of `dcg`); how many blocks it has in the stream (luma: its own are jm's bits
    // 16..31, the chroma ones' number in dcg's bits 8..12; chroma: the luma ones' number in jm's bits 8..12, its own are bits
    // 16..23); the lane's entry of s_tab (quantiser) and of s_y2dc
    auto prepare_mb = [&](const u32x4 m0, const u32x4 m1, const u32x4 y2a, const u32x4 y2b, u32 &jm, u32 &dcg, u32 &dq_out) {
        const u32 w0 = m0.x, w1 = m0.y;
        const int y_mode = w0 & 0xff;
        const bool skip = (w0 >> 24) & VP8IR_MB_SKIP;
        const bool has_y2 = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV;
        const int seg = w1 & 3;
        // (copies first: a select between by-reference captures would become a dynamic index into the closure, in scratch)
        const u32 s00 = dqs[0][0], s01 = dqs[0][1], s10 = dqs[1][0], s11 = dqs[1][1];
        const u32 s20 = dqs[2][0], s21 = dqs[2][1], s30 = dqs[3][0], s31 = dqs[3][1];
        const u32 dq0 = seg == 0 ? s00 : seg == 1 ? s10 : seg == 2 ? s20 : s30;
        const u32 dq1 = seg == 0 ? s01 : seg == 1 ? s11 : seg == 2 ? s21 : s31;
        // eobs (detokenize.c:363), a byte per block, 0..16: which blocks have a token at all.  A luma block of a macroblock with
        // Y2 always has its DC (idct_blk.c:20-44, decodframe.c:262-296).
        // Bits 16.. of the result: the blocks with more than a DC (eob >= 2), which alone go through the transform.
        u32 m = 0, nchroma = 0;
        if constexpr (LUMA) {
            const u32 e[4] = { m0.z, m0.w, m1.x, m1.y };
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 ge1 = ((e[q] + 0x7f7f7f7fu) & 0x80808080u) >> 7;        // bit 0 of each byte: eob >= 1
                const u32 ge2 = ((e[q] + 0x7e7e7e7eu) & 0x80808080u) >> 7;        // eob >= 2
                m |= (((ge1 * 0x00204081u) >> 21) & 0xfu) << (4 * q);
                m |= (((ge2 * 0x00204081u) >> 21) & 0xfu) << (16 + 4 * q);
            }
            if (has_y2) m |= 0xffffu;
            // the chroma blocks behind this macroblock's luma blocks in the stream
            nchroma = __builtin_popcount((m1.z + 0x7e7e7e7eu) & 0x80808080u) + __builtin_popcount((m1.w + 0x7e7e7e7eu) & 0x80808080u);
        } else {
            const u32 e[2] = { m1.z, m1.w };
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32 ge1 = ((e[q] + 0x7f7f7f7fu) & 0x80808080u) >> 7;
                const u32 ge2 = ((e[q] + 0x7e7e7e7eu) & 0x80808080u) >> 7;
                m |= (((ge1 * 0x00204081u) >> 21) & 0xfu) << (4 * q);
                m |= (((ge2 * 0x00204081u) >> 21) & 0xfu) << (16 + 4 * q);
            }
            // the luma blocks stored in front of the chroma ones, counted into bits 8..12
            const u32 el[4] = { m0.z, m0.w, m1.x, m1.y };
            u32 nl = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) nl += __builtin_popcount((el[q] + 0x7e7e7e7eu) & 0x80808080u);       // (eob >= 2: the blocks in the stream)
            m |= nl << 8;
        }
        if (skip) m = 0;
        jm = m;
        dcg = (LUMA && has_y2 && !skip ? 1u : 0u) | (skip ? 0u : nchroma << 8);
        s_tab[lane] = dq0;
        dq_out = dq0;
        if constexpr (LUMA) {
            // Y2: vp8_dequantize_b + vp8_short_inv_walsh4x4_c (idctllm.c:140-192) -> the 16 luma DCs.  With nothing but a DC
            // coefficient the full transform gives what vp8_short_inv_walsh4x4_1_c gives (decodframe.c:282-285).
            if (__builtin_amdgcn_ballot_w64((dcg & 1) != 0) != 0) {
                if (dcg & 1) {
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
            // without a Y2 block: vp8ir_mbx::y2 holds the sixteen luma blocks' lone first coefficients
            if (!has_y2 && !skip) {
                u32x4 *dst = (u32x4 *)(s_y2dc + lane * 8);
                dst[0] = y2a;
                dst[1] = y2b;
            }
        } else {
            // the eight chroma blocks' lone first coefficients (vp8ir_mbx::cdc)
            if (!skip) *(u32x4 *)(s_y2dc + lane * 8) = y2a;
        }
    };
    // the next macroblock's record (vp8ir_mbx), the pieces this role reads, requested into the lane's slots of s_desc: descriptor
    // words 0-3, 4-7; luma: the sub-block modes (words 10-13) and y2 (16-23); chroma: cdc (24-27)
    auto request_desc = [&](g_cu32p mb) {
        __builtin_amdgcn_global_load_lds((g_cvp)mb, (lds_vp)(s_desc), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_cvp)(mb + 4), (lds_vp)(s_desc + 256), 16, 0, 0);
        if constexpr (LUMA) {
            __builtin_amdgcn_global_load_lds((g_cvp)(mb + 10), (lds_vp)(s_desc + 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((g_cvp)(mb + 16), (lds_vp)(s_desc + 768), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((g_cvp)(mb + 20), (lds_vp)(s_desc + 1024), 16, 0, 0);
        } else
            __builtin_amdgcn_global_load_lds((g_cvp)(mb + 24), (lds_vp)(s_desc + 512), 16, 0, 0);
    };

    int c = -2 * pos - 1, V = pos;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        STAMP(0)
        if (++c == P) { c = 0; V += G; }
        // ---- what the lane above finished: prediction context (two steps ago: straight above; last step: above-right) and the
        // filter's context rows (two steps ago); then this lane's own offer: the macroblock held from the step before
        u32 nA[4], nAR = 0, tF[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) nA[i] = from_lane_above(h2[i]);
        if constexpr (LUMA) nAR = from_lane_above(h1[0]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < 4; i++) tF[j][i] = from_lane_above(hF[j][i]);
            if constexpr (LUMA) { hF[j][0] = pb[j][0]; hF[j][1] = pb[j][1]; hF[j][2] = pb[j][2]; hF[j][3] = sB[j]; }
            else { hF[j][0] = pb[j][0]; hF[j][1] = sB[j]; hF[j][2] = pb[j][1]; hF[j][3] = sB2[j]; }
        }
        const bool act = c >= 0 && c < cols && V < Vmax;
        const bool late = act && !p_more;        // first macroblock of a row: nothing was prepared a step ahead
        // A strand's first lane takes its context from the hand-over tile the strand's last lane wrote a round ago: the loads are issued
        // HERE, at the top of the step (a row's first macroblock: below, once the row's pointers are known) -- agent-scope: served by the L2,
        // see load_l2; six wide ones instead of 21 dwords -- and waited for where the context is needed
        u32x4 rb_a = { 0, 0, 0, 0 }, rb_0 = { 0, 0, 0, 0 }, rb_1 = { 0, 0, 0, 0 }, rb_2 = { 0, 0, 0, 0 }, rb_3 = { 0, 0, 0, 0 };
        u32 rb_r = 0;
        auto readback_issue = [&](const unsigned char *ha, const unsigned char *pa) {
            if constexpr (LUMA)
                asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                             "global_load_dword %5, %6, off offset:32 sc1\n\t"
                             "global_load_dwordx4 %1, %7, off offset:192 sc1\n\t"
                             "global_load_dwordx4 %2, %7, off offset:208 sc1\n\t"
                             "global_load_dwordx4 %3, %7, off offset:224 sc1\n\t"
                             "global_load_dwordx4 %4, %7, off offset:240 sc1"
                             : "=&v"(rb_a), "=&v"(rb_0), "=&v"(rb_1), "=&v"(rb_2), "=&v"(rb_3), "=&v"(rb_r) : "v"(ha), "v"(pa) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %5, off offset:16 sc1\n\t"
                             "global_load_dwordx4 %1, %6, off offset:320 sc1\n\t"
                             "global_load_dwordx4 %2, %6, off offset:336 sc1\n\t"
                             "global_load_dwordx4 %3, %6, off offset:352 sc1\n\t"
                             "global_load_dwordx4 %4, %6, off offset:368 sc1"
                             : "=&v"(rb_a), "=&v"(rb_0), "=&v"(rb_1), "=&v"(rb_2), "=&v"(rb_3) : "v"(ha), "v"(pa) : "memory");
        };
        if constexpr (RB_EARLY)
            if (act && !late && pos == 0 && r != 0)
                readback_issue((const unsigned char *)hp + (long)c * KH_BYTES - hrow, (const unsigned char *)tp + (long)c * VP8_TILE_BYTES - rowbytes);
        u32 jm = nx_jm, dc_given = nx_dcg, cur_dq = nx_dq;
        u32 cur_w0 = nx_w0, cur_w1 = nx_w1;
        u32x4 bm = nx_bm;
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
                else { d[0] = dqs[0][0]; d[1] = dqs[0][1]; d[2] = dqs[0][0]; }
                dqs[s][0] = LUMA ? d[0] : d[2]; dqs[s][1] = d[1];
            }
            if constexpr (INTER) {
                frame_levels_inter(h, lv_plain, lv_bpred, lv_mode);
                lv_delta = h.filter_level && h.mode_ref_lf_delta_enabled;
                ftype = vp8ir_lf_frame_type(&h);
            } else
                frame_levels(h, lv_plain, lv_bpred);
            sharp = h.sharpness_level; simple = h.filter_type == 1;
            mbp = (g_cu32p)(job->mbx + (long)r * cols);
            tp = (g_u8p)(job->tile + (long)r * rowbytes);
            hp = (g_u8p)(job->tile + (long)rows * rowbytes + (long)r * hrow);
            l0[0] = l0[1] = l0[2] = l0[3] = 0x81818181u;    // left border 129 (setupintrarecon.c:15-32)
            u32x4 m0 = *(g_cu32x4p)mbp, m1 = *(g_cu32x4p)(mbp + 4), b0 = { 0, 0, 0, 0 }, y2a = { 0, 0, 0, 0 }, y2b = { 0, 0, 0, 0 };
            u32 first = mbp[14];                // vp8ir_mb::sparse_first: where the row's blocks begin
            if constexpr (LUMA) { b0 = *(g_cu32x4p)(mbp + 10); y2a = *(g_cu32x4p)(mbp + 16); y2b = *(g_cu32x4p)(mbp + 20); }
            else y2a = *(g_cu32x4p)(mbp + 24);
            // consumed here, so that no pending load leaves the branch
            asm volatile("" : "+v"(m0), "+v"(m1), "+v"(b0), "+v"(y2a), "+v"(y2b), "+v"(first));
            bp = (g_cs16p)(job->blocks + (long)first * 16);
            cur_w0 = m0.x; cur_w1 = m0.y; bm = b0;
            prepare_mb(m0, m1, y2a, y2b, jm, dc_given, cur_dq);
        }
        if (!act) { jm = 0; dc_given = 0; }
        // the row starters' first phase joins the queue behind the blocks the others had transformed at the end of the last step
        if (__builtin_amdgcn_ballot_w64(late) != 0) {
            const int n0 = q_n;
            queue(bp, LUMA ? 0 : (int)((jm >> 8) & 0x1f), late ? (jm >> 16) & 0xf : 0, dc_given, n0);
            drain(LUMA ? 0 : 16, n0, 0);
        }
        // the descriptor of the macroblock after this one: on its way from here (a lane at the end of its row, or idle, fetches
        // whatever follows: never used)
        request_desc(mbp + VP8IR_MBX_WORDS);
        STAMP(1)
        const bool top = r == 0;
        const bool more = act && c + 1 < cols;
#ifdef KF_NO_READBACK      // (timing experiments only: what the L2 round trip below costs -- 4-5 % of the kernel; asking a step ahead
                           // costs 21 registers the luma wave does not have: +2 % key frames, +12 % inter frames with the spills)
        const bool readback = false;
#else
        const bool readback = act && pos == 0 && !top;          // first lane of a strand: its context comes from the hand-over tile
#endif
        const bool last_col = act && c == cols - 1, last_row = r == rows - 1;
        // bottom rows that no lane below takes over are written here: to the frame (last row), or to the hand-over tile
        const bool write_bottom = pos == G - 1 || last_row;
        const bool hand = act && pos == G - 1 && !last_row;
        const g_u8p tpc = tp + (long)(act ? c : 0) * VP8_TILE_BYTES;                 // this macroblock's tile
        const g_u8p hpc = hp + (long)(act ? c : 0) * KH_BYTES;                       // ... and unfiltered line
        if constexpr (RB_EARLY) { if (readback && late) readback_issue((const unsigned char *)hpc - hrow, (const unsigned char *)tpc - rowbytes); }
        // ---- macroblock descriptor; loop-filter parameters (vp8_loop_filter_frame, loopfilter.c:245-299)
        const int y_mode = cur_w0 & 0xff, uv_mode = (cur_w0 >> 8) & 0xff;
        const bool bpred = y_mode == VP8IR_B_PRED;
        const int ref_frame = INTER ? (int)((cur_w0 >> 16) & 3) : 0;
        const bool is_inter = INTER && act && ref_frame != VP8IR_INTRA_FRAME;
        // (launches with inter frames: most waves hold nothing but inter macroblocks, and skip the intra predictors)
        const bool any_intra = !INTER || __builtin_amdgcn_ballot_w64(act && !is_inter) != 0;
        int level;
        if constexpr (INTER) {
            // vp8_loop_filter_frame_init (loopfilter.c:117-201) for this macroblock's segment, reference frame and mode class
            level = (int)((lv_plain >> (8 * (cur_w1 & 3))) & 0xff);
            if (lv_delta) {
                const int m = ref_frame == VP8IR_INTRA_FRAME ? (bpred ? 0 : -1)
                                                             : (y_mode == VP8IR_ZEROMV ? 1 : (y_mode == VP8IR_SPLITMV ? 3 : 2));
                level += (int)(signed char)(lv_bpred >> (8 * ref_frame)) + (m >= 0 ? (int)(signed char)(lv_mode >> (8 * m)) : 0);
                level = level < 0 ? 0 : (level > 63 ? 63 : level);
            }
            if (!act) level = 0;
        } else
            level = act ? (int)(((bpred ? lv_bpred : lv_plain) >> (8 * (cur_w1 & 3))) & 0xff) : 0;
        const Lim L = mb_limits(sharp, level, INTER ? ftype : 0, one);
        const bool on = level != 0;
        const bool skip_lf = !bpred && y_mode != VP8IR_SPLITMV && ((cur_w0 >> 24) & VP8IR_MB_SKIP);
        const bool mbv = on && c > 0, inner = on && !skip_lf, mbh = on && !top;
        // the simple filter leaves chroma alone (loopfilter.c:283-299)
        const bool any_normal = __builtin_amdgcn_ballot_w64(on && !simple) != 0;
        const bool any_simple = LUMA && __builtin_amdgcn_ballot_w64(on && simple) != 0;
        auto gate = [](bool b) { return lf_gate(b); };
        const Gates gv = { gate(mbv && !simple), gate(inner && !simple), gate(LUMA && mbv && simple), gate(LUMA && inner && simple), any_normal, any_simple };
        const Gates gh = { gate(mbh && !simple), gate(inner && !simple), gate(LUMA && mbh && simple), gate(LUMA && inner && simple), any_normal, any_simple };
        // ---- unfiltered line above (127 above the frame; vp8_setup_intra_recon)
        u32 aA[4], arY = 0;
        if (top) {
            aA[0] = aA[1] = aA[2] = aA[3] = arY = 0x7f7f7f7fu;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) aA[i] = nA[i];
            arY = nAR;
        }
        if (readback) {
            if constexpr (!RB_EARLY) readback_issue((const unsigned char *)hpc - hrow, (const unsigned char *)tpc - rowbytes);
            static_assert(KH_Y == 0 && KH_BYTES == 32 && KT_Y_BOT == 192 && KH_U == 16 && KH_V == 24 && KT_U_BOT == 320, "offsets in the loads above");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb_a), "+v"(rb_0), "+v"(rb_1), "+v"(rb_2), "+v"(rb_3), "+v"(rb_r) :: "memory");
            if constexpr (LUMA) {
                aA[0] = rb_a.x; aA[1] = rb_a.y; aA[2] = rb_a.z; aA[3] = rb_a.w;
                arY = rb_r;
                const u32x4 vv[4] = { rb_0, rb_1, rb_2, rb_3 };
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    tF[j][0] = vv[j].x ^ VP8_LF_BIAS; tF[j][1] = vv[j].y ^ VP8_LF_BIAS;
                    tF[j][2] = vv[j].z ^ VP8_LF_BIAS; tF[j][3] = vv[j].w ^ VP8_LF_BIAS;
                }
            } else {
                const u32x4 va = rb_a, u0 = rb_0, u1 = rb_1, w0 = rb_2, w1 = rb_3;
                aA[0] = va.x; aA[1] = va.y; aA[2] = va.z; aA[3] = va.w;
                tF[0][0] = u0.x ^ VP8_LF_BIAS; tF[0][1] = u0.y ^ VP8_LF_BIAS; tF[1][0] = u0.z ^ VP8_LF_BIAS; tF[1][1] = u0.w ^ VP8_LF_BIAS;
                tF[2][0] = u1.x ^ VP8_LF_BIAS; tF[2][1] = u1.y ^ VP8_LF_BIAS; tF[3][0] = u1.z ^ VP8_LF_BIAS; tF[3][1] = u1.w ^ VP8_LF_BIAS;
                tF[0][2] = w0.x ^ VP8_LF_BIAS; tF[0][3] = w0.y ^ VP8_LF_BIAS; tF[1][2] = w0.z ^ VP8_LF_BIAS; tF[1][3] = w0.w ^ VP8_LF_BIAS;
                tF[2][2] = w1.x ^ VP8_LF_BIAS; tF[2][3] = w1.y ^ VP8_LF_BIAS; tF[3][2] = w1.z ^ VP8_LF_BIAS; tF[3][3] = w1.w ^ VP8_LF_BIAS;
            }
        }
        const int up = !top, lf = c > 0;
        STAMP(2)

        if constexpr (LUMA) {
            // ======================= luma: four block rows =======================
            // vp8_extend_mb_row (extend.c:160-185): right of the frame the line repeats its last pixel
            if (!top && c == cols - 1) arY = splat(aA[3] >> 24);
            const int tlY = top ? 127 : (c == 0 ? 129 : prevLast);
            int dcY = 128;
            if (up | lf) {
                const int shift = 3 + up + lf;
                const int s = (up ? sad4(aA[0]) + sad4(aA[1]) + sad4(aA[2]) + sad4(aA[3]) : 0)
                            + (lf ? sad4(l0[0]) + sad4(l0[1]) + sad4(l0[2]) + sad4(l0[3]) : 0);
                dcY = (s + (1 << (shift - 1))) >> shift;
            }
            // the branch-free predictor's per-macroblock inputs (pred4x4_net): which table entry a block takes -- a B_PRED macroblock's
            // sub-block modes; any other: 0 (the dword C below: DC_PRED's value or, V_PRED, the line above), PSEL_MB_H, or B_TM_PRED's for TM_PRED
            {
                const u32 emb = y_mode == VP8IR_H_PRED ? (u32)PSEL_MB_H * 0x01010101u : (y_mode == VP8IR_TM_PRED ? (u32)VP8IR_B_TM_PRED * 0x01010101u : 0u);
                if (!(bpred && act)) bm = (u32x4){ emb, emb, emb, emb };
            }
            const u32 dcs = perm((u32)dcY, (u32)dcY, 0u);
            const bool v_mb = y_mode == VP8IR_V_PRED;
            const u32 Cmb[4] = { v_mb ? aA[0] : dcs, v_mb ? aA[1] : dcs, v_mb ? aA[2] : dcs, v_mb ? aA[3] : dcs };
            u32 abv[4] = { aA[0], aA[1], aA[2], aA[3] };      // line above the current block row (B_PRED chain)
            int tlrow = tlY;                                   // top-left of the block row's first block
            u32 nl[4] = { 0, 0, 0, 0 };                        // right column of this macroblock = left of the next
            // the filter's rows above the first block row: rows 12..15 of the macroblock above (tF); from then on the block row before
            u32 sfix[4] = { 0, 0, 0, 0 };             // the left neighbour's last dword in the rows above the current block row, fixed up
            // where the rows above the current block row go: first the bottom rows of the macroblock above, then this one's window
            g_u8p prow = KF_ACT(act && !top) ? tpc - rowbytes + KT_Y_BOT : (g_u8p)dummy;
            int pstride = KF_ACT(act && !top) ? 16 : 0;
            if constexpr (INTER) {
                if (is_inter && late) {          // first macroblock of a row: nobody asked ahead
#pragma unroll
                    for (int j = 0; j < 4; j++) pr[j] = *(g_cu32x4p)(tpc + 16 * j);
                }
            }
#pragma unroll 1
            for (int by = 0; by < 4; by++) {
                // (landed before the next phase's coefficients are requested: nothing older than those may be waited for later)
                if constexpr (INTER) asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]));
                // ---- the block row's residuals; then the next phase's coefficients are requested (the next block row's, or the next
                // macroblock's first, whose descriptor has arrived by now: at least the 12 row stores below are younger)
                u32x4 rr[8];
                const u32 rmg = jm >> (by * 4), rmf = jm >> (16 + by * 4);
                // the block row's four DCs out of the Y2 block, for the blocks that have nothing else
                const u32 y2w0 = s_y2dc[lane * 8 + 2 * by], y2w1 = s_y2dc[lane * 8 + 2 * by + 1];
                if (by < 3) {
                    fetch(rr);
                    STAMP(8)
                    queue(bp, __builtin_popcount((jm >> 16) & 0xffffu & ((16u << (4 * by)) - 1)), (jm >> (4 * by + 20)) & 0xf, dc_given, 0);
                } else {
                    // (all of this macroblock's phases have been transformed: its entries of s_tab / s_y2dc are free)
                    u32 n_jm = 0, n_dcg = 0;
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    {
                        u32x4 m0, m1, y2a, y2b;
                        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                                     "ds_read_b128 %2, %4 offset:3072\n\tds_read_b128 %3, %4 offset:4096\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(m0), "=&v"(m1), "=&v"(y2a), "=&v"(y2b) : "v"(desc_lane) : "memory");
                        u32 n_dq = 0;
                        if (more) prepare_mb(m0, m1, y2a, y2b, n_jm, n_dcg, n_dq);
                        nx_w0 = m0.x; nx_w1 = m0.y; nx_jm = n_jm; nx_dcg = n_dcg; nx_dq = n_dq;
                    }
                    asm volatile("ds_read_b128 %0, %1 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=&v"(nx_bm) : "v"(desc_lane) : "memory");
                    STAMP(9)
                    fetch(rr);
                    STAMP(8)
                    // (this macroblock's blocks in the stream are behind us: the next one's follow them)
                    bp += (__builtin_popcount(jm >> 16) + (int)((dc_given >> 8) & 0x1f)) * 16;
                    queue(bp, 0, (n_jm >> 16) & 0xf, n_dcg, 0);
                }
                STAMP(3)
                const u32 lcur = l0[0];
                const u32 bmw = by == 0 ? bm.x : by == 1 ? bm.y : by == 2 ? bm.z : bm.w;
                u32 left = lcur;
                int tl = tlrow;
                u32 orow[4][4];                               // [row][block]
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 p[4] = { 0, 0, 0, 0 };
                    if (any_intra) {
                        // decodframe.c:200-236; above-right of the right-hand block column is the macroblock's own
                        // above-right for every block row (reconintra4x4.c:305-317)
                        const u32 em = (bmw >> (8 * k)) & 0xff;
                        const u32 a0 = abv[k], a1 = k < 3 ? abv[k + 1] : arY;
                        const u32 bdc = __builtin_amdgcn_sad_u8(left, 0u, __builtin_amdgcn_sad_u8(a0, 0u, 4u)) >> 3;      // B_DC_PRED
                        pred4x4_net((const u32x4 *)(s_psel + em * PSEL_WORDS), a0, a1, left, tl, em == 0, bpred ? perm(bdc, bdc, 0u) : Cmb[k],
                                    !bpred, lcur, p);
                        // TM_PRED of the macroblock (above = the macroblock's line above, left = its left column) and B_TM_PRED of a
                        // block (the running context): one formula
                        const bool tm = em == VP8IR_B_TM_PRED;

                        if (__builtin_amdgcn_ballot_w64(tm) != 0) {
                            const u32 at = bpred ? a0 : aA[k], lt = bpred ? left : lcur;
                            const int tt = bpred ? tl : tlY;
                            const v2s a01 = as_v2s(perm(at, at, 0x0c010c00u)), a23 = as_v2s(perm(at, at, 0x0c030c02u));
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const u32 t = tm_row(a01, a23, (int)((lt >> (8 * jj)) & 0xff) - tt);
                                p[jj] = tm ? t : p[jj];
                            }
                        }
                    }
                    if constexpr (INTER) {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
                            const u32 q = k == 0 ? pr[jj].x : k == 1 ? pr[jj].y : k == 2 ? pr[jj].z : pr[jj].w;
                            p[jj] = is_inter ? q : p[jj];
                        }
                    }
                    u32 o[4] = { p[0], p[1], p[2], p[3] };
                    const bool hasr = (rmg >> k) & 1;
                    if (__builtin_amdgcn_ballot_w64(hasr) != 0) {
                        if (hasr) {
                            u32x4 ra = rr[2 * k], rb = rr[2 * k + 1];
                            if (!((rmf >> k) & 1)) {
                                // a lone DC: (short)(q[0] * dq[0]) (idct_blk.c:34), or the Y2 block's; a1 = (dc + 4) >> 3 on every pixel
                                const u32 y2w = k < 2 ? y2w0 : y2w1;
                                const int raw = (short)(y2w >> (16 * (k & 1)));
                                const int dc = (dc_given & 1) ? raw : (short)__mul24(raw, (int)(cur_dq & 0xffff));
                                const u32 a1 = (u32)((dc + 4) >> 3) & 0xffffu, a2 = a1 | (a1 << 16);
                                ra = rb = (u32x4){ a2, a2, a2, a2 };
                            }
                            add_clamp_rows(p, ra, rb, o);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) orow[jj][k] = o[jj];
                    tl = abv[k] >> 24;
                    abv[k] = o[3];
                    left = right_column(o);
                }
                tlrow = lcur >> 24;
                l0[0] = l0[1]; l0[1] = l0[2]; l0[2] = l0[3];
                nl[0] = nl[1]; nl[1] = nl[2]; nl[2] = nl[3]; nl[3] = left;
                STAMP(4)
                // ---- loop filter: the vertical edges of this block row, the horizontal edge above it; the rows above are final
                const int byr = by < 3 ? by : 2;
                u32 sb[4] = { sF[(4 * byr) * 64], sF[(4 * byr + 1) * 64], sF[(4 * byr + 2) * 64], sF[(4 * byr + 3) * 64] }, d[4][4];
                if (by == 3) { sb[0] = sB[0]; sb[1] = sB[1]; sb[2] = sB[2]; sb[3] = sB[3]; }
                lf_block_row<4>(orow, sb, tF, by == 0, gv, gh, L, d);
                STAMP(5)
                // (launches with inter frames: the next part of the prediction is asked for IN FRONT of the row stores -- it lies in other
                // bytes of the tile than they write --, so that the four youngest memory operations the drain below leaves in flight are
                // always the stores: behind them, the wait for the coefficients was a wait for the stores to be acknowledged as well)
                if constexpr (INTER) {
                    const bool nx_inter = more && ((nx_w0 >> 16) & 3) != VP8IR_INTRA_FRAME;        // (by == 3: nx_w0 is the next macroblock's)
                    if (by < 3 ? is_inter : nx_inter) {
                        const g_u8p pp = by < 3 ? tpc + 64 * (by + 1) : tpc + VP8_TILE_BYTES;
#pragma unroll
                        for (int j = 0; j < 4; j++) pr[j] = *(g_cu32x4p)(pp + 16 * j);
                    }
                }
                const bool first = by == 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32x4 v = first ? (u32x4){ d[j][0], d[j][1], d[j][2], d[j][3] } : (u32x4){ sfix[j], d[j][0], d[j][1], d[j][2] };
                    *(g_u32x4p)(prow + j * pstride) = v ^ VP8_LF_BIAS;
                }
                // the rows just written keep their last dword for the next macroblock's left edge
#pragma unroll
                for (int j = 0; j < 4; j++) { if (act && !first) sF[(4 * by - 4 + j) * 64] = d[j][3]; sfix[j] = sb[j]; }
                prow = KF_ACT(act) ? (first ? tpc + KT_Y_WIN : prow + 64) : (g_u8p)dummy;
                pstride = KF_ACT(act) ? 16 : 0;
                STAMP(10)
                // ---- the next phase's residuals (its coefficients have landed: the four row stores above are younger)
                drain(by < 3 ? 4 * by + 4 : 0, 0, 4);
                STAMP(6)
            }
            // ---- the bottom four rows stay (the lane below finishes them); the left neighbour's are complete now
            {
                u32 (&e)[4][4] = tF;             // rows 12..15 as the last block row left them
#pragma unroll
                for (int j = 0; j < 4; j++) hF[j][3] = sfix[j];
                if (act) {
                    if (hand) *(g_u32x4p)(hpc + KH_Y) = (u32x4){ abv[0], abv[1], abv[2], abv[3] };     // unfiltered bottom line
                    if (write_bottom && c > 0) {     // nobody below takes the left neighbour's bottom rows over (the first lane of the
                                                     // strand will read them back, finish them and write them again)
                        g_u8p pbo = tpc - VP8_TILE_BYTES + KT_Y_BOT;
#pragma unroll
                        for (int j = 0; j < 4; j++) *(g_u32x4p)(pbo + j * 16) = (u32x4){ hF[j][0], hF[j][1], hF[j][2], hF[j][3] } ^ VP8_LF_BIAS;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) { sB[j] = e[j][3]; pb[j][0] = e[j][0]; pb[j][1] = e[j][1]; pb[j][2] = e[j][2]; }
                    if (last_col) {          // end of the row: nobody revisits the last dwords
#pragma unroll
                        for (int y = 0; y < 12; y++) *(GLOBAL_AS u32 *)(tpc + VP8_TILE_BYTES + KT_Y_WIN + 16 * y) = sF[y * 64] ^ VP8_LF_BIAS;
                        if (write_bottom) {
#pragma unroll
                            for (int j = 0; j < 4; j++) *(g_u32x4p)(tpc + KT_Y_BOT + j * 16) = (u32x4){ e[j][0], e[j][1], e[j][2], e[j][3] } ^ VP8_LF_BIAS;
                        }
                    }
                }
            }
            // ---- prediction context of the next step
#pragma unroll
            for (int i = 0; i < 4; i++) { l0[i] = nl[i]; h2[i] = h1[i]; h1[i] = act ? abv[i] : h1[i]; }
            prevLast = aA[3] >> 24;
        } else {
            // ======================= chroma: U, then V =======================
            // The loop body works on "U" (l0[0..1], aA[0..1], pb[j][0], hF[j][0..1], tF[j][0..1], ras); the two planes' state
            // changes places at its end.
            int tlA = top ? 127 : (c == 0 ? 129 : prevLast), tlB = top ? 127 : (c == 0 ? 129 : prevLast2);
            const int lastU = aA[1] >> 24, lastV = aA[3] >> 24;
            u32 bA[2] = { 0, 0 }, bB[2] = { 0, 0 };           // unfiltered bottom lines of the two planes
            if constexpr (INTER) {
                if (is_inter && late) {
                    pr[0] = *(g_cu32x4p)(tpc + 256); pr[1] = *(g_cu32x4p)(tpc + 272);
                    pr[2] = *(g_cu32x4p)(tpc + 320); pr[3] = *(g_cu32x4p)(tpc + 336);
                }
            }
#pragma unroll 1
            for (int pl = 0; pl < 2; pl++) {
                if constexpr (INTER) asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]));
                u32x4 rr[8];
                fetch(rr);
                const u32 rmg = jm >> (4 * pl), rmf = jm >> (16 + 4 * pl);
                // the plane's four lone first coefficients (read before the next macroblock's take their place)
                const u32 cdc0 = s_y2dc[lane * 8 + 2 * pl], cdc1 = s_y2dc[lane * 8 + 2 * pl + 1];
                if (pl == 0) queue(bp, (int)((jm >> 8) & 0x1f) + __builtin_popcount((jm >> 16) & 0xf), (jm >> 20) & 0xf, 0, 0);
                else {
                    u32 n_jm = 0, n_dcg = 0;
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // (the descriptor: U's four row stores are younger)
                    u32x4 m0, m1;
                    u32x4 cd;
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(m0), "=&v"(m1), "=&v"(cd) : "v"(desc_lane) : "memory");
                    const u32x4 z = { 0, 0, 0, 0 };
                    u32 n_dq = 0;
                    if (more) prepare_mb(m0, m1, cd, z, n_jm, n_dcg, n_dq);
                    nx_w0 = m0.x; nx_w1 = m0.y; nx_jm = n_jm; nx_dcg = 0; nx_dq = n_dq;
                    bp += ((int)((jm >> 8) & 0x1f) + __builtin_popcount((jm >> 16) & 0xff)) * 16;
                    queue(bp, (int)((n_jm >> 8) & 0x1f), (n_jm >> 16) & 0xf, 0, 0);
                }
                STAMP(3)
                const u32 aC0 = aA[0], aC1 = aA[1], lC0 = l0[0], lC1 = l0[1];
                int dcC = 128;
                if (up | lf) {
                    const int shift = 2 + up + lf;
                    const int s = (up ? sad4(aC0) + sad4(aC1) : 0) + (lf ? sad4(lC0) + sad4(lC1) : 0);
                    dcC = (s + (1 << (shift - 1))) >> shift;
                }
                const u32 dcsC = perm((u32)dcC, (u32)dcC, 0u);
                u32 bot[2] = { 0, 0 }, rc[2] = { 0, 0 };
                u32 o0[4][2], o1[4][2];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int bx = k & 1, byc = k >> 1;
                    u32 p[4] = { 0, 0, 0, 0 };
                    if (any_intra) mb_mode_pred_sel(uv_mode, bx ? aC1 : aC0, byc ? lC1 : lC0, tlA, dcsC, p);
                    if constexpr (INTER) {
                        const u32x4 ra = byc ? pr[2] : pr[0], rb = byc ? pr[3] : pr[1];
                        p[0] = is_inter ? (bx ? ra.y : ra.x) : p[0]; p[1] = is_inter ? (bx ? ra.w : ra.z) : p[1];
                        p[2] = is_inter ? (bx ? rb.y : rb.x) : p[2]; p[3] = is_inter ? (bx ? rb.w : rb.z) : p[3];
                    }
                    u32 o[4] = { p[0], p[1], p[2], p[3] };
                    const bool hasr = (rmg >> k) & 1;
                    if (__builtin_amdgcn_ballot_w64(hasr) != 0) {
                        if (hasr) {
                            u32x4 ra = rr[2 * k], rb = rr[2 * k + 1];
                            if (!((rmf >> k) & 1)) {          // a lone DC
                                const u32 cw = k < 2 ? cdc0 : cdc1;
                                const int dc = (short)__mul24((int)(short)(cw >> (16 * (k & 1))), (int)(cur_dq & 0xffff));
                                const u32 a1 = (u32)((dc + 4) >> 3) & 0xffffu, a2 = a1 | (a1 << 16);
                                ra = rb = (u32x4){ a2, a2, a2, a2 };
                            }
                            add_clamp_rows(p, ra, rb, o);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) { if (byc) o1[jj][bx] = o[jj]; else o0[jj][bx] = o[jj]; }
                    if (byc) bot[bx] = o[3];
                    if (bx) rc[byc] = right_column(o);
                }
                bA[0] = bot[0]; bA[1] = bot[1]; l0[0] = rc[0]; l0[1] = rc[1];
                STAMP(4)
                // (the other plane's prediction, or the next macroblock's: in front of this plane's stores, see the luma role)
                if constexpr (INTER) {
                    const bool nx_inter = more && ((nx_w0 >> 16) & 3) != VP8IR_INTRA_FRAME;        // (pl == 1: nx_w0 is the next macroblock's)
                    if (pl == 0 ? is_inter : nx_inter) {
                        const g_u8p pp = pl == 0 ? tpc + 288 : tpc + VP8_TILE_BYTES + 256;
                        pr[0] = *(g_cu32x4p)pp; pr[1] = *(g_cu32x4p)(pp + 16);
                        pr[2] = *(g_cu32x4p)(pp + 64); pr[3] = *(g_cu32x4p)(pp + 80);
                    }
                }
                // ---- loop filter of the plane, block row by block row
                const int poff = 32 * pl;
                if (hand) *(g_u32x2p)(hpc + KH_U + 8 * pl) = (u32x2){ bot[0], bot[1] };        // unfiltered bottom line: hand-over
                u32 Pc[4][2];
#pragma unroll
                for (int j = 0; j < 4; j++) { Pc[j][0] = tF[j][0]; Pc[j][1] = tF[j][1]; }
                u32 *const sP = sF + pl * 4 * 64;        // this plane's rows 0..3
                u32 s0[4] = { sP[0], sP[64], sP[128], sP[192] }, s1[4] = { sB[0], sB[1], sB[2], sB[3] }, d0[4][2], d1[4][2];
                lf_block_row<2>(o0, s0, Pc, true, gv, gh, L, d0);
                STAMP(11)
                {   // rows 4..7 of the macroblock above: final
                    g_u8p pa = KF_ACT(act && !top) ? tpc - rowbytes + KT_U_BOT + poff : (g_u8p)dummy;
                    const int st = KF_ACT(act && !top) ? 16 : 0;          // (two 8-byte rows per store)
#pragma unroll
                    for (int j = 0; j < 4; j += 2)
                        *(g_u32x4p)(pa + (j >> 1) * st) = (u32x4){ d0[j][0], d0[j][1], d0[j + 1][0], d0[j + 1][1] } ^ VP8_LF_BIAS;
                }
                STAMP(12)
                lf_block_row<2>(o1, s1, Pc, false, gv, gh, L, d1);
                STAMP(11)
                {   // rows 0..3: the left neighbour's last dword and this macroblock's first
                    g_u8p po = KF_ACT(act) ? tpc + KT_U_WIN + poff : (g_u8p)dummy;
                    const int st = KF_ACT(act) ? 16 : 0;
#pragma unroll
                    for (int j = 0; j < 4; j += 2)
                        *(g_u32x4p)(po + (j >> 1) * st) = (u32x4){ s0[j], d1[j][0], s0[j + 1], d1[j + 1][0] } ^ VP8_LF_BIAS;
                }
                STAMP(12)
                u32 (&e)[4][2] = Pc;             // rows 4..7 as the second block row left them
                // the left neighbour's bottom rows are complete now (a lane that is idle, or first in its row, changed nothing)
#pragma unroll
                for (int j = 0; j < 4; j++) hF[j][1] = s1[j];
                if (act) {
                    if (write_bottom && c > 0) {       // nobody below takes the left neighbour's bottom rows over
                        g_u8p pbo = tpc - VP8_TILE_BYTES + KT_U_BOT + poff;
#pragma unroll
                        for (int j = 0; j < 4; j += 2) *(g_u32x4p)(pbo + j * 8) = (u32x4){ hF[j][0], hF[j][1], hF[j + 1][0], hF[j + 1][1] } ^ VP8_LF_BIAS;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) { sP[j * 64] = d1[j][1]; sB[j] = e[j][1]; pb[j][0] = e[j][0]; }
                    if (last_col) {        // end of the row: nobody revisits the last dword
#pragma unroll
                        for (int j = 0; j < 4; j++) *(GLOBAL_AS u32 *)(tpc + VP8_TILE_BYTES + KT_U_WIN + poff + 8 * j) = d1[j][1] ^ VP8_LF_BIAS;
                        if (write_bottom) {
#pragma unroll
                            for (int j = 0; j < 4; j += 2) *(g_u32x4p)(tpc + KT_U_BOT + poff + j * 8) = (u32x4){ e[j][0], e[j][1], e[j + 1][0], e[j + 1][1] } ^ VP8_LF_BIAS;
                        }
                    }
                }
                // ---- the other plane's residuals (or the next macroblock's first): the four row stores above are younger
                drain(pl == 0 ? 20 : 16, 0, 4);
                STAMP(6)
                // ---- the planes change places
                SWAP_U32(bA[0], bB[0]) SWAP_U32(bA[1], bB[1]) SWAP_U32(l0[0], l0[2]) SWAP_U32(l0[1], l0[3])
                SWAP_U32(aA[0], aA[2]) SWAP_U32(aA[1], aA[3])
                { const int t_ = tlA; tlA = tlB; tlB = t_; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    SWAP_U32(tF[j][0], tF[j][2]) SWAP_U32(tF[j][1], tF[j][3]) SWAP_U32(hF[j][0], hF[j][2]) SWAP_U32(hF[j][1], hF[j][3])
                    SWAP_U32(pb[j][0], pb[j][1]) SWAP_U32(sB[j], sB2[j])
                }
            }
            // ---- prediction context of the next step
            h2[0] = h1[0]; h2[1] = h1[1]; h2[2] = h1[2]; h2[3] = h1[3];
            h1[0] = act ? bA[0] : h1[0]; h1[1] = act ? bA[1] : h1[1]; h1[2] = act ? bB[0] : h1[2]; h1[3] = act ? bB[1] : h1[3];
            prevLast = lastU; prevLast2 = lastV;
        }
        if (act) mbp += VP8IR_MBX_WORDS;
        p_more = more;
        STAMP(7)
    }
#ifdef VP8_STAMPS
    if (wave == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < VP8_NSTAMPS; i_++) atomicAdd(LUMA ? &vp8_stamps_recon[i_] : &vp8_stamps_lf[i_], st_acc[i_]); }
#endif
}

// One kernel, two roles.  A wave decides at run time whether it reconstructs luma or chroma, so that the two waves that
// share a SIMD always play different roles: launched as two kernels side by side, nothing keeps the dispatcher from putting
// two luma waves (or two chroma waves) on one SIMD and leaving another with a single wave, and a process in which it did
// took 62 ms per 8192 frames instead of 49.  The first wave to arrive on a SIMD (an agent-scope counter per SIMD, indexed
// by XCC / SE / SH / CU / SIMD id) takes luma, the second chroma, the third luma again ...; which STRANDS it works on comes
// from one counter per role (if its role has run out of work it takes the other).  grid = 2 * nwaves.
//   sched[0], sched[1]: next luma / chroma work item (zeroed before every launch); sched[16 + simd]: waves seen (never reset:
//   only the parity matters).
// Who is on a SIMD right now: luma waves in bits 15:0, chroma waves in bits 31:16, one word per SIMD of the device, for ALL launches
// of the process (every context, every stream: kernels of two contexts run side by side, and each must see the other's waves).
__device__ unsigned int vp8_simd_roles[16384];

template <bool INTER>
__device__ __forceinline__ void kf_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy,
                                          unsigned int *sched, int nwaves)
{
    __shared__ __attribute__((aligned(16))) u32 s_stage[4 * 2 * 64 * 4];
    __shared__ unsigned short s_queue[256];
    __shared__ u32 s_tab[64];
    __shared__ __attribute__((aligned(16))) u32 s_y2dc[64 * 8];
    __shared__ __attribute__((aligned(16))) u32 s_desc[5 * 64 * 4];
    __shared__ u32 s_sf[12 * 64];
    __shared__ __attribute__((aligned(16))) u32 s_psel[PSEL_MODES * PSEL_WORDS];
    for (int i = threadIdx.x; i < PSEL_MODES * PSEL_WORDS; i += 64) s_psel[i] = k_pred_sel[i];
    int role = 0, item = 0;
    // (the SIMD's word is looked up again when the wave leaves: a wave stays where it is, and the kernel has no register to carry it in)
    auto simd_word = [](u32 &hw, u32 &xcc) {
        hw = __builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_REG_HW_ID */);
        xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20 /* HW_REG_XCC_ID */);
        // SIMD_ID [5:4], CU_ID [11:8], SH_ID [12], SE_ID [15:13]; the pipe / queue fields in between differ from stream to stream
        return vp8_simd_roles + (((hw >> 4) & 3) | (((hw >> 8) & 0xff) << 2) | ((xcc & 15) << 10));
    };
    if (threadIdx.x == 0) {
        u32 hw, xcc;
        unsigned int *const here = simd_word(hw, xcc);
        // the wave joins its SIMD as the role the SIMD holds fewer of (an empty SIMD: luma) ...
        unsigned int seen = __hip_atomic_load(here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        do role = (seen & 0xffffu) <= (seen >> 16) ? 0 : 1;
        while (!__hip_atomic_compare_exchange_strong(here, &seen, seen + (role ? 0x10000u : 1u), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        item = (int)__hip_atomic_fetch_add(&sched[role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (item >= nwaves) {       // ... unless that role has run out of work in this launch
            (void)__hip_atomic_fetch_add(here, role ? 1u - 0x10000u : 0x10000u - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            role ^= 1;
            item = (int)__hip_atomic_fetch_add(&sched[role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef VP8_STAMPS       // diagnostic builds: where every wave ran and as what
        sched[16 + 16384 + 4 * blockIdx.x] = hw; sched[16 + 16384 + 4 * blockIdx.x + 1] = xcc;
        sched[16 + 16384 + 4 * blockIdx.x + 2] = (u32)role | (seen << 8); sched[16 + 16384 + 4 * blockIdx.x + 3] = (u32)item;
#endif
    }
    role = __builtin_amdgcn_readfirstlane(role);
    item = __builtin_amdgcn_readfirstlane(item);
    // (leaving: the SIMD's count of this role goes down again)
    auto leave = [&]() {
        u32 hw, xcc;
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(simd_word(hw, xcc), role ? 0u - 0x10000u : 0u - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (item >= nwaves) { leave(); return; }         // (cannot happen with grid = 2 * nwaves)
#ifdef KF_ONLY_ROLE     // (timing experiments only, the frames are wrong: one role's waves alone on their SIMDs -- DESIGN 8's issue model)
    if (role != KF_ONLY_ROLE) { leave(); return; }
#endif
#ifdef VP8_STAMPS
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // The luma wave of a SIMD is the longer one (twice the pixels: 80 M cycles against the chroma wave's 64 M per 8192 key frames,
    // 50 M against 31 M in launches with inter frames; tools/wave_times.py) and so is what the kernel takes: where the two compete
    // for an issue slot it goes first.  -4 % (key frames) / -6 % (inter frames) of the kernel's time, mostly off the slowest
    // luma waves.
#ifndef KF_LUMA_PRIO
#define KF_LUMA_PRIO 3
#endif
    if (role == 0) __builtin_amdgcn_s_setprio(KF_LUMA_PRIO);
    if (role == 0) kf_body<true, INTER>(jobs, njobs, g, lgG, P, nstrands, dummy, item, s_stage, s_queue, s_tab, s_y2dc, s_desc, s_sf, s_psel);
    else kf_body<false, INTER>(jobs, njobs, g, lgG, P, nstrands, dummy + 1024, item, s_stage, s_queue, s_tab, s_y2dc, s_desc, s_sf, s_psel);
    leave();
#ifdef VP8_STAMPS       // ... and for how long (units of 1024 cycles, above the work item's ten bits)
    if (threadIdx.x == 0) sched[16 + 16384 + 4 * blockIdx.x + 3] = (u32)item | ((u32)((__builtin_amdgcn_s_memtime() - t_begin) >> 10) << 10);
#endif
}

extern "C" __global__ void __launch_bounds__(64, 2)
vp8_keyframe_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy,
                    unsigned int *sched, int nwaves)
{
    kf_kernel<false>(jobs, njobs, g, lgG, P, nstrands, dummy, sched, nwaves);
}

// The same for launches with inter frames among them: a macroblock with a reference frame takes its prediction from its tile, where
// vp8_inter_pred_kernel (vp8_inter_pred.hip; launched in front of this kernel) left it, instead of predicting from its neighbours;
// residual, loop filter (levels by reference frame and mode, the inter frames' hev thresholds) and output are the key frames'.
// Intra macroblocks of inter frames, and whole key frames in such a launch, go the key-frame way.
extern "C" __global__ void __launch_bounds__(64, 2)
vp8_interframe_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy,
                      unsigned int *sched, int nwaves)
{
    kf_kernel<true>(jobs, njobs, g, lgG, P, nstrands, dummy, sched, nwaves);
}
