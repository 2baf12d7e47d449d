// Inter prediction of a launch's inter macroblocks, every one on its own, for gfx950: the first of the two kernels that decode
// large launches with inter frames (vp8_interframe_kernel, vp8_keyframe_simt.hip, is the second).
//
// What it replaces (paths relative to the reference tree): vp8_build_inter_predictors_mb (vp8/common/reconinter.c:560-606) --
// vp8_build_inter16x16_predictors_mb (:384-441), build_inter4x4_predictors_mb (:443-518), build_4x4uvmvs (:520-558), the MV
// clamps (:348-382) -- and behind them the sub-pixel filters of vp8/common/filter.c (six-tap :41-128, bilinear :376-397) as
// decode_macroblock (vp8/decoder/decodframe.c:112-296) calls them for a macroblock with ref_frame != INTRA_FRAME.
//
// An inter macroblock's prediction needs nothing of the frame being decoded, only the (finished, border-extended) reference
// frames, so this kernel is order-free and its lanes are cut to the filter, not to the macroblock order:
//   * a lane owns a COLUMN STRIP eight pixels wide -- half of a luma macroblock's 16 rows, a chroma plane's 8 rows -- or one 4x4
//     block of a SPLITMV macroblock, and streams down the source rows: one global_load_dwordx4 (4x4 blocks: dwordx3) per row (the
//     13 / 9 pixels the six taps need, as aligned dwords shifted into place), requested six rows ahead; the horizontal pass two
//     pixels per instruction on 16-bit lanes; a ring of the last six filtered rows in registers; the vertical pass.  21 source
//     rows for 16 output rows (the wave-per-row kernels' 4-row segments need 9 for 4), no LDS.  Output goes out in 16-byte
//     stores -- the two lanes of a luma macroblock swap half rows by DPP and store a whole row each, a chroma lane stores two
//     rows at once: as 8-byte stores (19 write requests per macroblock at the L2 instead of 6) they were 39 % of the kernel;
//   * both passes always run, as in the reference's C filters (filter.c:95-128): a whole-pixel offset has the taps {0,0,128,0,0,0},
//     which make a pass the identity, and the bilinear filters of versions 1-3 are the six-tap arithmetic with the taps
//     {0,0,128-16f,16f,0,0} (the same sums, the same rounding; the clamp never binds) -- one code path for every profile;
//   * a wave takes 64 consecutive macroblocks of a frame, sorts them by ballot into those with one motion vector -- with a fraction
//     (filtered) or without (copied: the reference's vp8_copy_mem branch, reconinter.c:402-417; round 6), luma and chroma apart -- and
//     the SPLITMV ones (intra macroblocks drop out), and works through the lists 32 (luma strips; then chroma strips) resp. 4 / 8
//     macroblocks (luma / chroma 4x4 blocks) at a time, all lanes busy on the same code;
//   * the reference frames are read as border-extended raster frames (vp8_inter_pred_kernel) or as the tiles a large launch left
//     (vp8_inter_pred_tiles_kernel): there the macroblocks with one vector fetch their source windows TOGETHER, four lanes a 64-byte
//     sector, and hand the rows -- normalised, edges replicated -- to the filtering lanes through LDS (WinGeo; round 6), the 4x4
//     blocks of SPLITMV macroblocks read per lane (TileSrc; round 5).
//
// Output: the prediction of macroblock (r, c) in ITS tile of the job's scratch frame (DevJob::tile, the key-frame kernels'
// rows x (cols + 1) tiles of VP8_TILE_BYTES): luma rows at 16 y; chroma in the arrangement the tiles' chroma has -- U rows 0..3
// at 256 + 8 y, V rows 0..3 at 288 + 8 y, U rows 4..7 at 320 + 8 (y - 4), V rows 4..7 at 352 + 8 (y - 4) -- so that
// vp8_interframe_kernel, which reads a block row's prediction right before it writes finished pixels into the same tile, never
// overwrites what it has not read yet.  Integer only; no MFMA by design.
#include "vp8_block_prims.hip.h"
#include <type_traits>
#include <utility>
#ifndef IP_AHEAD
#define IP_AHEAD 6
#endif
#ifndef IP_WAVES
#define IP_WAVES 4        // (five or six waves per SIMD only fit with spills: 9.9 / 10.0 ms against 9.1-9.6)
#endif
#ifndef IP_WAVES_TILES
#define IP_WAVES_TILES 3  // (the tile reader: 12 KB of LDS a wave for the rows on their way from loading to filtering lanes)
#endif

namespace {

typedef GLOBAL_AS const unsigned int *g_cmvp;       // vp8ir_mv {int16 row, col} read as one dword
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
typedef GLOBAL_AS const u32x3 *g_cu32x3p;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(4)));          // sixteen bytes at a four-byte boundary
typedef GLOBAL_AS const u32x4_u *g_cu32x4up;
typedef GLOBAL_AS const u32x4 *g_cu32x4p;

// filter taps as (t, t) pairs of 16-bit lanes: [0] six-tap (filter.c:27-39), [1] bilinear (filter.c:16-26) on the six-tap grid
#define T2_(a) (((u32)(unsigned short)(a)) * 0x10001u)
#define BIL_(f) { 0u, 0u, T2_(128 - 16 * (f)), T2_(16 * (f)), 0u, 0u }
__constant__ static const u32 k_taps2[2][8][6] = {
    { { T2_(0), T2_(0), T2_(128), T2_(0), T2_(0), T2_(0) }, { T2_(0), T2_(-6), T2_(123), T2_(12), T2_(-1), T2_(0) },
      { T2_(2), T2_(-11), T2_(108), T2_(36), T2_(-8), T2_(1) }, { T2_(0), T2_(-9), T2_(93), T2_(50), T2_(-6), T2_(0) },
      { T2_(3), T2_(-16), T2_(77), T2_(77), T2_(-16), T2_(3) }, { T2_(0), T2_(-6), T2_(50), T2_(93), T2_(-9), T2_(0) },
      { T2_(1), T2_(-8), T2_(36), T2_(108), T2_(-11), T2_(2) }, { T2_(0), T2_(-1), T2_(12), T2_(123), T2_(-6), T2_(0) } },
    { BIL_(0), BIL_(1), BIL_(2), BIL_(3), BIL_(4), BIL_(5), BIL_(6), BIL_(7) }
};
#undef BIL_
#undef T2_

struct Taps { v2u16 t[6]; };
__device__ __forceinline__ Taps load_taps(int bil, int f)
{
    Taps r;
    const u32 *p = k_taps2[bil][f];
#pragma unroll
    for (int k = 0; k < 6; k++) r.t[k] = __builtin_bit_cast(v2u16, p[k]);
    return r;
}

// four pixels of a filtered row as two pairs of 16-bit lanes, clamped to 0..255
struct Row4 { v2u16 a, b; };
// A pass's sum lies in -8160 .. 40864: biased by 8192 it is an unsigned 16-bit number, v_pk_mad_u16 wraps exactly, and
// (t + 8192) >> 7 == (t >> 7) + 64; a saturating -64 and a min 255 are the clamp (vp8_block_prims.hip.h).
__device__ __forceinline__ v2u16 finish2(v2u16 acc)
{
    const v2u16 c64 = { 64, 64 }, c255 = { 255, 255 };
    return __builtin_elementwise_min(__builtin_elementwise_sub_sat(acc >> 7, c64), c255);
}
// first pass (filter_block2d_first_pass, filter.c:41-79) for four output pixels: d = the three aligned dwords that hold the
// nine pixels from two left of the first output pixel on, sh = the byte offset of that pixel in d.x
__device__ __forceinline__ Row4 h_pass(u32x3 d, u32 sh, const Taps &tx)
{
    auto asv = [](u32 v) { return __builtin_bit_cast(v2u16, v); };
    const u32 w0 = __builtin_amdgcn_alignbyte(d.y, d.x, sh), w1 = __builtin_amdgcn_alignbyte(d.z, d.y, sh);
    const u32 w2 = __builtin_amdgcn_alignbyte(0u, d.z, sh);
    // P[k] = pixels (k, k+1) of the row, one per 16-bit lane
    const v2u16 P[8] = { asv(perm(w0, w0, 0x0c010c00u)), asv(perm(w0, w0, 0x0c020c01u)), asv(perm(w0, w0, 0x0c030c02u)),
                         asv(perm(w1, w0, 0x0c040c03u)), asv(perm(w1, w1, 0x0c010c00u)), asv(perm(w1, w1, 0x0c020c01u)),
                         asv(perm(w1, w1, 0x0c030c02u)), asv(perm(w2, w1, 0x0c040c03u)) };
    const v2u16 bias = { 64 + 8192, 64 + 8192 };
    v2u16 a01 = bias, a23 = bias;
#pragma unroll
    for (int k = 0; k < 6; k++) { a01 += P[k] * tx.t[k]; a23 += P[k + 2] * tx.t[k]; }
    return { finish2(a01), finish2(a23) };
}
// second pass (filter_block2d_second_pass, filter.c:81-128): H[0..5] = the filtered source rows -2 .. +3 of the output row
__device__ __forceinline__ u32 v_pass(const Row4 &h0, const Row4 &h1, const Row4 &h2, const Row4 &h3, const Row4 &h4, const Row4 &h5,
                                      const Taps &ty)
{
    const v2u16 bias = { 64 + 8192, 64 + 8192 };
    const v2u16 a01 = bias + h0.a * ty.t[0] + h1.a * ty.t[1] + h2.a * ty.t[2] + h3.a * ty.t[3] + h4.a * ty.t[4] + h5.a * ty.t[5];
    const v2u16 a23 = bias + h0.b * ty.t[0] + h1.b * ty.t[1] + h2.b * ty.t[2] + h3.b * ty.t[3] + h4.b * ty.t[4] + h5.b * ty.t[5];
    return perm(__builtin_bit_cast(u32, finish2(a23)), __builtin_bit_cast(u32, finish2(a01)), 0x06040200u);
}

// ---------------------------------------------------------------------------------------------------------------------
// Where a strip's source rows come from.  A strip asks its source for row i twice: issue(i) requests it -- AHEAD rows before it
// is filtered (the compiler would not move a load above the store of the output row before it -- for all it knows they alias --
// and one row in flight at a time is a latency chain) -- and finish(raw, i) hands out the aligned dwords that hold the row's
// pixels from byte `sh` of the first one on.
//
// RASTER: the reference's own frame buffer (yv12config.c:55-112) with its borders extended (extend.c, onyxd_if.c:607): a row is
// one load.  What vp8_build_inter_predictors_mb reads (xd->pre, reconinter.c:393-441).
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)          // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>()); static_for<N, I + 1>(f); }
}
template <int NDW> struct VecOf;
template <> struct VecOf<4> { typedef u32x4 T; typedef u32x4_u TU; };
template <> struct VecOf<3> { typedef u32x3 T; typedef u32x3 TU; };
template <int NDW>
struct RasterSrc {
    typedef typename VecOf<NDW>::T Vec;
    typedef typename VecOf<NDW>::TU Raw;
    static constexpr bool PIN = false;
    g_cu8p rp; int stride; u32 sh;
    // src = the pixel two left of and two above the strip's first one in the reference plane
    __device__ __forceinline__ RasterSrc(g_cu8p src, int stride_) : stride(stride_) { sh = (u32)(unsigned long)src & 3u; rp = src - sh; }
    __device__ __forceinline__ Raw issue(int i) const { return *(GLOBAL_AS const Raw *)(rp + (long)i * stride); }
    __device__ __forceinline__ Vec finish(const Raw &r, int) const { return r; }
};

// TILES, for the 4x4 blocks of SPLITMV macroblocks (macroblocks with one vector fetch their windows together: WinGeo, further down).
// The reference frame as a large launch left it (macroblock-window tiles, vp8_keyframe_simt.hip KT_*; round 5) -- n streams
// decoded in lock step predict every launch from what the launch before wrote.  A tile row piece is 16 (chroma: 8) bytes of ONE
// pixel row, the tile of the next macroblock 384 bytes on; rows 0..11 (chroma 0..3) of a macroblock row stand in the macroblocks'
// WINDOWS, four pixels to the left (pixel x at byte (x + 4) & 15 of tile (x + 4) >> 4), rows 12..15 (4..7) macroblock-aligned.  So a
// row of a block's strip -- nine pixels from any x -- is two loads, the tail of one tile's piece and the head of the next tile's,
// put together by selects, and which of the two column arrangements applies is decided row by row.  There are no borders: rows and
// columns beyond the frame are the frame's last (vp8_extend_mb_row / vp8_yv12_extend_frame_borders replicate, extend.c:160-185,
// yv12extend.c:24-90) -- a clamped row index, and for the few strips that reach past the left or right edge a fix-up of the loaded
// bytes.  Motion vectors that point further out than any border (no conforming stream has them: reconinter.c:348-382 clamps) read
// replicated pixels where the raster kernel reads whatever the clamped address holds: both are memory-safe, neither is specified.
template <bool CHROMA, int NDW>
struct TileSrc {
    static_assert(NDW == 3, "strips eight pixels wide go through WinGeo");
    typedef typename VecOf<NDW>::T Vec;
    static constexpr int NPX = 9;                               // pixels a row needs
    static constexpr int LGR = CHROMA ? 3 : 4, RMASK = (1 << LGR) - 1, BOT0 = CHROMA ? 4 : 12;     // pixel rows per tile row; first macroblock-aligned one
    struct Raw { u32 e[CHROMA ? 5 : 8]; int jsel; };
    // (the strip keeps the scheduler from moving a row's requests: with two loads and nine registers a row in flight it sinks them down to
    // their use to save registers, and every row then waits for its own loads)
    static constexpr bool PIN = true;
    g_cu8p rowp;            // TWO tiles before the plane's first row piece in the first tile of the tile row the strip begins in
    int yy0, tlo, thi;      // the strip's first row within its tile row; rows above the plane's first / below its last count as those
    u32 rowbytes;           // bytes per tile row
    u32 colW, colB; int jW, jB;     // window rows / bottom rows: byte offset of the first load from rowp, the first dword wanted within the piece
    u32 sh;
    u32 em[NDW], eselA, eselB;      // bytes to replace (per dword); v_perm selectors that splat the edge pixel out of dwords (0, 1) / (2)
    // tiles: the frame's tiles + the plane's offset in a tile (0, 256 U, 288 V); W, H: the plane's size; (x0, y0): the strip's first pixel
    __device__ __forceinline__ TileSrc(g_cu8p tiles, int cols, int W, int H, int x0, int y0)
    {
        rowbytes = (u32)(cols + 1) * VP8_TILE_BYTES;
        const int yc = max(0, min(y0, H - 1));
        const int ybase = yc & ~RMASK;
        yy0 = y0 - ybase; tlo = -ybase; thi = H - 1 - ybase;     // row i is row clamp(yy0 + i, tlo, thi) counted from the tile row's first
        rowp = tiles + (long)(yc >> LGR) * rowbytes - 2 * VP8_TILE_BYTES;
        x0 = max(-64, min(x0, W + 64));
        const int xl = max(min(x0, W - 1), 1 - NPX);            // what is loaded: NPX bytes from xl on, at least one of them inside
        sh = (u32)xl & 3u;
        constexpr int PW = CHROMA ? 8 : 16, LG = CHROMA ? 3 : 4;
        const int xs = xl + 4;
        // (+ 2 tiles: rowp stands two tiles early, so that the offsets are never negative -- a strip that begins twelve pixels left of
        // the frame has its first chroma piece in tile -2, and the loads are in bounds: VP8HIP_TILE_FRONT bytes lie in front of the pool)
        jW = (xs & (PW - 1)) >> 2; colW = (u32)(((xs >> LG) + 2) * VP8_TILE_BYTES);
        jB = (xl & (PW - 1)) >> 2; colB = (u32)(((xl >> LG) + 2) * VP8_TILE_BYTES + (CHROMA ? 32 : 0));     // (chroma rows 4..7: 288 + 8 yy)
        const bool edge = x0 < 0 || x0 + NPX > W;
        eselA = eselB = 0x0c0c0c0cu;
#pragma unroll
        for (int k = 0; k < NDW; k++) em[k] = 0;
        if (__builtin_amdgcn_ballot_w64(edge) != 0) {
            // byte t of the loaded dwords is pixel x0 + t - sh (xl + t - sh where that differs, all of them replaced then)
            const bool left = x0 < 0;
            const int n = left ? min(16, (int)sh - x0) : max(0, W - x0 + (int)sh);      // left: bytes [0, n) replaced; right: [n, 16)
            const int ed = (left ? -xl : W - 1 - xl) + (int)sh;                            // where pixel 0 / W - 1 lies
            // (a v_perm selector byte 0x0c gives 0: the splat comes out of one pair of dwords, zeros out of the other)
            const u32 spl = (u32)(ed & 7) * 0x01010101u;
            eselA = ed < 8 ? spl : 0x0c0c0c0cu; eselB = ed < 8 ? 0x0c0c0c0cu : spl;
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                const int nk = max(0, min(4, n - 4 * k));
                const u32 low = nk >= 4 ? 0xffffffffu : (1u << (8 * nk)) - 1u;
                em[k] = edge ? (left ? low : ~low) : 0u;
            }
        }
    }
    __device__ __forceinline__ Raw issue(int i) const
    {
        const int t = max(tlo, min(yy0 + i, thi));              // row i: t >> LGR tile rows on, row t & RMASK of it
        const u32 yy = (u32)t & RMASK;
        const bool bot = yy >= BOT0;
        const u32 off = __umul24((u32)t >> LGR, rowbytes) + (bot ? colB : colW) + (yy << LGR);
        g_cu8p p = rowp + off;
        Raw r;
        r.jsel = bot ? jB : jW;
        if constexpr (!CHROMA) {
            // this tile's row piece and what can be wanted of the next one's: dwords 0 .. jsel + NDW - 1 <= 5 of the eight.
            // (Not a dword more: a component of a load's result that nothing reads is a register the allocator hands out again
            // while the load is in flight, and the compiler then waits for the load -- for ALL loads -- right behind it.)
            const u32x4 a = *(g_cu32x4p)p;
#pragma unroll
            for (int k = 0; k < 4; k++) r.e[k] = a[k];
            const u32x2 b = *(GLOBAL_AS const u32x2 *)(p + VP8_TILE_BYTES);
            r.e[4] = b.x; r.e[5] = b.y; r.e[6] = r.e[7] = 0;
        } else {
            // the piece and the next tile's piece
            typedef GLOBAL_AS const u32x2 *g_cu32x2p;
            const u32x2 a = *(g_cu32x2p)p, b = *(g_cu32x2p)(p + VP8_TILE_BYTES);
            r.e[0] = a.x; r.e[1] = a.y; r.e[2] = b.x; r.e[3] = b.y; r.e[4] = 0u;
        }
        return r;
    }
    __device__ __forceinline__ Vec finish(const Raw &r, int) const
    {
        Vec d;
        if constexpr (!CHROMA) {
            // dwords jsel .. jsel + NDW - 1 of the eight: a shifter of two stages, as bit selects under masks (written as
            // `jsel & 2 ? e[k + 2] : e[k]` the compiler sees e[jsel + k] and puts the eight dwords in scratch to index them)
            const u32 m2 = (u32)__builtin_amdgcn_sbfe((int)r.jsel, 1, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)r.jsel, 0, 1);
            u32 f[NDW + 1];
#pragma unroll
            for (int k = 0; k < NDW + 1; k++) f[k] = (r.e[k + 2] & m2) | (r.e[k] & ~m2);
#pragma unroll
            for (int k = 0; k < NDW; k++) d[k] = (f[k + 1] & m1) | (f[k] & ~m1);
        } else {
            // dwords jsel .. of the five (jsel: 0 or 1)
            const u32 m1 = (u32)-(int)r.jsel;
#pragma unroll
            for (int k = 0; k < NDW; k++) d[k] = (r.e[k + 1] & m1) | (r.e[k] & ~m1);
        }
        // the edge pixel out of whichever dword holds it, four times, into the bytes beyond the edge.  Unconditional -- the masks of
        // a lane that reaches past no edge are empty --: a branch in the row loop, even a wave-uniform one without a load in it,
        // makes the compiler wait for EVERY load in flight at each row (s_waitcnt vmcnt(0)), and the strip is a latency chain
        const u32 ev = perm(d[1], d[0], eselA) | perm(0u, d[2], eselB);
#pragma unroll
        for (int k = 0; k < NDW; k++) d[k] = (d[k] & ~em[k]) | (ev & em[k]);
        return d;
    }
};

// A column strip of NOUT rows, four pixels wide: dst / dstride = where output row y goes (rows 4.. of an 8-row strip at
// dst + dhalf + dstride * (y - 4): the chroma layout)
template <int NOUT, class SRC>
__device__ __forceinline__ void strip(const SRC &src, const Taps &tx, const Taps &ty, g_u8p dst, int dstride, int dhalf)
{
    // (the tile reader's rows are two loads and nine registers each: five of them in flight keep the kernel, whose passes for one
    // motion vector want their registers for other things, from spilling)
    constexpr int NIN = NOUT + 5, WANT = SRC::PIN ? 5 : 8, AHEAD = NIN < WANT ? NIN : WANT;
    typename SRC::Raw q[AHEAD];
#pragma unroll
    for (int i = 0; i < AHEAD; i++) q[i] = src.issue(i);
    Row4 H[6];
#pragma unroll
    for (int i = 0; i < NIN; i++) {
        const u32x3 d = src.finish(q[i % AHEAD], i);
        if (i + AHEAD < NIN) q[i % AHEAD] = src.issue(i + AHEAD);
        if constexpr (SRC::PIN) __builtin_amdgcn_sched_barrier(0);
        H[i % 6] = h_pass(d, src.sh, tx);
        if (i >= 5) {
            const int y = i - 5;
            const u32 o = v_pass(H[(i + 1) % 6], H[(i + 2) % 6], H[(i + 3) % 6], H[(i + 4) % 6], H[(i + 5) % 6], H[i % 6], ty);
            g_u8p qd = (NOUT == 8 && y >= 4) ? dst + dhalf + dstride * (y - 4) : dst + dstride * y;
            *(GLOBAL_AS u32 *)qd = o;
        }
    }
}

// The same for a strip EIGHT pixels wide (the texture addresser is what bounds this kernel, and it works by the instruction:
// half as many loads and stores per macroblock): the 13 pixels a row needs lie in four aligned dwords
struct Row8 { Row4 l, r; };
__device__ __forceinline__ Row8 h_pass8(u32x4 d, u32 sh, const Taps &tx)
{
    auto asv = [](u32 v) { return __builtin_bit_cast(v2u16, v); };
    const u32 w0 = __builtin_amdgcn_alignbyte(d.y, d.x, sh), w1 = __builtin_amdgcn_alignbyte(d.z, d.y, sh);
    const u32 w2 = __builtin_amdgcn_alignbyte(d.w, d.z, sh), w3 = __builtin_amdgcn_alignbyte(0u, d.w, sh);
    const v2u16 P[12] = { asv(perm(w0, w0, 0x0c010c00u)), asv(perm(w0, w0, 0x0c020c01u)), asv(perm(w0, w0, 0x0c030c02u)),
                          asv(perm(w1, w0, 0x0c040c03u)), asv(perm(w1, w1, 0x0c010c00u)), asv(perm(w1, w1, 0x0c020c01u)),
                          asv(perm(w1, w1, 0x0c030c02u)), asv(perm(w2, w1, 0x0c040c03u)), asv(perm(w2, w2, 0x0c010c00u)),
                          asv(perm(w2, w2, 0x0c020c01u)), asv(perm(w2, w2, 0x0c030c02u)), asv(perm(w3, w2, 0x0c040c03u)) };
    const v2u16 bias = { 64 + 8192, 64 + 8192 };
    v2u16 a01 = bias, a23 = bias, a45 = bias, a67 = bias;
#pragma unroll
    for (int k = 0; k < 6; k++) { a01 += P[k] * tx.t[k]; a23 += P[k + 2] * tx.t[k]; a45 += P[k + 4] * tx.t[k]; a67 += P[k + 6] * tx.t[k]; }
    return { { finish2(a01), finish2(a23) }, { finish2(a45), finish2(a67) } };
}
// ... the same from dwords whose first byte IS the row's first pixel (the LDS-fed strips: WinGeo below normalises the rows)
__device__ __forceinline__ Row8 h_pass8n(u32x4 d, const Taps &tx)
{
    auto asv = [](u32 v) { return __builtin_bit_cast(v2u16, v); };
    const u32 w0 = d.x, w1 = d.y, w2 = d.z, w3 = d.w;
    const v2u16 P[12] = { asv(perm(w0, w0, 0x0c010c00u)), asv(perm(w0, w0, 0x0c020c01u)), asv(perm(w0, w0, 0x0c030c02u)),
                          asv(perm(w1, w0, 0x0c040c03u)), asv(perm(w1, w1, 0x0c010c00u)), asv(perm(w1, w1, 0x0c020c01u)),
                          asv(perm(w1, w1, 0x0c030c02u)), asv(perm(w2, w1, 0x0c040c03u)), asv(perm(w2, w2, 0x0c010c00u)),
                          asv(perm(w2, w2, 0x0c020c01u)), asv(perm(w2, w2, 0x0c030c02u)), asv(perm(w3, w2, 0x0c040c03u)) };
    const v2u16 bias = { 64 + 8192, 64 + 8192 };
    v2u16 a01 = bias, a23 = bias, a45 = bias, a67 = bias;
#pragma unroll
    for (int k = 0; k < 6; k++) { a01 += P[k] * tx.t[k]; a23 += P[k + 2] * tx.t[k]; a45 += P[k + 4] * tx.t[k]; a67 += P[k + 6] * tx.t[k]; }
    return { { finish2(a01), finish2(a23) }, { finish2(a45), finish2(a67) } };
}
// What a strip eight pixels wide does with its filtered source rows, one after the other: the ring of the last six, from the sixth
// on an output row, and the stores.  Output in 16-byte stores (8-byte ones -- 19 write requests per macroblock at the L2 instead of
// 6 -- cost the kernel 39 % of its time).  NOUT == 16, luma: dst = the macroblock's tile, lanes s = 0 / 1 hold the left / right half of
// its rows; after every second row they swap a half row (DPP), lane 0 stores the whole row y - 1 and lane 1 the whole row y.
// NOUT == 8, chroma: dst = the plane's rows 0..3 (rows 4..7 64 bytes on), a lane has whole rows: two of them are one store.
template <int NOUT>
struct Strip8Rows {
    Row8 H[6];
    u32x2 prev;
    g_u8p dst; int s;
    __device__ __forceinline__ Strip8Rows(g_u8p dst_, int s_) : dst(dst_), s(s_) { prev = (u32x2){ 0, 0 }; }
    // source row I, filtered horizontally
    template <int I>
    __device__ __forceinline__ void feed(const Row8 &h, const Taps &ty)
    {
        H[I % 6] = h;
        if constexpr (I >= 5) {
            const Row8 &h0 = H[(I + 1) % 6], &h1 = H[(I + 2) % 6], &h2 = H[(I + 3) % 6], &h3 = H[(I + 4) % 6], &h4 = H[(I + 5) % 6], &h5 = H[I % 6];
            u32x2 o;
            o.x = v_pass(h0.l, h1.l, h2.l, h3.l, h4.l, h5.l, ty);
            o.y = v_pass(h0.r, h1.r, h2.r, h3.r, h4.r, h5.r, ty);
            emit<I - 5>(o);
        }
    }
    // output row y of the strip, eight pixels
    template <int y>
    __device__ __forceinline__ void emit(const u32x2 o)
    {
        if constexpr (!(y & 1)) prev = o;
        else if constexpr (NOUT == 16) {
            const u32x2 give = s ? prev : o;
            const u32 tx_ = dpp_xor1(give.x), ty_ = dpp_xor1(give.y);
            const u32x4 v = s ? (u32x4){ tx_, ty_, o.x, o.y } : (u32x4){ prev.x, prev.y, tx_, ty_ };
            *(GLOBAL_AS u32x4 *)(dst + 16 * (y - 1 + s)) = v;
        } else {
            g_u8p qd = y - 1 >= 4 ? dst + 64 + 8 * (y - 1 - 4) : dst + 8 * (y - 1);
            *(GLOBAL_AS u32x4 *)qd = (u32x4){ prev.x, prev.y, o.x, o.y };
        }
    }
};
// A strip of a macroblock whose motion vector has no fraction: the reference copies (vp8_copy_mem16x16 / vp8_copy_mem8x8,
// reconinter.c:22-110, chosen at :402-417 by `mv.as_int & 0x00070007`) where its filters would multiply by {0, 0, 128, 0, 0, 0} --
// sixteen (eight) rows in, no halo, no arithmetic.  src = the strip's first pixel in a raster plane.
template <int NOUT>
__device__ __forceinline__ void copy_strip8(g_cu8p src, int stride, g_u8p dst, int s)
{
    const u32 sh = (u32)(unsigned long)src & 3u;
    g_cu8p rp = src - sh;
    constexpr int AHEAD = 8;
    u32x3 q[AHEAD];
#pragma unroll
    for (int y = 0; y < AHEAD; y++) q[y] = *(GLOBAL_AS const u32x3 *)(rp + (long)y * stride);
    Strip8Rows<NOUT> out(dst, s);
    static_for<NOUT>([&](auto ic) {
        constexpr int y = decltype(ic)::value;
        const u32x3 d = q[y % AHEAD];
        if constexpr (y + AHEAD < NOUT) q[y % AHEAD] = *(GLOBAL_AS const u32x3 *)(rp + (long)(y + AHEAD) * stride);
        out.template emit<y>((u32x2){ __builtin_amdgcn_alignbyte(d.y, d.x, sh), __builtin_amdgcn_alignbyte(d.z, d.y, sh) });
    });
}
template <int NOUT, class SRC>
__device__ __forceinline__ void strip8(const SRC &src, const Taps &tx, const Taps &ty, g_u8p dst, int s)
{
    constexpr int NIN = NOUT + 5, AHEAD = IP_AHEAD < NIN ? IP_AHEAD : NIN;
    Strip8Rows<NOUT> out(dst, s);
    typename SRC::Raw q[AHEAD];
#pragma unroll
    for (int i = 0; i < AHEAD; i++) q[i] = src.issue(i);
    // source row I: the horizontal pass, and from the sixth row on an output row
    static_for<NIN>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        const u32x4 d = src.finish(q[I % AHEAD], I);
        if constexpr (I + AHEAD < NIN) q[I % AHEAD] = src.issue(I + AHEAD);
        out.template feed<I>(h_pass8(d, src.sh, tx), ty);
    });
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the tile reader's macroblocks with ONE motion vector fetch their source rows TOGETHER, through LDS.
//
// What bounded the tile reader was the number of cache lines its loads touch (TA_BUSY 85 % of the kernel, 110 vector-cache accesses
// per macroblock against the raster reader's 54): a lane of a strip asks for 16 bytes of ONE pixel row, and the next row's 16 bytes
// -- the same 64-byte sector of the same tile -- with the next instruction.  Here four neighbouring lanes ask for the four rows of a
// sector with one instruction (luma: a macroblock's 21 x 21 source window is 6 sector rows x 3 tile pieces = 18 accesses where the
// strips made 48; chroma: a sector holds four rows of BOTH planes, 12 accesses for 42), and because the lane that loads a row is
// not the lane that filters it the rows go through LDS -- NORMALISED on the way: the loading lane picks the window's dwords out of
// the pieces (the two-stage shifter the strips ran per row and lane), shifts them to the window's first pixel (v_alignbyte: the
// strips' horizontal pass then needs none), replaces what lies beyond the frame's left or right edge by the edge pixel, and writes
// 24 (chroma 16) bytes a row.  A strip's row is then one aligned LDS read at 8 s.  Rows above and below the plane are clamped row
// indices of the loading lane, as before.  Three sector rows of 32 macroblocks x 4 rows x 32 bytes are resident per wave (12 KB:
// three waves per SIMD still fit); the lanes of a wave are up to three rows apart in their sector rows (the windows' first rows
// differ mod 4), so a slot is rewritten when the lanes that began earliest are done with it, and the loads for it are issued four
// rows before that.
//
// The source window of ONE macroblock plane as the lane that loads it sees it: NB bytes a row from pixel X0 on, rows from y0 on.
template <bool CHROMA>
struct WinGeo {
    static constexpr int LGR = CHROMA ? 3 : 4, RMASK = (1 << LGR) - 1, BOT0 = CHROMA ? 4 : 12;      // as TileSrc
    static constexpr int PW = CHROMA ? 8 : 16, LG = CHROMA ? 3 : 4;                                    // bytes of a tile's row piece
    static constexpr int NB = CHROMA ? 16 : 24, NDW = NB / 4;                                          // the normalised row
    static constexpr int NPX = CHROMA ? 13 : 21;                                                       // ... and how many of its pixels the strips read
    static constexpr int NE = CHROMA ? 6 : 9;                                                          // dwords loaded a row
    g_cu8p rowp;            // two tiles before the first tile of the tile row the window begins in (+ 256: chroma)
    int base4, tlo, thi;    // the window's first row floored to a multiple of four, and the plane's first / last row, all relative to that tile row
    u32 rowbytes, colW, colB;
    u32 a_sh;               // dword shift for window rows | for bottom rows << 2 | byte shift << 4
    bool edge;
    u32 edge_info;          // bytes to replace (left: the first n, right: from the n-th on) | where the edge pixel lies << 8 | left << 16
    __device__ __forceinline__ WinGeo(g_cu8p tiles, int cols, int W, int H, int X0, int y0)
    {
        rowbytes = (u32)(cols + 1) * VP8_TILE_BYTES;
        const int yc = max(0, min(y0, H - 1));
        const int ybase = yc & ~RMASK;
        base4 = (y0 - ybase) & ~3; tlo = -ybase; thi = H - 1 - ybase;
        rowp = tiles + (long)(yc >> LGR) * rowbytes - 2 * VP8_TILE_BYTES;
        // what is loaded: from xl on, at least one of the first NPX pixels inside the plane (the edge pixel has to be among the bytes
        // the shifter delivers whole: with the vector clamps of reconinter.c:348-382 a window begins up to 21 pixels left of the frame)
        const int xl = max(min(X0, W - 1), 1 - NPX);
        const int xs = xl + 4;
        colW = (u32)(((xs >> LG) + 2) * VP8_TILE_BYTES);
        colB = (u32)(((xl >> LG) + 2) * VP8_TILE_BYTES + (CHROMA ? 64 : 0));
        a_sh = (u32)((xs & (PW - 1)) >> 2) | ((u32)((xl & (PW - 1)) >> 2) << 2) | (((u32)xl & 3u) << 4);
        // byte t of the normalised row is pixel X0 + t (where xl differs from X0 every byte is beyond one edge)
        edge = X0 < 0 || X0 + NB > W;
        const bool left = X0 < 0;
        const int n = left ? min(NB, -X0) : max(0, W - X0);        // left: bytes [0, n) replaced; right: [n, NB)
        const int ed = left ? -xl : W - 1 - xl;                     // where pixel 0 / W - 1 lies
        edge_info = (u32)n | ((u32)ed << 8) | ((u32)left << 16);
    }
    // where row `v` (relative to the tile row, any value) lies -- the row piece of the window's first tile --, and whether it is one of
    // the macroblock-aligned rows.  Rows above / below the plane are its first / last (vp8_extend_mb_row, vp8_yv12_extend_frame_borders).
    // Chroma: the PAIR of rows (v, v + 1), v even -- 16 adjacent bytes; of a pair beyond the plane both rows are the plane's first
    // (the pair's first row) resp. last (its second): first_twice / second_twice
    __device__ __forceinline__ g_cu8p row_ptr(int v, bool &bot, bool &first_twice, bool &second_twice) const
    {
        int t = max(tlo, min(v, thi));
        first_twice = CHROMA && v + 1 < tlo; second_twice = CHROMA && v > thi;
        if constexpr (CHROMA) t &= ~1;            // (tlo is even, thi odd)
        const u32 yy = (u32)t & RMASK;
        bot = yy >= BOT0;
        return rowp + (__umul24((u32)t >> LGR, rowbytes) + (bot ? colB : colW) + (CHROMA ? (yy & 3u) << 3 : yy << 4));
    }
    // E[0 .. NE-1]: the row's loaded dwords (luma: two pieces and the first dword of the third; chroma: three pieces) -> N: the
    // normalised row.  any_edge: a lane of the wave has an edge in its window
    __device__ __forceinline__ void normalise(const u32 (&E)[NE], bool bot, bool any_edge, u32 (&N)[NDW]) const
    {
        const u32 a = bot ? (a_sh >> 2) & 3u : a_sh & 3u, sh = a_sh >> 4;
        u32 D[NDW + 1];
        if constexpr (!CHROMA) {
            // dwords a .. a + 5 of the nine (a + 6 only feeds bytes past the 21st pixel): a shifter of two stages under masks
            const u32 m2 = (u32)__builtin_amdgcn_sbfe((int)a, 1, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)a, 0, 1);
            u32 F[NDW + 1];
#pragma unroll
            for (int k = 0; k < NDW + 1; k++) F[k] = (E[k + 2] & m2) | (E[k] & ~m2);
#pragma unroll
            for (int k = 0; k < NDW; k++) D[k] = (F[k + 1] & m1) | (F[k] & ~m1);
            D[NDW] = 0;
        } else {
            const u32 m1 = (u32)-(int)a;                // (a: 0 or 1)
#pragma unroll
            for (int k = 0; k < NDW + 1; k++) D[k] = (E[k + 1] & m1) | (E[k] & ~m1);
        }
#pragma unroll
        for (int k = 0; k < NDW; k++) N[k] = __builtin_amdgcn_alignbyte(D[k + 1], D[k], sh);
        if (any_edge) {
            // (the masks are made here, by the few waves that have a window at an edge, rather than kept in registers by all)
            const int n = (int)(edge_info & 0xff), ed = (int)((edge_info >> 8) & 0xff);
            const bool left = (edge_info >> 16) & 1;
            const u32 spl = (u32)(ed & 7) * 0x01010101u;
            u32 ev = 0;
#pragma unroll
            for (int k = 0; k < NDW / 2; k++) ev |= perm(N[2 * k + 1], N[2 * k], (ed >> 3) == k ? spl : 0x0c0c0c0cu);
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                const int nk = max(0, min(4, n - 4 * k));
                const u32 low = nk >= 4 ? 0xffffffffu : (1u << (8 * nk)) - 1u;
                const u32 em = edge ? (left ? low : ~low) : 0u;
                N[k] = (N[k] & ~em) | (ev & em);
            }
        }
    }
};

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

} // namespace

// grid: any number of blocks of four waves (a wave takes units unit, unit + waves, ...); upf = units (64 macroblocks) per frame.
// TILES: the reference frames are read as tiles (DevJob::ref_tile), else as border-extended raster frames (DevJob::ref).
template <bool TILES>
__device__ __forceinline__ void inter_pred(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int upf)
{
    // (the tile reader's workgroup has to stay under a third of a CU's LDS -- 54,613 bytes -- for three waves per SIMD)
    __shared__ u32 s_rca[4][64], s_mva[4][64], s_cmva[4][64];
    __shared__ unsigned char s_lista[4][5][64];
    __shared__ __attribute__((aligned(16))) u32 s_ringa[TILES ? 4 : 1][TILES ? 3072 : 4];      // (the tile reader's rows on their way from loading to filtering lanes)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *const ring = s_ringa[TILES ? wv : 0];
    u32 *const s_rc = s_rca[wv], *const s_mv = s_mva[wv], *const s_cmv = s_cmva[wv];
    // the unit's macroblocks by what their prediction takes: one vector with a fraction / without one (luma), the same for the chroma
    // vector derived from it, SPLITMV
    unsigned char *const s_plain = s_lista[wv][0], *const s_whole = s_lista[wv][1], *const s_cplain = s_lista[wv][2],
                  *const s_cwhole = s_lista[wv][3], *const s_split = s_lista[wv][4];
    const int cols = g.mb_cols, rows = g.mb_rows, nmb = cols * rows;
    const long nunits = (long)njobs * upf;
    for (long unit = (long)blockIdx.x * 4 + wv; unit < nunits; unit += (long)gridDim.x * 4) {
        const int j = (int)(unit / upf), u = (int)(unit - (long)j * upf);
        const DevJob &job = jobs[j];
        if (job.hdr.frame_type == 0) continue;
        const bool bil = job.hdr.version != 0, fullpix = job.hdr.version == 3;
        g_cu32p mbs = (g_cu32p)job.mbx;
        g_cmvp mvs = (g_cmvp)job.mvs;
        const g_u8p tiles = (g_u8p)job.tile;
        const int mb_l = u * 64 + lane;
        const u32 w0_l = mb_l < nmb ? mbs[(long)mb_l * VP8IR_MBX_WORDS] : 0u;
        const bool inter_l = ((w0_l >> 16) & 0xff) != VP8IR_INTRA_FRAME;
        const bool split_l = inter_l && (w0_l & 0xff) == VP8IR_SPLITMV;
        const bool one_l = inter_l && !split_l;
        // Per macroblock of the unit, ONCE (the passes below look a macroblock up several times, the tile reader as filtering and as
        // loading lane): its row and column (a division), and -- one motion vector -- that vector as the luma predictor uses it
        // (clamp_mv_to_umv_border, reconinter.c:348-368, where the macroblock is flagged) and as the chroma predictor does
        // (reconinter.c:419-424: from the CLAMPED luma vector, rounded away from zero, halved; whole pixels in version 3)
        const int r_l = mb_l / cols, c_l = mb_l - r_l * cols;
        u32 mv_l = 0, cmv_l = 0;
        bool whole_l = false, cwhole_l = false;
        if (one_l) {
            const u32 mvw = mvs[(long)mb_l * 16];
            int mrow = sext16(mvw), mcol = hi16(mvw);
            if ((w0_l >> 24) & VP8IR_MB_CLAMP)
                clamp_luma_mv(mrow, mcol, -((c_l * 16) << 3), ((cols - 1 - c_l) * 16) << 3, -((r_l * 16) << 3), ((rows - 1 - r_l) * 16) << 3);
            mv_l = ((u32)mrow & 0xffffu) | ((u32)mcol << 16);
#ifndef IP_NOCOPY         // (diagnostic builds: every macroblock through the filters, as through round 5)
            whole_l = ((mrow | mcol) & 7) == 0;                 // vp8_build_inter16x16_predictors_mb: `mv.as_int & 0x00070007`
#endif
            mrow = (short)(mrow + (1 | (mrow >> 31)));
            mcol = (short)(mcol + (1 | (mcol >> 31)));
            mrow /= 2; mcol /= 2;
            if (fullpix) { mrow &= ~7; mcol &= ~7; }
            cmv_l = ((u32)mrow & 0xffffu) | ((u32)mcol << 16);
#ifndef IP_NOCOPY
            cwhole_l = ((mrow | mcol) & 7) == 0;
#endif
        }
        const bool is_l[5] = { one_l && !whole_l, one_l && whole_l, one_l && !cwhole_l, one_l && cwhole_l, split_l };
        unsigned long long bal[5];
        int cnt[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { bal[k] = __builtin_amdgcn_ballot_w64(is_l[k]); cnt[k] = __builtin_popcountll(bal[k]); }
        wave_lds_sync();                                    // the previous unit's readers are done
        // row << 20 | column << 8 | vectors to be clamped << 2 | reference frame (a frame has at most 1024 x 1024 macroblocks)
        s_rc[lane] = ((u32)r_l << 20) | ((u32)c_l << 8) | (((w0_l >> 24) & VP8IR_MB_CLAMP) ? 4u : 0u) | ((w0_l >> 16) & 3u);
        s_mv[lane] = mv_l; s_cmv[lane] = cmv_l;
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (is_l[k]) s_lista[wv][k][__builtin_amdgcn_mbcnt_hi((u32)(bal[k] >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal[k], 0u))] = (unsigned char)lane;
        wave_lds_sync();
        // What is left of the whole-pixel lists behind their last full pass of 32 fills the free places of the filtered lists' last pass
        // -- taps {0, 0, 128, 0, 0, 0} give the same pixels, as the reference's C filters would (filter.c:95-128; through round 5 every
        // macroblock went that way) -- and only what does not fit there makes a (short, cheap) copy pass of its own.
        const int mvW = min(cnt[1] & 31, -cnt[0] & 31), mvC = min(cnt[3] & 31, -cnt[2] & 31);
        const int nW = cnt[1] - mvW, nCW = cnt[3] - mvC, nP = cnt[0] + mvW, nCP = cnt[2] + mvC, nS = cnt[4];
        if (lane < mvW) s_plain[cnt[0] + lane] = s_whole[nW + lane];
        if (lane < mvC) s_cplain[cnt[2] + lane] = s_cwhole[nCW + lane];
        wave_lds_sync();

        // what a lane needs to know about macroblock `li` of the unit (mrow / mcol, crow / ccol: its one vector, luma / chroma form)
        struct Mb { int r, c; bool clamp; int mrow, mcol, crow, ccol; long idx; int e_left, e_right, e_top, e_bottom; g_cu8p ref; g_u8p tile; };
        auto mb_of = [&](int li) {
            Mb m;
            m.idx = (long)u * 64 + li;
            const u32 rc = s_rc[li], mv = s_mv[li], cmv = s_cmv[li];
            m.r = (int)(rc >> 20); m.c = (int)((rc >> 8) & 0xfffu);
            m.clamp = (rc & 4u) != 0;
            m.mrow = sext16(mv); m.mcol = hi16(mv); m.crow = sext16(cmv); m.ccol = hi16(cmv);
            m.e_left = -((m.c * 16) << 3); m.e_right = ((cols - 1 - m.c) * 16) << 3;
            m.e_top = -((m.r * 16) << 3); m.e_bottom = ((rows - 1 - m.r) * 16) << 3;
            const int rf = (int)(rc & 3u);
            if constexpr (TILES) m.ref = (g_cu8p)(rf == 1 ? job.ref_tile[0] : rf == 2 ? job.ref_tile[1] : job.ref_tile[2]);
            else m.ref = (g_cu8p)(rf == 1 ? job.ref[1] : rf == 2 ? job.ref[2] : job.ref[3]);
            m.tile = tiles + ((long)m.r * (cols + 1) + m.c) * VP8_TILE_BYTES;
            return m;
        };
        // a strip's source: NDW dwords a row (4: thirteen pixels, 3: nine) from (x, y) of the plane (0 Y, 1 U, 2 V) on
        auto luma_src = [&](auto ndw, const Mb &m, int x, int y) {
            constexpr int NDW = decltype(ndw)::value;
            if constexpr (TILES) return TileSrc<false, NDW>(m.ref, cols, g.aligned_w, g.aligned_h, x, y);
            else {
                // memory safety only (a conforming stream never triggers these): every tap inside the plane and its border
                const int sx = max(-32, min(x, g.aligned_w + 32 - 4 * NDW)), sy = max(-32, min(y, g.aligned_h + 32 - (NDW == 4 ? 21 : 9)));
                return RasterSrc<NDW>(m.ref + g.y_off + (long)sy * g.y_stride + sx, g.y_stride);
            }
        };
        auto chroma_src = [&](auto ndw, const Mb &m, int pl, int x, int y) {
            constexpr int NDW = decltype(ndw)::value;
            if constexpr (TILES) return TileSrc<true, NDW>(m.ref + 256 + 32 * pl, cols, g.aligned_w / 2, g.aligned_h / 2, x, y);
            else {
                const int sx = max(-16, min(x, g.aligned_w / 2 + 16 - 4 * NDW)), sy = max(-16, min(y, g.aligned_h / 2 + 16 - (NDW == 4 ? 13 : 9)));
                return RasterSrc<NDW>(m.ref + (pl ? g.v_off : g.u_off) + (long)sy * g.uv_stride + sx, g.uv_stride);
            }
        };
        typedef std::integral_constant<int, 4> W8;      // strips eight pixels wide
        typedef std::integral_constant<int, 3> W4;      // 4x4 blocks

        // ---- one motion vector: vp8_build_inter16x16_predictors_mb
        if constexpr (TILES) {
            // Luma: 32 macroblocks a pass; as a FILTERING lane: macroblock lane >> 1, strip lane & 1; as a LOADING lane, twice a sector
            // row: macroblocks lane >> 2 and 16 + (lane >> 2), row lane & 3 of the sector.  Positions behind the list's end work on its
            // last macroblock once more (the same bytes to the same places): no lane is ever masked.
            // LDS (dwords): ring[slot 3][macroblock 32][row 4][8]
            // COPY: the list of the vectors without a fraction -- the same windows (their halo rows and columns loaded for nothing: the
            // loader is one code path), but a filtering lane takes rows 2..17 from byte 2 on as they are.
            auto luma_pass = [&](const unsigned char *list, const int n, const bool COPY /* wave-uniform: ONE body -- the kernel has to stay inside the instruction cache */) {
                for (int i0 = 0; i0 < n; i0 += 32) {
                    wave_lds_sync();                                // the pass before has read its last rows
                    const int s = lane & 1;
                    const Mb m = mb_of(list[min(i0 + (lane >> 1), n - 1)]);
                    const Taps tx = load_taps(bil, COPY ? 0 : m.mcol & 7), ty = load_taps(bil, COPY ? 0 : m.mrow & 7);
                    const u32 m4 = (u32)(m.r * 16 + (m.mrow >> 3) - 2) & 3u;                    // the window's first row within its sector
                    const bool c1 = m4 >= 1, c2 = m4 >= 2, c3 = m4 >= 3;
                    const u32 kq[4] = { m4 * 8, ((m4 + 1) & 3) * 8, ((m4 + 2) & 3) * 8, ((m4 + 3) & 3) * 8 };
                    const u32 *const mine = ring + (lane >> 1) * 32 + 2 * s;
                    typedef WinGeo<false> Geo;
                    auto geo_of = [&](int hf) {
                        const Mb ml = mb_of(list[min(i0 + 16 * hf + (lane >> 2), n - 1)]);
                        return Geo(ml.ref, cols, g.aligned_w, g.aligned_h, ml.c * 16 + (ml.mcol >> 3) - 2, ml.r * 16 + (ml.mrow >> 3) - 2);
                    };
                    const Geo g0 = geo_of(0), g1 = geo_of(1);
                    const bool any_edge = __builtin_amdgcn_ballot_w64(g0.edge || g1.edge) != 0;
                    struct Raw { u32x4 a, b; u32 c; bool bot; };
                    auto issue = [&](const Geo &ge, int sr) {
                        bool ft, st;
                        Raw r;
                        g_cu8p p = ge.row_ptr(ge.base4 + 4 * sr + (lane & 3), r.bot, ft, st);
                        r.a = *(g_cu32x4p)p; r.b = *(g_cu32x4p)(p + VP8_TILE_BYTES); r.c = *(GLOBAL_AS const u32 *)(p + 2 * VP8_TILE_BYTES);
                        return r;
                    };
                    auto land = [&](const Geo &ge, const Raw &r, int sr, int hf) {
                        const u32 E[9] = { r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w, r.c };
                        u32 N[6];
                        ge.normalise(E, r.bot, any_edge, N);
                        u32 *q = ring + (sr % 3) * 1024 + (16 * hf + (lane >> 2)) * 32 + (lane & 3) * 8;
                        *(u32x4 *)q = (u32x4){ N[0], N[1], N[2], N[3] };
                        *(u32x2 *)(q + 4) = (u32x2){ N[4], N[5] };
                    };
                    Raw n0, n1;
                    {   // sector rows 0..2 into the three slots (two sector rows' loads in flight at a time)
                        Raw a0 = issue(g0, 0), a1 = issue(g1, 0), b0 = issue(g0, 1), b1 = issue(g1, 1);
                        land(g0, a0, 0, 0); land(g1, a1, 0, 1);
                        a0 = issue(g0, 2); a1 = issue(g1, 2);
                        land(g0, b0, 1, 0); land(g1, b1, 1, 1);
                        n0 = issue(g0, 3); n1 = issue(g1, 3);
                        land(g0, a0, 2, 0); land(g1, a1, 2, 1);
                    }
                    wave_lds_sync();
                    Strip8Rows<16> out(m.tile, s);
                    static_for<21>([&](auto ic) {
                        constexpr int I = decltype(ic)::value;
                        if constexpr (I == 4 || I == 8 || I == 12) {
                            // every lane is done with sector row I / 4 - 1: its slot takes sector row I / 4 + 2, whose loads were issued four
                            // rows ago; the next one's go out
                            wave_lds_sync();
                            land(g0, n0, I / 4 + 2, 0); land(g1, n1, I / 4 + 2, 1);
                            if constexpr (I < 12) { n0 = issue(g0, I / 4 + 3); n1 = issue(g1, I / 4 + 3); }
                            wave_lds_sync();
                        }
                        // row I of the window: sector row (m4 + I) >> 2, row (m4 + I) & 3 of it
                        constexpr int G = I >> 2, R = I & 3;
                        const bool carry = R == 0 ? false : R == 1 ? c3 : R == 2 ? c2 : c1;
                        const u32 off = carry ? ((G + 1) % 3) * 1024u : (G % 3) * 1024u;
                        const u32 *q = mine + off + kq[R];
                        const u32x2 lo = *(const u32x2 *)q, hi = *(const u32x2 *)(q + 2);
                        if (!COPY) out.template feed<I>(h_pass8n((u32x4){ lo.x, lo.y, hi.x, hi.y }, tx), ty);
                        else if constexpr (I >= 2 && I < 18)
                            out.template emit<I - 2>((u32x2){ __builtin_amdgcn_alignbyte(lo.y, lo.x, 2), __builtin_amdgcn_alignbyte(hi.x, lo.y, 2) });
                    });
                }
            };
            // Chroma: 32 macroblocks a pass; filtering lane: macroblock lane >> 1, plane lane & 1; loading lane: macroblocks lane >> 2 and
            // 16 + (lane >> 2), plane (lane >> 1) & 1, row pair lane & 1 of the sector (a sector: four rows of U, four rows of V).
            // LDS (dwords): ring[slot 3][macroblock 32][plane 2][row 4][4]
            auto chroma_pass = [&](const unsigned char *list, const int n, const bool COPY) {
                for (int i0 = 0; i0 < n; i0 += 32) {
                    wave_lds_sync();
                    const int pl = lane & 1;
                    const Mb m = mb_of(list[min(i0 + (lane >> 1), n - 1)]);
                    const Taps tx = load_taps(bil, COPY ? 0 : m.ccol & 7), ty = load_taps(bil, COPY ? 0 : m.crow & 7);
                    const u32 m4 = (u32)(m.r * 8 + (m.crow >> 3) - 2) & 3u;
                    const bool c1 = m4 >= 1, c2 = m4 >= 2, c3 = m4 >= 3;
                    const u32 kq[4] = { m4 * 4, ((m4 + 1) & 3) * 4, ((m4 + 2) & 3) * 4, ((m4 + 3) & 3) * 4 };
                    const u32 *const mine = ring + (lane >> 1) * 32 + pl * 16;
                    typedef WinGeo<true> Geo;
                    const int lpl = (lane >> 1) & 1, lrp = lane & 1;
                    auto geo_of = [&](int hf) {
                        const Mb ml = mb_of(list[min(i0 + 16 * hf + (lane >> 2), n - 1)]);
                        return Geo(ml.ref + 256 + 32 * lpl, cols, g.aligned_w / 2, g.aligned_h / 2, ml.c * 8 + (ml.ccol >> 3) - 2, ml.r * 8 + (ml.crow >> 3) - 2);
                    };
                    const Geo g0 = geo_of(0), g1 = geo_of(1);
                    const bool any_edge = __builtin_amdgcn_ballot_w64(g0.edge || g1.edge) != 0;
                    // (a lane of the wave loads rows above or below the plane: only then are a pair's rows anything but its two halves)
                    const bool any_yedge = __builtin_amdgcn_ballot_w64(g0.base4 < g0.tlo || g0.base4 + 15 > g0.thi || g1.base4 < g1.tlo || g1.base4 + 15 > g1.thi) != 0;
                    struct Raw { u32x4 a, b, c; bool bot, ft, st; };
                    auto issue = [&](const Geo &ge, int sr) {
                        Raw r;
                        g_cu8p p = ge.row_ptr(ge.base4 + 4 * sr + 2 * lrp, r.bot, r.ft, r.st);
                        r.a = *(g_cu32x4p)p; r.b = *(g_cu32x4p)(p + VP8_TILE_BYTES); r.c = *(g_cu32x4p)(p + 2 * VP8_TILE_BYTES);
                        return r;
                    };
                    auto land = [&](const Geo &ge, const Raw &r, int sr, int hf) {
                        // the pair's two rows: (x, y) of every piece the first, (z, w) the second -- both the same one beyond the plane
                        u32 EA[6] = { r.a.x, r.a.y, r.b.x, r.b.y, r.c.x, r.c.y }, EB[6] = { r.a.z, r.a.w, r.b.z, r.b.w, r.c.z, r.c.w };
                        if (any_yedge) {
#pragma unroll
                            for (int k = 0; k < 6; k++) { const u32 lo = EA[k], hi = EB[k]; EA[k] = r.st ? hi : lo; EB[k] = r.ft ? lo : hi; }
                        }
                        u32 NA[4], NB_[4];
                        ge.normalise(EA, r.bot, any_edge, NA);
                        ge.normalise(EB, r.bot, any_edge, NB_);
                        u32 *q = ring + (sr % 3) * 1024 + (16 * hf + (lane >> 2)) * 32 + lpl * 16 + lrp * 8;
                        *(u32x4 *)q = (u32x4){ NA[0], NA[1], NA[2], NA[3] };
                        *(u32x4 *)(q + 4) = (u32x4){ NB_[0], NB_[1], NB_[2], NB_[3] };
                    };
                    Raw n0, n1;
                    {
                        Raw a0 = issue(g0, 0), a1 = issue(g1, 0), b0 = issue(g0, 1), b1 = issue(g1, 1);
                        land(g0, a0, 0, 0); land(g1, a1, 0, 1);
                        a0 = issue(g0, 2); a1 = issue(g1, 2);
                        land(g0, b0, 1, 0); land(g1, b1, 1, 1);
                        n0 = issue(g0, 3); n1 = issue(g1, 3);
                        land(g0, a0, 2, 0); land(g1, a1, 2, 1);
                    }
                    wave_lds_sync();
                    Strip8Rows<8> out(m.tile + 256 + 32 * pl, 0);
                    static_for<13>([&](auto ic) {
                        constexpr int I = decltype(ic)::value;
                        if constexpr (I == 4) {
                            wave_lds_sync();
                            land(g0, n0, 3, 0); land(g1, n1, 3, 1);
                            wave_lds_sync();
                        }
                        constexpr int G = I >> 2, R = I & 3;
                        const bool carry = R == 0 ? false : R == 1 ? c3 : R == 2 ? c2 : c1;
                        const u32 off = carry ? ((G + 1) % 3) * 1024u : (G % 3) * 1024u;
                        const u32x4 d = *(const u32x4 *)(mine + off + kq[R]);
                        if (!COPY) out.template feed<I>(h_pass8n(d, tx), ty);
                        else if constexpr (I >= 2 && I < 10)
                            out.template emit<I - 2>((u32x2){ __builtin_amdgcn_alignbyte(d.y, d.x, 2), __builtin_amdgcn_alignbyte(d.z, d.y, 2) });
                    });
                }
            };
            // (one loop over both lists: one copy of the pass's code)
            for (int k = 0; k < 2; k++) luma_pass(k ? s_whole : s_plain, k ? nW : nP, k != 0);
            for (int k = 0; k < 2; k++) chroma_pass(k ? s_cwhole : s_cplain, k ? nCW : nCP, k != 0);
        } else {
            // Luma: 32 macroblocks x 2 strips
            for (int i0 = 0; i0 < nP; i0 += 32) {
                const int mi = i0 + (lane >> 1), s = lane & 1;
                if (mi < nP) {
                    const Mb m = mb_of(s_plain[mi]);
                    const Taps tx = load_taps(bil, m.mcol & 7), ty = load_taps(bil, m.mrow & 7);
                    strip8<16>(luma_src(W8(), m, m.c * 16 + (m.mcol >> 3) - 2 + 8 * s, m.r * 16 + (m.mrow >> 3) - 2), tx, ty, m.tile, s);
                }
            }
            // ... whose vector has no fraction: copied (memory safety as in luma_src: twelve bytes a row inside the plane and its border)
            for (int i0 = 0; i0 < nW; i0 += 32) {
                const int mi = i0 + (lane >> 1), s = lane & 1;
                if (mi < nW) {
                    const Mb m = mb_of(s_whole[mi]);
                    const int x = max(-32, min(m.c * 16 + (m.mcol >> 3) + 8 * s, g.aligned_w + 32 - 12)), y = max(-32, min(m.r * 16 + (m.mrow >> 3), g.aligned_h + 32 - 16));
                    copy_strip8<16>(m.ref + g.y_off + (long)y * g.y_stride + x, g.y_stride, m.tile, s);
                }
            }
            // chroma: 32 macroblocks x 2 planes
            for (int i0 = 0; i0 < nCP; i0 += 32) {
                const int mi = i0 + (lane >> 1), pl = lane & 1;
                if (mi < nCP) {
                    const Mb m = mb_of(s_cplain[mi]);
                    const Taps tx = load_taps(bil, m.ccol & 7), ty = load_taps(bil, m.crow & 7);
                    strip8<8>(chroma_src(W8(), m, pl, m.c * 8 + (m.ccol >> 3) - 2, m.r * 8 + (m.crow >> 3) - 2), tx, ty, m.tile + 256 + 32 * pl, 0);
                }
            }
            for (int i0 = 0; i0 < nCW; i0 += 32) {
                const int mi = i0 + (lane >> 1), pl = lane & 1;
                if (mi < nCW) {
                    const Mb m = mb_of(s_cwhole[mi]);
                    const int x = max(-16, min(m.c * 8 + (m.ccol >> 3), g.aligned_w / 2 + 16 - 12)), y = max(-16, min(m.r * 8 + (m.crow >> 3), g.aligned_h / 2 + 16 - 8));
                    copy_strip8<8>(m.ref + (pl ? g.v_off : g.u_off) + (long)y * g.uv_stride + x, g.uv_stride, m.tile + 256 + 32 * pl, 0);
                }
            }
        }
        // ---- SPLITMV: build_inter4x4_predictors_mb, a 4x4 block per lane (partitions of 8x8 / 16x8 / 8x16 carry their MV in every
        // block they cover: the filters are the same per pixel).  Luma: 4 macroblocks x 16 blocks
#ifndef IP_NOSPLIT        // (diagnostic builds: what the kernels need without the SPLITMV paths)
        for (int i0 = 0; i0 < nS; i0 += 4) {
            const int mi = i0 + (lane >> 4), b = lane & 15;
            if (mi < nS) {
                const Mb m = mb_of(s_split[mi]);
                const u32 mvw = mvs[m.idx * 16 + b];
                int mrow = sext16(mvw), mcol = hi16(mvw);
                if (m.clamp) clamp_luma_mv(mrow, mcol, m.e_left, m.e_right, m.e_top, m.e_bottom);
                const Taps tx = load_taps(bil, mcol & 7), ty = load_taps(bil, mrow & 7);
                strip<4>(luma_src(W4(), m, m.c * 16 + 4 * (b & 3) + (mcol >> 3) - 2, m.r * 16 + 4 * (b >> 2) + (mrow >> 3) - 2), tx, ty,
                         m.tile + 64 * (b >> 2) + 4 * (b & 3), 16, 0);
            }
        }
        // chroma: 8 macroblocks x 2 planes x 4 blocks; build_4x4uvmvs (reconinter.c:520-558): the UNclamped MVs of the four luma
        // blocks above a chroma block, averaged
        for (int i0 = 0; i0 < nS; i0 += 8) {
            const int mi = i0 + (lane >> 3), pl = (lane >> 2) & 1, blk = lane & 3;
            if (mi < nS) {
                const Mb m = mb_of(s_split[mi]);
                const int kq = (blk >> 1) * 8 + (blk & 1) * 2;
                g_cmvp mv = mvs + m.idx * 16;
                const u32 m0 = mv[kq], m1 = mv[kq + 1], m4 = mv[kq + 4], m5 = mv[kq + 5];
                int mrow = sext16(m0) + sext16(m1) + sext16(m4) + sext16(m5);
                int mcol = hi16(m0) + hi16(m1) + hi16(m4) + hi16(m5);
                mrow += 4 + ((mrow >> 31) << 3);
                mcol += 4 + ((mcol >> 31) << 3);
                mrow /= 8; mcol /= 8;
                if (fullpix) { mrow &= ~7; mcol &= ~7; }
                if (m.clamp) clamp_chroma_mv(mrow, mcol, m.e_left, m.e_right, m.e_top, m.e_bottom);
                const Taps tx = load_taps(bil, mcol & 7), ty = load_taps(bil, mrow & 7);
                strip<4>(chroma_src(W4(), m, pl, m.c * 8 + 4 * (blk & 1) + (mcol >> 3) - 2, m.r * 8 + 4 * (blk >> 1) + (mrow >> 3) - 2), tx, ty,
                         m.tile + 256 + 32 * pl + 64 * (blk >> 1) + 4 * (blk & 1), 8, 0);
            }
        }
#endif
    }
}

extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IP_WAVES, 8)))
vp8_inter_pred_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int upf)
{
    inter_pred<false>(jobs, njobs, g, upf);
}
// ... from reference frames that are there as tiles (the frames a large launch left: DevJob::ref_tile)
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IP_WAVES_TILES, 8)))
vp8_inter_pred_tiles_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int upf)
{
    inter_pred<true>(jobs, njobs, g, upf);
}
