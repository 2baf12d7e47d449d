// Entropy decoding on the device, one frame per LANE (gfx950): the macroblocks' modes, motion vectors and coefficient tokens of
// key and inter frames.
//
// What it replaces: the per-macroblock half of the reference's CPU front end -- vp8_kfread_modes (vp8/decoder/decodemv.c:50-173:
// segment id, skip flag, intra modes with the sub-block modes' above / left contexts), read_mb_modes for inter frames
// (decodemv.c:323-569: reference frame, the near / nearest candidates of vp8_find_near_mvs, NEWMV and SPLITMV vectors) and
// vp8_decode_mb_tokens (vp8/decoder/detokenize.c:183-405: the coefficient token tree over the bool decoder of
// vp8/decoder/dboolhuff.h:76-120) as decode_mb_row drives them (vp8/decoder/decodframe.c:293-470: left / above entropy contexts,
// vp8_reset_mb_tokens_context for skipped macroblocks, eobtotal == 0 turning a macroblock into a skipped one, token partitions
// taken round robin by macroblock row).  In this repository the same work is csrc/host/vp8_parser.c's read_modes / decode_row,
// at ~10 ms per 1080p key frame and host core; what the kernels write into a frame's IR slot is the DEVICE FORM of
// include/vp8_ir.h -- records, block stream, vectors: what vp8_parser_decode_mbs_compact writes on the host and what the pixel
// kernels read as it stands (tests/test_gpu_entropy.py).  The frame header stays on the host (csrc/host/vp8_parser.h:
// vp8_parser_export_entropy).
//
// A bool decoder is a serial machine: every decision needs range and window as the decision before left them.  So there is
// nothing to spread over lanes inside a partition, and a frame is one lane's work from its first macroblock to its last (the
// partitions of a frame with several are taken in macroblock-row order, as the reference's single thread takes them: the
// contexts of a row come from the row above, which belongs to another partition; vp8_entropy_parts_kernel gives each partition
// a lane instead, the lanes a macroblock behind each other).  The frames of a batch run side by side.  A lone wave issues a
// dependent instruction every ~7 cycles, so a lane's speed is its instruction count per decision:
// 32-bit window (a 64-bit one is two instructions per shift), the next three bytes of the partition requested when the three
// before are taken (the request has ~25 decisions to land), probabilities in LDS in rows of 12 bytes read as three words when a
// row is entered (per-lane tables 289 words apart: consecutive lanes on different banks), the macroblock descriptor and the
// record and the block being decoded assembled in LDS and written out whole (16-byte stores; a block only if it has more than
// a first coefficient: nothing is written for the zeros of the dense form), and a launch parameter for how many lanes of a
// wave carry frames.  Integer only; no MFMA.
#include "vp8_common.hip.h"
#include "vp8hip.h"

namespace {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef short __attribute__((may_alias)) coef_t;     // the block staged in LDS is zeroed and read back as words
typedef u32 __attribute__((may_alias)) row_t;        // probability rows are written as bytes and read as words

// Probabilities live in LDS in ROWS of 12 bytes -- the 11 node probabilities of one (block type, band, context), the 9 of one
// (above, left) pair of sub-block modes, the up to 11 of one extra-bits category -- so that what a token's decisions need comes
// with one three-word read when the row is entered, not with a byte read in front of every decision.
#define ENT_ROW 12
#define ENT_PROB_WORDS 289         // per lane: 96 rows = 288 words, + 1 so that consecutive lanes start on different banks
#define ENT_DESC_WORDS 33          // the macroblock's 128-byte record (vp8ir_mbx) + a word of padding
#define ENT_BLK_WORDS  8           // the block being decoded

__constant__ uint8_t k_kf_bmode_probs[900] = {
#include "../host/vp8_kf_bmode_probs.inc"
};

// Pcat1..Pcat6 (vp8/decoder/detokenize.c:52-64; RFC 6386 13.2), a row each
__constant__ uint8_t k_cat_rows[6 * ENT_ROW] = { 159, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,   165, 145, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                 173, 148, 140, 0, 0, 0, 0, 0, 0, 0, 0, 0,   176, 155, 140, 135, 0, 0, 0, 0, 0, 0, 0, 0,
                                                 180, 157, 141, 134, 130, 0, 0, 0, 0, 0, 0, 0,
                                                 254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 0 };

struct Row { u32 w0, w1, w2; };
__device__ __forceinline__ Row row_at(const row_t *p) { Row r; r.w0 = p[0]; r.w1 = p[1]; r.w2 = p[2]; return r; }
// node k of a row, k a constant: a bit-field extract
#define RP(r, k) ((((k) < 4 ? (r).w0 : (k) < 8 ? (r).w1 : (r).w2) >> (8 * ((k) & 3))) & 255u)

struct BD {
    u32 value;      // window: the active byte in bits 31..24
    int bits;       // valid bits below it; negative: refill before the next decision
    u32 range;      // 128..255
    u32 pos, end;   // next byte to request / end of the partition (offsets from the frame's first byte)
    u32 n01, n2;    // bytes pos-3, pos-2 (a little-endian pair) and pos-1 as requested at the refill before (not looked at until
                    // the next refill takes them: the requests have the decisions in between to land)
#ifdef ENT_STATS
    u32 count;      // (diagnostic builds: decisions taken)
#endif
};

// Request the three bytes from pos on.  No branches: they are requested whatever pos is (the launch's data are followed by
// padding, and `limit` keeps a decoder that runs on through a damaged partition inside it).
__device__ __forceinline__ void request3(BD &b, const uint8_t *__restrict__ data, u32 pos, u32 limit)
{
    typedef unsigned short __attribute__((aligned(1), may_alias)) u16u;
    const uint8_t *p = data + (pos < limit ? pos : limit);
    b.n01 = *(const u16u *)p; b.n2 = p[2];
}

// vp8dx_decode_bool (dboolhuff.h:76-120): split = 1 + (((range - 1) * probability) >> 8), the decision is whether the window is
// at or above split; range and window renormalised by the leading zeros of the new range
__device__ __forceinline__ int bd_get(BD &b, const uint8_t *__restrict__ data, u32 limit, u32 prob)
{
    const u32 split = 1u + (__umul24(b.range - 1u, prob) >> 8);
#ifdef ENT_STATS
    b.count++;
#endif
    if (__builtin_expect(b.bits < 0, 0)) {              // 1..8 bits of the top byte are spent: three more bytes under them
        u32 nxt = (b.n01 & 255u) << 16 | (b.n01 & 0xff00u) | b.n2;
        if (b.pos > b.end) {                            // (the partition ends inside these three, or has ended: zeros from there on)
            const u32 past = b.pos - b.end;
            nxt = past >= 3u ? 0u : nxt & ~(0xffffffu >> (8 * (3u - past)));
        }
        b.value |= nxt << (-b.bits);
        b.bits += 24;
        request3(b, data, b.pos, limit);
        b.pos += 3;
    }
    const u32 big = split << 24;
    const bool bit = b.value >= big;
    b.value -= bit ? big : 0u;
    b.range = bit ? b.range - split : split;
    const int shift = __builtin_clz(b.range) - 24;      // (range is never 0)
    b.range <<= shift;
    b.value <<= shift;
    b.bits -= shift;
    return bit ? 1 : 0;
}
#define GET(b, prob) bd_get(b, data, limit, prob)

// vp8dx_bool_error (dboolhuff.h:131-153) as csrc/host/vp8_boolreader.h states it: zeros from behind the end of the partition have
// reached the top byte.  Bytes are taken in order, so those taken from behind the end are what the position says (the three
// before `pos` are requested, not taken).
__device__ __forceinline__ bool bd_error(const BD &b)
{
    const int over = (int)(b.pos - 3u) - (int)b.end;
    return over > 0 && b.bits - 8 * over < 0;
}

// intra sub-block mode tree (vp8_bmode_tree, vp8/common/entropymode.c)
__device__ __forceinline__ int read_bmode(BD &b, const uint8_t *__restrict__ data, u32 limit, const Row pr)
{
    if (!GET(b, RP(pr, 0))) return VP8IR_B_DC_PRED;
    if (!GET(b, RP(pr, 1))) return VP8IR_B_TM_PRED;
    if (!GET(b, RP(pr, 2))) return VP8IR_B_VE_PRED;
    if (!GET(b, RP(pr, 3))) {
        if (!GET(b, RP(pr, 4))) return VP8IR_B_HE_PRED;
        return GET(b, RP(pr, 5)) ? VP8IR_B_VR_PRED : VP8IR_B_RD_PRED;
    }
    if (!GET(b, RP(pr, 6))) return VP8IR_B_LD_PRED;
    if (!GET(b, RP(pr, 7))) return VP8IR_B_VL_PRED;
    return GET(b, RP(pr, 8)) ? VP8IR_B_HU_PRED : VP8IR_B_HD_PRED;
}

// One 4x4 block (the body of vp8_decode_mb_tokens, detokenize.c:262-378).  pr0: the lane's rows of the block type in LDS
// ([band][context]); cat: the extra-bits rows; out: the lane's 16 coefficients in LDS, zeroed, in the IR's column-major order.
// Returns the reference's eob ("c" at BLOCK_FINISHED); nz = the first token was not EOB.
__device__ __forceinline__ int read_block(BD &b, const uint8_t *__restrict__ data, u32 limit, const row_t *pr0, const row_t *cat, int ctx,
                                          int first, coef_t *out, int &nz)
{
    constexpr u64 BANDS = 0x7666666665463210ull;    // vp8_coef_bands (entropy.c), a nibble per position
    constexpr u64 ZIGZAG = 0xFBEDA7369C852140ull;   // vp8_default_zig_zag1d with raster index r * 4 + c mapped to c * 4 + r
    int c = first;
    Row pr = row_at(pr0 + ((int)((BANDS >> (4 * c)) & 15) * 3 + ctx) * 3);
    nz = 0;
    if (!GET(b, RP(pr, 0))) return c;
    nz = 1;
    for (;;) {
        int v, next;
        while (!GET(b, RP(pr, 1))) {                // DCT_0: no EOB test follows a zero
            if (c == 15) return 15;                 // (a stream that codes a zero in the last position: as the reference)
            c++;
            pr = row_at(pr0 + (int)((BANDS >> (4 * c)) & 15) * 9);
        }
        if (!GET(b, RP(pr, 2))) {
            v = 1; next = 1;
        } else {
            next = 2;
            if (!GET(b, RP(pr, 3))) {
                if (!GET(b, RP(pr, 4))) v = 2;
                else v = 3 + GET(b, RP(pr, 5));
            } else {
                int k;                              // DCT_VAL_CATEGORY1..6: base value, number of extra bits, their probabilities
                if (!GET(b, RP(pr, 6))) k = GET(b, RP(pr, 7));
                else if (!GET(b, RP(pr, 8))) k = 2 + GET(b, RP(pr, 9));
                else k = 4 + GET(b, RP(pr, 10));
                const int nbits = k < 5 ? k + 1 : 11;
                const Row cr = row_at(cat + 3 * k);
                int x = 0;
                for (int i = 0; i < nbits; i++) {
                    const u32 w = i < 4 ? cr.w0 : i < 8 ? cr.w1 : cr.w2;
                    x = (x << 1) | GET(b, (w >> (8 * (i & 3))) & 255u);
                }
                v = (k < 5 ? 3 + (2 << k) : 67) + x;                      // 5, 7, 11, 19, 35, 67
            }
        }
        if (GET(b, 128)) v = -v;
        out[(int)((ZIGZAG >> (4 * c)) & 15)] = (coef_t)v;
        if (c == 15) return 15;                     // the reference leaves c at 15 here (detokenize.c:140-146)
        c++;
        pr = row_at(pr0 + ((int)((BANDS >> (4 * c)) & 15) * 3 + next) * 3);
        if (!GET(b, RP(pr, 0))) return c;
    }
}

// What the first partition says about one macroblock (vp8_kfread_modes, decodemv.c:50-173).  above / left: the sub-block modes
// of the row above at this column and of the macroblock to the left (four nibbles each), updated for the neighbours to come.
struct MbModes { int ymode, uvmode, seg, skip; u64 bm; int ref, clamp, part; };    // bm: B_PRED's sixteen modes, a nibble each
// (kept_seg: the segment id of a macroblock whose frame does not code one -- 0, or what the slot's record held: segmap_keep)
struct ModeParams { bool seg_map, has_skip; u32 p_skip, tp0, tp1, tp2; };
__device__ __forceinline__ MbModes read_mb_modes(BD &fb, const uint8_t *__restrict__ data, u32 limit, const ModeParams &P, const row_t *kfb,
                                                 u32 &above, u32 &lbm, int kept_seg)
{
    MbModes m;
    m.ref = VP8IR_INTRA_FRAME; m.clamp = 0; m.part = 0;
    m.seg = kept_seg;
    if (P.seg_map) m.seg = GET(fb, P.tp0) ? 2 + GET(fb, P.tp2) : GET(fb, P.tp1);
    m.skip = P.has_skip ? GET(fb, P.p_skip) : 0;
    if (!GET(fb, 145)) m.ymode = VP8IR_B_PRED;
    else if (!GET(fb, 156)) m.ymode = GET(fb, 163) ? VP8IR_V_PRED : VP8IR_DC_PRED;
    else m.ymode = GET(fb, 128) ? VP8IR_TM_PRED : VP8IR_H_PRED;
    m.bm = 0;
    if (m.ymode == VP8IR_B_PRED) {
        u64 bm = 0;
        for (int i = 0; i < 16; i++) {
            const int A = i < 4 ? (int)((above >> (4 * i)) & 15) : (int)((bm >> (4 * (i - 4))) & 15);
            const int L = (i & 3) == 0 ? (int)((lbm >> (i & 12)) & 15) : (int)((bm >> (4 * (i - 1))) & 15);
            bm |= (u64)read_bmode(fb, data, limit, row_at(kfb + (A * 10 + L) * 3)) << (4 * i);
        }
        above = (u32)(bm >> 48);
        lbm = (u32)((bm >> 12) & 15) | (u32)((bm >> 28) & 15) << 4 | (u32)((bm >> 44) & 15) << 8 | (u32)((bm >> 60) & 15) << 12;
        m.bm = bm;
    } else {
        // the sub-block mode a whole-macroblock mode stands for in its neighbours' contexts (findnearmv.h:131-188)
        const u32 im = m.ymode == VP8IR_V_PRED ? VP8IR_B_VE_PRED : m.ymode == VP8IR_H_PRED ? VP8IR_B_HE_PRED
                     : m.ymode == VP8IR_TM_PRED ? VP8IR_B_TM_PRED : VP8IR_B_DC_PRED;
        above = lbm = im * 0x1111u;
    }
    if (!GET(fb, 142)) m.uvmode = VP8IR_DC_PRED;
    else if (!GET(fb, 114)) m.uvmode = VP8IR_V_PRED;
    else m.uvmode = GET(fb, 183) ? VP8IR_TM_PRED : VP8IR_H_PRED;
    return m;
}

// ---- inter frames (read_mb_modes / read_mbinfo, decodemv.c:323-569; vp8_find_near_mvs, findnearmv.c:25-140, as decodemv.c inlines it)
// A motion vector is a word: row in the low half, column in the high half (vp8ir_mv).  What a macroblock leaves for its
// neighbours: its vector (SPLITMV: the last block's; intra: 0), reference frame and mode, and the four sub-block vectors on the
// side the neighbour touches (a macroblock that is not split: its vector four times).
struct Nb { u32 mv, bmv[4], ref, ymode; };
struct InterParams { u32 p_intra, p_last, p_gf; u32 ymode[4]; u32 uvmode[3]; u32 sign_bias; const uint8_t *mvc; };   // sign_bias: bit per reference frame

__device__ __forceinline__ u32 mv_neg(u32 mv) { return ((0u - (mv & 0xffffu)) & 0xffffu) | ((0u - (mv >> 16)) << 16); }
__device__ __forceinline__ u32 mv_add(u32 a, u32 b) { return ((a + b) & 0xffffu) | ((a >> 16) + (b >> 16)) << 16; }
__device__ __forceinline__ int mv_row(u32 mv) { return (int)(short)(mv & 0xffffu); }
__device__ __forceinline__ int mv_col(u32 mv) { return (int)(short)(mv >> 16); }
__device__ __forceinline__ u32 mv_make(int row, int col) { return ((u32)row & 0xffffu) | (u32)col << 16; }
struct Edges { int left, right, top, bottom; };
__device__ __forceinline__ u32 mv_clamp(u32 mv, const Edges &e)            // vp8_clamp_mv2, findnearmv.h:32-44
{
    int r = mv_row(mv), c = mv_col(mv);
    c = c < e.left ? e.left : c > e.right ? e.right : c;
    r = r < e.top ? e.top : r > e.bottom ? e.bottom : r;
    return mv_make(r, c);
}
__device__ __forceinline__ int mv_outside(u32 mv, const Edges &e)         // vp8_check_mv_bounds
{
    const int r = mv_row(mv), c = mv_col(mv);
    return (c < e.left) | (c > e.right) | (r < e.top) | (r > e.bottom);
}
// read_mvcomponent (decodemv.c:75-110): the long form's bits 0, 1, 2, then 9 .. 4, bit 3 last (and only if it can be zero)
__device__ __forceinline__ int read_mv_component(BD &fb, const uint8_t *__restrict__ data, u32 limit, const uint8_t *pr)
{
    int x = 0;
    if (GET(fb, pr[0])) {
        for (int i = 0; i < 3; i++) x += GET(fb, pr[9 + i]) << i;
        for (int i = 9; i > 3; i--) x += GET(fb, pr[9 + i]) << i;
        if (!(x & 0xfff0) || GET(fb, pr[9 + 3])) x += 8;
    } else {
        if (!GET(fb, pr[2])) {
            if (!GET(fb, pr[3])) x = GET(fb, pr[4]);
            else x = 2 + GET(fb, pr[5]);
        } else {
            if (!GET(fb, pr[6])) x = 4 + GET(fb, pr[7]);
            else x = 6 + GET(fb, pr[8]);
        }
    }
    if (x && GET(fb, pr[1])) x = -x;
    return x;
}
__device__ __forceinline__ u32 read_mv(BD &fb, const uint8_t *__restrict__ data, u32 limit, const uint8_t *mvc)
{
    const int r = read_mv_component(fb, data, limit, mvc) * 2;
    const int c = read_mv_component(fb, data, limit, mvc + 19) * 2;
    return mv_make(r, c);
}

__constant__ uint8_t k_mode_contexts[24] = { 7, 1, 1, 143, 14, 18, 14, 107, 135, 64, 57, 68, 60, 56, 128, 65, 159, 134, 128, 34, 234, 188, 128, 28 };
__constant__ uint8_t k_submv_prob[8][3] = { { 147, 136, 18 }, { 223, 1, 34 }, { 106, 145, 1 }, { 208, 1, 1 },
                                            { 179, 121, 1 }, { 223, 1, 34 }, { 179, 121, 1 }, { 208, 1, 1 } };       // vp8_sub_mv_ref_prob3

// One macroblock of an inter frame.  above / left / aboveleft: the neighbours' records; mvs: the lane's sixteen block vectors in
// LDS (out: what the macroblock's blocks use, vp8ir_mv order).  Returns the modes; `self` = the record for the neighbours to come
// (bmv as the macroblock's own sixteen say: the caller picks the side).
__device__ __forceinline__ MbModes read_mb_modes_inter(BD &fb, const uint8_t *__restrict__ data, u32 limit, const ModeParams &P,
                                                       const InterParams &I, const Nb &above, const Nb &left, const Nb &aboveleft,
                                                       int mb_row, int mb_col, int rows, int cols, u32 *mvs, Nb &self, int kept_seg)
{
    MbModes m;
    m.bm = 0; m.clamp = 0; m.part = 0; m.uvmode = VP8IR_DC_PRED;
    m.seg = kept_seg;
    if (P.seg_map) m.seg = GET(fb, P.tp0) ? 2 + GET(fb, P.tp2) : GET(fb, P.tp1);
    m.skip = P.has_skip ? GET(fb, P.p_skip) : 0;
    u32 mv = 0;
    m.ref = GET(fb, I.p_intra);
    if (m.ref) {
        if (GET(fb, I.p_last)) m.ref = 2 + GET(fb, I.p_gf);
        const u32 my_bias = (I.sign_bias >> m.ref) & 1u;
        u32 near[4] = { 0, 0, 0, 0 };
        int cnt[4] = { 0, 0, 0, 0 };
        int n = 0;                                  // the most recently added candidate: near[n], cnt[n]
        // (near / cnt indexed by n: four entries, selects)
#define NEAR_SET(i, v) do { near[0] = (i) == 0 ? (v) : near[0]; near[1] = (i) == 1 ? (v) : near[1]; near[2] = (i) == 2 ? (v) : near[2]; near[3] = (i) == 3 ? (v) : near[3]; } while (0)
#define NEAR_GET(i) ((i) == 0 ? near[0] : (i) == 1 ? near[1] : (i) == 2 ? near[2] : near[3])
#define CNT_ADD(i, v) do { cnt[0] += (i) == 0 ? (v) : 0; cnt[1] += (i) == 1 ? (v) : 0; cnt[2] += (i) == 2 ? (v) : 0; cnt[3] += (i) == 3 ? (v) : 0; } while (0)
        if (above.ref != VP8IR_INTRA_FRAME) {
            if (above.mv) {
                const u32 t = ((I.sign_bias >> above.ref) & 1u) != my_bias ? mv_neg(above.mv) : above.mv;
                n++; NEAR_SET(n, t);
            }
            CNT_ADD(n, 2);
        }
        if (left.ref != VP8IR_INTRA_FRAME) {
            if (left.mv) {
                const u32 t = ((I.sign_bias >> left.ref) & 1u) != my_bias ? mv_neg(left.mv) : left.mv;
                if (t != NEAR_GET(n)) { n++; NEAR_SET(n, t); }
                CNT_ADD(n, 2);
            } else
                cnt[0] += 2;
        }
        if (aboveleft.ref != VP8IR_INTRA_FRAME) {
            if (aboveleft.mv) {
                const u32 t = ((I.sign_bias >> aboveleft.ref) & 1u) != my_bias ? mv_neg(aboveleft.mv) : aboveleft.mv;
                if (t != NEAR_GET(n)) { n++; NEAR_SET(n, t); }
                CNT_ADD(n, 1);
            } else
                cnt[0] += 1;
        }
        if (GET(fb, k_mode_contexts[cnt[0] * 4 + 0])) {
            Edges e;
            e.left = -((mb_col * 16) << 3) - (16 << 3);
            e.right = (((cols - 1 - mb_col) * 16) << 3) + (16 << 3);
            e.top = -((mb_row * 16) << 3) - (16 << 3);
            e.bottom = (((rows - 1 - mb_row) * 16) << 3) + (16 << 3);
            // three distinct candidates: the above-left one counts for NEAREST when they are equal
            if (cnt[3] && NEAR_GET(n) == near[1]) cnt[1] += 1;
            cnt[3] = ((above.ymode == VP8IR_SPLITMV) + (left.ymode == VP8IR_SPLITMV)) * 2 + (aboveleft.ymode == VP8IR_SPLITMV);
            if (cnt[2] > cnt[1]) {
                const int t = cnt[1]; const u32 tm = near[1];
                cnt[1] = cnt[2]; cnt[2] = t;
                near[1] = near[2]; near[2] = tm;
            }
            if (GET(fb, k_mode_contexts[cnt[1] * 4 + 1])) {
                if (GET(fb, k_mode_contexts[cnt[2] * 4 + 2])) {
                    if (cnt[1] >= cnt[0]) near[0] = near[1];
                    const u32 best = mv_clamp(near[0], e);
                    if (GET(fb, k_mode_contexts[cnt[3] * 4 + 3])) {
                        // decode_split_mv (decodemv.c:252-321): 16x8, 8x16, 8x8 or 4x4; a part's vector from its left / above
                        // neighbours' or new
                        int sp = 3, nparts = 16;
                        if (GET(fb, 110)) {
                            sp = 2; nparts = 4;
                            if (GET(fb, 111)) { sp = GET(fb, 150); nparts = 2; }
                        }
                        for (int j = 0; j < nparts; j++) {
                            // the part's first block: 16x8 {0, 8}, 8x16 {0, 2}, 8x8 {0, 2, 8, 10}, 4x4 j
                            const int k = sp == 0 ? 8 * j : sp == 1 ? 2 * j : sp == 2 ? (j & 1) * 2 + (j >> 1) * 8 : j;
                            const u32 leftmv = (k & 3) ? mvs[k - 1] : left.bmv[k >> 2];
                            const u32 abovemv = k >= 4 ? mvs[k - 4] : above.bmv[k];
                            const uint8_t *pr = k_submv_prob[((abovemv == 0) << 2) | ((leftmv == 0) << 1) | (leftmv == abovemv)];
                            u32 v;
                            if (!GET(fb, pr[0])) v = leftmv;
                            else if (!GET(fb, pr[1])) v = abovemv;
                            else if (!GET(fb, pr[2])) v = 0;
                            else v = mv_add(read_mv(fb, data, limit, I.mvc), best);
                            m.clamp |= mv_outside(v, e);
                            for (int bb = 0; bb < 16; bb++) {
                                const int part = sp == 0 ? bb >> 3 : sp == 1 ? (bb >> 1) & 1 : sp == 2 ? ((bb >> 3) << 1) | ((bb >> 1) & 1) : bb;
                                if (part == j) mvs[bb] = v;
                            }
                        }
                        m.part = sp;
                        mv = mvs[15];
                        m.ymode = VP8IR_SPLITMV;
                    } else {
                        mv = mv_add(read_mv(fb, data, limit, I.mvc), best);
                        m.clamp = mv_outside(mv, e);
                        m.ymode = VP8IR_NEWMV;
                    }
                } else {
                    m.ymode = VP8IR_NEARMV;
                    mv = mv_clamp(near[2], e);
                }
            } else {
                m.ymode = VP8IR_NEARESTMV;
                mv = mv_clamp(near[1], e);
            }
        } else {
            m.ymode = VP8IR_ZEROMV;
            mv = 0;
        }
#undef NEAR_SET
#undef NEAR_GET
#undef CNT_ADD
        if (m.ymode != VP8IR_SPLITMV)
            for (int bb = 0; bb < 16; bb++) mvs[bb] = mv;
    } else {
        if (!GET(fb, I.ymode[0])) m.ymode = VP8IR_DC_PRED;
        else if (!GET(fb, I.ymode[1])) m.ymode = GET(fb, I.ymode[2]) ? VP8IR_H_PRED : VP8IR_V_PRED;
        else m.ymode = GET(fb, I.ymode[3]) ? VP8IR_B_PRED : VP8IR_TM_PRED;
        if (m.ymode == VP8IR_B_PRED) {
            Row pr;                                            // the nine fixed probabilities as a row (vp8_bmode_prob defaults)
            pr.w0 = 120u | 90u << 8 | 79u << 16 | 133u << 24; pr.w1 = 87u | 85u << 8 | 80u << 16 | 111u << 24; pr.w2 = 151u;
            u64 bm = 0;
            for (int i = 0; i < 16; i++) bm |= (u64)read_bmode(fb, data, limit, pr) << (4 * i);
            m.bm = bm;
        }
        if (!GET(fb, I.uvmode[0])) m.uvmode = VP8IR_DC_PRED;
        else if (!GET(fb, I.uvmode[1])) m.uvmode = VP8IR_V_PRED;
        else m.uvmode = GET(fb, I.uvmode[2]) ? VP8IR_TM_PRED : VP8IR_H_PRED;
        for (int bb = 0; bb < 16; bb++) mvs[bb] = 0;
    }
    self.mv = mv; self.ref = (u32)m.ref; self.ymode = (u32)m.ymode;
    return m;
}

// The macroblock's tokens (decode_macroblock, decodframe.c:100-130; vp8_decode_mb_tokens) and its place in the IR (the device
// form, include/vp8_ir.h): the record to out_mb (eight 16-byte pieces); a block with more than a first coefficient to the
// slot's block stream at index bw, which moves on; the Y2 block and lone first coefficients into the record.  A / lnz: the
// non-zero flags of the row above at this column and of the macroblock to the left (bits 0..3 Y, 4..5 U, 6..7 V, 8 Y2),
// updated.  desc / blk: the lane's staging in LDS.
__device__ __forceinline__ void read_mb_tokens(BD &tb, const uint8_t *__restrict__ data, u32 limit, const MbModes &m, const row_t *probs,
                                               const row_t *cat, u32 &A, u32 &lnz, u32 *desc, u32 *blk, u32x4 *out_blocks, u32 &bw, u32x4 *out_mb)
{
    const bool has_y2 = m.ymode != VP8IR_B_PRED && m.ymode != VP8IR_SPLITMV;
    int skip = m.skip;
#pragma unroll
    for (int i = 0; i < 32; i++) desc[i] = 0;
    if (m.ymode == VP8IR_B_PRED) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u32 four = (u32)(m.bm >> (16 * w)) & 0xffffu;  // modes 4w .. 4w+3, a nibble each -> a byte each (descriptor bytes 40..55)
            desc[10 + w] = (four & 15u) | (four >> 4 & 15u) << 8 | (four >> 8 & 15u) << 16 | (four >> 12 & 15u) << 24;
        }
    }
    desc[14] = bw;                                             // vp8ir_mb::sparse_first
    if (skip) {                                                // vp8_reset_mb_tokens_context (detokenize.c:70-85)
        A = has_y2 ? 0u : A & 0x100u;
        lnz = has_y2 ? 0u : lnz & 0x100u;
    } else if (bd_error(tb)) {
        // the partition has run out: no tokens, contexts and skip flag stay, no residual (decodframe.c:119-130)
    } else {
        int total = 0;
        for (int i = has_y2 ? -1 : 0; i < 24; i++) {
            // block order: Y2 (when there is one), 16 Y, 4 U, 4 V; its place among the 25 of the IR; its context bits
            const int k = i < 0 ? 24 : i;
            int abit, lbit, type, first = 0;
            if (i < 0) { abit = lbit = 8; type = 1; }
            else if (i < 16) { abit = i & 3; lbit = i >> 2; type = has_y2 ? 0 : 3; first = has_y2 ? 1 : 0; }
            else { const int j = i - 16; abit = 4 + ((j >> 2) << 1) + (j & 1); lbit = 4 + ((j >> 2) << 1) + ((j >> 1) & 1); type = 2; }
#pragma unroll
            for (int w = 0; w < 8; w++) blk[w] = 0;
            int nz;
            const int ctx = (int)((A >> abit) & 1) + (int)((lnz >> lbit) & 1);
            const int e = read_block(tb, data, limit, probs + type * 72, cat, ctx, first, (coef_t *)blk, nz);
            A = (A & ~(1u << abit)) | (u32)nz << abit;
            lnz = (lnz & ~(1u << lbit)) | (u32)nz << lbit;
            ((uint8_t *)desc)[8 + k] = (uint8_t)e;
            total += e;
            if (i < 0) {                                       // the Y2 block: with the record, whatever it holds (vp8ir_mbx::y2)
#pragma unroll
                for (int w = 0; w < 8; w++) desc[16 + w] = blk[w];
            } else if (e > 1) {
                out_blocks[2 * (size_t)bw] = (u32x4){ blk[0], blk[1], blk[2], blk[3] };
                out_blocks[2 * (size_t)bw + 1] = (u32x4){ blk[4], blk[5], blk[6], blk[7] };
                bw++;
            } else if (e == 1 && !(has_y2 && i < 16)) {        // a lone first coefficient: y2[k] (no Y2 block), cdc[k - 16]
                ((unsigned short *)desc)[i < 16 ? 32 + i : 48 + (i - 16)] = (unsigned short)(blk[0] & 0xffffu);
            }
        }
        if (has_y2) total -= 16;                               // (the sixteen luma blocks started at position 1)
        if (total == 0) {                                      // decodframe.c:129: nothing coded after all (no block, no lone coefficient: the record's are zeros)
            skip = 1;
#pragma unroll
            for (int w = 2; w < 9; w++) desc[w] = 0;           // (eobs live in bytes 8..32; 33..35 are reserved zeros)
        }
    }
    desc[0] = (u32)m.ymode | (u32)m.uvmode << 8 | (u32)m.ref << 16 | (u32)((skip ? VP8IR_MB_SKIP : 0) | (m.clamp ? VP8IR_MB_CLAMP : 0)) << 24;
    desc[1] = (u32)m.seg | (u32)m.part << 8;
#pragma unroll
    for (int w = 0; w < 8; w++) out_mb[w] = (u32x4){ desc[4 * w], desc[4 * w + 1], desc[4 * w + 2], desc[4 * w + 3] };
}

}  // namespace

extern "C" size_t vp8_entropy_lds_bytes(int lpw) { return (size_t)lpw * (ENT_PROB_WORDS + ENT_DESC_WORDS + ENT_BLK_WORDS + 17) * 4; }

// frames: `count` of them; frame f goes to the IR slot at slot_base + (first_slot + f) * slot_bytes (records at o_mbx, block
// stream at o_blocks, vectors at o_mvs).  pool (or null): the context's block pool (vp8hip_configure_pooled) -- the slots then
// have no block streams of their own; a lane takes a chunk of chunk_blocks blocks out of the pool (*pool_ctr: the next free
// chunk; pool_chunks of them, and one more behind them that takes what no longer fits) whenever what is left of its chunk
// would not hold a macroblock row's worst case, so a row's blocks stay together and sparse_first counts from the pool's start.  lpw: lanes of each wave that carry a frame (1..64).  scratch: per frame (8 * mb_cols +
// 64) words (the row above's sub-block modes and non-zero flags per macroblock column, the token partitions' decoder states,
// inter frames: the row above's records).
extern "C" __global__ void __launch_bounds__(64)
vp8_entropy_kernel(const vp8hip_entropy_frame *__restrict__ frames, int count, int lpw, const uint8_t *__restrict__ all_data, DevGeom g,
                   size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbx, size_t o_blocks, size_t o_mvs, int first_slot,
                   u32 *__restrict__ scratch, u32 *__restrict__ status, char *pool, u32 *pool_ctr, u32 pool_chunks, u32 chunk_blocks)
{
    // LDS by lanes that carry a frame (the launch says how much: vp8_entropy_lds_bytes): probabilities, descriptor, block
    extern __shared__ u32 s_dyn[];
    row_t *s_probs = (row_t *)s_dyn;
    u32 *s_desc = s_dyn + lpw * ENT_PROB_WORDS;
    u32 *s_blk = s_desc + lpw * ENT_DESC_WORDS;
    u32 *s_mvs = s_blk + lpw * ENT_BLK_WORDS;           // inter frames: the macroblock's sixteen block vectors (17 words a lane)
    __shared__ row_t s_kfb[100 * 3];                    // kf_bmode_probs, a row per (above, left)
    __shared__ row_t s_cat[6 * 3];
    const int lane = threadIdx.x;
    const int f = blockIdx.x * lpw + lane;
    for (int i = lane; i < 100 * ENT_ROW; i += 64) {
        const int row = i / ENT_ROW, k = i - row * ENT_ROW;
        ((uint8_t *)s_kfb)[i] = k < 9 ? k_kf_bmode_probs[row * 9 + k] : (uint8_t)0;
    }
    for (int i = lane; i < 6 * ENT_ROW; i += 64) ((uint8_t *)s_cat)[i] = k_cat_rows[i];
    __syncthreads();
    if (lane >= lpw || f >= count) return;
    const vp8hip_entropy_frame &F = frames[f];
    const int cols = g.mb_cols, rows = g.mb_rows;
    u32 *abm = scratch + (size_t)f * (8 * cols + 64);   // the row above: four sub-block modes per macroblock column, a nibble each
    u32 *anz = abm + cols;                              // ... and its non-zero flags: bits 0..3 Y, 4..5 U, 6..7 V, 8 Y2
    u32 *tst = anz + cols;                              // token partitions' states: 8 words each
    u32 *anb = tst + 64;                                // inter frames: the row above's records (Nb), six words a column
    u32 *mvs = s_mvs + lane * 17;
    row_t *probs = s_probs + lane * ENT_PROB_WORDS;
    u32 *desc = s_desc + lane * ENT_DESC_WORDS;
    u32 *blk = s_blk + lane * ENT_BLK_WORDS;
    for (int row = 0; row < 96; row++) {                               // [type][band][context] rows of 11 -> rows of 12
        const uint8_t *src = F.coef_probs + row * 11;
        uint8_t *dst = (uint8_t *)probs + row * ENT_ROW;
        for (int k = 0; k < 11; k++) dst[k] = src[k];
        dst[11] = 0;
    }
    for (int c = 0; c < cols; c++) { abm[c] = 0; anz[c] = 0; }         // outside the frame: B_DC_PRED, nothing coded
    const bool inter = F.hdr.frame_type != 0;
    if (inter) for (int i = 0; i < 6 * cols; i++) anb[i] = 0;          // ... intra, no vector
    // positions are relative to the frame's first byte; what may be read: to the end of the launch's data (followed by padding)
    const uint8_t *__restrict__ data = all_data + F.data_off;
    const u32 limit = data_bytes - F.data_off < 0xfffffff0ull ? (u32)(data_bytes - F.data_off) : 0xfffffff0u;
    const u32 base = 0;
    const int ntok = (int)F.num_tok;
    for (int k = 0; k < ntok; k++) {                                   // a fresh decoder per partition (vp8dx_start_decode)
        u32 *t = tst + 8 * k;
        t[0] = 0; t[1] = (u32)-8; t[2] = 255; t[3] = base + F.tok_pos[k]; t[4] = base + F.tok_end[k]; t[5] = 0;
    }
    BD fb;                                                             // first partition: where the host's header parse stopped
    fb.value = F.first_value; fb.bits = F.first_bits; fb.range = F.first_range; fb.end = base + F.first_end;
#ifdef ENT_STATS
    fb.count = 0; u32 tcount = 0;
#endif
    request3(fb, data, base + F.first_pos, limit);
    fb.pos = base + F.first_pos + 3;
    const ModeParams MP = { F.update_mb_segmentation_map != 0, F.mb_no_coeff_skip != 0, F.prob_skip_false, F.segment_tree_probs[0],
                            F.segment_tree_probs[1], F.segment_tree_probs[2] };
    char *slot = slot_base + slot_bytes * (size_t)(first_slot + f);
    u32x4 *out_mbs = (u32x4 *)(slot + o_mbx);
    u32x4 *out_blocks = pool ? (u32x4 *)pool : (u32x4 *)(slot + o_blocks);
    u32x4 *out_mvs = (u32x4 *)(slot + o_mvs);
    u32 bw = 0;                                                        // blocks written to the slot's stream so far (pool: where the next one goes)
    u32 bw_end = pool ? 0u : 0xffffffffu;                              // pool: the end of the lane's chunk
    bool pool_full = false;
    const InterParams IP = { F.prob_intra, F.prob_last, F.prob_gf, { F.ymode_prob[0], F.ymode_prob[1], F.ymode_prob[2], F.ymode_prob[3] },
                             { F.uvmode_prob[0], F.uvmode_prob[1], F.uvmode_prob[2] },
                             (u32)F.hdr.sign_bias_golden << VP8IR_GOLDEN_FRAME | (u32)F.hdr.sign_bias_alt << VP8IR_ALTREF_FRAME, &F.mvc[0][0] };
    bool bad = false;

    for (int r = 0; r < rows; r++) {
        BD tb;
        if (bw_end - bw < (u32)cols * VP8IR_MAX_BLOCKS_PER_MB) {      // (pool only) the next chunk
            u32 ch = atomicAdd(pool_ctr, 1u);
            if (ch >= pool_chunks) { ch = pool_chunks; pool_full = true; }
            bw = ch * chunk_blocks; bw_end = bw + chunk_blocks;
        }
        {
            const u32 *t = tst + 8 * (r & (ntok - 1));                 // round robin by row (decodframe.c:1116-1129)
            tb.value = t[0]; tb.bits = (int)t[1]; tb.range = t[2]; tb.end = t[4];
#ifdef ENT_STATS
            tb.count = 0;
#endif
            if (r < ntok) { request3(tb, data, t[3], limit); tb.pos = t[3] + 3; }
            else { tb.n01 = t[6] >> 8; tb.n2 = t[6] & 255u; tb.pos = t[3]; }
        }
        u32 lbm = 0, lnz = 0;                                          // left of the row: B_DC_PRED, nothing coded
        Nb left = { 0, { 0, 0, 0, 0 }, 0, 0 }, aboveleft = { 0, { 0, 0, 0, 0 }, 0, 0 };
        for (int c = 0; c < cols; c++) {
            const long n = (long)r * cols + c;
            MbModes m;
            // a kept segment map: the id the frame before left in this slot's record (descriptor byte 4), before it is overwritten
            const int kept_seg = F.segmap_keep ? (int)(((const u32 *)(out_mbs + n * 8))[1] & 3u) : 0;
            if (inter) {
                Nb above, self;
                u32 *a6 = anb + 6 * c;
                above.mv = a6[0]; above.bmv[0] = a6[1]; above.bmv[1] = a6[2]; above.bmv[2] = a6[3]; above.bmv[3] = a6[4];
                above.ref = a6[5] & 255u; above.ymode = a6[5] >> 8;
                m = read_mb_modes_inter(fb, data, limit, MP, IP, above, left, aboveleft, r, c, rows, cols, mvs, self, kept_seg);
                a6[0] = self.mv; a6[1] = mvs[12]; a6[2] = mvs[13]; a6[3] = mvs[14]; a6[4] = mvs[15]; a6[5] = self.ref | self.ymode << 8;
                aboveleft = above;
                left = self; left.bmv[0] = mvs[3]; left.bmv[1] = mvs[7]; left.bmv[2] = mvs[11]; left.bmv[3] = mvs[15];
#pragma unroll
                for (int w = 0; w < 4; w++) out_mvs[n * 4 + w] = (u32x4){ mvs[4 * w], mvs[4 * w + 1], mvs[4 * w + 2], mvs[4 * w + 3] };
            } else {
                u32 above = abm[c];
                m = read_mb_modes(fb, data, limit, MP, s_kfb, above, lbm, kept_seg);
                abm[c] = above;
            }
            u32 A = anz[c];
            read_mb_tokens(tb, data, limit, m, probs, s_cat, A, lnz, desc, blk, out_blocks, bw, out_mbs + n * 8);
            anz[c] = A;
        }
        bad |= bd_error(tb);
#ifdef ENT_STATS
        tcount += tb.count;
#endif
        {
            u32 *t = tst + 8 * (r & (ntok - 1));
            t[0] = tb.value; t[1] = (u32)tb.bits; t[2] = tb.range; t[3] = tb.pos; t[6] = tb.n01 << 8 | tb.n2;
        }
    }
    bad |= bd_error(fb);
#ifdef ENT_STATS      // decisions of the first partition in the low half, of the token partitions (the last row's) in the high half
    if (status) status[f] = (fb.count >> 4 & 0xffffu) | tcount >> 8 << 16;
#else
    if (status) status[f] = (bad ? 1u : 0u) | (pool_full ? 2u : 0u);
#endif
}

// Frames coded with several token partitions (2, 4 or 8: the encoder's --token-parts; macroblock row r is in partition r mod NP,
// decodframe.c:1116-1129), a PARTITION per lane.  The partitions of a frame are separate bool-coded streams; what ties them is
// the entropy context a macroblock takes from the one above it, which belongs to the partition before.  So the NP lanes of a
// frame walk their rows one macroblock behind each other -- lane p decodes macroblock (row, c) in the step after lane p - 1
// decoded (row - 1, c) -- like the lanes of the lane-per-row pixel kernels, the non-zero flags of the row above handed over
// through LDS (a word per macroblock column and frame).  64 / NP frames per wave.  The first partition (the modes: one stream)
// is decoded first, by each frame's lane 0, into a scratch array the token lanes read (12 bytes per macroblock).
// Every partition's blocks go to a region of the slot's block stream of its own (the worst case of its rows: the rows' blocks
// stay together, which is what the device form asks, include/vp8_ir.h).
// scratch per frame: mb_cols words (the modes' row above) + 3 words per macroblock.  cols <= ENT_PARTS_MAX_COLS.
#define ENT_PARTS_MAX_COLS 256
extern "C" __global__ void __launch_bounds__(64)
vp8_entropy_parts_kernel(const vp8hip_entropy_frame *__restrict__ frames, int count, int np, const uint8_t *__restrict__ all_data, DevGeom g,
                         size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbx, size_t o_blocks, int first_slot,
                         u32 *__restrict__ scratch, u32 *__restrict__ status)
{
    __shared__ row_t s_probs[32 * ENT_PROB_WORDS];     // per frame (at most 32 frames of two partitions in a wave)
    __shared__ u32 s_desc[64 * ENT_DESC_WORDS];
    __shared__ u32 s_blk[64 * ENT_BLK_WORDS];
    __shared__ row_t s_kfb[100 * 3];
    __shared__ row_t s_cat[6 * 3];
    __shared__ u32 s_anz[32 * ENT_PARTS_MAX_COLS / 2];  // per frame: cols words; frames * cols <= 32 * 128 = 8 * 512 words
    __shared__ u32 s_bad[32];
    const int lane = threadIdx.x;
    const int fpw = 64 / np, fi = lane / np, part = lane - fi * np;
    const int f = blockIdx.x * fpw + fi;
    const bool live = f < count;
    const int cols = g.mb_cols, rows = g.mb_rows, nmb = cols * rows;
    for (int i = lane; i < 100 * ENT_ROW; i += 64) {
        const int row = i / ENT_ROW, k = i - row * ENT_ROW;
        ((uint8_t *)s_kfb)[i] = k < 9 ? k_kf_bmode_probs[row * 9 + k] : (uint8_t)0;
    }
    for (int i = lane; i < 6 * ENT_ROW; i += 64) ((uint8_t *)s_cat)[i] = k_cat_rows[i];
    if (lane < 32) s_bad[lane] = 0;
    const vp8hip_entropy_frame &F = frames[live ? f : 0];
    row_t *probs = s_probs + fi * ENT_PROB_WORDS;
    u32 *anz = s_anz + fi * cols;
    for (int i = part; i < 96 * ENT_ROW; i += np) {     // the frame's lanes share the copying of its probabilities
        const int row = i / ENT_ROW, k = i - row * ENT_ROW;
        ((uint8_t *)probs)[i] = k < 11 ? F.coef_probs[row * 11 + k] : (uint8_t)0;
    }
    for (int c = part; c < cols; c += np) anz[c] = 0;
    u32 *abm = scratch + (size_t)(live ? f : 0) * (cols + 3 * (size_t)nmb);
    u32 *modes = abm + cols;
    const uint8_t *__restrict__ data = all_data + F.data_off;
    const u32 limit = data_bytes - F.data_off < 0xfffffff0ull ? (u32)(data_bytes - F.data_off) : 0xfffffff0u;
    u32 *desc = s_desc + lane * ENT_DESC_WORDS;
    u32 *blk = s_blk + lane * ENT_BLK_WORDS;
    bool bad = false;
    __syncthreads();

    // ---- the modes of the whole frame: lane 0 of the frame
    if (live && part == 0) {
        BD fb;
        fb.value = F.first_value; fb.bits = F.first_bits; fb.range = F.first_range; fb.end = F.first_end;
#ifdef ENT_STATS
        fb.count = 0;
#endif
        request3(fb, data, F.first_pos, limit);
        fb.pos = F.first_pos + 3;
        const ModeParams MP = { F.update_mb_segmentation_map != 0, F.mb_no_coeff_skip != 0, F.prob_skip_false, F.segment_tree_probs[0],
                                F.segment_tree_probs[1], F.segment_tree_probs[2] };
        for (int c = 0; c < cols; c++) abm[c] = 0;
        for (int r = 0; r < rows; r++) {
            u32 lbm = 0;
            for (int c = 0; c < cols; c++) {
                u32 above = abm[c];
                const int kept_seg = F.segmap_keep ? (int)(((const u32 *)((u32x4 *)(slot_base + slot_bytes * (size_t)(first_slot + f) + o_mbx) + ((size_t)r * cols + c) * 8))[1] & 3u) : 0;
                const MbModes m = read_mb_modes(fb, data, limit, MP, s_kfb, above, lbm, kept_seg);
                abm[c] = above;
                u32 *o = modes + 3 * ((size_t)r * cols + c);
                o[0] = (u32)m.ymode | (u32)m.uvmode << 8 | (u32)m.seg << 16 | (u32)m.skip << 24;
                o[1] = (u32)m.bm; o[2] = (u32)(m.bm >> 32);
            }
        }
        bad |= bd_error(fb);
    }
    // what lane 0 wrote to memory is read by the frame's other lanes from here on
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    // ---- the tokens: lane `part` takes rows part, part + np, ..., a macroblock behind the lane before it
    char *slot = slot_base + slot_bytes * (size_t)(first_slot + (live ? f : 0));
    u32x4 *out_mbs = (u32x4 *)(slot + o_mbx);
    u32x4 *out_blocks = (u32x4 *)(slot + o_blocks);
    // rows part, part + np, ...: the partitions before this one own ceil((rows - q) / np) rows each, q < part
    u32 bw = 0;
    for (int q = 0; q < part; q++) bw += (u32)((rows - q + np - 1) / np) * (u32)cols * VP8IR_MAX_BLOCKS_PER_MB;
    BD tb;
    tb.value = 0; tb.bits = -8; tb.range = 255; tb.end = F.tok_end[part];
#ifdef ENT_STATS
    tb.count = 0;
#endif
    request3(tb, data, F.tok_pos[part], limit);
    tb.pos = F.tok_pos[part] + 3;
    const int rounds = (rows + np - 1) / np, steps = rounds * cols + np - 1;
    int row = part, c = -part - 1;                      // (lane `part` starts `part` steps late)
    u32 lnz = 0;
    for (int t = 0; t < steps; t++) {
        if (++c == cols) { c = 0; row += np; lnz = 0; }
        const bool work = live && c >= 0 && row < rows;
        u32 A = 0;
        if (work) A = anz[c];                           // left there by the lane before, a step ago (or by nobody: row 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (work) {
            const long n = (long)row * cols + c;
            const u32 *mo = modes + 3 * n;
            const u32 m0 = mo[0];
            MbModes m;
            m.ymode = (int)(m0 & 255u); m.uvmode = (int)(m0 >> 8 & 255u); m.seg = (int)(m0 >> 16 & 255u); m.skip = (int)(m0 >> 24);
            m.ref = VP8IR_INTRA_FRAME; m.clamp = 0; m.part = 0;
            m.bm = (u64)mo[1] | (u64)mo[2] << 32;
            read_mb_tokens(tb, data, limit, m, probs, s_cat, A, lnz, desc, blk, out_blocks, bw, out_mbs + n * 8);
            anz[c] = A;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    bad |= bd_error(tb);
    if (live && bad) atomicOr(&s_bad[fi], 1u);
    __syncthreads();
    if (live && part == 0 && status) status[f] = s_bad[fi];
}
