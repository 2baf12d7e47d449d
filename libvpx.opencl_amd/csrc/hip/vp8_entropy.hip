// Entropy decoding of key frames on the device, one frame per LANE (gfx950).
//
// What it replaces: the per-macroblock half of the reference's CPU front end -- vp8_kfread_modes (vp8/decoder/decodemv.c:50-173:
// segment id, skip flag, intra modes with the sub-block modes' above / left contexts) and vp8_decode_mb_tokens
// (vp8/decoder/detokenize.c:183-405: the coefficient token tree over the bool decoder of vp8/decoder/dboolhuff.h:76-120) as
// decode_mb_row drives them (vp8/decoder/decodframe.c:293-470: left / above entropy contexts, vp8_reset_mb_tokens_context for
// skipped macroblocks, eobtotal == 0 turning a macroblock into a skipped one, token partitions taken round robin by macroblock
// row).  In this repository the same work is csrc/host/vp8_parser.c's read_modes / decode_row, at ~10 ms per 1080p frame and
// host core; what the kernel writes into a frame's IR slot -- descriptors and dense coefficients, include/vp8_ir.h -- is byte
// for byte what vp8_parser_decode_mbs writes (tests/test_gpu_entropy.py).
//
// A bool decoder is a serial machine: every decision needs range and window as the decision before left them.  So there is
// nothing to spread over lanes inside a partition, and a frame is one lane's work from its first macroblock to its last (the
// partitions of a frame with several are taken in macroblock-row order, as the reference's single thread takes them: the
// contexts of a row come from the row above, which belongs to another partition).  The frames of a batch run side by side.
// What decides the rate is the length of the dependent instruction chain per decision and how many different paths through
// the token tree the lanes of a wave are on at once, so: 32-bit window (a 64-bit one is two instructions per shift), the next
// three bytes of the partition requested when the three before are taken (the request has ~25 decisions to land), the frame's
// 1056 coefficient probabilities in LDS (a row of 1060 bytes per lane: consecutive lanes on different banks), the macroblock
// descriptor and the block being decoded assembled in LDS and written out whole (16-byte stores), and a launch parameter for
// how many lanes of a wave carry frames (fewer lanes: fewer paths per wave, more waves).  Integer only; no MFMA.
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
#define ENT_DESC_WORDS 17          // 64-byte descriptor + a word of padding
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
    if (b.bits < 0) {                                   // 1..8 bits of the top byte are spent: three more bytes under them
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

}  // namespace

// frames: `count` of them; frame f goes to the IR slot at slot_base + (first_slot + f) * slot_bytes (descriptors at o_mbs, dense
// coefficients at o_coef).  lpw: lanes of each wave that carry a frame (1..64).  scratch: per frame (2 * mb_cols + 64) words
// (the row above's sub-block modes and non-zero flags per macroblock column, the token partitions' decoder states).
extern "C" __global__ void __launch_bounds__(64)
vp8_entropy_kernel(const vp8hip_entropy_frame *__restrict__ frames, int count, int lpw, const uint8_t *__restrict__ all_data, DevGeom g,
                   size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbs, size_t o_coef, int first_slot,
                   u32 *__restrict__ scratch, u32 *__restrict__ status)
{
    __shared__ row_t s_probs[64 * ENT_PROB_WORDS];
    __shared__ u32 s_desc[64 * ENT_DESC_WORDS];
    __shared__ u32 s_blk[64 * ENT_BLK_WORDS];
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
    u32 *abm = scratch + (size_t)f * (2 * cols + 64);   // the row above: four sub-block modes per macroblock column, a nibble each
    u32 *anz = abm + cols;                              // ... and its non-zero flags: bits 0..3 Y, 4..5 U, 6..7 V, 8 Y2
    u32 *tst = anz + cols;                              // token partitions' states: 8 words each
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
    const bool seg_map = F.update_mb_segmentation_map != 0, has_skip = F.mb_no_coeff_skip != 0;
    const u32 p_skip = F.prob_skip_false, tp0 = F.segment_tree_probs[0], tp1 = F.segment_tree_probs[1], tp2 = F.segment_tree_probs[2];
    char *slot = slot_base + slot_bytes * (size_t)(first_slot + f);
    u32x4 *out_mbs = (u32x4 *)(slot + o_mbs);
    u32x4 *out_coef = (u32x4 *)(slot + o_coef);
    bool bad = false;

    for (int r = 0; r < rows; r++) {
        BD tb;
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
        for (int c = 0; c < cols; c++) {
            const long n = (long)r * cols + c;
            // ---- modes (vp8_kfread_modes, decodemv.c:50-173)
            int seg = 0;
            if (seg_map) seg = GET(fb, tp0) ? 2 + GET(fb, tp2) : GET(fb, tp1);
            int skip = has_skip ? GET(fb, p_skip) : 0;
            int ymode;
            if (!GET(fb, 145)) ymode = VP8IR_B_PRED;
            else if (!GET(fb, 156)) ymode = GET(fb, 163) ? VP8IR_V_PRED : VP8IR_DC_PRED;
            else ymode = GET(fb, 128) ? VP8IR_TM_PRED : VP8IR_H_PRED;
#pragma unroll
            for (int i = 0; i < 16; i++) desc[i] = 0;
            const u32 above = abm[c];
            if (ymode == VP8IR_B_PRED) {
                u64 bm = 0;                                            // the macroblock's sixteen modes, a nibble each
                for (int i = 0; i < 16; i++) {
                    const int A = i < 4 ? (int)((above >> (4 * i)) & 15) : (int)((bm >> (4 * (i - 4))) & 15);
                    const int L = (i & 3) == 0 ? (int)((lbm >> (i & 12)) & 15) : (int)((bm >> (4 * (i - 1))) & 15);
                    const int m = read_bmode(fb, data, limit, row_at(s_kfb + (A * 10 + L) * 3));
                    bm |= (u64)m << (4 * i);
                    ((uint8_t *)desc)[40 + i] = (uint8_t)m;
                }
                abm[c] = (u32)(bm >> 48);
                lbm = (u32)((bm >> 12) & 15) | (u32)((bm >> 28) & 15) << 4 | (u32)((bm >> 44) & 15) << 8 | (u32)((bm >> 60) & 15) << 12;
            } else {
                // the sub-block mode a whole-macroblock mode stands for in its neighbours' contexts (findnearmv.h:131-188)
                const u32 im = ymode == VP8IR_V_PRED ? VP8IR_B_VE_PRED : ymode == VP8IR_H_PRED ? VP8IR_B_HE_PRED
                             : ymode == VP8IR_TM_PRED ? VP8IR_B_TM_PRED : VP8IR_B_DC_PRED;
                abm[c] = lbm = im * 0x1111u;
            }
            int uvmode;
            if (!GET(fb, 142)) uvmode = VP8IR_DC_PRED;
            else if (!GET(fb, 114)) uvmode = VP8IR_V_PRED;
            else uvmode = GET(fb, 183) ? VP8IR_TM_PRED : VP8IR_H_PRED;

            // ---- tokens (decode_macroblock, decodframe.c:100-130; vp8_decode_mb_tokens)
            const bool has_y2 = ymode != VP8IR_B_PRED;
            u32 A = anz[c];
            if (skip) {                                                // vp8_reset_mb_tokens_context (detokenize.c:70-85)
                A = has_y2 ? 0u : A & 0x100u;
                lnz = has_y2 ? 0u : lnz & 0x100u;
            } else if (bd_error(tb)) {
                // the partition has run out: no tokens, contexts and skip flag stay, no residual (decodframe.c:119-130)
#pragma unroll
                for (int i = 0; i < 50; i++) out_coef[n * 50 + i] = (u32x4){ 0, 0, 0, 0 };
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
                    const int e = read_block(tb, data, limit, probs + type * 72, s_cat, ctx, first, (coef_t *)blk, nz);
                    A = (A & ~(1u << abit)) | (u32)nz << abit;
                    lnz = (lnz & ~(1u << lbit)) | (u32)nz << lbit;
                    ((uint8_t *)desc)[8 + k] = (uint8_t)e;
                    total += e;
                    out_coef[n * 50 + 2 * k] = (u32x4){ blk[0], blk[1], blk[2], blk[3] };
                    out_coef[n * 50 + 2 * k + 1] = (u32x4){ blk[4], blk[5], blk[6], blk[7] };
                }
                if (has_y2) total -= 16;                               // (the sixteen luma blocks started at position 1)
                else { out_coef[n * 50 + 48] = (u32x4){ 0, 0, 0, 0 }; out_coef[n * 50 + 49] = (u32x4){ 0, 0, 0, 0 }; }
                if (total == 0) {                                      // decodframe.c:129: nothing coded after all
                    skip = 1;
#pragma unroll
                    for (int w = 2; w < 9; w++) desc[w] = 0;           // (eobs live in bytes 8..32; 33..35 are reserved zeros)
                }
            }
            anz[c] = A;
            desc[0] = (u32)ymode | (u32)uvmode << 8 | (u32)(skip ? VP8IR_MB_SKIP : 0) << 24;
            desc[1] = (u32)seg;
#pragma unroll
            for (int w = 0; w < 4; w++) out_mbs[n * 4 + w] = (u32x4){ desc[4 * w], desc[4 * w + 1], desc[4 * w + 2], desc[4 * w + 3] };
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
    if (status) status[f] = bad ? 1u : 0u;
#endif
}
