/* Error concealment of the host feeder (VPX_CODEC_USE_ERROR_CONCEALMENT): motion vectors for macroblocks whose modes never
 * arrived, so that the pixel path -- unchanged: a concealed macroblock is an ordinary SPLITMV macroblock without residual -- has
 * something to predict from.  Included by vp8_parser.c behind its mbinfo type.
 *
 * Behavioural reference: vp8/decoder/error_concealment.c of a build configured --enable-error-concealment --
 *   vp8_estimate_missing_mvs (:408-416 with :205-406): a lost frame, or a frame whose first partition ends early.  Every 4x4
 *       block of the PREVIOUS frame that predicted from the last frame is moved on along its own motion vector; a block of the
 *       new frame gets the mean of the vectors of the blocks that land on it, weighted by the area they cover;
 *   vp8_interpolate_motion (:555-584 with :425-553): an intra macroblock whose residual is lost inside an otherwise decodable
 *       frame.  Its sixteen blocks get the mean of the vectors of the up to twenty blocks around the macroblock that refer to
 *       the last frame, weighted by the inverse distance.
 * Same integer arithmetic, rounding (C division) and traversal order as there, including the quirks a restatement has to keep to
 * land on the same vectors (noted inline); own data layout: an overlap keeps the vector's value, not a pointer to it. */
#ifndef VP8_EC_H
#define VP8_EC_H

#include <stdlib.h>
#include <string.h>

#define EC_MAX_OVERLAPS 16                       /* MAX_OVERLAPS, ec_types.h:14 */
typedef struct ec_node { int area; int32_t mv; } ec_node;           /* area 0: free (OVERLAP_NODE with bmi == NULL) */
typedef struct ec_block { ec_node n[EC_MAX_OVERLAPS]; } ec_block;   /* B_OVERLAP; sixteen of them are a macroblock's MB_OVERLAP */

static inline int ec_min(int a, int b) { return a < b ? a : b; }
static inline int ec_max(int a, int b) { return a > b ? a : b; }
static inline int ec_floor_q(int x, int q) { return x & -(1 << q); }         /* FLOOR(x, q), error_concealment.c:22 */
static inline int ec_row(int32_t mv) { return (int16_t)(mv & 0xffff); }
static inline int ec_col(int32_t mv) { return (int16_t)((uint32_t)mv >> 16); }
static inline int32_t ec_mv(int row, int col) { return (int32_t)((uint16_t)(int16_t)row | ((uint32_t)(uint16_t)(int16_t)col << 16)); }

/* vp8_check_mv_bounds (findnearmv.h:58-68) against the edges as given (no margins here) */
static inline int ec_out_of_bounds(int32_t mv, int left, int right, int top, int bottom)
{
    return (ec_col(mv) < left) | (ec_col(mv) > right) | (ec_row(mv) < top) | (ec_row(mv) > bottom);
}

/* A block of the previous frame at block position (b_row, b_col), moved back along `mv`: the areas it covers of the (at most four)
 * blocks it lands on are added to their lists.  blocks: the frame's ec_block array, 16 per macroblock, macroblocks in raster
 * order.  (vp8_calculate_overlaps + calculate_overlaps_mb + block_overlap + assign_overlap, error_concealment.c:68-245.) */
static void ec_spread_block(ec_block *blocks, int mb_rows, int mb_cols, int32_t mv, int b_row, int b_col)
{
    const int new_row = ((4 * b_row) << 3) - ec_row(mv), new_col = ((4 * b_col) << 3) - ec_col(mv);      /* Q3 pixels */
    int ob_row, ob_col, om_row, om_col, end_row, end_col, rr, rc;
    const long total = (long)mb_rows * mb_cols * 16;
    if (new_row >= ((16 * mb_rows) << 3) || new_col >= ((16 * mb_cols) << 3)) return;       /* landed outside the frame */
    if (new_row <= -32 || new_col <= -32) return;
    ob_row = ec_floor_q(new_row / 4, 3) >> 3;                /* block the upper-left corner lies in */
    ob_col = ec_floor_q(new_col / 4, 3) >> 3;
    om_row = ec_floor_q((ob_row * 8) / 4, 3) >> 3;          /* ... and its macroblock */
    om_col = ec_floor_q((ob_col * 8) / 4, 3) >> 3;
    end_row = ec_min(mb_rows - om_row, 2);
    end_col = ec_min(mb_cols - om_col, 2);
    /* a second macroblock is only reached from the last block row / column of the first */
    if (abs(new_row - 16 * om_row * 8) < ((3 * 4) << 3)) end_row = 1;
    if (abs(new_col - 16 * om_col * 8) < ((3 * 4) << 3)) end_col = 1;
    for (rr = 0; rr < end_row; rr++)
        for (rc = 0; rc < end_col; rc++) {
            const int m_row = om_row + rr, m_col = om_col + rc;
            /* (the reference steps the first BLOCK by one per MACROBLOCK step: from a last block row that is the next macroblock's
               first) */
            const int fb_row = ob_row + rr, fb_col = ob_col + rc;
            int rel_row, rel_col, first, e_row, e_col, r, c;
            if (m_row < 0 || m_col < 0) continue;
            rel_row = fb_row - m_row * 4;
            rel_col = fb_col - m_col * 4;
            first = ec_max(rel_row, 0) * 4 + ec_max(rel_col, 0);
            e_row = ec_min(4 + m_row * 4 - fb_row, 2);
            e_col = ec_min(4 + m_col * 4 - fb_col, 2);
            if (new_row >= 0 && (new_row & 0x1F) == 0) e_row = 1;        /* block-aligned: no second block */
            if (new_col >= 0 && (new_col & 0x1F) == 0) e_col = 1;
            if (new_row < m_row * 16 * 8) e_row = 1;                /* started in the macroblock before */
            if (new_col < m_col * 16 * 8) e_col = 1;
            for (r = 0; r < e_row; r++)
                for (c = 0; c < e_col; c++) {
                    const int b2_row = (fb_row + r) * 4 * 8, b2_col = (fb_col + c) * 4 * 8;
                    const int top = ec_max(new_row, b2_row), left = ec_max(new_col, b2_col);
                    const int right = ec_min(new_col + (4 << 3), b2_col + (4 << 3)), bottom = ec_min(new_row + (4 << 3), b2_row + (4 << 3));
                    const int area = (bottom - top) * (right - left);            /* Q6 */
                    const long at = ((long)m_row * mb_cols + m_col) * 16 + first + r * 4 + c;
                    int i;
                    if (area <= 0 || at < 0 || at >= total) continue;
                    for (i = 0; i < EC_MAX_OVERLAPS; i++)
                        if (blocks[at].n[i].area == 0) { blocks[at].n[i].area = area; blocks[at].n[i].mv = mv; break; }
                }
        }
}

/* estimate_mv (error_concealment.c:250-280) */
static int32_t ec_block_mv(const ec_block *b)
{
    int i, sum = 0, row_acc = 0, col_acc = 0;
    for (i = 0; i < EC_MAX_OVERLAPS && b->n[i].area; i++) {
        col_acc += b->n[i].area * ec_col(b->n[i].mv);
        row_acc += b->n[i].area * ec_row(b->n[i].mv);
        sum += b->n[i].area;
    }
    return sum > 0 ? ec_mv(row_acc / sum, col_acc / sum) : 0;          /* Q9 / Q6 = Q3 */
}

/* vp8_estimate_missing_mvs: the macroblocks from `first_corrupt` on become SPLITMV macroblocks predicted from the last frame.
 * mi / prev: macroblock (0, 0) of the current / previous frame's mode info, `stride` entries per row. */
static void ec_estimate_missing_mvs(ec_block *blocks, mbinfo *mi, const mbinfo *prev, int stride, int mb_rows, int mb_cols,
                                    unsigned first_corrupt)
{
    int r, c, k;
    memset(blocks, 0, sizeof(ec_block) * 16 * (size_t)mb_rows * mb_cols);
    for (r = 0; r < mb_rows; r++)
        for (c = 0; c < mb_cols; c++) {
            const mbinfo *pm = prev + r * stride + c;
            if (pm->ref_frame != VP8IR_LAST_FRAME) continue;              /* only vectors into the last frame can be carried on */
            for (k = 0; k < 16; k++) ec_spread_block(blocks, mb_rows, mb_cols, pm->b[k].mv, 4 * r + (k >> 2), 4 * c + (k & 3));
        }
    r = (int)(first_corrupt / (unsigned)mb_cols);
    c = (int)(first_corrupt - (unsigned)r * (unsigned)mb_cols);
    for (; r < mb_rows; r++, c = 0) {
        const int to_top = -(r * 16) * 8, to_bottom = ((mb_rows - 1 - r) * 16) << 3;
        for (; c < mb_cols; c++) {
            mbinfo *m = mi + r * stride + c;
            const int to_left = -(c * 16) * 8, to_right = ((mb_cols - 1 - c) * 16) << 3;
            const ec_block *b = blocks + ((long)r * mb_cols + c) * 16;
            int nz = 0;
            int16_t a_row = 0, a_col = 0;
            m->ref_frame = VP8IR_LAST_FRAME;
            m->y_mode = VP8IR_SPLITMV;
            m->uv_mode = VP8IR_DC_PRED;
            m->partitioning = 3;
            m->segment_id = 0;
            m->need_clamp = 0;
            for (k = 0; k < 16; k++) {                                     /* estimate_mb_mvs (:285-332) */
                const int row = k >> 2, col = k & 3;
                const int32_t mv = ec_block_mv(b + k);
                m->b[k].mv = mv;
                m->need_clamp |= (uint8_t)ec_out_of_bounds(mv, to_left + ((col * 4) << 3), to_right - ((col * 4) << 3),
                                                           to_top + ((row * 4) << 3), to_bottom - ((row * 4) << 3));
                /* the macroblock's own vector: the mean of the non-zero ones (accumulated in the 16 bits of an MV there) */
                if (mv != 0) { nz++; a_col = (int16_t)(a_col + ec_col(mv)); a_row = (int16_t)(a_row + ec_row(mv)); }
            }
            if (nz > 0) { a_col = (int16_t)(a_col / nz); a_row = (int16_t)(a_row / nz); }
            m->mv = ec_mv(a_row, a_col);
        }
    }
}

/* vp8_interpolate_motion for the macroblock m at (r, c): weights_q7[|dy|][|dx|] = round(128 / distance) */
static void ec_interpolate_motion(mbinfo *m, int stride, int r, int c, int mb_rows, int mb_cols)
{
    static const int weights_q7[5][5] = { { 0, 128, 64, 43, 32 }, { 128, 91, 57, 40, 31 }, { 64, 57, 45, 36, 29 },
                                          { 43, 40, 36, 30, 26 }, { 32, 31, 29, 26, 23 } };
    /* the twenty blocks around the macroblock, clockwise from the one above-left: position relative to the macroblock's first block */
    static const signed char pos[20][2] = { { -1, -1 }, { -1, 0 }, { -1, 1 }, { -1, 2 }, { -1, 3 }, { -1, 4 }, { 0, 4 }, { 1, 4 }, { 2, 4 },
                                            { 3, 4 }, { 4, 4 }, { 4, 3 }, { 4, 2 }, { 4, 1 }, { 4, 0 }, { 4, -1 }, { 3, -1 }, { 2, -1 },
                                            { 1, -1 }, { 0, -1 } };
    int ref[20];
    int32_t mvs[20];
    int i = 0, j, k;
    const int to_left = -(c * 16) * 8, to_right = ((mb_cols - 1 - c) * 16) << 3;
    const int to_top = -(r * 16) * 8, to_bottom = ((mb_rows - 1 - r) * 16) << 3;
    for (j = 0; j < 20; j++) { ref[j] = -1; mvs[j] = 0; }               /* -1: no such neighbour (MAX_REF_FRAMES there) */
#define EC_TAKE(mb, blk) do { const mbinfo *q_ = (mb); ref[i] = q_->ref_frame; mvs[i] = q_->b[blk].mv; } while (0)
    /* find_neighboring_blocks (:433-493): whatever the neighbours hold at this point of the frame -- decoded, estimated or
       interpolated before this macroblock */
    if (r > 0) {
        if (c > 0) EC_TAKE(m - stride - 1, 15);
        ++i;
        for (j = 12; j < 16; ++j, ++i) EC_TAKE(m - stride, j);
    } else
        i += 5;
    if (c < mb_cols - 1) {
        if (r > 0) EC_TAKE(m - stride + 1, 12);
        ++i;
        for (j = 0; j <= 12; j += 4, ++i) EC_TAKE(m + 1, j);
    } else
        i += 5;
    if (r < mb_rows - 1) {
        if (c < mb_cols - 1) EC_TAKE(m + stride + 1, 0);
        ++i;
        for (j = 0; j < 4; ++j, ++i) EC_TAKE(m + stride, j);
    } else
        i += 5;
    if (c > 0) {
        if (r < mb_rows - 1) EC_TAKE(m + stride - 1, 4);
        ++i;
        for (j = 3; j < 16; j += 4, ++i) EC_TAKE(m - 1, j);
    } else
        i += 5;
#undef EC_TAKE
    m->need_clamp = 0;
    for (k = 0; k < 16; k++) {                                           /* interpolate_mvs (:498-553) */
        const int row = k >> 2, col = k & 3;
        int w_sum = 0, row_sum = 0, col_sum = 0;
        m->b[k].mv = 0;
        for (j = 0; j < 20; j++) {
            const int w = weights_q7[abs(row - pos[j][0])][abs(col - pos[j][1])];
            if (ref[j] != VP8IR_LAST_FRAME) continue;
            w_sum += w;
            row_sum += w * ec_row(mvs[j]);                              /* Q7 * Q3 */
            col_sum += w * ec_col(mvs[j]);
        }
        if (w_sum > 0) {
            m->b[k].mv = ec_mv(row_sum / w_sum, col_sum / w_sum);
            m->need_clamp |= (uint8_t)ec_out_of_bounds(m->b[k].mv, to_left + ((col * 4) << 3), to_right - ((col * 4) << 3),
                                                       to_top + ((row * 4) << 3), to_bottom - ((row * 4) << 3));
        }
    }
    m->ref_frame = VP8IR_LAST_FRAME;
    m->y_mode = VP8IR_SPLITMV;
    m->uv_mode = VP8IR_DC_PRED;
    m->partitioning = 3;
    m->segment_id = 0;
}

#endif
