/* VP8 host feeder: compressed frame -> IR.  See vp8_parser.h.
 *
 * Written from the bitstream format (RFC 6386) with the reference's decoder as the behavioural
 * authority; the reference locations each step mirrors are cited inline.  Entropy decode is
 * inherently serial per partition and stays on the CPU.
 */
#include "vp8_parser.h"

#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vp8_boolreader.h"
#include "vp8_tables.h"

/* a polite spin: the x86 PAUSE hint where there is one, a compiler barrier elsewhere (the caller yields after a while) */
static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

/* ------------------------------------------------------------------------------------------
 * persistent per-MB state (the reference's MODE_INFO, vp8/common/blockd.h:168-184), kept with a
 * one-entry border on the top and left so neighbour look-ups need no edge tests
 * (alloccommon.c:111-120).
 * ---------------------------------------------------------------------------------------- */
typedef union bslot {
    int32_t mv;      /* low 16 bits row, high 16 bits col (MV {short row; short col}, mv.h:16-26) */
    uint8_t mode;
} bslot;

typedef struct mbinfo {
    uint8_t y_mode, uv_mode, ref_frame, skip;
    uint8_t segment_id, partitioning, need_clamp, rsv;
    int32_t mv;
    bslot   b[16];
} mbinfo;

#include "vp8_ec.h"              /* error concealment: motion vectors for macroblocks whose modes were lost */

typedef struct entropy_ctx {     /* ENTROPY_CONTEXT_PLANES (blockd.h:44-50) */
    uint8_t y[4], u[2], v[2], y2;
} entropy_ctx;

typedef struct frame_probs {     /* FRAME_CONTEXT (onyxc_int.h:45-57), the saveable part */
    uint8_t coef[4][8][3][11];
    uint8_t mvc[2][19];
    uint8_t ymode[4];
    uint8_t uvmode[3];
} frame_probs;

struct vp8_parser {
    int width, height, mb_cols, mb_rows;
    int have_key_frame;

    frame_probs fc, saved_fc;
    int restore_probs;           /* refresh_entropy_probs == 0 for the current frame */

    /* state that persists across frames (decodframe.c:826-919) */
    uint8_t segmentation_enabled, update_mb_segmentation_map, update_mb_segmentation_data;
    uint8_t mb_segment_abs_delta;
    int8_t  segment_quant[4], segment_lf[4];
    uint8_t segment_tree_probs[3];
    uint8_t mode_ref_lf_delta_enabled;
    int8_t  ref_lf_deltas[4], mode_lf_deltas[4];

    /* per-frame */
    vp8ir_frame_hdr hdr;
    int mb_no_coeff_skip;
    uint8_t prob_skip_false, prob_intra, prob_last, prob_gf;
    int sign_bias[4];
    vp8_boolreader first;        /* first partition (modes) */
    vp8_boolreader tok[8];       /* token partitions */
    const uint8_t *tok_start[8]; /* ... where each begins (vp8_parser_export_entropy) */
    const uint8_t *frame_data;   /* the frame, when it came as one buffer */
    int segmap_stale;            /* the last frame's macroblocks were decoded elsewhere (vp8_parser_export_entropy) */
    int device_segmap;           /* vp8_parser_set_device_segmap: the device keeps this stream's segment map (in the stream's IR slot) */
    int num_tok;
    int frame_open;

    mbinfo *mi_alloc, *mi;       /* mi points at MB (0,0); stride mb_cols + 1 */
    int mi_stride;
    entropy_ctx *above;
    int threads;                 /* vp8_parser_set_threads */
    int *progress;               /* per macroblock row, threaded token decode */

    /* error concealment (a reference build with CONFIG_ERROR_CONCEALMENT, decoder created with oxcf.error_concealment) */
    int ec_enabled;              /* vp8_parser_set_error_concealment */
    int key_concealed;           /* the key frame being decoded got inter macroblocks from vp8_interpolate_motion */
    int ec_active;               /* from the first inter frame after a complete key frame on (init_frame, decodframe.c:672-673) */
    int corrupted;               /* xd->corrupted of the frame being decoded */
    unsigned mvs_corrupt_from_mb;/* first macroblock whose modes the first partition no longer held (decodemv.c:639-655) */
    int frame_corrupt_residual;  /* a macroblock of this frame has lost its residual already */
    int independent_partitions, prev_independent;    /* decodframe.c:1036-1054, :1162-1166 */
    uint8_t prev_version;        /* what a lost frame inherits: vp8_setup_version is not run for it */
    int prev_log2n;              /* ... and the number of token partitions, when the bits that carry it are lost */
    mbinfo *prev_alloc, *prev_mi;/* the frame before's mode info (pc->prev_mi; the two arrays change places after every frame) */
    ec_block *overlaps;
    char err[96];
};

static const uint8_t kf_ymode_prob[4] = { 145, 156, 163, 128 };     /* RFC 6386 11.2 */
static const uint8_t kf_uvmode_prob[3] = { 142, 114, 183 };
static const uint8_t default_ymode_prob[4] = { 112, 86, 140, 37 };  /* RFC 6386 16.2 */
static const uint8_t default_uvmode_prob[3] = { 162, 101, 204 };
static const uint8_t inter_bmode_prob[9] = { 120, 90, 79, 133, 87, 85, 80, 111, 151 };

/* zig-zag scan giving the position inside OUR column-major 4x4 block:
 * reference order {0,1,4,8,5,2,3,6,9,12,13,10,7,11,14,15} (entropy.c vp8_default_zig_zag1d) with
 * each raster index r*4+c mapped to c*4+r. */
static const uint8_t zigzag_colmajor[16] = { 0, 4, 1, 2, 5, 8, 12, 9, 6, 3, 7, 10, 13, 14, 11, 15 };
static const uint8_t coef_band[16] = { 0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7 };

static int fail(vp8_parser *p, int code, const char *msg)
{
    snprintf(p->err, sizeof p->err, "%s", msg);
    p->frame_open = 0;
    return code;
}

const char *vp8_parser_error(const vp8_parser *p) { return p->err; }

vp8_parser *vp8_parser_create(void)
{
    vp8_parser *p = (vp8_parser *)calloc(1, sizeof *p);
    if (p) p->threads = 1;
    return p;
}

void vp8_parser_destroy(vp8_parser *p)
{
    if (!p) return;
    free(p->mi_alloc);
    free(p->prev_alloc);
    free(p->overlaps);
    free(p->above);
    free(p->progress);
    free(p);
}

int vp8_parser_peek(const uint8_t *data, size_t size, int *is_key, int *width, int *height)
{
    if (!data || size == 0) return VP8P_INVALID_PARAM;
    if (is_key) *is_key = 0;
    if (size >= 10 && !(data[0] & 1)) {
        const uint8_t *c = data + 3;
        int w, h;
        if (is_key) *is_key = 1;
        if (c[0] != 0x9d || c[1] != 0x01 || c[2] != 0x2a) return VP8P_UNSUP_BITSTREAM;
        w = (c[3] | (c[4] << 8)) & 0x3fff;
        h = (c[5] | (c[6] << 8)) & 0x3fff;
        if (width) *width = w;
        if (height) *height = h;
        if (!(w | h)) return VP8P_UNSUP_BITSTREAM;
        return VP8P_OK;
    }
    return VP8P_UNSUP_BITSTREAM;
}

static int resize(vp8_parser *p, int w, int h)
{
    int cols = (w + 15) >> 4, rows = (h + 15) >> 4;
    free(p->mi_alloc);
    free(p->above);
    free(p->prev_alloc);
    free(p->overlaps);
    p->prev_alloc = NULL; p->prev_mi = NULL; p->overlaps = NULL;
    p->mi_alloc = (mbinfo *)calloc((size_t)(cols + 1) * (rows + 1), sizeof(mbinfo));
    p->above = (entropy_ctx *)calloc((size_t)cols, sizeof(entropy_ctx));
    if (!p->mi_alloc || !p->above) return -1;
    p->mi_stride = cols + 1;
    p->mi = p->mi_alloc + p->mi_stride + 1;
    if (p->ec_enabled) {         /* vp8_alloc_frame_buffers (prev_mip, alloccommon.c) + vp8_alloc_overlap_lists (decodframe.c:793-802) */
        p->prev_alloc = (mbinfo *)calloc((size_t)(cols + 1) * (rows + 1), sizeof(mbinfo));
        p->overlaps = (ec_block *)calloc((size_t)cols * rows * 16, sizeof(ec_block));
        if (!p->prev_alloc || !p->overlaps) return -1;
        p->prev_mi = p->prev_alloc + p->mi_stride + 1;
    }
    p->width = w;
    p->height = h;
    p->mb_cols = cols;
    p->mb_rows = rows;
    return 0;
}

static int read_delta_q(vp8_boolreader *br)    /* get_delta_q, decodframe.c:307-324 */
{
    int v = 0;
    if (vp8br_bit(br)) {
        v = vp8br_literal(br, 4);
        if (vp8br_bit(br)) v = -v;
    }
    return v;
}

static int read_signed(vp8_boolreader *br, int nbits)
{
    int v = vp8br_literal(br, nbits);
    return vp8br_bit(br) ? -v : v;
}

/* ------------------------------------------------------------------------------------------
 * frame header
 * ---------------------------------------------------------------------------------------- */
int vp8_parser_begin_frame(vp8_parser *p, const uint8_t *data, size_t size, vp8ir_frame_hdr *out)
{
    return vp8_parser_begin_frame_fragments(p, &data, &size, 1, out);
}

/* read_is_valid (decodframe.c:445-449) */
static int span_ok(const uint8_t *start, size_t len, const uint8_t *end) { return start + len > start && start + len <= end; }

int vp8_parser_begin_frame_fragments(vp8_parser *p, const uint8_t *const *frags, const size_t *frag_sizes, int nfrags,
                                     vp8ir_frame_hdr *out)
{
    vp8ir_frame_hdr *h = &p->hdr;
    vp8_boolreader *br = &p->first;
    /* the frame header and the first partition are read from the first fragment only (decodframe.c:695-697) */
    static const uint8_t nothing[8] = { 0 };
    const uint8_t *data = nfrags > 0 && frags[0] ? frags[0] : nothing;
    const size_t size = nfrags > 0 && frags[0] ? frag_sizes[0] : 0;
    const uint8_t *end = data + size;
    const uint8_t *cur = data;
    size_t first_len;
    int is_key, i, j, lost = 0;
    /* error concealment is for whole buffers; with VPX_CODEC_USE_INPUT_FRAGMENTS the strict rules stay */
    p->frame_data = nfrags == 1 ? frags[0] : NULL;
#define EC_ON (p->ec_active && nfrags <= 1)

    p->frame_open = 0;
    p->err[0] = 0;
    memset(h, 0, sizeof *h);
    if (size < 3) {
        /* decodframe.c:709-724: a frame that never came is, with concealment, an inter frame whose every bit reads as zero and
           whose motion vectors are estimated; everything vp8_setup_version derives stays as the frame before left it */
        if (!EC_ON)
            return fail(p, VP8P_CORRUPT_FRAME, "Truncated packet");
        lost = 1;
        is_key = 0;
        h->frame_type = 1;
        h->version = p->prev_version;
        h->show_frame = 1;
        first_len = 0;
    } else {
    /* 3-byte frame tag (decodframe.c:727-731) */
    is_key = !(cur[0] & 1);
    h->frame_type = (uint8_t)(cur[0] & 1);
    h->version = (cur[0] >> 1) & 7;
    h->show_frame = (cur[0] >> 4) & 1;
    first_len = (size_t)((cur[0] | (cur[1] << 8) | (cur[2] << 16)) >> 5);
    cur += 3;
    /* (with concealment the reference goes on here and, the first partition being short, sets its token decoders up over
       memory behind the buffer, decodframe.c:533-541: not followed -- an error with or without concealment) */
    if (first_len > (size_t)(end - (data)))   /* reference checks data + len against data_end */
        return fail(p, VP8P_CORRUPT_FRAME, "Truncated packet or corrupt partition 0 length");
    }

    if (is_key) {
        int w, hgt;
        if (end - cur < 7)
            return fail(p, VP8P_CORRUPT_FRAME, "Truncated key frame header");
        if (cur[0] != 0x9d || cur[1] != 0x01 || cur[2] != 0x2a)
            return fail(p, VP8P_UNSUP_BITSTREAM, "Invalid frame sync code");
        w = (cur[3] | (cur[4] << 8)) & 0x3fff;      /* upper 2 bits: scaling, unused by the decoder */
        hgt = (cur[5] | (cur[6] << 8)) & 0x3fff;
        cur += 7;
        if (w <= 0) return fail(p, VP8P_CORRUPT_FRAME, "Invalid frame width");
        if (hgt <= 0) return fail(p, VP8P_CORRUPT_FRAME, "Invalid frame height");
        if (w != p->width || hgt != p->height) {
            if (resize(p, w, hgt))
                return fail(p, VP8P_MEM_ERROR, "Failed to allocate frame buffers");
        }
    }
    if ((!p->have_key_frame && !is_key) || p->width == 0 || p->height == 0)
        return fail(p, VP8P_CORRUPT_FRAME, "A stream must start with a complete key frame");

    h->width = (uint16_t)p->width;
    h->height = (uint16_t)p->height;
    h->mb_cols = (uint16_t)p->mb_cols;
    h->mb_rows = (uint16_t)p->mb_rows;

    /* init_frame (decodframe.c:605-687): key frames reset all adaptive state */
    p->corrupted = 0;
    p->prev_independent = p->independent_partitions;
    if (!is_key && p->have_key_frame && p->ec_enabled) p->ec_active = 1;
    if (is_key) {
        memcpy(p->fc.mvc, vp8t_default_mv_context, sizeof p->fc.mvc);
        memcpy(p->fc.ymode, default_ymode_prob, 4);
        memcpy(p->fc.uvmode, default_uvmode_prob, 3);
        memcpy(p->fc.coef, vp8t_default_coef_probs, sizeof p->fc.coef);
        memset(p->segment_quant, 0, 4);
        memset(p->segment_lf, 0, 4);
        p->mb_segment_abs_delta = 0;
        memset(p->ref_lf_deltas, 0, 4);
        memset(p->mode_lf_deltas, 0, 4);
        p->sign_bias[VP8IR_GOLDEN_FRAME] = 0;
        p->sign_bias[VP8IR_ALTREF_FRAME] = 0;
    }

    /* first partition */
    {
        const uint8_t *part0 = cur;
        size_t avail = (size_t)(end - cur);
        vp8br_init(br, part0, avail);    /* reference starts bc over [data, data_end) too */
    }
    if (is_key) {
        h->color_space = (uint8_t)vp8br_bit(br);
        h->clamping_type = (uint8_t)vp8br_bit(br);
    }

    /* segmentation (decodframe.c:826-875).  NB: update_mb_segmentation_map is deliberately NOT
       cleared when segmentation is disabled -- the reference leaves the stale value in place. */
    p->segmentation_enabled = (uint8_t)vp8br_bit(br);
    if (p->segmentation_enabled) {
        p->update_mb_segmentation_map = (uint8_t)vp8br_bit(br);
        p->update_mb_segmentation_data = (uint8_t)vp8br_bit(br);
        if (p->update_mb_segmentation_data) {
            p->mb_segment_abs_delta = (uint8_t)vp8br_bit(br);
            memset(p->segment_quant, 0, 4);
            memset(p->segment_lf, 0, 4);
            for (j = 0; j < 4; j++)
                p->segment_quant[j] = (int8_t)(vp8br_bit(br) ? read_signed(br, 7) : 0);
            for (j = 0; j < 4; j++)
                p->segment_lf[j] = (int8_t)(vp8br_bit(br) ? read_signed(br, 6) : 0);
        }
        if (p->update_mb_segmentation_map) {
            memset(p->segment_tree_probs, 255, 3);
            for (i = 0; i < 3; i++)
                if (vp8br_bit(br))
                    p->segment_tree_probs[i] = (uint8_t)vp8br_literal(br, 8);
        }
    }

    /* loop filter header (decodframe.c:877-919) */
    h->filter_type = (uint8_t)vp8br_bit(br);
    h->filter_level = (uint8_t)vp8br_literal(br, 6);
    h->sharpness_level = (uint8_t)vp8br_literal(br, 3);
    p->mode_ref_lf_delta_enabled = (uint8_t)vp8br_bit(br);
    if (p->mode_ref_lf_delta_enabled) {
        if (vp8br_bit(br)) {
            for (i = 0; i < 4; i++)
                if (vp8br_bit(br))
                    p->ref_lf_deltas[i] = (int8_t)read_signed(br, 6);
            for (i = 0; i < 4; i++)
                if (vp8br_bit(br))
                    p->mode_lf_deltas[i] = (int8_t)read_signed(br, 6);
        }
    }

    /* token partitions (setup_token_decoder, decodframe.c:501-592): the fragments are unpacked so that entry k of F / S is
       partition k (k = 0: header + first partition + size table); a fragment may hold several partitions, the sizes of all but
       the last come from the table, every one is checked against the end of the fragment it lies in */
    {
        int log2n = vp8br_literal(br, 2);
        int n, fi, none = 0;
        const uint8_t *sizes = lost ? data : data + 3 + (is_key ? 7 : 0) + first_len;
        /* (decodframe.c:510-514: the count only changes when the bits that carry it were really there) */
        if (!vp8br_error(br)) p->prev_log2n = log2n; else log2n = p->prev_log2n;
        n = 1 << log2n;
        const uint8_t *F[10];
        size_t S[10];
        if (nfrags > 9)
            return fail(p, VP8P_UNSUP_BITSTREAM, "Too many fragments");
        memset(F, 0, sizeof F);
        memset(S, 0, sizeof S);
        for (i = 0; i < nfrags; i++) { F[i] = frags[i]; S[i] = frag_sizes[i]; }
        if (lost) { F[0] = data; S[0] = size; }
        for (fi = 0; fi < nfrags && fi <= n; fi++) {
            size_t left = S[fi];
            const uint8_t *fend = F[fi] + left;
            if (fi == 0) {
                const size_t ext_first = (size_t)(sizes - F[0]) + (size_t)(3 * (n - 1));   /* first partition + the size table */
                if (sizes > end || ext_first > left) {
                    if (!(EC_ON && lost))
                        return fail(p, VP8P_CORRUPT_FRAME, "Truncated partition size data");
                    none = 1;               /* a lost frame has no token partitions (every residual is thrown away anyway) */
                    break;
                }
                left -= ext_first;
                if (left > 0) {                 /* the fragment goes on with token partitions */
                    S[0] = ext_first;
                    fi++;
                    F[fi] = F[0] + ext_first;
                }
            }
            while (left > 0) {
                /* read_available_partition_size (decodframe.c:456-497), partition fi - 1 */
                const int k = fi - 1;
                size_t len = (size_t)(fend - F[fi]);
                if (k < n - 1) {
                    if (span_ok(sizes + 3 * k, 3, end))
                        len = (size_t)(sizes[3 * k] | (sizes[3 * k + 1] << 8) | (sizes[3 * k + 2] << 16));
                    else if (!EC_ON)
                        return fail(p, VP8P_CORRUPT_FRAME, "Truncated partition size data");
                }
                if (!span_ok(F[fi], len, fend)) {
                    /* with concealment a partition is what is left of it (read_available_partition_size, decodframe.c:456-497) */
                    if (!EC_ON)
                        return fail(p, VP8P_CORRUPT_FRAME, "Truncated packet or corrupt partition length");
                    len = (size_t)(fend - F[fi]);
                }
                S[fi] = len;
                left -= len;
                if (left > 0) {
                    if (fi >= n)                /* more bytes than partitions: the reference asserts here */
                        return fail(p, VP8P_CORRUPT_FRAME, "Truncated packet or corrupt partition length");
                    fi++;
                    F[fi] = F[fi - 1] + len;
                }
            }
        }
        for (i = 0; i < n; i++) {     /* (a partition that never came: empty, the frame turns out corrupt) */
            if (none || !F[i + 1]) { vp8br_init(&p->tok[i], nothing, 0); p->tok_start[i] = NULL; }
            else { vp8br_init(&p->tok[i], F[i + 1], S[i + 1]); p->tok_start[i] = F[i + 1]; }
        }
        p->num_tok = n;
        h->num_token_partitions = (uint8_t)n;
    }

    /* quantiser indices (decodframe.c:926-943) */
    h->base_qindex = (uint8_t)vp8br_literal(br, 7);
    h->y1dc_delta_q = (int8_t)read_delta_q(br);
    h->y2dc_delta_q = (int8_t)read_delta_q(br);
    h->y2ac_delta_q = (int8_t)read_delta_q(br);
    h->uvdc_delta_q = (int8_t)read_delta_q(br);
    h->uvac_delta_q = (int8_t)read_delta_q(br);

    /* reference updates (decodframe.c:949-1018) */
    if (is_key) {
        h->refresh_golden = 1;
        h->refresh_alt = 1;
        h->copy_buffer_to_gf = 0;
        h->copy_buffer_to_arf = 0;
    } else {
        /* with concealment, a flag whose bit was not there takes the harmless value (decodframe.c:951-992): no golden / alt-ref
           refresh, no buffer copies */
        h->refresh_golden = (uint8_t)vp8br_bit(br);
        if (p->ec_enabled) p->corrupted |= vp8br_error(br);
        if (EC_ON && p->corrupted) h->refresh_golden = 0;
        h->refresh_alt = (uint8_t)vp8br_bit(br);
        if (p->ec_enabled) p->corrupted |= vp8br_error(br);
        if (EC_ON && p->corrupted) h->refresh_alt = 0;
        h->copy_buffer_to_gf = 0;
        if (!h->refresh_golden) h->copy_buffer_to_gf = (uint8_t)vp8br_literal(br, 2);
        if (p->ec_enabled) p->corrupted |= vp8br_error(br);
        if (EC_ON && p->corrupted) h->copy_buffer_to_gf = 0;
        h->copy_buffer_to_arf = 0;
        if (!h->refresh_alt) h->copy_buffer_to_arf = (uint8_t)vp8br_literal(br, 2);
        if (p->ec_enabled) p->corrupted |= vp8br_error(br);
        if (EC_ON && p->corrupted) h->copy_buffer_to_arf = 0;
        p->sign_bias[VP8IR_GOLDEN_FRAME] = vp8br_bit(br);
        p->sign_bias[VP8IR_ALTREF_FRAME] = vp8br_bit(br);
    }
    h->sign_bias_golden = (uint8_t)p->sign_bias[VP8IR_GOLDEN_FRAME];
    h->sign_bias_alt = (uint8_t)p->sign_bias[VP8IR_ALTREF_FRAME];

    p->restore_probs = !vp8br_bit(br);          /* refresh_entropy_probs == 0 */
    if (p->ec_enabled) p->corrupted |= vp8br_error(br);
    if (EC_ON && p->corrupted) p->restore_probs = 1;          /* (decodframe.c:998-1005) probabilities of a damaged frame do not stay */
    if (p->restore_probs)
        p->saved_fc = p->fc;
    h->refresh_last = (uint8_t)(is_key || vp8br_bit(br));
    if (p->ec_enabled) p->corrupted |= vp8br_error(br);
    if (EC_ON && p->corrupted) h->refresh_last = 1;           /* (:1013-1018) ... but the frame itself becomes the last frame */

    /* coefficient probability updates (decodframe.c:1036-1054); the token partitions are independent of each other's contexts
       when no probability depends on the context of the block before (what lets concealment keep the residual of intact partitions) */
    {
        uint8_t *cp = &p->fc.coef[0][0][0][0];
        p->independent_partitions = 1;
        for (i = 0; i < 1056; i++) {
            if (vp8br_get(br, vp8t_coef_update_probs[i]))
                cp[i] = (uint8_t)vp8br_literal(br, 8);
            if ((i / 11) % 3 > 0 && cp[i] != cp[i - 11]) p->independent_partitions = 0;
        }
    }

    p->mb_no_coeff_skip = vp8br_bit(br);

    /* mb_mode_mv_init (decodemv.c:178-224) */
    p->prob_skip_false = 0;
    if (p->mb_no_coeff_skip)
        p->prob_skip_false = (uint8_t)vp8br_literal(br, 8);
    if (!is_key) {
        p->prob_intra = (uint8_t)vp8br_literal(br, 8);
        p->prob_last = (uint8_t)vp8br_literal(br, 8);
        p->prob_gf = (uint8_t)vp8br_literal(br, 8);
        if (vp8br_bit(br))
            for (i = 0; i < 4; i++) p->fc.ymode[i] = (uint8_t)vp8br_literal(br, 8);
        if (vp8br_bit(br))
            for (i = 0; i < 3; i++) p->fc.uvmode[i] = (uint8_t)vp8br_literal(br, 8);
        for (i = 0; i < 2; i++)
            for (j = 0; j < 19; j++)
                if (vp8br_get(br, vp8t_mv_update_probs[i * 19 + j])) {
                    int x = vp8br_literal(br, 7);
                    p->fc.mvc[i][j] = (uint8_t)(x ? x << 1 : 1);
                }
    }

    /* publish persistent state into the IR header */
    h->segmentation_enabled = p->segmentation_enabled;
    h->mb_segment_abs_delta = p->mb_segment_abs_delta;
    memcpy(h->segment_quant, p->segment_quant, 4);
    memcpy(h->segment_lf, p->segment_lf, 4);
    h->mode_ref_lf_delta_enabled = p->mode_ref_lf_delta_enabled;
    memcpy(h->ref_lf_deltas, p->ref_lf_deltas, 4);
    memcpy(h->mode_lf_deltas, p->mode_lf_deltas, 4);

    if (!lost) p->prev_version = h->version;
    *out = *h;
    p->frame_open = 1;
    return VP8P_OK;
#undef EC_ON
}

int vp8_parser_conceals(const vp8_parser *p) { return p && p->ec_active; }

/* See vp8_parser.h.  The decoder states handed over are the host's, cut down to the device's 32-bit window: the bytes the
   64-bit window holds beyond that go back to the partition (the position moves back by whole bytes). */
int vp8_parser_export_entropy(vp8_parser *p, vp8hip_entropy_frame *out)
{
    const vp8_boolreader *br = &p->first;
    int i, bits, back;
    if (!p->frame_open) return fail(p, VP8P_ERROR, "export_entropy without begin_frame");
    if (!p->frame_data || p->ec_enabled ||
        (!p->device_segmap && p->hdr.frame_type != 0 && p->segmentation_enabled && !p->update_mb_segmentation_map)) {      /* (the frame stays open) */
        snprintf(p->err, sizeof p->err, "%s", "one buffer, no concealment, a segment map of its own: or the device cannot decode it");
        return VP8P_UNSUP_BITSTREAM;
    }
    if (br->zero_fill || p->corrupted) return fail(p, VP8P_CORRUPT_FRAME, "the frame header ran past the end of the data");
    memset(out, 0, sizeof *out);
    out->hdr = p->hdr;
    bits = br->bits;
    back = bits > 24 ? (bits - 24 + 7) >> 3 : 0;
    bits -= 8 * back;
    out->first_pos = (uint32_t)(br->cur - p->frame_data) - (uint32_t)back;
    out->first_end = (uint32_t)(br->end - p->frame_data);
    out->first_value = (uint32_t)(br->window >> 32) & ~((1u << (24 - (bits < 0 ? 0 : bits))) - 1u);
    if (bits < 0) out->first_value = (uint32_t)(br->window >> 32);
    out->first_bits = bits;
    out->first_range = br->range;
    out->num_tok = (uint32_t)p->num_tok;
    for (i = 0; i < p->num_tok; i++) {
        if (!p->tok_start[i]) return fail(p, VP8P_CORRUPT_FRAME, "a token partition is missing");
        out->tok_pos[i] = (uint32_t)(p->tok_start[i] - p->frame_data);
        out->tok_end[i] = (uint32_t)(p->tok[i].end - p->frame_data);
    }
    out->update_mb_segmentation_map = (uint8_t)(p->update_mb_segmentation_map && p->segmentation_enabled);
    /* a macroblock's segment id when the frame does not code it, as read_modes has it (decodemv.c:594-606): the id the frame
       before left -- with the update flag set and segmentation off, and in inter frames --, or 0 (key frames).  Kept ids live in
       the stream's IR slot, which only a caller that decodes every frame of the stream there may rely on */
    out->segmap_keep = (uint8_t)(p->device_segmap && !out->update_mb_segmentation_map &&
                                 (p->update_mb_segmentation_map || p->hdr.frame_type != 0));
    out->mb_no_coeff_skip = (uint8_t)p->mb_no_coeff_skip;
    out->prob_skip_false = p->prob_skip_false;
    memcpy(out->segment_tree_probs, p->segment_tree_probs, 3);
    memcpy(out->coef_probs, p->fc.coef, 1056);
    out->prob_intra = p->prob_intra; out->prob_last = p->prob_last; out->prob_gf = p->prob_gf;
    memcpy(out->ymode_prob, p->fc.ymode, 4);
    memcpy(out->uvmode_prob, p->fc.uvmode, 3);
    memcpy(out->mvc, p->fc.mvc, 38);
    /* the frame is over for the parser (what vp8_parser_decode_mbs does at its end).  The macroblocks' modes never came by here;
       the one thing a later frame takes from them is the segment map when it does not bring its own: such a frame is refused
       until a frame decoded on the host has brought one (segmap_stale) */
    if (p->restore_probs) {
        p->fc = p->saved_fc;
        p->independent_partitions = p->prev_independent;
    }
    if (p->hdr.frame_type == 0) p->have_key_frame = 1;
    p->segmap_stale = 1;
    p->frame_open = 0;
    return VP8P_OK;
}

void vp8_parser_frame_hdr(const vp8_parser *p, vp8ir_frame_hdr *out)
{
    *out = p->hdr;
    if (p->key_concealed) { out->frame_type = 1; out->lf_key_frame = 1; }
}

/* ------------------------------------------------------------------------------------------
 * modes and motion vectors
 * ---------------------------------------------------------------------------------------- */
static int read_bmode(vp8_boolreader *br, const uint8_t *pr)   /* vp8_bmode_tree, entropymode.c */
{
    if (!vp8br_get(br, pr[0])) return VP8IR_B_DC_PRED;
    if (!vp8br_get(br, pr[1])) return VP8IR_B_TM_PRED;
    if (!vp8br_get(br, pr[2])) return VP8IR_B_VE_PRED;
    if (!vp8br_get(br, pr[3])) {
        if (!vp8br_get(br, pr[4])) return VP8IR_B_HE_PRED;
        return vp8br_get(br, pr[5]) ? VP8IR_B_VR_PRED : VP8IR_B_RD_PRED;
    }
    if (!vp8br_get(br, pr[6])) return VP8IR_B_LD_PRED;
    if (!vp8br_get(br, pr[7])) return VP8IR_B_VL_PRED;
    return vp8br_get(br, pr[8]) ? VP8IR_B_HU_PRED : VP8IR_B_HD_PRED;
}

static int read_uvmode(vp8_boolreader *br, const uint8_t *pr)
{
    if (!vp8br_get(br, pr[0])) return VP8IR_DC_PRED;
    if (!vp8br_get(br, pr[1])) return VP8IR_V_PRED;
    return vp8br_get(br, pr[2]) ? VP8IR_TM_PRED : VP8IR_H_PRED;
}

/* implied sub-block mode of a non-B_PRED MB, for the key-frame contexts
 * (above_block_mode / left_block_mode, findnearmv.h:131-188) */
static int implied_bmode(int y_mode)
{
    switch (y_mode) {
    case VP8IR_V_PRED: return VP8IR_B_VE_PRED;
    case VP8IR_H_PRED: return VP8IR_B_HE_PRED;
    case VP8IR_TM_PRED: return VP8IR_B_TM_PRED;
    default: return VP8IR_B_DC_PRED;
    }
}

static void read_kf_modes(vp8_parser *p, mbinfo *m)            /* decodemv.c:50-73 */
{
    vp8_boolreader *br = &p->first;
    const mbinfo *above = m - p->mi_stride, *left = m - 1;
    m->ref_frame = VP8IR_INTRA_FRAME;
    m->mv = 0;
    if (!vp8br_get(br, kf_ymode_prob[0]))
        m->y_mode = VP8IR_B_PRED;
    else if (!vp8br_get(br, kf_ymode_prob[1]))
        m->y_mode = vp8br_get(br, kf_ymode_prob[2]) ? VP8IR_V_PRED : VP8IR_DC_PRED;
    else
        m->y_mode = vp8br_get(br, kf_ymode_prob[3]) ? VP8IR_TM_PRED : VP8IR_H_PRED;

    if (m->y_mode == VP8IR_B_PRED) {
        int i;
        for (i = 0; i < 16; i++) {
            int A, L;
            if (i < 4)
                A = above->y_mode == VP8IR_B_PRED ? above->b[i + 12].mode : implied_bmode(above->y_mode);
            else
                A = m->b[i - 4].mode;
            if (!(i & 3))
                L = left->y_mode == VP8IR_B_PRED ? left->b[i + 3].mode : implied_bmode(left->y_mode);
            else
                L = m->b[i - 1].mode;
            m->b[i].mv = 0;
            m->b[i].mode = (uint8_t)read_bmode(br, &vp8t_kf_bmode_probs[(A * 10 + L) * 9]);
        }
    }
    m->uv_mode = (uint8_t)read_uvmode(br, kf_uvmode_prob);
}

typedef struct mv16 { int16_t row, col; } mv16;
static inline int32_t mv_pack(mv16 v) { return (int32_t)((uint16_t)v.row | ((uint32_t)(uint16_t)v.col << 16)); }
static inline mv16 mv_unpack(int32_t x) { mv16 v; v.row = (int16_t)(x & 0xffff); v.col = (int16_t)((uint32_t)x >> 16); return v; }

static int read_mv_component(vp8_boolreader *br, const uint8_t *pr)   /* decodemv.c:75-110 */
{
    int x = 0;
    if (vp8br_get(br, pr[0])) {                 /* long form: bits 0,1,2 then 9..4, bit 3 last */
        int i;
        for (i = 0; i < 3; i++) x += vp8br_get(br, pr[9 + i]) << i;
        for (i = 9; i > 3; i--) x += vp8br_get(br, pr[9 + i]) << i;
        if (!(x & 0xFFF0) || vp8br_get(br, pr[9 + 3])) x += 8;
    } else {                                    /* short form: 3-level tree over pr[2..8] */
        if (!vp8br_get(br, pr[2])) {
            if (!vp8br_get(br, pr[3])) x = vp8br_get(br, pr[4]);
            else x = 2 + vp8br_get(br, pr[5]);
        } else {
            if (!vp8br_get(br, pr[6])) x = 4 + vp8br_get(br, pr[7]);
            else x = 6 + vp8br_get(br, pr[8]);
        }
    }
    if (x && vp8br_get(br, pr[1])) x = -x;
    return x;
}

static mv16 read_mv(vp8_parser *p)
{
    mv16 v;
    v.row = (int16_t)(read_mv_component(&p->first, p->fc.mvc[0]) * 2);
    v.col = (int16_t)(read_mv_component(&p->first, p->fc.mvc[1]) * 2);
    return v;
}

static inline int32_t bias_mv(vp8_parser *p, int32_t mv, int neigh_ref, int this_ref)
{
    if (p->sign_bias[neigh_ref] != p->sign_bias[this_ref]) {   /* mv_bias, findnearmv.h:20-28 */
        mv16 v = mv_unpack(mv);
        v.row = (int16_t)(v.row * -1);
        v.col = (int16_t)(v.col * -1);
        return mv_pack(v);
    }
    return mv;
}

typedef struct edges { int left, right, top, bottom; } edges;   /* incl. the 16-pel margin */

static inline int32_t clamp_to(int32_t mv, const edges *e)      /* vp8_clamp_mv2, findnearmv.h:32-44 */
{
    mv16 v = mv_unpack(mv);
    if (v.col < e->left) v.col = (int16_t)e->left; else if (v.col > e->right) v.col = (int16_t)e->right;
    if (v.row < e->top) v.row = (int16_t)e->top; else if (v.row > e->bottom) v.row = (int16_t)e->bottom;
    return mv_pack(v);
}

static inline int out_of_bounds(int32_t mv, const edges *e)     /* vp8_check_mv_bounds */
{
    mv16 v = mv_unpack(mv);
    return (v.col < e->left) | (v.col > e->right) | (v.row < e->top) | (v.row > e->bottom);
}

static const uint8_t submv_prob[8][3] = {       /* vp8_sub_mv_ref_prob3, decodemv.c:226-236 */
    { 147, 136, 18 }, { 223, 1, 34 }, { 106, 145, 1 }, { 208, 1, 1 },
    { 179, 121, 1 }, { 223, 1, 34 }, { 179, 121, 1 }, { 208, 1, 1 }
};

static inline int split_part(int s, int b)      /* vp8_mbsplits, entropymode.c:60-90 */
{
    switch (s) {
    case 0: return b >> 3;
    case 1: return (b >> 1) & 1;
    case 2: return ((b >> 3) << 1) | ((b >> 1) & 1);
    default: return b;
    }
}

static void read_split_mvs(vp8_parser *p, mbinfo *m, int32_t best, const edges *e)   /* decodemv.c:252-321 */
{
    vp8_boolreader *br = &p->first;
    static const uint8_t first_block[4][16] = {
        { 0, 8 }, { 0, 2 }, { 0, 2, 8, 10 }, { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }
    };
    int s = 3, nparts = 16, j, b;
    if (vp8br_get(br, 110)) {
        s = 2; nparts = 4;
        if (vp8br_get(br, 111)) { s = vp8br_get(br, 150); nparts = 2; }
    }
    m->need_clamp = 0;
    for (j = 0; j < nparts; j++) {
        int k = first_block[s][j];
        int32_t leftmv, abovemv, mv;
        const uint8_t *pr;
        if (!(k & 3)) {
            const mbinfo *l = m - 1;
            leftmv = l->y_mode == VP8IR_SPLITMV ? l->b[k + 3].mv : l->mv;
        } else
            leftmv = m->b[k - 1].mv;
        if (k < 4) {
            const mbinfo *a = m - p->mi_stride;
            abovemv = a->y_mode == VP8IR_SPLITMV ? a->b[k + 12].mv : a->mv;
        } else
            abovemv = m->b[k - 4].mv;
        pr = submv_prob[((abovemv == 0) << 2) | ((leftmv == 0) << 1) | (leftmv == abovemv)];
        if (!vp8br_get(br, pr[0]))
            mv = leftmv;
        else if (!vp8br_get(br, pr[1]))
            mv = abovemv;
        else if (!vp8br_get(br, pr[2]))
            mv = 0;
        else {
            mv16 d = read_mv(p), bm = mv_unpack(best);
            d.row = (int16_t)(d.row + bm.row);
            d.col = (int16_t)(d.col + bm.col);
            mv = mv_pack(d);
        }
        m->need_clamp |= (uint8_t)out_of_bounds(mv, e);
        for (b = 0; b < 16; b++)
            if (split_part(s, b) == j) m->b[b].mv = mv;
    }
    m->partitioning = (uint8_t)s;
}

static void read_inter_modes(vp8_parser *p, mbinfo *m, int mb_row, int mb_col)   /* decodemv.c:323-569 */
{
    vp8_boolreader *br = &p->first;
    m->ref_frame = (uint8_t)vp8br_get(br, p->prob_intra);
    if (m->ref_frame) {
        const mbinfo *above = m - p->mi_stride, *left = m - 1, *aboveleft = above - 1;
        int32_t near[4] = { 0, 0, 0, 0 };
        int cnt[4] = { 0, 0, 0, 0 };
        int n = 0;      /* index of the most recently added candidate: near[n], cnt[n] */
        m->need_clamp = 0;
        if (vp8br_get(br, p->prob_last))
            m->ref_frame = (uint8_t)(2 + vp8br_get(br, p->prob_gf));

        if (above->ref_frame != VP8IR_INTRA_FRAME) {
            if (above->mv) {
                near[++n] = bias_mv(p, above->mv, above->ref_frame, m->ref_frame);
            }
            cnt[n] += 2;
        }
        if (left->ref_frame != VP8IR_INTRA_FRAME) {
            if (left->mv) {
                int32_t t = bias_mv(p, left->mv, left->ref_frame, m->ref_frame);
                if (t != near[n]) near[++n] = t;
                cnt[n] += 2;
            } else
                cnt[0] += 2;
        }
        if (aboveleft->ref_frame != VP8IR_INTRA_FRAME) {
            if (aboveleft->mv) {
                int32_t t = bias_mv(p, aboveleft->mv, aboveleft->ref_frame, m->ref_frame);
                if (t != near[n]) near[++n] = t;
                cnt[n] += 1;
            } else
                cnt[0] += 1;
        }

        if (vp8br_get(br, vp8t_mode_contexts[cnt[0] * 4 + 0])) {
            edges e;
            e.left = -((mb_col * 16) << 3) - (16 << 3);
            e.right = (((p->mb_cols - 1 - mb_col) * 16) << 3) + (16 << 3);
            e.top = -((mb_row * 16) << 3) - (16 << 3);
            e.bottom = (((p->mb_rows - 1 - mb_row) * 16) << 3) + (16 << 3);

            /* three distinct candidates: merge above-left with NEAREST if equal */
            if (cnt[3] && near[n] == near[1]) cnt[1] += 1;
            cnt[3] = ((above->y_mode == VP8IR_SPLITMV) + (left->y_mode == VP8IR_SPLITMV)) * 2
                     + (aboveleft->y_mode == VP8IR_SPLITMV);
            if (cnt[2] > cnt[1]) {
                int t = cnt[1]; int32_t tm = near[1];
                cnt[1] = cnt[2]; cnt[2] = t;
                near[1] = near[2]; near[2] = tm;
            }
            if (vp8br_get(br, vp8t_mode_contexts[cnt[1] * 4 + 1])) {
                if (vp8br_get(br, vp8t_mode_contexts[cnt[2] * 4 + 2])) {
                    int32_t best;
                    if (cnt[1] >= cnt[0]) near[0] = near[1];
                    best = clamp_to(near[0], &e);
                    if (vp8br_get(br, vp8t_mode_contexts[cnt[3] * 4 + 3])) {
                        read_split_mvs(p, m, best, &e);
                        m->mv = m->b[15].mv;
                        m->y_mode = VP8IR_SPLITMV;
                    } else {
                        mv16 d = read_mv(p), bm = mv_unpack(best);
                        d.row = (int16_t)(d.row + bm.row);
                        d.col = (int16_t)(d.col + bm.col);
                        m->mv = mv_pack(d);
                        m->need_clamp = (uint8_t)out_of_bounds(m->mv, &e);
                        m->y_mode = VP8IR_NEWMV;
                    }
                } else {
                    m->y_mode = VP8IR_NEARMV;
                    m->mv = clamp_to(near[2], &e);
                }
            } else {
                m->y_mode = VP8IR_NEARESTMV;
                m->mv = clamp_to(near[1], &e);
            }
        } else {
            m->y_mode = VP8IR_ZEROMV;
            m->mv = 0;
        }
        m->uv_mode = VP8IR_DC_PRED;
    } else {
        const uint8_t *yp = p->fc.ymode;
        m->mv = 0;
        m->need_clamp = 0;
        if (!vp8br_get(br, yp[0]))
            m->y_mode = VP8IR_DC_PRED;
        else if (!vp8br_get(br, yp[1]))
            m->y_mode = vp8br_get(br, yp[2]) ? VP8IR_H_PRED : VP8IR_V_PRED;
        else
            m->y_mode = vp8br_get(br, yp[3]) ? VP8IR_B_PRED : VP8IR_TM_PRED;
        if (m->y_mode == VP8IR_B_PRED) {
            int j;
            for (j = 0; j < 16; j++) {
                m->b[j].mv = 0;
                m->b[j].mode = (uint8_t)read_bmode(br, inter_bmode_prob);
            }
        }
        m->uv_mode = (uint8_t)read_uvmode(br, p->fc.uvmode);
    }
}

static void read_modes(vp8_parser *p)           /* vp8_decode_mode_mvs, decodemv.c:622-672 */
{
    vp8_boolreader *br = &p->first;
    int is_key = p->hdr.frame_type == 0;
    int r, c;
    p->mvs_corrupt_from_mb = ~0u;
    for (r = 0; r < p->mb_rows; r++) {
        mbinfo *m = p->mi + r * p->mi_stride;
        for (c = 0; c < p->mb_cols; c++, m++) {
            if (p->update_mb_segmentation_map) {
                if (p->segmentation_enabled) {  /* read_mb_features, decodemv.c:571-585 */
                    if (vp8br_get(br, p->segment_tree_probs[0]))
                        m->segment_id = (uint8_t)(2 + vp8br_get(br, p->segment_tree_probs[2]));
                    else
                        m->segment_id = (uint8_t)vp8br_get(br, p->segment_tree_probs[1]);
                }
            } else if (is_key)
                m->segment_id = 0;
            m->skip = p->mb_no_coeff_skip ? (uint8_t)vp8br_get(br, p->prob_skip_false) : 0;
            if (is_key)
                read_kf_modes(p, m);
            else
                read_inter_modes(p, m, r, c);
            if (p->ec_enabled) {
                /* a build with concealment keeps every inter macroblock's vector per block too (decodemv.c:538-557), and stops at
                   the macroblock where the first partition ran out (:639-655): the modes from there on are the estimator's */
                if (!is_key && m->ref_frame != VP8IR_INTRA_FRAME && m->y_mode != VP8IR_SPLITMV) {
                    int k;
                    for (k = 0; k < 16; k++) m->b[k].mv = m->mv;
                }
                if (vp8br_error(br)) {
                    p->mvs_corrupt_from_mb = (unsigned)(r * p->mb_cols + c);
                    return;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * coefficient tokens
 * ---------------------------------------------------------------------------------------- */
static int read_extra(vp8_boolreader *br, const uint8_t *probs, int n)
{
    int v = 0, i;
    for (i = 0; i < n; i++) v = (v << 1) | vp8br_get(br, probs[i]);
    return v;
}

/* One 4x4 block (the body of vp8_decode_mb_tokens, detokenize.c:262-378).  Returns the
 * reference's eob value ("c" at BLOCK_FINISHED); *nz = 1 iff the first token was not EOB. */
static int read_block_tokens(vp8_boolreader *br, const uint8_t (*probs)[3][11], int ctx, int first,
                             int16_t *out, int *nz)
{
    static const uint8_t cat2[2] = { 165, 145 }, cat3[3] = { 173, 148, 140 }, cat4[4] = { 176, 155, 140, 135 },
                         cat5[5] = { 180, 157, 141, 134, 130 },
                         cat6[11] = { 254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129 };
    int c = first;
    const uint8_t *pr = probs[coef_band[c]][ctx];
    *nz = 0;
    if (!vp8br_get(br, pr[0])) return c;
    *nz = 1;
    for (;;) {
        int v, next;
        while (!vp8br_get(br, pr[1])) {          /* DCT_0: no EOB test follows a zero */
            if (c == 15) return 15;              /* malformed input guard, as the reference */
            c++;
            pr = probs[coef_band[c]][0];
        }
        if (!vp8br_get(br, pr[2])) {
            v = 1; next = 1;
        } else {
            next = 2;
            if (!vp8br_get(br, pr[3])) {
                if (!vp8br_get(br, pr[4])) v = 2;
                else v = 3 + vp8br_get(br, pr[5]);
            } else if (!vp8br_get(br, pr[6])) {
                if (!vp8br_get(br, pr[7])) v = 5 + vp8br_get(br, 159);
                else v = 7 + read_extra(br, cat2, 2);
            } else if (!vp8br_get(br, pr[8])) {
                if (!vp8br_get(br, pr[9])) v = 11 + read_extra(br, cat3, 3);
                else v = 19 + read_extra(br, cat4, 4);
            } else {
                if (!vp8br_get(br, pr[10])) v = 35 + read_extra(br, cat5, 5);
                else v = 67 + read_extra(br, cat6, 11);
            }
        }
        if (vp8br_get(br, 128)) v = -v;
        out[zigzag_colmajor[c]] = (int16_t)v;
        if (c == 15) return 15;                  /* reference leaves c at 15 here (detokenize.c:140-146) */
        c++;
        pr = probs[coef_band[c]][next];
        if (!vp8br_get(br, pr[0])) return c;
    }
}

static int read_mb_tokens(vp8_parser *p, vp8_boolreader *brp, const mbinfo *m, entropy_ctx *A, entropy_ctx *L,
                          int16_t *coef, uint8_t *eobs)
{
    /* the decoder state in locals for the macroblock (a few hundred to a few thousand bool decodes): in registers, not behind a
       pointer the coefficient stores might alias */
    vp8_boolreader local = *brp, *br = &local;
    int total = 0, i, nz, first = 0, ytype = 3;
    if (m->y_mode != VP8IR_B_PRED && m->y_mode != VP8IR_SPLITMV) {
        int e = read_block_tokens(br, p->fc.coef[1], A->y2 + L->y2, 0, coef + 24 * 16, &nz);
        A->y2 = L->y2 = (uint8_t)nz;
        eobs[24] = (uint8_t)e;
        total += e - 16;
        first = 1;
        ytype = 0;
    }
    for (i = 0; i < 16; i++) {
        uint8_t *a = &A->y[i & 3], *l = &L->y[i >> 2];
        int e = read_block_tokens(br, p->fc.coef[ytype], *a + *l, first, coef + i * 16, &nz);
        *a = *l = (uint8_t)nz;
        eobs[i] = (uint8_t)e;
        total += e;
    }
    for (i = 16; i < 24; i++) {
        int k = i - 16;
        uint8_t *a = k < 4 ? &A->u[k & 1] : &A->v[k & 1];
        uint8_t *l = k < 4 ? &L->u[(k >> 1) & 1] : &L->v[(k >> 1) & 1];
        int e = read_block_tokens(br, p->fc.coef[2], *a + *l, 0, coef + i * 16, &nz);
        *a = *l = (uint8_t)nz;
        eobs[i] = (uint8_t)e;
        total += e;
    }
    *brp = local;
    return total;
}

static int decode_mbs(vp8_parser *p, vp8ir_mb *mbs, int16_t *coef, vp8ir_mbx *mbx, int16_t *blocks, size_t cap_blocks, size_t *nblocks,
                      vp8ir_mv *mvs, int *corrupt);

int vp8_parser_decode_mbs(vp8_parser *p, vp8ir_mb *mbs, int16_t *coef, vp8ir_mv *mvs, int *corrupt)
{
    return decode_mbs(p, mbs, coef, NULL, NULL, 0, NULL, mvs, corrupt);
}

int vp8_parser_decode_mbs_compact(vp8_parser *p, vp8ir_mbx *mbx, int16_t *blocks, size_t cap_blocks, size_t *nblocks, vp8ir_mv *mvs,
                                  int *corrupt)
{
    if (!mbx || !blocks || !nblocks) return fail(p, VP8P_INVALID_PARAM, "decode_mbs_compact: no output streams");
    return decode_mbs(p, NULL, NULL, mbx, blocks, cap_blocks, nblocks, mvs, corrupt);
}

/* ---- token partitions -----------------------------------------------------------------------
 * Macroblock row r is coded in partition r mod N (decodframe.c:1116-1129), and the partitions are independent bool-coder
 * streams (setup_token_decoder, decodframe.c:501-592): the only thing row r needs from row r-1 is the "block had
 * coefficients" context of the macroblock straight above.  With vp8_parser_set_threads(T > 1) the rows of a frame with
 * several partitions are decoded by up to min(T, N) threads, each owning the partitions t, t+T, ..., rows in order;
 * a row runs at most as far as the row above has got (per-row progress counters, release / acquire), which is how the
 * reference's own multi-threaded decoder synchronises (vp8/decoder/threading.c, sync_range).  The output is the serial
 * decoder's in the dense form; in the device form every thread writes its rows' blocks into a region of the block stream of
 * its own, closed up afterwards (a row's blocks stay together, which is all the form asks: include/vp8_ir.h). */
#define PROGRESS_STRIDE 16       /* ints: a cache line per row counter */
typedef struct tok_stream {      /* where a thread's blocks go (device form) */
    int16_t *blocks;
    size_t first_block;              /* region start inside the caller's array (blocks) */
    size_t cap_blocks;               /* blocks this region may take */
    size_t nb;
} tok_stream;

typedef struct tok_job {
    vp8_parser *p;
    vp8ir_mb *mbs;                   /* dense form: descriptors, coefficients ... */
    int16_t *coef;
    vp8ir_mbx *mbx;                  /* ... or the device form: records (+ the threads' block streams) */
    vp8ir_mv *mvs;
    int nthreads;
    volatile int *progress;          /* per row: macroblocks finished */
    int overflow;                 /* set by any thread (atomically): a block stream ran out of room */
} tok_job;

typedef struct tok_worker { tok_job *job; int id; tok_stream out; int bad; pthread_t thread; } tok_worker;

/* one macroblock row; returns nonzero when the thread's region of the block stream is full */
static int decode_row(tok_job *j, tok_stream *out, int r, vp8_boolreader *br)
{
    vp8_parser *p = j->p;
    int16_t *coef = j->coef;
    vp8ir_mv *mvs = j->mvs;
    entropy_ctx left;
    mbinfo *m = p->mi + r * p->mi_stride;
    int c, i;
    memset(&left, 0, sizeof left);
    for (c = 0; c < p->mb_cols; c++, m++) {
        size_t n = (size_t)r * p->mb_cols + c;
        vp8ir_mbx *x = j->mbx ? &j->mbx[n] : NULL;
        vp8ir_mb *o = x ? &x->d : &j->mbs[n];
        entropy_ctx *A = &p->above[c];
        const unsigned mb_idx = (unsigned)n;
        int has_y2;
        if (p->ec_active && m->ref_frame == VP8IR_INTRA_FRAME && (p->hdr.frame_type != 0 || mvs)
            && ((!p->independent_partitions && p->frame_corrupt_residual) || vp8br_error(br))) {
            /* decode_mb_row, decodframe.c:365-392: an intra macroblock whose residual is lost is better predicted from the last
               frame with vectors interpolated from its neighbours.  (The first macroblock to lose its residual finds out too
               late for that, as there.)  In a KEY frame too: the frame then reads a reference like an inter frame, which is what
               its header says to the pixel path afterwards (vp8_parser_frame_hdr), with lf_key_frame for the loop filter */
            ec_interpolate_motion(m, p->mi_stride, r, c, p->mb_rows, p->mb_cols);
            if (p->hdr.frame_type == 0) p->key_concealed = 1;
        }
        has_y2 = m->y_mode != VP8IR_B_PRED && m->y_mode != VP8IR_SPLITMV;
        if (j->progress && r > 0 && (c & 3) == 0) {   /* the four macroblocks above have left their context in p->above[] */
            const int need = c + 4 < p->mb_cols ? c + 4 : p->mb_cols;
            int spins = 0;
            while (__atomic_load_n(&j->progress[(r - 1) * PROGRESS_STRIDE], __ATOMIC_ACQUIRE) < need) {
                if (__atomic_load_n(&j->overflow, __ATOMIC_ACQUIRE)) return 1;
                if (++spins < 2000) cpu_relax(); else sched_yield();
            }
        }
        if (x) { memset(x, 0, sizeof *x); o->sparse_first = (uint32_t)(out->first_block + out->nb); }
        else memset(o, 0, sizeof *o);
        if (m->skip) {                       /* vp8_reset_mb_tokens_context, detokenize.c:70-85 */
            uint8_t ay2 = A->y2, ly2 = left.y2;
            memset(A, 0, sizeof *A);
            memset(&left, 0, sizeof left);
            if (!has_y2) { A->y2 = ay2; left.y2 = ly2; }
        } else if (vp8br_error(br)) {
            /* decode_macroblock, decodframe.c:119-130: once the partition has run out no tokens are read -- the macroblock is
               neither reset nor marked skipped (its inner edges are loop-filtered), it gets no residual: the reference adds
               the all-zero qcoeff through whatever eobs the macroblock before left behind, which is the prediction unchanged */
            if (coef) memset(coef + n * VP8IR_COEF_PER_MB, 0, VP8IR_COEF_PER_MB * sizeof(int16_t));
        } else {
            int16_t local[VP8IR_COEF_PER_MB];
            int16_t *q = coef ? coef + n * VP8IR_COEF_PER_MB : local;
            memset(q, 0, VP8IR_COEF_PER_MB * sizeof(int16_t));
            if (read_mb_tokens(p, br, m, A, &left, q, o->eobs) == 0) {
                m->skip = 1;                 /* decodframe.c:129: eobtotal==0 forces skip */
                memset(o->eobs, 0, 25);
            } else if (p->ec_active && (mb_idx >= p->mvs_corrupt_from_mb || vp8br_error(br)
                                        || (!p->independent_partitions && p->frame_corrupt_residual))) {
                /* decode_macroblock, decodframe.c:158-187: with concealment, a macroblock whose modes were estimated, whose
                   partition has just run out or -- unless the partitions are independent -- that follows one that lost its
                   residual keeps the prediction alone; its skip flag stays what the tokens said */
                memset(o->eobs, 0, 25);
                if (coef) memset(q, 0, VP8IR_COEF_PER_MB * sizeof(int16_t));
            } else if (x) {
                /* the device form (vp8_ir.h): the Y2 block and the lone first coefficients with the record, the blocks with more
                   than that into the stream, in block order */
                int k;
                if (has_y2) memcpy(x->y2, q + 24 * 16, 32);
                for (k = 0; k < 24; k++) {
                    if (o->eobs[k] > 1) {
                        if (out->nb >= out->cap_blocks) { __atomic_store_n(&j->overflow, 1, __ATOMIC_RELEASE); return 1; }
                        memcpy(out->blocks + (out->first_block + out->nb) * 16, q + k * 16, 32);
                        out->nb++;
                    } else if (o->eobs[k] == 1 && !(has_y2 && k < 16)) {
                        if (k < 16) x->y2[k] = q[k * 16]; else x->cdc[k - 16] = q[k * 16];
                    }
                }
            }
        }
        if (p->ec_active && (mb_idx >= p->mvs_corrupt_from_mb || vp8br_error(br) || (!p->independent_partitions && p->frame_corrupt_residual)))
            p->frame_corrupt_residual = 1;       /* (whether or not this macroblock had a residual to lose) */
        if (j->progress && ((c & 3) == 3 || c == p->mb_cols - 1)) __atomic_store_n(&j->progress[r * PROGRESS_STRIDE], c + 1, __ATOMIC_RELEASE);
        o->y_mode = m->y_mode;
        o->uv_mode = m->uv_mode;
        o->ref_frame = m->ref_frame;
        o->flags = (uint8_t)((m->skip ? VP8IR_MB_SKIP : 0) | (m->need_clamp ? VP8IR_MB_CLAMP : 0));
        o->segment_id = m->segment_id;
        o->partitioning = m->y_mode == VP8IR_SPLITMV ? m->partitioning : 0;
        if (m->y_mode == VP8IR_B_PRED)
            for (i = 0; i < 16; i++) o->b_modes[i] = m->b[i].mode;
        if (mvs) {
            vp8ir_mv *mv = mvs + n * 16;
            for (i = 0; i < 16; i++) {
                int32_t x = m->ref_frame == VP8IR_INTRA_FRAME ? 0
                          : (m->y_mode == VP8IR_SPLITMV ? m->b[i].mv : m->mv);
                mv16 v = mv_unpack(x);
                mv[i].row = v.row;
                mv[i].col = v.col;
            }
        }
    }
    return 0;
}

/* which thread decodes row r: the owner of its partition */
static inline int row_owner(const vp8_parser *p, int r, int nthreads) { return (r & (p->num_tok - 1)) % nthreads; }

static void *tok_worker_main(void *arg)
{
    tok_worker *w = (tok_worker *)arg;
    tok_job *j = w->job;
    vp8_parser *p = j->p;
    int r;
    for (r = 0; r < p->mb_rows; r++) {
        vp8_boolreader *br = &p->tok[r & (p->num_tok - 1)];
        if (row_owner(p, r, j->nthreads) != w->id) continue;
        if (decode_row(j, &w->out, r, br)) break;
        w->bad |= vp8br_error(br);
    }
    return NULL;
}

void vp8_parser_set_error_concealment(vp8_parser *p, int on)
{
    if (p && !p->width) p->ec_enabled = on != 0;      /* before the first frame only: the mode-info arrays are allocated with it */
}

void vp8_parser_set_device_segmap(vp8_parser *p, int on)
{
    if (p) p->device_segmap = on != 0;
}

void vp8_parser_set_threads(vp8_parser *p, int threads)
{
    if (p) p->threads = threads < 1 ? 1 : (threads > 8 ? 8 : threads);
}

/* mbx == NULL: the dense form (mbs, coef); else the device form (mbx, blocks) of vp8_ir.h */
static int decode_mbs(vp8_parser *p, vp8ir_mb *mbs, int16_t *coef, vp8ir_mbx *mbx, int16_t *blocks, size_t cap_blocks, size_t *nblocks,
                      vp8ir_mv *mvs, int *corrupt)
{
    int r, t, bad = 0, nthreads;
    int is_key;
    size_t nb = 0;
    tok_job job;
    tok_worker w[8];
    if (!p->frame_open)
        return fail(p, VP8P_ERROR, "decode_mbs without begin_frame");
    is_key = p->hdr.frame_type == 0;
    if (!is_key && !mvs)
        return fail(p, VP8P_INVALID_PARAM, "inter frame needs an mv array");
    if (p->segmap_stale && !is_key && p->segmentation_enabled && !p->update_mb_segmentation_map)
        return fail(p, VP8P_UNSUP_BITSTREAM, "the segment map this frame keeps belongs to a frame that was decoded on the device");
    /* (only a frame that writes every segment id makes the host's map current again: a frame with segmentation off leaves it alone) */
    if (is_key || (p->segmentation_enabled && p->update_mb_segmentation_map)) p->segmap_stale = 0;

    read_modes(p);
    bad |= vp8br_error(&p->first) | p->corrupted;
    if (p->ec_active && p->mvs_corrupt_from_mb < (unsigned)(p->mb_rows * p->mb_cols) && p->prev_mi)
        /* decodframe.c:1079-1086: the modes that did not arrive are estimated from the frame before */
        ec_estimate_missing_mvs(p->overlaps, p->mi, p->prev_mi, p->mi_stride, p->mb_rows, p->mb_cols, p->mvs_corrupt_from_mb);
    p->frame_corrupt_residual = 0;
    p->key_concealed = 0;

    memset(p->above, 0, (size_t)p->mb_cols * sizeof(entropy_ctx));
    memset(&job, 0, sizeof job);
    memset(w, 0, sizeof w);
    job.p = p; job.mbs = mbs; job.coef = coef; job.mbx = mbx; job.mvs = mvs;
    nthreads = p->threads < p->num_tok ? p->threads : p->num_tok;
    if (nthreads > p->mb_rows) nthreads = p->mb_rows;
    if (p->ec_active) nthreads = 1;              /* what a lost residual does to the macroblocks after it is decided in frame order */
    /* a thread's region of the block stream has to hold the worst case of its rows; with a smaller array the frame is decoded serially */
    if (nthreads > 1 && mbx && cap_blocks < (size_t)p->mb_rows * p->mb_cols * VP8IR_MAX_BLOCKS_PER_MB) nthreads = 1;
    if (nthreads > 1) {
        int *pr = (int *)realloc(p->progress, (size_t)p->mb_rows * PROGRESS_STRIDE * sizeof(int));
        if (pr) p->progress = pr; else nthreads = 1;
    }
    if (nthreads <= 1) {
        w[0].out.blocks = blocks; w[0].out.cap_blocks = cap_blocks;
        for (r = 0; r < p->mb_rows; r++) {
            vp8_boolreader *br = &p->tok[r & (p->num_tok - 1)];   /* round-robin, decodframe.c:1116-1129 */
            if (decode_row(&job, &w[0].out, r, br)) return fail(p, VP8P_MEM_ERROR, "block stream overflow");
            bad |= vp8br_error(br);
        }
        nb = w[0].out.nb;
    } else {
        size_t at = 0;
        int started = 0;
        memset(p->progress, 0, (size_t)p->mb_rows * PROGRESS_STRIDE * sizeof(int));
        job.progress = p->progress;
        job.nthreads = nthreads;
        for (t = 0; t < nthreads; t++) {
            size_t rows_t = 0;
            for (r = 0; r < p->mb_rows; r++) rows_t += row_owner(p, r, nthreads) == t;
            w[t].job = &job; w[t].id = t;
            w[t].out.blocks = blocks;
            w[t].out.first_block = at;
            w[t].out.cap_blocks = rows_t * p->mb_cols * VP8IR_MAX_BLOCKS_PER_MB;
            at += w[t].out.cap_blocks;
        }
        for (t = 1; t < nthreads; t++) {
            if (pthread_create(&w[t].thread, NULL, tok_worker_main, &w[t])) break;
            started = t;
        }
        if (started != nthreads - 1) {                    /* could not start them all: nobody may wait for a missing row */
            __atomic_store_n(&job.overflow, 1, __ATOMIC_RELEASE);
            for (t = 1; t <= started; t++) pthread_join(w[t].thread, NULL);
            return fail(p, VP8P_MEM_ERROR, "cannot start the token partition threads");
        }
        tok_worker_main(&w[0]);
        for (t = 1; t < nthreads; t++) pthread_join(w[t].thread, NULL);
        if (__atomic_load_n(&job.overflow, __ATOMIC_ACQUIRE)) return fail(p, VP8P_MEM_ERROR, "block stream overflow");
        for (t = 0; t < nthreads; t++) bad |= w[t].bad;
        if (mbx) {
            /* close the gaps: thread t's blocks follow thread t-1's, and its macroblocks' indices move with them.
               (Row order inside the stream is by thread: the stream is addressed through sparse_first, row by row.) */
            for (t = 0; t < nthreads; t++) {
                const size_t db = w[t].out.first_block - nb;
                if (db) {
                    memmove(blocks + nb * 16, blocks + w[t].out.first_block * 16, w[t].out.nb * 32);
                    for (r = 0; r < p->mb_rows; r++)
                        if (row_owner(p, r, nthreads) == t) {
                            vp8ir_mbx *o = mbx + (size_t)r * p->mb_cols;
                            int c;
                            for (c = 0; c < p->mb_cols; c++) o[c].d.sparse_first -= (uint32_t)db;
                        }
                }
                nb += w[t].out.nb;
            }
        }
    }

    if (is_key && !bad) p->have_key_frame = 1;
    if (!p->have_key_frame)
        return fail(p, VP8P_CORRUPT_FRAME, "A stream must start with a complete key frame");
    if (p->restore_probs) {
        p->fc = p->saved_fc;
        p->independent_partitions = p->prev_independent;
    }
    if (p->ec_enabled && p->prev_mi) {
        /* onyxd_if.c:622-640: this frame's mode info is what the next frame's concealment looks back on; the array it gets
           to fill starts out with this frame's segment ids (a segment map that is not updated persists) */
        mbinfo *t = p->prev_mi, *ta = p->prev_alloc;
        p->prev_mi = p->mi; p->prev_alloc = p->mi_alloc;
        p->mi = t; p->mi_alloc = ta;
        for (r = 0; r < p->mb_rows; r++) {
            int c;
            for (c = 0; c < p->mb_cols; c++) p->mi[r * p->mi_stride + c].segment_id = p->prev_mi[r * p->mi_stride + c].segment_id;
        }
    }
    if (corrupt) *corrupt = bad;
    if (nblocks) *nblocks = nb;
    p->frame_open = 0;
    return VP8P_OK;
}

/* ------------------------------------------------------------------------------------------
 * reference-buffer bookkeeping
 * ---------------------------------------------------------------------------------------- */
void vp8_refs_init(vp8_refs *r)                 /* decoder creation: nothing allocated, all counts 0 */
{
    memset(r, 0, sizeof *r);
}

void vp8_refs_on_alloc(vp8_refs *r)             /* vp8_alloc_frame_buffers, alloccommon.c:87-95 */
{
    r->new_idx = 0; r->lst_idx = 1; r->gld_idx = 2; r->alt_idx = 3;
    r->ref_cnt[0] = r->ref_cnt[1] = r->ref_cnt[2] = r->ref_cnt[3] = 1;
}

void vp8_refs_release_new(vp8_refs *r)          /* error path of vp8dx_receive_compressed_data */
{
    if (r->ref_cnt[r->new_idx] > 0) r->ref_cnt[r->new_idx]--;
}

int vp8_refs_get_free(vp8_refs *r)
{
    int i;
    for (i = 0; i < 4; i++)
        if (r->ref_cnt[i] == 0) break;
    if (i == 4) return -1;
    r->ref_cnt[i] = 1;
    r->new_idx = i;
    return i;
}

static void retarget(vp8_refs *r, int *idx, int to)
{
    if (r->ref_cnt[*idx] > 0) r->ref_cnt[*idx]--;
    *idx = to;
    r->ref_cnt[to]++;
}

/* vp8dx_set_reference (onyxd_if.c:192-230): point one reference (1 last, 2 golden, 4 alt-ref) at a free buffer,
 * leaving whatever else still shares the old one untouched.  Returns the buffer to fill, or -1. */
int vp8_refs_retarget_free(vp8_refs *r, int which)
{
    int i, *idx = which == 1 ? &r->lst_idx : which == 2 ? &r->gld_idx : which == 4 ? &r->alt_idx : NULL;
    if (!idx) return -1;
    for (i = 0; i < 4; i++)
        if (r->ref_cnt[i] == 0) break;
    if (i == 4) return -1;
    retarget(r, idx, i);
    return i;
}

int vp8_refs_swap(vp8_refs *r, const vp8ir_frame_hdr *h)
{
    int err = 0;
    if (h->copy_buffer_to_arf) {
        int from = 0;
        if (h->copy_buffer_to_arf == 1) from = r->lst_idx;
        else if (h->copy_buffer_to_arf == 2) from = r->gld_idx;
        else err = -1;
        retarget(r, &r->alt_idx, from);
    }
    if (h->copy_buffer_to_gf) {
        int from = 0;
        if (h->copy_buffer_to_gf == 1) from = r->lst_idx;
        else if (h->copy_buffer_to_gf == 2) from = r->alt_idx;
        else err = -1;
        retarget(r, &r->gld_idx, from);
    }
    if (h->refresh_golden) retarget(r, &r->gld_idx, r->new_idx);
    if (h->refresh_alt) retarget(r, &r->alt_idx, r->new_idx);
    if (h->refresh_last) {
        retarget(r, &r->lst_idx, r->new_idx);
        r->show_idx = r->lst_idx;
    } else
        r->show_idx = r->new_idx;
    r->ref_cnt[r->new_idx]--;
    return err;
}
