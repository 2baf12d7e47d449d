/* The VP8 decoder algorithm interface `vpx_codec_vp8_dx`, backed by the MI355X HIP pixel path.
 *
 * Same entry points, argument meaning and error behaviour as the reference's
 *   vp8/vp8_dx_iface.c        vp8_init :188, vp8_destroy :221, vp8_peek_si :245, vp8_get_si :287,
 *                             vp8_decode :350, vp8_get_frame :485, controls :611-770, iface :776
 *   vp8/decoder/onyxd_if.c    vp8dx_receive_compressed_data :318 (frame lifecycle, buffer swap)
 * but the work is split differently: the host only entropy-decodes (vp8_parser) into pinned IR
 * staging; prediction, residual, loop filter and border extension run on the GPU through the
 * RTCD table (include/vp8_rtcd.h -> include/vp8hip.h); the reference-frame pool lives in HBM and
 * only the frame to show is copied back.  No CPU pixel fallback exists: if the GPU is
 * unavailable, decode() returns VPX_CODEC_ERROR with a detail string.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vp8_parser.h"
#include "vp8_rtcd.h"
#include "vpx/vp8dx.h"
#include "vpx_codec_internal.h"

#include "vp8_postproc_host.h"

#define VP8_CAP_POSTPROC VPX_CODEC_CAP_POSTPROC      /* vp8_dx_iface.c:25 with CONFIG_POSTPROC */
/* frame buffers: the reference's four (NUM_YV12_BUFFERS) + post_proc_buffer + a work buffer of the demacroblocking filter */
#define FB_DECODE 4
#define FB_POST   4
#define FB_PPTMP  5
#define FB_PPINT  6         /* post_proc_buffer_int: what VP8_MFQE hands to the deblocking filters */

struct vpx_codec_alg_priv {
    vpx_codec_priv_t        base;
    vpx_codec_dec_cfg_t     cfg;
    vpx_codec_stream_info_t si;
    int                     decoder_init;
    vp8_parser             *parser;
    vp8hip_ctx             *hip;
    vp8_refs                refs;
    vp8ir_geom              geom;
    int                     width, height;
    uint8_t                *host_frame;       /* frame_to_show copied back, vp8ir_geom layout */
    vpx_image_t             img;
    int                     img_avail;
    int                     fb_corrupted[4];
    int                     show_corrupted;
    int                     ref_updates, ref_used;
    /* VPX_CODEC_USE_INPUT_FRAGMENTS: the pieces of the frame being collected (borrowed until the flush call, like the
       reference's pbi->fragments, onyxd_if.c:336-366) */
    const uint8_t          *frag[9];
    size_t                  frag_sz[9];
    int                     num_frags;
    /* VPX_CODEC_USE_POSTPROC (vp8_dx_iface.c:65-66,421-431,446-466) */
    int                     postproc_cfg_set;
    vp8_postproc_cfg_t      postproc_cfg;
    vp8_pp_state           *pp;
    uint8_t                *mb_class;        /* VP8_MFQE: a byte per macroblock (vp8_pp_mfqe_classes) */
    size_t                  mb_class_cap;
    char                    detail[160];
    double                  t_parse, t_launch, t_down;   /* VP8HIP_TRACE: seconds per phase of vp8_decode */
    long                    t_frames;
};

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static vpx_codec_err_t set_detail(vpx_codec_alg_priv_t *p, vpx_codec_err_t code, const char *msg)
{
    snprintf(p->detail, sizeof p->detail, "%s", msg ? msg : "");
    p->base.err_detail = msg ? p->detail : NULL;
    return code;
}

static vpx_codec_err_t vp8_init(vpx_codec_ctx_t *ctx, void *mr_cfg)
{
    (void)mr_cfg;
    if (!ctx->priv) {
        vpx_codec_alg_priv_t *p = (vpx_codec_alg_priv_t *)calloc(1, sizeof *p);
        if (!p) return VPX_CODEC_MEM_ERROR;
        ctx->priv = &p->base;
        p->base.sz = sizeof *p;
        p->base.iface = ctx->iface;
        p->base.alg_priv = p;
        p->base.init_flags = ctx->init_flags;
        p->si.sz = sizeof p->si;
        if (ctx->config.dec) {           /* keep our own copy: the caller's struct may go away */
            p->cfg = *ctx->config.dec;
            ctx->config.dec = &p->cfg;
        }
    }
    return VPX_CODEC_OK;
}

static vpx_codec_err_t vp8_destroy(vpx_codec_alg_priv_t *p)
{
    if (getenv("VP8HIP_TRACE") && p->t_frames)
        fprintf(stderr, "[vp8_dx] %ld frames: parse %.3f ms, upload+launch %.3f ms, wait+download %.3f ms per frame\n", p->t_frames,
                p->t_parse / p->t_frames * 1e3, p->t_launch / p->t_frames * 1e3, p->t_down / p->t_frames * 1e3);
    if (p->hip) { vp8hip_host_free(p->hip, p->host_frame); vp8hip_destroy(p->hip); }
    vp8_parser_destroy(p->parser);
    free(p->pp);
    free(p->mb_class);
    free(p);
    return VPX_CODEC_OK;
}

static vpx_codec_err_t vp8_peek_si(const uint8_t *data, unsigned int data_sz, vpx_codec_stream_info_t *si)
{
    int is_kf = 0, w = 0, h = 0, rc;
    if (data + data_sz <= data) return VPX_CODEC_INVALID_PARAM;
    rc = vp8_parser_peek(data, data_sz, &is_kf, &w, &h);
    si->is_kf = (unsigned)is_kf;
    if (is_kf) { si->w = (unsigned)w; si->h = (unsigned)h; }
    return (vpx_codec_err_t)rc;
}

static vpx_codec_err_t vp8_get_si(vpx_codec_alg_priv_t *p, vpx_codec_stream_info_t *si)
{
    unsigned int sz = si->sz >= sizeof p->si ? sizeof p->si : sizeof(vpx_codec_stream_info_t);
    memcpy(si, &p->si, sz);
    si->sz = sz;
    return VPX_CODEC_OK;
}

static vpx_codec_err_t gpu_error(vpx_codec_alg_priv_t *p, const char *what)
{
    char msg[160];
    snprintf(msg, sizeof msg, "%s: %s", what, vp8hip_last_error(p->hip));
    return set_detail(p, VPX_CODEC_ERROR, msg);
}

/* yuvconfig2image (vp8_dx_iface.c:319-347): the image aliases the decoder's own frame memory */
static void publish_image(vpx_codec_alg_priv_t *p, void *user_priv)
{
    vpx_image_t *img = &p->img;
    const vp8ir_geom *g = &p->geom;
    memset(img, 0, sizeof *img);
    img->fmt = VPX_IMG_FMT_I420;
    img->w = (unsigned)g->y_stride;
    img->h = (unsigned)((g->aligned_h + 2 * VP8IR_BORDER + 15) & ~15);
    img->d_w = (unsigned)p->width;
    img->d_h = (unsigned)p->height;
    img->x_chroma_shift = img->y_chroma_shift = 1;
    img->planes[VPX_PLANE_Y] = p->host_frame + g->y_off;
    img->planes[VPX_PLANE_U] = p->host_frame + g->u_off;
    img->planes[VPX_PLANE_V] = p->host_frame + g->v_off;
    img->planes[VPX_PLANE_ALPHA] = NULL;
    img->stride[VPX_PLANE_Y] = g->y_stride;
    img->stride[VPX_PLANE_U] = img->stride[VPX_PLANE_V] = g->uv_stride;
    img->stride[VPX_PLANE_ALPHA] = g->y_stride;
    img->bps = 12;
    img->user_priv = user_priv;
    img->img_data = p->host_frame;
}

static vpx_codec_err_t vp8_decode(vpx_codec_alg_priv_t *p, const uint8_t *data, unsigned int data_sz, void *user_priv,
                                  long deadline)
{
    vp8ir_frame_hdr hdr, *h_hdr;
    vp8ir_mbx *h_mbx;
    int16_t *h_blocks;
    size_t cap_blocks, nblocks = 0;
    vp8ir_mv *h_mvs;
    vp8hip_job job;
    int rc, corrupt = 0, i, nmb, nfrags;
    double t0, t1, t2;
    (void)deadline;

    p->img_avail = 0;
    p->base.err_detail = NULL;

    if (!p->si.h) {                                   /* first call: needs a key frame (vp8_dx_iface.c:364) */
        vpx_codec_err_t res = vp8_peek_si(data, data_sz, &p->si);
        if (res) return res;
    }
    if (!p->decoder_init) {                           /* vp8dx_create_decompressor equivalent */
        int device = -1;
        const char *env = getenv("VP8HIP_DEVICE");
        if (env && *env) device = atoi(env);
        vpx_rtcd();
        p->parser = vp8_parser_create();
        if (!p->parser) return VPX_CODEC_MEM_ERROR;
        vp8_parser_set_threads(p->parser, (int)p->cfg.threads);   /* oxcf.max_threads = ctx->cfg.threads, vp8_dx_iface.c:413 */
        /* oxcf.error_concealment (vp8_dx_iface.c:414-415) */
        vp8_parser_set_error_concealment(p->parser, (p->base.init_flags & VPX_CODEC_USE_ERROR_CONCEALMENT) != 0);
        vp8_refs_init(&p->refs);
        p->decoder_init = 1;
        if (vp8hip_create(device, &p->hip)) {
            p->hip = NULL;
            return set_detail(p, VPX_CODEC_ERROR, vp8hip_last_error(NULL));
        }
    }
    if (!p->hip) return set_detail(p, VPX_CODEC_ERROR, "HIP pixel path unavailable (no CPU fallback)");

    if ((p->base.init_flags & VPX_CODEC_USE_INPUT_FRAGMENTS) && !(data == NULL && data_sz == 0)) {
        /* a piece of a frame: remember it and wait for the rest (onyxd_if.c:343-361) */
        if (p->num_frags >= 9) {
            p->num_frags = 0;
            return set_detail(p, VPX_CODEC_UNSUP_BITSTREAM, "Too many fragments");
        }
        p->frag[p->num_frags] = data;
        p->frag_sz[p->num_frags] = data_sz;
        p->num_frags++;
        return VPX_CODEC_OK;
    }
    if (!(p->base.init_flags & VPX_CODEC_USE_INPUT_FRAGMENTS)) {
        p->frag[0] = data;
        p->frag_sz[0] = data_sz;
        p->num_frags = data ? 1 : 0;
    }
    if (p->num_frags == 0 || (p->num_frags == 1 && p->frag_sz[0] == 0)) {
        if (!vp8_parser_conceals(p->parser) || !p->width) {
            /* missing frame (onyxd_if.c:375-407): mark the last reference corrupt, nothing to show */
            p->fb_corrupted[p->refs.lst_idx] = 1;
            p->num_frags = 0;
            return VPX_CODEC_OK;
        }
        /* with error concealment at work the lost frame is decoded: an inter frame out of estimated motion vectors */
        p->frag[0] = NULL;
        p->frag_sz[0] = 0;
        p->num_frags = 1;
    }
    nfrags = p->num_frags;
    p->num_frags = 0;                                 /* whatever happens below, the next call starts a new frame */

    if (vp8_refs_get_free(&p->refs) < 0) return set_detail(p, VPX_CODEC_ERROR, "no free frame buffer");
    t0 = now_s();
    rc = vp8_parser_begin_frame_fragments(p->parser, p->frag, p->frag_sz, nfrags, &hdr);
    if (rc) {
        vp8_refs_release_new(&p->refs);
        return set_detail(p, (vpx_codec_err_t)rc, vp8_parser_error(p->parser));
    }
    if (hdr.width != p->width || hdr.height != p->height) {       /* vp8_alloc_frame_buffers */
        if (vp8hip_configure(p->hip, hdr.width, hdr.height,
                             (p->base.init_flags & VPX_CODEC_USE_POSTPROC) ? FB_DECODE + 3 : FB_DECODE, 1)) {
            vp8_refs_release_new(&p->refs);
            p->width = p->height = 0;
            return gpu_error(p, "vp8hip_configure");
        }
        vp8hip_geometry(p->hip, &p->geom);
        vp8hip_host_free(p->hip, p->host_frame);
        p->host_frame = (uint8_t *)vp8hip_host_alloc(p->hip, (size_t)p->geom.frame_size);
        if (!p->host_frame) {
            vp8_refs_release_new(&p->refs);
            p->width = p->height = 0;                 /* the next frame allocates again */
            return VPX_CODEC_MEM_ERROR;
        }
        p->width = hdr.width;
        p->height = hdr.height;
        p->si.w = hdr.width;
        p->si.h = hdr.height;
        vp8_refs_on_alloc(&p->refs);
        memset(p->fb_corrupted, 0, sizeof p->fb_corrupted);
    }
    /* the feeder writes the device form of include/vp8_ir.h straight into the slot's pinned staging: what goes up is what the
       kernels read (0.4 of the bytes the dense arrays would cost PCIe), with one copy and nothing in between */
    if (vp8hip_ir_map_compact(p->hip, 0, &h_hdr, &h_mbx, &h_blocks, &cap_blocks, &h_mvs)) {
        vp8_refs_release_new(&p->refs);
        return gpu_error(p, "vp8hip_ir_map");
    }
    rc = vp8_parser_decode_mbs_compact(p->parser, h_mbx, h_blocks, cap_blocks, &nblocks, h_mvs, &corrupt);
    if (rc) {
        vp8_refs_release_new(&p->refs);
        return set_detail(p, (vpx_codec_err_t)rc, vp8_parser_error(p->parser));
    }
    vp8_parser_frame_hdr(p->parser, &hdr);          /* (a concealed key frame reads references: frame_type 1, lf_key_frame 1) */
    *h_hdr = hdr;

    /* which references does this frame read (vp8dx_references_buffer, onyxd_if.c:711-760) */
    nmb = hdr.mb_cols * hdr.mb_rows;
    p->ref_used = 0;
    if (hdr.frame_type != 0)
        for (i = 0; i < nmb; i++) {
            int rf = h_mbx[i].d.ref_frame;
            if (rf == VP8IR_LAST_FRAME) p->ref_used |= VP8_LAST_FRAME;
            else if (rf == VP8IR_GOLDEN_FRAME) p->ref_used |= VP8_GOLD_FRAME;
            else if (rf == VP8IR_ALTREF_FRAME) p->ref_used |= VP8_ALTR_FRAME;
        }
    p->ref_updates = (hdr.refresh_last ? VP8_LAST_FRAME : 0) | (hdr.refresh_golden ? VP8_GOLD_FRAME : 0)
                   | (hdr.refresh_alt ? VP8_ALTR_FRAME : 0);
    if (p->ref_used & VP8_LAST_FRAME) corrupt |= p->fb_corrupted[p->refs.lst_idx];
    if (p->ref_used & VP8_GOLD_FRAME) corrupt |= p->fb_corrupted[p->refs.gld_idx];
    if (p->ref_used & VP8_ALTR_FRAME) corrupt |= p->fb_corrupted[p->refs.alt_idx];
    p->fb_corrupted[p->refs.new_idx] = corrupt;

    /* (allocated before anything is decoded: a failure here must not drop a frame that has been decoded and swapped in) */
    if ((p->base.init_flags & VPX_CODEC_USE_POSTPROC) && !p->pp && !(p->pp = (vp8_pp_state *)calloc(1, sizeof *p->pp))) {
        vp8_refs_release_new(&p->refs);
        return VPX_CODEC_MEM_ERROR;
    }
    if ((p->base.init_flags & VPX_CODEC_USE_POSTPROC) && (size_t)hdr.mb_cols * hdr.mb_rows > p->mb_class_cap) {
        const size_t n = (size_t)hdr.mb_cols * hdr.mb_rows;
        uint8_t *m = (uint8_t *)realloc(p->mb_class, n);
        if (!m) { vp8_refs_release_new(&p->refs); return VPX_CODEC_MEM_ERROR; }
        p->mb_class = m;
        p->mb_class_cap = n;
    }
    t1 = now_s();
    if (vp8hip_ir_upload_compact(p->hip, 0, nblocks)) { vp8_refs_release_new(&p->refs); return gpu_error(p, "vp8hip_ir_upload_compact"); }
    job.ir_slot = 0;
    job.dst_fb = p->refs.new_idx;
    job.ref_fb[0] = -1;
    job.ref_fb[VP8IR_LAST_FRAME] = p->refs.lst_idx;
    job.ref_fb[VP8IR_GOLDEN_FRAME] = p->refs.gld_idx;
    job.ref_fb[VP8IR_ALTREF_FRAME] = p->refs.alt_idx;
    if (vp8_decode_frame_pixels(p->hip, &job, 1)) { vp8_refs_release_new(&p->refs); return gpu_error(p, "vp8hip_decode"); }

    /* (swap_frame_buffers, onyxd_if.c:261-316, gives the new buffer's reference back even when it reports bad copy flags) */
    if (vp8_refs_swap(&p->refs, &hdr)) return set_detail(p, VPX_CODEC_ERROR, "invalid buffer copy flags");
    p->show_corrupted = p->fb_corrupted[p->refs.show_idx];

    t2 = now_s();
    if (hdr.show_frame) {
        int show_fb = p->refs.show_idx;
        if (p->base.init_flags & VPX_CODEC_USE_POSTPROC) {
            /* vp8dx_get_raw_frame -> vp8_post_proc_frame (onyxd_if.c:723, postproc.c:903): the shown frame goes through the
               output filters into post_proc_buffer, and that is the image the application gets */
            vp8hip_pp pp;
            if (!p->postproc_cfg_set) {                       /* the reference's default (vp8_dx_iface.c:421-431) */
                p->postproc_cfg.post_proc_flag = VP8_DEBLOCK | VP8_DEMACROBLOCK | VP8_MFQE;
                p->postproc_cfg.deblocking_level = 4;
                p->postproc_cfg.noise_level = 0;
                p->postproc_cfg_set = 1;
            }
            {
                int qprev = 0;
                const int mfqe = vp8_pp_mfqe_step(p->pp, &p->postproc_cfg, hdr.base_qindex, &qprev);
                const int filters = vp8_pp_prepare(p->pp, &p->postproc_cfg, hdr.filter_level, p->geom.aligned_h, &pp);
                if (mfqe) {
                    /* postproc.c:948-969: the picture shown before, still in post_proc_buffer, is kept or blended in where the
                       new frame (coarser by 10 quantiser steps or more) differs little from it; the deblocking filters then
                       run on the result.  (Sizes that are not multiples of 16 included: the reference dies on those when both
                       are asked for -- its intermediate buffer is never allocated, :929-941 -- so that combination is the one
                       piece here no listing pins.) */
                    const int deblocking = filters & (VP8HIP_PP_DEBLOCK | VP8HIP_PP_DEMACROBLOCK);
                    vp8_pp_mfqe_classes(&hdr, h_mbx, sizeof *h_mbx, h_mvs, p->mb_class);
                    if (vp8hip_mfqe(p->hip, show_fb, FB_POST, deblocking ? FB_PPINT : FB_POST, p->mb_class, hdr.base_qindex, qprev))
                        return gpu_error(p, "vp8hip_mfqe");
                    if (deblocking) {
                        if (vp8hip_postproc(p->hip, FB_PPINT, FB_POST, FB_PPTMP, &pp)) return gpu_error(p, "vp8hip_postproc");
                    } else if (filters && vp8hip_postproc(p->hip, FB_POST, FB_POST, FB_PPTMP, &pp))
                        return gpu_error(p, "vp8hip_postproc");
                    show_fb = FB_POST;
                } else if (filters || (p->postproc_cfg.post_proc_flag & VP8_MFQE)) {
                    /* (with VP8_MFQE the buffer holds every shown frame, filtered or not: :982-986) */
                    if (vp8hip_postproc(p->hip, show_fb, FB_POST, FB_PPTMP, &pp)) return gpu_error(p, "vp8hip_postproc");
                    show_fb = FB_POST;
                }
            }
        }
        /* the whole frame buffer, borders included, in one linear copy into the pinned mirror (same vp8ir_geom layout) */
        if (vp8hip_frame_download(p->hip, show_fb, 1, p->host_frame, NULL, NULL, 0, 0))
            return gpu_error(p, "vp8hip_frame_download");
        publish_image(p, user_priv);
        p->img_avail = 1;
    } else if (vp8hip_sync(p->hip))
        return gpu_error(p, "vp8hip_sync");
    p->t_parse += t1 - t0; p->t_launch += t2 - t1; p->t_down += now_s() - t2; p->t_frames++;
    return VPX_CODEC_OK;
}

static vpx_image_t *vp8_get_frame(vpx_codec_alg_priv_t *p, vpx_codec_iter_t *iter)
{
    if (p->img_avail && !*iter) {          /* flip-flop iterator (vp8_dx_iface.c:485-503) */
        *iter = &p->img;
        return &p->img;
    }
    return NULL;
}

static vpx_codec_err_t ctl_get_int(vpx_codec_alg_priv_t *p, int ctrl_id, va_list ap)
{
    int *out = va_arg(ap, int *);
    if (!out) return VPX_CODEC_INVALID_PARAM;
    if (!p->decoder_init) return VPX_CODEC_ERROR;
    if (ctrl_id == VP8D_GET_FRAME_CORRUPTED) *out = p->show_corrupted;
    else if (ctrl_id == VP8D_GET_LAST_REF_UPDATES) *out = p->ref_updates;
    else *out = p->ref_used;
    return VPX_CODEC_OK;
}

static vpx_codec_err_t ctl_incapable(vpx_codec_alg_priv_t *p, int ctrl_id, va_list ap)
{
    (void)ctrl_id; (void)ap;
    /* the debug overlays of CONFIG_POSTPROC_VISUALIZER (vp8_dx_iface.c:674-697 answers the same way without it) */
    return set_detail(p, VPX_CODEC_INCAPABLE, "control not implemented by the HIP decoder");
}

/* vp8_set_postproc (vp8_dx_iface.c:653-672) */
static vpx_codec_err_t ctl_set_postproc(vpx_codec_alg_priv_t *p, int ctrl_id, va_list ap)
{
    vp8_postproc_cfg_t *data = va_arg(ap, vp8_postproc_cfg_t *);
    (void)ctrl_id;
    if (!data) return VPX_CODEC_INVALID_PARAM;
    p->postproc_cfg_set = 1;
    p->postproc_cfg = *data;
    return VPX_CODEC_OK;
}

/* VP8_COPY_REFERENCE / VP8_SET_REFERENCE (vp8_dx_iface.c:611-651 -> vp8dx_get_reference / vp8dx_set_reference,
 * onyxd_if.c:161-230).  The reference frames live in HBM; the image is staged through a host copy of one frame
 * buffer.  As in the reference the image must have the frame buffers' (16-aligned) dimensions, and a set
 * reference gets its borders extended (vp8_yv12_copy_frame = copy + extend). */
static void extend_host_plane(uint8_t *p, int stride, int w, int h, int border)
{
    int r, i;
    for (r = 0; r < h; r++) {
        memset(p + r * stride - border, p[r * stride], (size_t)border);
        memset(p + r * stride + w, p[r * stride + w - 1], (size_t)border);
    }
    for (i = 1; i <= border; i++) {
        memcpy(p - border - i * stride, p - border, (size_t)w + 2 * border);
        memcpy(p - border + (h - 1 + i) * stride, p - border + (h - 1) * stride, (size_t)w + 2 * border);
    }
}

static vpx_codec_err_t ctl_reference(vpx_codec_alg_priv_t *p, int ctrl_id, va_list ap)
{
    vpx_ref_frame_t *rf = va_arg(ap, vpx_ref_frame_t *);
    const vp8ir_geom *g = &p->geom;
    uint8_t *buf;
    int fb, r;
    if (!rf) return VPX_CODEC_INVALID_PARAM;
    if (!p->decoder_init || !p->hip) return set_detail(p, VPX_CODEC_ERROR, "no frame has been decoded yet");
    if (rf->frame_type != VP8_LAST_FRAME && rf->frame_type != VP8_GOLD_FRAME && rf->frame_type != VP8_ALTR_FRAME)
        return set_detail(p, VPX_CODEC_ERROR, "Invalid reference frame");
    if ((int)rf->img.d_w != g->aligned_w || (int)rf->img.d_h != g->aligned_h ||
        rf->img.x_chroma_shift != 1 || rf->img.y_chroma_shift != 1)
        return set_detail(p, VPX_CODEC_ERROR, "Incorrect buffer dimensions");
    buf = (uint8_t *)malloc((size_t)g->frame_size);
    if (!buf) return VPX_CODEC_MEM_ERROR;
    if (ctrl_id == VP8_COPY_REFERENCE) {
        fb = rf->frame_type == VP8_LAST_FRAME ? p->refs.lst_idx : rf->frame_type == VP8_GOLD_FRAME ? p->refs.gld_idx
                                                                                                   : p->refs.alt_idx;
        if (vp8hip_frame_download(p->hip, fb, 1, buf, NULL, NULL, 0, 0)) { free(buf); return gpu_error(p, "vp8hip_frame_download"); }
        for (r = 0; r < g->aligned_h; r++)
            memcpy(rf->img.planes[VPX_PLANE_Y] + (size_t)r * rf->img.stride[VPX_PLANE_Y],
                   buf + g->y_off + (size_t)r * g->y_stride, (size_t)g->aligned_w);
        for (r = 0; r < g->aligned_h / 2; r++) {
            memcpy(rf->img.planes[VPX_PLANE_U] + (size_t)r * rf->img.stride[VPX_PLANE_U],
                   buf + g->u_off + (size_t)r * g->uv_stride, (size_t)g->aligned_w / 2);
            memcpy(rf->img.planes[VPX_PLANE_V] + (size_t)r * rf->img.stride[VPX_PLANE_V],
                   buf + g->v_off + (size_t)r * g->uv_stride, (size_t)g->aligned_w / 2);
        }
    } else {
        fb = vp8_refs_retarget_free(&p->refs, (int)rf->frame_type);
        if (fb < 0) { free(buf); return set_detail(p, VPX_CODEC_ERROR, "no free frame buffer"); }
        memset(buf, 0, (size_t)g->frame_size);
        for (r = 0; r < g->aligned_h; r++)
            memcpy(buf + g->y_off + (size_t)r * g->y_stride,
                   rf->img.planes[VPX_PLANE_Y] + (size_t)r * rf->img.stride[VPX_PLANE_Y], (size_t)g->aligned_w);
        for (r = 0; r < g->aligned_h / 2; r++) {
            memcpy(buf + g->u_off + (size_t)r * g->uv_stride,
                   rf->img.planes[VPX_PLANE_U] + (size_t)r * rf->img.stride[VPX_PLANE_U], (size_t)g->aligned_w / 2);
            memcpy(buf + g->v_off + (size_t)r * g->uv_stride,
                   rf->img.planes[VPX_PLANE_V] + (size_t)r * rf->img.stride[VPX_PLANE_V], (size_t)g->aligned_w / 2);
        }
        extend_host_plane(buf + g->y_off, g->y_stride, g->aligned_w, g->aligned_h, VP8IR_BORDER);
        extend_host_plane(buf + g->u_off, g->uv_stride, g->aligned_w / 2, g->aligned_h / 2, VP8IR_BORDER / 2);
        extend_host_plane(buf + g->v_off, g->uv_stride, g->aligned_w / 2, g->aligned_h / 2, VP8IR_BORDER / 2);
        if (vp8hip_frame_upload(p->hip, fb, buf)) { free(buf); return gpu_error(p, "vp8hip_frame_upload"); }
        p->fb_corrupted[fb] = 0;
    }
    free(buf);
    return VPX_CODEC_OK;
}

static vpx_codec_ctrl_fn_map_t vp8_ctf_maps[] = {
    { VP8_SET_REFERENCE, ctl_reference },
    { VP8_COPY_REFERENCE, ctl_reference },
    { VP8_SET_POSTPROC, ctl_set_postproc },
    { VP8_SET_DBG_COLOR_REF_FRAME, ctl_incapable },
    { VP8_SET_DBG_COLOR_MB_MODES, ctl_incapable },
    { VP8_SET_DBG_COLOR_B_MODES, ctl_incapable },
    { VP8_SET_DBG_DISPLAY_MV, ctl_incapable },
    { VP8D_GET_LAST_REF_UPDATES, ctl_get_int },
    { VP8D_GET_FRAME_CORRUPTED, ctl_get_int },
    { VP8D_GET_LAST_REF_USED, ctl_get_int },
    { -1, NULL },
};

const struct vpx_codec_iface vpx_codec_vp8_dx_algo = {
    "MI355X HIP VP8 Decoder (gfx950) " "v1.0.0",
    VPX_CODEC_INTERNAL_ABI_VERSION,
    VPX_CODEC_CAP_DECODER | VP8_CAP_POSTPROC | VPX_CODEC_CAP_ERROR_CONCEALMENT | VPX_CODEC_CAP_INPUT_FRAGMENTS,
    vp8_init,
    vp8_destroy,
    vp8_ctf_maps,
    NULL,               /* get_mmap: XMA not supported */
    NULL,               /* set_mmap */
    { vp8_peek_si, vp8_get_si, vp8_decode, vp8_get_frame },
    { NULL, NULL, NULL, NULL, NULL, NULL, NULL },
};

vpx_codec_iface_t *vpx_codec_vp8_dx(void) { return &vpx_codec_vp8_dx_algo; }
