// vp8hip_postproc, vp8hip_mfqe (include/vp8hip.h): the host entry points of the output-side filters; the kernels are in
// vp8_postproc.hip.
#include "vp8hip_ctx.hip.h"

// vp8_postproc.hip
void vp8pp_down_and_across(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit);
void vp8pp_mb_across(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit);
void vp8pp_mb_down(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit, const short *rv);
void vp8pp_mfqe(hipStream_t st, const uint8_t *show, const uint8_t *prev, uint8_t *out, const DevGeom &g, const uint8_t *cls,
                int qcurr, int qprev);
void vp8pp_add_noise(hipStream_t st, uint8_t *plane, int stride, int rows, int cols, int clamp, const signed char *noise,
                     const uint8_t *row_offset);

// Output-side post-processing of one frame buffer into another (vp8_post_proc_frame, vp8/common/postproc.c:903-1000, minus
// the policy: the caller has turned the frame's quantiser into thresholds and drawn the random phases).
extern "C" int vp8hip_postproc(vp8hip_ctx *c, int src_fb, int dst_fb, int tmp_fb, const vp8hip_pp *pp)
{
    const int nfb = c ? (int)c->fb.size() : 0;
    if (!c || !pp || src_fb < 0 || src_fb >= nfb || dst_fb < 0 || dst_fb >= nfb)
        return fail(c, -2, "vp8hip_postproc: bad arguments");
    const bool demacro = pp->flags & VP8HIP_PP_DEMACROBLOCK, deblock = demacro || (pp->flags & VP8HIP_PP_DEBLOCK);
    if (dst_fb == src_fb && deblock)                     // (in place: the noise alone, on a picture vp8hip_mfqe left in dst_fb)
        return fail(c, -2, "vp8hip_postproc: the deblocking filters cannot run in place");
    const bool noise = pp->flags & VP8HIP_PP_ADDNOISE;
    if (demacro && (tmp_fb < 0 || tmp_fb >= nfb || tmp_fb == src_fb || tmp_fb == dst_fb || !pp->rv || pp->rv_offset < 0 || pp->rv_offset > 63))
        return fail(c, -2, "vp8hip_postproc: demacroblocking needs a third frame buffer and the dither table");
    const vp8ir_geom &g = c->geom;
    // the noise row of a line starts up to 255 entries into the 3072-entry table (the reference indexes past its end for
    // wider frames, postproc.c:499-510: no defined answer to reproduce)
    if (noise && (!pp->noise_rows || g.aligned_w + 255 > 3072 || g.aligned_h > 16384))
        return fail(c, -2, "vp8hip_postproc: noise needs the row phases and a frame at most 2816 wide");
    HIPCHK(c, hipSetDevice(c->device));
    {   // the filters read and write the raster form
        if (vp8hip_need_raster(c, src_fb, 1)) return -1;
        c->fb_state[(size_t)dst_fb] = FB_RASTER;
        if (demacro) c->fb_state[(size_t)tmp_fb] = FB_RASTER;
    }
    if (c->d2h_count) { HIPCHK(c, hipEventSynchronize(c->ev_d2h_done)); c->d2h_count = 0; }   // a batch download may be reading dst
    if (!c->d_pp || !c->h_pp || !c->ev_pp) {
        // (each piece on its own: a failure half way leaves what exists for the next call, never a null event to wait on)
        if (!c->d_pp) {
            HIPCHK(c, hipMalloc((void **)&c->d_pp, 1024 + 3072 + 16384));
            HIPCHK(c, hipMemsetAsync(c->d_pp, 0, 1024 + 3072 + 16384, c->stream));     // a noise table never sent is all zeros, as the reference's
        }
        // the caller's tables go through a pinned copy of our own, so that they may be reused the moment the call returns
        if (!c->h_pp) HIPCHK(c, hipHostMalloc((void **)&c->h_pp, 1024 + 3072 + 16384, hipHostMallocDefault));
        if (!c->ev_pp) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pp, hipEventDisableTiming));
    } else
        HIPCHK(c, hipEventSynchronize(c->ev_pp));       // the previous call's copies have left the pinned staging
    const short *d_rv = (const short *)c->d_pp;
    signed char *d_noise = (signed char *)c->d_pp + 1024;
    uint8_t *d_rows = (uint8_t *)c->d_pp + 1024 + 3072;
    if (vp8hip_raster_pool(c)) return -1;
    uint8_t *src = c->fb[src_fb], *dst = c->fb[dst_fb];
    const struct { int off, stride, rows, cols; } pl[3] = { { g.y_off, g.y_stride, g.aligned_h, g.aligned_w },
                                                            { g.u_off, g.uv_stride, g.aligned_h / 2, g.aligned_w / 2 },
                                                            { g.v_off, g.uv_stride, g.aligned_h / 2, g.aligned_w / 2 } };
    if (deblock) {
        for (int k = 0; k < 3; k++)
            vp8pp_down_and_across(c->stream, src + pl[k].off, dst + pl[k].off, pl[k].stride, pl[k].rows, pl[k].cols, pp->flimit);
        if (demacro) {       // luma only (vp8_deblock_and_de_macro_block, postproc.c:328-346)
            uint8_t *tmp = c->fb[tmp_fb];
            if (!c->pp_rv_loaded) {
                memcpy(c->h_pp, pp->rv, 440 * sizeof(short));
                HIPCHK(c, hipMemcpyAsync(c->d_pp, c->h_pp, 440 * sizeof(short), hipMemcpyHostToDevice, c->stream));
                c->pp_rv_loaded = true;
            }
            vp8pp_mb_across(c->stream, dst + pl[0].off, tmp + pl[0].off, pl[0].stride, pl[0].rows, pl[0].cols, pp->mb_flimit);
            vp8pp_mb_down(c->stream, tmp + pl[0].off, dst + pl[0].off, pl[0].stride, pl[0].rows, pl[0].cols, pp->mb_flimit,
                          d_rv + pp->rv_offset);
        }
    } else if (dst != src)      // vp8_yv12_copy_frame_ptr (postproc.c:982)
        HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)g.frame_size, hipMemcpyDeviceToDevice, c->stream));
    if (noise) {
        if (pp->noise) {
            memcpy(c->h_pp + 1024, pp->noise, 3072);
            HIPCHK(c, hipMemcpyAsync(d_noise, c->h_pp + 1024, 3072, hipMemcpyHostToDevice, c->stream));
        }
        memcpy(c->h_pp + 1024 + 3072, pp->noise_rows, (size_t)pl[0].rows);
        HIPCHK(c, hipMemcpyAsync(d_rows, c->h_pp + 1024 + 3072, (size_t)pl[0].rows, hipMemcpyHostToDevice, c->stream));
        vp8pp_add_noise(c->stream, dst + pl[0].off, pl[0].stride, pl[0].rows, pl[0].cols, pp->noise_clamp, d_noise, d_rows);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_pp, c->stream));
    return 0;
}

extern "C" int vp8hip_mfqe(vp8hip_ctx *c, int show_fb, int prev_fb, int dst_fb, const uint8_t *mb_class, int qcurr, int qprev)
{
    const int nfb = c ? (int)c->fb.size() : 0;
    if (!c || !mb_class || show_fb < 0 || show_fb >= nfb || prev_fb < 0 || prev_fb >= nfb || dst_fb < 0 || dst_fb >= nfb ||
        show_fb == prev_fb || show_fb == dst_fb || qcurr < 0 || qcurr > 127 || qprev < 0 || qprev > qcurr)
        return fail(c, -2, "vp8hip_mfqe: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    {
        const int need[2] = { show_fb, prev_fb };
        if (vp8hip_raster_pool(c) || vp8hip_need_raster_list(c, need, 2)) return -1;
        c->fb_state[(size_t)dst_fb] = FB_RASTER;
    }
    if (c->d2h_count) { HIPCHK(c, hipEventSynchronize(c->ev_d2h_done)); c->d2h_count = 0; }   // a batch download may be reading dst
    const int nmb = c->dg.mb_cols * c->dg.mb_rows;
    if (nmb > c->mfqe_cap) {
        if (c->ev_mfqe) HIPCHK(c, hipEventSynchronize(c->ev_mfqe));
        if (c->d_mfqe) (void)hipFree(c->d_mfqe);
        if (c->h_mfqe) (void)hipHostFree(c->h_mfqe);
        c->d_mfqe = c->h_mfqe = nullptr; c->mfqe_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_mfqe, (size_t)nmb));
        HIPCHK(c, hipHostMalloc((void **)&c->h_mfqe, (size_t)nmb, hipHostMallocDefault));
        c->mfqe_cap = nmb;
    }
    if (!c->ev_mfqe) HIPCHK(c, hipEventCreateWithFlags(&c->ev_mfqe, hipEventDisableTiming));
    else HIPCHK(c, hipEventSynchronize(c->ev_mfqe));    // the previous call's copy has left the pinned staging
    memcpy(c->h_mfqe, mb_class, (size_t)nmb);
    HIPCHK(c, hipMemcpyAsync(c->d_mfqe, c->h_mfqe, (size_t)nmb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_mfqe, c->stream));
    vp8pp_mfqe(c->stream, c->fb[show_fb], c->fb[prev_fb], c->fb[dst_fb], c->dg, c->d_mfqe, qcurr, qprev);
    HIPCHK(c, hipGetLastError());
    return 0;
}
