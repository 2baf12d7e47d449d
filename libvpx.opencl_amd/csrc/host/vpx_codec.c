/* Generic vpx codec API entry points: context checks, vtable dispatch, error bookkeeping, vpx_img_*.
 * Behavioural reference: vpx/src/vpx_codec.c, vpx/src/vpx_decoder.c, vpx/src/vpx_image.c (the argument
 * checks and returned error codes follow those files so applications see identical behaviour). */
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "vpx_codec_internal.h"

#define SAVE_STATUS(ctx, expr) ((ctx) ? ((ctx)->err = (expr)) : (expr))

int vpx_codec_version(void) { return (1 << 16) | (0 << 8) | 0; }      /* API level of libvpx v1.0.0 */
const char *vpx_codec_version_str(void) { return "v1.0.0-mi355x-hip"; }
const char *vpx_codec_version_extra_str(void) { return "mi355x-hip"; }
const char *vpx_codec_build_config(void) { return "--target=gfx950 (HIP pixel path, C host)"; }

const char *vpx_codec_iface_name(vpx_codec_iface_t *iface) { return iface ? iface->name : "<invalid interface>"; }

const char *vpx_codec_err_to_string(vpx_codec_err_t err)
{
    switch (err) {
    case VPX_CODEC_OK: return "Success";
    case VPX_CODEC_ERROR: return "Unspecified internal error";
    case VPX_CODEC_MEM_ERROR: return "Memory allocation error";
    case VPX_CODEC_ABI_MISMATCH: return "ABI version mismatch";
    case VPX_CODEC_INCAPABLE: return "Codec does not implement requested capability";
    case VPX_CODEC_UNSUP_BITSTREAM: return "Bitstream not supported by this decoder";
    case VPX_CODEC_UNSUP_FEATURE: return "Bitstream required feature not supported by this decoder";
    case VPX_CODEC_CORRUPT_FRAME: return "Corrupt frame detected";
    case VPX_CODEC_INVALID_PARAM: return "Invalid parameter";
    case VPX_CODEC_LIST_END: return "End of iterated list";
    }
    return "Unrecognized error code";
}

const char *vpx_codec_error(vpx_codec_ctx_t *ctx)
{
    return ctx ? vpx_codec_err_to_string(ctx->err) : vpx_codec_err_to_string(VPX_CODEC_INVALID_PARAM);
}

const char *vpx_codec_error_detail(vpx_codec_ctx_t *ctx)
{
    if (ctx && ctx->err) return ctx->priv ? ctx->priv->err_detail : ctx->err_detail;
    return NULL;
}

vpx_codec_err_t vpx_codec_destroy(vpx_codec_ctx_t *ctx)
{
    vpx_codec_err_t res;
    if (!ctx) res = VPX_CODEC_INVALID_PARAM;
    else if (!ctx->iface || !ctx->priv) res = VPX_CODEC_ERROR;
    else {
        if (ctx->priv->alg_priv) ctx->iface->destroy(ctx->priv->alg_priv);
        ctx->iface = NULL;
        ctx->name = NULL;
        ctx->priv = NULL;
        res = VPX_CODEC_OK;
    }
    return SAVE_STATUS(ctx, res);
}

vpx_codec_caps_t vpx_codec_get_caps(vpx_codec_iface_t *iface) { return iface ? iface->caps : 0; }

vpx_codec_err_t vpx_codec_control_(vpx_codec_ctx_t *ctx, int ctrl_id, ...)
{
    vpx_codec_err_t res;
    if (!ctx || !ctrl_id) res = VPX_CODEC_INVALID_PARAM;
    else if (!ctx->iface || !ctx->priv || !ctx->iface->ctrl_maps) res = VPX_CODEC_ERROR;
    else {
        vpx_codec_ctrl_fn_map_t *e;
        res = VPX_CODEC_ERROR;
        for (e = ctx->iface->ctrl_maps; e && e->fn; e++) {
            if (!e->ctrl_id || e->ctrl_id == ctrl_id) {
                va_list ap;
                va_start(ap, ctrl_id);
                res = e->fn(ctx->priv->alg_priv, ctrl_id, ap);
                va_end(ap);
                break;
            }
        }
    }
    return SAVE_STATUS(ctx, res);
}

vpx_codec_err_t vpx_codec_get_mem_map(vpx_codec_ctx_t *ctx, vpx_codec_mmap_t *mmap, vpx_codec_iter_t *iter)
{
    vpx_codec_err_t res = VPX_CODEC_OK;
    if (!ctx || !mmap || !iter || !ctx->iface) res = VPX_CODEC_INVALID_PARAM;
    else if (!(ctx->iface->caps & VPX_CODEC_CAP_XMA)) res = VPX_CODEC_ERROR;
    else res = ctx->iface->get_mmap(ctx, mmap, iter);
    return SAVE_STATUS(ctx, res);
}

vpx_codec_err_t vpx_codec_set_mem_map(vpx_codec_ctx_t *ctx, vpx_codec_mmap_t *mmaps, unsigned int num_maps)
{
    vpx_codec_err_t res = VPX_CODEC_MEM_ERROR;
    if (!ctx || !mmaps || !ctx->iface) res = VPX_CODEC_INVALID_PARAM;
    else if (!(ctx->iface->caps & VPX_CODEC_CAP_XMA)) res = VPX_CODEC_ERROR;
    else {
        unsigned int i;
        for (i = 0; i < num_maps; i++, mmaps++) {
            if (!mmaps->base) break;
            if ((res = ctx->iface->set_mmap(ctx, mmaps))) break;
        }
    }
    return SAVE_STATUS(ctx, res);
}

/* ---- decoder entry points (vpx/src/vpx_decoder.c:21-248) ---------------------------------- */
vpx_codec_err_t vpx_codec_dec_init_ver(vpx_codec_ctx_t *ctx, vpx_codec_iface_t *iface, vpx_codec_dec_cfg_t *cfg,
                                       vpx_codec_flags_t flags, int ver)
{
    vpx_codec_err_t res;
    if (ver != VPX_DECODER_ABI_VERSION) res = VPX_CODEC_ABI_MISMATCH;
    else if (!ctx || !iface) res = VPX_CODEC_INVALID_PARAM;
    else if (iface->abi_version != VPX_CODEC_INTERNAL_ABI_VERSION) res = VPX_CODEC_ABI_MISMATCH;
    else if ((flags & VPX_CODEC_USE_XMA) && !(iface->caps & VPX_CODEC_CAP_XMA)) res = VPX_CODEC_INCAPABLE;
    else if ((flags & VPX_CODEC_USE_POSTPROC) && !(iface->caps & VPX_CODEC_CAP_POSTPROC)) res = VPX_CODEC_INCAPABLE;
    else if ((flags & VPX_CODEC_USE_ERROR_CONCEALMENT) && !(iface->caps & VPX_CODEC_CAP_ERROR_CONCEALMENT))
        res = VPX_CODEC_INCAPABLE;
    else if ((flags & VPX_CODEC_USE_INPUT_FRAGMENTS) && !(iface->caps & VPX_CODEC_CAP_INPUT_FRAGMENTS))
        res = VPX_CODEC_INCAPABLE;
    else if (!(iface->caps & VPX_CODEC_CAP_DECODER)) res = VPX_CODEC_INCAPABLE;
    else {
        memset(ctx, 0, sizeof *ctx);
        ctx->iface = iface;
        ctx->name = iface->name;
        ctx->priv = NULL;
        ctx->init_flags = flags;
        ctx->config.dec = cfg;
        res = ctx->iface->init(ctx, NULL);
        if (res) {
            ctx->err_detail = ctx->priv ? ctx->priv->err_detail : NULL;
            vpx_codec_destroy(ctx);
        }
        if (ctx->priv) ctx->priv->iface = ctx->iface;
    }
    return SAVE_STATUS(ctx, res);
}

vpx_codec_err_t vpx_codec_peek_stream_info(vpx_codec_iface_t *iface, const uint8_t *data, unsigned int data_sz,
                                           vpx_codec_stream_info_t *si)
{
    if (!iface || !data || !data_sz || !si || si->sz < sizeof(vpx_codec_stream_info_t)) return VPX_CODEC_INVALID_PARAM;
    si->w = 0;
    si->h = 0;
    return iface->dec.peek_si(data, data_sz, si);
}

vpx_codec_err_t vpx_codec_get_stream_info(vpx_codec_ctx_t *ctx, vpx_codec_stream_info_t *si)
{
    vpx_codec_err_t res;
    if (!ctx || !si || si->sz < sizeof(vpx_codec_stream_info_t)) res = VPX_CODEC_INVALID_PARAM;
    else if (!ctx->iface || !ctx->priv) res = VPX_CODEC_ERROR;
    else {
        si->w = 0;
        si->h = 0;
        res = ctx->iface->dec.get_si(ctx->priv->alg_priv, si);
    }
    return SAVE_STATUS(ctx, res);
}

vpx_codec_err_t vpx_codec_decode(vpx_codec_ctx_t *ctx, const uint8_t *data, unsigned int data_sz, void *user_priv,
                                 long deadline)
{
    vpx_codec_err_t res;
    if (!ctx || (!data && data_sz)) res = VPX_CODEC_INVALID_PARAM;      /* NULL data only with data_sz == 0 */
    else if (!ctx->iface || !ctx->priv) res = VPX_CODEC_ERROR;
    else res = ctx->iface->dec.decode(ctx->priv->alg_priv, data, data_sz, user_priv, deadline);
    return SAVE_STATUS(ctx, res);
}

vpx_image_t *vpx_codec_get_frame(vpx_codec_ctx_t *ctx, vpx_codec_iter_t *iter)
{
    if (!ctx || !iter || !ctx->iface || !ctx->priv) return NULL;
    return ctx->iface->dec.get_frame(ctx->priv->alg_priv, iter);
}

vpx_codec_err_t vpx_codec_register_put_frame_cb(vpx_codec_ctx_t *ctx, vpx_codec_put_frame_cb_fn_t cb, void *user_priv)
{
    vpx_codec_err_t res;
    if (!ctx || !cb) res = VPX_CODEC_INVALID_PARAM;
    else if (!ctx->iface || !ctx->priv || !(ctx->iface->caps & VPX_CODEC_CAP_PUT_FRAME)) res = VPX_CODEC_ERROR;
    else {
        ctx->priv->dec.put_frame_cb.fn = cb;
        ctx->priv->dec.put_frame_cb.user_priv = user_priv;
        res = VPX_CODEC_OK;
    }
    return SAVE_STATUS(ctx, res);
}

vpx_codec_err_t vpx_codec_register_put_slice_cb(vpx_codec_ctx_t *ctx, vpx_codec_put_slice_cb_fn_t cb, void *user_priv)
{
    vpx_codec_err_t res;
    if (!ctx || !cb) res = VPX_CODEC_INVALID_PARAM;
    else if (!ctx->iface || !ctx->priv || !(ctx->iface->caps & VPX_CODEC_CAP_PUT_SLICE)) res = VPX_CODEC_ERROR;
    else {
        ctx->priv->dec.put_slice_cb.fn = cb;
        ctx->priv->dec.put_slice_cb.user_priv = user_priv;
        res = VPX_CODEC_OK;
    }
    return SAVE_STATUS(ctx, res);
}

/* ---- vpx_img_* (vpx/src/vpx_image.c) --------------------------------------------------------- */
static vpx_image_t *img_setup(vpx_image_t *img, vpx_img_fmt_t fmt, unsigned int d_w, unsigned int d_h,
                              unsigned int stride_align, unsigned char *img_data)
{
    unsigned int h, w, s, xcs, ycs, bps, align;
    if (!stride_align) stride_align = 1;
    if (stride_align & (stride_align - 1)) goto fail;
    switch (fmt) {
    case VPX_IMG_FMT_RGB32: case VPX_IMG_FMT_RGB32_LE: case VPX_IMG_FMT_ARGB: case VPX_IMG_FMT_ARGB_LE: bps = 32; break;
    case VPX_IMG_FMT_RGB24: case VPX_IMG_FMT_BGR24: bps = 24; break;
    case VPX_IMG_FMT_RGB565: case VPX_IMG_FMT_RGB565_LE: case VPX_IMG_FMT_RGB555: case VPX_IMG_FMT_RGB555_LE:
    case VPX_IMG_FMT_UYVY: case VPX_IMG_FMT_YUY2: case VPX_IMG_FMT_YVYU: bps = 16; break;
    case VPX_IMG_FMT_I420: case VPX_IMG_FMT_YV12: case VPX_IMG_FMT_VPXI420: case VPX_IMG_FMT_VPXYV12: bps = 12; break;
    default: bps = 16; break;
    }
    xcs = ycs = (fmt & VPX_IMG_FMT_PLANAR) ? 1 : 0;
    align = (1u << xcs) - 1;
    w = (d_w + align) & ~align;
    align = (1u << ycs) - 1;
    h = (d_h + align) & ~align;
    s = (fmt & VPX_IMG_FMT_PLANAR) ? w : bps * w / 8;
    s = (s + stride_align - 1) & ~(stride_align - 1);
    if (!img) {
        img = (vpx_image_t *)calloc(1, sizeof *img);
        if (!img) goto fail;
        img->self_allocd = 1;
    } else
        memset(img, 0, sizeof *img);
    img->img_data = img_data;
    if (!img_data) {
        img->img_data = (unsigned char *)malloc((fmt & VPX_IMG_FMT_PLANAR) ? h * w * bps / 8 : h * s);
        img->img_data_owner = 1;
    }
    if (!img->img_data) goto fail;
    img->fmt = fmt;
    img->w = w;
    img->h = h;
    img->x_chroma_shift = xcs;
    img->y_chroma_shift = ycs;
    img->bps = (int)bps;
    img->stride[VPX_PLANE_Y] = img->stride[VPX_PLANE_ALPHA] = (int)s;
    img->stride[VPX_PLANE_U] = img->stride[VPX_PLANE_V] = (int)(s >> xcs);
    if (!vpx_img_set_rect(img, 0, 0, d_w, d_h)) return img;
fail:
    vpx_img_free(img);
    return NULL;
}

vpx_image_t *vpx_img_alloc(vpx_image_t *img, vpx_img_fmt_t fmt, unsigned int d_w, unsigned int d_h, unsigned int align)
{
    return img_setup(img, fmt, d_w, d_h, align, NULL);
}

vpx_image_t *vpx_img_wrap(vpx_image_t *img, vpx_img_fmt_t fmt, unsigned int d_w, unsigned int d_h, unsigned int align,
                          unsigned char *img_data)
{
    return img_setup(img, fmt, d_w, d_h, align, img_data);
}

int vpx_img_set_rect(vpx_image_t *img, unsigned int x, unsigned int y, unsigned int w, unsigned int h)
{
    unsigned char *data;
    if (x + w > img->w || y + h > img->h) return -1;
    img->d_w = w;
    img->d_h = h;
    if (!(img->fmt & VPX_IMG_FMT_PLANAR)) {
        img->planes[VPX_PLANE_PACKED] = img->img_data + x * img->bps / 8 + y * img->stride[VPX_PLANE_PACKED];
        return 0;
    }
    data = img->img_data;
    if (img->fmt & VPX_IMG_FMT_HAS_ALPHA) {
        img->planes[VPX_PLANE_ALPHA] = data + x + y * img->stride[VPX_PLANE_ALPHA];
        data += img->h * img->stride[VPX_PLANE_ALPHA];
    }
    img->planes[VPX_PLANE_Y] = data + x + y * img->stride[VPX_PLANE_Y];
    data += img->h * img->stride[VPX_PLANE_Y];
    {
        int first = (img->fmt & VPX_IMG_FMT_UV_FLIP) ? VPX_PLANE_V : VPX_PLANE_U;
        int second = (img->fmt & VPX_IMG_FMT_UV_FLIP) ? VPX_PLANE_U : VPX_PLANE_V;
        img->planes[first] = data + (x >> img->x_chroma_shift) + (y >> img->y_chroma_shift) * img->stride[first];
        data += (img->h >> img->y_chroma_shift) * img->stride[first];
        img->planes[second] = data + (x >> img->x_chroma_shift) + (y >> img->y_chroma_shift) * img->stride[second];
    }
    return 0;
}

void vpx_img_flip(vpx_image_t *img)
{
    /* vertical flip by pointing at the last row and negating the strides */
    img->planes[VPX_PLANE_Y] += (signed)(img->d_h - 1) * img->stride[VPX_PLANE_Y];
    img->stride[VPX_PLANE_Y] = -img->stride[VPX_PLANE_Y];
    img->planes[VPX_PLANE_U] += (signed)((img->d_h >> img->y_chroma_shift) - 1) * img->stride[VPX_PLANE_U];
    img->stride[VPX_PLANE_U] = -img->stride[VPX_PLANE_U];
    img->planes[VPX_PLANE_V] += (signed)((img->d_h >> img->y_chroma_shift) - 1) * img->stride[VPX_PLANE_V];
    img->stride[VPX_PLANE_V] = -img->stride[VPX_PLANE_V];
    img->planes[VPX_PLANE_ALPHA] += (signed)(img->d_h - 1) * img->stride[VPX_PLANE_ALPHA];
    img->stride[VPX_PLANE_ALPHA] = -img->stride[VPX_PLANE_ALPHA];
}

void vpx_img_free(vpx_image_t *img)
{
    if (!img) return;
    if (img->img_data && img->img_data_owner) free(img->img_data);
    if (img->self_allocd) free(img);
}
