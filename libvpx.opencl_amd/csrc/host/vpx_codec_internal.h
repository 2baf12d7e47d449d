/* Internal side of the vpx codec API: the algorithm-interface vtable an implementation exports
 * (reference: vpx/internal/vpx_codec_internal.h:59-322, struct vpx_codec_iface; abi_version 4) and
 * the per-context private block.  Only the decoder half is populated in this library. */
#ifndef VPX_CODEC_INTERNAL_H
#define VPX_CODEC_INTERNAL_H
#include <stdarg.h>
#include "vpx/vpx_decoder.h"

#define VPX_CODEC_INTERNAL_ABI_VERSION (4)

typedef struct vpx_codec_alg_priv vpx_codec_alg_priv_t;

typedef vpx_codec_err_t (*vpx_codec_init_fn_t)(vpx_codec_ctx_t *ctx, void *mr_cfg);
typedef vpx_codec_err_t (*vpx_codec_destroy_fn_t)(vpx_codec_alg_priv_t *priv);
typedef vpx_codec_err_t (*vpx_codec_peek_si_fn_t)(const uint8_t *data, unsigned int data_sz, vpx_codec_stream_info_t *si);
typedef vpx_codec_err_t (*vpx_codec_get_si_fn_t)(vpx_codec_alg_priv_t *priv, vpx_codec_stream_info_t *si);
typedef vpx_codec_err_t (*vpx_codec_control_fn_t)(vpx_codec_alg_priv_t *priv, int ctrl_id, va_list ap);
typedef vpx_codec_err_t (*vpx_codec_decode_fn_t)(vpx_codec_alg_priv_t *priv, const uint8_t *data, unsigned int data_sz,
                                                 void *user_priv, long deadline);
typedef vpx_image_t *(*vpx_codec_get_frame_fn_t)(vpx_codec_alg_priv_t *priv, vpx_codec_iter_t *iter);
typedef vpx_codec_err_t (*vpx_codec_get_mmap_fn_t)(const vpx_codec_ctx_t *ctx, vpx_codec_mmap_t *mmap, vpx_codec_iter_t *iter);
typedef vpx_codec_err_t (*vpx_codec_set_mmap_fn_t)(vpx_codec_ctx_t *ctx, const vpx_codec_mmap_t *mmap);

typedef const struct vpx_codec_ctrl_fn_map {
    int                    ctrl_id;
    vpx_codec_control_fn_t fn;
} vpx_codec_ctrl_fn_map_t;

struct vpx_codec_iface {
    const char              *name;
    int                      abi_version;
    vpx_codec_caps_t         caps;
    vpx_codec_init_fn_t      init;
    vpx_codec_destroy_fn_t   destroy;
    vpx_codec_ctrl_fn_map_t *ctrl_maps;
    vpx_codec_get_mmap_fn_t  get_mmap;
    vpx_codec_set_mmap_fn_t  set_mmap;
    struct {
        vpx_codec_peek_si_fn_t   peek_si;
        vpx_codec_get_si_fn_t    get_si;
        vpx_codec_decode_fn_t    decode;
        vpx_codec_get_frame_fn_t get_frame;
    } dec;
    struct {                       /* encoder half: unused (NULL) in a decoder-only library */
        void *cfg_maps, *encode, *get_cx_data, *cfg_set, *get_glob_hdrs, *get_preview, *mr_get_mem_loc;
    } enc;
};

struct vpx_codec_priv {
    unsigned int          sz;
    vpx_codec_iface_t    *iface;
    vpx_codec_alg_priv_t *alg_priv;
    const char           *err_detail;
    vpx_codec_flags_t     init_flags;
    struct {
        struct { vpx_codec_put_frame_cb_fn_t fn; void *user_priv; } put_frame_cb;
        struct { vpx_codec_put_slice_cb_fn_t fn; void *user_priv; } put_slice_cb;
    } dec;
};

#endif
