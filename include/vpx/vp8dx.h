/* include/vpx/vp8dx.h -- the VP8 decoder algorithm interface.
 * Same entry points as the reference's vpx/vp8dx.h:31-80 / vp8/vp8_dx_iface.c:776-802
 * (CODEC_INTERFACE(vpx_codec_vp8_dx)); here it is backed by the MI355X HIP pixel path. */
#ifndef VP8DX_H
#define VP8DX_H
#include "vp8.h"
#ifdef __cplusplus
extern "C" {
#endif

extern vpx_codec_iface_t  vpx_codec_vp8_dx_algo;
extern vpx_codec_iface_t *vpx_codec_vp8_dx(void);

enum vp8_dec_control_id {
    VP8D_GET_LAST_REF_UPDATES = VP8_DECODER_CTRL_ID_START,
    VP8D_GET_FRAME_CORRUPTED,
    VP8D_GET_LAST_REF_USED,
    VP8_DECODER_CTRL_ID_MAX
};

VPX_CTRL_USE_TYPE(VP8D_GET_LAST_REF_UPDATES, int *)
VPX_CTRL_USE_TYPE(VP8D_GET_FRAME_CORRUPTED,  int *)
VPX_CTRL_USE_TYPE(VP8D_GET_LAST_REF_USED,    int *)

#ifdef __cplusplus
}
#endif
#endif
