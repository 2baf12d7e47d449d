/* RTCD table instance + vpx_rtcd() (see include/vp8_rtcd.h).  Reference: vp8/common/rtcd.c. */
#define RTCD_C
#include "vp8_rtcd.h"

int vp8_decode_mb_rows_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_RECON); }
int vp8_loop_filter_frame_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_LF); }
int vp8_yv12_extend_frame_borders_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_EXTEND); }
int vp8_decode_frame_pixels_hip(vp8hip_ctx *c, const vp8hip_job *j, int n) { return vp8hip_decode(c, j, n, VP8HIP_STAGE_ALL); }

void vpx_rtcd(void)
{
    /* one specialisation: gfx950 HIP.  (The reference's generated setter picks by CPU flags.) */
    vp8_decode_mb_rows = vp8_decode_mb_rows_hip;
    vp8_loop_filter_frame = vp8_loop_filter_frame_hip;
    vp8_yv12_extend_frame_borders_ptr = vp8_yv12_extend_frame_borders_hip;
    vp8_decode_frame_pixels = vp8_decode_frame_pixels_hip;
}
