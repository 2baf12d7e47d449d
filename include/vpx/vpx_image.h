/* include/vpx/vpx_image.h -- image descriptor of the vpx codec API.
 *
 * Interface-compatible restatement of the reference's vpx/vpx_image.h (enum values, struct layout
 * and function names are the ABI; see vpx/vpx_image.h:31-242 and vpx/src/vpx_image.c).
 * VPX_IMAGE_ABI_VERSION 1.
 */
#ifndef VPX_IMAGE_H
#define VPX_IMAGE_H
#ifdef __cplusplus
extern "C" {
#endif

#define VPX_IMAGE_ABI_VERSION (1)

#define VPX_IMG_FMT_PLANAR     0x100
#define VPX_IMG_FMT_UV_FLIP    0x200
#define VPX_IMG_FMT_HAS_ALPHA  0x400

typedef enum vpx_img_fmt {
    VPX_IMG_FMT_NONE, VPX_IMG_FMT_RGB24, VPX_IMG_FMT_RGB32, VPX_IMG_FMT_RGB565, VPX_IMG_FMT_RGB555,
    VPX_IMG_FMT_UYVY, VPX_IMG_FMT_YUY2, VPX_IMG_FMT_YVYU, VPX_IMG_FMT_BGR24, VPX_IMG_FMT_RGB32_LE,
    VPX_IMG_FMT_ARGB, VPX_IMG_FMT_ARGB_LE, VPX_IMG_FMT_RGB565_LE, VPX_IMG_FMT_RGB555_LE,
    VPX_IMG_FMT_YV12    = VPX_IMG_FMT_PLANAR | VPX_IMG_FMT_UV_FLIP | 1,
    VPX_IMG_FMT_I420    = VPX_IMG_FMT_PLANAR | 2,
    VPX_IMG_FMT_VPXYV12 = VPX_IMG_FMT_PLANAR | VPX_IMG_FMT_UV_FLIP | 3,
    VPX_IMG_FMT_VPXI420 = VPX_IMG_FMT_PLANAR | 4
} vpx_img_fmt_t;

#define VPX_PLANE_PACKED 0
#define VPX_PLANE_Y      0
#define VPX_PLANE_U      1
#define VPX_PLANE_V      2
#define VPX_PLANE_ALPHA  3

typedef struct vpx_image {
    vpx_img_fmt_t  fmt;
    unsigned int   w, h;            /* stored size (decoder: stride / padded height)  */
    unsigned int   d_w, d_h;        /* displayed size                                 */
    unsigned int   x_chroma_shift, y_chroma_shift;
    unsigned char *planes[4];       /* top-left pixel of each plane                   */
    int            stride[4];
    int            bps;
    void          *user_priv;
    unsigned char *img_data;        /* private */
    int            img_data_owner;  /* private */
    int            self_allocd;     /* private */
} vpx_image_t;

typedef struct vpx_image_rect { unsigned int x, y, w, h; } vpx_image_rect_t;

vpx_image_t *vpx_img_alloc(vpx_image_t *img, vpx_img_fmt_t fmt, unsigned int d_w, unsigned int d_h,
                           unsigned int align);
vpx_image_t *vpx_img_wrap(vpx_image_t *img, vpx_img_fmt_t fmt, unsigned int d_w, unsigned int d_h,
                          unsigned int align, unsigned char *img_data);
int  vpx_img_set_rect(vpx_image_t *img, unsigned int x, unsigned int y, unsigned int w, unsigned int h);
void vpx_img_flip(vpx_image_t *img);
void vpx_img_free(vpx_image_t *img);

#ifdef __cplusplus
}
#endif
#endif
