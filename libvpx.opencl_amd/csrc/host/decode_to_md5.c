/* decode_to_md5 <in.ivf> <out.md5> -- frame-by-frame MD5 of the decoded output, through the public
 * vpx codec API only.  Same command line and same output lines as the reference's generated example
 * (examples/decode_to_md5.txt:27-48 + decoder_tmpl.c:47-103): for every shown frame
 *     "<32 hex digits>  img-<d_w>x<d_h>-<frame count %04d>.i420"
 * Rows are hashed Y, then U, then V, honouring the image strides.  (Unlike the reference example this
 * tool has no 256 KiB frame-size limit.) */
#include <stdio.h>
#include <stdlib.h>
#define VPX_CODEC_DISABLE_COMPAT 1
#include "vpx/vpx_decoder.h"
#include "vpx/vp8dx.h"
#include "ivf.h"
#include "md5.h"

static void die_codec(vpx_codec_ctx_t *c, const char *s)
{
    const char *detail = vpx_codec_error_detail(c);
    fprintf(stderr, "%s: %s\n", s, vpx_codec_error(c));
    if (detail) fprintf(stderr, "    %s\n", detail);
    exit(EXIT_FAILURE);
}

int main(int argc, char **argv)
{
    ivf_reader in;
    FILE *out;
    vpx_codec_ctx_t codec;
    const uint8_t *frame;
    size_t frame_sz;
    int frame_cnt = 0, rc;

    if (argc != 3) { fprintf(stderr, "Usage: %s <infile> <outfile>\n", argv[0]); return EXIT_FAILURE; }
    if (ivf_open(&in, argv[1])) { fprintf(stderr, "%s is not an IVF file.\n", argv[1]); return EXIT_FAILURE; }
    if (!(out = fopen(argv[2], "wb"))) { fprintf(stderr, "Failed to open %s for writing\n", argv[2]); return EXIT_FAILURE; }
    printf("Using %s\n", vpx_codec_iface_name(vpx_codec_vp8_dx()));
    if (vpx_codec_dec_init(&codec, vpx_codec_vp8_dx(), NULL, 0)) die_codec(&codec, "Failed to initialize decoder");

    while ((rc = ivf_next(&in, &frame, &frame_sz)) == 1) {
        vpx_codec_iter_t iter = NULL;
        vpx_image_t *img;
        frame_cnt++;
        if (vpx_codec_decode(&codec, frame, (unsigned)frame_sz, NULL, 0)) die_codec(&codec, "Failed to decode frame");
        while ((img = vpx_codec_get_frame(&codec, &iter))) {
            md5_state md5;
            unsigned char sum[16];
            md5_init(&md5);
            for (int plane = 0; plane < 3; plane++) {
                const unsigned char *buf = img->planes[plane];
                unsigned rows = plane ? (img->d_h + 1) >> 1 : img->d_h, w = plane ? (img->d_w + 1) >> 1 : img->d_w;
                for (unsigned y = 0; y < rows; y++, buf += img->stride[plane]) md5_update(&md5, buf, w);
            }
            md5_final(&md5, sum);
            for (int i = 0; i < 16; i++) fprintf(out, "%02x", sum[i]);
            fprintf(out, "  img-%dx%d-%04d.i420\n", img->d_w, img->d_h, frame_cnt);
        }
    }
    if (rc < 0) fprintf(stderr, "Frame %d failed to read complete frame\n", frame_cnt + 1);
    printf("Processed %d frames.\n", frame_cnt);
    if (vpx_codec_destroy(&codec)) die_codec(&codec, "Failed to destroy codec");
    fclose(out);
    ivf_close(&in);
    return EXIT_SUCCESS;
}
