/* oracle/roi_enc.c -- TEST INFRASTRUCTURE ONLY: our harness around the REFERENCE encoder's public API (vpx/vpx_encoder.h,
 * vpx/vp8cx.h), linked against oracle/_ref/libvpxref.so, that makes a stream vpxenc cannot: one whose inter frames have
 * segmentation ON without coding a map.  The application installs a region-of-interest map (VP8E_SET_ROI_MAP: vp8_set_roimap,
 * vp8/encoder/onyx_if.c:5112-5153), which switches segmentation on; this encoder then clears the one-shot update flags of every
 * inter frame before it packs the header (encode_frame_to_data_rate, onyx_if.c:3213-3216) and switches segmentation off again at
 * key frames (setup_features, :305-323) -- so what reaches the stream is segmentation_enabled = 1, update_mb_segmentation_map = 0,
 * update_mb_segmentation_data = 0: frames that KEEP their segment map (vp8/decoder/decodemv.c:594-606), here one that no frame
 * ever coded.  That is the decoder path tests need a reference-made sample of (the exporter of the device's entropy decoder
 * refuses such frames unless the device keeps the map); streams whose kept map has content come from tests/vp8_writer.py.
 * tests/golden/make_fixtures.py runs this to produce p_roi_640x360.ivf.
 *
 *   roi_enc <w> <h> <in.i420> <out.ivf> <roi frame> [<second roi frame>]
 * encodes every frame of the raw I420 input (one pass, good quality, one thread, a key frame first and no other) and installs a
 * four-segment map before frame <roi frame> (1-based) and, optionally, a different one before <second roi frame>.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define VPX_CODEC_DISABLE_COMPAT 1
#include "vpx/vpx_encoder.h"
#include "vpx/vp8cx.h"

static void le16(unsigned char *p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }
static void le32(unsigned char *p, unsigned v) { le16(p, v & 0xffff); le16(p + 2, v >> 16); }

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s <w> <h> <in.i420> <out.ivf> <roi frame> [<second roi frame>]\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), roi_at = atoi(argv[5]), roi2_at = argc > 6 ? atoi(argv[6]) : -1;
    FILE *in = fopen(argv[3], "rb"), *out = fopen(argv[4], "wb");
    if (!in || !out) { fprintf(stderr, "cannot open files\n"); return 1; }
    vpx_codec_enc_cfg_t cfg;
    vpx_codec_ctx_t codec;
    if (vpx_codec_enc_config_default(vpx_codec_vp8_cx(), &cfg, 0)) return 1;
    cfg.g_w = w; cfg.g_h = h; cfg.g_threads = 1; cfg.g_lag_in_frames = 0; cfg.g_pass = VPX_RC_ONE_PASS;
    cfg.rc_target_bitrate = 700; cfg.kf_mode = VPX_KF_AUTO; cfg.kf_min_dist = 0; cfg.kf_max_dist = 9999;
    cfg.g_timebase.num = 1; cfg.g_timebase.den = 30;
    if (vpx_codec_enc_init(&codec, vpx_codec_vp8_cx(), &cfg, 0)) { fprintf(stderr, "enc_init: %s\n", vpx_codec_error(&codec)); return 1; }
    vpx_codec_control(&codec, VP8E_SET_CPUUSED, 2);
    vpx_codec_control(&codec, VP8E_SET_ENABLEAUTOALTREF, 0);
    vpx_image_t img;
    if (!vpx_img_alloc(&img, VPX_IMG_FMT_I420, w, h, 1)) return 1;
    unsigned char hdr[32] = { 'D', 'K', 'I', 'F' };
    le16(hdr + 4, 0); le16(hdr + 6, 32); memcpy(hdr + 8, "VP80", 4); le16(hdr + 12, w); le16(hdr + 14, h);
    le32(hdr + 16, 30); le32(hdr + 20, 1);
    fwrite(hdr, 1, 32, out);
    const int rows = (h + 15) / 16, cols = (w + 15) / 16;
    unsigned char *map = malloc((size_t)rows * cols);
    int n = 0, written = 0, eof = 0;
    for (;;) {
        vpx_image_t *pic = NULL;
        if (!eof) {
            size_t got = 0;
            for (int pl = 0; pl < 3; pl++) {
                const int pw = pl ? (w + 1) / 2 : w, ph = pl ? (h + 1) / 2 : h;
                for (int y = 0; y < ph; y++) got += fread(img.planes[pl] + (size_t)y * img.stride[pl], 1, pw, in);
            }
            if (got == (size_t)w * h + 2 * (size_t)((w + 1) / 2) * ((h + 1) / 2)) pic = &img; else eof = 1;
        }
        if (pic && (n + 1 == roi_at || n + 1 == roi2_at)) {
            vpx_roi_map_t roi;
            const int second = n + 1 == roi2_at;
            memset(&roi, 0, sizeof roi);
            roi.rows = rows; roi.cols = cols; roi.roi_map = map;
            for (int r = 0; r < rows; r++)
                for (int c = 0; c < cols; c++) map[r * cols + c] = (unsigned char)(second ? ((r / 2 + c / 3) & 3) : ((r + c) & 3));
            const int dq[4] = { 0, -6, 8, -12 }, dlf[4] = { 0, 4, -3, 7 };
            for (int k = 0; k < 4; k++) { roi.delta_q[k] = dq[(k + second) & 3]; roi.delta_lf[k] = dlf[(k + second) & 3]; roi.static_threshold[k] = 0; }
            if (vpx_codec_control(&codec, VP8E_SET_ROI_MAP, &roi)) { fprintf(stderr, "roi: %s\n", vpx_codec_error(&codec)); return 1; }
        }
        if (vpx_codec_encode(&codec, pic, n, 1, 0, VPX_DL_GOOD_QUALITY)) { fprintf(stderr, "encode: %s\n", vpx_codec_error(&codec)); return 1; }
        vpx_codec_iter_t it = NULL;
        const vpx_codec_cx_pkt_t *pkt;
        int got_pkt = 0;
        while ((pkt = vpx_codec_get_cx_data(&codec, &it)))
            if (pkt->kind == VPX_CODEC_CX_FRAME_PKT) {
                unsigned char fh[12];
                le32(fh, (unsigned)pkt->data.frame.sz); le32(fh + 4, (unsigned)pkt->data.frame.pts); le32(fh + 8, 0);
                fwrite(fh, 1, 12, out); fwrite(pkt->data.frame.buf, 1, pkt->data.frame.sz, out);
                written++; got_pkt = 1;
            }
        if (!pic && !got_pkt) break;
        if (pic) n++;
    }
    le32(hdr + 24, (unsigned)written);
    fseek(out, 0, SEEK_SET); fwrite(hdr, 1, 32, out);
    fclose(out); fclose(in);
    vpx_codec_destroy(&codec);
    fprintf(stderr, "%d frames in, %d written\n", n, written);
    return 0;
}
