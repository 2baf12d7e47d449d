/* decode_damaged [--ec] [--lose N,N,...] [--cut N:BYTES,...] <in.ivf> <out.md5>
 *
 * decode_to_md5 over a stream that is damaged on the way in, the way a lossy transport damages it: frames N (1-based) of --lose
 * never arrive -- the application says so with vpx_codec_decode(ctx, NULL, 0, ...), as vpx/vpx_decoder.h documents for lost
 * frames --, frames of --cut arrive with only their first BYTES bytes.  --ec initialises the decoder with
 * VPX_CODEC_USE_ERROR_CONCEALMENT.  The reference's examples/decode_with_drops (decode_with_drops.txt) drops frames without telling
 * the decoder; this tool is that example with the two things error concealment exists for.  Output: decode_to_md5's lines for
 * the frames the decoder shows; a frame whose decode call fails gives the line "decode-error <frame %04d>" and the tool goes on.
 * oracle/ref_md5.c takes the same options around the reference decoder: tests/test_gpu_concealment.py compares the listings. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define VPX_CODEC_DISABLE_COMPAT 1
#include "vpx/vpx_decoder.h"
#include "vpx/vp8dx.h"
#include "ivf.h"
#include "md5.h"

#define MAXD 64
static int lose[MAXD], nlose, cut_at[MAXD], cut_to[MAXD], ncut;

static void parse_list(const char *s, int pairs)
{
    while (*s) {
        char *e;
        long a = strtol(s, &e, 10), b = 0;
        if (e == s) break;
        if (pairs) { if (*e != ':') break; b = strtol(e + 1, &e, 10); }
        if (pairs && ncut < MAXD) { cut_at[ncut] = (int)a; cut_to[ncut++] = (int)b; }
        if (!pairs && nlose < MAXD) lose[nlose++] = (int)a;
        s = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
}

int main(int argc, char **argv)
{
    ivf_reader in;
    FILE *out;
    vpx_codec_ctx_t codec;
    const uint8_t *frame;
    size_t frame_sz;
    int frame_cnt = 0, rc, a = 1, ec = 0, i;

    for (; a < argc && argv[a][0] == '-' && argv[a][1] == '-'; a++) {
        if (!strcmp(argv[a], "--ec")) ec = 1;
        else if (!strcmp(argv[a], "--lose") && a + 1 < argc) parse_list(argv[++a], 0);
        else if (!strcmp(argv[a], "--cut") && a + 1 < argc) parse_list(argv[++a], 1);
        else break;
    }
    if (argc - a != 2) { fprintf(stderr, "Usage: %s [--ec] [--lose N,...] [--cut N:BYTES,...] <infile> <outfile>\n", argv[0]); return EXIT_FAILURE; }
    if (ivf_open(&in, argv[a])) { fprintf(stderr, "%s is not an IVF file.\n", argv[a]); return EXIT_FAILURE; }
    if (!(out = fopen(argv[a + 1], "wb"))) { fprintf(stderr, "Failed to open %s for writing\n", argv[a + 1]); return EXIT_FAILURE; }
    if (vpx_codec_dec_init(&codec, vpx_codec_vp8_dx(), NULL, ec ? VPX_CODEC_USE_ERROR_CONCEALMENT : 0)) {
        fprintf(stderr, "Failed to initialize decoder: %s\n", vpx_codec_error(&codec));
        return EXIT_FAILURE;
    }
    while ((rc = ivf_next(&in, &frame, &frame_sz)) == 1) {
        vpx_codec_iter_t iter = NULL;
        vpx_image_t *img;
        int lost = 0, err;
        frame_cnt++;
        for (i = 0; i < nlose; i++) lost |= lose[i] == frame_cnt;
        for (i = 0; i < ncut; i++)
            if (cut_at[i] == frame_cnt && (size_t)cut_to[i] < frame_sz) frame_sz = (size_t)cut_to[i];
        err = lost ? vpx_codec_decode(&codec, NULL, 0, NULL, 0) : vpx_codec_decode(&codec, frame, (unsigned)frame_sz, NULL, 0);
        if (err) {
            fprintf(stderr, "frame %d: %s\n", frame_cnt, vpx_codec_error(&codec));
            fprintf(out, "decode-error %04d\n", frame_cnt);
            continue;
        }
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
            for (i = 0; i < 16; i++) fprintf(out, "%02x", sum[i]);
            fprintf(out, "  img-%dx%d-%04d.i420\n", img->d_w, img->d_h, frame_cnt);
        }
    }
    vpx_codec_destroy(&codec);
    fclose(out);
    ivf_close(&in);
    return EXIT_SUCCESS;
}
