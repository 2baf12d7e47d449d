/* oracle/ref_md5.c -- TEST INFRASTRUCTURE.  Our harness around the REFERENCE's public
 * decoder API (linked against oracle/_ref/libvpxref.so, built from /root/reference).
 *
 * Prints one line per shown frame in the format of the reference's generated
 * `decode_to_md5` example (examples/decode_to_md5.txt:28-47):
 *     "<32 hex>  img-<d_w>x<d_h>-<frame_cnt %04d>.i420"
 * and, unlike that example (fixed 256 KiB frame buffer, decoder_tmpl.c:53,82), accepts
 * frames of any size.  Options:
 *     ref_md5 in.ivf out.md5            per-frame md5 listing
 *     ref_md5 --time N in.ivf           decode the file N times, print
 *                                       "frames pixels seconds" (timing brackets only
 *                                       vpx_codec_decode, like vpxdec.c:1041-1055)
 *     ref_md5 --dump K in.ivf out.i420  write shown frame K (1-based) as raw I420
 *     ref_md5 --damage [--ec] [--lose N,...] [--cut N:BYTES,...] in.ivf out.md5
 *                                       the options of the product's decode_damaged (csrc/host/decode_damaged.c) around the
 *                                       reference decoder: frames that never arrive (vpx_codec_decode(NULL, 0)), frames cut
 *                                       short, VPX_CODEC_USE_ERROR_CONCEALMENT; "decode-error NNNN" lines instead of giving up
 *                                       (oracle/_ref/ref_md5_ec is this file against the reference configured
 *                                       --enable-error-concealment)
 *     ref_md5 --pp FLAGS LEVEL NOISE in.ivf out.md5
 *                                       per-frame md5 listing of the POST-PROCESSED output: the decoder is initialised with
 *                                       VPX_CODEC_USE_POSTPROC and, unless FLAGS is -1 (the reference's default
 *                                       configuration, vp8_dx_iface.c:421-431), VP8_SET_POSTPROC {FLAGS, LEVEL, NOISE}
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#define VPX_CODEC_DISABLE_COMPAT 1
#include "vpx/vpx_decoder.h"
#include "vpx/vp8dx.h"
#include "md5_utils.h"

static unsigned rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

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
        if (*e != ',') break;
        s = e + 1;
    }
}

int main(int argc, char **argv) {
    int reps = 1, timing = 0, dumpk = 0, pp = 0, damage = 0, ec = 0;
    vp8_postproc_cfg_t ppcfg = { 0, 0, 0 };
    int ai = 1;
    if (argc > 2 && !strcmp(argv[1], "--time")) { timing = 1; reps = atoi(argv[2]); ai = 3; }
    else if (argc > 2 && !strcmp(argv[1], "--dump")) { dumpk = atoi(argv[2]); ai = 3; }
    else if (argc > 4 && !strcmp(argv[1], "--pp")) {
        pp = 1; ppcfg.post_proc_flag = atoi(argv[2]); ppcfg.deblocking_level = atoi(argv[3]); ppcfg.noise_level = atoi(argv[4]); ai = 5;
    }
    else if (argc > 1 && !strcmp(argv[1], "--damage")) {
        damage = 1;
        for (ai = 2; ai < argc && argv[ai][0] == '-' && argv[ai][1] == '-'; ai++) {
            if (!strcmp(argv[ai], "--ec")) ec = 1;
            else if (!strcmp(argv[ai], "--lose") && ai + 1 < argc) parse_list(argv[++ai], 0);
            else if (!strcmp(argv[ai], "--cut") && ai + 1 < argc) parse_list(argv[++ai], 1);
            else break;
        }
    }
    if (argc - ai < (timing ? 1 : 2)) { fprintf(stderr, "usage: see header comment\n"); return 2; }
    FILE *f = fopen(argv[ai], "rb");
    if (!f) { perror(argv[ai]); return 1; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *buf = malloc(n);
    if (fread(buf, 1, n, f) != (size_t)n) return 1;
    fclose(f);
    if (n < 32 || memcmp(buf, "DKIF", 4)) { fprintf(stderr, "not IVF\n"); return 1; }
    FILE *out = timing ? NULL : fopen(argv[ai + 1], "wb");
    double secs = 0; long frames = 0; double pixels = 0;
    for (int r = 0; r < reps; r++) {
        vpx_codec_ctx_t c;
        if (vpx_codec_dec_init(&c, vpx_codec_vp8_dx(), NULL, (pp ? VPX_CODEC_USE_POSTPROC : 0) | (ec ? VPX_CODEC_USE_ERROR_CONCEALMENT : 0))) {
            fprintf(stderr, "init failed: %s\n", vpx_codec_error(&c)); return 1;
        }
        if (pp && ppcfg.post_proc_flag >= 0 && vpx_codec_control(&c, VP8_SET_POSTPROC, &ppcfg)) { fprintf(stderr, "VP8_SET_POSTPROC failed\n"); return 1; }
        long pos = 32; int cnt = 0;
        while (pos + 12 <= n) {
            unsigned sz = rd32(buf + pos); pos += 12;
            if (pos + sz > n) break;
            cnt++;
            unsigned use = sz;
            int lost = 0;
            for (int i = 0; i < nlose; i++) lost |= lose[i] == cnt;
            for (int i = 0; i < ncut; i++) if (cut_at[i] == cnt && (unsigned)cut_to[i] < use) use = (unsigned)cut_to[i];
            double t0 = now();
            int err = lost ? vpx_codec_decode(&c, NULL, 0, NULL, 0) : vpx_codec_decode(&c, buf + pos, use, NULL, 0);
            secs += now() - t0;
            pos += sz;
            if (err && damage) {
                fprintf(stderr, "frame %d: %s\n", cnt, vpx_codec_error(&c));
                fprintf(out, "decode-error %04d\n", cnt);
                continue;
            }
            if (err) { fprintf(stderr, "decode error frame %d: %s\n", cnt, vpx_codec_error(&c)); return 1; }
            vpx_codec_iter_t it = NULL; vpx_image_t *img;
            while ((img = vpx_codec_get_frame(&c, &it))) {
                frames++; pixels += (double)img->d_w * img->d_h;
                if (timing) continue;
                if (dumpk) {
                    if (cnt != dumpk) continue;
                    for (int pl = 0; pl < 3; pl++) {
                        unsigned char *p = img->planes[pl];
                        unsigned w = pl ? (img->d_w + 1) >> 1 : img->d_w, h = pl ? (img->d_h + 1) >> 1 : img->d_h;
                        for (unsigned y = 0; y < h; y++, p += img->stride[pl]) fwrite(p, 1, w, out);
                    }
                    continue;
                }
                MD5Context m; unsigned char d[16];
                MD5Init(&m);
                for (int pl = 0; pl < 3; pl++) {
                    unsigned char *p = img->planes[pl];
                    unsigned w = pl ? (img->d_w + 1) >> 1 : img->d_w, h = pl ? (img->d_h + 1) >> 1 : img->d_h;
                    for (unsigned y = 0; y < h; y++, p += img->stride[pl]) MD5Update(&m, p, w);
                }
                MD5Final(d, &m);
                for (int i = 0; i < 16; i++) fprintf(out, "%02x", d[i]);
                fprintf(out, "  img-%dx%d-%04d.i420\n", img->d_w, img->d_h, cnt);
            }
        }
        vpx_codec_destroy(&c);
    }
    if (timing) printf("%ld %.0f %.6f\n", frames, pixels, secs);
    if (out) fclose(out);
    free(buf);
    return 0;
}
