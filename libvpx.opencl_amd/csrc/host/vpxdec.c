/* vpxdec -- command line decoder on top of the public vpx codec API (MI355X HIP backend).
 *
 * Command-line contract of the reference's vpxdec.c for the options the parity harness and the
 * benchmarks use (vpxdec.c:66-135 option table, :1034-1124 main loop): IVF input, --i420 / --yv12,
 * --md5 (one digest over all output, printed as "<md5>  <outfile>"), -o/--output, --noblit,
 * --summary / --progress (frames, microseconds inside vpx_codec_decode only, fps -- the same
 * bracket as vpxdec.c:1041-1055), --limit, --skip, -t/--threads (host threads for the token partitions of a frame: the
 * entropy decode is the CPU side of this decoder, vp8_parser_set_threads), --codec=vp8, -v, and the VP8 post-processing options --postproc, --deblock,
 * --demacroblock-level=<n>, --noise-level=<n>, --mfqe (vpxdec.c:111-133, 779-812, 983-1002).
 * Input: IVF or WebM, probed in that order like vpxdec.c:573-587 (webm.h; the reference reads WebM through its bundled
 * nestegg).  Headerless raw input is not provided.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#define VPX_CODEC_DISABLE_COMPAT 1
#include "vpx/vpx_decoder.h"
#include "vpx/vp8dx.h"
#include "ivf.h"
#include "webm.h"
#include "md5.h"

static const char *exec_name;

static void usage_exit(void)
{
    fprintf(stderr,
            "Usage: %s <options> filename\n\nOptions:\n"
            "      --codec=<arg>     Codec to use (vp8)\n"
            "      --yv12            Output raw YV12 frames\n"
            "      --i420            Output raw I420 frames\n"
            "      --flipuv          Flip the chroma planes in the output\n"
            "      --noblit          Don't process the decoded frames\n"
            "      --postproc        Postprocess decoded frames\n"
            "      --deblock         Enable VP8 deblocking\n"
            "      --demacroblock-level=<arg>  Enable VP8 demacroblocking, w/ level\n"
            "      --noise-level=<arg>         Enable VP8 postproc add noise\n"
            "      --progress        Show progress after each frame decodes\n"
            "      --limit=<arg>     Stop decoding after n frames\n"
            "      --skip=<arg>      Skip the first n input frames\n"
            "      --summary         Show timing summary\n"
            "  -o, --output=<arg>    Output file name\n"
            "  -t, --threads=<arg>   Max threads to use (token partitions of a frame are decoded in parallel)\n"
            "  -v, --verbose         Show version string\n"
            "      --md5             Compute the MD5 sum of the decoded frames\n\n"
            "Included decoders:\n\n    vp8    - %s\n", exec_name, vpx_codec_iface_name(vpx_codec_vp8_dx()));
    exit(EXIT_FAILURE);
}

static const char *optval(const char *arg, const char *name, char **argv, int *i, int argc, const char *shortname)
{
    size_t n = strlen(name);
    if (!strncmp(arg, name, n) && arg[n] == '=') return arg + n + 1;
    if ((!strcmp(arg, name) || (shortname && !strcmp(arg, shortname))) && *i + 1 < argc) return argv[++*i];
    return NULL;
}

static unsigned long now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (unsigned long)t.tv_sec * 1000000ul + (unsigned long)t.tv_nsec / 1000ul;
}

static void show_progress(int in, int out, unsigned long us)
{
    fprintf(stderr, "%d decoded frames/%d showed frames in %lu us (%.2f fps)\r", in, out, us,
            (float)out * 1000000.0 / (float)us);
}

int main(int argc, char **argv)
{
    const char *fn = NULL, *outfile = NULL, *v;
    int use_y4m_order = 0, flipuv = 0, noblit = 0, do_md5 = 0, progress = 0, summary = 0, verbose = 0;
    int stop_after = 0, skip = 0, frame_in = 0, frame_out = 0, frames_corrupted = 0, rc, postproc = 0;
    vp8_postproc_cfg_t pp_cfg = { 0, 0, 0 };
    unsigned long dx_time = 0;
    vpx_codec_ctx_t decoder;
    vpx_codec_dec_cfg_t cfg = { 0, 0, 0 };
    ivf_reader in;
    webm_reader wm;
    int is_webm = 0;
    FILE *out = NULL;
    md5_state md5;
    const uint8_t *buf;
    size_t buf_sz;

    exec_name = argv[0];
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "--yv12")) use_y4m_order = 0, flipuv = 1;
        else if (!strcmp(a, "--i420")) flipuv = 0;
        else if (!strcmp(a, "--flipuv")) flipuv = 1;
        else if (!strcmp(a, "--noblit")) noblit = 1;
        else if (!strcmp(a, "--progress")) progress = 1;
        else if (!strcmp(a, "--summary")) summary = 1;
        else if (!strcmp(a, "--md5")) do_md5 = 1;
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) verbose = 1;
        else if (!strcmp(a, "--postproc")) postproc = 1;
        else if (!strcmp(a, "--deblock")) postproc = 1, pp_cfg.post_proc_flag |= VP8_DEBLOCK;
        else if (!strcmp(a, "--mfqe")) postproc = 1, pp_cfg.post_proc_flag |= VP8_MFQE;
        else if ((v = optval(a, "--noise-level", argv, &i, argc, NULL)))
            postproc = 1, pp_cfg.post_proc_flag |= VP8_ADDNOISE, pp_cfg.noise_level = atoi(v);
        else if ((v = optval(a, "--demacroblock-level", argv, &i, argc, NULL)))
            postproc = 1, pp_cfg.post_proc_flag |= VP8_DEMACROBLOCK, pp_cfg.deblocking_level = atoi(v);
        else if ((v = optval(a, "--codec", argv, &i, argc, NULL))) {
            if (strcmp(v, "vp8")) { fprintf(stderr, "Error: Unrecognized argument (%s) to --codec\n", v); return EXIT_FAILURE; }
        } else if ((v = optval(a, "--limit", argv, &i, argc, NULL))) stop_after = atoi(v);
        else if ((v = optval(a, "--skip", argv, &i, argc, NULL))) skip = atoi(v);
        else if ((v = optval(a, "--output", argv, &i, argc, "-o"))) outfile = v;
        else if ((v = optval(a, "--threads", argv, &i, argc, "-t"))) cfg.threads = (unsigned)atoi(v);
        else if (a[0] == '-' && a[1]) { fprintf(stderr, "Error: Unrecognized option %s\n", a); usage_exit(); }
        else fn = a;
    }
    (void)use_y4m_order;
    if (!fn) usage_exit();
    rc = ivf_open(&in, fn);
    if (rc == -1) { fprintf(stderr, "Failed to open file '%s'\n", fn); return EXIT_FAILURE; }
    if (rc) {
        ivf_close(&in);
        if (!strcmp(fn, "-") || webm_open(&wm, fn)) { fprintf(stderr, "Unrecognized input file type.\n"); return EXIT_FAILURE; }
        is_webm = 1;
    } else if (in.fourcc != 0x30385056) fprintf(stderr, "Notice -- IVF header indicates codec: %08x\n", in.fourcc);
    if (!noblit) {
        if (do_md5) md5_init(&md5);
        else if (outfile) {
            out = strcmp(outfile, "-") ? fopen(outfile, "wb") : stdout;
            if (!out) { fprintf(stderr, "Failed to output file"); return EXIT_FAILURE; }
        } else { fprintf(stderr, "Not dumping raw video to your terminal. Use '-o -' to override.\n"); return EXIT_FAILURE; }
    }
    if (vpx_codec_dec_init(&decoder, vpx_codec_vp8_dx(), &cfg, postproc ? VPX_CODEC_USE_POSTPROC : 0)) {
        fprintf(stderr, "Failed to initialize decoder: %s\n", vpx_codec_error(&decoder));
        return EXIT_FAILURE;
    }
    if (pp_cfg.post_proc_flag && vpx_codec_control(&decoder, VP8_SET_POSTPROC, &pp_cfg)) {
        fprintf(stderr, "Failed to configure postproc: %s\n", vpx_codec_error(&decoder));
        return EXIT_FAILURE;
    }
    if (verbose) fprintf(stderr, "%s\n", decoder.name);

#define NEXT_FRAME() (is_webm ? webm_next(&wm, &buf, &buf_sz) : ivf_next(&in, &buf, &buf_sz))
    while (skip-- > 0 && NEXT_FRAME() == 1) { }
    while (NEXT_FRAME() == 1) {
        vpx_codec_iter_t iter = NULL;
        vpx_image_t *img;
        unsigned long t0 = now_us();
        int corrupted = 0;
        if (vpx_codec_decode(&decoder, buf, (unsigned)buf_sz, NULL, 0)) {
            const char *detail = vpx_codec_error_detail(&decoder);
            fprintf(stderr, "Failed to decode frame: %s\n", vpx_codec_error(&decoder));
            if (detail) fprintf(stderr, "  Additional information: %s\n", detail);
            goto fail;
        }
        dx_time += now_us() - t0;
        ++frame_in;
        if (vpx_codec_control(&decoder, VP8D_GET_FRAME_CORRUPTED, &corrupted)) {
            fprintf(stderr, "Failed VP8_GET_FRAME_CORRUPTED: %s\n", vpx_codec_error(&decoder));
            goto fail;
        }
        frames_corrupted += corrupted;
        if ((img = vpx_codec_get_frame(&decoder, &iter))) ++frame_out;
        if (progress) show_progress(frame_in, frame_out, dx_time);
        if (!noblit && img) {
            const int order[3] = { VPX_PLANE_Y, flipuv ? VPX_PLANE_V : VPX_PLANE_U, flipuv ? VPX_PLANE_U : VPX_PLANE_V };
            for (int k = 0; k < 3; k++) {
                const unsigned char *p = img->planes[order[k]];
                unsigned rows = k ? (1 + img->d_h) / 2 : img->d_h, w = k ? (1 + img->d_w) / 2 : img->d_w;
                for (unsigned y = 0; y < rows; y++, p += img->stride[order[k]]) {
                    if (do_md5) md5_update(&md5, p, w);
                    else fwrite(p, 1, w, out);
                }
            }
        }
        if (stop_after && frame_in >= stop_after) break;
    }
    if (summary || progress) { show_progress(frame_in, frame_out, dx_time); fprintf(stderr, "\n"); }
    if (frames_corrupted) fprintf(stderr, "WARNING: %d frames corrupted.\n", frames_corrupted);
fail:
    if (vpx_codec_destroy(&decoder)) {
        fprintf(stderr, "Failed to destroy decoder: %s\n", vpx_codec_error(&decoder));
        return EXIT_FAILURE;
    }
    if (!noblit) {
        if (do_md5) {
            unsigned char d[16];
            md5_final(&md5, d);
            for (int i = 0; i < 16; i++) printf("%02x", d[i]);
            printf("  %s\n", outfile ? outfile : fn);
        } else if (out && out != stdout) fclose(out);
    }
    ivf_close(&in);
    if (is_webm) webm_close(&wm);
    return frames_corrupted ? EXIT_FAILURE : EXIT_SUCCESS;
}
