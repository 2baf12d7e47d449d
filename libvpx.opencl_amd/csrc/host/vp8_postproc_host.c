/* See vp8_postproc_host.h. */
#include "vp8_postproc_host.h"
#include "vp8_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int vp8_pp_rand(vp8_pp_state *st)
{
    uint32_t v;
    if (!st->rng_ready) {
        int32_t w = 1;
        int i;
        st->rng[0] = 1;
        for (i = 1; i < 31; i++) {                       /* 16807 * w mod (2^31 - 1), Schrage's way like the C library */
            const int32_t hi = w / 127773, lo = w % 127773;
            w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            st->rng[i] = (uint32_t)w;
        }
        for (i = 31; i < 34; i++) st->rng[i] = st->rng[i - 31];
        st->rng_pos = 0;                                 /* rng[] is a ring of the last 34 words; pos = slot of word i */
        st->rng_ready = 1;
        for (i = 34; i < 344; i++) (void)vp8_pp_rand(st);
    }
    /* word i = word i-31 + word i-3; the ring holds words i-34 .. i-1 with word i-34 at rng_pos */
    v = st->rng[(st->rng_pos + 3) % 34] + st->rng[(st->rng_pos + 31) % 34];
    st->rng[st->rng_pos] = v;
    st->rng_pos = (st->rng_pos + 1) % 34;
    return (int)(v >> 1);
}

/* the deblocking threshold as a function of q (postproc.c:334-335,354-355) */
static int deblock_limit(int q)
{
    const double level = 6.0e-05 * q * q * q - .0067 * q * q + .306 * q + .0065;
    return (int)(level + .5);
}

/* q2mbl (postproc.c:223-228) */
static int demacroblock_limit(int x)
{
    if (x < 20) x = 20;
    x = 50 + (x - 50) * 10 / 8;
    return x * x / 3;
}

/* fillrd (postproc.c:410-465): 256 values distributed like a gaussian whose width grows with the noise level `a` and
 * shrinks with q, sampled 3072 times */
static void build_noise(vp8_pp_state *st, int q, int a)
{
    int8_t dist[300];
    const double sigma = a + .5 + .6 * (63 - q) / 63.0;
    int next = 0, i, j;
    for (i = -32; i < 32; i++) {
        const double x = i;
        const int n = (int)(.5 + 256 * (1 / (sigma * sqrt(2.0 * 3.14159265)) * exp(-x * x / (2 * sigma * sigma))));
        for (j = 0; j < n; j++) dist[next + j] = (int8_t)i;
        if (n > 0) next += n;
    }
    for (; next < 256; next++) dist[next] = 0;
    for (i = 0; i < 3072; i++) st->noise[i] = dist[vp8_pp_rand(st) & 0xff];
    st->clamp = -dist[0];
    st->last_q = q;
    st->last_noise = a;
}

int vp8_pp_prepare(vp8_pp_state *st, const vp8_postproc_cfg_t *cfg, int filter_level, int rows, vp8hip_pp *pp)
{
    /* (VP8_MFQE: vp8_pp_mfqe_step; the VP8_DEBUG_* overlays are not implemented: the flags are accepted and have no effect) */
    const int flags = cfg->post_proc_flag & (VP8_DEBLOCK | VP8_DEMACROBLOCK | VP8_ADDNOISE);
    int q = filter_level * 10 / 6, r;
    memset(pp, 0, sizeof *pp);
    if (!flags) return 0;
    if (q > 63) q = 63;
    if (flags & VP8_DEMACROBLOCK) {                       /* wins over VP8_DEBLOCK (postproc.c:970-981) */
        const int qd = q + (cfg->deblocking_level - 5) * 10;
        pp->flags |= VP8HIP_PP_DEMACROBLOCK;
        pp->flimit = deblock_limit(qd);
        pp->mb_flimit = demacroblock_limit(qd);
        pp->rv = vp8t_pp_rv;
        pp->rv_offset = 63 & vp8_pp_rand(st);                    /* vp8_mbpost_proc_down_c, postproc.c:286 */
    } else if (flags & VP8_DEBLOCK) {
        pp->flags |= VP8HIP_PP_DEBLOCK;
        pp->flimit = deblock_limit(q);
    }
    if (flags & VP8_ADDNOISE) {
        pp->flags |= VP8HIP_PP_ADDNOISE;
        /* postproc.c:988-993: the table is rebuilt when last_q differs from q -- and fillrd stores ITS argument, 63 - q */
        if (st->last_q != q || st->last_noise != cfg->noise_level) {
            build_noise(st, 63 - q, cfg->noise_level);
            pp->noise = st->noise;
        }
        if (rows > (int)sizeof st->noise_rows) rows = (int)sizeof st->noise_rows;
        for (r = 0; r < rows; r++) st->noise_rows[r] = (uint8_t)(vp8_pp_rand(st) & 0xff);
        pp->noise_rows = st->noise_rows;
        pp->noise_clamp = st->clamp;
    }
    return pp->flags;
}

int vp8_pp_mfqe_step(vp8_pp_state *st, const vp8_postproc_cfg_t *cfg, int base_qindex, int *qprev)
{
    st->shown++;
    if ((cfg->post_proc_flag & VP8_MFQE) && st->shown >= 2 && base_qindex - st->last_base_qindex >= 10) {
        *qprev = st->last_base_qindex;
        st->last_base_qindex = (3 * st->last_base_qindex + base_qindex) >> 2;
        return 1;
    }
    st->last_base_qindex = base_qindex;
    return 0;
}

void vp8_pp_mfqe_classes(const vp8ir_frame_hdr *hdr, const void *mb_array, size_t mb_stride, const vp8ir_mv *mvs, uint8_t *cls)
{
    const int n = hdr->mb_cols * hdr->mb_rows;
    int i;
    for (i = 0; i < n; i++) {
        const vp8ir_mb *mbs = (const vp8ir_mb *)((const char *)mb_array + (size_t)i * mb_stride) - i;   /* mbs[i] = descriptor i */
        int still = hdr->frame_type == 0;
        if (!still) {
            const int intra = mbs[i].ref_frame == VP8IR_INTRA_FRAME || !mvs;
            const int r = intra ? 0 : mvs[i * 16 + 15].row, c = intra ? 0 : mvs[i * 16 + 15].col;
            still = abs(r) <= 10 && abs(c) <= 10;
        }
        cls[i] = !still ? 0 : (mbs[i].y_mode == VP8IR_B_PRED || mbs[i].y_mode == VP8IR_SPLITMV) ? 2 : 1;
    }
}
