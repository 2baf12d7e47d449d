/* Host side of the output-side post-processing: what vp8_post_proc_frame (vp8/common/postproc.c:903-1000) decides per frame
 * before the filters run -- thresholds from the frame's loop-filter level, the random phases, the noise table -- handed to
 * the device as a vp8hip_pp (include/vp8hip.h). */
#ifndef VP8_POSTPROC_HOST_H
#define VP8_POSTPROC_HOST_H
#include "vp8hip.h"
#include "vpx/vp8.h"

typedef struct vp8_pp_state {           /* struct postproc_state, vp8/common/postproc.h:16-25 */
    uint32_t rng[34]; int rng_pos; int rng_ready;      /* vp8_pp_rand */
    int    last_q, last_noise;
    int    clamp;                       /* blackclamp[0] == whiteclamp[0] */
    int8_t noise[3072];
    uint8_t noise_rows[16384];
    int    shown;                       /* frames shown so far: cm->current_video_frame as vp8_post_proc_frame sees it */
    int    last_base_qindex;            /* the quantiser index the picture in the post-processing buffer stands for */
} vp8_pp_state;

/* The reference draws its dither and noise phases from the C library's rand() and never seeds it (postproc.c:286,456,499), so
 * what its binaries produce is defined by the C library's sequence for seed 1.  The GPU runtime shares the process and may
 * draw from rand() itself, so that sequence is reproduced here, per decoder: glibc's default generator (additive feedback
 * over 31 words, x[i] = x[i-31] + x[i-3], seeded by the 16807 Lehmer generator, first 310 outputs dropped, >> 1).
 * tests/test_abi_cpu.py compares it with the C library's. */
int vp8_pp_rand(vp8_pp_state *st);

/* Fill *pp for one shown frame decoded with loop-filter level `filter_level`, `rows` = its 16-aligned height.  Draws in the
 * reference's order: once per demacroblocked frame, 3072 times per new noise table, once per noisy row.  Returns the
 * effective flags (0: show the frame as it is). */
int vp8_pp_prepare(vp8_pp_state *st, const vp8_postproc_cfg_t *cfg, int filter_level, int rows, vp8hip_pp *pp);

/* VP8_MFQE, the policy (vp8_post_proc_frame, postproc.c:911-925,948-986): to be called once for every SHOWN frame of a decoder
 * created with VPX_CODEC_USE_POSTPROC, before vp8_pp_prepare.  Returns 1 when the frame goes through
 * vp8_multiframe_quality_enhance -- VP8_MFQE set, at least the second frame shown, base_qindex 10 or more above the running
 * index -- with *qprev = that index; the running index then moves a quarter of the way (:969), otherwise it becomes the frame's. */
int vp8_pp_mfqe_step(vp8_pp_state *st, const vp8_postproc_cfg_t *cfg, int base_qindex, int *qprev);
/* ... and the class of every macroblock for vp8hip_mfqe (postproc.c:834-843): cls[mb_rows * mb_cols]; mvs may be NULL on key
 * frames.  A macroblock's vector is its last block's (decodemv.c:490), zero for intra macroblocks (:563).  mb_array: the
 * macroblocks' descriptors (vp8ir_mb), mb_stride bytes apart (64: a dense array; 128: the records of the device form). */
void vp8_pp_mfqe_classes(const vp8ir_frame_hdr *hdr, const void *mb_array, size_t mb_stride, const vp8ir_mv *mvs, uint8_t *cls);

#endif
