/* VP8 host feeder: compressed frame -> frame-batched IR (include/vp8_ir.h).
 *
 * Pure CPU, no pixel work.  Replaces, for the feeder side only, the reference's
 *   vp8_decode_frame header part      vp8/decoder/decodframe.c:690-1077
 *   vp8_decode_mode_mvs               vp8/decoder/decodemv.c:622-672
 *   vp8_decode_mb_tokens              vp8/decoder/detokenize.c:183-405
 * Everything a pixel kernel needs leaves through the IR; the pixel path itself lives behind
 * include/vp8hip.h (HIP) -- this file never touches pixels.
 */
#ifndef VP8_PARSER_H
#define VP8_PARSER_H

#include <stddef.h>
#include <stdint.h>
#include "vp8_ir.h"
#include "vp8hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: numeric values of vpx_codec_err_t (vpx/vpx_codec.h:81-132) */
enum {
    VP8P_OK = 0,
    VP8P_ERROR = 1,
    VP8P_MEM_ERROR = 2,
    VP8P_UNSUP_BITSTREAM = 5,
    VP8P_CORRUPT_FRAME = 7,
    VP8P_INVALID_PARAM = 8
};

typedef struct vp8_parser vp8_parser;

vp8_parser *vp8_parser_create(void);
/* Token partitions of a frame are decoded by up to `threads` threads (1..8; default 1).  Only frames coded with several
 * partitions gain (decodframe.c:501-592); the output is the serial decoder's byte for byte.  This is what
 * vpx_codec_dec_cfg_t::threads asks for (vpx/vpx_decoder.h:101-106; the reference's vp8/decoder/threading.c). */
void vp8_parser_set_threads(vp8_parser *p, int threads);
void vp8_parser_destroy(vp8_parser *p);
/* Error concealment (VPX_CODEC_USE_ERROR_CONCEALMENT; a reference build configured --enable-error-concealment, decoder created
 * with oxcf.error_concealment, onyxd_if.c:111-113).  To be set before the first frame.  Once a key frame has been decoded
 * completely, from the first inter frame on (vp8_parser_conceals): a frame that never came -- vp8_parser_begin_frame with size 0
 * -- is decoded as an inter frame whose motion vectors are extrapolated from the frame before (vp8_estimate_missing_mvs,
 * error_concealment.c:408); a frame whose token partitions end early keeps the prediction, without residual, from the macroblock
 * where they end, intra macroblocks among those predicted from the last frame with vectors interpolated from their neighbours
 * (vp8_interpolate_motion, :555), in key frames too (vp8_parser_frame_hdr); reference-refresh flags whose bits are missing take
 * their harmless values.  A short FIRST partition stays an error (the reference reads behind the buffer there). */
void vp8_parser_set_error_concealment(vp8_parser *p, int on);
int  vp8_parser_conceals(const vp8_parser *p);
/* The header of the frame vp8_parser_decode_mbs[_compact] has just decoded, as the pixel path is to see it: what begin_frame
 * returned, except for a key frame in which concealment replaced intra macroblocks by inter ones (frame_type 1, lf_key_frame 1:
 * vp8_ir.h).  Callers that enable concealment pass THIS header on; key frames then need the mv array too. */
void vp8_parser_frame_hdr(const vp8_parser *p, vp8ir_frame_hdr *out);

/* vp8_peek_si (vp8/vp8_dx_iface.c:245-285): key-frame start code + 14-bit dimensions. */
int vp8_parser_peek(const uint8_t *data, size_t size, int *is_key, int *width, int *height);

/* Step 1: frame tag + frame header (everything up to, not including, per-MB modes).
 * Fills *hdr.  On a key frame with new dimensions the parser re-allocates its per-MB state.
 * `data` must stay valid until vp8_parser_decode_mbs returns. */
int vp8_parser_begin_frame(vp8_parser *p, const uint8_t *data, size_t size, vp8ir_frame_hdr *hdr);
/* The same for a frame handed over in pieces (VPX_CODEC_USE_INPUT_FRAGMENTS: vp8dx_receive_compressed_data,
 * vp8/decoder/onyxd_if.c:336-366; setup_token_decoder, vp8/decoder/decodframe.c:501-592): frags[0] holds the frame header and
 * the first partition (and possibly more), each further fragment one or several whole token partitions. */
int vp8_parser_begin_frame_fragments(vp8_parser *p, const uint8_t *const *frags, const size_t *frag_sizes, int nfrags,
                                     vp8ir_frame_hdr *hdr);

/* Step 2: per-MB modes / motion vectors (first partition) and coefficient tokens (token
 * partitions) for the frame opened by begin_frame, written to caller-owned arrays sized for
 * hdr->mb_cols * hdr->mb_rows macroblocks:
 *   mbs  [n]        always
 *   coef [n * 400]  int16, column-major 4x4 blocks; only non-skipped MBs are written
 *   mvs  [n * 16]   inter frames only (may be NULL on key frames)
 * Returns VP8P_OK, or an error; *corrupt (optional) reports a truncated partition. */
int vp8_parser_decode_mbs(vp8_parser *p, vp8ir_mb *mbs, int16_t *coef, vp8ir_mv *mvs, int *corrupt);

/* The same straight into the DEVICE FORM of include/vp8_ir.h -- what an IR slot holds in HBM and the pixel kernels read: a
 * feeder gives the pinned staging of a slot (vp8hip_ir_map_compact) and the upload is one copy.  mbx[n]: the records; `blocks`:
 * 16 int16 per block with more than one coded position (at most cap_blocks of them; 24 per macroblock is the worst case);
 * *nblocks how many.  With vp8_parser_set_threads the rows of different token partitions stand in the stream thread by thread
 * (every row's blocks together, found through mbx[row start].d.sparse_first, as the form allows). */
int vp8_parser_decode_mbs_compact(vp8_parser *p, vp8ir_mbx *mbx, int16_t *blocks, size_t cap_blocks, size_t *nblocks, vp8ir_mv *mvs,
                                  int *corrupt);

/* Step 2 on the device (include/vp8hip.h: vp8hip_entropy_decode): instead of decoding the macroblocks, hand over what the
 * frame header left behind -- the first partition's decoder state at the first macroblock, the token partitions' extents (all
 * relative to the start of the buffer begin_frame was given), the probabilities.  Frames given as one buffer, without
 * concealment, and -- inter frames -- bringing their segment map if segmentation is on (VP8P_UNSUP_BITSTREAM otherwise, and the
 * frame stays open for vp8_parser_decode_mbs); a frame whose header ran past its data is VP8P_CORRUPT_FRAME.  Closes the frame.
 * The frame header is all the parser needs of a frame to go on to the next, with one exception: an inter frame that keeps the
 * segment map of a frame decoded this way is refused, by this call and by vp8_parser_decode_mbs. */
int vp8_parser_export_entropy(vp8_parser *p, vp8hip_entropy_frame *out);

/* The segment map on the device.  A macroblock's segment id persists from frame to frame until a frame codes it anew
 * (decodemv.c:594-606).  With the entropy decoder on the device the ids of the frame before are where that decoder left them: in
 * the records of the IR slot the frame was decoded into.  A caller that decodes EVERY frame of a stream on the device, into ONE
 * slot (many streams side by side: a slot each), says so here; vp8_parser_export_entropy then hands over the frames that keep their
 * map as well (vp8hip_entropy_frame::segmap_keep: the kernel takes the ids out of the slot before it overwrites them) instead of
 * refusing them. */
void vp8_parser_set_device_segmap(vp8_parser *p, int on);

const char *vp8_parser_error(const vp8_parser *p);

/* Reference-buffer index bookkeeping shared by every decoder built on the parser:
 * swap_frame_buffers / get_free_fb / ref_cnt_fb (vp8/decoder/onyxd_if.c:238-316). */
typedef struct vp8_refs {
    int new_idx, lst_idx, gld_idx, alt_idx;
    int ref_cnt[4];
    int show_idx;        /* frame_to_show */
} vp8_refs;

/* Per frame: get_free -> begin_frame -> (dimensions changed? on_alloc) -> decode -> swap. */
void vp8_refs_init(vp8_refs *r);                  /* at decoder creation */
void vp8_refs_on_alloc(vp8_refs *r);              /* vp8_alloc_frame_buffers, alloccommon.c:87-95 */
int  vp8_refs_get_free(vp8_refs *r);              /* -> new_idx, or -1 */
void vp8_refs_release_new(vp8_refs *r);           /* frame failed: give new_idx back */
int  vp8_refs_swap(vp8_refs *r, const vp8ir_frame_hdr *hdr);
int  vp8_refs_retarget_free(vp8_refs *r, int which);   /* vp8dx_set_reference: which = 1 last, 2 golden, 4 alt-ref */

#ifdef __cplusplus
}
#endif
#endif
