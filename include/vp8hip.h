/* include/vp8hip.h -- C ABI of the MI355X (gfx950) VP8 pixel path: libvp8hip.so.
 *
 * Plain C, opaque handle, int status returns (0 = OK, negative = error; text via
 * vp8hip_last_error), no HIP / C++ / torch types in any signature.  This is the frame-granular
 * boundary that sits where the reference's OpenCL offload branches sit (SURVEY.md 8b "B3"):
 *
 *   vp8hip_create / vp8hip_destroy      <- vp8dx_create_decompressor / vp8dx_remove_decompressor
 *                                          (vp8/decoder/onyxd_if.c:73,136) + cl_init/cl_destroy
 *                                          (vp8/common/opencl/vp8_opencl.c:40-84,155-260)
 *   vp8hip_configure                    <- vp8_alloc_frame_buffers (vp8/common/alloccommon.c:59)
 *                                          incl. the per-frame-buffer device memory the reference
 *                                          attaches as YV12_BUFFER_CONFIG.buffer_mem
 *                                          (vpx_scale/yv12config.h:63-65, yv12config.c:97-105)
 *   vp8hip_ir_map_compact /             <- the per-MB submit points of the reference's CL path
 *   vp8hip_ir_upload_compact               (vp8/decoder/decodframe.c:149-156: qcoeff/eobs/MODE_INFO
 *   (vp8hip_ir_map / vp8hip_ir_upload:     handed to vp8_decode_macroblock_cl)
 *    the same from dense arrays)
 *   vp8hip_decode                       <- decode_mb_row x mb_rows (decodframe.c:1116-1129),
 *                                          vp8_loop_filter_frame (vp8/common/loopfilter.c:203; its
 *                                          CL diversion at :225-230) and
 *                                          vp8_yv12_extend_frame_borders_ptr (onyxd_if.c:607)
 *   vp8hip_frame_download               <- the read-back of loopfilter_cl.c:688-696 / the plane
 *                                          pointers vp8dx_get_raw_frame exposes (onyxd_if.c:707)
 *
 * There is NO CPU fallback behind this interface: if the GPU or the code object is unavailable
 * vp8hip_create fails and the decoder built on it reports VPX_CODEC_ERROR.
 *
 * Threading: one context per decoder / per GPU; a context is not thread-safe.
 */
#ifndef VP8HIP_H
#define VP8HIP_H

#include <stddef.h>
#include <stdint.h>
#include "vp8_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp8hip_ctx vp8hip_ctx;

#define VP8HIP_STAGE_RECON   1   /* dequant+IDCT/WHT, intra + inter prediction, add            */
#define VP8HIP_STAGE_LF      2   /* in-loop deblocking (no-op for frames with filter_level 0) */
#define VP8HIP_STAGE_EXTEND  4   /* 32/16-px border replication                               */
#define VP8HIP_STAGE_ALL     7

/* One frame of work.  Slots and frame buffers are indices into the pools sized by
 * vp8hip_configure.  ref_fb[VP8IR_LAST_FRAME..VP8IR_ALTREF_FRAME] are read by inter MBs only
 * (ref_fb[0] unused; -1 where a reference does not exist).  Jobs passed to ONE vp8hip_decode
 * call must be mutually independent (no job's dst_fb is another's ref_fb): they run concurrently. */
typedef struct vp8hip_job {
    int32_t ir_slot;
    int32_t dst_fb;
    int32_t ref_fb[4];
} vp8hip_job;

typedef struct vp8hip_stats {      /* filled by vp8hip_get_stats; times from HIP events, ms */
    float recon_ms, lf_ms, extend_ms;   /* last vp8hip_decode call; extend_ms = border extension (wave-per-row kernels), or the
                                           tiled -> raster pass + borders if the launch ran it at once (detile_pass) */
    int   recon_waves, lf_waves;        /* waves per workgroup (1 = the lane-per-row kernels ran, 4 = the cross-CU
                                           variant of the wave-per-row kernels: small launches) */
    int   workgroups;
    int   detile_pass;                  /* 1: the launch produced the raster form of its frames at once (VP8HIP_EAGER_RASTER); by
                                           default a large launch leaves tiles and the pass runs when the raster form is asked for */
    int   lf_kernels;                   /* loop-filter kernels launched: 1 (wave-per-row family, some frame filtered) or 0 */
    int   fused;                        /* 1: the lane-per-row kernels -- vp8_keyframe_kernel, or vp8_inter_pred_kernel +
                                           vp8_interframe_kernel -- reconstructed AND filtered the launch; recon_ms covers it,
                                           lf_ms is 0 */
    int   pred_tiles;                   /* 1: the launch predicted its inter macroblocks from reference frames read as TILES
                                           (vp8_inter_pred_tiles_kernel: what the launch before left, no tiled -> raster pass in
                                           between); 0: from their border-extended raster form */
} vp8hip_stats;

/* device < 0: use the current HIP device.  Returns 0 or a negative error. */
int  vp8hip_create(int device, vp8hip_ctx **out);
void vp8hip_destroy(vp8hip_ctx *ctx);
const char *vp8hip_last_error(const vp8hip_ctx *ctx);   /* ctx may be NULL: creation error */

/* (Re)allocate device state for frames of width x height: num_fb frame buffers (vp8ir_geom layout, borders included) and
 * num_slots IR slots.  A slot holds one frame's macroblock data in the DEVICE FORM of include/vp8_ir.h (records, block stream
 * sized for the worst case, vectors: 960 bytes per macroblock) and gets pinned host staging of the same layout when it is first
 * mapped.  Existing contents are discarded. */
int  vp8hip_configure(vp8hip_ctx *ctx, int width, int height, int num_fb, int num_slots);
/* The same with the slots' block streams out of ONE pool.  The worst case a slot of vp8hip_configure is sized for -- 24 blocks a
 * macroblock, 768 of its 960 bytes per macroblock -- is rarely what a frame needs (a 1080p key frame of 200 KB: 62 k blocks = 2.0 of
 * 6.3 MB), and frames in flight are what the device's entropy decoder lives on (vp8hip_entropy_decode: a frame per lane).  Here a
 * slot holds records and vectors only (192 bytes per macroblock) and the entropy decoder takes the blocks' room out of the pool as
 * it goes, a chunk (four macroblock rows' worst case) at a time: a row's blocks stay together, the records' sparse_first count
 * from the pool's start, and vp8hip_decode reads the slots as ever.  vp8hip_pool_reset (on the context's stream: after the launches
 * queued so far) empties the pool -- when the frames decoded out of it have been through vp8hip_decode; the caller says when.  A
 * frame that finds the pool empty gets bit 1 of its status word (vp8hip_entropy_status) and is not to be decoded for its pixels: the
 * blocks it could not place went into the spare chunk behind the pool's last (all such frames share it), so its records'
 * sparse_first still point INSIDE the pool's allocation -- a vp8hip_decode over such a slot, queued before the status has come back,
 * reads and writes nothing out of bounds; what it leaves in the frame buffer is garbage (tests/test_gpu_entropy.py).  The host-side
 * producers (vp8hip_ir_map*, vp8hip_ir_upload*) are refused on such a context; vp8hip_ir_copy copies the records, which then share
 * the blocks; frames with several token partitions are decoded a frame per lane.  vp8hip_pool_usage (synchronises): bytes taken
 * since the last reset (more than *pool_bytes: that much was asked for) and the pool's size. */
int  vp8hip_configure_pooled(vp8hip_ctx *ctx, int width, int height, int num_fb, int num_slots, size_t pool_bytes);
int  vp8hip_pool_reset(vp8hip_ctx *ctx);
int  vp8hip_pool_usage(vp8hip_ctx *ctx, size_t *used_bytes, size_t *pool_bytes);
/* The two forms' pools are allocated when a launch or a reader first needs them (the tiled forms with the first large launch, the
 * raster forms with the first small launch, inter frame, download or filter); vp8hip_reserve allocates them now -- beside a first
 * launch of the entropy decoder, for instance: tens of GB take the allocator a second or two. */
int  vp8hip_reserve(vp8hip_ctx *ctx, int tiled_form, int raster_form);
/* What the context holds on the device right now, in bytes: the raster forms' pool, the tiled forms' pool, the IR slots, the block
 * pool (vp8hip_configure_pooled), the entropy decoder's input buffers, the staging of packed downloads. */
typedef struct vp8hip_memory {
    size_t raster_pool, tile_pool, slots, block_pool, entropy_input, packed_staging;
} vp8hip_memory;
int  vp8hip_memory_usage(const vp8hip_ctx *ctx, vp8hip_memory *out);
/* packed_staging: the batch as packed I420 -- what vp8hip_frames_fetch_i420_async sends, and what a digest-only fetch of
 * VP8HIP_MD5_PACK_FROM (12,288) tiled frames and more is hashed from: 3.1 MB per 1080p frame, 51 GB at 16,384, allocated when first
 * needed (a fetch that finds no room for it hashes the tiles) and KEPT for the next batch.  It is a cache: the library frees it by
 * itself when one of the two pools cannot be allocated beside it, and vp8hip_release_staging frees it now (waits for fetches in
 * flight). */
int  vp8hip_release_staging(vp8hip_ctx *ctx);
int  vp8hip_geometry(const vp8hip_ctx *ctx, vp8ir_geom *g);

/* Pinned host staging of a slot in the device form, for a feeder to write into directly (vp8_parser_decode_mbs_compact): mbx[nmb],
 * blocks[*cap_blocks * 16] right behind them, mvs[nmb * 16].  vp8hip_ir_upload_compact sends header, records and the first
 * nblocks blocks with ONE asynchronous copy on the context's stream (+ one for the vectors of an inter frame); the pixel
 * kernels read what arrives, nothing on the device touches it in between.  The staging may be rewritten once that copy has run
 * (vp8hip_sync, or any later synchronisation of the stream). */
int  vp8hip_ir_map_compact(vp8hip_ctx *ctx, int slot, vp8ir_frame_hdr **hdr, vp8ir_mbx **mbx, int16_t **blocks, size_t *cap_blocks,
                           vp8ir_mv **mvs);
int  vp8hip_ir_upload_compact(vp8hip_ctx *ctx, int slot, size_t nblocks);
/* The same from the DENSE view (mbs[nmb], coef[nmb * 400]: what the oracle and the tests speak): host memory the caller fills;
 * vp8hip_ir_upload turns it into the device form on the host (vp8ir_compact_mb), in the slot's staging, and uploads that.  It
 * waits for the context's stream first (the staging may still be on its way from the upload before). */
int  vp8hip_ir_map(vp8hip_ctx *ctx, int slot, vp8ir_frame_hdr **hdr, vp8ir_mb **mbs,
                   int16_t **coef, vp8ir_mv **mvs);
int  vp8hip_ir_upload(vp8hip_ctx *ctx, int slot);
/* Device-to-device replication of a slot (synthetic looped streams: every key frame
 * is independently decodable, decodframe.c:610-639). */
int  vp8hip_ir_copy(vp8hip_ctx *ctx, int dst_slot, int src_slot);

/* Enqueue the pixel path for njobs independent frames on the context's stream.  `stages` is a
 * mask of VP8HIP_STAGE_*.  Asynchronous; see vp8hip_sync. */
int  vp8hip_decode(vp8hip_ctx *ctx, const vp8hip_job *jobs, int njobs, int stages);

/* Copy one frame buffer to the host.  full != 0: the whole buffer (frame_size bytes, borders
 * included) into y (u, v ignored).  Otherwise the visible planes: rows of width x height (Y) and
 * ((w+1)/2) x ((h+1)/2) (U, V) written with the given destination strides.  Synchronous. */
int  vp8hip_frame_download(vp8hip_ctx *ctx, int fb, int full, uint8_t *y, uint8_t *u, uint8_t *v,
                           int y_stride, int uv_stride);
/* Output-side post-processing (vp8/common/postproc.c): frame buffer src_fb -> dst_fb, never back into
 * decoding.  The policy stays with the caller, as in the reference where vp8_post_proc_frame (postproc.c:903-1000) sits above
 * the filters: it turns the frame's loop-filter level into the thresholds and draws the random phases.
 *   VP8HIP_PP_DEBLOCK       vp8_deblock (:348-362): vp8_post_proc_down_and_across on Y, U, V with `flimit`
 *   VP8HIP_PP_DEMACROBLOCK  vp8_deblock_and_de_macro_block (:328-346): the same, then vp8_mbpost_proc_across_ip and
 *                           vp8_mbpost_proc_down on Y with `mb_flimit`; needs tmp_fb (a third buffer) and the dither table
 *                           `rv` (vp8_rv, 440 entries) with this frame's rv_offset = 63 & rand() (:286)
 *   VP8HIP_PP_ADDNOISE      vp8_plane_add_noise (:489-513) on Y: `noise` = the 3072-entry table fillrd (:410-465) built (NULL:
 *                           unchanged since the previous call), noise_clamp = its blackclamp[0], noise_rows[r] = rand() & 0xff
 *                           per row of the 16-aligned height.  Frames wider than 2816 are refused: the reference indexes
 *                           past the end of its table for them.
 * With neither of the first two flags dst_fb becomes a copy of src_fb (:982), or, with dst_fb == src_fb, stays what it is (the
 * noise alone, in place: after vp8hip_mfqe).  The filters read nothing outside the pictures' coded area (rows above and below
 * are the edge rows, as a frame's borders would say).  Asynchronous on the context's stream like
 * vp8hip_decode; the host arrays may be reused when the call returns. */
#define VP8HIP_PP_DEBLOCK       1
#define VP8HIP_PP_DEMACROBLOCK  2
#define VP8HIP_PP_ADDNOISE      4
typedef struct vp8hip_pp {
    int32_t flags;
    int32_t flimit;
    int32_t mb_flimit;
    int32_t rv_offset;
    int32_t noise_clamp;
    const int16_t *rv;
    const int8_t  *noise;
    const uint8_t *noise_rows;
} vp8hip_pp;
int  vp8hip_postproc(vp8hip_ctx *ctx, int src_fb, int dst_fb, int tmp_fb, const vp8hip_pp *pp);
/* VP8_MFQE, vp8_multiframe_quality_enhance (postproc.c:802-900): the frame about to be shown (show_fb) against the picture that
 * was shown before it (prev_fb: the output of the previous call chain, noise and all), macroblock by macroblock -- where the two
 * differ little for the old picture's activity and the step from qprev to qcurr the old picture is kept or blended in (it was
 * coded with the finer quantiser), elsewhere the new one is copied -- into dst_fb, which may be prev_fb.  mb_class: a byte per
 * macroblock in raster order, 0 = copy (inter frame, a vector component above 10: :836-841), 1 = one 16x16 block, 2 = four 8x8
 * blocks (B_PRED, SPLITMV: :843).  When to call it is the caller's policy, as for the filters (vp8_post_proc_frame :948-969:
 * from the second shown frame on, when base_qindex is 10 or more above the running last_base_qindex; the filters then run on
 * its output: vp8hip_postproc(dst_fb -> prev_fb), or with src_fb == dst_fb for the noise alone).  qprev <= qcurr <= 127.
 * Asynchronous on the context's stream; mb_class may be reused when the call returns. */
int  vp8hip_mfqe(vp8hip_ctx *ctx, int show_fb, int prev_fb, int dst_fb, const uint8_t *mb_class, int qcurr, int qprev);

/* Entropy decoding on the device (key frames).  What it replaces: the per-macroblock half of the host feeder -- the reference's
 * vp8_kfread_modes (vp8/decoder/decodemv.c:50-173) and vp8_decode_mb_tokens (vp8/decoder/detokenize.c:183-405) driven by
 * decode_mb_row (vp8/decoder/decodframe.c:293-470) -- which at ~10 ms per 1080p frame and core is what bounds a pipeline that
 * starts from the compressed stream.  A bool decoder is a serial machine, so a frame is ONE LANE's work (its token partitions
 * decoded in macroblock-row order like the reference's single thread does); the frames of a batch run side by side, 64 to a
 * wave.  The frame header stays with the host (a few thousand bools: vp8_parser_begin_frame), which hands over what it leaves
 * behind (vp8_parser_export_entropy, csrc/host/vp8_parser.h): the decoder state of the first partition where the per-macroblock
 * data start, the token partitions' extents, the probabilities.  The kernel writes the frames' slots in the device form of
 * include/vp8_ir.h -- what vp8_parser_decode_mbs_compact writes on the host -- and vp8hip_decode reads it as it stands.  Inter
 * frames too (vp8_decode_mode_mvs' per-macroblock half: reference frame, the near / nearest candidates from the
 * macroblocks above, left and above-left, NEWMV / SPLITMV vectors: decodemv.c:323-569) -- which only pays where many frames are
 * independent of each other, as the same position of many streams is.  Integer only. */
typedef struct vp8hip_entropy_frame {
    vp8ir_frame_hdr hdr;                /* as vp8_parser_begin_frame returned it; a frame of the context's size */
    uint64_t data_off;                  /* the frame's first byte in the buffer handed to vp8hip_entropy_decode */
    uint32_t first_pos, first_end;      /* first partition, relative to data_off: the next byte the decoder takes, and its end */
    uint32_t first_value;               /* ... its window (32 bits, the active byte on top), */
    int32_t  first_bits;                /*     the valid bits below the top byte (negative: refill due), */
    uint32_t first_range;               /*     and its range, 128..255 */
    uint32_t num_tok;                   /* 1, 2, 4 or 8 token partitions */
    uint32_t tok_pos[8], tok_end[8];    /* their extents, relative to data_off */
    uint8_t  update_mb_segmentation_map, mb_no_coeff_skip, prob_skip_false;
    uint8_t  segmap_keep;               /* 1: a macroblock's segment id is the one the slot's record holds from the frame before
                                           (a stream decoded frame after frame into this slot: vp8_parser_set_device_segmap) */
    uint8_t  segment_tree_probs[3], rsv1;
    uint8_t  coef_probs[1056];          /* [block type 4][band 8][context 3][node 11] */
    /* inter frames (hdr.frame_type 1; mb_mode_mv_init, decodemv.c:178-224): */
    uint8_t  prob_intra, prob_last, prob_gf, rsv2;
    uint8_t  ymode_prob[4];
    uint8_t  uvmode_prob[3], rsv3;
    uint8_t  mvc[2][19], rsv4[2];       /* motion-vector probabilities, row then column */
    uint8_t  rsv5[4];
} vp8hip_entropy_frame;
/* frames[i] -> IR slot first_slot + i.  `data`: the compressed frames (data_bytes in all; any host memory -- page-locked memory
 * from vp8hip_host_alloc makes the copy asynchronous, and then `frames` and `data` have to stay untouched until the next
 * vp8hip_sync).  Asynchronous on the context's stream. */
int  vp8hip_entropy_decode(vp8hip_ctx *ctx, int first_slot, int count, const vp8hip_entropy_frame *frames, const uint8_t *data,
                           size_t data_bytes);
/* The same in two steps, for a pipeline that wants the NEXT launch's input on its way while it still queues and waits on behalf of
 * the current one: vp8hip_entropy_stage checks the frames and sends them to the device (its own copy stream, one of two buffers),
 * vp8hip_entropy_decode(ctx, first_slot, count, NULL, NULL, 0) launches the kernel over what was staged.  One staged input at a time. */
int  vp8hip_entropy_stage(vp8hip_ctx *ctx, int count, const vp8hip_entropy_frame *frames, const uint8_t *data, size_t data_bytes);
/* What became of the frames of the last vp8hip_entropy_decode, a word each: bit 1 = the block pool was empty (vp8hip_configure_pooled:
 * the frame's slot is not to be decoded); bit 0 = a partition of the frame ended early, the
 * frame is corrupt (what vp8_parser_decode_mbs reports through *corrupt).  Synchronous.  _async: the copy is queued on the
 * context's stream into page-locked memory of the caller's (vp8hip_host_alloc) and has landed after the next synchronisation
 * (vp8hip_sync, or vp8hip_download_wait for a fetch queued behind it). */
int  vp8hip_entropy_status(vp8hip_ctx *ctx, int count, uint32_t *status);
int  vp8hip_entropy_status_async(vp8hip_ctx *ctx, int count, uint32_t *status);
/* The IR of a slot as it stands on the device, expanded to the dense view on the host (tests, debugging): mbs[nmb],
 * coef[nmb * 400], zeros where a block has no coefficients.  Synchronous. */
int  vp8hip_ir_fetch(vp8hip_ctx *ctx, int slot, vp8ir_mb *mbs, int16_t *coef);
int  vp8hip_ir_fetch_mvs(vp8hip_ctx *ctx, int slot, vp8ir_mv *mvs);      /* ... and its vectors: mvs[nmb * 16] */

/* Batch form for pipelines (tools/e2e.py, bin/batch_md5): `count` consecutive frame buffers, whole, as ONE asynchronous copy on
 * a stream of its own -- it starts when everything queued on the context's stream so far has finished and runs beside later
 * uploads (PCIe is full duplex).  dst: page-locked memory (vp8hip_host_alloc), frame i at dst + i * vp8hip_frame_stride(ctx)
 * (frame_size rounded up to 256).  vp8hip_download_wait returns when the copy has landed; one copy in flight at a time; a
 * launch that writes one of these frame buffers waits for it by itself. */
size_t vp8hip_frame_stride(const vp8hip_ctx *ctx);
int  vp8hip_frames_download_async(vp8hip_ctx *ctx, int first_fb, int count, uint8_t *dst);
int  vp8hip_download_wait(vp8hip_ctx *ctx);
/* The same with the frames' MD5s computed on the device: what `vpxdec --md5` (vpxdec.c:1080-1101) and examples/decode_to_md5
 * (decode_to_md5.txt) hash on the host -- the visible rows of the Y, U and V planes, vpx_image_t d_w x d_h -- one 16-byte digest
 * per frame into digests[16 * i], a frame per lane (csrc/hip/vp8_md5.hip).  dst and digests may each be NULL (not both); both
 * have landed when vp8hip_download_wait returns.  Any display size: where a row is whole MD5 blocks (width a multiple of 128) the
 * kernel reads block by block, from whichever form the frames are in; other widths are hashed from the raster form by a kernel
 * whose blocks straddle rows (correct, slower: the conformance streams' odd sizes, not the throughput path). */
int  vp8hip_frames_fetch_async(vp8hip_ctx *ctx, int first_fb, int count, uint8_t *dst, uint8_t *digests);
/* The same with the frames delivered as PACKED I420: d_w x d_h luma, then the two (d_w / 2) x ((d_h + 1) / 2) chroma planes, back to
 * back, no borders, no strides -- what `vpxdec --i420` writes (vpxdec.c:1080-1101) and what the digests are taken over --,
 * vp8hip_i420_bytes() per frame, frame i at dst + i * vp8hip_i420_bytes().  A pass on the device packs the frames from whichever
 * form they are in (tiles as they are; no raster form is needed), the copy engines take them out: a tenth less over the link than
 * whole frame buffers (1080p: 3.11 MB a frame instead of 3.43), which is what a pipeline that downloads every frame is bound by.
 * Display widths that are not a multiple of 8 are refused with -3. */
size_t vp8hip_i420_bytes(const vp8hip_ctx *ctx);
int  vp8hip_frames_fetch_i420_async(vp8hip_ctx *ctx, int first_fb, int count, uint8_t *dst, uint8_t *digests);
/* The digests alone, of ANY n frame buffers (fbs[i]; not necessarily neighbours: the shown frames of many streams decoded side by
 * side, bin/batch_md5 --streams): digests[16 * i].  Same stream and same wait as vp8hip_frames_fetch_async. */
int  vp8hip_frames_md5_list_async(vp8hip_ctx *ctx, const int *fbs, int n, uint8_t *digests);
/* A frame buffer has two forms on the device: the RASTER form (vp8ir_geom: the reference's YV12 layout, borders included), which
 * the small-launch kernels write and everything that reads pixels by coordinate reads (inter prediction, vp8hip_frame_download,
 * the post-processing filters), and the TILED form a large launch leaves (macroblock-window tiles: the form in which a lane of
 * vp8_keyframe_kernel can write whole 64-byte sectors).  The library converts a frame when something needs the form it is not in
 * -- never behind the caller's back after a launch -- and reads tiles where it can: the MD5 kernel of vp8hip_frames_fetch_async
 * walks them.  vp8hip_frames_to_raster asks for the raster form of `count` consecutive frame buffers explicitly (asynchronous, on
 * the context's stream; a no-op for frames that have it).  vp8hip_set_direct_download(ctx, 1): a batch download of tiled frames
 * into page-locked memory IS the tiled -> raster pass, a kernel writing the host buffer (the frames' raster form never exists in
 * HBM; what lands in the destination's border bytes is then undefined) -- faster than the copy engines on an otherwise idle
 * device, slower beside other kernels, hence off by default (VP8HIP_DIRECT_DOWNLOAD=1 sets the default).
 * INTER PREDICTION reads a reference frame in either form (round 5).  A large launch ONE of whose references exists only as tiles
 * -- streams decoded in lock step: every launch predicts from what the launch before left -- reads all its references as tiles
 * (vp8_inter_pred_tiles_kernel; borders are address clamps): no tiled -> raster pass runs, and a reference that exists only in
 * raster form (a golden frame a small launch decoded, an uploaded one) is given its tiled form once (vp8_retile_kernel) and keeps
 * both.  Any other launch -- every reference has a raster form already, or the launch is a small one -- reads the raster form.
 * vp8hip_set_pred_tiles(ctx, mode): 1 that rule (the default; VP8HIP_PRED_TILES sets it), 2 large launches always read tiles
 * (references without them are retiled), 0 never -- the three give the same frames, bit for bit.  Frames 16 pixels wide are always
 * read in raster form (a strip of the tile reader replicates one horizontal edge, and theirs reach past both). */
int  vp8hip_frames_to_raster(vp8hip_ctx *ctx, int first_fb, int count);
int  vp8hip_set_direct_download(vp8hip_ctx *ctx, int on);
int  vp8hip_set_pred_tiles(vp8hip_ctx *ctx, int mode);
/* Upload a whole frame buffer (frame_size bytes) -- tests and VP8_SET_REFERENCE. */
int  vp8hip_frame_upload(vp8hip_ctx *ctx, int fb, const uint8_t *buf);
int  vp8hip_frame_copy(vp8hip_ctx *ctx, int dst_fb, int src_fb);

/* Page-locked host memory for the caller's side of vp8hip_frame_download / vp8hip_frame_upload (a download into pageable
 * memory runs at a fraction of the PCIe rate).  The reference keeps its frame buffers in host memory it owns
 * (vp8_yv12_alloc_frame_buffer, vpx_scale/generic/yv12config.c:45-110); this is the host mirror's allocator. */
void *vp8hip_host_alloc(vp8hip_ctx *ctx, size_t bytes);
void  vp8hip_host_free(vp8hip_ctx *ctx, void *p);

/* Waits for the context's work.  Also reports (-1 + vp8hip_last_error) if a kernel of the cross-CU family gave up on
 * a row hand-over -- a defect, not an input error; the frames of that launch are invalid.  vp8hip_frame_download checks
 * the same. */
int  vp8hip_sync(vp8hip_ctx *ctx);
int  vp8hip_get_stats(vp8hip_ctx *ctx, vp8hip_stats *st);
/* Stats of an earlier launch: back = 0 the last vp8hip_decode call, 1 the one before, ... (up to 31).  Waits
 * for that launch only, so a caller can time a pipelined sequence and read the kernel times afterwards. */
int  vp8hip_get_stats_at(vp8hip_ctx *ctx, int back, vp8hip_stats *st);
/* The HIP stream (hipStream_t, as void*) the work of this context is enqueued on, so callers can bracket it with their own
 * events.  (vp8hip_join: rounds 1-3 ran a pass on a second stream that callers with work of their own had to join; there is
 * no such stream any more and the call does nothing.  A caller that reads frame buffers with kernels of its own asks for
 * their raster form first: vp8hip_frames_to_raster.) */
void *vp8hip_stream(vp8hip_ctx *ctx);
int  vp8hip_join(vp8hip_ctx *ctx);

/* ---- per-block test surface of the lane-per-row arithmetic (csrc/hip/vp8_lane_blocks.hip): the per-lane code the large-launch
 * kernels are made of, one block / macroblock per lane, on the current HIP device.  Each call is a launch and a synchronisation;
 * returns 0 or a negative error.  Counterparts in the reference: vp8_loop_filter_frame's per-macroblock calls
 * (vp8/common/loopfilter.c:259-299: vp8_loop_filter_{mbv,bv,mbh,bh}[_simple]), vp8_intra4x4_predict (reconintra4x4.c:16),
 * vp8_dequant_idct_add_c (dequantize.c:29). */
/* in / out: n x 400 bytes = rows and columns -4..15 of a luma macroblock (20 x 20); par: n x 8 bytes = mblim, blim, lim, hev_thr,
 * left edge filtered, inner edges filtered, top edge filtered, filter type (0 normal, 1 simple) */
int  vp8hip_lane_loop_filter_mbs(const uint8_t *in, uint8_t *out, const uint8_t *par, int n);
/* mode: n B_PREDICTION_MODEs; ctx: n x 16 bytes = above[0..7], left[0..3], top_left, 3 bytes of padding; out: n x 16 bytes, row-major */
int  vp8hip_lane_intra4x4(const uint8_t *mode, const uint8_t *ctx, uint8_t *out, int n);
/* coef: n x 16 in IR order (column-major, vp8_ir.h); dq: n x (dc, ac); pred / out: n x 16 bytes, row-major */
int  vp8hip_lane_dequant_idct_add(const int16_t *coef, const int16_t *dq, const uint8_t *pred, uint8_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* VP8HIP_H */
