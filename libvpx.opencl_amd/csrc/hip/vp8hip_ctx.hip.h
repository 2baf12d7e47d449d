// Internal to libvp8hip.so: the context behind include/vp8hip.h's opaque handle, shared by the shim's translation units --
//   vp8hip.hip          context, pools, IR slots (upload / copy / fetch), frame buffers, statistics
//   vp8hip_launch.hip   vp8hip_decode: which kernels a launch runs; the two forms a frame buffer has on the device
//   vp8hip_entropy.hip  vp8hip_entropy_decode
//   vp8hip_postproc.hip vp8hip_postproc, vp8hip_mfqe
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "vp8hip.h"
#include "vp8_common.hip.h"

#define VP8HIP_STATS_RING 32
#define VP8HIP_NBUF 4          // device job tables in rotation (a launch's table may still be read while the next one is staged)
#ifdef VP8_STAMPS
#define VP8HIP_SCHED_WORDS (16 + 16384 + 4 * 4096)     // + the diagnostic builds' log: four words per wave
#else
#define VP8HIP_SCHED_WORDS (16 + 16384)     // vp8_keyframe_kernel: two work counters, one arrival counter per SIMD of the device
#endif

// An IR slot: one frame's macroblock data in the DEVICE FORM of include/vp8_ir.h -- [pad 128][mbx: nmb x 128][blocks: up to
// nmb x 24 x 32][mvs: nmb x 64] in one device block (vp8hip_ctx::slot_block_dev + slot * slot_bytes) -- and, once mapped, pinned
// host staging of the same layout that a feeder fills and one copy sends.  The dense view (vp8hip_ir_map: what the oracle and
// the tests speak) is host memory only; vp8hip_ir_upload converts it on the host.
struct Slot {
    vp8ir_mbx *d_mbx; int16_t *d_blocks; vp8ir_mv *d_mvs;
    char *h_block;                                     // pinned staging, allocated on first map: same offsets as the device block
    vp8ir_frame_hdr *h_hdr; vp8ir_mbx *h_mbx; int16_t *h_blocks; vp8ir_mv *h_mvs;
    char *h_dense; vp8ir_mb *h_mbs; int16_t *h_coef;   // the dense view (pageable), allocated on first vp8hip_ir_map
    vp8ir_frame_hdr hdr_copy;                          // header as of the last upload / copy / entropy launch (host side, for job setup)
    size_t nblocks;                                    // blocks in the device stream, as far as the host knows (NBLOCKS_UNKNOWN: written on the device)
};
#define NBLOCKS_UNKNOWN ((size_t)-1)

// Tuning / test knobs, read from the environment by vp8hip_configure (never per launch):
//   VP8HIP_RECON=simt|wave     force one of the two kernel families whatever the launch's size
//   VP8HIP_SIMT_LGG=1..6       lane-per-row kernels: lanes per strand (log2); VP8HIP_SIMT_WAVES=n  at most n waves per launch
//   VP8HIP_WG_PER_CU, VP8HIP_XCU, VP8HIP_XCU_S, VP8HIP_XCU_NW, VP8HIP_RECON_NW, VP8HIP_LF_NW   wave-per-row family shapes
//   VP8HIP_EAGER_RASTER=1      large launches produce the raster form of their frames at once (default: when something asks for it)
//   VP8HIP_D2H_STREAMS=1..4    a batch download goes in that many pieces on streams of their own (default 2)
//   VP8HIP_DIRECT_DOWNLOAD=0   batch downloads of tiled frames go through the raster form in HBM and a copy (default: the tiled ->
//                              raster pass writes the page-locked destination itself); VP8HIP_DOWNLOAD_BLOCKS=n  its workgroups
//   VP8HIP_INTER_SPLIT=N       launches of up to N frames with inter frames among them run vp8_inter_mb_kernel first (default 384; 0:
//                              never).  It shortens a frame's critical path (1080p P frames, 1..16 per launch: recon 0.91 -> 0.46-0.56
//                              ms; 128: 1.56 -> 1.45) and costs throughput in launches that fill the chip anyway (512: 3.82 -> 4.03)
#define VP8HIP_TILE_FRONT 4096
struct Knobs {
    int recon_force;       // 0 automatic, 1 lane-per-row, 2 wave-per-row
    int md5_pack_from;     // VP8HIP_MD5_PACK_FROM: batches of this many tiled frames and more are hashed from a packed copy (vp8hip.hip: fetch_impl)
    int pred_tiles;        // VP8HIP_PRED_TILES: 1 (default) a large launch whose references are all there as tiles, and not all as raster frames,
                           // predicts from the tiles; 2: whenever all are there as tiles; 0: never (the raster form is made first)
    int inter_split, eager_raster, direct_download, download_blocks, d2h_prio, d2h_streams;
    int lgG, simt_waves, wg_per_cu, xcu, xcu_S, xcu_NW, recon_nw, lf_nw;
};

struct vp8hip_ctx {
    int device;
    Knobs knobs;
    hipStream_t stream;
    // timing events of the last VP8HIP_STATS_RING launches: [0..3] on the main stream around recon / loop filter /
    // extend, [4..5] around the tiled -> raster pass on whichever stream it ran
    hipEvent_t evr[VP8HIP_STATS_RING][6];
    bool evr_tiled[VP8HIP_STATS_RING]; vp8hip_stats evr_stats[VP8HIP_STATS_RING];
    long ncalls;
    hipEvent_t ev_jobs2[VP8HIP_NBUF];   // the job table staged in h_jobs2[k] has been copied
    int parity;                    // job table used by the next launch
    char err[256];
    // geometry
    int width, height;
    vp8ir_geom geom;
    DevGeom dg;
    int nmb;
    // pools.  A frame buffer has TWO forms on the device (vp8hip_launch.hip): the raster form -- the reference's YV12 layout with
    // its borders, fb[i] -- and the tiled form the lane-per-row kernels write -- macroblock-window tiles, fb_tiles[i], allocated
    // with the first large launch --, and fb_state[i] says which of them hold the frame.  Whoever needs a form the frame is not
    // in asks for it (vp8hip_need_raster): the conversion runs when something reads the frame as raster, not after every launch.
    std::vector<uint8_t *> fb, fb_tiles;
    std::vector<uint8_t> fb_state;
    std::vector<Slot> slots;
    uint8_t *fb_block; char *slot_block_dev;
    uint8_t *tile_block; size_t tile_frame;          // the tiled forms of all frame buffers (tile_frame bytes each) + the dummy tile
    uint8_t *tile_alloc;                             // ... as allocated: VP8HIP_TILE_FRONT bytes in front of tile_block (the tile-reading predictor's loads left of a tile row)
    DevJob *d_conv_jobs, *h_conv_jobs; int conv_cap; hipEvent_t ev_conv;    // job table of a tiled -> raster pass
    size_t slot_bytes, o_mbx, o_blocks, o_mvs, cap_blocks;            // slot layout; cap_blocks = nmb * 24 (pooled: 0)
    // vp8hip_configure_pooled: the slots have no block streams of their own; the device's entropy decoder takes the blocks' room out
    // of this pool, chunk_blocks blocks at a time (pool_chunks chunks + one that takes what no longer fits); *d_pool_ctr = chunks taken
    char *pool; unsigned int *d_pool_ctr; unsigned int pool_chunks, chunk_blocks;
    // job staging
    // (device tables and page-locked stagings in rotation: a caller may queue VP8HIP_NBUF launches before it has to wait for the
    // device to have taken the first one's table -- behind an entropy launch of a quarter of a second, say)
    DevJob *d_jobs2[VP8HIP_NBUF]; DevJob *h_jobs2[VP8HIP_NBUF]; DevJob *d_jobs; DevJob *h_jobs; int jobs_cap;   // d_jobs / h_jobs = those of the call
    // launch configuration
    int num_cu, max_lds;
    int recon_nw, lf_nw;
    size_t recon_lds, lf_lds;
    vp8hip_stats stats;
    // Small launches spread every frame pair over several CUs (vp8_recon_xcu_kernel / vp8_loopfilter_xcu_kernel): granule
    // buffers of the row-to-row hand-over, the launch counter that tags the granules, and the status word a kernel
    // sets (host-mapped memory) when a hand-over does not arrive
    unsigned long long *gran_recon, *gran_lf; size_t gran_recon_cap, gran_lf_cap;
    unsigned int epoch;
    int *h_status, *d_status;
    // batch download of whole frame buffers on a stream of its own (vp8hip_frames_download_async): PCIe is full duplex, the next
    // batch's uploads run beside it
    hipStream_t stream_d2h;
    uint8_t *d_i420; size_t i420_cap; hipEvent_t ev_pack;     // vp8hip_frames_fetch_i420_async: the batch as packed I420, before it leaves
    hipStream_t stream_d2h_more[3]; hipEvent_t ev_d2h_more[3];      // a batch download in up to four pieces on streams of their own (a copy engine each)
    hipEvent_t ev_d2h_from, ev_d2h_done;
    int d2h_first, d2h_count;      // frame buffers of the copy in flight (count 0: none)
    std::vector<uint8_t> d2h_mask; bool d2h_listed;     // ... of a fetch by list (vp8hip_frames_md5_list_async): a flag per frame buffer
    uint8_t *d_md5; int md5_cap;   // vp8hip_frames_fetch_async: the batch's digests on the device
    int *d_md5_idx, *h_md5_idx; int md5_idx_cap;    // vp8hip_frames_md5_list_async: which frame buffers
    size_t fb_stride;
    unsigned int *d_intra_flags; int intra_flags_cap;       // per job of a launch: the frame has intra macroblocks (vp8_inter_mb_kernel)
    // vp8hip_postproc: dither table (440 shorts), noise table (3072) and per-row noise phases (16384) on the device
    char *d_pp, *h_pp; bool pp_rv_loaded; hipEvent_t ev_pp;
    uint8_t *d_mfqe, *h_mfqe; int mfqe_cap; hipEvent_t ev_mfqe;     // vp8hip_mfqe: the macroblock classes of the frame
    // vp8hip_entropy_decode: the frames' descriptions, their bytes, per-frame scratch and status on the device; the stream the
    // launch runs on (its own: beside the pixel path of other slots) and the events that order it against the main stream
    // (descriptions and bytes in TWO sets: a launch's input is copied on a stream of its own, stream_h2d, while the launch before
    // and its frames' pixel path still run -- 3 to 5 GB per launch of 24,576 1080p frames; ev_ent_in[k]: set k's copies have landed,
    // ev_ent_out[k]: the kernel that read set k is done)
    char *d_ent_frames2[2], *d_ent_data2[2]; size_t ent_frames_cap2[2], ent_data_cap2[2]; int ent_set;
    hipStream_t stream_h2d; hipEvent_t ev_ent_in[2], ev_ent_out[2];
    unsigned int *d_ent_scratch, *d_ent_status; size_t ent_status_cap, ent_scratch_cap; int ent_last_count;
    struct { int count, set, np; const vp8hip_entropy_frame *frames; size_t data_bytes; } ent_staged;    // input sent, kernel not yet launched (count 0: none)
    bool ent_tables_loaded, ent_parts_off; int ent_lpw;
    int ent_resident[3];           // waves of vp8_entropy_kernel the device holds at once with 64 / 32 / 16 lanes carrying a frame (LDS)
    unsigned int *d_sched;         // vp8_keyframe_kernel's role / work counters
};

int vp8hip_fail(vp8hip_ctx *c, int code, const char *fmt, ...);
#define fail vp8hip_fail
#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, -1, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define FB_RASTER 1u            // fb_state bits: the raster form holds the frame (borders included) ...
#define FB_TILES  2u            // ... the tiled form does
// vp8hip_launch.hip: make sure the raster form of frame buffers first .. first + count - 1 holds their frames (a tiled -> raster
// pass + border extension on the context's stream for those that are only there as tiles)
// The raster pool itself is allocated when something first needs it (vp8hip_raster_pool; c->fb[i] are null until then): a
// pipeline whose frames are written as tiles, hashed as tiles and never read by coordinate never pays for it -- at 1080p 3.2 MB
// per frame buffer, a quarter of what a frame in flight costs.
int vp8hip_raster_pool(vp8hip_ctx *c);
int vp8hip_drop_staging(vp8hip_ctx *c);       // vp8hip.hip: frees the packed staging (1: freed, 0: there was none)
int vp8hip_need_raster(vp8hip_ctx *c, int first, int count);
int vp8hip_need_raster_list(vp8hip_ctx *c, const int *fbs, int n);
// vp8hip.hip
int vp8hip_check_status(vp8hip_ctx *c);       // after a stream synchronisation: did a kernel of the cross-CU family give up on a hand-over?
