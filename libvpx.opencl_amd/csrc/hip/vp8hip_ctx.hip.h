// Internal to libvp8hip.so: the context behind include/vp8hip.h's opaque handle, shared by the shim's translation units --
//   vp8hip.hip          context, pools, IR slots (upload / copy / fetch), frame buffers, statistics
//   vp8hip_launch.hip   vp8hip_decode: which kernels a launch runs, and the tiled -> raster pass behind the large ones
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
#define VP8HIP_NBUF 3          // tile sets / job tables in rotation (the tiled -> raster pass of a launch runs beside the next one)
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
//   VP8HIP_DETILE_STREAM=0     run the tiled -> raster pass on the main stream
//   VP8HIP_DETILE_BLOCKS=n     workgroups of that pass (default: two per CU)
//   VP8HIP_INTER_SPLIT=N       launches of up to N frames with inter frames among them run vp8_inter_mb_kernel first (default 384; 0:
//                              never).  It shortens a frame's critical path (1080p P frames, 1..16 per launch: recon 0.91 -> 0.46-0.56
//                              ms; 128: 1.56 -> 1.45) and costs throughput in launches that fill the chip anyway (512: 3.82 -> 4.03)
struct Knobs {
    int recon_force;       // 0 automatic, 1 lane-per-row, 2 wave-per-row
    int inter_split, detile_blocks;
    int lgG, simt_waves, wg_per_cu, xcu, xcu_S, xcu_NW, recon_nw, lf_nw, detile_stream;
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
    hipEvent_t ev_jobs;            // job table of the previous call has been copied
    // The tiled -> raster pass of a large launch is memory-bound while the kernel in front of it is bound by arithmetic, so it
    // runs on a second stream beside the NEXT launch.  VP8HIP_NBUF tile sets and device job tables rotate; any other use of the
    // frame buffers first joins the second stream.
    hipStream_t stream2;
    hipEvent_t ev_lf_done, ev_detile_done[VP8HIP_NBUF];
    bool detile_used[VP8HIP_NBUF], detile_pending;
    // a tiled -> raster pass not launched yet (launches with inter frames: it goes out beside the NEXT launch's
    // vp8_interframe_kernel, behind its prediction kernel, or at the next join)
    struct { bool valid; DevJob *jobs; int njobs, extend, par; hipEvent_t *ev; } deferred;
    hipEvent_t ev_recon_done;
    int parity, last_par;        // set used by the next lane-per-row launch / by the last one
    char err[256];
    // geometry
    int width, height;
    vp8ir_geom geom;
    DevGeom dg;
    int nmb;
    // pools
    std::vector<uint8_t *> fb;
    // which tiled -> raster pass writes a frame buffer's raster: passes are numbered as they are issued (detile_gen); passes up to
    // detile_joined have been waited for by the main stream.  A launch that reads reference frames only has to join if one of
    // them is still to be written by a pass it has not waited for
    std::vector<unsigned> fb_detile_gen;
    unsigned detile_gen, detile_joined;
    std::vector<Slot> slots;
    uint8_t *fb_block; char *slot_block_dev;
    uint8_t *tile_block[VP8HIP_NBUF]; size_t tile_cap[VP8HIP_NBUF];   // macroblock-window tiles of the lane-per-row kernels
    size_t slot_bytes, o_mbx, o_blocks, o_mvs, cap_blocks;            // slot layout; cap_blocks = nmb * 24
    // job staging
    DevJob *d_jobs2[VP8HIP_NBUF]; DevJob *d_jobs; DevJob *h_jobs; int jobs_cap;   // d_jobs = d_jobs2[parity of the call]
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
    hipEvent_t ev_d2h_from, ev_d2h_done;
    int d2h_first, d2h_count;      // frame buffers of the copy in flight (count 0: none)
    uint8_t *d_md5; int md5_cap;   // vp8hip_frames_fetch_async: the batch's digests on the device
    size_t fb_stride;
    unsigned int *d_intra_flags; int intra_flags_cap;       // per job of a launch: the frame has intra macroblocks (vp8_inter_mb_kernel)
    // vp8hip_postproc: dither table (440 shorts), noise table (3072) and per-row noise phases (16384) on the device
    char *d_pp, *h_pp; bool pp_rv_loaded; hipEvent_t ev_pp;
    uint8_t *d_mfqe, *h_mfqe; int mfqe_cap; hipEvent_t ev_mfqe;     // vp8hip_mfqe: the macroblock classes of the frame
    // vp8hip_entropy_decode: the frames' descriptions, their bytes, per-frame scratch and status on the device; the stream the
    // launch runs on (its own: beside the pixel path of other slots) and the events that order it against the main stream
    char *d_ent_frames, *d_ent_data; unsigned int *d_ent_scratch, *d_ent_status; size_t ent_frames_cap, ent_data_cap, ent_scratch_cap;
    bool ent_tables_loaded, ent_parts_off; int ent_lpw;
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

// vp8hip_launch.hip
int vp8hip_join_detile(vp8hip_ctx *c);        // the main stream waits for a tiled -> raster pass still running (or not launched yet)
// vp8hip.hip
int vp8hip_check_status(vp8hip_ctx *c);       // after a stream synchronisation: did a kernel of the cross-CU family give up on a hand-over?
