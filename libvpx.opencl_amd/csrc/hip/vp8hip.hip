// C-ABI shim of the gfx950 VP8 pixel path (include/vp8hip.h).  Owns the HIP stream, the device
// frame-buffer pool (the decoder's yv12_fb[] lives in HBM), the IR slots with their pinned host
// staging mirrors, and launches the three kernels.  No CPU fallback: every failure is reported.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "vp8hip.h"
#include "vp8_common.hip.h"

#define VP8HIP_STATS_RING 32
#define VP8HIP_NBUF 3          // scratch frame sets / job tables in rotation (see vp8hip_ctx)

extern "C" __global__ void vp8_recon_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_recon_xcu_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                                                int S, int *err);
extern "C" __global__ void vp8_loopfilter_xcu_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned long long *gran,
                                                     unsigned int epoch, int S, int *err);
extern "C" __global__ void vp8_recon_intra_kernel(const DevJob *jobs, int njobs, DevGeom g, const unsigned int *intra_flags);
extern "C" __global__ void vp8_recon_intra_xcu_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                                                      int S, int *err, const unsigned int *intra_flags);
extern "C" __global__ void vp8_inter_mb_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned int *intra_flags);
extern "C" __global__ void vp8_keyframe_kernel(const DevJob *jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy,
                                               unsigned int *sched, int nwaves);
extern "C" __global__ void vp8_interframe_kernel(const DevJob *jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy,
                                               unsigned int *sched, int nwaves);
extern "C" __global__ void vp8_inter_pred_kernel(const DevJob *jobs, int njobs, DevGeom g, int upf);
extern "C" __global__ void vp8_entropy_kernel(const vp8hip_entropy_frame *frames, int count, int lpw, const uint8_t *data, DevGeom g,
                                              size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbs, size_t o_coef, size_t o_mvs,
                                              int first_slot, unsigned int *scratch, unsigned int *status);
extern "C" size_t vp8_entropy_lds_bytes(int lpw);
typedef unsigned int ent_u32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void vp8_entropy_sparse_kernel(const vp8hip_entropy_frame *frames, int count, int lpw, const uint8_t *data, DevGeom g,
                                                     size_t data_bytes, unsigned int *scratch, unsigned int *status, ent_u32x4 *mbs,
                                                     ent_u32x4 *blocks, short *dcs, unsigned int *cursors, unsigned int cap_blocks,
                                                     unsigned int cap_dcs);
extern "C" __global__ void vp8_entropy_parts_kernel(const vp8hip_entropy_frame *frames, int count, int np, const uint8_t *data, DevGeom g,
                                                    size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbs, size_t o_coef,
                                                    int first_slot, unsigned int *scratch, unsigned int *status);
extern "C" __global__ void vp8_md5_kernel(const uint8_t *frames, size_t fstride, int count, DevGeom g, int w, int h, uint8_t *out);
#ifdef VP8_STAMPS
#define VP8HIP_SCHED_WORDS (16 + 16384 + 4 * 4096)     // + the diagnostic builds' log: four words per wave
#else
#define VP8HIP_SCHED_WORDS (16 + 16384)     // vp8_keyframe_kernel: two work counters, one arrival counter per SIMD of the device
#endif
extern "C" __global__ void vp8_recon_simt_kernel(const DevJob *jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, uint8_t *dummy);
extern "C" __global__ void vp8_loopfilter_simt_luma_kernel(const DevJob *jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster);
extern "C" __global__ void vp8_loopfilter_simt_chroma_kernel(const DevJob *jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster);
extern "C" __global__ void vp8_loopfilter_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_extend_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_detile_kernel(const DevJob *jobs, int njobs, DevGeom g, int extend);
extern "C" __global__ void vp8_detile_kf_kernel(const DevJob *jobs, int njobs, DevGeom g);
// vp8_postproc.hip
void vp8pp_down_and_across(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit);
void vp8pp_mb_across(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit);
void vp8pp_mb_down(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit, const short *rv);
void vp8pp_mfqe(hipStream_t st, const uint8_t *show, const uint8_t *prev, uint8_t *out, const DevGeom &g, const uint8_t *cls,
                int qcurr, int qprev);
void vp8pp_add_noise(hipStream_t st, uint8_t *plane, int stride, int rows, int cols, int clamp, const signed char *noise,
                     const uint8_t *row_offset);

// Sparse coefficient streams -> the dense coefficient array the kernels read (include/vp8_ir.h): one thread per 16 bytes of
// output -- half a block --, which entry of which stream it comes from (or none: zeros) follows from the macroblock's descriptor.
// (The descriptors arrive in the same staging buffer -- one host-to-device copy per frame -- and are put in place here too.)
__global__ void __launch_bounds__(256)
vp8_ir_expand_kernel(const vp8ir_mb *__restrict__ mbs, const int16_t *__restrict__ blocks, const int16_t *__restrict__ dcs,
                     vp8ir_mb *__restrict__ mbs_out, int16_t *__restrict__ coef, int nmb)
{
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int mb = (int)(gid / 50), ch = (int)(gid % 50), k = ch >> 1;
    if (mb >= nmb) return;
    const vp8ir_mb &m = mbs[mb];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (ch < 4) ((u32x4 *)(mbs_out + mb))[ch] = ((const u32x4 *)&m)[ch];
    u32x4 v = { 0, 0, 0, 0 };
    const int kind = vp8ir_block_kind(&m, k);
    if (kind) {
        int rank = 0;
        for (int j = 0; j < k; j++) rank += vp8ir_block_kind(&m, j) == kind;
        if (kind == 2) v = *(const u32x4 *)(blocks + ((size_t)m.sparse_first + rank) * 16 + (ch & 1) * 8);
        else if (!(ch & 1)) v.x = (unsigned short)dcs[(size_t)m.dc_first + rank];      // IR order: the DC is the block's first entry
    }
    *(u32x4 *)(coef + (size_t)mb * VP8IR_COEF_PER_MB + k * 16 + (ch & 1) * 8) = v;
}

// The same for `gridDim.y` frames whose sparse form was written on the device (vp8_entropy_sparse_kernel): descriptors frame after
// frame in sp_mbs, the streams in arenas shared by all of them (sparse_first / dc_first index the arenas); frame y -> slot first_slot + y.
// Two kernels: the slots' coefficient arrays are cleared (most of the dense form is zeros: plain 16-byte stores, a row of the grid
// per frame), then a thread per macroblock walks its 25 eobs and puts the coded entries in place.
__global__ void __launch_bounds__(256)
vp8_ir_clear_kernel(char *slot_base, size_t slot_bytes, size_t o_coef, int first_slot, size_t coef_bytes)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 *p = (u32x4 *)(slot_base + slot_bytes * (size_t)(first_slot + (int)blockIdx.y) + o_coef);
    const size_t n = coef_bytes / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (u32x4){ 0, 0, 0, 0 };
}
__global__ void __launch_bounds__(256)
vp8_ir_expand_batch_kernel(const vp8ir_mb *__restrict__ sp_mbs, const int16_t *__restrict__ blocks, const int16_t *__restrict__ dcs,
                           char *slot_base, size_t slot_bytes, size_t o_mbs, size_t o_coef, int first_slot, int nmb)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int mb = (int)(blockIdx.x * 256 + threadIdx.x);
    if (mb >= nmb) return;
    const vp8ir_mb *m = sp_mbs + (size_t)blockIdx.y * nmb + mb;
    char *slot = slot_base + slot_bytes * (size_t)(first_slot + (int)blockIdx.y);
    const u32x4 d0 = ((const u32x4 *)m)[0], d1 = ((const u32x4 *)m)[1], d2 = ((const u32x4 *)m)[2], d3 = ((const u32x4 *)m)[3];
    u32x4 *md = (u32x4 *)((vp8ir_mb *)(slot + o_mbs) + mb);
    md[0] = d0; md[1] = d1; md[2] = d2; md[3] = d3;
    const unsigned int ymode = d0.x & 255u, flags = d0.x >> 24;
    if (flags & VP8IR_MB_SKIP) return;
    const bool has_y2 = ymode != VP8IR_B_PRED && ymode != VP8IR_SPLITMV;
    // eobs: descriptor bytes 8..32
    const unsigned int ew[7] = { d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x };
    size_t b = d3.z, d = d3.w;                                            // sparse_first, dc_first
    int16_t *coef = (int16_t *)(slot + o_coef) + (size_t)mb * VP8IR_COEF_PER_MB;
    for (int k = 0; k < 25; k++) {
        if (k == 24 && !has_y2) break;
        const unsigned int e = (ew[k >> 2] >> (8 * (k & 3))) & 255u;
        if (e > 1) {
            const u32x4 *src = (const u32x4 *)(blocks + b * 16);
            ((u32x4 *)(coef + k * 16))[0] = src[0];
            ((u32x4 *)(coef + k * 16))[1] = src[1];
            b++;
        } else if (e == 1 && !(has_y2 && k < 16)) {
            coef[k * 16] = dcs[d];
            d++;
        }
    }
}

// Packed coefficients: the form vp8_keyframe_kernel / vp8_interframe_kernel read a slot in.  Of a macroblock's blocks 0..23 those a
// kernel FETCHES move to the front of its 800 bytes, in block order: the luma blocks with more than a DC (a lone DC comes out of the
// Y2 block, decodframe.c:262-296, or -- below -- with the descriptor); the chroma blocks with any coefficient.
// The Y2 block stays where it is (block 24); in a macroblock WITHOUT one, block 24's place holds the first coefficients of the
// sixteen luma blocks instead (eob == 1: the lone DC; else 0), which the luma wave gets with the macroblock's descriptor anyway:
// a lone DC costs two bytes there, not a 32-byte block and its two requests.  With a third of the blocks coded the dense form
// makes a kernel fetch four of every five 128-byte lines of the array; packed, it fetches what it uses.  In place, one thread per
// macroblock: a block only ever moves towards the front, past blocks that have moved already.  `unpack` restores the dense form
// (zeros where a block has no coefficients) for the kernels that read that.  slots: indices into the slot pool.
static __device__ __forceinline__ unsigned int vp8_stored_blocks(const vp8ir_mb &m, bool &lone_dcs)
{
    lone_dcs = false;
    if (m.flags & VP8IR_MB_SKIP) return 0;
    const bool has_y2 = m.y_mode != VP8IR_B_PRED && m.y_mode != VP8IR_SPLITMV;
    unsigned int mask = 0;
    for (int k = 0; k < 16; k++) mask |= (unsigned int)(m.eobs[k] >= 2) << k;
    for (int k = 16; k < 24; k++) mask |= (unsigned int)(m.eobs[k] >= 1) << k;
    lone_dcs = !has_y2;
    return mask;
}
__global__ void __launch_bounds__(256)
vp8_ir_pack_kernel(char *slot_base, size_t slot_bytes, size_t o_mbs, size_t o_coef, const int *__restrict__ slots, int nslots, int nmb, int unpack)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int si = (int)(gid / nmb), mb = (int)(gid - (long)si * nmb);
    if (si >= nslots) return;
    char *slot = slot_base + slot_bytes * (size_t)slots[si];
    const vp8ir_mb &m = ((const vp8ir_mb *)(slot + o_mbs))[mb];
    int16_t *c16 = (int16_t *)(slot + o_coef) + (size_t)mb * VP8IR_COEF_PER_MB;
    u32x4 *cf = (u32x4 *)c16;      // two per block
    bool lone;
    const unsigned int mask = vp8_stored_blocks(m, lone);
    if (!unpack) {
        int16_t dc[16];
        if (lone)
            for (int k = 0; k < 16; k++) dc[k] = m.eobs[k] == 1 ? c16[k * 16] : (int16_t)0;
        int r = 0;
        for (int k = 0; k < 24; k++) {
            if (!((mask >> k) & 1)) continue;
            if (r != k) { const u32x4 a = cf[2 * k], b = cf[2 * k + 1]; cf[2 * r] = a; cf[2 * r + 1] = b; }
            r++;
        }
        if (lone)
            for (int k = 0; k < 16; k++) c16[384 + k] = dc[k];
    } else {
        if (m.flags & VP8IR_MB_SKIP) return;
        int16_t dc[16];
        if (lone)
            for (int k = 0; k < 16; k++) dc[k] = c16[384 + k];
        int r = __builtin_popcount(mask);
        for (int k = 23; k >= 0; k--) {
            if (!((mask >> k) & 1)) continue;
            r--;
            if (r != k) { const u32x4 a = cf[2 * r], b = cf[2 * r + 1]; cf[2 * k] = a; cf[2 * k + 1] = b; }
        }
        const u32x4 z = { 0, 0, 0, 0 };
        for (int k = 0; k < 24; k++)
            if (!((mask >> k) & 1)) { cf[2 * k] = z; cf[2 * k + 1] = z; }
        if (lone) {
            for (int k = 0; k < 16; k++)
                if (m.eobs[k] == 1) c16[k * 16] = dc[k];
            cf[48] = z; cf[49] = z;
        }
    }
}

static char g_create_error[256] = "";

struct Slot {
    // device
    vp8ir_frame_hdr *d_hdr; vp8ir_mb *d_mbs; int16_t *d_coef; vp8ir_mv *d_mvs;
    // pinned host mirror
    vp8ir_frame_hdr *h_hdr; vp8ir_mb *h_mbs; int16_t *h_coef; vp8ir_mv *h_mvs;
    vp8ir_frame_hdr hdr_copy;      // header as of the last upload / copy (host side, for job setup)
    char *h_block;                 // pinned mirror, allocated on first vp8hip_ir_map
    char *d_sparse;                // device staging of a sparse upload: descriptors, blocks, DCs; allocated on first vp8hip_ir_upload_sparse
    int16_t *h_dcs;                // pinned staging of the DC stream while the feeder writes it (behind the dense mirror in h_block)
    bool packed;                   // the coefficients on the device are in vp8_keyframe_kernel's packed form (vp8_ir_pack_kernel)
};

// Tuning / test knobs, read from the environment by vp8hip_configure (never per launch):
//   VP8HIP_RECON=simt|wave     force one of the two kernel families (simt: key-frame launches only)
//   VP8HIP_LF_RASTER=0         lane-per-row family: finish with the tiled -> raster pass instead of letting the loop filter write raster
//   VP8HIP_SIMT_LGG=1..6       lanes per strand (log2); VP8HIP_SIMT_WAVES=n  at most n waves per launch
//   VP8HIP_WG_PER_CU, VP8HIP_XCU, VP8HIP_XCU_S, VP8HIP_XCU_NW, VP8HIP_RECON_NW, VP8HIP_LF_NW   wave-per-row family shapes
//   VP8HIP_DETILE_STREAM=0 / VP8HIP_DETILE_DEFER=0   run the tiled -> raster pass on the main stream / at once
struct Knobs {
    int recon_force;       // 0 automatic, 1 lane-per-row, 2 wave-per-row
    int inter_split;  // VP8HIP_INTER_SPLIT=N: launches of up to N frames with inter frames among them run vp8_inter_mb_kernel first
                      // (default 384; 0: never).  It shortens a frame's critical path (1080p P frames, 1..16 per launch: recon 0.91 ->
                      // 0.46-0.56 ms; 128: 1.56 -> 1.45) and costs throughput in launches that fill the chip anyway (512 frames:
                      // 3.82 -> 4.03 ms, 1024: 7.2 -> 8.2)
    int inter_tiled;  // VP8HIP_INTER_TILED=N: launches of N or more frames with inter frames among them hand over to the lane-per-row loop
                      // filter through the tiled scratch frames (default 640; 0: never).  1080p P frames, recon + loop filter per launch:
                      // 512 frames 6.5 -> 7.4 ms, 768: 11.5 -> 10.6, 1024: 13.2 -> 10.9, 8192: 101.8 -> 73.2 (the recon's 4-byte stores
                      // complete 128-byte tile lines, which they never do in a raster frame)
    // VP8HIP_FUSED=0: all-key-frame launches of the lane-per-row family run reconstruction and loop filter as two kernels with the
    // tiled scratch frames between them (the round-1/2 pipeline) instead of vp8_keyframe_simt_kernel
    int fused;
    // VP8HIP_INTER_FUSED=N: launches of N or more frames with inter frames among them, both stages wanted, go the key frames' way --
    // vp8_inter_pred_kernel (every inter macroblock's prediction, order-free) + vp8_interframe_kernel (residual + loop filter, one
    // macroblock row per lane) -- instead of the wave-per-row recon + the lane-per-row loop filter.  Default -1: launches of more
    // than two frames per CU, as for key frames; 0: never.  VP8HIP_RECON=simt forces it at every size.
    int inter_fused;
    int detile_blocks;     // VP8HIP_DETILE_BLOCKS=n: workgroups of the key-frame kernel's tiled -> raster pass (default: two per CU)
    int lf_raster, lgG, simt_waves, wg_per_cu, xcu, xcu_S, xcu_NW, recon_nw, lf_nw, detile_stream, detile_defer;
};
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
static void read_knobs(Knobs &k)
{
    const char *e = getenv("VP8HIP_RECON");
    k.recon_force = !e ? 0 : !strcmp(e, "simt") ? 1 : !strcmp(e, "wave") ? 2 : 0;
    k.lf_raster = env_int("VP8HIP_LF_RASTER", 1) != 0;
    k.lgG = env_int("VP8HIP_SIMT_LGG", 0);
    k.simt_waves = env_int("VP8HIP_SIMT_WAVES", 0);
    k.wg_per_cu = env_int("VP8HIP_WG_PER_CU", 1);
    k.xcu = env_int("VP8HIP_XCU", 1) != 0;
    k.xcu_S = env_int("VP8HIP_XCU_S", 0);
    k.fused = env_int("VP8HIP_FUSED", 1) != 0;
    k.detile_blocks = env_int("VP8HIP_DETILE_BLOCKS", 0);
    k.inter_split = env_int("VP8HIP_INTER_SPLIT", 384);
    k.inter_tiled = env_int("VP8HIP_INTER_TILED", 640);
    k.inter_fused = env_int("VP8HIP_INTER_FUSED", -1);
    k.xcu_NW = env_int("VP8HIP_XCU_NW", 0);
    k.recon_nw = env_int("VP8HIP_RECON_NW", 0);
    k.lf_nw = env_int("VP8HIP_LF_NW", 0);
    k.detile_stream = env_int("VP8HIP_DETILE_STREAM", 1) != 0;
    k.detile_defer = env_int("VP8HIP_DETILE_DEFER", 1) != 0;
}

__device__ unsigned int vp8_gran_broken;      // see gran_wait (vp8_common.hip.h)

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
    // The tiled -> raster pass of the lane-per-row pipeline is memory-bound while recon and loop filter are
    // VALU-bound, so it runs on a second stream and overlaps the NEXT launch's recon.  Two scratch frame sets
    // and two device job tables alternate; any other use of the frame buffers first joins the second stream.
    hipStream_t stream2;
    hipEvent_t ev_lf_done, ev_detile_done[VP8HIP_NBUF];
    bool detile_used[VP8HIP_NBUF], detile_pending;
    // a tiled -> raster pass not launched yet: it goes out beside the NEXT launch's loop filter (or at the next join)
    struct { bool valid, kf; DevJob *jobs; int njobs, extend, par; hipEvent_t *ev; } deferred;
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
    uint8_t *tile_block[VP8HIP_NBUF]; size_t tile_cap[VP8HIP_NBUF];   // macroblock-tiled scratch frames of the lane-per-row pipeline
    size_t slot_bytes, o_mbs, o_coef, o_mvs;
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
    hipStream_t stream3; hipEvent_t ev_split_from, ev_split_done;     // chroma half of the split lane-per-row loop filter
    // vp8hip_postproc: dither table (440 shorts), noise table (3072) and per-row noise phases (16384) on the device
    char *d_pp, *h_pp; bool pp_rv_loaded; hipEvent_t ev_pp;
    uint8_t *d_mfqe, *h_mfqe; int mfqe_cap; hipEvent_t ev_mfqe;     // vp8hip_mfqe: the macroblock classes of the frame
    // vp8hip_entropy_decode: the frames' descriptions, their bytes, per-frame scratch and status on the device
    char *d_ent_frames, *d_ent_data; unsigned int *d_ent_scratch, *d_ent_status; size_t ent_frames_cap, ent_data_cap, ent_scratch_cap;
    bool ent_tables_loaded, ent_parts_off, ent_last_sparse; int ent_lpw;
    // vp8hip_entropy_decode_sparse: descriptors of the launch's frames, the two arenas, cursors; the frames' headers for vp8hip_ir_expand
    char *d_sp_mbs, *d_sp_blocks, *d_sp_dcs; unsigned int *d_sp_cursors; size_t sp_mbs_cap, sp_blocks_cap, sp_dcs_cap, sp_blocks_use, sp_dcs_use; int sp_count;
    std::vector<vp8ir_frame_hdr> sp_hdrs;
    unsigned int *d_sched;         // vp8_keyframe_kernel's role / work counters
    int *h_pack, *d_pack; int pack_cap;      // slots whose coefficients a launch has to pack / unpack first
};

static int fail(vp8hip_ctx *c, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_error, 256, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, -1, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *vp8hip_last_error(const vp8hip_ctx *ctx) { return ctx ? ctx->err : g_create_error; }

static void free_pools(vp8hip_ctx *c)
{
    if (c->fb_block) (void)hipFree(c->fb_block);
    for (int k = 0; k < VP8HIP_NBUF; k++) { if (c->tile_block[k]) (void)hipFree(c->tile_block[k]); c->tile_block[k] = nullptr; c->tile_cap[k] = 0; }
    if (c->slot_block_dev) (void)hipFree(c->slot_block_dev);
    if (c->gran_recon) (void)hipFree(c->gran_recon);
    if (c->gran_lf) (void)hipFree(c->gran_lf);
    if (c->d_intra_flags) (void)hipFree(c->d_intra_flags);
    c->d_intra_flags = nullptr; c->intra_flags_cap = 0;
    c->gran_recon = c->gran_lf = nullptr; c->gran_recon_cap = c->gran_lf_cap = 0;
    for (Slot &s : c->slots) {
        if (s.h_block) (void)hipHostFree(s.h_block);
        if (s.d_sparse) (void)hipFree(s.d_sparse);
    }
    c->fb_block = nullptr; c->slot_block_dev = nullptr;
    for (int k = 0; k < VP8HIP_NBUF; k++) { c->tile_block[k] = nullptr; c->tile_cap[k] = 0; }
    c->fb.clear(); c->slots.clear();
}

static void destroy_events(vp8hip_ctx *c)
{
    for (int r = 0; r < VP8HIP_STATS_RING; r++) for (int i = 0; i < 6; i++) if (c->evr[r][i]) (void)hipEventDestroy(c->evr[r][i]);
    if (c->ev_jobs) (void)hipEventDestroy(c->ev_jobs);
    if (c->ev_lf_done) (void)hipEventDestroy(c->ev_lf_done);
    if (c->ev_recon_done) (void)hipEventDestroy(c->ev_recon_done);
    for (int k = 0; k < VP8HIP_NBUF; k++) if (c->ev_detile_done[k]) (void)hipEventDestroy(c->ev_detile_done[k]);
}

extern "C" int vp8hip_create(int device, vp8hip_ctx **out)
{
    if (!out) return fail(nullptr, -2, "vp8hip_create: null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, -1, "no HIP device available (%s): the VP8 pixel path has no CPU fallback",
                    hipGetErrorString(e));
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) return fail(nullptr, -2, "device %d out of range (%d devices)", device, ndev);
    if ((e = hipSetDevice(device)) != hipSuccess)
        return fail(nullptr, -1, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(nullptr, -1, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(nullptr, -1, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
                    prop.gcnArchName);
    vp8hip_ctx *c = new vp8hip_ctx();
    memset(c->err, 0, sizeof c->err);
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->max_lds = 160 * 1024;
    c->fb_block = nullptr; c->slot_block_dev = nullptr;
    read_knobs(c->knobs);
    for (int k = 0; k < VP8HIP_NBUF; k++) { c->tile_block[k] = nullptr; c->tile_cap[k] = 0; }
    c->d_jobs = nullptr; c->h_jobs = nullptr; c->jobs_cap = 0;
    for (int k = 0; k < VP8HIP_NBUF; k++) c->d_jobs2[k] = nullptr;
    for (int k = 0; k < VP8HIP_NBUF; k++) c->detile_used[k] = false;
    c->detile_pending = false; c->parity = 0; c->last_par = 0;
    c->detile_gen = c->detile_joined = 0;
    c->d_md5 = nullptr; c->md5_cap = 0;
    c->deferred.valid = false;
    c->width = c->height = 0;
    c->ncalls = 0;
    c->gran_recon = c->gran_lf = nullptr; c->gran_recon_cap = c->gran_lf_cap = 0; c->epoch = 0;
    c->h_status = c->d_status = nullptr;
    memset(&c->stats, 0, sizeof c->stats);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(nullptr, -1, "hipStreamCreate: %s", hipGetErrorString(e));
        delete c;
        return -1;
    }
    c->stream2 = nullptr;      // created by the first launch that wants it
    c->stream_d2h = nullptr; c->ev_d2h_from = c->ev_d2h_done = nullptr; c->d2h_first = c->d2h_count = 0; c->fb_stride = 0;
    // events: every creation is checked; on failure whatever exists is destroyed again (null handles are skipped)
    for (int r = 0; r < VP8HIP_STATS_RING; r++) for (int i = 0; i < 6; i++) c->evr[r][i] = nullptr;
    c->ev_jobs = c->ev_lf_done = c->ev_recon_done = nullptr;
    for (int k = 0; k < VP8HIP_NBUF; k++) c->ev_detile_done[k] = nullptr;
    e = hipSuccess;
    for (int r = 0; r < VP8HIP_STATS_RING && e == hipSuccess; r++)
        for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&c->evr[r][i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_jobs, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_lf_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_recon_done, hipEventDisableTiming);
    for (int k = 0; k < VP8HIP_NBUF && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->ev_detile_done[k], hipEventDisableTiming);
    const char *what = "hipEventCreate";
    const void *big_lds[6] = { (const void *)vp8_recon_kernel, (const void *)vp8_recon_xcu_kernel,
                               (const void *)vp8_loopfilter_xcu_kernel, (const void *)vp8_loopfilter_kernel,
                               (const void *)vp8_recon_intra_kernel, (const void *)vp8_recon_intra_xcu_kernel };
    for (int i = 0; i < 6 && e == hipSuccess; i++) {
        what = "hipFuncSetAttribute(max dynamic LDS)";
        e = hipFuncSetAttribute(big_lds[i], hipFuncAttributeMaxDynamicSharedMemorySize, c->max_lds);
    }
    if (e != hipSuccess) {
        fail(nullptr, -1, "%s: %s", what, hipGetErrorString(e));
        destroy_events(c);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return -1;
    }
    *out = c;
    return 0;
}

static int join_detile(vp8hip_ctx *c);

extern "C" void vp8hip_destroy(vp8hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)join_detile(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    free_pools(c);
    for (int k = 0; k < VP8HIP_NBUF; k++) if (c->d_jobs2[k]) (void)hipFree(c->d_jobs2[k]);
    if (c->h_jobs) (void)hipHostFree(c->h_jobs);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->d_sched) (void)hipFree(c->d_sched);
    // the post-processing tables outlive reconfigurations: the caller's noise state does too (vp8/common/postproc.c keeps
    // postproc_state.noise across vp8_alloc_frame_buffers) and only sends the noise table again when q changes
    if (c->d_pp) (void)hipFree(c->d_pp);
    if (c->d_md5) (void)hipFree(c->d_md5);
    if (c->h_pp) (void)hipHostFree(c->h_pp);
    if (c->ev_pp) (void)hipEventDestroy(c->ev_pp);
    if (c->d_sp_mbs) (void)hipFree(c->d_sp_mbs);
    if (c->d_sp_blocks) (void)hipFree(c->d_sp_blocks);
    if (c->d_sp_dcs) (void)hipFree(c->d_sp_dcs);
    if (c->d_sp_cursors) (void)hipFree(c->d_sp_cursors);
    if (c->d_ent_frames) (void)hipFree(c->d_ent_frames);
    if (c->d_ent_data) (void)hipFree(c->d_ent_data);
    if (c->d_ent_scratch) (void)hipFree(c->d_ent_scratch);
    if (c->d_ent_status) (void)hipFree(c->d_ent_status);
    if (c->d_mfqe) (void)hipFree(c->d_mfqe);
    if (c->h_mfqe) (void)hipHostFree(c->h_mfqe);
    if (c->ev_mfqe) (void)hipEventDestroy(c->ev_mfqe);
    if (c->d_pack) (void)hipFree(c->d_pack);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    destroy_events(c);
    if (c->stream_d2h) { (void)hipStreamSynchronize(c->stream_d2h); (void)hipStreamDestroy(c->stream_d2h); }
    if (c->ev_d2h_from) (void)hipEventDestroy(c->ev_d2h_from);
    if (c->ev_d2h_done) (void)hipEventDestroy(c->ev_d2h_done);
    (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) {
        (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3);
        (void)hipEventDestroy(c->ev_split_from); (void)hipEventDestroy(c->ev_split_done);
    }
    delete c;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// per-wave LDS footprints; must match the kernels (WaveLds 2080 B + line slot, LfWaveLds 768 B)
static size_t recon_lds_bytes(int nw, int aligned_w) { return 1024 + (size_t)nw * 2 * (2080 + 2 * aligned_w + 96); }   // two frames per wave
static size_t lf_lds_bytes(int nw) { return 256 + (size_t)nw * 2 * 4096; }   // two frames per wave

static int configure_pools(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots);

extern "C" int vp8hip_configure(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots)
{
    const int rc = configure_pools(c, width, height, num_fb, num_slots);
    if (rc && c) {               // a failed (re)configuration leaves an UNconfigured context, not a half-allocated one
        free_pools(c);
        c->width = c->height = 0;
    }
    return rc;
}

static int configure_pools(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots)
{
    if (!c) return -2;
    if (width <= 0 || height <= 0 || width > 16383 || height > 16383 || num_fb < 1 || num_slots < 1)
        return fail(c, -2, "vp8hip_configure: bad arguments %dx%d fb=%d slots=%d", width, height, num_fb, num_slots);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->width && join_detile(c)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    c->detile_pending = false;
    for (int k = 0; k < VP8HIP_NBUF; k++) c->detile_used[k] = false;
    free_pools(c);
    read_knobs(c->knobs);
    c->width = width; c->height = height;
    vp8ir_geom_init(&c->geom, width, height);
    const vp8ir_geom &g = c->geom;
    c->dg.mb_cols = g.aligned_w / 16; c->dg.mb_rows = g.aligned_h / 16;
    c->dg.aligned_w = g.aligned_w; c->dg.aligned_h = g.aligned_h;
    c->dg.y_stride = g.y_stride; c->dg.uv_stride = g.uv_stride;
    c->dg.y_off = g.y_off; c->dg.u_off = g.u_off; c->dg.v_off = g.v_off;
    c->nmb = c->dg.mb_cols * c->dg.mb_rows;
    if (c->dg.mb_cols > 65535) return fail(c, -2, "frame too wide");

    // waves per workgroup: one wave per MB row in flight; as many as LDS allows, at most 16, and
    // no more than the frame has rows (rounded up to 2, the minimum the line-buffer ring needs)
    int nw = 12;       // __launch_bounds__(768) in vp8_recon.hip
    while (nw > 2 && (recon_lds_bytes(nw, g.aligned_w) > (size_t)c->max_lds)) nw -= 2;
    if (recon_lds_bytes(nw, g.aligned_w) > (size_t)c->max_lds)
        return fail(c, -2, "frame width %d needs more LDS than a CU has", width);
    while (nw > 2 && nw / 2 >= c->dg.mb_rows) nw /= 2;
    if (c->knobs.recon_nw >= 2 && c->knobs.recon_nw <= nw) nw = c->knobs.recon_nw;
    c->recon_nw = nw; c->recon_lds = recon_lds_bytes(nw, g.aligned_w);
    int lnw = 16;
    while (lnw > 2 && lnw / 2 >= c->dg.mb_rows) lnw /= 2;
    if (c->knobs.lf_nw >= 2 && c->knobs.lf_nw <= 16) lnw = c->knobs.lf_nw;
    c->lf_nw = lnw; c->lf_lds = lf_lds_bytes(lnw);

    // frame buffers: one block, each buffer 256-B aligned
    const size_t fbsz = align_up((size_t)g.frame_size, 256);
    HIPCHK(c, hipMalloc((void **)&c->fb_block, fbsz * num_fb));
    HIPCHK(c, hipMemsetAsync(c->fb_block, 0, fbsz * num_fb, c->stream));
    for (int i = 0; i < num_fb; i++) c->fb.push_back(c->fb_block + fbsz * i);
    c->fb_detile_gen.assign((size_t)num_fb, 0u); c->detile_gen = c->detile_joined = 0;
    c->fb_stride = fbsz;
    if (c->stream_d2h) HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
    c->d2h_count = 0;

    // IR slots
    const size_t o_mbs = 64, o_coef = o_mbs + align_up((size_t)c->nmb * sizeof(vp8ir_mb), 256);
    const size_t o_mvs = o_coef + align_up((size_t)c->nmb * VP8IR_COEF_PER_MB * sizeof(int16_t), 256);
    const size_t slotsz = align_up(o_mvs + (size_t)c->nmb * 16 * sizeof(vp8ir_mv), 256);
    HIPCHK(c, hipMalloc((void **)&c->slot_block_dev, slotsz * num_slots + 4096));   // + room for prefetches past the last macroblock
    c->slot_bytes = slotsz; c->o_mbs = o_mbs; c->o_coef = o_coef; c->o_mvs = o_mvs;
    c->slots.resize(num_slots);
    for (int i = 0; i < num_slots; i++) {
        char *d = c->slot_block_dev + slotsz * i;
        Slot &s = c->slots[i];
        s.d_hdr = (vp8ir_frame_hdr *)d; s.d_mbs = (vp8ir_mb *)(d + o_mbs);
        s.d_coef = (int16_t *)(d + o_coef); s.d_mvs = (vp8ir_mv *)(d + o_mvs);
        s.h_block = nullptr; s.h_hdr = nullptr; s.h_mbs = nullptr; s.h_coef = nullptr; s.h_mvs = nullptr;
        s.d_sparse = nullptr; s.h_dcs = nullptr; s.packed = false;
        memset(&s.hdr_copy, 0, sizeof s.hdr_copy);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int vp8hip_geometry(const vp8hip_ctx *c, vp8ir_geom *g)
{
    if (!c || !g || !c->width) return -2;
    *g = c->geom;
    return 0;
}

extern "C" int vp8hip_ir_map(vp8hip_ctx *c, int slot, vp8ir_frame_hdr **hdr, vp8ir_mb **mbs, int16_t **coef,
                             vp8ir_mv **mvs)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_map: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.h_block) {   // pinned staging is created on first use: device-only slots cost no host memory
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipHostMalloc((void **)&s.h_block, c->slot_bytes + (size_t)c->nmb * 50, hipHostMallocDefault));
        memset(s.h_block, 0, c->slot_bytes);
        s.h_dcs = (int16_t *)(s.h_block + c->slot_bytes);
        s.h_hdr = (vp8ir_frame_hdr *)s.h_block; s.h_mbs = (vp8ir_mb *)(s.h_block + c->o_mbs);
        s.h_coef = (int16_t *)(s.h_block + c->o_coef); s.h_mvs = (vp8ir_mv *)(s.h_block + c->o_mvs);
    }
    if (hdr) *hdr = s.h_hdr;
    if (mbs) *mbs = s.h_mbs;
    if (coef) *coef = s.h_coef;
    if (mvs) *mvs = s.h_mvs;
    return 0;
}

extern "C" int vp8hip_ir_upload(vp8hip_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_upload: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.h_block) return fail(c, -2, "vp8hip_ir_upload: slot %d was never mapped", slot);
    const vp8ir_frame_hdr &h = *s.h_hdr;
    if (h.mb_cols != c->dg.mb_cols || h.mb_rows != c->dg.mb_rows)
        return fail(c, -2, "vp8hip_ir_upload: header is %dx%d MBs, context configured for %dx%d", h.mb_cols,
                    h.mb_rows, c->dg.mb_cols, c->dg.mb_rows);
    HIPCHK(c, hipSetDevice(c->device));
    s.hdr_copy = h;
    s.packed = false;
    HIPCHK(c, hipMemcpyAsync(s.d_mbs, s.h_mbs, (size_t)c->nmb * sizeof(vp8ir_mb), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s.d_coef, s.h_coef, (size_t)c->nmb * VP8IR_COEF_PER_MB * sizeof(int16_t),
                             hipMemcpyHostToDevice, c->stream));
    if (h.frame_type != 0)
        HIPCHK(c, hipMemcpyAsync(s.d_mvs, s.h_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyHostToDevice,
                                 c->stream));
    return 0;
}

extern "C" int vp8hip_ir_map_sparse(vp8hip_ctx *c, int slot, vp8ir_frame_hdr **hdr, vp8ir_mb **mbs, int16_t **blocks,
                                    size_t *cap_blocks, int16_t **dcs, vp8ir_mv **mvs)
{
    int16_t *coef = nullptr;
    if (vp8hip_ir_map(c, slot, hdr, mbs, &coef, mvs)) return -2;
    if (blocks) *blocks = coef;                     // the block stream is staged where the dense mirror would be: never both at once
    if (cap_blocks) *cap_blocks = (size_t)c->nmb * 25;
    if (dcs) *dcs = c->slots[slot].h_dcs;
    return 0;
}

extern "C" int vp8hip_ir_upload_sparse(vp8hip_ctx *c, int slot, size_t nblocks, size_t ndcs)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_upload_sparse: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.h_block) return fail(c, -2, "vp8hip_ir_upload_sparse: slot %d was never mapped", slot);
    const vp8ir_frame_hdr &h = *s.h_hdr;
    if (h.mb_cols != c->dg.mb_cols || h.mb_rows != c->dg.mb_rows)
        return fail(c, -2, "vp8hip_ir_upload_sparse: header is %dx%d MBs, context configured for %dx%d", h.mb_cols,
                    h.mb_rows, c->dg.mb_cols, c->dg.mb_rows);
    if (nblocks + ndcs > (size_t)c->nmb * 25) return fail(c, -2, "vp8hip_ir_upload_sparse: %zu blocks + %zu DCs for %d macroblocks", nblocks, ndcs, c->nmb);
    HIPCHK(c, hipSetDevice(c->device));
    // One copy per frame: in the pinned mirror the descriptors are followed by the coefficient staging (c->o_coef), where the
    // feeder wrote the blocks; the DCs, written elsewhere because nobody knew where the blocks would end, are moved up behind them.
    const size_t o_blocks = c->o_coef - c->o_mbs, o_dcs = o_blocks + nblocks * 32, used = o_dcs + ((ndcs * 2 + 15) & ~(size_t)15);
    if (!s.d_sparse) HIPCHK(c, hipMalloc((void **)&s.d_sparse, o_blocks + (size_t)c->nmb * 25 * 32 + 64));
    memcpy((char *)s.h_mbs + o_dcs, s.h_dcs, ndcs * 2);
    const int16_t *d_blocks = (const int16_t *)(s.d_sparse + o_blocks), *d_dcs = (const int16_t *)(s.d_sparse + o_dcs);
    s.hdr_copy = h;
    s.packed = false;
    HIPCHK(c, hipMemcpyAsync(s.d_sparse, s.h_mbs, used, hipMemcpyHostToDevice, c->stream));
    if (h.frame_type != 0)
        HIPCHK(c, hipMemcpyAsync(s.d_mvs, s.h_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyHostToDevice, c->stream));
    const long chunks = (long)c->nmb * 50;
    hipLaunchKernelGGL(vp8_ir_expand_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, c->stream,
                       (const vp8ir_mb *)s.d_sparse, d_blocks, d_dcs, s.d_mbs, s.d_coef, c->nmb);
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int vp8hip_ir_copy(vp8hip_ctx *c, int dst, int src)
{
    if (!c || dst < 0 || src < 0 || dst >= (int)c->slots.size() || src >= (int)c->slots.size())
        return fail(c, -2, "vp8hip_ir_copy: bad slots %d <- %d", dst, src);
    if (dst == src) return 0;
    Slot &d = c->slots[dst], &s = c->slots[src];
    HIPCHK(c, hipSetDevice(c->device));
    d.hdr_copy = s.hdr_copy;
    d.packed = s.packed;
    HIPCHK(c, hipMemcpyAsync(d.d_mbs, s.d_mbs, (size_t)c->nmb * sizeof(vp8ir_mb), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d.d_coef, s.d_coef, (size_t)c->nmb * VP8IR_COEF_PER_MB * sizeof(int16_t),
                             hipMemcpyDeviceToDevice, c->stream));
    if (s.hdr_copy.frame_type != 0)
        HIPCHK(c, hipMemcpyAsync(d.d_mvs, s.d_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyDeviceToDevice,
                                 c->stream));
    return 0;
}

// Make the main stream wait for a tiled -> raster pass still running on the second stream.  Every entry point
// that reads or writes frame buffers (other than another lane-per-row launch, which is ordered behind it on
// the second stream anyway) calls this first.
// Launch the deferred tiled -> raster pass on the second stream, behind `after` (an event on the main stream).
// the tiled -> raster pass (+ border extension) of a lane-per-row launch: vp8_detile_kernel for the two-kernel pipeline's tiles,
// vp8_detile_kf_kernel + vp8_extend_kernel for the key-frame kernel's
static int launch_detile(vp8hip_ctx *c, hipStream_t st, DevJob *jobs, int njobs, int extend, bool kf)
{
    if (kf) {
        // Two workgroups per CU, each looping over macroblock rows: the pass runs beside the next launch's vp8_keyframe_kernel
        // (its waves need 16 registers: they fit in the gap two of that kernel's waves leave on a SIMD) and is to trickle -- the
        // pair is bound by HBM bandwidth when the pass goes at full speed, and the key-frame kernel then loses more than the pass
        // gains.  8192 1080p frames per launch, ms per step: 2 per CU 45.7-47.1, 4 per CU 49.2-51.5, all at once 49.9-50.2, 1 per
        // CU 65 (the pass becomes the longer one)
        long units = (long)c->dg.mb_rows * njobs;
        const int cap = c->knobs.detile_blocks > 0 ? c->knobs.detile_blocks : 2 * c->num_cu;
        if (units > cap) units = cap;
        hipLaunchKernelGGL(vp8_detile_kf_kernel, dim3((unsigned)units), dim3(256), 0, st, (const DevJob *)jobs, njobs, c->dg);
        if (extend) {
            int bx = (c->geom.aligned_h + 64) / 4;
            if (bx < 1) bx = 1;
            if (bx > 64) bx = 64;
            hipLaunchKernelGGL(vp8_extend_kernel, dim3(bx, njobs), dim3(256), 0, st, (const DevJob *)jobs, njobs, c->dg);
        }
    } else
        hipLaunchKernelGGL(vp8_detile_kernel, dim3(c->dg.mb_rows, njobs), dim3(256), 0, st, (const DevJob *)jobs, njobs, c->dg, extend);
    HIPCHK(c, hipGetLastError());
    return 0;
}
static int launch_deferred(vp8hip_ctx *c, hipEvent_t after)
{
    if (!c->deferred.valid) return 0;
    HIPCHK(c, hipStreamWaitEvent(c->stream2, after, 0));
    HIPCHK(c, hipEventRecord(c->deferred.ev[4], c->stream2));
    if (launch_detile(c, c->stream2, c->deferred.jobs, c->deferred.njobs, c->deferred.extend, c->deferred.kf)) return -1;
    HIPCHK(c, hipEventRecord(c->deferred.ev[5], c->stream2));
    HIPCHK(c, hipEventRecord(c->ev_detile_done[c->deferred.par], c->stream2));
    c->deferred.valid = false;
    return 0;
}
static int join_detile(vp8hip_ctx *c)
{
    if (c->deferred.valid) {
        HIPCHK(c, hipEventRecord(c->ev_lf_done, c->stream));
        if (launch_deferred(c, c->ev_lf_done)) return -1;
    }
    if (c->detile_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_detile_done[c->last_par], 0));
        c->detile_pending = false;
    }
    c->detile_joined = c->detile_gen;
    return 0;
}
extern "C" int vp8hip_join(vp8hip_ctx *c)
{
    if (!c) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    return join_detile(c);
}

// the stream of the chroma half of a split launch (fused key-frame kernels, lane-per-row loop filter)
static int ensure_stream3(vp8hip_ctx *c)
{
    if (c->stream3) return 0;
    // (a stream of the lowest priority class: the luma kernel, which takes longer, is served first where the two compete)
    int prio_least = 0, prio_greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIPCHK(c, hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_least));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_split_from, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_split_done, hipEventDisableTiming));
    return 0;
}

extern "C" int vp8hip_decode(vp8hip_ctx *c, const vp8hip_job *jobs, int njobs, int stages)
{
    if (!c || !jobs || njobs <= 0) return fail(c, -2, "vp8hip_decode: bad arguments");
    if (!c->width) return fail(c, -2, "vp8hip_decode: context not configured");
    HIPCHK(c, hipSetDevice(c->device));
    if (njobs > c->jobs_cap) {
        // the staging arrays are reused by in-flight launches: drain before growing
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
        for (int k = 0; k < VP8HIP_NBUF; k++) if (c->d_jobs2[k]) (void)hipFree(c->d_jobs2[k]);
        if (c->h_jobs) (void)hipHostFree(c->h_jobs);
        c->jobs_cap = njobs < 64 ? 64 : njobs;
        for (int k = 0; k < VP8HIP_NBUF; k++) HIPCHK(c, hipMalloc((void **)&c->d_jobs2[k], sizeof(DevJob) * c->jobs_cap));
        HIPCHK(c, hipHostMalloc((void **)&c->h_jobs, sizeof(DevJob) * c->jobs_cap, hipHostMallocDefault));
    } else {
        // h_jobs is read by an async copy of the previous call; wait for that copy only
        HIPCHK(c, hipEventSynchronize(c->ev_jobs));
    }
    const int nfb = (int)c->fb.size(), nsl = (int)c->slots.size();
    bool any_lf = false;
    // Which reconstruction kernel: the wave-per-MB-row kernels are the faster ones up to one frame pair per CU
    // (512 frames on an MI355X: 126 vs 85 Gpix/s at 1080p); beyond that they need a second round of workgroups and
    // the one-MB-row-per-lane kernels win (640 frames: 108 vs 85).
    // (Its inter prediction still works 4x4 block by 4x4 block and loses to the wave-per-row kernel on inter
    // frames, so launches that contain inter frames stay with the latter.)
    bool all_key = true;
    for (int i = 0; i < njobs && all_key; i++)
        if (jobs[i].ir_slot >= 0 && jobs[i].ir_slot < nsl) all_key = c->slots[jobs[i].ir_slot].hdr_copy.frame_type == 0;
    const Knobs &K = c->knobs;
    bool simt_recon = (stages & VP8HIP_STAGE_RECON) && all_key && njobs > 2 * c->num_cu;
    if (K.recon_force)           // tuning / test knob: force one of the two kernel families (lane-per-row: key frames only)
        simt_recon = (stages & VP8HIP_STAGE_RECON) && all_key && K.recon_force == 1;
    // Large launches with inter frames: the wave-per-row recon (inter prediction is its business) writes the tiled scratch
    // frames too, and the loop filter is the lane-per-row one, at half the time per frame of the wave-per-row filter once the
    // launch fills the chip
    // ... or, larger still, the key frames' way: all inter predictions first, then residual + loop filter in one pass
    bool inter_fused = (stages & VP8HIP_STAGE_RECON) && (stages & VP8HIP_STAGE_LF) && !all_key && K.fused
                       && (K.inter_fused < 0 ? njobs > 2 * c->num_cu : K.inter_fused > 0 && njobs >= K.inter_fused);
    if (K.recon_force)
        inter_fused = (stages & VP8HIP_STAGE_RECON) && (stages & VP8HIP_STAGE_LF) && !all_key && K.fused && K.recon_force == 1;
    const bool inter_tiled = (stages & VP8HIP_STAGE_RECON) && (stages & VP8HIP_STAGE_LF) && !all_key && K.inter_tiled > 0
                             && njobs >= K.inter_tiled && !inter_fused;
    // the lane-per-row kernels work on macroblock-tiled scratch frames; vp8_detile_kernel converts at the end
    const bool tiled = simt_recon || inter_tiled || inter_fused;
    // both stages wanted: one kernel reconstructs and filters, and writes the raster frame buffers itself
    const bool fused = (simt_recon && (stages & VP8HIP_STAGE_LF) && K.fused) || inter_fused;
    // When the loop filter runs at all (some frame of the launch has filter_level != 0), it writes its finished lines
    // straight into the raster frame buffers -- unfiltered frames are carried through with the filter gated off -- (rows of two neighbouring macroblocks back to back: 32-byte pieces) and the tiled -> raster
    // pass is skipped; only the border extension is left.  +6..10 % from 1536 frames per launch up (1080p), a tie at 2048,
    // -4 % at 1024.  VP8HIP_LF_RASTER=0 keeps the tiled -> raster pass.
    bool lf_raster = false;
    if (!fused && tiled && (stages & VP8HIP_STAGE_LF) && K.lf_raster)
        for (int i = 0; i < njobs && !lf_raster; i++)     // some frame is filtered: the loop filter kernel runs anyway
            if (jobs[i].ir_slot >= 0 && jobs[i].ir_slot < nsl) lf_raster = c->slots[jobs[i].ir_slot].hdr_copy.filter_level != 0;
    // scratch of a frame: a tile per macroblock -- the key-frame kernel's layout has one more per macroblock row and 32 bytes of
    // unfiltered line per tile behind them (vp8_keyframe_simt.hip)
    const size_t tile_frame = align_up((size_t)c->dg.mb_rows * (c->dg.mb_cols + 1) * (VP8_TILE_BYTES + 32), 256);
    const int par = c->parity;
    bool reads_pending = false;
    if (!all_key)
        for (int i = 0; i < njobs && !reads_pending; i++) {
            if (jobs[i].ir_slot < 0 || jobs[i].ir_slot >= nsl || c->slots[jobs[i].ir_slot].hdr_copy.frame_type == 0) continue;
            for (int k = 1; k < 4; k++) {
                const int f = jobs[i].ref_fb[k];
                if (f >= 0 && f < nfb && c->fb_detile_gen[f] > c->detile_joined) reads_pending = true;
            }
        }
    if (!tiled || lf_raster || reads_pending) {
        // this launch touches the raster frame buffers directly: it writes them, or (inter frames) reads reference frames a
        // tiled -> raster pass of an earlier launch is still to produce
        if (join_detile(c)) return -1;
    }
    if (tiled) {
        // scratch set and job table `par` were last read by the tiled -> raster pass VP8HIP_NBUF launches ago (three
        // sets: that pass, launched beside the previous launch's loop filter, may still be finishing)
        if (c->detile_used[par]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_detile_done[par], 0));
        if (c->tile_cap[par] < tile_frame * njobs) {
            // (re)allocate the scratch set in use (the other sets only exist once the tiled -> raster pass has rotated to them)
            if (join_detile(c)) return -1;          // a pass not launched yet still reads the old sets
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
            if (c->tile_cap[par] < tile_frame * njobs) {
                if (c->tile_block[par]) (void)hipFree(c->tile_block[par]);
                c->tile_block[par] = nullptr; c->tile_cap[par] = 0;
                // + 8 KB: the dummy tile idle lanes write, and room for the lane-per-row kernels' prefetches past the last tile
                HIPCHK(c, hipMalloc((void **)&c->tile_block[par], tile_frame * njobs + 8192));
                c->tile_cap[par] = tile_frame * njobs;
            }
        }
    }
    c->d_jobs = c->d_jobs2[par];
    for (int i = 0; i < njobs; i++) {
        const vp8hip_job &j = jobs[i];
        if (j.ir_slot < 0 || j.ir_slot >= nsl || j.dst_fb < 0 || j.dst_fb >= nfb)
            return fail(c, -2, "vp8hip_decode: job %d has slot %d / fb %d out of range", i, j.ir_slot, j.dst_fb);
        const Slot &s = c->slots[j.ir_slot];
        DevJob &d = c->h_jobs[i];
        d.hdr = s.hdr_copy;
        d.mbs = s.d_mbs; d.coef = s.d_coef; d.mvs = s.d_mvs;
        d.dst = c->fb[j.dst_fb];
        d.ref[0] = nullptr;
        d.tile = tiled ? c->tile_block[par] + tile_frame * i : nullptr;
        for (int k = 1; k < 4; k++) {
            d.ref[k] = nullptr;
            if (s.hdr_copy.frame_type == 0) continue;          // key frames read no reference
            int f = j.ref_fb[k];
            if (f >= nfb) return fail(c, -2, "vp8hip_decode: job %d ref %d out of range", i, f);
            if (f < 0) return fail(c, -2, "vp8hip_decode: inter frame job %d lacks reference %d", i, k);
            if (f == j.dst_fb) return fail(c, -2, "vp8hip_decode: job %d decodes into its own reference", i);
            d.ref[k] = c->fb[f];
        }
        any_lf |= s.hdr_copy.filter_level != 0;
    }
    if (c->d2h_count) {      // a batch download still in flight: a launch that writes one of its frame buffers waits for it
        bool hit = false;
        for (int i = 0; i < njobs && !hit; i++) hit = jobs[i].dst_fb >= c->d2h_first && jobs[i].dst_fb < c->d2h_first + c->d2h_count;
        if (hit) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2h_done, 0));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, c->h_jobs, sizeof(DevJob) * njobs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_jobs, c->stream));

    // ---- the coefficient form this launch's kernels read: packed for vp8_keyframe_kernel, dense for every other
    if (stages & VP8HIP_STAGE_RECON) {
        if (c->pack_cap < njobs) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_pack) (void)hipFree(c->d_pack);
            if (c->h_pack) (void)hipHostFree(c->h_pack);
            c->d_pack = nullptr; c->h_pack = nullptr; c->pack_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_pack, sizeof(int) * (size_t)(njobs < 64 ? 64 : njobs)));
            HIPCHK(c, hipHostMalloc((void **)&c->h_pack, sizeof(int) * (size_t)(njobs < 64 ? 64 : njobs), hipHostMallocDefault));
            c->pack_cap = njobs < 64 ? 64 : njobs;
        }
        int n = 0;
        for (int i = 0; i < njobs; i++) {
            Slot &sl = c->slots[jobs[i].ir_slot];
            if (sl.packed != fused) { sl.packed = fused; c->h_pack[n++] = jobs[i].ir_slot; }     // (a slot named twice converts once)
        }
        if (n) {
            // (h_pack is reused by the next call: the copy below is waited for through ev_jobs, recorded after it)
            HIPCHK(c, hipMemcpyAsync(c->d_pack, c->h_pack, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipEventRecord(c->ev_jobs, c->stream));
            const long threads = (long)n * c->nmb;
            hipLaunchKernelGGL(vp8_ir_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, c->slot_block_dev,
                               c->slot_bytes, c->o_mbs, c->o_coef, (const int *)c->d_pack, n, c->nmb, fused ? 0 : 1);
            HIPCHK(c, hipGetLastError());
        }
    }
    const int wg_per_cu = K.wg_per_cu >= 1 && K.wg_per_cu <= 8 ? K.wg_per_cu : 1;
    const int grid = njobs < c->num_cu * wg_per_cu ? njobs : c->num_cu * wg_per_cu;
    c->stats.workgroups = grid;
    c->stats.recon_waves = c->recon_nw;
    c->stats.lf_waves = c->lf_nw;
    // ---- small launches of the wave-per-row family: a frame pair is spread over S workgroups of XCU_NW waves on S CUs
    // of one XCD (round-robin placement: workgroups b, b+8, b+16, ... share an XCD) instead of living on one CU, so
    // that one 1080p frame keeps 68 SIMDs busy, not 4.  At most 32 CUs' worth of workgroups per XCD, one pair per group.
    int XCU_NW = 4;
    int xcu_S = 1, xcu_grid = 0;
    if (!tiled) {
        const int npairs = (njobs + 1) / 2, rows = c->dg.mb_rows, cols = c->dg.mb_cols;
        const int per_xcd = (npairs + 7) / 8;
        int S = (rows + XCU_NW - 1) / XCU_NW;                // a wave per row ...
        if (per_xcd > 32) S = 1;
        else if (S > 32 / per_xcd) S = 32 / per_xcd;         // ... or one workgroup on every CU of the XCD
        // fewer waves than rows: two waves per SIMD.  Worth it as long as a pair gets more waves than the twelve it
        // has on a single CU (a wave's macroblock step is a latency chain; throughput goes with the number of waves)
        if (S * XCU_NW < rows) XCU_NW = 8;
        if (S * XCU_NW <= c->recon_nw) S = 1;
        if (!K.xcu) S = 1;
        if (K.xcu_S >= 1 && K.xcu_S <= 64) S = K.xcu_S;
        if (K.xcu_NW == 4 || K.xcu_NW == 8) XCU_NW = K.xcu_NW;
        if (S > 1) {
            // the workgroups of a group wait for each other: all of them have to be resident at once, on this device as it
            // is (fewer CUs when partitioned), or the launch stays with one workgroup per pair
            int per_cu_r = 0, per_cu_l = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_r, vp8_recon_xcu_kernel, 64 * XCU_NW, 1024 + XCU_NW * 2 * 2080) != hipSuccess
                || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_l, vp8_loopfilter_xcu_kernel, 64 * XCU_NW, 256 + XCU_NW * 2 * 4096) != hipSuccess
                || 8 * S * per_xcd > c->num_cu * (per_cu_r < per_cu_l ? per_cu_r : per_cu_l))
                S = 1;
        }
        if (S > 1) {
            xcu_S = S; xcu_grid = 8 * S * per_xcd;
            if (!c->h_status) {
                HIPCHK(c, hipHostMalloc((void **)&c->h_status, sizeof(int), hipHostMallocMapped));
                *c->h_status = 0;
                HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_status, c->h_status, 0));
            }
            // granule buffers: recon one unfiltered pixel line per MB row (cols*8+2 granules), loop filter four
            // context rows per MB row (cols*32), per frame; zeroed once -- the tags of later launches never repeat
            const size_t need_r = (size_t)npairs * 2 * rows * (cols * 8 + 2) * 8, need_l = (size_t)npairs * 2 * rows * cols * 32 * 8;
            if (c->gran_recon_cap < need_r || c->gran_lf_cap < need_l) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (c->gran_recon) (void)hipFree(c->gran_recon);
                if (c->gran_lf) (void)hipFree(c->gran_lf);
                c->gran_recon = c->gran_lf = nullptr; c->gran_recon_cap = c->gran_lf_cap = 0;
                HIPCHK(c, hipMalloc((void **)&c->gran_recon, need_r));
                HIPCHK(c, hipMalloc((void **)&c->gran_lf, need_l));
                HIPCHK(c, hipMemsetAsync(c->gran_recon, 0, need_r, c->stream));
                HIPCHK(c, hipMemsetAsync(c->gran_lf, 0, need_l, c->stream));
                c->gran_recon_cap = need_r; c->gran_lf_cap = need_l;
                c->epoch = 0;
            }
            if (++c->epoch == 0) {          // 2^32 launches later: start over with clean buffers
                HIPCHK(c, hipMemsetAsync(c->gran_recon, 0, c->gran_recon_cap, c->stream));
                HIPCHK(c, hipMemsetAsync(c->gran_lf, 0, c->gran_lf_cap, c->stream));
                c->epoch = 1;
            }
            c->stats.workgroups = xcu_grid; c->stats.recon_waves = XCU_NW; c->stats.lf_waves = XCU_NW;
        }
    }
    hipEvent_t *ev = c->evr[c->ncalls % VP8HIP_STATS_RING];
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    // "one MB row per lane" kernels: G lanes per strand of frames, row period P >= max(cols, 2G+2).  G is at most
    // the largest value that costs no idle steps (cols >= 2G+2), and otherwise as small as it can be while every
    // strand of a full launch (4 waves per CU) still gets a frame: a small G means few pipeline-fill steps and a
    // better fit of the frame's rows into whole row periods.
    int lgG = 1;
    {
        const int cols = c->dg.mb_cols;
        int lgmax = 1;
        while (lgmax < 6 && 2 * (2 << lgmax) + 2 <= cols) lgmax++;
        const long lanes = (long)c->num_cu * 4 * 64;                      // one wave per SIMD
        while (lgG < 6 && (lanes >> lgG) > njobs) lgG++;                  // strands of a full launch <= frames
        if (lgG > lgmax && ((long)njobs << lgmax) >= lanes) lgG = lgmax;   // no idle steps, if that still fills every SIMD
        if (K.lgG >= 1 && K.lgG <= 6) lgG = K.lgG;
    }
    const int simtG = 1 << lgG, spw = 64 >> lgG;
    const int simtP = c->dg.mb_cols > 2 * simtG + 2 ? c->dg.mb_cols : 2 * simtG + 2;
    int simt_waves = (njobs + spw - 1) / spw;
    {
        int maxw = c->num_cu * 4;
        if (K.simt_waves >= 1) maxw = K.simt_waves;
        if (simt_waves > maxw) simt_waves = maxw;
    }
    if (tiled) { c->stats.workgroups = simt_waves; c->stats.recon_waves = simt_recon || inter_fused ? 1 : c->recon_nw; c->stats.lf_waves = 1; }
    c->stats.detile_pass = tiled && !lf_raster;
    if (stages & VP8HIP_STAGE_RECON) {
        if (fused) {
            // one kernel, two waves per SIMD: the first to arrive on a SIMD reconstructs luma, the second chroma (see the kernel)
            if (!c->d_sched) {
                HIPCHK(c, hipMalloc((void **)&c->d_sched, sizeof(unsigned int) * VP8HIP_SCHED_WORDS));
                HIPCHK(c, hipMemsetAsync(c->d_sched, 0, sizeof(unsigned int) * VP8HIP_SCHED_WORDS, c->stream));
            }
            HIPCHK(c, hipMemsetAsync(c->d_sched, 0, 2 * sizeof(unsigned int), c->stream));
            if (inter_fused) {
                // the inter macroblocks' predictions into their tiles: a wave per 64 macroblocks, at most 8 waves per SIMD's worth
                const int upf = (c->nmb + 63) / 64;
                long pgrid = ((long)njobs * upf + 3) / 4;
                if (pgrid > (long)c->num_cu * 8) pgrid = (long)c->num_cu * 8;
                hipLaunchKernelGGL(vp8_inter_pred_kernel, dim3((unsigned)pgrid), dim3(256), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                                   c->dg, upf);
                // the previous launch's tiled -> raster pass, if it was held back: beside vp8_interframe_kernel, which is bound by
                // arithmetic, not beside the prediction kernel, which is bound by memory bandwidth as the pass is
                if (c->deferred.valid) {
                    HIPCHK(c, hipEventRecord(c->ev_recon_done, c->stream));
                    if (launch_deferred(c, c->ev_recon_done)) return -1;
                }
                hipLaunchKernelGGL(vp8_interframe_kernel, dim3(2 * simt_waves), dim3(64), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                                   c->dg, lgG, simtP, simt_waves * spw, c->tile_block[par] + tile_frame * njobs + 4096,
                                   c->d_sched, simt_waves);
            } else
            hipLaunchKernelGGL(vp8_keyframe_kernel, dim3(2 * simt_waves), dim3(64), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                               c->dg, lgG, simtP, simt_waves * spw, c->tile_block[par] + tile_frame * njobs + 4096,
                               c->d_sched, simt_waves);
        } else if (simt_recon) {
            hipLaunchKernelGGL(vp8_recon_simt_kernel, dim3(simt_waves), dim3(64), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                               c->dg, lgG, simtP, simt_waves * spw, c->tile_block[par] + tile_frame * njobs + 4096);
        } else {
            const int npairs = (njobs + 1) / 2;          // two frames per wave
            // launches with inter frames: their inter macroblocks first, every one on its own (vp8_inter_mb_kernel), then the
            // row-ordered kernel for the intra macroblocks only
            const bool inter_first = !all_key && njobs <= K.inter_split;
            if (inter_first) {
                if (c->intra_flags_cap < njobs) {
                    if (c->d_intra_flags) (void)hipFree(c->d_intra_flags);
                    c->d_intra_flags = nullptr; c->intra_flags_cap = 0;
                    HIPCHK(c, hipMalloc((void **)&c->d_intra_flags, sizeof(unsigned int) * (size_t)njobs));
                    c->intra_flags_cap = njobs;
                }
                HIPCHK(c, hipMemsetAsync(c->d_intra_flags, 0, sizeof(unsigned int) * (size_t)njobs, c->stream));
                const long units = (long)njobs * ((c->nmb + 1) / 2);
                long igrid = (units + 3) / 4;
                if (igrid > (long)c->num_cu * 16) igrid = (long)c->num_cu * 16;
                hipLaunchKernelGGL(vp8_inter_mb_kernel, dim3((unsigned)igrid), dim3(256), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                                   c->dg, c->d_intra_flags);
            }
            if (xcu_S > 1) {
                if (inter_first)
                    hipLaunchKernelGGL(vp8_recon_intra_xcu_kernel, dim3(xcu_grid), dim3(64 * XCU_NW), 1024 + XCU_NW * 2 * 2080, c->stream,
                                       (const DevJob *)c->d_jobs, njobs, c->dg, c->gran_recon, c->epoch, xcu_S, c->d_status,
                                       (const unsigned int *)c->d_intra_flags);
                else
                hipLaunchKernelGGL(vp8_recon_xcu_kernel, dim3(xcu_grid), dim3(64 * XCU_NW), 1024 + XCU_NW * 2 * 2080, c->stream,
                                   (const DevJob *)c->d_jobs, njobs, c->dg, c->gran_recon, c->epoch, xcu_S, c->d_status);
            } else {
            const int rgrid = npairs < c->num_cu * wg_per_cu ? npairs : c->num_cu * wg_per_cu;
            if (inter_first)
                hipLaunchKernelGGL(vp8_recon_intra_kernel, dim3(rgrid), dim3(64 * c->recon_nw), c->recon_lds, c->stream,
                                   (const DevJob *)c->d_jobs, njobs, c->dg, (const unsigned int *)c->d_intra_flags);
            else
            hipLaunchKernelGGL(vp8_recon_kernel, dim3(rgrid), dim3(64 * c->recon_nw), c->recon_lds, c->stream,
                               (const DevJob *)c->d_jobs, njobs, c->dg);
            }
        }
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(ev[1], c->stream));
    if (tiled && c->deferred.valid) {          // the previous launch's tiled -> raster pass runs beside this loop filter
        HIPCHK(c, hipEventRecord(c->ev_recon_done, c->stream));
        if (launch_deferred(c, c->ev_recon_done)) return -1;
    }
    c->stats.fused = fused;
    if (fused) c->stats.lf_kernels = 0;
    if ((stages & VP8HIP_STAGE_LF) && any_lf && !fused) {
        c->stats.lf_kernels = tiled ? 2 : 1;
        if (tiled) {
            // luma and chroma as two kernels side by side: a luma wave (268 registers, 25.6 KB of LDS) and a chroma wave (187,
            // 9.2 KB) share a SIMD, so every SIMD has two instruction streams to issue from.  The chroma kernel goes out on
            // a stream of its own behind the recon, and the main stream takes it back in before anything reads the frames.
            if (ensure_stream3(c)) return -1;
            HIPCHK(c, hipEventRecord(c->ev_split_from, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_split_from, 0));
            hipLaunchKernelGGL(vp8_loopfilter_simt_luma_kernel, dim3(simt_waves), dim3(64), 0, c->stream, (const DevJob *)c->d_jobs,
                               njobs, c->dg, lgG, simtP, simt_waves * spw, lf_raster ? 1 : 0);
            hipLaunchKernelGGL(vp8_loopfilter_simt_chroma_kernel, dim3(simt_waves), dim3(64), 0, c->stream3, (const DevJob *)c->d_jobs,
                               njobs, c->dg, lgG, simtP, simt_waves * spw, lf_raster ? 1 : 0);
            HIPCHK(c, hipEventRecord(c->ev_split_done, c->stream3));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_split_done, 0));
        } else {
            const int npairs = (njobs + 1) / 2;          // the loop filter works on two frames per wave
            if (xcu_S > 1) {
                hipLaunchKernelGGL(vp8_loopfilter_xcu_kernel, dim3(xcu_grid), dim3(64 * XCU_NW), 256 + XCU_NW * 2 * 4096, c->stream,
                                   (const DevJob *)c->d_jobs, njobs, c->dg, c->gran_lf, c->epoch, xcu_S, c->d_status);
            } else {
            const int lfgrid = npairs < c->num_cu * wg_per_cu ? npairs : c->num_cu * wg_per_cu;
            hipLaunchKernelGGL(vp8_loopfilter_kernel, dim3(lfgrid), dim3(64 * c->lf_nw), c->lf_lds, c->stream,
                               (const DevJob *)c->d_jobs, njobs, c->dg);
            }
        }
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(ev[2], c->stream));
    if (lf_raster) {
        if (stages & VP8HIP_STAGE_EXTEND) {
            // on the main stream: 1.8 ms per 8192 1080p frames; beside the next launch's recon (second stream) it
            // stretched both by more than it takes alone
            int bx = (c->geom.aligned_h + 64) / 4;
            if (bx < 1) bx = 1;
            if (bx > 64) bx = 64;
            HIPCHK(c, hipEventRecord(ev[4], c->stream));
            hipLaunchKernelGGL(vp8_extend_kernel, dim3(bx, njobs), dim3(256), 0, c->stream, (const DevJob *)c->d_jobs, njobs, c->dg);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(ev[5], c->stream));
        }
        // no rotation: nothing reads this scratch set once the launch's loop filter is done
    } else if (tiled) {      // whatever stages ran, the frame buffer gets the result; borders are extended on the way
        const bool own_stream = K.detile_stream;
        if (own_stream && !c->stream2) {
            int prio_least = 0, prio_greatest = 0;
            HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            HIPCHK(c, hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least));
        }
        // deferred by default: the pass is launched with the NEXT lane-per-row launch, right after its recon, so
        // that it runs beside that launch's loop filter (which it disturbs less than the recon), or at the next join
        // (the key-frame kernel's pass goes out at once: its waves are small enough -- 16 registers -- to run in the gaps the next
        // launch's kernel leaves on every SIMD; after a launch with inter frames it waits for the next launch's prediction kernel)
        const bool defer = own_stream && K.detile_defer && (!fused || inter_fused);
        hipStream_t ds = own_stream ? c->stream2 : c->stream;
        if (defer) {
            c->deferred.valid = true; c->deferred.kf = fused; c->deferred.jobs = c->d_jobs; c->deferred.njobs = njobs;
            c->deferred.extend = (stages & VP8HIP_STAGE_EXTEND) ? 1 : 0; c->deferred.par = par; c->deferred.ev = ev;
        } else {
        if (own_stream) {
            HIPCHK(c, hipEventRecord(c->ev_lf_done, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_lf_done, 0));
        }
        HIPCHK(c, hipEventRecord(ev[4], ds));
        if (launch_detile(c, ds, c->d_jobs, njobs, (stages & VP8HIP_STAGE_EXTEND) ? 1 : 0, fused)) return -1;
        HIPCHK(c, hipEventRecord(ev[5], ds));
        HIPCHK(c, hipEventRecord(c->ev_detile_done[par], ds));
        }
        c->detile_used[par] = true; c->detile_pending = true; c->last_par = par; c->parity = (par + 1) % VP8HIP_NBUF;
        ++c->detile_gen;
        for (int i = 0; i < njobs; i++) c->fb_detile_gen[jobs[i].dst_fb] = c->detile_gen;
    } else if (stages & VP8HIP_STAGE_EXTEND) {
        int bx = (c->geom.aligned_h + 64) / 4;
        if (bx < 1) bx = 1;
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(vp8_extend_kernel, dim3(bx, njobs), dim3(256), 0, c->stream, (const DevJob *)c->d_jobs,
                           njobs, c->dg);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(ev[3], c->stream));
    c->evr_tiled[c->ncalls % VP8HIP_STATS_RING] = tiled && (!lf_raster || (stages & VP8HIP_STAGE_EXTEND));
    c->evr_stats[c->ncalls % VP8HIP_STATS_RING] = c->stats;
    c->ncalls++;
    return 0;
}

// after a stream synchronisation: did a kernel of the cross-CU family give up on a hand-over?
static int check_status(vp8hip_ctx *c)
{
    if (c->h_status && *c->h_status) {
        const int st = *c->h_status;
        *c->h_status = 0;
        return fail(c, -1, "a row hand-over between CUs did not arrive (%s kernel): the frames of that launch are invalid",
                    st == 1 ? "reconstruction" : "loop filter");
    }
    return 0;
}

extern "C" int vp8hip_sync(vp8hip_ctx *c)
{
    if (!c) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_status(c);
}

extern "C" int vp8hip_get_stats_at(vp8hip_ctx *c, int back, vp8hip_stats *st)
{
    if (!c || !st || back < 0 || back >= VP8HIP_STATS_RING) return -2;
    if (back >= c->ncalls) { memset(st, 0, sizeof *st); return c->ncalls ? fail(c, -2, "vp8hip_get_stats_at: only %ld launches so far", c->ncalls) : 0; }
    const int r = (int)((c->ncalls - 1 - back) % VP8HIP_STATS_RING);
    if (c->deferred.valid && join_detile(c)) return -1;      // its events are read below: it has to be launched
    hipEvent_t *ev = c->evr[r];
    vp8hip_stats out = c->evr_stats[r];
    HIPCHK(c, hipEventSynchronize(ev[3]));
    (void)hipEventElapsedTime(&out.recon_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&out.lf_ms, ev[1], ev[2]);
    if (c->evr_tiled[r]) {
        HIPCHK(c, hipEventSynchronize(ev[5]));
        (void)hipEventElapsedTime(&out.extend_ms, ev[4], ev[5]);
    } else
        (void)hipEventElapsedTime(&out.extend_ms, ev[2], ev[3]);
    *st = out;
    return 0;
}
extern "C" int vp8hip_get_stats(vp8hip_ctx *c, vp8hip_stats *st) { return vp8hip_get_stats_at(c, 0, st); }

extern "C" void *vp8hip_stream(vp8hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int vp8hip_frame_download(vp8hip_ctx *c, int fb, int full, uint8_t *y, uint8_t *u, uint8_t *v,
                                     int y_stride, int uv_stride)
{
    if (!c || fb < 0 || fb >= (int)c->fb.size() || !y) return fail(c, -2, "vp8hip_frame_download: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    const vp8ir_geom &g = c->geom;
    if (full) {
        HIPCHK(c, hipMemcpyAsync(y, c->fb[fb], (size_t)g.frame_size, hipMemcpyDeviceToHost, c->stream));
    } else {
        if (!u || !v) return fail(c, -2, "vp8hip_frame_download: null chroma pointers");
        const int cw = (c->width + 1) / 2, ch = (c->height + 1) / 2;
        HIPCHK(c, hipMemcpy2DAsync(y, y_stride, c->fb[fb] + g.y_off, g.y_stride, c->width, c->height,
                                   hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(u, uv_stride, c->fb[fb] + g.u_off, g.uv_stride, cw, ch, hipMemcpyDeviceToHost,
                                   c->stream));
        HIPCHK(c, hipMemcpy2DAsync(v, uv_stride, c->fb[fb] + g.v_off, g.uv_stride, cw, ch, hipMemcpyDeviceToHost,
                                   c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_status(c);
}

// Output-side post-processing of one frame buffer into another (vp8_post_proc_frame, vp8/common/postproc.c:903-1000, minus
// the policy: the caller has turned the frame's quantiser into thresholds and drawn the random phases).
extern "C" int vp8hip_postproc(vp8hip_ctx *c, int src_fb, int dst_fb, int tmp_fb, const vp8hip_pp *pp)
{
    const int nfb = c ? (int)c->fb.size() : 0;
    if (!c || !pp || src_fb < 0 || src_fb >= nfb || dst_fb < 0 || dst_fb >= nfb)
        return fail(c, -2, "vp8hip_postproc: bad arguments");
    const bool demacro = pp->flags & VP8HIP_PP_DEMACROBLOCK, deblock = demacro || (pp->flags & VP8HIP_PP_DEBLOCK);
    if (dst_fb == src_fb && deblock)                     // (in place: the noise alone, on a picture vp8hip_mfqe left in dst_fb)
        return fail(c, -2, "vp8hip_postproc: the deblocking filters cannot run in place");
    const bool noise = pp->flags & VP8HIP_PP_ADDNOISE;
    if (demacro && (tmp_fb < 0 || tmp_fb >= nfb || tmp_fb == src_fb || tmp_fb == dst_fb || !pp->rv || pp->rv_offset < 0 || pp->rv_offset > 63))
        return fail(c, -2, "vp8hip_postproc: demacroblocking needs a third frame buffer and the dither table");
    const vp8ir_geom &g = c->geom;
    // the noise row of a line starts up to 255 entries into the 3072-entry table (the reference indexes past its end for
    // wider frames, postproc.c:499-510: no defined answer to reproduce)
    if (noise && (!pp->noise_rows || g.aligned_w + 255 > 3072 || g.aligned_h > 16384))
        return fail(c, -2, "vp8hip_postproc: noise needs the row phases and a frame at most 2816 wide");
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    if (c->d2h_count) { HIPCHK(c, hipEventSynchronize(c->ev_d2h_done)); c->d2h_count = 0; }   // a batch download may be reading dst
    if (!c->d_pp || !c->h_pp || !c->ev_pp) {
        // (each piece on its own: a failure half way leaves what exists for the next call, never a null event to wait on)
        if (!c->d_pp) {
            HIPCHK(c, hipMalloc((void **)&c->d_pp, 1024 + 3072 + 16384));
            HIPCHK(c, hipMemsetAsync(c->d_pp, 0, 1024 + 3072 + 16384, c->stream));     // a noise table never sent is all zeros, as the reference's
        }
        // the caller's tables go through a pinned copy of our own, so that they may be reused the moment the call returns
        if (!c->h_pp) HIPCHK(c, hipHostMalloc((void **)&c->h_pp, 1024 + 3072 + 16384, hipHostMallocDefault));
        if (!c->ev_pp) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pp, hipEventDisableTiming));
    } else
        HIPCHK(c, hipEventSynchronize(c->ev_pp));       // the previous call's copies have left the pinned staging
    const short *d_rv = (const short *)c->d_pp;
    signed char *d_noise = (signed char *)c->d_pp + 1024;
    uint8_t *d_rows = (uint8_t *)c->d_pp + 1024 + 3072;
    uint8_t *src = c->fb[src_fb], *dst = c->fb[dst_fb];
    const struct { int off, stride, rows, cols; } pl[3] = { { g.y_off, g.y_stride, g.aligned_h, g.aligned_w },
                                                            { g.u_off, g.uv_stride, g.aligned_h / 2, g.aligned_w / 2 },
                                                            { g.v_off, g.uv_stride, g.aligned_h / 2, g.aligned_w / 2 } };
    if (deblock) {
        for (int k = 0; k < 3; k++)
            vp8pp_down_and_across(c->stream, src + pl[k].off, dst + pl[k].off, pl[k].stride, pl[k].rows, pl[k].cols, pp->flimit);
        if (demacro) {       // luma only (vp8_deblock_and_de_macro_block, postproc.c:328-346)
            uint8_t *tmp = c->fb[tmp_fb];
            if (!c->pp_rv_loaded) {
                memcpy(c->h_pp, pp->rv, 440 * sizeof(short));
                HIPCHK(c, hipMemcpyAsync(c->d_pp, c->h_pp, 440 * sizeof(short), hipMemcpyHostToDevice, c->stream));
                c->pp_rv_loaded = true;
            }
            vp8pp_mb_across(c->stream, dst + pl[0].off, tmp + pl[0].off, pl[0].stride, pl[0].rows, pl[0].cols, pp->mb_flimit);
            vp8pp_mb_down(c->stream, tmp + pl[0].off, dst + pl[0].off, pl[0].stride, pl[0].rows, pl[0].cols, pp->mb_flimit,
                          d_rv + pp->rv_offset);
        }
    } else if (dst != src)      // vp8_yv12_copy_frame_ptr (postproc.c:982)
        HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)g.frame_size, hipMemcpyDeviceToDevice, c->stream));
    if (noise) {
        if (pp->noise) {
            memcpy(c->h_pp + 1024, pp->noise, 3072);
            HIPCHK(c, hipMemcpyAsync(d_noise, c->h_pp + 1024, 3072, hipMemcpyHostToDevice, c->stream));
        }
        memcpy(c->h_pp + 1024 + 3072, pp->noise_rows, (size_t)pl[0].rows);
        HIPCHK(c, hipMemcpyAsync(d_rows, c->h_pp + 1024 + 3072, (size_t)pl[0].rows, hipMemcpyHostToDevice, c->stream));
        vp8pp_add_noise(c->stream, dst + pl[0].off, pl[0].stride, pl[0].rows, pl[0].cols, pp->noise_clamp, d_noise, d_rows);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_pp, c->stream));
    return 0;
}

extern "C" int vp8hip_mfqe(vp8hip_ctx *c, int show_fb, int prev_fb, int dst_fb, const uint8_t *mb_class, int qcurr, int qprev)
{
    const int nfb = c ? (int)c->fb.size() : 0;
    if (!c || !mb_class || show_fb < 0 || show_fb >= nfb || prev_fb < 0 || prev_fb >= nfb || dst_fb < 0 || dst_fb >= nfb ||
        show_fb == prev_fb || show_fb == dst_fb || qcurr < 0 || qcurr > 127 || qprev < 0 || qprev > qcurr)
        return fail(c, -2, "vp8hip_mfqe: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    if (c->d2h_count) { HIPCHK(c, hipEventSynchronize(c->ev_d2h_done)); c->d2h_count = 0; }   // a batch download may be reading dst
    const int nmb = c->dg.mb_cols * c->dg.mb_rows;
    if (nmb > c->mfqe_cap) {
        if (c->ev_mfqe) HIPCHK(c, hipEventSynchronize(c->ev_mfqe));
        if (c->d_mfqe) (void)hipFree(c->d_mfqe);
        if (c->h_mfqe) (void)hipHostFree(c->h_mfqe);
        c->d_mfqe = c->h_mfqe = nullptr; c->mfqe_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_mfqe, (size_t)nmb));
        HIPCHK(c, hipHostMalloc((void **)&c->h_mfqe, (size_t)nmb, hipHostMallocDefault));
        c->mfqe_cap = nmb;
    }
    if (!c->ev_mfqe) HIPCHK(c, hipEventCreateWithFlags(&c->ev_mfqe, hipEventDisableTiming));
    else HIPCHK(c, hipEventSynchronize(c->ev_mfqe));    // the previous call's copy has left the pinned staging
    memcpy(c->h_mfqe, mb_class, (size_t)nmb);
    HIPCHK(c, hipMemcpyAsync(c->d_mfqe, c->h_mfqe, (size_t)nmb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_mfqe, c->stream));
    vp8pp_mfqe(c->stream, c->fb[show_fb], c->fb[prev_fb], c->fb[dst_fb], c->dg, c->d_mfqe, qcurr, qprev);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// first_slot >= 0: into the IR slots (dense); first_slot < 0: into the context's sparse arenas (vp8hip_entropy_decode_sparse)
static int entropy_launch(vp8hip_ctx *c, int first_slot, int count, const vp8hip_entropy_frame *frames, const uint8_t *data, size_t data_bytes,
                          size_t blocks_cap, size_t dcs_cap)
{
    const bool sparse = first_slot < 0;
    bool any_inter = false;
    if (!c || !frames || !data || count < 1 || (!sparse && first_slot + count > (int)c->slots.size()))
        return fail(c, -2, "vp8hip_entropy_decode: bad arguments");
    for (int i = 0; i < count; i++) {
        const vp8hip_entropy_frame &f = frames[i];
        const vp8ir_frame_hdr &h = f.hdr;
        if (h.frame_type != 0) {
            any_inter = true;
            if (sparse) return fail(c, -2, "vp8hip_entropy_decode_sparse: frame %d is not a key frame", i);
        }
        if (h.mb_cols != c->dg.mb_cols || h.mb_rows != c->dg.mb_rows)
            return fail(c, -2, "vp8hip_entropy_decode: frame %d is %dx%d MBs, context configured for %dx%d", i, h.mb_cols, h.mb_rows,
                        c->dg.mb_cols, c->dg.mb_rows);
        bool ok = (f.num_tok == 1 || f.num_tok == 2 || f.num_tok == 4 || f.num_tok == 8) && f.data_off <= data_bytes &&
                  f.first_pos <= f.first_end && f.first_end <= data_bytes - f.data_off && data_bytes - f.data_off >= f.first_end && f.first_range >= 128 && f.first_range <= 255 &&
                  f.first_bits >= -8 && f.first_bits <= 24;
        for (unsigned k = 0; ok && k < f.num_tok; k++) ok = f.tok_pos[k] <= f.tok_end[k] && f.tok_end[k] <= data_bytes - f.data_off;
        if (!ok) return fail(c, -2, "vp8hip_entropy_decode: frame %d: partitions outside the data, or no decoder state", i);
    }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t fbytes = (size_t)count * sizeof(vp8hip_entropy_frame);
    // frames coded with several token partitions, all with the same number: a partition per lane (vp8_entropy_parts_kernel)
    int np = (int)frames[0].num_tok;
    for (int i = 1; i < count && np > 1; i++) if ((int)frames[i].num_tok != np) np = 1;
    // (the lanes of a frame follow each other a macroblock apart and lane 0 follows the last one into the next round of rows: rows
    // at least as long as the partitions are many; the row above's flags of a wave's frames in 16 KB of LDS)
    if (c->dg.mb_cols < np || c->dg.mb_cols > 256 || c->dg.mb_cols * (64 / np) > 4096 || c->ent_parts_off || sparse || any_inter) np = 1;
    const size_t swords = np > 1 ? (size_t)count * ((size_t)c->dg.mb_cols + 3 * (size_t)c->nmb) : (size_t)count * (8 * (size_t)c->dg.mb_cols + 64);
    if (fbytes > c->ent_frames_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_ent_frames) (void)hipFree(c->d_ent_frames);
        if (c->d_ent_status) (void)hipFree(c->d_ent_status);
        c->d_ent_frames = nullptr; c->d_ent_status = nullptr; c->ent_frames_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_frames, fbytes));
        HIPCHK(c, hipMalloc((void **)&c->d_ent_status, (size_t)count * 4));
        c->ent_frames_cap = fbytes;
    }
    if (data_bytes + 16 > c->ent_data_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_ent_data) (void)hipFree(c->d_ent_data);
        c->d_ent_data = nullptr; c->ent_data_cap = 0;
        const size_t cap = data_bytes + data_bytes / 4 + 4096;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_data, cap));
        c->ent_data_cap = cap;
    }
    if (swords > c->ent_scratch_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_ent_scratch) (void)hipFree(c->d_ent_scratch);
        c->d_ent_scratch = nullptr; c->ent_scratch_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_scratch, swords * 4));
        c->ent_scratch_cap = swords;
    }
    if (!c->ent_tables_loaded) {
        c->ent_tables_loaded = true;
        HIPCHK(c, hipFuncSetAttribute((const void *)vp8_entropy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vp8_entropy_lds_bytes(64)));
        HIPCHK(c, hipFuncSetAttribute((const void *)vp8_entropy_sparse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vp8_entropy_lds_bytes(64)));
        const char *e = getenv("VP8HIP_ENTROPY_LANES");     // lanes of a wave that carry a frame (a tuning knob: read once)
        c->ent_lpw = e ? atoi(e) : 0;
        const char *e2 = getenv("VP8HIP_ENTROPY_PARTS");   // 0: a frame per lane whatever the number of token partitions
        c->ent_parts_off = e2 && atoi(e2) == 0;
        if (c->ent_lpw < 1 || c->ent_lpw > 64) c->ent_lpw = 0;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ent_frames, frames, fbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ent_data, data, data_bytes, hipMemcpyHostToDevice, c->stream));
    if (sparse) {
        // arenas: descriptors for every frame; blocks and DCs by the caller's estimate, or by what key frames have been seen to need
        // (blocks: up to 0.41 per compressed byte, DCs up to 0.54), with room to spare (0.6, 0.8) and a chunk per lane on top
        // (never more than every block of every macroblock, plus the chunk a lane may leave unfinished)
        const size_t worst_b = (size_t)count * ((size_t)c->nmb * 25 + 2 * 256), worst_d = (size_t)count * ((size_t)c->nmb * 25 + 2 * 1024);
        size_t nb = blocks_cap ? blocks_cap : (size_t)(data_bytes * 0.6) + (size_t)count * 512, nd = dcs_cap ? dcs_cap : (size_t)(data_bytes * 0.8) + (size_t)count * 2048;
        if (nb > worst_b) nb = worst_b;
        if (nd > worst_d) nd = worst_d;
        if (nb > 0xffff0000ull) nb = 0xffff0000ull;          // (sparse_first / dc_first are 32-bit indices)
        if (nd > 0xffff0000ull) nd = 0xffff0000ull;
        if (vp8hip_entropy_reserve_sparse(c, count, nb, nd)) return -1;
        if (!c->d_sp_cursors) HIPCHK(c, hipMalloc((void **)&c->d_sp_cursors, 16));
        HIPCHK(c, hipMemsetAsync(c->d_sp_cursors, 0, 16, c->stream));
        c->sp_hdrs.resize((size_t)count);
        for (int i = 0; i < count; i++) c->sp_hdrs[(size_t)i] = frames[i].hdr;
        c->sp_count = count;
        // (the caller's caps are honoured as they are; without any, what has been reserved is there to be used)
        c->sp_blocks_use = blocks_cap ? nb : c->sp_blocks_cap; c->sp_dcs_use = dcs_cap ? nd : c->sp_dcs_cap;
    } else
        for (int i = 0; i < count; i++) {
            Slot &s = c->slots[first_slot + i];
            s.hdr_copy = frames[i].hdr;
            s.packed = false;
        }
    // Lanes per wave.  The lanes of a wave go through the macroblocks together, each macroblock taking as long as the slowest
    // lane's, so fewer frames to a wave waste less -- while there are CUs without a wave; several waves to a CU slow each other
    // down again (8192 1080p frames per launch, frames per second over a run: 64 lanes 15.4 k, 32: 16.9-17.9 k, 16: 16.1 k, 8: 12.7 k;
    // 4096 per launch with every frame downloaded: the same 9 k at 16 and 64)
    c->ent_last_sparse = sparse;
    int lpw = c->ent_lpw;
    if (!lpw) lpw = (count + 31) / 32 <= c->num_cu ? 32 : 64;
    if (sparse)
        hipLaunchKernelGGL(vp8_entropy_sparse_kernel, dim3((unsigned)((count + lpw - 1) / lpw)), dim3(64), vp8_entropy_lds_bytes(lpw), c->stream,
                           (const vp8hip_entropy_frame *)c->d_ent_frames, count, lpw, (const uint8_t *)c->d_ent_data, c->dg, data_bytes,
                           c->d_ent_scratch, c->d_ent_status, (ent_u32x4 *)c->d_sp_mbs, (ent_u32x4 *)c->d_sp_blocks, (short *)c->d_sp_dcs,
                           c->d_sp_cursors, (unsigned int)c->sp_blocks_use, (unsigned int)c->sp_dcs_use);
    else if (np > 1)
        hipLaunchKernelGGL(vp8_entropy_parts_kernel, dim3((unsigned)((count + 64 / np - 1) / (64 / np))), dim3(64), 0, c->stream,
                           (const vp8hip_entropy_frame *)c->d_ent_frames, count, np, (const uint8_t *)c->d_ent_data, c->dg, data_bytes,
                           c->slot_block_dev, c->slot_bytes, c->o_mbs, c->o_coef, first_slot, c->d_ent_scratch, c->d_ent_status);
    else
    hipLaunchKernelGGL(vp8_entropy_kernel, dim3((unsigned)((count + lpw - 1) / lpw)), dim3(64), vp8_entropy_lds_bytes(lpw), c->stream,
                       (const vp8hip_entropy_frame *)c->d_ent_frames, count, lpw, (const uint8_t *)c->d_ent_data, c->dg, data_bytes, c->slot_block_dev,
                       c->slot_bytes, c->o_mbs, c->o_coef, c->o_mvs, first_slot, c->d_ent_scratch, c->d_ent_status);
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int vp8hip_entropy_decode(vp8hip_ctx *c, int first_slot, int count, const vp8hip_entropy_frame *frames, const uint8_t *data,
                                     size_t data_bytes)
{
    if (first_slot < 0) return fail(c, -2, "vp8hip_entropy_decode: bad arguments");
    return entropy_launch(c, first_slot, count, frames, data, data_bytes, 0, 0);
}

extern "C" int vp8hip_entropy_decode_sparse(vp8hip_ctx *c, int count, const vp8hip_entropy_frame *frames, const uint8_t *data, size_t data_bytes,
                                            size_t blocks_cap, size_t dcs_cap)
{
    return entropy_launch(c, -1, count, frames, data, data_bytes, blocks_cap, dcs_cap);
}

extern "C" int vp8hip_entropy_reserve_sparse(vp8hip_ctx *c, int max_count, size_t blocks_cap, size_t dcs_cap)
{
    if (!c || max_count < 1 || !c->nmb) return fail(c, -2, "vp8hip_entropy_reserve_sparse: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t mbytes = (size_t)max_count * c->nmb * sizeof(vp8ir_mb);
    if (blocks_cap > 0xffff0000ull) blocks_cap = 0xffff0000ull;
    if (dcs_cap > 0xffff0000ull) dcs_cap = 0xffff0000ull;
    if (mbytes > c->sp_mbs_cap || blocks_cap > c->sp_blocks_cap || dcs_cap > c->sp_dcs_cap) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mbytes > c->sp_mbs_cap) {
        if (c->d_sp_mbs) (void)hipFree(c->d_sp_mbs);
        c->d_sp_mbs = nullptr; c->sp_mbs_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_sp_mbs, mbytes));
        c->sp_mbs_cap = mbytes;
    }
    if (blocks_cap > c->sp_blocks_cap) {
        if (c->d_sp_blocks) (void)hipFree(c->d_sp_blocks);
        c->d_sp_blocks = nullptr; c->sp_blocks_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_sp_blocks, blocks_cap * 32 + 64));
        c->sp_blocks_cap = blocks_cap;
    }
    if (dcs_cap > c->sp_dcs_cap) {
        if (c->d_sp_dcs) (void)hipFree(c->d_sp_dcs);
        c->d_sp_dcs = nullptr; c->sp_dcs_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_sp_dcs, dcs_cap * 2 + 64));
        c->sp_dcs_cap = dcs_cap;
    }
    return 0;
}

extern "C" int vp8hip_ir_expand(vp8hip_ctx *c, int first_frame, int first_slot, int n)
{
    if (!c || n < 1 || first_frame < 0 || first_frame + n > c->sp_count || first_slot < 0 || first_slot + n > (int)c->slots.size() || n > 65535)
        return fail(c, -2, "vp8hip_ir_expand: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < n; i++) {
        Slot &s = c->slots[first_slot + i];
        s.hdr_copy = c->sp_hdrs[(size_t)(first_frame + i)];
        s.packed = false;
    }
    hipLaunchKernelGGL(vp8_ir_clear_kernel, dim3(64, (unsigned)n), dim3(256), 0, c->stream, c->slot_block_dev, c->slot_bytes, c->o_coef, first_slot,
                       (size_t)c->nmb * VP8IR_COEF_PER_MB * sizeof(int16_t));
    hipLaunchKernelGGL(vp8_ir_expand_batch_kernel, dim3((unsigned)((c->nmb + 255) / 256), (unsigned)n), dim3(256), 0, c->stream,
                       (const vp8ir_mb *)c->d_sp_mbs + (size_t)first_frame * c->nmb, (const int16_t *)c->d_sp_blocks, (const int16_t *)c->d_sp_dcs,
                       c->slot_block_dev, c->slot_bytes, c->o_mbs, c->o_coef, first_slot, c->nmb);
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int vp8hip_entropy_status(vp8hip_ctx *c, int count, uint32_t *status)
{
    if (!c || !status || count < 1 || (size_t)count * sizeof(vp8hip_entropy_frame) > c->ent_frames_cap)
        return fail(c, -2, "vp8hip_entropy_status: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(status, c->d_ent_status, (size_t)count * 4, hipMemcpyDeviceToHost));
    if (c->ent_last_sparse) {                               // did the arenas hold?
        unsigned int cur[4] = { 0, 0, 0, 0 };
        HIPCHK(c, hipMemcpy(cur, c->d_sp_cursors, 16, hipMemcpyDeviceToHost));
        if (cur[2]) for (int i = 0; i < count; i++) status[i] |= 2u;
    }
    return 0;
}

extern "C" int vp8hip_ir_fetch_mvs(vp8hip_ctx *c, int slot, vp8ir_mv *mvs)
{
    if (!c || !mvs || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_fetch_mvs: bad slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(mvs, c->slots[slot].d_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vp8hip_ir_fetch(vp8hip_ctx *c, int slot, vp8ir_mb *mbs, int16_t *coef)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_fetch: bad slot %d", slot);
    Slot &s = c->slots[slot];
    HIPCHK(c, hipSetDevice(c->device));
    if (s.packed) {                 // (a large launch has consumed the slot: the dense form, which is what callers see, back in place)
        int *d_one = nullptr;
        HIPCHK(c, hipMalloc((void **)&d_one, sizeof(int)));
        HIPCHK(c, hipMemcpyAsync(d_one, &slot, sizeof(int), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(vp8_ir_pack_kernel, dim3((unsigned)((c->nmb + 255) / 256)), dim3(256), 0, c->stream, c->slot_block_dev, c->slot_bytes,
                           c->o_mbs, c->o_coef, (const int *)d_one, 1, c->nmb, 1);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(d_one);
        s.packed = false;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mbs) HIPCHK(c, hipMemcpy(mbs, s.d_mbs, (size_t)c->nmb * sizeof(vp8ir_mb), hipMemcpyDeviceToHost));
    if (coef) HIPCHK(c, hipMemcpy(coef, s.d_coef, (size_t)c->nmb * VP8IR_COEF_PER_MB * sizeof(int16_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" size_t vp8hip_frame_stride(const vp8hip_ctx *c) { return c ? c->fb_stride : 0; }

extern "C" int vp8hip_frames_fetch_async(vp8hip_ctx *c, int first_fb, int count, uint8_t *dst, uint8_t *digests)
{
    if (!c || first_fb < 0 || count < 1 || first_fb + count > (int)c->fb.size() || (!dst && !digests))
        return fail(c, -2, "vp8hip_frames_fetch_async: bad arguments");
    if (digests && (c->width & 127))
        return fail(c, -3, "vp8hip_frames_fetch_async: digests on the device need a display width that is a multiple of 128 (%d)", c->width);
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    if (!c->stream_d2h) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_d2h, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_from, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_done, hipEventDisableTiming));
    }
    if (c->d2h_count) HIPCHK(c, hipEventSynchronize(c->ev_d2h_done));  // one copy in flight at a time
    HIPCHK(c, hipEventRecord(c->ev_d2h_from, c->stream));             // everything queued so far: the frames' kernels
    HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_d2h_from, 0));
    if (dst) HIPCHK(c, hipMemcpyAsync(dst, c->fb[first_fb], c->fb_stride * (size_t)count, hipMemcpyDeviceToHost, c->stream_d2h));
    if (digests) {
        if (c->md5_cap < count) {
            HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
            if (c->d_md5) (void)hipFree(c->d_md5);
            c->d_md5 = nullptr; c->md5_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_md5, 16 * (size_t)(count < 64 ? 64 : count)));
            c->md5_cap = count < 64 ? 64 : count;
        }
        // a frame per lane: the frames' hashes run side by side, behind the copy of the frames themselves (if asked for)
        hipLaunchKernelGGL(vp8_md5_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, c->stream_d2h, (const uint8_t *)c->fb[first_fb],
                           c->fb_stride, count, c->dg, c->width, c->height, c->d_md5);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(digests, c->d_md5, 16 * (size_t)count, hipMemcpyDeviceToHost, c->stream_d2h));
    }
    HIPCHK(c, hipEventRecord(c->ev_d2h_done, c->stream_d2h));
    c->d2h_first = first_fb; c->d2h_count = count;
    return 0;
}

extern "C" int vp8hip_frames_download_async(vp8hip_ctx *c, int first_fb, int count, uint8_t *dst)
{
    if (!dst) return fail(c, -2, "vp8hip_frames_download_async: bad arguments");
    return vp8hip_frames_fetch_async(c, first_fb, count, dst, nullptr);
}

extern "C" int vp8hip_download_wait(vp8hip_ctx *c)
{
    if (!c) return -2;
    if (!c->d2h_count) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev_d2h_done));
    c->d2h_count = 0;
    return check_status(c);
}

extern "C" void *vp8hip_host_alloc(vp8hip_ctx *c, size_t bytes)
{
    void *p = nullptr;
    if (!c || !bytes) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { fail(c, -1, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

extern "C" void vp8hip_host_free(vp8hip_ctx *c, void *p)
{
    if (c && p) { (void)hipSetDevice(c->device); (void)hipHostFree(p); }
}

extern "C" int vp8hip_frame_upload(vp8hip_ctx *c, int fb, const uint8_t *buf)
{
    if (!c || fb < 0 || fb >= (int)c->fb.size() || !buf) return fail(c, -2, "vp8hip_frame_upload: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    HIPCHK(c, hipMemcpyAsync(c->fb[fb], buf, (size_t)c->geom.frame_size, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int vp8hip_frame_copy(vp8hip_ctx *c, int dst, int src)
{
    if (!c || dst < 0 || src < 0 || dst >= (int)c->fb.size() || src >= (int)c->fb.size())
        return fail(c, -2, "vp8hip_frame_copy: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (join_detile(c)) return -1;
    HIPCHK(c, hipMemcpyAsync(c->fb[dst], c->fb[src], (size_t)c->geom.frame_size, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

#ifdef VP8_STAMPS
extern "C" int vp8hip_debug_sched(vp8hip_ctx *c, unsigned int *out, int nwords)
{
    if (!c || !out || !c->d_sched || nwords > VP8HIP_SCHED_WORDS) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_sched, sizeof(unsigned int) * nwords, hipMemcpyDeviceToHost));
    return 0;
}
// diagnostic builds only (see vp8_common.hip.h): read and clear the stamp buckets; which = 0 recon, 1 loop filter
__device__ unsigned long long vp8_stamps_recon[VP8_NSTAMPS], vp8_stamps_lf[VP8_NSTAMPS];
extern "C" int vp8hip_debug_stamps(vp8hip_ctx *c, int which, unsigned long long *out)
{
    if (!c || !out) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long zero[VP8_NSTAMPS] = { 0 };
    if (which == 0) { HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(vp8_stamps_recon), sizeof zero)); HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(vp8_stamps_recon), zero, sizeof zero)); }
    else { HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(vp8_stamps_lf), sizeof zero)); HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(vp8_stamps_lf), zero, sizeof zero)); }
    return 0;
}
#endif
