// vp8hip_decode (include/vp8hip.h): which kernels a launch of independent frames runs, and the two forms a frame buffer has on
// the device.  What it stands for in the reference: decode_mb_row x mb_rows (vp8/decoder/decodframe.c:1116-1129),
// vp8_loop_filter_frame (vp8/common/loopfilter.c:203) and vp8_yv12_extend_frame_borders_ptr (vp8/decoder/onyxd_if.c:607) for
// every frame of the launch.
//
// TWO kernel families, one choice (launch_regime below):
//   * large launches -- more than two frames per CU, reconstruction and loop filter both wanted -- run the LANE-PER-ROW kernels:
//     vp8_keyframe_kernel (key frames only), or vp8_inter_pred_kernel + vp8_interframe_kernel (inter frames among them), which
//     leave a frame as macroblock-window tiles (vp8_keyframe_simt.hip: the form in which a lane can write whole 64-byte sectors);
//   * everything else -- few frames, one frame (a single stream through vpx_codec_decode), single stages -- runs the WAVE-PER-ROW
//     kernels (vp8_recon.hip, vp8_loopfilter.hip), which write the raster frame buffers: up to 128 frame pairs spread
//     over several CUs each (the _xcu variants), more than that one pair per workgroup; launches with inter frames of up to
//     VP8HIP_INTER_SPLIT frames reconstruct their inter macroblocks first, every one on its own (vp8_inter_mb_kernel).
// Every kernel reads the IR slots in the device form of include/vp8_ir.h, as the producers wrote them: nothing is converted here.
//
// TWO forms of a frame buffer (vp8hip_ctx::fb_state): the RASTER form is the reference's YV12 layout with its borders
// (vpx_scale/generic/yv12config.c:55-112) -- what the wave-per-row kernels read and write, what inter prediction reads, what
// vp8hip_frame_download and the post-processing filters see --, the TILED form is what a large launch leaves.  A frame is
// converted (vp8_detile_kf_kernel + vp8_extend_kernel, one pass over the frame) when something asks for the form it is not in
// (vp8hip_need_raster), not after every launch: the MD5 kernel reads tiles, a batch download writes raster rows straight into
// the caller's page-locked memory (vp8hip_frames_fetch_async), so a pipeline of large launches never pays the pass in HBM.
// Through round 3 the pass ran after every large launch, beside the next one: 38 % of a step's HBM traffic.
#include "vp8hip_ctx.hip.h"

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
extern "C" __global__ void vp8_inter_pred_tiles_kernel(const DevJob *jobs, int njobs, DevGeom g, int upf);
extern "C" __global__ void vp8_loopfilter_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_extend_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_detile_kf_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_retile_kernel(const DevJob *jobs, int njobs, DevGeom g);

// The raster form of the frame buffers fbs[0..n-1], for those that only exist as tiles: the tiled -> raster pass and the borders
// (vp8_yv12_extend_frame_borders, onyxd_if.c:607), on the context's stream.
static int tile_pool(vp8hip_ctx *c);
// to_tiles: the other direction -- the tiled form of frame buffers that only exist in raster form (vp8_retile_kernel), for a launch
// that reads its references as tiles
static int convert_list(vp8hip_ctx *c, const int *fbs, int n, bool to_tiles)
{
    const uint8_t lacks = to_tiles ? FB_RASTER : FB_TILES;        // the state of a frame that has only the other form
    int m = 0;
    for (int i = 0; i < n; i++) m += c->fb_state[(size_t)fbs[i]] == lacks;
    if (!m) return 0;
    if (to_tiles ? tile_pool(c) : vp8hip_raster_pool(c)) return -1;
    if (m > c->conv_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_conv_jobs) (void)hipFree(c->d_conv_jobs);
        if (c->h_conv_jobs) (void)hipHostFree(c->h_conv_jobs);
        c->d_conv_jobs = c->h_conv_jobs = nullptr; c->conv_cap = 0;
        const int cap = m < 64 ? 64 : m;
        HIPCHK(c, hipMalloc((void **)&c->d_conv_jobs, sizeof(DevJob) * (size_t)cap));
        HIPCHK(c, hipHostMalloc((void **)&c->h_conv_jobs, sizeof(DevJob) * (size_t)cap, hipHostMallocDefault));
        c->conv_cap = cap;
    } else
        HIPCHK(c, hipEventSynchronize(c->ev_conv));           // the table of the pass before has been copied
    m = 0;
    for (int i = 0; i < n; i++) {
        const int f = fbs[i];
        if (c->fb_state[(size_t)f] != lacks) continue;
        DevJob &d = c->h_conv_jobs[m++];
        memset(&d, 0, sizeof d);
        d.dst = c->fb[(size_t)f]; d.tile = c->fb_tiles[(size_t)f];
        c->fb_state[(size_t)f] = FB_TILES | FB_RASTER;          // (a frame buffer named twice converts once)
    }
    HIPCHK(c, hipMemcpyAsync(c->d_conv_jobs, c->h_conv_jobs, sizeof(DevJob) * (size_t)m, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_conv, c->stream));
    long units = (long)c->dg.mb_rows * m;
    if (units > 8L * c->num_cu) units = 8L * c->num_cu;
    if (to_tiles) {
        hipLaunchKernelGGL(vp8_retile_kernel, dim3((unsigned)units), dim3(256), 0, c->stream, (const DevJob *)c->d_conv_jobs, m, c->dg);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(vp8_detile_kf_kernel, dim3((unsigned)units), dim3(256), 0, c->stream, (const DevJob *)c->d_conv_jobs, m, c->dg);
    int bx = (c->geom.aligned_h + 64) / 4;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(vp8_extend_kernel, dim3(bx, m), dim3(256), 0, c->stream, (const DevJob *)c->d_conv_jobs, m, c->dg);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int vp8hip_need_raster_list(vp8hip_ctx *c, const int *fbs, int n) { return convert_list(c, fbs, n, false); }
int vp8hip_need_raster(vp8hip_ctx *c, int first, int count)
{
    bool any = false;
    for (int i = 0; i < count && !any; i++) any = c->fb_state[(size_t)(first + i)] == FB_TILES;
    if (!any) return 0;
    std::vector<int> v((size_t)count);
    for (int i = 0; i < count; i++) v[(size_t)i] = first + i;
    return vp8hip_need_raster_list(c, v.data(), count);
}
// (kept for callers of round 3's interface: there is no second stream to join any more -- conversions run on the context's stream)
extern "C" int vp8hip_join(vp8hip_ctx *c) { return c ? 0 : -2; }

// The ONE place that decides which kernels a launch runs (header of this file).  The threshold is where the families cross on an
// MI355X: the wave-per-row kernels are the faster ones up to one frame pair per CU (512 frames: 126 vs 85 Gpix/s at 1080p); beyond
// that they need a second round of workgroups and the lane-per-row kernels win (640 frames: 108 vs 85).
enum Regime { WAVE_PER_ROW = 0, LANE_KEY, LANE_INTER };
static Regime launch_regime(const vp8hip_ctx *c, int njobs, int stages, bool all_key)
{
    const bool both = (stages & VP8HIP_STAGE_RECON) && (stages & VP8HIP_STAGE_LF);
    bool lane = both && njobs > 2 * c->num_cu;
    // a context whose frames have only ever been tiles (the raster pool was never needed: vp8hip_raster_pool) stays with the
    // kernels that write tiles for the odd small launch of key frames -- the tail of a run -- rather than allocate the pool for it
    if (both && all_key && c->tile_block && !c->fb_block) lane = true;
    if (c->knobs.recon_force) lane = both && c->knobs.recon_force == 1;       // tuning / test knob: VP8HIP_RECON
    return !lane ? WAVE_PER_ROW : all_key ? LANE_KEY : LANE_INTER;
}

// the tiled forms of all frame buffers, with the first large launch: tile_frame bytes each (one tile per macroblock and one more
// per macroblock row, 32 bytes of unfiltered line per tile behind them: vp8_keyframe_simt.hip), + 8 KB: the dummy tile idle lanes
// write, and room for the kernels' prefetches past the last tile; VP8HIP_TILE_FRONT bytes in front: a strip of
// vp8_inter_pred_tiles_kernel that begins left of the frame loads from the tile before the row's first
static int tile_pool(vp8hip_ctx *c)
{
    if (c->tile_block) return 0;
    const int nfb = (int)c->fb.size();
    hipError_t e = hipMalloc((void **)&c->tile_alloc, c->tile_frame * (size_t)nfb + 8192 + VP8HIP_TILE_FRONT);
    if (e != hipSuccess && vp8hip_drop_staging(c) == 1) {       // (the packed staging is a cache: vp8hip.hip)
        (void)hipGetLastError();
        e = hipMalloc((void **)&c->tile_alloc, c->tile_frame * (size_t)nfb + 8192 + VP8HIP_TILE_FRONT);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->tile_alloc = nullptr;
        return fail(c, -1, "no device memory for the tiled form of %d frame buffers (%zu MB)", nfb, c->tile_frame * (size_t)nfb >> 20);
    }
    c->tile_block = c->tile_alloc + VP8HIP_TILE_FRONT;
    c->fb_tiles.resize((size_t)nfb);
    for (int i = 0; i < nfb; i++) c->fb_tiles[(size_t)i] = c->tile_block + c->tile_frame * (size_t)i;
    return 0;
}

// (include/vp8hip.h) the pools a pipeline is going to need, now -- while a first launch that needs neither is running, say: tens of
// GB take the allocator a second or two
extern "C" int vp8hip_reserve(vp8hip_ctx *c, int tiled_form, int raster_form)
{
    if (!c || !c->width) return fail(c, -2, "vp8hip_reserve: context not configured");
    HIPCHK(c, hipSetDevice(c->device));
    if (tiled_form && tile_pool(c)) return -1;
    if (raster_form && vp8hip_raster_pool(c)) return -1;
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
        for (int k = 0; k < VP8HIP_NBUF; k++) {
            if (c->d_jobs2[k]) (void)hipFree(c->d_jobs2[k]);
            if (c->h_jobs2[k]) (void)hipHostFree(c->h_jobs2[k]);
            c->d_jobs2[k] = nullptr; c->h_jobs2[k] = nullptr;
        }
        c->jobs_cap = njobs < 64 ? 64 : njobs;
        for (int k = 0; k < VP8HIP_NBUF; k++) {
            HIPCHK(c, hipMalloc((void **)&c->d_jobs2[k], sizeof(DevJob) * c->jobs_cap));
            HIPCHK(c, hipHostMalloc((void **)&c->h_jobs2[k], sizeof(DevJob) * c->jobs_cap, hipHostMallocDefault));
        }
    } else {
        // this call's staging was read by the copy of the call VP8HIP_NBUF calls ago; wait for that copy only
        HIPCHK(c, hipEventSynchronize(c->ev_jobs2[c->parity]));
    }
    const int nfb = (int)c->fb.size(), nsl = (int)c->slots.size();
    bool any_lf = false;
    bool all_key = true;
    for (int i = 0; i < njobs && all_key; i++)
        if (jobs[i].ir_slot >= 0 && jobs[i].ir_slot < nsl) all_key = c->slots[jobs[i].ir_slot].hdr_copy.frame_type == 0;
    const Knobs &K = c->knobs;
    const Regime regime = launch_regime(c, njobs, stages, all_key);
    const bool tiled = regime != WAVE_PER_ROW, inter_fused = regime == LANE_INTER;
    const int par = c->parity;
    c->parity = (par + 1) % VP8HIP_NBUF;
    for (int i = 0; i < njobs; i++)
        if (jobs[i].ir_slot < 0 || jobs[i].ir_slot >= nsl || jobs[i].dst_fb < 0 || jobs[i].dst_fb >= nfb)
            return fail(c, -2, "vp8hip_decode: job %d has slot %d / fb %d out of range", i, jobs[i].ir_slot, jobs[i].dst_fb);
    if (tiled && tile_pool(c)) return -1;
    // Inter prediction reads its reference frames in the form they are in: a large launch whose references all have a raster form
    // reads that (vp8_inter_pred_kernel: a row is one load); one with a reference that exists only as tiles -- streams decoded in
    // lock step: what the launch before left -- reads ALL its references as tiles (vp8_inter_pred_tiles_kernel), no tiled ->
    // raster pass runs, and those of its references that exist only in raster form are retiled once (they keep both forms); a
    // small launch gets the raster form of the references that lack it first.
    bool pred_tiles = false;
    {
        // what this launch reads as raster: its reference frames (borders included), and -- a wave-per-row launch of the loop filter
        // alone -- the frames it filters in place
        std::vector<int> need;
        bool pool = !tiled || K.eager_raster;               // (a large launch of key frames writes tiles only)
        std::vector<int> lack_tiles;                         // references that only exist in raster form
        // (frames one macroblock wide: a chroma strip reaches past BOTH vertical edges, and TileSrc replicates one)
        bool may_tiles = inter_fused && K.pred_tiles && c->tile_block && c->geom.aligned_w >= 32;
        for (int i = 0; i < njobs; i++) {
            if (c->slots[jobs[i].ir_slot].hdr_copy.frame_type != 0) {
                for (int k = 1; k < 4; k++) {
                    const int f = jobs[i].ref_fb[k];
                    if (f < 0 || f >= nfb) continue;
                    if (c->fb_state[(size_t)f] == FB_RASTER) lack_tiles.push_back(f);
                    if (c->fb_state[(size_t)f] == FB_TILES) need.push_back(f);
                }
            }
            if (!tiled && !(stages & VP8HIP_STAGE_RECON) && c->fb_state[(size_t)jobs[i].dst_fb] == FB_TILES) need.push_back(jobs[i].dst_fb);
        }
        // tiles where a reference would have to be converted to raster first (or: always, VP8HIP_PRED_TILES=2); the references that
        // have no tiled form -- a golden frame a small launch decoded long ago, an uploaded one -- get one (vp8_retile_kernel: once
        // per such frame, they keep both forms)
        pred_tiles = !all_key && may_tiles && (!need.empty() || K.pred_tiles == 2);
        if (pred_tiles) {
            need.clear();
            if (!c->fb_block) lack_tiles.clear();           // (no raster form has ever been made: such a reference was never decoded at all)
            if (!lack_tiles.empty() && convert_list(c, lack_tiles.data(), (int)lack_tiles.size(), true)) return -1;
        } else if (!all_key) pool = true;
        if (pool && vp8hip_raster_pool(c)) return -1;
        if (!need.empty() && vp8hip_need_raster_list(c, need.data(), (int)need.size())) return -1;
    }
    c->d_jobs = c->d_jobs2[par]; c->h_jobs = c->h_jobs2[par];
    for (int i = 0; i < njobs; i++) {
        const vp8hip_job &j = jobs[i];
        const Slot &s = c->slots[j.ir_slot];
        DevJob &d = c->h_jobs[i];
        d.hdr = s.hdr_copy;
        d.mbx = s.d_mbx; d.blocks = s.d_blocks; d.mvs = s.d_mvs;
        d.dst = c->fb[j.dst_fb];
        d.ref[0] = nullptr;
        d.tile = tiled ? c->fb_tiles[(size_t)j.dst_fb] : nullptr;
        for (int k = 1; k < 4; k++) {
            d.ref[k] = nullptr; d.ref_tile[k - 1] = nullptr;
            if (s.hdr_copy.frame_type == 0) continue;          // key frames read no reference
            int f = j.ref_fb[k];
            if (f >= nfb) return fail(c, -2, "vp8hip_decode: job %d ref %d out of range", i, f);
            if (f < 0) return fail(c, -2, "vp8hip_decode: inter frame job %d lacks reference %d", i, k);
            if (f == j.dst_fb) return fail(c, -2, "vp8hip_decode: job %d decodes into its own reference", i);
            d.ref[k] = c->fb.empty() || !c->fb_block ? nullptr : c->fb[f];
            d.ref_tile[k - 1] = c->fb_tiles.empty() ? nullptr : c->fb_tiles[(size_t)f];
        }
        any_lf |= s.hdr_copy.filter_level != 0;
    }
    if (c->d2h_count) {      // a batch download still in flight: a launch that writes one of its frame buffers waits for it
        bool hit = false;
        for (int i = 0; i < njobs && !hit; i++)
            hit = c->d2h_listed ? c->d2h_mask[(size_t)jobs[i].dst_fb] != 0 : jobs[i].dst_fb >= c->d2h_first && jobs[i].dst_fb < c->d2h_first + c->d2h_count;
        if (hit) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_d2h_done, 0));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, c->h_jobs, sizeof(DevJob) * njobs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_jobs2[par], c->stream));

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
    // "one MB row per lane" kernels: G lanes per strand of frames, row period P = max(cols, 2G + 2).  A strand's frames follow each
    // other row after row, G rows at a time, so a launch takes ceil(frames per strand x rows / G) rounds of P steps + 2 (G - 1) to
    // fill and drain -- 8192 1080p frames on 8 lanes each: 68 rows = 8.5 rounds, nine run; 16,384: two frames per strand, 17 whole
    // rounds -- and every strand's first lane goes through the hand-over tile (a little more per step the more strands a wave has:
    // 16,384 frames at G = 4, 8, 16: 58.1, 57.5, 58.2 ms).  G = the one that costs the fewest steps by that count.
    int lgG = 1;
    {
        const int cols = c->dg.mb_cols, rows = c->dg.mb_rows;
        const long lanes = (long)c->num_cu * 4 * 64;                      // one luma wave per SIMD
        double best = 0;
        for (int k = 1; k <= 6; k++) {
            const long G = 1L << k, strands = lanes >> k;
            const long per = (njobs + strands - 1) / strands;             // frames per strand (strands without a frame idle)
            const long P = cols > 2 * G + 2 ? cols : 2 * G + 2;
            const double steps = (double)((per * rows + G - 1) / G * P + 2 * (G - 1)) * (1.0 + 0.09 / (double)G);
            if (k == 1 || steps < best) { best = steps; lgG = k; }
        }
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
    if (tiled) { c->stats.workgroups = simt_waves; c->stats.recon_waves = 1; c->stats.lf_waves = 1; }
    c->stats.detile_pass = tiled && K.eager_raster;
    c->stats.pred_tiles = pred_tiles;

    if (stages & VP8HIP_STAGE_RECON) {
        if (tiled) {
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
                hipLaunchKernelGGL(pred_tiles ? vp8_inter_pred_tiles_kernel : vp8_inter_pred_kernel, dim3((unsigned)pgrid), dim3(256), 0, c->stream,
                                   (const DevJob *)c->d_jobs, njobs, c->dg, upf);
                hipLaunchKernelGGL(vp8_interframe_kernel, dim3(2 * simt_waves), dim3(64), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                                   c->dg, lgG, simtP, simt_waves * spw, c->tile_block + c->tile_frame * (size_t)nfb + 4096,
                                   c->d_sched, simt_waves);
            } else
                hipLaunchKernelGGL(vp8_keyframe_kernel, dim3(2 * simt_waves), dim3(64), 0, c->stream, (const DevJob *)c->d_jobs, njobs,
                                   c->dg, lgG, simtP, simt_waves * spw, c->tile_block + c->tile_frame * (size_t)nfb + 4096,
                                   c->d_sched, simt_waves);
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
    c->stats.fused = tiled;
    c->stats.lf_kernels = 0;
    if ((stages & VP8HIP_STAGE_LF) && any_lf && !tiled) {
        c->stats.lf_kernels = 1;
        const int npairs = (njobs + 1) / 2;          // the loop filter works on two frames per wave
        if (xcu_S > 1) {
            hipLaunchKernelGGL(vp8_loopfilter_xcu_kernel, dim3(xcu_grid), dim3(64 * XCU_NW), 256 + XCU_NW * 2 * 4096, c->stream,
                               (const DevJob *)c->d_jobs, njobs, c->dg, c->gran_lf, c->epoch, xcu_S, c->d_status);
        } else {
            const int lfgrid = npairs < c->num_cu * wg_per_cu ? npairs : c->num_cu * wg_per_cu;
            hipLaunchKernelGGL(vp8_loopfilter_kernel, dim3(lfgrid), dim3(64 * c->lf_nw), c->lf_lds, c->stream,
                               (const DevJob *)c->d_jobs, njobs, c->dg);
        }
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(ev[2], c->stream));
    if (tiled) {
        // the frames are there as tiles; their raster form when somebody asks (or at once: VP8HIP_EAGER_RASTER)
        std::vector<int> dsts((size_t)njobs);
        for (int i = 0; i < njobs; i++) { c->fb_state[(size_t)jobs[i].dst_fb] = FB_TILES; dsts[(size_t)i] = jobs[i].dst_fb; }
        HIPCHK(c, hipEventRecord(ev[4], c->stream));
        if (K.eager_raster && vp8hip_need_raster_list(c, dsts.data(), njobs)) return -1;
        HIPCHK(c, hipEventRecord(ev[5], c->stream));
    } else if (stages & VP8HIP_STAGE_EXTEND) {
        int bx = (c->geom.aligned_h + 64) / 4;
        if (bx < 1) bx = 1;
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(vp8_extend_kernel, dim3(bx, njobs), dim3(256), 0, c->stream, (const DevJob *)c->d_jobs,
                           njobs, c->dg);
        HIPCHK(c, hipGetLastError());
    }
    if (!tiled && (stages & (VP8HIP_STAGE_RECON | VP8HIP_STAGE_LF | VP8HIP_STAGE_EXTEND)))
        for (int i = 0; i < njobs; i++) c->fb_state[(size_t)jobs[i].dst_fb] = FB_RASTER;
    HIPCHK(c, hipEventRecord(ev[3], c->stream));
    c->evr_tiled[c->ncalls % VP8HIP_STATS_RING] = tiled;
    c->evr_stats[c->ncalls % VP8HIP_STATS_RING] = c->stats;
    c->ncalls++;
    return 0;
}
