// C-ABI shim of the gfx950 VP8 pixel path (include/vp8hip.h): context, device pools, IR slots, frame buffers.  Owns the HIP
// stream, the device frame-buffer pool (the decoder's yv12_fb[] lives in HBM) and the IR slots with their pinned host staging.
// The launches are in vp8hip_launch.hip (pixel path), vp8hip_entropy.hip and vp8hip_postproc.hip.  No CPU fallback: every
// failure is reported.
#include "vp8hip_ctx.hip.h"

extern "C" __global__ void vp8_recon_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_recon_xcu_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                                                int S, int *err);
extern "C" __global__ void vp8_loopfilter_xcu_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned long long *gran,
                                                     unsigned int epoch, int S, int *err);
extern "C" __global__ void vp8_recon_intra_kernel(const DevJob *jobs, int njobs, DevGeom g, const unsigned int *intra_flags);
extern "C" __global__ void vp8_recon_intra_xcu_kernel(const DevJob *jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                                                      int S, int *err, const unsigned int *intra_flags);
extern "C" __global__ void vp8_loopfilter_kernel(const DevJob *jobs, int njobs, DevGeom g);
extern "C" __global__ void vp8_md5_kernel(const uint8_t *frames, size_t fstride, const int *index, int first, int count, DevGeom g, int w, int h,
                                          uint8_t *out);
extern "C" __global__ void vp8_md5_any_kernel(const uint8_t *frames, size_t fstride, const int *index, int first, int count, DevGeom g, int w, int h,
                                              uint8_t *out);
extern "C" __global__ void vp8_md5_tiles_kernel(const uint8_t *tiles, size_t tstride, const int *index, int first, int count, DevGeom g, int w,
                                                int h, uint8_t *out);
extern "C" __global__ void vp8_detile_run_kernel(const uint8_t *tiles, size_t tstride, uint8_t *dst, size_t dstride, int count, DevGeom g);
extern "C" __global__ void vp8_pack_i420_tiles_kernel(const uint8_t *tiles, size_t tstride, uint8_t *dst, size_t dstride, int count, DevGeom g, int w, int h);
extern "C" __global__ void vp8_pack_i420_raster_kernel(const uint8_t *frames, size_t fstride, int first, uint8_t *dst, size_t dstride, int count, DevGeom g,
                                                       int w, int h);

static char g_create_error[256] = "";

static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
static void read_knobs(Knobs &k)
{
    const char *e = getenv("VP8HIP_RECON");
    k.recon_force = !e ? 0 : !strcmp(e, "simt") ? 1 : !strcmp(e, "wave") ? 2 : 0;
    k.pred_tiles = env_int("VP8HIP_PRED_TILES", 1);
    k.md5_pack_from = env_int("VP8HIP_MD5_PACK_FROM", 12288);
    k.lgG = env_int("VP8HIP_SIMT_LGG", 0);
    k.simt_waves = env_int("VP8HIP_SIMT_WAVES", 0);
    k.wg_per_cu = env_int("VP8HIP_WG_PER_CU", 1);
    k.xcu = env_int("VP8HIP_XCU", 1) != 0;
    k.xcu_S = env_int("VP8HIP_XCU_S", 0);
    k.eager_raster = env_int("VP8HIP_EAGER_RASTER", 0) != 0;
    k.direct_download = env_int("VP8HIP_DIRECT_DOWNLOAD", 0) != 0;
    k.download_blocks = env_int("VP8HIP_DOWNLOAD_BLOCKS", 0);
    k.d2h_prio = env_int("VP8HIP_D2H_PRIO", 1);
    k.d2h_streams = env_int("VP8HIP_D2H_STREAMS", 2);      // (1080p, every frame downloaded, the hash beside the copy: 13.3-13.4 k frames/s = 46 GB/s with one to four)
    if (k.d2h_streams < 1 || k.d2h_streams > 4) k.d2h_streams = 1;
    k.inter_split = env_int("VP8HIP_INTER_SPLIT", 384);
    k.xcu_NW = env_int("VP8HIP_XCU_NW", 0);
    k.recon_nw = env_int("VP8HIP_RECON_NW", 0);
    k.lf_nw = env_int("VP8HIP_LF_NW", 0);
}

__device__ unsigned int vp8_gran_broken;      // see gran_wait (vp8_common.hip.h)

int vp8hip_fail(vp8hip_ctx *c, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_error, 256, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *vp8hip_last_error(const vp8hip_ctx *ctx) { return ctx ? ctx->err : g_create_error; }

static void free_pools(vp8hip_ctx *c)
{
    if (c->fb_block) (void)hipFree(c->fb_block);
    if (c->tile_alloc) (void)hipFree(c->tile_alloc);
    c->tile_block = c->tile_alloc = nullptr;
    if (c->slot_block_dev) (void)hipFree(c->slot_block_dev);
    if (c->pool) (void)hipFree(c->pool);
    if (c->d_pool_ctr) (void)hipFree(c->d_pool_ctr);
    c->pool = nullptr; c->d_pool_ctr = nullptr; c->pool_chunks = c->chunk_blocks = 0;
    if (c->gran_recon) (void)hipFree(c->gran_recon);
    if (c->gran_lf) (void)hipFree(c->gran_lf);
    if (c->d_intra_flags) (void)hipFree(c->d_intra_flags);
    c->d_intra_flags = nullptr; c->intra_flags_cap = 0;
    c->gran_recon = c->gran_lf = nullptr; c->gran_recon_cap = c->gran_lf_cap = 0;
    for (Slot &s : c->slots) {
        if (s.h_block) (void)hipHostFree(s.h_block);
        free(s.h_dense);
    }
    c->fb_block = nullptr; c->slot_block_dev = nullptr;
    c->fb.clear(); c->fb_tiles.clear(); c->fb_state.clear(); c->slots.clear();
}

static void destroy_events(vp8hip_ctx *c)
{
    for (int r = 0; r < VP8HIP_STATS_RING; r++) for (int i = 0; i < 6; i++) if (c->evr[r][i]) (void)hipEventDestroy(c->evr[r][i]);
    for (int k = 0; k < VP8HIP_NBUF; k++) if (c->ev_jobs2[k]) (void)hipEventDestroy(c->ev_jobs2[k]);
    if (c->ev_conv) (void)hipEventDestroy(c->ev_conv);
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
    c->tile_block = c->tile_alloc = nullptr; c->tile_frame = 0;
    c->d_jobs = nullptr; c->h_jobs = nullptr; c->jobs_cap = 0;
    for (int k = 0; k < VP8HIP_NBUF; k++) { c->d_jobs2[k] = nullptr; c->h_jobs2[k] = nullptr; c->ev_jobs2[k] = nullptr; }
    c->parity = 0;
    c->d_md5 = nullptr; c->md5_cap = 0;
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
    c->stream_d2h = nullptr; c->ev_d2h_from = c->ev_d2h_done = nullptr; c->d2h_first = c->d2h_count = 0; c->fb_stride = 0;
    // events: every creation is checked; on failure whatever exists is destroyed again (null handles are skipped)
    for (int r = 0; r < VP8HIP_STATS_RING; r++) for (int i = 0; i < 6; i++) c->evr[r][i] = nullptr;
    c->ev_conv = nullptr;
    e = hipSuccess;
    for (int r = 0; r < VP8HIP_STATS_RING && e == hipSuccess; r++)
        for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&c->evr[r][i]);
    for (int k = 0; k < VP8HIP_NBUF && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->ev_jobs2[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_conv, hipEventDisableTiming);
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


extern "C" void vp8hip_destroy(vp8hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_pools(c);
    if (c->d_conv_jobs) (void)hipFree(c->d_conv_jobs);
    if (c->h_conv_jobs) (void)hipHostFree(c->h_conv_jobs);
    for (int k = 0; k < VP8HIP_NBUF; k++) if (c->d_jobs2[k]) (void)hipFree(c->d_jobs2[k]);
    for (int k = 0; k < VP8HIP_NBUF; k++) if (c->h_jobs2[k]) (void)hipHostFree(c->h_jobs2[k]);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->d_sched) (void)hipFree(c->d_sched);
    // the post-processing tables outlive reconfigurations: the caller's noise state does too (vp8/common/postproc.c keeps
    // postproc_state.noise across vp8_alloc_frame_buffers) and only sends the noise table again when q changes
    if (c->d_pp) (void)hipFree(c->d_pp);
    if (c->d_md5) (void)hipFree(c->d_md5);
    if (c->d_md5_idx) (void)hipFree(c->d_md5_idx);
    if (c->h_md5_idx) (void)hipHostFree(c->h_md5_idx);
    if (c->h_pp) (void)hipHostFree(c->h_pp);
    if (c->ev_pp) (void)hipEventDestroy(c->ev_pp);
    for (int k = 0; k < 2; k++) {
        if (c->d_ent_frames2[k]) (void)hipFree(c->d_ent_frames2[k]);
        if (c->d_ent_data2[k]) (void)hipFree(c->d_ent_data2[k]);
        if (c->ev_ent_in[k]) (void)hipEventDestroy(c->ev_ent_in[k]);
        if (c->ev_ent_out[k]) (void)hipEventDestroy(c->ev_ent_out[k]);
    }
    if (c->stream_h2d) (void)hipStreamDestroy(c->stream_h2d);
    if (c->d_i420) (void)hipFree(c->d_i420);
    if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
    for (int k = 0; k < 3; k++) {
        if (c->stream_d2h_more[k]) (void)hipStreamDestroy(c->stream_d2h_more[k]);
        if (c->ev_d2h_more[k]) (void)hipEventDestroy(c->ev_d2h_more[k]);
    }
    if (c->d_ent_scratch) (void)hipFree(c->d_ent_scratch);
    if (c->d_ent_status) (void)hipFree(c->d_ent_status);
    if (c->d_mfqe) (void)hipFree(c->d_mfqe);
    if (c->h_mfqe) (void)hipHostFree(c->h_mfqe);
    if (c->ev_mfqe) (void)hipEventDestroy(c->ev_mfqe);
    destroy_events(c);
    if (c->stream_d2h) { (void)hipStreamSynchronize(c->stream_d2h); (void)hipStreamDestroy(c->stream_d2h); }
    if (c->ev_d2h_from) (void)hipEventDestroy(c->ev_d2h_from);
    if (c->ev_d2h_done) (void)hipEventDestroy(c->ev_d2h_done);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

// per-wave LDS footprints; must match the kernels (WaveLds 2080 B + line slot, LfWaveLds 768 B)
static size_t recon_lds_bytes(int nw, int aligned_w) { return 1024 + (size_t)nw * 2 * (2080 + 2 * aligned_w + 96); }   // two frames per wave
static size_t lf_lds_bytes(int nw) { return 256 + (size_t)nw * 2 * 4096; }   // two frames per wave

static int configure_pools(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots, size_t pool_bytes);

extern "C" int vp8hip_configure(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots)
{
    const int rc = configure_pools(c, width, height, num_fb, num_slots, 0);
    if (rc && c) {               // a failed (re)configuration leaves an UNconfigured context, not a half-allocated one
        free_pools(c);
        c->width = c->height = 0;
    }
    return rc;
}

extern "C" int vp8hip_configure_pooled(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots, size_t pool_bytes)
{
    if (!c || !pool_bytes) return fail(c, -2, "vp8hip_configure_pooled: bad arguments");
    const int rc = configure_pools(c, width, height, num_fb, num_slots, pool_bytes);
    if (rc) {
        free_pools(c);
        c->width = c->height = 0;
    }
    return rc;
}

extern "C" int vp8hip_pool_reset(vp8hip_ctx *c)
{
    if (!c || !c->pool) return fail(c, -2, "vp8hip_pool_reset: the context has no block pool (vp8hip_configure_pooled)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_pool_ctr, 0, sizeof(unsigned int), c->stream));
    return 0;
}

extern "C" int vp8hip_pool_usage(vp8hip_ctx *c, size_t *used_bytes, size_t *pool_bytes)
{
    if (!c || !c->pool) return fail(c, -2, "vp8hip_pool_usage: the context has no block pool (vp8hip_configure_pooled)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_pool_ctr, sizeof n, hipMemcpyDeviceToHost));
    if (used_bytes) *used_bytes = (size_t)n * c->chunk_blocks * 32;          // (more than the pool holds: that much was asked for)
    if (pool_bytes) *pool_bytes = (size_t)c->pool_chunks * c->chunk_blocks * 32;
    return 0;
}

static int configure_pools(vp8hip_ctx *c, int width, int height, int num_fb, int num_slots, size_t pool_bytes)
{
    if (!c) return -2;
    if (width <= 0 || height <= 0 || width > 16383 || height > 16383 || num_fb < 1 || num_slots < 1)
        return fail(c, -2, "vp8hip_configure: bad arguments %dx%d fb=%d slots=%d", width, height, num_fb, num_slots);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_pools(c);
    c->ent_staged.count = 0;                 // (input staged for the geometry before: dropped)
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

    // frame buffers: one block, each buffer 256-B aligned -- allocated with the first use of a raster form (vp8hip_raster_pool);
    // the tiled forms come with the first large launch (vp8hip_launch.hip)
    const size_t fbsz = align_up((size_t)g.frame_size, 256);
    c->fb.assign((size_t)num_fb, (uint8_t *)nullptr);
    c->fb_state.assign((size_t)num_fb, (uint8_t)FB_RASTER);
    c->tile_frame = align_up((size_t)c->dg.mb_rows * (c->dg.mb_cols + 1) * (VP8_TILE_BYTES + 32), 256);
    c->fb_stride = fbsz;
    if (c->stream_d2h) HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
    c->d2h_count = 0;


    // IR slots: [pad][mbx][blocks][mvs], the records on 128-byte boundaries (a cache line each); with a block pool the slots
    // have no block streams of their own
    const size_t o_mbx = 128, o_blocks = o_mbx + (size_t)c->nmb * sizeof(vp8ir_mbx);
    c->cap_blocks = pool_bytes ? 0 : (size_t)c->nmb * VP8IR_MAX_BLOCKS_PER_MB;
    if (pool_bytes) {
        // a chunk: four macroblock rows' worst case (a lane asks for the next one when what it has left would not hold a row's)
        c->chunk_blocks = 4u * (unsigned)c->dg.mb_cols * VP8IR_MAX_BLOCKS_PER_MB;
        const size_t chunk_bytes = (size_t)c->chunk_blocks * 32;
        if (pool_bytes / chunk_bytes < 2 || pool_bytes / chunk_bytes > 0xffffffffull / c->chunk_blocks)
            return fail(c, -2, "vp8hip_configure_pooled: a pool of %zu bytes (chunks of %zu; at most 2^32 blocks)", pool_bytes, chunk_bytes);
        c->pool_chunks = (unsigned)(pool_bytes / chunk_bytes) - 1;              // (the last chunk takes what no longer fits)
        HIPCHK(c, hipMalloc((void **)&c->pool, ((size_t)c->pool_chunks + 1) * chunk_bytes + 8192));
        HIPCHK(c, hipMalloc((void **)&c->d_pool_ctr, 256));
        HIPCHK(c, hipMemsetAsync(c->d_pool_ctr, 0, 256, c->stream));
    }
    const size_t o_mvs = align_up(o_blocks + c->cap_blocks * 32, 256);
    const size_t slotsz = align_up(o_mvs + (size_t)c->nmb * 16 * sizeof(vp8ir_mv), 256);
    HIPCHK(c, hipMalloc((void **)&c->slot_block_dev, slotsz * num_slots + 4096));   // + room for prefetches past the last macroblock
    c->slot_bytes = slotsz; c->o_mbx = o_mbx; c->o_blocks = o_blocks; c->o_mvs = o_mvs;
    c->slots.resize(num_slots);
    for (int i = 0; i < num_slots; i++) {
        char *d = c->slot_block_dev + slotsz * i;
        Slot &s = c->slots[i];
        s.d_mbx = (vp8ir_mbx *)(d + o_mbx); s.d_blocks = c->pool ? (int16_t *)c->pool : (int16_t *)(d + o_blocks); s.d_mvs = (vp8ir_mv *)(d + o_mvs);
        s.h_block = nullptr; s.h_hdr = nullptr; s.h_mbx = nullptr; s.h_blocks = nullptr; s.h_mvs = nullptr;
        s.h_dense = nullptr; s.h_mbs = nullptr; s.h_coef = nullptr;
        s.nblocks = 0;
        memset(&s.hdr_copy, 0, sizeof s.hdr_copy);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The staging of packed downloads -- which a digest-only fetch of a large batch also allocates and keeps, 3.1 MB per 1080p frame
// (fetch_impl) -- is a cache: given back when a pool that the context cannot do without finds no room.
int vp8hip_drop_staging(vp8hip_ctx *c)
{
    if (!c->d_i420) return 0;
    if (c->stream_d2h) HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
    for (int k = 0; k < 3; k++) if (c->stream_d2h_more[k]) HIPCHK(c, hipStreamSynchronize(c->stream_d2h_more[k]));
    (void)hipFree(c->d_i420);
    c->d_i420 = nullptr; c->i420_cap = 0;
    return 1;
}
extern "C" int vp8hip_release_staging(vp8hip_ctx *c)
{
    if (!c) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    return vp8hip_drop_staging(c) < 0 ? -1 : 0;
}

int vp8hip_raster_pool(vp8hip_ctx *c)
{
    if (c->fb_block || c->fb.empty()) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = c->fb.size();
    hipError_t e = hipMalloc((void **)&c->fb_block, c->fb_stride * n);
    if (e != hipSuccess && vp8hip_drop_staging(c) == 1) { (void)hipGetLastError(); e = hipMalloc((void **)&c->fb_block, c->fb_stride * n); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->fb_block = nullptr;
        return fail(c, -1, "no device memory for the raster form of %zu frame buffers (%zu MB)", n, c->fb_stride * n >> 20);
    }
    HIPCHK(c, hipMemsetAsync(c->fb_block, 0, c->fb_stride * n, c->stream));
    for (size_t i = 0; i < n; i++) c->fb[i] = c->fb_block + c->fb_stride * i;
    return 0;
}

extern "C" int vp8hip_memory_usage(const vp8hip_ctx *c, vp8hip_memory *out)
{
    if (!c || !out) return -2;
    out->raster_pool = c->fb_block ? c->fb_stride * c->fb.size() : 0;
    out->tile_pool = c->tile_block ? c->tile_frame * c->fb.size() + 8192 + VP8HIP_TILE_FRONT : 0;
    out->slots = c->slot_block_dev ? c->slot_bytes * c->slots.size() + 4096 : 0;
    out->block_pool = c->pool ? ((size_t)c->pool_chunks + 1) * c->chunk_blocks * 32 + 8192 : 0;
    out->entropy_input = c->ent_frames_cap2[0] + c->ent_frames_cap2[1] + c->ent_data_cap2[0] + c->ent_data_cap2[1];
    out->packed_staging = c->i420_cap;
    return 0;
}

extern "C" int vp8hip_geometry(const vp8hip_ctx *c, vp8ir_geom *g)
{
    if (!c || !g || !c->width) return -2;
    *g = c->geom;
    return 0;
}

// pinned staging of a slot (same offsets as the device block), created on first use: device-only slots cost no host memory
static int map_staging(vp8hip_ctx *c, Slot &s)
{
    if (c->pool)
        return fail(c, -2, "a context with a block pool (vp8hip_configure_pooled) takes its IR from the device's entropy decoder only");
    if (s.h_block) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc((void **)&s.h_block, c->slot_bytes, hipHostMallocDefault));
    memset(s.h_block, 0, c->o_blocks);
    s.h_hdr = (vp8ir_frame_hdr *)s.h_block; s.h_mbx = (vp8ir_mbx *)(s.h_block + c->o_mbx);
    s.h_blocks = (int16_t *)(s.h_block + c->o_blocks); s.h_mvs = (vp8ir_mv *)(s.h_block + c->o_mvs);
    return 0;
}

extern "C" int vp8hip_ir_map_compact(vp8hip_ctx *c, int slot, vp8ir_frame_hdr **hdr, vp8ir_mbx **mbx, int16_t **blocks, size_t *cap_blocks,
                                     vp8ir_mv **mvs)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_map_compact: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (map_staging(c, s)) return -1;
    if (hdr) *hdr = s.h_hdr;
    if (mbx) *mbx = s.h_mbx;
    if (blocks) *blocks = s.h_blocks;
    if (cap_blocks) *cap_blocks = c->cap_blocks;
    if (mvs) *mvs = s.h_mvs;
    return 0;
}

extern "C" int vp8hip_ir_upload_compact(vp8hip_ctx *c, int slot, size_t nblocks)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_upload_compact: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.h_block || c->pool) return fail(c, -2, "vp8hip_ir_upload_compact: slot %d was never mapped", slot);
    const vp8ir_frame_hdr &h = *s.h_hdr;
    if (h.mb_cols != c->dg.mb_cols || h.mb_rows != c->dg.mb_rows)
        return fail(c, -2, "vp8hip_ir_upload_compact: header is %dx%d MBs, context configured for %dx%d", h.mb_cols, h.mb_rows,
                    c->dg.mb_cols, c->dg.mb_rows);
    if (nblocks > c->cap_blocks) return fail(c, -2, "vp8hip_ir_upload_compact: %zu blocks for %d macroblocks", nblocks, c->nmb);
    HIPCHK(c, hipSetDevice(c->device));
    s.hdr_copy = h;
    s.nblocks = nblocks;
    // ONE copy: the records and the blocks behind them, as the feeder left them; nothing on the device touches them before the
    // pixel kernels do
    HIPCHK(c, hipMemcpyAsync(s.d_mbx, s.h_mbx, (size_t)c->nmb * sizeof(vp8ir_mbx) + nblocks * 32, hipMemcpyHostToDevice, c->stream));
    if (h.frame_type != 0)
        HIPCHK(c, hipMemcpyAsync(s.d_mvs, s.h_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyHostToDevice, c->stream));
    return 0;
}

extern "C" int vp8hip_ir_map(vp8hip_ctx *c, int slot, vp8ir_frame_hdr **hdr, vp8ir_mb **mbs, int16_t **coef, vp8ir_mv **mvs)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_map: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (map_staging(c, s)) return -1;
    if (!s.h_dense) {
        const size_t mb_bytes = (size_t)c->nmb * sizeof(vp8ir_mb);
        s.h_dense = (char *)calloc(1, mb_bytes + (size_t)c->nmb * VP8IR_COEF_PER_MB * sizeof(int16_t));
        if (!s.h_dense) return fail(c, -1, "vp8hip_ir_map: out of host memory");
        s.h_mbs = (vp8ir_mb *)s.h_dense; s.h_coef = (int16_t *)(s.h_dense + mb_bytes);
    }
    if (hdr) *hdr = s.h_hdr;
    if (mbs) *mbs = s.h_mbs;
    if (coef) *coef = s.h_coef;
    if (mvs) *mvs = s.h_mvs;
    return 0;
}

// The dense view -> the device form, on the host (vp8ir_compact_mb), then the compact upload.  Tests, the oracle's IR, anything
// that speaks dense arrays; a feeder that cares about its time writes the device form itself (vp8_parser_decode_mbs_compact).
extern "C" int vp8hip_ir_upload(vp8hip_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_upload: bad slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.h_dense) return fail(c, -2, "vp8hip_ir_upload: slot %d was never mapped", slot);
    // the staging is read by the copy of the upload before this one
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t nb = 0;
    for (int i = 0; i < c->nmb; i++)
        nb += vp8ir_compact_mb(&s.h_mbs[i], s.h_coef + (size_t)i * VP8IR_COEF_PER_MB, (uint32_t)nb, &s.h_mbx[i], s.h_blocks);
    return vp8hip_ir_upload_compact(c, slot, nb);
}

extern "C" int vp8hip_ir_copy(vp8hip_ctx *c, int dst, int src)
{
    if (!c || dst < 0 || src < 0 || dst >= (int)c->slots.size() || src >= (int)c->slots.size())
        return fail(c, -2, "vp8hip_ir_copy: bad slots %d <- %d", dst, src);
    if (dst == src) return 0;
    Slot &d = c->slots[dst], &s = c->slots[src];
    HIPCHK(c, hipSetDevice(c->device));
    d.hdr_copy = s.hdr_copy;
    d.nblocks = s.nblocks;
    // (with a block pool the records say where in the pool the blocks are: the copy shares them)
    const size_t nb = c->pool ? 0 : s.nblocks == NBLOCKS_UNKNOWN ? c->cap_blocks : s.nblocks;
    HIPCHK(c, hipMemcpyAsync(d.d_mbx, s.d_mbx, (size_t)c->nmb * sizeof(vp8ir_mbx) + nb * 32, hipMemcpyDeviceToDevice, c->stream));
    if (s.hdr_copy.frame_type != 0)
        HIPCHK(c, hipMemcpyAsync(d.d_mvs, s.d_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyDeviceToDevice,
                                 c->stream));
    return 0;
}

int vp8hip_check_status(vp8hip_ctx *c)
{
    if (c->h_status && *c->h_status) {
        const int st = *c->h_status;
        *c->h_status = 0;
        return fail(c, -1, "a row hand-over between CUs did not arrive (%s kernel): the frames of that launch are invalid",
                    st == 1 ? "reconstruction" : "loop filter");
    }
    return 0;
}
#define check_status vp8hip_check_status

extern "C" int vp8hip_sync(vp8hip_ctx *c)
{
    if (!c) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_status(c);
}

extern "C" int vp8hip_get_stats_at(vp8hip_ctx *c, int back, vp8hip_stats *st)
{
    if (!c || !st || back < 0 || back >= VP8HIP_STATS_RING) return -2;
    if (back >= c->ncalls) { memset(st, 0, sizeof *st); return c->ncalls ? fail(c, -2, "vp8hip_get_stats_at: only %ld launches so far", c->ncalls) : 0; }
    const int r = (int)((c->ncalls - 1 - back) % VP8HIP_STATS_RING);
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
    if (vp8hip_raster_pool(c) || vp8hip_need_raster(c, fb, 1)) return -1;
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


extern "C" int vp8hip_ir_fetch_mvs(vp8hip_ctx *c, int slot, vp8ir_mv *mvs)
{
    if (!c || !mvs || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_fetch_mvs: bad slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(mvs, c->slots[slot].d_mvs, (size_t)c->nmb * 16 * sizeof(vp8ir_mv), hipMemcpyDeviceToHost));
    return 0;
}

// The slot as it stands on the device, expanded to the dense view on the host (tests, debugging)
extern "C" int vp8hip_ir_fetch(vp8hip_ctx *c, int slot, vp8ir_mb *mbs, int16_t *coef)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, -2, "vp8hip_ir_fetch: bad slot %d", slot);
    Slot &s = c->slots[slot];
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->pool) {
        // the records, then row by row the piece of the pool the row's blocks are in (a row's blocks are together)
        const int cols = c->dg.mb_cols;
        vp8ir_mbx *x = (vp8ir_mbx *)malloc((size_t)c->nmb * sizeof(vp8ir_mbx));
        int16_t *rowb = (int16_t *)malloc((size_t)cols * VP8IR_MAX_BLOCKS_PER_MB * 32);
        int16_t *one = (int16_t *)malloc(VP8IR_COEF_PER_MB * sizeof(int16_t));
        int rc = 0;
        if (!x || !rowb || !one) rc = fail(c, -1, "vp8hip_ir_fetch: out of host memory");
        if (!rc && hipMemcpy(x, s.d_mbx, (size_t)c->nmb * sizeof(vp8ir_mbx), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, -1, "vp8hip_ir_fetch: copy failed");
        const size_t pool_blocks = ((size_t)c->pool_chunks + 1) * c->chunk_blocks;
        for (int r = 0; r < c->dg.mb_rows && !rc; r++) {
            const uint32_t first = x[(size_t)r * cols].d.sparse_first;
            size_t nrow = 0;
            for (int i = r * cols; i < (r + 1) * cols; i++)
                for (int k = 0; k < 24; k++) nrow += vp8ir_block_kind(&x[i].d, k) == 2;
            if (first + nrow > pool_blocks) { rc = fail(c, -1, "vp8hip_ir_fetch: row %d of slot %d points outside the block pool", r, slot); break; }
            if (nrow && hipMemcpy(rowb, c->pool + (size_t)first * 32, nrow * 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(c, -1, "vp8hip_ir_fetch: copy failed"); break; }
            for (int i = r * cols; i < (r + 1) * cols; i++) {
                vp8ir_mbx m = x[i];
                if (m.d.sparse_first < first || m.d.sparse_first - first > nrow) { rc = fail(c, -1, "vp8hip_ir_fetch: macroblock %d of slot %d is not with its row", i, slot); break; }
                m.d.sparse_first -= first;
                vp8ir_expand_mb(&m, rowb, mbs ? &mbs[i] : nullptr, coef ? coef + (size_t)i * VP8IR_COEF_PER_MB : one);
            }
        }
        free(x); free(rowb); free(one);
        return rc;
    }
    const size_t nb = s.nblocks == NBLOCKS_UNKNOWN ? c->cap_blocks : s.nblocks;
    const size_t bytes = (size_t)c->nmb * sizeof(vp8ir_mbx) + nb * 32;
    char *tmp = (char *)malloc(bytes);
    int16_t *one = (int16_t *)malloc(VP8IR_COEF_PER_MB * sizeof(int16_t));
    if (!tmp || !one) { free(tmp); free(one); return fail(c, -1, "vp8hip_ir_fetch: out of host memory"); }
    const hipError_t e = hipMemcpy(tmp, s.d_mbx, bytes, hipMemcpyDeviceToHost);
    int rc = 0;
    if (e != hipSuccess) rc = fail(c, -1, "vp8hip_ir_fetch: %s", hipGetErrorString(e));
    const vp8ir_mbx *x = (const vp8ir_mbx *)tmp;
    const int16_t *blocks = (const int16_t *)(tmp + (size_t)c->nmb * sizeof(vp8ir_mbx));
    for (int i = 0; i < c->nmb && !rc; i++) {
        unsigned n2 = 0;
        for (int k = 0; k < 24; k++) n2 += vp8ir_block_kind(&x[i].d, k) == 2;
        if ((size_t)x[i].d.sparse_first + n2 > nb)
            rc = fail(c, -1, "vp8hip_ir_fetch: macroblock %d of slot %d points outside the block stream", i, slot);
        else
            vp8ir_expand_mb(&x[i], blocks, mbs ? &mbs[i] : nullptr, coef ? coef + (size_t)i * VP8IR_COEF_PER_MB : one);
    }
    free(tmp); free(one);
    return rc;
}

extern "C" size_t vp8hip_frame_stride(const vp8hip_ctx *c) { return c ? c->fb_stride : 0; }

extern "C" size_t vp8hip_i420_bytes(const vp8hip_ctx *c)
{
    return c && c->width ? (size_t)c->width * c->height + 2 * (size_t)((c->width + 1) / 2) * ((c->height + 1) / 2) : 0;
}

static int fetch_impl(vp8hip_ctx *c, int first_fb, int count, uint8_t *dst, uint8_t *digests, bool packed);
extern "C" int vp8hip_frames_fetch_async(vp8hip_ctx *c, int first_fb, int count, uint8_t *dst, uint8_t *digests)
{
    return fetch_impl(c, first_fb, count, dst, digests, false);
}
extern "C" int vp8hip_frames_fetch_i420_async(vp8hip_ctx *c, int first_fb, int count, uint8_t *dst, uint8_t *digests)
{
    if (c && (c->width & 7))
        return fail(c, -3, "vp8hip_frames_fetch_i420_async: packed frames need a display width that is a multiple of 8 (%d)", c->width);
    return fetch_impl(c, first_fb, count, dst, digests, dst != nullptr);
}

static int fetch_impl(vp8hip_ctx *c, int first_fb, int count, uint8_t *dst, uint8_t *digests, bool packed)
{
    if (!c || first_fb < 0 || count < 1 || first_fb + count > (int)c->fb.size() || (!dst && !digests))
        return fail(c, -2, "vp8hip_frames_fetch_async: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    // Frames that only exist as tiles (a large launch wrote them, nothing has asked for their raster form) are read as tiles: the
    // MD5 kernel walks the tiles, and -- direct downloads switched on (vp8hip_set_direct_download) -- the frames leave through a
    // kernel that writes raster rows straight into the caller's page-locked memory: the tiled -> raster pass happens where the
    // frame leaves the device and costs the HBM one read (44 GB/s over PCIe on an otherwise idle device, against the copy engines'
    // 25-33).  The default is the plain way -- raster form in HBM first (vp8hip_need_raster), then the copy engines -- because a
    // pass that waits on PCIe occupies wave slots beside whatever else runs: with the entropy decoder on the device the pipeline
    // of bin/batch_md5 does 9.6 k 1080p frames/s the plain way and 6.5-8.5 k with direct downloads (gpurun_out/r4e).
    const bool whole_blocks = (c->width & 127) == 0;        // a row is whole MD5 blocks (vp8_md5.hip); other widths: raster form, vp8_md5_any_kernel
    bool tiled = c->tile_block != nullptr && (whole_blocks || !digests);
    for (int i = 0; i < count && tiled; i++) tiled = c->fb_state[(size_t)(first_fb + i)] == FB_TILES;
    if (tiled && dst && !packed && !c->knobs.direct_download) tiled = false;
    if (tiled && dst && !packed) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) != hipSuccess) { (void)hipGetLastError(); tiled = false; }
        else tiled = at.type == hipMemoryTypeHost;
    }
    if (!tiled && (vp8hip_raster_pool(c) || vp8hip_need_raster(c, first_fb, count))) return -1;
    if (!c->stream_d2h) {
        // (a stream of another priority class than the context's: the runtime then gives it a hardware queue of its own, and what
        // it carries -- a pass that waits on PCIe for a third of a second per 4096 frames -- runs BESIDE the main stream's kernels)
        int prio_least = 0, prio_greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&c->stream_d2h, hipStreamNonBlocking, c->knobs.d2h_prio ? prio_least : 0));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_from, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_done, hipEventDisableTiming));
    }
    if (c->d2h_count) HIPCHK(c, hipEventSynchronize(c->ev_d2h_done));  // one copy in flight at a time
    HIPCHK(c, hipEventRecord(c->ev_d2h_from, c->stream));             // everything queued so far: the frames' kernels
    HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_d2h_from, 0));
    if (dst && packed) {
        // packed I420: a pass from whichever form the frames are in into a staging buffer on the device, then the copy engines, in
        // pieces on streams of their own -- a tenth less over the link than whole frame buffers
        const size_t fbytes = vp8hip_i420_bytes(c);
        if (fbytes * (size_t)count > c->i420_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
            if (c->d_i420) (void)hipFree(c->d_i420);
            c->d_i420 = nullptr; c->i420_cap = 0;
            if (hipMalloc((void **)&c->d_i420, fbytes * (size_t)count + 256) != hipSuccess) {
                (void)hipGetLastError();
                return fail(c, -1, "no device memory for %d packed frames (%zu MB)", count, fbytes * (size_t)count >> 20);
            }
            c->i420_cap = fbytes * (size_t)count;
        }
        long units = (long)count * (tiled ? c->dg.mb_rows : c->height + 2 * ((c->height + 1) / 2));
        if (units > 16L * c->num_cu) units = 16L * c->num_cu;
        if (tiled)
            hipLaunchKernelGGL(vp8_pack_i420_tiles_kernel, dim3((unsigned)units), dim3(256), 0, c->stream_d2h, (const uint8_t *)c->fb_tiles[(size_t)first_fb],
                               c->tile_frame, c->d_i420, fbytes, count, c->dg, c->width, c->height);
        else
            hipLaunchKernelGGL(vp8_pack_i420_raster_kernel, dim3((unsigned)units), dim3(256), 0, c->stream_d2h, (const uint8_t *)c->fb_block, c->fb_stride,
                               first_fb, c->d_i420, fbytes, count, c->dg, c->width, c->height);
        HIPCHK(c, hipGetLastError());
        const int pieces = count >= 64 ? c->knobs.d2h_streams : 1;
        const int per = (count + pieces - 1) / pieces;
        // (the packed copy is there: the copies' other streams, and the hash, wait for this)
        if (!c->ev_pack) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_pack, c->stream_d2h));
        for (int k = 1; k < pieces; k++) {
            if (!c->stream_d2h_more[k - 1]) {
                HIPCHK(c, hipStreamCreateWithFlags(&c->stream_d2h_more[k - 1], hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_more[k - 1], hipEventDisableTiming));
            }
            const int at = k * per, n = count - at < per ? count - at : per;
            if (n < 1) break;
            HIPCHK(c, hipStreamWaitEvent(c->stream_d2h_more[k - 1], c->ev_pack, 0));
            HIPCHK(c, hipMemcpyAsync(dst + fbytes * (size_t)at, c->d_i420 + fbytes * (size_t)at, fbytes * (size_t)n, hipMemcpyDeviceToHost, c->stream_d2h_more[k - 1]));
            HIPCHK(c, hipEventRecord(c->ev_d2h_more[k - 1], c->stream_d2h_more[k - 1]));
        }
        HIPCHK(c, hipMemcpyAsync(dst, c->d_i420, fbytes * (size_t)(per < count ? per : count), hipMemcpyDeviceToHost, c->stream_d2h));
        for (int k = 1; k < pieces && k * per < count; k++) HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_d2h_more[k - 1], 0));
    } else if (dst) {
        if (tiled) {
            // (the pass is bound by PCIe, not by the device: a workgroup per CU keeps the link full and leaves the SIMDs to the
            // launches that run beside it)
            long units = (long)c->dg.mb_rows * count;
            const long cap = c->knobs.download_blocks > 0 ? c->knobs.download_blocks : c->num_cu;
            if (units > cap) units = cap;
            hipLaunchKernelGGL(vp8_detile_run_kernel, dim3((unsigned)units), dim3(256), 0, c->stream_d2h, (const uint8_t *)c->fb_tiles[(size_t)first_fb],
                               c->tile_frame, dst, c->fb_stride, count, c->dg);
            HIPCHK(c, hipGetLastError());
        } else {
            // (a copy is one copy engine's work: 25-33 GB/s; in pieces on streams of their own the link's other engines take part)
            const int pieces = count >= 64 ? c->knobs.d2h_streams : 1;
            const int per = (count + pieces - 1) / pieces;
            for (int k = 1; k < pieces; k++) {
                if (!c->stream_d2h_more[k - 1]) {
                    HIPCHK(c, hipStreamCreateWithFlags(&c->stream_d2h_more[k - 1], hipStreamNonBlocking));
                    HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_more[k - 1], hipEventDisableTiming));
                }
                const int at = k * per, n = count - at < per ? count - at : per;
                if (n < 1) break;
                HIPCHK(c, hipStreamWaitEvent(c->stream_d2h_more[k - 1], c->ev_d2h_from, 0));
                HIPCHK(c, hipMemcpyAsync(dst + c->fb_stride * (size_t)at, c->fb[first_fb + at], c->fb_stride * (size_t)n, hipMemcpyDeviceToHost, c->stream_d2h_more[k - 1]));
                HIPCHK(c, hipEventRecord(c->ev_d2h_more[k - 1], c->stream_d2h_more[k - 1]));
            }
            HIPCHK(c, hipMemcpyAsync(dst, c->fb[first_fb], c->fb_stride * (size_t)(per < count ? per : count), hipMemcpyDeviceToHost, c->stream_d2h));
            for (int k = 1; k < pieces && k * per < count; k++) HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_d2h_more[k - 1], 0));
        }
    }
    if (digests) {
        if (c->md5_cap < count) {
            HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
            if (c->d_md5) (void)hipFree(c->d_md5);
            c->d_md5 = nullptr; c->md5_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_md5, 16 * (size_t)(count < 64 ? 64 : count)));
            c->md5_cap = count < 64 ? 64 : count;
        }
        // a frame per lane: the frames' hashes run side by side -- and, when the frames themselves are asked for as well, BESIDE their
        // copy, on a stream of their own: a hash is 70-90 ms whatever the batch, a fifth of what 4096 1080p frames take over the link
        hipStream_t ms = c->stream_d2h;
        if (dst) {
            if (!c->stream_d2h_more[2]) {
                HIPCHK(c, hipStreamCreateWithFlags(&c->stream_d2h_more[2], hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_more[2], hipEventDisableTiming));
            }
            ms = c->stream_d2h_more[2];
            HIPCHK(c, hipStreamWaitEvent(ms, c->ev_d2h_from, 0));
        }
        // Batches beyond what the Infinity Cache holds of tiles (vp8_md5.hip: a 16-byte row piece costs its 128-byte line, eight times
        // over -- 254 MB of lines in use at 16,384 1080p frames) are hashed from a PACKED copy: vp8_pack_i420_tiles_kernel reads every
        // line once, and the hash then streams its frame front to back (the raster reader over a geometry without borders).  Where the
        // frames were packed for the download anyway the copy is there; else it is made if the device has the room (3.1 MB a frame).
        bool from_packed = tiled && whole_blocks && dst && packed;
        if (tiled && whole_blocks && !dst && count >= c->knobs.md5_pack_from) {
            const size_t fbytes = vp8hip_i420_bytes(c);
            if (fbytes * (size_t)count > c->i420_cap) {
                HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
                if (c->d_i420) (void)hipFree(c->d_i420);
                c->d_i420 = nullptr; c->i420_cap = 0;
                if (hipMalloc((void **)&c->d_i420, fbytes * (size_t)count + 256) == hipSuccess) c->i420_cap = fbytes * (size_t)count;
                else { (void)hipGetLastError(); c->d_i420 = nullptr; }
            }
            if (c->d_i420 && fbytes * (size_t)count <= c->i420_cap) {
                long units = (long)count * c->dg.mb_rows;
                if (units > 16L * c->num_cu) units = 16L * c->num_cu;
                hipLaunchKernelGGL(vp8_pack_i420_tiles_kernel, dim3((unsigned)units), dim3(256), 0, ms, (const uint8_t *)c->fb_tiles[(size_t)first_fb],
                                   c->tile_frame, c->d_i420, fbytes, count, c->dg, c->width, c->height);
                HIPCHK(c, hipGetLastError());
                from_packed = true;
            }
        }
        if (from_packed) {
            // (the hash kernel of a packed batch download runs on a stream of its own: behind the pack pass)
            if (dst) HIPCHK(c, hipStreamWaitEvent(ms, c->ev_pack, 0));
            DevGeom pg = c->dg;
            const int cw = c->width / 2, ch = (c->height + 1) / 2;
            pg.y_off = 0; pg.y_stride = c->width; pg.uv_stride = cw;
            pg.u_off = c->width * c->height; pg.v_off = pg.u_off + cw * ch;
            hipLaunchKernelGGL(vp8_md5_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ms, (const uint8_t *)c->d_i420, vp8hip_i420_bytes(c),
                               (const int *)nullptr, 0, count, pg, c->width, c->height, c->d_md5);
        } else if (tiled)
            hipLaunchKernelGGL(vp8_md5_tiles_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ms,
                               (const uint8_t *)c->tile_block, c->tile_frame, (const int *)nullptr, first_fb, count, c->dg, c->width, c->height, c->d_md5);
        else
            hipLaunchKernelGGL(whole_blocks ? vp8_md5_kernel : vp8_md5_any_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ms,
                               (const uint8_t *)c->fb_block, c->fb_stride, (const int *)nullptr, first_fb, count, c->dg, c->width, c->height, c->d_md5);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(digests, c->d_md5, 16 * (size_t)count, hipMemcpyDeviceToHost, ms));
        if (dst) {
            HIPCHK(c, hipEventRecord(c->ev_d2h_more[2], ms));
            HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_d2h_more[2], 0));
        }
    }
    HIPCHK(c, hipEventRecord(c->ev_d2h_done, c->stream_d2h));
    c->d2h_first = first_fb; c->d2h_count = count; c->d2h_listed = false;
    return 0;
}

extern "C" int vp8hip_set_pred_tiles(vp8hip_ctx *c, int mode)
{
    if (!c || mode < 0 || mode > 2) return fail(c, -2, "vp8hip_set_pred_tiles: bad arguments");
    c->knobs.pred_tiles = mode;
    return 0;
}
extern "C" int vp8hip_set_direct_download(vp8hip_ctx *c, int on)
{
    if (!c) return -2;
    c->knobs.direct_download = on != 0;
    return 0;
}

extern "C" int vp8hip_frames_to_raster(vp8hip_ctx *c, int first_fb, int count)
{
    if (!c || first_fb < 0 || count < 1 || first_fb + count > (int)c->fb.size()) return fail(c, -2, "vp8hip_frames_to_raster: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    return vp8hip_need_raster(c, first_fb, count);
}

// The digests of any n frame buffers (fbs[i]: not necessarily neighbours -- the shown frames of many streams decoded side by
// side).  Same stream, same wait (vp8hip_download_wait) as vp8hip_frames_fetch_async; `fbs` may be reused when the call returns.
extern "C" int vp8hip_frames_md5_list_async(vp8hip_ctx *c, const int *fbs, int n, uint8_t *digests)
{
    if (!c || !fbs || n < 1 || !digests) return fail(c, -2, "vp8hip_frames_md5_list_async: bad arguments");
    for (int i = 0; i < n; i++)
        if (fbs[i] < 0 || fbs[i] >= (int)c->fb.size()) return fail(c, -2, "vp8hip_frames_md5_list_async: frame buffer %d out of range", fbs[i]);
    HIPCHK(c, hipSetDevice(c->device));
    const bool whole_blocks = (c->width & 127) == 0;
    bool tiled = c->tile_block != nullptr && whole_blocks;
    for (int i = 0; i < n && tiled; i++) tiled = (c->fb_state[(size_t)fbs[i]] & FB_TILES) != 0;
    if (!tiled && (vp8hip_raster_pool(c) || vp8hip_need_raster_list(c, fbs, n))) return -1;
    if (!c->stream_d2h) {
        int prio_least = 0, prio_greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&c->stream_d2h, hipStreamNonBlocking, c->knobs.d2h_prio ? prio_least : 0));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_from, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_d2h_done, hipEventDisableTiming));
    }
    if (c->d2h_count) HIPCHK(c, hipEventSynchronize(c->ev_d2h_done));  // one fetch in flight at a time
    if (c->md5_cap < n || c->md5_idx_cap < n) {
        HIPCHK(c, hipStreamSynchronize(c->stream_d2h));
        const int cap = n < 64 ? 64 : n;
        if (c->md5_cap < n) {
            if (c->d_md5) (void)hipFree(c->d_md5);
            c->d_md5 = nullptr; c->md5_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_md5, 16 * (size_t)cap));
            c->md5_cap = cap;
        }
        if (c->md5_idx_cap < n) {
            if (c->d_md5_idx) (void)hipFree(c->d_md5_idx);
            if (c->h_md5_idx) (void)hipHostFree(c->h_md5_idx);
            c->d_md5_idx = nullptr; c->h_md5_idx = nullptr; c->md5_idx_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_md5_idx, sizeof(int) * (size_t)cap));
            HIPCHK(c, hipHostMalloc((void **)&c->h_md5_idx, sizeof(int) * (size_t)cap, hipHostMallocDefault));
            c->md5_idx_cap = cap;
        }
    }
    memcpy(c->h_md5_idx, fbs, sizeof(int) * (size_t)n);      // (the previous fetch, which read this staging, has been waited for)
    HIPCHK(c, hipEventRecord(c->ev_d2h_from, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_d2h_from, 0));
    HIPCHK(c, hipMemcpyAsync(c->d_md5_idx, c->h_md5_idx, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream_d2h));
    if (tiled)
        hipLaunchKernelGGL(vp8_md5_tiles_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream_d2h, (const uint8_t *)c->tile_block,
                           c->tile_frame, (const int *)c->d_md5_idx, 0, n, c->dg, c->width, c->height, c->d_md5);
    else
        hipLaunchKernelGGL(whole_blocks ? vp8_md5_kernel : vp8_md5_any_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream_d2h,
                           (const uint8_t *)c->fb_block, c->fb_stride, (const int *)c->d_md5_idx, 0, n, c->dg, c->width, c->height, c->d_md5);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(digests, c->d_md5, 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream_d2h));
    HIPCHK(c, hipEventRecord(c->ev_d2h_done, c->stream_d2h));
    // (a launch that writes one of the listed frame buffers waits for this fetch)
    c->d2h_first = 0; c->d2h_count = (int)c->fb.size();
    c->d2h_mask.assign(c->fb.size(), (uint8_t)0);
    for (int i = 0; i < n; i++) c->d2h_mask[(size_t)fbs[i]] = 1;
    c->d2h_listed = true;
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
    if (vp8hip_raster_pool(c)) return -1;
    HIPCHK(c, hipMemcpyAsync(c->fb[fb], buf, (size_t)c->geom.frame_size, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fb_state[(size_t)fb] = FB_RASTER;
    return 0;
}

extern "C" int vp8hip_frame_copy(vp8hip_ctx *c, int dst, int src)
{
    if (!c || dst < 0 || src < 0 || dst >= (int)c->fb.size() || src >= (int)c->fb.size())
        return fail(c, -2, "vp8hip_frame_copy: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (dst == src) return 0;
    // in whatever form(s) the source holds the frame
    const uint8_t st = c->fb_state[(size_t)src];
    if ((st & FB_RASTER) && !c->fb_block) {                 // (never written: the raster form is zeros, and so will the copy's be)
        if (!(st & FB_TILES) && vp8hip_raster_pool(c)) return -1;
    }
    if ((st & FB_RASTER) && c->fb_block) HIPCHK(c, hipMemcpyAsync(c->fb[dst], c->fb[src], (size_t)c->geom.frame_size, hipMemcpyDeviceToDevice, c->stream));
    if (st & FB_TILES) HIPCHK(c, hipMemcpyAsync(c->fb_tiles[(size_t)dst], c->fb_tiles[(size_t)src], c->tile_frame, hipMemcpyDeviceToDevice, c->stream));
    c->fb_state[(size_t)dst] = st;
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
