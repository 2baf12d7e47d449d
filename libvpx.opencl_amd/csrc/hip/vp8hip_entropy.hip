// vp8hip_entropy_decode (include/vp8hip.h): the launch of the device's entropy decoder (vp8_entropy.hip), which writes the
// frames' IR slots in the device form of include/vp8_ir.h -- the form the pixel kernels read, nothing in between.
#include "vp8hip_ctx.hip.h"

extern "C" __global__ void vp8_entropy_kernel(const vp8hip_entropy_frame *frames, int count, int lpw, const uint8_t *data, DevGeom g,
                                              size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbx, size_t o_blocks, size_t o_mvs,
                                              int first_slot, unsigned int *scratch, unsigned int *status, char *pool, unsigned int *pool_ctr,
                                              unsigned int pool_chunks, unsigned int chunk_blocks);
extern "C" size_t vp8_entropy_lds_bytes(int lpw);
extern "C" __global__ void vp8_entropy_parts_kernel(const vp8hip_entropy_frame *frames, int count, int np, const uint8_t *data, DevGeom g,
                                                    size_t data_bytes, char *slot_base, size_t slot_bytes, size_t o_mbx, size_t o_blocks,
                                                    int first_slot, unsigned int *scratch, unsigned int *status);

// A launch in two steps: its input -- checked, and on its way to the device on the copy stream -- and the kernel.
// vp8hip_entropy_decode is both; vp8hip_entropy_stage + vp8hip_entropy_decode(..., NULL, NULL, 0) lets a caller send the input of
// the NEXT launch while it still has things to queue and wait for on behalf of the current one.
static int entropy_stage(vp8hip_ctx *c, int count, const vp8hip_entropy_frame *frames, const uint8_t *data, size_t data_bytes);
static int entropy_launch(vp8hip_ctx *c, int first_slot, int count, int set, int np, size_t data_bytes);

extern "C" int vp8hip_entropy_stage(vp8hip_ctx *c, int count, const vp8hip_entropy_frame *frames, const uint8_t *data, size_t data_bytes)
{
    if (!c || !frames || !data || count < 1) return fail(c, -2, "vp8hip_entropy_stage: bad arguments");
    if (c->ent_staged.count) return fail(c, -2, "vp8hip_entropy_stage: the input staged before has not been launched");
    return entropy_stage(c, count, frames, data, data_bytes);
}

extern "C" int vp8hip_entropy_decode(vp8hip_ctx *c, int first_slot, int count, const vp8hip_entropy_frame *frames, const uint8_t *data,
                                     size_t data_bytes)
{
    if (!c || count < 1 || first_slot < 0 || first_slot + count > (int)c->slots.size() || (!frames) != (!data))
        return fail(c, -2, "vp8hip_entropy_decode: bad arguments");
    if (frames) {
        if (c->ent_staged.count) return fail(c, -2, "vp8hip_entropy_decode: input was staged (vp8hip_entropy_stage): launch that first");
        if (entropy_stage(c, count, frames, data, data_bytes)) return -1;
    } else if (c->ent_staged.count != count)
        return fail(c, -2, "vp8hip_entropy_decode: %d frames asked for, %d staged", count, c->ent_staged.count);
    const int set = c->ent_staged.set, np = c->ent_staged.np;
    frames = c->ent_staged.frames; data_bytes = c->ent_staged.data_bytes;
    c->ent_staged.count = 0;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t swords = np > 1 ? (size_t)count * ((size_t)c->dg.mb_cols + 3 * (size_t)c->nmb) : (size_t)count * (8 * (size_t)c->dg.mb_cols + 64);
    if ((size_t)count > c->ent_status_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_ent_status) (void)hipFree(c->d_ent_status);
        c->d_ent_status = nullptr; c->ent_status_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_status, (size_t)count * 4));
        c->ent_status_cap = (size_t)count;
    }
    if (swords > c->ent_scratch_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_ent_scratch) (void)hipFree(c->d_ent_scratch);
        c->d_ent_scratch = nullptr; c->ent_scratch_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_scratch, swords * 4));
        c->ent_scratch_cap = swords;
    }
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ent_in[set], 0));
    for (int i = 0; i < count; i++) {
        Slot &s = c->slots[first_slot + i];
        s.hdr_copy = frames[i].hdr;
        s.nblocks = NBLOCKS_UNKNOWN;           // (the host never sees how many blocks the device wrote)
    }
    return entropy_launch(c, first_slot, count, set, np, data_bytes);
}

static int entropy_stage(vp8hip_ctx *c, int count, const vp8hip_entropy_frame *frames, const uint8_t *data, size_t data_bytes)
{
    bool any_inter = false;
    for (int i = 0; i < count; i++) {
        const vp8hip_entropy_frame &f = frames[i];
        const vp8ir_frame_hdr &h = f.hdr;
        if (h.frame_type != 0) any_inter = true;
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
    if (!c->ent_tables_loaded) {
        c->ent_tables_loaded = true;
        HIPCHK(c, hipFuncSetAttribute((const void *)vp8_entropy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vp8_entropy_lds_bytes(64)));
        const char *e = getenv("VP8HIP_ENTROPY_LANES");     // lanes of a wave that carry a frame (a tuning knob: read once)
        c->ent_lpw = e ? atoi(e) : 0;
        const char *e2 = getenv("VP8HIP_ENTROPY_PARTS");   // 0: a frame per lane whatever the number of token partitions
        c->ent_parts_off = e2 && atoi(e2) == 0;
        if (c->ent_lpw < 1 || c->ent_lpw > 64) c->ent_lpw = 0;
    }
    if (c->dg.mb_cols < np || c->dg.mb_cols > 256 || c->dg.mb_cols * (64 / np) > 4096 || c->ent_parts_off || any_inter) np = 1;
    if (c->pool) np = 1;           // (the partition-per-lane kernel gives every partition a worst-case region of the slot's own stream)
    if (!c->stream_h2d) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_h2d, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_ent_in[k], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_ent_out[k], hipEventDisableTiming));
        }
    }
    const int set = c->ent_set;
    c->ent_set ^= 1;
    if (fbytes > c->ent_frames_cap2[set]) {
        HIPCHK(c, hipEventSynchronize(c->ev_ent_out[set]));          // (the launch that read this set)
        if (c->d_ent_frames2[set]) (void)hipFree(c->d_ent_frames2[set]);
        c->d_ent_frames2[set] = nullptr; c->ent_frames_cap2[set] = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_frames2[set], fbytes));
        c->ent_frames_cap2[set] = fbytes;
    }
    if (data_bytes + 16 > c->ent_data_cap2[set]) {
        HIPCHK(c, hipEventSynchronize(c->ev_ent_out[set]));
        if (c->d_ent_data2[set]) (void)hipFree(c->d_ent_data2[set]);
        c->d_ent_data2[set] = nullptr; c->ent_data_cap2[set] = 0;
        const size_t cap = data_bytes + data_bytes / 4 + 4096;
        HIPCHK(c, hipMalloc((void **)&c->d_ent_data2[set], cap));
        c->ent_data_cap2[set] = cap;
    }
    // the launch's input: on the copy stream, as soon as the kernel that last read this set is done -- beside whatever the context's
    // stream still has to do before this launch
    hipStream_t cs = c->stream_h2d;        // (24,576 1080p frames per launch: 15 ms of every 380 against copies on the context's stream; the kernel beside a copy is 8 % slower)
    HIPCHK(c, hipStreamWaitEvent(cs, c->ev_ent_out[set], 0));
    HIPCHK(c, hipMemcpyAsync(c->d_ent_frames2[set], frames, fbytes, hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipMemcpyAsync(c->d_ent_data2[set], data, data_bytes, hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipEventRecord(c->ev_ent_in[set], cs));
    c->ent_staged.count = count; c->ent_staged.set = set; c->ent_staged.np = np; c->ent_staged.frames = frames; c->ent_staged.data_bytes = data_bytes;
    return 0;
}

static int entropy_launch(vp8hip_ctx *c, int first_slot, int count, int set, int np, size_t data_bytes)
{
    // Lanes per wave.  A launch lasts as long as its largest frame IF all its waves are on the device at once, and what bounds
    // that is LDS -- 1.4 KB a lane (the frame's coefficient probabilities: 1152 bytes), so a CU holds one wave of 64 lanes, three
    // of 32, seven of 16: 16,384 / 24,576 / 28,672 frames on the 256 CUs; a launch with more waves than fit runs in rounds, each
    // as long as ITS largest frame (24,576 1080p frames: 1.14 s with 32 lanes a wave, 1.97 s with 64; 640x360: the same steps,
    // tools/entropy_scale.sh).  Where they fit, more lanes to a wave are a little faster (8192 frames: 1.05 s at 64, 1.13 at 32,
    // 1.45 at 16).  So: the most lanes per wave with which the launch is one round.
    int lpw = c->ent_lpw;
    if (!lpw) {
        if (!c->ent_resident[0])
            for (int k = 0; k < 3; k++) {
                int nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, vp8_entropy_kernel, 64, vp8_entropy_lds_bytes(64 >> k)) != hipSuccess || nb < 1) {
                    (void)hipGetLastError();
                    nb = 1;
                }
                c->ent_resident[k] = nb * c->num_cu;
            }
        lpw = 64;
        for (int k = 0; k < 3; k++)
            if ((count + (64 >> k) - 1) / (64 >> k) <= c->ent_resident[k]) { lpw = 64 >> k; break; }
    }
    if (np > 1)
        hipLaunchKernelGGL(vp8_entropy_parts_kernel, dim3((unsigned)((count + 64 / np - 1) / (64 / np))), dim3(64), 0, c->stream,
                           (const vp8hip_entropy_frame *)c->d_ent_frames2[set], count, np, (const uint8_t *)c->d_ent_data2[set], c->dg, data_bytes,
                           c->slot_block_dev, c->slot_bytes, c->o_mbx, c->o_blocks, first_slot, c->d_ent_scratch, c->d_ent_status);
    else
        hipLaunchKernelGGL(vp8_entropy_kernel, dim3((unsigned)((count + lpw - 1) / lpw)), dim3(64), vp8_entropy_lds_bytes(lpw), c->stream,
                           (const vp8hip_entropy_frame *)c->d_ent_frames2[set], count, lpw, (const uint8_t *)c->d_ent_data2[set], c->dg, data_bytes,
                           c->slot_block_dev, c->slot_bytes, c->o_mbx, c->o_blocks, c->o_mvs, first_slot, c->d_ent_scratch, c->d_ent_status,
                           c->pool, c->d_pool_ctr, c->pool_chunks, c->chunk_blocks);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_ent_out[set], c->stream));
    c->ent_last_count = count;
    return 0;
}

extern "C" int vp8hip_entropy_status(vp8hip_ctx *c, int count, uint32_t *status)
{
    if (!c || !status || count < 1 || count > c->ent_last_count)
        return fail(c, -2, "vp8hip_entropy_status: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(status, c->d_ent_status, (size_t)count * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vp8hip_entropy_status_async(vp8hip_ctx *c, int count, uint32_t *status)
{
    if (!c || !status || count < 1 || count > c->ent_last_count)
        return fail(c, -2, "vp8hip_entropy_status_async: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(status, c->d_ent_status, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    return 0;
}
