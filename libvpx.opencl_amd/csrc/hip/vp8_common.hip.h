// Shared device-side definitions for the gfx950 VP8 pixel-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vp8_ir.h"

// One frame of work as the kernels see it (device memory, one entry per job of a launch).
struct DevJob {
    vp8ir_frame_hdr hdr;          // 64 B
    const vp8ir_mbx *mbx;         // the slot's macroblock records (include/vp8_ir.h, the device form): 128 B each
    const int16_t  *blocks;       // ... and its block stream: 32 B per block with eob > 1
    const vp8ir_mv *mvs;
    uint8_t        *dst;
    const uint8_t  *ref[4];       // [1..3] = last / golden / alt-ref frame buffers (inter frames); [0] unused
    uint8_t        *tile;         // lane-per-row (key-frame) pipeline: the job's macroblock-tiled scratch frame (VP8_TILE_BYTES per macroblock)
    const uint8_t  *ref_tile[3];  // the TILED forms of ref[1..3] (null where a reference has none): vp8_inter_pred_tiles_kernel
    // 64 + 8*8 + 4*8 = 160 B
};
static_assert(sizeof(DevJob) == 160, "DevJob layout");

// Frame geometry common to all jobs of a launch (vp8ir_geom, flattened for kernel args).
struct DevGeom {
    int mb_cols, mb_rows;
    int aligned_w, aligned_h;
    int y_stride, uv_stride;
    int y_off, u_off, v_off;
};

#define WAVE 64

// Macroblock tiles of the one-MB-row-per-lane pipeline (vp8_keyframe_simt.hip has the layout): three 128-byte lines per macroblock.
#define VP8_TILE_BYTES 384

// Pointers that come out of a DevJob (i.e. out of memory) are generic to the compiler, which then
// emits FLAT loads/stores: those count on lgkmcnt as well as vmcnt, so every LDS wait would also
// wait for the global prefetches and the frame write-out.  Casting to the global address space
// makes them global_load/global_store (vmcnt only).
#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS unsigned char *g_u8p;
typedef GLOBAL_AS const unsigned char *g_cu8p;
typedef GLOBAL_AS unsigned int *g_u32p;
typedef GLOBAL_AS const unsigned int *g_cu32p;
typedef GLOBAL_AS const short *g_cs16p;

// ---- intra-workgroup progress flags in LDS ------------------------------------------------
// A wave publishes "(row sequence number << 16) | MBs finished in that row"; finishing a row
// publishes (seq+1) << 16.  All waves of a workgroup live on one CU.
//
// Two flavours, chosen by where the handed-over DATA lives:
//  * data in LDS (recon kernel's line buffers): a wave's LDS operations are executed in issue
//    order, so "write data; write flag" / "read flag; read data" need NO s_waitcnt at all -- only
//    the compiler must be kept from reordering.  In particular the publisher does not wait for its
//    outstanding global stores (frame write-out) and the consumer does not drain its prefetches.
//  * data in global memory (loop-filter kernel's context rows): the publisher must have its stores
//    acknowledged (s_waitcnt vmcnt(0)) before the flag store; same CU => same L1/L2, so workgroup
//    scope needs no cache maintenance (LLVM AMDGPU memory model, non-tgsplit mode).
__device__ __forceinline__ void compiler_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef VP8_POLL_SLEEP
#define VP8_POLL_SLEEP 1   // s_sleep units (64 clocks) between polls of a progress flag
#endif
__device__ __forceinline__ void wg_wait_ge(int *flag, int value)
{
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < value)
        __builtin_amdgcn_s_sleep(VP8_POLL_SLEEP);
    compiler_fence();
}

__device__ __forceinline__ void wg_publish_lds(int *flag, int value, int lane)
{
    compiler_fence();
    if (lane == 0)
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void wg_publish_global(int *flag, int value, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    compiler_fence();
    if (lane == 0)
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Lanes of one wave exchanging data through LDS: the hardware issues a wave's LDS operations in
// order, but the COMPILER only promises per-thread ordering and may move one lane's load above
// another lane's store.  This is the (instruction-free) fence that pins the order.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// quantiser lookups (vp8/common/quant_common.c:14-37); VP8 format constants
__constant__ static const unsigned short k_dc_q[128] = {
    4, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13, 14, 15, 16, 17, 17, 18, 19, 20, 20, 21, 21, 22, 22, 23, 23, 24, 25, 25, 26,
    27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 46, 47, 48, 49, 50, 51, 52,
    53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 76, 77, 78, 79,
    80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 91, 93, 95, 96, 98, 100, 101, 102, 104, 106, 108, 110, 112, 114, 116,
    118, 122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157
};
__constant__ static const unsigned short k_ac_q[128] = {
    4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
    34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 60, 62, 64,
    66, 68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88, 90, 92, 94, 96, 98, 100, 102, 104, 106, 108, 110, 112, 114, 116,
    119, 122, 125, 128, 131, 134, 137, 140, 143, 146, 149, 152, 155, 158, 161, 164, 167, 170, 173, 177, 181, 185,
    189, 193, 197, 201, 205, 209, 213, 217, 221, 225, 229, 234, 239, 245, 249, 254, 259, 264, 269, 274, 279, 284
};

// ---- in-kernel stamps (diagnostic builds only: -DVP8_STAMPS; never in the product build) ------------------
// Where a lane-per-row kernel's step spends its cycles: STAMP(i) adds the shader cycles since the previous stamp
// to bucket i (wave-uniform, kept in SGPRs); the first wave of the grid adds its buckets to a device array of its
// own that no kernel reads (MI355X guide, "In-kernel stamps").  Shares only -- the build itself runs slower.
// ---- granule hand-over between the workgroups of a frame (vp8_recon_xcu_kernel / vp8_loopfilter_xcu_kernel) ----
// A granule is 4 bytes of data and the tag (launch counter) of the launch that wrote it in one 8-byte word, stored and
// loaded with one relaxed agent-scope access: the tag is the progress flag, there is no separate flag and no fence.
// A bounded poll turns a broken hand-over into an error status instead of a hang: the first poll that runs out sets
// *err (host-visible) and marks the launch in vp8_gran_broken; from then on every wait of that launch gives up after one
// look, so the kernel drains in milliseconds instead of repeating the full budget per macroblock.
typedef unsigned long long u64;
typedef GLOBAL_AS u64 *g_u64p;
extern __device__ unsigned int vp8_gran_broken;     // tag of the last launch in which a hand-over timed out (vp8hip.hip)
__device__ __forceinline__ void gran_store(g_u64p p, unsigned int data, unsigned int tag)
{
    __hip_atomic_store(p, (u64)data | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(g_u64p p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// v: what an earlier gran_load of *p returned; polls only if that was too early
__device__ __forceinline__ unsigned int gran_wait(g_u64p p, u64 v, unsigned int tag, int *err, int code)
{
    int budget = 0;
    for (int n = 0; (unsigned int)(v >> 32) != tag; ++n) {
        if (n == 0) budget = __hip_atomic_load(&vp8_gran_broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag ? 1 : (1 << 22);
        if (n >= budget) {
            *err = code;
            __hip_atomic_store(&vp8_gran_broken, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(VP8_POLL_SLEEP);
        v = gran_load(p);
    }
    return (unsigned int)v;
}

#ifdef VP8_STAMPS
#define VP8_NSTAMPS 16
extern __device__ unsigned long long vp8_stamps_recon[VP8_NSTAMPS], vp8_stamps_lf[VP8_NSTAMPS];
#define STAMP_DECL unsigned long long st_acc[VP8_NSTAMPS]; unsigned long long st_last; \
    for (int i_ = 0; i_ < VP8_NSTAMPS; i_++) st_acc[i_] = 0; \
    { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    st_acc[i] += t_ - st_last; st_last = t_; }
#define STAMP_FLUSH(arr) if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < VP8_NSTAMPS; i_++) atomicAdd(&arr[i_], st_acc[i_]); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(arr)
#endif
