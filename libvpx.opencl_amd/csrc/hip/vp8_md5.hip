// MD5 of decoded frames on the device, one frame per LANE (gfx950).
//
// What it replaces: the consumer side of the reference's conformance tools -- `vpxdec --md5` (vpxdec.c:1080-1101 with
// md5_utils.c) and examples/decode_to_md5 (decode_to_md5.txt: MD5Update over the visible rows of the Y, U and V planes of every
// frame vpx_codec_get_frame returns).  In a pipeline that decodes a thousand frames a second the hashing alone keeps a third of
// the host's cores busy (3.1 MB per 1080p frame at the ≈ 0.6 GB/s one core hashes), cores the entropy decoder needs; the GPU
// has the frames already and is idle nine tenths of the time in that pipeline.
//
// MD5 is a chain over the 64-byte blocks of ONE message, so a frame is a lane's work, block after block (RFC 1321: 64 steps per
// block); the frames of a batch run side by side in the lanes of a few waves.  All frames of a context have one geometry, so the
// walk over planes, rows and blocks is wave-uniform (scalar); only the frame's base address is per lane.  The blocks are
// requested AHEAD blocks before they are hashed (each lane reads its own 64 bytes: nothing coalesces, only latency matters).
// Where rows are whole blocks -- display width a multiple of 128 -- there are two readers: the raster form of a frame buffer
// (vp8_md5_kernel), and the TILED form a large launch leaves (vp8_md5_tiles_kernel: macroblock-window tiles,
// vp8_keyframe_simt.hip) -- the hash is the consumer that proves a frame, and it takes the frame as the decoder left it, without
// a tiled -> raster pass in between.  Any other width: vp8_md5_any_kernel, the same chain over a message whose blocks straddle
// rows (raster form; word by word, byte by byte across a row's end) -- the odd sizes of the conformance streams, not the
// throughput path.  Integer only.
#include "vp8_common.hip.h"

namespace {

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef GLOBAL_AS const u32x4 *g_cu32x4p;
typedef GLOBAL_AS const u32 *g_cu32p_;

__device__ __forceinline__ u32 rol(u32 x, int s) { return __builtin_rotateleft32(x, (u32)s); }

// one block: state (a, b, c, d) += the 64 steps over message words m[0..15] (RFC 1321 section 3.4)
__device__ __forceinline__ void md5_block(u32 &A, u32 &B, u32 &C, u32 &D, const u32 (&m)[16])
{
    static constexpr u32 K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
    static constexpr int S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                   4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
    u32 a = A, b = B, c = C, d = D;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        u32 f;
        int g;
        if (i < 16) { f = d ^ (b & (c ^ d)); g = i; }
        else if (i < 32) { f = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        const u32 t = a + f + K[i] + m[g];
        a = d; d = c; c = b;
        b = b + rol(t, S[i]);
    }
    A += a; B += b; C += c; D += d;
}

// The 64 bytes of block bx of pixel row `row` of plane pl, out of the frame's TILES (layout: vp8_keyframe_simt.hip, KT_*): rows
// 0..11 of a macroblock row (chroma: 0..3) stand in the macroblocks' WINDOWS, shifted four pixels = one dword to the left -- a
// block's sixteen dwords are dwords 1.. of five (chroma: nine) neighbouring tiles' row pieces --, rows 12..15 (4..7)
// macroblock-aligned.  All offsets are wave-uniform; only `base` is the lane's.
__device__ __forceinline__ void tile_block(g_cu8p base, int cols, int pl, int row, int bx, u32x4 (&q)[4])
{
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    typedef GLOBAL_AS const u32x2 *g_cu32x2p;
    u32 m[16];
    if (pl == 0) {
        const int yy = row & 15;
        g_cu8p t = base + ((long)(row >> 4) * (cols + 1) + 4 * bx) * VP8_TILE_BYTES;
        if (yy >= 12) {
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = *(g_cu32x4p)(t + k * VP8_TILE_BYTES + 192 + 16 * (yy - 12));
            return;
        }
        t += 16 * yy;
        const u32x4 a = *(g_cu32x4p)t, b = *(g_cu32x4p)(t + VP8_TILE_BYTES), c = *(g_cu32x4p)(t + 2 * VP8_TILE_BYTES),
                    d = *(g_cu32x4p)(t + 3 * VP8_TILE_BYTES);
        const u32 e = *(g_cu32p_)(t + 4 * VP8_TILE_BYTES);
        q[0] = (u32x4){ a.y, a.z, a.w, b.x }; q[1] = (u32x4){ b.y, b.z, b.w, c.x };
        q[2] = (u32x4){ c.y, c.z, c.w, d.x }; q[3] = (u32x4){ d.y, d.z, d.w, e };
        return;
    }
    const int yy = row & 7;
    g_cu8p t = base + ((long)(row >> 3) * (cols + 1) + 8 * bx) * VP8_TILE_BYTES + 32 * (pl - 1);
    if (yy >= 4) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32x2 v = *(g_cu32x2p)(t + k * VP8_TILE_BYTES + 320 + 8 * (yy - 4));
            m[2 * k] = v.x; m[2 * k + 1] = v.y;
        }
    } else {
        t += 256 + 8 * yy;
        u32 prev = *(g_cu32p_)(t + 4);                 // the second dword of the first window's row piece
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32x2 v = *(g_cu32x2p)(t + (k + 1) * VP8_TILE_BYTES);
            m[2 * k] = prev; m[2 * k + 1] = v.x;
            prev = v.y;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = (u32x4){ m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3] };
}

// frames: frame buffer 0 of the pool (TILED: its tiles); fstride: bytes from one to the next; index: which frame buffer lane f
// hashes (null: first + f); w, h: display size (w a multiple of 128); out: 16 bytes per frame.  One lane per frame, 64-thread blocks.
template <bool TILED>
__device__ __forceinline__ void md5_frames(const uint8_t *__restrict__ frames, size_t fstride, const int *__restrict__ index, int first, int count,
                                           DevGeom g, int w, int h, uint8_t *__restrict__ out)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    const bool live = f < count;
    const int fbi = index ? index[live ? f : 0] : first + (live ? f : 0);
    g_cu8p base = (g_cu8p)(frames + fstride * (size_t)fbi);
    u32 A = 0x67452301u, B = 0xefcdab89u, C = 0x98badcfeu, D = 0x10325476u;
    constexpr int AHEAD = 4;
    // the walk: plane 0 = Y (h rows of w / 64 blocks), 1 = U, 2 = V ((h + 1) / 2 rows of w / 128 blocks)
    const int cw = w >> 1, ch = (h + 1) >> 1;
    const long nblk = (long)h * (w >> 6) + 2L * ch * (cw >> 6);
    // fetch cursor (AHEAD blocks in front of the hash cursor); wave-uniform
    int pl = 0, row = 0, bx = 0;
    auto fetch = [&](u32x4 (&q)[4]) {
        if constexpr (TILED) tile_block(base, g.mb_cols, pl, row, bx, q);
        else {
            const long off = pl == 0 ? g.y_off + (long)row * g.y_stride : (pl == 1 ? g.u_off : g.v_off) + (long)row * g.uv_stride;
            g_cu32x4p p = (g_cu32x4p)(base + off + 64L * bx);
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = p[k];
        }
        const int per_row = (pl == 0 ? w : cw) >> 6, rows = pl == 0 ? h : ch;
        if (++bx == per_row) { bx = 0; if (++row == rows) { row = 0; pl++; } }
    };
    u32x4 q[AHEAD][4];
#pragma unroll
    for (int i = 0; i < AHEAD; i++) {
        if (i < nblk) fetch(q[i]);
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) q[i][k] = (u32x4){ 0, 0, 0, 0 };
        }
    }
    for (long blk = 0; blk < nblk; blk += AHEAD) {
#pragma unroll
        for (int i = 0; i < AHEAD; i++) {
            if (blk + i < nblk) {
                const u32 m[16] = { q[i][0].x, q[i][0].y, q[i][0].z, q[i][0].w, q[i][1].x, q[i][1].y, q[i][1].z, q[i][1].w,
                                    q[i][2].x, q[i][2].y, q[i][2].z, q[i][2].w, q[i][3].x, q[i][3].y, q[i][3].z, q[i][3].w };
                if (blk + i + AHEAD < nblk) fetch(q[i]);
                md5_block(A, B, C, D, m);
            }
        }
    }
    // padding (RFC 1321 3.1-3.2): the message is a whole number of blocks, so one more: 0x80, zeros, the length in bits
    {
        const unsigned long long bits = (unsigned long long)nblk * 512ull;
        const u32 m[16] = { 0x80u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (u32)bits, (u32)(bits >> 32) };
        md5_block(A, B, C, D, m);
    }
    if (live) *(GLOBAL_AS u32x4 *)(out + 16 * (size_t)f) = (u32x4){ A, B, C, D };
}

// Any display size, raster form: the message is the visible rows of Y, U ((w + 1) / 2 x (h + 1) / 2), V back to back, so a
// 64-byte block starts anywhere in a row and may run over its end.  The walk stays wave-uniform (one geometry): a word that lies
// within a row is one load at whatever alignment, the others are put together from bytes; behind the last byte come 0x80, zeros
// and, in the last block, the length (RFC 1321 3.1-3.2).
__device__ __forceinline__ void md5_frames_any(const uint8_t *__restrict__ frames, size_t fstride, const int *__restrict__ index, int first,
                                               int count, DevGeom g, int w, int h, uint8_t *__restrict__ out)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    const bool live = f < count;
    const int fbi = index ? index[live ? f : 0] : first + (live ? f : 0);
    g_cu8p base = (g_cu8p)(frames + fstride * (size_t)fbi);
    u32 A = 0x67452301u, B = 0xefcdab89u, C = 0x98badcfeu, D = 0x10325476u;
    const int cw = (w + 1) >> 1, ch = (h + 1) >> 1;
    const long nbytes = (long)w * h + 2L * cw * ch;
    const long nblk = (nbytes + 9 + 63) >> 6;
    long left = nbytes;                                   // bytes of the message still to come; -1 once the 0x80 is out
    int pl = 0, row = 0, col = 0, roww = w;
    g_cu8p rp = base + g.y_off;
    auto next_row = [&]() {
        col = 0;
        if (++row == (pl == 0 ? h : ch)) { row = 0; pl++; roww = cw; }
        rp = base + (pl == 0 ? g.y_off + (long)row * g.y_stride : (pl == 1 ? g.u_off : g.v_off) + (long)row * g.uv_stride);
    };
    auto word = [&]() -> u32 {
        u32 v = 0;
        if (left >= 4 && col + 4 <= roww) {
            __builtin_memcpy(&v, (const void *)(rp + col), 4);
            col += 4; left -= 4;
            if (col == roww && left > 0) next_row();
            return v;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 b = 0;
            if (left > 0) {
                b = rp[col];
                left--;
                if (++col == roww && left > 0) next_row();
            } else if (left == 0) {
                b = 0x80u; left = -1;
            }
            v |= b << (8 * k);
        }
        return v;
    };
    for (long blk = 0; blk < nblk; blk++) {
        u32 m[16];
#pragma unroll
        for (int j = 0; j < 16; j++) m[j] = word();
        if (blk == nblk - 1) {
            const unsigned long long bits = (unsigned long long)nbytes * 8ull;
            m[14] = (u32)bits; m[15] = (u32)(bits >> 32);
        }
        md5_block(A, B, C, D, m);
    }
    if (live) *(GLOBAL_AS u32x4 *)(out + 16 * (size_t)f) = (u32x4){ A, B, C, D };
}

} // namespace

extern "C" __global__ void __launch_bounds__(64)
vp8_md5_kernel(const uint8_t *__restrict__ frames, size_t fstride, const int *__restrict__ index, int first, int count, DevGeom g, int w, int h,
               uint8_t *__restrict__ out)
{
    md5_frames<false>(frames, fstride, index, first, count, g, w, h, out);
}
extern "C" __global__ void __launch_bounds__(64)
vp8_md5_tiles_kernel(const uint8_t *__restrict__ tiles, size_t tstride, const int *__restrict__ index, int first, int count, DevGeom g, int w, int h,
                     uint8_t *__restrict__ out)
{
    md5_frames<true>(tiles, tstride, index, first, count, g, w, h, out);
}
extern "C" __global__ void __launch_bounds__(64)
vp8_md5_any_kernel(const uint8_t *__restrict__ frames, size_t fstride, const int *__restrict__ index, int first, int count, DevGeom g, int w, int h,
                   uint8_t *__restrict__ out)
{
    md5_frames_any(frames, fstride, index, first, count, g, w, h, out);
}
