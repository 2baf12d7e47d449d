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
// requested AHEAD blocks before they are hashed (each lane reads its own 64 bytes: nothing coalesces, only latency matters;
// round 5: by loads of ONE shape outside any branch -- the round-4 kernel's ring drained at every block, see md5_run).
// What a batch costs (1080p, tiles): up to 4096 frames 37-38 ms -- the chain: 48.6 k blocks x 320 dependent-ish instructions of a lone
// wave --, 8192 frames 43.6, 12,288 59, 16,384 80.6 ms (round 4: 67 ms for 8192, 86 for 16,384).  The climb is the memory system, not the
// chain (loads compiled out: 36.8 ms for 16,384): a row piece is 16 bytes of a 128-byte line that holds eight rows of a tile, the line
// comes back for each of them, and between two rows of a frame all the other frames' lines go by -- 16,384 frames x 121 tiles x 128 B =
// 254 MB of lines in use, the size of the Infinity Cache; which is why fetch_impl (vp8hip.hip) hashes batches of 12,288 tiled frames and
// more from a packed I420 copy with vp8_md5_kernel (pack 25 ms + hash 40 ms = 65.6 ms for 16,384 frames).  A variant in which the WAVE fetched (one global_load_lds_dwordx4 for the
// next chunk of three frames' row, lanes hashing out of LDS regions) was built, bit-exact, and no faster: the same lines, the same
// wall, and 40 % more instructions beside the chain (50 ms for any batch up to 8192 frames; 82 for 16,384) -- not kept.
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


// One block: state (a, b, c, d) += the 64 steps over message words m[0..15] (RFC 1321 section 3.4; the reference's MD5Transform,
// md5_utils.c:167-245).  A step is five vector instructions, four of them a dependent chain -- the round function of (b, c, d) as ONE
// three-input boolean (v_bitop3_b32; truth table over (s0, s1, s2) = (b, c, d), bit s0 << 2 | s1 << 1 | s2 of the immediate: F
// (b & c) | (~b & d) = 0xca, G (b & d) | (c & ~d) = 0xe4, H b ^ c ^ d = 0x96, I c ^ (b | ~d) = 0x39), a + (m + K) + f as one
// v_add3_u32, the rotation (v_alignbit_b32 by 32 - s), + b -- and a frame's hash is 48.6 k blocks of them one after the other, so
// the chain IS the kernel's time: a lone wave issues a dependent vector instruction every ~7 cycles.  Written as assembly, four
// steps a statement, because nothing else keeps the fifth instruction, m + K, where it costs least (in the shadow of the round
// function): the compiler's schedulers see no latency between dependent vector instructions and hoist all 64 sums to the top of
// the block (+150 registers), and an empty asm to pin each costs a wait-state s_nop.  Registers rotate by role as in RFC 1321's
// own listing: step 4q writes a, 4q+1 d, 4q+2 c, 4q+3 b.  (The round-4 kernel: six instructions a step, five of them dependent.)
#define MD5_STEP(A, B, C, D, J, K, R, OP)                                           \
    "v_bitop3_b32 %[f], %[" B "], %[" C "], %[" D "] bitop3:" OP "\n\t"             \
    "v_add_u32_e32 %[u], " K ", %[x" J "]\n\t"                                      \
    "v_add3_u32 %[" A "], %[" A "], %[u], %[f]\n\t"                                 \
    "v_alignbit_b32 %[" A "], %[" A "], %[" A "], " R "\n\t"                        \
    "v_add_u32_e32 %[" A "], %[" A "], %[" B "]\n\t"
#define MD5_QUAD(K0, K1, K2, K3, R0, R1, R2, R3, OP, G0, G1, G2, G3)                                                                       \
    asm(MD5_STEP("a", "b", "c", "d", "0", K0, R0, OP) MD5_STEP("d", "a", "b", "c", "1", K1, R1, OP)                                        \
        MD5_STEP("c", "d", "a", "b", "2", K2, R2, OP) MD5_STEP("b", "c", "d", "a", "3", K3, R3, OP)                                        \
        : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [f] "=&v"(f), [u] "=&v"(u)                                                  \
        : [x0] "v"(m[G0]), [x1] "v"(m[G1]), [x2] "v"(m[G2]), [x3] "v"(m[G3]));
__device__ __forceinline__ void md5_block(u32 (&st)[4], const u32 (&m)[16])
{
    u32 a = st[0], b = st[1], c = st[2], d = st[3], f, u;
    // (K[4q .. 4q+3], 32 - s, the round function, the message words' indices g: RFC 1321 3.4, rounds 1-4)
    MD5_QUAD("0xd76aa478", "0xe8c7b756", "0x242070db", "0xc1bdceee", "25", "20", "15", "10", "0xca",  0,  1,  2,  3)
    MD5_QUAD("0xf57c0faf", "0x4787c62a", "0xa8304613", "0xfd469501", "25", "20", "15", "10", "0xca",  4,  5,  6,  7)
    MD5_QUAD("0x698098d8", "0x8b44f7af", "0xffff5bb1", "0x895cd7be", "25", "20", "15", "10", "0xca",  8,  9, 10, 11)
    MD5_QUAD("0x6b901122", "0xfd987193", "0xa679438e", "0x49b40821", "25", "20", "15", "10", "0xca", 12, 13, 14, 15)
    MD5_QUAD("0xf61e2562", "0xc040b340", "0x265e5a51", "0xe9b6c7aa", "27", "23", "18", "12", "0xe4",  1,  6, 11,  0)
    MD5_QUAD("0xd62f105d", "0x02441453", "0xd8a1e681", "0xe7d3fbc8", "27", "23", "18", "12", "0xe4",  5, 10, 15,  4)
    MD5_QUAD("0x21e1cde6", "0xc33707d6", "0xf4d50d87", "0x455a14ed", "27", "23", "18", "12", "0xe4",  9, 14,  3,  8)
    MD5_QUAD("0xa9e3e905", "0xfcefa3f8", "0x676f02d9", "0x8d2a4c8a", "27", "23", "18", "12", "0xe4", 13,  2,  7, 12)
    MD5_QUAD("0xfffa3942", "0x8771f681", "0x6d9d6122", "0xfde5380c", "28", "21", "16", "9", "0x96",  5,  8, 11, 14)
    MD5_QUAD("0xa4beea44", "0x4bdecfa9", "0xf6bb4b60", "0xbebfbc70", "28", "21", "16", "9", "0x96",  1,  4,  7, 10)
    MD5_QUAD("0x289b7ec6", "0xeaa127fa", "0xd4ef3085", "0x04881d05", "28", "21", "16", "9", "0x96", 13,  0,  3,  6)
    MD5_QUAD("0xd9d4d039", "0xe6db99e5", "0x1fa27cf8", "0xc4ac5665", "28", "21", "16", "9", "0x96",  9, 12, 15,  2)
    MD5_QUAD("0xf4292244", "0x432aff97", "0xab9423a7", "0xfc93a039", "26", "22", "17", "11", "0x39",  0,  7, 14,  5)
    MD5_QUAD("0x655b59c3", "0x8f0ccc92", "0xffeff47d", "0x85845dd1", "26", "22", "17", "11", "0x39", 12,  3, 10,  1)
    MD5_QUAD("0x6fa87e4f", "0xfe2ce6e0", "0xa3014314", "0x4e0811a1", "26", "22", "17", "11", "0x39",  8, 15,  6, 13)
    MD5_QUAD("0xf7537e82", "0xbd3af235", "0x2ad7d2bb", "0xeb86d391", "26", "22", "17", "11", "0x39",  4, 11,  2,  9)
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}
#undef MD5_QUAD
#undef MD5_STEP

// A frame's TILES (layout: vp8_keyframe_simt.hip, KT_*): rows 0..11 of a macroblock row (chroma: 0..3) stand in the macroblocks'
// WINDOWS, shifted four pixels = one dword to the left -- the sixteen dwords of a 64-byte block of such a row are dwords 1.. of five
// (chroma: nine) neighbouring tiles' row pieces --, rows 12..15 (chroma 4..7) are macroblock-aligned.  Where block bx of pixel row
// `row` of plane pl begins (wave-uniform; the loads' own offsets are immediates):
__device__ __forceinline__ long tile_offset(int cols, int pl, int row, int bx)
{
    if (pl == 0) return ((long)(row >> 4) * (cols + 1) + 4 * bx) * VP8_TILE_BYTES + 16 * (row & 15);
    const int yy = row & 7;
    return ((long)(row >> 3) * (cols + 1) + 8 * bx) * VP8_TILE_BYTES + 32 * (pl - 1) + (yy >= 4 ? 320 + 8 * (yy - 4) : 256 + 8 * yy);
}

// What a lane holds of a block on its way: the SAME loads whatever the row's kind (a load inside a branch, even a wave-uniform
// one, makes the compiler drain every load in flight where the paths join: the ring below would be no ring), taken apart when the
// block is hashed.  Luma (SHAPE 0): the row pieces of four tiles and the first dword of the fifth -- a window row's block is dwords
// 1..16 of those seventeen, a bottom row's dwords 0..15.  Chroma (SHAPE 1): eight 8-byte row pieces -- tiles 1..8 for a window row,
// 0..7 for a bottom row: the address differs, not the instruction -- and the second dword of tile 0's piece (window rows only).
// Raster form (SHAPE 2): the block as it lies.
template <int SHAPE> struct Blk;
template <> struct Blk<0> { u32x4 v[4]; u32 e; };
template <> struct Blk<1> { u32 x[8], y[8], p; };
template <> struct Blk<2> { u32x4 v[4]; };
template <int SHAPE>
__device__ __forceinline__ void load_block(g_cu8p t, bool win, Blk<SHAPE> &q)
{
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    typedef GLOBAL_AS const u32x2 *g_cu32x2p;
#ifdef MD5_NOLOAD          // (timing experiments only: the chain without its loads)
    if constexpr (SHAPE == 1) { q.p = (u32)(unsigned long)t; for (int k = 0; k < 8; k++) { q.x[k] = q.p + k; q.y[k] = q.p ^ k; } }
    else { for (int k = 0; k < 4; k++) q.v[k] = (u32x4){ (u32)(unsigned long)t, (u32)k, (u32)win, 7u }; if constexpr (SHAPE == 0) q.e = 1; }
    return;
#endif
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) q.v[k] = *(g_cu32x4p)(t + k * VP8_TILE_BYTES);
        q.e = *(g_cu32p_)(t + 4 * VP8_TILE_BYTES);
    } else if constexpr (SHAPE == 1) {
        q.p = *(g_cu32p_)(t + 4);
        g_cu8p t1 = t + (win ? VP8_TILE_BYTES : 0);
#pragma unroll
        for (int k = 0; k < 8; k++) { const u32x2 v = *(g_cu32x2p)(t1 + k * VP8_TILE_BYTES); q.x[k] = v.x; q.y[k] = v.y; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) q.v[k] = ((g_cu32x4p)t)[k];
    }
}
// (the kind of the row selects the words with a wave-uniform condition -- sixteen selects a block --, not the code: a second copy
// of the 640 instructions of md5_blocks per kind, ring slot and shape does not fit the instruction cache)
template <int SHAPE>
__device__ __forceinline__ void block_words(const Blk<SHAPE> &q, bool win, u32 (&m)[16])
{
    if constexpr (SHAPE == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            m[2 * k] = win ? (k ? q.y[k - 1] : q.p) : q.x[k];
            m[2 * k + 1] = win ? q.x[k] : q.y[k];
        }
    } else if constexpr (SHAPE == 0) {
        const u32 w[17] = { q.v[0].x, q.v[0].y, q.v[0].z, q.v[0].w, q.v[1].x, q.v[1].y, q.v[1].z, q.v[1].w, q.v[2].x, q.v[2].y, q.v[2].z, q.v[2].w,
                            q.v[3].x, q.v[3].y, q.v[3].z, q.v[3].w, q.e };
#pragma unroll
        for (int j = 0; j < 16; j++) m[j] = win ? w[j + 1] : w[j];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { m[4 * k] = q.v[k].x; m[4 * k + 1] = q.v[k].y; m[4 * k + 2] = q.v[k].z; m[4 * k + 3] = q.v[k].w; }
    }
}

// A run of `nb` blocks of one shape -- the luma plane, or the two chroma planes -- of the lane's frame: the blocks are requested
// AHEAD blocks before they are hashed (each lane reads its own 64 bytes: nothing coalesces, only latency matters), by loads that
// are unconditional and of one shape, so that the waits the compiler inserts count the loads behind them instead of draining.
// pl0 / npl: first plane and number of planes of the run, rows x per_row blocks each.
template <bool TILED, int SHAPE>
__device__ __forceinline__ void md5_run(g_cu8p base, u32 (&st)[4], const DevGeom &g, int pl0, int npl, int rows, int per_row)
{
    constexpr int AHEAD = 3;
    const long nb = (long)npl * rows * per_row;
    // fetch cursor, wave-uniform; behind the run's last block it stays on it
    int pl = pl0, row = 0, bx = 0;
    long fetched = 0;
    Blk<SHAPE> q[AHEAD];
    bool qwin[AHEAD];
    auto fetch = [&](Blk<SHAPE> &qq, bool &win) {
        long off;
        if constexpr (TILED) {
            off = tile_offset(g.mb_cols, pl, row, bx);
            win = pl == 0 ? (row & 15) < 12 : (row & 7) < 4;
        } else {
            off = (pl == 0 ? g.y_off + (long)row * g.y_stride : (pl == 1 ? g.u_off : g.v_off) + (long)row * g.uv_stride) + 64L * bx;
            win = false;
        }
        load_block<SHAPE>(base + off, win, qq);
        if (fetched + 1 < nb) {
            fetched++;
            if (++bx == per_row) { bx = 0; if (++row == rows) { row = 0; pl++; } }
        }
    };
    auto hash = [&](const Blk<SHAPE> &qq, bool win) {
        u32 m[16];
        block_words<SHAPE>(qq, win, m);
        md5_block(st, m);
    };
#pragma unroll
    for (int i = 0; i < AHEAD; i++) fetch(q[i], qwin[i]);
    long n = 0;
    for (; n + AHEAD <= nb; n += AHEAD) {
#pragma unroll
        for (int i = 0; i < AHEAD; i++) {
            hash(q[i], qwin[i]);
            fetch(q[i], qwin[i]);
        }
    }
    // the tail: fewer than AHEAD blocks, all of them in the ring
#pragma unroll
    for (int i = 0; i < AHEAD - 1; i++)
        if (n + i < nb) hash(q[i], qwin[i]);
}

// frames: frame buffer 0 of the pool (TILED: its tiles); fstride: bytes from one to the next; index: which frame buffer lane f
// hashes (null: first + f); w, h: display size (w a multiple of 128); out: 16 bytes per frame.  One lane per frame, 64-thread blocks.
// (Two frames per lane, their chains interleaved instruction by instruction, double what a SIMD hashes and slow every frame down by
// a third: worth it from 65,536 frames per launch on, where the chip runs out of SIMDs -- no pipeline here holds that many.)
template <bool TILED>
__device__ __forceinline__ void md5_frames(const uint8_t *__restrict__ frames, size_t fstride, const int *__restrict__ index, int first, int count,
                                           DevGeom g, int w, int h, uint8_t *__restrict__ out)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    const int fc = f < count ? f : count - 1;                       // (a lane without a frame reads the last one's and stores nothing)
    const int fbi = index ? index[fc] : first + fc;
    g_cu8p base = (g_cu8p)(frames + fstride * (size_t)fbi);
    u32 st[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
    // the message: plane 0 = Y (h rows of w / 64 blocks), 1 = U, 2 = V ((h + 1) / 2 rows of w / 128 blocks)
    const int cw = w >> 1, ch = (h + 1) >> 1;
    const long nblk = (long)h * (w >> 6) + 2L * ch * (cw >> 6);
    md5_run<TILED, TILED ? 0 : 2>(base, st, g, 0, 1, h, w >> 6);
    md5_run<TILED, TILED ? 1 : 2>(base, st, g, 1, 2, ch, cw >> 6);
    // padding (RFC 1321 3.1-3.2): the message is a whole number of blocks, so one more: 0x80, zeros, the length in bits
    {
        const unsigned long long bits = (unsigned long long)nblk * 512ull;
        const u32 m[16] = { 0x80u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (u32)bits, (u32)(bits >> 32) };
        md5_block(st, m);
    }
    if (f < count) *(GLOBAL_AS u32x4 *)(out + 16 * (size_t)f) = (u32x4){ st[0], st[1], st[2], st[3] };
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
        u32 st[4] = { A, B, C, D };
        md5_block(st, m);
        A = st[0]; B = st[1]; C = st[2]; D = st[3];
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
