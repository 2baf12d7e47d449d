/* batch_md5 [--threads T] [--batch B] [--loop N] [--host-md5] [--device-entropy [--entropy-batch E [--pool-mb M]] [--no-download]] [--gpus G] <in.ivf> <out.md5>
 * batch_md5 --streams S [--threads T] [--gpus G] <in.ivf> [<in2.ivf> ...] <out.md5>
 *
 * decode_to_md5 for streams of independently decodable frames (all key frames), at the rate the host can feed the
 * GPU: SURVEY.md 8(f)1.  The output file has decode_to_md5's lines ("<md5>  img-<w>x<h>-<%04d>.i420", one per frame,
 * in stream order) and is byte-identical to it.
 *
 * The reference decodes one frame at a time on one thread (decoder_tmpl.c:47-103).  Key frames reset every piece of
 * decoder state (decodframe.c:610-639), so here T feeder threads run the entropy decoder (vp8_parser.h) on different
 * frames at once, each writing the IR -- in the device form of include/vp8_ir.h, the form the kernels read -- straight into the
 * pinned staging of an IR slot (vp8hip_ir_map_compact); the main thread uploads a batch of B slots (a copy each), launches the pixel path for the whole batch (vp8hip_decode), downloads the previous
 * batch into pinned host memory, and the same pool hashes it.  Three slot / frame-buffer sets rotate: batch k+1 is
 * parsed while batch k is on the GPU and batch k-1 is downloaded and hashed.  --loop repeats the stream N times
 * (benchmarking; the listing then has N * frames lines).
 *
 * --device-entropy: the feeder threads only read the frame headers (vp8_parser_begin_frame, vp8_parser_export_entropy) and
 * copy the compressed frames into page-locked memory; the macroblocks' modes and coefficient tokens are decoded on the GPU, a
 * frame per lane (vp8hip_entropy_decode), straight into the IR slots the pixel path reads.  A lane takes about a second for a
 * large 1080p key frame whatever the batch, so this mode wants batches of thousands (one set of IR slots, two sets of frame
 * buffers).  --no-download: with the MD5s computed on the device the frames themselves stay there.  --entropy-batch E (a
 * multiple of B): the entropy decoder takes E frames per launch and the pixel path decodes them B at a time -- more frames in
 * flight (frames in flight over the time of the largest is what the entropy decoder's rate is; 24,576 is what the device holds
 * at once, vp8hip_entropy.hip).  The E slots hold records and vectors only and the blocks of a launch come out of ONE pool
 * (vp8hip_configure_pooled; --pool-mb M sets its size, by default what the largest launch is expected to need; a launch that finds
 * the pool empty ends the attempt, and the work is run again with twice the pool, up to three times: pooled_attempts): a frame in
 * flight costs what it needs, not the worst case.  With --no-download the launch's frames are hashed in one go (E frame buffers,
 * as tiles), beside the next launch of the entropy decoder.  Frames the device reports as cut short (vp8hip_entropy_status) are
 * counted and named on stderr, as the host feeder's *corrupt would.
 *
 * --streams S: S streams of ANY frame types decoded side by side, position t of all of them in one launch (stream s plays input
 * s mod inputs; all inputs of one frame size; the shortest sets the length): only the frame headers are read on the host, a
 * stream's frames go through ONE IR slot and four frame buffers of its own, the segment map a stream keeps from frame to frame
 * stays on the device with them (vp8_parser_set_device_segmap), every shown frame is hashed on the device
 * (vp8hip_frames_md5_list_async).  The listing has decode_to_md5's lines stream after stream ("stream<s>/img-..." labels when
 * S > 1).  This is how inter-frame streams use the device's entropy decoder: a stream's frames depend on each other, streams do not.
 *
 * --gpus G: G worker processes, one per device (one feeder pool and one context each), over contiguous shares of the work --
 * frames of the looped stream, or streams --; the listings are merged in order.  VP8BATCH_SINGLE_DEVICE=1 (test boxes with one
 * GPU) puts every worker on device 0.
 *
 * Prints frames, seconds and frames/s for the region "first byte parsed .. last digest done" on stderr. */
#define _GNU_SOURCE
#include <sched.h>
#include <sys/wait.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "ivf.h"
#include "md5.h"
#include "vp8_parser.h"
#include "vp8hip.h"

/* ---- a tiny pool: up to two "parallel for" tasks in flight, indices handed out one at a time ---------------------- */
typedef void (*task_fn)(void *arg, int index, int worker);
typedef struct task {
    task_fn fn; void *arg;
    int n, next, done;              /* guarded by pool.mu */
} task;
static struct {
    pthread_mutex_t mu; pthread_cond_t work, finished;
    task *slot[2];                  /* slot 0 is served first (the feeder), then slot 1 (hashing) */
    int stop;
} pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, { NULL, NULL }, 0 };

static void *worker_main(void *idp)
{
    const int id = (int)(size_t)idp;
    pthread_mutex_lock(&pool.mu);
    for (;;) {
        task *t = NULL;
        for (int s = 0; s < 2 && !t; s++)
            if (pool.slot[s] && pool.slot[s]->next < pool.slot[s]->n) t = pool.slot[s];
        if (!t) {
            if (pool.stop) break;
            pthread_cond_wait(&pool.work, &pool.mu);
            continue;
        }
        const int i = t->next++;
        pthread_mutex_unlock(&pool.mu);
        t->fn(t->arg, i, id);
        pthread_mutex_lock(&pool.mu);
        if (++t->done == t->n) pthread_cond_broadcast(&pool.finished);
    }
    pthread_mutex_unlock(&pool.mu);
    return NULL;
}
static void task_start(task *t, int slot, task_fn fn, void *arg, int n)
{
    pthread_mutex_lock(&pool.mu);
    t->fn = fn; t->arg = arg; t->n = n; t->next = 0; t->done = 0;
    pool.slot[slot] = t;
    pthread_cond_broadcast(&pool.work);
    pthread_mutex_unlock(&pool.mu);
}
static void task_wait(task *t, int slot)
{
    pthread_mutex_lock(&pool.mu);
    while (t->done < t->n) pthread_cond_wait(&pool.finished, &pool.mu);
    if (pool.slot[slot] == t) pool.slot[slot] = NULL;
    pthread_mutex_unlock(&pool.mu);
}

/* ---- the stream in memory -------------------------------------------------------------------------------------------- */
typedef struct frame { uint8_t *data; size_t size; } frame;
static frame *g_frames; static int g_nframes;
static long g_first;                                /* --gpus: this worker's share starts at frame g_first of the looped stream */
#define FRAME_AT(k) (&g_frames[(g_first + (k)) % g_nframes])

/* ---- per-run state shared with the workers ------------------------------------------------------------------------ */
static vp8hip_ctx *g_hip;
static vp8_parser **g_parsers;                      /* one per worker */
static int g_batch, g_width, g_height;
static vp8ir_geom g_geom;
/* 3 * batch slots: their pinned staging, in the device form of include/vp8_ir.h (0.4 of the dense bytes) */
static struct { vp8ir_frame_hdr *hdr; vp8ir_mbx *mbx; int16_t *blocks; size_t cap, nblocks; vp8ir_mv *mvs; } *g_maps;
static uint8_t *g_host[2];                          /* 2 x batch pinned frame buffers: one being filled by the GPU, one being hashed */
static size_t g_stride;                             /* bytes from one frame buffer to the next */
static int g_packed;                                /* downloads as packed I420 (vp8hip_frames_fetch_i420_async: a tenth less over the link than
                                                       whole frame buffers) -- whenever the digests come from the device and nobody here
                                                       reads the frames by their strides */
static unsigned char (*g_digest)[16];               /* one per frame of the whole run */
static uint8_t *g_dig[2];                           /* the digests of a batch as the device computed them (pinned), beside g_host[] */
static int g_dev_md5;                               /* hash on the device (vp8hip_frames_fetch_async: widths that are multiples of 128),
                                                       the feeder keeps the host's cores */
static volatile int g_failed;
static int g_dev_entropy;                           /* --device-entropy */
static int g_ebatch;                                /* --entropy-batch: frames per entropy launch (0: = g_batch, dense) */
static uint32_t *g_ent_status[3];                   /* per set: the device's status words of the launch (pinned) */
static long g_corrupt;                              /* frames whose partitions ended early */
static long *g_order;                               /* --device-entropy: which frame of the run the k-th processed one is.  A lane of the
                                                       entropy kernel is busy for as long as its frame is large and a launch lasts as long
                                                       as its longest lane, so the frames of SORT_WINDOW batches at a time are taken
                                                       largest first: the frames of a launch are of a size */
#define SORT_WINDOW 16
#define ESET(b) ((b) % 3)                           /* three page-locked sets of a launch's input: the headers of launch L + 1 are read while
                                                       launch L is being queued and launch L - 1 is still on the device */
static vp8hip_entropy_frame *g_ent[3];              /* per set: the frames' descriptions for vp8hip_entropy_decode (pinned) */
static uint8_t *g_ent_data[3];                      /* ... and their bytes, one after the other */
static size_t g_ent_cap;
static size_t g_pool_bytes;                         /* --entropy-batch: the block pool of the context */

typedef struct batch_ref { int b, n; long first; } batch_ref;     /* batch number, frames in it, index of its first frame */
/* The pipeline, batch by batch (three slot / frame-buffer sets, two pinned host sets):
 *   feeder threads   parse batch b+1 into slot set (b+1)%3            (the device form, include/vp8_ir.h)
 *   this thread      uploads batch b (one copy per frame), launches its pixel path, then asks for the
 *                    frames back: ONE asynchronous device-to-host copy of the whole batch on a stream of its own
 *   hash threads     MD5 of batch b-1, which that copy delivered during the previous iteration */

static inline long run_index(long k) { return g_order ? g_order[k] : k; }
static int by_size_desc(const void *pa, const void *pb)
{
    const long a = *(const long *)pa, b = *(const long *)pb;
    const size_t sa = FRAME_AT(a)->size, sb = FRAME_AT(b)->size;
    return sa != sb ? (sa < sb ? 1 : -1) : (a < b ? -1 : a > b);
}

static void parse_one(void *arg, int i, int worker)
{
    const batch_ref *br = (const batch_ref *)arg;
    const frame *f = FRAME_AT(br->first + i);
    const int slot = (br->b % 3) * g_batch + i;
    vp8ir_frame_hdr hdr;
    int rc = vp8_parser_begin_frame(g_parsers[worker], f->data, f->size, &hdr);
    if (!rc && (hdr.frame_type != 0 || hdr.width != g_width || hdr.height != g_height)) rc = VP8P_UNSUP_BITSTREAM;
    if (!rc) rc = vp8_parser_decode_mbs_compact(g_parsers[worker], g_maps[slot].mbx, g_maps[slot].blocks, g_maps[slot].cap,
                                                &g_maps[slot].nblocks, g_maps[slot].mvs, NULL);
    if (rc) { g_failed = 1; return; }
    *g_maps[slot].hdr = hdr;
}

/* --device-entropy: the header on the host, the rest of the frame handed over as it is (data_off was set by the main thread) */
static void export_one(void *arg, int i, int worker)
{
    const batch_ref *br = (const batch_ref *)arg;
    const frame *f = FRAME_AT(run_index(br->first + i));
    vp8hip_entropy_frame *e = &g_ent[ESET(br->b)][i];
    const uint64_t off = e->data_off;
    vp8ir_frame_hdr hdr;
    int rc = vp8_parser_begin_frame(g_parsers[worker], f->data, f->size, &hdr);
    if (!rc && (hdr.frame_type != 0 || hdr.width != g_width || hdr.height != g_height)) rc = VP8P_UNSUP_BITSTREAM;
    if (!rc) rc = vp8_parser_export_entropy(g_parsers[worker], e);
    if (rc) { g_failed = 1; return; }
    e->data_off = off;
    memcpy(g_ent_data[ESET(br->b)] + off, f->data, f->size);
}
/* where the frames of a batch go in the set's data buffer; returns the bytes in all */
static size_t place_frames(const batch_ref *br)
{
    size_t off = 0;
    for (int i = 0; i < br->n; i++) {
        g_ent[ESET(br->b)][i].data_off = off;
        off += FRAME_AT(run_index(br->first + i))->size;
    }
    return off;
}

static void hash_one(void *arg, int i, int worker)
{
    const batch_ref *br = (const batch_ref *)arg;
    const uint8_t *fb = g_host[br->b & 1] + (size_t)i * g_stride;
    md5_state md5;
    (void)worker;
    md5_init(&md5);
    for (int plane = 0; plane < 3; plane++) {
        const int w = plane ? (g_width + 1) >> 1 : g_width, rows = plane ? (g_height + 1) >> 1 : g_height;
        const int stride = plane ? g_geom.uv_stride : g_geom.y_stride;
        const uint8_t *p = fb + (plane == 0 ? g_geom.y_off : plane == 1 ? g_geom.u_off : g_geom.v_off);
        for (int y = 0; y < rows; y++, p += stride) md5_update(&md5, p, (size_t)w);
    }
    md5_final(&md5, g_digest[run_index(br->first + i)]);
}
static void take_digests(const batch_ref *br)
{
    for (int i = 0; i < br->n; i++) memcpy(g_digest[run_index(br->first + i)], g_dig[br->b & 1] + 16 * (size_t)i, 16);
}
#define EXIT_POOL_STARVED 75         /* (pooled_attempts, below) */
/* the status words of an entropy launch (set `set`, n frames from run position `first` on), once its copy has landed */
static void take_status(int set, long first, int n)
{
    for (int i = 0; i < n; i++) {
        if (g_ent_status[set][i] & 2u) {
            /* the launch's frames were decoded from half an IR: nothing of this attempt is kept -- the supervising process runs the
               work again with twice the pool (pooled_attempts) */
            fprintf(stderr, "frame %ld found the block pool (%zu MB) empty\n", run_index(first + i) + 1, g_pool_bytes >> 20);
            fflush(NULL);
            _exit(EXIT_POOL_STARVED);
        }
        if (g_ent_status[set][i] & 1u) {
            if (g_corrupt++ < 8) fprintf(stderr, "frame %ld: a partition ended early (corrupt)\n", run_index(first + i) + 1);
        }
    }
}

/* --entropy-batch: the block pool is sized from the stream (or by --pool-mb), and a launch that finds it empty has decoded some of
   its frames from half an IR.  The work therefore runs in a CHILD process (forked before anything touches the device); a child that
   ends with EXIT_POOL_STARVED is run again with twice the pool, up to three times.  Returns in the child, with the factor its pool
   is to be scaled by; the parent never returns. */
static size_t pooled_attempts(void)
{
    size_t scale = 1;
    for (int attempt = 0; attempt < 4; attempt++, scale *= 2) {
        fflush(NULL);
        const pid_t pid = fork();
        if (pid < 0) { fprintf(stderr, "fork failed\n"); exit(EXIT_FAILURE); }
        if (pid == 0) return scale;
        int st = 0;
        if (waitpid(pid, &st, 0) < 0) { fprintf(stderr, "waitpid failed\n"); exit(EXIT_FAILURE); }
        if (WIFEXITED(st) && WEXITSTATUS(st) == EXIT_POOL_STARVED && attempt < 3) {
            fprintf(stderr, "the block pool was too small: decoding again with a pool %zu times the size\n", 2 * scale);
            continue;
        }
        if (WIFEXITED(st) && WEXITSTATUS(st) == EXIT_POOL_STARVED)
            fprintf(stderr, "the block pool is still too small at %zu times its size: raise --pool-mb or lower --entropy-batch\n", scale);
        exit(WIFEXITED(st) ? WEXITSTATUS(st) : EXIT_FAILURE);
    }
    exit(EXIT_FAILURE);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(EXIT_FAILURE); } while (0)
#define HIP(call) do { if (call) DIE("%s: %s", #call, vp8hip_last_error(g_hip)); } while (0)


/* ---- --gpus: workers ------------------------------------------------------------------------------------------------------- */
static const char *g_stat_path;                     /* a worker leaves "frames seconds" here for the parent */
static void leave_stats(long frames, double seconds)
{
    if (!g_stat_path) return;
    FILE *f = fopen(g_stat_path, "w");
    if (f) { fprintf(f, "%ld %.6f\n", frames, seconds); fclose(f); }
}
/* this process's CPUs: share `g` of `G` of what it may run on (contiguous slices: the feeder threads of a device's worker stay
   together and off the other workers' cores) */
static void pin_share(int g, int G)
{
    cpu_set_t all, mine;
    if (getenv("VP8BATCH_NO_AFFINITY") || sched_getaffinity(0, sizeof all, &all)) return;
    const int n = CPU_COUNT(&all), per = n / G;
    if (per < 1) return;
    CPU_ZERO(&mine);
    int seen = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &all)) { if (seen >= g * per && seen < (g + 1) * per) CPU_SET(c, &mine); seen++; }
    (void)sched_setaffinity(0, sizeof mine, &mine);
}
/* Fork G workers over [0, total) (frames or streams).  In a worker: returns its index and sets *lo / *hi and VP8HIP_DEVICE; in the
   parent: waits, merges the workers' listings into `out_path`, prints the aggregate, and exits. */
static int fork_workers(int G, long total, long *lo, long *hi, const char *out_path, char *part_path, size_t part_len, const char *unit,
                        long pixels_per_unit_frame)
{
    pid_t pid[64];
    if (G > 64) G = 64;
    if (G > total) G = (int)total;
    const double t0 = now_s();
    for (int g = 0; g < G; g++) {
        fflush(NULL);
        pid[g] = fork();
        if (pid[g] < 0) DIE("fork failed");
        if (pid[g] == 0) {
            char dev[16];
            snprintf(dev, sizeof dev, "%d", getenv("VP8BATCH_SINGLE_DEVICE") ? 0 : g);
            setenv("VP8HIP_DEVICE", dev, 1);
            pin_share(g, G);
            const long base = total / G, extra = total % G;
            *lo = g * base + (g < extra ? g : extra);
            *hi = *lo + base + (g < extra ? 1 : 0);
            snprintf(part_path, part_len, "%s.part%d", out_path, g);
            static char stat[4096];
            snprintf(stat, sizeof stat, "%s.stat%d", out_path, g);
            g_stat_path = stat;
            return g;
        }
    }
    int bad = 0;
    for (int g = 0; g < G; g++) {
        int st = 0;
        if (waitpid(pid[g], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st)) bad = 1;
    }
    const double wall = now_s() - t0;
    if (bad) DIE("a worker failed");
    FILE *out = fopen(out_path, "wb");
    if (!out) DIE("Failed to open %s for writing", out_path);
    long frames = 0; double slowest = 0;
    for (int g = 0; g < G; g++) {
        char path[4096], buf[65536];
        snprintf(path, sizeof path, "%s.part%d", out_path, g);
        FILE *in = fopen(path, "rb");
        if (!in) DIE("worker %d left no listing", g);
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, in)) > 0) fwrite(buf, 1, n, out);
        fclose(in); remove(path);
        snprintf(path, sizeof path, "%s.stat%d", out_path, g);
        in = fopen(path, "r");
        long f = 0; double sec = 0;
        if (in) { if (fscanf(in, "%ld %lf", &f, &sec) != 2) f = 0; fclose(in); remove(path); }
        fprintf(stderr, "  device %d: %ld frames in %.3f s: %.1f frames/s\n", getenv("VP8BATCH_SINGLE_DEVICE") ? 0 : g, f, sec, sec > 0 ? f / sec : 0.0);
        frames += f; if (sec > slowest) slowest = sec;
    }
    fclose(out);
    fprintf(stderr, "%ld frames on %d GPUs (%s sharded in contiguous blocks): %.1f frames/s, %.1f Mpix/s over the slowest worker's %.3f s (%.3f s wall with start-up)\n",
            frames, G, unit, slowest > 0 ? frames / slowest : 0.0, slowest > 0 ? frames / slowest * pixels_per_unit_frame / 1e6 : 0.0, slowest, wall);
    exit(EXIT_SUCCESS);
}

/* ---- --streams: S streams side by side, entropy decode on the device -------------------------------------------------------- */
typedef struct input { frame *frames; int nframes; } input;
static struct {
    input *in; int nin; int S; long s0;               /* inputs; streams of this worker: global streams s0 .. s0 + S - 1 */
    int t;                                              /* position being parsed */
    vp8_parser **parser; vp8_refs *refs; vp8ir_frame_hdr *hdr; int *have_dims;
    vp8hip_entropy_frame *ent[2]; uint8_t *arena[2]; size_t *off;
} st;
static void stream_header_one(void *arg, int s, int worker)
{
    (void)arg; (void)worker;
    const input *in = &st.in[(st.s0 + s) % st.nin];
    const frame *f = &in->frames[st.t];
    vp8hip_entropy_frame *e = &st.ent[st.t & 1][s];
    if (vp8_refs_get_free(&st.refs[s]) < 0) { g_failed = 1; return; }
    int rc = vp8_parser_begin_frame(st.parser[s], f->data, f->size, &st.hdr[s]);
    if (!rc && (st.hdr[s].width != g_width || st.hdr[s].height != g_height)) rc = VP8P_UNSUP_BITSTREAM;
    if (!rc && !st.have_dims[s]) { st.have_dims[s] = 1; vp8_refs_on_alloc(&st.refs[s]); }
    if (!rc) rc = vp8_parser_export_entropy(st.parser[s], e);
    if (rc) { fprintf(stderr, "stream %ld frame %d: %s\n", st.s0 + s, st.t + 1, vp8_parser_error(st.parser[s])); g_failed = 1; return; }
    e->data_off = st.off[s];
    memcpy(st.arena[st.t & 1] + st.off[s], f->data, f->size);
}
static int run_streams(int S_total, int threads, int gpus, int argc, char **argv, int a)
{
    const int nin = argc - a - 1;
    const char *out_path = argv[argc - 1];
    if (nin < 1 || S_total < 1) DIE("--streams S <in.ivf> [<in2.ivf> ...] <out.md5>");
    input *in = calloc((size_t)nin, sizeof *in);
    int T = -1;
    for (int k = 0; k < nin; k++) {
        ivf_reader rd; const uint8_t *data; size_t size; int rc, cap = 0;
        if (ivf_open(&rd, argv[a + k])) DIE("%s is not an IVF file.", argv[a + k]);
        while ((rc = ivf_next(&rd, &data, &size)) == 1) {
            if (in[k].nframes == cap) { cap = cap ? 2 * cap : 64; in[k].frames = realloc(in[k].frames, sizeof(frame) * (size_t)cap); }
            if (in[k].nframes == 0) {
                int key, w, h;
                if (vp8_parser_peek(data, size, &key, &w, &h) || !key) DIE("%s does not start with a key frame", argv[a + k]);
                if (k == 0) { g_width = w; g_height = h; }
                else if (w != g_width || h != g_height) DIE("%s: all streams must have one frame size", argv[a + k]);
            }
            in[k].frames[in[k].nframes].data = malloc(size + 16);
            memcpy(in[k].frames[in[k].nframes].data, data, size);
            in[k].frames[in[k].nframes].size = size;
            in[k].nframes++;
        }
        ivf_close(&rd);
        if (rc < 0 || !in[k].nframes) DIE("failed to read %s", argv[a + k]);
        if (T < 0 || in[k].nframes < T) T = in[k].nframes;
    }
    long lo = 0, hi = S_total;
    char part[4096] = "";
    if (gpus > 1) { fork_workers(gpus, S_total, &lo, &hi, out_path, part, sizeof part, "streams", (long)g_width * g_height); out_path = part; }
    const int S = (int)(hi - lo);
    int device = -1;
    if (getenv("VP8HIP_DEVICE")) device = atoi(getenv("VP8HIP_DEVICE"));
    if (vp8hip_create(device, &g_hip)) DIE("vp8hip_create: %s (no CPU fallback)", vp8hip_last_error(NULL));
    HIP(vp8hip_configure(g_hip, g_width, g_height, 4 * S, S));
    st.in = in; st.nin = nin; st.S = S; st.s0 = lo;
    st.parser = calloc((size_t)S, sizeof *st.parser); st.refs = calloc((size_t)S, sizeof *st.refs);
    st.hdr = calloc((size_t)S, sizeof *st.hdr); st.have_dims = calloc((size_t)S, sizeof(int)); st.off = calloc((size_t)S + 1, sizeof(size_t));
    size_t arena_cap = 0;
    for (int t = 0; t < T; t++) {
        size_t sum = 0;
        for (int s = 0; s < S; s++) sum += in[(lo + s) % nin].frames[t].size;
        if (sum > arena_cap) arena_cap = sum;
    }
    uint32_t *status[2]; uint8_t *dig[2]; int *list[2];
    for (int k = 0; k < 2; k++) {
        st.ent[k] = vp8hip_host_alloc(g_hip, (size_t)S * sizeof(vp8hip_entropy_frame));
        st.arena[k] = vp8hip_host_alloc(g_hip, arena_cap + 16);
        status[k] = vp8hip_host_alloc(g_hip, (size_t)S * sizeof(uint32_t));
        dig[k] = vp8hip_host_alloc(g_hip, (size_t)S * 16);
        list[k] = malloc(sizeof(int) * (size_t)S);
        if (!st.ent[k] || !st.arena[k] || !status[k] || !dig[k]) DIE("vp8hip_host_alloc: %s", vp8hip_last_error(g_hip));
    }
    for (int s = 0; s < S; s++) {
        if (!(st.parser[s] = vp8_parser_create())) DIE("out of memory");
        vp8_parser_set_device_segmap(st.parser[s], 1);
        vp8_refs_init(&st.refs[s]);
    }
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    for (int t = 0; t < threads; t++) pthread_create(&tid[t], NULL, worker_main, (void *)(size_t)t);
    unsigned char (*digest)[16] = calloc((size_t)S * (size_t)T, 16);     /* [stream][shown frame] */
    int *nshown = calloc((size_t)S, sizeof(int));
    vp8hip_job *jobs = calloc((size_t)S, sizeof *jobs);
    long frames = 0, corrupt = 0;
    task parse_t;
    int prev_n = -1, prev_t = -1;                       /* the fetch in flight: shown frames, its position */
    int *prev_list = NULL;
#define PLACE(t_) do { size_t o_ = 0; for (int s_ = 0; s_ < S; s_++) { st.off[s_] = o_; o_ += in[(lo + s_) % nin].frames[t_].size; } st.off[S] = o_; } while (0)
#define TAKE_PREV() do { if (prev_n >= 0) { HIP(vp8hip_download_wait(g_hip));                                             \
        for (int i_ = 0; i_ < prev_n; i_++) { const int s_ = prev_list[i_] / 4; memcpy(digest[(size_t)s_ * T + nshown[s_]++], dig[prev_t & 1] + 16 * (size_t)i_, 16); } \
        for (int s_ = 0; s_ < S; s_++) if (status[prev_t & 1][s_] & 1u) { if (corrupt++ < 8) fprintf(stderr, "stream %ld frame %d: a partition ended early (corrupt)\n", lo + s_, prev_t + 1); } \
        prev_n = -1; } } while (0)
    const double t0 = now_s();
    st.t = 0; PLACE(0);
    size_t bytes = st.off[S];
    task_start(&parse_t, 0, stream_header_one, NULL, S);
    for (int t = 0; t < T; t++) {
        task_wait(&parse_t, 0);
        if (g_failed) DIE("a frame header of position %d failed to parse", t + 1);
        HIP(vp8hip_entropy_decode(g_hip, 0, S, st.ent[t & 1], st.arena[t & 1], bytes));
        if (t == 0) HIP(vp8hip_reserve(g_hip, S > 512, 1));       /* (the frame buffers' pools, while the first launch -- key frames: the long one -- runs) */
        /* (status and digests of position t - 1 are taken below, before their page-locked sets come round again at t + 1) */
        for (int s = 0; s < S; s++) {
            const vp8_refs *r = &st.refs[s];
            jobs[s].ir_slot = s; jobs[s].dst_fb = 4 * s + r->new_idx;
            jobs[s].ref_fb[0] = -1; jobs[s].ref_fb[1] = 4 * s + r->lst_idx; jobs[s].ref_fb[2] = 4 * s + r->gld_idx; jobs[s].ref_fb[3] = 4 * s + r->alt_idx;
        }
        HIP(vp8hip_decode(g_hip, jobs, S, VP8HIP_STAGE_ALL));
        TAKE_PREV();
        HIP(vp8hip_entropy_status_async(g_hip, S, status[t & 1]));
        int n = 0;
        for (int s = 0; s < S; s++) {
            vp8_refs_swap(&st.refs[s], &st.hdr[s]);
            if (st.hdr[s].show_frame) list[t & 1][n++] = 4 * s + st.refs[s].show_idx;
        }
        frames += S;
        if (n) HIP(vp8hip_frames_md5_list_async(g_hip, list[t & 1], n, dig[t & 1]));
        else HIP(vp8hip_sync(g_hip));
        prev_n = n; prev_t = t; prev_list = list[t & 1];
        if (t + 1 < T) {                                 /* the next position's headers while the device works on this one */
            st.t = t + 1; PLACE(t + 1); bytes = st.off[S];
            task_start(&parse_t, 0, stream_header_one, NULL, S);
        }
    }
    if (prev_n == 0) { prev_n = -1; for (int s_ = 0; s_ < S; s_++) if (status[prev_t & 1][s_] & 1u) corrupt++; }
    TAKE_PREV();
    HIP(vp8hip_sync(g_hip));
    const double dt = now_s() - t0;
    FILE *out = fopen(out_path, "wb");
    if (!out) DIE("Failed to open %s for writing", out_path);
    for (int s = 0; s < S; s++)
        for (int k = 0; k < nshown[s]; k++) {
            for (int i = 0; i < 16; i++) fprintf(out, "%02x", digest[(size_t)s * T + k][i]);
            if (S_total > 1) fprintf(out, "  stream%ld/img-%dx%d-%04d.i420\n", lo + s, g_width, g_height, k + 1);
            else fprintf(out, "  img-%dx%d-%04d.i420\n", g_width, g_height, k + 1);
        }
    fclose(out);
    fprintf(stderr, "%ld frames in %.3f s: %.1f frames/s, %.1f Mpix/s (%d streams of %d frames side by side, %d feeder threads, entropy decode on the device, MD5 on the device; %ld corrupt)\n",
            frames, dt, frames / dt, frames / dt * g_width * g_height / 1e6, S, T, threads, corrupt);
    leave_stats(frames, dt);
    pthread_mutex_lock(&pool.mu);
    pool.stop = 1;
    pthread_cond_broadcast(&pool.work);
    pthread_mutex_unlock(&pool.mu);
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    for (int s = 0; s < S; s++) vp8_parser_destroy(st.parser[s]);
    vp8hip_destroy(g_hip);
    return EXIT_SUCCESS;
}

int main(int argc, char **argv)
{
    int threads = 0, loop = 1, a = 1, host_md5 = 0, no_download = 0, streams = 0, gpus = 1;
    long pool_mb = 0;
    g_batch = 128;
    for (; a < argc && argv[a][0] == '-' && argv[a][1] == '-'; a++) {
        if (!strcmp(argv[a], "--threads") && a + 1 < argc) threads = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--host-md5")) host_md5 = 1;          /* hash on the host whatever the frame size */
        else if (!strcmp(argv[a], "--device-entropy")) g_dev_entropy = 1;
        else if (!strcmp(argv[a], "--no-download")) no_download = 1;
        else if (!strcmp(argv[a], "--entropy-batch") && a + 1 < argc) g_ebatch = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--entropy-dense")) ;                 /* (what --entropy-batch does anyway since the slots hold the compact form) */
        else if (!strcmp(argv[a], "--pool-mb") && a + 1 < argc) pool_mb = atol(argv[++a]);
        else if (!strcmp(argv[a], "--streams") && a + 1 < argc) streams = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--batch") && a + 1 < argc) g_batch = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--loop") && a + 1 < argc) loop = atoi(argv[++a]);
        else DIE("Usage: %s [--threads T] [--batch B] [--loop N] [--host-md5] [--device-entropy [--entropy-batch E [--pool-mb M]] [--no-download]] <in.ivf> <out.md5>", argv[0]);
    }
    if (threads < 1) {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        threads = n > 33 ? 32 : (n > 2 ? (int)n - 1 : 1);      /* more than ~32 feeders gain nothing: the host memory system is the limit */
        if (gpus > 1 && threads > 2 * gpus) threads = threads / gpus > 2 ? threads / gpus : 2;      /* a pool per device */
    }
    if (gpus < 1) gpus = 1;
    if (streams > 0) return run_streams(streams, threads, gpus, argc, argv, a);
    if (argc - a != 2 || g_batch < 1 || loop < 1)
        DIE("Usage: %s [--threads T] [--batch B] [--loop N] [--host-md5] [--device-entropy [--entropy-batch E [--pool-mb M]] [--no-download]] [--gpus G] <in.ivf> <out.md5>\n"
            "       %s --streams S [--threads T] [--gpus G] <in.ivf> [<in2.ivf> ...] <out.md5>", argv[0], argv[0]);

    /* ---- read the whole stream; every frame must be a key frame of one size */
    ivf_reader in;
    const uint8_t *data; size_t size; int rc, cap = 0;
    if (ivf_open(&in, argv[a])) DIE("%s is not an IVF file.", argv[a]);
    while ((rc = ivf_next(&in, &data, &size)) == 1) {
        int key, w, h;
        if (g_nframes == cap) { cap = cap ? 2 * cap : 64; g_frames = (frame *)realloc(g_frames, sizeof(frame) * cap); }
        if (vp8_parser_peek(data, size, &key, &w, &h) || !key)
            DIE("frame %d is not a key frame: batch_md5 needs independently decodable frames (use decode_to_md5)", g_nframes + 1);
        if (g_nframes == 0) { g_width = w; g_height = h; }
        else if (w != g_width || h != g_height) DIE("frame %d changes the frame size (use decode_to_md5)", g_nframes + 1);
        g_frames[g_nframes].data = (uint8_t *)malloc(size + 16);
        memcpy(g_frames[g_nframes].data, data, size);
        g_frames[g_nframes].size = size;
        g_nframes++;
    }
    ivf_close(&in);
    if (rc < 0 || !g_nframes) DIE("failed to read %s", argv[a]);
    long total = (long)g_nframes * loop;
    const char *out_path = argv[a + 1];
    char part[4096] = "";
    if (gpus > 1) {
        long lo = 0, hi = total;
        fork_workers(gpus, total, &lo, &hi, out_path, part, sizeof part, "frames", (long)g_width * g_height);
        g_first = lo; total = hi - lo; out_path = part;
    }
    if (g_batch > total) g_batch = (int)total;
    if (g_ebatch && (!g_dev_entropy || g_ebatch < g_batch || g_ebatch % g_batch)) DIE("--entropy-batch goes with --device-entropy and is a multiple of --batch");
    if (g_ebatch == g_batch) g_ebatch = 0;
    const int unit = g_ebatch ? g_ebatch : g_batch;            /* frames per entropy launch */

    /* ---- device and host state */
    const size_t pool_scale = g_ebatch ? pooled_attempts() : 1;
    int device = -1;
    if (getenv("VP8HIP_DEVICE")) device = atoi(getenv("VP8HIP_DEVICE"));
    if (vp8hip_create(device, &g_hip)) DIE("vp8hip_create: %s (no CPU fallback)", vp8hip_last_error(NULL));
    g_dev_md5 = !host_md5;
    if (g_ebatch && !g_dev_md5) DIE("--entropy-batch needs the MD5s computed on the device (no --host-md5)");
    if (no_download && !g_dev_md5) DIE("--no-download needs the MD5s computed on the device (no --host-md5)");
    /* slots and frame buffers: three sets for the host feeder (parsed / on the GPU / coming back); with the entropy decoder on
       the device the IR is written and read on one stream, one set does, and the frame buffers alternate between two */
    const int slot_sets = g_dev_entropy ? 1 : 3, fb_sets = g_dev_entropy ? 2 : 3;
    if (!g_ebatch) {
        HIP(vp8hip_configure(g_hip, g_width, g_height, fb_sets * g_batch, slot_sets * g_batch));
        HIP(vp8hip_geometry(g_hip, &g_geom));
    }
    if (g_dev_entropy) {
        {   /* a batch's bytes at most: the frames are taken in order of size within windows of SORT_WINDOW batches */
            g_order = (long *)malloc(sizeof(long) * (size_t)total);
            for (long k = 0; k < total; k++) g_order[k] = k;
            /* (whole launches: a launch that straddles two windows would get the smallest frames of one and the largest of the next) */
            long window = ((long)SORT_WINDOW * g_batch + unit - 1) / unit * unit;
            if (window < 4L * unit) window = 4L * unit;
            for (long w0 = 0; w0 < total; w0 += window) {
                const long wn = total - w0 < window ? total - w0 : window;
                qsort(g_order + w0, (size_t)wn, sizeof(long), by_size_desc);
            }
            for (long k0 = 0; k0 < total; k0 += unit) {
                size_t run = 0;
                for (long k = k0; k < k0 + unit && k < total; k++) run += FRAME_AT(g_order[k])->size;
                if (run > g_ent_cap) g_ent_cap = run;
            }
        }
        if (g_ebatch) {
            /* E frames per entropy launch: E slots without block streams of their own and ONE pool for the blocks of a launch
               (vp8hip_configure_pooled) -- a frame in flight costs what it needs (records 192 bytes a macroblock + its blocks), not
               the worst case (960), and frames in flight are what the entropy decoder's rate is made of.  The pool: what the
               launch that needs most is expected to need -- a frame's blocks are 6 to 13 times its compressed bytes in the
               fixtures: 14 times, capped by the worst case; / 0.75, because the kernel leaves a chunk as soon as what is left of it
               would not hold a macroblock row's WORST case (a quarter of a chunk; real rows take a third of that), + two chunks per
               frame for the chunks a frame begins and ends in. */
            const size_t nmb = (size_t)((g_width + 15) / 16) * (size_t)((g_height + 15) / 16);
            const size_t worst = nmb * 24 * 32, chunk = (size_t)4 * ((g_width + 15) / 16) * 24 * 32;
            size_t pool = 0;
            for (long k0 = 0; k0 < total; k0 += unit) {
                size_t need = 0;
                for (long k = k0; k < k0 + unit && k < total; k++) {
                    const size_t est = 14 * FRAME_AT(g_order[k])->size;
                    need += (est < worst ? est : worst) * 4 / 3 + 2 * chunk;
                }
                if (need > pool) pool = need;
            }
            pool += 4 * chunk;
            {   /* (a pool counts its blocks in 32 bits: 137 GB at most) */
                const size_t most = ((size_t)0xffffffffu / (chunk / 32) - 1) * chunk;
                if (pool > most) pool = most;
            }
            if (pool_mb > 0) pool = (size_t)pool_mb << 20;
            pool *= pool_scale;                      /* (an attempt after one that found the pool empty: pooled_attempts) */
            {
                const size_t most = ((size_t)0xffffffffu / (chunk / 32) - 1) * chunk;
                if (pool > most) pool = most;
            }
            g_pool_bytes = pool;
            /* frames that stay on the device are hashed a launch at a time (E frame buffers, tiles only); downloads go B at a time */
            HIP(vp8hip_configure_pooled(g_hip, g_width, g_height, no_download ? g_ebatch : 2 * g_batch, g_ebatch, pool));
            HIP(vp8hip_geometry(g_hip, &g_geom));
        }
        for (int k = 0; k < 3; k++) {
            if (!(g_ent[k] = (vp8hip_entropy_frame *)vp8hip_host_alloc(g_hip, (size_t)unit * sizeof(vp8hip_entropy_frame))) ||
                !(g_ent_data[k] = (uint8_t *)vp8hip_host_alloc(g_hip, g_ent_cap + 16)) ||
                !(g_ent_status[k] = (uint32_t *)vp8hip_host_alloc(g_hip, (size_t)unit * sizeof(uint32_t)))) DIE("vp8hip_host_alloc: %s", vp8hip_last_error(g_hip));
        }
    } else {
        g_maps = calloc((size_t)3 * g_batch, sizeof *g_maps);
        for (int s = 0; s < 3 * g_batch; s++)
            HIP(vp8hip_ir_map_compact(g_hip, s, &g_maps[s].hdr, &g_maps[s].mbx, &g_maps[s].blocks, &g_maps[s].cap, &g_maps[s].mvs));
    }
    g_stride = vp8hip_frame_stride(g_hip);
    g_packed = g_dev_md5 && !no_download && g_width % 8 == 0 && !getenv("VP8BATCH_WHOLE_BUFFERS");
    if (g_packed) g_stride = vp8hip_i420_bytes(g_hip);
    for (int k = 0; k < 2; k++) {
        if (!no_download && !(g_host[k] = (uint8_t *)vp8hip_host_alloc(g_hip, (size_t)g_batch * g_stride))) DIE("vp8hip_host_alloc: %s", vp8hip_last_error(g_hip));
        if (!(g_dig[k] = (uint8_t *)vp8hip_host_alloc(g_hip, (size_t)(g_ebatch && no_download ? g_ebatch : g_batch) * 16))) DIE("vp8hip_host_alloc: %s", vp8hip_last_error(g_hip));
    }
    g_digest = calloc((size_t)total, 16);
    g_parsers = calloc((size_t)threads, sizeof *g_parsers);
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    for (int t = 0; t < threads; t++) {
        if (!(g_parsers[t] = vp8_parser_create())) DIE("out of memory");
        pthread_create(&tid[t], NULL, worker_main, (void *)(size_t)t);
    }
    vp8hip_job *jobs = calloc((size_t)g_batch, sizeof *jobs);

    if (g_ebatch) {
        /* ---- the pipeline with the entropy decoder on the device in launches of up to E frames, the pixel path B at a time:
           headers of launch L+1 on the host while launch L is on the GPU; per part of a launch: decode, digests back. */
#define LAUNCH_FRAMES(first, n_out) do { (n_out) = (int)(total - (first) < g_ebatch ? total - (first) : g_ebatch); } while (0)
        task parse_t;
        batch_ref cur = { 0, 0, 0 }, prev = { -1, 0, 0 };
        long part_no = 0, L = 0, prev_launch = -1;
        /* the status words of a launch are good once a fetch queued behind their copy has come back: any part of that launch */
        struct { int valid, n; long first, launch; } pend[3] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#define TAKE_PENDING() do { for (int k_ = 0; k_ < 3; k_++) if (pend[k_].valid && pend[k_].launch <= prev_launch) {          \
                                take_status(k_, pend[k_].first, pend[k_].n); pend[k_].valid = 0; } } while (0)
        const double t0 = now_s();
        LAUNCH_FRAMES(0, cur.n);
        size_t bytes = place_frames(&cur), next_bytes = 0;
        task_start(&parse_t, 0, export_one, &cur, cur.n);
        batch_ref prev_launch_ref = { -1, 0, 0 };
        const int late_stage = getenv("VP8BATCH_LATE_STAGE") != NULL;      /* (experiments: a launch's input is sent when the launch is queued) */
        const int trace = getenv("VP8BATCH_TRACE") != NULL;          /* where the main thread's time goes, launch by launch */
        for (long done = 0; done < total; L++) {
            if (L == 0 || late_stage) {           /* (every later launch's input was sent while the launch before it was being queued: below) */
                task_wait(&parse_t, 0);
                if (g_failed) DIE("a frame of launch %ld failed to parse", L);
                HIP(vp8hip_entropy_stage(g_hip, cur.n, g_ent[ESET(cur.b)], g_ent_data[ESET(cur.b)], L == 0 ? bytes : next_bytes));
            }
            if (trace) fprintf(stderr, "launch %ld: at %.3f s\n", L, now_s() - t0);
            const batch_ref now = cur;
            done += now.n;
            const int have_next = done < total;
            if (have_next) {
                /* the headers of the next launch, at once: its page-locked set (one of three) belonged to launch L - 2, whose digests
                   came back an iteration ago -- the feeder threads work while this thread queues launch L and the device runs it */
                const long first = cur.first + cur.n;
                cur.b = cur.b + 1; cur.first = first;
                LAUNCH_FRAMES(first, cur.n);
                next_bytes = place_frames(&cur);
                task_start(&parse_t, 0, export_one, &cur, cur.n);
            }
            HIP(vp8hip_pool_reset(g_hip));               /* (on the stream: behind the pixel path of the launch before) */
            HIP(vp8hip_entropy_decode(g_hip, 0, now.n, NULL, NULL, 0));      /* (the kernel over the input staged before) */
            HIP(vp8hip_entropy_status_async(g_hip, now.n, g_ent_status[ESET(now.b)]));
            if (L == 0) HIP(vp8hip_reserve(g_hip, 1, !no_download && !g_packed));     /* (the frame buffers' pools, while the first launch runs) */
            pend[ESET(now.b)].valid = 1; pend[ESET(now.b)].n = now.n; pend[ESET(now.b)].first = now.first; pend[ESET(now.b)].launch = L;
            for (int at = 0; at < now.n; at += g_batch, part_no++) {
                const batch_ref part = { (int)(part_no & 0x3fffffff), now.n - at < g_batch ? now.n - at : g_batch, now.first + at };
                const int fb0 = no_download ? at : (part.b & 1) * g_batch;
                for (int i = 0; i < part.n; i++) {
                    jobs[i].ir_slot = at + i; jobs[i].dst_fb = fb0 + i;
                    jobs[i].ref_fb[0] = jobs[i].ref_fb[1] = jobs[i].ref_fb[2] = jobs[i].ref_fb[3] = -1;
                }
                const double td0 = now_s();
                HIP(vp8hip_decode(g_hip, jobs, part.n, VP8HIP_STAGE_ALL));
                if (trace) fprintf(stderr, "launch %ld: vp8hip_decode of part %d took the host %.3f s\n", L, at / g_batch, now_s() - td0);
                if (no_download) continue;
                if (prev.b >= 0) {
                    HIP(vp8hip_download_wait(g_hip));
                    take_digests(&prev);
                    TAKE_PENDING();
                }
                HIP((g_packed ? vp8hip_frames_fetch_i420_async : vp8hip_frames_fetch_async)(g_hip, fb0, part.n, g_host[part.b & 1], g_dig[part.b & 1]));
                prev = part; prev_launch = L;
            }
            if (have_next && !late_stage) {
                /* the next launch's input on its way now -- checked, and copied on the entropy decoder's copy stream while the
                   device is busy with this launch; its kernel is launched at the top of the next iteration */
                const double tw0 = now_s();
                task_wait(&parse_t, 0);
                if (g_failed) DIE("a frame of launch %ld failed to parse", L + 1);
                HIP(vp8hip_entropy_stage(g_hip, cur.n, g_ent[ESET(cur.b)], g_ent_data[ESET(cur.b)], next_bytes));
                if (trace) fprintf(stderr, "launch %ld: waited %.3f s for the next launch's headers, %.2f GB staged\n", L, now_s() - tw0, next_bytes / 1e9);
            }
            if (no_download) {
                /* the frames stay: ONE hash launch over the launch's frames, on the download stream -- beside the entropy decoder's
                   next launch, which is most of the time.  The launch before has come back by now: its page-locked set is free
                   for the headers of the next one */
                if (prev_launch_ref.b >= 0) {
                    const double tw1 = now_s();
                    HIP(vp8hip_download_wait(g_hip));
                    if (trace) fprintf(stderr, "launch %ld: queued at %.3f s, waited %.3f s for the digests of the launch before\n", L, tw1 - t0, now_s() - tw1);
                    take_digests(&prev_launch_ref);
                    prev_launch = L - 1;
                    TAKE_PENDING();
                }
                HIP(vp8hip_frames_fetch_async(g_hip, 0, now.n, NULL, g_dig[now.b & 1]));
                prev_launch_ref = now; prev = now; prev_launch = L;
            }
        }
        HIP(vp8hip_download_wait(g_hip));
        take_digests(&prev);
        TAKE_PENDING();
        const double dt = now_s() - t0;
        FILE *out = fopen(out_path, "wb");
        if (!out) DIE("Failed to open %s for writing", out_path);
        for (long f = 0; f < total; f++) {
            for (int i = 0; i < 16; i++) fprintf(out, "%02x", g_digest[f][i]);
            fprintf(out, "  img-%dx%d-%04ld.i420\n", g_width, g_height, g_first + f + 1);
        }
        fclose(out);
        fprintf(stderr, "%ld frames in %.3f s: %.1f frames/s, %.1f Mpix/s (%d feeder threads, %d frames per launch, entropy decode on the %s, MD5 on the %s%s; %d frames per entropy launch; %ld corrupt)\n",
                total, dt, total / dt, total / dt * g_width * g_height / 1e6, threads, g_batch, "device", "device",
                no_download ? ", frames not downloaded" : "", g_ebatch, g_corrupt);
        leave_stats(total, dt);
        pthread_mutex_lock(&pool.mu);
        pool.stop = 1;
        pthread_cond_broadcast(&pool.work);
        pthread_mutex_unlock(&pool.mu);
        for (int t = 0; t < threads; t++) { pthread_join(tid[t], NULL); vp8_parser_destroy(g_parsers[t]); }
        vp8hip_destroy(g_hip);
        return EXIT_SUCCESS;
    }

    /* ---- the pipeline */
    const long nbatch = (total + g_batch - 1) / g_batch;
    task parse_t, hash_t;
    batch_ref cur = { 0, (int)(total < g_batch ? total : g_batch), 0 }, nxt, prev = { -1, 0, 0 }, hashing = { -1, 0, 0 };
    const task_fn feed = g_dev_entropy ? export_one : parse_one;
    size_t ent_bytes = 0;
    const double t0 = now_s();
    if (g_dev_entropy) ent_bytes = place_frames(&cur);
    task_start(&parse_t, 0, feed, &cur, cur.n);
    for (long b = 0; b < nbatch; b++) {
        task_wait(&parse_t, 0);
        if (g_failed) DIE("a frame of batch %ld failed to parse", b);
        const batch_ref now = cur;
        if (b + 1 < nbatch && !g_dev_entropy) {
            /* the feeder goes on with the next batch at once -- slot set (b+1)%3 was last used by batch b-2, which came back an
               iteration ago -- while this thread downloads batch b-1 and uploads and launches batch b */
            const long first = cur.first + cur.n;
            nxt.b = cur.b + 1; nxt.first = first; nxt.n = (int)(total - first < g_batch ? total - first : g_batch);
            cur = nxt;
            task_start(&parse_t, 0, feed, &cur, cur.n);
        }
        const int fb0 = g_dev_entropy ? (now.b & 1) * g_batch : (now.b % 3) * g_batch;
        if (g_dev_entropy) {
            HIP(vp8hip_entropy_decode(g_hip, 0, now.n, g_ent[ESET(now.b)], g_ent_data[ESET(now.b)], ent_bytes));
            HIP(vp8hip_entropy_status_async(g_hip, now.n, g_ent_status[ESET(now.b)]));
        }
        for (int i = 0; i < now.n; i++) {
            const int s = g_dev_entropy ? i : (now.b % 3) * g_batch + i;
            if (!g_dev_entropy) HIP(vp8hip_ir_upload_compact(g_hip, s, g_maps[s].nblocks));
            jobs[i].ir_slot = s; jobs[i].dst_fb = fb0 + i;
            jobs[i].ref_fb[0] = jobs[i].ref_fb[1] = jobs[i].ref_fb[2] = jobs[i].ref_fb[3] = -1;
        }
        HIP(vp8hip_decode(g_hip, jobs, now.n, VP8HIP_STAGE_ALL));
        if (prev.b >= 0) {
            HIP(vp8hip_download_wait(g_hip));                               /* batch b-1 is in host set (b-1)&1 */
            if (g_dev_entropy) take_status(ESET(prev.b), prev.first, prev.n); /* (its status copy was queued in front of its fetch) */
            if (g_dev_md5) take_digests(&prev);
            else {
                if (hashing.b >= 0) task_wait(&hash_t, 1);                  /* batch b-2 hashed: host set b&1 is free again */
                hashing = prev;
                task_start(&hash_t, 1, hash_one, &hashing, hashing.n);
            }
        }
        if (b + 1 < nbatch && g_dev_entropy) {
            /* the headers of the next batch (little work).  The pinned set it is written to was handed to the GPU two batches
               ago, and that batch's digests have just come back: its copies are done */
            const long first = cur.first + cur.n;
            nxt.b = cur.b + 1; nxt.first = first; nxt.n = (int)(total - first < g_batch ? total - first : g_batch);
            cur = nxt;
            ent_bytes = place_frames(&cur);
            task_start(&parse_t, 0, feed, &cur, cur.n);
        }
        HIP((g_packed ? vp8hip_frames_fetch_i420_async : vp8hip_frames_fetch_async)(g_hip, fb0, now.n, no_download ? NULL : g_host[now.b & 1],
                                                                                     g_dev_md5 ? g_dig[now.b & 1] : NULL));
        prev = now;
    }
    HIP(vp8hip_download_wait(g_hip));
    if (g_dev_entropy) take_status(ESET(prev.b), prev.first, prev.n);
    if (g_dev_md5) take_digests(&prev);
    else {
        if (hashing.b >= 0) task_wait(&hash_t, 1);
        hashing = prev;
        task_start(&hash_t, 1, hash_one, &hashing, hashing.n);
        task_wait(&hash_t, 1);
    }
    const double dt = now_s() - t0;

    /* ---- the listing */
    FILE *out = fopen(out_path, "wb");
    if (!out) DIE("Failed to open %s for writing", out_path);
    for (long f = 0; f < total; f++) {
        for (int i = 0; i < 16; i++) fprintf(out, "%02x", g_digest[f][i]);
        fprintf(out, "  img-%dx%d-%04ld.i420\n", g_width, g_height, g_first + f + 1);
    }
    fclose(out);
    fprintf(stderr, "%ld frames in %.3f s: %.1f frames/s, %.1f Mpix/s (%d feeder threads, %d frames per launch, entropy decode on the %s, MD5 on the %s%s%s)\n",
            total, dt, total / dt, total / dt * g_width * g_height / 1e6, threads, g_batch, g_dev_entropy ? "device" : "host",
            g_dev_md5 ? "device" : "host", no_download ? ", frames not downloaded" : "", g_corrupt ? "; CORRUPT FRAMES, see above" : "");
    leave_stats(total, dt);

    pthread_mutex_lock(&pool.mu);
    pool.stop = 1;
    pthread_cond_broadcast(&pool.work);
    pthread_mutex_unlock(&pool.mu);
    for (int t = 0; t < threads; t++) { pthread_join(tid[t], NULL); vp8_parser_destroy(g_parsers[t]); }
    if (g_host[0]) vp8hip_host_free(g_hip, g_host[0]);
    if (g_host[1]) vp8hip_host_free(g_hip, g_host[1]);
    for (int k = 0; k < 2; k++) if (g_ent[k]) { vp8hip_host_free(g_hip, g_ent[k]); vp8hip_host_free(g_hip, g_ent_data[k]); }
    vp8hip_host_free(g_hip, g_dig[0]);
    vp8hip_host_free(g_hip, g_dig[1]);
    vp8hip_destroy(g_hip);
    return EXIT_SUCCESS;
}
