/* See webm.h. */
#include "webm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
    ID_EBML = 0x1A45DFA3, ID_SEGMENT = 0x18538067, ID_TRACKS = 0x1654AE6B, ID_TRACK_ENTRY = 0xAE, ID_TRACK_NUMBER = 0xD7,
    ID_TRACK_TYPE = 0x83, ID_CODEC_ID = 0x86, ID_VIDEO = 0xE0, ID_PIXEL_WIDTH = 0xB0, ID_PIXEL_HEIGHT = 0xBA,
    ID_CLUSTER = 0x1F43B675, ID_SIMPLE_BLOCK = 0xA3, ID_BLOCK_GROUP = 0xA0, ID_BLOCK = 0xA1
};
#define UNKNOWN_SIZE ((uint64_t)-1)

/* An element header at *pos: the ID keeps its length marker, the size loses it (all ones = unknown).  0 on success. */
static int read_header(const uint8_t *d, size_t end, size_t *pos, uint32_t *id, uint64_t *size)
{
    size_t p = *pos;
    int n, i;
    uint64_t v;
    if (p >= end) return -1;
    for (n = 1; n <= 4 && !(d[p] & (0x100 >> n)); n++) { }
    if (n > 4 || p + n > end) return -1;
    for (*id = 0, i = 0; i < n; i++) *id = (*id << 8) | d[p + i];
    p += n;
    if (p >= end) return -1;
    for (n = 1; n <= 8 && !(d[p] & (0x100 >> n)); n++) { }
    if (n > 8 || p + n > end) return -1;
    v = d[p] & (0xffu >> n);
    for (i = 1; i < n; i++) v = (v << 8) | d[p + i];
    *size = v == ((uint64_t)1 << (7 * n)) - 1 ? UNKNOWN_SIZE : v;
    *pos = p + n;
    return 0;
}

static uint64_t read_uint(const uint8_t *d, uint64_t n)
{
    uint64_t v = 0;
    while (n--) v = (v << 8) | *d++;
    return v;
}

/* one TrackEntry: takes it if it is the first video track coded as VP8 */
static void parse_track(webm_reader *r, size_t pos, size_t end)
{
    unsigned number = 0, type = 0, w = 0, h = 0;
    char codec[32] = "";
    uint32_t id;
    uint64_t sz;
    while (pos < end && !read_header(r->data, end, &pos, &id, &sz) && sz <= end - pos) {
        if (id == ID_TRACK_NUMBER) number = (unsigned)read_uint(r->data + pos, sz);
        else if (id == ID_TRACK_TYPE) type = (unsigned)read_uint(r->data + pos, sz);
        else if (id == ID_CODEC_ID) { const size_t n = sz < 31 ? (size_t)sz : 31; memcpy(codec, r->data + pos, n); codec[n] = 0; }
        else if (id == ID_VIDEO) {
            size_t q = pos;
            const size_t qe = pos + (size_t)sz;
            uint32_t vid;
            uint64_t vsz;
            while (q < qe && !read_header(r->data, qe, &q, &vid, &vsz) && vsz <= qe - q) {
                if (vid == ID_PIXEL_WIDTH) w = (unsigned)read_uint(r->data + q, vsz);
                else if (vid == ID_PIXEL_HEIGHT) h = (unsigned)read_uint(r->data + q, vsz);
                q += (size_t)vsz;
            }
        }
        pos += (size_t)sz;
    }
    if (!r->track && type == 1 && !strcmp(codec, "V_VP8")) {
        r->track = number; r->width = w; r->height = h;
        strcpy(r->codec, codec);
    }
}

int webm_open(webm_reader *r, const char *path)
{
    FILE *f = fopen(path, "rb");
    long n;
    size_t pos = 0, seg;
    uint32_t id;
    uint64_t sz;
    memset(r, 0, sizeof *r);
    if (!f) return -1;
    if (fseek(f, 0, SEEK_END) || (n = ftell(f)) < 0 || fseek(f, 0, SEEK_SET)) { fclose(f); return -1; }
    r->data = (uint8_t *)malloc((size_t)n + 1);
    if (!r->data || fread(r->data, 1, (size_t)n, f) != (size_t)n) { fclose(f); webm_close(r); return -1; }
    fclose(f);
    r->size = (size_t)n;
    if (read_header(r->data, r->size, &pos, &id, &sz) || id != ID_EBML || sz > r->size - pos) { webm_close(r); return -2; }
    pos += (size_t)sz;
    if (read_header(r->data, r->size, &pos, &id, &sz) || id != ID_SEGMENT) { webm_close(r); return -2; }
    if (sz != UNKNOWN_SIZE && sz < r->size - pos) r->size = pos + (size_t)sz;          /* the Segment bounds everything */
    seg = pos;
    while (pos < r->size && !read_header(r->data, r->size, &pos, &id, &sz)) {           /* top level of the Segment: find Tracks */
        if (id == ID_CLUSTER) break;
        if (sz == UNKNOWN_SIZE || sz > r->size - pos) break;
        if (id == ID_TRACKS) {
            size_t q = pos;
            const size_t qe = pos + (size_t)sz;
            uint32_t tid;
            uint64_t tsz;
            while (q < qe && !read_header(r->data, qe, &q, &tid, &tsz) && tsz <= qe - q) {
                if (tid == ID_TRACK_ENTRY) parse_track(r, q, q + (size_t)tsz);
                q += (size_t)tsz;
            }
        }
        pos += (size_t)sz;
    }
    if (!r->track) { webm_close(r); return -2; }
    r->pos = seg;
    return 0;
}

int webm_next(webm_reader *r, const uint8_t **data, size_t *size)
{
    uint32_t id;
    uint64_t sz;
    while (r->pos < r->size) {
        if (read_header(r->data, r->size, &r->pos, &id, &sz)) return -1;
        if (id == ID_CLUSTER || id == ID_BLOCK_GROUP) continue;             /* containers of blocks: descend */
        if (sz == UNKNOWN_SIZE || sz > r->size - r->pos) return -1;
        if (id == ID_SIMPLE_BLOCK || id == ID_BLOCK) {
            const uint8_t *b = r->data + r->pos;
            size_t hdr;
            uint64_t track;
            int n;
            r->pos += (size_t)sz;
            if (sz < 4) return -1;
            for (n = 1; n <= 8 && !(b[0] & (0x100 >> n)); n++) { }            /* track number: a size-style integer */
            if (n > 8 || (uint64_t)n + 3 > sz) return -1;
            track = b[0] & (0xffu >> n);
            for (hdr = 1; hdr < (size_t)n; hdr++) track = (track << 8) | b[hdr];
            if (track != r->track) continue;
            if (b[n + 2] & 0x06) return -1;                                  /* laced: not something a video muxer writes */
            *data = b + n + 3;                                               /* after the 16-bit timecode and the flags */
            *size = (size_t)sz - (size_t)n - 3;
            return 1;
        }
        r->pos += (size_t)sz;                                               /* anything else at these levels: skip */
    }
    return 0;
}

void webm_close(webm_reader *r)
{
    free(r->data);
    memset(r, 0, sizeof *r);
}
