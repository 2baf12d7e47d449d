/* IVF container reader (32-byte "DKIF" file header, 12-byte frame headers: LE32 size + LE64 pts),
 * the input format of vpxdec (vpxdec.c:386-441, :224-318) and decode_to_md5 (decoder_tmpl.c:47-103). */
#ifndef VP8HIP_IVF_H
#define VP8HIP_IVF_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct ivf_reader {
    FILE *f;
    unsigned fourcc, width, height, rate, scale, nframes;
    uint8_t *buf;
    size_t cap;
} ivf_reader;

static inline unsigned ivf_le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
static inline unsigned ivf_le16(const uint8_t *p) { return p[0] | (p[1] << 8); }

static inline int ivf_open(ivf_reader *r, const char *path)
{
    uint8_t h[32];
    memset(r, 0, sizeof *r);
    r->f = strcmp(path, "-") ? fopen(path, "rb") : stdin;
    if (!r->f) return -1;
    if (fread(h, 1, 32, r->f) != 32 || memcmp(h, "DKIF", 4)) return -2;
    r->fourcc = ivf_le32(h + 8);
    r->width = ivf_le16(h + 12);
    r->height = ivf_le16(h + 14);
    r->rate = ivf_le32(h + 16);
    r->scale = ivf_le32(h + 20);
    r->nframes = ivf_le32(h + 24);
    return 0;
}

/* returns 1 and sets *data,*size on success, 0 at end of file, -1 on a truncated frame */
static inline int ivf_next(ivf_reader *r, const uint8_t **data, size_t *size)
{
    uint8_t h[12];
    size_t sz;
    if (fread(h, 1, 12, r->f) != 12) return 0;
    sz = ivf_le32(h);
    if (sz > r->cap) {
        uint8_t *nb = (uint8_t *)realloc(r->buf, sz + 16);
        if (!nb) return -1;
        r->buf = nb;
        r->cap = sz;
    }
    if (fread(r->buf, 1, sz, r->f) != sz) return -1;
    *data = r->buf;
    *size = sz;
    return 1;
}

static inline void ivf_close(ivf_reader *r)
{
    if (r->f && r->f != stdin) fclose(r->f);
    free(r->buf);
    memset(r, 0, sizeof *r);
}
#endif
