/* WebM (Matroska) input for the command line tools: the frames of the first VP8 video track, in file order.  The reference's
 * vpxdec reads WebM through its bundled nestegg (vpxdec.c:443-571, file type probing :573-587); this is a reader for the subset
 * a VP8 elementary stream needs: EBML header, Segment, Tracks/TrackEntry (number, type, codec, pixel size), Cluster,
 * SimpleBlock and BlockGroup/Block without lacing.  Elements of unknown size (live muxing) are followed to the end of file. */
#ifndef VP8HIP_WEBM_H
#define VP8HIP_WEBM_H
#include <stddef.h>
#include <stdint.h>

typedef struct webm_reader {
    uint8_t *data;          /* the whole file */
    size_t size, pos;       /* pos: next element inside the Segment */
    unsigned track, width, height;
    char codec[32];
} webm_reader;

/* 0 = ok; -1 = cannot open / read; -2 = not a WebM file or no VP8 video track */
int  webm_open(webm_reader *r, const char *path);
/* 1 = *data,*size is the next frame (points into the file image); 0 = end of stream; -1 = damaged or laced block */
int  webm_next(webm_reader *r, const uint8_t **data, size_t *size);
void webm_close(webm_reader *r);
#endif
