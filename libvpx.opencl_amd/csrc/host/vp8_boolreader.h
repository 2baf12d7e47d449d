/* VP8 boolean entropy decoder (RFC 6386 section 7) -- host feeder, CPU only.
 *
 * Behavioural reference: vp8/decoder/dboolhuff.h:76-120 (vp8dx_decode_bool) and
 * dboolhuff.c:16-60.  Same arithmetic (split = 1 + (((range-1)*prob) >> 8), MSB-first
 * window, zero bytes past the end of the partition); own structure: a 64-bit window with
 * `bits` counting the valid bits BELOW the active top byte.
 */
#ifndef VP8_BOOLREADER_H
#define VP8_BOOLREADER_H

#include <stddef.h>
#include <stdint.h>

typedef struct vp8_boolreader {
    const uint8_t *cur, *end;
    uint64_t window;   /* active byte in bits 63..56 */
    int      bits;     /* valid bits below the top byte; < 0 => refill before next read */
    uint32_t range;    /* 128..255 after normalisation */
    int      zero_fill; /* bytes synthesised past the end (stream over-read indicator) */
} vp8_boolreader;

static inline void vp8br_refill(vp8_boolreader *br)
{
    /* the bulk of a partition: up to seven bytes with one load (cf. the reference's VP8DX_BOOL_DECODER_FILL, dboolhuff.h:51-76,
       which also shifts in as many whole bytes as fit) */
    if (br->end - br->cur >= 8 && br->bits <= 0) {
        uint64_t v;
        int take = (56 - br->bits) >> 3;            /* whole bytes that fit below the valid bits: 7 at bits <= 0 */
        __builtin_memcpy(&v, br->cur, 8);
        v = __builtin_bswap64(v) >> (64 - 8 * take);      /* the first `take` bytes, first byte on top */
        br->window |= v << (56 - br->bits - 8 * take);
        br->cur += take;
        br->bits += 8 * take;
        return;
    }
    while (br->bits <= 48) {
        uint64_t byte = 0;
        if (br->cur < br->end)
            byte = *br->cur++;
        else
            br->zero_fill++;
        br->window |= byte << (48 - br->bits);
        br->bits += 8;
    }
}

static inline void vp8br_init(vp8_boolreader *br, const uint8_t *data, size_t size)
{
    br->cur = data;
    br->end = data + size;
    br->window = 0;
    br->bits = -8;
    br->range = 255;
    br->zero_fill = 0;
    vp8br_refill(br);
}

static inline int vp8br_get(vp8_boolreader *br, int prob)
{
    uint32_t split = 1 + (((br->range - 1) * (uint32_t)prob) >> 8);
    uint64_t big;
    int bit, shift;
    if (br->bits < 0)
        vp8br_refill(br);
    big = (uint64_t)split << 56;
    if (br->window >= big) {
        br->window -= big;
        br->range -= split;
        bit = 1;
    } else {
        br->range = split;
        bit = 0;
    }
    shift = __builtin_clz(br->range) - 24;   /* 0..7: renormalise range into 128..255 */
    br->range <<= shift;
    br->window <<= shift;
    br->bits -= shift;
    return bit;
}

static inline int vp8br_bit(vp8_boolreader *br) { return vp8br_get(br, 128); }

static inline int vp8br_literal(vp8_boolreader *br, int nbits)
{
    int v = 0;
    while (nbits-- > 0)
        v = (v << 1) | vp8br_get(br, 128);
    return v;
}

/* vp8dx_bool_error (vp8/decoder/dboolhuff.h:131-153): the decoder has shifted bits that were never in the partition into its
 * top byte.  The reference keeps `count` = buffered bits below the top byte and adds VP8_LOTS_OF_BITS once a fill finds the
 * partition exhausted; its test `count > VP8_BD_VALUE_SIZE && count < VP8_LOTS_OF_BITS` then says: fewer than zero REAL bits
 * below the top byte.  Here the zero bytes of the refill are counted, so that is bits - 8 * zero_fill < 0, once the end has
 * been met (a window that merely waits for its next refill is not an error). */
static inline int vp8br_error(const vp8_boolreader *br)
{
    return br->zero_fill > 0 && br->bits - 8 * br->zero_fill < 0;
}

#endif
