/* VP8 bitstream-format constants used by the host feeder (see vp8_tables.c, generated). */
#ifndef VP8_TABLES_H
#define VP8_TABLES_H
#include <stdint.h>
extern const uint8_t  vp8t_coef_update_probs[1056];   /* RFC 6386 13.4  */
extern const uint8_t  vp8t_default_coef_probs[1056];  /* RFC 6386 13.5  */
extern const uint8_t  vp8t_kf_bmode_probs[900];       /* RFC 6386 11.5  */
extern const uint8_t  vp8t_default_mv_context[38];    /* RFC 6386 17.2  */
extern const uint8_t  vp8t_mv_update_probs[38];
extern const int32_t  vp8t_mode_contexts[24];         /* RFC 6386 16.3 (vp8_mode_contexts[6][4]) */
extern const uint16_t vp8t_dc_qlookup[128];           /* RFC 6386 14.1  */
extern const uint16_t vp8t_ac_qlookup[128];
extern const int16_t vp8t_pp_rv[440];       /* dither table of the demacroblocking post-filter (postproc.c:80-130) */
#endif
