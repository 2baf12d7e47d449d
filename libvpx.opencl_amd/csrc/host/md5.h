/* MD5 (RFC 1321), own implementation; used by vpxdec --md5 and decode_to_md5. */
#ifndef VP8HIP_MD5_H
#define VP8HIP_MD5_H
#include <stddef.h>
#include <stdint.h>
typedef struct md5_state { uint32_t h[4]; uint64_t nbytes; unsigned char buf[64]; } md5_state;
void md5_init(md5_state *s);
void md5_update(md5_state *s, const void *data, size_t len);
void md5_final(md5_state *s, unsigned char digest[16]);
#endif
