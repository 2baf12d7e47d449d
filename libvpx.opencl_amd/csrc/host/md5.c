/* MD5 message digest, written from RFC 1321. */
#include "md5.h"
#include <string.h>

static const uint32_t K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8,
    0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340,
    0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87,
    0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
    0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039,
    0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92,
    0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb,
    0xeb86d391
};
static const unsigned char S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14,
                                     20, 5, 9, 14, 20, 5, 9, 14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11,
                                     16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };

static void block(md5_state *s, const unsigned char *p)
{
    uint32_t m[16], a = s->h[0], b = s->h[1], c = s->h[2], d = s->h[3];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    for (int i = 0; i < 64; i++) {
        uint32_t f, t;
        int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        t = d; d = c; c = b;
        f += a + K[i] + m[g];
        b += (f << S[i]) | (f >> (32 - S[i]));
        a = t;
    }
    s->h[0] += a; s->h[1] += b; s->h[2] += c; s->h[3] += d;
}

void md5_init(md5_state *s)
{
    s->h[0] = 0x67452301; s->h[1] = 0xefcdab89; s->h[2] = 0x98badcfe; s->h[3] = 0x10325476;
    s->nbytes = 0;
}

void md5_update(md5_state *s, const void *data, size_t len)
{
    const unsigned char *p = (const unsigned char *)data;
    size_t fill = (size_t)(s->nbytes & 63);
    s->nbytes += len;
    if (fill) {
        size_t n = 64 - fill < len ? 64 - fill : len;
        memcpy(s->buf + fill, p, n);
        p += n; len -= n;
        if (fill + n < 64) return;
        block(s, s->buf);
    }
    for (; len >= 64; p += 64, len -= 64) block(s, p);
    memcpy(s->buf, p, len);
}

void md5_final(md5_state *s, unsigned char digest[16])
{
    uint64_t bits = s->nbytes * 8;
    unsigned char pad[72] = { 0x80 };
    size_t fill = (size_t)(s->nbytes & 63), n = (fill < 56 ? 56 : 120) - fill;
    md5_update(s, pad, n);
    for (int i = 0; i < 8; i++) pad[i] = (unsigned char)(bits >> (8 * i));
    md5_update(s, pad, 8);
    for (int i = 0; i < 16; i++) digest[i] = (unsigned char)(s->h[i >> 2] >> (8 * (i & 3)));
}
