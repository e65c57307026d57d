/* oracle/ -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CRC-32 (IEEE 802.3 reflected, poly 0xEDB88320), Adler-32 (RFC 1950) and crc32_combine,
 * the three checksum entry points the reference binds at zlib_ngmodule.c:1455-1596
 * (zng_crc32 / zng_adler32 / zng_crc32_combine call sites :1487, :1549, :1595, :1741). */
#include "oracle.h"

static uint32_t crc_table[256];
static int crc_table_ready = 0;

static void make_crc_table(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        crc_table[i] = c;
    }
    crc_table_ready = 1;
}

uint32_t za_o_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    if (!crc_table_ready) make_crc_table();
    uint32_t c = crc ^ 0xFFFFFFFFu;
    for (size_t i = 0; i < len; i++) c = crc_table[(c ^ buf[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

uint32_t za_o_adler32(uint32_t adler, const uint8_t *buf, size_t len)
{
    uint32_t a = (adler & 0xFFFF) % 65521u, b = ((adler >> 16) & 0xFFFF) % 65521u;
    while (len > 0) {
        size_t k = len < 5552 ? len : 5552;   /* largest n with 255n(n+1)/2 + (n+1)(65520) < 2^32 */
        len -= k;
        while (k--) { a += *buf++; b += a; }
        a %= 65521u; b %= 65521u;
    }
    return (b << 16) | a;
}

/* GF(2) polynomial arithmetic modulo the reflected CRC polynomial: a(x)*b(x) mod P(x). */
static uint32_t multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 0x80000000u, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? ((b >> 1) ^ 0xEDB88320u) : (b >> 1);
    }
    return p;
}

/* x^(8*len) mod P */
static uint32_t x8nmodp(uint64_t len)
{
    uint32_t p = 0x80000000u;          /* x^0 */
    uint32_t sq = 0x00800000u;         /* x^8  (bit 31 = x^0, so x^8 is bit 23) */
    while (len) {
        if (len & 1) p = multmodp(sq, p);
        sq = multmodp(sq, sq);
        len >>= 1;
    }
    return p;
}

uint32_t za_o_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    return multmodp(x8nmodp(len2), crc1) ^ crc2;
}
