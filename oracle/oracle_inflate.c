/* oracle/ -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * RFC 1951 inflate + RFC 1952 gzip member reader + RFC 1950 zlib container, restating what the
 * reference reaches through zng_inflate (call sites zlib_ngmodule.c:328, :2539) and the member
 * state machine GzipReader_read_into_buffer (zlib_ngmodule.c:2426-2637).
 * Decoder: canonical-Huffman decode with a 10-bit lookup table and a bit-serial slow path for
 * longer codes.  Error rules follow the published zlib behaviour the reference's tests rely on
 * (over-subscribed / incomplete sets, missing end-of-block, distance too far back, ...). */
#include "oracle.h"
#include <string.h>

typedef struct {
    const uint8_t *in; size_t in_len; size_t pos;   /* next byte to load */
    uint64_t bitbuf; int bitcnt;
    int overrun;                                    /* tried to read past the input */
} bitrd;

static inline void refill(bitrd *b)
{
    while (b->bitcnt <= 56 && b->pos < b->in_len) {
        b->bitbuf |= (uint64_t)b->in[b->pos++] << b->bitcnt;
        b->bitcnt += 8;
    }
}
static inline uint32_t getbits(bitrd *b, int n)
{
    if (b->bitcnt < n) { refill(b); if (b->bitcnt < n) { b->overrun = 1; return 0; } }
    uint32_t v = (uint32_t)(b->bitbuf & ((1ull << n) - 1));
    b->bitbuf >>= n; b->bitcnt -= n;
    return v;
}

#define LUT_BITS 10
typedef struct {
    uint16_t count[16];      /* number of codes of each length */
    uint16_t symbol[288];    /* symbols ordered by (length, symbol) */
    uint16_t lut[1 << LUT_BITS]; /* (sym<<4)|len, 0 = not resolvable within LUT_BITS */
    int maxlen;
} hufftab;

/* returns 0 complete, <0 over-subscribed, >0 incomplete */
static int build(hufftab *h, const uint8_t *lens, int n)
{
    int left = 1;
    uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[lens[i]]++;
    h->maxlen = 0;
    for (int l = 1; l <= 15; l++) if (h->count[l]) h->maxlen = l;
    if (h->count[0] == n) { memset(h->lut, 0, sizeof h->lut); return 0; } /* no codes */
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= h->count[l]; if (left < 0) return left; }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h->count[l];
    for (int i = 0; i < n; i++) if (lens[i]) h->symbol[offs[lens[i]]++] = (uint16_t)i;
    /* LUT: walk canonical codes, store bit-reversed */
    memset(h->lut, 0, sizeof h->lut);
    {
        uint32_t code = 0; int idx = 0;
        for (int l = 1; l <= LUT_BITS && l <= 15; l++) {
            for (int k = 0; k < h->count[l]; k++, code++, idx++) {
                uint32_t rev = 0;
                for (int b = 0; b < l; b++) if (code & (1u << b)) rev |= 1u << (l - 1 - b);
                for (uint32_t e = rev; e < (1u << LUT_BITS); e += 1u << l)
                    h->lut[e] = (uint16_t)((h->symbol[idx] << 4) | l);
            }
            code <<= 1;
        }
    }
    return left;
}

/* decode one symbol; -1 = invalid code / out of input */
static inline int decode(bitrd *b, const hufftab *h)
{
    if (b->bitcnt < 15) refill(b);
    uint16_t e = h->lut[b->bitbuf & ((1u << LUT_BITS) - 1)];
    if (e) {
        int l = e & 15;
        if (l > b->bitcnt) { b->overrun = 1; return -1; }
        b->bitbuf >>= l; b->bitcnt -= l;
        return e >> 4;
    }
    /* slow path: bit-serial canonical decode */
    int code = 0, first = 0, index = 0;
    uint64_t bb = b->bitbuf;
    for (int l = 1; l <= 15; l++) {
        if (l > b->bitcnt) { b->overrun = 1; return -1; }
        code |= (int)(bb & 1); bb >>= 1;
        int count = h->count[l];
        if (code - count < first) {
            b->bitbuf >>= l; b->bitcnt -= l;
            return h->symbol[index + (code - first)];
        }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

static const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t  len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t  dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

int za_o_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                     const uint8_t *dict, size_t dict_len,
                     size_t *in_used, size_t *out_len)
{
    bitrd b = { in, in_len, 0, 0, 0, 0 };
    size_t op = 0;
    int ret = ZA_OK;
    static hufftab fixl, fixd; static int fixed_ready = 0;
    hufftab dynl, dynd;
    if (dict_len > ZA_WIN) { dict += dict_len - ZA_WIN; dict_len = ZA_WIN; }
    if (!fixed_ready) {
        uint8_t l[288];
        int i = 0;
        for (; i < 144; i++) l[i] = 8;
        for (; i < 256; i++) l[i] = 9;
        for (; i < 280; i++) l[i] = 7;
        for (; i < 288; i++) l[i] = 8;
        build(&fixl, l, 288);
        for (i = 0; i < 30; i++) l[i] = 5;
        build(&fixd, l, 30);
        fixed_ready = 1;
    }
    for (;;) {
        int last = (int)getbits(&b, 1);
        int type = (int)getbits(&b, 2);
        if (b.overrun) { ret = ZA_BUF_ERROR; break; }
        if (type == 0) {
            int drop = b.bitcnt & 7;
            b.bitbuf >>= drop; b.bitcnt -= drop;
            uint32_t len = getbits(&b, 16), nlen = getbits(&b, 16);
            if (b.overrun) { ret = ZA_BUF_ERROR; break; }
            if ((len ^ 0xFFFF) != nlen) { ret = ZA_DATA_ERROR; break; }
            while (len) {
                if (b.bitcnt == 0) { refill(&b); if (b.bitcnt == 0) { ret = ZA_BUF_ERROR; goto done; } }
                if (op >= out_cap) { ret = ZA_BUF_ERROR; goto done; }
                out[op++] = (uint8_t)(b.bitbuf & 0xFF); b.bitbuf >>= 8; b.bitcnt -= 8; len--;
            }
        } else if (type == 3) {
            ret = ZA_DATA_ERROR; break;
        } else {
            const hufftab *hl = &fixl, *hd = &fixd;
            if (type == 2) {
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t lens[320];
                hufftab cl;
                int nlen = (int)getbits(&b, 5) + 257, ndist = (int)getbits(&b, 5) + 1, ncode = (int)getbits(&b, 4) + 4;
                if (b.overrun) { ret = ZA_BUF_ERROR; break; }
                if (nlen > 286 || ndist > 30) { ret = ZA_DATA_ERROR; break; }
                memset(lens, 0, 19);
                for (int i = 0; i < ncode; i++) lens[order[i]] = (uint8_t)getbits(&b, 3);
                if (b.overrun) { ret = ZA_BUF_ERROR; break; }
                if (build(&cl, lens, 19) != 0) { ret = ZA_DATA_ERROR; break; }   /* must be complete */
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = decode(&b, &cl);
                    if (sym < 0) { ret = b.overrun ? ZA_BUF_ERROR : ZA_DATA_ERROR; goto done; }
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) { ret = ZA_DATA_ERROR; goto done; }
                            val = lens[idx - 1]; rep = 3 + (int)getbits(&b, 2);
                        } else if (sym == 17) rep = 3 + (int)getbits(&b, 3);
                        else rep = 11 + (int)getbits(&b, 7);
                        if (b.overrun) { ret = ZA_BUF_ERROR; goto done; }
                        if (idx + rep > nlen + ndist) { ret = ZA_DATA_ERROR; goto done; }
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) { ret = ZA_DATA_ERROR; break; }       /* missing end-of-block */
                int e = build(&dynl, lens, nlen);
                if (e < 0 || (e > 0 && dynl.maxlen != 1)) { ret = ZA_DATA_ERROR; break; }
                e = build(&dynd, lens + nlen, ndist);
                if (e < 0 || (e > 0 && dynd.maxlen != 1)) { ret = ZA_DATA_ERROR; break; }
                hl = &dynl; hd = &dynd;
            }
            for (;;) {
                int sym = decode(&b, hl);
                if (sym < 0) { ret = b.overrun ? ZA_BUF_ERROR : ZA_DATA_ERROR; goto done; }
                if (sym < 256) {
                    if (op >= out_cap) { ret = ZA_BUF_ERROR; goto done; }
                    out[op++] = (uint8_t)sym;
                } else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) { ret = ZA_DATA_ERROR; goto done; }
                    uint32_t len = len_base[sym] + getbits(&b, len_extra[sym]);
                    int ds = decode(&b, hd);
                    if (ds < 0) { ret = b.overrun ? ZA_BUF_ERROR : ZA_DATA_ERROR; goto done; }
                    if (ds >= 30) { ret = ZA_DATA_ERROR; goto done; }
                    uint32_t dist = dist_base[ds] + getbits(&b, dist_extra[ds]);
                    if (b.overrun) { ret = ZA_BUF_ERROR; goto done; }
                    if (dist > op + dict_len) { ret = ZA_DATA_ERROR; goto done; }   /* too far back */
                    while (len--) {
                        if (op >= out_cap) { ret = ZA_BUF_ERROR; goto done; }
                        out[op] = (dist > op) ? dict[dict_len - (dist - op)] : out[op - dist];
                        op++;
                    }
                }
            }
        }
        if (last) { ret = ZA_STREAM_END; break; }
    }
done:
    /* give back whole unused bytes held in the bit buffer */
    {
        size_t unused = (size_t)(b.bitcnt >> 3);
        if (in_used) *in_used = b.pos - unused;
    }
    if (out_len) *out_len = op;
    return ret;
}

/* ---- gzip members: zlib_ngmodule.c:2443-2611 ---- */
#define FHCRC 2
#define FEXTRA 4
#define FNAME 8
#define FCOMMENT 16

int za_o_gunzip(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                size_t *out_len, int *n_members)
{
    size_t pos = 0, op = 0;
    int members = 0;
    if (out_len) *out_len = 0;
    if (n_members) *n_members = 0;
    for (;;) {
        /* HEADER (:2443-2532) */
        if (pos == in_len) break;                       /* clean EOF */
        if (in_len - pos < 10) return ZA_GZ_TRUNCATED;
        if (!(in[pos] == 0x1f && in[pos + 1] == 0x8b)) return ZA_GZ_BAD_MAGIC;
        if (in[pos + 2] != 8) return ZA_GZ_BAD_METHOD;
        int flags = in[pos + 3];
        size_t cur = pos + 10;
        if (flags & FEXTRA) {
            if (cur + 2 >= in_len) return ZA_GZ_TRUNCATED;
            size_t fl = in[cur] | (in[cur + 1] << 8);
            cur += 2;
            if (cur + fl >= in_len) return ZA_GZ_TRUNCATED;
            cur += fl;
        }
        if (flags & FNAME) {
            const uint8_t *z = memchr(in + cur, 0, in_len - cur);
            if (!z) return ZA_GZ_TRUNCATED;
            cur = (size_t)(z - in) + 1;
        }
        if (flags & FCOMMENT) {
            const uint8_t *z = memchr(in + cur, 0, in_len - cur);
            if (!z) return ZA_GZ_TRUNCATED;
            cur = (size_t)(z - in) + 1;
        }
        if (flags & FHCRC) {
            if (cur + 2 >= in_len) return ZA_GZ_TRUNCATED;
            uint16_t hc = (uint16_t)(in[cur] | (in[cur + 1] << 8));
            uint16_t c = (uint16_t)(za_o_crc32(0, in + pos, cur - pos) & 0xFFFF);
            if (hc != c) return ZA_GZ_BAD_HCRC;
            cur += 2;
        }
        /* DEFLATE (:2533-2576) */
        size_t used = 0, produced = 0;
        int r = za_o_inflate_raw(in + cur, in_len - cur, out + op, out_cap - op, NULL, 0, &used, &produced);
        if (r == ZA_BUF_ERROR) {
            if (out_len) *out_len = op + produced;
            return (op + produced >= out_cap) ? ZA_BUF_ERROR : ZA_GZ_TRUNCATED;
        }
        if (r != ZA_STREAM_END) return r;
        uint32_t crc = za_o_crc32(0, out + op, produced);
        cur += used;
        /* TRAILER (:2577-2599) */
        if (in_len - cur < 8) return ZA_GZ_TRUNCATED;
        uint32_t tc = in[cur] | (in[cur + 1] << 8) | (in[cur + 2] << 16) | ((uint32_t)in[cur + 3] << 24);
        uint32_t tl = in[cur + 4] | (in[cur + 5] << 8) | (in[cur + 6] << 16) | ((uint32_t)in[cur + 7] << 24);
        if (tc != crc) return ZA_GZ_BAD_CRC;
        if (tl != (uint32_t)(produced & 0xFFFFFFFFu)) return ZA_GZ_BAD_LENGTH;
        cur += 8;
        op += produced;
        members++;
        /* NULL_BYTES (:2601-2611) */
        while (cur < in_len && in[cur] == 0) cur++;
        pos = cur;
        if (out_len) *out_len = op;
        if (n_members) *n_members = members;
    }
    if (out_len) *out_len = op;
    if (n_members) *n_members = members;
    return ZA_OK;
}

int za_o_zlib_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len)
{
    if (out_len) *out_len = 0;
    if (in_len < 2) return ZA_BUF_ERROR;
    unsigned cmf = in[0], flg = in[1];
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0) return ZA_DATA_ERROR;
    if (flg & 0x20) return ZA_NEED_DICT;
    size_t used = 0, produced = 0;
    int r = za_o_inflate_raw(in + 2, in_len - 2, out, out_cap, NULL, 0, &used, &produced);
    if (out_len) *out_len = produced;
    if (r != ZA_STREAM_END) return r;
    size_t cur = 2 + used;
    if (in_len - cur < 4) return ZA_BUF_ERROR;
    uint32_t ad = ((uint32_t)in[cur] << 24) | (in[cur + 1] << 16) | (in[cur + 2] << 8) | in[cur + 3];
    if (ad != za_o_adler32(1, out, produced)) return ZA_DATA_ERROR;
    return ZA_OK;
}
