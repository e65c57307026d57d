// Streaming entry points of the C ABI: zngamd_stream_* with the calling convention and the return codes of the zng_*
// functions the reference's streaming objects are written against.  Product code, host side only
// (included by zng_amd.hip); every payload byte is compressed, decoded and checksummed by the kernels behind the batch entry
// points -- this file buffers, frames (RFC 1950 / 1952 headers and trailers) and keeps the resume state.
//
//   zngamd_stream_deflate*   replace zng_deflateInit2 / zng_deflate / zng_deflateSetDictionary / zng_deflateCopy / zng_deflateEnd
//                            as called by Compress.compress (zlib_ngmodule.c:530-575), Compress.flush (:718-784),
//                            Compress.copy (:795-850), compressobj (:376-430)
//   zngamd_stream_inflate*   replace zng_inflateInit2 / zng_inflate / zng_inflateSetDictionary / zng_inflateCopy / zng_inflateEnd
//                            as called by Decompress.decompress (:622-716), Decompress.flush (:959-1037), ZlibDecompressor (:1102-1195)
//
// A GPU wants large batches, a zng_stream caller feeds whatever it has: deflate collects input until a flush or 32 MiB and
// then runs one dictionary-chained engine batch that ends on a sync-flush boundary (so the pieces concatenate into one valid
// stream); inflate keeps the compressed bytes from the last block header on and decodes from there each call (bit offset +
// up to 32 KiB of history), handing out only what is new -- exactly what the Python objects did before, now behind the C ABI.
#include <memory>

// no C++ exception leaves the C ABI: running out of host memory is zlib's Z_MEM_ERROR
#define ZS_GUARD catch (const std::bad_alloc &) { return ZNGAMD_MEM_ERROR; } catch (...) { return ZNGAMD_STREAM_ERROR; }

#define ZS_NO_FLUSH 0
#define ZS_FULL_FLUSH 3
#define ZS_FINISH   4
#define ZS_BLOCK    5
#define ZS_BATCH    (32u << 20)

// byte vector whose resize() does not zero-fill (the engine writes into the new room straight away)
template <class T> struct ZsNoInit : std::allocator<T> {
    template <class U> struct rebind { using other = ZsNoInit<U>; };
    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
typedef std::vector<uint8_t, ZsNoInit<uint8_t>> ZsBytes;

struct ZsDeflate {
    int level = 6, kind = 1 /* 0 raw, 1 zlib, 2 gzip */, wb = 15;
    std::vector<uint8_t> pending, tail;      // pending = [pend_tail bytes of tail][collected input] (empty: nothing collected)
    size_t pend_tail = 0;
    bool started = false, finished = false, has_dict = false;
    uint32_t crc = 0, adler = 1, dictid = 0;
    uint64_t size = 0;
};
struct ZsInflate {
    int kind = 1 /* 0 raw, 1 zlib, 2 gzip, 3 auto */, wbits = 15;
    int kind0 = 1, wbits0 = 15;      // as given to inflate_init (auto-detection rewrites kind / wbits): what a reset goes back to
    std::vector<uint8_t> zdict, window, buf;
    uint32_t start_bit = 0;
    uint64_t skip = 0, total = 0;
    uint64_t ahead = 0;              // output the caller has announced it will take beyond avail_out (zngamd_stream_inflate_ahead)
    bool header_done = false, deflate_done = false, eof = false, want_dict = false;
    uint32_t check = 1;
};
struct zngamd_stream_state {
    zngamd_ctx *ctx = nullptr;
    bool is_deflate = false;
    ZsDeflate d;
    ZsInflate i;
    ZsBytes outq;                    // produced, not yet handed out
    size_t outpos = 0;
    std::string msg;
};

static int zs_msg(zngamd_stream *s, int code, const char *m)
{
    s->state->msg = m ? m : "";
    s->msg = m ? s->state->msg.c_str() : nullptr;
    return code;
}
// hand queued output to the caller; true if something moved
static bool zs_drain(zngamd_stream *s)
{
    zngamd_stream_state *st = s->state;
    const size_t have = st->outq.size() - st->outpos;
    const size_t k = std::min<size_t>(have, s->avail_out);
    if (k) {
        memcpy(s->next_out, st->outq.data() + st->outpos, k);
        s->next_out += k; s->avail_out -= (uint32_t)k; s->total_out += k; st->outpos += k;
    }
    if (st->outpos == st->outq.size()) { st->outq.clear(); st->outpos = 0; }
    return k != 0;
}
static void zs_put(zngamd_stream_state *st, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; st->outq.insert(st->outq.end(), b, b + n); }

// ---- deflate ---------------------------------------------------------------------------------------------------------
static void zs_zlib_header(zngamd_stream_state *st)
{
    const ZsDeflate &d = st->d;
    const int lv = d.level == -1 ? 6 : d.level;
    const unsigned flevel = lv < 2 ? 0 : lv < 6 ? 1 : lv == 6 ? 2 : 3;
    unsigned head = ((((unsigned)d.wb - 8u) << 4) | 8u) << 8 | (flevel << 6) | (d.has_dict ? 0x20u : 0u);
    head += 31 - head % 31;
    const uint8_t h[2] = {(uint8_t)(head >> 8), (uint8_t)head};
    zs_put(st, h, 2);
    if (d.has_dict) { const uint8_t id[4] = {(uint8_t)(d.dictid >> 24), (uint8_t)(d.dictid >> 16), (uint8_t)(d.dictid >> 8), (uint8_t)d.dictid}; zs_put(st, id, 4); }
}
static void zs_gzip_header(zngamd_stream_state *st)
{
    const int lv = st->d.level == -1 ? 6 : st->d.level;
    const uint8_t h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, (uint8_t)(lv == 9 ? 2 : lv == 1 ? 4 : 0), 3};
    zs_put(st, h, 10);
}
// one engine batch: `data` primed with the 32 KiB tail; appends the raw deflate bytes to the output queue
static int zs_deflate_batch(zngamd_stream *s, const uint8_t *data, size_t n, bool final)
{
    zngamd_stream_state *st = s->state;
    ZsDeflate &d = st->d;
    zngamd_ctx *c = st->ctx;
    // the engine's block length is a u32: pieces of at most 1 GiB, each primed by the 32 KiB in front of it
    size_t pos = 0;
    bool first = true;
    while (first || pos < n) {
        const size_t ln = std::min<size_t>(n - pos, 1u << 30);
        const bool last = pos + ln == n;
        // the first 128 KiB of a piece go through a small buffer behind the dictionary tail; the rest is compressed where it
        // lies in the caller's memory (every unit primed by the bytes in front of it)
        const size_t head = std::min<size_t>(ln, ZA_MAX_UNIT);
        std::vector<uint8_t> hb(d.tail.size() + head);
        if (!d.tail.empty()) memcpy(hb.data(), d.tail.data(), d.tail.size());
        if (head) memcpy(hb.data() + d.tail.size(), data + pos, head);
        const uint32_t wflag = ZNGAMD_FLAG_WBITS(d.wb);
        auto run = [&](const uint8_t *buf, size_t buf_len, size_t off, size_t len, size_t dict, bool fin) -> int {
            zngamd_block B; B.off = off; B.len = (uint32_t)len; B.dict_len = (uint32_t)dict; B.flags = wflag | (fin ? ZNGAMD_FLAG_FINAL : 0u); B.reserved = 0;
            const uint64_t cap = len + len / 8 + (len / ZA_MAX_UNIT + 2) * 64 + 64;
            const size_t at = st->outq.size();
            st->outq.resize(at + cap);
            uint32_t olen = 0, crc = 0;
            const int r = zngamd_deflate_blocks(c, buf, buf_len, &B, 1, d.level, st->outq.data() + at, cap, &olen, &crc);
            if (r != ZNGAMD_OK) { st->outq.resize(at); return r; }
            st->outq.resize(at + olen);
            d.crc = zngamd_crc32_combine(d.crc, crc, len);          // (from 0: the first piece's own value)
            d.size += len;
            return ZNGAMD_OK;
        };
        int r = run(hb.data(), hb.size(), d.tail.size(), head, d.tail.size(), final && last && head == ln);
        if (r) return r;
        if (ln > head) {
            r = run(data + pos, ln, head, ln - head, ZA_WIN, final && last);
            if (r) return r;
        }
        if (d.kind == 1 && ln) {
            uint32_t a = d.adler;
            r = zngamd_adler32(c, a, data + pos, ln, &a);
            if (r) return r;
            d.adler = a;
        }
        // the next piece is primed with the last 32 KiB of everything so far
        if (ln >= ZA_WIN) d.tail.assign(data + pos + ln - ZA_WIN, data + pos + ln);
        else {
            std::vector<uint8_t> t(d.tail);
            t.insert(t.end(), data + pos, data + pos + ln);
            if (t.size() > ZA_WIN) t.erase(t.begin(), t.end() - ZA_WIN);
            d.tail.swap(t);
        }
        pos += ln;
        first = false;
    }
    return ZNGAMD_OK;
}
// the collected input in ONE engine call: `pending` = [the 32 KiB tail][collected input], the block is primed from the tail
static int zs_deflate_pending(zngamd_stream *s, bool final)
{
    zngamd_stream_state *st = s->state;
    ZsDeflate &d = st->d;
    zngamd_ctx *c = st->ctx;
    if (d.pending.empty()) { d.pending = d.tail; d.pend_tail = d.tail.size(); }        // (final with nothing collected)
    const size_t tl = d.pend_tail, n = d.pending.size() - tl;
    if (n > (1u << 30)) return ZNGAMD_E_ARG;          // (cannot happen: collected input is emitted at 32 MiB, larger pieces go direct)
    zngamd_block B; B.off = tl; B.len = (uint32_t)n; B.dict_len = (uint32_t)tl; B.flags = ZNGAMD_FLAG_WBITS(d.wb) | (final ? ZNGAMD_FLAG_FINAL : 0u); B.reserved = 0;
    const uint64_t cap = n + n / 8 + (n / ZA_MAX_UNIT + 2) * 64 + 64;
    const size_t at = st->outq.size();
    st->outq.resize(at + cap);
    uint32_t olen = 0, crc = 0;
    int r = zngamd_deflate_blocks(c, d.pending.data(), d.pending.size(), &B, 1, d.level, st->outq.data() + at, cap, &olen, &crc);
    if (r != ZNGAMD_OK) { st->outq.resize(at); return r; }
    st->outq.resize(at + olen);
    d.crc = zngamd_crc32_combine(d.crc, crc, n);
    d.size += n;
    if (d.kind == 1 && n) {
        uint32_t a = d.adler;
        r = zngamd_adler32(c, a, d.pending.data() + tl, n, &a);
        if (r) return r;
        d.adler = a;
    }
    const size_t keep = std::min<size_t>(d.pending.size(), ZA_WIN);
    d.tail.assign(d.pending.end() - keep, d.pending.end());
    if (d.pending.capacity() > (4u << 20) && n < (1u << 20)) std::vector<uint8_t>().swap(d.pending);   // a small flush does not pin a large buffer
    d.pending.clear(); d.pend_tail = 0;
    return ZNGAMD_OK;
}
static int zs_deflate_emit(zngamd_stream *s, const uint8_t *direct, size_t direct_len, bool final)
{
    zngamd_stream_state *st = s->state;
    ZsDeflate &d = st->d;
    if (!d.started) {
        d.started = true;
        if (d.kind == 1) zs_zlib_header(st); else if (d.kind == 2) zs_gzip_header(st);
    }
    int r = ZNGAMD_OK;
    if (direct) { if (direct_len || final) r = zs_deflate_batch(s, direct, direct_len, final); }
    else if (!d.pending.empty() || final) r = zs_deflate_pending(s, final);
    if (r) return r;
    if (final) {
        d.finished = true;
        if (d.kind == 1) { const uint8_t t[4] = {(uint8_t)(d.adler >> 24), (uint8_t)(d.adler >> 16), (uint8_t)(d.adler >> 8), (uint8_t)d.adler}; zs_put(st, t, 4); }
        else if (d.kind == 2) {
            const uint32_t sz = (uint32_t)(d.size & 0xFFFFFFFFull);
            const uint8_t t[8] = {(uint8_t)d.crc, (uint8_t)(d.crc >> 8), (uint8_t)(d.crc >> 16), (uint8_t)(d.crc >> 24), (uint8_t)sz, (uint8_t)(sz >> 8), (uint8_t)(sz >> 16), (uint8_t)(sz >> 24)};
            zs_put(st, t, 8);
        }
    }
    return ZNGAMD_OK;
}

extern "C" {

int zngamd_stream_deflate_init(zngamd_ctx *c, zngamd_stream *s, int level, int method, int wbits, int mem_level, int strategy)
try {
    if (!c || !s) return ZNGAMD_STREAM_ERROR;
    s->state = nullptr; s->msg = nullptr; s->total_in = s->total_out = 0; s->adler = 1;
    if (level < -1 || level > 9 || method != 8 || mem_level < 1 || mem_level > 9 || strategy < 0 || strategy > 4) return ZNGAMD_STREAM_ERROR;
    int kind, wb;
    if (wbits >= 9 && wbits <= 15) { kind = 1; wb = wbits; }
    else if (wbits <= -9 && wbits >= -15) { kind = 0; wb = -wbits; }
    else if (wbits >= 25 && wbits <= 31) { kind = 2; wb = wbits - 16; }
    else if (wbits == 8) { kind = 1; wb = 9; }              // zlib promotes an 8-bit window to 9
    else if (wbits == 24) { kind = 2; wb = 9; }
    else return ZNGAMD_STREAM_ERROR;
    zngamd_stream_state *st = new zngamd_stream_state();
    st->ctx = c; st->is_deflate = true;
    st->d.level = level; st->d.kind = kind; st->d.wb = wb;
    s->state = st;
    s->adler = kind == 2 ? 0u : 1u;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_deflate_set_dictionary(zngamd_stream *s, const uint8_t *dict, uint32_t len)
try {
    if (!s || !s->state || !s->state->is_deflate || (!dict && len)) return ZNGAMD_STREAM_ERROR;
    ZsDeflate &d = s->state->d;
    // zng_deflateSetDictionary: before the first deflate call (nothing collected either), and never on a gzip stream -- no gzip
    // decoder could supply the dictionary (Z_STREAM_ERROR for wrap == 2; the reference then raises ValueError("Invalid dictionary"))
    if (d.started || d.kind == 2 || !d.pending.empty() || s->total_in != 0) return ZNGAMD_STREAM_ERROR;
    if (d.kind == 1) {
        uint32_t a = 1;
        const int r = zngamd_adler32(s->state->ctx, 1, dict, len, &a);
        if (r) return r;
        d.dictid = a; d.has_dict = true;
        s->adler = a;
    }
    const uint32_t keep = len > ZA_WIN ? (uint32_t)ZA_WIN : len;
    d.tail.assign(dict + (len - keep), dict + len);
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_deflate(zngamd_stream *s, int flush)
try {
    if (!s || !s->state || !s->state->is_deflate || flush < 0 || flush > ZS_BLOCK) return ZNGAMD_STREAM_ERROR;
    zngamd_stream_state *st = s->state;
    ZsDeflate &d = st->d;
    if ((!s->next_in && s->avail_in) || (!s->next_out && s->avail_out)) return ZNGAMD_STREAM_ERROR;
    if (d.finished && (flush != ZS_FINISH || s->avail_in)) return zs_msg(s, ZNGAMD_STREAM_ERROR, nullptr);
    const uint64_t in0 = s->avail_in, out0 = s->avail_out;
    int r = ZNGAMD_OK;
    if (!d.finished) {
        if (s->avail_in >= ZS_BATCH && !d.pending.empty()) {
            // a large piece behind a small one: what was collected goes first, as a batch of its own, so that the large piece is
            // compressed where it lies (and `pending`, whose length the engine takes as a u32, never holds more than 2 x 32 MiB)
            r = zs_deflate_emit(s, nullptr, 0, false);
        }
        if (r == ZNGAMD_OK && d.pending.empty() && s->avail_in >= ZS_BATCH) {
            // a large piece with nothing pending: compressed where it lies (no copy of the payload)
            r = zs_deflate_emit(s, s->next_in, s->avail_in, flush == ZS_FINISH);
            if (r == ZNGAMD_OK) { s->next_in += s->avail_in; s->total_in += s->avail_in; s->avail_in = 0; }
        } else if (r == ZNGAMD_OK) {
            if (s->avail_in) {
                if (d.pending.empty()) { d.pending = d.tail; d.pend_tail = d.tail.size(); }
                d.pending.insert(d.pending.end(), s->next_in, s->next_in + s->avail_in);
                s->next_in += s->avail_in; s->total_in += s->avail_in; s->avail_in = 0;
            }
            if (flush != ZS_NO_FLUSH || d.pending.size() - d.pend_tail >= ZS_BATCH) r = zs_deflate_emit(s, nullptr, 0, flush == ZS_FINISH);
        }
        if (r != ZNGAMD_OK) return zs_msg(s, r == ZNGAMD_E_HIP ? ZNGAMD_MEM_ERROR : r > 0 || r < -6 ? ZNGAMD_STREAM_ERROR : r, zngamd_last_error(st->ctx));
        // Z_FULL_FLUSH: decompression can restart at this point, so nothing behind it may refer to anything in front of it --
        // the history that would prime the next batch is forgotten (zng_deflate clears its hash table there)
        if (flush == ZS_FULL_FLUSH) { d.tail.clear(); d.pend_tail = 0; }
    }
    zs_drain(s);
    s->adler = d.kind == 1 ? d.adler : d.crc;        // zlib: Adler-32; gzip and raw: CRC-32 of what has been compressed so far
    if (d.finished && st->outq.empty()) return ZNGAMD_STREAM_END;
    if (in0 == s->avail_in && out0 == s->avail_out && in0 == 0 && flush == ZS_NO_FLUSH) return ZNGAMD_BUF_ERROR;     // no progress possible
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_pending(const zngamd_stream *s, uint64_t *pending)
try {
    if (!s || !s->state || !pending) return ZNGAMD_STREAM_ERROR;
    *pending = s->state->outq.size() - s->state->outpos;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_deflate_copy(zngamd_stream *dst, const zngamd_stream *src)
try {
    if (!dst || !src || !src->state || !src->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    *dst = *src;
    dst->state = new zngamd_stream_state(*src->state);
    dst->msg = nullptr;
    return ZNGAMD_OK;
} ZS_GUARD

// zng_deflateReset (zlib_ngmodule.c:1725): back to the state right behind deflate_init -- level, container and window kept,
// everything collected, the history, a preset dictionary and the checksums forgotten
int zngamd_stream_deflate_reset(zngamd_stream *s)
try {
    if (!s || !s->state || !s->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    zngamd_stream_state *st = s->state;
    ZsDeflate fresh;
    fresh.level = st->d.level; fresh.kind = st->d.kind; fresh.wb = st->d.wb;
    st->d = fresh;
    st->outq.clear(); st->outpos = 0; st->msg.clear();
    s->msg = nullptr; s->total_in = s->total_out = 0; s->adler = fresh.kind == 2 ? 0u : 1u;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_deflate_end(zngamd_stream *s)
try {
    if (!s || !s->state || !s->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    const bool busy = !s->state->d.finished && (s->state->d.started || !s->state->d.pending.empty());
    delete s->state;
    s->state = nullptr; s->msg = nullptr;
    return busy ? ZNGAMD_DATA_ERROR : ZNGAMD_OK;      // zng_deflateEnd: Z_DATA_ERROR when the stream was freed prematurely
} ZS_GUARD

// ---- inflate ---------------------------------------------------------------------------------------------------------
int zngamd_stream_inflate_init(zngamd_ctx *c, zngamd_stream *s, int wbits)
try {
    if (!c || !s) return ZNGAMD_STREAM_ERROR;
    s->state = nullptr; s->msg = nullptr; s->total_in = s->total_out = 0; s->adler = 1;
    int kind;
    if (wbits == 0 || (wbits >= 8 && wbits <= 15)) kind = 1;
    else if (wbits <= -8 && wbits >= -15) kind = 0;
    else if ((wbits >= 24 && wbits <= 31) || wbits == 16) kind = 2;
    else if ((wbits >= 40 && wbits <= 47) || wbits == 32) kind = 3;
    else return ZNGAMD_STREAM_ERROR;
    zngamd_stream_state *st = new zngamd_stream_state();
    st->ctx = c; st->is_deflate = false;
    st->i.kind = st->i.kind0 = kind; st->i.wbits = st->i.wbits0 = wbits; st->i.header_done = kind == 0;
    s->state = st;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_inflate_set_dictionary(zngamd_stream *s, const uint8_t *dict, uint32_t len)
try {
    if (!s || !s->state || s->state->is_deflate || (!dict && len)) return ZNGAMD_STREAM_ERROR;
    ZsInflate &I = s->state->i;
    if (I.kind == 0) {                      // raw stream: the history the first block may refer to; any time before decoding starts
        if (I.total || I.deflate_done) return ZNGAMD_STREAM_ERROR;
        const uint32_t keep = len > ZA_WIN ? (uint32_t)ZA_WIN : len;
        I.window.assign(dict + (len - keep), dict + len);
        return ZNGAMD_OK;
    }
    if (!I.want_dict) return ZNGAMD_STREAM_ERROR;          // zlib stream: only right after Z_NEED_DICT
    uint32_t a = 1;
    const int r = zngamd_adler32(s->state->ctx, 1, dict, len, &a);
    if (r) return r;
    if (a != s->adler) return ZNGAMD_DATA_ERROR;           // not the dictionary the stream was written with
    const uint32_t keep = len > ZA_WIN ? (uint32_t)ZA_WIN : len;
    I.window.assign(dict + (len - keep), dict + len);
    I.want_dict = false;
    I.buf.erase(I.buf.begin(), I.buf.begin() + 6);
    I.header_done = true; I.check = 1;
    return ZNGAMD_OK;
} ZS_GUARD

// gzip member header at buf[0..): 1 = complete (*start = first deflate byte), 0 = more input needed, < 0 = error (msg set)
static int zs_gzip_header_parse(zngamd_stream *s, const std::vector<uint8_t> &b, size_t *start)
{
    const size_t n = b.size();
    if (n >= 2 && !(b[0] == 0x1f && b[1] == 0x8b)) return zs_msg(s, ZNGAMD_DATA_ERROR, "incorrect header check");
    if (n >= 3 && b[2] != 8) return zs_msg(s, ZNGAMD_DATA_ERROR, "unknown compression method");
    if (n >= 4 && (b[3] & 0xE0)) return zs_msg(s, ZNGAMD_DATA_ERROR, "unknown header flags set");
    if (n < 10) return 0;
    const int flags = b[3];
    size_t cur = 10;
    if (flags & 4) {
        if (cur + 2 >= n) return 0;
        cur += 2 + (size_t)(b[cur] | (b[cur + 1] << 8));
        if (cur >= n) return 0;
    }
    for (int bit = 8; bit <= 16; bit <<= 1) {
        if (flags & bit) {
            const void *z = memchr(b.data() + cur, 0, n - cur);
            if (!z) return 0;
            cur = (size_t)((const uint8_t *)z - b.data()) + 1;
        }
    }
    if (flags & 2) {
        if (cur + 2 >= n) return 0;
        uint32_t hc = 0;
        const int r = zngamd_crc32(s->state->ctx, 0, b.data(), cur, &hc);
        if (r) return zs_msg(s, ZNGAMD_MEM_ERROR, zngamd_last_error(s->state->ctx));
        if ((hc & 0xFFFFu) != (uint32_t)(b[cur] | (b[cur + 1] << 8))) return zs_msg(s, ZNGAMD_DATA_ERROR, "header crc mismatch");
        cur += 2;
    }
    *start = cur;
    return 1;
}

// container header at the front of the buffer: 1 done, 0 need more input, Z_NEED_DICT, < 0 error
static int zs_inflate_header(zngamd_stream *s)
{
    ZsInflate &I = s->state->i;
    std::vector<uint8_t> &b = I.buf;
    if (I.kind == 3 && b.size() >= 2) { I.kind = (b[0] == 0x1f && b[1] == 0x8b) ? 2 : 1; I.wbits -= 32; }
    if (I.kind == 1) {
        if (b.size() < 2) return 0;
        const unsigned cmf = b[0], flg = b[1];
        if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31) return zs_msg(s, ZNGAMD_DATA_ERROR, "incorrect header check");
        if ((int)(cmf >> 4) + 8 > (I.wbits ? I.wbits : 15)) return zs_msg(s, ZNGAMD_DATA_ERROR, "invalid window size");
        if (flg & 0x20) {
            if (b.size() < 6) return 0;
            s->adler = ((uint32_t)b[2] << 24) | ((uint32_t)b[3] << 16) | ((uint32_t)b[4] << 8) | b[5];
            I.want_dict = true;
            return ZNGAMD_NEED_DICT;
        }
        b.erase(b.begin(), b.begin() + 2);
        I.check = 1;
    } else if (I.kind == 2) {
        size_t start = 0;
        const int r = zs_gzip_header_parse(s, b, &start);
        if (r <= 0) return r;
        b.erase(b.begin(), b.begin() + start);
        I.check = 0;
    } else return 0;
    I.header_done = true;
    return 1;
}

// trailer behind the deflate data (front of the buffer): 1 = end of stream verified (what follows is left in the buffer as
// unused input), 0 = more input needed, < 0 = error
static int zs_inflate_trailer(zngamd_stream *s)
{
    ZsInflate &I = s->state->i;
    const size_t need = I.kind == 1 ? 4 : I.kind == 2 ? 8 : 0;
    const std::vector<uint8_t> &b = I.buf;
    auto le32 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24); };
    if (I.kind == 2 && b.size() >= 4 && b.size() < 8 && le32(0) != I.check) return zs_msg(s, ZNGAMD_DATA_ERROR, "incorrect data check");   // inflate() compares the CRC as soon as its bytes are there
    if (b.size() < need) return 0;
    if (I.kind == 1 && ((((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]) != I.check)) return zs_msg(s, ZNGAMD_DATA_ERROR, "incorrect data check");
    if (I.kind == 2) {
        if (le32(0) != I.check) return zs_msg(s, ZNGAMD_DATA_ERROR, "incorrect data check");
        if (le32(4) != (uint32_t)(I.total & 0xFFFFFFFFull)) return zs_msg(s, ZNGAMD_DATA_ERROR, "incorrect length check");
    }
    I.buf.erase(I.buf.begin(), I.buf.begin() + need);
    I.eof = true;
    return 1;
}

int zngamd_stream_inflate(zngamd_stream *s, int flush)
try {
    if (!s || !s->state || s->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    zngamd_stream_state *st = s->state;
    ZsInflate &I = st->i;
    zngamd_ctx *c = st->ctx;
    if ((!s->next_in && s->avail_in) || (!s->next_out && s->avail_out)) return ZNGAMD_STREAM_ERROR;
    (void)flush;
    const uint32_t out0 = s->avail_out;
    // output decoded earlier (a call whose buffer was smaller than what one engine call produced)
    if (zs_drain(s) && s->avail_out == 0 && !st->outq.empty()) return ZNGAMD_OK;
    if (I.eof) return st->outq.empty() ? ZNGAMD_STREAM_END : ZNGAMD_OK;
    if (I.want_dict) return ZNGAMD_NEED_DICT;
    // all offered input moves into the buffer; what the stream does not need is handed back below (avail_in)
    const size_t fed = s->avail_in;
    const size_t before = I.buf.size();
    if (fed) { I.buf.insert(I.buf.end(), s->next_in, s->next_in + fed); s->next_in += fed; s->total_in += fed; s->avail_in = 0; }
    auto give_back = [&](size_t left) {       // the last `left` bytes of the buffer were not consumed: they stay the caller's
        left = std::min(left, fed);
        I.buf.resize(I.buf.size() - left);
        s->next_in -= left; s->total_in -= left; s->avail_in = (uint32_t)left;
    };
    if (!I.header_done) {
        const int r = zs_inflate_header(s);
        if (r == ZNGAMD_NEED_DICT) {
            // only the header stays here: what came behind it goes back to the caller, who offers it again once the dictionary
            // is set (zng_inflate stops consuming at the dictionary id too) -- bytes behind the stream's end must reach `unused_data`
            give_back(I.buf.size() > 6 ? I.buf.size() - 6 : 0);
            return r;
        }
        if (r < 0) return r;
        if (r == 0) return (fed || out0 != s->avail_out) ? ZNGAMD_OK : ZNGAMD_BUF_ERROR;
    }
    if (!I.deflate_done) {
        if (s->avail_out == 0) { give_back(I.buf.size() > before ? I.buf.size() - before : 0); return (out0 != s->avail_out) ? ZNGAMD_OK : ZNGAMD_BUF_ERROR; }
        if (I.buf.empty()) return (fed || out0 != s->avail_out) ? ZNGAMD_OK : ZNGAMD_BUF_ERROR;
        // decode from the last block header; `skip` bytes of that block were delivered before
        // the caller's buffer -- or what it has announced it will take by growing that buffer: one engine call then decodes that
        // far (a 16 KiB buffer doubled step by step would otherwise decode the same block again for every step)
        const uint64_t room = std::max<uint64_t>(s->avail_out, I.ahead);
        const uint64_t want = room > ~0ull - I.skip ? ~0ull : I.skip + room;
        uint64_t cap = std::max<uint64_t>(1u << 16, 8ull * I.buf.size() + I.skip);
        if (want < cap) cap = want;
        std::vector<uint8_t> out;
        uint64_t out_len = 0, in_bits = 0, bb = 0, bo = 0;
        int code;
        for (;;) {
            out.resize(cap ? cap : 1);
            code = zngamd_inflate_resume(c, I.buf.data(), I.buf.size(), I.start_bit, I.window.data(), (uint32_t)I.window.size(), out.data(), cap ? cap : 1,
                                         &out_len, &in_bits, &bb, &bo);
            if (code == ZNGAMD_E_OVERFLOW && cap < want) { cap = std::min<uint64_t>(cap * 4, want); continue; }      // our guess was short, not the caller's buffer
            break;
        }
        if (code == ZNGAMD_E_HIP || code == ZNGAMD_E_ARG || code == ZNGAMD_MEM_ERROR) return zs_msg(s, ZNGAMD_MEM_ERROR, zngamd_last_error(c));
        if (code == ZNGAMD_DATA_ERROR) return zs_msg(s, ZNGAMD_DATA_ERROR, nullptr);
        // "input ran out" is ZNGAMD_OK / ZNGAMD_BUF_ERROR and nothing else: any other code leaves the resume state as it is
        if (code != ZNGAMD_OK && code != ZNGAMD_BUF_ERROR && code != ZNGAMD_STREAM_END && code != ZNGAMD_E_OVERFLOW)
            return zs_msg(s, ZNGAMD_STREAM_ERROR, zngamd_last_error(c));
        const uint64_t nnew = out_len > I.skip ? out_len - I.skip : 0;
        if (nnew) {
            uint32_t v = I.check;
            int r = ZNGAMD_OK;
            if (I.kind == 2) r = zngamd_crc32(c, v, out.data() + I.skip, nnew, &v);
            else if (I.kind == 1) r = zngamd_adler32(c, v, out.data() + I.skip, nnew, &v);
            if (r) return zs_msg(s, ZNGAMD_MEM_ERROR, zngamd_last_error(c));
            I.check = v; I.total += nnew;
            I.ahead = I.ahead > nnew ? I.ahead - nnew : 0;
            zs_put(st, out.data() + I.skip, nnew);
        }
        auto set_window = [&](uint64_t upto) {       // history = old window + out[0 .. upto)
            std::vector<uint8_t> w(I.window);
            w.insert(w.end(), out.begin(), out.begin() + upto);
            if (w.size() > ZA_WIN) w.erase(w.begin(), w.end() - ZA_WIN);
            I.window.swap(w);
        };
        if (code == ZNGAMD_STREAM_END) {
            I.buf.erase(I.buf.begin(), I.buf.begin() + std::min<size_t>((size_t)((in_bits + 7) / 8), I.buf.size()));
            I.deflate_done = true; I.skip = 0;
        } else if (code == ZNGAMD_E_OVERFLOW) {
            // the caller's buffer is full in the middle of a block: the input behind the stop position goes back to the caller,
            // the resume point moves to the header of the block the decoder stopped in
            const size_t stop = std::min<size_t>((size_t)((in_bits + 7) / 8), I.buf.size());
            give_back(I.buf.size() - stop);
            set_window(bo);
            I.buf.erase(I.buf.begin(), I.buf.begin() + std::min<size_t>((size_t)(bb / 8), I.buf.size()));
            I.start_bit = (uint32_t)(bb & 7u);
            I.skip = out_len - bo;
        } else {                               // input ran out: keep what follows the last block header
            set_window(bo);
            I.buf.erase(I.buf.begin(), I.buf.begin() + std::min<size_t>((size_t)(bb / 8), I.buf.size()));
            I.start_bit = (uint32_t)(bb & 7u);
            I.skip = out_len - bo;
        }
    }
    if (I.deflate_done && !I.eof) {
        const int r = zs_inflate_trailer(s);
        if (r < 0) { zs_drain(s); return r; }
        if (r == 1) {
            // bytes behind the trailer are not this stream's: hand back what came in with this call, drop nothing else
            give_back(I.buf.size());
            I.buf.clear();
        }
    }
    zs_drain(s);
    s->adler = I.check;
    if (I.eof && st->outq.empty()) return ZNGAMD_STREAM_END;
    if (fed == s->avail_in && out0 == s->avail_out && fed == 0) return ZNGAMD_BUF_ERROR;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_inflate_ahead(zngamd_stream *s, uint64_t bytes)
try {
    if (!s || !s->state || s->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    s->state->i.ahead = bytes;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_inflate_copy(zngamd_stream *dst, const zngamd_stream *src)
try {
    if (!dst || !src || !src->state || src->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    *dst = *src;
    dst->state = new zngamd_stream_state(*src->state);
    dst->msg = nullptr;
    return ZNGAMD_OK;
} ZS_GUARD

// zng_inflateReset (zlib_ngmodule.c:2525, :2715): back to the state right behind inflate_init with the same wbits -- the next
// byte offered is the first byte of a new stream (the gzip reader calls it between members)
int zngamd_stream_inflate_reset(zngamd_stream *s)
try {
    if (!s || !s->state || s->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    zngamd_stream_state *st = s->state;
    ZsInflate fresh;
    fresh.wbits = st->i.wbits0; fresh.wbits0 = st->i.wbits0; fresh.kind = st->i.kind0; fresh.kind0 = st->i.kind0;
    fresh.header_done = fresh.kind == 0;
    st->i = fresh;
    st->outq.clear(); st->outpos = 0; st->msg.clear();
    s->msg = nullptr; s->total_in = s->total_out = 0; s->adler = 1;
    return ZNGAMD_OK;
} ZS_GUARD

int zngamd_stream_inflate_end(zngamd_stream *s)
try {
    if (!s || !s->state || s->state->is_deflate) return ZNGAMD_STREAM_ERROR;
    delete s->state;
    s->state = nullptr; s->msg = nullptr;
    return ZNGAMD_OK;
} ZS_GUARD

}  // extern "C"
