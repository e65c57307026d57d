/*
 * oracle/ -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the hot path of pycompression/python-zlib-ng
 * (DEFLATE / inflate / CRC-32 / Adler-32 / crc32_combine / gzip+zlib framing).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (python-zlib-ng_amd/) never links, imports or calls it.
 *
 * Where the algorithm comes from:
 *   - The reference's arithmetic lives in the third-party zlib-ng library, an
 *     UN-VENDORED submodule (reference .gitmodules:1-3, documented version
 *     2.1.5, CHANGELOG.rst:60) that is absent from /root/reference.  Inflate,
 *     CRC-32, Adler-32 and crc32_combine are restated from their published
 *     definitions (RFC 1950 / 1951 / 1952) and are pinned by the reference's
 *     own known-answer tests and data files (tests/golden/, see
 *     tests/test_oracle_pinning.py).
 *   - DEFLATE *compressed bytes* are pinned nowhere in the reference (every
 *     reference compression test is a round trip):
 *     "compressed-bytes parity unpinned".  The deflate restated here is this
 *     repo's own deterministic codec specification ("ZA codec", DESIGN.md
 *     section 3) which the HIP kernels must reproduce bit for bit; its output
 *     is pinned by round trips through this oracle's inflater and through the
 *     system zlib.
 *   - Call-site semantics follow the reference's C module:
 *       compress_and_crc     src/zlib_ng/zlib_ngmodule.c:1696-1782
 *       one-shot containers  src/zlib_ng/zlib_ngmodule.c:199-373
 *       gzip member reader   src/zlib_ng/zlib_ngmodule.c:2426-2637
 *       checksums            src/zlib_ng/zlib_ngmodule.c:1455-1596
 *       threaded framing     src/zlib_ng/gzip_ng_threaded.py:269-338
 */
#ifndef ZA_ORACLE_H
#define ZA_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- codec constants ---- */
#define ZA_SEG        2048      /* segment of a full unit: token boundaries are forced at segment ends (za_o_seg_shift: small units take smaller ones) */
#define ZA_SMALL_UNIT 16384     /* unit size of a one-shot stream of up to ZA_MAX_UNIT bytes (za_o_deflate_stream)  */
#define ZA_MAX_UNIT   131072    /* one codec unit = at most 64 segments             */
#define ZA_MAX_SEGS   64
#define ZA_WIN        32768
#define ZA_HASH_BITS  14
#define ZA_MIN_MATCH  3
#define ZA_HASH_BYTES_A 5       /* context bytes of link table A (chains), B and C (nearest occurrence only) */
#define ZA_HASH_BYTES_B 3
#define ZA_HASH_BYTES_C 12
#define ZA_MAX_MATCH  258
#define ZA_DP_WEAK_DIST 256     /* cost statistics: a 3-byte match farther back than this counts as literals */
#define ZA_DP_SUB      4         /* the dynamic programme also tries the 4 next shorter lengths of a position's match */

#define ZA_FLAG_FINAL 1         /* last block gets BFINAL=1, no sync-flush marker    */
#define ZA_FLAG_SEG2K 16        /* segments of 2 KiB whatever the unit's size (the threaded writer: its segment index counts in them) */
#define ZA_FLAG_FLATHDR 2       /* dynamic header in its flat form: the code-length code is the fixed 4-bit code of
                                   the symbols 0..15 (no run-length symbols), so every code length sits at a known bit
                                   offset and a decoder can read the header in parallel (indexed gzip members)        */
#define ZA_LIMIT_L    10        /* longest literal/length code: a 2^11-entry table decodes every symbol in one step */
#define ZA_LIMIT_D    9         /* longest distance code                                                            */
#define ZA_CHUNK_SHIFT 11       /* index granularity of indexed members: one entry per 2 KiB segment */
#define ZA_MAX_CHUNKS (ZA_MAX_UNIT >> ZA_CHUNK_SHIFT)

/* return codes, zlib numbering (zlib_ngmodule.c:68-95 maps them to messages) */
#define ZA_OK            0
#define ZA_STREAM_END    1
#define ZA_NEED_DICT     2
#define ZA_STREAM_ERROR (-2)
#define ZA_DATA_ERROR   (-3)
#define ZA_MEM_ERROR    (-4)
#define ZA_BUF_ERROR    (-5)

/* ---- checksums (zlib_ngmodule.c:1455-1596) ---- */
uint32_t za_o_crc32(uint32_t crc, const uint8_t *buf, size_t len);
uint32_t za_o_adler32(uint32_t adler, const uint8_t *buf, size_t len);
uint32_t za_o_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

/* ---- deflate: one unit (<= ZA_MAX_UNIT bytes) with up to 32 KiB of preceding dictionary.
 * `data` points at the unit; data[-dict_len .. -1] must be readable (the dictionary).
 * Mirrors compress_and_crc (zlib_ngmodule.c:1725-1742): reset, set dictionary, crc32, one
 * deflate(Z_SYNC_FLUSH).  Returns compressed size or a negative ZA_* code (ZA_BUF_ERROR when
 * `cap` is too small).  Optional debug outputs may be NULL. */
typedef struct {
    uint16_t *prevdist;   /* [dict_len + n]  links of table A (stage 1)             */
    uint32_t *best;       /* [n]  len<<16 | dist, 0 = none (stage 2, before the dynamic programme) */
    uint32_t *tokens;     /* [n]  tokens at segment slots seg*ZA_SEG (stage 3)       */
    uint32_t *seg_ntok;   /* [ZA_MAX_SEGS]                                           */
    uint32_t *hist;       /* [320] 0..285 lit/len, 288..317 dist (stage 3)           */
    uint8_t  *lens;       /* [320] code lengths, same layout (stage 4)               */
    uint32_t *seg_bits;   /* [ZA_MAX_SEGS+1] bit offset of each segment's first token
                             from the unit's first byte; [nseg] = offset of EOB      */
    int      *btype;      /* 0 stored, 1 fixed, 2 dynamic                            */
    uint32_t *chunk_idx;  /* [ZA_MAX_CHUNKS+1] entry c: bit offset (23 bits) of the first token that starts at
                             or behind output byte c*256, | (that token's start - c*256) << 23;
                             entry nchunk = offset of EOB (stage 5; Huffman blocks only)  */
    uint16_t *linkB;      /* [dict_len + n]  links of table B (stage 1)             */
    uint16_t *linkC;      /* [dict_len + n]  links of table C (stage 1; levels that use it) */
    uint32_t *best_dp;    /* [n]  the entries as the dynamic programme leaves them (stage 3a; = best on levels 1-3) */
    uint32_t *dp_cost;    /* [258] the unit's cost table in quarter bits: [0..255] literals, [256] match base (stage 3a) */
} za_o_debug;

int za_o_seg_shift(int n, int flags);      /* log2 of the unit's segment size: 5 .. 11 */
long za_o_deflate_unit(const uint8_t *data, int dict_len, int n, int level, int flags,
                       uint8_t *out, size_t cap, uint32_t *crc, za_o_debug *dbg);

/* Whole buffer as a chain of units (block_size per reference block, each block split into
 * units of <= ZA_MAX_UNIT, every unit primed with the previous 32 KiB of *input* as in
 * gzip_ng_threaded.py:317).  Output = concatenation, raw deflate ending in a sync flush
 * (or a final block when flags has ZA_FLAG_FINAL).  Returns size or negative code. */
long za_o_deflate_stream(const uint8_t *data, size_t n, int level, int flags,
                         uint8_t *out, size_t cap);

/* ---- inflate: raw RFC 1951.  Decodes until a final block ends (returns ZA_STREAM_END) or
 * input runs out (ZA_BUF_ERROR).  *in_used / *out_len are set in all cases. dict may be NULL. */
int za_o_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                     const uint8_t *dict, size_t dict_len,
                     size_t *in_used, size_t *out_len);

/* gzip multi-member reader, restating GzipReader_read_into_buffer
 * (zlib_ngmodule.c:2426-2637).  Error codes below ZA_* for the framing errors. */
#define ZA_GZ_BAD_MAGIC   (-101)
#define ZA_GZ_BAD_METHOD  (-102)
#define ZA_GZ_BAD_HCRC    (-103)
#define ZA_GZ_BAD_CRC     (-104)
#define ZA_GZ_BAD_LENGTH  (-105)
#define ZA_GZ_TRUNCATED   (-106)
int za_o_gunzip(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                size_t *out_len, int *n_members);

/* zlib container (RFC 1950) decode: header check + raw inflate + adler32 check. */
int za_o_zlib_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                         size_t *out_len);

/* CPU-baseline helper for bench.py: compress `n_blocks` blocks of `block` bytes (dictionary =
 * previous 32 KiB of input) with `threads` pthreads, then inflate them; returns seconds via
 * out params.  Only used by bench.py's cpu_baseline leg. */
int za_o_bench_blocks(const uint8_t *data, size_t n, size_t block, int level, int threads,
                      double *t_deflate, double *t_inflate, size_t *comp_bytes);

#ifdef __cplusplus
}
#endif
#endif
