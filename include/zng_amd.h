/*
 * zng_amd.h -- C ABI of the MI355X-native DEFLATE / inflate engine (libzng_amd.so).
 *
 * This is the drop-in boundary for the hot path of pycompression/python-zlib-ng: the entry points
 * below are what a binding of the reference would call instead of the zlib-ng `zng_*` functions it
 * uses today.  Plain pointers and sizes only; no Python, no torch types.  Every function that
 * compresses, decompresses or checksums runs hand-written HIP kernels on the GPU -- there is no CPU
 * fallback, and a missing / unusable GPU is an error (ZNGAMD_E_HIP).
 *
 * Reference interface each group replaces (file:line in the reference tree):
 *   zngamd_deflate_blocks*      ParallelCompress_compress_and_crc      src/zlib_ng/zlib_ngmodule.c:1696-1782
 *                               (zng_deflateReset :1725, zng_deflateSetDictionary :1735,
 *                                zng_crc32_z :1741, zng_deflate(Z_SYNC_FLUSH) :1742), batched over the
 *                               blocks that gzip_ng_threaded.py:299-322 cuts and round-robins
 *   zngamd_level_ok             level validation of zng_deflateInit2    zlib_ngmodule.c:1645-1658
 *   zngamd_deflate_stream       zlib_compress_impl body                 zlib_ngmodule.c:199-273
 *   zngamd_inflate*             zng_inflate call sites                  zlib_ngmodule.c:328, :2539
 *   zngamd_gzip_scan / _members GzipReader_read_into_buffer             zlib_ngmodule.c:2426-2637
 *   zngamd_crc32 / _adler32     zlib_crc32 / zlib_adler32               zlib_ngmodule.c:1455-1562
 *   zngamd_crc32_combine        zlib_crc32_combine                      zlib_ngmodule.c:1585-1596
 */
#ifndef ZNG_AMD_H
#define ZNG_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zngamd_ctx zngamd_ctx;

/* status codes: zlib numbering for the codec, -2xx for the engine */
#define ZNGAMD_OK              0
#define ZNGAMD_STREAM_END      1
#define ZNGAMD_NEED_DICT       2
#define ZNGAMD_STREAM_ERROR  (-2)
#define ZNGAMD_DATA_ERROR    (-3)
#define ZNGAMD_MEM_ERROR     (-4)
#define ZNGAMD_BUF_ERROR     (-5)
#define ZNGAMD_E_GZ_MAGIC    (-101)
#define ZNGAMD_E_GZ_METHOD   (-102)
#define ZNGAMD_E_GZ_HCRC     (-103)
#define ZNGAMD_E_GZ_CRC      (-104)
#define ZNGAMD_E_GZ_LENGTH   (-105)
#define ZNGAMD_E_GZ_TRUNC    (-106)
#define ZNGAMD_E_HIP         (-201)   /* HIP runtime error or no usable GPU */
#define ZNGAMD_E_ARG         (-202)
#define ZNGAMD_E_OVERFLOW    (-203)   /* a block's compressed output reached its buffer size */
#define ZNGAMD_E_INDEX       (-204)   /* a segment index does not fit its stream: decode the stream without the index */

#define ZNGAMD_FLAG_FINAL      1u     /* block ends the deflate stream (BFINAL=1, no sync flush) */
#define ZNGAMD_FLAG_FLATHDR    2u     /* dynamic block headers in their flat form: the code-length code is the fixed 4-bit
                                         code of the symbols 0..15, so a decoder finds every code length at a known bit offset
                                         (what zngamd_gzip_members* writes; any inflater reads it as an ordinary dynamic header) */
#define ZNGAMD_FLAG_SEG2K      16u    /* segments of 2 KiB whatever the block's size (small blocks take smaller ones otherwise): what the
                                       * segment index of a dict-chained stream is counted in (zngamd_deflate_index) */
#define ZNGAMD_FLAG_UNITS16K   32u    /* the block is cut into units of 16 KiB (deflate blocks of their own, each with the 32 KiB before it as
                                        its dictionary) instead of 128 KiB: latency before size -- what zngamd_deflate_stream does by itself for
                                        inputs of up to 128 KiB (r06; +0.2 .. 1.2 % of compressed size, a 64 KiB call 970 -> 435 us) */
/* window of the stream the blocks belong to: match distances stay within 2^bits (deflateInit2's windowBits 9..15).
 * Taken from the FIRST block of a call and applied to all of them; 0 = 15. */
#define ZNGAMD_FLAG_WBITS(bits) (((uint32_t)(bits) & 15u) << 8)

#define ZNGAMD_UNIT_MAX        131072u  /* largest span one kernel unit covers */
#define ZNGAMD_SLOT_STRIDE     131136u  /* bytes reserved per unit in a device slot buffer */
#define ZNGAMD_SEG             2048u    /* parse / index segment */

/* ---- context ---- */
int         zngamd_device_count(void);
int         zngamd_ctx_create(int device, zngamd_ctx **out);
void        zngamd_ctx_destroy(zngamd_ctx *ctx);
const char *zngamd_last_error(zngamd_ctx *ctx);     /* message of the calling thread's last failing call (thread-local) */
const char *zngamd_version(void);
/* run all work of this context on a caller-owned hipStream_t (pass NULL to go back to the own stream) */
int         zngamd_set_stream(zngamd_ctx *ctx, void *hip_stream);
int         zngamd_sync(zngamd_ctx *ctx);

/* device memory helpers for callers that have no allocator of their own */
int zngamd_dmalloc(zngamd_ctx *ctx, size_t bytes, void **dptr);
int zngamd_dfree(zngamd_ctx *ctx, void *dptr);
int zngamd_h2d(zngamd_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int zngamd_d2h(zngamd_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* device-to-device copy / fill on the context's stream (ordered with the engine's kernels, no host synchronisation), and the
 * device's free / total memory: what a harness needs to do without a tensor library (bench.py) */
int zngamd_d2d(zngamd_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
int zngamd_dmemset(zngamd_ctx *ctx, void *dst_dev, int value, size_t bytes);
int zngamd_mem_info(zngamd_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- checksums (GPU) ---- */
int      zngamd_crc32(zngamd_ctx *ctx, uint32_t crc, const uint8_t *buf, size_t len, uint32_t *out);
int      zngamd_adler32(zngamd_ctx *ctx, uint32_t adler, const uint8_t *buf, size_t len, uint32_t *out);
int      zngamd_crc32_dev(zngamd_ctx *ctx, uint32_t crc, const void *dbuf, size_t len, uint32_t *out);
/* scalar GF(2) arithmetic on three integers (no data pass) */
uint32_t zngamd_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);
/* the same over a run of pieces: crc of (what `crc` covers, followed by n pieces of lens[i] bytes whose CRC-32s are crcs[i]) -- the fold
 * the reference's writer thread does block by block (src/zlib_ng/gzip_ng_threaded.py:392-396), as one call */
uint32_t zngamd_crc32_combine_many(uint32_t crc, const uint32_t *crcs, const uint64_t *lens, uint32_t n);

/* ---- deflate ---- */
int zngamd_level_ok(int level);   /* 1 for -1..9, else 0 ("Bad compression level") */

typedef struct {
    uint64_t off;        /* offset of the block's first byte in `in` */
    uint32_t len;        /* block length (any size; cut into <=128 KiB units inside) */
    uint32_t dict_len;   /* bytes of `in` directly before `off` that prime the window (<= 32768) */
    uint32_t flags;      /* ZNGAMD_FLAG_FINAL | ZNGAMD_FLAG_WBITS(n) */
    uint32_t reserved;
} zngamd_block;

/* Compress n_blocks independent blocks of a host buffer.  Block b's raw-deflate bytes go to
 * out + b*out_cap_per_block, its size to out_len[b], the CRC-32 of its input to crc[b].  Each block
 * ends on a byte boundary with a sync-flush marker (00 00 FF FF) unless FINAL.  A block whose output
 * reaches out_cap_per_block gets out_len[b] = 0xFFFFFFFF and the call returns ZNGAMD_E_OVERFLOW. */
int zngamd_deflate_blocks(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len,
                          const zngamd_block *blocks, uint32_t n_blocks, int level,
                          uint8_t *out, uint64_t out_cap_per_block,
                          uint32_t *out_len, uint32_t *crc);

/* The same blocks, their outputs back to back in `out` (block b at the sum of out_len[0 .. b-1]; *total = all of them): what a
 * writer that only concatenates the blocks needs (the reference's writer thread, src/zlib_ng/gzip_ng_threaded.py:382-398) --
 * the packed stream is copied from the device straight into `out`, no per-block slots in between.  A block whose output
 * reaches block_cap is an overflow as above (ZNGAMD_E_OVERFLOW, out_len[b] = 0xFFFFFFFF; `out` is then not to be used);
 * ZNGAMD_BUF_ERROR with *total = the size needed when out_cap is too small. */
int zngamd_deflate_blocks_packed(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len,
                                 const zngamd_block *blocks, uint32_t n_blocks, int level,
                                 uint8_t *out, uint64_t out_cap, uint64_t block_cap,
                                 uint32_t *out_len, uint32_t *crc, uint64_t *total);

/* Device-resident form.  d_in holds the input; blocks are cut into units as above.
 * d_slots must hold n_units * ZNGAMD_SLOT_STRIDE bytes, where n_units = zngamd_count_units(...).
 * Per-unit results stay on the device: d_unit_len[u], d_unit_crc[u]; unit -> block map in h_unit_block
 * (host, optional).  zngamd_gather_dev then packs the used bytes of the slots contiguously. */
uint32_t zngamd_count_units(const zngamd_block *blocks, uint32_t n_blocks);
int zngamd_deflate_blocks_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len,
                              const zngamd_block *blocks, uint32_t n_blocks, int level,
                              void *d_slots, uint32_t *d_unit_len, uint32_t *d_unit_crc,
                              uint32_t *h_unit_block);
/* The same blocks straight into ONE contiguous stream at d_out (what zngamd_deflate_blocks_dev + zngamd_gather_dev leave at
 * d_dst, without the slots and without the copy: every unit's exact size is known before it is packed, a prefix sum places it).
 * Per unit: d_unit_len, d_unit_crc and (optional) d_unit_off = its byte offset in the stream; *total_bytes (host) = the
 * stream's size.  ZNGAMD_BUF_ERROR with *total_bytes = the size needed when out_cap is too small.  Replaces, for a batch of
 * blocks, the copies of ParallelCompress_compress_and_crc's results (reference src/zlib_ng/zlib_ngmodule.c:1765-1777) and the
 * in-order concatenation of the writer thread (src/zlib_ng/gzip_ng_threaded.py:382-398). */
int zngamd_deflate_blocks_packed_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len,
                                     const zngamd_block *blocks, uint32_t n_blocks, int level,
                                     void *d_out, uint64_t out_cap, uint32_t *d_unit_len, uint32_t *d_unit_crc,
                                     uint64_t *d_unit_off, uint64_t *total_bytes);
/* ---- the writer's segment index for a dict-chained stream (r06).  A batch compressed with ZNGAMD_FLAG_FLATHDR on its blocks
 * (one Huffman block per unit with the header in its flat form, 2 KiB segments) leaves, per unit, ZNGAMD_INDEX_STRIDE u32: entry s
 * = bit offset of the first token of segment s from the unit's first byte, entry nseg = bit offset of the end-of-block code, the
 * rest zeros (all zeros: a unit of stored blocks).  zngamd_deflate_index_dev copies the index of the context's LAST deflate call
 * (n_units as zngamd_count_units gave it) to device memory of the caller.  zngamd_inflate_units_indexed_dev decodes such a
 * stream -- what the reference's threaded writer frames as one gzip member (gzip_ng_threaded.py:299-338), and what
 * GzipReader_read_into_buffer (zlib_ngmodule.c:2426-2637) reads with one zng_inflate stream -- unit-parallel: a lane per 2 KiB
 * segment decodes, matches that reach in front of their unit become markers, the window kernels of the chunk-parallel inflate
 * resolve them.  unit_in_len / unit_out_len are HOST arrays (compressed bytes of a unit including its sync marker; its output
 * bytes), d_index device memory, d_dict / dict_len the history in front of the first unit (may be NULL / 0).  Returns
 * ZNGAMD_STREAM_END with *out_len, ZNGAMD_BUF_ERROR with the size needed, ZNGAMD_E_INDEX when stream and index do not fit (the
 * caller then decodes with zngamd_inflate_raw_dev), ZNGAMD_DATA_ERROR for invalid deflate data. */
#define ZNGAMD_INDEX_STRIDE 68
int zngamd_deflate_index_dev(zngamd_ctx *ctx, uint32_t *d_index, uint32_t n_units);
int zngamd_inflate_units_indexed_dev(zngamd_ctx *ctx, const void *d_def, uint64_t def_len, const uint32_t *unit_in_len,
                                     const uint32_t *unit_out_len, uint32_t n_units, const uint32_t *d_index,
                                     const void *d_dict, uint32_t dict_len, void *d_out, uint64_t out_cap, uint64_t *out_len);

/* Packs unit slots back to back at d_dst + dst_base; returns the total in *total_bytes (host).
 * d_unit_off (device, n_units x u64, may be NULL) receives each unit's byte offset. */
int zngamd_gather_dev(zngamd_ctx *ctx, const void *d_slots, const uint32_t *d_unit_len, uint32_t n_units,
                      void *d_dst, uint64_t dst_base, uint64_t dst_cap, uint64_t *d_unit_off,
                      uint64_t *total_bytes);

/* One whole raw-deflate stream from a host buffer: units chained through the previous 32 KiB of
 * input, last block FINAL.  window_bits 9..15 bounds match distances to 2^window_bits (the window a
 * decoder opened with that wbits keeps).  Returns the size in *out_len, CRC-32 and Adler-32 of the input. */
int zngamd_deflate_stream(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len, int level, int window_bits,
                          uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                          uint32_t *crc, uint32_t *adler);

/* ---- inflate ---- */
/* Raw RFC 1951 stream from a host buffer, optional preset dictionary.  Returns ZNGAMD_STREAM_END when
 * a final block ended, ZNGAMD_BUF_ERROR when input ran out or out_cap was reached (distinguish with
 * *out_len == out_cap), ZNGAMD_DATA_ERROR on invalid data.  *in_used = bytes consumed.
 * Complete streams of at least 64 KiB without a dictionary are decoded chunk-parallel (see zngamd_gunzip,
 * path 3); for those an output buffer that is too small gives ZNGAMD_BUF_ERROR with *out_len > out_cap =
 * the size the stream needs, and nothing is copied. */
int zngamd_inflate_raw(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len,
                       const uint8_t *dict, uint32_t dict_len,
                       uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint64_t *in_used,
                       uint32_t *crc, uint32_t *adler);

/* Resumable form for incremental readers (decompressobj).  Decoding starts `start_bit` (0..7) bits into
 * in[0] with `dict` as history.  Besides the totals it reports the last deflate-block header that was
 * entered (*block_bits from bit 0 of `in`, *block_out bytes produced before it): when the call ends with
 * ZNGAMD_BUF_ERROR (input ran out), the caller keeps in[*block_bits/8 ..], the last 32 KiB of output up to
 * *block_out, and calls again once more input has arrived.  ZNGAMD_E_OVERFLOW = out_cap reached.
 * Pieces of at least 64 KiB are decoded chunk-parallel up to their last complete block (then *out_len == *block_out:
 * the incomplete block is not decoded at all); smaller pieces run on the sequential wavefront decoder. */
int zngamd_inflate_resume(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len, uint32_t start_bit,
                          const uint8_t *dict, uint32_t dict_len, uint8_t *out, uint64_t out_cap,
                          uint64_t *out_len, uint64_t *in_bits, uint64_t *block_bits, uint64_t *block_out);

/* gzip member table entry produced by the index scan (pass 1) */
typedef struct {
    uint64_t in_off;       /* offset of the member's first deflate byte */
    uint64_t in_len;       /* deflate bytes (0 = unknown: decode sequentially)                     */
    uint64_t out_off;      /* offset of the member's first output byte                             */
    uint32_t out_len;      /* ISIZE                                                                 */
    uint32_t crc;          /* CRC-32 from the trailer                                               */
    uint32_t index_off;    /* bytes from the start of the chunk index to in_off, 0 = no index       */
    uint32_t nseg;         /* index entries - 1 = ceil(out_len / 256)                              */
} zngamd_member;

/* Pass 1 on the device: find every member of a multi-member stream written by this engine
 * (FEXTRA subfield 'Z','A' carrying member size, ISIZE and the chunk bit index).  d_members must
 * hold max_members entries; *n_members / *total_out on the host.  Returns ZNGAMD_E_ARG when the
 * stream is not fully made of indexed members (caller then uses the sequential reader). */
int zngamd_gzip_scan_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len,
                         zngamd_member *d_members, uint32_t max_members,
                         uint32_t *n_members, uint64_t *total_out);
/* Pass 2: decode all members (one wavefront per member, one lane per 2 KiB segment of the chunk index:
 * symbols decoded through LDS tables, matches resolved in an LDS image of the segment), verify CRC-32 and
 * ISIZE.  d_status[m] receives a ZNGAMD_* code per member. */
int zngamd_gzip_inflate_members_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len,
                                    const zngamd_member *d_members, uint32_t n_members,
                                    void *d_out, uint64_t out_cap, int32_t *d_status);

/* Pass 2 for members WITHOUT this engine's index whose extent is known (BGZF 'B','C' members, members written by any
 * gzip with their sizes recorded by the caller): d_members[i].index_off = 0, in_off / in_len = the member's deflate bytes
 * (the 8-byte trailer follows them), out_off / out_len = where its ISIZE bytes go.  One wavefront per member (64
 * self-synchronising sub-sequences inside every Huffman block), CRC-32 / ISIZE verified against the trailer. */
int zngamd_gzip_inflate_plain_members_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len,
                                          const zngamd_member *d_members, uint32_t n_members,
                                          void *d_out, uint64_t out_cap, int32_t *d_status);

/* One raw deflate stream that lies in device memory (d_in must be readable 64 bytes past in_len), decoded into device
 * memory: chunk-parallel where the stream offers block boundaries (sync-flush points, dynamic block headers), else on one
 * wavefront.  Returns ZNGAMD_STREAM_END when the final block ended; *out_len = bytes produced, *in_used = bytes consumed. */
int zngamd_inflate_raw_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len, void *d_out, uint64_t out_cap,
                           uint64_t *out_len, uint64_t *in_used);

/* CRC-32 of the concatenation of n pieces from their CRC-32s (a device array, e.g. the d_unit_crc of
 * zngamd_deflate_blocks_dev): pieces 0 .. n-2 are each_len bytes long, the last one last_len -- the in-order fold of the
 * reference's writer thread (gzip_ng_threaded.py:394), GF(2) arithmetic on the host. */
int zngamd_crc32_fold_dev(zngamd_ctx *ctx, const uint32_t *d_crcs, uint32_t n, uint64_t each_len, uint64_t last_len, uint32_t *crc);

/* Number of differing 4-byte words (bytes in the tail) of two device buffers: the round-trip check of device-resident callers. */
int zngamd_compare_dev(zngamd_ctx *ctx, const void *d_a, const void *d_b, uint64_t n, uint64_t *mismatches);

/* Host-buffer gzip reader: any multi-member gzip stream (headers with FEXTRA/FNAME/FCOMMENT/FHCRC,
 * NUL padding between members).  Four decode paths, picked per stream / member:
 *   1. this engine's indexed members          -> two-pass, lane-parallel inside each member
 *   2. BGZF-style members ('B','C' subfield)  -> one wavefront per member, one launch; runs of small ordinary members
 *      (no member is more than 1 MiB of input from the next) likewise, after a count launch that finds where each ends
 *   3. any other member of at least 64 KiB with enough block boundaries -> chunk-parallel: chunk starts are
 *      the positions after sync-flush markers (block-parallel writers: gzip_ng_threaded, pigz) and the bit
 *      offsets where a dynamic block header parses (ordinary gzip files); a count-only pass sizes and
 *      validates them, a marker pass decodes the chunks independently, the 32 KiB windows are propagated
 *      along the chain and the markers resolved
 *   4. anything else (small members, stored / fixed-Huffman only streams) -> the sequential wavefront decoder
 * Paths 1-3 hand over to 4 whenever something does not check out, so errors are always the sequential
 * reader's.  *out_len = bytes produced (also on error), except that ZNGAMD_BUF_ERROR with
 * *out_len > out_cap means "the stream needs *out_len bytes of output" (size known up front). */
int zngamd_gunzip(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len,
                  uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members);

/* The same reader for a WINDOW of a longer stream (bounded-memory readers: GzipReader_read_into_buffer keeps a fixed
 * input buffer, zlib_ngmodule.c:2426-2450): every member that is complete inside the window is decoded; an incomplete
 * last member (cut header, deflate data or trailer) is left alone.  Returns ZNGAMD_OK with *in_consumed = offset of the
 * first byte not consumed: feed the stream from there next time, with more input behind it.  *in_consumed == 0 means the
 * window holds no complete member yet.  Real errors are reported as by zngamd_gunzip. */
int zngamd_gunzip_partial(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len,
                          uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members,
                          uint64_t *in_consumed);

/* The windowed reader with state: also a member LARGER than the window is decoded, block-wise.  *st starts zeroed.
 * At a member boundary this is zngamd_gunzip_partial; when not even the first member of the window is complete, the
 * complete deflate blocks of that member are decoded (chunk-parallel) and handed out, and *st remembers the bit offset
 * of the next block header, the last 32 KiB of output (the history that block may reference), the CRC-32 and length so
 * far.  The next call continues there: `in` must start at byte *in_consumed of the previous input.  last != 0: no more
 * input exists (a stream that does not end is an error).  *in_consumed == 0 with ZNGAMD_OK: supply a larger window. */
typedef struct zngamd_gz_state {
    uint32_t in_member;      /* 0 at a member boundary, 1 inside a member's deflate data */
    uint32_t start_bit;      /* bit (0..7) of in[0] where the next block header starts */
    uint32_t crc;            /* CRC-32 of the member's output so far */
    uint32_t window_len;     /* valid bytes in window[] */
    uint64_t out_total;      /* bytes the member has produced so far */
    uint8_t  window[32768];  /* the last window_len bytes of the member's output */
    void    *index;          /* (r06, may be NULL) zngamd_index_create's handle for the member being read: the writer's segment index,
                              * taken from the file's trailing members by the caller; the units inside a window then decode side by side */
} zngamd_gz_state;
/* The index of one data member for the windowed reader: per unit its compressed bytes (sync marker included), its output bytes and
 * its ZNGAMD_INDEX_STRIDE u32 of segment bit offsets (host arrays; see zngamd_deflate_index, which gives them for the units of the
 * context's last deflate call).  The handle belongs to the caller (zngamd_index_destroy) and must outlive every state that names it. */
int zngamd_index_create(zngamd_ctx *ctx, uint32_t n_units, const uint32_t *unit_in_len, const uint32_t *unit_out_len,
                        const uint32_t *rows, void **handle);
void zngamd_index_destroy(void *handle);
int zngamd_deflate_index(zngamd_ctx *ctx, uint32_t n_units, uint32_t *unit_in_len, uint32_t *unit_out_len, uint32_t *rows);
/* zngamd_deflate_blocks_packed and zngamd_deflate_index of ITS units in one call (n_units = zngamd_count_units(blocks, n_blocks)):
 * what a writer that shares its context with other threads calls -- between two calls another thread's deflate could slip, and
 * zngamd_deflate_index answers for the context's last deflate call, whoever made it.  Replaces, with the index, the writer
 * thread's join of its blocks (gzip_ng_threaded.py:382-398). */
int zngamd_deflate_blocks_packed_indexed(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                                         int level, uint8_t *out, uint64_t out_cap, uint64_t block_cap, uint32_t *out_len, uint32_t *crc,
                                         uint64_t *total, uint32_t n_units, uint32_t *unit_in_len, uint32_t *unit_out_len, uint32_t *rows);
int zngamd_gunzip_stream(zngamd_ctx *ctx, zngamd_gz_state *st, const uint8_t *in, uint64_t in_len, int last,
                         uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members,
                         uint64_t *in_consumed);

/* Build one indexed gzip member stream (one member per block, FINAL blocks, 'ZA' index) from a host
 * buffer.  level as above; block_size <= 128 KiB. */
int zngamd_gzip_members(zngamd_ctx *ctx, const uint8_t *in, uint64_t in_len, uint32_t block_size,
                        int level, uint8_t *out, uint64_t out_cap, uint64_t *out_len);
/* Device-resident form of the same (input and output stay in HBM). */
int zngamd_gzip_members_dev(zngamd_ctx *ctx, const void *d_in, uint64_t in_len, uint32_t block_size,
                            int level, void *d_out, uint64_t out_cap, uint64_t *out_len,
                            uint32_t *n_members);

/* ---- streaming: the zng_stream calling convention ------------------------------------------------
 * What a binding of the reference swaps in for zng_deflateInit2 / zng_deflate / zng_deflateSetDictionary / zng_deflateCopy /
 * zng_deflateEnd (zlib_ngmodule.c:394, :552, :743, :401, :811) and zng_inflateInit2 / zng_inflate / zng_inflateSetDictionary /
 * zng_inflateCopy / zng_inflateEnd (:477, :667, :995, :1150, :446, :893): same fields, same flush values (Z_FULL_FLUSH forgets the history: nothing behind
 * the flush point refers to anything in front of it), same return codes
 * (ZNGAMD_OK 0, ZNGAMD_STREAM_END 1, ZNGAMD_NEED_DICT 2, ZNGAMD_STREAM_ERROR -2, ZNGAMD_DATA_ERROR -3, ZNGAMD_MEM_ERROR -4,
 * ZNGAMD_BUF_ERROR -5), `msg` set where zng_inflate sets it ("incorrect header check", "invalid window size", "incorrect data
 * check", ...).  deflate collects input until a flush or 32 MiB and compresses it as one dictionary-chained engine batch
 * (a piece of 32 MiB or more handed in at once is compressed where it lies); inflate keeps the compressed bytes from the last
 * block header on, decodes from there (bit offset + 32 KiB of history) and hands out what is new; input the stream does not
 * need yet comes back through avail_in, as with zng_inflate. */
typedef struct zngamd_stream_state zngamd_stream_state;
typedef struct zngamd_stream {
    const uint8_t *next_in;   /* next input byte */
    uint32_t avail_in;        /* bytes available at next_in */
    uint64_t total_in;
    uint8_t *next_out;        /* next output byte goes here */
    uint32_t avail_out;       /* room at next_out */
    uint64_t total_out;
    const char *msg;          /* last error message, NULL if none */
    zngamd_stream_state *state;
    uint32_t adler;           /* Adler-32 (zlib) or CRC-32 (gzip) of the uncompressed data so far; DICTID after ZNGAMD_NEED_DICT */
    uint32_t reserved;
    /* zng_stream's allocator hooks, which the reference sets (zlib_ngmodule.c:210-212, :391-393, :474-476): accepted and never
     * called -- the engine's memory is device memory and its own host buffers; a binding keeps those three lines as they are */
    void *zalloc, *zfree, *opaque;
} zngamd_stream;
#define ZNGAMD_NO_FLUSH 0
#define ZNGAMD_PARTIAL_FLUSH 1
#define ZNGAMD_SYNC_FLUSH 2
#define ZNGAMD_FULL_FLUSH 3
#define ZNGAMD_FINISH 4
#define ZNGAMD_BLOCK 5
int zngamd_stream_deflate_init(zngamd_ctx *ctx, zngamd_stream *strm, int level, int method, int wbits, int mem_level, int strategy);
int zngamd_stream_deflate(zngamd_stream *strm, int flush);
int zngamd_stream_deflate_set_dictionary(zngamd_stream *strm, const uint8_t *dict, uint32_t len);
int zngamd_stream_deflate_copy(zngamd_stream *dst, const zngamd_stream *src);
/* output produced and not yet handed out (zng_deflatePending; also valid for an inflate stream): lets the caller size its buffer once */
int zngamd_stream_pending(const zngamd_stream *strm, uint64_t *pending);
/* zng_deflateReset (zlib_ngmodule.c:1725): the stream as deflate_init left it -- same level, container and window */
int zngamd_stream_deflate_reset(zngamd_stream *strm);
int zngamd_stream_deflate_end(zngamd_stream *strm);
int zngamd_stream_inflate_init(zngamd_ctx *ctx, zngamd_stream *strm, int wbits);
int zngamd_stream_inflate(zngamd_stream *strm, int flush);
int zngamd_stream_inflate_set_dictionary(zngamd_stream *strm, const uint8_t *dict, uint32_t len);
/* A caller that grows its output buffer (arrange_output_buffer, zlib_ngmodule.c:142-197) announces how much output it will take
 * in total over the next calls (its max_length; UINT64_MAX = as much as the input yields): the engine then decodes that far in one
 * batch and hands the bytes out as buffers arrive (zngamd_stream_pending says how many wait), instead of decoding the current
 * block again for every doubling of a 16 KiB buffer.  0 (the default) = strictly avail_out, as zng_inflate. */
int zngamd_stream_inflate_ahead(zngamd_stream *strm, uint64_t bytes);
int zngamd_stream_inflate_copy(zngamd_stream *dst, const zngamd_stream *src);
/* zng_inflateReset (zlib_ngmodule.c:2525, :2715): the stream as inflate_init left it; the next input byte starts a new stream */
int zngamd_stream_inflate_reset(zngamd_stream *strm);
int zngamd_stream_inflate_end(zngamd_stream *strm);

/* ---- multi-GPU exchange: RCCL over xGMI, one process per GPU (gzip_ng_threaded.py:233-246 gives every worker thread a
 * compressor, :316-321 deals the blocks round-robin, :382-398 drains them in order; here every rank owns a contiguous block
 * range and the ranks reassemble the member stream with ONE exchange step) -------------------------------------------------
 * librccl is loaded on the first call (dlopen), so the library itself does not depend on it.  The 128-byte unique id is made
 * on one rank and handed to the others by the launcher's own means (bench.py: a TCP socket on MASTER_ADDR). */
typedef struct zngamd_comm zngamd_comm;
#define ZNGAMD_COMM_ID_BYTES 128
int zngamd_comm_unique_id(uint8_t id[ZNGAMD_COMM_ID_BYTES]);
/* collective over all ranks: communicator bound to ctx's device, with a stream of its own (exchanges overlap ctx's kernels) */
int zngamd_comm_create(zngamd_ctx *ctx, const uint8_t id[ZNGAMD_COMM_ID_BYTES], int rank, int world, zngamd_comm **out);
void zngamd_comm_destroy(zngamd_comm *comm);
const char *zngamd_comm_last_error(zngamd_comm *comm);
/* number of ranks the communicator really has, as RCCL reports it (ncclCommCount) -- not what the launcher's environment claims */
int zngamd_comm_count(zngamd_comm *comm, int *ranks);
/* layout of the one output stream: all-gather of {compressed bytes, CRC-32 of the rank's input, input bytes} (24 bytes per
 * rank), then on every rank: sizes[world], the offset of the own slice, the total, the CRC-32 of the whole input folded with
 * crc32_combine in rank order, and the whole input length -- what header / trailer and a positional write need */
int zngamd_comm_layout(zngamd_comm *comm, uint64_t local_len, uint32_t local_crc, uint64_t local_ulen, uint64_t *sizes,
                       uint64_t *my_off, uint64_t *total, uint32_t *whole_crc, uint64_t *whole_ulen);
/* exact-size exchange of the slices (no padding): every rank sends d_local[0 .. sizes[rank]) to every other rank and
 * receives their slices at their offsets, grouped ncclSend / ncclRecv over all links at once; afterwards d_stream holds the
 * whole stream on every rank.  Starts behind the work queued on ctx's stream so far and returns at once;
 * zngamd_comm_wait blocks until the exchange is done. */
int zngamd_comm_allgather_stream(zngamd_comm *comm, const void *d_local, const uint64_t *sizes, void *d_stream, uint64_t stream_cap);
/* where the slices lie in the assembled stream: offs[r] = sizes[0] + ... + sizes[r-1]; returns the total (pure arithmetic, no GPU:
 * the placement rule of the exchange above and of a positional write, testable on its own) */
uint64_t zngamd_comm_offsets(const uint64_t *sizes, int world, uint64_t *offs);
int zngamd_comm_wait(zngamd_comm *comm);
/* plumbing for a driver: barrier, and the maximum of one double over the ranks (step time of the slowest rank) */
int zngamd_comm_barrier(zngamd_comm *comm);
int zngamd_comm_max_f64(zngamd_comm *comm, double *value);

/* ---- measurement ---- */
/* With profiling on, every kernel launch is bracketed by HIP events on the context's stream. */
#define ZNGAMD_K_CHAINS 0
#define ZNGAMD_K_SEARCH 1
#define ZNGAMD_K_PARSE  2
#define ZNGAMD_K_PLAN   3
#define ZNGAMD_K_PACK   4
#define ZNGAMD_K_GATHER 5
#define ZNGAMD_K_SCAN   6
#define ZNGAMD_K_INFLATE 7
#define ZNGAMD_K_OTHER  8
#define ZNGAMD_K_OPTPARSE 9     /* the dynamic programme between search and parse (levels 4-9) */
#define ZNGAMD_K_COUNT  10
int zngamd_profiling(zngamd_ctx *ctx, int on);
/* accumulated milliseconds and launch counts per kernel class since the last reset.  The library writes
 * zngamd_kernel_class_count() elements into each array: a caller built against an older header (ZNGAMD_K_COUNT was 9 before the
 * dynamic programme's class) asks the library, not its own header, how much room to give -- or compares ZNGAMD_ABI with
 * zngamd_abi() once. */
int zngamd_kernel_times(zngamd_ctx *ctx, double *ms /*[ZNGAMD_K_COUNT]*/, uint64_t *launches /*[ZNGAMD_K_COUNT]*/, int reset);
int zngamd_kernel_class_count(void);
#define ZNGAMD_ABI 6            /* bumped whenever an array size, a struct layout or an argument list of this header changes */
int zngamd_abi(void);

/* how many gzip members zngamd_gunzip (and whole streams zngamd_inflate_raw) decoded through each path since the last reset:
 * ZA-indexed two-pass, BGZF one-launch, chunk-parallel (sync points / block finder), one sequential wavefront */
#define ZNGAMD_PATH_INDEXED    0
#define ZNGAMD_PATH_BGZF       1
#define ZNGAMD_PATH_CHUNKED    2
#define ZNGAMD_PATH_SEQUENTIAL 3
#define ZNGAMD_PATH_COUNT      4
int zngamd_decode_paths(zngamd_ctx *ctx, uint64_t *members /*[ZNGAMD_PATH_COUNT]*/, int reset);
/* units decoded with a writer's segment index (zngamd_inflate_units_indexed_dev, or the windowed reader with zngamd_gz_state.index) */
uint64_t zngamd_indexed_units(zngamd_ctx *ctx, int reset);

/* ---- debugging aid for the parity tests: copy a stage's intermediate of unit `u` of the last
 * deflate call to the host.  what: 0 links of table A (u16) 1 best(u32, as the parse kernel read it: behind the dynamic
 * programme on levels 4-9) 2 tokens(u32) 3 seg_ntok(u32) 4 hist(u32) 5 codes(u32) 6 seg_bits(u32) 7 plan(4 x u32)
 * 8 chunk index(u32) 9 / 10 links of tables B / C (u16) 11 best as the search left it (u32) 12 the dynamic programme's cost
 * table (258 x u32, levels 4-9).  Stages 0, 9, 10 and 11 exist only for calls made after zngamd_debug_keep(ctx, 1): without it the
 * token words are written over the link tables (a third of the workspace saved), and nothing copies the search results aside. */
int zngamd_debug_keep(zngamd_ctx *ctx, int on);
int zngamd_debug_fetch(zngamd_ctx *ctx, int what, uint32_t unit, void *host_dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
