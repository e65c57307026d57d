// libzng_amd.so -- host side of the C ABI declared in include/zng_amd.h.  Product code.
// One translation unit: the kernel files are included so that no relocatable device code is needed.
//
// Nothing in this library computes a checksum, a match, a Huffman code or a decoded byte on the
// CPU: the host code cuts blocks into units, owns device workspaces and streams, launches the
// kernels and assembles container framing bytes (gzip / zlib headers and trailers).
#include "za_deflate.hip"
#include "za_inflate.hip"
#include "za_inflate_units.hip"
#include "za_checksum.hip"
#include "../../include/zng_amd.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static_assert(sizeof(zngamd_member) == sizeof(ZaMember), "member layout");
static_assert(ZNGAMD_SLOT_STRIDE % 4 == 0 && ZNGAMD_SLOT_STRIDE >= ZA_MAX_UNIT + 32, "slot stride");
static_assert(ZNGAMD_UNIT_MAX == ZA_MAX_UNIT && ZNGAMD_SEG == ZA_SEG, "constants");

// the level table (the same numbers as the oracle's): chain steps over table A, nice, cap, table C, dynamic programme,
// too_far3, too_far4.  Within 2 % of zlib 1.2.11 at the same level on held-out real files, at least zlib on the synthetic corpora.
#define ZA_CH_STREAMS_PER_CU (ZA_HASH_BITS >= 14 ? 2u : 3u)     // what the chain kernel's LDS (table + 14 KiB) lets a CU hold
static const ZaLevel ZA_LEVELS[10] = {
    {0, 0, ZA_WIN, 0, 0, 0, 0, 0},
    {1, 16, ZA_WIN, 16, 0, 0, 256, 4096}, {2, 16, ZA_WIN, 16, 0, 0, 256, 4096}, {3, 16, ZA_WIN, 16, 0, 0, 256, 4096},
    {2, 16, ZA_WIN, 16, 0, 1, 4096, 32768}, {2, 16, ZA_WIN, 16, 1, 1, 4096, 32768}, {3, 16, ZA_WIN, 16, 1, 1, 4096, 32768},
    {4, 32, ZA_WIN, 258, 1, 1, 4096, 32768}, {8, 64, ZA_WIN, 258, 1, 1, 4096, 32768}, {12, 128, ZA_WIN, 258, 1, 1, 4096, 32768}};

template <typename T> struct DevBuf {
    T *p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        // (small buffers grow with an eighth of slack so that a caller whose batches creep up does not reallocate every call; the
        // large per-chunk workspaces are sized exactly: an eighth of 48 GB is memory another context on the device may need)
        size_t want = n * sizeof(T) >= ((size_t)64 << 20) ? n + 64 : n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) { want = n; e = hipMalloc((void **)&p, want * sizeof(T)); }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct EvPair { hipEvent_t a, b; int cls; };

// Message of the last failing call, kept per calling thread: a context is shared by many threads (readers, writers, the
// pump of the threaded reader), and the message is read after the call has left the context's mutex.
static thread_local std::string za_tl_err;
struct ZaErrSlot {
    ZaErrSlot &operator=(const char *m) { za_tl_err = m; return *this; }
    ZaErrSlot &operator=(const std::string &m) { za_tl_err = m; return *this; }
    const char *c_str() const { return za_tl_err.c_str(); }
};

struct zngamd_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    ZaErrSlot err;
    std::mutex mu;
    // constant tables
    uint32_t *d_crc_table = nullptr, *d_x8k = nullptr;
    uint32_t *d_crc_slice4 = nullptr;                                         // CRC slice-by-4 table of za_k_inflate_members
    // deflate workspaces (per chunk of units)
    // units per launch (ZNGAMD_CHUNK_UNITS).  Workspace per unit (r06): link tables 3 x 320 KiB (two below level 5), entries 512 KiB,
    // about 5 KiB of small arrays -- 1.44 MiB; the token words (512 KiB) live IN the link tables' memory, which nobody reads once the
    // search is through (r05: 1.94 MiB with an eighth of slack on top).  16 384 units = 24 GB: a 4 GiB shard takes two launches per
    // kernel and 2.1 % longer than in one (32 768: 47 GB; measured on the r06 kernels: deflate 64.5 -> 65.9 ms, the device holds
    // 58.7 instead of 83.5 GB during the bench; 8 192: 68.4 ms, 46.3 GB) -- the default since r06, the review's choice.  A chunk
    // that does not fit the device's free memory is halved until it does.
    uint32_t chunk_units = 16384;
    DevBuf<uint16_t> links; DevBuf<uint32_t> best, tok, segtok, hist, codes; DevBuf<ZaPlan> plan;
    uint16_t *prev_p = nullptr, *linkb_p = nullptr, *linkc_p = nullptr; uint32_t *tok_p = nullptr;     // where the current chunk size puts the tables inside `links`, and the token words (inside `links` too unless zngamd_debug_keep asked for all stages to stay)
    bool debug_keep = false, last_kept = false; DevBuf<uint32_t> best_keep, dpcost;
    std::vector<uint32_t> last_ulen_host; bool last_ulen_on_host = false;      // ... and on the host, where the call fetched them with its results (the device copy lies in a staging buffer that an inflate call of another thread may give back)
    const uint32_t *last_unit_len = nullptr;     // the last deflate call's per-unit compressed sizes (device; the caller's array or a staging buffer of this context)     // zngamd_debug_keep: the search results as they were before the dynamic programme, its cost tables
    // per call
    DevBuf<ZaUnit> units; DevBuf<uint32_t> segbits, cidx, status, runs;
    uint32_t last_units = 0; bool last_single_chunk = false;
    std::vector<ZaUnit> last_hu;                 // the units of the last deflate call as the kernels saw them (zngamd_debug_fetch)
    std::vector<ZaUnit> plan_in; std::vector<uint32_t> plan_runs; uint32_t plan_ch = 0;     // the unit table the device holds was planned from this one: a caller that compresses batch after batch of the same shape pays for the planning once
    std::vector<zngamd_block> blocks_in; std::vector<ZaUnit> blocks_hu; uint64_t blocks_len = 0;
    uint32_t chain_slots = 512;                  // chain-kernel workgroups the device holds at once: CUs x ZA_CH_STREAMS_PER_CU (two with the 64 KiB table)
    uint32_t chain_run = 0;                      // ZNGAMD_CHAIN_RUN: fixed run length of the chain kernel (0 = sized to the device)
    // staging
    DevBuf<uint8_t> st_in, st_out, st_slots, st_aux, hdr; DevBuf<uint32_t> st_len, st_crc; DevBuf<uint64_t> st_off;
    DevBuf<uint64_t> ccand, csurv; DevBuf<ZaChunkRes> cres; DevBuf<ZaChunk> cchunks; DevBuf<uint16_t> out16, ccomp; DevBuf<uint8_t> winbuf;
    DevBuf<uint16_t> uarea; uint32_t uarea_marked = 0;     // symbol areas of the indexed unit decoder (32 768 marker symbols + 131 072 per unit) and how many of them have their markers
    DevBuf<ZaCkPart> ck; DevBuf<uint32_t> matchq; DevBuf<ZaCand> cands; DevBuf<ZaMember> members; DevBuf<int32_t> mstatus;
    void *d_small = nullptr;     // 256 B scratch for counters / results
    // profiling
    uint64_t paths[4] = {0, 0, 0, 0};            // members decoded per path, see zngamd_decode_paths
    uint64_t indexed_units = 0;                  // units decoded with a writer's index (zngamd_indexed_units)
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;      // pinned host staging for device-to-host results (grow-only)
    uint8_t *h_up = nullptr; size_t h_up_cap = 0;            // pinned host staging for small uploads (r06): the input of a small call, the unit / run tables
    uint8_t *h_tab = nullptr; size_t h_tab_cap = 0;
    hipEvent_t ev_up = nullptr, ev_tab = nullptr; bool up_busy = false, tab_busy = false;      // behind the last copy out of either: a buffer is written again only once that copy has run
    hipEvent_t ev_copy[2] = {nullptr, nullptr};              // ends of the staged device-to-host pieces
    bool prof = false; std::vector<EvPair> evs; std::vector<hipEvent_t> pool;
    double ms[ZNGAMD_K_COUNT] = {0}; uint64_t launches[ZNGAMD_K_COUNT] = {0};
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return ZNGAMD_E_HIP; } } while (0)

static int fail(zngamd_ctx *c, int code, const char *msg) { c->err = msg; return code; }
// no C++ exception leaves the C ABI (host allocations of the batch paths can fail): reported as ZNGAMD_MEM_ERROR
#define ZA_ABI_GUARD catch (const std::bad_alloc &) { return ZNGAMD_MEM_ERROR; } catch (...) { return ZNGAMD_E_ARG; }


// ZNGAMD_TRACE=1: wall clock of the host phases of a call on stderr, the stream drained at the end of each so that a phase
// owns what it launched (diagnostics only: the drains cost time; profiles/time_threaded_rw.py and time_oneshot.py read it)
static bool trace_on() { static const bool on = getenv("ZNGAMD_TRACE") != nullptr; return on; }
struct PhaseClock {
    zngamd_ctx *c; const char *name; std::chrono::steady_clock::time_point t;
    PhaseClock(zngamd_ctx *c_, const char *n) : c(c_), name(n) { if (trace_on()) t = std::chrono::steady_clock::now(); }
    ~PhaseClock()
    {
        if (!trace_on()) return;
        if (c) (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "zng_amd trace: %-34s %9.3f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count());
    }
};

static hipEvent_t ev_get(zngamd_ctx *c)
{
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ProfScope {
    zngamd_ctx *c; int cls; hipEvent_t a = nullptr;
    ProfScope(zngamd_ctx *c_, int cls_) : c(c_), cls(cls_) { if (c->prof) { a = ev_get(c); (void)hipEventRecord(a, c->stream); } }
    ~ProfScope() { if (c->prof) { hipEvent_t b = ev_get(c); (void)hipEventRecord(b, c->stream); c->evs.push_back({a, b, cls}); } }
};
static void prof_collect(zngamd_ctx *c)
{
    for (auto &e : c->evs) {
        float t = 0; (void)hipEventSynchronize(e.b);
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) { c->ms[e.cls] += t; c->launches[e.cls]++; }
        c->pool.push_back(e.a); c->pool.push_back(e.b);
    }
    c->evs.clear();
}

extern "C" {

const char *zngamd_version(void) { return "zng_amd 0.6 (gfx950)"; }
int zngamd_abi(void) { return ZNGAMD_ABI; }
int zngamd_kernel_class_count(void) { return ZNGAMD_K_COUNT; }

int zngamd_device_count(void)
try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} ZA_ABI_GUARD

int zngamd_ctx_create(int device, zngamd_ctx **out)
try {
    if (!out) return ZNGAMD_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ZNGAMD_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return ZNGAMD_E_HIP;
    zngamd_ctx *c = new zngamd_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ZNGAMD_E_HIP; }
    c->stream = c->own_stream;
    if (const char *e = getenv("ZNGAMD_CHUNK_UNITS")) { long v = atol(e); if (v >= 1 && v <= (1 << 20)) c->chunk_units = (uint32_t)v; }
    if (const char *e = getenv("ZNGAMD_CHAIN_RUN")) { long v = atol(e); if (v >= 1 && v <= (1 << 20)) c->chain_run = (uint32_t)v; }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->chain_slots = (uint32_t)cus * ZA_CH_STREAMS_PER_CU; }
    // tables: CRC-32 byte table and x^(8*2048*k) mod P
    uint32_t tab[256], x8k[64];
    for (uint32_t i = 0; i < 256; i++) { uint32_t v = i; for (int k = 0; k < 8; k++) v = (v & 1) ? (0xEDB88320u ^ (v >> 1)) : (v >> 1); tab[i] = v; }
    {
        uint32_t xs = 0x80000000u, sq = 0x00800000u;      // x^(8*2048): square x^8 eleven times
        for (int i = 0; i < 11; i++) sq = za_multmodp(sq, sq);
        for (int k = 0; k < 64; k++) { x8k[k] = xs; xs = za_multmodp(xs, sq); }
    }
    if (hipMalloc((void **)&c->d_crc_table, sizeof tab) != hipSuccess || hipMalloc((void **)&c->d_x8k, sizeof x8k) != hipSuccess ||
        hipMalloc(&c->d_small, 256) != hipSuccess ||
        hipMemcpy(c->d_crc_table, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_x8k, x8k, sizeof x8k, hipMemcpyHostToDevice) != hipSuccess) {
        zngamd_ctx_destroy(c); return ZNGAMD_E_HIP;
    }
    {   // tables of the indexed-member decoder's CRC-32: [0, 1024) slice-by-4; [1024, 1152) the raw state advanced over 2 016 zero
        // bytes (what the other 63 lanes hold of a 2 KiB segment), one 16-entry table per nibble of the state; [1152, 1280)
        // x^(8 * 32 k) mod P for k < 128
        // [1280, 2304) and [2304, 2817): the checksum kernel's multipliers (za_checksum.hip)
        std::vector<uint32_t> s4(ZA_CK_TABS);
        for (int i = 0; i < 256; i++) s4[i] = tab[i];
        for (int t = 1; t < 4; t++) for (int i = 0; i < 256; i++) s4[256 * t + i] = (s4[256 * (t - 1) + i] >> 8) ^ tab[s4[256 * (t - 1) + i] & 0xFF];
        uint32_t x32 = 0x00800000u;                               // x^8 -> x^(8*32): squared five times
        for (int i = 0; i < 5; i++) x32 = za_multmodp(x32, x32);
        uint32_t xs = 0x80000000u;
        for (int k = 0; k < 128; k++) { s4[1152 + k] = xs; xs = za_multmodp(xs, x32); }
        const uint32_t x2016 = s4[1152 + 63];                     // x^(8 * 2016)
        for (int k = 0; k < 8; k++) for (uint32_t v = 0; v < 16; v++) s4[1024 + 16 * k + v] = za_multmodp(x2016, v << (4 * k));
        {
            uint32_t xt = 0x80000000u;                            // x^(8 t), t <= 512
            for (int t = 0; t <= 512; t++) { s4[ZA_CK_XT + t] = xt; xt = za_multmodp(xt, 0x00800000u); }
            for (int j = 0; j < 4; j++) {                         // x^(8 * (64 << j) * k), k < 256
                const uint32_t step = s4[ZA_CK_XT + (64 << j)];
                uint32_t xk = 0x80000000u;
                for (int k = 0; k < 256; k++) { s4[ZA_CK_XS + 256 * j + k] = xk; xk = za_multmodp(xk, step); }
            }
        }
        if (hipMalloc((void **)&c->d_crc_slice4, s4.size() * 4) != hipSuccess ||
            hipMemcpy(c->d_crc_slice4, s4.data(), s4.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            zngamd_ctx_destroy(c); return ZNGAMD_E_HIP;
        }
    }
    *out = c;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

void zngamd_ctx_destroy(zngamd_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    c->links.release(); c->best_keep.release(); c->dpcost.release(); c->hdr.release(); c->best.release(); c->tok.release(); c->segtok.release(); c->hist.release(); c->codes.release();
    c->plan.release(); c->units.release(); c->segbits.release(); c->cidx.release(); c->status.release();
    c->st_in.release(); c->st_out.release(); c->st_slots.release(); c->st_aux.release(); c->st_len.release(); c->st_crc.release();
    c->ccand.release(); c->csurv.release(); c->cres.release(); c->cchunks.release(); c->out16.release(); c->ccomp.release(); c->winbuf.release(); c->uarea.release();
    c->st_off.release(); c->runs.release(); c->ck.release(); c->matchq.release(); c->cands.release(); c->members.release(); c->mstatus.release();
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_up) (void)hipHostFree(c->h_up);
    if (c->h_tab) (void)hipHostFree(c->h_tab);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->ev_tab) (void)hipEventDestroy(c->ev_tab);
    for (auto &e : c->ev_copy) if (e) (void)hipEventDestroy(e);
    if (c->d_crc_table) (void)hipFree(c->d_crc_table);
    if (c->d_x8k) (void)hipFree(c->d_x8k);
    if (c->d_crc_slice4) (void)hipFree(c->d_crc_slice4);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char *zngamd_last_error(zngamd_ctx *c) { return c ? c->err.c_str() : "no context"; }

int zngamd_set_stream(zngamd_ctx *c, void *s)
try {
    if (!c) return ZNGAMD_E_ARG;
    (void)hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_sync(zngamd_ctx *c)
try {
    if (!c) return ZNGAMD_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_dmalloc(zngamd_ctx *c, size_t bytes, void **dptr)
try {
    if (!c || !dptr) return ZNGAMD_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(dptr, bytes ? bytes : 1));
    return ZNGAMD_OK;
} ZA_ABI_GUARD
int zngamd_dfree(zngamd_ctx *c, void *dptr) { if (!c) return ZNGAMD_E_ARG; HIPCHK(c, hipFree(dptr)); return ZNGAMD_OK; }
int zngamd_h2d(zngamd_ctx *c, void *dst, const void *src, size_t bytes)
try {
    if (!c) return ZNGAMD_E_ARG;
    if (bytes) { HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_d2h(zngamd_ctx *c, void *dst, const void *src, size_t bytes)
try {
    if (!c) return ZNGAMD_E_ARG;
    if (bytes) { HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

// device-to-device copy and fill on the context's stream (ordered with the engine's kernels; no host synchronisation), and the
// device's free / total memory: what a harness needs to do without a tensor library
int zngamd_d2d(zngamd_ctx *c, void *dst, const void *src, size_t bytes)
try {
    if (!c) return ZNGAMD_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_dmemset(zngamd_ctx *c, void *dst, int value, size_t bytes)
try {
    if (!c) return ZNGAMD_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes) HIPCHK(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_mem_info(zngamd_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
try {
    if (!c || !free_bytes || !total_bytes) return ZNGAMD_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_decode_paths(zngamd_ctx *c, uint64_t *members, int reset)
try {
    if (!c || !members) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    for (int i = 0; i < ZNGAMD_PATH_COUNT; i++) { members[i] = c->paths[i]; if (reset) c->paths[i] = 0; }
    return ZNGAMD_OK;
} ZA_ABI_GUARD
uint64_t zngamd_indexed_units(zngamd_ctx *c, int reset) { if (!c) return 0; std::lock_guard<std::mutex> g(c->mu); const uint64_t v = c->indexed_units; if (reset) c->indexed_units = 0; return v; }
int zngamd_profiling(zngamd_ctx *c, int on) { if (!c) return ZNGAMD_E_ARG; c->prof = on != 0; return ZNGAMD_OK; }
int zngamd_kernel_times(zngamd_ctx *c, double *ms, uint64_t *launches, int reset)
try {
    if (!c) return ZNGAMD_E_ARG;
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (int i = 0; i < ZNGAMD_K_COUNT; i++) { if (ms) ms[i] = c->ms[i]; if (launches) launches[i] = c->launches[i]; if (reset) { c->ms[i] = 0; c->launches[i] = 0; } }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

// ---------------------------------------------------------------------------------------------
// checksums
// ---------------------------------------------------------------------------------------------
uint32_t zngamd_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    uint32_t p = 0x80000000u, sq = 0x00800000u;
    while (len2) { if (len2 & 1) p = za_multmodp(sq, p); sq = za_multmodp(sq, sq); len2 >>= 1; }
    return za_multmodp(p, crc1) ^ crc2;
}

uint32_t zngamd_crc32_combine_many(uint32_t crc, const uint32_t *crcs, const uint64_t *lens, uint32_t n)
{
    if (!crcs || !lens) return crc;
    for (uint32_t i = 0; i < n; i++) crc = zngamd_crc32_combine(crc, crcs[i], lens[i]);
    return crc;
}

// run the checksum kernel over a device buffer and fold the per-span partials
// In two halves, so that a caller with kernels of its own to run can put them between the two and wait once: checksum_launch
// enqueues the kernel and the copy of its partial results, checksum_fold (behind a synchronisation of the stream) folds them.
static int checksum_launch(zngamd_ctx *c, const uint8_t *d, uint64_t n, bool want_crc, bool want_adler, std::vector<ZaCkPart> &parts)
{
    parts.clear();
    if (n == 0) return ZNGAMD_OK;
    const uint64_t nspan = (n + ZA_MAX_UNIT - 1) / ZA_MAX_UNIT;
    if (nspan > 0x7FFFFFFFull) return fail(c, ZNGAMD_E_ARG, "buffer too large");
    HIPCHK(c, c->ck.ensure(nspan));
    {
        ProfScope ps(c, ZNGAMD_K_OTHER);
        hipLaunchKernelGGL(za_k_checksum, dim3((uint32_t)nspan), dim3(256), 0, c->stream, d, n, c->d_crc_slice4, c->ck.p, want_crc ? 1 : 0, want_adler ? 1 : 0);
    }
    HIPCHK(c, hipGetLastError());
    parts.resize(nspan);
    HIPCHK(c, hipMemcpyAsync(parts.data(), c->ck.p, nspan * sizeof(ZaCkPart), hipMemcpyDeviceToHost, c->stream));
    return ZNGAMD_OK;
}

static void checksum_fold(const std::vector<ZaCkPart> &parts, uint32_t *crc_io, uint32_t *adler_io)
{
    const uint64_t nspan = parts.size();
    if (crc_io) {
        // all spans but the last are ZA_MAX_UNIT long: one multiplier, x^(8 * ZA_MAX_UNIT) mod P, serves them (2 048 spans of a
        // 256 MiB result cost 0.6 ms when each call worked its multiplier out again)
        uint32_t crc = *crc_io;
        uint32_t xp = 0x80000000u, sq = 0x00800000u;                          // x^0, x^8
        for (uint64_t k = ZA_MAX_UNIT; k; k >>= 1) { if (k & 1) xp = za_multmodp(sq, xp); sq = za_multmodp(sq, sq); }
        for (uint64_t i = 0; i < nspan; i++) {
            if (parts[i].len == ZA_MAX_UNIT) crc = za_multmodp(xp, crc) ^ parts[i].crc;
            else crc = zngamd_crc32_combine(crc, parts[i].crc, parts[i].len);
        }
        *crc_io = crc;
    }
    if (adler_io) {
        uint64_t A = *adler_io & 0xFFFF, B = (*adler_io >> 16) & 0xFFFF;
        for (uint64_t i = 0; i < nspan; i++) {
            B = (B + (uint64_t)(parts[i].len % 65521u) * A + parts[i].b) % 65521u;
            A = (A + parts[i].a) % 65521u;
        }
        *adler_io = (uint32_t)((B << 16) | A);
    }
}

static int checksum_dev(zngamd_ctx *c, const uint8_t *d, uint64_t n, uint32_t *crc_io, uint32_t *adler_io)
{
    if (n == 0) return ZNGAMD_OK;
    PhaseClock pc(c, "checksum of a device buffer");
    std::vector<ZaCkPart> parts;
    const int r = checksum_launch(c, d, n, crc_io != nullptr, adler_io != nullptr, parts);
    if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    checksum_fold(parts, crc_io, adler_io);
    return ZNGAMD_OK;
}

// a grow-only pinned buffer (uploads from pinned memory are queued and return; from pageable memory the call stages them itself
// and waits -- 15 to 30 us per copy of a small call)
static int pinned_ensure(zngamd_ctx *c, uint8_t **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return ZNGAMD_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(c, hipHostMalloc((void **)p, want, hipHostMallocDefault));
    *cap = want;
    return ZNGAMD_OK;
}

#define ZNGAMD_SMALL_UP (1u << 20)       // inputs up to this size travel through the pinned upload buffer (one queued copy, the 64 zero bytes behind the input included)
static int stage_in(zngamd_ctx *c, const uint8_t *in, uint64_t n, uint64_t pad_front = 0)
{
    PhaseClock pc(c, "stage_in (host -> device)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->st_in.ensure(pad_front + n + 64));
    if (n <= ZNGAMD_SMALL_UP) {
        // (nearly every call ends behind a synchronisation of the stream, but one that failed half way may have left its copy queued)
        if (!c->ev_up) HIPCHK(c, hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
        if (c->up_busy) { HIPCHK(c, hipEventSynchronize(c->ev_up)); c->up_busy = false; }
        const int r = pinned_ensure(c, &c->h_up, &c->h_up_cap, n + 64);
        if (r) return r;
        if (n) memcpy(c->h_up, in, n);
        memset(c->h_up + n, 0, 64);
        HIPCHK(c, hipMemcpyAsync(c->st_in.p + pad_front, c->h_up, n + 64, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_up, c->stream));
        c->up_busy = true;
        return ZNGAMD_OK;
    }
    if (n) HIPCHK(c, hipMemcpyAsync(c->st_in.p + pad_front, in, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->st_in.p + pad_front + n, 0, 64, c->stream));
    return ZNGAMD_OK;
}

int zngamd_crc32_dev(zngamd_ctx *c, uint32_t crc, const void *dbuf, size_t len, uint32_t *out)
try {
    if (!c || !out) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t v = crc;
    int r = checksum_dev(c, (const uint8_t *)dbuf, len, &v, nullptr);
    *out = v;
    return r;
} ZA_ABI_GUARD

int zngamd_crc32(zngamd_ctx *c, uint32_t crc, const uint8_t *buf, size_t len, uint32_t *out)
try {
    if (!c || !out || (!buf && len)) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    int r = stage_in(c, buf, len);
    if (r) return r;
    uint32_t v = crc;
    r = checksum_dev(c, c->st_in.p, len, &v, nullptr);
    *out = v;
    return r;
} ZA_ABI_GUARD

int zngamd_adler32(zngamd_ctx *c, uint32_t adler, const uint8_t *buf, size_t len, uint32_t *out)
try {
    if (!c || !out || (!buf && len)) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    int r = stage_in(c, buf, len);
    if (r) return r;
    // the seed halves are reduced even when there is no data (zlib semantics the reference inherits)
    uint32_t v = ((((adler >> 16) & 0xFFFFu) % 65521u) << 16) | ((adler & 0xFFFFu) % 65521u);
    r = checksum_dev(c, c->st_in.p, len, nullptr, &v);
    *out = v;
    return r;
} ZA_ABI_GUARD

// ---------------------------------------------------------------------------------------------
// deflate
// ---------------------------------------------------------------------------------------------
int zngamd_level_ok(int level) { return level >= -1 && level <= 9; }

static uint32_t unit_size_of(const zngamd_block &b) { return (b.flags & ZNGAMD_FLAG_UNITS16K) ? ZA_SMALL_UNIT : (uint32_t)ZA_MAX_UNIT; }
static uint32_t units_of(const zngamd_block &b) { const uint64_t U = unit_size_of(b); return b.len == 0 ? 1u : (uint32_t)(((uint64_t)b.len + U - 1) / U); }

uint32_t zngamd_count_units(const zngamd_block *blocks, uint32_t n_blocks)
{
    uint64_t t = 0;
    for (uint32_t i = 0; i < n_blocks; i++) t += units_of(blocks[i]);
    return t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}

static int build_units(zngamd_ctx *c, const zngamd_block *blocks, uint32_t n_blocks, uint64_t in_len, std::vector<ZaUnit> &hu)
{
    hu.clear();
    for (uint32_t b = 0; b < n_blocks; b++) {
        const zngamd_block &B = blocks[b];
        if (B.dict_len > ZA_WIN || B.dict_len > B.off || B.off + B.len > in_len) return fail(c, ZNGAMD_E_ARG, "block outside the input buffer");
        const uint32_t nu = units_of(B);
        const uint64_t U = unit_size_of(B);
        for (uint32_t k = 0; k < nu; k++) {
            ZaUnit u;
            const uint64_t rel = (uint64_t)k * U;
            u.in_off = B.off + rel;
            u.in_len = (uint32_t)std::min<uint64_t>(U, B.len - rel);
            u.dict_len = (uint32_t)std::min<uint64_t>(ZA_WIN, (uint64_t)B.dict_len + rel);
            u.flags = (B.flags & (ZNGAMD_FLAG_FLATHDR | ZNGAMD_FLAG_SEG2K)) | ((k == nu - 1) ? (B.flags & ZNGAMD_FLAG_FINAL) : 0u);
            u.flags |= (uint32_t)za_seg_shift_for(u.in_len, u.flags) << 8;       // the unit's segment size (za_common.h: small units take small segments)
            u.block = b;
            // the unit's whole 32 KiB dictionary is the tail of the unit in front of it: the chain tables may be carried over
            if (!hu.empty()) {
                const ZaUnit &pv = hu.back();
                if (u.dict_len == ZA_WIN && pv.in_len >= ZA_WIN && (pv.in_len & 3u) == 0u && pv.in_off + pv.in_len == u.in_off) u.flags |= ZA_FLAG_CARRY;   // (a multiple of 4: the search stages its byte ring in aligned dwords)
            }
            hu.push_back(u);
        }
    }
    return ZNGAMD_OK;
}

// launch the five deflate kernels over all units (in chunks that bound the workspace)
// `packed` set: no slots -- the plan kernel sizes every unit exactly, a prefix sum places it, the packer writes it there
struct PackedDst { uint8_t *d_dst = nullptr; uint64_t cap = 0; uint64_t *d_unit_off = nullptr;
                   uint64_t *d_total = nullptr; uint32_t *d_status = nullptr; };      // (optional) where the stream's size and the units' pack status go: a caller that fetches all results with one copy
static int deflate_units_dev(zngamd_ctx *c, const uint8_t *d_in, uint64_t in_len, const std::vector<ZaUnit> &hu, int level,
                             uint8_t *d_slots, uint32_t *d_unit_len, uint32_t *d_unit_crc, int max_dist = ZA_WIN,
                             const PackedDst *packed = nullptr)
{
    if (level == -1) level = 6;
    if (level < 0 || level > 9) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    const uint32_t n = (uint32_t)hu.size();
    if (n == 0) return ZNGAMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t ch = std::min(n, c->chunk_units);
    const size_t ntab = ZA_LEVELS[level].use_c ? 3 : 2;
    if (level > 0) {
        // a chunk's workspaces must fit what the device has free (plus what these buffers hold already): halve it until they do
        // (a hipMalloc failure further down is still a hard error, but no longer the first thing a smaller device or a second
        // context on this one meets)
        const size_t held = c->links.cap * 2 + c->best.cap * 4 + c->tok.cap * 4 + c->best_keep.cap * 4;
        size_t free_b = 0, total_b = 0;
        if (ch > 64 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {       // (a driver call: tens of microseconds, and a small call has nothing to halve)
            auto need = [&](uint32_t k) { return (size_t)k * (ntab * ZA_PREV_STRIDE * 2 + ZA_BEST_STRIDE * 4 + (c->debug_keep ? (ZA_BEST_STRIDE + ZA_TOK_STRIDE) * 4 : 0) + 8192); };
            while (ch > 64 && need(ch) > held && need(ch) - held > free_b - free_b / 16) ch = (ch + 1) / 2;
        }
    }
    HIPCHK(c, c->units.ensure(n)); HIPCHK(c, c->segbits.ensure((size_t)n * ZA_SEGB_STRIDE)); HIPCHK(c, c->cidx.ensure((size_t)n * ZA_CIDX_STRIDE)); HIPCHK(c, c->status.ensure(n));
    if (level > 0) {
        const size_t tab = (size_t)ch * ZA_PREV_STRIDE + 8;              // entries of one link table (+ slack for the search's four-link loads)
        HIPCHK(c, c->links.ensure(ntab * tab)); HIPCHK(c, c->best.ensure((size_t)ch * ZA_BEST_STRIDE));
        c->prev_p = c->links.p; c->linkb_p = c->links.p + tab; c->linkc_p = ntab > 2 ? c->links.p + 2 * tab : nullptr;
        if (ZA_LEVELS[level].dp) HIPCHK(c, c->dpcost.ensure((size_t)ch * ZA_DP_COSTS));
        if (c->debug_keep) {
            HIPCHK(c, c->best_keep.ensure((size_t)ch * ZA_BEST_STRIDE));
            HIPCHK(c, c->tok.ensure((size_t)ch * ZA_TOK_STRIDE));
            c->tok_p = c->tok.p;
        } else {
            // the token words (and the dynamic programme's acc[] scratch inside them) take the place of the link tables: written by
            // kernels that run after the search of the same chunk on the same stream, 512 KiB a unit inside >= 640 KiB a unit
            static_assert(2 * ZA_PREV_STRIDE * 2 >= ZA_TOK_STRIDE * 4, "token words fit two link tables");
            c->tok_p = (uint32_t *)c->links.p;
        }
    }
    HIPCHK(c, c->segtok.ensure((size_t)ch * ZA_MAX_SEGS)); HIPCHK(c, c->hist.ensure((size_t)ch * ZA_HIST_STRIDE));
    HIPCHK(c, c->codes.ensure((size_t)ch * ZA_CODE_STRIDE)); HIPCHK(c, c->plan.ensure(ch));
    uint64_t *d_run_total = (uint64_t *)((uint8_t *)c->d_small + 224);      // (packed) bytes of the launches so far
    uint64_t *d_offs = nullptr;
    if (packed) {
        HIPCHK(c, c->hdr.ensure((size_t)ch * ZA_HDR_STRIDE));
        d_offs = packed->d_unit_off;
        if (!d_offs) { HIPCHK(c, c->st_off.ensure(n)); d_offs = c->st_off.p; }
        if (packed->d_total) d_run_total = packed->d_total;
        // (no memset of the running total: the first launch's prefix sum starts from nothing and writes it)
    }
    uint32_t *d_status = (packed && packed->d_status) ? packed->d_status : c->status.p;
    // Runs of the chain kernel: one workgroup walks a run of consecutive units and carries its tables from unit to unit where
    // the next unit's dictionary is the tail of the one before (ZA_FLAG_CARRY); a run costs its first unit's dictionary
    // again, so longer runs save more (a quarter of the positions at most) -- but workgroups are handed out in order, and a
    // launch ends with its last workgroup: runs of L units for the first four fifths of a launch, short ones behind them to
    // fill the tail, and L bounded by what keeps every slot of the device busy at least twice.  Runs never cross a launch.
    const bool same_plan = c->plan_ch == ch && c->plan_in.size() == hu.size() && c->last_hu.size() == hu.size() &&
                           memcmp(c->plan_in.data(), hu.data(), hu.size() * sizeof(ZaUnit)) == 0;
    if (!same_plan) {
        // (the kept plan describes device tables that are about to be overwritten: should an upload fail half way, a later call
        // with the old shape must not find it valid)
        c->plan_in.clear(); c->plan_ch = 0; c->last_hu.clear();
        std::vector<ZaUnit> hv(hu);
        std::vector<uint32_t> run_start;
        for (uint32_t c0 = 0; c0 < n; c0 += ch) {
            const uint32_t m = std::min(ch, n - c0);
            const uint32_t L = c->chain_run ? c->chain_run : std::min<uint32_t>(8u, std::max<uint32_t>(1u, m / (2u * c->chain_slots)));
            const uint32_t Ls = c->chain_run ? c->chain_run : std::max<uint32_t>(1u, L / 4u);
            const uint32_t big_end = c->chain_run ? m : (uint32_t)((uint64_t)m * 4 / 5 / L * L);
            uint32_t next_cut = 0;
            for (uint32_t i = 0; i < m; i++) {
                ZaUnit &u = hv[c0 + i];
                const bool cut = i == next_cut || !(u.flags & ZA_FLAG_CARRY);
                if (cut) { u.flags |= ZA_FLAG_RUNHEAD; run_start.push_back(i); next_cut = i + (i < big_end ? L : Ls); }
            }
            run_start.push_back(m);                                  // (the runs of one launch: starts relative to the launch, then its end)
        }
        HIPCHK(c, c->runs.ensure(run_start.size()));
        if (n <= 4096) {
            // a small table travels from pinned memory: two queued copies and no wait (r06: a call whose shape differs from the
            // last one's paid 40 us for the two staged copies and the synchronisation)
            const size_t rb = run_start.size() * 4, ub = (size_t)n * sizeof(ZaUnit);
            if (!c->ev_tab) HIPCHK(c, hipEventCreateWithFlags(&c->ev_tab, hipEventDisableTiming));
            if (c->tab_busy) { HIPCHK(c, hipEventSynchronize(c->ev_tab)); c->tab_busy = false; }
            const int rp = pinned_ensure(c, &c->h_tab, &c->h_tab_cap, rb + ub);
            if (rp) return rp;
            memcpy(c->h_tab, hv.data(), ub);
            memcpy(c->h_tab + ub, run_start.data(), rb);
            HIPCHK(c, hipMemcpyAsync(c->units.p, c->h_tab, ub, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->runs.p, c->h_tab + ub, rb, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipEventRecord(c->ev_tab, c->stream));
            c->tab_busy = true;
        } else {
        HIPCHK(c, hipMemcpyAsync(c->runs.p, run_start.data(), run_start.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->units.p, hv.data(), (size_t)n * sizeof(ZaUnit), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));                  // (hv and run_start are locals)
        }
        c->last_hu.swap(hv);
        c->plan_runs.swap(run_start);
        c->plan_in = hu; c->plan_ch = ch;
    }
    const std::vector<uint32_t> &run_start = c->plan_runs;
    // (no memset of the slots: the pack kernel zeroes the few words it merges with atomic OR and writes the rest whole)
    ZaLevel L = ZA_LEVELS[level];
    L.max_dist = (max_dist < 1 || max_dist > ZA_WIN) ? ZA_WIN : max_dist;
    size_t run_pos = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += ch) {
        const uint32_t m = std::min(ch, n - c0);
        const ZaUnit *du = c->units.p + c0;
        uint32_t nruns = 0;
        while (run_start[run_pos + nruns] != m) nruns++;
        const uint32_t *d_runs = c->runs.p + run_pos;
        run_pos += nruns + 1;
        if (level > 0) {
            { ProfScope ps(c, ZNGAMD_K_CHAINS);
              hipLaunchKernelGGL(za_k_chains<ZA_TABLE_A>, dim3(nruns), dim3(256), 0, c->stream, d_in, du, d_runs, c->prev_p, L.dp ? c->dpcost.p : (uint32_t *)nullptr);
              hipLaunchKernelGGL(za_k_chains<ZA_TABLE_B>, dim3(nruns), dim3(256), 0, c->stream, d_in, du, d_runs, c->linkb_p, (uint32_t *)nullptr);
              if (L.use_c) hipLaunchKernelGGL(za_k_chains<ZA_TABLE_C>, dim3(nruns), dim3(256), 0, c->stream, d_in, du, d_runs, c->linkc_p, (uint32_t *)nullptr); }
            { ProfScope ps(c, ZNGAMD_K_SEARCH);
#define ZA_LAUNCH_SEARCH(...) hipLaunchKernelGGL((za_k_search<__VA_ARGS__>), dim3(nruns), dim3(ZA_SEARCH_THREADS), 0, c->stream, d_in, in_len, du, d_runs, \
                                                 c->prev_p, c->linkb_p, L.use_c ? c->linkc_p : c->linkb_p, c->best.p, c->dpcost.p, L)
              if (L.cap > 16) { if (L.use_c) ZA_LAUNCH_SEARCH(true, 0, true); else ZA_LAUNCH_SEARCH(true, 0, false); }
              else if (L.chain == 1) { if (L.use_c) ZA_LAUNCH_SEARCH(false, 1, true); else ZA_LAUNCH_SEARCH(false, 1, false); }
              else if (L.chain == 2) { if (L.use_c) ZA_LAUNCH_SEARCH(false, 2, true); else ZA_LAUNCH_SEARCH(false, 2, false); }
              else if (L.chain == 3) { if (L.use_c) ZA_LAUNCH_SEARCH(false, 3, true); else ZA_LAUNCH_SEARCH(false, 3, false); }
              else { if (L.use_c) ZA_LAUNCH_SEARCH(false, 0, true); else ZA_LAUNCH_SEARCH(false, 0, false); }
#undef ZA_LAUNCH_SEARCH
            }
            if (L.dp) {
                if (c->debug_keep) HIPCHK(c, hipMemcpyAsync(c->best_keep.p, c->best.p, (size_t)m * ZA_BEST_STRIDE * 4, hipMemcpyDeviceToDevice, c->stream));
                ProfScope ps(c, ZNGAMD_K_OPTPARSE);
                // (levels 4-6: the search took the statistics itself)
                if (L.cap > 16 || !ZA_STATS_FOLD) hipLaunchKernelGGL(za_k_dpstats, dim3(m), dim3(256), 0, c->stream, du, c->best.p, c->dpcost.p);
                hipLaunchKernelGGL(za_k_optparse, dim3(m), dim3(64), 0, c->stream, du, c->best.p, c->dpcost.p, c->tok_p, L);
            }
            { ProfScope ps(c, ZNGAMD_K_PARSE);
              hipLaunchKernelGGL(za_k_parse, dim3(m), dim3(64), 0, c->stream, du, c->best.p, c->tok_p, c->segtok.p, c->hist.p,
                                 d_unit_crc + c0, c->d_crc_table, c->d_x8k, L); }
        } else {       // level 0: stored blocks, nothing to search or parse -- only the units' CRC-32
            ProfScope ps(c, ZNGAMD_K_PARSE);
            hipLaunchKernelGGL(za_k_unit_crc, dim3(m), dim3(64), 0, c->stream, d_in, du, d_unit_crc + c0, c->d_crc_table, c->d_x8k);
        }
        if (packed) {
            { ProfScope ps(c, ZNGAMD_K_PLAN);
              hipLaunchKernelGGL(za_k_plan, dim3(m), dim3(64), 0, c->stream, du, c->hist.p, c->codes.p, c->plan.p,
                                 (uint8_t *)nullptr, 0u, level, c->hdr.p, d_unit_len + c0); }
            { ProfScope ps(c, ZNGAMD_K_GATHER);          // (what is left of the gather: the prefix sum)
              hipLaunchKernelGGL(za_k_offsets, dim3(1), dim3(m <= 64 ? 64 : 1024), 0, c->stream, d_unit_len + c0, m, 0u, 0ull, d_offs + c0, d_run_total,
                                 (const ZaUnit *)nullptr, c0 ? (const uint64_t *)d_run_total : (const uint64_t *)nullptr, 1); }
            { ProfScope ps(c, ZNGAMD_K_PACK);
              hipLaunchKernelGGL(za_k_pack, dim3(m), dim3(64), 0, c->stream, d_in, du, c->tok_p, c->segtok.p, c->codes.p, c->plan.p,
                                 c->segbits.p + (size_t)c0 * ZA_SEGB_STRIDE, c->cidx.p + (size_t)c0 * ZA_CIDX_STRIDE, packed->d_dst,
                                 0u, d_unit_len + c0, d_status + c0, (const uint64_t *)(d_offs + c0), packed->cap, (const uint8_t *)c->hdr.p); }
        } else {
        { ProfScope ps(c, ZNGAMD_K_PLAN);
          hipLaunchKernelGGL(za_k_plan, dim3(m), dim3(64), 0, c->stream, du, c->hist.p, c->codes.p, c->plan.p,
                             d_slots + (size_t)c0 * ZNGAMD_SLOT_STRIDE, (uint32_t)ZNGAMD_SLOT_STRIDE, level, (uint8_t *)nullptr, (uint32_t *)nullptr); }
        { ProfScope ps(c, ZNGAMD_K_PACK);
          hipLaunchKernelGGL(za_k_pack, dim3(m), dim3(64), 0, c->stream, d_in, du, c->tok_p, c->segtok.p, c->codes.p, c->plan.p,
                             c->segbits.p + (size_t)c0 * ZA_SEGB_STRIDE, c->cidx.p + (size_t)c0 * ZA_CIDX_STRIDE, d_slots + (size_t)c0 * ZNGAMD_SLOT_STRIDE,
                             (uint32_t)ZNGAMD_SLOT_STRIDE, d_unit_len + c0, c->status.p + c0, (const uint64_t *)nullptr, 0ull, (const uint8_t *)nullptr); }
        }
        HIPCHK(c, hipGetLastError());
    }
    c->last_units = n; c->last_single_chunk = (n <= ch); c->last_kept = c->debug_keep; c->last_unit_len = d_unit_len; c->last_ulen_on_host = false;
    return ZNGAMD_OK;
}

int zngamd_deflate_blocks_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                              int level, void *d_slots, uint32_t *d_unit_len, uint32_t *d_unit_crc, uint32_t *h_unit_block)
try {
    if (!c || (!blocks && n_blocks) || !d_slots || !d_unit_len || !d_unit_crc) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    // (a caller that compresses batch after batch with the same block table -- the reference's writer does, block size and
    // dictionary rule never change -- pays for cutting the blocks into units once)
    int r = ZNGAMD_OK;
    if (!(n_blocks && c->blocks_in.size() == n_blocks && c->blocks_len == in_len &&
          memcmp(c->blocks_in.data(), blocks, (size_t)n_blocks * sizeof(zngamd_block)) == 0)) {
        c->blocks_in.clear();
        r = build_units(c, blocks, n_blocks, in_len, c->blocks_hu);
        if (r) return r;
        c->blocks_in.assign(blocks, blocks + n_blocks); c->blocks_len = in_len;
    }
    const std::vector<ZaUnit> &hu = c->blocks_hu;
    if (h_unit_block) for (size_t i = 0; i < hu.size(); i++) h_unit_block[i] = hu[i].block;
    r = deflate_units_dev(c, (const uint8_t *)d_in, in_len, hu, level, (uint8_t *)d_slots, d_unit_len, d_unit_crc);
    if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_OK;
} ZA_ABI_GUARD

// The blocks' compressed bytes back to back at d_out (no slots, no gather pass): *total_bytes and, per unit, its size, CRC-32 and
// (optional) offset.  ZNGAMD_BUF_ERROR with *total_bytes = the size needed when out_cap is too small (units that did not fit are
// not written); ZNGAMD_E_OVERFLOW never (a unit has no cap of its own here).
int zngamd_deflate_blocks_packed_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                                     int level, void *d_out, uint64_t out_cap, uint32_t *d_unit_len, uint32_t *d_unit_crc,
                                     uint64_t *d_unit_off, uint64_t *total_bytes)
try {
    if (!c || (!blocks && n_blocks) || !d_out || !d_unit_len || !d_unit_crc || !total_bytes) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    int r = ZNGAMD_OK;
    if (!(n_blocks && c->blocks_in.size() == n_blocks && c->blocks_len == in_len &&
          memcmp(c->blocks_in.data(), blocks, (size_t)n_blocks * sizeof(zngamd_block)) == 0)) {
        c->blocks_in.clear();
        r = build_units(c, blocks, n_blocks, in_len, c->blocks_hu);
        if (r) return r;
        c->blocks_in.assign(blocks, blocks + n_blocks); c->blocks_len = in_len;
    }
    const std::vector<ZaUnit> &hu = c->blocks_hu;
    *total_bytes = 0;
    if (hu.empty()) return ZNGAMD_OK;
    PackedDst pd; pd.d_dst = (uint8_t *)d_out; pd.cap = out_cap; pd.d_unit_off = d_unit_off;
    r = deflate_units_dev(c, (const uint8_t *)d_in, in_len, hu, level, nullptr, d_unit_len, d_unit_crc, ZA_WIN, &pd);
    if (r) return r;
    const uint64_t *d_run_total = (const uint64_t *)((const uint8_t *)c->d_small + 224);
    std::vector<uint32_t> st(hu.size());
    HIPCHK(c, hipMemcpyAsync(total_bytes, d_run_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(st.data(), c->status.p, st.size() * 4ull, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (*total_bytes > out_cap) return fail(c, ZNGAMD_BUF_ERROR, "destination too small");
    // a unit whose packed size is not the planned one would have shifted everything behind it: never expected, always checked
    for (uint32_t v : st) if (v) return fail(c, ZNGAMD_E_HIP, "packed deflate: a unit's size differs from its plan");
    return ZNGAMD_OK;
} ZA_ABI_GUARD

static int gather_dev(zngamd_ctx *c, const uint8_t *d_slots, const uint32_t *d_unit_len, uint32_t n, uint32_t extra,
                      uint8_t *d_dst, uint64_t dst_base, uint64_t dst_cap, uint64_t *d_unit_off, uint64_t *total, bool do_copy,
                      const ZaUnit *d_units_for_index = nullptr)
{
    if (n == 0) { *total = 0; return ZNGAMD_OK; }
    uint64_t *offs = d_unit_off;
    if (!offs) { HIPCHK(c, c->st_off.ensure(n)); offs = c->st_off.p; }
    uint64_t *d_total = (uint64_t *)c->d_small;
    { ProfScope ps(c, ZNGAMD_K_GATHER);
      hipLaunchKernelGGL(za_k_offsets, dim3(1), dim3(1024), 0, c->stream, d_unit_len, n, extra, dst_base, offs, d_total, d_units_for_index); }
    HIPCHK(c, hipMemcpyAsync(total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (dst_base + *total > dst_cap) return fail(c, ZNGAMD_BUF_ERROR, "destination too small");
    if (do_copy) {
        ProfScope ps(c, ZNGAMD_K_GATHER);
        hipLaunchKernelGGL(za_k_gather, dim3(n), dim3(256), 0, c->stream, d_slots, (uint32_t)ZNGAMD_SLOT_STRIDE, d_unit_len, offs, d_dst);
    }
    HIPCHK(c, hipGetLastError());
    return ZNGAMD_OK;
}

int zngamd_gather_dev(zngamd_ctx *c, const void *d_slots, const uint32_t *d_unit_len, uint32_t n_units, void *d_dst,
                      uint64_t dst_base, uint64_t dst_cap, uint64_t *d_unit_off, uint64_t *total_bytes)
try {
    if (!c || !d_slots || !d_unit_len || !d_dst || !total_bytes) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    int r = gather_dev(c, (const uint8_t *)d_slots, d_unit_len, n_units, 0, (uint8_t *)d_dst, dst_base, dst_cap, d_unit_off, total_bytes, true);
    if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_OK;
} ZA_ABI_GUARD

// shared by the two host-buffer entry points: input already staged at st_in.p
// pinned host memory for results that are re-distributed on the host afterwards: a device-to-host copy into pinned memory
// runs at link speed, into fresh pageable memory several times slower
static int host_stage(zngamd_ctx *c, size_t bytes, uint8_t **p)
{
    if (bytes > c->h_stage_cap) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->h_stage_cap = 0;
        const size_t want = bytes + bytes / 4 + (1u << 20);
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage, want, hipHostMallocDefault));
        c->h_stage_cap = want;
    }
    *p = c->h_stage;
    return ZNGAMD_OK;
}

// Host copies of payload: several threads for large pieces.  A result usually lands in memory the process has never touched
// (a Python bytes object made for this call): the first touch of a page costs far more than the copy, one thread reaches
// 12 GB/s there, and the device-to-host DMA itself 56 GB/s (profiles/r03_pcie.txt).
static void par_memcpy(uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t piece = 4u << 20;
    static const long env_threads = [] { const char *e = getenv("ZNGAMD_COPY_THREADS"); return e ? atol(e) : 0L; }();      // (read once, not per piece)
    unsigned hw = std::thread::hardware_concurrency();
    size_t t = std::min<size_t>(std::min<size_t>(hw ? hw / 2 : 4, 8), n / piece);
    if (env_threads >= 1 && env_threads <= 64) t = std::min<size_t>((size_t)env_threads, std::max<size_t>(n / (1u << 20), 1));
    if (t <= 1) { memcpy(dst, src, n); return; }
    const size_t step = ((n / t) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    th.reserve(t);
    size_t started_to = step;                                  // bytes [0, started_to) have a copier (this thread takes the first piece)
    for (size_t i = 1; i < t; i++) {
        const size_t o = i * step;
        if (o >= n) break;
        // a thread that cannot be started (EAGAIN under a pids limit) must not unwind through the joinable ones -- that is
        // std::terminate(), the host process gone: whatever has no copier yet is copied here
        try { th.emplace_back([=] { memcpy(dst + o, src + o, std::min(step, n - o)); }); }
        catch (...) { break; }
        started_to = std::min(n, o + step);
    }
    memcpy(dst, src, std::min(step, n));
    if (started_to < n) memcpy(dst + started_to, src + started_to, n - started_to);
    for (auto &x : th) x.join();
}

// Decoded / compressed payload to the caller's buffer, PIPELINED through two halves of the pinned staging buffer: the DMA of
// piece k + 1 runs while the host threads copy piece k into the caller's (usually fresh) memory.  Copying straight into fresh
// pageable memory makes the driver pin its pages and unpin them when the object is cut to size afterwards -- 25 ms of fixed
// cost for a 3 MiB result, and 12 GB/s at best for a large one.
#define ZNGAMD_D2H_PIECE (16ull << 20)
static int d2h_payload(zngamd_ctx *c, uint8_t *dst, const uint8_t *src_dev, uint64_t n)
{
    if (!n) return ZNGAMD_OK;
    PhaseClock pc(c, "d2h_payload (device -> caller)");
    const uint64_t piece = std::min<uint64_t>(ZNGAMD_D2H_PIECE, n);
    uint8_t *st = nullptr;
    int r = host_stage(c, n <= piece ? n : 2 * piece, &st);
    if (r) return r;
    if (!c->ev_copy[0]) { HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy[0], hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy[1], hipEventDisableTiming)); }
    const uint64_t np = (n + piece - 1) / piece;
    double wait_ms = 0, copy_ms = 0;
    for (uint64_t k = 0; k <= np; k++) {
        if (k < np) {
            const uint64_t o = k * piece, ln = std::min(piece, n - o);
            HIPCHK(c, hipMemcpyAsync(st + (k & 1) * piece, src_dev + o, ln, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->ev_copy[k & 1], c->stream));
        }
        if (k > 0) {
            const uint64_t o = (k - 1) * piece, ln = std::min(piece, n - o);
            const auto t0 = std::chrono::steady_clock::now();
            HIPCHK(c, hipEventSynchronize(c->ev_copy[(k - 1) & 1]));
            const auto t1 = std::chrono::steady_clock::now();
            par_memcpy(dst + o, st + ((k - 1) & 1) * piece, ln);
            if (trace_on()) { wait_ms += std::chrono::duration<double, std::milli>(t1 - t0).count(); copy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(); }
        }
    }
    if (trace_on()) fprintf(stderr, "zng_amd trace:   d2h: waiting for the DMA %.3f ms, host copies %.3f ms (%llu pieces)\n", wait_ms, copy_ms, (unsigned long long)np);
    return ZNGAMD_OK;
}

// (optional) a checksum of the staged input worked out beside the deflate kernels; its partial results travel with theirs
struct CkReq { bool want_crc = false, want_adler = false; std::vector<ZaCkPart> parts; };

static int deflate_host_common(zngamd_ctx *c, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks, int level,
                               std::vector<ZaUnit> &hu, std::vector<uint32_t> &ulen, std::vector<uint32_t> &ucrc,
                               const uint8_t **packed, int max_dist = ZA_WIN,
                               uint8_t *direct_out = nullptr, uint64_t direct_cap = 0, uint64_t *direct_len = nullptr, CkReq *ck = nullptr)
{
    int r;
    { PhaseClock pc(nullptr, "build_units"); r = build_units(c, blocks, n_blocks, in_len, hu); }
    if (r) return r;
    const uint32_t n = (uint32_t)hu.size();
    // Packed: every unit's size is planned before it is packed and the packer writes at its final offset (no slots, no gather).
    // A unit is never larger than its stored form: its bytes + 5 per stored chunk of 65 535 + an empty stored block behind it.
    uint64_t bound = 64;
    for (const ZaUnit &u : hu) bound += (uint64_t)u.in_len + 32u;
    // Everything the host wants back lies in ONE device buffer, the results in front of the stream -- the stream's size, the units'
    // sizes, CRCs and pack status, the checksum kernel's partial sums -- so that a small call fetches results and bytes with one
    // copy (r06: five copies of a few bytes each, staged one after the other, cost a small call 60 us):
    //   [0, 16) size of the stream | [16, ..) n sizes | n CRCs | n status words | checksum parts | (to a multiple of 256) the stream
    const uint64_t nspan = ck ? (in_len + ZA_MAX_UNIT - 1) / ZA_MAX_UNIT : 0;
    const uint64_t o_len = 16, o_crc = o_len + 4ull * n, o_st = o_crc + 4ull * n, o_ck = (o_st + 4ull * n + 15) & ~15ull;
    const uint64_t head = (o_ck + nspan * sizeof(ZaCkPart) + 255) & ~255ull;
    HIPCHK(c, c->st_out.ensure(head + bound));
    uint8_t *d_head = c->st_out.p;
    PackedDst pd; pd.d_dst = d_head + head; pd.cap = bound; pd.d_unit_off = nullptr;
    pd.d_total = (uint64_t *)d_head; pd.d_status = (uint32_t *)(d_head + o_st);
    if (ck && nspan) {
        // the container's checksum: its kernel goes out in front of the deflate kernels and is waited for with them
        ProfScope ps(c, ZNGAMD_K_OTHER);
        hipLaunchKernelGGL(za_k_checksum, dim3((uint32_t)nspan), dim3(256), 0, c->stream, (const uint8_t *)c->st_in.p, in_len, c->d_crc_slice4,
                           (ZaCkPart *)(d_head + o_ck), ck->want_crc ? 1 : 0, ck->want_adler ? 1 : 0);
    }
    { PhaseClock pc(c, "deflate kernels");
      r = deflate_units_dev(c, c->st_in.p, in_len, hu, level, nullptr, (uint32_t *)(d_head + o_len), (uint32_t *)(d_head + o_crc), max_dist, &pd); }
    if (r) return r;
    const bool direct = direct_out != nullptr;
    // One round trip for the results, and for a small stream the bytes as well (its upper bound travels: the size is not known yet)
    const uint64_t EAGER = 512u << 10;
    uint8_t *stage = nullptr;
    const bool eager = bound <= EAGER;
    { int rs = host_stage(c, head + (eager ? bound : 0), &stage); if (rs) return rs; }
    {
        PhaseClock pc(c, "results to the host");
        HIPCHK(c, hipMemcpyAsync(stage, d_head, head + (eager ? bound : 0), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->up_busy = false; c->tab_busy = false;                       // (the stream has drained: the upload buffers are free)
    prof_collect(c);
    uint64_t total = 0;
    memcpy(&total, stage, 8);
    ulen.resize(n); ucrc.resize(n);
    if (n) { memcpy(ulen.data(), stage + o_len, 4ull * n); memcpy(ucrc.data(), stage + o_crc, 4ull * n); }
    c->last_ulen_host = ulen; c->last_ulen_on_host = true;
    const uint32_t *st = (const uint32_t *)(stage + o_st);
    for (uint32_t i = 0; i < n; i++) if (st[i]) return fail(c, ZNGAMD_E_HIP, "packed deflate: a unit's size differs from its plan");
    if (total > bound) return fail(c, ZNGAMD_E_HIP, "packed deflate: stream larger than its bound");
    if (ck) { ck->parts.resize(nspan); if (nspan) memcpy(ck->parts.data(), stage + o_ck, nspan * sizeof(ZaCkPart)); }
    // one-shot callers take the packed stream straight into their buffer (no intermediate copy)
    if (direct) { *direct_len = total; if (total > direct_cap) return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
    if (eager) {
        if (direct) memcpy(direct_out, stage + head, total); else *packed = stage + head;
        return ZNGAMD_OK;
    }
    if (!direct) { int rs = host_stage(c, total, &stage); if (rs) return rs; *packed = stage; }
    if (total && !direct) { HIPCHK(c, hipMemcpyAsync(stage, pd.d_dst, total, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    if (total && direct) { const int rc_ = d2h_payload(c, direct_out, pd.d_dst, total); if (rc_) return rc_; HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return ZNGAMD_OK;
}

int zngamd_deflate_blocks(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                          int level, uint8_t *out, uint64_t out_cap_per_block, uint32_t *out_len, uint32_t *crc)
try {
    if (!c || (!in && in_len) || (!blocks && n_blocks) || !out || !out_len || !crc) return ZNGAMD_E_ARG;
    if (!zngamd_level_ok(level)) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    std::lock_guard<std::mutex> g(c->mu);
    int r = stage_in(c, in, in_len);
    if (r) return r;
    std::vector<ZaUnit> hu; std::vector<uint32_t> ulen, ucrc; const uint8_t *packed = nullptr;
    const uint32_t wb = n_blocks ? (blocks[0].flags >> 8) & 15u : 0u;
    if (wb != 0 && wb < 9) return fail(c, ZNGAMD_STREAM_ERROR, "window bits must be 9..15");
    r = deflate_host_common(c, in_len, blocks, n_blocks, level, hu, ulen, ucrc, &packed, wb ? (1 << wb) : ZA_WIN);
    if (r) return r;
    int ret = ZNGAMD_OK;
    size_t pos = 0, u = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint64_t tot = 0; uint32_t bc = 0; size_t start = pos;
        bool first = true;
        for (; u < hu.size() && hu[u].block == b; u++) {
            tot += ulen[u]; pos += ulen[u];
            bc = first ? ucrc[u] : zngamd_crc32_combine(bc, ucrc[u], hu[u].in_len);
            first = false;
        }
        crc[b] = bc;
        if (tot >= out_cap_per_block) { out_len[b] = 0xFFFFFFFFu; ret = ZNGAMD_E_OVERFLOW; continue; }
        memcpy(out + (size_t)b * out_cap_per_block, packed + start, tot);
        out_len[b] = (uint32_t)tot;
    }
    if (ret) c->err = "Compressed output exceeds buffer size";
    return ret;
} ZA_ABI_GUARD

static int deflate_blocks_packed_locked(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                                        int level, uint8_t *out, uint64_t out_cap, uint64_t block_cap, uint32_t *out_len, uint32_t *crc, uint64_t *total)
{
    *total = 0;
    int r = stage_in(c, in, in_len);
    if (r) return r;
    std::vector<ZaUnit> hu; std::vector<uint32_t> ulen, ucrc; const uint8_t *packed = nullptr;
    const uint32_t wb = n_blocks ? (blocks[0].flags >> 8) & 15u : 0u;
    if (wb != 0 && wb < 9) return fail(c, ZNGAMD_STREAM_ERROR, "window bits must be 9..15");
    r = deflate_host_common(c, in_len, blocks, n_blocks, level, hu, ulen, ucrc, &packed, wb ? (1 << wb) : ZA_WIN, out, out_cap, total);
    if (r == ZNGAMD_BUF_ERROR && !hu.empty()) {
        // the packed stream does not fit: because a block outgrew its cap (an overflow, reported like the per-block form does),
        // or because the caller's buffer is simply too small (ZNGAMD_BUF_ERROR, *total = the size needed)
        // (the units' sizes came back with the results: deflate_host_common fills `ulen` before it looks at the caller's room)
        if (ulen.size() != hu.size()) return r;
        bool any = false;
        size_t u2 = 0;
        for (uint32_t b = 0; b < n_blocks; b++) {
            uint64_t tot = 0;
            for (; u2 < hu.size() && hu[u2].block == b; u2++) tot += ulen[u2];
            crc[b] = 0;
            out_len[b] = tot >= block_cap ? 0xFFFFFFFFu : (uint32_t)tot;
            any = any || tot >= block_cap;
        }
        if (any) { c->err = "Compressed output exceeds buffer size"; return ZNGAMD_E_OVERFLOW; }
    }
    if (r) return r;
    int ret = ZNGAMD_OK;
    size_t u = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint64_t tot = 0; uint32_t bc = 0;
        bool first = true;
        for (; u < hu.size() && hu[u].block == b; u++) {
            tot += ulen[u];
            bc = first ? ucrc[u] : zngamd_crc32_combine(bc, ucrc[u], hu[u].in_len);
            first = false;
        }
        crc[b] = bc;
        if (tot >= block_cap) { out_len[b] = 0xFFFFFFFFu; ret = ZNGAMD_E_OVERFLOW; continue; }
        out_len[b] = (uint32_t)tot;
    }
    if (ret) c->err = "Compressed output exceeds buffer size";
    return ret;
}

int zngamd_deflate_blocks_packed(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                                 int level, uint8_t *out, uint64_t out_cap, uint64_t block_cap, uint32_t *out_len, uint32_t *crc, uint64_t *total)
try {
    if (!c || (!in && in_len) || (!blocks && n_blocks) || !out || !out_len || !crc || !total) return ZNGAMD_E_ARG;
    if (!zngamd_level_ok(level)) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    std::lock_guard<std::mutex> g(c->mu);
    return deflate_blocks_packed_locked(c, in, in_len, blocks, n_blocks, level, out, out_cap, block_cap, out_len, crc, total);
} ZA_ABI_GUARD

static int deflate_index_locked(zngamd_ctx *c, uint32_t n_units, uint32_t *unit_in_len, uint32_t *unit_out_len, uint32_t *rows);
// The packed call and the segment index of ITS units in one: a context shared by several writer threads would otherwise have to
// keep them from slipping a deflate call of their own between a thread's two calls (zngamd_deflate_index answers for the
// context's last deflate call, whoever made it).
int zngamd_deflate_blocks_packed_indexed(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, const zngamd_block *blocks, uint32_t n_blocks,
                                         int level, uint8_t *out, uint64_t out_cap, uint64_t block_cap, uint32_t *out_len, uint32_t *crc, uint64_t *total,
                                         uint32_t n_units, uint32_t *unit_in_len, uint32_t *unit_out_len, uint32_t *rows)
try {
    if (!c || (!in && in_len) || (!blocks && n_blocks) || !out || !out_len || !crc || !total || !unit_in_len || !unit_out_len || !rows) return ZNGAMD_E_ARG;
    if (!zngamd_level_ok(level)) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    if (n_units != zngamd_count_units(blocks, n_blocks)) return fail(c, ZNGAMD_E_ARG, "n_units is not what zngamd_count_units says for these blocks");
    std::lock_guard<std::mutex> g(c->mu);
    const int r = deflate_blocks_packed_locked(c, in, in_len, blocks, n_blocks, level, out, out_cap, block_cap, out_len, crc, total);
    if (r) return r;
    return deflate_index_locked(c, n_units, unit_in_len, unit_out_len, rows);
} ZA_ABI_GUARD

int zngamd_deflate_stream(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, int level, int window_bits, uint8_t *out,
                          uint64_t out_cap, uint64_t *out_len, uint32_t *crc, uint32_t *adler)
try {
    if (!c || (!in && in_len) || !out || !out_len) return ZNGAMD_E_ARG;
    if (!zngamd_level_ok(level)) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    if (in_len > 0xFFFFFFFFull) return fail(c, ZNGAMD_E_ARG, "one-shot input limited to 4 GiB - 1");
    std::lock_guard<std::mutex> g(c->mu);
    int r = stage_in(c, in, in_len);
    if (r) return r;
    zngamd_block B; B.off = 0; B.len = (uint32_t)in_len; B.dict_len = 0; B.flags = ZNGAMD_FLAG_FINAL; B.reserved = 0;
    // a call of up to one unit's size is cut into units of 16 KiB (oracle: za_o_deflate_stream): eight wavefronts instead of one
    if (in_len <= ZA_MAX_UNIT) B.flags |= ZNGAMD_FLAG_UNITS16K;
    std::vector<ZaUnit> hu; std::vector<uint32_t> ulen, ucrc; const uint8_t *packed = nullptr;
    if (window_bits < 9 || window_bits > 15) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    // the zlib container's Adler-32: its kernel goes out in front of the deflate kernels and its sums come back with their results
    CkReq ck; ck.want_adler = adler != nullptr;
    r = deflate_host_common(c, in_len, &B, 1, level, hu, ulen, ucrc, &packed, 1 << window_bits, out, out_cap, out_len, adler ? &ck : nullptr);
    if (r) { (void)hipStreamSynchronize(c->stream); return r; }
    if (crc) { uint32_t v = 0; for (size_t u = 0; u < hu.size(); u++) v = u ? zngamd_crc32_combine(v, ucrc[u], hu[u].in_len) : ucrc[u]; *crc = v; }
    if (adler) { uint32_t a = 1; checksum_fold(ck.parts, nullptr, &a); *adler = a; }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_debug_keep(zngamd_ctx *c, int on)
try {
    if (!c) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->debug_keep = on != 0;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_debug_fetch(zngamd_ctx *c, int what, uint32_t unit, void *dst, size_t bytes)
try {
    if (!c || !dst) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (unit >= c->last_units || !c->last_single_chunk) return fail(c, ZNGAMD_E_ARG, "unit not resident");
    const void *src = nullptr; size_t lim = 0;
    switch (what) {
    case 0: src = c->prev_p ? c->prev_p + (size_t)unit * ZA_PREV_STRIDE : nullptr; lim = ZA_PREV_STRIDE * 2ull; break;
    case 1: src = c->best.p + (size_t)unit * ZA_BEST_STRIDE; lim = ZA_BEST_STRIDE * 4ull; break;
    case 2: src = c->tok_p ? c->tok_p + (size_t)unit * ZA_TOK_STRIDE : nullptr; lim = ZA_TOK_STRIDE * 4ull; break;
    case 3: src = c->segtok.p + (size_t)unit * ZA_MAX_SEGS; lim = ZA_MAX_SEGS * 4ull; break;
    case 4: src = c->hist.p + (size_t)unit * ZA_HIST_STRIDE; lim = ZA_HIST_STRIDE * 4ull; break;
    case 5: src = c->codes.p + (size_t)unit * ZA_CODE_STRIDE; lim = ZA_CODE_STRIDE * 4ull; break;
    case 6: src = c->segbits.p + (size_t)unit * ZA_SEGB_STRIDE; lim = ZA_SEGB_STRIDE * 4ull; break;
    case 7: src = c->plan.p + unit; lim = sizeof(ZaPlan); break;
    case 8: src = c->cidx.p + (size_t)unit * ZA_CIDX_STRIDE; lim = ZA_CIDX_STRIDE * 4ull; break;
    case 9: src = c->linkb_p ? c->linkb_p + (size_t)unit * ZA_PREV_STRIDE : nullptr; lim = ZA_PREV_STRIDE * 2ull; break;
    case 10: src = c->linkc_p ? c->linkc_p + (size_t)unit * ZA_PREV_STRIDE : nullptr; lim = ZA_PREV_STRIDE * 2ull; break;
    case 11: src = c->best_keep.p ? c->best_keep.p + (size_t)unit * ZA_BEST_STRIDE : nullptr; lim = ZA_BEST_STRIDE * 4ull; break;
    case 12: src = c->dpcost.p ? c->dpcost.p + (size_t)unit * ZA_DP_COSTS : nullptr; lim = ZA_DP_COSTS * 4ull; break;
    default: return fail(c, ZNGAMD_E_ARG, "unknown stage");
    }
    if (!src || bytes > lim) return fail(c, ZNGAMD_E_ARG, "stage not available");
    // (without zngamd_debug_keep the token words are written over the link tables: only one of the two survives a call)
    if (!c->last_kept && (what == 0 || what == 9 || what == 10)) return fail(c, ZNGAMD_E_ARG, "link tables are kept only after zngamd_debug_keep(1): the token words take their place");
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    if ((what == 0 || what == 9 || what == 10) && unit < c->last_hu.size() && (c->last_hu[unit].flags & ZA_FLAG_CARRY) && !(c->last_hu[unit].flags & ZA_FLAG_RUNHEAD) && unit > 0) {
        // a unit whose chain tables were carried over: the links of its dictionary are the links of the last 32 KiB of the unit in
        // front of it (that unit's row) -- all but the last few (context length - 1), which this unit inserted itself; links that
        // reach in front of the dictionary are "no link" for this unit
        const ZaUnit &u = c->last_hu[unit], &pv = c->last_hu[unit - 1];
        const uint16_t *tab = what == 0 ? c->prev_p : what == 9 ? c->linkb_p : c->linkc_p;
        const size_t late = (what == 0 ? ZA_HASH_BYTES_A : what == 9 ? ZA_HASH_BYTES_B : ZA_HASH_BYTES_C) - 1;
        const size_t nd = std::min<size_t>(bytes / 2, u.dict_len - late);
        uint16_t *d16 = (uint16_t *)dst;
        HIPCHK(c, hipMemcpy(d16, tab + (size_t)(unit - 1) * ZA_PREV_STRIDE + pv.dict_len + pv.in_len - u.dict_len, nd * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nd; i++) if (d16[i] > i) d16[i] = 0;
    }
    if (what == 1 || what == 11) {     // the kernels' entry (distance - 1 | length << 15 | the position's byte << 24) in the documented form len << 16 | dist
        uint32_t *e = (uint32_t *)dst;
        for (size_t i = 0; i < bytes / 4; i++) { const uint32_t lf = ZA_ELEN(e[i]); e[i] = lf ? (lf << 16) | ZA_EDIST(e[i]) : 0u; }
    }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

// ---------------------------------------------------------------------------------------------
// inflate
// ---------------------------------------------------------------------------------------------
static int map_status(int s)
{
    switch (s) {
    case ZA_I_OK: return ZNGAMD_OK;
    case ZA_I_END: return ZNGAMD_STREAM_END;
    case ZA_I_DATA: return ZNGAMD_DATA_ERROR;
    case ZA_I_INPUT: case ZA_I_OUTFULL: return ZNGAMD_BUF_ERROR;
    case ZA_I_CRC: return ZNGAMD_E_GZ_CRC;
    case ZA_I_LENGTH: return ZNGAMD_E_GZ_LENGTH;
    default: return ZNGAMD_DATA_ERROR;
    }
}

// sequential decode of one raw stream that already sits (padded) on the device
static int inflate_serial_dev(zngamd_ctx *c, const uint8_t *d_in, uint64_t in_len, const uint8_t *d_dict, uint32_t dict_len,
                              uint8_t *d_out, uint64_t out_cap, ZaInfResult *hres, uint32_t start_bit = 0)
{
    ZaInfResult *dres = (ZaInfResult *)((uint8_t *)c->d_small + 64);
    { ProfScope ps(c, ZNGAMD_K_INFLATE);
      hipLaunchKernelGGL(za_k_inflate_serial, dim3(1), dim3(ZA_SERIAL_THREADS), 0, c->stream, d_in, in_len, start_bit, d_dict, dict_len, d_out, out_cap, dres); }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hres, dres, sizeof(ZaInfResult), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_OK;
}

// a stream that is resumed at a block header (bit offset + history) and / or may run past the end of the buffer
struct ChunkOpts { uint32_t start_bit = 0; const uint8_t *d_dict = nullptr; uint32_t dict_len = 0; bool allow_cut = false; };
struct ChunkInfo { bool cut = false; bool ended = false; uint64_t end_bit = 0; };
// from this many chunks (or block candidates) on, the chunk kernels run in their small-LDS size: more than the 1 024
// wavefronts the large size keeps resident (256 CUs x 4 workgroups)
#ifndef ZNGAMD_CHUNKS_SMALL_FROM
#define ZNGAMD_CHUNKS_SMALL_FROM 1u             // (r03: with three passes at most and sweeps that follow one another the 512-bit footprint wins at every chunk count -- 512 chunks 3.94 -> 3.38 ms, 1 408 chunks 4.08 -> 3.76 ms; the 1 024-bit one, 4 per CU, stays compiled for comparison: -DZNGAMD_CHUNKS_SMALL_FROM=1536u)
#endif
#ifndef ZNGAMD_CHUNKS_MANY_FROM
#define ZNGAMD_CHUNKS_MANY_FROM 2049u          // more chunks than the middle footprint holds at once (8 per CU x 256): the smallest footprint (384-bit sub-sequences, queue of 768, 256 symbols of ring, decode tables of 9 / 8 index bits: 16 per CU) wins -- 320 MiB of this engine's stream 7.1 -> 4.9 ms, 1 GiB 15.2 -> 10.4 ms
#endif
#ifndef ZA_CHUNK_Q_M
#define ZA_CHUNK_Q_M 768               // the marker decoder where chunks outnumber the wavefronts: queue entries, symbols of history in LDS (768 / 256: 16 per CU, 1 GiB 12.4 -> 10.4 ms; 768 / 512 11.8, 1 024 / 256 12.1)
#endif
#ifndef ZA_CHUNK_RING_M
#define ZA_CHUNK_RING_M 256
#endif
#ifndef ZA_CHUNK_LB_2P
#define ZA_CHUNK_LB_2P ZA_LUT_L_BITS     // table index bits of the marker decoder behind a count pass (streams of other writers)
#define ZA_CHUNK_DB_2P ZA_LUT_D_BITS
#endif
#ifndef ZA_CHUNK_BITS_S
#define ZA_CHUNK_BITS_S 512          // the marker decoder where chunks are many: bits per sub-sequence, queue entries, symbols of history in LDS
#endif
#ifndef ZA_CHUNK_Q_S
#define ZA_CHUNK_Q_S 1536
#endif
#ifndef ZA_CHUNK_RING_S
#define ZA_CHUNK_RING_S 1024
#endif
static int inflate_chunked_dev(zngamd_ctx *c, const uint8_t *d_def, uint64_t avail, uint8_t *d_out, uint64_t out_room,
                               uint64_t *out_len, uint64_t *in_used, const ChunkOpts &o = ChunkOpts(), ChunkInfo *info = nullptr);
static int inflate_chunked_once(zngamd_ctx *c, const uint8_t *d_def, uint64_t avail, uint8_t *d_out, uint64_t out_room,
                                uint64_t *out_len, uint64_t *in_used, const ChunkOpts &o, ChunkInfo *info);

int zngamd_inflate_raw(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, const uint8_t *dict, uint32_t dict_len,
                       uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint64_t *in_used, uint32_t *crc, uint32_t *adler)
try {
    if (!c || (!in && in_len) || (!out && out_cap) || !out_len) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (dict_len > ZA_WIN) { dict += dict_len - ZA_WIN; dict_len = ZA_WIN; }
    // staging layout: [dictionary, padded to 64][stream][64 zero bytes]
    const uint64_t front = (dict_len + 63u) & ~63ull;
    int r = stage_in(c, in, in_len, front);
    if (r) return r;
    if (dict_len) HIPCHK(c, hipMemcpyAsync(c->st_in.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
    if (in_len < (1u << 16) && out_cap <= (512u << 10)) {
        // A small call (r06): the decoder, the checksum kernel -- which takes the output's length from where the decoder left it --
        // and ONE copy that brings result, checksum parts and the bytes up to the caller's limit, behind one synchronisation (the
        // result, the parts and the payload each had a round trip of their own: 60 of a 1 KiB call's 330 us).
        const uint64_t head = 256;                                   // [0, 40) the decoder's result | [64, 128) checksum parts of up to four spans | the output
        const uint64_t nspan = (crc || adler) ? (out_cap + ZA_MAX_UNIT - 1) / ZA_MAX_UNIT : 0;
        HIPCHK(c, c->st_out.ensure(head + out_cap + 64));
        ZaInfResult *dres = (ZaInfResult *)c->st_out.p;
        // (the output is assembled in LDS by za_k_inflate_serial_small; a stream that outgrows the image -- ZA_SMALL_IMG bytes --
        // comes back as "output full" at exactly that size and goes to the ordinary kernel: second == true)
        static const bool no_img = getenv("ZNGAMD_NO_SMALL_IMAGE") != nullptr;
        for (int second = no_img ? 1 : 0; second < 2; second++) {
        { ProfScope ps(c, ZNGAMD_K_INFLATE);
          if (second) hipLaunchKernelGGL(za_k_inflate_serial, dim3(1), dim3(ZA_SERIAL_THREADS), 0, c->stream, (const uint8_t *)(c->st_in.p + front), in_len, 0u, (const uint8_t *)c->st_in.p, dict_len,
                             c->st_out.p + head, out_cap, dres);
          else hipLaunchKernelGGL(za_k_inflate_serial_small, dim3(1), dim3(128), 0, c->stream, (const uint8_t *)(c->st_in.p + front), in_len, 0u, (const uint8_t *)c->st_in.p, dict_len,
                             c->st_out.p + head, out_cap, dres); }
        if (nspan) {
            ProfScope ps(c, ZNGAMD_K_OTHER);
            hipLaunchKernelGGL(za_k_checksum, dim3((uint32_t)nspan), dim3(256), 0, c->stream, (const uint8_t *)(c->st_out.p + head), out_cap, c->d_crc_slice4,
                               (ZaCkPart *)(c->st_out.p + 64), crc ? 1 : 0, adler ? 1 : 0, (const uint64_t *)&dres->out_len);
        }
        HIPCHK(c, hipGetLastError());
        uint8_t *stage = nullptr;
        r = host_stage(c, head + out_cap, &stage);
        if (r) return r;
        HIPCHK(c, hipMemcpyAsync(stage, c->st_out.p, head + out_cap, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->up_busy = false;
        prof_collect(c);
        ZaInfResult res;
        memcpy(&res, stage, sizeof res);
        if (!second && res.status == ZA_I_OUTFULL && res.out_len == (uint64_t)ZA_SMALL_IMG && out_cap > (uint64_t)ZA_SMALL_IMG) continue;      // the image was full, the caller's buffer is not
        c->paths[ZNGAMD_PATH_SEQUENTIAL]++;
        *out_len = res.out_len;
        if (in_used) *in_used = (res.in_bits + 7) >> 3;
        if (nspan) {
            std::vector<ZaCkPart> parts(nspan);
            memcpy(parts.data(), stage + 64, nspan * sizeof(ZaCkPart));
            uint32_t cv = 0, av = 1;
            checksum_fold(parts, crc ? &cv : nullptr, adler ? &av : nullptr);
            if (crc) *crc = cv;
            if (adler) *adler = av;
        }
        if (res.out_len) memcpy(out, stage + head, res.out_len <= out_cap ? res.out_len : out_cap);
        if (res.status == ZA_I_DATA) c->err = "invalid deflate data";
        return map_status(res.status);
        }
        return fail(c, ZNGAMD_E_HIP, "small inflate: no result");       // (not reached: the second pass always returns)
    }
    HIPCHK(c, c->st_out.ensure(out_cap + 64));
    ZaInfResult res;
    bool chunked = false;
    if (dict_len == 0) {
        // large complete streams: chunk-parallel (sync-flush points / dynamic block headers), see zngamd_gunzip
        uint64_t clen = 0, cused = 0;
        const int cr = inflate_chunked_dev(c, c->st_in.p + front, in_len, c->st_out.p, out_cap, &clen, &cused);
        if (cr < 0 && cr != ZNGAMD_BUF_ERROR) return cr;
        if (cr == ZNGAMD_BUF_ERROR) { *out_len = clen; if (in_used) *in_used = 0; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
        if (cr == 0) { chunked = true; res.status = ZA_I_END; res.out_len = clen; res.in_bits = cused * 8; res.block_bits = 0; res.block_out = 0; }
    }
    if (!chunked) {
        r = inflate_serial_dev(c, c->st_in.p + front, in_len, c->st_in.p, dict_len, c->st_out.p, out_cap, &res);
        if (r) return r;
    }
    c->paths[chunked ? ZNGAMD_PATH_CHUNKED : ZNGAMD_PATH_SEQUENTIAL]++;
    *out_len = res.out_len;
    if (in_used) *in_used = (res.in_bits + 7) >> 3;
    if (crc || adler) {                       // (before the payload leaves: the kernel is short, the copies keep the host busy for milliseconds)
        uint32_t cv = 0, av = 1;
        r = checksum_dev(c, c->st_out.p, res.out_len, crc ? &cv : nullptr, adler ? &av : nullptr);
        if (r) return r;
        if (crc) *crc = cv;
        if (adler) *adler = av;
    }
    if (res.out_len) { const int rc_ = d2h_payload(c, out, c->st_out.p, res.out_len); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (res.status == ZA_I_DATA) c->err = "invalid deflate data";
    return map_status(res.status);
} ZA_ABI_GUARD


int zngamd_inflate_resume(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, uint32_t start_bit, const uint8_t *dict, uint32_t dict_len,
                          uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint64_t *in_bits, uint64_t *block_bits, uint64_t *block_out)
try {
    if (!c || (!in && in_len) || (!out && out_cap) || !out_len || !in_bits || !block_bits || !block_out || start_bit > 7) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (dict_len > ZA_WIN) { dict += dict_len - ZA_WIN; dict_len = ZA_WIN; }
    const uint64_t front = (dict_len + 63u) & ~63ull;
    int r = stage_in(c, in, in_len, front);
    if (r) return r;
    if (dict_len) HIPCHK(c, hipMemcpyAsync(c->st_in.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->st_out.ensure(out_cap + 64));
    ZaInfResult res;
    bool chunked = false;
    if (in_len >= (1u << 16)) {
        // a large piece of a stream: its complete blocks are decoded chunk-parallel; what follows the last complete block
        // stays for the next call (the checkpoint is exactly that boundary)
        ChunkOpts o; o.start_bit = start_bit; o.d_dict = c->st_in.p; o.dict_len = dict_len; o.allow_cut = true;
        ChunkInfo ci;
        uint64_t clen = 0, cused = 0;
        const int cr = inflate_chunked_dev(c, c->st_in.p + front, in_len, c->st_out.p, out_cap, &clen, &cused, o, &ci);
        if (cr < 0 && cr != ZNGAMD_BUF_ERROR) return cr;
        if (cr == 0) {
            chunked = true;
            res.status = ci.ended ? ZA_I_END : ZA_I_INPUT; res.out_len = clen; res.in_bits = ci.end_bit;
            res.block_bits = ci.end_bit; res.block_out = clen;
        }   // (not enough room, or nothing to gain: the sequential decoder below fills out_cap / gives the verdict)
    }
    if (!chunked) {
        r = inflate_serial_dev(c, c->st_in.p + front, in_len, c->st_in.p, dict_len, c->st_out.p, out_cap, &res, start_bit);
        if (r) return r;
    }
    *out_len = res.out_len; *in_bits = res.in_bits; *block_bits = res.block_bits; *block_out = res.block_out;
    if (res.out_len) { const int rc_ = d2h_payload(c, out, c->st_out.p, res.out_len); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (res.status == ZA_I_OUTFULL) return ZNGAMD_E_OVERFLOW;          // out_cap reached: call again with more room
    if (res.status == ZA_I_DATA) c->err = "invalid deflate data";
    return map_status(res.status);
} ZA_ABI_GUARD

// ---- two-pass reader for indexed members -----------------------------------------------------
// allow_tail: the chain may stop before the end of the buffer (what follows is an incomplete member, or not an indexed
// one); *covered = bytes the chain spans.
static int scan_members_dev(zngamd_ctx *c, const uint8_t *d_in, uint64_t in_len, std::vector<ZaMember> &hm, uint64_t *total_out,
                            bool allow_tail = false, uint64_t *covered = nullptr)
{
    hm.clear(); *total_out = 0;
    if (in_len < ZA_MEMBER_FIXED + 8) return ZNGAMD_E_ARG;
    const uint32_t max_c = (uint32_t)std::min<uint64_t>(in_len / (ZA_MEMBER_FIXED + 8) + 1, 1u << 26);
    HIPCHK(c, c->cands.ensure(max_c));
    uint32_t *d_n = (uint32_t *)((uint8_t *)c->d_small + 128);
    HIPCHK(c, hipMemsetAsync(d_n, 0, 4, c->stream));
    const uint64_t threads = (in_len + 15) / 16;
    { ProfScope ps(c, ZNGAMD_K_SCAN);
      hipLaunchKernelGGL(za_k_scan_members, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, d_in, in_len, c->cands.p, max_c, d_n); }
    HIPCHK(c, hipGetLastError());
    uint32_t n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n == 0 || n > max_c) return ZNGAMD_E_ARG;
    std::vector<ZaCand> hc(n);
    HIPCHK(c, hipMemcpy(hc.data(), c->cands.p, (size_t)n * sizeof(ZaCand), hipMemcpyDeviceToHost));
    // order by offset (the kernel hands the candidates out in the order its atomics happened): no comparison sort of the whole
    // table -- it was a millisecond of every inflate step for 32 768 members -- but a counting pass over buckets of 16 KiB of
    // stream, which hold a member or two, and a sort inside the few buckets that hold more
    {
        const unsigned SH = 14;
        const size_t nbk = (size_t)(in_len >> SH) + 2;
        if (nbk <= 4 * (size_t)n + 1024) {
            std::vector<uint32_t> start(nbk + 1, 0u);
            for (const ZaCand &cd : hc) start[(size_t)(cd.off >> SH) + 1]++;
            for (size_t b = 0; b < nbk; b++) start[b + 1] += start[b];
            std::vector<ZaCand> so(n);
            std::vector<uint32_t> fill(start.begin(), start.end() - 1);
            for (const ZaCand &cd : hc) so[fill[(size_t)(cd.off >> SH)]++] = cd;
            for (size_t b = 0; b < nbk; b++)
                if (start[b + 1] - start[b] > 1)
                    std::sort(so.begin() + start[b], so.begin() + start[b + 1], [](const ZaCand &a, const ZaCand &b2) { return a.off < b2.off; });
            hc.swap(so);
        } else std::sort(hc.begin(), hc.end(), [](const ZaCand &a, const ZaCand &b) { return a.off < b.off; });
    }
    hm.reserve(n);
    // keep the chain that starts at offset 0 and tiles the stream exactly (signature hits inside
    // compressed data are skipped because nothing points at them)
    uint64_t pos = 0, outp = 0;
    size_t i = 0;
    while (pos < in_len) {
        while (i < hc.size() && hc[i].off < pos) i++;
        if (i == hc.size() || hc[i].off != pos) { if (allow_tail && !hm.empty()) break; return ZNGAMD_E_ARG; }
        ZaMember m;
        const uint32_t nchunk = (hc[i].isize + (1u << ZA_CHUNK_SHIFT) - 1u) >> ZA_CHUNK_SHIFT, hdr = ZA_MEMBER_HDR(nchunk);
        m.in_off = pos + hdr; m.in_len = hc[i].size - hdr - 8; m.out_off = outp;
        m.out_len = hc[i].isize; m.crc = 0; m.index_off = hdr - 28; m.nseg = nchunk;
        hm.push_back(m);
        pos += hc[i].size; outp += hc[i].isize;
    }
    if (pos != in_len && !allow_tail) return ZNGAMD_E_ARG;
    if (covered) *covered = pos;
    *total_out = outp;
    return ZNGAMD_OK;
}

int zngamd_gzip_scan_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, zngamd_member *d_members, uint32_t max_members,
                         uint32_t *n_members, uint64_t *total_out)
try {
    if (!c || !d_in || !d_members || !n_members || !total_out) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<ZaMember> hm;
    int r = scan_members_dev(c, (const uint8_t *)d_in, in_len, hm, total_out);
    prof_collect(c);
    if (r) return fail(c, r, "stream is not made of indexed members");
    if (hm.size() > max_members) return fail(c, ZNGAMD_BUF_ERROR, "member table too small");
    *n_members = (uint32_t)hm.size();
    HIPCHK(c, hipMemcpy(d_members, hm.data(), hm.size() * sizeof(ZaMember), hipMemcpyHostToDevice));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

static int inflate_members_dev(zngamd_ctx *c, const uint8_t *d_in, uint64_t in_len, const ZaMember *d_members, uint32_t n,
                               uint8_t *d_out, uint64_t out_cap, int32_t *d_status)
{
    const uint32_t ch = std::min<uint32_t>(n, 32768);          // members per launch: 176 KiB of match queue each (5.8 GB)
    HIPCHK(c, c->matchq.ensure((size_t)ch * 64 * ZA_MATCHQ_PER_SEG));
    for (uint32_t c0 = 0; c0 < n; c0 += ch) {
        const uint32_t m = std::min(ch, n - c0);
        ProfScope ps(c, ZNGAMD_K_INFLATE);
        hipLaunchKernelGGL(za_k_inflate_members, dim3(m), dim3(64), 0, c->stream, d_in, in_len, d_members + c0, d_out, out_cap,
                           c->matchq.p, c->d_crc_slice4, c->d_x8k, d_status + c0);
    }
    HIPCHK(c, hipGetLastError());
    return ZNGAMD_OK;
}

int zngamd_gzip_inflate_members_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, const zngamd_member *d_members,
                                    uint32_t n_members, void *d_out, uint64_t out_cap, int32_t *d_status)
try {
    if (!c || !d_in || !d_members || !d_out || !d_status) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    int r = inflate_members_dev(c, (const uint8_t *)d_in, in_len, (const ZaMember *)d_members, n_members, (uint8_t *)d_out, out_cap, d_status);
    if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_gzip_inflate_plain_members_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, const zngamd_member *d_members,
                                          uint32_t n_members, void *d_out, uint64_t out_cap, int32_t *d_status)
try {
    if (!c || !d_in || !d_members || !d_out || !d_status) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (n_members) {
        ProfScope ps(c, ZNGAMD_K_INFLATE);
        hipLaunchKernelGGL(za_k_inflate_serial_members, dim3(n_members), dim3(64), 0, c->stream, (const uint8_t *)d_in, in_len,
                           (const ZaMember *)d_members, (uint8_t *)d_out, out_cap, c->d_crc_table, c->d_x8k, d_status);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_inflate_raw_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, void *d_out, uint64_t out_cap, uint64_t *out_len,
                           uint64_t *in_used)
try {
    if (!c || !d_in || !d_out || !out_len) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t clen = 0, cused = 0;
    const int cr = inflate_chunked_dev(c, (const uint8_t *)d_in, in_len, (uint8_t *)d_out, out_cap, &clen, &cused);
    if (cr < 0 && cr != ZNGAMD_BUF_ERROR) return cr;
    if (cr == ZNGAMD_BUF_ERROR) { *out_len = clen; if (in_used) *in_used = 0; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
    if (cr == 0) { *out_len = clen; if (in_used) *in_used = cused; c->paths[ZNGAMD_PATH_CHUNKED]++; return ZNGAMD_STREAM_END; }
    ZaInfResult res;
    int r = inflate_serial_dev(c, (const uint8_t *)d_in, in_len, nullptr, 0, (uint8_t *)d_out, out_cap, &res);
    if (r) return r;
    c->paths[ZNGAMD_PATH_SEQUENTIAL]++;
    *out_len = res.out_len;
    if (in_used) *in_used = (res.in_bits + 7) >> 3;
    if (res.status == ZA_I_DATA) c->err = "invalid deflate data";
    return map_status(res.status);
} ZA_ABI_GUARD

// ---- the default writer's stream with the writer's index (za_inflate_units.hip)
int zngamd_deflate_index_dev(zngamd_ctx *c, uint32_t *d_index, uint32_t n_units)
try {
    if (!c || !d_index) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_units == 0) return ZNGAMD_OK;
    if (n_units != c->last_units || !c->cidx.p) return fail(c, ZNGAMD_E_ARG, "the index is that of the context's last deflate call: its unit count differs");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(d_index, c->cidx.p, (size_t)n_units * ZA_CIDX_STRIDE * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

static_assert(ZNGAMD_INDEX_STRIDE == ZA_CIDX_STRIDE, "index stride");
// (the caller holds the context's lock; device pointers, host size arrays; leaves the stream drained)
static int inflate_units_core(zngamd_ctx *c, const uint8_t *d_def, uint64_t def_len, const uint32_t *unit_in_len, const uint32_t *unit_out_len,
                              uint32_t n_units, const uint32_t *d_index, const void *d_dict, uint32_t dict_len,
                              uint8_t *d_out, uint64_t out_cap, uint64_t *out_len)
{
    *out_len = 0;
    uint64_t tin = 0, tout = 0;
    for (uint32_t u = 0; u < n_units; u++) {
        if (unit_out_len[u] > ZA_MAX_UNIT) return fail(c, ZNGAMD_E_ARG, "a unit is larger than 128 KiB");
        tin += unit_in_len[u]; tout += unit_out_len[u];
    }
    if (tin > def_len) return fail(c, ZNGAMD_E_ARG, "the units' bytes exceed the stream");
    *out_len = tout;
    if (tout > out_cap) return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small");
    constexpr uint64_t AREA = (uint64_t)ZA_WIN + ZA_MAX_UNIT;                 // symbols of a unit's area: markers, then the unit
    static const uint32_t batch = [] { const char *e = getenv("ZNGAMD_UNIT_BATCH"); long v = e ? atol(e) : 16384; return (uint32_t)(v < 64 ? 64 : v > 65536 ? 65536 : v); }();
    uint64_t in_at = 0, out_at = 0;
    std::vector<ZaMember> hm; std::vector<ZaChunk> chain; std::vector<ZaChunkRes> res; std::vector<uint32_t> uidx;
    for (uint32_t u0 = 0; u0 < n_units; u0 += batch) {
        const uint32_t u1 = std::min(n_units, u0 + batch);
        hm.clear(); chain.clear(); uidx.clear();
        uint64_t acc = 0;                                                     // output of this batch so far
        for (uint32_t u = u0; u < u1; u++) {
            if (unit_out_len[u] != 0) {
                ZaMember m;
                m.in_off = in_at; m.in_len = unit_in_len[u]; m.out_off = (uint64_t)hm.size() * AREA; m.out_len = unit_out_len[u];
                m.crc = (uint32_t)std::min<uint64_t>(ZA_WIN, out_at + acc + dict_len);       // (history in front of the unit)
                m.index_off = 0; m.nseg = (unit_out_len[u] + ZA_SEG - 1) >> ZA_SEG_SHIFT;
                ZaChunk ch; ch.in_bit = in_at * 8ull; ch.out_off = acc; ch.out_len = unit_out_len[u]; ch.end_bit = (in_at + unit_in_len[u]) * 8ull;
                ch.src_off = m.out_off + ZA_WIN;
                hm.push_back(m); chain.push_back(ch); uidx.push_back(u);
                acc += unit_out_len[u];
            }
            in_at += unit_in_len[u];
        }
        const uint32_t m = (uint32_t)hm.size();
        if (m == 0) continue;
        // the units' index rows are picked by unit number: the kernel reads row blockIdx.x, so a batch without empty units passes
        // its first row; one with empty units has its rows gathered
        const uint32_t *d_rows = d_index + (size_t)u0 * ZA_CIDX_STRIDE;
        if (m != u1 - u0) {
            HIPCHK(c, c->st_len.ensure((size_t)m * ZA_CIDX_STRIDE));
            for (uint32_t k = 0; k < m; k++)
                HIPCHK(c, hipMemcpyAsync(c->st_len.p + (size_t)k * ZA_CIDX_STRIDE, d_index + (size_t)uidx[k] * ZA_CIDX_STRIDE, ZA_CIDX_STRIDE * 4, hipMemcpyDeviceToDevice, c->stream));
            d_rows = c->st_len.p;
        }
        {
            // (grown with a quarter of head-room: the windows of a reader hold a few units more or fewer each time, and every
            // reallocation loses the areas' markers)
            const size_t want = (size_t)std::min<uint32_t>(batch, n_units) * AREA + 64, cap0 = c->uarea.cap;
            const uint16_t *p0 = c->uarea.p;
            if (want > cap0) HIPCHK(c, c->uarea.ensure(std::min<size_t>((size_t)batch * AREA + 64, want + want / 4)));
            if (c->uarea.cap != cap0 || c->uarea.p != p0) c->uarea_marked = 0;
        }
        if (c->uarea.cap < (size_t)m * AREA + 64) return fail(c, ZNGAMD_E_HIP, "unit areas");
        HIPCHK(c, c->members.ensure(m)); HIPCHK(c, c->cres.ensure(m)); HIPCHK(c, c->cchunks.ensure(m));
        HIPCHK(c, c->matchq.ensure((size_t)m * 64 * ZA_MATCHQ_PER_SEG));
        const uint32_t groups = (m + ZA_CHUNK_GROUP - 1) / ZA_CHUNK_GROUP;
        HIPCHK(c, c->winbuf.ensure((size_t)groups * ZA_WIN)); HIPCHK(c, c->ccomp.ensure((size_t)m * ZA_WIN));
        HIPCHK(c, hipMemcpyAsync(c->members.p, hm.data(), (size_t)m * sizeof(ZaMember), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->cchunks.p, chain.data(), (size_t)m * sizeof(ZaChunk), hipMemcpyHostToDevice, c->stream));
        { ProfScope ps(c, ZNGAMD_K_INFLATE);
          if (c->uarea_marked < m) {
              hipLaunchKernelGGL(za_k_fill_marker_prefix, dim3(m - c->uarea_marked), dim3(256), 0, c->stream, c->uarea.p + (uint64_t)c->uarea_marked * AREA, AREA);
              c->uarea_marked = m;
          }
          hipLaunchKernelGGL(za_k_inflate_units_marked, dim3(m), dim3(64), 0, c->stream, d_def, def_len, c->members.p, c->uarea.p, (uint64_t)c->uarea.cap,
                             c->matchq.p, d_rows, c->cres.p); }
        HIPCHK(c, hipGetLastError());
        res.resize(m);
        HIPCHK(c, hipMemcpyAsync(res.data(), c->cres.p, (size_t)m * sizeof(ZaChunkRes), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->indexed_units += m;
        for (uint32_t k = 0; k < m; k++) {
            if (res[k].status == ZA_I_DATA) return fail(c, ZNGAMD_DATA_ERROR, "invalid deflate data");
            if ((res[k].status != ZA_I_SYNC && res[k].status != ZA_I_END) || res[k].out_len != hm[k].out_len)
                return fail(c, ZNGAMD_E_INDEX, "the index does not fit the stream (or the stream is not this engine's indexed form): decode it without the index");
        }
        // windows in front of the units, then markers -> bytes (the chunk pipeline's kernels: a unit is a chunk)
        const uint8_t *bd = out_at ? d_out + out_at - std::min<uint64_t>(ZA_WIN, out_at) : (const uint8_t *)d_dict;
        const uint32_t bdl = out_at ? (uint32_t)std::min<uint64_t>(ZA_WIN, out_at) : dict_len;
        if (out_at && out_at < (uint64_t)ZA_WIN) return fail(c, ZNGAMD_E_ARG, "a batch of units shorter than the window");
        { ProfScope ps(c, ZNGAMD_K_INFLATE);
          hipLaunchKernelGGL(za_k_chunk_compose, dim3(groups), dim3(1024), 0, c->stream, c->uarea.p, c->cchunks.p, m, c->ccomp.p);
          hipLaunchKernelGGL(za_k_chunk_chain, dim3(1), dim3(1024), 0, c->stream, c->ccomp.p, m, c->winbuf.p, bd, bdl);
          hipLaunchKernelGGL(za_k_chunk_resolve, dim3(m), dim3(ZA_RESOLVE_THREADS), 0, c->stream, c->uarea.p, c->cchunks.p, c->ccomp.p, c->winbuf.p, d_out + out_at); }
        HIPCHK(c, hipGetLastError());
        out_at += acc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZNGAMD_STREAM_END;
}

int zngamd_inflate_units_indexed_dev(zngamd_ctx *c, const void *d_def, uint64_t def_len, const uint32_t *unit_in_len, const uint32_t *unit_out_len,
                                     uint32_t n_units, const uint32_t *d_index, const void *d_dict, uint32_t dict_len,
                                     void *d_out, uint64_t out_cap, uint64_t *out_len)
try {
    if (!c || !d_def || !unit_in_len || !unit_out_len || !d_index || !d_out || !out_len || dict_len > ZA_WIN) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const int r = inflate_units_core(c, (const uint8_t *)d_def, def_len, unit_in_len, unit_out_len, n_units, d_index, d_dict, dict_len, (uint8_t *)d_out, out_cap, out_len);
    if (r == ZNGAMD_STREAM_END) c->paths[ZNGAMD_PATH_CHUNKED]++;
    return r;
} ZA_ABI_GUARD

// ---- the index of a FILE's data member: what the writer leaves in the trailing empty members, parsed by the caller and handed to
// the windowed reader through zngamd_gz_state.index
struct ZaFileIndex {
    uint32_t n = 0; int device = 0;
    std::vector<uint32_t> in_len, out_len;
    std::vector<uint64_t> cum_in, cum_out;       // bytes in front of unit k (n + 1 entries)
    uint32_t *d_rows = nullptr;
    bool dead = false;                           // it did not fit the stream once: never tried again
};
int zngamd_index_create(zngamd_ctx *c, uint32_t n_units, const uint32_t *unit_in_len, const uint32_t *unit_out_len, const uint32_t *rows, void **handle)
try {
    if (!c || !handle || (n_units && (!unit_in_len || !unit_out_len || !rows))) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    ZaFileIndex *ix = new ZaFileIndex();
    ix->n = n_units; ix->device = c->device;
    ix->in_len.assign(unit_in_len, unit_in_len + n_units); ix->out_len.assign(unit_out_len, unit_out_len + n_units);
    ix->cum_in.resize((size_t)n_units + 1); ix->cum_out.resize((size_t)n_units + 1);
    ix->cum_in[0] = ix->cum_out[0] = 0;
    for (uint32_t u = 0; u < n_units; u++) {
        if (unit_out_len[u] > ZA_MAX_UNIT) { delete ix; return fail(c, ZNGAMD_E_ARG, "a unit is larger than 128 KiB"); }
        ix->cum_in[u + 1] = ix->cum_in[u] + unit_in_len[u]; ix->cum_out[u + 1] = ix->cum_out[u] + unit_out_len[u];
    }
    if (n_units) {
        if (hipMalloc((void **)&ix->d_rows, (size_t)n_units * ZA_CIDX_STRIDE * 4) != hipSuccess) { delete ix; return fail(c, ZNGAMD_MEM_ERROR, "index rows"); }
        if (hipMemcpy(ix->d_rows, rows, (size_t)n_units * ZA_CIDX_STRIDE * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(ix->d_rows); delete ix; return fail(c, ZNGAMD_E_HIP, "index rows"); }
    }
    *handle = ix;
    return ZNGAMD_OK;
} ZA_ABI_GUARD
void zngamd_index_destroy(void *handle)
{
    ZaFileIndex *ix = (ZaFileIndex *)handle;
    if (!ix) return;
    if (ix->d_rows) { (void)hipSetDevice(ix->device); (void)hipFree(ix->d_rows); }
    delete ix;
}
// the index of the context's last deflate call on the HOST, with the units' sizes (what a writer keeps for its file's trailing members)
static int deflate_index_locked(zngamd_ctx *c, uint32_t n_units, uint32_t *unit_in_len, uint32_t *unit_out_len, uint32_t *rows)
{
    if (n_units == 0) return ZNGAMD_OK;
    const bool on_host = c->last_ulen_on_host && c->last_ulen_host.size() == n_units;
    if (n_units != c->last_units || !c->cidx.p || (!on_host && !c->last_unit_len) || c->last_hu.size() != n_units) return fail(c, ZNGAMD_E_ARG, "the index is that of the context's last deflate call: its unit count differs");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(rows, c->cidx.p, (size_t)n_units * ZA_CIDX_STRIDE * 4, hipMemcpyDeviceToHost, c->stream));
    if (on_host) memcpy(unit_in_len, c->last_ulen_host.data(), (size_t)n_units * 4);
    else HIPCHK(c, hipMemcpyAsync(unit_in_len, c->last_unit_len, (size_t)n_units * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t u = 0; u < n_units; u++) unit_out_len[u] = c->last_hu[u].in_len;
    return ZNGAMD_OK;
}

int zngamd_deflate_index(zngamd_ctx *c, uint32_t n_units, uint32_t *unit_in_len, uint32_t *unit_out_len, uint32_t *rows)
try {
    if (!c || !unit_in_len || !unit_out_len || !rows) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    return deflate_index_locked(c, n_units, unit_in_len, unit_out_len, rows);
} ZA_ABI_GUARD

int zngamd_crc32_fold_dev(zngamd_ctx *c, const uint32_t *d_crcs, uint32_t n, uint64_t each_len, uint64_t last_len, uint32_t *crc)
try {
    if (!c || (!d_crcs && n) || !crc) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint32_t> v(n);
    if (n) HIPCHK(c, hipMemcpyAsync(v.data(), d_crcs, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // x^(8 each_len) once; then Horner: crc = crc * x^(8 len_i) ^ crc_i
    uint32_t xe = 0x80000000u, sq = 0x00800000u;
    for (uint64_t k = each_len; k; k >>= 1) { if (k & 1) xe = za_multmodp(sq, xe); sq = za_multmodp(sq, sq); }
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (i == 0) r = v[0];
        else if (i + 1 < n || last_len == each_len) r = za_multmodp(xe, r) ^ v[i];
        else r = zngamd_crc32_combine(r, v[i], last_len);
    }
    *crc = r;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_compare_dev(zngamd_ctx *c, const void *d_a, const void *d_b, uint64_t n, uint64_t *mismatches)
try {
    if (!c || (!d_a && n) || (!d_b && n) || !mismatches) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long *d_bad = (unsigned long long *)((uint8_t *)c->d_small + 192);
    HIPCHK(c, hipMemsetAsync(d_bad, 0, 8, c->stream));
    if (n) {
        const uint64_t threads = (n + 15) / 16;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((threads + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(za_k_compare, dim3(blocks), dim3(256), 0, c->stream, (const uint8_t *)d_a, (const uint8_t *)d_b, n, d_bad);
    }
    HIPCHK(c, hipGetLastError());
    unsigned long long bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *mismatches = bad;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

// ---- general gzip reader (host buffer) --------------------------------------------------------
// Restates the member state machine of GzipReader_read_into_buffer (zlib_ngmodule.c:2443-2611):
// header fields, inflate, CRC / ISIZE check, NUL padding.  Header bytes are parsed on the host (they
// are framing, not payload); every payload byte is decoded and checksummed on the GPU.
static int parse_gzip_header(const uint8_t *in, uint64_t in_len, uint64_t pos, uint64_t *data_off, bool *za_indexed, uint32_t *hcrc_len)
{
    *za_indexed = false; *hcrc_len = 0;
    if (in_len - pos < 10) return ZNGAMD_E_GZ_TRUNC;
    if (!(in[pos] == 0x1f && in[pos + 1] == 0x8b)) return ZNGAMD_E_GZ_MAGIC;
    if (in[pos + 2] != 8) return ZNGAMD_E_GZ_METHOD;
    const int flags = in[pos + 3];
    uint64_t cur = pos + 10;
    if (flags & 4) {
        if (cur + 2 >= in_len) return ZNGAMD_E_GZ_TRUNC;
        const uint64_t fl = in[cur] | (in[cur + 1] << 8);
        cur += 2;
        if (cur + fl >= in_len) return ZNGAMD_E_GZ_TRUNC;
        if (fl >= ZA_MEMBER_FIXED - 12 && in[cur] == 'Z' && in[cur + 1] == 'A' && flags == 4 && in[pos + 26] == 2) *za_indexed = true;
        cur += fl;
    }
    if (flags & 8) {
        const void *z = memchr(in + cur, 0, in_len - cur);
        if (!z) return ZNGAMD_E_GZ_TRUNC;
        cur = (uint64_t)((const uint8_t *)z - in) + 1;
    }
    if (flags & 16) {
        const void *z = memchr(in + cur, 0, in_len - cur);
        if (!z) return ZNGAMD_E_GZ_TRUNC;
        cur = (uint64_t)((const uint8_t *)z - in) + 1;
    }
    if (flags & 2) {
        if (cur + 2 >= in_len) return ZNGAMD_E_GZ_TRUNC;
        *hcrc_len = (uint32_t)(cur - pos);
        cur += 2;
    }
    *data_off = cur;
    return ZNGAMD_OK;
}

// why the chunk-parallel path handed over to the sequential decoder (ZNGAMD_DEBUG=1 prints it)
static int chunk_bail(int why)
{
    static const bool dbg = getenv("ZNGAMD_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "zng_amd: chunk-parallel inflate not used (reason %d)\n", why);
    return 1;
}

// Chunk-parallel inflate of one member: candidate chunk starts = positions after sync-flush
// markers and bit offsets where a dynamic block header parses; a count-only pass sizes every candidate up to the
// next listed boundary, the host follows the chain of real block boundaries and groups blocks into chunks.
// d_def = device pointer to the first deflate byte (inside a padded staging buffer), avail = bytes from there to
// the end of the input.  Returns 0 when the member was decoded (out_len / in_used set; CRC / ISIZE are checked by
// the caller), 1 when this path does not apply or anything looked odd (caller uses the sequential decoder, which
// also produces the exact error), <0 on engine errors, ZNGAMD_BUF_ERROR with *out_len = needed size when the
// output does not fit.
// Options: the stream may start at a bit offset inside the first byte with up to 32 KiB of history (a stream resumed at
// a block header), and with allow_cut a stream that runs past the end of the buffer is decoded up to its last complete
// block (info->cut, info->end_bit = where the next block header starts; *in_used = that bit's byte).
// A long stream goes through the pipeline in BATCHES of compressed input (ZNGAMD_CHUNK_BATCH_MIB, 512 MiB): every batch is a
// stream resumed at a block header with the 32 KiB in front of it as its history (the options above), cut in front of its last,
// incomplete piece -- so the scratch area of the one-pass marker decode (12 bytes per compressed byte) is bounded by the batch,
// not by the stream: 4 GiB of text asked for 18 GiB of it and went through the count pass instead (7.8 of 38 ms), a 16 GiB
// stream would not have fitted a busy GPU at all.
static int inflate_chunked_dev(zngamd_ctx *c, const uint8_t *d_def, uint64_t avail, uint8_t *d_out, uint64_t out_room,
                               uint64_t *out_len, uint64_t *in_used, const ChunkOpts &o, ChunkInfo *info)
{
    static const uint64_t batch = [] { const char *e = getenv("ZNGAMD_CHUNK_BATCH_MIB"); long v = e ? atol(e) : 512; return (uint64_t)(v < 16 ? 16 : v > 8192 ? 8192 : v) << 20; }();
    ChunkInfo dummy; if (!info) info = &dummy;
    if (avail <= batch + batch / 4) return inflate_chunked_once(c, d_def, avail, d_out, out_room, out_len, in_used, o, info);
    uint64_t pos_bit = o.start_bit, acc = 0;
    const uint8_t *dict = o.d_dict; uint32_t dict_len = o.dict_len;
    for (int b = 0;; b++) {
        const uint64_t byte0 = pos_bit >> 3;
        const bool lastb = avail - byte0 <= batch + batch / 4;
        ChunkOpts bo; bo.start_bit = (uint32_t)(pos_bit & 7ull); bo.d_dict = dict; bo.dict_len = dict_len; bo.allow_cut = lastb ? o.allow_cut : true;
        ChunkInfo bi; uint64_t bl = 0, bu = 0;
        const int r = inflate_chunked_once(c, d_def + byte0, lastb ? avail - byte0 : batch, d_out + acc, out_room - acc, &bl, &bu, bo, &bi);
        if (r == ZNGAMD_BUF_ERROR) {
            // the room this batch needs is known, the batches behind it are not decoded yet: their share is extrapolated from the
            // expansion so far (+ 1/16), so that a caller who resizes to the reported size retries once, not once per batch
            uint64_t need = acc + bl;
            if (!lastb) need = std::max<uint64_t>(need, (uint64_t)((long double)need * (long double)avail / (long double)(byte0 + batch) * 1.0625L) + (1u << 20));
            *out_len = need; *info = bi; return r;
        }
        if (r < 0) return r;
        if (r != 0 || (!lastb && !bi.ended && (bl == 0 || acc + bl < (uint64_t)ZA_WIN))) {
            // a batch this path does not take (or one that made no headway): the whole stream the old way, as one
            if (b == 0 && r != 0) { *info = bi; }
            return inflate_chunked_once(c, d_def, avail, d_out, out_room, out_len, in_used, o, info);
        }
        acc += bl;
        if (lastb || bi.ended) {
            *out_len = acc; *in_used = byte0 + bu;
            info->cut = bi.cut; info->ended = bi.ended; info->end_bit = byte0 * 8ull + bi.end_bit;
            return 0;
        }
        pos_bit = byte0 * 8ull + bi.end_bit;
        dict = d_out + acc - ZA_WIN; dict_len = ZA_WIN;               // (acc >= 32 KiB: checked above)
    }
}

static int inflate_chunked_once(zngamd_ctx *c, const uint8_t *d_def, uint64_t avail, uint8_t *d_out, uint64_t out_room,
                                uint64_t *out_len, uint64_t *in_used, const ChunkOpts &o, ChunkInfo *info)
{
    ChunkInfo dummy; if (!info) info = &dummy;
    *info = ChunkInfo();
    const bool resumed = o.start_bit != 0 || o.dict_len != 0 || o.allow_cut;
    if (avail < (1u << 16) || avail > (1ull << 36)) return chunk_bail(1);
    const uint32_t max_c = (uint32_t)std::min<uint64_t>(avail / 8 + 64, 1u << 24);
    const uint32_t max_s = (uint32_t)std::min<uint64_t>(avail / 4 + 1024, 1u << 25);
    HIPCHK(c, c->ccand.ensure((size_t)max_c + 1)); HIPCHK(c, c->csurv.ensure(max_s));
    uint32_t *d_n = (uint32_t *)((uint8_t *)c->d_small + 128);          // [0] candidates, [1] survivors
    PhaseClock pc_all(c, "chunk-parallel inflate, all of it");
    auto *pc = new PhaseClock(c, "  sync scan + candidates");
    struct PcOwner { PhaseClock *&p; ~PcOwner() { delete p; } } pc_owner{pc};
    auto phase = [&](const char *name) { delete pc; pc = nullptr; pc = new PhaseClock(c, name); };
    HIPCHK(c, hipMemsetAsync(d_n, 0, 8, c->stream));
    {   ProfScope ps(c, ZNGAMD_K_SCAN);
        const uint64_t threads = (avail + 15) / 16;
        hipLaunchKernelGGL(za_k_scan_sync, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, d_def, avail, c->ccand.p, max_c, d_n);
    }
    HIPCHK(c, hipGetLastError());
    uint32_t cnt[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(cnt, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (cnt[0] > max_c) return chunk_bail(2);
    // sync-scan hits are byte positions: turn them into bit offsets on the host
    std::vector<uint64_t> cand(cnt[0]);
    if (cnt[0]) HIPCHK(c, hipMemcpy(cand.data(), c->ccand.p, (size_t)cnt[0] * 8, hipMemcpyDeviceToHost));
    for (auto &q : cand) q *= 8ull;
    // A stream written block-parallel (the reference's threaded writer, pigz, this engine) has a sync point every block: when
    // no stretch of more than 2 MiB is without one, those are boundaries enough and the bit-level header finder (a pass over
    // every bit offset of the stream, about as dear as the decode itself) is not run.
    static const uint32_t dense_min = [] { const long v = getenv("ZNGAMD_DENSE_MIN") ? atol(getenv("ZNGAMD_DENSE_MIN")) : 8; return (uint32_t)(v < 1 ? 1 : v); }();
    bool dense = cnt[0] >= dense_min;
    if (dense) {
        std::sort(cand.begin(), cand.end());
        uint64_t prev = o.start_bit, gap = 0;
        for (const uint64_t q : cand) { if (q > prev && q - prev > gap) gap = q - prev; if (q > prev) prev = q; }
        if (avail * 8ull > prev && avail * 8ull - prev > gap) gap = avail * 8ull - prev;
        dense = gap <= (16ull << 20);                                   // bits: 2 MiB
    }
    if (!dense) {
        { ProfScope ps(c, ZNGAMD_K_SCAN);
          hipLaunchKernelGGL(za_k_find_blocks_a, dim3((uint32_t)((avail / 4 + 1 + ZA_FINDA_THREADS - 1) / ZA_FINDA_THREADS)), dim3(ZA_FINDA_THREADS), 0, c->stream, d_def, avail, c->csurv.p, max_s, d_n + 1); }      // a thread per aligned dword
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(cnt + 1, d_n + 1, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (cnt[1] > 0 && cnt[1] <= max_s) {
        HIPCHK(c, hipMemsetAsync(d_n, 0, 4, c->stream));
        { ProfScope ps(c, ZNGAMD_K_SCAN);
          hipLaunchKernelGGL(za_k_find_blocks_b, dim3((cnt[1] + 63) / 64), dim3(64), 0, c->stream, d_def, avail, c->csurv.p, cnt[1], c->ccand.p, max_c, d_n); }
        HIPCHK(c, hipGetLastError());
        uint32_t nb = 0;
        HIPCHK(c, hipMemcpyAsync(&nb, d_n, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (nb > 0 && nb <= max_c) {
            const size_t old = cand.size();
            cand.resize(old + nb);
            HIPCHK(c, hipMemcpy(cand.data() + old, c->ccand.p, (size_t)nb * 8, hipMemcpyDeviceToHost));
        }
    }
    cand.push_back(o.start_bit);
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    const uint32_t n = (uint32_t)cand.size();
    if (getenv("ZNGAMD_DEBUG")) fprintf(stderr, "zng_amd: chunk finder: %u sync hits, %u header survivors, %u candidates\n", cnt[0], cnt[1], n);
    if (n < 8 || n > max_c) return chunk_bail(3);                    // too few boundaries to be worth it
    std::vector<ZaChunk> chain;
    std::vector<ZaChunkRes> res;
    uint64_t acc = 0, end_bit = 0;
    bool ended = false, placed = false;
    // ONE decoding pass where the boundaries are the stream's own sync points and the pieces between them are large enough to be
    // chunks as they are (what block-parallel writers emit: this engine, the reference's threaded writer, pigz): every piece is
    // decoded to the next boundary into a scratch area of its own (6 symbols per compressed byte + 64 Ki; a piece that outgrows it
    // sends the stream to the two passes below), sizes and ends come out of that pass, and the window kernels read the symbols
    // where they lie.  The count pass it saves decoded everything once to learn 16 bytes per piece (1.9 of 6.9 ms for 256 MiB).
    static const bool one_pass_allowed = getenv("ZNGAMD_TWO_PASS") == nullptr;
    if (dense && one_pass_allowed && avail / n >= (8u << 10)) {
        std::vector<ZaChunk> pieces(n);
        uint64_t sc = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t ext = ((i + 1 < n ? cand[i + 1] : avail * 8ull) - cand[i] + 7) >> 3;
            ZaChunk pc; pc.in_bit = cand[i]; pc.out_off = 0; pc.out_len = 6 * ext + 65536; pc.end_bit = 0; pc.src_off = sc;
            pieces[i] = pc;
            sc += (pc.out_len + 7) & ~7ull;                            // (16-byte aligned symbol areas)
        }
        // (no room for the scratch area on this device right now: not an error -- the two passes below need a sixth of it)
        bool scratch_ok = sc * 2 <= (16ull << 30);
        if (scratch_ok && c->out16.ensure(sc + 64) != hipSuccess) { (void)hipGetLastError(); scratch_ok = false; }
        if (scratch_ok) {
            phase("  marker decode (one pass)");
            HIPCHK(c, c->cchunks.ensure(n)); HIPCHK(c, c->cres.ensure(n));
            HIPCHK(c, hipMemcpyAsync(c->ccand.p, cand.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->cchunks.p, pieces.data(), (size_t)n * sizeof(ZaChunk), hipMemcpyHostToDevice, c->stream));
            { ProfScope ps(c, ZNGAMD_K_INFLATE);
              if (n >= ZNGAMD_CHUNKS_MANY_FROM)
                  hipLaunchKernelGGL((za_k_chunk_decode<384, ZA_CHUNK_Q_M, ZA_CHUNK_RING_M, 9, 8>), dim3(n), dim3(64), 0, c->stream, d_def, avail, c->cchunks.p, c->out16.p, c->cres.p, (uint64_t)o.start_bit, o.dict_len, c->ccand.p, n);
              else if (n >= ZNGAMD_CHUNKS_SMALL_FROM)
                  hipLaunchKernelGGL((za_k_chunk_decode<ZA_CHUNK_BITS_S, ZA_CHUNK_Q_S, ZA_CHUNK_RING_S>), dim3(n), dim3(64), 0, c->stream, d_def, avail, c->cchunks.p, c->out16.p, c->cres.p, (uint64_t)o.start_bit, o.dict_len, c->ccand.p, n);
              else
                  hipLaunchKernelGGL((za_k_chunk_decode<1024, 3072, 4096>), dim3(n), dim3(64), 0, c->stream, d_def, avail, c->cchunks.p, c->out16.p, c->cres.p, (uint64_t)o.start_bit, o.dict_len, c->ccand.p, n); }
            HIPCHK(c, hipGetLastError());
            res.resize(n);
            HIPCHK(c, hipMemcpyAsync(res.data(), c->cres.p, (size_t)n * sizeof(ZaChunkRes), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            bool good = true, cut = false;
            size_t i = (size_t)(std::lower_bound(cand.begin(), cand.end(), (uint64_t)o.start_bit) - cand.begin());
            for (uint32_t guard = 0; guard <= n && good; guard++) {
                const ZaChunkRes &r = res[i];
                if (r.status == ZA_I_INPUT) {
                    cut = true;
                    if (o.allow_cut && !chain.empty()) { end_bit = cand[i]; break; }
                }
                if ((r.status != ZA_I_SYNC && r.status != ZA_I_END) || r.max_back > acc + o.dict_len) { good = false; break; }
                ZaChunk ch; ch.in_bit = cand[i]; ch.out_off = acc; ch.out_len = r.out_len; ch.end_bit = r.bits; ch.src_off = pieces[i].src_off;
                chain.push_back(ch);
                acc += r.out_len;
                if (r.status == ZA_I_END) { end_bit = r.bits; ended = true; break; }
                auto it = std::lower_bound(cand.begin(), cand.end(), r.bits);
                if (it == cand.end() || *it != r.bits) { good = false; break; }
                i = (size_t)(it - cand.begin());
            }
            if (good && (ended || (o.allow_cut && cut && !chain.empty())) && chain.size() >= (resumed ? 1u : 4u)) { placed = true; info->cut = cut; }
            else { chain.clear(); acc = 0; end_bit = 0; ended = false; }         // anything odd: the two passes decide
        }
    }
    if (!placed) {
    phase("  count pass");
    HIPCHK(c, hipMemcpyAsync(c->ccand.p, cand.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->cres.ensure(n));
    { ProfScope ps(c, ZNGAMD_K_INFLATE);
      if (n >= ZNGAMD_CHUNKS_SMALL_FROM)
          hipLaunchKernelGGL(za_k_chunk_count<512>, dim3(n), dim3(64), 0, c->stream, d_def, avail, c->ccand.p, n, c->cres.p, (uint64_t)o.start_bit, o.dict_len);
      else
          hipLaunchKernelGGL(za_k_chunk_count<1024>, dim3(n), dim3(64), 0, c->stream, d_def, avail, c->ccand.p, n, c->cres.p, (uint64_t)o.start_bit, o.dict_len); }
    HIPCHK(c, hipGetLastError());
    res.resize(n);
    HIPCHK(c, hipMemcpyAsync(res.data(), c->cres.p, (size_t)n * sizeof(ZaChunkRes), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // follow the chain of real block boundaries from the start of the stream, then group blocks into chunks:
    // about 4096 of them (12 marker decoders fit a CU), none below 32 KiB (a chunk's tail is the next window)
    phase("  chain of blocks on the host");
    std::vector<ZaChunk> blocks;
    {
        size_t i = (size_t)(std::lower_bound(cand.begin(), cand.end(), (uint64_t)o.start_bit) - cand.begin());
        for (uint32_t guard = 0; guard <= n; guard++) {
            const ZaChunkRes &r = res[i];
            if (r.status == ZA_I_INPUT) {
                info->cut = true;
                if (o.allow_cut && !blocks.empty()) { end_bit = cand[i]; break; }      // stop in front of the incomplete block
            }
            if (r.status != ZA_I_SYNC && r.status != ZA_I_END) return chunk_bail(4);
            ZaChunk b; b.in_bit = cand[i]; b.out_off = acc; b.out_len = r.out_len; b.end_bit = r.bits; b.src_off = acc;
            blocks.push_back(b);
            acc += r.out_len;
            if (r.status == ZA_I_END) { end_bit = r.bits; ended = true; break; }
            auto it = std::lower_bound(cand.begin(), cand.end(), r.bits);
            if (it == cand.end() || *it != r.bits) return chunk_bail(5);
            i = (size_t)(it - cand.begin());
        }
    }
    static const uint64_t chunk_div = [] { const long v = getenv("ZNGAMD_CHUNK_DIV") ? atol(getenv("ZNGAMD_CHUNK_DIV")) : 4096; return (uint64_t)(v < 1 ? 1 : v); }();
    const uint64_t target = std::max<uint64_t>(32u << 10, acc / chunk_div);
    for (size_t b = 0; b < blocks.size();) {
        ZaChunk cur = blocks[b++];
        while (b < blocks.size() && cur.out_len < target) { cur.out_len += blocks[b].out_len; cur.end_bit = blocks[b].end_bit; b++; }
        chain.push_back(cur);
    }
    if (chain.size() > 1 && chain.back().out_len < target) {       // short last chunk: merge into its predecessor
        ZaChunk last = chain.back(); chain.pop_back();
        chain.back().out_len += last.out_len; chain.back().end_bit = last.end_bit;
    }
    if (!ended && !(o.allow_cut && info->cut && !chain.empty())) return chunk_bail(6);
    if (chain.size() < (resumed ? 1u : 4u)) return chunk_bail(6);
    }
    info->ended = ended; info->end_bit = end_bit;
    *out_len = acc; *in_used = ended ? (end_bit + 7) >> 3 : end_bit >> 3;
    if (acc > out_room) return ZNGAMD_BUF_ERROR;
    const uint32_t m = (uint32_t)chain.size();
    const uint32_t groups = (m + ZA_CHUNK_GROUP - 1) / ZA_CHUNK_GROUP;
    HIPCHK(c, c->cchunks.ensure(m)); HIPCHK(c, c->winbuf.ensure((size_t)groups * ZA_WIN));
    HIPCHK(c, c->ccomp.ensure((size_t)m * ZA_WIN));
    if (!placed) {
        phase("  marker decode");
        HIPCHK(c, c->out16.ensure(acc + 64));
        HIPCHK(c, hipMemcpyAsync(c->cchunks.p, chain.data(), (size_t)m * sizeof(ZaChunk), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, c->cres.ensure(m));
        { ProfScope ps(c, ZNGAMD_K_INFLATE);
          if (m >= ZNGAMD_CHUNKS_MANY_FROM)
              hipLaunchKernelGGL((za_k_chunk_decode<384, ZA_CHUNK_Q_M, ZA_CHUNK_RING_M, 9, 8>), dim3(m), dim3(64), 0, c->stream, d_def, avail, c->cchunks.p, c->out16.p, c->cres.p, (uint64_t)o.start_bit, o.dict_len, (const uint64_t *)nullptr, 0u);
          else if (m >= ZNGAMD_CHUNKS_SMALL_FROM)
              hipLaunchKernelGGL((za_k_chunk_decode<ZA_CHUNK_BITS_S, ZA_CHUNK_Q_S, ZA_CHUNK_RING_S, ZA_CHUNK_LB_2P, ZA_CHUNK_DB_2P>), dim3(m), dim3(64), 0, c->stream, d_def, avail, c->cchunks.p, c->out16.p, c->cres.p, (uint64_t)o.start_bit, o.dict_len, (const uint64_t *)nullptr, 0u);
          else
              hipLaunchKernelGGL((za_k_chunk_decode<1024, 3072, 4096>), dim3(m), dim3(64), 0, c->stream, d_def, avail, c->cchunks.p, c->out16.p, c->cres.p, (uint64_t)o.start_bit, o.dict_len, (const uint64_t *)nullptr, 0u); }
        HIPCHK(c, hipGetLastError());
        res.resize(m);
        HIPCHK(c, hipMemcpyAsync(res.data(), c->cres.p, (size_t)m * sizeof(ZaChunkRes), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (uint32_t k = 0; k < m; k++) {
            const bool last = k + 1 == m;
            if (res[k].status != ((last && ended) ? ZA_I_END : ZA_I_SYNC) || res[k].out_len != chain[k].out_len || res[k].bits != chain[k].end_bit) return chunk_bail(7);
            if (res[k].max_back > chain[k].out_off + o.dict_len) return chunk_bail(8);       // reference before the start of the stream
        }
    } else HIPCHK(c, hipMemcpyAsync(c->cchunks.p, chain.data(), (size_t)m * sizeof(ZaChunk), hipMemcpyHostToDevice, c->stream));
    phase("  windows + resolve");
    { ProfScope ps(c, ZNGAMD_K_INFLATE);
      hipLaunchKernelGGL(za_k_chunk_compose, dim3(groups), dim3(1024), 0, c->stream, c->out16.p, c->cchunks.p, m, c->ccomp.p);
      hipLaunchKernelGGL(za_k_chunk_chain, dim3(1), dim3(1024), 0, c->stream, c->ccomp.p, m, c->winbuf.p, o.d_dict, o.dict_len);
      hipLaunchKernelGGL(za_k_chunk_resolve, dim3(m), dim3(ZA_RESOLVE_THREADS), 0, c->stream, c->out16.p, c->cchunks.p, c->ccomp.p, c->winbuf.p, d_out); }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return 0;
}

// BGZF hop: every member must be `1f 8b 08 04`, carry a 'B','C' subfield of length 2 (block size - 1) and
// tile the buffer exactly; anything else makes the caller use the general reader.
static bool hop_bgzf(const uint8_t *in, uint64_t in_len, std::vector<ZaMember> &hm, uint64_t *total_out,
                     bool allow_tail = false, uint64_t *covered = nullptr)
{
    auto le16 = [&](uint64_t o) { return (uint32_t)in[o] | ((uint32_t)in[o + 1] << 8); };
    auto le32 = [&](uint64_t o) { return (uint32_t)in[o] | ((uint32_t)in[o + 1] << 8) | ((uint32_t)in[o + 2] << 16) | ((uint32_t)in[o + 3] << 24); };
    uint64_t pos = 0, outp = 0;
    hm.clear();
    while (pos < in_len) {
        if (in_len - pos < 12 + 6 + 8) { if (allow_tail && !hm.empty()) break; return false; }
        if (in[pos] != 0x1f || in[pos + 1] != 0x8b || in[pos + 2] != 8 || in[pos + 3] != 4) { if (allow_tail && !hm.empty()) break; return false; }
        const uint64_t xlen = le16(pos + 10);
        uint64_t cur = pos + 12;
        const uint64_t end = cur + xlen;
        if (end + 8 > in_len) { if (allow_tail && !hm.empty()) break; return false; }
        long bsize = -1;
        bool bad = false;
        while (cur + 4 <= end) {
            const uint32_t sl = le16(cur + 2);
            if (cur + 4 + sl > end) { bad = true; break; }
            if (in[cur] == 'B' && in[cur + 1] == 'C' && sl == 2) bsize = (long)le16(cur + 4);
            cur += 4 + sl;
        }
        if (bad || bsize < 0) { if (allow_tail && !hm.empty()) break; return false; }
        const uint64_t msize = (uint64_t)bsize + 1;
        if (msize < (end - pos) + 8 || pos + msize > in_len) { if (allow_tail && !hm.empty()) break; return false; }
        ZaMember m;
        m.in_off = end; m.in_len = msize - (end - pos) - 8; m.out_off = outp;
        m.out_len = le32(pos + msize - 4); m.crc = le32(pos + msize - 8); m.index_off = 0; m.nseg = 0;
        hm.push_back(m);
        pos += msize; outp += m.out_len;
    }
    *total_out = outp;
    if (covered) *covered = pos;
    return pos == in_len || (allow_tail && !hm.empty());
}

// A file made of many small ordinary members (concatenated logs, `cat *.gz`, files written by appending): member by member
// every one costs a few launches and synchronisations (3.5 ms each; 2 000 members of 16 KiB took 7 s).  Where members start is
// not written anywhere, but every start shows the gzip magic: all places that look like a member header are sized at once by
// the count kernel (one wavefront each, to the end of their own stream), the host then follows the chain of members from
// offset 0 -- a false candidate inside compressed data is simply never reached -- and the members of the chain are decoded
// in one launch like BGZF members, CRC-32 and ISIZE checked in the kernel.  Returns the members of the longest chain prefix
// that is complete and unremarkable; the member loop of gunzip_impl goes on behind it (large members, errors, the tail).
static int hop_plain_members(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, std::vector<ZaMember> &hm, std::vector<uint64_t> &hpos,
                             uint64_t *total_out, uint64_t *covered)
{
    hm.clear(); hpos.clear(); *total_out = 0; *covered = 0;
    if (in_len < 40) return ZNGAMD_OK;
    struct Cand { uint64_t pos, doff; };
    std::vector<Cand> cv;
    uint64_t prev = 0, gap = 0;
    for (const uint8_t *p = in, *e = in + in_len - 18; p < e; p++) {
        p = (const uint8_t *)memchr(p, 0x1f, (size_t)(e - p));
        if (!p) break;
        if (p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) continue;
        const uint64_t pos = (uint64_t)(p - in);
        uint64_t doff = 0; bool za = false; uint32_t hl = 0;
        if (parse_gzip_header(in, in_len, pos, &doff, &za, &hl) != ZNGAMD_OK || hl) continue;     // (a header CRC is left to the member loop)
        if (pos - prev > gap) gap = pos - prev;
        prev = pos;
        cv.push_back(Cand{pos, doff});
        if (cv.size() > (1u << 20)) return ZNGAMD_OK;
    }
    if (in_len - prev > gap) gap = in_len - prev;
    // worth it for many small members only: a large member is better off chunk-parallel in the member loop
    if (cv.size() < 4 || cv[0].pos != 0 || gap > (1ull << 20)) return ZNGAMD_OK;
    const uint32_t n = (uint32_t)cv.size();
    std::vector<uint64_t> bits(n);
    for (uint32_t i = 0; i < n; i++) bits[i] = cv[i].doff * 8ull;
    HIPCHK(c, c->ccand.ensure((size_t)n + 1)); HIPCHK(c, c->cres.ensure(n));
    HIPCHK(c, hipMemcpyAsync(c->ccand.p, bits.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    { ProfScope ps(c, ZNGAMD_K_INFLATE);
      hipLaunchKernelGGL(za_k_chunk_count<512>, dim3(n), dim3(64), 0, c->stream, c->st_in.p, in_len, c->ccand.p, n, c->cres.p, 0ull, 0u, 1); }
    HIPCHK(c, hipGetLastError());
    std::vector<ZaChunkRes> res(n);
    HIPCHK(c, hipMemcpyAsync(res.data(), c->cres.p, (size_t)n * sizeof(ZaChunkRes), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    uint64_t pos = 0, outp = 0;
    size_t i = 0;
    while (pos < in_len) {
        while (i < n && cv[i].pos < pos) i++;
        if (i >= n || cv[i].pos != pos) break;                         // not a (plain) member header here: the member loop looks at it
        const ZaChunkRes &r = res[i];
        if (r.status != ZA_I_END || r.out_len > 0xFFFFFFFFull) break;  // truncated, damaged, huge: the member loop gives the verdict
        const uint64_t dend = (r.bits + 7) >> 3;                       // first byte behind the deflate stream
        if (dend + 8 > in_len) break;
        ZaMember m;
        m.in_off = cv[i].doff; m.in_len = dend - cv[i].doff; m.out_off = outp; m.out_len = (uint32_t)r.out_len;
        m.crc = 0; m.index_off = 0; m.nseg = 0;
        hm.push_back(m);
        hpos.push_back(cv[i].pos);
        outp += r.out_len;
        pos = dend + 8;
        while (pos < in_len && in[pos] == 0) pos++;
        *covered = pos;
    }
    *total_out = outp;
    return ZNGAMD_OK;
}

struct StreamRun { int status; uint64_t out_len; uint64_t in_bits; bool chunked; };
static int stream_finish(zngamd_ctx *c, zngamd_gz_state *st, const uint8_t *in, uint64_t in_len, uint64_t doff, bool last, const StreamRun &run,
                         uint32_t dl, uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members, uint64_t *in_consumed);

// partial: more input may follow -- complete members are decoded, an incomplete last one is left alone and
// *in_consumed tells where it starts (ZNGAMD_OK; nothing consumed = the window holds no complete member yet).
// With a stream state (st, partial only) a FIRST member that runs past the window is not left alone: the blocks of it that
// are complete were decoded on the way to finding that out, so they are handed out and the state says where the next
// window continues (what zngamd_gunzip_stream would otherwise do in a second pass over the same window).
struct ZaFileIndex;
static int decode_stream_prefix_indexed(zngamd_ctx *c, ZaFileIndex *ix, uint64_t out_total, const uint8_t *h_def, const uint8_t *d_def, uint64_t avail,
                                        const uint8_t *d_dict, uint32_t dict_len, uint8_t *d_out, uint64_t out_room, bool allow_cut,
                                        StreamRun *run, uint64_t *needed);
static int gunzip_impl(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, bool partial, uint8_t *out, uint64_t out_cap,
                       uint64_t *out_len, uint32_t *n_members, uint64_t *in_consumed, zngamd_gz_state *st = nullptr, void *index = nullptr)
{
    *out_len = 0;
    *in_consumed = 0;
    if (n_members) *n_members = 0;
    int r = stage_in(c, in, in_len);
    if (r) return r;
    HIPCHK(c, c->st_out.ensure(out_cap + 64));
    uint64_t pos = 0, op = 0, last_member_bytes = 0;
    uint32_t members = 0;
    int ret = ZNGAMD_OK;
    bool prefix_done = false;                        // the fast path for indexed members took the front of the buffer
    // fast path: the whole stream is indexed members -> two-pass scheme
    {
        uint64_t doff; bool za; uint32_t hl;
        if (in_len >= ZA_MEMBER_FIXED + 8 && parse_gzip_header(in, in_len, 0, &doff, &za, &hl) == ZNGAMD_OK && za) {
            std::vector<ZaMember> hm; uint64_t total = 0;
            uint64_t covered = 0;
            // (a run of indexed members at the front is enough: what follows them -- members of another writer appended to the
            // file -- goes to the member loop below)
            if (scan_members_dev(c, c->st_in.p, in_len, hm, &total, true, &covered) == ZNGAMD_OK) {
                if (total > out_cap) { *out_len = total; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
                HIPCHK(c, c->members.ensure(hm.size())); HIPCHK(c, c->mstatus.ensure(hm.size()));
                HIPCHK(c, hipMemcpyAsync(c->members.p, hm.data(), hm.size() * sizeof(ZaMember), hipMemcpyHostToDevice, c->stream));
                r = inflate_members_dev(c, c->st_in.p, in_len, c->members.p, (uint32_t)hm.size(), c->st_out.p, out_cap, c->mstatus.p);
                if (r) return r;
                std::vector<int32_t> st(hm.size());
                HIPCHK(c, hipMemcpyAsync(st.data(), c->mstatus.p, st.size() * 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                prof_collect(c);
                bool all_ok = true;
                for (size_t i = 0; i < st.size(); i++) if (st[i] != ZA_I_OK) { all_ok = false; break; }
                // NUL padding behind the members is skipped, as the reference's reader does (zlib_ngmodule.c:2604-2612) and as the member
                // loop below does behind every member it decodes
                uint64_t behind = covered;
                while (behind < in_len && ((const uint8_t *)in)[behind] == 0) behind++;
                if (all_ok && (behind == in_len || partial)) {
                    if (total) { const int rc_ = d2h_payload(c, out, c->st_out.p, total); if (rc_) return rc_; }
                    *out_len = total;
                    if (n_members) *n_members = (uint32_t)hm.size();
                    c->paths[ZNGAMD_PATH_INDEXED] += hm.size();
                    *in_consumed = behind == in_len ? in_len : covered;
                    return ZNGAMD_OK;
                }
                if (all_ok) {                        // the indexed members are a prefix of the buffer: their bytes stay where they are
                    c->paths[ZNGAMD_PATH_INDEXED] += hm.size();
                    op = total; members = (uint32_t)hm.size(); pos = behind;
                    prefix_done = true;
                }
                // anything unexpected (foreign 'ZA' field, stored blocks, corruption): the sequential
                // reader below decides, member by member
            }
        }
    }
    // second fast path: BGZF-style members ('B','C' subfield = block size - 1).  The member table comes from a
    // header hop over the host copy; all members are then decoded in one launch, one wavefront each.
    if (!prefix_done) {
        std::vector<ZaMember> hm; uint64_t total = 0;
        uint64_t covered = 0;
        if (hop_bgzf(in, in_len, hm, &total, partial, &covered) && hm.size() > 1) {
            if (total > out_cap) { *out_len = total; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
            const uint32_t n = (uint32_t)hm.size();
            HIPCHK(c, c->members.ensure(n)); HIPCHK(c, c->mstatus.ensure(n));
            HIPCHK(c, hipMemcpyAsync(c->members.p, hm.data(), (size_t)n * sizeof(ZaMember), hipMemcpyHostToDevice, c->stream));
            { ProfScope ps(c, ZNGAMD_K_INFLATE);
              hipLaunchKernelGGL(za_k_inflate_serial_members, dim3(n), dim3(64), 0, c->stream, c->st_in.p, in_len, c->members.p,
                                 c->st_out.p, out_cap, c->d_crc_table, c->d_x8k, c->mstatus.p); }
            HIPCHK(c, hipGetLastError());
            std::vector<int32_t> st(n);
            HIPCHK(c, hipMemcpyAsync(st.data(), c->mstatus.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            prof_collect(c);
            uint32_t good = 0;
            while (good < n && st[good] == ZA_I_OK) good++;
            const uint64_t produced = good < n ? hm[good].out_off : total;
            if (produced) { const int rc_ = d2h_payload(c, out, c->st_out.p, produced); if (rc_) return rc_; }
            *out_len = produced;
            if (n_members) *n_members = good;
            c->paths[ZNGAMD_PATH_BGZF] += good;
            if (good == n) { *in_consumed = covered; return ZNGAMD_OK; }
            const int code = st[good];
            if (code == ZA_I_CRC) { c->err = "CRC check failed"; return ZNGAMD_E_GZ_CRC; }
            if (code == ZA_I_LENGTH) { c->err = "Incorrect length of data produced"; return ZNGAMD_E_GZ_LENGTH; }
            if (code == ZA_I_INPUT) return ZNGAMD_E_GZ_TRUNC;
            c->err = "invalid deflate data";
            return ZNGAMD_DATA_ERROR;
        }
    }
    // third fast path: a run of small ordinary members at the front (see hop_plain_members); the loop below continues behind it
    if (!prefix_done) {
        std::vector<ZaMember> hm; std::vector<uint64_t> hpos; uint64_t total = 0, covered = 0;
        r = hop_plain_members(c, in, in_len, hm, hpos, &total, &covered);
        if (r) return r;
        if (hm.size() >= 2) {
            if (total > out_cap) { *out_len = total; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
            const uint32_t n = (uint32_t)hm.size();
            HIPCHK(c, c->members.ensure(n)); HIPCHK(c, c->mstatus.ensure(n));
            HIPCHK(c, hipMemcpyAsync(c->members.p, hm.data(), (size_t)n * sizeof(ZaMember), hipMemcpyHostToDevice, c->stream));
            { ProfScope ps(c, ZNGAMD_K_INFLATE);
              hipLaunchKernelGGL(za_k_inflate_serial_members, dim3(n), dim3(64), 0, c->stream, c->st_in.p, in_len, c->members.p,
                                 c->st_out.p, out_cap, c->d_crc_table, c->d_x8k, c->mstatus.p); }
            HIPCHK(c, hipGetLastError());
            std::vector<int32_t> st(n);
            HIPCHK(c, hipMemcpyAsync(st.data(), c->mstatus.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            prof_collect(c);
            uint32_t good = 0;
            while (good < n && st[good] == ZA_I_OK) good++;
            if (good) {                                   // the loop takes over at the first member that did not check out
                op = good < n ? hm[good].out_off : total;
                members = good;
                c->paths[ZNGAMD_PATH_BGZF] += good;
                pos = good == n ? covered : hpos[good];
            }
        }
    }
    for (;;) {
        if (pos == in_len) break;
        uint64_t doff = 0; bool za = false; uint32_t hl = 0;
        r = parse_gzip_header(in, in_len, pos, &doff, &za, &hl);
        if (r == ZNGAMD_E_GZ_TRUNC && partial) break;                     // incomplete header: wait for more input
        if (r) { ret = r; break; }
        if (hl) {
            uint32_t hc = 0;
            r = checksum_dev(c, c->st_in.p + pos, hl, &hc, nullptr);
            if (r) return r;
            const uint32_t want = in[pos + hl] | (in[pos + hl + 1] << 8);
            if ((hc & 0xFFFFu) != want) { ret = ZNGAMD_E_GZ_HCRC; break; }
        }
        ZaInfResult res;
        bool chunked = false;
        {   // chunk-parallel decode where the stream offers enough block boundaries, else one wavefront.
            // The finder and the count pass look at everything they are given: behind the first member they get a
            // bounded piece (twice the previous member, at least 4 MiB; four times more whenever the member proves to be
            // longer) instead of the whole rest of the file, which made a file of N large members cost N^2 / 2 passes.
            uint64_t clen = 0, cused = 0;
            ChunkInfo ci;
            const uint64_t rest = in_len - doff;
            uint64_t piece = rest;
            if (members > 0) piece = std::min<uint64_t>(rest, std::max<uint64_t>(4ull << 20, 2 * last_member_bytes));
            int cr;
            if (index && members == 0 && op == 0 && pos == 0) {
                // the file's first member with the writer's index (zngamd_gz_state.index): its units inside the window side by side --
                // all of the member if the window holds it, else as far as the window goes (the state then continues it)
                StreamRun run; uint64_t needed = 0;
                HIPCHK(c, c->st_out.ensure(out_cap + 64));
                const int ri = decode_stream_prefix_indexed(c, (ZaFileIndex *)index, 0, in + doff, c->st_in.p + doff, in_len - doff, nullptr, 0,
                                                            c->st_out.p, out_cap, partial && st, &run, &needed);
                if (ri == ZNGAMD_BUF_ERROR) { *out_len = needed; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
                if (ri < 0) return ri;
                if (ri == 1 && (run.status == ZA_I_END || (partial && st && run.out_len))) {
                    zngamd_gz_state tmp_state; zngamd_gz_state *ss = st;
                    if (!ss) { memset(&tmp_state, 0, sizeof tmp_state); ss = &tmp_state; }
                    ss->in_member = 0; ss->start_bit = 0; ss->crc = 0; ss->window_len = 0; ss->out_total = 0;
                    return stream_finish(c, ss, in, in_len, doff, !partial, run, 0, out, out_cap, out_len, n_members, in_consumed);
                }
            }
            const bool may_continue = partial && st && members == 0 && op == 0 && pos == 0;
            ChunkOpts copts; copts.allow_cut = may_continue;
            for (;;) {
                cr = inflate_chunked_dev(c, c->st_in.p + doff, piece, c->st_out.p + op, out_cap - op, &clen, &cused, copts, &ci);
                if (piece < rest && ci.cut) { piece = std::min<uint64_t>(rest, piece * 4); continue; }      // the member is longer than the piece: more
                break;
            }
            if (may_continue && cr == 0 && !ci.ended) {           // the member goes on behind the window: its complete blocks are out
                st->in_member = 0; st->start_bit = 0; st->crc = 0; st->window_len = 0; st->out_total = 0;
                StreamRun run; run.status = ZA_I_INPUT; run.out_len = clen; run.in_bits = ci.end_bit; run.chunked = true;
                return stream_finish(c, st, in, in_len, doff, false, run, 0, out, out_cap, out_len, n_members, in_consumed);
            }
            const bool cut = ci.cut;
            if (cr < 0 && cr != ZNGAMD_BUF_ERROR) return cr;
            if (cr == ZNGAMD_BUF_ERROR) { *out_len = op + clen; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
            if (cr == 0) { chunked = true; res.status = ZA_I_END; res.out_len = clen; res.in_bits = cused * 8; res.block_bits = 0; res.block_out = 0; }
            else if (cut && partial) break;                               // the member runs past the window: wait for more input
            else {
                r = inflate_serial_dev(c, c->st_in.p + doff, in_len - doff, nullptr, 0, c->st_out.p + op, out_cap - op, &res);
                if (r) return r;
            }
        }
        if (res.status == ZA_I_INPUT && partial) break;
        if (res.status != ZA_I_END) {
            *out_len = op + res.out_len;
            if (res.status == ZA_I_OUTFULL) ret = ZNGAMD_BUF_ERROR;
            else if (res.status == ZA_I_INPUT) ret = ZNGAMD_E_GZ_TRUNC;
            else ret = map_status(res.status);
            if (ret == ZNGAMD_BUF_ERROR || ret == ZNGAMD_E_GZ_TRUNC) {
                if (op + res.out_len) { const int rc_ = d2h_payload(c, out, c->st_out.p, op + res.out_len); if (rc_) return rc_; }
            }
            c->err = "gzip member did not end"; return ret;
        }
        uint32_t crc = 0;
        r = checksum_dev(c, c->st_out.p + op, res.out_len, &crc, nullptr);
        if (r) return r;
        uint64_t cur = doff + ((res.in_bits + 7) >> 3);
        if (in_len - cur < 8) { if (!partial) ret = ZNGAMD_E_GZ_TRUNC; break; }
        const uint32_t tc = in[cur] | (in[cur + 1] << 8) | (in[cur + 2] << 16) | ((uint32_t)in[cur + 3] << 24);
        const uint32_t tl = in[cur + 4] | (in[cur + 5] << 8) | (in[cur + 6] << 16) | ((uint32_t)in[cur + 7] << 24);
        if (tc != crc) { ret = ZNGAMD_E_GZ_CRC; char b[96]; snprintf(b, sizeof b, "CRC check failed %u != %u", tc, crc); c->err = b; break; }
        if (tl != (uint32_t)(res.out_len & 0xFFFFFFFFull)) { ret = ZNGAMD_E_GZ_LENGTH; c->err = "Incorrect length of data produced"; break; }
        cur += 8; op += res.out_len; members++;
        last_member_bytes = cur - pos;
        c->paths[chunked ? ZNGAMD_PATH_CHUNKED : ZNGAMD_PATH_SEQUENTIAL]++;
        while (cur < in_len && in[cur] == 0) cur++;
        pos = cur;
    }
    if (op) { const int rc_ = d2h_payload(c, out, c->st_out.p, op); if (rc_) return rc_; }
    *out_len = op;
    *in_consumed = pos;
    if (n_members) *n_members = members;
    return ret;
}

int zngamd_gunzip(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members)
try {
    if (!c || (!in && in_len) || (!out && out_cap) || !out_len) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    uint64_t used = 0;
    return gunzip_impl(c, in, in_len, false, out, out_cap, out_len, n_members, &used);
} ZA_ABI_GUARD

// ---- one member that is larger than the caller's window: decoded block-wise across calls -------------------

// Decodes the deflate stream that starts at bit `start_bit` of d_def[0] with `dict_len` bytes of history.  With allow_cut a
// stream that runs past the buffer is decoded up to its last complete block: status ZA_I_INPUT, in_bits = where the next
// block header starts.
static int decode_stream_prefix(zngamd_ctx *c, const uint8_t *d_def, uint64_t avail, uint32_t start_bit, const uint8_t *d_dict, uint32_t dict_len,
                                uint8_t *d_out, uint64_t out_room, bool allow_cut, StreamRun *run, uint64_t *needed)
{
    ChunkOpts o; o.start_bit = start_bit; o.d_dict = d_dict; o.dict_len = dict_len; o.allow_cut = allow_cut;
    ChunkInfo ci;
    uint64_t clen = 0, cused = 0;
    const int cr = inflate_chunked_dev(c, d_def, avail, d_out, out_room, &clen, &cused, o, &ci);
    if (cr < 0 && cr != ZNGAMD_BUF_ERROR) return cr;
    if (cr == ZNGAMD_BUF_ERROR) { *needed = clen; return ZNGAMD_BUF_ERROR; }
    if (cr == 0) { run->status = ci.ended ? ZA_I_END : ZA_I_INPUT; run->out_len = clen; run->in_bits = ci.end_bit; run->chunked = true; return ZNGAMD_OK; }
    ZaInfResult res;
    int r = inflate_serial_dev(c, d_def, avail, d_dict, dict_len, d_out, out_room, &res, start_bit);
    if (r) return r;
    run->chunked = false;
    if (res.status == ZA_I_OUTFULL) { *needed = 0; return ZNGAMD_BUF_ERROR; }
    if (res.status == ZA_I_INPUT && allow_cut) { run->status = ZA_I_INPUT; run->out_len = res.block_out; run->in_bits = res.block_bits; }   // the last block header entered
    else { run->status = res.status; run->out_len = res.out_len; run->in_bits = res.in_bits; }
    return ZNGAMD_OK;
}

// A window of a member whose units the writer indexed (zngamd_gz_state.index): the whole units inside the window, decoded side by
// side.  1 = done (*run says how far), 0 = not applicable here (the caller decodes the window without the index), < 0 = error.
static int decode_stream_prefix_indexed(zngamd_ctx *c, ZaFileIndex *ix, uint64_t out_total, const uint8_t *h_def, const uint8_t *d_def, uint64_t avail,
                                        const uint8_t *d_dict, uint32_t dict_len, uint8_t *d_out, uint64_t out_room, bool allow_cut,
                                        StreamRun *run, uint64_t *needed)
{
    if (!ix || ix->dead || ix->device != c->device) return 0;
    const size_t k0 = (size_t)(std::lower_bound(ix->cum_out.begin(), ix->cum_out.end(), out_total) - ix->cum_out.begin());
    if (k0 > ix->n || ix->cum_out[k0] != out_total) return 0;               // the reader does not stand at a unit's start
    size_t k1 = k0;
    while (k1 < ix->n && ix->cum_in[k1 + 1] - ix->cum_in[k0] <= avail && ix->cum_out[k1 + 1] - out_total <= out_room) k1++;
    if (k1 < ix->n && ix->cum_in[k1 + 1] - ix->cum_in[k0] <= avail && !allow_cut) {       // (room, not input, ended the run, and nothing may be left over)
        *needed = ix->cum_out[ix->n] - out_total; return ZNGAMD_BUF_ERROR;
    }
    uint64_t olen = 0;
    const uint64_t used = ix->cum_in[k1] - ix->cum_in[k0];
    if (k1 > k0) {
        const int r = inflate_units_core(c, d_def, used, ix->in_len.data() + k0, ix->out_len.data() + k0, (uint32_t)(k1 - k0),
                                         ix->d_rows + k0 * ZA_CIDX_STRIDE, d_dict, dict_len, d_out, out_room, &olen);
        // (nothing was handed out: the window is decoded again, the ordinary way -- also behind "invalid data", which a wrong index
        // entry looks like to the lane it sends into the middle of a code; real damage is then reported by the decoder that needs no index)
        if (r == ZNGAMD_E_INDEX || r == ZNGAMD_DATA_ERROR) { ix->dead = true; return 0; }
        if (r != ZNGAMD_STREAM_END) return r;
    }
    run->chunked = true; run->out_len = olen;
    if (k1 == ix->n) {
        // behind the last unit the writer closes the stream with an empty final block (03 00: gzip_ng_threaded.py:333)
        if (avail >= used + 2 && h_def[used] == 0x03 && h_def[used + 1] == 0x00) { run->status = ZA_I_END; run->in_bits = (used + 2) * 8ull; return 1; }
        if (avail >= used + 2) { ix->dead = true; if (k1 == k0) return 0; }  // something else follows: the rest without the index
    }
    run->status = ZA_I_INPUT; run->in_bits = used * 8ull;
    return 1;
}

static int stream_step(zngamd_ctx *c, zngamd_gz_state *st, const uint8_t *in, uint64_t in_len, uint64_t doff, bool last,
                       uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members, uint64_t *in_consumed)
{
    const uint64_t front = ZA_WIN + 64;                     // staging: [history][pad][window of the stream]
    int r = stage_in(c, in, in_len, front);
    if (r) return r;
    const uint32_t dl = st->window_len > ZA_WIN ? (uint32_t)ZA_WIN : st->window_len;
    if (dl) HIPCHK(c, hipMemcpyAsync(c->st_in.p, st->window, dl, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->st_out.ensure(out_cap + 64));
    StreamRun run; uint64_t needed = 0;
    r = 0;
    if (st->index && (st->start_bit & 7u) == 0u)
        r = decode_stream_prefix_indexed(c, (ZaFileIndex *)st->index, st->out_total, in + doff, c->st_in.p + front + doff, in_len - doff,
                                         c->st_in.p, dl, c->st_out.p, out_cap, !last, &run, &needed);
    if (r == 1) r = ZNGAMD_OK;
    else if (r == 0)
    r = decode_stream_prefix(c, c->st_in.p + front + doff, in_len - doff, st->start_bit & 7u, c->st_in.p, dl, c->st_out.p, out_cap, !last, &run, &needed);
    if (r == ZNGAMD_BUF_ERROR) { *out_len = needed > out_cap ? needed : out_cap; return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small"); }
    if (r) return r;
    return stream_finish(c, st, in, in_len, doff, last, run, dl, out, out_cap, out_len, n_members, in_consumed);
}

// What follows the decode of a piece of a member that is continued across windows (run: how far it got; the decoded bytes are
// at st_out): CRC fold, trailer check at the end of the member, hand-out, and the state the next window continues with.
static int stream_finish(zngamd_ctx *c, zngamd_gz_state *st, const uint8_t *in, uint64_t in_len, uint64_t doff, bool last, const StreamRun &run,
                         uint32_t dl, uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members, uint64_t *in_consumed)
{
    int r;
    if (run.status != ZA_I_END && run.status != ZA_I_INPUT) { c->err = "invalid deflate data"; return map_status(run.status); }
    if (run.status == ZA_I_INPUT && last) {
        if (run.out_len) { const int rc_ = d2h_payload(c, out, c->st_out.p, run.out_len); if (rc_) return rc_; }
        *out_len = run.out_len; c->err = "gzip member did not end"; return ZNGAMD_E_GZ_TRUNC;
    }
    if (run.status == ZA_I_INPUT && run.out_len == 0 && run.in_bits <= (st->start_bit & 7u)) return ZNGAMD_OK;   // not one whole block yet
    uint32_t cnew = 0;
    r = checksum_dev(c, c->st_out.p, run.out_len, &cnew, nullptr);
    if (r) return r;
    const uint32_t crc = st->out_total ? zngamd_crc32_combine(st->crc, cnew, run.out_len) : cnew;
    if (run.status == ZA_I_END) {
        uint64_t cur = doff + ((run.in_bits + 7) >> 3);
        if (in_len - cur < 8) { if (last) return ZNGAMD_E_GZ_TRUNC; return ZNGAMD_OK; }      // trailer not in the window yet
        const uint32_t tc = in[cur] | (in[cur + 1] << 8) | (in[cur + 2] << 16) | ((uint32_t)in[cur + 3] << 24);
        const uint32_t tl = in[cur + 4] | (in[cur + 5] << 8) | (in[cur + 6] << 16) | ((uint32_t)in[cur + 7] << 24);
        if (run.out_len) { const int rc_ = d2h_payload(c, out, c->st_out.p, run.out_len); if (rc_) return rc_; }
        *out_len = run.out_len;
        if (tc != crc) { char b[96]; snprintf(b, sizeof b, "CRC check failed %u != %u", tc, crc); c->err = b; return ZNGAMD_E_GZ_CRC; }
        if (tl != (uint32_t)((st->out_total + run.out_len) & 0xFFFFFFFFull)) { c->err = "Incorrect length of data produced"; return ZNGAMD_E_GZ_LENGTH; }
        cur += 8;
        while (cur < in_len && in[cur] == 0) cur++;
        *in_consumed = cur;
        if (n_members) *n_members = 1;
        c->paths[run.chunked ? ZNGAMD_PATH_CHUNKED : ZNGAMD_PATH_SEQUENTIAL]++;
        st->in_member = 0; st->start_bit = 0; st->crc = 0; st->window_len = 0; st->out_total = 0;
        st->index = nullptr;                         // (the index was this member's; the handle stays the caller's)
        return ZNGAMD_OK;
    }
    // the member goes on: hand out the complete blocks, remember where and with what history to continue
    if (run.out_len) { const int rc_ = d2h_payload(c, out, c->st_out.p, run.out_len); if (rc_) return rc_; }
    if (run.out_len >= ZA_WIN) { memcpy(st->window, out + run.out_len - ZA_WIN, ZA_WIN); st->window_len = ZA_WIN; }
    else {
        const uint32_t keep = std::min<uint32_t>(dl, (uint32_t)(ZA_WIN - run.out_len));
        memmove(st->window, st->window + (dl - keep), keep);
        memcpy(st->window + keep, out, run.out_len);
        st->window_len = keep + (uint32_t)run.out_len;
    }
    st->crc = crc; st->out_total += run.out_len; st->start_bit = (uint32_t)(run.in_bits & 7u); st->in_member = 1;
    *out_len = run.out_len;
    *in_consumed = doff + (run.in_bits >> 3);
    return ZNGAMD_OK;
}

int zngamd_gunzip_stream(zngamd_ctx *c, zngamd_gz_state *st, const uint8_t *in, uint64_t in_len, int last, uint8_t *out, uint64_t out_cap,
                         uint64_t *out_len, uint32_t *n_members, uint64_t *in_consumed)
try {
    if (!c || !st || (!in && in_len) || (!out && out_cap) || !out_len || !in_consumed) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    *out_len = 0; *in_consumed = 0;
    if (n_members) *n_members = 0;
    if (st->in_member) return stream_step(c, st, in, in_len, 0, last != 0, out, out_cap, out_len, n_members, in_consumed);
    int r = gunzip_impl(c, in, in_len, !last, out, out_cap, out_len, n_members, in_consumed, last ? nullptr : st, st->index);
    if (*in_consumed > 0 && !st->in_member) st->index = nullptr;      // (the index was the first member's, and that one is through)
    if (r != ZNGAMD_OK || last || *in_consumed > 0 || in_len == 0) return r;
    // not even the first member is complete in this window: start it and hand out the blocks that are
    uint64_t doff = 0; bool za = false; uint32_t hl = 0;
    r = parse_gzip_header(in, in_len, 0, &doff, &za, &hl);
    if (r == ZNGAMD_E_GZ_TRUNC) return ZNGAMD_OK;                      // the header itself is cut: more input first
    if (r) return r;
    if (hl) {
        int rs = stage_in(c, in, in_len);
        if (rs) return rs;
        uint32_t hc = 0;
        rs = checksum_dev(c, c->st_in.p, hl, &hc, nullptr);
        if (rs) return rs;
        const uint32_t want = in[hl] | (in[hl + 1] << 8);
        if ((hc & 0xFFFFu) != want) return ZNGAMD_E_GZ_HCRC;
    }
    st->in_member = 0; st->start_bit = 0; st->crc = 0; st->window_len = 0; st->out_total = 0;
    return stream_step(c, st, in, in_len, doff, false, out, out_cap, out_len, n_members, in_consumed);
} ZA_ABI_GUARD

int zngamd_gunzip_partial(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                          uint32_t *n_members, uint64_t *in_consumed)
try {
    if (!c || (!in && in_len) || (!out && out_cap) || !out_len || !in_consumed) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    return gunzip_impl(c, in, in_len, true, out, out_cap, out_len, n_members, in_consumed);
} ZA_ABI_GUARD

// ---- indexed member writer ---------------------------------------------------------------------
static int gzip_members_dev(zngamd_ctx *c, const uint8_t *d_in, uint64_t in_len, uint32_t block_size, int level,
                            uint8_t *d_out, uint64_t out_cap, uint64_t *out_len, uint32_t *n_members)
{
    if (block_size == 0 || block_size > ZA_MAX_UNIT) return fail(c, ZNGAMD_E_ARG, "block_size must be 1..131072");
    if (!zngamd_level_ok(level)) return fail(c, ZNGAMD_STREAM_ERROR, "Bad compression level");
    const uint64_t nb64 = in_len == 0 ? 1 : (in_len + block_size - 1) / block_size;
    if (nb64 > (1u << 26)) return fail(c, ZNGAMD_E_ARG, "too many members");
    const uint32_t nb = (uint32_t)nb64;
    std::vector<ZaUnit> hu(nb);
    for (uint32_t b = 0; b < nb; b++) {
        ZaUnit u; u.in_off = (uint64_t)b * block_size; u.in_len = (uint32_t)std::min<uint64_t>(block_size, in_len - u.in_off);
        u.dict_len = 0; u.flags = ZA_FLAG_FINAL | ZA_FLAG_FLATHDR; u.flags |= (uint32_t)za_seg_shift_for(u.in_len, u.flags) << 8; u.block = b; hu[b] = u;
    }
    HIPCHK(c, c->st_slots.ensure((size_t)nb * ZNGAMD_SLOT_STRIDE)); HIPCHK(c, c->st_len.ensure(nb)); HIPCHK(c, c->st_crc.ensure(nb));
    int r = deflate_units_dev(c, d_in, in_len, hu, level, c->st_slots.p, c->st_len.p, c->st_crc.p);
    if (r) return r;
    HIPCHK(c, c->st_off.ensure(nb));
    uint64_t total = 0;
    r = gather_dev(c, c->st_slots.p, c->st_len.p, nb, ZA_MEMBER_FIXED + 8, d_out, 0, out_cap, c->st_off.p, &total, false, c->units.p);
    if (r) return r;
    const int lv = level == -1 ? 6 : level;
    const uint8_t xfl = lv == 9 ? 2 : lv == 1 ? 4 : 0;
    { ProfScope ps(c, ZNGAMD_K_GATHER);
      hipLaunchKernelGGL(za_k_assemble_members, dim3(nb), dim3(256), 0, c->stream, c->st_slots.p, (uint32_t)ZNGAMD_SLOT_STRIDE, c->st_len.p,
                         c->st_crc.p, c->cidx.p, c->units.p, c->st_off.p, d_out, xfl); }
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> st(nb);
    HIPCHK(c, hipMemcpyAsync(st.data(), c->status.p, nb * 4ull, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    for (uint32_t i = 0; i < nb; i++) if (st[i]) return fail(c, ZNGAMD_E_OVERFLOW, "unit overflowed its slot");
    *out_len = total;
    if (n_members) *n_members = nb;
    return ZNGAMD_OK;
}

int zngamd_gzip_members_dev(zngamd_ctx *c, const void *d_in, uint64_t in_len, uint32_t block_size, int level, void *d_out,
                            uint64_t out_cap, uint64_t *out_len, uint32_t *n_members)
try {
    if (!c || (!d_in && in_len) || !d_out || !out_len) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return gzip_members_dev(c, (const uint8_t *)d_in, in_len, block_size, level, (uint8_t *)d_out, out_cap, out_len, n_members);
} ZA_ABI_GUARD

int zngamd_gzip_members(zngamd_ctx *c, const uint8_t *in, uint64_t in_len, uint32_t block_size, int level, uint8_t *out,
                        uint64_t out_cap, uint64_t *out_len)
try {
    if (!c || (!in && in_len) || !out || !out_len) return ZNGAMD_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    int r = stage_in(c, in, in_len);
    if (r) return r;
    const uint64_t nb = in_len == 0 ? 1 : (in_len + (block_size ? block_size : 1) - 1) / (block_size ? block_size : 1);
    const uint64_t bound = in_len + in_len / 32 + nb * (ZA_MEMBER_FIXED + 8 + 600);      // worst case stored + index + flat header
    HIPCHK(c, c->st_aux.ensure(bound));
    uint64_t total = 0;
    r = gzip_members_dev(c, c->st_in.p, in_len, block_size, level, c->st_aux.p, bound, &total, nullptr);
    if (r) return r;
    *out_len = total;
    if (total > out_cap) return fail(c, ZNGAMD_BUF_ERROR, "output buffer too small");
    if (total) { const int rc_ = d2h_payload(c, out, c->st_aux.p, total); if (rc_) return rc_; }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// multi-GPU exchange over RCCL (one process per GPU).  librccl is loaded with dlopen on first use.
// ---------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi *rccl_api(std::string *why)
{
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (api.lib) return &api;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { if (why) *why = std::string("librccl not found: ") + dlerror(); return nullptr; }
#define ZA_SYM(field, name) do { api.field = (decltype(api.field))dlsym(h, name); if (!api.field) { if (why) *why = std::string("librccl lacks ") + name; return nullptr; } } while (0)
    ZA_SYM(GetUniqueId, "ncclGetUniqueId"); ZA_SYM(CommInitRank, "ncclCommInitRank"); ZA_SYM(CommDestroy, "ncclCommDestroy");
    ZA_SYM(CommCount, "ncclCommCount");
    ZA_SYM(AllGather, "ncclAllGather"); ZA_SYM(AllReduce, "ncclAllReduce"); ZA_SYM(Send, "ncclSend"); ZA_SYM(Recv, "ncclRecv");
    ZA_SYM(GroupStart, "ncclGroupStart"); ZA_SYM(GroupEnd, "ncclGroupEnd"); ZA_SYM(GetErrorString, "ncclGetErrorString");
#undef ZA_SYM
    api.lib = h;
    return &api;
}

struct zngamd_comm {
    zngamd_ctx *ctx = nullptr; RcclApi *api = nullptr; ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr; hipEvent_t ev = nullptr;
    uint64_t *d_rec = nullptr;           // [3 * world] gathered layout records, [3] own record behind them
    double *d_val = nullptr;
    std::string err;
};
#define NCCLCHK(cm, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    (cm)->err = std::string(#call) + ": " + (cm)->api->GetErrorString(r_); return ZNGAMD_E_HIP; } } while (0)
#define HIPCHKM(cm, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (cm)->err = std::string(#call) + ": " + hipGetErrorString(e_); return ZNGAMD_E_HIP; } } while (0)

extern "C" {

int zngamd_comm_unique_id(uint8_t id[ZNGAMD_COMM_ID_BYTES])
try {
    static_assert(sizeof(ncclUniqueId) == ZNGAMD_COMM_ID_BYTES, "unique id size");
    if (!id) return ZNGAMD_E_ARG;
    RcclApi *a = rccl_api(nullptr);
    if (!a) return ZNGAMD_E_HIP;
    ncclUniqueId u;
    if (a->GetUniqueId(&u) != ncclSuccess) return ZNGAMD_E_HIP;
    memcpy(id, &u, sizeof u);
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_comm_create(zngamd_ctx *c, const uint8_t id[ZNGAMD_COMM_ID_BYTES], int rank, int world, zngamd_comm **out)
try {
    if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return ZNGAMD_E_ARG;
    *out = nullptr;
    std::string why;
    RcclApi *a = rccl_api(&why);
    if (!a) { c->err = why; return ZNGAMD_E_HIP; }
    zngamd_comm *m = new zngamd_comm();
    m->ctx = c; m->api = a; m->rank = rank; m->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&m->d_rec, (size_t)(3 * world + 3) * 8) != hipSuccess || hipMalloc((void **)&m->d_val, 16) != hipSuccess) {
        c->err = "comm: HIP resources"; zngamd_comm_destroy(m); return ZNGAMD_E_HIP;
    }
    const ncclResult_t r = a->CommInitRank(&m->comm, world, u, rank);
    if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + a->GetErrorString(r); m->comm = nullptr; zngamd_comm_destroy(m); return ZNGAMD_E_HIP; }
    *out = m;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

void zngamd_comm_destroy(zngamd_comm *m)
{
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm) (void)m->api->CommDestroy(m->comm);
    if (m->d_rec) (void)hipFree(m->d_rec);
    if (m->d_val) (void)hipFree(m->d_val);
    if (m->ev) (void)hipEventDestroy(m->ev);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

const char *zngamd_comm_last_error(zngamd_comm *m) { return m ? m->err.c_str() : "no communicator"; }

// the number of ranks RCCL itself reports for the communicator (ncclCommCount), not what the launcher's environment says
int zngamd_comm_count(zngamd_comm *m, int *ranks)
try {
    if (!m || !ranks) return ZNGAMD_E_ARG;
    *ranks = 0;
    NCCLCHK(m, m->api->CommCount(m->comm, ranks));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_comm_layout(zngamd_comm *m, uint64_t local_len, uint32_t local_crc, uint64_t local_ulen, uint64_t *sizes,
                       uint64_t *my_off, uint64_t *total, uint32_t *whole_crc, uint64_t *whole_ulen)
try {
    if (!m || !sizes || !my_off || !total) return ZNGAMD_E_ARG;
    HIPCHKM(m, hipSetDevice(m->ctx->device));
    const uint64_t mine[3] = {local_len, (uint64_t)local_crc, local_ulen};
    uint64_t *d_mine = m->d_rec + 3 * (size_t)m->world;
    HIPCHKM(m, hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, m->stream));
    NCCLCHK(m, m->api->AllGather(d_mine, m->d_rec, 3, ncclUint64, m->comm, m->stream));
    std::vector<uint64_t> rec(3 * (size_t)m->world);
    HIPCHKM(m, hipMemcpyAsync(rec.data(), m->d_rec, rec.size() * 8, hipMemcpyDeviceToHost, m->stream));
    HIPCHKM(m, hipStreamSynchronize(m->stream));
    uint64_t off = 0, tot = 0, ul = 0;
    uint32_t crc = 0;
    for (int r = 0; r < m->world; r++) {
        sizes[r] = rec[3 * r];
        if (r < m->rank) off += rec[3 * r];
        tot += rec[3 * r];
        crc = r ? zngamd_crc32_combine(crc, (uint32_t)rec[3 * r + 1], rec[3 * r + 2]) : (uint32_t)rec[1];
        ul += rec[3 * r + 2];
    }
    *my_off = off; *total = tot;
    if (whole_crc) *whole_crc = crc;
    if (whole_ulen) *whole_ulen = ul;
    return ZNGAMD_OK;
} ZA_ABI_GUARD

uint64_t zngamd_comm_offsets(const uint64_t *sizes, int world, uint64_t *offs)
{
    uint64_t run = 0;
    for (int r = 0; r < world; r++) { if (offs) offs[r] = run; run += sizes ? sizes[r] : 0; }
    return run;
}

int zngamd_comm_allgather_stream(zngamd_comm *m, const void *d_local, const uint64_t *sizes, void *d_stream, uint64_t stream_cap)
try {
    if (!m || !d_local || !sizes || !d_stream) return ZNGAMD_E_ARG;
    HIPCHKM(m, hipSetDevice(m->ctx->device));
    std::vector<uint64_t> offs((size_t)m->world);
    const uint64_t total = zngamd_comm_offsets(sizes, m->world, offs.data()), my_off = offs[(size_t)m->rank];
    if (total > stream_cap) { m->err = "stream buffer too small"; return ZNGAMD_BUF_ERROR; }
    // the slices are final once the work queued on the context's stream so far is done
    HIPCHKM(m, hipEventRecord(m->ev, m->ctx->stream));
    HIPCHKM(m, hipStreamWaitEvent(m->stream, m->ev, 0));
    uint8_t *dst = (uint8_t *)d_stream;
    if (sizes[m->rank] && dst + my_off != (const uint8_t *)d_local)
        HIPCHKM(m, hipMemcpyAsync(dst + my_off, d_local, sizes[m->rank], hipMemcpyDeviceToDevice, m->stream));
    if (m->world > 1) {
        NCCLCHK(m, m->api->GroupStart());
        // (a call that fails inside the group must not leave the group open: it is closed first, then the first error is reported)
        ncclResult_t bad = ncclSuccess;
        const char *what = "";
        for (int r = 0; r < m->world && bad == ncclSuccess; r++) {
            if (r != m->rank) {
                if (sizes[m->rank]) { bad = m->api->Send(d_local, sizes[m->rank], ncclUint8, r, m->comm, m->stream); what = "ncclSend"; }
                if (bad == ncclSuccess && sizes[r]) { bad = m->api->Recv(dst + offs[(size_t)r], sizes[r], ncclUint8, r, m->comm, m->stream); what = "ncclRecv"; }
            }
        }
        const ncclResult_t ended = m->api->GroupEnd();
        if (bad != ncclSuccess) { m->err = std::string(what) + ": " + m->api->GetErrorString(bad); return ZNGAMD_E_HIP; }
        NCCLCHK(m, ended);
    }
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_comm_wait(zngamd_comm *m)
try {
    if (!m) return ZNGAMD_E_ARG;
    HIPCHKM(m, hipStreamSynchronize(m->stream));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_comm_max_f64(zngamd_comm *m, double *value)
try {
    if (!m || !value) return ZNGAMD_E_ARG;
    HIPCHKM(m, hipSetDevice(m->ctx->device));
    HIPCHKM(m, hipMemcpyAsync(m->d_val, value, 8, hipMemcpyHostToDevice, m->stream));
    NCCLCHK(m, m->api->AllReduce(m->d_val, m->d_val + 1, 1, ncclFloat64, ncclMax, m->comm, m->stream));
    HIPCHKM(m, hipMemcpyAsync(value, m->d_val + 1, 8, hipMemcpyDeviceToHost, m->stream));
    HIPCHKM(m, hipStreamSynchronize(m->stream));
    return ZNGAMD_OK;
} ZA_ABI_GUARD

int zngamd_comm_barrier(zngamd_comm *m)
try {
    double v = 0;
    return zngamd_comm_max_f64(m, &v);
} ZA_ABI_GUARD

}  // extern "C"

#include "zng_stream.hip"

#ifdef ZA_PLAN_STATS
// profiling build only: the plan kernel's phase clocks (za_deflate.hip), read and cleared
extern "C" int zngamd_debug_plan_stats(unsigned long long *out16)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(za_plan_stat), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(za_plan_stat), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ZA_CH_STATS
// profiling build only: the chain kernel's clocks (za_deflate.hip), read and cleared
extern "C" int zngamd_debug_ch_stats(unsigned long long *out32)
{
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(za_ch_stat), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(za_ch_stat), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ZA_PS_STATS
// profiling build only (profiles/ps_stats.sh): counters of the parallel sweep, read and cleared
extern "C" int zngamd_debug_ps_stats(unsigned long long *out24)
{
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(za_ps_stat), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(za_ps_stat), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
