// common.h -- context, device buffers, error handling, small device helpers.
// Part of libshepseg_hip.so (gfx950 only; wavefront = 64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include <condition_variable>
#include "../../include/shepseg_hip.h"

#define WAVE 64
#define NULL_LAB 0xFFFFFFFFu   // CCL label of a null pixel
#define VIS_FLAG 0x80000000u   // set on labels assigned by the depth-first splitter
#define MAX_CLUMP_SIZE 10000u  // shepseg.py:481
#define PROF_N 16
#define PROF_POOL 32
#define SHP_PINNED_BYTES (1u << 20)   // pinned host staging per context
// The last 64 words of the staging block are MIRRORS: scalars the host needs after a phase (counts,
// scan totals) are stored there by the kernel that produces them (system-scope stores to the mapped
// pinned block) instead of by a copy command queued behind it -- under load every command on a
// stream costs 40-80 us, and there were ten such copies per tile.
#define PIN_MIRROR (SHP_PINNED_BYTES / 4u - 64u)
enum { MIR_NBIG = 0, MIR_RELABEL = 4, MIR_RUNS = 5 };
#define MIRROR_STORE(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct StageTimer {
    hipEvent_t a = nullptr, b = nullptr;
};

struct shp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // side stream for fork/join inside one call (created lazily)
    int stream_priority = 0;
    hipEvent_t evfork = nullptr, evjoin = nullptr;
    std::string err;
    std::vector<DevBuf *> bufs;
    // named workspace buffers (grow-only)
    DevBuf img, clus, lab, seg, aux, aux2, stack, scan_tmp, sort_k0, sort_k1, sort_v1, sort_hist,
        pix, segsz, origsz, off, ssum, chnext, chtail, mergeto, tcount, toff, tfill, tlist, tsorted,
        small, cen, fit_x, fit_lab, fit_part, fit_lb, big, srclist, tgtlist, bigbits, singles, dbg, snap;
    uint32_t *h_pinned = nullptr;   // SHP_PINNED_BYTES of pinned host staging (small transfers)
    int fit_path = 0;               // last k-means fit: 0 the fast (Lloyd) path, 1 the reference's Elkan path
    double *h_fit = nullptr;        // pinned, grow-only: the centred k-means sample
    size_t h_fit_cap = 0;
    hipEvent_t ev[16] = {};
    double timings[8] = {};
    // per-kernel device-time accounting (HIP events on this stream), see PROF_* below
    double prof_ms[PROF_N] = {};
    uint64_t prof_cnt[PROF_N] = {};
    hipEvent_t prof_ev[PROF_POOL][2] = {};
    int prof_id[PROF_POOL] = {};
    int prof_used = 0;
    uint32_t *scan_ctr = nullptr;   // device word, zero between scans: arrival counter of k_scan_local
    int gated = 0;      // this call takes part in the fill gate (tiled driver's worker calls)
    bool fill_held = false;
    bool gate_held = false;
    bool shared = false;    // a worker context without a stream of its own: it borrows one per phase
    int borrowed = 0;       // 0 = none (ctx->stream is the pool's idle stream), 1 = a fill stream, 2 = a walker stream
};

// Fill gate.  A tile alternates between phases that fill the GPU (HBM-bound passes over the whole
// tile: ~3.4 ms of device time) and phases that are one long dependent chain on a few wavefronts
// (the depth-first replay, the pass loop: ~35 ms).  Twenty worker streams started together stay
// in step -- every tile fills at once, then every tile waits at once, and the filling capacity
// idles a quarter of the time.  The gate lets only a few tiles be in a filling phase at any time
// (the others queue on the host), which staggers them: while some fill, the rest are in their
// chains.  Phases nearer the end of a tile go first.  SHEPSEG_FILL_MAX = 0 switches it off.
#define FILL_MAX_DEFAULT 6          // (round 4, with the shorter fit: 4 -> 6 fill and 12 -> 10 walker streams = -15 ms per step;
                                    //  fill + walker + 3 must stay <= 21 hardware queues: 7 + 12 costs 60 ms)
#define FILL_PRIOS 4
struct FillGate {
    std::mutex mu;
    std::condition_variable cv;
    int running = 0;
    int waiting[FILL_PRIOS] = {};
};
static FillGate g_fill;
static const int g_fill_max = getenv("SHEPSEG_FILL_MAX") ? atoi(getenv("SHEPSEG_FILL_MAX")) : FILL_MAX_DEFAULT;

// Stream pool.  A process gets about 24 hardware queues before the driver time-slices them, so the
// number of streams is what bounds the tiles in flight -- and a tile that owns a stream keeps it
// through its host-side waits (for the fill gate, for a pass-loop slot) as well.  A SHARED context
// has no stream of its own: inside a worker call of the tiled driver it borrows a fill stream for
// each GPU-filling phase (with the gate slot: there are as many as the gate admits) and a walker
// stream for each latency-bound kernel, and gives it back once the phase has left the GPU (every
// phase ends in a host synchronisation, so the next phase may run on any other stream).  More tiles
// than streams can then be in flight.  Outside worker calls a shared context uses the pool's idle
// stream, which all of them share.
#define WALK_STREAMS_DEFAULT 10
struct StreamPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<hipStream_t> idle[2];     // 0 = fill streams, 1 = walker streams
    int made[2] = {0, 0};
    int device = -1;
    hipStream_t misc = nullptr;
};
static StreamPool g_streams;
static const int g_walk_streams = getenv("SHEPSEG_WALK_STREAMS") ? atoi(getenv("SHEPSEG_WALK_STREAMS")) : WALK_STREAMS_DEFAULT;
static inline int stream_pool_cap(int cls)
{
    if (cls == 1) return g_walk_streams < 1 ? 1 : g_walk_streams;
    return g_fill_max > 0 ? g_fill_max : 8;
}
static inline void stream_take(shp_ctx *ctx, int cls)
{
    std::unique_lock<std::mutex> lk(g_streams.mu);
    for (;;) {
        if (!g_streams.idle[cls].empty()) {
            ctx->stream = g_streams.idle[cls].back();
            g_streams.idle[cls].pop_back();
            break;
        }
        if (g_streams.made[cls] < stream_pool_cap(cls)) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0) == hipSuccess) {
                g_streams.made[cls]++;
                ctx->stream = s;
                break;
            }
            (void)hipGetLastError();
            if (g_streams.made[cls] == 0) return;      // no stream of this class at all: stay on the idle stream
        }
        g_streams.cv.wait(lk);
    }
    ctx->borrowed = cls + 1;
}
// (the caller has synchronised the stream: the next borrower starts on an empty one)
static inline void stream_give(shp_ctx *ctx)
{
    if (!ctx->borrowed) return;
    {
        std::lock_guard<std::mutex> lk(g_streams.mu);
        g_streams.idle[ctx->borrowed - 1].push_back(ctx->stream);
        ctx->stream = g_streams.misc;
    }
    ctx->borrowed = 0;
    g_streams.cv.notify_all();
}
static inline bool stream_sharing(const shp_ctx *ctx) { return ctx->shared && ctx->gated; }

static inline bool fill_gating(const shp_ctx *ctx) { return g_fill_max > 0 && ctx->gated; }
static inline void fill_acquire(shp_ctx *ctx, int prio)
{
    if (ctx->fill_held || !(fill_gating(ctx) || stream_sharing(ctx))) return;
    if (fill_gating(ctx)) {
        std::unique_lock<std::mutex> lk(g_fill.mu);
        g_fill.waiting[prio]++;
        g_fill.cv.wait(lk, [&] {
            if (g_fill.running >= g_fill_max) return false;
            for (int p = prio + 1; p < FILL_PRIOS; p++)
                if (g_fill.waiting[p]) return false;
            return true;
        });
        g_fill.waiting[prio]--;
        g_fill.running++;
        ctx->gate_held = true;
        if (g_fill.running < g_fill_max) g_fill.cv.notify_all();
    }
    if (stream_sharing(ctx)) stream_take(ctx, 0);
    ctx->fill_held = true;
}
// sync: wait until the phase's kernels have left the GPU before letting the next tile in
static inline void fill_release(shp_ctx *ctx, bool sync)
{
    if (!ctx->fill_held) return;
    if (sync || ctx->borrowed) (void)hipStreamSynchronize(ctx->stream);
    stream_give(ctx);
    if (ctx->gate_held) {
        {
            std::lock_guard<std::mutex> lk(g_fill.mu);
            g_fill.running--;
        }
        ctx->gate_held = false;
        g_fill.cv.notify_all();
    }
    ctx->fill_held = false;
}
// a latency-bound kernel of a shared context runs on a borrowed walker stream
static inline void walk_begin(shp_ctx *ctx)
{
    if (stream_sharing(ctx) && !ctx->borrowed) stream_take(ctx, 1);
}
static inline void walk_end(shp_ctx *ctx)       // (after the caller's stream synchronisation)
{
    if (ctx->borrowed == 2) stream_give(ctx);
}
struct FillScope {          // an API call never leaves with the gate or a borrowed stream held (error paths)
    shp_ctx *ctx;
    FillScope(shp_ctx *c, int gated) : ctx(c) { c->gated = gated; }
    ~FillScope()
    {
        fill_release(ctx, false);
        if (ctx->borrowed) { (void)hipStreamSynchronize(ctx->stream); stream_give(ctx); }
        ctx->gated = 0;
    }
};

// kernels whose launch durations bench.py reports against the roofline
enum { PROF_ASSIGN = 0, PROF_CCL = 1, PROF_DFS = 2, PROF_SORT = 3, PROF_SPECTRA = 4,
       PROF_SMALL_LOOP = 5, PROF_SINGLE = 6, PROF_LABEL = 7, PROF_SEGSTATS = 8 };

static inline int prof_begin(shp_ctx *ctx, int id)
{
    if (ctx->prof_used >= PROF_POOL) return -1;
    const int slot = ctx->prof_used++;
    if (!ctx->prof_ev[slot][0]) { hipEventCreate(&ctx->prof_ev[slot][0]); hipEventCreate(&ctx->prof_ev[slot][1]); }
    ctx->prof_id[slot] = id;
    hipEventRecord(ctx->prof_ev[slot][0], ctx->stream);
    return slot;
}
static inline void prof_end(shp_ctx *ctx, int slot)
{
    if (slot >= 0) hipEventRecord(ctx->prof_ev[slot][1], ctx->stream);
}
// call after the stream has been synchronised
static inline void prof_collect(shp_ctx *ctx)
{
    for (int i = 0; i < ctx->prof_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->prof_ev[i][0], ctx->prof_ev[i][1]) == hipSuccess) {
            ctx->prof_ms[ctx->prof_id[i]] += ms;
            ctx->prof_cnt[ctx->prof_id[i]] += 1;
        }
    }
    ctx->prof_used = 0;
}

#define SHP_FAIL(ctx, code, ...)                                  \
    do {                                                          \
        char _b[512];                                             \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                    \
        (ctx)->err = _b;                                          \
        return (code);                                            \
    } while (0)

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess)                                                               \
            SHP_FAIL(ctx, SHP_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,        \
                     hipGetErrorString(_e));                                                \
    } while (0)

#define CHK(call)                    \
    do {                             \
        int _rc = (call);            \
        if (_rc != 0) return _rc;    \
    } while (0)

#define KCHK(ctx) HIPCHK(ctx, hipGetLastError())

// the side stream is only created when a call really forks: every stream costs a slot in the
// hardware queues (GPU_MAX_HW_QUEUES, 24 by default here) the worker streams are spread over
static inline int ensure_stream2(shp_ctx *ctx)
{
    if (ctx->stream2) return 0;
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, ctx->stream_priority));
    return 0;
}

static inline int buf_ensure(shp_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return 0;
    static const bool regrow_log = getenv("SHEPSEG_REGROW_LOG") != nullptr;      // diagnostic: which workspace still grows
    if (regrow_log) {
        int which = -1;
        for (size_t i = 0; i < ctx->bufs.size(); i++) if (ctx->bufs[i] == &b) which = (int)i;
        fprintf(stderr, "regrow: ctx %p buffer #%d %zu -> %zu bytes\n", (void *)ctx, which, b.cap, bytes);
    }
    if (b.p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr; b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        SHP_FAIL(ctx, SHP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return 0;
}

template <typename T> static inline T *bp(DevBuf &b) { return (T *)b.p; }

static inline unsigned grid_for(size_t n, unsigned block, unsigned cap = 0x7fffffffu)
{
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

static inline size_t dtype_size(int dtype)
{
    switch (dtype) {
    case SHP_U8: return 1;
    case SHP_I16: case SHP_U16: return 2;
    case SHP_I32: case SHP_U32: return 4;
    default: return 0;
    }
}

// ---- device helpers ---------------------------------------------------------------------
// Pixel fetch as int64 (uniform branch on dtype; all supported dtypes fit).
__device__ __forceinline__ long long ld_px(const void *__restrict__ img, int dtype, size_t i)
{
    switch (dtype) {
    case SHP_U8: return ((const uint8_t *)img)[i];
    case SHP_I16: return ((const int16_t *)img)[i];
    case SHP_U16: return ((const uint16_t *)img)[i];
    case SHP_I32: return ((const int32_t *)img)[i];
    default: return ((const uint32_t *)img)[i];
    }
}

// relaxed agent-scope load: served by L2, where the atomics land (a plain load may be answered
// by a stale line of the CU's vector L1)
#define L2LOAD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Typed pixel fetch for kernels whose inner loops issue many loads: with the pixel type a template
// parameter the loads of an unrolled loop sit in one basic block and overlap, whereas the
// run-time switch of ld_px puts every load behind its own branch.
template <int DT> struct PxT { typedef uint32_t type; };
template <> struct PxT<SHP_U8> { typedef uint8_t type; };
template <> struct PxT<SHP_I16> { typedef int16_t type; };
template <> struct PxT<SHP_U16> { typedef uint16_t type; };
template <> struct PxT<SHP_I32> { typedef int32_t type; };
template <int DT> __device__ __forceinline__ long long ld_t(const void *__restrict__ img, size_t i)
{
    return ((const typename PxT<DT>::type *)img)[i];
}
// expands STMT once per pixel type with `DT` a compile-time constant
#define DISPATCH_DTYPE(dtype, ...)                                    \
    switch (dtype) {                                                  \
    case SHP_U8: { constexpr int DT = SHP_U8; __VA_ARGS__; } break;          \
    case SHP_I16: { constexpr int DT = SHP_I16; __VA_ARGS__; } break;        \
    case SHP_U16: { constexpr int DT = SHP_U16; __VA_ARGS__; } break;        \
    case SHP_I32: { constexpr int DT = SHP_I32; __VA_ARGS__; } break;        \
    default: { constexpr int DT = SHP_U32; __VA_ARGS__; } break;             \
    }

// Where a tile's pixels live: band b, tile pixel p (row-major in a tile xs pixels wide) is element
// b * bstride + origin + (p / xs) * pitch + p % xs of the image buffer.  A compact tile image has
// pitch == xs, bstride == pixels per band, origin == 0; a window of a resident raster is read in
// place (pitch = raster width, origin = first pixel of the window) instead of being copied out.
struct ImgGeom {
    size_t bstride, origin;
    uint32_t pitch, xs;
    float inv_xs;
};
static inline ImgGeom geom_compact(size_t n, uint32_t xs)
{
    return ImgGeom{n, 0, xs, xs, xs ? 1.0f / (float)xs : 0.0f};
}
__device__ __forceinline__ size_t geom_off(const ImgGeom &g, uint32_t p)
{
    if (g.pitch == g.xs) return g.origin + p;
    // row = p / xs without an integer division: the float quotient is within 0.02 of the true one
    // (p / xs < 65536 rows, three roundings of 2^-24 each), so its floor is off by at most one
    uint32_t r = (uint32_t)((float)p * g.inv_xs);
    uint32_t rem = p - r * g.xs;
    if (rem >= g.xs) {
        if ((int32_t)rem < 0) { r -= 1u; rem += g.xs; }
        else { r += 1u; rem -= g.xs; }
    }
    return g.origin + (size_t)r * g.pitch + rem;
}

// ---- workgroup-local aggregation of per-id minima / maxima ---------------------------------------
// Atomics on one address serialise at L2, and the pixels that update one id's bounding box sit
// close together.  Kernels that reduce per-id extremes therefore walk 2-D patches (AGG_ROWS x 64
// pixels per 256-thread workgroup, a wavefront per image row so loads stay coalesced), combine
// candidates in a small LDS hash table with LDS atomics, and issue one global atomic per
// (workgroup, id, field).  A full table sends the candidate straight to global memory.
#define AGG_SLOTS 128u
#define AGG_ROWS 32u
struct AggTable {
    uint32_t key[AGG_SLOTS];        // 0 = empty
    uint32_t v[3][AGG_SLOTS];
};
__device__ __forceinline__ void agg_init(AggTable &t, uint32_t i0, uint32_t i1, uint32_t i2)
{
    for (uint32_t i = threadIdx.x; i < AGG_SLOTS; i += 256u) {
        t.key[i] = 0u; t.v[0][i] = i0; t.v[1][i] = i1; t.v[2][i] = i2;
    }
}
// slot of `key` (claimed if new) or -1 when the table is full
__device__ __forceinline__ int agg_slot(AggTable &t, uint32_t key)
{
    uint32_t h = (key * 2654435761u) >> 25;
    for (uint32_t probe = 0; probe < AGG_SLOTS; probe++) {
        const uint32_t old = atomicCAS(&t.key[h], 0u, key);
        if (old == 0u || old == key) return (int)h;
        h = (h + 1u) & (AGG_SLOTS - 1u);
    }
    return -1;
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}
