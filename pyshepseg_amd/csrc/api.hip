// api.hip -- the extern "C" boundary of libshepseg_hip.so (see include/shepseg_hip.h).
// gfx950 only.  No CPU fallback: without a usable device every call fails.
#include "common.h"
#include <utility>
#include "scan.h"
#include "sort.h"
#include "kmeans.h"
#include "clump.h"
#include "elim_single.h"
#include "elim_small.h"
#include "synth.h"
#include "stitch.h"
static int read_u32(shp_ctx *ctx, const uint32_t *d, uint32_t *h);
#include "segstats.h"
#include "subset.h"
#include "spatial.h"
#include "comm.h"

#define API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_create_err;

API int shp_version(void) { return 100; }

API int shp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ctx_create(int device, int high_priority, shp_ctx **out, bool shared = false);

API int shp_ctx_create(int device, shp_ctx **out) { return ctx_create(device, 0, out); }

// a worker context of the tiled drivers: device workspace without a stream of its own (see the
// stream pool in common.h); it falls back to an ordinary context on a second device of the process
API int shp_ctx_create_shared(int device, shp_ctx **out) { return ctx_create(device, 0, out, true); }

// a context whose streams are created with the highest stream priority (the tiled drivers use
// one for the sequential stitch chain so that its small kernels are not queued behind the
// worker streams' launches)
API int shp_ctx_create_priority(int device, shp_ctx **out) { return ctx_create(device, 1, out); }

static int ctx_create(int device, int high_priority, shp_ctx **out, bool shared)
{
    if (!out) return SHP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SHP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SHP_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return SHP_ERR_HIP;
    shp_ctx *ctx = new shp_ctx();
    ctx->device = device;
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // hi = numerically lowest
    ctx->stream_priority = high_priority ? prio_hi : 0;
    if (shared) {
        std::lock_guard<std::mutex> lk(g_streams.mu);
        if (g_streams.device < 0) {
            if (hipStreamCreateWithPriority(&g_streams.misc, hipStreamNonBlocking, 0) != hipSuccess) {
                delete ctx;
                return SHP_ERR_HIP;
            }
            g_streams.device = device;
        }
        if (g_streams.device == device) {
            ctx->shared = true;
            ctx->stream = g_streams.misc;
        }
    }
    if (!ctx->shared &&
        hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, ctx->stream_priority) != hipSuccess) {
        delete ctx;
        return SHP_ERR_HIP;
    }
    if (hipHostMalloc((void **)&ctx->h_pinned, SHP_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) {
        if (!ctx->shared) hipStreamDestroy(ctx->stream);
        delete ctx;
        return SHP_ERR_HIP;
    }
    if (hipMalloc((void **)&ctx->scan_ctr, 256) != hipSuccess || hipMemset(ctx->scan_ctr, 0, 256) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {       // (the context's stream does not wait for the null stream)
        hipHostFree(ctx->h_pinned);
        if (!ctx->shared) hipStreamDestroy(ctx->stream);
        delete ctx;
        return SHP_ERR_NOMEM;
    }
    for (int i = 0; i < 16; i++) hipEventCreate(&ctx->ev[i]);
    hipEventCreateWithFlags(&ctx->evfork, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->evjoin, hipEventDisableTiming);
    ctx->bufs = {&ctx->img, &ctx->clus, &ctx->lab, &ctx->seg, &ctx->aux, &ctx->aux2, &ctx->stack,
                 &ctx->scan_tmp, &ctx->sort_k0, &ctx->sort_k1, &ctx->sort_v1, &ctx->sort_hist,
                 &ctx->pix, &ctx->segsz, &ctx->origsz, &ctx->off, &ctx->ssum, &ctx->chnext,
                 &ctx->chtail, &ctx->mergeto, &ctx->tcount, &ctx->toff, &ctx->tfill, &ctx->tlist,
                 &ctx->tsorted, &ctx->small, &ctx->cen, &ctx->fit_x, &ctx->fit_lab, &ctx->fit_part, &ctx->fit_lb,
                 &ctx->big, &ctx->srclist, &ctx->tgtlist, &ctx->bigbits, &ctx->singles, &ctx->dbg, &ctx->snap};
    *out = ctx;
    return SHP_OK;
}

API void shp_ctx_destroy(shp_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (DevBuf *b : ctx->bufs)
        if (b->p) hipFree(b->p);
    for (int i = 0; i < 16; i++)
        if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < PROF_POOL; i++)
        for (int j = 0; j < 2; j++)
            if (ctx->prof_ev[i][j]) hipEventDestroy(ctx->prof_ev[i][j]);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    if (ctx->h_fit) hipHostFree(ctx->h_fit);
    if (ctx->scan_ctr) hipFree(ctx->scan_ctr);
    if (ctx->stream2) { hipStreamSynchronize(ctx->stream2); hipStreamDestroy(ctx->stream2); }
    if (ctx->evfork) hipEventDestroy(ctx->evfork);
    if (ctx->evjoin) hipEventDestroy(ctx->evjoin);
    if (!ctx->shared) hipStreamDestroy(ctx->stream);
    delete ctx;
}

API const char *shp_last_error(const shp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

API int shp_last_timings(const shp_ctx *ctx, double *out)
{
    if (!ctx || !out) return SHP_ERR_ARG;
    for (int i = 0; i < 8; i++) out[i] = ctx->timings[i];
    return SHP_OK;
}

static int enter(shp_ctx *ctx)
{
    if (!ctx) return SHP_ERR_ARG;
    ctx->err.clear();
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->prof_used) {                       // events of an earlier (synchronised) call
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        prof_collect(ctx);
    }
    return 0;
}

static int check_img_args(shp_ctx *ctx, const void *img, int dtype, int nb, int nr, int nc)
{
    if (!img) SHP_FAIL(ctx, SHP_ERR_ARG, "img is NULL");
    if (dtype_size(dtype) == 0) SHP_FAIL(ctx, SHP_ERR_ARG, "unsupported image dtype %d", dtype);
    if (nb < 1 || nr < 0 || nc < 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad image shape (%d,%d,%d)", nb, nr, nc);
    if ((uint64_t)nr * (uint64_t)nc >= 0x7fffffffull)
        SHP_FAIL(ctx, SHP_ERR_ARG, "tile too large: %d x %d", nr, nc);
    return 0;
}

static int upload_img(shp_ctx *ctx, const void *img, int dtype, int nb, size_t npix)
{
    const size_t bytes = (size_t)nb * npix * dtype_size(dtype);
    CHK(buf_ensure(ctx, ctx->img, bytes));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(ctx->img.p, img, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

static int read_u32(shp_ctx *ctx, const uint32_t *d, uint32_t *h)
{
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, d, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *h = ctx->h_pinned[0];
    return 0;
}

static void collect_timings(shp_ctx *ctx);
static float ev_ms(shp_ctx *ctx, int a, int b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) return 0.f;
    return ms;
}

// ---- k-means --------------------------------------------------------------------------------
API int shp_last_fit_path(const shp_ctx *ctx) { return ctx ? ctx->fit_path : -1; }

API int shp_kmeans_fit(shp_ctx *ctx, const double *xsample, int64_t nrows, int nbands, int k,
                       const double *init_centres, int max_iter, double tol_rel,
                       double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    CHK(enter(ctx));
    if (!xsample || !init_centres || !centres_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    return run_kmeans_fit(ctx, xsample, FIT_DT_F64, nrows, nbands, k, init_centres, max_iter, tol_rel,
                          centres_out, labels_out, n_iter_out);
}

// the same on rows of the image's own pixel type (no float64 copy of the sample on the host)
API int shp_kmeans_fit_typed(shp_ctx *ctx, const void *xsample, int dtype, int64_t nrows, int nbands,
                             int k, const double *init_centres, int max_iter, double tol_rel,
                             double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    CHK(enter(ctx));
    if (!xsample || !init_centres || !centres_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    if (dtype_size(dtype) == 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad pixel type %d", dtype);
    return run_kmeans_fit(ctx, xsample, dtype, nrows, nbands, k, init_centres, max_iter, tol_rel,
                          centres_out, labels_out, n_iter_out);
}

// The whole-image model of the tiled driver from the sub-sample as it leaves the device: nbands
// planes of npix pixels.  Rows with the null value in any band are dropped, the initial centres are
// diagonalClusterCentres of what is left (init_centres == NULL) or given; *nrows_out = rows fitted
// (labels_out, optional, holds that many).  Same arithmetic as shp_kmeans_fit_typed on the
// transposed rows, prepared by one host thread per band.
API int shp_kmeans_fit_planar(shp_ctx *ctx, const void *planes, int dtype, int64_t npix, int nbands,
                              int has_null, int64_t null_val, int k, const double *init_centres,
                              int max_iter, double tol_rel, double *centres_out, int32_t *labels_out,
                              int *n_iter_out, int64_t *nrows_out)
{
    CHK(enter(ctx));
    if (!planes || !centres_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    if (dtype_size(dtype) == 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad pixel type %d", dtype);
    if (nrows_out) *nrows_out = 0;
    return run_kmeans_fit(ctx, planes, dtype, npix, nbands, k, init_centres, max_iter, tol_rel, centres_out,
                          labels_out, n_iter_out, true, has_null, (long long)null_val, nrows_out);
}

// the same with the E-step sharded by sample rows over the ranks of a communicator: every rank passes the SAME
// sample and gets the same model (fit_elkan.h: FitShard); cm == NULL: shp_kmeans_fit_planar
API int shp_kmeans_fit_planar_dist(shp_ctx *ctx, shp_comm *cm, const void *planes, int dtype, int64_t npix, int nbands,
                                   int has_null, int64_t null_val, int k, const double *init_centres,
                                   int max_iter, double tol_rel, double *centres_out, int32_t *labels_out,
                                   int *n_iter_out, int64_t *nrows_out)
{
    CHK(enter(ctx));
    if (!planes || !centres_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    if (dtype_size(dtype) == 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad pixel type %d", dtype);
    if (nrows_out) *nrows_out = 0;
    FitShard sh;
    if (cm && cm->world > 1) {
        if (cm->ctx->device != ctx->device) SHP_FAIL(ctx, SHP_ERR_ARG, "the communicator lives on another device");
        sh.rank = cm->rank; sh.world = cm->world; sh.nc = (void *)cm->nc;
    }
    return run_kmeans_fit(ctx, planes, dtype, npix, nbands, k, init_centres, max_iter, tol_rel, centres_out,
                          labels_out, n_iter_out, true, has_null, (long long)null_val, nrows_out, sh);
}

API int shp_kmeans_assign(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows, int ncols,
                          const double *centres, int k, int has_null, int64_t null_val,
                          int32_t *clusters_out)
{
    CHK(enter(ctx));
    CHK(check_img_args(ctx, img, dtype, nbands, nrows, ncols));
    if (!centres || !clusters_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const size_t npix = (size_t)nrows * ncols;
    if (npix == 0) return 0;
    CHK(upload_img(ctx, img, dtype, nbands, npix));
    CHK(buf_ensure(ctx, ctx->aux, npix * 4));
    CHK(launch_assign(ctx, ctx->img.p, dtype, nbands, npix, centres, k, has_null, null_val, nullptr,
                      bp<int32_t>(ctx->aux)));
    HIPCHK(ctx, hipMemcpyAsync(clusters_out, ctx->aux.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- stages ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_i32_to_u16(const int32_t *__restrict__ in,
                                                    uint16_t *__restrict__ out, uint32_t n,
                                                    uint32_t *bad)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int32_t v = in[p];
    if (v < 0 || v > 65535) atomicAdd(bad, 1u);
    out[p] = (uint16_t)v;
}

API int shp_clump(shp_ctx *ctx, const int32_t *clusters, int nrows, int ncols, int four_connected,
                  uint32_t *seg_out, uint32_t *max_seg_id_out)
{
    CHK(enter(ctx));
    if (!clusters || !seg_out || !max_seg_id_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    if (nrows < 0 || ncols < 0 || (uint64_t)nrows * (uint64_t)ncols >= 0x7fffffffull)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad shape %d x %d", nrows, ncols);
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    *max_seg_id_out = 0;
    if (n == 0) return 0;
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->clus, (size_t)n * 2));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->small, 4096));
    uint32_t *bad = bp<uint32_t>(ctx->small);
    HIPCHK(ctx, hipMemcpyAsync(ctx->aux2.p, clusters, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(bad, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_i32_to_u16, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                       bp<int32_t>(ctx->aux2), bp<uint16_t>(ctx->clus), n, bad);
    KCHK(ctx);
    uint32_t nbad = 0;
    CHK(read_u32(ctx, bad, &nbad));
    if (nbad) SHP_FAIL(ctx, SHP_ERR_ARG, "cluster ids must lie in 0..65535 (%u do not)", nbad);
    CHK(run_clump(ctx, bp<uint16_t>(ctx->clus), nrows, ncols, four_connected, bp<uint32_t>(ctx->seg),
                  bad + 1));
    HIPCHK(ctx, hipMemcpyAsync(seg_out, ctx->seg.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    CHK(read_u32(ctx, bad + 1, max_seg_id_out));
    return 0;
}

API int shp_make_seg_size(shp_ctx *ctx, const uint32_t *seg, int64_t npix, uint32_t max_seg_id,
                          uint32_t *seg_size_out)
{
    CHK(enter(ctx));
    if (!seg || !seg_size_out || npix < 0 || npix >= 0x7fffffffll) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const uint32_t n = (uint32_t)npix;
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->segsz, ((size_t)max_seg_id + 2) * 4));
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    // ids above max_seg_id would index out of bounds: verify on the host copy first
    for (uint32_t i = 0; i < n; i++)
        if (seg[i] > max_seg_id) SHP_FAIL(ctx, SHP_ERR_ARG, "segment id %u > max_seg_id %u", seg[i], max_seg_id);
    CHK(run_seg_size(ctx, bp<uint32_t>(ctx->seg), n, max_seg_id, bp<uint32_t>(ctx->segsz)));
    HIPCHK(ctx, hipMemcpyAsync(seg_size_out, ctx->segsz.p, ((size_t)max_seg_id + 1) * 4,
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int check_seg_ids(shp_ctx *ctx, const uint32_t *seg, uint32_t n, uint32_t max_id)
{
    for (uint32_t i = 0; i < n; i++)
        if (seg[i] > max_id) SHP_FAIL(ctx, SHP_ERR_ARG, "segment id %u > max_seg_id %u", seg[i], max_id);
    return 0;
}

API int shp_eliminate_single(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows,
                             int ncols, int four_connected, uint32_t *seg_inout,
                             uint32_t *max_seg_id_inout)
{
    CHK(enter(ctx));
    CHK(check_img_args(ctx, img, dtype, nbands, nrows, ncols));
    if (!seg_inout || !max_seg_id_inout) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    if (n == 0) return 0;
    CHK(check_seg_ids(ctx, seg_inout, n, *max_seg_id_inout));
    CHK(upload_img(ctx, img, dtype, nbands, n));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg_inout, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    CHK(run_eliminate_single(ctx, ctx->img.p, dtype, nbands, nrows, ncols, four_connected,
                             bp<uint32_t>(ctx->seg), max_seg_id_inout));
    HIPCHK(ctx, hipMemcpyAsync(seg_inout, ctx->seg.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_eliminate_small(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows,
                            int ncols, int four_connected, int min_seg_size,
                            double max_spectral_diff, uint32_t *seg_inout,
                            uint32_t *max_seg_id_inout, int64_t *num_elim_out)
{
    CHK(enter(ctx));
    CHK(check_img_args(ctx, img, dtype, nbands, nrows, ncols));
    if (!seg_inout || !max_seg_id_inout) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    if (num_elim_out) *num_elim_out = 0;
    if (n == 0) return 0;
    CHK(check_seg_ids(ctx, seg_inout, n, *max_seg_id_inout));
    CHK(upload_img(ctx, img, dtype, nbands, n));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg_inout, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    int64_t ne = 0;
    CHK(run_eliminate_small(ctx, ctx->img.p, dtype, nbands, nrows, ncols, four_connected,
                            min_seg_size, max_spectral_diff, bp<uint32_t>(ctx->seg),
                            max_seg_id_inout, &ne));
    if (num_elim_out) *num_elim_out = ne;
    HIPCHK(ctx, hipMemcpyAsync(seg_inout, ctx->seg.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_segment_locations(shp_ctx *ctx, const uint32_t *seg, int nrows, int ncols, uint32_t max_seg_id,
                              uint32_t *offsets_out, uint32_t *pix_out)
{
    CHK(enter(ctx));
    if (!seg || !offsets_out || !pix_out || nrows < 0 || ncols < 0 ||
        (uint64_t)nrows * (uint64_t)ncols >= 0x7fffffffull)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    memset(offsets_out, 0, ((size_t)max_seg_id + 2) * 4);
    if (n == 0) return 0;
    CHK(check_seg_ids(ctx, seg, n, max_seg_id));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *pix = nullptr;
    CHK(run_segment_tables(ctx, bp<uint32_t>(ctx->seg), n, (uint32_t)ncols, max_seg_id, nullptr, 0, 0, &pix));
    HIPCHK(ctx, hipMemcpyAsync(offsets_out, ctx->off.p, ((size_t)max_seg_id + 2) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pix_out, pix, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_build_segment_spectra(shp_ctx *ctx, const uint32_t *seg, const void *img, int dtype, int nbands,
                                  int nrows, int ncols, uint32_t max_seg_id, float *spect_sum_out)
{
    CHK(enter(ctx));
    CHK(check_img_args(ctx, img, dtype, nbands, nrows, ncols));
    if (!seg || !spect_sum_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    const size_t nout = ((size_t)max_seg_id + 1) * nbands;
    memset(spect_sum_out, 0, nout * 4);
    if (n == 0) return 0;
    CHK(check_seg_ids(ctx, seg, n, max_seg_id));
    CHK(upload_img(ctx, img, dtype, nbands, n));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *pix = nullptr;
    CHK(run_segment_tables(ctx, bp<uint32_t>(ctx->seg), n, (uint32_t)ncols, max_seg_id, ctx->img.p, dtype, nbands, &pix));
    HIPCHK(ctx, hipMemcpyAsync(spect_sum_out, ctx->ssum.p, nout * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// Device-resident fused pipeline on ctx->img -> ctx->seg.  Records stage events 1..5.
static int segment_device(shp_ctx *ctx, const void *d_img, uint32_t *d_seg, int dtype, int nb,
                          uint32_t nrows, uint32_t ncols,
                          const double *centres, int k, int has_null, int64_t null_val, int four,
                          int min_seg_size, double msd, uint32_t *max_seg_id, int64_t *singles,
                          int64_t *small, uint32_t *nclumps_out, const uint16_t *clus_window = nullptr,
                          uint32_t clus_pitch = 0, const ImgGeom *geom = nullptr)
{
    const uint32_t n = nrows * ncols;
    CHK(buf_ensure(ctx, ctx->small, 4096));
    uint32_t *scal = bp<uint32_t>(ctx->small);
    hipEventRecord(ctx->ev[1], ctx->stream);
    const uint16_t *d_clus = clus_window;       // the tile's window of a raster-wide cluster map, read in place
    if (!d_clus) {
        CHK(buf_ensure(ctx, ctx->clus, (size_t)n * 2));
        CHK(launch_assign(ctx, d_img, dtype, nb, n, centres, k, has_null, null_val,
                          bp<uint16_t>(ctx->clus), nullptr));
        d_clus = bp<uint16_t>(ctx->clus);
        clus_pitch = ncols;
    }
    hipEventRecord(ctx->ev[2], ctx->stream);
    CHK(buf_ensure(ctx, ctx->segsz, ((size_t)n + 2) * 4));
    CHK(buf_ensure(ctx, ctx->singles, ((size_t)n + 2) * 4));
    CHK(run_clump(ctx, d_clus, nrows, ncols, four, d_seg, scal + 2, bp<uint32_t>(ctx->segsz),
                  bp<uint32_t>(ctx->singles), scal + 3, clus_pitch));
    // read back: number of clumps, number of one-pixel clumps, null-pixel count (scal[2..4])
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, scal + 2, 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t nclumps = ctx->h_pinned[0], nsingles = ctx->h_pinned[1], nnull = ctx->h_pinned[2];
    hipEventRecord(ctx->ev[3], ctx->stream);
    uint32_t max_id = nclumps;
    // (a lone null pixel is itself a size-1 "segment" 0: then fall back to the full scan, N4)
    CHK(run_eliminate_single(ctx, d_img, dtype, nb, nrows, ncols, four, d_seg, &max_id, 1, nnull != 1u,
                             nsingles, geom));
    hipEventRecord(ctx->ev[4], ctx->stream);
    if (singles) *singles = (int64_t)nclumps - (int64_t)max_id;        // shepseg.py:226-227
    int64_t ne = 0;
    CHK(run_eliminate_small(ctx, d_img, dtype, nb, nrows, ncols, four, min_seg_size, msd, d_seg,
                            &max_id, &ne, 1, geom));
    hipEventRecord(ctx->ev[5], ctx->stream);
    if (small) *small = ne;
    if (max_seg_id) *max_seg_id = max_id;
    if (nclumps_out) *nclumps_out = nclumps;
    return 0;
}

API int shp_segment_tile(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows, int ncols,
                         const double *centres, int k, int has_null, int64_t null_val,
                         int four_connected, int min_seg_size, double max_spectral_diff,
                         uint32_t *seg_out, uint32_t *max_seg_id_out, int64_t *singles_elim_out,
                         int64_t *small_elim_out, uint32_t *num_clumps_out)
{
    CHK(enter(ctx));
    CHK(check_img_args(ctx, img, dtype, nbands, nrows, ncols));
    if (!centres || !seg_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    if (max_seg_id_out) *max_seg_id_out = 0;
    if (singles_elim_out) *singles_elim_out = 0;
    if (small_elim_out) *small_elim_out = 0;
    if (num_clumps_out) *num_clumps_out = 0;
    if (n == 0) return 0;
    hipEventRecord(ctx->ev[0], ctx->stream);
    CHK(upload_img(ctx, img, dtype, nbands, n));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    CHK(segment_device(ctx, ctx->img.p, bp<uint32_t>(ctx->seg), dtype, nbands, nrows, ncols, centres, k, has_null, null_val,
                       four_connected, min_seg_size, max_spectral_diff, max_seg_id_out,
                       singles_elim_out, small_elim_out, num_clumps_out));
    HIPCHK(ctx, hipMemcpyAsync(seg_out, ctx->seg.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    hipEventRecord(ctx->ev[6], ctx->stream);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    collect_timings(ctx);
    return 0;
}

// ---- synthetic imagery ----------------------------------------------------------------------
API int shp_synthimg(shp_ctx *ctx, uint64_t seed, int nbands, int64_t y0, int64_t x0, int nrows,
                     int ncols, uint16_t *out_host)
{
    CHK(enter(ctx));
    if (!out_host || nbands < 1 || nrows < 0 || ncols < 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const size_t total = (size_t)nbands * nrows * ncols;
    if (total == 0) return 0;
    CHK(buf_ensure(ctx, ctx->img, total * 2));
    hipLaunchKernelGGL(k_synthimg, dim3(grid_for(total, 256, 65535u * 8u)), dim3(256), 0, ctx->stream,
                       seed, nbands, y0, x0, (uint32_t)nrows, (uint32_t)ncols, bp<uint16_t>(ctx->img));
    KCHK(ctx);
    HIPCHK(ctx, hipMemcpyAsync(out_host, ctx->img.p, total * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static void collect_timings(shp_ctx *ctx)
{
    prof_collect(ctx);
    ctx->timings[0] = ev_ms(ctx, 1, 2);
    ctx->timings[1] = ev_ms(ctx, 2, 3);
    ctx->timings[2] = ev_ms(ctx, 3, 4);
    ctx->timings[3] = ev_ms(ctx, 4, 5);
    ctx->timings[4] = ev_ms(ctx, 0, 1);
    ctx->timings[5] = ev_ms(ctx, 5, 6);
    ctx->timings[6] = ev_ms(ctx, 0, 6);
}

// ---- device-resident rasters (tiled driver, benchmark) ----------------------------------------
API int shp_dev_alloc(shp_ctx *ctx, size_t bytes, void **dptr)
{
    CHK(enter(ctx));
    if (!dptr) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) SHP_FAIL(ctx, SHP_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return 0;
}

API int shp_dev_free(shp_ctx *ctx, void *dptr)
{
    CHK(enter(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (dptr) HIPCHK(ctx, hipFree(dptr));
    return 0;
}

API int shp_dev_upload(shp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    CHK(enter(ctx));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_dev_download(shp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    CHK(enter(ctx));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// pinned (page-locked) host memory for the raster I/O pipeline: transfers from / to it run at full
// PCIe rate and asynchronously, pageable memory is staged by the runtime in small synchronous pieces
API int shp_host_alloc(shp_ctx *ctx, size_t bytes, void **hptr)
{
    CHK(enter(ctx));
    if (!hptr) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    *hptr = nullptr;
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) SHP_FAIL(ctx, SHP_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return 0;
}

API int shp_host_free(shp_ctx *ctx, void *hptr)
{
    CHK(enter(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (hptr) HIPCHK(ctx, hipHostFree(hptr));
    return 0;
}

API int shp_dev_copy(shp_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    CHK(enter(ctx));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_dev_memset(shp_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    CHK(enter(ctx));
    if (bytes) HIPCHK(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_dev_synthimg(shp_ctx *ctx, uint64_t seed, int nbands, int64_t y0, int64_t x0, int nrows,
                         int ncols, void *d_out)
{
    CHK(enter(ctx));
    if (!d_out || nbands < 1 || nrows < 0 || ncols < 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const size_t total = (size_t)nbands * nrows * ncols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_synthimg, dim3(grid_for(total, 256, 65535u * 8u)), dim3(256), 0, ctx->stream,
                       seed, nbands, y0, x0, (uint32_t)nrows, (uint32_t)ncols, (uint16_t *)d_out);
    KCHK(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// synthetic label raster for the statistics benchmark (BASELINE config 5): bh x bw-pixel blocks
// numbered row-major from 1; returns the largest id
__global__ __launch_bounds__(256) void k_block_labels(uint32_t nrows, uint32_t ncols, uint32_t bh, uint32_t bw,
                                                      uint32_t ncb, uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)nrows * ncols) return;
    const uint32_t r = (uint32_t)(i / ncols), c = (uint32_t)(i - (size_t)r * ncols);
    out[i] = (r / bh) * ncb + c / bw + 1u;
}

API int shp_dev_block_labels(shp_ctx *ctx, int nrows, int ncols, int block_rows, int block_cols,
                             uint32_t *d_out, uint32_t *max_id_out)
{
    CHK(enter(ctx));
    if (!d_out || nrows < 0 || ncols < 0 || block_rows < 1 || block_cols < 1) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const uint64_t ncb = ((uint64_t)ncols + block_cols - 1) / block_cols, nrb = ((uint64_t)nrows + block_rows - 1) / block_rows;
    if (ncb * nrb >= 0xffffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "too many blocks");
    if (max_id_out) *max_id_out = (uint32_t)(ncb * nrb);
    const size_t total = (size_t)nrows * ncols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_block_labels, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, (uint32_t)nrows,
                       (uint32_t)ncols, (uint32_t)block_rows, (uint32_t)block_cols, (uint32_t)ncb, d_out);
    KCHK(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// gather rows y0,y0+ystep.. / cols likewise of every band of a device raster into a host array
__global__ __launch_bounds__(256) void k_subsample(const void *__restrict__ img, int dtype, int nb,
                                                   uint32_t rows, uint32_t cols,
                                                   const uint32_t *__restrict__ ry,
                                                   const uint32_t *__restrict__ rx, uint32_t ny,
                                                   uint32_t nx, void *__restrict__ out)
{
    const size_t total = (size_t)nb * ny * nx;
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / ((size_t)ny * nx), r = (i / nx) % ny, c = i % nx;
    const size_t src = b * (size_t)rows * cols + (size_t)ry[r] * cols + rx[c];
    switch (dtype) {
    case SHP_U8: ((uint8_t *)out)[i] = ((const uint8_t *)img)[src]; break;
    case SHP_I16: case SHP_U16: ((uint16_t *)out)[i] = ((const uint16_t *)img)[src]; break;
    default: ((uint32_t *)out)[i] = ((const uint32_t *)img)[src]; break;
    }
}

API int shp_dev_subsample(shp_ctx *ctx, const void *d_img, int dtype, int nbands, int nrows, int ncols,
                          const uint32_t *row_idx, int ny, const uint32_t *col_idx, int nx,
                          void *out_host)
{
    CHK(enter(ctx));
    if (!d_img || !row_idx || !col_idx || !out_host || dtype_size(dtype) == 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const size_t total = (size_t)nbands * ny * nx;
    if (total == 0) return 0;
    CHK(buf_ensure(ctx, ctx->aux, total * dtype_size(dtype)));
    CHK(buf_ensure(ctx, ctx->aux2, ((size_t)ny + nx) * 4));
    uint32_t *ry = bp<uint32_t>(ctx->aux2), *rx = ry + ny;
    HIPCHK(ctx, hipMemcpyAsync(ry, row_idx, (size_t)ny * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(rx, col_idx, (size_t)nx * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_subsample, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, d_img, dtype,
                       nbands, (uint32_t)nrows, (uint32_t)ncols, ry, rx, (uint32_t)ny, (uint32_t)nx,
                       ctx->aux.p);
    KCHK(ctx);
    // through the context's pinned staging buffer: a pageable destination makes the runtime stage
    // the copy itself in small synchronous pieces (1-30 ms for 12 MB, erratic)
    const size_t bytes = total * dtype_size(dtype);
    if (ctx->h_fit_cap < bytes) {
        if (ctx->h_fit) hipHostFree(ctx->h_fit);
        ctx->h_fit = nullptr; ctx->h_fit_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_fit, bytes, hipHostMallocDefault) != hipSuccess)
            SHP_FAIL(ctx, SHP_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
        ctx->h_fit_cap = bytes;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_fit, ctx->aux.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out_host, ctx->h_fit, bytes);
    return 0;
}

// copy window (x, y, xs, ys) of every band of a device raster into a contiguous tile image:
// blockIdx.x = (band, window row), a thread moves 16 bytes of the row (one vector load / store
// when both ends are 16-byte aligned, element by element otherwise)
__global__ __launch_bounds__(256) void k_window(const uint8_t *__restrict__ img, uint32_t esize,
                                                uint32_t rows, uint32_t cols, uint32_t x, uint32_t y,
                                                uint32_t xs, uint32_t ys, uint8_t *__restrict__ out)
{
    const uint32_t br = blockIdx.x, b = br / ys, r = br - b * ys;
    const size_t rowbytes = (size_t)xs * esize;
    const size_t c0 = ((size_t)blockIdx.y * 256u + threadIdx.x) * 16u;         // byte offset in the row
    if (c0 >= rowbytes) return;
    const uint8_t *src = img + (((size_t)b * rows + (y + r)) * cols + x) * esize + c0;
    uint8_t *dst = out + (size_t)br * rowbytes + c0;
    if (c0 + 16u <= rowbytes && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0u) {
        *(uint4 *)dst = *(const uint4 *)src;
    } else {
        const uint32_t nbytes = rowbytes - c0 < 16u ? (uint32_t)(rowbytes - c0) : 16u;
        for (uint32_t i = 0; i < nbytes; i++) dst[i] = src[i];
    }
}

API int shp_segment_window_dev(shp_ctx *ctx, const void *d_img, int dtype, int nbands, int img_rows,
                               int img_cols, int x, int y, int xs, int ys, const double *centres, int k,
                               int has_null, int64_t null_val, int four_connected, int min_seg_size,
                               double max_spectral_diff, uint32_t *d_seg_out, uint32_t *max_seg_id_out,
                               int64_t *singles_elim_out, int64_t *small_elim_out,
                               uint32_t *num_clumps_out, const uint16_t *d_clusmap)
{
    CHK(enter(ctx));
    if (!d_img || !centres || !d_seg_out || dtype_size(dtype) == 0 || nbands < 1)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (x < 0 || y < 0 || xs < 0 || ys < 0 || (int64_t)x + xs > img_cols || (int64_t)y + ys > img_rows)
        SHP_FAIL(ctx, SHP_ERR_ARG, "window (%d,%d,%d,%d) outside the %d x %d raster", x, y, xs, ys, img_rows, img_cols);
    if ((uint64_t)xs * (uint64_t)ys >= 0x7fffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "tile too large");
    const uint32_t n = (uint32_t)xs * (uint32_t)ys;
    if (max_seg_id_out) *max_seg_id_out = 0;
    if (singles_elim_out) *singles_elim_out = 0;
    if (small_elim_out) *small_elim_out = 0;
    if (num_clumps_out) *num_clumps_out = 0;
    if (n == 0) return 0;
    FillScope fs(ctx, 1);
    fill_acquire(ctx, 0);
    hipEventRecord(ctx->ev[0], ctx->stream);
    // With the cluster map nothing reads the tile as a contiguous image any more: the single-pixel
    // and spectra kernels read the window of the resident raster in place (ImgGeom).  Without it
    // the assign step wants the compact copy.
    const void *tile_img = d_img;
    ImgGeom geom = geom_compact(n, (uint32_t)xs);
    if (d_clusmap) {
        geom.bstride = (size_t)img_rows * (size_t)img_cols;
        geom.origin = (size_t)y * (size_t)img_cols + (size_t)x;
        geom.pitch = (uint32_t)img_cols;
        // (the connected-component kernels read the window of the raster-wide cluster map in place too)
    } else if (!(x == 0 && y == 0 && xs == img_cols && ys == img_rows)) {
        const size_t total = (size_t)nbands * n;
        CHK(buf_ensure(ctx, ctx->img, total * dtype_size(dtype)));
        const size_t rowbytes = (size_t)xs * dtype_size(dtype);
        hipLaunchKernelGGL(k_window, dim3((unsigned)nbands * (unsigned)ys, grid_for((rowbytes + 15) / 16, 256)),
                           dim3(256), 0, ctx->stream, (const uint8_t *)d_img, (uint32_t)dtype_size(dtype),
                           (uint32_t)img_rows, (uint32_t)img_cols, (uint32_t)x, (uint32_t)y, (uint32_t)xs,
                           (uint32_t)ys, (uint8_t *)ctx->img.p);
        KCHK(ctx);
        tile_img = ctx->img.p;
    }
    const uint16_t *clus_window = d_clusmap ? (const uint16_t *)d_clusmap + (size_t)y * (size_t)img_cols + (size_t)x : nullptr;
    CHK(segment_device(ctx, tile_img, d_seg_out, dtype, nbands, ys, xs, centres, k, has_null, null_val,
                       four_connected, min_seg_size, max_spectral_diff, max_seg_id_out,
                       singles_elim_out, small_elim_out, num_clumps_out, clus_window, (uint32_t)img_cols, &geom));
    hipEventRecord(ctx->ev[6], ctx->stream);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    collect_timings(ctx);
    return 0;
}

// Clusters of rectangles of a device raster into a raster-wide cluster map (uint16, the raster's
// geometry): 0 = null pixel, else cluster index + 1, exactly what the per-tile assign step writes.
// rects: nrects x (x, y, xs, ys).  Synchronous.
API int shp_assign_rects_dev(shp_ctx *ctx, const void *d_img, int dtype, int nbands, int img_rows,
                             int img_cols, const int32_t *rects, int nrects, const double *centres,
                             int k, int has_null, int64_t null_val, uint16_t *d_clusmap)
{
    CHK(enter(ctx));
    if (!d_img || !centres || !d_clusmap || dtype_size(dtype) == 0 || nbands < 1 || nrects < 0 ||
        (nrects > 0 && !rects) || img_rows < 0 || img_cols < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    for (int r = 0; r < nrects; r++) {
        const int32_t *q = rects + 4 * r;
        if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[3] < 0 || (int64_t)q[0] + q[2] > img_cols ||
            (int64_t)q[1] + q[3] > img_rows)
            SHP_FAIL(ctx, SHP_ERR_ARG, "rectangle (%d,%d,%d,%d) outside the %d x %d raster", q[0], q[1], q[2],
                     q[3], img_rows, img_cols);
    }
    if (nrects == 0) return 0;
    FillScope fs(ctx, 1);
    fill_acquire(ctx, 0);
    CHK(launch_assign(ctx, d_img, dtype, nbands, (size_t)img_rows * (size_t)img_cols, centres, k, has_null,
                      null_val, d_clusmap, nullptr, rects, nrects, (uint32_t)img_cols));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// host tile in, device labels out (file-based tiled driver: read -> H2D -> segment, labels stay in HBM)
API int shp_segment_tile_to_dev(shp_ctx *ctx, const void *img, int dtype, int nbands, int nrows,
                                int ncols, const double *centres, int k, int has_null, int64_t null_val,
                                int four_connected, int min_seg_size, double max_spectral_diff,
                                uint32_t *d_seg_out, uint32_t *max_seg_id_out,
                                int64_t *singles_elim_out, int64_t *small_elim_out,
                                uint32_t *num_clumps_out)
{
    CHK(enter(ctx));
    CHK(check_img_args(ctx, img, dtype, nbands, nrows, ncols));
    if (!centres || !d_seg_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const uint32_t n = (uint32_t)nrows * (uint32_t)ncols;
    if (max_seg_id_out) *max_seg_id_out = 0;
    if (singles_elim_out) *singles_elim_out = 0;
    if (small_elim_out) *small_elim_out = 0;
    if (num_clumps_out) *num_clumps_out = 0;
    if (n == 0) return 0;
    FillScope fs(ctx, 1);
    hipEventRecord(ctx->ev[0], ctx->stream);
    CHK(upload_img(ctx, img, dtype, nbands, n));          // (PCIe: outside the gate)
    if (ctx->shared) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // (the phases run on borrowed streams)
    fill_acquire(ctx, 0);
    CHK(segment_device(ctx, ctx->img.p, d_seg_out, dtype, nbands, nrows, ncols, centres, k, has_null,
                       null_val, four_connected, min_seg_size, max_spectral_diff, max_seg_id_out,
                       singles_elim_out, small_elim_out, num_clumps_out));
    hipEventRecord(ctx->ev[6], ctx->stream);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    collect_timings(ctx);
    return 0;
}

API int shp_stitch_tile_dev(shp_ctx *ctx, uint32_t *d_tile, int ys, int xs, int overlap,
                            const uint32_t *d_top_b, int64_t top_pitch, const uint32_t *d_left_b,
                            int64_t left_pitch, uint32_t max_local, int simple_recode,
                            uint32_t *d_max_seg_id, int top, int bottom, int left, int right,
                            uint32_t *d_out, int64_t out_pitch, int xout, int yout)
{
    CHK(enter(ctx));
    if (!d_tile || !d_max_seg_id || !d_out || ys < 0 || xs < 0 || overlap < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (top < 0 || left < 0 || bottom > ys || right > xs || top > bottom || left > right)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad trimmed window");
    return run_stitch_tile(ctx, d_tile, (uint32_t)ys, (uint32_t)xs, (uint32_t)overlap, d_top_b,
                           (size_t)top_pitch, d_left_b, (size_t)left_pitch, max_local, simple_recode,
                           d_max_seg_id, (uint32_t)top, (uint32_t)bottom, (uint32_t)left,
                           (uint32_t)right, d_out, (size_t)out_pitch, (uint32_t)xout, (uint32_t)yout);
}

API int shp_sync(shp_ctx *ctx)
{
    CHK(enter(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) HIPCHK(ctx, hipStreamSynchronize(ctx->stream2));
    return 0;
}

API int shp_stitch_prepare_dev(shp_ctx *ctx, const uint32_t *d_tile, int ys, int xs, int overlap,
                               int has_top, int has_left, uint32_t max_local, int top, int bottom,
                               int left, int right, uint32_t *d_meta, uint32_t *cross_px_out)
{
    CHK(enter(ctx));
    if (!d_tile || !d_meta || ys < 0 || xs < 0 || overlap < 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (top < 0 || left < 0 || bottom > ys || right > xs || top > bottom || left > right)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad trimmed window");
    FillScope fs(ctx, 1);
    fill_acquire(ctx, 3);
    uint32_t *d_cross = nullptr;
    if (cross_px_out) {
        CHK(buf_ensure(ctx, ctx->small, 4096));
        d_cross = bp<uint32_t>(ctx->small) + 64;
        HIPCHK(ctx, hipMemsetAsync(d_cross, 0, 8, ctx->stream));
    }
    CHK(run_stitch_prepare(ctx, d_tile, (uint32_t)ys, (uint32_t)xs, (uint32_t)overlap, has_top, has_left,
                           max_local, (uint32_t)top, (uint32_t)bottom, (uint32_t)left, (uint32_t)right,
                           d_meta, d_cross));
    if (cross_px_out) HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, d_cross, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (cross_px_out) { cross_px_out[0] = ctx->h_pinned[0]; cross_px_out[1] = ctx->h_pinned[1]; }
    return 0;
}

API int shp_stitch_chain_dev(shp_ctx *ctx, const uint32_t *d_tile, int ys, int xs, int overlap,
                             const uint32_t *d_top_b, int64_t top_pitch, const uint32_t *d_left_b,
                             int64_t left_pitch, uint32_t max_local, int simple_recode,
                             uint32_t *d_max_seg_id, int top, int bottom, int left, int right,
                             uint32_t *d_meta, uint32_t *d_right_out, uint32_t *d_bottom_out,
                             uint32_t *d_out, int64_t out_pitch, int xout, int yout,
                             uint32_t top_cross_px, uint32_t left_cross_px)
{
    CHK(enter(ctx));
    if (!d_tile || !d_max_seg_id || !d_meta || ys < 0 || xs < 0 || overlap < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (top < 0 || left < 0 || bottom > ys || right > xs || top > bottom || left > right)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad trimmed window");
    CHK(run_stitch_chain(ctx, d_tile, (uint32_t)ys, (uint32_t)xs, (uint32_t)overlap, d_top_b,
                         (size_t)top_pitch, d_left_b, (size_t)left_pitch, max_local, simple_recode,
                         d_max_seg_id, (uint32_t)top, (uint32_t)bottom, (uint32_t)left, (uint32_t)right,
                         d_meta, d_right_out, d_bottom_out, top_cross_px, left_cross_px));
    if (d_out)
        CHK(run_stitch_finish(ctx, d_tile, (uint32_t)ys, (uint32_t)xs, max_local, (uint32_t)top,
                              (uint32_t)bottom, (uint32_t)left, (uint32_t)right, d_meta, d_out,
                              (size_t)out_pitch, (uint32_t)xout, (uint32_t)yout));
    return 0;
}

// one trimmed tile's contribution to one overview layer; runs behind the tile's output write on the
// context's side stream (shp_stitch_chain_dev), so the chain itself is not held up
// Parallel stitch.  After shp_stitch_chain_dev ran with
// *d_max_seg_id = base (the tile's provisional base): d_out2[0] = number of new ids the tile handed
// out, d_out2[1] = the largest of them (minus base) present in its trimmed window.  Asynchronous.
API int shp_stitch_counts_dev(shp_ctx *ctx, const uint32_t *d_meta, uint32_t max_local, uint32_t base,
                              uint32_t *d_out2)
{
    CHK(enter(ctx));
    if (!d_meta || !d_out2) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    const uint32_t nseg = max_local + 1u;
    HIPCHK(ctx, hipMemsetAsync(d_out2, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_lut_counts, dim3(grid_for(nseg, 256)), dim3(256), 0, ctx->stream,
                       d_meta + 3 * (size_t)nseg, d_meta, nseg, base, d_out2);
    KCHK(ctx);
    return 0;
}

// Provisional ids -> final ids over a device raster: id -> new_base[id / stride] + id % stride
// (0 stays 0; new_base: ntiles host values).  Synchronous.
API int shp_renumber_dev(shp_ctx *ctx, uint32_t *d_raster, int64_t npix, uint32_t stride,
                         const uint32_t *new_base, int ntiles)
{
    CHK(enter(ctx));
    if ((!d_raster && npix) || !new_base || ntiles < 1 || stride == 0 || npix < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (npix == 0) return 0;
    CHK(buf_ensure(ctx, ctx->tlist, (size_t)ntiles * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->tlist.p, new_base, (size_t)ntiles * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_renumber, dim3(grid_for((size_t)npix, 256)), dim3(256), 0, ctx->stream, d_raster,
                       (size_t)npix, stride, bp<uint32_t>(ctx->tlist), (uint32_t)ntiles);
    KCHK(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

API int shp_overview_window_dev(shp_ctx *ctx, const uint32_t *d_raster, int64_t pitch, int xout, int yout,
                                int w, int h, int level, uint32_t *d_ov, int ov_w, int ov_h)
{
    CHK(enter(ctx));
    if (!d_raster || !d_ov || pitch < 0 || xout < 0 || yout < 0 || w < 0 || h < 0 || level < 1 || ov_w < 0 || ov_h < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    const uint32_t o = (uint32_t)level / 2u;
    const uint64_t nsr = (uint32_t)h > o ? ((uint32_t)h - o + level - 1u) / level : 0u;
    const uint64_t nsc = (uint32_t)w > o ? ((uint32_t)w - o + level - 1u) / level : 0u;
    if (nsr * nsc == 0) return 0;
    if (nsr * nsc >= 0x7fffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "window too large");
    CHK(ensure_stream2(ctx));
    hipLaunchKernelGGL(k_overview_window, dim3(grid_for((size_t)(nsr * nsc), 256)), dim3(256), 0, ctx->stream2, d_raster,
                       (size_t)pitch, (uint32_t)xout, (uint32_t)yout, (uint32_t)w, (uint32_t)h, (uint32_t)level, d_ov,
                       (uint32_t)ov_w, (uint32_t)ov_h);
    KCHK(ctx);
    return 0;
}

// histogram of a device label raster: hist_out[0..max_seg_id], hist_out[0] = 0 (tiling.py:1915-1963).
// ncols > 0 (npix a multiple of it): the raster's row length, which lets the pixels of a segment
// be combined per 2-D patch before they reach the global counters; 0 = unknown (1-D runs).
// numpy's pairwise float64 sum (DOUBLE_pairwise_sum) of term(i), i in [lo, lo + n), for the host (recursion as numpy's)
template <class Term>
static double hist_pairwise(Term term, size_t lo, size_t n)
{
    if (n <= 128) return np_pairwise_sum_lv<0>(term, lo, n);
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return hist_pairwise(term, lo, n2) + hist_pairwise(term, lo + n2, n - n2);
}

API int shp_hist_stats(const uint32_t *hist, int64_t n, double *out)
{
    if (!hist || !out || n < 1) return SHP_ERR_ARG;
    // mask = hist > 0; nVals = hist.sum(); min / max = first / last non-zero bin; mode = argmax (first maximum)
    uint64_t nvals = 0, wsum = 0;
    int64_t first = -1, last = -1, mode = 0;
    uint32_t best = hist[0];
    for (int64_t i = 0; i < n; i++) {
        const uint32_t h = hist[i];
        nvals += h;
        wsum += (uint64_t)i * h;                        // (values * hist).sum(): int64, exact
        if (h) { if (first < 0) first = i; last = i; }
        if (h > best) { best = h; mode = i; }
    }
    if (first < 0) { first = 0; last = n - 1; }       // all zero: argmax of an all-False mask is 0 on both sides
    const double nv = (double)nvals;
    const double mean = (double)(int64_t)wsum / nv;
    // (hist * power(values - mean, 2)).sum(): float64 terms, pairwise
    auto term = [hist, mean](size_t i) {
        const double d = (double)(int64_t)i - mean;
        const double d2 = d * d;
        return (double)hist[i] * d2;
    };
    // a .sum() over a long contiguous array reaches the pairwise routine 8192 elements (the ufunc buffer
    // size) at a time, the blocks' sums added one after the other (checked against numpy 2.2 up to 2.5 M bins)
    double ssq = 0.0;
    for (size_t lo = 0; lo < (size_t)n; lo += NP_REDUCE_BLOCK) {
        const size_t m = (size_t)n - lo < NP_REDUCE_BLOCK ? (size_t)n - lo : NP_REDUCE_BLOCK;
        const double part = hist_pairwise(term, lo, m);
        ssq = lo ? ssq + part : part;
    }
    const double sd = __builtin_sqrt(ssq / nv);
    // first bin whose cumulative count reaches hist.sum() / 2 (a float64 comparison)
    const double middle = nv / 2.0;
    uint64_t cum = 0;
    int64_t median = 0;
    bool found = false;
    for (int64_t i = 0; i < n; i++) {
        cum += hist[i];
        if ((double)cum >= middle) { median = i; found = true; break; }
    }
    if (!found) return SHP_ERR_STATE;
    out[0] = (double)first; out[1] = (double)last; out[2] = mean; out[3] = sd; out[4] = (double)mode; out[5] = (double)median;
    return 0;
}

API int shp_histogram_dev(shp_ctx *ctx, const uint32_t *d_raster, int64_t npix, int64_t ncols,
                          uint32_t max_seg_id, uint32_t *hist_out_host)
{
    CHK(enter(ctx));
    if (!d_raster || !hist_out_host || npix < 0 || ncols < 0) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (ncols > 0 && (npix % ncols != 0 || ncols > 0x7fffffffll || npix / ncols > 0x7fffffffll))
        SHP_FAIL(ctx, SHP_ERR_ARG, "npix is not a whole number of rows of %lld pixels", (long long)ncols);
    static const bool io_timing = getenv("SHEPSEG_IO_TIMING") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (io_timing) fprintf(stderr, "    [hist] %-18s %.2f ms\n", what,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count());
    };
    CHK(buf_ensure(ctx, ctx->segsz, ((size_t)max_seg_id + 2) * 4));
    uint32_t *h = bp<uint32_t>(ctx->segsz);
    const size_t hbytes = ((size_t)max_seg_id + 1) * 4;
    lap("workspace");
    HIPCHK(ctx, hipMemsetAsync(h, 0, hbytes, ctx->stream));
    if (ncols > 0 && npix > 0) {
        const uint32_t nc = (uint32_t)ncols;
        const int64_t nrows = npix / ncols, RCH = 65535ll * AGG_ROWS;      // grid.y limit
        for (int64_t r = 0; r < nrows; r += RCH) {
            const uint32_t nr = (uint32_t)((nrows - r < RCH) ? (nrows - r) : RCH);
            hipLaunchKernelGGL(k_hist_patch, dim3(grid_for(nc, 64), grid_for(nr, AGG_ROWS)), dim3(256), 0,
                               ctx->stream, d_raster + (size_t)r * nc, nr, nc, h);
            KCHK(ctx);
        }
    } else {
        const int64_t CH = 1ll << 30;
        for (int64_t o = 0; o < npix; o += CH) {
            const uint32_t m = (uint32_t)((npix - o < CH) ? (npix - o) : CH);
            hipLaunchKernelGGL(k_run_count, dim3(grid_for(m, 256)), dim3(256), 0, ctx->stream, d_raster + o, m,
                               h, 0u, 1);
            KCHK(ctx);
        }
    }
    // read back through pinned staging (a pageable destination: 1-30 ms for 10 MB, erratic)
    if (ctx->h_fit_cap < hbytes) {
        if (ctx->h_fit) hipHostFree(ctx->h_fit);
        ctx->h_fit = nullptr; ctx->h_fit_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_fit, hbytes + hbytes / 4, hipHostMallocDefault) != hipSuccess)
            SHP_FAIL(ctx, SHP_ERR_NOMEM, "hipHostMalloc(%zu) failed", hbytes);
        ctx->h_fit_cap = hbytes + hbytes / 4;
    }
    lap("launched");
    if (io_timing) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); lap("kernel done"); }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_fit, h, hbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    lap("on the host");
    memcpy(hist_out_host, ctx->h_fit, hbytes);
    lap("copied out");
    hist_out_host[0] = 0;
    return 0;
}

// accumulated device time (ms) and launch count of the instrumented kernels of this context:
// ids: 0 assign, 1 ccl (init+merge+flatten), 2 dfs_split, 3 radix sort, 4 spectra,
//      5 small-segment pass loop, 6 (unused), 7 seed scan + final labels.  reset != 0 clears.
API int shp_prof_get(shp_ctx *ctx, double *ms_out, uint64_t *count_out, int n, int reset)
{
    if (!ctx || !ms_out || !count_out) return SHP_ERR_ARG;
    prof_collect(ctx);
    for (int i = 0; i < n && i < PROF_N; i++) { ms_out[i] = ctx->prof_ms[i]; count_out[i] = ctx->prof_cnt[i]; }
    if (reset) for (int i = 0; i < PROF_N; i++) { ctx->prof_ms[i] = 0; ctx->prof_cnt[i] = 0; }
    return 0;
}

// ---- per-segment statistics (tilingstats) -------------------------------------------------------
API int shp_segstats_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                         int64_t npix, uint32_t max_seg_id, int has_null, int64_t null_val,
                         const uint32_t *stats_sel, int nstats, int64_t missing,
                         int64_t *intcols_out, float *floatcols_out)
{
    CHK(enter(ctx));
    if (!d_seg || !d_band || !stats_sel || nstats < 1 || dtype_size(dtype) == 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (npix < 0 || npix >= 0xffffffffll) SHP_FAIL(ctx, SHP_ERR_ARG, "raster too large (%lld px)", (long long)npix);
    return run_segstats(ctx, d_seg, d_band, dtype, (uint32_t)npix, max_seg_id, has_null, null_val,
                        stats_sel, nstats, missing, intcols_out, floatcols_out);
}

// the same for a raster whose shape is known (nrows x ncols pixels, row-major): lets the library take the
// patch-by-patch path when the segments are small (segstats.h)
API int shp_segstats2d_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                           int64_t nrows, int64_t ncols, uint32_t max_seg_id, int has_null, int64_t null_val,
                           const uint32_t *stats_sel, int nstats, int64_t missing,
                           int64_t *intcols_out, float *floatcols_out)
{
    CHK(enter(ctx));
    if (!d_seg || !d_band || !stats_sel || nstats < 1 || dtype_size(dtype) == 0 || nrows < 0 || ncols < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (nrows > 0xffffffffll || ncols > 0xffffffffll || nrows * ncols >= 0xffffffffll)
        SHP_FAIL(ctx, SHP_ERR_ARG, "raster too large (%lld x %lld px)", (long long)nrows, (long long)ncols);
    return run_segstats(ctx, d_seg, d_band, dtype, (uint32_t)(nrows * ncols), max_seg_id, has_null, null_val,
                        stats_sel, nstats, missing, intcols_out, floatcols_out, (uint32_t)nrows, (uint32_t)ncols);
}

API int shp_segstats(shp_ctx *ctx, const uint32_t *seg, const void *band, int dtype, int64_t npix,
                     uint32_t max_seg_id, int has_null, int64_t null_val, const uint32_t *stats_sel,
                     int nstats, int64_t missing, int64_t *intcols_out, float *floatcols_out)
{
    CHK(enter(ctx));
    if (!seg || !band || !stats_sel || nstats < 1 || dtype_size(dtype) == 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (npix < 0 || npix >= 0xffffffffll) SHP_FAIL(ctx, SHP_ERR_ARG, "raster too large (%lld px)", (long long)npix);
    CHK(buf_ensure(ctx, ctx->seg, (size_t)npix * 4));
    CHK(buf_ensure(ctx, ctx->img, (size_t)npix * dtype_size(dtype)));
    if (npix) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg, (size_t)npix * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->img.p, band, (size_t)npix * dtype_size(dtype), hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    return run_segstats(ctx, bp<uint32_t>(ctx->seg), ctx->img.p, dtype, (uint32_t)npix, max_seg_id,
                        has_null, null_val, stats_sel, nstats, missing, intcols_out, floatcols_out);
}

API int shp_gather_flagged_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                               int64_t npix, uint32_t max_seg_id, const uint8_t *flags,
                               int64_t cap, uint32_t *seg_out, int64_t *val_out, int64_t *count_out)
{
    CHK(enter(ctx));
    if (!d_seg || !d_band || !flags || !count_out || dtype_size(dtype) == 0 || cap < 0 ||
        (cap > 0 && (!seg_out || !val_out)))
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (npix < 0 || npix >= 0xffffffffll || cap >= 0xffffffffll)
        SHP_FAIL(ctx, SHP_ERR_ARG, "raster too large (%lld px)", (long long)npix);
    return run_gather_flagged(ctx, d_seg, d_band, dtype, (uint32_t)npix, max_seg_id, flags, (uint32_t)cap,
                              seg_out, val_out, count_out);
}

// the multi-GPU split with everything left in device memory (segstats.h: run_dstats_local / run_dstats_merge)
API int shp_dstats_local_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype, int64_t nrows,
                             int64_t ncols, uint32_t max_seg_id, int has_null, int64_t null_val,
                             const uint32_t *stats_sel, int nstats, int64_t missing, const uint32_t *d_hist,
                             int keep_unheld, void *d_cols, void **d_pair_seg_out, void **d_pair_val_out,
                             int64_t *n_pairs_out, int64_t *n_straddlers_out)
{
    CHK(enter(ctx));
    if (!d_seg || !d_band || !stats_sel || !d_hist || !d_cols || !d_pair_seg_out || !d_pair_val_out || !n_pairs_out ||
        !n_straddlers_out || nstats < 1 || dtype_size(dtype) == 0 || nrows < 0 || ncols < 0)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (nrows > 0xffffffffll || ncols > 0xffffffffll || nrows * ncols >= 0xffffffffll)
        SHP_FAIL(ctx, SHP_ERR_ARG, "raster too large (%lld x %lld px)", (long long)nrows, (long long)ncols);
    for (int i = 0; i < nstats; i++)
        if (stats_sel[i * 5 + 1] > 7u || stats_sel[i * 5 + 2] > 1u) SHP_FAIL(ctx, SHP_ERR_ARG, "bad statsSelection entry %d", i);
    uint32_t *ps = nullptr;
    long long *pv = nullptr;
    CHK(run_dstats_local(ctx, d_seg, d_band, dtype, (uint32_t)nrows, (uint32_t)ncols, max_seg_id, has_null, null_val,
                         stats_sel, nstats, missing, d_hist, keep_unheld, d_cols, &ps, &pv, n_pairs_out, n_straddlers_out));
    *d_pair_seg_out = ps;
    *d_pair_val_out = pv;
    return 0;
}

API int shp_dstats_merge_dev(shp_ctx *ctx, const uint32_t *d_pair_seg, const int64_t *d_pair_val, int64_t slot, int world,
                             const uint32_t *counts, int dtype, uint32_t max_seg_id, int has_null, int64_t null_val,
                             const uint32_t *stats_sel, int nstats, int64_t missing, uint32_t id_lo, uint32_t id_hi,
                             void *d_cols, int64_t *n_merged_out, int64_t *n_ids_out)
{
    CHK(enter(ctx));
    if (!stats_sel || !counts || !d_cols || !n_merged_out || !n_ids_out || nstats < 1 || dtype_size(dtype) == 0 || slot < 0 || world < 1 ||
        slot >= 0xffffffffll || (slot > 0 && (!d_pair_seg || !d_pair_val)))
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    for (int i = 0; i < nstats; i++)
        if (stats_sel[i * 5 + 1] > 7u || stats_sel[i * 5 + 2] > 1u) SHP_FAIL(ctx, SHP_ERR_ARG, "bad statsSelection entry %d", i);
    for (int r = 0; r < world; r++) if ((int64_t)counts[r] > slot) SHP_FAIL(ctx, SHP_ERR_ARG, "counts[%d] exceeds the slot", r);
    return run_dstats_merge(ctx, d_pair_seg, (const long long *)d_pair_val, (uint32_t)slot, (uint32_t)world, counts, dtype,
                            max_seg_id, has_null, null_val, stats_sel, nstats, missing, id_lo, id_hi, d_cols, n_merged_out, n_ids_out);
}

// ---- subset ---------------------------------------------------------------------
static int subset_check(shp_ctx *ctx, int64_t img_rows, int64_t img_cols, int64_t tlx, int64_t tly,
                        int64_t xs, int64_t ys, int tile_size)
{
    if (img_rows < 0 || img_cols < 0 || tlx < 0 || tly < 0 || xs < 0 || ys < 0 || tile_size < 1)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (tlx + xs > img_cols || tly + ys > img_rows)
        SHP_FAIL(ctx, SHP_ERR_ARG, "Requested subset is not within input image");      // subset.py:86-88
    if ((uint64_t)xs * (uint64_t)ys >= 0xffffffffull || img_cols >= 0xffffffffll)
        SHP_FAIL(ctx, SHP_ERR_ARG, "subset too large");
    return 0;
}

API int shp_subset_recode_dev(shp_ctx *ctx, const uint32_t *d_seg, int64_t img_rows, int64_t img_cols,
                              int64_t tlx, int64_t tly, int64_t xs, int64_t ys, const uint8_t *d_mask,
                              int tile_size, uint32_t max_seg_id, uint32_t *d_out, uint32_t *orig_out,
                              uint32_t *hist_out, int64_t cap, uint32_t *n_new_out)
{
    CHK(enter(ctx));
    if (!d_seg || !d_out || !orig_out || !hist_out || !n_new_out || cap < 1)
        SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    CHK(subset_check(ctx, img_rows, img_cols, tlx, tly, xs, ys, tile_size));
    return run_subset_recode(ctx, d_seg, (uint32_t)img_cols, (uint32_t)tlx, (uint32_t)tly, (uint32_t)xs,
                             (uint32_t)ys, d_mask, (uint32_t)tile_size, max_seg_id, d_out, orig_out,
                             hist_out, cap, n_new_out);
}

API int shp_subset_recode(shp_ctx *ctx, const uint32_t *seg, int64_t img_rows, int64_t img_cols,
                          int64_t tlx, int64_t tly, int64_t xs, int64_t ys, const uint8_t *mask,
                          int tile_size, uint32_t max_seg_id, uint32_t *out, uint32_t *orig_out,
                          uint32_t *hist_out, int64_t cap, uint32_t *n_new_out)
{
    CHK(enter(ctx));
    if (!seg || !out || !orig_out || !hist_out || !n_new_out || cap < 1)
        SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    CHK(subset_check(ctx, img_rows, img_cols, tlx, tly, xs, ys, tile_size));
    const size_t n = (size_t)xs * (size_t)ys;
    *n_new_out = 0;
    if (n == 0) return 0;
    // only the window travels: rows of xs labels out of a raster of img_cols
    CHK(buf_ensure(ctx, ctx->seg, n * 4));
    CHK(buf_ensure(ctx, ctx->lab, n * 4));
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->seg.p, (size_t)xs * 4, seg + (size_t)tly * img_cols + tlx,
                                 (size_t)img_cols * 4, (size_t)xs * 4, (size_t)ys, hipMemcpyHostToDevice,
                                 ctx->stream));
    uint8_t *d_mask = nullptr;
    if (mask) {
        CHK(buf_ensure(ctx, ctx->clus, n));
        HIPCHK(ctx, hipMemcpyAsync(ctx->clus.p, mask, n, hipMemcpyHostToDevice, ctx->stream));
        d_mask = (uint8_t *)ctx->clus.p;
    }
    CHK(run_subset_recode(ctx, bp<uint32_t>(ctx->seg), (uint32_t)xs, 0u, 0u, (uint32_t)xs, (uint32_t)ys,
                          d_mask, (uint32_t)tile_size, max_seg_id, bp<uint32_t>(ctx->lab), orig_out, hist_out,
                          cap, n_new_out));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->lab.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- spatial statistics ------------------------------------------------------------
static int spatial_check(shp_ctx *ctx, int dtype, int64_t nrows, int64_t ncols, int func,
                         const double *params, int nint, int nflt)
{
    if (dtype_size(dtype) == 0 || nrows < 0 || ncols < 0 || !params || nint < 0 || nflt < 0 || nint + nflt < 1)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    if (func < 0 || func > 2) SHP_FAIL(ctx, SHP_ERR_ARG, "unknown built-in spatial function %d", func);
    if ((uint64_t)nrows * (uint64_t)ncols >= 0xffffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "raster too large");
    return 0;
}

API int shp_spatialstats_dev(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                             int64_t nrows, int64_t ncols, uint32_t max_seg_id, int64_t null_val,
                             int func, const double *params, int64_t missing, int nint, int nflt,
                             int64_t *intcols_out, float *floatcols_out)
{
    CHK(enter(ctx));
    if (!d_seg || !d_band || (nint && !intcols_out) || (nflt && !floatcols_out))
        SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    CHK(spatial_check(ctx, dtype, nrows, ncols, func, params, nint, nflt));
    return run_spatialstats(ctx, d_seg, d_band, dtype, (uint32_t)nrows, (uint32_t)ncols, max_seg_id,
                            null_val, func, params, missing, nint, nflt, intcols_out, floatcols_out);
}

API int shp_spatialstats(shp_ctx *ctx, const uint32_t *seg, const void *band, int dtype, int64_t nrows,
                         int64_t ncols, uint32_t max_seg_id, int64_t null_val, int func,
                         const double *params, int64_t missing, int nint, int nflt,
                         int64_t *intcols_out, float *floatcols_out)
{
    CHK(enter(ctx));
    if (!seg || !band || (nint && !intcols_out) || (nflt && !floatcols_out))
        SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    CHK(spatial_check(ctx, dtype, nrows, ncols, func, params, nint, nflt));
    const size_t npix = (size_t)nrows * (size_t)ncols;
    CHK(buf_ensure(ctx, ctx->seg, npix * 4));
    CHK(buf_ensure(ctx, ctx->img, npix * dtype_size(dtype)));
    if (npix) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->seg.p, seg, npix * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->img.p, band, npix * dtype_size(dtype), hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    return run_spatialstats(ctx, bp<uint32_t>(ctx->seg), ctx->img.p, dtype, (uint32_t)nrows, (uint32_t)ncols,
                            max_seg_id, null_val, func, params, missing, nint, nflt, intcols_out,
                            floatcols_out);
}

// Grow the context's workspace for tiles of up to npix pixels now (the buffers are grow-only and
// a regrow synchronises and reallocates in the middle of a run): the tiled drivers call this once
// per worker with the largest tile of the job, so that a pooled context does not keep growing
// until it has met that tile itself.
// the buffers one tile of npix pixels makes a worker context hold, and their sizes
typedef std::vector<std::pair<DevBuf shp_ctx::*, size_t>> ReservePlan;
static ReservePlan reserve_plan(int dtype, int nbands, size_t n)
{
    const size_t ns = n / 4 + 2;
    const uint32_t maxbig = (uint32_t)(n / (MAX_CLUMP_SIZE + 2u) + 1u);
    const size_t nblk = (n + SORT_TILE - 1) / SORT_TILE, nh = (size_t)256 * (nblk ? nblk : 1);
    ReservePlan pl;
    pl.push_back({&shp_ctx::img, (size_t)nbands * n * dtype_size(dtype)});
    pl.push_back({&shp_ctx::clus, n * 2});
    DevBuf shp_ctx::*perpix[] = {&shp_ctx::lab, &shp_ctx::aux, &shp_ctx::aux2, &shp_ctx::stack,
                                 &shp_ctx::sort_k0, &shp_ctx::sort_k1, &shp_ctx::sort_v1, &shp_ctx::pix};
    for (auto m : perpix) pl.push_back({m, n * 4});
    pl.push_back({&shp_ctx::segsz, (n + 2) * 4});
    pl.push_back({&shp_ctx::singles, (n + 2) * 4});
    pl.push_back({&shp_ctx::bigbits, (n / 32 + 2) * 4});
    pl.push_back({&shp_ctx::big, (size_t)maxbig * (sizeof(BigInfo) + 4) + 128});
    {       // the replay's bitmap snapshots: a walker-pool-sized slot per walker (run_clump)
        const size_t walkers = ((size_t)maxbig + DFS_WAVES - 1) / DFS_WAVES * DFS_WAVES;
        const size_t cap = (size_t)DFS_MAX_BLOCKS * DFS_WAVES;
        pl.push_back({&shp_ctx::snap, (walkers < cap ? walkers : cap) * DFS_POOL_GRANS_DEFAULT * DFS_GRAN_WORDS * 4});
    }
    pl.push_back({&shp_ctx::sort_hist, 2 * nh * 4});
    pl.push_back({&shp_ctx::scan_tmp, scan_tmp_bytes(n > nh ? n : nh)});
    pl.push_back({&shp_ctx::chnext, ns * 16});       // the pass loop's chunk records (uint4 per segment)
    DevBuf shp_ctx::*perseg[] = {&shp_ctx::origsz, &shp_ctx::mergeto,
                                 &shp_ctx::tcount, &shp_ctx::tfill, &shp_ctx::tsorted, &shp_ctx::srclist,
                                 &shp_ctx::tgtlist};
    for (auto m : perseg) pl.push_back({m, ns * 4});
    pl.push_back({&shp_ctx::off, ns * 4 + 16});
    pl.push_back({&shp_ctx::toff, ns * 4 + 16});
    pl.push_back({&shp_ctx::tlist, (ns + 32) * 4});
    pl.push_back({&shp_ctx::ssum, ns * (size_t)nbands * 4});
    pl.push_back({&shp_ctx::small, 8192});
    return pl;
}

API int shp_ctx_reserve(shp_ctx *ctx, int dtype, int nbands, int64_t npix)
{
    CHK(enter(ctx));
    if (dtype_size(dtype) == 0 || nbands < 1 || npix < 0 || npix >= 0x7fffffffll)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    for (const auto &e : reserve_plan(dtype, nbands, (size_t)npix)) CHK(buf_ensure(ctx, ctx->*(e.first), e.second));
    return 0;
}

// How much more device memory shp_ctx_reserve(dtype, nbands, npix) would allocate on this context
// (what it already holds counts), and the device's free / total memory: the tiled driver sizes its
// worker count with these instead of running out of memory on very large tiles.
API int shp_ctx_reserve_query(shp_ctx *ctx, int dtype, int nbands, int64_t npix, int64_t *extra_bytes,
                              int64_t *free_bytes, int64_t *total_bytes)
{
    CHK(enter(ctx));
    if (dtype_size(dtype) == 0 || nbands < 1 || npix < 0 || npix >= 0x7fffffffll)
        SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    size_t extra = 0;
    for (const auto &e : reserve_plan(dtype, nbands, (size_t)npix)) {
        const size_t bytes = e.second ? e.second : 16;
        if ((ctx->*(e.first)).cap < bytes) extra += bytes + bytes / 8 + 256;       // as buf_ensure grows
    }
    if (extra_bytes) *extra_bytes = (int64_t)extra;
    if (free_bytes || total_bytes) {
        size_t fr = 0, tot = 0;
        HIPCHK(ctx, hipMemGetInfo(&fr, &tot));
        if (free_bytes) *free_bytes = (int64_t)fr;
        if (total_bytes) *total_bytes = (int64_t)tot;
    }
    return 0;
}

// ---- RCCL communicator (multi-GPU stitch exchange) ----------------------------------------------
API int shp_comm_unique_id(void *id_out_128)
{
    if (!id_out_128) return SHP_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    return ncclGetUniqueId((ncclUniqueId *)id_out_128) == ncclSuccess ? SHP_OK : SHP_ERR_HIP;
}

API int shp_comm_create(shp_ctx *ctx, int rank, int world, const void *unique_id_128, shp_comm **out)
{
    CHK(enter(ctx));
    if (!out || !unique_id_128 || world < 1 || rank < 0 || rank >= world) SHP_FAIL(ctx, SHP_ERR_ARG, "bad argument");
    *out = nullptr;
    shp_comm *cm = new shp_comm();
    cm->ctx = ctx; cm->rank = rank; cm->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&cm->nc, world, id, rank);        // (the context's device is current)
    if (r != ncclSuccess) {
        delete cm;
        SHP_FAIL(ctx, SHP_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&cm->cstream, hipStreamNonBlocking) != hipSuccess) {
        ncclCommDestroy(cm->nc);
        delete cm;
        SHP_FAIL(ctx, SHP_ERR_HIP, "hipStreamCreate for the communicator failed");
    }
    *out = cm;
    return 0;
}

API void shp_comm_destroy(shp_comm *cm)
{
    if (!cm) return;
    if (cm->ctx) { hipSetDevice(cm->ctx->device); hipStreamSynchronize(cm->ctx->stream); }
    if (cm->cstream) { hipStreamSynchronize(cm->cstream); hipStreamDestroy(cm->cstream); }
    for (hipEvent_t e : cm->evpool) hipEventDestroy(e);
    if (cm->nc) ncclCommDestroy(cm->nc);
    delete cm;
}

// what RCCL itself says the communicator spans (ncclCommCount): bench lines quote it
API int shp_comm_count(shp_comm *cm, int *nranks_out)
{
    if (!cm || !nranks_out) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    NCCLCHK(cm->ctx, ncclCommCount(cm->nc, nranks_out));
    return 0;
}

// ncclGroupStart / ncclGroupEnd around the asynchronous calls (a send and its matching receive of ONE
// rank must be grouped; between different ranks the calls pair up by themselves)
API int shp_comm_group(shp_comm *cm, int begin)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    NCCLCHK(cm->ctx, begin ? ncclGroupStart() : ncclGroupEnd());
    return 0;
}

// Asynchronous send of a device buffer that `producer`'s stream is still writing: enqueued on the
// communicator's stream behind an event recorded on the producer's stream now.  Returns at once.
API int shp_comm_isend(shp_comm *cm, const void *d_buf, size_t bytes, int dst, shp_ctx *producer)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if ((!d_buf && bytes) || dst < 0 || dst >= cm->world) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "bad argument");
    if (producer) {
        hipEvent_t e = comm_event(cm);
        if (!e) SHP_FAIL(cm->ctx, SHP_ERR_HIP, "hipEventCreate failed");
        HIPCHK(cm->ctx, hipEventRecord(e, producer->stream));
        HIPCHK(cm->ctx, hipStreamWaitEvent(cm->cstream, e, 0));
    }
    NCCLCHK(cm->ctx, ncclSend(d_buf, bytes, ncclUint8, dst, cm->nc, cm->cstream));
    return 0;
}

// Asynchronous receive into a device buffer that `consumer`'s stream will read: enqueued on the
// communicator's stream; the consumer's stream is made to wait (on the device) for its completion.
API int shp_comm_irecv(shp_comm *cm, void *d_buf, size_t bytes, int src, shp_ctx *consumer)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if ((!d_buf && bytes) || src < 0 || src >= cm->world) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "bad argument");
    NCCLCHK(cm->ctx, ncclRecv(d_buf, bytes, ncclUint8, src, cm->nc, cm->cstream));
    if (consumer) {
        hipEvent_t e = comm_event(cm);
        if (!e) SHP_FAIL(cm->ctx, SHP_ERR_HIP, "hipEventCreate failed");
        HIPCHK(cm->ctx, hipEventRecord(e, cm->cstream));
        HIPCHK(cm->ctx, hipStreamWaitEvent(consumer->stream, e, 0));
    }
    return 0;
}

// `consumer`'s stream waits (on the device) for everything enqueued on the communicator's stream so far:
// for receives issued inside a group, whose ncclRecv is only enqueued by shp_comm_group(0)
API int shp_comm_wait(shp_comm *cm, shp_ctx *consumer)
{
    if (!cm || !consumer) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    hipEvent_t e = comm_event(cm);
    if (!e) SHP_FAIL(cm->ctx, SHP_ERR_HIP, "hipEventCreate failed");
    HIPCHK(cm->ctx, hipEventRecord(e, cm->cstream));
    HIPCHK(cm->ctx, hipStreamWaitEvent(consumer->stream, e, 0));
    return 0;
}

// host waits until every asynchronous operation issued so far has completed
API int shp_comm_drain(shp_comm *cm)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    HIPCHK(cm->ctx, hipStreamSynchronize(cm->cstream));
    return 0;
}

API int shp_comm_send(shp_comm *cm, const void *d_buf, size_t bytes, int dst)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if ((!d_buf && bytes) || dst < 0 || dst >= cm->world) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "bad argument");
    NCCLCHK(cm->ctx, ncclSend(d_buf, bytes, ncclUint8, dst, cm->nc, cm->ctx->stream));
    return comm_finish(cm);
}

API int shp_comm_recv(shp_comm *cm, void *d_buf, size_t bytes, int src)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if ((!d_buf && bytes) || src < 0 || src >= cm->world) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "bad argument");
    NCCLCHK(cm->ctx, ncclRecv(d_buf, bytes, ncclUint8, src, cm->nc, cm->ctx->stream));
    return comm_finish(cm);
}

API int shp_comm_bcast(shp_comm *cm, void *d_buf, size_t bytes, int root)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if ((!d_buf && bytes) || root < 0 || root >= cm->world) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "bad argument");
    NCCLCHK(cm->ctx, ncclBroadcast(d_buf, d_buf, bytes, ncclUint8, root, cm->nc, cm->ctx->stream));
    return comm_finish(cm);
}

// every rank contributes bytes_per_rank bytes; d_recv holds world * bytes_per_rank
API int shp_comm_allgather(shp_comm *cm, const void *d_send, void *d_recv, size_t bytes_per_rank)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if (!d_send || !d_recv) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "NULL argument");
    NCCLCHK(cm->ctx, ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8, cm->nc, cm->ctx->stream));
    return comm_finish(cm);
}

// in place; op 0 = sum of int64, 1 = max of float64
API int shp_comm_allreduce(shp_comm *cm, void *d_buf, size_t count, int op)
{
    if (!cm) return SHP_ERR_ARG;
    CHK(enter(cm->ctx));
    if ((!d_buf && count) || (op != 0 && op != 1)) SHP_FAIL(cm->ctx, SHP_ERR_ARG, "bad argument");
    NCCLCHK(cm->ctx, ncclAllReduce(d_buf, d_buf, count, op == 0 ? ncclInt64 : ncclFloat64,
                                   op == 0 ? ncclSum : ncclMax, cm->nc, cm->ctx->stream));
    return comm_finish(cm);
}
