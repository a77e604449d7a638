// api.hip -- the extern "C" boundary of libshepseg_hip.so (see include/shepseg_hip.h).
// gfx950 only.  No CPU fallback: without a usable device every call fails.
#include "common.h"
#include "scan.h"
#include "sort.h"
#include "kmeans.h"
#include "clump.h"
#include "elim_single.h"
#include "elim_small.h"
#include "synth.h"

#define API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_create_err;

API int shp_version(void) { return 100; }

API int shp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

API int shp_ctx_create(int device, shp_ctx **out)
{
    if (!out) return SHP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SHP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SHP_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return SHP_ERR_HIP;
    shp_ctx *ctx = new shp_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return SHP_ERR_HIP;
    }
    if (hipHostMalloc((void **)&ctx->h_pinned, 4096, hipHostMallocDefault) != hipSuccess) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return SHP_ERR_HIP;
    }
    for (int i = 0; i < 16; i++) hipEventCreate(&ctx->ev[i]);
    ctx->bufs = {&ctx->img, &ctx->clus, &ctx->lab, &ctx->seg, &ctx->aux, &ctx->aux2, &ctx->stack,
                 &ctx->scan_tmp, &ctx->sort_k0, &ctx->sort_k1, &ctx->sort_v1, &ctx->sort_hist,
                 &ctx->pix, &ctx->segsz, &ctx->origsz, &ctx->off, &ctx->ssum, &ctx->chnext,
                 &ctx->chtail, &ctx->mergeto, &ctx->tcount, &ctx->toff, &ctx->tfill, &ctx->tlist,
                 &ctx->tsorted, &ctx->small, &ctx->cen, &ctx->fit_x, &ctx->fit_lab, &ctx->fit_part,
                 &ctx->big};
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
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    hipStreamDestroy(ctx->stream);
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

static float ev_ms(shp_ctx *ctx, int a, int b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) return 0.f;
    return ms;
}

// ---- k-means --------------------------------------------------------------------------------
API int shp_kmeans_fit(shp_ctx *ctx, const double *xsample, int64_t nrows, int nbands, int k,
                       const double *init_centres, int max_iter, double tol_rel,
                       double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    CHK(enter(ctx));
    if (!xsample || !init_centres || !centres_out) SHP_FAIL(ctx, SHP_ERR_ARG, "NULL argument");
    return run_kmeans_fit(ctx, xsample, nrows, nbands, k, init_centres, max_iter, tol_rel,
                          centres_out, labels_out, n_iter_out);
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

// Device-resident fused pipeline on ctx->img -> ctx->seg.  Records stage events 1..5.
static int segment_device(shp_ctx *ctx, int dtype, int nb, uint32_t nrows, uint32_t ncols,
                          const double *centres, int k, int has_null, int64_t null_val, int four,
                          int min_seg_size, double msd, uint32_t *max_seg_id, int64_t *singles,
                          int64_t *small, uint32_t *nclumps_out)
{
    const uint32_t n = nrows * ncols;
    CHK(buf_ensure(ctx, ctx->clus, (size_t)n * 2));
    CHK(buf_ensure(ctx, ctx->seg, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->small, 4096));
    uint32_t *d_seg = bp<uint32_t>(ctx->seg);
    uint32_t *scal = bp<uint32_t>(ctx->small);
    hipEventRecord(ctx->ev[1], ctx->stream);
    CHK(launch_assign(ctx, ctx->img.p, dtype, nb, n, centres, k, has_null, null_val,
                      bp<uint16_t>(ctx->clus), nullptr));
    hipEventRecord(ctx->ev[2], ctx->stream);
    CHK(run_clump(ctx, bp<uint16_t>(ctx->clus), nrows, ncols, four, d_seg, scal + 2));
    uint32_t nclumps = 0;
    CHK(read_u32(ctx, scal + 2, &nclumps));
    hipEventRecord(ctx->ev[3], ctx->stream);
    uint32_t max_id = nclumps;
    CHK(run_eliminate_single(ctx, ctx->img.p, dtype, nb, nrows, ncols, four, d_seg, &max_id));
    hipEventRecord(ctx->ev[4], ctx->stream);
    if (singles) *singles = (int64_t)nclumps - (int64_t)max_id;        // shepseg.py:226-227
    int64_t ne = 0;
    CHK(run_eliminate_small(ctx, ctx->img.p, dtype, nb, nrows, ncols, four, min_seg_size, msd, d_seg,
                            &max_id, &ne));
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
    CHK(segment_device(ctx, dtype, nbands, nrows, ncols, centres, k, has_null, null_val,
                       four_connected, min_seg_size, max_spectral_diff, max_seg_id_out,
                       singles_elim_out, small_elim_out, num_clumps_out));
    HIPCHK(ctx, hipMemcpyAsync(seg_out, ctx->seg.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    hipEventRecord(ctx->ev[6], ctx->stream);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timings[0] = ev_ms(ctx, 1, 2);
    ctx->timings[1] = ev_ms(ctx, 2, 3);
    ctx->timings[2] = ev_ms(ctx, 3, 4);
    ctx->timings[3] = ev_ms(ctx, 4, 5);
    ctx->timings[4] = ev_ms(ctx, 0, 1);
    ctx->timings[5] = ev_ms(ctx, 5, 6);
    ctx->timings[6] = ev_ms(ctx, 0, 6);
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
