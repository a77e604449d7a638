// spatial.h -- calcPerSegmentSpatialStatsTiled with the reference's built-in user functions.
//
// Replaces the tile loop of tilingstats.calcPerSegmentSpatialStatsTiled (tilingstats.py:1262-1390:
// accumulateSegSpatial :1652-1699 collects every segment's (x, y, value) points across tiles
// until checkSegCompleteSpatial has seen them all) for userFuncMeanCoord (:1098-1142),
// userFuncNumEdgePixels (:1146-1216) and userFuncVariogram (:1037-1094).  The three are
// reductions over a segment's non-nodata pixels, so no point lists are built:
//   mean coordinates   n, sum(x), sum(y) per segment as integers (one atomic triple per run of a
//                      segment inside a wavefront), then transform applied to the sums in float64;
//   edge pixels        a pixel is an edge pixel when one of its 4 (or of the reference's 7: its
//                      8-connected test never looks at (y-1, x+1)) neighbours is not a non-nodata
//                      pixel of the same segment -- the bounding-box mask of the reference gives
//                      exactly that, the box border being made of such pixels;
//   variogram          for every pixel the (yo, xo) in 1..maxDist pairs inside the same segment,
//                      binned by floor(sqrt(yo^2 + xo^2)): integer count and integer sum of squared
//                      differences per (segment, bin) -- exact, so sqrt(sum / count) is the
//                      reference's value bit for bit.
// userFunc outputs land as the reference stores them: intArr int32 -> int64 column, floatArr
// float64 -> float32 column; unset entries and segments without a valid pixel hold `missing`.
#pragma once
#include "common.h"

struct SpatialGeom {
    const uint32_t *seg;
    const void *band;
    int dtype;
    uint32_t nrows, ncols, S;
    long long null_val;
};

__device__ __forceinline__ uint32_t spatial_member(const SpatialGeom &g, uint32_t p)
{
    const uint32_t s = g.seg[p];
    if (s == 0u || s > g.S) return 0u;
    return ld_px(g.band, g.dtype, p) != g.null_val ? s : 0u;
}

// run of equal non-zero keys inside a wavefront, broken at image row starts: head lanes get the
// run length (0 for every other lane)
__device__ __forceinline__ uint32_t spatial_runlen(uint32_t key, uint32_t col, bool inb)
{
    const unsigned lane = lane_id();
    const uint32_t pk = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || pk != key || col == 0u || !inb;
    const unsigned long long heads = __ballot(head);
    if (!head || !inb || key == 0u) return 0u;
    const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
    return (nxt ? (unsigned)__builtin_ctzll(nxt) : 64u) - lane;
}

__global__ __launch_bounds__(256) void k_spatial_sums(SpatialGeom g, uint32_t *cnt,
                                                      unsigned long long *sumx,
                                                      unsigned long long *sumy)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool inb = p < g.nrows * g.ncols;
    const uint32_t row = p / g.ncols, col = p - row * g.ncols;
    const uint32_t s = inb ? spatial_member(g, p) : 0u;
    const uint32_t len = spatial_runlen(s, col, inb);
    if (len) {
        atomicAdd(&cnt[s], len);
        if (sumx) {
            atomicAdd(&sumx[s], (unsigned long long)len * col + (unsigned long long)len * (len - 1u) / 2ull);
            atomicAdd(&sumy[s], (unsigned long long)len * row);
        }
    }
}

__global__ __launch_bounds__(256) void k_spatial_edges(SpatialGeom g, int four, uint32_t *edges)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool inb = p < g.nrows * g.ncols;
    const uint32_t row = p / g.ncols, col = p - row * g.ncols;
    const uint32_t s = inb ? spatial_member(g, p) : 0u;
    uint32_t key = 0u;
    if (s) {
        const bool up = row > 0u, dn = row + 1u < g.nrows, lf = col > 0u, rt = col + 1u < g.ncols;
#define MEM(ok, q) ((ok) && spatial_member(g, (q)) == s)
        bool inside = MEM(up, p - g.ncols) && MEM(dn, p + g.ncols) && MEM(lf, p - 1u) && MEM(rt, p + 1u);
        if (inside && !four)        // the reference's 8-connected test: (y-1, x+1) is never looked at
            inside = MEM(up && lf, p - g.ncols - 1u) && MEM(dn && rt, p + g.ncols + 1u) &&
                     MEM(dn && lf, p + g.ncols - 1u);
#undef MEM
        if (!inside) key = s;
    }
    const uint32_t len = spatial_runlen(key, col, inb);
    if (len) atomicAdd(&edges[key], len);
}

// offs: noffs packed (bin << 16 | yo << 8 | xo) sorted by bin (1-based), yo, xo
__global__ __launch_bounds__(256) void k_spatial_vario(SpatialGeom g, const uint32_t *__restrict__ offs,
                                                       uint32_t noffs, uint32_t maxd, uint32_t *vcnt,
                                                       unsigned long long *vsum)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= g.nrows * g.ncols) return;
    const uint32_t s = spatial_member(g, p);
    if (s == 0u) return;
    const uint32_t row = p / g.ncols, col = p - row * g.ncols;
    const long long v = ld_px(g.band, g.dtype, p);
    uint32_t bin = 0, c = 0;
    unsigned long long acc = 0;
    for (uint32_t i = 0; i <= noffs; i++) {
        const uint32_t o = i < noffs ? offs[i] : 0xFFFFFFFFu;
        const uint32_t b = o >> 16;
        if (b != bin) {
            if (c) {
                atomicAdd(&vcnt[(size_t)s * maxd + (bin - 1u)], c);
                atomicAdd(&vsum[(size_t)s * maxd + (bin - 1u)], acc);
            }
            bin = b; c = 0; acc = 0;
            if (i == noffs) break;
        }
        const uint32_t yo = (o >> 8) & 255u, xo = o & 255u;
        if (row + yo < g.nrows && col + xo < g.ncols) {
            const uint32_t q = p + yo * g.ncols + xo;
            if (spatial_member(g, q) == s) {
                const unsigned long long d = (unsigned long long)(v - ld_px(g.band, g.dtype, q));
                c++;
                acc += d * d;                 // (unsigned: the square of a 32-bit difference may wrap)
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_spatial_finish(
    int func, uint32_t S, const uint32_t *__restrict__ cnt, const unsigned long long *__restrict__ sumx,
    const unsigned long long *__restrict__ sumy, const uint32_t *__restrict__ edges,
    const uint32_t *__restrict__ vcnt, const unsigned long long *__restrict__ vsum, uint32_t maxd,
    const double *__restrict__ prm, long long missing, int nint, int nflt, long long *intcols,
    float *fltcols)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s > S) return;
    const size_t ns = (size_t)S + 1;
    for (int c = 0; c < nint; c++) intcols[(size_t)c * ns + s] = s ? missing : 0;
    for (int c = 0; c < nflt; c++) fltcols[(size_t)c * ns + s] = s ? (float)missing : 0.0f;
    if (s == 0u || cnt[s] == 0u) return;
    const double n = (double)cnt[s];
    if (func == 0) {
        const double sx = (double)sumx[s], sy = (double)sumy[s];
        if (nflt > 0) fltcols[s] = (float)((prm[0] * n + prm[1] * sx + prm[2] * sy) / n);
        if (nflt > 1) fltcols[ns + s] = (float)((prm[3] * n + prm[4] * sx + prm[5] * sy) / n);
    } else if (func == 1) {
        if (nint > 0) intcols[s] = (long long)(int)edges[s];
    } else {
        for (uint32_t d = 0; d < maxd && (int)d < nflt; d++) {
            const uint32_t c = vcnt[(size_t)s * maxd + d];
            if (c) fltcols[(size_t)d * ns + s] = (float)sqrt((double)vsum[(size_t)s * maxd + d] / (double)c);
        }
    }
}

// d_seg / d_band: device rasters (nrows x ncols).  Outputs are HOST arrays.
static int run_spatialstats(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                            uint32_t nrows, uint32_t ncols, uint32_t S, int64_t null_val, int func,
                            const double *params, int64_t missing, int nint, int nflt,
                            int64_t *intcols_out, float *fltcols_out)
{
    hipStream_t st = ctx->stream;
    const size_t ns = (size_t)S + 1;
    const uint32_t n = nrows * ncols;
    uint32_t maxd = 0;
    std::vector<uint32_t> offs;
    if (func == 2) {
        if (!(params[0] >= 1.0 && params[0] <= 255.0))
            SHP_FAIL(ctx, SHP_ERR_ARG, "variogram maxDist must be 1..255 (got %g)", params[0]);
        maxd = (uint32_t)params[0];
        for (uint32_t b = 1; b <= maxd; b++)
            for (uint32_t yo = 1; yo <= maxd; yo++)
                for (uint32_t xo = 1; xo <= maxd; xo++)
                    if ((uint32_t)__builtin_sqrt((double)(yo * yo + xo * xo)) == b)
                        offs.push_back((b << 16) | (yo << 8) | xo);
    }
    const size_t vrows = func == 2 ? ns * maxd : 1;
    CHK(buf_ensure(ctx, ctx->segsz, ns * 4));                                   // cnt
    CHK(buf_ensure(ctx, ctx->origsz, ns * 4));                                  // edges
    CHK(buf_ensure(ctx, ctx->aux, ns * 16));                                    // sumx | sumy
    CHK(buf_ensure(ctx, ctx->aux2, vrows * 12 + 64));                           // vsum | vcnt
    CHK(buf_ensure(ctx, ctx->small, 4096 + offs.size() * 4));
    CHK(buf_ensure(ctx, ctx->ssum, ((size_t)nint * 8 + (size_t)nflt * 4) * ns + 64));
    if (offs.size() * 4 + 64 > SHP_PINNED_BYTES) SHP_FAIL(ctx, SHP_ERR_ARG, "maxDist too large");
    uint32_t *cnt = bp<uint32_t>(ctx->segsz), *edges = bp<uint32_t>(ctx->origsz);
    unsigned long long *sumx = (unsigned long long *)ctx->aux.p, *sumy = sumx + ns;
    unsigned long long *vsum = (unsigned long long *)ctx->aux2.p;
    uint32_t *vcnt = (uint32_t *)(vsum + vrows);
    double *d_prm = (double *)ctx->small.p;
    uint32_t *d_offs = bp<uint32_t>(ctx->small) + 64;
    long long *d_int = (long long *)ctx->ssum.p;
    float *d_flt = (float *)(d_int + (size_t)nint * ns);
    HIPCHK(ctx, hipStreamSynchronize(st));
    double *pin = (double *)(ctx->h_pinned + 16);
    memcpy(pin, params, 48);
    if (!offs.empty()) memcpy(pin + 8, offs.data(), offs.size() * 4);
    HIPCHK(ctx, hipMemcpyAsync(d_prm, pin, 48, hipMemcpyHostToDevice, st));
    if (!offs.empty())
        HIPCHK(ctx, hipMemcpyAsync(d_offs, pin + 8, offs.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, ns * 4, st));
    SpatialGeom g{d_seg, d_band, dtype, nrows, ncols, S, (long long)null_val};
    const unsigned grid = grid_for(n, 256);
    if (func == 0) {
        HIPCHK(ctx, hipMemsetAsync(sumx, 0, ns * 16, st));
        if (n) hipLaunchKernelGGL(k_spatial_sums, dim3(grid), dim3(256), 0, st, g, cnt, sumx, sumy);
    } else {
        if (n) hipLaunchKernelGGL(k_spatial_sums, dim3(grid), dim3(256), 0, st, g, cnt,
                                  (unsigned long long *)nullptr, (unsigned long long *)nullptr);
        if (func == 1) {
            HIPCHK(ctx, hipMemsetAsync(edges, 0, ns * 4, st));
            if (n) hipLaunchKernelGGL(k_spatial_edges, dim3(grid), dim3(256), 0, st, g, params[0] != 0.0, edges);
        } else {
            HIPCHK(ctx, hipMemsetAsync(vsum, 0, vrows * 12, st));
            if (n) hipLaunchKernelGGL(k_spatial_vario, dim3(grid), dim3(256), 0, st, g, d_offs,
                                      (uint32_t)offs.size(), maxd, vcnt, vsum);
        }
    }
    KCHK(ctx);
    hipLaunchKernelGGL(k_spatial_finish, dim3(grid_for(ns, 256)), dim3(256), 0, st, func, S, cnt, sumx, sumy,
                       edges, vcnt, vsum, maxd, d_prm, (long long)missing, nint, nflt, d_int, d_flt);
    KCHK(ctx);
    if (nint) HIPCHK(ctx, hipMemcpyAsync(intcols_out, d_int, (size_t)nint * ns * 8, hipMemcpyDeviceToHost, st));
    if (nflt) HIPCHK(ctx, hipMemcpyAsync(fltcols_out, d_flt, (size_t)nflt * ns * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}
