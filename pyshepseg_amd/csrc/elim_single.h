// elim_single.h -- single-pixel elimination and segment-id compaction.
//
// Replaces shepseg.eliminateSinglePixels / mergeSinglePixels / findNearestNeighbourPixel /
// relabelSegments (shepseg.py:572-777).  Each pass of the reference is a Jacobi step: the scan
// phase reads a frozen (seg, segSize); the apply phase only relabels single pixels, whose
// targets (segments of size > 1) never move in the same pass, so one thread per pixel plus
// integer atomics on segSize reproduces it exactly.  Distances are exact int64,
// scan order rows-outer / cols-inner with strict '<' (N3); segment 0 (null) is a legal target
// when it holds more than one pixel (N4).
#pragma once
#include "common.h"
#include "scan.h"

#define NO_TARGET 0xFFFFFFFFu

__global__ __launch_bounds__(256) void k_single_scan(
    const void *__restrict__ img, int dtype, int nb, const uint32_t *__restrict__ seg,
    const uint32_t *__restrict__ segsz, uint32_t *__restrict__ tgt, uint32_t n, uint32_t nrows,
    uint32_t ncols, int four)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    uint32_t out = NO_TARGET;
    if (segsz[seg[p]] == 1u) {
        const uint32_t i = p / ncols, j = p - i * ncols;
        const uint32_t i0 = i > 0 ? i - 1 : 0, i1 = (i + 1 < nrows) ? i + 1 : nrows - 1;
        const uint32_t j0 = j > 0 ? j - 1 : 0, j1 = (j + 1 < ncols) ? j + 1 : ncols - 1;
        long long mind = -1;
        for (uint32_t a = i0; a <= i1; a++)
            for (uint32_t b = j0; b <= j1; b++) {
                if (four && a != i && b != j) continue;
                const uint32_t q = a * ncols + b;
                const uint32_t sn = seg[q];
                if (segsz[sn] > 1u) {
                    long long d = 0;
                    for (int k = 0; k < nb; k++) {
                        const long long t = ld_px(img, dtype, (size_t)k * n + p) -
                                            ld_px(img, dtype, (size_t)k * n + q);
                        d += t * t;
                    }
                    if (mind < 0 || d < mind) { mind = d; out = sn; }
                }
            }
    }
    tgt[p] = out;
}

__global__ __launch_bounds__(256) void k_single_apply(uint32_t *__restrict__ seg, uint32_t *segsz,
                                                      const uint32_t *__restrict__ tgt, uint32_t n,
                                                      uint32_t *nelim)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = (p < n) ? tgt[p] : NO_TARGET;
    const bool act = t != NO_TARGET;
    if (act) {
        const uint32_t old = seg[p];
        seg[p] = t;
        segsz[old] = 0;
        atomicAdd(&segsz[t], 1u);
    }
    // the host only needs "did this pass merge anything" (the count is oldMax - newMax later)
    if (act) *nelim = 1u;
}

// relabelSegments (shepseg.py:739-777): newid[k] = k - #{1 <= j < k : segsz[j] == 0}
struct EmptyFn {      // f(i) = 1 if id i (>= 1) is unused; f(0) = 0
    const uint32_t *segsz;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const
    {
        return (i >= 1u && segsz[i] == 0u) ? 1u : 0u;
    }
};

__global__ __launch_bounds__(256) void k_relabel(uint32_t *__restrict__ seg,
                                                 const uint32_t *__restrict__ sub, uint32_t n)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t s = seg[p];
    seg[p] = s - sub[s];
}

// Compacts ids in d_seg given segsz[0..max_id].  *new_max_host = max_id - (#unused ids >= 1)
// which equals seg.max() after the relabel (0 when every pixel is null).
static int run_relabel(shp_ctx *ctx, uint32_t *d_seg, uint32_t n, const uint32_t *d_segsz,
                       uint32_t max_id, uint32_t *new_max_host)
{
    const uint32_t ns = max_id + 1u;
    CHK(buf_ensure(ctx, ctx->toff, (size_t)ns * 4 + 16));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns)));
    uint32_t *sub = bp<uint32_t>(ctx->toff);
    uint32_t *tot = sub + ns;
    EmptyFn f{d_segsz};
    CHK(scan_exclusive(ctx, f, ns, sub, tot, bp<uint32_t>(ctx->scan_tmp)));
    if (n) {
        hipLaunchKernelGGL(k_relabel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_seg,
                           sub, n);
        KCHK(ctx);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, tot, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *new_max_host = max_id - ctx->h_pinned[0];
    return 0;
}

// segsz[0..max_id] = histogram of d_seg (makeSegSize, shepseg.py:544-569)
static int run_seg_size(shp_ctx *ctx, const uint32_t *d_seg, uint32_t n, uint32_t max_id,
                        uint32_t *d_segsz)
{
    HIPCHK(ctx, hipMemsetAsync(d_segsz, 0, ((size_t)max_id + 1) * 4, ctx->stream));
    if (n) {
        hipLaunchKernelGGL(k_run_count, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_seg, n,
                           d_segsz, 0u, 0);
        KCHK(ctx);
    }
    return 0;
}

// d_seg: clump ids (in place).  max_id in: largest id; out: largest id after relabel.
static int run_eliminate_single(shp_ctx *ctx, const void *d_img, int dtype, int nb, uint32_t nrows,
                                uint32_t ncols, int four, uint32_t *d_seg, uint32_t *max_id)
{
    const uint32_t n = nrows * ncols;
    CHK(buf_ensure(ctx, ctx->segsz, ((size_t)*max_id + 2) * 4));
    CHK(buf_ensure(ctx, ctx->aux, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->small, 4096));
    uint32_t *segsz = bp<uint32_t>(ctx->segsz), *tgt = bp<uint32_t>(ctx->aux);
    uint32_t *nelim = bp<uint32_t>(ctx->small);
    CHK(run_seg_size(ctx, d_seg, n, *max_id, segsz));
    if (n == 0) return 0;
    const unsigned g = grid_for(n, 256);
    for (;;) {
        HIPCHK(ctx, hipMemsetAsync(nelim, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_single_scan, dim3(g), dim3(256), 0, ctx->stream, d_img, dtype, nb,
                           d_seg, segsz, tgt, n, nrows, ncols, four);
        KCHK(ctx);
        hipLaunchKernelGGL(k_single_apply, dim3(g), dim3(256), 0, ctx->stream, d_seg, segsz, tgt, n,
                           nelim);
        KCHK(ctx);
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, nelim, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_pinned[0] == 0) break;
    }
    uint32_t new_max = 0;
    CHK(run_relabel(ctx, d_seg, n, segsz, *max_id, &new_max));
    *max_id = new_max;
    return 0;
}
