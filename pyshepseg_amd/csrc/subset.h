// subset.h -- subset.subsetImage's recode on the device.
//
// Replaces the tile loop of subset.subsetImage (subset.py:124-166) and its njit kernel
// processSubsetTile (subset.py:366-425): the window is visited tile by tile (tile rows outer,
// tile columns inner, raster order inside a tile) and an id gets the next new number the first
// time it is seen.  "First seen" is a minimum: every window pixel has a position in that visiting
// order (subset_key), first[id] = min key over the id's unmasked pixels (only pixels that have
// no unmasked same-id neighbour to the left or above inside their tile can hold it), and the new
// id is the rank of first[id] among the ids present -- one radix sort of (key, id) pairs.
// HBM-bound: 4 B (+1 B mask) in twice, 4 B out per pixel.
#pragma once
#include "common.h"
#include "scan.h"
#include "sort.h"
#include "clump.h"      // k_run_count
#include "elim_small.h" // bits_for

struct SubsetGeom {
    const uint32_t *seg;        // label raster, row pitch img_cols
    const uint8_t *mask;        // xs*ys bytes or nullptr
    uint32_t img_cols, tlx, tly, xs, ys, T;
};

__device__ __forceinline__ uint32_t subset_key(const SubsetGeom &g, uint32_t r, uint32_t c)
{
    const uint32_t tr = r / g.T, tc = c / g.T;
    const uint32_t th = g.ys - tr * g.T < g.T ? g.ys - tr * g.T : g.T;
    const uint32_t tw = g.xs - tc * g.T < g.T ? g.xs - tc * g.T : g.T;
    return tr * g.T * g.xs + tc * g.T * th + (r - tr * g.T) * tw + (c - tc * g.T);
}

__device__ __forceinline__ uint32_t subset_id(const SubsetGeom &g, uint32_t r, uint32_t c)
{
    if (g.mask && g.mask[(size_t)r * g.xs + c] == 0) return 0u;
    return g.seg[(size_t)(g.tly + r) * g.img_cols + (g.tlx + c)];
}

__global__ __launch_bounds__(256) void k_subset_first(SubsetGeom g, uint32_t max_id, uint32_t *first,
                                                      uint32_t *bad)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= g.xs * g.ys) return;
    const uint32_t r = p / g.xs, c = p - r * g.xs;
    const uint32_t s = subset_id(g, r, c);
    if (s == 0u) return;
    if (s > max_id) { *bad = 1u; return; }
    // a same-id pixel to the left / above in the same tile comes earlier in the visiting order
    if (c % g.T != 0u && subset_id(g, r, c - 1) == s) return;
    if (r % g.T != 0u && subset_id(g, r - 1, c) == s) return;
    const uint32_t key = subset_key(g, r, c);
    if (key < first[s]) atomicMin(&first[s], key);
}

struct PresentFn {
    const uint32_t *first;
    __device__ __forceinline__ uint32_t operator()(uint32_t s) const
    {
        return (s != 0u && first[s] != 0xFFFFFFFFu) ? 1u : 0u;
    }
};

__global__ __launch_bounds__(256) void k_subset_list(const uint32_t *__restrict__ first,
                                                     const uint32_t *__restrict__ slot, uint32_t max_id,
                                                     uint32_t *__restrict__ keys, uint32_t *__restrict__ ids)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s == 0u || s > max_id) return;
    const uint32_t f = first[s];
    if (f == 0xFFFFFFFFu) return;
    keys[slot[s]] = f;
    ids[slot[s]] = s;
}

// sorted ids -> lut[old] = rank + 1, orig[rank + 1] = old
__global__ __launch_bounds__(256) void k_subset_lut(const uint32_t *__restrict__ ids, uint32_t m,
                                                    uint32_t *__restrict__ lut, uint32_t *__restrict__ orig)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t s = ids[i];
    lut[s] = i + 1u;
    orig[i + 1u] = s;
}

__global__ __launch_bounds__(256) void k_subset_apply(SubsetGeom g, const uint32_t *__restrict__ lut,
                                                      uint32_t *__restrict__ out)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= g.xs * g.ys) return;
    const uint32_t r = p / g.xs, c = p - r * g.xs;
    const uint32_t s = subset_id(g, r, c);
    out[p] = s ? lut[s] : 0u;
}

// d_seg: device label raster (img_rows x img_cols); d_mask: device xs*ys bytes or nullptr;
// d_out: device xs*ys labels.  orig_out / hist_out: HOST arrays of cap entries.
static int run_subset_recode(shp_ctx *ctx, const uint32_t *d_seg, uint32_t img_cols, uint32_t tlx,
                             uint32_t tly, uint32_t xs, uint32_t ys, const uint8_t *d_mask,
                             uint32_t tile_size, uint32_t max_id, uint32_t *d_out, uint32_t *orig_out,
                             uint32_t *hist_out, int64_t cap, uint32_t *n_new_out)
{
    hipStream_t st = ctx->stream;
    const uint32_t n = xs * ys;
    const size_t ns = (size_t)max_id + 1;
    *n_new_out = 0;
    if (n == 0) return 0;
    CHK(buf_ensure(ctx, ctx->segsz, (ns + 1) * 4));           // first[]
    CHK(buf_ensure(ctx, ctx->off, (ns + 1) * 4 + 16));          // compaction slots
    CHK(buf_ensure(ctx, ctx->origsz, (ns + 1) * 4));            // lut
    CHK(buf_ensure(ctx, ctx->tlist, (ns + 1) * 4));             // keys
    CHK(buf_ensure(ctx, ctx->tsorted, (ns + 1) * 4));           // ids
    CHK(buf_ensure(ctx, ctx->mergeto, (ns + 1) * 4));           // orig
    CHK(buf_ensure(ctx, ctx->tcount, (ns + 1) * 4));            // hist
    CHK(buf_ensure(ctx, ctx->small, 64));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns)));
    uint32_t *first = bp<uint32_t>(ctx->segsz), *slot = bp<uint32_t>(ctx->off);
    uint32_t *lut = bp<uint32_t>(ctx->origsz), *keys = bp<uint32_t>(ctx->tlist);
    uint32_t *ids = bp<uint32_t>(ctx->tsorted), *orig = bp<uint32_t>(ctx->mergeto);
    uint32_t *hist = bp<uint32_t>(ctx->tcount), *scal = bp<uint32_t>(ctx->small);
    SubsetGeom g{d_seg, d_mask, img_cols, tlx, tly, xs, ys, tile_size};
    HIPCHK(ctx, hipMemsetAsync(first, 0xff, ns * 4, st));
    HIPCHK(ctx, hipMemsetAsync(scal, 0, 16, st));
    hipLaunchKernelGGL(k_subset_first, dim3(grid_for(n, 256)), dim3(256), 0, st, g, max_id, first, scal + 1);
    KCHK(ctx);
    PresentFn pf{first};
    CHK(scan_exclusive(ctx, pf, (uint32_t)ns, slot, scal, bp<uint32_t>(ctx->scan_tmp)));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, scal, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    const uint32_t m = ctx->h_pinned[0];
    if (ctx->h_pinned[1]) SHP_FAIL(ctx, SHP_ERR_ARG, "segment id above max_seg_id (%u) in the subset", max_id);
    if ((int64_t)m + 1 > cap)
        SHP_FAIL(ctx, SHP_ERR_ARG, "subset holds %u segments, output arrays hold %lld rows", m, (long long)cap);
    hipLaunchKernelGGL(k_subset_list, dim3(grid_for(ns, 256)), dim3(256), 0, st, first, slot, max_id, keys, ids);
    KCHK(ctx);
    uint32_t *ksorted = nullptr, *isorted = nullptr;
    CHK(sort_pairs(ctx, keys, ids, m, bits_for(n - 1u), &ksorted, &isorted));
    HIPCHK(ctx, hipMemsetAsync(orig, 0, 4, st));
    HIPCHK(ctx, hipMemsetAsync(hist, 0, ((size_t)m + 1) * 4, st));
    if (m) {
        hipLaunchKernelGGL(k_subset_lut, dim3(grid_for(m, 256)), dim3(256), 0, st, isorted, m, lut, orig);
        KCHK(ctx);
    }
    hipLaunchKernelGGL(k_subset_apply, dim3(grid_for(n, 256)), dim3(256), 0, st, g, lut, d_out); KCHK(ctx);
    hipLaunchKernelGGL(k_run_count, dim3(grid_for(n, 256)), dim3(256), 0, st, d_out, n, hist, 0u, 1); KCHK(ctx);
    HIPCHK(ctx, hipMemcpyAsync(orig_out, orig, ((size_t)m + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(hist_out, hist, ((size_t)m + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    *n_new_out = m;
    return 0;
}
