// elim_small.h -- iterative small-segment elimination.
//
// Replaces shepseg.eliminateSmallSegments / buildSegmentSpectra / makeSegmentLocations /
// findMergeSegment / doMerge (shepseg.py:780-1123).
//
// Device data layout
//   pix[]      pixel indices grouped by segment id, raster order inside a segment (a CSR built
//              with one stable radix sort; == the reference's segLoc at entry, shepseg.py:880).
//   off[s]     start of segment s in pix[]; origsz[s] its length at entry.
//   chnext/chtail  a merged segment's pixel list is the chain of the ORIGINAL segments it
//              absorbed, in merge order (doMerge appends the source's list to the target's,
//              shepseg.py:1102-1110); iteration order == the reference's list order (N7).
//   ssum[s][b] float32 spectral sums.  Built by an ordered float32 accumulation over the
//              raster-ordered list (N5) -- exact-integer fast path while |partial sums| < 2^24,
//              sequential adds after that -- and merged by float32 '+=' in ascending source id
//              (shepseg.py:1117-1120), which is the reference's merge-loop order.
// Per pass (find phase / merge phase of shepseg.py:983-994): state is frozen while every
// segment of the target size picks its neighbour (one thread per segment); sources are then
// grouped by target (count, scan, fill), ranked by id inside the group, and each target
// applies its sources in ascending id.  A target is never a source in the same pass.
#pragma once
#include "common.h"
#include "scan.h"
#include "sort.h"
#include "elim_single.h"

__device__ __forceinline__ float f32_acc(float acc, long long v)
{
    return (float)((double)acc + (double)v);     // numba: float32 + pixel, stored to float32 (N5)
}

// segments with <= 64 pixels: one thread each
__global__ __launch_bounds__(256) void k_spectra_small(
    const void *__restrict__ img, int dtype, int nb, uint32_t n, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ segsz, float *__restrict__ ssum,
    uint32_t S)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u;
    if (s > S) return;
    const uint32_t m = segsz[s];
    if (m > 64u) return;
    const uint32_t o = off[s];
    for (int b = 0; b < nb; b++) {
        float acc = 0.0f;
        for (uint32_t i = 0; i < m; i++) acc = f32_acc(acc, ld_px(img, dtype, (size_t)b * n + pix[o + i]));
        ssum[(size_t)s * nb + b] = acc;
    }
}

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// segments with > 64 pixels: one wavefront each
__global__ __launch_bounds__(256) void k_spectra_big(
    const void *__restrict__ img, int dtype, int nb, uint32_t n, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ segsz, float *__restrict__ ssum,
    uint32_t S)
{
    const uint32_t s =
        (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) / 64u)) + 1u;
    if (s > S) return;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)segsz[s]);
    if (m <= 64u) return;
    const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[s]);
    const unsigned lane = lane_id();
    for (int b = 0; b < nb; b++) {
        float acc = 0.0f;
        long long exact = 0, sabs = 0;
        for (uint32_t i0 = 0; i0 < m; i0 += 64u) {
            const bool valid = i0 + lane < m;
            const long long v = valid ? ld_px(img, dtype, (size_t)b * n + pix[o + i0 + lane]) : 0;
            const long long csum = wave_sum_ll(v);
            const long long cabs = wave_sum_ll(v < 0 ? -v : v);
            if (sabs + cabs < (1ll << 24)) {
                exact += csum;                 // every partial sum is an exact float32 integer
                acc = (float)exact;
            } else {
                const uint32_t cnt = (m - i0 < 64u) ? (m - i0) : 64u;
                const double dv = (double)v;
                const int lo = __double2loint(dv), hi = __double2hiint(dv);
                for (uint32_t j = 0; j < cnt; j++) {
                    const int l = __builtin_amdgcn_readlane(lo, j);
                    const int h = __builtin_amdgcn_readlane(hi, j);
                    acc = (float)((double)acc + __hiloint2double(h, l));
                }
            }
            sabs += cabs;
        }
        if (lane == 0) ssum[(size_t)s * nb + b] = acc;
    }
}

__global__ __launch_bounds__(256) void k_small_init(const uint32_t *__restrict__ segsz,
                                                    uint32_t *__restrict__ origsz,
                                                    uint32_t *__restrict__ chnext,
                                                    uint32_t *__restrict__ chtail,
                                                    uint32_t *__restrict__ mergeto,
                                                    uint32_t *__restrict__ tcount,
                                                    uint32_t *__restrict__ tfill, uint32_t *hist,
                                                    uint32_t S, uint32_t min_seg)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s > S) return;
    const uint32_t m = segsz[s];
    origsz[s] = m;
    chnext[s] = 0;
    chtail[s] = s;
    mergeto[s] = 0;
    tcount[s] = 0;
    tfill[s] = 0;
    if (s >= 1u && m < min_seg) atomicAdd(&hist[m], 1u);
}

// findMergeSegment (shepseg.py:1003-1063), one thread per segment of the target size
__global__ __launch_bounds__(256) void k_find_merge(
    const uint32_t *__restrict__ seg, const uint32_t *__restrict__ segsz,
    const float *__restrict__ ssum, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ origsz,
    const uint32_t *__restrict__ chnext, uint32_t *__restrict__ mergeto, uint32_t S,
    uint32_t target, int nb, uint32_t nrows, uint32_t ncols, int four, double thr2)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u;
    if (s > S) return;
    if (segsz[s] != target) return;
    const float nf = (float)target;
    uint32_t best = 0, last = 0;
    float bestd = 0.0f;
    for (uint32_t c = s; c != 0; c = chnext[c]) {
        const uint32_t o = off[c], m = origsz[c];
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t p = pix[o + i];
            const uint32_t r = p / ncols, cc = p - r * ncols;
            const uint32_t r0 = r > 0 ? r - 1 : 0, r1 = (r + 1 < nrows) ? r + 1 : nrows - 1;
            const uint32_t c0 = cc > 0 ? cc - 1 : 0, c1 = (cc + 1 < ncols) ? cc + 1 : ncols - 1;
            for (uint32_t ii = r0; ii <= r1; ii++)
                for (uint32_t jj = c0; jj <= c1; jj++) {
                    if (four && ii != r && jj != cc) continue;
                    const uint32_t nbid = seg[ii * ncols + jj];
                    if (nbid == s || nbid == 0 || nbid == last) continue;
                    last = nbid;       // re-evaluating the same neighbour can never win ('<' is strict)
                    const uint32_t szn = segsz[nbid];
                    if (szn > target) {
                        const float sf = (float)szn;
                        float d = 0.0f;
                        for (int b = 0; b < nb; b++) {
                            const float a = ssum[(size_t)s * nb + b] / nf;
                            const float e = ssum[(size_t)nbid * nb + b] / sf;
                            const float t = a - e;
                            const float t2 = t * t;
                            d = d + t2;
                        }
                        if (best == 0 || d < bestd) { bestd = d; best = nbid; }
                    }
                }
        }
    }
    if (best != 0 && (double)bestd > thr2) best = 0;
    mergeto[s] = best;
}

// merge phase, step 1: count sources per target; relabel the source's pixels (doMerge :1107-1109)
__global__ __launch_bounds__(256) void k_merge_mark(
    uint32_t *__restrict__ seg, const uint32_t *__restrict__ mergeto,
    const uint32_t *__restrict__ pix, const uint32_t *__restrict__ off,
    const uint32_t *__restrict__ origsz, const uint32_t *__restrict__ chnext, uint32_t *tcount,
    uint32_t S)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u;
    if (s > S) return;
    const uint32_t t = mergeto[s];
    if (t == 0) return;
    atomicAdd(&tcount[t], 1u);
    for (uint32_t c = s; c != 0; c = chnext[c]) {
        const uint32_t o = off[c], m = origsz[c];
        for (uint32_t i = 0; i < m; i++) seg[pix[o + i]] = t;
    }
}

__global__ __launch_bounds__(256) void k_merge_fill(const uint32_t *__restrict__ mergeto,
                                                    const uint32_t *__restrict__ toff,
                                                    uint32_t *tfill, uint32_t *__restrict__ tlist,
                                                    uint32_t S)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u;
    if (s > S) return;
    const uint32_t t = mergeto[s];
    if (t == 0) return;
    const uint32_t slot = atomicAdd(&tfill[t], 1u);
    tlist[toff[t] + slot] = s;
}

__global__ __launch_bounds__(256) void k_merge_rank(const uint32_t *__restrict__ mergeto,
                                                    const uint32_t *__restrict__ toff,
                                                    const uint32_t *__restrict__ tcount,
                                                    const uint32_t *__restrict__ tlist,
                                                    uint32_t *__restrict__ tsorted, uint32_t S)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u;
    if (s > S) return;
    const uint32_t t = mergeto[s];
    if (t == 0) return;
    const uint32_t base = toff[t], cnt = tcount[t];
    uint32_t rank = 0;
    for (uint32_t i = 0; i < cnt; i++) rank += (tlist[base + i] < s) ? 1u : 0u;
    tsorted[base + rank] = s;
}

// merge phase, step 2: each target absorbs its sources in ascending id (doMerge :1112-1123)
__global__ __launch_bounds__(256) void k_merge_apply(
    uint32_t *segsz, float *ssum, uint32_t *chnext, uint32_t *chtail, uint32_t *mergeto,
    uint32_t *tcount, uint32_t *tfill, const uint32_t *__restrict__ toff,
    const uint32_t *__restrict__ tsorted, uint32_t *hist, uint32_t *nelim, uint32_t S,
    uint32_t target, int nb, uint32_t min_seg)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x + 1u;
    if (t > S) return;
    const uint32_t cnt = tcount[t];
    if (cnt == 0) return;
    const uint32_t base = toff[t];
    const uint32_t a0 = segsz[t];
    uint32_t sz = a0, tail = chtail[t];
    for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t s = tsorted[base + i];
        for (int b = 0; b < nb; b++) {
            ssum[(size_t)t * nb + b] = ssum[(size_t)t * nb + b] + ssum[(size_t)s * nb + b];
            ssum[(size_t)s * nb + b] = 0.0f;
        }
        sz += segsz[s];
        segsz[s] = 0;
        chnext[tail] = s;
        tail = chtail[s];
        mergeto[s] = 0;
    }
    segsz[t] = sz;
    chtail[t] = tail;
    tcount[t] = 0;
    tfill[t] = 0;
    atomicSub(&hist[target], cnt);
    if (a0 < min_seg) atomicSub(&hist[a0], 1u);
    if (sz < min_seg) atomicAdd(&hist[sz], 1u);
    atomicAdd(nelim, cnt);
}

static inline int bits_for(uint32_t maxval)
{
    int b = 1;
    while (b < 32 && (maxval >> b) != 0) b++;
    return b;
}

// d_seg in place; *max_id in: seg.max(); out: seg.max() after the final relabel.
static int run_eliminate_small(shp_ctx *ctx, const void *d_img, int dtype, int nb, uint32_t nrows,
                               uint32_t ncols, int four, int min_seg_size, double max_spectral_diff,
                               uint32_t *d_seg, uint32_t *max_id, int64_t *num_elim)
{
    const uint32_t n = nrows * ncols;
    const uint32_t S = *max_id;
    const size_t ns = (size_t)S + 2;
    const uint32_t min_seg = (uint32_t)(min_seg_size < 1 ? 1 : min_seg_size);
    *num_elim = 0;
    CHK(buf_ensure(ctx, ctx->segsz, ns * 4));
    CHK(buf_ensure(ctx, ctx->origsz, ns * 4));
    CHK(buf_ensure(ctx, ctx->off, ns * 4 + 16));
    CHK(buf_ensure(ctx, ctx->ssum, ns * nb * 4));
    CHK(buf_ensure(ctx, ctx->chnext, ns * 4));
    CHK(buf_ensure(ctx, ctx->chtail, ns * 4));
    CHK(buf_ensure(ctx, ctx->mergeto, ns * 4));
    CHK(buf_ensure(ctx, ctx->tcount, ns * 4));
    CHK(buf_ensure(ctx, ctx->toff, ns * 4 + 16));
    CHK(buf_ensure(ctx, ctx->tfill, ns * 4));
    CHK(buf_ensure(ctx, ctx->tlist, ns * 4));
    CHK(buf_ensure(ctx, ctx->tsorted, ns * 4));
    CHK(buf_ensure(ctx, ctx->small, ((size_t)min_seg + 8) * 4));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns > n ? ns : n)));
    uint32_t *segsz = bp<uint32_t>(ctx->segsz), *origsz = bp<uint32_t>(ctx->origsz);
    uint32_t *off = bp<uint32_t>(ctx->off), *chnext = bp<uint32_t>(ctx->chnext);
    uint32_t *chtail = bp<uint32_t>(ctx->chtail), *mergeto = bp<uint32_t>(ctx->mergeto);
    uint32_t *tcount = bp<uint32_t>(ctx->tcount), *toff = bp<uint32_t>(ctx->toff);
    uint32_t *tfill = bp<uint32_t>(ctx->tfill), *tlist = bp<uint32_t>(ctx->tlist);
    uint32_t *tsorted = bp<uint32_t>(ctx->tsorted);
    float *ssum = bp<float>(ctx->ssum);
    uint32_t *hist = bp<uint32_t>(ctx->small);          // [0..min_seg] then nelim
    uint32_t *nelim = hist + min_seg + 1;
    hipStream_t st = ctx->stream;

    CHK(run_seg_size(ctx, d_seg, n, S, segsz));
    if (n == 0 || S == 0) return 0;
    // CSR: pixels grouped by segment id, raster order inside (stable sort of (seg, index))
    uint32_t *ksorted = nullptr, *pix = nullptr;
    int ps = prof_begin(ctx, PROF_SORT);
    CHK(sort_pairs(ctx, d_seg, nullptr, n, bits_for(S), &ksorted, &pix));
    prof_end(ctx, ps);
    uint32_t *stmp = bp<uint32_t>(ctx->scan_tmp);      // (fetched after sort_pairs: it may regrow)
    ArrFn szf{segsz};
    CHK(scan_exclusive(ctx, szf, S + 1u, off, nullptr, stmp));
    const unsigned gs = grid_for((size_t)S + 1, 256);
    HIPCHK(ctx, hipMemsetAsync(hist, 0, ((size_t)min_seg + 2) * 4, st));
    hipLaunchKernelGGL(k_small_init, dim3(gs), dim3(256), 0, st, segsz, origsz, chnext, chtail,
                       mergeto, tcount, tfill, hist, S, min_seg); KCHK(ctx);
    ps = prof_begin(ctx, PROF_SPECTRA);
    hipLaunchKernelGGL(k_spectra_small, dim3(gs), dim3(256), 0, st, d_img, dtype, nb, n, pix, off,
                       segsz, ssum, S); KCHK(ctx);
    hipLaunchKernelGGL(k_spectra_big, dim3(grid_for((size_t)S * 64, 256)), dim3(256), 0, st, d_img,
                       dtype, nb, n, pix, off, segsz, ssum, S); KCHK(ctx);
    prof_end(ctx, ps);
    ps = prof_begin(ctx, PROF_SMALL_LOOP);

    const double thr2 = max_spectral_diff * max_spectral_diff;       // float64 square (N8)
    std::vector<uint32_t> hhist((size_t)min_seg + 2, 0);
    auto read_hist = [&]() -> int {
        HIPCHK(ctx, hipMemcpyAsync(hhist.data(), hist, ((size_t)min_seg + 2) * 4,
                                   hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return 0;
    };
    CHK(read_hist());
    for (uint32_t target = 1; target < min_seg; target++) {
        long long count = hhist[target], prev = -1;
        int passes = 0;
        while (count != prev && passes < 10) {
            prev = count;
            if (count > 0) {
                hipLaunchKernelGGL(k_find_merge, dim3(gs), dim3(256), 0, st, d_seg, segsz, ssum, pix,
                                   off, origsz, chnext, mergeto, S, target, nb, nrows, ncols, four,
                                   thr2); KCHK(ctx);
                hipLaunchKernelGGL(k_merge_mark, dim3(gs), dim3(256), 0, st, d_seg, mergeto, pix,
                                   off, origsz, chnext, tcount, S); KCHK(ctx);
                ArrFn tf{tcount};
                CHK(scan_exclusive(ctx, tf, S + 1u, toff, nullptr, stmp));
                hipLaunchKernelGGL(k_merge_fill, dim3(gs), dim3(256), 0, st, mergeto, toff, tfill,
                                   tlist, S); KCHK(ctx);
                hipLaunchKernelGGL(k_merge_rank, dim3(gs), dim3(256), 0, st, mergeto, toff, tcount,
                                   tlist, tsorted, S); KCHK(ctx);
                hipLaunchKernelGGL(k_merge_apply, dim3(gs), dim3(256), 0, st, segsz, ssum, chnext,
                                   chtail, mergeto, tcount, tfill, toff, tsorted, hist, nelim, S,
                                   target, nb, min_seg); KCHK(ctx);
                CHK(read_hist());
                count = hhist[target];
            }
            passes++;
        }
    }
    prof_end(ctx, ps);
    *num_elim = (int64_t)hhist[min_seg + 1];
    uint32_t new_max = 0;
    CHK(run_relabel(ctx, d_seg, n, segsz, S, &new_max));
    *max_id = new_max;
    return 0;
}
