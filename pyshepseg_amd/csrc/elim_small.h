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
    __shared__ uint32_t lh[256];            // block-local histogram of sizes 1..255
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s <= S) {
        const uint32_t m = segsz[s];
        origsz[s] = m;
        chnext[s] = 0;
        chtail[s] = s;
        mergeto[s] = 0;
        tcount[s] = 0;
        tfill[s] = 0;
        if (s >= 1u && m < min_seg) {
            if (m < 256u) atomicAdd(&lh[m], 1u);
            else atomicAdd(&hist[m], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < min_seg && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// Device-side loop control of eliminateSmallSegments (shepseg.py:970-997).  The host enqueues
// identical "pass slots" without reading anything back; the one-thread control kernel at the
// head of each slot advances (target, prev, passes) exactly like the reference's for/while and
// stops at the next pass that has sources to merge.  Every other kernel of the slot exits at
// once when ctl->active == 0.
struct SmallCtl {
    uint32_t target;     // current targetSize
    int32_t prev;        // prevCount (-1 = none)
    uint32_t passes;     // numPasses for this target
    uint32_t active;     // this slot runs a find/merge pass
    uint32_t done;       // target reached minSegSize
    uint32_t nelim;      // numElim
    uint32_t nsrc, ntgt, bump, pad;
};

__global__ void k_small_ctl(SmallCtl *ctl, const uint32_t *__restrict__ hist, uint32_t min_seg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ctl->nsrc = 0; ctl->ntgt = 0; ctl->bump = 0;
    uint32_t target = ctl->target, passes = ctl->passes;
    int32_t prev = ctl->prev;
    uint32_t active = 0, done = 0;
    for (;;) {
        if (target >= min_seg) { done = 1; break; }
        const int32_t count = (int32_t)hist[target];
        if (count != prev && passes < 10u) {          // `while` condition, shepseg.py:980
            prev = count;
            passes++;
            if (count > 0) { active = 1; break; }     // a pass with sources: run the kernels
        } else {
            target++; prev = -1; passes = 0;          // next targetSize, shepseg.py:970
        }
    }
    ctl->target = target; ctl->prev = prev; ctl->passes = passes;
    ctl->active = active; ctl->done = done;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// findMergeSegment (shepseg.py:1003-1063) for one source, executed by a whole wavefront: one
// lane per pixel of its list.  The reference keeps the FIRST strict minimum in (list index k,
// ii outer, jj inner) order (N7) == the lexicographic minimum of (distSqr, k, neighbour
// position), found with one 64-bit wave reduction (distSqr >= +0, so its float32 bit pattern
// orders like the value).
__device__ __forceinline__ void find_merge_wave(
    uint32_t s, const uint32_t *__restrict__ seg, const uint32_t *__restrict__ segsz,
    const float *__restrict__ ssum, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ origsz,
    const uint32_t *__restrict__ chnext, uint32_t *__restrict__ mergeto, uint32_t target, int nb,
    uint32_t nrows, uint32_t ncols, int four, double thr2)
{
    const unsigned lane = lane_id();
    const float nf = (float)target;
    unsigned long long best = ~0ull;         // (float bits of distSqr << 32) | order
    uint32_t bestnb = 0;
    uint32_t k0 = 0;                         // list index of the current chunk's first pixel
    for (uint32_t c = s; c != 0; c = (uint32_t)__builtin_amdgcn_readfirstlane((int)chnext[c])) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[c]);
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)origsz[c]);
        for (uint32_t i0 = 0; i0 < m; i0 += 64u) {
            const uint32_t i = i0 + lane;
            if (i < m) {
                const uint32_t k = k0 + i;
                const uint32_t p = pix[o + i];
                const uint32_t r = p / ncols, cc = p - r * ncols;
                uint32_t last = 0, pos = 0;
                for (int di = -1; di <= 1; di++)
                    for (int dj = -1; dj <= 1; dj++) {
                        if (di == 0 && dj == 0) continue;
                        if (four && di != 0 && dj != 0) continue;
                        const uint32_t mypos = pos++;
                        const int ii = (int)r + di, jj = (int)cc + dj;
                        if (ii < 0 || jj < 0 || ii >= (int)nrows || jj >= (int)ncols) continue;
                        const uint32_t nbid = seg[(uint32_t)ii * ncols + (uint32_t)jj];
                        if (nbid == s || nbid == 0 || nbid == last) continue;
                        last = nbid;
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
                            const unsigned long long key =
                                ((unsigned long long)__float_as_uint(d) << 32) |
                                (unsigned long long)(k * 8u + mypos);
                            if (key < best) { best = key; bestnb = nbid; }
                        }
                    }
            }
        }
        k0 += m;
    }
    const unsigned long long wmin = wave_min_u64(best);
    if (wmin == ~0ull) { if (lane == 0) mergeto[s] = 0; return; }
    if (best == wmin) {                      // unique: (k, position) differs between lanes
        const float bd = __uint_as_float((uint32_t)(wmin >> 32));
        mergeto[s] = ((double)bd > thr2) ? 0u : bestnb;
    }
}

// find phase: each block finds the sources (size == target) among its 256 segment ids, appends
// them to the global source list and lets its four wavefronts run findMergeSegment on them.
__global__ __launch_bounds__(256) void k_find_merge(
    const SmallCtl *ctlp, uint32_t *cnts, const uint32_t *__restrict__ seg,
    const uint32_t *__restrict__ segsz, const float *__restrict__ ssum,
    const uint32_t *__restrict__ pix, const uint32_t *__restrict__ off,
    const uint32_t *__restrict__ origsz, const uint32_t *__restrict__ chnext,
    uint32_t *__restrict__ mergeto, uint32_t *__restrict__ srclist, uint32_t S, int nb,
    uint32_t nrows, uint32_t ncols, int four, double thr2)
{
    __shared__ uint32_t lsrc[256];
    __shared__ uint32_t lcnt;
    if (!ctlp->active) return;
    const uint32_t target = ctlp->target;
    if (threadIdx.x == 0) lcnt = 0;
    __syncthreads();
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u;
    const bool is = s <= S && segsz[s] == target;
    const unsigned long long m = __ballot(is);
    if (m != 0ull) {
        uint32_t lbase = 0, gbase = 0;
        if (lane_id() == 0) {
            lbase = atomicAdd(&lcnt, (uint32_t)__popcll(m));
            gbase = atomicAdd(&cnts[0], (uint32_t)__popcll(m));
        }
        lbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)lbase);
        gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
        if (is) {
            const uint32_t r = (uint32_t)__popcll(m & lanemask_lt());
            lsrc[lbase + r] = s;
            srclist[gbase + r] = s;
        }
    }
    __syncthreads();
    const uint32_t n = lcnt;
    for (uint32_t i = threadIdx.x >> 6; i < n; i += 4u) {
        const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)lsrc[i]);
        find_merge_wave(src, seg, segsz, ssum, pix, off, origsz, chnext, mergeto, target, nb, nrows,
                        ncols, four, thr2);
    }
}

// merge phase, step 1 (per source): count sources per target, list the targets, relabel the
// source's pixels (doMerge :1107-1109)
__global__ __launch_bounds__(256) void k_merge_mark(
    const SmallCtl *ctlp, uint32_t *__restrict__ seg,
    const uint32_t *__restrict__ mergeto, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ origsz,
    const uint32_t *__restrict__ chnext, uint32_t *tcount, const uint32_t *__restrict__ srclist,
    uint32_t *cnts, uint32_t *__restrict__ tgtlist)
{
    if (!ctlp->active) return;
    const uint32_t nsrc = cnts[0];
    for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < nsrc; w += gridDim.x * 256u) {
        const uint32_t s = srclist[w];
        const uint32_t t = mergeto[s];
        if (t == 0) continue;
        if (atomicAdd(&tcount[t], 1u) == 0u) tgtlist[atomicAdd(&cnts[1], 1u)] = t;
        for (uint32_t c = s; c != 0; c = chnext[c]) {
            const uint32_t o = off[c], m = origsz[c];
            for (uint32_t i = 0; i < m; i++) seg[pix[o + i]] = t;
        }
    }
}

// storage for each target's source list (bump allocation; order is irrelevant)
__global__ __launch_bounds__(256) void k_merge_alloc(const SmallCtl *ctlp,
                                                     const uint32_t *__restrict__ tgtlist,
                                                     const uint32_t *__restrict__ tcount,
                                                     uint32_t *__restrict__ toff, uint32_t *cnts)
{
    if (!ctlp->active) return;
    const uint32_t ntgt = cnts[1];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < ntgt; i += gridDim.x * 256u) {
        const uint32_t t = tgtlist[i];
        toff[t] = atomicAdd(&cnts[2], tcount[t]);
    }
}

__global__ __launch_bounds__(256) void k_merge_fill(const SmallCtl *ctlp,
                                                    const uint32_t *__restrict__ mergeto,
                                                    const uint32_t *__restrict__ toff,
                                                    uint32_t *tfill, uint32_t *__restrict__ tlist,
                                                    const uint32_t *__restrict__ srclist,
                                                    const uint32_t *__restrict__ cnts)
{
    if (!ctlp->active) return;
    const uint32_t nsrc = cnts[0];
    for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < nsrc; w += gridDim.x * 256u) {
        const uint32_t s = srclist[w];
        const uint32_t t = mergeto[s];
        if (t == 0) continue;
        const uint32_t slot = atomicAdd(&tfill[t], 1u);
        tlist[toff[t] + slot] = s;
    }
}

__global__ __launch_bounds__(256) void k_merge_rank(const SmallCtl *ctlp,
                                                    const uint32_t *__restrict__ mergeto,
                                                    const uint32_t *__restrict__ toff,
                                                    const uint32_t *__restrict__ tcount,
                                                    const uint32_t *__restrict__ tlist,
                                                    uint32_t *__restrict__ tsorted,
                                                    const uint32_t *__restrict__ srclist,
                                                    const uint32_t *__restrict__ cnts)
{
    if (!ctlp->active) return;
    const uint32_t nsrc = cnts[0];
    for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < nsrc; w += gridDim.x * 256u) {
        const uint32_t s = srclist[w];
        const uint32_t t = mergeto[s];
        if (t == 0) continue;
        const uint32_t base = toff[t], cnt = tcount[t];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < cnt; i++) rank += (tlist[base + i] < s) ? 1u : 0u;
        tsorted[base + rank] = s;
    }
}

// merge phase, step 2: each target absorbs its sources in ascending id (doMerge :1112-1123)
__global__ __launch_bounds__(256) void k_merge_apply(
    SmallCtl *ctlp, uint32_t *segsz, float *ssum, uint32_t *chnext, uint32_t *chtail,
    uint32_t *mergeto, uint32_t *tcount, uint32_t *tfill, const uint32_t *__restrict__ toff,
    const uint32_t *__restrict__ tsorted, uint32_t *hist, const uint32_t *__restrict__ tgtlist,
    const uint32_t *__restrict__ cnts, int nb, uint32_t min_seg)
{
    if (!ctlp->active) return;
    const uint32_t target = ctlp->target;
    const uint32_t ntgt = cnts[1];
    for (uint32_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < ntgt; i0 += gridDim.x * 256u) {
        const uint32_t t = tgtlist[i0];
        const uint32_t cnt = tcount[t];
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
        atomicAdd(&ctlp->nelim, cnt);
    }
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
    CHK(buf_ensure(ctx, ctx->srclist, ns * 4));
    CHK(buf_ensure(ctx, ctx->tgtlist, ns * 4));
    CHK(buf_ensure(ctx, ctx->small, ((size_t)min_seg + 32) * 4));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns > n ? ns : n)));
    uint32_t *segsz = bp<uint32_t>(ctx->segsz), *origsz = bp<uint32_t>(ctx->origsz);
    uint32_t *off = bp<uint32_t>(ctx->off), *chnext = bp<uint32_t>(ctx->chnext);
    uint32_t *chtail = bp<uint32_t>(ctx->chtail), *mergeto = bp<uint32_t>(ctx->mergeto);
    uint32_t *tcount = bp<uint32_t>(ctx->tcount), *toff = bp<uint32_t>(ctx->toff);
    uint32_t *tfill = bp<uint32_t>(ctx->tfill), *tlist = bp<uint32_t>(ctx->tlist);
    uint32_t *tsorted = bp<uint32_t>(ctx->tsorted);
    uint32_t *srclist = bp<uint32_t>(ctx->srclist), *tgtlist = bp<uint32_t>(ctx->tgtlist);
    float *ssum = bp<float>(ctx->ssum);
    uint32_t *hist = bp<uint32_t>(ctx->small);          // [0..min_seg] then nelim
    SmallCtl *ctl = (SmallCtl *)(hist + ((min_seg + 4u + 3u) & ~3u));
    uint32_t *cnts = &ctl->nsrc;                        // [0]=#sources [1]=#targets [2]=bump
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
    // pass slots: control kernel + find + merge kernels, enqueued SLOTS_PER_SYNC at a time
    SmallCtl hctl;
    memset(&hctl, 0, sizeof(hctl));
    hctl.target = 1; hctl.prev = -1;
    HIPCHK(ctx, hipMemcpyAsync(ctl, &hctl, sizeof(hctl), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    const unsigned gfix = 128;                         // grid-stride kernels
    const int SLOTS_PER_SYNC = 8;
    SmallCtl *pin = (SmallCtl *)ctx->h_pinned;
    for (int guard = 0; guard < 100000; guard++) {
        for (int k = 0; k < SLOTS_PER_SYNC; k++) {
            hipLaunchKernelGGL(k_small_ctl, dim3(1), dim3(64), 0, st, ctl, hist, min_seg); KCHK(ctx);
            hipLaunchKernelGGL(k_find_merge, dim3(gs), dim3(256), 0, st, ctl, cnts, d_seg, segsz, ssum,
                               pix, off, origsz, chnext, mergeto, srclist, S, nb, nrows, ncols, four,
                               thr2); KCHK(ctx);
            hipLaunchKernelGGL(k_merge_mark, dim3(gfix), dim3(256), 0, st, ctl, d_seg, mergeto, pix, off,
                               origsz, chnext, tcount, srclist, cnts, tgtlist); KCHK(ctx);
            hipLaunchKernelGGL(k_merge_alloc, dim3(gfix), dim3(256), 0, st, ctl, tgtlist, tcount, toff,
                               cnts); KCHK(ctx);
            hipLaunchKernelGGL(k_merge_fill, dim3(gfix), dim3(256), 0, st, ctl, mergeto, toff, tfill,
                               tlist, srclist, cnts); KCHK(ctx);
            hipLaunchKernelGGL(k_merge_rank, dim3(gfix), dim3(256), 0, st, ctl, mergeto, toff, tcount,
                               tlist, tsorted, srclist, cnts); KCHK(ctx);
            hipLaunchKernelGGL(k_merge_apply, dim3(gfix), dim3(256), 0, st, ctl, segsz, ssum, chnext,
                               chtail, mergeto, tcount, tfill, toff, tsorted, hist, tgtlist, cnts, nb,
                               min_seg); KCHK(ctx);
        }
        HIPCHK(ctx, hipMemcpyAsync(pin, ctl, sizeof(SmallCtl), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (pin->done) break;
    }
    if (!pin->done) SHP_FAIL(ctx, SHP_ERR_STATE, "small-segment loop did not terminate");
    prof_end(ctx, ps);
    *num_elim = (int64_t)pin->nelim;
    uint32_t new_max = 0;
    CHK(run_relabel(ctx, d_seg, n, segsz, S, &new_max));
    *max_id = new_max;
    return 0;
}
