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

// nearest spectrally-similar neighbouring pixel in a segment of size > 1 (shepseg.py:677-736).
// The 3 x 3 window is walked in the reference's order (rows outer, columns inner; first strict
// minimum wins) but evaluated without branches around the loads: neighbour ids, then their
// segment sizes, then the band values of all nine positions are fetched as independent loads
// (an excluded position re-reads the centre pixel and is ignored), so one candidate costs three
// memory round trips instead of 3 + 2 * nBands.  DT = pixel type (see ld_t).
// COH: seg / segsz are being changed by earlier passes of the SAME kernel (k_single_tail): read them
// at the L2 instead of through the read-only path.
template <int DT, bool COH = false>
__device__ __forceinline__ uint32_t single_target(const void *__restrict__ img, int nb,
                                                  const uint32_t *seg, const uint32_t *segsz, uint32_t p,
                                                  uint32_t n, uint32_t nrows, uint32_t ncols, int four,
                                                  const ImgGeom g)
{
    const uint32_t i = p / ncols, j = p - i * ncols;
    uint32_t q[9], sn[9];
    size_t qo[9];                   // image offsets of the nine positions (see ImgGeom)
    bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int da = k / 3 - 1, db = k % 3 - 1;
        const int a = (int)i + da, b = (int)j + db;
        ok[k] = k != 4 && a >= 0 && b >= 0 && a < (int)nrows && b < (int)ncols && !(four && da != 0 && db != 0);
        q[k] = ok[k] ? (uint32_t)a * ncols + (uint32_t)b : p;
        qo[k] = g.origin + (size_t)(ok[k] ? (uint32_t)a : i) * g.pitch + (ok[k] ? (uint32_t)b : j);
        sn[k] = COH ? L2LOAD(&seg[q[k]]) : seg[q[k]];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) ok[k] = ok[k] && (COH ? L2LOAD(&segsz[sn[k]]) : segsz[sn[k]]) > 1u;
    // the reference's dSqr is an int64 that wraps (32-bit imagery can get there) and is compared
    // as written, `minDsqr < 0 or dSqr < minDsqr`: a negative sum counts as "unset" for the next
    // candidate.  Unsigned arithmetic, converted at the end: signed overflow is undefined.
    unsigned long long d[9];
#pragma unroll
    for (int k = 0; k < 9; k++) d[k] = 0;
    for (int b = 0; b < nb; b++) {
        const long long vp = ld_t<DT>(img, (size_t)b * g.bstride + qo[4]);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const unsigned long long t = (unsigned long long)(vp - ld_t<DT>(img, (size_t)b * g.bstride + qo[k]));
            d[k] += t * t;
        }
    }
    uint32_t out = NO_TARGET;
    long long mind = -1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const long long dk = (long long)d[k];
        if (ok[k] && (mind < 0 || dk < mind)) { mind = dk; out = sn[k]; }
    }
    return out;
}

// first pass: every pixel.  Single pixels that find no target yet are appended to `rest`
// (they are the only candidates of the later passes: sizes never shrink to 1).
template <int DT>
__global__ __launch_bounds__(256) void k_single_scan(
    const void *__restrict__ img, int nb, const uint32_t *__restrict__ seg,
    const uint32_t *__restrict__ segsz, uint32_t *__restrict__ tgt, uint32_t n, uint32_t nrows,
    uint32_t ncols, int four, uint32_t *__restrict__ rest, uint32_t *nrest, const ImgGeom geom)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    uint32_t out = NO_TARGET;
    bool keep = false;
    if (p < n) {
        if (segsz[seg[p]] == 1u) {
            out = single_target<DT>(img, nb, seg, segsz, p, n, nrows, ncols, four, geom);
            keep = out == NO_TARGET;
        }
        tgt[p] = out;
    }
    const unsigned long long m = __ballot(keep);
    if (m != 0ull) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(nrest, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (keep) rest[base + (uint32_t)__popcll(m & lanemask_lt())] = p;
    }
}

__global__ __launch_bounds__(256) void k_single_apply(uint32_t *__restrict__ seg, uint32_t *segsz,
                                                      const uint32_t *__restrict__ tgt, uint32_t n,
                                                      uint32_t *nelim)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = (p < n) ? tgt[p] : NO_TARGET;
    if (t != NO_TARGET) {
        const uint32_t old = seg[p];
        seg[p] = t;
        segsz[old] = 0;
        atomicAdd(&segsz[t], 1u);
        // the host only needs "did this pass merge anything" (the count is oldMax - newMax later)
        *nelim = 1u;
    }
}

// later passes: only the remaining single pixels
template <int DT>
__global__ __launch_bounds__(256) void k_single_scan_list(
    const void *__restrict__ img, int nb, const uint32_t *__restrict__ seg,
    const uint32_t *__restrict__ segsz, uint32_t *__restrict__ tgt_l, uint32_t n, uint32_t nrows,
    uint32_t ncols, int four, const uint32_t *__restrict__ rest, uint32_t nrest, uint32_t *nelim,
    const ImgGeom geom)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) { nelim[0] = 0u; nelim[1] = 0u; }     // the apply kernel of this pass raises them (no memset launch)
    if (i >= nrest) return;
    const uint32_t p = rest[i];
    uint32_t out = NO_TARGET;
    if (segsz[seg[p]] == 1u) out = single_target<DT>(img, nb, seg, segsz, p, n, nrows, ncols, four, geom);
    tgt_l[i] = out;
}

// rest2 (optional): the single pixels of the list that found no target are appended there, their
// count in nelim[1] -- the candidates of the later passes.  A workgroup covers SINGLE_SPAN entries
// and appends with ONE global atomic (atomics on one counter serialise at the L2: ~0.1 us each).
#define SINGLE_SPAN 4096u
__global__ __launch_bounds__(256) void k_single_apply_list(uint32_t *__restrict__ seg, uint32_t *segsz,
                                                           const uint32_t *__restrict__ tgt_l,
                                                           const uint32_t *__restrict__ rest,
                                                           uint32_t nrest,
                                                           uint32_t *nelim, uint32_t *__restrict__ rest2)
{
    __shared__ uint32_t s_buf[SINGLE_SPAN];
    __shared__ uint32_t s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0u;
    __syncthreads();
    bool any = false;
    for (uint32_t it = 0; it < SINGLE_SPAN / 256u; it++) {
        const uint32_t i = blockIdx.x * SINGLE_SPAN + it * 256u + threadIdx.x;
        const uint32_t t = i < nrest ? tgt_l[i] : NO_TARGET;
        const uint32_t p = i < nrest ? rest[i] : 0u;
        if (t != NO_TARGET) {
            const uint32_t old = seg[p];
            seg[p] = t;
            segsz[old] = 0;
            atomicAdd(&segsz[t], 1u);
            any = true;
        }
        if (rest2) {
            const bool keep = i < nrest && t == NO_TARGET;
            const unsigned long long m = __ballot(keep);
            if (m != 0ull) {
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (keep) s_buf[base + (uint32_t)__popcll(m & lanemask_lt())] = p;
            }
        }
    }
    if (__ballot(any) != 0ull && lane_id() == 0) *nelim = 1u;
    if (!rest2) return;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(&nelim[1], s_cnt) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < s_cnt; i += 256u) rest2[s_base + i] = s_buf[i];
}

// Every later pass of the single-pixel stage in ONE workgroup: the candidates left after the first
// pass are few (a single pixel stays without a target only while all its neighbours are single
// pixels too), so the scan / apply / "did anything merge" round trip of a pass -- three commands and
// a host synchronisation each -- becomes two workgroup barriers.  Same Jacobi passes as the
// reference's loop (shepseg.py:572-611): all scans of a pass read the state its applies have not
// touched yet.
template <int DT>
__global__ __launch_bounds__(1024) void k_single_tail(
    const void *__restrict__ img, int nb, uint32_t *seg, uint32_t *segsz, uint32_t *tgt_l, uint32_t n,
    uint32_t nrows, uint32_t ncols, int four, uint32_t *rest2, const uint32_t *__restrict__ nrest2,
    const ImgGeom geom)
{
    __shared__ uint32_t s_merged, s_next;
    uint32_t nr = *nrest2;
    // The list shrinks from pass to pass: what merged, or stopped being a single pixel because a neighbour
    // merged INTO it, never comes back, so only the candidates that are still single and found no target yet
    // are carried on (compacted into the other half of the list's buffer; their order is immaterial: a pass
    // reads the state before any of its applies and the applies commute).  Without it every pass walked all
    // nr entries, five rounds per thread for the one or two hundred that were still alive.
    const bool compact = 2ull * nr <= (unsigned long long)n;
    uint32_t *cur = rest2, *nxt = rest2 + nr;
    for (;;) {
        if (threadIdx.x == 0) { s_merged = 0u; s_next = 0u; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nr; i += 1024u) {
            const uint32_t p = cur[i];
            uint32_t out = NO_TARGET;
            bool single = L2LOAD(&segsz[L2LOAD(&seg[p])]) == 1u;
            if (single)
                out = single_target<DT, true>(img, nb, seg, segsz, p, n, nrows, ncols, four, geom);
            tgt_l[i] = out;                       // (read back by this thread only)
            if (compact && single && out == NO_TARGET) nxt[atomicAdd(&s_next, 1u)] = p;
        }
        __threadfence();
        __syncthreads();
        bool any = false;
        for (uint32_t i = threadIdx.x; i < nr; i += 1024u) {
            const uint32_t t = tgt_l[i];
            if (t == NO_TARGET) continue;
            const uint32_t p = cur[i];
            const uint32_t old = L2LOAD(&seg[p]);
            __hip_atomic_store(&seg[p], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&segsz[old], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&segsz[t], 1u);
            any = true;
        }
        if (any) s_merged = 1u;
        __threadfence();
        __syncthreads();
        const bool more = s_merged != 0u;
        const uint32_t left = s_next;
        __syncthreads();
        if (!more) break;
        if (compact) {
            if (left == 0u) break;                // nobody left to look for a target
            nr = left;
            uint32_t *t = cur; cur = nxt; nxt = t;
        }
    }
}

// relabelSegments (shepseg.py:739-777): newid[k] = k - #{1 <= j < k : segsz[j] == 0}
struct EmptyFn {      // f(i) = 1 if id i (>= 1) is unused; f(0) = 0
    const uint32_t *segsz;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const
    {
        return (i >= 1u && segsz[i] == 0u) ? 1u : 0u;
    }
    __device__ __forceinline__ bool get4(uint32_t base, uint32_t v[4]) const
    {
        if (!scan_load4(segsz, base, v)) return false;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) v[i] = (base + i >= 1u && v[i] == 0u) ? 1u : 0u;
        return true;
    }
};

// sizes of the surviving ids under their new numbers (the reference recomputes makeSegSize)
// (boff: the block offsets of the scan that made `sub`, added here instead of by a launch)
__global__ __launch_bounds__(256) void k_compact_sizes(const uint32_t *__restrict__ segsz,
                                                       const uint32_t *__restrict__ sub, uint32_t ns,
                                                       uint32_t *__restrict__ out,
                                                       const uint32_t *__restrict__ boff)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= ns) return;
    const uint32_t v = segsz[k];
    if (k == 0u) out[0] = v;
    else if (v != 0u) out[k - sub[k] - (boff ? boff[k / SCAN_ITEMS] : 0u)] = v;
}

// (sizes_out: the k_compact_sizes pass over the ns ids rides in the same launch)
__global__ __launch_bounds__(256) void k_relabel(uint32_t *__restrict__ seg,
                                                 const uint32_t *__restrict__ sub, uint32_t n,
                                                 const uint32_t *__restrict__ boff,
                                                 const uint32_t *__restrict__ segsz, uint32_t ns,
                                                 uint32_t *__restrict__ sizes_out)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (sizes_out && p < ns) {
        const uint32_t v = segsz[p];
        if (p == 0u) sizes_out[0] = v;
        else if (v != 0u) sizes_out[p - sub[p] - (boff ? boff[p / SCAN_ITEMS] : 0u)] = v;
    }
    if (p >= n) return;
    const uint32_t s = seg[p];
    seg[p] = s - sub[s] - (boff ? boff[s / SCAN_ITEMS] : 0u);
}

// the same, four pixels per thread as one 16-byte load and store (seg 16-byte aligned; the tail by the last threads)
__global__ __launch_bounds__(256) void k_relabel4(uint32_t *__restrict__ seg,
                                                  const uint32_t *__restrict__ sub, uint32_t n,
                                                  const uint32_t *__restrict__ boff,
                                                  const uint32_t *__restrict__ segsz, uint32_t ns,
                                                  uint32_t *__restrict__ sizes_out)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (sizes_out && t < ns) {
        const uint32_t v = segsz[t];
        if (t == 0u) sizes_out[0] = v;
        else if (v != 0u) sizes_out[t - sub[t] - (boff ? boff[t / SCAN_ITEMS] : 0u)] = v;
    }
    const uint32_t p = t * 4u;
    if (p + 4u <= n) {
        uint4 s = *(const uint4 *)(seg + p);
        s.x = s.x - sub[s.x] - (boff ? boff[s.x / SCAN_ITEMS] : 0u);
        s.y = s.y - sub[s.y] - (boff ? boff[s.y / SCAN_ITEMS] : 0u);
        s.z = s.z - sub[s.z] - (boff ? boff[s.z / SCAN_ITEMS] : 0u);
        s.w = s.w - sub[s.w] - (boff ? boff[s.w / SCAN_ITEMS] : 0u);
        *(uint4 *)(seg + p) = s;
    } else {
        for (uint32_t q = p; q < n; q++) {
            const uint32_t v = seg[q];
            seg[q] = v - sub[v] - (boff ? boff[v / SCAN_ITEMS] : 0u);
        }
    }
}

// Compacts ids in d_seg given segsz[0..max_id].  *new_max_host = max_id - (#unused ids >= 1)
// which equals seg.max() after the relabel (0 when every pixel is null).
static int run_relabel(shp_ctx *ctx, uint32_t *d_seg, uint32_t n, const uint32_t *d_segsz,
                       uint32_t max_id, uint32_t *new_max_host, uint32_t *d_sizes_out = nullptr)
{
    const uint32_t ns = max_id + 1u;
    CHK(buf_ensure(ctx, ctx->toff, (size_t)ns * 4 + 16));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns)));
    uint32_t *sub = bp<uint32_t>(ctx->toff);
    uint32_t *tot = sub + ns;
    EmptyFn f{d_segsz};
    const uint32_t *boff = nullptr;
    uint32_t *mir = ctx->h_pinned + PIN_MIRROR + MIR_RELABEL;        // the scan stores its total there
    CHK(scan_exclusive(ctx, f, ns, sub, tot, bp<uint32_t>(ctx->scan_tmp), &boff, mir));
    if (n || d_sizes_out) {
        if (((uintptr_t)d_seg & 15u) == 0u) {
            const uint32_t n4 = (n + 3u) / 4u;
            hipLaunchKernelGGL(k_relabel4, dim3(grid_for(n4 > ns ? n4 : ns, 256)), dim3(256), 0, ctx->stream, d_seg,
                               sub, n, boff, d_segsz, ns, d_sizes_out);
        } else
            hipLaunchKernelGGL(k_relabel, dim3(grid_for(n > ns ? n : ns, 256)), dim3(256), 0, ctx->stream, d_seg,
                               sub, n, boff, d_segsz, ns, d_sizes_out);
        KCHK(ctx);
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *new_max_host = max_id - *(volatile uint32_t *)mir;
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
// sizes_ready: ctx->segsz already holds makeSegSize(d_seg) (run_clump provides it).
// On return ctx->origsz holds the segment sizes under the NEW ids (max_id+1 entries).
// singles_ready: ctx->singles holds nsingles one-pixel clumps (run_clump provides them) and
// they are the only size-1 segments (the caller checked that the null count is not 1).
static int run_eliminate_single(shp_ctx *ctx, const void *d_img, int dtype, int nb, uint32_t nrows,
                                uint32_t ncols, int four, uint32_t *d_seg, uint32_t *max_id,
                                int sizes_ready = 0, int singles_ready = 0, uint32_t nsingles = 0,
                                const ImgGeom *geom_in = nullptr)
{
    const uint32_t n = nrows * ncols;
    const ImgGeom geom = geom_in ? *geom_in : geom_compact(n, ncols);
    CHK(buf_ensure(ctx, ctx->segsz, ((size_t)*max_id + 2) * 4));
    CHK(buf_ensure(ctx, ctx->origsz, ((size_t)*max_id + 2) * 4));
    CHK(buf_ensure(ctx, ctx->aux, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->stack, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->small, 4096));
    uint32_t *segsz = bp<uint32_t>(ctx->segsz), *tgt = bp<uint32_t>(ctx->aux);
    uint32_t *rest = bp<uint32_t>(ctx->stack);
    uint32_t *nelim = bp<uint32_t>(ctx->small), *nrest = nelim + 1;
    if (!sizes_ready) CHK(run_seg_size(ctx, d_seg, n, *max_id, segsz));
    if (n == 0) return 0;
    const unsigned g = grid_for(n, 256);
    hipStream_t st = ctx->stream;
    uint32_t merged = 0, nr = 0;
    if (singles_ready) {
        // the candidates are known: the first pass walks the list and compacts what is left of it,
        // one workgroup runs the later passes -- three commands, no host round trip
        if (nsingles) {
            const uint32_t *list = bp<uint32_t>(ctx->singles);
            const unsigned gl = grid_for(nsingles, 256);
            DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_single_scan_list<DT>, dim3(gl), dim3(256), 0, st, d_img, nb,
                                                     d_seg, segsz, tgt, n, nrows, ncols, four, list, nsingles, nelim, geom));
            KCHK(ctx);
            hipLaunchKernelGGL(k_single_apply_list, dim3(grid_for(nsingles, SINGLE_SPAN)), dim3(256), 0, st, d_seg,
                               segsz, tgt, list, nsingles, nelim, rest); KCHK(ctx);
            DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_single_tail<DT>, dim3(1), dim3(1024), 0, st, d_img, nb, d_seg,
                                                     segsz, tgt, n, nrows, ncols, four, rest, nrest, geom));
            KCHK(ctx);
        }
    } else {
        HIPCHK(ctx, hipMemsetAsync(nelim, 0, 8, st));
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_single_scan<DT>, dim3(g), dim3(256), 0, st, d_img, nb, d_seg,
                                                 segsz, tgt, n, nrows, ncols, four, rest, nrest, geom));
        KCHK(ctx);
        hipLaunchKernelGGL(k_single_apply, dim3(g), dim3(256), 0, st, d_seg, segsz, tgt, n, nelim); KCHK(ctx);
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, nelim, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        merged = ctx->h_pinned[0];
        nr = ctx->h_pinned[1];
    }
    // later passes touch only the single pixels that could not merge in the first one
    while (merged != 0 && nr != 0) {
        const unsigned gl = grid_for(nr, 256);
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_single_scan_list<DT>, dim3(gl), dim3(256), 0, st, d_img, nb,
                                                 d_seg, segsz, tgt, n, nrows, ncols, four, rest, nr, nelim, geom));
        KCHK(ctx);
        hipLaunchKernelGGL(k_single_apply_list, dim3(grid_for(nr, SINGLE_SPAN)), dim3(256), 0, st, d_seg, segsz,
                           tgt, rest, nr, nelim, (uint32_t *)nullptr); KCHK(ctx);
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, nelim, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        merged = ctx->h_pinned[0];
    }
    uint32_t new_max = 0;
    CHK(run_relabel(ctx, d_seg, n, segsz, *max_id, &new_max, bp<uint32_t>(ctx->origsz)));
    *max_id = new_max;
    return 0;
}
