// kmeans.h -- k-means assign (predict) and Lloyd fit on the device.
//
// assign replaces shepseg.applySpectralClusters (shepseg.py:317-361) + sklearn
// KMeans.predict:  label = argmin_j ( |c_j|^2 - 2 x.c_j ), float64, first minimum wins,
// evaluated in the order of sklearn 0.24.2's chunked dgemm onto the pre-filled squared norms
// (oracle/shepseg_oracle.c orc_dist / orc_sqnorm, pinned by oracle/refgen/fit_probe.py): the
// dot product accumulated from zero by fma in band order with the operand -2c, added to |c|^2
// (one band: fma(x, -2c, |c|^2)); |c|^2 in numpy einsum's two-lane order.  SURVEY N13: any
// float64 evaluation reproduces sklearn on integer imagery with separated centres, float32 does
// not; on exact ties only this order does.
//
// Roofline: 2*nB*k flop/px in FP64 (720 at 6x60).  gfx950 FP64 vector and matrix peaks are
// equal (78.6 TFLOP/s), so the contraction runs on v_fma_f64 with the centroid operand in
// SGPRs (uniform scalar loads); bytes are nB*sizeof(px) in + 2 B (uint16 label) out.
#pragma once
#include "common.h"
#include <algorithm>
#include <chrono>
#include <thread>

#ifndef ASSIGN_PPT
#define ASSIGN_PPT 4   // pixels per thread (amortises the scalar centroid loads)
#endif

// |c|^2 as numpy 1.26 evaluates row_norms(C, squared=True) = einsum('ij,ij->i') on x86-64 (baseline
// SSE2, no fma): two lanes over the even / odd elements, products and sums rounded separately,
// whole blocks of 8 elements taken vector 3, 2, 1, 0, the rest in order; lanes added at the end.
__host__ __device__ inline double kmeans_sqnorm(const double *c, int nb)
{
    double a0 = 0.0, a1 = 0.0;
    int i = 0, count = nb;
    for (; count >= 8; count -= 8, i += 8) {
        double t0, t1, p;
        p = c[i + 6] * c[i + 6]; t0 = p + a0; p = c[i + 7] * c[i + 7]; t1 = p + a1;
        p = c[i + 4] * c[i + 4]; t0 = p + t0; p = c[i + 5] * c[i + 5]; t1 = p + t1;
        p = c[i + 2] * c[i + 2]; t0 = p + t0; p = c[i + 3] * c[i + 3]; t1 = p + t1;
        p = c[i + 0] * c[i + 0]; a0 = p + t0; p = c[i + 1] * c[i + 1]; a1 = p + t1;
    }
    for (; count > 0; count -= 2, i += 2) {
        const double p = c[i] * c[i];
        const double q = (count > 1) ? c[i + 1] * c[i + 1] : 0.0;
        a0 = p + a0;
        a1 = q + a1;
    }
    return a0 + a1;
}

// RECT: the pixels are a rectangle of a band-planar raster (x0, y0, w, h; `pitch` pixels per raster
// row; `npix` = pixels per band of the whole raster) and the clusters go to the same position of a
// cluster map with the raster's geometry -- the tiled driver assigns every pixel once, not once
// per overlapping tile (the model is global, so a pixel's cluster does not depend on the tile).
// Launched with a 2-D grid: x = columns / 256, y = rows / ASSIGN_PPT.
struct AssignRect { uint32_t x0, y0, w, h, pitch; };

// min of two distances (never NaN): the bare instruction -- __builtin_fmin would first canonicalise
// both operands (a v_max_f64 each), a tenth of the loop's instructions
__device__ __forceinline__ double assign_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int NB, int DT, bool RECT>
__global__ __launch_bounds__(256) void k_assign(
    const void *__restrict__ img, size_t npix, int nb_rt,
    const double *__restrict__ m2c, const double *__restrict__ cnorm, int k, int has_null,
    long long null_val, uint16_t *__restrict__ clus16, int32_t *__restrict__ clus32, AssignRect rc)
{
    const int nb = (NB > 0) ? NB : nb_rt;
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t p0 = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (!RECT && p0 >= npix) return;
    for (;;) {
        double x[ASSIGN_PPT][(NB > 0) ? NB : 1];
        bool isnull[ASSIGN_PPT], ok[ASSIGN_PPT];
        size_t off[ASSIGN_PPT];          // clamped inside the data: the loads need no branch
        double bestd[ASSIGN_PPT];
        int best[ASSIGN_PPT];
#pragma unroll
        for (int q = 0; q < ASSIGN_PPT; q++) {
            if (RECT) {
                const uint32_t col = blockIdx.x * 256u + threadIdx.x, row = blockIdx.y * ASSIGN_PPT + q;
                ok[q] = col < rc.w && row < rc.h;
                const uint32_t cc = col < rc.w ? col : rc.w - 1u, rr = row < rc.h ? row : rc.h - 1u;
                off[q] = (size_t)(rc.y0 + rr) * rc.pitch + rc.x0 + cc;
            } else {
                const size_t p = p0 + (size_t)q * stride;
                ok[q] = p < npix;
                off[q] = ok[q] ? p : npix - 1;
            }
            isnull[q] = false;
            best[q] = 0;
            bestd[q] = 0.0;
            if (NB > 0) {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const long long v = ld_t<DT>(img, (size_t)b * npix + off[q]);
                    if (has_null && v == null_val) isnull[q] = true;
                    x[q][b] = (double)v;
                }
            } else {
                for (int b = 0; b < nb; b++) {
                    const long long v = ld_t<DT>(img, (size_t)b * npix + off[q]);
                    if (has_null && v == null_val) isnull[q] = true;
                }
            }
        }
        for (int j = 0; j < k; j++) {
            double d[ASSIGN_PPT];
            const double cn = cnorm[j];
            const bool seeded = nb == 1;             // one band: fma(x, -2c, |c|^2); else dot from zero
#pragma unroll
            for (int q = 0; q < ASSIGN_PPT; q++) d[q] = seeded ? cn : 0.0;
            if (NB > 0) {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const double c = m2c[j * NB + b];
#pragma unroll
                    for (int q = 0; q < ASSIGN_PPT; q++) d[q] = __builtin_fma(x[q][b], c, d[q]);
                }
            } else {
                for (int b = 0; b < nb; b++) {
                    const double c = m2c[j * nb + b];
#pragma unroll
                    for (int q = 0; q < ASSIGN_PPT; q++) {
                        const double xv = (double)ld_t<DT>(img, (size_t)b * npix + off[q]);
                        d[q] = __builtin_fma(xv, c, d[q]);
                    }
                }
            }
            if (!seeded) {
#pragma unroll
                for (int q = 0; q < ASSIGN_PPT; q++) d[q] = cn + d[q];
            }
            // first minimum wins: the index moves only on a strict improvement, the value is a min
#pragma unroll
            for (int q = 0; q < ASSIGN_PPT; q++) {
                if (j == 0) { bestd[q] = d[q]; best[q] = 0; }
                else {
                    best[q] = d[q] < bestd[q] ? j : best[q];
                    bestd[q] = assign_min(d[q], bestd[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ASSIGN_PPT; q++) {
            if (ok[q]) {
                const int c = isnull[q] ? 0 : best[q] + 1;
                if (clus16) clus16[off[q]] = (uint16_t)c;
                if (clus32) clus32[off[q]] = c;
            }
        }
        if (RECT) break;
        p0 += stride * ASSIGN_PPT;
        if (p0 >= npix) break;
    }
}

// host-side preparation, same arithmetic as oracle orc_kmeans_prepare
static inline void kmeans_prepare_host(const double *centres, int k, int nb, double *m2c,
                                       double *cnorm)
{
    for (int j = 0; j < k; j++) {
        for (int b = 0; b < nb; b++) m2c[j * nb + b] = -2.0 * centres[j * nb + b];
        cnorm[j] = kmeans_sqnorm(centres + (size_t)j * nb, nb);
    }
}

// centres (host) -> ctx->cen (device: m2c[k*nb] then cnorm[k]); launches the assign kernel
// on the device image `d_img`; writes uint16 labels to d_clus16 and/or int32 to d_clus32.
// rects != nullptr: d_img is a whole raster of npix pixels per band and `pitch` pixels per row,
// the nrects rectangles (x, y, xs, ys each) are assigned into the cluster map d_clus16.
static int launch_assign(shp_ctx *ctx, const void *d_img, int dtype, int nb, size_t npix,
                         const double *centres, int k, int has_null, int64_t null_val,
                         uint16_t *d_clus16, int32_t *d_clus32, const int32_t *rects = nullptr,
                         int nrects = 0, uint32_t pitch = 0)
{
    if (k < 1 || k > 65534) SHP_FAIL(ctx, SHP_ERR_ARG, "numClusters %d out of range", k);
    const size_t hn = (size_t)k * nb + k;
    if (hn * 8 + 64 + 512 > SHP_PINNED_BYTES) SHP_FAIL(ctx, SHP_ERR_ARG, "k * nbands too large (%d x %d)", k, nb);
    double *h = (double *)(ctx->h_pinned + 16);       // pinned staging (stream-ordered reuse)
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // earlier users of the staging area are done
    kmeans_prepare_host(centres, k, nb, h, h + (size_t)k * nb);
    CHK(buf_ensure(ctx, ctx->cen, hn * sizeof(double) * 2));
    HIPCHK(ctx, hipMemcpyAsync(ctx->cen.p, h, hn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const double *m2c = bp<double>(ctx->cen), *cn = m2c + (size_t)k * nb;
    const unsigned grid = grid_for((npix + ASSIGN_PPT - 1) / ASSIGN_PPT, 256, 256u * 16u);
#define LA(NBT)                                                                                  \
    do {                                                                                         \
        if (!rects) {                                                                            \
            DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assign<NBT, DT, false>),  \
                           dim3(grid), dim3(256), 0, ctx->stream, d_img, npix, nb, m2c, cn, k,   \
                           has_null, (long long)null_val, d_clus16, d_clus32, AssignRect{}));    \
        } else {                                                                                 \
            for (int r = 0; r < nrects; r++) {                                                   \
                const AssignRect rc{(uint32_t)rects[4 * r], (uint32_t)rects[4 * r + 1],          \
                                    (uint32_t)rects[4 * r + 2], (uint32_t)rects[4 * r + 3], pitch}; \
                if (rc.w == 0u || rc.h == 0u) continue;                                          \
                const dim3 g2(grid_for(rc.w, 256), grid_for(rc.h, ASSIGN_PPT));                  \
                DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assign<NBT, DT, true>), \
                               g2, dim3(256), 0, ctx->stream, d_img, npix, nb, m2c, cn, k,       \
                               has_null, (long long)null_val, d_clus16, d_clus32, rc));          \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    const int ps = prof_begin(ctx, PROF_ASSIGN);
    switch (nb) {
    case 1: LA(1); break;
    case 2: LA(2); break;
    case 3: LA(3); break;
    case 4: LA(4); break;
    case 5: LA(5); break;
    case 6: LA(6); break;
    case 7: LA(7); break;
    case 8: LA(8); break;
    case 10: LA(10); break;
    case 12: LA(12); break;
    default: LA(0); break;
    }
#undef LA
    prof_end(ctx, ps);
    KCHK(ctx);
    return 0;
}

// ---------------------------------------------------------------------------------------
// Lloyd fit on the device.  Replaces sklearn KMeans(init=<array>, n_init=1).fit as called by
// shepseg.fitSpectralClusters (shepseg.py:305-312); algorithm restated in SURVEY Appendix D
// and oracle/shepseg_oracle.c orc_kmeans_fit (sklearn 0.24.2 semantics: centred data,
// tol = mean(var)*tol_rel, strict-convergence test, empty-cluster relocation, centres *= 1/w).
// E-step: one thread per sample row, same fma chain as predict.  M-step: the reference's sums --
// one chain per (cluster, band) over the cluster's rows in row order (fit_elkan.h:
// k_fit_sum_lists[_staged] on the row numbers sorted stably by label), so the centres are the
// reference's bits (one OpenMP thread) whenever the partitions are.  No float atomics.
// ---------------------------------------------------------------------------------------
#define FIT_BATCH 8                 // Lloyd iterations enqueued between two host synchronisations

// loop control of the Lloyd iterations, owned by the device between host synchronisations
struct FitCtl {
    uint32_t stop;      // 0 running; 1 labels unchanged (strict convergence); 2 an empty cluster:
                        // the host finishes this iteration; 3 centre shift <= tol
    uint32_t iters;     // completed iterations
    uint32_t ndiff;     // labels changed by the E-step of the running iteration
    uint32_t near;      // E-steps that met a sample with two centres within FIT_TIE_EPS (fit_elkan.h)
    double shift;
};

#include "fit_elkan.h"

#define FIT_RPT 4        // sample rows per thread in the E-step (independent fma chains in flight)
template <int NB>
__global__ __launch_bounds__(256) void k_fit_assign(const double *__restrict__ X, uint32_t n, int nb_rt,
                                                    const double *__restrict__ m2c,
                                                    const double *__restrict__ cnorm, int k,
                                                    int32_t *__restrict__ lab,
                                                    const int32_t *__restrict__ lab_old,
                                                    FitCtl *ctl)
{
    if (ctl && ctl->stop) return;               // the loop has ended: the rest of the batch is a no-op
    const int nb = (NB > 0) ? NB : nb_rt;
    const uint32_t i0 = blockIdx.x * (256u * FIT_RPT) + threadIdx.x;
    uint32_t ndiff = 0, near = 0;
    if (NB > 0) {
        double x[FIT_RPT][(NB > 0) ? NB : 1], bestd[FIT_RPT], second[FIT_RPT];
        int best[FIT_RPT];
        double cnmax = 0.0;
#pragma unroll
        for (int r = 0; r < FIT_RPT; r++) {
            const uint32_t i = i0 + (uint32_t)r * 256u;
            const uint32_t ic = i < n ? i : n - 1u;      // clamped: branch-free loads
#pragma unroll
            for (int b = 0; b < NB; b++) x[r][b] = X[(size_t)ic * NB + b];
            best[r] = 0; bestd[r] = 0.0; second[r] = __builtin_inf();
        }
        for (int j = 0; j < k; j++) {
            const double cn = cnorm[j];
            cnmax = cn > cnmax ? cn : cnmax;
            double d[FIT_RPT];
#pragma unroll
            for (int r = 0; r < FIT_RPT; r++) d[r] = (NB == 1) ? cn : 0.0;       // (see k_assign)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const double c = m2c[j * NB + b];
#pragma unroll
                for (int r = 0; r < FIT_RPT; r++) d[r] = __builtin_fma(x[r][b], c, d[r]);
            }
            if (NB != 1) {
#pragma unroll
                for (int r = 0; r < FIT_RPT; r++) d[r] = cn + d[r];
            }
#pragma unroll
            for (int r = 0; r < FIT_RPT; r++) {
                if (j == 0) { bestd[r] = d[r]; best[r] = 0; }
                else {
                    // (the runner-up, for the tie guard: the larger of d and the best so far, if smaller)
                    const double loser = d[r] < bestd[r] ? bestd[r] : d[r];
                    second[r] = assign_min(loser, second[r]);
                    best[r] = d[r] < bestd[r] ? j : best[r];
                    bestd[r] = assign_min(d[r], bestd[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FIT_RPT; r++) {
            const uint32_t i = i0 + (uint32_t)r * 256u;
            if (i < n) {
                lab[i] = best[r];
                if (lab_old && lab_old[i] != best[r]) ndiff++;
                double xn = 0.0;
#pragma unroll
                for (int b = 0; b < NB; b++) xn = __builtin_fma(x[r][b], x[r][b], xn);
                const double scale = __builtin_fabs(bestd[r]) + __builtin_fabs(second[r]) + xn + cnmax;
                if (k > 1 && second[r] - bestd[r] <= FIT_TIE_EPS * scale) near++;
            }
        }
    } else {
        for (int r = 0; r < FIT_RPT; r++) {
            const uint32_t i = i0 + (uint32_t)r * 256u;
            if (i >= n) continue;
            int best = 0;
            double bestd = 0.0, second = __builtin_inf(), cnmax = 0.0, xn = 0.0;
            for (int j = 0; j < k; j++) {
                double d = nb == 1 ? cnorm[j] : 0.0;
                for (int b = 0; b < nb; b++) d = __builtin_fma(X[(size_t)i * nb + b], m2c[j * nb + b], d);
                if (nb != 1) d = cnorm[j] + d;
                cnmax = cnorm[j] > cnmax ? cnorm[j] : cnmax;
                if (j == 0) { bestd = d; best = 0; }
                else if (d < bestd) { second = bestd; bestd = d; best = j; }
                else if (d < second) second = d;
            }
            lab[i] = best;
            if (lab_old && lab_old[i] != best) ndiff++;
            for (int b = 0; b < nb; b++) xn = __builtin_fma(X[(size_t)i * nb + b], X[(size_t)i * nb + b], xn);
            if (k > 1 && second - bestd <= FIT_TIE_EPS * (__builtin_fabs(bestd) + __builtin_fabs(second) + xn + cnmax)) near++;
        }
    }
    // one atomic per wavefront
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) ndiff += __shfl_xor(ndiff, s, 64);
    if (ndiff != 0u && lane_id() == 0 && ctl) atomicAdd(&ctl->ndiff, ndiff);
    if (__ballot(near != 0u) != 0ull && lane_id() == 0 && ctl) atomicAdd(&ctl->near, 1u);
}

static void launch_fit_assign(shp_ctx *ctx, unsigned g, const double *dX, uint32_t n, int nb,
                              const double *dm2c, const double *dcn, int k, int32_t *dlab,
                              const int32_t *dlab_old, FitCtl *ndiff)
{
#define FA(NBT)                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fit_assign<NBT>), dim3(g), dim3(256), 0, ctx->stream, dX, \
                       n, nb, dm2c, dcn, k, dlab, dlab_old, ndiff)
    switch (nb) {
    case 1: FA(1); break;
    case 2: FA(2); break;
    case 3: FA(3); break;
    case 4: FA(4); break;
    case 5: FA(5); break;
    case 6: FA(6); break;
    case 7: FA(7); break;
    case 8: FA(8); break;
    case 10: FA(10); break;
    case 12: FA(12); break;
    default: FA(0); break;
    }
#undef FA
}

// The M-step's sums are the reference's: one chain per (cluster, band) in ROW order (sklearn with one
// OpenMP thread; fit_elkan.h: the row numbers sorted stably by label, k_fit_sum_lists[_staged]), on
// this path too -- an earlier version added per-256-row partial sums, whose association left the
// centres 1e-9 away from the reference's bits.  pcount[j] = rows of cluster j from the list offsets.
__global__ __launch_bounds__(256) void k_fit_counts(const uint32_t *__restrict__ off, int k,
                                                    uint32_t *__restrict__ pcount, const uint32_t *stop)
{
    if (stop && *stop) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < k) pcount[j] = off[j + 1] - off[j];
}

// End of a Lloyd iteration in one workgroup: S[t] = sum over `ngroups` partial sums in order (one
// group = the row-order sums themselves: 0.0 + x is x), w[j] likewise, then -- unless a cluster came out empty, which is left to the host --
// centres = S * (1 / w), the squared centre shift, the convergence tests of sklearn 0.24.2
// (labels unchanged -> strict; shift <= tol) and the E-step operands of the next iteration
// (m2c = -2c, cnorm = |c|^2 as kmeans_prepare_host).  Every float64 operation and its order are
// those of the host code this replaces, so the iteration count and the centres are unchanged.
__global__ __launch_bounds__(256) void k_fit_update(const double *__restrict__ partial,
                                                    const uint32_t *__restrict__ pcount,
                                                    uint32_t ngroups, int k, int nb, double *S,
                                                    double *w, double *C, double *m2c, double *cnorm,
                                                    FitCtl *ctl, double tol, uint32_t it)
{
    if (ctl->stop) return;
    __shared__ int s_empty;
    const int kn = k * nb;
    if (threadIdx.x == 0) s_empty = 0;
    for (int t = threadIdx.x; t < kn; t += 256) {
        double acc = 0.0;
        uint32_t g = 0;
        for (; g + 8u <= ngroups; g += 8u) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(g + u) * kn + t];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; g < ngroups; g++) acc += partial[(size_t)g * kn + t];
        S[t] = acc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        uint32_t ww = 0;
        for (uint32_t g = 0; g < ngroups; g++) ww += pcount[(size_t)g * k + j];
        w[j] = (double)ww;
        if (ww == 0u) s_empty = 1;
    }
    __syncthreads();
    if (s_empty) {
        if (threadIdx.x == 0) ctl->stop = 2u;
        return;
    }
    for (int t = threadIdx.x; t < kn; t += 256) {
        const double alpha = 1.0 / w[t / nb];
        S[t] = S[t] * alpha;                                 // the new centres
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        const double *a = &S[j * nb], *c = &C[j * nb];
        double r = 0.0;
        int b = 0;
        for (; b + 4 <= nb; b += 4)
            r += ((a[b] - c[b]) * (a[b] - c[b]) + (a[b + 1] - c[b + 1]) * (a[b + 1] - c[b + 1]) +
                  (a[b + 2] - c[b + 2]) * (a[b + 2] - c[b + 2]) + (a[b + 3] - c[b + 3]) * (a[b + 3] - c[b + 3]));
        for (; b < nb; b++) r += (a[b] - c[b]) * (a[b] - c[b]);
        const double sq = __builtin_sqrt(r);
        w[j] = sq * sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double shift = np_pairwise_sum(w, (size_t)k);      // (center_shift**2).sum()
        const uint32_t nd = ctl->ndiff;
        ctl->shift = shift;
        ctl->iters = it;
        ctl->ndiff = 0u;
        if (nd == 0u) ctl->stop = 1u;
        else if (shift <= tol) ctl->stop = 3u;
    }
    for (int t = threadIdx.x; t < kn; t += 256) C[t] = S[t];
    for (int j = threadIdx.x; j < k; j += 256) {
        for (int b = 0; b < nb; b++) m2c[j * nb + b] = -2.0 * S[j * nb + b];
        cnorm[j] = kmeans_sqnorm(&S[j * nb], nb);
    }
}

// sklearn's X -= X.mean(axis=0) on rows of pixel type T (converted to float64 as the reference's
// check_array does): column sums in row order, X = x - mean, and the column sums of X.
template <class T>
static void fit_centre_rows(const T *xin, uint32_t n, int nb, double *X, std::vector<double> &mu,
                            std::vector<double> &acc)
{
    for (uint32_t i = 0; i < n; i++)
        for (int b = 0; b < nb; b++) acc[b] += (double)xin[(size_t)i * nb + b];
    for (int b = 0; b < nb; b++) { mu[b] = acc[b] / (double)n; acc[b] = 0.0; }
    for (uint32_t i = 0; i < n; i++)
        for (int b = 0; b < nb; b++) {
            const double xv = (double)xin[(size_t)i * nb + b] - mu[b];
            X[(size_t)i * nb + b] = xv;
            acc[b] += xv;
        }
}


// ---- sample preparation from the BAND-PLANAR form (the tiled driver's sub-sample as it leaves the
//      device: nb planes of m pixels), a host thread per band.  The same arithmetic as fit_centre_rows
//      -- every band's sums are one chain in row order, and the bands never mix -- without the
//      (m x nb) transposition on the host (5 ms for the benchmark's 10^6-row sample) and with the
//      three passes over the sample spread over nb cores (8 ms -> ~2 ms).  The centred sample is
//      written band-planar too (threads do not share cache lines) and transposed on the device.
struct PlanarPrep {
    uint32_t n = 0;                     // rows kept (non-null)
    std::vector<double> mu, var_sum;    // per band: mean, sum of squared deviations of the centred column
    std::vector<long long> vmin, vmax;  // per band, over the kept rows
};

template <class T>
static void planar_band(const T *plane, const uint32_t *idx, uint32_t n, double *Xb, double *mu_out,
                        double *var_out, long long *mn_out, long long *mx_out)
{
    double acc = 0.0;
    T mn = plane[idx ? idx[0] : 0], mx = mn;
    for (uint32_t i = 0; i < n; i++) {
        const T v = plane[idx ? idx[i] : i];
        acc += (double)v;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    const double mu = acc / (double)n;
    acc = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        const double xv = (double)plane[idx ? idx[i] : i] - mu;
        Xb[i] = xv;
        acc += xv;
    }
    const double m2 = acc / (double)n;
    double acc2 = 0.0;
    for (uint32_t i = 0; i < n; i++) { const double d = Xb[i] - m2; acc2 += d * d; }
    *mu_out = mu; *var_out = acc2; *mn_out = (long long)mn; *mx_out = (long long)mx;
}

// rows whose every band differs from the null value, in order (shepseg.py:283-299)
template <class T>
static void planar_keep(const T *planes, size_t m, int nb, long long null_val, std::vector<uint32_t> &idx)
{
    const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    std::vector<std::vector<uint32_t>> part(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            const size_t lo = m * t / nt, hi = m * (t + 1) / nt;
            std::vector<uint32_t> &out = part[t];
            for (size_t i = lo; i < hi; i++) {
                bool ok = true;
                for (int b = 0; b < nb && ok; b++) ok = (long long)planes[(size_t)b * m + i] != null_val;
                if (ok) out.push_back((uint32_t)i);
            }
        });
    for (auto &t : th) t.join();
    size_t tot = 0;
    for (auto &v : part) tot += v.size();
    idx.reserve(tot);
    for (auto &v : part) idx.insert(idx.end(), v.begin(), v.end());
}

// diagonalClusterCentres (shepseg.py:364-397) as numpy evaluates it on a sample of pixel type T:
// (max - min) in T (wrapping), / (k + 1) in float64, min + (j + 1) * step in float64, truncated to T
template <class T>
static void planar_diag_init(const PlanarPrep &pp, int nb, int k, double *init)
{
    typedef typename std::make_unsigned<T>::type UT;
    for (int b = 0; b < nb; b++) {
        const T mn = (T)pp.vmin[b], mx = (T)pp.vmax[b];
        const T diff = (T)(UT)((UT)mx - (UT)mn);
        const double step = (double)diff / (double)(k + 1);
        for (int j = 0; j < k; j++) {
            const double v = (double)mn + (double)(j + 1) * step;
            init[j * nb + b] = (double)(T)v;
        }
    }
}

// Xp (nb planes of n) -> X (n rows of nb)
__global__ __launch_bounds__(256) void k_fit_transpose(const double *__restrict__ Xp, uint32_t n, int nb,
                                                       double *__restrict__ X)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)n * nb) return;
    const uint32_t r = (uint32_t)(i / nb), b = (uint32_t)(i - (size_t)r * nb);
    X[i] = Xp[(size_t)b * n + r];
}

#define FIT_DT_F64 100        // xin holds float64 rows (shp_kmeans_fit); else one of the SHP_* pixel types
// planar: xin_any holds nb planes of nrows pixels (a pixel type, not FIT_DT_F64); rows with a null in
// any band are dropped when has_null; init == nullptr: diagonalClusterCentres of the kept rows;
// *nrows_kept_out = rows the model was fitted on (labels_out holds that many).
static int run_kmeans_fit(shp_ctx *ctx, const void *xin_any, int xdtype, int64_t nrows, int nb, int k,
                          const double *init, int max_iter, double tol_rel, double *centres_out,
                          int32_t *labels_out, int *n_iter_out, bool planar = false, int has_null = 0,
                          long long null_val = 0, int64_t *nrows_kept_out = nullptr, FitShard shard = FitShard())
{
    if (nrows < 1 || nrows > 0x7fffffffll || nb < 1 || k < 1)
        SHP_FAIL(ctx, SHP_ERR_ARG, "kmeans_fit: bad shape");
    if (!planar && nrows < k) SHP_FAIL(ctx, SHP_ERR_ARG, "n_samples=%lld should be >= n_clusters=%d",
                                       (long long)nrows, k);
    uint32_t n = (uint32_t)nrows;
    const int kn = k * nb;
    const auto t_begin = std::chrono::steady_clock::now();
    PlanarPrep pp;
    std::vector<uint32_t> keep_idx;
    std::vector<double> init_diag;
    if (planar) {
        // rows kept first (the buffers below are sized by them)
        if (has_null) {
            switch (xdtype) {
            case SHP_U8: planar_keep((const uint8_t *)xin_any, (size_t)nrows, nb, null_val, keep_idx); break;
            case SHP_I16: planar_keep((const int16_t *)xin_any, (size_t)nrows, nb, null_val, keep_idx); break;
            case SHP_U16: planar_keep((const uint16_t *)xin_any, (size_t)nrows, nb, null_val, keep_idx); break;
            case SHP_I32: planar_keep((const int32_t *)xin_any, (size_t)nrows, nb, null_val, keep_idx); break;
            case SHP_U32: planar_keep((const uint32_t *)xin_any, (size_t)nrows, nb, null_val, keep_idx); break;
            default: SHP_FAIL(ctx, SHP_ERR_ARG, "kmeans_fit: bad sample type %d", xdtype);
            }
            n = (uint32_t)keep_idx.size();
        }
        if (nrows_kept_out) *nrows_kept_out = (int64_t)n;
        if ((int64_t)n < (int64_t)k) SHP_FAIL(ctx, SHP_ERR_ARG, "n_samples=%lld should be >= n_clusters=%d",
                                              (long long)n, k);
    }
    // centre the data on the host (sklearn: X -= X.mean(axis=0)); rows outer / bands inner keeps
    // each band's additions in row order while nb independent chains are in flight.  X lives in
    // a pinned, grow-only buffer of the context: no page faults after the first call and the
    // upload runs at full PCIe speed.
    const size_t xbytes = (size_t)n * nb * 8;
    if (ctx->h_fit_cap < xbytes) {
        if (ctx->h_fit) hipHostFree(ctx->h_fit);
        ctx->h_fit = nullptr; ctx->h_fit_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_fit, xbytes + xbytes / 8, hipHostMallocDefault) != hipSuccess)
            SHP_FAIL(ctx, SHP_ERR_NOMEM, "hipHostMalloc(%zu) failed", xbytes);
        ctx->h_fit_cap = xbytes + xbytes / 8;
    }
    double *X = ctx->h_fit;
    std::vector<double> mu(nb, 0.0), acc(nb, 0.0), acc2(nb, 0.0);
    if (planar) {
        const uint32_t *ix = has_null ? keep_idx.data() : (const uint32_t *)nullptr;
        pp.n = n;
        pp.mu.assign(nb, 0.0); pp.var_sum.assign(nb, 0.0); pp.vmin.assign(nb, 0); pp.vmax.assign(nb, 0);
        std::vector<std::thread> th;
#define PLANAR_BANDS(T)                                                                              \
        for (int b = 0; b < nb; b++)                                                                     \
            th.emplace_back([&, b] {                                                                     \
                planar_band((const T *)xin_any + (size_t)b * (size_t)nrows, ix, n, X + (size_t)b * n,   \
                            &pp.mu[b], &pp.var_sum[b], &pp.vmin[b], &pp.vmax[b]);                        \
            })
        switch (xdtype) {
        case SHP_U8: PLANAR_BANDS(uint8_t); break;
        case SHP_I16: PLANAR_BANDS(int16_t); break;
        case SHP_U16: PLANAR_BANDS(uint16_t); break;
        case SHP_I32: PLANAR_BANDS(int32_t); break;
        case SHP_U32: PLANAR_BANDS(uint32_t); break;
        default: SHP_FAIL(ctx, SHP_ERR_ARG, "kmeans_fit: bad sample type %d", xdtype);
        }
#undef PLANAR_BANDS
        for (auto &t : th) t.join();
        for (int b = 0; b < nb; b++) { mu[b] = pp.mu[b]; acc2[b] = pp.var_sum[b]; }
        if (!init) {
            init_diag.resize(kn);
            switch (xdtype) {
            case SHP_U8: planar_diag_init<uint8_t>(pp, nb, k, init_diag.data()); break;
            case SHP_I16: planar_diag_init<int16_t>(pp, nb, k, init_diag.data()); break;
            case SHP_U16: planar_diag_init<uint16_t>(pp, nb, k, init_diag.data()); break;
            case SHP_I32: planar_diag_init<int32_t>(pp, nb, k, init_diag.data()); break;
            default: planar_diag_init<uint32_t>(pp, nb, k, init_diag.data()); break;
            }
            init = init_diag.data();
        }
    } else {
        switch (xdtype) {
        case FIT_DT_F64: fit_centre_rows((const double *)xin_any, n, nb, X, mu, acc); break;
        case SHP_U8: fit_centre_rows((const uint8_t *)xin_any, n, nb, X, mu, acc); break;
        case SHP_I16: fit_centre_rows((const int16_t *)xin_any, n, nb, X, mu, acc); break;
        case SHP_U16: fit_centre_rows((const uint16_t *)xin_any, n, nb, X, mu, acc); break;
        case SHP_I32: fit_centre_rows((const int32_t *)xin_any, n, nb, X, mu, acc); break;
        case SHP_U32: fit_centre_rows((const uint32_t *)xin_any, n, nb, X, mu, acc); break;
        default: SHP_FAIL(ctx, SHP_ERR_ARG, "kmeans_fit: bad sample type %d", xdtype);
        }
        for (int b = 0; b < nb; b++) acc[b] /= (double)n;
        for (uint32_t i = 0; i < n; i++)
            for (int b = 0; b < nb; b++) { const double d = X[(size_t)i * nb + b] - acc[b]; acc2[b] += d * d; }
    }
    if (!init) SHP_FAIL(ctx, SHP_ERR_ARG, "kmeans_fit: no initial centres");
    // (host X is row-major, or band-planar in the planar form)
    auto Xat = [&](uint32_t i, int b) -> double { return planar ? X[(size_t)b * n + i] : X[(size_t)i * nb + b]; };
    double tol = 0.0;
    for (int b = 0; b < nb; b++) tol += acc2[b] / (double)n;
    tol = tol / nb * tol_rel;
    std::vector<double> C(kn);
    for (int t = 0; t < kn; t++) C[t] = init[t] - mu[t % nb];

    if ((size_t)(2 * kn + 2 * k + 8) * 8 > SHP_PINNED_BYTES)
        SHP_FAIL(ctx, SHP_ERR_ARG, "k * nbands too large for the k-means fit (%d x %d)", k, nb);
    CHK(buf_ensure(ctx, ctx->fit_x, (size_t)n * nb * 8 + (size_t)n * 8));
    CHK(buf_ensure(ctx, ctx->fit_lab, (size_t)n * 4 * 2 + 64 + 1024));        // (128 words between the two label arrays: the
                                                                               //  sharded E-step's all-gather pads the first)
    CHK(buf_ensure(ctx, ctx->fit_part, (size_t)(kn + k) * 8 * 3 + ((size_t)2 * k + 4) * 4 + 512));
    CHK(buf_ensure(ctx, ctx->cen, (size_t)(kn + k) * 8 * 2));
    double *dX = bp<double>(ctx->fit_x), *ddist = dX + (size_t)n * nb;
    int32_t *dlabA = bp<int32_t>(ctx->fit_lab), *dlabB = dlabA + n + 128;
    FitCtl *dctl = (FitCtl *)(dlabB + n);
    // row-order sums | their counts as float64 (unused here) | S | w | counts | list offsets (k + 1) + a spare word
    double *dpart2 = bp<double>(ctx->fit_part), *dcntd = dpart2 + kn;
    double *dS = dcntd + k, *dw = dS + kn;
    uint32_t *dpc2 = (uint32_t *)(dw + k + 2);
    uint32_t *doff = dpc2 + k;
    double *dm2c = bp<double>(ctx->cen), *dcn = dm2c + kn, *dC = dcn + k;
    hipStream_t st = ctx->stream;
    // pinned staging: [0..] FitCtl, then m2c|cnorm|C (2kn+k doubles)
    FitCtl *pin_ctl = (FitCtl *)ctx->h_pinned;
    double *pin_up = (double *)(ctx->h_pinned + 16);
    HIPCHK(ctx, hipStreamSynchronize(st));           // earlier users of the staging area are done
    if (planar) {
        CHK(buf_ensure(ctx, ctx->aux, xbytes));
        HIPCHK(ctx, hipMemcpyAsync(ctx->aux.p, X, xbytes, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_fit_transpose, dim3(grid_for((size_t)n * nb, 256)), dim3(256), 0, st,
                           bp<double>(ctx->aux), n, nb, dX); KCHK(ctx);
    } else {
        HIPCHK(ctx, hipMemcpyAsync(dX, X, xbytes, hipMemcpyHostToDevice, st));
    }
    HIPCHK(ctx, hipMemsetAsync(dlabB, 0xff, (size_t)n * 4, st));        // labels_old = -1
    const unsigned g = grid_for(n, 256u * FIT_RPT);
    auto upload_centres = [&](const std::vector<double> &cc) -> int {
        kmeans_prepare_host(cc.data(), k, nb, pin_up, pin_up + kn);        // m2c | cnorm
        memcpy(pin_up + kn + k, cc.data(), (size_t)kn * 8);              // C  (dC follows dcn)
        HIPCHK(ctx, hipMemcpyAsync(dm2c, pin_up, (size_t)(2 * kn + k) * 8, hipMemcpyHostToDevice, st));
        return 0;
    };
    auto upload_ctl = [&](uint32_t iters) -> int {
        memset(pin_ctl, 0, sizeof(FitCtl));
        pin_ctl->iters = iters;
        HIPCHK(ctx, hipMemcpyAsync(dctl, pin_ctl, sizeof(FitCtl), hipMemcpyHostToDevice, st));
        return 0;
    };
    // The iterations run in batches of FIT_BATCH without a host round trip: the last kernel of an
    // iteration (k_fit_update) owns the convergence tests and, once it raises ctl->stop, the
    // kernels still queued behind it return at once.  Iteration `it` writes its labels to buffer
    // A when it is odd, B when even, and compares them with the other buffer (labels_old).
    const auto t_prep = std::chrono::steady_clock::now();
    bool strict = false, finished = false;
    int it_done = 0;
    // SHEPSEG_FIT_ALGO: auto (the fast path unless its tie guard fires; sklearn runs Lloyd for k == 1),
    // lloyd (never leave the fast path), elkan (always the reference's algorithm): fit_elkan.h
    const char *algo_env = getenv("SHEPSEG_FIT_ALGO");
    const bool force_lloyd = k == 1 || (algo_env && !strcmp(algo_env, "lloyd"));
    bool elkan = k > 1 && !force_lloyd && algo_env && !strcmp(algo_env, "elkan");
    const std::vector<double> C0 = C;
    ctx->fit_path = 0;
    CHK(upload_centres(C));
    CHK(upload_ctl(0));
    while (it_done < max_iter && !finished && !elkan) {
        // (the first batch is ONE iteration: where the tie guard fires at all -- integer-valued initial centres on
        //  integer imagery -- it fires in the very first E-step, and seven more iterations would be thrown away)
        const int batch = it_done == 0 ? 1 : FIT_BATCH;
        const int b_end = it_done + batch < max_iter ? it_done + batch : max_iter;
        for (int it = it_done + 1; it <= b_end; it++) {
            int32_t *dlab = (it & 1) ? dlabA : dlabB, *dlab_old = (it & 1) ? dlabB : dlabA;
            launch_fit_assign(ctx, g, dX, n, nb, dm2c, dcn, k, dlab, dlab_old, dctl); KCHK(ctx);
            {
                uint32_t *ks = nullptr, *rows = nullptr;
                CHK(sort_pairs(ctx, (const uint32_t *)dlab, nullptr, n, bits_for((uint32_t)(k - 1)), &ks, &rows));
                hipLaunchKernelGGL(k_elk_offsets, dim3(grid_for((size_t)k + 1, 256)), dim3(256), 0, st, ks, n, k, doff,
                                   doff + k + 1, &dctl->stop); KCHK(ctx);
                if (nb <= 64)
                    hipLaunchKernelGGL(k_fit_sum_lists_staged, dim3(k), dim3(FIT_SUM_THREADS), 0, st, dX, nb, rows, doff, dpart2,
                                       dcntd, &dctl->stop, FitDigits{nullptr, nullptr, 0u, n}, ElkEpilogue{});
                else
                    hipLaunchKernelGGL(k_fit_sum_lists, dim3(k, (nb + 63) / 64), dim3(64), 0, st, dX, nb, rows, doff,
                                       dpart2, dcntd, &dctl->stop, FitDigits{nullptr, nullptr, 0u, n});
                KCHK(ctx);
                hipLaunchKernelGGL(k_fit_counts, dim3(grid_for((size_t)k, 256)), dim3(256), 0, st, doff, k, dpc2,
                                   &dctl->stop); KCHK(ctx);
            }
            hipLaunchKernelGGL(k_fit_update, dim3(1), dim3(256), 0, st, dpart2, dpc2, 1u, k, nb, dS, dw,
                               dC, dm2c, dcn, dctl, tol, (uint32_t)it); KCHK(ctx);
        }
        HIPCHK(ctx, hipMemcpyAsync(pin_ctl, dctl, sizeof(FitCtl), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        const uint32_t stop = pin_ctl->stop;
        if (getenv("SHEPSEG_FIT_TIMING"))
            fprintf(stderr, "kmeans fit: batch to %d: stop %u iters %u ndiff %u near %u shift %.17g tol %.17g\n", b_end, stop,
                    pin_ctl->iters, pin_ctl->ndiff, pin_ctl->near, pin_ctl->shift, tol);
        if (pin_ctl->near != 0u && !force_lloyd) { elkan = true; break; }    // a (near) tie decided a label
        if (stop == 0u) { it_done = b_end; continue; }
        if (stop == 1u || stop == 3u) {
            it_done = (int)pin_ctl->iters;
            strict = stop == 1u;
            finished = true;
            break;
        }
        // stop == 2: an iteration found an empty cluster.  Relocations are the reference algorithm's
        // business (which samples move where depends on its distances and on numpy's partition order,
        // and what follows on Elkan's bounds): start over on that path (fit_elkan.h)
        if (force_lloyd && k > 1) SHP_FAIL(ctx, SHP_ERR_ARG, "kmeans_fit: an empty cluster needs SHEPSEG_FIT_ALGO=auto or elkan");
        elkan = true;
        break;
    }
    int it = it_done;
    int32_t *dlab = (it_done & 1) ? dlabA : dlabB;
    if (it_done == 0) dlab = dlabA;                  // max_iter < 1: labels of the initial centres
    if (elkan) {
        // the reference's algorithm from the initial centres (whatever the fast path did is dropped)
        C = C0;
        dlab = dlabA;
        if (shard.world == 1 && getenv("SHEPSEG_FIT_SHARDS") && atoi(getenv("SHEPSEG_FIT_SHARDS")) > 1) {
            // one process plays every rank in turn (or, SHEPSEG_FIT_SHARD_ONLY=r, rank r alone): fit_elkan.h FitShard
            shard.world = atoi(getenv("SHEPSEG_FIT_SHARDS"));
            shard.only = getenv("SHEPSEG_FIT_SHARD_ONLY") ? atoi(getenv("SHEPSEG_FIT_SHARD_ONLY")) : -1;
            if (shard.world > 64 || (size_t)shard.world > (size_t)n) shard = FitShard();
        }
        CHK(run_fit_elkan(ctx, dX, Xat, n, nb, k, C, max_iter, tol, dlab, ddist, &it, shard, dlabB));
        ctx->fit_path = 1;
        HIPCHK(ctx, hipMemcpyAsync(dC, C.data(), (size_t)kn * 8, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    } else if (!strict) {
        // extra E-step so that the labels match the final centres (the device already holds their
        // m2c | cnorm; either label buffer will do)
        launch_fit_assign(ctx, g, dX, n, nb, dm2c, dcn, k, dlab, (const int32_t *)nullptr, (FitCtl *)nullptr); KCHK(ctx);
    }
    HIPCHK(ctx, hipMemcpyAsync(pin_up, dC, (size_t)kn * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (int t = 0; t < kn; t++) C[t] = pin_up[t];
    if (labels_out) {       // via the pinned sample buffer (X is no longer needed): see shp_dev_subsample
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_fit, dlab, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        memcpy(labels_out, ctx->h_fit, (size_t)n * 4);
    }
    for (int t = 0; t < kn; t++) centres_out[t] = C[t] + mu[t % nb];
    if (n_iter_out) *n_iter_out = it;
    if (getenv("SHEPSEG_FIT_TIMING")) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "kmeans fit: n=%u k=%d iterations=%d  host prep + upload %.2f ms  Lloyd %.2f ms\n", n, k, it,
                std::chrono::duration<double, std::milli>(t_prep - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_prep).count());
    }
    return 0;
}
