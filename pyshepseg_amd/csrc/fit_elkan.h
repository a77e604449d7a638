// fit_elkan.h -- the reference's own k-means fit, operation by operation (included by kmeans.h).
//
// shepseg.fitSpectralClusters (shepseg.py:305-312) calls KMeans(n_init=1, init=<array>).fit with
// sklearn's default algorithm="auto", which in the reference's stack (sklearn 0.24.2, _kmeans.py:824)
// is ELKAN's variant for k > 1: _kmeans_single_elkan (:300-428) over _k_means_elkan.pyx
// init_bounds_dense / elkan_iter_chunked_dense.  In exact arithmetic it visits the partitions of
// Lloyd's algorithm.  In float64 it does not wherever a sample is (nearly) equidistant from two
// centres: distances are the direct sqrt(sum (x - c)^2), a sample keeps its label unless another
// centre is STRICTLY closer, and triangle-inequality bounds decide which distances are looked at at
// all -- on lattice-valued imagery (8-bit, few bands) exact ties are common and steer the whole fit
// (oracle/refgen/probe_elkan.py: the Lloyd restatement differs from the reference on 67 of 357 such
// fits, this one on none).
//
// run_kmeans_fit therefore has two paths with the same result where both apply:
//  * the fast one (kmeans.h: Lloyd E-step in the dgemm order, the same row-order M-step sums as below,
//    8 iterations per host round trip) with a guard: an E-step that meets a sample whose two nearest centres are
//    within FIT_TIE_EPS (relative) raises FitCtl::near.  Without such a sample every label is decided
//    by a margin far above the rounding of either evaluation, so both algorithms visit the same
//    partitions, compute the same centres from them (bit for bit: same sums in the same order) and
//    stop after the same iteration;
//  * this one, taken when the guard fires (SHEPSEG_FIT_ALGO=elkan: always): bounds kept per sample and
//    centre (k x n float64, cluster-major), M-step sums in ROW order per cluster (sklearn with one
//    OpenMP thread; row lists by one stable radix pass over the labels), the k x nb sized tail of an
//    iteration in a one-workgroup kernel (k_elk_update), eight iterations per host round trip; an
//    iteration that leaves a cluster empty is finished on the host.  Bit-identical to the oracle's
//    orc_kmeans_fit_elkan, i.e. to the reference with OMP_NUM_THREADS=1 (with more threads sklearn
//    adds per-thread partial sums in the order the threads finish: not reproducible run to run).
// Shared by both: empty-cluster relocation as numpy evaluates it (pairwise row sums, np.argpartition's
// introselect), center_shift_tot as numpy's pairwise sum.
#pragma once
#include "comm.h"        // the row-sharded E-step all-gathers its labels with RCCL (FitShard)

#define FIT_TIE_EPS 1e-12

// DOUBLE_pairwise_sum (numpy/core/src/umath/loops_utils.h.src): what a float64 .sum() over a
// contiguous run evaluates; plain left to right below 8 elements.  get(i) yields element i.
// (the halving above 128 elements is numpy's recursion; here its depth is a template parameter, so that
//  device code gets plain nested calls instead of a recursive function on a dynamic stack -- the
//  recursive form made k_fit_update, the one kernel that called it out of line, read its loop control
//  back wrongly.  LEVELS = 6 covers the 8192 elements numpy hands the routine at a time, see below.)
template <int LEVELS, class Get>
__host__ __device__ inline double np_pairwise_sum_lv(Get get, size_t lo, size_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (size_t i = 0; i < n; i++) res += get(lo + i);
        return res;
    }
    if (n <= 128) {
        double r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = get(lo + j);
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += get(lo + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += get(lo + i);
        return res;
    }
    if constexpr (LEVELS > 0) {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_lv<LEVELS - 1>(get, lo, n2) + np_pairwise_sum_lv<LEVELS - 1>(get, lo + n2, n - n2);
    } else {
        return __builtin_nan("");           // more than 128 * 2^10 elements: no caller gets here
    }
}
// A .sum() over a contiguous run longer than the ufunc buffer (8192 elements) reaches the routine above one
// buffer at a time, the blocks' sums added one after the other.
#define NP_REDUCE_BLOCK ((size_t)8192)
template <class Get>
__host__ __device__ inline double np_pairwise_sum_fn(Get get, size_t lo, size_t n)
{
    double res = 0.0;
    for (size_t o = 0; o < n || o == 0; o += NP_REDUCE_BLOCK) {
        const size_t m = n - o < NP_REDUCE_BLOCK ? n - o : NP_REDUCE_BLOCK;
        const double part = np_pairwise_sum_lv<6>(get, lo + o, m);
        res = o ? res + part : part;
        if (n == 0) break;
    }
    return res;
}
__host__ __device__ inline double np_pairwise_sum(const double *a, size_t n)
{
    return np_pairwise_sum_fn([a](size_t i) { return a[i]; }, 0, n);
}

// np.argpartition(v, kth) for float64 without NaNs: numpy/core/src/npysort/selection.cpp
// introselect_<double, arg> from the index array 0..num-1 (oracle: np_aintroselect, pinned there)
static void np_aintroselect(const double *v, int64_t *t, int64_t num, int64_t kth);
static int64_t np_amedian5(const double *v, int64_t *t)
{
    if (v[t[1]] < v[t[0]]) std::swap(t[1], t[0]);
    if (v[t[4]] < v[t[3]]) std::swap(t[4], t[3]);
    if (v[t[3]] < v[t[0]]) std::swap(t[3], t[0]);
    if (v[t[4]] < v[t[1]]) std::swap(t[4], t[1]);
    if (v[t[2]] < v[t[1]]) std::swap(t[2], t[1]);
    if (v[t[3]] < v[t[2]]) return (v[t[3]] < v[t[1]]) ? 1 : 3;
    return 2;
}
static int64_t np_amedian_of_median5(const double *v, int64_t *t, int64_t num)
{
    const int64_t nmed = num / 5;
    for (int64_t i = 0, subleft = 0; i < nmed; i++, subleft += 5) {
        const int64_t m = np_amedian5(v, t + subleft);
        std::swap(t[subleft + m], t[i]);
    }
    if (nmed > 2) np_aintroselect(v, t, nmed, nmed / 2);
    return nmed / 2;
}
static void np_aintroselect(const double *v, int64_t *t, int64_t num, int64_t kth)
{
    int64_t low = 0, high = num - 1;
    if (kth - low < 3) {
        const int64_t n2 = high - low + 1;
        for (int64_t i = 0; i <= kth - low; i++) {
            int64_t minidx = i;
            double minval = v[t[low + i]];
            for (int64_t q = i + 1; q < n2; q++)
                if (v[t[low + q]] < minval) { minidx = q; minval = v[t[low + q]]; }
            std::swap(t[low + i], t[low + minidx]);
        }
        return;
    }
    if (kth == num - 1) {
        int64_t maxidx = low;
        double maxval = v[t[low]];
        for (int64_t q = low + 1; q < num; q++)
            if (!(v[t[q]] < maxval)) { maxidx = q; maxval = v[t[q]]; }
        std::swap(t[kth], t[maxidx]);
        return;
    }
    int depth_limit = 0;
    for (uint64_t u = (uint64_t)num >> 1; u; u >>= 1) depth_limit++;
    depth_limit *= 2;
    for (; low + 1 < high;) {
        int64_t ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int64_t mid = low + (high - low) / 2;
            if (v[t[high]] < v[t[mid]]) std::swap(t[high], t[mid]);
            if (v[t[high]] < v[t[low]]) std::swap(t[high], t[low]);
            if (v[t[low]] < v[t[mid]]) std::swap(t[low], t[mid]);
            std::swap(t[mid], t[low + 1]);
        } else {
            const int64_t mid = ll + np_amedian_of_median5(v, t + ll, hh - ll);
            std::swap(t[mid], t[low]);
            ll--; hh++;
        }
        depth_limit--;
        const double pivot = v[t[low]];
        for (;;) {
            do ll++; while (v[t[ll]] < pivot);
            do hh--; while (pivot < v[t[hh]]);
            if (hh < ll) break;
            std::swap(t[hh], t[ll]);
        }
        std::swap(t[low], t[hh]);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1 && v[t[high]] < v[t[low]]) std::swap(t[high], t[low]);
}

// dist[i] = ((X_i - C[lab_i])**2).sum() as numpy sums a row (empty-cluster relocation)
__global__ __launch_bounds__(256) void k_fit_dist(const double *__restrict__ X, uint32_t n, int nb,
                                                  const int32_t *__restrict__ lab,
                                                  const double *__restrict__ C,
                                                  double *__restrict__ dist)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const double *x = X + (size_t)i * nb, *c = C + (size_t)lab[i] * nb;
    dist[i] = np_pairwise_sum_fn([x, c](size_t b) { const double t = x[b] - c[b]; return t * t; }, 0, (size_t)nb);
}

// The tail of an M-step on the host (k x nb numbers; both paths): empty clusters take the samples
// farthest from their OLD centres (_relocate_empty_clusters_dense: empty clusters ascending, the r-th
// gets np.argpartition(distances, -n_empty)[n - 1 - r]), centres = sums * (1 / weight)
// (_average_centers), shifts (_center_shift: _euclidean_dense_dense) and their squared sum.
// Cn: the cluster sums in, the new centres out.  fetch(dist, labels) brings the n distances and labels
// from the device when a cluster is empty.
template <class XAt, class Fetch>
static int fit_mstep_tail(int k, int nb, uint32_t n, std::vector<double> &Cn, std::vector<double> &w,
                          const std::vector<double> &C, XAt Xat, Fetch fetch, std::vector<double> &cshift,
                          double *shift_tot)
{
    std::vector<int> empties;
    for (int j = 0; j < k; j++) if (w[j] == 0.0) empties.push_back(j);
    const int n_empty = (int)empties.size();
    if (n_empty > 0) {
        std::vector<double> dist;
        std::vector<int32_t> hl;
        const int rc = fetch(dist, hl);
        if (rc) return rc;
        std::vector<int64_t> part(n);
        for (uint32_t i = 0; i < n; i++) part[i] = i;
        np_aintroselect(dist.data(), part.data(), (int64_t)n, (int64_t)n - n_empty);
        for (int r = 0; r < n_empty; r++) {
            const uint32_t f = (uint32_t)part[n - 1u - (uint32_t)r];
            const int e = empties[r], old = hl[f];
            for (int b = 0; b < nb; b++) {
                Cn[old * nb + b] -= Xat(f, b);
                Cn[e * nb + b] = Xat(f, b);
            }
            w[e] = 1.0; w[old] -= 1.0;
        }
    }
    for (int j = 0; j < k; j++)
        if (w[j] > 0.0) { const double alpha = 1.0 / w[j]; for (int b = 0; b < nb; b++) Cn[j * nb + b] *= alpha; }
    cshift.resize(k);
    std::vector<double> sq(k);
    for (int j = 0; j < k; j++) {
        const double *a = &Cn[j * nb], *c = &C[j * nb];
        double r = 0.0; int b = 0;
        for (; b + 4 <= nb; b += 4)
            r += ((a[b] - c[b]) * (a[b] - c[b]) + (a[b + 1] - c[b + 1]) * (a[b + 1] - c[b + 1]) +
                  (a[b + 2] - c[b + 2]) * (a[b + 2] - c[b + 2]) + (a[b + 3] - c[b + 3]) * (a[b + 3] - c[b + 3]));
        for (; b < nb; b++) r += (a[b] - c[b]) * (a[b] - c[b]);
        const double s = __builtin_sqrt(r);
        cshift[j] = s;
        sq[j] = s * s;
    }
    *shift_tot = np_pairwise_sum(sq.data(), (size_t)k);
    return 0;
}

// ---- Elkan's E-step ---------------------------------------------------------------------------
// _euclidean_dense_dense(x, c, nf, squared=False)
__host__ __device__ inline double elk_dist(const double *a, const double *b, int nf)
{
    const int n4 = nf / 4, rem = nf % 4;
    double result = 0.0;
    for (int i = 0; i < n4; i++) {
        result += ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) +
                   (a[2] - b[2]) * (a[2] - b[2]) + (a[3] - b[3]) * (a[3] - b[3]));
        a += 4; b += 4;
    }
    for (int i = 0; i < rem; i++) result += (a[i] - b[i]) * (a[i] - b[i]);
    return __builtin_sqrt(result);
}

// center_half_distances = euclidean_distances(centres) / 2 (-2 C.C^T + |c|^2 + |c|^2 clipped at 0, zero
// diagonal, sqrt; the product's terms multiplied and added one after the other in band order, as the
// reference stack's BLAS does for these shapes) and distance_next_center = the second smallest of every
// column (np.partition(half, 1, axis=0)[1]).  Host: k x k numbers.
static void elk_half_distances(const double *C, int k, int nb, double *half, double *next)
{
    std::vector<double> xx(k);
    for (int a = 0; a < k; a++) xx[a] = kmeans_sqnorm(C + (size_t)a * nb, nb);
    for (int a = 0; a < k; a++)
        for (int b = 0; b < k; b++) {
            double d = 0.0;
            for (int t = 0; t < nb; t++) d = d + C[(size_t)a * nb + t] * C[(size_t)b * nb + t];
            double v = -2.0 * d;
            v = v + xx[a];
            v = v + xx[b];
            if (!(v > 0.0)) v = 0.0;
            if (a == b) v = 0.0;
            half[(size_t)a * k + b] = __builtin_sqrt(v) / 2.0;
        }
    for (int l = 0; l < k; l++) {
        double m0 = half[l], m1 = -1.0;
        for (int a = 1; a < k; a++) {
            const double v = half[(size_t)a * k + l];
            if (v < m0) { m1 = m0; m0 = v; }
            else if (m1 < 0.0 || v < m1) m1 = v;
        }
        next[l] = k > 1 ? m1 : m0;
    }
}

#include "fit_bounds.h"        // k <= 64: the E-step without a table of exact bounds

// init_bounds_dense: lb is cluster-major (lb[j * n + i]), zero-filled by the caller
__global__ __launch_bounds__(256) void k_elk_init(const double *__restrict__ X, uint32_t n, int nb,
                                                  const double *__restrict__ C, int k,
                                                  const double *__restrict__ half,
                                                  int32_t *__restrict__ lab, double *__restrict__ ub,
                                                  double *__restrict__ lb)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const double *x = X + (size_t)i * nb;
    int best = 0;
    double min_dist = elk_dist(x, C, nb);
    lb[i] = min_dist;
    for (int j = 1; j < k; j++)
        if (min_dist > half[(size_t)best * k + j]) {
            const double dist = elk_dist(x, C + (size_t)j * nb, nb);
            lb[(size_t)j * n + i] = dist;
            if (dist < min_dist) { min_dist = dist; best = j; }
        }
    lab[i] = best;
    ub[i] = min_dist;
}

// _update_chunk_dense's relabelling, preceded by the bounds update that ends elkan_iter for the previous
// iteration when cshift != nullptr (upper += shift of the own centre; lower -= shift, clipped at 0), in
// ONE pass over the sample's k lower bounds (cluster-major: consecutive lanes read consecutive
// addresses), ELK_AHEAD loads in flight.  The reference updates all bounds first and relabels afterwards; the
// only place where the order shows is the write of the tightened upper bound into lb[label] while the
// scan is still below `label`: that entry is then already final when the scan reaches it (`fresh`).
// (A lazy form -- replaying the missed shifts only for samples whose upper bound does not clear the
// nearest other centre -- was tried: on the benchmark sample most samples fail that test in every
// iteration, and the replay's dependent loads made the kernel three times slower.)
// *ndiff += labels changed.
#define ELK_AHEAD_ANY 8        // the same for the one-pass kernel of k > 64
#define ELK_AHEAD 20          // bounds loads in flight per thread (8: 0.303 ms per pass on the benchmark sample, 20: 0.285)
__global__ __launch_bounds__(256) void k_elk_estep(const double *__restrict__ X, uint32_t n, int nb,
                                                   const double *__restrict__ C, int k,
                                                   const double *__restrict__ half,
                                                   const double *__restrict__ next,
                                                   const double *__restrict__ cshift,
                                                   int32_t *__restrict__ lab, double *__restrict__ ub,
                                                   double *__restrict__ lb, uint32_t *ndiff,
                                                   const uint32_t *stop)
{
    if (stop && *stop) return;                  // the loop has ended: the rest of the batch is a no-op
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t changed = 0;
    if (i < n) {
        const double *x = X + (size_t)i * nb;
        int label = lab[i];
        const int label0 = label;
        double upper = ub[i];
        if (cshift) upper += cshift[label];
        const bool open = !(next[label] >= upper);
        bool tight = false;
        int fresh = -1;
        double fresh_val = 0.0;
        double *lbi = lb + i;
        for (int j0 = 0; j0 < k; j0 += ELK_AHEAD_ANY) {
            double pre[ELK_AHEAD_ANY];
#pragma unroll
            for (int u = 0; u < ELK_AHEAD_ANY; u++) {
                const int jj = j0 + u < k ? j0 + u : k - 1;
                pre[u] = lbi[(size_t)jj * n];
            }
#pragma unroll
            for (int u = 0; u < ELK_AHEAD_ANY; u++) {
                const int j = j0 + u;
                if (j >= k) break;
                double v;
                if (j == fresh) v = fresh_val;          // (pre[u] may predate the write)
                else {
                    v = pre[u];
                    if (cshift) {
                        v -= cshift[j];
                        if (v < 0) v = 0;
                        lbi[(size_t)j * n] = v;
                    }
                }
                if (open && j != label && upper > v && upper > half[(size_t)label * k + j]) {
                    if (!tight) {
                        upper = elk_dist(x, C + (size_t)label * nb, nb);
                        lbi[(size_t)label * n] = upper;
                        if (label > j) { fresh = label; fresh_val = upper; }
                        tight = true;
                    }
                    if (upper > v || upper > half[(size_t)label * k + j]) {
                        const double dist = elk_dist(x, C + (size_t)j * nb, nb);
                        lbi[(size_t)j * n] = dist;
                        if (dist < upper) { label = j; upper = dist; }
                    }
                }
            }
        }
        if (open) lab[i] = label;
        ub[i] = upper;
        changed = label != label0;
    }
    const unsigned long long m = __ballot(changed != 0u);
    if (m != 0ull && lane_id() == 0) atomicAdd(ndiff, (uint32_t)__popcll(m));
}

// The same for k <= 64 in two phases, the reference's own order: first every lower bound is updated in a
// branch-free streaming pass that also notes, one bit per centre, where the (shifted) upper bound exceeds
// both the lower bound and the half distance from the sample's centre (the centres' half distances sit in
// LDS: a per-lane gather); then only those centres are visited, in index order, with the reference's
// tests on the current values.  The set is a superset of the centres the reference's scan stops at while
// the label stands, because the upper bound only falls during the scan -- except when the tightened
// distance comes out a rounding above the bound it replaces; after that, and after a relabelling (the half
// distances are then the new centre's), the rest of the set is rebuilt.  In the one-pass form nearly
// every centre's branch was taken by some lane of a wavefront; here a wavefront runs as many distance
// evaluations as its busiest lane needs.  (As two kernels -- eight bounds per thread in a pure streaming
// pass at 4.4 TB/s, then a thread per sample for the candidates -- the pass took 0.22 + 0.14 ms against
// 0.31 ms here: the candidates' scattered accesses hide behind the streaming when they share a kernel.)
__global__ __launch_bounds__(256) void k_elk_estep64(const double *__restrict__ X, uint32_t n, int nb,
                                                     const double *__restrict__ C, int k,
                                                     const double *__restrict__ half,
                                                     const double *__restrict__ next,
                                                     const double *__restrict__ cshift,
                                                     int32_t *__restrict__ lab, double *__restrict__ ub,
                                                     double *__restrict__ lb, uint32_t *ndiff,
                                                     const uint32_t *stop)
{
    if (stop && *stop) return;
    __shared__ double sh[64 * 64];
    for (int t = threadIdx.x; t < k * k; t += 256) sh[t] = half[t];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t changed = 0;
    if (i < n) {
        const double *x = X + (size_t)i * nb;
        int label = lab[i];
        const int label0 = label;
        double upper = ub[i];
        if (cshift) upper += cshift[label];
        double *lbi = lb + i;
        const double *hrow = sh + label * k;
        unsigned long long cand = 0ull;
        for (int j0 = 0; j0 < k; j0 += ELK_AHEAD) {
            double pre[ELK_AHEAD];
#pragma unroll
            for (int u = 0; u < ELK_AHEAD; u++) {
                const int jj = j0 + u < k ? j0 + u : k - 1;
                pre[u] = lbi[(size_t)jj * n];
            }
#pragma unroll
            for (int u = 0; u < ELK_AHEAD; u++) {
                const int j = j0 + u;
                if (j < k) {
                    double v = pre[u];
                    if (cshift) {
                        v -= cshift[j];
                        if (v < 0) v = 0;
                        lbi[(size_t)j * n] = v;
                    }
                    cand |= (upper > v && upper > hrow[j]) ? (1ull << j) : 0ull;
                }
            }
        }
        if (!(next[label] >= upper)) {
            bool tight = false;
            cand &= ~(1ull << label);
            // the centres after j that can pass the reference's first test with the current label and bound
            auto rebuild = [&](int j) {
                unsigned long long c2 = 0ull;
                for (int q = j + 1; q < k; q++)
                    if (q != label && upper > lbi[(size_t)q * n] && upper > sh[label * k + q]) c2 |= 1ull << q;
                return c2;
            };
            while (cand) {
                const int j = __builtin_ctzll(cand);
                cand &= cand - 1ull;
                if (j == label) continue;
                if (upper > lbi[(size_t)j * n] && upper > sh[label * k + j]) {
                    if (!tight) {
                        const double was = upper;
                        upper = elk_dist(x, C + (size_t)label * nb, nb);
                        lbi[(size_t)label * n] = upper;
                        tight = true;
                        if (upper > was) cand = rebuild(j);        // a rounding above the bound it replaces
                    }
                    if (upper > lbi[(size_t)j * n] || upper > sh[label * k + j]) {
                        const double dist = elk_dist(x, C + (size_t)j * nb, nb);
                        lbi[(size_t)j * n] = dist;
                        if (dist < upper) { label = j; upper = dist; cand = rebuild(j); }
                    }
                }
            }
            lab[i] = label;
        }
        ub[i] = upper;
        changed = label != label0;
    }
    const unsigned long long m = __ballot(changed != 0u);
    if (m != 0ull && lane_id() == 0) atomicAdd(ndiff, (uint32_t)__popcll(m));
}

// off[j] = first position of label j in the sorted labels (off[k] = n); *zero_me = 0
__global__ __launch_bounds__(256) void k_elk_offsets(const uint32_t *__restrict__ keys, uint32_t n, int k,
                                                     uint32_t *__restrict__ off, uint32_t *zero_me,
                                                     const uint32_t *stop)
{
    if (stop && *stop) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) *zero_me = 0u;          // the next E-step's change counter
    if (j > k) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (keys[mid] < (uint32_t)j) lo = mid + 1u; else hi = mid;
    }
    off[j] = lo;
}

// S[j][b] = sum of X[row][b] over the rows of cluster j in ROW order (rows: the stable sort of the row
// numbers by label), one chain per (cluster, band): a lane per band, FIT_ROWS_AHEAD independent loads in
// flight, the additions one after the other.  grid (k, ceil(nb / 64)).
#define FIT_ROWS_AHEAD 32
// (the clusters' ranges of the row list: off[j] .. off[j + 1], or -- dg.hscan given -- read off the scanned digit
//  histogram of the one-pass radix sort that made the list)
struct FitDigits { const uint32_t *hscan, *boff; uint32_t nblk, n; };
__device__ __forceinline__ void fit_range(const uint32_t *off, const FitDigits &dg, int j, uint32_t *q0, uint32_t *q1)
{
    if (dg.hscan) {
        *q0 = sort_digit_start(dg.hscan, dg.boff, dg.nblk, (uint32_t)j, dg.n);
        *q1 = sort_digit_start(dg.hscan, dg.boff, dg.nblk, (uint32_t)j + 1u, dg.n);
    } else { *q0 = off[j]; *q1 = off[j + 1]; }
}
// SHEPSEG_FIT_CHECK_DIGITS=1: the clusters' ranges as the sums kernels read them, for a comparison on the host
__global__ __launch_bounds__(256) void k_fit_digit_starts(FitDigits dg, int k, uint32_t *out)
{
    for (int j = threadIdx.x; j <= k; j += 256) out[j] = sort_digit_start(dg.hscan, dg.boff, dg.nblk, (uint32_t)j, dg.n);
}
__global__ __launch_bounds__(64) void k_fit_sum_lists(const double *__restrict__ X, int nb,
                                                      const uint32_t *__restrict__ rows,
                                                      const uint32_t *__restrict__ off,
                                                      double *__restrict__ S, double *__restrict__ cnt,
                                                      const uint32_t *stop, FitDigits dg)
{
    if (stop && *stop) return;
    const int j = blockIdx.x, b = blockIdx.y * 64 + threadIdx.x;
    uint32_t q0, q1;
    fit_range(off, dg, j, &q0, &q1);
    if (b == 0) cnt[j] = (double)(q1 - q0);
    if (b >= nb) return;
    double acc = 0.0;
    for (uint32_t q = q0; q < q1; q += FIT_ROWS_AHEAD) {
        double v[FIT_ROWS_AHEAD];
#pragma unroll
        for (int u = 0; u < FIT_ROWS_AHEAD; u++) {
            const uint32_t qq = q + (uint32_t)u < q1 ? q + (uint32_t)u : q1 - 1u;
            v[u] = X[(size_t)rows[qq] * nb + b];
        }
#pragma unroll
        for (int u = 0; u < FIT_ROWS_AHEAD; u++)
            if (q + (uint32_t)u < q1) acc += v[u];
    }
    S[(size_t)j * nb + b] = acc;
}

// The row-order sums as a chain of v_mfma_f64_4x4x4_4b_f64.  With B = 1.0 in every lane the instruction
// computes, for each of 16 (block, row) pairs m, D = (((C + A[m][0]) + A[m][1]) + A[m][2]) + A[m][3]: four
// IEEE float64 additions one after the other in ascending k, bit for bit what four `acc += x` statements
// give (tools/ubench/mfma_f64_order.hip checks exactly that on the hardware: one-hot inputs for the lane
// map, 200 000 random trials x 64 lanes with 60 binades of spread and near-cancellations against every
// permutation of the four; a x 1.0 is exact and a fused multiply-add of an exact product is an addition).
// A[m][k] is lane m + 16 k; the result for m appears in lanes 16 (m & 3) + 4 (m >> 2) + {0..3}, which is
// also where C is read, so the accumulator stays put.  A dependent chain of these takes 20.4 cycles per
// instruction = 5.1 cycles per row for 16 chains at once; the VALU form costs 5.5 for the dependent
// v_add_f64 PLUS 8.8 for the half ds_read_b128 that feeds it (a lone wavefront pays ~4.4 issue cycles per
// dword an LDS read returns, whatever the number of active lanes): 14.3.  Here one ds_read_b64 feeds four
// rows and issues in the shadow of the previous instruction.  Padding rows are -0.0: x + -0.0 == x for
// every x, -0.0 and +0.0 included.
__device__ __forceinline__ double fit_mfma4(double a, double acc)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, acc, 0, 0, 0);
}
// 32 n rows (n >= 1) of the 16 chains of one wavefront: lane m + 16 k reads p_m[4 i + k] at LDS byte address
// `addr` + 32 i.  Two register sets of eight operands, v[64:79] and v[80:95]; the reads of one are issued
// between the instructions that consume the other, eight instructions (~160 cycles) ahead of their use, and
// LDS reads return in order, so "at most seven reads outstanding" is the wait of every instruction.  A
// dependent v_mfma_f64_4x4x4 needs four wait states behind its producer (hipcc puts s_nop 3 there): the
// read, the wait and an s_nop 1 stand in them.
#define FCM_RD(r, off) "ds_read_b64 v[" #r ":" #r "+1], %[a] offset:" #off "\n"
#define FCM_MF(r) "s_waitcnt lgkmcnt(7)\n v_mfma_f64_4x4x4_4b_f64 %[acc], v[" #r ":" #r "+1], %[one], %[acc]\n"
#define FCM_MF_LAST(r) "v_mfma_f64_4x4x4_4b_f64 %[acc], v[" #r ":" #r "+1], %[one], %[acc]\n s_nop 3\n"
#define FCM_STEP(x, y, o) FCM_MF(x) FCM_RD(y, o) "s_nop 1\n"
#define FCM_HALF(x0, y0, o)                                                                                   \
    FCM_STEP(x0, y0, o) FCM_STEP(x0 + 2, y0 + 2, o + 32) FCM_STEP(x0 + 4, y0 + 4, o + 64)                      \
    FCM_STEP(x0 + 6, y0 + 6, o + 96) FCM_STEP(x0 + 8, y0 + 8, o + 128) FCM_STEP(x0 + 10, y0 + 10, o + 160)     \
    FCM_STEP(x0 + 12, y0 + 12, o + 192) FCM_STEP(x0 + 14, y0 + 14, o + 224)
#define FCM_TAIL(x0)                                                                                          \
    FCM_MF_LAST(x0) FCM_MF_LAST(x0 + 2) FCM_MF_LAST(x0 + 4) FCM_MF_LAST(x0 + 6) FCM_MF_LAST(x0 + 8)              \
    FCM_MF_LAST(x0 + 10) FCM_MF_LAST(x0 + 12) FCM_MF_LAST(x0 + 14)
__device__ __forceinline__ double fit_chain_mfma32(double acc, uint32_t addr, uint32_t n)
{
    uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    const double one = 1.0;
    asm volatile(
        FCM_RD(64, 0) FCM_RD(66, 32) FCM_RD(68, 64) FCM_RD(70, 96) FCM_RD(72, 128) FCM_RD(74, 160) FCM_RD(76, 192)
        FCM_RD(78, 224)
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 2f\n"
        "1:\n"
        FCM_HALF(64, 80, 256)
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 3f\n"
        "v_add_u32 %[a], 0x200, %[a]\n"
        FCM_HALF(80, 64, 0)
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc0 1b\n"
        "2:\n"
        "s_waitcnt lgkmcnt(0)\n"
        FCM_TAIL(64)
        "s_branch 4f\n"
        "3:\n"
        "s_waitcnt lgkmcnt(0)\n"
        FCM_TAIL(80)
        "4:\n"
        "s_nop 7\n"                                  // (the result is next read by code hipcc does not see behind)
        : [acc] "+v"(acc), [a] "+v"(addr), [n] "+s"(left)
        : [one] "v"(one)
        : "scc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
          "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91",
          "v92", "v93", "v94", "v95");
    return acc;
}

// The same for nb <= 64 with the rows staged through LDS, a workgroup per cluster.  The time of this kernel
// is the chain of dependent additions of the largest cluster, PROVIDED the adding wavefront never waits for
// memory: wave 0 only adds (fit_chain_mfma32: 16 bands' chains per instruction, ceil(nb / 16) accumulators),
// waves 1..7 only gather.  A block is B rows; while block s is summed from one LDS buffer, the gatherers
// store block s + 1 (loaded a step ago) into the other, issue the value loads of block s + 2 and the
// row-number loads of block s + 3, so every global load has a whole step to arrive.  On the benchmark sample
// (1 032 256 rows, 60 clusters of ~17 200): 190 us in round 2 (all threads gathered AND wave 0 added, two
// dependent round trips per block plus the chain), 129 us with the gatherers split off (the VALU chain:
// 14.4 cycles per row measured), 81 us with the MFMA chain (6.6 cycles per row).
#define FIT_STAGE_DOUBLES 6144u       // 48 KiB per buffer, two buffers
#ifdef FIT_SUM_DIAG
__device__ unsigned long long g_fit_diag[256 * 4];      // tools/ubench/sumlists.hip: cycles adding / at barriers
#endif
#define FIT_SUM_THREADS 512u
#define FIT_GATHERERS (FIT_SUM_THREADS - 64u)
#define FIT_STAGE_PER_THREAD ((FIT_STAGE_DOUBLES + FIT_GATHERERS - 1u) / FIT_GATHERERS)
// loop control of the Elkan iterations, owned by the device between host synchronisations
struct ElkCtl {
    uint32_t stop;      // 0 running; 1 labels unchanged (strict convergence); 2 an empty cluster: the host
                        // finishes this iteration; 3 centre shift <= tol
    uint32_t iters;     // completed iterations
    uint32_t nd[2];     // labels changed by the E-step of iteration it: nd[it & 1]
    double shift_tot;
};

struct ElkEpilogue {
    ElkCtl *ctl;                    // nullptr: no epilogue
    uint32_t *done;                 // zero-initialised device word: workgroups finished
    double *C, *cshift, *half, *next;
    double *scratch;                // 3 k doubles of device memory (the tail's small vectors when its tables are not in LDS)
    double tol;
    uint32_t it;
    ElkHist hist;
};
// ep.ctl != nullptr: the workgroup that finishes LAST also runs the end of the iteration (elk_update_body, the
// whole of k_elk_update) -- one launch and its gap less per iteration, and the tail's tables live in this
// kernel's LDS instead of crossing global memory between its phases.
__device__ void elk_update_body(double *lds, uint32_t nthreads, const ElkEpilogue &ep, double *S, const double *cnt, int k, int nb);
__global__ __launch_bounds__(FIT_SUM_THREADS) void k_fit_sum_lists_staged(const double *__restrict__ X, int nb,
                                                             const uint32_t *__restrict__ rows,
                                                             const uint32_t *__restrict__ off,
                                                             double *__restrict__ S, double *__restrict__ cnt,
                                                             const uint32_t *stop, FitDigits dg, ElkEpilogue ep)
{
    if (stop && *stop) return;
    constexpr uint32_t BUF = FIT_STAGE_DOUBLES + 4u * 64u;        // nb runs of B + 4 doubles, nb <= 64
    __shared__ __attribute__((aligned(16))) double sx[2u * BUF];
    const int j = blockIdx.x;
    uint32_t q0, q1;
    fit_range(off, dg, j, &q0, &q1);
    const uint32_t unb = (uint32_t)nb;
    const uint32_t B = (FIT_STAGE_DOUBLES / unb) & ~31u;         // rows per block (whole 32-row groups of the chain)
    const uint32_t pitch = B + 4u;                               // doubles between two bands' runs: bands 8 banks apart
    const uint32_t nblk = (q1 - q0 + B - 1u) / B;
    if (threadIdx.x == 0) cnt[j] = (double)(q1 - q0);
    static_assert(FIT_STAGE_DOUBLES / 64u >= 32u, "a block holds at least one 32-row group of 64 bands");
    if (nblk == 0u) {
        if (threadIdx.x < unb) S[(size_t)j * unb + threadIdx.x] = 0.0;
    } else if (threadIdx.x >= 64u) {
        // ---- gatherers.  Element e = g + FIT_GATHERERS u of a block is band eb[u] of its row er[u] (the same in
        //      every block); rows past the end of the list are clamped to its last one ----
        const uint32_t g = threadIdx.x - 64u;
        uint32_t er[FIT_STAGE_PER_THREAD], eb[FIT_STAGE_PER_THREAD];
        uint32_t ia[FIT_STAGE_PER_THREAD], ib[FIT_STAGE_PER_THREAD];      // row numbers: two register sets that
        double va[FIT_STAGE_PER_THREAD], vb[FIT_STAGE_PER_THREAD];       // swap roles every step (no copies:
#pragma unroll                                                           // a copy would wait for its loads)
        for (uint32_t u = 0; u < FIT_STAGE_PER_THREAD; u++) {
            uint32_t e = g + u * FIT_GATHERERS;
            if (e >= B * unb) e = B * unb - 1u;              // (stored twice with the same value)
            er[u] = e / unb; eb[u] = e - er[u] * unb;
        }
        auto load_idx = [&](uint32_t blk, uint32_t *idx) {
            if (blk >= nblk) return;
            const uint32_t q = q0 + blk * B;
            const uint32_t last = ((q1 - q < B) ? (q1 - q) : B) - 1u;
#pragma unroll
            for (uint32_t u = 0; u < FIT_STAGE_PER_THREAD; u++) idx[u] = rows[q + (er[u] < last ? er[u] : last)];
        };
        auto load_x = [&](uint32_t blk, const uint32_t *idx, double *v) {
            if (blk >= nblk) return;
#pragma unroll
            for (uint32_t u = 0; u < FIT_STAGE_PER_THREAD; u++) v[u] = X[(size_t)idx[u] * unb + eb[u]];
        };
        auto store = [&](uint32_t blk, const double *v) {
            if (blk >= nblk) return;
            double *dst = sx + (blk & 1u) * BUF;
#pragma unroll
            for (uint32_t u = 0; u < FIT_STAGE_PER_THREAD; u++) dst[eb[u] * pitch + er[u]] = v[u];
        };
        // block b's row numbers live in (b even ? ia : ib), its values in (b even ? va : vb).  (Loading the
        // values three blocks ahead instead of two changed nothing: once the E-step has swept the caches the
        // gatherers are bound by how fast ONE compute unit pulls 1 300 scattered lines per block, ~4.5 us
        // against 2.9 us of chain; the kernel takes 56 us with the sample in cache and 74 us behind a 1 GB write.)
        load_idx(0, ia); load_idx(1, ib);
        load_x(0, ia, va); load_x(1, ib, vb);
        load_idx(2, ia);
        store(0, va);
        __syncthreads();
        for (uint32_t s = 0; s < nblk; s += 2u) {
            load_x(s + 2u, ia, va);
            load_idx(s + 3u, ib);
            store(s + 1u, vb);
            __syncthreads();
            if (s + 1u >= nblk) break;
            load_x(s + 3u, ib, vb);
            load_idx(s + 4u, ia);
            store(s + 2u, va);
            __syncthreads();
        }
    } else {
        // ---- the adding wavefront ----
#ifdef FIT_SUM_DIAG
        unsigned long long tc = 0, tb = 0, t0 = __builtin_readcyclecounter();
#define FIT_DIAG_MARK(acc_t) { const unsigned long long t1 = __builtin_readcyclecounter(); acc_t += t1 - t0; t0 = t1; }
#else
#define FIT_DIAG_MARK(acc_t)
#endif
        // lane l feeds chain m = l & 15 (band m + 16 c of accumulator c) with row 4 i + (l >> 4) of step i
        const uint32_t m = threadIdx.x & 15u, kk = threadIdx.x >> 4;
        const uint32_t nch = (unb + 15u) / 16u;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        uint32_t boff[4];                                    // chains past nb shadow band 0 (results unused)
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++) boff[c] = ((m + 16u * c < unb) ? (m + 16u * c) : 0u) * pitch + kk;
        __syncthreads();
        FIT_DIAG_MARK(tb)
        for (uint32_t s = 0; s < nblk; s++) {
            const uint32_t q = q0 + s * B;
            const uint32_t nrows = (q1 - q < B) ? (q1 - q) : B;
            const double *buf = sx + (s & 1u) * BUF;
#pragma unroll
            for (uint32_t c = 0; c < 4u; c++) {
                if (c >= nch) break;
                const double *p = buf + boff[c];
                uint32_t r = 0;
                if (nrows >= 32u) {
                    acc[c] = fit_chain_mfma32(acc[c], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double *)p,
                                              nrows / 32u);
                    r = nrows & ~31u;
                }
                for (; r + 4u <= nrows; r += 4u) acc[c] = fit_mfma4(p[r], acc[c]);
                if (r < nrows) acc[c] = fit_mfma4(r + kk < nrows ? p[r] : -0.0, acc[c]);
            }
            FIT_DIAG_MARK(tc)
            __syncthreads();
            FIT_DIAG_MARK(tb)
        }
#ifdef FIT_SUM_DIAG
        if (threadIdx.x == 0) { g_fit_diag[j * 4] = tc; g_fit_diag[j * 4 + 1] = tb; g_fit_diag[j * 4 + 2] = nblk; g_fit_diag[j * 4 + 3] = q1 - q0; }
#endif
        // chain m's sum stands in lanes 16 (m & 3) + 4 (m >> 2) + {0..3}
        if ((threadIdx.x & 3u) == 0u) {
            const uint32_t mo = (threadIdx.x >> 4) + 4u * ((threadIdx.x >> 2) & 3u);
#pragma unroll
            for (uint32_t c = 0; c < 4u; c++)
                if (mo + 16u * c < unb) S[(size_t)j * unb + mo + 16u * c] = acc[c];
        }
    }
    if (ep.ctl) {
        // the last workgroup to finish ends the iteration (release / acquire at agent scope: gridbar.h)
        __shared__ uint32_t s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t arrived = __hip_atomic_fetch_add(ep.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            const uint32_t last = arrived == gridDim.x;
            if (last) __hip_atomic_store(ep.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = last;
        }
        __syncthreads();
        if (s_last) elk_update_body(sx, FIT_SUM_THREADS, ep, S, cnt, (int)gridDim.x, nb);
    }
}

// The end of an iteration in one workgroup, unless a cluster came out empty (left to the host:
// fit_mstep_tail): centres = sums * (1 / weight), the centres' shifts and their squared sum, the new
// centres' half distances and nearest-centre distances, sklearn's two convergence tests.  Every float64
// operation and its order are those of fit_mstep_tail / elk_half_distances.  ep.C: the old centres in, the
// new ones out.  lds: 2 k nb + k k + 3 k doubles of LDS (the tables of the tail never touch global memory
// between its phases), or nullptr: S, ep.C, ep.half and ep.scratch stand in (k, nb beyond the LDS form).
__device__ void elk_update_body(double *lds, uint32_t nthreads, const ElkEpilogue &ep, double *S, const double *cnt, int k, int nb)
{
    ElkCtl *ctl = ep.ctl;
    const uint32_t it = ep.it;
    if (ctl->stop) return;
    if (threadIdx.x == 0) ctl->nd[(it + 1u) & 1u] = 0u;          // the next E-step's change counter
    __shared__ int s_empty;
    const int kn = k * nb;
    if (threadIdx.x == 0) s_empty = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += nthreads) if (cnt[j] == 0.0) s_empty = 1;
    __syncthreads();
    if (s_empty) {
        if (threadIdx.x == 0) ctl->stop = 2u;
        return;
    }
    double *Sn = lds, *Co = lds + kn, *hf = Co + kn, *sq = hf + (size_t)k * k, *xx = sq + k, *cs = xx + k;
    if (!lds) { Sn = S; Co = ep.C; hf = ep.half; sq = ep.scratch; xx = sq + k; cs = xx + k; }
    for (int t = threadIdx.x; t < kn; t += nthreads) {
        const double alpha = 1.0 / cnt[t / nb];
        Sn[t] = S[t] * alpha;                                // the new centres
        if (lds) Co[t] = ep.C[t];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += nthreads) {
        const double *a = &Sn[j * nb], *c = &Co[j * nb];
        double r = 0.0;
        int b = 0;
        for (; b + 4 <= nb; b += 4)
            r += ((a[b] - c[b]) * (a[b] - c[b]) + (a[b + 1] - c[b + 1]) * (a[b + 1] - c[b + 1]) +
                  (a[b + 2] - c[b + 2]) * (a[b + 2] - c[b + 2]) + (a[b + 3] - c[b + 3]) * (a[b + 3] - c[b + 3]));
        for (; b < nb; b++) r += (a[b] - c[b]) * (a[b] - c[b]);
        const double sh = __builtin_sqrt(r);
        cs[j] = sh;
        sq[j] = sh * sh;
        xx[j] = kmeans_sqnorm(a, nb);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < k * k; e += nthreads) {
        const int a = e / k, b = e - a * k;
        double d = 0.0;
        for (int t = 0; t < nb; t++) d = d + Sn[(size_t)a * nb + t] * Sn[(size_t)b * nb + t];
        double v = -2.0 * d;
        v = v + xx[a];
        v = v + xx[b];
        if (!(v > 0.0)) v = 0.0;
        if (a == b) v = 0.0;
        const double h = __builtin_sqrt(v) / 2.0;
        hf[e] = h;
        if (lds) ep.half[e] = h;
    }
    __syncthreads();                                         // (without LDS Co IS ep.C: every shift has been taken)
    for (int t = threadIdx.x; t < kn; t += nthreads) { if (lds) S[t] = Sn[t]; ep.C[t] = Sn[t]; }
    for (int j = threadIdx.x; j < k; j += nthreads) ep.cshift[j] = cs[j];
    if (ep.hist.cs) {           // fit_bounds.h: this iteration's shifts, their running sums, the next E-step's centres
        for (int j = threadIdx.x; j < k; j += nthreads) {
            ep.hist.cs[(size_t)it * k + j] = cs[j];
            ep.hist.csT[(size_t)j * ep.hist.rows + it] = cs[j];
            ep.hist.cum[((size_t)it + 1u) * k + j] = ep.hist.cum[(size_t)it * k + j] + cs[j];
        }
        for (int t = threadIdx.x; t < kn; t += nthreads) ep.hist.cen[((size_t)it + 1u) * kn + t] = Sn[t];
    }
    __syncthreads();
    for (int l = threadIdx.x; l < k; l += nthreads) {
        double m0 = hf[l], m1 = -1.0;
        for (int a = 1; a < k; a++) {
            const double v = hf[(size_t)a * k + l];
            if (v < m0) { m1 = m0; m0 = v; }
            else if (m1 < 0.0 || v < m1) m1 = v;
        }
        ep.next[l] = k > 1 ? m1 : m0;
    }
    if (threadIdx.x == 0) {
        const double shift_tot = np_pairwise_sum(sq, (size_t)k);
        const uint32_t nd = ctl->nd[it & 1u];
        ctl->shift_tot = shift_tot;
        ctl->iters = it;
        if (it >= 2u && nd == 0u) ctl->stop = 1u;
        else if (shift_tot <= ep.tol) ctl->stop = 3u;
    }
}
#define ELK_UPDATE_LDS_DOUBLES(k, nb) ((size_t)2 * (k) * (nb) + (size_t)(k) * (k) + 3 * (size_t)(k))
// the same as a kernel of its own (k, nb beyond k_fit_sum_lists_staged; tables in dynamic LDS)
__global__ __launch_bounds__(1024) void k_elk_update(double *__restrict__ S, const double *__restrict__ cnt,
                                                    int k, int nb, ElkEpilogue ep)
{
    extern __shared__ double sh_upd[];
    elk_update_body(ep.scratch ? (double *)nullptr : sh_upd, 1024u, ep, S, cnt, k, nb);
}

#ifndef ELK_BATCH
#define ELK_BATCH 8                 // iterations enqueued between two host synchronisations
#endif

// The E-step sharded by sample rows over the ranks of a communicator (the one part of the fit that
// shards; sklearn's fit itself is one process, shepseg.py:305-312).  Rank r runs init / filter / visit on rows
// [r ns, (r + 1) ns) -- its own bounds, stamps, upper bounds; nothing of them ever leaves the rank -- then the
// labels are all-gathered IN PLACE on the fit's stream (ncclAllGather of ns int32 per rank, no host round
// trip), and every rank runs the M-step on the full label array: the same sort, the same row-order sums, the
// same tail on the same bits, so the centres agree without a broadcast.  The count of changed labels comes
// from comparing the gathered labels with the previous iteration's, on every rank alike.
//   nc == nullptr with world > 1: ONE process plays all the ranks in turn (tests of the shard arithmetic on a
// one-GPU box: SHEPSEG_FIT_SHARDS); only == r: it plays rank r alone (the timing of one rank's share:
// SHEPSEG_FIT_SHARD_ONLY; the other shards' labels go stale, the result is not a fit).
struct FitShard {
    int rank = 0, world = 1;
    void *nc = nullptr;             // ncclComm_t
    int only = -1;
};
__global__ __launch_bounds__(256) void k_fit_count_diff(const int32_t *__restrict__ a, const int32_t *__restrict__ b,
                                                        uint32_t n, uint32_t *count, const uint32_t *stop)
{
    if (stop && *stop) return;
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) c += a[i] != b[i];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane_id() == 0 && c) atomicAdd(count, c);
}

// The faithful path.  dX: the centred sample on the device (n rows of nb); X: the same on the host
// through Xat; C: the centred initial centres in, the final centred centres out; dlab: n labels out.
// Iterations run in batches of ELK_BATCH without a host round trip (k_elk_update owns the convergence
// tests; once it raises ctl->stop the kernels still queued behind it return at once); an iteration that
// leaves a cluster empty is finished on the host.
template <class XAt>
static int run_fit_elkan(shp_ctx *ctx, const double *dX, XAt Xat, uint32_t n, int nb, int k,
                         std::vector<double> &C, int max_iter, double tol, int32_t *dlab, double *ddist,
                         int *n_iter_out, FitShard shard = FitShard(), int32_t *dlab_prev = nullptr)
{
    const int kn = k * nb;
    hipStream_t st = ctx->stream;
    // k <= 64: no table of exact bounds (fit_bounds.h): n x k float32 brackets + n x k 2-byte stamps + the upper
    // bounds, and the history of shifts / their running sums / centres per iteration.  SHEPSEG_ELK_TABLE=1: the
    // round-3 form (the exact table, cluster-major), which k > 64 always takes.
    const bool lazy = k <= 64 && nb <= 64 && max_iter < 65000 &&
                      !(getenv("SHEPSEG_ELK_TABLE") && atoi(getenv("SHEPSEG_ELK_TABLE")) != 0);
    if (shard.world > 1 && (!lazy || !dlab_prev || shard.world > 64)) shard = FitShard();    // (the exact-table form is not sharded: every rank runs it whole)
    const bool sharded = shard.world > 1;
    const uint32_t ns = sharded ? (n + (uint32_t)shard.world - 1u) / (uint32_t)shard.world : n;      // rows per rank
    const size_t hist_rows = (size_t)max_iter + 3;
    const size_t hist_doubles = lazy ? hist_rows * ((size_t)3 * k + kn) : 0;
    if (lazy) CHK(buf_ensure(ctx, ctx->fit_lb, (size_t)n * k * 6 + 512 + (size_t)n * 8 * 3 + hist_doubles * 8 + 64));
    else CHK(buf_ensure(ctx, ctx->fit_lb, ((size_t)k * n + n) * 8));
    // small device block: C | cshift | half | next | S | cnt | scratch (2k) | ctl | off (k + 1)
    const size_t small_doubles = (size_t)kn + k + (size_t)k * k + k + kn + k + 3 * (size_t)k;
    CHK(buf_ensure(ctx, ctx->fit_part, small_doubles * 8 + sizeof(ElkCtl) + ((size_t)k + 4) * 4 + 64));
    double *dlb = bp<double>(ctx->fit_lb), *dub = dlb + (size_t)k * n;
    float *dA = nullptr;
    uint16_t *dstamps = nullptr;
    ElkHist hist{nullptr, nullptr, nullptr, nullptr, 0u};
    unsigned long long *ddiag = nullptr, *dcmask = nullptr, *dmmask = nullptr;
    if (lazy) {
        // ub | cmask | mmask | hist (cs, cum, cen) | diag | A | stamps
        dub = bp<double>(ctx->fit_lb);
        dcmask = (unsigned long long *)(dub + n);
        dmmask = dcmask + n;
        hist.cs = (double *)(dmmask + n);
        hist.cum = hist.cs + hist_rows * k;
        hist.cen = hist.cum + hist_rows * k;
        hist.csT = hist.cen + hist_rows * kn;
        hist.rows = (uint32_t)hist_rows;
        ddiag = (unsigned long long *)(hist.csT + hist_rows * k);
        dA = (float *)(((uintptr_t)(ddiag + 8) + 255u) & ~(uintptr_t)255u);
        dstamps = (uint16_t *)(dA + (size_t)n * k);
        dlb = nullptr;
    }
    double *dC = bp<double>(ctx->fit_part), *dcshift = dC + kn, *dhalf = dcshift + k, *dnext = dhalf + (size_t)k * k;
    double *dS = dnext + k, *dcnt = dS + kn, *dscr = dcnt + k;
    ElkCtl *dctl = (ElkCtl *)(dscr + 3 * (size_t)k);
    uint32_t *doff = (uint32_t *)(dctl + 1);
    uint32_t *dstop = &dctl->stop;
    // host staging (pinned when it fits the context's block): C | cshift | half | next up, S | cnt down, ctl
    const size_t up_doubles = (size_t)kn + k + (size_t)k * k + k, dn_doubles = (size_t)kn + k;
    std::vector<double> pageable;
    double *stage = (double *)ctx->h_pinned;
    const bool pinned = (up_doubles + dn_doubles) * 8 + sizeof(ElkCtl) + 256 <= (size_t)PIN_MIRROR * 4u;
    if (!pinned) { pageable.resize(up_doubles + dn_doubles + 8); stage = pageable.data(); }
    double *h_up = stage, *h_dn = stage + up_doubles;
    ElkCtl *h_ctl = (ElkCtl *)(h_dn + dn_doubles);
    std::vector<double> half((size_t)k * k), next(k), cshift(k, 0.0), Cn(kn), w(k);
    auto upload = [&]() -> int {                    // (the stream is idle whenever this runs)
        memcpy(h_up, C.data(), (size_t)kn * 8);
        memcpy(h_up + kn, cshift.data(), (size_t)k * 8);
        memcpy(h_up + kn + k, half.data(), (size_t)k * k * 8);
        memcpy(h_up + kn + k + (size_t)k * k, next.data(), (size_t)k * 8);
        HIPCHK(ctx, hipMemcpyAsync(dC, h_up, up_doubles * 8, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return 0;
    };
    auto upload_ctl = [&](uint32_t iters, uint32_t nd0, uint32_t nd1) -> int {
        memset(h_ctl, 0, sizeof(ElkCtl));
        h_ctl->iters = iters; h_ctl->nd[0] = nd0; h_ctl->nd[1] = nd1;
        HIPCHK(ctx, hipMemcpyAsync(dctl, h_ctl, sizeof(ElkCtl), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return 0;
    };
    HIPCHK(ctx, hipStreamSynchronize(st));          // earlier users of the pinned block are done
    elk_half_distances(C.data(), k, nb, half.data(), next.data());
    CHK(upload());                                  // (cshift = 0: the first E-step's bounds update changes nothing)
    CHK(upload_ctl(0, 0, 0));
    const unsigned g = grid_for(n, 256);
    const bool want_diag = lazy && getenv("SHEPSEG_FIT_TRACE");
    // k_elk2_filter: shifts, nearest-centre distances (float64) and the half distances (float32) in LDS;
    // k_elk2_visit: the half distances and the centres (float64), the shifts' running sums, the list of a chunk's visits
    const size_t lds_f = (size_t)2 * k * 8 + (size_t)k * k * 4, lds2 = ((size_t)k * k + kn + k) * 8 + ELK2_VCHUNK * 4;
    // the pixel-band count as a template parameter where it is small (the sample row then lives in registers)
    auto visit2 = nb == 1 ? k_elk2_visit<1> : nb == 2 ? k_elk2_visit<2> : nb == 3 ? k_elk2_visit<3> : nb == 4 ? k_elk2_visit<4> :
                  nb == 5 ? k_elk2_visit<5> : nb == 6 ? k_elk2_visit<6> : nb == 7 ? k_elk2_visit<7> : nb == 8 ? k_elk2_visit<8> :
                  nb == 10 ? k_elk2_visit<10> : nb == 12 ? k_elk2_visit<12> : k_elk2_visit<0>;
    unsigned g2 = g, gf = g, gv = (unsigned)grid_for(n, ELK2_VCHUNK);
    if (lazy) {
        // workgroups stride over chunks of samples: as many as stay resident
        int dev = 0, ncu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        auto resident = [&](const void *fn, size_t lds) -> unsigned {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                per_cu = 2;
            }
            return (unsigned)(ncu * per_cu);
        };
        const unsigned cap_f = resident((const void *)k_elk2_filter<0>, lds_f), cap_v = resident((const void *)visit2, lds2);
        const unsigned cap_i = resident((const void *)k_elk2_init, ((size_t)k * k + kn) * 8);
        if (gf > cap_f) gf = cap_f;
        if (gv > cap_v) gv = cap_v;
        if (g2 > cap_i) g2 = cap_i;
        // cum[1] = 0; cen[1] = the initial centres (dC, uploaded above)
        HIPCHK(ctx, hipMemsetAsync(hist.cum, 0, (size_t)2 * k * 8, st));
        HIPCHK(ctx, hipMemsetAsync(ddiag, 0, 64, st));
        HIPCHK(ctx, hipMemcpyAsync(hist.cen + kn, dC, (size_t)kn * 8, hipMemcpyDeviceToDevice, st));
        // (a shard = the same kernels on pointers moved to the shard's first row)
        for (int r = 0; r < shard.world; r++) {
            if (sharded && ((shard.nc && r != shard.rank) || (shard.only >= 0 && r != shard.only))) continue;
            const uint32_t i0 = sharded ? (uint32_t)r * ns : 0u;
            if (i0 >= n) continue;
            const uint32_t m = sharded ? (n - i0 < ns ? n - i0 : ns) : n;
            hipLaunchKernelGGL(k_elk2_init, dim3(g2), dim3(256), ((size_t)k * k + kn) * 8, st, dX + (size_t)i0 * nb, m, nb, dC, k, dhalf,
                               dlab + i0, dub + i0, dA + (size_t)i0 * k, dstamps + (size_t)i0 * k); KCHK(ctx);
        }
        if (sharded) HIPCHK(ctx, hipMemcpyAsync(dlab_prev, dlab, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    } else {
        HIPCHK(ctx, hipMemsetAsync(dlb, 0, (size_t)k * n * 8, st));
        hipLaunchKernelGGL(k_elk_init, dim3(g), dim3(256), 0, st, dX, n, nb, dC, k, dhalf, dlab, dub, dlb); KCHK(ctx);
    }
    auto estep = [&](uint32_t *nd, const uint32_t *stop, int it) -> int {
        if (lazy) {
            const double *csp = it > 1 ? hist.cs + (size_t)(it - 1) * k : (const double *)nullptr;
            for (int r = 0; r < shard.world; r++) {
                if (sharded && ((shard.nc && r != shard.rank) || (shard.only >= 0 && r != shard.only))) continue;
                const uint32_t i0 = sharded ? (uint32_t)r * ns : 0u;
                if (i0 >= n) continue;
                const uint32_t m = sharded ? (n - i0 < ns ? n - i0 : ns) : n;
                unsigned gfr = (unsigned)grid_for(m, 256), gvr = (unsigned)grid_for(m, ELK2_VCHUNK);
                if (gfr > gf) gfr = gf;
                if (gvr > gv) gvr = gv;
                hipLaunchKernelGGL(k_elk2_filter<0>, dim3(gfr), dim3(256), lds_f, st, m, k, dhalf, dnext, csp, hist.cum + (size_t)it * k,
                                   dlab + i0, dub + i0, dA + (size_t)i0 * k, dcmask + i0, dmmask + i0, stop); KCHK(ctx);
                hipLaunchKernelGGL(visit2, dim3(gvr), dim3(256), lds2, st, dX + (size_t)i0 * nb, m, nb, hist.cen + (size_t)it * kn, k, dhalf,
                                   hist.cum + (size_t)it * k, dlab + i0, dub + i0, dA + (size_t)i0 * k, dstamps + (size_t)i0 * k,
                                   dcmask + i0, dmmask + i0, hist, (uint32_t)it, sharded ? (uint32_t *)(ddiag + 6) : nd, stop,
                                   want_diag ? ddiag : (unsigned long long *)nullptr);
            }
            if (sharded) {
                // every rank's labels on every rank (in place, on this stream), then the changed-label count from the
                // labels themselves -- the same on all ranks
                if (shard.nc)
                    NCCLCHK(ctx, ncclAllGather(dlab + (size_t)shard.rank * ns, dlab, ns, ncclInt32, (ncclComm_t)shard.nc, st));
                hipLaunchKernelGGL(k_fit_count_diff, dim3(256), dim3(256), 0, st, dlab, dlab_prev, n, nd, stop); KCHK(ctx);
                HIPCHK(ctx, hipMemcpyAsync(dlab_prev, dlab, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            }
        }
        else if (k <= 64)
            hipLaunchKernelGGL(k_elk_estep64, dim3(g), dim3(256), 0, st, dX, n, nb, dC, k, dhalf, dnext, dcshift, dlab,
                               dub, dlb, nd, stop);
        else
            hipLaunchKernelGGL(k_elk_estep, dim3(g), dim3(256), 0, st, dX, n, nb, dC, k, dhalf, dnext, dcshift, dlab,
                               dub, dlb, nd, stop);
        KCHK(ctx);
        return 0;
    };
    // the end of an iteration: the last workgroup of the staged sums (nb <= 64, tables in its LDS), else k_elk_update
    const bool upd_lds = ELK_UPDATE_LDS_DOUBLES(k, nb) * 8 <= 64 * 1024;
    const bool fused = nb <= 64 && ELK_UPDATE_LDS_DOUBLES(k, nb) <= 2u * (FIT_STAGE_DOUBLES + 4u * 64u) &&
                       !(getenv("SHEPSEG_ELK_UNFUSED") && atoi(getenv("SHEPSEG_ELK_UNFUSED")) != 0);
    uint32_t *ddone = doff + (size_t)k + 1;
    HIPCHK(ctx, hipMemsetAsync(ddone, 0, 4, st));
    auto ep_of = [&](int it, bool on) -> ElkEpilogue {
        ElkEpilogue ep;
        ep.ctl = on ? dctl : nullptr; ep.done = ddone; ep.C = dC; ep.cshift = dcshift; ep.half = dhalf; ep.next = dnext;
        ep.scratch = (fused || upd_lds) ? nullptr : dscr; ep.tol = tol; ep.it = (uint32_t)it; ep.hist = hist;
        return ep;
    };
    const bool check_digits = getenv("SHEPSEG_FIT_CHECK_DIGITS") && atoi(getenv("SHEPSEG_FIT_CHECK_DIGITS")) != 0;
    uint32_t h_ctl_stop_seen = 0u;          // (the check above only judges iterations that really ran)
    bool strict = false, finished = false;
    int it_done = 0;
    while (it_done < max_iter && !finished) {
        const int b_end = check_digits ? it_done + 1 : (it_done + ELK_BATCH < max_iter ? it_done + ELK_BATCH : max_iter);
        for (int it = it_done + 1; it <= b_end; it++) {
            CHK(estep(&dctl->nd[it & 1], dstop, it));
            // row lists: the row numbers sorted stably by label
            uint32_t *ks = nullptr, *rows = nullptr;
            SortDigits sd;
            CHK(sort_pairs(ctx, (const uint32_t *)dlab, nullptr, n, bits_for((uint32_t)(k - 1)), &ks, &rows, false, &sd));
            // with one radix pass (k <= 256) the clusters' ranges are in the sort's own scanned histogram:
            // no search kernel (k_elk_update zeroes the next E-step's counter)
            FitDigits dg{nullptr, nullptr, 0u, n};
            if (sd.passes == 1) { dg.hscan = sd.hscan; dg.boff = sd.boff; dg.nblk = sd.nblk; }
            else {
                hipLaunchKernelGGL(k_elk_offsets, dim3(grid_for((size_t)k + 1, 256)), dim3(256), 0, st, ks, n, k, doff,
                                   &dctl->nd[(it + 1) & 1], dstop); KCHK(ctx);
            }
            if (sd.passes == 1 && check_digits) {
                // the ranges read off the sort's scanned histogram against a count of the labels themselves
                std::vector<uint32_t> got((size_t)k + 1);
                std::vector<int32_t> hl(n);
                hipLaunchKernelGGL(k_fit_digit_starts, dim3(1), dim3(256), 0, st, dg, k, doff); KCHK(ctx);
                HIPCHK(ctx, hipMemcpyAsync(got.data(), doff, ((size_t)k + 1) * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipMemcpyAsync(hl.data(), dlab, (size_t)n * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
                if (h_ctl_stop_seen == 0u) {
                    std::vector<uint32_t> want((size_t)k + 1, 0u);
                    for (uint32_t i = 0; i < n; i++) if ((uint32_t)hl[i] < (uint32_t)k) want[(size_t)hl[i] + 1]++;
                    for (int j = 0; j < k; j++) want[(size_t)j + 1] += want[j];
                    for (int j = 0; j <= k; j++)
                        if (got[j] != want[j])
                            SHP_FAIL(ctx, SHP_ERR_STATE, "fit: cluster %d's row list starts at %u in the sort's histogram, %u by count "
                                     "(the layout sort_digit_start assumes has changed)", j, got[j], want[j]);
                }
            }
            if (nb <= 64)
                hipLaunchKernelGGL(k_fit_sum_lists_staged, dim3(k), dim3(FIT_SUM_THREADS), 0, st, dX, nb, rows, doff, dS, dcnt, dstop, dg,
                                   ep_of(it, fused));
            else
                hipLaunchKernelGGL(k_fit_sum_lists, dim3(k, (nb + 63) / 64), dim3(64), 0, st, dX, nb, rows, doff, dS, dcnt,
                                   dstop, dg);
            KCHK(ctx);
            if (!fused) {
                hipLaunchKernelGGL(k_elk_update, dim3(1), dim3(1024), upd_lds ? ELK_UPDATE_LDS_DOUBLES(k, nb) * 8 : 0, st, dS, dcnt, k, nb,
                                   ep_of(it, true)); KCHK(ctx);
            }
        }
        HIPCHK(ctx, hipMemcpyAsync(h_ctl, dctl, sizeof(ElkCtl), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        const uint32_t stop = h_ctl->stop;
        h_ctl_stop_seen = stop;
        if (getenv("SHEPSEG_FIT_TRACE")) {
            std::vector<double> hs(k);
            HIPCHK(ctx, hipMemcpyAsync(hs.data(), dcshift, (size_t)k * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            int nz = 0;
            for (int j = 0; j < k; j++) nz += hs[j] == 0.0;
            fprintf(stderr, "elkan batch to %d: stop %u iters %u labels changed %u / %u shift %.17g, %d of %d centres did not move\n",
                    b_end, stop, h_ctl->iters, h_ctl->nd[0], h_ctl->nd[1], h_ctl->shift_tot, nz, k);
            if (want_diag) {
                unsigned long long hd[3];
                HIPCHK(ctx, hipMemcpyAsync(hd, ddiag, sizeof(hd), hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
                fprintf(stderr, "   since the start: %llu rows read, %llu samples visited, %llu comparisons recomputed exactly (n = %u)\n",
                        hd[0], hd[1], hd[2], n);
            }
        }
        if (stop == 0u) { it_done = b_end; continue; }
        if (stop == 1u || stop == 3u) {
            it_done = (int)h_ctl->iters;
            strict = stop == 1u;
            finished = true;
            break;
        }
        // stop == 2: iteration `it` left a cluster empty after its E-step and sums; the host finishes it
        const int it = (int)h_ctl->iters + 1;
        const uint32_t nd = h_ctl->nd[it & 1], nd_other = h_ctl->nd[(it + 1) & 1];
        HIPCHK(ctx, hipMemcpyAsync(h_dn, dS, dn_doubles * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(h_up, dC, (size_t)kn * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        memcpy(Cn.data(), h_dn, (size_t)kn * 8);
        memcpy(w.data(), h_dn + kn, (size_t)k * 8);
        memcpy(C.data(), h_up, (size_t)kn * 8);             // the centres this iteration started from
        double shift_tot = 0.0;
        auto fetch = [&](std::vector<double> &dist, std::vector<int32_t> &hl) -> int {
            hipLaunchKernelGGL(k_fit_dist, dim3(g), dim3(256), 0, st, dX, n, nb, dlab, dC, ddist); KCHK(ctx);
            dist.resize(n); hl.resize(n);
            HIPCHK(ctx, hipMemcpyAsync(dist.data(), ddist, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(hl.data(), dlab, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            return 0;
        };
        CHK(fit_mstep_tail(k, nb, n, Cn, w, C, Xat, fetch, cshift, &shift_tot));
        elk_half_distances(Cn.data(), k, nb, half.data(), next.data());
        C = Cn;
        CHK(upload());
        if (lazy) {             // this iteration's shifts, their running sums and the next E-step's centres (k_elk_update's part)
            std::vector<double> row(k);
            HIPCHK(ctx, hipMemcpyAsync(row.data(), hist.cum + (size_t)it * k, (size_t)k * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            for (int j = 0; j < k; j++) row[j] = row[j] + cshift[j];
            HIPCHK(ctx, hipMemcpyAsync(hist.cum + ((size_t)it + 1) * k, row.data(), (size_t)k * 8, hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipMemcpyAsync(hist.cs + (size_t)it * k, cshift.data(), (size_t)k * 8, hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipMemcpy2DAsync(hist.csT + it, hist_rows * 8, cshift.data(), 8, 8, (size_t)k, hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipMemcpyAsync(hist.cen + ((size_t)it + 1) * kn, C.data(), (size_t)kn * 8, hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
        }
        // (the counter the next E-step adds to was zeroed by this iteration's k_elk_offsets)
        CHK(upload_ctl((uint32_t)it, (it & 1) ? nd_other : nd, (it & 1) ? nd : nd_other));
        it_done = it;
        if (it >= 2 && nd == 0u) { strict = true; finished = true; }
        else if (shift_tot <= tol) finished = true;
    }
    if (!strict) CHK(estep(&dctl->nd[0], nullptr, it_done + 1));
    // the final centres
    HIPCHK(ctx, hipMemcpyAsync(h_up, dC, (size_t)kn * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    memcpy(C.data(), h_up, (size_t)kn * 8);
    *n_iter_out = it_done;
    return 0;
}
