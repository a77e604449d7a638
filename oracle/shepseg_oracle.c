/*
 * shepseg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the pyshepseg per-tile hot path
 * (ubarsc/pyshepseg v2.0.3).  It exists so that the HIP kernels in
 * pyshepseg_amd/csrc can be checked bit-for-bit on the GPU box, where the
 * Python/numba reference cannot travel.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path never does.
 *
 * Parity status: PINNED.  Every function below is checked against the
 * unmodified reference run under real numba 0.54.1 / sklearn 0.24.2 in the
 * build container (oracle/refgen/gen_golden*.py -> tests/golden/, and
 * oracle/refgen/fuzz_vs_reference.py: 400 + 300 random cases, every stage and
 * the k-means fit bit for bit).  The fit the reference runs is ELKAN's
 * k-means (sklearn 0.24.2 algorithm="auto"): orc_kmeans_fit_elkan, pinned on
 * top by oracle/refgen/probe_elkan.py with one OpenMP thread -- the only
 * reproducible setting of the reference itself.  orc_kmeans_fit /
 * orc_kmeans_fit_assoc restate Lloyd's algorithm (what the HIP fit runs while no
 * label hangs on a tie) and are pinned through it: same partitions, iteration
 * count and, with the row-order sums, centres as Elkan's wherever no sample is
 * equidistant from two centres.
 *
 * Each function cites the reference file:line it restates.  Numeric typing
 * follows SURVEY.md section 8(a0) (facts N1..N14, established by running the
 * reference, not by reading it).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))
#ifndef ORC_SKLEARN_GE_1
#define ORC_SKLEARN_GE_1 0     /* 0: sklearn 0.24.2 (pinned oracle stack); 1: sklearn >= 1.0 variants */
#endif
#define ORC_RELOCATE_GUARD ORC_SKLEARN_GE_1

enum { ORC_U8 = 0, ORC_I16 = 1, ORC_U16 = 2, ORC_I32 = 3, ORC_U32 = 4 };

/* ------------------------------------------------------------------ */
/* generic pixel fetch: all supported image dtypes fit in int64        */
/* ------------------------------------------------------------------ */
static inline int64_t px_get(const void *img, int dtype, size_t i)
{
    switch (dtype) {
    case ORC_U8:  return ((const uint8_t  *)img)[i];
    case ORC_I16: return ((const int16_t  *)img)[i];
    case ORC_U16: return ((const uint16_t *)img)[i];
    case ORC_I32: return ((const int32_t  *)img)[i];
    default:      return ((const uint32_t *)img)[i];
    }
}

ORC_API int orc_version(void) { return 1; }

/* ------------------------------------------------------------------ */
/* synthimg v1 -- integer-only synthetic image   */
/* ------------------------------------------------------------------ */
static inline uint64_t syn_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t syn_h(uint64_t seed, uint64_t b, uint64_t o, uint64_t y, uint64_t x)
{
    const uint64_t P = 1000003ULL;
    return syn_mix((((seed * P + b) * P + o) * P + y) * P + x);
}
static inline uint16_t syn_pixel(uint64_t seed, uint64_t b, uint64_t y, uint64_t x)
{
    static const int lgs[4] = {8, 6, 4, 2};
    static const uint64_t amps[4] = {2000, 1000, 500, 250};
    uint64_t v = 1000 + 300 * b;
    for (int o = 0; o < 4; o++) {
        int lg = lgs[o];
        uint64_t c = 1ULL << lg;
        uint64_t gy = y >> lg, gx = x >> lg, fy = y & (c - 1), fx = x & (c - 1);
        uint64_t v00 = syn_h(seed, b, o, gy, gx) >> 48;
        uint64_t v01 = syn_h(seed, b, o, gy, gx + 1) >> 48;
        uint64_t v10 = syn_h(seed, b, o, gy + 1, gx) >> 48;
        uint64_t v11 = syn_h(seed, b, o, gy + 1, gx + 1) >> 48;
        uint64_t interp = (v00 * (c - fy) * (c - fx) + v01 * (c - fy) * fx +
                           v10 * fy * (c - fx) + v11 * fy * fx) >> (2 * lg);
        v += (interp * amps[o]) >> 16;
    }
    v += ((syn_h(seed, b, 99, y, x) >> 48) * 120) >> 16;
    if (v > 65534) v = 65534;
    return (uint16_t)v;
}
/* window [y0,y0+rows) x [x0,x0+cols) of the infinite synthetic image, band-planar */
ORC_API void orc_synthimg(uint64_t seed, int nbands, int64_t y0, int64_t x0,
                          int rows, int cols, uint16_t *out)
{
    for (int b = 0; b < nbands; b++)
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++)
                out[((size_t)b * rows + r) * cols + c] =
                    syn_pixel(seed, (uint64_t)b, (uint64_t)(y0 + r), (uint64_t)(x0 + c));
}

/* ------------------------------------------------------------------ */
/* k-means assign: shepseg.py:317-361 applySpectralClusters +           */
/* sklearn KMeans.predict (3rd party, sklearn 0.24.2                     */
/* sklearn/cluster/_kmeans.py _labels_inertia -> _k_means_lloyd.pyx      */
/* _update_chunk_dense):                                                 */
/*   pairwise = |c_j|^2 (row_norms = numpy einsum 'ij,ij->i'), then      */
/*   BLAS dgemm(alpha = -2, X, C^T, beta = 1) onto it, argmin, first     */
/*   minimum wins.                                                       */
/* Evaluation order, pinned against the oracle stack (numpy 1.26.4,      */
/* scipy-bundled OpenBLAS) by oracle/refgen/fit_probe.py:                */
/*   * |c|^2: einsum's baseline-SSE2 loop -- two lanes (even / odd       */
/*     elements), product and sum rounded separately, whole blocks of 8  */
/*     elements taken vector 3,2,1,0, the rest in order, lanes added at  */
/*     the end;                                                          */
/*   * nBands >= 2: the dot product is accumulated from zero by fused    */
/*     multiply-adds in band order, then d = |c|^2 + (-2 dot) (the       */
/*     scaling by -2 is exact, so operand -2c gives the same bits);      */
/*   * nBands == 1 (a rank-1 update): d = fma(x, -2c, |c|^2).            */
/* On integer imagery with well separated centres any float64 order     */
/* gives the same labels; on exact ties (few grey levels,   */
/* duplicated centres) only this one does.                               */
/* ------------------------------------------------------------------ */
static double orc_sqnorm(const double *c, int nb)
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
        double p = c[i] * c[i];
        double q = (count > 1) ? c[i + 1] * c[i + 1] : 0.0;
        a0 = p + a0;
        a1 = q + a1;
    }
    return a0 + a1;
}

/* distance term of one sample against one centre (m2c = -2c, cn = |c|^2) */
static inline double orc_dist(const double *x, const double *m2c, int nb, double cn)
{
    if (nb == 1) return fma(x[0], m2c[0], cn);
    double dot = 0.0;
    for (int b = 0; b < nb; b++) dot = fma(x[b], m2c[b], dot);
    return cn + dot;
}

ORC_API void orc_kmeans_prepare(const double *centres, int k, int nbands,
                                double *m2c /* k*nbands: -2*c */, double *cnorm /* k */)
{
    for (int j = 0; j < k; j++) {
        for (int b = 0; b < nbands; b++) m2c[j * nbands + b] = -2.0 * centres[j * nbands + b];
        cnorm[j] = orc_sqnorm(centres + (size_t)j * nbands, nbands);
    }
}

ORC_API int orc_kmeans_assign(const void *img, int dtype, int nbands, int nrows, int ncols,
                              const double *centres, int k, int has_null, int64_t null_val,
                              int32_t *clusters_out)
{
    size_t npix = (size_t)nrows * ncols;
    double *m2c = (double *)malloc(sizeof(double) * k * nbands);
    double *cnorm = (double *)malloc(sizeof(double) * k);
    double x[64];
    if (nbands > 64) return -1;
    orc_kmeans_prepare(centres, k, nbands, m2c, cnorm);
    for (size_t p = 0; p < npix; p++) {
        int isnull = 0;
        for (int b = 0; b < nbands; b++) {
            int64_t v = px_get(img, dtype, (size_t)b * npix + p);
            if (has_null && v == null_val) isnull = 1;      /* shepseg.py:357-359 any band */
            x[b] = (double)v;
        }
        int best = 0;
        double bestd = 0.0;
        for (int j = 0; j < k; j++) {
            double d = orc_dist(x, m2c + (size_t)j * nbands, nbands, cnorm[j]);
            if (j == 0 || d < bestd) { bestd = d; best = j; }
        }
        clusters_out[p] = isnull ? 0 : best + 1;             /* shepseg.py:356 */
    }
    free(m2c); free(cnorm);
    return 0;
}

/* ------------------------------------------------------------------ */
/* clump: shepseg.py:452-541.  Raster scan + explicit LIFO stack,       */
/* neighbours pushed cx outer / cy inner, labelled when pushed,          */
/* MAX_CLUMP_SIZE=10000 counted on pixels added after the seed (N9).     */
/* Returns next clump id (highest used + 1).                             */
/* ------------------------------------------------------------------ */
ORC_API uint32_t orc_clump(const int32_t *img, int nrows, int ncols, int32_t ignore_val,
                           int four_connected, uint32_t clump_id, uint32_t *out)
{
    const int MAX_CLUMP_SIZE = 10000;
    size_t npix = (size_t)nrows * ncols;
    uint32_t *stack = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (npix ? npix : 1));
    memset(out, 0, sizeof(uint32_t) * npix);
    for (int y = 0; y < nrows; y++) {
        for (int x = 0; x < ncols; x++) {
            size_t p = (size_t)y * ncols + x;
            if (img[p] != ignore_val && out[p] == 0) {
                int32_t val = img[p];
                int clump_size = 0;
                size_t sp = 0;
                stack[0] = (uint32_t)y; stack[1] = (uint32_t)x; sp = 1;
                out[p] = clump_id;
                while (sp > 0 && clump_size < MAX_CLUMP_SIZE) {
                    sp--;
                    int sy = (int)stack[2 * sp], sx = (int)stack[2 * sp + 1];
                    int tlx = sx - 1 < 0 ? 0 : sx - 1;
                    int tly = sy - 1 < 0 ? 0 : sy - 1;
                    int brx = sx + 1 > ncols - 1 ? ncols - 1 : sx + 1;
                    int bry = sy + 1 > nrows - 1 ? nrows - 1 : sy + 1;
                    for (int cx = tlx; cx <= brx; cx++) {
                        for (int cy = tly; cy <= bry; cy++) {
                            int connected = !four_connected || (cy == sy || cx == sx);
                            size_t q = (size_t)cy * ncols + cx;
                            if (connected && img[q] != ignore_val && out[q] == 0 && img[q] == val) {
                                out[q] = clump_id;
                                clump_size++;
                                stack[2 * sp] = (uint32_t)cy; stack[2 * sp + 1] = (uint32_t)cx;
                                sp++;
                            }
                        }
                    }
                }
                clump_id++;
            }
        }
    }
    free(stack);
    return clump_id;
}

/* makeSegSize: shepseg.py:544-569.  seg_size has max_seg_id+1 entries. */
ORC_API void orc_make_seg_size(const uint32_t *seg, size_t npix, uint32_t max_seg_id,
                               uint32_t *seg_size)
{
    memset(seg_size, 0, sizeof(uint32_t) * ((size_t)max_seg_id + 1));
    for (size_t p = 0; p < npix; p++) seg_size[seg[p]]++;
}

ORC_API uint32_t orc_seg_max(const uint32_t *seg, size_t npix)
{
    uint32_t m = 0;
    for (size_t p = 0; p < npix; p++) if (seg[p] > m) m = seg[p];
    return m;
}

/* relabelSegments: shepseg.py:739-777 */
static void relabel_segments(uint32_t *seg, size_t npix, const uint32_t *seg_size,
                             size_t nseg /* len(segSize) */, uint32_t min_seg_id)
{
    uint32_t *sub = (uint32_t *)calloc(nseg ? nseg : 1, sizeof(uint32_t));
    for (size_t k = (size_t)min_seg_id + 1; k < nseg; k++) {
        sub[k] = sub[k - 1];
        if (seg_size[k - 1] == 0) sub[k]++;
    }
    for (size_t p = 0; p < npix; p++) seg[p] -= sub[seg[p]];
    free(sub);
}

/* findNearestNeighbourPixel: shepseg.py:677-736 (N2; N3: scan order; N4).
 * N2 precisely: numba evaluates dSqr as the int64 sum of the squared EXACT differences, wrapping
 * modulo 2^64 (only 32-bit imagery can get there: |difference| >= 2^31.5 in a band, or the sum
 * over the bands), and the reference compares it as written -- `minDsqr < 0 or dSqr < minDsqr` --
 * so a sum that wrapped negative is taken and then counts as "unset" for the next candidate.
 * Checked against the reference on 8000 limit-value cases: oracle/refgen/probe_dist_32bit.py,
 * results/dist_probe_32bit.txt.  Unsigned arithmetic here: signed overflow is undefined in C. */
static int find_nearest_nbr(const void *img, int dtype, int nbands, int nrows, int ncols,
                            const uint32_t *seg, int i, int j, const uint32_t *seg_size,
                            int four, int *oi, int *oj)
{
    size_t npix = (size_t)nrows * ncols;
    int64_t min_d = -1;
    int ii = -1, jj = -1;
    int i0 = i - 1 < 0 ? 0 : i - 1, i1 = i + 1 > nrows - 1 ? nrows - 1 : i + 1;
    int j0 = j - 1 < 0 ? 0 : j - 1, j1 = j + 1 > ncols - 1 ? ncols - 1 : j + 1;
    for (int a = i0; a <= i1; a++)
        for (int b = j0; b <= j1; b++) {
            int connected = !four || (a == i || b == j);
            if (!connected) continue;
            uint32_t nb = seg[(size_t)a * ncols + b];
            if (seg_size[nb] > 1) {
                uint64_t du = 0;
                for (int k = 0; k < nbands; k++) {
                    uint64_t t = (uint64_t)(px_get(img, dtype, (size_t)k * npix + (size_t)i * ncols + j) -
                                            px_get(img, dtype, (size_t)k * npix + (size_t)a * ncols + b));
                    du += t * t;
                }
                int64_t d = (int64_t)du;
                if (min_d < 0 || d < min_d) { min_d = d; ii = a; jj = b; }
            }
        }
    *oi = ii; *oj = jj;
    return ii >= 0 && jj >= 0;
}

/* eliminateSinglePixels: shepseg.py:572-615 (+ mergeSinglePixels :618-674).
 * seg_size has max_seg_id+1 entries and is updated in place (not relabelled,
 * as in the reference).  Returns total number of pixels merged. */
ORC_API int64_t orc_eliminate_single_pixels(const void *img, int dtype, int nbands, int nrows,
                                            int ncols, uint32_t *seg, uint32_t *seg_size,
                                            uint32_t min_seg_id, uint32_t max_seg_id, int four)
{
    size_t npix = (size_t)nrows * ncols;
    uint32_t *elim = (uint32_t *)malloc(sizeof(uint32_t) * 3 * ((size_t)max_seg_id + 1));
    int64_t total = 0;
    for (;;) {
        size_t n = 0;
        for (int i = 0; i < nrows; i++)
            for (int j = 0; j < ncols; j++) {
                uint32_t s = seg[(size_t)i * ncols + j];
                if (seg_size[s] == 1) {
                    int ii, jj;
                    if (find_nearest_nbr(img, dtype, nbands, nrows, ncols, seg, i, j, seg_size,
                                         four, &ii, &jj)) {
                        elim[3 * n] = (uint32_t)i; elim[3 * n + 1] = (uint32_t)j;
                        elim[3 * n + 2] = seg[(size_t)ii * ncols + jj];
                        n++;
                    }
                }
            }
        for (size_t k = 0; k < n; k++) {
            size_t p = (size_t)elim[3 * k] * ncols + elim[3 * k + 1];
            uint32_t ns = elim[3 * k + 2], os = seg[p];
            seg[p] = ns; seg_size[os] = 0; seg_size[ns]++;
        }
        if (n == 0) break;
        total += (int64_t)n;
    }
    free(elim);
    relabel_segments(seg, npix, seg_size, (size_t)max_seg_id + 1, min_seg_id);
    return total;
}

/* buildSegmentSpectra: shepseg.py:780-813 (N5: float32, raster order; the
 * add is evaluated as numba does for the unified type of float32+pixel). */
static float *build_segment_spectra(const uint32_t *seg, const void *img, int dtype, int nbands,
                                    size_t npix, uint32_t max_seg_id)
{
    float *ss = (float *)calloc(((size_t)max_seg_id + 1) * nbands, sizeof(float));
    for (size_t p = 0; p < npix; p++) {
        uint32_t s = seg[p];
        for (int k = 0; k < nbands; k++) {
            float *a = &ss[(size_t)s * nbands + k];
            *a = (float)((double)*a + (double)px_get(img, dtype, (size_t)k * npix + p));
        }
    }
    return ss;
}

/* buildSegmentSpectra as a test entry point: spect_sum_out has (max_seg_id + 1) * nbands floats */
ORC_API void orc_build_segment_spectra(const uint32_t *seg, const void *img, int dtype, int nbands,
                                       int nrows, int ncols, uint32_t max_seg_id, float *spect_sum_out)
{
    float *ss = build_segment_spectra(seg, img, dtype, nbands, (size_t)nrows * ncols, max_seg_id);
    memcpy(spect_sum_out, ss, sizeof(float) * ((size_t)max_seg_id + 1) * nbands);
    free(ss);
}

/* makeSegmentLocations (shepseg.py:880-915) as a CSR: off_out[s] .. off_out[s + 1] are segment s's
 * entries of rc_out ((row, col) pairs, raster order), s = 0 .. max_seg_id; the null segment has
 * none (the reference's dict holds ids >= MINSEGID only, and skips null pixels) */
ORC_API void orc_segment_locations(const uint32_t *seg, int nrows, int ncols, uint32_t max_seg_id,
                                   uint32_t *off_out /* max_seg_id + 2 */, uint32_t *rc_out)
{
    size_t npix = (size_t)nrows * ncols;
    uint32_t *fill = (uint32_t *)calloc((size_t)max_seg_id + 2, sizeof(uint32_t));
    memset(off_out, 0, sizeof(uint32_t) * ((size_t)max_seg_id + 2));
    for (size_t p = 0; p < npix; p++) if (seg[p] != 0) off_out[seg[p] + 1]++;
    for (uint32_t s = 1; s <= max_seg_id + 1; s++) off_out[s] += off_out[s - 1];
    for (int r = 0; r < nrows; r++)
        for (int c = 0; c < ncols; c++) {
            uint32_t s = seg[(size_t)r * ncols + c];
            if (s == 0) continue;
            uint32_t at = off_out[s] + fill[s]++;
            rc_out[2 * (size_t)at] = (uint32_t)r; rc_out[2 * (size_t)at + 1] = (uint32_t)c;
        }
    free(fill);
}

typedef struct { uint32_t n; uint32_t *rc; } seg_loc_t;   /* RowColArray shepseg.py:816-870 */

/* findMergeSegment: shepseg.py:1003-1063 (N6: float32 means / float32 sequential
 * band sum; N7: first strict minimum in list order, ii outer, jj inner; N8: float64 threshold) */
static uint32_t find_merge_segment(uint32_t seg_id, const seg_loc_t *loc, const uint32_t *seg,
                                   const uint32_t *seg_size, const float *ss, int nbands,
                                   int nrows, int ncols, double max_spectral_diff, int four)
{
    uint32_t best = 0;
    double best_d = 0.0;
    const uint32_t *rc = loc[seg_id].rc;
    uint32_t npx = loc[seg_id].n;
    float spect[64], nbr[64];
    for (int k = 0; k < nbands; k++) spect[k] = ss[(size_t)seg_id * nbands + k] / (float)npx;
    for (uint32_t k = 0; k < npx; k++) {
        int i = (int)rc[2 * k], j = (int)rc[2 * k + 1];
        int i0 = i - 1 < 0 ? 0 : i - 1, i1 = i + 2 > nrows ? nrows : i + 2;
        int j0 = j - 1 < 0 ? 0 : j - 1, j1 = j + 2 > ncols ? ncols : j + 2;
        for (int ii = i0; ii < i1; ii++)
            for (int jj = j0; jj < j1; jj++) {
                int connected = !four || (ii == i || jj == j);
                uint32_t nb = seg[(size_t)ii * ncols + jj];
                if (connected && nb != seg_id && nb != 0 && seg_size[nb] > seg_size[seg_id]) {
                    float d = 0.0f;
                    for (int b = 0; b < nbands; b++) {
                        nbr[b] = ss[(size_t)nb * nbands + b] / (float)seg_size[nb];
                        float t = spect[b] - nbr[b];
                        float t2 = t * t;
                        d = d + t2;
                    }
                    if (best == 0 || (double)d < best_d) { best_d = (double)d; best = nb; }
                }
            }
    }
    if (best_d > max_spectral_diff * max_spectral_diff) best = 0;
    return best;
}

/* doMerge: shepseg.py:1066-1123 */
static void do_merge(uint32_t s, uint32_t t, uint32_t *seg, uint32_t *seg_size, seg_loc_t *loc,
                     float *ss, int nbands, int ncols)
{
    uint32_t ns = loc[s].n, nt = loc[t].n;
    uint32_t *m = (uint32_t *)malloc(sizeof(uint32_t) * 2 * ((size_t)ns + nt));
    memcpy(m, loc[t].rc, sizeof(uint32_t) * 2 * nt);
    for (uint32_t k = 0; k < ns; k++) {
        uint32_t r = loc[s].rc[2 * k], c = loc[s].rc[2 * k + 1];
        seg[(size_t)r * ncols + c] = t;
        m[2 * (nt + k)] = r; m[2 * (nt + k) + 1] = c;
    }
    free(loc[t].rc); free(loc[s].rc);
    loc[t].rc = m; loc[t].n = ns + nt;
    loc[s].rc = NULL; loc[s].n = 0;
    for (int b = 0; b < nbands; b++) {
        ss[(size_t)t * nbands + b] += ss[(size_t)s * nbands + b];
        ss[(size_t)s * nbands + b] = 0.0f;
    }
    seg_size[t] += seg_size[s];
    seg_size[s] = 0;
}

/* eliminateSmallSegments: shepseg.py:918-1000.  Returns number eliminated;
 * seg relabelled contiguous in place. */
ORC_API int64_t orc_eliminate_small_segments(uint32_t *seg, const void *img, int dtype, int nbands,
                                             int nrows, int ncols, uint32_t max_seg_id,
                                             int min_seg_size, double max_spectral_diff, int four,
                                             uint32_t min_seg_id)
{
    size_t npix = (size_t)nrows * ncols;
    size_t nseg = (size_t)max_seg_id + 1;
    if (nbands > 64) return -1;
    float *ss = build_segment_spectra(seg, img, dtype, nbands, npix, max_seg_id);
    uint32_t *seg_size = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    orc_make_seg_size(seg, npix, max_seg_id, seg_size);
    /* makeSegmentLocations: shepseg.py:880-915 (raster order) */
    seg_loc_t *loc = (seg_loc_t *)calloc(nseg, sizeof(seg_loc_t));
    for (size_t s = 1; s < nseg; s++) {
        loc[s].rc = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (seg_size[s] ? seg_size[s] : 1));
        loc[s].n = 0;
    }
    for (int r = 0; r < nrows; r++)
        for (int c = 0; c < ncols; c++) {
            uint32_t s = seg[(size_t)r * ncols + c];
            if (s != 0) {
                loc[s].rc[2 * loc[s].n] = (uint32_t)r;
                loc[s].rc[2 * loc[s].n + 1] = (uint32_t)c;
                loc[s].n++;
            }
        }
    uint32_t *merge_seg = (uint32_t *)calloc(nseg, sizeof(uint32_t));
    int64_t num_elim = 0;
    for (int target = 1; target < min_seg_size; target++) {
        int64_t count = 0, prev = -1;
        for (size_t s = 0; s < nseg; s++) count += (seg_size[s] == (uint32_t)target);
        int passes = 0;
        while (count != prev && passes < 10) {
            prev = count;
            for (size_t s = min_seg_id; s < nseg; s++)
                if (seg_size[s] == (uint32_t)target)
                    merge_seg[s] = find_merge_segment((uint32_t)s, loc, seg, seg_size, ss, nbands,
                                                      nrows, ncols, max_spectral_diff, four);
            for (size_t s = min_seg_id; s < nseg; s++)
                if (merge_seg[s] != 0) {
                    do_merge((uint32_t)s, merge_seg[s], seg, seg_size, loc, ss, nbands, ncols);
                    merge_seg[s] = 0;
                    num_elim++;
                }
            count = 0;
            for (size_t s = 0; s < nseg; s++) count += (seg_size[s] == (uint32_t)target);
            passes++;
        }
    }
    relabel_segments(seg, npix, seg_size, nseg, min_seg_id);
    for (size_t s = 0; s < nseg; s++) free(loc[s].rc);
    free(loc); free(merge_seg); free(seg_size); free(ss);
    return num_elim;
}

/* doShepherdSegmentation with a supplied k-means model: shepseg.py:130-249
 * (stages :206 predict, :212 clump, :219 segSize, :225 single pixels, :235 small).
 * max_spectral_diff must already be resolved (autoMaxSpectralDiff is host code). */
ORC_API int orc_segment_tile(const void *img, int dtype, int nbands, int nrows, int ncols,
                             const double *centres, int k, int has_null, int64_t null_val,
                             int four, int min_seg_size, double max_spectral_diff,
                             uint32_t *seg_out, uint32_t *max_seg_id_out,
                             int64_t *singles_elim_out, int64_t *small_elim_out,
                             uint32_t *num_clumps_out)
{
    size_t npix = (size_t)nrows * ncols;
    int32_t *clusters = (int32_t *)malloc(sizeof(int32_t) * (npix ? npix : 1));
    int rc = orc_kmeans_assign(img, dtype, nbands, nrows, ncols, centres, k, has_null, null_val,
                               clusters);
    if (rc) { free(clusters); return rc; }
    uint32_t next = orc_clump(clusters, nrows, ncols, 0, four, 1, seg_out);
    free(clusters);
    uint32_t max_seg = next - 1;
    if (num_clumps_out) *num_clumps_out = max_seg;
    uint32_t *seg_size = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)max_seg + 1));
    orc_make_seg_size(seg_out, npix, max_seg, seg_size);
    orc_eliminate_single_pixels(img, dtype, nbands, nrows, ncols, seg_out, seg_size, 1, max_seg,
                                four);
    free(seg_size);
    uint32_t new_max = orc_seg_max(seg_out, npix);
    if (singles_elim_out) *singles_elim_out = (int64_t)max_seg - (int64_t)new_max; /* :226-227 */
    int64_t ne = orc_eliminate_small_segments(seg_out, img, dtype, nbands, nrows, ncols, new_max,
                                              min_seg_size, max_spectral_diff, four, 1);
    if (small_elim_out) *small_elim_out = ne;
    if (max_seg_id_out) *max_seg_id_out = orc_seg_max(seg_out, npix);
    return 0;
}

/* ------------------------------------------------------------------ */
/* k-means fit: sklearn KMeans(init=<array>, n_init=1).fit restated      */
/* (call site shepseg.py:305-312).  float64 Lloyd. */
/* x: nrows*nbands row-major float64 sample.  init: k*nbands.            */
/* Returns 0; centres_out k*nbands; labels_out may be NULL.              */
/* ------------------------------------------------------------------ */
static void lloyd_assign(const double *X, size_t n, int nb, const double *C, int k, int32_t *lab)
{
    /* the E-step of fit is the same chunked dgemm as predict (_k_means_lloyd.pyx), on the centred data */
    double *cn = (double *)malloc(sizeof(double) * k);
    double *m2c = (double *)malloc(sizeof(double) * k * nb);
    orc_kmeans_prepare(C, k, nb, m2c, cn);
    for (size_t i = 0; i < n; i++) {
        int best = 0; double bd = 0.0;
        for (int j = 0; j < k; j++) {
            double d = orc_dist(X + i * nb, m2c + (size_t)j * nb, nb, cn[j]);
            if (j == 0 || d < bd) { bd = d; best = j; }
        }
        lab[i] = best;
    }
    free(cn); free(m2c);
}

/* ---- two pieces of numpy 1.26 that decide which samples an empty cluster is moved to
 * (_k_means_fast.pyx _relocate_empty_clusters_dense: distances = ((X - centers_old[labels])**2).sum(axis=1);
 *  far_from_centers = np.argpartition(distances, -n_empty)[:-n_empty-1:-1]), restated because on
 * lattice-valued imagery many samples are equally far and the choice among them steers the rest of the
 * fit.  Both pinned against numpy itself by oracle/refgen/probe_elkan.py (experiments 3 and 4). ---- */

/* DOUBLE_pairwise_sum (numpy/core/src/umath/loops_utils.h.src) of a contiguous run: what a float64
 * .sum() over the last axis evaluates.  Plain left-to-right below 8 elements. */
static double np_pairwise_block(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_block(a, n2) + np_pairwise_block(a + n2, n - n2);
}
/* ... which a .sum() reaches one ufunc buffer (8192 elements) at a time, the blocks' sums added one after
 * the other (checked against numpy 2.2.6 with runs of up to 2.5 M elements, tests/test_tiling_host.py) */
static double np_pairwise_sum(const double *a, size_t n)
{
    double res = np_pairwise_block(a, n < 8192 ? n : 8192);
    for (size_t o = 8192; o < n; o += 8192) res += np_pairwise_block(a + o, n - o < 8192 ? n - o : 8192);
    return res;
}

/* np.argpartition(v, kth) for float64 without NaNs: numpy/core/src/npysort/selection.cpp
 * introselect_<npy::double_tag, arg=true> (median-of-3 quickselect, median of medians of 5 once the
 * depth limit is spent, selection by repeated minimum when kth is within 3 of the low end), the
 * index array starting as 0..num-1.  No pivot cache (a single kth). */
#define ASWAP(a, b) do { int64_t t_ = (a); (a) = (b); (b) = t_; } while (0)
static void np_aintroselect(const double *v, int64_t *t, int64_t num, int64_t kth);
static int64_t np_amedian5(const double *v, int64_t *t)
{
    if (v[t[1]] < v[t[0]]) ASWAP(t[1], t[0]);
    if (v[t[4]] < v[t[3]]) ASWAP(t[4], t[3]);
    if (v[t[3]] < v[t[0]]) ASWAP(t[3], t[0]);
    if (v[t[4]] < v[t[1]]) ASWAP(t[4], t[1]);
    if (v[t[2]] < v[t[1]]) ASWAP(t[2], t[1]);
    if (v[t[3]] < v[t[2]]) return (v[t[3]] < v[t[1]]) ? 1 : 3;
    return 2;
}
static int64_t np_amedian_of_median5(const double *v, int64_t *t, int64_t num)
{
    const int64_t right = num - 1, nmed = (right + 1) / 5;
    for (int64_t i = 0, subleft = 0; i < nmed; i++, subleft += 5) {
        const int64_t m = np_amedian5(v, t + subleft);
        ASWAP(t[subleft + m], t[i]);
    }
    if (nmed > 2) np_aintroselect(v, t, nmed, nmed / 2);
    return nmed / 2;
}
static void np_aintroselect(const double *v, int64_t *t, int64_t num, int64_t kth)
{
    int64_t low = 0, high = num - 1;
    if (kth - low < 3) {                                   /* dumb_select_ */
        const int64_t n2 = high - low + 1;
        for (int64_t i = 0; i <= kth - low; i++) {
            int64_t minidx = i;
            double minval = v[t[low + i]];
            for (int64_t q = i + 1; q < n2; q++)
                if (v[t[low + q]] < minval) { minidx = q; minval = v[t[low + q]]; }
            ASWAP(t[low + i], t[low + minidx]);
        }
        return;
    }
    if (kth == num - 1) {
        int64_t maxidx = low;
        double maxval = v[t[low]];
        for (int64_t q = low + 1; q < num; q++)
            if (!(v[t[q]] < maxval)) { maxidx = q; maxval = v[t[q]]; }
        ASWAP(t[kth], t[maxidx]);
        return;
    }
    int depth_limit = 0;
    for (uint64_t u = (uint64_t)num >> 1; u; u >>= 1) depth_limit++;      /* npy_get_msb */
    depth_limit *= 2;
    for (; low + 1 < high;) {
        int64_t ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int64_t mid = low + (high - low) / 2;    /* median3_swap_ */
            if (v[t[high]] < v[t[mid]]) ASWAP(t[high], t[mid]);
            if (v[t[high]] < v[t[low]]) ASWAP(t[high], t[low]);
            if (v[t[low]] < v[t[mid]]) ASWAP(t[low], t[mid]);
            ASWAP(t[mid], t[low + 1]);
        } else {
            const int64_t mid = ll + np_amedian_of_median5(v, t + ll, hh - ll);
            ASWAP(t[mid], t[low]);
            ll--; hh++;
        }
        depth_limit--;
        const double pivot = v[t[low]];
        for (;;) {                                         /* unguarded_partition_ */
            do ll++; while (v[t[ll]] < pivot);
            do hh--; while (pivot < v[t[hh]]);
            if (hh < ll) break;
            ASWAP(t[hh], t[ll]);
        }
        ASWAP(t[low], t[hh]);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1 && v[t[high]] < v[t[low]]) ASWAP(t[high], t[low]);
}
ORC_API void orc_np_argpartition(const double *v, int64_t num, int64_t kth, int64_t *out)
{
    for (int64_t i = 0; i < num; i++) out[i] = i;
    if (num > 0) np_aintroselect(v, out, num, kth);
}
ORC_API double orc_np_pairwise_sum(const double *a, int64_t n) { return np_pairwise_sum(a, (size_t)n); }

/* ---- Elkan's variant (what KMeans(algorithm="auto") runs for k > 1 in sklearn 0.24.2:
 * _kmeans.py:824-825, _kmeans_single_elkan :300-428, _k_means_elkan.pyx init_bounds_dense /
 * elkan_iter_chunked_dense / _update_chunk_dense).  In exact arithmetic it visits the partitions of
 * Lloyd's algorithm; in float64 it differs wherever a sample is (nearly) equidistant from two
 * centres: distances are the direct sqrt(sum (x - c)^2) of _euclidean_dense_dense instead of the
 * dgemm expansion, a sample keeps its label unless another centre is STRICTLY closer, and the
 * triangle-inequality bounds decide which distances are looked at at all.  Restated operation by
 * operation, bounds included, so that exact ties fall the way the reference's do. ---- */
static double elk_dist(const double *a, const double *b, int nf)          /* _euclidean_dense_dense, squared=False */
{
    int n4 = nf / 4, rem = nf % 4;
    double result = 0.0;
    for (int i = 0; i < n4; i++) {
        result += ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) +
                   (a[2] - b[2]) * (a[2] - b[2]) + (a[3] - b[3]) * (a[3] - b[3]));
        a += 4; b += 4;
    }
    for (int i = 0; i < rem; i++) result += (a[i] - b[i]) * (a[i] - b[i]);
    return sqrt(result);
}

/* center_half_distances = euclidean_distances(centers) / 2 (sklearn.metrics.pairwise: -2 X.X^T + |x|^2
 * + |y|^2, clipped at 0, zero diagonal, sqrt; the k x k product as this stack's BLAS evaluates it for
 * these shapes: products and sums one after the other in band order, no fma -- pinned by
 * oracle/refgen/probe_elkan.py) and distance_next_center = np.partition(half, 1, axis=0)[1] */
static void elk_half_distances(const double *C, int k, int nb, double *half, double *next)
{
    double *xx = (double *)malloc(sizeof(double) * k);
    for (int a = 0; a < k; a++) xx[a] = orc_sqnorm(C + (size_t)a * nb, nb);
    for (int a = 0; a < k; a++)
        for (int b = 0; b < k; b++) {
            double d = 0.0;
            for (int t = 0; t < nb; t++) d = d + C[(size_t)a * nb + t] * C[(size_t)b * nb + t];
            double v = -2.0 * d;
            v = v + xx[a];
            v = v + xx[b];
            if (!(v > 0.0)) v = 0.0;
            if (a == b) v = 0.0;
            half[(size_t)a * k + b] = sqrt(v) / 2.0;
        }
    for (int l = 0; l < k; l++) {          /* second smallest of column l (the diagonal zero is the smallest) */
        double m0 = half[l], m1 = -1.0;
        for (int a = 1; a < k; a++) {
            const double v = half[(size_t)a * k + l];
            if (v < m0) { m1 = m0; m0 = v; }
            else if (m1 < 0.0 || v < m1) m1 = v;
        }
        next[l] = k > 1 ? m1 : m0;
    }
    free(xx);
}

static void elk_init_bounds(const double *X, size_t n, int nb, const double *C, int k, const double *half,
                            int32_t *lab, double *ub, double *lb)
{
    for (size_t i = 0; i < n; i++) {
        int best = 0;
        double min_dist = elk_dist(X + i * nb, C, nb);
        lb[i * k] = min_dist;
        for (int j = 1; j < k; j++)
            if (min_dist > half[(size_t)best * k + j]) {
                const double dist = elk_dist(X + i * nb, C + (size_t)j * nb, nb);
                lb[i * k + j] = dist;
                if (dist < min_dist) { min_dist = dist; best = j; }
            }
        lab[i] = best;
        ub[i] = min_dist;
    }
}

static void elk_estep(const double *X, size_t n, int nb, const double *C, int k, const double *half,
                      const double *next, int32_t *lab, double *ub, double *lb)
{
    for (size_t i = 0; i < n; i++) {
        double upper = ub[i];
        int tight = 0, label = lab[i];
        if (!(next[label] >= upper)) {
            for (int j = 0; j < k; j++)
                if (j != label && upper > lb[i * k + j] && upper > half[(size_t)label * k + j]) {
                    if (!tight) {
                        upper = elk_dist(X + i * nb, C + (size_t)label * nb, nb);
                        lb[i * k + label] = upper;
                        tight = 1;
                    }
                    if (upper > lb[i * k + j] || upper > half[(size_t)label * k + j]) {
                        const double dist = elk_dist(X + i * nb, C + (size_t)j * nb, nb);
                        lb[i * k + j] = dist;
                        if (dist < upper) { label = j; upper = dist; }
                    }
                }
            lab[i] = label;
            ub[i] = upper;
        }
    }
}

/* chunk_rows == 0: the M-step sums rows one after the other, as sklearn does with one OpenMP
 * thread (_kmeans_lloyd.pyx lloyd_iter_chunked_dense / _update_chunk_dense; with more threads
 * sklearn adds per-thread partial sums in whatever order the threads finish, so the association
 * of these sums is not fixed by the reference).  chunk_rows > 0: the sums are associated as
 * the HIP fit does (kmeans.h k_fit_partial / k_fit_reduce1 / k_fit_update): per chunk of
 * chunk_rows rows in row order, the chunks of a group of group_chunks in chunk order, the groups
 * in order.  Everything else is identical; tests use the second form to show that where the
 * device and the row-order oracle part ways the association of these sums is the only cause. */
static int kmeans_fit_impl(const double *xin, int64_t nrows, int nbands, int k,
                           const double *init, int max_iter, double tol_rel,
                           int chunk_rows, int group_chunks, int elkan,
                           double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    size_t n = (size_t)nrows;
    int nb = nbands;
    double *X = (double *)malloc(sizeof(double) * n * nb);
    double *mu = (double *)calloc(nb, sizeof(double));
    double *C = (double *)malloc(sizeof(double) * k * nb);
    double *Cn = (double *)malloc(sizeof(double) * k * nb);
    double *w = (double *)malloc(sizeof(double) * k);
    int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *lab_old = (int32_t *)malloc(sizeof(int32_t) * n);
    /* centre the data (sklearn _kmeans.py fit: X -= X.mean(axis=0)) */
    for (int b = 0; b < nb; b++) {
        double s = 0.0;
        for (size_t i = 0; i < n; i++) s += xin[i * nb + b];
        mu[b] = s / (double)n;
    }
    for (size_t i = 0; i < n; i++)
        for (int b = 0; b < nb; b++) X[i * nb + b] = xin[i * nb + b] - mu[b];
    for (int j = 0; j < k; j++)
        for (int b = 0; b < nb; b++) C[j * nb + b] = init[j * nb + b] - mu[b];
    /* tol = mean(var(X, axis=0)) * tol_rel  (_tolerance) */
    double tol = 0.0;
    for (int b = 0; b < nb; b++) {
        double m = 0.0, v = 0.0;
        for (size_t i = 0; i < n; i++) m += X[i * nb + b];
        m /= (double)n;
        for (size_t i = 0; i < n; i++) { double d = X[i * nb + b] - m; v += d * d; }
        tol += v / (double)n;
    }
    tol = tol / nb * tol_rel;
    int strict = 0, it = 0, have_old = 0;
    double *half = NULL, *next = NULL, *ub = NULL, *lb = NULL, *cshift = NULL;
    if (elkan) {
        half = (double *)malloc(sizeof(double) * k * k);
        next = (double *)malloc(sizeof(double) * k);
        cshift = (double *)malloc(sizeof(double) * k);
        ub = (double *)calloc(n ? n : 1, sizeof(double));
        lb = (double *)calloc((n ? n : 1) * (size_t)k, sizeof(double));
        elk_half_distances(C, k, nb, half, next);
        elk_init_bounds(X, n, nb, C, k, half, lab, ub, lb);
    }
    for (it = 1; it <= max_iter; it++) {
        if (elkan) elk_estep(X, n, nb, C, k, half, next, lab, ub, lb);
        else lloyd_assign(X, n, nb, C, k, lab);
        memset(Cn, 0, sizeof(double) * k * nb);
        memset(w, 0, sizeof(double) * k);
        if (chunk_rows <= 0) {
            for (size_t i = 0; i < n; i++) {
                w[lab[i]] += 1.0;
                for (int b = 0; b < nb; b++) Cn[lab[i] * nb + b] += X[i * nb + b];
            }
        } else {
            const int kn = k * nb;
            double *P = (double *)malloc(sizeof(double) * kn), *P2 = (double *)malloc(sizeof(double) * kn);
            const size_t gl = (size_t)chunk_rows * (size_t)(group_chunks > 0 ? group_chunks : 1);
            for (size_t g0 = 0; g0 < n; g0 += gl) {
                const size_t g1 = g0 + gl < n ? g0 + gl : n;
                for (int t = 0; t < kn; t++) P2[t] = 0.0;
                for (size_t c0 = g0; c0 < g1; c0 += (size_t)chunk_rows) {
                    const size_t c1 = c0 + (size_t)chunk_rows < g1 ? c0 + (size_t)chunk_rows : g1;
                    for (int t = 0; t < kn; t++) P[t] = 0.0;
                    for (size_t i = c0; i < c1; i++) {
                        w[lab[i]] += 1.0;
                        for (int b = 0; b < nb; b++) P[lab[i] * nb + b] += X[i * nb + b];
                    }
                    for (int t = 0; t < kn; t++) P2[t] += P[t];
                }
                for (int t = 0; t < kn; t++) Cn[t] += P2[t];
            }
            free(P); free(P2);
        }
        int n_empty = 0;
        for (int j = 0; j < k; j++) n_empty += (w[j] == 0.0);
        if (n_empty > 0) {
            /* _relocate_empty_clusters_dense: farthest samples from their OLD centres */
            double *dist = (double *)malloc(sizeof(double) * n);
            double *sq = (double *)malloc(sizeof(double) * nb);
            double dmax = 0.0;
            for (size_t i = 0; i < n; i++) {
                for (int b = 0; b < nb; b++) {
                    double t = X[i * nb + b] - C[lab[i] * nb + b];
                    sq[b] = t * t;
                }
                const double d = np_pairwise_sum(sq, (size_t)nb);
                dist[i] = d; if (d > dmax) dmax = d;
            }
            free(sq);
            if (ORC_RELOCATE_GUARD == 0 || dmax > 0.0) {
                /* empty_clusters is taken once, ascending (np.where), before any move; the r-th of
                 * them gets far_from_centers[r] = np.argpartition(distances, -n_empty)[n - 1 - r] */
                int *empties = (int *)malloc(sizeof(int) * n_empty);
                int64_t *part = (int64_t *)malloc(sizeof(int64_t) * n);
                int ne = 0;
                for (int j = 0; j < k; j++) if (w[j] == 0.0) empties[ne++] = j;
                orc_np_argpartition(dist, (int64_t)n, (int64_t)n - n_empty, part);
                for (int r = 0; r < n_empty; r++) {
                    const size_t f = (size_t)part[n - 1 - (size_t)r];
                    int e = empties[r];
                    int old = lab[f];
                    for (int b = 0; b < nb; b++) {
                        Cn[old * nb + b] -= X[f * nb + b];
                        Cn[e * nb + b] = X[f * nb + b];
                    }
                    w[e] = 1.0; w[old] -= 1.0;
                }
                free(empties); free(part);
            }
            free(dist);
        }
        /* _average_centers (sklearn 0.24.2, the pinned oracle stack): centres *= 1/w for
         * w > 0; a cluster left with w == 0 keeps its residual sum (sklearn >= 1.0
         * instead copies the biggest cluster's centre: ORC_SKLEARN_GE_1). */
#if ORC_SKLEARN_GE_1
        int jmax = 0;
        for (int j = 1; j < k; j++) if (w[j] > w[jmax]) jmax = j;
#endif
        for (int j = 0; j < k; j++) {
            if (w[j] > 0.0) {
                double alpha = 1.0 / w[j];
                for (int b = 0; b < nb; b++) Cn[j * nb + b] *= alpha;
            }
#if ORC_SKLEARN_GE_1
            else for (int b = 0; b < nb; b++) Cn[j * nb + b] = Cn[jmax * nb + b];
#endif
        }
        /* center_shift[j] = euclidean norm (4-way unrolled in sklearn's
         * _euclidean_dense_dense), center_shift_tot = sum(center_shift**2) */
        /* ... center_shift_tot = (center_shift**2).sum(): numpy's pairwise sum */
        double shift = 0.0;
        double *shsq = (double *)malloc(sizeof(double) * k);
        for (int j = 0; j < k; j++) {
            const double *a = &Cn[j * nb], *c = &C[j * nb];
            double r = 0.0;
            int b = 0;
            for (; b + 4 <= nb; b += 4)
                r += ((a[b] - c[b]) * (a[b] - c[b]) + (a[b + 1] - c[b + 1]) * (a[b + 1] - c[b + 1]) +
                      (a[b + 2] - c[b + 2]) * (a[b + 2] - c[b + 2]) +
                      (a[b + 3] - c[b + 3]) * (a[b + 3] - c[b + 3]));
            for (; b < nb; b++) r += (a[b] - c[b]) * (a[b] - c[b]);
            double s = sqrt(r);
            shsq[j] = s * s;
            if (elkan) cshift[j] = s;
        }
        shift = np_pairwise_sum(shsq, (size_t)k);
        free(shsq);
        if (elkan) {
            /* end of elkan_iter: the bounds follow the centres; then the new centres' half distances */
            for (size_t i = 0; i < n; i++) {
                ub[i] += cshift[lab[i]];
                for (int j = 0; j < k; j++) {
                    lb[i * k + j] -= cshift[j];
                    if (lb[i * k + j] < 0) lb[i * k + j] = 0;
                }
            }
            elk_half_distances(Cn, k, nb, half, next);
        }
        memcpy(C, Cn, sizeof(double) * k * nb);
        if (have_old && memcmp(lab, lab_old, sizeof(int32_t) * n) == 0) { strict = 1; break; }
        if (shift <= tol) break;
        memcpy(lab_old, lab, sizeof(int32_t) * n); have_old = 1;
    }
    if (it > max_iter) it = max_iter;
    if (!strict) {
        if (elkan) elk_estep(X, n, nb, C, k, half, next, lab, ub, lb);
        else lloyd_assign(X, n, nb, C, k, lab);
    }
    free(half); free(next); free(ub); free(lb); free(cshift);
    for (int j = 0; j < k; j++)
        for (int b = 0; b < nb; b++) centres_out[j * nb + b] = C[j * nb + b] + mu[b];
    if (labels_out) memcpy(labels_out, lab, sizeof(int32_t) * n);
    if (n_iter_out) *n_iter_out = it;
    free(X); free(mu); free(C); free(Cn); free(w); free(lab); free(lab_old);
    return 0;
}

ORC_API int orc_kmeans_fit(const double *xin, int64_t nrows, int nbands, int k,
                           const double *init, int max_iter, double tol_rel,
                           double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    return kmeans_fit_impl(xin, nrows, nbands, k, init, max_iter, tol_rel, 0, 0, 0, centres_out, labels_out,
                           n_iter_out);
}

ORC_API int orc_kmeans_fit_assoc(const double *xin, int64_t nrows, int nbands, int k,
                                 const double *init, int max_iter, double tol_rel,
                                 int chunk_rows, int group_chunks,
                                 double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    return kmeans_fit_impl(xin, nrows, nbands, k, init, max_iter, tol_rel, chunk_rows, group_chunks, 0,
                           centres_out, labels_out, n_iter_out);
}

ORC_API int orc_kmeans_fit_elkan(const double *xin, int64_t nrows, int nbands, int k,
                                 const double *init, int max_iter, double tol_rel,
                                 int chunk_rows, int group_chunks,
                                 double *centres_out, int32_t *labels_out, int *n_iter_out)
{
    return kmeans_fit_impl(xin, nrows, nbands, k, init, max_iter, tol_rel, chunk_rows, group_chunks, 1,
                           centres_out, labels_out, n_iter_out);
}

/* ------------------------------------------------------------------ */
/* Stitch: tiling.py:1066-1306 recodeTile / recodeSharedSegments /       */
/* crossesMidline / relabelSegments, for ONE tile.                       */
/*  tile      ys*xs local segment ids (0 = null), not modified           */
/*  top_b     the saved (already recoded) bottom strip of the tile above */
/*            (ovr rows x xs, row pitch top_pitch) or NULL (tile row 0)  */
/*  left_b    saved right strip of the tile to the left (ys x ovc cols,  */
/*            row pitch left_pitch) or NULL (tile col 0)                 */
/*  overlap   overlapSize; the A strips are tile[:overlap,:] and         */
/*            tile[:, :overlap] (clipped to the tile)                    */
/*  out       ys*xs recoded copy (unlisted ids -> 0).  Returns newSegId. */
/* ------------------------------------------------------------------ */
static void recode_shared(const uint32_t *tile, int ys, int xs, int an_rows, int an_cols,
                          const uint32_t *b, size_t b_pitch, int horizontal, uint32_t max_local,
                          uint8_t *in_dict, uint32_t *recode)
{
    /* crossesMidline tiling.py:1271-1306: minN < mid && maxN >= mid along the short axis */
    (void)ys;
    int mid = horizontal ? an_rows / 2 : an_cols / 2;
    int *mn = (int *)malloc(sizeof(int) * ((size_t)max_local + 1));
    int *mx = (int *)malloc(sizeof(int) * ((size_t)max_local + 1));
    for (size_t s = 0; s <= max_local; s++) { mn[s] = 0x7fffffff; mx[s] = -1; }
    for (int r = 0; r < an_rows; r++)
        for (int c = 0; c < an_cols; c++) {
            uint32_t s = tile[(size_t)r * xs + c];
            int v = horizontal ? r : c;
            if (v < mn[s]) mn[s] = v;
            if (v > mx[s]) mx[s] = v;
        }
    /* per crossing segment: scipy.stats.mode of B under the segment's A pixels
     * (most frequent value, smallest on ties; may be 0)  tiling.py:1194-1203 */
    size_t npx = (size_t)an_rows * an_cols;
    uint32_t *cnt_seg = (uint32_t *)calloc((size_t)max_local + 2, sizeof(uint32_t));
    for (int r = 0; r < an_rows; r++)
        for (int c = 0; c < an_cols; c++) cnt_seg[tile[(size_t)r * xs + c] + 1]++;
    for (size_t s = 1; s <= (size_t)max_local + 1; s++) cnt_seg[s] += cnt_seg[s - 1];
    uint32_t *vals = (uint32_t *)malloc(sizeof(uint32_t) * (npx ? npx : 1));
    uint32_t *fill = (uint32_t *)calloc((size_t)max_local + 1, sizeof(uint32_t));
    for (int r = 0; r < an_rows; r++)
        for (int c = 0; c < an_cols; c++) {
            uint32_t s = tile[(size_t)r * xs + c];
            vals[cnt_seg[s] + fill[s]++] = b[(size_t)r * b_pitch + c];
        }
    for (size_t s = 1; s <= max_local; s++) {
        if (mx[s] < 0) continue;                       /* not in the strip */
        if (!(mn[s] < mid && mx[s] >= mid)) continue;
        uint32_t *v = vals + cnt_seg[s];
        uint32_t n = cnt_seg[s + 1] - cnt_seg[s];
        /* insertion-free mode: sort the small list */
        for (uint32_t i = 1; i < n; i++) {             /* simple insertion sort (test sizes) */
            uint32_t x = v[i]; uint32_t j = i;
            while (j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; j--; }
            v[j] = x;
        }
        uint32_t best = 0, bestc = 0;
        for (uint32_t i = 0; i < n;) {
            uint32_t j = i;
            while (j < n && v[j] == v[i]) j++;
            if (j - i > bestc) { bestc = j - i; best = v[i]; }
            i = j;
        }
        in_dict[s] = 1; recode[s] = best;
    }
    free(mn); free(mx); free(cnt_seg); free(vals); free(fill);
}

ORC_API uint32_t orc_recode_tile(const uint32_t *tile, int ys, int xs, int overlap,
                                 const uint32_t *top_b, size_t top_pitch,
                                 const uint32_t *left_b, size_t left_pitch,
                                 uint32_t max_seg_id, int top, int bottom, int left, int right,
                                 uint32_t *out)
{
    size_t npix = (size_t)ys * xs;
    uint32_t max_local = 0;
    for (size_t p = 0; p < npix; p++) if (tile[p] > max_local) max_local = tile[p];
    uint8_t *in_dict = (uint8_t *)calloc((size_t)max_local + 1, 1);
    uint32_t *recode = (uint32_t *)calloc((size_t)max_local + 1, sizeof(uint32_t));
    int an_rows = overlap < ys ? overlap : ys, an_cols = overlap < xs ? overlap : xs;
    if (top_b) recode_shared(tile, ys, xs, an_rows, xs, top_b, top_pitch, 1, max_local, in_dict, recode);
    if (left_b) recode_shared(tile, ys, xs, ys, an_cols, left_b, left_pitch, 0, max_local, in_dict, recode);
    /* relabelSegments tiling.py:1205-1269 */
    int *seg_top = (int *)malloc(sizeof(int) * ((size_t)max_local + 1));
    int *seg_left = (int *)malloc(sizeof(int) * ((size_t)max_local + 1));
    for (size_t s = 0; s <= max_local; s++) { seg_top[s] = 0x7fffffff; seg_left[s] = 0x7fffffff; }
    for (int r = 0; r < ys; r++)
        for (int c = 0; c < xs; c++) {
            uint32_t s = tile[(size_t)r * xs + c];
            if (r < seg_top[s]) seg_top[s] = r;
            if (c < seg_left[s]) seg_left[s] = c;
        }
    uint32_t *lut = (uint32_t *)calloc((size_t)max_local + 1, sizeof(uint32_t));
    uint32_t new_id = max_seg_id;
    for (size_t s = 1; s <= max_local; s++) {
        if (in_dict[s]) lut[s] = recode[s];
        else if (seg_left[s] >= left && seg_top[s] >= top && seg_left[s] < right && seg_top[s] < bottom)
            lut[s] = ++new_id;
    }
    for (size_t p = 0; p < npix; p++) out[p] = lut[tile[p]];
    free(in_dict); free(recode); free(seg_top); free(seg_left); free(lut);
    return new_id;
}

/* ------------------------------------------------------------------ */
/* Per-segment statistics: tilingstats.py accumulateSegDict :466-515,    */
/* SegmentStats :922-1008 (N11), RatPage :1972-2045.                      */
/*  seg[npix] uint32 labels (0 = null), band[npix] image values,          */
/*  stats_sel[nstats][5] = {globalCol, statId, colType, colArrayIdx,      */
/*  param} (makeFastStatsSelection :798-863); statId 0..7 = min, max,     */
/*  mean, stddev, median, mode, percentile, pixcount.                     */
/*  intcols_out[nint][max_seg_id+1] int64, floatcols_out[nflt][..] f32;   */
/*  row 0 is zero (the null segment, RatPage :1992-1996).                 */
/* ------------------------------------------------------------------ */
static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

ORC_API int orc_segstats(const uint32_t *seg, const void *band, int dtype, int64_t npix,
                         uint32_t max_seg_id, int has_null, int64_t null_val,
                         const uint32_t *stats_sel, int nstats, int64_t missing,
                         int64_t *intcols_out, float *floatcols_out)
{
    size_t ns = (size_t)max_seg_id + 1;
    int nint = 0, nflt = 0;
    for (int i = 0; i < nstats; i++) {
        if (stats_sel[i * 5 + 2] == 0) nint++; else nflt++;
    }
    /* group valid pixel values by segment (counting sort), then sort each group by value */
    size_t *off = (size_t *)calloc(ns + 1, sizeof(size_t));
    for (int64_t p = 0; p < npix; p++) {
        uint32_t s = seg[p];
        if (s == 0 || s > max_seg_id) continue;
        int64_t v = px_get(band, dtype, (size_t)p);
        if (has_null && v == null_val) continue;
        off[s + 1]++;
    }
    for (size_t s = 0; s < ns; s++) off[s + 1] += off[s];
    int64_t *vals = (int64_t *)malloc(sizeof(int64_t) * (off[ns] ? off[ns] : 1));
    size_t *fill = (size_t *)calloc(ns, sizeof(size_t));
    for (int64_t p = 0; p < npix; p++) {
        uint32_t s = seg[p];
        if (s == 0 || s > max_seg_id) continue;
        int64_t v = px_get(band, dtype, (size_t)p);
        if (has_null && v == null_val) continue;
        vals[off[s] + fill[s]++] = v;
    }
    for (int c = 0; c < nint; c++) intcols_out[(size_t)c * ns] = 0;
    for (int c = 0; c < nflt; c++) floatcols_out[(size_t)c * ns] = 0.0f;
    for (size_t s = 1; s < ns; s++) {
        int64_t *a = vals + off[s];
        size_t n = off[s + 1] - off[s];
        qsort(a, n, sizeof(int64_t), cmp_i64);
        /* SegmentStats.__init__ */
        int64_t vmin = missing, vmax = missing, vmode = missing, vmed = missing;
        float mean = (float)missing, stddev = (float)missing;
        if (n > 0) {
            vmin = a[0]; vmax = a[n - 1];
            int64_t sum = 0;
            for (size_t i = 0; i < n; i++) sum += a[i];
            mean = (float)((double)sum / (double)(uint32_t)n);
            /* variance (tilingstats.py:951): established by running the reference (golden
             * stats_*.npz, 403 segments): each bin's term counts*(pixVals-mean)**2 is
             * evaluated in float64 with the float32-rounded mean and stored as float32,
             * .sum() accumulates those in float32 in value order, the division by pixCount
             * and the sqrt are float64, the result is stored as float32. */
            float var = 0.0f;
            size_t bestc = 0;
            for (size_t i = 0; i < n;) {
                size_t j = i;
                while (j < n && a[j] == a[i]) j++;
                double d = (double)a[i] - (double)mean;
                float term = (float)((double)(uint32_t)(j - i) * (d * d));
                var = var + term;
                if (j - i > bestc) { bestc = j - i; vmode = a[i]; }
                i = j;
            }
            stddev = (float)sqrt((double)var / (double)(uint32_t)n);
        }
        for (int i = 0; i < nstats; i++) {
            uint32_t stat = stats_sel[i * 5 + 1], ctype = stats_sel[i * 5 + 2];
            uint32_t cidx = stats_sel[i * 5 + 3], param = stats_sel[i * 5 + 4];
            double val = 0.0;
            int is_pct = (stat == 4 || stat == 6);
            if (is_pct) {
                /* getPercentile :969-986 (median = getPercentile(50)); percentile 0 returns
                 * pixVals[-1] because the while loop never runs */
                if (n == 0) val = (double)missing;
                else {
                    double pc = (stat == 4) ? 50.0 : (double)param;
                    double t = (double)(uint32_t)n * (pc / 100.0);
                    size_t cum = 0, k = 0, i2 = 0;
                    int64_t pv = a[n - 1];
                    while ((double)cum < t) {
                        size_t j = i2;
                        while (j < n && a[j] == a[i2]) j++;
                        cum += j - i2; pv = a[i2]; i2 = j; k++;
                    }
                    (void)k;
                    val = (double)pv;
                    if (stat == 4) vmed = pv;
                }
            } else if (stat == 0) val = (double)vmin;
            else if (stat == 1) val = (double)vmax;
            else if (stat == 2) val = (double)mean;
            else if (stat == 3) val = (double)stddev;
            else if (stat == 5) val = (double)vmode;
            else if (stat == 7) val = (double)(uint32_t)n;
            if (ctype == 0) intcols_out[(size_t)cidx * ns + s] = (int64_t)val;
            else floatcols_out[(size_t)cidx * ns + s] = (float)val;
        }
        (void)vmed;
    }
    free(off); free(vals); free(fill);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * subset.subsetImage's recode (reference subset.py:124-166 tile loop + processSubsetTile
 * :366-425): the window (tlx, tly, xs, ys) of a label raster is visited tile by tile
 * (tile_size x tile_size, row-major over tiles, raster order inside a tile); masked-out
 * (mask == 0) and null pixels give 0, every other id gets the next new id the first time it is
 * seen.  orig_out[new id] = old id (orig_out[0] = 0), hist_out[new id] = pixel count; both must
 * hold xs*ys + 1 entries.  Returns the number of new ids.
 * ------------------------------------------------------------------------------------------- */
ORC_API uint32_t orc_subset_recode(const uint32_t *seg, int64_t img_cols, int64_t tlx, int64_t tly, int64_t xs,
                           int64_t ys, const uint8_t *mask, int64_t tile_size, uint32_t max_seg_id,
                           uint32_t *out, uint32_t *orig_out, uint32_t *hist_out)
{
    uint32_t *lut = (uint32_t *)calloc((size_t)max_seg_id + 1, sizeof(uint32_t));
    uint32_t nnew = 0;
    orig_out[0] = 0; hist_out[0] = 0;
    for (int64_t ty = 0; ty < ys; ty += tile_size)
        for (int64_t tx = 0; tx < xs; tx += tile_size) {
            const int64_t th = ys - ty < tile_size ? ys - ty : tile_size;
            const int64_t tw = xs - tx < tile_size ? xs - tx : tile_size;
            for (int64_t y = ty; y < ty + th; y++)
                for (int64_t x = tx; x < tx + tw; x++) {
                    const uint32_t s = seg[(size_t)(tly + y) * img_cols + (tlx + x)];
                    uint32_t v = 0;
                    if (!(mask && mask[(size_t)y * xs + x] == 0) && s != 0) {
                        if (lut[s] == 0) {
                            lut[s] = ++nnew;
                            orig_out[nnew] = s;
                            hist_out[nnew] = 0;
                        }
                        v = lut[s];
                        hist_out[v]++;
                    }
                    out[(size_t)y * xs + x] = v;
                }
        }
    free(lut);
    return nnew;
}

/* ---------------------------------------------------------------------------------------------
 * calcPerSegmentSpatialStatsTiled with the reference's three built-in user functions
 * (tilingstats.py:1262-1390 driver, accumulateSegSpatial :1652-1699, userFuncVariogram :1037-1094,
 * userFuncMeanCoord :1098-1142, userFuncNumEdgePixels :1146-1216).  A segment's point list holds
 * its non-nodata pixels in the order the tiles (tile_size x tile_size, row-major) deliver them,
 * raster order inside a tile.  func 0 = mean coordinates (params = GDAL transform[6]; float
 * columns 0, 1), 1 = number of edge pixels (params[0] = fourConnected; int column 0; the
 * 8-connected test omits (y-1, x+1) exactly as the reference does), 2 = variogram (params[0] =
 * maxDist; float columns 0..maxDist-1).  userFunc outputs: intArr int32, floatArr float64, stored
 * into int64 / float32 page columns; unset entries and segments without valid pixels hold
 * `missing`; row 0 is zero.
 * ------------------------------------------------------------------------------------------- */
ORC_API int orc_spatialstats(const uint32_t *seg, const void *band, int dtype, int64_t nrows,
                             int64_t ncols, uint32_t max_seg_id, int64_t null_val, int func,
                             const double *params, int64_t tile_size, int64_t missing, int nint,
                             int nflt, int64_t *intcols_out, float *floatcols_out)
{
    const size_t ns = (size_t)max_seg_id + 1;
    const size_t n = (size_t)nrows * (size_t)ncols;
    size_t *off = (size_t *)calloc(ns + 1, sizeof(size_t));
    size_t *fill = (size_t *)calloc(ns, sizeof(size_t));
    uint8_t *present = (uint8_t *)calloc(ns, 1);
    for (size_t p = 0; p < n; p++) {
        const uint32_t s = seg[p];
        if (s == 0 || s > max_seg_id) continue;
        present[s] = 1;
        if (px_get(band, dtype, p) != null_val) off[s + 1]++;
    }
    for (size_t s = 0; s < ns; s++) off[s + 1] += off[s];
    uint32_t *px = (uint32_t *)malloc((off[ns] + 1) * sizeof(uint32_t));
    uint32_t *py = (uint32_t *)malloc((off[ns] + 1) * sizeof(uint32_t));
    int64_t *pv = (int64_t *)malloc((off[ns] + 1) * sizeof(int64_t));
    for (int64_t ty = 0; ty < nrows; ty += tile_size)
        for (int64_t tx = 0; tx < ncols; tx += tile_size) {
            const int64_t th = nrows - ty < tile_size ? nrows - ty : tile_size;
            const int64_t tw = ncols - tx < tile_size ? ncols - tx : tile_size;
            for (int64_t y = ty; y < ty + th; y++)
                for (int64_t x = tx; x < tx + tw; x++) {
                    const size_t p = (size_t)y * ncols + x;
                    const uint32_t s = seg[p];
                    if (s == 0 || s > max_seg_id) continue;
                    const int64_t v = px_get(band, dtype, p);
                    if (v == null_val) continue;
                    const size_t k = off[s] + fill[s]++;
                    px[k] = (uint32_t)x; py[k] = (uint32_t)y; pv[k] = v;
                }
        }
    for (int c = 0; c < nint; c++) for (size_t s = 0; s < ns; s++) intcols_out[(size_t)c * ns + s] = s ? missing : 0;
    for (int c = 0; c < nflt; c++) for (size_t s = 0; s < ns; s++) floatcols_out[(size_t)c * ns + s] = s ? (float)missing : 0.0f;
    for (size_t s = 1; s < ns; s++) {
        const size_t m = off[s + 1] - off[s];
        if (m == 0) continue;
        const uint32_t *X = px + off[s], *Y = py + off[s];
        const int64_t *V = pv + off[s];
        if (func == 0) {
            double sumx = 0.0, sumy = 0.0;
            for (size_t i = 0; i < m; i++) {
                const double geox = params[0] + params[1] * (double)X[i] + params[2] * (double)Y[i];
                const double geoy = params[3] + params[4] * (double)X[i] + params[5] * (double)Y[i];
                sumx += geox; sumy += geoy;
            }
            if (nflt > 0) floatcols_out[s] = (float)(sumx / (double)m);
            if (nflt > 1) floatcols_out[ns + s] = (float)(sumy / (double)m);
            continue;
        }
        uint32_t xmin = X[0], xmax = X[0], ymin = Y[0], ymax = Y[0];
        for (size_t i = 1; i < m; i++) {
            if (X[i] < xmin) xmin = X[i]; else if (X[i] > xmax) xmax = X[i];
            if (Y[i] < ymin) ymin = Y[i]; else if (Y[i] > ymax) ymax = Y[i];
        }
        const int64_t W = (int64_t)xmax - xmin + 1, H = (int64_t)ymax - ymin + 1;
        if (func == 1) {
            uint8_t *mask = (uint8_t *)calloc((size_t)W * H, 1);
            for (size_t i = 0; i < m; i++) mask[(size_t)(Y[i] - ymin) * W + (X[i] - xmin)] = 1;
            const int four = params[0] != 0.0;
            int32_t total_edge = 0;
#define MK(yy, xx) ((int)mask[(size_t)(yy) * W + (xx)])
            for (int64_t y = 0; y < H; y++)
                for (int64_t x = 0; x < W; x++) {
                    if (!MK(y, x)) continue;
                    if (y == 0 || x == 0 || y == H - 1 || x == W - 1) { total_edge++; continue; }
                    int tot;
                    if (four) {
                        tot = MK(y - 1, x) + MK(y + 1, x) + MK(y, x - 1) + MK(y, x + 1);
                        if (tot != 4) total_edge++;
                    } else {                 /* sic: (y+1, x+1) twice, (y-1, x+1) never */
                        tot = MK(y - 1, x - 1) + MK(y - 1, x) + MK(y + 1, x + 1) + MK(y, x - 1) + MK(y, x + 1) +
                              MK(y + 1, x - 1) + MK(y + 1, x) + MK(y + 1, x + 1);
                        if (tot != 8) total_edge++;
                    }
                }
#undef MK
            free(mask);
            if (nint > 0) intcols_out[s] = (int64_t)total_edge;
        } else {
            const int64_t maxd = (int64_t)params[0];
            int64_t *tile = (int64_t *)malloc((size_t)W * H * sizeof(int64_t));
            for (size_t i = 0; i < (size_t)W * H; i++) tile[i] = null_val;
            for (size_t i = 0; i < m; i++) tile[(size_t)(Y[i] - ymin) * W + (X[i] - xmin)] = V[i];
            uint32_t *counts = (uint32_t *)calloc((size_t)maxd + 1, sizeof(uint32_t));
            double *sums = (double *)calloc((size_t)maxd + 1, sizeof(double));
            for (int64_t y = 0; y < H; y++)
                for (int64_t x = 0; x < W; x++) {
                    const int64_t val = tile[(size_t)y * W + x];
                    if (val == null_val) continue;
                    for (int64_t yo = 1; yo <= maxd; yo++)
                        for (int64_t xo = 1; xo <= maxd; xo++) {
                            if (y + yo >= H || x + xo >= W) continue;
                            const int64_t val2 = tile[(size_t)(y + yo) * W + (x + xo)];
                            if (val2 == null_val) continue;
                            const int64_t dist = (int64_t)sqrt((double)(yo * yo + xo * xo));
                            if (dist <= maxd && dist > 0) {
                                counts[dist - 1]++;
                                /* numba: an int64 product that wraps (32-bit imagery only), then float64 */
                                const uint64_t du = (uint64_t)(val - val2);
                                sums[dist - 1] += (double)(int64_t)(du * du);
                            }
                        }
                }
            for (int64_t d = 0; d < maxd && d < nflt; d++)
                if (counts[d] > 0) floatcols_out[(size_t)d * ns + s] = (float)sqrt(sums[d] / (double)counts[d]);
            free(tile); free(counts); free(sums);
        }
    }
    (void)present;
    free(off); free(fill); free(present); free(px); free(py); free(pv);
    return 0;
}
