/* How many bounds rows would a lazy Elkan E-step touch?  Design study for fit_elkan.h (round 4).
 *
 *   gcc -O2 -mfma -ffp-contract=off -o /tmp/elkan_filter_sim tools/elkan_filter_sim.c -lm
 *   /tmp/elkan_filter_sim sample.npy-raw nrows nbands k iters
 *
 * Runs the oracle's Elkan fit (oracle/shepseg_oracle.c, included for its static functions) on a sample
 * (raw uint16 rows of nbands) and, beside it, simulates filters that decide WITHOUT the sample's row of
 * lower bounds whether the reference can touch the sample in an iteration:
 *   truth : the sample has a candidate centre (gate open and some j with upper > lb_j and upper > half)
 *   A     : one scalar per sample -- min over j of max(lb_j, half[label][j]) at the last look, decayed
 *           by the largest centre shift per iteration
 *   B / C : the 1 / 2 most dangerous centres tracked exactly (their bound + the cumulated shift of that
 *           centre), the rest as in A
 * Per iteration: fraction with a true candidate, fraction each filter lets through (rows read).
 * Test infrastructure; nothing here is shipped. */
#include "../oracle/shepseg_oracle.c"
#include <stdio.h>

#define NT 4                /* designs: 0 = A (one scalar over max(lb, half)), 1..3 = half-protected centres as a bit mask tested
                               exactly + 0 / 1 / 2 lb-protected centres tracked exactly + one scalar for the rest */
typedef struct { double m_rest; unsigned long long H; int j[2]; int label; } Flt;

static void refresh(Flt *fl, int d, int a, int k, const double *row, const double *h)
{
    const int nt = d == 0 ? 0 : d - 1;
    double best[3] = {1e300, 1e300, 1e300}; int bj[3] = {-1, -1, -1};
    fl->H = 0ull;
    for (int j = 0; j < k; j++) {
        if (j == a) continue;
        double p;
        if (d == 0) p = row[j] > h[j] ? row[j] : h[j];
        else if (h[j] >= row[j]) { fl->H |= 1ull << j; continue; }
        else p = row[j];
        for (int q = 0; q <= nt; q++) if (p < best[q]) {
            for (int r = nt; r > q; r--) { best[r] = best[r - 1]; bj[r] = bj[r - 1]; }
            best[q] = p; bj[q] = j; break;
        }
    }
    fl->label = a; fl->m_rest = best[nt];
    fl->j[0] = fl->j[1] = -1;
    for (int q = 0; q < nt; q++) fl->j[q] = bj[q];
}
static int passes(const Flt *fl, int d, int a, double u, int k, const double *row, const double *h)
{
    if (fl->label != a) return 0;
    if (!(u <= fl->m_rest)) return 0;
    if (d == 0) return 1;
    for (int j = 0; j < k; j++) if (((fl->H >> j) & 1ull) && u > h[j]) return 0;
    for (int q = 0; q < d - 1; q++) { const int j = fl->j[q]; if (j >= 0 && u > row[j]) return 0; }
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s raw_u16 nrows nbands k iters\n", argv[0]); return 2; }
    size_t n = (size_t)atoll(argv[2]);
    int nb = atoi(argv[3]), k = atoi(argv[4]), iters = atoi(argv[5]);
    uint16_t *raw = malloc(n * nb * 2);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(raw, 2, n * nb, f) != n * nb) { fprintf(stderr, "read failed\n"); return 1; }
    fclose(f);
    double *X = malloc(sizeof(double) * n * nb), *mu = calloc(nb, sizeof(double));
    double *C = malloc(sizeof(double) * k * nb), *Cn = malloc(sizeof(double) * k * nb), *w = malloc(sizeof(double) * k);
    /* diagonalClusterCentres */
    for (int b = 0; b < nb; b++) {
        double s = 0.0; uint16_t mn = 65535, mx = 0;
        for (size_t i = 0; i < n; i++) { uint16_t v = raw[i * nb + b]; s += v; if (v < mn) mn = v; if (v > mx) mx = v; }
        mu[b] = s / (double)n;
        double step = (double)(mx - mn) / (double)(k + 1);
        for (int j = 0; j < k; j++) C[j * nb + b] = (double)(uint16_t)((double)mn + (double)(j + 1) * step) - mu[b];
    }
    for (size_t i = 0; i < n; i++) for (int b = 0; b < nb; b++) X[i * nb + b] = (double)raw[i * nb + b] - mu[b];
    int32_t *lab = malloc(sizeof(int32_t) * n);
    double *half = malloc(sizeof(double) * k * k), *next = malloc(sizeof(double) * k), *cshift = calloc(k, sizeof(double));
    double *ub = calloc(n, sizeof(double)), *lb = calloc(n * (size_t)k, sizeof(double));
    elk_half_distances(C, k, nb, half, next);
    elk_init_bounds(X, n, nb, C, k, half, lab, ub, lb);
    /* filter state per design: A uses m_rest only; B tracks j[0]; C tracks j[0], j[1] */
    Flt *F[NT];
    for (int d = 0; d < NT; d++) { F[d] = malloc(sizeof(Flt) * n); for (size_t i = 0; i < n; i++) { F[d][i].m_rest = -1.0; F[d][i].label = -1; F[d][i].H = 0; F[d][i].j[0] = F[d][i].j[1] = -1; } }
    printf("# it  cand%%  gateclosed%%  A%%  H%%  H+1%%  H+2%%   maxshift  medshift  changed\n");
    for (int it = 1; it <= iters; it++) {
        size_t ncand = 0, ngate = 0, npass[NT] = {0, 0, 0, 0}, nchanged = 0;
        for (size_t i = 0; i < n; i++) {
            const int a = lab[i];
            const double u = ub[i];
            const double *row = lb + i * k, *h = half + (size_t)a * k;
            int cand = 0;
            const int open = !(next[a] >= u);
            if (open) for (int j = 0; j < k; j++) if (j != a && u > row[j] && u > h[j]) { cand = 1; break; }
            ncand += cand; ngate += !open;
            for (int d = 0; d < NT; d++) {
                Flt *fl = &F[d][i];
                const int through = open && !passes(fl, d, a, u, k, row, h);
                if (through && !cand) refresh(fl, d, a, k, row, h);      /* the row was read: new summary */
                if (through && cand) fl->label = -1;                     /* refreshed after the visit */
                npass[d] += through;
            }
        }
        /* the E-step itself */
        int32_t *lab_old = malloc(sizeof(int32_t) * n); memcpy(lab_old, lab, sizeof(int32_t) * n);
        elk_estep(X, n, nb, C, k, half, next, lab, ub, lb);
        for (size_t i = 0; i < n; i++) nchanged += lab[i] != lab_old[i];
        free(lab_old);
        /* post-visit summaries for visited samples */
        for (size_t i = 0; i < n; i++) for (int d = 0; d < NT; d++) if (F[d][i].label < 0)
            refresh(&F[d][i], d, lab[i], k, lb + i * k, half + (size_t)lab[i] * k);
        /* M-step (row order) */
        memset(Cn, 0, sizeof(double) * k * nb); memset(w, 0, sizeof(double) * k);
        for (size_t i = 0; i < n; i++) { w[lab[i]] += 1.0; for (int b = 0; b < nb; b++) Cn[lab[i] * nb + b] += X[i * nb + b]; }
        for (int j = 0; j < k; j++) if (w[j] > 0.0) { double al = 1.0 / w[j]; for (int b = 0; b < nb; b++) Cn[j * nb + b] *= al; }
        double mx = 0.0, srt[1024];
        for (int j = 0; j < k; j++) { cshift[j] = elk_dist(Cn + j * nb, C + j * nb, nb); if (cshift[j] > mx) mx = cshift[j]; srt[j] = cshift[j]; }
        for (int a = 0; a < k; a++) for (int b = a + 1; b < k; b++) if (srt[b] < srt[a]) { double t = srt[a]; srt[a] = srt[b]; srt[b] = t; }
        for (size_t i = 0; i < n; i++) {
            ub[i] += cshift[lab[i]];
            for (int j = 0; j < k; j++) { lb[i * k + j] -= cshift[j]; if (lb[i * k + j] < 0) lb[i * k + j] = 0; }
            for (int d = 0; d < NT; d++) F[d][i].m_rest -= mx;
        }
        elk_half_distances(Cn, k, nb, half, next);
        memcpy(C, Cn, sizeof(double) * k * nb);
        printf("%3d  %6.2f  %6.2f  %6.2f %6.2f %6.2f %6.2f   %.4g %.4g  %zu\n", it, 100.0 * ncand / n, 100.0 * ngate / n,
               100.0 * npass[0] / n, 100.0 * npass[1] / n, 100.0 * npass[2] / n, 100.0 * npass[3] / n, mx, srt[k / 2], nchanged);
        fflush(stdout);
    }
    return 0;
}
