#include <math.h>
#include <stdint.h>
#include <stddef.h>
// E-step variants: labels of n rows (nb cols, float64) against k centres; cn = provided squared norms
void assign_variant(const double *X, size_t n, int nb, const double *C, int k, const double *cn, int variant, int32_t *lab)
{
    for (size_t i = 0; i < n; i++) {
        int best = 0; double bd = 0.0;
        for (int j = 0; j < k; j++) {
            double d;
            if (variant == 0) {           // oracle: fma chain seeded with cn, operand -2c
                d = cn[j];
                for (int b = 0; b < nb; b++) d = fma(X[i * nb + b], -2.0 * C[j * nb + b], d);
            } else if (variant == 1) {    // gemm-like: dot by fma from 0, then fma(-2, dot, cn)
                double dot = 0.0;
                for (int b = 0; b < nb; b++) dot = fma(X[i * nb + b], C[j * nb + b], dot);
                d = fma(-2.0, dot, cn[j]);
            } else if (variant == 2) {    // dot by fma, then cn + (-2*dot) rounded separately
                double dot = 0.0;
                for (int b = 0; b < nb; b++) dot = fma(X[i * nb + b], C[j * nb + b], dot);
                double t = -2.0 * dot; d = cn[j] + t;
            } else if (variant == 3) {    // dot mul+add (no fma), then fma(-2,dot,cn)
                double dot = 0.0;
                for (int b = 0; b < nb; b++) { double p = X[i * nb + b] * C[j * nb + b]; dot = dot + p; }
                d = fma(-2.0, dot, cn[j]);
            } else if (variant == 4) {    // alpha folded into A: dot of (-2x)*c by fma from 0, + cn
                double dot = 0.0;
                for (int b = 0; b < nb; b++) dot = fma(-2.0 * X[i * nb + b], C[j * nb + b], dot);
                d = cn[j] + dot;
            } else {                      // dot accumulated in reverse order
                double dot = 0.0;
                for (int b = nb - 1; b >= 0; b--) dot = fma(X[i * nb + b], C[j * nb + b], dot);
                d = fma(-2.0, dot, cn[j]);
            }
            if (j == 0 || d < bd) { bd = d; best = j; }
        }
        lab[i] = best;
    }
}
// squared-norm variants
void cn_variant(const double *C, int k, int nb, int variant, double *cn)
{
    for (int j = 0; j < k; j++) {
        double s = 0.0;
        if (variant == 0) for (int b = 0; b < nb; b++) s = fma(C[j*nb+b], C[j*nb+b], s);
        else if (variant == 1) for (int b = 0; b < nb; b++) { double p = C[j*nb+b]*C[j*nb+b]; s = s + p; }
        else if (variant == 2) for (int b = nb-1; b >= 0; b--) { double p = C[j*nb+b]*C[j*nb+b]; s = s + p; }
        else for (int b = nb-1; b >= 0; b--) s = fma(C[j*nb+b], C[j*nb+b], s);
        cn[j] = s;
    }
}
// numpy einsum 'ij,ij->i' baseline SSE2 (2 lanes, mul then add), 4-vector unrolled blocks in reverse
void cn_sse2(const double *C, int k, int nb, double *cn)
{
    for (int j = 0; j < k; j++) {
        const double *c = C + (size_t)j * nb;
        double a0 = 0.0, a1 = 0.0;
        int i = 0, count = nb;
        for (; count >= 8; count -= 8, i += 8) {
            double p;
            p = c[i+6]*c[i+6]; double t0 = p + a0; p = c[i+7]*c[i+7]; double t1 = p + a1;
            p = c[i+4]*c[i+4]; t0 = p + t0; p = c[i+5]*c[i+5]; t1 = p + t1;
            p = c[i+2]*c[i+2]; t0 = p + t0; p = c[i+3]*c[i+3]; t1 = p + t1;
            p = c[i+0]*c[i+0]; a0 = p + t0; p = c[i+1]*c[i+1]; a1 = p + t1;
        }
        for (; count > 0; count -= 2, i += 2) {
            double p = c[i]*c[i]; a0 = p + a0;
            double q = (count > 1) ? c[i+1]*c[i+1] : 0.0; a1 = q + a1;
        }
        cn[j] = a0 + a1;
    }
}
