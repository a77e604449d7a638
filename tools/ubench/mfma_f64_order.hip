// In which order does v_mfma_f64_4x4x4_4b_f64 add?  With B = 1.0 everywhere the instruction computes, for every
// output lane, C + the four A values of its (block, row); if the hardware accumulates them as four IEEE
// additions one after the other in ascending k, the instruction is a 4-step slice of a row-order sum
// (k_fit_sum_lists_staged).  Step 1 finds which four A lanes reach which output lane (one-hot A), step 2 runs
// random trials with wide exponent spread and cancellation and checks every permutation of the four against
// the hardware's bits.  Step 3 times a dependent chain of the instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
__global__ void k_once(const double *a, const double *b, const double *c, double *d, int trials)
{
    for (int t = 0; t < trials; t++)
        d[t * 64 + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + threadIdx.x], b[t * 64 + threadIdx.x],
                                                                     c[t * 64 + threadIdx.x], 0, 0, 0);
}
__global__ void k_chain(unsigned long long *out, double *sink, double a0)
{
    double acc = 0.0, a = a0 + threadIdx.x, one = 1.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 1000; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, one, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main()
{
    const int T1 = 64, T2 = 200000, T = T1 + T2;
    std::vector<double> a((size_t)T * 64, 0.0), b((size_t)T * 64, 1.0), c((size_t)T * 64, 0.0), d((size_t)T * 64);
    for (int t = 0; t < T1; t++) a[(size_t)t * 64 + t] = 1.0;                       // one-hot A
    std::mt19937_64 rng(7);
    auto rnd = [&]() {
        const int e = (int)(rng() % 60) - 30;
        double m = 1.0 + (double)(rng() >> 11) / 9007199254740992.0;
        if (rng() & 1) m = -m;
        return std::ldexp(m, e);
    };
    for (int t = T1; t < T; t++)
        for (int l = 0; l < 64; l++) {
            a[(size_t)t * 64 + l] = rnd(); c[(size_t)t * 64 + l] = rnd();
            if (t % 3 == 0 && (l & 4)) a[(size_t)t * 64 + l] = -a[(size_t)t * 64 + (l ^ 4)] * (1.0 + 1e-15 * (rng() % 7));   // near cancellation
        }
    double *da, *db, *dc, *dd;
    hipMalloc(&da, a.size() * 8); hipMalloc(&db, a.size() * 8); hipMalloc(&dc, a.size() * 8); hipMalloc(&dd, a.size() * 8);
    hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), a.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), a.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, da, db, dc, dd, T);
    hipMemcpy(d.data(), dd, a.size() * 8, hipMemcpyDeviceToHost);
    // step 1: contributors of every output lane
    int src[64][4], nsrc[64] = {0};
    for (int t = 0; t < T1; t++)
        for (int o = 0; o < 64; o++)
            if (d[(size_t)t * 64 + o] != 0.0 && nsrc[o] < 4) src[o][nsrc[o]++] = t;
    printf("A lanes reaching output lane o (B = 1):\n");
    for (int o = 0; o < 64; o++) {
        printf("  o=%2d <-", o);
        for (int q = 0; q < nsrc[o]; q++) printf(" %2d", src[o][q]);
        printf(o % 4 == 3 ? "\n" : "   ");
    }
    // step 2: which order?
    int perm[4] = {0, 1, 2, 3}, nperm = 0, okperms = 0;
    do {
        long bad = 0;
        for (int t = T1; t < T && bad == 0; t++)
            for (int o = 0; o < 64; o++) {
                if (nsrc[o] != 4) { bad++; break; }
                double r = c[(size_t)t * 64 + o];
                for (int q = 0; q < 4; q++) r = r + a[(size_t)t * 64 + src[o][perm[q]]];
                if (std::memcmp(&r, &d[(size_t)t * 64 + o], 8) != 0) { bad++; break; }
            }
        if (bad == 0) { printf("order (c + a[%d]) + a[%d] + a[%d] + a[%d] of the contributors listed above matches all %d trials x 64 lanes\n", perm[0], perm[1], perm[2], perm[3], T2); okperms++; }
        nperm++;
    } while (std::next_permutation(perm, perm + 4));
    if (!okperms) printf("NO sequential order matches: the instruction does not add one value after the other in float64\n");
    unsigned long long *dout, cyc; double *dsink;
    hipMalloc(&dout, 8); hipMalloc(&dsink, 64 * 8);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dout, dsink, 0.5);
    hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost);
    printf("dependent chain: %.1f cycles per v_mfma_f64_4x4x4_4b_f64 (= 4 rows of 16 chains)\n", (double)cyc / 16000.0);
    return 0;
}
