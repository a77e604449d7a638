// k_fit_sum_lists_staged (the k-means M-step's row-order sums) on a synthetic sample of the benchmark's shape:
// 1 032 256 rows x 6 bands, 60 clusters of random membership.  Prints the kernel's duration and, from the
// adding wavefront's cycle counter, how long it added and how long it stood at the barriers waiting for
// the gatherers (cluster 0 and the largest cluster).   hipcc -DFIT_SUM_DIAG ... -I pyshepseg_amd/csrc
#define FIT_SUM_DIAG 1
#include "common.h"
#include <utility>
#include "scan.h"
#include "sort.h"
#include "kmeans.h"
#include <cstdio>
#include <vector>
#include <random>
int main()
{
    const uint32_t n = 1032256; const int nb = 6, k = 60;
    std::mt19937 rng(5);
    std::vector<double> X((size_t)n * nb);
    for (auto &v : X) v = (double)(rng() % 60000) - 20000.25;
    std::vector<uint32_t> lab(n), rows(n), off(k + 1, 0);
    for (auto &l : lab) l = rng() % k;
    for (uint32_t i = 0; i < n; i++) off[lab[i] + 1]++;
    for (int j = 0; j < k; j++) off[j + 1] += off[j];
    { std::vector<uint32_t> fill(off.begin(), off.end() - 1); for (uint32_t i = 0; i < n; i++) rows[fill[lab[i]]++] = i; }
    double *dX, *dS, *dcnt; uint32_t *drows, *doff;
    hipMalloc(&dX, X.size() * 8); hipMalloc(&dS, k * nb * 8); hipMalloc(&dcnt, k * 8);
    hipMalloc(&drows, n * 4); hipMalloc(&doff, (k + 1) * 4);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(drows, rows.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, off.data(), (k + 1) * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++)
            hipLaunchKernelGGL(k_fit_sum_lists_staged, dim3(k), dim3(FIT_SUM_THREADS), 0, 0, dX, nb, drows, doff, dS, dcnt, nullptr, FitDigits{nullptr, nullptr, 0u, n}, ElkEpilogue{});
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_fit_sum_lists_staged: %.1f us per launch\n", ms * 1000.f / 20.f);
    }
    {   // the same behind a 1 GB write (the E-step streams 0.96 GB of bounds between two M-steps)
        void *big; hipMalloc(&big, 1u << 30);
        float tot = 0.f;
        for (int i = 0; i < 10; i++) {
            hipMemsetAsync(big, i, 1u << 30, 0);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fit_sum_lists_staged, dim3(k), dim3(FIT_SUM_THREADS), 0, 0, dX, nb, drows, doff, dS, dcnt, nullptr, FitDigits{nullptr, nullptr, 0u, n}, ElkEpilogue{});
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
        }
        printf("k_fit_sum_lists_staged behind a 1 GB memset: %.1f us per launch\n", tot * 1000.f / 10.f);
    }
    std::vector<double> S(k * nb);
    hipMemcpy(S.data(), dS, k * nb * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < k; j++) for (int b = 0; b < nb; b++) {
        double a = 0.0; for (uint32_t q = off[j]; q < off[j + 1]; q++) a += X[(size_t)rows[q] * nb + b];
        if (a != S[j * nb + b]) bad++;
    }
    printf("sums differing from the row-order sums of the host: %d of %d\n", bad, k * nb);
    unsigned long long d[256 * 4];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_fit_diag), sizeof d);
    for (int j = 0; j < k; j += 59)
        printf("cluster %d: %llu rows in %llu blocks, %llu cycles adding (%.1f per row), %llu at barriers\n", j, d[j * 4 + 3],
               d[j * 4 + 2], d[j * 4], (double)d[j * 4] / (double)d[j * 4 + 3], d[j * 4 + 1]);
    return 0;
}
