// The same question as mfma_f64_order.hip for float32: does v_mfma_f32_16x16x4_f32 (K = 4) / v_mfma_f32_32x32x2_f32
// (K = 2) with B = 1.0 add its K values of A to C one after the other, each sum rounded to float32 (= the ordered
// `acc = acc + x` chain of buildSegmentSpectra)?  Random trials with wide exponent spread against every order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(const float *a, const float *c, float *d, int trials)
{
    for (int t = 0; t < trials; t++) {
        f32x4 cc;
        for (int r = 0; r < 4; r++) cc[r] = c[(t * 64 + threadIdx.x) * 4 + r];
        f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t * 64 + threadIdx.x], 1.0f, cc, 0, 0, 0);
        for (int r = 0; r < 4; r++) d[(t * 64 + threadIdx.x) * 4 + r] = dd[r];
    }
}
int main()
{
    const int T = 100000;
    std::vector<float> a((size_t)T * 64), c((size_t)T * 256), d((size_t)T * 256);
    std::mt19937_64 rng(9);
    auto rnd = [&]() {
        const int e = (int)(rng() % 40) - 10;
        float m = 1.0f + (float)(rng() >> 41) / 8388608.0f;
        if (rng() & 1) m = -m;
        return std::ldexp(m, e);
    };
    for (auto &v : a) v = rnd();
    for (auto &v : c) v = rnd();
    float *da, *dc, *dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&dc, c.size() * 4); hipMalloc(&dd, c.size() * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, da, dc, dd, T);
    hipMemcpy(d.data(), dd, c.size() * 4, hipMemcpyDeviceToHost);
    // C/D of 16x16x4 f32: lane l, register r -> row 4 (l >> 4) + r, column l & 15; A: lane l holds A[l & 15][k = l >> 4]
    int perm[4] = {0, 1, 2, 3}, okperms = 0;
    long exact_wide = 0, total = 0;
    do {
        long bad = 0;
        for (int t = 0; t < T && !bad; t++)
            for (int l = 0; l < 64 && !bad; l++)
                for (int r = 0; r < 4; r++) {
                    const int row = 4 * (l >> 4) + r;
                    float acc = c[((size_t)t * 64 + l) * 4 + r];
                    for (int q = 0; q < 4; q++) acc = acc + a[(size_t)t * 64 + row + 16 * perm[q]];
                    if (std::memcmp(&acc, &d[((size_t)t * 64 + l) * 4 + r], 4) != 0) { bad++; break; }
                }
        if (!bad) { printf("16x16x4 f32: order c + a[k=%d] + a[k=%d] + a[k=%d] + a[k=%d], each sum rounded to float32, matches all trials\n", perm[0], perm[1], perm[2], perm[3]); okperms++; }
    } while (std::next_permutation(perm, perm + 4));
    for (int t = 0; t < T; t++)
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int row = 4 * (l >> 4) + r;
                double w = c[((size_t)t * 64 + l) * 4 + r];
                for (int q = 0; q < 4; q++) w += a[(size_t)t * 64 + row + 16 * q];
                const float f = (float)w;
                exact_wide += std::memcmp(&f, &d[((size_t)t * 64 + l) * 4 + r], 4) == 0; total++;
            }
    if (!okperms) printf("16x16x4 f32: NO sequential float32 order matches; the float64 sum rounded once matches %ld of %ld results\n", exact_wide, total);
    return 0;
}
