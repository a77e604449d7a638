// Cost of a taken branch for ONE wavefront as a function of the distance to its target: loops of
// N s_nop (4 bytes each) closed by a backward s_cbranch, and forward hops over M bytes of padding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2000
__global__ void k(unsigned long long *out)
{
    if (threadIdx.x >= 64) return;
    int slot = 0;
#define LOOP(N) { unsigned s = ITERS; unsigned long long t0 = __builtin_readcyclecounter(); \
        asm volatile("1:\n .rept " #N "\n s_nop 0\n .endr\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b" : "+s"(s) :: "scc"); \
        unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[slot] = t1 - t0; slot++; }
    LOOP(1) LOOP(4) LOOP(8) LOOP(12) LOOP(16) LOOP(24) LOOP(32) LOOP(48) LOOP(64) LOOP(96) LOOP(128) LOOP(256) LOOP(512)
#define HOP(M) { unsigned s = ITERS / 10; unsigned long long t0 = __builtin_readcyclecounter(); \
        asm volatile("1:\n .rept 10\n s_branch 2f\n .fill " #M ", 4, 0xbf800000\n 2:\n .endr\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b" : "+s"(s) :: "scc"); \
        unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[slot] = t1 - t0; slot++; }
    HOP(1) HOP(4) HOP(8) HOP(15) HOP(16) HOP(31) HOP(32) HOP(64) HOP(128)
}
int main()
{
    unsigned long long *d, h[32];
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int n[13] = {1, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 512};
    for (int i = 0; i < 13; i++) {
        const double per = (double)h[i] / ITERS;
        printf("loop of %3d s_nop + sub + cmp + branch (%4d bytes): %7.1f cycles/iter, %6.1f beyond 4.4/instr\n", n[i], (n[i] + 3) * 4, per, per - 4.4 * (n[i] + 3));
    }
    const int m[9] = {1, 4, 8, 15, 16, 31, 32, 64, 128};
    for (int i = 0; i < 9; i++) printf("forward hop over %3d dwords: %6.1f cycles per s_branch\n", m[i], (double)h[13 + i] / ITERS);
    return 0;
}
