// How fast can ONE wavefront run a dependent float64 addition chain (the k-means M-step's row-order sums)?
// (a) operands already in VGPRs, (b) operands in SGPR pairs, (c) operands streamed with s_load_dwordx16
// from global memory (8 doubles per load, loads issued AHEAD of use), (d) the LDS form of today's kernel
// (ds_read_b128: two doubles per read).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITERS 400
__global__ void k(unsigned long long *out, const double *__restrict__ g, double *sink)
{
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = g[i];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    double acc = 0.0;
    double v0 = g[threadIdx.x], v1 = g[64 + threadIdx.x];
    { unsigned long long t0 = __builtin_readcyclecounter();
      _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 32\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %2\n .endr" : "+v"(acc) : "v"(v0), "v"(v1));
      unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[0] = t1 - t0; }
    { double s0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x) + 1.5;
      unsigned long long t0 = __builtin_readcyclecounter();
      _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n v_add_f64 %0, %0, %1\n .endr" : "+v"(acc) : "s"(s0));
      unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[1] = t1 - t0; }
    { // (c) stream 8 doubles per s_load_dwordx16, two loads in flight ahead
      unsigned long long pp = (unsigned long long)g;
      unsigned long long t0 = __builtin_readcyclecounter();
      asm volatile(
          "s_load_dwordx16 s[40:55], s[36:37], 0x0\n"
          "s_load_dwordx16 s[56:71], s[36:37], 0x40\n"
          "s_add_u32 s36, s36, 0x80\n s_addc_u32 s37, s37, 0\n s_movk_i32 s38, 400\n"
          "1:\n"
          "s_waitcnt lgkmcnt(1)\n"
          "v_add_f64 %0, %0, s[40:41]\n v_add_f64 %0, %0, s[42:43]\n v_add_f64 %0, %0, s[44:45]\n v_add_f64 %0, %0, s[46:47]\n"
          "v_add_f64 %0, %0, s[48:49]\n v_add_f64 %0, %0, s[50:51]\n v_add_f64 %0, %0, s[52:53]\n v_add_f64 %0, %0, s[54:55]\n"
          "s_load_dwordx16 s[40:55], s[36:37], 0x0\n"
          "s_waitcnt lgkmcnt(1)\n"
          "v_add_f64 %0, %0, s[56:57]\n v_add_f64 %0, %0, s[58:59]\n v_add_f64 %0, %0, s[60:61]\n v_add_f64 %0, %0, s[62:63]\n"
          "v_add_f64 %0, %0, s[64:65]\n v_add_f64 %0, %0, s[66:67]\n v_add_f64 %0, %0, s[68:69]\n v_add_f64 %0, %0, s[70:71]\n"
          "s_load_dwordx16 s[56:71], s[36:37], 0x40\n"
          "s_add_u32 s36, s36, 0x80\n s_addc_u32 s37, s37, 0\n"
          "s_sub_u32 s38, s38, 1\n s_cmp_lg_u32 s38, 0\n s_cbranch_scc1 1b\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "+v"(acc), "+{s[36:37]}"(pp) :
          : "s38", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
            "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "scc", "memory");
      unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[2] = t1 - t0; }     // 400 * 16 adds
    { // (d) LDS: 8 x ds_read_b128 then 16 adds, as k_fit_sum_lists_staged
      const double *p = lds + (threadIdx.x & 7) * 16;
      unsigned long long t0 = __builtin_readcyclecounter();
      for (int i = 0; i < ITERS; i++) {
          double2 t[8];
#pragma unroll
          for (int u = 0; u < 8; u++) t[u] = *(const double2 *)(p + ((i * 16 + 2 * u) & 1023));
#pragma unroll
          for (int u = 0; u < 8; u++) { acc += t[u].x; acc += t[u].y; }
      }
      unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[3] = t1 - t0; }
    sink[threadIdx.x] = acc;
}
int main()
{
    unsigned long long *d, h[4]; double *g, *sink;
    std::vector<double> hg(1 << 20, 1.25);
    hipMalloc(&d, 32); hipMalloc(&g, hg.size() * 8); hipMalloc(&sink, 64 * 8);
    hipMemcpy(g, hg.data(), hg.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, g, sink); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("dependent v_add_f64, VGPR operands:      %6.2f cycles/add\n", (double)h[0] / (ITERS * 64.0));
    printf("dependent v_add_f64, SGPR-pair operand:  %6.2f cycles/add\n", (double)h[1] / (ITERS * 64.0));
    printf("s_load_dwordx16 stream + 16 adds/iter:   %6.2f cycles/add\n", (double)h[2] / (400 * 16.0));
    printf("ds_read_b128 x8 + 16 adds (today):       %6.2f cycles/add\n", (double)h[3] / (ITERS * 16.0));
    return 0;
}
