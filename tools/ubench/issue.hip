// Micro-benchmarks of instruction issue for ONE wavefront (the replay walkers are lone latency-bound
// waves): cycles per instruction for dependent / independent scalar and vector chains, taken and
// not-taken branches, VGPR <-> SGPR crossings, LDS round trips.  hipcc --offload-arch=gfx950 issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITERS 200
#define T0() unsigned long long t0 = __builtin_readcyclecounter()
#define T1(slot) do { unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[blockIdx.x * 32 + slot] = t1 - t0; } while (0)

__global__ void k(unsigned long long *out, uint32_t *sink, int nwaves_active)
{
    __shared__ uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = ((i + 17) & 1023) * 4;
    __syncthreads();
    if ((int)(threadIdx.x >> 6) >= nwaves_active) return;
    uint32_t s = __builtin_amdgcn_readfirstlane(threadIdx.x), s2 = 1, s3 = 2, s4 = 3;
    uint32_t v = threadIdx.x, v2 = 5;
    unsigned long long q = 0x123456789abcull;
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n .endr" : "+s"(s) :: "scc"); T1(0); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 16\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n .endr" : "+s"(s), "+s"(s2), "+s"(s3), "+s"(s4) :: "scc"); T1(1); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n v_add_u32 %0, %0, 1\n .endr" : "+v"(v)); T1(2); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 32\n v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n .endr" : "+v"(v), "+v"(v2)); T1(3); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_branch 1f\n s_nop 0\n 1:\n .endr" ::: "memory", "scc"); T1(4); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n .endr" :: "s"(s) : "scc"); T1(5); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n 1:\n .endr" :: "s"(s) : "scc"); T1(6); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_lshr_b64 %0, %0, 1\n s_lshl_b64 %0, %0, 1\n .endr" : "+s"(q) :: "scc"); T1(7); }          // 128 instrs
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n v_readlane_b32 %0, %1, 3\n s_add_u32 %0, %0, 1\n v_mov_b32 %1, %0\n .endr" : "+s"(s), "+v"(v) :: "scc"); T1(8); }   // 192 instrs
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n v_cmp_eq_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n .endr" : "+v"(v) : "s"(s), "v"(v2), "v"(v2) : "vcc", "scc"); T1(9); }  // 128
    { uint32_t a = (threadIdx.x & 63) * 4; T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(a) :: "memory", "scc"); T1(10); v += a; }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_bitcmp1_b64 %0, 5\n s_cselect_b32 %1, 1, 0\n .endr" : "+s"(q), "+s"(s2) :: "scc"); T1(11); }   // 128
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n .endr" : "+s"(s), "+v"(v) :: "scc"); T1(12); }      // 128 independent s/v interleaved
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n v_readlane_b32 %0, %1, %2\n .endr" : "=s"(s3) : "v"(v), "s"(s2 & 63) : "scc"); T1(13); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_nop 0\n .endr" ::: "scc"); T1(14); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n .endr" : "+s"(s) :: "scc"); T1(15); }  // dependent add+cmp+taken branch: 192 issued
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_and_b32 %0, %0, %1\n s_or_b32 %0, %0, %2\n .endr" : "+s"(s) : "s"(s2), "s"(s3) : "scc"); T1(16); }   // 128 dependent
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n v_mov_b32 %1, %0\n .endr" : "+s"(s), "+v"(v) :: "scc"); T1(17); }   // salu -> valu reads it: 128
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile("s_getpc_b64 s[40:41]\n .rept 64\n s_add_u32 s40, s40, 20\n s_addc_u32 s41, s41, 0\n s_setpc_b64 s[40:41]\n s_nop 0\n s_nop 0\n .endr" ::: "s40", "s41", "scc"); T1(18); }   // 192: add/addc/setpc (taken, +8 bytes skipped)
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile("s_mov_b32 m0, 5\n .rept 64\n v_writelane_b32 %0, %1, m0\n .endr" : "+v"(v) : "s"(s) : "scc"); T1(19); }
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_bfe_u64 s[40:41], %0, 0x30005\n s_lshl1_add_u32 s40, s40, s40\n .endr" :: "s"(q) : "s40", "s41", "scc"); T1(20); }  // 128
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n s_sub_u32 m0, %1, 2\n v_writelane_b32 %0, %1, m0\n v_readlane_b32 s40, %0, %1\n s_bfe_u64 s[42:43], s[40:41], 0x10003\n .endr" : "+v"(v) : "s"(s2 & 31) : "s40", "s41", "s42", "s43", "scc"); T1(21); }  // 256: vertical-move chain
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile(".rept 64\n v_add_u32 %0, 1, %0\n ds_write_b32 %1, %0\n .endr" : "+v"(v) : "v"(v2 & 1020) : "memory", "scc"); T1(22); }  // 128: push
    { T0(); _Pragma("nounroll") for (int i = 0; i < ITERS; i++) asm volatile("s_getpc_b64 s[40:41]\n .rept 64\n s_add_u32 s40, s40, 0x10c\n s_addc_u32 s41, s41, 0\n s_setpc_b64 s[40:41]\n .fill 64, 4, 0xbf800000\n .endr" ::: "s40", "s41", "scc"); T1(23); }   // 192: setpc jumping 256 bytes ahead each time
    if (threadIdx.x == 0) sink[blockIdx.x] = s + s2 + s3 + s4 + v + v2 + (uint32_t)q;
}

int main()
{
    unsigned long long *d; uint32_t *sink;
    hipMalloc(&d, 32 * 8 * 64); hipMalloc(&sink, 4 * 64);
    const char *names[24] = {"dep s_add x64", "4 indep s_add chains x64", "dep v_add x64", "2 indep v_add x64", "s_branch taken x64",
        "s_cmp+cbranch taken x64 (128 instr)", "s_cmp+cbranch not taken x64 (128)", "dep s_lshr/lshl_b64 x128", "readlane->s_add->v_mov x192",
        "v_cmp+v_cndmask x128", "ds_read dep chain x64", "s_bitcmp1_b64+cselect x128", "indep s_add/v_add interleaved x128", "v_readlane sgpr idx x64",
        "s_nop 0 x64", "dep s_add+s_cmp+taken branch x192", "dep s_and/s_or x128", "s_add -> v_mov x128", "s_add+s_addc+s_setpc (short hop) x192", "v_writelane m0 x64", "s_bfe_u64+s_lshl1_add x128", "m0/writelane/readlane/bfe chain x256", "v_add+ds_write x128", "s_add+s_addc+s_setpc (256-byte hop) x192"};
    const int ninstr[24] = {64, 64, 64, 64, 64, 128, 128, 128, 192, 128, 64, 128, 128, 64, 64, 192, 128, 128, 192, 64, 128, 256, 128, 192};
    for (int cfg = 0; cfg < 2; cfg++) {
        const int waves = cfg == 0 ? 1 : cfg == 1 ? 4 : 8;         // waves per workgroup (one CU): 1, 1 per SIMD, 2 per SIMD
        hipMemset(d, 0, 32 * 8 * 64);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, sink, waves);
        hipDeviceSynchronize();
        unsigned long long h[32];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("== %d active wave(s) in one workgroup\n", waves);
        for (int i = 0; i < 24; i++) printf("  %-40s %7.2f cycles/instr\n", names[i], (double)h[i] / (ITERS * (double)ninstr[i]));
    }
    return 0;
}
