// gridbar.h -- software grid barriers for persistent kernels (the pass loop of elim_small.h).  CTL is the kernel's control block in device memory; it must have the
// zero-initialised fields  uint32_t bar_count, bar_gen, fail, xcd_n[16], xcd_cnt[16].
// Agent-scope release / acquire as MI355X_MICROARCH.md and cdna_hip_programming.md Guideline 16
// prescribe; every spin is bounded: on a timeout the barrier sets ctl->fail and returns false in
// every workgroup, and the kernel leaves.  All workgroups of the launch must be co-resident.
#pragma once
#include "common.h"

#define GRID_SPIN_LIMIT (1u << 23)

template <class CTL>
__device__ __forceinline__ bool small_grid_barrier(CTL *ctl, uint32_t nblocks, int poll = 4)
{
    __shared__ uint32_t s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ok = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t gen = __hip_atomic_load(&ctl->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t arrived = __hip_atomic_fetch_add(&ctl->bar_count, 1u, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == nblocks) {
            __hip_atomic_store(&ctl->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&ctl->bar_gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            uint32_t spins = 0;
            // poll gently: 1280 workgroups of 20 concurrent loops polling device-coherent lines is
            // fabric traffic every other kernel pays for; the fail flag is looked at now and then
            while (__hip_atomic_load(&ctl->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                for (int q = 0; q < poll; q++) __builtin_amdgcn_s_sleep(8);
                if (++spins > GRID_SPIN_LIMIT ||
                    ((spins & 15u) == 0u &&
                     __hip_atomic_load(&ctl->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&ctl->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

// Two-level form of the grid barrier.  MI355X has eight XCDs, each with its own L2: the release
// side of an agent-scope barrier is an L2 write-back (buffer_wbl2) of the whole XCD -- the data of
// every other kernel running there included -- and the one-level barrier above executes it once per
// workgroup (65 per barrier, ~150 barriers per tile, 20 tiles in flight).  Here the workgroups of
// one XCD first meet on a per-XCD counter (their stores have drained into that XCD's L2 by then) and
// only the last one to arrive writes the L2 back and goes on to the global counter: 8 write-backs
// per barrier.  The acquire side (invalidate) stays per workgroup.
#define SMALL_GETREG_XCC_ID ((3u << 11) | 20u)      // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4)
struct SmallBar { uint32_t xcd, nx, nactive; int poll; };

template <class CTL>
__device__ __forceinline__ bool small_grid_barrier2(CTL *ctl, const SmallBar &b)
{
    __shared__ uint32_t s_ok2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores are in this XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ok = 1;
        const uint32_t gen = __hip_atomic_load(&ctl->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t a = __hip_atomic_fetch_add(&ctl->xcd_cnt[b.xcd], 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (a == b.nx) {                                      // last workgroup of this XCD
            __hip_atomic_store(&ctl->xcd_cnt[b.xcd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                // one L2 write-back per XCD
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t g = __hip_atomic_fetch_add(&ctl->bar_count, 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (g == b.nactive) {
                __hip_atomic_store(&ctl->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&ctl->bar_gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        uint32_t spins = 0;
        while (__hip_atomic_load(&ctl->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
            for (int q = 0; q < b.poll; q++) __builtin_amdgcn_s_sleep(8);
            if (++spins > GRID_SPIN_LIMIT ||
                ((spins & 15u) == 0u &&
                 __hip_atomic_load(&ctl->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ok2 = ok;
    }
    __syncthreads();
    return s_ok2 != 0;
}


