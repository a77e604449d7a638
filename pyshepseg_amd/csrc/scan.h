// scan.h -- exclusive prefix sum of uint32 values produced by a device functor.
// Three-phase (block-local scan, recursive scan of block totals, add-back); 8192 items per
// 256-thread workgroup.  HBM-bound: 4 B read (whatever the functor reads) + 4 B written/item.
#pragma once
#include "common.h"
#include <type_traits>

#define SCAN_SUB 1024u                       // items per inner step of a workgroup (4 per thread)
#define SCAN_STEPS 8u
#define SCAN_ITEMS (SCAN_SUB * SCAN_STEPS)   // items per workgroup: two levels cover 67 M items, so the
                                             // scans of a tile need 2 + 1 launches instead of 3 + 2

// A functor may offer get4(base, v): its four items base .. base + 3 (base a multiple of 4, all below n) from ONE
// 16-byte load -- 4-byte accesses per lane stream at ~2.8 TB/s on this chip, 16-byte ones at ~4.5 (k_relabel: 47 -> 31 us)
template <class F, class = void> struct scan_has_get4 : std::false_type {};
template <class F> struct scan_has_get4<F, std::void_t<decltype(&F::get4)>> : std::true_type {};
__device__ __forceinline__ bool scan_load4(const uint32_t *a, uint32_t base, uint32_t v[4])
{
    if (((uintptr_t)a & 15u) != 0u) return false;
    const uint4 x = *(const uint4 *)(a + base);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    return true;
}

// block-local exclusive scan of SCAN_ITEMS items (those of block `blk`) in SCAN_STEPS coalesced
// sub-tiles with a running carry; returns the block's total (in every thread)
template <class F>
__device__ __forceinline__ uint32_t scan_block(F f, uint32_t n, uint32_t *__restrict__ out, uint32_t blk)
{
    __shared__ uint32_t wsum[4];
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t st = 0; st < SCAN_STEPS; st++) {
        const uint32_t base = blk * SCAN_ITEMS + st * SCAN_SUB + threadIdx.x * 4u;
        if (st * SCAN_SUB + blk * SCAN_ITEMS >= n) break;                 // uniform: nothing left
        uint32_t v[4];
        bool got = false;
        if constexpr (scan_has_get4<F>::value) {
            if (base + 3u < n) got = f.get4(base, v);
        }
        if (!got) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = (base + i < n) ? f(base + i) : 0u;
        }
        const uint32_t tsum = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = tsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= (unsigned)d) incl += t;
        }
        __syncthreads();                                       // wsum of the previous step was read
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t woff = carry;
        for (unsigned i = 0; i < w; i++) woff += wsum[i];
        uint32_t run = woff + incl - tsum;
        if (base + 3u < n && ((uintptr_t)out & 15u) == 0u) {
            *(uint4 *)(out + base) = make_uint4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (base + i < n) out[base + i] = run;
                run += v[i];
            }
        }
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    return carry;
}

struct AgentArrFn {     // block totals written by other workgroups of the same launch
    const uint32_t *a;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return L2LOAD(&a[i]); }
};

// One launch for up to SCAN_ITEMS workgroups (67 M items): every workgroup scans its own items and
// publishes its total (agent-scope store, drained before it takes a ticket from `ctr`); the workgroup
// that takes the last ticket -- every total is in memory by then, and nobody waits for anybody --
// scans the totals into boff[], writes the grand total and leaves `ctr` at zero for the next scan.
// bsum[block] = the block's total; total_out (optional) = the sum of all.
template <class F>
__global__ __launch_bounds__(256) void k_scan_local(F f, uint32_t n, uint32_t *__restrict__ out,
                                                    uint32_t *bsum, uint32_t *total_out,
                                                    uint32_t *boff, uint32_t *ctr, uint32_t *total_host)
{
    __shared__ uint32_t s_last;
    const uint32_t carry = scan_block(f, n, out, blockIdx.x);
    const uint32_t nb = gridDim.x;
    if (nb == 1u) {
        if (threadIdx.x == 0) {
            bsum[0] = carry;
            if (total_out) *total_out = carry;
            if (total_host) MIRROR_STORE(total_host, carry);
        }
        return;
    }
    if (threadIdx.x == 0) {
        uint32_t last = 0;
        if (boff) {
            __hip_atomic_store(&bsum[blockIdx.x], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = (atomicAdd(ctr, 1u) == nb - 1u) ? 1u : 0u;
        } else {
            bsum[blockIdx.x] = carry;           // the totals are scanned by a second launch
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    AgentArrFn g{bsum};
    const uint32_t tot = scan_block(g, nb, boff, 0u);
    if (threadIdx.x == 0) {
        if (total_out) *total_out = tot;
        if (total_host) MIRROR_STORE(total_host, tot);
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t *__restrict__ out, uint32_t n,
                                                  const uint32_t *__restrict__ boff)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] += boff[i / SCAN_ITEMS];
}

struct ArrFn {
    const uint32_t *a;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return a[i]; }
    __device__ __forceinline__ bool get4(uint32_t base, uint32_t v[4]) const { return scan_load4(a, base, v); }
};

// bytes of scratch needed for scanning n items
static inline size_t scan_tmp_bytes(size_t n)
{
    size_t tot = 0;
    while (true) {
        size_t nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
        tot += 2 * nb + 2;
        if (nb <= 1) break;
        n = nb;
    }
    return (tot + 16) * sizeof(uint32_t);
}

// out[i] = sum_{j<i} f(j); *total_dev (optional, device pointer) = sum of all.
// tmp: device scratch of scan_tmp_bytes(n).
// lazy_boff (optional): the add-back pass is skipped and *lazy_boff receives the per-block offsets
// (nullptr when there is a single block): the consumer adds lazy_boff[i / SCAN_ITEMS] to out[i]
// itself, one launch less on a stream where every launch queues behind other tiles' kernels.
// total_host (optional): a word of the context's pinned block (see PIN_MIRROR) that receives the
// total as well, readable after the next stream synchronisation.
template <class F>
static int scan_exclusive(shp_ctx *ctx, F f, uint32_t n, uint32_t *out, uint32_t *total_dev,
                          uint32_t *tmp, const uint32_t **lazy_boff = nullptr, uint32_t *total_host = nullptr)
{
    if (lazy_boff) *lazy_boff = nullptr;
    if (n == 0) {
        if (total_dev) HIPCHK(ctx, hipMemsetAsync(total_dev, 0, 4, ctx->stream));
        if (total_host) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); *total_host = 0; }
        return 0;
    }
    const uint32_t nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    uint32_t *bsum = tmp;
    uint32_t *boff = tmp + nb + 1;
    static const int one_env = getenv("SHEPSEG_SCAN_ONE") ? atoi(getenv("SHEPSEG_SCAN_ONE")) : 1;
    const bool one_launch = one_env && nb > 1 && nb <= SCAN_ITEMS;    // the last workgroup scans the totals
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_local<F>), dim3(nb), dim3(256), 0, ctx->stream, f, n,
                       out, bsum, (nb == 1 || one_launch) ? total_dev : (uint32_t *)nullptr,
                       one_launch ? boff : (uint32_t *)nullptr, ctx->scan_ctr,
                       (nb == 1 || one_launch) ? total_host : (uint32_t *)nullptr);
    KCHK(ctx);
    if (nb == 1) return 0;                   // the kernel wrote the total itself
    if (!one_launch) {
        ArrFn g{bsum};
        CHK(scan_exclusive(ctx, g, nb, boff, total_dev, tmp + 2 * (size_t)nb + 2, nullptr, total_host));
    }
    if (lazy_boff) { *lazy_boff = boff; return 0; }
    hipLaunchKernelGGL(k_scan_add, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, out, n, boff);
    KCHK(ctx);
    return 0;
}
