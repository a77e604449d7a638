// csr.h -- pixels grouped by segment id, raster order inside a segment (the reference's segLoc at
// entry, shepseg.py:880), built from RUNS instead of pixels.  A run is a maximal sequence of
// consecutive linear pixel indices carrying one segment id, cut at multiples of 64 (so that a
// wavefront sees whole runs).  Runs come out of the raster in ascending start index; a stable
// sort of the runs by segment id therefore leaves every segment's runs in raster order, and
// expanding the sorted runs gives the same array as the stable sort of all pixels -- with 3-4x
// fewer items through the radix sort.  A run is one 32-bit value: start (26 bits) | (len-1) << 26.
#pragma once
#include "common.h"
#include "scan.h"
#include "sort.h"

#define RUN_TILE 4096u          // pixels per workgroup: 4 wavefronts x 16 rows of 64
#define RUN_ROWS 16u
#define RUN_POS_BITS 26
#define RUN_POS_MASK ((1u << RUN_POS_BITS) - 1u)

struct RunLenFn {
    const uint32_t *v;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return (v[i] >> RUN_POS_BITS) + 1u; }
    __device__ __forceinline__ bool get4(uint32_t base, uint32_t o[4]) const
    {
        if (!scan_load4(v, base, o)) return false;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) o[i] = (o[i] >> RUN_POS_BITS) + 1u;
        return true;
    }
};

__global__ __launch_bounds__(256) void k_run_tile_count(const uint32_t *__restrict__ seg, uint32_t n,
                                                        uint32_t *__restrict__ bcount,
                                                        uint32_t *zero, uint32_t nzero)
{
    __shared__ uint32_t wc[4];
    if (blockIdx.x == 0)                    // a caller's small array to clear (saves a fill launch)
        for (uint32_t q = threadIdx.x; q < nzero; q += 256u) zero[q] = 0u;
    const unsigned w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t base = blockIdx.x * RUN_TILE + w * (RUN_ROWS * 64u) + lane;
    uint32_t s[RUN_ROWS];
#pragma unroll
    for (unsigned r = 0; r < RUN_ROWS; r++) {
        const uint32_t i = base + r * 64u;
        s[r] = i < n ? seg[i] : 0xFFFFFFFFu;
    }
    uint32_t cnt = 0;
#pragma unroll
    for (unsigned r = 0; r < RUN_ROWS; r++) {
        const uint32_t prev = __shfl_up(s[r], 1, 64);
        const bool flag = (base + r * 64u) < n && (lane == 0 || s[r] != prev);
        cnt += (uint32_t)__popcll(__ballot(flag));
    }
    if (lane == 0) wc[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ __launch_bounds__(256) void k_run_tile_emit(const uint32_t *__restrict__ seg, uint32_t n,
                                                       const uint32_t *__restrict__ boff,
                                                       uint32_t *__restrict__ rkeys,
                                                       uint32_t *__restrict__ rvals)
{
    __shared__ uint32_t wc[4];
    const unsigned w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t base = blockIdx.x * RUN_TILE + w * (RUN_ROWS * 64u) + lane;
    uint32_t s[RUN_ROWS];
#pragma unroll
    for (unsigned r = 0; r < RUN_ROWS; r++) {
        const uint32_t i = base + r * 64u;
        s[r] = i < n ? seg[i] : 0xFFFFFFFFu;
    }
    unsigned long long m[RUN_ROWS];
    uint32_t cnt = 0;
#pragma unroll
    for (unsigned r = 0; r < RUN_ROWS; r++) {
        const uint32_t prev = __shfl_up(s[r], 1, 64);
        const bool flag = (base + r * 64u) < n && (lane == 0 || s[r] != prev);
        m[r] = __ballot(flag);
        cnt += (uint32_t)__popcll(m[r]);
    }
    if (lane == 0) wc[w] = cnt;
    __syncthreads();
    uint32_t j0 = boff[blockIdx.x];
    for (unsigned q = 0; q < w; q++) j0 += wc[q];
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (unsigned r = 0; r < RUN_ROWS; r++) {
        const uint32_t i = base + r * 64u;
        const uint32_t row0 = i - lane;                       // first pixel of this row of 64
        const uint32_t nvalid = row0 < n ? (n - row0 < 64u ? n - row0 : 64u) : 0u;
        if ((m[r] >> lane) & 1ull) {
            const unsigned long long rest = lane == 63u ? 0ull : (m[r] >> (lane + 1u));
            const uint32_t len = rest ? (uint32_t)__builtin_ctzll(rest) + 1u : nvalid - lane;
            const uint32_t j = j0 + (uint32_t)__popcll(m[r] & lt);
            rkeys[j] = s[r];
            rvals[j] = i | ((len - 1u) << RUN_POS_BITS);
        }
        j0 += (uint32_t)__popcll(m[r]);
    }
}

// one run per lane; the wavefront then writes its 64 runs one after the other, each as one
// contiguous store of len words (runs of a segment are consecutive, so the writes stream)
__global__ __launch_bounds__(256) void k_run_expand(const uint32_t *__restrict__ rvals,
                                                    const uint32_t *__restrict__ roff,
                                                    const uint32_t *__restrict__ boffp, uint32_t m,
                                                    uint32_t *__restrict__ pix)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const unsigned lane = lane_id();
    uint32_t v = 0, o = 0;
    if (r < m) {
        v = rvals[r];
        o = roff[r] + (boffp ? boffp[r / SCAN_ITEMS] : 0u);
    }
    // the wavefront's 64 runs hold T pixels (5 a run on the benchmark raster): output pixel t of them is written by
    // lane t % 64 -- it finds its run in the runs' prefix sums (six steps over a wavefront-private LDS row) -- so the
    // wavefront stores 64 pixels per instruction, where a store per run wrote five
    __shared__ uint32_t s_pre[4][64], s_v[4][64], s_o[4][64];
    const unsigned wv = threadIdx.x >> 6;
    const uint32_t len = r < m ? (v >> RUN_POS_BITS) + 1u : 0u;
    uint32_t incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += t;
    }
    s_pre[wv][lane] = incl - len;               // pixels of the wavefront's earlier runs
    s_v[wv][lane] = v & RUN_POS_MASK;
    s_o[wv][lane] = o;
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < T; t += 64u) {
        uint32_t q = 0;                         // the last run whose prefix is <= t
#pragma unroll
        for (uint32_t step = 32u; step >= 1u; step >>= 1)
            if (s_pre[wv][q + step] <= t && q + step < 64u) q += step;
        const uint32_t k = t - s_pre[wv][q];
        pix[s_o[wv][q] + k] = s_v[wv][q] + k;
    }
}

// *pix_out: n pixel indices grouped by segment id (0..S), raster order inside.  Uses ctx->aux /
// aux2 (runs), the sort workspaces and scan_tmp; the result lives in ctx->sort_v1.
// zero[0..nzero) is cleared on the way.
static int build_segment_csr(shp_ctx *ctx, const uint32_t *d_seg, uint32_t n, uint32_t S,
                             uint32_t **pix_out, uint32_t *zero = nullptr, uint32_t nzero = 0)
{
    static const int runs_env = getenv("SHEPSEG_CSR_RUNS") ? atoi(getenv("SHEPSEG_CSR_RUNS")) : 1;
    if (!runs_env || n == 0 || n > (1u << RUN_POS_BITS)) {
        if (nzero) HIPCHK(ctx, hipMemsetAsync(zero, 0, (size_t)nzero * 4, ctx->stream));
        CHK(sort_pairs(ctx, d_seg, nullptr, n, bits_for(S), nullptr, pix_out));
        return 0;
    }
    hipStream_t st = ctx->stream;
    const uint32_t nblk = (n + RUN_TILE - 1) / RUN_TILE;
    CHK(buf_ensure(ctx, ctx->aux, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->sort_v1, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->sort_k0, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->sort_hist, ((size_t)2 * nblk + 16) * 4));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(n)));
    uint32_t *bcount = bp<uint32_t>(ctx->sort_hist), *boff = bcount + nblk, *tot = boff + nblk;
    uint32_t *rkeys = bp<uint32_t>(ctx->aux), *rvals = bp<uint32_t>(ctx->aux2);
    hipLaunchKernelGGL(k_run_tile_count, dim3(nblk), dim3(256), 0, st, d_seg, n, bcount, zero, nzero); KCHK(ctx);
    ArrFn bf{bcount};
    uint32_t *mir = ctx->h_pinned + PIN_MIRROR + MIR_RUNS;        // the scan stores its total there
    CHK(scan_exclusive(ctx, bf, nblk, boff, tot, bp<uint32_t>(ctx->scan_tmp), nullptr, mir));
    hipLaunchKernelGGL(k_run_tile_emit, dim3(nblk), dim3(256), 0, st, d_seg, n, boff, rkeys, rvals); KCHK(ctx);
    HIPCHK(ctx, hipStreamSynchronize(st));
    const uint32_t m = *(volatile uint32_t *)mir;
    if (m == 0 || m > n) SHP_FAIL(ctx, SHP_ERR_STATE, "run count %u out of range (n = %u)", m, n);
    uint32_t *svals = nullptr;
    CHK(sort_pairs(ctx, rkeys, rvals, m, bits_for(S), nullptr, &svals));       // -> ctx->pix
    uint32_t *roff = bp<uint32_t>(ctx->sort_k0);
    RunLenFn lf{svals};
    const uint32_t *boffp = nullptr;
    CHK(scan_exclusive(ctx, lf, m, roff, nullptr, bp<uint32_t>(ctx->scan_tmp), &boffp));
    uint32_t *pix = bp<uint32_t>(ctx->sort_v1);
    hipLaunchKernelGGL(k_run_expand, dim3(grid_for(m, 256)), dim3(256), 0, st, svals, roff, boffp, m, pix);
    KCHK(ctx);
    *pix_out = pix;
    return 0;
}
