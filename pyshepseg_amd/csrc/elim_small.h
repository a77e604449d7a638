// elim_small.h -- iterative small-segment elimination.
//
// Replaces shepseg.eliminateSmallSegments / buildSegmentSpectra / makeSegmentLocations /
// findMergeSegment / doMerge (shepseg.py:780-1123).
//
// Device data layout
//   pix[]      pixel indices grouped by segment id, raster order inside a segment (a CSR built
//              with one stable radix sort; == the reference's segLoc at entry, shepseg.py:880).
//   off[s]     start of segment s in pix[]; origsz[s] its length at entry.
//   ch[] (records {next, pixels, offset, tail})  a merged segment's pixel list is the chain of the ORIGINAL segments it
//              absorbed, in merge order (doMerge appends the source's list to the target's,
//              shepseg.py:1102-1110); iteration order == the reference's list order (N7).
//   ssum[s][b] float32 spectral sums.  Built by an ordered float32 accumulation over the
//              raster-ordered list (N5) -- exact-integer fast path while |partial sums| < 2^24,
//              sequential adds after that -- and merged by float32 '+=' in ascending source id
//              (shepseg.py:1117-1120), which is the reference's merge-loop order.
// Per pass (find phase / merge phase of shepseg.py:983-994): state is frozen while every
// segment of the target size picks its neighbour (one thread per segment); sources are then
// grouped by target (count, scan, fill), ranked by id inside the group, and each target
// applies its sources in ascending id.  A target is never a source in the same pass.
#pragma once
#include "common.h"
#include "scan.h"
#include "sort.h"
#include "csr.h"
#include "gridbar.h"
#include "elim_single.h"
#include <type_traits>

__device__ __forceinline__ float f32_acc(float acc, long long v)
{
    return (float)((double)acc + (double)v);     // numba: float32 + pixel, stored to float32 (N5)
}

// Spectral sums of every segment (buildSegmentSpectra, shepseg.py:780-806).
//   k_spectra_small  a segment of <= 64 pixels is summed by its own thread (pixel list read once,
//                    all bands of the group accumulated together);
//   k_big_seg_list   the ids of the larger ones are compacted into a list (one global atomic per
//                    4096 ids);
//   k_spectra_big    a persistent grid of wavefronts takes list entries round-robin, 64 pixels a
//                    step.  While sum(|v|) stays below 2^24 every float32 partial sum is an exact
//                    integer, so lanes keep private integer partial sums (512 pixels = 8 independent
//                    steps in flight) and one wave reduction per 512 pixels checks the bound
//                    (summed over lanes of the per-lane band maximum: an upper bound, so the test
//                    is conservative).  Past the bound the order of the float32 additions matters:
//                    the values of a step go through LDS transposed, lane b then adds band b's 64
//                    values in list order while the next step's gathers are already in flight.
// WIDE = 32-bit pixel types, whose values need the float64 form of the reference's
// `float32 + pixel` (N5); for 8/16-bit types a float32 add is the same correctly rounded sum.
#define SPECTRA_BG 8        // bands per pass
#define SPECTRA_VAL 4       // ordered phase: image values are loaded this many 64-pixel steps ahead of their chain,
#define SPECTRA_PIX 3       // their pixel indices this many steps ahead of the values (SPECTRA_PIX + 1 == SPECTRA_VAL)
#ifndef SPECTRA_GRID
#define SPECTRA_GRID 4096   // workgroups of the persistent k_spectra_big (latency-bound: fill the wave slots)
#endif
template <typename T> __device__ __forceinline__ T wave_sum_t(T v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// BG = bands per pass, a compile-time constant (= nb when nb <= 8) so that the loads of one pixel
// are a straight run of independent instructions; a last pass with fewer bands than BG re-reads
// band 0 for the surplus slots and drops the result.
template <int DT, int BG>
__global__ __launch_bounds__(256) void k_spectra_small(
    const void *__restrict__ img, int nb, uint32_t n, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ segsz, float *__restrict__ ssum,
    uint32_t S, const ImgGeom g, uint32_t first)
{
    constexpr bool WIDE = DT == SHP_I32 || DT == SHP_U32;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x + first;     // first = 1, or 0 to include the null segment
    if (s > S) return;
    const uint32_t m = segsz[s];
    if (m == 0u || m > 64u) return;
    const uint32_t o = off[s];
    for (int b0 = 0; b0 < nb; b0 += BG) {
        const int bg = nb - b0 < BG ? nb - b0 : BG;
        const size_t base = (size_t)b0 * g.bstride;
        float acc[BG];
#pragma unroll
        for (int j = 0; j < BG; j++) acc[j] = 0.0f;
        // (the pixel index is loaded two pixels ahead of its values, the values one pixel ahead of their
        //  additions: a pixel otherwise costs two dependent memory round trips)
        uint32_t p1 = pix[o], p2 = pix[o + (m > 1u ? 1u : 0u)];
        long long v[BG];
        {
            const size_t idx = geom_off(g, p1);
#pragma unroll
            for (int j = 0; j < BG; j++) v[j] = ld_t<DT>(img, base + (size_t)(j < bg ? j : 0) * g.bstride + idx);
        }
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t p3 = pix[o + (i + 2u < m ? i + 2u : m - 1u)];
            long long vn[BG];
            const size_t idx = geom_off(g, p2);
#pragma unroll
            for (int j = 0; j < BG; j++) vn[j] = ld_t<DT>(img, base + (size_t)(j < bg ? j : 0) * g.bstride + idx);
#pragma unroll
            for (int j = 0; j < BG; j++) acc[j] = WIDE ? f32_acc(acc[j], v[j]) : acc[j] + (float)(int)v[j];
#pragma unroll
            for (int j = 0; j < BG; j++) v[j] = vn[j];
            p2 = p3;
        }
#pragma unroll
        for (int j = 0; j < BG; j++)
            if (j < bg) ssum[(size_t)s * nb + b0 + j] = acc[j];
    }
}

// The same for 8/16-bit pixel types, eight lanes per segment.  With |v| <= 65535 and at most 64 pixels every
// partial sum of the reference's float32 accumulation is an integer below 2^24, hence exact, hence the order
// of the additions is immaterial: the eight lanes of a group read consecutive list entries (one coalesced
// load, and their image values lie in one or two cache lines per band, where a thread per segment had its
// wavefront touch 64 unrelated lines per load and run as long as its longest segment), keep integer partial
// sums and combine them with three shuffles.  The float32 conversion of the exact total is the reference's
// value bit for bit.  (32-bit types keep the ordered kernel above.)
template <int DT, int BG>
__global__ __launch_bounds__(256) void k_spectra_small_grp(
    const void *__restrict__ img, int nb, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ segsz, float *__restrict__ ssum,
    uint32_t S, const ImgGeom g, uint32_t first)
{
    static_assert(DT != SHP_I32 && DT != SHP_U32, "exact only while 64 values stay below 2^24");
    const uint32_t gl = threadIdx.x & 7u;
    const uint32_t s = (blockIdx.x * 256u + threadIdx.x) / 8u + first;
    uint32_t m = 0, o = 0;
    if (s <= S) { m = segsz[s]; o = off[s]; }
    if (m > 64u) m = 0u;                                  // (the larger ones: k_spectra_big)
    for (int b0 = 0; b0 < nb; b0 += BG) {
        const int bg = nb - b0 < BG ? nb - b0 : BG;
        const size_t base = (size_t)b0 * g.bstride;
        int acc[BG];
#pragma unroll
        for (int j = 0; j < BG; j++) acc[j] = 0;
        for (uint32_t i = gl; i < m; i += 8u) {
            const size_t idx = geom_off(g, pix[o + i]);
#pragma unroll
            for (int j = 0; j < BG; j++) acc[j] += (int)ld_t<DT>(img, base + (size_t)(j < bg ? j : 0) * g.bstride + idx);
        }
#pragma unroll
        for (int j = 0; j < BG; j++) {
            acc[j] += __shfl_xor(acc[j], 4, 64);
            acc[j] += __shfl_xor(acc[j], 2, 64);
            acc[j] += __shfl_xor(acc[j], 1, 64);
        }
        if (gl == 0u && m != 0u) {
#pragma unroll
            for (int j = 0; j < BG; j++)
                if (j < bg) ssum[(size_t)s * nb + b0 + j] = (float)acc[j];
        }
    }
}

// list[0] = number of segments with > 64 pixels (zeroed by k_small_init), ids from list[16]
__global__ __launch_bounds__(256) void k_big_seg_list(const uint32_t *__restrict__ segsz, uint32_t S,
                                                      uint32_t *list, uint32_t first)
{
    __shared__ uint32_t s_buf[4096];
    __shared__ uint32_t s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const unsigned lane = lane_id();
    for (uint32_t it = 0; it < 16u; it++) {
        const uint32_t s = blockIdx.x * 4096u + it * 256u + threadIdx.x + first;
        const bool big = s <= S && segsz[s] > 64u;
        const unsigned long long mb = __ballot(big);
        if (mb != 0ull) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(mb));
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            if (big) s_buf[wbase + (uint32_t)__popcll(mb & lanemask_lt())] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(&list[0], s_cnt) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < s_cnt; i += 256u) list[16u + s_base + i] = s_buf[i];
}

template <int DT, int BG>
__global__ __launch_bounds__(256) void k_spectra_big(
    const void *__restrict__ img, int nb, uint32_t n, const uint32_t *__restrict__ pix,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ segsz, float *__restrict__ ssum,
    const uint32_t *__restrict__ list, const ImgGeom g)
{
    constexpr bool WIDE = DT == SHP_I32 || DT == SHP_U32;
    typedef typename std::conditional<WIDE, long long, int>::type IT;
    typedef typename std::conditional<WIDE, double, float>::type FT;
    __shared__ __attribute__((aligned(16))) FT tv[4][BG][68];     // 68: rows 16-byte aligned, lanes (= bands) read theirs conflict-free
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t nbig = list[0];
    const IT LIM = (IT)1 << 24;
    for (uint32_t e = blockIdx.x * 4u + wv; e < nbig; e += gridDim.x * 4u) {
        const uint32_t bs = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[16u + e]);
        const uint32_t bm = (uint32_t)__builtin_amdgcn_readfirstlane((int)segsz[bs]);
        const uint32_t bo = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[bs]);
        for (int b0 = 0; b0 < nb; b0 += BG) {
            const int bg = nb - b0 < BG ? nb - b0 : BG;
            size_t boff[BG];                         // band offsets; surplus slots re-read band 0
#pragma unroll
            for (int j = 0; j < BG; j++) boff[j] = (size_t)(b0 + (j < bg ? j : 0)) * g.bstride;
            // ---- exact phase: private integer partial sums, bound checked per 512 pixels ----
            IT psum[BG], pabs[BG];
#pragma unroll
            for (int j = 0; j < BG; j++) { psum[j] = 0; pabs[j] = 0; }
            uint32_t i0 = 0;
            while (i0 < bm) {
                const uint32_t gend = bm - i0 > 512u ? i0 + 512u : bm;
                IT ts[BG], ta[BG];
#pragma unroll
                for (int j = 0; j < BG; j++) { ts[j] = psum[j]; ta[j] = pabs[j]; }
                size_t idx[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t i = i0 + (uint32_t)u * 64u + lane;
                    ok[u] = i < gend;
                    idx[u] = geom_off(g, pix[bo + (ok[u] ? i : i0)]);   // unconditional loads: they all overlap
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int j = 0; j < BG; j++) {
                        IT v = (IT)ld_t<DT>(img, boff[j] + idx[u]);
                        v = ok[u] ? v : (IT)0;
                        ts[j] += v;
                        ta[j] += v < 0 ? -v : v;
                    }
                }
                IT mx = 0;
#pragma unroll
                for (int j = 0; j < BG; j++) mx = ta[j] > mx ? ta[j] : mx;
                // per-lane maxima stay below 2^24 while the phase lasts, so 64 of them fit IT
                const IT bound = wave_sum_t<IT>(mx);
                if (bound >= LIM) break;
#pragma unroll
                for (int j = 0; j < BG; j++) { psum[j] = ts[j]; pabs[j] = ta[j]; }
                i0 = gend;
            }
            float acc = 0.0f;                        // lane j < bg: running float32 sum of band b0 + j
#pragma unroll
            for (int j = 0; j < BG; j++) {
                const IT t = wave_sum_t<IT>(psum[j]);
                if (lane == (unsigned)j) acc = (float)t;
            }
            // ---- ordered phase: list order float32 additions, lane = band ----
            if (i0 < bm) {
                // Two-stage software pipeline.  A step's image values hang on its pixel indices (two
                // dependent memory round trips): issued together once per step they cost the
                // wavefront a full memory latency per 64 pixels -- and the 10000-pixel pieces of the
                // depth-first cut, ~150 such steps for ONE wavefront, are this kernel's duration.  So
                // the indices are loaded SPECTRA_PIX steps before the values that need them, the
                // values SPECTRA_VAL steps before the chain that adds them.
                IT V[SPECTRA_VAL][BG];
                uint32_t P[SPECTRA_VAL];              // (ring of SPECTRA_PIX + 1 = SPECTRA_VAL slots)
                const uint32_t last = bm - 1u;
#define SP_PIX(cs) pix[bo + ((cs) + lane < bm ? (cs) + lane : last)]
#define SP_VAL(dst, p)                                                                   \
                { const size_t ix_ = geom_off(g, (p));                                    \
                  _Pragma("unroll") for (int j = 0; j < BG; j++) dst[j] = (IT)ld_t<DT>(img, boff[j] + ix_); }
                {
                    uint32_t p0[SPECTRA_VAL];
#pragma unroll
                    for (int d = 0; d < SPECTRA_VAL; d++) p0[d] = SP_PIX(i0 + (uint32_t)d * 64u);
#pragma unroll
                    for (int d = 0; d < SPECTRA_PIX; d++) P[d] = SP_PIX(i0 + (uint32_t)(SPECTRA_VAL + d) * 64u);
#pragma unroll
                    for (int d = 0; d < SPECTRA_VAL; d++) SP_VAL(V[d], p0[d]);
                }
                for (uint32_t c0 = i0; c0 < bm; c0 += 64u * SPECTRA_VAL) {
#pragma unroll
                    for (int d = 0; d < SPECTRA_VAL; d++) {
                        const uint32_t cs = c0 + (uint32_t)d * 64u;
                        if (cs >= bm) break;                         // (uniform)
#pragma unroll
                        for (int j = 0; j < BG; j++) tv[wv][j][lane] = (FT)V[d][j];
                        __builtin_amdgcn_wave_barrier();
                        // indices of step s + VAL + PIX into the slot its predecessor just left, values of
                        // step s + VAL from the indices loaded PIX steps ago
                        const uint32_t pnow = P[d];
                        P[(d + SPECTRA_PIX) % SPECTRA_VAL] = SP_PIX(cs + (uint32_t)(SPECTRA_VAL + SPECTRA_PIX) * 64u);
                        SP_VAL(V[d], pnow);
                        const uint32_t cnt = bm - cs < 64u ? bm - cs : 64u;      // lanes >= cnt staged junk
                        if (lane < (unsigned)bg) {
                            const FT *row = tv[wv][lane];
                            uint32_t q = 0;
                            if (!WIDE && cnt == 64u) {
                                // a whole step: the row comes in as sixteen 16-byte LDS reads issued together
                                // (eight separate waits for eight values each made the 64 dependent adds a
                                // 3000-cycle step: the 10000-pixel pieces of the depth-first cut, ~150 steps
                                // for one wavefront, set this kernel's duration)
                                float4 r4[16];
#pragma unroll
                                for (int u = 0; u < 16; u++) r4[u] = ((const float4 *)row)[u];
#pragma unroll
                                for (int u = 0; u < 16; u++) {
                                    acc = acc + r4[u].x; acc = acc + r4[u].y; acc = acc + r4[u].z; acc = acc + r4[u].w;
                                }
                                q = 64u;
                            }
                            for (; q + 8u <= cnt; q += 8u) {
                                FT r[8];
#pragma unroll
                                for (int u = 0; u < 8; u++) r[u] = row[q + u];
#pragma unroll
                                for (int u = 0; u < 8; u++) {
                                    if (WIDE) acc = (float)((double)acc + (double)r[u]);
                                    else acc = acc + (float)r[u];
                                }
                            }
                            for (; q < cnt; q++) {
                                if (WIDE) acc = (float)((double)acc + (double)row[q]);
                                else acc = acc + (float)row[q];
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
#undef SP_PIX
#undef SP_VAL
            }
            if (lane < (unsigned)bg) ssum[(size_t)bs * nb + b0 + lane] = acc;
        }
    }
}

template <int DT, int BG>
static void launch_spectra_small(hipStream_t st, unsigned gs, const void *d_img, int nb, uint32_t n, const uint32_t *pix,
                                 const uint32_t *off, const uint32_t *segsz, float *ssum, uint32_t S,
                                 const ImgGeom &geom, uint32_t first)
{
    if constexpr (DT == SHP_I32 || DT == SHP_U32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spectra_small<DT, BG>), dim3(gs), dim3(256), 0, st, d_img, nb, n, pix, off,
                           segsz, ssum, S, geom, first);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spectra_small_grp<DT, BG>), dim3(grid_for(((size_t)S + 1) * 8, 256)),
                           dim3(256), 0, st, d_img, nb, pix, off, segsz, ssum, S, geom, first);
}

// float32 sums of segments first .. S (buildSegmentSpectra): the ids above 64 pixels are compacted into
// biglist (its counter word biglist[0] must be zero), then the two kernels above
static int launch_spectra(shp_ctx *ctx, const void *d_img, int dtype, int nb, uint32_t n, const uint32_t *pix,
                          const uint32_t *off, const uint32_t *segsz, float *ssum, uint32_t S,
                          const ImgGeom &geom, uint32_t *biglist, uint32_t first)
{
    hipStream_t st = ctx->stream;
    const unsigned gs = grid_for((size_t)S + 1, 256);
    hipLaunchKernelGGL(k_big_seg_list, dim3(grid_for((size_t)S + 1, 4096)), dim3(256), 0, st, segsz, S, biglist, first);
    KCHK(ctx);
#define SPECTRA_LAUNCH(BGN)                                                                           \
    DISPATCH_DTYPE(dtype,                                                                             \
        launch_spectra_small<DT, BGN>(st, gs, d_img, nb, n, pix, off, segsz, ssum, S, geom, first);    \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spectra_big<DT, BGN>), dim3(SPECTRA_GRID), dim3(256), 0, st, \
                           d_img, nb, n, pix, off, segsz, ssum, biglist, geom))
    switch (nb >= SPECTRA_BG ? SPECTRA_BG : nb) {
    case 1: SPECTRA_LAUNCH(1); break;
    case 2: SPECTRA_LAUNCH(2); break;
    case 3: SPECTRA_LAUNCH(3); break;
    case 4: SPECTRA_LAUNCH(4); break;
    case 5: SPECTRA_LAUNCH(5); break;
    case 6: SPECTRA_LAUNCH(6); break;
    case 7: SPECTRA_LAUNCH(7); break;
    default: SPECTRA_LAUNCH(8); break;
    }
#undef SPECTRA_LAUNCH
    KCHK(ctx);
    return 0;
}

// per-segment state of the pass loop; sizes come from `src` (segsz itself, or origsz when the
// single-pixel stage left them there: then this is also the copy), and thread 0 sets up the loop
// control block -- neither needs a copy launch of its own
__global__ __launch_bounds__(256) void k_small_init(const uint32_t *src, uint32_t *segsz,
                                                    uint32_t *origsz,
                                                    uint4 *__restrict__ ch,
                                                    uint32_t *__restrict__ mergeto,
                                                    uint32_t *__restrict__ tcount,
                                                    uint32_t *__restrict__ tfill, uint32_t *hist,
                                                    uint32_t S, uint32_t min_seg, uint32_t *tlist,
                                                    uint32_t *ctlwords, uint32_t nctlwords,
                                                    uint32_t *off, const uint32_t *__restrict__ off_boff)
{
    __shared__ uint32_t lh[256];            // block-local histogram of sizes 1..255
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0) {                  // SmallCtl: all zero but st[0] = {target 1, prev -1, ...}
        for (uint32_t i = threadIdx.x; i < nctlwords; i += 256u)
            ctlwords[i] = (i == 0u) ? 1u : (i == 1u) ? 0xFFFFFFFFu : 0u;
    }
    if (s <= S) {
        const uint32_t m = src[s];
        segsz[s] = m;
        origsz[s] = m;
        mergeto[s] = 0;
        tcount[s] = 0;
        tfill[s] = 0;
        if (s == 0u) tlist[0] = 0u;             // counter of k_big_seg_list
        uint32_t o = off[s];
        if (off_boff) { o += off_boff[s / SCAN_ITEMS]; off[s] = o; }     // second level of the offset scan
        // the segment's pixel list as ONE chunk: {next chunk, pixels, offset in pix[], last chunk of the chain}
        ch[s] = make_uint4(0u, m, o, s);
        if (s >= 1u && m < min_seg) {
            if (m < 256u) atomicAdd(&lh[m], 1u);
            else atomicAdd(&hist[m], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < min_seg && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// The find/merge pass loop of eliminateSmallSegments (shepseg.py:970-997) as ONE persistent
// kernel per tile.  A pass needs three grid-wide dependencies (find -> link sources to targets
// in ascending id + relabel pixels -> apply); as separate launches that was hundreds of tiny
// dispatches per tile, each costing 0.1-0.3 ms once 16 tiles share the GPU (an earlier version
// built the per-target source lists with count / allocate / fill / rank phases: six barriers;
// a lock-free sorted insert does it in one).  Here SMALL_BLOCKS workgroups stay resident and
// meet at software grid barriers (agent-scope release / acquire as MI355X_MICROARCH.md and
// cdna_hip_programming.md Guideline 16 prescribe: every storing wave drains vmcnt, workgroup
// barrier, one lane's release fence, arrive; poll; one lane's acquire fence, drain, workgroup
// barrier).  Every spin is bounded; on a timeout the kernel sets ctl->fail and every workgroup
// leaves.  The loop control (target, prevCount, numPasses) is recomputed identically by every
// workgroup from the device-side size histogram, so no launch and no host read-back happens
// between passes.  The host caps the number of concurrently running persistent kernels so
// that all their workgroups are co-resident.
// ---------------------------------------------------------------------------------------------
#define SMALL_BLOCKS 64u
#define SMALL_BALANCED_MAX 2048u      // passes with at most this many sources deal them round robin (three barriers)
#define SMALL_SPIN_LIMIT GRID_SPIN_LIMIT

struct SmallCnt { uint32_t nsrc, ntgt, nmerge, work; };   // nmerge: sources that found a target; work: the
                                                          // next candidate of the size's lists nobody took yet
struct SmallState { uint32_t target; int32_t prev; uint32_t passes; uint32_t pad; };
struct SmallCtl {
    SmallState st[2];        // double-buffered loop state (slot parity)
    SmallCnt cnt[3];         // per-pass counters, ring of three: a pass without merges skips its last two
                             // barriers, so a workgroup can be one pass ahead when it zeroes the next slot
    uint32_t bar_count, bar_gen;
    uint32_t nelim, fail, done, slots;
    uint32_t xcd_n[16];             // workgroups of this launch per XCD (counted at kernel start)
    uint32_t xcd_cnt[16];           // per-XCD arrival counters of the two-level grid barrier
    unsigned long long tphase[4];   // SHEPSEG_SMALL_TIMING: wall_clock64 ticks spent in control+find / link / apply
    uint32_t plog[64][4];           // per pass: target, sources, find ticks, merge-phase ticks
    uint32_t phops[64];             // per pass: chunk-chain hops of the find phase (diagnostic)
    uint32_t hopcnt, hoppad;        // the running counter behind phops
    unsigned long long prof[16];    // SMALL_PROF build (make PROF=1): cycles of workgroup 0's wave 0 per phase, passes >= 15
};

struct SmallArgs {
    SmallCtl *ctl;
    uint32_t *pin;      // the host's copy of *ctl (pinned block), written by the kernel when the loop is done
    uint32_t *hist;
    uint32_t *seg, *segsz;
    float *ssum;
    const uint32_t *pix, *off, *origsz;
    uint4 *ch;          // chunk records {next, pixels, offset, tail}: a merged segment's pixel list is the chain of
                        // the original segments' runs in pix[] (one 16-byte load per hop)
    uint32_t *mergeto, *tcount, *toff, *tfill, *tlist, *tsorted, *srclist, *tgtlist;
    uint32_t S, min_seg, nrows, ncols;
    int nb, four;
    double thr2;
    int poll;           // s_sleep(8) repetitions between two polls of a grid barrier
    int bar2;           // two-level (per-XCD) grid barrier
    int lists;          // per-size source lists instead of a scan of the size table per pass
    uint32_t *hopstat;  // SHEPSEG_SMALL_TIMING: counts chunk-chain hops of the find phase (else null)
};

// (the software grid barriers live in gridbar.h)
#ifdef SMALL_PROF
#define SP_DECL unsigned long long sp_t0 = __builtin_readcyclecounter();
#define SP_MARK(ctl_, slot, on) { const unsigned long long sp_t1 = __builtin_readcyclecounter(); if (on) (ctl_)->prof[slot] += sp_t1 - sp_t0; sp_t0 = sp_t1; }
#else
#define SP_DECL
#define SP_MARK(ctl_, slot, on)
#endif

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// A source that merges links itself into its target's list at once (the find phase does not read
// these lists): the list is kept in ascending source id by a lock-free sorted insert (head in
// tfill[t], links in tlist[s]; inserts only, so a failed CAS simply retries), the first source to
// arrive registers the target.
// Returns true if s is the first source of t in this pass: the caller registers the target (register_targets:
// one atomic per wavefront, not per target -- tens of thousands of returning atomics on the one counter were a
// large part of the early passes).
__device__ __forceinline__ bool link_source(const SmallArgs &a, SmallCnt *cnt, uint32_t s, uint32_t t)
{
    bool first = false;
    for (uint32_t tries = 0;; tries++) {
        uint32_t prev = 0, cur = L2LOAD(&a.tfill[t]);
        while (cur != 0 && cur < s) { prev = cur; cur = L2LOAD(&a.tlist[cur]); }
        __hip_atomic_store(&a.tlist[s], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the link must have reached the coherent level before s becomes reachable: both are
        // agent-scope atomics, so draining this lane's stores is enough (a release fence
        // would also write back the XCD's whole L2, once per source)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t *slot = prev ? &a.tlist[prev] : &a.tfill[t];
        if (atomicCAS(slot, cur, s) == cur) {
            first = prev == 0 && cur == 0;                                          // list was empty
            break;
        }
        if (tries > SMALL_SPIN_LIMIT) {
            __hip_atomic_store(&a.ctl->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    return first;
}
// the lanes whose source opened its target's list append the targets to the pass's list (called by the whole wave)
__device__ __forceinline__ void register_targets(const SmallArgs &a, SmallCnt *cnt, bool first, uint32_t t)
{
    const unsigned long long m = __ballot(first);
    if (m == 0ull) return;
    uint32_t base = 0;
    if (lane_id() == (unsigned)__builtin_ctzll(m)) base = atomicAdd(&cnt->ntgt, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), 64);
    if (first) a.tgtlist[base + (uint32_t)__popcll(m & lanemask_lt())] = t;
}

// distSqr between the mean spectra of source s (n pixels) and of U neighbour segments (shepseg.py:1041-1049: band
// by band in ascending order, float32 throughout, means by division).  All loads of a chunk of eight bands --
// the source's row and the U neighbours' rows -- are issued before the first use: written band by band the
// loop waited for memory once per band and per neighbour, a dozen dependent round trips where one does.
// Rows of neighbours that do not qualify are passed as s itself (a valid row; the result is ignored).
template <int U>
__device__ __forceinline__ void seg_dist_multi(const SmallArgs &a, uint32_t s, float nf, const uint32_t (&nbe)[U],
                                               const float (&sf)[U], float (&d)[U])
{
#pragma unroll
    for (int u = 0; u < U; u++) d[u] = 0.0f;
    const float *srow = a.ssum + (size_t)s * a.nb;
    for (int b0 = 0; b0 < a.nb; b0 += 8) {
        float xs[8], es[U][8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = b0 + j < a.nb ? b0 + j : a.nb - 1;
            xs[j] = srow[b];
#pragma unroll
            for (int u = 0; u < U; u++) es[u][j] = a.ssum[(size_t)nbe[u] * a.nb + b];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (b0 + j < a.nb) {
                const float x = xs[j] / nf;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float e2 = es[u][j] / sf[u];
                    const float t = x - e2;
                    const float t2 = t * t;
                    d[u] = d[u] + t2;
                }
            }
        }
    }
}

// findMergeSegment (shepseg.py:1003-1063) for one source by one wavefront.  The source's pixel
// list (chunk chain = the reference's list order) is gathered 64 entries at a time into this
// wave's LDS slice; lanes then own (pixel k, neighbour position) pairs.  The reference keeps
// the FIRST strict minimum in (k, ii outer, jj inner) order (N7) == the lexicographic minimum
// of (distSqr, k, position): one 64-bit wave reduction (distSqr >= +0, so its float32 bit
// pattern orders like the value).
__device__ __forceinline__ bool find_merge_wave(uint32_t s, uint32_t target, const SmallArgs &a,
                                                uint32_t *wpix, SmallCnt *cnt, bool prof = false)
{
    SP_DECL
    const unsigned lane = lane_id();
    const float nf = (float)target;
    const uint32_t nq = a.four ? 4u : 8u;
    unsigned long long best = ~0ull;         // (float bits of distSqr << 32) | order
    uint32_t bestnb = 0;
    uint32_t c = s, ci = 0;                  // chain cursor: chunk id, index inside the chunk
    uint4 rec = a.ch[c];
    uint32_t co = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.z);
    uint32_t cm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.y);
    uint32_t cn = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.x);
    SP_MARK(a.ctl, 8, prof)
    for (uint32_t kbase = 0; kbase < target; kbase += 64u) {
        const uint32_t want = (target - kbase < 64u) ? (target - kbase) : 64u;
        // gather list entries kbase .. kbase+want-1
        uint32_t got = 0;
        while (got < want) {
            if (ci >= cm) {
                c = cn;
                if (c == 0) break;
                if (a.hopstat && lane == 0) atomicAdd(a.hopstat, 1u);
                ci = 0;
                rec = a.ch[c];
                co = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.z);
                cm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.y);
                cn = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.x);
                continue;
            }
            uint32_t take = cm - ci;
            if (take > want - got) take = want - got;
            if (lane < take) wpix[got + lane] = a.pix[co + ci + lane];
            got += take;
            ci += take;
        }
        __builtin_amdgcn_wave_barrier();
        SP_MARK(a.ctl, 9, prof)
        // the (pixel, neighbour) pairs of these <= 64 list entries, four 64-pair steps at a time and
        // stage by stage (neighbour id, its size, its sums), so that a round trip to memory is paid
        // per stage and not per step: the pass loop is a chain of such round trips
        const uint32_t npairs = got * nq;
        for (uint32_t q0 = 0; q0 < npairs; q0 += 256u) {
            uint32_t nbid[4], szn[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t q = q0 + u * 64u + lane;
                nbid[u] = 0u;
                if (q < npairs) {
                    const uint32_t kk = q / nq, pos = q - kk * nq;
                    const uint32_t p = wpix[kk];
                    const uint32_t r = p / a.ncols, cc = p - r * a.ncols;
                    int di, dj;                  // neighbour `pos` in (ii outer, jj inner) order
                    if (a.four) {
                        di = (pos == 0u) ? -1 : (pos == 3u) ? 1 : 0;
                        dj = (pos == 1u) ? -1 : (pos == 2u) ? 1 : 0;
                    } else {
                        const uint32_t e = pos < 4u ? pos : pos + 1u;      // skip the centre
                        di = (int)(e / 3u) - 1;
                        dj = (int)(e % 3u) - 1;
                    }
                    const int ii = (int)r + di, jj = (int)cc + dj;
                    if (ii >= 0 && jj >= 0 && ii < (int)a.nrows && jj < (int)a.ncols)
                        nbid[u] = a.seg[(uint32_t)ii * a.ncols + (uint32_t)jj];
                }
            }
            SP_MARK(a.ctl, 10, prof && nbid[0] != 0xFFFFFFFFu)
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                if (nbid[u] == s) nbid[u] = 0u;
                szn[u] = nbid[u] ? a.segsz[nbid[u]] : 0u;
            }
            SP_MARK(a.ctl, 11, prof && szn[0] != 0xFFFFFFFFu)
            {
                uint32_t nbe[4];
                float sfv[4], dv[4];
                bool any = false;
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const bool elig = szn[u] > target;
                    any |= elig;
                    nbe[u] = elig ? nbid[u] : s;
                    sfv[u] = elig ? (float)szn[u] : nf;
                }
                if (__any(any)) {
                    seg_dist_multi<4>(a, s, nf, nbe, sfv, dv);
#pragma unroll
                    for (uint32_t u = 0; u < 4u; u++) {
                        if (szn[u] > target) {
                            const uint32_t q = q0 + u * 64u + lane;
                            const uint32_t kk = q / nq, pos = q - kk * nq;
                            const unsigned long long key =
                                ((unsigned long long)__float_as_uint(dv[u]) << 32) |
                                (unsigned long long)((kbase + kk) * 8u + pos);
                            if (key < best) { best = key; bestnb = nbid[u]; }
                        }
                    }
                }
            }
            SP_MARK(a.ctl, 12, prof && best != 1ull)
        }
        __builtin_amdgcn_wave_barrier();
        if (c == 0) break;
    }
    const unsigned long long wmin = wave_min_u64(best);
    SP_MARK(a.ctl, 13, prof)
    if (wmin == ~0ull) { if (lane == 0) a.mergeto[s] = 0; return false; }
    const float bd = __uint_as_float((uint32_t)(wmin >> 32));
    const bool merges = !((double)bd > a.thr2);
    bool first = false;
    if (best == wmin) {                      // unique: (k, position) differs between lanes
        a.mergeto[s] = merges ? bestnb : 0u;
        if (merges) first = link_source(a, cnt, s, bestnb);
    }
    register_targets(a, cnt, first, bestnb);
    SP_MARK(a.ctl, 14, prof)
    return merges;                           // wave-uniform: does s merge in this pass?
}

// The same for sources of at most 256 (pixel, neighbour) pairs, as many of them at once as fit 256 slots: a
// wavefront owns four slots per lane, slot q belongs to pair q % npairs of source q / npairs of the batch (the
// sources of a pass all have `target` pixels), and every stage -- chunk heads, chain hops, pixel, neighbour id,
// its size, the spectra -- issues the loads of all four slots before it waits: one round trip to memory per
// stage and BATCH, where a wavefront per source (or a group of lanes per source, one batch of groups after the
// other) paid them per source; two-pixel sources go 32 to a batch.  Each source's minimum of (distSqr, k,
// position) is a 64-bit ds_min in the wavefront's LDS slice; the slot that holds it records the target and
// links the source.  ids: this wavefront's list of `count` sources in LDS.  Returns how many of them merge.
#define SMALL_BATCH_IDS 256u
#define SMALL_BATCH_SRC 32u
template <int U>
__device__ __forceinline__ void seg_dist_pairs(const SmallArgs &a, const uint32_t (&src)[U], float nf,
                                               const uint32_t (&nbe)[U], const float (&sf)[U], float (&d)[U])
{
#pragma unroll
    for (int u = 0; u < U; u++) d[u] = 0.0f;
    for (int b0 = 0; b0 < a.nb; b0 += 4) {           // four bands at a time: 8 U values in registers
        float xs[U][4], es[U][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = b0 + j < a.nb ? b0 + j : a.nb - 1;
#pragma unroll
            for (int u = 0; u < U; u++) {
                xs[u][j] = a.ssum[(size_t)src[u] * a.nb + b];
                es[u][j] = a.ssum[(size_t)nbe[u] * a.nb + b];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (b0 + j < a.nb) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float x = xs[u][j] / nf;
                    const float e2 = es[u][j] / sf[u];
                    const float t = x - e2;
                    const float t2 = t * t;
                    d[u] = d[u] + t2;
                }
            }
        }
    }
}

template <int U>
__device__ __forceinline__ uint32_t find_merge_batch(const uint32_t *ids, uint32_t count, uint32_t target,
                                                     const SmallArgs &a, SmallCnt *cnt, unsigned long long *wkey,
                                                     bool prof = false)
{
    SP_DECL
    const unsigned lane = lane_id();
    const uint32_t nq = a.four ? 4u : 8u;
    const float nf = (float)target;
    const uint32_t npairs = target * nq;                       // <= 64 U
    uint32_t nsb = (64u * (uint32_t)U) / npairs;
    nsb = nsb > SMALL_BATCH_SRC ? SMALL_BATCH_SRC : nsb;
    uint32_t merges = 0;
    for (uint32_t i0 = 0; i0 < count; i0 += nsb) {
        const uint32_t nsrc = count - i0 < nsb ? count - i0 : nsb;
        const uint32_t nslots = nsrc * npairs;
        if (lane < SMALL_BATCH_SRC) wkey[lane] = ~0ull;
        uint32_t src[U], si[U], kidx[U], kk[U], pos[U], c[U], cm[U], co[U];
        bool valid[U];
        // ---- the slots' sources and the heads of their pixel lists ----
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t q = (uint32_t)u * 64u + lane;
            valid[u] = q < nslots;
            si[u] = q / npairs;
            const uint32_t pr = q - si[u] * npairs;
            kidx[u] = pr / nq;
            pos[u] = pr - kidx[u] * nq;
            kk[u] = kidx[u];
            src[u] = valid[u] ? ids[i0 + si[u]] : ids[i0];
            c[u] = src[u];
        }
        SP_MARK(a.ctl, 8, prof)
        uint32_t cx[U];                              // the chunks' successors
#pragma unroll
        for (int u = 0; u < U; u++) { const uint4 r = a.ch[c[u]]; cx[u] = r.x; cm[u] = r.y; co[u] = r.z; }
        // ---- chain hops (the chunk chain = the reference's list order), all slots together ----
        bool need[U], hopped = false;
#pragma unroll
        for (int u = 0; u < U; u++) need[u] = valid[u] && kk[u] >= cm[u];
        bool anyneed = false;
#pragma unroll
        for (int u = 0; u < U; u++) anyneed |= need[u];
        while (__any(anyneed)) {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (need[u]) {
                    kk[u] -= cm[u];
                    c[u] = cx[u];
                    if (cx[u] == 0u) { valid[u] = false; need[u] = false; c[u] = src[u]; }
                }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (need[u]) { const uint4 r = a.ch[c[u]]; cx[u] = r.x; cm[u] = r.y; co[u] = r.z; }
#pragma unroll
            for (int u = 0; u < U; u++) need[u] = need[u] && kk[u] >= cm[u];
            anyneed = false;
#pragma unroll
            for (int u = 0; u < U; u++) anyneed |= need[u];
            hopped = true;
        }
        (void)hopped;
        // ---- pixel, neighbour id, neighbour size ----
        SP_MARK(a.ctl, 9, prof && cm[0] != 0xFFFFFFFFu)
        uint32_t p[U], nbid[U], szn[U];
#pragma unroll
        for (int u = 0; u < U; u++) p[u] = a.pix[co[u] + (valid[u] ? kk[u] : 0u)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            nbid[u] = 0u;
            const uint32_t r = p[u] / a.ncols, cc = p[u] - r * a.ncols;
            int di, dj;                          // neighbour `pos` in (ii outer, jj inner) order
            if (a.four) {
                di = (pos[u] == 0u) ? -1 : (pos[u] == 3u) ? 1 : 0;
                dj = (pos[u] == 1u) ? -1 : (pos[u] == 2u) ? 1 : 0;
            } else {
                const uint32_t e = pos[u] < 4u ? pos[u] : pos[u] + 1u;      // skip the centre
                di = (int)(e / 3u) - 1;
                dj = (int)(e % 3u) - 1;
            }
            const int ii = (int)r + di, jj = (int)cc + dj;
            if (valid[u] && ii >= 0 && jj >= 0 && ii < (int)a.nrows && jj < (int)a.ncols)
                nbid[u] = a.seg[(uint32_t)ii * a.ncols + (uint32_t)jj];
        }
        SP_MARK(a.ctl, 10, prof && nbid[0] != 0xFFFFFFFFu)
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (nbid[u] == src[u]) nbid[u] = 0u;
            szn[u] = nbid[u] ? a.segsz[nbid[u]] : 0u;
        }
        SP_MARK(a.ctl, 11, prof && szn[0] != 0xFFFFFFFFu)
        // ---- distances, each source's minimum ----
        unsigned long long key[U];
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += 4) {
            uint32_t srcg[4], nbe[4];
            float sfv[4], dv[4];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool elig = szn[u0 + u] > target;
                any |= elig;
                srcg[u] = src[u0 + u];
                nbe[u] = elig ? nbid[u0 + u] : src[u0 + u];
                sfv[u] = elig ? (float)szn[u0 + u] : nf;
                key[u0 + u] = ~0ull;
            }
            if (__any(any)) {
                seg_dist_pairs<4>(a, srcg, nf, nbe, sfv, dv);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (szn[u0 + u] > target) {
                        key[u0 + u] = ((unsigned long long)__float_as_uint(dv[u]) << 32) |
                                      (unsigned long long)(kidx[u0 + u] * 8u + pos[u0 + u]);
                        atomicMin(&wkey[si[u0 + u]], key[u0 + u]);
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        SP_MARK(a.ctl, 12, prof)
        // ---- the slot that holds its source's minimum (unique: (k, position) differs between slots) hands the
        //      neighbour over; then a LANE PER SOURCE records the decision and links: all sources of the batch
        //      in one round of the sorted insert's round trips (slot by slot it was up to U rounds) ----
        uint32_t *wtgt = (uint32_t *)(wkey + SMALL_BATCH_SRC);
#pragma unroll
        for (int u = 0; u < U; u++)
            if ((uint32_t)u * 64u + lane < nslots && key[u] != ~0ull && key[u] == wkey[si[u]]) wtgt[si[u]] = nbid[u];
        __builtin_amdgcn_wave_barrier();
        {
            bool won = false, first = false;
            uint32_t t = 0u;
            if (lane < nsrc) {
                const uint32_t s = ids[i0 + lane];
                const unsigned long long wm = wkey[lane];
                if (wm != ~0ull) {
                    const float bd = __uint_as_float((uint32_t)(wm >> 32));
                    won = !((double)bd > a.thr2);
                    t = wtgt[lane];
                }
                a.mergeto[s] = won ? t : 0u;
                if (won) first = link_source(a, cnt, s, t);
            }
            register_targets(a, cnt, first, t);
            merges += (uint32_t)__popcll(__ballot(won));
        }
        SP_MARK(a.ctl, 14, prof)
        __builtin_amdgcn_wave_barrier();
    }
    return merges;
}

#ifndef SMALL_MINWAVES
#define SMALL_MINWAVES 4
#endif
__global__ __launch_bounds__(256, SMALL_MINWAVES) void k_small_loop(SmallArgs a)
{
    __shared__ uint32_t wpix[4][64];
    __shared__ uint32_t wids[4][SMALL_BATCH_IDS];
    __shared__ unsigned long long wkeys[4][SMALL_BATCH_SRC + SMALL_BATCH_SRC / 2u];     // keys, then the winners' targets
    __shared__ uint32_t s_target, s_done, s_count;
    __shared__ uint32_t lhist[256];
    SmallCtl *ctl = a.ctl;
    __builtin_amdgcn_s_setprio(2);
    const uint32_t G = gridDim.x;
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x, gthreads = G * 256u;
    const uint32_t gwave = gtid >> 6, gwaves = gthreads >> 6;
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    // where does this workgroup run?  count the launch's workgroups per XCD, meet once with the
    // one-level barrier so that everybody sees the final counts, then use the two-level barrier
    __shared__ SmallBar s_bar;
    if (threadIdx.x == 0) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_s_getreg(SMALL_GETREG_XCC_ID) & 15u;
        s_bar.xcd = x;
        __hip_atomic_fetch_add(&ctl->xcd_n[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!small_grid_barrier(ctl, G, a.poll)) return;
    if (threadIdx.x == 0) {
        uint32_t nact = 0;
        for (int i = 0; i < 16; i++)
            nact += __hip_atomic_load(&ctl->xcd_n[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        s_bar.nx = __hip_atomic_load(&ctl->xcd_n[s_bar.xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_bar.nactive = nact;
        s_bar.poll = a.poll;
    }
    __syncthreads();
    const SmallBar bar = s_bar;
    // ---- per-size source lists.  A pass needs the segments of exactly `target` pixels.  Sizes only grow, so
    //      they are: the segments that HAD that size at entry (bysize: the ids below min_seg counting-sorted by
    //      size, built here) plus the targets that GREW to it in an earlier pass (grown[size]: appended by the
    //      merge phase, at most `cap` per size; a size that overflows is scanned for as before).  Either kind
    //      may have grown on since: a candidate counts if its size is still the target's. ----
    uint32_t *bstart = a.tcount, *gcount = a.tcount + (a.min_seg + 1u), *gover = gcount + a.min_seg,
             *cursor = gover + a.min_seg;
    uint32_t *bysize = a.tsorted, *grown = a.toff;
    const uint32_t cap = (a.S + 1u) / (a.min_seg ? a.min_seg : 1u);
    const bool lists = a.lists && a.min_seg >= 2u && a.min_seg <= 256u && 4u * a.min_seg + 1u <= a.S + 1u && cap >= 64u;
    if (lists) {
        if (blockIdx.x == 0) {
            for (uint32_t u = threadIdx.x; u < a.min_seg; u += 256u) { gcount[u] = 0u; gover[u] = 0u; cursor[u] = 0u; }
            if (threadIdx.x == 0) {
                uint32_t acc = 0;
                bstart[0] = 0u;
                for (uint32_t u = 1; u <= a.min_seg; u++) { bstart[u] = acc; if (u < a.min_seg) acc += a.hist[u]; }
            }
        }
        if (!(a.bar2 ? small_grid_barrier2(ctl, bar) : small_grid_barrier(ctl, G, a.poll))) return;
        // counting sort, a contiguous range of ids per workgroup: sizes counted in LDS, one global atomic per
        // (workgroup, size) to reserve the run (a global atomic per id put 77 000 of them on ONE address for the
        // two-pixel segments of a tile -- a millisecond, and far more with twelve loops at it)
        __shared__ uint32_t lbase[256];
        const uint32_t per = (a.S + G - 1u) / G, lo = blockIdx.x * per + 1u;
        const uint32_t hi = lo + per - 1u < a.S ? lo + per - 1u : a.S;
        lhist[threadIdx.x] = 0u;
        __syncthreads();
        for (uint32_t sid = lo + threadIdx.x; sid <= hi; sid += 256u) {
            const uint32_t m = a.segsz[sid];
            if (m >= 1u && m < a.min_seg) atomicAdd(&lhist[m], 1u);
        }
        __syncthreads();
        if (threadIdx.x < a.min_seg) {
            const uint32_t c = lhist[threadIdx.x];
            lbase[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0u;
            lhist[threadIdx.x] = 0u;
        }
        __syncthreads();
        for (uint32_t sid = lo + threadIdx.x; sid <= hi; sid += 256u) {
            const uint32_t m = a.segsz[sid];
            if (m >= 1u && m < a.min_seg) bysize[bstart[m] + lbase[m] + atomicAdd(&lhist[m], 1u)] = sid;
        }
        __syncthreads();
        if (!(a.bar2 ? small_grid_barrier2(ctl, bar) : small_grid_barrier(ctl, G, a.poll))) return;
    }
    unsigned long long tmark = wall_clock64();
    SP_DECL
    for (uint32_t slot = 0;; slot++) {
        const uint32_t par = slot & 1u;
#ifdef SMALL_PROF
#ifdef SMALL_PROF_EARLY
        const bool sprof = gtid == 0u && (uint32_t)ctl->tphase[3] < 8u;
#else
        const bool sprof = gtid == 0u && (uint32_t)ctl->tphase[3] >= 15u;
#endif
#endif
        SP_MARK(ctl, 15, false)
        // ---- loop control (identical in every workgroup; shepseg.py:970-997) ----
        if (threadIdx.x == 0) {
            uint32_t target = ctl->st[par].target, passes = ctl->st[par].passes;
            int32_t prev = ctl->st[par].prev;
            uint32_t done = 0;
            for (;;) {
                if (target >= a.min_seg) { done = 1; break; }
                const int32_t count = (int32_t)a.hist[target];
                if (count != prev && passes < 10u) {          // `while` condition, shepseg.py:980
                    prev = count;
                    passes++;
                    if (count > 0) break;                      // a pass with sources
                } else {
                    target++; prev = -1; passes = 0;          // next targetSize, shepseg.py:970
                }
            }
            s_target = target; s_done = done; s_count = done ? 0u : (uint32_t)prev;      // sources of this pass
            if (blockIdx.x == 0) {
                ctl->st[par ^ 1u].target = target; ctl->st[par ^ 1u].prev = prev;
                ctl->st[par ^ 1u].passes = passes;
                SmallCnt *nx = &ctl->cnt[(slot + 1u) % 3u];
                nx->nsrc = 0; nx->ntgt = 0; nx->nmerge = 0; nx->work = 0;
                if (done) { ctl->done = 1; ctl->slots = slot; }
            }
        }
        __syncthreads();
        if (s_done) {
            // workgroup 0 hands the control block to the host (every other workgroup's counters
            // reached the L2 before the last barrier)
            if (blockIdx.x == 0)
                for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(SmallCtl) / 4); i += 256u)
                    MIRROR_STORE(&a.pin[i], L2LOAD(&((const uint32_t *)ctl)[i]));
            break;
        }
        const uint32_t target = s_target;
        SmallCnt *cnt = &ctl->cnt[slot % 3u];
        SP_MARK(ctl, 0, sprof)
        // ---- find phase: sources = segments of the target size.  Every wavefront scans its own
        //      64-id slices of the size table (four slices in flight: the scan of ~2.5 M ids is
        //      repeated every pass and is pure load latency).  With many sources a wavefront handles
        //      what it finds at once.  With few (the ~40 late passes of a tile: a few hundred sources of
        //      30-50 pixels) the pass lasts as long as the wavefront that happened to find the most --
        //      four or five dependent find chains where the average is one -- so the sources are only
        //      LISTED by the scan, the grid meets once more, and the list is dealt round robin. ----
        const bool balanced = s_count <= SMALL_BALANCED_MAX;
        const bool from_lists = lists && gover[target] == 0u;
        if (from_lists) {
            // the candidates of this size, dealt in even chunks of at most 256; no scan, no listing barrier
            const uint32_t b0 = bstart[target], nb0 = bstart[target + 1u] - b0;
            const uint32_t nc = nb0 + gcount[target];            // (no overflow: gcount <= cap)
            const uint32_t npairs = target * (a.four ? 4u : 8u);
            // (a third to a half of the candidates are stale, unevenly: a static deal left some wavefront with
            //  three or four sources of 45 pixels where the average is one; so the wavefronts TAKE chunks from
            //  a counter, about four per wavefront)
            uint32_t chunk = (nc + 4u * gwaves - 1u) / (4u * gwaves);
            chunk = chunk < 1u ? 1u : chunk > SMALL_BATCH_IDS ? SMALL_BATCH_IDS : chunk;
            uint32_t wmerges = 0;
            for (;;) {
                uint32_t c0 = 0;
                if (lane == 0) c0 = atomicAdd(&cnt->work, chunk);
                c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
                if (c0 >= nc) break;
                const uint32_t nn = nc - c0 < chunk ? nc - c0 : chunk;
                uint32_t id[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t o = u * 64u + lane, ci = c0 + o;
                    id[u] = 0u;
                    if (o < nn) id[u] = ci < nb0 ? bysize[b0 + ci] : grown[(size_t)target * cap + (ci - nb0)];
                }
                uint32_t szv[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) szv[u] = id[u] ? a.segsz[id[u]] : 0xFFFFFFFFu;
                uint32_t nfound = 0;
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const bool ok = id[u] != 0u && szv[u] == target;
                    const unsigned long long m = __ballot(ok);
                    if (ok) wids[w][nfound + (uint32_t)__popcll(m & lanemask_lt())] = id[u];
                    nfound += (uint32_t)__popcll(m);
                }
                SP_MARK(ctl, 1, sprof)
                if (nfound == 0u) continue;
                __builtin_amdgcn_wave_barrier();
                uint32_t gbase = 0;
                if (lane == 0) gbase = atomicAdd(&cnt->nsrc, nfound);
                gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
                for (uint32_t q = lane; q < nfound; q += 64u) a.srclist[gbase + q] = wids[w][q];
                SP_MARK(ctl, 2, sprof)
#ifdef SMALL_PROF
                if (npairs <= 256u) wmerges += find_merge_batch<4>(wids[w], nfound, target, a, cnt, wkeys[w], sprof);
#else
                if (npairs <= 256u) wmerges += find_merge_batch<4>(wids[w], nfound, target, a, cnt, wkeys[w]);
#endif
                else for (uint32_t q = 0; q < nfound; q++) {
#ifdef SMALL_PROF
                    wmerges += find_merge_wave((uint32_t)__builtin_amdgcn_readfirstlane((int)wids[w][q]), target, a, wpix[w],
                                               cnt, sprof) ? 1u : 0u;
                    if (sprof) ctl->prof[7] += 1;
#else
                    wmerges += find_merge_wave((uint32_t)__builtin_amdgcn_readfirstlane((int)wids[w][q]), target, a, wpix[w],
                                               cnt) ? 1u : 0u;
#endif
                }
                __builtin_amdgcn_wave_barrier();
                SP_MARK(ctl, 3, sprof)
            }
            SP_MARK(ctl, 1, sprof)
            if (wmerges && lane == 0) atomicAdd(&cnt->nmerge, wmerges);
        } else {
            const uint32_t stride = gwaves * 64u;
            uint32_t wmerges = 0;
            for (uint32_t base = gwave * 64u; base < a.S; base += 4u * stride) {
                uint32_t sz[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t sid = base + (uint32_t)u * stride + lane + 1u;
                    sz[u] = sid <= a.S ? a.segsz[sid] : 0xFFFFFFFFu;
                }
                const uint32_t npairs = target * (a.four ? 4u : 8u);
                uint32_t nfound = 0;                 // sources of these four slices, gathered in wids[w]
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    unsigned long long m = __ballot(sz[u] == target);
                    if (m == 0ull) continue;
                    const uint32_t b0 = base + (uint32_t)u * stride;
                    uint32_t gbase = 0;
                    if (lane == 0) gbase = atomicAdd(&cnt->nsrc, (uint32_t)__popcll(m));
                    gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
                    const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
                    if (sz[u] == target) a.srclist[gbase + rank] = b0 + lane + 1u;
                    if (balanced) continue;
                    if (npairs <= 256u) {
                        if (sz[u] == target) wids[w][nfound + rank] = b0 + lane + 1u;
                        nfound += (uint32_t)__popcll(m);
                    } else while (m) {
                        const uint32_t src = b0 + (uint32_t)__builtin_ctzll(m) + 1u;
                        m &= m - 1ull;
                        wmerges += find_merge_wave(src, target, a, wpix[w], cnt) ? 1u : 0u;
                    }
                }
                if (nfound) {
                    __builtin_amdgcn_wave_barrier();
                    wmerges += find_merge_batch<4>(wids[w], nfound, target, a, cnt, wkeys[w]);
                }
            }
            if (balanced) {
                SP_MARK(ctl, 1, sprof)
                if (!(a.bar2 ? small_grid_barrier2(ctl, bar) : small_grid_barrier(ctl, G, a.poll))) return;
                SP_MARK(ctl, 2, sprof)
                const uint32_t nlisted = cnt->nsrc;
                const uint32_t npairs = target * (a.four ? 4u : 8u);
                if (npairs <= 128u) {
                    // several sources fit a batch: deal chunks of them (no larger than an even share).  (Eight
                    // slots per lane, i.e. two sources of 33..64 pixels at once, need 250 registers -- two such
                    // loops per CU -- or spill at 128: the late passes came out 10 % slower than with a
                    // wavefront per source.)
                    uint32_t chunk = 256u / npairs;
                    chunk = chunk > SMALL_BATCH_SRC ? SMALL_BATCH_SRC : chunk;
                    const uint32_t share = (nlisted + gwaves - 1u) / gwaves;
                    chunk = chunk > share ? (share ? share : 1u) : chunk;
                    for (uint32_t i = gwave * chunk; i < nlisted; i += gwaves * chunk) {
                        const uint32_t nn = nlisted - i < chunk ? nlisted - i : chunk;
                        if (lane < nn) wids[w][lane] = a.srclist[i + lane];
                        __builtin_amdgcn_wave_barrier();
                        wmerges += find_merge_batch<4>(wids[w], nn, target, a, cnt, wkeys[w]);
                    }
                } else
                for (uint32_t i = gwave; i < nlisted; i += gwaves) {
                    const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.srclist[i]);
#ifdef SMALL_PROF
                    wmerges += find_merge_wave(src, target, a, wpix[w], cnt, sprof) ? 1u : 0u;
                    if (sprof) ctl->prof[7] += 1;
#else
                    wmerges += find_merge_wave(src, target, a, wpix[w], cnt) ? 1u : 0u;
#endif
                }
                SP_MARK(ctl, 3, sprof)
            }
            if (wmerges && lane == 0) atomicAdd(&cnt->nmerge, wmerges);
        }
        if (!(a.bar2 ? small_grid_barrier2(ctl, bar) : small_grid_barrier(ctl, G, a.poll))) return;
        SP_MARK(ctl, 4, sprof)
        const uint32_t nsrc = cnt->nsrc;
        if (gtid == 0) {
            const unsigned long long t = wall_clock64();
            const uint32_t pi = (uint32_t)ctl->tphase[3] & 63u;
            ctl->plog[pi][0] = target; ctl->plog[pi][1] = nsrc; ctl->plog[pi][2] = (uint32_t)(t - tmark); ctl->plog[pi][3] = 0;
            if (a.hopstat) ctl->phops[pi] = atomicExch(a.hopstat, 0u);
            ctl->tphase[0] += t - tmark; tmark = t;
        }
        if (cnt->nmerge == 0u) {
            // nobody merges (every candidate was further than maxSpectralDiff, or had no larger
            // neighbour): nothing to link, relabel or absorb, the size histogram is unchanged
            if (gtid == 0) ctl->tphase[3] += 1;
            continue;
        }
        // ---- merge phase.  The sources linked themselves to their targets during the find phase, so
        //      one more barrier is all a pass needs: the sources' pixels are relabelled (doMerge
        //      :1107-1109; a source's chunk chain ends at chtail[s] -- what the targets append behind
        //      it meanwhile is not followed) while each target absorbs its sources in ascending id
        //      (doMerge :1112-1123).  The two touch disjoint state. ----
        if (target <= 16u) {                   // a thread per source: at most 16 pixels each
            for (uint32_t i = gtid; i < nsrc; i += gthreads) {
                const uint32_t s = a.srclist[i];
                const uint32_t t = a.mergeto[s];
                if (t == 0) continue;
                uint4 r = a.ch[s];
                const uint32_t last = r.w;
                for (uint32_t c = s;;) {
                    for (uint32_t j = 0; j < r.y; j++) a.seg[a.pix[r.z + j]] = t;
                    if (c == last || r.x == 0u) break;
                    c = r.x;
                    r = a.ch[c];
                }
            }
        } else {
            for (uint32_t i = gwave; i < nsrc; i += gwaves) {
                const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.srclist[i]);
                const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.mergeto[s]);
                if (t == 0) continue;
                // (a chunk's record is one load; the successor's is fetched while this chunk's pixels are on
                //  their way)
                uint32_t c = s;
                uint4 r = a.ch[c];
                const uint32_t last = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w);
                for (;;) {
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.z);
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.y);
                    const uint32_t nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.x);
                    const bool more = c != last && nx != 0u;
                    uint4 r2 = r;
                    if (more) r2 = a.ch[nx];
                    for (uint32_t j = lane; j < m; j += 64u) a.seg[a.pix[o + j]] = t;
                    if (!more) break;
                    c = nx; r = r2;
                }
            }
        }
        SP_MARK(ctl, 5, sprof)
        const uint32_t ntgt = cnt->ntgt;
        // ---- merge step 2: each target absorbs its sources in ascending id (doMerge :1112-1123)
        //      size-histogram updates go through LDS, numElim through a wave reduction ----
        for (uint32_t b = threadIdx.x; b < 256u; b += 256u) lhist[b] = 0;
        __syncthreads();
        uint32_t my_elim = 0;
        for (uint32_t i0 = gtid; i0 < ntgt; i0 += gthreads) {
            const uint32_t t = a.tgtlist[i0];
            const uint32_t a0 = a.segsz[t];
            uint32_t *chw = (uint32_t *)a.ch;        // the records' words: [4 c] next, [4 c + 3] tail
            uint32_t sz = a0, tail = chw[4u * (size_t)t + 3u], n = 0;
            // the target's sums stay in registers (eight bands at a time) while its sources are added in
            // ascending id; everything a source contributes -- its link, size, chain tail and sums -- is loaded
            // together (band by band, with stores in between, every band was a round trip of its own)
            for (int b0 = 0; b0 < a.nb; b0 += 8) {
                float acc[8];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = b0 + j < a.nb ? a.ssum[(size_t)t * a.nb + b0 + j] : 0.0f;
                for (uint32_t s = a.tfill[t]; s != 0;) {
                    const uint32_t nxt = a.tlist[s];
                    float add[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) add[j] = b0 + j < a.nb ? a.ssum[(size_t)s * a.nb + b0 + j] : 0.0f;
                    if (b0 == 0) {
                        const uint32_t ssz = a.segsz[s], stail = chw[4u * (size_t)s + 3u];
                        sz += ssz;
                        a.segsz[s] = 0;
                        chw[4u * (size_t)tail] = s;
                        tail = stail;
                        n++;                         // (mergeto[s] stays: the relabel beside us reads it,
                                                     //  and every pass rewrites it for its own sources)
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (b0 + j < a.nb) { acc[j] = acc[j] + add[j]; a.ssum[(size_t)s * a.nb + b0 + j] = 0.0f; }
                    s = nxt;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) if (b0 + j < a.nb) a.ssum[(size_t)t * a.nb + b0 + j] = acc[j];
            }
            a.segsz[t] = sz;
            chw[4u * (size_t)t + 3u] = tail;
            a.tfill[t] = 0;
            my_elim += n;
            if (lists && sz < a.min_seg) {        // a source of a later pass: onto its size's list
                const uint32_t gi = atomicAdd(&gcount[sz], 1u);
                if (gi < cap) grown[(size_t)sz * cap + gi] = t; else gover[sz] = 1u;
            }
            // histogram of sizes < min_seg: bins < 256 via LDS (signed deltas as uint32 wrap)
            if (a0 < a.min_seg) { if (a0 < 256u) atomicSub(&lhist[a0], 1u); else atomicSub(&a.hist[a0], 1u); }
            if (sz < a.min_seg) { if (sz < 256u) atomicAdd(&lhist[sz], 1u); else atomicAdd(&a.hist[sz], 1u); }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) my_elim += __shfl_xor(my_elim, d, 64);
        if (lane == 0 && my_elim) {
            atomicAdd(&ctl->nelim, my_elim);
            if (target < 256u) atomicSub(&lhist[target], my_elim); else atomicSub(&a.hist[target], my_elim);
        }
        __syncthreads();
        if (threadIdx.x < a.min_seg && lhist[threadIdx.x] != 0u)
            atomicAdd(&a.hist[threadIdx.x], lhist[threadIdx.x]);          // wrapping add of the delta
        SP_MARK(ctl, 6, sprof)
        if (!(a.bar2 ? small_grid_barrier2(ctl, bar) : small_grid_barrier(ctl, G, a.poll))) return;
        SP_MARK(ctl, 15, sprof)
        if (gtid == 0) {
            const unsigned long long t = wall_clock64();
            ctl->plog[(uint32_t)ctl->tphase[3] & 63u][3] = (uint32_t)(t - tmark);
            ctl->tphase[1] += t - tmark; tmark = t; ctl->tphase[3] += 1;
        }
    }
}

// at most this many persistent loop kernels run at once (co-residency of all their workgroups)
#include <mutex>
#include <condition_variable>
static std::mutex g_small_mu;
static std::condition_variable g_small_cv;
static int g_small_running = 0;
#define SMALL_MAX_CONCURRENT 20
static const int g_small_max = getenv("SHEPSEG_SMALL_MAX") ? atoi(getenv("SHEPSEG_SMALL_MAX")) : SMALL_MAX_CONCURRENT;

// d_seg in place; *max_id in: seg.max(); out: seg.max() after the final relabel.
// sizes_in_origsz: ctx->origsz already holds makeSegSize(d_seg) (run_eliminate_single leaves it)
static int run_eliminate_small(shp_ctx *ctx, const void *d_img, int dtype, int nb, uint32_t nrows,
                               uint32_t ncols, int four, int min_seg_size, double max_spectral_diff,
                               uint32_t *d_seg, uint32_t *max_id, int64_t *num_elim,
                               int sizes_in_origsz = 0, const ImgGeom *geom_in = nullptr)
{
    const uint32_t n = nrows * ncols;
    const ImgGeom geom = geom_in ? *geom_in : geom_compact(n, ncols);
    const uint32_t S = *max_id;
    const size_t ns = (size_t)S + 2;
    const uint32_t min_seg = (uint32_t)(min_seg_size < 1 ? 1 : min_seg_size);
    *num_elim = 0;
    CHK(buf_ensure(ctx, ctx->segsz, ns * 4));
    CHK(buf_ensure(ctx, ctx->origsz, ns * 4));
    CHK(buf_ensure(ctx, ctx->off, ns * 4 + 16));
    CHK(buf_ensure(ctx, ctx->ssum, ns * nb * 4));
    CHK(buf_ensure(ctx, ctx->chnext, ns * 16));           // the chunk records (uint4)
    CHK(buf_ensure(ctx, ctx->mergeto, ns * 4));
    CHK(buf_ensure(ctx, ctx->tcount, ns * 4));
    CHK(buf_ensure(ctx, ctx->toff, ns * 4 + 16));
    CHK(buf_ensure(ctx, ctx->tfill, ns * 4));
    CHK(buf_ensure(ctx, ctx->tlist, (ns + 32) * 4));      // + header of the big-segment list
    CHK(buf_ensure(ctx, ctx->tsorted, ns * 4));
    CHK(buf_ensure(ctx, ctx->srclist, ns * 4));
    CHK(buf_ensure(ctx, ctx->tgtlist, ns * 4));
    CHK(buf_ensure(ctx, ctx->small, ((size_t)min_seg + 160) * 4));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns > n ? ns : n)));
    uint32_t *segsz = bp<uint32_t>(ctx->segsz), *origsz = bp<uint32_t>(ctx->origsz);
    uint32_t *off = bp<uint32_t>(ctx->off);
    uint4 *ch = (uint4 *)ctx->chnext.p;
    uint32_t *mergeto = bp<uint32_t>(ctx->mergeto);
    uint32_t *tcount = bp<uint32_t>(ctx->tcount), *toff = bp<uint32_t>(ctx->toff);
    uint32_t *tfill = bp<uint32_t>(ctx->tfill), *tlist = bp<uint32_t>(ctx->tlist);
    uint32_t *tsorted = bp<uint32_t>(ctx->tsorted);
    uint32_t *srclist = bp<uint32_t>(ctx->srclist), *tgtlist = bp<uint32_t>(ctx->tgtlist);
    float *ssum = bp<float>(ctx->ssum);
    uint32_t *hist = bp<uint32_t>(ctx->small);          // [0..min_seg] then nelim
    SmallCtl *ctl = (SmallCtl *)(hist + ((min_seg + 4u + 3u) & ~3u));
    hipStream_t st = ctx->stream;

    const uint32_t *sizes = sizes_in_origsz ? origsz : segsz;      // k_small_init copies them to both
    if (!sizes_in_origsz) CHK(run_seg_size(ctx, d_seg, n, S, segsz));
    if (n == 0 || S == 0) return 0;
    // CSR: pixels grouped by segment id, raster order inside (stable sort of (seg, index))
    uint32_t *pix = nullptr;
    int ps = prof_begin(ctx, PROF_SORT);
    CHK(build_segment_csr(ctx, d_seg, n, S, &pix, hist, min_seg + 2u));    // (also zeroes hist)
    prof_end(ctx, ps);
    uint32_t *stmp = bp<uint32_t>(ctx->scan_tmp);      // (fetched after sort_pairs: it may regrow)
    ArrFn szf{sizes};
    const uint32_t *off_boff = nullptr;
    CHK(scan_exclusive(ctx, szf, S + 1u, off, nullptr, stmp, &off_boff));
    const unsigned gs = grid_for((size_t)S + 1, 256);
    static_assert(sizeof(SmallCtl) % 4 == 0 && offsetof(SmallCtl, st) == 0, "SmallCtl layout");
    hipLaunchKernelGGL(k_small_init, dim3(gs), dim3(256), 0, st, sizes, segsz, origsz, ch,
                       mergeto, tcount, tfill, hist, S, min_seg, tlist, (uint32_t *)ctl,
                       (uint32_t)(sizeof(SmallCtl) / 4), off, off_boff); KCHK(ctx);
    ps = prof_begin(ctx, PROF_SPECTRA);
    CHK(launch_spectra(ctx, d_img, dtype, nb, n, pix, off, segsz, ssum, S, geom, tlist /* free until the pass loop */, 1u));
    prof_end(ctx, ps);

    const double thr2 = max_spectral_diff * max_spectral_diff;       // float64 square (N8)
    // one persistent kernel runs every pass (see k_small_loop)
    SmallCtl *pin = (SmallCtl *)(ctx->h_pinned + 16);            // (the kernel above initialised ctl)
    SmallArgs args;
    args.ctl = ctl; args.hist = hist; args.seg = d_seg; args.segsz = segsz; args.ssum = ssum;
    args.pix = pix; args.off = off; args.origsz = origsz; args.ch = ch;
    args.mergeto = mergeto; args.tcount = tcount; args.toff = toff; args.tfill = tfill;
    args.tlist = tlist; args.tsorted = tsorted; args.srclist = srclist; args.tgtlist = tgtlist;
    args.S = S; args.min_seg = min_seg; args.nrows = nrows; args.ncols = ncols;
    args.nb = nb; args.four = four; args.thr2 = thr2;
    static const int poll_env = getenv("SHEPSEG_SMALL_POLL") ? atoi(getenv("SHEPSEG_SMALL_POLL")) : 4;
    args.poll = poll_env < 1 ? 1 : poll_env;
    static const int bar2_env = getenv("SHEPSEG_SMALL_BAR2") ? atoi(getenv("SHEPSEG_SMALL_BAR2")) : 1;
    args.bar2 = bar2_env;
    static const int lists_env = getenv("SHEPSEG_SMALL_LISTS") ? atoi(getenv("SHEPSEG_SMALL_LISTS")) : 1;
    args.lists = lists_env;
    args.pin = (uint32_t *)pin;
    args.hopstat = getenv("SHEPSEG_SMALL_TIMING") ? &ctl->hopcnt : nullptr;
    pin->done = 0; pin->fail = 0;       // (a loop that gives up at a barrier leaves them so)
    fill_release(ctx, true);            // the pass loop is a latency-bound phase
    static const unsigned small_blocks = getenv("SHEPSEG_SMALL_BLOCKS") ? (unsigned)atoi(getenv("SHEPSEG_SMALL_BLOCKS")) : SMALL_BLOCKS;
    // how many loops fit the device at once, from the kernel's own occupancy (its register count decides:
    // 168 VGPRs = 3 workgroups per CU = 768 on the device = 12 loops of 64); never more than SHEPSEG_SMALL_MAX
    static const int cap = [] {
        int per_cu = 0, dev = 0, ncu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_small_loop, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 64;
        int c = (int)((unsigned)(per_cu * ncu) / (small_blocks ? small_blocks : 1u));
        c = c < 1 ? 1 : c;
        return c < g_small_max ? c : g_small_max;
    }();
    {
        std::unique_lock<std::mutex> lk(g_small_mu);
        g_small_cv.wait(lk, [] { return g_small_running < cap; });
        g_small_running++;
    }
    static const int dbg_skip_loop = getenv("SHEPSEG_DBG_SKIP_SMALL") ? atoi(getenv("SHEPSEG_DBG_SKIP_SMALL")) : 0;
    if (dbg_skip_loop) args.min_seg = 1;      // diagnostic only: the loop ends at once (wrong labels)
    walk_begin(ctx);
    st = ctx->stream;
    ps = prof_begin(ctx, PROF_SMALL_LOOP);           // events hug the kernel: no copies, no host waits
    hipLaunchKernelGGL(k_small_loop, dim3(small_blocks), dim3(256), 0, st, args);
    hipError_t lerr = hipGetLastError();
    prof_end(ctx, ps);
    hipError_t cerr = hipSuccess;
    hipError_t serr = hipStreamSynchronize(st);
    walk_end(ctx);
    {
        std::lock_guard<std::mutex> lk(g_small_mu);
        g_small_running--;
    }
    g_small_cv.notify_one();
    HIPCHK(ctx, lerr); HIPCHK(ctx, cerr); HIPCHK(ctx, serr);
    fill_acquire(ctx, 2);
    if (pin->fail || !pin->done)
        SHP_FAIL(ctx, SHP_ERR_STATE, "small-segment loop: grid barrier timed out (fail=%u done=%u)",
                 pin->fail, pin->done);
    if (getenv("SHEPSEG_SMALL_TIMING"))
        fprintf(stderr, "small loop: passes %llu  find+link %.2f ms  relabel+apply %.2f ms (S=%u)\n",
                pin->tphase[3], pin->tphase[0] / 1e5, pin->tphase[1] / 1e5, S);
    if (getenv("SHEPSEG_SMALL_TIMING")) {
        for (unsigned i = 0; i < 64u && i < pin->tphase[3]; i++)
            fprintf(stderr, "  pass %u: target %u, %u sources, find %.1f us, merge %.1f us, %u chain hops\n", i, pin->plog[i][0],
                    pin->plog[i][1], pin->plog[i][2] / 100.0, pin->plog[i][3] / 100.0, pin->phops[i]);
#ifdef SMALL_PROF
        fprintf(stderr, "  wave 0, passes >= 15, cycles: control %llu, scan | take + validate %llu, list barrier | record %llu, finds %llu (%llu sources), find barrier %llu, relabel %llu, apply %llu, merge barrier %llu\n",
                pin->prof[0], pin->prof[1], pin->prof[2], pin->prof[3], pin->prof[7], pin->prof[4], pin->prof[5], pin->prof[6], pin->prof[15]);
        fprintf(stderr, "    inside the finds (a wavefront per source: head, gather, ...; batches: slots, chunk records + hops, pixels + neighbour ids, sizes, distances + minima, -, decide + link): %llu, %llu, %llu, %llu, %llu, %llu, %llu\n",
                pin->prof[8], pin->prof[9], pin->prof[10], pin->prof[11], pin->prof[12], pin->prof[13], pin->prof[14]);
#endif
        fprintf(stderr, "  workgroups per XCD:");
        for (int i = 0; i < 16; i++) fprintf(stderr, " %u", pin->xcd_n[i]);
        fprintf(stderr, "\n");
    }
    *num_elim = (int64_t)pin->nelim;
    uint32_t new_max = 0;
    CHK(run_relabel(ctx, d_seg, n, segsz, S, &new_max));
    *max_id = new_max;
    return 0;
}

// makeSegmentLocations / buildSegmentSpectra on the device structures of the elimination stage: the
// CSR of pixels by segment (csr.h) with its offsets, and the ordered float32 sums.  d_seg: n labels
// 0..S on the device; on return ctx->off holds S + 2 offsets (exclusive scan of the sizes), *pix_out
// the n pixel indices, ctx->ssum (when d_img) the (S + 1) * nb sums, row 0 = the null segment's.
static int run_segment_tables(shp_ctx *ctx, const uint32_t *d_seg, uint32_t n, uint32_t ncols, uint32_t S,
                              const void *d_img, int dtype, int nb, uint32_t **pix_out)
{
    const size_t ns = (size_t)S + 2;
    CHK(buf_ensure(ctx, ctx->segsz, (ns + 1) * 4));
    CHK(buf_ensure(ctx, ctx->off, (ns + 1) * 4 + 16));
    CHK(buf_ensure(ctx, ctx->tlist, (ns + 32) * 4));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns > n ? ns : n)));
    uint32_t *segsz = bp<uint32_t>(ctx->segsz), *off = bp<uint32_t>(ctx->off);
    CHK(run_seg_size(ctx, d_seg, n, S + 1u, segsz));                 // (one entry past S: zero)
    uint32_t *pix = nullptr;
    CHK(build_segment_csr(ctx, d_seg, n, S, &pix));
    ArrFn szf{segsz};
    CHK(scan_exclusive(ctx, szf, S + 2u, off, nullptr, bp<uint32_t>(ctx->scan_tmp)));
    if (d_img) {
        CHK(buf_ensure(ctx, ctx->ssum, ns * nb * 4));
        uint32_t *biglist = bp<uint32_t>(ctx->tlist);
        HIPCHK(ctx, hipMemsetAsync(biglist, 0, 64, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->ssum.p, 0, ns * nb * 4, ctx->stream));      // ids without pixels: 0
        CHK(launch_spectra(ctx, d_img, dtype, nb, n, pix, off, segsz, bp<float>(ctx->ssum), S,
                           geom_compact(n, ncols), biglist, 0u));
    }
    *pix_out = pix;
    return 0;
}
