// sort.h -- stable LSD radix sort of (uint32 key, uint32 value) pairs, 8-bit digits.
// Wave-level multi-split: for each 64-item row the lanes that share a digit are found with
// 8 ballots (match-any), ranked with popcount, and placed after the running per-wave digit
// offset, so the sort is stable without any per-thread counters.  2048 items per workgroup.
// Per pass: histogram kernel, exclusive scan of the (digit-major) block histograms, scatter.
#pragma once
#include "common.h"
#include "scan.h"

#define SORT_ROWS 8u
#define SORT_TILE (4u * SORT_ROWS * 64u)   // items per 256-thread workgroup

template <unsigned ROWS>
__global__ __launch_bounds__(256) void k_sort_hist(const uint32_t *__restrict__ keys, uint32_t n,
                                                   int shift, uint32_t *__restrict__ hist,
                                                   uint32_t nblk)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t base = blockIdx.x * (4u * ROWS * 64u) + w * ROWS * 64u + lane;
    uint32_t d[ROWS];
#pragma unroll
    for (unsigned r = 0; r < ROWS; r++) {
        const uint32_t idx = base + r * 64u;
        d[r] = (idx < n) ? ((keys[idx] >> shift) & 255u) : 256u;
    }
#pragma unroll
    for (unsigned r = 0; r < ROWS; r++) {
        // label rasters come in runs: the lanes that share the first lane's digit add once (64
        // LDS atomics on one address would serialise), the others one by one
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d[r]);
        const unsigned long long same = __ballot(d[r] == d0);
        if (d0 < 256u && lane == 0) atomicAdd(&h[d0], (uint32_t)__popcll(same));
        if (d[r] != d0 && d[r] < 256u) atomicAdd(&h[d[r]], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid)
{
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// Direct form: every item goes straight to its place (label keys come in runs, so the lanes of a
// digit already write neighbouring addresses; 4 KiB of LDS per workgroup).
__global__ __launch_bounds__(256) void k_sort_scatter(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n, int shift,
    const uint32_t *__restrict__ histscan, uint32_t nblk, const uint32_t *__restrict__ boff)
{
    __shared__ uint32_t wcount[4][256];
    const unsigned w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t base = blockIdx.x * SORT_TILE + w * SORT_ROWS * 64u + lane;
    uint32_t k[SORT_ROWS], v[SORT_ROWS];
#pragma unroll
    for (unsigned r = 0; r < SORT_ROWS; r++) {
        const uint32_t idx = base + r * 64u;
        k[r] = (idx < n) ? keys_in[idx] : 0u;
        v[r] = (idx < n) ? (vals_in ? vals_in[idx] : idx) : 0u;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) wcount[i][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long lt = lanemask_lt();
    // per row: the lanes sharing this lane's digit (peers) and how many items of that digit the
    // wave's earlier rows hold (rowbase); the leader of each group bumps the wave's digit count
    unsigned long long peers[SORT_ROWS];
    uint32_t rowbase[SORT_ROWS];
#pragma unroll
    for (unsigned r = 0; r < SORT_ROWS; r++) {
        const bool valid = (base + r * 64u) < n;
        const uint32_t d = (k[r] >> shift) & 255u;
        peers[r] = match_digit(d, valid);
        uint32_t rb = 0;
        if (valid) rb = wcount[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers[r] & lt) == 0ull) wcount[w][d] = rb + (uint32_t)__popcll(peers[r]);
        __builtin_amdgcn_wave_barrier();
        rowbase[r] = rb;
    }
    __syncthreads();
    {
        const size_t hi = (size_t)threadIdx.x * nblk + blockIdx.x;
        uint32_t run = histscan[hi] + (boff ? boff[hi / SCAN_ITEMS] : 0u);      // lazy add-back of the scan
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t c = wcount[i][threadIdx.x];
            wcount[i][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (unsigned r = 0; r < SORT_ROWS; r++) {
        const bool valid = (base + r * 64u) < n;
        const uint32_t d = (k[r] >> shift) & 255u;
        if (valid) {
            const uint32_t pos = wcount[w][d] + rowbase[r] + (uint32_t)__popcll(peers[r] & lt);
            if (keys_out) keys_out[pos] = k[r];
            vals_out[pos] = v[r];
        }
    }
}

// Staged form of the scatter (keys without locality: the statistics' (value, segment) sorts).
// The scatter goes through LDS: the workgroup's 2048 items are first put in digit order there (the
// per-wave ranks give every item its place), then written out in that order, so that consecutive
// lanes write consecutive addresses within a digit's run -- with random digits a direct scatter
// issued 4-byte writes to 2048 unrelated addresses per workgroup, the staged one writes ~256 runs of
// 8 items (and whole 256-byte spans when the keys come in runs, as labels do).
template <unsigned ROWS>
__global__ __launch_bounds__(256) void k_sort_scatter_staged(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n, int shift,
    const uint32_t *__restrict__ histscan, uint32_t nblk, const uint32_t *__restrict__ boff)
{
    constexpr unsigned TILE = 4u * ROWS * 64u;
    __shared__ uint32_t wcount[4][256];
    __shared__ uint32_t gdelta[256];        // global position - position in the workgroup's digit order
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t skeys[TILE], svals[TILE];
    const unsigned w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t base = blockIdx.x * TILE + w * ROWS * 64u + lane;
    uint32_t k[ROWS], v[ROWS];
#pragma unroll
    for (unsigned r = 0; r < ROWS; r++) {
        const uint32_t idx = base + r * 64u;
        k[r] = (idx < n) ? keys_in[idx] : 0u;
        v[r] = (idx < n) ? (vals_in ? vals_in[idx] : idx) : 0u;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) wcount[i][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long lt = lanemask_lt();
    // per row: the lanes sharing this lane's digit (peers) and how many items of that digit the
    // wave's earlier rows hold (rowbase); the leader of each group bumps the wave's digit count
    unsigned long long peers[ROWS];
    uint32_t rowbase[ROWS];
#pragma unroll
    for (unsigned r = 0; r < ROWS; r++) {
        const bool valid = (base + r * 64u) < n;
        const uint32_t d = (k[r] >> shift) & 255u;
        peers[r] = match_digit(d, valid);
        uint32_t rb = 0;
        if (valid) rb = wcount[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers[r] & lt) == 0ull) wcount[w][d] = rb + (uint32_t)__popcll(peers[r]);
        __builtin_amdgcn_wave_barrier();
        rowbase[r] = rb;
    }
    __syncthreads();
    {
        // thread d: the workgroup's count of digit d, its exclusive scan over the digits (the digit's
        // first place in the staged order), and where each wave's items of that digit start
        const uint32_t c0 = wcount[0][threadIdx.x], c1 = wcount[1][threadIdx.x], c2 = wcount[2][threadIdx.x],
                       c3 = wcount[3][threadIdx.x];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t t = __shfl_up(incl, dd, 64);
            if (lane >= (unsigned)dd) incl += t;
        }
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        uint32_t lstart = incl - tot;
        for (unsigned i = 0; i < w; i++) lstart += wtot[i];
        const size_t hi = (size_t)threadIdx.x * nblk + blockIdx.x;
        const uint32_t gbase = histscan[hi] + (boff ? boff[hi / SCAN_ITEMS] : 0u);      // lazy add-back of the scan
        gdelta[threadIdx.x] = gbase - lstart;
        wcount[0][threadIdx.x] = lstart;
        wcount[1][threadIdx.x] = lstart + c0;
        wcount[2][threadIdx.x] = lstart + c0 + c1;
        wcount[3][threadIdx.x] = lstart + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (unsigned r = 0; r < ROWS; r++) {
        const bool valid = (base + r * 64u) < n;
        const uint32_t d = (k[r] >> shift) & 255u;
        if (valid) {
            const uint32_t lp = wcount[w][d] + rowbase[r] + (uint32_t)__popcll(peers[r] & lt);
            skeys[lp] = k[r];
            svals[lp] = v[r];
        }
    }
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * TILE;
    const uint32_t nvalid = tile0 < n ? (n - tile0 < TILE ? n - tile0 : TILE) : 0u;
#pragma unroll
    for (unsigned j = 0; j < TILE / 256u; j++) {
        const uint32_t i = threadIdx.x + j * 256u;
        if (i < nvalid) {
            const uint32_t key = skeys[i];
            const uint32_t pos = i + gdelta[(key >> shift) & 255u];
            if (keys_out) keys_out[pos] = key;
            vals_out[pos] = svals[i];
        }
    }
}

static inline int bits_for(uint32_t maxval)      // significant bits of the largest key
{
    int b = 1;
    while (b < 32 && (maxval >> b) != 0) b++;
    return b;
}

// Sort n pairs by the low `bits` bits of the key.  vals_in == nullptr means value = index.
// Uses ctx->sort_k0/sort_k1/sort_v1/pix as ping-pong storage and ctx->sort_hist/scan_tmp as
// scratch.  On return *keys_sorted / *vals_sorted point at the buffers holding the result
// (vals always end up in ctx->pix or ctx->sort_v1).  keys_sorted == nullptr: the caller only wants
// the values, the last pass does not write the keys.
// digits_out (optional): where the LAST pass's scanned digit histogram lies -- with a single pass (bits <= 8)
// entry d * nblk of it (+ the scan's block offset) is the position of the first key equal to d.
// (fit_elkan.h reads the clusters' ranges of the k-means row lists straight from these -- the LAST pass's scanned
//  histogram in ctx->sort_hist / scan_tmp, entry d * nblk, NON-staged layout -- between sort_pairs and the sums
//  kernel: whoever changes the histogram layout or lets another sort run in between must change sort_digit_start
//  and its caller with it; SHEPSEG_FIT_CHECK_DIGITS=1 compares them with a binary search of the sorted keys)
struct SortDigits { const uint32_t *hscan = nullptr, *boff = nullptr; uint32_t nblk = 0; int passes = 0; };
__device__ __forceinline__ uint32_t sort_digit_start(const uint32_t *hscan, const uint32_t *boff, uint32_t nblk,
                                                     uint32_t d, uint32_t n)
{
    if (d >= 256u) return n;
    const uint32_t hi = d * nblk;
    return hscan[hi] + (boff ? boff[hi / SCAN_ITEMS] : 0u);
}
static int sort_pairs(shp_ctx *ctx, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t n,
                      int bits, uint32_t **keys_sorted, uint32_t **vals_sorted, bool staged = false,
                      SortDigits *digits_out = nullptr)
{
    // SHEPSEG_SORT_WIDE=1: the staged form with 4096 items per workgroup (half the block histograms, 64-byte
    // runs per digit).  Measured on C5: histogram passes 3.8 -> 3.2 ms, scatter passes 9.2 -> 11.2 ms (116
    // VGPRs, 38 KiB of LDS: four workgroups per CU): slower overall, so off by default.
    static const bool wide = getenv("SHEPSEG_SORT_WIDE") && atoi(getenv("SHEPSEG_SORT_WIDE")) != 0;
    const uint32_t tile = (staged && wide) ? 2u * SORT_TILE : SORT_TILE;
    const uint32_t nblk = (n + tile - 1) / tile;
    int passes = (bits + 7) / 8;
    if (passes < 1) passes = 1;
    CHK(buf_ensure(ctx, ctx->sort_k0, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->sort_k1, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->sort_v1, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->pix, (size_t)n * 4));
    const size_t nh = (size_t)256 * (nblk ? nblk : 1);
    CHK(buf_ensure(ctx, ctx->sort_hist, 2 * nh * 4));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(nh)));
    uint32_t *hist = bp<uint32_t>(ctx->sort_hist), *hscan = hist + nh;
    // choose ping-pong so the values finish in ctx->pix
    uint32_t *kbuf[2] = {bp<uint32_t>(ctx->sort_k0), bp<uint32_t>(ctx->sort_k1)};
    uint32_t *vbuf[2];
    if (passes & 1) { vbuf[0] = bp<uint32_t>(ctx->pix); vbuf[1] = bp<uint32_t>(ctx->sort_v1); }
    else            { vbuf[0] = bp<uint32_t>(ctx->sort_v1); vbuf[1] = bp<uint32_t>(ctx->pix); }
    const uint32_t *kin = keys_in, *vin = vals_in;
    if (n == 0) { if (keys_sorted) *keys_sorted = kbuf[0]; *vals_sorted = bp<uint32_t>(ctx->pix); return 0; }
    for (int p = 0; p < passes; p++) {
        uint32_t *kout = (!keys_sorted && p == passes - 1) ? nullptr : kbuf[p & 1], *vout = vbuf[p & 1];
        if (tile == SORT_TILE)
            hipLaunchKernelGGL(k_sort_hist<SORT_ROWS>, dim3(nblk), dim3(256), 0, ctx->stream, kin, n, p * 8, hist, nblk);
        else
            hipLaunchKernelGGL(k_sort_hist<2u * SORT_ROWS>, dim3(nblk), dim3(256), 0, ctx->stream, kin, n, p * 8, hist, nblk);
        KCHK(ctx);
        ArrFn f{hist};
        const uint32_t *boff = nullptr;
        CHK(scan_exclusive(ctx, f, (uint32_t)nh, hscan, nullptr, bp<uint32_t>(ctx->scan_tmp), &boff));
        if (staged && tile != SORT_TILE)
            hipLaunchKernelGGL(k_sort_scatter_staged<2u * SORT_ROWS>, dim3(nblk), dim3(256), 0, ctx->stream, kin, vin,
                               kout, vout, n, p * 8, hscan, nblk, boff);
        else if (staged)
            hipLaunchKernelGGL(k_sort_scatter_staged<SORT_ROWS>, dim3(nblk), dim3(256), 0, ctx->stream, kin, vin, kout,
                               vout, n, p * 8, hscan, nblk, boff);
        else
            hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, ctx->stream, kin, vin, kout,
                               vout, n, p * 8, hscan, nblk, boff);
        KCHK(ctx);
        kin = kout; vin = vout;
        if (digits_out) { digits_out->hscan = hscan; digits_out->boff = boff; digits_out->nblk = nblk; digits_out->passes = passes; }
    }
    if (keys_sorted) *keys_sorted = (uint32_t *)kin;
    *vals_sorted = (uint32_t *)vin;
    return 0;
}
