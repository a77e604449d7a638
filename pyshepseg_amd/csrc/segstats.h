// segstats.h -- per-segment statistics of one image band (the "tilingstats" reductions).
//
// Replaces tilingstats.accumulateSegDict / checkSegComplete / calcStatsForCompletedSegs /
// SegmentStats / RatPage (tilingstats.py:466-617, :866-1008, :1949-2045).  The reference builds
// a dict-of-dicts histogram per segment tile by tile; here the exact per-segment, value-sorted
// multiset is produced by two stable radix sorts (by value, then by segment id), after which a
// segment's valid pixel values are one contiguous ascending run:
//   pixcount = run length, min / max = ends, percentile p = element ceil(n*p/100)-1 (p = 0 gives
//   the LAST element: the reference's while loop never runs, tilingstats.py:979-986), median =
//   percentile 50, mode = first longest run of equal values, mean = exact int64 sum / n
//   (float64) stored float32, stddev with the reference's mixed float64/float32 evaluation
//   (see oracle/shepseg_oracle.c orc_segstats; pinned against the reference's goldens).
// Nodata pixels and the null segment are keyed to segment 0 before sorting and never counted.
// HBM-bound: 6 B/px in, ~3+3 radix passes of 16 B/px, 4*nCols B per segment out.
#pragma once
#include "common.h"
#include "scan.h"
#include "sort.h"
#include "clump.h"      // k_run_count
#include "elim_small.h"

__global__ __launch_bounds__(256) void k_stats_keys(const uint32_t *__restrict__ seg,
                                                    const void *__restrict__ band, int dtype,
                                                    uint32_t n, uint32_t S, int has_null,
                                                    long long null_val, long long bias,
                                                    uint32_t *__restrict__ kseg,
                                                    uint32_t *__restrict__ kval)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const long long v = ld_px(band, dtype, p);
    uint32_t s = seg[p];
    if (s > S || (has_null && v == null_val)) s = 0;
    kseg[p] = s;
    kval[p] = (uint32_t)(v - bias);
}

// the statistics of one segment into their columns (a: its ascending value run, n values)
__device__ __forceinline__ void seg_stats_emit(uint32_t s, uint32_t n, const uint32_t *a, long long bias,
                                               const uint32_t *__restrict__ sel, int nstats, long long missing,
                                               long long *__restrict__ intcols, float *__restrict__ fltcols,
                                               size_t ns, long long vmin, long long vmax, long long vmode,
                                               float mean, float stddev)
{
    for (int i = 0; i < nstats; i++) {
        const uint32_t stat = sel[i * 5 + 1], ctype = sel[i * 5 + 2], cidx = sel[i * 5 + 3];
        const uint32_t param = sel[i * 5 + 4];
        double val = 0.0;
        if (stat == 4u || stat == 6u) {
            if (n == 0) val = (double)missing;
            else {
                const double pc = (stat == 4u) ? 50.0 : (double)param;
                const double t = (double)n * (pc / 100.0);
                uint32_t idx = n - 1;                            // t == 0: loop never runs
                if (t > 0.0) {
                    double ct = ceil(t);
                    if (ct > (double)n) ct = (double)n;
                    idx = (uint32_t)ct - 1u;
                }
                val = (double)((long long)a[idx] + bias);
            }
        } else if (stat == 0u) val = (double)vmin;
        else if (stat == 1u) val = (double)vmax;
        else if (stat == 2u) val = (double)mean;
        else if (stat == 3u) val = (double)stddev;
        else if (stat == 5u) val = (double)vmode;
        else if (stat == 7u) val = (double)n;
        if (ctype == 0u) intcols[(size_t)cidx * ns + s] = (long long)val;
        else fltcols[(size_t)cidx * ns + s] = (float)val;
    }
}

// the statistics of one segment from its ascending value run (a[0..n), biased values), into their columns
__device__ __forceinline__ void seg_stats_of_run(uint32_t s, uint32_t n, const uint32_t *a, long long bias,
                                                 const uint32_t *__restrict__ sel, int nstats, long long missing,
                                                 long long *__restrict__ intcols, float *__restrict__ fltcols, size_t ns)
{
    long long vmin = missing, vmax = missing, vmode = missing;
    float mean = (float)missing, stddev = (float)missing;
    if (n > 0) {
        vmin = (long long)a[0] + bias;
        vmax = (long long)a[n - 1] + bias;
        long long sum = 0;
        for (uint32_t i = 0; i < n; i++) sum += (long long)a[i] + bias;
        mean = (float)((double)sum / (double)n);
        // one pass, closing a group of equal values at its last member (a nested "while the next one is equal" loop
        // had the lanes of a wavefront -- a segment each -- wait for one another at every group)
        float var = 0.0f;
        uint32_t bestc = 0, c = 0;
        uint32_t x = a[0];
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t nx = i + 1u < n ? a[i + 1u] : ~x;
            c++;
            if (nx != x) {
                const double d = (double)((long long)x + bias) - (double)mean;
                const float term = (float)((double)c * (d * d));
                var = var + term;
                if (c > bestc) { bestc = c; vmode = (long long)x + bias; }
                c = 0;
            }
            x = nx;
        }
        stddev = (float)sqrt((double)var / (double)n);
    }
    seg_stats_emit(s, n, a, bias, sel, nstats, missing, intcols, fltcols, ns, vmin, vmax, vmode, mean, stddev);
}

#define SEGSTATS_BIG 4096u       // segments above this many valid pixels go to k_seg_stats_big
#define SEGSTATS_STAGE 3072u     // values a wavefront stages in LDS (12 KiB; 48 KiB per workgroup)
// one thread per segment over its ascending value run
__global__ __launch_bounds__(256) void k_seg_stats(const uint32_t *__restrict__ vals,
                                                   const uint32_t *__restrict__ off,
                                                   const uint32_t *__restrict__ cnt, uint32_t S,
                                                   long long bias, const uint32_t *__restrict__ sel,
                                                   int nstats, long long missing,
                                                   long long *__restrict__ intcols,
                                                   float *__restrict__ fltcols, uint32_t *biglist,
                                                   const uint8_t *__restrict__ only)
{
    // (only != nullptr: the rows of unflagged segments belong to somebody else -- the patch path below)
    // The 64 segments of a wavefront hold one contiguous span of the value array.  When that span is
    // short (many small segments: 50 M segments of 32 pixels in the C5 workload) the wavefront loads
    // it into LDS with coalesced reads and every thread walks its own run there; a thread reading
    // its 128 bytes straight from memory, twice, moved a cache line per load.
    __shared__ uint32_t stage[4][SEGSTATS_STAGE];
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t s_first = s - lane, s_last = s_first + 63u < S ? s_first + 63u : S;
    const uint32_t span0 = s_first <= S ? off[s_first] : 0u;
    const uint32_t span1 = s_first <= S ? off[s_last] + cnt[s_last] : 0u;
    const bool staged = span1 - span0 <= SEGSTATS_STAGE;          // (uniform per wavefront)
    if (staged) {
        for (uint32_t i = lane; i < span1 - span0; i += 64u) stage[wv][i] = vals[span0 + i];
        __builtin_amdgcn_wave_barrier();
    }
    if (s > S) return;
    const size_t ns = (size_t)S + 1;
    if (only && s != 0u && !only[s]) return;
    if (s == 0) {                       // null segment row: zeros (RatPage :1992-1996)
        for (int i = 0; i < nstats; i++) {
            if (sel[i * 5 + 2] == 0) intcols[(size_t)sel[i * 5 + 3] * ns] = 0;
            else fltcols[(size_t)sel[i * 5 + 3] * ns] = 0.0f;
        }
        return;
    }
    const uint32_t n = cnt[s];
    if (n > SEGSTATS_BIG) {             // a long run is a wavefront's work (k_seg_stats_big), not a thread's
        biglist[1u + atomicAdd(&biglist[0], 1u)] = s;
        return;
    }
    const uint32_t *a = staged ? &stage[wv][off[s] - span0] : vals + off[s];
    seg_stats_of_run(s, n, a, bias, sel, nstats, missing, intcols, fltcols, ns);
}

// Segments with long value runs, one wavefront each (a persistent grid over biglist: [0] = count,
// ids from [1]).  The sum is a parallel integer sum; the variance and the mode need the runs of equal
// values IN ORDER (the reference adds one float32 term per distinct value, ascending): the wavefront
// reads 64 values a step, finds the run boundaries with one ballot and closes the runs that end in
// the step one after the other -- at most one per distinct value in all, whatever the pixel count.
__global__ __launch_bounds__(256) void k_seg_stats_big(const uint32_t *__restrict__ vals,
                                                       const uint32_t *__restrict__ off,
                                                       const uint32_t *__restrict__ cnt, uint32_t S,
                                                       long long bias, const uint32_t *__restrict__ sel,
                                                       int nstats, long long missing,
                                                       long long *__restrict__ intcols,
                                                       float *__restrict__ fltcols, const uint32_t *biglist)
{
    const unsigned lane = lane_id();
    const uint32_t nbig = biglist[0];
    const size_t ns = (size_t)S + 1;
    for (uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6); e < nbig; e += gridDim.x * 4u) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)biglist[1u + e]);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[s]);
        const uint32_t *a = vals + (uint32_t)__builtin_amdgcn_readfirstlane((int)off[s]);
        long long sum = 0;
        for (uint32_t i = lane; i < n; i += 64u) sum += (long long)a[i] + bias;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        const float mean = (float)((double)sum / (double)n);
        float var = 0.0f;
        uint32_t bestc = 0, curx = (uint32_t)__builtin_amdgcn_readfirstlane((int)a[0]), curc = 0;
        long long vmode = missing;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + lane;
            const bool valid = i < n;
            const uint32_t x = valid ? a[i] : 0u;
            uint32_t prev = __shfl_up(x, 1, 64);
            if (lane == 0) prev = curx;
            unsigned long long heads = __ballot(valid && x != prev);
            const uint32_t nvalid = n - base < 64u ? n - base : 64u;
            uint32_t start = 0;
            while (heads) {
                const uint32_t h = (uint32_t)__builtin_ctzll(heads);
                heads &= heads - 1ull;
                curc += h - start;
                const double dd = (double)((long long)curx + bias) - (double)mean;
                var = var + (float)((double)curc * (dd * dd));
                if (curc > bestc) { bestc = curc; vmode = (long long)curx + bias; }
                curx = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)h);
                curc = 0;
                start = h;
            }
            curc += nvalid - start;
        }
        {
            const double dd = (double)((long long)curx + bias) - (double)mean;
            var = var + (float)((double)curc * (dd * dd));
            if (curc > bestc) { bestc = curc; vmode = (long long)curx + bias; }
        }
        if (lane == 0) {
            const float stddev = (float)sqrt((double)var / (double)n);
            seg_stats_emit(s, n, a, bias, sel, nstats, missing, intcols, fltcols, ns, (long long)a[0] + bias,
                           (long long)a[n - 1] + bias, vmode, mean, stddev);
        }
    }
}

// ---- small segments: statistics patch by patch, without the global sorts -----------------------------
// Where the average segment is a few dozen pixels (the C5 workload: 50 M segments of 4 x 8 pixels) nearly
// every segment lies inside one 32 x 64-pixel patch, and six radix passes over (segment, value) pairs of the
// WHOLE raster -- 16 B per pixel and pass -- sort what is already together.  Here a workgroup takes a patch:
// its pixels stay in registers (eight per thread), the distinct labels get slots in an LDS hash table with
// their pixel counts; a label whose count equals its count in the whole raster (tot[], one histogram pass
// before) is COMPLETE here: its valid values (at most SPP_MAXRUN) are gathered into one LDS run, sorted by
// rank, and reduced with the same code as k_seg_stats.  The pixels of every other label --
// segments that straddle patches or are too long -- are appended to a (segment, value) list that goes through
// the sorts as before; flagged[] tells the two paths' rows apart.  HBM traffic: the rasters twice (histogram,
// patches) + the columns, instead of ~100 B per pixel.
#define SPP_H 32u
#define SPP_W 64u
#define SPP_PPT 8u                  // pixels per thread: SPP_H * SPP_W / 256
#define SPP_SLOTS 1024u
#define SPP_MAXDIST 704u            // distinct labels a patch may hold before it gives up (load factor 0.69)
#define SPP_MAXRUN 64u
#define SPP_EMPTY 0xFFFFFFFFu

// tot[label] = pixels carrying it (labels above S are nobody's), one atomic per run of equal labels per wavefront
__global__ __launch_bounds__(256) void k_label_hist(const uint32_t *__restrict__ seg, uint32_t n, uint32_t S,
                                                    uint32_t *tot)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool inb = p < n;
    const unsigned lane = lane_id();
    const uint32_t v = inb ? seg[p] : 0u;
    const uint32_t pv = __shfl_up(v, 1, 64);
    const bool head = lane == 0 || pv != v || !inb;
    const unsigned long long heads = __ballot(head);
    if (head && inb && v != 0u && v <= S) {
        const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const unsigned nl = nxt ? (unsigned)__builtin_ctzll(nxt) : 64u;
        atomicAdd(&tot[v], nl - lane);
    }
}

// The same over a raster of known shape, a workgroup per 32 x 64-pixel patch (the patches of k_stats_patch): the runs'
// counts are first combined in an LDS hash table, and a label costs ONE global atomic per patch it touches instead of
// one per row run (C5: 4 x 8-pixel blocks, 200 M run heads -> 50 M atomics).  A label that finds no slot (a patch
// with more than 1024 labels) is added directly.
__global__ __launch_bounds__(256) void k_label_hist_patch(const uint32_t *__restrict__ seg, uint32_t nrows, uint32_t ncols,
                                                          uint32_t S, uint32_t *tot)
{
    __shared__ uint32_t key[SPP_SLOTS], cnt[SPP_SLOTS];
    for (uint32_t i = threadIdx.x; i < SPP_SLOTS; i += 256u) { key[i] = SPP_EMPTY; cnt[i] = 0u; }
    __syncthreads();
    const uint32_t x0 = blockIdx.x * SPP_W, y0 = blockIdx.y * SPP_H;
    const unsigned lane = lane_id();
    uint32_t sv[SPP_PPT];
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++) {          // (a wavefront = one row of the patch)
        const uint32_t pl = k * 256u + threadIdx.x;
        const uint32_t y = y0 + pl / SPP_W, x = x0 + (pl % SPP_W);
        sv[k] = (y < nrows && x < ncols) ? seg[(size_t)y * ncols + x] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++) {
        const uint32_t v = sv[k];
        const uint32_t pv = __shfl_up(v, 1, 64);
        const bool head = lane == 0 || pv != v;
        const unsigned long long heads = __ballot(head);
        if (!head || v == 0u || v > S) continue;
        const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const uint32_t nl = (nxt ? (unsigned)__builtin_ctzll(nxt) : 64u) - lane;
        uint32_t h = (v * 2654435761u) >> 22;
        bool placed = false;
        for (uint32_t probe = 0; probe < SPP_SLOTS && !placed; probe++) {
            const uint32_t kk = key[h];
            if (kk == v) placed = true;
            else if (kk == SPP_EMPTY) {
                const uint32_t old = atomicCAS(&key[h], SPP_EMPTY, v);
                placed = old == SPP_EMPTY || old == v;
            }
            if (!placed) h = (h + 1u) & (SPP_SLOTS - 1u);
        }
        if (placed) atomicAdd(&cnt[h], nl);
        else atomicAdd(&tot[v], nl);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < SPP_SLOTS; i += 256u)
        if (key[i] != SPP_EMPTY) atomicAdd(&tot[key[i]], cnt[i]);
}

// every row as for a segment without valid pixels; the paths below overwrite what they compute
__global__ __launch_bounds__(256) void k_stats_prefill(uint32_t S, long long bias, const uint32_t *__restrict__ sel,
                                                       int nstats, long long missing, long long *__restrict__ intcols,
                                                       float *__restrict__ fltcols)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s > S) return;
    const size_t ns = (size_t)S + 1;
    if (s == 0) {
        for (int i = 0; i < nstats; i++) {
            if (sel[i * 5 + 2] == 0) intcols[(size_t)sel[i * 5 + 3] * ns] = 0;
            else fltcols[(size_t)sel[i * 5 + 3] * ns] = 0.0f;
        }
        return;
    }
    seg_stats_of_run(s, 0u, nullptr, bias, sel, nstats, missing, intcols, fltcols, ns);
}

// (eight workgroups per CU -- 64 VGPRs, 20 KiB of LDS -- is worth more here than any unrolling: 23.4 ms at five,
//  21.5 at eight on C5 with the same instructions)
// (DT: the band's pixel type at compile time -- eight loads issued together instead of eight trips through a
//  run-time switch, each with its own wait)
template <int DT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_stats_patch(const uint32_t *__restrict__ seg, const void *__restrict__ band,
                                                     uint32_t nrows, uint32_t ncols, uint32_t S,
                                                     int has_null, long long null_val, long long bias,
                                                     const uint32_t *__restrict__ tot, const uint32_t *__restrict__ sel,
                                                     int nstats, long long missing, long long *__restrict__ intcols,
                                                     float *__restrict__ fltcols, uint8_t *__restrict__ flagged,
                                                     uint32_t *__restrict__ left_seg, uint32_t *__restrict__ left_val,
                                                     uint32_t *left_count)
{
    // (LDS per workgroup sets how many of them a CU holds, and the kernel lives on that: the fill counters share
    //  cnt[]'s low half once the pixel totals have been compared, the offsets are 16-bit)
    __shared__ uint32_t key[SPP_SLOTS], cnt[SPP_SLOTS];
    __shared__ uint16_t offs[SPP_SLOTS];
    // (runs padded to 16 bytes for four-value LDS reads in the rank loop were tried: 23.6 ms on C5 against 21.5 --
    //  the LDS the padding takes costs more in resident workgroups than the wider reads save)
    // (and one unused entry behind every run: runs of equal length would start at multiples of it, and the labels of
    //  a wavefront read the same few banks all through the rank loop)
    __shared__ uint32_t runs[SPP_H * SPP_W + SPP_MAXDIST];
    __shared__ uint32_t s_ndist, s_left, s_leftbase, s_wsum[4], s_nc;
    __shared__ uint16_t clist[SPP_SLOTS];       // the complete labels' slots, compacted: consecutive threads take them
    for (uint32_t i = threadIdx.x; i < SPP_SLOTS; i += 256u) { key[i] = SPP_EMPTY; cnt[i] = 0u; }
    if (threadIdx.x == 0) { s_ndist = 0u; s_left = 0u; s_nc = 0u; }
    __syncthreads();
    const uint32_t x0 = blockIdx.x * SPP_W, y0 = blockIdx.y * SPP_H;
    constexpr bool narrow = DT != SHP_I32 && DT != SHP_U32;         // biased values below 2^16
    // this thread's pixels: (segment, biased value, slot); the same thread keeps them through every phase
    uint32_t ps[SPP_PPT], pv[SPP_PPT], pslot[SPP_PPT];
    bool pvalid[SPP_PPT];
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++) {
        const uint32_t pl = k * 256u + threadIdx.x;
        const uint32_t y = y0 + pl / SPP_W, x = x0 + (pl % SPP_W);
        ps[k] = 0u; pv[k] = 0u; pvalid[k] = false;
        if (y < nrows && x < ncols) {
            const size_t p = (size_t)y * ncols + x;
            const uint32_t sg = seg[p];
            const long long v = ld_t<DT>(band, p);
            if (sg != 0u && sg <= S) {
                ps[k] = sg;
                pv[k] = (uint32_t)(v - bias);
                pvalid[k] = !(has_null && v == null_val);
            }
        }
    }
    // ---- slots and counts (all pixels in the low half, valid ones in the high half) ----
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++) {
        pslot[k] = SPP_SLOTS;
        if (ps[k] == 0u) continue;
        uint32_t h = (ps[k] * 2654435761u) >> 22;
        for (uint32_t probe = 0; probe < SPP_SLOTS; probe++) {
            const uint32_t kk = key[h];
            if (kk == ps[k]) { pslot[k] = h; break; }
            if (kk == SPP_EMPTY) {
                const uint32_t old = atomicCAS(&key[h], SPP_EMPTY, ps[k]);
                if (old == SPP_EMPTY) { atomicAdd(&s_ndist, 1u); pslot[k] = h; break; }
                if (old == ps[k]) { pslot[k] = h; break; }
            }
            h = (h + 1u) & (SPP_SLOTS - 1u);
        }
        if (pslot[k] < SPP_SLOTS) atomicAdd(&cnt[pslot[k]], 1u + (pvalid[k] ? 65536u : 0u));
    }
    __syncthreads();
    const bool crowded = s_ndist > SPP_MAXDIST;          // (a pixel may then have found no slot at all)
    // ---- which labels are complete here; run offsets by a scan of their valid counts ----
    uint32_t mine[4], msum = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) {
        const uint32_t sl = threadIdx.x * 4u + q;
        const uint32_t kk = key[sl], c = cnt[sl];
        uint32_t nv = 0;
        if (kk != SPP_EMPTY) {
            const bool complete = !crowded && (c & 0xFFFFu) == tot[kk] && (c >> 16) <= SPP_MAXRUN;
            if (complete) { nv = (c >> 16) + 1u; clist[atomicAdd(&s_nc, 1u)] = (uint16_t)sl; }
            else flagged[kk] = 1;
            // bit 31: complete; the low half (all pixels) has served and becomes the run's fill counter
            cnt[sl] = complete ? ((c & 0x7FFF0000u) | 0x80000000u) : (c & 0x7FFFFFFFu);
        }
        mine[q] = nv;
        msum += nv;
    }
    uint32_t incl = msum;                                  // inclusive scan over the workgroup's 256 threads
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t base = incl - msum;
    for (unsigned w2 = 0; w2 < wv; w2++) base += s_wsum[w2];
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) { offs[threadIdx.x * 4u + q] = (uint16_t)base; base += mine[q]; }
    __syncthreads();
    // ---- values of complete labels into their runs, the rest into the list ----
    // (a run's first lane probing and adding for the run -- one atomic per run instead of one per pixel -- was built
    //  twice, rounds 3 and 4: 27.1 -> 30.7 ms and 15.9 -> 17.9 ms; the masks and shuffles cost more than the
    //  eight-way same-address atomics they save)
    // rpos[k]: the place in the label's run; 0x80000000 | place in the left-over list; 0xFFFFFFFF: neither
    uint32_t rpos[SPP_PPT];
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++) {
        rpos[k] = 0xFFFFFFFFu;
        if (ps[k] == 0u || !pvalid[k]) continue;           // (nodata pixels count for completeness only)
        const uint32_t sl = pslot[k];
        if (sl < SPP_SLOTS && (cnt[sl] & 0x80000000u)) {
            rpos[k] = atomicAdd(&cnt[sl], 1u) & 0xFFFFu;
            runs[offs[sl] + rpos[k]] = narrow ? (pv[k] << 6) | rpos[k] : pv[k];
        } else rpos[k] = 0x80000000u | atomicAdd(&s_left, 1u);
    }
    if (crowded) {                                          // pixels without a slot: their labels are flagged here
#pragma unroll
        for (uint32_t k = 0; k < SPP_PPT; k++)
            if (ps[k] != 0u && pslot[k] >= SPP_SLOTS) flagged[ps[k]] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_left) s_leftbase = atomicAdd(left_count, s_left);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++)
        if (rpos[k] != 0xFFFFFFFFu && (rpos[k] & 0x80000000u)) {
            const uint32_t li = s_leftbase + (rpos[k] & 0x7FFFFFFFu);
            left_seg[li] = ps[k]; left_val[li] = pv[k];
            rpos[k] = 0xFFFFFFFFu;                           // (not a member of a run)
        }
    // ---- the runs sorted by RANK: every pixel counts the values of its run that come before it (smaller, or equal
    //      and stored earlier) -- all threads busy on independent LDS reads, where a thread per run doing an
    //      insertion sort was a chain of dependent LDS round trips on a quarter of the lanes ----
    // (the rank takes the place of rpos[k] once it is known, and the final place offs + rank the place of that)
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++) {
        if (rpos[k] == 0xFFFFFFFFu) continue;
        const uint32_t sl = pslot[k];
        const uint32_t n = (cnt[sl] >> 16) & 0x7FFFu, o = offs[sl];
        const uint32_t v = pv[k];
        uint32_t r = 0;
        const uint32_t *q = &runs[o];
        if (narrow) {
            // 8/16-bit bands: the run holds value << 6 | position (positions < SPP_MAXRUN = 64), and "smaller, or equal
            // and stored earlier" is ONE unsigned compare -- this loop is n iterations for each of the n values, a third
            // of the kernel's instructions: plain offsets from one base, so that a value costs a compare and an add
            const uint32_t me = (v << 6) | rpos[k];
            uint32_t j = 0;
            for (; j + 8u <= n; j += 8u) {
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) r += q[j + u] < me ? 1u : 0u;
            }
            for (; j < n; j++) r += q[j] < me ? 1u : 0u;
        } else {
            const uint32_t rp = rpos[k];
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t w = q[j];
                r += (w < v || (w == v && j < rp)) ? 1u : 0u;
            }
        }
        rpos[k] = o + r;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SPP_PPT; k++)
        if (rpos[k] != 0xFFFFFFFFu) runs[rpos[k]] = pv[k];
    __syncthreads();
    // ---- a thread per complete label -- consecutive threads take consecutive entries of the compacted list, so
    //      that a wavefront's lanes all work (a thread per SLOT left 4 lanes in 64 busy, four times over) ----
    const size_t ns = (size_t)S + 1;
    for (uint32_t ci = threadIdx.x; ci < s_nc; ci += 256u) {
        const uint32_t sl = clist[ci];
        const uint32_t n = (cnt[sl] >> 16) & 0x7FFFu;
        seg_stats_of_run(key[sl], n, &runs[offs[sl]], bias, sel, nstats, missing, intcols, fltcols, ns);
    }
}

// d_seg / d_band: device rasters of n pixels.  Outputs are HOST arrays.
// nrows x ncols = n when the caller knows the raster's shape (0, 0 otherwise): with small segments the
// statistics are then computed patch by patch (k_stats_patch) and only what is left over is sorted.
// The result columns go to pageable host arrays (the caller's numpy columns): the runtime stages such a copy
// through its own pinned buffers, one pipeline per stream, so the integer and the float columns travel on two
// streams at once (the second waits on an event for the kernels that wrote them).
static int segstats_download(shp_ctx *ctx, void *intcols_out, const void *d_int, size_t int_bytes,
                             void *fltcols_out, const void *d_flt, size_t flt_bytes)
{
    hipStream_t st = ctx->stream;
    if (int_bytes && flt_bytes && int_bytes + flt_bytes > (size_t)(8u << 20)) {
        CHK(ensure_stream2(ctx));
        hipEvent_t ev = nullptr;
        HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ev, st));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ev, 0));
        HIPCHK(ctx, hipMemcpyAsync(fltcols_out, d_flt, flt_bytes, hipMemcpyDeviceToHost, ctx->stream2));
        HIPCHK(ctx, hipMemcpyAsync(intcols_out, d_int, int_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream2));
        HIPCHK(ctx, hipEventDestroy(ev));
        return 0;
    }
    if (int_bytes) HIPCHK(ctx, hipMemcpyAsync(intcols_out, d_int, int_bytes, hipMemcpyDeviceToHost, st));
    if (flt_bytes) HIPCHK(ctx, hipMemcpyAsync(fltcols_out, d_flt, flt_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

static int run_segstats(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                        uint32_t n, uint32_t S, int has_null, int64_t null_val,
                        const uint32_t *sel_host, int nstats, int64_t missing,
                        int64_t *intcols_out, float *fltcols_out, uint32_t nrows = 0, uint32_t ncols = 0,
                        long long **dev_int = nullptr, float **dev_flt = nullptr)
{
    // (dev_int / dev_flt given: the columns stay in the context's workspace, nothing is copied to the host)
    hipStream_t st = ctx->stream;
    const size_t ns = (size_t)S + 1;
    int nint = 0, nflt = 0;
    for (int i = 0; i < nstats; i++) {
        const uint32_t stat = sel_host[i * 5 + 1], ctype = sel_host[i * 5 + 2];
        if (stat > 7u || ctype > 1u) SHP_FAIL(ctx, SHP_ERR_ARG, "bad statsSelection entry %d", i);
        if (ctype == 0) nint++; else nflt++;
    }
    if ((size_t)nstats * 20 + 64 > SHP_PINNED_BYTES) SHP_FAIL(ctx, SHP_ERR_ARG, "too many statistics");
    long long bias = 0;
    int valbits = 32;
    switch (dtype) {
    case SHP_U8: valbits = 8; break;
    case SHP_U16: valbits = 16; break;
    case SHP_I16: valbits = 16; bias = -32768; break;
    case SHP_I32: bias = -2147483648ll; break;
    default: break;
    }
    CHK(buf_ensure(ctx, ctx->aux, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->segsz, (ns + 1) * 4));
    CHK(buf_ensure(ctx, ctx->off, (ns + 1) * 4 + 16));
    CHK(buf_ensure(ctx, ctx->small, 4096 + (size_t)nstats * 20));
    CHK(buf_ensure(ctx, ctx->ssum, ((size_t)nint * 8 + (size_t)nflt * 4) * ns + 64));
    uint32_t *kval = bp<uint32_t>(ctx->aux), *kseg = bp<uint32_t>(ctx->aux2);
    uint32_t *cnt = bp<uint32_t>(ctx->segsz), *off = bp<uint32_t>(ctx->off);
    uint32_t *d_sel = bp<uint32_t>(ctx->small) + 256;
    long long *d_int = (long long *)ctx->ssum.p;
    float *d_flt = (float *)(d_int + (size_t)nint * ns);
    HIPCHK(ctx, hipStreamSynchronize(st));
    uint32_t *pin = ctx->h_pinned + 16;
    memcpy(pin, sel_host, (size_t)nstats * 20);
    HIPCHK(ctx, hipMemcpyAsync(d_sel, pin, (size_t)nstats * 20, hipMemcpyHostToDevice, st));
    const int ps = prof_begin(ctx, PROF_SEGSTATS);      // device time of the kernels (keys .. statistics)
    // (SHEPSEG_STATS_PATCH=0: never; =1: whenever the shape is known; default: when the average segment has
    //  at most SPP_MAXRUN pixels)
    const int patch_env = getenv("SHEPSEG_STATS_PATCH") ? atoi(getenv("SHEPSEG_STATS_PATCH")) : -1;
    const bool patches = n && nrows && ncols && (uint64_t)nrows * ncols == n && patch_env != 0 &&
                         (patch_env == 1 || (uint64_t)n <= (uint64_t)SPP_MAXRUN * ((uint64_t)S + 1));
    const uint8_t *only = nullptr;
    uint32_t nsort = n;                                   // pairs that go through the sorts
    if (patches) {
        CHK(buf_ensure(ctx, ctx->tcount, (ns + 1) * 4));
        CHK(buf_ensure(ctx, ctx->mergeto, ns + 64));
        uint32_t *tot = bp<uint32_t>(ctx->tcount);
        uint8_t *flagged = bp<uint8_t>(ctx->mergeto);
        uint32_t *d_left = bp<uint32_t>(ctx->small);      // (word 0; the selection sits behind word 256)
        HIPCHK(ctx, hipMemsetAsync(tot, 0, ns * 4, st));
        HIPCHK(ctx, hipMemsetAsync(flagged, 0, ns, st));
        HIPCHK(ctx, hipMemsetAsync(d_left, 0, 4, st));
        hipLaunchKernelGGL(k_label_hist_patch, dim3(grid_for(ncols, SPP_W), grid_for(nrows, SPP_H)), dim3(256), 0, st, d_seg,
                           (uint32_t)nrows, (uint32_t)ncols, S, tot); KCHK(ctx);
        hipLaunchKernelGGL(k_stats_prefill, dim3(grid_for(ns, 256)), dim3(256), 0, st, S, bias, d_sel, nstats,
                           (long long)missing, d_int, d_flt); KCHK(ctx);
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_stats_patch<DT>, dim3(grid_for(ncols, SPP_W), grid_for(nrows, SPP_H)),
                                                 dim3(256), 0, st, d_seg, d_band, (uint32_t)nrows, (uint32_t)ncols, S, has_null,
                                                 (long long)null_val, bias, tot, d_sel, nstats, (long long)missing, d_int,
                                                 d_flt, flagged, kseg, kval, d_left));
        KCHK(ctx);
        CHK(read_u32(ctx, d_left, &nsort));
        only = flagged;
    } else if (n) {
        hipLaunchKernelGGL(k_stats_keys, dim3(grid_for(n, 256)), dim3(256), 0, st, d_seg, d_band, dtype, n,
                           S, has_null, (long long)null_val, bias, kseg, kval); KCHK(ctx);
    }
    if (patches && nsort == 0) {                          // every segment was complete in its patch
        prof_end(ctx, ps);
        if (dev_int) { *dev_int = d_int; *dev_flt = d_flt; return 0; }
        return segstats_download(ctx, intcols_out, d_int, (size_t)nint * ns * 8, fltcols_out, d_flt, (size_t)nflt * ns * 4);
    }
    n = nsort;
    // sort by value (payload: segment key), then stably by segment key (payload: value)
    uint32_t *k1 = nullptr, *v1 = nullptr, *k2 = nullptr, *v2 = nullptr;
    CHK(sort_pairs(ctx, kval, kseg, n, valbits, &k1, &v1, true));            // k1 = values, v1 = seg keys
    // the second sort reads the first one's result where it lies unless its own first pass would
    // write into the same ping-pong buffers (that depends on the two pass counts' parities)
    const uint32_t *in_seg = v1, *in_val = k1;
    {
        const int passes2 = (bits_for(S) + 7) / 8;
        const uint32_t *out_k = bp<uint32_t>(ctx->sort_k0);
        const uint32_t *out_v = (passes2 & 1) ? bp<uint32_t>(ctx->pix) : bp<uint32_t>(ctx->sort_v1);
        if (n && (k1 == out_k || k1 == out_v || v1 == out_k || v1 == out_v)) {
            HIPCHK(ctx, hipMemcpyAsync(kval, k1, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            HIPCHK(ctx, hipMemcpyAsync(kseg, v1, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            in_seg = kseg; in_val = kval;
        }
    }
    CHK(sort_pairs(ctx, in_seg, in_val, n, bits_for(S), &k2, &v2, true));    // k2 = seg keys, v2 = values
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(ns)));
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, ns * 4, st));
    if (n) {
        hipLaunchKernelGGL(k_run_count, dim3(grid_for(n, 256)), dim3(256), 0, st, k2, n, cnt, 0u, 0); KCHK(ctx);
    }
    ArrFn cf{cnt};
    CHK(scan_exclusive(ctx, cf, (uint32_t)ns, off, nullptr, bp<uint32_t>(ctx->scan_tmp)));
    // (the list of long segments: in the first sort's key buffer, free by now)
    uint32_t *biglist = kval;
    HIPCHK(ctx, hipMemsetAsync(biglist, 0, 4, st));
    hipLaunchKernelGGL(k_seg_stats, dim3(grid_for(ns, 256)), dim3(256), 0, st, v2, off, cnt, S, bias, d_sel,
                       nstats, (long long)missing, d_int, d_flt, biglist, only); KCHK(ctx);
    hipLaunchKernelGGL(k_seg_stats_big, dim3(512), dim3(256), 0, st, v2, off, cnt, S, bias, d_sel, nstats,
                       (long long)missing, d_int, d_flt, biglist); KCHK(ctx);      // (only flagged ones got onto the list)
    prof_end(ctx, ps);
    if (dev_int) { *dev_int = d_int; *dev_flt = d_flt; return 0; }
    return segstats_download(ctx, intcols_out, d_int, (size_t)nint * ns * 8, fltcols_out, d_flt, (size_t)nflt * ns * 4);
}

// ---- multi-GPU split: the pixels of segments that straddle a rank boundary ------------------
// (seg id, band value) of every pixel whose segment is flagged, compacted (any order) with one
// global atomic per 4096 pixels.  count may exceed cap: only the first cap pairs are stored.
__global__ __launch_bounds__(256) void k_gather_flagged(const uint32_t *__restrict__ seg,
                                                        const void *__restrict__ band, int dtype,
                                                        uint32_t n, uint32_t S,
                                                        const uint8_t *__restrict__ flags,
                                                        uint32_t *__restrict__ out_seg,
                                                        long long *__restrict__ out_val,
                                                        uint32_t cap, uint32_t *count)
{
    __shared__ uint32_t s_buf[4096];
    __shared__ uint32_t s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const unsigned lane = lane_id();
    for (uint32_t it = 0; it < 16u; it++) {
        const uint32_t p = blockIdx.x * 4096u + it * 256u + threadIdx.x;
        bool take = false;
        if (p < n) {
            const uint32_t sg = seg[p];
            take = sg != 0u && sg <= S && flags[sg] != 0;
        }
        const unsigned long long m = __ballot(take);
        if (m != 0ull) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            if (take) s_buf[wbase + (uint32_t)__popcll(m & lanemask_lt())] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(count, s_cnt) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < s_cnt; i += 256u) {
        const uint32_t o = s_base + i;
        if (o < cap) {
            const uint32_t p = s_buf[i];
            out_seg[o] = seg[p];
            out_val[o] = ld_px(band, dtype, p);
        }
    }
}

static int run_gather_flagged(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype,
                              uint32_t n, uint32_t S, const uint8_t *flags_host, uint32_t cap,
                              uint32_t *seg_out, int64_t *val_out, int64_t *count_out)
{
    hipStream_t st = ctx->stream;
    const size_t ns = (size_t)S + 1;
    CHK(buf_ensure(ctx, ctx->small, ns + 64));
    CHK(buf_ensure(ctx, ctx->aux, (size_t)cap * 4 + 64));
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)cap * 8 + 64));
    uint8_t *d_flags = bp<uint8_t>(ctx->small) + 64;
    uint32_t *d_count = bp<uint32_t>(ctx->small);
    HIPCHK(ctx, hipMemcpyAsync(d_flags, flags_host, ns, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(d_count, 0, 4, st));
    if (n) {
        hipLaunchKernelGGL(k_gather_flagged, dim3(grid_for(n, 4096)), dim3(256), 0, st, d_seg, d_band, dtype,
                           n, S, d_flags, bp<uint32_t>(ctx->aux), (long long *)ctx->aux2.p, cap, d_count);
        KCHK(ctx);
    }
    uint32_t cnt = 0;
    CHK(read_u32(ctx, d_count, &cnt));
    const uint32_t take = cnt < cap ? cnt : cap;
    if (take) {
        HIPCHK(ctx, hipMemcpyAsync(seg_out, ctx->aux.p, (size_t)take * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(val_out, ctx->aux2.p, (size_t)take * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    *count_out = (int64_t)cnt;
    return 0;
}

// ---- multi-GPU split, device-resident (pyshepseg_amd/distributed.py) ---------------------------
// Rank-local part.  The statistics of this rank's rows are computed as on one GPU; then every id is
// classified against the global histogram (the reference's segSize, tilingstats.py:165): a segment whose
// local pixel count equals it is complete here and its row is final (checkSegComplete, :518-553); a segment
// with fewer is a straddler -- its row is cleared and its pixels are packed as (id, value) pairs for the
// exchange; ids nobody holds (global count 0, row 0 among them) keep their "missing" row on the rank that is
// told to (keep_unheld), so that the columns of all ranks ADD UP to the single-GPU columns.
// cols: nint int64 columns then nflt float columns of S + 1 rows, in device memory of the caller.
__global__ __launch_bounds__(256) void k_dstats_classify(const uint32_t *__restrict__ lh,
                                                         const uint32_t *__restrict__ gh, uint32_t S,
                                                         int keep_unheld, int nint, int nflt,
                                                         const long long *__restrict__ src_int,
                                                         const float *__restrict__ src_flt,
                                                         long long *__restrict__ dst_int,
                                                         float *__restrict__ dst_flt,
                                                         uint8_t *__restrict__ flags,
                                                         unsigned long long *counters)
{
    const size_t ns = (size_t)S + 1;
    const size_t id = (size_t)blockIdx.x * 256u + threadIdx.x;
    bool strad = false;
    uint32_t l = 0;
    if (id < ns) {
        l = id == 0 ? 0u : lh[id];
        const uint32_t g = id == 0 ? 0u : gh[id];
        strad = l > 0u && l < g;
        const bool keep = (l > 0u && l == g) || (keep_unheld && g == 0u);
        flags[id] = strad ? 1 : 0;
        for (int c = 0; c < nint; c++) dst_int[(size_t)c * ns + id] = keep ? src_int[(size_t)c * ns + id] : 0ll;
        for (int c = 0; c < nflt; c++) dst_flt[(size_t)c * ns + id] = keep ? src_flt[(size_t)c * ns + id] : 0.0f;
    }
    // counters[0] += pixels of straddlers, counters[1] += straddling segments (one atomic pair per wavefront)
    const unsigned long long m = __ballot(strad);
    if (m != 0ull) {
        unsigned long long px = strad ? (unsigned long long)l : 0ull;
        for (int o = 32; o > 0; o >>= 1) px += __shfl_xor(px, o);
        if (lane_id() == 0) { atomicAdd(&counters[0], px); atomicAdd(&counters[1], (unsigned long long)__popcll(m)); }
    }
}

static int run_dstats_local(shp_ctx *ctx, const uint32_t *d_seg, const void *d_band, int dtype, uint32_t nrows,
                            uint32_t ncols, uint32_t S, int has_null, int64_t null_val, const uint32_t *sel_host,
                            int nstats, int64_t missing, const uint32_t *d_hist, int keep_unheld, void *d_cols,
                            uint32_t **d_pair_seg, long long **d_pair_val, int64_t *n_pairs, int64_t *n_strad)
{
    hipStream_t st = ctx->stream;
    const size_t ns = (size_t)S + 1;
    const uint32_t n = nrows * ncols;
    int nint = 0, nflt = 0;
    for (int i = 0; i < nstats; i++) { if (sel_host[i * 5 + 2] == 0) nint++; else nflt++; }
    long long *di = nullptr;
    float *df = nullptr;
    CHK(run_segstats(ctx, d_seg, d_band, dtype, n, S, has_null, null_val, sel_host, nstats, missing, nullptr, nullptr,
                     nrows, ncols, &di, &df));
    // the local label histogram (all pixels of a label, valid or not) | flags | two counters
    CHK(buf_ensure(ctx, ctx->chnext, ns * 4 + 64));
    CHK(buf_ensure(ctx, ctx->chtail, ns + 64));
    uint32_t *lh = bp<uint32_t>(ctx->chnext);
    unsigned long long *ctr = (unsigned long long *)bp<uint8_t>(ctx->chtail);
    uint8_t *flags = bp<uint8_t>(ctx->chtail) + 64;
    HIPCHK(ctx, hipMemsetAsync(lh, 0, ns * 4, st));
    HIPCHK(ctx, hipMemsetAsync(ctr, 0, 16, st));
    if (n) { hipLaunchKernelGGL(k_label_hist, dim3(grid_for(n, 256)), dim3(256), 0, st, d_seg, n, S, lh); KCHK(ctx); }
    hipLaunchKernelGGL(k_dstats_classify, dim3(grid_for(ns, 256)), dim3(256), 0, st, lh, d_hist, S, keep_unheld, nint, nflt,
                       di, df, (long long *)d_cols, (float *)((long long *)d_cols + (size_t)nint * ns), flags, ctr); KCHK(ctx);
    unsigned long long h[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, ctr, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (h[0] >= 0xffffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "too many straddling pixels (%llu)", h[0]);
    const uint32_t cap = (uint32_t)h[0];
    // the straddlers' pixels, packed (the sorts of run_segstats are done with aux / aux2)
    CHK(buf_ensure(ctx, ctx->aux, (size_t)cap * 4 + 64));
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)cap * 8 + 64));
    uint32_t *d_count = (uint32_t *)(ctr + 4);
    HIPCHK(ctx, hipMemsetAsync(d_count, 0, 4, st));
    if (n && cap) {
        hipLaunchKernelGGL(k_gather_flagged, dim3(grid_for(n, 4096)), dim3(256), 0, st, d_seg, d_band, dtype, n, S, flags,
                           bp<uint32_t>(ctx->aux), (long long *)ctx->aux2.p, cap, d_count); KCHK(ctx);
        uint32_t got = 0;
        CHK(read_u32(ctx, d_count, &got));
        if (got != cap) SHP_FAIL(ctx, SHP_ERR_STATE, "straddler gather found %u pixels, the histogram says %u", got, cap);
    }
    *d_pair_seg = bp<uint32_t>(ctx->aux);
    *d_pair_val = (long long *)ctx->aux2.p;
    *n_pairs = (int64_t)cap;
    *n_strad = (int64_t)h[1];
    return 0;
}

// Merge part.  pairs: `world` slots of `slot` (id, value) pairs as the all-gather left them, counts[r] valid
// in slot r.  The pairs whose id lies in [id_lo, id_hi) -- this rank's share of the id space -- are compacted
// into a 1 x m raster, reduced by the same statistics code, and the rows of the ids that occur are written
// into cols (they were cleared by every rank's classify step).
__global__ __launch_bounds__(256) void k_dstats_pick(const uint32_t *__restrict__ pseg, const long long *__restrict__ pval,
                                                     uint32_t slot, uint32_t world, const uint32_t *__restrict__ counts,
                                                     uint32_t id_lo, uint32_t id_hi, int dtype,
                                                     uint32_t *__restrict__ out_seg, void *__restrict__ out_band,
                                                     uint32_t *count)
{
    const size_t q = (size_t)blockIdx.x * 256u + threadIdx.x;
    bool take = false;
    uint32_t sg = 0;
    long long v = 0;
    if (q < (size_t)slot * world) {
        const uint32_t r = (uint32_t)(q / slot), e = (uint32_t)(q - (size_t)r * slot);
        if (e < counts[r]) { sg = pseg[q]; v = pval[q]; take = sg >= id_lo && sg < id_hi; }
    }
    const unsigned long long m = __ballot(take);
    if (m == 0ull) return;
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (take) {
        const uint32_t o = base + (uint32_t)__popcll(m & lanemask_lt());
        out_seg[o] = sg;
        switch (dtype) {
        case SHP_U8: ((uint8_t *)out_band)[o] = (uint8_t)v; break;
        case SHP_I16: ((int16_t *)out_band)[o] = (int16_t)v; break;
        case SHP_U16: ((uint16_t *)out_band)[o] = (uint16_t)v; break;
        case SHP_I32: ((int32_t *)out_band)[o] = (int32_t)v; break;
        default: ((uint32_t *)out_band)[o] = (uint32_t)v; break;
        }
    }
}
__global__ __launch_bounds__(256) void k_dstats_take(const uint32_t *__restrict__ present, uint32_t S, int nint, int nflt,
                                                     const long long *__restrict__ src_int, const float *__restrict__ src_flt,
                                                     long long *__restrict__ dst_int, float *__restrict__ dst_flt,
                                                     uint32_t *n_ids)
{
    const size_t ns = (size_t)S + 1;
    const size_t id = (size_t)blockIdx.x * 256u + threadIdx.x;
    const bool here = id != 0 && id < ns && present[id] != 0u;
    const unsigned long long m = __ballot(here);
    if (m != 0ull && lane_id() == 0) atomicAdd(n_ids, (uint32_t)__popcll(m));
    if (!here) return;
    for (int c = 0; c < nint; c++) dst_int[(size_t)c * ns + id] = src_int[(size_t)c * ns + id];
    for (int c = 0; c < nflt; c++) dst_flt[(size_t)c * ns + id] = src_flt[(size_t)c * ns + id];
}

static int run_dstats_merge(shp_ctx *ctx, const uint32_t *d_pseg, const long long *d_pval, uint32_t slot, uint32_t world,
                            const uint32_t *counts_host, int dtype, uint32_t S, int has_null, int64_t null_val,
                            const uint32_t *sel_host, int nstats, int64_t missing, uint32_t id_lo, uint32_t id_hi,
                            void *d_cols, int64_t *n_merged, int64_t *n_ids)
{
    hipStream_t st = ctx->stream;
    const size_t ns = (size_t)S + 1;
    int nint = 0, nflt = 0;
    for (int i = 0; i < nstats; i++) { if (sel_host[i * 5 + 2] == 0) nint++; else nflt++; }
    *n_merged = 0;
    *n_ids = 0;
    const size_t total = (size_t)slot * world;
    if (total == 0 || id_lo >= id_hi) return 0;
    if (total >= 0xffffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "too many gathered pairs");
    CHK(buf_ensure(ctx, ctx->lab, total * 4 + 64));
    CHK(buf_ensure(ctx, ctx->img, total * dtype_size(dtype) + 64));
    CHK(buf_ensure(ctx, ctx->chtail, (size_t)world * 4 + 128));
    uint32_t *d_counts = bp<uint32_t>(ctx->chtail) + 16, *d_n = bp<uint32_t>(ctx->chtail);
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipMemcpyAsync(d_counts, counts_host, (size_t)world * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(d_n, 0, 4, st));
    hipLaunchKernelGGL(k_dstats_pick, dim3(grid_for(total, 256)), dim3(256), 0, st, d_pseg, d_pval, slot, world, d_counts,
                       id_lo, id_hi, dtype, bp<uint32_t>(ctx->lab), ctx->img.p, d_n); KCHK(ctx);
    uint32_t m = 0;
    CHK(read_u32(ctx, d_n, &m));
    *n_merged = (int64_t)m;
    if (m == 0) return 0;
    long long *di = nullptr;
    float *df = nullptr;
    CHK(run_segstats(ctx, bp<uint32_t>(ctx->lab), ctx->img.p, dtype, m, S, has_null, null_val, sel_host, nstats, missing,
                     nullptr, nullptr, 0, 0, &di, &df));
    CHK(buf_ensure(ctx, ctx->chnext, ns * 4 + 64));
    uint32_t *present = bp<uint32_t>(ctx->chnext);
    HIPCHK(ctx, hipMemsetAsync(present, 0, ns * 4, st));
    hipLaunchKernelGGL(k_label_hist, dim3(grid_for(m, 256)), dim3(256), 0, st, bp<uint32_t>(ctx->lab), m, S, present); KCHK(ctx);
    HIPCHK(ctx, hipMemsetAsync(d_n, 0, 4, st));
    hipLaunchKernelGGL(k_dstats_take, dim3(grid_for(ns, 256)), dim3(256), 0, st, present, S, nint, nflt, di, df,
                       (long long *)d_cols, (float *)((long long *)d_cols + (size_t)nint * ns), d_n); KCHK(ctx);
    uint32_t ids = 0;
    CHK(read_u32(ctx, d_n, &ids));
    *n_ids = (int64_t)ids;
    return 0;
}
