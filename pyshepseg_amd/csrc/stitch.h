// stitch.h -- cross-tile segment-id stitch for one tile, entirely on the device.
//
// Replaces tiling.recodeTile / recodeSharedSegments / crossesMidline / relabelSegments
// (tiling.py:1066-1306) and the per-tile part of stitchTiles (tiling.py:1029-1043).
//   * a local segment "crosses the midline" of the top (left) overlap strip when its pixels in
//     the strip span rows (cols) min < mid <= max            -> two integer atomics per pixel
//   * it is recoded to scipy.stats.mode of the neighbour tile's already-recoded labels under
//     its strip pixels (most frequent, smallest on ties, may be 0): exact (label, value) pair
//     counts in an open-addressing hash table (64-bit CAS, run-aggregated adds), then a
//     64-bit atomicMax of (count << 32 | ~value) per segment; top strip first, left overwrites
//   * every other segment whose bounding-box corner lies in the trimmed window gets the next
//     new id in ascending local-id order (flag + exclusive scan); the rest become 0
//   * the LUT is applied in place, the trimmed window is written to the output raster and
//     maxSegId advances to the largest id present in the trimmed window (device scalar).
// No host synchronisation: the whole tile is a chain of launches on the context's stream.
#pragma once
#include "common.h"
#include "scan.h"

__device__ __forceinline__ uint32_t hash64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

// Per segment of a strip (rows < srows, cols < scols of the tile): min and max+1 of the row index
// (horizontal) or of the column index (vertical); mx = 0 means absent.
// Only corner pixels can hold an extreme: the topmost pixel of a segment's leftmost column has
// neither the segment above it nor to its left, and so on for the other extremes.  That leaves a
// handful of candidates per segment; they are combined per 32 x 64 patch in LDS (AggTable) and
// flushed with one pruned global atomic per (patch, segment, field).
__device__ __forceinline__ void strip_minmax_body(const uint32_t *__restrict__ tile, uint32_t xs,
                                                  uint32_t srows, uint32_t scols, int horizontal,
                                                  uint32_t *mn, uint32_t *mx, uint32_t bx, uint32_t by)
{
    __shared__ AggTable tab;
    agg_init(tab, 0xFFFFFFFFu, 0u, 0u);
    __syncthreads();
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t c = bx * 64u + lane;
    const uint32_t r0 = by * AGG_ROWS + wv * (AGG_ROWS / 4u);
    const bool cin = c < scols;
    // the wavefront's rows r0 - 1 .. r0 + 8 and, in lanes 0 / 63, the pixels beside them: all loads in flight together
    constexpr uint32_t NR = AGG_ROWS / 4u;
    uint32_t rv[NR + 2u], ev[NR];
#pragma unroll
    for (uint32_t i = 0; i < NR + 2u; i++) {
        const uint32_t r = r0 + i - 1u;                      // (r0 == 0: wraps, and the test fails)
        rv[i] = (cin && r < srows) ? tile[(size_t)r * xs + c] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < NR; i++) {
        const uint32_t r = r0 + i;
        ev[i] = 0u;
        if (r < srows) {
            if (lane == 0 && cin && c > 0u) ev[i] = tile[(size_t)r * xs + c - 1u];
            if (lane == 63 && c + 1u < scols) ev[i] = tile[(size_t)r * xs + c + 1u];
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < NR; i++) {
        const uint32_t r = r0 + i;
        if (r >= srows) break;                               // uniform per wavefront
        const uint32_t above = rv[i], below = rv[i + 2u];
        const uint32_t s = rv[i + 1u];
        uint32_t lf = __shfl_up(s, 1, 64), rt = __shfl_down(s, 1, 64);
        if (lane == 0) lf = ev[i];
        if (lane == 63) rt = ev[i];
        if (cin && c + 1u == scols) rt = 0u;                 // the strip ends here
        if (s != 0u) {
            const bool ldiff = lf != s, udiff = above != s;
            bool lo, hi;
            uint32_t v;
            if (horizontal) { v = r; lo = ldiff && udiff; hi = ldiff && below != s; }
            else            { v = c; lo = ldiff && udiff; hi = udiff && rt != s; }
            if (lo || hi) {
                const int h = agg_slot(tab, s);
                if (h >= 0) {
                    if (lo) atomicMin(&tab.v[0][h], v);
                    if (hi) atomicMax(&tab.v[1][h], v + 1u);
                } else {
                    if (lo) atomicMin(&mn[s], v);
                    if (hi) atomicMax(&mx[s], v + 1u);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < AGG_SLOTS; i += 256u) {
        const uint32_t s = tab.key[i];
        if (s == 0u) continue;
        if (tab.v[0][i] < mn[s]) atomicMin(&mn[s], tab.v[0][i]);
        if (tab.v[1][i] > mx[s]) atomicMax(&mx[s], tab.v[1][i]);
    }
}
__global__ __launch_bounds__(256) void k_strip_minmax(const uint32_t *__restrict__ tile, uint32_t xs,
                                                      uint32_t srows, uint32_t scols, int horizontal,
                                                      uint32_t *mn, uint32_t *mx)
{
    strip_minmax_body(tile, xs, srows, scols, horizontal, mn, mx, blockIdx.x, blockIdx.y);
}
// both overlap strips of a tile in one launch: blockIdx.z = 0 the top strip (rows < g.rows0, horizontal),
// 1 the left strip (columns < g.cols1); each has its own (mn, mx) pair behind mnmx: [z][mn | mx][nseg]
struct StripPair { uint32_t rows0, cols0, rows1, cols1; };      // a strip that is absent has 0 rows
__global__ __launch_bounds__(256) void k_strip_minmax2(const uint32_t *__restrict__ tile, uint32_t xs,
                                                       StripPair g, uint32_t *mnmx, uint32_t nseg)
{
    const uint32_t z = blockIdx.z;
    const uint32_t srows = z ? g.rows1 : g.rows0, scols = z ? g.cols1 : g.cols0;
    if (blockIdx.x * 64u >= scols || blockIdx.y * AGG_ROWS >= srows) return;       // (uniform per workgroup)
    uint32_t *mn = mnmx + (size_t)z * 2u * nseg;
    strip_minmax_body(tile, xs, srows, scols, z == 0u, mn, mn + nseg, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void k_pair_count(
    const uint32_t *__restrict__ tile, uint32_t xs, uint32_t srows, uint32_t scols,
    const uint32_t *__restrict__ B, size_t bpitch, const uint32_t *__restrict__ mn,
    const uint32_t *__restrict__ mx, uint32_t mid, unsigned long long *keys, uint32_t *cnts,
    uint32_t hmask)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool inb = i < srows * scols;
    unsigned long long key = 0;
    if (inb) {
        const uint32_t r = i / scols, c = i - r * scols;
        const uint32_t s = tile[r * xs + c];
        if (s != 0 && mn[s] < mid && mx[s] >= mid + 1u)
            key = ((unsigned long long)s << 32) | (unsigned long long)B[(size_t)r * bpitch + c];
    }
    // aggregate runs of equal keys inside the wavefront
    const unsigned lane = lane_id();
    const unsigned long long pk = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || pk != key;
    const unsigned long long heads = __ballot(head);
    if (head && key != 0) {
        const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const uint32_t len = (nxt ? (unsigned)__builtin_ctzll(nxt) : 64u) - lane;
        uint32_t h = hash64(key) & hmask;
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[h], 0ull, key);
            if (old == 0ull || old == key) { atomicAdd(&cnts[h], len); break; }
            h = (h + 1u) & hmask;
        }
    }
}

__global__ __launch_bounds__(256) void k_pair_best(const unsigned long long *__restrict__ keys,
                                                   const uint32_t *__restrict__ cnts, uint32_t hsize,
                                                   unsigned long long *best)
{
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h >= hsize) return;
    const unsigned long long key = keys[h];
    if (key == 0ull) return;
    const uint32_t s = (uint32_t)(key >> 32), b = (uint32_t)key;
    atomicMax(&best[s], ((unsigned long long)cnts[h] << 32) | (unsigned long long)(~b));
}

__global__ __launch_bounds__(256) void k_best_to_dict(const unsigned long long *__restrict__ best,
                                                      uint32_t nseg, uint32_t *in_dict,
                                                      uint32_t *recode)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    const unsigned long long b = best[s];
    if (b != 0ull) { in_dict[s] = 1u; recode[s] = ~(uint32_t)b; }
}

__global__ __launch_bounds__(256) void k_seg_topleft(const uint32_t *__restrict__ tile, uint32_t ys,
                                                     uint32_t xs, uint32_t *segtop, uint32_t *segleft)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= ys * xs) return;
    const uint32_t s = tile[p];
    if (s == 0) return;
    const uint32_t r = p / xs, c = p - r * xs;
    if (!(c == 0 || tile[p - 1] != s) || !(r == 0 || tile[p - xs] != s)) return;   // corners only
    if (r < segtop[s]) atomicMin(&segtop[s], r);
    if (c < segleft[s]) atomicMin(&segleft[s], c);
}

struct OwnFn {
    const uint32_t *in_dict, *segtop, *segleft;
    uint32_t top, bottom, left, right;
    __device__ __forceinline__ uint32_t operator()(uint32_t s) const
    {
        if (s == 0 || in_dict[s]) return 0u;
        const uint32_t t = segtop[s], l = segleft[s];
        return (l >= left && t >= top && l < right && t < bottom) ? 1u : 0u;
    }
};

__global__ __launch_bounds__(256) void k_build_lut(OwnFn own, const uint32_t *__restrict__ rank,
                                                   const uint32_t *__restrict__ recode,
                                                   const uint32_t *__restrict__ max_seg_id,
                                                   uint32_t nseg, uint32_t *__restrict__ lut)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    uint32_t v = 0;
    if (s != 0) {
        if (own.in_dict[s]) v = recode[s];
        else if (own(s)) v = *max_seg_id + rank[s] + 1u;
    }
    lut[s] = v;
}

__global__ __launch_bounds__(256) void k_simple_recode(uint32_t *tile, uint32_t n,
                                                       const uint32_t *__restrict__ max_seg_id)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t s = tile[p];
    if (s != 0) tile[p] = s + *max_seg_id;          // tiling.py:1024-1027
}

// tile[p] = lut[tile[p]] (lut == nullptr: keep); trimmed window -> output raster; tmax = max id
__global__ __launch_bounds__(256) void k_apply_lut(uint32_t *tile, uint32_t ys, uint32_t xs,
                                                   const uint32_t *__restrict__ lut, uint32_t top,
                                                   uint32_t bottom, uint32_t left, uint32_t right,
                                                   uint32_t *__restrict__ out, size_t opitch,
                                                   uint32_t xout, uint32_t yout, uint32_t *tmax)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    uint32_t m = 0;
    if (p < ys * xs) {
        const uint32_t r = p / xs, c = p - r * xs;
        const uint32_t v = lut ? lut[tile[p]] : tile[p];
        if (lut) tile[p] = v;
        if (r >= top && r < bottom && c >= left && c < right) {
            out[(size_t)(yout + r - top) * opitch + (xout + c - left)] = v;
            m = v;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    // one atomic per wavefront only if it can still raise the maximum (plain pre-read: a stale
    // smaller value only costs a redundant atomic)
    if (lane_id() == 0 && m != 0 && m > *(volatile uint32_t *)tmax) atomicMax(tmax, m);
}

__global__ void k_max_merge(uint32_t *max_seg_id, const uint32_t *tmax)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && *tmax > *max_seg_id) *max_seg_id = *tmax;
}

static int stitch_strip(shp_ctx *ctx, const uint32_t *tile, uint32_t xs, uint32_t srows,
                        uint32_t scols, int horizontal, const uint32_t *B, size_t bpitch,
                        uint32_t nseg, uint32_t *mn, uint32_t *mx, unsigned long long *best,
                        unsigned long long *keys, uint32_t *cnts, uint32_t hsize, uint32_t *in_dict,
                        uint32_t *recode)
{
    hipStream_t st = ctx->stream;
    const uint32_t npx = srows * scols;
    if (npx == 0) return 0;
    const uint32_t mid = (horizontal ? srows : scols) / 2u;          // int(n / 2), tiling.py:1296
    HIPCHK(ctx, hipMemsetAsync(mn, 0xff, (size_t)nseg * 4, st));
    HIPCHK(ctx, hipMemsetAsync(mx, 0, (size_t)nseg * 4, st));
    HIPCHK(ctx, hipMemsetAsync(best, 0, (size_t)nseg * 8, st));
    HIPCHK(ctx, hipMemsetAsync(keys, 0, (size_t)hsize * 8, st));
    HIPCHK(ctx, hipMemsetAsync(cnts, 0, (size_t)hsize * 4, st));
    const unsigned g = grid_for(npx, 256);
    hipLaunchKernelGGL(k_strip_minmax, dim3(grid_for(scols, 64), grid_for(srows, AGG_ROWS)), dim3(256), 0, st, tile, xs, srows, scols, horizontal, mn, mx); KCHK(ctx);
    hipLaunchKernelGGL(k_pair_count, dim3(g), dim3(256), 0, st, tile, xs, srows, scols, B, bpitch, mn, mx,
                       mid, keys, cnts, hsize - 1u); KCHK(ctx);
    hipLaunchKernelGGL(k_pair_best, dim3(grid_for(hsize, 256)), dim3(256), 0, st, keys, cnts, hsize, best); KCHK(ctx);
    hipLaunchKernelGGL(k_best_to_dict, dim3(grid_for(nseg, 256)), dim3(256), 0, st, best, nseg, in_dict, recode); KCHK(ctx);
    return 0;
}

// One tile of stitchTiles.  d_tile: ys*xs local ids, recoded in place.
static int run_stitch_tile(shp_ctx *ctx, uint32_t *d_tile, uint32_t ys, uint32_t xs, uint32_t overlap,
                           const uint32_t *d_top_b, size_t top_pitch, const uint32_t *d_left_b,
                           size_t left_pitch, uint32_t max_local, int simple, uint32_t *d_max_seg_id,
                           uint32_t top, uint32_t bottom, uint32_t left, uint32_t right,
                           uint32_t *d_out, size_t out_pitch, uint32_t xout, uint32_t yout)
{
    hipStream_t st = ctx->stream;
    const uint32_t n = ys * xs;
    if (n == 0) return 0;
    const uint32_t nseg = max_local + 1u;
    const uint32_t an_rows = overlap < ys ? overlap : ys, an_cols = overlap < xs ? overlap : xs;
    uint32_t maxstrip = 0;
    if (d_top_b) maxstrip = an_rows * xs;
    if (d_left_b && ys * an_cols > maxstrip) maxstrip = ys * an_cols;
    uint32_t hsize = 1024;
    while (hsize < 2u * maxstrip) hsize <<= 1;
    // workspace carve-up (aux: per-segment tables, aux2: hash table)
    CHK(buf_ensure(ctx, ctx->aux, (size_t)nseg * 4 * 10 + 256));
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)hsize * 12 + 256));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(nseg)));
    CHK(buf_ensure(ctx, ctx->small, 4096));
    uint32_t *w = bp<uint32_t>(ctx->aux);
    uint32_t *in_dict = w, *recode = w + nseg, *mn = w + 2 * (size_t)nseg, *mx = w + 3 * (size_t)nseg;
    uint32_t *segtop = w + 4 * (size_t)nseg, *segleft = w + 5 * (size_t)nseg;
    uint32_t *rank = w + 6 * (size_t)nseg, *lut = w + 7 * (size_t)nseg;
    unsigned long long *best = (unsigned long long *)(w + 8 * (size_t)nseg);   // 32*nseg B: 8-aligned
    unsigned long long *keys = bp<unsigned long long>(ctx->aux2);
    uint32_t *cnts = (uint32_t *)(keys + hsize);
    uint32_t *tmax = bp<uint32_t>(ctx->small) + 32;
    HIPCHK(ctx, hipMemsetAsync(tmax, 0, 4, st));
    const unsigned g = grid_for(n, 256);
    if (simple) {
        hipLaunchKernelGGL(k_simple_recode, dim3(g), dim3(256), 0, st, d_tile, n, d_max_seg_id); KCHK(ctx);
        hipLaunchKernelGGL(k_apply_lut, dim3(g), dim3(256), 0, st, d_tile, ys, xs, (const uint32_t *)nullptr,
                           top, bottom, left, right, d_out, out_pitch, xout, yout, tmax); KCHK(ctx);
    } else {
        HIPCHK(ctx, hipMemsetAsync(in_dict, 0, (size_t)nseg * 4, st));
        if (d_top_b)
            CHK(stitch_strip(ctx, d_tile, xs, an_rows, xs, 1, d_top_b, top_pitch, nseg, mn, mx, best, keys,
                             cnts, hsize, in_dict, recode));
        if (d_left_b)
            CHK(stitch_strip(ctx, d_tile, xs, ys, an_cols, 0, d_left_b, left_pitch, nseg, mn, mx, best, keys,
                             cnts, hsize, in_dict, recode));
        HIPCHK(ctx, hipMemsetAsync(segtop, 0xff, (size_t)nseg * 8, st));      // segtop + segleft
        hipLaunchKernelGGL(k_seg_topleft, dim3(g), dim3(256), 0, st, d_tile, ys, xs, segtop, segleft); KCHK(ctx);
        OwnFn own{in_dict, segtop, segleft, top, bottom, left, right};
        CHK(scan_exclusive(ctx, own, nseg, rank, nullptr, bp<uint32_t>(ctx->scan_tmp)));
        hipLaunchKernelGGL(k_build_lut, dim3(grid_for(nseg, 256)), dim3(256), 0, st, own, rank, recode,
                           d_max_seg_id, nseg, lut); KCHK(ctx);
        hipLaunchKernelGGL(k_apply_lut, dim3(g), dim3(256), 0, st, d_tile, ys, xs, lut, top, bottom, left,
                           right, d_out, out_pitch, xout, yout, tmax); KCHK(ctx);
    }
    hipLaunchKernelGGL(k_max_merge, dim3(1), dim3(64), 0, st, d_max_seg_id, tmax); KCHK(ctx);
    return 0;
}

// =============================================================================================
// Three-phase stitch.  The reference's stitch is sequential in tile order, but most of a tile's
// work does not depend on its neighbours:
//   prepare (worker stream, right after the tile is segmented; purely local):
//       which local segments cross the midline of the top / left overlap strip, every segment's
//       bounding-box corner, and whether it has a pixel in the trimmed window
//   chain   (the one sequential stream; needs the recoded strips of the tiles above / left and
//       the running maxSegId): pair counts over the two strips -> modes, new-id ranks, the LUT,
//       maxSegId advance (a per-SEGMENT reduction: no pixel pass), and the recoded right /
//       bottom overlap strips that later tiles will read
//   finish  (side stream): LUT over the trimmed window -> output raster
// Per-tile meta block (uint32 arrays of max_local+1 entries): flags | segtop | segleft | lut.
// =============================================================================================
#define META_CROSS_TOP 1u
#define META_CROSS_LEFT 2u
#define META_IN_TRIM 4u

// every per-segment table of the prepare stage in one launch (memsets are launches too, and every
// launch boundary is a cache write-back / invalidate for the whole device)
__global__ __launch_bounds__(256) void k_meta_init(uint32_t nseg, uint32_t *__restrict__ flags,
                                                   uint32_t *__restrict__ segtop,
                                                   uint32_t *__restrict__ segleft,
                                                   uint32_t *__restrict__ mn, uint32_t *__restrict__ mx)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    flags[s] = 0u; segtop[s] = 0xFFFFFFFFu; segleft[s] = 0xFFFFFFFFu;
    if (mn) { mn[s] = 0xFFFFFFFFu; mx[s] = 0u; }
}
// the same with the (mn, mx) pairs of both strips: mnmx = [z][mn | mx][nseg]
__global__ __launch_bounds__(256) void k_meta_init2(uint32_t nseg, uint32_t *__restrict__ flags,
                                                    uint32_t *__restrict__ segtop,
                                                    uint32_t *__restrict__ segleft, uint32_t *__restrict__ mnmx)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    flags[s] = 0u; segtop[s] = 0xFFFFFFFFu; segleft[s] = 0xFFFFFFFFu;
    mnmx[s] = 0xFFFFFFFFu; mnmx[(size_t)nseg + s] = 0u;
    mnmx[2 * (size_t)nseg + s] = 0xFFFFFFFFu; mnmx[3 * (size_t)nseg + s] = 0u;
}

// crossesMidline for every segment of the strip; leaves mn / mx reset for the next strip
__global__ __launch_bounds__(256) void k_meta_cross(uint32_t *__restrict__ mn, uint32_t *__restrict__ mx,
                                                    uint32_t mid, uint32_t nseg, uint32_t bit,
                                                    uint32_t *flags)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    if (s != 0 && mn[s] < mid && mx[s] >= mid + 1u) flags[s] |= bit;
    mn[s] = 0xFFFFFFFFu; mx[s] = 0u;
}
// crossesMidline against both strips' midlines at once (mid0 / mid1; a strip that is absent has on = 0)
__global__ __launch_bounds__(256) void k_meta_cross2(const uint32_t *__restrict__ mnmx, uint32_t nseg,
                                                     uint32_t mid0, int on0, uint32_t mid1, int on1,
                                                     uint32_t *flags)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg || s == 0u) return;
    uint32_t f = 0u;
    if (on0 && mnmx[s] < mid0 && mnmx[(size_t)nseg + s] >= mid0 + 1u) f |= META_CROSS_TOP;
    if (on1 && mnmx[2 * (size_t)nseg + s] < mid1 && mnmx[3 * (size_t)nseg + s] >= mid1 + 1u) f |= META_CROSS_LEFT;
    if (f) flags[s] |= f;
}

// hist[id] += pixels of id (id 0 is not counted) over a raster of `ncols` columns: run lengths per
// 64-px row piece are combined per 32 x 64-px patch in an LDS hash table, so a segment costs one
// global atomic per patch it touches instead of one per run of pixels.
__global__ __launch_bounds__(256) void k_hist_patch(const uint32_t *__restrict__ ras, uint32_t nrows,
                                                    uint32_t ncols, uint32_t *hist)
{
    __shared__ AggTable tab;
    agg_init(tab, 0u, 0u, 0u);
    __syncthreads();
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * 64u + lane;
    const uint32_t r0 = blockIdx.y * AGG_ROWS + wv * (AGG_ROWS / 4u);
    const bool cin = c < ncols;
    uint32_t sv[AGG_ROWS / 4u];                 // the wavefront's eight rows, their loads in flight together
#pragma unroll
    for (uint32_t i = 0; i < AGG_ROWS / 4u; i++) sv[i] = (cin && r0 + i < nrows) ? ras[(size_t)(r0 + i) * ncols + c] : 0u;
#pragma unroll
    for (uint32_t i = 0; i < AGG_ROWS / 4u; i++) {
        const uint32_t r = r0 + i;
        if (r >= nrows) break;                               // uniform per wavefront
        const uint32_t s = sv[i];
        const uint32_t pv = __shfl_up(s, 1, 64);
        const bool head = lane == 0 || pv != s;
        const unsigned long long heads = __ballot(head);
        if (head && s != 0u) {
            const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
            const uint32_t len = (nxt ? (uint32_t)__builtin_ctzll(nxt) : 64u) - lane;
            const int h = agg_slot(tab, s);
            if (h >= 0) atomicAdd(&tab.v[0][h], len);
            else atomicAdd(&hist[s], len);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < AGG_SLOTS; i += 256u) {
        const uint32_t s = tab.key[i];
        if (s != 0u) atomicAdd(&hist[s], tab.v[0][i]);
    }
}


// bounding-box corner of every segment + "has a pixel in the trimmed window".
// Only corner pixels act (see k_strip_minmax): the top row and the left column of a segment are
// both found on pixels that have neither the segment above nor to the left, and the topmost-
// leftmost pixel of (segment n window) has the same property relative to the window.  Their
// candidates are combined per 32 x 64 patch in LDS (AggTable) before touching global memory.
__global__ __launch_bounds__(256) void k_meta_pixels(const uint32_t *__restrict__ tile, uint32_t ys,
                                                     uint32_t xs, uint32_t top, uint32_t bottom,
                                                     uint32_t left, uint32_t right, uint32_t *segtop,
                                                     uint32_t *segleft, uint32_t *flags)
{
    __shared__ AggTable tab;
    agg_init(tab, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u);
    __syncthreads();
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * 64u + lane;
    const uint32_t r0 = blockIdx.y * AGG_ROWS + wv * (AGG_ROWS / 4u);
    const bool cin = c < xs;
    uint32_t above = (cin && r0 > 0u && r0 <= ys) ? tile[(size_t)(r0 - 1u) * xs + c] : 0u;
    // the wavefront's eight rows (and, in lane 0, the pixels left of them): all loads in flight together
    uint32_t sv[AGG_ROWS / 4u], ev[AGG_ROWS / 4u];
#pragma unroll
    for (uint32_t i = 0; i < AGG_ROWS / 4u; i++) {
        const uint32_t r = r0 + i;
        sv[i] = (cin && r < ys) ? tile[(size_t)r * xs + c] : 0u;
        ev[i] = (lane == 0 && cin && c > 0u && r < ys) ? tile[(size_t)r * xs + c - 1u] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < AGG_ROWS / 4u; i++) {
        const uint32_t r = r0 + i;
        if (r >= ys) break;                                  // uniform per wavefront
        const uint32_t s = sv[i];
        uint32_t lf = __shfl_up(s, 1, 64);
        if (lane == 0) lf = ev[i];
        if (s != 0u) {
            const bool ldiff = c == 0u || lf != s;
            const bool udiff = r == 0u || above != s;
            const bool corner = ldiff && udiff;
            const bool inwin = r >= top && r < bottom && c >= left && c < right;
            const bool wcorner = inwin && (ldiff || c == left) && (udiff || r == top);
            if (corner || wcorner) {
                const int h = agg_slot(tab, s);
                if (h >= 0) {
                    if (corner) { atomicMin(&tab.v[0][h], r); atomicMin(&tab.v[1][h], c); }
                    if (wcorner) tab.v[2][h] = 1u;
                } else {
                    if (corner) { atomicMin(&segtop[s], r); atomicMin(&segleft[s], c); }
                    if (wcorner) atomicOr(&flags[s], META_IN_TRIM);
                }
            }
        }
        above = s;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < AGG_SLOTS; i += 256u) {
        const uint32_t s = tab.key[i];
        if (s == 0u) continue;
        // plain pre-reads prune most of what is left (a stale value only costs a redundant atomic)
        if (tab.v[0][i] < segtop[s]) atomicMin(&segtop[s], tab.v[0][i]);
        if (tab.v[1][i] < segleft[s]) atomicMin(&segleft[s], tab.v[1][i]);
        if (tab.v[2][i] && !(flags[s] & META_IN_TRIM)) atomicOr(&flags[s], META_IN_TRIM);
    }
}

// pixels of a strip whose segment crosses the strip's midline: an upper bound of the distinct
// (segment, neighbour id) pairs the chain step will hash, so that it can size (and clear) a table of
// that order instead of one sized by the whole strip
__global__ __launch_bounds__(256) void k_cross_count(const uint32_t *__restrict__ tile, uint32_t xs,
                                                     uint32_t srows, uint32_t scols,
                                                     const uint32_t *__restrict__ flags, uint32_t bit,
                                                     uint32_t *count)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0, sv[4];
#pragma unroll
    for (uint32_t e = 0; e < 4u; e++) {                      // the four labels first, then their flags: two round trips
        const uint32_t i = blockIdx.x * 1024u + e * 256u + threadIdx.x;
        sv[e] = 0u;
        if (i < srows * scols) {
            const uint32_t r = i / scols, c = i - r * scols;
            sv[e] = tile[r * xs + c];
        }
    }
#pragma unroll
    for (uint32_t e = 0; e < 4u; e++) mine += (sv[e] != 0u && (flags[sv[e]] & bit)) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane_id() == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(count, s_cnt);
}
// both strips in one launch: blockIdx.y = strip, count[0] the top strip's, count[1] the left strip's
__global__ __launch_bounds__(256) void k_cross_count2(const uint32_t *__restrict__ tile, uint32_t xs, StripPair g,
                                                      const uint32_t *__restrict__ flags, uint32_t *count)
{
    __shared__ uint32_t s_cnt;
    const uint32_t z = blockIdx.y;
    const uint32_t srows = z ? g.rows1 : g.rows0, scols = z ? g.cols1 : g.cols0;
    const uint32_t bit = z ? META_CROSS_LEFT : META_CROSS_TOP;
    if ((size_t)blockIdx.x * 1024u >= (size_t)srows * scols) return;             // (uniform per workgroup)
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x, e = 0; e < 4u; e++, i += 256u) {
        if (i < srows * scols) {
            const uint32_t r = i / scols, c = i - r * scols;
            const uint32_t s = tile[r * xs + c];
            mine += (s != 0u && (flags[s] & bit)) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane_id() == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(count + z, s_cnt);
}

static int run_stitch_prepare(shp_ctx *ctx, const uint32_t *d_tile, uint32_t ys, uint32_t xs,
                              uint32_t overlap, int has_top, int has_left, uint32_t max_local,
                              uint32_t top, uint32_t bottom, uint32_t left, uint32_t right,
                              uint32_t *d_meta, uint32_t *d_cross /* 2 counters, zeroed by the caller; may be NULL */)
{
    hipStream_t st = ctx->stream;
    const uint32_t n = ys * xs, nseg = max_local + 1u;
    uint32_t *flags = d_meta, *segtop = d_meta + nseg, *segleft = d_meta + 2 * (size_t)nseg;
    CHK(buf_ensure(ctx, ctx->aux, (size_t)nseg * 16 + 64));
    uint32_t *mnmx = bp<uint32_t>(ctx->aux);
    hipLaunchKernelGGL(k_meta_init2, dim3(grid_for(nseg, 256)), dim3(256), 0, st, nseg, flags, segtop, segleft,
                       mnmx); KCHK(ctx);
    if (n == 0) return 0;
    // both overlap strips per launch (three launches instead of six: a launch costs the fill phase 50-100 us
    // of queueing under load, whatever it does)
    const uint32_t an_rows = overlap < ys ? overlap : ys, an_cols = overlap < xs ? overlap : xs;
    StripPair g{0u, 0u, 0u, 0u};
    if (has_top && an_rows * xs != 0u) { g.rows0 = an_rows; g.cols0 = xs; }
    if (has_left && ys * an_cols != 0u) { g.rows1 = ys; g.cols1 = an_cols; }
    if (g.rows0 || g.rows1) {
        const uint32_t mc = g.cols0 > g.cols1 ? g.cols0 : g.cols1, mr = g.rows0 > g.rows1 ? g.rows0 : g.rows1;
        hipLaunchKernelGGL(k_strip_minmax2, dim3(grid_for(mc, 64), grid_for(mr, AGG_ROWS), 2), dim3(256), 0, st,
                           d_tile, xs, g, mnmx, nseg); KCHK(ctx);
        hipLaunchKernelGGL(k_meta_cross2, dim3(grid_for(nseg, 256)), dim3(256), 0, st, mnmx, nseg, g.rows0 / 2u,
                           g.rows0 != 0u, g.cols1 / 2u, g.rows1 != 0u, flags); KCHK(ctx);
        if (d_cross) {
            const size_t m0 = (size_t)g.rows0 * g.cols0, m1 = (size_t)g.rows1 * g.cols1;
            hipLaunchKernelGGL(k_cross_count2, dim3(grid_for(m0 > m1 ? m0 : m1, 1024), 2), dim3(256), 0, st, d_tile,
                               xs, g, flags, d_cross); KCHK(ctx);
        }
    }
    hipLaunchKernelGGL(k_meta_pixels, dim3(grid_for(xs, 64), grid_for(ys, AGG_ROWS)), dim3(256), 0, st, d_tile,
                       ys, xs, top, bottom, left, right, segtop, segleft, flags); KCHK(ctx);
    return 0;
}

__global__ __launch_bounds__(256) void k_pair_count_flag(
    const uint32_t *__restrict__ tile, uint32_t xs, uint32_t srows, uint32_t scols,
    const uint32_t *__restrict__ B, size_t bpitch, const uint32_t *__restrict__ flags, uint32_t bit,
    unsigned long long *keys, uint32_t *cnts, uint32_t hmask)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool inb = i < srows * scols;
    unsigned long long key = 0;
    if (inb) {
        const uint32_t r = i / scols, c = i - r * scols;
        const uint32_t s = tile[r * xs + c];
        if (s != 0 && (flags[s] & bit))
            key = ((unsigned long long)s << 32) | (unsigned long long)B[(size_t)r * bpitch + c];
    }
    const unsigned lane = lane_id();
    const unsigned long long pk = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || pk != key;
    const unsigned long long heads = __ballot(head);
    if (head && key != 0) {
        const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const uint32_t len = (nxt ? (unsigned)__builtin_ctzll(nxt) : 64u) - lane;
        uint32_t h = hash64(key) & hmask;
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[h], 0ull, key);
            if (old == 0ull || old == key) { atomicAdd(&cnts[h], len); break; }
            h = (h + 1u) & hmask;
        }
    }
}

__global__ __launch_bounds__(256) void k_lut_simple(uint32_t *__restrict__ lut, uint32_t nseg,
                                                    const uint32_t *__restrict__ max_seg_id)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    lut[s] = s ? s + *max_seg_id : 0u;
}

// largest recoded id among segments present in the trimmed window (== trimmed.max())
__global__ __launch_bounds__(256) void k_tmax_segments(const uint32_t *__restrict__ lut,
                                                       const uint32_t *__restrict__ flags,
                                                       uint32_t nseg, uint32_t *tmax)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    uint32_t m = (s < nseg && s != 0 && (flags[s] & META_IN_TRIM)) ? lut[s] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if (lane_id() == 0 && m != 0) atomicMax(tmax, m);
}

// out[r][c] = lut[tile[(r0 + r) * xs + c0 + c]] for an (nr x nc) sub-window -> dense strip
__global__ __launch_bounds__(256) void k_recode_window(const uint32_t *__restrict__ tile, uint32_t xs,
                                                       uint32_t r0, uint32_t c0, uint32_t nr,
                                                       uint32_t nc, const uint32_t *__restrict__ lut,
                                                       uint32_t *__restrict__ out, size_t opitch)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nr * nc) return;
    const uint32_t r = i / nc, c = i - r * nc;
    out[(size_t)r * opitch + c] = lut[tile[(r0 + r) * xs + c0 + c]];
}

// ---- the chain step's kernels: both strips per launch (blockIdx.y = 0 top strip, 1 left strip) ----
struct StripArgs {
    const uint32_t *B;              // the neighbour's recoded strip (nullptr: this tile has none)
    size_t pitch;
    uint32_t srows, scols, bit;
    unsigned long long *keys;       // hsize slots
    uint32_t *cnts;
    unsigned long long *best;       // nseg entries
};

__global__ __launch_bounds__(256) void k_pair_count2(const uint32_t *__restrict__ tile, uint32_t xs,
                                                     const uint32_t *__restrict__ flags,
                                                     const StripArgs a0, const StripArgs a1, uint32_t hmask)
{
    const StripArgs &a = blockIdx.y ? a1 : a0;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (!a.B || blockIdx.x * 256u >= a.srows * a.scols) return;         // (uniform per workgroup)
    const bool inb = i < a.srows * a.scols;
    unsigned long long key = 0;
    if (inb) {
        const uint32_t r = i / a.scols, c = i - r * a.scols;
        const uint32_t s = tile[r * xs + c];
        if (s != 0 && (flags[s] & a.bit))
            key = ((unsigned long long)s << 32) | (unsigned long long)a.B[(size_t)r * a.pitch + c];
    }
    const unsigned lane = lane_id();
    const unsigned long long pk = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || pk != key;
    const unsigned long long heads = __ballot(head);
    if (head && key != 0) {
        const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const uint32_t len = (nxt ? (unsigned)__builtin_ctzll(nxt) : 64u) - lane;
        uint32_t h = hash64(key) & hmask;
        for (;;) {
            const unsigned long long old = atomicCAS(&a.keys[h], 0ull, key);
            if (old == 0ull || old == key) { atomicAdd(&a.cnts[h], len); break; }
            h = (h + 1u) & hmask;
        }
    }
}

// best[seg] = max over the table of (count << 32 | ~value): most frequent, smallest on ties
__global__ __launch_bounds__(256) void k_pair_best2(const StripArgs a0, const StripArgs a1, uint32_t hsize)
{
    const StripArgs &a = blockIdx.y ? a1 : a0;
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (!a.B || h >= hsize) return;
    const unsigned long long k = a.keys[h];
    if (k == 0ull) return;
    const uint32_t s = (uint32_t)(k >> 32), v = (uint32_t)k;
    atomicMax(&a.best[s], ((unsigned long long)a.cnts[h] << 32) | (unsigned long long)(~v));
}

// a segment keeps its own (new) id when neither strip recodes it and its bounding-box corner lies in
// the trimmed window
struct OwnFn2 {
    const unsigned long long *best_top, *best_left;
    const uint32_t *segtop, *segleft;
    uint32_t top, bottom, left, right;
    __device__ __forceinline__ uint32_t operator()(uint32_t s) const
    {
        if (s == 0 || best_top[s] != 0ull || best_left[s] != 0ull) return 0u;
        const uint32_t t = segtop[s], l = segleft[s];
        return (l >= left && t >= top && l < right && t < bottom) ? 1u : 0u;
    }
};

// the LUT: the left strip's mode overrides the top strip's (recodeSharedSegments runs top first, then
// left, tiling.py:1129-1203), owned segments get max_seg_id + rank + 1, the rest 0
__global__ __launch_bounds__(256) void k_build_lut2(OwnFn2 own, const uint32_t *__restrict__ rank,
                                                    const uint32_t *__restrict__ rank_boff,
                                                    const uint32_t *__restrict__ max_seg_id,
                                                    uint32_t nseg, uint32_t *__restrict__ lut)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    uint32_t v = 0;
    if (s != 0) {
        const unsigned long long bl = own.best_left[s], bt = own.best_top[s];
        if (bl != 0ull) v = ~(uint32_t)bl;
        else if (bt != 0ull) v = ~(uint32_t)bt;
        else if (own(s)) v = *max_seg_id + rank[s] + (rank_boff ? rank_boff[s / SCAN_ITEMS] : 0u) + 1u;
    }
    lut[s] = v;
}

// maxSegId = max(maxSegId, largest recoded id among the segments present in the trimmed window)
__global__ __launch_bounds__(256) void k_tmax_into(const uint32_t *__restrict__ lut,
                                                   const uint32_t *__restrict__ flags, uint32_t nseg,
                                                   uint32_t *max_seg_id)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    uint32_t m = (s < nseg && s != 0 && (flags[s] & META_IN_TRIM)) ? lut[s] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if (lane_id() == 0 && m != 0) atomicMax(max_seg_id, m);
}

// the recoded right strip (ys x an_cols) and bottom strip (an_rows x xs) in one launch
__global__ __launch_bounds__(256) void k_recode_strips(const uint32_t *__restrict__ tile, uint32_t ys, uint32_t xs,
                                                       uint32_t an_rows, uint32_t an_cols,
                                                       const uint32_t *__restrict__ lut,
                                                       uint32_t *__restrict__ right_out,
                                                       uint32_t *__restrict__ bottom_out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.y == 0) {
        if (!right_out || i >= ys * an_cols) return;
        const uint32_t r = i / an_cols, c = i - r * an_cols;
        right_out[i] = lut[tile[r * xs + xs - an_cols + c]];
    } else {
        if (!bottom_out || i >= an_rows * xs) return;
        bottom_out[i] = lut[tile[(ys - an_rows) * xs + i]];
    }
}

static int run_stitch_chain(shp_ctx *ctx, const uint32_t *d_tile, uint32_t ys, uint32_t xs,
                            uint32_t overlap, const uint32_t *d_top_b, size_t top_pitch,
                            const uint32_t *d_left_b, size_t left_pitch, uint32_t max_local, int simple,
                            uint32_t *d_max_seg_id, uint32_t top, uint32_t bottom, uint32_t left,
                            uint32_t right, uint32_t *d_meta, uint32_t *d_right_out,
                            uint32_t *d_bottom_out, uint32_t top_cross_px = 0xFFFFFFFFu,
                            uint32_t left_cross_px = 0xFFFFFFFFu)
{
    hipStream_t st = ctx->stream;
    const uint32_t n = ys * xs;
    if (n == 0) return 0;
    const uint32_t nseg = max_local + 1u;
    uint32_t *flags = d_meta, *segtop = d_meta + nseg, *segleft = d_meta + 2 * (size_t)nseg;
    uint32_t *lut = d_meta + 3 * (size_t)nseg;
    const uint32_t an_rows = overlap < ys ? overlap : ys, an_cols = overlap < xs ? overlap : xs;
    if (simple) {
        hipLaunchKernelGGL(k_lut_simple, dim3(grid_for(nseg, 256)), dim3(256), 0, st, lut, nseg, d_max_seg_id); KCHK(ctx);
    } else {
        // Nine commands per tile (twenty-one before: under load every command of this one sequential
        // stream queues behind the other streams' kernels, and the chain fell 120 tiles behind):
        // one clear of every table, both strips per launch, the modes folded into the LUT build.
        // The pair table holds one entry per distinct (crossing segment, neighbour id): at most the
        // strip's pixels of crossing segments, which the prepare step counted (else the whole strip)
        uint32_t maxstrip = 0;
        if (d_top_b) maxstrip = top_cross_px < an_rows * xs ? top_cross_px : an_rows * xs;
        if (d_left_b) {
            const uint32_t bb = left_cross_px < ys * an_cols ? left_cross_px : ys * an_cols;
            if (bb > maxstrip) maxstrip = bb;
        }
        uint32_t hsize = 1024;
        while (hsize < 2u * maxstrip) hsize <<= 1;
        const size_t zbytes = (size_t)nseg * 16 + (size_t)hsize * 24;
        CHK(buf_ensure(ctx, ctx->aux, (size_t)nseg * 4 + 256));
        CHK(buf_ensure(ctx, ctx->aux2, zbytes + 256));
        CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(nseg)));
        uint32_t *rank = bp<uint32_t>(ctx->aux);
        unsigned long long *best_top = bp<unsigned long long>(ctx->aux2), *best_left = best_top + nseg;
        unsigned long long *keys0 = best_left + nseg, *keys1 = keys0 + hsize;
        uint32_t *cnts0 = (uint32_t *)(keys1 + hsize), *cnts1 = cnts0 + hsize;
        HIPCHK(ctx, hipMemsetAsync(ctx->aux2.p, 0, zbytes, st));
        const StripArgs a0{d_top_b, top_pitch, an_rows, xs, META_CROSS_TOP, keys0, cnts0, best_top};
        const StripArgs a1{d_left_b, left_pitch, ys, an_cols, META_CROSS_LEFT, keys1, cnts1, best_left};
        if (d_top_b || d_left_b) {
            const uint32_t n0 = d_top_b ? an_rows * xs : 0u, n1 = d_left_b ? ys * an_cols : 0u;
            hipLaunchKernelGGL(k_pair_count2, dim3(grid_for(n0 > n1 ? n0 : n1, 256), 2), dim3(256), 0, st, d_tile, xs,
                               flags, a0, a1, hsize - 1u); KCHK(ctx);
            hipLaunchKernelGGL(k_pair_best2, dim3(grid_for(hsize, 256), 2), dim3(256), 0, st, a0, a1, hsize); KCHK(ctx);
        }
        OwnFn2 own{best_top, best_left, segtop, segleft, top, bottom, left, right};
        const uint32_t *rank_boff = nullptr;
        CHK(scan_exclusive(ctx, own, nseg, rank, nullptr, bp<uint32_t>(ctx->scan_tmp), &rank_boff));
        hipLaunchKernelGGL(k_build_lut2, dim3(grid_for(nseg, 256)), dim3(256), 0, st, own, rank, rank_boff,
                           d_max_seg_id, nseg, lut); KCHK(ctx);
    }
    hipLaunchKernelGGL(k_tmax_into, dim3(grid_for(nseg, 256)), dim3(256), 0, st, lut, flags, nseg, d_max_seg_id); KCHK(ctx);
    // the recoded overlap strips that the tiles to the right / below will read
    uint32_t *ro = (d_right_out && an_cols) ? d_right_out : nullptr, *bo = (d_bottom_out && an_rows) ? d_bottom_out : nullptr;
    if (ro || bo) {
        const uint32_t n0 = ro ? ys * an_cols : 0u, n1 = bo ? an_rows * xs : 0u;
        hipLaunchKernelGGL(k_recode_strips, dim3(grid_for(n0 > n1 ? n0 : n1, 256), 2), dim3(256), 0, st, d_tile, ys, xs,
                           an_rows, an_cols, lut, ro, bo); KCHK(ctx);
    }
    return 0;
}

// ---- parallel (provisional-id) stitch of the sharded driver -----------------------------------------
// out[0] = largest new id the chain step handed out, minus base (= how many: they are base + 1 ..),
// out[1] = the same over the segments present in the trimmed window.  With base = the tile's
// provisional base every inherited id is below it.  out[] must be zero.
__global__ __launch_bounds__(256) void k_lut_counts(const uint32_t *__restrict__ lut,
                                                    const uint32_t *__restrict__ flags, uint32_t nseg,
                                                    uint32_t base, uint32_t *out)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const uint32_t l = (s < nseg && s != 0u) ? lut[s] : 0u;
    uint32_t a = l > base ? l - base : 0u;
    uint32_t b = (a != 0u && (flags[s] & META_IN_TRIM)) ? a : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t oa = __shfl_xor(a, d, 64), ob = __shfl_xor(b, d, 64);
        a = oa > a ? oa : a;
        b = ob > b ? ob : b;
    }
    if (lane_id() == 0) {
        if (a) atomicMax(&out[0], a);
        if (b) atomicMax(&out[1], b);
    }
}

// id -> table[id / stride] + id % stride (0 stays 0): provisional ids (tile index * stride + rank)
// to the final ones once every tile's count is known
__global__ __launch_bounds__(256) void k_renumber(uint32_t *__restrict__ ras, size_t n, uint32_t stride,
                                                  const uint32_t *__restrict__ table, uint32_t ntiles)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = ras[i];
    if (v == 0u) return;
    const uint32_t t = v / stride;
    ras[i] = t < ntiles ? table[t] + (v - t * stride) : 0u;
}

// trimmed window of the tile through the LUT into the output raster, on the side stream
static int run_stitch_finish(shp_ctx *ctx, const uint32_t *d_tile, uint32_t ys, uint32_t xs,
                             uint32_t max_local, uint32_t top, uint32_t bottom, uint32_t left,
                             uint32_t right, const uint32_t *d_meta, uint32_t *d_out, size_t out_pitch,
                             uint32_t xout, uint32_t yout)
{
    if (bottom <= top || right <= left) return 0;
    const uint32_t nseg = max_local + 1u;
    const uint32_t *lut = d_meta + 3 * (size_t)nseg;
    CHK(ensure_stream2(ctx));
    HIPCHK(ctx, hipEventRecord(ctx->evfork, ctx->stream));          // after this tile's chain step
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->evfork, 0));
    const uint32_t nr = bottom - top, nc = right - left;
    hipLaunchKernelGGL(k_recode_window, dim3(grid_for(nr * nc, 256)), dim3(256), 0, ctx->stream2, d_tile, xs,
                       top, left, nr, nc, lut, d_out + (size_t)yout * out_pitch + xout, out_pitch); KCHK(ctx);
    (void)ys;
    return 0;
}

// ---- overview layers (tiling.py:1360-1383 writeOverviews) ------------------------------------------
// The reference sub-samples every stitched, trimmed tile by taking every lvl-th pixel from offset
// lvl / 2 of THE TILE and writes the block at (xout / lvl, yout / lvl) of the overview band, clipped
// to the band: ov[yout / lvl + r][xout / lvl + c] = raster[yout + lvl / 2 + r * lvl][xout + lvl / 2 + c * lvl].
__global__ __launch_bounds__(256) void k_overview_window(const uint32_t *__restrict__ ras, size_t pitch,
                                                         uint32_t xout, uint32_t yout, uint32_t w, uint32_t h,
                                                         uint32_t lvl, uint32_t *__restrict__ ov, uint32_t ovw,
                                                         uint32_t ovh)
{
    const uint32_t o = lvl / 2u;
    const uint32_t nsr = h > o ? (h - o + lvl - 1u) / lvl : 0u, nsc = w > o ? (w - o + lvl - 1u) / lvl : 0u;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nsr * nsc) return;
    const uint32_t r = i / nsc, c = i - r * nsc;
    const uint32_t dr = yout / lvl + r, dc = xout / lvl + c;
    if (dr >= ovh || dc >= ovw) return;
    ov[(size_t)dr * ovw + dc] = ras[(size_t)(yout + o + r * lvl) * pitch + xout + o + c * lvl];
}
