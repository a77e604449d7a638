// clump.h -- connected-component labelling with the reference's size-capped depth-first cut.
//
// Replaces shepseg.clump (shepseg.py:452-541).  The reference is a raster scan that grows each
// clump with an explicit LIFO stack and stops growing after MAX_CLUMP_SIZE pixels were added
// (SURVEY N9), so a true component of > 10001 pixels is cut into pieces whose membership
// depends on the visit order.  Design:
//   1. union-find CCL on the cluster image (labels = smallest linear index of the component);
//      run heads inside a wavefront are resolved with ballots, vertical links are pruned when
//      the 2x2 neighbourhood already implies them, the rest goes through atomicMin hooks.
//   2. component sizes by run-aggregated atomics (one atomic per run per wavefront).
//   3. every component that can be cut (size >= MAX_CLUMP_SIZE + 2) is replayed exactly by ONE
//      wavefront: lane 0 runs the depth-first walk with the reference's push order, all 64
//      lanes search the bounding box for the next unvisited pixel in raster order (next seed).
//      Components are independent, so the chip runs hundreds of replays concurrently.
//   4. a piece's id is the raster rank of its seed pixel: flag seeds, exclusive scan, gather.
// All of it is integer work bound by HBM/L2 latency, not by bandwidth; algorithmic bytes are
// 2 B (cluster id) in + 4 B (label) out per pixel.
#pragma once
#include "common.h"
#include "scan.h"
#include <stdlib.h>

struct BigInfo {
    uint32_t root, size, off, minc, maxc, maxr;
};

__device__ __forceinline__ uint32_t uf_find(const uint32_t *lab, uint32_t x)
{
    uint32_t p = lab[x];
    while (p != x) { x = p; p = lab[x]; }
    return x;
}

__device__ __forceinline__ void uf_merge(uint32_t *lab, uint32_t a, uint32_t b)
{
    a = uf_find(lab, a);
    b = uf_find(lab, b);
    while (a != b) {
        if (a < b) { uint32_t t = a; a = b; b = t; }       // a > b: hook a under b
        const uint32_t old = atomicMin(&lab[a], b);
        if (old == a) break;                                // a was a root: done
        a = old;                                            // a was hooked meanwhile: merge old with b
        a = uf_find(lab, a);
        b = uf_find(lab, b);
    }
}

// ---- union-find CCL, patch-local first ------------------------------------------------------------
// The image is cut into 32 x 64 patches (one 256-thread workgroup each, a wavefront per image row
// so loads stay coalesced).  k_ccl_local labels a patch entirely in LDS (run heads by ballot, the
// pruned vertical / diagonal links as LDS atomicMin hooks), counts the size of every patch-local
// component in LDS and writes, per pixel, lab = global index of its local root and csize = that
// local size at local roots, 0 elsewhere.  k_ccl_border then hooks the links that cross a patch
// border with global atomics (~10 % of the pixels), and k_ccl_flatten resolves every pixel to its
// global root and moves the size of each local component that is not its own global root to that
// root -- one global atomic per such component instead of one per run of pixels, none at all for
// the majority of components that live inside one patch.
// Link rule for pixel p of cluster c (L, U, UL, UR = that neighbour exists and has cluster c):
//   horizontal runs are linked to their head; U is linked unless L and UL (then p-1's U link and
//   the row runs already imply it); 8-connectivity adds, only when U is absent, UL unless L, and UR.
#define CCL_ROWS 32u
__device__ __forceinline__ uint32_t lds_find(const uint32_t *L, uint32_t x)
{
    uint32_t p = L[x];
    while (p != x) { x = p; p = L[x]; }
    return x;
}
__device__ __forceinline__ void lds_merge(uint32_t *L, uint32_t a, uint32_t b)
{
    a = lds_find(L, a);
    b = lds_find(L, b);
    while (a != b) {
        if (a < b) { uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(&L[a], b);
        if (old == a) break;
        a = lds_find(L, old);
        b = lds_find(L, b);
    }
}

__global__ __launch_bounds__(256) void k_ccl_local(const uint16_t *__restrict__ clus,
                                                   uint32_t *__restrict__ lab,
                                                   uint32_t *__restrict__ csize, uint32_t nrows,
                                                   uint32_t ncols, int four, uint32_t *zero4,
                                                   uint32_t *zero_a, uint32_t *zero_b, uint32_t cpitch)
{
    // the scalars of the later clump kernels are zeroed here instead of by memset launches
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8u) {
        zero4[threadIdx.x] = 0u;          // (eight words: counters[0..4] of run_clump and spares)
        if (threadIdx.x == 0) { if (zero_a) *zero_a = 0u; if (zero_b) { zero_b[0] = 0u; zero_b[1] = 0u; } }
    }
    __shared__ uint32_t L[CCL_ROWS * 64u];
    __shared__ uint32_t sz[CCL_ROWS * 64u];
    __shared__ uint16_t cl[CCL_ROWS * 64u];
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * 64u + lane;
    const uint32_t prow0 = blockIdx.y * CCL_ROWS;
    const bool cin = c < ncols;
    uint32_t cv[CCL_ROWS / 4u];
#pragma unroll
    for (uint32_t i = 0; i < CCL_ROWS / 4u; i++) {
        const uint32_t lr = wv * (CCL_ROWS / 4u) + i, r = prow0 + lr;
        const uint32_t v = (cin && r < nrows) ? clus[(size_t)r * cpitch + c] : 0u;
        cv[i] = v;
        cl[lr * 64u + lane] = (uint16_t)v;
        sz[lr * 64u + lane] = 0u;
        // row-run heads by ballot
        const uint32_t vl = __shfl_up(v, 1, 64);
        const bool leftsame = lane > 0 && v != 0u && vl == v;
        const unsigned long long heads = __ballot(!leftsame);
        const unsigned long long m = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        const unsigned start = 63u - (unsigned)__clzll(m);
        L[lr * 64u + lane] = lr * 64u + start;          // null pixels: their own (unused) singleton
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < CCL_ROWS / 4u; i++) {
        const uint32_t lr = wv * (CCL_ROWS / 4u) + i;
        const uint32_t v = cv[i];
        if (v == 0u || lr == 0u) continue;
        const uint32_t x = lr * 64u + lane;
        const bool Lk = lane > 0 && cl[x - 1u] == v;
        const bool U = cl[x - 64u] == v;
        const bool UL = lane > 0 && cl[x - 65u] == v;
        if (U) {
            if (!(Lk && UL)) lds_merge(L, x, x - 64u);
        } else if (!four) {
            if (UL && !Lk) lds_merge(L, x, x - 65u);
            if (lane < 63 && cl[x - 63u] == v) lds_merge(L, x, x - 63u);
        }
    }
    __syncthreads();
    uint32_t root[CCL_ROWS / 4u];
#pragma unroll
    for (uint32_t i = 0; i < CCL_ROWS / 4u; i++) {
        const uint32_t x = (wv * (CCL_ROWS / 4u) + i) * 64u + lane;
        root[i] = cv[i] ? lds_find(L, x) : 0xFFFFFFFFu;
        // one LDS atomic per run of equal roots in the row
        const uint32_t pr = __shfl_up(root[i], 1, 64);
        const bool head = lane == 0 || pr != root[i];
        const unsigned long long heads = __ballot(head);
        if (head && cv[i]) {
            const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
            atomicAdd(&sz[root[i]], (nxt ? (unsigned)__builtin_ctzll(nxt) : 64u) - lane);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < CCL_ROWS / 4u; i++) {
        const uint32_t lr = wv * (CCL_ROWS / 4u) + i, r = prow0 + lr;
        if (!cin || r >= nrows) continue;
        const size_t p = (size_t)r * ncols + c;
        if (cv[i] == 0u) { lab[p] = NULL_LAB; csize[p] = 0u; continue; }
        const uint32_t rt = root[i];
        lab[p] = (prow0 + (rt >> 6)) * ncols + blockIdx.x * 64u + (rt & 63u);
        csize[p] = (rt == lr * 64u + lane) ? sz[rt] : 0u;
    }
}

// links that cross a patch border (same rule as inside the patches).  Only border pixels are
// launched: first every pixel of the patches' top rows, then the first / last column of every
// patch for the remaining rows.
__global__ __launch_bounds__(256) void k_ccl_border(const uint16_t *__restrict__ clus, uint32_t *lab,
                                                    uint32_t nrows, uint32_t ncols, int four,
                                                    uint32_t ntop, uint32_t npc, uint32_t cpitch)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t row, col;
    if (i < ntop * ncols) {
        row = (i / ncols) * CCL_ROWS;
        col = i % ncols;
    } else {
        const uint32_t j = i - ntop * ncols;
        row = j / (2u * npc);
        const uint32_t k = j - row * (2u * npc);
        col = (k >> 1) * 64u + ((k & 1u) ? 63u : 0u);
        if (row >= nrows || col >= ncols || (row % CCL_ROWS) == 0u) return;
    }
    const uint32_t p = row * ncols + col;
    const bool topb = (row % CCL_ROWS) == 0u, leftb = (col & 63u) == 0u, rightb = (col & 63u) == 63u;
    const size_t cp = (size_t)row * cpitch + col;          // (the cluster ids may be a window of a wider map)
    const uint32_t v = clus[cp];
    if (v == 0u) return;
    const bool L = col > 0 && clus[cp - 1] == v;
    const bool U = row > 0 && clus[cp - cpitch] == v;
    const bool UL = row > 0 && col > 0 && clus[cp - cpitch - 1] == v;
    if (L && leftb) uf_merge(lab, p, p - 1);                       // row runs cut at a patch column
    if (U) {
        if (topb && !(L && UL)) uf_merge(lab, p, p - ncols);
    } else if (!four) {
        if (UL && !L && (topb || leftb)) uf_merge(lab, p, p - ncols - 1);
        const bool UR = row > 0 && col + 1 < ncols && clus[cp - cpitch + 1] == v;
        if (UR && (topb || rightb)) uf_merge(lab, p, p - ncols + 1);
    }
}

// lab[p] = global root; a patch-local component that is not its own global root hands its size
// (csize at its local root, 0 at every other pixel) to that root
// (four CONSECUTIVE pixels per thread: labels and sizes come and go as 16-byte accesses, and the four chains of
//  parent links are followed side by side -- one pixel per thread had one dependent 4-byte load in flight per lane)
#define FLAT_PPT 4u
__global__ __launch_bounds__(256) void k_ccl_flatten(uint32_t *lab, uint32_t n, uint32_t *csize,
                                                     uint32_t *__restrict__ bigbits)
{
    const uint32_t p0 = (blockIdx.x * 256u + threadIdx.x) * FLAT_PPT;
    if (p0 >= n) return;
    if ((p0 & 31u) == 0u) bigbits[p0 >> 5] = 0u;              // the cut-able-root bitmap starts empty
    uint32_t x[FLAT_PPT], px[FLAT_PPT], sz[FLAT_PPT];
    const bool vec = p0 + FLAT_PPT <= n && (((uintptr_t)lab | (uintptr_t)csize) & 15u) == 0u;
    if (vec) {
        const uint4 l = *(const uint4 *)(lab + p0), c = *(const uint4 *)(csize + p0);
        px[0] = l.x; px[1] = l.y; px[2] = l.z; px[3] = l.w;
        sz[0] = c.x; sz[1] = c.y; sz[2] = c.z; sz[3] = c.w;
    } else {
#pragma unroll
        for (uint32_t q = 0; q < FLAT_PPT; q++) {
            px[q] = p0 + q < n ? lab[p0 + q] : NULL_LAB;
            sz[q] = p0 + q < n ? csize[p0 + q] : 0u;
        }
    }
    bool on[FLAT_PPT];
#pragma unroll
    for (uint32_t q = 0; q < FLAT_PPT; q++) { x[q] = p0 + q; on[q] = px[q] != NULL_LAB; }
    for (;;) {
        bool any = false;
#pragma unroll
        for (uint32_t q = 0; q < FLAT_PPT; q++) {
            if (on[q] && px[q] != x[q]) { x[q] = px[q]; px[q] = lab[x[q]]; any = true; }
        }
        if (!any) break;
    }
    if (vec) {
        *(uint4 *)(lab + p0) = make_uint4(on[0] ? x[0] : NULL_LAB, on[1] ? x[1] : NULL_LAB, on[2] ? x[2] : NULL_LAB,
                                          on[3] ? x[3] : NULL_LAB);
    } else {
#pragma unroll
        for (uint32_t q = 0; q < FLAT_PPT; q++)
            if (on[q] && p0 + q < n) lab[p0 + q] = x[q];
    }
#pragma unroll
    for (uint32_t q = 0; q < FLAT_PPT; q++)
        if (on[q] && x[q] != p0 + q && sz[q]) atomicAdd(&csize[x[q]], sz[q]);
}

// cnt[key[p]] += 1 for every pixel, one atomic per run of equal keys per wavefront.
// Pixels whose key == skip are not counted.
__global__ __launch_bounds__(256) void k_run_count(const uint32_t *__restrict__ key, uint32_t n,
                                                   uint32_t *cnt, uint32_t skip, int use_skip)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool inb = p < n;
    const unsigned lane = lane_id();
    const uint32_t v = inb ? key[p] : 0u;
    const uint32_t pv = __shfl_up(v, 1, 64);
    const bool head = lane == 0 || pv != v || !inb;
    const unsigned long long heads = __ballot(head);
    if (head && inb && !(use_skip && v == skip)) {
        const unsigned long long nxt = (lane == 63) ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const unsigned nl = nxt ? (unsigned)__builtin_ctzll(nxt) : 64u;
        atomicAdd(&cnt[v], nl - lane);
    }
}

__global__ __launch_bounds__(256) void k_big_list(const uint32_t *__restrict__ lab,
                                                  uint32_t *csize, uint32_t n, uint32_t ncols,
                                                  BigInfo *big, uint32_t *counters, uint32_t *bigbits,
                                                  uint32_t bigmin)
{
    // (four consecutive pixels per thread, one 16-byte load: the kernel is a stream over the labels that stops at roots)
    const uint32_t p0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (p0 >= n) return;
    uint32_t l[4];
    if (p0 + 4u <= n && ((uintptr_t)lab & 15u) == 0u) {
        const uint4 v = *(const uint4 *)(lab + p0);
        l[0] = v.x; l[1] = v.y; l[2] = v.z; l[3] = v.w;
    } else {
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) l[q] = p0 + q < n ? lab[p0 + q] : NULL_LAB;
    }
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) {
        const uint32_t p = p0 + q;
        if (l[q] != p) continue;                      // (NULL_LAB is no pixel index)
        const uint32_t s = csize[p];
        if (s < bigmin) continue;
        const uint32_t bi = atomicAdd(&counters[0], 1u);
        const uint32_t off = atomicAdd(&counters[1], s);
        BigInfo b;
        b.root = p; b.size = s; b.off = off; b.minc = ncols; b.maxc = 0; b.maxr = 0;
        big[bi] = b;
        csize[p] = VIS_FLAG | bi;
        atomicOr(&bigbits[p >> 5], 1u << (p & 31u));      // N-bit map of cut-able roots (L2-resident)
    }
}

// Bounding box (min col, max col, max row; the min row is the root's) of every cut-able
// component.  Only corner pixels can hold an extreme (the topmost pixel of the leftmost column
// has no member above or to its left, ...); their candidates are combined per 32 x 64 patch in
// LDS (AggTable, keyed by root) and flushed with one pruned global atomic per (patch, component,
// field): atomics on one address serialise at L2.
__global__ __launch_bounds__(256) void k_big_bbox(const uint32_t *__restrict__ lab,
                                                  const uint32_t *__restrict__ csize, uint32_t nrows,
                                                  uint32_t ncols, BigInfo *big,
                                                  const uint32_t *__restrict__ bigbits)
{
    __shared__ AggTable tab;
    agg_init(tab, 0xFFFFFFFFu, 0u, 0u);             // min col | max col + 1 | max row + 1
    __syncthreads();
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * 64u + lane;
    const uint32_t r0 = blockIdx.y * AGG_ROWS + wv * (AGG_ROWS / 4u);
    const bool cin = c < ncols;
    uint32_t above = (cin && r0 > 0u && r0 <= nrows) ? lab[(size_t)(r0 - 1u) * ncols + c] : NULL_LAB;
    uint32_t cur = (cin && r0 < nrows) ? lab[(size_t)r0 * ncols + c] : NULL_LAB;
    for (uint32_t i = 0; i < AGG_ROWS / 4u; i++) {
        const uint32_t row = r0 + i;
        if (row >= nrows) break;                             // uniform per wavefront
        const uint32_t below = (cin && row + 1u < nrows) ? lab[(size_t)(row + 1u) * ncols + c] : NULL_LAB;
        const uint32_t r = cur;
        uint32_t lf = __shfl_up(r, 1, 64), rt = __shfl_down(r, 1, 64);
        if (lane == 0) lf = (cin && c > 0u) ? lab[(size_t)row * ncols + c - 1u] : NULL_LAB;
        if (lane == 63) rt = (c + 1u < ncols) ? lab[(size_t)row * ncols + c + 1u] : NULL_LAB;
        if (cin && c + 1u == ncols) rt = NULL_LAB;
        if (r != NULL_LAB && ((bigbits[r >> 5] >> (r & 31u)) & 1u)) {    // 2 MB bitmap, not a 4N-byte gather
            const bool ldiff = lf != r, rdiff = rt != r, udiff = above != r, ddiff = below != r;
            const bool cminc = ldiff && udiff, cmaxc = rdiff && udiff, cmaxr = ddiff && ldiff;
            if (cminc || cmaxc || cmaxr) {
                const int h = agg_slot(tab, r + 1u);                     // key 0 means empty
                if (h >= 0) {
                    if (cminc) atomicMin(&tab.v[0][h], c);
                    if (cmaxc) atomicMax(&tab.v[1][h], c + 1u);
                    if (cmaxr) atomicMax(&tab.v[2][h], row + 1u);
                } else {
                    const uint32_t cs = csize[r];
                    if (cs & VIS_FLAG) {
                        const uint32_t bi = cs & ~VIS_FLAG;
                        if (cminc) atomicMin(&big[bi].minc, c);
                        if (cmaxc) atomicMax(&big[bi].maxc, c);
                        if (cmaxr) atomicMax(&big[bi].maxr, row);
                    }
                }
            }
        }
        above = cur;
        cur = below;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < AGG_SLOTS; i += 256u) {
        if (tab.key[i] == 0u) continue;
        const uint32_t cs = csize[tab.key[i] - 1u];
        if (!(cs & VIS_FLAG)) continue;
        const uint32_t bi = cs & ~VIS_FLAG;
        const uint32_t mn = tab.v[0][i], mx = tab.v[1][i], mr = tab.v[2][i];
        if (mn != 0xFFFFFFFFu && mn < big[bi].minc) atomicMin(&big[bi].minc, mn);
        if (mx != 0u && mx - 1u > big[bi].maxc) atomicMax(&big[bi].maxc, mx - 1u);
        if (mr != 0u && mr - 1u > big[bi].maxr) atomicMax(&big[bi].maxr, mr - 1u);
    }
}

// One wavefront per cut-able component: exact replay of shepseg.py:490-539 restricted to the
// component (unvisited member pixels are exactly those with lab == root).
//   * the explicit LIFO stack lives in an LDS window (oldest half spilled to / refilled from
//     a per-component slice of a global scratch array), entries are packed (row << 16 | col)
//     so that no integer division sits on the dependent path;
//   * lanes 0..3 (0..7 for 8-connectivity) each own one neighbour in the reference's push
//     order (cx outer, cy inner): one ballot decides which are unvisited members, they are
//     labelled together and pushed in lane order, the last one becoming the next pop;
//   * all 64 lanes scan the bounding box for the next seed in raster order.
#define DFS_SWN 512u         // stack window entries in LDS per walker (2 KiB)
__device__ __forceinline__ unsigned long long dfs_bitmap_words(const BigInfo &B, uint32_t ncols);

__device__ __forceinline__ void dfs_split_global(uint32_t *lab, const BigInfo &B, uint32_t *sw,
                                                 uint32_t *stackbuf, uint32_t nrows, uint32_t ncols,
                                                 int four, uint32_t *singles, uint32_t *nsingles,
                                                 uint32_t *csize)
{
    const uint32_t root = B.root;
    uint32_t *gstack = stackbuf + B.off;
    const unsigned lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    // this lane's neighbour offset in push order
    int dy = 0, dx = 0;
    const unsigned nq = four ? 4u : 8u;
    if (four) {
        dy = (lane == 1) ? -1 : (lane == 2) ? 1 : 0;
        dx = (lane == 0) ? -1 : (lane == 3) ? 1 : 0;
    } else {
        const int t8y[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
        const int t8x[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        dy = t8y[lane & 7u];
        dx = t8x[lane & 7u];
    }
    uint32_t cursor = root;
    for (;;) {
        // ---- next seed: first unvisited member at or after cursor, raster order ----
        uint32_t seed = NULL_LAB;
        const uint32_t crow = cursor / ncols, ccol = cursor - crow * ncols;
        for (uint32_t row = crow; row <= B.maxr && seed == NULL_LAB; row++) {
            uint32_t c0 = B.minc;
            if (row == crow && ccol > c0) c0 = ccol;
            for (; c0 <= B.maxc; c0 += 64u) {
                const uint32_t c = c0 + lane;
                const bool ok = c <= B.maxc && lab[row * ncols + c] == root;
                const unsigned long long m = __ballot(ok);
                if (m) { seed = row * ncols + c0 + (uint32_t)__builtin_ctzll(m); break; }
            }
        }
        if (seed == NULL_LAB) break;
        const uint32_t FL = seed | VIS_FLAG;
        if (lane == 0) lab[seed] = FL;
        uint32_t sp_l = 0, sp_g = 0;            // entries in the LDS window / spilled to global
        uint32_t ty = seed / ncols, tx = seed - ty * ncols;    // current pop (uniform)
        bool have = true;
        uint32_t cnt = 0;
        while (have && cnt < MAX_CLUMP_SIZE) {
            const int ny = (int)ty + dy, nx = (int)tx + dx;
            const bool valid = lane < nq && ny >= 0 && nx >= 0 && ny < (int)nrows && nx < (int)ncols;
            const uint32_t q = (uint32_t)ny * ncols + (uint32_t)nx;
            const bool avail = valid && lab[q] == root;
            const unsigned long long m = __ballot(avail);
            if (avail) lab[q] = FL;
            const uint32_t npush = (uint32_t)__popcll(m);
            if (npush == 0) {
                if (sp_l == 0 && sp_g > 0) {            // refill the window from the spill area
                    const uint32_t k = sp_g < DFS_SWN / 2u ? sp_g : DFS_SWN / 2u;
                    for (uint32_t i = lane; i < k; i += 64u) sw[i] = gstack[sp_g - k + i];
                    sp_g -= k;
                    sp_l = k;
                    __builtin_amdgcn_wave_barrier();
                }
                if (sp_l > 0) {
                    const uint32_t e = sw[--sp_l];
                    ty = e >> 16; tx = e & 0xffffu;
                } else {
                    have = false;
                }
            } else {
                if (sp_l + 8u > DFS_SWN) {              // spill the oldest half of the window
                    for (uint32_t i = lane; i < DFS_SWN / 2u; i += 64u) gstack[sp_g + i] = sw[i];
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t i0 = 0; i0 + DFS_SWN / 2u < sp_l; i0 += 64u) {
                        const uint32_t i = i0 + lane;
                        uint32_t v = 0;
                        if (i + DFS_SWN / 2u < sp_l) v = sw[i + DFS_SWN / 2u];
                        __builtin_amdgcn_wave_barrier();
                        if (i + DFS_SWN / 2u < sp_l) sw[i] = v;
                    }
                    sp_g += DFS_SWN / 2u;
                    sp_l -= DFS_SWN / 2u;
                    __builtin_amdgcn_wave_barrier();
                }
                const unsigned last = 63u - (unsigned)__clzll(m);
                const uint32_t packed = ((uint32_t)ny << 16) | (uint32_t)nx;
                if (avail && lane != last) sw[sp_l + (uint32_t)__popcll(m & lt)] = packed;
                sp_l += npush - 1u;
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)packed, (int)last);
                ty = e >> 16; tx = e & 0xffffu;
                cnt += npush;
                __builtin_amdgcn_wave_barrier();
            }
        }
        // the piece's size is known here: cnt pixels were added after the seed (makeSegSize for
        // free); a piece that never grew is a one-pixel clump = candidate of the single-pixel pass
        if (lane == 0) {
            csize[seed] = cnt + 1u;
            if (cnt == 0 && singles) singles[atomicAdd(nsingles, 1u)] = seed;
        }
        __threadfence();
        cursor = seed + 1;
        if (cursor >= nrows * ncols) break;
    }
}

// The stack window of a walker (DFS_SWN entries in LDS): the oldest half is spilled to / refilled
// from the component's slice of a global scratch array.  Out of line: they run once per few
// hundred steps and must not weigh on the registers and wait counters of the walk.
// (stack pointers travel by value, packed sp_l | sp_g << 32: a reference parameter of an
// out-of-line function would pin them to scratch memory in the walk)
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))      // wave-uniform by construction
#define UNI64(x) ((unsigned long long)UNI((uint32_t)(x)) | ((unsigned long long)UNI((uint32_t)((x) >> 32)) << 32))
__device__ __noinline__ unsigned long long dfs_spill(uint32_t *sw, uint32_t *gstack, uint32_t sp_l, uint32_t sp_g)
{
    const unsigned lane = lane_id();
    for (uint32_t i = lane; i < DFS_SWN / 2u; i += 64u) gstack[sp_g + i] = sw[i];
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i0 = 0; i0 + DFS_SWN / 2u < sp_l; i0 += 64u) {
        const uint32_t i = i0 + lane;
        uint32_t v = 0;
        if (i + DFS_SWN / 2u < sp_l) v = sw[i + DFS_SWN / 2u];
        __builtin_amdgcn_wave_barrier();
        if (i + DFS_SWN / 2u < sp_l) sw[i] = v;
    }
    __builtin_amdgcn_wave_barrier();
    return (unsigned long long)(sp_l - DFS_SWN / 2u) | ((unsigned long long)(sp_g + DFS_SWN / 2u) << 32);
}
__device__ __noinline__ unsigned long long dfs_refill(uint32_t *sw, const uint32_t *gstack, uint32_t sp_g)
{
    const unsigned lane = lane_id();
    const uint32_t k = sp_g < DFS_SWN / 2u ? sp_g : DFS_SWN / 2u;
    for (uint32_t i = lane; i < k; i += 64u) sw[i] = gstack[sp_g - k + i];
    __builtin_amdgcn_wave_barrier();
    return (unsigned long long)k | ((unsigned long long)(sp_g - k) << 32);
}

// LDS variant of the replay for components whose bounding box fits the walker pool (all of
// them on the benchmark imagery): membership/unvisited state is one bit per bounding-box pixel
// in LDS, so a pop costs one LDS round trip instead of a dependent HBM/L2 access, and the walk
// no longer suffers when other streams pollute L2.  Labels are still written to `lab` with
// fire-and-forget stores.  Components that do not fit take dfs_split_global.

// The bitmap carries a one-bit border of zeros on every side, so the walk needs no bounds checks.
__device__ __forceinline__ unsigned long long dfs_bitmap_words(const BigInfo &B, uint32_t ncols)
{
    const uint32_t minr = B.root / ncols;
    const uint32_t H = B.maxr - minr + 1u, W = B.maxc - B.minc + 1u;
    const uint32_t wpr = (W + 2u + 31u) >> 5;
    return (unsigned long long)(H + 2u) * (wpr < 2u ? 2u : wpr);      // (a tile row of the walk is two words)
}

__device__ __forceinline__ void dfs_split_lds(uint32_t *lab, const BigInfo &B, uint32_t *bm,
                                              uint32_t *sw, uint32_t *stackbuf, uint32_t ncols,
                                              int four, uint32_t *singles, uint32_t *nsingles,
                                              uint32_t *csize)
{
    const uint32_t root = B.root;
    uint32_t *gstack = stackbuf + B.off;
    const unsigned lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    const uint32_t minr = root / ncols, minc = B.minc;
    const uint32_t H = B.maxr - minr + 1u, W = B.maxc - minc + 1u, wpr = (W + 2u + 31u) >> 5;
    const uint32_t nwords = (H + 2u) * wpr;
    // padded coordinates: member (y, x) of the box sits at row y + 1, bit column x + 1;
    // its pixel index is gbase + (y + 1) * ncols + (x + 1)   (mod 2^32)
    const uint32_t gbase = (minr - 1u) * ncols + minc - 1u;
    // ---- build the member bitmap from the flattened CCL labels: eight rows per step, so that
    //      eight independent 256-byte loads are in flight (a dependent load per row made the build
    //      as long as a short walk) ----
    for (uint32_t pr0 = 0; pr0 < H + 2u; pr0 += 8u) {
        for (uint32_t c0 = 0; c0 < wpr * 32u; c0 += 64u) {
            const uint32_t bc = c0 + lane;
            const bool cin = bc >= 1u && bc <= W;
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t pr = pr0 + u;
                v[u] = (cin && pr >= 1u && pr <= H) ? lab[gbase + pr * ncols + bc] : NULL_LAB;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t pr = pr0 + u;
                const unsigned long long m = __ballot(v[u] == root);
                if (pr < H + 2u && lane < 2u && (c0 >> 5) + lane < wpr)
                    bm[pr * wpr + (c0 >> 5) + lane] = lane ? (uint32_t)(m >> 32) : (uint32_t)m;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // Positions are kept in three forms at once, each a uniform value plus a per-lane neighbour delta:
    //   q  = bit index in the padded bitmap (row pitch P = 32 * wpr bits)   -> LDS word and bit
    //   g  = pixel index in the tile                                        -> label store
    //   pk = (row << 16 | column), padded coordinates                       -> stack entry
    // Lanes 0..nq-1 own the neighbours in the reference's push order (cx outer, cy inner); the other
    // lanes carry zero deltas, i.e. they probe the current position itself, whose bit is already
    // cleared: they never find anything and need no masking.
    int dy = 0, dx = 0;
    if (four) {
        dy = (lane == 1) ? -1 : (lane == 2) ? 1 : 0;
        dx = (lane == 0) ? -1 : (lane == 3) ? 1 : 0;
    } else if (lane < 8u) {
        const unsigned l8 = lane;              // (dy,dx) in push order: dx outer, dy inner
        dx = (l8 < 3u) ? -1 : (l8 < 5u) ? 0 : 1;
        dy = (l8 == 0u || l8 == 3u || l8 == 5u) ? -1 : (l8 == 1u || l8 == 6u) ? 0 : 1;
    }
    if (lane >= (four ? 4u : 8u)) { dy = 0; dx = 0; }
    const uint32_t P = wpr << 5;
    const uint32_t dq = (uint32_t)(dy * (int)P + dx);
    const uint32_t dg = (uint32_t)(dy * (int)ncols + dx);
    const uint32_t dpk = (uint32_t)(dy * 65536 + dx);
    const uint32_t ltm = (uint32_t)lt;           // (only lanes 0..7 ever push)
    uint32_t wcur = 0;                           // first word that can still hold a set bit
    for (;;) {
        // ---- next seed = first set bit in raster order ----
        uint32_t sword = 0xFFFFFFFFu, sbits = 0;
        for (uint32_t w0 = wcur; w0 < nwords; w0 += 64u) {
            const uint32_t wi = w0 + lane;
            const uint32_t v = wi < nwords ? bm[wi] : 0u;
            const unsigned long long m = __ballot(v != 0u);
            if (m) {
                const int fl = __builtin_ctzll(m);
                sword = w0 + (uint32_t)fl;
                sbits = (uint32_t)__builtin_amdgcn_readlane((int)v, fl);
                break;
            }
        }
        if (sword == 0xFFFFFFFFu) break;
        wcur = sword;
        const uint32_t sy = sword / wpr, sx = ((sword - sy * wpr) << 5) + (uint32_t)__builtin_ctz(sbits);
        const uint32_t seed = gbase + sy * ncols + sx;
        const uint32_t FL = seed | VIS_FLAG;
        if (lane == 0) {
            lab[seed] = FL;
            bm[sword] = sbits & (sbits - 1u);   // clear the seed's (lowest) bit
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t sp_l = 0, sp_g = 0;             // stack entries in the LDS window / spilled
        uint32_t cnt = 0;
        // current position (uniform)
        uint32_t cq = UNI(sy * P + sx), cg = UNI(seed), cpk = UNI((sy << 16) | sx);
        uint32_t q = cq + dq;
        uint32_t word = bm[q >> 5];              // probe of the current position's neighbours
        for (;;) {
            const bool avail = ((word >> (q & 31u)) & 1u) != 0u;
            const uint32_t m = (uint32_t)__ballot(avail);
            if (m == 0u) {
                // dead end: pop.  (refill the window from the spill area first when it ran dry)
                if (sp_l == 0u) {
                    if (sp_g == 0u) break;                       // stack empty: the piece is complete
                    const unsigned long long r = dfs_refill(sw, gstack, sp_g);
                    sp_l = UNI((uint32_t)r); sp_g = UNI((uint32_t)(r >> 32));
                }
                sp_l -= 1u;
                const uint32_t e = UNI(sw[sp_l]);
                cpk = e;
                cq = __umul24(e >> 16, P) + (e & 0xffffu);
                cg = gbase + __umul24(e >> 16, ncols) + (e & 0xffffu);
                q = cq + dq;
                word = bm[q >> 5];
                continue;
            }
            // this step's neighbours: clear their bits first (the next probe must see them gone; a
            // set bit is cleared by xor, two lanes may share a word), then move to the last one
            // pushed -- it is the next pop -- and probe from there while the labels and the stack
            // entries of this step are still being written
            const uint32_t g = cg + dg, pk = cpk + dpk;
            if (avail) atomicXor(&bm[q >> 5], 1u << (q & 31u));
            const int last = 31 - __builtin_clz(m);
            const uint32_t np = (uint32_t)__builtin_popcount(m);
            cq = (uint32_t)__builtin_amdgcn_readlane((int)q, last);
            cg = (uint32_t)__builtin_amdgcn_readlane((int)g, last);
            cpk = (uint32_t)__builtin_amdgcn_readlane((int)pk, last);
            q = cq + dq;
            word = bm[q >> 5];
            if (avail) {
                lab[g] = FL;
                // (the last one is written too, one slot above the new top: never read)
                sw[sp_l + (uint32_t)__builtin_popcount(m & ltm)] = pk;
            }
            sp_l += np - 1u;
            cnt += np;
            if (cnt >= MAX_CLUMP_SIZE) break;
            if (sp_l + 9u > DFS_SWN) {
                const unsigned long long r = dfs_spill(sw, gstack, sp_l, sp_g);
                sp_l = UNI((uint32_t)r); sp_g = UNI((uint32_t)(r >> 32));
            }
        }
        if (lane == 0) {
            csize[seed] = cnt + 1u;
            if (cnt == 0 && singles) singles[atomicAdd(nsingles, 1u)] = seed;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the replay with its neighbourhood in registers ------------------------------------------------
// The walk above pays an LDS round trip and ~45 dependent instructions per step.  A walker is one
// wavefront on one dependent chain, so it is paced by instruction ISSUE: 4.3-4.5 cycles per scalar
// or vector instruction for a lone wave, ~10 for a conditional branch that falls through, ~25 for
// a taken one, 8.8 for a v_readlane, 52 for an LDS round trip (tools/ubench/issue.hip).  Here:
//   * the walker keeps a 64 x 64-bit TILE of its bitmap in two VGPRs (lane i = padded row tr0 + i,
//     bits = padded columns 32 * twc .. + 63) and the three rows around the current position
//     (U, C, D) in SGPR pairs.  A step is scalar bit arithmetic on U / C / D -- which neighbours
//     are unvisited members, clear them, move to the last one pushed; a vertical move shifts the
//     window with one v_readlane pair (entering row) and one v_writelane pair (leaving row), a
//     horizontal move costs nothing; LDS is read only when the tile is re-centred;
//   * for 4-connectivity the steps run inside ONE assembly block (dfs_walk4_asm.h, generated by
//     tools/gen_dfs_walk4.py): the 4-bit neighbour mask indexes a table of 16 code blocks that know
//     statically what to clear, stack and where to move, each ending in the next step's table
//     jump.  It returns on a dead end or when its GUARD runs out: a budget, decremented by the
//     pixels each step marks, that is the minimum of what is left of the 10000-pixel cap, of the
//     room in the stack window and of the distance to the tile's rim -- one untaken branch per step
//     instead of three tests;
//   * labels are NOT stored per step.  A piece's pixels are exactly the bits it cleared, so at the
//     piece's end the rows it stood on are swept once with all 64 lanes: a pixel that still holds
//     the component's root in `lab` and whose bit is gone belongs to this piece (coalesced loads;
//     pixels of earlier pieces already carry their seed);
//   * a dead end does not pop one entry at a time: the 64 entries on top of the stack are tested
//     together, one lane each, and everything above the topmost entry that still has an unvisited
//     neighbour is dropped at once -- popping a dead entry has no effect in the reference either
//     (shepseg.py:507-536: nothing is marked, clumpSize stands), and most stacked pixels are dead
//     by the time they surface (mean run of 6-7 dead pops).
// Bit-exact replay of shepseg.py:490-539 like the walk above.
#include "dfs_walk4_asm.h"
// GLB: the bitmap does not fit the walker pool and lives in global memory (column-block-major, no snapshot): the same
// walk -- a step touches registers only, so the backing store's latency is paid at re-centrings, dead ends,
// seeds and piece ends -- with the bitmap read past this CU's vector cache (agent scope: lanes read words
// that other lanes of the wavefront stored).
template <bool FOUR, bool GLB = false>
__device__ __forceinline__ void dfs_split_win(uint32_t *lab, const BigInfo &B, uint32_t *bm,
                                              uint32_t *sw, uint32_t *stackbuf, uint32_t ncols,
                                              uint32_t *singles, uint32_t *nsingles, uint32_t *csize,
                                              uint32_t *snap, unsigned long long *prof)
{
    typedef unsigned long long u64;
#define BMLD(P) (GLB ? __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(P))
    // where word Wd of padded row R lives.  In LDS: row-major.  GLB: column-block-major (the 32-column word Wd of
    // every row, then word Wd + 1 of every row), so that the 64 rows of a tile are two runs of 256 bytes -- four
    // cache lines per tile load or write-back where the row-major bitmap took 64 to 128, and a re-centring is a
    // round trip of exactly those transactions issued by one wavefront.  BML: the same for a raster word index.
#define BMI(R, Wd) (GLB ? (Wd) * nprow + (R) : (R) * wpr + (Wd))
#define BML(I) (GLB ? bml_(I) : (I))
#ifdef DFS_PROF     // diagnostic build (make PROF=1): cycles per phase and event counts per component
    u64 pf_t = __builtin_readcyclecounter(), pf_build = 0, pf_dead = 0, pf_label = 0, pf_seed = 0, pf_rim = 0, pf_walk = 0, pf_asm = 0;
    u64 pf_nstep = 0, pf_ndead = 0, pf_nbulk = 0, pf_nrim = 0, pf_npiece = 0, pf_nrun = 0;
#define PF_LAP(acc) do { const u64 t_ = __builtin_readcyclecounter(); acc += t_ - pf_t; pf_t = t_; } while (0)
#define PF_CNT(c) (c)++
#define PF_ADD(c, n) (c) += (n)
#else
#define PF_LAP(acc) do { } while (0)
#define PF_CNT(c) do { } while (0)
#define PF_ADD(c, n) do { } while (0)
#endif
    const uint32_t root = B.root;
    uint32_t *gstack = stackbuf + B.off;
    const unsigned lane = lane_id();
    const uint32_t minr = root / ncols, minc = B.minc;
    const uint32_t H = B.maxr - minr + 1u, W = B.maxc - minc + 1u;
    uint32_t wpr_ = (W + 2u + 31u) >> 5;
    if (wpr_ < 2u) wpr_ = 2u;                   // a tile row is two words
    const uint32_t wpr = wpr_;
    const uint32_t nprow = H + 2u, nwords = nprow * wpr;
    const uint32_t gbase = (minr - 1u) * ncols + minc - 1u;
    // ---- member bitmap from the flattened CCL labels (eight row loads in flight) ----
    for (uint32_t pr0 = 0; pr0 < nprow; pr0 += 8u) {
        for (uint32_t c0 = 0; c0 < wpr * 32u; c0 += 64u) {
            const uint32_t bc = c0 + lane;
            const bool cin = bc >= 1u && bc <= W;
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t pr = pr0 + u;
                v[u] = (cin && pr >= 1u && pr <= H) ? lab[gbase + pr * ncols + bc] : NULL_LAB;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t pr = pr0 + u;
                const u64 mm = __ballot(v[u] == root);
                if (pr < nprow && lane < 2u && (c0 >> 5) + lane < wpr) {
                    const uint32_t wv = lane ? (uint32_t)(mm >> 32) : (uint32_t)mm;
                    bm[BMI(pr, (c0 >> 5) + lane)] = wv;
                    if (!GLB) snap[pr * wpr + (c0 >> 5) + lane] = wv;       // the bitmap as of the last piece's end
                }
            }
        }
    }
    if (GLB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint32_t wpr_inv = 0xFFFFFFFFu / wpr + 1u;              // __umulhi(i, wpr_inv) == i / wpr for i < 2^32 / wpr
    auto bml_ = [&](uint32_t i) { const uint32_t r = __umulhi(i, wpr_inv); return (i - r * wpr) * nprow + r; };
    // per-lane constants (8-connectivity and the dead-end test): lanes 0..nq-1 own the neighbours in
    // the reference's push order (cx outer, cy inner); dpk = offset in packed (row << 16 | col) form
    constexpr uint32_t NQ = FOUR ? 4u : 8u;
    int dy = 0, dx = 0;
    if (FOUR) {
        dy = (lane == 1) ? -1 : (lane == 2) ? 1 : 0;
        dx = (lane == 0) ? -1 : (lane == 3) ? 1 : 0;
    } else if (lane < 8u) {
        dx = (lane < 3u) ? -1 : (lane < 5u) ? 0 : 1;
        dy = (lane == 0u || lane == 3u || lane == 5u) ? -1 : (lane == 1u || lane == 6u) ? 0 : 1;
    }
    if (lane >= NQ) { dy = 0; dx = 0; }
    const uint32_t dpk = (uint32_t)(dy * 65536 + dx);
    const uint32_t lanebit = lane < NQ ? (1u << lane) : 0u;
    const uint32_t ltm = lane < 32u ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
    const uint32_t trmax = nprow > 64u ? nprow - 64u : 0u;
    // (readfirstlane: the walker's window address comes from threadIdx.x / 64, uniform but not provably so)
    const uint32_t sw_addr = UNI((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)sw);
    // the tile (registers) and where it sits
    uint32_t tlo = 0, thi = 0, tr0 = 0, twc32 = 0;
    uint32_t olo = 0, ohi = 0;                  // GLB: the tile as it was loaded / last written back
    // (RB / CB: where in the tile the position is put -- 32 / 16 centres it; a walk that left the last tile through
    //  its lower rim is put 8 rows below the new tile's top, and so on: a straight walk then re-centres every ~54
    //  steps instead of every ~31, and a re-centring is a round trip to the bitmap)
#define DFSW_TILE_LOAD(PR, PC) DFSW_TILE_LOAD_AT(PR, PC, 32, 16)
#define DFSW_TILE_LOAD_AT(PR, PC, RB, CB)                                                           \
    do {                                                                                            \
        int t_ = (int)(PR) - (RB);                                                                  \
        t_ = t_ < 0 ? 0 : t_;                                                                       \
        tr0 = UNI((uint32_t)t_ > trmax ? trmax : (uint32_t)t_);                                     \
        int w_ = ((int)(PC) - (CB)) >> 5;                                                           \
        w_ = w_ < 0 ? 0 : w_;                                                                       \
        const uint32_t twc_ = (uint32_t)w_ > wpr - 2u ? wpr - 2u : (uint32_t)w_;                    \
        twc32 = UNI(twc_ << 5);                                                                     \
        const uint32_t r_ = tr0 + lane;                                                             \
        const bool ok_ = r_ < nprow;                                                                \
        const uint32_t a_ = BMI(r_, twc_), a1_ = BMI(r_, twc_ + 1u);                                \
        tlo = ok_ ? BMLD(&bm[a_]) : 0u;                                                             \
        thi = ok_ ? BMLD(&bm[a1_]) : 0u;                                                            \
        if (GLB) { olo = tlo; ohi = thi; }                                                          \
    } while (0)
    // GLB: a piece's pixels are labelled HERE, from the registers -- the bits that went since the tile was loaded
    // (or last written back).  Round 3 swept the bitmap against a snapshot at the piece's end, cell by cell with a
    // lane per row: 256 cache-line transactions per cell of 64 rows x 64 columns, ~320 cells for a piece of a
    // uniform region (one row plus a two-pixel strip of 3000 rows) -- 54 % of the time of a tile that is ONE
    // component.  The difference costs nothing to form here, and only the label stores touch memory.
#define DFSW_TILE_FLUSH()                                                                           \
    do {                                                                                            \
        const uint32_t r_ = tr0 + lane;                                                             \
        uint32_t d0_ = 0u, d1_ = 0u;                                                                \
        if (r_ < nprow) {                                                                           \
            bm[BMI(r_, twc32 >> 5)] = tlo;                                                          \
            bm[BMI(r_, (twc32 >> 5) + 1u)] = thi;                                                   \
            if (GLB) { d0_ = olo & ~tlo; d1_ = ohi & ~thi; }                                        \
        }                                                                                           \
        if (GLB) {                                                                                  \
            /* few rows changed (a sweep along a row): a row at a time, lane = column, one store;   \
               many (a run down a column: a bit or two per row): every lane walks its own bits */   \
            u64 rows_ = __ballot((d0_ | d1_) != 0u);                                                \
            if (__builtin_popcountll(rows_) <= 6) {                                                 \
                while (rows_) {                                                                     \
                    const int rr_ = __builtin_ctzll(rows_);                                         \
                    rows_ &= rows_ - 1ull;                                                          \
                    const uint32_t x0_ = (uint32_t)__builtin_amdgcn_readlane((int)d0_, rr_);        \
                    const uint32_t x1_ = (uint32_t)__builtin_amdgcn_readlane((int)d1_, rr_);        \
                    const uint32_t xb_ = lane < 32u ? x0_ >> lane : x1_ >> (lane - 32u);            \
                    if (xb_ & 1u) lab[gbase + (tr0 + (uint32_t)rr_) * ncols + twc32 + lane] = FL;   \
                }                                                                                   \
            } else {                                                                                \
                const uint32_t pb_ = gbase + r_ * ncols + twc32;                                    \
                while (d0_) { lab[pb_ + (uint32_t)__builtin_ctz(d0_)] = FL; d0_ &= d0_ - 1u; }      \
                while (d1_) { lab[pb_ + 32u + (uint32_t)__builtin_ctz(d1_)] = FL; d1_ &= d1_ - 1u; } \
            }                                                                                       \
            olo = tlo; ohi = thi;                                                                   \
        }                                                                                           \
        /* (GLB: no wait -- the loads that follow come from this wavefront too, and a wavefront's       \
            accesses to one address reach the L2 in program order) */                                \
        __builtin_amdgcn_wave_barrier();                                                            \
    } while (0)
#define DFSW_ROW_GET(R) ((u64)(uint32_t)__builtin_amdgcn_readlane((int)tlo, (int)(R)) |             \
                         ((u64)(uint32_t)__builtin_amdgcn_readlane((int)thi, (int)(R)) << 32))
#define DFSW_ROW_PUT(R, V)                                                                          \
    do {      /* (clang has no writelane builtin: a compare and two selects) */                     \
        const bool me_ = lane == (R);                                                               \
        tlo = me_ ? (uint32_t)(V) : tlo;                                                            \
        thi = me_ ? (uint32_t)((V) >> 32) : thi;                                                    \
    } while (0)
    // the window rows that are valid go back into the tile (on the rim one of them lies outside)
#define DFSW_WINDOW_PUT()                                                                           \
    do {                                                                                            \
        if (ry == 63u) { DFSW_ROW_PUT(62u, U); DFSW_ROW_PUT(63u, C); }                              \
        else if (ry == 0u) { DFSW_ROW_PUT(0u, C); DFSW_ROW_PUT(1u, D); }                            \
        else { DFSW_ROW_PUT(ry - 1u, U); DFSW_ROW_PUT(ry, C); DFSW_ROW_PUT(ry + 1u, D); }           \
    } while (0)
    uint32_t wcur = 0;                           // first word that can still hold a set bit
    PF_LAP(pf_build);
    for (;;) {
        // ---- next seed = first set bit in raster order (the bitmap in LDS is current here) ----
        uint32_t sword = 0xFFFFFFFFu, sbits = 0;
        for (uint32_t w0 = wcur; w0 < nwords; w0 += 64u) {
            const uint32_t wi = w0 + lane;
            const uint32_t v = wi < nwords ? BMLD(&bm[BML(wi)]) : 0u;
            const u64 mm = __ballot(v != 0u);
            if (mm) {
                const int fl = __builtin_ctzll(mm);
                sword = w0 + (uint32_t)fl;
                sbits = (uint32_t)__builtin_amdgcn_readlane((int)v, fl);
                break;
            }
        }
        if (sword == 0xFFFFFFFFu) break;
        wcur = sword;
        const uint32_t sy = sword / wpr, sx = ((sword - sy * wpr) << 5) + (uint32_t)__builtin_ctz(sbits);
        const uint32_t seed = gbase + sy * ncols + sx;
        const uint32_t FL = seed | VIS_FLAG;
        if (lane == 0) bm[BML(sword)] = sbits & (sbits - 1u);     // clear the seed's (lowest) bit
        if (GLB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the store has left before the tile is read)
        __builtin_amdgcn_wave_barrier();
        uint32_t sp_l = 0, sp_g = 0;             // stack entries in the LDS window / spilled
        uint32_t cnt = 0;
        uint32_t cpk = UNI((sy << 16) | sx);     // current position, packed padded coordinates
        uint32_t rmin = UNI(sy), rmax = rmin;    // rows the piece has stood on
        DFSW_TILE_LOAD(sy, sx);
        uint32_t lpr = sy, lpc = sx;              // where the walk stood when the tile was loaded
        uint32_t ry = UNI(sy - tr0), b = UNI(sx - twc32);        // both in 1 .. 62
        u64 U = DFSW_ROW_GET(ry - 1u), C = DFSW_ROW_GET(ry), D = DFSW_ROW_GET(ry + 1u);
        PF_LAP(pf_seed); PF_CNT(pf_npiece);
        for (;;) {
            if (FOUR) {
                // ---- the steps, in the assembly block (dfs_walk4_asm.h); it comes back for the events
                //      below.  Slack offsets: a side of the tile that coincides with the bitmap's own
                //      zero border cannot be crossed and never binds ----
                const int big = 1 << 20;
                int up_off = tr0 == 0u ? -big : 1, dn_off = tr0 + 64u >= nprow ? big : 62;
                int lf_off = twc32 == 0u ? -big : 1, rt_off = (twc32 >> 5) + 2u >= wpr ? big : 62;
                uint32_t rylo = UNI(ry + 1u), ryhi = rylo, why;
                uint32_t ssp = UNI(sw_addr + (sp_l << 2)), sw_end = UNI(sw_addr + (DFS_SWN << 2)), sw_base = sw_addr;
                uint32_t vcpk = cpk, vsp = ssp, vt, vt2, vlane = lane, a_tr0 = UNI(tr0), a_twc = UNI(twc32);
                // (every scalar operand through readfirstlane: the register constraints need values
                //  the compiler can PROVE uniform)
                U = UNI64(U); C = UNI64(C); D = UNI64(D); cnt = UNI(cnt);
                up_off = (int)UNI(up_off); dn_off = (int)UNI(dn_off); lf_off = (int)UNI(lf_off); rt_off = (int)UNI(rt_off);
                PF_LAP(pf_walk); PF_CNT(pf_nrun);
                asm volatile(DFS_WALK4_ASM
                             : "+{s[64:65]}"(U), "+{s[66:67]}"(C), "+{s[68:69]}"(D), "+{s73}"(rylo), "+{s74}"(ryhi), "={s75}"(why),
                               "+{s88}"(cnt), "+{s89}"(ssp),
                               [tlo] "+v"(tlo), [thi] "+v"(thi), [vcpk] "+v"(vcpk), [vsp] "+v"(vsp), [vt] "=&v"(vt),
                               [vt2] "=&v"(vt2)
                             : [vlane] "v"(vlane), "{s90}"(a_tr0), "{s91}"(a_twc), "{s92}"(up_off), "{s93}"(dn_off), "{s94}"(lf_off), "{s95}"(rt_off),
                               "{s96}"(sw_end), "{s99}"(sw_base)
                             : "s70", "s71", "s72", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86",
                               "s87", "s97", "s98", "scc", "memory");
                PF_LAP(pf_asm);
                // (and back: the compiler takes every result of an asm block with vector outputs as divergent)
                U = UNI64(U); C = UNI64(C); D = UNI64(D);
                cnt = UNI(cnt); ssp = UNI(ssp); rylo = UNI(rylo); ryhi = UNI(ryhi); why = UNI(why);
                cpk = UNI(vcpk);
                sp_l = (ssp - sw_addr) >> 2;
                ry = (cpk >> 16) - tr0; b = (cpk & 0xffffu) - twc32;
                {
                    const uint32_t a0 = tr0 + rylo - 1u, a1 = tr0 + ryhi - 1u;
                    rmin = a0 < rmin ? a0 : rmin;
                    rmax = a1 > rmax ? a1 : rmax;
                }
                if (why == 2u) {                 // the cap
                    DFSW_WINDOW_PUT();
                    DFSW_TILE_FLUSH();
                    break;
                }
                if (why == 3u) {                 // the stack window is nearly full
                    const u64 r = dfs_spill(sw, gstack, sp_l, sp_g);
                    sp_l = UNI((uint32_t)r); sp_g = UNI((uint32_t)(r >> 32));
                    continue;
                }
                if (why == 1u || why == 4u) {    // re-centre the tile / an entry outside it was popped
                    PF_LAP(pf_walk); PF_CNT(pf_nrim);
                    if (why == 1u) DFSW_WINDOW_PUT();           // (4: the block wrote the window back)
                    DFSW_TILE_FLUSH();
                    const uint32_t pr = cpk >> 16, pc = cpk & 0xffffu;
                    rmin = pr < rmin ? pr : rmin;
                    rmax = pr > rmax ? pr : rmax;
                    if (why == 1u) {
                        // which rim was it (ry, b: the position inside the tile that is left), and did the walk come a
                        // long way towards it since the tile was loaded?  (a walk that only turned round near the rim of
                        // an off-centre tile gets a centred one, or two tiles would hand it back and forth)
                        const int rb_ = (ry >= 50u && pr >= lpr + 24u) ? 8 : (ry <= 13u && lpr >= pr + 24u) ? 55 : 32;
                        const int cb_ = (b >= 50u && pc >= lpc + 24u) ? 8 : (b <= 13u && lpc >= pc + 24u) ? 24 : 16;
                        DFSW_TILE_LOAD_AT(pr, pc, rb_, cb_);
                        lpr = pr; lpc = pc;
                    } else {
                        DFSW_TILE_LOAD(pr, pc);
                        lpr = pr; lpc = pc;
                    }
                    ry = UNI(pr - tr0); b = UNI(pc - twc32);
                    U = DFSW_ROW_GET(ry - 1u); C = DFSW_ROW_GET(ry); D = DFSW_ROW_GET(ry + 1u);
                    PF_LAP(pf_rim);
                    continue;
                }
            } else {
                PF_CNT(pf_nstep);
                const uint32_t sh = b - 1u;
                const uint32_t u3 = (uint32_t)(U >> sh) & 7u, c3 = (uint32_t)(C >> sh) & 7u, d3 = (uint32_t)(D >> sh) & 7u;
                // unvisited member neighbours, bit = push order
                const uint32_t m = (u3 & 1u) | ((c3 & 1u) << 1) | ((d3 & 1u) << 2) | ((u3 & 2u) << 2) | ((d3 & 2u) << 3) |
                                   ((u3 & 4u) << 3) | ((c3 & 4u) << 4) | ((d3 & 4u) << 5);
                if (m != 0u) {
                    // ---- mark this step's neighbours (clear their bits), stack all but the last, move there ----
                    C &= ~(5ull << sh); U &= ~(7ull << sh); D &= ~(7ull << sh);
                    const uint32_t np = (uint32_t)__builtin_popcount(m);
                    // (the last one is stored too, one slot above the new top: never read)
                    if (m & lanebit) sw[sp_l + (uint32_t)__builtin_popcount(m & ltm)] = cpk + dpk;
                    sp_l += np - 1u;
                    cnt += np;
                    const int last = 31 - __builtin_clz(m);
                    cpk += (uint32_t)__builtin_amdgcn_readlane((int)dpk, last);
                    const uint32_t nry = (cpk >> 16) - tr0;
                    const int vy = (int)nry - (int)ry;
                    b = (cpk & 0xffffu) - twc32;
                    if (vy > 0) {
                        DFSW_ROW_PUT(ry - 1u, U);
                        U = C; C = D; ry += 1u;
                        const uint32_t pr = cpk >> 16;
                        rmax = pr > rmax ? pr : rmax;
                        if (ry < 63u) D = DFSW_ROW_GET(ry + 1u);
                    } else if (vy < 0) {
                        DFSW_ROW_PUT(ry + 1u, D);
                        D = C; C = U; ry -= 1u;
                        const uint32_t pr = cpk >> 16;
                        rmin = pr < rmin ? pr : rmin;
                        if (ry > 0u) U = DFSW_ROW_GET(ry - 1u);
                    }
                    if (!((ry - 1u) < 62u && (b - 1u) < 62u)) {
                        // the position reached the tile's rim: write the window back, re-centre
                        PF_LAP(pf_walk); PF_CNT(pf_nrim);
                        DFSW_WINDOW_PUT();
                        DFSW_TILE_FLUSH();
                        const uint32_t pr = cpk >> 16, pc = cpk & 0xffffu;
                        DFSW_TILE_LOAD(pr, pc);
                        lpr = pr; lpc = pc;
                        ry = UNI(pr - tr0); b = UNI(pc - twc32);
                        U = DFSW_ROW_GET(ry - 1u); C = DFSW_ROW_GET(ry); D = DFSW_ROW_GET(ry + 1u);
                        PF_LAP(pf_rim);
                    }
                    if (cnt >= MAX_CLUMP_SIZE) {
                        DFSW_WINDOW_PUT();
                        DFSW_TILE_FLUSH();
                        break;
                    }
                    if (sp_l + 9u > DFS_SWN) {
                        const u64 r = dfs_spill(sw, gstack, sp_l, sp_g);
                        sp_l = UNI((uint32_t)r); sp_g = UNI((uint32_t)(r >> 32));
                    }
                    continue;
                }
            }
            // ---- dead end: make the bitmap current, then find the topmost stack entry that still
            //      has an unvisited neighbour (everything above it pops without effect) ----
            PF_LAP(pf_walk); PF_CNT(pf_ndead);
            DFSW_WINDOW_PUT();
            DFSW_TILE_FLUSH();
            uint32_t e = 0xFFFFFFFFu;
            for (;;) {
                PF_CNT(pf_nbulk);
                if (sp_l == 0u) {
                    if (sp_g == 0u) break;                       // stack empty: the piece is complete
                    const u64 r = dfs_refill(sw, gstack, sp_g);
                    sp_l = UNI((uint32_t)r); sp_g = UNI((uint32_t)(r >> 32));
                }
                const uint32_t k = sp_l < 64u ? sp_l : 64u;
                const uint32_t base = sp_l - k;
                bool alive = false;
                uint32_t ent = 0;
                if (lane < k) {
                    ent = sw[base + lane];
                    const uint32_t er = ent >> 16, ec = ent & 0xffffu;
#pragma unroll
                    for (uint32_t q = 0; q < NQ; q++) {
                        int qy, qx;
                        if (FOUR) { qy = (q == 1u) ? -1 : (q == 2u) ? 1 : 0; qx = (q == 0u) ? -1 : (q == 3u) ? 1 : 0; }
                        else { qx = (q < 3u) ? -1 : (q < 5u) ? 0 : 1; qy = (q == 0u || q == 3u || q == 5u) ? -1 : (q == 1u || q == 6u) ? 0 : 1; }
                        const uint32_t rr = (uint32_t)((int)er + qy), cc = (uint32_t)((int)ec + qx);
                        alive = alive || (((BMLD(&bm[BMI(rr, cc >> 5)]) >> (cc & 31u)) & 1u) != 0u);
                    }
                }
                const u64 am = __ballot(alive);
                if (am) {
                    const int top = 63 - __builtin_clzll(am);
                    e = (uint32_t)__builtin_amdgcn_readlane((int)ent, top);
                    sp_l = base + (uint32_t)top;
                    break;
                }
                sp_l = base;
            }
            if (e == 0xFFFFFFFFu) { PF_LAP(pf_dead); break; }
            cpk = e;
            {
                const uint32_t pr = e >> 16, pc = e & 0xffffu;
                rmin = pr < rmin ? pr : rmin;
                rmax = pr > rmax ? pr : rmax;
                ry = pr - tr0; b = pc - twc32;
                if (!((ry - 1u) < 62u && (b - 1u) < 62u)) {
                    DFSW_TILE_LOAD(pr, pc);
                    lpr = pr; lpc = pc;
                    ry = UNI(pr - tr0); b = UNI(pc - twc32);
                }
            }
            U = DFSW_ROW_GET(ry - 1u); C = DFSW_ROW_GET(ry); D = DFSW_ROW_GET(ry + 1u);
            PF_LAP(pf_dead);
        }
        // ---- the piece is complete and the bitmap in LDS is current: label its pixels ----
        PF_LAP(pf_walk);
        if (FOUR) PF_ADD(pf_nstep, cnt);
        if (cnt == 0u) {
            if (lane == 0) {
                lab[seed] = FL;
                csize[seed] = 1u;
                if (!GLB) snap[sword] = sbits & (sbits - 1u);          // (the snapshot follows: only the seed's bit went)
                if (singles) singles[atomicAdd(nsingles, 1u)] = seed;
            }
        } else if (GLB) {
            // (every pixel but the seed was labelled when its tile was written back: DFSW_TILE_FLUSH)
            if (lane == 0) { lab[seed] = FL; csize[seed] = cnt + 1u; }
        } else {
            // the piece's pixels = the bits that went since the snapshot (the bitmap at the last piece's
            // end, in global memory), in the rows the piece stood on +- 1: a word per lane, four loads
            // in flight; the snapshot moves on as it is compared
            const uint32_t r0 = rmin > 1u ? rmin - 1u : 1u, r1 = rmax + 1u < H ? rmax + 1u : H;
            const uint32_t iend = (r1 + 1u) * wpr;
            for (uint32_t i0 = r0 * wpr; i0 < iend; i0 += 256u) {
                uint32_t sv[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t i = i0 + u * 64u + lane;
                    // (agent scope: past this CU's vector cache, which the stores below do not update)
                    sv[u] = i < iend ? __hip_atomic_load(&snap[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t i = i0 + u * 64u + lane;
                    const uint32_t cur = i < iend ? BMLD(&bm[i]) : 0u;
                    uint32_t d = sv[u] & ~cur;
                    if (d) {
                        snap[i] = cur;
                        const uint32_t pr = __umulhi(i, wpr_inv);
                        const uint32_t pbase = gbase + pr * ncols + ((i - pr * wpr) << 5);
                        do {
                            lab[pbase + (uint32_t)__builtin_ctz(d)] = FL;
                            d &= d - 1u;
                        } while (d);
                    }
                }
            }
            if (lane == 0) csize[seed] = cnt + 1u;
        }
        __builtin_amdgcn_wave_barrier();
        PF_LAP(pf_label);
    }
#ifdef DFS_PROF
    if (prof && lane == 0) {
        prof[0] = pf_build; prof[1] = pf_walk; prof[2] = pf_dead; prof[3] = pf_label; prof[4] = pf_seed; prof[5] = pf_rim;
        prof[6] = pf_nstep; prof[7] = pf_ndead; prof[8] = pf_nbulk; prof[9] = pf_nrim; prof[10] = pf_npiece; prof[11] = pf_asm; prof[12] = pf_nrun;
    }
#endif
#undef PF_LAP
#undef PF_CNT
#undef PF_ADD
#undef BMLD
#undef BMI
#undef BML
#undef DFSW_TILE_LOAD
#undef DFSW_MARK
#undef DFSW_MARK_ONE
#undef DFSW_TILE_FLUSH
#undef DFSW_ROW_GET
#undef DFSW_ROW_PUT
#undef DFSW_WINDOW_PUT
}

// order[rank] = component index, largest first: the longest replays start first (LPT), which
// shortens the makespan whenever there are more components than resident workgroups
// counters[2] = how many of them do not fit the walker pool (bitmap above bmw_small words)
// (mirror: the component count for the host, see PIN_MIRROR)
__global__ __launch_bounds__(256) void k_big_order(const BigInfo *__restrict__ big,
                                                   uint32_t *counters,
                                                   uint32_t *__restrict__ order, uint32_t ncols,
                                                   uint32_t bmw_small, uint32_t *mirror)
{
    const uint32_t nbig = counters[0];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u && mirror) MIRROR_STORE(mirror, nbig);
    if (i >= nbig) return;
    if (dfs_bitmap_words(big[i], ncols) > bmw_small) atomicAdd(&counters[2], 1u);
    const uint32_t si = big[i].size;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nbig; j++) {
        const uint32_t sj = big[j].size;
        rank += (sj > si || (sj == si && j < i)) ? 1u : 0u;
    }
    order[rank] = i;
}

// One launch per tile.  A workgroup holds DFS_WAVES independent walkers (one wavefront each) that
// share a pool of LDS granules: a walker takes a component, claims the contiguous granules its
// bounding-box bitmap needs (a bit mask in LDS, claimed with one atomic OR), walks, gives them back,
// takes the next component.  The bitmaps are sized by the component (median 6 KiB, 99 % below 28 KiB
// on the benchmark imagery) instead of by a launch-wide worst case (40 / 64 KiB in round 1), so a
// CU carries eight walkers in 84 KiB where it carried three in 138 KiB, and the rest of the LDS
// stays free for the other streams' kernels; the LDS footprint above 80 KiB keeps a second walker
// workgroup off the same CU.
// Components are taken largest first.  The first pass is static and interleaved (walker w of
// workgroup b takes rank w * gridDim.x + b), which hands every workgroup one component of each
// size class instead of the eight largest to workgroup 0; whatever is left is pulled from a
// counter.  A walker that finds no room waits for its neighbours' releases (they never wait for
// anything, so the wait is finite); after DFS_ALLOC_SPINS polls, or for a bitmap larger than the
// whole pool, it walks in global memory instead.
#define DFS_WAVES 8u
#define DFS_DBG_WORDS 20u        // SHEPSEG_DFS_STATS: 6 words per component + 11 of the DFS_PROF build
#define DFS_GRAN_WORDS 512u          // granule = 2 KiB
#define DFS_POOL_GRANS_DEFAULT 34u   // 68 KiB pool (+ 16 KiB of stack windows + the mask = 84 KiB)
#define DFS_ALLOC_SPINS 200000u
#define DFS_MAX_BLOCKS 256u          // 2048 walkers: 8 per CU

// start granule of `need` contiguous free granules, or -1; wave-uniform, lane 0 talks to the mask
__device__ __forceinline__ int dfs_pool_alloc(unsigned long long *mask, uint32_t need, uint32_t ngrans,
                                              uint32_t max_spins)
{
    int res = -1;
    if (lane_id() == 0) {
        const unsigned long long all = ngrans >= 64u ? ~0ull : ((1ull << ngrans) - 1ull);
        const unsigned long long ones = need >= 64u ? ~0ull : ((1ull << need) - 1ull);
        for (uint32_t spin = 0; spin < max_spins; spin++) {
            const unsigned long long m = __hip_atomic_load(mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long fr = ~m & all;
            unsigned long long f = fr;          // bit j of f: granules j .. j + need - 1 are free
            for (uint32_t have = 1; have < need;) {
                const uint32_t sh = need - have < have ? need - have : have;
                f &= f >> sh;
                have += sh;
            }
            if (f) {
                const int j = __builtin_ctzll(f);
                const unsigned long long claim = ones << j;
                const unsigned long long old = __hip_atomic_fetch_or(mask, claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!(old & claim)) { res = j; break; }
                __hip_atomic_fetch_and(mask, ~(claim & ~old), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    return __builtin_amdgcn_readfirstlane(res);
}

__global__ __launch_bounds__(DFS_WAVES * 64) void k_dfs_pool(        // blockDim.x / 64 walkers (<= DFS_WAVES)

    uint32_t *lab, const BigInfo *__restrict__ big, uint32_t *counters, uint32_t *stackbuf, uint32_t nrows,
    uint32_t ncols, int four, uint32_t pool_grans, uint32_t *singles, uint32_t *nsingles,
    const uint32_t *__restrict__ order, uint32_t *csize, unsigned long long *dbg, int oldwalk, uint32_t *snap,
    uint32_t *gscratch, uint32_t gscratch_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dfs_lds[];
    unsigned long long *mask = (unsigned long long *)dfs_lds;             // 4 words (2 used)
    // (readfirstlane: the walker index is wave-uniform, and everything derived from it -- the
    //  component record, the bitmap's geometry -- must be PROVABLY so to stay in scalar registers)
    const unsigned w = UNI(threadIdx.x >> 6), lane = lane_id();
    const uint32_t nwalk = blockDim.x >> 6;
    uint32_t *sw = dfs_lds + 4u + w * DFS_SWN;
    uint32_t *pool = dfs_lds + 4u + nwalk * DFS_SWN;
    uint32_t *turn = dfs_lds + 2u;          // first-pass allocations go in walker order (largest first)
    if (threadIdx.x == 0) { *mask = 0ull; *turn = 0u; }
    __syncthreads();                        // the only workgroup-wide rendezvous: walkers are independent
    const uint32_t nbig = counters[0];
    __builtin_amdgcn_s_setprio(3);          // lone latency-bound waves: win issue arbitration
    uint32_t idx = w * gridDim.x + blockIdx.x;
    bool first = true;
    for (;;) {
        if (first) {
            // walker w holds the workgroup's w-th largest component: it claims its bitmap before the
            // smaller ones behind it (a free-for-all let them crowd the largest one out for
            // milliseconds); one attempt each, whoever finds no room then retries at leisure
            if (lane == 0) {
                uint32_t spin = 0;
                while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != w && ++spin < DFS_ALLOC_SPINS)
                    __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (idx >= nbig) {
            if (first && lane == 0) __hip_atomic_fetch_add(turn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        const BigInfo B = big[order[idx]];
        const unsigned long long words = dfs_bitmap_words(B, ncols);
        int g0 = -1;
        uint32_t need = 0;
        const unsigned long long t0 = dbg ? wall_clock64() : 0ull;
        if (words <= (unsigned long long)pool_grans * DFS_GRAN_WORDS) {
            need = ((uint32_t)words + DFS_GRAN_WORDS - 1u) / DFS_GRAN_WORDS;
            g0 = dfs_pool_alloc(mask, need, pool_grans, first ? 1u : DFS_ALLOC_SPINS);
            if (first) {
                if (lane == 0) __hip_atomic_fetch_add(turn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (g0 < 0) g0 = dfs_pool_alloc(mask, need, pool_grans, DFS_ALLOC_SPINS);
            }
        } else if (first && lane == 0) {
            __hip_atomic_fetch_add(turn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        first = false;
        const unsigned long long t1 = dbg ? wall_clock64() : 0ull;
        if (g0 >= 0) {
            uint32_t *bmw = pool + (uint32_t)g0 * DFS_GRAN_WORDS;
            uint32_t *snapw = snap + (size_t)(blockIdx.x * nwalk + w) * ((size_t)pool_grans * DFS_GRAN_WORDS);
            if (oldwalk) dfs_split_lds(lab, B, bmw, sw, stackbuf, ncols, four, singles, nsingles, csize);
            else if (four) dfs_split_win<true>(lab, B, bmw, sw, stackbuf, ncols, singles, nsingles, csize, snapw, dbg ? dbg + (size_t)idx * DFS_DBG_WORDS + 6u : nullptr);
            else dfs_split_win<false>(lab, B, bmw, sw, stackbuf, ncols, singles, nsingles, csize, snapw, dbg ? dbg + (size_t)idx * DFS_DBG_WORDS + 6u : nullptr);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const unsigned long long ones = need >= 64u ? ~0ull : ((1ull << need) - 1ull);
                __hip_atomic_fetch_and(mask, ~(ones << g0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // the bitmap does not fit the pool (or no room came free): bitmap in global memory,
            // carved from the scratch block (counters[4] = words taken); the plain global walk when that is full
            uint32_t goff = 0xFFFFFFFFu;
            if (!oldwalk && words <= 0x3FFFFFFFull && gscratch) {
                if (lane == 0) {
                    // a CAS loop that only advances on success: a counter that kept adding after the block was
                    // full could wrap in 32 bits and hand a later component an offset inside live bitmaps
                    const uint32_t need2 = (uint32_t)words;
                    uint32_t cur = __hip_atomic_load(&counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        if ((unsigned long long)cur + need2 > (unsigned long long)gscratch_words) break;
                        const uint32_t seen = atomicCAS(&counters[4], cur, cur + need2);
                        if (seen == cur) { goff = cur; break; }
                        cur = seen;
                    }
                }
                goff = (uint32_t)__builtin_amdgcn_readfirstlane((int)goff);
            }
            if (goff != 0xFFFFFFFFu) {
                uint32_t *gbm = gscratch + goff, *gsnap = nullptr;      // (labels go out at tile write-back: no snapshot)
                if (four) dfs_split_win<true, true>(lab, B, gbm, sw, stackbuf, ncols, singles, nsingles, csize, gsnap, dbg ? dbg + (size_t)idx * DFS_DBG_WORDS + 6u : nullptr);
                else dfs_split_win<false, true>(lab, B, gbm, sw, stackbuf, ncols, singles, nsingles, csize, gsnap, dbg ? dbg + (size_t)idx * DFS_DBG_WORDS + 6u : nullptr);
            } else {
                dfs_split_global(lab, B, sw, stackbuf, nrows, ncols, four, singles, nsingles, csize);
            }
        }
        if (dbg && lane == 0) {          // SHEPSEG_DFS_STATS: size, bitmap words, wait / walk ticks (100 MHz), start
            unsigned long long *d = dbg + (size_t)idx * DFS_DBG_WORDS;
            d[0] = B.size; d[1] = words; d[2] = t1 - t0; d[3] = wall_clock64() - t1; d[4] = t0;
            d[5] = ((unsigned long long)blockIdx.x << 8) | w | (g0 < 0 ? 1ull << 40 : 0ull);
        }
        uint32_t nx = 0;
        if (lane == 0) nx = atomicAdd(&counters[3], 1u);
        idx = nwalk * gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
    }
}

struct SeedFn {
    const uint32_t *lab;
    __device__ __forceinline__ uint32_t operator()(uint32_t p) const
    {
        const uint32_t l = lab[p];
        return (l != NULL_LAB && (l & ~VIS_FLAG) == p) ? 1u : 0u;
    }
    __device__ __forceinline__ bool get4(uint32_t base, uint32_t v[4]) const
    {
        if (!scan_load4(lab, base, v)) return false;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) v[i] = (v[i] != NULL_LAB && (v[i] & ~VIS_FLAG) == base + i) ? 1u : 0u;
        return true;
    }
};

// seg[p] = raster rank of the piece's seed + 1, and the segment-size table of the clumps
// (makeSegSize, shepseg.py:544-569) in the same pass: every piece's size sits at its seed
// (uncut components: k_run_count; cut pieces: the replay kernel), so each id gets exactly one
// plain store; only the null pixels are counted here (one atomic per run per wavefront).
#define FINAL_SPAN 16u       // 256-pixel steps per workgroup
__global__ __launch_bounds__(256) void k_clump_final(const uint32_t *__restrict__ lab,
                                                     const uint32_t *__restrict__ rank,
                                                     const uint32_t *__restrict__ csize,
                                                     uint32_t *__restrict__ seg, uint32_t *segsz,
                                                     uint32_t n, uint32_t *__restrict__ singles,
                                                     uint32_t *nsingles,
                                                     const uint32_t *__restrict__ rank_boff)
{
    // one-pixel clumps and the null count are gathered in LDS over the workgroup's whole span and
    // flushed with ONE global atomic each: atomics on a single counter serialise at L2, a
    // per-wavefront (or even per-256-pixel) update costs more than the rest of the kernel
    __shared__ uint32_t s_buf[256u * FINAL_SPAN];
    __shared__ uint32_t s_cnt, s_base, s_null;
    if (threadIdx.x == 0) { s_cnt = 0; s_null = 0; }
    __syncthreads();
    const unsigned lane = lane_id();
    const uint32_t base = blockIdx.x * (256u * FINAL_SPAN);
    for (uint32_t it = 0; it < FINAL_SPAN; it++) {
        const uint32_t p = base + it * 256u + threadIdx.x;
        bool single = false;                    // one-pixel UNCUT clump (cut ones come from the replay)
        bool isnull = false;
        if (p < n) {
            const uint32_t l = lab[p];
            if (l == NULL_LAB) { seg[p] = 0u; isnull = true; }
            else {
                const uint32_t seed = l & ~VIS_FLAG;
                // (rank_boff: the seed scan's block offsets, added here instead of by a launch)
                const uint32_t id = rank[seed] + (rank_boff ? rank_boff[seed / SCAN_ITEMS] : 0u) + 1u;
                seg[p] = id;
                if (seed == p) {
                    const uint32_t sz = csize[p];
                    segsz[id] = sz;
                    single = sz == 1u && !(l & VIS_FLAG);
                }
            }
        }
        const unsigned long long mn = __ballot(isnull);
        if (mn != 0ull && lane == 0) atomicAdd(&s_null, (uint32_t)__popcll(mn));
        const unsigned long long ms = singles ? __ballot(single) : 0ull;
        if (ms != 0ull) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(ms));
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            if (single) s_buf[wbase + (uint32_t)__popcll(ms & lanemask_lt())] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_null) {
            atomicAdd(&segsz[0], s_null);
            if (nsingles) atomicAdd(&nsingles[1], s_null);      // mirror: one read-back gets all three
        }
        s_base = s_cnt ? atomicAdd(nsingles, s_cnt) : 0u;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < s_cnt; i += 256u) singles[s_base + i] = s_buf[i];
}

// d_clus (uint16, 0 = null) -> ctx->seg (uint32 clump ids 1..nclumps, 0 = null).
// *nclumps_dev: device uint32 receiving the number of clumps.
// d_segsz (optional): receives the clump sizes, must hold n + 2 entries.
static int run_clump(shp_ctx *ctx, const uint16_t *d_clus, uint32_t nrows, uint32_t ncols, int four,
                     uint32_t *d_seg, uint32_t *nclumps_dev, uint32_t *d_segsz = nullptr,
                     uint32_t *d_singles = nullptr, uint32_t *d_nsingles = nullptr, uint32_t cpitch = 0)
{
    if (cpitch == 0u) cpitch = ncols;            // d_clus: nrows rows of ncols ids, cpitch ids apart
    const uint64_t n64 = (uint64_t)nrows * ncols;
    if (n64 >= 0x7fffffffull) SHP_FAIL(ctx, SHP_ERR_ARG, "tile too large (%llu px)", (unsigned long long)n64);
    if (nrows > 65535u || ncols > 65535u)
        SHP_FAIL(ctx, SHP_ERR_ARG, "tile dimensions above 65535 are not supported (%u x %u)", nrows, ncols);
    const uint32_t n = (uint32_t)n64;
    if (n == 0) { HIPCHK(ctx, hipMemsetAsync(nclumps_dev, 0, 4, ctx->stream)); return 0; }
    const uint32_t maxbig = n / (MAX_CLUMP_SIZE + 2u) + 1u;
    CHK(buf_ensure(ctx, ctx->lab, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->aux, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->aux2, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->stack, (size_t)n * 4));
    CHK(buf_ensure(ctx, ctx->bigbits, ((size_t)n / 32 + 2) * 4));
    if (!d_segsz) {
        CHK(buf_ensure(ctx, ctx->segsz, ((size_t)n + 2) * 4));
        d_segsz = bp<uint32_t>(ctx->segsz);
    }
    CHK(buf_ensure(ctx, ctx->big, (size_t)maxbig * (sizeof(BigInfo) + 4) + 128));
    CHK(buf_ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(n)));
    uint32_t *lab = bp<uint32_t>(ctx->lab), *csize = bp<uint32_t>(ctx->aux);
    uint32_t *counters = (uint32_t *)((char *)ctx->big.p + (size_t)maxbig * sizeof(BigInfo));
    BigInfo *big = bp<BigInfo>(ctx->big);
    const unsigned g = grid_for(n, 256);
    hipStream_t st = ctx->stream;
    int ps = prof_begin(ctx, PROF_CCL);
    hipLaunchKernelGGL(k_ccl_local, dim3(grid_for(ncols, 64), grid_for(nrows, CCL_ROWS)), dim3(256), 0, st, d_clus,
                       lab, csize, nrows, ncols, four, counters, d_segsz, d_nsingles, cpitch); KCHK(ctx);
    {
        const uint32_t ntop = (nrows + CCL_ROWS - 1u) / CCL_ROWS, npc = (ncols + 63u) / 64u;
        const size_t nborder = (size_t)ntop * ncols + (size_t)nrows * 2u * npc;
        hipLaunchKernelGGL(k_ccl_border, dim3(grid_for(nborder, 256)), dim3(256), 0, st, d_clus, lab, nrows, ncols,
                           four, ntop, npc, cpitch); KCHK(ctx);
    }
    uint32_t *bigbits = bp<uint32_t>(ctx->bigbits), *rank = bp<uint32_t>(ctx->aux2);
    hipLaunchKernelGGL(k_ccl_flatten, dim3(grid_for(n, 256u * FLAT_PPT)), dim3(256), 0, st, lab, n, csize, bigbits); KCHK(ctx);
    prof_end(ctx, ps);
    // (SHEPSEG_DBG_SKIP_DFS: diagnostic only -- no component is cut, wrong labels, times the rest)
    static const uint32_t bigmin = getenv("SHEPSEG_DBG_SKIP_DFS") ? 0x7fffffffu : MAX_CLUMP_SIZE + 2u;
    hipLaunchKernelGGL(k_big_list, dim3(grid_for(n, 1024)), dim3(256), 0, st, lab, csize, n, ncols, big, counters, bigbits, bigmin); KCHK(ctx);
    hipLaunchKernelGGL(k_big_bbox, dim3(grid_for(ncols, 64), grid_for(nrows, AGG_ROWS)), dim3(256), 0, st, lab,
                       csize, nrows, ncols, big, bigbits); KCHK(ctx);
    uint32_t *order = (uint32_t *)((char *)ctx->big.p + (size_t)maxbig * sizeof(BigInfo) + 64);
    uint32_t *mir_nbig = ctx->h_pinned + PIN_MIRROR + MIR_NBIG;
    hipLaunchKernelGGL(k_big_order, dim3(grid_for(maxbig, 256)), dim3(256), 0, st, big, counters, order, ncols,
                       DFS_POOL_GRANS_DEFAULT * DFS_GRAN_WORDS, mir_nbig); KCHK(ctx);
    // the replay is a latency-bound phase: outside the fill gate.  The component count comes back
    // first, so that the launch is sized exactly (a workgroup per DFS_WAVES components; every one of
    // them reserves the whole walker pool in LDS, so none is launched for nothing).
    HIPCHK(ctx, hipStreamSynchronize(st));
    const uint32_t nbig_h = *(volatile uint32_t *)mir_nbig;
    fill_release(ctx, false);
    if (nbig_h) walk_begin(ctx);
    st = ctx->stream;
    ps = prof_begin(ctx, PROF_DFS);              // events hug the kernel
    unsigned long long *dbg = nullptr;
    if (getenv("SHEPSEG_DFS_STATS") && nbig_h) {
        CHK(buf_ensure(ctx, ctx->dbg, (size_t)nbig_h * DFS_DBG_WORDS * 8u));
        HIPCHK(ctx, hipMemsetAsync(ctx->dbg.p, 0, (size_t)nbig_h * DFS_DBG_WORDS * 8u, st));
        dbg = bp<unsigned long long>(ctx->dbg);
    }
    if (nbig_h) {
        static const uint32_t pool_grans = getenv("SHEPSEG_DFS_POOL") ? (uint32_t)atoi(getenv("SHEPSEG_DFS_POOL")) : DFS_POOL_GRANS_DEFAULT;
        static const uint32_t per_wg = getenv("SHEPSEG_DFS_PER_WG") ? (uint32_t)atoi(getenv("SHEPSEG_DFS_PER_WG")) : DFS_WAVES;
        static const int oldwalk = getenv("SHEPSEG_DFS_OLDWALK") ? atoi(getenv("SHEPSEG_DFS_OLDWALK")) : 0;
        const uint32_t pg = pool_grans < 1u ? 1u : pool_grans > 64u ? 64u : pool_grans;
        const uint32_t pw = per_wg < 1u ? 1u : per_wg > DFS_WAVES ? DFS_WAVES : per_wg;
        const size_t lds = (4u + pw * DFS_SWN + (size_t)pg * DFS_GRAN_WORDS) * 4u;
        static bool attr_set = false;       // (benign race: the attribute is idempotent)
        if (!attr_set) {
            HIPCHK(ctx, hipFuncSetAttribute((const void *)k_dfs_pool, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            attr_set = true;
        }
        // at most DFS_MAX_BLOCKS workgroups (the rest of the components is pulled from the counter):
        // every walker owns a slot of the snapshot buffer as large as the walker pool
        uint32_t nblk = (nbig_h + pw - 1u) / pw;
        nblk = nblk > DFS_MAX_BLOCKS ? DFS_MAX_BLOCKS : nblk;
        CHK(buf_ensure(ctx, ctx->snap, (size_t)nblk * pw * pg * DFS_GRAN_WORDS * 4u));
        hipLaunchKernelGGL(k_dfs_pool, dim3(nblk), dim3(pw * 64u), lds, st, lab, big,
                           counters, bp<uint32_t>(ctx->stack), nrows, ncols, four, pg, d_singles, d_nsingles,
                           order, csize, dbg, oldwalk, bp<uint32_t>(ctx->snap), rank /* aux2: free until the seed scan */,
                           n); KCHK(ctx);
    }
    prof_end(ctx, ps);
    if (dbg) {
        const size_t DW = DFS_DBG_WORDS;
        std::vector<unsigned long long> h((size_t)nbig_h * DW);
        HIPCHK(ctx, hipMemcpyAsync(h.data(), dbg, h.size() * 8u, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        unsigned long long tmin = ~0ull, tend = 0, swait = 0, swalk = 0, spx = 0;
        for (uint32_t i = 0; i < nbig_h; i++) {
            tmin = h[i * DW + 4] < tmin ? h[i * DW + 4] : tmin;
            const unsigned long long e = h[i * DW + 4] + h[i * DW + 2] + h[i * DW + 3];
            tend = e > tend ? e : tend;
            swait += h[i * DW + 2]; swalk += h[i * DW + 3]; spx += h[i * DW];
        }
        fprintf(stderr, "dfs: %u components, %llu px, span %.2f ms, sum wait %.2f ms, sum walk %.2f ms (%.1f ns/px)\n",
                nbig_h, spx, (tend - tmin) / 1e5, swait / 1e5, swalk / 1e5, swalk * 10.0 / (double)(spx ? spx : 1));
        for (uint32_t i = 0; i < nbig_h && i < 12u; i++)
            fprintf(stderr, "  rank %u: %llu px, %llu words, start %.2f wait %.2f walk %.2f ms (%.1f ns/px) wg %llu wave %llu%s\n", i,
                    h[i * DW], h[i * DW + 1], (h[i * DW + 4] - tmin) / 1e5, h[i * DW + 2] / 1e5, h[i * DW + 3] / 1e5,
                    h[i * DW + 3] * 10.0 / (double)h[i * DW], (h[i * DW + 5] >> 8) & 0xffffffffull, h[i * DW + 5] & 255ull,
                    (h[i * DW + 5] >> 40) ? " GLOBAL" : "");
    #ifdef DFS_PROF
        {
            unsigned long long a[13] = {0};
            for (uint32_t i = 0; i < nbig_h; i++) for (int j = 0; j < 13; j++) a[j] += h[i * DW + 6 + j];
            fprintf(stderr, "  prof (Mcycles): build %.1f walk %.1f dead %.1f label %.1f seed %.1f rim %.1f | steps %llu dead ends %llu bulk iters %llu rims %llu pieces %llu | %.1f cycles/step, %.0f cycles/dead end, %.0f cycles/piece label\n",
                    a[0] / 1e6, a[1] / 1e6, a[2] / 1e6, a[3] / 1e6, a[4] / 1e6, a[5] / 1e6, a[6], a[7], a[8], a[9], a[10],
                    (double)a[1] / (double)(a[6] ? a[6] : 1), (double)a[2] / (double)(a[7] ? a[7] : 1), (double)a[3] / (double)(a[10] ? a[10] : 1));
            fprintf(stderr, "  asm runs %llu, %.1f Mcycles inside (%.1f per marked px), %.1f outside per run\n", a[12], a[11] / 1e6,
                    (double)a[11] / (double)(a[6] ? a[6] : 1), (double)a[1] / (double)(a[12] ? a[12] : 1));
            const unsigned long long *p0 = &h[6];
            fprintf(stderr, "  rank 0 prof (Mcycles): build %.2f walk %.2f dead %.2f label %.2f seed %.2f rim %.2f | steps %llu dead %llu bulk %llu rims %llu pieces %llu\n",
                    p0[0] / 1e6, p0[1] / 1e6, p0[2] / 1e6, p0[3] / 1e6, p0[4] / 1e6, p0[5] / 1e6, p0[6], p0[7], p0[8], p0[9], p0[10]);
        }
#endif
    }
    if (fill_gating(ctx) || stream_sharing(ctx)) {
        HIPCHK(ctx, hipStreamSynchronize(st));
        walk_end(ctx);
        fill_acquire(ctx, 1);
        st = ctx->stream;
    }
    ps = prof_begin(ctx, PROF_LABEL);
    // seed rank -> clump id
    SeedFn sf{lab};
    const uint32_t *rank_boff = nullptr;
    CHK(scan_exclusive(ctx, sf, n, rank, nclumps_dev, bp<uint32_t>(ctx->scan_tmp), &rank_boff));
    hipLaunchKernelGGL(k_clump_final, dim3(grid_for(n, 256u * FINAL_SPAN)), dim3(256), 0, st, lab, rank, csize, d_seg, d_segsz, n,
                       d_singles, d_nsingles, rank_boff); KCHK(ctx);
    prof_end(ctx, ps);
    return 0;
}
