// fit_bounds.h -- Elkan's E-step WITHOUT a stored table of exact lower bounds (k <= 64; included by
// fit_elkan.h, which keeps the exact-table kernels for k > 64).
//
// What the reference keeps (sklearn 0.24.2 _k_means_elkan.pyx): lower_bounds[i][j], a float64 per sample
// and centre, set to the distance whenever that distance is computed and lowered by the centre's shift
// (clipped at 0) at the end of EVERY iteration.  Round 3 kept that table (k x n float64, 495 MB for the
// benchmark sample) and streamed all of it through the E-step every iteration: 0.96 GB per pass, 260 GB
// per fit, 72 % of it rows the reference does not look at.
//
// What decides the reference's result is not the bounds' values but the outcomes of its comparisons
// `upper > lower_bounds[i][j]`.  A bound is a pure function of (the iteration tau it was last set in, the
// distance computed then, the shifts of centre j since):
//       lb_ij(t) = clip(... clip(clip(d(x_i, C_tau[j]) - s_tau[j]) - s_tau+1[j]) ... - s_t-1[j])
// so it can be RECOMPUTED bit for bit from a 2-byte stamp tau_ij, the history of the centres (k x nb
// float64 per iteration) and of their shifts -- and it can be BRACKETED without any of that by
//       A_ij - S_j(t) -/+ eps,   A_ij = float32(d + S_j(tau)) rounded down,  S_j(t) = sum of s_1..t-1[j]
// (both ends clipped at 0; a pair never computed is exactly 0).  The bracket is ~1e-3 wide; a comparison
// whose `upper` falls outside it -- all but a handful per million -- is decided by the bracket, the rest
// by the exact recomputation.  Nothing is updated per iteration: a row changes only where the reference
// computes a distance.  Per iteration the E-step reads n x k float32 (248 MB) instead of moving 960 MB,
// and rows are sample-major, so work that skips a sample skips its lines.
//
// An E-step is a FILTER with a lane per centre (rows streamed, coalesced: which samples can the reference
// touch at all, and through which centres) and a VISIT with a lane per visited sample (the reference's scan,
// decisions bit for bit those of _update_chunk_dense): k_elk2_filter / k_elk2_visit below.
#pragma once

struct ElkHist {
    double *cs;         // [it * k + j]        shift of centre j in iteration it (1 ..)
    double *cum;        // [t * k + j]         S_j(t) = cs[1][j] + ... + cs[t-1][j]   (cum[1] = 0)
    double *cen;        // [t * k * nb + ...]  the centres iteration t's E-step runs with (cen[1] = initial)
    double *csT;        // [j * rows + it]     cs again, a centre's shifts adjacent (elk2_exact's replay)
    uint32_t rows;      // iterations the history has room for
};

#define ELK2_EPS_ABS 1e-6
#define ELK2_EPS_REL 1e-9

__device__ __forceinline__ float elk2_f32_down(double s)
{
    float f = (float)s;
    if ((double)f > s) f = __uint_as_float(__float_as_uint(f) - 1u);        // (s >= 0: f > 0 here)
    return f;
}
__device__ __forceinline__ double rl_f64(double v, int lane)
{
    lane = __builtin_amdgcn_readfirstlane(lane);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// one lane per centre: the distances of sample row x (uniform address) to every centre, each in the
// operation order of _euclidean_dense_dense (elk_dist).  ct: the centres band-major (ct[b * k + lane]).
__device__ __forceinline__ double elk2_dist_lane(const double *__restrict__ x, const double *ct, int k, int nb, int lane)
{
    const int n4 = nb / 4, rem = nb % 4;
    double result = 0.0;
    int b = 0;
    for (int i = 0; i < n4; i++, b += 4) {
        const double d0 = x[b] - ct[b * k + lane], d1 = x[b + 1] - ct[(b + 1) * k + lane];
        const double d2 = x[b + 2] - ct[(b + 2) * k + lane], d3 = x[b + 3] - ct[(b + 3) * k + lane];
        result += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
    }
    for (int i = 0; i < rem; i++, b++) { const double d = x[b] - ct[b * k + lane]; result += d * d; }
    return __builtin_sqrt(result);
}

// lb_ij(t) recomputed exactly (rare: a handful of comparisons per million).  The replay of the shifts is one
// dependent chain per call and a call holds its wavefront, so the shifts of a centre lie adjacent (csT) and
// are fetched ELK2_REPLAY at a time, the next batch in flight while this one is applied; a bound that
// reaches 0 stays there.
#define ELK2_REPLAY 16
__device__ __noinline__ double elk2_exact(const double *__restrict__ X, int nb, int k, uint32_t i, int j, uint32_t t,
                                          const uint16_t *__restrict__ stamps, ElkHist h)
{
    const uint32_t tau = stamps[(size_t)i * k + j];
    if (tau == 0u) return 0.0;
    double v = elk_dist(X + (size_t)i * nb, h.cen + ((size_t)tau * k + j) * nb, nb);
    const double *cs = h.csT + (size_t)j * h.rows;
    double a[ELK2_REPLAY], b[ELK2_REPLAY];
#pragma unroll
    for (int u = 0; u < ELK2_REPLAY; u++) a[u] = cs[tau + (uint32_t)u < t ? tau + (uint32_t)u : tau];
    for (uint32_t tt = tau; tt < t; tt += ELK2_REPLAY) {
#pragma unroll
        for (int u = 0; u < ELK2_REPLAY; u++) {
            const uint32_t q = tt + ELK2_REPLAY + (uint32_t)u;
            b[u] = cs[q < t ? q : tau];
        }
#pragma unroll
        for (int u = 0; u < ELK2_REPLAY; u++)
            if (tt + (uint32_t)u < t) { v -= a[u]; if (v < 0) v = 0; }
        if (v == 0.0) return 0.0;
#pragma unroll
        for (int u = 0; u < ELK2_REPLAY; u++) a[u] = b[u];
    }
    return v;
}

// init_bounds_dense.  A: n x k float32 (-inf = never computed), stamps: n x k (0 = never).
// block 256 (a wavefront per 64 samples), workgroups stride over chunks of 256 samples.  cen: the initial centres.
__global__ __launch_bounds__(256) void k_elk2_init(const double *__restrict__ X, uint32_t n, int nb,
                                                   const double *__restrict__ cen, int k,
                                                   const double *__restrict__ half,
                                                   int32_t *__restrict__ lab, double *__restrict__ ub,
                                                   float *__restrict__ A, uint16_t *__restrict__ stamps)
{
    extern __shared__ double sh2[];
    double *sh_half = sh2, *sh_ct = sh2 + k * k;
    for (int t = threadIdx.x; t < k * k; t += 256) sh_half[t] = half[t];
    for (int t = threadIdx.x; t < k * nb; t += 256) { const int j = t / nb, b = t - j * nb; sh_ct[b * k + j] = cen[t]; }
    __syncthreads();
    const int lane = (int)lane_id();
    const int cl = lane < k ? lane : k - 1;
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * 256u < n; chunk += gridDim.x) {
    const uint32_t base = chunk * 256u + (threadIdx.x & ~63u);
    int my_lab = 0;
    double my_ub = 0.0;
    for (int s = 0; s < 64; s++) {
        const uint32_t i = base + (uint32_t)s;
        if (i >= n) break;
        const double *x = X + (size_t)__builtin_amdgcn_readfirstlane((int)i) * nb;
        const double D = elk2_dist_lane(x, sh_ct, k, nb, cl);
        int best = 0;
        double min_dist = rl_f64(D, 0);
        unsigned long long computed = 1ull;
        for (int j = 1; j < k; j++)
            if (min_dist > sh_half[best * k + j]) {
                const double dist = rl_f64(D, j);
                computed |= 1ull << j;
                if (dist < min_dist) { min_dist = dist; best = j; }
            }
        if (lane < k) {
            const bool c = (computed >> lane) & 1ull;
            A[(size_t)i * k + lane] = c ? elk2_f32_down(D) : -__builtin_inff();      // S_j(1) = 0
            stamps[(size_t)i * k + lane] = c ? (uint16_t)1 : (uint16_t)0;
        }
        if (lane == s) { my_lab = best; my_ub = min_dist; }
    }
    const uint32_t i = base + (uint32_t)lane;
    if (i < n) { lab[i] = my_lab; ub[i] = my_ub; }
    }
}

// `count` doubles of a small global table into registers first, then wherever `put` takes them: the loads of a
// thread are independent and all in flight (a load -> LDS store loop pays a memory round trip per 256 entries,
// 15 of them for the half distances -- a third of a late iteration's visit kernel)
template <class Put>
__device__ __forceinline__ void elk2_stage(const double *__restrict__ src, int count, Put put)
{
    for (int q0 = 0; q0 < count; q0 += 256 * 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int q = q0 + u * 256 + (int)threadIdx.x;
            v[u] = src[q < count ? q : count - 1];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int q = q0 + u * 256 + (int)threadIdx.x;
            if (q < count) put(q, v[u]);
        }
    }
}

#define ELK2_ROWS 8                     // bounds rows per group
#define ELK2_WIN 4                      // groups in flight per wavefront (8 groups = the 64 samples of a chunk)

__device__ __forceinline__ float elk2_f32_up(double s)       // the float32 >= s (s finite, > -FLT_MAX)
{
    float f = (float)s;
    if ((double)f < s) f = f >= 0.0f ? __uint_as_float(__float_as_uint(f) + 1u) : __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

// One E-step of iteration t (t >= 1) is two kernels.
//
// k_elk2_filter, a lane per centre: upper += the previous iteration's shift of the sample's centre (the end of
// elkan_iter; stored), then for every sample whose gate `next[label] >= upper` is open its row of brackets:
//   m0 = the centres whose bound MAY lie below the sample's upper bound (A_ij < (upper + S_j(t))(1 + rel) + abs),
//   c  = those of m0 that also clear the half-distance test with the sample's label.
// c == 0: the reference does not touch the sample.  Otherwise c and m0 go to cmask[i] / mmask[i]; m0 is a
// superset of every centre the reference's scan can stop at while upper only falls, WHATEVER label the scan
// moves to.  The filter only has to be a superset, so it runs in float32 with the roundings pushed outwards
// (upper and the threshold up, the half distances down): per row three lane reads, an LDS read, one fma, two
// compares and -- for rows with candidates only -- the mask hand-over.
//   The rows come through a rolling window of ELK2_WIN groups of ELK2_ROWS rows per wavefront (32 rows, 7.5 KB,
// in flight at any time, across chunk boundaries).  Every row of a chunk is loaded, whatever its sample's gate
// says (a closed gate filters with upper = -inf), so a load depends on nothing but the chunk number; rows past
// the end re-read the last row.  The vector-memory counter retires in order, so whatever is consumed first is
// issued first (a chunk's scalars before its rows; shifts and nearest-centre distances sit in LDS, which counts
// separately) and the code is straight-line: the wait before a group's first use is a count of the newer loads.
// PROBE (timing experiments, SHEPSEG_ELK2_PROBE=1, stores nothing): 1 = as is; 2 = the scalars only.
template <int PROBE = 0>
__global__ __launch_bounds__(256) void k_elk2_filter(uint32_t n, int k, const double *__restrict__ half,
                                                     const double *__restrict__ next,
                                                     const double *__restrict__ csprev,
                                                     const double *__restrict__ cumt,
                                                     const int32_t *__restrict__ lab, double *__restrict__ ub,
                                                     const float *__restrict__ A,
                                                     unsigned long long *__restrict__ cmask,
                                                     unsigned long long *__restrict__ mmask, const uint32_t *stop)
{
    if (stop && *stop) return;
    extern __shared__ double sh2[];
    double *sh_csp = sh2, *sh_next = sh2 + k;
    float *sh_hf = (float *)(sh_next + k);
    for (int q = threadIdx.x; q < k; q += 256) { sh_csp[q] = csprev ? csprev[q] : 0.0; sh_next[q] = next[q]; }
    // half distances rounded down; +inf on the diagonal (`upper > half[label][j]` then fails for j == label by itself)
    elk2_stage(half, k * k, [&](int q, double h) {
        float f = (float)h;
        if ((double)f > h) f = __uint_as_float(__float_as_uint(f) - 1u);       // (h >= 0: f > 0 here)
        sh_hf[q] = (q / k == q % k) ? __builtin_inff() : f;
    });
    __syncthreads();
    const int lane = (int)lane_id();
    const int cl = lane < k ? lane : k - 1;
    // the threshold (upper + S_j)(1 + rel) + abs as ONE float32 fma rounded outwards: upf * E + cumqf
    const float E = 1.0f + 4.76837158e-07f;                 // 1 + 2^-21
    const float cumqf = lane < k ? elk2_f32_up((cumt[cl] * (1.0 + ELK2_EPS_REL) + ELK2_EPS_ABS) * (1.0 + 4.76837158e-07))
                                 : -__builtin_inff();
    float R[ELK2_WIN][ELK2_ROWS];
    const uint32_t wave_off = threadIdx.x & ~63u;
#define ELK2_LOADG(slot, sbase, g)                                                                               \
    _Pragma("unroll") for (int q = 0; q < ELK2_ROWS; q++) {                                                       \
        uint32_t ri = (sbase) + (uint32_t)((g) * ELK2_ROWS + q);                                                 \
        ri = ri < n ? ri : n - 1u;                                                                               \
        const float *rp = A + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ri) * (uint32_t)k;           \
        R[slot][q] = rp[cl];                                                                                     \
    }
    int nlab;
    double nub;
    {
        const uint32_t b0 = blockIdx.x * 256u + wave_off;
        const uint32_t li = b0 + (uint32_t)lane < n ? b0 + (uint32_t)lane : n - 1u;
        nlab = lab[li]; nub = ub[li];
        ELK2_LOADG(0, b0, 0) ELK2_LOADG(1, b0, 1) ELK2_LOADG(2, b0, 2) ELK2_LOADG(3, b0, 3)
    }
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * 256u < n; chunk += gridDim.x) {
        const uint32_t base = chunk * 256u + wave_off;
        const uint32_t nbase = (chunk + gridDim.x) * 256u + wave_off;        // (past the end: clamped loads nobody uses)
        // ---- a lane per sample: the scalars ----
        const uint32_t my_i = base + (uint32_t)lane;
        const int my_lab = nlab;
        const double my_ub = nub + sh_csp[my_lab];         // (t = 1: + 0.0)
        bool open = my_i < n && !(sh_next[my_lab] >= my_ub);
        if (PROBE == 2) open = false;
        const float my_uf = open ? elk2_f32_up(my_ub) : -__builtin_inff();
        // ---- a lane per centre ----
        int c_lo = 0, c_hi = 0, m_lo = 0, m_hi = 0;         // lane s: its sample's masks c (0: not visited) and m0
#define ELK2_FILTG(slot, g)                                                                                      \
        _Pragma("unroll") for (int q = 0; q < ELK2_ROWS; q++) {                                                   \
            const int sq = (g) * ELK2_ROWS + q;                                                                  \
            const int label = __builtin_amdgcn_readlane(my_lab, sq);                                             \
            const float upf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_uf), sq));              \
            const float hq = sh_hf[label * k + cl];                                                              \
            const unsigned long long m0 = __ballot(R[slot][q] < __builtin_fmaf(upf, E, cumqf));                  \
            const unsigned long long c = __ballot(upf > hq) & m0;                                                \
            if (c != 0ull) {                                                                                     \
                const bool mine = lane == sq;                                                                    \
                c_lo = mine ? (int)(uint32_t)c : c_lo;                                                           \
                c_hi = mine ? (int)(uint32_t)(c >> 32) : c_hi;                                                   \
                m_lo = mine ? (int)(uint32_t)m0 : m_lo;                                                          \
                m_hi = mine ? (int)(uint32_t)(m0 >> 32) : m_hi;                                                  \
            }                                                                                                    \
        }
        ELK2_FILTG(0, 0) ELK2_LOADG(0, base, 4)
        ELK2_FILTG(1, 1) ELK2_LOADG(1, base, 5)
        ELK2_FILTG(2, 2) ELK2_LOADG(2, base, 6)
        ELK2_FILTG(3, 3) ELK2_LOADG(3, base, 7)
        {
            const uint32_t li = nbase + (uint32_t)lane < n ? nbase + (uint32_t)lane : n - 1u;
            nlab = lab[li]; nub = ub[li];
        }
        ELK2_FILTG(0, 4) ELK2_LOADG(0, nbase, 0)
        ELK2_FILTG(1, 5) ELK2_LOADG(1, nbase, 1)
        ELK2_FILTG(2, 6) ELK2_LOADG(2, nbase, 2)
        ELK2_FILTG(3, 7) ELK2_LOADG(3, nbase, 3)
        if (PROBE == 0 && my_i < n) {
            const unsigned long long c = ((unsigned long long)(uint32_t)c_hi << 32) | (unsigned long long)(uint32_t)c_lo;
            ub[my_i] = my_ub;
            cmask[my_i] = c;
            if (c != 0ull) mmask[my_i] = ((unsigned long long)(uint32_t)m_hi << 32) | (unsigned long long)(uint32_t)m_lo;
        }
    }
#undef ELK2_LOADG
#undef ELK2_FILTG
}

// k_elk2_visit, a lane per VISITED sample: a workgroup takes 1024 samples at a time, lists those with a mask in
// LDS and runs the reference's scan (_update_chunk_dense) for each listed sample: over c's bits in index order
// (after a relabelling: over the rest of m0), state (label, upper, tightened) per lane, its two tests exact --
// the half distance from LDS, `upper > lb_j` by the bracket of A[i][j] (a 4-byte gather) and, inside the
// bracket, by elk2_exact; distances in _euclidean_dense_dense's operation order.
// (A first version ran the visit with a lane per centre as well -- every distance of a sample at once, the
// scan as one vector round per tightening / relabelling: bit-exact, and 2.2 us per visited sample per
// wavefront, because ~400 wave instructions serve ONE sample: 0.66 ms per E-step while most samples have
// candidates.  The second had filter and visit in one kernel, a lane per sample for the visit: the lanes of
// the few visited samples of a late iteration each held their whole wavefront for a chain of gathers, chunk
// after chunk: 75 us.  Listed, the visits of 1024 samples fill the lanes of one or two wavefronts.)
// *ndiff += labels changed.  diag (optional): [1] samples visited, [2] comparisons decided by elk2_exact.
#define ELK2_VCHUNK 1024u
template <int NBT>
__global__ __launch_bounds__(256) void k_elk2_visit(const double *__restrict__ X, uint32_t n, int nb_arg,
                                                    const double *__restrict__ cen, int k,
                                                    const double *__restrict__ half,
                                                    const double *__restrict__ cumt,
                                                    int32_t *__restrict__ lab, double *__restrict__ ub,
                                                    float *__restrict__ A, uint16_t *__restrict__ stamps,
                                                    const unsigned long long *__restrict__ cmask,
                                                    const unsigned long long *__restrict__ mmask,
                                                    ElkHist hist, uint32_t t, uint32_t *ndiff,
                                                    const uint32_t *stop, unsigned long long *diag)
{
    if (stop && *stop) return;
    const int nb = NBT > 0 ? NBT : nb_arg;
    extern __shared__ double sh2[];
    double *sh_half = sh2, *sh_ct = sh2 + k * k, *sh_cum = sh_ct + k * nb;
    uint32_t *sh_list = (uint32_t *)(sh_cum + k);
    __shared__ uint32_t sh_cnt, sh_changed, sh_exact;
    elk2_stage(half, k * k, [&](int q, double h) { sh_half[q] = h; });
    elk2_stage(cen, k * nb, [&](int q, double c) { const int j = q / nb, b = q - j * nb; sh_ct[b * k + j] = c; });
    for (int q = threadIdx.x; q < k; q += 256) sh_cum[q] = cumt[q];
    if (threadIdx.x == 0) { sh_cnt = 0u; sh_changed = 0u; sh_exact = 0u; }
    __syncthreads();
    const unsigned long long kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    uint32_t n_visit = 0;
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * ELK2_VCHUNK < n; chunk += gridDim.x) {
        // ---- list the chunk's samples that have a mask (a wavefront claims its slots with one LDS atomic) ----
#pragma unroll
        for (uint32_t r = 0; r < ELK2_VCHUNK / 256u; r++) {
            const uint32_t i = chunk * ELK2_VCHUNK + r * 256u + threadIdx.x;
            const bool v = i < n && cmask[i] != 0ull;
            const unsigned long long bal = __ballot(v);
            uint32_t pos = 0;
            if (lane_id() == 0 && bal) pos = atomicAdd(&sh_cnt, (uint32_t)__popcll(bal));
            pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
            if (v) sh_list[pos + (uint32_t)__popcll(bal & lanemask_lt())] = i;
        }
        __syncthreads();
        const uint32_t cnt = sh_cnt;
        n_visit += cnt;
        for (uint32_t e = threadIdx.x; e < cnt; e += 256u) {
            const uint32_t i = sh_list[e];
            unsigned long long mask = cmask[i];
            const unsigned long long m0 = mmask[i];
            int label = lab[i];
            const int lab0 = label;
            double upper = ub[i];
            const double *x = X + (size_t)i * nb;
            double xr[NBT > 0 ? NBT : 1];
            if (NBT > 0) {
#pragma unroll
                for (int b = 0; b < NBT; b++) xr[b] = x[b];
            }
            // the distance to centre j, in _euclidean_dense_dense's operation order (elk_dist)
            auto dist_to = [&](int j) -> double {
                const int n4 = nb / 4, rem = nb % 4;
                double result = 0.0;
                int b = 0;
                for (int q = 0; q < n4; q++, b += 4) {
                    const double d0 = (NBT > 0 ? xr[NBT > 0 ? b : 0] : x[b]) - sh_ct[b * k + j];
                    const double d1 = (NBT > 0 ? xr[NBT > 0 ? b + 1 : 0] : x[b + 1]) - sh_ct[(b + 1) * k + j];
                    const double d2 = (NBT > 0 ? xr[NBT > 0 ? b + 2 : 0] : x[b + 2]) - sh_ct[(b + 2) * k + j];
                    const double d3 = (NBT > 0 ? xr[NBT > 0 ? b + 3 : 0] : x[b + 3]) - sh_ct[(b + 3) * k + j];
                    result += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
                }
                for (int q = 0; q < rem; q++, b++) {
                    const double d = (NBT > 0 ? xr[NBT > 0 ? b : 0] : x[b]) - sh_ct[b * k + j];
                    result += d * d;
                }
                return __builtin_sqrt(result);
            };
            bool tight = false;
            float *Ai = A + (size_t)i * k;
            uint16_t *Si = stamps + (size_t)i * k;
            uint32_t n_exact = 0;
            while (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                if (j == label) continue;
                const double hj = sh_half[label * k + j];
                if (!(upper > hj)) continue;
                // `upper > lb_j`: the bracket of A[i][j]; inside it, the bound itself
                const double cj = sh_cum[j];
                const float af = Ai[j];
                const float afu = af < 0.0f ? af : __uint_as_float(__float_as_uint(af) + 1u);
                const double eps = ELK2_EPS_ABS + ELK2_EPS_REL * ((double)afu < 0.0 ? 0.0 : (double)afu);
                double L = (double)af - cj - eps, U = (double)afu - cj + eps;
                if (!(L > 0.0)) L = 0.0;
                if (!(U > 0.0)) U = 0.0;
                bool exact_known = false;
                double lbj = 0.0;
                bool above;
                if (upper > U) above = true;
                else if (!(upper > L)) above = false;
                else { lbj = elk2_exact(X, nb, k, i, j, t, stamps, hist); exact_known = true; above = upper > lbj; n_exact++; }
                if (!above) continue;
                if (!tight) {
                    const double was = upper;
                    upper = dist_to(label);
                    Ai[label] = elk2_f32_down(upper + sh_cum[label]);
                    Si[label] = (uint16_t)t;
                    tight = true;
                    if (upper > was) mask = kmask & ~(j >= 63 ? ~0ull : ((2ull << j) - 1ull));     // a rounding above the bound it replaces
                    // the second test sees the tightened bound
                    if (upper > U) above = true;
                    else if (!(upper > L)) above = false;
                    else {
                        if (!exact_known) { lbj = elk2_exact(X, nb, k, i, j, t, stamps, hist); n_exact++; }
                        above = upper > lbj;
                    }
                }
                if (above || upper > hj) {
                    const double dist = dist_to(j);
                    Ai[j] = elk2_f32_down(dist + cj);
                    Si[j] = (uint16_t)t;
                    if (dist < upper) {             // the new label's half distances judge the rest: back to the superset
                        label = j; upper = dist;
                        mask = m0 & ~(j >= 63 ? ~0ull : ((2ull << j) - 1ull));
                    }
                }
            }
            if (tight) ub[i] = upper;
            if (label != lab0) { lab[i] = label; atomicAdd(&sh_changed, 1u); }
            if (n_exact) atomicAdd(&sh_exact, n_exact);
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_cnt = 0u;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (sh_changed) atomicAdd(ndiff, sh_changed);
        if (diag) { atomicAdd(&diag[1], (unsigned long long)n_visit); if (sh_exact) atomicAdd(&diag[2], (unsigned long long)sh_exact); }
    }
}
