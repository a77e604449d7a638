// synth.h -- `synthimg v1`: integer-only synthetic imagery, deterministic in
// (seed, band, y, x) so any window can be generated independently, bit-identically on host or
// device.  Used only to make benchmark / test inputs on the device.
#pragma once
#include "common.h"

__device__ __forceinline__ uint64_t syn_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t syn_h(uint64_t seed, uint64_t b, uint64_t o, uint64_t y, uint64_t x)
{
    const uint64_t P = 1000003ull;
    return syn_mix((((seed * P + b) * P + o) * P + y) * P + x);
}
__device__ __forceinline__ uint16_t syn_pixel(uint64_t seed, uint64_t b, uint64_t y, uint64_t x)
{
    uint64_t v = 1000 + 300 * b;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int lg = 8 - 2 * o;
        const uint64_t amp = 2000ull >> o;
        const uint64_t c = 1ull << lg;
        const uint64_t gy = y >> lg, gx = x >> lg, fy = y & (c - 1), fx = x & (c - 1);
        const uint64_t v00 = syn_h(seed, b, o, gy, gx) >> 48;
        const uint64_t v01 = syn_h(seed, b, o, gy, gx + 1) >> 48;
        const uint64_t v10 = syn_h(seed, b, o, gy + 1, gx) >> 48;
        const uint64_t v11 = syn_h(seed, b, o, gy + 1, gx + 1) >> 48;
        const uint64_t interp = (v00 * (c - fy) * (c - fx) + v01 * (c - fy) * fx +
                                 v10 * fy * (c - fx) + v11 * fy * fx) >> (2 * lg);
        v += (interp * amp) >> 16;
    }
    v += ((syn_h(seed, b, 99, y, x) >> 48) * 120) >> 16;
    return (uint16_t)(v > 65534 ? 65534 : v);
}

// out: band-planar (nbands, nrows, ncols) window starting at (y0, x0)
__global__ __launch_bounds__(256) void k_synthimg(uint64_t seed, int nbands, int64_t y0, int64_t x0,
                                                  uint32_t nrows, uint32_t ncols,
                                                  uint16_t *__restrict__ out)
{
    const size_t npix = (size_t)nrows * ncols;
    const size_t total = npix * nbands;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (size_t)gridDim.x * 256u) {
        const size_t b = i / npix, p = i - b * npix;
        const size_t r = p / ncols, c = p - r * ncols;
        out[i] = syn_pixel(seed, b, (uint64_t)(y0 + (int64_t)r), (uint64_t)(x0 + (int64_t)c));
    }
}
