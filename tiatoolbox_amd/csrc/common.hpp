// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tiatoolbox_amd.h"

namespace tia {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Order-preserving map f64 -> u64 (total order, -0 < +0), and back.
__device__ __forceinline__ unsigned long long f64_key(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_down(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned w = __shfl_up(v, o, 64);
        if (lane_id() >= o) v += w;
    }
    return v;
}

// Visit every pixel of one HWC uint8 image with `f(idx, r, g, b)`; block-strided.
// Fast path reads 12 contiguous bytes (4 pixels) per lane as three dwords: lanes of a wave
// cover 768 contiguous bytes per iteration (coalesced), and pixels never straddle lanes.
template <int NT, class F>
__device__ __forceinline__ void for_each_pixel(const uint8_t* __restrict__ p, long hw, F&& f) {
    if (((hw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
        const long ng = hw >> 2;
        for (long g = threadIdx.x; g < ng; g += NT) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + g * 12);
            const uint32_t a = q[0], b = q[1], c = q[2];
            f(g * 4 + 0, a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
            f(g * 4 + 1, a >> 24, b & 255u, (b >> 8) & 255u);
            f(g * 4 + 2, (b >> 16) & 255u, b >> 24, c & 255u);
            f(g * 4 + 3, (c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += NT) {
            f(i, (uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2]);
        }
    }
}

}  // namespace tia
