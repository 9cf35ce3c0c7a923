// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "dev_env.hpp"
#include "tiatoolbox_amd.h"

#ifndef TIA_UNIFORM
#define TIA_UNIFORM 1
#endif

namespace tia {

constexpr int kWave = 64;

// A batch whose input exceeds the kernels' 32-bit byte offsets goes in groups of at most `max_group` images.  Equal groups, not
// "full groups and a remainder": 4096 images at 2047 per group are 1366 + 1366 + 1364, not 2047 + 2047 + 2 -- a two-image
// launch runs on whichever kernel serves tiny batches and costs a launch of its own (measured in the bench trace: 0.5 % of a step).
inline long even_group(long n, long max_group) {
    if (max_group < 1 || n <= max_group) return max_group;
    const long k = (n + max_group - 1) / max_group;
    return (n + k - 1) / k;
}

// Per-device one-time host set-up (hipFuncSetAttribute is a per-DEVICE property of a kernel; a process-wide `static bool`
// would leave the second GPU of a process without it).  `ensure(setup)` runs `setup()` -> bool until it has succeeded once on
// the calling thread's current device; concurrent first calls may both run it (the set-ups are idempotent), the flag itself
// is atomic.
struct DeviceOnce {
    std::atomic<unsigned char> done[64] = {};
    template <class F>
    bool ensure(F&& setup) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
        if (done[d].load(std::memory_order_acquire)) return true;
        if (!setup()) return false;
        done[d].store(1, std::memory_order_release);
        return true;
    }
};

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

// LDS histogram increment that survives value spikes (white background, saturated pixels).
// `WaveGroup` says whether every active lane of the wave is looking at the same 12 bytes in this
// iteration (one ballot per 4 pixels); if so one lane adds the population count instead of 64
// serialised same-address atomics.
struct WaveGroup {
    bool uniform;
    int leader;
    unsigned count;
};
__device__ __forceinline__ void hist_add(unsigned* bins, int bin, bool valid, const WaveGroup& wg) {
    if (!valid) return;
    if (wg.uniform) {
        if (lane_id() == wg.leader) atomicAdd(&bins[bin], wg.count);
    } else {
        atomicAdd(&bins[bin], 1u);
    }
}

// Visit every pixel of one HWC uint8 image with `f(idx, r, g, b)`; block-strided.
// Fast path reads 12 contiguous bytes (4 pixels) per lane as three dwords: lanes of a wave
// cover 768 contiguous bytes per iteration (coalesced), and pixels never straddle lanes.
// The next group's dwords are requested before the current group is processed (software
// prefetch), so the L2/HBM latency hides behind the per-pixel arithmetic even at 2-4 waves/SIMD.
template <int NT, class F>
__device__ __forceinline__ void for_each_pixel(const uint8_t* __restrict__ p, long hw, F&& f) {
    if (((hw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
        const long ng = hw >> 2;
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
        long g = threadIdx.x;
        uint32_t a = 0, b = 0, c = 0;
        if (g < ng) {
            a = q[g * 3 + 0];
            b = q[g * 3 + 1];
            c = q[g * 3 + 2];
        }
        while (g < ng) {
            const long gn = g + NT;
            uint32_t na = 0, nb = 0, nc = 0;
            if (gn < ng) {
                na = q[gn * 3 + 0];
                nb = q[gn * 3 + 1];
                nc = q[gn * 3 + 2];
            }
            f(g * 4 + 0, a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
            f(g * 4 + 1, a >> 24, b & 255u, (b >> 8) & 255u);
            f(g * 4 + 2, (b >> 16) & 255u, b >> 24, c & 255u);
            f(g * 4 + 3, (c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
            a = na;
            b = nb;
            c = nc;
            g = gn;
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += NT) {
            f(i, (uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2]);
        }
    }
}

// Same sweep, additionally telling `f` whether the wave's lanes all hold identical pixels.
template <int NT, class F>
__device__ __forceinline__ void for_each_pixel_w(const uint8_t* __restrict__ p, long hw, F&& f) {
    if (((hw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
        const long ng = hw >> 2;
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
        long g = threadIdx.x;
        uint32_t a = 0, b = 0, c = 0;
        if (g < ng) {
            a = q[g * 3 + 0];
            b = q[g * 3 + 1];
            c = q[g * 3 + 2];
        }
        while (g < ng) {
            const long gn = g + NT;
            uint32_t na = 0, nb = 0, nc = 0;
            if (gn < ng) {
                na = q[gn * 3 + 0];
                nb = q[gn * 3 + 1];
                nc = q[gn * 3 + 2];
            }
#ifndef TIA_UNIFORM
#define TIA_UNIFORM 1
#endif
            WaveGroup wg{false, 0, 1u};
            if (TIA_UNIFORM) {
                const unsigned long long act = __ballot(1);
                wg.leader = __ffsll((long long)act) - 1;
                const uint32_t a0 = __builtin_amdgcn_readlane(a, wg.leader);
                const uint32_t b0 = __builtin_amdgcn_readlane(b, wg.leader);
                const uint32_t c0 = __builtin_amdgcn_readlane(c, wg.leader);
                wg.uniform = __ballot(a == a0 && b == b0 && c == c0) == act;
                wg.count = (unsigned)__popcll(act);
            }
            f(g * 4 + 0, a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, wg);
            f(g * 4 + 1, a >> 24, b & 255u, (b >> 8) & 255u, wg);
            f(g * 4 + 2, (b >> 16) & 255u, b >> 24, c & 255u, wg);
            f(g * 4 + 3, (c >> 8) & 255u, (c >> 16) & 255u, c >> 24, wg);
            a = na;
            b = nb;
            c = nc;
            g = gn;
        }
    } else {
        WaveGroup wg{false, 0, 1u};
        for (long i = threadIdx.x; i < hw; i += NT) {
            f(i, (uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2], wg);
        }
    }
}

// Group-level sweep for straight-line kernels: `f(g, a, b, c, wg)` receives the three dwords holding
// pixels 4g..4g+3 (bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3).  Requires hw % 4 == 0 and a
// 4-byte aligned image (check with `groups_ok`).  Next group's dwords are prefetched.
__device__ __forceinline__ bool groups_ok(const uint8_t* p, long hw) {
    return ((hw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 3) == 0);
}
template <int NT, class F>
__device__ __forceinline__ void for_each_group(const uint8_t* __restrict__ p, long hw, F&& f) {
    const long ng = hw >> 2;
    const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
    long g = threadIdx.x;
    uint32_t a = 0, b = 0, c = 0;
    if (g < ng) {
        a = q[g * 3 + 0];
        b = q[g * 3 + 1];
        c = q[g * 3 + 2];
    }
    while (g < ng) {
        const long gn = g + NT;
        uint32_t na = 0, nb = 0, nc = 0;
        if (gn < ng) {
            na = q[gn * 3 + 0];
            nb = q[gn * 3 + 1];
            nc = q[gn * 3 + 2];
        }
        WaveGroup wg{false, 0, 1u};
        if (TIA_UNIFORM) {
            const unsigned long long act = __ballot(1);
            wg.leader = __ffsll((long long)act) - 1;
            const uint32_t a0 = __builtin_amdgcn_readlane(a, wg.leader);
            const uint32_t b0 = __builtin_amdgcn_readlane(b, wg.leader);
            const uint32_t c0 = __builtin_amdgcn_readlane(c, wg.leader);
            wg.uniform = __ballot(a == a0 && b == b0 && c == c0) == act;
            wg.count = (unsigned)__popcll(act);
        }
        f(g, a, b, c, wg);
        a = na;
        b = nb;
        c = nc;
        g = gn;
    }
}
// channel bytes of the four pixels of a group
__device__ __forceinline__ void unpack_group(uint32_t a, uint32_t b, uint32_t c, uint32_t (&r)[4], uint32_t (&g)[4],
                                             uint32_t (&bl)[4]) {
    r[0] = a & 255u;          g[0] = (a >> 8) & 255u;   bl[0] = (a >> 16) & 255u;
    r[1] = a >> 24;           g[1] = b & 255u;          bl[1] = (b >> 8) & 255u;
    r[2] = (b >> 16) & 255u;  g[2] = b >> 24;           bl[2] = c & 255u;
    r[3] = (c >> 8) & 255u;   g[3] = (c >> 16) & 255u;  bl[3] = c >> 24;
}

// ---- LDS-resident labelling of small planes (imgops.hip; internal interface shared with hover_post.hip) ---------------------
constexpr long kCclTileMaxPixels = 36864;  // int32 union-find of one plane in LDS: 147,456 of the CU's 163,840 bytes
// Whether the tile-resident (LDS) kernels serve the current device: false with the developer switch TIA_NO_CCL_TILE=1 (parity
// audit of the multi-launch path) and on a device that refuses their dynamic LDS size (up to 147 KB; gfx950 has 160 KB per
// CU) -- the callers then take the multi-launch path instead of failing.  Sets the kernels' LDS attribute once per device.
bool ccl_tile_enabled();
// labels (1-based, raster order of the components' first pixels; 0 = background or removed) of n planes of h x w <= 36,864
// pixels in one launch.  src_kind 0: uint8 mask != 0; 1: uint8 mask == 0; 2: float32 map >= 0.5.  min_keep > 0: components with
// fewer pixels become 0 (their numbers are not re-used); areas (nullable): [n][h*w + 1] component areas by label.
// offs / bbox (nullable): heap-segment offsets and bounding-box reset of the labels (HoVer-Net's blob stage)
int ccl_tile_label(const void* src, int src_kind, long n, int h, int w, int conn, int min_keep, int* labels, int* count, int* areas,
                   hipStream_t st, int* offs = nullptr, int* bbox = nullptr);
int fill_holes_tile(const uint8_t* mask, long n, int h, int w, uint8_t* out, hipStream_t st);
// HoVer-Net's marker pipeline (fill holes -> 5x5 elliptical opening -> label -> area filter) of n planes in one launch;
// union-find + one byte plane in LDS: h * w <= 32,000
constexpr long kMarkerTileMaxPixels = 32000;
// blob / inst / bbox (nullable together): also write the watershed's initial state (inst, blob bounding boxes)
int marker_tile(const uint8_t* marker0, long n, int h, int w, int min_keep, int* labels, int* count, int* areas, hipStream_t st,
                const int* blob = nullptr, int* inst = nullptr, int* bbox = nullptr);

}  // namespace tia
