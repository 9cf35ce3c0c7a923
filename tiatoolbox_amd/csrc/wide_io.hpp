// 16-byte global accesses for HWC uint8 pixel streams (gfx950): a wave takes 3072 contiguous bytes (1024 RGB pixels) per step as
// three coalesced 16 B/lane loads, and a wave-private 3 KB LDS region turns "lane l holds bytes 16l.." into "lane l holds its 16
// pixels' 48 contiguous bytes" (the 12-dword lane stride makes the three ds_read_b128 conflict-free).  12-byte loads / 4-byte stores
// plateau far below what 16-byte accesses reach on MI355X (MI355X_MICROARCH.md: 8-B accesses run at 0.54-0.70x the 16-B rate);
// every streaming kernel over RGB bytes goes through this header.
#pragma once
#include "common.hpp"

namespace tia {

using v4u = __attribute__((ext_vector_type(4))) unsigned;
constexpr int kRgbChunk = 3072;  // bytes per wave step
constexpr int kPxChunk = 1024;   // pixels per wave step

// The hand-off region is written and read by different lanes of one wave: the compiler has to treat the barrier as a memory
// clobber (in-thread alias analysis alone would let it hoist the read-backs above the stores).  Wavefront scope: no cache
// maintenance, no extra waits (LDS operations of a wave execute in order).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RgbChunk {
    v4u in[3];
};

// issue the three loads of the chunk at `src` (16-byte aligned); nothing waits here: call it one step ahead
__device__ __forceinline__ void rgb_chunk_issue(RgbChunk& c, const uint8_t* __restrict__ src) {
    const v4u* g = reinterpret_cast<const v4u*>(src);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; ++k) c.in[k] = __builtin_nontemporal_load(g + k * 64 + lane);
}
// same, through the caches (data that is read again soon)
__device__ __forceinline__ void rgb_chunk_issue_cached(RgbChunk& c, const uint8_t* __restrict__ src) {
    const v4u* g = reinterpret_cast<const v4u*>(src);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; ++k) c.in[k] = g[k * 64 + lane];
}

// lane's 16 pixels as 12 dwords (w[3q..3q+2] = pixels 4q..4q+3: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3); `mine` = the wave's
// 3072-byte LDS region (16-byte aligned).  The region is free again on return.
__device__ __forceinline__ void rgb_chunk_transpose(const RgbChunk& c, uint8_t* mine, uint32_t (&w)[12]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<v4u*>(mine + k * 1024 + lane * 16) = c.in[k];
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const v4u t = *reinterpret_cast<const v4u*>(mine + lane * 48 + j * 16);
        w[j * 4 + 0] = t.x;
        w[j * 4 + 1] = t.y;
        w[j * 4 + 2] = t.z;
        w[j * 4 + 3] = t.w;
    }
    wave_lds_fence();
}

// the reverse: lane's 48 result bytes (12 dwords, same layout) -> three coalesced 16 B/lane stores at `dst`
template <bool NT = true>
__device__ __forceinline__ void rgb_chunk_store(const uint32_t (&w)[12], uint8_t* mine, uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v4u t;
        t.x = w[j * 4 + 0];
        t.y = w[j * 4 + 1];
        t.z = w[j * 4 + 2];
        t.w = w[j * 4 + 3];
        *reinterpret_cast<v4u*>(mine + lane * 48 + j * 16) = t;
    }
    wave_lds_fence();
    v4u* g = reinterpret_cast<v4u*>(dst);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const v4u t = *reinterpret_cast<const v4u*>(mine + k * 1024 + lane * 16);
        if constexpr (NT) __builtin_nontemporal_store(t, g + k * 64 + lane);
        else g[k * 64 + lane] = t;
    }
    wave_lds_fence();
}

// the four pixels of a 3-dword group, each as a dword whose low three bytes are r, g, b (top byte: junk from the neighbour)
__device__ __forceinline__ void group_pixels(uint32_t a, uint32_t b, uint32_t c, uint32_t (&p)[4]) {
    p[0] = a;
    p[1] = __builtin_amdgcn_alignbyte(b, a, 3);
    p[2] = __builtin_amdgcn_alignbyte(c, b, 2);
    p[3] = c >> 8;
}

// cv2.cvtColor(COLOR_RGB2GRAY), 8-bit: (R*9798 + G*19235 + B*3735 + 2^14) >> 15 of a pixel dword (r | g << 8 | b << 16 | junk << 24)
// as two byte dot products (coefficients split into high and low bytes; the fourth coefficient is zero)
__device__ __forceinline__ uint32_t gray_of_px(uint32_t px) {
    constexpr uint32_t kHi = (9798u >> 8) | ((19235u >> 8) << 8) | ((3735u >> 8) << 16);
    constexpr uint32_t kLo = (9798u & 255u) | ((19235u & 255u) << 8) | ((3735u & 255u) << 16);
    const uint32_t hi = __builtin_amdgcn_udot4(px, kHi, 0u, false);
    const uint32_t lo = __builtin_amdgcn_udot4(px, kLo, 1u << 14, false);
    return ((hi << 8) + lo) >> 15;
}

}  // namespace tia
