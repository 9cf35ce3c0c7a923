// The ResNet stem in ONE kernel on the gfx950 matrix cores: uint8 (or float32) NHWC patches in, pooled float32 NHWC out.
//
//   y = maxpool3x3/s2/p1( relu( conv7x7/s2/p3( x / 255 ) + bias ) )        3 -> 64 channels, BatchNorm folded into w / bias
//
// Reference: CNNModel.forward -> torchvision resnet conv1 / bn1 / relu / maxpool behind `ToTensor`
// (models/architecture/vanilla.py:242-245, 300-316; models/dataset/classification.py:27-32: uint8 HWC -> float32 / 255).
// Arithmetic: float32 throughout; the division by 255 is the correctly rounded one (a 256-entry table built with IEEE
// division, = torch's `.float().div(255)`), the convolution is an fmaf chain on v_mfma_f32_32x32x2_f32 in the order
// (ky, kx, c), bias added after the sum, then max(., 0), then the 3x3 maximum -- the order of the unfused torch ops.
//
// GEMM view per workgroup iteration: M = 2 conv rows x 128 conv columns (8 MFMA tiles of 32 pixels), N = 64 channels
// (2 tiles), K = 7 * 21 = 147 (+ 1 zero row).  A workgroup (4 waves, one column group of 32 conv columns each) walks down
// a strip of an image two conv rows = one pooled row at a time:
//   * the 9 input rows a pair of conv rows needs sit in LDS as float32 (converted ONCE per workgroup iteration, pad
//     columns / rows as zeros); MFMA lane (i, h) reads pixel column i, reduction index 2 s + h: with the reduction ordered
//     (ky, kx, c) a tap row is 21 consecutive floats, so the A operand is `ring[6 i + const]` -- one ds_read_b32 with an
//     immediate offset and NO address arithmetic in the loop (three "wrap" steps where h = 0 / 1 straddle two tap rows use
//     a second per-lane base); B (148 x 64 weights) stays in LDS for the whole kernel
//   * each wave owns the SAME 32 columns of both conv rows, so the vertical part of the 3x3 maximum is register-local:
//     V = max(previous iteration's second row, row 2p, row 2p+1); only V goes to LDS (aliasing the input rows, which are
//     dead by then) for the horizontal maximum of three columns, written out as whole 16 KB pooled rows
//   * the next iteration's input bytes are requested before the 296 MFMAs of the current one and converted afterwards
//   * LDS 70 KB -> two workgroups per CU: one converts / pools while the other multiplies
// Wider images are cut into column strips of <= 128 conv columns (64 / 63 pooled columns), taller ones into row chunks
// (one extra warm-up iteration per chunk supplies the carried row).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tiatoolbox_amd.h"
#include "common.hpp"

namespace {

constexpr int NTH = 256;
constexpr int RS = 784;          // floats per staged input row: (2 * 128 + 5) pixels * 3 = 783, + 1 (read by the zero k row)
constexpr int WROWS = 9;         // input rows under two conv rows: 2 * 2 + 5
constexpr int KROWS = 148;       // 147 taps*channels + one zero row (the MFMA reduces two k per step)
constexpr int COUT = 64;
constexpr int REGION = 128 * 64;  // floats: V tile [128 conv columns][64 channels]  (>= WROWS * RS = 7056)
constexpr int LUT_OFF = REGION;
constexpr int W_OFF = REGION + 256;
constexpr int LDS_FLOATS = W_OFF + KROWS * COUT;

// half-precision matrix-core variant (MMA = 1: fp16, 2: bf16; `compute_dtype="float16" | "bfloat16"` of the engines): the staged
// rows and the weights are halves (x / 255 rounded to half = what `model.half()(ToTensor(x).half())` feeds its first convolution),
// the reduction is ordered (ky, 24-padded kx * 3 + c): 7 * 24 = 168 -> 11 k-steps of v_mfma_f32_32x32x16 (float32 accumulate);
// an MFMA lane's 8 k-values are the 8 consecutive halves ring16[ky][6 lc + {0, 8, 16} ..] (dword-aligned: four ds_read_b32)
constexpr int RS16 = 792;         // halves per staged row: 198 units of 4 (783 used; the k-padding reads up to element 785)
constexpr int KH = 176;           // 7 * 24 = 168, padded to whole k-steps of 16
constexpr int LUT16_OFF_B = REGION * 4;                 // byte offsets in the half variant's LDS
constexpr int W16_OFF_B = LUT16_OFF_B + 512;
constexpr int LDS_BYTES_H = W16_OFF_B + KH * COUT * 2;  // 55,808 B

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8v = __attribute__((ext_vector_type(8))) _Float16;
using b8v = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int MMA>
__device__ __forceinline__ unsigned short to_half_bits(float x) {  // round to nearest even (finite inputs)
    if constexpr (MMA == 2) {
        unsigned u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    } else {
        const _Float16 h = (_Float16)x;
        unsigned short v;
        __builtin_memcpy(&v, &h, 2);
        return v;
    }
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains vmcnt, i.e. it makes
// every wave wait for the acknowledgements of the pooled-row stores it has just issued; the loop's barriers only protect the LDS
// regions (input rows <-> V tile), and a wave's own global loads are consumed through registers (the compiler's vmcnt waits).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Phase timing (developer builds only: -DTIA_STEM_TIMING=1, build.build(defines=...)): thread 0 of two workgroups prints the mean
// shader-clock cycles per loop iteration of: request + MFMA loop, barrier, V tile, horizontal maximum + stores, conversion.
#ifndef TIA_STEM_TIMING
#define TIA_STEM_TIMING 0
#endif
#if TIA_STEM_TIMING
#define SSTAMP(i) { const long long now_ = clock64(); if (threadIdx.x == 0) tm_[i] += now_ - tl_; tl_ = now_; }
#else
#define SSTAMP(i)
#endif

struct StemDims {
    int n, h, w, ho, wo, hp, wp;
    int chunks, rows_per_chunk;  // row chunks per image, pooled rows per chunk
    unsigned x_bytes;            // extent of the input buffer of this launch (< 2^31), rounded up to whole dwords
    int x_shift;                 // uint8 input: bytes between the (dword-aligned) buffer base and the first image
    int out_dtype;               // TIA_DT_F32 | TIA_DT_F16 | TIA_DT_BF16: type of the pooled output (arithmetic is float32 either way)
};

template <bool U8, int MMA = 0>
__global__ __launch_bounds__(NTH, 2) void stem7x7_pool_kernel(const void* __restrict__ xin, const void* __restrict__ wpk_v,
                                                             const float* __restrict__ bias, void* __restrict__ yout, float* __restrict__ yconv, StemDims d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool HALF = MMA != 0;
    constexpr int UPR = HALF ? 198 : 196;  // 4-element staging units per row
    constexpr int RSE = HALF ? RS16 : RS;  // elements per staged row
    const float* wpk = static_cast<const float*>(wpk_v);
    float* ring = smem;
    unsigned short* ring16 = reinterpret_cast<unsigned short*>(smem);
    float* lut = smem + LUT_OFF;
    unsigned short* lut16 = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(smem) + LUT16_OFF_B);
    float* Wl = smem + W_OFF;
    const u32x4* Wh = reinterpret_cast<const u32x4*>(reinterpret_cast<unsigned char*>(smem) + W16_OFF_B);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x / d.chunks, chunk = blockIdx.x - img * d.chunks;
    const int strip = blockIdx.y;
    const int p0 = strip == 0 ? 0 : 64 + 63 * (strip - 1);
    const int p1 = min(d.wp, strip == 0 ? 64 : p0 + 63);
    const int c_start = max(0, 2 * p0 - 1);
    const int ncols = min(d.wo, 2 * p1) - c_start;  // conv columns [c_start, c_start + ncols), <= 128
    const int ixlo = 2 * c_start - 3;               // input pixel column under ring float 0
    const int ixlo_c = max(ixlo, 0);
    const int f0 = (ixlo_c - ixlo) * 3;                              // ring float of the first in-image element of a row
    const int nb = max(0, (min(ixlo + 261, d.w) - ixlo_c) * 3);      // in-image elements of a row that the strip needs
    const int q0 = chunk * d.rows_per_chunk, q1 = min(d.hp, q0 + d.rows_per_chunk);
    if (q0 >= q1) return;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xin), 0, (int)d.x_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    // ---- weights and the /255 table into LDS (once) ----
    if constexpr (HALF) {
        for (int i = tid; i < KH * COUT * 2 / 16; i += NTH) const_cast<u32x4*>(Wh)[i] = static_cast<const u32x4*>(wpk_v)[i];
        lut16[tid] = to_half_bits<MMA>(__fdiv_rn((float)tid, 255.0f));
    } else {
        for (int i = tid; i < KROWS * COUT / 4; i += NTH)
            reinterpret_cast<float4*>(Wl)[i] = reinterpret_cast<const float4*>(wpk)[i];
        lut[tid] = __fdiv_rn((float)tid, 255.0f);
    }

    // ---- staging of the 9 input rows of pooled row `py`: request (registers), later convert + write (LDS) ----
    // uint8: a unit = four consecutive ring floats (one 16-byte LDS store) = four consecutive input bytes, which lie in two
    // aligned dwords (funnel-shifted together); 9 rows x 196 units.  float32: a unit = one ring float.
    constexpr int NU = U8 ? 7 : 28;  // 7 * 256 >= 9 * 198 units; 28 * 256 >= 9 * 792 elements
    unsigned ld[NU], ld2[U8 ? NU : 1];
    auto issue_loads = [&](int py) {
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            const int u = tid + NTH * q;
            if constexpr (U8) {
                const int wr = u / UPR, g = u - wr * UPR;
                const int iy = 4 * py - 3 + wr;
                const bool ok = wr < WROWS && iy >= 0 && iy < d.h;
                const unsigned gb = (unsigned)(((img * d.h + iy) * d.w + ixlo_c) * 3 + d.x_shift);
                const int t = (int)(gb & 3u) + 4 * g - f0;           // byte distance from the row's aligned base
                const int voff = (int)(gb & ~3u) + ((t >> 2) << 2);  // may lie in front of the buffer: reads as zero
                ld[q] = __builtin_amdgcn_raw_buffer_load_b32(rx, (ok && voff >= 0) ? voff : OOB, 0, 0);
                ld2[q] = __builtin_amdgcn_raw_buffer_load_b32(rx, (ok && voff + 4 >= 0) ? voff + 4 : OOB, 0, 0);
            } else {
                const int wr = u / RSE, ri = u - wr * RSE;
                const int iy = 4 * py - 3 + wr, bi = ri - f0;
                const bool ok = wr < WROWS && iy >= 0 && iy < d.h && bi >= 0 && bi < nb;
                const int voff = (((img * d.h + iy) * d.w + ixlo_c) * 3 + bi) * 4;
                ld[q] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? voff : OOB, 0, 0);
            }
        }
    };
    // uint8 units: which of a unit's four bytes are image bytes of ITS row (the others belong to the neighbouring rows of the flat
    // NHWC buffer, or to the padding columns) and the funnel-shift amount do not change from one pooled row to the next (the row
    // base moves by 4 * w * 3 bytes): one byte mask and one shift per unit, computed once.  A masked byte reads table entry 0 = 0.0,
    // and rows outside the image were loaded as zeros -- the conversion needs no per-element predicate.
    unsigned cmask[U8 ? NU : 1], shv = 0;
    if constexpr (U8) {
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            const int u = tid + NTH * q;
            const int wr = u / UPR, g = u - wr * UPR;
            const int bi = 4 * g - f0;
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) m |= (bi + e >= 0 && bi + e < nb) ? 0xffu << (8 * e) : 0u;
            cmask[q] = m;
            const unsigned gb = (unsigned)(((img * d.h + (wr - 3)) * d.w + ixlo_c) * 3 + d.x_shift);  // row base at py = 0 (mod 4: any py)
            shv |= (((gb & 3u) + 4u * g - (unsigned)f0) & 3u) << (2 * q);
        }
    }
    auto write_ring = [&](int py) {
        if constexpr (U8) {
            (void)py;
            float f[NU][4];
            unsigned short hq[NU][4];
#pragma unroll
            for (int q = 0; q < NU; ++q) {  // all table look-ups first (independent), then the stores
                const unsigned wbytes = __builtin_amdgcn_alignbyte(ld2[q], ld[q], (shv >> (2 * q)) & 3u) & cmask[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (HALF) hq[q][e] = lut16[(wbytes >> (8 * e)) & 255u];
                    else f[q][e] = lut[(wbytes >> (8 * e)) & 255u];
                }
            }
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const int u = tid + NTH * q;
                if constexpr (HALF) {
                    if (u < WROWS * UPR)  // ring half 4 u = row wr, half 4 g
                        reinterpret_cast<uint2*>(ring16)[u] = make_uint2((unsigned)hq[q][0] | ((unsigned)hq[q][1] << 16),
                                                                         (unsigned)hq[q][2] | ((unsigned)hq[q][3] << 16));
                } else {
                    if (u < WROWS * UPR) reinterpret_cast<float4*>(ring)[u] = float4{f[q][0], f[q][1], f[q][2], f[q][3]};  // row wr, float 4 g
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const int u = tid + NTH * q;
                if (u < WROWS * RSE) {  // out-of-image slots were loaded as zeros
                    if constexpr (HALF) ring16[u] = to_half_bits<MMA>(__uint_as_float(ld[q]));
                    else ring[u] = __uint_as_float(ld[q]);
                }
            }
        }
    };

    f32x16 carry[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) carry[0][e] = carry[1][e] = 0.0f;

    const int hh = lane >> 5;
    const int lc = 32 * wave + (lane & 31);  // local conv column of this lane's A rows
    const float* a_ptr = ring + 6 * lc + hh;
    const float* aw_ptr = ring + 6 * lc + hh * (RS - 20);  // wrap steps: h = 1 starts the next tap row
    const float* w_ptr = Wl + hh * COUT + (lane & 31);
    const float bv0 = bias[lane & 31], bv1 = bias[32 + (lane & 31)];

    const int it0 = q0 > 0 ? q0 - 1 : 0;
    issue_loads(it0);
    __syncthreads();  // lut ready
    write_ring(it0);
    __syncthreads();

#if TIA_STEM_TIMING
    long long tm_[6] = {0, 0, 0, 0, 0, 0}, tl_ = clock64();
    const long long t0c_ = tl_, t0w_ = wall_clock64();  // shader clock vs the constant 100 MHz clock: the sustained frequency
#endif
    for (int py = it0; py < q1; ++py) {
        SSTAMP(5)
        issue_loads(py + 1 < q1 ? py + 1 : py);  // next window's bytes fly behind the MFMAs (the last one re-reads its own)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.0f;
        if constexpr (HALF) {
            // k-step s, lane half h: k0 = 16 s + 8 h = 24 ky + j0 with j0 in {0, 8, 16}; h = 1 is "+8 halves" of h = 0 except
            // when s % 3 == 1, where it is the start of the next tap row (second base pointer, like the wrap steps above)
            const unsigned* pa = reinterpret_cast<const unsigned*>(ring16 + 6 * lc + 8 * hh);
            const unsigned* pb = reinterpret_cast<const unsigned*>(ring16 + 6 * lc + hh * (RS16 - 16));
            const u32x4* wv = Wh + hh * COUT + (lane & 31);
#pragma unroll
            for (int s = 0; s < KH / 16; ++s) {
                const int k0 = 16 * s, ky = k0 / 24, j0 = k0 - 24 * ky;       // of the h = 0 half
                const unsigned* p = (s % 3 == 1 ? pb : pa) + (ky * RS16 + j0) / 2;
                u32x4 a0 = u32x4{p[0], p[1], p[2], p[3]};
                u32x4 a1 = u32x4{p[RS16], p[RS16 + 1], p[RS16 + 2], p[RS16 + 3]};  // second conv row: two input rows down
                if (s == KH / 16 - 1) {  // k >= 168 (h = 1 of the last step): zero weights, and the row read does not exist
                    a0 = hh ? u32x4{0u, 0u, 0u, 0u} : a0;
                    a1 = hh ? u32x4{0u, 0u, 0u, 0u} : a1;
                }
                const u32x4 b0 = wv[2 * s * COUT], b1 = wv[2 * s * COUT + 32];
                if constexpr (MMA == 2) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<b8v*>(&a0), *reinterpret_cast<const b8v*>(&b0), acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<b8v*>(&a0), *reinterpret_cast<const b8v*>(&b1), acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<b8v*>(&a1), *reinterpret_cast<const b8v*>(&b0), acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<b8v*>(&a1), *reinterpret_cast<const b8v*>(&b1), acc[1][1], 0, 0, 0);
                } else {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<h8v*>(&a0), *reinterpret_cast<const h8v*>(&b0), acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<h8v*>(&a0), *reinterpret_cast<const h8v*>(&b1), acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<h8v*>(&a1), *reinterpret_cast<const h8v*>(&b0), acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<h8v*>(&a1), *reinterpret_cast<const h8v*>(&b1), acc[1][1], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int s = 0; s < KROWS / 2; ++s) {
            const int k0 = 2 * s, ky = k0 / 21, j0 = k0 - 21 * ky;
            const bool wrap = j0 == 20 && s != KROWS / 2 - 1;  // the last step's h = 1 is the zero row: any finite float will do
            const int off = ky * RS + j0;
            const float a0 = wrap ? aw_ptr[off] : a_ptr[off];
            const float a1 = wrap ? aw_ptr[off + 2 * RS] : a_ptr[off + 2 * RS];
            const float b0 = w_ptr[k0 * COUT], b1 = w_ptr[k0 * COUT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        }
        SSTAMP(0)
        lds_barrier();  // every wave is done with the input rows: the region becomes the V tile
        SSTAMP(1)

        // ---- bias + ReLU, vertical maximum in registers (C/D layout: channel = lane & 31, pixel = (e&3) + 8 (e>>2) + 4 h) ----
        const bool row0 = 2 * py < d.ho, row1 = 2 * py + 1 < d.ho;
        // both conv rows on the map, all 32 columns of the wave on the map, no pre-pool output wanted (wave-uniform, the common
        // case): 5 vector instructions per value instead of ~10 -- this phase competes for VALU issue with the MFMA stream of the
        // CU's other workgroup and ran 2.2 x slower beside it than alone (timing build)
        if (row0 && row1 && ncols - 32 * wave >= 32 && yconv == nullptr) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = j == 0 ? bv0 : bv1;
                float* vdst = ring + (32 * wave + 4 * hh) * COUT + j * 32 + (lane & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float r0 = fmaxf(acc[0][j][e] + bv, 0.0f), r1 = fmaxf(acc[1][j][e] + bv, 0.0f);
                    vdst[((e & 3) + 8 * (e >> 2)) * COUT] = fmaxf(carry[j][e], fmaxf(r0, r1));
                    carry[j][e] = r1;
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = j == 0 ? bv0 : bv1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int lcol = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const bool colok = lcol < ncols;
                float r0 = acc[0][j][e] + bv, r1 = acc[1][j][e] + bv;
                r0 = (row0 && colok && r0 > 0.0f) ? r0 : 0.0f;
                r1 = (row1 && colok && r1 > 0.0f) ? r1 : 0.0f;
                const float v = fmaxf(carry[j][e], fmaxf(r0, r1));
                carry[j][e] = r1;
                ring[lcol * COUT + j * 32 + (lane & 31)] = v;
                if (yconv != nullptr && py >= q0 && colok) {  // the pre-pool activation too (UNet's first skip connection)
                    float* o = yconv + (((long)img * d.ho + 2 * py) * d.wo + c_start + lcol) * COUT + j * 32 + (lane & 31);
                    if (row0) o[0] = r0;
                    if (row1) o[(long)d.wo * COUT] = r1;
                }
            }
        }
        }
        lds_barrier();
        SSTAMP(2)

        // ---- horizontal maximum of three columns, one pooled row (<= 64 x 64 floats) written as float4 ----
        if (py >= q0) {
            const int c4 = tid & 15;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int px = p0 + pass * 16 + (tid >> 4);
                if (px < p1) {
                    const int lcm = 2 * px - c_start;
                    const float4* vrow = reinterpret_cast<const float4*>(ring) + c4;
                    float4 m = vrow[lcm * (COUT / 4)];
                    const float4 r = vrow[(lcm + 1) * (COUT / 4)];  // lcm + 1 <= 127: zeros where the column does not exist
                    m.x = fmaxf(m.x, r.x), m.y = fmaxf(m.y, r.y), m.z = fmaxf(m.z, r.z), m.w = fmaxf(m.w, r.w);
                    if (lcm > 0) {
                        const float4 l = vrow[(lcm - 1) * (COUT / 4)];
                        m.x = fmaxf(m.x, l.x), m.y = fmaxf(m.y, l.y), m.z = fmaxf(m.z, l.z), m.w = fmaxf(m.w, l.w);
                    }
                    const long o = (((long)img * d.hp + py) * d.wp + px) * COUT + 4 * c4;
                    if (d.out_dtype == TIA_DT_F32) {
                        *reinterpret_cast<float4*>(static_cast<float*>(yout) + o) = m;
                    } else {  // one rounding to half (round to nearest even) for the fp16 / bf16 trunk
                        unsigned short hv[4];
                        const float mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (d.out_dtype == TIA_DT_BF16) {
                                unsigned u = __float_as_uint(mv[k]);
                                u += 0x7fffu + ((u >> 16) & 1u);  // values are finite and >= 0 here (after ReLU)
                                hv[k] = (unsigned short)(u >> 16);
                            } else {
                                const _Float16 hh = (_Float16)mv[k];
                                __builtin_memcpy(&hv[k], &hh, 2);
                            }
                        }
                        *reinterpret_cast<uint2*>(static_cast<unsigned short*>(yout) + o) =
                            make_uint2((unsigned)hv[0] | ((unsigned)hv[1] << 16), (unsigned)hv[2] | ((unsigned)hv[3] << 16));
                    }
                }
            }
        }
        lds_barrier();
        SSTAMP(3)
        if (py + 1 < q1) write_ring(py + 1);
        lds_barrier();
        SSTAMP(4)
    }
#if TIA_STEM_TIMING
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 700))
        printf("stem wg %d: iters %d  loads+mfma %lld  barrier %lld  vtile %lld  hmax+store %lld  write_ring %lld  loop %lld  | shader clock %.0f MHz\n",
               (int)blockIdx.x, q1 - it0, tm_[0] / (q1 - it0), tm_[1] / (q1 - it0), tm_[2] / (q1 - it0), tm_[3] / (q1 - it0),
               tm_[4] / (q1 - it0), tm_[5] / (q1 - it0), 100.0 * (double)(clock64() - t0c_) / (double)(wall_clock64() - t0w_));
#endif
}

}  // namespace

namespace {
// OIHW [64][3][7][7] -> [ky][kx][c][cout] = [147][64], followed by one zero row
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KROWS * COUT) return;
    const int o = i % COUT, k = i / COUT;
    float v = 0.0f;
    if (k < 147) {
        const int ky = k / 21, r = k - 21 * ky, kx = r / 3, c = r - 3 * kx;
        v = w[((o * 3 + c) * 7 + ky) * 7 + kx];
    }
    out[i] = v;
}
// OIHW [64][3][7][7] -> [k / 8][cout][8] halves with k = 24 ky + 3 kx + c (kx * 3 + c >= 21 and k >= 168: zeros)
template <int MMA>
__global__ __launch_bounds__(256) void stem_pack_h_kernel(const float* __restrict__ w, unsigned short* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KH * COUT) return;
    const int e = i & 7, o = (i >> 3) % COUT, chunk = i / (8 * COUT);
    const int k = chunk * 8 + e, ky = k / 24, j = k - 24 * ky;
    float v = 0.0f;
    if (ky < 7 && j < 21) {
        const int kx = j / 3, c = j - 3 * kx;
        v = w[((o * 3 + c) * 7 + ky) * 7 + kx];
    }
    out[i] = to_half_bits<MMA>(v);
}
}  // namespace

extern "C" int tia_stem_pack_weights_f32(const float* d_w_oihw, float* d_packed, void* stream) {
    if (!d_w_oihw || !d_packed) return TIA_EINVAL;
    hipLaunchKernelGGL(stem_pack_kernel, dim3((KROWS * COUT + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_w_oihw, d_packed);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}


extern "C" int tia_stem_pack_weights_h(const float* d_w_oihw, int32_t dtype, void* d_packed, void* stream) {
    if (!d_w_oihw || !d_packed || (dtype != TIA_DT_F16 && dtype != TIA_DT_BF16)) return TIA_EINVAL;
    const dim3 grid((KH * COUT + 255) / 256);
    if (dtype == TIA_DT_BF16)
        hipLaunchKernelGGL(stem_pack_h_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, d_w_oihw, (unsigned short*)d_packed);
    else
        hipLaunchKernelGGL(stem_pack_h_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d_w_oihw, (unsigned short*)d_packed);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

// mma: 0 = float32 matrix cores (weights [148][64] float32), TIA_DT_F16 / TIA_DT_BF16 = half matrix cores (weights packed by
// tia_stem_pack_weights_h; the output type is then that half type)
static int stem_impl(const void* d_x, int32_t x_is_u8, const void* d_w_packed, const float* d_bias, void* d_y, int32_t y_dtype,
                     float* d_conv_out, int64_t n, int64_t h, int64_t w, int32_t mma, void* stream) {
    if (!d_x || !d_w_packed || !d_bias || !d_y || n <= 0 || h <= 0 || w <= 0) return TIA_EINVAL;
    if (y_dtype != TIA_DT_F32 && y_dtype != TIA_DT_F16 && y_dtype != TIA_DT_BF16) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_w_packed) | reinterpret_cast<uintptr_t>(d_y)) & 15) return TIA_EINVAL;
    if (!x_is_u8 && (reinterpret_cast<uintptr_t>(d_x) & 3)) return TIA_EINVAL;
    const long ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;  // (h + 6 - 7) / 2 + 1
    const long hp = (ho - 1) / 2 + 1, wp = (wo - 1) / 2 + 1;  // (ho + 2 - 3) / 2 + 1
    const long esz = x_is_u8 ? 1 : 4;
    const long image_bytes = h * w * 3 * esz;
    if (image_bytes > 0x7fffffffL) return TIA_ESIZE;
    long group = 0x7fffffffL / image_bytes;  // 32-bit byte offsets inside a launch
    if (group > 0x7fffffffL / (h * w * 3)) group = 0x7fffffffL / (h * w * 3);
    group = tia::even_group(n, group);
    const long strips = wp <= 64 ? 1 : 1 + (wp - 64 + 62) / 63;
    using Kernel = void (*)(const void*, const void*, const float*, void*, float*, StemDims);
    const Kernel kernels[3][2] = {{stem7x7_pool_kernel<false, 0>, stem7x7_pool_kernel<true, 0>},
                                  {stem7x7_pool_kernel<false, 1>, stem7x7_pool_kernel<true, 1>},
                                  {stem7x7_pool_kernel<false, 2>, stem7x7_pool_kernel<true, 2>}};
    static tia::DeviceOnce attr_once;  // the dynamic-LDS attribute is per device
    if (!attr_once.ensure([&] {
            for (int m = 0; m < 3; ++m)
                for (int u = 0; u < 2; ++u)
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[m][u]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            m == 0 ? (int)(LDS_FLOATS * sizeof(float)) : LDS_BYTES_H) != hipSuccess)
                        return false;
            return true;
        }))
        return TIA_ELAUNCH;
    const int mi = mma == TIA_DT_F16 ? 1 : (mma == TIA_DT_BF16 ? 2 : 0);
    const size_t lds = mi == 0 ? (size_t)LDS_FLOATS * sizeof(float) : (size_t)LDS_BYTES_H;
    hipStream_t st = (hipStream_t)stream;
    for (long first = 0; first < n; first += group) {
        const long nb = n - first < group ? n - first : group;
        // enough workgroups for two per CU and a second round: cut images into row chunks when the batch is small
        long chunks = (1024 + nb * strips - 1) / (nb * strips);
        const long max_chunks = hp >= 8 ? hp / 8 : 1;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
        const long rows = (hp + chunks - 1) / chunks;
        chunks = (hp + rows - 1) / rows;
        const char* xg = static_cast<const char*>(d_x) + first * image_bytes;
        const int shift = x_is_u8 ? (int)(reinterpret_cast<uintptr_t>(xg) & 3) : 0;
        xg -= shift;
        StemDims d{(int)nb, (int)h, (int)w, (int)ho, (int)wo, (int)hp, (int)wp, (int)chunks, (int)rows,
                   (unsigned)((nb * image_bytes + shift + 3) & ~3L), shift, (int)y_dtype};
        char* yg = static_cast<char*>(d_y) + first * hp * wp * COUT * (y_dtype == TIA_DT_F32 ? 4 : 2);
        float* cg = d_conv_out ? d_conv_out + first * ho * wo * COUT : nullptr;
        const dim3 grid((unsigned)(nb * chunks), (unsigned)strips);
        hipLaunchKernelGGL(kernels[mi][x_is_u8 ? 1 : 0], grid, dim3(NTH), lds, st, static_cast<const void*>(xg), d_w_packed, d_bias,
                           static_cast<void*>(yg), cg, d);
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_stem_conv7x7_pool_nhwc(const void* d_x, int32_t x_is_u8, const float* d_w_packed, const float* d_bias, void* d_y,
                                          int32_t y_dtype, float* d_conv_out, int64_t n, int64_t h, int64_t w, void* stream) {
    return stem_impl(d_x, x_is_u8, d_w_packed, d_bias, d_y, y_dtype, d_conv_out, n, h, w, 0, stream);
}

extern "C" int tia_stem_conv7x7_pool_nhwc_h(const void* d_x, int32_t x_is_u8, const void* d_w_packed_h, const float* d_bias, void* d_y,
                                            int32_t dtype, int64_t n, int64_t h, int64_t w, void* stream) {
    if (dtype != TIA_DT_F16 && dtype != TIA_DT_BF16) return TIA_EINVAL;
    return stem_impl(d_x, x_is_u8, d_w_packed_h, d_bias, d_y, dtype, nullptr, n, h, w, dtype, stream);
}
