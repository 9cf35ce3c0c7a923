// NHWC fp16 / bf16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16 / _bf16: half inputs, float32
// accumulate), with the ResNet block epilogue -- bias + residual + ReLU, all in float32 before the one rounding to half -- fused in.
// The half-precision sibling of conv_mfma.hip (reference call site: CNNModel.forward -> torchvision BasicBlock / Bottleneck,
// models/architecture/vanilla.py:300-316; the reference itself runs float32 -- this is the `compute_dtype="float16"|"bfloat16"`
// extension of the engines, checked against the float32 probabilities to the reference's own 1e-3 tolerance,
// tests/engines/test_patch_predictor.py:712-722).
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, reduced tap by tap in slices of 32 input channels (64 bytes per
// pixel: NHWC keeps a tap's channels contiguous).
//   * workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); tile 128 pixels x BN channels (BN = 128 | 64); a wave owns 64 x BN/2
//     outputs = 2 x (BN/64) MFMA tiles of 32x32, accumulators in registers; a slice is two k-steps of 16
//   * operands of an MFMA lane are 8 consecutive halves = ONE 16-byte LDS read each:
//       A slice [128 pixels][4 chunks of 8 channels] with the chunk index XOR-ed by (pixel >> 2) & 3: the 16 lanes a
//         ds_read_b128 services together (rows distinct mod 16) then cover all 16 quads of the 256-byte bank row;
//       B slice [4 k-chunks][BN columns][8 halves]: the weights are pre-packed [tap][cin/8][cout][8], so a lane's 8 k-values
//         of its column are contiguous in memory AND in LDS, and lanes walk the columns (linear, conflict-free)
//   * a ring of three LDS stages filled by LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS without passing through registers or the
//     ds_write path, which at ~80 B/clk/CU was the bottleneck of the register-staged first version: 508 -> see profiles/r03*):
//     the DMA of slice s+2 is issued before the MFMAs of slice s, one raw barrier and one counted vmcnt per slice.  The
//     LDS image of a DMA is lane-linear, so the XOR swizzle of the A slice sits on the SOURCE side: the thread that fills
//     position P fetches chunk (P & 3) ^ ((P >> 4) & 3) of pixel P >> 2.  Loads go through buffer descriptors (per slot: byte
//     offset of tap (0,0) + one bit per kernel row / column, as in conv_mfma.hip); a padding tap gets an out-of-range offset
//     and the DMA writes zeros
//   * epilogue through LDS in two column halves: accumulators -> float32 tile [128][BN/2] -> rows re-read 8 columns per lane,
//     + bias + residual (16-byte loads), ReLU, ONE rounding to half, 16-byte stores (whole 128- or 256-byte output rows)
//   * blockIdx remapped so that each XCD walks a contiguous range of pixel tiles
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tiatoolbox_amd.h"
#include "common.hpp"
#include "conv3x3_spatial.hpp"

namespace {

constexpr int BM = 128;
constexpr int NTH = 256;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using b8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct ConvDimsH {
    int n, h, w, cin, cout, ho, wo, kh, kw, stride, pad_y, pad_x;
    unsigned x_bytes, w_bytes;
};
constexpr int OOB = (int)0x80000000;

template <bool BF>
__device__ __forceinline__ f32x16 mma(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const b8*>(&a), *reinterpret_cast<const b8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&a), *reinterpret_cast<const h8*>(&b), c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ float half_to_f32(unsigned short v) {
    if constexpr (BF) return __uint_as_float((unsigned)v << 16);
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
template <bool BF>
__device__ __forceinline__ unsigned short f32_to_half(float x) {  // round to nearest even
    if constexpr (BF) {
        unsigned u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    } else {
        const _Float16 h = (_Float16)x;
        unsigned short v;
        __builtin_memcpy(&v, &h, 2);
        return v;
    }
}

// 16 bytes per lane from a buffer straight into LDS (buffer_load_dwordx4 ... lds): the wave's 64 lanes fill the 1 KB at
// `lds_wave_base` in lane order; an out-of-range `voffset` writes zeros.  (A __device__ function: the builtin must not be seen by
// the host pass, which otherwise drops the kernel's launch stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

template <int BN, bool BF, int BK>
__global__ __launch_bounds__(NTH, 2) void conv_mfma_h_kernel(const void* __restrict__ x, const void* __restrict__ wk,
                                                          const float* __restrict__ bias, const void* __restrict__ res,
                                                          void* __restrict__ y, ConvDimsH d, int relu, int m_tiles) {
    static_assert(BK == 32 || BK == 64, "slices of 32 or 64 input channels");
    constexpr int NTILE = BN / 64;
    constexpr int CH = BK / 8;                       // 16-byte chunks (8 halves) per pixel and slice
    constexpr int KS = BK / 16;                      // MFMA k-steps per slice
    constexpr int SH = CH == 4 ? 2 : 1;              // swizzle: chunk ^ ((pixel >> SH) & (CH - 1)) spreads 16 rows over 16 quads
    constexpr int A_BYTES = BM * BK * 2;             // 8 | 16 KB per stage
    constexpr int B_BYTES = BK * BN * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_SLOTS = BM * CH / NTH;           // 16-byte pixel chunks per thread and slice: 2 | 4
    constexpr int B_SLOTS = CH * BN / NTH;           // 16-byte weight chunks per thread and slice
    constexpr int NSTAGE = BK == 32 ? 3 : 2;         // LDS ring (BK 32: slice s is consumed while s+1 and s+2 are in flight)
    constexpr int DMA_PER_SLICE = A_SLOTS + B_SLOTS; // LDS-DMA instructions a thread issues per slice (vmcnt bookkeeping)
    constexpr int LDS_BYTES = NSTAGE * STAGE > BM * (BN / 2) * 4 ? NSTAGE * STAGE : BM * (BN / 2) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int mt_id = (bid % 8) * per_xcd + bid / 8;
    if (mt_id >= m_tiles) return;
    const long m0 = (long)mt_id * BM;
    const int n0 = blockIdx.y * BN;
    const long m_total = (long)d.n * d.ho * d.wo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wk), 0, (int)d.w_bytes, 0x00020000);

    // ---- A staging: thread -> LDS positions P = tid + 256 r (16-byte units, lane-linear as the DMA writes them):
    //      pixel = P / CH = tid / CH + (256 / CH) r, source chunk = (P % CH) ^ ((pixel >> SH) & (CH - 1)), the same for every r ----
    const int chunk = (tid & (CH - 1)) ^ (((tid / CH) >> SH) & (CH - 1));
    int cen[A_SLOTS];
    unsigned msk[A_SLOTS];
#pragma unroll
    for (int r = 0; r < A_SLOTS; ++r) {
        const long m = m0 + tid / CH + (NTH / CH) * r;
        const bool pvalid = m < m_total;
        const int mm = pvalid ? (int)m : 0;
        const int b = mm / (d.ho * d.wo);
        const int rem = mm - b * d.ho * d.wo;
        const int oy = rem / d.wo, ox = rem - oy * d.wo;
        const int iy0 = oy * d.stride - d.pad_y, ix0 = ox * d.stride - d.pad_x;
        cen[r] = (((b * d.h + iy0) * d.w + ix0) * d.cin + 8 * chunk) * 2;
        unsigned rows = 0, cols = 0;
        for (int t = 0; t < d.kh; ++t) rows |= (unsigned)((unsigned)(iy0 + t) < (unsigned)d.h) << t;
        for (int t = 0; t < d.kw; ++t) cols |= (unsigned)((unsigned)(ix0 + t) < (unsigned)d.w) << (16 + t);
        msk[r] = pvalid ? (rows | cols) : 0u;
    }
    // B slots: linear index idx = tid + 256 r over [CH k-chunks][BN columns]; global: ((tap * cin/8 + c0/8 + kc) * cout + n0 + col) * 16
    int b_off[B_SLOTS];
#pragma unroll
    for (int r = 0; r < B_SLOTS; ++r) {
        const int idx = tid + NTH * r;
        const int kc = idx / BN, col = idx - kc * BN;
        b_off[r] = (kc * d.cout + n0 + col) * 16;
    }

    int s_kh = 0, s_kw = 0, s_c0 = 0;
    // DMA of the slice under the cursor into `stage`: wave w, instruction r writes the 1 KB at position (w * 64 + 256 r) * 16
    auto dma_slice = [&](int stage) {
        const int sdelta = ((s_kh * d.w + s_kw) * d.cin + s_c0) * 2;
        const unsigned sel = (1u << s_kh) | (1u << (16 + s_kw));
        const int swrow = ((s_kh * d.kw + s_kw) * (d.cin >> 3) + (s_c0 >> 3)) * d.cout * 16;
        unsigned char* sa = smem + stage * STAGE + wave * 1024;
        unsigned char* sb = smem + stage * STAGE + A_BYTES + wave * 1024;
#pragma unroll
        for (int r = 0; r < A_SLOTS; ++r) {
            const bool ok = (msk[r] & sel) == sel;
            dma16(rx, sa + r * 4096, ok ? cen[r] + sdelta : OOB, 0);
        }
#pragma unroll
        for (int r = 0; r < B_SLOTS; ++r) dma16(rw, sb + r * 4096, b_off[r], swrow);
    };
    auto next_slice = [&]() {
        int c0 = s_c0 + BK, kw = s_kw, kh = s_kh;
        if (c0 == d.cin) { c0 = 0; ++kw; }
        if (kw == d.kw) { kw = 0; ++kh; }
        if (kh < d.kh) { s_c0 = c0; s_kw = kw; s_kh = kh; }
    };

    f32x16 acc[2][NTILE];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment positions (16-byte units): A tile i, k-step q: pixel p = wm*64 + i*32 + (lane & 31), chunk c = 2 q + (lane >> 5)
    int fa[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int p = wm * 64 + i * 32 + (lane & 31), c = 2 * q + (lane >> 5);
            fa[i][q] = p * CH + (c ^ ((p >> SH) & (CH - 1)));
        }
    // B tile j, k-step q: k-chunk 2 q + (lane >> 5), column wn * (BN/2) + j*32 + (lane & 31)
    const int fb0 = (lane >> 5) * BN + wn * (BN / 2) + (lane & 31);

    auto compute = [&](int stage) {
        const u32x4* sa = reinterpret_cast<const u32x4*>(smem + stage * STAGE);
        const u32x4* sb = reinterpret_cast<const u32x4*>(smem + stage * STAGE + A_BYTES);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            u32x4 a[2], b[NTILE];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sa[fa[i][q]];
#pragma unroll
            for (int j = 0; j < NTILE; ++j) b[j] = sb[fb0 + 2 * q * BN + j * 32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTILE; ++j) acc[i][j] = mma<BF>(a[i], b[j], acc[i][j]);
        }
    };

    const int n_slices = d.kh * d.kw * (d.cin / BK);
    if constexpr (NSTAGE == 3) {
        // Three-stage ring with COUNTED waits: the DMA of slice s+2 is issued at the top of iteration s, and the iteration ends
        // by waiting only for slice s+1 (vmcnt(DMA_PER_SLICE): the instructions just issued for s+2 stay in flight across the
        // barrier).  Raw s_barrier: __syncthreads() would drain the DMA queue (an LDS-DMA is a pending LDS write).  A stage is
        // read in the iteration AFTER the wait + barrier that retired it; every LDS read of a stage has returned (lgkmcnt(0))
        // before the wave arrives at the barrier behind which the stage is refilled.
        dma_slice(0);
        next_slice();
        dma_slice(1);
        if constexpr (DMA_PER_SLICE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int sidx = 0; sidx < n_slices; ++sidx) {
            next_slice();
            int nxt2 = cur + 2;
            nxt2 = nxt2 >= NSTAGE ? nxt2 - NSTAGE : nxt2;
            dma_slice(nxt2);  // past the last slice the cursor stays put: the tail re-fetches the last slice (no control flow)
            compute(cur);
            if constexpr (DMA_PER_SLICE == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        }
    } else {
        // Two stages of 64-channel slices (whole 128-byte lines per pixel, 4 k-steps = 16 MFMAs per wave between barriers): the
        // DMA of slice s+1 is issued before the MFMAs of slice s and must have landed at the barrier that ends the iteration.
        dma_slice(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int sidx = 0; sidx < n_slices; ++sidx) {
            const int cur = sidx & 1;
            next_slice();
            dma_slice(cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's surplus DMAs must land before the epilogue reuses the LDS
    __syncthreads();

    // ---- epilogue: per column half h (= the waves with wn == h): accumulators -> float32 LDS tile [128][BN/2], then every thread
    //      takes rows x 8-column chunks: + bias + residual, ReLU, round once, 16-byte stores ----
    constexpr int HB = BN / 2;                 // columns per half
    constexpr int CHUNKS = BM * HB / 8;        // 8-column chunks of the half tile
    float* tile = reinterpret_cast<float*>(smem);
    const unsigned short* resh = reinterpret_cast<const unsigned short*>(res);
    unsigned short* yh = reinterpret_cast<unsigned short*>(y);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (wn == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTILE; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        tile[row * HB + j * 32 + (lane & 31)] = acc[i][j][e];
                    }
        }
        __syncthreads();
        for (int idx = tid; idx < CHUNKS; idx += NTH) {
            const int row = idx / (HB / 8), cc = idx - row * (HB / 8);
            const long m = m0 + row;
            if (m < m_total) {
                const int col0 = n0 + h * HB + cc * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(tile + row * HB + cc * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(tile + row * HB + cc * 8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias + col0), b1 = *reinterpret_cast<const float4*>(bias + col0 + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (res) {
                    const u32x4 rv = *reinterpret_cast<const u32x4*>(resh + m * d.cout + col0);
                    const unsigned rw4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] += half_to_f32<BF>((unsigned short)(rw4[k] & 0xffffu));
                        v[2 * k + 1] += half_to_f32<BF>((unsigned short)(rw4[k] >> 16));
                    }
                }
                unsigned o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a0 = v[2 * k], a1 = v[2 * k + 1];
                    if (relu) {
                        a0 = a0 > 0.0f ? a0 : 0.0f;
                        a1 = a1 > 0.0f ? a1 : 0.0f;
                    }
                    o[k] = (unsigned)f32_to_half<BF>(a0) | ((unsigned)f32_to_half<BF>(a1) << 16);
                }
                *reinterpret_cast<u32x4*>(yh + m * d.cout + col0) = u32x4{o[0], o[1], o[2], o[3]};
            }
        }
        __syncthreads();
    }
}


// OIHW float32 -> [kh][kw][cin/8][cout][8] halves (the GEMM's B matrix with a lane's 8 k-values contiguous)
template <bool BF>
__global__ __launch_bounds__(256) void pack_weights_h_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw,
                                                             unsigned short* __restrict__ out) {
    const long total = (long)cout * cin * kh * kw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k8 = (int)(i & 7);
        long t = i >> 3;
        const int o = (int)(t % cout);
        t /= cout;
        const int c8 = (int)(t % (cin >> 3));
        t /= (cin >> 3);
        const int xk = (int)(t % kw), yk = (int)(t / kw);
        const int c = c8 * 8 + k8;
        out[i] = f32_to_half<BF>(w[(((long)o * cin + c) * kh + yk) * kw + xk]);
    }
}

}  // namespace

extern "C" int tia_conv_pack_weights_h(const float* d_w_oihw, int64_t cout, int64_t cin, int64_t kh, int64_t kw, int32_t dtype,
                                       void* d_packed, void* stream) {
    if (!d_w_oihw || !d_packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return TIA_EINVAL;
    if (dtype != TIA_DT_F16 && dtype != TIA_DT_BF16) return TIA_EINVAL;
    if (cin % 8 != 0) return TIA_ESIZE;
    const long total = (long)cout * cin * kh * kw;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == TIA_DT_BF16)
        hipLaunchKernelGGL(pack_weights_h_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_w_oihw, (int)cout,
                           (int)cin, (int)kh, (int)kw, (unsigned short*)d_packed);
    else
        hipLaunchKernelGGL(pack_weights_h_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_w_oihw, (int)cout,
                           (int)cin, (int)kh, (int)kw, (unsigned short*)d_packed);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_conv2d_nhwc_h(const void* d_x, const void* d_w_packed, const float* d_bias, const void* d_residual, void* d_y,
                                 int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                                 int64_t pad, int32_t dtype, int32_t relu, void* stream) {
    if (!d_x || !d_w_packed || !d_y || n <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return TIA_EINVAL;
    if (dtype != TIA_DT_F16 && dtype != TIA_DT_BF16) return TIA_EINVAL;
    if (cin % 32 != 0 || cout % 64 != 0) return TIA_ESIZE;
    if (((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_w_packed) | reinterpret_cast<uintptr_t>(d_y) |
          reinterpret_cast<uintptr_t>(d_residual) | reinterpret_cast<uintptr_t>(d_bias)) & 15) != 0)
        return TIA_EINVAL;
    if (kh > 16 || kw > 16 || pad >= kh || pad >= kw) return TIA_EINVAL;
    const long ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return TIA_EINVAL;
    const long image_bytes = h * w * cin * 2, w_bytes = kh * kw * cin * cout * 2;
    if (image_bytes > 0x7fffffffL || w_bytes > 0x7fffffffL || ho * wo > 0x7fffffffL / 4) return TIA_ESIZE;
    long group = 0x7fffffffL / image_bytes;
    if (group * ho * wo > 0x7fffffffL / 2) group = 0x7fffffffL / 2 / (ho * wo);
    if (group < 1) return TIA_ESIZE;
    group = tia::even_group(n, group);
    hipStream_t st = (hipStream_t)stream;
    const bool bf = dtype == TIA_DT_BF16;
    for (long first = 0; first < n; first += group) {
        const long nb = n - first < group ? n - first : group;
        const long m_total = nb * ho * wo;
        const long m_tiles = (m_total + BM - 1) / BM;
        ConvDimsH d{(int)nb, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)kh, (int)kw, (int)stride, (int)pad, (int)pad,
                    (unsigned)(nb * image_bytes), (unsigned)w_bytes};
        const char* xg = static_cast<const char*>(d_x) + first * image_bytes;
        const char* rg = d_residual ? static_cast<const char*>(d_residual) + first * ho * wo * cout * 2 : nullptr;
        char* yg = static_cast<char*>(d_y) + first * ho * wo * cout * 2;
        // 3x3 / stride 1 on maps that 16 x 16 pixel blocks cover with little waste: the tap-reuse form (conv3x3_spatial.hip)
        if (tia::conv3x3_spatial_ok(kh, kw, stride, h, w, ho, wo, pad, pad, false) &&
            tia::conv3x3_spatial_launch(xg, d_w_packed, d_bias, rg, yg, nb, h, w, cin, cout, pad, pad, ho, wo, dtype, relu, st))
            continue;
        const long grid_x = ((m_tiles + 7) / 8) * 8;
        // 64-channel slices (two stages, 16 MFMAs per barrier, whole cache lines per pixel) measured no faster than 32-channel
        // slices in a three-stage ring (profiles/r03e_perf_conv_h*.txt: 527 vs 539 TF/s over the resnet18 trunk; slower on the
        // 1x1 convolutions): both sit on the global -> LDS byte rate, not on latency or barriers.  Kept as a developer switch.
        static const bool want64 = tia::dev_env("TIA_CONVH_BK64") != nullptr;
        const bool bk64 = cin % 64 == 0 && want64;
        const bool wide = cout % 128 == 0;
        const dim3 grid((unsigned)grid_x, (unsigned)(cout / (wide ? 128 : 64)));
#define TIA_LAUNCH_H(BN_, BF_, BK_) \
    hipLaunchKernelGGL((conv_mfma_h_kernel<BN_, BF_, BK_>), grid, dim3(NTH), 0, st, xg, d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles)
        if (wide) {
            if (bf) { if (bk64) TIA_LAUNCH_H(128, true, 64); else TIA_LAUNCH_H(128, true, 32); }
            else { if (bk64) TIA_LAUNCH_H(128, false, 64); else TIA_LAUNCH_H(128, false, 32); }
        } else {
            if (bf) { if (bk64) TIA_LAUNCH_H(64, true, 64); else TIA_LAUNCH_H(64, true, 32); }
            else { if (bk64) TIA_LAUNCH_H(64, false, 64); else TIA_LAUNCH_H(64, false, 32); }
        }
#undef TIA_LAUNCH_H
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
