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


// ---- 3x3 / stride 1 / pad 1 with tap reuse: a workgroup owns a 16 x 16 block of output pixels of ONE image ------------------
// The byte rate of the global -> LDS path bounds the kernel above (profiles/r03f_*: matrix pipe 35 % busy, LDS 22 %, waves parked
// on the DMA counters 58 %), so this form moves fewer bytes per flop: the 18 x 18 input patch of a 32-channel slice is brought
// in ONCE and read by all nine taps (20.7 KB instead of 9 x 16 KB), and 256 pixels share every weight slice (8 KB per tap and
// channel slice) -- 0.36 KB per output pixel, tap group and channel slice instead of 1.1 KB.
//   * 512 threads = 8 waves as 4 (M: four pixel rows each) x 2 (N); MFMA tile i of a wave = two 16-pixel rows
//   * patch in LDS, in 16-byte units: pixel (py, px) at py * 96 + px * 5 (+ chunk 0..3; the fifth unit is padding): the 16 lanes a
//     ds_read_b128 services together hold pixels {0-3, 12-15} of one row and {4-11} of the next; with a pixel pitch of 5 units
//     and a row pitch of 0 mod 16 they fall into 16 different 16-byte bank groups for every tap shift, and a tap, a k-step or the
//     second MFMA tile is an IMMEDIATE offset on one base register per lane.  Double-buffered; the next channel slice's patch
//     arrives in four LDS-DMA pieces behind taps 0-3 (the DMA image is lane-linear: padding units fetch out of range = zeros)
//   * weights: ring of three 8 KB stages, one LDS-DMA instruction per thread and tap; counted vmcnt per tap position
template <int BN, bool BF>
__global__ __launch_bounds__(512, 4) void conv3x3_h_kernel(const void* __restrict__ x, const void* __restrict__ wk,
                                                          const float* __restrict__ bias, const void* __restrict__ res,
                                                          void* __restrict__ y, ConvDimsH d, int relu, int m_tiles, int tiles_x,
                                                          int tiles_per_image) {
    constexpr int NT = 512, NTILE = BN / 64, PW = 18, PIX = 5, ROW = 96;
    constexpr int A_UNITS = PW * ROW;     // 1728 units: 3 whole DMA rounds of 512 + 192
    constexpr int A_BYTES = A_UNITS * 16;
    constexpr int B_BYTES = NT * 16;      // one unit per thread: [4 k-chunks][BN columns][8 halves] (BN = 64: the upper half idles)
    constexpr int DUMP = 2 * A_BYTES + 3 * B_BYTES;  // 1 KB that the idle waves of the fourth patch piece write their zeros to
    constexpr int LDS_BYTES = DUMP + 1024;
    static_assert(LDS_BYTES >= 256 * (BN / 2) * 4, "epilogue tile");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int mt_id = (bid % 8) * per_xcd + bid / 8;
    if (mt_id >= m_tiles) return;
    const int img = mt_id / tiles_per_image, trem = mt_id - img * tiles_per_image;
    const int ty0 = (trem / tiles_x) * 16, tx0 = (trem - (trem / tiles_x) * tiles_x) * 16;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wk), 0, (int)d.w_bytes, 0x00020000);

    // patch staging: unit U = 512 r + tid -> row U / 96, pixel (U % 96) / 5, chunk (U % 96) % 5 (4 = padding)
    int cen[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = NT * r + tid;
        const int py = u / ROW, rem = u - py * ROW;
        const int px = rem / PIX, chunk = rem - px * PIX;
        const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
        const bool inside = py < PW && px < PW && chunk < 4 && (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
        cen[r] = inside ? (((img * d.h + iy) * d.w + ix) * d.cin + 8 * chunk) * 2 : OOB;
    }
    const int b_kc = tid / BN, b_col = tid - b_kc * BN;
    const int b_off = b_kc < 4 ? (b_kc * d.cout + n0 + b_col) * 16 : OOB;
    const int n_cs = d.cin >> 5, last = 9 * n_cs - 1;

    unsigned char* const abuf0 = smem;
    unsigned char* const bring = smem + 2 * A_BYTES;
    auto dma_a = [&](int buf, int r, int cs) {
        // the fourth piece covers units 1536 .. 1727 (waves 0-2); the other waves' lanes are all out of range: zeros to the dump
        unsigned char* dst = (r == 3 && wave >= 3) ? smem + DUMP : abuf0 + buf * A_BYTES + r * (NT * 16) + wave * 1024;
        dma16(rx, dst, cen[r], cs * 64);
    };
    // weight slice of flattened step s = cs * 9 + tap (clamped: the tail re-fetches the last slice)
    auto dma_b = [&](int stage, int s) {
        s = s < last ? s : last;
        const int cs = s / 9, tap = s - cs * 9;
        dma16(rw, bring + stage * B_BYTES + wave * 1024, b_off, (tap * (d.cin >> 3) + cs * 4) * d.cout * 16);
    };

    f32x16 acc[2][NTILE];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // one base per lane (units): MFMA row = lane & 31 -> pixel row 4 wm + (row >> 4) (+ 2 i), column row & 15; k-chunk lane >> 5 (+ 2 q)
    const int fa0 = (4 * wm + ((lane & 31) >> 4)) * ROW + (lane & 15) * PIX + (lane >> 5);
    const int fb0 = (lane >> 5) * BN + wn * (BN / 2) + (lane & 31);

    auto compute = [&](int buf, int stage, int tap) {
        const u32x4* sa = reinterpret_cast<const u32x4*>(abuf0 + buf * A_BYTES) + fa0;
        const u32x4* sb = reinterpret_cast<const u32x4*>(bring + stage * B_BYTES) + fb0;
        const int shift = (tap / 3) * ROW + (tap % 3) * PIX;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 a[2], b[NTILE];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sa[shift + i * 2 * ROW + 2 * q];
#pragma unroll
            for (int j = 0; j < NTILE; ++j) b[j] = sb[2 * q * BN + j * 32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTILE; ++j) acc[i][j] = mma<BF>(a[i], b[j], acc[i][j]);
        }
    };

    // prologue: patch of slice 0, weight slices 0 and 1
    dma_a(0, 0, 0);
    dma_a(0, 1, 0);
    dma_a(0, 2, 0);
    dma_a(0, 3, 0);
    dma_b(0, 0);
    dma_b(1, 1);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int cs = 0; cs < n_cs; ++cs) {
        const int buf = cs & 1, s0 = cs * 9;
        const int cs_next = cs + 1 < n_cs ? cs + 1 : cs;  // past the end: the idle buffer is refilled with the same slice
        // Per tap: the weight slice two steps ahead goes out first, then (taps 0-3) one piece of the next patch; the wait at the end
        // lets exactly the instructions younger than weight slice s + 1 stay in flight (queue, oldest first, "|" = must have landed:
        // t=0: B(s+1) | B(s+2) A0;   t=1: B(s+2) | A0 B(s+3) A1;   t=2: A0 B(s+3) | A1 B(s+4) A2;   t=3: A1 B(s+4) | A2 B(s+5) A3;
        // t=4: A2 B(s+5) | A3 B(s+6);   t=5: A3 B(s+6) | B(s+7);   t>=6: B(s+1) | B(s+2)).
        // lgkmcnt(0): every LDS read of the stage refilled next has returned before the barrier.
#define TIA_TAP(T, VM)                                                              \
        dma_b((T + 2) % 3, s0 + T + 2);                                             \
        if (T < 4) dma_a(buf ^ 1, T, cs_next);                                      \
        compute(buf, T % 3, T);                                                     \
        asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_s_barrier();
        TIA_TAP(0, 2)
        TIA_TAP(1, 3)
        TIA_TAP(2, 3)
        TIA_TAP(3, 3)
        TIA_TAP(4, 2)
        TIA_TAP(5, 1)
        TIA_TAP(6, 1)
        TIA_TAP(7, 1)
        TIA_TAP(8, 1)
#undef TIA_TAP
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue (as above, 256 rows): row m of the block = pixel (ty0 + m / 16, tx0 + m % 16) ----
    constexpr int HB = BN / 2, CHUNKS = 256 * HB / 8;
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
        for (int idx = tid; idx < CHUNKS; idx += NT) {
            const int row = idx / (HB / 8), cc = idx - row * (HB / 8);
            const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
            if (oy < d.ho && ox < d.wo) {
                const long m = ((long)img * d.ho + oy) * d.wo + ox;
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
        // 3x3 / stride 1 / pad 1 on maps that 16 x 16 pixel blocks cover with little waste: the tap-reuse form
        static const bool no_spatial = getenv("TIA_CONVH_NO_SPATIAL") != nullptr;
        const long tiles_y = (ho + 15) / 16, tiles_x = (wo + 15) / 16;
        if (!no_spatial && kh == 3 && kw == 3 && stride == 1 && pad == 1 && 4 * ho * wo >= 3 * tiles_y * tiles_x * 256) {
            const long tiles = nb * tiles_y * tiles_x;
            const dim3 sgrid((unsigned)(((tiles + 7) / 8) * 8), (unsigned)(cout / (cout % 128 == 0 ? 128 : 64)));
#define TIA_LAUNCH_S(BN_, BF_) \
    hipLaunchKernelGGL((conv3x3_h_kernel<BN_, BF_>), sgrid, dim3(512), 0, st, xg, d_w_packed, d_bias, rg, yg, d, relu, (int)tiles, \
                       (int)tiles_x, (int)(tiles_y * tiles_x))
            if (cout % 128 == 0) { if (bf) TIA_LAUNCH_S(128, true); else TIA_LAUNCH_S(128, false); }
            else { if (bf) TIA_LAUNCH_S(64, true); else TIA_LAUNCH_S(64, false); }
#undef TIA_LAUNCH_S
            continue;
        }
        const long grid_x = ((m_tiles + 7) / 8) * 8;
        // 64-channel slices (two stages, 16 MFMAs per barrier, whole cache lines per pixel) measured no faster than 32-channel
        // slices in a three-stage ring (profiles/r03e_perf_conv_h*.txt: 527 vs 539 TF/s over the resnet18 trunk; slower on the
        // 1x1 convolutions): both sit on the global -> LDS byte rate, not on latency or barriers.  Kept as a developer switch.
        static const bool want64 = getenv("TIA_CONVH_BK64") != nullptr;
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
