// 3x3 / stride-1 NHWC convolution with TAP REUSE on the gfx950 matrix cores, float32 (v_mfma_f32_32x32x2_f32: the reference's
// arithmetic) and fp16 / bf16 (v_mfma_f32_32x32x16_*, float32 accumulate), bias + residual + ReLU fused.  Call sites in the
// reference: the 3x3 convolutions of torchvision's BasicBlock / Bottleneck behind CNNModel.forward (models/architecture/
// vanilla.py:300-316), UNetModel's encoder / decoder blocks (unet.py:356-417), HoVerNet's residual units (hovernet.py:405-454).
//
// The slice kernels (conv_mfma.hip, conv_mfma_h.hip) fetch a [128 pixels] x [32 channels] operand per tap, so a pixel's
// channels cross the global -> LDS path nine times, and they pay a barrier pair (f32) or a DMA-bound slice (half) per 64 / 8
// MFMAs.  Here a workgroup owns a 16 x 16 block of output pixels of ONE image:
//   * the 18 x 18 input patch of a 64-byte channel slice (16 float32 / 32 half channels) is brought in ONCE and read by all nine
//     taps; 256 pixels share every weight slice (8 KB per tap and channel slice): 0.36 KB moved per output pixel, tap group and
//     slice instead of 1.1 KB
//   * 512 threads = 8 waves as 4 (M: four pixel rows each) x 2 (N); MFMA tile i of a wave = two 16-pixel rows; BN = 128 | 64
//   * patch in LDS, in 16-byte units: pixel (py, px) at py * 96 + px * 5 (+ unit 0..3; the fifth unit is padding): the 16 lanes
//     a ds_read_b128 services together hold pixels {0-3, 12-15} of one row and {4-11} of the next; with a pixel pitch of 5 units
//     and a row pitch of 0 mod 16 they fall into 16 different 16-byte bank groups for every tap shift, and a tap, a k-step or
//     the second MFMA tile is an IMMEDIATE offset on one base register per lane.  Double-buffered; the next slice's patch arrives
//     by LDS-DMA in four pieces behind taps 0-3 (the DMA image is lane-linear: padding units fetch out of range = zeros)
//   * weights: ring of three 8 KB stages, one LDS-DMA instruction per thread and tap; one raw s_barrier and one COUNTED vmcnt per
//     tap (the younger DMAs stay in flight across the barrier)
//   * float32: a lane's k values of a slice are channels 8 hi .. 8 hi + 7 (hi = lane >> 5): two 16-byte reads feed eight
//     MFMAs per tile; the weight stage is the [16 channels][BN] block of the [tap][cin][cout] packing, read by dword
//     half: a lane's 8 halves of k-chunk 2 q + hi are one 16-byte read on both sides ([tap][cin/8][cout][8] packing)
//   * epilogue through LDS in two column halves: float32 tile [256][BN/2] -> + bias + residual, ReLU, (one rounding), 16-byte stores
//   * blockIdx remapped so that each XCD walks a contiguous range of pixel blocks
#include "conv3x3_spatial.hpp"
#include <atomic>

#include <stdlib.h>

#include "../../include/tiatoolbox_amd.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using b8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int OOB = (int)0x80000000;
constexpr int K_F32 = 0, K_F16 = 1, K_BF16 = 2;

struct SpDims {
    int n, h, w, cin, cout, ho, wo, pad_y, pad_x;
    unsigned x_bytes, w_bytes;
    // band geometry (GB / GB7 below): strip width, rows per band, LDS row pitch in 16-byte units, image pitch h + 1, strips per row
    int bw, br, brow, bhp1, strips, bpix;
    float inv_bw, inv_brow, inv_bhp1;  // reciprocals for fdiv() below
    int bpack;                         // 1: a band is `br` REAL output rows of the batch (plan kind 4), 0: `br` virtual rows (kind 3)
    int bgap;                          // kind 4: virtual rows per image beyond its ho output rows (1: "same", the zero row; 2: valid)
    float inv_ho;
};

// n / d for 0 <= n < 2^24, 0 < d < 2^24 with the reciprocal computed on the host: a float estimate and one correction step each
// way (|n * inv - n / d| < 1 for these ranges) -- 8 instructions instead of the ~35 of a run-time integer division; the band
// geometry's prologue and epilogue are full of divisions by run-time strip widths and image pitches.
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

template <int KIND>
__device__ __forceinline__ f32x16 mma_h(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (KIND == K_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const b8*>(&a), *reinterpret_cast<const b8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&a), *reinterpret_cast<const h8*>(&b), c, 0, 0, 0);
}
template <int KIND>
__device__ __forceinline__ float half_to_f32(unsigned short v) {
    if constexpr (KIND == K_BF16) return __uint_as_float((unsigned)v << 16);
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
template <int KIND>
__device__ __forceinline__ unsigned short f32_to_half(float x) {  // round to nearest even
    if constexpr (KIND == K_BF16) {
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

// 16 bytes per lane from a buffer straight into LDS: the wave's 64 lanes fill the 1 KB at `lds_wave_base` in lane order; an
// out-of-range `voffset` writes zeros.  (A __device__ function: the builtin must not be seen by the host pass.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// Block geometries.  G16: one image, 16 x 16 output pixels, 8 waves (4 x 2).  G8 (maps of at most 8 x 8, e.g. resnet layer 4 at
// 256^2 patches): TWO images of 8 x 8, 4 waves (2 x 2, wave row = image); an MFMA tile is four 8-pixel rows, and the row pitch is
// 8 mod 16 units, which puts the four rows' lanes of a ds_read_b128 group ({0-3} {12-15} {20-23} {24-27}) on 16 different bank
// groups the same way (k-space: {0-3}, 8+{4-7}, 16+{4-7}, 24+{0-3}).
// WN = waves along the channel dimension.  GBN (band geometry, 64 output channels) puts four waves along the pixels: a wave then
// owns 64 pixels x 64 channels = 2 x 2 MFMA tiles like a wave of the 128-channel kernels (32 MFMAs per tap and barrier instead of
// 16, one LDS fragment read per MFMA instead of 1.5).  Measured (profiles/r04za_n64_*.txt): +1 % on 56^2 maps, +3.6 % on
// HoVer-Net's 164^2 decoder layer; the same form of the 16 x 16 geometry was -1 % and is not kept -- the 64-channel layers' gap to
// the 128-channel ones (130 vs 146 TFLOP/s) is not a barrier-interval effect.
struct G16 {
    static constexpr int NT = 512, G = 1, TH = 16, TW = 16, PH = 18, PWD = 18, ROW = 96, IMG = 18 * 96, MROWS = 2, WAVES_M = 4, WN = 2;
    static constexpr bool BAND = false;
};
struct G8 {
    static constexpr int NT = 256, G = 2, TH = 8, TW = 8, PH = 10, PWD = 10, ROW = 56, IMG = 10 * 56, MROWS = 4, WAVES_M = 2, WN = 2;
    static constexpr bool BAND = false;
};
// Band geometry ("same" 3x3 convolutions on maps that 16 x 16 blocks cover badly: 56 / 28 / 14 / 7 of 224^2 patches).  The
// images of the batch are stacked into one tall virtual image with ONE zero row between neighbours (image pitch h + 1 rows: the
// zero row is the bottom halo of one image and the top halo of the next), cut into column strips of `bw` columns, and a block
// owns `br` consecutive virtual rows of a strip -- its br * bw <= 256 output pixels are the block's GEMM rows in row-major
// order (linearised: an MFMA tile is 32 consecutive pixels, wherever the row ends), rows >= br * bw idle.  Nothing has to divide
// anything: bands run across image boundaries (gap rows are computed and dropped: 1 / (h + 1) of the work), so a 28-wide strip
// takes br = 9 (252 of 256 rows busy) at 96.7 % (56^2 maps, two strips) / 95.0 % (28^2), 14-wide br = 18 at 91.9 %, 7-wide br = 36
// at 86.1 % -- against 76.6 % for 16 x 16 blocks on all four.  LDS row pitch = 5 bw mod 16 units (>= 5 (bw + 2)): unit address
// = 5 p + const mod 16 for the linear pixel index p, so the 16 lanes of a ds_read_b128 group (pixels p0 + {0-3, 12-15, 20-27})
// still hit 16 different bank groups for every tap shift.  That matters for the half kernels only (their MFMAs are 16 x shorter);
// float32 takes a pixel pitch of 4 units (no padding unit, 4-way conflicts on 4 reads per 32 MFMAs: 3 % of the LDS time), which
// is what lets the 7-wide band (38 rows) fit the patch buffer with two workgroups per CU.  A one-workgroup-per-CU form with a
// larger patch was measured and lost to the slice kernel (profiles/r04c_*: 106.6 vs 114.0 TFLOP/s on 512 -> 512 @ 7 x 7).
// Round 5 (plan kind 4, `bpack`): a band is `br` consecutive REAL output rows of the batch instead -- the rows between two images
// (the zero row of "same" padding, or the two extra input rows of a VALID 3x3 convolution) are in the block's LDS patch but not
// among its GEMM rows: 100 % busy on 56^2 (8 x 32) and 28^2 (4 x 64), 98.4 % on 14^2 (14 x 18) and 7^2 (7 x 36: up to five zero
// rows inside the 43-row patch), 99.6 % on HoVer-Net's valid 92 -> 90 maps (15 x 17).  Measured at 4096-patch launches (profiles/
// r05zb_band_ab.txt): 128.3 -> 131.1 / 139.4 -> 145.4 / 134.6 -> 145.5 / 127.1 (gather ring) -> 141.2 TFLOP/s on the four maps.
struct GB {
    static constexpr int NT = 512, G = 1, TH = 16, TW = 16, PH = 0, PWD = 0, ROW = 0, IMG = 1728, MROWS = 0, WAVES_M = 4, WN = 2;
    static constexpr bool BAND = true;
};
struct GBN {
    static constexpr int NT = 256, G = 1, TH = 16, TW = 16, PH = 0, PWD = 0, ROW = 0, IMG = 1728, MROWS = 0, WAVES_M = 4, WN = 1;
    static constexpr bool BAND = true;
};

// s_waitcnt vmcnt(VM) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] = 7 (no wait) | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int VM>
__device__ __forceinline__ void wait_vm_lgkm0() {
    __builtin_amdgcn_s_waitcnt((VM & 15) | (7 << 4) | ((VM >> 4) << 14));
    asm volatile("" ::: "memory");
}

// Phase timing (developer builds only: -DTIA_SP_TIMING=1, build.build(defines=...)): thread 0 of two workgroups prints the
// shader-clock cycles of set-up, first-data wait, tap loop and epilogue, and the shader clock itself (against the constant 100 MHz
// clock) -- the sustained frequency under this kernel's load, which is what the MFMA peak scales with.
#ifndef TIA_SP_TIMING
#define TIA_SP_TIMING 0
#endif
#if TIA_SP_TIMING
#define PSTAMP(i) { const long long now_ = clock64(); tm_[i] = now_ - tl_; tl_ = now_; }
#else
#define PSTAMP(i)
#endif

template <int BN, int KIND, typename GEO>
__global__ __launch_bounds__(GEO::NT, GEO::NT == 512 ? 4 : 2) void conv3x3_spatial_kernel(const void* __restrict__ x, const void* __restrict__ wk,
                                                                const float* __restrict__ bias, const void* __restrict__ res,
                                                                void* __restrict__ y, SpDims d, int relu, int m_tiles, int tiles_x,
                                                                int tiles_per_image) {
    constexpr bool F32 = KIND == K_F32;
    constexpr int ES = F32 ? 4 : 2;       // bytes per element
    constexpr int SC = 64 / ES;           // channels per 64-byte slice: 16 | 32
    constexpr int NT = GEO::NT, WN = GEO::WN, NTILE = BN / (32 * WN);
    static_assert(WN == 2 || (WN == 1 && BN == 64), "all waves along the pixels: 64-channel tiles only");
    constexpr bool BAND = GEO::BAND;
    const int PIX = BAND ? d.bpix : 5;  // units per pixel in the LDS patch
    const int ROW = BAND ? d.brow : GEO::ROW;  // (a compile-time constant for the fixed geometries)
    constexpr int BLOCK_PX = GEO::G * GEO::TH * GEO::TW;                 // 256 | 128 output pixels = GEMM rows of the block
    constexpr int A_UNITS = (GEO::G * GEO::IMG + 63) / 64 * 64;           // patch units, whole waves: 1728 | 1152
    constexpr int NA = (A_UNITS + NT - 1) / NT;                           // DMA pieces per patch: 4 | 5 (the last one partial)
    constexpr int B_UNITS = F32 ? 16 * BN / 4 : 4 * BN;                   // weight slice: 512 units (BN = 128) | 256
    constexpr int NB = (B_UNITS + NT - 1) / NT;                           // DMA pieces per weight slice: 1 | 2
    constexpr int A_BYTES = A_UNITS * 16;
    constexpr int B_BYTES = 512 * 16;     // (BN = 64: the upper half idles)
    constexpr int DUMP = 2 * A_BYTES + 3 * B_BYTES;  // 1 KB that the idle waves of the last patch piece write their zeros to
    constexpr int LDS_BYTES = DUMP + 1024;
    static_assert(NA <= 7, "the last patch piece must have landed by tap 8");
    static_assert(LDS_BYTES >= BLOCK_PX * (BN / 2) * 4, "epilogue tile");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

#if TIA_SP_TIMING
    long long tm_[4] = {0, 0, 0, 0}, tl_ = clock64();
    const long long t0c_ = tl_, t0w_ = wall_clock64();
#endif
    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int mt_id = (bid % 8) * per_xcd + bid / 8;
    if (mt_id >= m_tiles) return;
    // G16: block = (image, 16 x 16 tile); G8: block = images 2 mt_id, 2 mt_id + 1 (tiles_per_image = 1, tile origin 0)
    // band: block = (band of br virtual rows, strip); ty0 = first virtual output row
    const int img = BAND ? 0 : (GEO::G == 1 ? mt_id / tiles_per_image : mt_id * GEO::G);
    const int trem = GEO::G == 1 ? mt_id - img * tiles_per_image : 0;
    // band: virtual rows = the batch's INPUT rows stacked with image pitch bhp1 (h + 1 for "same" padding: one zero row between
    // neighbours; h for a valid convolution); an output row's virtual row = the input row of its centre ("same") / top (valid) tap.
    // ty0 = first virtual output row, vlast = the last virtual row of the patch; packed bands (d.bpack) own the real output rows
    // R0 .. R0 + br - 1 of the batch (row rr of image rr / ho), v(rr) = rr + bgap (rr / ho)
    int ty0, R0 = 0, vlast = 0;
    if constexpr (BAND) {
        const int bi = mt_id / d.strips;
        if (d.bpack) {
            R0 = bi * d.br;
            ty0 = R0 + d.bgap * fdiv(R0, d.ho, d.inv_ho);
            const int rl = min(R0 + d.br, d.n * d.ho) - 1;
            vlast = rl + d.bgap * fdiv(rl, d.ho, d.inv_ho) - d.pad_y + 2;
        } else {
            ty0 = bi * d.br;
            vlast = ty0 + d.br;
        }
    } else {
        ty0 = (trem / tiles_x) * GEO::TH;
    }
    const int tx0 = BAND ? (mt_id % d.strips) * d.bw : (trem - (trem / tiles_x) * tiles_x) * GEO::TW;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wk), 0, (int)d.w_bytes, 0x00020000);

    // patch staging: unit U = NT r + tid -> image U / IMG, row (U % IMG) / ROW, pixel (.. % ROW) / 5, unit-of-slice .. % 5 (4 = padding)
    int cen[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int u = NT * r + tid;
        if constexpr (BAND) {
            const int py = fdiv(u, ROW, d.inv_brow), rem = u - py * ROW;
            const int px = PIX == 4 ? rem >> 2 : rem / 5, chunk = rem - px * PIX;
            const int vy = ty0 - d.pad_y + py, ix = tx0 - d.pad_x + px;  // virtual input row; "same" padding: one row / column in front
            const int g = vy >= 0 ? fdiv(vy, d.bhp1, d.inv_bhp1) : 0, iy = vy - g * d.bhp1;  // image, row in it (== h: the zero row)
            const bool inside = vy >= 0 && vy <= vlast && px < d.bw + 2 && chunk < 4 && g < d.n && iy < d.h && (unsigned)ix < (unsigned)d.w;
            cen[r] = inside ? ((g * d.h + iy) * d.w + ix) * d.cin * ES + 16 * chunk : OOB;
        } else {
            const int g = u / GEO::IMG, ug = u - g * GEO::IMG;
            const int py = ug / ROW, rem = ug - py * ROW;
            const int px = rem / PIX, chunk = rem - px * PIX;
            const int iy = ty0 - d.pad_y + py, ix = tx0 - d.pad_x + px;
            const bool inside = g < GEO::G && img + g < d.n && py < GEO::PH && px < GEO::PWD && chunk < 4 && (unsigned)iy < (unsigned)d.h &&
                                (unsigned)ix < (unsigned)d.w;
            cen[r] = inside ? (((img + g) * d.h + iy) * d.w + ix) * d.cin * ES + 16 * chunk : OOB;
        }
    }
    // weight staging: one unit per thread and tap
    //   half:    [4 k-chunks][BN columns] units of 8 halves; global ((tap * cin/8 + 4 cs + kc) * cout + n0 + col) * 16
    //   float32: [16 channels][BN / 4] units of 4 columns; global ((tap * cin + 16 cs + k) * cout + n0) * 4 + 16 * colunit
    int b_off[NB], b_tap_stride, b_cs_stride;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int idx = NT * p + tid;
        if constexpr (F32) {
            const int k = idx / (BN / 4), cu = idx - k * (BN / 4);
            b_off[p] = k < 16 ? (k * d.cout + n0) * 4 + 16 * cu : OOB;
        } else {
            const int kc = idx / BN, col = idx - kc * BN;
            b_off[p] = kc < 4 ? (kc * d.cout + n0 + col) * 16 : OOB;
        }
    }
    if constexpr (F32) {
        b_tap_stride = d.cin * d.cout * 4;
        b_cs_stride = 16 * d.cout * 4;
    } else {
        b_tap_stride = (d.cin >> 3) * d.cout * 16;
        b_cs_stride = 4 * d.cout * 16;
    }
    const int n_cs = d.cin / SC, last = 9 * n_cs - 1;

    unsigned char* const abuf0 = smem;
    unsigned char* const bring = smem + 2 * A_BYTES;
    auto dma_a = [&](int buf, int r, int cs) {
        // the last piece is partial: the waves past the end of the patch (all their lanes out of range) send their zeros to the dump
        unsigned char* dst = (NT * r + wave * 64 >= A_UNITS) ? smem + DUMP : abuf0 + buf * A_BYTES + r * (NT * 16) + wave * 1024;
        dma16(rx, dst, cen[r], cs * 64);
    };
    // weight slice of flattened step s = cs * 9 + tap (clamped: the tail re-fetches the last slice)
    auto dma_b = [&](int stage, int s) {
        s = s < last ? s : last;
        const int cs = s / 9, tap = s - cs * 9;
#pragma unroll
        for (int p = 0; p < NB; ++p)
            dma16(rw, bring + stage * B_BYTES + p * (NT * 16) + wave * 1024, b_off[p], tap * b_tap_stride + cs * b_cs_stride);
    };

    f32x16 acc[2][NTILE];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // one base per lane (units): a wave owns 64 GEMM rows = 64 / TW pixel rows (G16: rows 4 wm ..; G8: image wm); MFMA row =
    // lane & 31 -> pixel row (row / TW) (+ MROWS i), column row % TW; half: k-chunk (lane >> 5) + 2 q; float32: units 2 (lane >> 5), + 1
    const int hi = lane >> 5;
    int fa[2];  // per MFMA tile of the wave: LDS unit of the lane's pixel at tap (0, 0)
    if constexpr (BAND) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = wm * 64 + i * 32 + (lane & 31);  // linear pixel of the block; idle rows read pixel 0 (results dropped)
            m = m < d.br * d.bw ? m : 0;
            int r = fdiv(m, d.bw, d.inv_bw), col = m - r * d.bw;
            if (d.bpack) {  // patch row of real row R0 + r (rows past the batch's end: pixel 0 again)
                int rr = R0 + r;
                if (rr >= d.n * d.ho) rr = R0, col = 0;
                r = rr + d.bgap * fdiv(rr, d.ho, d.inv_ho) - ty0;
            }
            fa[i] = r * ROW + col * PIX + (F32 ? 2 * hi : hi);
        }
    } else {
        const int wave_base = GEO::G == 1 ? (64 / GEO::TW) * wm * ROW : wm * GEO::IMG;
        fa[0] = wave_base + ((lane & 31) / GEO::TW) * ROW + ((lane & 31) % GEO::TW) * PIX + (F32 ? 2 * hi : hi);
        fa[1] = fa[0] + GEO::MROWS * ROW;
    }

    auto compute = [&](int buf, int stage, int tap) {
        const u32x4* sa = reinterpret_cast<const u32x4*>(abuf0 + buf * A_BYTES);
        const int shift = (tap / 3) * ROW + (tap % 3) * PIX;
        if constexpr (F32) {
            const float* sb = reinterpret_cast<const float*>(bring + stage * B_BYTES) + (8 * hi) * BN + wn * (BN / WN) + (lane & 31);
            u32x4 a[2][2];
            float b[NTILE][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][0] = sa[fa[i] + shift];
                a[i][1] = sa[fa[i] + shift + 1];
            }
#pragma unroll
            for (int j = 0; j < NTILE; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) b[j][k] = sb[k * BN + j * 32];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NTILE; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][k >> 2][k & 3]), b[j][k], acc[i][j], 0, 0, 0);
        } else {
            const u32x4* sb = reinterpret_cast<const u32x4*>(bring + stage * B_BYTES) + hi * BN + wn * (BN / WN) + (lane & 31);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 a[2], b[NTILE];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = sa[fa[i] + shift + 2 * q];
#pragma unroll
                for (int j = 0; j < NTILE; ++j) b[j] = sb[2 * q * BN + j * 32];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NTILE; ++j) acc[i][j] = mma_h<KIND>(a[i], b[j], acc[i][j]);
            }
        }
    };

    // prologue: patch of slice 0, weight slices 0 and 1
#pragma unroll
    for (int r = 0; r < NA; ++r) dma_a(0, r, 0);
    dma_b(0, 0);
    dma_b(1, 1);
    PSTAMP(0)
    wait_vm_lgkm0<NB>();  // everything but weight slice 1
    __builtin_amdgcn_s_barrier();
    PSTAMP(1)
    for (int cs = 0; cs + 1 < n_cs; ++cs) {
        const int buf = cs & 1, s0 = cs * 9;
        const int cs_next = cs + 1;
        // Per tap T: the weight slice two steps ahead goes out first (NB instructions), then (T < NA) one piece of the next patch.
        // The wait at the end lets exactly the instructions YOUNGER than weight slice s + 1 stay in flight: piece T - 1 of the patch
        // (issued right after slice s + 1), slice s + 2, piece T.  (G16: 2 3 3 3 2 1 1 1 1; G8: 3 4 4 4 4 3 2 2 2.)  A patch piece is
        // thus complete two taps after its issue, the last one by tap NA + 1 <= 8.
        // lgkmcnt(0): every LDS read of the stage refilled next has returned before the barrier.
#define TIA_TAP(T)                                                                          \
        dma_b((T + 2) % 3, s0 + T + 2);                                                     \
        if (T < NA) dma_a(buf ^ 1, T, cs_next);                                             \
        compute(buf, T % 3, T);                                                             \
        wait_vm_lgkm0<((T >= 1 && T - 1 < NA) ? 1 : 0) + NB + (T < NA ? 1 : 0)>();          \
        __builtin_amdgcn_s_barrier();
        TIA_TAP(0)
        TIA_TAP(1)
        TIA_TAP(2)
        TIA_TAP(3)
        TIA_TAP(4)
        TIA_TAP(5)
        TIA_TAP(6)
        TIA_TAP(7)
        TIA_TAP(8)
#undef TIA_TAP
    }
    {
        // the last channel slice: no patch follows (round 3 refilled the idle buffer with the same slice to keep one loop body:
        // 1 / n_cs more patch traffic -- a quarter for the 64-channel layers -- for nothing but power); only the weight ring runs on
        // (its last two requests re-fetch the last slice), so exactly slice s + 2 may stay in flight
        const int buf = (n_cs - 1) & 1, s0 = (n_cs - 1) * 9;
#define TIA_TAP_LAST(T)                                                                     \
        dma_b((T + 2) % 3, s0 + T + 2);                                                     \
        compute(buf, T % 3, T);                                                             \
        wait_vm_lgkm0<NB>();                                                                \
        __builtin_amdgcn_s_barrier();
        TIA_TAP_LAST(0)
        TIA_TAP_LAST(1)
        TIA_TAP_LAST(2)
        TIA_TAP_LAST(3)
        TIA_TAP_LAST(4)
        TIA_TAP_LAST(5)
        TIA_TAP_LAST(6)
        TIA_TAP_LAST(7)
        TIA_TAP_LAST(8)
#undef TIA_TAP_LAST
    }
    PSTAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: per column half (= the waves with wn == half): accumulators -> float32 LDS tile [BLOCK_PX][BN/2], then every
    //      thread takes rows x 8-column chunks: + bias + residual, ReLU, (round once), 16-byte stores.
    //      Row m of the block = image m / (TH TW), pixel (ty0 + (m % (TH TW)) / TW, tx0 + m % TW).
    //      Order within a half (timing build: the epilogue was 27 % of a 64-channel block's life, almost all of it memory latency):
    //      the residual and bias of ALL the thread's chunks are requested first, then the accumulators go to the tile, and only
    //      then the sums are formed -- one exposed round trip per half instead of one per chunk; the barriers around the tile wait for
    //      LDS only (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would also wait for the acknowledgement of every store). ----
    constexpr int HB = BN / 2, CHUNKS = BLOCK_PX * HB / 8, ITER = CHUNKS / NT;
    static_assert(CHUNKS % NT == 0 && NT % (HB / 8) == 0, "whole chunk rounds; a thread keeps its column chunk");
    float* tile = reinterpret_cast<float*>(smem);
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int cc = tid % (HB / 8);  // (idx = tid + NT * it: the same column chunk in every round)
    // software pipeline over the thread's chunks: PF requests in flight (all of them for the 64-channel tiles; two for the
    // 128-channel tiles, whose second-half accumulators are still in registers)
    constexpr int PF = (BN == 128 && ITER > 2) ? ((F32 && BAND) ? 1 : 2) : ITER;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int col0 = n0 + half * HB + cc * 8;
        int mpix[ITER];  // output pixel of the chunk's row (< 2^31: the callers split the batch), -1: row not on the map
        u32x4 rq[ITER][F32 ? 2 : 1];
        float4 b0 = float4{0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0;
        if (bias) {
            b0 = *reinterpret_cast<const float4*>(bias + col0);
            b1 = *reinterpret_cast<const float4*>(bias + col0 + 4);
        }
        auto request = [&](int it) {
            const int row = (tid + NT * it) / (HB / 8);
            int g, oy, ox;
            bool live;
            if constexpr (BAND) {
                const int r = fdiv(row, d.bw, d.inv_bw);
                ox = tx0 + (row - r * d.bw);
                if (d.bpack) {
                    g = fdiv(R0 + r, d.ho, d.inv_ho);
                    oy = R0 + r - g * d.ho;
                    live = row < d.br * d.bw && g < d.n;
                } else {
                    const int vy = ty0 + r;
                    g = fdiv(vy, d.bhp1, d.inv_bhp1);
                    oy = vy - g * d.bhp1;
                    live = row < d.br * d.bw && oy < d.ho && g < d.n;  // (oy == ho: the zero row between two images)
                }
            } else {
                g = row / (GEO::TH * GEO::TW);
                const int rg = row - g * (GEO::TH * GEO::TW);
                oy = ty0 + rg / GEO::TW;
                ox = tx0 + rg % GEO::TW;
                live = oy < d.ho && ox < d.wo && img + g < d.n;
            }
            mpix[it] = live ? ((img + g) * d.ho + oy) * d.wo + ox : -1;
#pragma unroll
            for (int q = 0; q < (F32 ? 2 : 1); ++q) rq[it][q] = u32x4{0u, 0u, 0u, 0u};
            if (res && live) {
                const long off = (long)mpix[it] * d.cout + col0;
                if constexpr (F32) {
                    const u32x4* rp = reinterpret_cast<const u32x4*>(static_cast<const float*>(res) + off);
                    rq[it][0] = rp[0];
                    rq[it][1] = rp[1];
                } else {
                    rq[it][0] = *reinterpret_cast<const u32x4*>(static_cast<const unsigned short*>(res) + off);
                }
            }
        };
#pragma unroll
        for (int it = 0; it < PF; ++it) request(it);
        if (WN == 1 || wn == half) {
            // WN == 2: the waves of this column half hold it in all their tiles; WN == 1: every wave holds it in tiles half * NH ..
            constexpr int NH = WN == 2 ? NTILE : NTILE / 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NH; ++jj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        tile[row * HB + jj * 32 + (lane & 31)] = acc[i][WN == 2 ? jj : half * NH + jj][e];
                    }
        }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            if (it + PF < ITER) request(it + PF);
            const int row = (tid + NT * it) / (HB / 8);
            const float4 v0 = *reinterpret_cast<const float4*>(tile + row * HB + cc * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(tile + row * HB + cc * 8 + 4);
            float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
            if (mpix[it] >= 0) {
                const long off = (long)mpix[it] * d.cout + col0;
                if constexpr (F32) {
                    float* yo = static_cast<float*>(y) + off;
                    if (res) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[k] += __uint_as_float(rq[it][0][k]);
                            v[4 + k] += __uint_as_float(rq[it][1][k]);
                        }
                    }
                    if (relu) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
                    }
                    *reinterpret_cast<float4*>(yo) = float4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<float4*>(yo + 4) = float4{v[4], v[5], v[6], v[7]};
                } else {
                    unsigned short* yh = static_cast<unsigned short*>(y);
                    if (res) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[2 * k] += half_to_f32<KIND>((unsigned short)(rq[it][0][k] & 0xffffu));
                            v[2 * k + 1] += half_to_f32<KIND>((unsigned short)(rq[it][0][k] >> 16));
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
                        o[k] = (unsigned)f32_to_half<KIND>(a0) | ((unsigned)f32_to_half<KIND>(a1) << 16);
                    }
                    *reinterpret_cast<u32x4*>(yh + off) = u32x4{o[0], o[1], o[2], o[3]};
                }
            }
        }
        if (half == 0) lds_barrier();  // the tile is re-used by the second half
    }
#if TIA_SP_TIMING
    PSTAMP(3)
    if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 64 || blockIdx.x == 3001))
        printf("spatial wg %d (BN %d, NT %d, cin %d): setup %lld  first-data wait %lld  tap loop %lld  epilogue %lld  | shader clock %.0f MHz\n",
               (int)blockIdx.x, BN, NT, d.cin, tm_[0], tm_[1], tm_[2], tm_[3],
               100.0 * (double)(clock64() - t0c_) / (double)(wall_clock64() - t0w_));
#endif
}


// ---- 1x1 convolutions (float32): the same machinery without taps ---------------------------------------------------------------
// A plain GEMM over 256 consecutive output pixels x BN channels: per 16-channel slice the pixels' 64-byte runs (pixel pitch 5
// units, the same bank argument) and the [16][BN] weight block arrive by LDS-DMA into the idle one of two stages while the eight
// waves issue the 32 MFMAs of the current one; one barrier and one vmcnt(0) per slice (the DMA has a whole MFMA phase to land)
// instead of the slice kernel's register staging and barrier pair.  Strided 1x1 (ResNet down-sampling) only changes which input
// pixel an output pixel reads.
// The same ring also GATHERS: for a kh x kw convolution (any stride, front padding < kernel) the reduction runs over (tap, 16-channel
// slice) in the slice kernel's order, a slice's DMA address is the pixel's tap-(0, 0) offset plus a scalar tap delta, and a tap
// outside the image is an out-of-range offset (the DMA writes zeros) chosen with one mask test per unit -- the implicit GEMM for
// the layers the tap-reuse kernel does not serve (stride-2 3x3, maps 16 x 16 blocks cover badly), bit-identical to the slice kernel.
struct PwDims {
    int n, h, w, cin, cout, ho, wo, stride;
    unsigned x_bytes, w_bytes;
    int kh, kw, pad_y, pad_x;
};

template <int BN>
__global__ __launch_bounds__(512, 4) void conv1x1_ring_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             float* __restrict__ y, PwDims d, int relu, int m_tiles) {
    constexpr int NT = 512, NTILE = BN / 64, PIX = 5;
    constexpr int A_UNITS = 256 * PIX;   // 1280 units: two whole DMA rounds of 512 + 256
    constexpr int A_BYTES = A_UNITS * 16;
    constexpr int B_BYTES = NT * 16;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int DUMP = 2 * STAGE;
    constexpr int EPI = 256 * (BN / 2) * 4;
    constexpr int LDS_BYTES = (DUMP + 1024) > EPI ? (DUMP + 1024) : EPI;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int mt_id = (bid % 8) * per_xcd + bid / 8;
    if (mt_id >= m_tiles) return;
    const long m0 = (long)mt_id * 256;
    const long m_total = (long)d.n * d.ho * d.wo;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wk), 0, (int)d.w_bytes, 0x00020000);

    // per DMA unit: byte offset of tap (0, 0) of its pixel (may lie before the buffer: only used when the tap is inside the image)
    // and one bit per kernel row / column saying whether that row / column of taps is inside (all set for a 1x1)
    int cen[3];
    unsigned msk[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int u = NT * r + tid;
        const int p = u / PIX, chunk = u - p * PIX;
        const long m = m0 + p;
        const bool ok = p < 256 && chunk < 4 && m < m_total;
        const int mm = ok ? (int)m : 0;
        const int b = mm / (d.ho * d.wo), rem = mm - b * d.ho * d.wo;
        const int oy = rem / d.wo, ox = rem - oy * d.wo;
        const int iy0 = oy * d.stride - d.pad_y, ix0 = ox * d.stride - d.pad_x;
        cen[r] = (((b * d.h + iy0) * d.w + ix0) * d.cin) * 4 + 16 * chunk;
        unsigned rows = 0, cols = 0;
        for (int t = 0; t < d.kh; ++t) rows |= (unsigned)((unsigned)(iy0 + t) < (unsigned)d.h) << t;
        for (int t = 0; t < d.kw; ++t) cols |= (unsigned)((unsigned)(ix0 + t) < (unsigned)d.w) << (16 + t);
        msk[r] = ok ? (rows | cols) : 0u;
    }
    const int bk = tid / (BN / 4), bcu = tid - bk * (BN / 4);
    const int b_off = bk < 16 ? (bk * d.cout + n0) * 4 + 16 * bcu : OOB;
    const int n_cs = d.cin >> 4;
    const int n_slices = d.kh * d.kw * n_cs;

    // slice cursor (scalar): tap (s_kh, s_kw), channel slice s_cs of the slice that is requested next
    int s_kh = 0, s_kw = 0, s_cs = 0;
    auto dma_stage = [&](int stage) {
        unsigned char* sa = smem + stage * STAGE;
        const int sdelta = (s_kh * d.w + s_kw) * d.cin * 4;
        const unsigned sel = (1u << s_kh) | (1u << (16 + s_kw));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            unsigned char* dst = (NT * r + wave * 64 >= A_UNITS) ? smem + DUMP : sa + r * (NT * 16) + wave * 1024;
            dma16(rx, dst, (msk[r] & sel) == sel ? cen[r] + sdelta : OOB, s_cs * 64);
        }
        dma16(rw, sa + A_BYTES + wave * 1024, b_off, (((s_kh * d.kw + s_kw) * d.cin) + s_cs * 16) * d.cout * 4);
    };
    // past the last slice the cursor stays there: the idle stage is refilled with the same slice
    auto next_slice = [&]() {
        int cs = s_cs + 1, kw = s_kw, kh = s_kh;
        if (cs == n_cs) { cs = 0; ++kw; }
        if (kw == d.kw) { kw = 0; ++kh; }
        if (kh < d.kh) { s_cs = cs; s_kw = kw; s_kh = kh; }
    };

    f32x16 acc[2][NTILE];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int hi = lane >> 5;
    const int fa0 = (wm * 64 + (lane & 31)) * PIX + 2 * hi;  // MFMA row = pixel wm * 64 + 32 i + (lane & 31); channels 8 hi .. 8 hi + 7
    const int fb0 = (8 * hi) * BN + wn * (BN / 2) + (lane & 31);

    dma_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < n_slices; ++it) {
        const int stage = it & 1;
        next_slice();
        dma_stage(stage ^ 1);
        const u32x4* sa = reinterpret_cast<const u32x4*>(smem + stage * STAGE) + fa0;
        const float* sb = reinterpret_cast<const float*>(smem + stage * STAGE + A_BYTES) + fb0;
        u32x4 a[2][2];
        float b[NTILE][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i][0] = sa[i * 32 * PIX];
            a[i][1] = sa[i * 32 * PIX + 1];
        }
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) b[j][k] = sb[k * BN + j * 32];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTILE; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][k >> 2][k & 3]), b[j][k], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __syncthreads();

    constexpr int HB = BN / 2, CHUNKS = 256 * HB / 8;
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wn == half) {
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
            const long m = m0 + row;
            if (m < m_total) {
                const int col0 = n0 + half * HB + cc * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(tile + row * HB + cc * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(tile + row * HB + cc * 8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias + col0), b1 = *reinterpret_cast<const float4*>(bias + col0 + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                float* yo = y + m * d.cout + col0;
                if (res) {
                    const float* rp = res + m * d.cout + col0;
                    const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                    v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                }
                if (relu) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
                }
                *reinterpret_cast<float4*>(yo) = float4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4*>(yo + 4) = float4{v[4], v[5], v[6], v[7]};
            }
        }
        __syncthreads();
    }
}

}  // namespace

namespace tia {

// Compute units of the calling thread's CURRENT device, cached per device index (a process may drive several GPUs; the first caller
// may be the host-only route query).  Without a usable device (build container): MI355X's 256.
static long device_cu_count() {
    static std::atomic<int> cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int cus = cached[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
        cached[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

bool conv3x3_spatial_serves(long nb, long h, long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int dtype) {
    static const bool disabled = tia::dev_env("TIA_CONV_NO_SPATIAL") != nullptr;
    const int es = dtype == TIA_DT_F32 ? 4 : 2;
    if (disabled || cin % (64 / es) != 0 || cout % 64 != 0 || pad_top > 2 || pad_left > 2) return false;
    const SpPlan plan = conv3x3_spatial_plan(3, 3, 1, h, w, ho, wo, pad_top, pad_left, dtype == TIA_DT_F32);
    if (plan.kind == 0) return false;
    if (plan.kind >= 3 && nb * (h + 1) + 256 >= (1L << 24)) return false;  // fdiv() range of the band geometry (rows of the last band included; the callers keep the input below 2 GiB)
    if (plan.kind >= 3) {
        // A launch of between one and two rounds of workgroups (two per CU) leaves the second round mostly empty, and the band
        // blocks are the coarsest work units of all the convolution kernels: 512 -> 512 on 7 x 7 maps of a 1024-patch batch is
        // 800 workgroups = 1.56 rounds -- measured 113.7 TFLOP/s against 120.2 on the slice kernel (1,568 smaller workgroups),
        // while the 4096-patch batch runs at 141.2 against 127.1 (profiles/r05zb_band_ab.txt).  One round, or three and more, are fine.
        const long bands = plan.kind == 4 ? (nb * ho + plan.br - 1) / plan.br : (nb * (h + 1) - 1 + plan.br - 1) / plan.br;
        const long wgs = bands * plan.strips * (cout / (cout % 128 == 0 ? 128 : 64)), slots = 2 * device_cu_count();
        if (wgs > slots && wgs <= 2 * slots && wgs * 100 < 2 * slots * 85) return false;
    }
    return true;
}

bool conv3x3_spatial_launch(const void* x, const void* w_packed, const float* bias, const void* residual, void* y, long nb, long h,
                            long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int dtype, int relu,
                            hipStream_t stream) {
    const int es = dtype == TIA_DT_F32 ? 4 : 2;
    if (!conv3x3_spatial_serves(nb, h, w, cin, cout, pad_top, pad_left, ho, wo, dtype)) return false;
    const SpPlan plan = conv3x3_spatial_plan(3, 3, 1, h, w, ho, wo, pad_top, pad_left, dtype == TIA_DT_F32);
    const bool small = plan.kind == 2;  // G8: two images of (at most) 8 x 8 per block
    const bool band = plan.kind >= 3;
    const long tiles_y = small ? 1 : (ho + 15) / 16, tiles_x = small ? 1 : (wo + 15) / 16;
    // band: the batch as one image of nb * (h + 1) - 1 rows (no zero row behind the last image), cut into bands of br rows
    // (kind 4: nb * h real rows in bands of br)
    const bool pack = plan.kind == 4;
    const long tiles = pack ? ((nb * ho + plan.br - 1) / plan.br) * plan.strips
                       : band ? ((nb * (h + 1) - 1 + plan.br - 1) / plan.br) * plan.strips : (small ? (nb + 1) / 2 : nb * tiles_y * tiles_x);
    const SpDims d{(int)nb, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)pad_top, (int)pad_left,
                   (unsigned)(nb * h * w * cin * es), (unsigned)(9 * cin * cout * es), plan.bw, plan.br, plan.brow, (int)(h + pad_top), plan.strips,
                   plan.bpix, band ? 1.0f / (float)plan.bw : 0.0f, band ? 1.0f / (float)plan.brow : 0.0f, 1.0f / (float)(h + pad_top),
                   pack ? 1 : 0, (int)(h + pad_top - ho), 1.0f / (float)ho};
    // 128-channel column tiles, unless that leaves fewer workgroups than the device has CUs (small batches of small maps: UNet-R50's
    // 512 -> 512 @ 32^2 at batch 8 is 128 workgroups for 256 CUs: 73 TFLOP/s): then 64-channel tiles, twice as many workgroups
    static const bool no_narrow = tia::dev_env("TIA_CONV_NO_NARROW_FILL") != nullptr;  // developer switch (A/B measurements)
    const bool wide = cout % 128 == 0 && (no_narrow || tiles * (cout / 128) >= device_cu_count());
    static const bool wide8 = tia::dev_env("TIA_CONV_N64_8WAVES") != nullptr;  // developer switch: 64-channel band tiles on the 8-wave (4 x 2) form
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8), (unsigned)(cout / (wide ? 128 : 64)));
#define TIA_LAUNCH_GEO(BN_, KIND_, GEO_)                                                                                           \
    hipLaunchKernelGGL((conv3x3_spatial_kernel<BN_, KIND_, GEO_>), grid, dim3(GEO_::NT), 0, stream, x, w_packed, bias, residual, y, d, \
                       relu, (int)tiles, (int)tiles_x, (int)(tiles_y * tiles_x))
#define TIA_LAUNCH_SP(BN_, KIND_)                                                                                                  \
    do {                                                                                                                           \
        switch (plan.kind) {                                                                                                       \
            case 1: TIA_LAUNCH_GEO(BN_, KIND_, G16); break;                                                                        \
            case 2: TIA_LAUNCH_GEO(BN_, KIND_, G8); break;                                                                         \
            default: if (BN_ == 64 && !wide8) TIA_LAUNCH_GEO(64, KIND_, GBN); else TIA_LAUNCH_GEO(BN_, KIND_, GB); break;          \
        }                                                                                                                          \
    } while (0)
    if (dtype == TIA_DT_F32) {
        if (wide) TIA_LAUNCH_SP(128, K_F32); else TIA_LAUNCH_SP(64, K_F32);
    } else if (dtype == TIA_DT_F16) {
        if (wide) TIA_LAUNCH_SP(128, K_F16); else TIA_LAUNCH_SP(64, K_F16);
    } else {
        if (wide) TIA_LAUNCH_SP(128, K_BF16); else TIA_LAUNCH_SP(64, K_BF16);
    }
#undef TIA_LAUNCH_SP
#undef TIA_LAUNCH_GEO
    return true;
}

bool conv_ring_ok(long nb, long cin, long cout, long kh, long kw, long ho, long wo) {
    static const bool disabled = tia::dev_env("TIA_CONV_NO_RING") != nullptr;
    static const bool no_taps = tia::dev_env("TIA_CONV_NO_GATHER_RING") != nullptr;  // developer switch (A/B measurements)
    if (disabled || cin % 16 != 0 || cout % 128 != 0 || kh > 16 || kw > 16) return false;
    if ((kh != 1 || kw != 1) && no_taps) return false;
    const long m_total = nb * ho * wo, tiles = (m_total + 255) / 256;
    // measured (profiles/r03u_unet_layers*.txt): with 64 output channels (half the MFMAs per barrier) and with fewer than ~1.5
    // workgroups per CU (256-pixel blocks: small maps at small batches) the slice kernel is the faster one
    if (tiles * (cout / 128) < 384) return false;
    if (kh != 1 || kw != 1) {
        // gathering taps, the ring beats the slice kernel only while its rounds of 2 workgroups per CU are full (measured,
        // profiles/r04y_*_probe.txt: +5..+28 % at >= 0.875 full, -3..-14 % at 0.77: 7 x 7 outputs of a 1024-patch batch)
        const long slots = 2 * device_cu_count();
        const long wgs = tiles * (cout / 128), rounds = (wgs + slots - 1) / slots;
        if (wgs * 100 < rounds * slots * 85) return false;
    }
    return true;
}

bool conv_ring_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, long nb, long h, long w,
                      long cin, long cout, long kh, long kw, long stride, long pad_top, long pad_left, long ho, long wo, int relu,
                      hipStream_t stream) {
    if (!conv_ring_ok(nb, cin, cout, kh, kw, ho, wo)) return false;
    const long tiles = (nb * ho * wo + 255) / 256;
    const PwDims d{(int)nb, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)stride, (unsigned)(nb * h * w * cin * 4),
                   (unsigned)(kh * kw * cin * cout * 4), (int)kh, (int)kw, (int)pad_top, (int)pad_left};
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8), (unsigned)(cout / 128));
    hipLaunchKernelGGL(conv1x1_ring_kernel<128>, grid, dim3(512), 0, stream, x, w_packed, bias, residual, y, d, relu, (int)tiles);
    return true;
}

bool conv1x1_ring_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, long nb, long h, long w,
                         long cin, long cout, long stride, long ho, long wo, int relu, hipStream_t stream) {
    return conv_ring_launch(x, w_packed, bias, residual, y, nb, h, w, cin, cout, 1, 1, stride, 0, 0, ho, wo, relu, stream);
}

}  // namespace tia
