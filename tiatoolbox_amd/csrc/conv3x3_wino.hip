// Winograd F(2x2, 3x3) form of the 3x3 / stride-1 NHWC convolution in float32 on the gfx950 matrix cores -- 16 multiplies per
// 2 x 2 outputs and (cin, cout) pair instead of 36, i.e. 2.25 x fewer MFMA operations than the direct tap-reuse kernel
// (conv3x3_spatial.hip), which sits at ~90 % of a float32 MFMA wall that equals the vector rate.  float32 Winograd is what the
// vendor libraries run for exactly these layers, so it is the reference's arithmetic CLASS (float32 in, float32 accumulate), not
// its operation order: results differ from a direct float32 convolution in the last bits (measured and bounded in
// tests/test_engine.py; the engines take this path only when asked: `conv_algo="winograd"`).
// Call sites in the reference: the 3x3 convolutions behind CNNModel.forward (models/architecture/vanilla.py:242-253,300-316).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A     per 2 x 2 output tile, 4 x 4 input tile d, 3 x 3 filter g   (Lavin & Gray)
//
// * weights: U = G g G^T computed ONCE at pack time in float64, rounded to float32 (tia_conv_pack_weights_wino_f32), stored in the
//   exact LDS image of a weight stage: [pos 16][cin/16][cout/64] blocks of 4 KB = [hi 2][kq 2][64 cout][4 channels]
//   (channel = 16 cs + 8 hi + 4 kq + c4): a stage is a plain contiguous LDS-DMA copy, a lane's 8 k-values of a position are two
//   conflict-free ds_read_b128.
// * a 512-thread workgroup owns 64 tiles (16 x 16 output pixels of one image, or four images of <= 8 x 8) x 64 output channels
//   x all 16 Winograd positions; 8 waves = 2 position groups (rows i in {0, 1} / {2, 3} of the 4 x 4 position grid) x 2 tile halves
//   x 2 channel halves; a wave keeps 8 positions x (32 tiles x 32 channels) = 128 accumulator registers.  One workgroup per CU
//   (two waves per SIMD, <= 256 registers each).
// * the raw 18 x 18 (4 x 10 x 10) input patch of a 16-channel slice arrives by LDS-DMA exactly as in the direct kernel
//   (pixel pitch 5 units, double-buffered); the input transform V = B^T d B is done IN REGISTERS on the way to the MFMA: per
//   step (one position row i, four positions j) a lane reads two patch rows x four columns of its tile (16 ds_read_b128),
//   forms R_i = d[ra] +- d[rb] and V_ij = R[c] +- R[c'] (64 adds) and feeds 32 MFMAs -- no transformed tensor ever exists in
//   memory (the non-fused form moves 4 x the input and 4 x the output through HBM and loses to the direct kernel).
// * per step a stage of 8 positions x 4 KB = 32 KB of weights (both position groups) streams in behind the MFMAs (two stages);
//   one barrier + vmcnt(0) per step = per 32 MFMAs of a wave.
// * epilogue: output transform A^T M A in registers per position group, the two groups' partial sums meet in the LDS tile
//   [256 pixels][64 channels] (group 0 stores, group 1 adds), then + bias + residual, ReLU, 16-byte stores as in the direct kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tiatoolbox_amd.h"
#include "conv3x3_wino.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int OOB = (int)0x80000000;

struct WinoDims {
    int n, h, w, cin, cout, ho, wo, pad_y, pad_x;
    unsigned x_bytes, u_bytes;
    int pos_stride;  // bytes between consecutive positions of the packed weights: (cin / 16) * (cout / 64) * 4096
    int abl;         // timing builds only (TIA_WINO_ABL): 1 no weight DMA in the loop, 2 no patch DMA, 4 weight DMA out of range (zeros), 8 half of it
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// W16: one image, 16 x 16 output pixels = 8 x 8 tiles, patch 18 x 18, row pitch 96 units (2 * ROW = 0 mod 16: the 16 lanes of a
// ds_read_b128 group fall on every even 16-byte bank group exactly twice -- a tile's pixels are two apart, so an inherent 2-way).
// W8: FOUR images of at most 8 x 8 = 4 x (4 x 4) tiles, patch 10 x 10 each, row pitch 52 (= 4 mod 8: the same 2-way argument
// across tile rows and images), image pitch 520.
struct W16 {
    static constexpr int G = 1, TH = 16, TW = 16, PH = 18, PWD = 18, ROW = 96, IMG = 18 * 96;
};
struct W8 {
    static constexpr int G = 4, TH = 8, TW = 8, PH = 10, PWD = 10, ROW = 52, IMG = 10 * 52;
};

// s_waitcnt vmcnt(VM) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] = 7 (no wait) | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int VM>
__device__ __forceinline__ void wait_vm_lgkm0() {
    __builtin_amdgcn_s_waitcnt((VM & 15) | (7 << 4) | ((VM >> 4) << 14));
    asm volatile("" ::: "memory");
}

// Phase timing (developer builds only: -DTIA_WINO_TIMING=1): thread 0 of two workgroups prints shader-clock cycles of the prologue,
// of the steps' compute and of their waits (vmcnt + barrier), of the epilogue, and the shader clock against the 100 MHz clock.
#ifndef TIA_WINO_TIMING
#define TIA_WINO_TIMING 0
#endif
#if TIA_WINO_TIMING
#define WSTAMP(var) { const long long now_ = clock64(); var += now_ - tl_; tl_ = now_; }
#else
#define WSTAMP(var)
#endif

template <typename GEO, int NSTAGE>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                              const float* __restrict__ bias, const float* __restrict__ res,
                                                              float* __restrict__ y, WinoDims d, int relu, int m_tiles, int tiles_x,
                                                              int tiles_per_image) {
    constexpr int NT = 512, PIX = 5, ROW = GEO::ROW, BN = 64;
    constexpr int BLOCK_PX = GEO::G * GEO::TH * GEO::TW;          // 256 output pixels = 64 tiles
    constexpr int A_UNITS = (GEO::G * GEO::IMG + 63) / 64 * 64;   // patch units (16 bytes), whole waves: 1728 | 2112
    constexpr int NA = (A_UNITS + NT - 1) / NT;                   // DMA pieces per patch: 4 | 5 (the last one partial)
    constexpr int A_BYTES = A_UNITS * 16;
    constexpr int W_STAGE = 8 * 4096;                             // 8 positions x [16 channels][64 columns] float32
    constexpr int DUMP = 2 * A_BYTES + NSTAGE * W_STAGE;          // 1 KB the idle waves of the last patch piece write their zeros to
    constexpr int EPI_BYTES = 2 * BLOCK_PX * BN * 4;              // the epilogue's two float32 tiles (one per position group)
    constexpr int LDS_BYTES = DUMP + 1024 > EPI_BYTES ? DUMP + 1024 : EPI_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(NA >= 2 && NA <= 6, "patch pieces are spread over the two steps of a slice");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "weight ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if TIA_WINO_TIMING
    long long tm_pro = 0, tm_comp = 0, tm_wait = 0, tm_epi = 0, tl_ = clock64();
    const long long t0c_ = tl_, t0w_ = wall_clock64();
#endif

    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int mt_id = (bid % 8) * per_xcd + bid / 8;  // every XCD walks a contiguous range of pixel blocks
    if (mt_id >= m_tiles) return;
    const int img = GEO::G == 1 ? mt_id / tiles_per_image : mt_id * GEO::G;
    const int trem = GEO::G == 1 ? mt_id - img * tiles_per_image : 0;
    const int ty0 = (trem / tiles_x) * GEO::TH;
    const int tx0 = (trem - (trem / tiles_x) * tiles_x) * GEO::TW;
    const int cb = blockIdx.y, n0 = cb * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;  // position group, tile half, channel half
    const int hi = lane >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u), 0, (int)d.u_bytes, 0x00020000);

    // patch staging (as the direct kernel): unit U = NT r + tid -> image U / IMG, row (U % IMG) / ROW, pixel (.. % ROW) / 5, unit-of-slice
    // .. % 5 (4 = padding); outside the image / patch: an out-of-range offset (the DMA writes zeros)
    int cen[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int un = NT * r + tid;
        const int g = un / GEO::IMG, ug = un - g * GEO::IMG;
        const int py = ug / ROW, rem = ug - py * ROW;
        const int px = rem / PIX, chunk = rem - px * PIX;
        const int iy = ty0 - d.pad_y + py, ix = tx0 - d.pad_x + px;
        const bool inside = g < GEO::G && img + g < d.n && py < GEO::PH && px < GEO::PWD && chunk < 4 && (unsigned)iy < (unsigned)d.h &&
                            (unsigned)ix < (unsigned)d.w;
        cen[r] = inside ? (((img + g) * d.h + iy) * d.w + ix) * d.cin * 4 + 16 * chunk : OOB;
    }
    const int n_cs = d.cin >> 4, n_cb = d.cout >> 6;
    // weight staging: a stage = 8 blocks of 4 KB in the order [pg][j]; DMA round q (0..3) moves blocks (pg = q >> 1, j = 2 (q & 1) +
    // (wave >> 2)): per lane the offset inside the block + (wave >> 2) positions; the rest is scalar
    const int w_voff = (wave & 3) * 1024 + lane * 16 + (wave >> 2) * d.pos_stride;

    unsigned char* const abuf0 = smem;
    unsigned char* const wst0 = smem + 2 * A_BYTES;
    auto dma_a = [&](int buf, int r, int cs) {
#if TIA_WINO_TIMING
        if ((d.abl & 2) && cs > 0) return;
#endif
        unsigned char* dst = (NT * r + wave * 64 >= A_UNITS) ? smem + DUMP : abuf0 + buf * A_BYTES + r * (NT * 16) + wave * 1024;
        dma16(rx, dst, cen[r], cs * 64);
    };
    // weights of flattened step s = 2 cs + half
    auto dma_w = [&](int stage, int s) {
        const int cs = s >> 1, half = s & 1;
#if TIA_WINO_TIMING
        if ((d.abl & 1) && s > 0) return;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 2 * (q >> 1) + half, j0 = 2 * (q & 1);
            const int soff = (i * 4 + j0) * d.pos_stride + (cs * n_cb + cb) * 4096;
#if TIA_WINO_TIMING
            if ((d.abl & 8) && (q & 1) && s > 0) continue;
            dma16(ru, wst0 + stage * W_STAGE + q * 8192 + wave * 1024, ((d.abl & 4) && s > 0) ? OOB : w_voff, soff);
#else
            dma16(ru, wst0 + stage * W_STAGE + q * 8192 + wave * 1024, w_voff, soff);
#endif
        }
    };

    f32x16 acc[2][4];  // [row of the group: i = 2 pg + half][j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.0f;

    // the lane's tile: MFMA row = lane & 31 -> tile 32 wm + (lane & 31); its 4 x 4 input tile starts at patch pixel (2 ty, 2 tx)
    const int t = 32 * wm + (lane & 31);
    int fa;
    if constexpr (GEO::G == 1) {
        fa = 2 * (t >> 3) * ROW + 2 * (t & 7) * PIX + 2 * hi;
    } else {
        fa = (t >> 4) * GEO::IMG + 2 * ((t >> 2) & 3) * ROW + 2 * (t & 3) * PIX + 2 * hi;
    }
    // row transform of position row i: R = d[ra] + sg * d[rb]   (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
    //   i = 0: d0 - d2    1: d1 + d2    2: d2 - d1    3: d1 - d3
    int ra_u[2], rb_u[2];
    float sg[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int i = 2 * pg + half;
        const int ra = i == 0 ? 0 : (i == 2 ? 2 : 1), rb = i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
        ra_u[half] = __builtin_amdgcn_readfirstlane(ra * ROW);
        rb_u[half] = __builtin_amdgcn_readfirstlane(rb * ROW);
        sg[half] = i == 1 ? 1.0f : -1.0f;
    }
    // weights of the lane: block (pg, j) of the stage, units [hi][kq][column]: two 16-byte reads per position
    const int fb = pg * 4 * 256 + hi * 128 + wn * 32 + (lane & 31);

    // One step = position row i of the group, positions j = 0..3: 16 patch reads (two rows x four columns x two 16-byte units),
    // 8 weight reads, 64 adds, 32 MFMAs.  Software-pipelined so that no LDS round trip is exposed inside the step (the first
    // version read each position's operands right in front of its MFMAs: eight exposed waits per step, matrix pipe 50-65 % busy):
    //   request columns 0 and 2 (what V_0 = R0 - R2 needs) + weights of j = 0, then column 1 + weights of j = 1;
    //   behind the eight MFMAs of j = 0: request column 3 and the weights of j = 2, 3; form R1, V1;
    //   j = 1 .. 3 then run out of registers.  sched_group_barrier pins that order.
    auto compute = [&](int buf, int stage, auto half_c) {
        constexpr int HALF = decltype(half_c)::value;
        const u32x4* sa = reinterpret_cast<const u32x4*>(abuf0 + buf * A_BYTES) + fa;
        const u32x4* sb = reinterpret_cast<const u32x4*>(wst0 + stage * W_STAGE) + fb;
        const int ra = ra_u[HALF], rb = rb_u[HALF];
        const float sgn = sg[HALF];
        u32x4 pa[4][2], pb[4][2], wq[4][2];
        auto load_col = [&](int c) {
            pa[c][0] = sa[ra + c * PIX], pa[c][1] = sa[ra + c * PIX + 1];
            pb[c][0] = sa[rb + c * PIX], pb[c][1] = sa[rb + c * PIX + 1];
        };
        auto load_w = [&](int j) { wq[j][0] = sb[j * 256], wq[j][1] = sb[j * 256 + 64]; };
        float R[4][8];
        auto row_tf = [&](int c) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                R[c][k] = __builtin_fmaf(__uint_as_float(pb[c][k >> 2][k & 3]), sgn, __uint_as_float(pa[c][k >> 2][k & 3]));
        };
        auto mma8 = [&](int j, const float (&v)[8]) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc[HALF][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[k], __uint_as_float(wq[j][k >> 2][k & 3]), acc[HALF][j], 0, 0, 0);
        };
        __builtin_amdgcn_sched_barrier(0);
        load_col(0), load_col(2), load_w(0);   // 10 reads
        load_col(1), load_w(1);                // 6 reads
        row_tf(0), row_tf(2);
        float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v0[k] = R[0][k] - R[2][k];
        mma8(0, v0);
        load_col(3), load_w(2), load_w(3);     // 8 reads, behind the MFMAs of j = 0
        row_tf(1);
#pragma unroll
        for (int k = 0; k < 8; ++k) v1[k] = R[1][k] + R[2][k];
        mma8(1, v1);
#pragma unroll
        for (int k = 0; k < 8; ++k) v2[k] = R[2][k] - R[1][k];
        mma8(2, v2);
        row_tf(3);
#pragma unroll
        for (int k = 0; k < 8; ++k) v3[k] = R[1][k] - R[3][k];
        mma8(3, v3);
        // order for the scheduler (masks: 0x100 DS read, 0x002 VALU, 0x008 MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);  // columns 0, 2, 1 + weights 0, 1
        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);  // R0, R2, V0
#pragma unroll
        for (int k = 0; k < 8; ++k) {                        // j = 0: each MFMA followed by one of the 8 late reads and two of the
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 16 VALU operations of R1, V1
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                        // j = 1 (+ V2)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                        // j = 2 (+ R3, V3)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // j = 3
        __builtin_amdgcn_sched_barrier(0);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
#if TIA_WINO_TIMING
    long long tm_vm = 0;
#endif
    auto step_end = [&] {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if TIA_WINO_TIMING
        { const long long now_ = clock64(); tm_vm += now_ - tl_; }
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if constexpr (NSTAGE == 2) {
        // two weight stages: the weights of step s + 1 and the next slice's patch are requested at the start of step s and must have
        // landed by its end (vmcnt(0)): one step of look-ahead
#pragma unroll
        for (int r = 0; r < NA; ++r) dma_a(0, r, 0);
        dma_w(0, 0);
        step_end();
        WSTAMP(tm_pro)
        constexpr int NA0 = (NA + 1) / 2;  // patch pieces issued in the first step of a slice; the rest in the second
        for (int cs = 0; cs < n_cs; ++cs) {
            const int buf = cs & 1;
            const bool more = cs + 1 < n_cs;
            // step 2 cs (rows i = 0 / 2): weights of step 2 cs + 1 -> stage 1, the first pieces of the next slice's patch
            dma_w(1, 2 * cs + 1);
            if (more) {
#pragma unroll
                for (int r = 0; r < NA0; ++r) dma_a(buf ^ 1, r, cs + 1);
            }
            compute(buf, 0, H0{});
            WSTAMP(tm_comp)
            step_end();
            WSTAMP(tm_wait)
            // step 2 cs + 1 (rows i = 1 / 3): weights of step 2 cs + 2 -> stage 0, the remaining pieces
            if (more) {
                dma_w(0, 2 * cs + 2);
#pragma unroll
                for (int r = NA0; r < NA; ++r) dma_a(buf ^ 1, r, cs + 1);
            }
            compute(buf, 1, H1{});
            WSTAMP(tm_comp)
            step_end();
            WSTAMP(tm_wait)
        }
    } else {
        // three weight stages, TWO steps of look-ahead: at step s the weights of step s + 2 go out; the next slice's whole patch goes
        // out at the slice's first step (its buffer was released by the barrier before).  The counted waits let exactly the requests
        // of the current step stay in flight: after an even step (4 weight rounds + NA patch pieces issued) everything older -- the
        // weights of step s + 1 -- has landed; after an odd step (4 issued) the weights of step s + 2 and the patch have.
#pragma unroll
        for (int r = 0; r < NA; ++r) dma_a(0, r, 0);
        dma_w(0, 0);
        dma_w(1, 1);
        step_end();
        WSTAMP(tm_pro)
        int st = 0;  // stage of the current step
        for (int cs = 0; cs + 1 < n_cs; ++cs) {
            const int buf = cs & 1;
            const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
            dma_w(st2, 2 * cs + 2);
#pragma unroll
            for (int r = 0; r < NA; ++r) dma_a(buf ^ 1, r, cs + 1);
            compute(buf, st, H0{});
            WSTAMP(tm_comp)
            wait_vm_lgkm0<4 + NA>();
            __builtin_amdgcn_s_barrier();
            WSTAMP(tm_wait)
            dma_w(st, 2 * cs + 3);  // (the stage this step's predecessor just finished with)
            compute(buf, st1, H1{});
            WSTAMP(tm_comp)
            wait_vm_lgkm0<4>();
            __builtin_amdgcn_s_barrier();
            WSTAMP(tm_wait)
            st = st2;
        }
        {
            const int buf = (n_cs - 1) & 1;
            const int st1 = st == 2 ? 0 : st + 1;
            compute(buf, st, H0{});
            WSTAMP(tm_comp)
            step_end();
            WSTAMP(tm_wait)
            compute(buf, st1, H1{});
            WSTAMP(tm_comp)
            step_end();
            WSTAMP(tm_wait)
        }
    }

    // ---- output transform (A^T = [1 1 1 0; 0 1 -1 -1]) ------------------------------------------------------------------------
    // rows of the group: Z_h[b] = sum_j M[h][j] A[j][b];   group 0 (i = 0, 1): Y[0][b] += Z0 + Z1, Y[1][b] += Z1
    //                                                       group 1 (i = 2, 3): Y[0][b] += Z0,      Y[1][b] += -Z0 - Z1
    f32x16 yp[2][2];
    {
        const f32x16 z00 = acc[0][0] + acc[0][1] + acc[0][2], z01 = acc[0][1] - acc[0][2] - acc[0][3];
        const f32x16 z10 = acc[1][0] + acc[1][1] + acc[1][2], z11 = acc[1][1] - acc[1][2] - acc[1][3];
        if (pg == 0) {
            yp[0][0] = z00 + z10, yp[0][1] = z01 + z11, yp[1][0] = z10, yp[1][1] = z11;
        } else {
            yp[0][0] = z00, yp[0][1] = z01, yp[1][0] = -z00 - z10, yp[1][1] = -z01 - z11;
        }
    }
    // The two position groups' partial sums go to TWO float32 tiles [BLOCK_PX][64] (group 0: smem + 0, group 1: smem + 64 KB), one
    // barrier, and the read-out adds them (the first version stored, synchronised, added in place, synchronised again).  The
    // residual and bias of all four chunks of a thread are requested BEFORE the tiles are written: one exposed round trip.
    // block pixel m = image m / (TH TW), (ty0 + (m % (TH TW)) / TW, tx0 + m % TW)
    constexpr int CHUNKS = BLOCK_PX * BN / 8, ITER = CHUNKS / NT;  // 2048 chunks of 8 columns, 4 per thread
    static_assert(CHUNKS % NT == 0 && NT % (BN / 8) == 0, "whole chunk rounds; a thread keeps its column chunk");
    const int cc = tid % (BN / 8);
    const int col0 = n0 + cc * 8;
    float4 b0 = float4{0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0;
    if (bias) {
        b0 = *reinterpret_cast<const float4*>(bias + col0);
        b1 = *reinterpret_cast<const float4*>(bias + col0 + 4);
    }
    int mpix[ITER];
    u32x4 rq[ITER][2];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = (tid + NT * it) / (BN / 8);
        const int g = row / (GEO::TH * GEO::TW), rg = row - g * (GEO::TH * GEO::TW);
        const int oy = ty0 + rg / GEO::TW, ox = tx0 + rg % GEO::TW;
        const bool live = oy < d.ho && ox < d.wo && img + g < d.n;
        mpix[it] = live ? ((img + g) * d.ho + oy) * d.wo + ox : -1;
        rq[it][0] = rq[it][1] = u32x4{0u, 0u, 0u, 0u};
        if (res && live) {
            const u32x4* rp = reinterpret_cast<const u32x4*>(res + (long)mpix[it] * d.cout + col0);
            rq[it][0] = rp[0];
            rq[it][1] = rp[1];
        }
    }
    float* tile = reinterpret_cast<float*>(smem) + pg * (BLOCK_PX * BN);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tt = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * hi;  // MFMA result row -> tile
        int m00;
        if constexpr (GEO::G == 1) {
            m00 = 2 * (tt >> 3) * GEO::TW + 2 * (tt & 7);
        } else {
            m00 = (tt >> 4) * (GEO::TH * GEO::TW) + 2 * ((tt >> 2) & 3) * GEO::TW + 2 * (tt & 3);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) tile[(m00 + a * GEO::TW + b) * BN + wn * 32 + (lane & 31)] = yp[a][b][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* t0 = reinterpret_cast<const float*>(smem);
    const float* t1 = t0 + BLOCK_PX * BN;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = (tid + NT * it) / (BN / 8);
        const float4 p0 = *reinterpret_cast<const float4*>(t0 + row * BN + cc * 8), p1 = *reinterpret_cast<const float4*>(t0 + row * BN + cc * 8 + 4);
        const float4 q0 = *reinterpret_cast<const float4*>(t1 + row * BN + cc * 8), q1 = *reinterpret_cast<const float4*>(t1 + row * BN + cc * 8 + 4);
        // (group 0's sum + group 1's sum) + bias: Y[0][b] = (Z0 + Z1) + Z2, Y[1][b] = Z1 + (-Z2 - Z3)
        float v[8] = {(p0.x + q0.x) + b0.x, (p0.y + q0.y) + b0.y, (p0.z + q0.z) + b0.z, (p0.w + q0.w) + b0.w,
                      (p1.x + q1.x) + b1.x, (p1.y + q1.y) + b1.y, (p1.z + q1.z) + b1.z, (p1.w + q1.w) + b1.w};
        if (mpix[it] >= 0) {
            float* yo = y + (long)mpix[it] * d.cout + col0;
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
        }
    }
#if TIA_WINO_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP(tm_epi)
    if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 64 || blockIdx.x == 1001))
        printf("wino wg %d (cin %d, stages %d, abl %d): prologue %lld  compute %lld  step waits %lld (of which vmcnt %lld)  epilogue %lld  (steps %d) | shader clock %.0f MHz\n",
               (int)blockIdx.x, d.cin, NSTAGE, d.abl, tm_pro, tm_comp, tm_wait, tm_vm, tm_epi, 2 * n_cs,
               100.0 * (double)(clock64() - t0c_) / (double)(wall_clock64() - t0w_));
#endif
}

// U = G g G^T (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]) in float64, rounded once; one thread per (cout, cin) pair
__global__ void wino_pack_kernel(const float* __restrict__ w_oihw, int cout, int cin, float* __restrict__ packed) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)cout * cin) return;
    const int o = (int)(idx / cin), c = (int)(idx - (long)o * cin);
    double g[3][3], t[4][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = (double)w_oihw[(idx * 3 + r) * 3 + s];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5 * (g[0][s] + g[1][s] + g[2][s]);
        t[2][s] = 0.5 * (g[0][s] - g[1][s] + g[2][s]);
        t[3][s] = g[2][s];
    }
    const int n_cs = cin >> 4, n_cb = cout >> 6;
    const int cs = c >> 4, hi = (c >> 3) & 1, kq = (c >> 2) & 1, c4 = c & 3, cb = o >> 6, col = o & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double uu[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long block = ((long)(i * 4 + j) * n_cs + cs) * n_cb + cb;
            packed[block * 1024 + ((hi * 2 + kq) * 64 + col) * 4 + c4] = (float)uu[j];
        }
    }
}

}  // namespace

namespace tia {

// dynamic LDS of the kernel: two patch buffers + two weight stages + the dump KB, at least the epilogue's two 64 KB tiles
static constexpr int wino_lds_bytes(int patch_units, int stages) {
    const int main_loop = 2 * ((patch_units + 63) / 64 * 64) * 16 + stages * 32768 + 1024;
    return main_loop > 2 * 256 * 64 * 4 ? main_loop : 2 * 256 * 64 * 4;
}

bool conv3x3_wino_serves(long nb, long h, long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo) {
    if (cin % 16 != 0 || cout % 64 != 0 || pad_top < 0 || pad_left < 0 || pad_top > 2 || pad_left > 2 || ho <= 0 || wo <= 0) return false;
    // the patch of a block must cover what its outputs read: any 3x3 / stride-1 geometry does (18 = 16 + 2)
    return nb > 0 && h > 0 && w > 0 && 16L * cin * cout * 4 <= 0x7fffffffL;
}

int conv3x3_wino_launch(const float* x, const float* u_packed, const float* bias, const float* residual, float* y, long nb, long h,
                        long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int relu, hipStream_t stream) {
    if (!conv3x3_wino_serves(nb, h, w, cin, cout, pad_top, pad_left, ho, wo)) return TIA_ESIZE;
    const bool small = ho <= 8 && wo <= 8;
    const long tiles_y = small ? 1 : (ho + 15) / 16, tiles_x = small ? 1 : (wo + 15) / 16;
    const long tiles = small ? (nb + 3) / 4 : nb * tiles_y * tiles_x;
    const WinoDims d{(int)nb, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)pad_top, (int)pad_left,
                     (unsigned)(nb * h * w * cin * 4), (unsigned)(16 * cin * cout * 4), (int)((cin / 16) * (cout / 64) * 4096),
                     getenv("TIA_WINO_ABL") ? atoi(getenv("TIA_WINO_ABL")) : 0};
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8), (unsigned)(cout / 64));
    // weight ring: three stages (two steps of look-ahead) where the LDS holds them -- the 16 x 16 geometry; the four-image geometry's
    // larger patch buffers leave room for two.  TIA_WINO_STAGES=2 forces two (developer switch, A/B measurements).
    static const bool two = getenv("TIA_WINO_STAGES") != nullptr && atoi(getenv("TIA_WINO_STAGES")) == 2;
    static tia::DeviceOnce attr16, attr16s2, attr8;  // the dynamic-LDS attribute is per device
#define TIA_WINO_LAUNCH(GEO_, NS_, ONCE_)                                                                                            \
    do {                                                                                                                             \
        constexpr int lds = wino_lds_bytes(GEO_::G * GEO_::IMG, NS_);                                                                \
        static_assert(lds <= 160 * 1024, "LDS");                                                                                     \
        if (!ONCE_.ensure([] {                                                                                                       \
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<GEO_, NS_>),                            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;                           \
            }))                                                                                                                      \
            return TIA_ELAUNCH;                                                                                                      \
        hipLaunchKernelGGL((conv3x3_wino_kernel<GEO_, NS_>), grid, dim3(512), lds, stream, x, u_packed, bias, residual, y, d, relu,   \
                           (int)tiles, (int)tiles_x, (int)(tiles_y * tiles_x));                                                      \
    } while (0)
    if (small)
        TIA_WINO_LAUNCH(W8, 2, attr8);
    else if (two)
        TIA_WINO_LAUNCH(W16, 2, attr16s2);
    else
        TIA_WINO_LAUNCH(W16, 3, attr16);
#undef TIA_WINO_LAUNCH
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia

extern "C" int tia_conv_pack_weights_wino_f32(const float* d_w_oihw, int64_t cout, int64_t cin, float* d_packed, void* stream) {
    if (!d_w_oihw || !d_packed || cout <= 0 || cin <= 0) return TIA_EINVAL;
    if (cin % 16 != 0 || cout % 64 != 0 || 16 * cin * cout * 4 > 0x7fffffffL) return TIA_ESIZE;
    const long total = (long)cout * cin;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w_oihw, (int)cout,
                       (int)cin, d_packed);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_conv3x3_wino_nhwc_f32(const float* d_x, const float* d_u_packed, const float* d_bias, const float* d_residual,
                                         float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t pad_top,
                                         int64_t pad_left, int64_t ho, int64_t wo, int32_t relu, void* stream) {
    if (!d_x || !d_u_packed || !d_y || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return TIA_EINVAL;
    if (ho <= 0 || wo <= 0 || pad_top < 0 || pad_left < 0 || pad_top > 2 || pad_left > 2) return TIA_EINVAL;
    if (ho - 1 - pad_top >= h || wo - 1 - pad_left >= w) return TIA_EINVAL;  // every output sees at least its first tap row / column start on the map
    if (((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_u_packed) | reinterpret_cast<uintptr_t>(d_y) |
          reinterpret_cast<uintptr_t>(d_residual) | reinterpret_cast<uintptr_t>(d_bias)) & 15) != 0)
        return TIA_EINVAL;
    if (cin % 16 != 0 || cout % 64 != 0) return TIA_ESIZE;
    // 32-bit byte offsets into the input: images go in groups of < 2 GiB (and < 2^31 / 4 output pixels)
    const long image_bytes = h * w * cin * 4;
    if (image_bytes > 0x7fffffffL || ho * wo > 0x7fffffffL / 4) return TIA_ESIZE;
    long group = 0x7fffffffL / image_bytes;
    if (group * ho * wo > 0x7fffffffL / 2) group = 0x7fffffffL / 2 / (ho * wo);
    if (group < 1) return TIA_ESIZE;
    if (ho <= 8 && wo <= 8 && group > 4) group -= group % 4;  // whole blocks of four images
    for (long first = 0; first < n; first += group) {
        const long nb = n - first < group ? n - first : group;
        const int rc = tia::conv3x3_wino_launch(d_x + first * h * w * cin, d_u_packed, d_bias, d_residual ? d_residual + first * ho * wo * cout : nullptr,
                                                d_y + first * ho * wo * cout, nb, h, w, cin, cout, pad_top, pad_left, ho, wo, relu,
                                                (hipStream_t)stream);
        if (rc != TIA_OK) return rc;
    }
    return TIA_OK;
}
