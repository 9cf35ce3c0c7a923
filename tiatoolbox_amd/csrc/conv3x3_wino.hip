// Winograd F(2x2, 3x3) form of the 3x3 / stride-1 NHWC convolution in float32 on the gfx950 matrix cores -- 16 multiplies per
// 2 x 2 outputs and (cin, cout) pair instead of 36, i.e. 2.25 x fewer MFMA operations than the direct tap-reuse kernel
// (conv3x3_spatial.hip), which sits at ~90 % of a float32 MFMA wall that equals the vector rate.  float32 Winograd is what the
// vendor libraries run for exactly these layers, so it is the reference's arithmetic CLASS (float32 in, float32 accumulate), not
// its operation order: results differ from a direct float32 convolution in the last bits (measured and bounded in
// tests/test_engine.py; the engines' default `conv_algo="auto"` takes this path on every layer it serves, `"direct"` never).
// Call sites in the reference: the 3x3 convolutions behind CNNModel.forward (models/architecture/vanilla.py:242-253,300-316).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A     per 2 x 2 output tile, 4 x 4 input tile d, 3 x 3 filter g   (Lavin & Gray)
//
// * weights: U = G g G^T computed ONCE at pack time in float64, rounded to float32 (tia_conv_pack_weights_wino_f32), stored in the
//   exact LDS image of a weight stage: [pos 16][cin/16][h8 2][cout/64] blocks of 2 KB = [hi 2][64 cout][4 channels]
//   (channel = 16 cs + 8 h8 + 4 hi + c4): a stage is a plain contiguous LDS-DMA copy, a lane's 4 k-values of a (position, channel
//   tile) are ONE conflict-free ds_read_b128.
// * a 512-thread workgroup owns 64 tiles (16 x 16 output pixels of one image, or four images of <= 8 x 8) x 64 output channels
//   x all 16 Winograd positions; 8 waves = 4 position ROWS i (positions (i, 0..3)) x 2 tile halves; a wave keeps 4 positions x
//   (32 tiles x 64 channels) = 128 accumulator registers.  One workgroup per CU (two waves per SIMD, <= 256 registers each).
// * the raw 18 x 18 (4 x 10 x 10) input patch of a 16-channel slice arrives by LDS-DMA (pairs of pixels at a pitch of 9 units:
//   conflict-free reads, see W16 / W8 below; double-buffered); the input transform V = B^T d B is done IN REGISTERS on the way to the
//   MFMA -- no transformed tensor ever exists in memory (the non-fused form moves 4 x the input and 4 x the output through HBM and
//   loses to the direct kernel).
// * a STEP covers 8 input channels (two per 16-channel slice): a lane reads two patch rows x four columns of its tile (8
//   ds_read_b128: its 4 channels), forms R = d[ra] +- d[rb] and V_j = R[c] +- R[c'] (16 packed adds), reads 8 weight units and
//   issues 32 MFMAs (4 positions x 2 channel tiles x 4 k-steps).  What bounds this kernel is the NUMBER of vector / LDS
//   instructions per MFMA: on a SIMD they are not hidden behind float32 MFMAs but ADD to them (measured: the whole kernel without a
//   single MFMA still takes 68 % of its time, profiles/r05j_wino_ablation.txt; ~13 cycles per instruction) -- 32 per 32 MFMAs in this
//   form, 56 in the first one (a wave = two position rows x 32 channels: every V was computed by two waves), 20 in the direct kernel.
// * the two halves of the position grid (rows {0, 1} / {2, 3}) run half a step apart (load phase | MFMA phase, one barrier per
//   half-step); per step a stage of 16 positions x 2 KB = 32 KB of weights streams in (two stages).
// * epilogue: column transform (over j) in registers, the row transform (over i = the four waves of a tile half) through two LDS
//   tiles [256 pixels][64 channels] (rows {0, 1} -> tile A, {2, 3} -> tile B; one store round, one add round), then + bias + residual,
//   ReLU, 16-byte stores as in the direct kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/tiatoolbox_amd.h"
#include "conv3x3_wino.hpp"
#include "dev_env.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int OOB = (int)0x80000000;

struct WinoDims {
    int n, h, w, cin, cout, ho, wo, pad_y, pad_x;
    unsigned x_bytes, u_bytes;
    int pos_stride;  // bytes between consecutive positions of the packed weights: (cin / 16) * (cout / 64) * 4096
    int abl;         // timing builds only (TIA_WINO_ABL): 1 no weight DMA in the loop, 2 no patch DMA, 4 weight DMA out of range (zeros), 8 half of it
    // window geometry (WR below): a block = `wg` windows of wty x wtx tiles, taken in order from the batch's window grid (wins_x x
    // wins_y windows per image); LDS: window pitch `wimg` units, row pitch `wrow` units; the reciprocals serve fdiv()
    int wg, wty, wtx, wrow, wimg, wins_x, wins_y, n_windows;
    float inv_wimg, inv_wrow, inv_wt, inv_wtx, inv_wins, inv_wins_x;
};

// n / d for 0 <= n < 2^24, 0 < d < 2^24 with the reciprocal computed on the host: a float estimate and one correction step each way
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

// Packed float32 add / subtract (two channels per instruction).  Inline assembly: the compiler splits a v2f32 subtraction into two
// scalar v_sub_f32 (48 of the 56 vector instructions of a load phase), and it is the NUMBER of vector instructions issued beside the
// SIMD partner's MFMA stream that stretches that phase.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// LDS patch layout, in 16-byte units: a pixel is 4 units (16 float32 channels); pixels are stored in PAIRS of 9 units (two pixels +
// one padding unit): pixel px of a row at (px >> 1) * 9 + (px & 1) * 4.  A Winograd tile's pixels are two apart, so with the direct
// kernel's pitch of 5 units per pixel every lane of a ds_read_b128 lands on an even unit (the first version: an inherent 2-way bank
// conflict on all 16 patch reads of a step, SQ_LDS_BANK_CONFLICT = 38 % of the LDS cycles); with 9 units per pair the tiles
// tx = 0..3 of a row sit at units {0, 9, 2, 11} and tx = 4..7 at {4, 13, 6, 15} (mod 16), and the four translates of {0, 2, 9, 11}
// by 0, 4, 8, 12 tile Z_16 exactly -- so the row / image pitches below make the 16 lanes of every service group of a ds_read_b128
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32) hit 16 different 16-byte bank groups, for every row, column and
// channel-half offset (those shift all lanes alike).
//   W16: one image, 16 x 16 output pixels = 8 x 8 tiles, patch 18 x 18: 9 pairs = 81 units per row, ROW = 84 (2 ROW = 8 mod 16:
//        tile rows ty = 0..3 of a group shift by {0, 8, 0, 8} -> translates {0, 12, 4, 8} and {4, 8, 0, 12})
//   W8:  FOUR images of at most 8 x 8 = 4 x (4 x 4) tiles, patch 10 x 10 each: 5 pairs = 45 units per row, ROW = 50 (2 ROW = 4 mod
//        16), image pitch 512 (= 0 mod 16): translates {0, 12, 4, 8} and {4, 8, 0, 12} over (image, tile row)
struct W16 {
    static constexpr int G = 1, TH = 16, TW = 16, PH = 18, PWD = 18, ROW = 84, IMG = 18 * 84;
};
struct W8 {
    static constexpr int G = 4, TH = 8, TW = 8, PH = 10, PWD = 10, ROW = 50, IMG = 512;
};
// WR: maps that 16 x 16 blocks cover badly (56 / 28 / 14 of 224^2 patches: 76.6 %): a block is `wg` WINDOWS of wty x wtx tiles
// (wg * wty * wtx <= 64) taken consecutively from the window grid of the whole batch -- e.g. two windows of 4 x 7 tiles on a 56^2 map,
// four of 2 x 7 on 28^2, eight of 1 x 7 on 14^2: 87.5 % of the MFMA rows busy, blocks run across image boundaries.  Run-time
// geometry (WinoDims), host-computed reciprocals; each window keeps its own (2 wty + 2) x (2 wtx + 2) pixel patch (pair layout, no bank
// analysis: a few 2-way conflicts).  Block pixel m = 4 * tile + 2 a + b.
struct WR {
    static constexpr int G = 0, TH = 16, TW = 16, PH = 0, PWD = 0, ROW = 0, IMG = 2560;  // IMG: the LDS patch allocation in units
};
// unit offset of pixel column px inside a row
__device__ __forceinline__ constexpr int px_unit(int px) { return (px >> 1) * 9 + (px & 1) * 4; }

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
// Epilogue ablations (developer builds only: -DTIA_WINO_ABLATE=1, then TIA_WINO_ABL bits 128: nothing after the column transform,
// 256: no exchange, 512: no read-out) -- the persistent form has no registers left for the timing build's counters.
#ifndef TIA_WINO_ABLATE
#define TIA_WINO_ABLATE 0
#endif
#if TIA_WINO_TIMING
#define WSTAMP(var) { const long long now_ = clock64(); var += now_ - tl_; tl_ = now_; }
#define WSTAMP_ROLE(mfma_if_pg0) { const long long now_ = clock64(); if ((pg == 0) == (mfma_if_pg0)) tm_comp += now_ - tl_; else tm_load += now_ - tl_; tl_ = now_; }
#else
#define WSTAMP(var)
#define WSTAMP_ROLE(x)
#endif

// PERSIST: one workgroup per CU walks a list of (pixel block, 64-channel tile) items.  The flattened step sequence simply runs on
// across items: the last two steps of an item request the NEXT item's first patch slice and first two weight stages instead of its
// own, so the ~8 k cycles a fresh workgroup spends waiting for its first HBM round trip (and the dispatch of a workgroup per item)
// disappear behind matrix work; the epilogue's 64 KB (exchange area, then tile) lie over the patch buffer the next item does not need yet.
template <typename GEO, int NSTAGE, bool PERSIST>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                              const float* __restrict__ bias, const float* __restrict__ res,
                                                              float* __restrict__ y, WinoDims d, int relu, int m_tiles, int tiles_x,
                                                              int tiles_per_image) {
    constexpr bool RT = GEO::G == 0;                              // run-time window geometry
    constexpr int NT = 512, BN = 64;
    const int ROW = RT ? d.wrow : GEO::ROW;                       // (a compile-time constant for the fixed geometries)
    constexpr int BLOCK_PX = 256;                                 // 256 output pixels = 64 tiles
    constexpr int A_UNITS = ((RT ? 1 : GEO::G) * GEO::IMG + 63) / 64 * 64;   // patch units (16 bytes), whole waves: 1536 | 2048 | 2560
    constexpr int NA = (A_UNITS + NT - 1) / NT;                   // DMA pieces per patch: 3 | 4
    constexpr int A_BYTES = A_UNITS * 16;
    constexpr int W_STAGE = 16 * 2048;                            // 16 positions x [8 channels][64 columns] float32
    // LDS map.  one block per workgroup: [patch 0][patch 1][stage 0][stage 1][dump], the epilogue's 64 KB alias the front of it.
    // PERSIST: [patch 0][stage 0][stage 1][patch 1 ...] with the epilogue's 64 KB starting at patch 1 (patch 0 and the stages
    // hold the next item's first operands while the epilogue runs), the dump KB behind the tile.
    constexpr int EPI_TILE = BLOCK_PX * BN * 4;
    // LATE (the larger patches of the 8 x 8 and window geometries): [patch 0][stage 0][patch 1][stage 1] -- the epilogue's 64 KB also
    // cover the front of stage 1, whose first refill for the next item is requested BEHIND the epilogue (one more barrier per item, and
    // part of that round trip exposed) instead of in the item's last step.
    constexpr bool LATE = PERSIST && A_BYTES + NSTAGE * W_STAGE + (A_BYTES > EPI_TILE ? A_BYTES : EPI_TILE) + 1024 > 160 * 1024;
    constexpr int OFF_W = PERSIST ? A_BYTES : 2 * A_BYTES;
    constexpr int OFF_A1 = PERSIST ? (LATE ? A_BYTES + W_STAGE : A_BYTES + NSTAGE * W_STAGE) : A_BYTES;
    constexpr int OFF_W1 = LATE ? OFF_A1 + A_BYTES : OFF_W + W_STAGE;
    constexpr int OFF_EPI = PERSIST ? OFF_A1 : 0;
    constexpr int EPI_BYTES = EPI_TILE;                            // the epilogue's exchange area, then its float32 tile
    constexpr int MAIN_END = PERSIST ? (LATE ? OFF_W1 + W_STAGE : OFF_A1 + A_BYTES) : 2 * A_BYTES + NSTAGE * W_STAGE;
    constexpr int DUMP = PERSIST ? (MAIN_END > OFF_EPI + EPI_BYTES ? MAIN_END : OFF_EPI + EPI_BYTES) : MAIN_END;  // 1 KB the idle waves of the last patch piece write their zeros to
    constexpr int LDS_BYTES = DUMP + 1024 > OFF_EPI + EPI_BYTES ? DUMP + 1024 : OFF_EPI + EPI_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(NA >= 2 && NA <= 6, "patch pieces are spread over the two steps of a slice");
    static_assert(NSTAGE == 2, "weight ring: two stages (a third one was measured and lost, profiles/r05d_perf_wino256_3stage.txt)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if TIA_WINO_TIMING
    long long tm_pro = 0, tm_comp = 0, tm_load = 0, tm_wait = 0, tm_epi = 0, tm_e0 = 0, tm_e1 = 0, tl_ = clock64();
    int n_items_ = 0;
    const long long t0c_ = tl_, t0w_ = wall_clock64();
#endif

    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int n_cs = d.cin >> 4, n_cb = d.cout >> 6;
    // an ITEM = (pixel block mt_id, channel tile cb).  One block per workgroup: the grid is the item list.  PERSIST: workgroup q of XCD x
    // (consecutive workgroup ids go round the XCDs) takes items q, q + Q, .. of the XCD's contiguous range of pixel blocks, channel tile
    // fastest -- the workgroups of an XCD that run at the same time share patches (one HBM read, then L2 hits).
    int item = 0, item_end = 1, item_step = 1, mt_lo = 0;
    if constexpr (PERSIST) {
        mt_lo = (bid & 7) * per_xcd;
        const int mt_hi = mt_lo + per_xcd < m_tiles ? mt_lo + per_xcd : m_tiles;
        item = bid >> 3, item_step = (int)(gridDim.x >> 3), item_end = (mt_hi - mt_lo) * n_cb;
        if (item >= item_end) return;
    } else {
        if ((bid % 8) * per_xcd + bid / 8 >= m_tiles) return;  // every XCD walks a contiguous range of pixel blocks
    }
    int mt_id, cb, img, ty0, tx0;
    auto decode = [&](int it) {  // (wave-uniform: scalar registers)
        if constexpr (PERSIST) {
            const int q = it / n_cb;
            mt_id = mt_lo + q, cb = it - q * n_cb;
        } else {
            mt_id = (bid % 8) * per_xcd + bid / 8, cb = (int)blockIdx.y;
        }
        img = RT ? 0 : (GEO::G == 1 ? mt_id / tiles_per_image : mt_id * GEO::G);  // (WR: windows carry their own image)
        const int trem = GEO::G == 1 ? mt_id - img * tiles_per_image : 0;
        ty0 = (trem / tiles_x) * GEO::TH;
        tx0 = (trem - (trem / tiles_x) * tiles_x) * GEO::TW;
    };
    decode(item);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // (kept in a vector register: with a scalar wave index the role
    // branches below become scalar branches and the register allocator spills 106 registers over them; 238 without)
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);  // the same number in a scalar register: LDS-DMA destinations (M0) and offsets
    const int irow = wave >> 1, wm = wave & 1, pg = wave >> 2;  // position row i, tile half; group = which half of the position grid (ping-pong)
    const int hi = lane >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u), 0, (int)d.u_bytes, 0x00020000);

    // patch staging: unit U = NT r + tid -> image U / IMG, row (U % IMG) / ROW, pixel pair (.. % ROW) / 9, pixel and unit-of-slice from
    // the rest (layout above); outside the image / patch, padding units: an out-of-range offset (the DMA writes zeros)
    int cen[NA];
    auto make_cen = [&] {  // for the item `decode` has just set
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int un = NT * r + tid;
        if constexpr (RT) {
            const int g = fdiv(un, d.wimg, d.inv_wimg), ug = un - g * d.wimg;
            const int py = fdiv(ug, d.wrow, d.inv_wrow), rem = ug - py * d.wrow;
            const int pair = rem / 9, r9 = rem - pair * 9;
            const int px = 2 * pair + (r9 >> 2), chunk = r9 == 8 ? 4 : (r9 & 3);
            const int win = mt_id * d.wg + g;                                    // window of the batch
            const int wi = fdiv(win, d.wins_x * d.wins_y, d.inv_wins), wr = win - wi * d.wins_x * d.wins_y;  // image, window in it
            const int wy = fdiv(wr, d.wins_x, d.inv_wins_x), wx = wr - wy * d.wins_x;
            const int iy = wy * 2 * d.wty - d.pad_y + py, ix = wx * 2 * d.wtx - d.pad_x + px;
            const bool inside = g < d.wg && win < d.n_windows && py < 2 * d.wty + 2 && px < 2 * d.wtx + 2 && chunk < 4 &&
                                (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
            cen[r] = inside ? ((wi * d.h + iy) * d.w + ix) * d.cin * 4 + 16 * chunk : OOB;
        } else {
            const int g = un / GEO::IMG, ug = un - g * GEO::IMG;
            const int py = ug / ROW, rem = ug - py * ROW;
            const int pair = rem / 9, r9 = rem - pair * 9;
            const int px = 2 * pair + (r9 >> 2), chunk = r9 == 8 ? 4 : (r9 & 3);  // (unit 8 of a pair: padding)
            const int iy = ty0 - d.pad_y + py, ix = tx0 - d.pad_x + px;
            const bool inside = g < GEO::G && img + g < d.n && py < GEO::PH && px < GEO::PWD && chunk < 4 && (unsigned)iy < (unsigned)d.h &&
                                (unsigned)ix < (unsigned)d.w;
            cen[r] = inside ? (((img + g) * d.h + iy) * d.w + ix) * d.cin * 4 + 16 * chunk : OOB;
        }
    }
    };
    make_cen();
    // weight staging: a stage = 16 position blocks of 2 KB; DMA round q (0..3) moves positions 4 q + (wave >> 1): per lane the offset
    // inside the block + (wave >> 1) positions; the rest is scalar
    const int w_voff = (wave_s & 1) * 1024 + lane * 16 + (wave_s >> 1) * d.pos_stride;

    unsigned char* const abuf0 = smem;
    constexpr int A_PITCH = OFF_A1;  // bytes between the two patch buffers
    auto wst = [&](int stage) -> unsigned char* { return smem + (stage ? OFF_W1 : OFF_W); };  // (stage: 0 | 1)
    auto dma_a = [&](int buf, int r, int cs) {
#if TIA_WINO_TIMING || TIA_WINO_ABLATE
        if ((d.abl & 2) && cs > 0) return;
#endif
        if constexpr (RT) {
            if (NT * r >= d.wg * d.wimg) return;  // (scalar) pieces past the block's windows: nothing of them is ever read
        }
        unsigned char* dst = (NT * r + wave_s * 64 >= A_UNITS) ? smem + DUMP : abuf0 + buf * A_PITCH + r * (NT * 16) + wave_s * 1024;
        dma16(rx, dst, cen[r], cs * 64);
    };
    // weights of flattened step s = 2 cs + h8 (h8: which 8 channels of the 16-channel slice)
    auto dma_w = [&](int stage, int s, int cbi) {
#if TIA_WINO_TIMING || TIA_WINO_ABLATE
        if ((d.abl & 1) && s > 0) return;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(ru, wst(stage) + q * 8192 + wave_s * 1024, w_voff, 4 * q * d.pos_stride + (s * n_cb + cbi) * 2048);
    };

    f32x16 acc[4][2];  // [position j of the wave's row][channel tile]
    // (a macro, not a lambda: with `acc` captured by a lambda the register allocator ends up 50 registers higher and spills)
#define TIA_WINO_CLEAR_ACC()                                              \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                      \
        _Pragma("unroll") for (int ct_ = 0; ct_ < 2; ++ct_)               \
            _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) acc[j_][ct_][e_] = 0.0f

    // the lane's tile: MFMA row = lane & 31 -> tile 32 wm + (lane & 31); its 4 x 4 input tile starts at patch pixel (2 ty, 2 tx); the
    // lane's k values of a step are channels 8 h8 + 4 hi + 0..3 = unit 2 h8 + hi of the pixel
    const int t = 32 * wm + (lane & 31);
    int fa;
    if constexpr (RT) {
        const int wt = d.wty * d.wtx;
        const int tv = t < d.wg * wt ? t : 0;  // idle MFMA rows read tile 0 (their results are dropped)
        const int g = fdiv(tv, wt, d.inv_wt), rr = tv - g * wt;
        const int tyy = fdiv(rr, d.wtx, d.inv_wtx), txx = rr - tyy * d.wtx;
        fa = g * d.wimg + 2 * tyy * d.wrow + txx * 9 + hi;
    } else if constexpr (GEO::G == 1) {
        fa = 2 * (t >> 3) * ROW + (t & 7) * 9 + hi;
    } else {
        fa = (t >> 4) * GEO::IMG + 2 * ((t >> 2) & 3) * ROW + (t & 3) * 9 + hi;
    }
    // weights of the lane: position block (i, j) of the stage, units [hi][column]: one 16-byte read per (position, channel tile)
    const int fb = irow * 4 * 128 + hi * 64 + (lane & 31);

    // A wave's step k (8 input channels) is software-pipelined against its own MFMAs:
    //   top of step k (behind the barrier: every DMA issued a step ago has landed, every LDS read of the last step has returned):
    //     DMA requests (weights of step k + 2 into the stage step k just released; at even k the patch of the next slice),
    //     the 8 patch reads of step k + 1 (two rows x four columns of the lane's tile);
    //   for each position j: its 8 MFMAs of step k, then the 2 weight reads of step k + 1 into the registers they just freed;
    //   the input transform of step k + 1 (R = d[ra] +- d[rb] per column, V_j = R0 - R2 | R1 + R2 | R2 - R1 | R1 - R3: 16 packed adds);
    //   s_waitcnt + ONE barrier.
    // History (profiles/r05*_wino_*): reading a position's operands right in front of its MFMAs, then all reads up front, then the
    // two halves of the position grid half a step apart ("ping-pong": load phase | MFMA phase, a barrier per half-step) all ended at
    // ~5,700-6,100 cycles per 4,096 cycles of matrix work: beside a float32 MFMA stream the SIMD partner's vector / LDS instructions
    // take 2-3 x as long, so phases that were meant to overlap mostly added up.  Here a wave's LDS round trips hide behind its OWN
    // MFMAs and only ~30 vector / LDS issue slots per 32 MFMAs are exposed.
    f32x2 vreg[4][2];  // V_j of the step whose MFMAs are next, four channels as two pairs
    u32x4 wq[4][2];    // [j][channel tile]: four k values each
    u32x4 pa[4], pb[4];  // raw patch units of the NEXT step (rows ra / rb, columns 0..3)
    const int RA0 = 0 * ROW, RB0 = 2 * ROW, RA1 = 1 * ROW, RB1 = 2 * ROW, RA2 = 2 * ROW, RB2 = 1 * ROW, RA3 = 1 * ROW, RB3 = 3 * ROW;
    // R = d[ra] +- d[rb]: i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3 (per-wave constants: the row offsets go into one register each)
    const int ra_off = irow == 0 ? RA0 : (irow == 1 ? RA1 : (irow == 2 ? RA2 : RA3));
    const int rb_off = irow == 0 ? RB0 : (irow == 1 ? RB1 : (irow == 2 ? RB2 : RB3));
    const bool plus = irow == 1;
    const int fa_a = fa + ra_off, fa_b = fa + rb_off;
    auto patch_reads = [&](int s) {  // step s: slice s >> 1 (buffer (s >> 1) & 1), channels 8 (s & 1) ..
        const u32x4* sa = reinterpret_cast<const u32x4*>(abuf0 + ((s >> 1) & 1) * A_PITCH) + 2 * (s & 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) pa[c] = sa[fa_a + px_unit(c)], pb[c] = sa[fa_b + px_unit(c)];
    };
    auto weight_reads = [&](int s, int j) {
        const u32x4* sb = reinterpret_cast<const u32x4*>(wst(s & 1)) + fb;
        wq[j][0] = sb[j * 128], wq[j][1] = sb[j * 128 + 32];
    };
    auto transform = [&] {
        auto pair_of = [](const u32x4& q, int k) { return f32x2{__uint_as_float(q[2 * k]), __uint_as_float(q[2 * k + 1])}; };
        f32x2 R[4][2];
        if (plus) {  // (wave-uniform)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 2; ++k) R[c][k] = pk_add(pair_of(pa[c], k), pair_of(pb[c], k));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 2; ++k) R[c][k] = pk_sub(pair_of(pa[c], k), pair_of(pb[c], k));
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            vreg[0][k] = pk_sub(R[0][k], R[2][k]);
            vreg[1][k] = pk_add(R[1][k], R[2][k]);
            vreg[2][k] = pk_sub(R[2][k], R[1][k]);
            vreg[3][k] = pk_sub(R[1][k], R[3][k]);
        }
    };
    auto mfma_j = [&](int j) {
#if TIA_WINO_TIMING || TIA_WINO_ABLATE
        if (d.abl & 64) return;  // no MFMAs
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[j][k >> 1][k & 1], __uint_as_float(wq[j][ct][k]), acc[j][ct], 0, 0, 0);
    };
#if TIA_WINO_TIMING
    long long tm_vm = 0;
#endif
    auto step_end = [&] {
        // (the builtin, not inline assembly: the compiler's own wait-count bookkeeping sees it -- behind an opaque asm it assumes the LDS-DMA
        // requests are still in flight and puts a vmcnt(0) in front of the epilogue's LDS accesses, i.e. waits for the residual loads)
        wait_vm_lgkm0<0>();
#if TIA_WINO_TIMING
        { const long long now_ = clock64(); tm_vm += now_ - tl_; }
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // prologue: patch of slice 0, weights of steps 0 and 1; the operands of step 0
    const int n_steps = 2 * n_cs;
#pragma unroll
    for (int r = 0; r < NA; ++r) dma_a(0, r, 0);
    dma_w(0, 0, cb);
    dma_w(1, 1, cb);
    step_end();
    TIA_WINO_CLEAR_ACC();
    for (bool first_item = true;; first_item = false) {  // (one round unless PERSIST)
    patch_reads(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) weight_reads(0, j);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    transform();
    // (every wave has read stage 0's weights: step 0 may refill it; PERSIST, later items: every wave has left the epilogue's tile, which
    // step 0's patch DMA overwrites -- the epilogue's global stores need not have landed for that)
    if (first_item || LATE) step_end(); else lds_barrier();  // (LATE: stage 1's refill, requested behind the epilogue, has to have landed)
    WSTAMP(tm_pro)
    // PERSIST: the item after this one is decoded in step n_steps - 3 (behind that step's MFMAs: every patch request of the current item
    // has been issued by then, so its offsets `cen` can be overwritten) and requested by the last slice
    const int cur_mt = mt_id, cur_cb = cb, cur_img = img, cur_ty0 = ty0, cur_tx0 = tx0;
    const bool has_next = PERSIST && item + item_step < item_end;
    // The two waves of a SIMD (w and w + 4: position rows {0, 1} and {2, 3}) take the step's non-MFMA work at DIFFERENT times: rows
    // {2, 3} issue the MFMAs of j = 0, 1 first and request DMAs / patch reads after them, rows {0, 1} the other way round -- while one
    // wave of the SIMD issues requests the other one feeds the matrix pipe (after a barrier both used to start with ~40 scalar /
    // vector instructions of request work, the pipe idle).
    for (int k = 0; k < n_steps; ++k) {
        const bool next = k + 1 < n_steps;
        auto requests = [&] {
            if (k + 2 < n_steps) {
                dma_w(k & 1, k + 2, cur_cb);
                if ((k & 1) == 0) {  // first step of slice k / 2: the next slice's patch into the other buffer
#pragma unroll
                    for (int r = 0; r < NA; ++r) dma_a(((k >> 1) & 1) ^ 1, r, (k >> 1) + 1);
                }
            } else if (PERSIST && has_next) {  // the last slice (n_cs is even: it sits in buffer 1): the next item's first operands
                if (!LATE || (k & 1) == 0) dma_w(k & 1, k + 2 - n_steps, cb);  // (LATE: stage 1 lies under the epilogue -- requested behind it)
                if ((k & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < NA; ++r) dma_a(0, r, 0);
                }
            }
            if (next) patch_reads(k + 1);
        };
        auto mma = [&](int j) {
            mfma_j(j);
            if (next) weight_reads(k + 1, j);
        };
        if (pg == 1) {
            mma(0), mma(1);
            requests();
        } else {
            requests();
            mma(0), mma(1);
        }
        mma(2), mma(3);
        if constexpr (PERSIST) {
            if (has_next && k == n_steps - 3) {
                item += item_step;
                decode(item);
                make_cen();
            }
        }
        WSTAMP(tm_comp)
        if (next) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            transform();
        }
        WSTAMP(tm_load)
        step_end();
        WSTAMP(tm_wait)
    }

    // ---- output transform (A^T = [1 1 1 0; 0 1 -1 -1]) ------------------------------------------------------------------------
    // column transform in registers (per channel tile): Z[b] = sum_j M[i][j] A[j][b] = M0 + M1 + M2 | M1 - M2 - M3; the row transform
    // Y[0][b] = Z(0) + Z(1) + Z(2), Y[1][b] = Z(1) - Z(2) - Z(3) runs over the four waves of a tile half, through LDS (below)
    f32x16 z[2][2];  // [b][channel tile]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        z[0][ct] = acc[0][ct] + acc[1][ct] + acc[2][ct];
        z[1][ct] = acc[1][ct] - acc[2][ct] - acc[3][ct];
    }
    // The four waves of a tile half hold Z(i) in the SAME lane layout, so the middle rows travel lane to lane: waves i = 1, 2 park
    // Z(1), Z(2) in a 64 KB exchange area (16 conflict-free ds_write_b128 per lane), wave i = 0 forms Y[0][b] = (Z0 + Z1) + Z2 and wave
    // i = 3 Y[1][b] = Z1 + (-Z3 - Z2) in registers (16 + 16 ds_read_b128), and only those FINAL rows go through the float32 tile
    // [BLOCK_PX][64] (it takes the exchange area's place) for the 16-byte read-out.  (The first form moved every Z through two such
    // tiles with 4-byte stores and read-modify-writes: 1,024 LDS instructions per workgroup in its two rounds against 448 here, and
    // twice the tile bytes to read out.)  (y0 / y1 = the tile's output rows 2 ty / 2 ty + 1, columns 2 tx + b.)  The residual and
    // bias of all four chunks of a thread are requested BEFORE the exchange: one exposed round trip.
    if (TIA_WINO_ABLATE && (d.abl & 128)) {
        if (d.n < 0) y[tid] = z[0][0][0] + z[1][1][5] + z[0][1][3] + z[1][0][7];  // (never: keeps the accumulators alive)
    } else {
    // block pixel m = image m / (TH TW), (ty0 + (m % (TH TW)) / TW, tx0 + m % TW)
    constexpr int CHUNKS = BLOCK_PX * BN / 8, ITER = CHUNKS / NT;  // 2048 chunks of 8 columns, 4 per thread
    static_assert(CHUNKS % NT == 0 && NT % (BN / 8) == 0, "whole chunk rounds; a thread keeps its column chunk");
    const int cc = tid % (BN / 8);
    const int col0 = cur_cb * BN + cc * 8;
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
        if constexpr (RT) {  // block pixel = 4 * tile + 2 a + b
            const int tt = row >> 2, wt = d.wty * d.wtx;
            const int g = fdiv(tt, wt, d.inv_wt), rr = tt - g * wt;
            const int tyy = fdiv(rr, d.wtx, d.inv_wtx), txx = rr - tyy * d.wtx;
            const int win = cur_mt * d.wg + g;
            const int wi = fdiv(win, d.wins_x * d.wins_y, d.inv_wins), wr = win - wi * d.wins_x * d.wins_y;
            const int wy = fdiv(wr, d.wins_x, d.inv_wins_x), wx = wr - wy * d.wins_x;
            const int oy = 2 * (wy * d.wty + tyy) + ((row >> 1) & 1), ox = 2 * (wx * d.wtx + txx) + (row & 1);
            const bool live = tt < d.wg * wt && win < d.n_windows && oy < d.ho && ox < d.wo;
            mpix[it] = live ? (wi * d.ho + oy) * d.wo + ox : -1;
        } else {
            const int g = row / (GEO::TH * GEO::TW), rg = row - g * (GEO::TH * GEO::TW);
            const int oy = cur_ty0 + rg / GEO::TW, ox = cur_tx0 + rg % GEO::TW;
            const bool live = oy < d.ho && ox < d.wo && cur_img + g < d.n;
            mpix[it] = live ? ((cur_img + g) * d.ho + oy) * d.wo + ox : -1;
        }
        const bool live = mpix[it] >= 0;
        rq[it][0] = rq[it][1] = u32x4{0u, 0u, 0u, 0u};
        if (res && live) {
            const u32x4* rp = reinterpret_cast<const u32x4*>(res + (long)mpix[it] * d.cout + col0);
            rq[it][0] = rp[0];
            rq[it][1] = rp[1];
        }
    }
    WSTAMP(tm_e0)
    float* tile = reinterpret_cast<float*>(smem + OFF_EPI);
    if (!(TIA_WINO_ABLATE && (d.abl & 256))) {   // exchange area: [Z(1) | Z(2)][tile half][q 16][lane 64] float4; unit q of a lane = z[q >> 3][(q >> 2) & 1][4 (q & 3) ..]
        unsigned xoff = OFF_EPI + (wm * (16 * 64) + lane) * 16;
        asm volatile("" : "+v"(xoff));  // ONE base register + immediate offsets (left alone, the compiler hoists 16 addresses out of the
                                        // item loop and spills them)
        float4* const xch = reinterpret_cast<float4*>(smem + xoff);
        constexpr int SLOT = 2 * 16 * 64;
        if (irow == 1 || irow == 2) {
            float4* dst = xch + (irow - 1) * SLOT;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x16& zz = z[q >> 3][(q >> 2) & 1];
                dst[q * 64] = float4{zz[4 * (q & 3)], zz[4 * (q & 3) + 1], zz[4 * (q & 3) + 2], zz[4 * (q & 3) + 3]};
            }
        }
        lds_barrier();
        // Y[0] = (Z0 + Z1) + Z2 by wave i = 0, Y[1] = Z1 + (-Z3 - Z2) by wave i = 3 (two exec-masked blocks: one block with a per-lane
        // select computes both forms for every element)
        auto combine = [&](bool top) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 z1 = xch[q * 64], z2 = xch[SLOT + q * 64];
                f32x16& zz = z[q >> 3][(q >> 2) & 1];
                const float a1[4] = {z1.x, z1.y, z1.z, z1.w}, a2[4] = {z2.x, z2.y, z2.z, z2.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = zz[4 * (q & 3) + k];
                    zz[4 * (q & 3) + k] = top ? (v + a1[k]) + a2[k] : a1[k] + (-v - a2[k]);
                }
            }
        };
        if (irow == 0) combine(true);
        if (irow == 3) combine(false);
        lds_barrier();  // every exchange read has returned: the tile may take the area's place
    }
    // output row `a` of this wave's tiles -> the tile
    auto to_tile = [&](int a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tt = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * hi;  // MFMA result row -> tile
            int m00;
            constexpr int ASTEP = RT ? 2 : GEO::TW;  // block pixels between the tile's two output rows
            if constexpr (RT) {
                m00 = 4 * tt;
            } else if constexpr (GEO::G == 1) {
                m00 = 2 * (tt >> 3) * GEO::TW + 2 * (tt & 7);
            } else {
                m00 = (tt >> 4) * (GEO::TH * GEO::TW) + 2 * ((tt >> 2) & 3) * GEO::TW + 2 * (tt & 3);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) tile[(m00 + a * ASTEP + b) * BN + ct * 32 + (lane & 31)] = z[b][ct][e];
        }
    };
    if (irow == 0) to_tile(0);
    if (irow == 3) to_tile(1);
    lds_barrier();
    WSTAMP(tm_e1)
    const float* t0 = reinterpret_cast<const float*>(smem + OFF_EPI);
    if (!(TIA_WINO_ABLATE && (d.abl & 512)))
    {
        // Every chunk's value is finished (residual consumed) BEFORE the first store: loads and stores share vmcnt on gfx9 and may retire
        // out of order against each other, so a load result used behind a store costs an s_waitcnt vmcnt(0) -- the store's round trip,
        // once per chunk in the first form of this loop.
        float4 o[ITER][2];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = (tid + NT * it) / (BN / 8);
            const float4 p0 = *reinterpret_cast<const float4*>(t0 + row * BN + cc * 8), p1 = *reinterpret_cast<const float4*>(t0 + row * BN + cc * 8 + 4);
            float v[8] = {p0.x + b0.x, p0.y + b0.y, p0.z + b0.z, p0.w + b0.w, p1.x + b1.x, p1.y + b1.y, p1.z + b1.z, p1.w + b1.w};
            if (res) {  // (dead pixels: zeros, never stored)
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
            o[it][0] = float4{v[0], v[1], v[2], v[3]};
            o[it][1] = float4{v[4], v[5], v[6], v[7]};
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            if (mpix[it] >= 0) {
                float4* yo = reinterpret_cast<float4*>(y + (long)mpix[it] * d.cout + col0);
                yo[0] = o[it][0];
                yo[1] = o[it][1];
            }
        }
    }
    }
#if TIA_WINO_TIMING
    ++n_items_;
#endif
    if (!has_next) break;
    if constexpr (LATE) {  // every wave has read the tile: stage 1 may take the next item's second step
        lds_barrier();
        dma_w(1, 1, cb);
    }
    TIA_WINO_CLEAR_ACC();
    WSTAMP(tm_epi)
    }  // items
#undef TIA_WINO_CLEAR_ACC
#if TIA_WINO_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP(tm_epi)
    if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.y == 0 && blockIdx.x == 64)
        printf("wino wg %d wave %d (cin %d, abl %d, %d items): prologue %lld  MFMA phases %lld  load phases %lld  waits %lld (of which before the barrier %lld)  epilogue: requests + column transform %lld, rounds %lld, read-out %lld  (steps %d) | shader clock %.0f MHz\n",
               (int)blockIdx.x, (int)(threadIdx.x >> 6), d.cin, d.abl, n_items_, tm_pro, tm_comp, tm_load, tm_wait, tm_vm, tm_e0, tm_e1, tm_epi, 2 * n_cs,
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
    const int cs = c >> 4, h8 = (c >> 3) & 1, hi = (c >> 2) & 1, c4 = c & 3, cb = o >> 6, col = o & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double uu[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long block = (((long)(i * 4 + j) * n_cs + cs) * 2 + h8) * n_cb + cb;  // 2 KB = 512 floats
            packed[block * 512 + (hi * 64 + col) * 4 + c4] = (float)uu[j];
        }
    }
}

}  // namespace

namespace tia {

// dynamic LDS of the kernel: two patch buffers + two weight stages + the dump KB, at least the epilogue's 64 KB; the
// persistent form keeps the epilogue's 64 KB at the second patch buffer, behind the first buffer and the stages (the kernel's map)
static constexpr int wino_lds_bytes(int patch_units, int stages, bool persist) {
    const int a_bytes = ((patch_units + 63) / 64 * 64) * 16, tile = 256 * 64 * 4;
    if (persist) {
        const int full = a_bytes + stages * 32768 + (a_bytes > tile ? a_bytes : tile) + 1024;
        if (full <= 160 * 1024) return full;
        return a_bytes + 32768 + (a_bytes + 32768 > tile ? a_bytes + 32768 : tile) + 1024;  // LATE: [patch 0][stage 0][patch 1][stage 1]
    }
    const int main_loop = 2 * a_bytes + stages * 32768 + 1024;
    return main_loop > tile ? main_loop : tile;
}

static long wino_cu_count() {  // compute units of the current device (MI355X: 256), cached per device index
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

bool conv3x3_wino_serves(long nb, long h, long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo) {
    if (cin % 16 != 0 || cout % 64 != 0 || pad_top < 0 || pad_left < 0 || pad_top > 2 || pad_left > 2 || ho <= 0 || wo <= 0) return false;
    // the patch of a block must cover what its outputs read: any 3x3 / stride-1 geometry does (18 = 16 + 2)
    return nb > 0 && h > 0 && w > 0 && 16L * cin * cout * 4 <= 0x7fffffffL;
}

// Window geometry for maps that 16 x 16 blocks cover badly: the (wg, wty, wtx) with wg * wty * wtx <= 64 that keeps most MFMA rows
// busy over the whole batch (blocks run across images), patch within the LDS allocation; ties: the smaller patch.
struct WinoPlan {
    int kind;  // 0: 16 x 16 blocks, 1: four images of <= 8 x 8, 2: windows
    int wg, wty, wtx, wrow, wimg, wins_x, wins_y;
    double busy;
};
static WinoPlan wino_plan(long nb, long ho, long wo) {
    static const bool no_windows = tia::dev_env("TIA_WINO_NO_WINDOWS") != nullptr;  // developer switch (A/B measurements)
    if (ho <= 8 && wo <= 8) return WinoPlan{1, 0, 0, 0, 0, 0, 0, 0, (double)(ho * wo) / 64.0};
    const long tiles_y = (ho + 1) / 2, tiles_x = (wo + 1) / 2;
    const double busy16 = (double)(ho * wo) / (double)(((ho + 15) / 16) * ((wo + 15) / 16) * 256);
    WinoPlan best{0, 0, 0, 0, 0, 0, 0, 0, busy16};
    if (no_windows || busy16 >= 0.9) return best;
    long best_units = 0;
    for (int wty = 1; wty <= 8; ++wty)
        for (int wtx = 1; wtx <= 16; ++wtx) {
            const int wt = wty * wtx;
            if (wt > 64) continue;
            const int wrow = (wtx + 1) * 9, wimg = ((2 * wty + 2) * wrow + 15) / 16 * 16;
            const long wins_x = (tiles_x + wtx - 1) / wtx, wins_y = (tiles_y + wty - 1) / wty;
            for (int wg = 64 / wt; wg >= 1 && wg > 64 / wt - 3; --wg) {  // (a window less than fit the MFMA rows may fit the LDS)
                if ((long)wg * wimg > 2560 || wg > 16) continue;
                const long blocks = (nb * wins_x * wins_y + wg - 1) / wg;
                const double busy = (double)(nb * ho * wo) / (double)(blocks * 256);
                const long units = (long)wg * wimg;
                if (busy > best.busy + 0.02 || (best.kind == 2 && busy > best.busy - 1e-9 && units < best_units)) {
                    best = WinoPlan{2, wg, wty, wtx, wrow, wimg, (int)wins_x, (int)wins_y, busy};
                    best_units = units;
                }
            }
        }
    return best;
}

int conv3x3_wino_launch(const float* x, const float* u_packed, const float* bias, const float* residual, float* y, long nb, long h,
                        long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int relu, hipStream_t stream) {
    if (!conv3x3_wino_serves(nb, h, w, cin, cout, pad_top, pad_left, ho, wo)) return TIA_ESIZE;
    const WinoPlan plan = wino_plan(nb, ho, wo);
    const bool small = plan.kind == 1;
    const long tiles_y = small ? 1 : (ho + 15) / 16, tiles_x = small ? 1 : (wo + 15) / 16;
    const long n_windows = plan.kind == 2 ? nb * plan.wins_x * plan.wins_y : 0;
    const long tiles = plan.kind == 2 ? (n_windows + plan.wg - 1) / plan.wg : (small ? (nb + 3) / 4 : nb * tiles_y * tiles_x);
    WinoDims d{(int)nb, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)pad_top, (int)pad_left,
               (unsigned)(nb * h * w * cin * 4), (unsigned)(16 * cin * cout * 4), (int)((cin / 16) * (cout / 64) * 4096),
               tia::dev_env("TIA_WINO_ABL") ? atoi(tia::dev_env("TIA_WINO_ABL")) : 0,
               1, 1, 1, 9, 16, 1, 1, 0, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
    if (plan.kind == 2) {
        if (n_windows >= (1L << 24)) return TIA_ESIZE;  // fdiv() range (the callers keep the input below 2 GiB)
        d.wg = plan.wg, d.wty = plan.wty, d.wtx = plan.wtx, d.wrow = plan.wrow, d.wimg = plan.wimg;
        d.wins_x = plan.wins_x, d.wins_y = plan.wins_y, d.n_windows = (int)n_windows;
        d.inv_wimg = 1.0f / (float)plan.wimg, d.inv_wrow = 1.0f / (float)plan.wrow, d.inv_wt = 1.0f / (float)(plan.wty * plan.wtx);
        d.inv_wtx = 1.0f / (float)plan.wtx, d.inv_wins = 1.0f / (float)(plan.wins_x * plan.wins_y), d.inv_wins_x = 1.0f / (float)plan.wins_x;
    }
    // Persistent form (one workgroup per CU walks the items, the next item's first operands requested behind the current one's last
    // steps): 16 x 16 blocks and the four-image blocks of small maps, an even number of 16-channel slices (the patch buffers alternate
    // per slice and an item must end on buffer 1), at least two rounds of items -- everything else one block per workgroup.
    static const bool no_persist = tia::dev_env("TIA_WINO_NO_PERSIST") != nullptr;  // developer switch (A/B measurements)
    const long cus = wino_cu_count() / 8 * 8;
    // (the window geometry keeps one block per workgroup: measured on the 56^2 / 28^2 / 14^2 maps of 224^2 patches the persistent form
    // is -1 % / +1.6 % / +4 % there -- its per-item decode is all run-time divisions -- profiles/r06k_wino_persist_ab_224.txt)
    const bool persist = !no_persist && plan.kind != 2 && (cin / 16) % 2 == 0 && cus >= 8 && tiles * (cout / 64) >= 2 * cus;
    const dim3 grid = persist ? dim3((unsigned)cus) : dim3((unsigned)(((tiles + 7) / 8) * 8), (unsigned)(cout / 64));
    static tia::DeviceOnce attr16, attr8, attrw, attr16p, attr8p;  // the dynamic-LDS attribute is per device
#define TIA_WINO_LAUNCH(GEO_, NS_, PERSIST_, ONCE_)                                                                                  \
    do {                                                                                                                             \
        constexpr int lds = wino_lds_bytes((GEO_::G == 0 ? 1 : GEO_::G) * GEO_::IMG, NS_, PERSIST_);                                \
        static_assert(lds <= 160 * 1024, "LDS");                                                                                     \
        if (!ONCE_.ensure([] {                                                                                                       \
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<GEO_, NS_, PERSIST_>),                  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;                           \
            }))                                                                                                                      \
            return TIA_ELAUNCH;                                                                                                      \
        hipLaunchKernelGGL((conv3x3_wino_kernel<GEO_, NS_, PERSIST_>), grid, dim3(512), lds, stream, x, u_packed, bias, residual, y,  \
                           d, relu, (int)tiles, (int)tiles_x, (int)(tiles_y * tiles_x));                                             \
    } while (0)
    if (plan.kind == 1 && persist)
        TIA_WINO_LAUNCH(W8, 2, true, attr8p);
    else if (plan.kind == 1)
        TIA_WINO_LAUNCH(W8, 2, false, attr8);
    else if (plan.kind == 2)
        TIA_WINO_LAUNCH(WR, 2, false, attrw);
    else if (persist)
        TIA_WINO_LAUNCH(W16, 2, true, attr16p);
    else
        TIA_WINO_LAUNCH(W16, 2, false, attr16);
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
    if (const long even = tia::even_group(n, group); even < group)
        group = (ho <= 8 && wo <= 8 && even > 4) ? (even + 3) / 4 * 4 : even;  // equal groups (still whole blocks, still <= the limit)
    for (long first = 0; first < n; first += group) {
        const long nb = n - first < group ? n - first : group;
        const int rc = tia::conv3x3_wino_launch(d_x + first * h * w * cin, d_u_packed, d_bias, d_residual ? d_residual + first * ho * wo * cout : nullptr,
                                                d_y + first * ho * wo * cout, nb, h, w, cin, cout, pad_top, pad_left, ho, wo, relu,
                                                (hipStream_t)stream);
        if (rc != TIA_OK) return rc;
    }
    return TIA_OK;
}
