// Internal interface of conv3x3_spatial.hip (not part of the C ABI): the tap-reuse 3x3 / stride-1 convolution that both
// tia_conv2d_nhwc_f32* (conv_mfma.hip) and tia_conv2d_nhwc_h (conv_mfma_h.hip) dispatch to for maps it covers well.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dev_env.hpp"

namespace tia {

// Which block geometry of the tap-reuse kernel serves a 3x3 / stride-1 convolution, if any.
//   kind 1: 16 x 16 pixel blocks of one image;  2: two images of at most 8 x 8;  3: bands of `br` rows of a `bw`-column strip of
//   the batch stacked into one tall image with a zero row between neighbours (conv3x3_spatial.hip: "same" padding only; LDS pixel
//   pitch `bpix` units: 4 for float32, 5 = bank-conflict free for the half kernels);  4: the same with bands of `br` REAL output rows
//   (the rows between two images are in the LDS patch but not among the GEMM rows: busy = br * bw / 256 -- 98-100 % on 56 / 28 / 14 /
//   7 maps); "same" padding or a valid convolution (ho = h - 2: HoVer-Net's decoders, 87-94 % on 16 x 16 blocks).  The fixed geometries need >= 7/8 of their pixels on the map (measured: below that the
//   slice kernels win -- e.g. 76.6 % on the 56 / 28 / 14 / 7 maps of 224^2 patches); the band geometry is taken when it keeps more
//   of the block busy than they do: busy = (br * bw / 256) * (h / (h + 1)).
struct SpPlan {
    int kind, bw, br, brow, strips, bpix;
};
inline SpPlan conv3x3_spatial_plan(long kh, long kw, long stride, long h, long w, long ho, long wo, long pad_top, long pad_left,
                                   bool f32) {
    SpPlan none{0, 0, 0, 0, 0, 0};
    if (kh != 3 || kw != 3 || stride != 1) return none;
    long num = 0, den = 1;  // busy fraction of the best fixed geometry
    int kind = 0;
    if (ho <= 8 && wo <= 8) {
        num = ho * wo, den = 64, kind = 2;
    } else {
        num = ho * wo, den = ((ho + 15) / 16) * ((wo + 15) / 16) * 256, kind = 1;
    }
    SpPlan best = (8 * num >= 7 * den) ? SpPlan{kind, 0, 0, 0, 0, 0} : none;
    double busy = best.kind ? (double)num / (double)den : 0.0;
    static const bool no_band = tia::dev_env("TIA_CONV_NO_BAND") != nullptr;   // developer switches (A/B measurements)
    static const bool no_pack = tia::dev_env("TIA_CONV_BAND_GAPS") != nullptr;  // keep the round-4 form of the band geometry (gap rows computed)
    // bands: "same" padding (one zero row / column all round) or, kind 4 only, a valid convolution (HoVer-Net's decoders)
    const bool same = pad_top == 1 && pad_left == 1 && ho == h && wo == w, valid = pad_top == 0 && pad_left == 0 && ho == h - 2 && wo == w - 2;
    if (!no_band && (same || valid) && ho >= 2) {
        const long gap = same ? 1 : 2;  // virtual (input) rows per image beyond its ho output rows
        SpPlan band = none;
        double band_busy = 0.0;
        static const long max_strips = tia::dev_env("TIA_CONV_BAND_MAX_STRIPS") ? atol(tia::dev_env("TIA_CONV_BAND_MAX_STRIPS")) : 8;
        for (long strips = 1; strips <= max_strips; ++strips) {
            if (wo % strips) continue;
            const long bw = wo / strips;
            if (bw > 128 || bw < 4) continue;
            const long bpix = f32 ? 4 : 5;
            long brow = bpix * (bw + 2);
            if (!f32)
                while (brow % 16 != (5 * bw) % 16) ++brow;  // unit address = 5 p + const (mod 16) along the linear pixel index p
            // kind 3: bands of the stacked batch INCLUDING the zero row between two images (computed and dropped)
            if (const long br = 256 / bw; same && (br + 2) * brow <= 1728) {
                const double b = ((double)(br * bw) / 256.0) * ((double)h / (double)(h + 1));
                if (b > band_busy) band_busy = b, band = SpPlan{3, (int)bw, (int)br, (int)brow, (int)strips, (int)bpix};
            }
            // kind 4 (round 5): bands of `br` REAL output rows -- the patch holds the rows between two images they straddle (at most
            // (br + ho - 2) / ho image boundaries of `gap` rows), the GEMM rows do not; the largest br whose patch fits
            for (long br = no_pack ? 0 : 256 / bw; br >= 1; --br) {
                if ((br + gap * ((br + ho - 2) / ho) + 2) * brow > 1728) continue;
                const double b = (double)(br * bw) / 256.0;
                if (b > band_busy) band_busy = b, band = SpPlan{4, (int)bw, (int)br, (int)brow, (int)strips, (int)bpix};
                break;
            }
        }
        // take a band only for a clear gain over a fixed geometry, and only above 0.88: measured at batch 1024 (profiles/
        // r04e_conv_probe.txt) the tap-reuse kernel sustains ~130 TFLOP/s of raw MFMA work against ~117 for the slice kernel, so a
        // 0.861-busy band (7-wide strips of the round-4 form: 112.4 TFLOP/s) loses to the slice kernel (117.2)
        // (+0.04: a valid 64 -> 62 map is 0.938 busy on 16 x 16 blocks and 0.969 on 31 x 8 bands -- measured 137.7 vs 135.3 TFLOP/s,
        //  profiles/r05zd_hovernet_layers.txt: the fixed geometries' immediate-offset addressing is worth that much)
        if (band.kind && band_busy > busy + 0.04 && band_busy >= 0.88) best = band;
    }
    return best;
}
inline bool conv3x3_spatial_ok(long kh, long kw, long stride, long h, long w, long ho, long wo, long pad_top, long pad_left, bool f32) {
    return conv3x3_spatial_plan(kh, kw, stride, h, w, ho, wo, pad_top, pad_left, f32).kind != 0;
}

// The whole dispatch condition of the tap-reuse kernel for one launch over `nb` images (plan + channel multiples + padding range +
// the band geometry's row-count limit + the developer switch): what conv3x3_spatial_launch checks before it launches.
bool conv3x3_spatial_serves(long nb, long h, long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int dtype);

// One launch over `nb` images (input extent < 2 GiB: the callers split the batch).  dtype: TIA_DT_F32 | _F16 | _BF16.
//   float32: weights packed [3][3][cin][cout] (tia_conv_pack_weights_f32), cin % 16 == 0
//   half:    weights packed [3][3][cin/8][cout][8] (tia_conv_pack_weights_h), cin % 32 == 0
// pad_top / pad_left in {0, 1, 2} zero rows / columns in front; ho / wo: output size (rows / columns past the image read as
// zeros).  Returns false if disabled by the developer switch TIA_CONV_NO_SPATIAL (the caller then uses its slice kernel).
bool conv3x3_spatial_launch(const void* x, const void* w_packed, const float* bias, const void* residual, void* y, long nb, long h,
                            long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int dtype, int relu,
                            hipStream_t stream);

// 1x1 convolutions (float32, stride >= 1, no padding) as a GEMM over 256-pixel blocks with both operands arriving by LDS-DMA in a
// two-stage ring; weights packed [cin][cout] (tia_conv_pack_weights_f32 of a 1x1 kernel).  false: disabled (TIA_CONV_NO_RING).
bool conv1x1_ring_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, long nb, long h, long w,
                         long cin, long cout, long stride, long ho, long wo, int relu, hipStream_t stream);

// The same ring gathering kh x kw taps (any stride, front padding < kernel): the implicit GEMM for the 3x3 layers the tap-reuse
// kernel does not serve.  false: disabled, or a shape the slice kernel handles better (cout % 128 != 0, few workgroups).
bool conv_ring_ok(long nb, long cin, long cout, long kh, long kw, long ho, long wo);  // the dispatch rule alone
bool conv_ring_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, long nb, long h, long w,
                      long cin, long cout, long kh, long kw, long stride, long pad_top, long pad_left, long ho, long wo, int relu,
                      hipStream_t stream);

}  // namespace tia
