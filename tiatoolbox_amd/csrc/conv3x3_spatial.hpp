// Internal interface of conv3x3_spatial.hip (not part of the C ABI): the tap-reuse 3x3 / stride-1 convolution that both
// tia_conv2d_nhwc_f32* (conv_mfma.hip) and tia_conv2d_nhwc_h (conv_mfma_h.hip) dispatch to for maps it covers well.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tia {

// 3x3, stride 1, and the kernel's pixel blocks -- 16 x 16 of one image, or two images of at most 8 x 8 -- waste at most an eighth
// of their pixels on this map (measured: below that the slice kernels win)
inline bool conv3x3_spatial_ok(long kh, long kw, long stride, long ho, long wo) {
    if (kh != 3 || kw != 3 || stride != 1) return false;
    if (ho <= 8 && wo <= 8) return 8 * ho * wo >= 7 * 64;
    const long tiles = ((ho + 15) / 16) * ((wo + 15) / 16);
    return 8 * ho * wo >= 7 * tiles * 256;
}

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

}  // namespace tia
