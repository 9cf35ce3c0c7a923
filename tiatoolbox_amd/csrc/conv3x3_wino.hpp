// Internal interface of conv3x3_wino.hip (not part of the C ABI): the Winograd F(2x2, 3x3) float32 convolution.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.hpp"

namespace tia {

// cin % 16 == 0, cout % 64 == 0, front padding 0..2, stride 1 (any map size; maps of at most 8 x 8 go four images per block)
bool conv3x3_wino_serves(long nb, long h, long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo);

// One launch over `nb` images (input extent < 2 GiB: the caller splits the batch).  `u_packed`: tia_conv_pack_weights_wino_f32.
int conv3x3_wino_launch(const float* x, const float* u_packed, const float* bias, const float* residual, float* y, long nb, long h,
                        long w, long cin, long cout, long pad_top, long pad_left, long ho, long wo, int relu, hipStream_t stream);

}  // namespace tia
