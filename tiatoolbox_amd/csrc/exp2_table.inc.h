// 255 * 2^(i/1024): the table of the float64 stain_apply kernel (generated values in exp2_table.inc).
#pragma once
__device__ const double kExp2Tab255[1024] = {
#include "exp2_table.inc"
};
