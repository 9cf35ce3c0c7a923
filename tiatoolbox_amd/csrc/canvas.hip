// Overlap-average stitching of patch probabilities into a whole-slide canvas on gfx950.
// Reference: models/engine/semantic_segmentor.py:1141-1263 (merge_batch_to_canvas / merge_horizontal),
// :1398-1534 (merge_vertical_chunkwise), patch_predictor.py:382-446 (argmax).
// Gather formulation: one thread owns one canvas element and sums its (<= a few) contributing blocks in
// the reference's order, so results are deterministic and bit-identical to the NumPy path -- no float atomics.
#include "common.hpp"

#pragma clang fp contract(off)

namespace tia {

constexpr int CT = 256;

__global__ __launch_bounds__(CT) void block_flags_kernel(const float* __restrict__ blocks, long per_block, int* __restrict__ flags) {
    // flags[b] = any(block b != 0)
    const float* p = blocks + (size_t)blockIdx.y * per_block;
    int any = 0;
    for (long i = (long)blockIdx.x * CT + threadIdx.x; i < per_block; i += (long)gridDim.x * CT) any |= (p[i] != 0.0f);
    if (__ballot(any) != 0ull && lane_id() == 0) atomicOr(&flags[blockIdx.y], 1);
}

__global__ __launch_bounds__(CT) void row_merge_kernel(const float* __restrict__ blocks, const int* __restrict__ xs,
                                                        const int* __restrict__ flags, int n, int oh, int ow, int c, int width,
                                                        float* __restrict__ row, uint8_t* __restrict__ cnt) {
    const long total = (long)oh * width;
    for (long i = (long)blockIdx.x * CT + threadIdx.x; i < total; i += (long)gridDim.x * CT) {
        const int y = (int)(i / width), x = (int)(i - (long)y * width);
        float acc[8];
        for (int k = 0; k < c; ++k) acc[k] = 0.0f;
        unsigned count = 0;
        for (int b = 0; b < n; ++b) {
            const int x0 = xs[b];
            if (x < x0 || x >= x0 + ow || !flags[b]) continue;
            const float* src = blocks + (((size_t)b * oh + y) * ow + (x - x0)) * c;
            for (int k = 0; k < c; ++k) acc[k] = acc[k] + src[k];
            ++count;
        }
        float* dst = row + (size_t)i * c;
        for (int k = 0; k < c; ++k) dst[k] = acc[k];
        cnt[i] = (uint8_t)count;
    }
}

__global__ __launch_bounds__(CT) void finalize_kernel(const float* __restrict__ row_a, const uint8_t* __restrict__ cnt_a, long ys_a,
                                                       const float* __restrict__ row_b, const uint8_t* __restrict__ cnt_b, long ys_b,
                                                       int oh, int width, int c, long y_begin, long y_end,
                                                       float* __restrict__ probs, uint8_t* __restrict__ pred) {
    const long total = (y_end - y_begin) * width;
    for (long i = (long)blockIdx.x * CT + threadIdx.x; i < total; i += (long)gridDim.x * CT) {
        const long y = y_begin + i / width;
        const int x = (int)(i % width);
        const long ya = y - ys_a;
        const bool has_a = ya >= 0 && ya < oh;
        const long yb = y - ys_b;
        const bool has_b = row_b != nullptr && yb >= 0 && yb < oh;
        unsigned count = (has_a ? cnt_a[ya * width + x] : 0u) + (has_b ? cnt_b[yb * width + x] : 0u);
        count = (uint8_t)count;          // numpy uint8 arithmetic
        const float denom = (float)(count == 0 ? 1u : count);
        int best = 0;
        float bestv = 0.0f;
        for (int k = 0; k < c; ++k) {
            float v = has_a ? row_a[((size_t)ya * width + x) * c + k] : 0.0f;
            if (has_b) v = v + row_b[((size_t)yb * width + x) * c + k];
            v = v / denom;
            if (probs) probs[((size_t)y * width + x) * c + k] = v;
            if (k == 0 || v > bestv) {
                bestv = v;
                best = k;
            }
        }
        pred[(size_t)y * width + x] = (uint8_t)best;
    }
}


// ---- patch reads from an in-memory slide level ---------------------------------------------------------------------
// WSIPatchDataset.__getitem__ / read_bounds(..., pad_constant_values=255) (models/dataset/dataset_abc.py:418-448) for a
// whole batch of equally sized bounds at once: out[m, y, x, :] = slide[y0+y, x0+x, :] or `pad` outside the slide.
// One thread produces 4 consecutive output bytes (one dword store; rows of ph*pw*c bytes are dword multiples, checked
// by the launcher), reading the source with one dword load when the source run is in bounds and aligned.
__global__ __launch_bounds__(CT) void gather_patches_kernel(const uint8_t* __restrict__ slide, int sh, int sw, int c,
                                                             const int* __restrict__ bounds, int ph, int pw, int pad,
                                                             uint8_t* __restrict__ out) {
    const long row_bytes = (long)pw * c;
    const long patch_bytes = (long)ph * row_bytes;
    const int m = blockIdx.y;
    const int x0 = bounds[m * 4 + 0], y0 = bounds[m * 4 + 1];
    uint8_t* dst = out + (size_t)m * patch_bytes;
    const uint32_t pad4 = 0x01010101u * (uint32_t)(pad & 255);
    for (long i = ((long)blockIdx.x * CT + threadIdx.x) * 4; i < patch_bytes; i += (long)gridDim.x * CT * 4) {
        const int y = (int)(i / row_bytes);
        const long xb = i - (long)y * row_bytes;       // byte offset inside the patch row
        const int sy = y0 + y;
        uint32_t v = pad4;
        if (sy >= 0 && sy < sh) {
            const long sb = (long)x0 * c + xb;         // byte offset inside the slide row (may be negative)
            const uint8_t* srow = slide + (size_t)sy * sw * c;
            const long lim = (long)sw * c;
            if (sb >= 0 && sb + 4 <= lim && xb + 4 <= row_bytes) {
                const uint8_t* sp = srow + sb;
                if ((reinterpret_cast<uintptr_t>(sp) & 3) == 0) {
                    v = *reinterpret_cast<const uint32_t*>(sp);
                } else {
                    v = (uint32_t)sp[0] | ((uint32_t)sp[1] << 8) | ((uint32_t)sp[2] << 16) | ((uint32_t)sp[3] << 24);
                }
            } else {  // the dword straddles the slide edge or the end of the patch row: byte by byte
                v = 0;
                for (int k = 0; k < 4; ++k) {
                    long xk = xb + k;
                    int yk = y;
                    if (xk >= row_bytes) {  // next patch row
                        xk -= row_bytes;
                        ++yk;
                    }
                    const int syk = y0 + yk;
                    const long sbk = (long)x0 * c + xk;
                    uint32_t b = (uint32_t)(pad & 255);
                    if (yk < ph && syk >= 0 && syk < sh && sbk >= 0 && sbk < lim) b = slide[(size_t)syk * sw * c + sbk];
                    v |= b << (8 * k);
                }
            }
        } else if (xb + 4 > row_bytes) {  // padded row whose dword runs into the next (possibly valid) row
            v = 0;
            for (int k = 0; k < 4; ++k) {
                long xk = xb + k;
                int yk = y;
                if (xk >= row_bytes) {
                    xk -= row_bytes;
                    ++yk;
                }
                const int syk = y0 + yk;
                const long sbk = (long)x0 * c + xk;
                uint32_t b = (uint32_t)(pad & 255);
                if (yk < ph && syk >= 0 && syk < sh && sbk >= 0 && sbk < (long)sw * c) b = slide[(size_t)syk * sw * c + sbk];
                v |= b << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t*>(dst + i) = v;
    }
}

}  // namespace tia

using namespace tia;

extern "C" int tia_canvas_row_merge_f32(const float* d_blocks, const int32_t* d_xs, int64_t n, int64_t oh, int64_t ow, int64_t c,
                                         int64_t width, float* d_row, uint8_t* d_cnt, int32_t* d_flags, void* stream) {
    if (!d_blocks || !d_xs || !d_row || !d_cnt || !d_flags) return TIA_EINVAL;
    if (n <= 0 || oh <= 0 || ow <= 0 || c <= 0 || c > 8 || width <= 0 || n > 65535) return TIA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_flags, 0, (size_t)n * sizeof(int32_t), st) != hipSuccess) return TIA_ELAUNCH;
    const long per_block = (long)oh * ow * c;
    hipLaunchKernelGGL(block_flags_kernel, dim3(64, (unsigned)n), dim3(CT), 0, st, d_blocks, per_block, d_flags);
    const long total = (long)oh * width;
    long nb = (total + CT - 1) / CT;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(row_merge_kernel, dim3((unsigned)nb), dim3(CT), 0, st, d_blocks, d_xs, d_flags, (int)n, (int)oh, (int)ow, (int)c,
                       (int)width, d_row, d_cnt);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_canvas_finalize_f32(const float* d_row_a, const uint8_t* d_cnt_a, int64_t ys_a, const float* d_row_b,
                                        const uint8_t* d_cnt_b, int64_t ys_b, int64_t oh, int64_t width, int64_t c, int64_t y_begin,
                                        int64_t y_end, float* d_probs, uint8_t* d_pred, void* stream) {
    if (!d_row_a || !d_cnt_a || !d_pred || (d_row_b && !d_cnt_b)) return TIA_EINVAL;
    if (oh <= 0 || width <= 0 || c <= 0 || c > 255 || y_end < y_begin) return TIA_EINVAL;
    if (y_end == y_begin) return TIA_OK;
    const long total = (y_end - y_begin) * width;
    long nb = (total + CT - 1) / CT;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nb), dim3(CT), 0, (hipStream_t)stream, d_row_a, d_cnt_a, (long)ys_a, d_row_b,
                       d_cnt_b, (long)ys_b, (int)oh, (int)width, (int)c, (long)y_begin, (long)y_end, d_probs, d_pred);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_gather_patches_u8(const uint8_t* d_slide, int64_t sh, int64_t sw, int64_t c, const int32_t* d_bounds,
                                     int64_t m, int64_t ph, int64_t pw, int32_t pad, uint8_t* d_out, void* stream) {
    if (!d_slide || !d_bounds || !d_out || sh <= 0 || sw <= 0 || c <= 0 || m < 0 || ph <= 0 || pw <= 0) return TIA_EINVAL;
    if (m == 0) return TIA_OK;
    const long patch_bytes = (long)ph * pw * c;
    if ((patch_bytes & 3) != 0 || m > 65535 || sh > 0x7fffffffL || sw * c > 0x7fffffffL) return TIA_ESIZE;
    if ((reinterpret_cast<uintptr_t>(d_out) & 3) != 0) return TIA_EINVAL;
    long blocks = (patch_bytes / 4 + tia::CT - 1) / tia::CT;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(tia::gather_patches_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(tia::CT), 0, (hipStream_t)stream,
                       d_slide, (int)sh, (int)sw, (int)c, d_bounds, (int)ph, (int)pw, pad, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
