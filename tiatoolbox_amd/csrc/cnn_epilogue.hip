// Fused CNN epilogues on gfx950 for NHWC activations (fp32 / fp16 / bf16), HBM-bound:
//   bias_act      : x = relu(x + bias[c] (+ residual))           1 read (+1) + 1 write, in place
//   stem epilogue : out = maxpool3x3s2p1(relu(x + bias[c]))      reads the conv1 output once
// 16 bytes per lane per access (8 halves / 4 floats); consecutive lanes walk the channel axis so a
// wave covers 1 KiB of contiguous NHWC memory per instruction.  Arithmetic in fp32, same order as
// the unfused torch ops ((x + b) + r, then max(.,0)).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace tia {

constexpr int ET = 256;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

template <class T>
struct Vec;  // 16-byte vector of T <-> float lanes
template <>
struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const u4& v, float (&f)[8]) {
        f[0] = __uint_as_float(v.x);
        f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z);
        f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ u4 pack(const float (&f)[8]) {
        u4 v;
        v.x = __float_as_uint(f[0]);
        v.y = __float_as_uint(f[1]);
        v.z = __float_as_uint(f[2]);
        v.w = __float_as_uint(f[3]);
        return v;
    }
};
template <>
struct Vec<__half> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const u4& v, float (&f)[8]) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __half2float(__ushort_as_half((unsigned short)(w[i] & 0xffffu)));
            f[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(w[i] >> 16)));
        }
    }
    static __device__ __forceinline__ u4 pack(const float (&f)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (unsigned)__half_as_ushort(__float2half_rn(f[2 * i])) |
                   ((unsigned)__half_as_ushort(__float2half_rn(f[2 * i + 1])) << 16);
        u4 v;
        v.x = w[0];
        v.y = w[1];
        v.z = w[2];
        v.w = w[3];
        return v;
    }
};
template <>
struct Vec<__hip_bfloat16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const u4& v, float (&f)[8]) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ unsigned bf(float x) {  // round to nearest even
        unsigned u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    }
    static __device__ __forceinline__ u4 pack(const float (&f)[8]) {
        u4 v;
        v.x = bf(f[0]) | (bf(f[1]) << 16);
        v.y = bf(f[2]) | (bf(f[3]) << 16);
        v.z = bf(f[4]) | (bf(f[5]) << 16);
        v.w = bf(f[6]) | (bf(f[7]) << 16);
        return v;
    }
};

template <class T, bool RES>
__global__ __launch_bounds__(ET) void bias_act_kernel(u4* __restrict__ x, const u4* __restrict__ bias, const u4* __restrict__ res,
                                                       long nvec, int cvec, int relu) {
    constexpr int N = Vec<T>::N;
    const long stride = (long)gridDim.x * ET;
    for (long i = (long)blockIdx.x * ET + threadIdx.x; i < nvec; i += stride) {
        float a[8], b[8], r[8];
        const u4 xv = x[i];
        Vec<T>::unpack(xv, a);
        Vec<T>::unpack(bias[i % cvec], b);
        if (RES) Vec<T>::unpack(__builtin_nontemporal_load(res + i), r);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float v = a[k] + b[k];
            if (RES) v = v + r[k];
            a[k] = (relu && !(v > 0.0f)) ? 0.0f : v;
        }
        x[i] = Vec<T>::pack(a);
    }
}

template <class T>
__global__ __launch_bounds__(ET) void stem_pool_kernel(const u4* __restrict__ x, const u4* __restrict__ bias, int h, int w, int cvec,
                                                        int ho, int wo, long nvec_out, u4* __restrict__ out) {
    constexpr int N = Vec<T>::N;
    const long stride = (long)gridDim.x * ET;
    for (long i = (long)blockIdx.x * ET + threadIdx.x; i < nvec_out; i += stride) {
        const int cv = (int)(i % cvec);
        long t = i / cvec;
        const int ox = (int)(t % wo);
        t /= wo;
        const int oy = (int)(t % ho);
        const long img = t / ho;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -3.4028234663852886e38f;
        // max over the valid 3x3 window of the raw conv output: +bias and ReLU are monotone, so they
        // commute with the maximum and are applied once afterwards
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * oy + dy;
            if (yy < 0 || yy >= h) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = 2 * ox + dx;
                if (xx < 0 || xx >= w) continue;
                float a[8];
                Vec<T>::unpack(x[((img * h + yy) * w + xx) * cvec + cv], a);
#pragma unroll
                for (int k = 0; k < N; ++k) m[k] = a[k] > m[k] ? a[k] : m[k];
            }
        }
        float b[8];
        Vec<T>::unpack(bias[cv], b);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float v = m[k] + b[k];
            m[k] = v > 0.0f ? v : 0.0f;
        }
        out[i] = Vec<T>::pack(m);
    }
}

static inline unsigned eblocks(long n) {
    long b = (n + ET - 1) / ET;
    if (b > 256L * 32) b = 256L * 32;
    return (unsigned)(b < 1 ? 1 : b);
}

template <class T>
static int launch_bias_act(void* x, const void* bias, const void* res, long rows, long c, int relu, hipStream_t st) {
    constexpr int N = Vec<T>::N;
    if (c % N) return TIA_EINVAL;
    const long nvec = rows * c / N;
    if (res)
        hipLaunchKernelGGL((bias_act_kernel<T, true>), dim3(eblocks(nvec)), dim3(ET), 0, st, (u4*)x, (const u4*)bias, (const u4*)res,
                           nvec, (int)(c / N), relu);
    else
        hipLaunchKernelGGL((bias_act_kernel<T, false>), dim3(eblocks(nvec)), dim3(ET), 0, st, (u4*)x, (const u4*)bias, (const u4*)nullptr,
                           nvec, (int)(c / N), relu);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

template <class T>
static int launch_stem(const void* x, const void* bias, long n, long h, long w, long c, void* out, hipStream_t st) {
    constexpr int N = Vec<T>::N;
    if (c % N) return TIA_EINVAL;
    const long ho = (h + 1) / 2, wo = (w + 1) / 2;
    const long nvec = n * ho * wo * c / N;
    hipLaunchKernelGGL((stem_pool_kernel<T>), dim3(eblocks(nvec)), dim3(ET), 0, st, (const u4*)x, (const u4*)bias, (int)h, (int)w,
                       (int)(c / N), (int)ho, (int)wo, nvec, (u4*)out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}


// y = relu(x * scale[c] + shift[c])  -- an inference BatchNorm (+ ReLU) that cannot be folded into a convolution (the
// pre-activation units of HoVer-Net: BN -> ReLU -> conv).  float32, 16 bytes per lane, y may alias x.  The product and
// the sum are rounded separately (contract off above), like torch's eval-mode batch_norm followed by relu.
__global__ __launch_bounds__(ET) void scale_shift_act_kernel(const u4* __restrict__ x, const u4* __restrict__ scale,
                                                              const u4* __restrict__ shift, long total_v, int cv, int relu,
                                                              u4* __restrict__ y) {
    for (long i = (long)blockIdx.x * ET + threadIdx.x; i < total_v; i += (long)gridDim.x * ET) {
        const int c = (int)(i % cv);
        float f[8], sc[8], sh[8];
        Vec<float>::unpack(x[i], f);
        Vec<float>::unpack(scale[c], sc);
        Vec<float>::unpack(shift[c], sh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = f[k] * sc[k];
            v = v + sh[k];
            f[k] = relu ? (v > 0.0f ? v : 0.0f) : v;
        }
        y[i] = Vec<float>::pack(f);
    }
}


// out[b, Y, X, :] = x[b, Y / 2, X / 2, :] + y[b, Y, X, :]  -- nearest-neighbour x2 upsampling fused with the skip-connection
// add of the HoVer-Net decoders (hovernet.py:447-449: `upsample2x(d3) + d2`); y may be a centre-cropped VIEW (row and image
// strides given in elements, channels contiguous).  float32, 16 bytes per lane.
// (optionally followed by relu(. * scale[c] + shift[c]): the pre-activation of the UNet decoder blocks, unet.py:193-240)
__global__ __launch_bounds__(ET) void upsample2x_add_kernel(const u4* __restrict__ x, const float* __restrict__ y, long y_sb, long y_sy,
                                                             int n, int h, int w, int cv, const u4* __restrict__ scale,
                                                             const u4* __restrict__ shift, u4* __restrict__ out) {
    const long total = (long)n * (2 * h) * (2 * w) * cv;
    for (long i = (long)blockIdx.x * ET + threadIdx.x; i < total; i += (long)gridDim.x * ET) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int X = (int)(t % (2 * w));
        t /= 2 * w;
        const int Y = (int)(t % (2 * h)), b = (int)(t / (2 * h));
        float a[8], r[8];
        Vec<float>::unpack(x[(((long)b * h + (Y >> 1)) * w + (X >> 1)) * cv + c], a);
        Vec<float>::unpack(*reinterpret_cast<const u4*>(y + (long)b * y_sb + (long)Y * y_sy + ((long)X * cv + c) * 4), r);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = a[k] + r[k];
        if (scale) {
            float sc[8], sh[8];
            Vec<float>::unpack(scale[c], sc);
            Vec<float>::unpack(shift[c], sh);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = a[k] * sc[k];
                v = v + sh[k];
                a[k] = v > 0.0f ? v : 0.0f;
            }
        }
        out[i] = Vec<float>::pack(a);
    }
}


// The same on a VIEW of an NHWC tensor (a channel prefix and a spatial window of a wider buffer: image / row / pixel
// strides in elements), written densely: what lets the dense units of HoVer-Net grow their feature stack in place
// instead of re-concatenating it (hovernet.py:93-98).
__global__ __launch_bounds__(ET) void scale_shift_act_view_kernel(const float* __restrict__ x, long sb, long sy, long sp,
                                                                   const u4* __restrict__ scale, const u4* __restrict__ shift, int n,
                                                                   int h, int w, int cv, int relu, u4* __restrict__ y) {
    const long total = (long)n * h * w * cv;
    for (long i = (long)blockIdx.x * ET + threadIdx.x; i < total; i += (long)gridDim.x * ET) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int px = (int)(t % w);
        t /= w;
        const int py = (int)(t % h), b = (int)(t / h);
        float f[8], sc[8], sh[8];
        Vec<float>::unpack(*reinterpret_cast<const u4*>(x + (long)b * sb + (long)py * sy + (long)px * sp + 4 * c), f);
        Vec<float>::unpack(scale[c], sc);
        Vec<float>::unpack(shift[c], sh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = f[k] * sc[k];
            v = v + sh[k];
            f[k] = relu ? (v > 0.0f ? v : 0.0f) : v;
        }
        y[i] = Vec<float>::pack(f);
    }
}


// Grouped "valid" k x k convolution with few channels per group (HoVer-Net's dense units: 128 -> 32 channels in 4 groups of
// 32 -> 8, hovernet.py:86-88), float32 NHWC.  Too narrow for the 32-wide MFMA tile (a block-diagonal dense formulation
// wastes three quarters of the matrix work), so: one thread per output pixel and group, 8 accumulators, the pixel's 32
// input channels as eight 16-byte loads per tap, the group's weights wave-uniform (scalar loads), fmaf chain in
// (tap, channel) order.  The result may go into a channel slice / spatial window of a wider buffer (strides in elements).
template <int CPG, int OPG>
__global__ __launch_bounds__(ET) void grouped_conv_valid_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                 float* __restrict__ y, int n, int h, int w, int groups, int k,
                                                                 long y_sb, long y_sy, long y_sp) {
    const int g = blockIdx.y;
    const int ho = h - k + 1, wo = w - k + 1;
    const long m_total = (long)n * ho * wo;
    const long m = (long)blockIdx.x * ET + threadIdx.x;
    const bool valid = m < m_total;
    const long mm = valid ? m : 0;
    const int b = (int)(mm / ((long)ho * wo));
    const int rem = (int)(mm - (long)b * ho * wo);
    const int oy = rem / wo, ox = rem - oy * wo;
    const int cin = groups * CPG;
    const float* xp = x + (((long)b * h + oy) * w + ox) * cin + g * CPG;
    const float* wg = wpk + (long)g * k * k * CPG * OPG;
    float acc[OPG];
#pragma unroll
    for (int j = 0; j < OPG; ++j) acc[j] = 0.0f;
    for (int ky = 0; ky < k; ++ky) {
        for (int kx = 0; kx < k; ++kx) {
            const float4* xr = reinterpret_cast<const float4*>(xp + ((long)ky * w + kx) * cin);
            const float* wr = wg + (long)(ky * k + kx) * CPG * OPG;
#pragma unroll
            for (int c4 = 0; c4 < CPG / 4; ++c4) {
                const float4 v = xr[c4];
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < OPG; ++j) acc[j] = fmaf(e[q], wr[(4 * c4 + q) * OPG + j], acc[j]);
            }
        }
    }
    if (valid) {
        float* yp = y + (long)b * y_sb + (long)oy * y_sy + (long)ox * y_sp + g * OPG;
#pragma unroll
        for (int j = 0; j < OPG; j += 4) *reinterpret_cast<float4*>(yp + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
}


// ---- class heads: 1x1 convolution 64 -> COUT <= 8 channels, optionally with the preceding BatchNorm + ReLU on load ----------
// HBM-bound (256 B read per pixel, 4 * COUT written).  16 lanes share a pixel (one float4 of its 64 channels each: a wave
// reads 4 pixels = 1 KB contiguous), multiply by their slice of the weights and fold the 16 partial sums with a butterfly
// of lane exchanges; lanes 0 .. COUT-1 of each 16 then write the pixel's outputs (4 * COUT contiguous floats per wave).
template <int COUT>
__global__ __launch_bounds__(256) void head1x1_kernel(const float4* __restrict__ x, long npix, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ ps,
                                                      const float* __restrict__ pt, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, q = lane & 15, sub = lane >> 4;
    float wr[COUT][4];
#pragma unroll
    for (int o = 0; o < COUT; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[o][i] = w[o * 64 + 4 * q + i];
    float sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, sh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool pre = ps != nullptr;
    if (pre) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = ps[4 * q + i];
            sh[i] = pt[4 * q + i];
        }
    }
    float bo = 0.0f;
    if (bias && q < COUT) bo = bias[q];
    const long groups = (npix + 3) / 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long)gridDim.x * 4;
    constexpr int U = 4;  // groups in flight per wave
    for (long g0 = wave * U; g0 < groups; g0 += waves * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = (g0 + u) * 4 + sub;
            v[u] = pix < npix ? x[pix * 16 + q] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (pre) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = __fadd_rn(__fmul_rn(a[i], sc[i]), sh[i]);  // rounded like batch_norm, then relu
                    a[i] = t > 0.0f ? t : 0.0f;
                }
            }
            float part[COUT];
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                float t = a[0] * wr[o][0];
                t = fmaf(a[1], wr[o][1], t);
                t = fmaf(a[2], wr[o][2], t);
                part[o] = fmaf(a[3], wr[o][3], t);
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                for (int o = 0; o < COUT; ++o) part[o] += __shfl_xor(part[o], m, 64);
            float mine = part[0];
#pragma unroll
            for (int o = 1; o < COUT; ++o) mine = q == o ? part[o] : mine;
            const long pix = (g0 + u) * 4 + sub;
            if (q < COUT && pix < npix) y[pix * COUT + q] = mine + bo;
        }
    }
}

template <int COUT>
static void launch_head(const float* x, long npix, const float* w, const float* bias, const float* ps, const float* pt, float* y,
                        hipStream_t st) {
    long blocks = ((npix + 3) / 4 + 15) / 16;  // 4 waves x 4 groups per pass
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipLaunchKernelGGL(head1x1_kernel<COUT>, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)x, npix, w, bias, ps, pt, y);
}
}  // namespace tia

using namespace tia;

extern "C" int tia_bias_act_nhwc(void* d_x, const void* d_bias, const void* d_residual, int64_t rows, int64_t c, int32_t dtype,
                                  int32_t relu, void* stream) {
    if (!d_x || !d_bias || rows <= 0 || c <= 0) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_bias) | reinterpret_cast<uintptr_t>(d_residual)) & 15)
        return TIA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case TIA_DT_F32: return launch_bias_act<float>(d_x, d_bias, d_residual, rows, c, relu, st);
        case TIA_DT_F16: return launch_bias_act<__half>(d_x, d_bias, d_residual, rows, c, relu, st);
        case TIA_DT_BF16: return launch_bias_act<__hip_bfloat16>(d_x, d_bias, d_residual, rows, c, relu, st);
        default: return TIA_EINVAL;
    }
}

extern "C" int tia_bias_relu_maxpool_nhwc(const void* d_x, const void* d_bias, int64_t n, int64_t h, int64_t w, int64_t c,
                                           int32_t dtype, void* d_out, void* stream) {
    if (!d_x || !d_bias || !d_out || n <= 0 || h <= 0 || w <= 0 || c <= 0) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_bias) | reinterpret_cast<uintptr_t>(d_out)) & 15)
        return TIA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case TIA_DT_F32: return launch_stem<float>(d_x, d_bias, n, h, w, c, d_out, st);
        case TIA_DT_F16: return launch_stem<__half>(d_x, d_bias, n, h, w, c, d_out, st);
        case TIA_DT_BF16: return launch_stem<__hip_bfloat16>(d_x, d_bias, n, h, w, c, d_out, st);
        default: return TIA_EINVAL;
    }
}

extern "C" int tia_scale_shift_act_nhwc_f32(const float* d_x, const float* d_scale, const float* d_shift, float* d_y, int64_t rows,
                                             int64_t c, int32_t relu, void* stream) {
    if (!d_x || !d_scale || !d_shift || !d_y || rows <= 0 || c <= 0 || (c & 3) != 0) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_scale) | reinterpret_cast<uintptr_t>(d_shift) |
         reinterpret_cast<uintptr_t>(d_y)) & 15)
        return TIA_EINVAL;
    const long total_v = rows * (c / 4);
    long blocks = (total_v + ET - 1) / ET;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL(scale_shift_act_kernel, dim3((unsigned)blocks), dim3(ET), 0, (hipStream_t)stream, (const u4*)d_x,
                       (const u4*)d_scale, (const u4*)d_shift, total_v, (int)(c / 4), relu, (u4*)d_y);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_upsample2x_add_act_nhwc_f32(const float* d_x, const float* d_y, int64_t y_image_stride, int64_t y_row_stride,
                                                const float* d_scale, const float* d_shift, float* d_out, int64_t n, int64_t h,
                                                int64_t w, int64_t c, void* stream) {
    if (!d_x || !d_y || !d_out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) != 0) return TIA_EINVAL;
    if ((d_scale == nullptr) != (d_shift == nullptr)) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_scale) | reinterpret_cast<uintptr_t>(d_shift)) & 15) return TIA_EINVAL;
    if ((y_image_stride & 3) != 0 || (y_row_stride & 3) != 0 || y_row_stride < 2 * w * c) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y) | reinterpret_cast<uintptr_t>(d_out)) & 15) return TIA_EINVAL;
    const long total = n * 4 * h * w * (c / 4);
    long blocks = (total + ET - 1) / ET;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3((unsigned)blocks), dim3(ET), 0, (hipStream_t)stream, (const u4*)d_x, d_y,
                       (long)y_image_stride, (long)y_row_stride, (int)n, (int)h, (int)w, (int)(c / 4), (const u4*)d_scale,
                       (const u4*)d_shift, (u4*)d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_upsample2x_add_nhwc_f32(const float* d_x, const float* d_y, int64_t y_image_stride, int64_t y_row_stride,
                                            float* d_out, int64_t n, int64_t h, int64_t w, int64_t c, void* stream) {
    return tia_upsample2x_add_act_nhwc_f32(d_x, d_y, y_image_stride, y_row_stride, nullptr, nullptr, d_out, n, h, w, c, stream);
}

extern "C" int tia_scale_shift_act_view_nhwc_f32(const float* d_x, int64_t x_image_stride, int64_t x_row_stride, int64_t x_pixel_stride,
                                                  const float* d_scale, const float* d_shift, float* d_y, int64_t n, int64_t h,
                                                  int64_t w, int64_t c, int32_t relu, void* stream) {
    if (!d_x || !d_scale || !d_shift || !d_y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) != 0) return TIA_EINVAL;
    if (((x_image_stride | x_row_stride | x_pixel_stride) & 3) != 0 || x_pixel_stride < c) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_scale) | reinterpret_cast<uintptr_t>(d_shift) |
         reinterpret_cast<uintptr_t>(d_y)) & 15)
        return TIA_EINVAL;
    const long total = n * h * w * (c / 4);
    long blocks = (total + ET - 1) / ET;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL(scale_shift_act_view_kernel, dim3((unsigned)blocks), dim3(ET), 0, (hipStream_t)stream, d_x, (long)x_image_stride,
                       (long)x_row_stride, (long)x_pixel_stride, (const u4*)d_scale, (const u4*)d_shift, (int)n, (int)h, (int)w,
                       (int)(c / 4), relu, (u4*)d_y);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_grouped_conv_valid_nhwc_f32(const float* d_x, const float* d_w_packed, float* d_y, int64_t y_image_stride,
                                                int64_t y_row_stride, int64_t y_pixel_stride, int64_t n, int64_t h, int64_t w,
                                                int64_t groups, int64_t cin_per_group, int64_t cout_per_group, int64_t k,
                                                void* stream) {
    if (!d_x || !d_w_packed || !d_y || n <= 0 || groups <= 0 || groups > 65535 || k <= 0 || h < k || w < k) return TIA_EINVAL;
    if (cin_per_group != 32 || cout_per_group != 8) return TIA_ESIZE;
    if (((y_image_stride | y_row_stride | y_pixel_stride) & 3) != 0 || y_pixel_stride < groups * cout_per_group) return TIA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_w_packed) | reinterpret_cast<uintptr_t>(d_y)) & 15) return TIA_EINVAL;
    const long m_total = n * (h - k + 1) * (w - k + 1);
    const long blocks = (m_total + ET - 1) / ET;
    if (blocks > 0x7fffffffL) return TIA_ESIZE;
    hipLaunchKernelGGL((grouped_conv_valid_kernel<32, 8>), dim3((unsigned)blocks, (unsigned)groups), dim3(ET), 0, (hipStream_t)stream, d_x,
                       d_w_packed, d_y, (int)n, (int)h, (int)w, (int)groups, (int)k, (long)y_image_stride, (long)y_row_stride,
                       (long)y_pixel_stride);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_conv1x1_head_nhwc_f32(const float* d_x, int64_t npix, const float* d_w, const float* d_bias, const float* d_pre_scale,
                                         const float* d_pre_shift, int32_t cout, float* d_y, void* stream) {
    if (!d_x || !d_w || !d_y || npix <= 0 || cout < 1 || cout > 8 || ((d_pre_scale == nullptr) != (d_pre_shift == nullptr))) return TIA_EINVAL;
    if (reinterpret_cast<uintptr_t>(d_x) & 15) return TIA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (cout) {
        case 1: launch_head<1>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        case 2: launch_head<2>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        case 3: launch_head<3>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        case 4: launch_head<4>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        case 5: launch_head<5>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        case 6: launch_head<6>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        case 7: launch_head<7>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
        default: launch_head<8>(d_x, npix, d_w, d_bias, d_pre_scale, d_pre_shift, d_y, st); break;
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
