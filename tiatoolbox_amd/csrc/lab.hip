// OpenCV 8-bit RGB<->Lab on gfx950 (integer fixed point, all tables in LDS) and the Reinhard
// normaliser built on it.  Reference: tools/stainnorm.py:222-367 (cv2.cvtColor RGB2LAB / LAB2RGB,
// cv2.meanStdDev).  Per pixel everything is integer table look-ups + integer MACs: HBM-bound.
#include "common.hpp"

#pragma clang fp contract(off)  // the LUT arithmetic restates NumPy float32/float64 expressions term by term

namespace tia {

constexpr int LT = 256;

struct LabLds {
    uint16_t gamma[256];
    uint16_t cbrt[3072];
    uint16_t lab_y[256];
    uint16_t lab_ify[256];
    uint8_t inv_gamma[4096];
    int c_fwd[9];
    int c_inv[9];
};

__device__ __forceinline__ void load_tables(LabLds& s, const tia_lab_tables* __restrict__ t) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(t);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s);
    static_assert(sizeof(LabLds) == sizeof(tia_lab_tables), "table layout");
    for (int i = threadIdx.x; i < (int)(sizeof(LabLds) / 4); i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ void rgb2lab(const LabLds& s, uint32_t r, uint32_t g, uint32_t b, int& L, int& A, int& B) {
    const int R = s.gamma[r], G = s.gamma[g], Bl = s.gamma[b];
    const int fx = s.cbrt[descale(R * s.c_fwd[0] + G * s.c_fwd[1] + Bl * s.c_fwd[2], 12)];
    const int fy = s.cbrt[descale(R * s.c_fwd[3] + G * s.c_fwd[4] + Bl * s.c_fwd[5], 12)];
    const int fz = s.cbrt[descale(R * s.c_fwd[6] + G * s.c_fwd[7] + Bl * s.c_fwd[8], 12)];
    L = sat_u8(descale(296 * fy - 1336934, 15));
    A = sat_u8(descale(500 * (fx - fy) + 128 * (1 << 15), 15));
    B = sat_u8(descale(200 * (fy - fz) + 128 * (1 << 15), 15));
}

// abToXZ_b of OpenCV computed directly (C integer division truncates toward zero)
__device__ __forceinline__ int ab_to_xz(int i) {
    if (i <= 3390) return i * 108 / 841 - (16384 * 16 / 116 * 108 / 841);
    const long long q = (long long)i * i / 16384;
    return (int)(q * i / 16384);
}

__device__ __forceinline__ void lab2rgb(const LabLds& s, int L, int A, int B, int& r, int& g, int& b) {
    const int y = s.lab_y[L], ify = s.lab_ify[L];
    const int adiv = ((5 * A * 53687 + (1 << 7)) >> 13) - 128 * 16384 / 500;
    const int bdiv = ((B * 41943 + (1 << 4)) >> 9) - 128 * 16384 / 200 + 1;
    const int x = ab_to_xz(ify + adiv), z = ab_to_xz(ify - bdiv);
    int ro = descale(s.c_inv[0] * x + s.c_inv[1] * y + s.c_inv[2] * z, 14);
    int go = descale(s.c_inv[3] * x + s.c_inv[4] * y + s.c_inv[5] * z, 14);
    int bo = descale(s.c_inv[6] * x + s.c_inv[7] * y + s.c_inv[8] * z, 14);
    ro = ro < 0 ? 0 : (ro > 4095 ? 4095 : ro);
    go = go < 0 ? 0 : (go > 4095 ? 4095 : go);
    bo = bo < 0 ? 0 : (bo > 4095 ? 4095 : bo);
    r = s.inv_gamma[ro];
    g = s.inv_gamma[go];
    b = s.inv_gamma[bo];
}

__global__ __launch_bounds__(LT) void lab_hist_kernel(const uint8_t* __restrict__ img, long hw, const tia_lab_tables* __restrict__ tab,
                                                       uint32_t* __restrict__ hist) {
    __shared__ LabLds s;
    __shared__ unsigned h[3][256];
    load_tables(s, tab);
    for (int i = threadIdx.x; i < 768; i += LT) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = img + (size_t)blockIdx.y * hw * 3;
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < hw; i += (long)gridDim.x * LT) {
        int L, A, B;
        rgb2lab(s, p[3 * i], p[3 * i + 1], p[3 * i + 2], L, A, B);
        atomicAdd(&h[0][L], 1u);
        atomicAdd(&h[1][A], 1u);
        atomicAdd(&h[2][B], 1u);
    }
    __syncthreads();
    uint32_t* out = hist + (size_t)blockIdx.y * 768;
    for (int i = threadIdx.x; i < 768; i += LT) {
        const unsigned v = (&h[0][0])[i];
        if (v) atomicAdd(&out[i], v);
    }
}

__global__ __launch_bounds__(LT) void reinhard_apply_kernel(const uint8_t* __restrict__ img, long hw, const tia_lab_tables* __restrict__ tab,
                                                             const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
    __shared__ LabLds s;
    __shared__ uint8_t l[3][256];
    load_tables(s, tab);
    for (int i = threadIdx.x; i < 768; i += LT) (&l[0][0])[i] = lut[(size_t)blockIdx.y * 768 + i];
    __syncthreads();
    const uint8_t* p = img + (size_t)blockIdx.y * hw * 3;
    uint8_t* o = out + (size_t)blockIdx.y * hw * 3;
    auto px = [&](uint32_t r, uint32_t g, uint32_t b, uint32_t& ro, uint32_t& go, uint32_t& bo) {
        int L, A, B, rr, gg, bb;
        rgb2lab(s, r, g, b, L, A, B);
        lab2rgb(s, l[0][L], l[1][A], l[2][B], rr, gg, bb);
        ro = (uint32_t)rr;
        go = (uint32_t)gg;
        bo = (uint32_t)bb;
    };
    if ((hw & 3) == 0) {
        const long ng = hw >> 2;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        uint32_t* w = reinterpret_cast<uint32_t*>(o);
        for (long g = (long)blockIdx.x * LT + threadIdx.x; g < ng; g += (long)gridDim.x * LT) {
            const uint32_t a = q[g * 3], b = q[g * 3 + 1], c = q[g * 3 + 2];
            uint32_t r0, g0, b0, r1, g1, b1, r2, g2, b2, r3, g3, b3;
            px(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, r0, g0, b0);
            px(a >> 24, b & 255u, (b >> 8) & 255u, r1, g1, b1);
            px((b >> 16) & 255u, b >> 24, c & 255u, r2, g2, b2);
            px((c >> 8) & 255u, (c >> 16) & 255u, c >> 24, r3, g3, b3);
            w[g * 3] = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
            w[g * 3 + 1] = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
            w[g * 3 + 2] = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
        }
    } else {
        for (long i = (long)blockIdx.x * LT + threadIdx.x; i < hw; i += (long)gridDim.x * LT) {
            uint32_t r, g, b;
            px(p[3 * i], p[3 * i + 1], p[3 * i + 2], r, g, b);
            o[3 * i] = (uint8_t)r;
            o[3 * i + 1] = (uint8_t)g;
            o[3 * i + 2] = (uint8_t)b;
        }
    }
}

// Per-image statistics and look-up tables of ReinhardNormalizer.transform (stainnorm.py:277-292, 336-339) from
// the Lab byte histograms: cv2.meanStdDev of the float32 channels == moments of 256 weighted values (f64,
// summed in NumPy's pairwise order for 256 elements so host and device agree to the bit), then the float32
// chain ((chan - mean) * (t_std / std) + t_mean), back to Lab bytes (x2.55 or +128, clip, truncate).
struct ReinhardTarget {
    double mean[3];
    double stdv[3];
};
__device__ __forceinline__ double np_pairwise_256(const double* __restrict__ a) {
    double total = 0.0;
    for (int half = 0; half < 2; ++half) {
        const double* p = a + half * 128;
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = p[j];
        for (int i = 8; i < 128; i += 8)
            for (int j = 0; j < 8; ++j) r[j] += p[i + j];
        const double part = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        total = half == 0 ? part : total + part;
    }
    return total;
}
__global__ __launch_bounds__(256) void reinhard_lut_kernel(const uint32_t* __restrict__ hist, const float* __restrict__ chan_vals,
                                                           const ReinhardTarget tgt, uint8_t* __restrict__ lut,
                                                           double* __restrict__ meanstd, int* __restrict__ flags) {
    __shared__ double prod[3][2][256];
    __shared__ double stat[3][2];  // mean, std
    const uint32_t* hs = hist + (size_t)blockIdx.x * 768;
    const int v = threadIdx.x;
    for (int c = 0; c < 3; ++c) {
        const double val = (double)chan_vals[c * 256 + v];
        const double hv = (double)hs[c * 256 + v] * val;
        prod[c][0][v] = hv;
        prod[c][1][v] = hv * val;
    }
    __syncthreads();
    if (v < 3) {
        long long cnt = 0;
        for (int i = 0; i < 256; ++i) cnt += hs[v * 256 + i];
        const double n = (double)cnt;
        const double mean = np_pairwise_256(prod[v][0]) / n;
        double var = np_pairwise_256(prod[v][1]) / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double sd = sqrt(var);
        stat[v][0] = mean;
        stat[v][1] = sd;
        if (meanstd) {
            meanstd[(size_t)blockIdx.x * 6 + v] = mean;
            meanstd[(size_t)blockIdx.x * 6 + 3 + v] = sd;
        }
        if (flags && sd == 0.0) atomicOr(&flags[blockIdx.x], 1);
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
        const float mean32 = (float)stat[c][0];
        const float ratio32 = (float)(tgt.stdv[c] / stat[c][1]);
        const float tmean32 = (float)tgt.mean[c];
        float norm = (chan_vals[c * 256 + v] - mean32) * ratio32 + tmean32;
        norm = c == 0 ? norm * 2.55f : norm + 128.0f;
        norm = norm < 0.0f ? 0.0f : (norm > 255.0f ? 255.0f : norm);  // NaN (std == 0) falls through; flagged above
        lut[(size_t)blockIdx.x * 768 + c * 256 + v] = (uint8_t)(int)norm;
    }
}

__global__ __launch_bounds__(LT) void lab_convert_kernel(const uint8_t* __restrict__ src, long npix, const tia_lab_tables* __restrict__ tab,
                                                          int dir, uint8_t* __restrict__ dst) {
    __shared__ LabLds s;
    load_tables(s, tab);
    __syncthreads();
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < npix; i += (long)gridDim.x * LT) {
        int a, b, c;
        if (dir == 0) rgb2lab(s, src[3 * i], src[3 * i + 1], src[3 * i + 2], a, b, c);
        else lab2rgb(s, src[3 * i], src[3 * i + 1], src[3 * i + 2], a, b, c);
        dst[3 * i] = (uint8_t)a;
        dst[3 * i + 1] = (uint8_t)b;
        dst[3 * i + 2] = (uint8_t)c;
    }
}

static inline unsigned lab_blocks(long work, long n) {
    long per = (long)LT * 16;
    long maxb = (work + per - 1) / per;
    long want = (2048 + n - 1) / n;
    long b = want < maxb ? want : maxb;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace tia

using namespace tia;

extern "C" int tia_lab_hist_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                                uint32_t* d_hist, void* stream) {
    if (!d_img || !d_tables || !d_hist || n <= 0 || h <= 0 || w <= 0 || n > 65535) return TIA_EINVAL;
    const long hw = (long)h * w;
    hipLaunchKernelGGL(lab_hist_kernel, dim3(lab_blocks(hw, n), (unsigned)n), dim3(LT), 0, (hipStream_t)stream, d_img, hw, d_tables,
                       d_hist);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_reinhard_apply_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                                      const uint8_t* d_lut, uint8_t* d_out, void* stream) {
    if (!d_img || !d_tables || !d_lut || !d_out || n <= 0 || h <= 0 || w <= 0 || n > 65535) return TIA_EINVAL;
    const long hw = (long)h * w;
    hipLaunchKernelGGL(reinhard_apply_kernel, dim3(lab_blocks(hw >> 2 ? hw >> 2 : 1, n), (unsigned)n), dim3(LT), 0,
                       (hipStream_t)stream, d_img, hw, d_tables, d_lut, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_reinhard_luts(const uint32_t* d_hist, int64_t n, const float* d_chan_vals, const double* target_means,
                                 const double* target_stds, uint8_t* d_lut, double* d_meanstd, int32_t* d_flags,
                                 void* stream) {
    if (!d_hist || !d_chan_vals || !target_means || !target_stds || !d_lut || n <= 0 || n > 2147483647LL) return TIA_EINVAL;
    ReinhardTarget t;
    for (int c = 0; c < 3; ++c) {
        t.mean[c] = target_means[c];
        t.stdv[c] = target_stds[c];
    }
    hipLaunchKernelGGL(reinhard_lut_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, d_hist, d_chan_vals, t, d_lut,
                       d_meanstd, d_flags);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_lab_convert_u8(const uint8_t* d_src, int64_t npix, const tia_lab_tables* d_tables, int32_t dir,
                                   uint8_t* d_dst, void* stream) {
    if (!d_src || !d_tables || !d_dst || npix <= 0 || (dir != 0 && dir != 1)) return TIA_EINVAL;
    long nb = (npix + LT * 8 - 1) / (LT * 8);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(lab_convert_kernel, dim3((unsigned)(nb < 1 ? 1 : nb)), dim3(LT), 0, (hipStream_t)stream, d_src, (long)npix,
                       d_tables, dir, d_dst);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
