// HoVer-Net post-processing on gfx950 (reference: models/architecture/hovernet.py:502-748).
// Dense stages are one-thread-per-pixel kernels over [n,h,w] planes; the watershed is a
// priority flood run independently per connected mask blob (the global skimage heap restricted
// to one blob pops in the same order as a private heap, because a blob's entries are only ever
// inserted by pops of that blob), one lane per blob with its heap segment in global memory.
#include <cstdlib>

#include "common.hpp"

#pragma clang fp contract(off)  // OpenCV/NumPy evaluate these expressions without contraction

namespace tia {

constexpr int HT = 256;

static inline unsigned hblocks(long n, int per = HT, long cap = 4096) {
    long b = (n + per - 1) / per;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - i : i;
}

// the same for -n < i < 2 n - 1 (one reflection at most): no integer division
__device__ __forceinline__ int reflect101_near(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(256) void np_threshold_kernel(const float* __restrict__ np_map, long n, uint8_t* __restrict__ mask) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) mask[i] = np_map[i] >= 0.5f ? 1 : 0;
}

__global__ __launch_bounds__(256) void blob_indicator_kernel(const int* __restrict__ lab, long n, int32_t* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = lab[i] > 0 ? 1 : 0;
}

// ---- per-plane min / max ---------------------------------------------------------------------------------
// out[plane*2] = min, out[plane*2+1] = max (as f64) of two maps per launch.  One workgroup per plane,
// coalesced sweeps with four loads in flight; NaNs are never selected (v < mn / v > mx comparisons).
struct MinMax2 {
    double v[4];  // min0, max0, min1, max1
    __device__ void init() {
        v[0] = v[2] = 1.0 / 0.0;
        v[1] = v[3] = -1.0 / 0.0;
    }
    __device__ void add(double a, double b) {
        v[0] = a < v[0] ? a : v[0];
        v[1] = a > v[1] ? a : v[1];
        v[2] = b < v[2] ? b : v[2];
        v[3] = b > v[3] ? b : v[3];
    }
    __device__ void merge(const double o[4]) {
        v[0] = o[0] < v[0] ? o[0] : v[0];
        v[1] = o[1] > v[1] ? o[1] : v[1];
        v[2] = o[2] < v[2] ? o[2] : v[2];
        v[3] = o[3] > v[3] ? o[3] : v[3];
    }
};
__device__ __forceinline__ void minmax2_finish(MinMax2 m, double* __restrict__ out0, double* __restrict__ out1) {
    __shared__ double red[16][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __shfl_down(m.v[k], o, 64);
        m.merge(t);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave_id()][k] = m.v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m.merge(red[i]);
        out0[blockIdx.x * 2] = m.v[0];
        out0[blockIdx.x * 2 + 1] = m.v[1];
        out1[blockIdx.x * 2] = m.v[2];
        out1[blockIdx.x * 2 + 1] = m.v[3];
    }
}
// both channels of the interleaved (h, v) map
__global__ __launch_bounds__(1024) void minmax_hv_kernel(const float* __restrict__ hv, long hw, double* __restrict__ out_h,
                                                          double* __restrict__ out_v) {
    const float2* s = reinterpret_cast<const float2*>(hv + (size_t)blockIdx.x * hw * 2);
    MinMax2 m;
    m.init();
    long i = threadIdx.x;
    for (; i + 3072 < hw; i += 4096) {
        const float2 a = s[i], b = s[i + 1024], c = s[i + 2048], d = s[i + 3072];
        m.add((double)a.x, (double)a.y);
        m.add((double)b.x, (double)b.y);
        m.add((double)c.x, (double)c.y);
        m.add((double)d.x, (double)d.y);
    }
    for (; i < hw; i += 1024) {
        const float2 a = s[i];
        m.add((double)a.x, (double)a.y);
    }
    minmax2_finish(m, out_h, out_v);
}
// two f64 planes
__global__ __launch_bounds__(1024) void minmax_pair_kernel(const double* __restrict__ pa, const double* __restrict__ pb, long hw,
                                                            double* __restrict__ out_a, double* __restrict__ out_b) {
    const double* a = pa + (size_t)blockIdx.x * hw;
    const double* b = pb + (size_t)blockIdx.x * hw;
    MinMax2 m;
    m.init();
    long i = threadIdx.x;
    for (; i + 1024 < hw; i += 2048) {
        const double a0 = a[i], b0 = b[i], a1 = a[i + 1024], b1 = b[i + 1024];
        m.add(a0, b0);
        m.add(a1, b1);
    }
    for (; i < hw; i += 1024) m.add(a[i], b[i]);
    minmax2_finish(m, out_a, out_b);
}

// cv2.normalize(NORM_MINMAX, 0..1): scale = 1/(max-min) (0 if the range is < DBL_EPSILON), shift = -min*scale
__device__ __forceinline__ void norm_params(const double* __restrict__ mm, double& scale, double& shift) {
    const double rng = mm[1] - mm[0];
    scale = rng > 2.220446049250313e-16 ? 1.0 / rng : 0.0;
    shift = 0.0 - mm[0] * scale;
}

// Filter taps travel as a kernel argument (scalar loads; exact small integers in f64).
struct SobelTaps {
    double v[32];
};

// ---- Sobel (separable, f64 accumulate) -----------------------------------------------------------------------
// Row pass over the min-max-normalised f32 map (normalisation in f32 arithmetic, as convertTo does for
// CV_32F sources): buf = sum_k kx[k] * S[x + k - anchor], taps in ascending order.
__global__ __launch_bounds__(HT) void sobel_row_kernel(const float* __restrict__ hv, int channel, int h, int w,
                                                        const double* __restrict__ mm, const SobelTaps kx,
                                                        int ksize, double* __restrict__ buf) {
    const long hw = (long)h * w;
    const float* src = hv + (size_t)blockIdx.y * hw * 2 + channel;
    double* dst = buf + (size_t)blockIdx.y * hw;
    double scale, shift;
    norm_params(mm + blockIdx.y * 2, scale, shift);
    const float a = (float)scale, b = (float)shift;
    const int anchor = ksize / 2;
    for (long i = (long)blockIdx.x * HT + threadIdx.x; i < hw; i += (long)gridDim.x * HT) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        const float* row = src + (size_t)y * w * 2;
        double acc = 0.0;
        for (int k = 0; k < ksize; ++k) {
            const int xx = reflect101(x + k - anchor, w);
            const float v = row[(size_t)xx * 2] * a + b;
            const double t = kx.v[k] * (double)v;
            acc = (k == 0) ? t : acc + t;
        }
        dst[i] = acc;
    }
}

// Column pass, OpenCV SymmColumnFilter: centre tap, then ky[c+k]*(S[+k] +/- S[-k]).
__global__ __launch_bounds__(HT) void sobel_col_kernel(const double* __restrict__ buf, int h, int w,
                                                        const SobelTaps ky, int ksize, int symmetric,
                                                        double* __restrict__ out) {
    const long hw = (long)h * w;
    const double* src = buf + (size_t)blockIdx.y * hw;
    double* dst = out + (size_t)blockIdx.y * hw;
    const int c = ksize / 2;
    for (long i = (long)blockIdx.x * HT + threadIdx.x; i < hw; i += (long)gridDim.x * HT) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        double acc = symmetric ? (ky.v[c] * src[i] + 0.0) : 0.0;
        for (int k = 1; k <= c; ++k) {
            const double up = src[(long)reflect101(y + k, h) * w + x];
            const double dn = src[(long)reflect101(y - k, h) * w + x];
            acc = acc + ky.v[c + k] * (symmetric ? (up + dn) : (up - dn));
        }
        dst[i] = acc;
    }
}

// ---- energy landscape (hovernet.py:571-603) ---------------------------------------------------------------------
__global__ __launch_bounds__(HT) void energy_kernel(const double* __restrict__ sob_h, const double* __restrict__ sob_v,
                                                     const double* __restrict__ mm_h, const double* __restrict__ mm_v,
                                                     const int* __restrict__ blob, long hw, double* __restrict__ dist0,
                                                     uint8_t* __restrict__ marker0) {
    const size_t off = (size_t)blockIdx.y * hw;
    double sh_s, sh_b, sv_s, sv_b;
    norm_params(mm_h + blockIdx.y * 2, sh_s, sh_b);
    norm_params(mm_v + blockIdx.y * 2, sv_s, sv_b);
    for (long i = (long)blockIdx.x * HT + threadIdx.x; i < hw; i += (long)gridDim.x * HT) {
        const int blb = blob[off + i] > 0 ? 1 : 0;
        const float nh = (float)(sob_h[off + i] * sh_s + sh_b);  // f64 -> f32 convertTo
        const float nv = (float)(sob_v[off + i] * sv_s + sv_b);
        const float sh = 1.0f - nh, sv = 1.0f - nv;              // 1 - float32
        const float ov32 = sh > sv ? sh : sv;                    // np.maximum (float32)
        double overall = (double)ov32 - (double)(1 - blb);       // float32 - int32 -> float64
        if (overall < 0.0) overall = 0.0;
        dist0[off + i] = (1.0 - overall) * (double)blb;
        int mk = blb - (overall >= 0.4 ? 1 : 0);
        marker0[off + i] = mk < 0 ? 0 : (uint8_t)mk;
    }
}

// dist = -GaussianBlur(dist0, (3,3), 0): [1/4,1/2,1/4] separable, S0*k0 + (S-1 + S+1)*k1, REFLECT_101
__global__ __launch_bounds__(HT) void gauss3_neg_kernel(const double* __restrict__ src, int h, int w,
                                                         double* __restrict__ dst) {
    const long hw = (long)h * w;
    const double* s = src + (size_t)blockIdx.y * hw;
    double* d = dst + (size_t)blockIdx.y * hw;
    for (long i = (long)blockIdx.x * HT + threadIdx.x; i < hw; i += (long)gridDim.x * HT) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
        double r[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double* row = s + (long)reflect101(y + j - 1, h) * w;
            r[j] = row[x] * 0.5 + (row[xm] + row[xp]) * 0.25;
        }
        d[i] = -(r[1] * 0.5 + (r[0] + r[2]) * 0.25);
    }
}

// ---- small planes: min / max, both Sobel filters, their min / max and the energy landscape in ONE launch ------------------
// One 1024-thread workgroup per plane.  The separable filter runs in bands of output rows: the row pass of the band's rows
// (+ the ksize / 2 rows above and below) goes to an LDS buffer, the column pass reads it from there (the multi-launch form
// sends the whole f64 row-pass plane through HBM).  Per-pixel arithmetic = sobel_row_kernel / sobel_col_kernel / energy_kernel
// above, operation for operation; min / max are order-independent.  After the last band the workgroup knows the range of
// both gradient planes and turns them into the energy landscape + marker seed (re-reading the planes it just wrote).
__device__ __forceinline__ void block_minmax2(MinMax2 m, double (*red)[4], double* out4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __shfl_down(m.v[k], o, 64);
        m.merge(t);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave_id()][k] = m.v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m.merge(red[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) out4[k] = m.v[k];
    }
    __syncthreads();
}

// KS: the filter size at compile time (21: HoVer-Net, 11: HoVerNet+) -- the tap loops are then fully unrolled and the taps are
// scalar registers; with a run-time index every tap is a scalar load from the kernel arguments.  KS = 0: any size.
template <int KS>
__global__ __launch_bounds__(1024) void sobel_energy_tile_kernel(const float* __restrict__ hv, int h, int w, const SobelTaps kd,
                                                                  const SobelTaps ks, int ksize_rt, int band, int n,
                                                                  const int* __restrict__ blob, double* __restrict__ sob_h,
                                                                  double* __restrict__ sob_v, double* __restrict__ dist0,
                                                                  uint8_t* __restrict__ marker0, double* __restrict__ mm,
                                                                  double* __restrict__ dist) {
    extern __shared__ double rb[];  // row-pass band: (band + ksize - 1) rows x w doubles, then the same rows of the normalised input
    __shared__ double red[16][4];
    __shared__ double s_in[4], s_out[4];  // min / max: (h, v) of the input maps, then of the two gradient planes
    const int ksize = KS ? KS : ksize_rt;
    const int plane = blockIdx.x, tid = threadIdx.x, hw = h * w, anchor = ksize / 2;
    float* inb = reinterpret_cast<float*>(rb + (size_t)(band + ksize - 1) * w);
    const bool near = h > anchor && w > anchor;  // border taps reflect once at most (otherwise: the general form)
    auto refl = [&](int i, int nn) { return near ? reflect101_near(i, nn) : reflect101(i, nn); };
    const size_t off = (size_t)plane * hw;
    const float2* in2 = reinterpret_cast<const float2*>(hv + off * 2);
    {
        MinMax2 m;
        m.init();
#pragma unroll 8  // one workgroup per CU: the loads of several rounds must be in flight together
        for (int i = tid; i < hw; i += 1024) {
            const float2 a = in2[i];
            m.add((double)a.x, (double)a.y);
        }
        block_minmax2(m, red, s_in);
    }
    if (tid == 0) {
        mm[plane * 2] = s_in[0];
        mm[plane * 2 + 1] = s_in[1];
        mm[2 * n + plane * 2] = s_in[2];
        mm[2 * n + plane * 2 + 1] = s_in[3];
    }
    MinMax2 mo;
    mo.init();
#pragma unroll  // both copies: which tap set is the row filter must be known at compile time (a run-time choice turns every tap
                // into a scalar load from the kernel arguments, ~200 cycles that one workgroup per CU cannot hide)
    for (int ch = 0; ch < 2; ++ch) {
        // h: dx = 1 -> row taps = derivative, column taps = smoothing (symmetric); v: the other way round (antisymmetric)
        const SobelTaps& kx = ch == 0 ? kd : ks;
        const SobelTaps& ky = ch == 0 ? ks : kd;
        const bool symmetric = ch == 0;
        double scale, shift;
        norm_params(s_in + 2 * ch, scale, shift);
        const float a = (float)scale, b = (float)shift;
        const float* src = hv + off * 2 + ch;
        double* dst = (ch == 0 ? sob_h : sob_v) + off;
        double vmin = 1.0 / 0.0, vmax = -1.0 / 0.0;
        for (int y0 = 0; y0 < h; y0 += band) {
            const int y1 = min(h, y0 + band);  // output rows [y0, y1)
            const int lo = max(0, y0 - anchor), hi = min(h - 1, y1 - 1 + anchor);
            const int nrow = (hi - lo + 1) * w;
            // the band's input rows, normalised (f32 arithmetic, as convertTo does for CV_32F sources), to LDS
#pragma unroll 8
            for (int i = tid; i < nrow; i += 1024) inb[i] = src[((size_t)lo * w + i) * 2] * a + b;
            __syncthreads();
            // row pass, two pixels per lane and round (independent accumulation chains); taps in ascending order
            for (int i = tid; i < nrow; i += 2048) {
                const int i2 = i + 1024;
                const bool two = i2 < nrow;
                const int r0 = i / w, x0 = i - r0 * w, r1 = two ? i2 / w : r0, x1 = two ? i2 - r1 * w : x0;
                const float* p0 = inb + r0 * w;
                const float* p1 = inb + r1 * w;
                double acc0 = 0.0, acc1 = 0.0;
                if (x0 >= anchor && x0 + anchor < w && x1 >= anchor && x1 + anchor < w) {
                    p0 += x0 - anchor;
                    p1 += x1 - anchor;
#pragma unroll
                    for (int k = 0; k < ksize; ++k) {
                        const double t0 = kx.v[k] * (double)p0[k];
                        const double t1 = kx.v[k] * (double)p1[k];
                        acc0 = (k == 0) ? t0 : acc0 + t0;
                        acc1 = (k == 0) ? t1 : acc1 + t1;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < ksize; ++k) {
                        const double t0 = kx.v[k] * (double)p0[refl(x0 + k - anchor, w)];
                        const double t1 = kx.v[k] * (double)p1[refl(x1 + k - anchor, w)];
                        acc0 = (k == 0) ? t0 : acc0 + t0;
                        acc1 = (k == 0) ? t1 : acc1 + t1;
                    }
                }
                rb[i] = acc0;
                if (two) rb[i2] = acc1;
            }
            __syncthreads();
            const int nout = (y1 - y0) * w;
            for (int i = tid; i < nout; i += 2048) {
                const int i2 = i + 1024;
                const bool two = i2 < nout;
                const int r0 = i / w, x0 = i - r0 * w, r1 = two ? i2 / w : r0, x1 = two ? i2 - r1 * w : x0;
                const int ya = y0 + r0, yb = y0 + r1;
                const double* c0 = rb + (ya - lo) * w + x0;
                const double* c1 = rb + (yb - lo) * w + x1;
                double acc0 = symmetric ? (ky.v[anchor] * c0[0] + 0.0) : 0.0;
                double acc1 = symmetric ? (ky.v[anchor] * c1[0] + 0.0) : 0.0;
                if (ya >= anchor && ya + anchor < h && yb >= anchor && yb + anchor < h) {
#pragma unroll
                    for (int k = 1; k <= anchor; ++k) {
                        const double up0 = c0[k * w], dn0 = c0[-k * w], up1 = c1[k * w], dn1 = c1[-k * w];
                        acc0 = acc0 + ky.v[anchor + k] * (symmetric ? (up0 + dn0) : (up0 - dn0));
                        acc1 = acc1 + ky.v[anchor + k] * (symmetric ? (up1 + dn1) : (up1 - dn1));
                    }
                } else {
#pragma unroll
                    for (int k = 1; k <= anchor; ++k) {
                        const double up0 = rb[(refl(ya + k, h) - lo) * w + x0], dn0 = rb[(refl(ya - k, h) - lo) * w + x0];
                        const double up1 = rb[(refl(yb + k, h) - lo) * w + x1], dn1 = rb[(refl(yb - k, h) - lo) * w + x1];
                        acc0 = acc0 + ky.v[anchor + k] * (symmetric ? (up0 + dn0) : (up0 - dn0));
                        acc1 = acc1 + ky.v[anchor + k] * (symmetric ? (up1 + dn1) : (up1 - dn1));
                    }
                }
                dst[ya * w + x0] = acc0;
                vmin = acc0 < vmin ? acc0 : vmin;
                vmax = acc0 > vmax ? acc0 : vmax;
                if (two) {
                    dst[yb * w + x1] = acc1;
                    vmin = acc1 < vmin ? acc1 : vmin;
                    vmax = acc1 > vmax ? acc1 : vmax;
                }
            }
            __syncthreads();
        }
        mo.v[2 * ch] = vmin;
        mo.v[2 * ch + 1] = vmax;
    }
    // (workgroup scope: an agent-scope fence writes the XCD's whole L2 back -- measured 150 us here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the gradient planes are re-read below by other lanes of this workgroup
    block_minmax2(mo, red, s_out);
    if (tid == 0) {
        mm[4 * n + plane * 2] = s_out[0];
        mm[4 * n + plane * 2 + 1] = s_out[1];
        mm[6 * n + plane * 2] = s_out[2];
        mm[6 * n + plane * 2 + 1] = s_out[3];
    }
    double sh_s, sh_b, sv_s, sv_b;
    norm_params(s_out, sh_s, sh_b);
    norm_params(s_out + 2, sv_s, sv_b);
    // Plain loads are coherent here: the planes were written by THIS workgroup (fence + barrier above), this CU has not read
    // these addresses before in this launch, and its L1 starts a launch empty -- no stale line can exist.
    const double* gh = sob_h + off;
    const double* gv = sob_v + off;
#pragma unroll 4
    for (int i = tid; i < hw; i += 1024) {
        const int blb = blob[off + i] > 0 ? 1 : 0;
        const float nh = (float)(gh[i] * sh_s + sh_b);  // f64 -> f32 convertTo
        const float nv = (float)(gv[i] * sv_s + sv_b);
        const float sh = 1.0f - nh, sv = 1.0f - nv;      // 1 - float32
        const float ov32 = sh > sv ? sh : sv;            // np.maximum (float32)
        double overall = (double)ov32 - (double)(1 - blb);
        if (overall < 0.0) overall = 0.0;
        dist0[off + i] = (1.0 - overall) * (double)blb;
        int mk = blb - (overall >= 0.4 ? 1 : 0);
        marker0[off + i] = mk < 0 ? 0 : (uint8_t)mk;
    }
    if (dist == nullptr) return;
    // dist = -GaussianBlur3x3(dist0) (= gauss3_neg_kernel), reading back the plane this workgroup just wrote
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const double* sg = dist0 + off;
#pragma unroll 2
    for (int i = tid; i < hw; i += 1024) {
        const int y = i / w, x = i - y * w;
        const int xm = refl(x - 1, w), xp = refl(x + 1, w);
        double r[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double* row = sg + (long)refl(y + j - 1, h) * w;
            r[j] = row[x] * 0.5 + (row[xm] + row[xp]) * 0.25;
        }
        dist[off + i] = -(r[1] * 0.5 + (r[0] + r[2]) * 0.25);
    }
}

// ---- watershed -----------------------------------------------------------------------------------------------------
struct HeapItem {
    double value;
    int age;
    int index;
};
__device__ __forceinline__ bool heap_smaller(const HeapItem& a, const HeapItem& b) {
    if (a.value != b.value) return a.value < b.value;
    return a.age < b.age;
}
// Heap procedures = skimage's heap_general.pxi, because entries that tie on (value, age) -- the initial markers of a
// flat plateau all carry age 0 -- leave the queue in an order that depends on the heap's internal arrangement:
// push = append + move up while smaller than the parent; pop = the last item replaces the root, then at every node
// the smallest of node / left / right moves up until the node itself is the smallest (not CPython heapq's
// bubble-to-a-leaf-and-sift-back, which pops such ties in a different order: 4.62 vs 4.84 ms per 256 x 164^2 batch,
// profiles/r02a_hover_*.jsonl; tests/test_hovernet_post.py forces the ties).
// One lane owns one heap, kept in the lane's global-memory segment.  (Keeping the top levels of every
// heap in LDS was measured and lost: 5.6 ms -> 8.0 ms per 256 x 164^2 batch; the extra branches and the 63 KB
// LDS footprint cost more than the L2 round trips they save.)
// inst = where(mask, markers, 0) with mask pixels still to be flooded marked -1 (so the flood needs a single
// load per neighbour; every -1 is gone when the flood ends); blob bounding boxes
__global__ __launch_bounds__(HT) void ws_init_kernel(const int* __restrict__ blob, const int* __restrict__ marker, int h, int w,
                                                      int* __restrict__ inst, int* __restrict__ bbox) {
    const long hw = (long)h * w;
    const size_t off = (size_t)blockIdx.y * hw;
    int* bb = bbox + (size_t)blockIdx.y * (hw + 1) * 4;
    const int lane = lane_id();
    for (long base = (long)blockIdx.x * HT; base < hw; base += (long)gridDim.x * HT) {  // uniform trip count
        const long i = base + threadIdx.x;
        const int b = i < hw ? blob[off + i] : 0;
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        if (i < hw) {
            const int mk = b > 0 ? marker[off + i] : 0;
            inst[off + i] = b > 0 ? (mk > 0 ? mk : -1) : 0;
        }
        // bounding boxes: one set of atomics per horizontal run of a blob within the wave
        const int pb = __shfl_up(b, 1);
        const bool head = lane == 0 || b != pb || x == 0;
        const unsigned long long heads = __ballot(head);
        if (head && b > 0) {
            const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
            const int len = (above ? __builtin_ctzll(above) : 64) - lane;
            atomicMin(&bb[b * 4 + 0], y);
            atomicMax(&bb[b * 4 + 1], y);
            atomicMin(&bb[b * 4 + 2], x);
            atomicMax(&bb[b * 4 + 3], x + len - 1);
        }
    }
}
__global__ __launch_bounds__(HT) void bbox_reset_kernel(int* __restrict__ bbox, long n_entries) {
    for (long i = (long)blockIdx.x * HT + threadIdx.x; i < n_entries; i += (long)gridDim.x * HT) {
        bbox[i * 4 + 0] = 0x7fffffff;
        bbox[i * 4 + 1] = -1;
        bbox[i * 4 + 2] = 0x7fffffff;
        bbox[i * 4 + 3] = -1;
    }
}

// heap segment offsets: exclusive scan over labels of (area if the blob survived the size filter)
__global__ __launch_bounds__(1024) void ws_offsets_kernel(const int* __restrict__ areas, const int* __restrict__ count, long hw,
                                                           int min_keep, int* __restrict__ offs) {
    __shared__ unsigned wtot[16];
    const int* a = areas + (size_t)blockIdx.x * (hw + 1);
    int* o = offs + (size_t)blockIdx.x * (hw + 1);
    const long k = (long)count[blockIdx.x] + 1;  // labels 0..count
    const long chunk = (k + 1023) / 1024;
    const long lo = (long)threadIdx.x * chunk, hi = lo + chunk < k ? lo + chunk : k;
    unsigned c = 0;
    for (long i = lo; i < hi; ++i) c += (i > 0 && a[i] >= min_keep) ? (unsigned)a[i] : 0u;
    const unsigned incl = wave_incl_scan_u32(c);
    if (lane_id() == 63) wtot[wave_id()] = incl;
    __syncthreads();
    unsigned before = incl - c;
    for (int wv = 0; wv < wave_id(); ++wv) before += wtot[wv];
    for (long i = lo; i < hi; ++i) {
        o[i] = (int)before;
        before += (i > 0 && a[i] >= min_keep) ? (unsigned)a[i] : 0u;
    }
}

__global__ __launch_bounds__(64) void ws_flood_kernel(const int* __restrict__ blob, const double* __restrict__ dist,
                                                       const int* __restrict__ areas, const int* __restrict__ offs,
                                                       const int* __restrict__ count, const int* __restrict__ bbox, int h, int w,
                                                       int min_keep, HeapItem* __restrict__ heaps, int* __restrict__ inst) {
    const long hw = (long)h * w;
    // Plane index fastest: workgroups go round-robin to the 8 XCDs in launch order, and in a batch of patches
    // only the first label chunk of every plane has work -- with the chunk index fastest all of it lands on one XCD.
    const int plane = blockIdx.x;
    for (int label = blockIdx.y * 64 + threadIdx.x + 1; label <= count[plane]; label += gridDim.y * 64) {
    const int area = areas[(size_t)plane * (hw + 1) + label];
    if (area < min_keep) continue;
    const size_t off = (size_t)plane * hw;
    const int* bl = blob + off;
    const double* ds = dist + off;
    int* out = inst + off;
    HeapItem* glob = heaps + off + offs[(size_t)plane * (hw + 1) + label];
    // skimage heap_general.pxi: push = append + sift towards the root
    auto heap_push = [&](int& items, const HeapItem& e) {
        int pos = items++;
        while (pos > 0) {
            const int parent = (pos - 1) >> 1;
            const HeapItem p = glob[parent];
            if (!heap_smaller(e, p)) break;
            glob[pos] = p;
            pos = parent;
        }
        glob[pos] = e;
    };
    // pop (heap_general.pxi): root out, last item to the root, sift down
    auto heap_pop = [&](int& items) {
        const HeapItem top = glob[0];
        --items;
        if (items == 0) return top;
        const HeapItem last = glob[items];
        {
            int at = 0;
            while (2 * at + 1 < items) {
                const int left = 2 * at + 1, right = left + 1;
                const HeapItem lc = glob[left];
                int best_i = at;
                HeapItem best = last;  // the item conceptually sitting at `at`
                if (heap_smaller(lc, best)) {
                    best = lc;
                    best_i = left;
                }
                if (right < items) {
                    const HeapItem rc = glob[right];
                    if (heap_smaller(rc, best)) {
                        best = rc;
                        best_i = right;
                    }
                }
                if (best_i == at) break;
                glob[at] = best;
                at = best_i;
            }
            glob[at] = last;
            return top;
        }
    };
    const int* bb = bbox + ((size_t)plane * (hw + 1) + label) * 4;
    const int y0 = bb[0], y1 = bb[1], x0 = bb[2], x1 = bb[3];
    int items = 0, age = 0;
    // initial queue: the blob's marker pixels in raster order, all with age 0 (as skimage does)
    for (int y = y0; y <= y1; ++y) {
        for (int x = x0; x <= x1; ++x) {
            const long i = (long)y * w + x;
            if (bl[i] == label && out[i] > 0) {
                HeapItem e;
                e.value = ds[i];
                e.age = 0;
                e.index = (int)i;
                heap_push(items, e);
            }
        }
    }
    if (items == 0) {  // a blob without markers stays background
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const long i = (long)y * w + x;
                if (bl[i] == label) out[i] = 0;
            }
        continue;
    }
    while (items > 0) {
        const HeapItem e = heap_pop(items);
        const int lab = out[e.index];
        const int y = e.index / w, x = e.index - y * w;
        // neighbour order of skimage's raveled offsets for connectivity 1: up, left, right, down.
        // The four neighbours are distinct pixels only this lane can change: fetch state and value of all
        // of them before acting, so one memory latency covers the whole step.
        long ni[4];
        int state[4];
        double dv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = y + (j == 0 ? -1 : (j == 3 ? 1 : 0));
            const int xx = x + (j == 1 ? -1 : (j == 2 ? 1 : 0));
            const bool inb = yy >= 0 && yy < h && xx >= 0 && xx < w;
            ni[j] = inb ? (long)yy * w + xx : (long)e.index;
            state[j] = inb ? out[ni[j]] : 0;
            dv[j] = ds[ni[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (state[j] != -1) continue;  // outside the mask, or labelled already
            ++age;
            out[ni[j]] = lab;
            HeapItem ne;
            ne.value = dv[j];
            ne.age = age;
            ne.index = (int)ni[j];
            heap_push(items, ne);
        }
    }
    }  // labels of this lane
}

// ---- watershed, one WAVE per blob ----------------------------------------------------------------------------------
// Same priority flood, but a whole wave owns one blob: lane 0 runs the (inherently sequential) algorithm and does
// every global access -- so ordering is plain single-thread program order -- while the 64 lanes together are the
// register file of the heap's top six levels (node i lives in lane i; v_readlane / a one-lane write instead of a
// memory round trip per sift step).  Control flow is wave-uniform: everything lane 0 loads is broadcast with
// v_readfirstlane.  Nodes >= 64 spill to the blob's global segment.  Many more waves are resident than with one
// lane per blob, which is what hides the remaining neighbour-load latency.
__device__ __forceinline__ int bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double bcast0(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int lane_get(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_get(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__global__ __launch_bounds__(64) void ws_flood_wave_kernel(const int* __restrict__ blob, const double* __restrict__ dist,
                                                            const int* __restrict__ areas, const int* __restrict__ offs,
                                                            const int* __restrict__ count, const int* __restrict__ bbox, int h,
                                                            int w, int min_keep, HeapItem* __restrict__ heaps,
                                                            const int* __restrict__ done, int* __restrict__ inst) {
    const long hw = (long)h * w;
    const int plane = blockIdx.x;  // plane fastest: consecutive workgroups (= consecutive XCDs) take different planes
    const int lane = threadIdx.x;
    const int n_labels = count[plane];
    for (int label = blockIdx.y + 1; label <= n_labels; label += gridDim.y) {
        const int area = areas[(size_t)plane * (hw + 1) + label];
        if (area < min_keep) continue;
        if (done && done[(size_t)plane * (hw + 1) + label]) continue;  // settled by ws_relax_wave_kernel
        const size_t off = (size_t)plane * hw;
        const int* bl = blob + off;
        const double* ds = dist + off;
        int* out = inst + off;
        HeapItem* glob = heaps + off + offs[(size_t)plane * (hw + 1) + label];
        double hv = 0.0;  // this lane's heap node
        int ha = 0, hx = 0;
        auto hget = [&](int i) -> HeapItem {
            HeapItem e;
            if (i < 64) {
                e.value = lane_get(hv, i);
                e.age = lane_get(ha, i);
                e.index = lane_get(hx, i);
            } else {
                double v = 0.0;
                int a = 0, x = 0;
                if (lane == 0) {
                    const HeapItem g = glob[i];
                    v = g.value;
                    a = g.age;
                    x = g.index;
                }
                e.value = bcast0(v);
                e.age = bcast0(a);
                e.index = bcast0(x);
            }
            return e;
        };
        auto hput = [&](int i, const HeapItem& e) {
            if (i < 64) {
                if (lane == i) {
                    hv = e.value;
                    ha = e.age;
                    hx = e.index;
                }
            } else if (lane == 0) {
                glob[i] = e;
            }
        };
        auto heap_push = [&](int& items, const HeapItem& e) {  // skimage heap_general.pxi: append + sift up
            int pos = items++;
            while (pos > 0) {
                const int parent = (pos - 1) >> 1;
                const HeapItem p = hget(parent);
                if (!heap_smaller(e, p)) break;
                hput(pos, p);
                pos = parent;
            }
            hput(pos, e);
        };
        auto heap_pop = [&](int& items) {  // heap_general.pxi: root out, last item to the root, sift down
            const HeapItem top = hget(0);
            --items;
            if (items == 0) return top;
            const HeapItem last = hget(items);
            {
                int at = 0;
                while (2 * at + 1 < items) {
                    const int left = 2 * at + 1, right = left + 1;
                    const HeapItem lc = hget(left);
                    int best_i = at;
                    HeapItem best = last;  // the item conceptually sitting at `at`
                    if (heap_smaller(lc, best)) {
                        best = lc;
                        best_i = left;
                    }
                    if (right < items) {
                        const HeapItem rc = hget(right);
                        if (heap_smaller(rc, best)) {
                            best = rc;
                            best_i = right;
                        }
                    }
                    if (best_i == at) break;
                    hput(at, best);
                    at = best_i;
                }
                hput(at, last);
                return top;
            }
        };
        const int* bb = bbox + ((size_t)plane * (hw + 1) + label) * 4;
        const int y0 = bb[0], y1 = bb[1], x0 = bb[2], x1 = bb[3];
        int items = 0, age = 0;
        // initial queue: marker pixels of the blob in raster order, age 0.  64 pixels of a row are examined at once;
        // the hits are pushed in lane (= raster) order.
        for (int y = y0; y <= y1; ++y) {
            for (int xb = x0; xb <= x1; xb += 64) {
                const int x = xb + lane;
                const long i = (long)y * w + x;
                const bool hit = x <= x1 && bl[i] == label && out[i] > 0;
                const double val = hit ? ds[i] : 0.0;
                unsigned long long m = __ballot(hit);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    HeapItem e;
                    e.value = lane_get(val, src);
                    e.age = 0;
                    e.index = (int)((long)y * w + xb + src);
                    heap_push(items, e);
                }
            }
        }
        if (items == 0) {  // a blob without markers stays background
            for (int y = y0; y <= y1; ++y)
                for (int x = x0 + lane; x <= x1; x += 64) {
                    const long i = (long)y * w + x;
                    if (bl[i] == label) out[i] = 0;
                }
            continue;
        }
        while (items > 0) {
            const HeapItem e = heap_pop(items);
            const int y = e.index / w, x = e.index - y * w;
            int lab = 0, st[4] = {0, 0, 0, 0};
            double dv[4] = {0.0, 0.0, 0.0, 0.0};
            long ni[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // up, left, right, down (skimage's raveled offsets)
                const int yy = y + (j == 0 ? -1 : (j == 3 ? 1 : 0));
                const int xx = x + (j == 1 ? -1 : (j == 2 ? 1 : 0));
                const bool inb = yy >= 0 && yy < h && xx >= 0 && xx < w;
                ni[j] = inb ? (long)yy * w + xx : (long)e.index;
            }
            if (lane == 0) {  // all nine loads in flight together
                lab = out[e.index];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st[j] = ni[j] != (long)e.index ? out[ni[j]] : 0;
                    dv[j] = ds[ni[j]];
                }
            }
            lab = bcast0(lab);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (bcast0(st[j]) != -1) continue;  // outside the mask, or labelled already
                ++age;
                if (lane == 0) out[ni[j]] = lab;
                HeapItem ne;
                ne.value = bcast0(dv[j]);
                ne.age = age;
                ne.index = (int)ni[j];
                heap_push(items, ne);
            }
        }
    }
}

// ---- watershed by minimax relaxation, one wave per blob, all lanes working ---------------------------------------------
// The priority flood labels an unlabelled pixel n at the pop of its FIRST-popped neighbour, and pops follow (value, age).
// Let L(p) = the highest value on p's claim chain back to its marker pixel (L = own value for marker pixels); it equals
// the minimax path value from the markers, so it does not depend on how ties are broken.  A neighbour with strictly
// smaller L pops strictly earlier (everything on its chain lies below the other chain's highest pixel, which therefore
// cannot reach the front of the queue first), hence n's claimer is one of its minimum-L neighbours.  So: relax
// (L, D, label) over the unlabelled pixels until nothing changes -- each takes its neighbour with the smallest (L, D),
// L(n) = max(value(n), L), D = hops since the chain's highest pixel (strictly growing along a chain, which keeps the
// parent pointers acyclic inside equal-L regions) -- and then CHECK every unlabelled pixel: if all its minimum-L
// neighbours carry its own label, induction over the true pop order shows the flood produces exactly this labelling,
// whatever the tie-breaks.  If some pixel has two minimum-L neighbours with different labels (an exact tie that the
// heap's age / arrangement would decide), the blob is left untouched and flagged for the sequential heap flood below.
// tests/test_flood_relaxation_model.py checks the argument on a NumPy model against the oracle; on HoVer-Net maps no blob
// needs the fallback.  State lives in two plane-sized arrays (L as f64 bits; D and label packed), read and written with
// agent-scope relaxed atomics so that a lane sees what another lane of its wave stored in an earlier pass.
__device__ __forceinline__ unsigned long long relax_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void relax_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct RelaxNb {
    double l;
    int d, lab;
};
// state of neighbour q of a pixel as the relaxation sees it: marker pixels are sources (L = own value, D = 0)
__device__ __forceinline__ RelaxNb relax_neighbour(bool inb, long q, const int* __restrict__ out, const double* __restrict__ ds,
                                                   const unsigned long long* stl, const unsigned long long* stdl) {
    const int o = out[q];
    const double dq = ds[q];
    const unsigned long long lq = relax_load(stl + q), dlq = relax_load(stdl + q);
    RelaxNb r;
    const bool marker = inb && o > 0, open = inb && o < 0;
    r.l = marker ? dq : (open ? __longlong_as_double((long long)lq) : __builtin_huge_val());
    r.d = marker ? 0 : (int)(unsigned)(dlq >> 32);
    r.lab = marker ? o : (int)(unsigned)dlq;
    return r;
}
// NT = 64: one wave per blob (blobs of up to RELAX_BIG pixels); NT = 1024: one workgroup per bigger blob.  Sweeps
// alternate between raster and reverse raster order; a blob whose relaxation has cost more passes than the heap flood
// would have (long thin chains) is abandoned to the heap as well.
constexpr int RELAX_BIG = 4096;
template <int NT>
__global__ __launch_bounds__(NT) void ws_relax_kernel(const int* __restrict__ blob, const double* __restrict__ dist,
                                                       const int* __restrict__ areas, const int* __restrict__ offs,
                                                       const int* __restrict__ count, const int* __restrict__ bbox, int h, int w,
                                                       int min_keep, HeapItem* __restrict__ heaps,
                                                       unsigned long long* __restrict__ st_l,
                                                       unsigned long long* __restrict__ st_dl, int* __restrict__ done,
                                                       int* __restrict__ inst) {
    constexpr bool BLOCK = NT > 64;
    __shared__ int sh_count, sh_marker;
    const long hw = (long)h * w;
    const int plane = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_labels = count[plane];
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long inf_bits = (unsigned long long)__double_as_longlong(__builtin_huge_val());
    // "does any thread of the group say yes" -- a ballot for one wave, a barrier reduction for a workgroup
    auto any_of = [&](bool v) -> bool {
        if constexpr (BLOCK) return __syncthreads_or(v ? 1 : 0) != 0;
        else return __ballot(v) != 0ull;
    };
    for (int label = blockIdx.y + 1; label <= n_labels; label += gridDim.y) {
        const size_t slot = (size_t)plane * (hw + 1) + label;
        const int area = areas[slot];
        if (area < min_keep || (area > RELAX_BIG) != BLOCK) continue;
        const size_t off = (size_t)plane * hw;
        const int* bl = blob + off;
        const double* ds = dist + off;
        int* out = inst + off;
        unsigned long long* stl = st_l + off;
        unsigned long long* stdl = st_dl + off;
        int* list = reinterpret_cast<int*>(heaps + off + offs[slot]);  // the blob's heap segment holds the pixel list meanwhile
        const int* bb = bbox + slot * 4;
        const int y0 = bb[0], y1 = bb[1], x0 = bb[2], x1 = bb[3];
        // 1. the blob's unlabelled pixels (raster order for one wave; any order for a workgroup); are there markers at all?
        int n_open = 0;
        bool any_marker = false;
        if constexpr (BLOCK) {
            if (tid == 0) sh_count = 0, sh_marker = 0;
            __syncthreads();
            const int bw = x1 - x0 + 1;
            const long box = (long)(y1 - y0 + 1) * bw;
            for (long base = 0; base < box; base += NT) {  // uniform trip count
                const long t = base + tid;
                const int yy = (int)(t / bw), xx = (int)(t - (long)yy * bw);
                const long i = (long)(y0 + yy) * w + x0 + xx;
                const bool mine = t < box && bl[i] == label;
                const int o = mine ? out[i] : 0;
                const bool open = o < 0;
                const unsigned long long m = __ballot(open);
                int wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(&sh_count, __builtin_popcountll(m));
                wbase = __shfl(wbase, 0);
                if (open) {
                    __hip_atomic_store(list + wbase + __builtin_popcountll(m & lt_mask), (int)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    relax_store(stl + i, inf_bits);
                    relax_store(stdl + i, 0ull);
                }
                const unsigned long long mk = __ballot(o > 0);
                if (lane == 0 && mk) sh_marker = 1;
            }
            __syncthreads();
            n_open = sh_count;
            any_marker = sh_marker != 0;
            __syncthreads();  // sh_* are reset by the next big blob
        } else {
            for (int y = y0; y <= y1; ++y) {
                for (int xb = x0; xb <= x1; xb += 64) {
                    const int x = xb + lane;
                    const long i = (long)y * w + x;
                    const bool mine = x <= x1 && bl[i] == label;
                    const int o = mine ? out[i] : 0;
                    any_marker = any_marker || __ballot(o > 0) != 0ull;
                    const bool open = o < 0;
                    const unsigned long long m = __ballot(open);
                    if (open) {
                        __hip_atomic_store(list + n_open + __builtin_popcountll(m & lt_mask), (int)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        relax_store(stl + i, inf_bits);
                        relax_store(stdl + i, 0ull);
                    }
                    n_open += __builtin_popcountll(m);
                }
            }
        }
        if (!any_marker) {  // a blob without markers stays background
            for (int k = tid; k < n_open; k += NT) out[__hip_atomic_load(list + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = 0;
            if (tid == 0) done[slot] = 1;
            continue;
        }
        if (n_open == 0) {
            if (tid == 0) done[slot] = 1;
            continue;
        }
        // 2. relax until a whole sweep changes nothing
        bool failed = false;
        // a pass costs about one memory round trip, like a pop of the heap flood: past 2 x area passes the heap is cheaper
        const long budget = 2L * area + 64;
        long passes = 0;
        for (int sweep = 0;; ++sweep) {
            bool changed = false;
            for (int base = 0; base < n_open; base += NT) {
                const int k0 = base + tid;
                const bool active = k0 < n_open;
                const int k = !active ? 0 : ((sweep & 1) ? n_open - 1 - k0 : k0);
                const long p = __hip_atomic_load(list + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int y = (int)(p / w), x = (int)(p - (long)y * w);
                const double v = ds[p];
                const unsigned long long cur_l = relax_load(stl + p), cur_dl = relax_load(stdl + p);
                double best_l = __builtin_huge_val();
                int best_d = 0, best_lab = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // up, left, right, down
                    const int yy = y + (j == 0 ? -1 : (j == 3 ? 1 : 0));
                    const int xx = x + (j == 1 ? -1 : (j == 2 ? 1 : 0));
                    const bool inb = yy >= 0 && yy < h && xx >= 0 && xx < w;
                    const RelaxNb nb = relax_neighbour(inb, inb ? (long)yy * w + xx : p, out, ds, stl, stdl);
                    const bool better = nb.l < best_l || (nb.l == best_l && nb.l < __builtin_huge_val() && nb.d < best_d);
                    best_l = better ? nb.l : best_l;
                    best_d = better ? nb.d : best_d;
                    best_lab = better ? nb.lab : best_lab;
                }
                const bool up = v > best_l;
                const double new_l = up ? v : best_l;
                const int new_d = up ? 0 : best_d + 1;
                const unsigned long long new_lb = (unsigned long long)__double_as_longlong(new_l);
                const unsigned long long new_dl = ((unsigned long long)(unsigned)new_d << 32) | (unsigned)best_lab;
                const bool ch = active && best_l < __builtin_huge_val() && (new_lb != cur_l || new_dl != cur_dl);
                if (ch) {
                    relax_store(stl + p, new_lb);
                    relax_store(stdl + p, new_dl);
                }
                changed = changed || ch;
            }
            passes += (n_open + NT - 1) / NT;
            if (!any_of(changed)) break;  // (a workgroup's barrier also orders this sweep's stores before the next sweep's loads)
            if (passes > budget) {
                failed = true;
                break;
            }
        }
        // 3. every unlabelled pixel: reached, and all its minimum-L neighbours carry its label?
        bool bad_any = false;
        for (int base = 0; base < n_open && !failed; base += NT) {
            const int k = base + tid;
            const bool active = k < n_open;
            const long p = __hip_atomic_load(list + (active ? k : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int y = (int)(p / w), x = (int)(p - (long)y * w);
            const unsigned long long cur_l = relax_load(stl + p), cur_dl = relax_load(stdl + p);
            const int my_lab = (int)(unsigned)cur_dl;
            RelaxNb nb[4];
            double min_l = __builtin_huge_val();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = y + (j == 0 ? -1 : (j == 3 ? 1 : 0));
                const int xx = x + (j == 1 ? -1 : (j == 2 ? 1 : 0));
                const bool inb = yy >= 0 && yy < h && xx >= 0 && xx < w;
                nb[j] = relax_neighbour(inb, inb ? (long)yy * w + xx : p, out, ds, stl, stdl);
                min_l = nb[j].l < min_l ? nb[j].l : min_l;
            }
            bool bad = cur_l == inf_bits || my_lab <= 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) bad = bad || (nb[j].l == min_l && nb[j].lab != my_lab);
            bad_any = bad_any || (active && bad);
        }
        if (!failed) failed = any_of(bad_any);
        if (failed) {
            if (tid == 0) done[slot] = 0;
            continue;
        }
        // 4. commit
        for (int k = tid; k < n_open; k += NT) {
            const int p = __hip_atomic_load(list + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[p] = (int)(unsigned)relax_load(stdl + p);
        }
        if (tid == 0) done[slot] = 1;
    }
}

// ---- instance statistics ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HT) void inst_stats_init_kernel(long long* __restrict__ stats, long n_entries) {
    for (long i = (long)blockIdx.x * HT + threadIdx.x; i < n_entries; i += (long)gridDim.x * HT) {
        stats[i * 8 + 1] = 0x7fffffffLL;
        stats[i * 8 + 2] = 0x7fffffffLL;
        stats[i * 8 + 3] = -1;
        stats[i * 8 + 4] = -1;
    }
}
// Runs of equal (id, type) along a row are merged across the lanes of a wave (ballot), so the atomics are
// issued once per run instead of once per pixel.
__global__ __launch_bounds__(HT) void inst_stats_kernel(const int* __restrict__ inst, const uint8_t* __restrict__ type, int h, int w,
                                                         int max_inst, int num_types, long long* __restrict__ stats,
                                                         int* __restrict__ types) {
    const long hw = (long)h * w;
    const size_t off = (size_t)blockIdx.y * hw;
    long long* st = stats + (size_t)blockIdx.y * (max_inst + 1) * 8;
    int* ty = types ? types + (size_t)blockIdx.y * (max_inst + 1) * num_types : nullptr;
    const int lane = lane_id();
    for (long base = (long)blockIdx.x * HT; base < hw; base += (long)gridDim.x * HT) {  // uniform trip count
        const long i = base + threadIdx.x;
        int id = i < hw ? inst[off + i] : 0;
        if (id < 0 || id > max_inst) id = 0;
        const int t = (ty && type && id > 0) ? (int)type[off + i] : 0;
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        const int pid = __shfl_up(id, 1), pt = __shfl_up(t, 1);
        const bool head = lane == 0 || id != pid || t != pt || x == 0;
        const unsigned long long heads = __ballot(head);
        if (!head || id == 0) continue;
        const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
        const long long len = (above ? __builtin_ctzll(above) : 64) - lane;
        long long* s = st + (size_t)id * 8;
        atomicAdd((unsigned long long*)&s[0], (unsigned long long)len);
        atomicMin(&s[1], (long long)x);
        atomicMin(&s[2], (long long)y);
        atomicMax(&s[3], (long long)x + len - 1);
        atomicMax(&s[4], (long long)y);
        atomicAdd((unsigned long long*)&s[5], (unsigned long long)(len * x + len * (len - 1) / 2));
        atomicAdd((unsigned long long*)&s[6], (unsigned long long)(len * y));
        if (ty && type && t < num_types) atomicAdd(&ty[(size_t)id * num_types + t], (int)len);
    }
}

// getSobelKernels (OpenCV deriv.cpp): order-1 (derivative) and order-0 (smoothing) integer taps
static void sobel_taps_host(int ksize, SobelTaps& kd, SobelTaps& ks) {
    for (int order = 0; order < 2; ++order) {
        long long ker[64];
        for (int i = 0; i <= ksize; ++i) ker[i] = 0;
        ker[0] = 1;
        for (int i = 0; i < ksize - order - 1; ++i) {
            long long oldval = ker[0];
            for (int j = 1; j <= ksize; ++j) {
                const long long newval = ker[j] + ker[j - 1];
                ker[j - 1] = oldval;
                oldval = newval;
            }
        }
        for (int i = 0; i < order; ++i) {
            long long oldval = -ker[0];
            for (int j = 1; j <= ksize; ++j) {
                const long long newval = ker[j - 1] - ker[j];
                ker[j - 1] = oldval;
                oldval = newval;
            }
        }
        SobelTaps& dst = order ? kd : ks;
        for (int i = 0; i < 32; ++i) dst.v[i] = i < ksize ? (double)ker[i] : 0.0;
    }
}

struct HoverWs {
    size_t total;
    size_t blb_mask, tmp_a, tmp_b, blob_lab, mark_lab, ws_int, bbox, offs, cnt_blob, rowbuf, sob_h, sob_v, dist0, dist, heaps,
        mm, taps, se_offs;
};
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static HoverWs hover_layout(long n, long hw) {
    HoverWs L{};
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align256(o + bytes);
        return at;
    };
    L.blb_mask = take((size_t)n * hw);
    L.tmp_a = take((size_t)n * hw);
    L.tmp_b = take((size_t)n * hw);
    L.blob_lab = take((size_t)n * hw * 4);
    L.mark_lab = take((size_t)n * hw * 4);
    L.ws_int = take(((size_t)2 * n * hw + 2 * n + (size_t)n * (hw + 1)) * 4);  // CCL / fill-holes / area scratch
    L.bbox = take((size_t)n * (hw + 1) * 16);
    L.offs = take((size_t)n * (hw + 1) * 4);
    L.cnt_blob = take((size_t)n * 4 * 2);
    L.rowbuf = take((size_t)n * hw * 8);
    L.sob_h = take((size_t)n * hw * 8);
    L.sob_v = take((size_t)n * hw * 8);
    L.dist0 = take((size_t)n * hw * 8);
    L.dist = take((size_t)n * hw * 8);
    L.heaps = take((size_t)n * hw * sizeof(HeapItem));
    L.mm = take((size_t)n * 2 * 8 * 4);
    L.taps = take(64 * 8 * 2);
    L.se_offs = take(64 * 8);
    L.total = o;
    return L;
}

}  // namespace tia

using namespace tia;

extern "C" size_t tia_hover_workspace_bytes(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return hover_layout((long)n, (long)h * w).total;
}

// watershed(image, markers, mask = blob labels > 0): ws_init + one priority flood per blob.  `areas` = per-label pixel
// counts of blob_lab ([n][hw+1]), `offs` = their exclusive scan (heap segment offsets), min_keep = smallest blob kept.
// `relax_l` / `relax_dl` ([n][hw] 64-bit words each) and `done` ([n][hw+1] ints) serve the relaxation pass.
static int launch_watershed(const int* blob_lab, const int* mark_lab, const double* dist, const int* areas, const int* offs,
                            const int* cnt_blob, int* bbox, HeapItem* heaps, unsigned long long* relax_l,
                            unsigned long long* relax_dl, int* done, int* d_inst, long n, int h, int w, int min_keep,
                            hipStream_t st, bool init_done = false) {
    const long hw = (long)h * w;
    dim3 grid(hblocks(hw), (unsigned)n);
    if (!init_done) hipLaunchKernelGGL(ws_init_kernel, grid, dim3(HT), 0, st, blob_lab, mark_lab, h, w, d_inst, bbox);
    const long max_labels = hw / 2 + 2;
    long fx = (max_labels + 63) / 64, fcap = 65536 / n > 4 ? 65536 / n : 4;  // lanes stride over labels beyond the cap
    dim3 fgrid((unsigned)n, (unsigned)(fx < fcap ? fx : fcap));
    static const int wave_per_blob = [] {
        const char* e = tia::dev_env("TIA_FLOOD_WAVE");  // developer switch: 0 = one lane per blob
        return e ? atoi(e) : 1;
    }();
    static const int relax = [] {
        const char* e = tia::dev_env("TIA_FLOOD_RELAX");  // developer switch: 0 = sequential heap flood for every blob
        return e ? atoi(e) : 1;
    }();
    if (wave_per_blob) {
        long wy = max_labels < 4096 ? max_labels : 4096, wcap = 262144 / n > 16 ? 262144 / n : 16;
        dim3 wgrid((unsigned)n, (unsigned)(wy < wcap ? wy : wcap));
        // all blobs by parallel relaxation; the (rare) blobs whose labelling hinges on an exact tie are left to the heap
        if (relax) {
            hipLaunchKernelGGL(ws_relax_kernel<64>, wgrid, dim3(64), 0, st, blob_lab, dist, areas, offs, cnt_blob, bbox, h, w,
                               min_keep, heaps, relax_l, relax_dl, done, d_inst);
            if (hw > RELAX_BIG) {  // blobs of more than RELAX_BIG pixels: one 1024-thread workgroup each
                const long by = hw / RELAX_BIG < 64 ? hw / RELAX_BIG : 64;
                hipLaunchKernelGGL(ws_relax_kernel<1024>, dim3((unsigned)n, (unsigned)(by < 1 ? 1 : by)), dim3(1024), 0, st, blob_lab,
                                   dist, areas, offs, cnt_blob, bbox, h, w, min_keep, heaps, relax_l, relax_dl, done, d_inst);
            }
        }
        hipLaunchKernelGGL(ws_flood_wave_kernel, wgrid, dim3(64), 0, st, blob_lab, dist, areas, offs, cnt_blob, bbox, h, w,
                           min_keep, heaps, relax ? done : nullptr, d_inst);
    } else {
        hipLaunchKernelGGL(ws_flood_kernel, fgrid, dim3(64), 0, st, blob_lab, dist, areas, offs, cnt_blob, bbox, h, w,
                           min_keep, heaps, d_inst);
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

// Optional copies of the pipeline's intermediate planes (any pointer may be null): what the parity tests compare with
// the oracle stage by stage, so that a compensating error cannot hide behind an identical final label map.
struct HoverTaps {
    double* sobel_h;   // cv2.Sobel(normalised h, CV_64F, 1, 0, ksize)   [n,h,w]
    double* sobel_v;   // cv2.Sobel(normalised v, CV_64F, 0, 1, ksize)   [n,h,w]
    double* dist;      // -GaussianBlur((1 - overall) * blb)            [n,h,w]
    int32_t* markers;  // labelled, size-filtered markers                [n,h,w]
    int32_t* blobs;    // 1 where blb (after the size filter), else 0    [n,h,w]
};

static int hover_proc_impl(const float* d_np, const float* d_hv, int64_t n, int64_t h, int64_t w, int32_t ksize,
                           int32_t obj_size, int32_t* d_inst, int32_t* d_ninst, void* d_ws, size_t ws_bytes, void* stream,
                           const HoverTaps& taps) {
    if (!d_np || !d_hv || !d_inst || !d_ninst || !d_ws) return TIA_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || n > 65535 || ksize < 5 || ksize > 31 || (ksize & 1) == 0) return TIA_EINVAL;
    const long hw = (long)h * w;
    if (hw > 0x3fffffffL) return TIA_ESIZE;
    const HoverWs L = hover_layout((long)n, hw);
    if (ws_bytes < L.total) return TIA_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)d_ws;
    uint8_t* blb_mask = (uint8_t*)(base + L.blb_mask);
    uint8_t* tmp_a = (uint8_t*)(base + L.tmp_a);
    uint8_t* tmp_b = (uint8_t*)(base + L.tmp_b);
    int* blob_lab = (int*)(base + L.blob_lab);
    int* mark_lab = (int*)(base + L.mark_lab);
    int* ws_int = (int*)(base + L.ws_int);
    int* bbox = (int*)(base + L.bbox);
    int* offs = (int*)(base + L.offs);
    int* cnt_blob = (int*)(base + L.cnt_blob);
    double* rowbuf = (double*)(base + L.rowbuf);
    double* sob_h = (double*)(base + L.sob_h);
    double* sob_v = (double*)(base + L.sob_v);
    double* dist0 = (double*)(base + L.dist0);
    double* dist = (double*)(base + L.dist);
    HeapItem* heaps = (HeapItem*)(base + L.heaps);
    double* mm = (double*)(base + L.mm);  // [4][n][2]: h raw, v raw, sobel h, sobel v
    int* se_offs = (int*)(base + L.se_offs);
    int* areas = ws_int;  // tia_label_area_filter_i32 leaves the per-label areas here
    const size_t plane_f64 = (size_t)n * hw * sizeof(double), plane_i32 = (size_t)n * hw * sizeof(int32_t);
    auto tap = [&](void* dst, const void* src, size_t bytes) {
        return !dst || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    };

    dim3 grid(hblocks(hw), (unsigned)n);
    int rc;
    // 1. blb = remove_small_objects(label(np >= 0.5), max_size=9) > 0
    const bool tile = ccl_tile_enabled() && hw <= kCclTileMaxPixels;  // small planes: threshold + labelling + area filter in ONE launch
    // the fully fused path: every stage of a plane as a tile-resident kernel (6 launches in all)
    const long band_rows_max = (144L * 1024) / (w * 12) - (ksize - 1);  // f64 row-pass band + f32 input band in LDS
    const bool fused = tile && hw <= kMarkerTileMaxPixels && band_rows_max >= 8;
    int* areas_keep = (int*)dist;
    if (fused) {
        // threshold + labelling + area filter + heap offsets + bounding-box reset (areas stay in ws_int: nothing below reuses it)
        rc = ccl_tile_label(d_np, 2, (long)n, (int)h, (int)w, 4, 10, blob_lab, cnt_blob, ws_int, st, offs, bbox);
        if (rc != TIA_OK) return rc;
    } else {
        if (tile) {
            rc = ccl_tile_label(d_np, 2, (long)n, (int)h, (int)w, 4, 10, blob_lab, cnt_blob, ws_int, st);  // areas -> ws_int
            if (rc != TIA_OK) return rc;
        } else {
            hipLaunchKernelGGL(np_threshold_kernel, dim3(hblocks((long)n * hw, HT, 65535)), dim3(HT), 0, st, d_np, (long)n * hw, blb_mask);
            rc = tia_ccl_label_i32(blb_mask, n, h, w, 4, blob_lab, cnt_blob, ws_int, st);
            if (rc != TIA_OK) return rc;
            rc = tia_label_area_filter_i32(blob_lab, n, h, w, 10, ws_int, st);  // areas stay in ws_int
            if (rc != TIA_OK) return rc;
        }
        // blob bounding boxes + heap offsets need the areas: do them before ws_int is reused
        hipLaunchKernelGGL(bbox_reset_kernel, dim3(hblocks((long)n * (hw + 1), HT, 65535)), dim3(HT), 0, st, bbox, (long)n * (hw + 1));
        hipLaunchKernelGGL(ws_offsets_kernel, dim3((unsigned)n), dim3(1024), 0, st, areas, cnt_blob, hw, 10, offs);
        // ws_int is scratch for the marker pipeline below: park the blob areas in `dist` (unused until step 5)
        if (hipMemcpyAsync(areas_keep, areas, (size_t)n * (hw + 1) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return TIA_ELAUNCH;
    }

    // 2. Sobel of the normalised h / v maps, 3. energy + marker seed
    SobelTaps kd, ks;
    sobel_taps_host(ksize, kd, ks);
    // small planes: min / max + both filters + energy in one launch, the row pass staged in LDS in bands of rows
    using SobelKernel = void (*)(const float*, int, int, const SobelTaps, const SobelTaps, int, int, int, const int*, double*, double*,
                                 double*, uint8_t*, double*, double*);
    static const SobelKernel variants[3] = {sobel_energy_tile_kernel<21>, sobel_energy_tile_kernel<11>, sobel_energy_tile_kernel<0>};
    static DeviceOnce sobel_once;
    bool tile_sobel = tile && band_rows_max >= 8;
    if (tile_sobel && !sobel_once.ensure([] {
            for (const SobelKernel k : variants)
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess)
                    return false;
            return true;
        })) {
        (void)hipGetLastError();  // a device without 144 KB of LDS per workgroup: the multi-launch Sobel below serves it
        tile_sobel = false;
    }
    if (tile_sobel) {
        const int band = (int)(band_rows_max < h ? band_rows_max : h);
        const size_t lds = (size_t)(band + ksize - 1) * w * 12;
        hipLaunchKernelGGL(variants[ksize == 21 ? 0 : (ksize == 11 ? 1 : 2)], dim3((unsigned)n), dim3(1024), lds, st, d_hv, (int)h, (int)w, kd,
                           ks, ksize, band, (int)n, (const int*)blob_lab, sob_h, sob_v, dist0, tmp_a, mm, fused ? dist : (double*)nullptr);
        if (!tap(taps.sobel_h, sob_h, plane_f64) || !tap(taps.sobel_v, sob_v, plane_f64)) return TIA_ELAUNCH;
    } else {
    hipLaunchKernelGGL(minmax_hv_kernel, dim3((unsigned)n), dim3(1024), 0, st, d_hv, hw, mm, mm + 2 * n);
    // h: dx=1 -> kx = derivative taps, ky = smoothing taps (symmetric column filter)
    hipLaunchKernelGGL(sobel_row_kernel, grid, dim3(HT), 0, st, d_hv, 0, (int)h, (int)w, mm, kd, ksize, rowbuf);
    hipLaunchKernelGGL(sobel_col_kernel, grid, dim3(HT), 0, st, rowbuf, (int)h, (int)w, ks, ksize, 1, sob_h);
    // v: dy=1 -> kx = smoothing taps, ky = derivative taps (antisymmetric column filter)
    hipLaunchKernelGGL(sobel_row_kernel, grid, dim3(HT), 0, st, d_hv, 1, (int)h, (int)w, mm + 2 * n, ks, ksize, rowbuf);
    hipLaunchKernelGGL(sobel_col_kernel, grid, dim3(HT), 0, st, rowbuf, (int)h, (int)w, kd, ksize, 0, sob_v);
    hipLaunchKernelGGL(minmax_pair_kernel, dim3((unsigned)n), dim3(1024), 0, st, sob_h, sob_v, hw, mm + 4 * n, mm + 6 * n);
    if (!tap(taps.sobel_h, sob_h, plane_f64) || !tap(taps.sobel_v, sob_v, plane_f64)) return TIA_ELAUNCH;

    // 3. energy, marker seed
    hipLaunchKernelGGL(energy_kernel, grid, dim3(HT), 0, st, sob_h, sob_v, mm + 4 * n, mm + 6 * n, blob_lab, hw, dist0, tmp_a);
    }

    // 4. marker = label(open5x5(fill_holes(marker0))), small objects removed
    if (fused) {  // the whole marker pipeline of a plane + the watershed's initial state in one launch, resident in LDS
        rc = marker_tile(tmp_a, (long)n, (int)h, (int)w, obj_size, mark_lab, d_ninst, nullptr, st, blob_lab, d_inst, bbox);
        if (rc != TIA_OK) return rc;
    } else if (tile && hw <= kMarkerTileMaxPixels) {
        rc = marker_tile(tmp_a, (long)n, (int)h, (int)w, obj_size, mark_lab, d_ninst, ws_int, st);
        if (rc != TIA_OK) return rc;
    } else {
        rc = tia_fill_holes_u8(tmp_a, n, h, w, tmp_b, ws_int, st);
        if (rc != TIA_OK) return rc;
        {
            // 5x5 ellipse: rows 00100 / 11111 / 11111 / 11111 / 00100 (cv2.getStructuringElement)
            static const int host_offs[17 * 2] = {-2, 0,  -1, -2, -1, -1, -1, 0, -1, 1, -1, 2, 0, -2, 0, -1, 0, 0,
                                                  0,  1,  0,  2,  1,  -2, 1,  -1, 1, 0, 1,  1, 1, 2,  2, 0};
            if (hipMemcpyAsync(se_offs, host_offs, sizeof(host_offs), hipMemcpyHostToDevice, st) != hipSuccess) return TIA_ELAUNCH;
        }
        rc = tia_binary_morph_u8(tmp_b, n, h, w, se_offs, 17, 1, tmp_a, st);
        if (rc != TIA_OK) return rc;
        rc = tia_binary_morph_u8(tmp_a, n, h, w, se_offs, 17, 0, tmp_b, st);
        if (rc != TIA_OK) return rc;
        if (tile) {
            rc = ccl_tile_label(tmp_b, 0, (long)n, (int)h, (int)w, 4, obj_size, mark_lab, d_ninst, ws_int, st);
            if (rc != TIA_OK) return rc;
        } else {
            rc = tia_ccl_label_i32(tmp_b, n, h, w, 4, mark_lab, d_ninst, ws_int, st);
            if (rc != TIA_OK) return rc;
            rc = tia_label_area_filter_i32(mark_lab, n, h, w, obj_size, ws_int, st);
            if (rc != TIA_OK) return rc;
        }
    }
    if (!tap(taps.markers, mark_lab, plane_i32)) return TIA_ELAUNCH;

    // 5. watershed(dist, markers, mask = blb)
    if (!fused) {
        // areas_keep lives in `dist`: move it to ws_int (free again) before dist is written
        if (hipMemcpyAsync(ws_int, areas_keep, (size_t)n * (hw + 1) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return TIA_ELAUNCH;
        hipLaunchKernelGGL(gauss3_neg_kernel, grid, dim3(HT), 0, st, dist0, (int)h, (int)w, dist);
    }
    if (!tap(taps.dist, dist, plane_f64)) return TIA_ELAUNCH;
    if (taps.blobs)
        hipLaunchKernelGGL(blob_indicator_kernel, dim3(hblocks((long)n * hw, HT, 65535)), dim3(HT), 0, st, blob_lab, (long)n * hw,
                           taps.blobs);
    // the Sobel planes and the row buffer are free by now: relaxation state and per-blob flags
    return launch_watershed(blob_lab, mark_lab, dist, ws_int, offs, cnt_blob, bbox, heaps, (unsigned long long*)sob_h,
                            (unsigned long long*)sob_v, (int*)rowbuf, d_inst, (long)n, (int)h, (int)w, 10, st, /*init_done=*/fused);
}

extern "C" int tia_hover_proc_np_hv_f32(const float* d_np, const float* d_hv, int64_t n, int64_t h, int64_t w,
                                         int32_t ksize, int32_t obj_size, int32_t* d_inst, int32_t* d_ninst, void* d_ws,
                                         size_t ws_bytes, void* stream) {
    return hover_proc_impl(d_np, d_hv, n, h, w, ksize, obj_size, d_inst, d_ninst, d_ws, ws_bytes, stream, HoverTaps{});
}

extern "C" int tia_hover_proc_np_hv_stages_f32(const float* d_np, const float* d_hv, int64_t n, int64_t h, int64_t w,
                                                int32_t ksize, int32_t obj_size, int32_t* d_inst, int32_t* d_ninst,
                                                double* d_sobel_h, double* d_sobel_v, double* d_dist, int32_t* d_markers,
                                                int32_t* d_blobs, void* d_ws, size_t ws_bytes, void* stream) {
    HoverTaps taps{d_sobel_h, d_sobel_v, d_dist, d_markers, d_blobs};
    return hover_proc_impl(d_np, d_hv, n, h, w, ksize, obj_size, d_inst, d_ninst, d_ws, ws_bytes, stream, taps);
}

// ---- stand-alone marker-controlled watershed ---------------------------------------------------------------------------
struct WatershedWs {
    size_t total, blob_lab, ws_int, bbox, offs, cnt, heaps, relax_l, relax_dl, done;
};
static WatershedWs watershed_layout(long n, long hw) {
    WatershedWs L{};
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align256(o + bytes);
        return at;
    };
    L.blob_lab = take((size_t)n * hw * 4);
    L.ws_int = take(((size_t)2 * n * hw + 2 * n + (size_t)n * (hw + 1)) * 4);
    L.bbox = take((size_t)n * (hw + 1) * 16);
    L.offs = take((size_t)n * (hw + 1) * 4);
    L.cnt = take((size_t)n * 4 * 2);
    L.heaps = take((size_t)n * hw * sizeof(HeapItem));
    L.relax_l = take((size_t)n * hw * 8);
    L.relax_dl = take((size_t)n * hw * 8);
    L.done = take((size_t)n * (hw + 1) * 4);
    L.total = o;
    return L;
}

extern "C" size_t tia_watershed_workspace_bytes(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return watershed_layout((long)n, (long)h * w).total;
}

extern "C" int tia_watershed_blobs_f64(const double* d_image, const int32_t* d_markers, const uint8_t* d_mask, int64_t n,
                                        int64_t h, int64_t w, int32_t* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_image || !d_markers || !d_mask || !d_out || !d_ws) return TIA_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || n > 65535) return TIA_EINVAL;
    const long hw = (long)h * w;
    if (hw > 0x3fffffffL) return TIA_ESIZE;
    const WatershedWs L = watershed_layout((long)n, hw);
    if (ws_bytes < L.total) return TIA_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)d_ws;
    int* blob_lab = (int*)(base + L.blob_lab);
    int* ws_int = (int*)(base + L.ws_int);
    int* bbox = (int*)(base + L.bbox);
    int* offs = (int*)(base + L.offs);
    int* cnt = (int*)(base + L.cnt);
    HeapItem* heaps = (HeapItem*)(base + L.heaps);
    // connectivity-1 floods never cross between 4-connected components of the mask: one private queue per component
    int rc = tia_ccl_label_i32(d_mask, n, h, w, 4, blob_lab, cnt, ws_int, st);
    if (rc != TIA_OK) return rc;
    rc = tia_label_area_filter_i32(blob_lab, n, h, w, 1, ws_int, st);  // keeps every blob; leaves the areas in ws_int
    if (rc != TIA_OK) return rc;
    hipLaunchKernelGGL(bbox_reset_kernel, dim3(hblocks((long)n * (hw + 1), HT, 65535)), dim3(HT), 0, st, bbox, (long)n * (hw + 1));
    hipLaunchKernelGGL(ws_offsets_kernel, dim3((unsigned)n), dim3(1024), 0, st, ws_int, cnt, hw, 1, offs);
    return launch_watershed(blob_lab, d_markers, d_image, ws_int, offs, cnt, bbox, heaps, (unsigned long long*)(base + L.relax_l),
                            (unsigned long long*)(base + L.relax_dl), (int*)(base + L.done), d_out, (long)n, (int)h, (int)w, 1, st);
}

extern "C" int tia_hover_instance_stats(const int32_t* d_inst, const uint8_t* d_type, int64_t n, int64_t h, int64_t w,
                                         int32_t max_inst, int32_t num_types, int64_t* d_stats, int32_t* d_types, void* stream) {
    if (!d_inst || !d_stats || n <= 0 || h <= 0 || w <= 0 || n > 65535 || max_inst < 0) return TIA_EINVAL;
    if (d_type && (!d_types || num_types <= 0)) return TIA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long entries = (long)n * (max_inst + 1);
    hipLaunchKernelGGL(inst_stats_init_kernel, dim3(hblocks(entries, HT, 65535)), dim3(HT), 0, st, (long long*)d_stats, entries);
    dim3 grid(hblocks((long)h * w), (unsigned)n);
    hipLaunchKernelGGL(inst_stats_kernel, grid, dim3(HT), 0, st, d_inst, d_type, (int)h, (int)w, max_inst, num_types,
                       (long long*)d_stats, d_type ? d_types : (int*)nullptr);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
