// Streaming per-pixel stain kernels on gfx950 (1 read + 1 write per pixel).
//   stain_apply   : out = 255*exp(-(OD . M))            tools/stainnorm.py:102-113
//   concentrations: C = OD . pinv                         tools/stainnorm.py:49-66
//   augment       : C[mask]*alpha+beta, recompose         tools/stainaugment.py:177-206
//   luminosity mask                                       utils/misc.py:261-290
// Layout: NHWC uint8 in; lanes read 12 contiguous bytes (4 whole pixels) so a wave covers 768
// contiguous bytes per load and no pixel straddles two lanes.  The OD look-up table lives in
// LDS, replicated once per bank (lut[v][lane&31]) so data-dependent look-ups never conflict.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.hpp"
#include "wide_io.hpp"

#pragma clang fp contract(off)

namespace tia {

constexpr int AT = 256;  // threads per block for the streaming kernels
#ifndef TIA_LUT_REP
#define TIA_LUT_REP 8
#endif
constexpr int REP = TIA_LUT_REP;  // copies of the f32 OD table in LDS (one per bank when 32)

struct StainT {
    double s[6];
};

// ---- per-block folded luminance tables (same arithmetic as stain_stats P1) --------------------
__device__ __forceinline__ void build_ty(int (*ty)[256], const tia_stain_tables* __restrict__ tab,
                                         double plow, double phigh, bool z1) {
    for (int tid = threadIdx.x; tid < 256; tid += blockDim.x) {
        int v = tid;
        if (z1 && v == 0) v = 1;
        int ce = v;
        if (phigh > plow) {
            double x = (double)v;
            x = x < plow ? plow : (x > phigh ? phigh : x);
            x = (x - plow) / (phigh - plow);
            x = x * 255.0 + 0.0;
            ce = (int)x;
        }
        ty[0][tid] = tab->ty[0][ce];
        ty[1][tid] = tab->ty[1][ce];
        ty[2][tid] = tab->ty[2][ce];
    }
}

// ---- output writers -----------------------------------------------------------------------------
template <int OUT>
struct Out;

template <>
struct Out<TIA_OUT_U8> {
    using T = uint8_t;
    template <class F>
    static __device__ __forceinline__ T cvt(F v) { return (T)(unsigned)v; }  // astype(uint8): truncation
};
template <>
struct Out<TIA_OUT_F32> {
    using T = float;
    template <class F>
    static __device__ __forceinline__ T cvt(F v) { return (float)v; }
};
template <>
struct Out<TIA_OUT_F64> {
    using T = double;
    template <class F>
    static __device__ __forceinline__ T cvt(F v) { return (double)v; }
};
// ToTensor() of the truncated uint8: float32(u8)/255, then rounded to the storage type.
// x/255 for the 256 possible bytes: q = x*(1/255f) followed by one Markstein correction step
// (e = fma(-255,q,x); q' = fma(e,1/255f,q)) is the correctly rounded float32 quotient for every byte,
// and for the 16-bit storage types the uncorrected product already rounds to the same half/bfloat16
// (checked exhaustively on the host, tests/test_stain_gpu.py::test_unit_outputs_equal_totensor).
__device__ __forceinline__ float trunc_nonneg(float v) { return floorf(v); }
__device__ __forceinline__ float trunc_nonneg(double v) { return (float)floor(v); }
__device__ __forceinline__ float unit_q(float x) { return x * 0.00392156885936856269836425781250f; }
template <>
struct Out<TIA_OUT_UNIT_F32> {
    using T = float;
    template <class F>
    static __device__ __forceinline__ T cvt(F v) {
        const float x = trunc_nonneg(v), r = 0.00392156885936856269836425781250f;
        const float q = x * r;
        return __builtin_fmaf(__builtin_fmaf(-255.0f, q, x), r, q);
    }
};
template <>
struct Out<TIA_OUT_UNIT_F16> {
    using T = __half;
    template <class F>
    static __device__ __forceinline__ T cvt(F v) { return __float2half_rn(unit_q(trunc_nonneg(v))); }
};
template <>
struct Out<TIA_OUT_UNIT_BF16> {
    using T = __hip_bfloat16;
    template <class F>
    static __device__ __forceinline__ T cvt(F v) { return __float2bfloat16(unit_q(trunc_nonneg(v))); }
};

template <class T, int BYTES = sizeof(T) * 12>
struct Pack12 {
    alignas(16) T v[12];
};

template <class T>
__device__ __forceinline__ void store12(T* __restrict__ dst, const Pack12<T>& pk) {
    constexpr int bytes = sizeof(T) * 12;
    if constexpr (bytes == 12) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(pk.v);
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        __builtin_nontemporal_store(w[0], d);
        __builtin_nontemporal_store(w[1], d + 1);
        __builtin_nontemporal_store(w[2], d + 2);
    } else if constexpr (bytes == 24) {
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(pk.v);
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
        __builtin_nontemporal_store(w[0], d);
        __builtin_nontemporal_store(w[1], d + 1);
        __builtin_nontemporal_store(w[2], d + 2);
    } else {
        using v4 = __attribute__((ext_vector_type(4))) unsigned;
        const v4* w = reinterpret_cast<const v4*>(pk.v);
        v4* d = reinterpret_cast<v4*>(dst);
#pragma unroll
        for (int i = 0; i < bytes / 16; ++i) __builtin_nontemporal_store(w[i], d + i);
    }
}

// Wave-private LDS exchange point: every lane's LDS stores before it are visible to every lane's LDS loads after it.
// The fences are what tells the COMPILER: a lane never reads back an address it wrote itself in these transposes (the data
// comes from other lanes), so in-thread alias analysis alone would let it hoist the loads out of the loop / above the stores
// (seen: one of the three read-backs became loop-invariant).  Wavefront scope: no cache maintenance, no extra waits (LDS
// operations of a wave execute in order).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- stain apply ---------------------------------------------------------------------------------
template <int MATH>
struct ApplyCtx;

// f32: fused 3x3 matrix, pre-scaled by -log2(e); hardware exp2.
template <>
struct ApplyCtx<TIA_MATH_F32> {
    float m[9];
    const float* lut;  // bank-private: lut[v*32 + (lane&31)]
    int bank;
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, float (&o)[3]) const {
        const float x = lut[r * REP + bank], y = lut[g * REP + bank], z = lut[b * REP + bank];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float u = __builtin_fmaf(z, m[6 + c], __builtin_fmaf(y, m[3 + c], x * m[c]));
            float v = 255.0f * __builtin_amdgcn_exp2f(u);
            v = v > 255.0f ? 255.0f : v;  // trans[trans > 255] = 255
            v = v < 0.0f ? 0.0f : v;      // trans[trans < 0] = 0
            o[c] = v;
        }
    }
};

// f64, libm: the reference's order of operations (concentrations, rescale, recomposition, exp) with the device
// library's exp() -- TIA_MATH_F64_REF, and the fall-back of TIA_MATH_F64 for patches whose exponent range is not safe.
template <>
struct ApplyCtx<TIA_MATH_F64_REF> {
    double p[6], sc[2], st[6];
    const double* lut;
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, double (&o)[3]) const {
        const double x = lut[r], y = lut[g], z = lut[b];
        double c0 = x * p[0] + y * p[2] + z * p[4];
        double c1 = x * p[1] + y * p[3] + z * p[5];
        c0 *= sc[0];
        c1 *= sc[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double t = c0 * st[c] + c1 * st[3 + c];
            double v = 255.0 * exp(-1.0 * t);
            v = v > 255.0 ? 255.0 : v;
            v = v < 0.0 ? 0.0 : v;
            o[c] = v;
        }
    }
};

// f64, own exp (TIA_MATH_F64): 255 * exp(-(OD . M)) = T[n & 1023] * 2^(n >> 10) * (1 + r q(r)) with
//   a = OD . (M * -1024/ln 2) = n + r,  n = rint(a) (magic-number add: no conversion instruction),  |r| <= 1/2,
//   T[i] = 255 * 2^(i/1024) (exp2_table.inc, correctly rounded),  1 + r q(r) = the cubic Taylor polynomial of
//   2^(r/1024) (remainder (ln2/2048)^4/24 = 5.5e-16 relative), 2^(n >> 10) added into the exponent field.
// 16 float64 operations per channel instead of the ~45 of the library's exp(); relative error <= 4e-15 for
// |a| < 2^15 (the rounding of a dominates), i.e. < 1e-12 on the 0..255 scale -- the 1e-4 contract and every
// tolerance of tests/test_stain_gpu.py hold unchanged.  M = TIA_ST_M (pinv . diag(scale) . S_target fused by the
// statistics kernel in float64); the product differs from the reference's order of operations by a few ulp of t.
// Safe while 5.5414 * sum_j |m[j][c]| < 2^19 (exponent arithmetic stays inside the normal range; results above
// 255 are clipped like the reference's): checked per patch by the caller, which otherwise takes the libm context.
#include "exp2_table.inc.h"
constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
struct ApplyCtxFast {
    double m[9];
    const double* lut;
    const double* etab;
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, double (&o)[3]) const {
        const double x = lut[r], y = lut[g], z = lut[b];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // exp(-t) >= 1 <=> a >= 0: clipping a at 0 IS the reference's `trans[trans > 255] = 255` (n = 0, r = 0 gives
            // T[0] = 255 exactly), one v_min_f64 on a canonical operand instead of a min on the bit-assembled result
            const double a = __builtin_fmin(__builtin_fma(z, m[6 + c], __builtin_fma(y, m[3 + c], x * m[c])), 0.0);
            const double big = a + kMagic;
            const int lo = __double2loint(big);
            const double rr = a - (big - kMagic);
            const double q = __builtin_fma(rr, __builtin_fma(rr, 0x1.c6b08d704a0bfp-35, 0x1.ebfbdff82c58ep-23), 0x1.62e42fefa39efp-11);
            const double t = etab[lo & 1023];
            const double v = __builtin_fma(t, rr * q, t);
            // exponent add in unsigned arithmetic (lo >> 10 is negative for every pixel darker than white: shifting it left as a
            // signed int would be undefined behaviour; two's-complement wrap-around is what the add needs)
            o[c] = __hiloint2double((int)((unsigned)__double2hiint(v) + ((unsigned)(lo >> 10) << 20)), __double2loint(v));
        }
    }
    // m = M * (-1024 / ln 2); returns whether the exponent arithmetic is safe for every byte value
    __device__ __forceinline__ bool load(const double* __restrict__ st) {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                m[j * 3 + c] = st[TIA_ST_M + j * 3 + c] * -0x1.71547652b82fep+10;
                s += fabs(m[j * 3 + c]);
            }
            ok = ok && (s * 5.5414 < 524288.0);  // also false for NaN / inf
        }
        return ok;
    }
};

// Product-of-tables form of the same float64 recomposition (round 5): the output optical density is LINEAR in the
// three input optical densities, OD'_c = sum_j LUT[v_j] m[j][c], and each LUT[v_j] takes 256 values, so
//   255 exp(-OD'_c) = (255 exp(-LUT[r] m[0][c])) * exp(-LUT[g] m[1][c]) * exp(-LUT[b] m[2][c]) = T_0c[r] * T_1c[g] * T_2c[b]
// with nine 256-entry float64 tables per patch (2,304 libm `exp` per workgroup instead of three exponentials per pixel): per
// pixel 9 LDS reads, 6 multiplies and the reference's clip -- the kernel stops being bound by float64 vector arithmetic (60
// operations per pixel, §4.10) and runs at the memory system's pace.  Each factor is a correctly rounded-ish exp (< 1 ulp), the
// product three roundings more: relative error <= ~3e-16, i.e. < 1e-13 on the 0..255 scale (the exponent-trick form: 1e-12).
// Valid while every |LUT[v] m[j][c]| stays far inside exp's range (every |m| < 126, i.e. 5.5414 |m| < 700: checked per patch --
// NaN / inf entries fail the test -- else the libm context).
struct ApplyCtxTab {
    const double* pt;  // [9][256]: pt[(3 j + c) * 256 + v]
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, double (&o)[3]) const {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = __builtin_fmin(pt[c * 256 + r] * pt[(3 + c) * 256 + g] * pt[(6 + c) * 256 + b], 255.0);
    }
};

// one 12-byte group (4 whole pixels) per lane and step, U independent groups in flight
template <int OUT, int U, class Ctx, class F>
__device__ __forceinline__ void sweep12(const Ctx& ctx, const uint8_t* __restrict__ src, typename Out<OUT>::T* __restrict__ dst,
                                        long hw) {
    using O = Out<OUT>;
    using T = typename O::T;
    if ((hw & 3) == 0) {
        const long ng = hw >> 2;
        const long stride = (long)gridDim.x * AT;
        for (long g0 = (long)blockIdx.x * AT + threadIdx.x; g0 < ng; g0 += stride * U) {
            uint32_t a[U], b[U], c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long g = g0 + u * stride;
                if (g < ng) {
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(src + g * 12);
                    a[u] = __builtin_nontemporal_load(q);
                    b[u] = __builtin_nontemporal_load(q + 1);
                    c[u] = __builtin_nontemporal_load(q + 2);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long g = g0 + u * stride;
                if (g >= ng) break;
                F o[4][3];
                ctx.pixel(a[u] & 255u, (a[u] >> 8) & 255u, (a[u] >> 16) & 255u, o[0]);
                ctx.pixel(a[u] >> 24, b[u] & 255u, (b[u] >> 8) & 255u, o[1]);
                ctx.pixel((b[u] >> 16) & 255u, b[u] >> 24, c[u] & 255u, o[2]);
                ctx.pixel((c[u] >> 8) & 255u, (c[u] >> 16) & 255u, c[u] >> 24, o[3]);
                Pack12<T> pk;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) pk.v[i * 3 + ch] = O::cvt(o[i][ch]);
                store12<T>(dst + g * 12, pk);
            }
        }
    } else {
        for (long i = (long)blockIdx.x * AT + threadIdx.x; i < hw; i += (long)gridDim.x * AT) {
            F o[3];
            ctx.pixel(src[3 * i], src[3 * i + 1], src[3 * i + 2], o);
            dst[3 * i + 0] = O::cvt(o[0]);
            dst[3 * i + 1] = O::cvt(o[1]);
            dst[3 * i + 2] = O::cvt(o[2]);
        }
    }
}

__device__ __forceinline__ void load_ref_ctx(ApplyCtx<TIA_MATH_F64_REF>& ctx, const double* __restrict__ st, const StainT& tgt,
                                             const double* lut) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        ctx.p[i] = st[TIA_ST_PINV + i];
        ctx.st[i] = tgt.s[i];
    }
    ctx.sc[0] = st[TIA_ST_SCALE + 0];
    ctx.sc[1] = st[TIA_ST_SCALE + 1];
    ctx.lut = lut;
}

template <int MATH, int OUT>
__global__ __launch_bounds__(AT) void stain_apply_kernel(const uint8_t* __restrict__ img, long hw,
                                                          const tia_stain_tables* __restrict__ tab,
                                                          const double* __restrict__ stats, StainT tgt,
                                                          void* __restrict__ out_v) {
    using T = typename Out<OUT>::T;
    using F = typename std::conditional<MATH == TIA_MATH_F32, float, double>::type;
    constexpr int LUTN = (MATH == TIA_MATH_F32) ? 256 * REP : 256;
    __shared__ F lut[LUTN];

    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    T* dst = reinterpret_cast<T*>(out_v) + (size_t)patch * (size_t)hw * 3u;
    if constexpr (MATH == TIA_MATH_F32) {
        ApplyCtx<TIA_MATH_F32> ctx;
        for (int i = threadIdx.x; i < LUTN; i += AT) lut[i] = tab->od_lut_f32[i / REP];
        const double nl2e = -1.4426950408889634;
#pragma unroll
        for (int i = 0; i < 9; ++i) ctx.m[i] = (float)(st[TIA_ST_M + i] * nl2e);
        ctx.lut = lut;
        ctx.bank = threadIdx.x & (REP - 1);
        __syncthreads();
        sweep12<OUT, 4, ApplyCtx<TIA_MATH_F32>, float>(ctx, src, dst, hw);
    } else if constexpr (MATH == TIA_MATH_F64_REF) {
        for (int i = threadIdx.x; i < LUTN; i += AT) lut[i] = tab->od_lut[i];
        ApplyCtx<TIA_MATH_F64_REF> ctx;
        load_ref_ctx(ctx, st, tgt, lut);
        __syncthreads();
        sweep12<OUT, 2, ApplyCtx<TIA_MATH_F64_REF>, double>(ctx, src, dst, hw);
    } else {
        __shared__ double etab[1024];
        for (int i = threadIdx.x; i < LUTN; i += AT) lut[i] = tab->od_lut[i];
        for (int i = threadIdx.x; i < 1024; i += AT) etab[i] = kExp2Tab255[i];
        ApplyCtxFast fast;
        const bool ok = fast.load(st);
        fast.lut = lut;
        fast.etab = etab;
        __syncthreads();
        if (ok) {
            sweep12<OUT, 2, ApplyCtxFast, double>(fast, src, dst, hw);
        } else {
            ApplyCtx<TIA_MATH_F64_REF> ctx;
            load_ref_ctx(ctx, st, tgt, lut);
            sweep12<OUT, 1, ApplyCtx<TIA_MATH_F64_REF>, double>(ctx, src, dst, hw);
        }
    }
}

// ---- wide variant: 16-byte global accesses through a wave-private LDS transpose -------------------
// A wave owns 3072 contiguous input bytes (1024 pixels) per step: three fully coalesced 16 B/lane
// loads land in LDS linearly, each lane then reads back ITS 48 contiguous bytes (16 whole pixels;
// the 12-dword lane stride makes the three ds_read_b128 conflict-free), computes, writes its
// 48*sizeof(T) output bytes back to LDS and the wave stores them as coalesced 16 B/lane rows.
// 12-byte loads / 8-byte stores plateau at ~4.4 TB/s on MI355X; 16-byte accesses are what the
// memory path is built for (MI355X_MICROARCH.md: 8-B accesses run at 0.54-0.70x the 16-B rate).
template <int OUT, class Ctx, class F>
__device__ __forceinline__ void sweep_wide(const Ctx& ctx, uint8_t* mine, const uint8_t* __restrict__ src,
                                           uint8_t* __restrict__ dst, long hw) {
    using O = Out<OUT>;
    using T = typename O::T;
    constexpr int TS = sizeof(T);
    static_assert(TS == 1 || TS == 2, "wide path is for 1- and 2-byte outputs");
    constexpr int CHUNK = 3072;  // input bytes per wave step
    using v4 = __attribute__((ext_vector_type(4))) unsigned;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long nchunks = hw * 3 / CHUNK;
    const long wstride = (long)gridDim.x * (AT / 64);
    for (long c = (long)blockIdx.x * (AT / 64) + wv; c < nchunks; c += wstride) {
        const v4* gsrc = reinterpret_cast<const v4*>(src + c * CHUNK);
        v4 in[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) in[k] = __builtin_nontemporal_load(gsrc + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<v4*>(mine + k * 1024 + lane * 16) = in[k];
        wave_lds_sync();
        uint32_t w[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const v4 t = *reinterpret_cast<const v4*>(mine + lane * 48 + j * 16);
            w[j * 4 + 0] = t.x;
            w[j * 4 + 1] = t.y;
            w[j * 4 + 2] = t.z;
            w[j * 4 + 3] = t.w;
        }
        wave_lds_sync();  // everyone has read its pixels before the region is reused
        alignas(16) uint32_t res[12 * TS];  // the lane's 48 output values, packed
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // four groups of 4 pixels (12 bytes = 3 dwords each)
            const uint32_t a = w[q * 3], b = w[q * 3 + 1], cc = w[q * 3 + 2];
            F o[4][3];
            ctx.pixel(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, o[0]);
            ctx.pixel(a >> 24, b & 255u, (b >> 8) & 255u, o[1]);
            ctx.pixel((b >> 16) & 255u, b >> 24, cc & 255u, o[2]);
            ctx.pixel((cc >> 8) & 255u, (cc >> 16) & 255u, cc >> 24, o[3]);
            // explicit packing, group by group (left to the vectoriser, the uint8 / float64 instantiation gathered the bytes of
            // all 16 pixels first: 268 live registers, one wave per SIMD)
            uint32_t bits[12];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const T t = O::cvt(o[i][ch]);
                    if constexpr (TS == 1) {
                        bits[i * 3 + ch] = (uint32_t)t;
                    } else {
                        unsigned short h;
                        __builtin_memcpy(&h, &t, 2);
                        bits[i * 3 + ch] = h;
                    }
                }
            if constexpr (TS == 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    res[q * 3 + k] = bits[4 * k] | (bits[4 * k + 1] << 8) | (bits[4 * k + 2] << 16) | (bits[4 * k + 3] << 24);
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) res[q * 6 + k] = bits[2 * k] | (bits[2 * k + 1] << 16);
            }
            if constexpr (sizeof(F) == 8) __builtin_amdgcn_sched_barrier(0);  // float64: one group at a time
        }
        const v4* rv = reinterpret_cast<const v4*>(res);
#pragma unroll
        for (int j = 0; j < 3 * TS; ++j) *reinterpret_cast<v4*>(mine + lane * 48 * TS + j * 16) = rv[j];
        wave_lds_sync();
        v4* gdst = reinterpret_cast<v4*>(dst + c * CHUNK * TS);
#pragma unroll
        for (int k = 0; k < 3 * TS; ++k) {
            const v4 t = *reinterpret_cast<const v4*>(mine + k * 1024 + lane * 16);
            __builtin_nontemporal_store(t, gdst + k * 64 + lane);
        }
        wave_lds_sync();
    }
}

template <int MATH, int OUT, bool PT = false>
__global__ __launch_bounds__(AT) void stain_apply_wide_kernel(const uint8_t* __restrict__ img, long hw,
                                                               const tia_stain_tables* __restrict__ tab,
                                                               const double* __restrict__ stats, StainT tgt,
                                                               void* __restrict__ out_v) {
    constexpr int TS = sizeof(typename Out<OUT>::T);
    __shared__ __attribute__((aligned(16))) uint8_t stage[AT / 64][3072 * TS];
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    uint8_t* mine = stage[threadIdx.x >> 6];
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    uint8_t* dst = reinterpret_cast<uint8_t*>(out_v) + (size_t)patch * (size_t)hw * 3u * TS;
    if constexpr (MATH == TIA_MATH_F32) {
        __shared__ float lut[256 * REP];
        ApplyCtx<TIA_MATH_F32> ctx;
        for (int i = threadIdx.x; i < 256 * REP; i += AT) lut[i] = tab->od_lut_f32[i / REP];
        const double nl2e = -1.4426950408889634;
#pragma unroll
        for (int i = 0; i < 9; ++i) ctx.m[i] = (float)(st[TIA_ST_M + i] * nl2e);
        ctx.lut = lut;
        ctx.bank = threadIdx.x & (REP - 1);
        __syncthreads();
        sweep_wide<OUT, ApplyCtx<TIA_MATH_F32>, float>(ctx, mine, src, dst, hw);
    } else if constexpr (PT) {
        __shared__ double ptab[9 * 256];
        __shared__ double lut[256];  // (only the libm fall-back reads it)
        bool ok = true;  // every |LUT[v] m| < 5.5414 * 126 < 700: far inside exp's range; a NaN / inf entry fails the comparison
#pragma unroll
        for (int k = 0; k < 9; ++k) ok = ok && fabs(st[TIA_ST_M + k]) < 126.0;
        for (int i = threadIdx.x; i < 256; i += AT) lut[i] = tab->od_lut[i];
        if (ok) {
            for (int i = threadIdx.x; i < 9 * 256; i += AT) {
                const int k = i >> 8, v = i & 255;
                const double e = exp(-(tab->od_lut[v] * st[TIA_ST_M + k]));
                ptab[i] = k < 3 ? 255.0 * e : e;
            }
        }
        __syncthreads();
        if (ok) {
            ApplyCtxTab ctx{ptab};
            sweep_wide<OUT, ApplyCtxTab, double>(ctx, mine, src, dst, hw);
        } else {
            ApplyCtx<TIA_MATH_F64_REF> ctx;
            load_ref_ctx(ctx, st, tgt, lut);
            sweep12<OUT, 1, ApplyCtx<TIA_MATH_F64_REF>, double>(ctx, src, reinterpret_cast<typename Out<OUT>::T*>(dst), hw);
        }
    } else {
        __shared__ double lut[256];
        __shared__ double etab[1024];
        for (int i = threadIdx.x; i < 256; i += AT) lut[i] = tab->od_lut[i];
        for (int i = threadIdx.x; i < 1024; i += AT) etab[i] = kExp2Tab255[i];
        ApplyCtxFast fast;
        const bool ok = fast.load(st);
        fast.lut = lut;
        fast.etab = etab;
        __syncthreads();
        if (ok) {
            sweep_wide<OUT, ApplyCtxFast, double>(fast, mine, src, dst, hw);
        } else {
            ApplyCtx<TIA_MATH_F64_REF> ctx;
            load_ref_ctx(ctx, st, tgt, lut);
            sweep12<OUT, 1, ApplyCtx<TIA_MATH_F64_REF>, double>(ctx, src, reinterpret_cast<typename Out<OUT>::T*>(dst), hw);
        }
    }
}

// ---- augment, f32 / 16-byte-access variant -------------------------------------------------------
// Same data movement as stain_apply_wide_kernel; per pixel: concentrations (6 FMA), tissue test on the
// folded luminance tables, alpha/beta on the selected pixels, recomposition with the source stain matrix
// pre-scaled by -log2(e), hardware exp2.  uint8 output (tools/stainaugment.py:177-206).
struct AugCtx {
    float p[6], s[6], al[2], be[2];
    const float* lut;
    const int (*ty)[256];
    int bank, y_thr, augment_background;
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, float (&o)[3]) const {
        const float x = lut[r * REP + bank], y = lut[g * REP + bank], z = lut[b * REP + bank];
        const int t = ty[0][r] + ty[1][g] + ty[2][b];
        const bool sel = augment_background || (((t + (1 << 11)) >> 12) < y_thr);
        float c0 = __builtin_fmaf(z, p[4], __builtin_fmaf(y, p[2], x * p[0]));
        float c1 = __builtin_fmaf(z, p[5], __builtin_fmaf(y, p[3], x * p[1]));
        c0 = __builtin_fmaf(c0, sel ? al[0] : 1.0f, sel ? be[0] : 0.0f);
        c1 = __builtin_fmaf(c1, sel ? al[1] : 1.0f, sel ? be[1] : 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = 255.0f * __builtin_amdgcn_exp2f(__builtin_fmaf(c1, s[3 + c], c0 * s[c]));
            v = v > 255.0f ? 255.0f : v;
            o[c] = v < 0.0f ? 0.0f : v;
        }
    }
};

__global__ __launch_bounds__(AT) void stain_augment_wide_kernel(const uint8_t* __restrict__ img, long hw,
                                                                 const tia_stain_tables* __restrict__ tab,
                                                                 const double* __restrict__ stats,
                                                                 const double* __restrict__ alpha_beta, int y_thr,
                                                                 int augment_background, int z1,
                                                                 uint8_t* __restrict__ out) {
    constexpr int CHUNK = 3072;
    __shared__ float lut[256 * REP];
    __shared__ int ty[3][256];
    __shared__ __attribute__((aligned(16))) uint8_t stage[AT / 64][CHUNK];
    using v4 = __attribute__((ext_vector_type(4))) unsigned;
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    for (int i = threadIdx.x; i < 256 * REP; i += AT) lut[i] = tab->od_lut_f32[i / REP];
    build_ty(ty, tab, st[TIA_ST_PLOW], st[TIA_ST_PHIGH], z1 != 0);
    AugCtx ctx;
    const double nl2e = -1.4426950408889634;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        ctx.p[i] = (float)st[TIA_ST_PINV + i];
        ctx.s[i] = (float)(st[TIA_ST_STAIN + i] * nl2e);
    }
    ctx.al[0] = (float)alpha_beta[patch * 4 + 0];
    ctx.al[1] = (float)alpha_beta[patch * 4 + 1];
    ctx.be[0] = (float)alpha_beta[patch * 4 + 2];
    ctx.be[1] = (float)alpha_beta[patch * 4 + 3];
    ctx.lut = lut;
    ctx.ty = ty;
    ctx.bank = threadIdx.x & (REP - 1);
    ctx.y_thr = y_thr;
    ctx.augment_background = augment_background;
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* mine = stage[wv];
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    uint8_t* dst = out + (size_t)patch * (size_t)hw * 3u;
    const long nchunks = hw * 3 / CHUNK;
    const long wstride = (long)gridDim.x * (AT / 64);
    for (long c = (long)blockIdx.x * (AT / 64) + wv; c < nchunks; c += wstride) {
        const v4* gsrc = reinterpret_cast<const v4*>(src + c * CHUNK);
        v4 in[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) in[k] = __builtin_nontemporal_load(gsrc + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<v4*>(mine + k * 1024 + lane * 16) = in[k];
        wave_lds_sync();
        uint32_t w[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const v4 t = *reinterpret_cast<const v4*>(mine + lane * 48 + j * 16);
            w[j * 4 + 0] = t.x;
            w[j * 4 + 1] = t.y;
            w[j * 4 + 2] = t.z;
            w[j * 4 + 3] = t.w;
        }
        wave_lds_sync();
        alignas(16) uint8_t res[48];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t a = w[q * 3], b = w[q * 3 + 1], cc = w[q * 3 + 2];
            float o[4][3];
            ctx.pixel(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, o[0]);
            ctx.pixel(a >> 24, b & 255u, (b >> 8) & 255u, o[1]);
            ctx.pixel((b >> 16) & 255u, b >> 24, cc & 255u, o[2]);
            ctx.pixel((cc >> 8) & 255u, (cc >> 16) & 255u, cc >> 24, o[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) res[q * 12 + i * 3 + ch] = (uint8_t)(unsigned)o[i][ch];
        }
        const v4* rv = reinterpret_cast<const v4*>(res);
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<v4*>(mine + lane * 48 + j * 16) = rv[j];
        wave_lds_sync();
        v4* gdst = reinterpret_cast<v4*>(dst + c * CHUNK);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const v4 t = *reinterpret_cast<const v4*>(mine + k * 1024 + lane * 16);
            __builtin_nontemporal_store(t, gdst + k * 64 + lane);
        }
        wave_lds_sync();
    }
}

// ---- augment, float64 / product-of-tables / 16-byte accesses -----------------------------------------------------------------------
// The output optical density of StainAugmentor.augment is affine in the three input optical densities, separately for the pixels the
// augmentation selects (tissue, or all with augment_background) and for the others:
//   selected:  OD'_c = sum_j LUT[v_j] (pinv[j][0] a0 S[0][c] + pinv[j][1] a1 S[1][c]) + (b0 S[0][c] + b1 S[1][c])
//   others:    OD'_c = sum_j LUT[v_j] (pinv[j][0] S[0][c] + pinv[j][1] S[1][c])
// so 255 exp(-OD'_c) is a product of three table entries (the constant folded into the first): two sets of nine 256-entry float64
// tables per patch (4,608 libm `exp` per workgroup instead of three per pixel -- the per-pixel libm form is bound by float64
// vector arithmetic at 16 % of the HBM roof), per pixel the tissue test (3 look-ups), 9 look-ups, 6 multiplies and the
// reference's clip.  Same error budget as the apply kernel's table form (< 1e-13 on the 0..255 scale); valid while every exponent
// stays far inside exp's range, else the caller launches the libm kernel.
struct AugCtxTab {
    const double* pt;  // [2][9][256]: set 0 = not selected, set 1 = selected; pt[(set * 9 + 3 j + c) * 256 + v]
    const int (*ty)[256];
    long bound;
    int augment_background;
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, double (&o)[3]) const {
        const long t = (long)ty[0][r] + ty[1][g] + ty[2][b];
        // (the set is an INDEX offset, not a pointer: with a selected 64-bit pointer per pixel the address arithmetic of the 36 look-ups
        // of a group is 64-bit and the kernel needs more than 256 registers -- one wave per SIMD for a loop that lives on LDS latency)
        // (`t` is looked up whether or not the background is augmented too: a uniform branch around the three look-ups splits the
        // group into basic blocks, the products sink to the end of the chunk and all sixteen pixels' 144 table reads stay live)
        const unsigned set = ((t < bound) | (augment_background != 0)) ? 9u * 256u : 0u;
        const unsigned ir = set + r, ig = set + g, ib = set + b;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = __builtin_fmin(pt[c * 256 + ir] * pt[(3 + c) * 256 + ig] * pt[(6 + c) * 256 + ib], 255.0);
    }
};

// the reference's per-pixel float64 arithmetic (stainaugment.py:177-206), libm exp: the table form's fall-back
struct AugCtxRef {
    double p[6], sm[6], al[2], be[2];
    const double* lut;
    const int (*ty)[256];
    long bound;
    int augment_background;
    __device__ __forceinline__ void pixel(uint32_t r, uint32_t g, uint32_t b, double (&o)[3]) const {
        const double x = lut[r], y = lut[g], z = lut[b];
        const long t = (long)ty[0][r] + ty[1][g] + ty[2][b];
        double c0 = x * p[0] + y * p[2] + z * p[4];
        double c1 = x * p[1] + y * p[3] + z * p[5];
        if (augment_background || t < bound) {
            c0 *= al[0];
            c0 += be[0];
            c1 *= al[1];
            c1 += be[1];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double tt = c0 * sm[c] + c1 * sm[3 + c];
            double v = 255.0 * exp(-1.0 * tt);
            o[c] = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);  // np.clip(., 0, 255)
        }
    }
};

// The patch's fused matrices of the table form (m[set][3 j + c]: tissue / background set), the constant exponent k0, and whether every
// factor stays within e^+-100 (otherwise, or for NaN / inf, the patch takes the reference's per-pixel arithmetic).
__device__ __forceinline__ bool augment_tables_ok(const double* __restrict__ st, const double* __restrict__ alpha_beta, long patch,
                                                  double (&m)[2][9], double (&k0)[3]) {
    const double a0 = alpha_beta[patch * 4 + 0], a1 = alpha_beta[patch * 4 + 1], b0 = alpha_beta[patch * 4 + 2], b1 = alpha_beta[patch * 4 + 3];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double s0 = st[TIA_ST_STAIN + c], s1 = st[TIA_ST_STAIN + 3 + c];
        k0[c] = b0 * s0 + b1 * s1;
        ok = ok && fabs(k0[c]) < 600.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double p0 = st[TIA_ST_PINV + 2 * j], p1 = st[TIA_ST_PINV + 2 * j + 1];
            m[0][3 * j + c] = p0 * s0 + p1 * s1;
            m[1][3 * j + c] = p0 * a0 * s0 + p1 * a1 * s1;
            ok = ok && fabs(m[0][3 * j + c]) < 18.0 && fabs(m[1][3 * j + c]) < 18.0;  // 5.5414 * 18 < 100: every factor within e^+-100
        }
    }
    return ok;
}

// Two kernels, launched back to back over the same grid: the table form for the patches whose exponents are safe, the reference's
// per-pixel libm arithmetic for the rest (a workgroup of the other kind returns at once).  As ONE kernel the libm path set the register
// allocation of both: 256 VGPRs + 160 bytes of scratch, one wave per SIMD for a loop that lives on LDS look-up latency.
__global__ __launch_bounds__(AT) void stain_augment_f64_wide_kernel(const uint8_t* __restrict__ img, long hw,
                                                                     const tia_stain_tables* __restrict__ tab,
                                                                     const double* __restrict__ stats,
                                                                     const double* __restrict__ alpha_beta, int y_thr,
                                                                     int augment_background, int z1, uint8_t* __restrict__ out) {
    __shared__ double ptab[2 * 9 * 256];
    __shared__ int ty[3][256];
    __shared__ __attribute__((aligned(16))) uint8_t stage[AT / 64][3072];
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    double m[2][9], k0[3];
    if (!augment_tables_ok(st, alpha_beta, patch, m, k0)) return;  // (uniform per workgroup) -> stain_augment_f64_ref_wide_kernel
    build_ty(ty, tab, st[TIA_ST_PLOW], st[TIA_ST_PHIGH], z1 != 0);
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    uint8_t* dst = out + (size_t)patch * (size_t)hw * 3u;
    // The 4,608 entries in a ROLLED loop (one exp call site: unrolled, eighteen inlined libm exponentials push the kernel past 256
    // registers and its main loop to one wave per SIMD), the matrices read from LDS (indexed by a run-time (set, kk) in registers they
    // would live in scratch memory).
    __shared__ double s_m[2 * 9], s_k[3];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 18; ++i) s_m[i] = m[i / 9][i % 9];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_k[c] = 255.0 * exp(-k0[c]);
    }
    __syncthreads();
#pragma unroll 1
    for (int i = threadIdx.x; i < 2 * 9 * 256; i += AT) {
        const int sk = i >> 8, kk = sk >= 9 ? sk - 9 : sk, v = i & 255;
        const double scale = kk < 3 ? (sk >= 9 ? s_k[kk] : 255.0) : 1.0;
        ptab[i] = exp(-(tab->od_lut[v] * s_m[sk])) * scale;
    }
    __syncthreads();
    AugCtxTab ctx{ptab, ty, ((long)y_thr << 12) - (1 << 11), augment_background};
    sweep_wide<TIA_OUT_U8, AugCtxTab, double>(ctx, stage[threadIdx.x >> 6], src, dst, hw);
}

__global__ __launch_bounds__(AT) void stain_augment_f64_ref_wide_kernel(const uint8_t* __restrict__ img, long hw,
                                                                         const tia_stain_tables* __restrict__ tab,
                                                                         const double* __restrict__ stats,
                                                                         const double* __restrict__ alpha_beta, int y_thr,
                                                                         int augment_background, int z1, uint8_t* __restrict__ out) {
    __shared__ double lut[256];
    __shared__ int ty[3][256];
    __shared__ __attribute__((aligned(16))) uint8_t stage[AT / 64][3072];
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    double m[2][9], k0[3];
    if (augment_tables_ok(st, alpha_beta, patch, m, k0)) return;  // (uniform per workgroup) the table form has done this patch
    build_ty(ty, tab, st[TIA_ST_PLOW], st[TIA_ST_PHIGH], z1 != 0);
    for (int i = threadIdx.x; i < 256; i += AT) lut[i] = tab->od_lut[i];
    __syncthreads();
    AugCtxRef ref;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        ref.p[i] = st[TIA_ST_PINV + i];
        ref.sm[i] = st[TIA_ST_STAIN + i];
    }
    ref.al[0] = alpha_beta[patch * 4 + 0];
    ref.al[1] = alpha_beta[patch * 4 + 1];
    ref.be[0] = alpha_beta[patch * 4 + 2];
    ref.be[1] = alpha_beta[patch * 4 + 3];
    ref.lut = lut;
    ref.ty = ty;
    ref.bound = ((long)y_thr << 12) - (1 << 11);
    ref.augment_background = augment_background;
    sweep_wide<TIA_OUT_U8, AugCtxRef, double>(ref, stage[threadIdx.x >> 6], img + (size_t)patch * (size_t)hw * 3u,
                                             out + (size_t)patch * (size_t)hw * 3u, hw);
}

// ---- concentrations ------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void stain_conc_kernel(const uint8_t* __restrict__ img, long hw,
                                                         const tia_stain_tables* __restrict__ tab,
                                                         const double* __restrict__ stats,
                                                         double* __restrict__ conc) {
    __shared__ double lut[256];
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    for (int i = threadIdx.x; i < 256; i += AT) lut[i] = tab->od_lut[i];
    double p[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) p[i] = st[TIA_ST_PINV + i];
    __syncthreads();
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    double2* dst = reinterpret_cast<double2*>(conc) + (size_t)patch * (size_t)hw;
    for (long i = (long)blockIdx.x * AT + threadIdx.x; i < hw; i += (long)gridDim.x * AT) {
        const double x = lut[src[3 * i]], y = lut[src[3 * i + 1]], z = lut[src[3 * i + 2]];
        double2 c;
        c.x = x * p[0] + y * p[2] + z * p[4];
        c.y = x * p[1] + y * p[3] + z * p[5];
        dst[i] = c;
    }
}

// ---- augment + luminosity mask ----------------------------------------------------------------------
template <bool MASK_ONLY>
__global__ __launch_bounds__(AT) void stain_augment_kernel(const uint8_t* __restrict__ img, long hw,
                                                            const tia_stain_tables* __restrict__ tab,
                                                            const double* __restrict__ stats,
                                                            const double* __restrict__ alpha_beta,
                                                            int y_thr, int augment_background, int z1,
                                                            uint8_t* __restrict__ out) {
    __shared__ double lut[256];
    __shared__ int ty[3][256];
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    for (int i = threadIdx.x; i < 256; i += AT) lut[i] = tab->od_lut[i];
    build_ty(ty, tab, st[TIA_ST_PLOW], st[TIA_ST_PHIGH], z1 != 0);
    double p[6], sm[6], al[2] = {1.0, 1.0}, be[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        p[i] = st[TIA_ST_PINV + i];
        sm[i] = st[TIA_ST_STAIN + i];
    }
    if (!MASK_ONLY) {
        al[0] = alpha_beta[patch * 4 + 0];
        al[1] = alpha_beta[patch * 4 + 1];
        be[0] = alpha_beta[patch * 4 + 2];
        be[1] = alpha_beta[patch * 4 + 3];
    }
    __syncthreads();
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    if constexpr (MASK_ONLY) {
        // 4 pixels per lane: three dword loads in, one dword of mask bytes out
        if ((hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(out)) & 3) == 0) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(src);
            uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)patch * (size_t)hw);
            auto tis = [&](uint32_t r, uint32_t g, uint32_t b) -> uint32_t {
                return (((ty[0][r] + ty[1][g] + ty[2][b] + (1 << 11)) >> 12) < y_thr) ? 1u : 0u;
            };
            for (long g4 = (long)blockIdx.x * AT + threadIdx.x; g4 < (hw >> 2); g4 += (long)gridDim.x * AT) {
                const uint32_t a = q[g4 * 3], b = q[g4 * 3 + 1], c = q[g4 * 3 + 2];
                o[g4] = tis(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u) | (tis(a >> 24, b & 255u, (b >> 8) & 255u) << 8) |
                        (tis((b >> 16) & 255u, b >> 24, c & 255u) << 16) | (tis((c >> 8) & 255u, (c >> 16) & 255u, c >> 24) << 24);
            }
            return;
        }
    }
    for (long i = (long)blockIdx.x * AT + threadIdx.x; i < hw; i += (long)gridDim.x * AT) {
        const uint32_t r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
        const int t = ty[0][r] + ty[1][g] + ty[2][b];
        const bool tissue = ((t + (1 << 11)) >> 12) < y_thr;
        if (MASK_ONLY) {
            out[(size_t)patch * (size_t)hw + i] = tissue ? 1 : 0;
        } else {
            const double x = lut[r], y = lut[g], z = lut[b];
            double c0 = x * p[0] + y * p[2] + z * p[4];
            double c1 = x * p[1] + y * p[3] + z * p[5];
            if (tissue || augment_background) {
                c0 *= al[0];
                c0 += be[0];
                c1 *= al[1];
                c1 += be[1];
            }
            uint8_t* d = out + ((size_t)patch * (size_t)hw + i) * 3u;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double tt = c0 * sm[c] + c1 * sm[3 + c];
                double v = 255.0 * exp(-1.0 * tt);
                v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);  // np.clip(., 0, 255)
                d[c] = (uint8_t)(unsigned)v;
            }
        }
    }
}

// ---- luminosity mask, 16-byte accesses ------------------------------------------------------------------------------------
// 48 bytes of RGB in / 16 mask bytes out per lane and step (wide_io.hpp); per pixel three table look-ups (the folded
// contrast-stretch + Y-row tables of build_ty) and one comparison: tissue <=> ((t + 2^11) >> 12) < y_thr <=> t < (y_thr << 12) - 2^11
// for the non-negative integer sum t.  Needs h*w % 1024 == 0 and 16-byte aligned buffers (the launcher checks).
__global__ __launch_bounds__(AT) void luminosity_mask_wide_kernel(const uint8_t* __restrict__ img, long hw,
                                                                   const tia_stain_tables* __restrict__ tab,
                                                                   const double* __restrict__ stats, int y_thr, int z1,
                                                                   uint8_t* __restrict__ out) {
    __shared__ int ty[3][256];
    __shared__ __attribute__((aligned(16))) uint8_t stage[AT / 64][kRgbChunk];
    const long patch = blockIdx.y;
    const double* st = stats + patch * TIA_STATS_STRIDE;
    build_ty(ty, tab, st[TIA_ST_PLOW], st[TIA_ST_PHIGH], z1 != 0);
    __syncthreads();
    const long bound = ((long)y_thr << 12) - (1 << 11);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
    uint8_t* dst = out + (size_t)patch * (size_t)hw;
    const long nchunks = hw / kPxChunk, wstride = (long)gridDim.x * (AT / 64);
    RgbChunk cur, nxt;
    long c = (long)blockIdx.x * (AT / 64) + wv;
    if (c < nchunks) rgb_chunk_issue(cur, src + c * kRgbChunk);
    for (; c < nchunks; c += wstride) {
        const long cn = c + wstride;
        if (cn < nchunks) rgb_chunk_issue(nxt, src + cn * kRgbChunk);
        uint32_t w[12];
        rgb_chunk_transpose(cur, stage[wv], w);
        v4u m;
        uint32_t* mo = reinterpret_cast<uint32_t*>(&m);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t p[4];
            group_pixels(w[3 * q], w[3 * q + 1], w[3 * q + 2], p);
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long t = (long)ty[0][p[i] & 255u] + ty[1][(p[i] >> 8) & 255u] + ty[2][(p[i] >> 16) & 255u];
                bits |= (t < bound ? 1u : 0u) << (8 * i);
            }
            mo[q] = bits;
        }
        __builtin_nontemporal_store(m, reinterpret_cast<v4u*>(dst + c * kPxChunk) + lane);
        cur = nxt;
    }
}

// ---- stand-alone rgb2od (utils/transforms.py:209-231) ----------------------------------------------
// One dword (4 bytes) per lane and step: 4 table look-ups, two 16-byte stores (a lane's 32 output bytes are
// contiguous, a wave's 2 KB too).  `mutate` reproduces the reference's side effect `img[img == 0] = 1`
// (the dword is only written back when it held a zero byte).  Output-bound: 8 bytes out per byte in.
__global__ __launch_bounds__(AT) void rgb2od_kernel(uint8_t* __restrict__ img, long nbytes,
                                                     const tia_stain_tables* __restrict__ tab, int mutate,
                                                     double* __restrict__ od, int vec_ok) {
    __shared__ double lut[256];
    for (int i = threadIdx.x; i < 256; i += AT) lut[i] = tab->od_lut[i];
    __syncthreads();
    const long stride = (long)gridDim.x * AT;
    long done = 0;
    if (vec_ok) {
        const long nd = nbytes >> 2;
        uint32_t* q = reinterpret_cast<uint32_t*>(img);
        double2* o = reinterpret_cast<double2*>(od);
        for (long d = (long)blockIdx.x * AT + threadIdx.x; d < nd; d += stride) {
            const uint32_t w = q[d];
            double2 lo, hi;
            lo.x = lut[w & 255u];
            lo.y = lut[(w >> 8) & 255u];
            hi.x = lut[(w >> 16) & 255u];
            hi.y = lut[w >> 24];
            __builtin_nontemporal_store(lo.x, &o[2 * d].x);
            __builtin_nontemporal_store(lo.y, &o[2 * d].y);
            __builtin_nontemporal_store(hi.x, &o[2 * d + 1].x);
            __builtin_nontemporal_store(hi.y, &o[2 * d + 1].y);
            if (mutate) {
                // a byte is zero <=> (w - 0x01010101) & ~w & 0x80808080 has its top bit set (exact per byte only
                // below the first zero byte, so build the replacement byte by byte)
                uint32_t z = 0;
                if ((w & 0x000000ffu) == 0) z |= 0x00000001u;
                if ((w & 0x0000ff00u) == 0) z |= 0x00000100u;
                if ((w & 0x00ff0000u) == 0) z |= 0x00010000u;
                if ((w & 0xff000000u) == 0) z |= 0x01000000u;
                if (z) q[d] = w | z;
            }
        }
        done = nd << 2;
    }
    for (long i = done + (long)blockIdx.x * AT + threadIdx.x; i < nbytes; i += stride) {
        const uint32_t v = img[i];
        od[i] = lut[v];
        if (mutate && v == 0) img[i] = 1;
    }
}

static inline unsigned blocks_x(long work_items, long n_patches) {
    // enough workgroups to fill 256 CUs several times over, but few per patch when the batch is
    // large so the per-block table set-up is amortised.
    long per_block = (long)AT * 4;
    long maxb = (work_items + per_block - 1) / per_block;
    static long target = 0;
    if (target == 0) {
        const char* e = tia::dev_env("TIA_APPLY_BLOCKS");
        target = e ? atol(e) : 4096;
    }
    long want = (target + n_patches - 1) / n_patches;
    long b = want < maxb ? want : maxb;
    return (unsigned)(b < 1 ? 1 : b);
}

template <int MATH>
static int launch_apply(const uint8_t* d_img, int64_t n, long hw, const tia_stain_tables* tab,
                        const double* stats, const StainT& tgt, void* out, int out_kind,
                        hipStream_t stream) {
    const long ng = (hw & 3) == 0 ? (hw >> 2) : hw;
    dim3 grid(blocks_x(ng, n), (unsigned)n);
    if constexpr (MATH != TIA_MATH_F64_REF) {
        static const bool no_wide = tia::dev_env("TIA_APPLY_NO_WIDE") != nullptr;  // developer switch (A/B measurements)
        const bool aligned = ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
        if ((hw * 3) % 3072 == 0 && aligned && !(no_wide && MATH == TIA_MATH_F64) &&
            (out_kind == TIA_OUT_U8 || out_kind == TIA_OUT_UNIT_F16 || out_kind == TIA_OUT_UNIT_BF16)) {
            const long nchunks = hw * 3 / 3072;
            long bx = (nchunks + 3) / 4;                       // one step per wave ...
            const long want = (4096 + (long)n - 1) / (long)n;  // ... unless the batch already fills the chip
            if (bx > want) bx = want;
            dim3 wgrid((unsigned)(bx < 1 ? 1 : bx), (unsigned)n);
            if constexpr (MATH == TIA_MATH_F64) {
                // float64: the product-of-tables form (TIA_APPLY_NO_PTAB keeps the exponent-trick form: developer switch, A/B measurements)
                static const bool no_ptab = tia::dev_env("TIA_APPLY_NO_PTAB") != nullptr;
                if (!no_ptab) {
                    if (out_kind == TIA_OUT_U8)
                        hipLaunchKernelGGL((stain_apply_wide_kernel<MATH, TIA_OUT_U8, true>), wgrid, dim3(AT), 0, stream, d_img, hw, tab, stats, tgt, out);
                    else if (out_kind == TIA_OUT_UNIT_F16)
                        hipLaunchKernelGGL((stain_apply_wide_kernel<MATH, TIA_OUT_UNIT_F16, true>), wgrid, dim3(AT), 0, stream, d_img, hw, tab, stats, tgt, out);
                    else
                        hipLaunchKernelGGL((stain_apply_wide_kernel<MATH, TIA_OUT_UNIT_BF16, true>), wgrid, dim3(AT), 0, stream, d_img, hw, tab, stats, tgt, out);
                    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
                }
            }
            if (out_kind == TIA_OUT_U8)
                hipLaunchKernelGGL((stain_apply_wide_kernel<MATH, TIA_OUT_U8>), wgrid, dim3(AT), 0, stream, d_img, hw, tab, stats, tgt, out);
            else if (out_kind == TIA_OUT_UNIT_F16)
                hipLaunchKernelGGL((stain_apply_wide_kernel<MATH, TIA_OUT_UNIT_F16>), wgrid, dim3(AT), 0, stream, d_img, hw, tab, stats, tgt, out);
            else
                hipLaunchKernelGGL((stain_apply_wide_kernel<MATH, TIA_OUT_UNIT_BF16>), wgrid, dim3(AT), 0, stream, d_img, hw, tab, stats, tgt, out);
            return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
        }
    }
#define TIA_LAUNCH(OUTK)                                                                              \
    case OUTK:                                                                                        \
        hipLaunchKernelGGL((stain_apply_kernel<MATH, OUTK>), grid, dim3(AT), 0, stream, d_img, hw,   \
                           tab, stats, tgt, out);                                                     \
        break;
    switch (out_kind) {
        TIA_LAUNCH(TIA_OUT_U8)
        TIA_LAUNCH(TIA_OUT_F32)
        TIA_LAUNCH(TIA_OUT_F64)
        TIA_LAUNCH(TIA_OUT_UNIT_F16)
        TIA_LAUNCH(TIA_OUT_UNIT_BF16)
        TIA_LAUNCH(TIA_OUT_UNIT_F32)
        default:
            return TIA_EINVAL;
    }
#undef TIA_LAUNCH
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia

static bool bad_dims(int64_t n, int64_t h, int64_t w) {
    return n <= 0 || h <= 0 || w <= 0 || n > 65535;
}

extern "C" int tia_abi_version(void) { return TIA_ABI_VERSION; }

extern "C" int tia_stain_apply_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                   const tia_stain_tables* d_tables, const double* d_stats,
                                   const double* h_target_stain, void* d_out, int32_t out_kind,
                                   int32_t math, void* stream) {
    if (!d_img || !d_tables || !d_stats || !d_out) return TIA_EINVAL;
    if (bad_dims(n, h, w)) return n > 65535 ? TIA_ESIZE : TIA_EINVAL;
    tia::StainT tgt{};
    if (math == TIA_MATH_F64 || math == TIA_MATH_F64_REF) {
        if (!h_target_stain) return TIA_EINVAL;
        for (int i = 0; i < 6; ++i) tgt.s[i] = h_target_stain[i];
        if (math == TIA_MATH_F64_REF)
            return tia::launch_apply<TIA_MATH_F64_REF>(d_img, n, (long)h * w, d_tables, d_stats, tgt, d_out, out_kind,
                                                       (hipStream_t)stream);
        return tia::launch_apply<TIA_MATH_F64>(d_img, n, (long)h * w, d_tables, d_stats, tgt, d_out,
                                               out_kind, (hipStream_t)stream);
    }
    if (math == TIA_MATH_F32)
        return tia::launch_apply<TIA_MATH_F32>(d_img, n, (long)h * w, d_tables, d_stats, tgt, d_out,
                                               out_kind, (hipStream_t)stream);
    return TIA_EINVAL;
}

extern "C" int tia_stain_concentrations_f64(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                             const tia_stain_tables* d_tables, const double* d_stats,
                                             double* d_conc, void* stream) {
    if (!d_img || !d_tables || !d_stats || !d_conc) return TIA_EINVAL;
    if (bad_dims(n, h, w)) return n > 65535 ? TIA_ESIZE : TIA_EINVAL;
    const long hw = (long)h * w;
    dim3 grid(tia::blocks_x(hw, n), (unsigned)n);
    hipLaunchKernelGGL(tia::stain_conc_kernel, grid, dim3(tia::AT), 0, (hipStream_t)stream, d_img, hw,
                       d_tables, d_stats, d_conc);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_stain_augment_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                     const tia_stain_tables* d_tables, const double* d_stats,
                                     const double* d_alpha_beta, int32_t y_thr,
                                     int32_t augment_background, int32_t zero_to_one, uint8_t* d_out,
                                     int32_t math, void* stream) {
    if (!d_img || !d_tables || !d_stats || !d_alpha_beta || !d_out) return TIA_EINVAL;
    if (math != TIA_MATH_F64 && math != TIA_MATH_F32) return TIA_EINVAL;
    if (bad_dims(n, h, w)) return n > 65535 ? TIA_ESIZE : TIA_EINVAL;
    const long hw = (long)h * w;
    if (math == TIA_MATH_F32) {
        // the f32 path exists only in the 16-byte-access form: whole 3072-byte chunks, 16-byte aligned buffers
        const bool aligned = ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
        if ((hw * 3) % 3072 != 0 || !aligned) return TIA_ESIZE;
        const long nchunks = hw * 3 / 3072;
        long bx = (nchunks + 3) / 4;
        const long want = (4096 + (long)n - 1) / (long)n;
        if (bx > want) bx = want;
        hipLaunchKernelGGL(tia::stain_augment_wide_kernel, dim3((unsigned)(bx < 1 ? 1 : bx), (unsigned)n), dim3(tia::AT), 0,
                           (hipStream_t)stream, d_img, hw, d_tables, d_stats, d_alpha_beta, y_thr, augment_background,
                           zero_to_one, d_out);
        return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
    }
    {
        // float64: the product-of-tables / 16-byte-access form where the shape allows (whole 3072-byte chunks, aligned buffers);
        // the tables cost 4,608 exponentials per workgroup: one workgroup per patch unless the batch is small
        const bool aligned = ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
        static const bool no_tab = tia::dev_env("TIA_AUGMENT_NO_TABLES") != nullptr;  // developer switch: the per-pixel libm kernel
        if ((hw * 3) % 3072 == 0 && aligned && !no_tab) {
            const long nchunks = hw * 3 / 3072;
            long bx = (nchunks + 3) / 4;
            const long want = (1024 + (long)n - 1) / (long)n;
            if (bx > want) bx = want;
            const dim3 grid_w((unsigned)(bx < 1 ? 1 : bx), (unsigned)n);
            hipLaunchKernelGGL(tia::stain_augment_f64_wide_kernel, grid_w, dim3(tia::AT), 0, (hipStream_t)stream, d_img, hw, d_tables,
                               d_stats, d_alpha_beta, y_thr, augment_background, zero_to_one, d_out);
            hipLaunchKernelGGL(tia::stain_augment_f64_ref_wide_kernel, grid_w, dim3(tia::AT), 0, (hipStream_t)stream, d_img, hw, d_tables,
                               d_stats, d_alpha_beta, y_thr, augment_background, zero_to_one, d_out);
            return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
        }
    }
    dim3 grid(tia::blocks_x(hw, n), (unsigned)n);
    hipLaunchKernelGGL((tia::stain_augment_kernel<false>), grid, dim3(tia::AT), 0, (hipStream_t)stream,
                       d_img, hw, d_tables, d_stats, d_alpha_beta, y_thr, augment_background,
                       zero_to_one, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}


extern "C" int tia_luminosity_mask_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                       const tia_stain_tables* d_tables, const double* d_stats,
                                       int32_t y_thr, int32_t zero_to_one, uint8_t* d_mask,
                                       void* stream) {
    if (!d_img || !d_tables || !d_stats || !d_mask) return TIA_EINVAL;
    if (bad_dims(n, h, w)) return n > 65535 ? TIA_ESIZE : TIA_EINVAL;
    const long hw = (long)h * w;
    if (hw % tia::kPxChunk == 0 && ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_mask)) & 15) == 0) {
        // 16-byte accesses: a few steps per wave, enough workgroups to fill the chip
        long per = hw / tia::kPxChunk / 4;  // one step per wave
        long want = (4096 + n - 1) / n;
        long bx = per < want ? per : want;
        dim3 gridw((unsigned)(bx < 1 ? 1 : bx), (unsigned)n);
        hipLaunchKernelGGL(tia::luminosity_mask_wide_kernel, gridw, dim3(tia::AT), 0, (hipStream_t)stream, d_img, hw, d_tables,
                           d_stats, y_thr, zero_to_one, d_mask);
        return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
    }
    dim3 grid(tia::blocks_x(hw, n), (unsigned)n);
    hipLaunchKernelGGL((tia::stain_augment_kernel<true>), grid, dim3(tia::AT), 0, (hipStream_t)stream,
                       d_img, hw, d_tables, d_stats, (const double*)nullptr, y_thr, 0, zero_to_one,
                       d_mask);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_rgb2od_u8(uint8_t* d_img, int64_t nbytes, const tia_stain_tables* d_tables, int32_t mutate,
                             double* d_od, void* stream) {
    if (!d_img || !d_tables || !d_od || nbytes <= 0) return TIA_EINVAL;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(d_img) & 3) == 0 && (reinterpret_cast<uintptr_t>(d_od) & 15) == 0) ? 1 : 0;
    long blocks = (nbytes / 4 + tia::AT - 1) / tia::AT;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(tia::rgb2od_kernel, dim3((unsigned)blocks), dim3(tia::AT), 0, (hipStream_t)stream, d_img, (long)nbytes,
                       d_tables, mutate, d_od, vec_ok);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

// Clears (and returns) the HIP runtime's sticky last-error value of THIS library's runtime instance -- the one every entry
// point's `hipGetLastError() == hipSuccess` check reads.  A host-side call that was refused on purpose (hipHostRegister of
// memory that is already pinned) must not turn the next launch into a spurious TIA_ELAUNCH.
extern "C" int tia_clear_last_error(void) { return (int)hipGetLastError(); }
