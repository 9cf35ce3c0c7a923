// OpenCV 8-bit RGB<->Lab on gfx950 and the Reinhard normaliser built on it.  Reference: tools/stainnorm.py:222-367
// (cv2.cvtColor RGB2LAB / LAB2RGB, cv2.meanStdDev).
//
// What bounds these kernels is the VECTOR ALU, not HBM: the first-generation kernels spent ~110 VALU instructions per pixel (a
// dozen of them quarter-rate 32-bit multiplies and 64-bit multiply-adds of the Lab->XYZ cube), i.e. ~0.9 ms per 4096 x 224^2
// batch at 100 % VALU occupancy.  This version:
//   * RGB->Lab in float32 where float32 is EXACT: gamma-table values are integers <= 2040 and every coefficient row sums to 4096,
//     so R*c0 + G*c1 + B*c2 (+ rounding) < 2^24 -- the fused multiply-adds reproduce the integer sums bit for bit (coefficients
//     pre-divided by 4096, a power of two), and two pixels share one v_pk_fma_f32.  L, a, b = round-half-up of multiples of 2^-15
//     below 256: adding 2^-16 makes round-to-nearest-even equal to floor(x + 0.5) and v_cvt_pk_u8_f32 rounds, saturates to
//     [0, 255] and packs the byte in one instruction (semantics checked on the hardware: scripts/isa_probe.hip).
//   * Lab->RGB keeps integers but in 24-bit multiplies (full rate): every operand is below 2^23, the cube's intermediates fit 32
//     bits (i <= 26870: i*i < 2^30, (i*i >> 14) * i < 2^31), the linear branch (i <= 3390, near-black pixels) sits behind a wave-
//     uniform branch.
//   * the per-image part of Reinhard (the three 256-entry float32 chains) is folded with LabToYF_b / the a,b pre-scaling into
//     per-image tables: L -> (ify, y), a -> adiv, b -> bdiv.
//   * ReinhardNormalizer.transform is ONE launch for patches.  Up to 256 x 256 pixels a persistent 1024-thread workgroup keeps the
//     patch's Lab image in registers between the passes (`reinhard_resident_kernel`); above that (to 2^18 pixels) persistent
//     256-thread workgroups park it, one dword per pixel, in a per-workgroup scratch slot (`reinhard_fused_kernel`).  L is counted
//     in an LDS histogram (16 interleaved copies: bounded same-address serialisation whatever the image), the a / b moments are
//     exact integer sums (byte dot products), then moments -> tables -> the way back.  HBM sees the algorithmic 3 + 3 bytes per pixel.
//     Measured (4096 x 224^2): RGB->Lab + statistics 0.28 ms, the way back 0.2 ms of arithmetic + the store; the kernel issues ~63
//     VALU instructions and ~70 LDS cycles (half of them bank conflicts of the data-dependent look-ups) per 64 pixels: both units
//     run at ~50 %, the phases of one workgroup per CU do not overlap.
//   * all streams use 16-byte accesses through the wave-private LDS transpose of wide_io.hpp.
// Large single images (and shapes the wide path does not take) go through the three-launch form: histogram kernel over
// ceil(pixels / 64 Ki) workgroups per image -> moments / tables -> apply kernel, same arithmetic.
#include "common.hpp"
#include "wide_io.hpp"

#pragma clang fp contract(off)  // the LUT arithmetic restates NumPy float32/float64 expressions term by term

namespace tia {

constexpr int LT = 256;   // threads of the streaming kernels
constexpr int FT = 256;   // threads of the fused (one workgroup per patch) kernel: 53 KB of LDS, three workgroups per CU
typedef float f2 __attribute__((ext_vector_type(2)));
struct U3 {  // 12 bytes, 4-byte aligned (a 3-element ext_vector is padded to 16 bytes: pointer arithmetic on it would stride 16)
    uint32_t x, y, z;
};
static_assert(sizeof(U3) == 12, "12-byte lane accesses");

// ---- fixed tables in LDS ------------------------------------------------------------------------------------------------------
constexpr int kCbrtN = 2048;  // descale(R*c0 + G*c1 + B*c2, 12) <= 2040: gamma <= 2040, each coefficient row sums to 4096
constexpr int kGammaRep = 1;  // (experiment, kept at 1: pixel values of a wave CLUSTER, so the plain table -- bank = v mod 32 -- conflicts less than
                              // interleaved copies, whose bank is 8 (v mod 4) + copy: measured 64 M vs 78 M conflict cycles) interleaved copies of the gamma table (a lane reads copy lane & 7): a 32-lane LDS group spreads its
                              // 4 lanes per copy over 4 banks instead of 32 lanes over 32 -- 0.9 instead of 2.5 expected conflict cycles
struct LabFwdLds {
    float gamma[256 * kGammaRep];  // sRGBGammaTab_b, entry v of copy s at v * 8 + s
    float cbrt[kCbrtN];            // LabCbrtTab_b[0 .. 2047]
};
struct LabInvLds {
    uint8_t inv_gamma[4096];  // sRGBInvGammaTab_b
};
struct LabImgLds {  // per-image tables of the way back: Lab byte -> what Lab2RGBinteger needs
    int4 ly[256];   // of the (table-mapped) L byte: ify, and y's three matrix terms c1*y + 2^13, c4*y + 2^13, c7*y + 2^13
    int ta[256];    // adiv of the (table-mapped) a byte
    int tb[256];    // bdiv of the (table-mapped) b byte
};
struct LabCoef {
    float f[9];  // forward coefficients / 4096
    int i[9];    // inverse coefficients
};

__device__ __forceinline__ void load_fwd(LabFwdLds& s, const tia_lab_tables* __restrict__ t) {
    for (int i = threadIdx.x; i < 256 * kGammaRep; i += blockDim.x) s.gamma[i] = (float)t->gamma[i / kGammaRep];
    for (int i = threadIdx.x; i < kCbrtN; i += blockDim.x) s.cbrt[i] = (float)t->cbrt[i];
}
__device__ __forceinline__ void load_inv(LabInvLds& s, const tia_lab_tables* __restrict__ t) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(t->inv_gamma);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s.inv_gamma);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ LabCoef load_coef(const tia_lab_tables* __restrict__ t) {
    LabCoef c;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        c.f[k] = (float)t->c_fwd[k] * (1.0f / 4096.0f);
        c.i[k] = t->c_inv[k];
    }
    return c;
}

// adiv / bdiv of Lab2RGBinteger for one byte
__device__ __forceinline__ int lab_adiv(int a) { return ((5 * a * 53687 + (1 << 7)) >> 13) - 128 * 16384 / 500; }
__device__ __forceinline__ int lab_bdiv(int b) { return ((b * 41943 + (1 << 4)) >> 9) - 128 * 16384 / 200 + 1; }

__device__ __forceinline__ void set_img_entry(LabImgLds& s, const tia_lab_tables* __restrict__ t, int v, int l, int a, int b) {
    const int y = (int)t->lab_y[l];
    s.ly[v] = make_int4((int)t->lab_ify[l], t->c_inv[1] * y + (1 << 13), t->c_inv[4] * y + (1 << 13), t->c_inv[7] * y + (1 << 13));
    s.ta[v] = lab_adiv(a);
    s.tb[v] = lab_bdiv(b);
}

// per-image tables from three byte->byte tables (nullptr: identity, the plain LAB2RGB conversion)
__device__ __forceinline__ void build_img_tables(LabImgLds& s, const tia_lab_tables* __restrict__ t, const uint8_t* lut) {
    for (int v = threadIdx.x; v < 256; v += blockDim.x) {
        const int l = lut ? lut[v] : v, a = lut ? lut[256 + v] : v, b = lut ? lut[512 + v] : v;
        set_img_entry(s, t, v, l, a, b);
    }
}

// ---- RGB -> Lab, four pixels at a time ---------------------------------------------------------------------------------------------
// p = pixel dword (r | g << 8 | b << 16 | anything << 24); result = L | a << 8 | b << 16.  The two table stages are issued as blocks
// (twelve look-ups in flight, then the arithmetic): left to itself the scheduler waits for each pixel's look-ups before it
// issues the next pixel's, and the kernel is bound by LDS round trips instead of LDS throughput.
#define TIA_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
// CR interleaved copies of the cube-root table (entry i of copy c at i * CR + c; `cbrt` already points at the lane's copy): its
// indices scatter over ~50 consecutive entries within a wave (1 % pixel noise on 11-bit values), i.e. over the 32 banks like random
// numbers -- 2.5 expected conflict cycles per 32-lane group with one table, 0.5 with 16 copies (each bank pair serves two lanes).
template <int CR>
__device__ __forceinline__ void lab_fwd4t(const float* gamma, const float* cbrt, const LabCoef& k, const uint32_t (&p)[4], uint32_t (&o)[4]) {
    const float* gam = gamma + (kGammaRep > 1 ? (threadIdx.x & (kGammaRep - 1)) : 0);
    float R[4], G[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        R[i] = gam[(p[i] & 255u) * kGammaRep];
        G[i] = gam[((p[i] >> 8) & 255u) * kGammaRep];
        B[i] = gam[((p[i] >> 16) & 255u) * kGammaRep];
    }
    TIA_STAGE_FENCE();
    const f2 half = {0.5f, 0.5f};
    float fx[4], fy[4], fz[4];
    unsigned ix[3][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2 r2 = {R[2 * h], R[2 * h + 1]}, g2 = {G[2 * h], G[2 * h + 1]}, b2 = {B[2 * h], B[2 * h + 1]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const f2 c0 = {k.f[3 * r], k.f[3 * r]}, c1 = {k.f[3 * r + 1], k.f[3 * r + 1]}, c2 = {k.f[3 * r + 2], k.f[3 * r + 2]};
            // exact: integer sums < 2^24 scaled by 2^-12, + 0.5; truncation = descale(., 12)
            const f2 v = __builtin_elementwise_fma(b2, c2, __builtin_elementwise_fma(g2, c1, __builtin_elementwise_fma(r2, c0, half)));
            ix[r][2 * h] = (unsigned)v.x;
            ix[r][2 * h + 1] = (unsigned)v.y;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fx[i] = cbrt[ix[0][i] * CR];
        fy[i] = cbrt[ix[1][i] * CR];
        fz[i] = cbrt[ix[2][i] * CR];
    }
    TIA_STAGE_FENCE();
    // L = descale(296 fy - 1336934, 15), a = descale(500 (fx - fy) + 128 << 15, 15), b = descale(200 (fy - fz) + 128 << 15, 15),
    // saturated: x + 2^-16 rounded to nearest even == floor(x + 0.5) for the multiples x of 2^-15 (exact below 256; above, the
    // rounding is monotone and the byte saturates)
    const f2 kl = {296.0f / 32768.0f, 296.0f / 32768.0f}, cl = {-2673867.0f / 65536.0f, -2673867.0f / 65536.0f};
    const f2 ka = {500.0f / 32768.0f, 500.0f / 32768.0f}, kb = {200.0f / 32768.0f, 200.0f / 32768.0f};
    const f2 c128 = {128.0f + 1.0f / 65536.0f, 128.0f + 1.0f / 65536.0f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2 x2 = {fx[2 * h], fx[2 * h + 1]}, y2 = {fy[2 * h], fy[2 * h + 1]}, z2 = {fz[2 * h], fz[2 * h + 1]};
        const f2 Lf = __builtin_elementwise_fma(y2, kl, cl);
        const f2 Af = __builtin_elementwise_fma(x2 - y2, ka, c128);
        const f2 Bf = __builtin_elementwise_fma(y2 - z2, kb, c128);
        o[2 * h] = __builtin_amdgcn_cvt_pk_u8_f32(Bf.x, 2, __builtin_amdgcn_cvt_pk_u8_f32(Af.x, 1, __builtin_amdgcn_cvt_pk_u8_f32(Lf.x, 0, 0u)));
        o[2 * h + 1] = __builtin_amdgcn_cvt_pk_u8_f32(Bf.y, 2, __builtin_amdgcn_cvt_pk_u8_f32(Af.y, 1, __builtin_amdgcn_cvt_pk_u8_f32(Lf.y, 0, 0u)));
    }
}
__device__ __forceinline__ void lab_fwd4(const LabFwdLds& s, const LabCoef& k, const uint32_t (&p)[4], uint32_t (&o)[4]) {
    lab_fwd4t<1>(s.gamma, s.cbrt, k, p, o);
}
__device__ __forceinline__ void lab_fwd2(const LabFwdLds& s, const LabCoef& k, uint32_t p0, uint32_t p1, uint32_t& o0, uint32_t& o1) {
    const uint32_t p[4] = {p0, p1, p0, p1};
    uint32_t o[4];
    lab_fwd4(s, k, p, o);
    o0 = o[0];
    o1 = o[1];
}

// ---- Lab -> RGB --------------------------------------------------------------------------------------------------------------------
// abToXZ_b of OpenCV computed directly.  Cube branch: i in (3390, 26870] -- 24-bit multiplies, 32-bit intermediates.  The linear
// branch (i <= 3390: X/Xn or Z/Zn below 0.008856, near-black pixels) costs two quarter-rate multiplies; it sits behind a
// WAVE-UNIFORM test (the compiler turns a per-lane `if` into exec masking that every wave walks through).
__device__ __forceinline__ int ab_cube(int i) {
    const unsigned q = __umul24((unsigned)i, (unsigned)i) >> 14;
    return (int)(__umul24(q, (unsigned)i) >> 14);
}
__device__ __forceinline__ int ab_linear(int i) {
    return i * 108 / 841 - (16384 * 16 / 116 * 108 / 841);  // C integer division truncates toward zero
}
__device__ __forceinline__ int clamp4095(int v) { return v < 0 ? 0 : (v > 4095 ? 4095 : v); }

// lab = L | a << 8 | b << 16 (top byte ignored); result r | g << 8 | b << 16.  N pixels at a time, ONE wave-uniform test for the
// linear branch per call: a test per pixel splits the code into a basic block per pixel and serialises their two LDS round trips
// (measured: the way back ran at half the rate of its instruction count).
template <int N>
__device__ __forceinline__ void lab_inv(const LabInvLds& s, const LabImgLds& im, const LabCoef& k, const uint32_t (&lab)[N], uint32_t (&rgb)[N]) {
    int4 ly[N];
    int adiv[N], bdiv[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
        ly[p] = im.ly[lab[p] & 255u];
        adiv[p] = im.ta[(lab[p] >> 8) & 255u];
        bdiv[p] = im.tb[(lab[p] >> 16) & 255u];
    }
    TIA_STAGE_FENCE();
    int i[N], j[N], x[N], z[N];
    bool lin = false;
#pragma unroll
    for (int p = 0; p < N; ++p) {
        i[p] = ly[p].x + adiv[p];
        j[p] = ly[p].x - bdiv[p];
        x[p] = ab_cube(i[p]);
        z[p] = ab_cube(j[p]);
        lin = lin || i[p] <= 3390 || j[p] <= 3390;
    }
    if (__builtin_amdgcn_ballot_w64(lin) != 0) {
#pragma unroll
        for (int p = 0; p < N; ++p) {
            if (i[p] <= 3390) x[p] = ab_linear(i[p]);
            if (j[p] <= 3390) z[p] = ab_linear(j[p]);
        }
    }
    int ro[N], go[N], bo[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
        ro[p] = clamp4095((__mul24(k.i[0], x[p]) + (__mul24(k.i[2], z[p]) + ly[p].y)) >> 14);
        go[p] = clamp4095((__mul24(k.i[3], x[p]) + (__mul24(k.i[5], z[p]) + ly[p].z)) >> 14);
        bo[p] = clamp4095((__mul24(k.i[6], x[p]) + (__mul24(k.i[8], z[p]) + ly[p].w)) >> 14);
    }
    uint32_t r8[N], g8[N], b8[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
        r8[p] = s.inv_gamma[ro[p]];
        g8[p] = s.inv_gamma[go[p]];
        b8[p] = s.inv_gamma[bo[p]];
    }
    TIA_STAGE_FENCE();
#pragma unroll
    for (int p = 0; p < N; ++p) rgb[p] = r8[p] | (g8[p] << 8) | (b8[p] << 16);
}
__device__ __forceinline__ uint32_t lab_inv1(const LabInvLds& s, const LabImgLds& im, const LabCoef& k, uint32_t lab) {
    const uint32_t in[1] = {lab};
    uint32_t out[1];
    lab_inv<1>(s, im, k, in, out);
    return out[0];
}

// four pixel dwords (low three bytes each) -> the three dwords of a 12-byte group
__device__ __forceinline__ void pack_group(const uint32_t (&p)[4], uint32_t& a, uint32_t& b, uint32_t& c) {
    a = (p[0] & 0xffffffu) | (p[1] << 24);
    b = ((p[1] >> 8) & 0xffffu) | (p[2] << 16);
    c = ((p[2] >> 16) & 0xffu) | (p[3] << 8);
}

// ---- channel histograms: 8 interleaved copies ----------------------------------------------------------------------------------
// counter (channel c, value v, copy s) at dword c * 2048 + v * 8 + s; a lane uses copy lane & 7.  Whatever the image (flat
// background: all 64 lanes on one value), an LDS atomic of a wave serialises at most 8 same-address updates.
constexpr int kHistCopies = 8;
constexpr int kHistDwords = 3 * 256 * kHistCopies;
__device__ __forceinline__ void lab_hist_add(unsigned* hist_lane /* = hist + (lane & 7) */, uint32_t lab) {
    atomicAdd(&hist_lane[(lab & 255u) * kHistCopies], 1u);
    atomicAdd(&hist_lane[2048 + ((lab >> 8) & 255u) * kHistCopies], 1u);
    atomicAdd(&hist_lane[4096 + ((lab >> 16) & 255u) * kHistCopies], 1u);
}
__device__ __forceinline__ unsigned lab_hist_fold(const unsigned* hist, int c, int v) {
    unsigned t = 0;
#pragma unroll
    for (int s = 0; s < kHistCopies; ++s) t += hist[c * 2048 + v * kHistCopies + s];
    return t;
}

// The fused kernel counts only L that way (16 copies).  The a and b channel values are the integers v - 128, so both of their moment
// sums are integers far below 2^53: NumPy's pairwise float64 sums of hist[v] * val and hist[v] * val * val are exact whatever the
// order, and equal sum(a) - 128 n and sum(a^2) - 256 sum(a) + 16384 n accumulated per lane in integers -- four byte dot products per
// four pixels instead of two LDS atomics per pixel.  (L's values are the float32 quotients v / 2.55: its sum of squares rounds per
// bin, so L keeps its histogram and the pairwise order.)
constexpr int kLCopies = 16;
constexpr int kLHistDwords = 256 * kLCopies;
struct AbSums {
    uint32_t a, a2, b, b2;
};
__device__ __forceinline__ void ab_accumulate(AbSums& m, const uint32_t (&l)[4]) {
    const uint32_t t01 = __builtin_amdgcn_perm(l[1], l[0], 0x06020501u);  // a0 a1 b0 b1
    const uint32_t t23 = __builtin_amdgcn_perm(l[3], l[2], 0x06020501u);
    const uint32_t a4 = __builtin_amdgcn_perm(t23, t01, 0x05040100u), b4 = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
    m.a = __builtin_amdgcn_udot4(a4, 0x01010101u, m.a, false);
    m.a2 = __builtin_amdgcn_udot4(a4, a4, m.a2, false);
    m.b = __builtin_amdgcn_udot4(b4, 0x01010101u, m.b, false);
    m.b2 = __builtin_amdgcn_udot4(b4, b4, m.b2, false);
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// ---- moments and per-image tables (stainnorm.py:277-292, 336-339) -------------------------------------------------------------
// cv2.meanStdDev of the float32 channels == moments of 256 weighted values (f64, summed in NumPy's pairwise order for 256
// elements so host and device agree to the bit), then the float32 chain ((chan - mean) * (t_std / std) + t_mean), back to Lab
// bytes (x2.55 or +128, clip, truncate).
struct ReinhardTarget {
    double mean[3];
    double stdv[3];
};
// NumPy's pairwise sum of 256 doubles, serial form (one thread)
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
// the same sums for NQ arrays by 16 NQ threads: thread (q, half, j) owns accumulator r[j] of one half of array q -- the additions
// and their order are those of np_pairwise_256.  prod: [NQ][256]; part: [NQ][2][8]; sums: [NQ].
template <int NQ>
__device__ __forceinline__ void pairwise_n(const double* prod, double* part, double* sums) {
    const int tid = threadIdx.x;
    if (tid < 16 * NQ) {
        const int q = tid >> 4, half = (tid >> 3) & 1, j = tid & 7;
        const double* p = prod + q * 256 + half * 128;
        double r = p[j];
        for (int i = 8; i < 128; i += 8) r += p[i + j];
        part[tid] = r;
    }
    __syncthreads();
    if (tid < NQ) {
        const double* r = part + tid * 16;
        const double p0 = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        const double p1 = ((r[8] + r[9]) + (r[10] + r[11])) + ((r[12] + r[13]) + (r[14] + r[15]));
        sums[tid] = p0 + p1;
    }
    __syncthreads();
}

// byte tables of one image from its statistics: thread v < 256 writes lut[c][v]
__device__ __forceinline__ uint8_t reinhard_lut_entry(const float* __restrict__ chan_vals, int c, int v, double mean, double sd,
                                                      const ReinhardTarget& tgt) {
    const float mean32 = (float)mean;
    const float ratio32 = (float)(tgt.stdv[c] / sd);
    const float tmean32 = (float)tgt.mean[c];
    float norm = (chan_vals[c * 256 + v] - mean32) * ratio32 + tmean32;
    norm = c == 0 ? norm * 2.55f : norm + 128.0f;
    norm = norm < 0.0f ? 0.0f : (norm > 255.0f ? 255.0f : norm);  // NaN (std == 0) falls through; flagged by the caller
    return (uint8_t)(int)norm;
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
    for (int c = 0; c < 3; ++c) lut[(size_t)blockIdx.x * 768 + c * 256 + v] = reinhard_lut_entry(chan_vals, c, v, stat[c][0], stat[c][1], tgt);
}

// ---- the fused transform: one persistent workgroup per patch ------------------------------------------------------------------
// LDS (bytes): forward tables 16384 | inverse 4096 | L histogram 16384 (re-used for the moment products and then the image's
// tables) | wave stages FT/64 * 3072: 49 KB, three workgroups per CU.
struct FusedLds {
    LabFwdLds fwd;
    LabInvLds inv;
    union {
        unsigned hist[kLHistDwords];  // pass 1: L counts, 16 copies
        struct {                      // between the passes: L's moment products; pass 2: the image's tables
            double prod[2 * 256];
            double part[32];
            double sums[2];
            double stat[6];
            unsigned long long ab[4];  // sum a, sum a^2, sum b, sum b^2 (raw bytes)
            LabImgLds img;
        } m;
    } u;
    unsigned long long ab_acc[4];
};

// Shared by the two one-launch kernels: after pass 1 (L counted in s.u.hist, a / b sums in each lane's `ab`) -> moments (written to
// meanstd / flags when given) -> the image's tables in s.u.m.img (not STATS_ONLY).  Ends with a barrier.
template <bool STATS_ONLY>
__device__ __forceinline__ void fused_moments(FusedLds& s, const AbSums& ab, long hw, long patch, const tia_lab_tables* __restrict__ tab,
                                              const float* __restrict__ chan_vals, const ReinhardTarget& tgt,
                                              double* __restrict__ meanstd, int* __restrict__ flags) {
    const int tid = threadIdx.x, lane = tid & 63;
    {
        const unsigned long long ra = wave_sum_u64(ab.a), ra2 = wave_sum_u64(ab.a2), rb = wave_sum_u64(ab.b), rb2 = wave_sum_u64(ab.b2);
        if (lane == 0) {
            atomicAdd(&s.ab_acc[0], ra);
            atomicAdd(&s.ab_acc[1], ra2);
            atomicAdd(&s.ab_acc[2], rb);
            atomicAdd(&s.ab_acc[3], rb2);
        }
    }
    __syncthreads();
    unsigned cnt = 0;
    if (tid < 256) {
#pragma unroll
        for (int c2 = 0; c2 < kLCopies; ++c2) cnt += s.u.hist[tid * kLCopies + c2];
    }
    __syncthreads();  // histogram folded into registers: its LDS becomes the products
    if (tid < 256) {
        const double val = (double)chan_vals[tid];
        const double hv = (double)cnt * val;
        s.u.m.prod[tid] = hv;
        s.u.m.prod[256 + tid] = hv * val;
    }
    if (tid < 4) s.u.m.ab[tid] = s.ab_acc[tid];
    __syncthreads();
    pairwise_n<2>(s.u.m.prod, s.u.m.part, s.u.m.sums);
    if (tid < 3) {
        const double npx = (double)hw;  // every pixel is counted once per channel
        double s1, s2;
        if (tid == 0) {
            s1 = s.u.m.sums[0];
            s2 = s.u.m.sums[1];
        } else {  // a, b: exact integer sums of (v - 128) and (v - 128)^2
            const long long sv = (long long)s.u.m.ab[2 * tid - 2], sv2 = (long long)s.u.m.ab[2 * tid - 1];
            s1 = (double)(sv - 128ll * hw);
            s2 = (double)(sv2 - 256ll * sv + 16384ll * hw);
        }
        const double mean = s1 / npx;
        double var = s2 / npx - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double sd = sqrt(var);
        s.u.m.stat[tid] = mean;
        s.u.m.stat[3 + tid] = sd;
        if (meanstd) {
            meanstd[(size_t)patch * 6 + tid] = mean;
            meanstd[(size_t)patch * 6 + 3 + tid] = sd;
        }
        if (flags && sd == 0.0) atomicOr(&flags[patch], 1);
    }
    __syncthreads();
    if (STATS_ONLY) return;
    if (tid < 256) {
        const int l = reinhard_lut_entry(chan_vals, 0, tid, s.u.m.stat[0], s.u.m.stat[3], tgt);
        const int a = reinhard_lut_entry(chan_vals, 1, tid, s.u.m.stat[1], s.u.m.stat[4], tgt);
        const int b = reinhard_lut_entry(chan_vals, 2, tid, s.u.m.stat[2], s.u.m.stat[5], tgt);
        set_img_entry(s.u.m.img, tab, tid, l, a, b);
    }
    __syncthreads();
}

// STATS_ONLY: the statistics half alone (get_mean_std / fit): nothing parked, nothing written but six doubles per image.
// Requires hw % 1024 == 0 and 16-byte aligned images (the launcher checks).  scratch: gridDim.x slots of hw dwords.
template <bool STATS_ONLY>
__global__ __launch_bounds__(FT) void reinhard_fused_kernel(const uint8_t* __restrict__ img, long n, long hw,
                                                             const tia_lab_tables* __restrict__ tab,
                                                             const float* __restrict__ chan_vals, const ReinhardTarget tgt,
                                                             uint32_t* __restrict__ scratch, uint8_t* __restrict__ out,
                                                             double* __restrict__ meanstd, int* __restrict__ flags) {
    __shared__ FusedLds s;
    load_fwd(s.fwd, tab);
    if (!STATS_ONLY) load_inv(s.inv, tab);
    const LabCoef k = load_coef(tab);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = FT / 64;
    const long ngroups = hw / 256;  // wave steps: 64 lanes x 4 pixels
    v4u* slot = reinterpret_cast<v4u*>(scratch + (size_t)blockIdx.x * (size_t)hw);  // step g: 64 lanes x 16 bytes
    unsigned* hist_lane = s.u.hist + (lane & (kLCopies - 1));
    for (long patch = blockIdx.x; patch < n; patch += gridDim.x) {
        const uint8_t* src = img + (size_t)patch * (size_t)hw * 3u;
        for (int i = tid; i < kLHistDwords; i += FT) s.u.hist[i] = 0;
        if (tid < 4) s.ab_acc[tid] = 0ull;
        __syncthreads();  // tables loaded (first patch); histogram clear
        // ---- pass 1: RGB -> Lab (parked, one dword per pixel), L histogram, a / b integer moments.  A lane takes 4 pixels (12
        // bytes, one dwordx3 load; a wave 768 contiguous bytes) per step: no LDS staging -- the LDS is this kernel's bottleneck --
        // and few registers per step.
        AbSums ab = {0u, 0u, 0u, 0u};
        {
            const U3* g3 = reinterpret_cast<const U3*>(src);
            U3 cur, nxt;
            long g = wv;
            if (g < ngroups) cur = g3[g * 64 + lane];
            for (; g < ngroups; g += NW) {
                const long gn = g + NW;
                if (gn < ngroups) nxt = g3[gn * 64 + lane];
                uint32_t p[4], l[4];
                group_pixels(cur.x, cur.y, cur.z, p);
                lab_fwd4(s.fwd, k, p, l);
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(&hist_lane[(l[i] & 255u) * kLCopies], 1u);
                ab_accumulate(ab, l);
                if (!STATS_ONLY) {
                    v4u t;
                    t.x = l[0];
                    t.y = l[1];
                    t.z = l[2];
                    t.w = l[3];
                    slot[g * 64 + lane] = t;
                }
                cur = nxt;
            }
        }
        fused_moments<STATS_ONLY>(s, ab, hw, patch, tab, chan_vals, tgt, meanstd, flags);
        if (STATS_ONLY) continue;
        // ---- pass 2: parked Lab -> tables -> RGB
        {
            U3* d3 = reinterpret_cast<U3*>(out + (size_t)patch * (size_t)hw * 3u);
            v4u cur, nxt;
            long g = wv;
            if (g < ngroups) cur = slot[g * 64 + lane];
            for (; g < ngroups; g += NW) {
                const long gn = g + NW;
                if (gn < ngroups) nxt = slot[gn * 64 + lane];
                const uint32_t lb[4] = {cur.x, cur.y, cur.z, cur.w};
                uint32_t p[4];
                lab_inv<4>(s.inv, s.u.m.img, k, lb, p);
                U3 o;
                pack_group(p, o.x, o.y, o.z);
                d3[g * 64 + lane] = o;
                cur = nxt;
            }
        }
        __syncthreads();  // the image's tables share LDS with the next patch's histogram
    }
}

// ---- the register-resident transform: patches up to 256 x 256 never leave the CU ---------------------------------------------------
// One 1024-thread workgroup per patch and CU (persistent over the batch).  Thread t owns the 4-pixel groups t, t + 1024, ...: their
// Lab dwords stay in NG x 4 registers between the passes, so HBM (and the fabric) carry exactly 3 bytes in + 3 bytes out per pixel --
// the scratch-slot kernel above moves 8 more bytes per pixel through L2 and was bound by that.  NG = ceil(pixels / 4096) is a
// template parameter (register arrays need static indices: both passes are fully unrolled).
constexpr int RT = 1024;
constexpr int kResCbrtRep = 16;  // 128 KB of dynamic LDS: one workgroup per CU has the room
template <int NG, bool STATS_ONLY>
__global__ __launch_bounds__(RT) void reinhard_resident_kernel(const uint8_t* __restrict__ img, long n, long hw,
                                                                const tia_lab_tables* __restrict__ tab,
                                                                const float* __restrict__ chan_vals, const ReinhardTarget tgt,
                                                                uint8_t* __restrict__ out, double* __restrict__ meanstd,
                                                                int* __restrict__ flags) {
    __shared__ FusedLds s;  // (its single cube-root table stays unused here)
    extern __shared__ float cbrt_rep[];  // [kCbrtN][kResCbrtRep]
    for (int i = threadIdx.x; i < 256 * kGammaRep; i += RT) s.fwd.gamma[i] = (float)tab->gamma[i / kGammaRep];
    for (int i = threadIdx.x; i < kCbrtN * kResCbrtRep; i += RT) cbrt_rep[i] = (float)tab->cbrt[i / kResCbrtRep];
    if (!STATS_ONLY) load_inv(s.inv, tab);
    const LabCoef k = load_coef(tab);
    const int tid = threadIdx.x, lane = tid & 63;
    const float* cbrt_lane = cbrt_rep + (lane & (kResCbrtRep - 1));
    const long ngroups = hw >> 2;
    unsigned* hist_lane = s.u.hist + (lane & (kLCopies - 1));
    for (long patch = blockIdx.x; patch < n; patch += gridDim.x) {
        // Addresses = the patch's (wave-uniform) base + a 32-bit lane offset: one SGPR pair + one VGPR per access (saddr form).  Written
        // as 64-bit pointers per group, the compiler hoists 2 NG address pairs out of the patch loop and spills them -- every access of
        // the NG = 13 / 16 kernels then started with a scratch reload and an s_waitcnt vmcnt(0).
        const uint8_t* const src = img + (size_t)patch * (size_t)hw * 3u;
        unsigned toff = (unsigned)tid * 12u;
        asm volatile("" : "+v"(toff));
        // The thread's groups are requested kAhead groups ahead of their use, each into a register triple of its own (statically
        // indexed: the waits count down as the groups are used; the first form copied `cur = nxt` behind an s_waitcnt vmcnt(0)).
        constexpr int kAhead = NG < 2 ? NG : 2;  // (measured 1, 2, 3, 6, all: 0.63-0.69 ms per 4096 x 224^2 -- the kernel is not waiting for HBM)
        U3 raw[NG];
        auto request = [&](int j) {
            raw[j] = U3{0u, 0u, 0u};
            if ((long)j * RT + tid < ngroups) raw[j] = *reinterpret_cast<const U3*>(src + (toff + (unsigned)j * (RT * 12u)));
        };
#pragma unroll
        for (int j = 0; j < kAhead; ++j) request(j);
        for (int i = tid; i < kLHistDwords; i += RT) s.u.hist[i] = 0;
        if (tid < 4) s.ab_acc[tid] = 0ull;
        __syncthreads();  // tables loaded (first patch); histogram clear; previous patch's tables no longer read
        uint32_t lab[NG][4];
        AbSums ab = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const long g = (long)j * RT + tid;
            if (j + kAhead < NG) request(j + kAhead);
            if (g < ngroups) {
                uint32_t p[4];
                group_pixels(raw[j].x, raw[j].y, raw[j].z, p);
                lab_fwd4t<kResCbrtRep>(s.fwd.gamma, cbrt_lane, k, p, lab[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(&hist_lane[(lab[j][i] & 255u) * kLCopies], 1u);
                ab_accumulate(ab, lab[j]);
            }
        }
        fused_moments<STATS_ONLY>(s, ab, hw, patch, tab, chan_vals, tgt, meanstd, flags);
        if (STATS_ONLY) continue;
        uint8_t* const dst = out + (size_t)patch * (size_t)hw * 3u;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const long g = (long)j * RT + tid;
            if (g < ngroups) {
                uint32_t p[4];
                lab_inv<4>(s.inv, s.u.m.img, k, lab[j], p);
                U3 o;
                pack_group(p, o.x, o.y, o.z);
                *reinterpret_cast<U3*>(dst + (toff + (unsigned)j * (RT * 12u))) = o;
            }
        }
        // (the next patch's barrier after its histogram clear orders these table reads before the tables are overwritten: the clear
        // touches s.u.hist, which the tables share -- so clear only after everyone is done)
        __syncthreads();
    }
}

// ---- three-launch form: histogram over many workgroups per image -> reinhard_lut_kernel -> apply ------------------------------
// grid (bx, n); wide path when hw % 1024 == 0 and the image base is 16-byte aligned, scalar otherwise
__global__ __launch_bounds__(LT) void lab_hist_kernel(const uint8_t* __restrict__ img, long hw, const tia_lab_tables* __restrict__ tab,
                                                       uint32_t* __restrict__ hist, int wide) {
    __shared__ LabFwdLds fwd;
    __shared__ unsigned h[kHistDwords];
    __shared__ __attribute__((aligned(16))) uint8_t stage[LT / 64][kRgbChunk];
    load_fwd(fwd, tab);
    const LabCoef k = load_coef(tab);
    for (int i = threadIdx.x; i < kHistDwords; i += LT) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned* hist_lane = h + (lane & (kHistCopies - 1));
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    long done = 0;
    if (wide) {
        const long nchunks = hw / kPxChunk, wstride = (long)gridDim.x * (LT / 64);
        RgbChunk cur, nxt;
        long c = (long)blockIdx.x * (LT / 64) + wv;
        if (c < nchunks) rgb_chunk_issue(cur, p + c * kRgbChunk);
        for (; c < nchunks; c += wstride) {
            const long cn = c + wstride;
            if (cn < nchunks) rgb_chunk_issue(nxt, p + cn * kRgbChunk);
            uint32_t w[12];
            rgb_chunk_transpose(cur, stage[wv], w);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t px[4], l[4];
                group_pixels(w[3 * q], w[3 * q + 1], w[3 * q + 2], px);
                lab_fwd4(fwd, k, px, l);
#pragma unroll
                for (int i = 0; i < 4; ++i) lab_hist_add(hist_lane, l[i]);
            }
            cur = nxt;
        }
        done = nchunks * kPxChunk;
    }
    for (long i = done + (long)blockIdx.x * LT + threadIdx.x; i < hw; i += (long)gridDim.x * LT) {
        const uint32_t px = (uint32_t)p[3 * i] | ((uint32_t)p[3 * i + 1] << 8) | ((uint32_t)p[3 * i + 2] << 16);
        uint32_t l0, l1;
        lab_fwd2(fwd, k, px, px, l0, l1);
        lab_hist_add(hist_lane, l0);
    }
    __syncthreads();
    uint32_t* out = hist + (size_t)blockIdx.y * 768;
    for (int i = threadIdx.x; i < 768; i += LT) {
        const unsigned v = lab_hist_fold(h, i >> 8, i & 255);
        if (v) atomicAdd(&out[i], v);
    }
}

// out = LAB2RGB(lut[RGB2LAB(img)]) (lut != nullptr), or one of the plain conversions: MODE 0 Reinhard apply, 1 RGB->Lab, 2 Lab->RGB.
// grid (bx, n) over images of hw pixels.
template <int MODE>
__global__ __launch_bounds__(LT) void lab_stream_kernel(const uint8_t* __restrict__ img, long hw, const tia_lab_tables* __restrict__ tab,
                                                         const uint8_t* __restrict__ lut, uint8_t* __restrict__ out, int wide) {
    __shared__ LabFwdLds fwd;
    __shared__ LabInvLds inv;
    __shared__ LabImgLds im;
    __shared__ __attribute__((aligned(16))) uint8_t stage[LT / 64][kRgbChunk];
    if (MODE != 2) load_fwd(fwd, tab);
    if (MODE != 1) {
        load_inv(inv, tab);
        build_img_tables(im, tab, MODE == 0 ? lut + (size_t)blockIdx.y * 768 : nullptr);
    }
    const LabCoef k = load_coef(tab);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    uint8_t* o = out + (size_t)blockIdx.y * (size_t)hw * 3u;
    auto one = [&](uint32_t px) -> uint32_t {
        uint32_t l0 = px, l1;
        if (MODE != 2) lab_fwd2(fwd, k, px, px, l0, l1);
        return MODE == 1 ? l0 : lab_inv1(inv, im, k, l0);
    };
    long done = 0;
    if (wide) {
        const long nchunks = hw / kPxChunk, wstride = (long)gridDim.x * (LT / 64);
        RgbChunk cur, nxt;
        long c = (long)blockIdx.x * (LT / 64) + wv;
        if (c < nchunks) rgb_chunk_issue(cur, p + c * kRgbChunk);
        for (; c < nchunks; c += wstride) {
            const long cn = c + wstride;
            if (cn < nchunks) rgb_chunk_issue(nxt, p + cn * kRgbChunk);
            uint32_t w[12];
            rgb_chunk_transpose(cur, stage[wv], w);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t px[4], l[4];
                group_pixels(w[3 * q], w[3 * q + 1], w[3 * q + 2], px);
                if (MODE != 2) {
                    lab_fwd4(fwd, k, px, l);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) l[i] = px[i];
                }
                if (MODE != 1) {
                    uint32_t o4[4];
                    lab_inv<4>(inv, im, k, l, o4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) l[i] = o4[i];
                }
                pack_group(l, w[3 * q], w[3 * q + 1], w[3 * q + 2]);
            }
            rgb_chunk_store(w, stage[wv], o + c * kRgbChunk);
            cur = nxt;
        }
        done = nchunks * kPxChunk;
    }
    for (long i = done + (long)blockIdx.x * LT + threadIdx.x; i < hw; i += (long)gridDim.x * LT) {
        const uint32_t r = one((uint32_t)p[3 * i] | ((uint32_t)p[3 * i + 1] << 8) | ((uint32_t)p[3 * i + 2] << 16));
        o[3 * i] = (uint8_t)r;
        o[3 * i + 1] = (uint8_t)(r >> 8);
        o[3 * i + 2] = (uint8_t)(r >> 16);
    }
}

// workgroups per image: enough to fill the chip several times over for a batch, ceil(pixels / 64 Ki) for a single large image
static inline unsigned lab_blocks(long hw, long n) {
    const long maxb = (hw + 4 * kPxChunk - 1) / (4 * kPxChunk);  // one step per wave
    long want = (3072 + n - 1) / n;
    const long big = (hw + 65535) / 65536;
    if (want < big) want = big;
    const long b = want < maxb ? want : maxb;
    return (unsigned)(b < 1 ? 1 : b);
}
static inline bool wide_ok(const void* a, const void* b, long hw) {
    return hw % kPxChunk == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

static_assert(sizeof(FusedLds) <= 32768, "five workgroups of the fused kernel per CU (160 KB of LDS)");
// persistent workgroups of the fused kernel: as many as are resident at once (the runtime's occupancy answer for the kernel's
// registers and LDS -- a workgroup beyond that would only start when another ends), never more than there are patches
static long fused_grid(long n) {
    static std::atomic<int> per_device[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int g = per_device[dev].load(std::memory_order_relaxed);
    if (g == 0) {
        int cus = 256, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinhard_fused_kernel<false>, FT, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
        g = cus * per_cu;
        per_device[dev].store(g, std::memory_order_relaxed);
    }
    return n < g ? n : g;
}
constexpr long kFusedMaxPixels = 1L << 18;  // a patch's scratch slot: 4 bytes per pixel, <= 1 MiB (x 768 workgroups)

}  // namespace tia

using namespace tia;

extern "C" int tia_lab_hist_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                                uint32_t* d_hist, void* stream) {
    if (!d_img || !d_tables || !d_hist || n <= 0 || h <= 0 || w <= 0 || n > 65535) return TIA_EINVAL;
    const long hw = (long)h * w;
    hipLaunchKernelGGL(lab_hist_kernel, dim3(lab_blocks(hw, n), (unsigned)n), dim3(LT), 0, (hipStream_t)stream, d_img, hw, d_tables,
                       d_hist, wide_ok(d_img, d_img, hw) ? 1 : 0);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_reinhard_apply_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                                      const uint8_t* d_lut, uint8_t* d_out, void* stream) {
    if (!d_img || !d_tables || !d_lut || !d_out || n <= 0 || h <= 0 || w <= 0 || n > 65535) return TIA_EINVAL;
    const long hw = (long)h * w;
    hipLaunchKernelGGL(lab_stream_kernel<0>, dim3(lab_blocks(hw, n), (unsigned)n), dim3(LT), 0, (hipStream_t)stream, d_img, hw,
                       d_tables, d_lut, d_out, wide_ok(d_img, d_out, hw) ? 1 : 0);
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
    // one "image" of npix pixels; the wide path takes the multiple of 1024 pixels, the scalar tail the rest
    const long hw = (long)npix;
    long nb = (hw + 8 * kPxChunk - 1) / (8 * kPxChunk);
    nb = nb > 4096 ? 4096 : (nb < 1 ? 1 : nb);
    const int wide = ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15) == 0 ? 1 : 0;
    if (dir == 0)
        hipLaunchKernelGGL(lab_stream_kernel<1>, dim3((unsigned)nb), dim3(LT), 0, (hipStream_t)stream, d_src, hw, d_tables,
                           (const uint8_t*)nullptr, d_dst, wide);
    else
        hipLaunchKernelGGL(lab_stream_kernel<2>, dim3((unsigned)nb), dim3(LT), 0, (hipStream_t)stream, d_src, hw, d_tables,
                           (const uint8_t*)nullptr, d_dst, wide);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

// ---- dispatch of the one-launch kernels --------------------------------------------------------------------------------------------
constexpr long kResidentMaxPixels = 16L * 4096;  // NG <= 16: 64 Lab registers per thread
static inline bool resident_ok(long hw) { return hw % 4 == 0 && hw <= kResidentMaxPixels; }
static inline bool scratch_ok(long hw) { return hw % 256 == 0 && hw <= kFusedMaxPixels; }
static long resident_grid(long n) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return n < cus ? n : cus;  // 1024 threads x up to 128 registers: one workgroup per CU
}
constexpr size_t kResidentDynLds = (size_t)kCbrtN * kResCbrtRep * sizeof(float);
template <bool STATS_ONLY>
static bool launch_resident(long hw, long n, hipStream_t st, const uint8_t* img, const tia_lab_tables* tab, const float* chan,
                            const ReinhardTarget& t, uint8_t* out, double* meanstd, int* flags) {
    const long need = ((hw >> 2) + RT - 1) / RT;
    const dim3 grid((unsigned)resident_grid(n)), block(RT);
#define TIA_RESIDENT(NG)                                                                                                         \
    if (need <= NG) {                                                                                                            \
        static DeviceOnce once;                                                                                                  \
        if (!once.ensure([] {                                                                                                    \
                return hipFuncSetAttribute(reinterpret_cast<const void*>(reinhard_resident_kernel<NG, STATS_ONLY>),            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResidentDynLds) == hipSuccess;      \
            }))                                                                                                                  \
            return false;                                                                                                        \
        hipLaunchKernelGGL((reinhard_resident_kernel<NG, STATS_ONLY>), grid, block, kResidentDynLds, st, img, n, hw, tab, chan, t, \
                           out, meanstd, flags);                                                                                 \
        return true;                                                                                                             \
    }
    TIA_RESIDENT(1)
    TIA_RESIDENT(2)
    TIA_RESIDENT(4)
    TIA_RESIDENT(6)
    TIA_RESIDENT(8)
    TIA_RESIDENT(10)
    TIA_RESIDENT(13)
    TIA_RESIDENT(16)
#undef TIA_RESIDENT
    return false;
}

extern "C" size_t tia_reinhard_workspace_bytes(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const long hw = (long)h * w;
    if (resident_ok(hw) || !scratch_ok(hw)) return 0;  // register-resident / three-launch form: no workspace
    return (size_t)fused_grid(n) * (size_t)hw * 4u;
}

extern "C" int tia_reinhard_transform_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                                          const float* d_chan_vals, const double* target_means, const double* target_stds,
                                          uint8_t* d_out, double* d_meanstd, int32_t* d_flags, void* d_workspace,
                                          size_t workspace_bytes, void* stream) {
    if (!d_img || !d_tables || !d_chan_vals || !target_means || !target_stds || !d_out || n <= 0 || h <= 0 || w <= 0)
        return TIA_EINVAL;
    const long hw = (long)h * w;
    if (((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_out)) & 3) != 0 || !(resident_ok(hw) || scratch_ok(hw)))
        return TIA_ESIZE;  // the caller takes the three-launch form
    ReinhardTarget t;
    for (int c = 0; c < 3; ++c) {
        t.mean[c] = target_means[c];
        t.stdv[c] = target_stds[c];
    }
    if (resident_ok(hw)) {
        if (!launch_resident<false>(hw, (long)n, (hipStream_t)stream, d_img, d_tables, d_chan_vals, t, d_out, d_meanstd, d_flags)) {
            (void)hipGetLastError();
            return TIA_ELAUNCH;
        }
        return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
    }
    const long grid = fused_grid(n);
    if (!d_workspace || workspace_bytes < (size_t)grid * (size_t)hw * 4u || (reinterpret_cast<uintptr_t>(d_workspace) & 15) != 0)
        return TIA_EINVAL;
    hipLaunchKernelGGL(reinhard_fused_kernel<false>, dim3((unsigned)grid), dim3(FT), 0, (hipStream_t)stream, d_img,
                       (long)n, hw, d_tables, d_chan_vals, t, reinterpret_cast<uint32_t*>(d_workspace), d_out, d_meanstd, d_flags);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_lab_moments_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                                   const float* d_chan_vals, double* d_meanstd, int32_t* d_flags, void* stream) {
    if (!d_img || !d_tables || !d_chan_vals || !d_meanstd || n <= 0 || h <= 0 || w <= 0) return TIA_EINVAL;
    const long hw = (long)h * w;
    if ((reinterpret_cast<uintptr_t>(d_img) & 3) != 0 || !(resident_ok(hw) || scratch_ok(hw)))
        return TIA_ESIZE;  // the caller takes tia_lab_hist_u8 + tia_reinhard_luts
    ReinhardTarget t = {{0.0, 0.0, 0.0}, {1.0, 1.0, 1.0}};
    if (resident_ok(hw)) {
        if (!launch_resident<true>(hw, (long)n, (hipStream_t)stream, d_img, d_tables, d_chan_vals, t, nullptr, d_meanstd, d_flags)) {
            (void)hipGetLastError();
            return TIA_ELAUNCH;
        }
    } else
        hipLaunchKernelGGL(reinhard_fused_kernel<true>, dim3((unsigned)fused_grid(n)), dim3(FT), 0, (hipStream_t)stream, d_img,
                           (long)n, hw, d_tables, d_chan_vals, t, (uint32_t*)nullptr, (uint8_t*)nullptr, d_meanstd, d_flags);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
