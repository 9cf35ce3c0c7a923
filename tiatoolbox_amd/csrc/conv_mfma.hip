// NHWC float32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain -- the reference's float32 arithmetic, vanilla.py:242), with the bias / residual-add / ReLU
// epilogue of the ResNet BasicBlock fused in (reference: CNNModel.forward -> torchvision resnet BasicBlock,
// models/architecture/vanilla.py:300-316).
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, reduced tap by tap in slices of 32 input channels
// (NHWC keeps a tap's channels contiguous: one 128-byte run per pixel and slice).
//   * workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); tile 128 pixels x BN channels (BN = 64 | 128)
//   * a wave owns 64 x BN/2 outputs = 2 x (BN/64) MFMA tiles of 32x32, accumulators in registers
//   * A slice (128 x 32) staged in LDS row-major with 33-dword rows: MFMA lane i reads row i, bank (33 i + k) % 32 --
//     conflict-free without transposing the NHWC run; B slice (32 x BN, weights pre-packed [tap][cin][cout]) is read
//     along cout, contiguous per lane
//   * the next slice's global loads are issued before the MFMA loop of the current one (register staging)
//   * blockIdx is remapped so that each XCD (its own L2) walks a contiguous range of pixel tiles: neighbouring tiles
//     share their input halo rows
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tiatoolbox_amd.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int NTH = 256;
constexpr int LDA = BK + 1;

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ConvDims {
    int n, h, w, cin, cout, ho, wo, kh, kw, stride, pad;
};

template <int BN>
__global__ __launch_bounds__(NTH, 2) void conv_mfma_f32_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ y, ConvDims d, int relu, int m_tiles) {
    constexpr int NTILE = BN / 64;       // 32-wide MFMA tiles per wave along N
    constexpr int BQ = BN / 4;           // float4 per B row
    constexpr int B_PER_THREAD = BK * BQ / NTH;
    __shared__ float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    // XCD-aware tile order: workgroups go round-robin to the 8 XCDs; give each XCD a contiguous range of pixel tiles
    const int bid = blockIdx.x;
    const int per_xcd = (m_tiles + 7) / 8;
    const int mt_id = (bid % 8) * per_xcd + bid / 8;
    if (mt_id >= m_tiles) return;
    const long m0 = (long)mt_id * BM;
    const int n0 = blockIdx.y * BN;
    const long m_total = (long)d.n * d.ho * d.wo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- A staging: thread -> 4 (pixel, channel-quad) slots; everything per slot lives in named scalars (arrays indexed
    //      through lambdas ended up in scratch memory, whose accesses share the vmcnt counter with the global loads) ----
    const int quad = tid & 7;
    const int slices_per_tap = d.cin / BK;
    const int n_slices = d.kh * d.kw * slices_per_tap;
#define TIA_SLOT_INIT(R)                                                                       \
    int iy0_##R, ix0_##R;                                                                      \
    long pbase_##R;                                                                            \
    bool pvalid_##R;                                                                           \
    {                                                                                          \
        const long m = m0 + (tid >> 3) + 32 * R;                                               \
        pvalid_##R = m < m_total;                                                              \
        const long mm = pvalid_##R ? m : 0;                                                    \
        const int b = (int)(mm / ((long)d.ho * d.wo));                                         \
        const int rem = (int)(mm - (long)b * d.ho * d.wo);                                     \
        const int oy = rem / d.wo, ox = rem - oy * d.wo;                                       \
        iy0_##R = oy * d.stride - d.pad;                                                       \
        ix0_##R = ox * d.stride - d.pad;                                                       \
        pbase_##R = (long)b * d.h * d.w;                                                       \
    }
    TIA_SLOT_INIT(0)
    TIA_SLOT_INIT(1)
    TIA_SLOT_INIT(2)
    TIA_SLOT_INIT(3)
#undef TIA_SLOT_INIT
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    bool ok0, ok1, ok2, ok3;
    rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    // branch-free A load: padding taps read a valid dummy address and are zeroed when the slice is written to LDS -- a
    // conditional load would make the compiler wait for it right here, in front of the MFMA loop
#define TIA_LOAD_A(R)                                                                                         \
    {                                                                                                         \
        const int iy = iy0_##R + kh, ix = ix0_##R + kw;                                                       \
        ok##R = pvalid_##R && (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;                   \
        const long pix = ok##R ? pbase_##R + (long)iy * d.w + ix : 0;                                         \
        ra##R = *reinterpret_cast<const float4*>(x + (pix * d.cin + c0 + 4 * quad));                          \
    }
#define TIA_LOAD_B(R)                                                                                         \
    {                                                                                                         \
        const int s = tid + NTH * R;                                                                          \
        const int row = s / BQ, c4 = s - row * BQ;                                                            \
        rb##R = *reinterpret_cast<const float4*>(wrow + (long)row * d.cout + 4 * c4);                         \
    }
#define TIA_LOAD_SLICE(SIDX)                                                       \
    {                                                                              \
        const int tap = (SIDX) / slices_per_tap;                                   \
        const int c0 = ((SIDX)-tap * slices_per_tap) * BK;                         \
        const int kh = tap / d.kw, kw = tap - kh * d.kw;                           \
        TIA_LOAD_A(0) TIA_LOAD_A(1) TIA_LOAD_A(2) TIA_LOAD_A(3)                    \
        const float* wrow = wk + ((long)tap * d.cin + c0) * d.cout + n0;           \
        TIA_LOAD_B(0) TIA_LOAD_B(1)                                                \
        if (B_PER_THREAD == 4) { TIA_LOAD_B(2) TIA_LOAD_B(3) }                     \
    }
#define TIA_STORE_A(R)                                                             \
    {                                                                              \
        float* dst = As + ((tid >> 3) + 32 * R) * LDA + 4 * quad;                  \
        dst[0] = ok##R ? ra##R.x : 0.0f;                                           \
        dst[1] = ok##R ? ra##R.y : 0.0f;                                           \
        dst[2] = ok##R ? ra##R.z : 0.0f;                                           \
        dst[3] = ok##R ? ra##R.w : 0.0f;                                           \
    }
#define TIA_STORE_SLICE()                                                                       \
    {                                                                                           \
        TIA_STORE_A(0) TIA_STORE_A(1) TIA_STORE_A(2) TIA_STORE_A(3)                             \
        *reinterpret_cast<float4*>(Bs + 4 * (tid + NTH * 0)) = rb0;                             \
        *reinterpret_cast<float4*>(Bs + 4 * (tid + NTH * 1)) = rb1;                             \
        if (B_PER_THREAD == 4) {                                                                \
            *reinterpret_cast<float4*>(Bs + 4 * (tid + NTH * 2)) = rb2;                         \
            *reinterpret_cast<float4*>(Bs + 4 * (tid + NTH * 3)) = rb3;                         \
        }                                                                                       \
    }

    f32x16 acc[2][NTILE];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const float* a_ptr = As + (wm * 64 + (lane & 31)) * LDA + (lane >> 5);
    const float* b_ptr = Bs + (lane >> 5) * BN + wn * (BN / 2) + (lane & 31);

    TIA_LOAD_SLICE(0)
    TIA_STORE_SLICE()
    __syncthreads();
    for (int sidx = 0; sidx < n_slices; ++sidx) {
        // always stage a "next" slice (the last iteration re-reads the final one): no control flow around the staging
        // registers, and their loads stay in flight while the matrix cores work
        const int nxt = sidx + 1 < n_slices ? sidx + 1 : sidx;
        TIA_LOAD_SLICE(nxt)
        __builtin_amdgcn_sched_barrier(0);  // the loads are issued HERE, not sunk below the MFMAs by the scheduler
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], b[NTILE];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = a_ptr[i * 32 * LDA + kk];
#pragma unroll
            for (int j = 0; j < NTILE; ++j) b[j] = b_ptr[kk * BN + j * 32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTILE; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // pin the first use of the staged registers BEHIND the MFMA loop: otherwise the compiler hoists the padding selects
        // (and with them the wait for the global loads) in front of the loop
        asm volatile("" : "+v"(ra0.x), "+v"(ra0.y), "+v"(ra0.z), "+v"(ra0.w), "+v"(ra1.x), "+v"(ra1.y), "+v"(ra1.z), "+v"(ra1.w));
        asm volatile("" : "+v"(ra2.x), "+v"(ra2.y), "+v"(ra2.z), "+v"(ra2.w), "+v"(ra3.x), "+v"(ra3.y), "+v"(ra3.z), "+v"(ra3.w));
        __syncthreads();
        TIA_STORE_SLICE()
        __syncthreads();
    }
#undef TIA_LOAD_A
#undef TIA_LOAD_B
#undef TIA_LOAD_SLICE
#undef TIA_STORE_A
#undef TIA_STORE_SLICE

    // ---- epilogue: bias (+ residual) (+ ReLU); C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    // Per 32x32 tile all 16 residual loads are issued together (rows beyond the end are clamped, their results unused):
    // one memory latency per tile instead of one per element.
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long mrow = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
            float rv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                long m = mrow + (e & 3) + 8 * (e >> 2);
                m = m < m_total ? m : m_total - 1;
                rv[e] = res ? res[m * d.cout + n] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long m = mrow + (e & 3) + 8 * (e >> 2);
                float v = acc[i][j][e] + bv;
                v = v + rv[e];
                if (relu) v = v > 0.0f ? v : 0.0f;
                if (m < m_total) y[m * d.cout + n] = v;
            }
        }
    }
}

// OIHW -> [kh][kw][cin][cout] (the GEMM's B matrix)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw,
                                                           float* __restrict__ out) {
    const long total = (long)cout * cin * kh * kw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % cout);
        long t = i / cout;
        const int c = (int)(t % cin);
        t /= cin;
        const int x = (int)(t % kw), yy = (int)(t / kw);
        out[i] = w[(((long)o * cin + c) * kh + yy) * kw + x];
    }
}

}  // namespace

extern "C" int tia_conv_pack_weights_f32(const float* d_w_oihw, int64_t cout, int64_t cin, int64_t kh, int64_t kw,
                                         float* d_packed, void* stream) {
    if (!d_w_oihw || !d_packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return TIA_EINVAL;
    const long total = (long)cout * cin * kh * kw;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_w_oihw, (int)cout, (int)cin,
                       (int)kh, (int)kw, d_packed);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_conv2d_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                                   float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                                   int64_t kw, int64_t stride, int64_t pad, int32_t relu, void* stream) {
    if (!d_x || !d_w_packed || !d_y || n <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return TIA_EINVAL;
    if (cin % BK != 0 || cout % 64 != 0) return TIA_ESIZE;
    if (((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_w_packed)) & 15) != 0) return TIA_EINVAL;
    const long ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return TIA_EINVAL;
    const long m_total = n * ho * wo;
    const long m_tiles = (m_total + BM - 1) / BM;
    if (m_tiles > 0x7ffffff0L / 8 || n * h * w * cin > 0x7fffffffffffL) return TIA_ESIZE;
    ConvDims d{(int)n, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)kh, (int)kw, (int)stride, (int)pad};
    const long grid_x = ((m_tiles + 7) / 8) * 8;  // whole rounds over the 8 XCDs (surplus workgroups exit at once)
    hipStream_t st = (hipStream_t)stream;
    if (cout % 128 == 0)
        hipLaunchKernelGGL(conv_mfma_f32_kernel<128>, dim3((unsigned)grid_x, (unsigned)(cout / 128)), dim3(NTH), 0, st, d_x, d_w_packed,
                           d_bias, d_residual, d_y, d, relu, (int)m_tiles);
    else
        hipLaunchKernelGGL(conv_mfma_f32_kernel<64>, dim3((unsigned)grid_x, (unsigned)(cout / 64)), dim3(NTH), 0, st, d_x, d_w_packed,
                           d_bias, d_residual, d_y, d, relu, (int)m_tiles);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
