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
//   * the next slice's global loads are issued before the MFMA loop of the current one (register staging), through buffer
//     descriptors: per slice a slot costs an AND, a compare, an add and a select (the per-pixel part is computed once),
//     and the bounds check zero-fills padding taps
//   * measured and rejected on MI355X (profiles/r02h_perf_conv_v*.txt): double-buffered LDS with the staging write in the
//     middle of the MFMA loop and one barrier per slice (2 workgroups / CU instead of 3: -5 %), s_setprio around the MFMA
//     phase (-4 %), 256 x 64 tiles for cout = 64 (-3 %), forcing 4 / 6 waves per SIMD by register cap (spills: -1 / -13 %);
//     128 x 64 tiles everywhere (TIA_CONV_BN64=1): +15-20 % on the three 1x1 down-sampling convolutions, +3 % on the last 3x3
//     layer, -3 % on the 28^2 / 14^2 layers -- the trunk total is unchanged (profiles/r02q_perf_conv_bn64.txt)
//   * blockIdx is remapped so that each XCD (its own L2) walks a contiguous range of pixel tiles: neighbouring tiles
//     share their input halo rows
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tiatoolbox_amd.h"
#include "common.hpp"
#include "conv3x3_spatial.hpp"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int NTH = 256;
constexpr int LDA = BK + 1;

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ConvDims {
    int n, h, w, cin, cout, ho, wo, kh, kw, stride, pad_y, pad_x;  // pad_* = zero rows / columns in front (top, left)
    unsigned x_bytes, w_bytes;  // buffer extents of this launch (both < 2^31: the host splits the batch)
    int pstride;  // floats between horizontally adjacent input pixels: cin, except for the row-packed thin-input form below
};

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int OOB = (int)0x80000000;  // voffset beyond every buffer extent: the load returns zeros (padding taps)

// second, "post-activated" output of the epilogue: y2 = relu(v * scale[c] + shift[c]) of the value v that goes to y -- the
// BatchNorm + ReLU that FOLLOWS a residual sum in a pre-activation network (HoVer-Net: the next unit's "preact" or the
// block's "blk_bna"), produced while the sum is still in registers.  y itself may then be null (only y2 wanted).
// `pre_*` (PRE kernels, 1x1 only): the mirror image on the INPUT side -- the A operand is relu(x * pre_scale[c] + pre_shift[c]),
// applied between the global load and the LDS store, so a pre-activation unit reads the raw residual sum directly and the
// activated copy never exists in memory.
struct ConvPost {
    const float* scale;
    const float* shift;
    float* y2;
    const float* pre_scale;
    const float* pre_shift;
};

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BN, bool POST = false, bool PRE = false>
__global__ __launch_bounds__(NTH, 2) void conv_mfma_f32_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ y, ConvDims d, int relu, int m_tiles,
                                                            ConvPost post = ConvPost{nullptr, nullptr, nullptr, nullptr, nullptr}) {
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

    // Both operands come through buffer descriptors (built from kernel arguments only, so they live in SGPRs): a 32-bit
    // per-lane byte offset is all the address arithmetic a load needs, and an out-of-range offset reads as zero -- which
    // is exactly what a padding tap must contribute, without a select after the load.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wk), 0, (int)d.w_bytes, 0x00020000);

    // ---- A staging: thread -> 4 (pixel, channel-quad) slots.  Per slot: the byte offset of tap (0, 0) of its pixel
    //      (may lie before the buffer: only used when the tap is inside the image) and one bit per kernel row / column
    //      saying whether that row / column of taps falls inside the image.  A slice then costs an AND, a compare, an
    //      add and a select per slot; everything else about the slice is wave-uniform (scalar unit). ----
    const int quad = tid & 7;
#define TIA_SLOT_INIT(R)                                                                       \
    int cen_##R;                                                                               \
    unsigned msk_##R;                                                                          \
    {                                                                                          \
        const long m = m0 + (tid >> 3) + 32 * R;                                               \
        const bool pvalid = m < m_total;                                                       \
        const int mm = pvalid ? (int)m : 0;                                                    \
        const int b = mm / (d.ho * d.wo);                                                      \
        const int rem = mm - b * d.ho * d.wo;                                                  \
        const int oy = rem / d.wo, ox = rem - oy * d.wo;                                       \
        const int iy0 = oy * d.stride - d.pad_y, ix0 = ox * d.stride - d.pad_x;                    \
        cen_##R = (((b * d.h + iy0) * d.w + ix0) * d.pstride + 4 * quad) * 4;                      \
        unsigned rows = 0, cols = 0;                                                           \
        for (int t = 0; t < d.kh; ++t) rows |= (unsigned)((unsigned)(iy0 + t) < (unsigned)d.h) << t;        \
        for (int t = 0; t < d.kw; ++t) cols |= (unsigned)((unsigned)(ix0 + t) < (unsigned)d.w) << (16 + t); \
        msk_##R = pvalid ? (rows | cols) : 0u;                                                 \
    }
    TIA_SLOT_INIT(0)
    TIA_SLOT_INIT(1)
    TIA_SLOT_INIT(2)
    TIA_SLOT_INIT(3)
#undef TIA_SLOT_INIT
    const int bvoff = ((tid / BQ) * d.cout + 4 * (tid % BQ)) * 4;  // B: row tid / BQ (+ NTH / BQ per further slot), one float4
    const int brow_step = (NTH / BQ) * d.cout * 4;

    // product and sum rounded separately, like batch_norm + relu (the same arithmetic as the POST epilogue and the stand-alone
    // scale_shift_act kernel: the three forms are interchangeable bit for bit)
    auto pre_act = [](float v, float sc, float sh) {
        if constexpr (PRE) {
            const float a = __fadd_rn(__fmul_rn(v, sc), sh);
            return a > 0.0f ? a : 0.0f;
        } else {
            return v;
        }
    };
    // slice cursor (scalar): tap (kh, kw) and first channel c0 of the slice that is staged next
    int s_kh = 0, s_kw = 0, s_c0 = 0;
    u32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    rb2 = rb3 = u32x4{0u, 0u, 0u, 0u};
    f32x4 rps = {1.0f, 1.0f, 1.0f, 1.0f}, rpt = {0.0f, 0.0f, 0.0f, 0.0f};  // PRE: scale / shift of this thread's channel quad
#define TIA_LOAD_A(R)                                                                                         \
    {                                                                                                         \
        const bool ok = (msk_##R & sel) == sel;                                                               \
        ra##R = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? cen_##R + sdelta : OOB, 0, 0);                 \
    }
#define TIA_LOAD_SLICE()                                                                                      \
    {                                                                                                         \
        const int sdelta = ((s_kh * d.w + s_kw) * d.pstride + s_c0) * 4;                                          \
        const unsigned sel = (1u << s_kh) | (1u << (16 + s_kw));                                              \
        const int swrow = (((s_kh * d.kw + s_kw) * d.cin + s_c0) * d.cout + n0) * 4;                          \
        TIA_LOAD_A(0) TIA_LOAD_A(1) TIA_LOAD_A(2) TIA_LOAD_A(3)                                               \
        if constexpr (PRE) {                                                                                  \
            rps = *reinterpret_cast<const f32x4*>(post.pre_scale + s_c0 + 4 * quad);                          \
            rpt = *reinterpret_cast<const f32x4*>(post.pre_shift + s_c0 + 4 * quad);                          \
        }                                                                                                     \
        rb0 = __builtin_amdgcn_raw_buffer_load_b128(rw, bvoff, swrow, 0);                                     \
        rb1 = __builtin_amdgcn_raw_buffer_load_b128(rw, bvoff, swrow + brow_step, 0);                         \
        if (B_PER_THREAD == 4) {                                                                              \
            rb2 = __builtin_amdgcn_raw_buffer_load_b128(rw, bvoff, swrow + 2 * brow_step, 0);                 \
            rb3 = __builtin_amdgcn_raw_buffer_load_b128(rw, bvoff, swrow + 3 * brow_step, 0);                 \
        }                                                                                                     \
    }
    // advance the cursor; past the last slice it stays there (the final iteration re-stages the last slice, so the loop
    // body has no control flow around its loads)
#define TIA_NEXT_SLICE()                                                                                      \
    {                                                                                                         \
        int c0 = s_c0 + BK, kw = s_kw, kh = s_kh;                                                             \
        if (c0 == d.cin) { c0 = 0; ++kw; }                                                                    \
        if (kw == d.kw) { kw = 0; ++kh; }                                                                     \
        if (kh < d.kh) { s_c0 = c0; s_kw = kw; s_kh = kh; }                                                   \
    }
#define TIA_STORE_A(R)                                                             \
    {                                                                              \
        float* dst = As + ((tid >> 3) + 32 * R) * LDA + 4 * quad;                  \
        dst[0] = pre_act(__uint_as_float(ra##R.x), rps.x, rpt.x);                  \
        dst[1] = pre_act(__uint_as_float(ra##R.y), rps.y, rpt.y);                  \
        dst[2] = pre_act(__uint_as_float(ra##R.z), rps.z, rpt.z);                  \
        dst[3] = pre_act(__uint_as_float(ra##R.w), rps.w, rpt.w);                  \
    }
#define TIA_STORE_SLICE()                                                                       \
    {                                                                                           \
        TIA_STORE_A(0) TIA_STORE_A(1) TIA_STORE_A(2) TIA_STORE_A(3)                             \
        *reinterpret_cast<u32x4*>(Bs + 4 * (tid + NTH * 0)) = rb0;                              \
        *reinterpret_cast<u32x4*>(Bs + 4 * (tid + NTH * 1)) = rb1;                              \
        if (B_PER_THREAD == 4) {                                                                \
            *reinterpret_cast<u32x4*>(Bs + 4 * (tid + NTH * 2)) = rb2;                          \
            *reinterpret_cast<u32x4*>(Bs + 4 * (tid + NTH * 3)) = rb3;                          \
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
    const int n_slices = d.kh * d.kw * (d.cin / BK);

    TIA_LOAD_SLICE()
    TIA_STORE_SLICE()
    __syncthreads();
    for (int sidx = 0; sidx < n_slices; ++sidx) {
        TIA_NEXT_SLICE()
        TIA_LOAD_SLICE()
        __builtin_amdgcn_sched_barrier(0);  // the loads are issued HERE, not sunk below the MFMAs by the scheduler
        // software pipeline over PAIRS of k-steps: the fragments of pair kp + 1 (2 + 2 * NTILE LDS reads: both k-steps of
        // an A row come from one ds_read2) are requested before the 4 * NTILE MFMAs of pair kp are issued, and
        // sched_group_barrier keeps that order -- an LDS round trip then hides behind >= 256 cycles of matrix work
        float a[2][2][2], b[2][2][NTILE];  // [buffer][k-step of the pair][tile]
        auto frags = [&](int buf, int kp) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[buf][q][i] = a_ptr[i * 32 * LDA + 4 * kp + 2 * q];
#pragma unroll
                for (int j = 0; j < NTILE; ++j) b[buf][q][j] = b_ptr[(4 * kp + 2 * q) * BN + j * 32];
            }
        };
        frags(0, 0);
#pragma unroll
        for (int kp = 0; kp < BK / 4; ++kp) {
            const int cur = kp & 1;
            if (kp + 1 < BK / 4) frags(cur ^ 1, kp + 1);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NTILE; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q][i], b[cur][q][j], acc[i][j], 0, 0, 0);
            if (kp + 1 < BK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * NTILE, 0);  // DS reads of the next pair
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTILE, 0);                             // then this pair's MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        TIA_STORE_SLICE()
        __syncthreads();
    }
#undef TIA_LOAD_A
#undef TIA_LOAD_SLICE
#undef TIA_NEXT_SLICE
#undef TIA_STORE_A
#undef TIA_STORE_SLICE

    // ---- epilogue: bias (+ residual) (+ ReLU); C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    // Per 32x32 tile all 16 residual loads are issued together (rows beyond the end are clamped, their results unused):
    // one memory latency per tile instead of one per element.
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        const float bv = bias ? bias[n] : 0.0f;
        float ps = 1.0f, pt = 0.0f;
        if constexpr (POST) {
            ps = post.scale[n];
            pt = post.shift[n];
        }
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
                if constexpr (POST) {
                    if (m < m_total) {
                        if (y) y[m * d.cout + n] = v;
                        float a = __fmul_rn(v, ps);  // product and sum rounded separately, like batch_norm + relu
                        a = __fadd_rn(a, pt);
                        post.y2[m * d.cout + n] = a > 0.0f ? a : 0.0f;
                    }
                } else {
                    if (m < m_total) y[m * d.cout + n] = v;
                }
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

// ---- the ONE dispatch decision: conv2d_impl launches by it, tia_conv2d_route_f32 reports it ---------------------------------------
enum ConvRoute { ROUTE_SLICE = 0, ROUTE_SPATIAL = 1, ROUTE_RING = 2 };

// Shape checks shared by the entry points and the route query (pointer checks stay with the callers).
static int conv2d_check_shape(int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                              int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad_top < 0 || pad_left < 0) return TIA_EINVAL;
    if (cin % BK != 0 || cout % 64 != 0) return TIA_ESIZE;
    // every output pixel must see at least its first tap row / column start inside [-(k-1), h): rows and columns beyond the
    // image on either side read as zeros (that is how asymmetric "same" padding is expressed: pad_top / pad_left + ho / wo)
    if (ho <= 0 || wo <= 0 || kh > 16 || kw > 16 || pad_top >= kh || pad_left >= kw) return TIA_EINVAL;
    if ((ho - 1) * stride - pad_top >= h || (wo - 1) * stride - pad_left >= w) return TIA_EINVAL;
    return TIA_OK;
}

// Images per launch: the kernels address their input with 32-bit byte offsets, so a batch goes in groups of < 2 GiB of input
// (and < 2^30 output pixels).  0: a single image is already too large.
static long conv2d_group(int64_t h, int64_t w, int64_t pstride, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t ho, int64_t wo) {
    const long image_bytes = h * w * pstride * 4, w_bytes = kh * kw * cin * cout * 4;
    if (image_bytes > 0x7fffffffL || w_bytes > 0x7fffffffL || ho * wo > 0x7fffffffL / 4) return 0;
    long group = 0x7fffffffL / image_bytes;
    if (group * ho * wo > 0x7fffffffL / 2) group = 0x7fffffffL / 2 / (ho * wo);
    return group;
}

// Which kernel serves ONE launch over `nb` images.  `plain` = no second epilogue output, no activation on load, dense pixels
// (pstride == cin): only those forms exist on the tap-reuse and ring kernels.
static ConvRoute conv2d_route(bool plain, long nb, long h, long w, long cin, long cout, long kh, long kw, long stride, long pad_top,
                              long pad_left, long ho, long wo) {
    if (!plain) return ROUTE_SLICE;
    // 3x3 / stride 1 on maps that 16 x 16 pixel blocks (or bands) cover with little waste: the tap-reuse kernel (conv3x3_spatial.hip)
    if (tia::conv3x3_spatial_ok(kh, kw, stride, h, w, ho, wo, pad_top, pad_left, true) &&
        tia::conv3x3_spatial_serves(nb, h, w, cin, cout, pad_top, pad_left, ho, wo, TIA_DT_F32))
        return ROUTE_SPATIAL;
    // 1x1 (any stride), and the kh x kw layers the tap-reuse kernel left: the LDS-DMA ring GEMM of conv3x3_spatial.hip
    if (tia::conv_ring_ok(nb, cin, cout, kh, kw, ho, wo)) return ROUTE_RING;
    return ROUTE_SLICE;
}

static int conv2d_impl(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y, int64_t n,
                       int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t stride, int64_t pad_top,
                       int64_t pad_left, int64_t ho, int64_t wo, int32_t relu, const float* d_post_scale, const float* d_post_shift,
                       float* d_y2, void* stream, int64_t pstride = 0, const float* d_pre_scale = nullptr,
                       const float* d_pre_shift = nullptr) {
    const bool with_post = d_y2 != nullptr;
    const bool with_pre = d_pre_scale != nullptr;
    if (pstride <= 0) pstride = cin;
    // activation on load: 1x1 without padding only (a padding tap must contribute zero, not relu(shift))
    if (with_pre && (!d_pre_shift || with_post || kh != 1 || kw != 1 || pad_top != 0 || pad_left != 0 || pstride != cin ||
                     ((reinterpret_cast<uintptr_t>(d_pre_scale) | reinterpret_cast<uintptr_t>(d_pre_shift)) & 15) != 0))
        return TIA_EINVAL;
    if (with_post && (!d_post_scale || !d_post_shift)) return TIA_EINVAL;
    if (!d_x || !d_w_packed || (!d_y && !with_post)) return TIA_EINVAL;
    if (const int rc = conv2d_check_shape(n, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo); rc != TIA_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(d_w_packed) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_x) & (pstride % 4 == 0 ? 15 : 3)) != 0)
        return TIA_EINVAL;
    const long image_bytes = h * w * pstride * 4, w_bytes = kh * kw * cin * cout * 4;
    const long group = tia::even_group(n, conv2d_group(h, w, pstride, cin, cout, kh, kw, ho, wo));
    if (group < 1) return TIA_ESIZE;
    const bool plain = !with_post && !with_pre && pstride == cin;
    hipStream_t st = (hipStream_t)stream;
    for (long first = 0; first < n; first += group) {
        const long nb = n - first < group ? n - first : group;
        const long m_total = nb * ho * wo;
        const long m_tiles = (m_total + BM - 1) / BM;
        ConvDims d{(int)nb, (int)h, (int)w, (int)cin, (int)cout, (int)ho, (int)wo, (int)kh, (int)kw, (int)stride, (int)pad_top,
                   (int)pad_left, (unsigned)(nb * image_bytes), (unsigned)w_bytes, (int)pstride};
        const float* xg = d_x + first * h * w * pstride;
        const float* rg = d_residual ? d_residual + first * ho * wo * cout : nullptr;
        float* yg = d_y ? d_y + first * ho * wo * cout : nullptr;
        const ConvPost post{d_post_scale, d_post_shift, with_post ? d_y2 + first * ho * wo * cout : nullptr, d_pre_scale, d_pre_shift};
        const ConvRoute route = conv2d_route(plain, nb, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo);
        if (route == ROUTE_SPATIAL &&
            tia::conv3x3_spatial_launch(xg, d_w_packed, d_bias, rg, yg, nb, h, w, cin, cout, pad_top, pad_left, ho, wo, TIA_DT_F32, relu, st))
            continue;
        if (route == ROUTE_RING &&
            tia::conv_ring_launch(xg, d_w_packed, d_bias, rg, yg, nb, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo, relu, st))
            continue;
        const long grid_x = ((m_tiles + 7) / 8) * 8;  // whole rounds over the 8 XCDs (surplus workgroups exit at once)
        static const bool force64 = tia::dev_env("TIA_CONV_BN64") != nullptr;  // developer switches (tile-shape experiments)
        static const bool no_rule = tia::dev_env("TIA_CONV_NO_1X1_RULE") != nullptr;
        // 1x1 convolutions with few input channels have only cin / 32 slices per tile: the narrower tile (more workgroups,
        // 5 instead of 3 per CU) hides their prologue / epilogue better (+15-20 % on resnet18's down-sampling convolutions)
        const bool narrow = force64 || (!no_rule && kh == 1 && kw == 1 && cin <= 256);
        const ConvPost none{nullptr, nullptr, nullptr, nullptr, nullptr};
        if (with_pre) {
            if (cout % 128 == 0 && !narrow)
                hipLaunchKernelGGL((conv_mfma_f32_kernel<128, false, true>), dim3((unsigned)grid_x, (unsigned)(cout / 128)), dim3(NTH), 0,
                                   st, xg, d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles, post);
            else
                hipLaunchKernelGGL((conv_mfma_f32_kernel<64, false, true>), dim3((unsigned)grid_x, (unsigned)(cout / 64)), dim3(NTH), 0, st,
                                   xg, d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles, post);
        } else if (cout % 128 == 0 && !narrow) {
            if (with_post)
                hipLaunchKernelGGL((conv_mfma_f32_kernel<128, true>), dim3((unsigned)grid_x, (unsigned)(cout / 128)), dim3(NTH), 0, st, xg,
                                   d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles, post);
            else
                hipLaunchKernelGGL((conv_mfma_f32_kernel<128, false>), dim3((unsigned)grid_x, (unsigned)(cout / 128)), dim3(NTH), 0, st,
                                   xg, d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles, none);
        } else {
            if (with_post)
                hipLaunchKernelGGL((conv_mfma_f32_kernel<64, true>), dim3((unsigned)grid_x, (unsigned)(cout / 64)), dim3(NTH), 0, st, xg,
                                   d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles, post);
            else
                hipLaunchKernelGGL((conv_mfma_f32_kernel<64, false>), dim3((unsigned)grid_x, (unsigned)(cout / 64)), dim3(NTH), 0, st, xg,
                                   d_w_packed, d_bias, rg, yg, d, relu, (int)m_tiles, none);
        }
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_conv2d_nhwc_f32_ex(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                                      float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                                      int64_t kw, int64_t stride, int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo,
                                      int32_t relu, void* stream) {
    if (!d_y) return TIA_EINVAL;
    return conv2d_impl(d_x, d_w_packed, d_bias, d_residual, d_y, n, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo, relu,
                       nullptr, nullptr, nullptr, stream);
}

extern "C" int tia_conv3x3_geometry(int64_t h, int64_t w, int64_t ho, int64_t wo, int64_t pad_top, int64_t pad_left, int32_t geom[4]) {
    const tia::SpPlan plan = tia::conv3x3_spatial_plan(3, 3, 1, h, w, ho, wo, pad_top, pad_left, true);
    if (geom) {
        geom[0] = plan.bw;
        geom[1] = plan.br;
        geom[2] = plan.brow;
        geom[3] = plan.strips;
    }
    return plan.kind;
}

extern "C" int tia_conv2d_route_f32(int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                                    int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo) {
    // the same checks, the same batch split and the same decision function as tia_conv2d_nhwc_f32(_ex): shapes the entry point
    // rejects are rejected here with the same code
    if (const int rc = conv2d_check_shape(n, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo); rc != TIA_OK) return rc;
    const long group = tia::even_group(n, conv2d_group(h, w, cin, cin, cout, kh, kw, ho, wo));
    if (group < 1) return TIA_ESIZE;
    // a batch beyond 2 GiB of input runs in EQUAL groups (tia::even_group; the last one at most k - 1 images shorter): the answer
    // is the route of the first group
    return (int)conv2d_route(true, n < group ? n : group, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo);
}

// Thin-input form (an RGB stem: c = 3): in NHWC the kw * c values under one row of taps are CONTIGUOUS, so a kh x kw
// convolution over c channels is a kh x 1 convolution over 32 "row-packed" channels whose pixels lie c floats apart --
// the same kernel with a pixel stride, weights [kh][32][cout] with rows >= kw * c zero (what they multiply is the rest of
// the 32-float read: finite image data, or zeros beyond the buffer).  The caller pads the rows horizontally.
extern "C" int tia_conv2d_thin_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, float* d_y, int64_t n, int64_t h,
                                        int64_t w, int64_t c, int64_t cout, int64_t kh, int64_t kw, int64_t stride, int64_t pad_top,
                                        int64_t ho, int64_t wo, int32_t relu, void* stream) {
    if (!d_y || c <= 0 || kw <= 0 || c * kw > BK || stride <= 0 || wo <= 0) return TIA_EINVAL;
    // horizontally "valid" (the padding columns are in the input), and the 32-float read of the last output column stays in its row
    if ((wo - 1) * stride + kw > w || (w - (wo - 1) * stride) * c < BK) return TIA_EINVAL;
    return conv2d_impl(d_x, d_w_packed, d_bias, nullptr, d_y, n, h, w, BK, cout, kh, 1, stride, pad_top, 0, ho, wo, relu, nullptr,
                       nullptr, nullptr, stream, c);
}

extern "C" int tia_conv2d_post_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                                        float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                                        int64_t kw, int64_t stride, int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo,
                                        int32_t relu, const float* d_post_scale, const float* d_post_shift, float* d_y2,
                                        void* stream) {
    if (!d_y2) return TIA_EINVAL;
    return conv2d_impl(d_x, d_w_packed, d_bias, d_residual, d_y, n, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo, relu,
                       d_post_scale, d_post_shift, d_y2, stream);
}

extern "C" int tia_conv1x1_pre_nhwc_f32(const float* d_x, const float* d_pre_scale, const float* d_pre_shift, const float* d_w_packed,
                                       const float* d_bias, const float* d_residual, float* d_y, int64_t n, int64_t h, int64_t w,
                                       int64_t cin, int64_t cout, int64_t stride, int32_t relu, void* stream) {
    if (!d_y || !d_pre_scale || !d_pre_shift || h <= 0 || w <= 0 || stride <= 0) return TIA_EINVAL;
    const long ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    return conv2d_impl(d_x, d_w_packed, d_bias, d_residual, d_y, n, h, w, cin, cout, 1, 1, stride, 0, 0, ho, wo, relu, nullptr, nullptr,
                       nullptr, stream, 0, d_pre_scale, d_pre_shift);
}

extern "C" int tia_conv2d_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                                   float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                                   int64_t kw, int64_t stride, int64_t pad, int32_t relu, void* stream) {
    if (h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return TIA_EINVAL;
    const long ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    return tia_conv2d_nhwc_f32_ex(d_x, d_w_packed, d_bias, d_residual, d_y, n, h, w, cin, cout, kh, kw, stride, pad, pad, ho, wo, relu,
                                  stream);
}
