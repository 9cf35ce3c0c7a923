// Stain statistics of LARGE single images (tools/stainnorm.py:68-113 takes ONE image per call: the reference fits on a 1000 x 1000
// target, normalises slide tiles and thumbnails of any size).  The per-patch kernels give an image one workgroup -- a 4096 x 4096
// tile kept 1 of 256 CUs busy for ~70 ms.  Here every sweep of the same algorithm runs over ceil(pixels / 64 Ki) workgroups per image
// and the per-image decisions between the sweeps are one-workgroup kernels on merged state, all stream-ordered (no host round trip):
//   P1  byte histogram (counts add exactly)            -> contrast-enhancer percentiles, folded luminance tables
//   P2  tissue mask + float64 moments of the tissue OD -> per-workgroup partial sums, merged IN WORKGROUP ORDER (deterministic for a
//       given grid), covariance, 3 x 3 eigen-decomposition
//   P3/P4  exact order statistics of the pseudo-angle key over the tissue pixels: radix selection on the order-preserving 64-bit
//       image of the float64 key, six digit passes (11, 11, 11, 11, 11, 9 bits); a pass = one sweep that counts the current digit
//       of the pixels matching the digits fixed so far (LDS histogram -> global atomics) + one small kernel that walks the merged
//       2048 counts to the bin holding each rank.  All four ranks (floor / ceil neighbours of both percentiles) share the sweeps.
//   P5/P6  the same selection for the 99th percentile of both stain concentrations over ALL pixels
//   tail   pseudo-inverse, concentration scale, fused matrix -- the record of the per-patch kernels (include/tiatoolbox_amd.h).
// Every value that enters a result is float64 from the same helpers as the per-patch kernels (np_index, np_lerp, pseudo_angle,
// angle_of_key, jacobi3, dot3); only the summation order of the P2 moments differs, i.e. results agree with those kernels to a few
// ulp and with the oracle to the same 1e-9 as they do (tests/test_stain_gpu.py).  Modes: Macenko, fixed and given stain matrices;
// Vahadane's dictionary learning stays on its own kernels.
#include "stain_stats_common.hpp"

#pragma clang fp contract(off)

namespace tia {

constexpr int GT = 256;        // threads of the sweep kernels
constexpr int kDigitBits = 11;
constexpr int kDigitBins = 1 << kDigitBits;
constexpr int kSelTargets = 4;
constexpr int kMaxBigGroups = 1024;  // workgroups per image

struct BigState {
    unsigned hist[256];                    // P1
    unsigned sel[kSelTargets][kDigitBins];  // digit counts of the current pass (rows shared by targets with equal prefixes)
    unsigned long long prefix[kSelTargets];  // digits fixed so far (order-preserving key space)
    unsigned long long rank[kSelTargets];    // 0-based rank inside the group the prefix selects
    int row[kSelTargets];
    int nkeys;                             // 1: every target ranks the same key; 2: targets 0, 1 rank key 0 and 2, 3 key 1
    int shift, bits;                       // current digit: key >> shift, `bits` wide; shift < 0: selection complete
    int skip;                              // empty tissue mask: the record is final
    int bmin, bmax;
    unsigned flags;
    double plow, phigh;
    double e1[3], e2[3];
    double S[6], P[6];
    double gm[2];
    double nt;
    double partial[kMaxBigGroups][10];
};

__device__ __forceinline__ void big_build_tables(double* od, int (*ty)[256], const tia_stain_tables* __restrict__ tab, double plow, double phigh,
                                                 bool z1) {
    for (int t = threadIdx.x; t < 256; t += blockDim.x) {
        od[t] = tab->od_lut[t];
        int v = t;
        if (z1 && v == 0) v = 1;
        int ce = v;
        if (phigh > plow) {  // contrast_enhancer LUT (utils/misc.py:438-444 + skimage rescale_intensity), folded into the Y-row tables
            double x = (double)v;
            x = x < plow ? plow : (x > phigh ? phigh : x);
            x = (x - plow) / (phigh - plow);
            x = x * 255.0 + 0.0;
            ce = (int)x;
        }
        ty[0][t] = tab->ty[0][ce];
        ty[1][t] = tab->ty[1][ce];
        ty[2][t] = tab->ty[2][ce];
    }
}

// this workgroup's share of the image: pixels [lo, hi), a multiple of 4 apart from the image's end
__device__ __forceinline__ void big_span(long hw, long& lo, long& hi) {
    const long per = (((hw + gridDim.x - 1) / gridDim.x) + 3) & ~3L;
    lo = (long)blockIdx.x * per;
    hi = lo + per < hw ? lo + per : hw;
    if (lo > hw) lo = hw;
}

// ---- P1 --------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GT) void big_hist_kernel(const uint8_t* __restrict__ img, long hw, BigState* __restrict__ states, int z1) {
    __shared__ unsigned bins[256 * 32];  // 32 copies: lane l adds to copy l & 31 of bin v at v * 32 + (l & 31)
    BigState& st = states[blockIdx.y];
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    for (int i = threadIdx.x; i < 256 * 32; i += GT) bins[i] = 0u;
    __syncthreads();
    unsigned* hs = bins + (threadIdx.x & 31);
    long lo, hi;
    big_span(hw, lo, hi);
    const long b0 = lo * 3, b1 = hi * 3;  // bytes: the percentiles are over the flattened image
    const bool al = (reinterpret_cast<uintptr_t>(p) & 3) == 0;
    long done = b0;
    if (al) {  // b0 is a multiple of 12
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        const long d1 = b1 >> 2;
        for (long d = (b0 >> 2) + threadIdx.x; d < d1; d += GT) {
            const uint32_t wv = q[d];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t v = (wv >> (8 * e)) & 255u;
                if (z1) v = v ? v : 1u;
                atomicAdd(hs + v * 32u, 1u);
            }
        }
        done = d1 << 2;
    }
    for (long i = done + threadIdx.x; i < b1; i += GT) {
        uint32_t v = p[i];
        if (z1) v = v ? v : 1u;
        atomicAdd(hs + v * 32u, 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        unsigned tot = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) tot += bins[threadIdx.x * 32 + ((c + threadIdx.x) & 31)];
        if (tot) atomicAdd(&st.hist[threadIdx.x], tot);
    }
}

// percentiles of the contrast enhancer from the merged counts (one workgroup per image): the arithmetic of the per-patch kernels
__global__ __launch_bounds__(256) void big_p1_finish_kernel(long hw, tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
    __shared__ unsigned cum[256];
    __shared__ int ibc[8];
    BigState& st = states[blockIdx.x];
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    const int tid = threadIdx.x;
    // TIA_MODE_GIVEN: the caller's stain matrix arrives in the record itself
    if (prm.mode == TIA_MODE_GIVEN && tid < 6) st.S[tid] = out[TIA_ST_STAIN + tid];
    if (prm.mode == TIA_MODE_FIXED && tid < 6) st.S[tid] = prm.stain_fixed[tid];
    __syncthreads();
    if (tid < TIA_STATS_STRIDE) out[tid] = 0.0;
    if (tid < 64) {
        const unsigned h0 = st.hist[tid * 4], h1 = st.hist[tid * 4 + 1], h2 = st.hist[tid * 4 + 2], h3 = st.hist[tid * 4 + 3];
        const unsigned incl = wave_incl_scan_u32(h0 + h1 + h2 + h3);
        const unsigned base = incl - (h0 + h1 + h2 + h3);
        cum[tid * 4] = base + h0;
        cum[tid * 4 + 1] = base + h0 + h1;
        cum[tid * 4 + 2] = base + h0 + h1 + h2;
        cum[tid * 4 + 3] = incl;
    }
    __syncthreads();
    const unsigned long long nbytes = (unsigned long long)hw * 3ull;
    unsigned long long kp[2], kn[2];
    double gm[2];
    np_index(nbytes, prm.q_img_lo, kp[0], kn[0], gm[0]);
    np_index(nbytes, prm.q_img_hi, kp[1], kn[1], gm[1]);
    {
        const unsigned long long c1 = cum[tid], c0 = tid ? cum[tid - 1] : 0;
        if (c0 <= kp[0] && kp[0] < c1) ibc[0] = tid;
        if (c0 <= kn[0] && kn[0] < c1) ibc[1] = tid;
        if (c0 <= kp[1] && kp[1] < c1) ibc[2] = tid;
        if (c0 <= kn[1] && kn[1] < c1) ibc[3] = tid;
        if (c0 == 0 && c1 > 0) ibc[4] = tid;                                 // min byte
        if (c1 == (unsigned)nbytes && c0 < (unsigned)nbytes) ibc[5] = tid;  // max byte
    }
    __syncthreads();
    if (tid == 0) {
        double plow = np_lerp((double)ibc[0], (double)ibc[1], gm[0]);
        double phigh = np_lerp((double)ibc[2], (double)ibc[3], gm[1]);
        if (plow >= phigh) {
            plow = (double)ibc[4];
            phigh = (double)ibc[5];
        }
        st.plow = plow;
        st.phigh = phigh;
        st.bmin = ibc[4];
        st.bmax = ibc[5];
        st.skip = 0;
        st.flags = 0;
        st.shift = -1;
        out[TIA_ST_PLOW] = plow;
        out[TIA_ST_PHIGH] = phigh;
    }
}

// ---- P2 --------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GT) void big_moments_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                         tia_stain_params prm, BigState* __restrict__ states) {
    __shared__ double od[256];
    __shared__ int ty[3][256];
    __shared__ double red[GT / 64][10];
    BigState& st = states[blockIdx.y];
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    big_build_tables(od, ty, tab, st.plow, st.phigh, prm.zero_to_one != 0);
    __syncthreads();
    const int y_thr = prm.y_thr;
    double acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0;
    long lo, hi;
    big_span(hw, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += GT) {
        const uint32_t r = p[3 * i], g = p[3 * i + 1], b = p[3 * i + 2];
        const int t = ty[0][r] + ty[1][g] + ty[2][b];
        if (((t + (1 << 11)) >> 12) < y_thr) {
            const double x = od[r], y = od[g], z = od[b];
            acc[0] += 1.0;
            acc[1] += x;
            acc[2] += y;
            acc[3] += z;
            acc[4] = __builtin_fma(x, x, acc[4]);
            acc[5] = __builtin_fma(x, y, acc[5]);
            acc[6] = __builtin_fma(x, z, acc[6]);
            acc[7] = __builtin_fma(y, y, acc[7]);
            acc[8] = __builtin_fma(y, z, acc[8]);
            acc[9] = __builtin_fma(z, z, acc[9]);
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const double w = wave_sum(acc[i]);
        if (lane_id() == 0) red[wave_id()][i] = w;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        double t = 0.0;
        for (int w = 0; w < GT / 64; ++w) t += red[w][threadIdx.x];  // fixed order
        st.partial[blockIdx.x][threadIdx.x] = t;
    }
}

// targets that rank the same key and agree in the digits fixed so far count into ONE row (the smallest such target's)
__device__ __forceinline__ void big_share_rows(BigState& st) {
    for (int t = 0; t < kSelTargets; ++t) {
        int r = t;
        for (int u = 0; u < t; ++u)
            if ((st.nkeys == 1 || (u >> 1) == (t >> 1)) && st.prefix[u] == st.prefix[t]) {
                r = u;
                break;
            }
        st.row[t] = r;
    }
}
__device__ __forceinline__ void big_start_selection(BigState& st, const unsigned long long (&ranks)[kSelTargets], int nkeys) {
    st.nkeys = nkeys;
    for (int t = 0; t < kSelTargets; ++t) {
        st.prefix[t] = 0ull;
        st.rank[t] = ranks[t];
    }
    big_share_rows(st);
    st.shift = 64 - kDigitBits;
    st.bits = kDigitBits;
}

__global__ __launch_bounds__(64) void big_eigen_kernel(int groups, tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
    BigState& st = states[blockIdx.x];
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    if (threadIdx.x != 0) return;
    double acc[10];
    for (int i = 0; i < 10; ++i) {
        double t = 0.0;
        for (int g = 0; g < groups; ++g) t += st.partial[g][i];  // workgroup order: deterministic for a given grid
        acc[i] = t;
    }
    const double nt = acc[0];
    const unsigned long long n_tissue = (unsigned long long)nt;
    st.nt = nt;
    if (n_tissue == 0) {
        out[TIA_ST_NTISSUE] = 0.0;
        out[TIA_ST_FLAGS] = (double)TIA_FLAG_EMPTY_MASK;
        st.skip = 1;
        return;
    }
    if (n_tissue < 2) st.flags |= TIA_FLAG_DEGENERATE;
    const double mx = acc[1] / nt, my = acc[2] / nt, mz = acc[3] / nt;
    const double f = 1.0 / (nt - 1.0);
    double cov[6];
    cov[0] = (acc[4] - nt * mx * mx) * f;
    cov[1] = (acc[5] - nt * mx * my) * f;
    cov[2] = (acc[6] - nt * mx * mz) * f;
    cov[3] = (acc[7] - nt * my * my) * f;
    cov[4] = (acc[8] - nt * my * mz) * f;
    cov[5] = (acc[9] - nt * mz * mz) * f;
    double w[3], v[3][3];
    jacobi3(cov, w, v);
    // eigh: ascending eigenvalues; reference takes columns [2,1] = largest, 2nd largest
    int i0 = 0, i1 = 1, i2 = 2;
    if (w[i0] < w[i1]) { int t = i0; i0 = i1; i1 = t; }
    if (w[i0] < w[i2]) { int t = i0; i0 = i2; i2 = t; }
    if (w[i1] < w[i2]) { int t = i1; i1 = i2; i2 = t; }
    double e1[3] = {v[0][i0], v[1][i0], v[2][i0]};
    double e2[3] = {v[0][i1], v[1][i1], v[2][i1]};
    if (e1[0] < 0) { e1[0] = -e1[0]; e1[1] = -e1[1]; e1[2] = -e1[2]; }
    if (e2[0] < 0) { e2[0] = -e2[0]; e2[1] = -e2[1]; e2[2] = -e2[2]; }
    for (int i = 0; i < 6; ++i) out[TIA_ST_COV + i] = cov[i];
    for (int i = 0; i < 3; ++i) {
        st.e1[i] = e1[i];
        st.e2[i] = e2[i];
        out[TIA_ST_EVEC + i] = e1[i];
        out[TIA_ST_EVEC + 3 + i] = e2[i];
    }
    out[TIA_ST_NTISSUE] = nt;
    // ranks of the angular percentiles (floor / ceil neighbours of both)
    unsigned long long kp[2], kn[2];
    np_index(n_tissue, prm.q_phi_lo, kp[0], kn[0], st.gm[0]);
    np_index(n_tissue, prm.q_phi_hi, kp[1], kn[1], st.gm[1]);
    const unsigned long long ranks[kSelTargets] = {kp[0], kn[0], kp[1], kn[1]};
    big_start_selection(st, ranks, 1);
}

// ---- radix selection ------------------------------------------------------------------------------------------------------------
// KIND 0: key = pseudo-angle of the tissue pixel's OD in the eigen-plane (all four targets); KIND 1: key0 / key1 = the two stain
// concentrations of every pixel (targets 0, 1 / 2, 3).
template <int KIND>
__global__ __launch_bounds__(GT) void big_select_sweep_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                              tia_stain_params prm, BigState* __restrict__ states) {
    __shared__ double od[256];
    __shared__ int ty[3][256];
    __shared__ unsigned bins[kSelTargets][kDigitBins];
    BigState& st = states[blockIdx.y];
    if (st.skip || st.shift < 0) return;  // (uniform)
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    big_build_tables(od, ty, tab, st.plow, st.phigh, prm.zero_to_one != 0);
    for (int i = threadIdx.x; i < kSelTargets * kDigitBins; i += GT) (&bins[0][0])[i] = 0u;
    const int shift = st.shift, bits = st.bits;
    const unsigned mask = (1u << bits) - 1u;
    const int hs = shift + bits;  // digits above the current one are fixed (hs == 64: none yet)
    unsigned long long pre[kSelTargets];
    int row[kSelTargets];
    bool own[kSelTargets];  // a target counts into its row only when no earlier target shares it
#pragma unroll
    for (int t = 0; t < kSelTargets; ++t) {
        pre[t] = hs < 64 ? st.prefix[t] >> hs : 0ull;
        row[t] = st.row[t];
        own[t] = row[t] == t;
    }
    double a[6], c[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = KIND == 0 ? st.e1[i] : st.P[2 * i];
        c[i] = KIND == 0 ? st.e2[i] : st.P[2 * i + 1];
    }
    __syncthreads();
    const int y_thr = prm.y_thr;
    long lo, hi;
    big_span(hw, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += GT) {
        const uint32_t r = p[3 * i], g = p[3 * i + 1], b = p[3 * i + 2];
        if (KIND == 0) {
            const int t = ty[0][r] + ty[1][g] + ty[2][b];
            if (!(((t + (1 << 11)) >> 12) < y_thr)) continue;
        }
        const double ox = od[r], oy = od[g], oz = od[b];
        const double p0 = dot3(ox, oy, oz, a[0], a[1], a[2]);
        const double p1 = dot3(ox, oy, oz, c[0], c[1], c[2]);
        unsigned long long key[2];
        if (KIND == 0) {
            key[0] = key[1] = f64_key(pseudo_angle(p1, p0));
        } else {
            key[0] = f64_key(p0);
            key[1] = f64_key(p1);
        }
#pragma unroll
        for (int t = 0; t < kSelTargets; ++t) {
            if (!own[t]) continue;
            const unsigned long long k = key[KIND == 0 ? 0 : (t >> 1)];
            if (hs < 64 && (k >> hs) != pre[t]) continue;
            atomicAdd(&bins[t][(unsigned)(k >> shift) & mask], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kSelTargets; ++t) {
        if (!own[t]) continue;
        for (int i = threadIdx.x; i < kDigitBins; i += GT) {
            const unsigned v = bins[t][i];
            if (v) atomicAdd(&st.sel[t][i], v);
        }
    }
}

// one workgroup per image: walk each target's merged counts to the bin holding its rank, fix that digit, move to the next digit
__global__ __launch_bounds__(256) void big_select_step_kernel(BigState* __restrict__ states) {
    __shared__ unsigned part[256];
    __shared__ unsigned found_bin[kSelTargets];
    __shared__ unsigned long long found_below[kSelTargets];
    BigState& st = states[blockIdx.x];
    if (st.skip || st.shift < 0) return;
    const int tid = threadIdx.x;
    const int shift = st.shift;
    constexpr int PER = kDigitBins / 256;
    for (int t = 0; t < kSelTargets; ++t) {
        const unsigned* bins = st.sel[st.row[t]];
        unsigned loc[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            loc[j] = bins[tid * PER + j];
            sum += loc[j];
        }
        part[tid] = sum;
        __syncthreads();
        unsigned long long before = 0;
        for (int j = 0; j < tid; ++j) before += part[j];  // (256 x 256 adds once per pass: negligible)
        const unsigned long long r = st.rank[t];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (before <= r && r < before + loc[j]) {
                found_bin[t] = (unsigned)(tid * PER + j);
                found_below[t] = before;
            }
            before += loc[j];
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int t = 0; t < kSelTargets; ++t) {
            st.prefix[t] |= (unsigned long long)found_bin[t] << shift;
            st.rank[t] -= found_below[t];
        }
        big_share_rows(st);
        if (shift == 0) {
            st.shift = -1;  // all 64 bits fixed: prefix[t] is the key of rank t
        } else {
            const int nb = shift >= kDigitBits ? kDigitBits : shift;
            st.shift = shift - nb;
            st.bits = nb;
        }
    }
    __syncthreads();
    for (int i = tid; i < kSelTargets * kDigitBins; i += 256) (&st.sel[0][0])[i] = 0u;
}

// ---- after the angular selection: stain vectors, pseudo-inverse, start of the concentration selection ----------------------------
__device__ __forceinline__ void big_pinv(const double (&S)[6], double (&P)[6]) {
    const double a = S[0] * S[0] + S[1] * S[1] + S[2] * S[2];
    const double bb = S[0] * S[3] + S[1] * S[4] + S[2] * S[5];
    const double d = S[3] * S[3] + S[4] * S[4] + S[5] * S[5];
    const double det = a * d - bb * bb;
    const double g00 = d / det, g01 = -bb / det, g11 = a / det;
    for (int j = 0; j < 3; ++j) {
        P[j * 2 + 0] = S[j] * g00 + S[3 + j] * g01;
        P[j * 2 + 1] = S[j] * g01 + S[3 + j] * g11;
    }
}

__global__ __launch_bounds__(64) void big_vectors_kernel(long hw, tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
    BigState& st = states[blockIdx.x];
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    if (threadIdx.x != 0 || st.skip) return;
    double S[6];
    if (prm.mode == TIA_MODE_MACENKO) {
        const double vp0 = key_f64(st.prefix[0]), vn0 = key_f64(st.prefix[1]), vp1 = key_f64(st.prefix[2]), vn1 = key_f64(st.prefix[3]);
        const double min_phi = np_lerp(angle_of_key(vp0), angle_of_key(vn0), st.gm[0]);
        const double max_phi = np_lerp(angle_of_key(vp1), angle_of_key(vn1), st.gm[1]);
        out[TIA_ST_MINPHI] = min_phi;
        out[TIA_ST_MAXPHI] = max_phi;
        const double c1 = cos(min_phi), s1 = sin(min_phi), c2 = cos(max_phi), s2 = sin(max_phi);
        const double* e1 = st.e1;
        const double* e2 = st.e2;
        double v1[3] = {e1[0] * c1 + e2[0] * s1, e1[1] * c1 + e2[1] * s1, e1[2] * c1 + e2[2] * s1};
        double v2[3] = {e1[0] * c2 + e2[0] * s2, e1[1] * c2 + e2[1] * s2, e1[2] * c2 + e2[2] * s2};
        const bool first = v1[0] > v2[0];
        const double* h = first ? v1 : v2;
        const double* e = first ? v2 : v1;
        const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
        const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        for (int i = 0; i < 3; ++i) {
            S[i] = h[i] / nh;
            S[3 + i] = e[i] / ne;
        }
    } else {
        for (int i = 0; i < 6; ++i) S[i] = st.S[i];
    }
    double P[6];
    big_pinv(S, P);
    for (int i = 0; i < 6; ++i) {
        st.S[i] = S[i];
        st.P[i] = P[i];
    }
    // 99th percentile of both concentrations over ALL pixels
    unsigned long long kp, kn;
    double gm;
    np_index((unsigned long long)hw, prm.q_conc, kp, kn, gm);
    st.gm[0] = st.gm[1] = gm;
    const unsigned long long ranks[kSelTargets] = {kp, kn, kp, kn};
    big_start_selection(st, ranks, 2);
}

__global__ __launch_bounds__(64) void big_final_kernel(tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
    BigState& st = states[blockIdx.x];
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    if (threadIdx.x != 0 || st.skip) return;
    double maxc[2];
    maxc[0] = np_lerp(key_f64(st.prefix[0]), key_f64(st.prefix[1]), st.gm[0]);
    maxc[1] = np_lerp(key_f64(st.prefix[2]), key_f64(st.prefix[3]), st.gm[1]);
    unsigned flags = st.flags;
    const double* S = st.S;
    const double* P = st.P;
    for (int i = 0; i < 6; ++i) {
        out[TIA_ST_STAIN + i] = S[i];
        out[TIA_ST_PINV + i] = P[i];
    }
    out[TIA_ST_MAXC + 0] = maxc[0];
    out[TIA_ST_MAXC + 1] = maxc[1];
    bool finite = true;
    for (int i = 0; i < 6; ++i) finite = finite && isfinite(S[i]) && isfinite(P[i]);
    finite = finite && isfinite(maxc[0]) && isfinite(maxc[1]);
    if (!finite) flags |= TIA_FLAG_DEGENERATE;
    if (prm.has_target) {
        const double sc0 = prm.target_maxc[0] / maxc[0], sc1 = prm.target_maxc[1] / maxc[1];
        if (!(isfinite(sc0) && isfinite(sc1))) flags |= TIA_FLAG_DEGENERATE;  // zero 99th-percentile concentration
        out[TIA_ST_SCALE + 0] = sc0;
        out[TIA_ST_SCALE + 1] = sc1;
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < 3; ++c)
                out[TIA_ST_M + j * 3 + c] = P[j * 2 + 0] * sc0 * prm.target_stain[c] + P[j * 2 + 1] * sc1 * prm.target_stain[3 + c];
    }
    out[TIA_ST_FLAGS] = (double)flags;
}

size_t stain_stats_big_workspace_bytes(long n) { return (size_t)n * sizeof(BigState); }

// workgroups per image: one per 64 Ki pixels, more when the batch is small (>= ~768 in all), at most kMaxBigGroups
static int big_groups(long n, long hw) {
    long g = (hw + 65535) / 65536;
    const long want = (768 + n - 1) / n;
    if (g < want) g = want;
    const long cap = (hw + 4095) / 4096;  // at least 4096 pixels per workgroup
    if (g > cap) g = cap;
    if (g > kMaxBigGroups) g = kMaxBigGroups;
    return (int)(g < 1 ? 1 : g);
}

int launch_stain_stats_big(const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm,
                           double* d_stats, void* d_ws, hipStream_t st) {
    BigState* states = reinterpret_cast<BigState*>(d_ws);
    if (hipMemsetAsync(states, 0, stain_stats_big_workspace_bytes(n), st) != hipSuccess) return TIA_ELAUNCH;
    const int groups = big_groups(n, hw);
    const dim3 grid((unsigned)groups, (unsigned)n), one((unsigned)n);
    const int z1 = prm.zero_to_one != 0 ? 1 : 0;
    hipLaunchKernelGGL(big_hist_kernel, grid, dim3(GT), 0, st, d_img, hw, states, z1);
    hipLaunchKernelGGL(big_p1_finish_kernel, one, dim3(256), 0, st, hw, prm, states, d_stats);
    constexpr int kPasses = (64 + kDigitBits - 1) / kDigitBits;
    if (prm.mode == TIA_MODE_MACENKO) {
        hipLaunchKernelGGL(big_moments_kernel, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states);
        hipLaunchKernelGGL(big_eigen_kernel, one, dim3(64), 0, st, groups, prm, states, d_stats);
        for (int pass = 0; pass < kPasses; ++pass) {
            hipLaunchKernelGGL(big_select_sweep_kernel<0>, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states);
            hipLaunchKernelGGL(big_select_step_kernel, one, dim3(256), 0, st, states);
        }
    }
    hipLaunchKernelGGL(big_vectors_kernel, one, dim3(64), 0, st, hw, prm, states, d_stats);
    for (int pass = 0; pass < kPasses; ++pass) {
        hipLaunchKernelGGL(big_select_sweep_kernel<1>, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states);
        hipLaunchKernelGGL(big_select_step_kernel, one, dim3(256), 0, st, states);
    }
    hipLaunchKernelGGL(big_final_kernel, one, dim3(64), 0, st, prm, states, d_stats);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia
