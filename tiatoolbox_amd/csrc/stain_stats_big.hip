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
constexpr int kLinBins = 16384;      // linear bins of a selection's first sweep: 16384 for one key, 8192 per key for two (64 KB of LDS)
constexpr int kCandCap = 16384;      // members of a rank's bin collected for the exact selection (per list: selected in 128 KB of LDS
                                     // by one workgroup); more than that (a huge image, massive ties): radix passes over the image

struct BigState {
    unsigned hist[256];                    // P1
    unsigned sel[kSelTargets][kDigitBins];  // digit counts of the current pass (rows shared by targets with equal prefixes)
    unsigned long long prefix[kSelTargets];  // digits fixed so far (order-preserving key space)
    unsigned long long rank[kSelTargets];    // 0-based rank inside the group the prefix selects
    int row[kSelTargets];
    int nkeys;                             // 1: every target ranks the same key; 2: targets 0, 1 rank key 0 and 2, 3 key 1
    int shift, bits;                       // current digit: key >> shift, `bits` wide; shift < 0: selection complete
    // fast form of a selection (two sweeps instead of six): the first sweep bins every key linearly over its value range (counts in
    // `lin`) and caches the bin code per pixel; the second collects the exact keys of the pixels in the ranks' bins (a few thousand)
    // and one workgroup selects among them.  `fast`: 1 = binning sweep pending, 2 = collecting sweep pending, 0 = off (the radix
    // passes run when shift >= 0: the fall-back for lists that overflow -- massive ties).
    alignas(16) unsigned lin[2][kLinBins];  // (16-byte aligned: the decision step reads and clears it in 16-byte units)
    double lin_lo[2], lin_scale[2];
    int nbins;
    unsigned long long rank0[kSelTargets];
    int fast;
    int tbin[kSelTargets], tlist[kSelTargets];
    unsigned lcount[kSelTargets];
    unsigned diag[2][1 + kSelTargets];     // per selection (angles, concentrations): fell back to radix passes; list sizes
    int skip;                              // empty tissue mask: the record is final
    unsigned ticket;                       // workgroups of the current sweep that have delivered (the last one runs the step)
    unsigned gen;                          // radix fall-back: digit passes completed (the grid barrier of big_select_sweep_kernel)
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
    const long per = (((hw + gridDim.x - 1) / gridDim.x) + 15) & ~15L;  // (16 pixels: whole 16-byte units of image bytes and of bin codes)
    lo = (long)blockIdx.x * per;
    hi = lo + per < hw ? lo + per : hw;
    if (lo > hw) lo = hw;
}

// `f(r, g, b)` for every pixel of [lo, hi): 4-pixel groups as three dword loads per lane (lo is a multiple of 4), the rest bytewise
template <class F>
__device__ __forceinline__ void big_for_each_pixel(const uint8_t* __restrict__ p, long lo, long hi, F&& f) {
    long done = lo;
    if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
        const long g1 = hi >> 2;
        auto four = [&](uint32_t a, uint32_t b, uint32_t c) {
            f(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
            f(a >> 24, b & 255u, (b >> 8) & 255u);
            f((b >> 16) & 255u, b >> 24, c & 255u);
            f((c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
        };
        long g = (lo >> 2) + threadIdx.x;
        for (; g + 3 * GT < g1; g += 4 * GT) {  // four groups requested before the first is used (the sweeps are latency-bound)
            uint32_t w[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k) w[u][k] = q[(g + u * GT) * 3 + k];
#pragma unroll
            for (int u = 0; u < 4; ++u) four(w[u][0], w[u][1], w[u][2]);
        }
        for (; g < g1; g += GT) four(q[g * 3], q[g * 3 + 1], q[g * 3 + 2]);
        done = g1 << 2;
        if (done < lo) done = lo;
    }
    for (long i = done + threadIdx.x; i < hi; i += GT) f((uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2]);
}

// What a sweep's workgroups deliver to the image's state travels as device-scope atomics / write-through stores; waiting for their
// acknowledgements orders them before the ticket (no release fence: on gfx950 that writes the whole L2 back).  Returns true in the
// workgroup that delivered LAST: it runs the per-image decision on the merged state (coherent loads) instead of a kernel of its own.
__device__ __forceinline__ bool big_last_workgroup(BigState& st) {
    __shared__ unsigned s_ticket;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(&st.ticket, 1u);
    __syncthreads();
    const bool last = s_ticket == gridDim.x - 1;
    if (last) {
        if (threadIdx.x == 0) st.ticket = 0u;
        // one agent-scope acquire (drops this CU's / XCD's possibly stale lines), then PLAIN loads of the merged state: they pipeline
        // and hit L2 on the second touch -- coherent loads one at a time cost ~1 us each and made this step 0.5-0.9 ms
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    return last;
}
__device__ __forceinline__ unsigned big_ld(const unsigned* p) { return *p; }
__device__ __forceinline__ double big_ldd(const double* p) { return *p; }

// ---- P1 --------------------------------------------------------------------------------------------------------------------------
__device__ void big_p1_finish(long hw, const tia_stain_params& prm, BigState& st, double* __restrict__ out);

__global__ __launch_bounds__(GT) void big_hist_kernel(const uint8_t* __restrict__ img, long hw, BigState* __restrict__ states, int z1,
                                                      tia_stain_params prm, double* __restrict__ stats) {
    __shared__ unsigned bins[256 * 32];  // 32 copies: lane l adds to copy l & 31 of bin v at v * 32 + (l & 31)
    BigState& st = states[blockIdx.y];
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    for (int i = threadIdx.x; i < 256 * 32; i += GT) bins[i] = 0u;
    __syncthreads();
    unsigned* hs = bins + (threadIdx.x & 31);
    long lo, hi;
    big_span(hw, lo, hi);
    const long b0 = lo * 3, b1 = hi * 3;  // bytes: the percentiles are over the flattened image
    const bool al = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    long done = b0;
    if (al) {  // b0 is a multiple of 48: 16-byte loads, two in flight per thread (4-byte loads made this sweep issue-bound)
        const uint4* q = reinterpret_cast<const uint4*>(p);
        const long d1 = b1 >> 4;
        auto sixteen = [&](const uint4& w) {
            const uint32_t a[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                uint32_t v = (a[e >> 2] >> (8 * (e & 3))) & 255u;
                if (z1) v = v ? v : 1u;
                atomicAdd(hs + v * 32u, 1u);
            }
        };
        long d = (b0 >> 4) + threadIdx.x;
        for (; d + GT < d1; d += 2 * GT) {
            const uint4 w0 = q[d], w1 = q[d + GT];
            sixteen(w0);
            sixteen(w1);
        }
        for (; d < d1; d += GT) sixteen(q[d]);
        done = d1 << 4;
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
__global__ __launch_bounds__(GT) void big_p1_finish_kernel(long hw, tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
    big_p1_finish(hw, prm, states[blockIdx.x], stats + (size_t)blockIdx.x * TIA_STATS_STRIDE);
}

// percentiles of the contrast enhancer from the merged counts (one workgroup per image): the arithmetic of the per-patch kernels
__device__ void big_p1_finish(long hw, const tia_stain_params& prm, BigState& st, double* __restrict__ out) {
    __shared__ unsigned cum[256];
    __shared__ int ibc[8];
    const int tid = threadIdx.x;
    // TIA_MODE_GIVEN: the caller's stain matrix arrives in the record itself
    if (prm.mode == TIA_MODE_GIVEN && tid < 6) st.S[tid] = out[TIA_ST_STAIN + tid];
    if (prm.mode == TIA_MODE_FIXED && tid < 6) st.S[tid] = prm.stain_fixed[tid];
    __syncthreads();
    if (tid < TIA_STATS_STRIDE) out[tid] = 0.0;
    if (tid < 64) {
        const unsigned h0 = big_ld(&st.hist[tid * 4]), h1 = big_ld(&st.hist[tid * 4 + 1]), h2 = big_ld(&st.hist[tid * 4 + 2]),
                       h3 = big_ld(&st.hist[tid * 4 + 3]);
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
__device__ void big_eigen(int groups, const tia_stain_params& prm, BigState& st, double* __restrict__ out);

__global__ __launch_bounds__(GT) void big_moments_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                         tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
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
    big_for_each_pixel(p, lo, hi, [&](uint32_t r, uint32_t g, uint32_t b) {
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
    });
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const double w = wave_sum(acc[i]);
        if (lane_id() == 0) red[wave_id()][i] = w;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        double t = 0.0;
        for (int w = 0; w < GT / 64; ++w) t += red[w][threadIdx.x];  // fixed order
        __hip_atomic_store(&st.partial[blockIdx.x][threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(GT) void big_eigen_kernel(int groups, tia_stain_params prm, BigState* __restrict__ states, double* __restrict__ stats) {
    BigState& st = states[blockIdx.x];
    if (st.skip) return;  // (uniform)
    big_eigen(groups, prm, st, stats + (size_t)blockIdx.x * TIA_STATS_STRIDE);
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
__device__ __forceinline__ void big_start_radix(BigState& st) {
    for (int t = 0; t < kSelTargets; ++t) {
        st.prefix[t] = 0ull;
        st.rank[t] = st.rank0[t];
    }
    big_share_rows(st);
    st.shift = 64 - kDigitBits;
    st.bits = kDigitBits;
    st.fast = 0;
}
// lo / hi: value range of each key (anything outside lands in the edge bins: the selection stays exact)
__device__ __forceinline__ void big_start_selection(BigState& st, const unsigned long long (&ranks)[kSelTargets], int nkeys, const double (&lo)[2],
                                                    const double (&hi)[2]) {
    st.nkeys = nkeys;
    for (int t = 0; t < kSelTargets; ++t) {
        st.rank0[t] = ranks[t];
        st.rank[t] = ranks[t];
        st.prefix[t] = 0ull;
    }
    st.nbins = kLinBins / nkeys;
    for (int k = 0; k < 2; ++k) {
        st.lin_lo[k] = lo[k];
        st.lin_scale[k] = (double)st.nbins / (hi[k] - lo[k]);
    }
    st.shift = -1;
    st.fast = 1;
}

__device__ void big_eigen(int groups, const tia_stain_params& prm, BigState& st, double* __restrict__ out) {
    __shared__ double s_acc[10];
    __shared__ double s_part[kMaxBigGroups * 10];
    // the partial sums come in with all 256 threads at once and are added from LDS: ten columns side by side, each in workgroup order
    // (deterministic for a given grid).  Ten threads walking global memory eight loads at a time took 35 us for 768 workgroups.
    {
        const double* flat = &st.partial[0][0];
        for (int i = threadIdx.x; i < groups * 10; i += GT) s_part[i] = big_ldd(&flat[i]);
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        double t = 0.0;
        for (int g = 0; g < groups; ++g) t += s_part[g * 10 + threadIdx.x];
        s_acc[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double acc[10];
    for (int i = 0; i < 10; ++i) acc[i] = s_acc[i];
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
    const double lo[2] = {-2.0009765625, -2.0009765625}, hi[2] = {2.0009765625, 2.0009765625};
    big_start_selection(st, ranks, 1, lo, hi);
}

// ---- fast selection: linear bins + cached codes, then exact selection among the members of the ranks' bins -------------------------
// key(s) of one pixel: KIND 0 = pseudo-angle in the eigen-plane (tissue pixels), KIND 1 = the two stain concentrations (all pixels)
template <int KIND>
__device__ __forceinline__ void big_keys(const double* od, const double (&a)[3], const double (&c)[3], uint32_t r, uint32_t g, uint32_t b,
                                         double (&x)[2]) {
    const double ox = od[r], oy = od[g], oz = od[b];
    const double p0 = dot3(ox, oy, oz, a[0], a[1], a[2]);
    const double p1 = dot3(ox, oy, oz, c[0], c[1], c[2]);
    if (KIND == 0) {
        x[0] = x[1] = pseudo_angle(p1, p0);
    } else {
        x[0] = p0;
        x[1] = p1;
    }
}
__device__ __forceinline__ int big_lin_bin(double x, double lo, double scale, int nbins) {
    const double d = (x - lo) * scale;
    if (!(d >= 0.0)) return 0;
    if (d >= (double)nbins) return nbins - 1;
    return (int)d;
}

__device__ void big_lin_step(BigState& st);
__device__ void big_exact_step(BigState& st, const unsigned long long* __restrict__ cand, int l);

// sweep A: bin every key, count, cache the codes ([key][pixel] uint16; 0xffff = not a member, i.e. no tissue)
template <int KIND>
__global__ __launch_bounds__(GT) void big_lin_sweep_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                           tia_stain_params prm, BigState* __restrict__ states, uint16_t* __restrict__ codes_all) {
    constexpr int NK = KIND == 0 ? 1 : 2;
    __shared__ double od[256];
    __shared__ int ty[3][256];
    constexpr int NBINS = kLinBins / NK;
    // 16-bit counters, two per dword (32 KB: four workgroups per CU instead of two -- the sweep is float64 arithmetic behind table
    // look-ups and needs the waves); a workgroup's span is swept in rounds of at most kRound pixels so that no counter can wrap
    constexpr long kRound = 65528;
    __shared__ unsigned bins[NK][NBINS / 2];
    BigState& st = states[blockIdx.y];
    if (st.skip || st.fast != 1) return;  // (uniform)
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    uint16_t* codes = codes_all + (size_t)blockIdx.y * 2u * (size_t)hw;
    big_build_tables(od, ty, tab, st.plow, st.phigh, prm.zero_to_one != 0);
    double a[3], c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = KIND == 0 ? st.e1[i] : st.P[2 * i];
        c[i] = KIND == 0 ? st.e2[i] : st.P[2 * i + 1];
    }
    const double lo0 = st.lin_lo[0], sc0 = st.lin_scale[0], lo1 = st.lin_lo[1], sc1 = st.lin_scale[1];
    const int y_thr = prm.y_thr;
    long span_lo, span_hi;
    big_span(hw, span_lo, span_hi);
    for (long lo = span_lo; lo < span_hi; lo += kRound) {
    const long hi = lo + kRound < span_hi ? lo + kRound : span_hi;
    for (int i = threadIdx.x; i < NK * NBINS / 2; i += GT) (&bins[0][0])[i] = 0u;
    __syncthreads();
    // (a group's four codes leave as ONE 8-byte store per plane when the planes are 8-byte aligned)
    const bool packed = (hw & 3) == 0 && (reinterpret_cast<uintptr_t>(codes) & 7) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0;  // (then every pixel is in a group)
    unsigned out0 = 0u, out1 = 0u;
    auto one = [&](long idx, uint32_t r, uint32_t g, uint32_t b) {
        unsigned c0 = 0xffffu, c1 = 0xffffu;
        bool member = true;
        if (KIND == 0) {
            const int t = ty[0][r] + ty[1][g] + ty[2][b];
            member = ((t + (1 << 11)) >> 12) < y_thr;
        }
        if (member) {
            double x[2];
            big_keys<KIND>(od, a, c, r, g, b, x);
            c0 = (unsigned)big_lin_bin(x[0], lo0, sc0, NBINS);
            atomicAdd(&bins[0][c0 >> 1], 1u << (16u * (c0 & 1u)));
            if (NK == 2) {
                c1 = (unsigned)big_lin_bin(x[1], lo1, sc1, NBINS);
                atomicAdd(&bins[NK - 1][c1 >> 1], 1u << (16u * (c1 & 1u)));
            }
        }
        if (!packed) {
            codes[idx] = (uint16_t)c0;
            if (NK == 2) codes[(size_t)hw + idx] = (uint16_t)c1;
        }
        out0 = c0, out1 = c1;
    };
    long idx = 0;
    (void)idx;
    // (pixel index needed for the code cache: the group form of the sweep, spelled out)
    long done = lo;
    if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
        const long g1 = hi >> 2;
        auto four = [&](long g, uint32_t wa, uint32_t wb, uint32_t wc) {
            unsigned k0[4], k1[4];
            one(4 * g, wa & 255u, (wa >> 8) & 255u, (wa >> 16) & 255u);
            k0[0] = out0, k1[0] = out1;
            one(4 * g + 1, wa >> 24, wb & 255u, (wb >> 8) & 255u);
            k0[1] = out0, k1[1] = out1;
            one(4 * g + 2, (wb >> 16) & 255u, wb >> 24, wc & 255u);
            k0[2] = out0, k1[2] = out1;
            one(4 * g + 3, (wc >> 8) & 255u, (wc >> 16) & 255u, wc >> 24);
            k0[3] = out0, k1[3] = out1;
            if (packed) {
                *reinterpret_cast<uint2*>(codes + 4 * g) = uint2{k0[0] | (k0[1] << 16), k0[2] | (k0[3] << 16)};
                if (NK == 2) *reinterpret_cast<uint2*>(codes + (size_t)hw + 4 * g) = uint2{k1[0] | (k1[1] << 16), k1[2] | (k1[3] << 16)};
            }
        };
        long g = (lo >> 2) + threadIdx.x;
        for (; g + GT < g1; g += 2 * GT) {  // two groups requested before the first is used
            const uint32_t a0 = q[g * 3], b0 = q[g * 3 + 1], c0 = q[g * 3 + 2];
            const uint32_t a1 = q[(g + GT) * 3], b1 = q[(g + GT) * 3 + 1], c1 = q[(g + GT) * 3 + 2];
            four(g, a0, b0, c0);
            four(g + GT, a1, b1, c1);
        }
        for (; g < g1; g += GT) four(g, q[g * 3], q[g * 3 + 1], q[g * 3 + 2]);
        done = g1 << 2;
        if (done < lo) done = lo;
    }
    for (long i = done + threadIdx.x; i < hi; i += GT) one(i, (uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2]);
    __syncthreads();
    for (int i = threadIdx.x; i < NK * NBINS / 2; i += GT) {  // key k's counts at st.lin[0] + k * NBINS (one flat area)
        const unsigned v = (&bins[0][0])[i];
        if (v & 0xffffu) atomicAdd(&(&st.lin[0][0])[2 * i], v & 0xffffu);
        if (v >> 16) atomicAdd(&(&st.lin[0][0])[2 * i + 1], v >> 16);
    }
    __syncthreads();  // (the next round clears the counters)
    }
}
// (the decision steps of the two selection sweeps are kernels of their own, one workgroup per image: fused into the sweep by the
// last-workgroup pattern they set its register and LDS footprint -- 248 VGPRs + scratch for the binning sweep, 131 KB of LDS = ONE
// workgroup per CU for the collecting sweep, which then streamed its 33 MB of bin codes at 0.27 TB/s)
__global__ __launch_bounds__(GT) void big_lin_step_kernel(BigState* __restrict__ states) {
    BigState& st = states[blockIdx.x];
    if (st.skip || st.fast != 1) return;  // (uniform)
    big_lin_step(st);
}

// after sweep A (one workgroup): the bin of every rank, the rank inside it, one candidate list per distinct (key, bin).  A key's merged
// counts are read ONCE (16-byte loads of the thread's own run of bins), scanned across the workgroup, and serve all of its targets.
__device__ void big_lin_step(BigState& st) {
    __shared__ unsigned long long wtot[GT / 64];
    __shared__ unsigned found_bin[kSelTargets];
    __shared__ unsigned long long found_below[kSelTargets];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int PER = kLinBins / 256;  // bins per thread when a key has all kLinBins (a multiple of 4)
    static_assert(PER % 4 == 0 && GT == 256, "16-byte loads of a thread's run of bins");
    __shared__ unsigned long long c_rank[kSelTargets];
    __shared__ int c_nb, c_nkeys;
    if (tid < kSelTargets) c_rank[tid] = st.rank[tid];
    if (tid == 0) {
        c_nb = st.nbins;
        c_nkeys = st.nkeys;
    }
    __syncthreads();
    const int nb = c_nb, nkeys = c_nkeys;
    const int per = nb / 256;  // (nbins is kLinBins or half of it: whole 16-byte units per thread)
    for (int key = 0; key < nkeys; ++key) {
        const uint4* bins4 = reinterpret_cast<const uint4*>(&st.lin[0][0] + key * nb + tid * per);
        unsigned loc[PER];
        unsigned long long sum = 0;
#pragma unroll
        for (int j = 0; j < PER / 4; ++j) {
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (4 * j < per) v = bins4[j];
            loc[4 * j] = v.x, loc[4 * j + 1] = v.y, loc[4 * j + 2] = v.z, loc[4 * j + 3] = v.w;
            sum += (unsigned long long)v.x + v.y + v.z + v.w;
        }
        // exclusive prefix of the threads' sums: wave scan, then the waves' totals
        unsigned long long incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = (unsigned long long)__shfl_up((long long)incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        unsigned long long before0 = incl - sum;
        for (int q = 0; q < wv; ++q) before0 += wtot[q];
        for (int t = 0; t < kSelTargets; ++t) {
            if (nkeys == 2 && (t >> 1) != key) continue;  // (uniform) targets 0, 1 rank key 0 and 2, 3 key 1
            const unsigned long long r = c_rank[t];
            unsigned long long before = before0;
            if (before <= r && r < before + sum) {
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    if (before <= r && r < before + loc[j]) {
                        found_bin[t] = (unsigned)(tid * per + j);
                        found_below[t] = before;
                    }
                    before += loc[j];
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int t = 0; t < kSelTargets; ++t) {
            st.tbin[t] = (int)found_bin[t];
            st.rank[t] = c_rank[t] - found_below[t];
            int l = t;
            for (int u = 0; u < t; ++u)
                if ((nkeys == 1 || (u >> 1) == (t >> 1)) && found_bin[u] == found_bin[t]) {
                    l = u;
                    break;
                }
            st.tlist[t] = l;
            st.lcount[t] = 0u;
        }
        st.fast = 2;
    }
    __syncthreads();
    uint4* z = reinterpret_cast<uint4*>(&st.lin[0][0]);
    for (int i = tid; i < 2 * kLinBins / 4; i += 256) z[i] = uint4{0u, 0u, 0u, 0u};
}

// sweep B: the cached codes pick the members of the ranks' bins; their exact keys go to the lists
template <int KIND>
__global__ __launch_bounds__(GT) void big_collect_sweep_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                               tia_stain_params prm, BigState* __restrict__ states,
                                                               const uint16_t* __restrict__ codes_all, unsigned long long* __restrict__ cand_all) {
    constexpr int NK = KIND == 0 ? 1 : 2;
    __shared__ double od[256];
    // A workgroup's members are gathered in LDS and appended to the image's lists with ONE global atomic per list: every member
    // taking its slot with a returning atomic on the list's counter serialised ~10 k same-address round trips per list at the L2
    // (8192^2: 220 us of a sweep whose loads take 20) -- the sweep's time was the list length, whatever the loop did.
    constexpr int kLocalCap = 192;
    __shared__ unsigned long long l_key[kSelTargets][kLocalCap];
    __shared__ unsigned l_cnt[kSelTargets], l_base[kSelTargets];
    BigState& st = states[blockIdx.y];
    if (st.skip || st.fast != 2) return;  // (uniform)
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    const uint16_t* codes = codes_all + (size_t)blockIdx.y * 2u * (size_t)hw;
    unsigned long long* cand = cand_all + (size_t)blockIdx.y * kSelTargets * (size_t)kCandCap;
    for (int t = threadIdx.x; t < 256; t += GT) od[t] = tab->od_lut[t];
    if (threadIdx.x < kSelTargets) l_cnt[threadIdx.x] = 0u;
    double a[3], c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = KIND == 0 ? st.e1[i] : st.P[2 * i];
        c[i] = KIND == 0 ? st.e2[i] : st.P[2 * i + 1];
    }
    int tb[kSelTargets], tl[kSelTargets];
#pragma unroll
    for (int t = 0; t < kSelTargets; ++t) {
        tb[t] = st.tbin[t];
        tl[t] = st.tlist[t];
    }
    __syncthreads();
    long lo, hi;
    big_span(hw, lo, hi);
    auto visit = [&](long i, unsigned c0, unsigned c1) {
        bool hit = false;
#pragma unroll
        for (int t = 0; t < kSelTargets; ++t) hit = hit || (tl[t] == t && (int)((NK == 2 && t >= 2) ? c1 : c0) == tb[t]);
        if (!hit) return;
        double x[2];
        big_keys<KIND>(od, a, c, (uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2], x);
#pragma unroll
        for (int t = 0; t < kSelTargets; ++t) {
            if (tl[t] != t) continue;  // lists are owned by their first target
            const bool second = NK == 2 && t >= 2;
            if ((int)(second ? c1 : c0) != tb[t]) continue;
            const unsigned long long k = f64_key(x[second ? 1 : 0]);
            const unsigned lp = atomicAdd(&l_cnt[t], 1u);
            if (lp < (unsigned)kLocalCap) {
                l_key[t][lp] = k;
            } else {  // (a workgroup with more members than its LDS list holds: the rest one by one)
                const unsigned pos = atomicAdd(&st.lcount[t], 1u);
                if (pos < (unsigned)kCandCap) __hip_atomic_store(&cand[(size_t)t * kCandCap + pos], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    // eight codes per 16-byte load, four loads in flight per thread (the loop is bound by load latency: members are a fraction of a per
    // cent); lo is a multiple of 8
    long done = lo;
    if ((hw & 7) == 0 && (reinterpret_cast<uintptr_t>(codes) & 15) == 0) {
        const uint4* q0 = reinterpret_cast<const uint4*>(codes);
        const uint4* q1 = reinterpret_cast<const uint4*>(codes + (size_t)hw);
        const long g1 = hi >> 3;
        // one test for the whole unit first (a code equals a list's bin in a fraction of a per cent of the pixels; testing every code
        // against every target was 12 vector instructions per pixel and bound this sweep): x ^ pattern has a zero half-word
        unsigned pat[kSelTargets];
#pragma unroll
        for (int t = 0; t < kSelTargets; ++t) pat[t] = tl[t] == t ? ((unsigned)tb[t] | ((unsigned)tb[t] << 16)) : 0xfffefffeu;  // (0xfffe: no code)
        auto zero_half = [](unsigned x) { return (x - 0x00010001u) & ~x & 0x80008000u; };
        auto eight = [&](long g, const uint4& w0, const uint4& w1) {
            const unsigned a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
            unsigned any = 0u;
#pragma unroll
            for (int t = 0; t < kSelTargets; ++t) {
                if (t > 0 && pat[t] == pat[0] && (NK == 1 || t < 2)) continue;  // (uniform) the common case: two lists, two patterns
                const unsigned* plane = (NK == 2 && t >= 2) ? a1 : a0;
#pragma unroll
                for (int k = 0; k < 4; ++k) any |= zero_half(plane[k] ^ pat[t]);
            }
            if (any == 0u) return;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                visit(8 * g + e, (a0[e >> 1] >> (16 * (e & 1))) & 0xffffu, (a1[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        };
        long g = (lo >> 3) + threadIdx.x;
        for (; g + 3 * GT < g1; g += 4 * GT) {
            uint4 w0[4], w1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w0[u] = q0[g + u * GT];
                w1[u] = NK == 2 ? q1[g + u * GT] : uint4{~0u, ~0u, ~0u, ~0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) eight(g + u * GT, w0[u], w1[u]);
        }
        for (; g < g1; g += GT) eight(g, q0[g], NK == 2 ? q1[g] : uint4{~0u, ~0u, ~0u, ~0u});
        done = g1 << 3;
        if (done < lo) done = lo;
    }
    for (long i = done + threadIdx.x; i < hi; i += GT) visit(i, codes[i], NK == 2 ? (unsigned)codes[(size_t)hw + i] : 0xffffu);
    __syncthreads();
    if (threadIdx.x < kSelTargets) {
        const unsigned n = l_cnt[threadIdx.x] < (unsigned)kLocalCap ? l_cnt[threadIdx.x] : (unsigned)kLocalCap;
        l_base[threadIdx.x] = n ? atomicAdd(&st.lcount[threadIdx.x], n) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kSelTargets; ++t) {
        const unsigned n = l_cnt[t] < (unsigned)kLocalCap ? l_cnt[t] : (unsigned)kLocalCap, base = l_base[t];
        for (unsigned i = threadIdx.x; i < n; i += GT)
            if (base + i < (unsigned)kCandCap) cand[(size_t)t * kCandCap + base + i] = l_key[t][i];
    }
}
// grid (images, kSelTargets): one workgroup per candidate LIST (the floor / ceil ranks of a percentile share one; two lists are the rule)
__global__ __launch_bounds__(GT) void big_exact_step_kernel(BigState* __restrict__ states, const unsigned long long* __restrict__ cand_all) {
    BigState& st = states[blockIdx.x];
    if (st.skip || st.fast != 2) return;  // (uniform; nothing below changes `fast` unless a list overflowed)
    big_exact_step(st, cand_all + (size_t)blockIdx.x * kSelTargets * (size_t)kCandCap, (int)blockIdx.y);
}

// after sweep B (one workgroup): each list is copied to LDS once and its ranks found by radix selection there (8-bit digits, the
// digit counts scanned by the 256 threads); the rank right after a selected one -- the ceil neighbour of a percentile -- needs no
// second selection: it is the same key while equal members remain, else the smallest key above it (one more pass).  A list that
// overflowed hands the whole selection to the radix passes over the image.
__device__ void big_exact_step(BigState& st, const unsigned long long* __restrict__ cand, const int l) {
    __shared__ unsigned long long s_list[kCandCap];
    __shared__ unsigned dig[256], wsum[4];
    __shared__ unsigned long long s_prefix, s_rank, s_min[4];
    __shared__ unsigned s_equal;
    __shared__ int s_overflow;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the image's state is read ONCE (global loads right after the acquire cost ~1 us each and the loops below are uniform chains)
    __shared__ int c_list[kSelTargets];
    __shared__ unsigned c_count[kSelTargets];
    __shared__ unsigned long long c_rank[kSelTargets];
    __shared__ int c_nkeys;
    if (tid < kSelTargets) {
        c_list[tid] = st.tlist[tid];
        c_count[tid] = big_ld(&st.lcount[tid]);
        c_rank[tid] = st.rank[tid];
    }
    if (tid == 0) c_nkeys = st.nkeys;
    __syncthreads();
    int first_used = kSelTargets;  // (every workgroup of the image computes the same: the lists' counts are final)
    for (int t = kSelTargets - 1; t >= 0; --t) first_used = c_list[t] < first_used ? c_list[t] : first_used;
    if (tid == 0) {
        s_overflow = 0;
        for (int t = 0; t < kSelTargets; ++t)
            if (c_count[c_list[t]] > (unsigned)kCandCap) s_overflow = 1;
        if (l == first_used) {
            unsigned* dg = st.diag[c_nkeys - 1];
            for (int t = 0; t < kSelTargets; ++t) dg[1 + t] = c_count[c_list[t]];
            dg[0] = (unsigned)s_overflow;
        }
    }
    __syncthreads();
    if (s_overflow) {  // the whole selection goes to the radix passes over the image
        if (tid == 0 && l == first_used) big_start_radix(st);
        return;
    }
    {
        bool used = false;
        for (int t = 0; t < kSelTargets; ++t) used = used || c_list[t] == l;
        if (!used) return;  // (uniform)
        const unsigned cnt = c_count[l];
        const unsigned long long* list = cand + (size_t)l * kCandCap;
        for (unsigned i = tid; i < cnt; i += 256) s_list[i] = list[i];
        __syncthreads();
        unsigned long long prev_rank = ~0ull, prev_key = 0ull;
        unsigned prev_left = 0;  // members equal to prev_key at ranks above prev_rank
        for (int t = 0; t < kSelTargets; ++t) {
            if (c_list[t] != l) continue;  // (uniform)
            const unsigned long long r = c_rank[t];
            unsigned long long key;
            if (r == prev_rank) {
                key = prev_key;
            } else if (r == prev_rank + 1 && prev_left > 0) {
                key = prev_key;
                prev_left -= 1;
                prev_rank = r;
            } else if (r == prev_rank + 1) {  // the smallest key above prev_key
                unsigned long long m = ~0ull;
                for (unsigned i = tid; i < cnt; i += 256) {
                    const unsigned long long k = s_list[i];
                    m = (k > prev_key && k < m) ? k : m;
                }
                m = wave_min_u64(m);
                if (lane == 0) s_min[wv] = m;
                __syncthreads();
                key = s_min[0];
                for (int q = 1; q < 4; ++q) key = s_min[q] < key ? s_min[q] : key;
                __syncthreads();
                // members equal to the new key: counted on demand only if yet another neighbour is asked for (not in this use)
                prev_key = key;
                prev_rank = r;
                prev_left = 0;
            } else {
                if (tid == 0) {
                    s_prefix = 0ull;
                    s_rank = r;
                }
                dig[tid] = 0u;
                __syncthreads();
                for (int shift = 56; shift >= 0; shift -= 8) {  // three barriers per digit (each thread clears its own counter when it reads it)
                    const unsigned long long pre = s_prefix, rr = s_rank;
                    const int hs = shift + 8;
                    for (unsigned i = tid; i < cnt; i += 256) {
                        const unsigned long long k = s_list[i];
                        const bool match = hs >= 64 ? true : (k >> (hs & 63)) == (pre >> (hs & 63));
                        if (match) atomicAdd(&dig[(unsigned)(k >> shift) & 255u], 1u);
                    }
                    __syncthreads();
                    const unsigned mine = dig[tid];
                    dig[tid] = 0u;
                    const unsigned incl = wave_incl_scan_u32(mine);
                    if (lane == 63) wsum[wv] = incl;
                    __syncthreads();
                    unsigned before = incl - mine;
                    for (int q = 0; q < wv; ++q) before += wsum[q];
                    if ((unsigned long long)before <= rr && rr < (unsigned long long)before + mine) {
                        s_prefix = pre | ((unsigned long long)tid << shift);
                        s_rank = rr - before;
                        s_equal = mine;
                    }
                    __syncthreads();
                }
                key = s_prefix;
                prev_key = key;
                prev_rank = r;
                prev_left = s_equal - 1u - (unsigned)s_rank;  // equal members at higher ranks
                __syncthreads();
            }
            if (tid == 0) st.prefix[t] = key;
        }
    }
    // (`fast` stays 2 and `shift` -1: the next selection's start sets both, the fall-back kernel looks at `shift` alone)
}

// ---- radix selection ------------------------------------------------------------------------------------------------------------
// KIND 0: key = pseudo-angle of the tissue pixel's OD in the eigen-plane (all four targets); KIND 1: key0 / key1 = the two stain
// concentrations of every pixel (targets 0, 1 / 2, 3).
__device__ void big_select_step(BigState& st);

template <int KIND>
__global__ __launch_bounds__(GT) void big_select_sweep_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                              tia_stain_params prm, BigState* __restrict__ states) {
    __shared__ double od[256];
    __shared__ int ty[3][256];
    __shared__ unsigned bins[kSelTargets][kDigitBins];
    BigState& st = states[blockIdx.y];
    if (st.skip || st.shift < 0) return;  // (uniform; the common case: no list overflowed, nothing to do)
    const uint8_t* p = img + (size_t)blockIdx.y * (size_t)hw * 3u;
    big_build_tables(od, ty, tab, st.plow, st.phigh, prm.zero_to_one != 0);
    // ALL digit passes in this one launch (the host launches the fall-back unconditionally: six launches per selection that return at
    // once cost 4.5 us each).  Between passes the image's workgroups meet at a barrier built on `gen`: the workgroup that delivers last
    // runs the decision step, publishes it (release fence) and advances `gen`; the others sleep-spin on it.  The workgroups of an
    // image are at most one per CU and dispatched in index order, so the ones a spinning workgroup waits for are resident or next in line.
    unsigned my_gen = __hip_atomic_load(&st.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
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
    big_for_each_pixel(p, lo, hi, [&](uint32_t r, uint32_t g, uint32_t b) {
        if (KIND == 0) {
            const int t = ty[0][r] + ty[1][g] + ty[2][b];
            if (!(((t + (1 << 11)) >> 12) < y_thr)) return;
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
    });
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kSelTargets; ++t) {
        if (!own[t]) continue;
        for (int i = threadIdx.x; i < kDigitBins; i += GT) {
            const unsigned v = bins[t][i];
            if (v) atomicAdd(&st.sel[t][i], v);
        }
    }
    if (big_last_workgroup(st)) {
        big_select_step(st);
        __threadfence();  // the step's plain stores (and the cleared counters) reach the other XCDs' view before `gen` moves
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&st.gen, my_gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
        while (__hip_atomic_load(&st.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    my_gen += 1u;
    if (__hip_atomic_load(&st.shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) break;  // (uniform) all 64 bits fixed
    __syncthreads();  // (the counters are cleared at the top of the next pass)
    }
}

// one workgroup per image: walk each target's merged counts to the bin holding its rank, fix that digit, move to the next digit
__device__ void big_select_step(BigState& st) {
    __shared__ unsigned part[256];
    __shared__ unsigned found_bin[kSelTargets];
    __shared__ unsigned long long found_below[kSelTargets];
    const int tid = threadIdx.x;
    const int shift = st.shift;
    constexpr int PER = kDigitBins / 256;
    for (int t = 0; t < kSelTargets; ++t) {
        const unsigned* bins = st.sel[st.row[t]];
        unsigned loc[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            loc[j] = big_ld(&bins[tid * PER + j]);
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

__global__ __launch_bounds__(64) void big_vectors_kernel(long hw, tia_stain_params prm, const tia_stain_tables* __restrict__ tab,
                                                         BigState* __restrict__ states, double* __restrict__ stats) {
    const double* od_lut = tab->od_lut;
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
    // rigorous value bounds from the byte range: od in [od(bmax), od(bmin)] (the OD table is decreasing in the byte)
    double lo[2] = {0.0, 0.0}, hi[2] = {0.0, 0.0};
    const double oa = od_lut[st.bmax], ob = od_lut[st.bmin];
    for (int t = 0; t < 2; ++t) {
        for (int j = 0; j < 3; ++j) {
            const double u = P[j * 2 + t] * oa, w = P[j * 2 + t] * ob;
            lo[t] += u < w ? u : w;
            hi[t] += u < w ? w : u;
        }
        const double pad = 1e-9 * (fabs(lo[t]) + fabs(hi[t])) + 1e-12;
        lo[t] -= pad;
        hi[t] += pad;
        if (!(hi[t] > lo[t])) hi[t] = lo[t] + 1.0;
    }
    big_start_selection(st, ranks, 2, lo, hi);
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
    // diagnostics of this path in the instrumentation slots: per selection, whether it fell back to the radix passes and the sizes
    // of its candidate lists
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 1 + kSelTargets; ++i) out[TIA_ST_CYCLES + k * (1 + kSelTargets) + i] = (double)st.diag[k][i];
}

// [states][candidate lists: 4 x kCandCap keys per image][bin codes: 2 x pixels uint16 per image]
static size_t big_states_bytes(long n) { return ((size_t)n * sizeof(BigState) + 255) & ~(size_t)255; }
static size_t big_cand_bytes(long n) { return (size_t)n * kSelTargets * (size_t)kCandCap * sizeof(unsigned long long); }
size_t stain_stats_big_workspace_bytes(long n, long hw) {
    return big_states_bytes(n) + big_cand_bytes(n) + (size_t)n * 2u * (size_t)hw * sizeof(uint16_t);
}

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
    unsigned long long* cand = reinterpret_cast<unsigned long long*>((char*)d_ws + big_states_bytes(n));
    uint16_t* codes = reinterpret_cast<uint16_t*>((char*)d_ws + big_states_bytes(n) + big_cand_bytes(n));
    if (hipMemsetAsync(states, 0, big_states_bytes(n), st) != hipSuccess) return TIA_ELAUNCH;
    const int groups = big_groups(n, hw);
    const dim3 grid((unsigned)groups, (unsigned)n), one((unsigned)n);
    const int z1 = prm.zero_to_one != 0 ? 1 : 0;
    hipLaunchKernelGGL(big_hist_kernel, grid, dim3(GT), 0, st, d_img, hw, states, z1, prm, d_stats);
    hipLaunchKernelGGL(big_p1_finish_kernel, one, dim3(GT), 0, st, hw, prm, states, d_stats);
    // the radix fall-back is launched unconditionally (the host does not know whether a list overflowed) and returns at once in the
    // common case; all its digit passes run inside one launch, on few workgroups per image (see the kernel)
    const dim3 grid_fb((unsigned)(groups > 256 ? 256 : groups), (unsigned)n);
    if (prm.mode == TIA_MODE_MACENKO) {
        hipLaunchKernelGGL(big_moments_kernel, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states, d_stats);
        hipLaunchKernelGGL(big_eigen_kernel, one, dim3(GT), 0, st, groups, prm, states, d_stats);
        hipLaunchKernelGGL(big_lin_sweep_kernel<0>, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states, codes);
        hipLaunchKernelGGL(big_lin_step_kernel, one, dim3(GT), 0, st, states);
        hipLaunchKernelGGL(big_collect_sweep_kernel<0>, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states, codes, cand);
        hipLaunchKernelGGL(big_exact_step_kernel, dim3((unsigned)n, kSelTargets), dim3(GT), 0, st, states, cand);
        hipLaunchKernelGGL(big_select_sweep_kernel<0>, grid_fb, dim3(GT), 0, st, d_img, hw, d_tables, prm, states);  // (returns at once unless a list overflowed)
    }
    hipLaunchKernelGGL(big_vectors_kernel, one, dim3(64), 0, st, hw, prm, d_tables, states, d_stats);
    hipLaunchKernelGGL(big_lin_sweep_kernel<1>, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states, codes);
    hipLaunchKernelGGL(big_lin_step_kernel, one, dim3(GT), 0, st, states);
    hipLaunchKernelGGL(big_collect_sweep_kernel<1>, grid, dim3(GT), 0, st, d_img, hw, d_tables, prm, states, codes, cand);
    hipLaunchKernelGGL(big_exact_step_kernel, dim3((unsigned)n, kSelTargets), dim3(GT), 0, st, states, cand);
    hipLaunchKernelGGL(big_select_sweep_kernel<1>, grid_fb, dim3(GT), 0, st, d_img, hw, d_tables, prm, states);
    hipLaunchKernelGGL(big_final_kernel, one, dim3(64), 0, st, prm, states, d_stats);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia
