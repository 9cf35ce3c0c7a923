// Per-patch stain statistics on gfx950: one 512-thread workgroup per patch (two resident per CU), all per-patch
// state in LDS, per-patch statistics in f64.  See include/tiatoolbox_amd.h for the contract
// and DESIGN.md ("stain_stats") for the pass structure:
//   P1 byte histogram -> contrast-enhancer percentiles -> folded luminance tables
//   P2 tissue mask + OD moments (f64) -> covariance -> 3x3 eigen-decomposition
//   P3/P4 exact angular percentiles (histogram refine + LDS bitonic select) on a monotone
//         pseudo-angle key (one f64 division per pixel instead of atan2)
//   P5/P6 exact 99th percentile of both stain concentrations
// Reference: tools/stainextract.py:177-227, tools/stainnorm.py:49-66,81-85,103,
//            utils/misc.py:261-290,405-444, utils/transforms.py:209-231.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

// numpy evaluates these expressions without fused multiply-add; keep the per-patch
// statistics free of contraction so table entries / lerps round exactly as the reference.
#pragma clang fp contract(off)

namespace tia {

constexpr int NT = 512;          // threads per workgroup; two workgroups per CU (4 waves/SIMD, 128 VGPRs)
constexpr int NW = NT / 64;
constexpr int NB = 4096;         // histogram bins per selection target per level
constexpr int CAP = 1024;        // candidates sorted in LDS
constexpr int MAXLEVEL = 7;      // 4096^6 > 2^64: deeper levels cannot split an f64 range further
constexpr int BPT = NB / NT;     // bins per thread in the scan
// P2 records the tissue mask as bits in LDS (8 lanes x 4 pixels = one word); the window sweeps of the angular
// selection test a bit instead of repeating three table look-ups per pixel.
constexpr int MASK_WORDS = 2048;  // tissue-mask bits of patches up to 65536 pixels (256x256); larger ones recompute
constexpr int SNB = 1024;         // bins of the sample histograms that place the selection windows
constexpr int SAMPLE_TARGET = 4096;  // pixels sampled to place a window
constexpr int MODE_VTAIL = 100;      // internal mode of stain_stats_kernel<false>: second launch of the Vahadane pair (below)

struct SelState {
    double lo[2][MAXLEVEL + 1];
    double scale[2][MAXLEVEL + 1];
    int sel[2][MAXLEVEL + 1];
    double olo[2], ohi[2];  // rigorous bounds of the current member set (edge bins are open-ended)
    int level[2];
    int collapsed[2];
    int need_hist[2];
    unsigned long long r[2];
    unsigned long long cnt[2];
    unsigned ncand[2];
    int fast;          // 1: collect through the per-pixel bin cache
    int sel_hi[2];     // last bin collected on the fast path (next non-empty bin when k+1 leaves the bin)
    unsigned long long above_key[2];
    unsigned long long member_key[2];
};

#ifndef TIA_OD_REP
#define TIA_OD_REP 1
#endif
constexpr int ODR = TIA_OD_REP;  // copies of the f64 OD table: lane l reads copy l%ODR, which spreads the
                                 // data-dependent look-ups over the LDS banks (the kernel is LDS-bound)

struct Smem {
    double od[256 * ODR];
    int ty[3][256];
    unsigned hist[256];
    unsigned hist3[3][256];
    unsigned cum[256];
    unsigned bins[2][NB];
    double cand[2][CAP];
    double red[NW][16];
    double red16[16][10];  // P2: partial sums of the 16 virtual waves (canonical order shared with the register-resident kernel)
    unsigned wtot[NW];
    SelState st;
    double bc[48];
    double chm[6];      // per-channel sum(od), sum(od^2) over all pixels
    unsigned long long ubc[8];
    int ibc[8];
    unsigned mbits[MASK_WORDS];  // tissue mask bits of the patch (when it fits)
    unsigned sbins[2][SNB];      // sample histograms (window placement)
    double wlo[2], whi[2];       // selection windows: candidates have wlo <= key <= whi
    double smin[2], sscale[2];   // sample histogram binning
    unsigned long long wbelow[2];
    unsigned wn[2];
    int wok;
    unsigned wcnt[NW];           // entries in each wave's private segment of the sweep list
#if TIA_STATS_TIMING
    long long tm[16];   // per-phase cycle accumulators (thread 0)
    long long tlast;
#endif
};

// Phase timing (developer builds only: -DTIA_STATS_TIMING=1, see build.build(defines=...)): thread 0 adds the shader-clock
// cycles since the previous stamp to slot `i` and the totals land in the statistics record (TIA_ST_CYCLES).  The product
// library is built without it: no clock reads, no extra live state in the kernel.
#ifndef TIA_STATS_TIMING
#define TIA_STATS_TIMING 0
#endif
__device__ __forceinline__ void stamp(Smem& s, int i) {
#if TIA_STATS_TIMING
    if (threadIdx.x == 0) {
        const long long now = clock64();
        s.tm[i] += now - s.tlast;
        s.tlast = now;
    }
#else
    (void)s;
    (void)i;
#endif
}
enum { TM_P1 = 0, TM_LUT, TM_P2, TM_EIG, TM_SEL_HIST, TM_SEL_FIND, TM_SEL_COLLECT, TM_SEL_SORT, TM_PHI_TOTAL,
       TM_CONC_TOTAL, TM_TOTAL };

// ---------------------------------------------------------------------------------------
// block-wide helpers (all threads must call)
// ---------------------------------------------------------------------------------------
template <int N, class SM>
__device__ __forceinline__ void block_sum(double (&v)[N], SM& s) {
    static_assert(N <= 16, "reduction scratch too small");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double w = wave_sum(v[i]);
        if (lane_id() == 0) s.red[wave_id()][i] = w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        for (int w = 0; w < NW; ++w) acc += s.red[w][i];  // fixed order: deterministic
        v[i] = acc;
    }
    __syncthreads();
}

// Projections and moment accumulations use explicit fused multiply-adds: f64 runs at half rate on gfx950 and
// these sweeps are VALU-bound, so a*b+c as one instruction is a third fewer issue slots.  The reference's
// BLAS dot products fix no particular rounding order either; what matters is that the histogram pass and
// the collect pass evaluate a pixel's key with the SAME instruction sequence, hence one shared helper.
__device__ __forceinline__ double dot3(double x, double y, double z, double a, double b, double c) {
    return __builtin_fma(z, c, __builtin_fma(y, b, x * a));
}

// numpy's _lerp (numpy/lib/_function_base_impl.py): a + (b-a)*t, or b - (b-a)*(1-t) for t>=0.5
__device__ __forceinline__ double np_lerp(double a, double b, double t) {
    const double d = b - a;
    return (t >= 0.5) ? (b - d * (1.0 - t)) : (a + d * t);
}

// numpy 'linear' percentile index: vi=(n-1)*q; prev=floor(vi), next=prev+1 (clamped), gamma
__device__ __forceinline__ void np_index(unsigned long long n, double q, unsigned long long& prev,
                                         unsigned long long& next, double& gamma) {
    const double vi = (double)(n - 1) * q;
    if (vi >= (double)(n - 1)) {
        // numpy (_get_indexes): both neighbours become the last element, so the lerp weight is moot
        prev = next = n - 1;
        gamma = 0.0;
        return;
    }
    const double fl = floor(vi);
    prev = (unsigned long long)fl;
    next = prev + 1;
    gamma = vi - fl;
}

__device__ __forceinline__ int bin_of(double x, double lo, double scale) {
    const double d = (x - lo) * scale;
    if (!(d >= 0.0)) return 0;
    if (d >= (double)NB) return NB - 1;
    return (int)d;
}

// Find the bin holding 0-based rank r in bins[NB]; thread that owns it publishes
// (bin, rank-within-bin, bin count) through ibc/ubc.  All threads call; result visible after return.
__device__ __forceinline__ void find_bin(const unsigned* __restrict__ bins, unsigned long long r,
                                         Smem& s, int slot) {
    unsigned local[BPT];
    unsigned sum = 0;
    const int base = threadIdx.x * BPT;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        local[i] = bins[base + i];
        sum += local[i];
    }
    unsigned incl = wave_incl_scan_u32(sum);
    if (lane_id() == 63) s.wtot[wave_id()] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (int w = 0; w < wave_id(); ++w) before += s.wtot[w];
    before += incl - sum;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        if (r >= before && r < before + local[i]) {
            s.ibc[slot] = base + i;
            s.ubc[slot * 2 + 0] = r - before;
            s.ubc[slot * 2 + 1] = local[i];
        }
        before += local[i];
    }
    __syncthreads();
}

// Exact order statistics sorted[k] and sorted[k+1] (k+1 clamped to n-1) for up to two targets in
// one sweep family.  `valf(idx,r,g,b,x)` returns a 2-bit validity mask and fills x[0], x[1].
// Multi-level linear-histogram refinement over pixel passes until the bin holding rank k has
// <= CAP members, then one collect pass + an LDS bitonic sort.  Everything is exact: the bin
// function is monotone in x, so bins partition the sorted order.
template <class VF, class H0>
__device__ __forceinline__ void select2(const uint8_t* __restrict__ p, long hw, VF&& valf, H0&& hist0, Smem& s,
                        const unsigned long long (&k)[2], const unsigned long long (&n)[2],
                        const double (&lo0)[2], const double (&hi0)[2], const double (&olo0)[2],
                        const double (&ohi0)[2], bool shared_values, uint16_t* __restrict__ bincache,
                        double (&vprev)[2], double (&vnext)[2]) {
    SelState& st = s.st;
    // `bincache` ([2][hw] uint16, may be null): the level-0 bin of every pixel, written by the first
    // histogram pass, lets the collect pass skip the value computation for everything but the few
    // members of the selected bin(s).
    const bool can_cache = bincache != nullptr && (hw & 3) == 0;
    if (threadIdx.x == 0) st.fast = 0;
    if (threadIdx.x < 2) {
        const int t = threadIdx.x;
        st.level[t] = 0;
        st.cnt[t] = n[t];
        st.r[t] = k[t];
        st.lo[t][0] = lo0[t];
        st.olo[t] = olo0[t];
        st.ohi[t] = ohi0[t];
        const double sc = (double)NB / (hi0[t] - lo0[t]);
        const bool ok = (hi0[t] > lo0[t]) && (sc > 0.0) && (sc < 1.0e300);
        st.scale[t][0] = ok ? sc : 0.0;
        st.collapsed[t] = ok ? 0 : 1;
    }
    __syncthreads();

    for (int iter = 0; iter < MAXLEVEL; ++iter) {
        if (threadIdx.x < 2) {
            const int t = threadIdx.x;
            st.need_hist[t] = (st.cnt[t] > (unsigned long long)CAP && !st.collapsed[t] &&
                               st.level[t] < MAXLEVEL) ? 1 : 0;
        }
        __syncthreads();
        const int nh0 = st.need_hist[0], nh1 = st.need_hist[1];
        if (!nh0 && !nh1) break;
        stamp(s, TM_SEL_SORT);
        const int lv0 = st.level[0], lv1 = st.level[1];
        // one shared histogram while both targets still see the same values and the same binning
        const bool shared = shared_values && nh0 && nh1 && lv0 == 0 && lv1 == 0;
        for (int i = threadIdx.x; i < NB; i += NT) {
            s.bins[0][i] = 0;
            s.bins[1][i] = 0;
        }
        __syncthreads();
        // values outside the histogram window belong to the (open-ended) edge bins; they are counted
        // in registers so that e.g. a large background population does not serialise on one address
        unsigned below[2] = {0, 0}, above[2] = {0, 0};
        const bool write_cache = can_cache && iter == 0;
        unsigned long long codes[2] = {0ull, 0ull};
        // straight-line level-0 pass supplied by the caller (4 pixels per step, all table look-ups issued
        // together); the generic per-pixel loop below handles every other case
        const bool handled = write_cache && nh0 && nh1 && hist0(below, above);
        if (!handled)
        for_each_pixel_w<NT>(p, hw, [&](long idx, uint32_t r, uint32_t g, uint32_t b, const WaveGroup& wg) {
            double x[2];
            const unsigned vm = valf(idx, r, g, b, x);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int nh = t ? nh1 : nh0;
                const int lv = t ? lv1 : lv0;
                if (!nh || (shared && t == 1)) continue;  // wave-uniform
                bool member = ((vm >> t) & 1u) != 0;
                for (int l = 0; l < lv && member; ++l)
                    member = bin_of(x[t], st.lo[t][l], st.scale[t][l]) == st.sel[t][l];
                const double d = (x[t] - st.lo[t][lv]) * st.scale[t][lv];
                const bool lowv = member && !(d >= 0.0);
                const bool highv = member && (d >= (double)NB);
                below[t] += lowv ? 1u : 0u;
                above[t] += highv ? 1u : 0u;
                hist_add(s.bins[t], (int)d, member && !lowv && !highv, wg);
                if (write_cache) {
                    const unsigned code = !member ? 0xffffu : (lowv ? 0u : (highv ? (unsigned)(NB - 1) : (unsigned)(int)d));
                    codes[t] |= (unsigned long long)code << (16 * (int)(idx & 3));
                    if ((idx & 3) == 3) {
                        *reinterpret_cast<unsigned long long*>(bincache + (size_t)t * hw + (idx - 3)) = codes[t];
                        codes[t] = 0ull;
                    }
                }
            }
        });
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (below[t]) atomicAdd(&s.bins[t][0], below[t]);
            if (above[t]) atomicAdd(&s.bins[t][NB - 1], above[t]);
        }
        __syncthreads();
        stamp(s, TM_SEL_HIST);
        for (int t = 0; t < 2; ++t) {
            if (!(t ? nh1 : nh0)) continue;
            const unsigned* hb = (shared && t == 1) ? s.bins[0] : s.bins[t];
            find_bin(hb, st.r[t], s, t);
            if (threadIdx.x == 0) {
                const int lv = st.level[t];
                const int b = s.ibc[t];
                st.sel[t][lv] = b;
                st.r[t] = s.ubc[t * 2 + 0];
                st.cnt[t] = s.ubc[t * 2 + 1];
                const double lo = st.lo[t][lv], sc = st.scale[t][lv];
                // edge bins also hold everything clamped into them: extend to the rigorous bound
                const double nlo = (b == 0) ? st.olo[t] : lo + (double)b / sc;
                const double nhi = (b == NB - 1) ? st.ohi[t] : lo + (double)(b + 1) / sc;
                st.olo[t] = nlo;
                st.ohi[t] = nhi;
                const double nsc = (double)NB / (nhi - nlo);
                const bool ok = (nhi > nlo) && (nsc > 0.0) && (nsc < 1.0e300);
                st.lo[t][lv + 1] = nlo;
                st.scale[t][lv + 1] = ok ? nsc : 0.0;
                if (!ok) st.collapsed[t] = 1;
                st.level[t] = lv + 1;
            }
            __syncthreads();
        }
        // fast path: exactly one histogram level for every target that needed one -> the cached bins are
        // exactly the membership test.  If rank k is the last member of its bin, the bin holding k+1 (the
        // next non-empty one) is collected too, so no separate "minimum above" search is needed.
        if (write_cache && nh0 && nh1) {
            if (threadIdx.x < 2) st.sel_hi[threadIdx.x] = NB;
            __syncthreads();
            bool want[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                want[t] = st.level[t] == 1 && st.cnt[t] <= (unsigned long long)CAP && st.r[t] + 1 == st.cnt[t] &&
                          k[t] + 1 < n[t];
                if (want[t]) {
                    const unsigned* hb = (shared && t == 1) ? s.bins[0] : s.bins[t];
                    const int sel = st.sel[t][0];
                    int first = NB;
                    for (int i = threadIdx.x * BPT; i < threadIdx.x * BPT + BPT; ++i)
                        if (i > sel && hb[i] != 0 && i < first) first = i;
                    if (first < NB) atomicMin(&st.sel_hi[t], first);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int ok = 1;
                for (int t = 0; t < 2; ++t) {
                    if (st.level[t] != 1 || st.cnt[t] > (unsigned long long)CAP) ok = 0;
                    if (want[t]) {
                        const unsigned* hb = (shared && t == 1) ? s.bins[0] : s.bins[t];
                        if (st.sel_hi[t] >= NB || st.cnt[t] + hb[st.sel_hi[t]] > (unsigned long long)CAP) ok = 0;
                        else st.cnt[t] += hb[st.sel_hi[t]];
                    } else {
                        st.sel_hi[t] = st.sel[t][0];
                    }
                }
                st.fast = ok;
                if (!ok)  // restore the counts the generic path expects
                    for (int t = 0; t < 2; ++t)
                        if (want[t] && st.sel_hi[t] < NB) {
                            const unsigned* hb = (shared && t == 1) ? s.bins[0] : s.bins[t];
                            if (st.cnt[t] > hb[st.sel[t][0]]) st.cnt[t] = hb[st.sel[t][0]];
                        }
            }
            __syncthreads();
        }
    }

    // ---- collect pass ------------------------------------------------------------------
    if (threadIdx.x < 2) {
        const int t = threadIdx.x;
        st.ncand[t] = 0;
        st.above_key[t] = ~0ull;
        st.member_key[t] = ~0ull;
    }
    __syncthreads();
    if (st.fast) {
        // group-level sweep over the cached bins only: pixel bytes are fetched for the (rare) members
        const int slo[2] = {st.sel[0][0], st.sel[1][0]}, shi[2] = {st.sel_hi[0], st.sel_hi[1]};
        const unsigned long long* c0p = reinterpret_cast<const unsigned long long*>(bincache);
        const unsigned long long* c1p = shared_values ? c0p : reinterpret_cast<const unsigned long long*>(bincache + (size_t)hw);
        const long ng = hw >> 2;
        constexpr int CU4 = 4;  // independent code loads in flight per lane (the loop is pure latency otherwise)
        for (long g0 = threadIdx.x; g0 < ng; g0 += (long)NT * CU4) {
            unsigned long long q0[CU4], q1[CU4];
#pragma unroll
            for (int u = 0; u < CU4; ++u) {
                const long g = g0 + (long)u * NT;
                q0[u] = g < ng ? c0p[g] : ~0ull;
                q1[u] = shared_values ? q0[u] : (g < ng ? c1p[g] : ~0ull);
            }
#pragma unroll
            for (int u = 0; u < CU4; ++u) {
                const long g = g0 + (long)u * NT;
                unsigned hit = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int a0 = (int)((q0[u] >> (16 * i)) & 0xffffu), a1 = (int)((q1[u] >> (16 * i)) & 0xffffu);
                    hit |= (a0 >= slo[0] && a0 <= shi[0]) ? (1u << i) : 0u;
                    hit |= (a1 >= slo[1] && a1 <= shi[1]) ? (16u << i) : 0u;
                }
                if (hit) {
                    for (int i = 0; i < 4; ++i) {
                        if (!((hit >> i) & 0x11u)) continue;
                        const long idx = g * 4 + i;
                        double x[2];
                        valf(idx, (uint32_t)p[3 * idx], (uint32_t)p[3 * idx + 1], (uint32_t)p[3 * idx + 2], x);
                        if ((hit >> i) & 1u) {
                            const unsigned pos = atomicAdd(&st.ncand[0], 1u);
                            if (pos < (unsigned)CAP) s.cand[0][pos] = x[0];
                        }
                        if ((hit >> i) & 16u) {
                            const unsigned pos = atomicAdd(&st.ncand[1], 1u);
                            if (pos < (unsigned)CAP) s.cand[1][pos] = x[1];
                        }
                    }
                }
            }
        }
    } else {
        const int lv[2] = {st.level[0], st.level[1]};
        const bool store[2] = {st.cnt[0] <= (unsigned long long)CAP, st.cnt[1] <= (unsigned long long)CAP};
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        double amin[2] = {inf, inf}, mmin[2] = {inf, inf};
        for_each_pixel<NT>(p, hw, [&](long idx, uint32_t r, uint32_t g, uint32_t b) {
            double x[2];
            const unsigned vm = valf(idx, r, g, b, x);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (!((vm >> t) & 1u)) continue;
                int cls = 0;  // 0 member, 1 above, -1 below
                for (int l = 0; l < lv[t]; ++l) {
                    const int bb = bin_of(x[t], st.lo[t][l], st.scale[t][l]);
                    if (bb != st.sel[t][l]) {
                        cls = bb > st.sel[t][l] ? 1 : -1;
                        break;
                    }
                }
                if (cls == 0) {
                    if (store[t]) {
                        const unsigned pos = atomicAdd(&st.ncand[t], 1u);
                        if (pos < (unsigned)CAP) s.cand[t][pos] = x[t];
                    } else {
                        mmin[t] = x[t] < mmin[t] ? x[t] : mmin[t];
                    }
                } else if (cls > 0) {
                    amin[t] = x[t] < amin[t] ? x[t] : amin[t];
                }
            }
        });
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned long long a = wave_min_u64(f64_key(amin[t]));
            const unsigned long long m = wave_min_u64(f64_key(mmin[t]));
            if (lane_id() == 0) {
                atomicMin(&st.above_key[t], a);
                atomicMin(&st.member_key[t], m);
            }
        }
    }
    __syncthreads();
    stamp(s, TM_SEL_COLLECT);

    // ---- sort candidates (both targets at once) and pick -----------------------------------
    unsigned pmax = 2;
    for (int t = 0; t < 2; ++t) {
        if (st.cnt[t] <= (unsigned long long)CAP) {
            unsigned c = (unsigned)st.cnt[t];
            unsigned pp = 2;
            while (pp < c) pp <<= 1;
            pmax = pp > pmax ? pp : pmax;
        }
    }
    for (int t = 0; t < 2; ++t) {
        if (st.cnt[t] <= (unsigned long long)CAP) {
            for (unsigned i = (unsigned)st.cnt[t] + threadIdx.x; i < pmax; i += NT)
                s.cand[t][i] = __longlong_as_double(0x7ff0000000000000ll);  // +inf padding
        }
    }
    __syncthreads();
    for (unsigned kk = 2; kk <= pmax; kk <<= 1) {
        for (unsigned j = kk >> 1; j > 0; j >>= 1) {
            for (unsigned i = threadIdx.x; i < pmax; i += NT) {
                const unsigned partner = i ^ j;
                if (partner > i) {
                    const bool asc = (i & kk) == 0;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (st.cnt[t] > (unsigned long long)CAP) continue;
                        const double a = s.cand[t][i], b = s.cand[t][partner];
                        if ((a > b) == asc) {
                            s.cand[t][i] = b;
                            s.cand[t][partner] = a;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (n[t] == 0) {
            vprev[t] = vnext[t] = 0.0;
            continue;
        }
        const unsigned long long r = st.r[t], c = st.cnt[t];
        const bool has_next = (k[t] + 1 < n[t]);
        const double above = key_f64(st.above_key[t]);
        if (c <= (unsigned long long)CAP) {
            vprev[t] = s.cand[t][r];
            vnext[t] = !has_next ? vprev[t] : ((r + 1 < c) ? s.cand[t][r + 1] : above);
        } else {
            const double m = key_f64(st.member_key[t]);  // collapsed range: members identical
            vprev[t] = m;
            vnext[t] = !has_next ? m : ((r + 1 < c) ? m : above);
        }
    }
    __syncthreads();
    stamp(s, TM_SEL_SORT);
}


// Sample k of a window-placing sample: one pixel from every window [k stride, (k + 1) stride) of the flat pixel index, at a
// pseudo-random offset inside it (a stratified sample).  A plain multiple of the stride is a set of image COLUMNS whenever the
// stride divides the row length (256 x 256: stride 16 = 16 columns out of 256): neighbouring rows are correlated in real and
// synthetic tissue alike, the sample then carries far fewer independent values than its size, the 3.5-sigma rank window derived
// from that size is too narrow, and the selection has to be redone by the histogram path (measured: 15 % of the selections at
// 256 x 256, none at 224 x 224 where the stride of 13 walks diagonally).  Results never depend on the sample; only the cost does.
__device__ __forceinline__ long sample_index(long k, long stride) {
    const unsigned h = ((unsigned)k * 2654435761u) >> 8;
    return k * stride + (long)(stride > 1 ? h % (unsigned)stride : 0u);
}

// ---------------------------------------------------------------------------------------------------------------------
// Window selection: the same exact order statistics as select2 from ONE sweep over the pixels.
//   1. a <= 4096-pixel sample, evaluated in float32 on the VALU, places per target a key window [wlo, whi] that holds
//      ranks k and k+1 with overwhelming probability (3.5 sigma of the sample-rank distribution, widened to the edges of
//      a 1024-bin sample histogram plus one bin of slack);
//   2. `sweep` classifies EVERY pixel against the windows with float32 arithmetic on the VALU only (no table look-ups):
//      definitely below -> counted, definitely above -> ignored, anything within the float32 error bound of a window
//      edge or inside the window -> its pixel index goes to an LDS list;
//   3. the listed pixels (a few per cent) get their exact float64 key (`exact`) and are classified exactly: below /
//      above / candidate;
//   4. ranks k, k+1 must fall inside the candidate set (checked from the exact counts); a 1024-bin histogram of the
//      candidates then isolates the one or two bins holding them and a single wave orders those few values.
// Whenever a precondition fails (sample too small, list or candidate overflow, ranks outside the window, a crowded bin)
// the function returns false and the caller runs select2.  Results never depend on the sample or on float32 rounding --
// only the cost does (tests: bitwise audit of both paths).
template <class SAMPLE32, class EXACT, class SWEEP>
__device__ __forceinline__ bool window_select2(const uint8_t* __restrict__ p, long hw, SAMPLE32&& sample32, EXACT&& exact,
                                               SWEEP&& sweep, Smem& s, const unsigned long long (&k)[2],
                                               const unsigned long long (&n)[2], double (&vprev)[2], double (&vnext)[2]) {
    const int tid = threadIdx.x;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const float finf = __int_as_float(0x7f800000);
    if (n[0] == 0 || n[1] == 0 || !groups_ok(p, hw) || hw >= (1L << 24)) return false;  // the sweep works on 4-pixel groups
    constexpr int SPT = SAMPLE_TARGET / NT;  // samples per thread
    const long stride = (hw + SAMPLE_TARGET - 1) / SAMPLE_TARGET;
    if (tid < 2) {
        s.st.above_key[tid] = 0ull;     // running max (as key)
        s.st.member_key[tid] = ~0ull;   // running min (as key)
        s.wn[tid] = 0u;
        s.wbelow[tid] = 0ull;
    }
    for (int i = tid; i < 2 * SNB; i += NT) (&s.sbins[0][0])[i] = 0u;
    // ---- sample (float32): all byte loads in flight together; the values wait in the (still unused) histogram area --------
    float* sbuf = reinterpret_cast<float*>(&s.bins[0][0]);  // [2][SAMPLE_TARGET]; NaN = not a member
    const float fnan = __int_as_float(0x7fc00000);
    {
        uint32_t rgb[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            // branch-free (clamped index): a conditional load would be waited for on its own, one memory latency per sample
            const long idx = sample_index((long)j * NT + tid, stride);
            const long ic = idx < hw ? idx : hw - 1;
            rgb[j] = (uint32_t)p[3 * ic] | ((uint32_t)p[3 * ic + 1] << 8) | ((uint32_t)p[3 * ic + 2] << 16);
        }
        float mn[2] = {finf, finf}, mx[2] = {-finf, -finf};
        unsigned cnt[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const long idx = sample_index((long)j * NT + tid, stride);
            float v[2] = {0.0f, 0.0f};
            const unsigned valid = idx < hw ? sample32(idx, rgb[j] & 255u, (rgb[j] >> 8) & 255u, (rgb[j] >> 16) & 255u, v) : 0u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool ok = (valid >> t) & 1u;
                sbuf[t * SAMPLE_TARGET + j * NT + tid] = ok ? v[t] : fnan;
                mn[t] = ok ? fminf(mn[t], v[t]) : mn[t];
                mx[t] = ok ? fmaxf(mx[t], v[t]) : mx[t];
                cnt[t] += ok ? 1u : 0u;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn[t] = fminf(mn[t], __shfl_down(mn[t], o, 64));
                mx[t] = fmaxf(mx[t], __shfl_down(mx[t], o, 64));
                cnt[t] += __shfl_down(cnt[t], o, 64);
            }
        }
        __syncthreads();  // zeroing above done
        if (lane_id() == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                atomicMin(&s.st.member_key[t], f64_key((double)mn[t]));
                atomicMax(&s.st.above_key[t], f64_key((double)mx[t]));
                atomicAdd(&s.wn[t], cnt[t]);
            }
        }
    }
    __syncthreads();
    const unsigned ns[2] = {s.wn[0], s.wn[1]};
    if (ns[0] < 64u || ns[1] < 64u) {  // too small to place a window: the histogram path handles it
#if TIA_STATS_TIMING
        if (tid == 0) s.tm[13] += 2;
#endif
        return false;
    }
    float smin[2], sscale[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const double lo = key_f64(s.st.member_key[t]), hi = key_f64(s.st.above_key[t]);
        const double sc = (double)SNB / (hi - lo);
        smin[t] = (float)lo;
        sscale[t] = (hi > lo && sc > 0.0 && sc < 1.0e30) ? (float)sc : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
        for (int i = tid; i < SAMPLE_TARGET; i += NT) {
            const float v = sbuf[t * SAMPLE_TARGET + i];
            if (v == v) {
                const float d = (v - smin[t]) * sscale[t];
                const int b = !(d >= 0.0f) ? 0 : (d >= (float)SNB ? SNB - 1 : (int)d);
                atomicAdd(&s.sbins[t][b], 1u);
            }
        }
    __syncthreads();
    // ---- windows: wave t places the window of target t ------------------------------------------------------------------
    constexpr int PER = SNB / 64;
    // bin holding rank r (0-based) of a 1024-bin histogram held 16 bins per lane: first bin whose inclusive count exceeds r
    auto bin_of_rank = [&](const unsigned (&local)[PER], unsigned incl, unsigned sum, unsigned r, unsigned& before_bin) -> int {
        unsigned before = incl - sum;
        int found = SNB;
        unsigned fb = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const unsigned after = before + local[i];
            if (found == SNB && after > r) {  // after > r >= before implies local[i] != 0
                found = lane_id() * PER + i;
                fb = before;
            }
            before = after;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(found, o, 64);
            const unsigned ob = __shfl_xor(fb, o, 64);
            if (other < found) {
                found = other;
                fb = ob;
            }
        }
        before_bin = fb;
        return found;
    };
    if (wave_id() < 2) {
        const int t = wave_id();
        const int lane = lane_id();
        unsigned local[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            local[i] = s.sbins[t][lane * PER + i];
            sum += local[i];
        }
        const unsigned incl = wave_incl_scan_u32(sum);
        const double q = ((double)k[t] + 0.5) / (double)n[t];
        const double centre = q * (double)ns[t];
        const double sigma = sqrt((double)ns[t] * q * (1.0 - q));
        const double rlo = floor(centre - 3.5 * sigma - 2.0), rhi = ceil(centre + 3.5 * sigma + 2.0);
        unsigned dummy;
        const int blo = rlo < 0.0 ? -1 : bin_of_rank(local, incl, sum, (unsigned)rlo, dummy);
        const int bhi = rhi >= (double)ns[t] ? SNB : bin_of_rank(local, incl, sum, (unsigned)rhi, dummy);
        if (lane == 0) {
            const double sc = (double)sscale[t];
            const bool flat = !(sc > 0.0);
            // one extra bin of slack on either side; the outermost bins are open-ended
            s.wlo[t] = (flat || blo <= 1) ? -inf : (double)smin[t] + (double)(blo - 1) / sc;
            s.whi[t] = (flat || bhi >= SNB - 2) ? inf : (double)smin[t] + (double)(bhi + 2) / sc;
        }
    }
    __syncthreads();
    if (tid < 2) {
        s.wn[tid] = 0u;
        s.st.ncand[tid] = 0u;
        s.st.above_key[tid] = 0ull;
        s.st.member_key[tid] = ~0ull;
    }
    for (int i = tid; i < 2 * SNB; i += NT) (&s.sbins[0][0])[i] = 0u;
    __syncthreads();
    stamp(s, TM_SEL_FIND);
    // ---- the float32 sweep: counts "definitely below", lists everything within the error bound of a window -------------
    // The list lives in the histogram area; every wave appends to its own segment with a register-resident count (no
    // atomics, nothing to wait for in the hot loop).  Entry = pixel-group index | 8 need-bits << 22 (target t of pixel i
    // of the group: bit 2i+t).
    unsigned* list = &s.bins[0][0];
    constexpr unsigned SEG = 2u * NB / NW;
    sweep(list, SEG);
    __syncthreads();
    stamp(s, TM_SEL_HIST);
    {
        bool over = false;
        for (int w = 0; w < NW; ++w) over = over || s.wcnt[w] > SEG;
#if TIA_STATS_TIMING
        if (over && tid == 0) s.tm[13] += 30;
#endif
        if (over) return false;  // uniform
    }
    // ---- exact classification of the listed pixels ----------------------------------------------------------------------
    {
        unsigned bl[2] = {0u, 0u};
        unsigned long long mn[2] = {~0ull, ~0ull}, mx[2] = {0ull, 0ull};
        unsigned pre[NW + 1];
        pre[0] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) pre[w + 1] = pre[w] + s.wcnt[w];
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
        const unsigned total = pre[NW];
        constexpr int EU = 4;  // listed groups per thread whose pixel words are fetched together (one latency, not four)
        for (unsigned i0 = tid; i0 < total; i0 += NT * EU) {
            unsigned ent[EU];
            uint32_t wa[EU], wb[EU], wc[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const unsigned i = i0 + (unsigned)u * NT;
                const unsigned ic = i < total ? i : total - 1;
                int w = 0;
                unsigned base = 0;
#pragma unroll
                for (int v = 1; v < NW; ++v)
                    if (ic >= pre[v]) {
                        w = v;
                        base = pre[v];
                    }
                ent[u] = i < total ? list[w * SEG + (ic - base)] : 0u;  // 0: no need-bits
            }
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const long g = (long)(ent[u] & 0x3fffffu);
                wa[u] = q[g * 3];
                wb[u] = q[g * 3 + 1];
                wc[u] = q[g * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const unsigned e = ent[u];
                const long g = (long)(e & 0x3fffffu);
                uint32_t rr[4], gg[4], bb[4];
                unpack_group(wa[u], wb[u], wc[u], rr, gg, bb);
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const unsigned need = (e >> (22 + 2 * px)) & 3u;
                    if (!need) continue;
                    double x[2];
                    const unsigned vm = exact(g * 4 + px, rr[px], gg[px], bb[px], x);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (!((need >> t) & 1u) || !((vm >> t) & 1u)) continue;
                        if (x[t] < s.wlo[t]) {
                            ++bl[t];
                        } else if (!(x[t] > s.whi[t])) {
                            const unsigned pos = atomicAdd(&s.wn[t], 1u);
                            if (pos < (unsigned)CAP) s.cand[t][pos] = x[t];
                            const unsigned long long key = f64_key(x[t]);
                            mn[t] = key < mn[t] ? key : mn[t];
                            mx[t] = key > mx[t] ? key : mx[t];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned c = bl[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
            const unsigned long long a2 = wave_min_u64(mn[t]);
            const unsigned long long b2 = ~wave_min_u64(~mx[t]);
            if (lane_id() == 0) {
                if (c) atomicAdd(&s.wbelow[t], (unsigned long long)c);
                atomicMin(&s.st.member_key[t], a2);
                atomicMax(&s.st.above_key[t], b2);
            }
        }
    }
    __syncthreads();
    stamp(s, TM_SEL_COLLECT);
    if (tid == 0) {
        int ok = 1;
        for (int t = 0; t < 2; ++t) {
            const unsigned long long below = s.wbelow[t], nc = s.wn[t];
            const bool has_next = k[t] + 1 < n[t];
            if (nc > (unsigned long long)CAP || k[t] < below || k[t] + (has_next ? 1 : 0) >= below + nc) ok = 0;
#if TIA_STATS_TIMING
            if (nc > (unsigned long long)CAP) s.tm[13] += 400;
            else if (k[t] < below || k[t] + (has_next ? 1 : 0) >= below + nc) s.tm[13] += 5000;
            s.tm[14] += (long long)nc;
#endif
        }
        s.wok = ok;
    }
    __syncthreads();
    if (!s.wok) return false;
    // ---- refine inside the candidate set: histogram -> the bin(s) of local ranks r, r+1 -> one wave orders them ----------
    const unsigned nc[2] = {s.wn[0], s.wn[1]};
    double clo[2], csc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const double lo = key_f64(s.st.member_key[t]), hi = key_f64(s.st.above_key[t]);
        const double sc = (double)SNB / (hi - lo);
        clo[t] = lo;
        csc[t] = (hi > lo && sc > 0.0 && sc < 1.0e300) ? sc : 0.0;
    }
    auto cbin = [&](int t, double x) -> int {
        const double d = (x - clo[t]) * csc[t];
        return !(d >= 0.0) ? 0 : (d >= (double)SNB ? SNB - 1 : (int)d);
    };
#pragma unroll
    for (int t = 0; t < 2; ++t)
        for (unsigned i = tid; i < nc[t]; i += NT) atomicAdd(&s.sbins[t][cbin(t, s.cand[t][i])], 1u);
    if (tid < 2) s.st.ncand[tid] = 0u;
    __syncthreads();
    if (wave_id() < 2) {
        const int t = wave_id();
        const int lane = lane_id();
        unsigned local[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            local[i] = s.sbins[t][lane * PER + i];
            sum += local[i];
        }
        const unsigned incl = wave_incl_scan_u32(sum);
        const unsigned long long r = k[t] - s.wbelow[t];
        const bool has_next = k[t] + 1 < n[t];
        unsigned before_a = 0, before_b = 0;
        const int ba = bin_of_rank(local, incl, sum, (unsigned)r, before_a);
        const int bb = has_next ? bin_of_rank(local, incl, sum, (unsigned)r + 1u, before_b) : ba;
        if (lane == 0) {
            s.st.sel[t][0] = ba;
            s.st.sel_hi[t] = bb;
            s.st.r[t] = r - before_a;  // rank inside the picked set (bins ba and, if different, bb; nothing in between)
        }
    }
    __syncthreads();
    // gather the members of the picked bins (a few values) behind the candidates' own storage: s.red / s.bc are too small,
    // the sample histogram of the OTHER kind is free: reuse s.bins (the list is consumed)
    double* small = reinterpret_cast<double*>(&s.bins[0][0]);  // [2][64]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ba = s.st.sel[t][0], bb = s.st.sel_hi[t];
        for (unsigned i = tid; i < nc[t]; i += NT) {
            const double x = s.cand[t][i];
            const int b = cbin(t, x);
            if (b == ba || b == bb) {
                const unsigned pos = atomicAdd(&s.st.ncand[t], 1u);
                if (pos < 64u) small[t * 64 + pos] = x;
            }
        }
    }
    __syncthreads();
    if (s.st.ncand[0] > 64u || s.st.ncand[1] > 64u) {  // a crowded bin (massive ties): order the whole candidate set instead
        unsigned pmax = 2;
        for (int t = 0; t < 2; ++t) {
            unsigned pp = 2;
            while (pp < nc[t]) pp <<= 1;
            pmax = pp > pmax ? pp : pmax;
        }
        for (int t = 0; t < 2; ++t)
            for (unsigned i = nc[t] + tid; i < pmax; i += NT) s.cand[t][i] = inf;
        __syncthreads();
        for (unsigned kk = 2; kk <= pmax; kk <<= 1) {
            for (unsigned j = kk >> 1; j > 0; j >>= 1) {
                for (unsigned i = tid; i < pmax; i += NT) {
                    const unsigned partner = i ^ j;
                    if (partner > i) {
                        const bool asc = (i & kk) == 0;
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const double a = s.cand[t][i], b = s.cand[t][partner];
                            if ((a > b) == asc) {
                                s.cand[t][i] = b;
                                s.cand[t][partner] = a;
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned long long r = k[t] - s.wbelow[t];
            vprev[t] = s.cand[t][r];
            vnext[t] = (k[t] + 1 < n[t]) ? s.cand[t][r + 1] : vprev[t];
        }
        __syncthreads();
        stamp(s, TM_SEL_SORT);
        return true;
    }
    if (wave_id() < 2) {  // rank by counting inside one wave: value of lane i, number of values ordered before it
        const int t = wave_id();
        const int lane = lane_id();
        const unsigned m = s.st.ncand[t];
        const double x = (unsigned)lane < m ? small[t * 64 + lane] : inf;
        unsigned rank = 0;
        for (unsigned j = 0; j < m; ++j) {
            const double y = small[t * 64 + j];
            rank += (y < x || (y == x && j < (unsigned)lane)) ? 1u : 0u;
        }
        const unsigned long long r = s.st.r[t];
        if ((unsigned)lane < m && rank == (unsigned)r) s.bc[40 + 2 * t] = x;
        if ((unsigned)lane < m && rank == (unsigned)r + 1u) s.bc[41 + 2 * t] = x;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        vprev[t] = s.bc[40 + 2 * t];
        vnext[t] = (k[t] + 1 < n[t]) ? s.bc[41 + 2 * t] : vprev[t];
    }
    __syncthreads();
    stamp(s, TM_SEL_SORT);
    return true;
}

// Monotone pseudo-angle: strictly increasing in atan2(y, x) over (-pi, pi], range [-2, 2].
//   x >= 0:  r            (phi in [-pi/2, pi/2]),   r = y / (|x| + |y|)
//   x <  0:  2 - r (y>=0) or -2 - r (y<0)
// Ordering pixels by this key orders them by phi, so the order statistics are selected on the
// key (1 division) and only the two selected values per percentile are turned back into angles.
__device__ __forceinline__ double pseudo_angle(double y, double x) {
    const double d = fabs(x) + fabs(y);
    if (!(d > 0.0)) return 0.0;  // atan2(0, 0) = 0
    const double r = y / d;
    if (x >= 0.0) return r;
    return (y >= 0.0) ? (2.0 - r) : (-2.0 - r);
}
__device__ double angle_of_key(double k) {
    // inverse of pseudo_angle: (|x|, y) proportional to (1 - |r|, r)
    if (k > 1.0) {
        const double r = 2.0 - k;
        return atan2(r, -(1.0 - fabs(r)));
    }
    if (k < -1.0) {
        const double r = -2.0 - k;
        return atan2(r, -(1.0 - fabs(r)));
    }
    return atan2(k, 1.0 - fabs(k));
}

// 3x3 symmetric eigen-decomposition (cyclic Jacobi, f64).  a = xx,xy,xz,yy,yz,zz.
// Outputs eigenvalues w[3] (unsorted) and eigenvectors as columns of v[3][3].
__device__ void jacobi3(const double (&a6)[6], double (&w)[3], double (&v)[3][3]) {
    double a[3][3] = {{a6[0], a6[1], a6[2]}, {a6[1], a6[3], a6[4]}, {a6[2], a6[4], a6[5]}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p) {
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double sn = t * c;
                const double app = a[p][p], aqq = a[q][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                const int r = 3 - p - q;
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - sn * arq;
                a[r][q] = a[q][r] = sn * arp + c * arq;
                for (int i = 0; i < 3; ++i) {
                    const double vip = v[i][p], viq = v[i][q];
                    v[i][p] = c * vip - sn * viq;
                    v[i][q] = sn * vip + c * viq;
                }
            }
        }
    }
    w[0] = a[0][0];
    w[1] = a[1][1];
    w[2] = a[2][2];
}


// ---------------------------------------------------------------------------------------
// Vahadane: sklearn.decomposition.DictionaryLearning restated (stainextract.py:305-316)
// ---------------------------------------------------------------------------------------
// One target of LassoLars(alpha/N, fit_intercept=False, precompute=gram).fit(D^T, x, Xy=cov) with two atoms, i.e. the
// minimiser of 0.5 w'Gw - c'w + alpha |w|_1 (sklearn scales the squared error by 1/(2N), alpha by 1/N: same problem).
// The lasso path ends at the unique minimiser; with two variables it is one of nine orthant-face minimisers, and the
// global one is the face minimiser that lies in its own (closed) orthant with the smallest objective.
__device__ void lasso2(double g00, double g01, double g11, double c0, double c1, double alpha, double (&w)[2]) {
    auto obj = [&](double a, double b) {
        return 0.5 * (g00 * a * a + 2.0 * g01 * a * b + g11 * b * b) - (c0 * a + c1 * b) + alpha * (fabs(a) + fabs(b));
    };
    double best = 0.0;  // w = 0
    w[0] = 0.0;
    w[1] = 0.0;
    if (fabs(c0) > alpha && g00 > 0.0) {
        const double a = (c0 - (c0 > 0.0 ? alpha : -alpha)) / g00;
        const double f = obj(a, 0.0);
        if (f < best) { best = f; w[0] = a; w[1] = 0.0; }
    }
    if (fabs(c1) > alpha && g11 > 0.0) {
        const double b = (c1 - (c1 > 0.0 ? alpha : -alpha)) / g11;
        const double f = obj(0.0, b);
        if (f < best) { best = f; w[0] = 0.0; w[1] = b; }
    }
    const double det = g00 * g11 - g01 * g01;
    if (det > 0.0) {
        for (int k = 0; k < 4; ++k) {
            const double s0 = (k & 1) ? -1.0 : 1.0, s1 = (k & 2) ? -1.0 : 1.0;
            const double r0 = c0 - alpha * s0, r1 = c1 - alpha * s1;
            const double a = (g11 * r0 - g01 * r1) / det, b = (g00 * r1 - g01 * r0) / det;
            if (a * s0 > 0.0 && b * s1 > 0.0) {
                const double f = obj(a, b);
                if (f < best) { best = f; w[0] = a; w[1] = b; }
            }
        }
    }
}
// counter-based generator for the (rare) "atom never used" branch of _update_dict (:527-536).  The reference leaves
// DictionaryLearning unseeded, so no particular random stream is the right one; this one is a function of
// (seed, patch, iteration, atom, pixel) only, hence deterministic and independent of scheduling.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double unit_open(unsigned long long z) { return ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
__device__ __noinline__ double normal_of(unsigned long long key) {
    const double u1 = unit_open(mix64(key)), u2 = unit_open(mix64(key ^ 0xd1b54a32d192ed03ull));
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// ---------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------
// DL = true: the TIA_MODE_VAHADANE instantiation (dictionary learning instead of the Macenko branch); kept apart so that
// its extra live state does not cost the Macenko / fixed-matrix kernel registers.
// Sweep over a patch TOGETHER WITH its per-pixel float64 pairs (the Vahadane dictionary: 2 x N as double2[N]), software-pipelined:
// a lane owns 4-pixel groups (12 image bytes and 64 contiguous dictionary bytes), and the NEXT group's image words and dictionary
// entries are requested before the current group is processed.  `f(idx, r, g, b, d)` is called per pixel in ascending order
// (the per-pixel sweep's order, so thread-local sums come out bit-identical) and may modify `d`; with STORE every entry of the
// group is written back (entries `f` does not touch -- non-tissue pixels -- are rewritten with what was loaded).  The dictionary-
// learning instantiation runs one workgroup per CU at two waves per SIMD: a sweep that tests the tissue mask first and only then
// asks for the pixel's dictionary entry pays one full memory latency per PIXEL with nothing to hide it behind (measured: 4.5 ms per
// sweep over 8192 x 256^2 for 26 GB/s x ... of traffic); here one latency per group is overlapped with the previous group's work.
template <int NT_, bool LOAD, bool STORE, class F>
__device__ __forceinline__ void for_each_pixel_dict(const uint8_t* __restrict__ p, long hw, double2* __restrict__ dict, F&& f) {
    if (!groups_ok(p, hw)) {
        for (long i = threadIdx.x; i < hw; i += NT_) {
            double2 d = LOAD ? dict[i] : make_double2(0.0, 0.0);
            f(i, (uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2], d);
            if (STORE) dict[i] = d;
        }
        return;
    }
    const long ng = hw >> 2;
    const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
    long g = threadIdx.x;
    uint32_t a = 0, b = 0, c = 0;
    double2 d[4], nd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = nd[i] = make_double2(0.0, 0.0);
    if (g < ng) {
        a = q[g * 3 + 0];
        b = q[g * 3 + 1];
        c = q[g * 3 + 2];
        if (LOAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = dict[g * 4 + i];
        }
    }
    while (g < ng) {
        const long gn = g + NT_;
        uint32_t na = 0, nb = 0, nc = 0;
        if (gn < ng) {
            na = q[gn * 3 + 0];
            nb = q[gn * 3 + 1];
            nc = q[gn * 3 + 2];
            if (LOAD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nd[i] = dict[gn * 4 + i];
            }
        }
        uint32_t rr[4], gg[4], bb[4];
        unpack_group(a, b, c, rr, gg, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) f(g * 4 + i, rr[i], gg[i], bb[i], d[i]);
        if (STORE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dict[g * 4 + i] = d[i];
        }
        a = na;
        b = nb;
        c = nc;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = nd[i];
        g = gn;
    }
}

template <bool DL>
#ifndef TIA_STATS_WPE
#define TIA_STATS_WPE 4  // waves per SIMD the Macenko / fixed-matrix instantiation is compiled for (2 work-groups per CU by LDS)
#endif
#ifndef TIA_STATS_WPE_DL
#define TIA_STATS_WPE_DL 2  // ... and the dictionary-learning instantiation (2: 256 VGPRs, one work-group per CU)
#endif
__global__ __launch_bounds__(NT, DL ? TIA_STATS_WPE_DL : TIA_STATS_WPE) void stain_stats_kernel(const uint8_t* __restrict__ img, long hw,
                                                          const tia_stain_tables* __restrict__ tab,
                                                          tia_stain_params prm,
                                                          double* __restrict__ stats,
                                                          uint16_t* __restrict__ binws,
                                                          double2* __restrict__ dictws,
                                                          const int* __restrict__ redo = nullptr) {
    // second launch behind stain_stats_reg_kernel / vahadane_dl_kernel: only the patches that kernel handed back are recomputed
    // here; the MODE_VTAIL launch is the complement (it completes the records of the patches that were NOT handed back)
    if (redo != nullptr && ((redo[blockIdx.x] == 0) != (prm.mode == MODE_VTAIL))) return;
    __shared__ Smem s;
    const uint8_t* p = img + (size_t)blockIdx.x * (size_t)hw * 3u;
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    uint16_t* bincache = binws ? binws + (size_t)blockIdx.x * (size_t)hw * 2u : nullptr;
    const int tid = threadIdx.x;
    const bool z1 = prm.zero_to_one != 0;

    // TIA_MODE_GIVEN: the caller's per-patch stain matrix arrives in the statistics record itself; MODE_VTAIL (internal): the
    // record comes from vahadane_dl_kernel -- stain matrix, tissue count, iteration count and flags -- and this launch adds the
    // rest (percentiles of the contrast enhancer, pseudo-inverse, concentration percentiles, fused matrix)
    double s_given[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double keep_nt = 0.0, keep_iter = 0.0;
    unsigned keep_flags = 0;
    if (!DL && (prm.mode == TIA_MODE_GIVEN || prm.mode == MODE_VTAIL)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s_given[i] = out[TIA_ST_STAIN + i];
        if (prm.mode == MODE_VTAIL) {
            keep_nt = out[TIA_ST_NTISSUE];
            keep_iter = out[TIA_ST_MINPHI];
            keep_flags = (unsigned)out[TIA_ST_FLAGS];
        }
        __syncthreads();
    }
    if (tid < TIA_STATS_STRIDE) out[tid] = 0.0;
    for (int i = tid; i < 256 * ODR; i += NT) s.od[i] = tab->od_lut[i / ODR];
    const int odl = tid & (ODR - 1);
#define OD(v) s.od[(v) * ODR + odl]
#if TIA_STATS_TIMING
    if (tid == 0) {
        for (int i = 0; i < 16; ++i) s.tm[i] = 0;
        s.tlast = clock64();
    }
    const long long t_begin = clock64();
#endif

    const bool grp = groups_ok(p, hw);
    if constexpr (!DL) {
        // ---- P1: byte histogram of all three channels together (the contrast-enhancer percentiles are over the flattened image;
        //      only the dictionary-learning instantiation needs per-channel sums).  32 copies in the 32 KB bins area: lane l adds
        //      to copy l & 31 of bin v at v * 32 + (l & 31), so the 32 lanes an LDS atomic services together never share a bank
        //      whatever the bytes are -- the layout the register-resident kernel uses (per-wave per-channel copies sat on bank
        //      conflicts of data-dependent addresses: 102 k cycles per 256 x 256 patch against ~25 k, profiles/r04f_*).
        unsigned* hs = &s.bins[0][0];
        for (int i = tid; i < 2 * NB; i += NT) hs[i] = 0u;
        __syncthreads();
        hs += lane_id() & 31;
        auto add = [&](uint32_t v) {
            if (z1) v = v ? v : 1u;
            atomicAdd(hs + v * 32u, 1u);
        };
        if (grp) {
            for_each_group<NT>(p, hw, [&](long, uint32_t a, uint32_t b, uint32_t c, const WaveGroup&) {
                const uint32_t w[3] = {a, b, c};
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) add((w[d] >> (8 * e)) & 255u);
            });
        } else {
            for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
                add(r);
                add(g);
                add(b);
            });
        }
        __syncthreads();
        stamp(s, TM_P1);
        if (tid < 256) {
            unsigned tot = 0;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) tot += (&s.bins[0][0])[tid * 32 + ((c + lane_id()) & 31)];  // rotated: no bank conflicts
            s.hist[tid] = tot;
        }
        __syncthreads();
    } else {
    // ---- P1: per-channel byte histograms (per-wave private copies in the bins area) ------------
    // (Two copies per wave -- even / odd lanes -- were measured: no change, 79.7 k cycles either way; the pass
    //  sits on the LDS atomic issue rate, ~11 cycles per wave instruction per CU, not on address conflicts.)
    unsigned* wh = &s.bins[0][0] + wave_id() * 768;
    for (int i = tid; i < NW * 768; i += NT) (&s.bins[0][0])[i] = 0;
    __syncthreads();
    if (grp) {
        for_each_group<NT>(p, hw, [&](long, uint32_t a, uint32_t b, uint32_t c, const WaveGroup& wg) {
            uint32_t rr[4], gg[4], bb[4];
            unpack_group(a, b, c, rr, gg, bb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (z1) {
                    rr[i] = rr[i] ? rr[i] : 1u;
                    gg[i] = gg[i] ? gg[i] : 1u;
                    bb[i] = bb[i] ? bb[i] : 1u;
                }
                hist_add(wh, (int)rr[i], true, wg);
                hist_add(wh + 256, (int)gg[i], true, wg);
                hist_add(wh + 512, (int)bb[i], true, wg);
            }
        });
    } else
    for_each_pixel_w<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b, const WaveGroup& wg) {
        if (z1) {
            r = r ? r : 1u;
            g = g ? g : 1u;
            b = b ? b : 1u;
        }
        hist_add(wh, (int)r, true, wg);
        hist_add(wh + 256, (int)g, true, wg);
        hist_add(wh + 512, (int)b, true, wg);
    });
    __syncthreads();
    stamp(s, TM_P1);
    {
    double chm[6] = {0, 0, 0, 0, 0, 0};  // per-channel sum(od), sum(od^2) over ALL pixels
    if (tid < 256) {
        unsigned tot = 0;
        const double o = OD(tid);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            unsigned h = 0;
            for (int w = 0; w < NW; ++w) h += (&s.bins[0][0])[w * 768 + c * 256 + tid];
            s.hist3[c][tid] = h;
            tot += h;
            chm[c] = (double)h * o;
            chm[3 + c] = (double)h * o * o;
        }
        s.hist[tid] = tot;
    }
    block_sum(chm, s);
    if (tid < 6) s.chm[tid] = chm[tid];
    }
    __syncthreads();
    }
    if (tid < 64) {  // inclusive prefix over 256 bins: 4 consecutive bins per lane + one wave scan
        const unsigned h0 = s.hist[tid * 4], h1 = s.hist[tid * 4 + 1], h2 = s.hist[tid * 4 + 2], h3 = s.hist[tid * 4 + 3];
        const unsigned incl = wave_incl_scan_u32(h0 + h1 + h2 + h3);
        const unsigned base = incl - (h0 + h1 + h2 + h3);
        s.cum[tid * 4] = base + h0;
        s.cum[tid * 4 + 1] = base + h0 + h1;
        s.cum[tid * 4 + 2] = base + h0 + h1 + h2;
        s.cum[tid * 4 + 3] = incl;
    }
    __syncthreads();
    {
        const unsigned long long nbytes = (unsigned long long)hw * 3ull;
        unsigned long long kp[2], kn[2];
        double gm[2];
        np_index(nbytes, prm.q_img_lo, kp[0], kn[0], gm[0]);
        np_index(nbytes, prm.q_img_hi, kp[1], kn[1], gm[1]);
        if (tid < 256) {
            const unsigned long long c1 = s.cum[tid], c0 = tid ? s.cum[tid - 1] : 0;
            if (c0 <= kp[0] && kp[0] < c1) s.ibc[0] = tid;
            if (c0 <= kn[0] && kn[0] < c1) s.ibc[1] = tid;
            if (c0 <= kp[1] && kp[1] < c1) s.ibc[2] = tid;
            if (c0 <= kn[1] && kn[1] < c1) s.ibc[3] = tid;
            if (c0 == 0 && c1 > 0) s.ibc[4] = tid;                                 // min byte
            if (c1 == (unsigned)nbytes && c0 < (unsigned)nbytes) s.ibc[5] = tid;  // max byte
        }
        __syncthreads();
        if (tid == 0) {
            // uint8 subtraction b-a is non-negative here (sorted), so no wrap-around to mimic
            double plow = np_lerp((double)s.ibc[0], (double)s.ibc[1], gm[0]);
            double phigh = np_lerp((double)s.ibc[2], (double)s.ibc[3], gm[1]);
            if (plow >= phigh) {
                plow = (double)s.ibc[4];
                phigh = (double)s.ibc[5];
            }
            s.bc[0] = plow;
            s.bc[1] = phigh;
            out[TIA_ST_PLOW] = plow;
            out[TIA_ST_PHIGH] = phigh;
        }
        __syncthreads();
    }
    const int bmin = s.ibc[4], bmax = s.ibc[5];
    if (tid < 256) {
        // contrast_enhancer LUT (utils/misc.py:438-444 + skimage rescale_intensity), folded into
        // the Y-row luminance tables: ty[c][v] = C[3+c]*sRGBGamma[ce(v)]
        const double plow = s.bc[0], phigh = s.bc[1];
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
        s.ty[0][tid] = tab->ty[0][ce];
        s.ty[1][tid] = tab->ty[1][ce];
        s.ty[2][tid] = tab->ty[2][ce];
    }
    __syncthreads();

    stamp(s, TM_LUT);
    const int y_thr = prm.y_thr;
    auto is_tissue = [&](uint32_t r, uint32_t g, uint32_t b) -> bool {
        const int t = s.ty[0][r] + s.ty[1][g] + s.ty[2][b];
        return ((t + (1 << 11)) >> 12) < y_thr;
    };
    // P2 records the mask as bits in LDS; later passes test a bit instead of three table look-ups
    const bool use_bits = hw <= (long)MASK_WORDS * 32;
    if (use_bits && !grp && !DL && prm.mode == TIA_MODE_MACENKO) {  // the per-pixel path ORs single bits
        for (int i = tid; i < MASK_WORDS; i += NT) s.mbits[i] = 0;
        __syncthreads();
    }
    auto is_tissue_cached = [&](long idx, uint32_t r, uint32_t g, uint32_t b) -> bool {
        if (use_bits) return (s.mbits[idx >> 5] >> (idx & 31)) & 1u;
        return is_tissue(r, g, b);
    };
    // float32 optical density on the VALU (no table): -ln(max(v,1)/255) clamped at 1e-6 like rgb2od; |error| < 5e-7
    // (v_log_f32 is accurate to 1 ulp).  Only ever used to CLASSIFY pixels against selection windows, with that error
    // bound (and a wide margin) built into the comparison; every value that enters a result is float64 from the table.
    auto od32 = [](uint32_t v) -> float {
        const float f = (float)(v ? v : 1u) * (1.0f / 255.0f);
        const float o = -0.69314718f * __log2f(f);
        return o > 1e-6f ? o : 1e-6f;
    };
    // append one entry per lane that needs it to this wave's private list segment: position = wave count (uniform, in a
    // register) + number of needing lanes below this one (v_mbcnt); no atomics, no cross-lane traffic
    auto seg_push = [&](bool need, unsigned entry, unsigned* seg, unsigned cap, unsigned& count) {
        const unsigned long long m = __ballot(need);
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const unsigned pos = count + before;
        if (need && pos < cap) seg[pos] = entry;
        count += (unsigned)__popcll(m);
    };

    double S[6];  // source stain matrix rows H,E
    unsigned flags = 0;

    if (!DL && prm.mode == TIA_MODE_MACENKO) {
        // ---- P2: tissue mask + OD moments -----------------------------------------------------
        double acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0.0;
        if (grp) {
            // Canonical accumulation order, shared with stain_stats_reg_kernel so that both give the same bits: VIRTUAL thread v
            // of 1024 owns the groups v + 1024 j in ascending order; this thread is virtual thread tid (even visits) and
            // tid + 512 (odd visits), with one accumulator set each; the 16 virtual waves are then summed in order.
            double accB[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) accB[i] = 0.0;
            auto visit = [&](long g, uint32_t a, uint32_t b, uint32_t c, double (&ac)[10]) {
                uint32_t rr[4], gg[4], bb[4];
                unpack_group(a, b, c, rr, gg, bb);
                double x[4], y[4], z[4];
                int lum[4];
                unsigned nib = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // every look-up of the group is in flight before the first use
                    x[i] = OD(rr[i]);
                    y[i] = OD(gg[i]);
                    z[i] = OD(bb[i]);
                    lum[i] = s.ty[0][rr[i]] + s.ty[1][gg[i]] + s.ty[2][bb[i]];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (((lum[i] + (1 << 11)) >> 12) < y_thr) {
                        nib |= 1u << i;
                        ac[0] += 1.0;
                        ac[1] += x[i];
                        ac[2] += y[i];
                        ac[3] += z[i];
                        ac[4] = __builtin_fma(x[i], x[i], ac[4]);
                        ac[5] = __builtin_fma(x[i], y[i], ac[5]);
                        ac[6] = __builtin_fma(x[i], z[i], ac[6]);
                        ac[7] = __builtin_fma(y[i], y[i], ac[7]);
                        ac[8] = __builtin_fma(y[i], z[i], ac[8]);
                        ac[9] = __builtin_fma(z[i], z[i], ac[9]);
                    }
                }
                if (use_bits) {  // 8 consecutive lanes hold 32 consecutive pixels: one mask word
                    // OR over the octet with DPP row shifts (lane i receives lane i+n; VALU only, no LDS crossbar traffic)
                    int word = (int)(nib << (4 * (lane_id() & 7)));
                    word |= __builtin_amdgcn_update_dpp(0, word, 0x101, 0xf, 0xf, true);  // row_shl:1
                    word |= __builtin_amdgcn_update_dpp(0, word, 0x102, 0xf, 0xf, true);  // row_shl:2
                    word |= __builtin_amdgcn_update_dpp(0, word, 0x104, 0xf, 0xf, true);  // row_shl:4
                    if ((lane_id() & 7) == 0) s.mbits[g >> 3] = (unsigned)word;
                }
            };
            const long ngr = hw >> 2;
            const uint32_t* __restrict__ qq = reinterpret_cast<const uint32_t*>(p);
            for (long g = tid; g < ngr; g += 2 * NT) {
                const long g2 = g + NT;
                const bool has2 = g2 < ngr;
                const long g2c = has2 ? g2 : g;
                const uint32_t a0 = qq[g * 3], b0 = qq[g * 3 + 1], c0 = qq[g * 3 + 2];
                const uint32_t a1 = qq[g2c * 3], b1 = qq[g2c * 3 + 1], c1 = qq[g2c * 3 + 2];
                visit(g, a0, b0, c0, acc);
                if (has2) visit(g2, a1, b1, c1, accB);
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const double w1 = wave_sum(acc[i]), w2 = wave_sum(accB[i]);
                if (lane_id() == 0) {
                    s.red16[wave_id()][i] = w1;
                    s.red16[NW + wave_id()][i] = w2;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                double t = 0.0;
                for (int w = 0; w < 16; ++w) t += s.red16[w][i];
                acc[i] = t;
            }
            __syncthreads();
        } else {
        for_each_pixel<NT>(p, hw, [&](long idx, uint32_t r, uint32_t g, uint32_t b) {
            const double x = OD(r), y = OD(g), z = OD(b);
            if (is_tissue(r, g, b)) {
                if (use_bits) atomicOr(&s.mbits[idx >> 5], 1u << (idx & 31));
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
        block_sum(acc, s);
        }
        stamp(s, TM_P2);
        const double nt = acc[0];
        const unsigned long long n_tissue = (unsigned long long)nt;
        if (n_tissue == 0) {
            if (tid == 0) {
                out[TIA_ST_NTISSUE] = 0.0;
                out[TIA_ST_FLAGS] = (double)TIA_FLAG_EMPTY_MASK;
            }
            return;  // uniform across the block
        }
        if (n_tissue < 2) flags |= TIA_FLAG_DEGENERATE;
        if (tid == 0) {
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
                s.bc[2 + i] = e1[i];
                s.bc[5 + i] = e2[i];
                out[TIA_ST_EVEC + i] = e1[i];
                out[TIA_ST_EVEC + 3 + i] = e2[i];
            }
            out[TIA_ST_NTISSUE] = nt;
        }
        __syncthreads();
        stamp(s, TM_EIG);
        const double e1x = s.bc[2], e1y = s.bc[3], e1z = s.bc[4];
        const double e2x = s.bc[5], e2y = s.bc[6], e2z = s.bc[7];

        // ---- P3/P4: exact percentiles of phi = atan2(od.e2, od.e1) over tissue pixels, selected on
        //      the monotone pseudo-angle key ------------------------------------------------------
        unsigned long long kp[2], kn[2], nn[2] = {n_tissue, n_tissue};
        double gm[2];
        np_index(n_tissue, prm.q_phi_lo, kp[0], kn[0], gm[0]);
        np_index(n_tissue, prm.q_phi_hi, kp[1], kn[1], gm[1]);
        const double lo0[2] = {-2.0009765625, -2.0009765625}, hi0[2] = {2.0009765625, 2.0009765625};
        double vp[2], vn[2];
        bool phi_done = false;
        if (prm.select_mode != 1) {
            // float32 classification: with L_c = log2(max(v_c, 1)) the projections are x = Kx - sum_c ex_c L_c (ex = ln2 e1,
            // Kx = log2(255) sum_c ex_c), likewise y; for window edges kb in [-1, 1] and x > 0, key < kb <=> y - kb (|x|+|y|) < 0.
            // Error budget of s = y - kb d: |dL| <= 1 ulp(8) = 9.6e-7, constants rounded to float32 (6e-8 x 8), three FMA
            // roundings (6e-8 x 10 each), the 1e-6 clamp of od(255): |dx|, |dy| <= 7e-6, |ds| <= 3 x 7e-6; four-fold margin.
            const float ln2 = 0.6931471805599453f, l255 = 7.994353436858858f;
            const float ex0 = ln2 * (float)e1x, ex1 = ln2 * (float)e1y, ex2 = ln2 * (float)e1z;
            const float ey0 = ln2 * (float)e2x, ey1 = ln2 * (float)e2y, ey2 = ln2 * (float)e2z;
            const float kx = l255 * (ex0 + ex1 + ex2), ky = l255 * (ey0 + ey1 + ey2);
            const float tol = 8.0e-5f;
            auto proj = [&](uint32_t r, uint32_t g, uint32_t b, float& x, float& y) {
                const float lr = __log2f(fmaxf((float)r, 1.0f)), lg = __log2f(fmaxf((float)g, 1.0f)),
                            lb = __log2f(fmaxf((float)b, 1.0f));
                x = fmaf(-ex2, lb, fmaf(-ex1, lg, fmaf(-ex0, lr, kx)));
                y = fmaf(-ey2, lb, fmaf(-ey1, lg, fmaf(-ey0, lr, ky)));
            };
            phi_done = window_select2(
                p, hw,
                [&](long idx, uint32_t r, uint32_t g, uint32_t b, float (&v)[2]) -> unsigned {
                    if (!is_tissue_cached(idx, r, g, b)) return 0u;
                    float x, y;
                    proj(r, g, b, x, y);
                    const float d = fabsf(x) + fabsf(y);
                    const float q = d > 0.0f ? y / d : 0.0f;
                    v[0] = v[1] = x >= 0.0f ? q : (y >= 0.0f ? 2.0f - q : -2.0f - q);
                    return 3u;
                },
                [&](long idx, uint32_t r, uint32_t g, uint32_t b, double (&x)[2]) -> unsigned {
                    if (!is_tissue_cached(idx, r, g, b)) return 0u;
                    const double ox = OD(r), oy = OD(g), oz = OD(b);
                    const double p0 = dot3(ox, oy, oz, e1x, e1y, e1z);
                    const double p1 = dot3(ox, oy, oz, e2x, e2y, e2z);
                    x[0] = x[1] = pseudo_angle(p1, p0);
                    return 3u;
                },
                [&](unsigned* list, unsigned seg_cap) {
                    // window edges outside [-1, 1] (keys of the x < 0 half plane) are not handled by the cross-product
                    // test: every tissue pixel then becomes a candidate, the overflow check falls back to select2
                    const double w[4] = {s.wlo[0], s.whi[0], s.wlo[1], s.whi[1]};
                    bool edges_ok = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i) edges_ok = edges_ok && (!(fabs(w[i]) < 1e300) || fabs(w[i]) <= 1.0);
                    const float lo0 = (float)w[0], hi0 = (float)w[1], lo1 = (float)w[2], hi1 = (float)w[3];
                    unsigned bl0 = 0, bl1 = 0, count = 0;  // wave-uniform (scalar population counts of the masks)
                    unsigned* seg = list + wave_id() * seg_cap;
                    for_each_group<NT>(p, hw, [&](long g, uint32_t a, uint32_t b, uint32_t c, const WaveGroup&) {
                        uint32_t rr[4], gg[4], bb[4];
                        unpack_group(a, b, c, rr, gg, bb);
                        unsigned nib;
                        if (use_bits) {
                            nib = (s.mbits[g >> 3] >> (4 * (int)(g & 7))) & 15u;
                        } else {
                            nib = 0;
#pragma unroll
                            for (int i = 0; i < 4; ++i) nib |= is_tissue(rr[i], gg[i], bb[i]) ? 1u << i : 0u;
                        }
                        unsigned flags = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool tissue = (nib >> i) & 1u;
                            float x, y;
                            proj(rr[i], gg[i], bb[i], x, y);
                            const float d = fabsf(x) + fabsf(y);
                            const bool plain = edges_ok && x > tol;  // otherwise: exact classification
                            const bool below0 = plain && fmaf(-lo0, d, y) < -tol, above0 = plain && fmaf(-hi0, d, y) > tol;
                            const bool below1 = plain && fmaf(-lo1, d, y) < -tol, above1 = plain && fmaf(-hi1, d, y) > tol;
                            bl0 += (unsigned)__popcll(__ballot(tissue && below0));
                            bl1 += (unsigned)__popcll(__ballot(tissue && below1));
                            const unsigned need = (tissue && !below0 && !above0 ? 1u : 0u) | (tissue && !below1 && !above1 ? 2u : 0u);
                            flags |= need << (2 * i);
                        }
                        seg_push(flags != 0u, (unsigned)g | (flags << 22), seg, seg_cap, count);
                    });
                    if (lane_id() == 0) {
                        s.wcnt[wave_id()] = count;
                        if (bl0) atomicAdd(&s.wbelow[0], (unsigned long long)bl0);
                        if (bl1) atomicAdd(&s.wbelow[1], (unsigned long long)bl1);
                    }
                },
                s, kp, nn, vp, vn);
        }
        if (!phi_done)
        select2(p, hw,
                [&](long idx, uint32_t r, uint32_t g, uint32_t b, double (&x)[2]) -> unsigned {
                    if (!is_tissue_cached(idx, r, g, b)) return 0u;
                    const double ox = OD(r), oy = OD(g), oz = OD(b);
                    const double p0 = dot3(ox, oy, oz, e1x, e1y, e1z);
                    const double p1 = dot3(ox, oy, oz, e2x, e2y, e2z);
                    x[0] = x[1] = pseudo_angle(p1, p0);
                    return 3u;
                },
                [&](unsigned (&below)[2], unsigned (&above)[2]) -> bool {
                    if (!grp) return false;
                    const double lo = s.st.lo[0][0], sc = s.st.scale[0][0];
                    unsigned bl = 0, ab = 0;
                    for_each_group<NT>(p, hw, [&](long g, uint32_t a, uint32_t b, uint32_t c, const WaveGroup& wg) {
                        uint32_t rr[4], gg[4], bb[4];
                        unpack_group(a, b, c, rr, gg, bb);
                        double ox[4], oy[4], oz[4];
                        int lum[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ox[i] = OD(rr[i]);
                            oy[i] = OD(gg[i]);
                            oz[i] = OD(bb[i]);
                            lum[i] = s.ty[0][rr[i]] + s.ty[1][gg[i]] + s.ty[2][bb[i]];
                        }
                        unsigned long long codes = 0ull;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool tissue = ((lum[i] + (1 << 11)) >> 12) < y_thr;
                            const double p0 = dot3(ox[i], oy[i], oz[i], e1x, e1y, e1z);
                            const double p1 = dot3(ox[i], oy[i], oz[i], e2x, e2y, e2z);
                            const double d = (pseudo_angle(p1, p0) - lo) * sc;
                            const bool low = !(d >= 0.0), high = d >= (double)NB;
                            const int bin = low ? 0 : (high ? NB - 1 : (int)d);
                            bl += (tissue && low) ? 1u : 0u;
                            ab += (tissue && high) ? 1u : 0u;
                            hist_add(s.bins[0], bin, tissue && !low && !high, wg);
                            codes |= (unsigned long long)(tissue ? (unsigned)bin : 0xffffu) << (16 * i);
                        }
                        *reinterpret_cast<unsigned long long*>(bincache + g * 4) = codes;
                    });
                    below[0] = bl;
                    above[0] = ab;
                    return true;
                },
                s, kp, nn, lo0, hi0, lo0, hi0, true, bincache, vp, vn);
#if TIA_STATS_TIMING
        if (tid == 0) {
            s.tm[TM_PHI_TOTAL] = clock64() - t_begin;
            s.tm[15] = s.st.level[0] * 1000000 + s.st.level[1] * 100000 + (long long)s.st.cnt[0] + (long long)s.st.cnt[1] * 0;
        }
#endif
        if (tid == 0) {
            const double min_phi = np_lerp(angle_of_key(vp[0]), angle_of_key(vn[0]), gm[0]);
            const double max_phi = np_lerp(angle_of_key(vp[1]), angle_of_key(vn[1]), gm[1]);
            out[TIA_ST_MINPHI] = min_phi;
            out[TIA_ST_MAXPHI] = max_phi;
            const double c1 = cos(min_phi), s1 = sin(min_phi), c2 = cos(max_phi), s2 = sin(max_phi);
            double v1[3] = {e1x * c1 + e2x * s1, e1y * c1 + e2y * s1, e1z * c1 + e2z * s1};
            double v2[3] = {e1x * c2 + e2x * s2, e1y * c2 + e2y * s2, e1z * c2 + e2z * s2};
            const bool first = v1[0] > v2[0];
            const double* h = first ? v1 : v2;
            const double* e = first ? v2 : v1;
            const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
            const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            for (int i = 0; i < 3; ++i) {
                s.bc[8 + i] = h[i] / nh;
                s.bc[11 + i] = e[i] / ne;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 6; ++i) S[i] = s.bc[8 + i];
    } else if (DL) {
        // ---- Vahadane: X = OD[tissue].T (3 samples x N pixel features); code, dictionary = dict_learning(X, 2 atoms,
        //      alpha, max_iter, tol, method="lasso_lars", positive_dict=True); the stain matrix is the CODE (3 x 2)
        //      transposed (stainextract.py:316).  The dictionary (2 x N, f64) lives in the per-patch scratch `dict`,
        //      everything else is a handful of whole-patch reductions between sweeps.
        double2* __restrict__ dict = dictws + (size_t)blockIdx.x * (size_t)hw;
        const double alpha = prm.dl_alpha;
        // S0: uncentred second moments of the tissue OD (X X^T), tissue sums, all-pixel cross moments
        double acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0.0;
        for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
            if (is_tissue(r, g, b)) {
                const double x = OD(r), y = OD(g), z = OD(b);
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
        block_sum(acc, s);
        if (tid < 10) s.bc[24 + tid] = acc[tid];  // tissue count, sums and second moments (unused-atom re-draw)
        stamp(s, TM_P2);
        const double nt = acc[0];
        if (nt == 0.0) {
            if (tid == 0) {
                out[TIA_ST_NTISSUE] = 0.0;
                out[TIA_ST_FLAGS] = (double)TIA_FLAG_EMPTY_MASK;
            }
            return;  // uniform across the block
        }
        // SVD of X through the eigen-decomposition of X X^T: U = eigenvectors (descending), s_k = sqrt(w_k),
        // s_k * Vt_k = u_k^T X; svd_flip makes the largest-magnitude entry of every u_k positive (_dict_learning :592-596)
        if (tid == 0) {
            const double g6[6] = {acc[4], acc[5], acc[6], acc[7], acc[8], acc[9]};
            double w[3], v[3][3];
            jacobi3(g6, w, v);
            int i0 = 0, i1 = 1, i2 = 2;
            if (w[i0] < w[i1]) { int t = i0; i0 = i1; i1 = t; }
            if (w[i0] < w[i2]) { int t = i0; i0 = i2; i2 = t; }
            if (w[i1] < w[i2]) { int t = i1; i1 = i2; i2 = t; }
            const int order[2] = {i0, i1};
            for (int k = 0; k < 2; ++k) {
                double u[3] = {v[0][order[k]], v[1][order[k]], v[2][order[k]]};
                int m = 0;
                if (fabs(u[1]) > fabs(u[m])) m = 1;
                if (fabs(u[2]) > fabs(u[m])) m = 2;
                const double sg = u[m] < 0.0 ? -1.0 : 1.0;
                for (int c = 0; c < 3; ++c) s.bc[16 + c * 2 + k] = u[c] * sg;  // code[c][k]
            }
            out[TIA_ST_NTISSUE] = nt;
        }
        __syncthreads();
        stamp(s, TM_EIG);
        double code[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            code[c][0] = s.bc[16 + c * 2];
            code[c][1] = s.bc[16 + c * 2 + 1];
        }
        // S1: dictionary_k = u_k^T X, with the Gram matrix and covariance the first sparse coding needs
        double gc[9];  // g00 g01 g11 | cov[k][c] = d_k . x_c
        auto gram_cov_reset = [&]() {
#pragma unroll
            for (int i = 0; i < 9; ++i) gc[i] = 0.0;
        };
        auto gram_cov_add = [&](double d0, double d1, double x, double y, double z) {
            gc[0] = __builtin_fma(d0, d0, gc[0]);
            gc[1] = __builtin_fma(d0, d1, gc[1]);
            gc[2] = __builtin_fma(d1, d1, gc[2]);
            gc[3] = __builtin_fma(d0, x, gc[3]);
            gc[4] = __builtin_fma(d0, y, gc[4]);
            gc[5] = __builtin_fma(d0, z, gc[5]);
            gc[6] = __builtin_fma(d1, x, gc[6]);
            gc[7] = __builtin_fma(d1, y, gc[7]);
            gc[8] = __builtin_fma(d1, z, gc[8]);
        };
        gram_cov_reset();
        for_each_pixel_dict<NT, false, true>(p, hw, dict, [&](long, uint32_t r, uint32_t g, uint32_t b, double2& d) {
            if (!is_tissue(r, g, b)) return;
            const double x = OD(r), y = OD(g), z = OD(b);
            const double d0 = dot3(x, y, z, code[0][0], code[1][0], code[2][0]);
            const double d1 = dot3(x, y, z, code[0][1], code[1][1], code[2][1]);
            d = make_double2(d0, d1);
            gram_cov_add(d0, d1, x, y, z);
        });
        block_sum(gc, s);
        double cost_prev = 0.0;
        int n_iter = 0;
        for (int it = 0; it < prm.dl_max_iter; ++it) {
            n_iter = it + 1;
            // sparse coding of the three samples (R, G, B rows of X) against the two atoms
            // (one lane per sample; the codes travel through LDS so the solver is not inlined three times per lane)
            __syncthreads();
            if (tid < 3) {
                double wv[2];
                lasso2(gc[0], gc[1], gc[2], gc[3 + tid], gc[6 + tid], alpha, wv);
                s.bc[16 + tid * 2] = wv[0];
                s.bc[16 + tid * 2 + 1] = wv[1];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                code[c][0] = s.bc[16 + c * 2];
                code[c][1] = s.bc[16 + c * 2 + 1];
            }
            // _update_dict (:519-545): A = code^T code, B = X^T code; atoms updated one after the other
            double A[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                A[0][0] += code[c][0] * code[c][0];
                A[0][1] += code[c][0] * code[c][1];
                A[1][1] += code[c][1] * code[c][1];
            }
            A[1][0] = A[0][1];
            const bool last = it + 1 == prm.dl_max_iter;
            if (last) {  // the returned code only sees _update_dict through the zeroing of unused atoms
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (!(A[k][k] > 1e-6))
                        for (int c = 0; c < 3; ++c) code[c][k] = 0.0;
                break;
            }
            double nrm0 = 1.0, nrm1 = 1.0;
            auto update_atom = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const bool used = A[k][k] > 1e-6;
                int pick = 0;
                double level = 0.0;
                if (!used) {  // atom (almost) never used: re-draw it from the data plus a little noise
                    const unsigned long long key = mix64((unsigned long long)prm.dl_seed * 0x100000001b3ull ^
                                                         ((unsigned long long)blockIdx.x << 20) ^ (unsigned long long)(it * 2 + k));
                    pick = (int)(key % 3ull);
                    const double m1 = s.bc[24 + 1 + pick] / nt, m2 = s.bc[24 + (pick == 0 ? 4 : (pick == 1 ? 7 : 9))] / nt;
                    double var = m2 - m1 * m1;
                    var = var > 0.0 ? var : 0.0;
                    const double sd = sqrt(var);
                    level = 0.01 * (sd != 0.0 ? sd : 1.0);
                    for (int c = 0; c < 3; ++c) code[c][k] = 0.0;
                }
                const double akk = A[k][k], ak0 = A[k][0], ak1 = A[k][1];
                const double ck0 = code[0][k], ck1 = code[1][k], ck2 = code[2][k];
                const double n0 = nrm0;  // atom 0 is divided by its norm lazily, while atom 1 is updated
                double nn2[1] = {0.0};
                const unsigned long long nkey = mix64((unsigned long long)prm.dl_seed ^ ((unsigned long long)blockIdx.x << 32) ^
                                                      (unsigned long long)(it * 2 + k + 1));
                if (used) {
                    for_each_pixel_dict<NT, true, true>(p, hw, dict, [&](long, uint32_t r, uint32_t g, uint32_t b, double2& d) {
                        if (!is_tissue(r, g, b)) return;
                        const double x = OD(r), y = OD(g), z = OD(b);
                        if (k == 1) d.x = d.x / n0;  // dictionary[0] /= max(norm, 1)
                        const double bk = x * ck0 + y * ck1 + z * ck2;           // B[:, k]
                        const double ad = ak0 * d.x + ak1 * d.y;                 // A[k] @ dictionary
                        double v = (k == 0 ? d.x : d.y) + (bk - ad) / akk;
                        v = v < 0.0 ? 0.0 : v;  // positive_dict
                        if (k == 0) d.x = v; else d.y = v;
                        nn2[0] = __builtin_fma(v, v, nn2[0]);
                    });
                } else {  // rare: plain loop, keeps the transcendental code out of the unrolled sweep
                    for (long idx = tid; idx < hw; idx += NT) {
                        const uint32_t r = p[3 * idx], g = p[3 * idx + 1], b = p[3 * idx + 2];
                        if (!is_tissue(r, g, b)) continue;
                        double2 d = dict[idx];
                        if (k == 1) d.x = d.x / n0;
                        const double base = pick == 0 ? OD(r) : (pick == 1 ? OD(g) : OD(b));
                        double v = base + level * normal_of(nkey + (unsigned long long)idx * 0x9e3779b97f4a7c15ull);
                        v = v < 0.0 ? 0.0 : v;
                        if (k == 0) d.x = v; else d.y = v;
                        dict[idx] = d;
                        nn2[0] = __builtin_fma(v, v, nn2[0]);
                    }
                }
                block_sum(nn2, s);
                const double nv = sqrt(nn2[0]);
                (k == 0 ? nrm0 : nrm1) = nv > 1.0 ? nv : 1.0;
            };
            update_atom(std::integral_constant<int, 0>{});
            update_atom(std::integral_constant<int, 1>{});
            // atom 1's normalisation is applied in the sweep that evaluates the cost and prepares the next coding
            const double n1 = nrm1;
            double cst[1] = {0.0};
            gram_cov_reset();
            for_each_pixel_dict<NT, true, true>(p, hw, dict, [&](long, uint32_t r, uint32_t g, uint32_t b, double2& d) {
                if (!is_tissue(r, g, b)) return;
                const double x = OD(r), y = OD(g), z = OD(b);
                d.y = d.y / n1;
                const double ex = x - (code[0][0] * d.x + code[0][1] * d.y);
                const double ey = y - (code[1][0] * d.x + code[1][1] * d.y);
                const double ez = z - (code[2][0] * d.x + code[2][1] * d.y);
                cst[0] += ex * ex + ey * ey + ez * ez;
                gram_cov_add(d.x, d.y, x, y, z);
            });
            block_sum(cst, s);
            block_sum(gc, s);
            double l1 = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) l1 += fabs(code[c][0]) + fabs(code[c][1]);
            const double cost = 0.5 * cst[0] + alpha * l1;
            if (it > 0 && (cost_prev - cost) < prm.dl_tol * cost) break;  // :657-665
            cost_prev = cost;
        }
        // dictionary = code.T; H first (dl_output_for_h_and_e, :53-68); unit rows (:322)
        {
            double h[3] = {code[0][0], code[1][0], code[2][0]}, e[3] = {code[0][1], code[1][1], code[2][1]};
            const bool swap = h[0] < e[0];
            const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
            const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double hv = h[i] / nh, ev = e[i] / ne;
                S[i] = swap ? ev : hv;
                S[3 + i] = swap ? hv : ev;
            }
        }
        if (tid == 0) {
            out[TIA_ST_MINPHI] = (double)n_iter;  // Vahadane: number of dictionary-learning iterations run
#if TIA_STATS_TIMING
            s.tm[TM_PHI_TOTAL] = clock64() - t_begin;
#endif
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) S[i] = (prm.mode == TIA_MODE_GIVEN || prm.mode == MODE_VTAIL) ? s_given[i] : prm.stain_fixed[i];
        if (prm.mode == MODE_VTAIL) {
            if (keep_flags & TIA_FLAG_EMPTY_MASK) {  // the record the one-kernel form leaves for an empty tissue mask
                if (tid == 0) {
                    out[TIA_ST_NTISSUE] = 0.0;
                    out[TIA_ST_FLAGS] = (double)TIA_FLAG_EMPTY_MASK;
                }
                return;  // uniform across the block
            }
            if (tid == 0) {
                out[TIA_ST_NTISSUE] = keep_nt;
                out[TIA_ST_MINPHI] = keep_iter;
            }
        }
        stamp(s, TM_P2);
    }

    // ---- pseudo-inverse: C = OD . P,  P = S^T (S S^T)^-1  (lstsq of stainnorm.py:65) ----------
    double P[6];
    {
        const double a = S[0] * S[0] + S[1] * S[1] + S[2] * S[2];
        const double bb = S[0] * S[3] + S[1] * S[4] + S[2] * S[5];
        const double d = S[3] * S[3] + S[4] * S[4] + S[5] * S[5];
        const double det = a * d - bb * bb;
        const double g00 = d / det, g01 = -bb / det, g11 = a / det;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            P[j * 2 + 0] = S[j] * g00 + S[3 + j] * g01;
            P[j * 2 + 1] = S[j] * g01 + S[3 + j] * g11;
        }
    }

    // ---- P5/P6: exact percentile of both concentration channels over ALL pixels ----------------
    double maxc[2];
    {
        const unsigned long long npx = (unsigned long long)hw;
        unsigned long long kp[2], kn[2], nn[2] = {npx, npx};
        double gm[2];
        np_index(npx, prm.q_conc, kp[0], kn[0], gm[0]);
        kp[1] = kp[0];
        kn[1] = kn[0];
        gm[1] = gm[0];
        // value bounds and the first histogram window of the histogram path -- needed only when that path runs
        double lo0[2], hi0[2], olo0[2], ohi0[2];
        auto histogram_ranges = [&]() {
            if constexpr (!DL) {
                // per-channel moments of od over ALL pixels (the dictionary-learning instantiation has them from its per-channel
                // byte histograms; here P1 keeps one histogram of all bytes): one extra sweep, on the fall-back / audit path only.
                // They only place the first histogram window: the selection is exact for any window.
                double cm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
                    const double x = OD(r), y = OD(g), z = OD(b);
                    cm[0] += x;
                    cm[1] += y;
                    cm[2] += z;
                    cm[3] = __builtin_fma(x, x, cm[3]);
                    cm[4] = __builtin_fma(y, y, cm[4]);
                    cm[5] = __builtin_fma(z, z, cm[5]);
                });
                block_sum(cm, s);
                if (tid < 6) s.chm[tid] = cm[tid];
                __syncthreads();
            }
            // rigorous value bounds from the byte range: od in [od(bmax), od(bmin)]
            const double oa = OD(bmax), ob = OD(bmin);
            const double inv_n = 1.0 / (double)hw;
    #pragma unroll
            for (int t = 0; t < 2; ++t) {
                double lo = 0.0, hi = 0.0, mu = 0.0;
                double mj[3];
    #pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double c = P[j * 2 + t];
                    const double u = c * oa, w = c * ob;
                    lo += u < w ? u : w;
                    hi += u < w ? w : u;
                    mj[j] = s.chm[j] * inv_n;
                    mu += c * mj[j];
                }
                // sigma(C_t) <= sum_c |P[c][t]| sigma(od_c) (per-channel moments come from the byte histograms of P1)
                const double cxx = s.chm[3] * inv_n - mj[0] * mj[0], cyy = s.chm[4] * inv_n - mj[1] * mj[1];
                const double czz = s.chm[5] * inv_n - mj[2] * mj[2];
                const double sdev = fabs(P[0 + t]) * sqrt(cxx > 0.0 ? cxx : 0.0) + fabs(P[2 + t]) * sqrt(cyy > 0.0 ? cyy : 0.0) +
                                    fabs(P[4 + t]) * sqrt(czz > 0.0 ? czz : 0.0);
                const double var = sdev * sdev;
                const double sg = sqrt(var) * 1.000001 + 1e-12 * (fabs(mu) + 1.0);
                const double pad = 1e-9 * (fabs(lo) + fabs(hi)) + 1e-12;
                olo0[t] = lo - pad;
                ohi0[t] = hi + pad;
                // histogram window: Chebyshev keeps the 99th percentile inside mu + 12 sigma
                double wlo = mu - 8.0 * sg, whi = mu + 12.0 * sg;
                wlo = wlo > olo0[t] ? wlo : olo0[t];
                whi = whi < ohi0[t] ? whi : ohi0[t];
                if (!(whi > wlo)) {
                    wlo = olo0[t];
                    whi = ohi0[t];
                }
                lo0[t] = wlo;
                hi0[t] = whi;
            }
        };
        double vp[2], vn[2];
        bool conc_done = false;
        if (prm.select_mode != 1) {
            // C_t = sum_c P[c][t] od_c = K_t - sum_c pt_c L_c with pt = ln2 P (see the angular sweep for the error budget):
            // |dC_t| <= (9.6e-7 + 5e-7 + 1e-6 / ln2) |pt|_1 + 4 roundings of |C| <= ~4e-6 |P column|_1; eight-fold margin.
            const float ln2 = 0.6931471805599453f, l255 = 7.994353436858858f;
            const float a0 = ln2 * (float)P[0], a1 = ln2 * (float)P[2], a2 = ln2 * (float)P[4];
            const float b0 = ln2 * (float)P[1], b1 = ln2 * (float)P[3], b2 = ln2 * (float)P[5];
            const float ka = l255 * (a0 + a1 + a2), kb = l255 * (b0 + b1 + b2);
            const float tol0 = 3.2e-5f * (fabsf((float)P[0]) + fabsf((float)P[2]) + fabsf((float)P[4])) + 1e-7f;
            const float tol1 = 3.2e-5f * (fabsf((float)P[1]) + fabsf((float)P[3]) + fabsf((float)P[5])) + 1e-7f;
            auto conc32 = [&](uint32_t r, uint32_t g, uint32_t b, float& c0, float& c1) {
                const float lr = __log2f(fmaxf((float)r, 1.0f)), lg = __log2f(fmaxf((float)g, 1.0f)),
                            lb = __log2f(fmaxf((float)b, 1.0f));
                c0 = fmaf(-a2, lb, fmaf(-a1, lg, fmaf(-a0, lr, ka)));
                c1 = fmaf(-b2, lb, fmaf(-b1, lg, fmaf(-b0, lr, kb)));
            };
            conc_done = window_select2(
                p, hw,
                [&](long, uint32_t r, uint32_t g, uint32_t b, float (&v)[2]) -> unsigned {
                    conc32(r, g, b, v[0], v[1]);
                    return 3u;
                },
                [&](long, uint32_t r, uint32_t g, uint32_t b, double (&x)[2]) -> unsigned {
                    const double ox = OD(r), oy = OD(g), oz = OD(b);
                    x[0] = dot3(ox, oy, oz, P[0], P[2], P[4]);
                    x[1] = dot3(ox, oy, oz, P[1], P[3], P[5]);
                    return 3u;
                },
                [&](unsigned* list, unsigned seg_cap) {
                    const float lo0 = (float)s.wlo[0], hi0 = (float)s.whi[0], lo1 = (float)s.wlo[1], hi1 = (float)s.whi[1];
                    // the float32 images of the window edges are themselves rounded: 1.2e-7 relative
                    auto slack = [](float v) { return fabsf(v) < 3e38f ? 2.4e-7f * fabsf(v) : 0.0f; };
                    const float t0 = tol0 + slack(lo0) + slack(hi0), t1 = tol1 + slack(lo1) + slack(hi1);
                    unsigned bl0 = 0, bl1 = 0, count = 0;
                    unsigned* seg = list + wave_id() * seg_cap;
                    for_each_group<NT>(p, hw, [&](long g, uint32_t a, uint32_t b, uint32_t c, const WaveGroup&) {
                        uint32_t rr[4], gg[4], bb[4];
                        unpack_group(a, b, c, rr, gg, bb);
                        unsigned flags = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float c0, c1;
                            conc32(rr[i], gg[i], bb[i], c0, c1);
                            const bool below0 = c0 + t0 < lo0, above0 = c0 - t0 > hi0;
                            const bool below1 = c1 + t1 < lo1, above1 = c1 - t1 > hi1;
                            bl0 += (unsigned)__popcll(__ballot(below0));
                            bl1 += (unsigned)__popcll(__ballot(below1));
                            flags |= ((!below0 && !above0 ? 1u : 0u) | (!below1 && !above1 ? 2u : 0u)) << (2 * i);
                        }
                        seg_push(flags != 0u, (unsigned)g | (flags << 22), seg, seg_cap, count);
                    });
                    if (lane_id() == 0) {
                        s.wcnt[wave_id()] = count;
                        if (bl0) atomicAdd(&s.wbelow[0], (unsigned long long)bl0);
                        if (bl1) atomicAdd(&s.wbelow[1], (unsigned long long)bl1);
                    }
                },
                s, kp, nn, vp, vn);
        }
        if (!conc_done) histogram_ranges();
        if (!conc_done)
        select2(p, hw,
                [&](long, uint32_t r, uint32_t g, uint32_t b, double (&x)[2]) -> unsigned {
                    const double ox = OD(r), oy = OD(g), oz = OD(b);
                    x[0] = dot3(ox, oy, oz, P[0], P[2], P[4]);
                    x[1] = dot3(ox, oy, oz, P[1], P[3], P[5]);
                    return 3u;
                },
                [&](unsigned (&below)[2], unsigned (&above)[2]) -> bool {
                    if (!grp) return false;
                    const double l0 = s.st.lo[0][0], s0 = s.st.scale[0][0], l1 = s.st.lo[1][0], s1 = s.st.scale[1][0];
                    unsigned bl0 = 0, ab0 = 0, bl1 = 0, ab1 = 0;
                    for_each_group<NT>(p, hw, [&](long g, uint32_t a, uint32_t b, uint32_t c, const WaveGroup& wg) {
                        uint32_t rr[4], gg[4], bb[4];
                        unpack_group(a, b, c, rr, gg, bb);
                        double ox[4], oy[4], oz[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ox[i] = OD(rr[i]);
                            oy[i] = OD(gg[i]);
                            oz[i] = OD(bb[i]);
                        }
                        unsigned long long code0 = 0ull, code1 = 0ull;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double d0 = (dot3(ox[i], oy[i], oz[i], P[0], P[2], P[4]) - l0) * s0;
                            const double d1 = (dot3(ox[i], oy[i], oz[i], P[1], P[3], P[5]) - l1) * s1;
                            const bool low0 = !(d0 >= 0.0), high0 = d0 >= (double)NB;
                            const bool low1 = !(d1 >= 0.0), high1 = d1 >= (double)NB;
                            const int b0 = low0 ? 0 : (high0 ? NB - 1 : (int)d0);
                            const int b1 = low1 ? 0 : (high1 ? NB - 1 : (int)d1);
                            bl0 += low0 ? 1u : 0u;
                            ab0 += high0 ? 1u : 0u;
                            bl1 += low1 ? 1u : 0u;
                            ab1 += high1 ? 1u : 0u;
                            hist_add(s.bins[0], b0, !low0 && !high0, wg);
                            hist_add(s.bins[1], b1, !low1 && !high1, wg);
                            code0 |= (unsigned long long)(unsigned)b0 << (16 * i);
                            code1 |= (unsigned long long)(unsigned)b1 << (16 * i);
                        }
                        *reinterpret_cast<unsigned long long*>(bincache + g * 4) = code0;
                        *reinterpret_cast<unsigned long long*>(bincache + (size_t)hw + g * 4) = code1;
                    });
                    below[0] = bl0;
                    above[0] = ab0;
                    below[1] = bl1;
                    above[1] = ab1;
                    return true;
                },
                s, kp, nn, lo0, hi0, olo0, ohi0, false, bincache, vp, vn);
#if TIA_STATS_TIMING
        if (tid == 0) {
            s.tm[11] = s.st.level[0];
            s.tm[12] = s.st.level[1];  // (13: failure codes of the window selections, 14: their candidate counts)
        }
#endif
        maxc[0] = np_lerp(vp[0], vn[0], gm[0]);
        maxc[1] = np_lerp(vp[1], vn[1], gm[1]);
    }

    if (tid == 0) {
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
                    out[TIA_ST_M + j * 3 + c] = P[j * 2 + 0] * sc0 * prm.target_stain[c] +
                                                P[j * 2 + 1] * sc1 * prm.target_stain[3 + c];
        }
        out[TIA_ST_FLAGS] = (double)flags;
#if TIA_STATS_TIMING
        s.tm[TM_TOTAL] = clock64() - t_begin;
        s.tm[TM_CONC_TOTAL] = s.tm[TM_TOTAL] - s.tm[TM_PHI_TOTAL];
        for (int i = 0; i < 16; ++i) out[TIA_ST_CYCLES + i] = (double)s.tm[i];
#endif
    }
}


// =====================================================================================================================
// Vahadane, first kernel of the pair: dictionary learning WITHOUT a dictionary in memory
// =====================================================================================================================
// In sklearn's _update_dict the value of atom k at pixel p only depends on x_p (the pixel's three OD values) and on
// per-iteration SCALARS (codes, A = code^T code, atom norms): d_k <- max(0, d_k + (x_p . c_k - A_k . d) / A_kk), then
// / max(norm, 1).  stain_stats_kernel<true> keeps the 2 x N float64 dictionary in a workspace and reads + writes it on every one
// of the 7 sweeps after its initialisation: 32 bytes per pixel and sweep, which is what bounds it (3.8 TB/s of dictionary
// traffic once the sweeps were software-pipelined, profiles/r04i_*).  Here every sweep RECOMPUTES a pixel's atom values from
// x_p by replaying the updates recorded so far (18 scalars per iteration in scalar registers): the same operations in the same
// order on the same values, hence the same bits -- no dictionary traffic at all, only the 3 bytes per pixel of the patch, which
// stay in L2 / MALL across the sweeps.  The replay costs arithmetic (up to two recorded iterations per pixel and sweep), so
// this kernel carries NOTHING but the learning loop -- no selection machinery, 128 registers, two workgroups per CU (the
// one-kernel form runs 254 registers at two waves per SIMD: replaying there was measured 2x SLOWER than the dictionary
// traffic, profiles/r04h_*); the statistics record is completed by stain_stats_kernel<false> in MODE_VTAIL (same tail code as
// every other mode).  Divisions: every divisor of the updates is one of those scalars, so they are done with Markstein's
// sequence q = a y, r = fma(-b, q, a), q' = fma(r, y, q) on the correctly rounded reciprocal y = 1 / b (computed once per sweep):
// q' is the correctly rounded quotient, in 3 full-rate instructions instead of ~15 partly quarter-rate ones.
// Handed back to stain_stats_kernel<true> through the redo flags (same results, the materialised form): a patch whose atom
// becomes unused (A_kk <= 1e-6: its re-draw adds pseudo-random values per pixel, not replayable from scalars) and runs with more
// than DL_HIST + 1 iterations.  tests: bit-identity of the pair with the one-kernel form on every Vahadane case.
constexpr int DL_HIST = 2;  // recorded iterations (the reference runs max_iter = 3: two rounds of updates, stainextract.py:313)

struct SmemV {
    double od[256];
    int ty[3][256];
    unsigned hist[256];
    unsigned cum[256];
    unsigned hstripe[256 * 32];
    double red[NW][16];
    double bc[48];
    double dlh[DL_HIST][18];  // c0[3] a00 a01 | c1[3] a10 a11 | akk0 akk1 n0 n1 | 1/akk0 1/akk1 1/n0 1/n1
    int ibc[8];
};

#ifndef TIA_DL_WPE
#define TIA_DL_WPE 2
#endif
__global__ __launch_bounds__(NT, TIA_DL_WPE) void vahadane_dl_kernel(const uint8_t* __restrict__ img, long hw, const tia_stain_tables* __restrict__ tab,
                                                            tia_stain_params prm, double* __restrict__ stats, int* __restrict__ redo) {
    __shared__ SmemV s;
    const uint8_t* p = img + (size_t)blockIdx.x * (size_t)hw * 3u;
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    const int tid = threadIdx.x;
    const bool z1 = prm.zero_to_one != 0;
    auto hand_back = [&]() {
        if (tid == 0) redo[blockIdx.x] = 1;
    };
    if (prm.dl_max_iter - 1 > DL_HIST) {  // uniform
        hand_back();
        return;
    }
    if (tid < TIA_STATS_STRIDE) out[tid] = 0.0;
    for (int i = tid; i < 256; i += NT) s.od[i] = tab->od_lut[i];
    for (int i = tid; i < 256 * 32; i += NT) s.hstripe[i] = 0u;
    __syncthreads();
    // ---- P1 + contrast-enhancer tables: as in stain_stats_kernel<false> (same histogram, same percentile arithmetic) -----------
    {
        unsigned* hs = s.hstripe + (lane_id() & 31);
        auto add = [&](uint32_t v) {
            if (z1) v = v ? v : 1u;
            atomicAdd(hs + v * 32u, 1u);
        };
        if (groups_ok(p, hw)) {
            for_each_group<NT>(p, hw, [&](long, uint32_t a, uint32_t b, uint32_t c, const WaveGroup&) {
                const uint32_t w[3] = {a, b, c};
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) add((w[d] >> (8 * e)) & 255u);
            });
        } else {
            for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
                add(r);
                add(g);
                add(b);
            });
        }
    }
    __syncthreads();
    if (tid < 256) {
        unsigned tot = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) tot += s.hstripe[tid * 32 + ((c + lane_id()) & 31)];
        s.hist[tid] = tot;
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned h0 = s.hist[tid * 4], h1 = s.hist[tid * 4 + 1], h2 = s.hist[tid * 4 + 2], h3 = s.hist[tid * 4 + 3];
        const unsigned incl = wave_incl_scan_u32(h0 + h1 + h2 + h3);
        const unsigned base = incl - (h0 + h1 + h2 + h3);
        s.cum[tid * 4] = base + h0;
        s.cum[tid * 4 + 1] = base + h0 + h1;
        s.cum[tid * 4 + 2] = base + h0 + h1 + h2;
        s.cum[tid * 4 + 3] = incl;
    }
    __syncthreads();
    {
        const unsigned long long nbytes = (unsigned long long)hw * 3ull;
        unsigned long long kp[2], kn[2];
        double gm[2];
        np_index(nbytes, prm.q_img_lo, kp[0], kn[0], gm[0]);
        np_index(nbytes, prm.q_img_hi, kp[1], kn[1], gm[1]);
        if (tid < 256) {
            const unsigned long long c1 = s.cum[tid], c0 = tid ? s.cum[tid - 1] : 0;
            if (c0 <= kp[0] && kp[0] < c1) s.ibc[0] = tid;
            if (c0 <= kn[0] && kn[0] < c1) s.ibc[1] = tid;
            if (c0 <= kp[1] && kp[1] < c1) s.ibc[2] = tid;
            if (c0 <= kn[1] && kn[1] < c1) s.ibc[3] = tid;
            if (c0 == 0 && c1 > 0) s.ibc[4] = tid;
            if (c1 == (unsigned)nbytes && c0 < (unsigned)nbytes) s.ibc[5] = tid;
        }
        __syncthreads();
        if (tid == 0) {
            double plow = np_lerp((double)s.ibc[0], (double)s.ibc[1], gm[0]);
            double phigh = np_lerp((double)s.ibc[2], (double)s.ibc[3], gm[1]);
            if (plow >= phigh) {
                plow = (double)s.ibc[4];
                phigh = (double)s.ibc[5];
            }
            s.bc[0] = plow;
            s.bc[1] = phigh;
        }
        __syncthreads();
    }
    if (tid < 256) {
        const double plow = s.bc[0], phigh = s.bc[1];
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
        s.ty[0][tid] = tab->ty[0][ce];
        s.ty[1][tid] = tab->ty[1][ce];
        s.ty[2][tid] = tab->ty[2][ce];
    }
    __syncthreads();
    const int y_thr = prm.y_thr;
    auto is_tissue = [&](uint32_t r, uint32_t g, uint32_t b) -> bool {
        const int t = s.ty[0][r] + s.ty[1][g] + s.ty[2][b];
        return ((t + (1 << 11)) >> 12) < y_thr;
    };
#define ODV(v) s.od[(v)]
    const double alpha = prm.dl_alpha;
    // ---- S0: uncentred second moments of the tissue OD (as the one-kernel form) ------------------------------------------------
    double acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0;
    for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
        if (is_tissue(r, g, b)) {
            const double x = ODV(r), y = ODV(g), z = ODV(b);
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
    block_sum(acc, s);
    const double nt = acc[0];
    if (nt == 0.0) {
        if (tid == 0) {
            out[TIA_ST_NTISSUE] = 0.0;
            out[TIA_ST_FLAGS] = (double)TIA_FLAG_EMPTY_MASK;
        }
        return;  // uniform across the block
    }
    if (tid == 0) {
        const double g6[6] = {acc[4], acc[5], acc[6], acc[7], acc[8], acc[9]};
        double w[3], v[3][3];
        jacobi3(g6, w, v);
        int i0 = 0, i1 = 1, i2 = 2;
        if (w[i0] < w[i1]) { int t = i0; i0 = i1; i1 = t; }
        if (w[i0] < w[i2]) { int t = i0; i0 = i2; i2 = t; }
        if (w[i1] < w[i2]) { int t = i1; i1 = i2; i2 = t; }
        const int order[2] = {i0, i1};
        for (int k = 0; k < 2; ++k) {
            double u[3] = {v[0][order[k]], v[1][order[k]], v[2][order[k]]};
            int m = 0;
            if (fabs(u[1]) > fabs(u[m])) m = 1;
            if (fabs(u[2]) > fabs(u[m])) m = 2;
            const double sg = u[m] < 0.0 ? -1.0 : 1.0;
            for (int c = 0; c < 3; ++c) s.bc[16 + c * 2 + k] = u[c] * sg;  // code[c][k]
        }
    }
    __syncthreads();
    double code[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        code[c][0] = s.bc[16 + c * 2];
        code[c][1] = s.bc[16 + c * 2 + 1];
    }
    const double u00 = code[0][0], u10 = code[1][0], u20 = code[2][0], u01 = code[0][1], u11 = code[1][1], u21 = code[2][1];
    // ---- replay ------------------------------------------------------------------------------------------------------------------
    auto div_by = [](double a, double b, double y) -> double {
        const double q = a * y;
        return __builtin_fma(__builtin_fma(-b, q, a), y, q);
    };
    double h[DL_HIST][18];  // the recorded scalars as wave-uniform values (scalar registers), refreshed from LDS before every sweep
    auto load_hist = [&]() {
#pragma unroll
        for (int j = 0; j < DL_HIST; ++j)
#pragma unroll
            for (int c = 0; c < 18; ++c) {
                const long long bits = __double_as_longlong(s.dlh[j][c]);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bits), hi = __builtin_amdgcn_readfirstlane((unsigned)(bits >> 32));
                h[j][c] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
    };
    // one recorded iteration applied to (d.x, d.y), up to and including step `st` (1: atom 0; 2: + its normalisation and atom 1;
    // 3: + atom 1's normalisation) -- operation for operation what the sweeps of the one-kernel form do to `dict[idx]`
    auto replay_step = [&](const double (&hj)[18], int st, double x, double y, double z, double2& d) {
        {
            const double bk = x * hj[0] + y * hj[1] + z * hj[2];
            const double ad = hj[3] * d.x + hj[4] * d.y;
            const double v = d.x + div_by(bk - ad, hj[10], hj[14]);
            d.x = v < 0.0 ? 0.0 : v;
        }
        if (st == 1) return;
        {
            d.x = div_by(d.x, hj[12], hj[16]);
            const double bk = x * hj[5] + y * hj[6] + z * hj[7];
            const double ad = hj[8] * d.x + hj[9] * d.y;
            const double v = d.y + div_by(bk - ad, hj[11], hj[15]);
            d.y = v < 0.0 ? 0.0 : v;
        }
        if (st == 2) return;
        d.y = div_by(d.y, hj[13], hj[17]);
    };
    static_assert(DL_HIST == 2, "replay() spells its two recorded iterations out (constant indices keep them in registers)");
    // the pixel's atom values after `full` completed iterations plus `stage` steps of the next one
    auto replay = [&](double x, double y, double z, int full, int stage) -> double2 {
        double2 d;
        d.x = dot3(x, y, z, u00, u10, u20);
        d.y = dot3(x, y, z, u01, u11, u21);
        const int st0 = full > 0 ? 3 : stage;
        if (st0) replay_step(h[0], st0, x, y, z, d);
        const int st1 = full > 1 ? 3 : (full == 1 ? stage : 0);
        if (st1) replay_step(h[1], st1, x, y, z, d);
        return d;
    };
    // ---- S1: Gram matrix and covariance of the initial dictionary (u_k^T X) -------------------------------------------------------
    double gc[9];
    auto gram_cov_reset = [&]() {
#pragma unroll
        for (int i = 0; i < 9; ++i) gc[i] = 0.0;
    };
    auto gram_cov_add = [&](double d0, double d1, double x, double y, double z) {
        gc[0] = __builtin_fma(d0, d0, gc[0]);
        gc[1] = __builtin_fma(d0, d1, gc[1]);
        gc[2] = __builtin_fma(d1, d1, gc[2]);
        gc[3] = __builtin_fma(d0, x, gc[3]);
        gc[4] = __builtin_fma(d0, y, gc[4]);
        gc[5] = __builtin_fma(d0, z, gc[5]);
        gc[6] = __builtin_fma(d1, x, gc[6]);
        gc[7] = __builtin_fma(d1, y, gc[7]);
        gc[8] = __builtin_fma(d1, z, gc[8]);
    };
    gram_cov_reset();
    for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
        if (!is_tissue(r, g, b)) return;
        const double x = ODV(r), y = ODV(g), z = ODV(b);
        const double d0 = dot3(x, y, z, u00, u10, u20);
        const double d1 = dot3(x, y, z, u01, u11, u21);
        gram_cov_add(d0, d1, x, y, z);
    });
    block_sum(gc, s);
    double cost_prev = 0.0;
    int n_iter = 0;
    for (int it = 0; it < prm.dl_max_iter; ++it) {
        n_iter = it + 1;
        __syncthreads();
        if (tid < 3) {
            double wv[2];
            lasso2(gc[0], gc[1], gc[2], gc[3 + tid], gc[6 + tid], alpha, wv);
            s.bc[16 + tid * 2] = wv[0];
            s.bc[16 + tid * 2 + 1] = wv[1];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            code[c][0] = s.bc[16 + c * 2];
            code[c][1] = s.bc[16 + c * 2 + 1];
        }
        double A[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A[0][0] += code[c][0] * code[c][0];
            A[0][1] += code[c][0] * code[c][1];
            A[1][1] += code[c][1] * code[c][1];
        }
        A[1][0] = A[0][1];
        const bool last = it + 1 == prm.dl_max_iter;
        if (last) {  // the returned code only sees _update_dict through the zeroing of unused atoms
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (!(A[k][k] > 1e-6))
                    for (int c = 0; c < 3; ++c) code[c][k] = 0.0;
            break;
        }
        if (!(A[0][0] > 1e-6) || !(A[1][1] > 1e-6)) {  // an unused atom is re-drawn with per-pixel noise: the materialised form
            hand_back();
            return;  // uniform across the block
        }
        __syncthreads();
        if (tid == 0) {
            double* hj = s.dlh[it];
            hj[0] = code[0][0], hj[1] = code[1][0], hj[2] = code[2][0], hj[3] = A[0][0], hj[4] = A[0][1];
            hj[5] = code[0][1], hj[6] = code[1][1], hj[7] = code[2][1], hj[8] = A[1][0], hj[9] = A[1][1];
            hj[10] = A[0][0], hj[11] = A[1][1], hj[12] = 1.0, hj[13] = 1.0;
            hj[14] = 1.0 / A[0][0], hj[15] = 1.0 / A[1][1], hj[16] = 1.0, hj[17] = 1.0;
        }
        __syncthreads();
        load_hist();
        double nrm0 = 1.0, nrm1 = 1.0;
        auto update_atom = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double akk = A[k][k], ak0 = A[k][0], ak1 = A[k][1];
            const double ck0 = code[0][k], ck1 = code[1][k], ck2 = code[2][k];
            const double n0 = nrm0, inv_n0 = 1.0 / nrm0, inv_akk = 1.0 / akk;
            double nn2[1] = {0.0};
            for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
                if (!is_tissue(r, g, b)) return;
                const double x = ODV(r), y = ODV(g), z = ODV(b);
                double2 d = replay(x, y, z, it, k);
                if (k == 1) d.x = div_by(d.x, n0, inv_n0);      // dictionary[0] /= max(norm, 1)
                const double bk = x * ck0 + y * ck1 + z * ck2;  // B[:, k]
                const double ad = ak0 * d.x + ak1 * d.y;        // A[k] @ dictionary
                double v = (k == 0 ? d.x : d.y) + div_by(bk - ad, akk, inv_akk);
                v = v < 0.0 ? 0.0 : v;  // positive_dict
                nn2[0] = __builtin_fma(v, v, nn2[0]);
            });
            block_sum(nn2, s);
            const double nv = sqrt(nn2[0]);
            (k == 0 ? nrm0 : nrm1) = nv > 1.0 ? nv : 1.0;
            if (tid == 0) {  // the norm joins the iteration's record (block_sum ended with a barrier: nobody is reading s.dlh)
                s.dlh[it][12 + k] = nv > 1.0 ? nv : 1.0;
                s.dlh[it][16 + k] = 1.0 / (nv > 1.0 ? nv : 1.0);
            }
            __syncthreads();
            load_hist();
        };
        update_atom(std::integral_constant<int, 0>{});
        update_atom(std::integral_constant<int, 1>{});
        const double n1 = nrm1, inv_n1 = 1.0 / nrm1;
        double cst[1] = {0.0};
        gram_cov_reset();
        for_each_pixel<NT>(p, hw, [&](long, uint32_t r, uint32_t g, uint32_t b) {
            if (!is_tissue(r, g, b)) return;
            const double x = ODV(r), y = ODV(g), z = ODV(b);
            double2 d = replay(x, y, z, it, 2);
            d.y = div_by(d.y, n1, inv_n1);
            const double ex = x - (code[0][0] * d.x + code[0][1] * d.y);
            const double ey = y - (code[1][0] * d.x + code[1][1] * d.y);
            const double ez = z - (code[2][0] * d.x + code[2][1] * d.y);
            cst[0] += ex * ex + ey * ey + ez * ez;
            gram_cov_add(d.x, d.y, x, y, z);
        });
        block_sum(cst, s);
        block_sum(gc, s);
        double l1 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) l1 += fabs(code[c][0]) + fabs(code[c][1]);
        const double cost = 0.5 * cst[0] + alpha * l1;
        if (it > 0 && (cost_prev - cost) < prm.dl_tol * cost) break;  // :657-665
        cost_prev = cost;
    }
#undef ODV
    // dictionary = code.T; H first (dl_output_for_h_and_e, :53-68); unit rows (:322)
    if (tid == 0) {
        double hh[3] = {code[0][0], code[1][0], code[2][0]}, e[3] = {code[0][1], code[1][1], code[2][1]};
        const bool swap = hh[0] < e[0];
        const double nh = sqrt(hh[0] * hh[0] + hh[1] * hh[1] + hh[2] * hh[2]);
        const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        for (int i = 0; i < 3; ++i) {
            const double hv = hh[i] / nh, ev = e[i] / ne;
            out[TIA_ST_STAIN + i] = swap ? ev : hv;
            out[TIA_ST_STAIN + 3 + i] = swap ? hv : ev;
        }
        out[TIA_ST_NTISSUE] = nt;
        out[TIA_ST_MINPHI] = (double)n_iter;  // Vahadane: number of dictionary-learning iterations run
    }
}


// =====================================================================================================================
// Register-resident variant (Macenko / fixed / given modes, patches of <= 65536 pixels with hw % 4 == 0)
// =====================================================================================================================
// ONE 1024-thread workgroup per patch reads the patch from HBM exactly once: thread t keeps the 4-pixel groups
// t, t + 1024, t + 2048, ... (three dwords each, at most 16 of them: 48 VGPRs) for the whole kernel, and every sweep --
// byte histogram, tissue mask + OD moments, the float32 classification sweeps of both selections -- runs out of registers.
// What leaves the registers is small: histogram increments (LDS atomics on a 32-way striped table: lane l owns copy
// l & 31, so the 32 lanes serviced together never share a bank), and the few per cent of pixels that lie inside a selection
// window, appended WITH their colour (r | g << 8 | b << 16 | need-bits << 24) to per-wave LDS lists, so the exact float64
// pass can hand them to any thread.  The statistics are the same numbers, bit for bit, as stain_stats_kernel<false>:
// same table look-ups, same per-pixel instruction sequences, the same fixed accumulation order (the moments of virtual
// thread t = groups t + 1024 j in ascending order, a wave's shuffle tree, the 16 waves in order -- the streaming kernel
// accumulates in this order too), and exact order statistics.  If a selection window cannot be placed or overflows, the
// patch is flagged in `redo` and the streaming kernel recomputes it (launched right after with the flag array).
constexpr int RT = 1024;
constexpr int RW = RT / 64;
constexpr int RG = 16;        // groups per thread
constexpr int RSEG = 384;     // list entries (16 bytes: one 4-pixel group + need-bits) per wave
constexpr int HCOPY = 32;     // histogram / OD-table copies: lane l uses copy l & 31, so a half-wave never shares a bank
constexpr int L2COPY = 8;     // copies of the float32 log2 table (the sweeps' three transcendental instructions per pixel become look-ups)
constexpr int RCAP = 2048;    // candidates per target (windows over all 65536 pixels of a 256 x 256 patch hold ~800 + slack)

struct SmemR {
    double od[256];
    int ty[3][256];
    unsigned hist[256];
    unsigned cum[256];
    union {  // one 96 KB region, used by one phase at a time
        unsigned hstripe[256 * HCOPY];   // P1: byte histogram, copy (lane & 31) of bin v at v * 32 + (lane & 31)
        double odstripe[256 * HCOPY];    // P2: the float64 OD table, striped the same way (conflict-free look-ups)
        float sbuf[2][SAMPLE_TARGET];    // selections: sample keys (window placement) ...
        uint4 list[RW * RSEG];           // ... then the sweep's lists (the sample is consumed before the sweep starts)
    };
    unsigned sbins[2][SNB];
    double cand[2][RCAP];
    float l2[256 * L2COPY];  // log2(max(v, 1)) of every byte value as the float32 sweeps compute it, copy (lane & 7) of v at v * 8 + copy
    double small[2][64];
    double red[RW][16];
    double tot[16];
    double bc[48];
    int ibc[8];
    double wlo[2], whi[2];
    unsigned long long wbelow[2];
    unsigned long long key_min[2], key_max[2];
    unsigned long long selr[2];
    unsigned wn[2];
    unsigned ncand[2];
    int sel_lo[2], sel_hi[2];
    unsigned wcnt[RW];
    int wok;
#if TIA_STATS_TIMING
    long long tm[16];
    long long tlast;
#endif
};
#if TIA_STATS_TIMING
#define RSTAMP(i)                                  \
    if (threadIdx.x == 0) {                        \
        const long long now_ = clock64();          \
        s.tm[i] += now_ - s.tlast;                 \
        s.tlast = now_;                            \
    }
#else
#define RSTAMP(i)
#endif

template <int N>
__device__ __forceinline__ void block_sum_r(double (&v)[N], SmemR& s) {
    static_assert(N <= 16, "reduction scratch too small");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double w = wave_sum(v[i]);
        if (lane_id() == 0) s.red[wave_id()][i] = w;
    }
    __syncthreads();
    // thread i adds the 16 wave partials of value i in wave order (the order block_sum uses), so nobody holds 16 x N values
    if (threadIdx.x < N) {
        double acc = 0.0;
        for (int w = 0; w < RW; ++w) acc += s.red[w][threadIdx.x];
        s.tot[threadIdx.x] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = s.tot[i];
    __syncthreads();
}

// window_select2 for register-resident pixels.  `sample32(idx, r, g, b, v)` and `exact(r, g, b, x)` as there;
// `sweep(seg, cap, count, below0, below1)` classifies the calling thread's own pixels in float32 and appends the undecided
// ones (colour + need-bits) to this wave's list segment.  Returns false (workgroup-uniform) when a precondition fails.
template <class FETCH, class SAMPLE32, class EXACT, class SWEEP>
__device__ __forceinline__ bool window_select_reg(FETCH&& fetch, long hw, bool shared_keys, SAMPLE32&& sample32,
                                                  EXACT&& exact, SWEEP&& sweep, SmemR& s, const unsigned long long (&k)[2],
                                                  const unsigned long long (&n)[2], double (&vprev)[2], double (&vnext)[2]) {
    const int tid = threadIdx.x;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const float finf = __int_as_float(0x7f800000);
    if (n[0] == 0 || n[1] == 0) return false;
    constexpr int SPT = SAMPLE_TARGET / RT;  // samples per thread
    if (tid < 2) {
        s.key_max[tid] = 0ull;
        s.key_min[tid] = ~0ull;
        s.wn[tid] = 0u;
        s.wbelow[tid] = 0ull;
    }
    for (int i = tid; i < 2 * SNB; i += RT) (&s.sbins[0][0])[i] = 0u;
    const float fnan = __int_as_float(0x7fc00000);
    {
        // the sample comes out of the thread's own registers: thread t holds the groups t + 1024 j -- one every 16 rows of a 256-wide
        // patch, at a column position that runs over the whole row with t -- so one pixel from each quarter of its slots, at a hashed
        // slot and pixel, is a stratified sample over the image (`fetch`).  (Round 3 re-read a strided sample from memory: every
        // sampled byte pulled a whole 64-byte sector, i.e. both selections together re-read ~1.6x the patch: the kernel's HBM-side
        // traffic was 2.8x the patch instead of ~1.2x, profiles/r04s_stain_pmc_*.)
        uint32_t rgb[SPT];
        bool have[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) have[j] = fetch(j, rgb[j]);
        float mn[2] = {finf, finf}, mx[2] = {-finf, -finf};
        unsigned cnt[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            float v[2] = {0.0f, 0.0f};
            const unsigned valid = have[j] ? sample32(0L, rgb[j] & 255u, (rgb[j] >> 8) & 255u, (rgb[j] >> 16) & 255u, v) : 0u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool ok = (valid >> t) & 1u;
                s.sbuf[t][j * RT + tid] = ok ? v[t] : fnan;
                mn[t] = ok ? fminf(mn[t], v[t]) : mn[t];
                mx[t] = ok ? fmaxf(mx[t], v[t]) : mx[t];
                cnt[t] += ok ? 1u : 0u;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn[t] = fminf(mn[t], __shfl_down(mn[t], o, 64));
                mx[t] = fmaxf(mx[t], __shfl_down(mx[t], o, 64));
                cnt[t] += __shfl_down(cnt[t], o, 64);
            }
        }
        __syncthreads();  // zeroing above done
        if (lane_id() == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                atomicMin(&s.key_min[t], f64_key((double)mn[t]));
                atomicMax(&s.key_max[t], f64_key((double)mx[t]));
                atomicAdd(&s.wn[t], cnt[t]);
            }
        }
    }
    __syncthreads();
    const unsigned ns[2] = {s.wn[0], s.wn[1]};
    if (ns[0] < 64u || ns[1] < 64u) return false;
    float smin[2], sscale[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const double lo = key_f64(s.key_min[t]), hi = key_f64(s.key_max[t]);
        const double sc = (double)SNB / (hi - lo);
        smin[t] = (float)lo;
        sscale[t] = (hi > lo && sc > 0.0 && sc < 1.0e30) ? (float)sc : 0.0f;
    }
    // (both targets of the angular selection see the same keys: one histogram then serves both)
    for (int t = 0; t < (shared_keys ? 1 : 2); ++t)
        for (int i = tid; i < SAMPLE_TARGET; i += RT) {
            const float v = s.sbuf[t][i];
            if (v == v) {
                const float d = (v - smin[t]) * sscale[t];
                const int b = !(d >= 0.0f) ? 0 : (d >= (float)SNB ? SNB - 1 : (int)d);
                atomicAdd(&s.sbins[t][b], 1u);
            }
        }
    __syncthreads();
    constexpr int PER = SNB / 64;
    auto bin_of_rank = [&](const unsigned (&local)[PER], unsigned incl, unsigned sum, unsigned r, unsigned& before_bin) -> int {
        unsigned before = incl - sum;
        int found = SNB;
        unsigned fb = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const unsigned after = before + local[i];
            if (found == SNB && after > r) {
                found = lane_id() * PER + i;
                fb = before;
            }
            before = after;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(found, o, 64);
            const unsigned ob = __shfl_xor(fb, o, 64);
            if (other < found) {
                found = other;
                fb = ob;
            }
        }
        before_bin = fb;
        return found;
    };
    if (wave_id() < 2) {
        const int t = wave_id();
        const int lane = lane_id();
        unsigned local[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            local[i] = s.sbins[shared_keys ? 0 : t][lane * PER + i];
            sum += local[i];
        }
        const unsigned incl = wave_incl_scan_u32(sum);
        const double q = ((double)k[t] + 0.5) / (double)n[t];
        const double centre = q * (double)ns[t];
        const double sigma = sqrt((double)ns[t] * q * (1.0 - q));
        const double rlo = floor(centre - 3.5 * sigma - 2.0), rhi = ceil(centre + 3.5 * sigma + 2.0);
        unsigned dummy;
        const int blo = rlo < 0.0 ? -1 : bin_of_rank(local, incl, sum, (unsigned)rlo, dummy);
        const int bhi = rhi >= (double)ns[t] ? SNB : bin_of_rank(local, incl, sum, (unsigned)rhi, dummy);
        if (lane == 0) {
            const double sc = (double)sscale[t];
            const bool flat = !(sc > 0.0);
            s.wlo[t] = (flat || blo <= 1) ? -inf : (double)smin[t] + (double)(blo - 1) / sc;
            s.whi[t] = (flat || bhi >= SNB - 2) ? inf : (double)smin[t] + (double)(bhi + 2) / sc;
        }
    }
    __syncthreads();
    if (tid < 2) {
        s.wn[tid] = 0u;
        s.ncand[tid] = 0u;
        s.key_max[tid] = 0ull;
        s.key_min[tid] = ~0ull;
    }
    for (int i = tid; i < 2 * SNB; i += RT) (&s.sbins[0][0])[i] = 0u;
    __syncthreads();
    RSTAMP(TM_SEL_FIND)
    // ---- the float32 sweep over the thread's own pixels ---------------------------------------------------------------------
    {
        unsigned count = 0, bl0 = 0, bl1 = 0;  // count: wave-uniform; bl0 / bl1: per-lane counts of "definitely below"
        sweep(s.list + wave_id() * RSEG, (unsigned)RSEG, count, bl0, bl1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            bl0 += __shfl_down(bl0, o, 64);
            bl1 += __shfl_down(bl1, o, 64);
        }
        if (lane_id() == 0) {
            s.wcnt[wave_id()] = count;
            if (bl0) atomicAdd(&s.wbelow[0], (unsigned long long)bl0);
            if (bl1) atomicAdd(&s.wbelow[1], (unsigned long long)bl1);
        }
    }
    __syncthreads();
    RSTAMP(TM_SEL_HIST)
    {
        bool over = false;
        for (int w = 0; w < RW; ++w) over = over || s.wcnt[w] > (unsigned)RSEG;
        if (over) return false;  // uniform
    }
    // ---- exact classification of the listed pixels: entry e of the concatenated lists goes to thread e % 1024 -----------------
    {
        unsigned bl[2] = {0u, 0u};
        unsigned long long mn[2] = {~0ull, ~0ull}, mx[2] = {0ull, 0ull};
        unsigned total = 0;
        for (int w = 0; w < RW; ++w) total += s.wcnt[w];
        for (unsigned i = tid; i < total; i += RT) {
            int w = 0;
            unsigned base = 0, acc = 0;
            for (int v = 0; v < RW; ++v) {
                if (i >= acc) {
                    w = v;
                    base = acc;
                }
                acc += s.wcnt[v];
            }
            const uint4 en = s.list[w * RSEG + (i - base)];
            uint32_t rr[4], gg[4], bb[4];
            unpack_group(en.x, en.y, en.z, rr, gg, bb);
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const unsigned need = (en.w >> (2 * px)) & 3u;
                if (!need) continue;
                double x[2];
                exact(rr[px], gg[px], bb[px], x);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (!((need >> t) & 1u)) continue;
                    if (x[t] < s.wlo[t]) {
                        ++bl[t];
                    } else if (!(x[t] > s.whi[t])) {
                        const unsigned pos = atomicAdd(&s.wn[t], 1u);
                        if (pos < (unsigned)RCAP) s.cand[t][pos] = x[t];
                        const unsigned long long key = f64_key(x[t]);
                        mn[t] = key < mn[t] ? key : mn[t];
                        mx[t] = key > mx[t] ? key : mx[t];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned c = bl[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
            const unsigned long long a2 = wave_min_u64(mn[t]);
            const unsigned long long b2 = ~wave_min_u64(~mx[t]);
            if (lane_id() == 0) {
                if (c) atomicAdd(&s.wbelow[t], (unsigned long long)c);
                atomicMin(&s.key_min[t], a2);
                atomicMax(&s.key_max[t], b2);
            }
        }
    }
    __syncthreads();
    RSTAMP(TM_SEL_COLLECT)
    if (tid == 0) {
        int ok = 1;
        for (int t = 0; t < 2; ++t) {
            const unsigned long long below = s.wbelow[t], nc = s.wn[t];
            const bool has_next = k[t] + 1 < n[t];
            if (nc > (unsigned long long)RCAP || k[t] < below || k[t] + (has_next ? 1 : 0) >= below + nc) ok = 0;
        }
        s.wok = ok;
#if TIA_STATS_TIMING
        s.tm[11] += s.wn[0];
        s.tm[12] += s.wn[1];
        for (int w = 0; w < RW; ++w) s.tm[13] += s.wcnt[w];
#endif
    }
    __syncthreads();
    if (!s.wok) return false;
    // ---- refine inside the candidate set ------------------------------------------------------------------------------------
    const unsigned nc[2] = {s.wn[0], s.wn[1]};
    double clo[2], csc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const double lo = key_f64(s.key_min[t]), hi = key_f64(s.key_max[t]);
        const double sc = (double)SNB / (hi - lo);
        clo[t] = lo;
        csc[t] = (hi > lo && sc > 0.0 && sc < 1.0e300) ? sc : 0.0;
    }
    auto cbin = [&](int t, double x) -> int {
        const double d = (x - clo[t]) * csc[t];
        return !(d >= 0.0) ? 0 : (d >= (double)SNB ? SNB - 1 : (int)d);
    };
#pragma unroll
    for (int t = 0; t < 2; ++t)
        for (unsigned i = tid; i < nc[t]; i += RT) atomicAdd(&s.sbins[t][cbin(t, s.cand[t][i])], 1u);
    if (tid < 2) s.ncand[tid] = 0u;
    __syncthreads();
    if (wave_id() < 2) {
        const int t = wave_id();
        const int lane = lane_id();
        unsigned local[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            local[i] = s.sbins[t][lane * PER + i];
            sum += local[i];
        }
        const unsigned incl = wave_incl_scan_u32(sum);
        const unsigned long long r = k[t] - s.wbelow[t];
        const bool has_next = k[t] + 1 < n[t];
        unsigned before_a = 0, before_b = 0;
        const int ba = bin_of_rank(local, incl, sum, (unsigned)r, before_a);
        const int bb = has_next ? bin_of_rank(local, incl, sum, (unsigned)r + 1u, before_b) : ba;
        if (lane == 0) {
            s.sel_lo[t] = ba;
            s.sel_hi[t] = bb;
            s.selr[t] = r - before_a;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ba = s.sel_lo[t], bb = s.sel_hi[t];
        for (unsigned i = tid; i < nc[t]; i += RT) {
            const double x = s.cand[t][i];
            const int b = cbin(t, x);
            if (b == ba || b == bb) {
                const unsigned pos = atomicAdd(&s.ncand[t], 1u);
                if (pos < 64u) s.small[t][pos] = x;
            }
        }
    }
    __syncthreads();
    if (s.ncand[0] > 64u || s.ncand[1] > 64u) {  // a crowded bin (massive ties): order the whole candidate set instead
        unsigned pmax = 2;
        for (int t = 0; t < 2; ++t) {
            unsigned pp = 2;
            while (pp < nc[t]) pp <<= 1;
            pmax = pp > pmax ? pp : pmax;
        }
        for (int t = 0; t < 2; ++t)
            for (unsigned i = nc[t] + tid; i < pmax; i += RT) s.cand[t][i] = inf;
        __syncthreads();
        for (unsigned kk = 2; kk <= pmax; kk <<= 1) {
            for (unsigned j = kk >> 1; j > 0; j >>= 1) {
                for (unsigned i = tid; i < pmax; i += RT) {
                    const unsigned partner = i ^ j;
                    if (partner > i) {
                        const bool asc = (i & kk) == 0;
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const double a = s.cand[t][i], b = s.cand[t][partner];
                            if ((a > b) == asc) {
                                s.cand[t][i] = b;
                                s.cand[t][partner] = a;
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned long long r = k[t] - s.wbelow[t];
            vprev[t] = s.cand[t][r];
            vnext[t] = (k[t] + 1 < n[t]) ? s.cand[t][r + 1] : vprev[t];
        }
        __syncthreads();
        return true;
    }
    if (wave_id() < 2) {  // rank by counting inside one wave
        const int t = wave_id();
        const int lane = lane_id();
        const unsigned m = s.ncand[t];
        const double x = (unsigned)lane < m ? s.small[t][lane] : inf;
        unsigned rank = 0;
        for (unsigned j = 0; j < m; ++j) {
            const double y = s.small[t][j];
            rank += (y < x || (y == x && j < (unsigned)lane)) ? 1u : 0u;
        }
        const unsigned long long r = s.selr[t];
        if ((unsigned)lane < m && rank == (unsigned)r) s.bc[40 + 2 * t] = x;
        if ((unsigned)lane < m && rank == (unsigned)r + 1u) s.bc[41 + 2 * t] = x;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        vprev[t] = s.bc[40 + 2 * t];
        vnext[t] = (k[t] + 1 < n[t]) ? s.bc[41 + 2 * t] : vprev[t];
    }
    __syncthreads();
    RSTAMP(TM_SEL_SORT)
    return true;
}

// The patch registers are LLVM vectors: a loop over the groups with a (wave-uniform) run-time index then compiles to indexed
// register moves (s_set_gpr_idx / v_movrel) instead of 16 unrolled copies of every sweep -- unrolled, the kernel is > 100 KB of
// straight-line code that every wave streams through the instruction cache once per patch.
using u32x16 = uint32_t __attribute__((ext_vector_type(16)));

// Workgroup-uniform values (read from LDS or computed from such) moved to scalar registers: they are live across the sweeps,
// and the vector registers are needed for the patch.
__device__ __forceinline__ float uni(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ double uni(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Single-lane arithmetic of the register-resident kernel as REAL calls: their (large) register needs then do not add to the
// kernel's own allocation, which is dominated by the 48 registers of the patch; only the calling wave pays the call.
__device__ __attribute__((noinline)) void reg_eigen(const double (&acc)[10], double (&cov)[6], double (&e1)[3], double (&e2)[3]) {
    const double nt = acc[0];
    const double mx = acc[1] / nt, my = acc[2] / nt, mz = acc[3] / nt;
    const double f = 1.0 / (nt - 1.0);
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
    e1[0] = v[0][i0]; e1[1] = v[1][i0]; e1[2] = v[2][i0];
    e2[0] = v[0][i1]; e2[1] = v[1][i1]; e2[2] = v[2][i1];
    if (e1[0] < 0) { e1[0] = -e1[0]; e1[1] = -e1[1]; e1[2] = -e1[2]; }
    if (e2[0] < 0) { e2[0] = -e2[0]; e2[1] = -e2[1]; e2[2] = -e2[2]; }
}
__device__ __attribute__((noinline)) void reg_stain_from_angles(const double (&vp)[2], const double (&vn)[2], const double (&gm)[2],
                                                                const double (&e1)[3], const double (&e2)[3], double (&phi)[2],
                                                                double (&hv)[3], double (&ev)[3]) {
    const double min_phi = np_lerp(angle_of_key(vp[0]), angle_of_key(vn[0]), gm[0]);
    const double max_phi = np_lerp(angle_of_key(vp[1]), angle_of_key(vn[1]), gm[1]);
    phi[0] = min_phi;
    phi[1] = max_phi;
    const double c1 = cos(min_phi), s1 = sin(min_phi), c2 = cos(max_phi), s2 = sin(max_phi);
    double v1[3] = {e1[0] * c1 + e2[0] * s1, e1[1] * c1 + e2[1] * s1, e1[2] * c1 + e2[2] * s1};
    double v2[3] = {e1[0] * c2 + e2[0] * s2, e1[1] * c2 + e2[1] * s2, e1[2] * c2 + e2[2] * s2};
    const bool first = v1[0] > v2[0];
    const double* h = first ? v1 : v2;
    const double* e = first ? v2 : v1;
    const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    for (int i = 0; i < 3; ++i) {
        hv[i] = h[i] / nh;
        ev[i] = e[i] / ne;
    }
}

__global__ __launch_bounds__(RT) void stain_stats_reg_kernel(const uint8_t* __restrict__ img, long hw,
                                                              const tia_stain_tables* __restrict__ tab, tia_stain_params prm,
                                                              double* __restrict__ stats, int* __restrict__ redo,
                                                              uint32_t* __restrict__ sample_ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    SmemR& s = *reinterpret_cast<SmemR*>(smem_raw);
    const uint8_t* p = img + (size_t)blockIdx.x * (size_t)hw * 3u;
    const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
    double* out = stats + (size_t)blockIdx.x * TIA_STATS_STRIDE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool z1 = prm.zero_to_one != 0;
    const int ng = (int)(hw >> 2);

    // ---- the patch: groups tid + 1024 j, all loads in flight together ------------------------------------------------------
    static_assert(RG == 16, "the patch registers are 16-wide vectors");
    u32x16 pa, pb, pc;
#pragma unroll
    for (int j = 0; j < RG; ++j) {
        const int g = tid + RT * j;
        const int gc = g < ng ? g : ng - 1;  // clamped (branch-free loads); slots beyond the patch are never used
        pa[j] = q[gc * 3 + 0];
        pb[j] = q[gc * 3 + 1];
        pc[j] = q[gc * 3 + 2];
    }
    const int n_slots = (ng + RT - 1) / RT;  // group slots in use (workgroup-uniform)
    // The window-placing sample (4 pixels per thread: the stratified sample of sample_index) is requested HERE, right behind the
    // patch itself -- its bytes sit in lines the workgroup's own coalesced loads are bringing into L2 at this moment -- and parked
    // in 16 KB of the patch's (otherwise unused) bin-cache workspace once P1 has run; each selection reads its 4 values back with
    // one coalesced load.  (Round 3 re-read the sample from the image in front of each selection, long after the lines had left
    // L2: every sampled byte pulled a whole 64-byte sector, both selections together re-read ~1.6x the patch and the kernel's
    // HBM-side traffic was 2.8x the patch, profiles/r04s_stain_pmc_*.  Selecting the sample out of the patch registers instead
    // -- a per-lane select chain over the 48 registers -- was measured too: at the point of use it spilled 45 more registers (3.97 ms),
    // at kernel start it made P1 wait for the whole patch and slowed the later sweeps (2.95 ms), profiles/r04t_*, r04u_*.)
    uint32_t* __restrict__ my_samples = sample_ws + (size_t)blockIdx.x * (size_t)hw;  // the patch's own 4 hw bytes of the bin cache
    constexpr int SPT_R = SAMPLE_TARGET / RT;
    uint32_t srgb[SPT_R];
    {
        const long sstride = (hw + SAMPLE_TARGET - 1) / SAMPLE_TARGET;
#pragma unroll
        for (int k = 0; k < SPT_R; ++k) {
            const long idx = sample_index((long)k * RT + tid, sstride);
            const long ic = idx < hw ? idx : hw - 1;
            srgb[k] = ((uint32_t)p[3 * ic] | ((uint32_t)p[3 * ic + 1] << 8) | ((uint32_t)p[3 * ic + 2] << 16)) |
                      (idx < hw ? 0x80000000u : 0u);  // bit 31: a pixel of the patch
        }
    }
    auto fetch_sample = [&](int k, uint32_t& rgb) -> bool {
        const uint32_t v = my_samples[k * RT + tid];
        rgb = v & 0xffffffu;
        return (v >> 31) != 0u;
    };
    double s_given[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (prm.mode == TIA_MODE_GIVEN) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s_given[i] = out[TIA_ST_STAIN + i];
        __syncthreads();
    }
    auto give_up = [&]() {  // workgroup-uniform: hand the patch to the streaming kernel
        if (tid == 0) {
            if (prm.mode == TIA_MODE_GIVEN)
                for (int i = 0; i < 6; ++i) out[TIA_ST_STAIN + i] = s_given[i];
            redo[blockIdx.x] = 1;
        }
    };
#if TIA_STATS_TIMING
    if (tid == 0) {
        for (int i = 0; i < 16; ++i) s.tm[i] = 0;
        s.tlast = clock64();
    }
    const long long t_begin = clock64();
#endif
    if (tid < TIA_STATS_STRIDE) out[tid] = 0.0;
    if (tid < 256) s.od[tid] = tab->od_lut[tid];
    // the SAME instruction the streaming kernel issues per pixel and channel, evaluated once per byte value: identical bits, and
    // the two classification sweeps (12 of their ~58 issue slots per pixel were v_log_f32) read it back from LDS
    for (int i = tid; i < 256 * L2COPY; i += RT) s.l2[i] = __log2f(fmaxf((float)(i / L2COPY), 1.0f));
    for (int i = tid; i < 256 * HCOPY; i += RT) s.hstripe[i] = 0u;
    __syncthreads();

    // ---- P1: byte histogram of all three channels together (the percentiles are over the flattened image) ------------------
    {
        unsigned* hs = s.hstripe + (lane & (HCOPY - 1));
#pragma unroll 1
        for (int j = 0; j < n_slots; ++j) {
            if (tid + RT * j < ng) {
                const uint32_t w[3] = {pa[j], pb[j], pc[j]};
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t v = (w[d] >> (8 * e)) & 255u;
                        if (z1) v = v ? v : 1u;
                        atomicAdd(hs + v * HCOPY, 1u);
                    }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT_R; ++k) my_samples[k * RT + tid] = srgb[k];
    RSTAMP(TM_P1)
    if (tid < 256) {
        unsigned tot = 0;
#pragma unroll 8
        for (int c = 0; c < HCOPY; ++c) tot += s.hstripe[tid * HCOPY + ((c + lane) & (HCOPY - 1))];  // rotated: no bank conflicts
        s.hist[tid] = tot;
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned h0 = s.hist[tid * 4], h1 = s.hist[tid * 4 + 1], h2 = s.hist[tid * 4 + 2], h3 = s.hist[tid * 4 + 3];
        const unsigned incl = wave_incl_scan_u32(h0 + h1 + h2 + h3);
        const unsigned base = incl - (h0 + h1 + h2 + h3);
        s.cum[tid * 4] = base + h0;
        s.cum[tid * 4 + 1] = base + h0 + h1;
        s.cum[tid * 4 + 2] = base + h0 + h1 + h2;
        s.cum[tid * 4 + 3] = incl;
    }
    __syncthreads();
    {
        const unsigned long long nbytes = (unsigned long long)hw * 3ull;
        unsigned long long kp[2], kn[2];
        double gm[2];
        np_index(nbytes, prm.q_img_lo, kp[0], kn[0], gm[0]);
        np_index(nbytes, prm.q_img_hi, kp[1], kn[1], gm[1]);
        if (tid < 256) {
            const unsigned long long c1 = s.cum[tid], c0 = tid ? s.cum[tid - 1] : 0;
            if (c0 <= kp[0] && kp[0] < c1) s.ibc[0] = tid;
            if (c0 <= kn[0] && kn[0] < c1) s.ibc[1] = tid;
            if (c0 <= kp[1] && kp[1] < c1) s.ibc[2] = tid;
            if (c0 <= kn[1] && kn[1] < c1) s.ibc[3] = tid;
            if (c0 == 0 && c1 > 0) s.ibc[4] = tid;
            if (c1 == (unsigned)nbytes && c0 < (unsigned)nbytes) s.ibc[5] = tid;
        }
        __syncthreads();
        if (tid == 0) {
            double plow = np_lerp((double)s.ibc[0], (double)s.ibc[1], gm[0]);
            double phigh = np_lerp((double)s.ibc[2], (double)s.ibc[3], gm[1]);
            if (plow >= phigh) {
                plow = (double)s.ibc[4];
                phigh = (double)s.ibc[5];
            }
            s.bc[0] = plow;
            s.bc[1] = phigh;
            out[TIA_ST_PLOW] = plow;
            out[TIA_ST_PHIGH] = phigh;
        }
        __syncthreads();
    }
    if (tid < 256) {  // contrast_enhancer LUT folded into the luminance tables (see stain_stats_kernel)
        const double plow = s.bc[0], phigh = s.bc[1];
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
        s.ty[0][tid] = tab->ty[0][ce];
        s.ty[1][tid] = tab->ty[1][ce];
        s.ty[2][tid] = tab->ty[2][ce];
    }
    __syncthreads();
    RSTAMP(TM_LUT)
    const int y_thr = prm.y_thr;
    // append one 16-byte entry (a 4-pixel group + its need-bits) per lane that has one to this wave's private list segment:
    // position = wave count (uniform) + number of appending lanes below this one (v_mbcnt); no atomics
    auto seg_push = [&](bool need, const uint4& entry, uint4* seg, unsigned cap, unsigned& count) {
        const unsigned long long m = __ballot(need);
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const unsigned pos = count + before;
        if (need && pos < cap) seg[pos] = entry;
        count += (unsigned)__popcll(m);
    };
    auto sgn = [](float v) -> unsigned { return __float_as_uint(v) >> 31; };  // 1 iff v < 0 (v is never NaN where it counts)

    unsigned flags = 0;
    if (prm.mode == TIA_MODE_MACENKO) {
        // ---- P2: tissue mask (kept as bits in two registers) + OD moments, out of the registers --------------------------------
        for (int i = tid; i < 256 * HCOPY; i += RT) s.odstripe[i] = s.od[i / HCOPY];  // the histogram is consumed: its LDS takes
        __syncthreads();                                                              // the striped OD table
        const double* ods = s.odstripe + (lane & (HCOPY - 1));
        double acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0.0;
        unsigned long long tmask = 0ull;
#pragma unroll 1
        for (int j = 0; j < n_slots; ++j) {
            const bool valid = tid + RT * j < ng;
            uint32_t rr[4], gg[4], bb[4];
            unpack_group(pa[j], pb[j], pc[j], rr, gg, bb);
            // the luminance look-ups of all four pixels and the OD look-ups of two at a time are in flight together; the
            // accumulation order (pixel 0, 1, 2, 3 of the group) is that of stain_stats_kernel
            int lum[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) lum[i] = s.ty[0][rr[i]] + s.ty[1][gg[i]] + s.ty[2][bb[i]];
            unsigned nib = 0;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                double x[2], y[2], z[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    x[i] = ods[rr[2 * h2 + i] * HCOPY];
                    y[i] = ods[gg[2 * h2 + i] * HCOPY];
                    z[i] = ods[bb[2 * h2 + i] * HCOPY];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (valid && ((lum[2 * h2 + i] + (1 << 11)) >> 12) < y_thr) {
                        nib |= 1u << (2 * h2 + i);
                        acc[0] += 1.0;
                        acc[1] += x[i];
                        acc[2] += y[i];
                        acc[3] += z[i];
                        acc[4] = __builtin_fma(x[i], x[i], acc[4]);
                        acc[5] = __builtin_fma(x[i], y[i], acc[5]);
                        acc[6] = __builtin_fma(x[i], z[i], acc[6]);
                        acc[7] = __builtin_fma(y[i], y[i], acc[7]);
                        acc[8] = __builtin_fma(y[i], z[i], acc[8]);
                        acc[9] = __builtin_fma(z[i], z[i], acc[9]);
                    }
                }
            }
            tmask |= (unsigned long long)nib << (4 * j);
        }
        block_sum_r(acc, s);
        RSTAMP(TM_P2)
        const double nt = acc[0];
        const unsigned long long n_tissue = (unsigned long long)nt;
        if (n_tissue == 0) {
            if (tid == 0) {
                out[TIA_ST_NTISSUE] = 0.0;
                out[TIA_ST_FLAGS] = (double)TIA_FLAG_EMPTY_MASK;
            }
            return;  // uniform across the block
        }
        if (n_tissue < 2) flags |= TIA_FLAG_DEGENERATE;
        if (tid == 0) {
            double cov[6], e1[3], e2[3], mom[10];
            for (int i = 0; i < 10; ++i) mom[i] = acc[i];  // only this lane's copy goes through memory (the callee takes references)
            reg_eigen(mom, cov, e1, e2);
            for (int i = 0; i < 6; ++i) out[TIA_ST_COV + i] = cov[i];
            for (int i = 0; i < 3; ++i) {
                s.bc[2 + i] = e1[i];
                s.bc[5 + i] = e2[i];
                out[TIA_ST_EVEC + i] = e1[i];
                out[TIA_ST_EVEC + 3 + i] = e2[i];
            }
            out[TIA_ST_NTISSUE] = nt;
        }
        __syncthreads();
        RSTAMP(TM_EIG)
        // the eigenvectors stay in LDS (s.bc[2..7]); only the float32 images the sweep needs go to (scalar) registers

        // ---- P3: exact percentiles of phi over the tissue pixels (see stain_stats_kernel for the error budget) ------------------
        unsigned long long kp[2], kn[2], nn[2] = {n_tissue, n_tissue};
        double gm[2];
        np_index(n_tissue, prm.q_phi_lo, kp[0], kn[0], gm[0]);
        np_index(n_tissue, prm.q_phi_hi, kp[1], kn[1], gm[1]);
        double vp[2], vn[2];
        const float ln2 = 0.6931471805599453f, l255 = 7.994353436858858f;
        const float ex0 = uni(ln2 * (float)s.bc[2]), ex1 = uni(ln2 * (float)s.bc[3]), ex2 = uni(ln2 * (float)s.bc[4]);
        const float ey0 = uni(ln2 * (float)s.bc[5]), ey1 = uni(ln2 * (float)s.bc[6]), ey2 = uni(ln2 * (float)s.bc[7]);
        const float kx = uni(l255 * (ex0 + ex1 + ex2)), ky = uni(l255 * (ey0 + ey1 + ey2));
        const float tol = 8.0e-5f;
        const float* l2t = s.l2 + (lane & (L2COPY - 1));
        auto proj = [&](uint32_t r, uint32_t g, uint32_t b, float& x, float& y) {
            const float lr = l2t[r * L2COPY], lg = l2t[g * L2COPY], lb = l2t[b * L2COPY];
            x = fmaf(-ex2, lb, fmaf(-ex1, lg, fmaf(-ex0, lr, kx)));
            y = fmaf(-ey2, lb, fmaf(-ey1, lg, fmaf(-ey0, lr, ky)));
        };
        const bool ok = window_select_reg(
            fetch_sample, hw, true,
            [&](long, uint32_t r, uint32_t g, uint32_t b, float (&v)[2]) -> unsigned {
                const int t = s.ty[0][r] + s.ty[1][g] + s.ty[2][b];
                if (!(((t + (1 << 11)) >> 12) < y_thr)) return 0u;
                float x, y;
                proj(r, g, b, x, y);
                const float d = fabsf(x) + fabsf(y);
                const float qv = d > 0.0f ? y / d : 0.0f;
                v[0] = v[1] = x >= 0.0f ? qv : (y >= 0.0f ? 2.0f - qv : -2.0f - qv);
                return 3u;
            },
            [&](uint32_t r, uint32_t g, uint32_t b, double (&x)[2]) {
                const double ox = s.od[r], oy = s.od[g], oz = s.od[b];
                const double p0 = dot3(ox, oy, oz, s.bc[2], s.bc[3], s.bc[4]);
                const double p1 = dot3(ox, oy, oz, s.bc[5], s.bc[6], s.bc[7]);
                x[0] = x[1] = pseudo_angle(p1, p0);
            },
            [&](uint4* seg, unsigned cap, unsigned& count, unsigned& bl0, unsigned& bl1) {
                const double w[4] = {s.wlo[0], s.whi[0], s.wlo[1], s.whi[1]};
                bool edges_ok = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) edges_ok = edges_ok && (!(fabs(w[i]) < 1e300) || fabs(w[i]) <= 1.0);
                const unsigned eok = edges_ok ? 1u : 0u;
                const float lo0 = uni((float)w[0]), hi0 = uni((float)w[1]), lo1 = uni((float)w[2]), hi1 = uni((float)w[3]);
                // all predicates as 0 / 1 integers from sign bits (VALU only: no compare -> scalar mask -> select round trips):
                // below <=> s + tol < 0, above <=> tol - s < 0, plain <=> tol - x < 0; NaNs (an infinite edge times d = 0) can
                // only arise where plain = 0, which masks them
#pragma unroll 1
                for (int j = 0; j < n_slots; ++j) {
                    const unsigned valid = tid + RT * j < ng ? 1u : 0u;
                    uint32_t rr[4], gg[4], bb[4];
                    unpack_group(pa[j], pb[j], pc[j], rr, gg, bb);
                    unsigned fl = 0u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned tb = (unsigned)(tmask >> (4 * j + i)) & valid;
                        float x, y;
                        proj(rr[i], gg[i], bb[i], x, y);
                        const float d = fabsf(x) + fabsf(y);
                        const unsigned pl = sgn(tol - x) & eok;
                        const unsigned bel0 = sgn(fmaf(-lo0, d, y) + tol), abv0 = sgn(tol - fmaf(-hi0, d, y));
                        const unsigned bel1 = sgn(fmaf(-lo1, d, y) + tol), abv1 = sgn(tol - fmaf(-hi1, d, y));
                        bl0 += tb & pl & bel0;
                        bl1 += tb & pl & bel1;
                        const unsigned dec0 = pl & (bel0 | abv0), dec1 = pl & (bel1 | abv1);
                        fl |= ((tb & (dec0 ^ 1u)) | ((tb & (dec1 ^ 1u)) << 1)) << (2 * i);
                    }
                    seg_push(fl != 0u, make_uint4(pa[j], pb[j], pc[j], fl), seg, cap, count);
                }
            },
            s, kp, nn, vp, vn);
        if (!ok) {
            give_up();
            return;
        }
        if (tid == 0) {
            const double e1[3] = {s.bc[2], s.bc[3], s.bc[4]}, e2[3] = {s.bc[5], s.bc[6], s.bc[7]};
            double phi[2], hv[3], ev[3];
            reg_stain_from_angles(vp, vn, gm, e1, e2, phi, hv, ev);
            out[TIA_ST_MINPHI] = phi[0];
            out[TIA_ST_MAXPHI] = phi[1];
            for (int i = 0; i < 3; ++i) {
                s.bc[8 + i] = hv[i];
                s.bc[11 + i] = ev[i];
            }
        }
    } else {
        if (tid == 0)
            for (int i = 0; i < 6; ++i) s.bc[8 + i] = prm.mode == TIA_MODE_GIVEN ? s_given[i] : prm.stain_fixed[i];
    }

#if TIA_STATS_TIMING
    if (tid == 0) s.tm[TM_PHI_TOTAL] = clock64() - t_begin;
#endif
    // ---- pseudo-inverse (stain matrix S = s.bc[8..13], P = s.bc[14..19]: both stay in LDS) ---------------------------------------
    if (tid == 0) {
        double S[6], P[6];
        for (int i = 0; i < 6; ++i) S[i] = s.bc[8 + i];
        const double a = S[0] * S[0] + S[1] * S[1] + S[2] * S[2];
        const double bb = S[0] * S[3] + S[1] * S[4] + S[2] * S[5];
        const double d = S[3] * S[3] + S[4] * S[4] + S[5] * S[5];
        const double det = a * d - bb * bb;
        const double g00 = d / det, g01 = -bb / det, g11 = a / det;
        for (int j = 0; j < 3; ++j) {
            P[j * 2 + 0] = S[j] * g00 + S[3 + j] * g01;
            P[j * 2 + 1] = S[j] * g01 + S[3 + j] * g11;
        }
        for (int i = 0; i < 6; ++i) s.bc[14 + i] = P[i];
    }
    __syncthreads();
    const double* P = &s.bc[14];

    // ---- P4: exact percentile of both concentration channels over ALL pixels ----------------------------------------------------
    double maxc[2];
    {
        const unsigned long long npx = (unsigned long long)hw;
        unsigned long long kp[2], kn[2], nn[2] = {npx, npx};
        double gm[2];
        np_index(npx, prm.q_conc, kp[0], kn[0], gm[0]);
        kp[1] = kp[0];
        kn[1] = kn[0];
        gm[1] = gm[0];
        double vp[2], vn[2];
        const float ln2 = 0.6931471805599453f, l255 = 7.994353436858858f;
        const float a0 = uni(ln2 * (float)P[0]), a1 = uni(ln2 * (float)P[2]), a2 = uni(ln2 * (float)P[4]);
        const float b0 = uni(ln2 * (float)P[1]), b1 = uni(ln2 * (float)P[3]), b2 = uni(ln2 * (float)P[5]);
        const float ka = uni(l255 * (a0 + a1 + a2)), kb = uni(l255 * (b0 + b1 + b2));
        const float tol0 = uni(3.2e-5f * (fabsf((float)P[0]) + fabsf((float)P[2]) + fabsf((float)P[4])) + 1e-7f);
        const float tol1 = uni(3.2e-5f * (fabsf((float)P[1]) + fabsf((float)P[3]) + fabsf((float)P[5])) + 1e-7f);
        const float* l2c = s.l2 + (lane & (L2COPY - 1));
        auto conc32 = [&](uint32_t r, uint32_t g, uint32_t b, float& c0, float& c1) {
            const float lr = l2c[r * L2COPY], lg = l2c[g * L2COPY], lb = l2c[b * L2COPY];
            c0 = fmaf(-a2, lb, fmaf(-a1, lg, fmaf(-a0, lr, ka)));
            c1 = fmaf(-b2, lb, fmaf(-b1, lg, fmaf(-b0, lr, kb)));
        };
        const bool ok = window_select_reg(
            fetch_sample, hw, false,
            [&](long, uint32_t r, uint32_t g, uint32_t b, float (&v)[2]) -> unsigned {
                conc32(r, g, b, v[0], v[1]);
                return 3u;
            },
            [&](uint32_t r, uint32_t g, uint32_t b, double (&x)[2]) {
                const double ox = s.od[r], oy = s.od[g], oz = s.od[b];
                x[0] = dot3(ox, oy, oz, P[0], P[2], P[4]);  // P read from LDS at the point of use
                x[1] = dot3(ox, oy, oz, P[1], P[3], P[5]);
            },
            [&](uint4* seg, unsigned cap, unsigned& count, unsigned& bl0, unsigned& bl1) {
                const float lo0 = uni((float)s.wlo[0]), hi0 = uni((float)s.whi[0]), lo1 = uni((float)s.wlo[1]), hi1 = uni((float)s.whi[1]);
                auto slack = [](float v) { return fabsf(v) < 3e38f ? 2.4e-7f * fabsf(v) : 0.0f; };
                const float t0 = uni(tol0 + slack(lo0) + slack(hi0)), t1 = uni(tol1 + slack(lo1) + slack(hi1));
                // below <=> (c + t) - lo < 0, above <=> hi - (c - t) < 0 (differences of finite / infinite floats: never NaN here)
#pragma unroll 1
                for (int j = 0; j < n_slots; ++j) {
                    const unsigned valid = tid + RT * j < ng ? 1u : 0u;
                    uint32_t rr[4], gg[4], bb[4];
                    unpack_group(pa[j], pb[j], pc[j], rr, gg, bb);
                    unsigned fl = 0u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float c0, c1;
                        conc32(rr[i], gg[i], bb[i], c0, c1);
                        const unsigned bel0 = sgn((c0 + t0) - lo0), abv0 = sgn(hi0 - (c0 - t0));
                        const unsigned bel1 = sgn((c1 + t1) - lo1), abv1 = sgn(hi1 - (c1 - t1));
                        bl0 += valid & bel0;
                        bl1 += valid & bel1;
                        fl |= ((valid & ((bel0 | abv0) ^ 1u)) | ((valid & ((bel1 | abv1) ^ 1u)) << 1)) << (2 * i);
                    }
                    seg_push(fl != 0u, make_uint4(pa[j], pb[j], pc[j], fl), seg, cap, count);
                }
            },
            s, kp, nn, vp, vn);
        if (!ok) {
            give_up();
            return;
        }
        maxc[0] = np_lerp(vp[0], vn[0], gm[0]);
        maxc[1] = np_lerp(vp[1], vn[1], gm[1]);
    }

    if (tid == 0) {
        const double* S = &s.bc[8];
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
            if (!(isfinite(sc0) && isfinite(sc1))) flags |= TIA_FLAG_DEGENERATE;
            out[TIA_ST_SCALE + 0] = sc0;
            out[TIA_ST_SCALE + 1] = sc1;
            for (int j = 0; j < 3; ++j)
                for (int c = 0; c < 3; ++c)
                    out[TIA_ST_M + j * 3 + c] = P[j * 2 + 0] * sc0 * prm.target_stain[c] +
                                                P[j * 2 + 1] * sc1 * prm.target_stain[3 + c];
        }
        out[TIA_ST_FLAGS] = (double)flags;
#if TIA_STATS_TIMING
        s.tm[TM_TOTAL] = clock64() - t_begin;
        s.tm[TM_CONC_TOTAL] = s.tm[TM_TOTAL] - s.tm[TM_PHI_TOTAL];
        for (int i = 0; i < 16; ++i) out[TIA_ST_CYCLES + i] = (double)s.tm[i];
#endif
    }
}

}  // namespace tia

static size_t align256s(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t tia_stain_stats_workspace_bytes(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)n * (size_t)h * (size_t)w * 2u * sizeof(uint16_t);
}

extern "C" size_t tia_stain_stats_workspace_bytes_mode(int64_t n, int64_t h, int64_t w, int32_t mode) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    // [bin cache][dictionary (Vahadane only)][per-patch redo flags of the register-resident kernel]
    size_t total = align256s(tia_stain_stats_workspace_bytes(n, h, w));
    if (mode == TIA_MODE_VAHADANE) total += align256s((size_t)n * (size_t)h * (size_t)w * sizeof(double2));
    return total + align256s((size_t)n * sizeof(int));
}

// patches of this shape (in whole 4-pixel groups, window selection on) go through the register-resident kernel: every size it
// can hold (16 groups per thread: up to 256 x 256).  Round 3 stopped at 224 x 224 because one patch in eight was handed back at
// 256 x 256 -- the window-placing sample was column-aligned there (sample_index) -- which made the pair slower than the streaming
// kernel alone; with the stratified sample nothing is handed back and both take the same time (2.14 ms per 4096 x 256^2,
// profiles/r04q_*), the register-resident one with a third of the HBM-side traffic.  select_mode 2 keeps the streaming kernel.
static bool stats_reg_shape(long hw, const tia_stain_params* params) {
    static const bool no_reg = getenv("TIA_STATS_NO_REG") != nullptr;  // developer switch (A/B measurements)
    const long reg_limit = (long)tia::RT * tia::RG * 4;
    // (>= 4096 pixels: the sample of 4096 dwords is parked in the patch's 4 hw bytes of workspace)
    return !no_reg && params->mode != TIA_MODE_VAHADANE && params->select_mode == 0 && (hw & 3) == 0 && hw <= reg_limit &&
           hw >= tia::SAMPLE_TARGET;
}

extern "C" int tia_stain_stats_path(int64_t h, int64_t w, const tia_stain_params* params) {
    if (!params || h <= 0 || w <= 0) return TIA_EINVAL;
    return stats_reg_shape((long)h * (long)w, params) ? 1 : 0;
}

extern "C" int tia_stain_stats_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                   const tia_stain_tables* d_tables, const tia_stain_params* params,
                                   double* d_stats, void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_img || !d_tables || !params || !d_stats) return TIA_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0) return TIA_EINVAL;
    if (params->mode < TIA_MODE_MACENKO || params->mode > TIA_MODE_GIVEN) return TIA_EINVAL;
    const long hw = (long)h * (long)w;
    if ((unsigned long long)hw * 3ull >= 0xffffffffull) return TIA_ESIZE;  // 32-bit histogram counts
    if (n > 0x7fffffffll) return TIA_ESIZE;
    const bool aligned = d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 15) == 0;
    // the per-pixel bin cache is optional: without (enough) workspace the kernel recomputes values
    uint16_t* binws = (aligned && ws_bytes >= tia_stain_stats_workspace_bytes(n, h, w)) ? (uint16_t*)d_ws : nullptr;
    double2* dictws = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (params->mode == TIA_MODE_VAHADANE) {  // the dictionary scratch is not optional
        if (!aligned || ws_bytes < tia_stain_stats_workspace_bytes_mode(n, h, w, TIA_MODE_VAHADANE)) return TIA_ESIZE;
        if (params->dl_max_iter < 1 || !(params->dl_alpha >= 0.0)) return TIA_EINVAL;
        dictws = (double2*)((char*)d_ws + align256s(tia_stain_stats_workspace_bytes(n, h, w)));
        static const bool env_one = getenv("TIA_DL_ONE_KERNEL") != nullptr;  // developer switch (A/B measurements)
        if (params->dl_one_kernel || env_one) {
            hipLaunchKernelGGL(tia::stain_stats_kernel<true>, dim3((unsigned)n), dim3(tia::NT), 0, st, d_img, hw, d_tables, *params, d_stats,
                               binws, dictws, (const int*)nullptr);
            return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
        }
        // the kernel pair: dictionary learning by replay (no dictionary traffic), then the common tail in MODE_VTAIL; what the first
        // kernel hands back (an unused atom's re-draw, more iterations than it records) goes through the one-kernel form
        int* redo = (int*)((char*)dictws + align256s((size_t)n * (size_t)hw * sizeof(double2)));
        if (hipMemsetAsync(redo, 0, (size_t)n * sizeof(int), st) != hipSuccess) return TIA_ELAUNCH;
        hipLaunchKernelGGL(tia::vahadane_dl_kernel, dim3((unsigned)n), dim3(tia::NT), 0, st, d_img, hw, d_tables, *params, d_stats, redo);
        tia_stain_params tail = *params;
        tail.mode = tia::MODE_VTAIL;
        hipLaunchKernelGGL(tia::stain_stats_kernel<false>, dim3((unsigned)n), dim3(tia::NT), 0, st, d_img, hw, d_tables, tail, d_stats, binws,
                           (double2*)nullptr, (const int*)redo);
        hipLaunchKernelGGL(tia::stain_stats_kernel<true>, dim3((unsigned)n), dim3(tia::NT), 0, st, d_img, hw, d_tables, *params, d_stats,
                           binws, dictws, (const int*)redo);
        return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
    }
    // Patches of up to 13 x 4096 pixels (224 x 224 and smaller) in whole 4-pixel groups go through the register-resident kernel
    // (the patch is read from HBM once); whatever it hands back -- and every other case -- through the streaming kernel.  Both
    // give the same bits.  Measured on MI355X (profiles/r03d_perf_stain*.txt): both kernels are bound by the vector ALU (about 150
    // instructions per pixel over the four sweeps, ~4 cycles each per wave), so the single read buys HBM traffic (1.25 GB instead of
    // 3.9 GB per 4096 x 224^2), not time; at 256 x 256 one 1024-thread workgroup per CU (16 register slots per thread) is slower than
    // two streaming workgroups that overlap each other's single-lane phases, so those patches stay on the streaming kernel.
    const bool reg_ok = stats_reg_shape(hw, params) &&
                        (reinterpret_cast<uintptr_t>(d_img) & 3) == 0 && aligned &&
                        ws_bytes >= tia_stain_stats_workspace_bytes_mode(n, h, w, params->mode);
    if (reg_ok) {
        int* redo = (int*)((char*)d_ws + align256s(tia_stain_stats_workspace_bytes(n, h, w)));
        static tia::DeviceOnce attr_once;  // the dynamic-LDS attribute is per device
        if (!attr_once.ensure([] {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&tia::stain_stats_reg_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(tia::SmemR)) == hipSuccess;
            }))
            return TIA_ELAUNCH;
        if (hipMemsetAsync(redo, 0, (size_t)n * sizeof(int), st) != hipSuccess) return TIA_ELAUNCH;
        // (the register-resident kernel has no use for the bin cache: 16 KB per patch of it hold the window-placing sample)
        hipLaunchKernelGGL(tia::stain_stats_reg_kernel, dim3((unsigned)n), dim3(tia::RT), sizeof(tia::SmemR), st, d_img, hw, d_tables,
                           *params, d_stats, redo, reinterpret_cast<uint32_t*>(d_ws));
        hipLaunchKernelGGL(tia::stain_stats_kernel<false>, dim3((unsigned)n), dim3(tia::NT), 0, st, d_img, hw, d_tables, *params,
                           d_stats, binws, dictws, (const int*)redo);
    } else {
        hipLaunchKernelGGL(tia::stain_stats_kernel<false>, dim3((unsigned)n), dim3(tia::NT), 0, st, d_img, hw, d_tables, *params,
                           d_stats, binws, dictws, (const int*)nullptr);
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
