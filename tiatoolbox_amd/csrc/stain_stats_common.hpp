// Shared device code of the per-patch stain-statistics kernels (round 5: stain_stats.hip was one 3,600-line file; it is now
//   stain_stats_common.hpp   -- this file: LDS layouts, exact order statistics (histogram selection `select2`, window selection
//                               `window_select2`), numpy-exact percentile helpers, the 3x3 eigen-solver, the two-atom lasso
//   stain_stats_reg.hip      -- PRODUCT default for Macenko / fixed / given modes: `stain_stats_reg_kernel` (patch in registers)
//   stain_stats_stream.hip   -- `stain_stats_kernel<DL>`: <false> = PRODUCT for patches the register kernel cannot hold or hands
//                               back, and the common tail of the Vahadane pair (MODE_VTAIL); <true> = the one-kernel Vahadane form
//                               (fallback for patches the replay kernel hands back; AUDIT form behind `dl_one_kernel`)
//   stain_stats_vahadane.hip -- PRODUCT default for Vahadane: `vahadane_dl_kernel` (dictionary learning by replay)
//   stain_stats.hip          -- the C entry points and the dispatch between the kernels).
// Audit switches that select a non-default form are all run-time (`select_mode`, `dl_one_kernel`, TIA_STATS_NO_REG) and tested for
// bit-identity with the defaults (tests/test_stain_gpu.py); `-DTIA_STATS_TIMING=1` is the phase-stamp build.
#pragma once
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

// ---- launchers (one per translation unit; stain_stats.hip dispatches) -----------------------------------------------------------
// streaming kernel: dl = the one-kernel Vahadane form; redo = nullptr (every patch) or the flag array of a first kernel
int launch_stain_stats_stream(bool dl, const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm,
                              double* d_stats, uint16_t* binws, double2* dictws, const int* redo, hipStream_t st);
int launch_vahadane_dl(const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm, double* d_stats,
                       int* redo, hipStream_t st);
int launch_stain_stats_reg(const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm,
                           double* d_stats, int* redo, uint32_t* ws, hipStream_t st);
long stain_stats_reg_pixel_limit();  // largest patch (pixels) the register-resident kernel holds
// large single images: multi-workgroup sweeps (stain_stats_big.hip); d_ws holds n state blocks
constexpr long kBigImagePixels = 4L * 256 * 256;
size_t stain_stats_big_workspace_bytes(long n, long hw);
int launch_stain_stats_big(const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm,
                           double* d_stats, void* d_ws, hipStream_t st);

}  // namespace tia
