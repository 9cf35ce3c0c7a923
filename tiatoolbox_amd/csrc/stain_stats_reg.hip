// Register-resident per-patch statistics kernel (PRODUCT default for Macenko / fixed / given modes, patches of <= 65,536 pixels):
// one 1024-thread workgroup reads the patch from HBM once and keeps it in registers.  DESIGN.md 4.9 / 4.10.
// Reference: tools/stainextract.py:177-227, tools/stainnorm.py:49-66,81-85,103, utils/misc.py:261-290,405-444.
#include "stain_stats_common.hpp"

namespace tia {

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

long stain_stats_reg_pixel_limit() { return (long)RT * RG * 4; }

int launch_stain_stats_reg(const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm,
                           double* d_stats, int* redo, uint32_t* ws, hipStream_t st) {
    static DeviceOnce attr_once;  // the dynamic-LDS attribute is per device
    if (!attr_once.ensure([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&stain_stats_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(SmemR)) == hipSuccess;
        }))
        return TIA_ELAUNCH;
    hipLaunchKernelGGL(stain_stats_reg_kernel, dim3((unsigned)n), dim3(RT), sizeof(SmemR), st, d_img, hw, d_tables, prm, d_stats, redo, ws);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia
