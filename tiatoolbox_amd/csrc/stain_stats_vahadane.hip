// Vahadane dictionary learning on the device, first kernel of the pair (PRODUCT default for TIA_MODE_VAHADANE): scikit-learn's
// DictionaryLearning as configured at tools/stainextract.py:305-316, with NO dictionary in memory (per-pixel replay).  DESIGN.md 4.2.
#include "stain_stats_common.hpp"

namespace tia {

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


int launch_vahadane_dl(const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm, double* d_stats,
                       int* redo, hipStream_t st) {
    hipLaunchKernelGGL(vahadane_dl_kernel, dim3((unsigned)n), dim3(NT), 0, st, d_img, hw, d_tables, prm, d_stats, redo);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia
