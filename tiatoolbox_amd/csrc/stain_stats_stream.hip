// Streaming per-patch statistics kernel: one 512-thread workgroup per patch (two resident per CU), all per-patch state in LDS, the
// patch re-read from L2 / MALL on every sweep.  Roles (see stain_stats_common.hpp): <false> serves patches the register-resident
// kernel cannot hold or hands back, every `select_mode` audit, and the tail of the Vahadane pair; <true> is the one-kernel
// Vahadane form (dictionary in a workspace).  DESIGN.md 4.1 / 4.2.
//   P1 byte histogram -> contrast-enhancer percentiles -> folded luminance tables
//   P2 tissue mask + OD moments (f64) -> covariance -> 3x3 eigen-decomposition
//   P3/P4 exact angular percentiles on a monotone pseudo-angle key (window selection; histogram refine as fallback)
//   P5/P6 exact 99th percentile of both stain concentrations
// Reference: tools/stainextract.py:177-227,281-322, tools/stainnorm.py:49-66,81-85,103, utils/misc.py:261-290,405-444.
#include "stain_stats_common.hpp"

namespace tia {

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


int launch_stain_stats_stream(bool dl, const uint8_t* d_img, long n, long hw, const tia_stain_tables* d_tables, const tia_stain_params& prm,
                              double* d_stats, uint16_t* binws, double2* dictws, const int* redo, hipStream_t st) {
    if (dl)
        hipLaunchKernelGGL(stain_stats_kernel<true>, dim3((unsigned)n), dim3(NT), 0, st, d_img, hw, d_tables, prm, d_stats, binws, dictws, redo);
    else
        hipLaunchKernelGGL(stain_stats_kernel<false>, dim3((unsigned)n), dim3(NT), 0, st, d_img, hw, d_tables, prm, d_stats, binws, dictws, redo);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

}  // namespace tia
