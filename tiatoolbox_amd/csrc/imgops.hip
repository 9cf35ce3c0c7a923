// Image primitives on gfx950: grey conversion, byte histogram, thresholding, connected-component
// labelling (union-find with atomicMin, raster-ordered relabel), label-area filtering, binary
// morphology with an arbitrary structuring element, hole filling.
// Reference call sites: tools/tissuemask.py:99-164,270-306; models/architecture/hovernet.py:541-545,
// 604-614 (scipy.ndimage.label / binary_fill_holes, skimage remove_small_objects, cv2.morphologyEx).
#include "common.hpp"
#include "wide_io.hpp"

namespace tia {

constexpr int BT = 256;

static inline unsigned nblocks(long n, int per = BT, long cap = 65535L * 16) {
    long b = (n + per - 1) / per;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

// ---- grey / histogram / threshold ----------------------------------------------------------------------
__device__ __forceinline__ uint32_t gray_of(uint32_t r, uint32_t g, uint32_t b) {
    return (r * 9798u + g * 19235u + b * 3735u + (1u << 14)) >> 15;
}

__global__ __launch_bounds__(BT) void rgb2gray_kernel(const uint8_t* __restrict__ img, long npix,
                                                       uint8_t* __restrict__ gray) {
    const long ng = npix >> 2;
    const long stride = (long)gridDim.x * BT;
    const bool fast = ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(gray)) & 3) == 0;
    if (fast) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(img);
        uint32_t* o = reinterpret_cast<uint32_t*>(gray);
        for (long g = (long)blockIdx.x * BT + threadIdx.x; g < ng; g += stride) {
            const uint32_t a = q[g * 3], b = q[g * 3 + 1], c = q[g * 3 + 2];
            const uint32_t g0 = gray_of(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
            const uint32_t g1 = gray_of(a >> 24, b & 255u, (b >> 8) & 255u);
            const uint32_t g2 = gray_of((b >> 16) & 255u, b >> 24, c & 255u);
            const uint32_t g3 = gray_of((c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
            o[g] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        }
        for (long i = ng * 4 + (long)blockIdx.x * BT + threadIdx.x; i < npix; i += stride)
            gray[i] = (uint8_t)gray_of(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
    } else {
        for (long i = (long)blockIdx.x * BT + threadIdx.x; i < npix; i += stride)
            gray[i] = (uint8_t)gray_of(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
    }
}

__global__ __launch_bounds__(BT) void hist256_kernel(const uint8_t* __restrict__ data, long n,
                                                      uint32_t* __restrict__ hist) {
    __shared__ unsigned h[4][256];  // one private copy per wave
    for (int i = threadIdx.x; i < 1024; i += BT) (&h[0][0])[i] = 0;
    __syncthreads();
    unsigned* mine = h[wave_id()];
    const long stride = (long)gridDim.x * BT;
    const long nw = ((reinterpret_cast<uintptr_t>(data) & 3) == 0) ? (n >> 2) : 0;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(data);
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < nw; i += stride) {
        const uint32_t v = q[i];
        atomicAdd(&mine[v & 255u], 1u);
        atomicAdd(&mine[(v >> 8) & 255u], 1u);
        atomicAdd(&mine[(v >> 16) & 255u], 1u);
        atomicAdd(&mine[v >> 24], 1u);
    }
    for (long i = nw * 4 + (long)blockIdx.x * BT + threadIdx.x; i < n; i += stride) atomicAdd(&mine[data[i]], 1u);
    __syncthreads();
    const unsigned t = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (t) atomicAdd(&hist[threadIdx.x], t);
}

__global__ __launch_bounds__(BT) void threshold_lt_kernel(const uint8_t* __restrict__ src, long npix, int is_rgb,
                                                           int thr, const int* __restrict__ thr_dev, uint8_t* __restrict__ mask) {
    if (thr_dev) thr = *thr_dev;
    const long stride = (long)gridDim.x * BT;
    const bool fast = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(mask)) & 3) == 0;
    long done = 0;
    if (fast) {  // 4 pixels per lane: 3 (or 1) dword loads, 1 dword store
        const long ng = npix >> 2;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src);
        uint32_t* o = reinterpret_cast<uint32_t*>(mask);
        for (long g = (long)blockIdx.x * BT + threadIdx.x; g < ng; g += stride) {
            uint32_t g0, g1, g2, g3;
            if (is_rgb) {
                const uint32_t a = q[g * 3], b = q[g * 3 + 1], c = q[g * 3 + 2];
                g0 = gray_of(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
                g1 = gray_of(a >> 24, b & 255u, (b >> 8) & 255u);
                g2 = gray_of((b >> 16) & 255u, b >> 24, c & 255u);
                g3 = gray_of((c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
            } else {
                const uint32_t a = q[g];
                g0 = a & 255u;
                g1 = (a >> 8) & 255u;
                g2 = (a >> 16) & 255u;
                g3 = a >> 24;
            }
            o[g] = ((int)g0 < thr ? 1u : 0u) | ((int)g1 < thr ? 1u << 8 : 0u) | ((int)g2 < thr ? 1u << 16 : 0u) |
                   ((int)g3 < thr ? 1u << 24 : 0u);
        }
        done = ng * 4;
    }
    for (long i = done + (long)blockIdx.x * BT + threadIdx.x; i < npix; i += stride) {
        const int g = is_rgb ? (int)gray_of(src[3 * i], src[3 * i + 1], src[3 * i + 2]) : (int)src[i];
        mask[i] = g < thr ? 1 : 0;
    }
}

// ---- one-pass Otsu fit: grey + histogram fused, 16-byte loads, conflict-bounded LDS counters ---------------------------------------
// Each wave keeps 16 private sub-histograms (lane & 15 picks one; 256 counters of 16 bits, two per dword, interleaved so that
// sub-histogram s of counter pair k sits at dword 16 k + s): a ds_add of the wave touches every LDS bank exactly twice whatever
// the pixel values are -- a smooth image (every lane on the same grey level) costs the same as noise, where one shared histogram
// serialises up to 64 same-address updates.  16-bit counters: a sub-histogram sees 64 pixels per wave step, so a wave may run
// kGrayHistMaxSteps steps (the host sizes the grid accordingly).
constexpr int kGrayHistMaxSteps = 1000;

__device__ __forceinline__ void sub_hist_add(unsigned* sub, uint32_t bin) {
    // counter `bin` of this lane's sub-histogram: dword (bin >> 1) * 16, half (bin & 1)
    atomicAdd(&sub[(bin >> 1) * 16], 1u << ((bin & 1u) * 16u));
}

__device__ __forceinline__ void otsu_from_counts(unsigned long long c, int* __restrict__ out);  // (below)

// `fit_out` (nullable): the workgroup that finishes LAST (a ticket counter in hist[256]) also runs Otsu's arithmetic on the complete
// counts -- OtsuTissueMasker.fit is one launch; the counter is reset, so that a later call that adds another image to the same
// counts recomputes the threshold.
template <int CH>  // 3: RGB pixels, grey conversion fused; 1: a grey plane
__global__ __launch_bounds__(BT) void gray_hist_kernel(const uint8_t* __restrict__ img, long npix, uint32_t* __restrict__ hist,
                                                       int* __restrict__ fit_out) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[BT / 64][kRgbChunk];
    __shared__ unsigned sub[BT / 64][128 * 16];
    for (int i = threadIdx.x; i < (BT / 64) * 128 * 16; i += BT) (&sub[0][0])[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned* mine = &sub[wv][lane & 15];
    const long waves = (long)gridDim.x * (BT / 64), wave = (long)blockIdx.x * (BT / 64) + wv;
    long done = 0;  // pixels covered by the wide path
    if constexpr (CH == 3) {
        if ((reinterpret_cast<uintptr_t>(img) & 15) == 0) {
            const long nchunks = npix / kPxChunk;
            RgbChunk cur, nxt;
            long c = wave;
            if (c < nchunks) rgb_chunk_issue(cur, img + c * kRgbChunk);
            for (; c < nchunks; c += waves) {
                const long cn = c + waves;
                if (cn < nchunks) rgb_chunk_issue(nxt, img + cn * kRgbChunk);
                uint32_t w[12];
                rgb_chunk_transpose(cur, stage[wv], w);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t p[4];
                    group_pixels(w[3 * q], w[3 * q + 1], w[3 * q + 2], p);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sub_hist_add(mine, gray_of_px(p[i]));
                }
                cur = nxt;
            }
            done = nchunks * kPxChunk;
        }
    } else {
        if ((reinterpret_cast<uintptr_t>(img) & 15) == 0) {
            const long nv = npix >> 4;  // 16 bytes per lane and step
            const v4u* q = reinterpret_cast<const v4u*>(img);
            for (long i = wave * 64 + lane; i < nv; i += waves * 64) {
                const v4u t = __builtin_nontemporal_load(q + i);
                const uint32_t d[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sub_hist_add(mine, d[k] & 255u);
                    sub_hist_add(mine, (d[k] >> 8) & 255u);
                    sub_hist_add(mine, (d[k] >> 16) & 255u);
                    sub_hist_add(mine, d[k] >> 24);
                }
            }
            done = nv << 4;
        }
    }
    for (long i = done + wave * 64 + lane; i < npix; i += waves * 64) {
        const uint32_t g = CH == 3 ? gray_of((uint32_t)img[3 * i], (uint32_t)img[3 * i + 1], (uint32_t)img[3 * i + 2]) : (uint32_t)img[i];
        sub_hist_add(mine, g);
    }
    __syncthreads();
    const int b = threadIdx.x;  // BT == 256 bins
    unsigned t = 0;
#pragma unroll
    for (int wq = 0; wq < BT / 64; ++wq)
#pragma unroll
        for (int s = 0; s < 16; ++s) t += (sub[wq][(b >> 1) * 16 + s] >> ((b & 1) * 16)) & 0xffffu;
    if (t) atomicAdd(&hist[b], t);
    if (fit_out == nullptr) return;
    __shared__ unsigned ticket;
    // The counts travel as device-scope atomics (performed at the memory side, coherent across the XCDs' L2s): waiting for their
    // acknowledgements (vmcnt) orders them before the ticket -- no release fence, which on gfx950 writes the whole L2 back (measured:
    // the fit twice as slow with a __threadfence() per workgroup).
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(&hist[256], 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    const unsigned long long c = __hip_atomic_load(&hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (L2, not this CU's L1)
    if (threadIdx.x == 0) hist[256] = 0u;
    otsu_from_counts(c, fit_out);
}

// skimage.filters.threshold_otsu on a 256-bin byte histogram, on the device (tools/tissuemask.py:131-134; the arithmetic of
// scikit-image's `threshold_otsu(hist=...)` restricted to the occupied range [lo, hi], as the reference's image path bins it):
// weight1 = cumsum(counts), weight2 = reversed cumsum, mean1 = cumsum(counts * centers) / weight1, mean2 likewise from the top,
// variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:])**2, threshold = centers[argmax] (first maximum).  The counts and
// their products with the integer bin centres are integers far below 2^53, so every partial sum is exact in float64 whatever the
// order: integer prefix sums here equal NumPy's sequential float64 cumsum bit for bit; the divisions and products are the same
// IEEE operations in the same order.  out[0] = threshold (the only occupied bin when there is just one), out[1] = occupied bins.
// (device function: runs in one 256-thread workgroup -- the stand-alone kernel below, or the last workgroup of the fit kernel)
__device__ __forceinline__ void otsu_from_counts(unsigned long long c, int* __restrict__ out) {
    __shared__ unsigned long long w1[256], s1[256];
    __shared__ unsigned long long wkey[4];
    __shared__ int widx[4];
    __shared__ int lohi[2], nz;
    const int i = threadIdx.x, lane = i & 63, wv = i >> 6;
    if (i == 0) {
        lohi[0] = 256;
        lohi[1] = -1;
        nz = 0;
    }
    __syncthreads();
    if (c) {
        atomicMin(&lohi[0], i);
        atomicMax(&lohi[1], i);
        atomicAdd(&nz, 1);
    }
    w1[i] = c;
    s1[i] = c * (unsigned long long)i;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive scans (Hillis-Steele; integer: exact)
        const unsigned long long a = i >= o ? w1[i - o] : 0ull, b = i >= o ? s1[i - o] : 0ull;
        __syncthreads();
        w1[i] += a;
        s1[i] += b;
        __syncthreads();
    }
    const int lo = lohi[0], hi = lohi[1], nzc = nz;
    // first maximum of variance12 over [lo, hi): order-preserving key of the (non-negative) float64, ties to the smaller index
    unsigned long long key = 0ull;  // below every key of a value >= 0 (f64_key(0.0) = 0x8000...)
    int idx = i;
    if (nzc > 1 && i >= lo && i < hi) {
        const double wt1 = (double)w1[i], wt2 = (double)(w1[255] - w1[i]);
        const double m1 = (double)s1[i] / wt1, m2 = (double)(s1[255] - s1[i]) / wt2;
        const double d = m1 - m2;
        key = f64_key(wt1 * wt2 * (d * d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long k2 = __shfl_down(key, o, 64);
        const int i2 = __shfl_down(idx, o, 64);
        if (k2 > key || (k2 == key && i2 < idx)) {
            key = k2;
            idx = i2;
        }
    }
    if (lane == 0) {
        wkey[wv] = key;
        widx[wv] = idx;
    }
    __syncthreads();
    if (i == 0) {
        int best = lo;
        if (nzc > 1) {
            unsigned long long bk = wkey[0];
            best = widx[0];
            for (int q = 1; q < 4; ++q)
                if (wkey[q] > bk) {  // waves hold ascending index ranges: strict > keeps the first maximum
                    bk = wkey[q];
                    best = widx[q];
                }
        }
        out[0] = nzc == 0 ? 0 : best;
        out[1] = nzc;
    }
}

__global__ __launch_bounds__(256) void otsu_threshold_kernel(const uint32_t* __restrict__ hist, int* __restrict__ out) {
    otsu_from_counts((unsigned long long)hist[threadIdx.x], out);
}

// mask = grey < thr, 16-byte accesses: 48 bytes of RGB in, 16 mask bytes out per lane and step (a lane's 16 pixels are contiguous
// in the mask).  thr_dev (nullable): the threshold comes from device memory (tia_otsu_threshold_u32's output).
__global__ __launch_bounds__(BT) void threshold_wide_kernel(const uint8_t* __restrict__ src, long npix, int thr,
                                                             const int* __restrict__ thr_dev, uint8_t* __restrict__ mask) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[BT / 64][kRgbChunk];
    if (thr_dev) thr = *thr_dev;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long waves = (long)gridDim.x * (BT / 64), wave = (long)blockIdx.x * (BT / 64) + wv;
    const long nchunks = npix / kPxChunk;
    RgbChunk cur, nxt;
    long c = wave;
    if (c < nchunks) rgb_chunk_issue(cur, src + c * kRgbChunk);
    for (; c < nchunks; c += waves) {
        const long cn = c + waves;
        if (cn < nchunks) rgb_chunk_issue(nxt, src + cn * kRgbChunk);
        uint32_t w[12];
        rgb_chunk_transpose(cur, stage[wv], w);
        v4u m;
        uint32_t* mo = reinterpret_cast<uint32_t*>(&m);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t p[4];
            group_pixels(w[3 * q], w[3 * q + 1], w[3 * q + 2], p);
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) bits |= ((int)gray_of_px(p[i]) < thr ? 1u : 0u) << (8 * i);
            mo[q] = bits;
        }
        __builtin_nontemporal_store(m, reinterpret_cast<v4u*>(mask + c * kPxChunk) + lane);
        cur = nxt;
    }
    for (long i = nchunks * kPxChunk + wave * 64 + lane; i < npix; i += waves * 64)
        mask[i] = (int)gray_of((uint32_t)src[3 * i], (uint32_t)src[3 * i + 1], (uint32_t)src[3 * i + 2]) < thr ? 1 : 0;
}

// ---- connected components ------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int* __restrict__ L, int i) {
    int p = L[i];
    while (p != i) {
        i = p;
        p = L[i];
    }
    return i;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    while (true) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[a], b);  // hook the larger root under the smaller index
        if (old == a) return;
        a = old;
    }
}

// Initial forest: every foreground pixel points at the first pixel of its horizontal run, cut at row starts
// and at 64-pixel (wave) boundaries -- one ballot per wave, no memory traffic.  The merge pass then only
// has to (a) re-join runs cut at a wave boundary and (b) hook vertically / diagonally adjacent runs once
// per overlap, instead of issuing a union per pixel and neighbour.
// INVERT: label the background (zero pixels) instead (used by fill-holes)
template <bool INVERT>
__global__ __launch_bounds__(BT) void ccl_init_kernel(const uint8_t* __restrict__ mask, long hw, int w, int* __restrict__ L) {
    const uint8_t* m = mask + (size_t)blockIdx.y * hw;
    int* l = L + (size_t)blockIdx.y * hw;
    const int lane = lane_id();
    const long stride = (long)gridDim.x * BT;
    const long rounds = (hw + stride - 1) / stride;  // same trip count for every lane: the ballots need whole waves
    for (long r = 0; r < rounds; ++r) {
        const long i = r * stride + (long)blockIdx.x * BT + threadIdx.x;
        const bool inb = i < hw;
        const bool fg = inb && (INVERT ? (m[i] == 0) : (m[i] != 0));
        const int x = inb ? (int)(i % w) : 0;
        const bool prev_fg = __shfl_up((int)fg, 1) != 0;
        const bool start = fg && (lane == 0 || x == 0 || !prev_fg);
        const unsigned long long starts = __ballot(start);
        if (!inb) continue;
        if (fg) {
            const unsigned long long below = starts & ((2ull << lane) - 1ull);  // a start exists at or below a fg lane
            l[i] = (int)(i - lane + (63 - __builtin_clzll(below)));
        } else {
            l[i] = -1;
        }
    }
}

__global__ __launch_bounds__(BT) void ccl_merge_kernel(int* __restrict__ L, int h, int w, int conn8) {
    const long hw = (long)h * w;
    int* l = L + (size_t)blockIdx.y * hw;
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < hw; i += (long)gridDim.x * BT) {
        if (l[i] < 0) continue;
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        const bool left = x > 0 && l[i - 1] >= 0;
        if (left && (i & 63) == 0) uf_union(l, (int)i, (int)i - 1);  // run cut at a wave boundary by the init pass
        if (y == 0) continue;
        const bool up = l[i - w] >= 0;
        const bool upleft = x > 0 && l[i - w - 1] >= 0;
        if (up) {
            // first column of the overlap between this run and the run above
            if (!left || !upleft) uf_union(l, (int)i, (int)(i - w));
        } else if (conn8) {
            if (upleft && !left) uf_union(l, (int)i, (int)(i - w - 1));
            if (x < w - 1 && l[i - w + 1] >= 0 && !(l[i + 1] >= 0)) uf_union(l, (int)i, (int)(i - w + 1));
        }
    }
}

__global__ __launch_bounds__(BT) void ccl_flatten_kernel(int* __restrict__ L, long hw) {
    int* l = L + (size_t)blockIdx.y * hw;
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < hw; i += (long)gridDim.x * BT)
        if (l[i] >= 0) l[i] = uf_find(l, (int)i);
}

// One workgroup per plane: rank the roots (pixels with L[i]==i) in raster order.
// rank[root] = 1-based component number; count[plane] = number of components.
// The plane is swept in coalesced 8192-pixel tiles (8 consecutive pixels per lane) with a running
// total; wave scan + per-wave totals in LDS (double-buffered: one barrier per tile).
__global__ __launch_bounds__(1024) void ccl_rank_kernel(const int* __restrict__ L, long hw, int* __restrict__ rank,
                                                         int* __restrict__ count) {
    __shared__ unsigned wtot[2][16];
    const int* l = L + (size_t)blockIdx.x * hw;
    int* r = rank + (size_t)blockIdx.x * hw;
    const bool vec = (hw & 3) == 0;  // plane base and every tile offset are then 16-byte aligned
    const int t = threadIdx.x, wv = wave_id();
    unsigned running = 0;
    int buf = 0;
    for (long base = 0; base < hw; base += 8192, buf ^= 1) {
        const long i0 = base + 8L * t;
        int v[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        if (vec) {
            if (i0 < hw) {
                const int4 q = *reinterpret_cast<const int4*>(l + i0);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            }
            if (i0 + 4 < hw) {
                const int4 q = *reinterpret_cast<const int4*>(l + i0 + 4);
                v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + k < hw) v[k] = l[i0 + k];
        }
        unsigned f[8], c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f[k] = (v[k] == (int)(i0 + k)) ? 1u : 0u;
            c += f[k];
        }
        const unsigned incl = wave_incl_scan_u32(c);
        if (lane_id() == 63) wtot[buf][wv] = incl;
        __syncthreads();
        unsigned before = running + incl - c, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned x = wtot[buf][q];
            before += q < wv ? x : 0u;
            tot += x;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (f[k]) r[i0 + k] = (int)(++before);
        running += tot;
    }
    if (t == 0) count[blockIdx.x] = (int)running;
}

__global__ __launch_bounds__(BT) void ccl_apply_rank_kernel(const int* __restrict__ L, const int* __restrict__ rank,
                                                             long hw, int* __restrict__ labels) {
    const size_t off = (size_t)blockIdx.y * hw;
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < hw; i += (long)gridDim.x * BT) {
        const int root = L[off + i];
        labels[off + i] = root >= 0 ? rank[off + root] : 0;
    }
}

static int ccl_run(const uint8_t* d_mask, long n, int h, int w, int conn, int* d_labels, int* d_count, int* d_ws,
                   bool invert, hipStream_t st) {
    const long hw = (long)h * w;
    dim3 grid(nblocks(hw, BT, 4096), (unsigned)n);
    // d_labels doubles as the union-find array; d_ws holds the ranks
    if (invert)
        hipLaunchKernelGGL(ccl_init_kernel<true>, grid, dim3(BT), 0, st, d_mask, hw, w, d_labels);
    else
        hipLaunchKernelGGL(ccl_init_kernel<false>, grid, dim3(BT), 0, st, d_mask, hw, w, d_labels);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(BT), 0, st, d_labels, h, w, conn == 8 ? 1 : 0);
    hipLaunchKernelGGL(ccl_flatten_kernel, grid, dim3(BT), 0, st, d_labels, hw);
    hipLaunchKernelGGL(ccl_rank_kernel, dim3((unsigned)n), dim3(1024), 0, st, d_labels, hw, d_ws, d_count);
    // in-place: labels[i] = rank[root]; safe because rank lives in d_ws and roots are read before written
    // only within the same element (labels[i] is overwritten after reading L[i] == labels[i]); other threads
    // still need L[root] == root semantics?  No: apply reads L[i] only (already flattened).
    hipLaunchKernelGGL(ccl_apply_rank_kernel, grid, dim3(BT), 0, st, d_labels, d_ws, hw, d_labels);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

// ---- small planes: the whole labelling in ONE launch, union-find resident in LDS ----------------------------------------
// One 1024-thread workgroup per plane of at most 36,864 pixels (HoVer-Net's 164 x 164 head maps): forest initialisation (runs cut
// at wave boundaries, as above), merge, flatten, raster-order ranking of the roots and -- optionally -- the area filter
// (skimage.morphology.remove_small_objects) without leaving the CU.  Same labels as ccl_run + tia_label_area_filter_i32:
// a component's root is its raster-first pixel, ranks follow the roots' raster order (= skimage.measure.label's numbering).
// After flattening, a root's slot is re-used as its record: 0x80000000 | area << 15 | rank (rank < 2^15: at most hw / 2
// components; area < 2^16), background stays 0xffffffff, every other foreground pixel holds the index of its root.
//   SRC 0: foreground = mask byte != 0;  1: foreground = mask byte == 0 (background labelling);  2: float32 map >= 0.5
__device__ __forceinline__ int lds_find(const int* L, int i) {
    int p = L[i];
    while (p != i) {
        i = p;
        p = L[i];
    }
    return i;
}
__device__ __forceinline__ void lds_union(int* L, int a, int b) {
    while (true) {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}
// fg(i): is pixel i (row-major in the h x w plane) foreground?  Called once per pixel, by whole waves.
template <class FG>
__device__ __forceinline__ void tile_forest_fn(int h, int w, int conn8, int* L, FG fg_of) {
    const int hw = h * w, tid = threadIdx.x, lane = tid & 63;
#pragma unroll 4  // one workgroup per CU: several rounds of source loads in flight
    for (int base = 0; base < hw; base += 1024) {  // same trip count for every lane: the ballots need whole waves
        const int i = base + tid;
        const bool inb = i < hw;
        const bool fg = inb && fg_of(i);
        const int x = inb ? i % w : 0;
        const bool prev_fg = __shfl_up((int)fg, 1) != 0;
        const bool start = fg && (lane == 0 || x == 0 || !prev_fg);
        const unsigned long long starts = __ballot(start);
        if (!inb) continue;
        if (fg) {
            const unsigned long long below = starts & ((2ull << lane) - 1ull);
            L[i] = i - lane + (63 - __builtin_clzll(below));
        } else {
            L[i] = -1;
        }
    }
    __syncthreads();
    for (int i = tid; i < hw; i += 1024) {
        if (L[i] < 0) continue;
        const int y = i / w, x = i - y * w;
        const bool left = x > 0 && L[i - 1] >= 0;
        if (left && (i & 63) == 0) lds_union(L, i, i - 1);  // run cut at a wave boundary by the initialisation
        if (y == 0) continue;
        const bool up = L[i - w] >= 0;
        const bool upleft = x > 0 && L[i - w - 1] >= 0;
        if (up) {
            if (!left || !upleft) lds_union(L, i, i - w);  // first column of the overlap with the run above
        } else if (conn8) {
            if (upleft && !left) lds_union(L, i, i - w - 1);
            if (x < w - 1 && L[i - w + 1] >= 0 && !(L[i + 1] >= 0)) lds_union(L, i, i - w + 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < hw; i += 1024)
        if (L[i] >= 0) L[i] = lds_find(L, i);
    __syncthreads();
}
template <int SRC>
__device__ __forceinline__ void tile_forest(const void* __restrict__ src, long plane_off, int h, int w, int conn8, int* L) {
    tile_forest_fn(h, w, conn8, L, [&](int i) -> bool {
        if constexpr (SRC == 2) return static_cast<const float*>(src)[plane_off + i] >= 0.5f;
        else if constexpr (SRC == 1) return static_cast<const uint8_t*>(src)[plane_off + i] == 0;
        else return static_cast<const uint8_t*>(src)[plane_off + i] != 0;
    });
}

// second half of the small-plane labelling (after tile_forest): rank the roots in raster order, component areas, area filter,
// labels / count / areas to global memory
template <typename Emit>
__device__ __forceinline__ unsigned tile_rank_filter(int* L, int hw, int min_keep, int* __restrict__ count_slot, int* __restrict__ a,
                                                     Emit emit) {
    __shared__ unsigned wtot[16];
    __shared__ unsigned s_running;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // rank the roots in raster order (4 consecutive pixels per lane and round), leave each root's record in its slot
    if (tid == 0) s_running = 0;
    __syncthreads();
    for (int base = 0; base < hw; base += 4096) {
        const int i0 = base + 4 * tid;
        unsigned f[4], c = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[k] = (i0 + k < hw && L[i0 + k] == i0 + k) ? 1u : 0u;
            c += f[k];
        }
        const unsigned incl = wave_incl_scan_u32(c);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        unsigned before = s_running + incl - c, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned x = wtot[q];
            before += q < wv ? x : 0u;
            tot += x;
        }
        __syncthreads();
        if (tid == 0) s_running += tot;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (f[k]) L[i0 + k] = (int)(0x80000000u | ++before);
        __syncthreads();
    }
    if (tid == 0) *count_slot = (int)s_running;
    // areas: one LDS add per foreground pixel on its root's record (the root counts itself)
    for (int i = tid; i < hw; i += 1024) {
        const int v = L[i];
        if (v == -1) continue;
        atomicAdd(reinterpret_cast<unsigned*>(&L[v < 0 ? i : v]), 1u << 15);
    }
    __syncthreads();
    if (a && tid == 0) a[0] = 0;
    for (int base = 0; base < hw; base += 1024) {  // whole waves (emit may ballot)
        const int i = base + tid;
        const bool valid = i < hw;
        const int v = valid ? L[i] : -1;
        int lab = 0;
        if (v != -1) {
            const unsigned rec = (unsigned)(v < 0 ? v : L[v]);
            const int rank = (int)(rec & 0x7fffu), area = (int)((rec >> 15) & 0xffffu);
            if (v < 0 && a) a[rank] = area;  // the root publishes its component's area
            lab = area >= min_keep ? rank : 0;
        }
        emit(i, valid, lab);
    }
    return s_running;
}

// offs / bbox (both nullable, HoVer-Net's blob stage): per-label heap-segment offsets (exclusive scan of the surviving components'
// areas = ws_offsets_kernel) and the reset of the labels' bounding boxes, so that the watershed needs no launch of its own for them
template <int SRC>
__global__ __launch_bounds__(1024) void ccl_tile_kernel(const void* __restrict__ src, int h, int w, int conn8, int min_keep,
                                                         int* __restrict__ labels, int* __restrict__ count, int* __restrict__ areas,
                                                         int* __restrict__ offs, int* __restrict__ bbox) {
    extern __shared__ int L[];
    __shared__ unsigned otot[16];
    const int hw = h * w, tid = threadIdx.x;
    const long plane_off = (long)blockIdx.x * hw;
    tile_forest<SRC>(src, plane_off, h, w, conn8, L);
    int* out = labels + plane_off;
    int* a = areas ? areas + (size_t)blockIdx.x * (hw + 1) : nullptr;
    const unsigned ncomp = tile_rank_filter(L, hw, min_keep, count + blockIdx.x, a, [&](int i, bool valid, int lab) {
        if (valid) out[i] = lab;
    });
    if (offs == nullptr || a == nullptr) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (not __threadfence(): agent scope writes the whole L2 back)
    __syncthreads();  // the areas (written by the roots' lanes) are complete
    const int* av = a;  // written by this workgroup before the fence + barrier; never read by this CU before: plain loads are coherent
    int* o = offs + (size_t)blockIdx.x * (hw + 1);
    const int k = (int)ncomp + 1;  // labels 0 .. ncomp
    const int chunk = (k + 1023) / 1024;
    const int lo = tid * chunk, hi = lo + chunk < k ? lo + chunk : k;
    unsigned c = 0;
    for (int i = lo; i < hi; ++i) {
        const int ar = av[i];
        c += (i > 0 && ar >= min_keep) ? (unsigned)ar : 0u;
    }
    const unsigned incl = wave_incl_scan_u32(c);
    if ((tid & 63) == 63) otot[tid >> 6] = incl;
    __syncthreads();
    unsigned before = incl - c;
    for (int wv = 0; wv < (tid >> 6); ++wv) before += otot[wv];
    for (int i = lo; i < hi; ++i) {
        o[i] = (int)before;
        const int ar = av[i];
        before += (i > 0 && ar >= min_keep) ? (unsigned)ar : 0u;
    }
    if (bbox) {
        int4* bb = reinterpret_cast<int4*>(bbox + (size_t)blockIdx.x * (hw + 1) * 4);
        for (int l = tid; l < k; l += 1024) bb[l] = make_int4(0x7fffffff, -1, 0x7fffffff, -1);
    }
}

// HoVer-Net's marker pipeline on a small plane in ONE launch (hovernet.py:604-614): binary_fill_holes -> 5x5 elliptical opening
// (cv2.morphologyEx(MORPH_OPEN): erode, then dilate; outside the image the erosion sees 1, the dilation 0) -> label (4-connectivity)
// -> remove_small_objects.  LDS: the union-find array (re-used as the second byte plane during the opening) + one byte plane.
// blob / inst / bbox (nullable together): also the watershed's initial state (= ws_init_kernel, hover_post.hip): inst = marker
// label inside a blob, -1 for unlabelled blob pixels, 0 outside; bounding boxes of the blobs (one set of atomics per run).
__global__ __launch_bounds__(1024) void marker_tile_kernel(const uint8_t* __restrict__ marker0, int h, int w, int min_keep,
                                                           int* __restrict__ labels, int* __restrict__ count, int* __restrict__ areas,
                                                           const int* __restrict__ blob, int* __restrict__ inst, int* __restrict__ bbox) {
    extern __shared__ int L[];
    const int hw = h * w, tid = threadIdx.x;
    const long plane_off = (long)blockIdx.x * hw;
    uint8_t* A = reinterpret_cast<uint8_t*>(L + hw);
    uint8_t* B = reinterpret_cast<uint8_t*>(L);
    // fill holes: background components that do not reach the frame
    tile_forest<1>(marker0, plane_off, h, w, 0, L);
    const int nb = 2 * w + 2 * h;
    for (int k = tid; k < nb; k += 1024) {
        int i;
        if (k < w) i = k;
        else if (k < 2 * w) i = (h - 1) * w + (k - w);
        else if (k < 2 * w + h) i = (k - 2 * w) * w;
        else i = (k - 2 * w - h) * w + (w - 1);
        const int v = L[i];
        if (v >= 0) atomicOr(&L[v & 0x3fffffff], 0x40000000);
    }
    __syncthreads();
    for (int i = tid; i < hw; i += 1024) {
        const int v = L[i];
        const bool hole = v >= 0 && (L[v & 0x3fffffff] & 0x40000000) == 0;
        A[i] = (marker0[plane_off + i] != 0 || hole) ? 1 : 0;
    }
    __syncthreads();
    // 5x5 ellipse: rows 00100 / 11111 / 11111 / 11111 / 00100 (cv2.getStructuringElement(MORPH_ELLIPSE, (5, 5)))
    auto morph = [&](const uint8_t* src, uint8_t* dst, bool erode) {
        for (int i = tid; i < hw; i += 1024) {
            const int y = i / w, x = i - y * w;
            bool all = true, any = false;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                const int r = (dy == -2 || dy == 2) ? 0 : 2;
#pragma unroll
                for (int dx = -2; dx <= 2; ++dx) {
                    if (dx < -r || dx > r) continue;
                    const int yy = y + dy, xx = x + dx;
                    const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
                    const bool v = inside ? src[yy * w + xx] != 0 : erode;
                    all = all && v;
                    any = any || v;
                }
            }
            dst[i] = (erode ? all : any) ? 1 : 0;
        }
        __syncthreads();
    };
    morph(A, B, true);
    morph(B, A, false);
    tile_forest<0>(A, 0, h, w, 0, L);
    int* out = labels + plane_off;
    int* bb = bbox ? bbox + (size_t)blockIdx.x * (hw + 1) * 4 : nullptr;
    const int lane = tid & 63;
    tile_rank_filter(L, hw, min_keep, count + blockIdx.x, areas ? areas + (size_t)blockIdx.x * (hw + 1) : nullptr,
                     [&](int i, bool valid, int lab) {
                         if (valid) out[i] = lab;
                         if (blob == nullptr) return;
                         const int b = valid ? blob[plane_off + i] : 0;
                         if (valid) inst[plane_off + i] = b > 0 ? (lab > 0 ? lab : -1) : 0;
                         const int y = i / w, x = i - y * w;
                         const int pb = __shfl_up(b, 1);
                         const bool head = lane == 0 || b != pb || x == 0;
                         const unsigned long long heads = __ballot(head);
                         if (head && b > 0) {
                             const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
                             const int len = (above ? __builtin_ctzll(above) : 64) - lane;
                             atomicMin(&bb[b * 4 + 0], y);
                             atomicMax(&bb[b * 4 + 1], y);
                             atomicMin(&bb[b * 4 + 2], x);
                             atomicMax(&bb[b * 4 + 3], x + len - 1);
                         }
                     });
}

// ---- MorphologicalMasker.transform in ONE launch (tools/tissuemask.py:270-306) ---------------------------------------------------------
// mask = dilate(remove_small_objects(grey < thr, min_size K, connectivity 8), element).  Both steps are LOCAL when K is small: a
// component with fewer than K pixels lies inside the (2K - 1)^2 window around any of its pixels.  Each workgroup takes a core
// rectangle of the image plus a halo of H = R + K - 1 pixels (R = the element's reach) and decides every component of that
// extended tile exactly:
//   * it does not touch the tile's inner frame (frame sides on the image border do not count) -> it is complete, its area is exact;
//   * it touches the inner frame -> for any of its pixels within R of the core, the 8-connected path to the frame has at least K
//     pixels, so the component has >= K pixels: kept.  (Pixels further out may be judged wrongly; nothing reads them.)
// Most of the work is done on BIT ROWS (the tile's 256 x 128 foreground bits are 1024 words in LDS, one per thread):
//   1. threshold: 12-byte pixel loads (a wave reads 768 contiguous bytes of an image row), four foreground bits per lane;
//   2. certificates: a pixel that starts a full bw x bh rectangle (bw bh >= K) belongs to a component of >= K pixels -- an
//      erosion of the bit rows (shifts and ANDs);
//   3. those seeds flood their components: big |= dilate3x3(big) & foreground, a few dozen shift/OR rounds until nothing changes;
//   4. what is left (specks, thin lines: a few per cent of the foreground) goes through the LDS union-find of the small-plane
//      kernels, components below K pixels that are complete in the tile are dropped (if the flood did not converge within its
//      round limit -- a maze -- the whole foreground takes this route: slower, same result);
//   5. dilation on the bit rows (one funnel shift per element offset), then the core's mask bytes.
// One read of the image (x the halo overlap), one write of the mask; label / area / dilation planes never exist in HBM.
constexpr int kMorphTileW = 256, kMorphTileH = 128;  // extended tile: 32,768 pixels (128 KB of LDS for the union-find)
constexpr int kMorphMaxHalo = 40;                    // beyond that the core is under 40 % of the tile: multi-launch form
constexpr int kMorphWords = kMorphTileW * kMorphTileH / 32;
constexpr int kMorphFloodRounds = 96;

// word (r, wx) of a bit plane shifted so that bit x holds the plane's bit x + dx (|dx| < 32); rows / words outside the tile read 0
__device__ __forceinline__ unsigned bit_row_shift(const unsigned* plane, int r, int wx, int dx) {
    if (r < 0 || r >= kMorphTileH) return 0u;
    const unsigned* row = plane + r * 8;
    const unsigned mid = row[wx];
    if (dx == 0) return mid;
    if (dx > 0) {
        const unsigned right = wx < 7 ? row[wx + 1] : 0u;
        return (mid >> dx) | (right << (32 - dx));
    }
    const unsigned left = wx > 0 ? row[wx - 1] : 0u;
    return (mid << -dx) | (left >> (32 + dx));
}

template <int CH>  // 3: RGB, grey conversion fused; 1: grey plane
__global__ __launch_bounds__(1024) void morph_mask_tile_kernel(const uint8_t* __restrict__ img, int h, int w, int thr,
                                                               const int* __restrict__ thr_dev, int min_keep, int halo, int bw, int bh,
                                                               const int* __restrict__ offs, int n_off, int tiles_x, int tiles_y,
                                                               long total_bytes, int force_uf, uint8_t* __restrict__ mask) {
    extern __shared__ int L[];
    __shared__ unsigned fgb[kMorphWords], big[2][kMorphWords], keepb[kMorphWords];
    __shared__ int s_off[512];  // (dy, dx) of the element, first 256 entries
    if (thr_dev) thr = *thr_dev;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cw = kMorphTileW - 2 * halo, chh = kMorphTileH - 2 * halo;  // core
    const int tile = blockIdx.x % (tiles_x * tiles_y), plane = blockIdx.x / (tiles_x * tiles_y);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * chh - halo, x0 = tx * cw - halo;  // image coordinates of the extended tile's corner
    const size_t plane_off = (size_t)plane * (size_t)h * w;
    const uint8_t* src = img + plane_off * CH;
    constexpr int EW = kMorphTileW, EH = kMorphTileH;
    const bool base_aligned = (reinterpret_cast<uintptr_t>(img) & 3) == 0;
    for (int k = tid; k < 2 * n_off && k < 512; k += 1024) s_off[k] = offs[k];
    fgb[tid] = 0u;
    keepb[tid] = 0u;
    __syncthreads();
    // ---- 1. threshold -> foreground bit rows.  Wave wv takes rows wv, wv + 16, ...; lane l the pixels 4l .. 4l + 3 of the row.
#pragma unroll  // all eight rows of a wave in flight: the HBM latency is paid once per tile
    for (int rr = 0; rr < EH / 16; ++rr) {
        const int r = wv + 16 * rr;
        const int y = y0 + r, x = x0 + 4 * lane;
        unsigned bits = 0u;
        if (y >= 0 && y < h && x + 3 >= 0 && x < w) {
            const size_t pix = (size_t)y * w + x;  // (may be "negative" by up to 3 at the left edge: only used when x >= 0)
            const long byte0 = (long)(plane_off + pix) * CH;
            if (CH == 3 && x >= 0 && x + 3 < w && base_aligned && ((byte0 & ~3L) + 16 <= total_bytes)) {
                // the group's 12 bytes from the enclosing aligned dwords (the row start is not 4-byte aligned in general)
                const uint32_t* q = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(src + pix * 3) & ~(uintptr_t)3);
                const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(src + pix * 3) & 3u);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
                uint32_t p[4];
                group_pixels(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                             __builtin_amdgcn_alignbyte(d3, d2, sh), p);
#pragma unroll
                for (int k = 0; k < 4; ++k) bits |= ((int)gray_of_px(p[k]) < thr ? 1u : 0u) << k;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (x + k < 0 || x + k >= w) continue;
                    const uint8_t* p = src + (pix + k) * CH;
                    const int g = CH == 3 ? (int)gray_of((uint32_t)p[0], (uint32_t)p[1], (uint32_t)p[2]) : (int)p[0];
                    bits |= (g < thr ? 1u : 0u) << k;
                }
            }
        }
        if (bits) atomicOr(&fgb[r * 8 + (lane >> 3)], bits << (4 * (lane & 7)));
    }
    __syncthreads();
    const int r = tid >> 3, wx = tid & 7;  // this thread's word of the bit planes
    const unsigned fg = fgb[tid];
    // ---- 2. + 3. certificates and their flood
    int cur = 0;
    bool converged = true;
    if (force_uf) {
        big[0][tid] = 0u;
        __syncthreads();
    } else if (min_keep <= 1) {
        big[0][tid] = fg;  // every component is kept
        __syncthreads();
    } else {
        unsigned hr = fg;  // bit x: the bw pixels x .. x + bw - 1 of the row are foreground
        for (int sft = 1; sft < bw; ++sft) hr &= bit_row_shift(fgb, r, wx, sft);
        big[1][tid] = hr;
        __syncthreads();
        unsigned seed = hr;
        for (int d = 1; d < bh; ++d) seed &= (r + d < EH) ? big[1][tid + 8 * d] : 0u;
        big[0][tid] = seed;
        __syncthreads();
        converged = false;
        for (int it = 0; it < kMorphFloodRounds; ++it) {
            const unsigned* b = big[cur];
            unsigned acc = 0u;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) acc |= bit_row_shift(b, r + dy, wx, -1) | bit_row_shift(b, r + dy, wx, 0) | bit_row_shift(b, r + dy, wx, 1);
            const unsigned nb = (acc & fg) | b[tid];
            big[cur ^ 1][tid] = nb;
            const int changed = __syncthreads_or(nb != b[tid]);
            cur ^= 1;
            if (!changed) {
                converged = true;
                break;
            }
        }
        if (!converged) {  // a maze: label everything
            __syncthreads();
            big[cur][tid] = 0u;
            __syncthreads();
        }
    }
    const unsigned bigw = big[cur][tid];
    const unsigned left = fg & ~bigw;
    unsigned* leftp = big[cur ^ 1];
    __syncthreads();  // everyone has read its word of big[cur ^ 1]'s predecessor state
    leftp[tid] = left;
    const int any_left = __syncthreads_or(left != 0u);
    // ---- 4. the rest: union-find, areas on the roots' slots, complete small components dropped
    if (any_left) {
        tile_forest_fn(EH, EW, 1, L, [&](int i) -> bool { return (leftp[i >> 5] >> (i & 31)) & 1u; });
        constexpr int EN = EW * EH;
        // root slots: 0x80000000 | touches-the-inner-frame << 30 | area (background stays 0xffffffff: area field all ones)
        for (int i = tid; i < EN; i += 1024)
            if (L[i] == i) L[i] = (int)0x80000001u;
        __syncthreads();
        for (int base = 0; base < EN; base += 1024) {  // whole waves: one add per run of equal roots (64 lanes = a quarter row)
            const int i = base + tid;
            const int v = L[i];
            const int root = v == -1 ? -1 : (v < 0 ? i : v);
            const int prev = __shfl_up(root, 1);
            const bool head = lane == 0 || root != prev;
            const unsigned long long heads = __ballot(head);
            const int ey = i >> 8, ex = i & (EW - 1);
            const int y = y0 + ey, x = x0 + ex;
            // the inner frame: the tile's outermost ring where the image continues beyond it
            const bool frame = root >= 0 && ((ey == 0 && y > 0) || (ey == EH - 1 && y < h - 1) || (ex == 0 && x > 0) || (ex == EW - 1 && x < w - 1));
            if (root >= 0) {
                unsigned* rec = reinterpret_cast<unsigned*>(&L[root]);
                if (head) {
                    const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
                    const unsigned len = (unsigned)((above ? __builtin_ctzll(above) : 64) - lane);
                    const unsigned add = len - ((v < 0 && root == i) ? 1u : 0u);  // the root counted itself
                    if (add) atomicAdd(rec, add);
                } else if (v < 0) {
                    atomicAdd(rec, 0xffffffffu);  // a root inside a run: its run head counted it as well
                }
                if (frame) atomicOr(rec, 1u << 30);
            }
        }
        __syncthreads();
        for (int base = 0; base < EN; base += 1024) {
            const int i = base + tid;
            const int v = L[i];
            bool keep = false;
            if (v != -1) {
                const unsigned rec = (unsigned)(v < 0 ? v : L[v]);
                keep = (rec & (1u << 30)) || (int)(rec & 0x3fffffffu) >= min_keep;
            }
            const unsigned long long kb = __ballot(keep);  // 64 consecutive pixels = words (i >> 5) and (i >> 5) + 1
            if (lane == 0) keepb[i >> 5] = (unsigned)kb;
            if (lane == 32) keepb[i >> 5] = (unsigned)(kb >> 32);
        }
        __syncthreads();
    }
    keepb[tid] |= bigw;
    __syncthreads();
    // ---- 5. dilation on the bit rows: out(q) = OR over the element of keep(q + off); outside the image = 0 (cv2.dilate's default)
    unsigned out = 0u;
    if (r >= halo && r < EH - halo) {
        if (n_off <= 256) {
            for (int k = 0; k < n_off; ++k) out |= bit_row_shift(keepb, r + s_off[2 * k], wx, s_off[2 * k + 1]);
        } else {
            for (int k = 0; k < n_off; ++k) out |= bit_row_shift(keepb, r + offs[2 * k], wx, offs[2 * k + 1]);
        }
        const int y = y0 + r;
        if (y < h) {  // (y >= 0: core rows start inside the image)
            uint8_t* drow = mask + plane_off + (size_t)y * w;
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                const int ex = 32 * wx + 4 * nb, x = x0 + ex;
                if (ex + 3 < halo || ex >= EW - halo || x >= w) continue;
                const unsigned q4 = (out >> (4 * nb)) & 0xfu;
                const bool whole = ex >= halo && ex + 3 < EW - halo && x + 3 < w && ((reinterpret_cast<uintptr_t>(drow + x) & 3) == 0);
                if (whole) {
                    *reinterpret_cast<uint32_t*>(drow + x) = (q4 & 1u) | ((q4 & 2u) << 7) | ((q4 & 4u) << 14) | ((q4 & 8u) << 21);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (ex + k >= halo && ex + k < EW - halo && x + k < w) drow[x + k] = (uint8_t)((q4 >> k) & 1u);
                }
            }
        }
    }
}

// scipy.ndimage.binary_fill_holes on a small plane: background components (4-connectivity) that do not reach the frame
__global__ __launch_bounds__(1024) void fill_holes_tile_kernel(const uint8_t* __restrict__ mask, int h, int w, uint8_t* __restrict__ out) {
    extern __shared__ int L[];
    const int hw = h * w, tid = threadIdx.x;
    const long plane_off = (long)blockIdx.x * hw;
    tile_forest<1>(mask, plane_off, h, w, 0, L);
    const int nb = 2 * w + 2 * h;
    for (int k = tid; k < nb; k += 1024) {
        int i;
        if (k < w) i = k;
        else if (k < 2 * w) i = (h - 1) * w + (k - w);
        else if (k < 2 * w + h) i = (k - 2 * w) * w;
        else i = (k - 2 * w - h) * w + (w - 1);
        const int v = L[i];
        if (v >= 0) atomicOr(&L[v & 0x3fffffff], 0x40000000);  // v: the root (a root's own slot may already carry the flag)
    }
    __syncthreads();
    for (int i = tid; i < hw; i += 1024) {
        const int v = L[i];
        const bool hole = v >= 0 && (L[v & 0x3fffffff] & 0x40000000) == 0;
        out[plane_off + i] = (mask[plane_off + i] != 0 || hole) ? 1 : 0;
    }
}

// ---- label areas -----------------------------------------------------------------------------------------
// Areas by run aggregation: a lane holds 4 consecutive pixels; lanes whose 4 pixels carry one label merge
// with their neighbours through a ballot (one update per run of lanes instead of one per pixel), and every
// lane keeps the last (label, count) it was about to add in registers across iterations, flushing only when
// the label changes -- a slide-sized component would otherwise serialise ~10^5 atomics on one address.
__global__ __launch_bounds__(BT) void area_count_kernel(const int* __restrict__ labels, long hw, int* __restrict__ areas) {
    const int* lb = labels + (size_t)blockIdx.y * hw;
    int* a = areas + (size_t)blockIdx.y * (hw + 1);
    const bool vec = (hw & 3) == 0;
    const int lane = lane_id();
    const long stride = (long)gridDim.x * BT * 4;
    int ckey = 0, ccnt = 0;
    for (long base = (long)blockIdx.x * BT * 4; base < hw; base += stride) {  // uniform trip count per block
        const long i0 = base + 4L * threadIdx.x;
        int v[4] = {0, 0, 0, 0};
        if (vec) {
            if (i0 < hw) {
                const int4 q = *reinterpret_cast<const int4*>(lb + i0);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < hw) v[k] = lb[i0 + k];
        }
        const bool uniform = v[0] == v[1] && v[1] == v[2] && v[2] == v[3];
        const int key = uniform ? v[0] : -1;
        const int prev = __shfl_up(key, 1);
        const bool head = lane == 0 || key != prev || key < 0;
        const unsigned long long heads = __ballot(head);
        if (uniform) {
            if (head && key > 0) {
                const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
                const int add = 4 * ((above ? __builtin_ctzll(above) : 64) - lane);
                if (key == ckey) {
                    ccnt += add;
                } else {
                    if (ccnt) atomicAdd(&a[ckey], ccnt);
                    ckey = key;
                    ccnt = add;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] > 0) atomicAdd(&a[v[k]], 1);
        }
    }
    if (ccnt) atomicAdd(&a[ckey], ccnt);
}
__global__ __launch_bounds__(BT) void area_filter_kernel(int* __restrict__ labels, long hw, const int* __restrict__ areas,
                                                          int min_keep) {
    const size_t off = (size_t)blockIdx.y * hw;
    const int* a = areas + (size_t)blockIdx.y * (hw + 1);
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < hw; i += (long)gridDim.x * BT) {
        const int l = labels[off + i];
        if (l > 0 && a[l] < min_keep) labels[off + i] = 0;
    }
}

// ---- binary morphology -------------------------------------------------------------------------------------
// Small structuring elements (every |dx| <= 4, at most 16 rows): the element is folded into one bit mask per
// row; a lane produces 4 adjacent output pixels from 3 aligned dword loads per element row (12 input bytes
// compressed to 12 window bits) instead of one byte load per element entry and pixel.  Anything else takes the
// generic per-entry loop.  Outside the image: dilate sees 0, erode sees 1 (OpenCV's default border).
__global__ __launch_bounds__(BT) void morph_kernel(const uint8_t* __restrict__ src, int h, int w,
                                                    const int* __restrict__ offs, int n_off, int op, int vec_ok,
                                                    uint8_t* __restrict__ dst) {
    __shared__ int s_dy[16];
    __shared__ unsigned s_mask[16];  // bit (dx + 4) set <=> (dy, dx) belongs to the element
    __shared__ int s_rows, s_fast;
    if (threadIdx.x == 0) {
        int rows = 0, fast = vec_ok;
        for (int k = 0; k < n_off && fast; ++k) {
            const int dy = offs[2 * k], dx = offs[2 * k + 1];
            if (dx < -4 || dx > 4) { fast = 0; break; }
            int r = 0;
            while (r < rows && s_dy[r] != dy) ++r;
            if (r == rows) {
                if (rows == 16) { fast = 0; break; }
                s_dy[rows] = dy;
                s_mask[rows] = 0u;
                ++rows;
            }
            s_mask[r] |= 1u << (dx + 4);
        }
        s_rows = rows;
        s_fast = fast;
    }
    __syncthreads();
    const long hw = (long)h * w;
    const uint8_t* s = src + (size_t)blockIdx.y * hw;
    uint8_t* d = dst + (size_t)blockIdx.y * hw;
    if (s_fast) {
        const int wq = w >> 2, rows = s_rows;
        const long nq = (long)h * wq;
        const uint32_t* sq = reinterpret_cast<const uint32_t*>(s);
        uint32_t* dq = reinterpret_cast<uint32_t*>(d);
        for (long q = (long)blockIdx.x * BT + threadIdx.x; q < nq; q += (long)gridDim.x * BT) {
            const int y = (int)(q / wq), xq = (int)(q - (long)y * wq);
            unsigned any4 = 0u, all4 = 0xfu;
            for (int r = 0; r < rows; ++r) {
                const int yy = y + s_dy[r];
                if (yy < 0 || yy >= h) continue;
                const uint32_t* row = sq + (long)yy * wq;
                unsigned win = 0u, valid = 0u;  // bit i <-> column 4*xq - 4 + i
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int c = xq - 1 + k;
                    if (c < 0 || c >= wq) continue;
                    const uint32_t v = row[c];
                    const uint32_t t = ((((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) >> 7) & 0x01010101u;  // byte != 0
                    win |= ((t * 0x01020408u) >> 24) << (4 * k);
                    valid |= 0xfu << (4 * k);
                }
                const unsigned m = s_mask[r];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned sel = m << j;  // element bits aligned to the window for output pixel j
                    if (win & sel) any4 |= 1u << j;
                    if ((~win & valid) & sel) all4 &= ~(1u << j);
                }
            }
            const unsigned res = op == 0 ? any4 : all4;
            dq[q] = (res & 1u) | ((res & 2u) << 7) | ((res & 4u) << 14) | ((res & 8u) << 21);
        }
        return;
    }
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < hw; i += (long)gridDim.x * BT) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        bool any = false, all = true;
        for (int k = 0; k < n_off; ++k) {
            const int yy = y + offs[2 * k], xx = x + offs[2 * k + 1];
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;  // dilate: outside = 0; erode: outside = 1
            const bool v = s[(long)yy * w + xx] != 0;
            any = any || v;
            all = all && v;
        }
        d[i] = (op == 0 ? any : all) ? 1 : 0;
    }
}

// ---- fill holes ------------------------------------------------------------------------------------------------
// after CCL of the background: mark background components that touch the border
__global__ __launch_bounds__(BT) void border_mark_kernel(const int* __restrict__ labels, int h, int w, int* __restrict__ flag) {
    const long hw = (long)h * w;
    const int* l = labels + (size_t)blockIdx.y * hw;
    int* f = flag + (size_t)blockIdx.y * hw;
    const long nb = 2L * w + 2L * h;
    for (long k = (long)blockIdx.x * BT + threadIdx.x; k < nb; k += (long)gridDim.x * BT) {
        long i;
        if (k < w) i = k;
        else if (k < 2L * w) i = (long)(h - 1) * w + (k - w);
        else if (k < 2L * w + h) i = (k - 2L * w) * w;
        else i = (k - 2L * w - h) * w + (w - 1);
        const int lab = l[i];
        if (lab > 0) f[lab - 1] = 1;
    }
}
__global__ __launch_bounds__(BT) void fill_apply_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ labels,
                                                         const int* __restrict__ flag, long hw, uint8_t* __restrict__ out) {
    const size_t off = (size_t)blockIdx.y * hw;
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < hw; i += (long)gridDim.x * BT) {
        const int lab = labels[off + i];  // > 0 only on background pixels
        const bool hole = lab > 0 && flag[off + lab - 1] == 0;
        out[off + i] = (mask[off + i] != 0 || hole) ? 1 : 0;
    }
}

}  // namespace tia

using namespace tia;

// ---- host side of the LDS-resident small-plane kernels (internal interface, common.hpp) -------------------------------------
namespace tia {
template <typename K>
static bool tile_lds_ok(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}
bool ccl_tile_enabled() {
    static const bool on = tia::dev_env("TIA_NO_CCL_TILE") == nullptr;
    if (!on) return false;
    static DeviceOnce once;
    static std::atomic<unsigned char> refused[64] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    if (refused[d].load(std::memory_order_acquire)) return false;
    const bool ok = once.ensure([] {
        const size_t cap = (size_t)kCclTileMaxPixels * sizeof(int);
        return tile_lds_ok(ccl_tile_kernel<0>, cap) && tile_lds_ok(ccl_tile_kernel<1>, cap) && tile_lds_ok(ccl_tile_kernel<2>, cap) &&
               tile_lds_ok(fill_holes_tile_kernel, cap) && tile_lds_ok(marker_tile_kernel, (size_t)kMarkerTileMaxPixels * 5) &&
               tile_lds_ok(morph_mask_tile_kernel<3>, (size_t)kMorphTileW * kMorphTileH * sizeof(int)) &&
               tile_lds_ok(morph_mask_tile_kernel<1>, (size_t)kMorphTileW * kMorphTileH * sizeof(int));
    });
    if (!ok) {
        (void)hipGetLastError();  // the refusal is handled here (multi-launch path): do not leave it for the next launch check
        refused[d].store(1, std::memory_order_release);
    }
    return ok;
}
int ccl_tile_label(const void* src, int src_kind, long n, int h, int w, int conn, int min_keep, int* labels, int* count, int* areas,
                   hipStream_t st, int* offs, int* bbox) {
    const size_t lds = (size_t)h * w * sizeof(int);
    if (!ccl_tile_enabled()) return TIA_ELAUNCH;  // callers ask ccl_tile_enabled() first
    const int c8 = conn == 8 ? 1 : 0;
    if (src_kind == 2)
        hipLaunchKernelGGL(ccl_tile_kernel<2>, dim3((unsigned)n), dim3(1024), lds, st, src, h, w, c8, min_keep, labels, count, areas, offs,
                           bbox);
    else if (src_kind == 1)
        hipLaunchKernelGGL(ccl_tile_kernel<1>, dim3((unsigned)n), dim3(1024), lds, st, src, h, w, c8, min_keep, labels, count, areas, offs,
                           bbox);
    else
        hipLaunchKernelGGL(ccl_tile_kernel<0>, dim3((unsigned)n), dim3(1024), lds, st, src, h, w, c8, min_keep, labels, count, areas, offs,
                           bbox);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
int marker_tile(const uint8_t* marker0, long n, int h, int w, int min_keep, int* labels, int* count, int* areas, hipStream_t st,
                const int* blob, int* inst, int* bbox) {
    if (!ccl_tile_enabled()) return TIA_ELAUNCH;
    hipLaunchKernelGGL(marker_tile_kernel, dim3((unsigned)n), dim3(1024), (size_t)h * w * 5, st, marker0, h, w, min_keep, labels, count, areas,
                       blob, inst, bbox);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
int fill_holes_tile(const uint8_t* mask, long n, int h, int w, uint8_t* out, hipStream_t st) {
    if (!ccl_tile_enabled()) return TIA_ELAUNCH;
    hipLaunchKernelGGL(fill_holes_tile_kernel, dim3((unsigned)n), dim3(1024), (size_t)h * w * sizeof(int), st, mask, h, w, out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
}  // namespace tia

static bool bad3(int64_t n, int64_t h, int64_t w) {
    return n <= 0 || h <= 0 || w <= 0 || n > 65535 || h * w > 0x7fffffffLL;
}

// ---- per-image byte look-up table (contrast_enhancer's intensity map, utils/misc.py:436-444) ------------------------
// out[i, j] = lut[i, img[i, j]]: the image's 256-byte table sits in LDS, bytes go through 16 at a time.
__global__ __launch_bounds__(BT) void lut_apply_kernel(const uint8_t* __restrict__ img, long len, const uint8_t* __restrict__ lut,
                                                        uint8_t* __restrict__ out) {
    __shared__ uint8_t t[256];
    const long base = (long)blockIdx.y * len;
    t[threadIdx.x] = lut[(long)blockIdx.y * 256 + threadIdx.x];
    __syncthreads();
    const uint8_t* src = img + base;
    uint8_t* dst = out + base;
    const long stride = (long)gridDim.x * BT;
    long done = 0;
    if ((((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0)) {
        const long nv = len >> 4;
        const uint4* q = reinterpret_cast<const uint4*>(src);
        uint4* o = reinterpret_cast<uint4*>(dst);
        for (long g = (long)blockIdx.x * BT + threadIdx.x; g < nv; g += stride) {
            const uint4 v = q[g];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[k] = (uint32_t)t[w[k] & 255u] | ((uint32_t)t[(w[k] >> 8) & 255u] << 8) | ((uint32_t)t[(w[k] >> 16) & 255u] << 16) |
                       ((uint32_t)t[w[k] >> 24] << 24);
            o[g] = make_uint4(r[0], r[1], r[2], r[3]);
        }
        done = nv << 4;
    }
    for (long i = done + (long)blockIdx.x * BT + threadIdx.x; i < len; i += stride) dst[i] = t[src[i]];
}

// ---- box down-sampling by an integer factor (slide thumbnail: cv2.INTER_AREA at an integer scale) -----------------------
// OpenCV's ResizeAreaFast for uint8: integer sum of the factor x factor box, times the float scale 1 / area, cvRound
// (round half to even) -- `saturate_cast<uchar>(sum * scale)` (modules/imgproc/src/resize.cpp).  One thread per output value.
__global__ __launch_bounds__(BT) void box_downsample_kernel(const uint8_t* __restrict__ src, int w, int c, int factor, int th, int tw,
                                                             uint8_t* __restrict__ out) {
    const long total = (long)th * tw * c;
    const float scale = 1.0f / (float)(factor * factor);
    for (long i = (long)blockIdx.x * BT + threadIdx.x; i < total; i += (long)gridDim.x * BT) {
        const int ch = (int)(i % c);
        const long p = i / c;
        const int ox = (int)(p % tw), oy = (int)(p / tw);
        const uint8_t* s = src + ((long)oy * factor * w + (long)ox * factor) * c + ch;
        unsigned sum = 0;
        for (int dy = 0; dy < factor; ++dy)
            for (int dx = 0; dx < factor; ++dx) sum += s[((long)dy * w + dx) * c];
        float v = rintf((float)sum * scale);
        out[i] = (uint8_t)(v > 255.0f ? 255.0f : v);
    }
}

extern "C" int tia_rgb2gray_u8(const uint8_t* d_img, int64_t npix, uint8_t* d_gray, void* stream) {
    if (!d_img || !d_gray || npix <= 0) return TIA_EINVAL;
    hipLaunchKernelGGL(rgb2gray_kernel, dim3(nblocks(npix >> 2 ? npix >> 2 : 1)), dim3(BT), 0, (hipStream_t)stream, d_img,
                       (long)npix, d_gray);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_hist256_u8(const uint8_t* d_data, int64_t n, uint32_t* d_hist, void* stream) {
    if (!d_data || !d_hist || n <= 0) return TIA_EINVAL;
    hipLaunchKernelGGL(hist256_kernel, dim3(nblocks(n >> 2 ? n >> 2 : 1, BT, 2048)), dim3(BT), 0, (hipStream_t)stream,
                       d_data, (long)n, d_hist);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

static int launch_threshold(const uint8_t* d_src, int64_t npix, int32_t is_rgb, int32_t thr, const int32_t* d_thr, uint8_t* d_mask,
                            void* stream) {
    if (!d_src || !d_mask || npix <= 0) return TIA_EINVAL;
    const bool wide = is_rgb && ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_mask)) & 15) == 0 && npix >= kPxChunk;
    if (wide) {
        long nb = (npix / kPxChunk + 15) / 16;  // four steps per wave
        nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
        hipLaunchKernelGGL(threshold_wide_kernel, dim3((unsigned)nb), dim3(BT), 0, (hipStream_t)stream, d_src, (long)npix, thr, d_thr,
                           d_mask);
    } else {
        hipLaunchKernelGGL(threshold_lt_kernel, dim3(nblocks((npix + 3) / 4)), dim3(BT), 0, (hipStream_t)stream, d_src, (long)npix,
                           is_rgb, thr, d_thr, d_mask);
    }
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_threshold_lt_u8(const uint8_t* d_src, int64_t npix, int32_t is_rgb, int32_t thr, uint8_t* d_mask,
                                   void* stream) {
    return launch_threshold(d_src, npix, is_rgb, thr, nullptr, d_mask, stream);
}

extern "C" int tia_threshold_lt_dev_u8(const uint8_t* d_src, int64_t npix, int32_t is_rgb, const int32_t* d_thr, uint8_t* d_mask,
                                       void* stream) {
    if (!d_thr) return TIA_EINVAL;
    return launch_threshold(d_src, npix, is_rgb, 0, d_thr, d_mask, stream);
}

extern "C" int tia_morph_mask_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, int32_t channels, int32_t thr,
                                 const int32_t* d_thr, int32_t min_region, const int32_t* d_offsets, int32_t n_off, int32_t reach,
                                 uint8_t* d_mask, void* stream) {
    if (!d_img || !d_mask || !d_offsets || n_off <= 0 || reach < 0 || min_region < 0 || (channels != 1 && channels != 3)) return TIA_EINVAL;
    if (bad3(n, h, w)) return TIA_ESIZE;
    const int halo = reach + (min_region > 1 ? min_region - 1 : 0);
    if (halo > kMorphMaxHalo || reach > 31 || !ccl_tile_enabled()) return TIA_ESIZE;  // the caller takes the multi-launch form
    const int cw = kMorphTileW - 2 * halo, ch = kMorphTileH - 2 * halo;
    const long tiles_x = (w + cw - 1) / cw, tiles_y = (h + ch - 1) / ch;
    const long blocks = tiles_x * tiles_y * n;
    if (blocks > 2147483647L) return TIA_ESIZE;
    // certificate rectangle: bw x bh >= min_region pixels, as square as possible (both <= 41 here)
    int bw = 1;
    while (bw * bw < min_region) ++bw;
    const int bh = min_region > 0 ? (min_region + bw - 1) / bw : 1;
    static const int force_uf = tia::dev_env("TIA_MORPH_FORCE_UF") ? 1 : 0;  // developer switch: every tile through the union-find
    const size_t lds = (size_t)kMorphTileW * kMorphTileH * sizeof(int);
    const long total_bytes = (long)n * h * w * channels;
    if (channels == 3)
        hipLaunchKernelGGL(morph_mask_tile_kernel<3>, dim3((unsigned)blocks), dim3(1024), lds, (hipStream_t)stream, d_img, (int)h, (int)w,
                           thr, d_thr, min_region, halo, bw, bh, d_offsets, n_off, (int)tiles_x, (int)tiles_y, total_bytes, force_uf, d_mask);
    else
        hipLaunchKernelGGL(morph_mask_tile_kernel<1>, dim3((unsigned)blocks), dim3(1024), lds, (hipStream_t)stream, d_img, (int)h, (int)w,
                           thr, d_thr, min_region, halo, bw, bh, d_offsets, n_off, (int)tiles_x, (int)tiles_y, total_bytes, force_uf, d_mask);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

// a wave may run kGrayHistMaxSteps steps of 1024 pixels (16-bit sub-histogram counters); otherwise ~8 steps per wave, at most three
// resident workgroups per CU
static long gray_hist_blocks(int64_t npix) {
    const long steps = (npix + tia::kPxChunk - 1) / tia::kPxChunk;
    long nb = (steps + 31) / 32;
    nb = nb > 768 ? 768 : nb;
    const long need = (steps + 4L * tia::kGrayHistMaxSteps - 1) / (4L * tia::kGrayHistMaxSteps);
    nb = nb < need ? need : nb;
    return nb < 1 ? 1 : nb;
}
static int launch_gray_hist(const uint8_t* d_img, int64_t npix, int32_t channels, uint32_t* d_hist, int32_t* d_fit, long nb, void* stream) {
    if (channels == 3)
        hipLaunchKernelGGL(tia::gray_hist_kernel<3>, dim3((unsigned)nb), dim3(tia::BT), 0, (hipStream_t)stream, d_img, (long)npix, d_hist, d_fit);
    else
        hipLaunchKernelGGL(tia::gray_hist_kernel<1>, dim3((unsigned)nb), dim3(tia::BT), 0, (hipStream_t)stream, d_img, (long)npix, d_hist, d_fit);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_gray_hist_u8(const uint8_t* d_img, int64_t npix, int32_t channels, uint32_t* d_hist, void* stream) {
    if (!d_img || !d_hist || npix <= 0 || (channels != 1 && channels != 3)) return TIA_EINVAL;
    const long nb = gray_hist_blocks(npix);
    if (nb > 2147483647L) return TIA_ESIZE;
    return launch_gray_hist(d_img, npix, channels, d_hist, nullptr, nb, stream);
}

extern "C" int tia_otsu_fit_u8(const uint8_t* d_img, int64_t npix, int32_t channels, uint32_t* d_hist, int32_t* d_out, void* stream) {
    if (!d_img || !d_hist || !d_out || npix <= 0 || (channels != 1 && channels != 3)) return TIA_EINVAL;
    const long nb = gray_hist_blocks(npix);
    if (nb > 2147483647L) return TIA_ESIZE;
    return launch_gray_hist(d_img, npix, channels, d_hist, d_out, nb, stream);
}

extern "C" int tia_otsu_threshold_u32(const uint32_t* d_hist, int32_t* d_out, void* stream) {
    if (!d_hist || !d_out) return TIA_EINVAL;
    hipLaunchKernelGGL(otsu_threshold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_hist, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_ccl_label_i32(const uint8_t* d_mask, int64_t n, int64_t h, int64_t w, int32_t connectivity,
                                  int32_t* d_labels, int32_t* d_count, int32_t* d_ws, void* stream) {
    if (!d_mask || !d_labels || !d_count || !d_ws) return TIA_EINVAL;
    if (bad3(n, h, w)) return TIA_ESIZE;
    if (connectivity != 4 && connectivity != 8) return TIA_EINVAL;
    if (ccl_tile_enabled() && h * w <= kCclTileMaxPixels)  // small planes: one launch, union-find in LDS (d_ws stays unused)
        return ccl_tile_label(d_mask, 0, n, (int)h, (int)w, connectivity, 0, d_labels, d_count, nullptr, (hipStream_t)stream);
    return ccl_run(d_mask, n, (int)h, (int)w, connectivity, d_labels, d_count, d_ws, false, (hipStream_t)stream);
}

extern "C" int tia_label_area_filter_i32(int32_t* d_labels, int64_t n, int64_t h, int64_t w, int32_t min_keep,
                                          int32_t* d_ws, void* stream) {
    if (!d_labels || !d_ws) return TIA_EINVAL;
    if (bad3(n, h, w)) return TIA_ESIZE;
    const long hw = (long)h * w;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_ws, 0, (size_t)n * (hw + 1) * sizeof(int32_t), st) != hipSuccess) return TIA_ELAUNCH;
    dim3 grid(nblocks(hw, BT, 4096), (unsigned)n);
    // few workgroups per plane (many iterations per lane) so the per-lane run cache gets to merge
    dim3 cgrid(nblocks(hw, BT * 4 * 16, n >= 64 ? 8 : 64), (unsigned)n);
    hipLaunchKernelGGL(area_count_kernel, cgrid, dim3(BT), 0, st, d_labels, hw, d_ws);
    hipLaunchKernelGGL(area_filter_kernel, grid, dim3(BT), 0, st, d_labels, hw, d_ws, min_keep);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_binary_morph_u8(const uint8_t* d_src, int64_t n, int64_t h, int64_t w, const int32_t* d_offsets,
                                    int32_t n_off, int32_t op, uint8_t* d_dst, void* stream) {
    if (!d_src || !d_dst || !d_offsets || n_off <= 0 || (op != 0 && op != 1)) return TIA_EINVAL;
    if (bad3(n, h, w)) return TIA_ESIZE;
    dim3 grid(nblocks((long)h * w, BT, 4096), (unsigned)n);
    const int vec_ok = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 3) == 0;
    hipLaunchKernelGGL(morph_kernel, grid, dim3(BT), 0, (hipStream_t)stream, d_src, (int)h, (int)w, d_offsets, n_off,
                       op, vec_ok, d_dst);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_fill_holes_u8(const uint8_t* d_mask, int64_t n, int64_t h, int64_t w, uint8_t* d_out,
                                  int32_t* d_ws, void* stream) {
    if (!d_mask || !d_out || !d_ws) return TIA_EINVAL;
    if (bad3(n, h, w)) return TIA_ESIZE;
    const long hw = (long)h * w;
    hipStream_t st = (hipStream_t)stream;
    if (ccl_tile_enabled() && hw <= kCclTileMaxPixels) return fill_holes_tile(d_mask, n, (int)h, (int)w, d_out, st);
    int32_t* labels = d_ws;
    int32_t* aux = d_ws + (size_t)n * hw;        // ranks, then border flags
    int32_t* counts = d_ws + 2 * (size_t)n * hw;  // [n] component counts (unused by the caller)
    const int rc = ccl_run(d_mask, n, (int)h, (int)w, 4, labels, counts, aux, true, st);
    if (rc != TIA_OK) return rc;
    if (hipMemsetAsync(aux, 0, (size_t)n * hw * sizeof(int32_t), st) != hipSuccess) return TIA_ELAUNCH;
    dim3 gb(nblocks(2L * (h + w), BT, 64), (unsigned)n);
    hipLaunchKernelGGL(border_mark_kernel, gb, dim3(BT), 0, st, labels, (int)h, (int)w, aux);
    dim3 grid(nblocks(hw, BT, 4096), (unsigned)n);
    hipLaunchKernelGGL(fill_apply_kernel, grid, dim3(BT), 0, st, d_mask, labels, aux, hw, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_lut_apply_u8(const uint8_t* d_img, int64_t n, int64_t len, const uint8_t* d_lut, uint8_t* d_out, void* stream) {
    if (!d_img || !d_lut || !d_out || n <= 0 || len <= 0) return TIA_EINVAL;
    if (n > 65535) return TIA_ESIZE;
    dim3 grid(nblocks((len + 15) / 16, BT, 2048), (unsigned)n);
    hipLaunchKernelGGL(lut_apply_kernel, grid, dim3(BT), 0, (hipStream_t)stream, d_img, (long)len, d_lut, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_box_downsample_u8(const uint8_t* d_src, int64_t h, int64_t w, int64_t c, int64_t factor, uint8_t* d_out,
                                      void* stream) {
    if (!d_src || !d_out || h <= 0 || w <= 0 || c <= 0 || factor <= 0) return TIA_EINVAL;
    if (factor > 4096 || w > 0x7fffffffL || h / factor <= 0 || w / factor <= 0) return TIA_ESIZE;
    const long th = h / factor, tw = w / factor;
    hipLaunchKernelGGL(box_downsample_kernel, dim3(nblocks(th * tw * c)), dim3(BT), 0, (hipStream_t)stream, d_src, (int)w, (int)c,
                       (int)factor, (int)th, (int)tw, d_out);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
