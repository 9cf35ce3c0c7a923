// Contour polygons of label-map instances for HoVerNet.get_instance_info
// (reference tiatoolbox/models/architecture/hovernet.py:685-692:
//  cv2.findContours(inst_map, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]).
//
// Border following (Suzuki & Abe 1985, the algorithm behind findContours) is sequential along a
// border, so the parallel axis is the instance: one lane owns one (plane, id), scans the instance's
// bounding box in raster order, follows every outer and hole border it meets (the marks a hole border
// leaves are what stops the pixel right of a hole from being mistaken for a new outer border) and
// remembers the top-level outer border found last -- the element OpenCV returns first, because it links
// each new border at the front of its parent's child list.  Instances are disjoint, so all of them share
// one int8 mark plane per image: a pixel's mark only counts for the instance whose label it carries.
//
// The tree bookkeeping collapses to two border classes: 2 = outer border whose parent is the frame,
// 3 = any other border.  A new outer border is top-level iff the last border met on the row (LNBD)
// is the frame or of class 2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tiatoolbox_amd.h"

namespace {

constexpr int CT = 64;  // one wave per workgroup: lanes diverge completely, spread them over the chip

__device__ __constant__ int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__device__ __constant__ int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Plane {
    const int* lab;
    signed char* mark;  // nullptr: unmarked view (1 on the instance, 0 elsewhere)
    int h, w, id;
    __device__ int get(int x, int y) const {
        if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return 0;
        const long p = (long)y * w + x;
        if (lab[p] != id) return 0;
        return mark ? (int)mark[p] : 1;
    }
    __device__ void set(int x, int y, int v) const { mark[(long)y * w + x] = (signed char)v; }
};

// Follows one border from (x0, y0).  MARK: leave +-cls on the visited pixels.  Returns the number of
// CHAIN_APPROX_SIMPLE points; writes them to `out` (x, y pairs) when it is not null.
template <bool MARK, bool SIMPLE = true, class P = Plane>
__device__ int follow_border(const P& f, int x0, int y0, bool is_hole, int cls, long limit, int* out) {
    int s_end = is_hole ? 0 : 4, s = s_end;
    int x1, y1, v1;
    do {
        s = (s - 1) & 7;
        x1 = x0 + kDx[s];
        y1 = y0 + kDy[s];
        v1 = f.get(x1, y1);
    } while (v1 == 0 && s != s_end);
    if (v1 == 0) {  // isolated pixel
        if (MARK) f.set(x0, y0, -cls);
        if (out) { out[0] = x0; out[1] = y0; }
        return 1;
    }
    int npts = 0, x3 = x0, y3 = y0, prev_s = s ^ 4;
    for (long it = 0; it < limit; ++it) {
        int x4, y4;
        do {  // counter-clockwise from the direction after the one we arrived by; (x1,y1)/(previous) stops it
            ++s;
            x4 = x3 + kDx[s & 7];
            y4 = y3 + kDy[s & 7];
        } while (f.get(x4, y4) == 0 && s < 15);
        const bool passed_east = s > 8;  // direction 0 was examined and was zero
        s &= 7;
        if (MARK) {
            if (passed_east) f.set(x3, y3, -cls);
            else if (f.get(x3, y3) == 1) f.set(x3, y3, cls);
        }
        if (!SIMPLE || s != prev_s) {  // CHAIN_APPROX_NONE keeps every border pixel visited
            if (out) { out[2 * npts] = x3; out[2 * npts + 1] = y3; }
            ++npts;
            prev_s = s;
        }
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
    }
    return npts;
}

__global__ __launch_bounds__(CT) void contour_scan_kernel(const int* __restrict__ inst, signed char* __restrict__ mark,
                                                          const long long* __restrict__ stats, int h, int w,
                                                          int max_inst, long entries, int* __restrict__ meta) {
    const long e = (long)blockIdx.x * CT + threadIdx.x;
    if (e >= entries) return;
    const long plane = e / (max_inst + 1);
    const int id = (int)(e - plane * (max_inst + 1));
    int* m = meta + e * 4;
    m[0] = m[1] = m[2] = m[3] = 0;
    const long long* st = stats + e * 8;
    const long long area = st[0];
    if (id == 0 || area <= 0) return;
    const int xmin = (int)st[1], ymin = (int)st[2], xmax = (int)st[3], ymax = (int)st[4];
    Plane f{inst + plane * (long)h * w, mark + plane * (long)h * w, h, w, id};
    const long limit = 8 * area + 16;
    int bx = 0, by = 0, bn = 0;
    for (int y = ymin; y <= ymax; ++y) {
        int prev = 0, lnbd = 2;  // the frame counts as a top-level border
        for (int x = xmin; x <= xmax + 1; ++x) {
            int p = f.get(x, y);
            if (p != prev) {
                if (prev == 0 && p == 1) {
                    const int cls = lnbd == 2 ? 2 : 3;
                    const int n = follow_border<true>(f, x, y, false, cls, limit, nullptr);
                    if (cls == 2) { bx = x; by = y; bn = n; }
                    p = f.get(x, y);
                } else if (p == 0 && prev >= 1) {
                    if (prev > 1) lnbd = prev;
                    follow_border<true>(f, x - 1, y, true, 3, limit, nullptr);
                }
            }
            prev = p;
            if (p != 0 && p != 1) lnbd = p < 0 ? -p : p;
        }
    }
    m[0] = bx;
    m[1] = by;
    m[2] = bn;
}

// exclusive prefix sum of meta[:, 2] into meta[:, 3] and the grand total (one workgroup; the table is small)
__global__ __launch_bounds__(1024) void contour_offsets_kernel(int* __restrict__ meta, long entries, long long* __restrict__ total) {
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const long chunk = (entries + 1023) / 1024;
    const long lo = (long)t * chunk, hi = lo + chunk < entries ? lo + chunk : entries;
    long long s = 0;
    for (long i = lo; i < hi; ++i) s += meta[i * 4 + 2];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long run = part[t] - s;
    for (long i = lo; i < hi; ++i) {
        meta[i * 4 + 3] = (int)run;
        run += meta[i * 4 + 2];
    }
    if (t == 1023) *total = part[1023];
}

__global__ __launch_bounds__(CT) void contour_write_kernel(const int* __restrict__ inst, const long long* __restrict__ stats,
                                                           int h, int w, int max_inst, long entries,
                                                           const int* __restrict__ meta, long long capacity,
                                                           int* __restrict__ points) {
    const long e = (long)blockIdx.x * CT + threadIdx.x;
    if (e >= entries) return;
    const int* m = meta + e * 4;
    const int n = m[2];
    if (n <= 0 || (long long)m[3] + n > capacity) return;
    const long plane = e / (max_inst + 1);
    const int id = (int)(e - plane * (max_inst + 1));
    Plane f{inst + plane * (long)h * w, nullptr, h, w, id};
    follow_border<false>(f, m[0], m[1], false, 2, 8 * stats[e * 8] + 16, points + 2 * (long)m[3]);
}


// ---- all borders of binary planes: cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE | CHAIN_APPROX_NONE) ------------
// (reference hovernetplus.py:222-226).  A border's trace does not depend on the marks other borders leave, and the set of
// borders Suzuki-Abe's raster scan discovers is: the outer border of every 8-connected foreground component, started at
// the component's raster-first pixel, and the hole border of every 4-connected background component that does not reach
// the image frame, started at the pixel left of its raster-first pixel.  So the components come from the labelling
// kernels, the starts from one atomicMin pass, and every border is followed by its own lane.
struct MaskPlane {
    const uint8_t* m;
    int h, w;
    __device__ int get(int x, int y) const {
        if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return 0;
        return m[(long)y * w + x] != 0 ? 1 : 0;
    }
    __device__ void set(int, int, int) const {}
};

// first[plane][label] = smallest raster index of the label; edge[plane][label] = 1 if it touches the image frame
__global__ __launch_bounds__(256) void label_first_pixel_kernel(const int* __restrict__ lab, int h, int w, int kmax,
                                                                int* __restrict__ first, int* __restrict__ edge) {
    const long hw = (long)h * w;
    const int* l = lab + (size_t)blockIdx.y * hw;
    int* fp = first + (size_t)blockIdx.y * (kmax + 1);
    int* ep = edge + (size_t)blockIdx.y * (kmax + 1);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const int v = l[i];
        if (v <= 0 || v > kmax) continue;
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        // only a run's first pixel can be the component's first pixel
        if (x == 0 || l[i - 1] != v) atomicMin(&fp[v], (int)i);
        if (x == 0 || y == 0 || x == w - 1 || y == h - 1) ep[v] = 1;
    }
}

__global__ __launch_bounds__(CT) void border_trace_kernel(const uint8_t* __restrict__ mask, int h, int w,
                                                          const int* __restrict__ starts, long nb, int simple,
                                                          int* __restrict__ counts, const long long* __restrict__ offsets,
                                                          long long capacity, int* __restrict__ points) {
    const long e = (long)blockIdx.x * CT + threadIdx.x;
    if (e >= nb) return;
    const int* st = starts + e * 4;
    MaskPlane f{mask + (size_t)st[0] * h * w, h, w};
    const long limit = 8L * h * w + 16;
    int* out = nullptr;
    if (points) {
        if (offsets[e] + counts[e] > capacity) return;
        out = points + 2 * offsets[e];
    }
    const int n = simple ? follow_border<false, true>(f, st[1], st[2], st[3] != 0, 2, limit, out)
                         : follow_border<false, false>(f, st[1], st[2], st[3] != 0, 2, limit, out);
    if (!points) counts[e] = n;
}

}  // namespace

extern "C" int tia_hover_contour_scan(const int32_t* d_inst, int64_t n, int64_t h, int64_t w, int32_t max_inst,
                                      const int64_t* d_stats, int8_t* d_mark, int32_t* d_meta, int64_t* d_total,
                                      void* stream) {
    if (!d_inst || !d_stats || !d_mark || !d_meta || !d_total || n <= 0 || h <= 0 || w <= 0 || max_inst < 0)
        return TIA_EINVAL;
    const long entries = (long)n * (max_inst + 1);
    if (entries > (long)CT * 2147483647L) return TIA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_mark, 1, (size_t)n * h * w, st) != hipSuccess) return TIA_ELAUNCH;
    hipLaunchKernelGGL(contour_scan_kernel, dim3((unsigned)((entries + CT - 1) / CT)), dim3(CT), 0, st, d_inst,
                       (signed char*)d_mark, (const long long*)d_stats, (int)h, (int)w, max_inst, entries, d_meta);
    hipLaunchKernelGGL(contour_offsets_kernel, dim3(1), dim3(1024), 0, st, d_meta, entries, (long long*)d_total);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_hover_contour_write(const int32_t* d_inst, int64_t n, int64_t h, int64_t w, int32_t max_inst,
                                       const int64_t* d_stats, const int32_t* d_meta, int64_t capacity,
                                       int32_t* d_points, void* stream) {
    if (!d_inst || !d_stats || !d_meta || n <= 0 || h <= 0 || w <= 0 || max_inst < 0 || capacity < 0) return TIA_EINVAL;
    if (capacity == 0) return TIA_OK;
    if (!d_points) return TIA_EINVAL;
    const long entries = (long)n * (max_inst + 1);
    hipLaunchKernelGGL(contour_write_kernel, dim3((unsigned)((entries + CT - 1) / CT)), dim3(CT), 0, (hipStream_t)stream,
                       d_inst, (const long long*)d_stats, (int)h, (int)w, max_inst, entries, d_meta, (long long)capacity,
                       d_points);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_label_first_pixel_i32(const int32_t* d_labels, int64_t n, int64_t h, int64_t w, int32_t kmax,
                                         int32_t* d_first, int32_t* d_edge, void* stream) {
    if (!d_labels || !d_first || !d_edge || n <= 0 || h <= 0 || w <= 0 || kmax < 0 || n > 65535) return TIA_EINVAL;
    if ((long)h * w > 0x7fffffffL) return TIA_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)n * (kmax + 1);
    if (hipMemsetAsync(d_first, 0x7f, cells * 4, st) != hipSuccess) return TIA_ELAUNCH;  // 0x7f7f7f7f: "no pixel"
    if (hipMemsetAsync(d_edge, 0, cells * 4, st) != hipSuccess) return TIA_ELAUNCH;
    long blocks = ((long)h * w + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(label_first_pixel_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, st, d_labels, (int)h, (int)w,
                       kmax, d_first, d_edge);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}

extern "C" int tia_border_trace_u8(const uint8_t* d_mask, int64_t n, int64_t h, int64_t w, const int32_t* d_starts,
                                   int64_t nb, int32_t simple, int32_t* d_counts, const int64_t* d_offsets,
                                   int64_t capacity, int32_t* d_points, void* stream) {
    if (!d_mask || !d_starts || !d_counts || n <= 0 || h <= 0 || w <= 0 || nb < 0) return TIA_EINVAL;
    if (d_points && (!d_offsets || capacity < 0)) return TIA_EINVAL;
    if (nb == 0) return TIA_OK;
    hipLaunchKernelGGL(border_trace_kernel, dim3((unsigned)((nb + CT - 1) / CT)), dim3(CT), 0, (hipStream_t)stream, d_mask,
                       (int)h, (int)w, d_starts, (long)nb, simple, d_counts, (const long long*)d_offsets, (long long)capacity,
                       d_points);
    return hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH;
}
