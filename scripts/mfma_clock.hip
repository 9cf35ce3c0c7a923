// Developer probe: what does the matrix pipe sustain?  Every wave issues back-to-back independent MFMAs (no memory traffic);
// the shader clock during the run = s_memtime ticks / s_memrealtime ticks x 100 MHz.  Prints the sustained rate next to
// the nominal peak so that a roofline fraction can be read against what the silicon holds under load.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_clock.hip -o scripts/bin/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int KIND>
__global__ __launch_bounds__(256) void probe(long iters, float* sink, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    f16x8 ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(1.0f + e * 0.001f); bh[e] = (_Float16)0.5f; }
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
            }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main(int argc, char** argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 20000;
    const int waves_per_simd = argc > 2 ? atoi(argv[2]) : 2;
    const int blocks = 256 * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
    float* sink; unsigned long long* clk;
    hipMalloc(&sink, 4); hipMalloc(&clk, blocks * 16);
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, iters * 4, sink, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 2);
            hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0;
            for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
            const double mhz = cyc / wall * 100.0;
            const double flop_per = kind == 0 ? 32.0 * 32 * 2 * 2 : 32.0 * 32 * 16 * 2;
            const double flops = (double)blocks * 4 * (kind == 0 ? iters : iters * 4) * 16 * flop_per;
            const double nominal = kind == 0 ? 157.3 : 2516.6;
            printf("%s  waves/SIMD %d  %.3f ms  %.1f TFLOP/s (%.1f%% of the nominal %.1f)  shader clock under load %.0f MHz\n",
                   kind == 0 ? "v_mfma_f32_32x32x2_f32  " : "v_mfma_f32_32x32x16_f16", waves_per_simd, ms, flops / ms / 1e9,
                   100.0 * flops / ms / 1e9 / nominal, nominal, mhz);
        }
    }
    return 0;
}
