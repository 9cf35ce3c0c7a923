// Instruction-semantics probe for gfx950 (kernel work aid, not product code): rounding of v_cvt_pk_u8_f32, unaligned ds_read_u16,
// v_mul_hi_u32_u24.   hipcc --offload-arch=gfx950 -O2 scripts/isa_probe.hip -o scripts/bin/isa_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

__global__ void probe(const float* in, int n, uint32_t* out, uint32_t* misc) {
    __shared__ uint16_t tab[64];
    int t = threadIdx.x;
    if (t < 64) tab[t] = (uint16_t)(0x1000 + t);
    __syncthreads();
    if (t < n) {
        uint32_t r = 0;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(in[t]), "v"(0xAABBCCDDu));
        out[t] = r;
    }
    if (t == 0) {
        // unaligned ds_read_u16 at byte offset 3 of tab (bytes: 00 10 01 10 02 10 ...) -> expect 0x0210 if unaligned access works
        uint32_t addr = (uint32_t)(uintptr_t)tab + 3, v = 0;
        asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
        misc[0] = v;
        uint32_t a = 0x00ABCDEF, b = 1u << 21, hi = 0;
        asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
        misc[1] = hi;  // expect a >> 11
        misc[2] = a >> 11;
    }
}

int main() {
    float h[] = {0.5f, 1.5f, 2.5f, 3.5f, 0.49999f, 254.5f, 255.5f, 256.0f, 300.0f, -0.4f, -3.0f, 127.50001f, 2.4999998f, 1e9f, 0.0f / 0.0f, 100.0f};
    const int n = sizeof(h) / 4;
    float* d;
    uint32_t *o, *m;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&o, n * 4);
    hipMalloc(&m, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, o, m);
    uint32_t ho[n], hm[3];
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hm, m, 12, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) -> %08x (byte1 = %u)\n", h[i], ho[i], (ho[i] >> 8) & 255);
    printf("unaligned ds_read_u16 @3 = %04x (0210 if supported)\nmul_hi_u32_u24 = %x expect %x\n", hm[0], hm[1], hm[2]);
    return 0;
}
