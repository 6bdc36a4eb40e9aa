// intrate.hip — issue cost of the integer multiplies Philox is made of (gfx950), one wave's view: 8 independent chains of one instruction kind,
// 6 waves per SIMD like k_path.  Prints G wave-instructions/s per kind; 1024 SIMDs x clock / that = cycles per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/intrate tools/intrate.hip     Run: tools/intrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t iters, uint32_t seed, uint32_t* out) {
    uint32_t a[8];
    unsigned long long w[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; w[i] = a[i]; }
    const uint32_t m = 0xD2511F53u;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(w[i]) : "v"(a[i]), "v"(m) : "vcc");
                if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32);
    if (s == 0x12345u) out[0] = s;
}

int main() {
    CK(hipSetDevice(0));
    uint32_t* out;
    CK(hipMalloc(&out, 64));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const uint32_t iters = 4000, grid = 256 * 24;
    const char* names[5] = {"v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_add_u32", "v_xor_b32"};
    for (int kind = 0; kind < 5; ++kind)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, 0));
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, iters, 1u, out);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, iters, 1u, out);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, iters, 1u, out);
            if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, iters, 1u, out);
            if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), 0, 0, iters, 1u, out);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double insts = (double)grid * iters * 128;
            if (rep) std::printf("%-14s %8.1f G wave-inst/s  (%.2f cycles per instruction per SIMD at 2.2 GHz)\n", names[kind], insts / (ms * 1e-3) / 1e9, 1024 * 2.2 / (insts / (ms * 1e-3) / 1e9));
        }
    return 0;
}
