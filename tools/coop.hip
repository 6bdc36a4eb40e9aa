// coop.hip — what a CU's vector-memory pipe charges for a load, and what a random gather of 64-B records reaches when every wave
// follows a dependent chain (the next index comes out of the record just read, as in a traversal) with 4 loads in flight.
// Random gather of 64-B records from a table of a given size, two ways, same bytes into the same number of VGPRs, same instruction count:
//   lane   every lane fetches its own record with 4 x buffer_load_dwordx4 (what k_path's node fetch does): 64 segments per instruction
//   quad   the 4 lanes of a quad fetch the records of the quad together: in instruction k every lane reads its 16-B quarter of
//          quad-lane k's record, so an instruction touches 16 segments, each with 4 adjacent lanes
// 6 waves per SIMD, every wave `iters` dependent rounds (the next index comes from the data just read, as in a traversal).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/coop tools/coop.hip      Run: tools/coop <table KiB> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4v quad16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// QUAD 2 / 3: the per-lane fetch with 16 of the 64 lanes active — lanes 0..15 (four whole quads) / every fourth lane (one per quad)
template <int QUAD>
__global__ __launch_bounds__(64) void k(const float4* table, uint32_t mask, uint32_t iters, float* out) {
    if (QUAD == 2 && (threadIdx.x & 63u) >= 16u) return;
    if (QUAD == 3 && (threadIdx.x & 3u) != 0u) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)((mask + 1u) * 64u), 0x00020000);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = mix(blockIdx.x * 64u + lane) & mask;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        f4v a, b, c, d;
        if (QUAD == 1) {
            const uint32_t q = lane & 3u;
            const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);  // quad_perm [0,0,0,0]
            const uint32_t i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);  // [1,1,1,1]
            const uint32_t i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true);  // [2,2,2,2]
            const uint32_t i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);  // [3,3,3,3]
            a = quad16(r, (i0 << 6) + 16u * q);
            b = quad16(r, (i1 << 6) + 16u * q);
            c = quad16(r, (i2 << 6) + 16u * q);
            d = quad16(r, (i3 << 6) + 16u * q);
        } else {
            a = quad16(r, idx << 6);
            b = quad16(r, (idx << 6) + 16u);
            c = quad16(r, (idx << 6) + 32u);
            d = quad16(r, (idx << 6) + 48u);
        }
        const float s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
        acc += s;
        idx = mix(idx + __float_as_uint(s) + it) & mask;  // dependent: the next record follows from this one
    }
    if (acc == 12345.678f) out[0] = acc;
}

// the per-lane fetch with loads of W dwords (4 loads per round at offsets 0, 16, 32, 48 of the record): what a load costs by its width
template <int W>
__global__ __launch_bounds__(64) void kw(const float4* table, uint32_t mask, uint32_t iters, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)((mask + 1u) * 64u), 0x00020000);
    uint32_t idx = mix(blockIdx.x * 64u + (threadIdx.x & 63u)) & mask;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = (int)(idx << 6) + 16 * q;
            if (W == 1) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
            if (W == 2) { typedef float f2v __attribute__((ext_vector_type(2))); const f2v v = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0)); s += v.x + v.y; }
            if (W == 3) { typedef float f3v __attribute__((ext_vector_type(3))); const f3v v = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0)); s += v.x + v.y + v.z; }
        }
        acc += s;
        idx = mix(idx + __float_as_uint(s) + it) & mask;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t kib = argc > 1 ? (size_t)std::atoll(argv[1]) : 1024;
    const uint32_t iters = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 2000;
    CK(hipSetDevice(0));
    if (kib > 2048 * 1024) { std::fprintf(stderr, "tables up to 2 GiB (a buffer resource addresses 4 GiB - 1)\n"); return 2; }
    size_t n = kib * 1024 / 64, p = 1;
    while (p * 2 <= n) p *= 2;
    n = p;
    std::vector<float> h(n * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    float4* table;
    float* out;
    CK(hipMalloc(&table, n * 64));
    CK(hipMalloc(&out, 64));
    CK(hipMemcpy(table, h.data(), n * 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t grid = 256 * 24;
    const char* names[4] = {"lane", "quad", "lane_16_first", "lane_16_spread"};
    for (int quad = 0; quad < 4; ++quad)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (quad == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            else if (quad == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            else if (quad == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double loads = (double)grid * iters * 4;  // wave-level load instructions
            if (rep) std::printf("{\"table_kib\": %zu, \"mode\": \"%s\", \"ms\": %.3f, \"g_wave_loads_per_s\": %.2f, \"cycles_per_load_per_cu\": %.1f, \"gb_per_s\": %.0f}\n", n * 64 / 1024, names[quad],
                                 ms, loads / (ms * 1e-3) / 1e9, 256 * 2.3e9 / (loads / (ms * 1e-3)), loads * 1024 / (ms * 1e-3) / 1e9);
        }
    for (int w = 1; w <= 3; ++w)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (w == 1) hipLaunchKernelGGL(kw<1>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            if (w == 2) hipLaunchKernelGGL(kw<2>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            if (w == 3) hipLaunchKernelGGL(kw<3>, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double loads = (double)grid * iters * 4;
            if (rep) std::printf("{\"table_kib\": %zu, \"mode\": \"lane_dwordx%d\", \"ms\": %.3f, \"g_wave_loads_per_s\": %.2f, \"cycles_per_load_per_cu\": %.1f}\n", n * 64 / 1024, w, ms,
                                 loads / (ms * 1e-3) / 1e9, 256 * 2.3e9 / (loads / (ms * 1e-3)));
        }
    return 0;
}
