// calib.hip — micro-kernels that calibrate the two roofs bench.py prices k_path against (gfx950):
//   stream    wide coalesced 16-B-per-lane read of a buffer          -> what FETCH_SIZE reports per streamed byte
//   gather64  one random 64-B record (4 x 16 B) per lane, every record of the table read exactly once,
//             the access shape of k_path's node fetch           -> what FETCH_SIZE reports per gathered byte, and the
//             rate at which HBM / Infinity Cache serve such gathers (the roof of the C5 soup)
//   gather128 / gather256   the same with records of 2 / 4 ADJACENT 64-B sectors (aligned to their size): what a node record
//             laid out next to the record visited after it would cost — the roof for that access shape (round 3)
//   valu      independent plain fp32 v_mul / v_add (inline asm), 6 waves per SIMD -> VALU wave-instruction issue peak
//   valu_pk   the same with packed v_pk_mul_f32 / v_pk_add_f32 (two operations per lane per instruction)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/calib tools/calib.hip
// Run:   tools/calib <stream|gather64|gather128|gather256|valu|valu_pk> [MiB] [reps] [noflush]   (one JSON line per repetition; also under rocprofv3 --pmc FETCH_SIZE)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ src, size_t n, float* __restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;  // never true: keeps the loads
}

// record index of lane i: an odd multiplier modulo a power of two is a permutation, so every record is read once
__global__ __launch_bounds__(64) void k_gather64(const float4* __restrict__ table, uint32_t n_records_log2, uint32_t per_lane, float* __restrict__ out) {
    const uint32_t mask = (1u << n_records_log2) - 1u;
    const uint32_t lanes = gridDim.x * blockDim.x;
    const uint32_t me = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t i = (k * lanes + me);
        const uint32_t rec = (i * 2654435761u + 12345u) & mask;
        const float4* p = table + (size_t)rec * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// S adjacent 64-B sectors per lane: one record of 64 * S bytes, aligned to its size, every record read exactly once
template <int S>
__global__ __launch_bounds__(64) void k_gather_adj(const float4* __restrict__ table, uint32_t n_records_log2, uint32_t per_lane, float* __restrict__ out) {
    const uint32_t mask = (1u << n_records_log2) - 1u;
    const uint32_t lanes = gridDim.x * blockDim.x;
    const uint32_t me = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t i = (k * lanes + me);
        const uint32_t rec = (i * 2654435761u + 12345u) & mask;
        const float4* p = table + (size_t)rec * (4 * S);
        float4 v[4 * S];
#pragma unroll
        for (int j = 0; j < 4 * S; ++j) v[j] = p[j];
#pragma unroll
        for (int j = 0; j < 4 * S; ++j) acc += v[j].x + v[j].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// Independent plain fp32 VALU instructions written as inline asm, so that the compiler can neither pack them into
// v_pk_* pairs nor fold them: 8 accumulators x 16 = 128 wave instructions per iteration, alternately v_mul_f32 / v_add_f32.
__global__ __launch_bounds__(64) void k_valu(uint32_t iters, float seed, float* __restrict__ out) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 0.25f;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(m));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(m));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(m));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(c));
        }
    }
    const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 123.456f) out[0] = s;
}

// the same with packed instructions (v_pk_mul_f32 / v_pk_add_f32: two fp32 operations per lane each)
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_valu_pk(uint32_t iters, float seed, float* __restrict__ out) {
    float2v a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float2v m = {1.0000001f, 0.9999999f}, c = {0.25f, 0.125f};
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a7) : "v"(c));
        }
    }
    const float2v t = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (t.x + t.y == 123.456f) out[0] = t.x;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "stream";
    const size_t mib = argc > 2 ? (size_t)std::atoll(argv[2]) : 4096;
    const int reps = argc > 3 ? std::atoi(argv[3]) : 3;
    const bool noflush = argc > 4 && !std::strcmp(argv[4], "noflush");  // keep the table in L2 / Infinity Cache between repetitions
    CK(hipSetDevice(0));
    float* out = nullptr;
    CK(hipMalloc(&out, 256));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    if (!std::strcmp(what, "valu") || !std::strcmp(what, "valu_pk")) {
        const bool pk = !std::strcmp(what, "valu_pk");
        const uint32_t iters = 20000, grid = 256 * 24;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(a, 0));
            if (pk) hipLaunchKernelGGL(k_valu_pk, dim3(grid), dim3(64), 0, 0, iters, 1.0f, out);
            else hipLaunchKernelGGL(k_valu, dim3(grid), dim3(64), 0, 0, iters, 1.0f, out);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double insts = (double)grid * iters * 128;
            std::printf("{\"kernel\": \"%s\", \"waves\": %u, \"wave_insts\": %.0f, \"ms\": %.3f, \"g_wave_insts_per_s\": %.1f}\n", what, grid, insts, ms,
                        insts / (ms * 1e-3) / 1e9);
        }
        return 0;
    }
    size_t bytes = mib << 20;
    uint32_t log2rec = 0;
    while (((size_t)64 << (log2rec + 1)) <= bytes) ++log2rec;
    const int sectors = !std::strcmp(what, "gather128") ? 2 : !std::strcmp(what, "gather256") ? 4 : 1;
    if (sectors > 1) log2rec -= sectors == 2 ? 1 : 2;  // records of 128 / 256 bytes
    if (!std::strncmp(what, "gather", 6)) bytes = ((size_t)64 * sectors) << log2rec;  // power-of-two record count
    float4* buf = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    // a second buffer written between repetitions pushes the table out of L2 / Infinity Cache
    float4* flush = nullptr;
    const size_t flush_bytes = (size_t)1 << 30;
    CK(hipMalloc(&flush, flush_bytes));
    for (int r = 0; r < reps; ++r) {
        if (!noflush) CK(hipMemset(flush, r, flush_bytes));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        if (!std::strcmp(what, "stream")) {
            hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, buf, bytes / 16, out);
        } else {
            const uint32_t n_rec = 1u << log2rec, lanes = 256 * 24 * 64;
            const uint32_t per_lane = n_rec / lanes;  // table sizes here are multiples of 24 MiB x 2^k only approximately: the rest is not read
            if (sectors == 2) hipLaunchKernelGGL(k_gather_adj<2>, dim3(256 * 24), dim3(64), 0, 0, buf, log2rec, per_lane, out);
            else if (sectors == 4) hipLaunchKernelGGL(k_gather_adj<4>, dim3(256 * 24), dim3(64), 0, 0, buf, log2rec, per_lane, out);
            else hipLaunchKernelGGL(k_gather64, dim3(256 * 24), dim3(64), 0, 0, buf, log2rec, per_lane, out);
            bytes = (size_t)per_lane * lanes * 64 * sectors;
        }
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        std::printf("{\"kernel\": \"%s\", \"table_mib\": %zu, \"bytes_read\": %zu, \"ms\": %.3f, \"gb_per_s\": %.1f, \"g_records_per_s\": %.2f}\n", what, mib, bytes,
                    ms, bytes / (ms * 1e-3) / 1e9, bytes / (64.0 * sectors) / (ms * 1e-3) / 1e9);
    }
    return 0;
}
