// pt_kernels.h — device data layout and the host-callable launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptcore.h"
#include "pt_pixelmap.h"

namespace pt {

enum { PRIM_TRIANGLE = 0, PRIM_SPHERE = 1 };

// Read-only scene data resident in HBM (replicated per GPU).  See DESIGN.md "Data layout".
struct DevScene {
    const float4* pairs;       // 4 x float4 (64 B) per internal node: {Lmin,Lref}{Lmax,0}{Rmin,Rref}{Rmax,0}
    const float4* prim;        // triangles: kTriStride (4) x float4 per slot {p0.xyz,e1.x}{e1.yz,e2.xy}{e2.z,id,mat,0}{unused}
                               // spheres:   1 x float4 per slot {c.xyz, r}
    const uint32_t* prim_meta; // spheres: 2 x u32 per slot {object id, material}
    const float* nrm;          // triangles: 9 floats per slot (normal0..2), leaf order
    const uint32_t* big_leaf;  // (first, count) of leaves with > 31 primitives
    const pt_material* mats;
    const float* sky;          // 2500 x 1250 x 3 floats + zero padding up to kSkyMaxIndex
    const uint32_t* lights;    // slots of emissive primitives
    uint32_t n_lights;
    uint32_t n_prims;
    uint32_t n_pairs;          // records in `pairs` (one more is allocated behind them)
    uint32_t root_ref;
    uint32_t stack_cap;        // most entries a traversal can postpone (= tree levels - 1)
    float root_min[3], root_max[3];
    // When the NEE block of camera.rs:314-338 may be left out (its weight min(1 - speculaty, 0) is exactly 0): only while
    // every component of transport is at most nee_thr in magnitude (then the block's product is finite and 0 * it adds
    // nothing), and never when nee_literal is set (a light whose area or colour is not finite).  See shade_core.
    float nee_thr;
    uint32_t nee_literal;
    uint32_t skip_repeated_miss;  // PT_OPT_SKIP_REPEATED_MISS (opt-in): see shade_core
};

// Paths a kernel without the NEE code could not finish (transport beyond nee_thr at a diffuse hit): their slots, for k_redo.
struct RedoList {
    uint32_t* list;
    uint32_t* count;   // entries asked for (may exceed cap: the host then runs the batch again with the NEE kernels)
    uint32_t cap;
};

// Camera after update() (src/camera.rs:271-292), in device memory.
struct DevCamera {
    float origin[3], p1[3], p2[3], p3[3], right[3], up[3], direction[3], target[3];
    float focal_distance, lens_size;
    uint32_t width, height;
};

// Per-path state, structure of float4 arrays indexed by slot = s_local * npix_pad + idx.
struct PathState {
    float4* ro;   // origin.xyz, ray.distance
    float4* rd;   // direction.xyz, cos_intersection saved across the look-ahead trace
    float4* hit;  // bits(slot | inside << 31) or PT_MISS, ray.distance after the trace, u, v
    float4* tp;   // transport.rgb, bits(iteration | phase << 16 | material << 17)
    float4* out;  // radiance of the finished path
    float4* acc;  // accumalated_color while the path is alive (only scenes whose NEE weight is non-zero)
    uint2* pxs;   // pixel index (y * width + x) and global sample index: the RNG counter words
    RedoList redo;
};

// CNT_NODE_TESTS / CNT_PRIM_TESTS are the round kernels' (k_extend, k_trace); the fused tail kernel keeps its own
enum { CNT_PATHS = 0, CNT_RAYS_CLOSEST, CNT_RAYS_ANY, CNT_NODE_TESTS, CNT_PRIM_TESTS, CNT_MAX_NODE_TESTS, CNT_TAIL_RAYS,
       CNT_TAIL_NODE_TESTS, CNT_TAIL_PRIM_TESTS,
       CNT_PHASE,  // diagnostics of a -DPT_PHASE_STATS build: 6 phases x {executions, lanes, clocks}
       CNT_N = CNT_PHASE + 18 };

struct LaunchCfg {
    uint32_t grid;          // workgroups of the persistent traversal launch
    unsigned long long* spill;  // overflow of the per-lane LDS stacks: [level][grid * block]
    uint32_t spill_stride;  // grid * block
};

#ifndef PT_TRACE_BLOCK
#define PT_TRACE_BLOCK 64
#endif
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12
#endif
#ifndef PT_WAVES_PER_CU
#define PT_WAVES_PER_CU 24
#endif
#ifndef PT_TRI_STRIDE
#define PT_TRI_STRIDE 4
#endif
#ifndef PT_NRM_STRIDE
#define PT_NRM_STRIDE 9
#endif
#ifndef PT_COOP_LEAF_FROM
#define PT_COOP_LEAF_FROM 16
#endif
constexpr uint32_t kCoopLeafFrom = PT_COOP_LEAF_FROM;  // triangle leaves of at least this many primitives are tested by the whole wave (0 = never)
constexpr int kTriStride = PT_TRI_STRIDE;     // float4 per triangle slot: 4 = one 64-byte sector each (the record is 48 B; packed, every other
                                              // one straddled two sectors: C3 +0.9 %, C5 +1.2 %, profiles/r2_ab_triangle_record_64B.txt)
constexpr int kNrmStride = PT_NRM_STRIDE;     // floats per slot of the normals array: 9 = packed; 16 = {n0,_}{n1,_}{n2,_}{_} in one 64-byte sector
                                              // measured 0.5 % slower on C3 / C4 (profiles/r2_ab_normal_record_64B.txt): shading reads them once per hit
constexpr int kTraceBlock = PT_TRACE_BLOCK;  // threads per traversal workgroup
constexpr int kLdsStack = PT_LDS_STACK;      // stack entries per lane kept in LDS
constexpr int kWavesPerCU = PT_WAVES_PER_CU; // persistent grid: waves launched per CU (what fits is decided by LDS and VGPRs)
// k_path's second form for scenes whose records come from HBM (their steps wait longest, so one more wave per SIMD pays more than
// the 8 registers it takes away: C5 +3.4 %, profiles/r3_ab_waves_per_simd_final_kernels.txt): 7 waves per SIMD at 72 VGPRs, and one LDS
// stack level less so that 28 waves fit a CU's 160 KB
constexpr int kLdsStack7 = PT_LDS_STACK - 1;
constexpr int kWavesPerCU7 = 28;

// parity seams
void launch_trace(int prim, bool any_hit, bool count, const DevScene& S, const pt_ray* rays, uint64_t n, pt_hit* hits,
                  const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st);
void launch_generate(const DevCamera* cam, const uint32_t* xy, uint64_t n, uint32_t sample, uint32_t k0, uint32_t k1, pt_ray* rays,
                     hipStream_t st);
void launch_sky(const DevScene& S, const float* dirs, uint64_t n, float* rgb, hipStream_t st);
void launch_focus(int prim, const DevScene& S, DevCamera* cam, const LaunchCfg& cfg, hipStream_t st);

// the wavefront pipeline
void launch_raygen(const DevCamera* cam, const PixelMap& M, const PathState& P, uint32_t n_samples, uint32_t sample_base, uint32_t k0,
                   uint32_t k1, bool nee, uint32_t* queue, uint32_t* qn, unsigned long long* counters, hipStream_t st);
void launch_extend(int prim, bool count, bool simple, const DevScene& S, const PathState& P, const uint32_t* queue, const uint32_t* qn,
                   uint32_t* cursor, const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st);
void launch_shade(int prim, bool nee, const DevScene& S, const PathState& P, const PixelMap& M, const uint32_t* q_in, const uint32_t* qn_in,
                  uint32_t* q_out, uint32_t* qn_out, uint32_t depth, uint32_t sample_base, uint32_t k0, uint32_t k1,
                  const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st);
void launch_tail(int prim, bool nee, bool count, int mode, const DevScene& S, const PathState& P, const uint32_t* queue, const uint32_t* qn,
                 uint32_t* cursor, uint32_t n_hint, uint32_t depth, uint32_t k0, uint32_t k1, const LaunchCfg& cfg, unsigned long long* counters,
                 hipStream_t st);
void launch_path(int prim, bool nee, bool count, const DevScene& S, const DevCamera* cam, const PixelMap& M, uint32_t n_samples,
                 uint32_t sample_base, uint32_t depth, uint32_t k0, uint32_t k1, float4* out, uint32_t* cursor, const RedoList& redo,
                 const LaunchCfg& cfg, bool seven_waves, unsigned long long* counters, hipStream_t st);
// the paths on a redo list, start to end, with the NEE code (one lane per path); pixel_major = the slot layout of k_path
void launch_redo(int prim, bool count, const DevScene& S, const DevCamera* cam, const PixelMap& M, uint32_t n_samples, bool pixel_major,
                 uint32_t sample_base, uint32_t depth, uint32_t k0, uint32_t k1, float4* out, const uint32_t* list, uint32_t n,
                 const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st);
// flag[0] |= 1 if a light's Primitive::area() is not a finite number >= 0 (such a scene always runs the NEE block)
void launch_light_check(int prim, const DevScene& S, uint32_t* flag, hipStream_t st);
void launch_accumulate(const PixelMap& M, const float4* out, uint32_t n_samples, bool pixel_major, float* accum, hipStream_t st);
void launch_shard_copy(const PixelMap& M, float* accum, float* slab, bool unpack, hipStream_t st);
void launch_resolve(const float* accum, uint32_t width, uint32_t height, float scale, uint8_t* rgb, uint64_t pitch, hipStream_t st);

}  // namespace pt
