// pt_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the path-tracing hot path.
//
//   k_raygen      Camera::generate                      src/camera.rs:426-449
//   k_extend      BVH::intersect_closest                src/bvh.rs:180-226  (+ triangle.rs:41-79, sphere.rs:34-80)
//   k_shade       body of the loop in Camera::sample    src/camera.rs:299-421 (+ scene.rs:162-169)
//                 and wave64 ballot/mbcnt compaction of the surviving paths
//   k_accumulate  chunk[idx] += sample                  src/main.rs:159
//   k_resolve     vec_to_rgb(scale * accum)             src/main.rs:71-75,160-164
//   k_trace / k_generate / k_sky / k_focus              parity seams and Camera::update + focus
//
// Traversal keeps the reference's exact visiting order (near child first by `tlmin <= trmin`,
// last accepted primitive wins, no distance test for triangles), so hit ids are bit-exact.
// The only restructurings are ones that cannot change a result:
//   * 1/direction is computed once per ray instead of once per AABB test;
//   * the reference re-tests a node's AABB when it is popped (bvh.rs:196).  tmin/tmax of that
//     test do not depend on ray.distance, so the re-test reduces to `!(ray.distance < tmin)`
//     with the tmin computed when the node was pushed; tmin is kept on the stack instead of
//     re-fetching the node.  The near child is visited directly (its re-test happens with an
//     unchanged ray.distance and always passes);
//   * normal / area of a hit are computed once for the winning primitive, in the shade kernel.
// Built with -ffp-contract=off: every float op is a single IEEE-754 operation.
#include "pt_kernels.h"

#include "pt_bvh.h"
#include "pt_device.h"

#pragma clang fp contract(off)

// k_path: 6 waves per SIMD (80 VGPRs, the shading temporaries spill) measured faster than 4 waves
// without spills: the kernel lives on latency hiding.
#ifndef PT_PATH_ATTR
#define PT_PATH_ATTR __attribute__((amdgpu_waves_per_eu(SEVEN ? 7 : 6, SEVEN ? 7 : 6)))  // SEVEN: k_path's template argument (pt_kernels.h, kLdsStack7)
#endif

namespace pt {

__device__ __forceinline__ V3 xyz(float4 v) { return V3{v.x, v.y, v.z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }

// Node and triangle records through buffer loads (PT_BUFFER_LOADS): the record's byte offset is ONE 32-bit shift of its index
// (records are 64 B, at most 2^26 of them: the offset fits 32 bits) and the base lives in a scalar resource descriptor, where
// a flat global load needs a 64-bit shift and a 64-bit add per record — quarter-rate instructions — in the hottest loop.
#ifndef PT_BUFFER_LOADS
#define PT_BUFFER_LOADS 1  // C3 +4.5 %, C4 +3.3 %, C5 +4.1 % (profiles/r3_ab_buffer_loads.txt)
#endif
#if PT_BUFFER_LOADS
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t record_rsrc(const float4* base, uint32_t n_records) {
    const unsigned long long bytes = (unsigned long long)n_records * 64ull;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0xffffffffull ? (int)0xffffffffu : (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ float4 record_quad(__amdgpu_buffer_rsrc_t r, uint32_t index, int quad) {
    const f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(index << 6) + 16 * quad, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
// PT_TOUCH_NEXT: at the end of a traversal turn of k_path a lane knows the record it will read at its next turn (T.cur).  A one-dword
// buffer load of that record WITH LDS AS ITS DESTINATION (no VGPR, nothing waits for it) starts the fetch a scheduling round early.
// Not speculative: it is the record the lane reads next, whatever happens.  1 = node records, 2 = also the first triangle of a leaf.
#ifndef PT_TOUCH_NEXT
#define PT_TOUCH_NEXT 0
#endif
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void record_touch(__amdgpu_buffer_rsrc_t r, uint32_t index, lds_u32* wave_dump) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, wave_dump, 4, (int)(index << 6), 0, 0, 0);
}
#endif

// ------------------------------------------------------------------------------------------------
// Per-lane traversal stack: the first kLdsStack levels live in LDS laid out [level][thread] so that
// a wave's accesses hit 64 consecutive 8-byte words whatever level each lane is at (no bank
// conflicts under divergence); deeper levels spill to a global array with the same shape.
// One entry = (child reference, tmin bits) packed in 64 bits.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
struct LaneStack {
    lds_u64* lds;         // &smem[threadIdx.x], kept in the LDS address space so accesses are ds_read/ds_write_b64
    unsigned long long* spill;  // the launch's spill area (wave-uniform; a lane's entries are at [level][global thread id])
    uint32_t spill_stride;
    uint32_t levels;      // levels kept in LDS (a constant at every call site of make_stack: kLdsStack, or kLdsStack7 in k_path's 7-wave form)
    __device__ __forceinline__ unsigned long long* spill_at(uint32_t level) const {
        return spill + ((size_t)(level - levels) * spill_stride + (blockIdx.x * blockDim.x + threadIdx.x));
    }
    __device__ __forceinline__ void put(uint32_t level, uint32_t ref, float tmin) const {
        const unsigned long long e = (unsigned long long)ref | ((unsigned long long)__float_as_uint(tmin) << 32);
        if (level < levels) lds[level * kTraceBlock] = e;
        else *spill_at(level) = e;
    }
    __device__ __forceinline__ void get(uint32_t level, uint32_t* ref, float* tmin) const {
        unsigned long long e;
        if (level < levels) e = lds[level * kTraceBlock];
        else e = *spill_at(level);
        *ref = (uint32_t)e;
        *tmin = __uint_as_float((uint32_t)(e >> 32));
    }
};

struct Hit {
    uint32_t slot;    // leaf-order primitive slot, PT_MISS if none
    float u, v;       // barycentrics of the winning triangle
    uint32_t inside;  // Intersection::inside
};

// BVH::intersect_closest (ANY = false) / intersect_any (ANY = true).  `dist` is ray.distance
// (in/out).  COUNT adds the algorithmic work counters used for the roofline.
template <int PRIM, bool ANY, bool COUNT>
__device__ __forceinline__ void traverse(const DevScene& S, V3 o, V3 d, float& dist, Hit& hit, LaneStack stk,
                                         uint32_t& n_node_tests, uint32_t& n_prim_tests) {
    hit.slot = PT_MISS;
    hit.u = hit.v = 0.0f;
    hit.inside = 0;
    const V3 rcp = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    float tmin;
    if (COUNT) n_node_tests += 1;
    if (!slab(S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1], S.root_max[2], o, rcp, dist, &tmin)) return;
    uint32_t cur = S.root_ref;
    uint32_t sp = 0;
    while (true) {
        bool need_pop;
        if (!(cur & kLeafBit)) {
#if PT_BUFFER_LOADS
            const __amdgpu_buffer_rsrc_t rp = record_rsrc(S.pairs, S.n_pairs + 1u);
            const float4 a = record_quad(rp, cur, 0), b = record_quad(rp, cur, 1), c = record_quad(rp, cur, 2), e = record_quad(rp, cur, 3);
#else
            const float4* rec = S.pairs + (size_t)cur * 4;
            const float4 a = rec[0], b = rec[1], c = rec[2], e = rec[3];
#endif
            float tl, tr;
            const bool hl = slab(a.x, a.y, a.z, b.x, b.y, b.z, o, rcp, dist, &tl);
            const bool hr = slab(c.x, c.y, c.z, e.x, e.y, e.z, o, rcp, dist, &tr);
            if (COUNT) n_node_tests += 2;
            const uint32_t lref = __float_as_uint(a.w), rref = __float_as_uint(c.w);
            if (hl && hr) {
                const bool left_first = tl <= tr;  // bvh.rs:211
                stk.put(sp, left_first ? rref : lref, left_first ? tr : tl);
                sp += 1;
                cur = left_first ? lref : rref;
                need_pop = false;
            } else if (hl) {
                cur = lref;
                need_pop = false;
            } else if (hr) {
                cur = rref;
                need_pop = false;
            } else {
                need_pop = true;
            }
        } else {
            uint32_t first = cur & kLeafFirstMask;
            uint32_t count = (cur >> kLeafCountShift) & 31u;
            if (count == 0) {
                count = S.big_leaf[2 * first + 1];
                first = S.big_leaf[2 * first];
            }
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t slot = first + k;
                if (COUNT) n_prim_tests += 1;  // per call: intersect_any leaves the leaf at its first hit (bvh.rs:246-248)
                if (PRIM == PRIM_TRIANGLE) {
#if PT_BUFFER_LOADS
                    const __amdgpu_buffer_rsrc_t rt = record_rsrc(S.prim, S.n_prims);
                    const float4 a = record_quad(rt, slot, 0), b = record_quad(rt, slot, 1), c = record_quad(rt, slot, 2);
#else
                    const float4* tp = S.prim + (size_t)slot * kTriStride;
                    const float4 a = tp[0], b = tp[1], c = tp[2];
#endif
                    float t, u, v, det;
                    if (tri_hit(V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, V3{b.z, b.w, c.x}, o, d, &t, &u, &v, &det)) {
                        dist = t;  // unconditionally, triangle.rs:72
                        hit.slot = slot;
                        hit.u = u;
                        hit.v = v;
                        hit.inside = det < 0.0f ? 1u : 0u;
                        if (ANY) return;
                    }
                } else {
                    const float4 s = S.prim[slot];
                    float t;
                    bool inside;
                    if (sphere_hit(V3{s.x, s.y, s.z}, s.w, o, d, dist, &t, &inside)) {
                        dist = t;
                        hit.slot = slot;
                        hit.inside = inside ? 1u : 0u;
                        if (ANY) return;
                    }
                }
            }
            need_pop = true;
        }
        if (need_pop) {
            // bvh.rs:194-196: pop, re-test against the current ray.distance
            while (true) {
                if (sp == 0) return;
                sp -= 1;
                uint32_t ref;
                float tpush;
                stk.get(sp, &ref, &tpush);
                if (!(dist < tpush)) {
                    cur = ref;
                    break;
                }
            }
        }
    }
}

__device__ __forceinline__ LaneStack make_stack(unsigned long long* smem, const LaunchCfg& cfg, uint32_t levels = (uint32_t)kLdsStack) {
    LaneStack s;
    s.lds = (lds_u64*)smem + threadIdx.x;
    s.spill = cfg.spill;
    s.spill_stride = cfg.spill_stride;
    s.levels = levels;
    return s;
}

// wave-level sum -> one atomic per wave
__device__ __forceinline__ void wave_add(unsigned long long* dst, uint32_t v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, s);
}

// ------------------------------------------------------------------------------------------------
// parity seam: BVH::intersect_closest / intersect_any over a batch of caller rays
template <int PRIM, bool ANY, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_trace(DevScene S, const pt_ray* __restrict__ rays, uint64_t n, pt_hit* __restrict__ hits,
                                                       LaunchCfg cfg, unsigned long long* counters) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    LaneStack stk = make_stack(smem, cfg);
    uint32_t nn = 0, np = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kTraceBlock) {
        const pt_ray r = rays[i];
        float dist = r.tmax;
        Hit h;
        traverse<PRIM, ANY, COUNT>(S, V3{r.o[0], r.o[1], r.o[2]}, V3{r.d[0], r.d[1], r.d[2]}, dist, h, stk, nn, np);
        pt_hit out;
        out.prim = PT_MISS;
        out.t = dist;
        out.inside = 0;
        if (h.slot != PT_MISS) {
            out.prim = PRIM == PRIM_TRIANGLE ? __float_as_uint(S.prim[(size_t)h.slot * kTriStride + 2].y) : S.prim_meta[2 * h.slot];
            out.inside = h.inside;
        }
        hits[i] = out;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counters + (ANY ? CNT_RAYS_ANY : CNT_RAYS_CLOSEST), (unsigned long long)n);
    if (COUNT) {
        wave_add(counters + CNT_NODE_TESTS, nn);
        wave_add(counters + CNT_PRIM_TESTS, np);
    }
}

// ------------------------------------------------------------------------------------------------
// material / geometry of a hit (winning primitive only)
struct Surface {
    V3 normal;
    float area;
    uint32_t mat;
};
__device__ __forceinline__ V3 tri_normal_at(const DevScene& S, uint32_t slot, float u, float v) {
    if (kNrmStride == 16) {
        const float4* n = (const float4*)S.nrm + (size_t)slot * 4;
        const float4 a = n[0], b = n[1], c = n[2];
        return tri_normal(V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, V3{c.x, c.y, c.z}, u, v);
    }
    const float* n = S.nrm + (size_t)slot * 9;
    return tri_normal(ld3(n), ld3(n + 3), ld3(n + 6), u, v);
}

template <int PRIM>
__device__ __forceinline__ Surface surface_at(const DevScene& S, uint32_t slot, float u, float v, V3 o, V3 d, float t, bool want_area) {
    Surface s;
    if (PRIM == PRIM_TRIANGLE) {
        const float4* tp = S.prim + (size_t)slot * kTriStride;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        s.normal = tri_normal_at(S, slot, u, v);
        s.mat = __float_as_uint(c.z);
        s.area = want_area ? tri_area(V3{a.w, b.x, b.y}, V3{b.z, b.w, c.x}) : 0.0f;
    } else {
        const float4 sp = S.prim[slot];
        s.normal = sphere_normal(V3{sp.x, sp.y, sp.z}, o, d, t);
        s.mat = S.prim_meta[2 * slot + 1];
        s.area = want_area ? sphere_area(sp.w) : 0.0f;
    }
    return s;
}

__device__ __forceinline__ V3 sky_lookup(const DevScene& S, V3 d) {
    uint32_t idx = sky_index(d);
    if (idx > kSkyMaxIndex) return V3{0.f, 0.f, 0.f};
    const float* p = S.sky + (size_t)idx * 3;
    return V3{p[0], p[1], p[2]};
}

// Camera::generate, src/camera.rs:426-449
__device__ __forceinline__ void generate_ray(const DevCamera& cam, uint32_t x, uint32_t y, uint32_t sample, uint32_t k0, uint32_t k1,
                                             V3* o_out, V3* d_out) {
    const U4 r = philox4x32_10(k0, k1, y * cam.width + x, sample, 0u, 0u);
    const float r0 = closed01(r.v[0]);
    const float r1 = closed01(r.v[1]);
    const float r2 = closed01(r.v[2]) - 0.5f;
    const float r3 = closed01(r.v[3]) - 0.5f;
    const float u = ((float)x + r0) / (float)cam.width;
    const float v = ((float)y + r1) / (float)cam.height;
    const V3 p1 = ld3(cam.p1), p2 = ld3(cam.p2), p3 = ld3(cam.p3);
    const V3 target = (p1 + u * (p2 - p1)) + v * (p3 - p1);
    const V3 origin = ld3(cam.origin) + cam.lens_size * (r2 * ld3(cam.right) + r3 * ld3(cam.up));
    *o_out = origin;
    *d_out = normalize(target - origin);
}

__global__ void k_generate(const DevCamera* __restrict__ camp, const uint32_t* __restrict__ xy, uint64_t n, uint32_t sample, uint32_t k0,
                           uint32_t k1, pt_ray* __restrict__ rays) {
    const DevCamera cam = *camp;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        V3 o, d;
        generate_ray(cam, xy[2 * i], xy[2 * i + 1], sample, k0, k1, &o, &d);
        pt_ray r;
        r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z;
        r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
        r.tmax = kInf;
        rays[i] = r;
    }
}

__global__ void k_sky(DevScene S, const float* __restrict__ dirs, uint64_t n, float* __restrict__ rgb) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const V3 c = sky_lookup(S, ld3(dirs + 3 * i));
        rgb[3 * i] = c.x;
        rgb[3 * i + 1] = c.y;
        rgb[3 * i + 2] = c.z;
    }
}

// Camera::update + Camera::focus, src/camera.rs:214-292.  One lane walks the (at most five)
// deterministic autofocus segments with the same traversal the renderer uses.
template <int PRIM>
__global__ __launch_bounds__(kTraceBlock) void k_focus(DevScene S, DevCamera* camp, LaunchCfg cfg) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    LaneStack stk = make_stack(smem, cfg);
    DevCamera cam = *camp;
    const V3 origin = ld3(cam.origin), target = ld3(cam.target);
    const V3 direction = normalize(target - origin);
    const V3 right = cross(V3{0.f, 1.f, 0.f}, direction);
    const V3 up = cross(direction, right);
    const float aspect = (float)cam.width / (float)cam.height;

    V3 o = origin, d = direction;
    float distance = 0.0f;
    int depth = 5;
    float focus;
    uint32_t nn = 0, np = 0;
    while (true) {
        if (depth == 0) { focus = distance; break; }
        float dist = kInf;
        Hit h;
        traverse<PRIM, false, false>(S, o, d, dist, h, stk, nn, np);
        if (h.slot == PT_MISS) { focus = distance == 0.0f ? kInf : distance; break; }
        distance = distance + dist;
        const Surface sf = surface_at<PRIM>(S, h.slot, h.u, h.v, o, d, dist, false);
        const pt_material m = S.mats[sf.mat];
        const V3 P = o + dist * d;
        V3 nd;
        if (m.kind == PT_MAT_DIFFUSE && m.a > 0.5f) {
            nd = reflect(d, sf.normal);
        } else if (m.kind == PT_MAT_DIELECTRIC) {
            const bool inside = h.inside != 0;
            const V3 nf = inside ? -sf.normal : sf.normal;
            const float na = inside ? m.b : m.a, nb = inside ? m.a : m.b;
            V3 refr;
            if (refract(d, nf, na, nb, &refr)) nd = 0.5f < schlick(d, nf, na, nb) ? reflect(d, nf) : refr;
            else nd = reflect(d, nf);
        } else {
            focus = distance == 0.0f ? kInf : distance;
            break;
        }
        o = advance(P, nd);
        d = nd;
        depth -= 1;
    }
    const float fd = fmin_(20.0f, focus);
    const V3 c = origin + fd * direction;
    const V3 p1 = (c + (-0.5f * fd * aspect) * right) + (0.5f * fd) * up;
    const V3 p2 = (c + (0.5f * fd * aspect) * right) + (0.5f * fd) * up;
    const V3 p3 = (c + (-0.5f * fd * aspect) * right) + (-0.5f * fd) * up;
    cam.focal_distance = fd;
    cam.direction[0] = direction.x; cam.direction[1] = direction.y; cam.direction[2] = direction.z;
    cam.right[0] = right.x; cam.right[1] = right.y; cam.right[2] = right.z;
    cam.up[0] = up.x; cam.up[1] = up.y; cam.up[2] = up.z;
    cam.p1[0] = p1.x; cam.p1[1] = p1.y; cam.p1[2] = p1.z;
    cam.p2[0] = p2.x; cam.p2[1] = p2.y; cam.p2[2] = p2.z;
    cam.p3[0] = p3.x; cam.p3[1] = p3.y; cam.p3[2] = p3.z;
    *camp = cam;
}

// ------------------------------------------------------------------------------------------------
// wave64 stream compaction: every lane of the wave must call this (uniform control flow).
// One atomic per wave; lanes get consecutive places in `q` in lane order.
__device__ __forceinline__ void wave_enqueue(bool keep, uint32_t value, uint32_t* __restrict__ q, uint32_t* __restrict__ qn) {
    const unsigned long long m = __ballot(keep);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(qn, (uint32_t)__popcll(m));
    base = __shfl(base, 0, 64);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (keep) q[base + rank] = value;
}

constexpr uint32_t kPhaseBit = 1u << 16;
constexpr uint32_t kMatShift = 17;

__global__ __launch_bounds__(256) void k_raygen(const DevCamera* __restrict__ camp, PixelMap M, PathState P, uint32_t n_samples,
                                                uint32_t sample_base, uint32_t k0, uint32_t k1, int nee, int identity,
                                                uint32_t* __restrict__ queue, uint32_t* __restrict__ qn, unsigned long long* counters) {
    const DevCamera cam = *camp;
    const uint32_t total = n_samples * M.npix_pad;
    uint32_t made = 0;
    // identity: every slot maps to a pixel (no padding), so queue[slot] = slot needs no compaction
    if (identity && blockIdx.x == 0 && threadIdx.x == 0) *qn = total;
    for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += gridDim.x * blockDim.x) {
        const uint32_t slot = base + threadIdx.x;
        bool valid = false;
        if (slot < total) {
            const uint32_t s_local = slot / M.npix_pad, idx = slot - s_local * M.npix_pad;
            uint32_t x, y;
            valid = map_pixel(M, idx, &x, &y);
            if (valid) {
                V3 o, d;
                const uint32_t sample = sample_base + s_local;
                generate_ray(cam, x, y, sample, k0, k1, &o, &d);
                P.ro[slot] = make_float4(o.x, o.y, o.z, kInf);
                P.rd[slot] = make_float4(d.x, d.y, d.z, 0.0f);
                P.tp[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(0u));
                P.pxs[slot] = make_uint2(y * M.width + x, sample);
                if (nee) P.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                made += 1;
            }
        }
        if (identity) {
            if (valid) queue[slot] = slot;
        } else {
            wave_enqueue(valid, slot, queue, qn);
        }
    }
    wave_add(counters + CNT_PATHS, made);
}

// BVH::intersect_closest for every queued path
template <int PRIM, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_extend(DevScene S, PathState P, const uint32_t* __restrict__ queue,
                                                        const uint32_t* __restrict__ qn, LaunchCfg cfg, unsigned long long* counters) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    LaneStack stk = make_stack(smem, cfg);
    const uint32_t n = *qn;
    uint32_t nn = 0, np = 0, worst = 0;
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const uint32_t slot = queue[i];
        const float4 ro = P.ro[slot], rd = P.rd[slot];
        float dist = ro.w;
        Hit h;
        const uint32_t nn0 = nn;
        traverse<PRIM, false, COUNT>(S, xyz(ro), xyz(rd), dist, h, stk, nn, np);
        if (COUNT) worst = nn - nn0 > worst ? nn - nn0 : worst;
        const uint32_t bits = h.slot == PT_MISS ? PT_MISS : (h.slot | (h.inside << 31));
        P.hit[slot] = make_float4(__uint_as_float(bits), dist, h.u, h.v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(counters + CNT_RAYS_CLOSEST, (unsigned long long)n);
    if (COUNT) {
        wave_add(counters + CNT_NODE_TESTS, nn);
        wave_add(counters + CNT_PRIM_TESTS, np);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t other = __shfl_down(worst, off, 64);
            worst = other > worst ? other : worst;
        }
        if ((threadIdx.x & 63) == 0 && worst) atomicMax(counters + CNT_MAX_NODE_TESTS, (unsigned long long)worst);
    }
}

// ------------------------------------------------------------------------------------------------
// k_extend_pw — the production traversal kernel: persistent waves, per-lane ray refill, while-while.
//
// With one ray per lane and no refill, a wave lives as long as its longest ray (60-100 loop
// iterations for a mean of ~10), so most lanes idle.  Here every lane owns a small traversal state
// machine; when at least kRefillLanes lanes of the wave are idle they take the next rays from the
// queue with one wave-aggregated atomic.  Between refills the wave takes traversal steps, each of
// the kind more of its lanes wait for: an internal-node step or one leaf primitive (wave_step).
// A lane only ever waits — its own sequence of node visits, primitive tests and ray.distance
// updates is exactly the one of traverse<>, so the results are identical.
struct LaneTrav {
    V3 o, d, rcp;
    float dist;
    uint32_t cur, sp;
    uint32_t hslot;   // winning primitive slot | Intersection::inside << 31 (slots have 26 bits), PT_MISS if none
    float hu, hv;
    uint32_t slot;  // path slot the ray belongs to
};

// pop the next node still reachable with the current ray.distance (bvh.rs:194-196); false = stack empty
__device__ __forceinline__ bool lane_pop(LaneTrav& T, const LaneStack& stk) {
    while (T.sp != 0) {
        T.sp -= 1;
        uint32_t ref;
        float tpush;
        stk.get(T.sp, &ref, &tpush);
        if (!(T.dist < tpush)) {
            T.cur = ref;
            return true;
        }
    }
    return false;
}

#ifndef PT_REFILL_LANES
#define PT_REFILL_LANES 20
#endif
#ifndef PT_SHADE_LANES
#define PT_SHADE_LANES 16
#endif
constexpr int kRefillLanes = PT_REFILL_LANES;  // refill when at least this many lanes of the wave are idle
#ifndef PT_FETCH_CHUNK
#define PT_FETCH_CHUNK 256
#endif
constexpr int kFetchChunk = PT_FETCH_CHUNK;               // rays a wave reserves per atomic on the queue cursor
constexpr int kShadeLanes = PT_SHADE_LANES;    // fused kernels: shade when at least this many lanes wait for it
#ifndef PT_STEP_BIAS_NUM
#define PT_STEP_BIAS_NUM 1
#endif
#ifndef PT_STEP_BIAS_DEN
#define PT_STEP_BIAS_DEN 2
#endif
#ifndef PT_PATH_REFILL_LANES
#define PT_PATH_REFILL_LANES 8
#endif
#ifndef PT_PATH_SHADE_LANES
#define PT_PATH_SHADE_LANES 24
#endif
constexpr int kPathRefillLanes = PT_PATH_REFILL_LANES, kPathShadeLanes = PT_PATH_SHADE_LANES;  // k_path's own thresholds (swept on C3)
#ifndef PT_INT_REPEAT
#define PT_INT_REPEAT 3
#endif
constexpr int kInternalRepeat = PT_INT_REPEAT;  // persistent kernels: internal-node steps a lane may take per turn (most nodes lead to another one)
constexpr int kStepBiasNum = PT_STEP_BIAS_NUM, kStepBiasDen = PT_STEP_BIAS_DEN;  // persistent kernels: internal step when n_int * NUM >= n_leaf * DEN

// load a ray into a lane and do the root test; false = the ray misses the whole scene
template <bool COUNT>
__device__ __forceinline__ bool lane_start_ray(const DevScene& S, LaneTrav& T, float4 ro, float4 rd, uint32_t& nn, uint32_t& mine) {
    T.o = xyz(ro);
    T.d = xyz(rd);
    T.rcp = V3{1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z};
    T.dist = ro.w;
    T.hslot = PT_MISS;
    T.hu = T.hv = 0.f;
    T.sp = 0;
    float tmin;
    if (COUNT) { nn += 1; mine = 1; }
    if (slab(S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1], S.root_max[2], T.o, T.rcp, T.dist, &tmin)) {
        T.cur = S.root_ref;
        return true;
    }
    return false;
}

// Diagnostics (-DPT_PHASE_STATS builds only): per phase, wave-level executions, lanes that took part, clocks.
struct PhaseStats {
    unsigned long long exec[6], lanes[6], clk[6];
    __device__ __forceinline__ void init() { for (int i = 0; i < 6; ++i) exec[i] = lanes[i] = clk[i] = 0; }
    __device__ __forceinline__ void add(int p, unsigned long long mask, unsigned long long t0) {
#ifdef PT_PHASE_STATS
        exec[p] += 1;
        lanes[p] += (unsigned long long)__popcll(mask);
        clk[p] += __builtin_amdgcn_s_memtime() - t0;
#endif
    }
    __device__ __forceinline__ unsigned long long now() const {
#ifdef PT_PHASE_STATS
        return __builtin_amdgcn_s_memtime();
#else
        return 0;
#endif
    }
    __device__ __forceinline__ void flush(unsigned long long* counters) const {
#ifdef PT_PHASE_STATS
        if ((threadIdx.x & 63u) == 0)
            for (int i = 0; i < 6; ++i) {
                atomicAdd(counters + CNT_PHASE + 3 * i, exec[i]);
                atomicAdd(counters + CNT_PHASE + 3 * i + 1, lanes[i]);
                atomicAdd(counters + CNT_PHASE + 3 * i + 2, clk[i]);
            }
#endif
    }
};

// one internal-node step of one lane (bvh.rs:198-222) on its node's record: test both children, descend into the nearer, push the farther
template <bool COUNT>
__device__ __forceinline__ void step_internal_on(float4 a, float4 b, float4 c, float4 e, LaneTrav& T, bool& active, const LaneStack& stk, uint32_t& nn,
                                                 uint32_t& mine) {
    float tl, tr;
    const bool hl = slab(a.x, a.y, a.z, b.x, b.y, b.z, T.o, T.rcp, T.dist, &tl);
    const bool hr = slab(c.x, c.y, c.z, e.x, e.y, e.z, T.o, T.rcp, T.dist, &tr);
    if (COUNT) { nn += 2; mine += 2; }
    const uint32_t lref = __float_as_uint(a.w), rref = __float_as_uint(c.w);
    if (hl && hr) {
        const bool left_first = tl <= tr;  // bvh.rs:211
        stk.put(T.sp, left_first ? rref : lref, left_first ? tr : tl);
        T.sp += 1;
        T.cur = left_first ? lref : rref;
    } else if (hl) {
        T.cur = lref;
    } else if (hr) {
        T.cur = rref;
    } else if (!lane_pop(T, stk)) {
        active = false;
    }
}
template <bool COUNT>
__device__ __forceinline__ void step_internal(const DevScene& S, LaneTrav& T, bool& active, const LaneStack& stk, uint32_t& nn, uint32_t& mine) {
#if PT_BUFFER_LOADS
    const __amdgpu_buffer_rsrc_t rp = record_rsrc(S.pairs, S.n_pairs + 1u);
    const float4 a = record_quad(rp, T.cur, 0), b = record_quad(rp, T.cur, 1), c = record_quad(rp, T.cur, 2), e = record_quad(rp, T.cur, 3);
#else
    const float4* rec = S.pairs + (size_t)T.cur * 4;
    const float4 a = rec[0], b = rec[1], c = rec[2], e = rec[3];
#endif
    step_internal_on<COUNT>(a, b, c, e, T, active, stk, nn, mine);
}

#ifndef PT_DRAIN_GHOST
#define PT_DRAIN_GHOST 0  // measured and not adopted (profiles/r3_ab_drained_read_ahead.txt): see below
#endif
// The internal step of a wave that is only emptying (k_path once the batch has no more paths to hand out): a handful of
// long paths in a wave of idle lanes, every step a dependent fetch that pays a whole round trip to memory.  The pair records
// are in split-event order, i.e. the tree in DFS pre-order (pt_bvh.h, PT_PAIR_DFS): behind a node's record come its left
// child's, that one's children ... and near the leaves the whole subtree.  So the lanes that have nothing to do read ahead:
// each takes the nearest traversing lane below it and fetches the record as many places behind that lane's as it is lanes
// away, in the same load instruction as the real fetches.  Nothing is done with what they fetch — it is a prefetch into L1 /
// L2, the visiting order is untouched — but the next steps of the traversal find their records there.
// Result (round 3, A/B builds): nothing.  A blocking 1-spp call of C3 takes 9.5 ms with and without it, C4 11.3, C5 16.9, and
// the 256-spp rates are equal: a wave that is emptying does not wait for pair records that miss (its few paths stay inside a
// small part of the tree, which L2 holds); its ~2.6 k clocks per turn are the ~500 dependent instructions of a turn issued
// by one wave alone on its SIMD, plus cache-hit latencies.  Kept as a build option for the record.
template <bool COUNT>
__device__ __forceinline__ void step_internal_readahead(const DevScene& S, LaneTrav& T, bool& active, bool at_internal, unsigned long long mi,
                                                        const LaneStack& stk, uint32_t& nn, uint32_t& mine) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = mi & ((2ull << lane) - 1ull);  // traversing lanes at or below this one
    const uint32_t owner = below ? 63u - (uint32_t)__clzll((long long)below) : 63u - (uint32_t)__clzll((long long)mi);
    const uint32_t ahead = (lane - owner) & 63u;
    const uint32_t cur_owner = (uint32_t)__shfl((int)T.cur, (int)owner, 64);
    uint32_t idx = at_internal ? T.cur : cur_owner + ahead;
    idx = idx < S.n_pairs ? idx : S.n_pairs - 1u;
    const float4* rec = S.pairs + (size_t)idx * 4;
    const float4 a = rec[0], b = rec[1], c = rec[2], e = rec[3];
    if (at_internal) step_internal_on<COUNT>(a, b, c, e, T, active, stk, nn, mine);
}

// one primitive of a leaf against the lane's ray; an accepted hit replaces the previous one (last accepted wins)
template <int PRIM>
__device__ __forceinline__ void test_prim(const DevScene& S, LaneTrav& T, uint32_t ps) {
    if (PRIM == PRIM_TRIANGLE) {
#if PT_BUFFER_LOADS
        static_assert(kTriStride == 4, "buffer loads address 64-byte triangle slots");
        const __amdgpu_buffer_rsrc_t rt = record_rsrc(S.prim, S.n_prims);
        const float4 a = record_quad(rt, ps, 0), b = record_quad(rt, ps, 1), c = record_quad(rt, ps, 2);
#else
        const float4* tp = S.prim + (size_t)ps * kTriStride;
        const float4 a = tp[0], b = tp[1], c = tp[2];
#endif
        float t, u, v, det;
        if (tri_hit(V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, V3{b.z, b.w, c.x}, T.o, T.d, &t, &u, &v, &det)) {
            T.dist = t;  // unconditionally, triangle.rs:72
            T.hslot = ps | (det < 0.0f ? 0x80000000u : 0u);
            T.hu = u;
            T.hv = v;
        }
    } else {
        const float4 sp4 = S.prim[ps];
        float t;
        bool inside;
        if (sphere_hit(V3{sp4.x, sp4.y, sp4.z}, sp4.w, T.o, T.d, T.dist, &t, &inside)) {
            T.dist = t;
            T.hslot = ps | (inside ? 0x80000000u : 0u);
        }
    }
}

// One primitive step of a lane that stands at a leaf: test the leaf's next (up to two) primitives; the leaf
// reference itself is the cursor (first += 2, count -= 2), the lane pops after the last one.
// Leaves of more than 31 primitives (count field 0) are rare and done in one go.
template <int PRIM, bool COUNT, bool ESCAPE = true>
__device__ __forceinline__ void step_prim(const DevScene& S, LaneTrav& T, bool& active, const LaneStack& stk, uint32_t& np) {
    const uint32_t first = T.cur & kLeafFirstMask;
    const uint32_t count = (T.cur >> kLeafCountShift) & 31u;
    bool done;
    if (ESCAPE && count == 0) {  // ESCAPE = false: the caller takes the escape-encoded leaves itself (step_prim_wave)
        const uint32_t bf = S.big_leaf[2 * first], bc = S.big_leaf[2 * first + 1];
        if (COUNT) np += bc;
        for (uint32_t k = 0; k < bc; ++k) test_prim<PRIM>(S, T, bf + k);
        done = true;
    } else {
        // up to two primitives per step: the builder stops splitting at two (bvh.rs:56), so most leaves are
        // done in one step, with both records requested before the first test (C3: 9.9 -> 10.3 Grays/s)
        if (COUNT) np += count >= 2 ? 2u : 1u;
        test_prim<PRIM>(S, T, first);
        if (count >= 2) test_prim<PRIM>(S, T, first + 1);
        done = count <= 2;
        T.cur = T.cur + 2u - (2u << kLeafCountShift);
    }
    if (done && !lane_pop(T, stk)) active = false;
}

#ifndef PT_COOP_LANES_PER_PRIM
#define PT_COOP_LANES_PER_PRIM 2
#endif
constexpr int kCoopLanesPerPrim = PT_COOP_LANES_PER_PRIM;  // a large leaf is taken cooperatively when lanes-at-large-leaves * this <= its primitives

__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

// The primitive step of a whole wave (every lane of the wave must call it; at_leaf says which lanes stand at a leaf).
// Lanes at small leaves take the ordinary step above.  A lane at a LARGE leaf — the builder does not split
// primitives whose centres coincide (bvh.rs:56,158-162: fans around one vertex, stacked copies), so leaves
// can hold dozens — would keep the other 63 lanes waiting for dozens of sequential tests.  Triangle::intersect
// does not look at ray.distance (triangle.rs:41-72), so which primitive of a leaf wins does not depend on the order
// of evaluation: it is the LAST one in leaf order that accepts the ray, with that primitive's t.  The wave
// therefore takes the large leaves one at a time: the owner's ray is broadcast, lane k tests primitive k, and the
// highest accepting lane is the winner — the same hit, t, u, v and inside flag as the sequential loop, in one
// test's time.  (Spheres do compare with ray.distance, sphere.rs:54,66: they stay sequential.)
template <int PRIM, bool COUNT>
__device__ __forceinline__ void step_prim_wave(const DevScene& S, LaneTrav& T, bool& active, bool at_leaf, const LaneStack& stk, uint32_t& np) {
    if (PRIM != PRIM_TRIANGLE || kCoopLeafFrom == 0) {  // PT_COOP_LEAF_FROM=0: A/B builds without the cooperative path
        if (at_leaf) step_prim<PRIM, COUNT>(S, T, active, stk, np);
        return;
    }
    // large leaves carry the escape encoding (count field 0; the host marks them when it flattens the tree)
    const bool esc = at_leaf && ((T.cur >> kLeafCountShift) & 31u) == 0u;
    if (at_leaf && !esc) step_prim<PRIM, COUNT, false>(S, T, active, stk, np);
    const unsigned long long escm = __ballot(esc);
    if (escm == 0) return;
    uint32_t first = 0, count = 0;
    if (esc) {
        const uint32_t e = T.cur & kLeafFirstMask;
        first = S.big_leaf[2 * e];
        count = S.big_leaf[2 * e + 1];
    }
    // One cooperative pass per lane costs about two ordinary steps.  When many lanes of the wave stand at large
    // leaves at once (coherent rays reach the same fan together), the ordinary lock-step steps of two primitives are
    // cheaper for the medium-sized ones: such a lane gets its leaf back as an ordinary reference and takes part in the
    // next leaf turns.  Leaves above 31 primitives cannot be written that way and are always taken cooperatively.
    const uint32_t n_esc = (uint32_t)__popcll(escm);
    const bool coop = esc && (count > 31u || n_esc * (uint32_t)kCoopLanesPerPrim <= count);
    if (esc && !coop) T.cur = kLeafBit | (count << kLeafCountShift) | first;
    unsigned long long todo = __ballot(coop);
    const uint32_t lane = threadIdx.x & 63u;
    while (todo) {
        const int src = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)todo) - 1);
        todo &= todo - 1ull;
        const uint32_t f = lane_bcast(first, src), n = lane_bcast(count, src);
        const V3 o = V3{lane_bcast(T.o.x, src), lane_bcast(T.o.y, src), lane_bcast(T.o.z, src)};
        const V3 d = V3{lane_bcast(T.d.x, src), lane_bcast(T.d.y, src), lane_bcast(T.d.z, src)};
        uint32_t win = PT_MISS;  // wave-uniform: the last accepting primitive so far, with its hit values
        float wt = 0.f, wu = 0.f, wv = 0.f, wdet = 0.f;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t k = base + lane;
            float t = 0.f, u = 0.f, v = 0.f, det = 0.f;
            bool hit = false;
            if (k < n) {
#if PT_BUFFER_LOADS
                const __amdgpu_buffer_rsrc_t rt = record_rsrc(S.prim, S.n_prims);
                const float4 a = record_quad(rt, f + k, 0), b = record_quad(rt, f + k, 1), c = record_quad(rt, f + k, 2);
#else
                const float4* tp = S.prim + (size_t)(f + k) * kTriStride;
                const float4 a = tp[0], b = tp[1], c = tp[2];
#endif
                hit = tri_hit(V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, V3{b.z, b.w, c.x}, o, d, &t, &u, &v, &det);
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                const int w = __builtin_amdgcn_readfirstlane(63 - (int)__clzll((long long)m));
                win = f + base + (uint32_t)w;
                wt = lane_bcast(t, w);
                wu = lane_bcast(u, w);
                wv = lane_bcast(v, w);
                wdet = lane_bcast(det, w);
            }
        }
        if ((int)lane == src) {
            if (COUNT) np += n;
            if (win != PT_MISS) {
                T.dist = wt;  // unconditionally, triangle.rs:72
                T.hslot = win | (wdet < 0.0f ? 0x80000000u : 0u);
                T.hu = wu;
                T.hv = wv;
            }
            if (!lane_pop(T, stk)) active = false;
        }
    }
}

// One traversal step of a wave in the persistent kernels: the wave does whichever kind of step
// (internal node / one leaf primitive) more of its traversing lanes are waiting for.
template <int PRIM, bool COUNT>
__device__ __forceinline__ void wave_step(const DevScene& S, LaneTrav& T, bool& active, const LaneStack& stk, unsigned long long actv,
                                          uint32_t& nn, uint32_t& np, uint32_t& mine) {
    const bool at_internal = active && !(T.cur & kLeafBit);
    const bool at_leaf = active && !at_internal;
    const int n_int = __popcll(__ballot(at_internal)), n_leaf = __popcll(actv) - n_int;
    if (n_int * kStepBiasNum >= n_leaf * kStepBiasDen) {
        if (at_internal) {
            step_internal<COUNT>(S, T, active, stk, nn, mine);
#pragma unroll
            for (int rep = 1; rep < kInternalRepeat; ++rep)
                if (active && !(T.cur & kLeafBit)) step_internal<COUNT>(S, T, active, stk, nn, mine);
        }
    } else {
        step_prim_wave<PRIM, COUNT>(S, T, active, at_leaf, stk, np);
    }
}

__device__ __forceinline__ void lane_store_hit(const PathState& P, const LaneTrav& T) {
    const uint32_t bits = T.hslot;
    P.hit[T.slot] = make_float4(__uint_as_float(bits), T.dist, T.hu, T.hv);
}

// wave-uniform: make sure this wave owns a piece of the queue; returns false once the queue is used up
__device__ __forceinline__ bool wave_fetch_chunk(uint32_t* cursor, uint32_t n, uint32_t chunk, uint32_t& chunk_next, uint32_t& chunk_end) {
    if (chunk_next < chunk_end) return true;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0) base = atomicAdd(cursor, chunk);
    base = __builtin_amdgcn_readfirstlane(__shfl(base, 0, 64));
    if (base >= n) {
        chunk_next = chunk_end = 0;
        return false;
    }
    chunk_next = base;
    chunk_end = base + chunk < n ? base + chunk : n;
    return true;
}

// XCD-aware variant for k_path: the slot range is cut into 8 * strips strips; strip s belongs to the
// region of XCD s % 8.  A wave draws chunks from the region of the XCD it runs on, so that an XCD's own
// L2 sees one part of the scene (C3: 11.0 -> 11.4 Grays/s), and moves on to the next region when
// its own is used up.  cursor[0..7] count the chunks handed out per region.
__device__ __forceinline__ bool wave_fetch_chunk_xcd(uint32_t* cursor, uint32_t n, uint32_t chunk, uint32_t strips, uint32_t& region, uint32_t& tried,
                                                     uint32_t& chunk_next, uint32_t& chunk_end) {
    if (chunk_next < chunk_end) return true;
    const uint32_t n_chunks = (n + chunk - 1u) / chunk;
    const uint32_t per_strip = (n_chunks + 8u * strips - 1u) / (8u * strips);  // chunks per strip
    while (tried < 8u) {
        uint32_t c = 0;
        if ((threadIdx.x & 63u) == 0) c = atomicAdd(cursor + region, 1u);
        c = __builtin_amdgcn_readfirstlane(__shfl(c, 0, 64));
        const uint32_t strip = (c / per_strip) * 8u + region;   // the region's strips are 8 apart
        const uint32_t g = strip * per_strip + c % per_strip;   // global chunk index
        if (c < per_strip * strips && g < n_chunks) {
            chunk_next = g * chunk;
            chunk_end = chunk_next + chunk < n ? chunk_next + chunk : n;
            return true;
        }
        // g grows with c, so this region is used up: go on with the next one
        region = (region + 1u) & 7u;
        tried += 1u;
    }
    chunk_next = chunk_end = 0;
    return false;
}

__device__ __forceinline__ void reduce_worst(unsigned long long* counters, uint32_t worst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t other = __shfl_down(worst, off, 64);
        worst = other > worst ? other : worst;
    }
    if ((threadIdx.x & 63) == 0 && worst) atomicMax(counters + CNT_MAX_NODE_TESTS, (unsigned long long)worst);
}

template <int PRIM, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_extend_pw(DevScene S, PathState P, const uint32_t* __restrict__ queue,
                                                           const uint32_t* __restrict__ qn, uint32_t* __restrict__ cursor, LaunchCfg cfg,
                                                           unsigned long long* counters) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    LaneStack stk = make_stack(smem, cfg);
    const uint32_t n = *qn;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nn = 0, np = 0, worst = 0, mine = 0;
    LaneTrav T;
    T.cur = 0; T.sp = 0; T.dist = 0.f; T.hslot = PT_MISS; T.hu = T.hv = 0.f; T.slot = 0;
    T.o = T.d = T.rcp = V3{0.f, 0.f, 0.f};
    bool active = false;
    bool drained = false;  // wave-uniform: the queue has no more rays for this wave
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform: this wave's current piece of the queue
    // rays reserved per atomic: large queues take kFetchChunk at a time (few atomics on one address),
    // small ones are spread over all waves of the grid
    const uint32_t per_wave = n / (gridDim.x * (kTraceBlock / 64)) + 1;
    const uint32_t chunk = per_wave >= (uint32_t)kFetchChunk ? (uint32_t)kFetchChunk : (per_wave < 32u ? 32u : per_wave);
    while (true) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (!drained && n_idle >= kRefillLanes) {
            if (!wave_fetch_chunk(cursor, n, chunk, chunk_next, chunk_end)) drained = true;
            if (!active && !drained) {
                const uint32_t idx = chunk_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (idx < chunk_end) {
                    T.slot = queue[idx];
                    active = lane_start_ray<COUNT>(S, T, P.ro[T.slot], P.rd[T.slot], nn, mine);
                    if (!active) lane_store_hit(P, T);  // misses the root box: done at once
                }
            }
            chunk_next = chunk_next + (uint32_t)n_idle < chunk_end ? chunk_next + (uint32_t)n_idle : chunk_end;
        }
        const unsigned long long act = __ballot(active);
        if (act == 0) {
            if (drained) break;
            continue;
        }
        // traversal steps (inner hot loop) until enough lanes are free for a refill
        while (true) {
            const unsigned long long actv = __ballot(active);
            if (actv == 0 || (!drained && 64 - __popcll(actv) >= kRefillLanes)) break;
            wave_step<PRIM, COUNT>(S, T, active, stk, actv, nn, np, mine);
            // lanes whose stack ran empty in this step are done: store the hit record (once)
            if (!active && ((actv >> lane) & 1ull)) {
                lane_store_hit(P, T);
                if (COUNT) worst = mine > worst ? mine : worst;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(counters + CNT_RAYS_CLOSEST, (unsigned long long)n);
    if (COUNT) {
        wave_add(counters + CNT_NODE_TESTS, nn);
        wave_add(counters + CNT_PRIM_TESTS, np);
        reduce_worst(counters, worst);
    }
}

template <int PRIM>
__device__ __forceinline__ uint32_t prim_material(const DevScene& S, uint32_t slot) {
    return PRIM == PRIM_TRIANGLE ? __float_as_uint(S.prim[(size_t)slot * kTriStride + 2].z) : S.prim_meta[2 * slot + 1];
}

// The state of one path between two traces — what Camera::sample keeps in locals
// (src/camera.rs:296-298) plus the ray, whether it lives in registers (k_path) or in HBM (k_shade).
struct PathVals {
    V3 o, d;          // the ray that was traced / the next ray to trace
    float tmax;       // ray.distance to start the next trace with
    float cos_i;      // cos_intersection saved across the look-ahead trace (camera.rs:358)
    V3 transport;
    uint32_t flags;   // iteration | phase << 16 | material << 17
    V3 acc;           // accumalated_color (only non-zero with NEE)
    uint32_t pixel, sample;  // RNG counter words
};
struct HitVals {
    uint32_t bits;    // slot | inside << 31, or PT_MISS
    float t, u, v;    // ray.distance after the trace, barycentrics
};

// One step of the loop body of Camera::sample (src/camera.rs:299-421) for one path.
// Returns SHADE_ALIVE when the path stays alive (pv then holds its next ray), SHADE_END when it ended (radiance then
// holds what Camera::sample returns), and — kernels without the NEE code only — SHADE_REDO when the path met a diffuse
// hit at which the NEE block of camera.rs:314-338 may not be left out: the caller puts its slot on the redo list and
// k_redo runs the whole path again with the NEE code.
//
// When the block may be left out.  Its contribution is min(1 - speculaty, 0) * n_lights * transport (*) (cos / mis) * light
// colour (*) brdf (camera.rs:333).  With the weight exactly 0 that is 0 * X, and acc + (+-0) == acc bit for bit — provided X
// is finite; 0 * inf and 0 * NaN are NaN, and the reference then returns a NaN pixel.  cos / mis <= pi (both pdfs are >= 0
// once cos_i > 0 and cos_l > 0 hold and the light's area is a finite number >= 0), colour (*) brdf is bounded by a per-scene
// constant, so X is finite whenever |transport| <= nee_thr (pt_api.cpp: FLT_MAX / (3.25 * largest |light colour * brdf|)).
// A transport beyond that (inf and NaN included: the unguarded light pdf of camera.rs:361-370 makes some) runs the block,
// and so does every diffuse hit of a scene whose lights have a non-finite area or colour (nee_literal).
// What is left out of this argument: cos / mis is also non-finite when mis == 0, i.e. the light pdf is 0 (the sampled point
// exactly f32::EPSILON from the surface) AND cos_i / pi underflows to 0; no input is known to reach it.
enum { SHADE_END = 0, SHADE_ALIVE = 1, SHADE_REDO = 2 };
// the work of the NEE blocks a shade_core call ran, for the counters of instrumented launches
struct NeeWork {
    uint32_t blocks;            // one Primitive::intersect on the light itself per block (camera.rs:321)
    uint32_t node_tests, prim_tests;  // of its intersect_any traversal
};
template <int PRIM, bool NEE>
__device__ __forceinline__ int shade_core(const DevScene& S, PathVals& pv, const HitVals& hv, uint32_t depth, uint32_t k0, uint32_t k1,
                                          LaneStack stk, unsigned long long* counters, V3& radiance, NeeWork& nee_work) {
    const V3 o = pv.o, d = pv.d;
    V3 transport = pv.transport;
    const uint32_t flags = pv.flags;
    uint32_t it = flags & 0xffffu;
    const uint32_t hbits = hv.bits;
    const bool is_hit = hbits != PT_MISS;
    const uint32_t hslot = hbits & 0x7fffffffu;
    const bool inside = (hbits >> 31) != 0;
    const float t = hv.t;  // ray.distance after the trace
    V3 acc = V3{0.f, 0.f, 0.f};
    if (NEE) acc = pv.acc;
    const uint2 px = make_uint2(pv.pixel, pv.sample);

    if (flags & kPhaseBit) {
        // second half of the diffuse branch, camera.rs:361-372: the look-ahead trace has returned
        const float cos_i = pv.cos_i;
        float light_pdf = 0.0f;
        if (is_hit) {
            const uint32_t lm = prim_material<PRIM>(S, hslot);
            if (S.mats[lm].kind == PT_MAT_EMISSIVE) {
                const Surface sf = surface_at<PRIM>(S, hslot, hv.u, hv.v, o, d, t, true);
                const float cos_l = -dot(sf.normal, d);
                const float solid = (cos_l * sf.area) / (t * t);
                light_pdf = 1.0f / solid;
            }
        }
        const pt_material m = S.mats[flags >> kMatShift];
        const V3 brdf = kFrac1Pi * V3{m.rgb[0], m.rgb[1], m.rgb[2]};
        const float hemi_pdf = kFrac1Pi * cos_i;
        const float mis = light_pdf + hemi_pdf;
        transport = mul_ew(transport, (cos_i / mis) * brdf);
        it += 1;
        if (it >= depth) { radiance = acc; return SHADE_END; }
        // Opt-in (PT_OPT_SKIP_REPEATED_MISS, never the default and never in a metric run): the reference now traces this very
        // ray again (quirk Q3: the look-ahead hit is dropped, camera.rs:298-300 against :362).  If the look-ahead trace missed,
        // ray.distance is still inf and the second traversal is the first one repeated — it misses again and the path ends
        // in the sky (camera.rs:301-304).  Taking that result at once gives the same sums with one traversal less per such
        // bounce; the executed-ray count then differs from the reference's by exactly those repeated misses.
        if (S.skip_repeated_miss != 0u && !is_hit) {
            radiance = acc + mul_ew(transport, 0.1f * sky_lookup(S, d));
            return SHADE_END;
        }
        // the next iteration re-traces this ray with the shortened ray.distance (quirk Q3)
        pv.tmax = t;
        pv.transport = transport;
        pv.flags = it;
        return SHADE_ALIVE;
    }

    if (!is_hit) {  // camera.rs:301-304
        radiance = acc + mul_ew(transport, 0.1f * sky_lookup(S, d));
        return SHADE_END;
    }
    const V3 Pt = o + t * d;  // ray.intersection(), ray.rs:33-35
    const Surface sf = surface_at<PRIM>(S, hslot, hv.u, hv.v, o, d, t, false);
    const V3 normal = sf.normal;
    const pt_material m = S.mats[sf.mat];
    const V3 color = V3{m.rgb[0], m.rgb[1], m.rgb[2]};
    V3 nd;

    if (m.kind == PT_MAT_EMISSIVE) {  // camera.rs:308-311
        radiance = acc + mul_ew(transport, color);
        return SHADE_END;
    } else if (m.kind == PT_MAT_DIFFUSE) {  // camera.rs:312-374
        const float speculaty = m.a;
        if (inside) { radiance = acc; return SHADE_END; }
        // a transport the block's product may not be finite with (NaN fails the comparisons too)
        const bool beyond = S.n_lights != 0 && !(__builtin_fabsf(transport.x) <= S.nee_thr && __builtin_fabsf(transport.y) <= S.nee_thr &&
                                                 __builtin_fabsf(transport.z) <= S.nee_thr);
        if (!NEE && beyond) return SHADE_REDO;
        if (NEE) {
            // camera.rs:314-338.  Its weight min(1 - speculaty, 0) is exactly 0 unless speculaty > 1; the block is left out
            // only where that provably adds nothing (see above).
            const float nee_w = fmin_(1.0f - speculaty, 0.0f);
            if (S.n_lights != 0 && (nee_w != 0.0f || S.nee_literal != 0u || beyond)) {
                nee_work.blocks += 1;
                const U4 rn = philox4x32_10(k0, k1, px.x, px.y, 1u + it, 1u);
                // BVH::random_light, bvh.rs:273-284: Range<usize>::ind_sample rejects draws at or above the largest
                // multiple of the range and returns draw % range (rand 0.3.15); rejected draws are replaced by
                // word 0 of the blocks (iteration, stream 2, 3, ...) of the counter-based generator
                uint32_t pick = rn.v[0];
                const uint32_t zone = 0xffffffffu - 0xffffffffu % S.n_lights;
                for (uint32_t j = 0; pick >= zone; ++j) pick = philox4x32_10(k0, k1, px.x, px.y, 1u + it, 2u + j).v[0];
                const uint32_t lslot = S.lights[pick % S.n_lights];
                V3 Q, ln;
                float larea, gdist = kInf;
                bool lhit;
                if (PRIM == PRIM_TRIANGLE) {
                    const float4* tpp = S.prim + (size_t)lslot * kTriStride;
                    const float4 a = tpp[0], b = tpp[1], c = tpp[2];
                    const V3 p0 = V3{a.x, a.y, a.z}, e1 = V3{a.w, b.x, b.y}, e2 = V3{b.z, b.w, c.x};
                    const float u = open01(rn.v[1]);
                    float v = open01(rn.v[2]);
                    v = (1.0f - u) * v;
                    Q = (p0 + u * e1) + v * e2;  // triangle.rs:97-104
                    const V3 ldir = normalize(Q - Pt);
                    const V3 go = Pt + (20.0f * kEps) * ldir;
                    float tt, uu, vv, det;
                    lhit = tri_hit(p0, e1, e2, go, ldir, &tt, &uu, &vv, &det);
                    if (lhit) {
                        gdist = tt;
                        ln = tri_normal_at(S, lslot, uu, vv);
                        larea = tri_area(e1, e2);
                    }
                } else {
                    // sphere.rs:98-118 (volume sample, x and y both use cos(phi))
                    const float4 sp = S.prim[lslot];
                    const float phi = 0.0f + (2.0f * kPi - 0.0f) * half_open01(rn.v[1]);
                    const float cos_theta = -1.0f + (1.0f - -1.0f) * half_open01(rn.v[2]);
                    const float uu = 0.0f + (1.0f - 0.0f) * half_open01(rn.v[3]);
                    const float theta = pt_acosf(cos_theta);
                    const float rr = sp.w * pt_cbrtf(uu);
                    float st, ct, sph, cph;
                    pt_sincosf(theta, &st, &ct);
                    pt_sincosf(phi, &sph, &cph);
                    Q = V3{rr * st * cph + sp.x, rr * st * cph + sp.y, rr * ct + sp.z};
                    const V3 ldir = normalize(Q - Pt);
                    const V3 go = Pt + (20.0f * kEps) * ldir;
                    float tt;
                    bool lin;
                    lhit = sphere_hit(V3{sp.x, sp.y, sp.z}, sp.w, go, ldir, gdist, &tt, &lin);
                    if (lhit) {
                        gdist = tt;
                        ln = sphere_normal(V3{sp.x, sp.y, sp.z}, go, ldir, tt);
                        larea = sphere_area(sp.w);
                    }
                }
                if (lhit) {
                    const V3 ldir = normalize(Q - Pt);
                    const V3 go = Pt + (20.0f * kEps) * ldir;
                    const float cos_i = dot(normal, ldir);
                    const float cos_l = -dot(ln, ldir);
                    if (cos_i > 0.0f && cos_l > 0.0f) {
                        gdist -= kEps;
                        const float gd = gdist;
                        Hit sh;
                        float dist2 = gdist;
                        traverse<PRIM, true, true>(S, go, ldir, dist2, sh, stk, nee_work.node_tests, nee_work.prim_tests);
                        atomicAdd(counters + CNT_RAYS_ANY, 1ull);
                        if (sh.slot == PT_MISS) {
                            const V3 brdf = kFrac1Pi * color;
                            const pt_material lmat = S.mats[prim_material<PRIM>(S, lslot)];
                            const V3 lc = V3{lmat.rgb[0], lmat.rgb[1], lmat.rgb[2]};
                            const float solid = (cos_l * larea) / (gd * gd);
                            const float light_pdf = 1.0f / solid;
                            const float hemi_pdf = kFrac1Pi * cos_i;
                            const float mis = light_pdf + hemi_pdf;
                            const V3 nee = (nee_w * (float)S.n_lights) * mul_ew(transport, (cos_i / mis) * mul_ew(lc, brdf));
                            acc = acc + nee;
                        }
                    }
                }
            }
        }
        const U4 r = philox4x32_10(k0, k1, px.x, px.y, 1u + it, 0u);
        if (closed01(r.v[0]) < speculaty) {  // camera.rs:341-345
            nd = reflect(d, normal);
            transport = mul_ew(transport, color);
        } else {  // camera.rs:346-373 up to the look-ahead trace
            const float rr = closed01(r.v[1]);
            float tm = transport.x > transport.y ? transport.x : transport.y;
            tm = tm > transport.z ? tm : transport.z;
            const float survival = fmax_(tm, 0.1f);
            if (rr < survival) transport = transport / survival;
            else { radiance = acc; return SHADE_END; }
            nd = cosine_weighted_diffuse(normal, closed01(r.v[2]), closed01(r.v[3]));
            const V3 no = advance(Pt, nd);
            const float cos_i = dot(nd, normal);
            pv.o = no;
            pv.d = nd;
            pv.tmax = kInf;
            pv.cos_i = cos_i;
            pv.transport = transport;
            pv.flags = it | kPhaseBit | (sf.mat << kMatShift);
            if (NEE) pv.acc = acc;
            return SHADE_ALIVE;
        }
    } else {  // Dielectric, camera.rs:375-417
        const float n1 = m.a, n2 = m.b;
        if (inside) {
            const V3 ab = (V3{-1.f, -1.f, -1.f} + color) * t;
            transport = mul_ew(transport, V3{pt_expf(ab.x), pt_expf(ab.y), pt_expf(ab.z)});
        }
        const V3 nf = inside ? -normal : normal;
        const float na = inside ? n2 : n1, nb = inside ? n1 : n2;
        V3 refr;
        if (refract(d, nf, na, nb, &refr)) {
            const float sr = schlick(d, nf, na, nb);
            const U4 r = philox4x32_10(k0, k1, px.x, px.y, 1u + it, 0u);
            if (closed01(r.v[0]) < sr) {
                if (!inside) transport = mul_ew(transport, color);
                nd = reflect(d, nf);
            } else {
                nd = refr;
            }
        } else {
            nd = reflect(d, nf);
        }
    }
    // ray.reset(P, nd, inf) and on to the next iteration
    it += 1;
    if (it >= depth) { radiance = acc; return SHADE_END; }
    pv.o = advance(Pt, nd);
    pv.d = nd;
    pv.tmax = kInf;
    pv.cos_i = 0.0f;
    pv.transport = transport;
    pv.flags = it;
    if (NEE) pv.acc = acc;
    return SHADE_ALIVE;
}

__device__ __forceinline__ void redo_push(const RedoList& R, uint32_t slot) {
    const uint32_t at = atomicAdd(R.count, 1u);
    if (at < R.cap) R.list[at] = slot;
}

// shade_core on a path whose state lives in HBM (the wavefront round kernel and the queue-fed tail kernel)
template <int PRIM, bool NEE>
__device__ __forceinline__ bool shade_one(const DevScene& S, const PathState& P, uint32_t slot, uint32_t depth, uint32_t k0, uint32_t k1,
                                          LaneStack stk, unsigned long long* counters) {
    const float4 h = P.hit[slot], ro = P.ro[slot], rd = P.rd[slot], tp = P.tp[slot];
    const uint2 px = P.pxs[slot];
    PathVals pv;
    pv.o = xyz(ro);
    pv.d = xyz(rd);
    pv.tmax = ro.w;
    pv.cos_i = rd.w;
    pv.transport = xyz(tp);
    pv.flags = __float_as_uint(tp.w);
    pv.acc = V3{0.f, 0.f, 0.f};
    if (NEE) pv.acc = xyz(P.acc[slot]);
    pv.pixel = px.x;
    pv.sample = px.y;
    HitVals hv;
    hv.bits = __float_as_uint(h.x);
    hv.t = h.y;
    hv.u = h.z;
    hv.v = h.w;
    V3 radiance;
    NeeWork nee_work{0, 0, 0};
    const int status = shade_core<PRIM, NEE>(S, pv, hv, depth, k0, k1, stk, counters, radiance, nee_work);
    if (status == SHADE_REDO) {  // k_redo finishes this path (and writes its P.out) before the batch is accumulated
        redo_push(P.redo, slot);
        return false;
    }
    if (status == SHADE_END) {
        P.out[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
        return false;
    }
    P.ro[slot] = make_float4(pv.o.x, pv.o.y, pv.o.z, pv.tmax);
    P.rd[slot] = make_float4(pv.d.x, pv.d.y, pv.d.z, pv.cos_i);
    P.tp[slot] = make_float4(pv.transport.x, pv.transport.y, pv.transport.z, __uint_as_float(pv.flags));
    if (NEE) P.acc[slot] = make_float4(pv.acc.x, pv.acc.y, pv.acc.z, 0.f);
    return true;
}


template <int PRIM, bool NEE>
__global__ __launch_bounds__(kTraceBlock) void k_shade(DevScene S, PathState P, const uint32_t* __restrict__ q_in,
                                                       const uint32_t* __restrict__ qn_in, uint32_t* __restrict__ q_out,
                                                       uint32_t* __restrict__ qn_out, uint32_t depth, uint32_t k0, uint32_t k1,
                                                       LaunchCfg cfg, unsigned long long* counters) {
    LaneStack stk;
    stk.lds = nullptr;
    stk.spill = nullptr;
    stk.spill_stride = 0;
    if constexpr (NEE) {
        __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
        stk = make_stack(smem, cfg);
    }
    const uint32_t n = *qn_in;
    for (uint32_t base = blockIdx.x * kTraceBlock; base < n; base += gridDim.x * kTraceBlock) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = q_in[i];
            alive = shade_one<PRIM, NEE>(S, P, slot, depth, k0, k1, stk, counters);
        }
        wave_enqueue(alive, slot, q_out, qn_out);
    }
}

// The tail of a batch: once few paths are left, per-round launches are bound by the latency of
// the longest single traversal.  This persistent kernel runs whole remaining paths instead:
// every lane repeats {trace, shade} on its path and, when the path ends, fetches the next one
// from the queue (one wave-aggregated atomic per refill), so waves stay full until the queue is
// empty.  Same two device functions as the round kernels, hence the same results.
template <int PRIM, bool NEE, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_tail(DevScene S, PathState P, const uint32_t* __restrict__ queue,
                                                      const uint32_t* __restrict__ qn, uint32_t* __restrict__ cursor, uint32_t depth,
                                                      uint32_t k0, uint32_t k1, LaunchCfg cfg, unsigned long long* counters) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    LaneStack stk = make_stack(smem, cfg);
    const uint32_t n = *qn;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nn = 0, np = 0, worst = 0, rays = 0;
    uint32_t slot = 0;
    bool alive = false, drained = false;
    while (true) {
        // refill: lanes without a path take consecutive queue entries
        const unsigned long long want = __ballot(!alive && !drained);
        if (want) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cursor, (uint32_t)__popcll(want));
            base = __shfl(base, 0, 64);
            if (!alive && !drained) {
                const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                if (idx < n) {
                    slot = queue[idx];
                    alive = true;
                } else {
                    drained = true;
                }
            }
        }
        if (__ballot(alive) == 0) break;
        if (alive) {
            const float4 ro = P.ro[slot], rd = P.rd[slot];
            float dist = ro.w;
            Hit h;
            const uint32_t nn0 = nn;
            traverse<PRIM, false, COUNT>(S, xyz(ro), xyz(rd), dist, h, stk, nn, np);
            if (COUNT) worst = nn - nn0 > worst ? nn - nn0 : worst;
            rays += 1;
            const uint32_t bits = h.slot == PT_MISS ? PT_MISS : (h.slot | (h.inside << 31));
            P.hit[slot] = make_float4(__uint_as_float(bits), dist, h.u, h.v);
            alive = shade_one<PRIM, NEE>(S, P, slot, depth, k0, k1, stk, counters);
        }
    }
    wave_add(counters + CNT_RAYS_CLOSEST, rays);
    wave_add(counters + CNT_TAIL_RAYS, rays);
    if (COUNT) {
        wave_add(counters + CNT_TAIL_NODE_TESTS, nn);
        wave_add(counters + CNT_TAIL_PRIM_TESTS, np);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t other = __shfl_down(worst, off, 64);
            worst = other > worst ? other : worst;
        }
        if ((threadIdx.x & 63) == 0 && worst) atomicMax(counters + CNT_MAX_NODE_TESTS, (unsigned long long)worst);
    }
}

// k_tail_pw — the tail as ONE persistent kernel with phase scheduling.  Every lane is idle, traversing
// or waiting to be shaded.  The wave runs the shared internal / leaf phases of k_extend_pw for the
// traversing lanes; when kShadeLanes lanes wait for shading (or nobody is traversing any more) it
// shades them together and puts the survivors' next rays straight back into traversal; idle lanes
// take fresh paths from the queue.  Same shade_one / phase functions as the round kernels.
template <int PRIM, bool NEE, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_tail_pw(DevScene S, PathState P, const uint32_t* __restrict__ queue,
                                                         const uint32_t* __restrict__ qn, uint32_t* __restrict__ cursor, uint32_t depth,
                                                         uint32_t k0, uint32_t k1, LaunchCfg cfg, unsigned long long* counters) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    LaneStack stk = make_stack(smem, cfg);
    const uint32_t n = *qn;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nn = 0, np = 0, worst = 0, mine = 0, rays = 0;
    LaneTrav T;
    T.cur = 0; T.sp = 0; T.dist = 0.f; T.hslot = PT_MISS; T.hu = T.hv = 0.f; T.slot = 0;
    T.o = T.d = T.rcp = V3{0.f, 0.f, 0.f};
    bool active = false;   // traversing
    bool pending = false;  // traversal finished, hit record stored, waits for shade_one
    bool drained = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    const uint32_t chunk = 32;
    while (true) {
        unsigned long long pend = __ballot(pending);
        unsigned long long act = __ballot(active);
        if (pend != 0 && (__popcll(pend) >= kShadeLanes || act == 0)) {
            if (pending) {
                pending = false;
                if (shade_one<PRIM, NEE>(S, P, T.slot, depth, k0, k1, stk, counters)) {
                    rays += 1;
                    active = lane_start_ray<COUNT>(S, T, P.ro[T.slot], P.rd[T.slot], nn, mine);
                    if (!active) {
                        lane_store_hit(P, T);
                        pending = true;
                    }
                }
            }
            continue;
        }
        const unsigned long long idle = __ballot(!active && !pending);
        const int n_idle = __popcll(idle);
        if (!drained && n_idle >= kRefillLanes) {
            if (!wave_fetch_chunk(cursor, n, chunk, chunk_next, chunk_end)) drained = true;
            if (!active && !pending && !drained) {
                const uint32_t idx = chunk_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (idx < chunk_end) {
                    T.slot = queue[idx];
                    rays += 1;
                    active = lane_start_ray<COUNT>(S, T, P.ro[T.slot], P.rd[T.slot], nn, mine);
                    if (!active) {
                        lane_store_hit(P, T);
                        pending = true;
                    }
                }
            }
            chunk_next = chunk_next + (uint32_t)n_idle < chunk_end ? chunk_next + (uint32_t)n_idle : chunk_end;
            continue;
        }
        if (act == 0) {
            if (pend != 0) continue;  // (cannot happen: pend != 0 with act == 0 is shaded above)
            if (drained) break;
            continue;  // fewer than kRefillLanes idle cannot be: everyone is idle here
        }
        // traversal steps (inner hot loop) until enough lanes wait for shading or are free for a refill
        while (true) {
            const unsigned long long actv = __ballot(active);
            if (actv == 0) break;
            const int n_wait = __popcll(__ballot(pending));
            if (n_wait >= kShadeLanes || (!drained && 64 - __popcll(actv) - n_wait >= kRefillLanes)) break;
            wave_step<PRIM, COUNT>(S, T, active, stk, actv, nn, np, mine);
            if (!active && ((actv >> lane) & 1ull)) {
                lane_store_hit(P, T);
                pending = true;
                if (COUNT) worst = mine > worst ? mine : worst;
            }
        }
    }
    wave_add(counters + CNT_RAYS_CLOSEST, rays);
    wave_add(counters + CNT_TAIL_RAYS, rays);
    if (COUNT) {
        wave_add(counters + CNT_TAIL_NODE_TESTS, nn);
        wave_add(counters + CNT_TAIL_PRIM_TESTS, np);
        reduce_worst(counters, worst);
    }
}

// k_path — the whole of generate() + sample() for a batch of paths in ONE persistent kernel, path
// state in registers.  Lanes are idle, traversing, or waiting to be shaded (as in k_tail_pw), but
//   * an idle lane takes the next path slot of the batch and generates its primary ray itself
//     (Camera::generate), so there is no raygen pass, no ray queue and no compaction;
//   * shade_core works on the lane's registers and puts the next ray straight back into traversal;
//   * the only per-path HBM traffic is the 16-byte radiance written when the path ends.
// slots are pixel-major (8x8-tile pixel order, then the samples of the pixel), and every value is
// computed by the same device functions, so the sums are bit-identical.
template <int PRIM, bool NEE, bool COUNT, bool SEVEN>
__global__ __launch_bounds__(kTraceBlock) PT_PATH_ATTR void k_path(DevScene S, const DevCamera* __restrict__ camp, PixelMap M, uint32_t n_samples, FastDiv fd_samples,
                                                      uint32_t sample_base, uint32_t depth, uint32_t k0, uint32_t k1,
                                                      float4* __restrict__ out, uint32_t* __restrict__ cursor, RedoList redo, LaunchCfg cfg,
                                                      unsigned long long* counters) {
    constexpr int kLevels = SEVEN ? kLdsStack7 : kLdsStack;
    __shared__ unsigned long long smem[kLevels * kTraceBlock];
#if PT_TOUCH_NEXT
    __shared__ uint32_t touch_dump[kTraceBlock];  // where the touches land: a dword per lane, never read
    lds_u32* const wave_dump = (lds_u32*)touch_dump + (threadIdx.x & ~63u);
#endif
    LaneStack stk = make_stack(smem, cfg, (uint32_t)kLevels);
    const DevCamera cam = *camp;
    const uint32_t n = n_samples * M.npix_pad;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nn = 0, np = 0, worst = 0, mine = 0;
    uint32_t rays = 0, paths = 0;  // wave-uniform tallies (scalar registers)
    LaneTrav T;
    T.cur = 0; T.sp = 0; T.dist = 0.f; T.hslot = PT_MISS; T.hu = T.hv = 0.f; T.slot = 0;
    T.o = T.d = T.rcp = V3{0.f, 0.f, 0.f};
    PathVals pv;
    pv.o = pv.d = pv.transport = pv.acc = V3{0.f, 0.f, 0.f};
    pv.tmax = pv.cos_i = 0.f;
    pv.flags = pv.pixel = pv.sample = 0;
    bool active = false, pending = false, drained = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    // wave-uniform: the XCD this wave runs on (HW_REG_XCC_ID[3:0]) picks its region of the slot range
    uint32_t region = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u, tried = 0;
    // one band of the image per XCD keeps the most locality (256 spp: +3 %); with few samples per pixel the bands
    // are too uneven, so each XCD then gets 8 strips spread over the image (64 spp: +1 % instead of -3 %)
    const uint32_t xcd_strips = n_samples >= 128u ? 1u : 8u;
    PhaseStats st;
    st.init();
    const unsigned long long t_begin = st.now();
    const uint32_t per_wave = n / (gridDim.x * (kTraceBlock / 64)) + 1;
    const uint32_t chunk = per_wave >= (uint32_t)kFetchChunk ? (uint32_t)kFetchChunk : (per_wave < 64u ? 64u : (per_wave & ~63u));
    while (true) {
        const unsigned long long pend = __ballot(pending);
        const unsigned long long act = __ballot(active);
        // once the batch has no more paths to hand out the wave only empties: from then on it runs
        // for latency (shade at once, both kinds of traversal step every turn), not for full waves
        if (pend != 0 && (__popcll(pend) >= kPathShadeLanes || act == 0 || drained)) {
            const unsigned long long t0 = st.now();
            bool traced = false;
            if (pending) {
                pending = false;
                HitVals hv;
                hv.bits = T.hslot;
                hv.t = T.dist;
                hv.u = T.hu;
                hv.v = T.hv;
                pv.o = T.o;
                pv.d = T.d;
                {   // the RNG counter words (pixel, sample) follow from the slot: not kept across the traversal loop
                    const uint32_t pix_idx = fastdiv(T.slot, fd_samples);
                    uint32_t x, y;
                    map_pixel(M, pix_idx, &x, &y);
                    pv.pixel = y * M.width + x;
                    pv.sample = sample_base + (T.slot - pix_idx * n_samples);
                }
                V3 radiance;
                NeeWork nee_work{0, 0, 0};
                const int status = shade_core<PRIM, NEE>(S, pv, hv, depth, k0, k1, stk, counters, radiance, nee_work);
                if (COUNT && NEE) { nn += nee_work.node_tests; np += nee_work.blocks + nee_work.prim_tests; }
                if (status == SHADE_ALIVE) {
                    traced = true;
                    active = lane_start_ray<COUNT>(S, T, make_float4(pv.o.x, pv.o.y, pv.o.z, pv.tmax), make_float4(pv.d.x, pv.d.y, pv.d.z, 0.f), nn, mine);
                    pending = !active;  // a ray that misses the root box is shaded as a miss
                } else if (status == SHADE_END) {
                    out[T.slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
                } else {
                    redo_push(redo, T.slot);  // k_redo runs this path again, with the NEE code, before the batch is accumulated
                }
            }
            rays += (uint32_t)__popcll(__ballot(traced));
            st.add(0, pend, t0);
            continue;
        }
        const unsigned long long idle = __ballot(!active && !pending);
        const int n_idle = __popcll(idle);
        if (!drained && n_idle >= kPathRefillLanes) {
            const unsigned long long t0 = st.now();
            if (!wave_fetch_chunk_xcd(cursor, n, chunk, xcd_strips, region, tried, chunk_next, chunk_end)) drained = true;
            bool made = false;
            if (!active && !pending && !drained) {
                const uint32_t idx = chunk_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (idx < chunk_end) {
                    // pixel-major slots: the samples of a pixel are neighbours, so the lanes of a wave start
                    // with (nearly) the same primary ray and walk the same nodes
                    const uint32_t pix_idx = fastdiv(idx, fd_samples), s_local = idx - pix_idx * n_samples;
                    uint32_t x, y;
                    if (map_pixel(M, pix_idx, &x, &y)) {
                        T.slot = idx;
                        V3 o, d;
                        const uint32_t sample = sample_base + s_local;
                        generate_ray(cam, x, y, sample, k0, k1, &o, &d);
                        pv.transport = V3{1.f, 1.f, 1.f};
                        pv.flags = 0;
                        pv.cos_i = 0.f;
                        pv.acc = V3{0.f, 0.f, 0.f};
                        made = true;
                        if (depth == 0) {  // `for _ in 0..0`: the sample is (0,0,0) and nothing is traced
                            out[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
                        } else {
                            active = lane_start_ray<COUNT>(S, T, make_float4(o.x, o.y, o.z, kInf), make_float4(d.x, d.y, d.z, 0.f), nn, mine);
                            pending = !active;
                        }
                    }
                }
            }
            chunk_next = chunk_next + (uint32_t)n_idle < chunk_end ? chunk_next + (uint32_t)n_idle : chunk_end;
            const uint32_t n_made = (uint32_t)__popcll(__ballot(made));
            paths += n_made;
            if (depth != 0) rays += n_made;
            st.add(1, idle, t0);
            continue;
        }
        if (act == 0) {
            if (drained) break;
            continue;
        }
        // Traversal steps until enough lanes wait for shading or for a new path.  Each step the wave
        // does whichever kind (internal node / leaf primitive) more of its lanes are waiting for.
        // This is the hot loop: it is an inner loop of its own so that register allocation favours it.
        while (true) {
            const unsigned long long actv = __ballot(active);
            if (actv == 0) break;
            const int n_wait = __popcll(__ballot(pending));
            if (n_wait >= kPathShadeLanes || (drained && n_wait != 0)) break;
            if (!drained && 64 - __popcll(actv) - n_wait >= kPathRefillLanes) break;
            const bool at_internal = active && !(T.cur & kLeafBit);
            const bool at_leaf = active && !at_internal;
            const unsigned long long mi = __ballot(at_internal);
            const int n_int = __popcll(mi), n_leaf = __popcll(actv) - n_int;
            const bool internal_turn = n_int * kStepBiasNum >= n_leaf * kStepBiasDen;
#ifdef PT_PHASE_STATS
            if (drained) {
                if (st.exec[4] == 0) st.clk[4] = st.now();  // becomes the time spent drained at flush
                st.exec[4] += 1;
                st.lanes[4] += (unsigned long long)(__popcll(actv) + n_wait);
            }
#endif
            if (PT_DRAIN_GHOST && kPairDfs && drained) {
                const unsigned long long t0 = st.now();
                if (mi) {  // (wave-uniform)
                    step_internal_readahead<COUNT>(S, T, active, at_internal, mi, stk, nn, mine);
#pragma unroll
                    for (int rep = 1; rep < kInternalRepeat; ++rep) {
                        const bool again = at_internal && active && !(T.cur & kLeafBit);
                        const unsigned long long m2 = __ballot(again);
                        if (m2) step_internal_readahead<COUNT>(S, T, active, again, m2, stk, nn, mine);
                    }
                }
                st.add(2, mi, t0);
            } else if (internal_turn || drained) {
                const unsigned long long t0 = st.now();
                if (at_internal) {
                    step_internal<COUNT>(S, T, active, stk, nn, mine);
                    // most internal nodes lead to another one: a lane takes up to kInternalRepeat of them per turn
                    // (a fixed count measured faster than "while most of the starters still descend")
#pragma unroll
                    for (int rep = 1; rep < kInternalRepeat; ++rep)
                        if (active && !(T.cur & kLeafBit)) step_internal<COUNT>(S, T, active, stk, nn, mine);
                }
                st.add(2, mi, t0);
            }
            if (!internal_turn || drained) {
                const unsigned long long t0 = st.now();
                step_prim_wave<PRIM, COUNT>(S, T, active, at_leaf, stk, np);
                st.add(3, actv & ~mi, t0);
            }
            if (!active && ((actv >> lane) & 1ull)) {
                pending = true;
                if (COUNT) worst = mine > worst ? mine : worst;
            }
#if PT_TOUCH_NEXT
            if (active && !(T.cur & kLeafBit)) record_touch(record_rsrc(S.pairs, S.n_pairs + 1u), T.cur, wave_dump);
#if PT_TOUCH_NEXT >= 2
            if (PRIM == PRIM_TRIANGLE && active && (T.cur & kLeafBit) && ((T.cur >> kLeafCountShift) & 31u) != 0u)
                record_touch(record_rsrc(S.prim, S.n_prims), T.cur & kLeafFirstMask, wave_dump);
#endif
#endif
        }
    }
#ifdef PT_PHASE_STATS
    if (st.exec[4]) st.clk[4] = st.now() - st.clk[4];
    if ((threadIdx.x & 63u) == 0) {
        atomicMax(counters + CNT_MAX_NODE_TESTS, st.exec[4]);       // most drained-mode iterations of any wave
        atomicMax(counters + CNT_NODE_TESTS, st.clk[4]);            // longest drained time of any wave (clocks)
    }
#endif
    st.add(5, ~0ull, t_begin);  // wave lifetime
    st.flush(counters);
    if (lane == 0) {
        atomicAdd(counters + CNT_PATHS, (unsigned long long)paths);
        atomicAdd(counters + CNT_RAYS_CLOSEST, (unsigned long long)rays);
        atomicAdd(counters + CNT_TAIL_RAYS, (unsigned long long)rays);
    }
    if (COUNT) {
        wave_add(counters + CNT_TAIL_NODE_TESTS, nn);
        wave_add(counters + CNT_TAIL_PRIM_TESTS, np);
        reduce_worst(counters, worst);
    }
}

// k_redo — the paths a kernel without the NEE code gave up (SHADE_REDO), run again from generate() with the NEE code,
// one lane per path.  A path is a function of (pixel, sample, seed), so the second run reaches the giving-up point in the
// same state, takes the NEE block there and goes on.  The first run has already counted the path and every ray up to
// that point; this kernel counts only what comes after (the block runs for the first time exactly there: before it the
// transport was within nee_thr, and these scenes have no non-zero NEE weight).  A handful of paths per frame at most
// (C3: the unguarded light pdf of camera.rs:361-370 makes ~0.3 non-finite samples per million), so nothing here is tuned.
template <int PRIM, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_redo(DevScene S, const DevCamera* __restrict__ camp, PixelMap M, uint32_t n_samples, int pixel_major,
                                                      uint32_t sample_base, uint32_t depth, uint32_t k0, uint32_t k1, float4* __restrict__ out,
                                                      const uint32_t* __restrict__ list, uint32_t n, LaunchCfg cfg, unsigned long long* counters) {
    __shared__ unsigned long long smem[kLdsStack * kTraceBlock];
    LaneStack stk = make_stack(smem, cfg);
    const DevCamera cam = *camp;
    uint32_t rays = 0, nn = 0, np = 0;
    for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const uint32_t slot = list[i];
        // slab layout: [pixel][sample] from k_path, [sample][pixel] from the wavefront kernels
        const uint32_t pix_idx = pixel_major ? slot / n_samples : slot % M.npix_pad;
        const uint32_t s_local = pixel_major ? slot % n_samples : slot / M.npix_pad;
        uint32_t x, y;
        map_pixel(M, pix_idx, &x, &y);
        PathVals pv;
        pv.pixel = y * M.width + x;
        pv.sample = sample_base + s_local;
        generate_ray(cam, x, y, pv.sample, k0, k1, &pv.o, &pv.d);
        pv.tmax = kInf;
        pv.cos_i = 0.f;
        pv.transport = V3{1.f, 1.f, 1.f};
        pv.flags = 0;
        pv.acc = V3{0.f, 0.f, 0.f};
        V3 radiance = V3{0.f, 0.f, 0.f};
        bool counting = false;
        while (true) {
            float dist = pv.tmax;
            Hit h;
            uint32_t a = 0, b = 0;
            traverse<PRIM, false, COUNT>(S, pv.o, pv.d, dist, h, stk, a, b);
            if (counting) { rays += 1; nn += a; np += b; }
            HitVals hv;
            hv.bits = h.slot == PT_MISS ? PT_MISS : (h.slot | (h.inside << 31));
            hv.t = dist;
            hv.u = h.u;
            hv.v = h.v;
            NeeWork nee_work{0, 0, 0};
            const int status = shade_core<PRIM, true>(S, pv, hv, depth, k0, k1, stk, counters, radiance, nee_work);
            if (nee_work.blocks) { counting = true; nn += nee_work.node_tests; np += nee_work.blocks + nee_work.prim_tests; }
            if (status != SHADE_ALIVE) break;
        }
        out[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
    wave_add(counters + CNT_RAYS_CLOSEST, rays);
    wave_add(counters + CNT_TAIL_RAYS, rays);
    if (COUNT) {
        wave_add(counters + CNT_TAIL_NODE_TESTS, nn);
        wave_add(counters + CNT_TAIL_PRIM_TESTS, np);
    }
}

// Primitive::area() of every light (triangle.rs:105-113, sphere.rs:119-121) must be a finite number >= 0 for the bound
// on cos / mis that lets the NEE block be left out (shade_core); otherwise flag[0] is set and the block always runs.
template <int PRIM>
__global__ void k_light_check(DevScene S, uint32_t* __restrict__ flag) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < S.n_lights; i += gridDim.x * blockDim.x) {
        const uint32_t slot = S.lights[i];
        float area;
        if (PRIM == PRIM_TRIANGLE) {
            const float4* tp = S.prim + (size_t)slot * kTriStride;
            const float4 a = tp[0], b = tp[1], c = tp[2];
            area = tri_area(V3{a.w, b.x, b.y}, V3{b.z, b.w, c.x});
        } else {
            area = sphere_area(S.prim[slot].w);
        }
        if (!(area >= 0.0f && area <= 3.402823466e38f)) atomicOr(flag, 1u);
    }
}

// chunk[idx] += sample, src/main.rs:159 — samples of one pixel are added in sample order, so the
// sum is the one the reference's frame loop builds one frame at a time.
__global__ __launch_bounds__(64) void k_accumulate(PixelMap M, const float4* __restrict__ out, uint32_t n_samples, int pixel_major,
                                                    float* __restrict__ accum) {
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < M.npix_pad; idx += gridDim.x * blockDim.x) {
        uint32_t x, y;
        if (!map_pixel(M, idx, &x, &y)) continue;
        float* a = accum + ((size_t)y * M.width + x) * 3;
        V3 sum = ld3(a);
        // slab layout: [sample][pixel] from the wavefront kernels, [pixel][sample] from k_path
        if (pixel_major) {
            // a lane owns 16 * n_samples contiguous bytes: fetch a whole 128-byte line (8 samples) at a
            // time so every line is asked for once, then add in sample order
            const float4* src = out + (size_t)idx * n_samples;
            uint32_t s = 0;
            for (; s + 8 <= n_samples; s += 8) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = src[s + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) sum = sum + xyz(v[k]);
            }
            for (; s < n_samples; ++s) sum = sum + xyz(src[s]);
        } else {
            const float4* src = out + idx;
            for (uint32_t s = 0; s < n_samples; ++s) sum = sum + xyz(src[(size_t)s * M.npix_pad]);
        }
        a[0] = sum.x;
        a[1] = sum.y;
        a[2] = sum.z;
    }
}

// Tile-sharded frames: a rank's tiles as one compact slab in the shard's own pixel order (tile by tile,
// 8x8 blocks inside, padding pixels zero) — what is gathered over RCCL — and back into a frame.
// unpack = 0: slab <- accum;  unpack = 1: accum <- slab (the un-swizzle on the gathering rank).
__global__ __launch_bounds__(256) void k_shard_copy(PixelMap M, float* __restrict__ accum, float* __restrict__ slab, int unpack) {
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < M.npix_pad; idx += gridDim.x * blockDim.x) {
        uint32_t x, y;
        const bool valid = map_pixel(M, idx, &x, &y);
        float* s = slab + (size_t)idx * 3;
        if (!valid) {
            if (!unpack) s[0] = s[1] = s[2] = 0.0f;
            continue;
        }
        float* a = accum + ((size_t)y * M.width + x) * 3;
        if (unpack) { a[0] = s[0]; a[1] = s[1]; a[2] = s[2]; }
        else { s[0] = a[0]; s[1] = a[1]; s[2] = a[2]; }
    }
}

__global__ __launch_bounds__(256) void k_resolve(const float* __restrict__ accum, uint32_t width, uint32_t height, float scale,
                                                 uint8_t* __restrict__ rgb, uint64_t pitch) {
    const uint32_t n = width * height;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t x = i % width, y = i / width;
        const V3 v = scale * ld3(accum + (size_t)i * 3);
        uint8_t* p = rgb + (size_t)y * pitch + (size_t)x * 3;
        p[0] = tonemap_channel(v.x);
        p[1] = tonemap_channel(v.y);
        p[2] = tonemap_channel(v.z);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
static inline uint32_t blocks_for(uint64_t n, uint32_t block, uint32_t cap) {
    uint64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    return (uint32_t)(b < cap ? b : cap);
}

void launch_trace(int prim, bool any_hit, bool count, const DevScene& S, const pt_ray* rays, uint64_t n, pt_hit* hits,
                  const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st) {
    const uint32_t grid = blocks_for(n, kTraceBlock, cfg.grid);
#define PT_TRACE(P, A, C) hipLaunchKernelGGL((k_trace<P, A, C>), dim3(grid), dim3(kTraceBlock), 0, st, S, rays, n, hits, cfg, counters)
    if (prim == PRIM_TRIANGLE) {
        if (any_hit) { if (count) PT_TRACE(PRIM_TRIANGLE, true, true); else PT_TRACE(PRIM_TRIANGLE, true, false); }
        else { if (count) PT_TRACE(PRIM_TRIANGLE, false, true); else PT_TRACE(PRIM_TRIANGLE, false, false); }
    } else {
        if (any_hit) { if (count) PT_TRACE(PRIM_SPHERE, true, true); else PT_TRACE(PRIM_SPHERE, true, false); }
        else { if (count) PT_TRACE(PRIM_SPHERE, false, true); else PT_TRACE(PRIM_SPHERE, false, false); }
    }
#undef PT_TRACE
}

void launch_generate(const DevCamera* cam, const uint32_t* xy, uint64_t n, uint32_t sample, uint32_t k0, uint32_t k1, pt_ray* rays,
                     hipStream_t st) {
    hipLaunchKernelGGL(k_generate, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, st, cam, xy, n, sample, k0, k1, rays);
}

void launch_sky(const DevScene& S, const float* dirs, uint64_t n, float* rgb, hipStream_t st) {
    hipLaunchKernelGGL(k_sky, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, st, S, dirs, n, rgb);
}

void launch_focus(int prim, const DevScene& S, DevCamera* cam, const LaunchCfg& cfg, hipStream_t st) {
    if (prim == PRIM_TRIANGLE) hipLaunchKernelGGL((k_focus<PRIM_TRIANGLE>), dim3(1), dim3(64), 0, st, S, cam, cfg);
    else hipLaunchKernelGGL((k_focus<PRIM_SPHERE>), dim3(1), dim3(64), 0, st, S, cam, cfg);
}

void launch_raygen(const DevCamera* cam, const PixelMap& M, const PathState& P, uint32_t n_samples, uint32_t sample_base, uint32_t k0,
                   uint32_t k1, bool nee, uint32_t* queue, uint32_t* qn, unsigned long long* counters, hipStream_t st) {
    const uint64_t total = (uint64_t)n_samples * M.npix_pad;
    const int identity = (M.width % M.tile == 0 && M.height % M.tile == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_raygen, dim3(blocks_for(total, 256, 8192)), dim3(256), 0, st, cam, M, P, n_samples, sample_base, k0, k1,
                       nee ? 1 : 0, identity, queue, qn, counters);
}

void launch_extend(int prim, bool count, bool simple, const DevScene& S, const PathState& P, const uint32_t* queue, const uint32_t* qn,
                   uint32_t* cursor, const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st) {
    if (simple) {
#define PT_EXT(PR, C) hipLaunchKernelGGL((k_extend<PR, C>), dim3(cfg.grid), dim3(kTraceBlock), 0, st, S, P, queue, qn, cfg, counters)
        if (prim == PRIM_TRIANGLE) { if (count) PT_EXT(PRIM_TRIANGLE, true); else PT_EXT(PRIM_TRIANGLE, false); }
        else { if (count) PT_EXT(PRIM_SPHERE, true); else PT_EXT(PRIM_SPHERE, false); }
#undef PT_EXT
        return;
    }
#define PT_EXT(PR, C) hipLaunchKernelGGL((k_extend_pw<PR, C>), dim3(cfg.grid), dim3(kTraceBlock), 0, st, S, P, queue, qn, cursor, cfg, counters)
    if (prim == PRIM_TRIANGLE) { if (count) PT_EXT(PRIM_TRIANGLE, true); else PT_EXT(PRIM_TRIANGLE, false); }
    else { if (count) PT_EXT(PRIM_SPHERE, true); else PT_EXT(PRIM_SPHERE, false); }
#undef PT_EXT
}

void launch_shade(int prim, bool nee, const DevScene& S, const PathState& P, const PixelMap& M, const uint32_t* q_in, const uint32_t* qn_in,
                  uint32_t* q_out, uint32_t* qn_out, uint32_t depth, uint32_t sample_base, uint32_t k0, uint32_t k1,
                  const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st) {
    (void)M;
    (void)sample_base;
#define PT_SH(PR, N) hipLaunchKernelGGL((k_shade<PR, N>), dim3(cfg.grid), dim3(kTraceBlock), 0, st, S, P, q_in, qn_in, q_out, qn_out, depth, k0, k1, cfg, counters)
    if (prim == PRIM_TRIANGLE) { if (nee) PT_SH(PRIM_TRIANGLE, true); else PT_SH(PRIM_TRIANGLE, false); }
    else { if (nee) PT_SH(PRIM_SPHERE, true); else PT_SH(PRIM_SPHERE, false); }
#undef PT_SH
}

void launch_tail(int prim, bool nee, bool count, int mode, const DevScene& S, const PathState& P, const uint32_t* queue, const uint32_t* qn,
                 uint32_t* cursor, uint32_t n_hint, uint32_t depth, uint32_t k0, uint32_t k1, const LaunchCfg& cfg,
                 unsigned long long* counters, hipStream_t st) {
    // n_hint lanes; further paths are fetched as lanes free up
    const uint32_t grid = blocks_for(n_hint, kTraceBlock, cfg.grid);
#define PT_TAIL(K, PR, N, C) hipLaunchKernelGGL((K<PR, N, C>), dim3(grid), dim3(kTraceBlock), 0, st, S, P, queue, qn, cursor, depth, k0, k1, cfg, counters)
#define PT_TAIL_ALL(K)                                                                                                  \
    if (prim == PRIM_TRIANGLE) {                                                                                        \
        if (nee) { if (count) PT_TAIL(K, PRIM_TRIANGLE, true, true); else PT_TAIL(K, PRIM_TRIANGLE, true, false); }     \
        else { if (count) PT_TAIL(K, PRIM_TRIANGLE, false, true); else PT_TAIL(K, PRIM_TRIANGLE, false, false); }       \
    } else {                                                                                                            \
        if (nee) { if (count) PT_TAIL(K, PRIM_SPHERE, true, true); else PT_TAIL(K, PRIM_SPHERE, true, false); }         \
        else { if (count) PT_TAIL(K, PRIM_SPHERE, false, true); else PT_TAIL(K, PRIM_SPHERE, false, false); }           \
    }
    if (mode == 2) { PT_TAIL_ALL(k_tail) } else { PT_TAIL_ALL(k_tail_pw) }
#undef PT_TAIL_ALL
#undef PT_TAIL
}

void launch_path(int prim, bool nee, bool count, const DevScene& S, const DevCamera* cam, const PixelMap& M, uint32_t n_samples,
                 uint32_t sample_base, uint32_t depth, uint32_t k0, uint32_t k1, float4* out, uint32_t* cursor, const RedoList& redo,
                 const LaunchCfg& cfg, bool seven_waves, unsigned long long* counters, hipStream_t st) {
    const uint64_t total = (uint64_t)n_samples * M.npix_pad;
    const uint32_t grid = blocks_for(total, kTraceBlock, cfg.grid);
#define PT_PATH4(PR, N, C, W) hipLaunchKernelGGL((k_path<PR, N, C, W>), dim3(grid), dim3(kTraceBlock), 0, st, S, cam, M, n_samples, fastdiv_make(n_samples), sample_base, depth, k0, k1, out, cursor, redo, cfg, counters)
#define PT_PATH(PR, N, C) do { if (seven_waves) PT_PATH4(PR, N, C, true); else PT_PATH4(PR, N, C, false); } while (0)
    if (prim == PRIM_TRIANGLE) {
        if (nee) { if (count) PT_PATH(PRIM_TRIANGLE, true, true); else PT_PATH(PRIM_TRIANGLE, true, false); }
        else { if (count) PT_PATH(PRIM_TRIANGLE, false, true); else PT_PATH(PRIM_TRIANGLE, false, false); }
    } else {
        if (nee) { if (count) PT_PATH(PRIM_SPHERE, true, true); else PT_PATH(PRIM_SPHERE, true, false); }
        else { if (count) PT_PATH(PRIM_SPHERE, false, true); else PT_PATH(PRIM_SPHERE, false, false); }
    }
#undef PT_PATH
#undef PT_PATH4
}

void launch_redo(int prim, bool count, const DevScene& S, const DevCamera* cam, const PixelMap& M, uint32_t n_samples, bool pixel_major,
                 uint32_t sample_base, uint32_t depth, uint32_t k0, uint32_t k1, float4* out, const uint32_t* list, uint32_t n,
                 const LaunchCfg& cfg, unsigned long long* counters, hipStream_t st) {
    const uint32_t grid = blocks_for(n, kTraceBlock, cfg.grid < 1024u ? cfg.grid : 1024u);  // a lane's spill column is its global thread id: stay inside cfg.grid
#define PT_REDO(PR, C) hipLaunchKernelGGL((k_redo<PR, C>), dim3(grid), dim3(kTraceBlock), 0, st, S, cam, M, n_samples, pixel_major ? 1 : 0, sample_base, depth, k0, k1, out, list, n, cfg, counters)
    if (prim == PRIM_TRIANGLE) { if (count) PT_REDO(PRIM_TRIANGLE, true); else PT_REDO(PRIM_TRIANGLE, false); }
    else { if (count) PT_REDO(PRIM_SPHERE, true); else PT_REDO(PRIM_SPHERE, false); }
#undef PT_REDO
}

void launch_light_check(int prim, const DevScene& S, uint32_t* flag, hipStream_t st) {
    const uint32_t grid = blocks_for(S.n_lights, 256, 1024);
    if (prim == PRIM_TRIANGLE) hipLaunchKernelGGL((k_light_check<PRIM_TRIANGLE>), dim3(grid), dim3(256), 0, st, S, flag);
    else hipLaunchKernelGGL((k_light_check<PRIM_SPHERE>), dim3(grid), dim3(256), 0, st, S, flag);
}

void launch_accumulate(const PixelMap& M, const float4* out, uint32_t n_samples, bool pixel_major, float* accum, hipStream_t st) {
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(M.npix_pad, 64, 65536)), dim3(64), 0, st, M, out, n_samples, pixel_major ? 1 : 0, accum);
}

void launch_shard_copy(const PixelMap& M, float* accum, float* slab, bool unpack, hipStream_t st) {
    hipLaunchKernelGGL(k_shard_copy, dim3(blocks_for(M.npix_pad, 256, 8192)), dim3(256), 0, st, M, accum, slab, unpack ? 1 : 0);
}

void launch_resolve(const float* accum, uint32_t width, uint32_t height, float scale, uint8_t* rgb, uint64_t pitch, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for((uint64_t)width * height, 256, 8192)), dim3(256), 0, st, accum, width, height, scale, rgb, pitch);
}

}  // namespace pt
