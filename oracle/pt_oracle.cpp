// pt_oracle.cpp — CPU restatement of ReinierMaas/PathTracer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (pathtracer_amd/, libptcore.so) may
// include, link, import or execute this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it, and only as the checker / reported baseline.
//
// What it follows (all paths under /root/reference, studied as text; the Rust crate cannot be
// built here: no rustc/cargo, nightly-only feature gate, 10 crates.io dependencies, no network):
//   src/primitive/aabb.rs:14-66      AABB new/extent/combine/size/area/intersect
//   src/primitive/triangle.rs:41-113 Triangle intersect/centre/bounds/random_point/area
//   src/primitive/sphere.rs:34-121   Sphere   intersect/centre/bounds/random_point/area
//   src/ray.rs:20-36                 Ray new/reset/intersection
//   src/material.rs:5-27             Material, LIGHT_COLOR
//   src/bvh.rs:25-284                BVH new/subdivide/partition/SAH/closest/any/random_light
//   src/camera.rs:39-86,214-292,295-449  BSDF helpers, focus, update, sample, generate
//   src/scene.rs:162-169             sample_skybox
//   src/main.rs:71-75,143-170        vec_to_rgb, frame loop (accumulate, static row bands)
//
// Third-party arithmetic not under /root/reference, restated from the published behaviour of
// the pinned versions (Cargo.lock): cgmath 0.12.0 — dot = (x*x' + y*y') + z*z',
// cross = (y z' - z y', z x' - x z', x y' - y x'), magnitude = sqrt(dot), normalize =
// v * (1/|v|), element-wise mul/div, Array::max = partial_max chain; rand 0.3.15 — replaced by
// a counter-based Philox4x32-10 stream (the reference's thread_rng is OS-seeded, so its images
// are not reproducible run to run; parity with the Rust BINARY at the RNG boundary is unpinned).
//
// Pinning: the 24 known-answer rays of the reference's own inline tests (aabb.rs:96-133,
// triangle.rs:116-151, sphere.rs:124-160) are checked in tests/test_oracle_kat.py.  Nothing in
// the reference pins BVH build, traversal order, shading, sky lookup or tonemap: for those this
// file is a line-by-line restatement and "parity unpinned" beyond it.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/ptcore.h"          // POD struct layouts only (pt_ray, pt_hit, ...)
#include "../include/ptcore_detmath.h"  // the shared definition of the fp32 transcendentals

namespace {

constexpr float INF = std::numeric_limits<float>::infinity();
constexpr float EPS = 1.1920929e-7f;  // f32::EPSILON
constexpr float FRAC_1_PI = 0.318309886183790671538f;
constexpr float PI = 3.14159265358979323846f;

// ---------------------------------------------------------------- cgmath 0.12 restatement
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 mul_ew(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float magnitude(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }
// Rust f32::min/max (llvm.minnum / maxnum as rustc's x86 backend lowers them): the non-NaN operand wins, and of two
// equal operands (+0 / -0 included) the FIRST is returned.  Written out because libm disagrees on the tie: glibc's
// fminf(-0, +0) is +0 (second operand), LLVM's lowering gives -0 (first); only the sign of a zero can differ.
inline float fmin_(float a, float b) { return a != a ? b : (b < a ? b : a); }
inline float fmax_(float a, float b) { return a != a ? b : (b > a ? b : a); }

// ---------------------------------------------------------------- math mode
enum { MATH_DET = 0, MATH_LIBM = 1 };
struct Math {
    int mode = MATH_DET;
    void sincos(float x, float* s, float* c) const {
        if (mode == MATH_DET) pt_sincosf(x, s, c);
        else { *s = std::sin(x); *c = std::cos(x); }
    }
    float exp(float x) const { return mode == MATH_DET ? pt_expf(x) : std::exp(x); }
    float acos(float x) const { return mode == MATH_DET ? pt_acosf(x) : std::acos(x); }
    float atan2(float y, float x) const { return mode == MATH_DET ? pt_atan2f(y, x) : std::atan2(y, x); }
    float cbrt(float x) const { return mode == MATH_DET ? pt_cbrtf(x) : std::cbrt(x); }
};

// ---------------------------------------------------------------- ray.rs
struct Ray {
    V3 origin, direction;
    float distance;
    void reset(V3 o, V3 d, float dist) {  // ray.rs:28-32
        origin = o + (20.0f * EPS) * d;
        direction = d;
        distance = dist;
    }
    V3 intersection() const { return origin + distance * direction; }  // ray.rs:33-35
};

struct Intersection {  // ray.rs:6-11 (+ prim id for the parity seam)
    V3 normal;
    bool inside;
    float area;
    uint32_t material;
    uint32_t prim;
};

// ---------------------------------------------------------------- aabb.rs
struct AABB {
    V3 min, max;
    static AABB empty() { return {{INF, INF, INF}, {-INF, -INF, -INF}}; }  // aabb.rs:15-20
    AABB extent(V3 p) const {                                                // aabb.rs:21-29
        return {{fmin_(min.x, p.x), fmin_(min.y, p.y), fmin_(min.z, p.z)},
                {fmax_(max.x, p.x), fmax_(max.y, p.y), fmax_(max.z, p.z)}};
    }
    AABB combine(const AABB& o) const {  // aabb.rs:30-38
        return {{fmin_(min.x, o.min.x), fmin_(min.y, o.min.y), fmin_(min.z, o.min.z)},
                {fmax_(max.x, o.max.x), fmax_(max.y, o.max.y), fmax_(max.z, o.max.z)}};
    }
    V3 size() const { return max - min; }  // aabb.rs:39-41
    float area() const {                   // aabb.rs:42-45
        V3 e = size();
        return (2.0f * e.x * e.y + 2.0f * e.x * e.z) + 2.0f * e.y * e.z;
    }
    // aabb.rs:47-66
    bool intersect(const Ray& ray, float* tmin_out, float* tmax_out) const {
        V3 rcp = {1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
        V3 mn = min - ray.origin;
        V3 mx = max - ray.origin;
        V3 imin = {mn.x * rcp.x, mn.y * rcp.y, mn.z * rcp.z};
        V3 imax = {mx.x * rcp.x, mx.y * rcp.y, mx.z * rcp.z};
        float tmin = fmax_(fmax_(fmin_(imin.x, imax.x), fmin_(imin.y, imax.y)), fmin_(imin.z, imax.z));
        float tmax = fmin_(fmin_(fmax_(imin.x, imax.x), fmax_(imin.y, imax.y)), fmax_(imin.z, imax.z));
        if (tmax < 0.0f || tmin > tmax || ray.distance < tmin) return false;
        *tmin_out = tmin;
        *tmax_out = tmax;
        return true;
    }
};

// ---------------------------------------------------------------- RNG (counter based)
// Philox4x32-10 (Salmon et al., SC'11).  Stream layout shared with the device code:
//   key = (seed_lo, seed_hi), counter = (pixel, sample, block, stream)
//   stream 0, block 0      : generate()  -> r0 r1 r2 r3             camera.rs:431-435
//   stream 0, block 1 + it : sample() iteration `it` -> [0] specular / Fresnel choice
//                            (camera.rs:340,392) [1] russian roulette (:348)
//                            [2],[3] cosine_weighted_diffuse (:76,:78)
//   stream 1, block 1 + it : NEE of iteration `it` -> [0] light pick (bvh.rs:277-280)
//                            [1..3] Primitive::random_point (triangle.rs:98-99, sphere.rs:107-109)
struct U4 { uint32_t v[4]; };
inline U4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{{c0, c1, c2, c3}};
}
// rand 0.3 Closed01<f32> / Open01<f32> / gen::<f32>() shapes on a 24-bit mantissa draw
inline float closed01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777215.0f); }
inline float open01(uint32_t x) { return ((float)(x >> 8) + 0.25f) * (1.0f / 16777216.0f); }
inline float half_open01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

struct RngKey { uint32_t k0, k1, pixel, sample; };

// ---------------------------------------------------------------- primitives
struct Triangle {
    V3 p0, p1, p2, n0, n1, n2;
    uint32_t material;
};
struct Sphere {
    V3 position;
    float radius;
    uint32_t material;
};

inline float triangle_area(const Triangle& t) {  // triangle.rs:105-113
    V3 e1 = t.p1 - t.p0, e2 = t.p2 - t.p0;
    float len1 = magnitude(e1), len2 = magnitude(e2);
    e1 = e1 / len1;
    e2 = e2 / len2;
    return 0.5f * (1.0f - dot(e1, e2)) * len1 * len2;
}
inline float sphere_area(const Sphere& s) { return PI * s.radius * s.radius; }  // sphere.rs:119-121

// triangle.rs:41-79; also reports u, v, a for tests
inline bool triangle_intersect(const Triangle& tr, Ray& ray, Intersection* out, float* uv = nullptr) {
    V3 edge1 = tr.p1 - tr.p0;
    V3 edge2 = tr.p2 - tr.p0;
    V3 h = cross(ray.direction, edge2);
    float a = dot(edge1, h);
    if (a > -EPS && a < EPS) return false;
    float f = 1.0f / a;
    V3 s = ray.origin - tr.p0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, edge1);
    float v = f * dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot(edge2, q);
    if (t < 0.0f) return false;
    ray.distance = t;
    out->normal = normalize(((1.0f - u - v) * tr.n0 + u * tr.n1) + v * tr.n2);
    out->inside = a < 0.0f;
    out->area = triangle_area(tr);
    out->material = tr.material;
    if (uv) { uv[0] = u; uv[1] = v; uv[2] = a; }
    return true;
}

// sphere.rs:34-80
inline bool sphere_intersect(const Sphere& sp, Ray& ray, Intersection* out) {
    V3 distance = sp.position - ray.origin;
    float tca = dot(distance, ray.direction);
    float d2 = dot(distance, distance) - tca * tca;
    if (d2 > sp.radius * sp.radius) return false;
    float thc = std::sqrt(sp.radius * sp.radius - d2);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 >= 0.0f) {
        if (t0 >= ray.distance) return false;
        ray.distance = t0;
        out->normal = normalize(ray.origin + ray.direction * t0 - sp.position);
        out->inside = false;
    } else if (t1 >= 0.0f) {
        if (t1 >= ray.distance) return false;
        ray.distance = t1;
        out->normal = normalize(ray.origin + ray.direction * t1 - sp.position);
        out->inside = true;
    } else {
        return false;
    }
    out->area = sphere_area(sp);
    out->material = sp.material;
    return true;
}

struct Counters {
    uint64_t paths = 0, rays_closest = 0, rays_any = 0, node_tests = 0, prim_tests = 0;
    uint64_t aabb_calls = 0;   // literal AABB::intersect calls (3 per internal node entered + pops)
    uint64_t max_stack = 0;    // deepest literal node_stack
    uint64_t stack_hist[64] = {0};  // histogram of per-traversal max stack depth (far-only pushes)
    void add(const Counters& o) {
        paths += o.paths; rays_closest += o.rays_closest; rays_any += o.rays_any;
        node_tests += o.node_tests; prim_tests += o.prim_tests; aabb_calls += o.aabb_calls;
        max_stack = std::max(max_stack, o.max_stack);
        for (int i = 0; i < 64; ++i) stack_hist[i] += o.stack_hist[i];
    }
};

struct BVHNode {  // bvh.rs:10-14
    AABB bounds;
    uint32_t left_first, count;
};

}  // namespace

struct orc_ctx {
    Math math;
    int prim_kind = -1;  // 0 triangles, 1 spheres
    std::vector<Triangle> tris;
    std::vector<Sphere> spheres;
    std::vector<pt_material> mats;
    std::vector<float> sky;  // zero padded like the reference's oversize Vec (scene.rs:174)
    uint32_t sky_w = 0, sky_h = 0;

    // BVH (bvh.rs:17-22)
    std::vector<uint64_t> indices;
    std::vector<uint64_t> lights;
    std::vector<BVHNode> nodes;
    std::vector<AABB> obj_bounds;   // Primitive::bounds(), cached (pure function of the object)
    std::vector<V3> obj_centre;     // Primitive::centre(), cached
    bool built = false;
    bool hazard = false;
    uint32_t max_depth = 0;
    // nee_mode 1 (the device's rule for leaving the NEE block out), set by nee_rule() after build()
    float nee_thr = 3.402823466e38f;
    bool nee_literal = false;

    // camera (camera.rs:18-37)
    V3 origin{}, target{}, direction{}, p1{}, p2{}, p3{}, up{}, right{};
    float focal_distance = 0.f, lens_size = 0.04f;
    uint32_t width = 0, height = 0;
    bool have_camera = false;

    // frame state (main.rs:100-101)
    std::vector<V3> accum;
    float spp = 0.f;
    Counters counters;

    size_t n_prims() const { return prim_kind == 0 ? tris.size() : spheres.size(); }

    // ---- Primitive trait dispatch (primitive/mod.rs:12-19)
    bool prim_intersect(uint64_t obj, Ray& ray, Intersection* out) const {
        bool h = prim_kind == 0 ? triangle_intersect(tris[obj], ray, out) : sphere_intersect(spheres[obj], ray, out);
        if (h) out->prim = (uint32_t)obj;
        return h;
    }
    V3 prim_centre(uint64_t obj) const {
        if (prim_kind == 0) {  // triangle.rs:80-82
            const Triangle& t = tris[obj];
            return ((t.p0 + t.p1) + t.p2) * (1.0f / 3.0f);
        }
        return spheres[obj].position;  // sphere.rs:81-83
    }
    AABB prim_bounds(uint64_t obj) const {
        if (prim_kind == 0) {  // triangle.rs:83-90
            const Triangle& t = tris[obj];
            return {{fmin_(fmin_(t.p0.x, t.p1.x), t.p2.x), fmin_(fmin_(t.p0.y, t.p1.y), t.p2.y),
                     fmin_(fmin_(t.p0.z, t.p1.z), t.p2.z)},
                    {fmax_(fmax_(t.p0.x, t.p1.x), t.p2.x), fmax_(fmax_(t.p0.y, t.p1.y), t.p2.y),
                     fmax_(fmax_(t.p0.z, t.p1.z), t.p2.z)}};
        }
        const Sphere& s = spheres[obj];  // sphere.rs:84-91
        return {{s.position.x - s.radius, s.position.y - s.radius, s.position.z - s.radius},
                {s.position.x + s.radius, s.position.y + s.radius, s.position.z + s.radius}};
    }
    bool prim_is_light(uint64_t obj, V3* color) const {  // triangle.rs:91-96, sphere.rs:92-97
        uint32_t m = prim_kind == 0 ? tris[obj].material : spheres[obj].material;
        if (mats[m].kind != PT_MAT_EMISSIVE) return false;
        if (color) *color = v3(mats[m].rgb[0], mats[m].rgb[1], mats[m].rgb[2]);
        return true;
    }
    float prim_area(uint64_t obj) const {
        return prim_kind == 0 ? triangle_area(tris[obj]) : sphere_area(spheres[obj]);
    }
    V3 prim_random_point(uint64_t obj, const U4& r) const {
        if (prim_kind == 0) {  // triangle.rs:97-104
            const Triangle& t = tris[obj];
            float u = open01(r.v[1]);
            float v = open01(r.v[2]);
            v = (1.0f - u) * v;
            V3 e1 = t.p1 - t.p0, e2 = t.p2 - t.p0;
            return (t.p0 + u * e1) + v * e2;
        }
        // sphere.rs:98-118 (volume sample; x and y both use cos(phi), as written)
        const Sphere& s = spheres[obj];
        float phi = 0.0f + (2.0f * PI - 0.0f) * half_open01(r.v[1]);
        float cos_theta = -1.0f + (1.0f - -1.0f) * half_open01(r.v[2]);
        float u = 0.0f + (1.0f - 0.0f) * half_open01(r.v[3]);
        float theta = math.acos(cos_theta);
        float rr = s.radius * math.cbrt(u);
        float st, ct, sp, cp;
        math.sincos(theta, &st, &ct);
        math.sincos(phi, &sp, &cp);
        float x = rr * st * cp;
        float y = rr * st * cp;
        float z = rr * ct;
        return v3(x + s.position.x, y + s.position.y, z + s.position.z);
    }

    // ---- BVH build (bvh.rs:25-179)
    bool surface_area_heuristic(const AABB& bounds, size_t first, size_t count, int* axis_out, float* pivot_out) const {
        float sah_parent = bounds.area() * (float)count;  // bvh.rs:105
        AABB centre_bound = AABB::empty();
        for (size_t i = first; i < first + count; ++i) centre_bound = centre_bound.extent(obj_centre[indices[i]]);
        V3 axis_length = centre_bound.size();
        V3 axis_delta = axis_length / 8.0f;
        AABB bins[24];
        uint32_t prims[24];
        for (int b = 0; b < 24; ++b) { bins[b] = AABB::empty(); prims[b] = 0; }
        for (size_t i = first; i < first + count; ++i) {
            uint64_t obj = indices[i];
            V3 centre = obj_centre[obj];
            const AABB& bound = obj_bounds[obj];
            for (int axis = 0; axis < 3; ++axis) {
                float q = (centre[axis] - centre_bound.min[axis]) / axis_delta[axis];
                // `.max(0.).min(7.) as usize`: NaN -> 0 via max, then saturating cast
                float c = fmin_(fmax_(q, 0.0f), 7.0f);
                int bin = axis * 8 + (int)c;
                bins[bin] = bins[bin].combine(bound);
                prims[bin] += 1;
            }
        }
        int min_axis = -1, min_bin = -1;
        float min_sah = sah_parent;
        for (int axis = 0; axis < 3; ++axis) {
            int off = axis * 8;
            AABB bounds_right[8];
            uint32_t prims_right[8];
            bounds_right[7] = bins[off + 7];
            prims_right[7] = prims[off + 7];
            for (int b = 6; b >= 0; --b) {
                bounds_right[b] = bounds_right[b + 1].combine(bins[off + b]);
                prims_right[b] = prims_right[b + 1] + prims[off + b];
            }
            AABB bounds_left = AABB::empty();
            uint32_t prims_left = 0;
            for (int b = 0; b < 8; ++b) {
                bounds_left = bounds_left.combine(bins[off + b]);
                prims_left += prims[off + b];
                float sah = (float)prims_left * bounds_left.area() + (float)prims_right[b] * bounds_right[b].area();
                if (sah < min_sah) { min_axis = axis; min_bin = b; min_sah = sah; }
            }
        }
        if (min_axis < 0) return false;
        *axis_out = min_axis;
        *pivot_out = centre_bound.min[min_axis] + (float)(1 + min_bin) * axis_delta[min_axis];  // bvh.rs:166
        return true;
    }
    bool partition(size_t node_index) {  // bvh.rs:63-103
        AABB bounds = nodes[node_index].bounds;
        size_t first = nodes[node_index].left_first;
        size_t count = nodes[node_index].count;
        int axis;
        float pivot;
        if (!surface_area_heuristic(bounds, first, count, &axis, &pivot)) return false;
        AABB left_bound = AABB::empty(), right_bound = AABB::empty();
        size_t pivot_index = first;
        for (size_t index = first; index < first + count; ++index) {
            const AABB& bound = obj_bounds[indices[index]];
            float c = obj_centre[indices[index]][axis];
            if (c <= pivot) {
                left_bound = left_bound.combine(bound);
                std::swap(indices[pivot_index], indices[index]);
                pivot_index += 1;
            } else {
                right_bound = right_bound.combine(bound);
            }
        }
        size_t left_index = nodes.size();
        nodes[node_index].left_first = (uint32_t)left_index;
        nodes[node_index].count = 0;
        uint32_t left_count = (uint32_t)(pivot_index - first);
        nodes.push_back({left_bound, (uint32_t)first, left_count});
        nodes.push_back({right_bound, (uint32_t)pivot_index, (uint32_t)(count - left_count)});
        if (left_count == 0 || left_count == count) hazard = true;  // see SURVEY 3.5 "Hazard"
        return true;
    }
    int build() {  // bvh.rs:25-62
        size_t n = n_prims();
        if (n == 0 || prim_kind < 0) return PT_ERR_INVALID;
        for (size_t i = 0; i < n; ++i) {
            uint32_t m = prim_kind == 0 ? tris[i].material : spheres[i].material;
            if (m >= mats.size()) return PT_ERR_INVALID;
        }
        indices.resize(n);
        lights.clear();
        obj_bounds.resize(n);
        obj_centre.resize(n);
        AABB root = AABB::empty();
        for (size_t i = 0; i < n; ++i) {
            indices[i] = i;
            if (prim_is_light(i, nullptr)) lights.push_back(i);
            obj_bounds[i] = prim_bounds(i);
            obj_centre[i] = prim_centre(i);
            root = root.combine(obj_bounds[i]);
        }
        nodes.clear();
        nodes.reserve(2 * n);
        nodes.push_back({root, 0, (uint32_t)n});
        hazard = false;
        std::vector<size_t> stack{0};
        std::vector<uint32_t> depth_of(1, 1);
        max_depth = 1;
        while (!stack.empty()) {
            size_t ni = stack.back();
            stack.pop_back();
            if (nodes[ni].count > 2 && partition(ni)) {
                if (hazard) return PT_ERR_BVH;  // SURVEY 3.5 "Hazard": the reference would split the full child for ever
                size_t left = nodes[ni].left_first;
                depth_of.resize(nodes.size());
                depth_of[left] = depth_of[left + 1] = depth_of[ni] + 1;
                max_depth = std::max(max_depth, depth_of[left]);
                stack.push_back(left + 1);
                stack.push_back(left);
            }
        }
        built = true;
        return hazard ? PT_ERR_BVH : PT_OK;
    }

    // ---- traversal (bvh.rs:180-271), literal: node stack, re-test on pop, last accepted hit wins
    bool traverse(Ray& ray, bool any, Intersection* out, Counters& c, std::vector<uint32_t>& node_stack) const {
        bool found = false;
        node_stack.clear();
        node_stack.push_back(0);
        c.node_tests += 1;
        size_t far_depth = 0, far_max = 0;  // depth of a stack that only holds postponed (far) nodes
        (any ? c.rays_any : c.rays_closest) += 1;
        while (!node_stack.empty()) {
            c.max_stack = std::max<uint64_t>(c.max_stack, node_stack.size());
            uint32_t ni = node_stack.back();
            node_stack.pop_back();
            const BVHNode& node = nodes[ni];
            float t0, t1;
            c.aabb_calls += 1;
            if (!node.bounds.intersect(ray, &t0, &t1)) continue;
            if (node.count != 0) {
                for (uint32_t i = node.left_first; i < node.left_first + node.count; ++i) {
                    c.prim_tests += 1;
                    Intersection tmp;
                    if (prim_intersect(indices[i], ray, &tmp)) {
                        *out = tmp;
                        found = true;
                        if (any) goto done;
                    }
                }
            } else {
                uint32_t left = node.left_first;
                float tlmin, tlmax, trmin, trmax;
                c.node_tests += 2;
                c.aabb_calls += 2;
                bool hl = nodes[left].bounds.intersect(ray, &tlmin, &tlmax);
                bool hr = nodes[left + 1].bounds.intersect(ray, &trmin, &trmax);
                if (hl && hr) {
                    if (tlmin <= trmin) { node_stack.push_back(left + 1); node_stack.push_back(left); }
                    else { node_stack.push_back(left); node_stack.push_back(left + 1); }
                } else if (hl) node_stack.push_back(left);
                else if (hr) node_stack.push_back(left + 1);
            }
            // size of the stack once the next node to visit is taken off it
            far_depth = node_stack.empty() ? 0 : node_stack.size() - 1;
            far_max = std::max(far_max, far_depth);
        }
    done:
        c.stack_hist[std::min<size_t>(far_max, 63)] += 1;
        return found;
    }

    // ---- scene.rs:162-169
    V3 sample_skybox(V3 d) const {
        float fu = 2500.0f * 0.5f * (1.0f + math.atan2(d.x, -d.z) * FRAC_1_PI);
        float fv = 1250.0f * (math.acos(d.y) * FRAC_1_PI);
        // Rust `as usize`: NaN -> 0, negative -> 0, saturating
        uint64_t u = (fu != fu || fu <= 0.0f) ? 0 : (uint64_t)fu;
        uint64_t v = (fv != fv || fv <= 0.0f) ? 0 : (uint64_t)fv;
        uint64_t idx = u + v * 2500;
        if (idx * 3 + 2 >= sky.size()) return v3(0, 0, 0);  // unreachable for finite inputs (u<=2500, v<=1250)
        return v3(sky[idx * 3 + 0], sky[idx * 3 + 1], sky[idx * 3 + 2]);
    }

    // ---- camera.rs:39-86
    static bool refract(V3 direction, V3 normal, float n1, float n2, V3* out) {
        float div = n1 / n2;
        float cosi = dot(direction, normal);
        float sin_t2 = div * div * (1.0f - cosi * cosi);
        if (sin_t2 <= 1.0f) {
            *out = div * direction - (div * cosi + std::sqrt(1.0f - sin_t2)) * normal;
            return true;
        }
        return false;
    }
    static V3 reflect(V3 direction, V3 normal) { return direction - (2.0f * dot(direction, normal)) * normal; }
    static float schlick(V3 direction, V3 normal, float n1, float n2) {
        float div = (n1 - n2) / (n1 + n2);
        float r0 = div * div;
        float cosi = -dot(direction, normal);
        float x = 1.0f - cosi;
        float x2 = x * x;
        float x5 = x * (x2 * x2);  // powi(5) == __powisf2: r = x; a = x^2; a = x^4; r *= a
        return r0 + (1.0f - r0) * x5;
    }
    static V3 from_tangent_to_local(V3 normal, V3 tangent) {
        V3 t = normalize(cross(normal, std::fabs(normal.x) > 0.99f ? v3(0, 1, 0) : v3(1, 0, 0)));
        V3 b = cross(normal, t);
        return (tangent.x * t + tangent.y * b) + tangent.z * normal;
    }
    V3 cosine_weighted_diffuse(V3 normal, float r0, float r1) const {
        float r = std::sqrt(r0);
        float phi = 2.0f * PI * r1;
        float s, c;
        math.sincos(phi, &s, &c);
        return from_tangent_to_local(normal, v3(c * r, s * r, std::sqrt(1.0f - r0)));
    }

    // ---- camera.rs:214-270 (tail recursion written as a loop)
    float focus(Ray ray, float distance, int depth, Counters& c, std::vector<uint32_t>& st) const {
        while (true) {
            if (depth == 0) return distance;
            Intersection it;
            if (!traverse(ray, false, &it, c, st)) return distance == 0.0f ? INF : distance;
            distance = distance + ray.distance;
            const pt_material& m = mats[it.material];
            if (m.kind == PT_MAT_DIFFUSE && m.a > 0.5f) {
                V3 r = reflect(ray.direction, it.normal);
                V3 p = ray.intersection();
                ray.reset(p, r, INF);
            } else if (m.kind == PT_MAT_DIELECTRIC) {
                V3 normal = it.normal;
                V3 nf = it.inside ? -normal : normal;
                float na = it.inside ? m.b : m.a, nb = it.inside ? m.a : m.b;
                V3 p = ray.intersection();
                V3 refr;
                if (refract(ray.direction, nf, na, nb, &refr)) {
                    float sr = schlick(ray.direction, nf, na, nb);
                    if (0.5f < sr) ray.reset(p, reflect(ray.direction, nf), INF);
                    else ray.reset(p, refr, INF);
                } else {
                    ray.reset(p, reflect(ray.direction, nf), INF);
                }
            } else {
                return distance == 0.0f ? INF : distance;
            }
            depth -= 1;
        }
    }
    void update() {  // camera.rs:271-292
        direction = normalize(target - origin);
        right = cross(v3(0, 1, 0), direction);
        up = cross(direction, right);
        Ray ray{origin, direction, INF};
        float aspect = (float)width / (float)height;
        Counters c;
        std::vector<uint32_t> st;
        focal_distance = fmin_(20.0f, focus(ray, 0.0f, 5, c, st));
        V3 cc = origin + focal_distance * direction;
        p1 = (cc + (-0.5f * focal_distance * aspect) * right) + (0.5f * focal_distance) * up;
        p2 = (cc + (0.5f * focal_distance * aspect) * right) + (0.5f * focal_distance) * up;
        p3 = (cc + (-0.5f * focal_distance * aspect) * right) + (-0.5f * focal_distance) * up;
    }

    // ---- camera.rs:426-449
    Ray generate(uint32_t x, uint32_t y, const RngKey& k) const {
        U4 r = philox4x32_10(k.k0, k.k1, k.pixel, k.sample, 0, 0);
        float r0 = closed01(r.v[0]);
        float r1 = closed01(r.v[1]);
        float r2 = closed01(r.v[2]) - 0.5f;
        float r3 = closed01(r.v[3]) - 0.5f;
        float u = ((float)x + r0) / (float)width;
        float v = ((float)y + r1) / (float)height;
        V3 tgt = (p1 + u * (p2 - p1)) + v * (p3 - p1);
        V3 org = origin + lens_size * (r2 * right + r3 * up);
        V3 dir = normalize(tgt - org);
        return Ray{org, dir, INF};
    }

    // The device's rule for nee_mode 1, restated: the NEE block (camera.rs:314-338) adds weight * X with weight =
    // min(1 - speculaty, 0).  Where the weight is 0 it adds +-0 — unless X is not finite (0 * inf = NaN).  X = transport *
    // (cos_i / mis) * (light colour * brdf); cos_i / mis <= pi when both pdfs are >= 0, so X is finite while |transport| <=
    // FLT_MAX / (3.25 * lcb), lcb = the largest |light colour * diffuse colour / pi| of the scene.  Lights whose area is not
    // a finite number >= 0, or an lcb that is not finite, make every block run (nee_literal).
    void nee_rule() {
        nee_literal = false;
        float lcb = 0.0f;
        for (const pt_material& e : mats) {
            if (e.kind != PT_MAT_EMISSIVE) continue;
            for (const pt_material& d : mats) {
                if (d.kind != PT_MAT_DIFFUSE) continue;
                for (int ch = 0; ch < 3; ++ch) {
                    float v = std::fabs(e.rgb[ch] * (FRAC_1_PI * d.rgb[ch]));
                    if (!(v <= 3.402823466e38f)) nee_literal = true;
                    else lcb = std::max(lcb, v);
                }
            }
        }
        double t = lcb > 0.0f ? 3.4028234663852886e38 / (3.25 * (double)lcb) : 3.4028234663852886e38;
        nee_thr = t >= 3.4028234663852886e38 ? 3.402823466e38f : (float)t;
        for (uint64_t l : lights) {
            float a = prim_area(l);
            if (!(a >= 0.0f && a <= 3.402823466e38f)) nee_literal = true;
        }
    }

    // ---- camera.rs:295-423
    // nee_mode 0: literal (the NEE block runs whenever the scene has a light), camera.rs:314-338 as written
    // nee_mode 1: the device's rule — the block runs where its weight is non-zero, where the transport is beyond nee_thr
    //             (inf and NaN included) and everywhere in a nee_literal scene; same sums as mode 0 (see nee_rule),
    //             fewer any-hit rays
    // nee_mode 2: round 2's rule, kept as the negative control of tests/test_nee_rule_cpu.py — the block runs only where its
    //             weight is non-zero; differs from the reference where a transport is not finite
    V3 sample(Ray& ray, uint32_t depth, const RngKey& k, int nee_mode, Counters& c, std::vector<uint32_t>& st) const {
        V3 acc = v3(0, 0, 0);
        V3 transport = v3(1, 1, 1);
        for (uint32_t it = 0; it < depth; ++it) {
            Intersection hit;
            if (!traverse(ray, false, &hit, c, st)) {
                acc = acc + mul_ew(transport, 0.1f * sample_skybox(ray.direction));
                break;
            }
            V3 normal = hit.normal;
            bool inside = hit.inside;
            V3 P = ray.intersection();
            const pt_material& m = mats[hit.material];
            V3 color = v3(m.rgb[0], m.rgb[1], m.rgb[2]);
            if (m.kind == PT_MAT_EMISSIVE) {
                acc = acc + mul_ew(transport, color);
                break;
            } else if (m.kind == PT_MAT_DIFFUSE) {
                float speculaty = m.a;
                if (inside) break;
                float nee_w = fmin_(1.0f - speculaty, 0.0f);
                bool beyond = !(std::fabs(transport.x) <= nee_thr && std::fabs(transport.y) <= nee_thr && std::fabs(transport.z) <= nee_thr);
                if (!lights.empty() && (nee_mode == 0 || nee_w != 0.0f || (nee_mode == 1 && (nee_literal || beyond)))) {
                    U4 rn = philox4x32_10(k.k0, k.k1, k.pixel, k.sample, 1 + it, 1);
                    // bvh.rs:273-284: Range::new(0, light_count).ind_sample() — rand 0.3.15 rejects draws at or above
                    // the largest multiple of the range (`zone`) and returns low + v % range.  The generator is the
                    // counter-based stream here, so the draws are its 32-bit words: first word 0 of the NEE block,
                    // then word 0 of blocks (iteration, stream 2, 3, ...) for as long as a draw is rejected.
                    size_t nl = lights.size();
                    uint32_t pick = rn.v[0];
                    const uint32_t zone = 0xffffffffu - 0xffffffffu % (uint32_t)nl;
                    for (uint32_t j = 0; pick >= zone; ++j) pick = philox4x32_10(k.k0, k.k1, k.pixel, k.sample, 1 + it, 2 + j).v[0];
                    uint64_t light = lights[pick % nl];
                    V3 Q = prim_random_point(light, rn);
                    V3 light_dir = normalize(Q - P);
                    Ray god{P + (20.0f * EPS) * light_dir, light_dir, INF};
                    Intersection il;
                    c.prim_tests += 1;
                    if (prim_intersect(light, god, &il)) {
                        float cos_i = dot(normal, light_dir);
                        float cos_l = -dot(il.normal, light_dir);
                        if (cos_i > 0.0f && cos_l > 0.0f) {
                            god.distance -= EPS;
                            Intersection tmp;
                            if (!traverse(god, true, &tmp, c, st)) {
                                V3 brdf = FRAC_1_PI * color;
                                float area = prim_area(light);
                                V3 lc = v3(0, 0, 0);
                                prim_is_light(light, &lc);
                                float solid = (cos_l * area) / (god.distance * god.distance);
                                float light_pdf = 1.0f / solid;
                                float hemi_pdf = FRAC_1_PI * cos_i;
                                float mis = light_pdf + hemi_pdf;
                                V3 nee = (nee_w * (float)nl) * mul_ew(transport, (cos_i / mis) * mul_ew(lc, brdf));
                                acc = acc + nee;
                            }
                        }
                    }
                }
                U4 r = philox4x32_10(k.k0, k.k1, k.pixel, k.sample, 1 + it, 0);
                float r0 = closed01(r.v[0]);
                if (r0 < speculaty) {
                    V3 rd = reflect(ray.direction, normal);
                    transport = mul_ew(transport, color);
                    ray.reset(P, rd, INF);
                } else {
                    float rr = closed01(r.v[1]);
                    float tm = transport.x > transport.y ? transport.x : transport.y;  // cgmath partial_max
                    tm = tm > transport.z ? tm : transport.z;
                    float survival = fmax_(tm, 0.1f);
                    if (rr < survival) transport = transport / survival;
                    else break;
                    V3 dd = cosine_weighted_diffuse(normal, closed01(r.v[2]), closed01(r.v[3]));
                    ray.reset(P, dd, INF);
                    float cos_i = dot(dd, normal);
                    V3 brdf = FRAC_1_PI * color;
                    float hemi_pdf = FRAC_1_PI * cos_i;
                    float light_pdf = 0.0f;
                    Intersection la;
                    // look-ahead trace (camera.rs:362); its result is dropped (shadowed binding),
                    // only ray.distance stays shortened for the next iteration's trace (Q3)
                    if (traverse(ray, false, &la, c, st) && mats[la.material].kind == PT_MAT_EMISSIVE) {
                        float cos_l = -dot(la.normal, ray.direction);
                        float solid = (cos_l * la.area) / (ray.distance * ray.distance);
                        light_pdf = 1.0f / solid;
                    }
                    float mis = light_pdf + hemi_pdf;
                    transport = mul_ew(transport, (cos_i / mis) * brdf);
                }
            } else {  // Dielectric, camera.rs:375-417
                float n1 = m.a, n2 = m.b;
                if (inside) {
                    V3 ab = (v3(-1, -1, -1) + color) * ray.distance;
                    transport = mul_ew(transport, v3(math.exp(ab.x), math.exp(ab.y), math.exp(ab.z)));
                }
                V3 nf = inside ? -normal : normal;
                float na = inside ? n2 : n1, nb = inside ? n1 : n2;
                V3 refr;
                if (refract(ray.direction, nf, na, nb, &refr)) {
                    float sr = schlick(ray.direction, nf, na, nb);
                    U4 r = philox4x32_10(k.k0, k.k1, k.pixel, k.sample, 1 + it, 0);
                    float r0 = closed01(r.v[0]);
                    if (r0 < sr) {
                        if (!inside) transport = mul_ew(transport, color);
                        ray.reset(P, reflect(ray.direction, nf), INF);
                    } else {
                        ray.reset(P, refr, INF);
                    }
                } else {
                    ray.reset(P, reflect(ray.direction, nf), INF);
                }
            }
        }
        return acc;
    }
};

// ---------------------------------------------------------------------------- C API (ctypes)
extern "C" {

orc_ctx* orc_create() { return new orc_ctx(); }
void orc_destroy(orc_ctx* c) { delete c; }
void orc_set_math(orc_ctx* c, int mode) { c->math.mode = mode; }

int orc_set_materials(orc_ctx* c, const pt_material* m, uint32_t n) {
    c->mats.assign(m, m + n);
    c->built = false;
    return 0;
}
int orc_set_triangles(orc_ctx* c, const float* pos, const float* nrm, const uint32_t* mat, uint64_t n) {
    c->prim_kind = 0;
    c->spheres.clear();
    c->tris.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = pos + 9 * i;
        const float* q = nrm + 9 * i;
        c->tris[i] = Triangle{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]},
                              {q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}, mat[i]};
    }
    c->built = false;
    return 0;
}
int orc_set_spheres(orc_ctx* c, const float* cr, const uint32_t* mat, uint64_t n) {
    c->prim_kind = 1;
    c->tris.clear();
    c->spheres.resize(n);
    for (uint64_t i = 0; i < n; ++i) c->spheres[i] = Sphere{{cr[4 * i], cr[4 * i + 1], cr[4 * i + 2]}, cr[4 * i + 3], mat[i]};
    c->built = false;
    return 0;
}
int orc_set_sky(orc_ctx* c, const float* rgb, uint32_t w, uint32_t h) {
    // reference: floats = bytes.len()/2 zero-initialised, first half filled (scene.rs:174-181)
    size_t n = (size_t)w * h * 3;
    size_t need = (size_t)3 * (2500 + (size_t)1250 * 2500) + 3;
    c->sky.assign(std::max(2 * n, need), 0.0f);
    std::memcpy(c->sky.data(), rgb, n * sizeof(float));
    c->sky_w = w;
    c->sky_h = h;
    return 0;
}
int orc_build(orc_ctx* c) {
    int rc = c->build();
    if (rc == PT_OK) c->nee_rule();
    return rc;
}
int orc_bvh_sizes(orc_ctx* c, uint64_t* n_nodes, uint64_t* n_indices, uint32_t* max_depth) {
    *n_nodes = c->nodes.size();
    *n_indices = c->indices.size();
    *max_depth = c->max_depth;
    return 0;
}
int orc_get_bvh(orc_ctx* c, pt_bvh_node* nodes, uint64_t* indices) {
    for (size_t i = 0; i < c->nodes.size(); ++i) {
        const BVHNode& n = c->nodes[i];
        nodes[i] = pt_bvh_node{{n.bounds.min.x, n.bounds.min.y, n.bounds.min.z}, n.left_first,
                               {n.bounds.max.x, n.bounds.max.y, n.bounds.max.z}, n.count};
    }
    std::memcpy(indices, c->indices.data(), c->indices.size() * sizeof(uint64_t));
    return 0;
}
uint64_t orc_num_lights(orc_ctx* c) { return c->lights.size(); }

int orc_set_camera(orc_ctx* c, const float* o, const float* t, uint32_t w, uint32_t h, float lens) {
    if (!c->built) return PT_ERR_STATE;
    c->origin = v3(o[0], o[1], o[2]);
    c->target = v3(t[0], t[1], t[2]);
    c->width = w;
    c->height = h;
    c->lens_size = lens;
    c->update();
    c->have_camera = true;
    c->accum.assign((size_t)w * h, v3(0, 0, 0));
    c->spp = 0.f;
    return 0;
}
static void put3(float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
int orc_get_camera(orc_ctx* c, pt_camera_state* s) {
    put3(s->origin, c->origin); put3(s->target, c->target); put3(s->direction, c->direction);
    put3(s->right, c->right); put3(s->up, c->up);
    put3(s->p1, c->p1); put3(s->p2, c->p2); put3(s->p3, c->p3);
    s->focal_distance = c->focal_distance;
    s->lens_size = c->lens_size;
    s->width = c->width;
    s->height = c->height;
    return 0;
}
int orc_generate(orc_ctx* c, const uint32_t* xy, uint64_t n, uint32_t sample, uint64_t seed, pt_ray* rays) {
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t x = xy[2 * i], y = xy[2 * i + 1];
        RngKey k{(uint32_t)seed, (uint32_t)(seed >> 32), y * c->width + x, sample};
        Ray r = c->generate(x, y, k);
        put3(rays[i].o, r.origin);
        put3(rays[i].d, r.direction);
        rays[i].tmax = r.distance;
    }
    return 0;
}
// counters_out: optional 5 x u64 {rays, node_tests, prim_tests, aabb_calls, max_stack}
int orc_trace(orc_ctx* c, const pt_ray* rays, uint64_t n, int any_hit, pt_hit* hits, uint64_t* counters_out) {
    if (!c->built) return PT_ERR_STATE;
    Counters cnt;
    std::vector<uint32_t> st;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r{{rays[i].o[0], rays[i].o[1], rays[i].o[2]}, {rays[i].d[0], rays[i].d[1], rays[i].d[2]}, rays[i].tmax};
        Intersection it;
        bool h = c->traverse(r, any_hit != 0, &it, cnt, st);
        hits[i].prim = h ? it.prim : PT_MISS;
        hits[i].t = r.distance;
        hits[i].inside = h ? (it.inside ? 1u : 0u) : 0u;
    }
    if (counters_out) {
        counters_out[0] = cnt.rays_closest + cnt.rays_any;
        counters_out[1] = cnt.node_tests;
        counters_out[2] = cnt.prim_tests;
        counters_out[3] = cnt.aabb_calls;
        counters_out[4] = cnt.max_stack;
    }
    return 0;
}
// per-traversal max depth histogram (64 bins) of a far-only stack, for sizing the LDS stack
int orc_trace_stack_hist(orc_ctx* c, const pt_ray* rays, uint64_t n, uint64_t* hist64) {
    Counters cnt;
    std::vector<uint32_t> st;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r{{rays[i].o[0], rays[i].o[1], rays[i].o[2]}, {rays[i].d[0], rays[i].d[1], rays[i].d[2]}, rays[i].tmax};
        Intersection it;
        c->traverse(r, false, &it, cnt, st);
    }
    std::memcpy(hist64, cnt.stack_hist, sizeof(cnt.stack_hist));
    return 0;
}
int orc_sample_skybox(orc_ctx* c, const float* dirs, uint64_t n, float* rgb) {
    for (uint64_t i = 0; i < n; ++i) put3(rgb + 3 * i, c->sample_skybox(v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2])));
    return 0;
}

// camera.sample for chosen pixels of one sample index (one radiance triple per pixel)
int orc_sample_pixels(orc_ctx* c, const uint32_t* xy, uint64_t n, uint32_t sample, uint32_t depth, uint64_t seed,
                      int nee_mode, float* rgb) {
    if (!c->have_camera) return PT_ERR_STATE;
    Counters cnt;
    std::vector<uint32_t> st;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t x = xy[2 * i], y = xy[2 * i + 1];
        RngKey k{(uint32_t)seed, (uint32_t)(seed >> 32), y * c->width + x, sample};
        Ray r = c->generate(x, y, k);
        put3(rgb + 3 * i, c->sample(r, depth, k, nee_mode, cnt, st));
    }
    return 0;
}

// The frame loop of main.rs:143-170 run `spp` times: static row bands over `nthreads` threads
// (band = H / nthreads rows as main.rs:154-155; the H % nthreads leftover rows the reference
// drops are given to the last band so the image is complete).
int orc_render(orc_ctx* c, uint32_t spp, uint32_t depth, uint64_t seed, uint32_t sample_begin, uint32_t nthreads,
               int nee_mode) {
    if (!c->have_camera) return PT_ERR_STATE;
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    nthreads = std::min<uint32_t>(nthreads, c->height);
    uint32_t W = c->width, H = c->height;
    std::vector<Counters> per(nthreads);
    for (uint32_t s = 0; s < spp; ++s) {
        uint32_t sample = sample_begin + s;
        std::vector<std::thread> pool;
        uint32_t band = H / nthreads;
        for (uint32_t core = 0; core < nthreads; ++core) {
            uint32_t y0 = core * band, y1 = core + 1 == nthreads ? H : y0 + band;
            pool.emplace_back([=, &per]() {
                std::vector<uint32_t> st;
                Counters& cnt = per[core];
                for (uint32_t y = y0; y < y1; ++y)
                    for (uint32_t x = 0; x < W; ++x) {
                        RngKey k{(uint32_t)seed, (uint32_t)(seed >> 32), y * W + x, sample};
                        Ray r = c->generate(x, y, k);
                        cnt.paths += 1;
                        V3 v = c->sample(r, depth, k, nee_mode, cnt, st);
                        V3& a = c->accum[(size_t)y * W + x];
                        a = a + v;
                    }
            });
        }
        for (auto& t : pool) t.join();
        c->spp += 1.0f;
    }
    for (auto& p : per) c->counters.add(p);
    return 0;
}
int orc_reset_accum(orc_ctx* c) {
    std::fill(c->accum.begin(), c->accum.end(), v3(0, 0, 0));
    c->spp = 0.f;
    return 0;
}
int orc_read_accum(orc_ctx* c, float* rgb, float* spp) {
    std::memcpy(rgb, c->accum.data(), c->accum.size() * sizeof(V3));
    if (spp) *spp = c->spp;
    return 0;
}
// main.rs:71-75: (255f32).min(256 * 1.5 * sqrt(x)) as u8   (NaN -> 255 via min; `as u8` truncates, saturates)
static uint8_t to_u8(float x) {
    float v = fmin_(255.0f, 256.0f * 1.5f * std::sqrt(x));
    if (v != v || v <= 0.0f) return 0;
    return (uint8_t)v;
}
int orc_read_rgb8(orc_ctx* c, uint8_t* out, uint64_t pitch) {
    float scale = 1.0f / c->spp;  // main.rs:145
    for (uint32_t y = 0; y < c->height; ++y)
        for (uint32_t x = 0; x < c->width; ++x) {
            V3 v = scale * c->accum[(size_t)y * c->width + x];
            uint8_t* p = out + y * pitch + 3 * x;
            p[0] = to_u8(v.x); p[1] = to_u8(v.y); p[2] = to_u8(v.z);
        }
    return 0;
}
int orc_get_counters(orc_ctx* c, pt_counters* out, uint64_t* stack_hist64) {
    std::memset(out, 0, sizeof(*out));
    out->paths = c->counters.paths;
    out->rays_closest = c->counters.rays_closest;
    out->rays_any = c->counters.rays_any;
    out->node_tests = c->counters.node_tests;
    out->prim_tests = c->counters.prim_tests;
    if (stack_hist64) std::memcpy(stack_hist64, c->counters.stack_hist, sizeof(c->counters.stack_hist));
    return 0;
}
int orc_reset_counters(orc_ctx* c) { c->counters = Counters(); return 0; }

// ---- unit seams for the reference's known-answer tests
int orc_aabb_intersect(const float* mn, const float* mx, const pt_ray* r, float* tmin, float* tmax) {
    AABB b{{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
    Ray ray{{r->o[0], r->o[1], r->o[2]}, {r->d[0], r->d[1], r->d[2]}, r->tmax};
    return b.intersect(ray, tmin, tmax) ? 1 : 0;
}
// out: t, u, v, a, inside, nx, ny, nz, area
int orc_triangle_intersect(const float* p, const float* n, const pt_ray* r, float* out) {
    Triangle t{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}, {n[0], n[1], n[2]}, {n[3], n[4], n[5]}, {n[6], n[7], n[8]}, 0};
    Ray ray{{r->o[0], r->o[1], r->o[2]}, {r->d[0], r->d[1], r->d[2]}, r->tmax};
    Intersection it;
    float uv[3];
    if (!triangle_intersect(t, ray, &it, uv)) return 0;
    out[0] = ray.distance; out[1] = uv[0]; out[2] = uv[1]; out[3] = uv[2];
    out[4] = it.inside ? 1.f : 0.f; out[5] = it.normal.x; out[6] = it.normal.y; out[7] = it.normal.z; out[8] = it.area;
    return 1;
}
// out: t, inside, nx, ny, nz, area
int orc_sphere_intersect(const float* cr, const pt_ray* r, float* out) {
    Sphere s{{cr[0], cr[1], cr[2]}, cr[3], 0};
    Ray ray{{r->o[0], r->o[1], r->o[2]}, {r->d[0], r->d[1], r->d[2]}, r->tmax};
    Intersection it;
    if (!sphere_intersect(s, ray, &it)) return 0;
    out[0] = ray.distance; out[1] = it.inside ? 1.f : 0.f; out[2] = it.normal.x; out[3] = it.normal.y; out[4] = it.normal.z; out[5] = it.area;
    return 1;
}
void orc_philox(const uint32_t* key, const uint32_t* ctr, uint32_t* out) {
    U4 r = philox4x32_10(key[0], key[1], ctr[0], ctr[1], ctr[2], ctr[3]);
    std::memcpy(out, r.v, 16);
}
// fn: 0 sin 1 cos 2 exp 3 acos 4 atan2(x, y); mode 0 det 1 libm
void orc_math(int fn, int mode, const float* x, const float* y, uint64_t n, float* out) {
    Math m;
    m.mode = mode;
    for (uint64_t i = 0; i < n; ++i) {
        float s, c;
        switch (fn) {
            case 0: m.sincos(x[i], &s, &c); out[i] = s; break;
            case 1: m.sincos(x[i], &s, &c); out[i] = c; break;
            case 2: out[i] = m.exp(x[i]); break;
            case 3: out[i] = m.acos(x[i]); break;
            case 5: out[i] = m.cbrt(x[i]); break;
            default: out[i] = m.atan2(x[i], y[i]); break;
        }
    }
}
// BSDF helper seams (camera.rs:39-86): fn 0 refract(d,n,n1,n2)->xyz,ok  1 reflect  2 schlick  3 cosine_weighted_diffuse(n,r0,r1)
void orc_bsdf(orc_ctx* c, int fn, const float* d, const float* n, float a, float b, float* out) {
    V3 dd = v3(d[0], d[1], d[2]), nn = v3(n[0], n[1], n[2]);
    V3 r = v3(0, 0, 0);
    switch (fn) {
        case 0: out[3] = orc_ctx::refract(dd, nn, a, b, &r) ? 1.f : 0.f; put3(out, r); break;
        case 1: put3(out, orc_ctx::reflect(dd, nn)); break;
        case 2: out[0] = orc_ctx::schlick(dd, nn, a, b); break;
        default: put3(out, c->cosine_weighted_diffuse(nn, a, b)); break;
    }
}
uint8_t orc_tonemap(float x) { return to_u8(x); }

}  // extern "C"
