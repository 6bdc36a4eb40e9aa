// pt_device.h — fp32 building blocks of the hot path for gfx950 (also compilable on the host).
//
// Every formula keeps the operation order of the reference so that results are bit-identical
// to a plain IEEE-754 evaluation of the Rust source (no FMA contraction: build with
// -ffp-contract=off, never with fast-math).  cgmath 0.12 conventions: dot = (x*x'+y*y')+z*z',
// normalize = v * (1/|v|).  File:line citations are into /root/reference.
#pragma once

#include <stdint.h>

#include "../../include/ptcore.h"
#include "../../include/ptcore_detmath.h"
#include "pt_pixelmap.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_DEV __host__ __device__ __forceinline__
#else
#define PT_DEV inline
#endif

#pragma clang fp contract(off)

namespace pt {

constexpr float kInf = __builtin_huge_valf();
constexpr float kEps = 1.1920929e-7f;  // f32::EPSILON
constexpr float kFrac1Pi = 0.318309886183790671538f;
constexpr float kPi = 3.14159265358979323846f;

struct V3 {
    float x, y, z;
};
PT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
PT_DEV V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
PT_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_DEV V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PT_DEV V3 mul_ew(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_DEV float magnitude(V3 a) { return __builtin_sqrtf(dot(a, a)); }
PT_DEV V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }
// Rust f32::min / f32::max: the non-NaN operand wins (IEEE minNum / maxNum)
PT_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

// ---- Philox4x32-10, counter = (pixel, sample, block, stream); layout documented in DESIGN.md
struct U4 {
    uint32_t v[4];
};
PT_DEV uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// Both halves of a 32x32 product.  PT_PHILOX_WIDE=1 takes them from one v_mad_u64_u32 (same issue cost as v_mul_hi_u32 alone,
// tools/intrate.hip) where the compiler emits v_mul_hi_u32 + v_mul_lo_u32.  Measured 0.5-1 % SLOWER on C3/C4/C5
// (profiles/r3_ab_philox_wide.txt: the kernel waits on memory, not on the multiplier, and the 64-bit results cost registers), so off.
#ifndef PT_PHILOX_WIDE
#define PT_PHILOX_WIDE 0
#endif
PT_DEV void mulwide32(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__) && PT_PHILOX_WIDE
    unsigned long long w;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w) : "v"(a), "v"(b) : "vcc");
    hi = (uint32_t)(w >> 32);
    lo = (uint32_t)w;
#else
    hi = mulhi32(a, b);
    lo = a * b;
#endif
}
PT_DEV U4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0, l0, h1, l1;
        mulwide32(0xD2511F53u, c0, h0, l0);
        mulwide32(0xCD9E8D57u, c2, h1, l1);
        c0 = h1 ^ c1 ^ k0;
        c1 = l1;
        c2 = h0 ^ c3 ^ k1;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{{c0, c1, c2, c3}};
}
PT_DEV float closed01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777215.0f); }
PT_DEV float open01(uint32_t x) { return ((float)(x >> 8) + 0.25f) * (1.0f / 16777216.0f); }
PT_DEV float half_open01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ---- AABB::intersect, src/primitive/aabb.rs:47-66.  rcp = 1/direction is hoisted out of the
// traversal loop (the reference recomputes the same three quotients on every call).
PT_DEV bool slab(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, V3 o, V3 rcp, float dist, float* tmin_out) {
    float ix0 = (mnx - o.x) * rcp.x, iy0 = (mny - o.y) * rcp.y, iz0 = (mnz - o.z) * rcp.z;
    float ix1 = (mxx - o.x) * rcp.x, iy1 = (mxy - o.y) * rcp.y, iz1 = (mxz - o.z) * rcp.z;
    float tmin = fmax_(fmax_(fmin_(ix0, ix1), fmin_(iy0, iy1)), fmin_(iz0, iz1));
    float tmax = fmin_(fmin_(fmax_(ix0, ix1), fmax_(iy0, iy1)), fmax_(iz0, iz1));
    *tmin_out = tmin;
    return !(tmax < 0.0f || tmin > tmax || dist < tmin);
}

// ---- Triangle::intersect, src/primitive/triangle.rs:41-72 (hit test part; edges precomputed,
// p1 - p0 and p2 - p0 are the same two subtractions the reference performs per call).
// Accepts without comparing t to ray.distance (quirk Q1).  a < 0 <=> Intersection::inside.
PT_DEV bool tri_hit(V3 p0, V3 e1, V3 e2, V3 o, V3 d, float* t_out, float* u_out, float* v_out, float* a_out) {
    V3 h = cross(d, e2);
    float a = dot(e1, h);
    if (a > -kEps && a < kEps) return false;
    float f = 1.0f / a;
    V3 s = o - p0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, e1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot(e2, q);
    if (t < 0.0f) return false;
    *t_out = t;
    *u_out = u;
    *v_out = v;
    *a_out = a;
    return true;
}
// Triangle::area, src/primitive/triangle.rs:105-113 (not the true area; reproduced as written)
PT_DEV float tri_area(V3 e1, V3 e2) {
    float len1 = magnitude(e1), len2 = magnitude(e2);
    e1 = e1 / len1;
    e2 = e2 / len2;
    return 0.5f * (1.0f - dot(e1, e2)) * len1 * len2;
}
// interpolated normal, src/primitive/triangle.rs:74
PT_DEV V3 tri_normal(V3 n0, V3 n1, V3 n2, float u, float v) {
    return normalize(((1.0f - u - v) * n0 + u * n1) + v * n2);
}

// ---- Sphere::intersect, src/primitive/sphere.rs:34-80 (does compare with ray.distance, >=)
PT_DEV bool sphere_hit(V3 c, float radius, V3 o, V3 d, float dist, float* t_out, bool* inside_out) {
    V3 L = c - o;
    float tca = dot(L, d);
    float d2 = dot(L, L) - tca * tca;
    if (d2 > radius * radius) return false;
    float thc = __builtin_sqrtf(radius * radius - d2);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 >= 0.0f) {
        if (t0 >= dist) return false;
        *t_out = t0;
        *inside_out = false;
        return true;
    } else if (t1 >= 0.0f) {
        if (t1 >= dist) return false;
        *t_out = t1;
        *inside_out = true;
        return true;
    }
    return false;
}
PT_DEV V3 sphere_normal(V3 c, V3 o, V3 d, float t) { return normalize(o + d * t - c); }
PT_DEV float sphere_area(float radius) { return kPi * radius * radius; }  // sphere.rs:119-121

// ---- BSDF helpers, src/camera.rs:39-86
PT_DEV bool refract(V3 direction, V3 normal, float n1, float n2, V3* out) {
    float div = n1 / n2;
    float cosi = dot(direction, normal);
    float sin_t2 = div * div * (1.0f - cosi * cosi);
    if (sin_t2 <= 1.0f) {
        *out = div * direction - (div * cosi + __builtin_sqrtf(1.0f - sin_t2)) * normal;
        return true;
    }
    return false;
}
PT_DEV V3 reflect(V3 direction, V3 normal) { return direction - (2.0f * dot(direction, normal)) * normal; }
PT_DEV float schlick(V3 direction, V3 normal, float n1, float n2) {
    float div = (n1 - n2) / (n1 + n2);
    float r0 = div * div;
    float cosi = -dot(direction, normal);
    float x = 1.0f - cosi;
    float x2 = x * x;
    float x5 = x * (x2 * x2);  // powi(5)
    return r0 + (1.0f - r0) * x5;
}
PT_DEV V3 from_tangent_to_local(V3 normal, V3 tangent) {
    V3 t = normalize(cross(normal, __builtin_fabsf(normal.x) > 0.99f ? v3(0.f, 1.f, 0.f) : v3(1.f, 0.f, 0.f)));
    V3 b = cross(normal, t);
    return (tangent.x * t + tangent.y * b) + tangent.z * normal;
}
PT_DEV V3 cosine_weighted_diffuse(V3 normal, float r0, float r1) {
    float r = __builtin_sqrtf(r0);
    float phi = 2.0f * kPi * r1;
    float s, c;
    pt_sincosf(phi, &s, &c);
    return from_tangent_to_local(normal, v3(c * r, s * r, __builtin_sqrtf(1.0f - r0)));
}
// Ray::reset origin push, src/ray.rs:29
PT_DEV V3 advance(V3 o, V3 d) { return o + (20.0f * kEps) * d; }

// Scene::sample_skybox texel index, src/scene.rs:163-165 (`as usize`: NaN / negative -> 0)
PT_DEV uint32_t sky_index(V3 d) {
    float fu = 2500.0f * 0.5f * (1.0f + pt_atan2f(d.x, -d.z) * kFrac1Pi);
    float fv = 1250.0f * (pt_acosf(d.y) * kFrac1Pi);
    uint32_t u = (fu != fu || fu <= 0.0f) ? 0u : (uint32_t)fu;
    uint32_t v = (fv != fv || fv <= 0.0f) ? 0u : (uint32_t)fv;
    return u + v * 2500u;
}
constexpr uint32_t kSkyMaxIndex = 2500u + 1250u * 2500u;  // largest index a finite direction can give

// vec_to_rgb channel, src/main.rs:71-75
PT_DEV uint8_t tonemap_channel(float c) {
    float v = fmin_(255.0f, 256.0f * 1.5f * __builtin_sqrtf(c));
    if (v != v || v <= 0.0f) return 0;
    return (uint8_t)v;
}

// n / d for n < 2^31 (pt_pixelmap.h)
PT_DEV uint32_t fastdiv(uint32_t n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return f.m ? (__umulhi(n, f.m) >> f.sh) : n;
#else
    return f.m ? (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.sh : n;
#endif
}

// slot index inside one sample -> pixel
PT_DEV bool map_pixel(const PixelMap& M, uint32_t idx, uint32_t* x_out, uint32_t* y_out) {
    const uint32_t per_tile = M.tile * M.tile;
    const uint32_t local_tile = fastdiv(idx, M.fd_per_tile), within = idx - local_tile * per_tile;
    const uint32_t T = local_tile * M.nranks + M.rank;
    const uint32_t ty = fastdiv(T, M.fd_tiles_x), tx = T - ty * M.tiles_x;
    const uint32_t sub = within >> 6, lane = within & 63u;
    const uint32_t subs = M.tile >> 3;
    const uint32_t sy = fastdiv(sub, M.fd_subs), sx = sub - sy * subs;
    const uint32_t x = tx * M.tile + sx * 8 + (lane & 7u);
    const uint32_t y = ty * M.tile + sy * 8 + (lane >> 3);
    *x_out = x;
    *y_out = y;
    return x < M.width && y < M.height;
}


}  // namespace pt
