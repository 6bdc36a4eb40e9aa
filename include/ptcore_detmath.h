/* ptcore_detmath.h — deterministic fp32 transcendentals shared by host and gfx950 device code.
 *
 * The reference calls Rust's f32::{sin,cos,exp,acos,atan2} (platform libm) in its shading code:
 *   cosine_weighted_diffuse  /root/reference/src/camera.rs:74-81   (sin, cos)
 *   Beer absorption          /root/reference/src/camera.rs:376-380 (exp)
 *   Scene::sample_skybox     /root/reference/src/scene.rs:162-169  (atan2, acos)
 * libm and the GPU's ocml disagree in the last ulp, which would make the CPU restatement and
 * the HIP kernels diverge on a handful of paths.  These five functions are built ONLY from
 * IEEE-754 fp32 +,-,*,/ and sqrt (all correctly rounded on x86-64 SSE and on gfx950 with
 * -ffp-contract=off), so every caller that includes this header gets bit-identical results.
 * Accuracy is <= 2 ulp over the ranges the path uses (Cephes single-precision kernels);
 * the oracle can also be switched to libm (ORC_MATH_LIBM) to show the image is statistically
 * unchanged by this substitution.
 *
 * Compile every translation unit that includes this with -ffp-contract=off and without
 * any fast-math flag.
 */
#ifndef PTCORE_DETMATH_H
#define PTCORE_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ static inline
#else
#define PT_HD static inline
#endif

#if defined(__cplusplus)
PT_HD uint32_t pt_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float pt_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
#else /* plain C hosts: type punning through a union is defined behaviour in C */
PT_HD uint32_t pt_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
PT_HD float pt_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
#endif

/* round-to-nearest-even to an integer-valued float, valid for |x| < 2^22; plain IEEE adds so
 * the result cannot depend on which rint implementation a compiler picks. */
PT_HD float pt_rint_small(float x) { return (x + 12582912.0f) - 12582912.0f; }

#define PT_PI_F 3.14159265358979323846f
#define PT_FRAC_1_PI_F 0.318309886183790671538f
#define PT_PI_2_F 1.57079632679489661923f
#define PT_PI_4_F 0.78539816339744830962f

/* sin and cos of x, |x| <= ~8192 (the path only passes phi in [0, 2*pi]). */
PT_HD void pt_sincosf(float x, float *s_out, float *c_out) {
    float kf = pt_rint_small(x * 0.636619772367581343f); /* 2/pi */
    float r = x - kf * 1.5703125f; /* pi/2 in three parts (Cody-Waite): 1.5703125 + 4.8375e-4 + 7.5498e-8 */
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    cp = cp - 0.5f * z;
    cp = cp + 1.0f;
    int k = (int)kf & 3;
    float s = (k & 1) ? cp : sp;
    float c = (k & 1) ? sp : cp;
    if (k == 1 || k == 2) c = -c;
    if (k >= 2) s = -s;
    *s_out = s;
    *c_out = c;
}

/* 2^n as a float for n in [-126, 127] */
PT_HD float pt_pow2i(int n) { return pt_u2f((uint32_t)(n + 127) << 23); }

PT_HD float pt_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return pt_u2f(0x7f800000u);
    if (x < -103.972076f) return 0.0f;
    float nf = pt_rint_small(x * 1.44269504088896341f);
    float r = x - nf * 0.693359375f;
    r = r - nf * -2.12194440e-4f;
    float z = r * r;
    float p = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r +
                 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f) * z;
    p = p + r;
    p = p + 1.0f;
    int n = (int)nf;
    int n1 = n / 2;
    int n2 = n - n1;
    return (p * pt_pow2i(n1)) * pt_pow2i(n2);
}

/* asin kernel, |a| <= 0.5 */
PT_HD float pt_asin_kernel(float a) {
    float z = a * a;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z +
             7.4953002686e-2f) * z + 1.6666752422e-1f) * z * a + a;
}

/* acos(x); |x| > 1 gives NaN (as libm), which the sky lookup turns into texel row 0. */
PT_HD float pt_acosf(float x) {
    if (x < -0.5f) return PT_PI_F - 2.0f * pt_asin_kernel(__builtin_sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * pt_asin_kernel(__builtin_sqrtf(0.5f * (1.0f - x)));
    return PT_PI_2_F - pt_asin_kernel(x);
}

PT_HD float pt_atanf(float t) {
    uint32_t sign = pt_f2u(t) & 0x80000000u;
    float a = pt_u2f(pt_f2u(t) & 0x7fffffffu);
    float y0;
    if (a > 2.414213562373095f) {
        y0 = PT_PI_2_F;
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) {
        y0 = PT_PI_4_F;
        a = (a - 1.0f) / (a + 1.0f);
    } else {
        y0 = 0.0f;
    }
    float z = a * a;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z -
               3.33329491539e-1f) * z * a + a;
    float r = y0 + p;
    return pt_u2f(pt_f2u(r) ^ sign);
}

/* atan2(y, x) with C99 zero/sign conventions for finite inputs. */
PT_HD float pt_atan2f(float y, float x) {
    if (x != x || y != y) return x + y;
    uint32_t ysign = pt_f2u(y) & 0x80000000u;
    if (x == 0.0f) {
        if (y == 0.0f) {
            float r = (pt_f2u(x) & 0x80000000u) ? PT_PI_F : 0.0f;
            return pt_u2f(pt_f2u(r) | ysign);
        }
        return pt_u2f(pt_f2u(PT_PI_2_F) | ysign);
    }
    float z = pt_atanf(y / x);
    if (x < 0.0f) z = ysign ? z - PT_PI_F : z + PT_PI_F;
    return z;
}

/* cbrt(x) (sphere.rs:108: radius * u.cbrt() of a light sample inside a sphere).  x = m * 2^(3q + r), m in [0.5, 1):
 * three Newton steps on m from a linear start (error 6e-3 -> 4e-5 -> 1e-9), then the factor cbrt(2^r) and 2^q. */
PT_HD float pt_cbrtf(float x) {
    uint32_t sign = pt_f2u(x) & 0x80000000u;
    float a = pt_u2f(pt_f2u(x) & 0x7fffffffu);
    if (!(a > 0.0f) || a == pt_u2f(0x7f800000u)) return x; /* zeros, NaN, infinities */
    float scale = 1.0f;
    if (a < 1.17549435e-38f) { /* subnormal: times 2^48, result times 2^-16 */
        a = a * 281474976710656.0f;
        scale = 1.52587890625e-05f;
    }
    uint32_t u = pt_f2u(a);
    int e = (int)(u >> 23) - 126;
    float m = pt_u2f((u & 0x007fffffu) | 0x3f000000u);
    int q = (e + 300) / 3 - 100;
    int r = e - 3 * q;
    float y = 0.5874010519681994f + 0.4125989480318006f * m;
    for (int k = 0; k < 3; ++k) {
        float y2 = y * y;
        y = y - (y2 * y - m) / (3.0f * y2);
    }
    y = y * (r == 0 ? 1.0f : (r == 1 ? 1.2599210498948732f : 1.5874010519681994f));
    y = (y * pt_pow2i(q)) * scale;
    return pt_u2f(pt_f2u(y) | sign);
}

#endif /* PTCORE_DETMATH_H */
