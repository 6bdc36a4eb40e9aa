// which NaN an invalid operation produces on gfx950 and what a subtraction does to a NaN operand (measured: the same 0xffc00000 as x86 SSE from
// every invalid operation; 1 - NaN comes out with the sign flipped on the GPU, unchanged on x86): hipcc --offload-arch=gfx950 tools/nan_bits.hip -o tools/nan_bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#pragma clang fp contract(off)
__global__ void k(const float* in, unsigned* out) {
    const float zero = in[0], inf = in[1], one = in[2];
    float r[12];
    r[0] = zero / zero;
    r[1] = inf / inf;
    r[2] = inf - inf;
    r[3] = zero * inf;
    r[4] = sqrtf(-one);
    r[5] = -(inf - inf);
    r[6] = (inf - inf) + one;
    r[7] = __fmaf_rn(zero, inf, one);
    const float nan = inf - inf;
    r[8] = one - nan;
    r[9] = nan - one;
    r[10] = one * nan;
    r[11] = fminf(nan, one) + (one - nan) * zero;
    for (int i = 0; i < 12; ++i) out[i] = __float_as_uint(r[i]);
}
int main() {
    float h[3] = {0.0f, INFINITY, 1.0f}, *d;
    unsigned *o, ho[12];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&o, sizeof(ho)) != hipSuccess) return 1;
    if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    if (hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[12] = {"0/0", "inf/inf", "inf-inf", "0*inf", "sqrt(-1)", "-(inf-inf)", "(inf-inf)+1", "fma(0,inf,1)", "1-nan", "nan-1", "1*nan", "(1-nan)*0+1"};
    volatile float z = 0.0f, i = INFINITY, one = 1.0f;
    volatile float nn = i - i;
    float c[12] = {z / z, i / i, i - i, z * i, sqrtf(-one), -(i - i), (i - i) + one, fmaf(z, i, one), one - nn, nn - one, one * nn, fminf(nn, one) + (one - nn) * z};
    for (int n = 0; n < 12; ++n) { unsigned u; memcpy(&u, &c[n], 4); printf("%-14s gpu %08x   host %08x\n", names[n], ho[n], u); }
    return 0;
}
