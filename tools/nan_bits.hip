// which NaN an invalid operation produces on gfx950 (x86 SSE gives the negative "real indefinite" 0xffc00000 for all of them): hipcc --offload-arch=gfx950 tools/nan_bits.hip -o tools/nan_bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#pragma clang fp contract(off)
__global__ void k(const float* in, unsigned* out) {
    const float zero = in[0], inf = in[1], one = in[2];
    float r[8];
    r[0] = zero / zero;
    r[1] = inf / inf;
    r[2] = inf - inf;
    r[3] = zero * inf;
    r[4] = sqrtf(-one);
    r[5] = -(inf - inf);
    r[6] = (inf - inf) + one;
    r[7] = __fmaf_rn(zero, inf, one);
    for (int i = 0; i < 8; ++i) out[i] = __float_as_uint(r[i]);
}
int main() {
    float h[3] = {0.0f, INFINITY, 1.0f}, *d;
    unsigned *o, ho[8];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const char* names[8] = {"0/0", "inf/inf", "inf-inf", "0*inf", "sqrt(-1)", "-(inf-inf)", "(inf-inf)+1", "fma(0,inf,1)"};
    volatile float z = 0.0f, i = INFINITY, one = 1.0f;
    float c[8] = {z / z, i / i, i - i, z * i, sqrtf(-one), -(i - i), (i - i) + one, fmaf(z, i, one)};
    for (int n = 0; n < 8; ++n) { unsigned u; memcpy(&u, &c[n], 4); printf("%-14s gpu %08x   host %08x\n", names[n], ho[n], u); }
    return 0;
}
