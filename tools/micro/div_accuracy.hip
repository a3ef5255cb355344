// Accuracy of the reciprocal-based divisions of rates_device.hpp against the IEEE quotient, on the GPU.
//   hipcc -O3 --offload-arch=gfx950 -I pyc2ray_amd/csrc -I include -o /tmp/div_accuracy tools/micro/div_accuracy.hip && /tmp/div_accuracy
// Prints, over 2^26 random operand pairs of ordinary magnitude: the largest relative error of v_rcp_f64 itself, and for the
// quotient with ONE and with TWO Newton steps on the reciprocal (each followed by the correction of the quotient) the share
// of results that differ from the correctly rounded quotient and the largest difference in ulps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ inline double div1(double x, double y)
{
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    const double q = x * r;
    return fma(fma(-y, q, x), r, q);
}
__device__ inline double div2(double x, double y)
{
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    r = fma(fma(-y, r, 1.0), r, r);
    const double q = x * r;
    return fma(fma(-y, q, x), r, q);
}
__device__ inline uint64_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ inline double operand(uint64_t &s)
{   // mantissa uniform, exponent in [-200, 200]
    const uint64_t m = rng(s) >> 12;
    const int e = (int)(rng(s) % 401) - 200;
    return ldexp(1.0 + (double)m * 0x1p-52, e);
}
__global__ void check(unsigned long long *out, double *worst_rcp)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bad1 = 0, bad2 = 0, ulp1 = 0, ulp2 = 0;
    double wr = 0.0;
    for (int it = 0; it < 1024; ++it) {
        const double x = operand(s), y = operand(s);
        const double q = x / y;                         // IEEE (v_div_scale / v_div_fmas / v_div_fixup)
        const double a = div1(x, y), b = div2(x, y);
        const long long qa = llabs(__double_as_longlong(a) - __double_as_longlong(q));
        const long long qb = llabs(__double_as_longlong(b) - __double_as_longlong(q));
        bad1 += qa != 0; bad2 += qb != 0;
        ulp1 = max(ulp1, (unsigned long long)qa); ulp2 = max(ulp2, (unsigned long long)qb);
        wr = fmax(wr, fabs(fma(-y, __builtin_amdgcn_rcp(y), 1.0)));
    }
    atomicAdd(out + 0, bad1); atomicAdd(out + 1, bad2);
    atomicMax(out + 2, ulp1); atomicMax(out + 3, ulp2);
    // (doubles >= 0 order like their bit patterns)
    atomicMax((unsigned long long *)worst_rcp, (unsigned long long)__double_as_longlong(wr));
}
int main()
{
    unsigned long long *out; double *wr;
    hipMalloc(&out, 4 * sizeof(*out)); hipMalloc(&wr, sizeof(double));
    hipMemset(out, 0, 4 * sizeof(*out)); hipMemset(wr, 0, sizeof(double));
    hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, 0, out, wr);
    unsigned long long h[4]; double hw;
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hw, wr, sizeof(hw), hipMemcpyDeviceToHost);
    const double n = 256.0 * 256.0 * 1024.0;
    printf("v_rcp_f64: largest |1 - y*rcp(y)| = %.3e (2^%.1f)\n", hw, log2(hw));
    printf("one Newton step  + quotient correction: %.3e of the quotients differ from IEEE, at most %llu ulp\n", h[0] / n, h[2]);
    printf("two Newton steps + quotient correction: %.3e of the quotients differ from IEEE, at most %llu ulp\n", h[1] / n, h[3]);
    return 0;
}
