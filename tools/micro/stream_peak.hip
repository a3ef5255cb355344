// Practical HBM streaming rate of the box for the mix of the fused chemistry pass: 5 read streams and 7 write streams of N^3
// doubles each (plus a plain copy for reference), grid-stride kernels at the occupancy the pass uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) mix(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c,
                                           const double *__restrict__ d, const double *__restrict__ e, double *o0, double *o1, double *o2,
                                           double *o3, double *o4, double *o5, double *o6, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double s = a[i] + b[i] * c[i] + d[i] + e[i];
        o0[i] = s; o1[i] = s + 1.0; o2[i] = s * 2.0; o3[i] = s - 1.0; o4[i] = 0.0; o5[i] = 0.0; o6[i] = s * s;
    }
}
// the same with non-temporal stores (and loads): does taking the streams past the caches change what the memory side delivers?
template <bool NT_LOADS, bool NT_STORES>
__global__ void __launch_bounds__(256) mix_nt(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c,
                                              const double *__restrict__ d, const double *__restrict__ e, double *o0, double *o1, double *o2,
                                              double *o3, double *o4, double *o5, double *o6, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double va, vb, vc, vd, ve;
        if (NT_LOADS) { va = __builtin_nontemporal_load(a + i); vb = __builtin_nontemporal_load(b + i); vc = __builtin_nontemporal_load(c + i);
                        vd = __builtin_nontemporal_load(d + i); ve = __builtin_nontemporal_load(e + i); }
        else { va = a[i]; vb = b[i]; vc = c[i]; vd = d[i]; ve = e[i]; }
        const double s = va + vb * vc + vd + ve;
        if (NT_STORES) {
            __builtin_nontemporal_store(s, o0 + i); __builtin_nontemporal_store(s + 1.0, o1 + i); __builtin_nontemporal_store(s * 2.0, o2 + i);
            __builtin_nontemporal_store(s - 1.0, o3 + i); __builtin_nontemporal_store(0.0, o4 + i); __builtin_nontemporal_store(0.0, o5 + i);
            __builtin_nontemporal_store(s * s, o6 + i);
        } else { o0[i] = s; o1[i] = s + 1.0; o2[i] = s * 2.0; o3[i] = s - 1.0; o4[i] = 0.0; o5[i] = 0.0; o6[i] = s * s; }
    }
}
__global__ void __launch_bounds__(256) copy(const double2 *__restrict__ a, double2 *o, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
int main()
{
    const size_t n = 256ull * 256 * 256;
    double *buf[12];
    for (auto &p : buf) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        float best_mix = 1e9f, best_copy = 1e9f, best_nt = 1e9f, best_ntl = 1e9f, best_l = 1e9f;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL(mix, dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9], buf[10], buf[11], n);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < best_mix) best_mix = ms;
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL(copy, dim3(grid), dim3(256), 0, 0, (const double2 *)buf[0], (double2 *)buf[5], n / 2);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < best_copy) best_copy = ms;
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL((mix_nt<false, true>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9], buf[10], buf[11], n);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < best_nt) best_nt = ms;
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL((mix_nt<true, true>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9], buf[10], buf[11], n);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < best_ntl) best_ntl = ms;
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL((mix_nt<true, false>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9], buf[10], buf[11], n);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < best_l) best_l = ms;
        }
        printf("grid %5d: non-temporal LOADS only: %.3f ms = %.2f TB/s\n", grid, best_l, 12.0 * n * 8 / best_l * 1e-9);
        printf("grid %5d: the same with non-temporal stores: %.3f ms = %.2f TB/s;  non-temporal stores and loads: %.3f ms = %.2f TB/s\n", grid, best_nt,
               12.0 * n * 8 / best_nt * 1e-9, best_ntl, 12.0 * n * 8 / best_ntl * 1e-9);
        printf("grid %5d: 5 reads + 7 writes of 128 MiB: %.3f ms = %.2f TB/s;  copy 128 MiB: %.3f ms = %.2f TB/s\n", grid, best_mix,
               12.0 * n * 8 / best_mix * 1e-9, best_copy, 2.0 * n * 8 / best_copy * 1e-9);
    }
    return 0;
}
