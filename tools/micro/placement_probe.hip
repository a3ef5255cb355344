// Does the streaming rate of the fused pass's mix (5 read + 7 write streams of 128 MiB) depend on WHERE the buffers lie?
// Within one process: allocate the twelve buffers, time the kernel, keep them (so the next set lies elsewhere), repeat; then
// free everything and allocate again.  If the sets differ among each other within seconds, the "state of the box" that moves the
// fused pass by 15-20 % between runs is the placement of its grids; if they agree, it is something that changes with time.
//   hipcc -O3 --offload-arch=gfx950 -o placement_probe placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) mix(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c,
                                           const double *__restrict__ d, const double *__restrict__ e, double *o0, double *o1, double *o2,
                                           double *o3, double *o4, double *o5, double *o6, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double s = __builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i) * __builtin_nontemporal_load(c + i) +
                         __builtin_nontemporal_load(d + i) + __builtin_nontemporal_load(e + i);
        o0[i] = s; o1[i] = s + 1.0; o2[i] = s * 2.0; o3[i] = s - 1.0; o4[i] = 0.0; o5[i] = 0.0; o6[i] = s * s;
    }
}
__global__ void __launch_bounds__(256) read_one(const double *__restrict__ a, double *out, size_t n)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += __builtin_nontemporal_load(a + i);
    if (s == 12345.678) out[0] = s;
}
// one load per PAGE_BYTES of the twelve buffers (argv[2] = "t"): what address translation costs on this placement
__global__ void __launch_bounds__(256) touch_pages(const double *const *bufs, size_t n, size_t stride, double *out)
{
    double s = 0.0;
    const size_t pages = n / stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 12 * pages; i += (size_t)gridDim.x * 256)
        s += __builtin_nontemporal_load(bufs[i % 12] + (i / 12) * stride);
    if (s == 12345.678) out[0] = s;
}
__global__ void __launch_bounds__(256) write_one(double *a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = 0.0;
}
int main(int argc, char **argv)
{
    const size_t n = 256ull * 256 * 256;
    const int sets = argc > 1 ? atoi(argv[1]) : 8;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    std::vector<double *> all;
    auto time_set = [&](double **buf, float &best, float &worst) -> int {
        best = 1e9f; worst = 0.0f;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(t0));
            hipLaunchKernelGGL(mix, dim3(1024), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9], buf[10], buf[11], n);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep >= 2) { if (ms < best) best = ms; if (ms > worst) worst = ms; }
        }
        return 0;
    };
    for (int round = 0; round < 2; ++round) {
        std::vector<double *> first;
        for (int s = 0; s < sets; ++s) {
            double *buf[12];
            // argv[3] = "arena[:SKEW_BYTES]": the twelve buffers carved out of ONE allocation, buffer q starting SKEW * q bytes later
            const bool arena = argc > 3 && argv[3][0] == 'a';
            const size_t skew = arena && argv[3][5] == ':' ? (size_t)atol(argv[3] + 6) : 0;
            if (arena) {
                char *base = nullptr;
                const int slots = argc > 4 ? atoi(argv[4]) : 12;          // argv[4]: size of the arena in buffers (the first twelve are used)
                CK(hipMalloc(&base, (size_t)slots * (n * 8 + skew) + 256)); CK(hipMemset(base, 0, 12 * (n * 8 + skew)));
                all.push_back(reinterpret_cast<double *>(base));
                for (int q = 0; q < 12; ++q) buf[q] = reinterpret_cast<double *>(base + (size_t)q * (n * 8 + skew));
            } else
                for (auto &p : buf) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); all.push_back(p); }
            if (s == 0) first.assign(buf, buf + 12);
            float best, worst;
            if (time_set(buf, best, worst)) return 1;
            printf("round %d set %d (first buffer at %p): %.3f ... %.3f ms = %.2f TB/s\n", round, s, (void *)buf[0], best, worst, 12.0 * n * 8 / best * 1e-9);
            if (argc > 2 && argv[2][0] == 't') {
                const double **dl = nullptr; CK(hipMalloc(&dl, 12 * sizeof(double *))); CK(hipMemcpy(dl, buf, 12 * sizeof(double *), hipMemcpyHostToDevice));
                printf("   one load per 4 KiB / 64 KiB / 2 MiB of every buffer, us:");
                for (size_t stride : {(size_t)512, (size_t)8192, (size_t)262144}) {
                    float b = 1e9f;
                    for (int rep = 0; rep < 6; ++rep) {
                        float ms;
                        CK(hipEventRecord(t0)); hipLaunchKernelGGL(touch_pages, dim3(256), dim3(256), 0, 0, (const double *const *)dl, n, stride, buf[0]); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
                        CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < b) b = ms;
                    }
                    printf(" %.1f", b * 1e3);
                }
                printf("\n");
                CK(hipFree(dl));
            }
            if (argc > 2 && argv[2][0] == 'p') {       // every buffer of the set by itself: read it, write it (microseconds, best of 6)
                printf("   per buffer read/write us:");
                for (int q = 0; q < 12; ++q) {
                    float br = 1e9f, bw = 1e9f;
                    for (int rep = 0; rep < 7; ++rep) {
                        float ms;
                        CK(hipEventRecord(t0)); hipLaunchKernelGGL(read_one, dim3(2048), dim3(256), 0, 0, buf[q], buf[(q + 1) % 12], n); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
                        CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < br) br = ms;
                        CK(hipEventRecord(t0)); hipLaunchKernelGGL(write_one, dim3(2048), dim3(256), 0, 0, buf[q], n); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
                        CK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < bw) bw = ms;
                    }
                    printf(" %.0f/%.0f", br * 1e3, bw * 1e3);
                }
                printf("\n");
            }
        }
        {   // the first set of the round once more, now that the others exist: the same placement a second later
            float best, worst;
            if (time_set(first.data(), best, worst)) return 1;
            printf("round %d set 0 again: %.3f ... %.3f ms = %.2f TB/s\n", round, best, worst, 12.0 * n * 8 / best * 1e-9);
        }
        for (double *p : all) CK(hipFree(p));
        all.clear();
    }
    return 0;
}
