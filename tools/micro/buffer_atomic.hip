// Semantics of buffer_atomic_add_f64 through a raw buffer descriptor on gfx950, as the raytrace kernel uses it:
// lanes with an offset inside [0, num_records) add; lanes with the offset 0x80000000 (beyond any descriptor of <= 2 GiB)
// are dropped by the range check.  (An offset of -1 is NOT: offset + 8 wraps around in the 32-bit range check, the access goes
// to base + 4 GiB and the process dies of a memory fault -- seen once, with the first build of the kernel.)  Prints "OK".   hipcc --offload-arch=gfx950 -O2 buffer_atomic.hip -o buffer_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern "C" __device__ double buffer_atomic_fadd_f64(double, __amdgpu_buffer_rsrc_t, int, int, int)
    __asm("llvm.amdgcn.raw.ptr.buffer.atomic.fadd.f64");
__global__ void k(double *p, unsigned nbytes, int n, int mode)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)nbytes, 0x00020000);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = (t % 3) != 0;                                  // every third lane has nothing to add
    if (mode == 0) { if (ok) (void)buffer_atomic_fadd_f64(1.0 + t, r, (int)((t % n) * 8u), 0, 0); return; }
    (void)buffer_atomic_fadd_f64(1.0 + t, r, ok ? (int)((t % n) * 8u) : (int)0x80000000u, 0, 0);
}
int main()
{
    const int n = 1000, threads = 256, blocks = 64;
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int mode = 0; mode < 2; ++mode) {
        printf("mode %d (0: in-range lanes under a branch, 1: all lanes issue, those without a value at offset 2 GiB)\n", mode);
        double *d;
        if (hipMalloc(&d, (n + 64) * sizeof(double)) != hipSuccess) return 2;
        (void)hipMemset(d, 0, (n + 64) * sizeof(double));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, (unsigned)(n * sizeof(double)), n, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
        std::vector<double> h(n + 64);
        (void)hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
        std::vector<double> want(n + 64, 0.0);
        for (int t = 0; t < blocks * threads; ++t) if (t % 3) want[t % n] += 1.0 + t;
        int bad = 0;
        for (int i = 0; i < n + 64; ++i) if (h[i] != want[i]) { if (bad < 5) printf("cell %d: %g, expected %g\n", i, h[i], want[i]); ++bad; }
        printf("%d cells differ; beyond the range: %g %g\n", bad, h[n], h[n + 1]);
        if (bad) { printf("FAILED\n"); return 1; }
        (void)hipFree(d);
    }
    printf("OK\n");
    return 0;
}
