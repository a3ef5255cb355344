// Can a stream wait for a counter that a RUNNING kernel of another stream increments (hipStreamWaitValue64 on signal memory)?
// Stream A: a long kernel whose workgroups add 1 to a counter as they finish.  Stream B: wait until the counter reaches
// half / all of the workgroups, then a small kernel that records the time.  Prints when B's kernels ran relative to A's end.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void worker(unsigned long long *counter, double *sink, int spin)
{
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    if (x == 12345.678) sink[0] = x;
    __threadfence();
    if (threadIdx.x == 0) atomicAdd(counter, 1ULL);
}
__global__ void stamp(long long *out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    unsigned long long *sig = nullptr;
    CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    double *sink; long long *t; CK(hipMalloc(&sink, 8)); CK(hipMalloc(&t, 4 * sizeof(long long)));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int blocks = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamWriteValue64(a, sig, 0, 0));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, t + 0);
        hipLaunchKernelGGL(worker, dim3(blocks), dim3(256), 0, a, sig, sink, 20000);
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, t + 1);
        CK(hipStreamWaitValue64(b, sig, blocks / 2, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFULL));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, t + 2);
        CK(hipStreamWaitValue64(b, sig, blocks, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFULL));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, t + 3);
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        long long h[4]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
        const double tick_us = 0.01;      // wall_clock64 counts at 100 MHz
        printf("kernel A ran %.1f us; B passed 'half' %.1f us and 'all' %.1f us after A's start\n", (h[1] - h[0]) * tick_us,
               (h[2] - h[0]) * tick_us, (h[3] - h[0]) * tick_us);
    }
    printf("OK\n");
    return 0;
}
