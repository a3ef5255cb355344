// Host <-> device transfer rates of the box for the drop-in do_all_sources (2 x N^3 doubles across PCIe per call):
// pageable hipMemcpy, hipHostRegister cost, registered (pinned) async copies, both directions at once.
// build + run: hipcc -O2 --offload-arch=gfx950 tools/micro/pcie.hip -o /tmp/pcie && /tmp/pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = (size_t)256 * 256 * 256 * 8;
    char *a = (char *)malloc(bytes), *b = (char *)malloc(bytes);
    memset(a, 1, bytes); memset(b, 2, bytes);
    void *d0, *d1;
    CK(hipMalloc(&d0, bytes)); CK(hipMalloc(&d1, bytes));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); CK(hipMemcpy(d0, a, bytes, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(b, d1, bytes, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable hipMemcpy 128 MiB: H2D %.3f ms (%.1f GB/s)  D2H %.3f ms (%.1f GB/s)\n", (t1 - t) * 1e3, bytes / (t1 - t) / 1e9,
               (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    {   // pageable async: when does the call return, when is the copy done
        double t = now(); CK(hipMemcpyAsync(d0, a, bytes, hipMemcpyHostToDevice, s0)); double t1 = now();
        CK(hipStreamSynchronize(s0)); double t2 = now();
        printf("pageable hipMemcpyAsync H2D: call returns after %.3f ms, complete after %.3f ms\n", (t1 - t) * 1e3, (t2 - t) * 1e3);
        t = now(); CK(hipMemcpyAsync(b, d1, bytes, hipMemcpyDeviceToHost, s1)); t1 = now();
        CK(hipStreamSynchronize(s1)); t2 = now();
        printf("pageable hipMemcpyAsync D2H: call returns after %.3f ms, complete after %.3f ms\n", (t1 - t) * 1e3, (t2 - t) * 1e3);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); CK(hipHostRegister(a, bytes, hipHostRegisterDefault)); double t1 = now();
        CK(hipHostRegister(b, bytes, hipHostRegisterDefault)); double t2 = now();
        CK(hipMemcpyAsync(d0, a, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double t3 = now();
        CK(hipMemcpyAsync(b, d1, bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double t4 = now();
        CK(hipMemcpyAsync(d0, a, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(b, d1, bytes, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double t5 = now();
        // chunked: 8 x 16 MiB each way, alternating, as a pipelined call would issue them
        for (int c = 0; c < 8; ++c) {
            CK(hipMemcpyAsync((char *)d0 + c * (bytes / 8), a + c * (bytes / 8), bytes / 8, hipMemcpyHostToDevice, s0));
            CK(hipMemcpyAsync(b + c * (bytes / 8), (char *)d1 + c * (bytes / 8), bytes / 8, hipMemcpyDeviceToHost, s1));
        }
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double t6 = now();
        CK(hipHostUnregister(a)); CK(hipHostUnregister(b)); double t7 = now();
        printf("hipHostRegister 128 MiB: %.3f / %.3f ms; registered H2D %.3f ms (%.1f GB/s), D2H %.3f ms (%.1f GB/s); both at once %.3f ms; "
               "8+8 chunks of 16 MiB %.3f ms; unregister both %.3f ms\n",
               (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, bytes / (t4 - t3) / 1e9,
               (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3);
    }
    {   // a pinned staging buffer of the library's own: CPU copy in + DMA
        void *pin; CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
        double t = now(); memcpy(pin, a, bytes); double t1 = now();
        CK(hipMemcpyAsync(d0, pin, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double t2 = now();
        printf("staging: CPU memcpy into pinned %.3f ms (%.1f GB/s), DMA from pinned %.3f ms (%.1f GB/s)\n", (t1 - t) * 1e3,
               bytes / (t1 - t) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    return 0;
}
