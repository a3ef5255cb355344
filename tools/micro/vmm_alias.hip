// Can one physical allocation be mapped into several virtual ranges, each followed by pages of its own (HIP virtual memory
// management)?  What a table shared between the sign variants of a clipped window would need: [shared prefix | own tail] contiguous
// in every variant's address range, with no change to the kernel that walks it.   hipcc --offload-arch=gfx950 -o vmm_alias vmm_alias.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned *p, size_t n, unsigned v) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i; }
__global__ void sum(const unsigned *p, size_t n, unsigned long long *out) { unsigned long long s = 0; for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; atomicAdd(out, s); }
int main()
{
    int dev = 0; CK(hipSetDevice(dev));
    int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    const size_t prefix = 8 * gran, tail = gran;
    hipMemGenericAllocationHandle_t hp, ht[3];
    CK(hipMemCreate(&hp, prefix, &prop, 0));
    for (int v = 0; v < 3; ++v) CK(hipMemCreate(&ht[v], tail, &prop, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va[3];
    for (int v = 0; v < 3; ++v) {
        CK(hipMemAddressReserve(&va[v], prefix + tail, 0, nullptr, 0));
        CK(hipMemMap(va[v], prefix, 0, hp, 0));
        CK(hipMemMap((char *)va[v] + prefix, tail, 0, ht[v], 0));
        CK(hipMemSetAccess(va[v], prefix + tail, &acc, 1));
    }
    // write the prefix through range 0, each tail through its own range; read everything through every range
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, (unsigned *)va[0], prefix / 4, 1000u);
    for (int v = 0; v < 3; ++v) hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, (unsigned *)((char *)va[v] + prefix), tail / 4, 7u * (v + 1));
    CK(hipDeviceSynchronize());
    unsigned long long *d = nullptr, h[3];
    CK(hipMalloc(&d, 3 * sizeof *d)); CK(hipMemset(d, 0, 3 * sizeof *d));
    for (int v = 0; v < 3; ++v) hipLaunchKernelGGL(sum, dim3(256), dim3(256), 0, 0, (const unsigned *)va[v], (prefix + tail) / 4, d + v);
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    const unsigned long long np = prefix / 4, nt = tail / 4;
    for (int v = 0; v < 3; ++v) {
        const unsigned long long want = 1000ull * np + np * (np - 1) / 2 + 7ull * (v + 1) * nt + nt * (nt - 1) / 2;
        printf("range %d: sum %llu, expected %llu %s\n", v, h[v], want, h[v] == want ? "ok" : "MISMATCH");
    }
    // host copy out of an aliased range
    std::vector<unsigned> host((prefix + tail) / 4);
    CK(hipMemcpy(host.data(), va[2], prefix + tail, hipMemcpyDeviceToHost));
    printf("memcpy from an aliased range: first %u last of prefix %u first of tail %u\n", host[0], host[np - 1], host[np]);
    for (int v = 0; v < 3; ++v) { CK(hipMemUnmap(va[v], prefix + tail)); CK(hipMemAddressFree(va[v], prefix + tail)); CK(hipMemRelease(ht[v])); }
    CK(hipMemRelease(hp));
    printf("done\n");
    return 0;
}
