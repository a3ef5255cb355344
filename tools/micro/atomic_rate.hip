// Microbenchmark: rate of no-return global_atomic_add_f64 for the access shape of the raytrace kernel
// (each wave adds to rows of ROW consecutive doubles that start at arbitrary 8-byte alignment, rows scattered over
// a 256^3 x 2 grid).  Prints atomics/s and 64-B requests/s (lines touched per wave-instruction, counted on the host).
// build+run: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>

__global__ void __launch_bounds__(256) adds(double *grid, const unsigned *idx, int per_lane, double v)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    for (int k = 0; k < per_lane; ++k) unsafeAtomicAdd(grid + idx[(size_t)k * nthreads + t], v);
}

int main(int argc, char **argv)
{
    const int row = argc > 1 ? atoi(argv[1]) : 40;          // doubles per row
    const size_t edge = argc > 2 ? (size_t)atoi(argv[2]) : 256;   // footprint of the target: 2 x edge^3 doubles (256: 256 MiB, the Infinity Cache's size)
    const size_t ncell = (size_t)2 * edge * edge * edge;
    const int blocks = 256 * 16, threads = 256, per_lane = 64;
    const size_t nthreads = (size_t)blocks * threads, n = nthreads * per_lane;
    std::vector<unsigned> h(n);
    srand(1);
    size_t requests = 0;
    // consecutive lanes walk along rows; a new row starts at a random cell
    for (int k = 0; k < per_lane; ++k) {
        unsigned pos = 0; int left = 0;
        for (size_t t = 0; t < nthreads; ++t) {
            if (left == 0) { pos = (unsigned)((((size_t)rand() << 16) ^ rand()) % (ncell - 4096)); left = row; }
            h[(size_t)k * nthreads + t] = pos++; --left;
        }
    }
    for (int k = 0; k < per_lane; ++k)
        for (size_t w = 0; w < nthreads; w += 64) {
            std::set<unsigned> lines;
            for (int l = 0; l < 64; ++l) lines.insert(h[(size_t)k * nthreads + w + l] >> 3);
            requests += lines.size();
        }
    double *g; unsigned *d;
    hipMalloc(&d, n * sizeof(unsigned)); hipMemcpy(d, h.data(), n * sizeof(unsigned), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // argv[3] = PLACEMENTS: the same stream of atomics on that many allocations of the target, all kept (each lies elsewhere):
    // does the rate at which the memory side takes the requests depend on WHERE the target lies, as streaming does (placement_probe.hip)?
    const int placements = argc > 3 ? atoi(argv[3]) : 1;
    for (int pl = 0; pl + 1 < placements; ++pl) {
        hipMalloc(&g, ncell * sizeof(double)); hipMemset(g, 0, ncell * sizeof(double));
        adds<<<blocks, threads>>>(g, d, per_lane, 1.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) adds<<<blocks, threads>>>(g, d, per_lane, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("placement %2d: %.3f ms per launch, %.3e requests/s\n", pl, ms, requests / (ms * 1e-3));
    }
    hipMalloc(&g, ncell * sizeof(double)); hipMemset(g, 0, ncell * sizeof(double));
    adds<<<blocks, threads>>>(g, d, per_lane, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) adds<<<blocks, threads>>>(g, d, per_lane, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("target 2 x %zu^3 doubles, row=%d doubles: %.3f ms per launch, %.3e atomics/s, %.3e requests/s (%.2f doubles per request)\n", edge, row, ms,
           n / (ms * 1e-3), requests / (ms * 1e-3), (double)n / requests);
    return 0;
}
