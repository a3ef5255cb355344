// grid_kernels.hip -- the N^3 helper passes around the raytrace: nHI = ndens (1 - xh_av) in both layouts, the [i][j][k] <-> [k][j][i]
// transposes, the folds of the z-faces' transposed rate accumulator.  All streaming, tiled 32 x 32 through LDS.
#include "asora_internal.hpp"

#include <algorithm>
#include <cmath>

namespace asora {

// ---------------------------------------------------------------------------------------------
// N^3 helper kernels
// ---------------------------------------------------------------------------------------------

// nhi[i][j][k] = ndens*(1-xh_av);  nhi_t[k][j][i] = same (tiled transpose of the (i,k) planes), for the planes
// [i_begin, i_end).  block (32,8): tile 32(i) x 32(k) of one j.  ZERO: also zero both layouts of an accumulator
// on those planes (a multi-GPU rank only touches the planes its sources reach).
template <bool WITH_T, bool ZERO>
__global__ void __launch_bounds__(256) prepare_nhi_kernel(const double *__restrict__ nd, const double *__restrict__ xh,
                                                          double *__restrict__ nhi, double *__restrict__ nhi_t, int N,
                                                          int i_begin, int i_end, double *__restrict__ acc, size_t ncell,
                                                          const int *__restrict__ done = nullptr)
{
    if (done && *done) return;                                  // device loop: the time step has converged
    __shared__ double tile[32][33];
    const int j = blockIdx.y;
    const int ib = i_begin + blockIdx.z * 32, kb = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = ib + r, k = kb + threadIdx.x;
        if (i < i_end && k < N) {
            const size_t idx = ((size_t)i * N + j) * N + k;
            const double v = nd[idx] * (1.0 - xh[idx]);       // raytracing.cu:276
            nhi[idx] = v;
            if (ZERO) acc[idx] = 0.0;
            if (WITH_T) tile[r][threadIdx.x] = v;
        }
    }
    if (!WITH_T) return;
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = kb + r, i = ib + threadIdx.x;
        if (i < i_end && k < N) {
            const size_t o = ((size_t)k * N + j) * N + i;
            nhi_t[o] = tile[threadIdx.x][r];
            if (ZERO) acc[ncell + o] = 0.0;
        }
    }
}

// dst[k][j][i] (op)= src[i][j][k]
template <bool ACCUMULATE>
__global__ void __launch_bounds__(256) transpose_ik_kernel(const double *__restrict__ src, double *__restrict__ dst, int N)
{
    __shared__ double tile[32][33];
    const int j = blockIdx.y;
    const int ib = blockIdx.z * 32, kb = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = ib + r, k = kb + threadIdx.x;
        if (i < N && k < N) tile[r][threadIdx.x] = src[((size_t)i * N + j) * N + k];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = kb + r, i = ib + threadIdx.x;
        if (i < N && k < N) {
            const size_t o = ((size_t)k * N + j) * N + i;
            if (ACCUMULATE) dst[o] += tile[threadIdx.x][r];
            else dst[o] = tile[threadIdx.x][r];
        }
    }
}

static dim3 tile_grid(int N) { const unsigned t = (N + 31) / 32; return dim3(t, N, t); }

int launch_prepare_nhi_from(State &st, const double *xh_av, bool need_transposed)
{
    KernelTimer kt(ASORA_KERNEL_PREP);
    const int N = st.N;
    if (need_transposed)
        hipLaunchKernelGGL((prepare_nhi_kernel<true, false>), tile_grid(N), dim3(32, 8), 0, st.stream,
                           st.grid[ASORA_GRID_NDENS], xh_av, st.nhi, st.nhi_t, N, 0, N, (double *)nullptr, st.ncell, (const int *)nullptr);
    else
        hipLaunchKernelGGL((prepare_nhi_kernel<false, false>), tile_grid(N), dim3(32, 8), 0, st.stream,
                           st.grid[ASORA_GRID_NDENS], xh_av, st.nhi, st.nhi_t, N, 0, N, (double *)nullptr, st.ncell, (const int *)nullptr);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_prepare_nhi(State &st, bool need_transposed)
{
    return launch_prepare_nhi_from(st, st.grid[ASORA_GRID_XH_AV], need_transposed);
}

// nHI in both layouts on the planes [i_begin, i_begin + i_count) only; zero_acc: also zero both layouts of `acc` there
int launch_prepare_range(State &st, int i_begin, int i_count, bool zero_acc, double *acc, const int *done)
{
    if (i_count <= 0) return 0;
    KernelTimer kt(ASORA_KERNEL_PREP);
    const int N = st.N;
    const unsigned t = (N + 31) / 32;
    const dim3 grid(t, N, (i_count + 31) / 32);
    if (zero_acc)
        hipLaunchKernelGGL((prepare_nhi_kernel<true, true>), grid, dim3(32, 8), 0, st.stream, st.grid[ASORA_GRID_NDENS],
                           st.grid[ASORA_GRID_XH_AV], st.nhi, st.nhi_t, N, i_begin, i_begin + i_count, acc, st.ncell, done);
    else
        hipLaunchKernelGGL((prepare_nhi_kernel<true, false>), grid, dim3(32, 8), 0, st.stream, st.grid[ASORA_GRID_NDENS],
                           st.grid[ASORA_GRID_XH_AV], st.nhi, st.nhi_t, N, i_begin, i_begin + i_count, acc, st.ncell, done);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_finish_phi(State &st)
{
    KernelTimer kt(ASORA_KERNEL_FINISH);
    // phi[i][j][k] += phi_t[k][j][i]  (the transpose is an involution on the index pair)
    hipLaunchKernelGGL(transpose_ik_kernel<true>, tile_grid(st.N), dim3(32, 8), 0, st.stream,
                       (const double *)st.phi_t, st.grid[ASORA_GRID_PHI_ION], st.N);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fold_transposed(State &st, const double *src_t, double *dst)
{
    KernelTimer kt(ASORA_KERNEL_FINISH);
    hipLaunchKernelGGL(transpose_ik_kernel<true>, tile_grid(st.N), dim3(32, 8), 0, st.stream, src_t, dst, st.N);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// dst[i][j][k] += src_t[k][j][i] for i in [i_begin, i_begin + i_count): the slab's share of the fold
__global__ void __launch_bounds__(256) fold_range_kernel(const double *__restrict__ src_t, double *__restrict__ dst, int N,
                                                         int i_begin, int i_end)
{
    __shared__ double tile[32][33];
    const int j = blockIdx.y;
    const int kb = blockIdx.z * 32, ib = i_begin + blockIdx.x * 32;       // src_t tile: rows k, columns i
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = kb + r, i = ib + threadIdx.x;
        if (k < N && i < i_end) tile[r][threadIdx.x] = src_t[((size_t)k * N + j) * N + i];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = ib + r, k = kb + threadIdx.x;
        if (i < i_end && k < N) dst[((size_t)i * N + j) * N + k] += tile[threadIdx.x][r];
    }
}

int launch_fold_range(State &st, const double *src_t, double *dst, int i_begin, int i_count)
{
    if (i_count <= 0) return 0;
    KernelTimer kt(ASORA_KERNEL_FINISH);
    const unsigned tk = (st.N + 31) / 32, ti = (i_count + 31) / 32;
    hipLaunchKernelGGL(fold_range_kernel, dim3(ti, st.N, tk), dim3(32, 8), 0, st.stream, src_t, dst, st.N, i_begin,
                       i_begin + i_count);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// Multi-GPU device loop (asora_evolve_slab_*): the rates this rank traced onto planes [i_begin, i_end) that ANOTHER rank owns,
// out[i][j][k] = a[i][j][k] + a_t[k][j][i] -- the message to the owner, a[] and a_t[] left intact -- and, in the same sweep, the
// OTHER accumulator pair zeroed on those planes for the next iteration's trace (what the fused pass does on the own planes).
__global__ void __launch_bounds__(256) fold_out_kernel(const double *__restrict__ a, const double *__restrict__ a_t, double *__restrict__ out,
                                                       double *__restrict__ z_a, double *__restrict__ z_t, int N, int i_begin, int i_end,
                                                       const int *__restrict__ done)
{
    if (done && *done) return;
    __shared__ double tile[32][33];
    const int j = blockIdx.y;
    const int kb = blockIdx.z * 32, ib = i_begin + blockIdx.x * 32;       // a_t tile: rows k, columns i
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = kb + r, i = ib + threadIdx.x;
        if (k < N && i < i_end) { const size_t o = ((size_t)k * N + j) * N + i; tile[r][threadIdx.x] = a_t[o]; if (z_t) z_t[o] = 0.0; }
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = ib + r, k = kb + threadIdx.x;
        if (i < i_end && k < N) { const size_t o = ((size_t)i * N + j) * N + k; out[o] = a[o] + tile[threadIdx.x][r]; if (z_a) z_a[o] = 0.0; }
    }
}

int launch_fold_out(State &st, const double *a, const double *a_t, double *out, double *z_a, double *z_t, int i_begin, int i_count,
                    const int *done)
{
    if (i_count <= 0) return 0;
    KernelTimer kt(ASORA_KERNEL_FINISH);
    const unsigned tk = (st.N + 31) / 32, ti = (i_count + 31) / 32;
    hipLaunchKernelGGL(fold_out_kernel, dim3(ti, st.N, tk), dim3(32, 8), 0, st.stream, a, a_t, out, z_a, z_t, st.N, i_begin,
                       i_begin + i_count, done);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// dst[q] += src[q], q < n (n even: whole planes of an even N^2, or handled by the tail): the rates another rank sent for planes
// this rank owns, added to its own accumulator in the order the host issues the calls
__global__ void __launch_bounds__(256) add_planes_kernel(double *__restrict__ dst, const double *__restrict__ src, size_t n,
                                                         const int *__restrict__ done)
{
    if (done && *done) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) dst[q] += src[q];
}

int launch_add_planes(State &st, double *dst, const double *src, size_t n, const int *done)
{
    if (n == 0) return 0;
    KernelTimer kt(ASORA_KERNEL_FINISH);
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)st.cu_count * 32);
    hipLaunchKernelGGL(add_planes_kernel, dim3(blocks), dim3(256), 0, st.stream, dst, src, n, done);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// dst[i][j][k] = a[i][j][k] + b_t[k][j][i] (the two accumulators of the device loop, left intact, into the rate grid)
__global__ void __launch_bounds__(256) fold_sum_kernel(const double *__restrict__ a, const double *__restrict__ b_t,
                                                       double *__restrict__ dst, int N)
{
    __shared__ double tile[32][33];
    const int j = blockIdx.y;
    const int kb = blockIdx.z * 32, ib = blockIdx.x * 32;       // b_t tile: rows k, columns i
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = kb + r, i = ib + threadIdx.x;
        if (k < N && i < N) tile[r][threadIdx.x] = b_t[((size_t)k * N + j) * N + i];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = ib + r, k = kb + threadIdx.x;
        if (i < N && k < N) { const size_t o = ((size_t)i * N + j) * N + k; dst[o] = a[o] + tile[threadIdx.x][r]; }
    }
}

int launch_fold_sum(State &st, const double *a, const double *b_t, double *dst)
{
    KernelTimer kt(ASORA_KERNEL_FINISH);
    hipLaunchKernelGGL(fold_sum_kernel, tile_grid(st.N), dim3(32, 8), 0, st.stream, a, b_t, dst, st.N);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// grid *= factor in place (the dilution of the density in a cosmological run, ref: pyc2ray/c2ray_base.py:248, for a grid that
// lives on the device)
__global__ void __launch_bounds__(256) scale_kernel(double *__restrict__ a, size_t n, double factor)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] *= factor;
}

int launch_scale(State &st, double *a, size_t n, double factor)
{
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, st.stream, a, n, factor);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_transpose(State &st, const double *src, double *dst, int N)
{
    hipLaunchKernelGGL(transpose_ik_kernel<false>, tile_grid(N), dim3(32, 8), 0, st.stream, src, dst, N);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Lines of the rate accumulators the sources can touch (State::reach_mask)
// ---------------------------------------------------------------------------------------------
// One thread per (source, transverse offset pair (u, v) of the square [-S, S]^2, layout): the chord of the sphere |d|^2 <= R2hi
// along the layout's contiguous axis at that (u, v), marked line by line.  A SUPERSET of what the raytrace rates (no periodic
// window, no octahedron bound, the generous radius of the geometry tables): a line marked in vain only costs its 32 bytes.
__global__ void __launch_bounds__(256) reach_mask_kernel(const int32_t *__restrict__ src_pos, int src_begin, int src_count, int N, int NL,
                                                         int S, double R2hi, unsigned char *__restrict__ mask, size_t bytes_one_layout)
{
    const int side = 2 * S + 1;
    const int per_src = 2 * side * side;                                          // (S <= N/2 <= 640: < 2^22)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= per_src) return;
    const int layout = r / (side * side);
    const int u = (r % (side * side)) / side - S, v = (r % side) - S;         // u: offset along the slowest axis of the layout, v: along j
    const double rest = R2hi - (double)u * u - (double)v * v;
    if (rest < 0.0) return;
    int half = (int)floor(sqrt(rest));
    if (half > N / 2) half = N / 2;                                           // (the whole axis)
    for (int sl = blockIdx.y; sl < src_count; sl += gridDim.y) {              // sources on the grid's second dimension (any count)
        const int s = src_begin + sl;
        const int i0 = src_pos[3 * s], j0 = src_pos[3 * s + 1], k0 = src_pos[3 * s + 2];
        // layout 0: line (i, j, k >> 3), chord along k;  layout 1: line (k, j, i >> 3), chord along i
        const int outer0 = layout == 0 ? i0 : k0, fast0 = layout == 0 ? k0 : i0;
        int outer = (outer0 + u) % N; if (outer < 0) outer += N;
        int j = (j0 + v) % N; if (j < 0) j += N;
        unsigned char *row = mask + (size_t)layout * bytes_one_layout + ((size_t)outer * N + j) * NL;
        for (int c = -half; c <= half; ++c) {
            int f = (fast0 + c) % N; if (f < 0) f += N;
            row[f >> 3] = 1;                                                   // (every writer writes 1)
        }
    }
}

// number of marked lines (both layouts)
__global__ void __launch_bounds__(256) reach_count_kernel(const unsigned char *__restrict__ mask, size_t n, unsigned long long *out)
{
    unsigned int c = 0;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) c += mask[q];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

int launch_reach_count(State &st, const unsigned char *mask, size_t bytes_both_layouts, unsigned long long *out_dev)
{
    ASORA_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), st.stream));
    hipLaunchKernelGGL(reach_count_kernel, dim3(1024), dim3(256), 0, st.stream, mask, bytes_both_layouts, out_dev);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_reach_mask(State &st, const int32_t *src_pos, int src_begin, int src_count, double R, unsigned char *mask, size_t bytes_one_layout)
{
    ASORA_HIP_TRY(hipMemsetAsync(mask, 0, 2 * bytes_one_layout, st.stream));
    if (src_count <= 0) return 0;
    const int N = st.N, NL = (N + 7) / 8;
    const double R2hi = R * R * (1.0 + 1e-9) + 1e-9;                          // as the geometry tables' outer bound (geometry.hip)
    const int S = (int)std::min((double)(N / 2), std::floor(std::sqrt(R2hi)));
    const int per_src = 2 * (2 * S + 1) * (2 * S + 1);
    hipLaunchKernelGGL(reach_mask_kernel, dim3((unsigned)((per_src + 255) / 256), (unsigned)std::min(src_count, 65535)), dim3(256), 0, st.stream,
                       src_pos, src_begin, src_count, N, NL, S, R2hi, mask, bytes_one_layout);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace asora
