// chemistry.hip -- fused photo-ionisation chemistry pass for gfx950 (MI355X).
//
// One kernel does what the reference does on the host in four steps per outer iteration
// (pyc2ray/evolve.py:210-217): global_pass + evolve0D_global + do_chemistry + doric
// (src/c2ray/chemistry.f90:13-316) for every cell, the count of non-converged cells
// (chemistry.f90:99-104) and the two sums sum(xh_intermed), sum(1-xh_intermed) that
// evolve3D forms with numpy afterwards.  The reference runs this serially on one CPU core
// ("GPU chemistry" is a TODO in its README); here it is one cell per lane, 56 B of HBM traffic
// per cell (5 loads, 2 stores), reductions fused.
//
// The reductions are two-stage and order-fixed (per-block partials, then one block), so every
// rank of a multi-GPU run derives bit-identical convergence scalars from the same grids.
#include "asora_internal.hpp"

#include <algorithm>

namespace asora {

constexpr int CH_THREADS = 256;

// doric's temperature-only factors (chemistry.f90:257-262) and the temperature term of do_chemistry's exit test
// (chemistry.f90:185-186).  The path is isothermal (the temperature never changes inside do_chemistry) and most
// grids are uniform in T, so a lane keeps the factors of the last temperature it saw: pow, sqrt and exp -- half of the
// pass's instructions -- are then evaluated once per lane instead of once per cell, with bit-identical results.
struct TempFactors {
    double T = -1.0;           // temperature the factors below belong to (no cell has T = -1: first use always computes)
    double brech0 = 0.0, acolh0 = 0.0;
    bool t_ok = false;
};
__device__ __forceinline__ void temperature_factors(const ChemParams &p, double T, TempFactors &f)
{
    if (T == f.T) return;
    f.T = T;
    f.brech0 = 1.0 * p.bh00 * pow(T / 1e4, p.albpow);
    f.acolh0 = p.colh0 * sqrt(T) * exp(-p.temph0 / T);
    // |(T - T)/T| < min_frac_change: 0 unless T is 0, infinite or NaN (then NaN: the test fails, as in the reference)
    f.t_ok = fabs((T - T) / T) < (double)1.0e-3f;
}

// One cell: do_chemistry + the convergence test of evolve0D_global.
__device__ __forceinline__ void chemistry_cell(const ChemParams &p, double n, double x0, double gamma,
                                               double &xav, double &xint, unsigned int &nconv, const TempFactors &tf)
{
    // single-precision parameters promoted to double, chemistry.f90:9-10
    const double min_frac_change = (double)1.0e-3f;
    const double min_frac_atoms = (double)1.0e-8f;
    const double eps = 1e-14;                                        // chemistry.f90:8

    const double xav_start = xav;                                    // chemistry.f90:91,99
    const double yh_av = 1.0 - xav;                                  // chemistry.f90:93

    const double brech0 = tf.brech0, acolh0 = tf.acolh0;
    const bool t_ok = tf.t_ok;

    int nit = 0;
    for (;;) {                                                       // do_chemistry, chemistry.f90:146-203
        nit += 1;
        const double xav_old = xav;
        const double de = n * (xav + p.abu_c);                       // chemistry.f90:162
        // doric, chemistry.f90:279-311
        const double aih0 = gamma + de * acolh0;
        const double delth = aih0 + de * brech0;
        const double eqxh = aih0 / delth;
        const double deltht = delth * p.dt;
        const double ee = exp(-deltht);
        xint = (x0 - eqxh) * ee + eqxh;
        if (xint < eps) xint = eps;
        const double avg = (deltht < (double)1.0e-8f) ? 1.0 : (1.0 - ee) / deltht;
        xav = eqxh + (x0 - eqxh) * avg;
        if (xav < eps) xav = eps;
        if ((fabs((xav - xav_old) / (1.0 - xav)) < min_frac_change || (1.0 - xav < min_frac_atoms)) && t_ok)
            break;                                                   // chemistry.f90:182-189
        if (nit > 400) break;                                        // chemistry.f90:192
    }
    if (fabs(xav - xav_start) > min_frac_change && fabs((xav - xav_start) / yh_av) > min_frac_change &&
        yh_av > min_frac_atoms)
        nconv += 1;                                                  // chemistry.f90:100-104
}

// Two consecutive cells per lane: 16-byte loads and stores (the grids are 256-byte aligned).
__global__ void __launch_bounds__(CH_THREADS) chemistry_kernel(const ChemParams p)
{
    double sum1 = 0.0, sum0 = 0.0;
    unsigned int nconv = 0;
    TempFactors tf;

    const size_t npair = p.ncell / 2;
    const size_t stride = (size_t)gridDim.x * CH_THREADS;
    const double2 *nd2 = reinterpret_cast<const double2 *>(p.ndens);
    const double2 *tp2 = reinterpret_cast<const double2 *>(p.temp);
    const double2 *x02 = reinterpret_cast<const double2 *>(p.xh);
    const double2 *ph2 = reinterpret_cast<const double2 *>(p.phi);
    double2 *xa2 = reinterpret_cast<double2 *>(p.xh_av);
    double2 *xi2 = reinterpret_cast<double2 *>(p.xh_intermed);
    for (size_t q = (size_t)blockIdx.x * CH_THREADS + threadIdx.x; q < npair; q += stride) {
        const double2 n = nd2[q], T = tp2[q], x0 = x02[q], g = ph2[q];
        double2 xav = xa2[q], xint;
        temperature_factors(p, T.x, tf);
        chemistry_cell(p, n.x, x0.x, g.x, xav.x, xint.x, nconv, tf);
        temperature_factors(p, T.y, tf);
        chemistry_cell(p, n.y, x0.y, g.y, xav.y, xint.y, nconv, tf);
        xi2[q] = xint;                                               // chemistry.f90:107-108
        xa2[q] = xav;
        sum1 += xint.x; sum0 += 1.0 - xint.x;                        // evolve.py:216-217
        sum1 += xint.y; sum0 += 1.0 - xint.y;
    }
    if ((p.ncell & 1) && blockIdx.x == 0 && threadIdx.x == 0) {      // odd cell count: the last cell
        const size_t idx = p.ncell - 1;
        double xav = p.xh_av[idx], xint;
        temperature_factors(p, p.temp[idx], tf);
        chemistry_cell(p, p.ndens[idx], p.xh[idx], p.phi[idx], xav, xint, nconv, tf);
        p.xh_intermed[idx] = xint;
        p.xh_av[idx] = xav;
        sum1 += xint; sum0 += 1.0 - xint;
    }

    // block reduction in a fixed order
    __shared__ double r1[CH_THREADS], r0[CH_THREADS];
    __shared__ unsigned int rc[CH_THREADS];
    r1[threadIdx.x] = sum1; r0[threadIdx.x] = sum0; rc[threadIdx.x] = nconv;
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            r1[threadIdx.x] += r1[threadIdx.x + off];
            r0[threadIdx.x] += r0[threadIdx.x + off];
            rc[threadIdx.x] += rc[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.red_partial[blockIdx.x] = r1[0];
        p.red_partial[p.red_blocks + blockIdx.x] = r0[0];
        p.red_partial[2 * p.red_blocks + blockIdx.x] = (double)rc[0];   // exact: < 2^53
    }
}

// out[q] = (accumulate ? out[q] : 0) + sum of the per-block partials, one block, fixed order.
// With `status` the convergence test of the evolve loop follows at once (pyc2ray/evolve.py:216-236):
// relative change of the two sums against the previous iteration, `conv_flag < conv_criterion or both changes
// below convergence_fraction`; the iteration is booked in the status block's history ring.
constexpr int RED_THREADS = 1024;
__global__ void __launch_bounds__(RED_THREADS) chemistry_reduce_kernel(const double *partial, int nblocks, double *out,
                                                                       int accumulate, EvolveStatus *status,
                                                                       const EvolveStatus *gate = nullptr)
{
    if (status && status->done) return;
    if (gate && gate->done) return;
    // fixed order: thread t sums partials t, t + 1024, ...; lanes of a wave are combined by a butterfly of DPP-free
    // shuffles in a fixed pattern, the 16 waves through LDS -- the same bits on every launch and on every rank
    double v[3];
    for (int q = 0; q < 3; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += RED_THREADS) s += partial[(size_t)q * nblocks + b];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        v[q] = s;
    }
    __shared__ double r[3][RED_THREADS / 64];
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 3; ++q) r[q][threadIdx.x >> 6] = v[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[3];
        for (int q = 0; q < 3; ++q) {
            double s = 0.0;
            for (int w = 0; w < RED_THREADS / 64; ++w) s += r[q][w];
            tot[q] = s;
            out[q] = (accumulate ? out[q] : 0.0) + s;
        }
        if (status) {
            const double sum1 = tot[0], sum0 = tot[1], nconv = tot[2];
            const double rel1 = sum1 > 0.0 ? fabs((sum1 - status->prev1) / sum1) : 1.0;      // evolve.py:219-227
            const double rel0 = sum0 > 0.0 ? fabs((sum0 - status->prev0) / sum0) : 1.0;
            const bool converged = (nconv < status->conv_criterion) ||
                                   (rel1 < status->conv_fraction && rel0 < status->conv_fraction);   // evolve.py:232
            double *h = status->hist[status->niter % EVOLVE_HIST];
            h[0] = nconv; h[1] = sum1; h[2] = sum0; h[3] = rel1; h[4] = rel0;
            status->prev1 = sum1; status->prev0 = sum0;                                       // evolve.py:234-235
            status->niter += 1;
            if (converged) status->done = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The tiled pass: planes [i_begin, i_end), one 32(i) x 32(k) tile of one j per workgroup iteration
// ---------------------------------------------------------------------------------------------
// What the host of the reference does between the raytrace and the next raytrace (pyc2ray/evolve.py:200-240:
// reshape phi_ion, global_pass, the two sums, ravel xh_av for the next upload) plus this build's own helper
// passes (fold of the z-face accumulator, nHI in both layouts, zeroing of the accumulators) in ONE sweep over
// the grids: per cell 6 loads (ndens, temp, xh, xh_av, two accumulators) and 6 stores (xh_av, xh_intermed,
// nHI twice, two zeros) = 96 B (88 without the temperature), against 56 + 32 + 24 + 16 B of the four separate passes.
// The folded rate itself is NOT stored: the pass leaves the accumulators it read intact and zeroes the OTHER pair for the
// next raytrace (ChemTileParams.zero_a / zero_t), so the rates of the last iteration can be folded when someone asks.
// The [k][j][i] twins are read and written through LDS tiles so that their rows are contiguous as well.
// waves per SIMD the register allocation of the tiled pass must leave room for (a streaming pass: more waves, more
// bytes in flight)
#ifndef ASORA_CHEM_MIN_WAVES
#define ASORA_CHEM_MIN_WAVES 1
#endif
// The grids the pass reads are streams: every element once, nothing another workgroup wants again.  Loaded non-temporal they do not
// take the place of the lines the seven write streams are being combined in: a bare kernel with the pass's 5 read and 7 write streams
// moves 6.3 TB/s instead of 5.3 (tools/micro/stream_peak.hip; non-temporal STORES cost 2-5 % there, and in the pass itself -- any of
// its three groups of write streams -- change nothing beyond the noise: profiles/r05_ab_chem_ntstores.txt).  The pass: -8 % in either
// state of a box (0.248 -> 0.229 ms, 0.306 -> 0.281 ms; profiles/r05_ab_chem_ntloads.txt).
#ifndef ASORA_CHEM_NT_LOADS
#define ASORA_CHEM_NT_LOADS 1
#endif
__device__ __forceinline__ double stream_load(const double *q)
{
#if ASORA_CHEM_NT_LOADS
    return __builtin_nontemporal_load(q);
#else
    return *q;
#endif
}

template <bool FOLD, bool EMIT, bool UNIFORM_T>
__global__ void __launch_bounds__(CH_THREADS, ASORA_CHEM_MIN_WAVES) chemistry_tile_kernel(const ChemTileParams p)
{
    if (p.status && p.status->done) return;
    __shared__ double tile_g[32][33], tile_n[32][33];
    __shared__ double r1[CH_THREADS], r0[CH_THREADS];
    __shared__ unsigned int rc[CH_THREADS];

    const int N = p.N;
    const int NL = (N + 7) >> 3;                                   // lines of 8 cells per row (ChemTileParams::reach_a / reach_t)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int kb = blockIdx.x * 32, ib = p.i_begin + blockIdx.z * 32;
    ChemParams cp;
    cp.dt = p.dt; cp.bh00 = p.bh00; cp.albpow = p.albpow; cp.colh0 = p.colh0; cp.temph0 = p.temph0; cp.abu_c = p.abu_c;

    double sum1 = 0.0, sum0 = 0.0;
    unsigned int nconv = 0;
    // UNIFORM_T: the whole grid has ONE temperature (probed when the grid was uploaded, temp_probe_kernel): its factors
    // come with the parameters -- no temperature load (96 B per cell instead of 104), no pow / sqrt / exp code in the
    // kernel (fewer registers: more waves in flight)
    TempFactors tf;
    if (UNIFORM_T) { tf.T = p.uniform_T; tf.brech0 = p.uniform_brech0; tf.acolh0 = p.uniform_acolh0; tf.t_ok = p.uniform_t_ok != 0; }
    for (int j = blockIdx.y; j < N; j += gridDim.y) {
        if (FOLD) {
            if (j != (int)blockIdx.y) __syncthreads();             // the tiles of the previous j have been consumed
            for (int r = ty; r < 32; r += 8) {
                const int k = kb + r, i = ib + tx;
                if (k < N && i < p.i_end) {
                    const size_t o = ((size_t)k * N + j) * N + i;
                    // a line no source reaches holds zeros in both accumulator pairs and keeps them: neither read nor zeroed
                    const bool reached = !p.reach_t || p.reach_t[((size_t)k * N + j) * NL + (i >> 3)] != 0;
                    tile_g[r][tx] = reached ? stream_load(p.gamma_t + o) : 0.0;
                    if (EMIT && reached) p.zero_t[o] = 0.0;
                }
            }
            __syncthreads();
        }
        for (int r = ty; r < 32; r += 8) {
            const int i = ib + r, k = kb + tx;
            if (i < p.i_end && k < N) {
                const size_t idx = ((size_t)i * N + j) * N + k;
                const bool reached = !FOLD || !p.reach_a || p.reach_a[((size_t)i * N + j) * NL + (k >> 3)] != 0;
                double g = reached ? stream_load(p.gamma + idx) : 0.0;
                if (FOLD) g += tile_g[tx][r];
                if (p.phi_out) p.phi_out[idx] = g;            // (the summed rates, where someone keeps them: all-reduce loop)
                if (EMIT && reached) p.zero_a[idx] = 0.0;
                const double n = stream_load(p.ndens + idx);
                double xav = stream_load(p.xh_av_in + idx), xint;
                if (!UNIFORM_T) temperature_factors(cp, stream_load(p.temp + idx), tf);
                chemistry_cell(cp, n, stream_load(p.xh + idx), g, xav, xint, nconv, tf);
                p.xh_intermed[idx] = xint;                           // chemistry.f90:107-108
                p.xh_av[idx] = xav;
                sum1 += xint; sum0 += 1.0 - xint;                    // evolve.py:216-217
                if (EMIT) {
                    const double v = n * (1.0 - xav);                // raytracing.cu:276, for the next raytrace
                    p.nhi[idx] = v;
                    tile_n[r][tx] = v;
                }
            }
        }
        if (EMIT) {
            __syncthreads();
            for (int r = ty; r < 32; r += 8) {
                const int k = kb + r, i = ib + tx;
                if (k < N && i < p.i_end) p.nhi_t[((size_t)k * N + j) * N + i] = tile_n[tx][r];
            }
        }
    }

    r1[threadIdx.x] = sum1; r0[threadIdx.x] = sum0; rc[threadIdx.x] = nconv;
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            r1[threadIdx.x] += r1[threadIdx.x + off];
            r0[threadIdx.x] += r0[threadIdx.x + off];
            rc[threadIdx.x] += rc[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        p.red_partial[b] = r1[0];
        p.red_partial[(size_t)p.red_stride + b] = r0[0];
        p.red_partial[2 * (size_t)p.red_stride + b] = (double)rc[0];
    }
}

// ---------------------------------------------------------------------------------------------
// Is the temperature grid uniform?  (once per upload of the grid and set of chemistry constants)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) temp_minmax_kernel(const double *__restrict__ temp, size_t n, double *partial)
{
    double lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    for (size_t q = (size_t)blockIdx.x * CH_THREADS + threadIdx.x; q < n; q += (size_t)gridDim.x * CH_THREADS) {
        const double t = temp[q];
        nan = nan || (t != t);
        lo = fmin(lo, t); hi = fmax(hi, t);
    }
    __shared__ double slo[CH_THREADS], shi[CH_THREADS];
    __shared__ int snan[CH_THREADS];
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; snan[threadIdx.x] = nan ? 1 : 0;
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + off]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + off]);
            snan[threadIdx.x] |= snan[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = slo[0];
        partial[gridDim.x + blockIdx.x] = shi[0];
        partial[2 * gridDim.x + blockIdx.x] = (double)snan[0];
    }
}

// out[0] = 1 if every cell holds the same (non-NaN) temperature, out[1..4] = that temperature and its factors, evaluated
// by the same device code the general path runs per cell (bit-identical results on both paths)
__global__ void __launch_bounds__(CH_THREADS) temp_probe_final_kernel(const double *partial, int nblocks, double bh00, double albpow,
                                                                      double colh0, double temph0, double *out)
{
    double lo = INFINITY, hi = -INFINITY, nan = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += CH_THREADS) {
        lo = fmin(lo, partial[b]); hi = fmax(hi, partial[nblocks + b]); nan += partial[2 * nblocks + b];
    }
    __shared__ double slo[CH_THREADS], shi[CH_THREADS], snan[CH_THREADS];
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; snan[threadIdx.x] = nan;
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + off]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + off]);
            snan[threadIdx.x] += snan[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    lo = slo[0]; hi = shi[0]; nan = snan[0];
    ChemParams p;
    p.bh00 = bh00; p.albpow = albpow; p.colh0 = colh0; p.temph0 = temph0;
    TempFactors tf;
    temperature_factors(p, lo, tf);
    out[0] = (lo == hi && nan == 0.0) ? 1.0 : 0.0;
    out[1] = lo; out[2] = tf.brech0; out[3] = tf.acolh0; out[4] = tf.t_ok ? 1.0 : 0.0;
}

int launch_temp_probe(State &st, const double *temp, size_t n, double bh00, double albpow, double colh0, double temph0,
                      double *out_dev)
{
    const int blocks = (int)std::min<size_t>(std::max<size_t>(1, (n + CH_THREADS - 1) / CH_THREADS), (size_t)st.red_blocks);
    hipLaunchKernelGGL(temp_minmax_kernel, dim3(blocks), dim3(CH_THREADS), 0, st.stream, temp, n, st.red_partial);
    ASORA_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(temp_probe_final_kernel, dim3(1), dim3(CH_THREADS), 0, st.stream, (const double *)st.red_partial, blocks, bh00,
                       albpow, colh0, temph0, out_dev);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Sum of a grid in a fixed order (the means of the reference's log line, evolve.py:160, without a host pass)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) grid_sum_kernel(const double *__restrict__ a, size_t n, double *partial)
{
    double s = 0.0;
    for (size_t q = (size_t)blockIdx.x * CH_THREADS + threadIdx.x; q < n; q += (size_t)gridDim.x * CH_THREADS) s += a[q];
    __shared__ double sh[CH_THREADS];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(CH_THREADS) grid_sum_final_kernel(const double *partial, int nblocks, double *out)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += CH_THREADS) s += partial[b];
    __shared__ double sh[CH_THREADS];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

int launch_grid_sum(State &st, const double *a, size_t n, double *out_dev)
{
    const int blocks = (int)std::min<size_t>(std::max<size_t>(1, (n + CH_THREADS - 1) / CH_THREADS), (size_t)st.red_blocks);
    hipLaunchKernelGGL(grid_sum_kernel, dim3(blocks), dim3(CH_THREADS), 0, st.stream, a, n, st.red_partial);
    ASORA_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grid_sum_final_kernel, dim3(1), dim3(CH_THREADS), 0, st.stream, (const double *)st.red_partial, blocks, out_dev);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int chemistry_reduction_blocks(const State &st) { return st.cu_count * 8; }

int launch_chemistry(State &st, ChemParams &p, hipStream_t stream)
{
    const size_t want = std::max<size_t>(1, (p.ncell / 2 + CH_THREADS - 1) / CH_THREADS);
    const int blocks = (int)std::min<size_t>(want, (size_t)p.red_blocks);
    ChemParams q = p;
    q.red_blocks = blocks;
    {
        KernelTimer kt(ASORA_KERNEL_CHEMISTRY);
        hipLaunchKernelGGL(chemistry_kernel, dim3(blocks), dim3(CH_THREADS), 0, stream, q);
        ASORA_HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(chemistry_reduce_kernel, dim3(1), dim3(RED_THREADS), 0, stream,
                       (const double *)p.red_partial, blocks, p.red_final, p.accumulate, (EvolveStatus *)nullptr,
                       (const EvolveStatus *)nullptr);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// Workgroups of a tiled pass over `planes` i-planes: (k tiles, j chunks, i tiles).  Every workgroup walks its
// (i,k) tile through N / (j chunks) values of j; the chunk count gives ~16 workgroups per CU (8 measured 10 % slower).
static dim3 tile_pass_grid(const State &st, int N, int planes)
{
    const unsigned kt = (N + 31) / 32, it = (planes + 31) / 32;
    const unsigned want = (unsigned)st.cu_count * 16u;
    unsigned jc = std::max(1u, std::min((unsigned)N, (want + kt * it - 1) / (kt * it)));
    return dim3(kt, jc, it);
}

size_t chemistry_tile_blocks(const State &st, int N, int planes)
{
    const dim3 g = tile_pass_grid(st, N, planes);
    return (size_t)g.x * g.y * g.z;
}

int launch_chemistry_tiles(State &st, ChemTileParams &p, hipStream_t stream)
{
    const int planes = p.i_end - p.i_begin;
    if (planes <= 0) return 0;
    const dim3 grid = tile_pass_grid(st, p.N, planes);
    const size_t blocks = (size_t)grid.x * grid.y * grid.z;
    if (3 * blocks > st.red_cap) return fail(11, "chemistry: reduction buffer too small (internal error)");
    ChemTileParams q = p;
    q.red_stride = (int)blocks;
    {
        KernelTimer kt(ASORA_KERNEL_CHEMISTRY, stream);
        const bool u = p.uniform != 0;
        if (p.fold && p.emit) {
            if (u) hipLaunchKernelGGL((chemistry_tile_kernel<true, true, true>), grid, dim3(CH_THREADS), 0, stream, q);
            else   hipLaunchKernelGGL((chemistry_tile_kernel<true, true, false>), grid, dim3(CH_THREADS), 0, stream, q);
        } else if (!p.fold && !p.emit) {
            if (u) hipLaunchKernelGGL((chemistry_tile_kernel<false, false, true>), grid, dim3(CH_THREADS), 0, stream, q);
            else   hipLaunchKernelGGL((chemistry_tile_kernel<false, false, false>), grid, dim3(CH_THREADS), 0, stream, q);
        } else if (!p.fold && p.emit) {       // rates already summed over both layouts (and over the ranks): asora_evolve_slab_fold_all
            if (u) hipLaunchKernelGGL((chemistry_tile_kernel<false, true, true>), grid, dim3(CH_THREADS), 0, stream, q);
            else   hipLaunchKernelGGL((chemistry_tile_kernel<false, true, false>), grid, dim3(CH_THREADS), 0, stream, q);
        } else return fail(11, "chemistry: unsupported fold/emit combination (internal error)");
        ASORA_HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(chemistry_reduce_kernel, dim3(1), dim3(RED_THREADS), 0, stream,
                       (const double *)p.red_partial, (int)blocks, p.red_final, p.accumulate,
                       p.local_sums ? (EvolveStatus *)nullptr : p.status, (const EvolveStatus *)(p.local_sums ? p.status : nullptr));
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// The convergence test of the evolve loop on its own (pyc2ray/evolve.py:216-236): multi-GPU, where the three sums of a rank's
// slab pass are first summed over the ranks (in place, RCCL) -- every rank then evaluates this on identical bits and takes
// the same decision.  Same arithmetic and bookkeeping as the tail of chemistry_reduce_kernel.
__global__ void convergence_test_kernel(const double *sums, EvolveStatus *status)
{
    if (status->done) return;
    const double sum1 = sums[0], sum0 = sums[1], nconv = sums[2];
    const double rel1 = sum1 > 0.0 ? fabs((sum1 - status->prev1) / sum1) : 1.0;      // evolve.py:219-227
    const double rel0 = sum0 > 0.0 ? fabs((sum0 - status->prev0) / sum0) : 1.0;
    const bool converged = (nconv < status->conv_criterion) ||
                           (rel1 < status->conv_fraction && rel0 < status->conv_fraction);   // evolve.py:232
    double *h = status->hist[status->niter % EVOLVE_HIST];
    h[0] = nconv; h[1] = sum1; h[2] = sum0; h[3] = rel1; h[4] = rel0;
    status->prev1 = sum1; status->prev0 = sum0;                                       // evolve.py:234-235
    status->niter += 1;
    if (converged) status->done = 1;
}

int launch_convergence_test(State &st, const double *sums, EvolveStatus *status)
{
    hipLaunchKernelGGL(convergence_test_kernel, dim3(1), dim3(1), 0, st.stream, sums, status);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace asora
