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

namespace asora {

constexpr int CH_THREADS = 256;

// One cell: do_chemistry + the convergence test of evolve0D_global.
__device__ __forceinline__ void chemistry_cell(const ChemParams &p, double n, double T, double x0, double gamma,
                                               double &xav, double &xint, unsigned int &nconv)
{
    // single-precision parameters promoted to double, chemistry.f90:9-10
    const double min_frac_change = (double)1.0e-3f;
    const double min_frac_atoms = (double)1.0e-8f;
    const double eps = 1e-14;                                        // chemistry.f90:8

    const double xav_start = xav;                                    // chemistry.f90:91,99
    const double yh_av = 1.0 - xav;                                  // chemistry.f90:93

    // doric's temperature-only factors, chemistry.f90:257-262 (isothermal: same every iteration)
    const double brech0 = 1.0 * p.bh00 * pow(T / 1e4, p.albpow);
    const double acolh0 = p.colh0 * sqrt(T) * exp(-p.temph0 / T);
    // temperature convergence term of do_chemistry, chemistry.f90:185-186 (0 unless T is 0/NaN)
    const bool t_ok = fabs((T - T) / T) < min_frac_change;

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
        chemistry_cell(p, n.x, T.x, x0.x, g.x, xav.x, xint.x, nconv);
        chemistry_cell(p, n.y, T.y, x0.y, g.y, xav.y, xint.y, nconv);
        xi2[q] = xint;                                               // chemistry.f90:107-108
        xa2[q] = xav;
        sum1 += xint.x; sum0 += 1.0 - xint.x;                        // evolve.py:216-217
        sum1 += xint.y; sum0 += 1.0 - xint.y;
    }
    if ((p.ncell & 1) && blockIdx.x == 0 && threadIdx.x == 0) {      // odd cell count: the last cell
        const size_t idx = p.ncell - 1;
        double xav = p.xh_av[idx], xint;
        chemistry_cell(p, p.ndens[idx], p.temp[idx], p.xh[idx], p.phi[idx], xav, xint, nconv);
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

// out[q] = (accumulate ? out[q] : 0) + sum of the per-block partials, one block, fixed order
__global__ void __launch_bounds__(CH_THREADS) chemistry_reduce_kernel(const double *partial, int nblocks, double *out,
                                                                      int accumulate)
{
    __shared__ double r[3][CH_THREADS];
    for (int q = 0; q < 3; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += CH_THREADS) s += partial[q * nblocks + b];
        r[q][threadIdx.x] = s;
    }
    __syncthreads();
    for (int off = CH_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int q = 0; q < 3; ++q) r[q][threadIdx.x] += r[q][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) out[threadIdx.x] = (accumulate ? out[threadIdx.x] : 0.0) + r[threadIdx.x][0];
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
    hipLaunchKernelGGL(chemistry_reduce_kernel, dim3(1), dim3(CH_THREADS), 0, stream,
                       (const double *)p.red_partial, blocks, p.red_final, p.accumulate);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace asora
