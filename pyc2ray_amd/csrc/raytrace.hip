// raytrace.hip -- short-characteristics raytracing for gfx950 (MI355X).
//
// What is computed is exactly what the reference's GPU path computes
// (src/asora/raytracing.cu:79-339: do_all_sources_gpu + evolve0D_gpu + cinterp_gpu, and
//  src/asora/rates.cu:16-83): for every source, the incoming/outgoing HI column density of
// every cell by short-characteristics interpolation and the photon-conserving rate Gamma,
// summed over sources into phi_ion.
//
// How it is computed is different:
//   * work item = (source, octant).  The 8 sign-octants of a source only share the three
//     coordinate planes through the source, and a cell on such a plane depends only on cells
//     of the same plane (the "upstream" neighbour across a zero offset has bilinear weight
//     exactly 0, cinterp_gpu raytracing.cu:378-408 with sign(0)=+1), so each octant re-derives
//     its boundary planes and is otherwise independent: 8*NumSrc independent workgroups, no
//     inter-workgroup communication.
//   * inside an octant the sweep runs over CHEBYSHEV shells s = max(|di|,|dj|,|dk|) instead of
//     the reference's octahedral shells q = |di|+|dj|+|dk|.  The interpolation of a cell whose
//     dominant offset is s reads only cells whose dominant offset is s-1, so ONE trailing shell
//     is live (the octahedral order keeps three) and there are R+1 instead of ~sqrt(3)R+1
//     barriers.  Both orders are valid topological orders of the same dependency graph, so the
//     values are the same.
//   * the live shell (three faces: dk=s, dj=s, di=s) is double-buffered in LDS; the reference's
//     NUM_SRC_PAR x N^3 global scratch (memory.cu:65) does not exist.
//   * only cells that can receive a rate are evaluated: |d|^2 <= R^2, inside the periodic window
//     and inside the reference's octahedron q <= q_max.  Every upstream neighbour of such a cell
//     is strictly closer to the source, so the pruned cells never feed a kept one and phi_ion is
//     unchanged (the reference evaluates them into scratch and then discards them,
//     raytracing.cu:311-315).
//   * faces dj=s and di=s are rows along k, contiguous in the [i][j][k] grid.  Faces dk=s are
//     rows along i, so they read nHI and accumulate Gamma through [k][j][i] transposed copies
//     (rows contiguous again); the transposed accumulator is folded back once per call.
//   * nHI = ndens*(1-xh_av) is formed once per call (raytracing.cu:275-276 forms it per visit).
#include "asora_internal.hpp"

namespace asora {

constexpr int RT_THREADS = 256;
constexpr double FOURPI = 12.566370614359172463991853874177;   // raytracing.cu:12

// ---------------------------------------------------------------------------------------------
// Rates (src/asora/rates.cu)
// ---------------------------------------------------------------------------------------------

// photo_lookuptable, rates.cu:70-83 (== photorates.f90:130-147).  Indices are clamped to the
// last table element: the reference reads one past the end when NumTau == len(table) and
// tau >= 10^maxlogtau.
__device__ __forceinline__ double table_lookup(const double *__restrict__ table, double tau,
                                               const RtParams &p)
{
    const double logtau = log10(fmax(1.0e-20, tau));
    const double real_i = fmin(p.numtau_f, fmax(0.0, 1.0 + (logtau - p.minlogtau) / p.dlogtau));
    int i0 = (int)real_i;
    int i1 = min(p.NumTau, i0 + 1);
    const double residual = real_i - (double)i0;
    const int last = p.table_len - 1;
    i0 = min(i0, last);
    i1 = min(i1, last);
    const double t0 = table[i0];
    const double t1 = table[i1];
    return t0 + residual * (t1 - t0);
}

// photoion_rates_gpu rates.cu:16-41 / photoion_rates_test_gpu rates.cu:48-64
__device__ __forceinline__ double photo_rate(double flux, double cd_in, double cd_out, double vol,
                                             const RtParams &p)
{
    const double tau_in = cd_in * p.sig;
    const double tau_out = cd_out * p.sig;
    // TAU_PHOTO_LIMIT: rates.cu:7 (double 1e-7) or photorates.f90:69 (single 1e-7 promoted)
    const double limit = p.fortran_consts ? (double)1.0e-7f : 1.0e-7;
    if (p.grey) {
        const double prefact = flux * 1e48 / vol;
        if (fabs(tau_out - tau_in) > limit) return prefact * (exp(-tau_in) - exp(-tau_out));
        return prefact * (tau_out - tau_in) * exp(-tau_in);
    }
    const double prefact = flux / vol;
    if (fabs(tau_out - tau_in) > limit) {
        const double phi_in = prefact * table_lookup(p.thick, tau_in, p);
        const double phi_out = prefact * table_lookup(p.thick, tau_out, p);
        return phi_in - phi_out;
    }
    // rates.cu:37 uses tau_out, photorates.f90:121 uses tau_in
    return prefact * (tau_out - tau_in) * table_lookup(p.thin, p.fortran_consts ? tau_in : tau_out, p);
}

// The reference's distance test dist2/(dr*dr) <= Rmax*Rmax (raytracing.cu:302-305,315) with
// its own rounding: products and sums are kept un-fused so that a cell sitting exactly on the
// sphere is classified as the (un-contracted) reference classifies it.
__device__ __forceinline__ double dist2_unfused(int a, int b, int c, double dr)
{
#pragma clang fp contract(off)
    const double xs = dr * (double)a, ys = dr * (double)b, zs = dr * (double)c;
    return xs * xs + ys * ys + zs * zs;
}
__device__ __forceinline__ bool inside_radius(double dist2, double dr, double R2)
{
#pragma clang fp contract(off)
    return dist2 / (dr * dr) <= R2;
}

__device__ __forceinline__ int wrap_once(int x, int N)
{
    return x < 0 ? x + N : (x >= N ? x - N : x);
}

// ---------------------------------------------------------------------------------------------
// The octant kernel
// ---------------------------------------------------------------------------------------------
// Shell buffer layout (doubles), per shell t, three faces of stride W:
//   z-face (dk = t)          cell (a,b,t)  -> b*W + a          a fastest (row along i)
//   y-face (dj = t, dk < t)  cell (a,t,c)  -> W*W + a*W + c    c fastest (row along k)
//   x-face (di = t, dj,dk<t) cell (t,b,c)  -> 2*W*W + b*W + c  c fastest (row along k)
// a,b,c = |di|,|dj|,|dk|.  Face membership follows the reference's branch order z, y, x
// (raytracing.cu:394,446,491): ties go to z, then y.
template <bool GLOBAL_SCRATCH, bool DUMP>
__global__ void __launch_bounds__(RT_THREADS) raytrace_octant_kernel(const RtParams p)
{
    extern __shared__ double lds_shell[];

    const int blk = blockIdx.x;
    // blocks b and b+8 share an XCD (round-robin dispatch): keep the 8 octants of one source
    // on one XCD so that they share its L2 lines of nHI.  Speed only, never correctness.
    const int src_local = (blk & 7) + 8 * (blk >> 6);
    const int oct = (blk >> 3) & 7;
    if (src_local >= p.src_count) return;
    const int ns = p.src_begin + src_local;

    const int N = p.N;
    const int i0 = p.src_pos[3 * ns + 0];
    const int j0 = p.src_pos[3 * ns + 1];
    const int k0 = p.src_pos[3 * ns + 2];
    const double flux = p.src_flux[ns];
    const int sa = (oct & 1) ? -1 : 1, sb = (oct & 2) ? -1 : 1, sc = (oct & 4) ? -1 : 1;
    // periodic window of the reference (raytracing.cu:122-123,241)
    const int Ea = sa > 0 ? p.ext_pos : p.ext_neg;
    const int Eb = sb > 0 ? p.ext_pos : p.ext_neg;
    const int Ec = sc > 0 ? p.ext_pos : p.ext_neg;

    const int W = p.W, WW = W * W;
    double *prev = GLOBAL_SCRATCH ? p.shell_scratch + (size_t)blk * 6 * WW : lds_shell;
    double *cur = prev + 3 * WW;

    const double sig = p.sig, dr = p.dr;
    const double maxcd = p.fortran_consts ? (double)2e30f : 2e30;                    // raytracing.cu:15
    const double r3 = p.fortran_consts ? (double)1.7320507764816284 : 1.73205080757; // f90:608 / cu:435
    const double r2 = p.fortran_consts ? (double)1.4142135381698608 : 1.41421356237; // f90:609 / cu:439
    // integer |d|^2 this far from R^2 needs no floating-point classification
    const double R2lo = p.R2 * (1.0 - 1e-9) - 1e-9, R2hi = p.R2 * (1.0 + 1e-9) + 1e-9;

    unsigned int n_gamma = 0, n_eval = 0;

    // ---- shell 0: the source cell (raytracing.cu:285-294) -----------------------------------
    if (threadIdx.x == 0) {
        const size_t idx = ((size_t)i0 * N + j0) * N + k0;
        const double nHI = p.nhi[idx];
        const double path = 0.5 * dr;
        const double cd_out = 0.0 + nHI * path;
        prev[0] = cd_out;
        ++n_eval;
        if (oct == 0) {
            if (DUMP) p.dump[idx] = cd_out;
            const double phi = photo_rate(flux, 0.0, cd_out, dr * dr * dr, p) / nHI;
            unsafeAtomicAdd(&p.phi[idx], phi);
            ++n_gamma;
        }
    }
    __syncthreads();

    for (int s = 1; s <= p.S; ++s) {
        const double sd = (double)s;
        const double alam = (sd - 0.5) / sd;                 // raytracing.cu:397 in source-relative form
        const double rem = R2hi - sd * sd;
        const int um = rem >= 0.0 ? (int)fmin(sqrt(rem), 1.0e6) : -1;   // largest transverse offset inside the sphere
        const int m = min(s, um), m1 = min(s - 1, um);
        const int Az = min(m, Ea), Bz = min(m, Eb);
        const int Ay = min(m, Ea), Cy = min(m1, Ec);
        const int Bx = min(m1, Eb), Cx = min(m1, Ec);
        const int nz = (s <= Ec && um >= 0) ? (Az + 1) * (Bz + 1) : 0;
        const int ny = (s <= Eb && um >= 0) ? (Ay + 1) * (Cy + 1) : 0;
        const int nx = (s <= Ea && um >= 0) ? (Bx + 1) * (Cx + 1) : 0;
        const int ntot = nz + ny + nx;
        if (ntot == 0) break;                                // uniform: nothing further out either

        for (int t = threadIdx.x; t < ntot; t += RT_THREADS) {
            int a, b, c, U, V, face;
            if (t < nz) {
                const int row = t / (Az + 1);
                a = t - row * (Az + 1); b = row; c = s;
                U = a; V = b; face = 2;
            } else if (t < nz + ny) {
                const int r = t - nz, row = r / (Cy + 1);
                c = r - row * (Cy + 1); a = row; b = s;
                U = a; V = c; face = 1;
            } else {
                const int r = t - nz - ny, row = r / (Cx + 1);
                c = r - row * (Cx + 1); b = row; a = s;
                U = b; V = c; face = 0;
            }
            if (a + b + c > p.q_max) continue;                       // raytracing.cu:101,198
            const double dist2 = dist2_unfused(a, b, c, dr);
            const double dn2 = (double)(a * a + b * b + c * c);
            if (dn2 > R2hi) continue;
            if (dn2 >= R2lo && !inside_radius(dist2, dr, p.R2)) continue;   // raytracing.cu:315

            // ---- cinterp_gpu, raytracing.cu:345-535, in source-relative octant coordinates ---
            const double u = (double)U, v = (double)V;
            const double de = 2.0 * fabs(alam * u - (u - 0.5));
            const double df = 2.0 * fabs(alam * v - (v - 0.5));
            double w1 = (1. - de) * (1. - df);
            double w2 = (1. - df) * de;
            double w3 = (1. - de) * df;
            double w4 = de * df;
            // Upstream corners live in shell s-1.  A corner that would step across a zero offset
            // (U==0 -> U-1) or keep a transverse offset equal to s carries weight exactly 0
            // (de==1 resp. de==0 above) and is not fetched.
            const bool em = U >= 1, e0 = U <= s - 1, fm = V >= 1, f0 = V <= s - 1;
            const int sm = s - 1;
            int o1, o2, o3, o4;   // slots of (U-1,V-1) (U,V-1) (U-1,V) (U,V) in shell s-1
            if (face == 2) {
                o1 = (V - 1) * W + (U - 1); o2 = (V - 1) * W + U; o3 = V * W + (U - 1); o4 = V * W + U;
            } else if (face == 1) {
                // neighbour (a', sm, c'): on the z-face when c' == sm (tie -> z), else y-face
                const int ym = (V - 1 == sm) ? sm * W : WW + (V - 1);     // c' = V-1
                const int y0 = (V == sm) ? sm * W : WW + V;               // c' = V
                const bool zm = (V - 1 == sm), z0 = (V == sm);
                o1 = zm ? ym + (U - 1) : ym + (U - 1) * W;
                o2 = zm ? ym + U : ym + U * W;
                o3 = z0 ? y0 + (U - 1) : y0 + (U - 1) * W;
                o4 = z0 ? y0 + U : y0 + U * W;
            } else {
                // neighbour (sm, b', c'): z-face when c' == sm, else y-face when b' == sm, else x-face
                auto slot = [&](int bb, int cc) -> int {
                    return (cc == sm) ? bb * W + sm : ((bb == sm) ? WW + sm * W + cc : 2 * WW + bb * W + cc);
                };
                o1 = slot(U - 1, V - 1); o2 = slot(U, V - 1); o3 = slot(U - 1, V); o4 = slot(U, V);
            }
            const double c1 = (em && fm) ? prev[o1] : 0.0;
            const double c2 = (e0 && fm) ? prev[o2] : 0.0;
            const double c3 = (em && f0) ? prev[o3] : 0.0;
            const double c4 = (e0 && f0) ? prev[o4] : 0.0;
            w1 *= 1.0 / fmax(0.6, c1 * sig);                  // weightf_gpu raytracing.cu:33
            w2 *= 1.0 / fmax(0.6, c2 * sig);
            w3 *= 1.0 / fmax(0.6, c3 * sig);
            w4 *= 1.0 / fmax(0.6, c4 * sig);
            double cd_in = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
            if (s == 1 && (U == 1 || V == 1)) cd_in = ((U == 1 && V == 1) ? r3 : r2) * cd_in;
            const double path = sqrt((u * u + v * v) / (sd * sd) + 1.0) * dr;

            // ---- the cell itself, raytracing.cu:270-276,311-328 -----------------------------
            const int i = wrap_once(i0 + sa * a, N), j = wrap_once(j0 + sb * b, N), k = wrap_once(k0 + sc * c, N);
            const size_t idx = ((size_t)i * N + j) * N + k;
            const size_t idx_t = ((size_t)k * N + j) * N + i;
            const bool zt = p.z_transposed && face == 2;
            const double nHI = zt ? p.nhi_t[idx_t] : p.nhi[idx];
            const double cd_out = cd_in + nHI * path;
            const int own = face == 2 ? b * W + a : (face == 1 ? WW + a * W + c : 2 * WW + b * W + c);
            cur[own] = cd_out;
            ++n_eval;
            // a cell on an octant-boundary plane is rated by the octant with the + sign there
            const bool owner = (a > 0 || sa > 0) && (b > 0 || sb > 0) && (c > 0 || sc > 0);
            if (owner) {
                if (DUMP) p.dump[idx] = cd_out;
                if (cd_in <= maxcd) {
                    const double vol = dist2 * path * FOURPI;                 // raytracing.cu:307
                    const double phi = photo_rate(flux, cd_in, cd_out, vol, p) / nHI;
                    unsafeAtomicAdd(zt ? &p.phi_t[idx_t] : &p.phi[idx], phi);
                    ++n_gamma;
                }
            }
        }
        __syncthreads();
        double *tmp = prev; prev = cur; cur = tmp;
    }

    // work accounting: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_gamma += __shfl_down(n_gamma, off);
        n_eval += __shfl_down(n_eval, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&p.counters[0], (unsigned long long)n_gamma);
        atomicAdd(&p.counters[1], (unsigned long long)n_eval);
    }
}

// ---------------------------------------------------------------------------------------------
// N^3 helper kernels
// ---------------------------------------------------------------------------------------------

// nhi[i][j][k] = ndens*(1-xh_av);  nhi_t[k][j][i] = same (tiled transpose of the (i,k) planes).
// block (32,8): tile 32(i) x 32(k) of one j.
template <bool WITH_T>
__global__ void __launch_bounds__(256) prepare_nhi_kernel(const double *__restrict__ nd, const double *__restrict__ xh,
                                                          double *__restrict__ nhi, double *__restrict__ nhi_t, int N)
{
    __shared__ double tile[32][33];
    const int j = blockIdx.y;
    const int ib = blockIdx.z * 32, kb = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = ib + r, k = kb + threadIdx.x;
        if (i < N && k < N) {
            const size_t idx = ((size_t)i * N + j) * N + k;
            const double v = nd[idx] * (1.0 - xh[idx]);       // raytracing.cu:276
            nhi[idx] = v;
            if (WITH_T) tile[r][threadIdx.x] = v;
        }
    }
    if (!WITH_T) return;
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = kb + r, i = ib + threadIdx.x;
        if (i < N && k < N) nhi_t[((size_t)k * N + j) * N + i] = tile[threadIdx.x][r];
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

int launch_prepare_nhi(State &st, bool need_transposed)
{
    KernelTimer kt(ASORA_KERNEL_PREP);
    const int N = st.N;
    if (need_transposed)
        hipLaunchKernelGGL(prepare_nhi_kernel<true>, tile_grid(N), dim3(32, 8), 0, st.stream,
                           st.grid[ASORA_GRID_NDENS], st.grid[ASORA_GRID_XH_AV], st.nhi, st.nhi_t, N);
    else
        hipLaunchKernelGGL(prepare_nhi_kernel<false>, tile_grid(N), dim3(32, 8), 0, st.stream,
                           st.grid[ASORA_GRID_NDENS], st.grid[ASORA_GRID_XH_AV], st.nhi, st.nhi_t, N);
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

int launch_transpose(State &st, const double *src, double *dst, int N)
{
    hipLaunchKernelGGL(transpose_ik_kernel<false>, tile_grid(N), dim3(32, 8), 0, st.stream, src, dst, N);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Host driver of the octant kernel (the role of do_all_sources_gpu's batch loop,
// raytracing.cu:101-143)
// ---------------------------------------------------------------------------------------------
static const size_t LDS_LIMIT_BYTES = 160 * 1024;

int launch_raytrace(State &st, RtParams &p, bool dump)
{
    const int N = p.N;
    // geometry shared by all sources
    p.q_max = (int)std::ceil(1.73205080757 * std::min(p.R, 1.73205080757 * N / 2.0));   // raytracing.cu:14,101
    p.ext_pos = N / 2 - 1 + (N % 2);                                                      // raytracing.cu:122
    p.ext_neg = N / 2;                                                                    // raytracing.cu:123
    p.R2 = p.R * p.R;
    const double R2hi = p.R2 * (1.0 + 1e-9) + 1e-9;
    const int Emax = std::max(p.ext_pos, p.ext_neg);
    int S = 0;
    if (std::isfinite(R2hi)) S = (int)std::min((double)Emax, std::floor(std::sqrt(R2hi)));
    else S = Emax;
    int W = 1;
    for (int s = 1; s <= S; ++s) {
        const double rem = R2hi - (double)s * s;
        if (rem < 0) break;
        const double umd = std::sqrt(rem);
        const int um = umd > 1e9 ? 1000000000 : (int)umd;
        W = std::max(W, std::min(std::min(s, um), Emax) + 1);
    }
    p.S = S;
    p.W = W;

    const size_t shell_bytes = (size_t)6 * W * W * sizeof(double);
    const bool use_lds = shell_bytes <= LDS_LIMIT_BYTES;

    int done = 0;
    while (done < p.src_count || (p.src_count == 0 && done == 0)) {
        if (p.src_count == 0) break;
        int batch = p.src_count - done;
        if (!use_lds) {
            // bound the global shell scratch to ~2 GiB per launch (the reference's source batching,
            // raytracing.cu:126, reappears only for traces whose shells outgrow LDS)
            const size_t per_src = 8 * shell_bytes;
            const size_t budget = (size_t)2 << 30;
            int max_batch = (int)std::max<size_t>(8, (budget / per_src) / 8 * 8);
            batch = std::min(batch, max_batch);
            const size_t need = (size_t)64 * ((batch + 7) / 8) * shell_bytes;
            if (need > st.shell_scratch_bytes) {
                if (st.shell_scratch) ASORA_HIP_TRY(hipFree(st.shell_scratch));
                st.shell_scratch = nullptr;
                st.shell_scratch_bytes = 0;
                ASORA_HIP_TRY(hipMalloc(&st.shell_scratch, need));
                st.shell_scratch_bytes = need;
            }
        }
        RtParams q = p;
        q.src_begin = p.src_begin + done;
        q.src_count = batch;
        q.shell_scratch = use_lds ? nullptr : st.shell_scratch;
        const unsigned grid = 64u * (unsigned)((batch + 7) / 8);
        {
            KernelTimer kt(ASORA_KERNEL_RAYTRACE);
            if (use_lds) {
                if (dump) {
                    ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<false, true>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shell_bytes));
                    hipLaunchKernelGGL((raytrace_octant_kernel<false, true>), dim3(grid), dim3(RT_THREADS),
                                       shell_bytes, st.stream, q);
                } else {
                    ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<false, false>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shell_bytes));
                    hipLaunchKernelGGL((raytrace_octant_kernel<false, false>), dim3(grid), dim3(RT_THREADS),
                                       shell_bytes, st.stream, q);
                }
            } else {
                if (dump)
                    hipLaunchKernelGGL((raytrace_octant_kernel<true, true>), dim3(grid), dim3(RT_THREADS), 0,
                                       st.stream, q);
                else
                    hipLaunchKernelGGL((raytrace_octant_kernel<true, false>), dim3(grid), dim3(RT_THREADS), 0,
                                       st.stream, q);
            }
            ASORA_HIP_TRY(hipGetLastError());
        }
        done += batch;
    }
    return 0;
}

} // namespace asora
