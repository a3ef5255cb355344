// subbox.hip -- the reference's CPU raytracing semantics on gfx950 (MI355X).
//
// libc2ray.raytracing.do_all_sources (src/c2ray/raytracing.f90:52-567) differs from the ASORA GPU path in
// what it traverses and in what it returns:
//   * the trace of a source covers a CUBE around it, grown in sub-boxes of `subboxsize` cells until the
//     photons leaving the current box drop below loss_fraction of the source's output or the range
//     min(max_subbox, N/2) is reached (do_source, f90:127-249); rates are only deposited inside
//     R_max_LLS and below the column-density cap (f90:474-478), but column densities are carried on;
//   * it returns the number of sub-boxes used, the photon loss through the last boxes, the heating rates
//     and the column-density grid of the last source;
//   * constants of the Fortran flavour (f90:608-609 single-precision sqrt(2), sqrt(3); photorates.f90:69,121).
//
// Nested sub-boxes are exactly ranges of Chebyshev shells s = max(|di|,|dj|,|dk|), the order the ASORA kernel
// (raytrace.hip) already sweeps in, so sub-box n of a source is "shells (b_{n-1}, b_n] of its 8 octants" and the
// box boundary is shell b_n.  The decision to grow a box needs the loss summed over the 8 octant workgroups
// of the source, so the sweep is cut into one launch per sub-box: a launch sweeps the shell range of every
// source that is still active, keeps the trailing shell of each octant in a global (L2-resident) buffer for the
// next launch and adds the boundary loss per source; a one-wave kernel then decides per source.  No workgroup
// ever waits for another one.
//
// Geometry is generated on the fly (a cube has no sphere to prune against and its tables would be N^3/8
// entries): shell s of an octant is its z-face {(a,b,s)}, y-face {(a,s,c), c<s} and x-face {(s,b,c), b,c<s},
// restricted to the window; slot numbering inside a shell buffer is fixed (pitch W).
#include "asora_internal.hpp"
#include "rates_device.hpp"

#include <algorithm>
#include <cmath>

namespace asora {

constexpr double S_STAR = 1e48;                     // photon-number normalisation, f90:28 / photorates.f90

// dist2 exactly as the reference forms it (f90:452-456): no fused multiply-adds
__device__ __forceinline__ double dist2_reference(int a, int b, int c, double dr)
{
    const double xs = mul_unfused(dr, (double)a), ys = mul_unfused(dr, (double)b), zs = mul_unfused(dr, (double)c);
    return add_unfused(add_unfused(mul_unfused(xs, xs), mul_unfused(ys, ys)), mul_unfused(zs, zs));
}

// SB_THREADS = 256, or 1024 when there are fewer workgroups than CUs (a handful of sources: the launch then lasts
// as long as one workgroup, so wider workgroups shorten it)
// HEAT: heating tables are looked up and heating rates deposited.  LDS_SHELLS: the two shell buffers (3 W^2 doubles each)
// live in LDS for the launch (they fit up to W ~ 55) and only the trailing shell goes through the global buffer that
// hands it to the next sub-box's launch; otherwise both stay in the global, L2-resident buffer.
template <int SB_THREADS, bool HEAT, bool LDS_SHELLS>
__global__ void __launch_bounds__(SB_THREADS) subbox_sweep_kernel(const SubboxParams p)
{
    __shared__ double2 logtab[LOG_TABLE_SIZE];
    __shared__ double red[SB_THREADS / 64];
    extern __shared__ double lds_shells[];

    const int blk = blockIdx.x;
    const int src_local = (blk & 7) + 8 * (blk >> 6);       // the 8 octants of a source share an XCD
    const int oct = (blk >> 3) & 7;
    if (src_local >= p.src_count) return;
    if (!p.active[src_local]) return;
    const int ns = p.src_begin + src_local;

    const int N = p.N, W = p.W, WW = W * W;
    const int i0 = p.src_pos[3 * ns + 0], j0 = p.src_pos[3 * ns + 1], k0 = p.src_pos[3 * ns + 2];
    const double flux = p.src_flux[p.flux_src >= 0 ? p.flux_src : ns];           // f90:500,503
    const int sa = (oct & 1) ? -1 : 1, sb = (oct & 2) ? -1 : 1, sc = (oct & 4) ? -1 : 1;
    const int Ea = sa > 0 ? p.ext_r : p.ext_l, Eb = sb > 0 ? p.ext_r : p.ext_l, Ec = sc > 0 ? p.ext_r : p.ext_l;
    // box faces of this sub-box on the sides this octant looks at (f90:199-200,541)
    const int fa = sa > 0 ? p.edge_r : p.edge_l, fb = sb > 0 ? p.edge_r : p.edge_l, fc = sc > 0 ? p.edge_r : p.edge_l;
    const double sig = p.sig, dr = p.dr;
    const double R2 = p.R * p.R;
    const double maxcd = (double)2e30f;                                  // f90:368 (single precision, promoted)
    const double limit = (double)1.0e-7f;                                // photorates.f90:69
    const double r3 = (double)1.7320507764816284, r2 = (double)1.4142135381698608;   // f90:608-609
    const bool dump = p.dump != nullptr && ns == p.dump_src;

    for (int t = threadIdx.x; t < LOG_TABLE_SIZE; t += SB_THREADS) logtab[t] = p.logtab[t];

    double *buf0 = p.scratch + (size_t)blk * p.unit_stride;
    double *buf1 = buf0 + 3 * (size_t)WW;
    double *gprev = (p.s_begin & 1) ? buf1 : buf0;          // after k shells the trailing one sits in buffer k & 1
    double *prev = gprev;
    double *cur = (p.s_begin & 1) ? buf0 : buf1;
    if (LDS_SHELLS) {
        prev = lds_shells; cur = lds_shells + 3 * WW;
        if (p.s_begin > 0)                                  // the trailing shell of the previous sub-box
            for (int t = threadIdx.x; t < 3 * WW; t += SB_THREADS) prev[t] = gprev[t];
    }
    __syncthreads();

    double loss = 0.0;

    // One cell, given its incoming column density: outgoing column density, rates, loss (evolve0D, f90:484-543).
    // idx addresses nHI and the rate grids ([i][j][k], or the [k][j][i] copies for dk = s cells), idx_plain the
    // column-density output grid.
    auto deposit = [&](double cd_in, double path, double vol, bool stop, bool owner, bool on_edge, unsigned idx,
                       unsigned idx_plain) -> double {
        const double nHI = p.nhi[idx];
        const double cd_out = cd_in + nHI * path;                                     // f90:488
        if (!owner) return cd_out;
        if (dump) p.dump[idx_plain] = cd_out;
        double phi_out = 0.0;     // a cell that deposits nothing leaves the reference's phi_out undefined: 0 here
        if (!stop) {
            // (un-fused products: tau_out - tau_in is the difference of the two ROUNDED optical depths, as in the reference)
            const double tau_in = mul_unfused(cd_in, sig), tau_out = mul_unfused(cd_out, sig);
            const double dtau = tau_out - tau_in;
            const bool thick = fabs(dtau) > limit;
            double phi, heat = 0.0;
            if (p.grey) {                                                             // photorates.f90:13-57
                const double pref = flux * S_STAR / vol;
                const double phi_in = pref * exp(-tau_in);
                if (thick) { const double eo = exp(-tau_out); phi_out = pref * eo; phi = pref * (exp(-tau_in) - eo); }
                else       { phi = pref * dtau * exp(-tau_in); phi_out = phi_in - phi; }
            } else {                                                                  // photorates.f90:62-125
                const double pref = flux / vol;
                const Lookup A = lookup_issue<HEAT>(p.tables, tau_in, p, logtab);     // thick table at tau_in
                const Lookup B = thick ? lookup_issue<HEAT>(p.tables, tau_out, p, logtab)
                                       : lookup_issue<HEAT>(p.tables, tau_in, p, logtab, table_stride(p.table_len));   // thin, tau_in
                const double phi_in = pref * lookup_value(A);
                if (thick) {
                    // (pref*(T_in - T_out): phi_in - phi_out would be fused into fma(pref, T_in, -phi_out) and leave the
                    //  rounding error of phi_out where the two table values are equal, e.g. beyond the last entry)
                    const double tb = lookup_value(B);
                    phi_out = pref * tb;
                    phi = pref * (lookup_value(A) - tb);
                    if (HEAT) heat = pref * (lookup_heat(A) - lookup_heat(B));
                } else {
                    phi = pref * dtau * lookup_value(B);
                    phi_out = phi_in - phi;
                    if (HEAT) heat = pref * dtau * lookup_heat(B);
                }
            }
            // (a rate of exactly +0 -- both lookups beyond the last table entry -- changes nothing: it is not added
            //  with ASORA_OPT_SKIP_ZERO_RATES; NaN != 0, so the reference's NaN for nHI = 0 is still deposited)
            const double dphi = phi / nHI;                                            // f90:531-535
            if (p.add_zero || dphi != 0.0) unsafeAtomicAdd(p.phi + idx, dphi);
            if (HEAT) { const double dheat = heat / nHI; if (p.add_zero || dheat != 0.0) unsafeAtomicAdd(p.heat_grid + idx, dheat); }
        }
        if (on_edge) loss += phi_out;                                                 // f90:541-543
        return cd_out;
    };

    // ---- shell 0: the source cell (f90:430-439) ---------------------------------------------
    if (p.s_begin == 0) {
        if (threadIdx.x == 0) {
            const unsigned idx = ((unsigned)i0 * N + j0) * N + k0;
            prev[0] = deposit(0.0, 0.5 * dr, dr * dr * dr, false, oct == 0, false, idx, idx);
        }
        __syncthreads();
    }

    for (int s = p.s_begin + 1; s <= p.s_end; ++s) {
        const double sd = (double)s;
        const double alam = (sd - 0.5) / sd;                 // f90:612 in source-relative form
        // transverse offsets beyond sqrt(R^2 - s^2) lie beyond the radius (see `beyond` below): unless this source's column
        // densities go back to the caller, the faces are only walked up to there (+1: the exact test is `beyond`'s)
        const double q2 = R2 * (1.0 + 1e-9) + 1e-9 - sd * sd;
        const int qlim = dump ? N : (q2 < 0.0 ? -1 : (int)sqrt(q2) + 1);
        const int Az = min(min(s, Ea), qlim), Bz = min(min(s, Eb), qlim);
        const int Ay = min(min(s, Ea), qlim), Cy = min(min(s - 1, Ec), qlim);
        const int Bx = min(min(s - 1, Eb), qlim), Cx = min(min(s - 1, Ec), qlim);
        const int nz = (s <= Ec && qlim >= 0) ? (Az + 1) * (Bz + 1) : 0;
        const int ny = (s <= Eb && qlim >= 0) ? (Ay + 1) * (Cy + 1) : 0;
        const int nx = (s <= Ea && qlim >= 0) ? (Bx + 1) * (Cx + 1) : 0;
        const int ntot = nz + ny + nx;
        const int sm = s - 1;
        // row = t / rowlength without an integer division: (t + 1/2) * (1/rowlength) in single precision is off by
        // < 2e-5 for t < 2^17 and rows < 2^8 cells, while the quotient stays >= 1/(2 rowlength) away from an integer
        const float inv_z = 1.0f / (float)(Az + 1), inv_y = 1.0f / (float)(Cy + 1), inv_x = 1.0f / (float)(Cx + 1);

        for (int t = threadIdx.x; t < ntot; t += SB_THREADS) {
            int a, b, c, U, V, face;
            if (t < nz) {            // dk = s: rows along a (contiguous in the [k][j][i] copies)
                const int row = (int)(((float)t + 0.5f) * inv_z);
                a = t - row * (Az + 1); b = row; c = s;
                U = a; V = b; face = 2;
            } else if (t < nz + ny) { // dj = s: rows along c
                const int r = t - nz, row = (int)(((float)r + 0.5f) * inv_y);
                c = r - row * (Cy + 1); a = row; b = s;
                U = a; V = c; face = 1;
            } else {                  // di = s
                const int r = t - nz - ny, row = (int)(((float)r + 0.5f) * inv_x);
                c = r - row * (Cx + 1); b = row; a = s;
                U = b; V = c; face = 0;
            }

            // A cell beyond R_max_LLS gets no rate (f90:474-478) and adds nothing to the photon loss; its column density
            // is only ever read by cells further out (every upstream corner has coordinates <= its own, and the rounded
            // distance is monotone in each of them), i.e. by cells beyond the radius as well.  So unless this source's
            // column densities are what the caller gets back (the last source of the call), it is not evaluated: for a
            // box of +-R that is half of the cube.
            const double dist2 = dist2_reference(a, b, c, dr);
            const bool beyond = dist2 / (dr * dr) > R2;                                         // f90:474
            if (beyond && !dump) continue;

            // ---- cinterp, f90:576-815, in source-relative octant coordinates -----------------
            const double u = (double)U, v = (double)V;
            // corners that would step across a zero offset, or keep a transverse offset equal to s, carry
            // weight 0 (up to rounding) and do not exist in shell s-1: not fetched
            const bool em = U >= 1, e0 = U <= sm, fm = V >= 1, f0 = V <= sm;
            // A transverse offset equal to s (cube edges, the diagonal): alam * s - (s - 1/2) is 0 in exact arithmetic but a few
            // 1e-17 s in floating point for most s, which the reference multiplies with the column density of a REAL neighbour
            // (its traversal holds the whole cube) -- a 1e-16 effect.  The corner not being fetched here, its value is 0 and its
            // weight 1 / max(0.6, 0) instead of 1 / (c sigma): in a medium with tau ~ 200 per cell that amplified the speck by
            // c sigma / 0.6 ~ 600 s, 5e-9 of the column density at the corner of a +-64 cube (round 6, found by the full-size
            // column-density fixture).  Its weight is therefore set to the exact 0.
            const double de = e0 ? 2.0 * fabs(alam * u - (u - 0.5)) : 0.0;
            const double df = f0 ? 2.0 * fabs(alam * v - (v - 0.5)) : 0.0;
            double w1 = (1. - de) * (1. - df);
            double w2 = (1. - df) * de;
            double w3 = (1. - de) * df;
            double w4 = de * df;
            int o1, o2, o3, o4;   // slots of (U-1,V-1) (U,V-1) (U-1,V) (U,V) in shell s-1
            if (face == 2) {
                o1 = (V - 1) * W + (U - 1); o2 = (V - 1) * W + U; o3 = V * W + (U - 1); o4 = V * W + U;
            } else if (face == 1) {
                // neighbour (a', sm, c'): on the z-face when c' == sm (ties go to z), else on the y-face
                const bool zm = (V - 1 == sm), z0 = (V == sm);
                const int ym = zm ? sm * W : WW + (V - 1);
                const int y0 = z0 ? sm * W : WW + V;
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
            // w_n = s_n / max(0.6, c_n sig) (weightf, f90:823-835) and cdensi = sum(c_n w_n) / sum(w_n) (f90:641), with
            // numerator and denominator multiplied through by the four max() terms: one division instead of five
            const double m1 = fmax(0.6, c1 * sig), m2 = fmax(0.6, c2 * sig), m3 = fmax(0.6, c3 * sig), m4 = fmax(0.6, c4 * sig);
            const double m12 = m1 * m2, m34 = m3 * m4;
            w1 *= m2 * m34; w2 *= m1 * m34; w3 *= m12 * m4; w4 *= m12 * m3;
            double cd_in = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
            if (s == 1 && (U == 1 || V == 1)) cd_in = ((U == 1 && V == 1) ? r3 : r2) * cd_in;   // f90:648-658
            const double path = sqrt((u * u + v * v) / (sd * sd) + 1.0) * dr;                   // f90:661 * dr

            // ---- the cell itself, f90:441-543 --------------------------------------------------
            const int i = wrap_once(i0 + sa * a, N), j = wrap_once(j0 + sb * b, N), k = wrap_once(k0 + sc * c, N);
            const unsigned idx_plain = ((unsigned)i * N + j) * N + k;
            const unsigned idx = face == 2 ? ((unsigned)k * N + j) * N + i + p.ncell : idx_plain;
            const double vol = dist2 * path * FOURPI;                                          // f90:457
            const bool stop = beyond || cd_in > maxcd;                                          // f90:474-478
            // a cell on an octant-boundary plane is the business of the octant with the + sign there
            const bool owner = (a > 0 || sa > 0) && (b > 0 || sb > 0) && (c > 0 || sc > 0);
            const bool on_edge = a == fa || b == fb || c == fc;
            const int own = face == 2 ? b * W + a : (face == 1 ? WW + a * W + c : 2 * WW + b * W + c);
            cur[own] = deposit(cd_in, path, vol, stop, owner, on_edge, idx, idx_plain);
        }
        __syncthreads();
        double *tmp = prev; prev = cur; cur = tmp;
    }

    if (LDS_SHELLS) {       // hand the trailing shell to the next sub-box's launch: after k shells it sits in buffer k & 1
        double *gnext = (p.s_end & 1) ? buf1 : buf0;
        for (int t = threadIdx.x; t < 3 * WW; t += SB_THREADS) gnext[t] = prev[t];
    }

    // ---- photon loss of this octant through the faces of the sub-box ------------------------------
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int w = 0; w < SB_THREADS / 64; ++w) total += red[w];
        if (total != 0.0) unsafeAtomicAdd(p.loss + src_local, total * (dr * dr * dr));
    }
}

// Per source, after sub-box n: book the box, keep its loss, decide whether a further box is traced
// (the while condition of do_source, f90:193-195).  mode 0 initialises (loss = all photons, f90:188).
__global__ void subbox_decide_kernel(int mode, int count, const double *src_flux, int src_begin, double loss_fraction,
                                     int more_range, int *active, double *loss, double *loss_final, int *nbox,
                                     int *n_active)
{
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < count; s += gridDim.x * blockDim.x) {
        const double all = src_flux[src_begin + s] * S_STAR;
        if (mode == 0) { loss[s] = all; loss_final[s] = all; nbox[s] = 0; active[s] = 1; }
        else if (active[s]) { nbox[s] += 1; loss_final[s] = loss[s]; }
        if (active[s]) {
            const bool go = more_range && loss[s] > loss_fraction * src_flux[src_begin + s] * S_STAR;
            active[s] = go ? 1 : 0;
            if (go) { loss[s] = 0.0; atomicAdd(n_active, 1); }
        }
    }
}

// threads per workgroup of the sweep when there are enough sources to fill the chip (A/B: tools/ab_oneoff.sh)
#ifndef ASORA_SB_THREADS
#define ASORA_SB_THREADS 256
#endif

template <int T>
static int launch_subbox_variant(State &st, const SubboxParams &p, unsigned grid, bool lds, size_t lds_bytes, hipStream_t stream)
{
#define ASORA_SB_LAUNCH(HT, LD)                                                                                          \
    do {                                                                                                                 \
        if (LD) ASORA_HIP_TRY(hipFuncSetAttribute((const void *)subbox_sweep_kernel<T, HT, LD>,                          \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));          \
        hipLaunchKernelGGL((subbox_sweep_kernel<T, HT, LD>), dim3(grid), dim3(T), LD ? lds_bytes : 0, stream, p);        \
    } while (0)
    if (p.heat) { if (lds) ASORA_SB_LAUNCH(true, true); else ASORA_SB_LAUNCH(true, false); }
    else        { if (lds) ASORA_SB_LAUNCH(false, true); else ASORA_SB_LAUNCH(false, false); }
#undef ASORA_SB_LAUNCH
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_subbox_sweep(State &st, const SubboxParams &p, hipStream_t side)
{
    const unsigned grid = 64u * (unsigned)((p.src_count + 7) / 8);
    hipStream_t stream = side ? side : st.stream;
    KernelTimer kt(ASORA_KERNEL_RAYTRACE, stream);
    // both shell buffers in LDS up to 56 KB per workgroup (W <= 34: a +-32 box, the benchmark's)
    const size_t lds_bytes = (size_t)6 * p.W * p.W * sizeof(double);
    const bool lds = lds_bytes <= 56 * 1024 && !st.opt[ASORA_OPT_SUBBOX_GLOBAL_SHELLS];
    if ((long)p.src_count * 8 < (long)st.cu_count) return launch_subbox_variant<1024>(st, p, grid, lds, lds_bytes, stream);
    // small boxes: shells of a few hundred cells fill 128 threads better (+-16: 0.65 -> 0.59 ms per 1000 sources; +-32: 256)
    if (ASORA_SB_THREADS == 256 && p.W <= 20) return launch_subbox_variant<128>(st, p, grid, lds, lds_bytes, stream);
    return launch_subbox_variant<ASORA_SB_THREADS>(st, p, grid, lds, lds_bytes, stream);
}

int launch_subbox_decide(State &st, int mode, int count, const double *src_flux, int src_begin, double loss_fraction,
                         int more_range, int *active, double *loss, double *loss_final, int *nbox, int *n_active)
{
    ASORA_HIP_TRY(hipMemsetAsync(n_active, 0, sizeof(int), st.stream));
    hipLaunchKernelGGL(subbox_decide_kernel, dim3(std::max(1, std::min(1024, (count + 255) / 256))), dim3(256), 0,
                       st.stream, mode, count, src_flux, src_begin, loss_fraction, more_range, active, loss,
                       loss_final, nbox, n_active);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace asora
