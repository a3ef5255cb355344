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
//   * the live shell is double-buffered in LDS; the reference's NUM_SRC_PAR x N^3 global scratch
//     (memory.cu:65) does not exist.
//   * only cells that can receive a rate are evaluated: |d|^2 <= R^2, inside the periodic window
//     and inside the reference's octahedron q <= q_max.  Every upstream neighbour of such a cell
//     is strictly closer to the source, so the pruned cells never feed a kept one and phi_ion is
//     unchanged (the reference evaluates them into scratch and then discards them,
//     raytracing.cu:311-315).
//   * everything about a cell that does not depend on the source or on the medium -- which cells
//     a shell holds, their bilinear interpolation weights, their path length, |d|^2 and the
//     shell-buffer slots of their four upstream corners -- is the same for all sources.  It is
//     tabulated ONCE per (N, R) on the host (build_octant_geometry below, with the reference's
//     own expressions) and streamed from L2 by every workgroup; the kernel does the
//     medium-dependent arithmetic only.
//   * faces dj=s and di=s are rows along k, contiguous in the [i][j][k] grid.  Faces dk=s are
//     rows along i, so they read nHI and accumulate Gamma through [k][j][i] transposed copies
//     (rows contiguous again); the transposed accumulator is folded back once per call.
//   * nHI = ndens*(1-xh_av) is formed once per call (raytracing.cu:275-276 forms it per visit).
#include "asora_internal.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace asora {

constexpr int RT_THREADS = 256;
constexpr double FOURPI = 12.566370614359172463991853874177;   // raytracing.cu:12
constexpr int LOG_TABLE_BITS = 7;
constexpr int LOG_TABLE_SIZE = 1 << LOG_TABLE_BITS;

// ---------------------------------------------------------------------------------------------
// Rates (src/asora/rates.cu)
// ---------------------------------------------------------------------------------------------

// log2 of a positive normal double: exponent + table (2^7 intervals of the mantissa: 1/c and
// log2 c at the interval centres, staged in LDS) + degree-6 series in r = m/c - 1, |r| < 2^-8
// (truncation 3e-18).  Absolute error ~1 ulp of the result, like libm's log10; it replaces
// log10 in the table lookup because two of them per cell dominated the instruction count.
__device__ __forceinline__ double log2_pos(double x, const double2 *__restrict__ logtab)
{
    const long long bits = __double_as_longlong(x);
    const int e = (int)(bits >> 52) - 1023;
    const int idx = (int)(bits >> (52 - LOG_TABLE_BITS)) & (LOG_TABLE_SIZE - 1);
    const double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    const double2 t = logtab[idx];                 // {1/c, log2 c}
    const double r = fma(m, t.x, -1.0);
    // log2(1+r) = r/ln2 * (1 - r/2 + r^2/3 - r^3/4 + r^4/5 - r^5/6)
    const double C1 = 1.4426950408889634074, C2 = -0.72134752044448170368, C3 = 0.48089834696298780245,
                 C4 = -0.36067376022224085184, C5 = 0.28853900817779268147, C6 = -0.24044917348149390123;
    const double p = r * fma(r, fma(r, fma(r, fma(r, fma(r, C6, C5), C4), C3), C2), C1);
    return (double)e + (t.y + p);
}

// photo_lookuptable, rates.cu:70-83 (== photorates.f90:130-147).  The reference forms
// 1 + (log10(tau) - minlogtau)/dlogtau; here that is one fused multiply-add on log2(tau) with
// k1 = log10(2)/dlogtau, k0 = 1 - minlogtau/dlogtau.  Indices are clamped to the last table
// element (the reference reads one past the end when NumTau == len(table), tau >= 10^maxlogtau).
__device__ __forceinline__ double table_lookup(const double *__restrict__ table, double tau, const RtParams &p,
                                               const double2 *__restrict__ logtab)
{
    const double l2 = log2_pos(fmax(1.0e-20, tau), logtab);
    const double real_i = fmin(p.numtau_f, fmax(0.0, fma(l2, p.lut_k1, p.lut_k0)));
    int i0 = (int)real_i;
    int i1 = min(p.NumTau, i0 + 1);
    const double residual = real_i - (double)i0;
    const int last = p.table_len - 1;
    i0 = min(i0, last);
    i1 = min(i1, last);
    const double t0 = table[i0];
    const double t1 = table[i1];
    return fma(residual, t1 - t0, t0);
}

// photoion_rates_gpu rates.cu:16-41 / photoion_rates_test_gpu rates.cu:48-64, divided by nHI
// (raytracing.cu:324): pref = flux/(vol*nHI) replaces the reference's two divisions by one.
__device__ __forceinline__ double photo_rate_per_atom(double flux, double cd_in, double cd_out, double vol_nhi,
                                                      const RtParams &p, const double2 *__restrict__ logtab)
{
    const double tau_in = cd_in * p.sig;
    const double tau_out = cd_out * p.sig;
    // TAU_PHOTO_LIMIT: rates.cu:7 (double 1e-7) or photorates.f90:69 (single 1e-7 promoted)
    const double limit = p.fortran_consts ? (double)1.0e-7f : 1.0e-7;
    if (p.grey) {
        const double pref = flux * 1e48 / vol_nhi;
        if (fabs(tau_out - tau_in) > limit) return pref * (exp(-tau_in) - exp(-tau_out));
        return pref * (tau_out - tau_in) * exp(-tau_in);
    }
    const double pref = flux / vol_nhi;
    if (fabs(tau_out - tau_in) > limit) {
        const double t_in = table_lookup(p.thick, tau_in, p, logtab);
        const double t_out = table_lookup(p.thick, tau_out, p, logtab);
        return pref * t_in - pref * t_out;
    }
    // rates.cu:37 uses tau_out, photorates.f90:121 uses tau_in
    return pref * (tau_out - tau_in) * table_lookup(p.thin, p.fortran_consts ? tau_in : tau_out, p, logtab);
}

__device__ __forceinline__ int wrap_once(int x, int N)
{
    return x < 0 ? x + N : (x >= N ? x - N : x);
}

// ---------------------------------------------------------------------------------------------
// The octant kernel
// ---------------------------------------------------------------------------------------------
// Dynamic LDS: [shell buffer 0: max_cells+1 doubles][shell buffer 1: same]   (unless GLOBAL_SCRATCH)
//              [log table: 128 x {1/c, log2 c}][wrapped i(a), j(b), k(c): 3*(S+1) ints]
// Slot max_cells of each shell buffer holds 0.0: upstream corners of weight 0 point there.
template <bool GLOBAL_SCRATCH, bool DUMP>
__global__ void __launch_bounds__(RT_THREADS) raytrace_octant_kernel(const RtParams p)
{
    extern __shared__ double lds_raw[];

    const int blk = blockIdx.x;
    // blocks b and b+8 share an XCD (round-robin dispatch): keep the 8 octants of one source
    // on one XCD so that they share its L2 lines of nHI.  Speed only, never correctness.
    const int src_local = (blk & 7) + 8 * (blk >> 6);
    const int oct = (blk >> 3) & 7;
    if (src_local >= p.src_count) return;
    const int ns = p.src_begin + src_local;

    const OctGeomDev G = p.geom[oct];
    const int N = p.N;
    const int i0 = p.src_pos[3 * ns + 0];
    const int j0 = p.src_pos[3 * ns + 1];
    const int k0 = p.src_pos[3 * ns + 2];
    const double flux = p.src_flux[ns];
    const int sa = (oct & 1) ? -1 : 1, sb = (oct & 2) ? -1 : 1, sc = (oct & 4) ? -1 : 1;

    const int slots = p.max_cells + 1;
    double *prev, *cur, *after;
    if (GLOBAL_SCRATCH) {
        prev = p.shell_scratch + (size_t)blk * 2 * slots;
        cur = prev + slots;
        after = lds_raw;
    } else {
        prev = lds_raw;
        cur = prev + slots;
        after = cur + slots;
    }
    double2 *logtab = reinterpret_cast<double2 *>(after);
    int *wi = reinterpret_cast<int *>(logtab + LOG_TABLE_SIZE);
    int *wj = wi + (p.S + 1);
    int *wk = wj + (p.S + 1);

    for (int t = threadIdx.x; t < LOG_TABLE_SIZE; t += RT_THREADS) logtab[t] = p.logtab[t];
    for (int t = threadIdx.x; t <= p.S; t += RT_THREADS) {
        wi[t] = wrap_once(i0 + sa * t, N);      // periodic position of offset t along each axis
        wj[t] = wrap_once(j0 + sb * t, N);      // (|offset| <= N/2: one wrap suffices, raytracing.cu:270-272)
        wk[t] = wrap_once(k0 + sc * t, N);
    }
    if (threadIdx.x == 0) { prev[p.max_cells] = 0.0; cur[p.max_cells] = 0.0; }
    __syncthreads();

    const double sig = p.sig, dr = p.dr, dr2 = dr * dr;
    const double maxcd = p.fortran_consts ? (double)2e30f : 2e30;                    // raytracing.cu:15
    const double r3 = p.fortran_consts ? (double)1.7320507764816284 : 1.73205080757; // f90:608 / cu:435
    const double r2 = p.fortran_consts ? (double)1.4142135381698608 : 1.41421356237; // f90:609 / cu:439
    const unsigned negmask = (sa < 0 ? 1u : 0u) | (sb < 0 ? 2u : 0u) | (sc < 0 ? 4u : 0u);

    unsigned int n_gamma = 0, n_eval = 0;

    // ---- shell 0: the source cell (raytracing.cu:285-294) -----------------------------------
    if (threadIdx.x == 0) {
        const unsigned idx = ((unsigned)i0 * N + j0) * N + k0;
        const double nHI = p.nhi[idx];
        const double path = 0.5 * dr;
        const double cd_out = 0.0 + nHI * path;
        prev[0] = cd_out;
        ++n_eval;
        if (oct == 0) {
            if (DUMP) p.dump[idx] = cd_out;
            const double phi = photo_rate_per_atom(flux, 0.0, cd_out, dr * dr * dr * nHI, p, logtab);
            unsafeAtomicAdd(&p.phi[idx], phi);
            ++n_gamma;
        }
    }
    __syncthreads();

    for (int s = 1; s <= G.S; ++s) {
        const unsigned off = G.shell_off[s];
        const int ncell = (int)(G.shell_off[s + 1] - off);
        if (ncell == 0) break;                               // uniform: nothing further out either

        for (int t = threadIdx.x; t < ncell; t += RT_THREADS) {
            const unsigned g = off + t;
            const unsigned abc = G.abc[g];
            const int a = abc & 1023, b = (abc >> 10) & 1023, c = (abc >> 20) & 1023;
            const unsigned face = abc >> 30;                 // 2: dk = s, 1: dj = s, 0: di = s
            const uint4 nb = G.nbr[g];

            // ---- cinterp_gpu, raytracing.cu:345-535 ------------------------------------------
            // w_n = s_n / max(0.6, c_n*sig) (raytracing.cu:33,422-425) and
            // cdensi = sum(c_n w_n)/sum(w_n) (raytracing.cu:428), with numerator and denominator
            // multiplied through by the four max() terms: one division instead of five.
            const double c1 = prev[nb.x], c2 = prev[nb.y], c3 = prev[nb.z], c4 = prev[nb.w];
            const double m1 = fmax(0.6, c1 * sig), m2 = fmax(0.6, c2 * sig);
            const double m3 = fmax(0.6, c3 * sig), m4 = fmax(0.6, c4 * sig);
            const double m12 = m1 * m2, m34 = m3 * m4;
            const double q1 = G.w1[g] * (m2 * m34), q2 = G.w2[g] * (m1 * m34);
            const double q3 = G.w3[g] * (m12 * m4), q4 = G.w4[g] * (m12 * m3);
            double cd_in = (c1 * q1 + c2 * q2 + c3 * q3 + c4 * q4) / (q1 + q2 + q3 + q4);
            if (s == 1) {                                    // diagonal neighbours of the source, cu:431-441
                const int nz = (a == 0) + (b == 0) + (c == 0);
                if (nz < 2) cd_in = (nz == 0 ? r3 : r2) * cd_in;
            }
            const double path = G.path[g] * dr;

            // ---- the cell itself, raytracing.cu:270-276,311-328 -----------------------------
            const unsigned i = wi[a], j = wj[b], k = wk[c];
            const bool zt = p.z_transposed && face == 2;
            const unsigned idx = zt ? (k * N + j) * N + i : (i * N + j) * N + k;
            const double nHI = (zt ? p.nhi_t : p.nhi)[idx];
            const double cd_out = fma(nHI, path, cd_in);
            cur[t] = cd_out;
            ++n_eval;
            // a cell on an octant-boundary plane is rated by the octant with the + sign there
            const unsigned zmask = (a == 0 ? 1u : 0u) | (b == 0 ? 2u : 0u) | (c == 0 ? 4u : 0u);
            if ((zmask & negmask) == 0) {
                if (DUMP) p.dump[(i * N + j) * N + k] = cd_out;
                if (cd_in <= maxcd) {
                    const double vol_nhi = G.n2[g] * dr2 * path * FOURPI * nHI;          // raytracing.cu:302-307
                    const double phi = photo_rate_per_atom(flux, cd_in, cd_out, vol_nhi, p, logtab);
                    unsafeAtomicAdd((zt ? p.phi_t : p.phi) + idx, phi);
                    ++n_gamma;
                }
            }
        }
        __syncthreads();
        double *tmp = prev; prev = cur; cur = tmp;
    }

    // work accounting: one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        n_gamma += __shfl_down(n_gamma, o);
        n_eval += __shfl_down(n_eval, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&p.counters[0], (unsigned long long)n_gamma);
        atomicAdd(&p.counters[1], (unsigned long long)n_eval);
    }
}

// ---------------------------------------------------------------------------------------------
// Source-independent geometry of one octant (host)
// ---------------------------------------------------------------------------------------------
// Faces of shell s follow the reference's branch order z, y, x (raytracing.cu:394,446,491; ties go
// to z, then y):  z-face dk = s: (a,b,s), a,b <= s;  y-face dj = s: (a,s,c), c < s;  x-face di = s:
// (s,b,c), b,c < s.  Inside a face the fastest index is the one that is contiguous in memory
// (a for the transposed z-face, c otherwise).  The weights are the reference's expressions
// (raytracing.cu:397-408,444) in source-relative coordinates |d|; corners that would step across
// a zero offset or keep a transverse offset equal to s get weight exactly 0 from those
// expressions and are given the buffer's zero slot.
namespace {

struct HostGeom {
    std::vector<uint32_t> shell_off, abc;
    std::vector<double> w1, w2, w3, w4, path, n2;
    std::vector<uint4> nbr;
    int S = 0;
    uint32_t max_cells = 1;
};

inline bool inside_radius_reference(int a, int b, int c, double dr, double R2)
{
    // raytracing.cu:302-305,315 as the reference evaluates it (un-fused on the host)
    volatile double xs = dr * (double)a, ys = dr * (double)b, zs = dr * (double)c;
    volatile double xx = xs * xs, yy = ys * ys, zz = zs * zs;
    volatile double d2 = xx + yy;
    d2 = d2 + zz;
    volatile double den = dr * dr;
    return d2 / den <= R2;
}

void build_octant_geometry(HostGeom &h, int Ea, int Eb, int Ec, double R, double dr, int q_max, uint32_t zero_slot_marker)
{
    const double R2 = R * R;
    const double R2hi = R2 * (1.0 + 1e-9) + 1e-9;
    const int Emax = std::max(Ea, std::max(Eb, Ec));
    int S = Emax;
    if (std::isfinite(R2hi)) S = (int)std::min((double)Emax, std::floor(std::sqrt(R2hi)));
    h.S = S;
    h.shell_off.assign(S + 2, 0);
    h.shell_off[0] = 0;
    h.shell_off[1] = 0;           // shell 0 (the source cell) is handled by the kernel prologue
    // slot maps of the previous / current shell: face*(P*P) + u*P + v with P = S+1
    const size_t P = (size_t)S + 1;
    std::vector<uint32_t> slot_prev(3 * P * P, zero_slot_marker), slot_cur(3 * P * P, zero_slot_marker);
    slot_prev[0] = 0;             // (0,0,0) sits in slot 0 (z-face entry a=0,b=0 of shell 0)
    auto slot_of = [&](const std::vector<uint32_t> &m, int a, int b, int c, int t) -> uint32_t {
        // cell (a,b,c) of shell t = max(a,b,c)
        if (c == t) return m[0 * P * P + (size_t)b * P + a];
        if (b == t) return m[1 * P * P + (size_t)a * P + c];
        return m[2 * P * P + (size_t)b * P + c];
    };
    auto in_sphere = [&](int a, int b, int c) -> bool {
        if (a + b + c > q_max) return false;                                   // raytracing.cu:101,198
        const double n2 = (double)a * a + (double)b * b + (double)c * c;
        if (n2 > R2hi) return false;
        if (n2 < R2 * (1.0 - 1e-9) - 1e-9) return true;
        return inside_radius_reference(a, b, c, dr, R2);
    };
    for (int s = 1; s <= S; ++s) {
        std::fill(slot_cur.begin(), slot_cur.end(), zero_slot_marker);
        const double sd = (double)s;
        const double alam = (sd - 0.5) / sd;                                   // raytracing.cu:397
        uint32_t count = 0;
        auto emit = [&](int a, int b, int c, int face, int U, int V) {
            if (!in_sphere(a, b, c)) return;
            const double u = (double)U, v = (double)V;
            const double de = 2.0 * std::fabs(alam * u - (u - 0.5));           // raytracing.cu:399-403
            const double df = 2.0 * std::fabs(alam * v - (v - 0.5));
            double s1 = (1. - de) * (1. - df), s2 = (1. - df) * de, s3 = (1. - de) * df, s4 = de * df;
            const bool em = U >= 1, e0 = U <= s - 1, fm = V >= 1, f0 = V <= s - 1;
            auto corner = [&](int uu, int vv) -> uint32_t {
                int aa, bb, cc;
                if (face == 2) { aa = uu; bb = vv; cc = s - 1; }
                else if (face == 1) { aa = uu; bb = s - 1; cc = vv; }
                else { aa = s - 1; bb = uu; cc = vv; }
                return slot_of(slot_prev, aa, bb, cc, s - 1);
            };
            uint4 nb;
            nb.x = (em && fm) ? corner(U - 1, V - 1) : zero_slot_marker;
            nb.y = (e0 && fm) ? corner(U, V - 1) : zero_slot_marker;
            nb.z = (em && f0) ? corner(U - 1, V) : zero_slot_marker;
            nb.w = (e0 && f0) ? corner(U, V) : zero_slot_marker;
            if (nb.x == zero_slot_marker) s1 = 0.0;
            if (nb.y == zero_slot_marker) s2 = 0.0;
            if (nb.z == zero_slot_marker) s3 = 0.0;
            if (nb.w == zero_slot_marker) s4 = 0.0;
            h.abc.push_back((uint32_t)a | ((uint32_t)b << 10) | ((uint32_t)c << 20) | ((uint32_t)face << 30));
            h.w1.push_back(s1); h.w2.push_back(s2); h.w3.push_back(s3); h.w4.push_back(s4);
            h.path.push_back(std::sqrt((u * u + v * v) / (sd * sd) + 1.0));    // raytracing.cu:444
            h.n2.push_back((double)a * a + (double)b * b + (double)c * c);
            h.nbr.push_back(nb);
            size_t key = face == 2 ? (0 * P * P + (size_t)b * P + a)
                       : face == 1 ? (1 * P * P + (size_t)a * P + c) : (2 * P * P + (size_t)b * P + c);
            slot_cur[key] = count++;
        };
        if (s <= Ec)
            for (int b = 0; b <= std::min(s, Eb); ++b)
                for (int a = 0; a <= std::min(s, Ea); ++a) emit(a, b, s, 2, a, b);
        if (s <= Eb)
            for (int a = 0; a <= std::min(s, Ea); ++a)
                for (int c = 0; c <= std::min(s - 1, Ec); ++c) emit(a, s, c, 1, a, c);
        if (s <= Ea)
            for (int b = 0; b <= std::min(s - 1, Eb); ++b)
                for (int c = 0; c <= std::min(s - 1, Ec); ++c) emit(s, b, c, 0, b, c);
        h.shell_off[s + 1] = h.shell_off[s] + count;
        h.max_cells = std::max(h.max_cells, count);
        slot_prev.swap(slot_cur);
    }
}

template <typename T>
int upload(const std::vector<T> &v, const T *&dev_out, std::vector<void *> &owned)
{
    void *d = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    ASORA_HIP_TRY(hipMalloc(&d, bytes));
    owned.push_back(d);
    if (!v.empty()) ASORA_HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    dev_out = static_cast<const T *>(d);
    return 0;
}

} // namespace

void release_geometry(State &st)
{
    for (void *q : st.geom_owned) (void)hipFree(q);
    st.geom_owned.clear();
    st.geom_dev = nullptr;
    st.logtab_dev = nullptr;
    st.geom_valid = false;
}

// Build (or reuse) the geometry tables for this (N, R, dr).  dr only enters through the
// classification of cells sitting exactly on the sphere (see inside_radius_reference).
static int ensure_geometry(State &st, RtParams &p)
{
    const int N = p.N;
    const int q_max = (int)std::ceil(1.73205080757 * std::min(p.R, 1.73205080757 * N / 2.0));   // raytracing.cu:14,101
    const int ext_pos = N / 2 - 1 + (N % 2);                                                      // raytracing.cu:122
    const int ext_neg = N / 2;                                                                    // raytracing.cu:123
    if (st.geom_valid && st.geom_N == N && st.geom_R == p.R && st.geom_dr == p.dr) {
        p.geom = st.geom_dev; p.logtab = st.logtab_dev; p.S = st.geom_S; p.max_cells = st.geom_max_cells;
        return 0;
    }
    release_geometry(st);

    // octants with the same periodic window share one table; when the sphere does not reach the
    // window on any axis all eight are identical
    const double R2hi_all = p.R * p.R * (1.0 + 1e-9) + 1e-9;
    const bool unclipped = std::isfinite(R2hi_all) && std::floor(std::sqrt(R2hi_all)) <= (double)std::min(ext_pos, ext_neg);
    auto ext = [&](int oct, int ax) { return ((oct >> ax) & 1) ? ext_neg : ext_pos; };
    int owner[8];
    for (int oct = 0; oct < 8; ++oct) {
        owner[oct] = oct;
        for (int o2 = 0; o2 < oct; ++o2)
            if (unclipped || (ext(oct, 0) == ext(o2, 0) && ext(oct, 1) == ext(o2, 1) && ext(oct, 2) == ext(o2, 2))) {
                owner[oct] = owner[o2];
                break;
            }
    }
    HostGeom hg[8];
    OctGeomDev od[8];
    int Smax = 0;
    uint32_t max_cells = 1;
    const uint32_t MARK = 0xffffffffu;
    for (int oct = 0; oct < 8; ++oct) {
        if (owner[oct] != oct) continue;
        build_octant_geometry(hg[oct], ext(oct, 0), ext(oct, 1), ext(oct, 2), p.R, p.dr, q_max, MARK);
        Smax = std::max(Smax, hg[oct].S);
        max_cells = std::max(max_cells, hg[oct].max_cells);
    }
    // zero-slot marker -> max_cells (the slot that holds 0.0), then upload
    for (int oct = 0; oct < 8; ++oct) {
        if (owner[oct] != oct) continue;
        HostGeom &h = hg[oct];
        for (auto &nb : h.nbr) {
            if (nb.x == MARK) nb.x = max_cells;
            if (nb.y == MARK) nb.y = max_cells;
            if (nb.z == MARK) nb.z = max_cells;
            if (nb.w == MARK) nb.w = max_cells;
        }
        OctGeomDev d;
        d.S = h.S;
        if (int rc = upload(h.shell_off, d.shell_off, st.geom_owned)) return rc;
        if (int rc = upload(h.abc, d.abc, st.geom_owned)) return rc;
        if (int rc = upload(h.w1, d.w1, st.geom_owned)) return rc;
        if (int rc = upload(h.w2, d.w2, st.geom_owned)) return rc;
        if (int rc = upload(h.w3, d.w3, st.geom_owned)) return rc;
        if (int rc = upload(h.w4, d.w4, st.geom_owned)) return rc;
        if (int rc = upload(h.path, d.path, st.geom_owned)) return rc;
        if (int rc = upload(h.n2, d.n2, st.geom_owned)) return rc;
        if (int rc = upload(h.nbr, d.nbr, st.geom_owned)) return rc;
        od[oct] = d;
    }
    for (int oct = 0; oct < 8; ++oct) od[oct] = od[owner[oct]];
    std::vector<OctGeomDev> odv(od, od + 8);
    const OctGeomDev *gd = nullptr;
    if (int rc = upload(odv, gd, st.geom_owned)) return rc;

    // log2 table: interval centres c = 1 + (i + 1/2)/128, entries {1/c, log2 c}
    std::vector<double2> lt(LOG_TABLE_SIZE);
    for (int i = 0; i < LOG_TABLE_SIZE; ++i) {
        const long double c = 1.0L + ((long double)i + 0.5L) / (long double)LOG_TABLE_SIZE;
        lt[i].x = (double)(1.0L / c);
        lt[i].y = (double)std::log2(c);
    }
    const double2 *ltd = nullptr;
    if (int rc = upload(lt, ltd, st.geom_owned)) return rc;

    st.geom_dev = gd; st.logtab_dev = ltd;
    st.geom_N = N; st.geom_R = p.R; st.geom_dr = p.dr; st.geom_S = Smax; st.geom_max_cells = (int)max_cells;
    st.geom_valid = true;
    p.geom = gd; p.logtab = ltd; p.S = Smax; p.max_cells = (int)max_cells;
    return 0;
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
    if (int rc = ensure_geometry(st, p)) return rc;
    p.lut_k1 = 0.30102999566398119521 / p.dlogtau;      // log10(2)/dlogtau
    p.lut_k0 = 1.0 - p.minlogtau / p.dlogtau;

    const size_t slots = (size_t)p.max_cells + 1;
    const size_t fixed_bytes = LOG_TABLE_SIZE * sizeof(double2) + 3 * (size_t)(p.S + 1) * sizeof(int);
    const size_t shell_bytes = 2 * slots * sizeof(double);
    const bool use_lds = shell_bytes + fixed_bytes <= LDS_LIMIT_BYTES;
    const size_t lds_bytes = (use_lds ? shell_bytes : 0) + fixed_bytes;

    int done = 0;
    while (done < p.src_count) {
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
#define ASORA_LAUNCH(GS, DP)                                                                                   \
    do {                                                                                                       \
        ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<GS, DP>,                        \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));        \
        hipLaunchKernelGGL((raytrace_octant_kernel<GS, DP>), dim3(grid), dim3(RT_THREADS), lds_bytes,          \
                           st.stream, q);                                                                      \
    } while (0)
            if (use_lds) { if (dump) ASORA_LAUNCH(false, true); else ASORA_LAUNCH(false, false); }
            else         { if (dump) ASORA_LAUNCH(true, true);  else ASORA_LAUNCH(true, false); }
#undef ASORA_LAUNCH
            ASORA_HIP_TRY(hipGetLastError());
        }
        done += batch;
    }
    return 0;
}

} // namespace asora
