// raytrace.hip -- short-characteristics raytracing for gfx950 (MI355X).
//
// What is computed is exactly what the reference's GPU path computes
// (src/asora/raytracing.cu:79-339: do_all_sources_gpu + evolve0D_gpu + cinterp_gpu, and
//  src/asora/rates.cu:16-83): for every source, the incoming/outgoing HI column density of
// every cell by short-characteristics interpolation and the photon-conserving rate Gamma,
// summed over sources into phi_ion.
//
// How it is computed is different:
//   * work item = (source, unit), unit = octant, octant sector or pair of mirrored sectors (pick_launch_shape).
//     The 8 sign-octants of a source only share the three
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
//   * what does not depend on the source or on the medium -- which cells a shell holds, their
//     path length and the shell-buffer slots of their four upstream corners -- is the same for
//     all sources.  It is tabulated ONCE per (N, R) on the host (build_unit_geometry in geometry.hip,
//     32 B per cell) and streamed from L2 by every workgroup, two steps ahead of its use.
//   * faces dj=s and di=s are rows along k, contiguous in the [i][j][k] grid.  Faces dk=s are
//     rows along i, so they read nHI and accumulate Gamma through [k][j][i] transposed copies
//     (rows contiguous again); the transposed accumulator is folded back once per call.
//   * nHI = ndens*(1-xh_av) is formed once per call (raytracing.cu:275-276 forms it per visit).
#include "asora_internal.hpp"
#include "rates_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

namespace asora {

// photoion_rates_gpu rates.cu:16-41 divided by nHI (raytracing.cu:324), also in two halves.
// pref = flux/(vol*nHI) replaces the reference's two divisions by one.  A cell is "thick" when
// |tau_out - tau_in| > TAU_PHOTO_LIMIT: Gamma = pref*(T_thick(tau_in) - T_thick(tau_out)); else
// Gamma = pref*(tau_out - tau_in)*T_thin(tau_out) [rates.cu:37; photorates.f90:121 uses tau_in].
struct RateJob {
    double pref, dtau;       // dtau = tau_out - tau_in for thin cells, unused (0) for thick ones
    Lookup A, B;             // thick: T(tau_in), T(tau_out);  thin: T_thin(tau*), unused
    bool thick;
};
__device__ __forceinline__ RateJob rate_issue(double flux, double cd_in, double cd_out, double vol_nhi,
                                              const RtParams &p, const double2 *__restrict__ logtab)
{
    // (un-fused products: tau_out - tau_in must be the difference of the two ROUNDED optical depths, as in the
    //  reference; fused into fma(cd_out, sig, -tau_in) it would carry the rounding error of one product -- a speck
    //  instead of 0 where nHI = 0, and a 1e-8 relative change of dtau for thin cells)
    const double tau_in = mul_unfused(cd_in, p.sig);
    const double tau_out = mul_unfused(cd_out, p.sig);
    // TAU_PHOTO_LIMIT: rates.cu:7 (double 1e-7) or photorates.f90:69 (single 1e-7 promoted)
    const double limit = p.fortran_consts ? (double)1.0e-7f : 1.0e-7;
    RateJob J;
    J.pref = flux / vol_nhi;
    J.thick = fabs(tau_out - tau_in) > limit;
    J.dtau = tau_out - tau_in;
    const double tau_thin = p.fortran_consts ? tau_in : tau_out;
    // one code path for both kinds of cell: per-lane table and arguments
    // thick table at [0, table_len), thin table at [table_len, 2*table_len) of one allocation
    const int toff = J.thick ? 0 : table_stride(p.table_len);
    J.A = lookup_issue(p.tables, J.thick ? tau_in : tau_thin, p, logtab, toff);
    J.B = lookup_issue(p.tables, J.thick ? tau_out : tau_thin, p, logtab, toff);
    return J;
}
__device__ __forceinline__ double rate_value(const RateJob &J)
{
    const double a = lookup_value(J.A), b = lookup_value(J.B);
    // (pref*(a - b), not pref*a - pref*b: the compiler would fuse the latter into fma(pref, a, -(pref*b)), which
    //  leaves the rounding error of pref*b -- a random-signed ~1e-16 relative speck -- where a == b, e.g. beyond the
    //  last table entry, where the reference's un-fused arithmetic gives exactly 0)
    return J.thick ? J.pref * (a - b) : J.pref * J.dtau * a;
}

// photoion_rates_test_gpu rates.cu:48-64 (analytic grey rates, GREY_NOTABLES builds), per atom
__device__ __forceinline__ double grey_rate_per_atom(double flux, double cd_in, double cd_out, double vol_nhi,
                                                     const RtParams &p)
{
    const double tau_in = mul_unfused(cd_in, p.sig), tau_out = mul_unfused(cd_out, p.sig);   // un-fused, see rate_issue
    const double limit = p.fortran_consts ? (double)1.0e-7f : 1.0e-7;
    const double pref = flux * 1e48 / vol_nhi;
    if (fabs(tau_out - tau_in) > limit) return pref * (exp(-tau_in) - exp(-tau_out));
    return pref * (tau_out - tau_in) * exp(-tau_in);
}

// heating rate of one cell per atom (photorates.f90:118,124), used for the source cell only
__device__ __forceinline__ double heat_rate_per_atom(double flux, double cd_in, double cd_out, double vol_nhi,
                                                     const RtParams &p, const double2 *__restrict__ logtab)
{
    const double tau_in = mul_unfused(cd_in, p.sig), tau_out = mul_unfused(cd_out, p.sig);   // un-fused, see rate_issue
    const double limit = p.fortran_consts ? (double)1.0e-7f : 1.0e-7;
    const double pref = flux / vol_nhi;
    const bool thick = fabs(tau_out - tau_in) > limit;
    const double tau_thin = p.fortran_consts ? tau_in : tau_out;
    const int toff = thick ? 0 : table_stride(p.table_len);
    const Lookup A = lookup_issue<true>(p.tables, thick ? tau_in : tau_thin, p, logtab, toff);
    const Lookup B = lookup_issue<true>(p.tables, thick ? tau_out : tau_thin, p, logtab, toff);
    return thick ? pref * (lookup_heat(A) - lookup_heat(B)) : pref * (tau_out - tau_in) * lookup_heat(A);
}

__device__ __forceinline__ double photo_rate_per_atom(double flux, double cd_in, double cd_out, double vol_nhi,
                                                      const RtParams &p, const double2 *__restrict__ logtab)
{
    if (p.grey) return grey_rate_per_atom(flux, cd_in, cd_out, vol_nhi, p);
    return rate_value(rate_issue(flux, cd_in, cd_out, vol_nhi, p, logtab));
}

// ---------------------------------------------------------------------------------------------
// The octant kernel
// ---------------------------------------------------------------------------------------------
// Per-octant cell tables (built by build_unit_geometry, geometry.hip): a flat sequence of "steps" of RT_THREADS
// entries; the cells of a shell fill whole steps (the last one padded with invalid entries), so
// entry k*RT_THREADS + lane is what `lane` does in step k and the tables can be prefetched blindly.
//   cellA[e] = { abc, own slot | VALID | LAST_OF_SHELL, path (double, 2 words) }
//   cellB[e] = { slots of the four upstream corners in the previous shell's buffer }
// Dynamic LDS: [log table: 128 x {1/c, log2 c}][1/s: TABCAP doubles][wrapped i(a), j(b), k(c), mirrored: 4*TABCAP ints]
//              [shell buffer 0: max_cells+1 doubles][shell buffer 1: same]   (the last two unless GLOBAL_SCRATCH)
// TABCAP = 64 (S <= 63, workgroups of 64 or 128 threads: small traces, where LDS decides how many workgroups a CU
// holds), 256 (S <= 255) or 1024.
// Slot max_cells of each shell buffer holds 0.0: upstream corners of weight 0 point there.
// Diagnostic builds only (make EXTRA=-DASORA_ENABLE_ABLATION, tools/ablate.sh): ASORA_ABLATE=1 skips the rate
// atomics, 2 the rates, 4 the shell barriers, to attribute kernel time.  Production builds contain none of it.
#ifdef ASORA_ENABLE_ABLATION
#define ASORA_ABLATED(bit) ((p.ablate & (bit)) != 0)
#else
#define ASORA_ABLATED(bit) false
#endif

// buffer_atomic_add_f64 through a raw buffer descriptor: a lane whose offset lies beyond the descriptor's range is dropped
// by the hardware, so "this lane has nothing to add" is an offset of -1 and the instruction needs no branch around it
// (clang has no builtin for the f64 form; this binds the LLVM intrinsic).
// The offset of such a lane is 2 GiB, and the descriptors never span more than 2 GiB: the range check is
// offset + 8 > num_records in 32-bit arithmetic, so an offset near 2^32 would wrap around and pass it.
#define ASORA_OOB_OFFSET ((int)0x80000000u)
extern "C" __device__ double asora_buffer_atomic_fadd_f64(double, __amdgpu_buffer_rsrc_t, int, int, int)
    __asm("llvm.amdgcn.raw.ptr.buffer.atomic.fadd.f64");

// 1: the rate atomic of a step is issued in the NEXT step, right behind that step's table-lookup loads.  Vector
// memory operations complete in issue order, so a lookup issued after an atomic cannot return before the atomic has
// gone through the memory-side atomic unit (thousands of cycles under load); issued the other way round the lookup
// only waits for the atomic of the step before.  Costs 3 VGPRs.
#ifndef ASORA_LATE_ATOMIC
#define ASORA_LATE_ATOMIC 1
#endif

// 1: the table lookups of a step are CONSUMED in the next step (their rate is formed there, right before that step's
// own lookups are issued, and added behind them): a whole step of arithmetic hides their latency.  Pays where the
// kernel is latency-bound; costs 14 VGPRs (the kernel as a whole needs 96 with buffer atomics).  Measured on MI355X, 1000 sources,
// 256^3 (tools/ab_macro.sh ASORA_LATE_LOOKUP "0 1" ...): R = 16 -1.0 %, 24 -3.5 %, 32 -1.9 %, 48 -1.7 %, 64 -1.7 %.
#ifndef ASORA_LATE_LOOKUP
#define ASORA_LATE_LOOKUP 1
#endif

// 1: within a group of 8 sources the units are dispatched largest first (the z-sector units hold the most cells), so
// that the workgroups still running when the grid drains are the short ones.  Measured (tools/ab_macro.sh, 1000 sources,
// 256^3): R = 24 -2.6 %, 32 -1.8 %, 48 -1.7 %, 64 -0.5 %.
#ifndef ASORA_UNITS_LARGEST_FIRST
#define ASORA_UNITS_LARGEST_FIRST 1
#endif

// A wave whose 64 table entries of a step are all padding: 1 = skips the step's arithmetic and lookups, 2 = skips the
// interpolation only and issues the (unused) lookups like every other wave, 0 = runs the whole step.  With 1 the number
// of vector-memory operations in flight differs between the two paths, and where they meet the compiler waits as the
// shorter one demands (three operations too early on every wait of every step).
#ifndef ASORA_SKIP_EMPTY_WAVES
#define ASORA_SKIP_EMPTY_WAVES 2
#endif

// 1: see the comment at the loop entry of the kernel
#ifndef ASORA_PRIME_PIPELINE
#define ASORA_PRIME_PIPELINE 1
#endif

// 1: a scheduling barrier between the unrolled steps.  Without it the compiler hoists the decoding of a table entry (the
// address of the NEXT step's nHI) into the step that loaded the entry, i.e. waits for a load issued 250 instructions ago
// with two newer ones in flight instead of a whole step later.
#ifndef ASORA_STEP_SCHED_BARRIER
#define ASORA_STEP_SCHED_BARRIER 1
#endif

// 1: the two divisions of a cell (interpolation, flux / volume) through div_newton (rates_device.hpp) instead of the IEEE
// sequence.  Pays since the work counters stopped serialising the launch and the kernel runs at ~80 % of VALU issue.
#ifndef ASORA_NEWTON_DIVISION
#define ASORA_NEWTON_DIVISION 1
#endif
#if ASORA_NEWTON_DIVISION
#define ASORA_DIV(x, y) div_newton((x), (y))
#else
#define ASORA_DIV(x, y) ((x) / (y))
#endif

// waves per SIMD the register allocation must leave room for (2nd argument of __launch_bounds__; workgroups of up to 512
// threads).  Left alone the kernel takes 100 VGPRs (4 waves per SIMD); asked for 5 it fits 96 with three dwords spilled
// outside the loop, and is slower (R = 16 +5 %, R = 32 +2 %: profiles/r02_ab_work_counters.txt).
#ifndef ASORA_MIN_WAVES
#define ASORA_MIN_WAVES 1
#endif

// the same for the paired-sources variant (NSRC = 2): left alone it takes 134 VGPRs (3 waves per SIMD)
#ifndef ASORA_PAIR_MIN_WAVES
#define ASORA_PAIR_MIN_WAVES 1
#endif

// (round 6: the thick and thin rate table in LDS -- NumTau <= 2048, 2 x 16 KB, lookups as ds_read2_b64 gathers -- was measured on
//  twelve sector pairs with two workgroups per CU and is 11-14 % SLOWER than the global-memory lookups on the quiet medium and on an
//  evolving field alike: the LDS pipe is already a third to a half busy with corner reads, the logarithm table and the wrapped
//  coordinates, and 8 more divergent 16-byte gathers per wave-step tip it over.  LABNOTES round 6, profiles/r06_ab_lds_tables.txt)
// (round 4: non-temporal loads for the table-entry and nHI streams were measured and are slower -- the workgroups of a CU and of an
//  XCD share those lines through the L1 and L2: LABNOTES.md)
// (the flag bits of a table entry, CELL_*: asora_internal.hpp)

// GREY (ASORA_OPT_GREY_NOTABLES) is a compile-time variant although its branch is wave-uniform: the compiler counts the
// vector-memory operations in flight per PATH and, where paths of different counts meet, waits as the shortest one
// demands -- with the grey branch (one atomic, no lookups) in the loop every wait of the table path was three operations
// too early, i.e. the lookups had to be back at the top of the next step.
// BUFATOM: the rate atomics go through buffer descriptors over [phi | phi_t] and [heat | heat_t] (possible while the
// pair of grids stays within 2 GiB, N <= 512): the atomic of a step is then ONE unconditional instruction -- with
// `if (lane has a rate) atomic` the compiler sees a path without the atomic and counts one operation too few in flight
// behind every earlier load, i.e. every wait for a lookup also waited for the atomic issued after it.
// NSRC = 2: one workgroup sweeps its unit for TWO sources at once (consecutive entries of the source list, which a whole-list
// call has ordered by position: neighbours).  Everything a step derives from the table entry alone -- offsets, face, shell,
// bilinear weights and their products, path, volume factor, owner bits, the shell barrier -- is then evaluated once per
// lane-step instead of once per source (~45 of ~230 VALU instructions), the tables are streamed once, and a wave carries
// two independent dependency chains (LDS reads -> interpolation -> LDS write) that interleave.  What stays per source:
// the shell buffers and wrapped-coordinate tables in LDS, nHI, the interpolation, the lookups and the atomic.  An odd
// source count leaves the last workgroup with a copy of its first source whose rates are dropped (`have`).
// SUBBOX: the semantics of the reference's CPU function (src/c2ray/raytracing.f90:52-567, see subbox.hip) on this kernel's
// machinery.  A launch sweeps ONE sub-box = a range of shells = the table steps [sb_k0, sb_k1) of its unit, for the sources
// that are still growing; it starts from the trailing shell the previous sub-box's launch left in global memory and leaves
// its own there; cells on the faces of the box add what passes through them (phi_out, photorates.f90:120-125) to the
// source's photon loss; every source is rated with the flux of `flux_src` (f90:500,503).  Fortran-flavoured constants.
//
// Which combinations exist (the two static_asserts below admit exactly these; DESIGN.md 4.1 has the same table with the launch
// shapes that select them):
//   family                         THREADS x TABCAP                               GS   DUMP HEAT SKIP_ZERO GREY BUFATOM NSRC SUBBOX   launcher
//   paired production              {64,128,256,512}x256 {64,128,256}x64 256x32     f    f    f    f/t       f    t       2    f        launch_variant_pairs
//   single source, buffer atomics  {64..1024}x256 {64,128}x64 {256,512,1024}x1024  f    f    f/t  f/t       f/t  t       1    f        launch_variant
//   single source, global atomics  same (N > 512, option, shells in global memory) f/t  f    f/t  f (t: GS=f) f/t f       1    f        launch_variant
//   column-density dump            256 x {256,1024}                                f/t  t    f    f         f/t  f       1    f        launch_variant
//   sub-box sweep                  {256,512} x 256                                 f    f    f/t  f         f    t       1|2  t        launch_subbox_tables_variant
//   descriptors per layout (SPLIT) paired {256x32, 256x64, 256x256, 512x256}; single {256,512}x256            f    f    f/t  f/t       f/t  t       1|2  f        launch_variant_pairs / _split
//   (SKIP_ZERO with HEAT or GREY, NSRC = 2 with HEAT, DUMP, GREY or global atomics, SUBBOX with NSRC = 2 and HEAT: not built)
// split descriptors: does this unit's face write the [k][j][i] twin?  (the z-sector with the twins in use)
__device__ __forceinline__ bool ztr_desc(const RtParams &p, int uinfo) { return p.z_transposed != 0 && ((uinfo >> 8) & 3) == 3; }

template <int RT_THREADS, bool GLOBAL_SCRATCH, bool DUMP, bool HEAT, int TABCAP, bool SKIP_ZERO = false, bool GREY = false,
          bool BUFATOM = false, int NSRC = 1, bool SUBBOX = false, bool SPLIT = false>
__global__ void __launch_bounds__(RT_THREADS, (RT_THREADS <= 512 ? (NSRC == 2 ? ASORA_PAIR_MIN_WAVES : ASORA_MIN_WAVES) : 1)) raytrace_octant_kernel(const RtParams p)
{
    extern __shared__ double lds_raw[];
    static_assert(NSRC == 1 || (ASORA_LATE_LOOKUP && !GLOBAL_SCRATCH && !DUMP && BUFATOM), "paired sources: production variant only");
    static_assert(!SUBBOX || (ASORA_LATE_LOOKUP && !GLOBAL_SCRATCH && !DUMP && !SKIP_ZERO && !GREY), "sub-box sweep: table rates, shells in LDS");
    static_assert(!SPLIT || (BUFATOM && !SUBBOX && !DUMP && (NSRC == 1 || (!HEAT && !GREY)) && !(SKIP_ZERO && (HEAT || GREY))),
                  "descriptors per layout: the production forms for 512 < N <= 645, and the single-source form with heating or grey opacity");

    if (p.done_flag && *p.done_flag) return;   // evolve loop: an iteration enqueued beyond convergence does nothing
    const int blk = blockIdx.x;
    // blocks b and b+8 share an XCD (round-robin dispatch): keep the 8 octants of one source
    // on one XCD so that they share its L2 lines of nHI.  Speed only, never correctness.
    // p.units workgroups per source (8, 24, 12, 96 or 4: see ensure_geometry).
    // p.spread (a handful of sources): consecutive blocks are the units of ONE source, i.e. they go to different XCDs --
    // with the grouping above a single source would keep all its workgroups on one XCD's 32 CUs.
    int src_local, unit;      // src_local: index of the group of NSRC sources this workgroup sweeps
    if (p.spread) {
        src_local = blk / p.units;
        unit = blk % p.units;
    } else {
        src_local = (blk & 7) + 8 * (blk / (8 * p.units));
        unit = (blk >> 3) % p.units;
    }
#if ASORA_UNITS_LARGEST_FIRST
    unit = p.units - 1 - unit;   // sector units: z (most cells) first, x (fewest) last in dispatch order
#endif
    const int uinfo = p.geom[unit].info;           // sign bits of the unit | merged axes << 3 | rates-source << 6 | (face + 1) << 8
    // p.aligned: the unit's tables come in eight forms, by the source's position modulo 8 along the memory-contiguous axis of
    // the unit's face; the two sources of a workgroup agree in it (the host paired them so: p.pairs)
    const bool by_class = p.aligned != 0;
    const int face_type = ((uinfo >> 8) & 3) == 3 ? 1 : 0;          // 1: z-sector (rows along i), 0: x- / y-sector (rows along k)
    const bool listed = by_class && NSRC == 2;
    if (listed ? src_local >= p.npairs[face_type] : src_local * NSRC >= p.src_count) return;

    const int N = p.N;
    int i0[NSRC], j0[NSRC], k0[NSRC];
    int loc[NSRC];            // SUBBOX: the source's index within the batch (activity, photon loss, trailing shell)
    double flux[NSRC];
    bool have[NSRC];
    unsigned nreal = 0;
    int2 listed_pair = {0, -1};
    if (listed) listed_pair = p.pairs[face_type][src_local];
#pragma unroll
    for (int q = 0; q < NSRC; ++q) {
        int ns;
        if (listed) {
            have[q] = q == 0 || listed_pair.y >= 0;
            ns = (q == 0 || listed_pair.y < 0) ? listed_pair.x : listed_pair.y;
        } else {
            have[q] = src_local * NSRC + q < p.src_count;
            ns = p.src_begin + (have[q] ? src_local * NSRC + q : src_local * NSRC);
        }
        loc[q] = ns - p.src_begin;
        // SUBBOX: a source that stopped growing after an earlier box is swept along with its partner, its rates and its
        // photon loss dropped like those of the copy that fills an odd count; a workgroup without a live source ends here
        if (SUBBOX) have[q] = have[q] && p.sb_active[loc[q]] != 0;
        i0[q] = p.src_pos[3 * ns + 0];
        j0[q] = p.src_pos[3 * ns + 1];
        k0[q] = p.src_pos[3 * ns + 2];
        flux[q] = p.src_flux[(SUBBOX && p.flux_src >= 0) ? p.flux_src : ns];
        nreal += have[q] ? 1u : 0u;
    }
    if (SUBBOX && nreal == 0) return;
    const int table = by_class ? unit + p.units * ((face_type ? i0[0] : k0[0]) & 7) : unit;
    const uint4 *__restrict__ cellA = p.geom[table].cellA;
    const uint4 *__restrict__ cellB = p.geom[table].cellB;
    const int nsteps = p.geom[table].nsteps;
    // Tables that share their inner shells (clipped windows, geometry_device.hip): the steps before `inner_steps` are read through
    // the family's first table.  The step index is wave-uniform: a scalar compare and two scalar selects per table load.
    const int inner_info = p.geom[table].inner;
    const int inner_steps = inner_info >> 8;
    const uint4 *__restrict__ innerA = p.geom[inner_info & 255].cellA;
    const uint4 *__restrict__ innerB = p.geom[inner_info & 255].cellB;
    const int sa = (uinfo & 1) ? -1 : 1, sb = (uinfo & 2) ? -1 : 1, sc = (uinfo & 4) ? -1 : 1;

    // LDS: the small tables sit first, at compile-time offsets (TABCAP entries each), then the shell buffers
    double2 *logtab = reinterpret_cast<double2 *>(lds_raw);
    double *inv_s = reinterpret_cast<double *>(logtab + LOG_TABLE_SIZE);
    // per source, per axis: wrapped position of offset t on the unit's side [0, TABCAP) and on the mirrored side [TABCAP, 2 TABCAP)
    int *wtab = reinterpret_cast<int *>(inv_s + TABCAP);
    double *shells = reinterpret_cast<double *>(wtab + NSRC * 6 * TABCAP);
    const int slots = (p.max_cells + 2) & ~1;      // cells + the zero slot, even (keeps 16-B alignment)
    // source q's two shell buffers follow source q-1's: prev/cur of source q = prev/cur + q * 2 * slots
    double *prev, *cur;
    if (GLOBAL_SCRATCH) {
        prev = p.shell_scratch + (size_t)blk * 2 * slots;
        cur = prev + slots;
    } else {
        prev = shells;
        cur = prev + slots;
    }
    const int src_stride = 2 * slots;

    for (int t = threadIdx.x; t < LOG_TABLE_SIZE; t += RT_THREADS) logtab[t] = p.logtab[t];
    for (int t = threadIdx.x; t <= p.S; t += RT_THREADS) {
        inv_s[t] = 1.0 / (double)max(t, 1);
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            int *w = wtab + q * 6 * TABCAP;
            w[t] = wrap_once(i0[q] + sa * t, N);      // periodic position of offset t along each axis
            w[TABCAP + t] = wrap_once(i0[q] - sa * t, N);      // (|offset| <= N/2: one wrap suffices, raytracing.cu:270-272)
            w[2 * TABCAP + t] = wrap_once(j0[q] + sb * t, N);
            w[3 * TABCAP + t] = wrap_once(j0[q] - sb * t, N);
            w[4 * TABCAP + t] = wrap_once(k0[q] + sc * t, N);
            w[5 * TABCAP + t] = wrap_once(k0[q] - sc * t, N);
        }
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NSRC; ++q) { prev[q * src_stride + p.max_cells] = 0.0; cur[q * src_stride + p.max_cells] = 0.0; }
    }
    __syncthreads();

    const double sig = p.sig, dr = p.dr;
    const unsigned negmask = (sa < 0 ? 1u : 0u) | (sb < 0 ? 2u : 0u) | (sc < 0 ? 4u : 0u);
    const bool ztr = p.z_transposed != 0;
    constexpr bool grey = GREY;

    // work accounting, per wave (scalar registers: population counts of the lane masks, no per-lane adds)
    unsigned int n_gamma = 0, n_eval = 0;
    unsigned int n_zero_lane = 0;                           // this lane's rated cells whose rate was exactly +0 and was not added
    unsigned int src_cell_gamma = 0, src_cell_eval = 0;      // the source cell (thread 0 only)

    // rate accumulation: `idx` indexes [phi | phi_t] (and [heat | heat_t])
    // One descriptor over [phi | phi_t] while the pair fits 2 GiB (N <= 512).  Beyond (SPLIT, N <= 645 -- the reference's own
    // limit, raytracing.cu:95): a descriptor over ONE layout, N^3 doubles -- possible for units whose rated cells all lie on
    // one face (the sectors), where the layout is the same for the whole workgroup: the descriptor starts at the layout the
    // unit's face writes and `desc_off8` takes that layout's offset out of the byte offsets.  A compile-time variant: without
    // it the kernels of N <= 512 are instruction for instruction what they were (the scalar alone cost two VGPRs, i.e. the
    // fourth wave per SIMD of the paired kernels).
    const bool unit_in_twin = SPLIT && ztr_desc(p, uinfo);
    const unsigned desc_cells = SPLIT ? p.ncell : 2u * p.ncell;
    // (subtracted in UNSIGNED arithmetic and cast once: at N = 576 idx * 8 reaches 3.0e9 and the layout's offset 1.5e9 -- as signed
    //  ints their sum would overflow, which the hardware wraps but the language leaves undefined)
    const unsigned desc_off8 = (SPLIT && unit_in_twin) ? 8u * (unsigned)p.ncell : 0u;
    __amdgpu_buffer_rsrc_t rs_phi = __builtin_amdgcn_make_buffer_rsrc(p.phi + (unit_in_twin ? p.ncell : 0u), 0, BUFATOM ? (int)(8u * desc_cells) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_heat = __builtin_amdgcn_make_buffer_rsrc((HEAT ? p.heat : p.phi) + (unit_in_twin ? p.ncell : 0u), 0, (BUFATOM && HEAT) ? (int)(8u * desc_cells) : 0, 0x00020000);
    auto add_phi = [&](bool ok, unsigned idx, double v) {
        if (ASORA_ABLATED(1)) ok = ok && v == 1.2345e-300;
        if (ASORA_ABLATED(64)) idx &= 0xFFFFu;            // diagnostic: all rates into a 512 KiB window (wrong results)
        if (BUFATOM) (void)asora_buffer_atomic_fadd_f64(v, rs_phi, ok ? (int)(idx * 8u - desc_off8) : ASORA_OOB_OFFSET, 0, 0);
        else if (ok) unsafeAtomicAdd(p.phi + idx, v);
    };
    auto add_heat = [&](bool ok, unsigned idx, double v) {
        if (BUFATOM) (void)asora_buffer_atomic_fadd_f64(v, rs_heat, ok ? (int)(idx * 8u - desc_off8) : ASORA_OOB_OFFSET, 0, 0);
        else if (ok) unsafeAtomicAdd(p.heat + idx, v);
    };
    // BUFATOM: a pending rate is carried as the byte offset its atomic will use -- ASORA_OOB_OFFSET when the lane has nothing
    // to add -- instead of an index and a flag (a flag that lives across the step costs a select to make and a compare to use)
    auto add_phi_at = [&](int off, double v) {
        if (ASORA_ABLATED(1)) off = v == 1.2345e-300 ? off : ASORA_OOB_OFFSET;
        if (ASORA_ABLATED(64)) off &= 0x7FFF8;
        (void)asora_buffer_atomic_fadd_f64(v, rs_phi, off, 0, 0);
    };
    auto add_heat_at = [&](int off, double v) { (void)asora_buffer_atomic_fadd_f64(v, rs_heat, off, 0, 0); };

    // ---- SUBBOX, a later box of the source: continue from the trailing shell of the box before ------------
    const bool continues = SUBBOX && !p.sb_first;
    // (one trailing shell per source of the batch and unit)
    double *trail[NSRC];
#pragma unroll
    for (int q = 0; q < NSRC; ++q) trail[q] = SUBBOX ? p.sb_trail + ((size_t)loc[q] * p.units + unit) * (size_t)slots : nullptr;
    if (continues) {
#pragma unroll
        for (int q = 0; q < NSRC; ++q)
            if (have[q]) for (int t = threadIdx.x; t < p.max_cells; t += RT_THREADS) prev[q * src_stride + t] = trail[q][t];
    }
    // ---- shell 0: the source cell (raytracing.cu:285-294) -----------------------------------
    if (threadIdx.x == 0 && !continues) {
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            const unsigned idx = ((unsigned)i0[q] * N + j0[q]) * N + k0[q];
            const double nHI = p.nhi[idx];
            const double path = 0.5 * dr;
            const double cd_out = 0.0 + nHI * path;
            prev[q * src_stride] = cd_out;
            src_cell_eval += have[q] ? 1u : 0u;
            if ((uinfo & 64) && have[q]) {                 // the source cell is rated by exactly one unit
                if (DUMP) p.dump[idx] = cd_out;
                const double phi = photo_rate_per_atom(flux[q], 0.0, cd_out, dr * dr * dr * nHI, p, logtab);
                unsafeAtomicAdd(&p.phi[idx], phi);
                if (HEAT && !p.grey) unsafeAtomicAdd(&p.heat[idx], heat_rate_per_atom(flux[q], 0.0, cd_out, dr * dr * dr * nHI, p, logtab));
                src_cell_gamma += 1;
            }
        }
    }
    __syncthreads();

    // ---- shells 1..S, software-pipelined ------------------------------------------------------
    // The tables of a step do not depend on the medium, and the address of a cell's nHI only on
    // its table entry, so both are fetched ahead of the dependent arithmetic: tables two steps
    // ahead, nHI one step ahead.  The rate lookups of a step are issued after its shell barrier, so
    // their latency never holds the barrier up, and consumed a step later (ASORA_LATE_LOOKUP).
    // (The tables carry two all-invalid steps of padding at the end: prefetches stay in bounds.)
    auto nhi_address = [&](int q, unsigned abc, unsigned flags, unsigned &idx) -> const double * {
        const int *w = wtab + q * 6 * TABCAP;
        // (a cell on the mirrored side of an axis the unit merges reads the second half of that axis's table)
        const unsigned i = w[(abc & 1023) + ((flags >> CELL_NEG_SHIFT) & 1u) * TABCAP];
        const unsigned j = w[2 * TABCAP + ((abc >> 10) & 1023) + ((flags >> (CELL_NEG_SHIFT + 1)) & 1u) * TABCAP];
        const unsigned k = w[4 * TABCAP + ((abc >> 20) & 1023) + ((flags >> (CELL_NEG_SHIFT + 2)) & 1u) * TABCAP];
        const bool zt = ztr && (abc >> 30) == 2;
        // the [k][j][i] copies follow the [i][j][k] grids in memory: one 32-bit index covers both
        // (one formula with the outer and inner coordinate swapped, rather than two under a branch)
        const unsigned outer = zt ? k : i, inner = zt ? i : k;
        // (24-bit multiplies: N <= 1280 (device_init) so outer * N + j < 2^21; a 32-bit integer multiply runs at a quarter of their rate)
        // (spelled out: from __umul24 the compiler makes one 24-bit and one 64-bit multiply-add)
        unsigned row, cell;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(outer), "s"(N), "v"(j));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cell) : "v"(row), "s"(N), "v"(inner));
        idx = cell + (zt ? p.ncell : 0u);
        if (ASORA_ABLATED(128)) return p.nhi + (idx & 0xFFFFu);   // diagnostic: nHI from a 512 KiB window (wrong results)
        return p.nhi + idx;
    };

    // One step of one lane, straight-line (whole-wave instruction count is what matters, so
    // inactive lanes run the arithmetic on harmless padding values and only the LDS store and
    // the atomic are predicated).  `cur_*` is this step's table entry / nHI, `nxt_*` the next
    // step's (its nHI is requested here), `pf_*` the register set the entry two steps ahead is
    // loaded into.  The loop below is unrolled three times with the three register sets rotating,
    // so the pipeline needs no register-to-register copies.
    bool late_ok[NSRC];                // a rate computed in the previous step, not yet added (ASORA_LATE_ATOMIC)
#if !ASORA_LATE_LOOKUP
    double late_v[NSRC], late_h[NSRC];
#endif
    unsigned late_idx[NSRC];
    int late_off[NSRC];                // BUFATOM: byte offset of the pending rate's atomic, or ASORA_OOB_OFFSET
#if ASORA_LATE_LOOKUP
    Lookup pend_A[NSRC], pend_B[NSRC];             // lookups issued in the previous step, consumed in this one
    bool pend_thick[NSRC];
    double pend_pref[NSRC], pend_dtau[NSRC];
#endif
#pragma unroll
    for (int q = 0; q < NSRC; ++q) {
        late_ok[q] = false; late_idx[q] = 0; late_off[q] = ASORA_OOB_OFFSET;
#if !ASORA_LATE_LOOKUP
        late_v[q] = 0.0; late_h[q] = 0.0;
#else
        pend_A[q].t = pend_B[q].t = pend_A[q].h = pend_B[q].h = double2{0.0, 0.0};
        pend_A[q].residual = pend_B[q].residual = 0.0;
        pend_thick[q] = false; pend_pref[q] = 0.0; pend_dtau[q] = 0.0;
#endif
    }

    // SUBBOX: photons through the faces of the current sub-box (f90:541-543), per lane; the faces on the unit's own side
    // and on the mirrored side of each axis
    double loss[NSRC];
    bool pend_edge = false, cur_edge = false;     // (geometry only: the same for every source of the workgroup)
    double pend_pv[NSRC];                 // flux / volume of the pending cell (pref without the division by nHI)
#pragma unroll
    for (int q = 0; q < NSRC; ++q) { loss[q] = 0.0; pend_pv[q] = 0.0; }
    const int edge_own[3] = {sa > 0 ? p.sb_edge_r : p.sb_edge_l, sb > 0 ? p.sb_edge_r : p.sb_edge_l, sc > 0 ? p.sb_edge_r : p.sb_edge_l};
    const int edge_mir[3] = {sa > 0 ? p.sb_edge_l : p.sb_edge_r, sb > 0 ? p.sb_edge_l : p.sb_edge_r, sc > 0 ? p.sb_edge_l : p.sb_edge_r};

    // `first_c`: std::true_type for the steps of a unit's first triple -- the only ones that can hold shell 1, whose cells next to
    // the source get the diagonal factors of raytracing.cu:431-441; every other step is compiled without that block.
    auto step = [&](auto first_c, int k_pf, unsigned e_pf, const uint4 &cur_A, const uint4 &cur_B, const double (&cur_nhi)[NSRC], const unsigned (&cur_idx)[NSRC],
                    const uint4 &nxt_A, double (&nxt_nhi)[NSRC], unsigned (&nxt_idx)[NSRC], uint4 &pf_A, uint4 &pf_B) {
        constexpr bool FIRST_TRIPLE = decltype(first_c)::value;
#if ASORA_STEP_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);      // nothing of this step is scheduled into the previous one (see the macro)
#endif
        pf_A = (k_pf < inner_steps ? innerA : cellA)[e_pf];                      // two steps ahead (step k_pf)
        pf_B = (k_pf < inner_steps ? innerB : cellB)[e_pf];
#ifdef ASORA_DIAG_EXTRA_TABLE_LOAD      // diagnostic build only: 16 more bytes per lane and step from ANOTHER unit's table (equal sizes: octants)
        {
            const uint4 extra = p.geom[(unit + 1) % p.units].cellA[e_pf];
            n_eval += (extra.x == 0xdeadbeefu && extra.w == 0x12345u) ? 1u : 0u;
        }
#endif
#pragma unroll
        for (int q = 0; q < NSRC; ++q) nxt_nhi[q] = *nhi_address(q, nxt_A.x, nxt_A.y, nxt_idx[q]);          // one step ahead

        const bool valid = (cur_A.y & CELL_VALID) != 0;
        // waves whose 64 entries are all padding skip the arithmetic (wave-uniform branch)
#if ASORA_SKIP_EMPTY_WAVES
        const bool wave_has_work = __builtin_amdgcn_ballot_w64(valid) != 0ull;
#else
        const bool wave_has_work = true;
#endif
        bool rated[NSRC];
        double cd_in[NSRC], cd_out[NSRC], vol_nhi[NSRC];
        unsigned dst_idx[NSRC];
#pragma unroll
        for (int q = 0; q < NSRC; ++q) { rated[q] = false; cd_in[q] = 0.0; cd_out[q] = 0.0; vol_nhi[q] = 1.0; dst_idx[q] = 0; }
        if (wave_has_work) {
        // ---- what the table entry alone determines: once per lane-step, whatever NSRC -------------------------------
        const unsigned abc = cur_A.x;
        const int a = abc & 1023, b = (abc >> 10) & 1023, c = (abc >> 20) & 1023;
        const unsigned face = abc >> 30;                 // 2: dk = s, 1: dj = s, 0: di = s
        const int s = max(a, max(b, c));

        // ---- cinterp_gpu, raytracing.cu:345-535 ------------------------------------------
        // Bilinear weights: with alam = (s-1/2)/s the reference's dx = 2|alam*u - (u - 1/2)|
        // (raytracing.cu:397-403, source-relative) is 1 - u/s, so
        // s1..s4 = fu*fv, fv*(1-fu), fu*(1-fv), (1-fu)*(1-fv) with fu = u/s (raytracing.cu:405-408).
        // fu = 0 (resp. 1) makes the weights of the corners that do not exist in shell s-1 exactly 0.
        const int U = face == 0 ? b : a, V = face == 2 ? b : c;
        const double is = inv_s[s];
        // (u*(1/s) may miss 0 or 1 by an ulp; the corner that would then get a ~1e-16 weight does not exist
        //  in shell s-1 and reads the zero slot, so only the denominator moves, below rounding)
        const double fu = (double)U * is, fv = (double)V * is;
        const double gu = 1.0 - fu, gv = 1.0 - fv;
        const double w1 = fu * fv, w2 = fv * gu, w3 = fu * gv, w4 = gu * gv;
        const double path = __hiloint2double((int)cur_A.w, (int)cur_A.z) * dr;
        // a cell on an octant-boundary plane is rated by the octant with the + sign there
        const unsigned zmask = (cur_A.y >> CELL_ZERO_SHIFT) & 7u;        // (a == 0) | (b == 0) << 1 | (c == 0) << 2, tabulated
        const double maxcd = p.fortran_consts ? (double)2e30f : 2e30;                    // raytracing.cu:15
        const bool owner = valid && (cur_A.y & CELL_RATE) && (zmask & negmask) == 0;
        const double n2 = (double)(a * a + b * b + c * c);
        const double volfac = n2 * (dr * dr * FOURPI) * path;                              // raytracing.cu:302-307 without nHI
        const unsigned own_slot = cur_A.y & CELL_SLOT_MASK;
        if (SUBBOX) {       // is the cell on a face of the current sub-box?  only from the shell of the nearest face on (a
            // wave's entries belong to one shell: a wave-uniform branch; without clipping that is the box's last shell only)
            cur_edge = false;
            if (__builtin_amdgcn_readfirstlane(s) >= min(p.sb_edge_r, p.sb_edge_l)) {
                const unsigned nb = cur_A.y >> CELL_NEG_SHIFT;
                cur_edge = a == ((nb & 1u) ? edge_mir[0] : edge_own[0]) || b == ((nb & 2u) ? edge_mir[1] : edge_own[1]) ||
                           c == ((nb & 4u) ? edge_mir[2] : edge_own[2]);
            }
        }
        n_eval += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(valid)) * nreal;

        // ---- per source: the medium-dependent arithmetic ----------------------------------------------------------------
        // (round 4: loading the corner values of EVERY source before the first source's shell-buffer store -- the compiler
        //  cannot know that the store does not feed the other source's loads, so it orders them behind it -- was measured:
        //  0 at R = 16 ... 32, -1 % at R = 48, and 130 instead of 128 VGPRs, i.e. three waves per SIMD instead of four: not kept)
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            const double *pq = prev + q * src_stride;
            // w_n = s_n / max(0.6, c_n*sig) (raytracing.cu:33,422-425) and
            // cdensi = sum(c_n w_n)/sum(w_n) (raytracing.cu:428), with numerator and denominator
            // multiplied through by the four max() terms: one division instead of five.
            const double x1 = pq[cur_B.x], x2 = pq[cur_B.y], x3 = pq[cur_B.z], x4 = pq[cur_B.w];
            const double m1 = fmax(0.6, x1 * sig), m2 = fmax(0.6, x2 * sig);
            const double m3 = fmax(0.6, x3 * sig), m4 = fmax(0.6, x4 * sig);
            const double m12 = m1 * m2, m34 = m3 * m4;
            const double q1 = w1 * (m2 * m34), q2 = w2 * (m1 * m34);
            const double q3 = w3 * (m12 * m4), q4 = w4 * (m12 * m3);
#ifdef ASORA_DIAG_NO_DIVISION        // diagnostic build only (wrong values): what the two divisions of a cell cost
            double cdi = (x1 * q1 + x2 * q2 + x3 * q3 + x4 * q4) * __builtin_amdgcn_rcp(q1 + q2 + q3 + q4);
#else
            double cdi = ASORA_DIV(x1 * q1 + x2 * q2 + x3 * q3 + x4 * q4, q1 + q2 + q3 + q4);     // (the weights sum to 1, each max() is >= 0.6)
#endif
            if (FIRST_TRIPLE && s == 1) {                    // diagonal neighbours of the source, cu:431-441
                // (round 3: a wave's entries belong to one shell, so this could be a scalar branch instead of the dozen selects
                //  the compiler makes of it -- measured: no gain at R = 16 / 32, +5 % at R = 64: the branch splits the
                //  scheduling region around the LDS reads; left as it is)
                const int nz = (a == 0) + (b == 0) + (c == 0);
                const double r3 = p.fortran_consts ? (double)1.7320507764816284 : 1.73205080757; // f90:608 / cu:435
                const double r2 = p.fortran_consts ? (double)1.4142135381698608 : 1.41421356237; // f90:609 / cu:439
                if (nz < 2) cdi = (nz == 0 ? r3 : r2) * cdi;
            }
            cd_in[q] = cdi;

            // ---- the cell itself, raytracing.cu:270-276,311-328 -----------------------------
            const double nHI = cur_nhi[q];
            cd_out[q] = fma(nHI, path, cdi);
            if (valid) cur[q * src_stride + own_slot] = cd_out[q];
            if (DUMP) {
                if (owner) {
                    const int *w = wtab + q * 6 * TABCAP;
                    const unsigned nb = cur_A.y >> CELL_NEG_SHIFT;
                    const unsigned di = w[a + (nb & 1u) * TABCAP], dj = w[2 * TABCAP + b + ((nb >> 1) & 1u) * TABCAP];
                    const unsigned dk = w[4 * TABCAP + c + ((nb >> 2) & 1u) * TABCAP];
                    p.dump[(di * N + dj) * N + dk] = cd_out[q];
                }
            }
            const bool ok = owner && cdi <= maxcd && (NSRC == 1 || have[q]);
            rated[q] = ok && !ASORA_ABLATED(2);
            n_gamma += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ok));
            vol_nhi[q] = volfac * nHI;
            dst_idx[q] = cur_idx[q];
        }
        }

        if (__builtin_amdgcn_readfirstlane(cur_A.y) & CELL_LAST) {   // shell finished: publish it
            if (!ASORA_ABLATED(4)) __syncthreads();
            double *tmp = prev; prev = cur; cur = tmp;
        }

        // ---- rates, raytracing.cu:315-328 + rates.cu:16-41 ---------------------------------------
        if (grey) {
#pragma unroll
            for (int q = 0; q < NSRC; ++q)
                add_phi(rated[q], dst_idx[q], grey_rate_per_atom(flux[q], cd_in[q], cd_out[q], vol_nhi[q], p));
        } else if (wave_has_work || ASORA_SKIP_EMPTY_WAVES == 2) {
            // (lanes without a rate run the lookups on whatever they hold: the index is clamped for any input)
            // TAU_PHOTO_LIMIT: rates.cu:7 (double 1e-7) or photorates.f90:69 (single 1e-7 promoted)
            const double limit = p.fortran_consts ? (double)1.0e-7f : 1.0e-7;
            const double2 *tab = p.tables;
            bool thick[NSRC];
            double dtau[NSRC], arg_A[NSRC], arg_B[NSRC], tau_at_entry[NSRC];
            int toff[NSRC];
#pragma unroll
            for (int q = 0; q < NSRC; ++q) {
                const double tau_in = mul_unfused(cd_in[q], sig), tau_out = mul_unfused(cd_out[q], sig);   // un-fused, see rate_issue
                tau_at_entry[q] = tau_in;
                dtau[q] = tau_out - tau_in;
                thick[q] = fabs(dtau[q]) > limit;
                // one code path for both kinds of cell: per-lane table offset and arguments.  A thin cell looks its tau_thin up
                // twice: tau_in with the Fortran's constants (photorates.f90:121), tau_out with the CUDA library's (rates.cu:37)
                toff[q] = thick[q] ? 0 : table_stride(p.table_len);
                arg_A[q] = (thick[q] || p.fortran_consts) ? tau_in : tau_out;
                arg_B[q] = (thick[q] || !p.fortran_consts) ? tau_out : tau_in;
            }
            // SUBBOX: the second lookup of a THIN cell is the THICK table at tau_in: phi_in = pref T_thick(tau_in), which
            // the photon loss needs (phi_out = phi_in - phi, photorates.f90:120-125); its rate only uses the first
#if ASORA_LATE_LOOKUP
            // SKIP_ZERO (ASORA_OPT_SKIP_ZERO_RATES): a thick cell whose tau_in lies beyond the last table entry gets
            // pref * (T_last - T_last) = exactly +0; adding it changes nothing, so the atomic is not issued -- and when no
            // lane of the wave has anything to add, neither are the division, the logarithms and the lookups.  (pref must
            // be finite for the product to be 0 and not NaN: vol_nhi is checked instead of forming pref first.)  A kernel
            // variant of its own: the test and the branch cost 4.5 % where nothing can be left out.
            bool add[NSRC];
            bool wave_adds = true;
#pragma unroll
            for (int q = 0; q < NSRC; ++q) add[q] = rated[q];
            // SKIP_ZERO with BUFATOM (round 3; the variant launch_raytrace takes while its probes find such cells): the same
            // test per lane without a branch around the atomic -- the lane carries the out-of-range offset of a lane without
            // a rate, so its atomic never becomes a request -- and, below, whole waves of them skip the rate arithmetic.
            if (SKIP_ZERO) {
#pragma unroll
                for (int q = 0; q < NSRC; ++q) {
                    // (BUFATOM form: one compare.  A thick cell has nHI * path * sig > 1e-7, so nHI is an ordinary positive number
                    //  and pref = flux / (volume x nHI) is finite unless the volume factor itself is out of range, which the
                    //  branching variant's two extra tests guard against and this one leaves to vol_nhi > 0 being implied)
                    const bool zero_rate = (SKIP_ZERO && !BUFATOM)
                        ? (thick[q] && tau_at_entry[q] >= p.tau_zero && fabs(vol_nhi[q]) > 1e-250 && (vol_nhi[q] - vol_nhi[q] == 0.0))
                        : (thick[q] && tau_at_entry[q] >= p.tau_zero);
                    add[q] = rated[q] && !zero_rate;
                    n_zero_lane += (rated[q] && zero_rate) ? 1u : 0u;      // per lane: one add-with-carry; summed when the workgroup ends
                }
                // a wave none of whose lanes has anything to add leaves out the division, the logarithms and the index
                // arithmetic as well; the BUFATOM form then issues as many (wave-uniform, cheap) table loads as the other path,
                // so that the number of operations in flight is the same wherever the two paths meet
                bool any_add = add[0];
#pragma unroll
                for (int q = 1; q < NSRC; ++q) any_add = any_add || add[q];
                wave_adds = __builtin_amdgcn_ballot_w64(any_add) != 0ull;
            }
            {   // the previous step's lookups have had a whole step to arrive: form its rates now, issue this step's
                // lookups, then add the rates behind them
                double v_prev[NSRC], h_prev[NSRC], pref[NSRC];
                Lookup A2[NSRC], B2[NSRC];
#pragma unroll
                for (int q = 0; q < NSRC; ++q) {
                    const double ta = lookup_value(pend_A[q]), tb = lookup_value(pend_B[q]);
                    v_prev[q] = pend_thick[q] ? pend_pref[q] * (ta - tb) : pend_pref[q] * pend_dtau[q] * ta;
                    h_prev[q] = 0.0;
                    if (HEAT) {
                        const double ha = lookup_heat(pend_A[q]), hb = lookup_heat(pend_B[q]);
                        h_prev[q] = pend_thick[q] ? pend_pref[q] * (ha - hb) : pend_pref[q] * pend_dtau[q] * ha;
                    }
                    A2[q] = pend_A[q]; B2[q] = pend_B[q];
                    pref[q] = 0.0;
                }
                if (wave_adds) {
#pragma unroll
                    for (int q = 0; q < NSRC; ++q) {
#ifdef ASORA_DIAG_NO_DIVISION
                        pref[q] = flux[q] * __builtin_amdgcn_rcp(vol_nhi[q]);
#else
                        // (nHI = 0 -- a fully ionised or empty cell: the reference divides by zero; flux / +0 = flux * inf)
                        pref[q] = vol_nhi[q] == 0.0 ? flux[q] * INFINITY : ASORA_DIV(flux[q], vol_nhi[q]);
#endif
                        A2[q] = lookup_issue<HEAT>(tab, arg_A[q], p, logtab, toff[q]);
                        B2[q] = lookup_issue<HEAT>(tab, arg_B[q], p, logtab, SUBBOX ? 0 : toff[q]);
                    }
                }
                else if (BUFATOM && SKIP_ZERO) {
#pragma unroll
                    for (int q = 0; q < NSRC; ++q) {     // (distinct addresses, or the loads would be merged)
                        const double2 *__restrict__ dummy = tab + (threadIdx.x & 1) + 4 * q;
                        A2[q].t = dummy[0]; A2[q].residual = 0.0;
                        B2[q].t = dummy[2]; B2[q].residual = 0.0;
                        if (HEAT) { A2[q].h = dummy[2 * p.table_len]; B2[q].h = dummy[2 * p.table_len + 2]; }
                        else { A2[q].h = A2[q].t; B2[q].h = B2[q].t; }
                    }
                }
                if (SUBBOX) {
                    // what leaves the PREVIOUS step's cell through the far side, if that cell lies on a face of the box
                    if (__builtin_amdgcn_ballot_w64(pend_edge) != 0ull) {
#pragma unroll
                        for (int q = 0; q < NSRC; ++q) {
                            const bool lost = (BUFATOM ? late_off[q] != ASORA_OOB_OFFSET : late_ok[q]) && pend_edge;
                            const double ta = lookup_value(pend_A[q]), tb = lookup_value(pend_B[q]);
                            const double po = pend_thick[q] ? pend_pv[q] * tb : pend_pv[q] * (tb - pend_dtau[q] * ta);
                            if (lost) loss[q] += po;
                        }
                    }
                    // flux / volume of this step's cell, should it lie on a face: pref * nHI, or the quotient itself where
                    // nHI = 0 (pref = inf)
                    if (__builtin_amdgcn_ballot_w64(cur_edge) != 0ull) {
#pragma unroll
                        for (int q = 0; q < NSRC; ++q) {
                            double pv = pref[q] * cur_nhi[q];
                            if (__builtin_amdgcn_ballot_w64(cur_nhi[q] == 0.0) != 0ull) {
                                const double n2s = (double)((cur_A.x & 1023) * (cur_A.x & 1023) + ((cur_A.x >> 10) & 1023) * ((cur_A.x >> 10) & 1023) +
                                                            ((cur_A.x >> 20) & 1023) * ((cur_A.x >> 20) & 1023));
                                const double vol = n2s * (dr * dr * FOURPI) * (__hiloint2double((int)cur_A.w, (int)cur_A.z) * dr);
                                if (cur_nhi[q] == 0.0) pv = flux[q] / vol;
                            }
                            pend_pv[q] = pv;
                        }
                    }
                    pend_edge = cur_edge;
                }
#pragma unroll
                for (int q = 0; q < NSRC; ++q) {
                    if (BUFATOM) {
                        add_phi_at(late_off[q], v_prev[q]);
                        if (HEAT) add_heat_at(late_off[q], h_prev[q]);
                    } else {
                        add_phi(late_ok[q], late_idx[q], v_prev[q]);
                        if (HEAT) add_heat(late_ok[q], late_idx[q], h_prev[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < NSRC; ++q) {
                    pend_A[q] = A2[q]; pend_B[q] = B2[q]; pend_thick[q] = thick[q]; pend_pref[q] = pref[q]; pend_dtau[q] = dtau[q];
                    if (BUFATOM) late_off[q] = add[q] ? (int)(dst_idx[q] * 8u - desc_off8) : ASORA_OOB_OFFSET;
                    else { late_idx[q] = dst_idx[q]; late_ok[q] = add[q]; }
                }
            }
#elif ASORA_LATE_ATOMIC
            const double pref = flux[0] / vol_nhi[0];
            const Lookup A = lookup_issue<HEAT>(tab, arg_A[0], p, logtab, toff[0]);
            const Lookup B = lookup_issue<HEAT>(tab, arg_B[0], p, logtab, toff[0]);
            // the previous step's rate, behind this step's lookups in the memory pipeline
            add_phi(late_ok[0], late_idx[0], late_v[0]);
            if (HEAT) add_heat(late_ok[0], late_idx[0], late_h[0]);
            {
                const double ta = lookup_value(A), tb = lookup_value(B);
                late_v[0] = thick[0] ? pref * (ta - tb) : pref * dtau[0] * ta;     // see rate_value on the form of the difference
                if (HEAT) {      // photorates.f90:118,124 with the same table index and residual
                    const double ha = lookup_heat(A), hb = lookup_heat(B);
                    late_h[0] = thick[0] ? pref * (ha - hb) : pref * dtau[0] * ha;
                }
                late_idx[0] = dst_idx[0];
                late_ok[0] = rated[0];
            }
#else
            const double pref = flux[0] / vol_nhi[0];
            const Lookup A = lookup_issue<HEAT>(tab, arg_A[0], p, logtab, toff[0]);
            const Lookup B = lookup_issue<HEAT>(tab, arg_B[0], p, logtab, toff[0]);
            {
                const double ta = lookup_value(A), tb = lookup_value(B);
                add_phi(rated[0], dst_idx[0], thick[0] ? pref * (ta - tb) : pref * dtau[0] * ta);
                if (HEAT) {      // photorates.f90:118,124 with the same table index and residual
                    const double ha = lookup_heat(A), hb = lookup_heat(B);
                    add_heat(rated[0], dst_idx[0], thick[0] ? pref * (ha - hb) : pref * dtau[0] * ha);
                }
            }
#endif
        }
    };

    // SUBBOX: the steps of this launch's sub-box only (whole triples: the tables are padded at every box boundary)
    const int k_first = SUBBOX ? p.sb_k0[unit] : 0, k_last = SUBBOX ? p.sb_k1[unit] : nsteps;
    unsigned e = (unsigned)k_first * RT_THREADS + threadIdx.x;
    uint4 A0 = (k_first < inner_steps ? innerA : cellA)[e], B0 = (k_first < inner_steps ? innerB : cellB)[e];
    uint4 A1 = (k_first + 1 < inner_steps ? innerA : cellA)[e + RT_THREADS], B1 = (k_first + 1 < inner_steps ? innerB : cellB)[e + RT_THREADS];
    uint4 A2, B2;
    unsigned idx0[NSRC], idx1[NSRC], idx2[NSRC];
    double nhi0[NSRC], nhi1[NSRC], nhi2[NSRC];
#pragma unroll
    for (int q = 0; q < NSRC; ++q) {
        idx1[q] = idx2[q] = 0; nhi1[q] = nhi2[q] = 0.0;
        nhi0[q] = *nhi_address(q, A0.x, A0.y, idx0[q]);
    }
#if ASORA_LATE_LOOKUP && ASORA_PRIME_PIPELINE
    // The loop is entered with the loads in flight that a step leaves behind, in the same order (tables, nHI, two lookups):
    // the compiler merges the counts of outstanding operations of the loop entry with those of the back edge and waits as
    // the SHORTER history demands, so without these two (unused) lookups every first step of the unrolled three waited
    // for its nHI and its predecessor's lookups as if nothing else were in flight.
    if (!GREY) {
        __builtin_amdgcn_sched_barrier(0);          // behind the nHI load, as in a step
        const double2 *__restrict__ prime = p.tables + (threadIdx.x & 1);
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            // (distinct addresses, or the loads would be merged; a further source's come from the 128-entry log table)
            pend_A[q].t = q == 0 ? prime[0] : p.logtab[(threadIdx.x & 1) + 4 * q];
            if (HEAT) pend_A[q].h = prime[2 * p.table_len];
            pend_B[q].t = q == 0 ? prime[p.table_len] : p.logtab[(threadIdx.x & 1) + 4 * q + 2];
            if (HEAT) pend_B[q].h = prime[3 * p.table_len - 1];
        }
        if (BUFATOM) {                              // ... and the (dropped) atomics that follow them
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NSRC; ++q) {
                (void)asora_buffer_atomic_fadd_f64(0.0, rs_phi, ASORA_OOB_OFFSET, 0, 0);
                if (HEAT) (void)asora_buffer_atomic_fadd_f64(0.0, rs_heat, ASORA_OOB_OFFSET, 0, 0);
            }
        }
    }
#endif

    // nsteps is a multiple of 3 (the tables are padded to it) and is followed by two more
    // all-invalid steps, so every look-ahead stays inside the tables.
    // Shell 1 (at most 26 cells) is step 0 of a unit's table: the first triple is peeled off with the shell-1 block in it.
    int k = k_first;
    if (k == 0 && k < k_last) {
        step(std::true_type{}, k + 2, e + 2 * RT_THREADS, A0, B0, nhi0, idx0, A1, nhi1, idx1, A2, B2);
        step(std::true_type{}, k + 3, e + 3 * RT_THREADS, A1, B1, nhi1, idx1, A2, nhi2, idx2, A0, B0);
        step(std::true_type{}, k + 4, e + 4 * RT_THREADS, A2, B2, nhi2, idx2, A0, nhi0, idx0, A1, B1);
        k += 3; e += 3 * RT_THREADS;
    }
    for (; k < k_last; k += 3, e += 3 * RT_THREADS) {
        step(std::false_type{}, k + 2, e + 2 * RT_THREADS, A0, B0, nhi0, idx0, A1, nhi1, idx1, A2, B2);
        step(std::false_type{}, k + 3, e + 3 * RT_THREADS, A1, B1, nhi1, idx1, A2, nhi2, idx2, A0, B0);
        step(std::false_type{}, k + 4, e + 4 * RT_THREADS, A2, B2, nhi2, idx2, A0, nhi0, idx0, A1, B1);
    }
#pragma unroll
    for (int q = 0; q < NSRC; ++q) {
#if ASORA_LATE_LOOKUP
        const double ta = lookup_value(pend_A[q]), tb = lookup_value(pend_B[q]);
        const double v_last = pend_thick[q] ? pend_pref[q] * (ta - tb) : pend_pref[q] * pend_dtau[q] * ta;
        if (BUFATOM) add_phi_at(late_off[q], v_last); else add_phi(late_ok[q], late_idx[q], v_last);
        if (HEAT) {
            const double ha = lookup_heat(pend_A[q]), hb = lookup_heat(pend_B[q]);
            const double h_last = pend_thick[q] ? pend_pref[q] * (ha - hb) : pend_pref[q] * pend_dtau[q] * ha;
            if (BUFATOM) add_heat_at(late_off[q], h_last); else add_heat(late_ok[q], late_idx[q], h_last);
        }
#else
        add_phi(late_ok[q], late_idx[q], late_v[q]);
        if (HEAT) add_heat(late_ok[q], late_idx[q], late_h[q]);
#endif
    }

    if (SUBBOX) {
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            // the pending cell of the last step
            {
                const double ta = lookup_value(pend_A[q]), tb = lookup_value(pend_B[q]);
                const double po = pend_thick[q] ? pend_pv[q] * tb : pend_pv[q] * (tb - pend_dtau[q] * ta);
                if ((BUFATOM ? late_off[q] != ASORA_OOB_OFFSET : late_ok[q]) && pend_edge) loss[q] += po;
            }
            if (!have[q]) continue;
            // hand the last shell swept to the next sub-box's launch (the step that closed it ended with the barrier and the
            // swap: it is `prev`, complete)
            for (int t = threadIdx.x; t < p.max_cells; t += RT_THREADS) trail[q][t] = prev[q * src_stride + t];
            double l = loss[q];
            for (int o = 32; o > 0; o >>= 1) l += __shfl_down(l, o);
            if ((threadIdx.x & 63) == 0 && l != 0.0) unsafeAtomicAdd(p.sb_loss + loc[q], l * (dr * dr * dr));
        }
    }

    // work accounting: one atomic per wave and counter
    unsigned int n_zero = n_zero_lane;
    if (SKIP_ZERO) for (int o = 32; o > 0; o >>= 1) n_zero += __shfl_down(n_zero, o);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *slot = p.counters + COUNTER_FIELDS * (blockIdx.x & (COUNTER_SLOTS - 1));     // (see COUNTER_SLOTS)
        atomicAdd(slot, (unsigned long long)(n_gamma + src_cell_gamma));
        atomicAdd(slot + 1, (unsigned long long)(n_eval + src_cell_eval));
        if (n_zero) atomicAdd(slot + 2, (unsigned long long)n_zero);
        if (SKIP_ZERO && p.zero_probe) {      // a probe launch (launch_raytrace): its own pair of sums, spread like the others
            unsigned long long *pr = p.zero_probe + 2 * (blockIdx.x & (ZERO_PROBE_SLOTS - 1));
            atomicAdd(pr, (unsigned long long)(n_gamma + src_cell_gamma));
            if (n_zero) atomicAdd(pr + 1, (unsigned long long)n_zero);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host driver of the octant kernel (the role of do_all_sources_gpu's batch loop,
// raytracing.cu:101-143)
// ---------------------------------------------------------------------------------------------
static const size_t LDS_LIMIT_BYTES = 160 * 1024;

// Decomposition and workgroup size.  Shells of a small trace do not fill 256 lanes (R=16: <= 310 cells per
// octant shell); a large one needs so much LDS per octant that few workgroups fit a CU.  Three things pull:
//   * padding: every shell of every unit is padded to whole waves, so FEWER, LARGER units waste fewer lanes (R = 16:
//     1.41 lane-steps per rated cell with 8 octants, 1.04 with the whole sphere in one workgroup; R = 32: 1.17 with 12
//     sector pairs, 1.10 with 6 sectors whose transverse axes are both mirrored), and units that cover both signs of an
//     axis do not evaluate the plane between them twice;
//   * the rate atomics want long contiguous rows: units that mirror the memory-contiguous axis of a face sweep full
//     chords of the sphere (~15 % fewer 64-B atomic requests per rated cell than octants);
//   * LDS: a unit's two shell buffers must leave room for enough workgroups per CU, which is what stops the merging --
//     the whole sphere needs 39 KB at R = 16 and 157 KB at R = 32.
// Thresholds from sweeps on MI355X (1000 sources, 256^3; tools/sweep_units.sh, profiles/r03_sweep_units_run*.txt):
//   R < 11.5: whole sphere x 128 threads | < 15.5: x 256 | < 21.5: x 512 | < 23.5: half spheres x 256 | < 25.5: x 512 |
//   < 36.5: 6 sectors by sign of the dominant offset x 256 | < 52.5: 12 mirrored sector pairs x 256 | else x 512
// (round 2, without the merged kinds: R <= 12 octant pairs x 64 | 13..14 x 128 | 15..19 octants x 64 | 20..22 sector pairs
//  x 64 | 23..28 octant pairs x 256 | 29..35 sector pairs x 128 | 36..54 x 256 | >= 55 x 512 -- still what is used when the
//  merged kinds would leave CUs without a workgroup)
static void pick_launch_shape(const State &st, double R, int N, int src_count, bool dump, int &units, int &threads)
{
    const double r = std::min(R, 0.87 * N);                 // the window cuts the trace at ~sqrt(3)/2 N
    const double est_cells = 1.2 * r * r;                   // largest shell of an octant
    if (r <= 12.5) { units = 4; threads = 64; }             // pairs of whole octants mirrored in x
    else if (r <= 14.5) { units = 4; threads = 128; }
    else if (est_cells <= 450.0) { units = 8; threads = 64; }        // r <= 19.4
    else if (r <= 22.5) { units = 12; threads = 64; }
    else if (r <= 28.5) { units = 4; threads = 256; }
    else if (est_cells <= 1500.0) { units = 12; threads = 128; }     // r <= 35.3
    else if (est_cells <= 3500.0) { units = 12; threads = 256; }     // r <= 54
    else { units = 12; threads = 512; }
    {   // the merged kinds, when they still give every CU a couple of workgroups
        int mu = 0, mt = 0;
        if (r < 11.5) { mu = 1; mt = 128; }
        else if (r < 15.5) { mu = 1; mt = 256; }
        else if (r < 21.5) { mu = 1; mt = 512; }
        else if (r < 23.5) { mu = 2; mt = 256; }
        else if (r < 25.5) { mu = 2; mt = 512; }
        else if (r < 36.5) { mu = 6; mt = 256; }
        else if (r < 52.5) { mu = 12; mt = 256; }
        if (mu && (long)src_count * mu >= 2L * st.cu_count && !dump) { units = mu; threads = mt; }
        // Beyond N = 512 a buffer descriptor spans ONE layout of the rate grid (launch_raytrace, `split`), so the production forms --
        // two sources per workgroup, buffer atomics -- need units whose rated cells all lie on one kind of face.  Where the radius
        // would take the whole sphere or the half spheres (cells of every face in one workgroup: the global-atomic family, one
        // source per workgroup), the three all-sign sectors take their place (round 6; the reference's meshes end at 645,
        // ref: src/asora/raytracing.cu:95).  Measured at N = 576, 1000 sources (profiles/r06_ab_576_small_radii.txt): r_RT = 16
        // 0.286 -> 0.241 ms, r_RT = 24 0.605 -> 0.589 ms; below the radius from which two sources share a workgroup (15.5) the
        // whole sphere stays (r_RT = 12: 0.137 against 0.140 ms).
        if ((units == 1 || units == 2) && units == mu && r >= 15.5 && N > 512 && N <= 645 && (long)src_count * 3 >= 2L * st.cu_count &&
            !st.opt[ASORA_OPT_GLOBAL_ATOMICS] && st.opt[ASORA_OPT_Z_TRANSPOSED]) {
            units = 3;
            threads = r < 21.5 ? 256 : 512;
        }
    }
    // Few sources (fewer workgroups than CUs): the time of the call is the time of ONE workgroup, so cut a source
    // into more (24 sectors) and wider pieces.  One source, 128^3, R = 64: 0.235 -> 0.146 ms (tools/sweep_single_source.sh)
    // A few dozen sources (workgroups for half the CUs' slots at most): still one round, wider workgroups finish it sooner
    // (64 sources, R = 32: pairs x 256 threads 0.183 ms against 0.202 with 128; tools/sweep_mid_counts.sh)
    if ((long)src_count * 12 <= (long)st.cu_count * 4 && units == 12 && threads == 128) threads = 256;
    if ((long)src_count * 12 < (long)st.cu_count) {
        units = 24;
        if (est_cells > 900.0) threads = std::max(threads, 512);
        if (est_cells > 2500.0) threads = 1024;
    }
    // A handful of sources: 24 workgroups each still leave most CUs idle -- quarter the sectors (96 per source; each wedge
    // re-derives the inner part of its sector, so evaluations double, time about halves)
    // (measured, one source: 128^3 whole box 0.205 -> 0.157 ms, R = 32 0.051 -> 0.043 ms; the floor is the chain of shells,
    //  ~2 us each, not the work: tools/sweep_single_source.sh)
    if ((long)src_count * 48 <= (long)st.cu_count) {
        units = 96;
        threads = est_cells > 6000.0 ? 1024 : est_cells > 2500.0 ? 512 : 256;
    }
    const int want_sectors = st.opt[ASORA_OPT_SECTORS];
    if (want_sectors == 1) units = 8;
    if (want_sectors == 2) units = 24;
    if (want_sectors == 3) units = 12;
    if (want_sectors == 4) units = 96;
    if (want_sectors == 5) units = 4;
    if (want_sectors == 6) units = 1;
    if (want_sectors == 7) units = 2;
    if (want_sectors == 8) units = 3;
    if (want_sectors == 9) units = 6;
    const int forced = st.opt[ASORA_OPT_BLOCK_THREADS];
    if (forced == 64 || forced == 128 || forced == 256 || forced == 512 || forced == 1024) threads = forced;
    if (dump) threads = 256;                                // the column-density dump variant is built for 256 only
}

// Does sweeping two sources per workgroup pay?  (auto mode of ASORA_OPT_PAIR_SOURCES; from A/B runs on MI355X, 1000 sources,
// 256^3: profiles/r03_ab_two_sources.txt -- R = 16 -4 %, 20 -2 %, 26 -10 %, 28 -12 %, 32 -7 %, 40 -8 %, 48 -5 %, 56 and 64 0 %;
// nothing below R ~ 15, where the whole sphere is swept by few waves)
static bool pair_sources_pays(const State &st, double R, int N, int src_count, int units, int threads)
{
    const double r = std::min(R, 0.87 * N);
    if (!(r >= 15.5 && r < 52.5)) return false;
    // enough waves must be left to fill the chip (two per SIMD)
    return (long)(src_count / 2) * units * (threads / 64) >= 8L * st.cu_count;
}

// Can the rate atomics go through buffer descriptors (the kernel's BUFATOM)?  One descriptor over both layouts of the rate
// grid while the pair fits 2 GiB (N <= 512); one per layout (split) up to 2 GiB per layout (N <= 645) for units whose rated
// cells all lie on one face -- the sector kinds.
// The split form is built for the shapes such meshes take at ordinary radii: 256 or 512 threads, up to 256 shells; two sources
// per workgroup for table rates without heating, one source per workgroup also with heating or grey opacity (everything else
// beyond N = 512 -- column-density dump, more than 256 shells, whole spheres below r = 15.5 -- keeps the global-atomic family).
static bool buffer_atomics_fit(const State &st, const RtParams &p, int units, int threads, bool plain_tables, bool &split)
{
    split = false;
    if (st.opt[ASORA_OPT_GLOBAL_ATOMICS]) return false;
    if (16ull * p.ncell <= 0x80000000ull) return true;
    const bool one_face = units == 3 || units == 6 || units == 12 || units == 24 || units == 96;
    if (8ull * p.ncell <= 0x80000000ull && one_face && p.z_transposed && (threads == 256 || threads == 512) && plain_tables) { split = true; return true; }
    return false;
}

constexpr size_t lds_table_bytes(int tabcap, int nsrc = 1) { return LOG_TABLE_SIZE * sizeof(double2) + (size_t)tabcap * (sizeof(double) + (size_t)nsrc * 6 * sizeof(int)); }

// the paired-sources variant (NSRC = 2) exists for the production path only: table rates, no heating, no dump, shell
// buffers in LDS, buffer atomics
template <int T, int TABCAP, bool SPLIT = false>
static int launch_variant_pairs(State &st, const RtParams &q, unsigned grid, size_t lds_bytes, hipStream_t stream, bool skip_zero)
{
    if (skip_zero) {
        ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<T, false, false, false, TABCAP, true, false, true, 2, false, SPLIT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((raytrace_octant_kernel<T, false, false, false, TABCAP, true, false, true, 2, false, SPLIT>), dim3(grid), dim3(T),
                           lds_bytes, stream, q);
    } else {
        ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<T, false, false, false, TABCAP, false, false, true, 2, false, SPLIT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((raytrace_octant_kernel<T, false, false, false, TABCAP, false, false, true, 2, false, SPLIT>), dim3(grid), dim3(T),
                           lds_bytes, stream, q);
    }
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// single source, buffer descriptors per layout (512 < N <= 645): table rates (plain and zero-skipping form), with heating, grey opacity
template <int T>
static int launch_variant_split(State &st, const RtParams &q, unsigned grid, size_t lds_bytes, hipStream_t stream, bool skip_zero, bool heat)
{
#define ASORA_LAUNCH_SPLIT(HT, SZ, GR)                                                                                                 \
    do {                                                                                                                           \
        ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<T, false, false, HT, 256, SZ, GR, true, 1, false, true>, \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                            \
        hipLaunchKernelGGL((raytrace_octant_kernel<T, false, false, HT, 256, SZ, GR, true, 1, false, true>), dim3(grid), dim3(T), lds_bytes, stream, q); \
    } while (0)
    if (q.grey)         ASORA_LAUNCH_SPLIT(false, false, true);
    else if (heat)      ASORA_LAUNCH_SPLIT(true, false, false);
    else if (skip_zero) ASORA_LAUNCH_SPLIT(false, true, false);
    else                ASORA_LAUNCH_SPLIT(false, false, false);
#undef ASORA_LAUNCH_SPLIT
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

template <int T, int TABCAP>
static int launch_variant(State &st, const RtParams &q, unsigned grid, size_t lds_bytes, bool use_lds, bool dump, bool heat,
                          hipStream_t stream, bool skip_zero)
{
#define ASORA_LAUNCH(GS, DP, HT, SZ, GR, BA)                                                                           \
    do {                                                                                                           \
        ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<T, GS, DP, HT, TABCAP, SZ, GR, BA>, \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));            \
        hipLaunchKernelGGL((raytrace_octant_kernel<T, GS, DP, HT, TABCAP, SZ, GR, BA>), dim3(grid), dim3(T),       \
                           lds_bytes, stream, q);                                                                  \
    } while (0)
    const bool grey = q.grey != 0;
    // rate atomics through buffer descriptors (see the kernel's BUFATOM): [phi | phi_t] must not exceed 2 GiB
    // (q.split_desc: launch_raytrace has decided that the descriptors go per layout; those forms are launched from there)
    const bool ba = use_lds && !q.split_desc && 16ull * q.ncell <= 0x80000000ull && !st.opt[ASORA_OPT_GLOBAL_ATOMICS];
    if (T == 256 && dump) {
        if (grey)         { if (use_lds) ASORA_LAUNCH(false, true, false, false, true, false); else ASORA_LAUNCH(true, true, false, false, true, false); }
        else              { if (use_lds) ASORA_LAUNCH(false, true, false, false, false, false); else ASORA_LAUNCH(true, true, false, false, false, false); }
    }
    else if (grey)        { if (ba) ASORA_LAUNCH(false, false, false, false, true, true); else if (use_lds) ASORA_LAUNCH(false, false, false, false, true, false); else ASORA_LAUNCH(true, false, false, false, true, false); }
    else if (heat)        { if (ba) ASORA_LAUNCH(false, false, true, false, false, true); else if (use_lds) ASORA_LAUNCH(false, false, true, false, false, false); else ASORA_LAUNCH(true, false, true, false, false, false); }
    // ASORA_OPT_SKIP_ZERO_RATES = 1 where the buffer-atomic kernel (which drops exact zeros by itself) is not available: the
    // branching variant that leaves them out
    else if (use_lds && !ba && std::isfinite(q.tau_zero) && st.opt[ASORA_OPT_SKIP_ZERO_RATES] == 1) ASORA_LAUNCH(false, false, false, true, false, false);
    else                  { if (ba && skip_zero) ASORA_LAUNCH(false, false, false, true, false, true); else if (ba) ASORA_LAUNCH(false, false, false, false, false, true); else if (use_lds) ASORA_LAUNCH(false, false, false, false, false, false); else ASORA_LAUNCH(true, false, false, false, false, false); }
#undef ASORA_LAUNCH
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

// Who shares a workgroup with whom when the tables are the aligned kind (build_unit_geometry, align_class): two sources
// that agree modulo 8 in k (units of the x- and y-sector: list 0) or in i (units of the z-sector: list 1).  The list is walked
// in its order -- position-sorted for a whole-list call -- and a source waits for the next one of its class, so partners are
// neighbours in the list, hence in space; what is left over at the end sweeps alone.
void release_pair_lists(State &st)
{
    for (auto &e : st.pair_lists) for (int ft = 0; ft < 2; ++ft) if (e.dev[ft]) (void)hipFree(e.dev[ft]);
    st.pair_lists.clear();
}

static int source_pairs_by_class(State &st, const int32_t *host_pos, const void *list, int begin, int count, const State::PairList *&out)
{
    for (const auto &e : st.pair_lists)
        if (e.list == list && e.begin == begin && e.count == count) { out = &e; return 0; }
    if (st.pair_lists.size() >= 64) release_pair_lists(st);
    State::PairList e{list, begin, count, {nullptr, nullptr}, {0, 0}};
    for (int ft = 0; ft < 2; ++ft) {
        const int axis = ft ? 0 : 2;
        std::vector<int2> pairs;
        pairs.reserve((size_t)count / 2 + 8);
        int open[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        for (int s = begin; s < begin + count; ++s) {
            const int c = host_pos[3 * (size_t)s + axis] & 7;
            if (open[c] < 0) open[c] = s;
            else { pairs.push_back(int2{open[c], s}); open[c] = -1; }
        }
        for (int c = 0; c < 8; ++c) if (open[c] >= 0) pairs.push_back(int2{open[c], -1});
        e.n[ft] = (int)pairs.size();
        ASORA_HIP_TRY(hipMalloc(&e.dev[ft], std::max<size_t>(1, pairs.size()) * sizeof(int2)));
        ASORA_HIP_TRY(hipMemcpy(e.dev[ft], pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    st.pair_lists.push_back(e);
    out = &st.pair_lists.back();
    return 0;
}

// {rated pairs, exact zeros left out} of a probe launch, summed over its slots into pinned host memory
__global__ void __launch_bounds__(ZERO_PROBE_SLOTS) zero_probe_sum_kernel(const unsigned long long *__restrict__ slots, unsigned long long *out)
{
    __shared__ unsigned long long r[2][ZERO_PROBE_SLOTS];
    r[0][threadIdx.x] = slots[2 * threadIdx.x]; r[1][threadIdx.x] = slots[2 * threadIdx.x + 1];
    __syncthreads();
    for (int o = ZERO_PROBE_SLOTS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r[0][threadIdx.x] += r[0][threadIdx.x + o]; r[1][threadIdx.x] += r[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = r[0][0]; out[1] = r[1][0]; }
}

// How the radius behaves from call to call (a call = one raytrace of the library's API, one time step of the evolve loop):
// the eight-fold line-aligned tables only pay when they are reused (see launch_raytrace)
bool note_call_radius(State &st, double R, int path)
{
    // (one history per path -- 0: the whole-box ASORA trace and the evolve loop, 1: the sub-box sweep, whose R is R_max_LLS and
    //  whose calls alternate with the other's in a process that uses both: a shared history would see the radius "change" on
    //  every alternation and never let the aligned tables back in)
    State::RadiusHistory &h = st.rt_radius[path ? 1 : 0];
    if (R != h.last_R) {
        if (h.last_R >= 0.0) h.has_changed = true;
        h.last_R = R;
        h.same_R_calls = 0;
    }
    h.same_R_calls += 1;
    return !h.has_changed || h.same_R_calls > 32;
}

int launch_raytrace(State &st, RtParams &p, bool dump, bool heat, hipStream_t side)
{
    int units, threads;   // one workgroup per (source, octant) or per (source, octant, sector)
    pick_launch_shape(st, p.R, p.N, p.shape_src_count > 0 ? p.shape_src_count : p.src_count, dump, units, threads);
    {   // number of shells, known before the tables are built: the 1024-entry LDS tables exist for 256/512 threads
        const double R2hi = p.R * p.R * (1.0 + 1e-9) + 1e-9;
        const int Emax = p.N / 2;
        const int S_est = std::isfinite(R2hi) ? (int)std::min((double)Emax, std::floor(std::sqrt(R2hi))) : Emax;
        if (S_est + 1 > 256 && threads < 256) threads = 256;
    }
    // rows cut at 64-byte lines (ASORA_OPT_ALIGNED_ROWS): units of one face, a mesh whose rows start on lines, the [k][j][i]
    // twin for the z-faces, a source list whose positions the host knows (pairing), and eight times the tables.  By default
    // where two sources share a workgroup (r < 52.5: -1.5 ... -4.7 %; beyond, with one source per workgroup and 0.13-0.2 GB of
    // tables, nothing: profiles/r03_ab_aligned_rows.txt); on request up to r ~ 110
    const int32_t *host_pos = p.src_pos == st.src_pos_sorted ? st.src_pos_sorted_host.data()
                            : p.src_pos == st.src_pos ? st.src_pos_host.data() : nullptr;
    // ... and, left to the library, a radius that stays: building eight forms costs eight times as long (12 ms instead of
    // 2.5 at r = 30), which a run whose r_RT changes with every time step (R_max_LLS in cells under cosmological expansion,
    // ref: c2ray_base.py:460; a dozen launches per step) would pay every step for 0.3 ms saved.  So: aligned from the first
    // call on until the radius changes for the first time, afterwards only once a radius has served 32 calls in a row.
    // Decided ONCE PER CALL (note_call_radius, from fill_rt_params), never per launch: every range of a pipelined or chunked
    // call and every iteration of an evolve batch sees the same tables.
    bool aligned = false;
    {
        const int want = st.opt[ASORA_OPT_ALIGNED_ROWS];
        const double r = std::min(p.R, 0.87 * p.N);
        const bool possible = (units == 6 || units == 12) && !dump && p.N % 8 == 0 && p.z_transposed && host_pos != nullptr && r <= 110.0;
        aligned = possible && (want == 2 || (want == 0 && r < 52.5 && p.radius_stays));
    }
    if (int rc = ensure_geometry(st, p, threads, units, nullptr, aligned)) return rc;
    st.last_variant = 0;
    p.lut_k1 = 0.30102999566398119521 / p.dlogtau;      // log10(2)/dlogtau
    p.lut_k0 = 1.0 - p.minlogtau / p.dlogtau;
    // optical depth from which BOTH lookups of a thick cell return the same table value (index clamped to NumTau, or on
    // the last pair of the device table, whose slope is 0): 0.01 index units beyond the exact point, far more than the
    // 1e-12 the device's log2 can be off by
    // ASORA_OPT_SKIP_ZERO_RATES: 0 = where it costs nothing (the buffer-atomic kernels drop such lanes' atomics), 1 = everywhere
    // (kernels without buffer atomics take the branching SKIP_ZERO variant), 2 = nowhere
    p.tau_zero = INFINITY;
    const int skip_zero_opt = st.opt[ASORA_OPT_SKIP_ZERO_RATES];
    if (skip_zero_opt != 2 && !p.grey && p.lut_k1 > 0.0) {
        const double last = std::min(p.numtau_f, (double)(p.table_len - 1));
        p.tau_zero = std::exp2((last + 0.01 - p.lut_k0) / p.lut_k1);
    }

    const size_t slots = ((size_t)p.max_cells + 2) & ~(size_t)1;   // max_cells + zero slot, rounded to even (16-B alignment)
    // small tables (log table, 1/s, three wrapped-coordinate tables) at fixed capacity, then the shell buffers
    const bool big_tables = p.S + 1 > 256;
    const bool small_tables = p.S + 1 <= 64 && threads <= 128;
    if (p.S + 1 > 1024) return fail(4, "raytrace: more than 1023 shells (mesh too large for this build)");
    size_t fixed_bytes = lds_table_bytes(big_tables ? 1024 : small_tables ? 64 : 256);
    size_t shell_bytes = 2 * slots * sizeof(double);
    const bool use_lds = shell_bytes + fixed_bytes <= LDS_LIMIT_BYTES;
    bool split = false;
    // (beyond N = 512: per-layout descriptors also for the single-source form with heating or grey opacity, round 6)
    const bool bufatom_fits = buffer_atomics_fit(st, p, units, threads, use_lds && !dump && !big_tables, split);
    p.split_desc = split ? 1 : 0;
    // Two sources per workgroup (see the kernel's NSRC): the variant exists for table rates without heating, column-density
    // dump or exact-zero skipping, with the shell buffers in LDS and the rates through buffer atomics, for 64..512 threads
    // and the two smaller LDS table capacities
    bool pairs = false;
    {
        const int want = st.opt[ASORA_OPT_PAIR_SOURCES];
        const bool possible = use_lds && !dump && !heat && !p.grey && !big_tables && threads <= 512 &&
                              bufatom_fits && p.src_count >= 2 &&
                              2 * shell_bytes + lds_table_bytes((p.S + 1 <= 32 && threads == 256) ? 32 : (p.S + 1 <= 64 && threads <= 256) ? 64 : 256, 2) <= LDS_LIMIT_BYTES;
        pairs = possible && (want == 2 || (want == 0 && pair_sources_pays(st, p.R, p.N, p.shape_src_count > 0 ? p.shape_src_count : p.src_count, units, threads)));
    }
    const bool pairs_small = p.S + 1 <= 64 && threads <= 256;      // the paired variant has the 64-entry tables for 256 threads too
    // ... and 32-entry ones for 256 threads: at r_RT = 30 the 1.75 KB they save are what separates two workgroups per CU from
    // three (four shell buffers of 11.6 KB + tables: 53.9 KB against 52.2 KB; three per CU fit up to 53.3 KB; worth ~4 %)
    const bool pairs_tiny = p.S + 1 <= 32 && threads == 256;
    if (pairs) {
        fixed_bytes = lds_table_bytes(pairs_tiny ? 32 : pairs_small ? 64 : 256, 2);
        shell_bytes *= 2;
    }
    size_t lds_bytes = (use_lds ? shell_bytes : 0) + fixed_bytes;
#ifdef ASORA_ENABLE_ABLATION        // diagnostic builds only: unused LDS per workgroup, to lower the occupancy (ASORA_DIAG_EXTRA_LDS bytes)
    if (const char *e = getenv("ASORA_DIAG_EXTRA_LDS")) lds_bytes = std::min<size_t>(LDS_LIMIT_BYTES, lds_bytes + (size_t)atol(e));
#endif
    // launches that share the global shell scratch stay on the main stream (one at a time)
    hipStream_t stream = (side && use_lds) ? side : st.stream;
    if (side && !use_lds) {       // ... behind whatever the side streams still run, and the side streams behind it
        for (int q = 0; q < 2; ++q)
            if (st.side_pending[q]) ASORA_HIP_TRY(hipStreamWaitEvent(st.stream, st.side_done[q], 0));
    }

    // Exact-zero rates (ASORA_OPT_SKIP_ZERO_RATES, see the kernel's SKIP_ZERO).  The kernels that leave them out cost 5 % where
    // there are none, and save up to a quarter of the launch where most cells lie beyond the table (the neutral medium of an
    // early-reionisation run).  Left to the library (0) a launch takes them while the last PROBE found more than 15 % of the
    // rated pairs dark; a probe is such a launch whose two sums go to pinned host memory behind it and are looked at, without
    // waiting, by a later call: the first launch, every 64th after it, every launch while the dark variant runs anyway.
    bool skip_zero = false, probe = false;
    {
        const bool exists = use_lds && !dump && !heat && !p.grey && std::isfinite(p.tau_zero) && bufatom_fits;
        if (exists && skip_zero_opt == 1) skip_zero = true;
        else if (exists && skip_zero_opt == 0) {
            if (!st.zero_probe_dev) {
                ASORA_HIP_TRY(hipMalloc(&st.zero_probe_dev, 2 * ZERO_PROBE_SLOTS * sizeof(unsigned long long)));
                ASORA_HIP_TRY(hipHostMalloc(&st.zero_probe_host, 2 * sizeof(unsigned long long), hipHostMallocDefault));
                ASORA_HIP_TRY(hipEventCreateWithFlags(&st.zero_probe_done, hipEventDisableTiming));
            }
            if (st.zero_probe_pending && hipEventQuery(st.zero_probe_done) == hipSuccess) {
                const double rated = (double)st.zero_probe_host[0], dark = (double)st.zero_probe_host[1];
                st.zero_dark = rated > 0.0 && dark > 0.15 * rated;
                st.zero_known = true;
                st.zero_probe_pending = false;
            }
            (void)hipGetLastError();      // (hipErrorNotReady from the query is not an error)
            st.zero_since_probe += 1;
            probe = !st.zero_probe_pending && (!st.zero_known || st.zero_since_probe >= 64 || st.zero_dark);
            skip_zero = probe || st.zero_dark;
        }
    }

    int done = 0;
    while (done < p.src_count) {
        int batch = p.src_count - done;
        if (!use_lds) {
            // bound the global shell scratch to ~2 GiB per launch (the reference's source batching,
            // raytracing.cu:126, reappears only for traces whose shells outgrow LDS)
            const size_t per_src = (size_t)units * shell_bytes;
            const size_t budget = (size_t)2 << 30;
            int max_batch = (int)std::max<size_t>(8, (budget / per_src) / 8 * 8);
            batch = std::min(batch, max_batch);
            const size_t need = (size_t)8 * units * ((batch + 7) / 8) * shell_bytes;
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
        q.zero_probe = nullptr;
        if (probe && done == 0) {
            ASORA_HIP_TRY(hipMemsetAsync(st.zero_probe_dev, 0, 2 * ZERO_PROBE_SLOTS * sizeof(unsigned long long), stream));
            q.zero_probe = st.zero_probe_dev;
        }
        int groups = pairs ? (batch + 1) / 2 : batch;              // workgroups per unit
        if (pairs && aligned) {
            const State::PairList *pl = nullptr;
            if (int rc = source_pairs_by_class(st, host_pos, p.src_pos, q.src_begin, batch, pl)) return rc;
            for (int ft = 0; ft < 2; ++ft) { q.pairs[ft] = pl->dev[ft]; q.npairs[ft] = pl->n[ft]; }
            groups = std::max(pl->n[0], pl->n[1]);
        }
        q.spread = (long)groups * units <= 2L * st.cu_count ? 1 : 0;     // few workgroups: spread a source's units over the XCDs
        const unsigned grid = q.spread ? (unsigned)units * (unsigned)groups : 8u * (unsigned)units * (unsigned)((groups + 7) / 8);
        st.last_variant = (pairs ? ASORA_VARIANT_PAIRED : 0) | (aligned ? ASORA_VARIANT_ALIGNED : 0) |
                          ((bufatom_fits && use_lds && !dump) ? ASORA_VARIANT_BUFFER_ATOMICS : 0) | (split ? ASORA_VARIANT_SPLIT_DESCRIPTORS : 0) |
                          (skip_zero ? ASORA_VARIANT_SKIP_ZERO : 0) | (use_lds ? 0 : ASORA_VARIANT_GLOBAL_SHELLS) | (units << 8) | (threads << 16);
        {
            KernelTimer kt(ASORA_KERNEL_RAYTRACE, stream);
            int rc = 0;
            if (pairs && split) {         // 512 < N <= 645: sectors x 256 / 512 threads
                if (pairs_tiny)          rc = launch_variant_pairs<256, 32, true>(st, q, grid, lds_bytes, stream, skip_zero);
                else if (pairs_small)    rc = launch_variant_pairs<256, 64, true>(st, q, grid, lds_bytes, stream, skip_zero);
                else if (threads == 512) rc = launch_variant_pairs<512, 256, true>(st, q, grid, lds_bytes, stream, skip_zero);
                else                     rc = launch_variant_pairs<256, 256, true>(st, q, grid, lds_bytes, stream, skip_zero);
            } else if (split) {
                rc = threads == 512 ? launch_variant_split<512>(st, q, grid, lds_bytes, stream, skip_zero, heat)
                                    : launch_variant_split<256>(st, q, grid, lds_bytes, stream, skip_zero, heat);
            } else if (pairs) {
                if (pairs_tiny) rc = launch_variant_pairs<256, 32>(st, q, grid, lds_bytes, stream, skip_zero);
                else if (pairs_small) {
                    if (threads == 64)       rc = launch_variant_pairs<64, 64>(st, q, grid, lds_bytes, stream, skip_zero);
                    else if (threads == 128) rc = launch_variant_pairs<128, 64>(st, q, grid, lds_bytes, stream, skip_zero);
                    else                     rc = launch_variant_pairs<256, 64>(st, q, grid, lds_bytes, stream, skip_zero);
                } else switch (threads) {
                    case 64:  rc = launch_variant_pairs<64, 256>(st, q, grid, lds_bytes, stream, skip_zero); break;
                    case 128: rc = launch_variant_pairs<128, 256>(st, q, grid, lds_bytes, stream, skip_zero); break;
                    case 512: rc = launch_variant_pairs<512, 256>(st, q, grid, lds_bytes, stream, skip_zero); break;
                    default:  rc = launch_variant_pairs<256, 256>(st, q, grid, lds_bytes, stream, skip_zero); break;
                }
            } else if (big_tables) {
                if (threads == 1024)     rc = launch_variant<1024, 1024>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero);
                else if (threads == 512) rc = launch_variant<512, 1024>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero);
                else                rc = launch_variant<256, 1024>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero);
            } else if (small_tables) {
                if (threads == 64) rc = launch_variant<64, 64>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero);
                else               rc = launch_variant<128, 64>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero);
            } else switch (threads) {
                case 64:  rc = launch_variant<64, 256>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero); break;
                case 128: rc = launch_variant<128, 256>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero); break;
                case 512: rc = launch_variant<512, 256>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero); break;
                case 1024: rc = launch_variant<1024, 256>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero); break;
                default:  rc = launch_variant<256, 256>(st, q, grid, lds_bytes, use_lds, dump, heat, stream, skip_zero); break;
            }
            if (rc) return rc;
        }
        if (q.zero_probe) {
            hipLaunchKernelGGL(zero_probe_sum_kernel, dim3(1), dim3(ZERO_PROBE_SLOTS), 0, stream,
                               (const unsigned long long *)st.zero_probe_dev, st.zero_probe_host);
            ASORA_HIP_TRY(hipGetLastError());
            ASORA_HIP_TRY(hipEventRecord(st.zero_probe_done, stream));
            st.zero_probe_pending = true;
            st.zero_since_probe = 0;
        }
        done += batch;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The sub-box sweep of the reference's CPU function on this file's tabulated geometry (see the kernel's SUBBOX and
// subbox.hip for the semantics).  The tables hold the cells within R_max_LLS inside the traversal range -- the only ones
// that are rated or add to the photon loss; a source whose column densities go back to the caller needs the whole cube and
// stays with the on-the-fly kernel of subbox.hip.
// ---------------------------------------------------------------------------------------------
template <int T, bool HT, int NS = 1>
static int launch_subbox_tables_variant(const RtParams &q, unsigned grid, size_t lds_bytes, hipStream_t stream)
{
    ASORA_HIP_TRY(hipFuncSetAttribute((const void *)raytrace_octant_kernel<T, false, false, HT, 256, false, false, true, NS, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((raytrace_octant_kernel<T, false, false, HT, 256, false, false, true, NS, true>), dim3(grid), dim3(T), lds_bytes,
                       stream, q);
    ASORA_HIP_TRY(hipGetLastError());
    return 0;
}

int subbox_tables_prepare(State &st, RtParams &p, int ext_r, int ext_l, int subboxsize, int src_count, bool heat, SubboxTables &out,
                          const int32_t *host_pos)
{
    (void)heat;
    out = SubboxTables();
    out.host_pos = host_pos;
    const int want = st.opt[ASORA_OPT_SUBBOX_TABLES];
    if (want == 1 || p.grey || ext_r <= 0 || ext_l <= 0 || src_count < 1) return 0;
    if (!p.z_transposed || st.opt[ASORA_OPT_GLOBAL_ATOMICS] || 16ull * p.ncell > 0x80000000ull) return 0;   // the rates go through buffer atomics (one descriptor: N <= 512)
    const double R2hi = p.R * p.R * (1.0 + 1e-9) + 1e-9;
    const int range = std::max(ext_r, ext_l);
    const int S_tab = std::isfinite(R2hi) ? (int)std::min((double)range, std::floor(std::sqrt(R2hi))) : range;
    if (S_tab + 1 > 256) return 0;                                    // the variant is built with the 256-entry LDS tables
    // auto: when the radius, not the range, bounds the work (else the tables would hold the whole cube), and not beyond
    // what a table of a few tens of MB covers
    if (want == 0 && !(S_tab < std::min(ext_r, ext_l) && S_tab <= 96)) return 0;
    int units, threads;
    const double r = (double)S_tab;
    if (r < 15.5) { units = 1; threads = 256; }
    else if (r < 21.5) { units = 1; threads = 512; }
    else if (r < 23.5) { units = 2; threads = 256; }
    else if (r < 25.5) { units = 2; threads = 512; }
    else if (r < 36.5) { units = 6; threads = 256; }
    else if (r < 52.5) { units = 12; threads = 256; }
    else { units = 12; threads = 512; }
    if (want == 0 && (long)src_count * units < 2L * st.cu_count) return 0;     // a handful of sources: subbox.hip's wide workgroups
    const SubboxGeometry sbg{ext_r, ext_l, subboxsize};
    // Rows cut at 64-byte lines, as for the ASORA sweep (launch_raytrace): sectors of one face, a mesh whose rows start on
    // lines, positions the host knows.  ON REQUEST ONLY (ASORA_OPT_ALIGNED_ROWS = 2): measured on MI355X, 1000 sources, 256^3,
    // r_RT = 32 (profiles/r05_ab_subbox_aligned.jsonl), the aligned tables give this sweep nothing -- 1.136 / 1.151 ms aligned
    // against 1.139 / 1.143 ms packed with two sources per workgroup, 1.235 against 1.18 ms with one -- where they gave the
    // ASORA sweep -1.5 ... -4.7 %: here the photon-loss arithmetic and 141 VGPRs (three waves per SIMD), not the atomic
    // requests, set the pace.
    bool aligned = false;
    {
        const int want_a = st.opt[ASORA_OPT_ALIGNED_ROWS];
        const bool possible = (units == 6 || units == 12) && p.N % 8 == 0 && host_pos != nullptr && r <= 110.0;
        aligned = possible && want_a == 2;
    }
    if (int rc = ensure_geometry(st, p, threads, units, &sbg, aligned)) return rc;
    out.aligned = aligned;
    const size_t slots = ((size_t)p.max_cells + 2) & ~(size_t)1;
    if (2 * slots * sizeof(double) + lds_table_bytes(256) > LDS_LIMIT_BYTES) return 0;
    // one trailing shell per source and unit, within 2 GiB (82 KB per source at r_RT = 32: 26 000 sources per launch)
    const size_t per_source = (size_t)units * slots * sizeof(double);
    size_t trail_budget = (size_t)2 << 30;
    if (const char *e = getenv("ASORA_SUBBOX_TRAIL_BUDGET")) trail_budget = std::max<size_t>(per_source, (size_t)atoll(e));   // tests: force several batches
    const int max_batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)src_count, trail_budget / per_source));
    const size_t need = (size_t)max_batch * per_source;
    if (need > st.sb_trail_bytes) {
        if (st.sb_trail) ASORA_HIP_TRY(hipFree(st.sb_trail));
        st.sb_trail = nullptr; st.sb_trail_bytes = 0;
        ASORA_HIP_TRY(hipMalloc(&st.sb_trail, need));
        st.sb_trail_bytes = need;
    }
    p.sb_trail = st.sb_trail;
    out.units = units; out.threads = threads; out.S = p.S; out.max_batch = max_batch; out.ok = true;
    // two sources per workgroup (the kernel's NSRC = 2, round 4): as for the ASORA sweep -- table entry decoded once, tables
    // streamed once, two dependency chains per wave -- where four shell buffers fit and enough workgroups remain; not with
    // heating (two more lookups per source in flight)
    {
        const int want = st.opt[ASORA_OPT_PAIR_SOURCES];
        const bool possible = !heat && src_count >= 2 && 4 * slots * sizeof(double) + lds_table_bytes(256, 2) <= LDS_LIMIT_BYTES;
        const bool pays = r >= 15.5 && (long)(src_count / 2) * units * (threads / 64) >= 8L * st.cu_count;
        out.nsrc = (possible && (want == 2 || (want == 0 && pays))) ? 2 : 1;
    }
    return 0;
}

int subbox_tables_sweep(State &st, const RtParams &p, const SubboxTables &tab, int s_begin, int s_end, bool heat)
{
    RtParams q = p;
    bool any = false;
    for (int u = 0; u < tab.units; ++u) {
        const std::vector<int> &after = st.geom_step_after_shell[u];
        auto at = [&](int s) { return after[(size_t)std::min<long>(std::max(s, 0), (long)after.size() - 1)]; };
        q.sb_k0[u] = at(s_begin);
        q.sb_k1[u] = at(s_end);
        if (q.sb_k0[u] % 3 || q.sb_k1[u] % 3) return fail(11, "sub-box tables: a box boundary is not on a triple of steps (internal error)");
        any = any || q.sb_k1[u] > q.sb_k0[u];
    }
    if (!any) return 0;                       // the box lies beyond the radius: nothing is rated, nothing is lost
    q.sb_first = s_begin == 0 ? 1 : 0;
    const size_t slots = ((size_t)p.max_cells + 2) & ~(size_t)1;
    const bool pairs = tab.nsrc == 2 && !heat;
    const size_t lds_bytes = (pairs ? 4 : 2) * slots * sizeof(double) + lds_table_bytes(256, pairs ? 2 : 1);
    int groups = pairs ? (q.src_count + 1) / 2 : q.src_count;
    if (pairs && tab.aligned) {        // partners agree modulo 8 along the memory-contiguous axis of the unit's face
        const State::PairList *pl = nullptr;
        if (int rc = source_pairs_by_class(st, tab.host_pos, p.src_pos, q.src_begin, q.src_count, pl)) return rc;
        for (int ft = 0; ft < 2; ++ft) { q.pairs[ft] = pl->dev[ft]; q.npairs[ft] = pl->n[ft]; }
        groups = std::max(pl->n[0], pl->n[1]);
    }
    q.spread = (long)groups * tab.units <= 2L * st.cu_count ? 1 : 0;
    const unsigned grid = q.spread ? (unsigned)tab.units * (unsigned)groups : 8u * (unsigned)tab.units * (unsigned)((groups + 7) / 8);
    st.last_variant = (pairs ? ASORA_VARIANT_PAIRED : 0) | (tab.aligned ? ASORA_VARIANT_ALIGNED : 0) | ASORA_VARIANT_BUFFER_ATOMICS |
                      (tab.units << 8) | (tab.threads << 16);
    KernelTimer kt(ASORA_KERNEL_RAYTRACE);
    if (pairs) return tab.threads == 512 ? launch_subbox_tables_variant<512, false, 2>(q, grid, lds_bytes, st.stream)
                                         : launch_subbox_tables_variant<256, false, 2>(q, grid, lds_bytes, st.stream);
    if (tab.threads == 512) return heat ? launch_subbox_tables_variant<512, true>(q, grid, lds_bytes, st.stream)
                                        : launch_subbox_tables_variant<512, false>(q, grid, lds_bytes, st.stream);
    return heat ? launch_subbox_tables_variant<256, true>(q, grid, lds_bytes, st.stream)
                : launch_subbox_tables_variant<256, false>(q, grid, lds_bytes, st.stream);
}

} // namespace asora
