/*
 * asora_hip.h -- C-ABI of libasora_hip.so, the MI355X (gfx950) implementation of pyc2ray's
 * raytracing + chemistry hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes, and returns 0 on success or
 * a non-zero error code; asora_last_error() then holds a message.  Nothing throws across the
 * boundary (the reference throws std::runtime_error through the CPython C-API uncaught,
 * src/asora/memory.cu:72, src/asora/raytracing.cu:136).
 *
 * Section A is the drop-in boundary: one function per method of the reference's two extension
 * modules, with the reference binding each replaces cited (paths relative to the reference
 * checkout).  Section B is the device-resident extension the fused evolve loop uses (no
 * per-iteration PCIe copies).  Section C holds measurement/diagnostic hooks.
 *
 * Host buffers are owned by the caller and only read/written during the call.  Device memory
 * is process-global library state between asora_device_init and asora_device_close
 * (as in src/asora/memory.cu:20-29).  Not re-entrant, one grid size per process, one GPU per
 * process (asora_device_init_ex selects it).
 *
 * Grid element order: C-order logical [i][j][k], flat index N*N*i + N*j + k
 * (src/asora/raytracing.cu:30).  Sources: int32 0-based, xyz-interleaved
 * [x0,y0,z0,x1,...] (pyc2ray/utils/sourceutils.py:30); flux in units of 1e48 photons/s.
 */
#ifndef ASORA_HIP_H
#define ASORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* A. Drop-in boundary                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* libasora.device_init(N, num_src_par)            src/asora/python_module.cu:73-82 -> memory.cu:34-80.
 * num_src_par is accepted for signature compatibility; this build keeps the per-source
 * column-density scratch in LDS, so no num_src_par*N^3 slab is allocated. */
int asora_device_init(int N, int num_src_par);
/* Same, on an explicit device (one process per GPU under torch.distributed / MPI). */
int asora_device_init_ex(int N, int num_src_par, int device_id);

/* libasora.device_close()                         python_module.cu:87-92 -> memory.cu:119-129 */
int asora_device_close(void);

/* libasora.density_to_device(ndens_flat, N)       python_module.cu:97-109 -> memory.cu:85-88 */
int asora_density_to_device(const double *ndens, int N);

/* libasora.photo_table_to_device(thin, thick, NumTau)   python_module.cu:114-128 -> memory.cu:90-98.
 * NumTau = number of elements of each table. */
int asora_photo_table_to_device(const double *thin_table, const double *thick_table, int NumTau);

/* Heating tables (no counterpart in the reference's GPU library, which lists GPU heating as TODO,
 * pyc2ray/c2ray_base.py:424-426; the Fortran CPU path has them: src/c2ray/photorates.f90:118,124).
 * Same length and tau grid as the photo tables, which must have been uploaded first. */
int asora_heat_table_to_device(const double *heat_thin_table, const double *heat_thick_table, int NumTau);

/* libasora.source_data_to_device(pos, flux, NumSrc)     python_module.cu:133-148 -> memory.cu:99-114 */
int asora_source_data_to_device(const int32_t *pos, const double *flux, int NumSrc);

/* libasora.do_all_sources(R, coldensh_out, sig, dr, ndens, xh_av, phi_ion, NumSrc, m1,
 *                         minlogtau, dlogtau, NumTau)   python_module.cu:21-68 -> raytracing.cu:79-148.
 * coldensh_out and ndens are ignored, as in the reference (raytracing.cu:116: the density must
 * already be on the device).  Uploads xh_av, raytraces, downloads phi_ion (in place). */
int asora_do_all_sources(double R, double *coldensh_out, double sig, double dr, const double *ndens,
                         const double *xh_av, double *phi_ion, int NumSrc, int m1,
                         double minlogtau, double dlogtau, int NumTau);

/* libc2ray.chemistry.global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow,
 *                                colh0, temph0, abu_c) -> conv_flag
 * f2py wrapper of src/c2ray/chemistry.f90:13-48.  Elementwise: all six grids must share one
 * storage order; xh_av and xh_intermed are updated in place.  Runs on the GPU (uploads the six
 * grids, runs the fused kernel, downloads two); needs no prior asora_device_init. */
int c2ray_global_pass(double dt, const double *ndens, const double *temp, const double *xh,
                      double *xh_av, double *xh_intermed, const double *phi_ion,
                      double bh00, double albpow, double colh0, double temph0, double abu_c,
                      int m1, int m2, int m3, int *conv_flag);

/* libc2ray.raytracing.do_all_sources(normflux, srcpos, max_subbox, subboxsize, coldensh_out, sig, dr, ndens,
 *        xh_av, phi_ion, phi_heat, loss_fraction, photo_thin_table, photo_thick_table, heat_thin_table,
 *        heat_thick_table, minlogtau, dlogtau, R_max_LLS) -> (sum_nbox, photon_loss)
 * f2py wrapper of src/c2ray/raytracing.f90:52-119 (built with -DUSE_SUBBOX, src/c2ray/Makefile:3): the
 * reference's CPU raytracer.  Same semantics, evaluated on the GPU: per source a cube of half-width
 * min(max_subbox, N/2) is traced in sub-boxes of `subboxsize` cells until the photons crossing the box faces
 * fall to loss_fraction of the source's output (do_source, f90:193-221); rates are deposited only within
 * R_max_LLS (cells) and below the column-density cap; Fortran-flavoured constants.
 *   srcpos        (3,NumSrc) column-major, 1-based                       (f90:64)
 *   grids         Fortran order (m1,m2,m3), m1 == m2 == m3               (f90:65-70)
 *   coldensh_out  out: outgoing column density of the cells the LAST source reached, 0 elsewhere (f90:181)
 *   phi_ion       out (zeroed first, f90:95);  phi_heat  in/out (accumulated onto)
 *   xh_av, ndens  in
 *   sum_nbox, photon_loss  out: sub-boxes used by all sources / photons lost through their last boxes
 * Every cell's rate uses the flux of the LAST source, as the reference does (f90:500,503); set
 * ASORA_OPT_C2RAY_OWN_FLUX = 1 for each source's own flux.  Cells on a box face that deposit nothing
 * (beyond R_max_LLS or above the cap) contribute 0 to the loss (the reference adds an undefined value there).
 * Needs no prior asora_device_init: it initialises the library for m1 itself (and again when a later call
 * comes with another m1); a library initialised by asora_device_init for another mesh size makes it fail.
 * Overwrites the device grids NDENS, XH_AV, PHI_ION, PHI_HEAT. */
int c2ray_do_all_sources(const double *normflux, const int32_t *srcpos, int max_subbox, int subboxsize,
                         double *coldensh_out, double sig, double dr, const double *ndens, const double *xh_av,
                         double *phi_ion, double *phi_heat, float loss_fraction,
                         const double *photo_thin_table, const double *photo_thick_table,
                         const double *heat_thin_table, const double *heat_thick_table,
                         double minlogtau, double dlogtau, double R_max_LLS,
                         int NumTau, int NumSrc, int m1, int m2, int m3,
                         int *sum_nbox, double *photon_loss);

/* Message of the last failing call in this thread's process ("" if none). */
const char *asora_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* B. Device-resident extension (fused evolve loop; pyc2ray/evolve.py:168-240 restated so that  */
/*    only three scalars cross PCIe per iteration)                                              */
/* ------------------------------------------------------------------------------------------ */

/* Grid selectors for asora_grid_to_device / asora_grid_to_host / asora_device_ptr */
enum {
    ASORA_GRID_NDENS = 0,       /* hydrogen density                 (memory.cu n_dev)   */
    ASORA_GRID_XH_AV = 1,       /* time-averaged ionised fraction   (memory.cu x_dev)   */
    ASORA_GRID_PHI_ION = 2,     /* photo-ionisation rate            (memory.cu phi_dev) */
    ASORA_GRID_TEMP = 3,        /* temperature                                          */
    ASORA_GRID_XH = 4,          /* ionised fraction at start of the step                */
    ASORA_GRID_XH_INTERMED = 5, /* end-of-step ionised fraction                         */
    ASORA_GRID_PHI_HEAT = 6,    /* photo-heating rate (only filled when heating is on)  */
    ASORA_GRID_COUNT = 7
};

/* Upload / download one N^3 grid.  order = 'C': buffer is logical [i][j][k] C-contiguous;
 * order = 'F': buffer is Fortran-contiguous (element (i,j,k) at i + N*j + N*N*k) and is
 * transposed on the device. */
int asora_grid_to_device(int which, const double *host, int N, char order);
int asora_grid_to_host(int which, double *host, int N, char order);
/* Device-to-device copy between two grids (e.g. xh -> xh_av at the start of a step). */
int asora_grid_copy(int dst, int src);
/* grid *= factor on the device: the dilution of a device-resident density in a cosmological run
 * (ref: pyc2ray/c2ray_base.py:244-248 scales the host array). */
int asora_grid_scale(int which, double factor);
/* Sum over all cells of a grid, evaluated on the device in a fixed order (the two means of the reference's log line,
 * pyc2ray/evolve.py:160 `ndens.mean()`, `xh.mean()`, without a host pass over N^3 values). */
int asora_grid_sum(int which, double *sum);
/* Page-locked host memory for the grids a caller hands to asora_grid_to_host / asora_grid_to_device: copies into
 * pageable memory that has never been touched run at a fraction of the PCIe rate (page faults inside the copy).
 * The host side keeps a small pool of these behind the arrays evolve3D returns (pyc2ray_amd/_pinned.py). */
int asora_host_alloc(size_t bytes, void **host);
int asora_host_free(void *host);
/* Raw device pointer of a grid (for zero-copy views, e.g. an RCCL all-reduce of phi_ion
 * through torch.distributed).  NULL when not initialised. */
void *asora_device_ptr(int which);

/* Raytrace all uploaded sources from the device-resident ndens/xh_av into the device-resident
 * phi_ion (zeroed first).  Same arithmetic as asora_do_all_sources without the PCIe copies.
 * Sources [src_begin, src_begin+src_count) of the uploaded list are traced. */
int asora_raytrace_device(double R, double sig, double dr, int src_begin, int src_count,
                          double minlogtau, double dlogtau, int NumTau);

/* The same raytrace in three parts, for callers that overlap the multi-GPU sum of the rate grid with the trace
 * (one process per GPU; pyc2ray_amd/dist.py): with the sources uploaded in ascending order of their first
 * coordinate, the planes phi_ion[i][:][:] a chunk of sources can no longer reach are final and can be summed
 * across GPUs (RCCL on asora_device_ptr(ASORA_GRID_PHI_ION), ordered after asora_stream()) while the next chunk
 * is being traced.
 *   asora_raytrace_begin  zeroes the accumulators, forms nHI, fixes the parameters of the call;
 *   asora_raytrace_range  traces sources [src_begin, src_begin + src_count) into the accumulators (asynchronous);
 *   asora_raytrace_fold   completes planes [i_begin, i_begin + i_count) of phi_ion (adds the z-face accumulator,
 *                         which is kept in [k][j][i] order); every plane must be folded exactly once per call.
 * asora_raytrace_device(...) == begin; range(all); fold(0, N). */
int asora_raytrace_begin(double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau);
int asora_raytrace_range(int src_begin, int src_count);
/* asora_raytrace_begin for a rank of a multi-GPU run that only works on SOME planes of the grids: nHI is formed
 * from ndens and xh_av, and both rate accumulators are zeroed, on the `nruns` runs of planes
 * [runs[2q], runs[2q] + runs[2q+1]) only -- the planes this rank's sources reach plus the planes whose rates it
 * collects (pyc2ray_amd/dist.py, SlabPlan).  Everything else of the accumulators must already be zero (it stays
 * zero: nothing is traced into it), which one call of asora_raytrace_begin at the start of a time step ensures. */
int asora_raytrace_begin_planes(double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                                const int *runs, int nruns);
int asora_raytrace_fold(int i_begin, int i_count);
/* The HIP stream (hipStream_t) all of the library's work is ordered on. */
void *asora_stream(void);

/* One chemistry pass on the device-resident grids (ndens, temp, xh, xh_av, xh_intermed,
 * phi_ion): global_pass + the three reductions of pyc2ray/evolve.py:216-217.
 * Outputs: conv_flag (chemistry.f90:99-104), sum(xh_intermed), sum(1-xh_intermed). */
/* The raytracer of libc2ray.raytracing.do_all_sources (cubic sub-boxes, photon loss; see c2ray_do_all_sources in
 * section A) on device-resident inputs: NDENS and XH_AV on the device, tables from asora_photo_table_to_device
 * [+ asora_heat_table_to_device with ASORA_OPT_HEATING], sources [src_begin, src_begin + src_count) of
 * asora_source_data_to_device (0-based positions).  Leaves the rates in ASORA_GRID_PHI_ION (and PHI_HEAT, zeroed
 * first); returns the number of sub-boxes used and the photon loss.  This is what evolve3D(use_gpu=False) iterates. */
int asora_subbox_raytrace_device(int max_subbox, int subboxsize, float loss_fraction, double R_max_LLS, double sig,
                                 double dr, double minlogtau, double dlogtau, int NumTau, int src_begin, int src_count,
                                 int *sum_nbox, double *photon_loss);
/* asora_device_init for callers that have no device_init of their own (the libc2ray-compatible entry points):
 * initialises the library for N if it is not initialised, re-initialises it if it was initialised this way for
 * another N, and fails if asora_device_init was called for another N. */
int asora_device_init_auto(int N);

int asora_chemistry_device(double dt, double bh00, double albpow, double colh0, double temph0,
                           double abu_c, int *conv_flag, double *sum_xh1, double *sum_xh0);

/* The same pass slab by slab (planes [i_begin, i_begin + i_count) of every grid), for callers that start the
 * chemistry of the planes whose rates are already summed across GPUs while the rest is still in flight.
 * `first` != 0 resets the three reductions, later calls add to them in call order; asora_chemistry_finish
 * returns them (and is the only call of the group that waits for the device). */
int asora_chemistry_range(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                          int i_begin, int i_count, int first);
int asora_chemistry_finish(int *conv_flag, double *sum_xh1, double *sum_xh0);
/* Device address of the three reductions {sum(xh_intermed), sum(1 - xh_intermed), conv_flag} (doubles) that
 * asora_chemistry_range accumulates: a multi-GPU caller sums them across ranks in place (one RCCL all-reduce ordered after
 * asora_stream()) and reads them back once, instead of asora_chemistry_finish + a host round trip per rank. */
void *asora_reduction_ptr(void);

/* The whole outer loop of evolve3D (pyc2ray/evolve.py:168-240) on the device.  Per iteration the host of the
 * reference uploads xh_av, downloads phi_ion, reshapes, runs global_pass, forms two sums with numpy and tests
 * convergence (evolve.py:187,200,210,216-236); here an iteration is three launches -- the raytrace, ONE pass over
 * the grids that folds the rate accumulators, solves the chemistry, forms nHI of the new xh_av in both layouts for
 * the next raytrace and zeroes the accumulators, and a one-workgroup kernel that finishes the three reductions and
 * evaluates the convergence test -- so several iterations can be enqueued without the host in between; launches
 * enqueued beyond convergence see the device flag and do nothing, so the iteration count is exactly the reference's.
 *   asora_evolve_begin    NDENS, TEMP, XH must be on the device; xh_av = xh_intermed = xh (evolve.py:136-137) is implied.
 *                         conv_criterion = min(int(convergence_fraction N^3), (NumSrc-1)/3) (evolve.py:127) is the
 *                         caller's to compute (NumSrc is the TOTAL source count, evolve.py:346).
 *   asora_evolve_enqueue  enqueues `iterations` (1..32) outer iterations; asynchronous.
 *   asora_evolve_poll     waits for what was enqueued; returns the iterations carried out so far, whether the test has
 *                         passed, and one row {conv_flag, sum(xh_intermed), sum(1-xh_intermed), rel_change_xh1,
 *                         rel_change_xh0} per iteration not reported yet (at most history_rows rows; history = NULL: the
 *                         rows are given up).  At most 64 iterations may be enqueued between two polls.  The pass does
 *                         not store the folded rates (the accumulators alternate between two pairs instead, so the last
 *                         iteration's survive): the poll folds them into ASORA_GRID_PHI_ION, which is valid from then on.
 * Results: ASORA_GRID_XH_INTERMED (xh_new), ASORA_GRID_XH_AV, and -- after a poll -- ASORA_GRID_PHI_ION. */
int asora_evolve_begin(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                       double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                       int src_begin, int src_count, double conv_criterion, double convergence_fraction);
int asora_evolve_enqueue(int iterations);
int asora_evolve_poll(int *niter, int *converged, double *history, int history_rows, int *rows_written);

/* The same device-resident loop with the sources sharded over several GPUs (pyc2ray/evolve.py:249-498 evolve3D_MPI: sources cut
 * into contiguous blocks :360-371, rate grids summed over the ranks every iteration :433-437, results broadcast :480-497).
 * Here a rank traces its block of sources, owns the chemistry of the planes [own_begin, own_begin + own_count), sends the
 * rates it traced onto other ranks' planes to their owners and receives the new xh_av of the foreign planes its sources
 * reach (pyc2ray_amd/dist.py: SlabPlan, TorchComm).  One iteration, all calls asynchronous on asora_stream():
 *   asora_evolve_slab_trace       traces sources [src_begin, src_begin + src_count) of the step (one call, or one per chunk)
 *   asora_evolve_slab_fold_out    planes owned by ANOTHER rank: the rates traced onto them are summed over the two accumulator
 *                                 layouts into the out-box (asora_evolve_slab_outbox(): an N^3 grid, plane i at i*N*N), from
 *                                 where the caller sends them; the accumulators of the NEXT iteration are zeroed there
 *   asora_evolve_slab_add         rates received for OWN planes (a device buffer of i_count*N*N doubles; _add_host: a host
 *                                 buffer), added in call order
 *   asora_evolve_slab_pass        the fused pass of the one-GPU loop on the own planes (rates folded, chemistry, nHI of the new
 *                                 xh_av in both layouts, next accumulators zeroed); leaves THIS RANK'S {sum x, sum 1-x, conv_flag}
 *                                 at asora_reduction_ptr()
 *   asora_evolve_slab_nhi         planes whose new xh_av has just arrived in ASORA_GRID_XH_AV: nHI for the next trace
 *   asora_evolve_slab_close       host_sums = NULL: the caller has summed the three doubles at asora_reduction_ptr() over the
 *                                 ranks IN PLACE (one all-reduce ordered on asora_stream()); else the three sums over all ranks,
 *                                 from the host.  Evaluates the convergence test of evolve.py:216-236 on them on the device.
 * Every launch is gated by the device's `done` flag, so -- as on one GPU -- several iterations can be enqueued per
 * asora_evolve_poll, and every rank, working from identical sums, stops at the same iteration.  asora_evolve_poll folds the last
 * iteration's rates into ASORA_GRID_PHI_ION (complete on the own planes).  ASORA_GRID_XH_INTERMED / _XH_AV: own planes. */
int asora_evolve_begin_slab(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                            double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                            int src_begin, int src_count, double conv_criterion, double convergence_fraction,
                            int own_begin, int own_count);
int asora_evolve_slab_trace(int src_begin, int src_count);
int asora_evolve_slab_fold_out(int i_begin, int i_count);
/* The full-grid exchange of the reference (pyc2ray/evolve.py:433-437: MPI Allreduce of the rate grid, chemistry on identical data)
 * on the same loop, for a step begun with own_begin = 0, own_count = N on every rank: asora_evolve_slab_fold_all sums the two
 * accumulator layouts of ALL planes into the out-box; the caller all-reduces the out-box over the ranks IN PLACE (ordered on
 * asora_stream()); asora_evolve_slab_pass then takes the rates from the out-box, keeps them in ASORA_GRID_PHI_ION, and its three
 * sums are those of the whole grid already: asora_evolve_slab_close(NULL) without a reduction.  One iteration:
 * trace, fold_all, [all-reduce], pass, close; asora_evolve_poll then has nothing left to fold.  asora_evolve_slab_outbox_from_host: the out-box planes written from the host (for
 * transports that sum on the host). */
int asora_evolve_slab_fold_all(void);
void *asora_evolve_slab_outbox(void);
int asora_evolve_slab_outbox_to_host(int i_begin, int i_count, double *host);
int asora_evolve_slab_outbox_from_host(int i_begin, int i_count, const double *host);
int asora_evolve_slab_add(int i_begin, int i_count, const double *dev_planes);
int asora_evolve_slab_add_host(int i_begin, int i_count, const double *host_planes);
int asora_evolve_slab_pass(void);
int asora_evolve_slab_nhi(int i_begin, int i_count);
int asora_evolve_slab_close(const double *host_sums);

/* Runs of i-planes [i_begin, i_begin + i_count) of a grid to / from a host buffer of i_count*N*N doubles (C order).
 * What multi-GPU ranks exchange are such runs: the planes a rank's sources reach, the planes whose chemistry it owns. */
int asora_planes_to_host(int which, int i_begin, int i_count, double *host);
int asora_planes_to_device(int which, int i_begin, int i_count, const double *host);

/* ------------------------------------------------------------------------------------------ */
/* C. Options, measurement and diagnostics                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Behaviour options (asora_set_option).  Defaults follow the CUDA library being replaced. */
enum {
    /* 1: sqrt(2)/sqrt(3), 1e-7, 2e30 as the Fortran's single-precision parameters
     *    (src/c2ray/raytracing.f90:368,608-609, photorates.f90:69) and the thin-cell table
     *    argument tau_in (photorates.f90:121);  0 (default): CUDA literals and tau_out
     *    (src/asora/raytracing.cu:15,435,439, rates.cu:7,37). */
    ASORA_OPT_FORTRAN_CONSTANTS = 0,
    /* 1: analytic grey rates (GREY_NOTABLES builds, rates.cu:48-64); 0 (default): tables. */
    ASORA_OPT_GREY_NOTABLES = 1,
    /* 1: record HIP events around each kernel launch (asora_kernel_time_ms). */
    ASORA_OPT_TIMING = 2,
    /* 0: z-faces read/accumulate through the [k][j][i] transposed copies (default 1). */
    ASORA_OPT_Z_TRANSPOSED = 3,
    /* Workgroup size of the raytrace kernel: 0 (default) = chosen from R and the number of sources;
     * 64, 128, 256, 512 or 1024 force it. */
    ASORA_OPT_BLOCK_THREADS = 4,
    /* Decomposition of a source: 0 (default) = chosen from R; 1 = one workgroup per octant;
     * 2 = one per octant and dominant-axis sector (24 per source, diagonal planes re-derived);
     * 3 = one per pair of mirrored sectors (12 per source; rows are full chords of the sphere);
     * 4 = one per quarter of a sector plus the cells it reads (96 per source: a handful of sources);
     * 5 = one per pair of whole octants mirrored in x (4 per source);
     * 6 = ONE workgroup per source: the whole sphere (nothing evaluated twice, every row a full chord, the least padding;
     *     the shell buffers must fit LDS: small radii);
     * 7 = one per half sphere (x and z mirrored, split by the sign of dj: 2 per source);
     * 8 = one per dominant-axis sector with all signs (3 per source);
     * 9 = one per sector and sign of its dominant offset, both transverse axes mirrored (6 per source). */
    ASORA_OPT_SECTORS = 5,
    /* 1: the raytrace also accumulates the photo-heating rate into ASORA_GRID_PHI_HEAT
     *    (src/c2ray/photorates.f90:118,124; src/c2ray/raytracing.f90:532,537); needs heat tables. */
    ASORA_OPT_HEATING = 6,
    /* c2ray_do_all_sources only.  0 (default): every source's rates use the flux of the LAST source, as the
     *    reference does (src/c2ray/raytracing.f90:500,503 pass normflux(NumSrc));  1: each source its own flux. */
    ASORA_OPT_C2RAY_OWN_FLUX = 7,
    /* 1: never take the uniform-temperature form of the tiled chemistry pass (diagnostics / tests: both forms give
     *    bit-identical results).  0 (default): a temperature grid found uniform when uploaded is not read again. */
    ASORA_OPT_NO_UNIFORM_T = 8,
    /* 1: the sub-box raytracer keeps its shell buffers in global memory even when they would fit LDS (tests). */
    ASORA_OPT_SUBBOX_GLOBAL_SHELLS = 9,
    /* 1 (default): asora_do_all_sources overlaps the upload of xh_av, the trace and the download of phi_ion slab by slab
     *    (when all uploaded sources are traced and R is small against the mesh); 0: upload, trace, download in turn. */
    ASORA_OPT_PIPELINED_COPIES = 10,
    /* A rate that is exactly +0 -- a thick cell whose optical depth lies beyond the last table entry, where both lookups
     * return the same value -- need not be added: the grid is bit-identical without it.  The kernels whose rate atomics go
     * through buffer descriptors (table rates, shells in LDS, N <= 512: the production path) exist in a second form that gives
     * such a lane the out-of-range offset of a lane without a rate, so that its atomic never leaves the wave, and in which a
     * wave with nothing to add skips the rate arithmetic; it costs 5 % where no such cell exists and saves a quarter of the
     * launch where two thirds of the pairs lie beyond the table (the benchmark medium at r_RT = 64: -24 %).
     * 0 (default): the library decides -- it takes that form while its probes (the first launch, every 64th after it, sooner
     *    after an upload; read back without waiting) find more than 15 % of the rated pairs dark.
     * 1: always; kernels without buffer atomics take round 2's variant that tests and branches.
     * 2: never: every rated cell is looked up and added, as the reference does (rates.cu:16-41, raytracing.cu:328). */
    ASORA_OPT_SKIP_ZERO_RATES = 11,
    /* 1: the rate atomics are global_atomic_add_f64 under `if (lane has a rate)` even where the grids are small enough for
     *    the default, buffer_atomic_add_f64 through a descriptor over [phi | phi_t] with out-of-range offsets for lanes
     *    without a rate (N <= 512).  Same arithmetic; for A/B runs and the parity test of the two forms. */
    ASORA_OPT_GLOBAL_ATOMICS = 12,
    /* raytrace: one workgroup sweeps its unit for TWO consecutive sources at once (the source-independent geometry is
     * decoded once per lane-step, two dependency chains per wave).  0 = the library decides from the radius and the
     * number of sources (default), 1 = never, 2 = whenever the variant exists (table rates, no heating, shell buffers
     * in LDS, buffer atomics).  Same rates either way, up to the order in which the atomics add them up.  Since round 4
     * the option also governs the tabulated sub-box sweep (ASORA_OPT_SUBBOX_TABLES): photon loss, trailing shell and
     * activity are kept per source, a source that stopped growing is swept along with its partner. */
    ASORA_OPT_PAIR_SOURCES = 13,
    /* c2ray_do_all_sources / asora_subbox_raytrace_device: sweep the sub-boxes on the tabulated geometry of the ASORA
     * kernel (cells within R_max_LLS only: what is rated or lost) instead of generating the cube's geometry on the fly.
     * 0 = the library decides (radius inside the traversal range, enough sources), 1 = never, 2 = whenever the variant
     * exists (table rates, N <= 512).  The source whose column densities are returned always takes the on-the-fly kernel. */
    ASORA_OPT_SUBBOX_TABLES = 14,
    /* 15: rows of the rate grid cut at 64-byte lines: 0 = the library decides (units of one face, i.e. six sectors or twelve
     * sector pairs per source, mesh a multiple of 8, r_RT < 52.5 cells, and a radius that does not change from call to call: after the first change only
     * once a radius has served 32 CALLS in a row -- decided once per call, never per launch), 1 = never, 2 = whenever possible (r_RT <= 110).  The geometry
     * tables then exist in eight forms, by the source's position modulo 8 along the axis that is contiguous in memory for
     * the unit's face; in each, the cells of one row that fall into one 64-byte line of the rate grid never straddle two
     * waves, so every wave's rate atomics leave as whole-line requests (about 8 % fewer requests; the memory side's
     * request rate is what bounds the trace).  Two sources share a workgroup only when they agree modulo 8.  Results are
     * the same sums in a different order. */
    ASORA_OPT_ALIGNED_ROWS = 15,
    /* 1: the geometry tables of the raytrace are built by the host-side builder (geometry.hip: the statement of what a table
     * holds, and the checker of the device-side builder, geometry_device.hip, which is what runs by default and produces the
     * same tables bit for bit -- tests/test_gpu_geometry.py) */
    ASORA_OPT_GEOMETRY_ON_HOST = 16,
    /* How many allocations of the grid arena asora_device_init may try before it keeps the one on which a kernel with the fused
     * pass's stream mix runs fastest (api.hip choose_arena; set BEFORE device_init): 0 (default) = up to 8, 1 = take the first
     * allocation (memory-tight or shared-GPU runs), n = up to n (at most 32).  Whatever the value, what is held during the probe
     * stays within an eighth of the free device memory, and the probe stops once it holds a placement 7 % faster than another.
     * Same results on any placement. */
    ASORA_OPT_PLACEMENT_CANDIDATES = 17,
    ASORA_OPT_COUNT = 18
};
int asora_set_option(int option, int value);
int asora_get_option(int option);

/* Kernel selectors for asora_kernel_time_ms */
enum { ASORA_KERNEL_RAYTRACE = 0, ASORA_KERNEL_CHEMISTRY = 1, ASORA_KERNEL_PREP = 2,
       ASORA_KERNEL_FINISH = 3, ASORA_KERNEL_COUNT = 4 };
/* Sum of HIP-event durations (ms) and number of launches of a kernel since the last reset,
 * measured on the library's stream (ASORA_OPT_TIMING must be 1). */
int asora_kernel_time_ms(int kernel, double *total_ms, long *launches);
int asora_kernel_time_reset(void);
/* hipDeviceSynchronize on the library's device. */
int asora_synchronize(void);

/* Work accounting of the last raytrace: (source,cell) pairs that received a rate
 * (|d|<=R inside the periodic window: the Gamma-contributing set of raytracing.cu:315), and
 * (source,cell) column-density evaluations actually performed (incl. octant-boundary planes). */
int asora_last_raytrace_counts(long long *gamma_cells, long long *evaluated_cells);
/* The same, and how many of the rate-receiving pairs got a rate of exactly +0 that was therefore not added to the grid
 * (ASORA_OPT_SKIP_ZERO_RATES; they are part of gamma_cells). */
int asora_last_raytrace_counts_ex(long long *gamma_cells, long long *evaluated_cells, long long *zero_rates_left_out);

/* Outgoing column density of ONE source over the cells its trace covers, written into a host
 * N^3 grid (zero elsewhere), C-order.  For parity tests of the column density
 * (the reference keeps it in cdh_dev, src/asora/memory.cu:20, and never downloads it). */
int asora_debug_coldens(double R, double sig, double dr, int source_index, double *coldens_out, int N);

/* Which form of the raytrace kernel the last launch took (measurement and tests; no reference counterpart): a bit set of
 * ASORA_VARIANT_* | workgroups per source << 8 | threads per workgroup << 16.  0 before the first launch. */
enum {
    ASORA_VARIANT_PAIRED = 1,             /* two sources per workgroup */
    ASORA_VARIANT_ALIGNED = 2,            /* line-aligned geometry tables (eight forms by source position) */
    ASORA_VARIANT_BUFFER_ATOMICS = 4,     /* rate atomics through buffer descriptors (else global_atomic_add_f64 under a branch) */
    ASORA_VARIANT_SPLIT_DESCRIPTORS = 8,  /* N > 512: one descriptor per layout of the rate grid */
    ASORA_VARIANT_SKIP_ZERO = 16,         /* the form that leaves exact-zero rates out */
    ASORA_VARIANT_GLOBAL_SHELLS = 32      /* shell buffers in global memory (they exceed LDS) */
};
int asora_last_raytrace_variant(void);

/* The geometry tables the last raytrace launch used (tests: device-built against host-built tables).  *ntables = tables of the
 * launch shape (units x 8 when line-aligned); for 0 <= table < *ntables: `words` receives 8 uint32 per entry (the 16-byte
 * cell-A and cell-B records, raytrace.hip) for up to capacity_entries entries, *entries the table's entry count, *nsteps its
 * steps, *shells / *max_cells the launch's shell count and zero slot.  table = -1 only reports the counts. */
/* Device memory the geometry tables of the last launch shape occupy (a part shared between tables counts once). */
size_t asora_debug_geometry_bytes(void);
/* How device_init placed the grids (api.hip choose_arena): allocations tried, probe time of the one kept and of the slowest. */
void asora_debug_placement(int *candidates, double *chosen_probe_ms, double *slowest_probe_ms);
/* Host wall clock (ms) of the last asora_device_init and, inside it, of the placement probe (0 when the first allocation was taken). */
void asora_debug_init_cost(double *device_init_ms, double *placement_probe_ms);
int asora_debug_geometry_table(int table, uint32_t *words, size_t capacity_entries, size_t *entries, int *nsteps, int *ntables,
                               int *shells, int *max_cells, int *threads);

/* Which build this is: a hash over every source and header of the library and the compiler flags (pyc2ray_amd/csrc/Makefile),
 * and those flags.  Measurement hygiene, no reference counterpart: the committed counter summaries under profiles/ name the
 * build they were collected on, and bench.py reports `traffic` only from a summary whose id equals this. */
const char *asora_build_id(void);
const char *asora_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif /* ASORA_HIP_H */
