// Internal declarations shared by the translation units of libasora_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/asora_hip.h"

namespace asora {

// Source-independent geometry of one octant, tabulated on the host (raytrace.hip), device pointers.
// A flat sequence of steps of workgroup-size entries; see the table description above the kernel.
struct OctGeomDev {
    const uint4 *cellA;          // { |di| | |dj|<<10 | |dk|<<20 | face<<30, own slot | flags, path (double) }
    const uint4 *cellB;          // shell-buffer slots of the four upstream corners in shell s-1
    int nsteps;
    int info;                    // sign bits of the unit | (merge axis + 1) << 3 | rates-the-source-cell << 5
    int inner;                   // tables that share their inner shells (geometry_device.hip): steps [0, inner >> 8) are read through
                                 // cellA / cellB of table (inner & 255) of the same array, the rest through this table's own
                                 // pointers (which are shifted: entry e of the table is cellA[e] on either side).  0: all its own
    int reserved_;
};

constexpr int MAX_UNITS = 96;   // workgroups per source: 8 octants, 12 mirrored sector pairs, 24 sectors or 96 sector wedges

// Parameters of one raytrace launch (raytrace.hip)
// The work counters of a trace (rated pairs, evaluated cells): every wave adds its share when it ends.  On TWO addresses those
// adds serialise in the memory-side atomic unit at ~12 ns each and set a floor of 25 ns per workgroup under the whole
// launch (0.2 ms for 8000 workgroups, whatever the radius); spread over COUNTER_SLOTS addresses by workgroup index they cost
// nothing, and asora_last_raytrace_counts sums the slots.
constexpr int COUNTER_SLOTS = 4096;
constexpr int ZERO_PROBE_SLOTS = 256;   // RtParams.zero_probe: {rated pairs, exact zeros left out} of ONE launch, spread over these slots
constexpr int COUNTER_FIELDS = 3;   // per slot: rated pairs, evaluated cells, rated pairs whose rate was exactly +0 and was not added

// Upper limit of the real table index in photo_lookuptable: min(float(NumTau), ...) of rates.cu:79 / real(NumTau) of
// photorates.f90:141, and never beyond the last element the device table holds (the reference reads one past the end when the
// caller passes NumTau = len(table), asora_core.py:54; index table_len - 1 is the pair {T[last], 0}).  With the limit applied to
// the REAL index the integer index needs no clamp of its own (rates_device.hpp, lookup_issue).
inline double lut_index_limit(int NumTau, int table_len)
{
    const double by_reference = (double)(float)NumTau, by_table = (double)(table_len > 0 ? table_len - 1 : 0);
    return by_reference < by_table ? by_reference : by_table;
}

struct RtParams {
    int N;
    int S;                 // last Chebyshev shell over all octants
    int max_cells;         // largest shell; slot max_cells of a shell buffer holds 0.0
    double R;
    double sig, dr;
    double minlogtau, dlogtau, numtau_f;
    double lut_k1, lut_k0; // table index = 1 + (log10 tau - minlogtau)/dlogtau = lut_k1*log2(tau) + lut_k0
    double tau_zero;       // ASORA_OPT_SKIP_ZERO_RATES: thick cells with tau_in >= tau_zero get exactly +0 (both lookups clamp) and are not added; +inf = add everything
    int NumTau, table_len;
    int fortran_consts, grey, z_transposed;
    int src_begin, src_count;
    int shape_src_count;   // source count the launch shape is chosen for (the whole call's, not a pipelined range's)
    int split_desc;        // 1: N > 512 -- the buffer descriptors of the rate atomics span ONE layout of the grid each (raytrace.hip)
    int radius_stays;      // decided once per call (note_call_radius): the line-aligned tables may be built for this radius
    int ablate;            // diagnostics only (env ASORA_ABLATE): 1 = no rate atomics, 2 = no rates
    OctGeomDev geom[MAX_UNITS]; // by value: pointers read from the kernarg segment are known-global to the compiler
    int units;                  // workgroups per source: 8 octants, 24 octant-sectors, 12 mirrored sector pairs, 96 sector wedges
    int spread;                 // 1: block b = (source b / units, unit b % units) -- a source's units on different XCDs
    const double2 *logtab;      // 128 x {1/c, log2 c}
    unsigned ncell;             // N^3: the [k][j][i] copy of a grid starts ncell elements after its [i][j][k] form
    const double *nhi;          // nHI, [i][j][k] then [k][j][i]
    double *phi;                // Gamma accumulator, [i][j][k] then [k][j][i]
    const double2 *tables;      // pairs {T[i], T[i+1]-T[i]}: thick at [0, len), thin at [len, 2 len), heat thick at [2 len, 3 len), heat thin at [3 len, 4 len)
    double *heat;               // heating accumulator (HEAT kernels), [i][j][k] then [k][j][i]
    const int32_t *src_pos;
    const double *src_flux;
    double *dump;               // debug: outgoing column density (N^3) or nullptr
    double *shell_scratch;      // global shell buffers when they do not fit LDS, else nullptr
    unsigned long long *counters;
    const int *done_flag;       // evolve loop: device flag "the step has converged" -> the launch does nothing; or nullptr
    unsigned long long *zero_probe;   // SKIP_ZERO kernels: where a probe launch sums what it rated and what it left out, or nullptr
    // ---- rows cut at 64-byte lines (ASORA_OPT_ALIGNED_ROWS; launch_raytrace) ----
    int aligned;                // 1: geom[] holds 8 x units tables, [class * units + unit], class = source position & 7 along the
                                //    memory-contiguous axis of the unit's face (k for the x- and y-sectors, i for the z-sector)
    const int2 *pairs[2];       // NSRC = 2 && aligned: the workgroup's two sources (indices into src_pos; .y < 0: only one) for
    int npairs[2];              //    [0] units of the x- / y-sector (equal k & 7), [1] units of the z-sector (equal i & 7)
    // ---- SUBBOX kernels only (the reference's CPU semantics on the tabulated geometry, raytrace.hip / subbox.hip) ----
    int sb_k0[12], sb_k1[12];   // per unit: this launch sweeps the table steps [k0, k1) = the shells of one sub-box (multiples of 3)
    int sb_first;               // 1: the launch starts at the source cell; 0: it continues from the trailing shell in sb_trail
    int sb_edge_r, sb_edge_l;   // faces of the current sub-box on the + / - side of every axis (raytracing.f90:199-200)
    int flux_src;               // >= 0: every source shines with the flux of this one (f90:500,503); -1: its own
    const int *sb_active;       // per source of the batch: still growing?
    double *sb_loss;            // per source of the batch: photons through the faces of the current sub-box
    double *sb_trail;           // per (source, unit): the last shell swept, handed to the next sub-box's launch
};

// Device-side bookkeeping of the evolve loop (asora_evolve_begin / _enqueue / _poll): the convergence test of
// pyc2ray/evolve.py:216-236 is evaluated on the device right behind the chemistry reductions, so that several outer
// iterations can be enqueued at once; launches that come after convergence see `done` and do nothing.
constexpr int EVOLVE_HIST = 64;
struct EvolveStatus {
    int niter;                         // outer iterations carried out
    int done;                          // 1 once the test of evolve.py:231-232 has passed
    double prev1, prev0;               // sum(xh_intermed), sum(1 - xh_intermed) of the previous iteration (evolve.py:130-131,234-235)
    double conv_criterion, conv_fraction;
    double hist[EVOLVE_HIST][5];       // ring by iteration: conv_flag, sum1, sum0, rel_change_xh1, rel_change_xh0
};

// ---------------------------------------------------------------------------------------------
// Process-global device state (the role of src/asora/memory.cu:20-29)
// ---------------------------------------------------------------------------------------------
struct State {
    bool init = false;
    bool auto_init = false;        // initialised by c2ray_do_all_sources on its own (may re-initialise for another N)
    int device = 0;
    int N = 0;
    size_t ncell = 0;
    int num_src_par = 0;           // accepted, unused (no per-source N^3 scratch in this build)
    int cu_count = 256;

    double *grid[ASORA_GRID_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool grid_valid[ASORA_GRID_COUNT] = {false, false, false, false, false, false, false};

    // derived per raytrace call
    // each of these is the second half of a 2 N^3 allocation whose first half is the [i][j][k] grid
    double *nhi = nullptr;     // ndens*(1-xh_av), [i][j][k] (owns the allocation)
    double *nhi_t = nullptr;   // = nhi + N^3: same, transposed [k][j][i]
    double *phi_t = nullptr;   // = grid[PHI_ION] + N^3: rate accumulator for z-faces, transposed [k][j][i]
    double *heat_t = nullptr;  // = grid[PHI_HEAT] + N^3: heating-rate accumulator for z-faces, transposed
    bool have_heat_tables = false;
    double *staging = nullptr; // N^3 staging grid for 'F'-order transfers / debug dumps

    double2 *tables = nullptr;     // [thick | thin | heat thick | heat thin] as pairs {T[i], T[i+1]-T[i]}, each table_len long
    int table_len = 0;

    int32_t *src_pos = nullptr;
    double *src_flux = nullptr;
    int num_src = 0;
    // the same sources ordered by their first coordinate (pipelined asora_do_all_sources)
    int32_t *src_pos_sorted = nullptr;
    double *src_flux_sorted = nullptr;
    std::vector<int> src_i0_sorted;
    std::vector<hipEvent_t> pipe_events;

    // raytracing geometry tables (built once per (N, R, dr), see raytrace.hip)
    std::vector<void *> geom_owned;
    size_t geom_bytes = 0;                  // device memory the current tables occupy (shells shared by several tables count once)
    OctGeomDev geom_host[MAX_UNITS];        // device pointers of the unit tables ([class * units + unit] when geom_aligned)
    bool geom_aligned = false;
    // how the radius has behaved across raytrace launches (launch_raytrace: the eight-fold tables only pay when they are reused)
    // exact-zero rates (ASORA_OPT_SKIP_ZERO_RATES = 0): launch_raytrace takes the kernels that leave them out while its probes
    // find enough of them
    unsigned long long *zero_probe_dev = nullptr;    // [2 * ZERO_PROBE_SLOTS]
    unsigned long long *zero_probe_host = nullptr;   // pinned: {rated, left out} of the last probe
    hipEvent_t zero_probe_done = nullptr;
    bool zero_probe_pending = false, zero_known = false, zero_dark = false;
    int zero_since_probe = 0;
    int last_variant = 0;                   // ASORA_VARIANT_* bits | units << 8 | threads << 16 of the last raytrace launch
    struct RadiusHistory { double last_R = -1.0; long same_R_calls = 0; bool has_changed = false; };
    RadiusHistory rt_radius[2];             // [0]: whole-box traces and the evolve loop, [1]: the sub-box sweep (note_call_radius)
    // host copies of the two source lists (as uploaded, and in lexicographic order of the position) and, for the paired-sources
    // variant on aligned tables, who shares a workgroup with whom: built once per (list, range), see source_pairs_by_class
    std::vector<int32_t> src_pos_host, src_pos_sorted_host;
    struct PairList { const void *list; int begin, count; int2 *dev[2]; int n[2]; };
    std::vector<PairList> pair_lists;      // (a sharded or chunked trace asks for the same few ranges every iteration)
    int geom_units = 0;
    double2 *logtab_dev = nullptr;          // log2 table (ensure_logtab), lives until the runtime is torn down
    bool geom_valid = false;
    bool geom_on_host = false;              // the cached tables were built by the host builder (ASORA_OPT_GEOMETRY_ON_HOST)
    // table entries of cells that sit (to rounding) exactly ON the sphere: whether such a cell gets a rate is decided by
    // the reference's floating-point distance test, whose outcome depends on dr.  They are always tabulated (evaluated);
    // their RATE bit is re-decided in place when dr changes (a cosmological run: every time step) -- no rebuild.
    struct SphereCell { uint32_t *dev_word; uint32_t word_without_rate; int a, b, c; };
    std::vector<SphereCell> geom_sphere;
    void *geom_patch_dev = nullptr;         // staging for the patch kernel: {address, value} pairs
    size_t geom_patch_cap = 0;
    int geom_N = 0, geom_S = 0, geom_max_cells = 0, geom_threads = 0;
    // sub-box tables (subbox_tables_prepare): traversal range instead of the periodic window, no octahedron bound, every
    // sub-box boundary padded to whole triples of steps; step_after_shell[u][s] = first table step behind shell s of unit u
    int geom_subbox = 0, geom_ext_r = 0, geom_ext_l = 0, geom_boxsize = 0;
    std::vector<int> geom_step_after_shell[12];
    double geom_R = 0.0, geom_dr = 0.0;

    // a raytrace call in progress (asora_raytrace_begin ... _range ... _fold)
    RtParams rt_params;
    bool rt_open = false, rt_heat = false;
    // pipelined calls put consecutive source ranges on two side streams, so that the tail of one range and the
    // head of the next overlap; folds (main stream) wait for the ranges issued before them
    bool rt_pipelined = false;
    bool rt_by_planes = false;              // begun by asora_raytrace_begin_planes: ranges share one launch shape
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t side_done[2] = {nullptr, nullptr}, main_ready = nullptr;
    bool side_pending[2] = {false, false};
    int side_next = 0;

    // per-source bookkeeping of the sub-box raytracer (subbox.hip), sized for the largest batch so far
    int *sb_active = nullptr, *sb_nbox = nullptr, *sb_nactive = nullptr;
    double *sb_loss = nullptr, *sb_loss_final = nullptr;
    size_t subbox_cap = 0;

    double *sb_trail = nullptr;             // per (source, unit): the trailing shell between two sub-box launches (table path)
    size_t sb_trail_bytes = 0;

    // shell scratch for traces whose shell buffers exceed LDS
    double *shell_scratch = nullptr;
    size_t shell_scratch_bytes = 0;

    // chemistry reductions
    double *red_partial = nullptr; // [3][red_cap]: per-workgroup partial sums (red_cap >= red_blocks)
    size_t red_cap = 0;
    double *red_final = nullptr;   // [3]
    double *red_host = nullptr;    // pinned [3]
    int red_blocks = 0;

    unsigned long long *counters = nullptr; // [COUNTER_FIELDS * COUNTER_SLOTS] device: gamma cells, evaluated cells, exact zeros left out, spread over the slots
    long long last_gamma_cells = 0, last_eval_cells = 0;

    // fused evolve loop (asora_evolve_*): raytrace accumulators of their own ([i][j][k] then [k][j][i]; the chemistry
    // kernel folds them into PHI_ION and zeroes them), device-side convergence bookkeeping
    // TWO sets of accumulators, each [i][j][k] then [k][j][i] (4 N^3 doubles, allocated on first use).  Iteration k of a
    // time step traces into set (ev_base + k - 1) & 1; its fused pass reads that set WITHOUT destroying it and zeroes the
    // other one for iteration k + 1.  The rates of the last iteration carried out therefore survive in their set until the
    // host asks for them (asora_evolve_poll folds them into PHI_ION): the pass writes no rate grid (88 instead of 96 B per cell).
    double *acc = nullptr;
    // every N^3 grid of the hot loop lives in ONE allocation (api.hip: choose_arena), picked among several candidates by a probe
    char *arena = nullptr;
    size_t arena_bytes = 0;
    int arena_candidates = 0;               // how many were tried
    double arena_probe_ms = 0.0, arena_probe_worst_ms = 0.0;     // the chosen one's probe time, the slowest candidate's
    double arena_probe_wall_ms = 0.0, device_init_wall_ms = 0.0; // host wall clock of the whole probe / of the last device_init
    int ev_base = 0;                        // set of the first iteration of the current time step
    bool ev_clean[2] = {false, false};      // set known to be all zero (when nothing is enqueued)
    bool ev_sets_known = false;             // the bookkeeping above is valid (a poll has happened since the last enqueue)
    int ev_folded_iter = -1;                // PHI_ION holds the rates of this iteration of the current step (0: none yet)
    EvolveStatus *ev_status = nullptr;      // device
    EvolveStatus *ev_host = nullptr;        // pinned
    bool ev_open = false, ev_first = true;
    // temperature probe of the TEMP grid: valid for the upload `temp_generation` and the constants in temp_consts
    double *temp_probe_dev = nullptr;       // [8] device: the probe's five results, [5] = asora_grid_sum's
    double temp_probe[5] = {0, 0, 0, 0, 0}; // host copy: uniform?, T, brech0, acolh0, t_ok
    bool temp_probe_valid = false;
    double temp_consts[4] = {0, 0, 0, 0};
    int ev_reported = 0;                    // iterations already handed to the caller by asora_evolve_poll
    int ev_enqueued = 0;                    // upper bound of the iterations carried out (enqueued) in this step
    double ev_chem[6] = {0, 0, 0, 0, 0, 0}; // dt, bh00, albpow, colh0, temph0, abu_c
    int ev_src_begin = 0, ev_src_count = 0;
    RtParams ev_rt;
    // multi-GPU form of the loop (asora_evolve_begin_slab): this rank runs the fused pass on the planes it owns only
    bool ev_slab = false;
    int ev_own_begin = 0, ev_own_count = 0;
    bool ev_slab_passed = false;            // the current iteration's pass has been enqueued (close comes next)
    bool ev_folded_all = false;             // ... and this iteration's has been enqueued
    bool ev_rates_in_outbox = false;        // asora_evolve_slab_fold_all: the passes of this step read the out-box and keep PHI_ION
    // Which 64-byte lines of the rate accumulators the sources of the current step can touch at all (round 4): one byte per
    // line of 8 cells, [i][j][k >> 3] for the plain layout and, behind it, [k][j][i >> 3] for the transposed one.  The fused
    // pass neither reads nor zeroes the lines no source reaches (they are zero and stay zero): 32 of its 88 bytes per cell.
    // Valid for (source upload, source range, R); nullptr in the pass = every line is reachable.
    unsigned char *reach_mask = nullptr;
    size_t reach_bytes = 0;                 // of ONE layout
    bool reach_valid = false, reach_in_use = false;
    bool reach_pays = false;                // enough lines out of reach for the mask to pay (counted when the mask is built)
    unsigned long long *reach_count_dev = nullptr;
    long reach_src_generation = -1;
    int reach_src_begin = 0, reach_src_count = 0;
    double reach_R = -1.0;
    long src_generation = 0;                // counts asora_source_data_to_device calls

    hipStream_t stream = nullptr;
    struct PendingTimer { int which; hipEvent_t e0, e1; };
    std::vector<PendingTimer> pending_timers;     // recorded, not yet resolved
    std::vector<hipEvent_t> free_events;
    int opt[ASORA_OPT_COUNT] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0};
    double k_ms[ASORA_KERNEL_COUNT] = {0, 0, 0, 0};
    long k_n[ASORA_KERNEL_COUNT] = {0, 0, 0, 0};
};

State &state();
int fail(int code, const std::string &msg);   // records the message, returns code
void clear_error();

#define ASORA_HIP_TRY(expr)                                                                     \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ::asora::fail(10, std::string(#expr) + ": " + hipGetErrorName(e__) + " - " + \
                                         hipGetErrorString(e__));                               \
    } while (0)

// Scoped HIP-event timer around kernel launches on the library stream.
struct KernelTimer {
    int which;
    bool on;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit KernelTimer(int w, hipStream_t s = nullptr);   // s = nullptr: the library's main stream
    ~KernelTimer();
};

// ---------------------------------------------------------------------------------------------
// Raytracing (raytrace.hip)
// ---------------------------------------------------------------------------------------------

// flag bits of a geometry-table entry (cellA.y; raytrace.hip describes the tables, geometry.hip builds them)
constexpr unsigned CELL_VALID = 1u << 30, CELL_LAST = 1u << 31, CELL_RATE = 1u << 29,
                   CELL_SPHERE = 1u << 28,            // host bookkeeping: the cell sits on the sphere, RATE depends on dr (the kernel ignores it)
                   CELL_ZERO_SHIFT = 25,              // bits 25..27: which of the offsets (a, b, c) are zero
                   CELL_NEG_SHIFT = 22,               // bits 22..24: the cell lies on the mirrored side of axis 0 / 1 / 2 (axes the unit merges)
                   CELL_SLOT_MASK = (1u << 22) - 1;

// What one geometry table holds (geometry.hip: UnitSpec + the alignment class): a dependency-closed part of the sphere
struct GeomTableSpec {
    int face = -1;            // -1: whole octant(s); 0 / 1 / 2: the x- / y- / z-sector
    int merge_mask = 0;       // bit ax: the unit covers BOTH signs of axis ax
    int ext[3] = {0, 0, 0};   // periodic-window extent of each axis on the side the unit looks at
    int ext_neg = 0;          // extent on the mirrored side of a merged axis
    int wedge = -1;           // 0..3: a quarter of the sector, -1: the whole unit
    int align_class = -1;     // 0..7: line-aligned form for sources at this position modulo 8; -1: densely packed
};
// geometry_device.hip: the DISTINCT tables of a launch shape built on the GPU (bit-identical to geometry.hip's host builder)
int build_geometry_on_device(State &st, const std::vector<GeomTableSpec> &specs, double R, double dr, int q_max, int threads,
                             int boxsize, std::vector<OctGeomDev> &out, std::vector<std::vector<int>> &step_after, int &S_all,
                             uint32_t &max_cells_all);
struct SubboxGeometry { int ext_r, ext_l, boxsize; };     // sub-box tables: the traversal range of raytracing.f90:174-175, the box size
// Build (or reuse) the geometry tables for (N, R, dr, threads, units): geometry.hip
int ensure_geometry(State &st, RtParams &p, int threads, int units, const SubboxGeometry *sbg = nullptr, bool aligned = false);
void release_geometry(State &st);
bool note_call_radius(State &st, double R, int path);   // once per API call: may the eight-fold aligned tables be built for this radius?
void release_pair_lists(State &st);     // with every change of the source lists
int launch_fold_range(State &st, const double *src_t, double *dst, int i_begin, int i_count);   // dst[i][j][k] += src_t[k][j][i], i in the range
int ensure_logtab(State &st);
int launch_prepare_nhi(State &st, bool need_transposed);
int launch_finish_phi(State &st);
int launch_raytrace(State &st, RtParams &p, bool dump, bool heat, hipStream_t side = nullptr);   // side: stream to launch on when no shared scratch is needed
// The sub-box sweep on tabulated geometry (raytrace.hip): prepare -> tables for (N, R, range, box size), then one launch per
// sub-box.  `shells` = (s_begin, s_end] of the box; returns without launching when the tables hold nothing there.
struct SubboxTables { int units = 0, threads = 0, S = 0, max_batch = 0, nsrc = 1; bool ok = false, aligned = false; const int32_t *host_pos = nullptr; };   // max_batch: sources per launch the trailing-shell scratch holds; nsrc: sources per workgroup (2: paired sweep); host_pos: the source positions on the host (0-based, xyz-interleaved; pairing for the aligned tables) or nullptr
int subbox_tables_prepare(State &st, RtParams &p, int ext_r, int ext_l, int subboxsize, int src_count, bool heat, SubboxTables &out,
                          const int32_t *host_pos);
int subbox_tables_sweep(State &st, const RtParams &p, const SubboxTables &tab, int s_begin, int s_end, bool heat);
int launch_fold_transposed(State &st, const double *src_t, double *dst);   // dst[i][j][k] += src_t[k][j][i]
int launch_fold_sum(State &st, const double *a, const double *b_t, double *dst);   // dst[i][j][k] = a[i][j][k] + b_t[k][j][i]
int launch_transpose(State &st, const double *src, double *dst, int N);   // dst[k][j][i] = src[i][j][k]

// ---------------------------------------------------------------------------------------------
// Raytracing with the reference's CPU semantics: cubic sub-boxes, photon loss (subbox.hip)
// ---------------------------------------------------------------------------------------------
struct SubboxParams {
    int N, W;                   // mesh size; row pitch of the shell-buffer slot layout (largest shell + 1)
    int ext_r, ext_l;           // traversal range on the + / - side of every axis (raytracing.f90:174-175)
    int s_begin, s_end;         // this launch sweeps the Chebyshev shells (s_begin, s_end]
    int edge_r, edge_l;         // faces of the current sub-box: last_r - src, src - last_l (f90:199-200)
    double sig, dr, R;          // R = R_max_LLS in cells
    double numtau_f, lut_k1, lut_k0;
    int table_len, ablate;
    int grey, heat, add_zero;   // add_zero = 0: ASORA_OPT_SKIP_ZERO_RATES
    int src_begin, src_count;   // batch of sources
    int flux_src;               // >= 0: every source shines with the flux of this one (f90:500,503); -1: its own
    int dump_src;               // source whose column densities are returned (the last one), or -1
    unsigned ncell;
    const double *nhi;          // [i][j][k] then [k][j][i]
    double *phi, *heat_grid;    // same layout
    double *dump;               // [i][j][k]
    const double2 *tables, *logtab;
    const int32_t *src_pos;
    const double *src_flux;
    double *scratch;            // per workgroup: two shell buffers of 3 W^2 doubles
    size_t unit_stride;         // = 6 W^2
    const int *active;          // per source of the batch
    double *loss;               // per source of the batch: photons through the current box faces
};
int launch_subbox_sweep(State &st, const SubboxParams &p, hipStream_t side = nullptr);   // side: the stream to launch on (default: the library's)
int launch_subbox_decide(State &st, int mode, int count, const double *src_flux, int src_begin, double loss_fraction,
                         int more_range, int *active, double *loss, double *loss_final, int *nbox, int *n_active);

// ---------------------------------------------------------------------------------------------
// Chemistry (chemistry.hip)
// ---------------------------------------------------------------------------------------------
struct ChemParams {
    size_t ncell;
    double dt, bh00, albpow, colh0, temph0, abu_c;
    const double *ndens, *temp, *xh, *phi;
    double *xh_av, *xh_intermed;
    double *red_partial, *red_final;
    int red_blocks;
    int accumulate = 0;        // 1: add this launch's reductions to red_final (slab-wise passes) instead of replacing it
};
int launch_chemistry(State &st, ChemParams &p, hipStream_t stream);
int chemistry_reduction_blocks(const State &st);

// The same pass over the planes [i_begin, i_end) of the N^3 grids, tiled so that the [k][j][i] twins can be read and
// written with contiguous rows as well (chemistry.hip: chemistry_tile_kernel).
//   fold   : the rate of a cell is gamma[i][j][k] + gamma_t[k][j][i] (the raytrace's two accumulators), written to phi_out
//   emit   : additionally form nHI = ndens (1 - xh_av) of the NEW xh_av in both layouts for the next raytrace and
//            zero both accumulators -- everything the host did between two iterations (pyc2ray/evolve.py:200-240)
//   status : evaluate the convergence test on the device behind the reductions; launches do nothing once it has passed
struct ChemTileParams {
    int N = 0, i_begin = 0, i_end = 0;
    double dt = 0, bh00 = 0, albpow = 0, colh0 = 0, temph0 = 0, abu_c = 0;
    const double *ndens = nullptr, *temp = nullptr, *xh = nullptr, *xh_av_in = nullptr;
    double *gamma = nullptr, *gamma_t = nullptr;     // rates [i][j][k] (+ [k][j][i] accumulator when fold)
    double *phi_out = nullptr;                       // fold: folded rates go here as well (nullptr: nowhere)
    double *zero_a = nullptr, *zero_t = nullptr;     // emit: the accumulator pair to zero for the next raytrace
    double *xh_av = nullptr, *xh_intermed = nullptr;
    double *nhi = nullptr, *nhi_t = nullptr;         // emit
    double *red_partial = nullptr, *red_final = nullptr;
    int red_stride = 0;                              // number of workgroups of the launch (set by the launcher)
    int accumulate = 0;
    EvolveStatus *status = nullptr;
    bool local_sums = false;                         // status only gates the launch: the reductions stop at red_final, the convergence
                                                     // test follows later, on the sums over all ranks (launch_convergence_test)
    const unsigned char *reach_a = nullptr, *reach_t = nullptr;   // fold + emit: lines of the accumulators any source reaches (State::reach_mask), or nullptr
    bool fold = false, emit = false;
    // the grid has one temperature (launch_temp_probe): its factors, evaluated on the device, travel with the parameters
    int uniform = 0, uniform_t_ok = 0;
    double uniform_T = 0, uniform_brech0 = 0, uniform_acolh0 = 0;
};
int launch_grid_sum(State &st, const double *a, size_t n, double *out_dev);
int launch_scale(State &st, double *a, size_t n, double factor);
int launch_temp_probe(State &st, const double *temp, size_t n, double bh00, double albpow, double colh0, double temph0,
                      double *out_dev);
int launch_chemistry_tiles(State &st, ChemTileParams &p, hipStream_t stream);
size_t chemistry_tile_blocks(const State &st, int N, int planes);
int launch_prepare_nhi_from(State &st, const double *xh_av, bool need_transposed);
int launch_prepare_range(State &st, int i_begin, int i_count, bool zero_acc, double *acc, const int *done = nullptr);
// multi-GPU device loop: foreign planes folded into the out-box (+ the other accumulator pair zeroed there); received planes added
int launch_fold_out(State &st, const double *a, const double *a_t, double *out, double *z_a, double *z_t, int i_begin, int i_count,
                    const int *done);
int launch_add_planes(State &st, double *dst, const double *src, size_t n, const int *done);
// the convergence test of evolve.py:216-236 on sums[3] = {sum x, sum 1-x, conv_flag} (summed over the ranks beforehand)
int launch_convergence_test(State &st, const double *sums, EvolveStatus *status);
// mask[0 .. N*N*NL) for [i][j][k >> 3], mask[N*N*NL .. ) for [k][j][i >> 3], NL = (N + 7) / 8: 1 where a source of the range reaches
int launch_reach_mask(State &st, const int32_t *src_pos, int src_begin, int src_count, double R, unsigned char *mask, size_t bytes_one_layout);
int launch_reach_count(State &st, const unsigned char *mask, size_t bytes_both_layouts, unsigned long long *out_dev);

} // namespace asora
