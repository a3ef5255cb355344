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
};

// Parameters of one raytrace launch (raytrace.hip)
struct RtParams {
    int N;
    int S;                 // last Chebyshev shell over all octants
    int max_cells;         // largest shell; slot max_cells of a shell buffer holds 0.0
    double R;
    double sig, dr;
    double minlogtau, dlogtau, numtau_f;
    double lut_k1, lut_k0; // table index = 1 + (log10 tau - minlogtau)/dlogtau = lut_k1*log2(tau) + lut_k0
    int NumTau, table_len;
    int fortran_consts, grey, z_transposed;
    int src_begin, src_count;
    int shape_src_count;   // source count the launch shape is chosen for (the whole call's, not a pipelined range's)
    int ablate;            // diagnostics only (env ASORA_ABLATE): 1 = no rate atomics, 2 = no rates
    OctGeomDev geom[24];        // by value: pointers read from the kernarg segment are known-global to the compiler
    int units;                  // workgroups per source: 8 octants, 24 octant-sectors, or 12 mirrored sector pairs
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

    // raytracing geometry tables (built once per (N, R, dr), see raytrace.hip)
    std::vector<void *> geom_owned;
    OctGeomDev geom_host[24];               // device pointers of the unit tables
    int geom_units = 0;
    double2 *logtab_dev = nullptr;          // log2 table (ensure_logtab), lives until the runtime is torn down
    bool geom_valid = false;
    bool geom_dr_matters = true;
    int geom_N = 0, geom_S = 0, geom_max_cells = 0, geom_threads = 0;
    double geom_R = 0.0, geom_dr = 0.0;

    // a raytrace call in progress (asora_raytrace_begin ... _range ... _fold)
    RtParams rt_params;
    bool rt_open = false, rt_heat = false;
    // pipelined calls put consecutive source ranges on two side streams, so that the tail of one range and the
    // head of the next overlap; folds (main stream) wait for the ranges issued before them
    bool rt_pipelined = false;
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t side_done[2] = {nullptr, nullptr}, main_ready = nullptr;
    bool side_pending[2] = {false, false};
    int side_next = 0;

    // per-source bookkeeping of the sub-box raytracer (subbox.hip), sized for the largest batch so far
    int *sb_active = nullptr, *sb_nbox = nullptr, *sb_nactive = nullptr;
    double *sb_loss = nullptr, *sb_loss_final = nullptr;
    size_t subbox_cap = 0;

    // shell scratch for traces whose shell buffers exceed LDS
    double *shell_scratch = nullptr;
    size_t shell_scratch_bytes = 0;

    // chemistry reductions
    double *red_partial = nullptr; // [3][red_blocks]
    double *red_final = nullptr;   // [3]
    double *red_host = nullptr;    // pinned [3]
    int red_blocks = 0;

    unsigned long long *counters = nullptr; // [2] device: gamma cells, evaluated cells
    long long last_gamma_cells = 0, last_eval_cells = 0;

    hipStream_t stream = nullptr;
    struct PendingTimer { int which; hipEvent_t e0, e1; };
    std::vector<PendingTimer> pending_timers;     // recorded, not yet resolved
    std::vector<hipEvent_t> free_events;
    int opt[ASORA_OPT_COUNT] = {0, 0, 0, 1, 0, 0, 0, 0};
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

void release_geometry(State &st);
int launch_fold_range(State &st, const double *src_t, double *dst, int i_begin, int i_count);   // dst[i][j][k] += src_t[k][j][i], i in the range
int ensure_logtab(State &st);
int launch_prepare_nhi(State &st, bool need_transposed);
int launch_finish_phi(State &st);
int launch_raytrace(State &st, RtParams &p, bool dump, bool heat, hipStream_t side = nullptr);   // side: stream to launch on when no shared scratch is needed
int launch_fold_transposed(State &st, const double *src_t, double *dst);   // dst[i][j][k] += src_t[k][j][i]
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
    int grey, heat;
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
int launch_subbox_sweep(State &st, const SubboxParams &p);
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

} // namespace asora
