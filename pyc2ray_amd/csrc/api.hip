// api.hip -- device-state manager and the extern "C" boundary of libasora_hip.so.
// The reference counterparts are src/asora/memory.cu (state) and
// src/asora/python_module.cu (CPython wrappers); see include/asora_hip.h for the mapping.
#include "asora_internal.hpp"
#include "rates_device.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>

namespace asora {

static State g_state;
static std::string g_error;

State &state() { return g_state; }
int fail(int code, const std::string &msg) { g_error = msg; return code; }
void clear_error() { g_error.clear(); }

// Kernel timing with HIP events on the library's stream.  Events are recorded without any host
// synchronisation (so that enabling the timers does not perturb what is being timed) and resolved
// when the totals are queried, or when the pool of pending pairs is full.
static int flush_timers()
{
    State &st = g_state;
    if (st.pending_timers.empty()) return 0;
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    for (hipStream_t s : st.side) if (s) ASORA_HIP_TRY(hipStreamSynchronize(s));
    for (auto &pt : st.pending_timers) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pt.e0, pt.e1) == hipSuccess) {
            st.k_ms[pt.which] += (double)ms;
            st.k_n[pt.which] += 1;
        }
        st.free_events.push_back(pt.e0);
        st.free_events.push_back(pt.e1);
    }
    st.pending_timers.clear();
    return 0;
}

static hipEvent_t take_event()
{
    State &st = g_state;
    if (!st.free_events.empty()) { hipEvent_t e = st.free_events.back(); st.free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

KernelTimer::KernelTimer(int w, hipStream_t s)
    : which(w), on(g_state.opt[ASORA_OPT_TIMING] != 0 && g_state.stream != nullptr), stream(s ? s : g_state.stream)
{
    if (!on) return;
    if (g_state.pending_timers.size() >= 4096) (void)flush_timers();
    e0 = take_event();
    e1 = take_event();
    (void)hipEventRecord(e0, stream);
}
KernelTimer::~KernelTimer()
{
    if (!on) return;
    (void)hipEventRecord(e1, stream);
    g_state.pending_timers.push_back({which, e0, e1});
}

// stream, events and the chemistry reduction buffers: needed with or without device_init
static int ensure_runtime()
{
    State &st = g_state;
    if (st.stream) return 0;
    ASORA_HIP_TRY(hipSetDevice(st.device));
    hipDeviceProp_t prop;
    ASORA_HIP_TRY(hipGetDeviceProperties(&prop, st.device));
    st.cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ASORA_HIP_TRY(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) {
        ASORA_HIP_TRY(hipStreamCreateWithFlags(&st.side[q], hipStreamNonBlocking));
        ASORA_HIP_TRY(hipEventCreateWithFlags(&st.side_done[q], hipEventDisableTiming));
    }
    ASORA_HIP_TRY(hipEventCreateWithFlags(&st.main_ready, hipEventDisableTiming));
    st.red_blocks = chemistry_reduction_blocks(st);
    st.red_cap = 3 * (size_t)st.red_blocks;
    ASORA_HIP_TRY(hipMalloc(&st.red_partial, sizeof(double) * st.red_cap));
    ASORA_HIP_TRY(hipMalloc(&st.red_final, sizeof(double) * 3));
    ASORA_HIP_TRY(hipHostMalloc(&st.red_host, sizeof(double) * 3, hipHostMallocDefault));
    ASORA_HIP_TRY(hipMalloc(&st.counters, sizeof(unsigned long long) * COUNTER_FIELDS * COUNTER_SLOTS));
    ASORA_HIP_TRY(hipMemset(st.counters, 0, sizeof(unsigned long long) * COUNTER_FIELDS * COUNTER_SLOTS));
    return 0;
}

// One temperature for the whole grid?  Probed once per upload of TEMP and set of chemistry constants (one pass over the
// grid + a 40-byte read-back); the tiled chemistry pass then needs neither the temperature loads nor pow/sqrt/exp.
static int ensure_temp_probe(double bh00, double albpow, double colh0, double temph0)
{
    State &st = g_state;
    const double c[4] = {bh00, albpow, colh0, temph0};
    if (st.temp_probe_valid && std::memcmp(c, st.temp_consts, sizeof c) == 0) return 0;
    if (!st.temp_probe_dev) ASORA_HIP_TRY(hipMalloc(&st.temp_probe_dev, sizeof(double) * 8));
    if (int rc = launch_temp_probe(st, st.grid[ASORA_GRID_TEMP], st.ncell, bh00, albpow, colh0, temph0, st.temp_probe_dev)) return rc;
    ASORA_HIP_TRY(hipMemcpyAsync(st.temp_probe, st.temp_probe_dev, sizeof(double) * 5, hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    std::memcpy(st.temp_consts, c, sizeof c);
    st.temp_probe_valid = true;
    return 0;
}
static void set_uniform_temperature(ChemTileParams &p)
{
    State &st = g_state;
    p.uniform = (st.temp_probe_valid && st.temp_probe[0] != 0.0 && !st.opt[ASORA_OPT_NO_UNIFORM_T]) ? 1 : 0;
    p.uniform_T = st.temp_probe[1]; p.uniform_brech0 = st.temp_probe[2]; p.uniform_acolh0 = st.temp_probe[3];
    p.uniform_t_ok = st.temp_probe[4] != 0.0 ? 1 : 0;
}

// per-workgroup partial sums of the chemistry passes: room for `entries` doubles
static int ensure_red_capacity(size_t entries)
{
    State &st = g_state;
    if (entries <= st.red_cap) return 0;
    if (st.stream) ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    if (st.red_partial) { (void)hipFree(st.red_partial); st.red_partial = nullptr; st.red_cap = 0; }
    ASORA_HIP_TRY(hipMalloc(&st.red_partial, sizeof(double) * entries));
    st.red_cap = entries;
    return 0;
}

// The heating-rate grid and its [k][j][i] twin (2 N^3 doubles: 2 GiB at 512^3) exist only once something heats: the heating
// tables are uploaded, or a caller hands the grid over (evolve3D never does)
static int ensure_heat_grid()
{
    State &st = g_state;
    if (st.grid[ASORA_GRID_PHI_HEAT]) return 0;
    ASORA_HIP_TRY(hipMalloc(&st.grid[ASORA_GRID_PHI_HEAT], 2 * st.ncell * sizeof(double)));
    st.heat_t = st.grid[ASORA_GRID_PHI_HEAT] + st.ncell;
    st.grid_valid[ASORA_GRID_PHI_HEAT] = false;
    return 0;
}

static int release_all()
{
    State &st = g_state;
    auto drop = [](auto *&ptr) { if (ptr) { (void)hipFree(ptr); ptr = nullptr; } };
    // (the grids of the hot loop are parts of the arena; the heating grid is an allocation of its own)
    drop(st.grid[ASORA_GRID_PHI_HEAT]);
    for (int g = 0; g < ASORA_GRID_COUNT; ++g) { st.grid[g] = nullptr; st.grid_valid[g] = false; }
    st.nhi = st.staging = st.acc = nullptr;
    drop(st.arena); st.arena_bytes = 0;
    st.ev_clean[0] = st.ev_clean[1] = false; st.ev_sets_known = false;
    drop(st.reach_mask); drop(st.reach_count_dev); st.reach_bytes = 0; st.reach_valid = false; st.reach_in_use = false; st.reach_pays = false;
    st.ev_open = false;
    st.temp_probe_valid = false;
    st.nhi_t = st.phi_t = st.heat_t = nullptr;    // second halves of nhi / phi_ion / phi_heat
    st.have_heat_tables = false;
    drop(st.tables); st.table_len = 0;
    drop(st.src_pos); drop(st.src_flux); drop(st.src_pos_sorted); drop(st.src_flux_sorted); st.src_i0_sorted.clear(); st.num_src = 0;
    st.src_pos_host.clear(); st.src_pos_sorted_host.clear();
    release_pair_lists(st);
    drop(st.shell_scratch); st.shell_scratch_bytes = 0;
    drop(st.sb_trail); st.sb_trail_bytes = 0;
    if (st.geom_patch_dev) { (void)hipFree(st.geom_patch_dev); st.geom_patch_dev = nullptr; st.geom_patch_cap = 0; }
    drop(st.sb_active); drop(st.sb_nbox); drop(st.sb_loss); drop(st.sb_loss_final); st.subbox_cap = 0;
    release_geometry(st);
    st.init = false; st.N = 0; st.ncell = 0;
    st.rt_radius[0] = State::RadiusHistory(); st.rt_radius[1] = State::RadiusHistory();
    if (st.zero_probe_dev) { (void)hipFree(st.zero_probe_dev); st.zero_probe_dev = nullptr; }
    if (st.zero_probe_host) { (void)hipHostFree(st.zero_probe_host); st.zero_probe_host = nullptr; }
    if (st.zero_probe_done) { (void)hipEventDestroy(st.zero_probe_done); st.zero_probe_done = nullptr; }
    st.zero_probe_pending = false; st.zero_known = false; st.zero_dark = false; st.zero_since_probe = 0;
    st.rt_open = false;
    return 0;
}

static int require_init(const char *who)
{
    if (!g_state.init) return fail(2, std::string(who) + ": device not initialised (call asora_device_init first)");
    return 0;
}

static int check_N(const char *who, int N)
{
    if (N != g_state.N)
        return fail(3, std::string(who) + ": mesh size " + std::to_string(N) + " does not match device_init(" +
                           std::to_string(g_state.N) + ")");
    return 0;
}

// The parameter block of a raytrace of the uploaded sources into PHI_ION (+ its [k][j][i] twin).  radius_path: which radius
// history the call belongs to (note_call_radius; exactly one call of it per API call: here) -- 0 the whole-box entry points,
// 1 the sub-box sweep
static void fill_rt_params(RtParams &p, double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau, int radius_path = 0)
{
    State &st = g_state;
    std::memset(&p, 0, sizeof p);
    p.N = st.N;
    p.R = R; p.sig = sig; p.dr = dr;
    p.minlogtau = minlogtau; p.dlogtau = dlogtau;
    p.table_len = st.table_len > 0 ? st.table_len : 1;
    p.NumTau = NumTau; p.numtau_f = lut_index_limit(NumTau, p.table_len);
    p.fortran_consts = st.opt[ASORA_OPT_FORTRAN_CONSTANTS];
    p.grey = st.opt[ASORA_OPT_GREY_NOTABLES];
    p.z_transposed = st.opt[ASORA_OPT_Z_TRANSPOSED] != 0 ? 1 : 0;
    p.ncell = (unsigned)st.ncell;
    p.nhi = st.nhi;
    p.phi = st.grid[ASORA_GRID_PHI_ION];
    p.tables = st.tables;
    p.heat = st.grid[ASORA_GRID_PHI_HEAT];
    p.src_pos = st.src_pos; p.src_flux = st.src_flux;
    p.counters = st.counters;
    p.radius_stays = note_call_radius(st, R, radius_path) ? 1 : 0;
#ifdef ASORA_ENABLE_ABLATION
    { const char *ab = getenv("ASORA_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }
#endif
}

// A raytrace call in three parts, so that a caller can overlap the multi-GPU sum of finished slabs of the
// rate grid with the tracing of later sources (asora_raytrace_begin / _range / _fold):
//   rt_begin  checks, zeroes the accumulators (raytracing.cu:113), forms nHI, fixes the parameters;
//   rt_range  traces a range of the uploaded sources into the accumulators (asynchronous);
//   rt_fold   adds the [k][j][i] accumulator of the z-faces into phi_ion for a slab of i-planes.
static int rt_begin(double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau, double *dump,
                    bool pipelined = false)
{
    State &st = g_state;
    st.rt_open = false;
    if (!st.grid_valid[ASORA_GRID_NDENS]) return fail(4, "raytrace: density not on device (density_to_device)");
    if (!st.grid_valid[ASORA_GRID_XH_AV]) return fail(4, "raytrace: xh_av not on device");
    if (!st.opt[ASORA_OPT_GREY_NOTABLES] && !st.tables)
        return fail(4, "raytrace: radiation tables not on device (photo_table_to_device)");
    if (!(R >= 0.0)) return fail(4, "raytrace: R must be >= 0");
    if (NumTau < 1 && !st.opt[ASORA_OPT_GREY_NOTABLES]) return fail(4, "raytrace: NumTau must be >= 1");

    const bool zt = st.opt[ASORA_OPT_Z_TRANSPOSED] != 0;
    const bool heat = st.opt[ASORA_OPT_HEATING] != 0 && dump == nullptr;
    if (heat && (!st.have_heat_tables || st.opt[ASORA_OPT_GREY_NOTABLES]))
        return fail(4, "raytrace: heating requested but no heating tables on device (heat_table_to_device)");
    const size_t bytes = st.ncell * sizeof(double);
    ASORA_HIP_TRY(hipMemsetAsync(st.grid[ASORA_GRID_PHI_ION], 0, bytes, st.stream));      // raytracing.cu:113
    if (zt) ASORA_HIP_TRY(hipMemsetAsync(st.phi_t, 0, bytes, st.stream));
    if (heat) {
        ASORA_HIP_TRY(hipMemsetAsync(st.grid[ASORA_GRID_PHI_HEAT], 0, bytes, st.stream));
        if (zt) ASORA_HIP_TRY(hipMemsetAsync(st.heat_t, 0, bytes, st.stream));
    }
    ASORA_HIP_TRY(hipMemsetAsync(st.counters, 0, sizeof(unsigned long long) * COUNTER_FIELDS * COUNTER_SLOTS, st.stream));
    if (int rc = launch_prepare_nhi(st, zt)) return rc;

    RtParams &p = st.rt_params;
    fill_rt_params(p, R, sig, dr, minlogtau, dlogtau, NumTau);
    p.dump = dump;
    st.rt_heat = heat;
    st.rt_pipelined = pipelined;
    st.rt_by_planes = false;
    if (pipelined) {       // the side streams start behind the zeroed accumulators and nHI
        ASORA_HIP_TRY(hipEventRecord(st.main_ready, st.stream));
        for (int q = 0; q < 2; ++q) {
            ASORA_HIP_TRY(hipStreamWaitEvent(st.side[q], st.main_ready, 0));
            st.side_pending[q] = false;
        }
        st.side_next = 0;
    }
    st.rt_open = true;
    return 0;
}

static int rt_range(int src_begin, int src_count)
{
    State &st = g_state;
    if (!st.rt_open) return fail(4, "raytrace_range: no raytrace in progress (call asora_raytrace_begin)");
    if (src_begin < 0 || src_count < 0 || src_begin + src_count > st.num_src)
        return fail(4, "raytrace: source range [" + std::to_string(src_begin) + "," +
                           std::to_string(src_begin + src_count) + ") outside the " + std::to_string(st.num_src) +
                           " uploaded sources (source_data_to_device)");
    if (src_count == 0) return 0;
    RtParams p = st.rt_params;
    p.src_pos = st.src_pos; p.src_flux = st.src_flux;
    // the whole list: in the spatially ordered copy (a column-density dump is of the caller's LAST source: caller's order)
    if (src_begin == 0 && src_count == st.num_src && st.src_pos_sorted && !p.dump) { p.src_pos = st.src_pos_sorted; p.src_flux = st.src_flux_sorted; }
    p.src_begin = src_begin; p.src_count = src_count;
    // one launch shape (one set of geometry tables) per call: a call that traces its sources in several ranges (pipelined
    // all-reduce, chunked slab exchange) is sized by all of the rank's sources
    p.shape_src_count = (st.rt_pipelined || st.rt_by_planes) ? st.num_src : src_count;
    if (!st.rt_pipelined) return launch_raytrace(st, p, p.dump != nullptr, st.rt_heat);
    const int q = st.side_next;
    st.side_next ^= 1;
    if (int rc = launch_raytrace(st, p, p.dump != nullptr, st.rt_heat, st.side[q])) return rc;
    ASORA_HIP_TRY(hipEventRecord(st.side_done[q], st.side[q]));
    st.side_pending[q] = true;
    return 0;
}

static int rt_fold(int i_begin, int i_count)
{
    State &st = g_state;
    if (!st.rt_open) return fail(4, "raytrace_fold: no raytrace in progress (call asora_raytrace_begin)");
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N) return fail(4, "raytrace_fold: bad plane range");
    for (int q = 0; q < 2; ++q)       // everything traced so far must have landed
        if (st.side_pending[q]) {
            ASORA_HIP_TRY(hipStreamWaitEvent(st.stream, st.side_done[q], 0));
            st.side_pending[q] = false;
        }
    if (st.rt_params.z_transposed && i_count > 0) {
        if (int rc = launch_fold_range(st, st.phi_t, st.grid[ASORA_GRID_PHI_ION], i_begin, i_count)) return rc;
        if (st.rt_heat)
            if (int rc = launch_fold_range(st, st.heat_t, st.grid[ASORA_GRID_PHI_HEAT], i_begin, i_count)) return rc;
    }
    st.grid_valid[ASORA_GRID_PHI_ION] = true;
    if (st.rt_heat) st.grid_valid[ASORA_GRID_PHI_HEAT] = true;
    return 0;
}

static int do_raytrace(double R, double sig, double dr, int src_begin, int src_count, double minlogtau,
                       double dlogtau, int NumTau, double *dump)
{
    State &st = g_state;
    if (src_begin < 0 || src_count < 0 || src_begin + src_count > st.num_src)
        return fail(4, "raytrace: source range [" + std::to_string(src_begin) + "," +
                           std::to_string(src_begin + src_count) + ") outside the " + std::to_string(st.num_src) +
                           " uploaded sources (source_data_to_device)");
    if (int rc = rt_begin(R, sig, dr, minlogtau, dlogtau, NumTau, dump)) return rc;
    if (int rc = rt_range(src_begin, src_count)) return rc;
    if (int rc = rt_fold(0, st.N)) return rc;
    st.rt_open = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Host driver of subbox.hip on device-resident inputs (do_all_sources / do_source,
// src/c2ray/raytracing.f90:52-249): NDENS and XH_AV on the device, tables and sources given as device pointers.
// ---------------------------------------------------------------------------------------------
struct SubboxCall {
    int max_subbox, subboxsize;
    float loss_fraction;
    double sig, dr, R, minlogtau, dlogtau;
    int NumTau, table_len;
    const double2 *tables;          // [thick | thin | heat thick | heat thin] pairs, table_len each
    const int32_t *src_pos;         // 0-based, xyz-interleaved
    const int32_t *host_pos;        // the same list on the host (pairing of sources for the line-aligned tables), or nullptr
    const double *src_flux;
    int src_begin, src_count;
    bool heat, keep_heat;           // keep_heat: add onto PHI_HEAT as it stands (f2py intent(inout)) instead of zeroing it
    double *dump;                   // N^3 grid receiving the column densities of the last source, or nullptr
};

static int subbox_core(const SubboxCall &c, long long &total_nbox, double &total_loss)
{
    State &st = g_state;
    const int N = st.N;
    const size_t bytes = st.ncell * sizeof(double);
    const bool grey = st.opt[ASORA_OPT_GREY_NOTABLES] != 0;
    ASORA_HIP_TRY(hipMemsetAsync(st.grid[ASORA_GRID_PHI_ION], 0, 2 * bytes, st.stream));          // f90:95 (+ its [k][j][i] twin)
    if (c.heat) {
        if (!c.keep_heat) ASORA_HIP_TRY(hipMemsetAsync(st.grid[ASORA_GRID_PHI_HEAT], 0, bytes, st.stream));
        ASORA_HIP_TRY(hipMemsetAsync(st.heat_t, 0, bytes, st.stream));
    }
    if (c.dump) ASORA_HIP_TRY(hipMemsetAsync(c.dump, 0, bytes, st.stream));
    if (int rc = launch_prepare_nhi(st, true)) return rc;
    if (int rc = ensure_logtab(st)) return rc;

    // traversal range per axis side, f90:174-175
    const int ext_r = std::min(c.max_subbox, N / 2 - 1 + N % 2);
    const int ext_l = std::min(c.max_subbox, N / 2);
    const int S_all = std::max(ext_r, ext_l);
    const bool range_open = ext_r > 0 && ext_l > 0;      // else the while loop of do_source never runs (f90:193-195)

    SubboxParams p;
    std::memset(&p, 0, sizeof p);
    p.N = N; p.W = std::max(S_all, 0) + 1;
    p.ext_r = ext_r; p.ext_l = ext_l;
    p.sig = c.sig; p.dr = c.dr; p.R = c.R;
    p.numtau_f = lut_index_limit(c.NumTau, c.table_len);                   // photorates.f90:141 real(NumTau)
    p.lut_k1 = 0.30102999566398119521 / c.dlogtau;
    p.lut_k0 = 1.0 - c.minlogtau / c.dlogtau;
    p.table_len = c.table_len;
    p.grey = grey ? 1 : 0; p.heat = c.heat ? 1 : 0; p.add_zero = st.opt[ASORA_OPT_SKIP_ZERO_RATES] == 1 ? 0 : 1;
    const int last = c.src_begin + c.src_count - 1;
    p.flux_src = st.opt[ASORA_OPT_C2RAY_OWN_FLUX] ? -1 : last;            // f90:500,503
    p.dump_src = last;
    p.ncell = (unsigned)st.ncell;
    p.nhi = st.nhi; p.phi = st.grid[ASORA_GRID_PHI_ION]; p.heat_grid = st.grid[ASORA_GRID_PHI_HEAT];
    p.dump = c.dump;
    p.tables = c.tables; p.logtab = st.logtab_dev;
    p.src_pos = c.src_pos; p.src_flux = c.src_flux;
    p.unit_stride = (size_t)6 * p.W * p.W;

    total_nbox = 0;
    total_loss = 0.0;
    // (pair lists are cached by the address of the source list: a caller's temporary list must not meet an older one's entries)
    struct DropPairs { State &s; bool on; ~DropPairs() { if (on) release_pair_lists(s); } } drop_pairs{st, c.src_pos != st.src_pos};
    if (drop_pairs.on) release_pair_lists(st);
    // Round 3: the sources whose column densities do not go back to the caller are swept on the ASORA kernel's tabulated
    // geometry (cells within R_max_LLS only; raytrace.hip, SUBBOX) when that applies; the dumped source -- it needs the
    // whole cube -- and everything else stay with the on-the-fly kernel of subbox.hip
    RtParams tp;
    fill_rt_params(tp, c.R, c.sig, c.dr, c.minlogtau, c.dlogtau, c.NumTau, 1);
    tp.numtau_f = p.numtau_f; tp.lut_k1 = p.lut_k1; tp.lut_k0 = p.lut_k0; tp.tau_zero = INFINITY;
    tp.table_len = c.table_len; tp.tables = c.tables;
    tp.fortran_consts = 1; tp.grey = grey ? 1 : 0; tp.z_transposed = 1;
    tp.logtab = st.logtab_dev;
    tp.src_pos = c.src_pos; tp.src_flux = c.src_flux;
    tp.flux_src = p.flux_src;
    const bool has_dump = c.dump != nullptr;
    SubboxTables tab;
    const int table_sources = c.src_count - (has_dump ? 1 : 0);
    {
        if (range_open && table_sources > 0)
            if (int rc = subbox_tables_prepare(st, tp, ext_r, ext_l, c.subboxsize, table_sources, c.heat, tab, c.host_pos)) return rc;
    }

    // sources in batches bounded by the scratch: the on-the-fly kernel keeps 8 octants x 2 buffers x 3 W^2 doubles per source
    // (6.4 MB at 256^3), the tabulated sweep one trailing shell per source and unit (tab.max_batch)
    const size_t per_src = 8 * p.unit_stride * sizeof(double);
    const size_t budget = (size_t)4 << 30;
    // (one batch when the trailing shells of all tabulated sources fit: the dumped source then runs beside them)
    const int max_batch = tab.ok ? (tab.max_batch >= table_sources ? std::max(c.src_count, 1) : tab.max_batch)
                                 : (int)std::max<size_t>(8, std::min<size_t>((budget / per_src) / 8 * 8, 1 << 20));
    const int cap = std::min(std::max(c.src_count, 1), max_batch);
    if ((size_t)cap > st.subbox_cap) {                   // per-source bookkeeping of a batch, kept between calls
        for (void *q : {(void *)st.sb_active, (void *)st.sb_nbox, (void *)st.sb_loss, (void *)st.sb_loss_final})
            if (q) (void)hipFree(q);
        st.sb_active = st.sb_nbox = nullptr; st.sb_loss = st.sb_loss_final = nullptr; st.subbox_cap = 0;
        ASORA_HIP_TRY(hipMalloc(&st.sb_active, sizeof(int) * cap));
        ASORA_HIP_TRY(hipMalloc(&st.sb_nbox, sizeof(int) * cap));
        ASORA_HIP_TRY(hipMalloc(&st.sb_loss, sizeof(double) * cap));
        ASORA_HIP_TRY(hipMalloc(&st.sb_loss_final, sizeof(double) * cap));
        st.subbox_cap = (size_t)cap;
    }
    if (!st.sb_nactive) ASORA_HIP_TRY(hipMalloc(&st.sb_nactive, sizeof(int)));
    std::vector<int> h_nbox((size_t)cap);
    std::vector<double> h_loss((size_t)cap);

    for (int done = 0; done < c.src_count;) {
        const int batch = std::min(c.src_count - done, max_batch);
        // with the tables, the on-the-fly kernel only sweeps the dumped source (the last one of the call)
        const bool dump_here = has_dump && done + batch == c.src_count;
        const int fly_count = tab.ok ? (dump_here ? 1 : 0) : batch;
        const int fly_first = tab.ok ? batch - fly_count : 0;          // batch-local index of the first source swept on the fly
        const size_t need = (size_t)8 * ((std::max(fly_count, 1) + 7) / 8) * per_src;
        if (need > st.shell_scratch_bytes) {
            if (st.shell_scratch) ASORA_HIP_TRY(hipFree(st.shell_scratch));
            st.shell_scratch = nullptr; st.shell_scratch_bytes = 0;
            ASORA_HIP_TRY(hipMalloc(&st.shell_scratch, need));
            st.shell_scratch_bytes = need;
        }
        const int first = c.src_begin + done;
        p.scratch = st.shell_scratch;
        p.src_begin = first + fly_first; p.src_count = fly_count;
        p.active = st.sb_active + fly_first; p.loss = st.sb_loss + fly_first;
        tp.src_begin = first; tp.src_count = batch - fly_count;
        tp.sb_active = st.sb_active; tp.sb_loss = st.sb_loss;
        int n_active = 0;
        if (int rc = launch_subbox_decide(st, 0, batch, c.src_flux, first, (double)c.loss_fraction, range_open ? 1 : 0,
                                          st.sb_active, st.sb_loss, st.sb_loss_final, st.sb_nbox, st.sb_nactive)) return rc;
        ASORA_HIP_TRY(hipMemcpyAsync(&n_active, st.sb_nactive, sizeof(int), hipMemcpyDeviceToHost, st.stream));
        ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
        long long box = 0;                                    // half-width of the current sub-box, f90:199-200
        while (n_active > 0) {
            const long long prev_box = box;
            box += c.subboxsize;
            p.s_begin = (int)std::min<long long>(prev_box, S_all);
            p.s_end = (int)std::min<long long>(box, S_all);
            p.edge_r = (int)std::min<long long>(box, ext_r);
            p.edge_l = (int)std::min<long long>(box, ext_l);
            // the dumped source's sweep (8 wide workgroups: as long as ONE workgroup lasts) runs beside the tabulated sweep of
            // all the others, on a side stream; both add into the same rate grids
            const bool beside = fly_count > 0 && tab.ok && tp.src_count > 0;
            if (beside) {
                ASORA_HIP_TRY(hipEventRecord(st.main_ready, st.stream));
                ASORA_HIP_TRY(hipStreamWaitEvent(st.side[0], st.main_ready, 0));
            }
            if (fly_count > 0) { if (int rc = launch_subbox_sweep(st, p, beside ? st.side[0] : nullptr)) return rc; }
            if (beside) ASORA_HIP_TRY(hipEventRecord(st.side_done[0], st.side[0]));
            if (tab.ok && tp.src_count > 0) {
                tp.sb_edge_r = p.edge_r; tp.sb_edge_l = p.edge_l;
                if (int rc = subbox_tables_sweep(st, tp, tab, p.s_begin, p.s_end, c.heat)) return rc;
            }
            if (beside) ASORA_HIP_TRY(hipStreamWaitEvent(st.stream, st.side_done[0], 0));
            const int more_range = (box < ext_r && box < ext_l) ? 1 : 0;          // f90:194-195
            if (int rc = launch_subbox_decide(st, 1, batch, c.src_flux, first, (double)c.loss_fraction, more_range,
                                              st.sb_active, st.sb_loss, st.sb_loss_final, st.sb_nbox, st.sb_nactive)) return rc;
            ASORA_HIP_TRY(hipMemcpyAsync(&n_active, st.sb_nactive, sizeof(int), hipMemcpyDeviceToHost, st.stream));
            ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
        }
        ASORA_HIP_TRY(hipMemcpy(h_nbox.data(), st.sb_nbox, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost));
        ASORA_HIP_TRY(hipMemcpy(h_loss.data(), st.sb_loss_final, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost));
        for (int s = 0; s < batch; ++s) { total_nbox += h_nbox[s]; total_loss += h_loss[s]; }   // f90:246-247, in source order
        done += batch;
    }

    // fold the [k][j][i] accumulators
    if (int rc = launch_finish_phi(st)) return rc;
    st.grid_valid[ASORA_GRID_PHI_ION] = true;
    if (c.heat) {
        if (int rc = launch_fold_transposed(st, st.heat_t, st.grid[ASORA_GRID_PHI_HEAT])) return rc;
        st.grid_valid[ASORA_GRID_PHI_HEAT] = true;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The drop-in asora_do_all_sources with its two PCIe copies hidden behind the trace
// ---------------------------------------------------------------------------------------------
// The reference's call uploads xh_av (N^3 doubles), traces, downloads phi_ion (raytracing.cu:117-146): at 256^3 the two
// copies take 2 x 2.4 ms at the link's ~56 GB/s against 1.4 ms of tracing 1000 sources.  A source at plane i0 only needs
// nHI on, and only rates, the planes within R of it.  So the grid is cut into K slabs of planes; the slabs of xh_av are
// uploaded one after the other on a copy stream, nHI of a slab is formed as soon as it has arrived, the sources of a
// slab (a second copy of the source list, ordered by first coordinate) are traced as soon as the slabs they reach are
// there, and a slab of phi_ion is folded and sent to the host on a second copy stream as soon as the last source that
// reaches it has been traced -- upload, trace and download overlap (PCIe is full duplex).  The host buffers are
// registered (pinned) for the duration of the call so that the copies are asynchronous; re-registering a buffer the
// driver has seen before costs microseconds (tools/micro/pcie.hip).  done = false: conditions not met, nothing was
// started, the caller takes the plain path.
static int do_all_sources_pipelined(double R, double sig, double dr, const double *xh_av, double *phi_ion, int NumSrc,
                                    double minlogtau, double dlogtau, int NumTau, bool &done)
{
    State &st = g_state;
    done = false;
    const int N = st.N;
    constexpr int KMAX = 16;
    const char *kenv = getenv("ASORA_PIPELINE_SLABS");
    const int K = kenv ? std::max(2, std::min(KMAX, atoi(kenv))) : 8;
    if (!st.opt[ASORA_OPT_PIPELINED_COPIES] || !st.opt[ASORA_OPT_Z_TRANSPOSED] || st.opt[ASORA_OPT_HEATING]) return 0;
    if (NumSrc != st.num_src || NumSrc < 1 || !st.src_pos_sorted || N < 8 * K) return 0;
    if (!std::isfinite(R) || !(R >= 0.0)) return 0;
    const int m = (int)std::floor(R);                       // a source rates the planes i0 - floor(R) ... i0 + floor(R)
    if (2 * m + N / K >= N) return 0;                       // every slab of sources reaches (nearly) every plane
    if (!st.grid_valid[ASORA_GRID_NDENS]) return fail(4, "raytrace: density not on device (density_to_device)");
    if (!st.opt[ASORA_OPT_GREY_NOTABLES] && !st.tables)
        return fail(4, "raytrace: radiation tables not on device (photo_table_to_device)");
    if (NumTau < 1 && !st.opt[ASORA_OPT_GREY_NOTABLES]) return fail(4, "raytrace: NumTau must be >= 1");

    const size_t bytes = st.ncell * sizeof(double);
    if (hipHostRegister((void *)xh_av, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipHostRegister((void *)phi_ion, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError(); (void)hipHostUnregister((void *)xh_av); return 0;
    }
    // on EVERY exit path -- also the error returns below, with copies into and out of the caller's buffers possibly still in
    // flight -- the three streams are drained before the buffers are unregistered and handed back
    struct Unpin {
        const void *a, *b;
        ~Unpin()
        {
            State &s = g_state;
            for (hipStream_t q : {s.side[0], s.side[1], s.stream}) if (q) (void)hipStreamSynchronize(q);
            (void)hipHostUnregister((void *)a); (void)hipHostUnregister((void *)b);
        }
    } unpin{xh_av, phi_ion};

    while ((int)st.pipe_events.size() < 2 * K) {
        hipEvent_t e = nullptr;
        ASORA_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        st.pipe_events.push_back(e);
    }
    hipStream_t up = st.side[0], down = st.side[1];
    const size_t plane = (size_t)N * N;
    int lo[KMAX + 1];
    for (int c = 0; c <= K; ++c) lo[c] = c * N / K;
    // sources of slab c: [sb[c], sb[c+1]) of the sorted list
    int sb[KMAX + 1];
    for (int c = 0; c <= K; ++c)
        sb[c] = (int)(std::lower_bound(st.src_i0_sorted.begin(), st.src_i0_sorted.end(), lo[c]) - st.src_i0_sorted.begin());
    // reach[c][d]: do sources of slab c touch planes of slab d (within m planes, periodically)
    bool reach[KMAX][KMAX];
    for (int c = 0; c < K; ++c)
        for (int d = 0; d < K; ++d) {
            bool hit = false;
            if (sb[c + 1] > sb[c])
                for (int q = lo[d]; q < lo[d + 1] && !hit; ++q) {
                    // distance from plane q to the interval [lo[c], lo[c+1]) on the ring
                    int dist = 0;
                    if (q < lo[c]) dist = std::min(lo[c] - q, q + N - (lo[c + 1] - 1));
                    else if (q >= lo[c + 1]) dist = std::min(q - (lo[c + 1] - 1), lo[c] + N - q);
                    hit = dist <= m;
                }
            reach[c][d] = hit;
        }

    ASORA_HIP_TRY(hipMemsetAsync(st.grid[ASORA_GRID_PHI_ION], 0, 2 * bytes, st.stream));      // raytracing.cu:113 (+ twin)
    ASORA_HIP_TRY(hipMemsetAsync(st.counters, 0, sizeof(unsigned long long) * COUNTER_FIELDS * COUNTER_SLOTS, st.stream));
    RtParams base;
    fill_rt_params(base, R, sig, dr, minlogtau, dlogtau, NumTau);
    base.src_pos = st.src_pos_sorted; base.src_flux = st.src_flux_sorted;
    base.shape_src_count = NumSrc;
    st.rt_open = false;
    // the copy streams start behind whatever the main stream has done so far (earlier calls may still own the grids)
    ASORA_HIP_TRY(hipEventRecord(st.main_ready, st.stream));
    ASORA_HIP_TRY(hipStreamWaitEvent(up, st.main_ready, 0));
    ASORA_HIP_TRY(hipStreamWaitEvent(down, st.main_ready, 0));

    bool prepped[KMAX] = {}, traced[KMAX] = {}, sent[KMAX] = {};
    auto try_traces = [&]() -> int {
        for (int c = 0; c < K; ++c) {
            if (traced[c]) continue;
            bool ready = true;
            for (int d = 0; d < K; ++d) if (reach[c][d] && !prepped[d]) ready = false;
            if (!ready) continue;
            if (sb[c + 1] > sb[c]) {
                RtParams p = base;
                p.src_begin = sb[c]; p.src_count = sb[c + 1] - sb[c];
                if (int rc = launch_raytrace(st, p, false, false)) return rc;
            }
            traced[c] = true;
        }
        return 0;
    };
    // A device-to-host copy blocks the calling thread until it has run (measured; the uploads do not): so all uploads are
    // enqueued first, a slab's fold is enqueued as soon as the slab is final, and its download is only ISSUED one round
    // later, after the next round's kernels have been enqueued -- the host then waits in the copy while the GPU traces.
    // (Letting the fold kernel write the slab straight into the pinned host buffer instead was measured as well: 5.9 ms
    //  per call against 5.0 ms this way -- that kernel does not overlap with the uploads either.)
    bool folded[KMAX] = {};
    std::vector<int> to_send;
    auto try_folds = [&]() -> int {
        for (int d = 0; d < K; ++d) {
            if (folded[d]) continue;
            bool final_ = true;
            for (int c = 0; c < K; ++c) if (reach[c][d] && !traced[c]) final_ = false;
            if (!final_) continue;
            if (int rc = launch_fold_range(st, st.phi_t, st.grid[ASORA_GRID_PHI_ION], lo[d], lo[d + 1] - lo[d])) return rc;
            ASORA_HIP_TRY(hipEventRecord(st.pipe_events[K + d], st.stream));
            folded[d] = true;
            to_send.push_back(d);
        }
        return 0;
    };
    auto send = [&](int d) -> int {
        ASORA_HIP_TRY(hipStreamWaitEvent(down, st.pipe_events[K + d], 0));
        ASORA_HIP_TRY(hipMemcpyAsync(phi_ion + (size_t)lo[d] * plane, st.grid[ASORA_GRID_PHI_ION] + (size_t)lo[d] * plane,
                                     (size_t)(lo[d + 1] - lo[d]) * plane * sizeof(double), hipMemcpyDeviceToHost, down));   // cu:146
        sent[d] = true;
        return 0;
    };
    // upload order: the slabs the sources of slab 0 reach back into first (K-w ... K-1), then 0, 1, ...
    int w = 0;                                    // how many slabs back the sources of a slab reach
    for (int c = 0; c < K; ++c)
        for (int d = 0; d < K; ++d) {
            const int back = (c - d + K) % K;     // d lies `back` slabs behind c (more than half the ring: it lies ahead)
            if (reach[c][d] && back <= K / 2) w = std::max(w, back);
        }
    for (int q = 0; q < K; ++q) {
        const int c = (q + K - w) % K;
        ASORA_HIP_TRY(hipMemcpyAsync(st.grid[ASORA_GRID_XH_AV] + (size_t)lo[c] * plane, xh_av + (size_t)lo[c] * plane,
                                     (size_t)(lo[c + 1] - lo[c]) * plane * sizeof(double), hipMemcpyHostToDevice, up));   // cu:117
        ASORA_HIP_TRY(hipEventRecord(st.pipe_events[c], up));
    }
    for (int q = 0; q < K; ++q) {
        const int c = (q + K - w) % K;
        ASORA_HIP_TRY(hipStreamWaitEvent(st.stream, st.pipe_events[c], 0));
        if (int rc = launch_prepare_range(st, lo[c], lo[c + 1] - lo[c], false, nullptr)) return rc;
        prepped[c] = true;
        const std::vector<int> ready = to_send;       // final since the previous round: their folds are already enqueued
        to_send.clear();
        if (int rc = try_traces()) return rc;
        if (int rc = try_folds()) return rc;
        for (int d : ready) if (int rc = send(d)) return rc;
    }
    for (int d : to_send) if (int rc = send(d)) return rc;
    for (int c = 0; c < K; ++c) if (!traced[c] || !sent[c]) return fail(11, "do_all_sources: pipeline schedule incomplete (internal error)");
    ASORA_HIP_TRY(hipStreamSynchronize(down));
    ASORA_HIP_TRY(hipStreamSynchronize(up));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    st.grid_valid[ASORA_GRID_XH_AV] = true;
    st.grid_valid[ASORA_GRID_PHI_ION] = true;
    done = true;
    return 0;
}

} // namespace asora

namespace {
struct DeviceBuffers {      // frees what the call allocated, on every exit path
    std::vector<void *> ptrs;
    ~DeviceBuffers() { for (void *q : ptrs) (void)hipFree(q); }
    template <typename T> int alloc(T *&out, size_t count)
    {
        void *d = nullptr;
        ASORA_HIP_TRY(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
        ptrs.push_back(d);
        out = static_cast<T *>(d);
        return 0;
    }
};
}

using namespace asora;

extern "C" {

// ---------------------------------------------------------------------------------------------
// Where the grids lie.  The fused pass streams 5 grids in and 6-7 out at once, and how fast the memory side takes that mix depends
// on WHERE those grids lie physically: the same kernel on the same box moves 5.2-5.4 TB/s on most placements and 6.0-6.2 TB/s on
// a fifth to a third of them, stable for the life of the allocation (tools/micro/placement_probe.hip; some boxes offer one kind only;
// with one hipMalloc per grid the sets spread between the two, which is what earlier rounds recorded as the "state of the box":
// fused pass 0.25 ... 0.31 ms from run to run).  Nothing a process can read tells the two kinds apart beforehand, so device_init
// tries a FEW allocations of the whole arena, three launches of a kernel with the pass's stream mix on each, keeps the fastest and
// frees the others.  Round 6 bounds (round 5 tried up to 32 within a quarter of the free memory: 56 GB and 96 probe launches on a
// box of one kind, for a 4 % spread):
//   * at most ASORA_OPT_PLACEMENT_CANDIDATES allocations (0 = default 8; 1 = take the first; the environment variable
//     ASORA_PLACEMENT_CANDIDATES does the same for a process that cannot call asora_set_option before device_init);
//   * what is held during the probe stays within an EIGHTH of the free device memory (512^3: two candidates of 14 GiB);
//   * it stops as soon as it holds a candidate 7 % faster than another (both kinds seen, one of the fast kind in hand); a box of one
//     kind costs all eight (10 ms at 256^3: one placement in eight to one in three is of the fast kind where both occur, so fewer
//     tries would miss it too often).
// The losers are held until the probe ends: a freed arena's pages are what the next allocation of that size gets back, so
// freeing as it goes would time the same placement again and again.  Meshes below 128^3 take the first allocation (their grids
// sit in the caches).  asora_debug_placement reports what was tried and what the probe cost.
// On a box with both kinds, alternating processes (profiles/r05_ab_placement.txt): first allocation taken 1.392-1.395 ms per step
// in five runs of six (fused pass 0.282, trace 1.079), probed 1.333-1.337 in six of six (0.245, 1.060; 2-6 candidates tried).
// ---------------------------------------------------------------------------------------------
constexpr int ARENA_CANDIDATES = 8;        // default bound (one placement in eight to one in three is of the fast kind where both occur)
constexpr int ARENA_SLOTS = 14;            // ndens, xh, xh_av, temp, xh_intermed, 4 accumulators, nhi x 2, phi_ion x 2, staging
__global__ void __launch_bounds__(256) placement_probe_kernel(char *arena, size_t slot, size_t n)
{
    // 5 read streams (non-temporal, as the pass loads them) and 7 write streams over the first 12 slots
    const double *in[5]; double *out[7];
    for (int q = 0; q < 5; ++q) in[q] = reinterpret_cast<const double *>(arena + q * slot);
    for (int q = 0; q < 7; ++q) out[q] = reinterpret_cast<double *>(arena + (5 + q) * slot);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) s += __builtin_nontemporal_load(in[q] + i);
#pragma unroll
        for (int q = 0; q < 7; ++q) out[q][i] = s;
    }
}

static int choose_arena(State &st, size_t slot, int slots)
{
    const auto wall0 = std::chrono::steady_clock::now();
    const size_t total = slot * (size_t)slots;
    int want = st.opt[ASORA_OPT_PLACEMENT_CANDIDATES] > 0 ? st.opt[ASORA_OPT_PLACEMENT_CANDIDATES] : ARENA_CANDIDATES;
    if (const char *e = getenv("ASORA_PLACEMENT_CANDIDATES")) want = std::max(1, atoi(e));
    want = std::min(want, 32);
    if (st.N < 128) want = 1;
    size_t free_b = 0, all_b = 0;
    if (want > 1 && hipMemGetInfo(&free_b, &all_b) == hipSuccess && total > 0)
        want = (int)std::max<size_t>(1, std::min<size_t>((size_t)want, (size_t)(0.125 * (double)free_b) / total));     // (several ranks may share a GPU)
    st.arena_candidates = 0; st.arena_probe_ms = st.arena_probe_worst_ms = st.arena_probe_wall_ms = 0.0;
    if (want == 1) {
        ASORA_HIP_TRY(hipMalloc(&st.arena, total));
        st.arena_bytes = total; st.arena_candidates = 1;
        return 0;
    }
    hipEvent_t t0, t1;
    ASORA_HIP_TRY(hipEventCreate(&t0)); ASORA_HIP_TRY(hipEventCreate(&t1));
    std::vector<char *> cand;
    std::vector<float> ms;
    int best = -1;
    float worst = 0.0f;
    const size_t n = st.ncell;
    for (int c = 0; c < want; ++c) {
        char *a = nullptr;
        if (hipMalloc(&a, total) != hipSuccess) { (void)hipGetLastError(); break; }
        cand.push_back(a);
        float t = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(t0, st.stream);
            hipLaunchKernelGGL(placement_probe_kernel, dim3(1024), dim3(256), 0, st.stream, a, slot, n);
            (void)hipEventRecord(t1, st.stream);
            if (hipEventSynchronize(t1) != hipSuccess) { t = 1e30f; break; }
            float e = 0.0f;
            (void)hipEventElapsedTime(&e, t0, t1);
            if (rep >= 1) t = std::min(t, e);              // (the first launch maps the pages)
        }
        ms.push_back(t);
        if (best < 0 || t < ms[(size_t)best]) best = c;
        worst = std::max(worst, t);
        if (c >= 1 && ms[(size_t)best] <= 0.93f * worst) break;     // both kinds seen, and one of the fast kind in hand
    }
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    if (best < 0) return fail(2, "device_init: out of device memory for the grids");
    for (size_t c = 0; c < cand.size(); ++c) if ((int)c != best) (void)hipFree(cand[c]);
    st.arena = cand[(size_t)best];
    st.arena_bytes = total;
    st.arena_candidates = (int)cand.size();
    st.arena_probe_ms = ms[(size_t)best]; st.arena_probe_worst_ms = worst;
    st.arena_probe_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return 0;
}

void asora_debug_placement(int *candidates, double *chosen_probe_ms, double *slowest_probe_ms)
{
    if (candidates) *candidates = g_state.arena_candidates;
    if (chosen_probe_ms) *chosen_probe_ms = g_state.arena_probe_ms;
    if (slowest_probe_ms) *slowest_probe_ms = g_state.arena_probe_worst_ms;
}

void asora_debug_init_cost(double *device_init_ms, double *placement_probe_ms)
{
    if (device_init_ms) *device_init_ms = g_state.device_init_wall_ms;
    if (placement_probe_ms) *placement_probe_ms = g_state.arena_probe_wall_ms;
}

const char *asora_last_error(void) { return g_error.c_str(); }

int asora_device_init_ex(int N, int num_src_par, int device_id)
{
    clear_error();
    State &st = g_state;
    const auto init_wall0 = std::chrono::steady_clock::now();
    // validate everything first: a refused re-initialisation leaves the working state as it was
    if (N < 2 || N > 1280) return fail(1, "device_init: N must be in [2, 1280] (32-bit cell indices over 2 N^3)");
    if (st.stream && device_id != st.device) return fail(1, "device_init: the device cannot change within a process");
    if (st.init) release_all();
    st.device = device_id;
    if (int rc = ensure_runtime()) return rc;
    st.N = N;
    st.ncell = (size_t)N * N * N;
    st.num_src_par = num_src_par;
    st.auto_init = false;
    const size_t bytes = st.ncell * sizeof(double);
    // ONE allocation for every N^3 grid of the hot loop (choose_arena); the rate grids and nHI carry their [k][j][i] twin directly
    // behind them (one 32-bit index reaches both)
    const size_t slot = (bytes + 4095) / 4096 * 4096;
    if (int rc = choose_arena(st, slot, ARENA_SLOTS)) return rc;
    {
        size_t at = 0;
        auto take = [&](size_t grids) { double *q = reinterpret_cast<double *>(st.arena + at); at += grids * slot; return q; };
        // (two-grid buffers: the second half starts ncell doubles behind the first -- inside the two slots either way)
        st.grid[ASORA_GRID_NDENS] = take(1); st.grid[ASORA_GRID_XH] = take(1); st.grid[ASORA_GRID_XH_AV] = take(1);
        st.grid[ASORA_GRID_TEMP] = take(1); st.grid[ASORA_GRID_XH_INTERMED] = take(1);
        st.acc = take(4);
        st.nhi = take(2);
        st.grid[ASORA_GRID_PHI_ION] = take(2);
        st.staging = take(1);
    }
    for (int g = 0; g < ASORA_GRID_COUNT; ++g)
        if (g != ASORA_GRID_PHI_HEAT && !st.grid[g]) return fail(11, "device_init: a grid without a place in the arena (internal error)");
    st.nhi_t = st.nhi + st.ncell;
    st.phi_t = st.grid[ASORA_GRID_PHI_ION] + st.ncell;
    st.heat_t = nullptr;
    st.ev_sets_known = false;
    {   // per-workgroup partials of the tiled chemistry pass: the j-chunk count is rounded up per (k tile x i tile), so a
        // RANGE of planes (asora_chemistry_range: a multi-GPU rank's slab) can need more workgroups than the whole grid
        size_t worst = 0;
        for (int planes = 1; planes <= N; ++planes) worst = std::max(worst, chemistry_tile_blocks(st, N, planes));
        if (int rc = ensure_red_capacity(3 * worst)) return rc;
    }
    st.init = true;
    st.device_init_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - init_wall0).count();
    return 0;
}

int asora_device_init(int N, int num_src_par)
{
    int dev = g_state.stream ? g_state.device : 0;
    if (!g_state.stream) {
        // the reference uses the current device (memory.cu:39)
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    return asora_device_init_ex(N, num_src_par, dev);
}

int asora_device_close(void)
{
    clear_error();
    if (int rc = require_init("device_close")) return rc;
    ASORA_HIP_TRY(hipStreamSynchronize(g_state.stream));
    return release_all();
}

int asora_grid_to_device(int which, const double *host, int N, char order)
{
    clear_error();
    if (int rc = require_init("grid_to_device")) return rc;
    if (int rc = check_N("grid_to_device", N)) return rc;
    if (which < 0 || which >= ASORA_GRID_COUNT) return fail(3, "grid_to_device: bad grid selector");
    if (!host) return fail(3, "grid_to_device: null host pointer");
    State &st = g_state;
    if (which == ASORA_GRID_PHI_HEAT) { if (int rc = ensure_heat_grid()) return rc; }
    st.zero_since_probe = std::max(st.zero_since_probe, 48);   // new medium: look again for cells beyond the table soon (launch_raytrace: at 64)
    const size_t bytes = st.ncell * sizeof(double);
    if (order == 'C' || order == 'c') {
        ASORA_HIP_TRY(hipMemcpyAsync(st.grid[which], host, bytes, hipMemcpyHostToDevice, st.stream));
    } else if (order == 'F' || order == 'f') {
        ASORA_HIP_TRY(hipMemcpyAsync(st.staging, host, bytes, hipMemcpyHostToDevice, st.stream));
        if (int rc = launch_transpose(st, st.staging, st.grid[which], N)) return rc;
    } else
        return fail(3, "grid_to_device: order must be 'C' or 'F'");
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    st.grid_valid[which] = true;
    if (which == ASORA_GRID_TEMP) st.temp_probe_valid = false;
    return 0;
}

int asora_grid_to_host(int which, double *host, int N, char order)
{
    clear_error();
    if (int rc = require_init("grid_to_host")) return rc;
    if (int rc = check_N("grid_to_host", N)) return rc;
    if (which < 0 || which >= ASORA_GRID_COUNT) return fail(3, "grid_to_host: bad grid selector");
    if (!host) return fail(3, "grid_to_host: null host pointer");
    State &st = g_state;
    if (!st.grid[which] || !st.grid_valid[which]) return fail(3, "grid_to_host: grid " + std::to_string(which) + " holds no data");
    const size_t bytes = st.ncell * sizeof(double);
    if (order == 'C' || order == 'c') {
        ASORA_HIP_TRY(hipMemcpyAsync(host, st.grid[which], bytes, hipMemcpyDeviceToHost, st.stream));
    } else if (order == 'F' || order == 'f') {
        if (int rc = launch_transpose(st, st.grid[which], st.staging, N)) return rc;
        ASORA_HIP_TRY(hipMemcpyAsync(host, st.staging, bytes, hipMemcpyDeviceToHost, st.stream));
    } else
        return fail(3, "grid_to_host: order must be 'C' or 'F'");
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    return 0;
}

int asora_grid_copy(int dst, int src)
{
    clear_error();
    if (int rc = require_init("grid_copy")) return rc;
    if (dst < 0 || dst >= ASORA_GRID_COUNT || src < 0 || src >= ASORA_GRID_COUNT || dst == src)
        return fail(3, "grid_copy: bad grid selectors");
    State &st = g_state;
    if (!st.grid_valid[src]) return fail(3, "grid_copy: source grid holds no data");
    if (dst == ASORA_GRID_PHI_HEAT) { if (int rc = ensure_heat_grid()) return rc; }
    ASORA_HIP_TRY(hipMemcpyAsync(st.grid[dst], st.grid[src], st.ncell * sizeof(double), hipMemcpyDeviceToDevice,
                                 st.stream));
    st.grid_valid[dst] = true;
    if (dst == ASORA_GRID_TEMP) st.temp_probe_valid = false;
    return 0;
}

int asora_grid_scale(int which, double factor)
{
    clear_error();
    if (int rc = require_init("grid_scale")) return rc;
    if (which < 0 || which >= ASORA_GRID_COUNT) return fail(3, "grid_scale: bad grid selector");
    State &st = g_state;
    if (!st.grid_valid[which]) return fail(3, "grid_scale: grid " + std::to_string(which) + " holds no data");
    if (int rc = launch_scale(st, st.grid[which], st.ncell, factor)) return rc;
    if (which == ASORA_GRID_TEMP) st.temp_probe_valid = false;
    return 0;
}

int asora_grid_sum(int which, double *sum)
{
    clear_error();
    if (int rc = require_init("grid_sum")) return rc;
    if (which < 0 || which >= ASORA_GRID_COUNT) return fail(3, "grid_sum: bad grid selector");
    if (!sum) return fail(3, "grid_sum: null output pointer");
    State &st = g_state;
    if (!st.grid_valid[which]) return fail(3, "grid_sum: grid " + std::to_string(which) + " holds no data");
    if (!st.temp_probe_dev) ASORA_HIP_TRY(hipMalloc(&st.temp_probe_dev, sizeof(double) * 8));
    if (int rc = launch_grid_sum(st, st.grid[which], st.ncell, st.temp_probe_dev + 5)) return rc;
    ASORA_HIP_TRY(hipMemcpyAsync(sum, st.temp_probe_dev + 5, sizeof(double), hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    return 0;
}

int asora_host_alloc(size_t bytes, void **host)
{
    clear_error();
    if (!host || bytes == 0) return fail(3, "host_alloc: null pointer or zero size");
    *host = nullptr;
    ASORA_HIP_TRY(hipHostMalloc(host, bytes, hipHostMallocDefault));
    return 0;
}

int asora_host_free(void *host)
{
    clear_error();
    if (!host) return 0;
    ASORA_HIP_TRY(hipHostFree(host));
    return 0;
}

void *asora_device_ptr(int which)
{
    if (!g_state.init || which < 0 || which >= ASORA_GRID_COUNT) return nullptr;
    return g_state.grid[which];
}

int asora_density_to_device(const double *ndens, int N)
{
    return asora_grid_to_device(ASORA_GRID_NDENS, ndens, N, 'C');
}

int asora_photo_table_to_device(const double *thin_table, const double *thick_table, int NumTau)
{
    clear_error();
    if (int rc = require_init("photo_table_to_device")) return rc;
    if (NumTau < 1 || !thin_table || !thick_table) return fail(3, "photo_table_to_device: empty table");
    State &st = g_state;
    st.zero_since_probe = std::max(st.zero_since_probe, 48);
    if (st.tables) { (void)hipFree(st.tables); st.tables = nullptr; }
    // device layout: rates_device.hpp (one 16-byte load serves the linear interpolation of photo_lookuptable, rates.cu:82)
    // (at least 16 elements: the kernels' pipeline-priming loads read a few fixed small offsets whatever the table's length)
    std::vector<double2> pairs(std::max<size_t>(4 * (size_t)NumTau, 16), double2{0.0, 0.0});   // [thick | thin | heat thick | heat thin]
    pack_rate_table(pairs.data(), 0, thick_table, NumTau);
    pack_rate_table(pairs.data(), 1, thin_table, NumTau);
    ASORA_HIP_TRY(hipMalloc(&st.tables, pairs.size() * sizeof(double2)));
    ASORA_HIP_TRY(hipMemcpy(st.tables, pairs.data(), pairs.size() * sizeof(double2), hipMemcpyHostToDevice));
    st.table_len = NumTau;
    st.have_heat_tables = false;
    return 0;
}

int asora_heat_table_to_device(const double *heat_thin_table, const double *heat_thick_table, int NumTau)
{
    clear_error();
    if (int rc = require_init("heat_table_to_device")) return rc;
    State &st = g_state;
    if (!st.tables) return fail(4, "heat_table_to_device: upload the photo tables first (photo_table_to_device)");
    if (NumTau != st.table_len || !heat_thin_table || !heat_thick_table)
        return fail(3, "heat_table_to_device: the heating tables must have the length of the photo tables (" +
                           std::to_string(st.table_len) + ")");
    if (int rc = ensure_heat_grid()) return rc;
    std::vector<double2> pairs(4 * (size_t)NumTau, double2{0.0, 0.0});
    pack_rate_table(pairs.data(), 2, heat_thick_table, NumTau);
    pack_rate_table(pairs.data(), 3, heat_thin_table, NumTau);
    const size_t lo = rate_table_byte_offset(2, NumTau), hi = rate_table_byte_offset(4, NumTau);
    ASORA_HIP_TRY(hipMemcpy(reinterpret_cast<char *>(st.tables) + lo, reinterpret_cast<const char *>(pairs.data()) + lo, hi - lo,
                            hipMemcpyHostToDevice));
    st.have_heat_tables = true;
    return 0;
}

int asora_source_data_to_device(const int32_t *pos, const double *flux, int NumSrc)
{
    clear_error();
    if (int rc = require_init("source_data_to_device")) return rc;
    if (NumSrc < 0 || (NumSrc > 0 && (!pos || !flux))) return fail(3, "source_data_to_device: bad arguments");
    State &st = g_state;
    st.zero_since_probe = std::max(st.zero_since_probe, 48);
    // validate on the host before anything reaches a kernel: positions index the grid directly
    for (int s = 0; s < NumSrc; ++s)
        for (int ax = 0; ax < 3; ++ax)
            if (pos[3 * s + ax] < 0 || pos[3 * s + ax] >= st.N)
                return fail(3, "source_data_to_device: source " + std::to_string(s) + " lies outside the mesh (0-based " +
                                   std::to_string(pos[3 * s + ax]) + " on axis " + std::to_string(ax) + ")");
    if (st.src_pos) { (void)hipFree(st.src_pos); st.src_pos = nullptr; }          // memory.cu:102-103
    if (st.src_flux) { (void)hipFree(st.src_flux); st.src_flux = nullptr; }
    if (st.src_pos_sorted) { (void)hipFree(st.src_pos_sorted); st.src_pos_sorted = nullptr; }
    if (st.src_flux_sorted) { (void)hipFree(st.src_flux_sorted); st.src_flux_sorted = nullptr; }
    st.src_i0_sorted.clear();
    st.src_pos_host.clear(); st.src_pos_sorted_host.clear();
    release_pair_lists(st);
    st.num_src = 0;
    if (NumSrc == 0) return 0;
    ASORA_HIP_TRY(hipMalloc(&st.src_pos, sizeof(int32_t) * 3 * (size_t)NumSrc));
    ASORA_HIP_TRY(hipMalloc(&st.src_flux, sizeof(double) * (size_t)NumSrc));
    ASORA_HIP_TRY(hipMemcpy(st.src_pos, pos, sizeof(int32_t) * 3 * (size_t)NumSrc, hipMemcpyHostToDevice));
    ASORA_HIP_TRY(hipMemcpy(st.src_flux, flux, sizeof(double) * (size_t)NumSrc, hipMemcpyHostToDevice));
    {   // a second copy in lexicographic order of the position (the sum over sources does not depend on their order): what
        // a call that traces the WHOLE list works from -- sources that run side by side are then neighbours in space and
        // share nHI and rate lines (measured -2 % on the trace at r_RT = 16 and 32) -- and, ordered by the first
        // coordinate, what the pipelined asora_do_all_sources cuts into slabs
        std::vector<int> order((size_t)NumSrc);
        for (int s = 0; s < NumSrc; ++s) order[s] = s;
        std::stable_sort(order.begin(), order.end(), [pos](int a, int b) {
            if (pos[3 * a] != pos[3 * b]) return pos[3 * a] < pos[3 * b];
            if (pos[3 * a + 1] != pos[3 * b + 1]) return pos[3 * a + 1] < pos[3 * b + 1];
            return pos[3 * a + 2] < pos[3 * b + 2];
        });
        std::vector<int32_t> ps(3 * (size_t)NumSrc);
        std::vector<double> fs((size_t)NumSrc);
        st.src_i0_sorted.resize((size_t)NumSrc);
        for (int s = 0; s < NumSrc; ++s) {
            const int o = order[s];
            ps[3 * s] = pos[3 * o]; ps[3 * s + 1] = pos[3 * o + 1]; ps[3 * s + 2] = pos[3 * o + 2];
            fs[s] = flux[o];
            st.src_i0_sorted[s] = pos[3 * o];
        }
        ASORA_HIP_TRY(hipMalloc(&st.src_pos_sorted, sizeof(int32_t) * 3 * (size_t)NumSrc));
        ASORA_HIP_TRY(hipMalloc(&st.src_flux_sorted, sizeof(double) * (size_t)NumSrc));
        ASORA_HIP_TRY(hipMemcpy(st.src_pos_sorted, ps.data(), sizeof(int32_t) * 3 * (size_t)NumSrc, hipMemcpyHostToDevice));
        ASORA_HIP_TRY(hipMemcpy(st.src_flux_sorted, fs.data(), sizeof(double) * (size_t)NumSrc, hipMemcpyHostToDevice));
        st.src_pos_sorted_host.swap(ps);
    }
    st.src_pos_host.assign(pos, pos + 3 * (size_t)NumSrc);
    st.num_src = NumSrc;
    st.src_generation += 1;
    return 0;
}

int asora_raytrace_device(double R, double sig, double dr, int src_begin, int src_count, double minlogtau,
                          double dlogtau, int NumTau)
{
    clear_error();
    if (int rc = require_init("raytrace_device")) return rc;
    return do_raytrace(R, sig, dr, src_begin, src_count, minlogtau, dlogtau, NumTau, nullptr);
}

int asora_raytrace_begin(double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau)
{
    clear_error();
    if (int rc = require_init("raytrace_begin")) return rc;
    return rt_begin(R, sig, dr, minlogtau, dlogtau, NumTau, nullptr, true);
}

int asora_raytrace_begin_planes(double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                                const int *runs, int nruns)
{
    clear_error();
    if (int rc = require_init("raytrace_begin_planes")) return rc;
    State &st = g_state;
    st.rt_open = false;
    if (!st.grid_valid[ASORA_GRID_NDENS]) return fail(4, "raytrace_begin_planes: density not on device");
    if (!st.grid_valid[ASORA_GRID_XH_AV]) return fail(4, "raytrace_begin_planes: xh_av not on device");
    if (!st.opt[ASORA_OPT_GREY_NOTABLES] && !st.tables) return fail(4, "raytrace_begin_planes: radiation tables not on device");
    if (!(R >= 0.0)) return fail(4, "raytrace_begin_planes: R must be >= 0");
    if (NumTau < 1 && !st.opt[ASORA_OPT_GREY_NOTABLES]) return fail(4, "raytrace_begin_planes: NumTau must be >= 1");
    if (!st.opt[ASORA_OPT_Z_TRANSPOSED]) return fail(4, "raytrace_begin_planes: needs the [k][j][i] twins (ASORA_OPT_Z_TRANSPOSED = 1)");
    if (st.opt[ASORA_OPT_HEATING]) return fail(4, "raytrace_begin_planes: no heating rates on this path");
    if (nruns < 0 || (nruns > 0 && !runs)) return fail(3, "raytrace_begin_planes: bad plane runs");
    for (int q = 0; q < nruns; ++q)
        if (runs[2 * q] < 0 || runs[2 * q + 1] < 0 || runs[2 * q] + runs[2 * q + 1] > st.N)
            return fail(3, "raytrace_begin_planes: plane run outside the mesh");
    ASORA_HIP_TRY(hipMemsetAsync(st.counters, 0, sizeof(unsigned long long) * COUNTER_FIELDS * COUNTER_SLOTS, st.stream));
    for (int q = 0; q < nruns; ++q)
        if (int rc = launch_prepare_range(st, runs[2 * q], runs[2 * q + 1], true, st.grid[ASORA_GRID_PHI_ION])) return rc;
    fill_rt_params(st.rt_params, R, sig, dr, minlogtau, dlogtau, NumTau);
    st.rt_heat = false;
    st.rt_pipelined = false;
    st.rt_by_planes = true;
    st.rt_open = true;
    return 0;
}

int asora_raytrace_range(int src_begin, int src_count)
{
    clear_error();
    if (int rc = require_init("raytrace_range")) return rc;
    return rt_range(src_begin, src_count);
}

int asora_raytrace_fold(int i_begin, int i_count)
{
    clear_error();
    if (int rc = require_init("raytrace_fold")) return rc;
    return rt_fold(i_begin, i_count);
}

void *asora_stream(void) { return (void *)g_state.stream; }

int asora_do_all_sources(double R, double *coldensh_out, double sig, double dr, const double *ndens,
                         const double *xh_av, double *phi_ion, int NumSrc, int m1, double minlogtau,
                         double dlogtau, int NumTau)
{
    (void)coldensh_out; (void)ndens;      // ignored by the reference too (raytracing.cu:116)
    clear_error();
    if (int rc = require_init("do_all_sources")) return rc;
    if (int rc = check_N("do_all_sources", m1)) return rc;
    if (!xh_av || !phi_ion) return fail(3, "do_all_sources: null xh_av / phi_ion");
    State &st = g_state;
    if (NumSrc > st.num_src)
        return fail(3, "do_all_sources: NumSrc=" + std::to_string(NumSrc) + " exceeds the " +
                           std::to_string(st.num_src) + " sources on the device");
    const size_t bytes = st.ncell * sizeof(double);
    {
        bool done = false;
        if (int rc = do_all_sources_pipelined(R, sig, dr, xh_av, phi_ion, NumSrc, minlogtau, dlogtau, NumTau, done)) return rc;
        if (done) return 0;
    }
    ASORA_HIP_TRY(hipMemcpyAsync(st.grid[ASORA_GRID_XH_AV], xh_av, bytes, hipMemcpyHostToDevice, st.stream)); // cu:117
    st.grid_valid[ASORA_GRID_XH_AV] = true;
    if (int rc = do_raytrace(R, sig, dr, 0, NumSrc, minlogtau, dlogtau, NumTau, nullptr)) return rc;
    ASORA_HIP_TRY(hipMemcpyAsync(phi_ion, st.grid[ASORA_GRID_PHI_ION], bytes, hipMemcpyDeviceToHost, st.stream)); // cu:146
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    return 0;
}

int asora_chemistry_device(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                           int *conv_flag, double *sum_xh1, double *sum_xh0)
{
    clear_error();
    if (int rc = require_init("chemistry_device")) return rc;
    State &st = g_state;
    static const int need[] = {ASORA_GRID_NDENS, ASORA_GRID_TEMP, ASORA_GRID_XH, ASORA_GRID_XH_AV, ASORA_GRID_PHI_ION};
    for (int g : need)
        if (!st.grid_valid[g]) return fail(4, "chemistry_device: grid " + std::to_string(g) + " holds no data");
    st.grid_valid[ASORA_GRID_XH_INTERMED] = true;
    ChemParams p;
    p.ncell = st.ncell;
    p.dt = dt; p.bh00 = bh00; p.albpow = albpow; p.colh0 = colh0; p.temph0 = temph0; p.abu_c = abu_c;
    p.ndens = st.grid[ASORA_GRID_NDENS]; p.temp = st.grid[ASORA_GRID_TEMP]; p.xh = st.grid[ASORA_GRID_XH];
    p.phi = st.grid[ASORA_GRID_PHI_ION];
    p.xh_av = st.grid[ASORA_GRID_XH_AV]; p.xh_intermed = st.grid[ASORA_GRID_XH_INTERMED];
    p.red_partial = st.red_partial; p.red_final = st.red_final; p.red_blocks = st.red_blocks;
    if (int rc = launch_chemistry(st, p, st.stream)) return rc;
    ASORA_HIP_TRY(hipMemcpyAsync(st.red_host, st.red_final, sizeof(double) * 3, hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    if (sum_xh1) *sum_xh1 = st.red_host[0];
    if (sum_xh0) *sum_xh0 = st.red_host[1];
    if (conv_flag) *conv_flag = (int)st.red_host[2];
    return 0;
}

int asora_chemistry_range(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                          int i_begin, int i_count, int first)
{
    clear_error();
    if (int rc = require_init("chemistry_range")) return rc;
    State &st = g_state;
    // (xh_intermed is only ever written by the pass: chemistry.f90:107)
    static const int need[] = {ASORA_GRID_NDENS, ASORA_GRID_TEMP, ASORA_GRID_XH, ASORA_GRID_XH_AV, ASORA_GRID_PHI_ION};
    for (int g : need)
        if (!st.grid_valid[g]) return fail(4, "chemistry_range: grid " + std::to_string(g) + " holds no data");
    st.grid_valid[ASORA_GRID_XH_INTERMED] = true;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N) return fail(4, "chemistry_range: bad plane range");
    if (i_count == 0 && !first) return 0;
    if (i_count == 0) {       // an empty first slab still resets the reductions
        ASORA_HIP_TRY(hipMemsetAsync(st.red_final, 0, sizeof(double) * 3, st.stream));
        return 0;
    }
    ChemTileParams p;
    p.N = st.N; p.i_begin = i_begin; p.i_end = i_begin + i_count;
    p.dt = dt; p.bh00 = bh00; p.albpow = albpow; p.colh0 = colh0; p.temph0 = temph0; p.abu_c = abu_c;
    p.ndens = st.grid[ASORA_GRID_NDENS]; p.temp = st.grid[ASORA_GRID_TEMP]; p.xh = st.grid[ASORA_GRID_XH];
    p.xh_av_in = st.grid[ASORA_GRID_XH_AV];
    p.gamma = st.grid[ASORA_GRID_PHI_ION];
    p.xh_av = st.grid[ASORA_GRID_XH_AV]; p.xh_intermed = st.grid[ASORA_GRID_XH_INTERMED];
    if (int rc = ensure_red_capacity(3 * chemistry_tile_blocks(st, st.N, i_count))) return rc;    // (sized for every range at init)
    p.red_partial = st.red_partial; p.red_final = st.red_final;
    p.accumulate = first ? 0 : 1;
    if (int rc = ensure_temp_probe(bh00, albpow, colh0, temph0)) return rc;
    set_uniform_temperature(p);
    return launch_chemistry_tiles(st, p, st.stream);
}

int asora_chemistry_finish(int *conv_flag, double *sum_xh1, double *sum_xh0)
{
    clear_error();
    if (int rc = require_init("chemistry_finish")) return rc;
    State &st = g_state;
    ASORA_HIP_TRY(hipMemcpyAsync(st.red_host, st.red_final, sizeof(double) * 3, hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    if (sum_xh1) *sum_xh1 = st.red_host[0];
    if (sum_xh0) *sum_xh0 = st.red_host[1];
    if (conv_flag) *conv_flag = (int)st.red_host[2];
    return 0;
}

void *asora_reduction_ptr(void) { return g_state.init ? (void *)g_state.red_final : nullptr; }

int c2ray_global_pass(double dt, const double *ndens, const double *temp, const double *xh, double *xh_av,
                      double *xh_intermed, const double *phi_ion, double bh00, double albpow, double colh0,
                      double temph0, double abu_c, int m1, int m2, int m3, int *conv_flag)
{
    clear_error();
    if (m1 < 1 || m2 < 1 || m3 < 1) return fail(3, "global_pass: bad mesh size");
    if (!ndens || !temp || !xh || !xh_av || !xh_intermed || !phi_ion) return fail(3, "global_pass: null grid");
    if (int rc = ensure_runtime()) return rc;
    State &st = g_state;
    const size_t ncell = (size_t)m1 * m2 * m3, bytes = ncell * sizeof(double);
    double *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const double *h[6] = {ndens, temp, xh, xh_av, xh_intermed, phi_ion};
    int rc = 0;
    auto cleanup = [&]() { for (auto *q : d) if (q) (void)hipFree(q); };
    for (int g = 0; g < 6 && !rc; ++g) {
        hipError_t e = hipMalloc(&d[g], bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(d[g], h[g], bytes, hipMemcpyHostToDevice, st.stream);
        if (e != hipSuccess) rc = fail(10, std::string("global_pass: ") + hipGetErrorString(e));
    }
    if (rc) { cleanup(); return rc; }
    ChemParams p;
    p.ncell = ncell;
    p.dt = dt; p.bh00 = bh00; p.albpow = albpow; p.colh0 = colh0; p.temph0 = temph0; p.abu_c = abu_c;
    p.ndens = d[0]; p.temp = d[1]; p.xh = d[2]; p.xh_av = d[3]; p.xh_intermed = d[4]; p.phi = d[5];
    p.red_partial = st.red_partial; p.red_final = st.red_final; p.red_blocks = st.red_blocks;
    rc = launch_chemistry(st, p, st.stream);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(xh_av, d[3], bytes, hipMemcpyDeviceToHost, st.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(xh_intermed, d[4], bytes, hipMemcpyDeviceToHost, st.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(st.red_host, st.red_final, sizeof(double) * 3, hipMemcpyDeviceToHost, st.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(st.stream);
        if (e != hipSuccess) rc = fail(10, std::string("global_pass: ") + hipGetErrorString(e));
    }
    if (!rc && conv_flag) *conv_flag = (int)st.red_host[2];
    cleanup();
    return rc;
}

// ---------------------------------------------------------------------------------------------
// libc2ray.raytracing.do_all_sources: the host driver of subbox.hip (do_all_sources / do_source,
// src/c2ray/raytracing.f90:52-249)
// ---------------------------------------------------------------------------------------------
int c2ray_do_all_sources(const double *normflux, const int32_t *srcpos, int max_subbox, int subboxsize,
                         double *coldensh_out, double sig, double dr, const double *ndens, const double *xh_av,
                         double *phi_ion, double *phi_heat, float loss_fraction,
                         const double *photo_thin_table, const double *photo_thick_table,
                         const double *heat_thin_table, const double *heat_thick_table,
                         double minlogtau, double dlogtau, double R_max_LLS,
                         int NumTau, int NumSrc, int m1, int m2, int m3,
                         int *sum_nbox, double *photon_loss)
{
    clear_error();
    State &st = g_state;
    const char *who = "c2ray_do_all_sources";
    if (m1 != m2 || m1 != m3) return fail(3, std::string(who) + ": the mesh must be cubic (raytracing.f90:174-175 use m1 for every axis)");
    if (NumSrc < 0 || (NumSrc > 0 && (!normflux || !srcpos))) return fail(3, std::string(who) + ": bad source arguments");
    if (!ndens || !xh_av || !phi_ion || !coldensh_out) return fail(3, std::string(who) + ": null grid");
    if (subboxsize < 1) return fail(3, std::string(who) + ": subboxsize must be >= 1");
    const bool grey = st.opt[ASORA_OPT_GREY_NOTABLES] != 0;
    if (!grey && (NumTau < 1 || !photo_thin_table || !photo_thick_table))
        return fail(3, std::string(who) + ": empty photo-ionisation tables");
    // Heating tables that are identically zero (what the reference's evolve3D passes, pyc2ray/evolve.py:193: "eventually
    // we'll add heating tables here") add exactly 0 to phi_heat: the grid is then neither uploaded, nor rated, nor
    // downloaded -- two 128 MiB transfers at 256^3 and the slower kernel variant for nothing.
    bool heat = !grey && phi_heat && heat_thin_table && heat_thick_table;
    if (heat) {
        bool any = false;
        for (int i = 0; i < NumTau && !any; ++i) any = heat_thin_table[i] != 0.0 || heat_thick_table[i] != 0.0;
        heat = any;
    }
    if (int rc = asora_device_init_auto(m1)) return rc;
    const int N = st.N;
    for (int s = 0; s < NumSrc; ++s)
        for (int ax = 0; ax < 3; ++ax)
            if (srcpos[3 * s + ax] < 1 || srcpos[3 * s + ax] > N)
                return fail(3, std::string(who) + ": source " + std::to_string(s + 1) + " lies outside the mesh (1-based " +
                                   std::to_string(srcpos[3 * s + ax]) + " on axis " + std::to_string(ax + 1) + ")");

    DeviceBuffers tmp;
    const size_t bytes = st.ncell * sizeof(double);
    // inputs: grids in Fortran order, sources 1-based
    if (int rc = asora_grid_to_device(ASORA_GRID_NDENS, ndens, N, 'F')) return rc;
    if (int rc = asora_grid_to_device(ASORA_GRID_XH_AV, xh_av, N, 'F')) return rc;
    if (heat) { if (int rc = asora_grid_to_device(ASORA_GRID_PHI_HEAT, phi_heat, N, 'F')) return rc; }

    int32_t *d_pos = nullptr; double *d_flux = nullptr; double2 *d_tables = nullptr;
    std::vector<int32_t> host_pos0;
    if (NumSrc > 0) {
        host_pos0.resize(3 * (size_t)NumSrc);
        std::vector<int32_t> &pos0 = host_pos0;
        for (size_t q = 0; q < pos0.size(); ++q) pos0[q] = srcpos[q] - 1;
        if (int rc = tmp.alloc(d_pos, pos0.size())) return rc;
        if (int rc = tmp.alloc(d_flux, (size_t)NumSrc)) return rc;
        ASORA_HIP_TRY(hipMemcpy(d_pos, pos0.data(), pos0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        ASORA_HIP_TRY(hipMemcpy(d_flux, normflux, (size_t)NumSrc * sizeof(double), hipMemcpyHostToDevice));
    }
    const int len = grey ? 1 : NumTau;
    {   // [thick | thin | heat thick | heat thin] as pairs {T[i], T[i+1]-T[i]} (see asora_photo_table_to_device)
        std::vector<double2> pairs(std::max<size_t>(4 * (size_t)len, 16), double2{0.0, 0.0});
        const double *src[4] = {photo_thick_table, photo_thin_table, heat ? heat_thick_table : nullptr,
                                heat ? heat_thin_table : nullptr};
        for (int t = 0; t < 4 && !grey; ++t)
            if (src[t]) pack_rate_table(pairs.data(), t, src[t], len);
        if (int rc = tmp.alloc(d_tables, pairs.size())) return rc;
        ASORA_HIP_TRY(hipMemcpy(d_tables, pairs.data(), pairs.size() * sizeof(double2), hipMemcpyHostToDevice));
    }

    SubboxCall c;
    c.max_subbox = max_subbox; c.subboxsize = subboxsize; c.loss_fraction = loss_fraction;
    c.sig = sig; c.dr = dr; c.R = R_max_LLS; c.minlogtau = minlogtau; c.dlogtau = dlogtau; c.NumTau = NumTau;
    c.table_len = len; c.tables = d_tables; c.src_pos = d_pos; c.src_flux = d_flux;
    c.host_pos = host_pos0.empty() ? nullptr : host_pos0.data();
    c.src_begin = 0; c.src_count = NumSrc;
    c.heat = heat; c.keep_heat = true;                  // phi_heat is intent(inout): added onto what was uploaded
    c.dump = st.staging;                                // column densities of the last source
    long long total_nbox = 0;
    double total_loss = 0.0;
    if (int rc = subbox_core(c, total_nbox, total_loss)) return rc;

    // the last source's column densities sit in the staging grid, which the 'F' download path below reuses:
    // take them out first, through nHI's transposed half (free once the sweep is over)
    if (int rc = launch_transpose(st, st.staging, st.nhi_t, N)) return rc;
    ASORA_HIP_TRY(hipMemcpyAsync(coldensh_out, st.nhi_t, bytes, hipMemcpyDeviceToHost, st.stream));
    if (int rc = asora_grid_to_host(ASORA_GRID_PHI_ION, phi_ion, N, 'F')) return rc;
    if (heat) { if (int rc = asora_grid_to_host(ASORA_GRID_PHI_HEAT, phi_heat, N, 'F')) return rc; }
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    if (sum_nbox) *sum_nbox = (int)total_nbox;
    if (photon_loss) *photon_loss = total_loss;
    return 0;
}

int asora_device_init_auto(int N)
{
    clear_error();
    State &st = g_state;
    if (!st.init || (st.auto_init && st.N != N)) {
        // stateless for the caller, like the f2py functions it serves: the library sets itself up for this mesh
        if (int rc = asora_device_init(N, 1)) return rc;
        st.auto_init = true;
        return 0;
    }
    return check_N("device_init_auto", N);
}

int asora_subbox_raytrace_device(int max_subbox, int subboxsize, float loss_fraction, double R_max_LLS, double sig, double dr,
                                 double minlogtau, double dlogtau, int NumTau, int src_begin, int src_count,
                                 int *sum_nbox, double *photon_loss)
{
    clear_error();
    if (int rc = require_init("subbox_raytrace_device")) return rc;
    State &st = g_state;
    const char *who = "subbox_raytrace_device";
    if (!st.grid_valid[ASORA_GRID_NDENS]) return fail(4, std::string(who) + ": density not on device");
    if (!st.grid_valid[ASORA_GRID_XH_AV]) return fail(4, std::string(who) + ": xh_av not on device");
    if (subboxsize < 1) return fail(3, std::string(who) + ": subboxsize must be >= 1");
    const bool grey = st.opt[ASORA_OPT_GREY_NOTABLES] != 0;
    if (!grey && (!st.tables || NumTau < 1)) return fail(4, std::string(who) + ": radiation tables not on device");
    if (src_begin < 0 || src_count < 0 || src_begin + src_count > st.num_src)
        return fail(4, std::string(who) + ": source range outside the uploaded sources");
    const bool heat = !grey && st.opt[ASORA_OPT_HEATING] != 0;
    if (heat && !st.have_heat_tables) return fail(4, std::string(who) + ": heating requested but no heating tables on device");
    SubboxCall c;
    c.max_subbox = max_subbox; c.subboxsize = subboxsize; c.loss_fraction = loss_fraction;
    c.sig = sig; c.dr = dr; c.R = R_max_LLS; c.minlogtau = minlogtau; c.dlogtau = dlogtau; c.NumTau = NumTau;
    c.table_len = st.table_len > 0 ? st.table_len : 1; c.tables = st.tables;
    c.src_pos = st.src_pos; c.src_flux = st.src_flux; c.src_begin = src_begin; c.src_count = src_count;
    c.host_pos = st.src_pos_host.empty() ? nullptr : st.src_pos_host.data();
    c.heat = heat; c.keep_heat = false; c.dump = nullptr;
    long long total_nbox = 0;
    double total_loss = 0.0;
    if (int rc = subbox_core(c, total_nbox, total_loss)) return rc;
    if (sum_nbox) *sum_nbox = (int)total_nbox;
    if (photon_loss) *photon_loss = total_loss;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The evolve loop on the device (pyc2ray/evolve.py:168-240): raytrace -> fused chemistry -> convergence test,
// nothing in between and nothing on the host
// ---------------------------------------------------------------------------------------------
static int evolve_begin_impl(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                             double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                             int src_begin, int src_count, double conv_criterion, double convergence_fraction,
                             bool slab, int own_begin, int own_count)
{
    clear_error();
    if (int rc = require_init("evolve_begin")) return rc;
    State &st = g_state;
    st.ev_open = false;
    if (slab) {
        if (own_begin < 0 || own_count < 0 || own_begin + own_count > st.N) return fail(4, "evolve_begin_slab: bad range of own planes");
        if (!st.opt[ASORA_OPT_Z_TRANSPOSED]) return fail(4, "evolve_begin_slab: needs the [k][j][i] twins (ASORA_OPT_Z_TRANSPOSED = 1)");
    }
    static const int need[] = {ASORA_GRID_NDENS, ASORA_GRID_TEMP, ASORA_GRID_XH};
    for (int g : need)
        if (!st.grid_valid[g]) return fail(4, "evolve_begin: grid " + std::to_string(g) + " holds no data");
    if (!st.opt[ASORA_OPT_GREY_NOTABLES] && !st.tables)
        return fail(4, "evolve_begin: radiation tables not on device (photo_table_to_device)");
    if (!(R >= 0.0)) return fail(4, "evolve_begin: R must be >= 0");
    if (NumTau < 1 && !st.opt[ASORA_OPT_GREY_NOTABLES]) return fail(4, "evolve_begin: NumTau must be >= 1");
    if (src_begin < 0 || src_count < 0 || src_begin + src_count > st.num_src)
        return fail(4, "evolve_begin: source range outside the " + std::to_string(st.num_src) + " uploaded sources");
    if (st.opt[ASORA_OPT_HEATING]) return fail(4, "evolve_begin: the fused loop carries no heating rates (use raytrace_device)");

    if (int rc = ensure_temp_probe(bh00, albpow, colh0, temph0)) return rc;
    const size_t bytes = st.ncell * sizeof(double);
    // two accumulator pairs (State::acc): the first trace needs a zeroed pair; every fused pass zeroes the pair the next
    // trace adds into, iterations beyond convergence touch nothing
    if (!st.ev_sets_known) {          // iterations were enqueued and never polled: which pair holds what is not known
        ASORA_HIP_TRY(hipMemsetAsync(st.acc, 0, 4 * bytes, st.stream));                      // raytracing.cu:113
        st.ev_clean[0] = st.ev_clean[1] = true;
        st.ev_sets_known = true;
    }
    // Which lines of the accumulators this step's sources can touch (State::reach_mask): rebuilt when the sources, their range
    // or the radius change, and USED while at least 45 % of the lines are out of reach (counted then).  Measured at 256^3 with 1000
    // sources (profiles/r04_ab_reach_mask.txt): r_RT = 8 (20 % of the lines reached) pass -12 ... -19 %, 12 (46 %) -3 ... -7 %,
    // 16 (74 %) +2 ... +5 %, 32 (100 %) +10 %: where the spheres cover the box the two mask bytes per cell only cost -- none of the
    // BASELINE configurations gains, sparse runs (few sources, small radii) do.  Whenever the set of lines the
    // passes zero changes, BOTH pairs are zeroed once: the dirty pair of the previous step may hold rates where the new
    // sources do not reach.  Not for traces that cover (nearly) the whole box, nor with ASORA_REACH_MASK=0 (2: whenever built).
    if (slab) {
        // multi-GPU: the pass sweeps the own planes only and the out-box folds zero the foreign ones (asora_evolve_slab_fold_out);
        // which planes those are changes with the plan, so a step simply starts from two zeroed pairs (256 MiB of stores at
        // 256^3, once per time step), and no reach mask
        if (!(st.ev_clean[0] && st.ev_clean[1])) {
            ASORA_HIP_TRY(hipMemsetAsync(st.acc, 0, 4 * bytes, st.stream));
            st.ev_clean[0] = st.ev_clean[1] = true;
        }
        st.reach_in_use = false;
    } else {
        static const int mode = []() { const char *v = getenv("ASORA_REACH_MASK"); return v ? atoi(v) : 1; }();
        const bool possible = mode != 0 && std::isfinite(R) && 2.0 * R + 2.0 < (double)st.N && st.opt[ASORA_OPT_Z_TRANSPOSED] != 0;
        const bool same = st.reach_valid && st.reach_src_generation == st.src_generation && st.reach_src_begin == src_begin &&
                          st.reach_src_count == src_count && st.reach_R == R;
        // where the spheres together hold more cells than the box, (nearly) every line is reached: the mask can not pay and is
        // not built (in a cosmological run R changes every step, and every build ends with a blocking read-back)
        // (forced use, ASORA_REACH_MASK=2, always builds it: a mask in use must be the mask of THIS source set and radius)
        const bool covers = possible && mode != 2 && (double)src_count * (4.0 / 3.0) * 3.14159265358979 * R * R * R >= (double)st.ncell;
        if (possible && !same && covers) {
            st.reach_pays = false;
            st.reach_valid = true; st.reach_src_generation = st.src_generation; st.reach_src_begin = src_begin;
            st.reach_src_count = src_count; st.reach_R = R;
        }
        if (possible && !same && !covers) {
            const size_t one = (size_t)st.N * st.N * ((st.N + 7) / 8);
            if (!st.reach_mask) { ASORA_HIP_TRY(hipMalloc(&st.reach_mask, 2 * one)); st.reach_bytes = one; }
            if (!st.reach_count_dev) ASORA_HIP_TRY(hipMalloc(&st.reach_count_dev, sizeof(unsigned long long)));
            if (int rc = launch_reach_mask(st, st.src_pos, src_begin, src_count, R, st.reach_mask, one)) return rc;
            if (int rc = launch_reach_count(st, st.reach_mask, 2 * one, st.reach_count_dev)) return rc;
            unsigned long long marked = 0;
            ASORA_HIP_TRY(hipMemcpyAsync(&marked, st.reach_count_dev, sizeof marked, hipMemcpyDeviceToHost, st.stream));
            ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
            st.reach_pays = (double)marked <= 0.55 * (double)(2 * one);
            st.reach_valid = true; st.reach_src_generation = st.src_generation; st.reach_src_begin = src_begin;
            st.reach_src_count = src_count; st.reach_R = R;
        }
        const bool wanted = possible && (st.reach_pays || mode == 2);
        if ((wanted && !same) || (wanted != st.reach_in_use)) {      // the set of lines the passes zero changes: start from zeroed pairs
            if (!(st.ev_clean[0] && st.ev_clean[1])) {
                ASORA_HIP_TRY(hipMemsetAsync(st.acc, 0, 4 * bytes, st.stream));
                st.ev_clean[0] = st.ev_clean[1] = true;
            }
        }
        st.reach_in_use = wanted;
    }
    if (!st.ev_clean[0] && !st.ev_clean[1]) return fail(11, "evolve_begin: no clean accumulator pair (internal error)");
    st.ev_base = st.ev_clean[0] ? 0 : 1;
    st.ev_folded_iter = 0;
    if (!st.ev_status) {
        ASORA_HIP_TRY(hipMalloc(&st.ev_status, sizeof(EvolveStatus)));
        ASORA_HIP_TRY(hipHostMalloc(&st.ev_host, sizeof(EvolveStatus), hipHostMallocDefault));
    }
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));            // ev_host may still be the target of an earlier poll
    std::memset(st.ev_host, 0, sizeof(EvolveStatus));
    st.ev_host->prev1 = 2.0 * (double)st.ncell;                // evolve.py:130-131
    st.ev_host->prev0 = 2.0 * (double)st.ncell;
    st.ev_host->conv_criterion = conv_criterion;
    st.ev_host->conv_fraction = convergence_fraction;
    ASORA_HIP_TRY(hipMemcpyAsync(st.ev_status, st.ev_host, sizeof(EvolveStatus), hipMemcpyHostToDevice, st.stream));
    ASORA_HIP_TRY(hipMemsetAsync(st.counters, 0, sizeof(unsigned long long) * COUNTER_FIELDS * COUNTER_SLOTS, st.stream));
    // xh_av = copy(xh) (evolve.py:136) is not materialised: nHI of the first trace is formed from xh and the first
    // chemistry pass takes xh as its starting xh_av; xh_intermed (evolve.py:137) is only ever written
    if (int rc = launch_prepare_nhi_from(st, st.grid[ASORA_GRID_XH], st.opt[ASORA_OPT_Z_TRANSPOSED] != 0)) return rc;

    fill_rt_params(st.ev_rt, R, sig, dr, minlogtau, dlogtau, NumTau);
    st.ev_rt.phi = st.acc + (size_t)st.ev_base * 2 * st.ncell;       // (each iteration sets its own pair, asora_evolve_enqueue)
    st.ev_rt.done_flag = &st.ev_status->done;
    st.ev_rt.src_begin = src_begin; st.ev_rt.src_count = src_count; st.ev_rt.shape_src_count = src_count;
    if (src_begin == 0 && src_count == st.num_src && st.src_pos_sorted) { st.ev_rt.src_pos = st.src_pos_sorted; st.ev_rt.src_flux = st.src_flux_sorted; }
    st.ev_src_begin = src_begin; st.ev_src_count = src_count;
    st.ev_chem[0] = dt; st.ev_chem[1] = bh00; st.ev_chem[2] = albpow; st.ev_chem[3] = colh0; st.ev_chem[4] = temph0;
    st.ev_chem[5] = abu_c;
    st.ev_first = true;
    st.ev_reported = 0;
    st.ev_enqueued = 0;
    st.ev_slab = slab; st.ev_own_begin = own_begin; st.ev_own_count = own_count; st.ev_slab_passed = false;
    st.ev_rates_in_outbox = false; st.ev_folded_all = false;
    st.ev_open = true;
    return 0;
}

int asora_evolve_begin(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                       double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                       int src_begin, int src_count, double conv_criterion, double convergence_fraction)
{
    return evolve_begin_impl(dt, bh00, albpow, colh0, temph0, abu_c, R, sig, dr, minlogtau, dlogtau, NumTau, src_begin, src_count,
                             conv_criterion, convergence_fraction, false, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// The same loop when the sources are sharded over several GPUs (pyc2ray/evolve.py:249-498; pyc2ray_amd/dist.py SlabPlan): a
// rank traces ITS sources, owns the chemistry of ITS planes, and one iteration is the sequence
//   asora_evolve_slab_trace      (once, or per chunk of sources)     -> rates into the iteration's accumulator pair
//   asora_evolve_slab_fold_out   per run of foreign planes           -> out-box planes to send; the other pair zeroed there
//   asora_evolve_slab_add        per run received from another rank  -> added to the own planes of the pair
//   asora_evolve_slab_pass                                            -> the fused pass of the one-GPU loop on the own planes
//   asora_evolve_slab_nhi        per run of xh_av received           -> nHI of the halo planes for the next trace
//   asora_evolve_slab_close                                           -> convergence test on the sums over all ranks
// all asynchronous on the library's stream and all gated by the status block's `done`, so that -- as on one GPU -- a caller
// enqueues several iterations and reads the status back once (asora_evolve_poll; it folds the own rates into PHI_ION).
// ---------------------------------------------------------------------------------------------
int asora_evolve_begin_slab(double dt, double bh00, double albpow, double colh0, double temph0, double abu_c,
                            double R, double sig, double dr, double minlogtau, double dlogtau, int NumTau,
                            int src_begin, int src_count, double conv_criterion, double convergence_fraction,
                            int own_begin, int own_count)
{
    return evolve_begin_impl(dt, bh00, albpow, colh0, temph0, abu_c, R, sig, dr, minlogtau, dlogtau, NumTau, src_begin, src_count,
                             conv_criterion, convergence_fraction, true, own_begin, own_count);
}

static int require_slab(const char *who)
{
    if (int rc = require_init(who)) return rc;
    if (!g_state.ev_open || !g_state.ev_slab)
        return fail(4, std::string(who) + ": no multi-GPU evolve step in progress (call asora_evolve_begin_slab)");
    if (g_state.ev_enqueued - g_state.ev_reported + 1 > EVOLVE_HIST)
        return fail(4, std::string(who) + ": " + std::to_string(EVOLVE_HIST) + " iterations enqueued since the last asora_evolve_poll (poll first)");
    return 0;
}
static double *slab_pair(int which)            // 0: the pair the current iteration traces into, 1: the other one
{
    State &st = g_state;
    const int set = ((st.ev_base + st.ev_enqueued) & 1) ^ which;
    return st.acc + (size_t)set * 2 * st.ncell;
}

int asora_evolve_slab_trace(int src_begin, int src_count)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_trace")) return rc;
    State &st = g_state;
    if (st.ev_slab_passed) return fail(4, "evolve_slab_trace: the iteration's pass has been enqueued already (close it first)");
    if (src_begin < st.ev_src_begin || src_count < 0 || src_begin + src_count > st.ev_src_begin + st.ev_src_count)
        return fail(4, "evolve_slab_trace: source range outside the step's sources");
    if (src_count == 0) return 0;
    st.ev_sets_known = false;
    RtParams p = st.ev_rt;
    p.phi = slab_pair(0);
    p.src_begin = src_begin; p.src_count = src_count;          // (shape_src_count stays the rank's whole share: one launch shape)
    if (!(src_begin == 0 && src_count == st.num_src)) { p.src_pos = st.src_pos; p.src_flux = st.src_flux; }
    return launch_raytrace(st, p, false, false);
}

int asora_evolve_slab_fold_out(int i_begin, int i_count)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_fold_out")) return rc;
    State &st = g_state;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N) return fail(4, "evolve_slab_fold_out: bad plane range");
    if (i_count > 0 && i_begin < st.ev_own_begin + st.ev_own_count && st.ev_own_begin < i_begin + i_count)
        return fail(4, "evolve_slab_fold_out: the range holds planes this rank owns (their rates stay: the pass folds them)");
    st.ev_sets_known = false;
    double *cur = slab_pair(0), *nxt = slab_pair(1);
    return launch_fold_out(st, cur, cur + st.ncell, st.staging, nxt, nxt + st.ncell, i_begin, i_count, &st.ev_status->done);
}

// The full-grid exchange (pyc2ray/evolve.py:433-437: every rank all-reduces the rate grid) on the same loop: ALL planes folded
// into the out-box, which the caller then sums over the ranks in place; the pass reads the out-box (one layout, nothing left to
// fold) and keeps the summed rates in PHI_ION itself.  (An all-reduce is not gated by `done`: iterations enqueued beyond convergence
// sum the stale out-box once more.  The pass is gated, so PHI_ION keeps what the last iteration carried out has read.)
int asora_evolve_slab_fold_all(void)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_fold_all")) return rc;
    State &st = g_state;
    if (st.ev_slab_passed) return fail(4, "evolve_slab_fold_all: the iteration's pass has been enqueued already");
    if (st.ev_own_begin != 0 || st.ev_own_count != st.N)
        return fail(4, "evolve_slab_fold_all: the step must own every plane (asora_evolve_begin_slab(..., 0, N)): the chemistry is replicated");
    st.ev_sets_known = false;
    st.ev_rates_in_outbox = true; st.ev_folded_all = true;
    double *cur = slab_pair(0), *nxt = slab_pair(1);
    // (the pass zeroes the other pair's [i][j][k] layout as it goes; the transposed layout is zeroed here)
    return launch_fold_out(st, cur, cur + st.ncell, st.staging, nullptr, nxt + st.ncell, 0, st.N, &st.ev_status->done);
}

void *asora_evolve_slab_outbox(void) { return g_state.init ? (void *)g_state.staging : nullptr; }

int asora_evolve_slab_outbox_from_host(int i_begin, int i_count, const double *host)
{
    clear_error();
    if (int rc = require_init("evolve_slab_outbox_from_host")) return rc;
    State &st = g_state;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N || (i_count > 0 && !host)) return fail(3, "evolve_slab_outbox_from_host: bad arguments");
    if (i_count == 0) return 0;
    const size_t plane = (size_t)st.N * st.N;
    ASORA_HIP_TRY(hipMemcpyAsync(st.staging + (size_t)i_begin * plane, host, (size_t)i_count * plane * sizeof(double), hipMemcpyHostToDevice, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));              // (the host buffer may be pageable)
    return 0;
}

int asora_evolve_slab_outbox_to_host(int i_begin, int i_count, double *host)
{
    clear_error();
    if (int rc = require_init("evolve_slab_outbox_to_host")) return rc;
    State &st = g_state;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N || (i_count > 0 && !host)) return fail(3, "evolve_slab_outbox_to_host: bad arguments");
    if (i_count == 0) return 0;
    const size_t plane = (size_t)st.N * st.N;
    ASORA_HIP_TRY(hipMemcpyAsync(host, st.staging + (size_t)i_begin * plane, (size_t)i_count * plane * sizeof(double), hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    return 0;
}

int asora_evolve_slab_add(int i_begin, int i_count, const double *dev_planes)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_add")) return rc;
    State &st = g_state;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N || (i_count > 0 && !dev_planes)) return fail(4, "evolve_slab_add: bad arguments");
    if (st.ev_slab_passed) return fail(4, "evolve_slab_add: the iteration's pass has been enqueued already");
    if (i_count > 0 && (i_begin < st.ev_own_begin || i_begin + i_count > st.ev_own_begin + st.ev_own_count))
        return fail(4, "evolve_slab_add: rates received for planes this rank does not own");
    st.ev_sets_known = false;
    const size_t plane = (size_t)st.N * st.N;
    return launch_add_planes(st, slab_pair(0) + (size_t)i_begin * plane, dev_planes, (size_t)i_count * plane, &st.ev_status->done);
}

int asora_evolve_slab_add_host(int i_begin, int i_count, const double *host_planes)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_add_host")) return rc;
    State &st = g_state;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N || (i_count > 0 && !host_planes)) return fail(4, "evolve_slab_add_host: bad arguments");
    if (i_count == 0) return 0;
    // through the out-box: what is added belongs to planes this rank owns, what the out-box holds to planes it does not
    const size_t plane = (size_t)st.N * st.N;
    double *tmp = st.staging + (size_t)i_begin * plane;
    ASORA_HIP_TRY(hipMemcpyAsync(tmp, host_planes, (size_t)i_count * plane * sizeof(double), hipMemcpyHostToDevice, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));              // (the host buffer may be pageable)
    return asora_evolve_slab_add(i_begin, i_count, tmp);
}

int asora_evolve_slab_pass(void)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_pass")) return rc;
    State &st = g_state;
    if (st.ev_slab_passed) return fail(4, "evolve_slab_pass: already enqueued for this iteration");
    if (st.ev_rates_in_outbox && !st.ev_folded_all)
        return fail(4, "evolve_slab_pass: this step exchanges whole grids (asora_evolve_slab_fold_all), and this iteration's fold has not been enqueued");
    st.ev_sets_known = false;
    st.ev_slab_passed = true;
    st.grid_valid[ASORA_GRID_XH_AV] = st.grid_valid[ASORA_GRID_XH_INTERMED] = true;
    st.grid_valid[ASORA_GRID_PHI_ION] = false;
    if (st.ev_own_count == 0) {           // nothing to own (more ranks than planes): this rank's share of the sums is zero
        ASORA_HIP_TRY(hipMemsetAsync(st.red_final, 0, sizeof(double) * 3, st.stream));
        return 0;
    }
    double *cur = slab_pair(0), *nxt = slab_pair(1);
    ChemTileParams c;
    c.N = st.N; c.i_begin = st.ev_own_begin; c.i_end = st.ev_own_begin + st.ev_own_count;
    c.dt = st.ev_chem[0]; c.bh00 = st.ev_chem[1]; c.albpow = st.ev_chem[2]; c.colh0 = st.ev_chem[3];
    c.temph0 = st.ev_chem[4]; c.abu_c = st.ev_chem[5];
    c.ndens = st.grid[ASORA_GRID_NDENS]; c.temp = st.grid[ASORA_GRID_TEMP]; c.xh = st.grid[ASORA_GRID_XH];
    c.xh_av_in = st.ev_first ? st.grid[ASORA_GRID_XH] : st.grid[ASORA_GRID_XH_AV];
    c.gamma = cur; c.gamma_t = cur + st.ncell; c.phi_out = nullptr;
    if (st.ev_rates_in_outbox) { c.gamma = st.staging; c.gamma_t = nullptr; c.phi_out = st.grid[ASORA_GRID_PHI_ION]; }
    c.zero_a = nxt; c.zero_t = nxt + st.ncell;
    c.xh_av = st.grid[ASORA_GRID_XH_AV]; c.xh_intermed = st.grid[ASORA_GRID_XH_INTERMED];
    c.nhi = st.nhi; c.nhi_t = st.nhi_t;
    if (int rc = ensure_red_capacity(3 * chemistry_tile_blocks(st, st.N, st.ev_own_count))) return rc;    // (sized for every range at init)
    c.red_partial = st.red_partial; c.red_final = st.red_final;
    c.status = st.ev_status; c.local_sums = true;
    c.fold = !st.ev_rates_in_outbox; c.emit = true;
    set_uniform_temperature(c);
    return launch_chemistry_tiles(st, c, st.stream);
}

int asora_evolve_slab_nhi(int i_begin, int i_count)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_nhi")) return rc;
    State &st = g_state;
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N) return fail(4, "evolve_slab_nhi: bad plane range");
    return launch_prepare_range(st, i_begin, i_count, false, nullptr, &st.ev_status->done);
}

int asora_evolve_slab_close(const double *host_sums)
{
    clear_error();
    if (int rc = require_slab("evolve_slab_close")) return rc;
    State &st = g_state;
    if (!st.ev_slab_passed) return fail(4, "evolve_slab_close: the iteration's pass has not been enqueued");
    if (host_sums) {          // summed over the ranks on the host (gloo rehearsals, mpi4py): {sum x, sum 1-x, conv_flag}
        ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
        std::memcpy(st.red_host, host_sums, sizeof(double) * 3);
        ASORA_HIP_TRY(hipMemcpyAsync(st.red_final, st.red_host, sizeof(double) * 3, hipMemcpyHostToDevice, st.stream));
    }
    if (int rc = launch_convergence_test(st, st.red_final, st.ev_status)) return rc;
    st.ev_first = false;
    st.ev_slab_passed = false; st.ev_folded_all = false;
    st.ev_enqueued += 1;
    return 0;
}

int asora_evolve_enqueue(int iterations)
{
    clear_error();
    if (int rc = require_init("evolve_enqueue")) return rc;
    State &st = g_state;
    if (!st.ev_open) return fail(4, "evolve_enqueue: no evolve step in progress (call asora_evolve_begin)");
    if (st.ev_slab) return fail(4, "evolve_enqueue: the step was begun with asora_evolve_begin_slab (use the asora_evolve_slab_* calls)");
    if (iterations < 1 || iterations > EVOLVE_HIST / 2) return fail(3, "evolve_enqueue: between 1 and 32 iterations per call");
    // the per-iteration history is a ring of EVOLVE_HIST rows on the device: rows not yet handed out by asora_evolve_poll
    // must not be overwritten
    if (st.ev_enqueued - st.ev_reported + iterations > EVOLVE_HIST)
        return fail(4, "evolve_enqueue: " + std::to_string(st.ev_enqueued - st.ev_reported) + " iterations enqueued since the last "
                           "asora_evolve_poll; the history ring holds " + std::to_string(EVOLVE_HIST) + " (poll first)");
    st.ev_sets_known = false;                        // until the next poll tells how many of these were carried out
    for (int it = 0; it < iterations; ++it) {
        // iteration k = ev_enqueued + it + 1 of the step (as long as the step has not converged: then nothing runs anyway)
        const int set = (st.ev_base + st.ev_enqueued + it) & 1;
        double *acc_cur = st.acc + (size_t)set * 2 * st.ncell, *acc_next = st.acc + (size_t)(set ^ 1) * 2 * st.ncell;
        if (st.ev_src_count > 0) {
            RtParams p = st.ev_rt;
            p.phi = acc_cur;
            if (int rc = launch_raytrace(st, p, false, false)) return rc;
        }
        ChemTileParams c;
        c.N = st.N; c.i_begin = 0; c.i_end = st.N;
        c.dt = st.ev_chem[0]; c.bh00 = st.ev_chem[1]; c.albpow = st.ev_chem[2]; c.colh0 = st.ev_chem[3];
        c.temph0 = st.ev_chem[4]; c.abu_c = st.ev_chem[5];
        c.ndens = st.grid[ASORA_GRID_NDENS]; c.temp = st.grid[ASORA_GRID_TEMP]; c.xh = st.grid[ASORA_GRID_XH];
        c.xh_av_in = st.ev_first ? st.grid[ASORA_GRID_XH] : st.grid[ASORA_GRID_XH_AV];
        c.gamma = acc_cur; c.gamma_t = acc_cur + st.ncell; c.phi_out = nullptr;
        {   // A/B only (tools/ab_chem_store.sh): also store the folded rates every iteration, as round 2 did
            static const bool store_always = getenv("ASORA_DIAG_STORE_PHI") != nullptr;
            if (store_always) c.phi_out = st.grid[ASORA_GRID_PHI_ION];
        }
        c.zero_a = acc_next; c.zero_t = acc_next + st.ncell;
        if (st.reach_in_use) { c.reach_a = st.reach_mask; c.reach_t = st.reach_mask + st.reach_bytes; }
        c.xh_av = st.grid[ASORA_GRID_XH_AV]; c.xh_intermed = st.grid[ASORA_GRID_XH_INTERMED];
        c.nhi = st.nhi; c.nhi_t = st.nhi_t;
        c.red_partial = st.red_partial; c.red_final = st.red_final;
        c.status = st.ev_status;
        c.fold = true; c.emit = true;
        set_uniform_temperature(c);
        if (int rc = launch_chemistry_tiles(st, c, st.stream)) return rc;
        st.ev_first = false;
    }
    st.ev_enqueued += iterations;
    st.grid_valid[ASORA_GRID_XH_AV] = st.grid_valid[ASORA_GRID_XH_INTERMED] = true;
    st.grid_valid[ASORA_GRID_PHI_ION] = false;       // until asora_evolve_poll folds the last iteration's accumulators
    return 0;
}

int asora_evolve_poll(int *niter, int *converged, double *history, int history_rows, int *rows_written)
{
    clear_error();
    if (int rc = require_init("evolve_poll")) return rc;
    State &st = g_state;
    if (!st.ev_open) return fail(4, "evolve_poll: no evolve step in progress (call asora_evolve_begin)");
    ASORA_HIP_TRY(hipMemcpyAsync(st.ev_host, st.ev_status, sizeof(EvolveStatus), hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    const EvolveStatus &h = *st.ev_host;
    // The rates of the last iteration carried out sit, unfolded, in its accumulator pair; the other pair is zero (the pass
    // of that iteration zeroed it; iterations enqueued beyond convergence did nothing).  Fold them into PHI_ION now.
    if (h.niter > 0) {
        const int set = (st.ev_base + h.niter - 1) & 1;
        if (st.ev_slab && st.ev_rates_in_outbox) st.ev_folded_iter = h.niter;      // (the pass has kept the summed rates in PHI_ION)
        if (st.ev_folded_iter != h.niter) {
            const double *a = st.acc + (size_t)set * 2 * st.ncell;
            if (int rc = launch_fold_sum(st, a, a + st.ncell, st.grid[ASORA_GRID_PHI_ION])) return rc;
            st.ev_folded_iter = h.niter;
        }
        st.grid_valid[ASORA_GRID_PHI_ION] = true;
        st.ev_clean[set] = false; st.ev_clean[set ^ 1] = true;
    }
    st.ev_sets_known = true;
    int rows = 0;
    for (int it = st.ev_reported; it < h.niter && history && rows < history_rows; ++it, ++rows)
        for (int q = 0; q < 5; ++q) history[5 * rows + q] = h.hist[it % EVOLVE_HIST][q];
    // everything enqueued has run by now (iterations enqueued beyond convergence did nothing and never will)
    st.ev_enqueued = h.niter;
    if (history) st.ev_reported += rows;
    else st.ev_reported = h.niter;           // a caller that does not ask for the rows gives them up
    if (rows_written) *rows_written = rows;
    if (niter) *niter = h.niter;
    if (converged) *converged = h.done;
    return 0;
}

// Contiguous runs of i-planes of a grid to / from the host (C order: plane i is N*N consecutive doubles).  What a
// multi-GPU rank exchanges are such runs (the planes its sources reach, the planes whose chemistry it owns).
int asora_planes_to_host(int which, int i_begin, int i_count, double *host)
{
    clear_error();
    if (int rc = require_init("planes_to_host")) return rc;
    State &st = g_state;
    if (which < 0 || which >= ASORA_GRID_COUNT) return fail(3, "planes_to_host: bad grid selector");
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N) return fail(3, "planes_to_host: bad plane range");
    if (i_count == 0) return 0;
    if (!host) return fail(3, "planes_to_host: null host pointer");
    if (!st.grid_valid[which]) return fail(3, "planes_to_host: grid " + std::to_string(which) + " holds no data");
    const size_t plane = (size_t)st.N * st.N;
    ASORA_HIP_TRY(hipMemcpyAsync(host, st.grid[which] + (size_t)i_begin * plane, (size_t)i_count * plane * sizeof(double),
                                 hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    return 0;
}

int asora_planes_to_device(int which, int i_begin, int i_count, const double *host)
{
    clear_error();
    if (int rc = require_init("planes_to_device")) return rc;
    State &st = g_state;
    if (which < 0 || which >= ASORA_GRID_COUNT) return fail(3, "planes_to_device: bad grid selector");
    if (i_begin < 0 || i_count < 0 || i_begin + i_count > st.N) return fail(3, "planes_to_device: bad plane range");
    if (i_count == 0) return 0;
    if (!host) return fail(3, "planes_to_device: null host pointer");
    if (which == ASORA_GRID_PHI_HEAT) { if (int rc = ensure_heat_grid()) return rc; }
    const size_t plane = (size_t)st.N * st.N;
    ASORA_HIP_TRY(hipMemcpyAsync(st.grid[which] + (size_t)i_begin * plane, host, (size_t)i_count * plane * sizeof(double),
                                 hipMemcpyHostToDevice, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    st.grid_valid[which] = true;             // (the caller vouches for the planes it did not write)
    if (which == ASORA_GRID_TEMP) st.temp_probe_valid = false;
    return 0;
}

int asora_set_option(int option, int value)
{
    clear_error();
    if (option < 0 || option >= ASORA_OPT_COUNT) return fail(3, "set_option: unknown option");
    g_state.opt[option] = value;
    return 0;
}

int asora_get_option(int option)
{
    if (option < 0 || option >= ASORA_OPT_COUNT) return -1;
    return g_state.opt[option];
}

int asora_kernel_time_ms(int kernel, double *total_ms, long *launches)
{
    clear_error();
    if (kernel < 0 || kernel >= ASORA_KERNEL_COUNT) return fail(3, "kernel_time_ms: unknown kernel");
    if (int rc = flush_timers()) return rc;
    if (total_ms) *total_ms = g_state.k_ms[kernel];
    if (launches) *launches = g_state.k_n[kernel];
    return 0;
}

int asora_kernel_time_reset(void)
{
    (void)flush_timers();
    for (int k = 0; k < ASORA_KERNEL_COUNT; ++k) { g_state.k_ms[k] = 0.0; g_state.k_n[k] = 0; }
    return 0;
}

int asora_synchronize(void)
{
    clear_error();
    if (!g_state.stream) return 0;
    ASORA_HIP_TRY(hipStreamSynchronize(g_state.stream));
    ASORA_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int asora_last_raytrace_counts(long long *gamma_cells, long long *evaluated_cells)
{
    clear_error();
    if (int rc = require_init("last_raytrace_counts")) return rc;
    long long zero = 0;
    return asora_last_raytrace_counts_ex(gamma_cells, evaluated_cells, &zero);
}

int asora_last_raytrace_counts_ex(long long *gamma_cells, long long *evaluated_cells, long long *zero_rates_left_out)
{
    clear_error();
    if (int rc = require_init("last_raytrace_counts")) return rc;
    std::vector<unsigned long long> h((size_t)COUNTER_FIELDS * COUNTER_SLOTS, 0ULL);
    ASORA_HIP_TRY(hipStreamSynchronize(g_state.stream));
    ASORA_HIP_TRY(hipMemcpy(h.data(), g_state.counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tot[COUNTER_FIELDS] = {0ULL, 0ULL, 0ULL};
    for (int q = 0; q < COUNTER_SLOTS; ++q) for (int f = 0; f < COUNTER_FIELDS; ++f) tot[f] += h[(size_t)COUNTER_FIELDS * q + f];
    if (gamma_cells) *gamma_cells = (long long)tot[0];
    if (evaluated_cells) *evaluated_cells = (long long)tot[1];
    if (zero_rates_left_out) *zero_rates_left_out = (long long)tot[2];
    return 0;
}

int asora_last_raytrace_variant(void) { return g_state.last_variant; }

int asora_debug_coldens(double R, double sig, double dr, int source_index, double *coldens_out, int N)
{
    clear_error();
    if (int rc = require_init("debug_coldens")) return rc;
    if (int rc = check_N("debug_coldens", N)) return rc;
    if (!coldens_out) return fail(3, "debug_coldens: null output");
    State &st = g_state;
    const size_t bytes = st.ncell * sizeof(double);
    ASORA_HIP_TRY(hipMemsetAsync(st.staging, 0, bytes, st.stream));
    // the column density does not depend on the tables: trace with whatever is loaded
    const int numtau = st.table_len > 0 ? st.table_len : 1;
    const int grey_save = st.opt[ASORA_OPT_GREY_NOTABLES];
    if (!st.tables) st.opt[ASORA_OPT_GREY_NOTABLES] = 1;
    int rc = do_raytrace(R, sig, dr, source_index, 1, -20.0, 1.0, numtau, st.staging);
    st.opt[ASORA_OPT_GREY_NOTABLES] = grey_save;
    if (rc) return rc;
    ASORA_HIP_TRY(hipMemcpyAsync(coldens_out, st.staging, bytes, hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    return 0;
}

} // extern "C"
