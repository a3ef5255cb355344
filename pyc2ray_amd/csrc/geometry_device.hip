// geometry_device.hip -- the raytrace kernel's geometry tables built ON THE DEVICE (round 5).
//
// The tables (format: raytrace.hip; semantics: build_unit_geometry in geometry.hip, which stays as the host-side statement of
// what a table holds and as the checker of this file -- tests compare the two bit for bit) are a function of (N, R, dr, launch
// shape) only.  Building them on the host cost 0.2-0.65 s for a whole-box trace (0.5-1.3 GB through PCIe) and 2.5-12 ms at
// r_RT = 30 -- per change of the radius, i.e. per time step of a cosmological run.  The reference derives its geometry inside the
// kernel (src/asora/raytracing.cu:39-59,228-238) and has no such cost; here the derivation stays outside the sweep but moves to
// the GPU: two passes over the CANDIDATE cells of every shell of every distinct table, all (table, shell) pairs side by side.
//
//   candidates of shell s : the cells the host builder's loops visit, in its order -- per face (z, y, x), per sign of the
//                           dominant offset, rows of the memory-contiguous transverse axis (shell_blocks below);
//   pass A (count)        : kept? (inside the sphere / octahedron / window; the sector's own face or a plane it reads) and, by an
//                           ordered scan, the RANK among the kept cells of the shell = the cell's slot in the kernel's shell
//                           buffer; for line-aligned tables also the position after the host builder's repacking (a sequential
//                           greedy packing of row pieces into waves, done by one lane per shell);
//   host                  : reads the per-shell counts (a few KB), lays the shells out (steps, padding, sub-box triples), allocates;
//   pass B (write)        : the entries -- packed offsets, flags, path, and the slots of the four upstream corners, looked up
//                           in the rank array of shell s-1 -- at their final positions.
// Quarter-sector tables (a handful of sources) are cut out of full sector tables as restrict_to_wedge does: marking backwards
// shell by shell what a wedge's rated cells read, renumbering, rewriting.
#include "asora_internal.hpp"
#include "rates_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace asora {

namespace {

constexpr uint32_t MARK = 0xffffffffu;
constexpr int GB_THREADS = 512;

struct BuildParams {
    double R2, R2hi, R2lo, dr;
    int q_max, threads, boxsize, S;
};

// The candidate cells of shell s of a table, as up to six blocks (face, sign of the dominant offset) of rows: exactly the loops
// of build_unit_geometry (geometry.hip): faces z, y, x; the dominant offset positive, then (merged axis only) negative; rows of
// the slow transverse axis, each along the fast (memory-contiguous) one.
struct ShellBlocks {
    int nblk;
    int face[6], dsgn[6], lo_slow[6], n_slow[6], lo_fast[6], n_fast[6];
    int start[7];
};

__host__ __device__ inline void shell_blocks(const GeomTableSpec &us, int s, ShellBlocks &B)
{
    B.nblk = 0;
    B.start[0] = 0;
    for (int face = 2; face >= 0; --face) {
        if (us.face == 2 && face != 2) continue;          // the z-sector holds z-face cells only
        if (us.face == 1 && face == 0) continue;          // the y-sector never needs x-face cells
        const int d = face;
        const int fast = face == 2 ? 0 : 2;
        const int slow = face == 2 ? 1 : (face == 1 ? 0 : 1);
        const int max_fast = (face == 2) ? s : s - 1;
        const int max_slow = (face == 0) ? s - 1 : s;
        for (int dsgn = 1; dsgn >= -1; dsgn -= 2) {
            if (dsgn < 0 && !((us.merge_mask >> d) & 1)) break;
            const int ext_d = dsgn > 0 ? us.ext[d] : us.ext_neg;
            if (s > ext_d) continue;
            const int lo_s = ((us.merge_mask >> slow) & 1) ? -(max_slow < us.ext_neg ? max_slow : us.ext_neg) : 0;
            const int hi_s = max_slow < us.ext[slow] ? max_slow : us.ext[slow];
            const int lo_f = ((us.merge_mask >> fast) & 1) ? -(max_fast < us.ext_neg ? max_fast : us.ext_neg) : 0;
            const int hi_f = max_fast < us.ext[fast] ? max_fast : us.ext[fast];
            const int b = B.nblk++;
            B.face[b] = face; B.dsgn[b] = dsgn;
            B.lo_slow[b] = lo_s; B.n_slow[b] = hi_s >= lo_s ? hi_s - lo_s + 1 : 0;
            B.lo_fast[b] = lo_f; B.n_fast[b] = hi_f >= lo_f ? hi_f - lo_f + 1 : 0;
            B.start[b + 1] = B.start[b] + B.n_slow[b] * B.n_fast[b];
        }
    }
}

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// candidate -> signed offsets and face
__device__ __forceinline__ void cand_cell(const ShellBlocks &B, int cand, int s, int x[3], int &face)
{
    int b = 0;
    while (b + 1 < B.nblk && cand >= B.start[b + 1]) ++b;
    const int r = cand - B.start[b];
    const int row = r / B.n_fast[b];
    face = B.face[b];
    const int fast = face == 2 ? 0 : 2, slow = face == 2 ? 1 : (face == 1 ? 0 : 1);
    x[face] = B.dsgn[b] * s;
    x[slow] = B.lo_slow[b] + row;
    x[fast] = B.lo_fast[b] + (r - row * B.n_fast[b]);
}

// 0: outside, 1: inside, 2: ON the sphere to rounding (geometry.hip, in_sphere)
__device__ __forceinline__ int in_sphere(const BuildParams &P, int a, int b, int c)
{
    if (a + b + c > P.q_max) return 0;
    const double n2 = (double)a * a + (double)b * b + (double)c * c;
    if (n2 > P.R2hi) return 0;
    if (n2 < P.R2lo) return 1;
    return 2;
}

// is the candidate part of the table?  (where = in_sphere; foreign = a cell of another face that this sector reads)
__device__ __forceinline__ bool cand_kept(const GeomTableSpec &us, const BuildParams &P, const int x[3], int face, int &where, bool &foreign)
{
    const int a = iabs(x[0]), b = iabs(x[1]), c = iabs(x[2]);
    where = in_sphere(P, a, b, c);
    foreign = false;
    if (!where) return false;
    if (us.face >= 0 && face != us.face) {
        const bool keep = (us.face == 1) ? (face == 2 && b == c)
                                         : (us.face == 0) ? ((face == 2 && a == c) || (face == 1 && a == b)) : false;
        if (!keep) return false;
        foreign = true;
    }
    return true;
}

// the candidate index, in shell `sp`, of the cell with offsets n (|n| has max-norm sp >= 1), or -1 when the shell's loops do not
// visit it
__device__ __forceinline__ int cand_of_cell(const ShellBlocks &Bp, const int n[3], int sp)
{
    const int bb = iabs(n[1]), cc = iabs(n[2]);
    const int face = (cc == sp) ? 2 : (bb == sp) ? 1 : 0;          // ties z, then y (raytracing.cu:394,446,491)
    const int fast = face == 2 ? 0 : 2, slow = face == 2 ? 1 : (face == 1 ? 0 : 1);
    const int dsgn = n[face] < 0 ? -1 : 1;
    for (int b = 0; b < Bp.nblk; ++b) {
        if (Bp.face[b] != face || Bp.dsgn[b] != dsgn) continue;
        const int rs = n[slow] - Bp.lo_slow[b], rf = n[fast] - Bp.lo_fast[b];
        if (rs < 0 || rs >= Bp.n_slow[b] || rf < 0 || rf >= Bp.n_fast[b]) return -1;
        return Bp.start[b] + rs * Bp.n_fast[b] + rf;
    }
    return -1;
}

// inside_radius_reference (geometry.hip): every operation rounded on its own
__device__ __forceinline__ bool inside_radius_device(int a, int b, int c, double dr, double R2)
{
    const double xs = mul_unfused(dr, (double)a), ys = mul_unfused(dr, (double)b), zs = mul_unfused(dr, (double)c);
    const double d2 = add_unfused(add_unfused(mul_unfused(xs, xs), mul_unfused(ys, ys)), mul_unfused(zs, zs));
    const double den = mul_unfused(dr, dr);
    return d2 / den <= R2;
}

// ordered exclusive prefix of a flag over the workgroup; `total` = number of set flags.  (two barriers)
__device__ __forceinline__ int block_rank(bool flag, int *wave_tot, int &total)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    __syncthreads();                                  // the previous round's totals have been read
    if (lane == 0) wave_tot[wave] = __builtin_popcountll(m);
    __syncthreads();
    int base = 0;
    total = 0;
    for (int w = 0; w < GB_THREADS / 64; ++w) {
        const int t = wave_tot[w];
        if (w < wave) base += t;
        total += t;
    }
    return base + before;
}

struct TableJob {
    GeomTableSpec spec;
    size_t cand_base;            // start of the table's candidates in the rank / pos arrays
    size_t meta_base;            // start of its per-shell records (S + 1 each)
    int S_built;                 // pass B: shells first_shell..S_built are written
    int first_shell;             // 1, or m + 1 for a table whose shells 1..m live in memory it shares with its family (share_prefixes)
    uint4 *cellA, *cellB;        // pass B
};

// ---- pass A ----------------------------------------------------------------------------------------------------------
// per (table, shell): rank of every kept candidate, the shell's cell count and its entry count after the aligned repacking
__global__ void __launch_bounds__(GB_THREADS) geometry_count_kernel(const TableJob *jobs, const size_t *shell_cand, BuildParams P,
                                                                     uint32_t *rank, uint32_t *pos, int *count, int *packed)
{
    __shared__ ShellBlocks B;
    __shared__ int wave_tot[GB_THREADS / 64];
    const TableJob &J = jobs[blockIdx.y];
    const int s = blockIdx.x + 1;
    const GeomTableSpec us = J.spec;
    if (threadIdx.x == 0) shell_blocks(us, s, B);
    __syncthreads();
    const int ncand = B.start[B.nblk];
    const size_t base = J.cand_base + shell_cand[J.meta_base + s];
    int running = 0;
    for (int c0 = 0; c0 < ncand; c0 += GB_THREADS) {
        const int cand = c0 + threadIdx.x;
        bool k = false;
        if (cand < ncand) {
            int x[3], face, where; bool foreign;
            cand_cell(B, cand, s, x, face);
            k = cand_kept(us, P, x, face, where, foreign);
        }
        int total;
        const int r = block_rank(k, wave_tot, total);
        if (cand < ncand) rank[base + cand] = k ? (uint32_t)(running + r) : MARK;
        running += total;
    }
    if (threadIdx.x == 0) count[J.meta_base + s] = running;
    if (us.align_class < 0 || us.face < 0) { if (threadIdx.x == 0) packed[J.meta_base + s] = running; return; }
    // ---- line-aligned tables: the host builder's repacking of the shell, by one lane (the packing is a sequential greedy) ----
    __syncthreads();                                   // the ranks of the shell are visible
    if (threadIdx.x != 0) return;
    const int fast = us.face == 2 ? 0 : 2;
    int size = 0;                                      // entries of the repacked shell so far
    int run_first = -1, run_n = 0, run_line = 0, run_prev_fast = 0;
    bool run_rated = false;
    int run_key[3] = {0, 0, 0};                        // |slow|, sign bits of the other axes, block: what "the same row" means
    auto flush = [&]() {
        if (run_n == 0) return;
        const int p0 = size & 63;
        if (p0 + run_n > 64) size = (size + 63) & ~63;
        for (int i = 0; i < run_n; ++i) pos[base + run_first + i] = (uint32_t)(size + i);
        size += run_n;
        run_n = 0;
    };
    for (int cand = 0; cand < ncand; ++cand) {
        if (rank[base + cand] == MARK) continue;
        int x[3], face, where; bool foreign;
        cand_cell(B, cand, s, x, face);
        (void)cand_kept(us, P, x, face, where, foreign);
        const bool rated = !foreign && face == us.face;               // (RATE | SPHERE) of an own-face cell
        const int f = x[fast];
        const int line = (us.align_class + f + 8192) >> 3;
        // the same row: same magnitudes of the other two offsets and the same face (the entry's x word without the fast
        // field), same signs of the other two axes (their NEG bits)
        int key[3];
        {
            const int o1 = fast == 0 ? 1 : 0, o2 = fast == 2 ? 1 : 2;
            key[0] = iabs(x[o1]) | (iabs(x[o2]) << 10) | (face << 20);
            key[1] = ((((us.merge_mask >> o1) & 1) && x[o1] < 0) ? 1 : 0) | ((((us.merge_mask >> o2) & 1) && x[o2] < 0) ? 2 : 0);
            key[2] = 0;
        }
        const bool cont = run_n > 0 && run_rated && rated && key[0] == run_key[0] && key[1] == run_key[1] &&
                          f == run_prev_fast + 1 && line == run_line;
        if (!cont) {
            flush();
            run_first = cand; run_rated = rated; run_line = line;
            run_key[0] = key[0]; run_key[1] = key[1];
        }
        run_n += 1;
        run_prev_fast = f;
    }
    flush();
    packed[J.meta_base + s] = size;
}

// ---- pass B ----------------------------------------------------------------------------------------------------------
struct SphereRecord { unsigned long long entry; uint32_t word_without_rate; int a, b, c; uint32_t table; };

__global__ void __launch_bounds__(GB_THREADS) geometry_write_kernel(const TableJob *jobs, const size_t *shell_cand, const size_t *shell_entry,
                                                                     const int *shell_len, BuildParams P, const uint32_t *rank,
                                                                     const uint32_t *pos, uint32_t max_cells, int *flags,
                                                                     SphereRecord *sphere, unsigned *n_sphere, unsigned sphere_cap)
{
    __shared__ ShellBlocks B, Bp;
    const TableJob &J = jobs[blockIdx.y];
    const int s = blockIdx.x + 1;
    if (s > J.S_built || s < J.first_shell) return;
    const GeomTableSpec us = J.spec;
    if (threadIdx.x == 0) { shell_blocks(us, s, B); if (s > 1) shell_blocks(us, s - 1, Bp); }
    __syncthreads();
    const int ncand = B.start[B.nblk];
    const size_t base = J.cand_base + shell_cand[J.meta_base + s];
    const size_t base_prev = J.cand_base + shell_cand[J.meta_base + s - 1];
    const size_t e0 = shell_entry[J.meta_base + s];
    const bool aligned = us.align_class >= 0 && us.face >= 0;
    const double sd = (double)s;
    const int len = shell_len[J.meta_base + s];         // entries of the shell's steps proper (before any sub-box padding)
    for (int cand = threadIdx.x; cand < ncand; cand += GB_THREADS) {
        const uint32_t slot = rank[base + cand];
        if (slot == MARK) continue;
        int x[3], face, where; bool foreign;
        cand_cell(B, cand, s, x, face);
        (void)cand_kept(us, P, x, face, where, foreign);
        const int a = iabs(x[0]), b = iabs(x[1]), c = iabs(x[2]);
        bool rate = where == 1 || inside_radius_device(a, b, c, P.dr, P.R2);
        bool sph = where == 2;
        if (foreign) { rate = false; sph = false; }
        const int d = face, e = face == 0 ? 1 : 0, f = face == 2 ? 1 : 2;           // transverse axes (e, f): x:(y,z) y:(x,z) z:(x,y)
        const int U = iabs(x[e]), V = iabs(x[f]);
        const int sgd = x[d] < 0 ? -1 : 1, sge = x[e] < 0 ? -1 : 1, sgf = x[f] < 0 ? -1 : 1;
        const bool em = U >= 1, e0c = U <= s - 1, fm = V >= 1, f0c = V <= s - 1;
        auto corner = [&](int uu, int vv) -> uint32_t {
            if (s == 1) return 0u;                          // the source cell: slot 0 of shell 0
            int n[3];
            n[d] = sgd * (s - 1); n[e] = sge * uu; n[f] = sgf * vv;
            const int cp = cand_of_cell(Bp, n, s - 1);
            return cp < 0 ? MARK : rank[base_prev + cp];
        };
        uint4 nb;
        nb.x = (em && fm) ? corner(U - 1, V - 1) : MARK;
        nb.y = (e0c && fm) ? corner(U, V - 1) : MARK;
        nb.z = (em && f0c) ? corner(U - 1, V) : MARK;
        nb.w = (e0c && f0c) ? corner(U, V) : MARK;
        // (a transverse offset equal to s: the corners at that offset alias their existing neighbour -- see build_unit_geometry)
        if (!e0c) { nb.y = nb.x; nb.w = nb.z; }
        if (!f0c) { nb.z = nb.x; nb.w = nb.y; }
        const double u = (double)U, v = (double)V;
        {   // every corner that carries weight must be part of this table
            const double fu = U == s ? 1.0 : u / sd, fv = V == s ? 1.0 : v / sd;
            const double wts[4] = {fu * fv, fv * (1.0 - fu), fu * (1.0 - fv), (1.0 - fu) * (1.0 - fv)};
            const uint32_t sl[4] = {nb.x, nb.y, nb.z, nb.w};
            for (int q = 0; q < 4; ++q) if (wts[q] != 0.0 && sl[q] == MARK) atomicOr(flags, 1);
        }
        // raytracing.cu:444; u*u + v*v and sd*sd are exact in double, so the contraction the device compiler may apply changes nothing
        const double path = sqrt((u * u + v * v) / (sd * sd) + 1.0);
        const unsigned long long pbits = (unsigned long long)__double_as_longlong(path);
        uint32_t negbits = 0;
        for (int ax = 0; ax < 3; ++ax) if (((us.merge_mask >> ax) & 1) && x[ax] < 0) negbits |= 1u << ax;
        uint4 ca;
        ca.x = (uint32_t)a | ((uint32_t)b << 10) | ((uint32_t)c << 20) | ((uint32_t)face << 30);
        ca.y = slot | CELL_VALID | (rate ? CELL_RATE : 0u) | (sph ? CELL_SPHERE : 0u) | (negbits << CELL_NEG_SHIFT) |
               (((a == 0 ? 1u : 0u) | (b == 0 ? 2u : 0u) | (c == 0 ? 4u : 0u)) << CELL_ZERO_SHIFT);
        ca.z = (uint32_t)(pbits & 0xffffffffull);
        ca.w = (uint32_t)(pbits >> 32);
        if (nb.x == MARK) nb.x = max_cells;
        if (nb.y == MARK) nb.y = max_cells;
        if (nb.z == MARK) nb.z = max_cells;
        if (nb.w == MARK) nb.w = max_cells;
        const uint32_t in_shell = aligned ? pos[base + cand] : slot;
        const size_t at = e0 + in_shell;
        J.cellA[at] = ca;
        J.cellB[at] = nb;
        if (s == 1 && in_shell >= (uint32_t)(3 * P.threads)) atomicOr(flags, 2);   // shell 1 within the first three steps
        if (sph) {
            // (the word the patch writes: as the entry will stand once the shell's last step has been flagged, below)
            const uint32_t last = (int)in_shell >= len - P.threads ? CELL_LAST : 0u;
            const unsigned q = atomicAdd(n_sphere, 1u);
            if (q < sphere_cap) sphere[q] = SphereRecord{(unsigned long long)at, (ca.y & ~CELL_RATE) | last, a, b, c, blockIdx.y};
        }
    }
    // every entry of the shell's last step carries CELL_LAST (padding included)
    __syncthreads();
    for (int q = threadIdx.x; q < P.threads; q += GB_THREADS) J.cellA[e0 + (size_t)(len - P.threads + q)].y |= CELL_LAST;
}

__global__ void fill_pad_kernel(uint4 *b, size_t n, uint32_t v)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) b[q] = uint4{v, v, v, v};
}

// ---- quarter sectors (restrict_to_wedge, geometry.hip) ---------------------------------------------------------------
// keep[] has one byte per ENTRY of the full sector table (entry = its position in the table; shells are dense, slot == rank).
__device__ __forceinline__ bool entry_in_wedge(const uint4 &a, int wedge)
{
    if (!(a.y & (CELL_RATE | CELL_SPHERE))) return false;
    const int ca = a.x & 1023, cb = (a.x >> 10) & 1023, cc = (a.x >> 20) & 1023, face = a.x >> 30;
    const int s = max(ca, max(cb, cc));
    const int U = face == 0 ? cb : ca, V = face == 2 ? cb : cc;
    return ((2 * U > s ? 1 : 0) | (2 * V > s ? 2 : 0)) == wedge;
}

struct WedgeJob {
    const uint4 *fullA, *fullB;      // the sector's full table
    size_t meta_base;                // its per-shell records (entry offsets, counts)
    int S_full;                      // shells 1..S_full hold cells
    int wedge;
    size_t keep_base;                // this wedge's keep bytes (one per entry of the full table)
    size_t wmeta_base;               // the wedge table's per-shell records
    int S_built;
    int first_shell;                 // as TableJob::first_shell
    uint4 *cellA, *cellB;
};

// keep[dst] |= keep[src] over n bytes: the inner shells of a family of wedge tables are kept as the UNION of what its members need
// (a union of dependency-closed sets is closed), so that those shells come out identical and can share their memory
__global__ void keep_or_kernel(unsigned char *keep, size_t dst, size_t src, size_t n)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) keep[dst + q] |= keep[src + q];
}

__global__ void wedge_seed_kernel(const WedgeJob *jobs, const size_t *shell_entry, const int *count, unsigned char *keep)
{
    const WedgeJob &J = jobs[blockIdx.y];
    const int s = blockIdx.x + 1;
    if (s > J.S_full) return;
    const size_t e0 = shell_entry[J.meta_base + s];
    const int n = count[J.meta_base + s];
    for (int q = threadIdx.x; q < n; q += blockDim.x) keep[J.keep_base + e0 + q] = entry_in_wedge(J.fullA[e0 + q], J.wedge) ? 1 : 0;
}

// shell s -> shell s-1: what a kept cell reads is kept
__global__ void wedge_mark_kernel(const WedgeJob *jobs, const size_t *shell_entry, const int *count, unsigned char *keep, int s, uint32_t zero_slot)
{
    const WedgeJob &J = jobs[blockIdx.y];
    if (s > J.S_full || s < 2) return;
    const size_t e0 = shell_entry[J.meta_base + s], ep = shell_entry[J.meta_base + s - 1];
    const int n = count[J.meta_base + s];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        if (!keep[J.keep_base + e0 + q]) continue;
        const uint4 b = J.fullB[e0 + q];
        if (b.x != zero_slot) keep[J.keep_base + ep + b.x] = 1;
        if (b.y != zero_slot) keep[J.keep_base + ep + b.y] = 1;
        if (b.z != zero_slot) keep[J.keep_base + ep + b.z] = 1;
        if (b.w != zero_slot) keep[J.keep_base + ep + b.w] = 1;
    }
}

// new slot of every kept entry of a shell (ordered), and the shell's count
__global__ void __launch_bounds__(GB_THREADS) wedge_count_kernel(const WedgeJob *jobs, const size_t *shell_entry, const int *count,
                                                                  const unsigned char *keep, uint32_t *newslot, int *wcount)
{
    __shared__ int wave_tot[GB_THREADS / 64];
    const WedgeJob &J = jobs[blockIdx.y];
    const int s = blockIdx.x + 1;
    if (s > J.S_full) { if (threadIdx.x == 0) wcount[J.wmeta_base + s] = 0; return; }
    const size_t e0 = shell_entry[J.meta_base + s];
    const int n = count[J.meta_base + s];
    int running = 0;
    for (int c0 = 0; c0 < n; c0 += GB_THREADS) {
        const int q = c0 + threadIdx.x;
        const bool k = q < n && keep[J.keep_base + e0 + q] != 0;
        int total;
        const int r = block_rank(k, wave_tot, total);
        if (q < n) newslot[J.keep_base + e0 + q] = k ? (uint32_t)(running + r) : MARK;
        running += total;
    }
    if (threadIdx.x == 0) wcount[J.wmeta_base + s] = running;
}

__global__ void __launch_bounds__(GB_THREADS) wedge_write_kernel(const WedgeJob *jobs, const size_t *shell_entry, const int *count,
                                                                  const size_t *wshell_entry, const int *wshell_len, const uint32_t *newslot,
                                                                  uint32_t zero_slot_full, uint32_t max_cells, int threads, int *flags,
                                                                  SphereRecord *sphere, unsigned *n_sphere, unsigned sphere_cap, unsigned table0)
{
    const WedgeJob &J = jobs[blockIdx.y];
    const int s = blockIdx.x + 1;
    if (s > J.S_built || s < J.first_shell) return;
    const size_t e0 = shell_entry[J.meta_base + s], ep = shell_entry[J.meta_base + s - 1];
    const size_t w0 = wshell_entry[J.wmeta_base + s];
    const int n = count[J.meta_base + s];
    const int len = wshell_len[J.wmeta_base + s];
    for (int q = threadIdx.x; q < n; q += GB_THREADS) {
        const uint32_t ns = newslot[J.keep_base + e0 + q];
        if (ns == MARK) continue;
        uint4 a = J.fullA[e0 + q], b = J.fullB[e0 + q];
        const uint32_t fl = a.y & ((7u << CELL_NEG_SHIFT) | (7u << CELL_ZERO_SHIFT));
        const bool mine = entry_in_wedge(a, J.wedge);
        a.y = ns | CELL_VALID | fl | ((mine && (a.y & CELL_RATE)) ? CELL_RATE : 0u) | ((mine && (a.y & CELL_SPHERE)) ? CELL_SPHERE : 0u);
        auto remap = [&](uint32_t slot) -> uint32_t {
            if (slot == zero_slot_full) return max_cells;
            if (s == 1) return slot;                     // corners of shell 1 point into shell 0 (the source cell, slot 0)
            const uint32_t m = newslot[J.keep_base + ep + slot];
            if (m == MARK) { atomicOr(flags, 1); return max_cells; }
            return m;
        };
        b.x = remap(b.x); b.y = remap(b.y); b.z = remap(b.z); b.w = remap(b.w);
        J.cellA[w0 + ns] = a;
        J.cellB[w0 + ns] = b;
        if (s == 1 && ns >= (uint32_t)(3 * threads)) atomicOr(flags, 2);
        if (a.y & CELL_SPHERE) {
            const uint32_t last = (int)ns >= len - threads ? CELL_LAST : 0u;
            const unsigned k = atomicAdd(n_sphere, 1u);
            if (k < sphere_cap) sphere[k] = SphereRecord{(unsigned long long)(w0 + ns), (a.y & ~CELL_RATE) | last, (int)(a.x & 1023), (int)((a.x >> 10) & 1023),
                                                         (int)((a.x >> 20) & 1023), table0 + blockIdx.y};
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < threads; q += GB_THREADS) J.cellA[w0 + (size_t)(len - threads + q)].y |= CELL_LAST;
}

template <typename T>
struct DevArray {
    T *p = nullptr;
    ~DevArray() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { ASORA_HIP_TRY(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T))); return 0; }
    int upload(const std::vector<T> &v, hipStream_t s)
    {
        if (int rc = alloc(v.size())) return rc;
        if (!v.empty()) ASORA_HIP_TRY(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
        return 0;
    }
};

// lay the shells of a table out: entry offset of every shell, entries of its steps proper, total steps; the host builder's
// padding rules (whole steps per shell; whole triples behind a sub-box boundary; whole triples at the end; four closing steps)
struct Layout { std::vector<size_t> entry, after; std::vector<int> len; std::vector<int> step_after_shell; int S_built = 0, nsteps = 0; uint32_t max_cells = 1; size_t entries = 0; };   // after[s]: entries up to and including shell s (its steps and any sub-box padding)
Layout lay_out(const int *count, const int *packed, int S, int threads, int boxsize)
{
    Layout L;
    L.entry.assign((size_t)S + 2, 0);
    L.after.assign((size_t)S + 2, 0);
    L.len.assign((size_t)S + 2, 0);
    size_t at = 0;
    for (int s = 1; s <= S; ++s) {
        if (count[s] == 0) break;                         // nothing further out either
        L.S_built = s;
        L.entry[(size_t)s] = at;
        const size_t len = ((size_t)packed[s] + threads - 1) / threads * threads;
        L.len[(size_t)s] = (int)len;
        at += len;
        if (boxsize > 0 && s % boxsize == 0) while ((at / (size_t)threads) % 3) at += (size_t)threads;
        L.after[(size_t)s] = at;
        L.max_cells = std::max<uint32_t>(L.max_cells, (uint32_t)count[s]);
        L.step_after_shell.resize((size_t)s + 1, 0);
        L.step_after_shell[(size_t)s] = (int)(at / (size_t)threads);
    }
    while ((at / (size_t)threads) % 3) at += (size_t)threads;
    L.nsteps = (int)(at / (size_t)threads);
    if (L.step_after_shell.empty()) L.step_after_shell.push_back(0);
    L.step_after_shell.back() = L.nsteps;
    L.entries = at + 4 * (size_t)threads;
    return L;
}

// ---- tables that share their inner shells ---------------------------------------------------------------------------
// Tables of one family -- same face, merged axes, wedge, alignment class; window extents that differ -- hold the same entries
// for every shell up to m = the smallest extent in the family: the window does not bind there (shell_blocks: min(maxmag <= s, ext)).
// A trace beyond the box on an even mesh is the case that matters: the window is [-N/2, N/2 - 1], every sign variant of a unit
// gets a table of its own, and they differ in the LAST shell only (0.56 GB of tables for 12 units at 256^3, 1.3 GB for 96).
// The family's first member holds the whole table; every other member holds only what follows shell m and tells the kernel to
// read the steps before through the first member (OctGeomDev::inner -- a wave-uniform choice between two base pointers per
// step).  A member's own pointers are shifted back by the shared entries, so that entry e of a table is cellA[e] whichever side
// it lies on and nothing below (write kernels, on-sphere patches) needs to know.
// (Mapping one physical allocation into every member's address range -- hipMemCreate / hipMemMap -- needs no kernel change, and was
//  how this was first done; ranges reserved after earlier ones had been unmapped and freed read back OTHER allocations' contents on
//  this software stack: tools/micro/vmm_many_small.hip, profiles/r05_vmm_many_small.txt.)
struct Family { std::vector<int> members; int m = 0; size_t prefix_entries = 0; };

static std::vector<Family> find_families(const std::vector<GeomTableSpec> &specs, const std::vector<Layout> &L, int threads)
{
    std::vector<Family> fams;
    static const bool no_sharing = getenv("ASORA_GEOMETRY_NO_SHARING") != nullptr;        // (A/B of the memory figure; tests)
    if (no_sharing) return fams;
    std::vector<char> taken(specs.size(), 0);
    auto min_ext = [](const GeomTableSpec &g) { int m = std::min(g.ext[0], std::min(g.ext[1], g.ext[2])); return g.merge_mask ? std::min(m, g.ext_neg) : m; };
    for (size_t t = 0; t < specs.size(); ++t) {
        if (taken[t]) continue;
        Family f;
        for (size_t u = t; u < specs.size(); ++u)
            if (!taken[u] && specs[u].face == specs[t].face && specs[u].merge_mask == specs[t].merge_mask && specs[u].wedge == specs[t].wedge &&
                specs[u].align_class == specs[t].align_class) { f.members.push_back((int)u); taken[u] = 1; }
        if (f.members.size() < 2) continue;
        f.m = min_ext(specs[(size_t)f.members[0]]);
        for (int u : f.members) f.m = std::min(f.m, min_ext(specs[(size_t)u]));
        // shared: the entries before shell m + 1 -- the same offset in every member's layout, whole steps
        bool ok = f.m >= 1;
        for (int u : f.members) ok = ok && L[(size_t)u].S_built >= f.m;
        if (!ok) continue;
        auto prefix_of = [&](int u) { return L[(size_t)u].after[(size_t)f.m]; };
        f.prefix_entries = prefix_of(f.members[0]);
        for (int u : f.members) ok = ok && prefix_of(u) == f.prefix_entries;
        if (!ok || f.prefix_entries == 0 || f.prefix_entries % (size_t)threads != 0 || f.prefix_entries / (size_t)threads >= (1u << 23)) continue;
        fams.push_back(f);
    }
    return fams;
}

// Allocate cellA / cellB of every table.  first_shell[t] = first shell table t writes itself, fill_from[t] = its first own entry,
// inner[t] = OctGeomDev::inner with the index of the family's first member among `specs`.
static int allocate_tables(State &st, const std::vector<GeomTableSpec> &specs, const std::vector<Layout> &L, int threads,
                           std::vector<uint4 *> &A, std::vector<uint4 *> &B, std::vector<int> &first_shell, std::vector<size_t> &fill_from,
                           std::vector<int> &inner)
{
    const size_t nt = specs.size();
    A.assign(nt, nullptr); B.assign(nt, nullptr); first_shell.assign(nt, 1); fill_from.assign(nt, 0); inner.assign(nt, 0);
    if (nt <= 256)
        for (const Family &f : find_families(specs, L, threads))
            for (size_t q = 1; q < f.members.size(); ++q) {
                const size_t u = (size_t)f.members[q];
                first_shell[u] = f.m + 1;
                fill_from[u] = f.prefix_entries;
                inner[u] = (int)(f.prefix_entries / (size_t)threads) << 8 | f.members[0];
            }
    for (size_t t = 0; t < nt; ++t) {
        const size_t own = L[t].entries - fill_from[t];
        uint4 *a = nullptr, *b = nullptr;
        ASORA_HIP_TRY(hipMalloc(&a, own * sizeof(uint4))); st.geom_owned.push_back(a);
        ASORA_HIP_TRY(hipMalloc(&b, own * sizeof(uint4))); st.geom_owned.push_back(b);
        st.geom_bytes += 2 * own * sizeof(uint4);
        // (entry e >= fill_from of the table is own[e - fill_from]; the shifted pointer is never dereferenced below fill_from)
        A[t] = reinterpret_cast<uint4 *>(reinterpret_cast<uintptr_t>(a) - fill_from[t] * sizeof(uint4));
        B[t] = reinterpret_cast<uint4 *>(reinterpret_cast<uintptr_t>(b) - fill_from[t] * sizeof(uint4));
    }
    return 0;
}

} // namespace

// Build the tables of `specs` (the DISTINCT ones of a launch shape; quarter sectors: specs[].wedge >= 0, each cut out of the full
// sector of the same face / extents) on the device.  out[t]: device pointers and step count; step_after[t]: steps up to and
// including every shell.  The tables are owned by st.geom_owned; st.geom_sphere gets the on-sphere entries.
int build_geometry_on_device(State &st, const std::vector<GeomTableSpec> &specs, double R, double dr, int q_max, int threads,
                             int boxsize, std::vector<OctGeomDev> &out, std::vector<std::vector<int>> &step_after, int &S_all,
                             uint32_t &max_cells_all)
{
    const int nt = (int)specs.size();
    out.assign((size_t)nt, OctGeomDev{nullptr, nullptr, 0, 0, 0, 0});
    step_after.assign((size_t)nt, std::vector<int>());
    BuildParams P;
    P.R2 = R * R; P.R2hi = P.R2 * (1.0 + 1e-9) + 1e-9; P.R2lo = P.R2 * (1.0 - 1e-9) - 1e-9; P.dr = dr;
    P.q_max = q_max; P.threads = threads; P.boxsize = boxsize;
    const bool wedges = nt > 0 && specs[0].wedge >= 0;
    // the full tables to build: the specs themselves, or -- quarter sectors -- the distinct whole sectors they are cut from
    std::vector<GeomTableSpec> full;
    std::vector<int> full_of((size_t)nt, -1);
    for (int t = 0; t < nt; ++t) {
        GeomTableSpec w = specs[(size_t)t];
        if (wedges) { w.wedge = -1; w.align_class = -1; }
        int found = -1;
        if (wedges)
            for (size_t q = 0; q < full.size(); ++q)
                if (full[q].face == w.face && full[q].merge_mask == w.merge_mask && full[q].ext[0] == w.ext[0] && full[q].ext[1] == w.ext[1] &&
                    full[q].ext[2] == w.ext[2] && full[q].ext_neg == w.ext_neg) { found = (int)q; break; }
        if (found < 0) { found = (int)full.size(); full.push_back(w); }
        full_of[(size_t)t] = found;
    }
    const int nf = (int)full.size();
    int S = 0;
    for (const auto &us : full) {
        int Emax = std::max(us.ext[0], std::max(us.ext[1], us.ext[2]));
        if (us.merge_mask) Emax = std::max(Emax, us.ext_neg);
        int S_t = Emax;
        if (std::isfinite(P.R2hi)) S_t = (int)std::min((double)Emax, std::floor(std::sqrt(P.R2hi)));
        S = std::max(S, S_t);
    }
    P.S = S;
    S_all = S;
    max_cells_all = 1;
    if (S < 1) {       // radius below one cell: tables of padding only (the source cell is the kernel's own business)
        for (int t = 0; t < nt; ++t) {
            uint4 *a = nullptr, *b = nullptr;
            const size_t n = 4 * (size_t)threads;
            ASORA_HIP_TRY(hipMalloc(&a, n * sizeof(uint4))); st.geom_owned.push_back(a);
            ASORA_HIP_TRY(hipMalloc(&b, n * sizeof(uint4))); st.geom_owned.push_back(b);
            st.geom_bytes += 2 * n * sizeof(uint4);
            ASORA_HIP_TRY(hipMemsetAsync(a, 0, n * sizeof(uint4), st.stream));
            hipLaunchKernelGGL(fill_pad_kernel, dim3(4), dim3(256), 0, st.stream, b, n, 1u);
            out[(size_t)t] = OctGeomDev{a, b, 0, 0};
            step_after[(size_t)t] = std::vector<int>(1, 0);
        }
        ASORA_HIP_TRY(hipStreamSynchronize(st.stream));      // (a pipelined call launches on side streams: the tables must be complete)
        return 0;
    }

    // candidates per shell and table (closed form: the block sizes)
    std::vector<TableJob> jobs((size_t)nf);
    std::vector<size_t> shell_cand((size_t)nf * (S + 2), 0);
    size_t total_cand = 0;
    for (int t = 0; t < nf; ++t) {
        jobs[(size_t)t].spec = full[(size_t)t];
        jobs[(size_t)t].cand_base = total_cand;
        jobs[(size_t)t].meta_base = (size_t)t * (S + 2);
        jobs[(size_t)t].S_built = 0; jobs[(size_t)t].first_shell = 1; jobs[(size_t)t].cellA = jobs[(size_t)t].cellB = nullptr;
        size_t at = 0;
        shell_cand[jobs[(size_t)t].meta_base + 0] = 0;              // (shell 0 has no candidates: the corners of shell 1 are the source cell)
        for (int s = 1; s <= S; ++s) {
            ShellBlocks B;
            shell_blocks(full[(size_t)t], s, B);
            shell_cand[jobs[(size_t)t].meta_base + s] = at;
            at += (size_t)B.start[B.nblk];
        }
        shell_cand[jobs[(size_t)t].meta_base + S + 1] = at;
        total_cand += at;
    }
    bool any_aligned = false;
    for (const auto &us : full) any_aligned = any_aligned || (us.align_class >= 0 && us.face >= 0);

    DevArray<TableJob> d_jobs;
    DevArray<size_t> d_shell_cand, d_shell_entry;
    DevArray<uint32_t> d_rank, d_pos;
    DevArray<int> d_count, d_packed, d_len, d_flags;
    if (int rc = d_jobs.upload(jobs, st.stream)) return rc;
    if (int rc = d_shell_cand.upload(shell_cand, st.stream)) return rc;
    if (int rc = d_rank.alloc(total_cand)) return rc;
    if (int rc = d_pos.alloc(any_aligned ? total_cand : 1)) return rc;
    if (int rc = d_count.alloc((size_t)nf * (S + 2))) return rc;
    if (int rc = d_packed.alloc((size_t)nf * (S + 2))) return rc;
    hipLaunchKernelGGL(geometry_count_kernel, dim3((unsigned)S, (unsigned)nf), dim3(GB_THREADS), 0, st.stream,
                       (const TableJob *)d_jobs.p, (const size_t *)d_shell_cand.p, P, d_rank.p, d_pos.p, d_count.p, d_packed.p);
    ASORA_HIP_TRY(hipGetLastError());
    std::vector<int> h_count((size_t)nf * (S + 2), 0), h_packed((size_t)nf * (S + 2), 0);
    ASORA_HIP_TRY(hipMemcpyAsync(h_count.data(), d_count.p, h_count.size() * sizeof(int), hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipMemcpyAsync(h_packed.data(), d_packed.p, h_packed.size() * sizeof(int), hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));

    // layout of the full tables
    std::vector<Layout> L((size_t)nf);
    std::vector<size_t> shell_entry((size_t)nf * (S + 2), 0);
    std::vector<int> shell_len((size_t)nf * (S + 2), 0);
    uint32_t max_cells_full = 1;
    for (int t = 0; t < nf; ++t) {
        const size_t mb = jobs[(size_t)t].meta_base;
        h_count[mb] = 0; h_packed[mb] = 0;
        L[(size_t)t] = lay_out(h_count.data() + mb, h_packed.data() + mb, S, threads, wedges ? 0 : boxsize);
        for (int s = 0; s <= S + 1; ++s) { shell_entry[mb + s] = L[(size_t)t].entry[(size_t)s]; shell_len[mb + s] = L[(size_t)t].len[(size_t)s]; }
        max_cells_full = std::max(max_cells_full, L[(size_t)t].max_cells);
    }
    // (the zero slot of the FINAL tables: the largest shell over all of them -- for quarter sectors that of the wedge tables,
    //  known only after their own count; the full sectors then keep MARK-free temporaries with their own zero slot)
    std::vector<void *> temp_tables;
    struct FreeTemps { std::vector<void *> &v; ~FreeTemps() { for (void *q : v) (void)hipFree(q); } } free_temps{temp_tables};
    std::vector<uint4 *> tabA, tabB;
    std::vector<int> first_shell((size_t)nf, 1), inner((size_t)nf, 0);
    std::vector<size_t> fill_from((size_t)nf, 0);
    if (wedges) {           // the full sectors are temporaries
        tabA.assign((size_t)nf, nullptr); tabB.assign((size_t)nf, nullptr);
        for (int t = 0; t < nf; ++t) {
            ASORA_HIP_TRY(hipMalloc(&tabA[(size_t)t], L[(size_t)t].entries * sizeof(uint4))); temp_tables.push_back(tabA[(size_t)t]);
            ASORA_HIP_TRY(hipMalloc(&tabB[(size_t)t], L[(size_t)t].entries * sizeof(uint4))); temp_tables.push_back(tabB[(size_t)t]);
        }
    } else if (int rc = allocate_tables(st, full, L, threads, tabA, tabB, first_shell, fill_from, inner)) return rc;
    for (int t = 0; t < nf; ++t) {
        // padding everywhere first (a member of a family: behind the shells it shares, which the family's first member pads)
        const size_t n = L[(size_t)t].entries - fill_from[(size_t)t];
        uint4 *a = tabA[(size_t)t] + fill_from[(size_t)t], *b = tabB[(size_t)t] + fill_from[(size_t)t];
        ASORA_HIP_TRY(hipMemsetAsync(a, 0, n * sizeof(uint4), st.stream));
        hipLaunchKernelGGL(fill_pad_kernel, dim3((unsigned)std::min<size_t>(1024, (n + 255) / 256)), dim3(256), 0, st.stream, b, n, max_cells_full);
        jobs[(size_t)t].cellA = tabA[(size_t)t]; jobs[(size_t)t].cellB = tabB[(size_t)t]; jobs[(size_t)t].S_built = L[(size_t)t].S_built;
        jobs[(size_t)t].first_shell = first_shell[(size_t)t];
    }
    ASORA_HIP_TRY(hipMemcpyAsync(d_jobs.p, jobs.data(), jobs.size() * sizeof(TableJob), hipMemcpyHostToDevice, st.stream));
    if (int rc = d_shell_entry.upload(shell_entry, st.stream)) return rc;
    if (int rc = d_len.upload(shell_len, st.stream)) return rc;
    std::vector<int> zero2(2, 0);
    if (int rc = d_flags.upload(zero2, st.stream)) return rc;
    // on-sphere entries: a handful per table (lattice points with |d|^2 = R^2 to rounding)
    const unsigned sphere_cap = 1u << 20;
    DevArray<SphereRecord> d_sphere;
    DevArray<unsigned> d_nsphere;
    if (int rc = d_sphere.alloc(sphere_cap)) return rc;
    std::vector<unsigned> zero1(1, 0u);
    if (int rc = d_nsphere.upload(zero1, st.stream)) return rc;
    hipLaunchKernelGGL(geometry_write_kernel, dim3((unsigned)S, (unsigned)nf), dim3(GB_THREADS), 0, st.stream, (const TableJob *)d_jobs.p,
                       (const size_t *)d_shell_cand.p, (const size_t *)d_shell_entry.p, (const int *)d_len.p, P, (const uint32_t *)d_rank.p,
                       (const uint32_t *)d_pos.p, max_cells_full, d_flags.p, d_sphere.p, d_nsphere.p, wedges ? 0u : sphere_cap);
    ASORA_HIP_TRY(hipGetLastError());

    std::vector<uint4 *> final_A((size_t)nt, nullptr);
    if (!wedges) {
        max_cells_all = max_cells_full;
        for (int t = 0; t < nt; ++t) {
            out[(size_t)t] = OctGeomDev{jobs[(size_t)t].cellA, jobs[(size_t)t].cellB, L[(size_t)t].nsteps, 0, inner[(size_t)t], 0};
            step_after[(size_t)t] = L[(size_t)t].step_after_shell;
            final_A[(size_t)t] = jobs[(size_t)t].cellA;
        }
    } else {
        // ---- quarter sectors: mark, renumber, rewrite ----
        std::vector<WedgeJob> wj((size_t)nt);
        size_t keep_total = 0;
        for (int t = 0; t < nt; ++t) {
            const int f = full_of[(size_t)t];
            WedgeJob &w = wj[(size_t)t];
            w.fullA = jobs[(size_t)f].cellA; w.fullB = jobs[(size_t)f].cellB;
            w.meta_base = jobs[(size_t)f].meta_base; w.S_full = L[(size_t)f].S_built; w.wedge = specs[(size_t)t].wedge;
            w.keep_base = keep_total; keep_total += L[(size_t)f].entries;
            w.wmeta_base = (size_t)t * (S + 2); w.S_built = 0; w.first_shell = 1; w.cellA = w.cellB = nullptr;
        }
        DevArray<WedgeJob> d_wj;
        DevArray<unsigned char> d_keep;
        DevArray<uint32_t> d_newslot;
        DevArray<int> d_wcount, d_wlen;
        DevArray<size_t> d_wentry;
        if (int rc = d_wj.upload(wj, st.stream)) return rc;
        if (int rc = d_keep.alloc(keep_total)) return rc;
        if (int rc = d_newslot.alloc(keep_total)) return rc;
        if (int rc = d_wcount.alloc((size_t)nt * (S + 2))) return rc;
        ASORA_HIP_TRY(hipMemsetAsync(d_keep.p, 0, keep_total, st.stream));
        hipLaunchKernelGGL(wedge_seed_kernel, dim3((unsigned)S, (unsigned)nt), dim3(256), 0, st.stream, (const WedgeJob *)d_wj.p,
                           (const size_t *)d_shell_entry.p, (const int *)d_count.p, d_keep.p);
        for (int s = S; s >= 2; --s)
            hipLaunchKernelGGL(wedge_mark_kernel, dim3(32, (unsigned)nt), dim3(256), 0, st.stream, (const WedgeJob *)d_wj.p,
                               (const size_t *)d_shell_entry.p, (const int *)d_count.p, d_keep.p, s, max_cells_full);
        {   // Families of wedge tables (the octant variants of one sector and wedge under a clipped window): their full sectors
            // hold the same entries up to shell m, but what a wedge KEEPS of them follows from its outer shells, which differ.  Keep
            // the union of the members' needs there: the inner shells of all members then come out identical (and are shared below).
            std::vector<Layout> FL((size_t)nt);                   // the layout of each wedge table's FULL sector
            for (int t = 0; t < nt; ++t) FL[(size_t)t] = L[(size_t)full_of[(size_t)t]];
            for (const Family &f : find_families(specs, FL, threads)) {
                    const size_t n = f.prefix_entries;               // (entries of the full sectors before shell m + 1)
                    const unsigned blocks = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
                    const size_t first = wj[(size_t)f.members[0]].keep_base;
                    for (size_t q = 1; q < f.members.size(); ++q)
                        hipLaunchKernelGGL(keep_or_kernel, dim3(blocks), dim3(256), 0, st.stream, d_keep.p, first, wj[(size_t)f.members[q]].keep_base, n);
                    for (size_t q = 1; q < f.members.size(); ++q)
                        ASORA_HIP_TRY(hipMemcpyAsync(d_keep.p + wj[(size_t)f.members[q]].keep_base, d_keep.p + first, n, hipMemcpyDeviceToDevice, st.stream));
                }
        }
        hipLaunchKernelGGL(wedge_count_kernel, dim3((unsigned)S, (unsigned)nt), dim3(GB_THREADS), 0, st.stream, (const WedgeJob *)d_wj.p,
                           (const size_t *)d_shell_entry.p, (const int *)d_count.p, (const unsigned char *)d_keep.p, d_newslot.p, d_wcount.p);
        ASORA_HIP_TRY(hipGetLastError());
        std::vector<int> h_wcount((size_t)nt * (S + 2), 0);
        ASORA_HIP_TRY(hipMemcpyAsync(h_wcount.data(), d_wcount.p, h_wcount.size() * sizeof(int), hipMemcpyDeviceToHost, st.stream));
        ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
        std::vector<Layout> WL((size_t)nt);
        std::vector<size_t> wentry((size_t)nt * (S + 2), 0);
        std::vector<int> wlen((size_t)nt * (S + 2), 0);
        for (int t = 0; t < nt; ++t) {
            const size_t mb = wj[(size_t)t].wmeta_base;
            h_wcount[mb] = 0;
            WL[(size_t)t] = lay_out(h_wcount.data() + mb, h_wcount.data() + mb, S, threads, 0);
            for (int s = 0; s <= S + 1; ++s) { wentry[mb + s] = WL[(size_t)t].entry[(size_t)s]; wlen[mb + s] = WL[(size_t)t].len[(size_t)s]; }
            max_cells_all = std::max(max_cells_all, WL[(size_t)t].max_cells);
        }
        std::vector<uint4 *> wA, wB;
        std::vector<int> wfirst, winner;
        std::vector<size_t> wfill;
        if (int rc = allocate_tables(st, specs, WL, threads, wA, wB, wfirst, wfill, winner)) return rc;
        for (int t = 0; t < nt; ++t) {
            const size_t n = WL[(size_t)t].entries - wfill[(size_t)t];
            uint4 *a = wA[(size_t)t] + wfill[(size_t)t], *b = wB[(size_t)t] + wfill[(size_t)t];
            ASORA_HIP_TRY(hipMemsetAsync(a, 0, n * sizeof(uint4), st.stream));
            hipLaunchKernelGGL(fill_pad_kernel, dim3((unsigned)std::min<size_t>(1024, (n + 255) / 256)), dim3(256), 0, st.stream, b, n, max_cells_all);
            wj[(size_t)t].cellA = wA[(size_t)t]; wj[(size_t)t].cellB = wB[(size_t)t]; wj[(size_t)t].S_built = WL[(size_t)t].S_built;
            wj[(size_t)t].first_shell = wfirst[(size_t)t];
            out[(size_t)t] = OctGeomDev{wA[(size_t)t], wB[(size_t)t], WL[(size_t)t].nsteps, 0, winner[(size_t)t], 0};
            step_after[(size_t)t] = WL[(size_t)t].step_after_shell;
            final_A[(size_t)t] = wA[(size_t)t];
        }
        ASORA_HIP_TRY(hipMemcpyAsync(d_wj.p, wj.data(), wj.size() * sizeof(WedgeJob), hipMemcpyHostToDevice, st.stream));
        ASORA_HIP_TRY(hipMemsetAsync(d_nsphere.p, 0, sizeof(unsigned), st.stream));       // (the full sectors' pass counted theirs)
        if (int rc = d_wentry.upload(wentry, st.stream)) return rc;
        if (int rc = d_wlen.upload(wlen, st.stream)) return rc;
        hipLaunchKernelGGL(wedge_write_kernel, dim3((unsigned)S, (unsigned)nt), dim3(GB_THREADS), 0, st.stream, (const WedgeJob *)d_wj.p,
                           (const size_t *)d_shell_entry.p, (const int *)d_count.p, (const size_t *)d_wentry.p, (const int *)d_wlen.p,
                           (const uint32_t *)d_newslot.p, max_cells_full, max_cells_all, threads, d_flags.p, d_sphere.p, d_nsphere.p,
                           sphere_cap, 0u);
        ASORA_HIP_TRY(hipGetLastError());
        ASORA_HIP_TRY(hipStreamSynchronize(st.stream));            // the temporaries (full sectors, keep bytes) go out of scope below
    }

    // flags and the on-sphere entries
    int h_flags[2] = {0, 0};
    unsigned n_sph = 0;
    ASORA_HIP_TRY(hipMemcpyAsync(h_flags, d_flags.p, sizeof h_flags, hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipMemcpyAsync(&n_sph, d_nsphere.p, sizeof n_sph, hipMemcpyDeviceToHost, st.stream));
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    if (h_flags[0] & 1) return fail(11, "raytrace geometry: a cell of a unit reads a corner outside the unit (internal error)");
    if (h_flags[0] & 2) return fail(11, "raytrace geometry: a cell of shell 1 lies beyond the first three steps (internal error)");
    if (n_sph > sphere_cap) return fail(11, "raytrace geometry: more on-sphere entries than the list holds (internal error)");
    if (n_sph) {
        std::vector<SphereRecord> rec(n_sph);
        ASORA_HIP_TRY(hipMemcpy(rec.data(), d_sphere.p, n_sph * sizeof(SphereRecord), hipMemcpyDeviceToHost));
        for (const auto &r : rec)
            st.geom_sphere.push_back({reinterpret_cast<uint32_t *>(final_A[r.table] + r.entry) + 1, r.word_without_rate, r.a, r.b, r.c});
    }
    return 0;
}

} // namespace asora
