// geometry.hip -- the source-independent geometry tables of the raytrace kernel (host side), built once per (N, R, dr, launch
// shape) and kept on the device: which cells a shell of a unit holds, their path length and the shell-buffer slots of their
// four upstream corners.  The kernel that reads them is in raytrace.hip; the table format is described above it.
#include "asora_internal.hpp"
#include "rates_device.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace asora {

// ---------------------------------------------------------------------------------------------
// Source-independent geometry of one octant (host)
// ---------------------------------------------------------------------------------------------
// Faces of shell s follow the reference's branch order z, y, x (raytracing.cu:394,446,491; ties go
// to z, then y):  z-face dk = s: (a,b,s), a,b <= s;  y-face dj = s: (a,s,c), c < s;  x-face di = s:
// (s,b,c), b,c < s.  Inside a face the fastest index is the one that is contiguous in memory
// (a for the transposed z-face, c otherwise).  Tabulated per cell: packed offsets + face, the path
// length (raytracing.cu:444) and the shell-buffer slots of the four upstream corners; corners
// that would step across a zero offset or keep a transverse offset equal to s have bilinear weight
// exactly 0 (raytracing.cu:397-408) and are given the buffer's zero slot.
namespace {

struct HostGeom {
    std::vector<uint4> cellA, cellB;      // step-padded, see the kernel's table description
    int S = 0;
    int nsteps = 0;
    uint32_t max_cells = 1;
    bool inconsistent = false;   // a corner of non-zero weight was not found in the unit
    bool on_sphere = false;      // some cell needed the floating-point distance test (its result depends on dr): flagged CELL_SPHERE
    std::vector<int> step_after_shell;   // [s]: number of table steps up to and including shell s ([0] = 0)
};

inline bool inside_radius_reference(int a, int b, int c, double dr, double R2)
{
    // raytracing.cu:302-305,315 evaluated with every operation rounded on its own, i.e. as the Fortran path
    // (raytracing.f90:452-456,474) and the oracle evaluate it.  The CUDA library itself is built with nvcc's default
    // -fmad=true, which may contract xs*xs + ys*ys + zs*zs into fused multiply-adds: for a lattice point EXACTLY on the
    // sphere (integer R with integer solutions, e.g. (6,8,0) at R = 10) a CUDA build can classify a handful of surface
    // cells differently.  Parity on such cells is claimed against the un-fused evaluation only.
    volatile double xs = dr * (double)a, ys = dr * (double)b, zs = dr * (double)c;
    volatile double xx = xs * xs, yy = ys * ys, zz = zs * zs;
    volatile double d2 = xx + yy;
    d2 = d2 + zz;
    volatile double den = dr * dr;
    return d2 / den <= R2;
}

// A unit is a dependency-closed set of cells that one workgroup sweeps:
//   face = -1 : a whole octant;
//   face =  2 : the z-sector = all dk = s cells (closed: their corners are dk = s-1 cells);
//   face =  1 : the y-sector = the dj = s cells plus the plane {|dj| = |dk|} of the z-sector they read
//               (that plane only reads itself);
//   face =  0 : the x-sector = the di = s cells plus the planes {|di| = |dk|} (z-sector) and {|di| = |dj|}
//               (y-sector) they read (each of which only reads itself and the main diagonal, which is in
//               {|di| = |dk|}).
//   merge_mask bit ax : the unit covers BOTH signs of axis ax (mirrored octants / sectors in one workgroup, the plane
//               between them evaluated once), so that its rows along that axis -- when it is the memory-contiguous one for
//               a face -- are full chords of the sphere: fewer 64-B atomic requests per cell; and fewer, larger shells:
//               fewer lanes of padding.  All three axes merged: the whole sphere in one workgroup, nothing evaluated twice.
// A cell is RATED by its home unit only (its own face's sector); the copies a sector keeps of another
// sector's plane are evaluated for their column density but not rated.
struct UnitSpec {
    int face = -1;
    int merge_mask = 0;       // bit ax: the unit covers BOTH signs of axis ax (its own side + the mirrored one)
    int ext[3] = {0, 0, 0};   // periodic-window extent of each axis on the side this unit looks at
    int ext_neg = 0;          // extent on the mirrored side of a merged axis
    int wedge = -1;           // 0..3: a quarter of the sector (restrict_to_wedge), -1: the whole unit
};

// boxsize > 0 (sub-box tables): every shell that closes a sub-box (a multiple of boxsize) is followed by all-invalid steps
// up to a whole triple of steps, so that a launch can sweep exactly one sub-box with the three-step pipeline
// align_class in 0..7 (units of ONE face only): the tables of the sources whose position along the memory-contiguous axis of
// that face (k for the x- and y-sector, i for the z-sector, whose rates go to the [k][j][i] twin) is align_class modulo 8.
// The rated cells of a row that fall into one 64-byte line of the rate grid then never straddle two 64-lane waves (invalid
// entries fill the wave up instead: ~3 % more lane-steps), so a wave's atomics leave as whole-line requests: 6.4 instead of
// 5.9 doubles per request at r_RT = 32.  -1: entries packed densely.
void build_unit_geometry(HostGeom &h, const UnitSpec &us, double R, double dr, int q_max, uint32_t zero_slot_marker,
                         int RT_THREADS, int boxsize = 0, int align_class = -1)
{
    const double R2 = R * R;
    const double R2hi = R2 * (1.0 + 1e-9) + 1e-9;
    int Emax = std::max(us.ext[0], std::max(us.ext[1], us.ext[2]));
    if (us.merge_mask) Emax = std::max(Emax, us.ext_neg);
    int S = Emax;
    if (std::isfinite(R2hi)) S = (int)std::min((double)Emax, std::floor(std::sqrt(R2hi)));
    h.S = S;
    static const int DOM[3] = {0, 1, 2};                 // dominant axis of face 0 (x), 1 (y), 2 (z)
    static const int TE[3] = {1, 0, 0}, TF[3] = {2, 2, 1};   // transverse axes (e,f): x:(y,z) y:(x,z) z:(x,y)
    // slot maps of the previous / current shell, keyed by face, sign of the dominant offset and the signed
    // transverse offsets
    const size_t P2 = 2 * (size_t)S + 1;
    const size_t map_size = 3 * 2 * P2 * P2;
    std::vector<uint32_t> slot_prev(map_size, zero_slot_marker), slot_cur(map_size, zero_slot_marker);
    auto key_of = [&](const int x[3]) -> size_t {
        // face by magnitudes, ties z, then y (raytracing.cu:394,446,491)
        const int aa = std::abs(x[0]), bb = std::abs(x[1]), cc = std::abs(x[2]);
        const int t = std::max(aa, std::max(bb, cc));
        const int face = (cc == t) ? 2 : (bb == t) ? 1 : 0;
        const int d = DOM[face], e = TE[face], f = TF[face];
        const size_t neg = x[d] < 0 ? 1 : 0;
        return ((size_t)(face * 2) + neg) * P2 * P2 + (size_t)(x[e] + S) * P2 + (size_t)(x[f] + S);
    };
    {   const int origin[3] = {0, 0, 0};
        slot_prev[key_of(origin)] = 0;       // the source cell sits in slot 0 of shell 0
    }
    // 0: outside, 1: inside, 2: ON the sphere to rounding -- tabulated and evaluated whatever the floating-point test says
    // (nothing that is kept reads it: its readers lie strictly further out), rated as inside_radius_reference decides for
    // the current dr; ensure_geometry re-decides these cells in place when dr changes
    auto in_sphere = [&](int a, int b, int c) -> int {
        if (a + b + c > q_max) return 0;                                       // raytracing.cu:101,198
        const double n2 = (double)a * a + (double)b * b + (double)c * c;
        if (n2 > R2hi) return 0;
        if (n2 < R2 * (1.0 - 1e-9) - 1e-9) return 1;
        return 2;
    };
    const uint4 pad_a = {0u, 0u, 0u, 0u};
    const uint4 pad_b = {zero_slot_marker, zero_slot_marker, zero_slot_marker, zero_slot_marker};
    for (int s = 1; s <= S; ++s) {
        std::fill(slot_cur.begin(), slot_cur.end(), zero_slot_marker);
        const double sd = (double)s;
        uint32_t count = 0;
        auto emit = [&](const int x[3], int face) {
            const int a = std::abs(x[0]), b = std::abs(x[1]), c = std::abs(x[2]);
            const int where = in_sphere(a, b, c);
            if (!where) return;
            bool rate = where == 1 || inside_radius_reference(a, b, c, dr, R2);
            bool sphere = where == 2;
            if (us.face >= 0 && face != us.face) {
                // a foreign cell: kept only if this sector reads it
                const bool keep = (us.face == 1) ? (face == 2 && b == c)
                                                 : (us.face == 0) ? ((face == 2 && a == c) || (face == 1 && a == b)) : false;
                if (!keep) return;
                rate = false;
                sphere = false;
            }
            if (sphere) h.on_sphere = true;
            const int d = DOM[face], e = TE[face], f = TF[face];
            const int U = std::abs(x[e]), V = std::abs(x[f]);
            const int sgd = x[d] < 0 ? -1 : 1, sge = x[e] < 0 ? -1 : 1, sgf = x[f] < 0 ? -1 : 1;
            const double u = (double)U, v = (double)V;
            const bool em = U >= 1, e0 = U <= s - 1, fm = V >= 1, f0 = V <= s - 1;
            auto corner = [&](int uu, int vv) -> uint32_t {
                int n[3];
                n[d] = sgd * (s - 1); n[e] = sge * uu; n[f] = sgf * vv;
                return slot_prev[key_of(n)];
            };
            uint4 nb;
            nb.x = (em && fm) ? corner(U - 1, V - 1) : zero_slot_marker;
            nb.y = (e0 && fm) ? corner(U, V - 1) : zero_slot_marker;
            nb.z = (em && f0) ? corner(U - 1, V) : zero_slot_marker;
            nb.w = (e0 && f0) ? corner(U, V) : zero_slot_marker;
            // A transverse offset EQUAL to s (cube edges, the diagonal): the two corners at that offset do not exist in shell s-1 and
            // carry weight 1 - s * (1/s), which is 0 for most s and 2^-53 for s = 49, 98, 103, 107, ... (230 values below 2048).  The
            // reference multiplies such a speck with the column density of a real neighbour; pointed at the zero slot it would meet
            // the value 0, i.e. a weight 1 / max(0.6, 0) instead of 1 / (c sigma), amplified by c sigma / 0.6 (round 6: 5e-9 at the
            // corners of a +-64 cube in tau = 200 cells, found in the sub-box kernel).  So they ALIAS their existing neighbour at
            // offset s - 1: same LDS word, a weight of 0 or 2^-53 on a real value.
            if (!e0) { nb.y = nb.x; nb.w = nb.z; }
            if (!f0) { nb.z = nb.x; nb.w = nb.y; }
            {   // every corner that carries weight must be part of this unit
                const double fu = U == s ? 1.0 : u / sd, fv = V == s ? 1.0 : v / sd;
                const double wts[4] = {fu * fv, fv * (1.0 - fu), fu * (1.0 - fv), (1.0 - fu) * (1.0 - fv)};
                const uint32_t sl[4] = {nb.x, nb.y, nb.z, nb.w};
                for (int q = 0; q < 4; ++q) if (wts[q] != 0.0 && sl[q] == zero_slot_marker) h.inconsistent = true;
            }
            const double path = std::sqrt((u * u + v * v) / (sd * sd) + 1.0);  // raytracing.cu:444
            uint64_t pbits;
            std::memcpy(&pbits, &path, sizeof pbits);
            uint32_t negbits = 0;
            for (int ax = 0; ax < 3; ++ax) if (((us.merge_mask >> ax) & 1) && x[ax] < 0) negbits |= 1u << ax;
            uint4 ca;
            ca.x = (uint32_t)a | ((uint32_t)b << 10) | ((uint32_t)c << 20) | ((uint32_t)face << 30);
            ca.y = count | CELL_VALID | (rate ? CELL_RATE : 0u) | (sphere ? CELL_SPHERE : 0u) | (negbits << CELL_NEG_SHIFT) |
                   (((a == 0 ? 1u : 0u) | (b == 0 ? 2u : 0u) | (c == 0 ? 4u : 0u)) << CELL_ZERO_SHIFT);
            ca.z = (uint32_t)(pbits & 0xffffffffu);
            ca.w = (uint32_t)(pbits >> 32);
            h.cellA.push_back(ca);
            h.cellB.push_back(nb);
            slot_cur[key_of(x)] = count++;
        };
        // signed range of a transverse axis whose magnitude may reach `maxmag`
        auto lo_of = [&](int axis, int maxmag) { return ((us.merge_mask >> axis) & 1) ? -std::min(maxmag, us.ext_neg) : 0; };
        auto hi_of = [&](int axis, int maxmag) { return std::min(maxmag, us.ext[axis]); };
        for (int face = 2; face >= 0; --face) {
            if (us.face == 2 && face != 2) continue;          // the z-sector holds z-face cells only
            if (us.face == 1 && face == 0) continue;          // the y-sector never needs x-face cells
            const int d = DOM[face];
            // the memory-contiguous transverse axis runs fastest: a (axis 0) on the z-face, c (axis 2) otherwise
            const int fast = face == 2 ? 0 : 2;
            const int slow = face == 2 ? 1 : (face == 1 ? 0 : 1);
            const int max_fast = (face == 2) ? s : s - 1;                     // y/x-face: |dk| < s
            const int max_slow = (face == 0) ? s - 1 : s;                     // x-face: |dj| < s
            for (int dsgn = 1; dsgn >= -1; dsgn -= 2) {
                if (dsgn < 0 && !((us.merge_mask >> d) & 1)) break;
                const int ext_d = dsgn > 0 ? us.ext[d] : us.ext_neg;
                if (s > ext_d) continue;
                for (int sl = lo_of(slow, max_slow); sl <= hi_of(slow, max_slow); ++sl)
                    for (int fa = lo_of(fast, max_fast); fa <= hi_of(fast, max_fast); ++fa) {
                        int x[3];
                        x[d] = dsgn * s; x[slow] = sl; x[fast] = fa;
                        emit(x, face);
                    }
            }
        }
        if (count == 0) break;                        // nothing further out either
        if (align_class >= 0 && us.face >= 0) {
            // repack the shell (it starts on a step, hence a wave boundary): runs of rated own-face cells of one row within one
            // 64-byte line stay in one wave
            const size_t end = h.cellA.size(), begin = end - count;
            const int fast = us.face == 2 ? 0 : 2;
            const uint32_t fast_mask = ~((1023u << (10 * fast)));
            const uint32_t other_neg = (7u & ~(1u << fast)) << CELL_NEG_SHIFT;
            auto fast_of = [&](const uint4 &ca) -> int {
                const int mag = (int)((ca.x >> (10 * fast)) & 1023u);
                return ((ca.y >> (CELL_NEG_SHIFT + fast)) & 1u) ? -mag : mag;
            };
            auto line_of = [&](int f) -> int { return (align_class + f + 8192) >> 3; };
            auto rated = [&](const uint4 &ca) -> bool { return (ca.y & (CELL_RATE | CELL_SPHERE)) != 0 && (int)(ca.x >> 30) == us.face; };
            std::vector<uint4> ra, rb;
            ra.reserve(count + count / 8); rb.reserve(count + count / 8);
            size_t q = begin;
            while (q < end) {
                size_t e = q + 1;
                if (rated(h.cellA[q])) {
                    const int line = line_of(fast_of(h.cellA[q]));
                    while (e < end && rated(h.cellA[e]) && (h.cellA[e].x & fast_mask) == (h.cellA[q].x & fast_mask) &&
                           (h.cellA[e].y & other_neg) == (h.cellA[q].y & other_neg) &&
                           fast_of(h.cellA[e]) == fast_of(h.cellA[e - 1]) + 1 && line_of(fast_of(h.cellA[e])) == line) ++e;
                }
                const size_t n = e - q, pos = ra.size() % 64;
                if (pos + n > 64) for (size_t t = pos; t < 64; ++t) { ra.push_back(pad_a); rb.push_back(pad_b); }
                for (size_t t = q; t < e; ++t) { ra.push_back(h.cellA[t]); rb.push_back(h.cellB[t]); }
                q = e;
            }
            h.cellA.resize(begin); h.cellB.resize(begin);
            h.cellA.insert(h.cellA.end(), ra.begin(), ra.end());
            h.cellB.insert(h.cellB.end(), rb.begin(), rb.end());
        }
        // pad the shell to whole steps and flag every entry of its last step
        while (h.cellA.size() % RT_THREADS) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
        for (size_t q = h.cellA.size() - RT_THREADS; q < h.cellA.size(); ++q) h.cellA[q].y |= CELL_LAST;
        if (boxsize > 0 && s % boxsize == 0)
            while ((h.cellA.size() / (size_t)RT_THREADS) % 3) for (int q = 0; q < RT_THREADS; ++q) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
        h.max_cells = std::max(h.max_cells, count);
        slot_prev.swap(slot_cur);
        h.step_after_shell.resize((size_t)s + 1, 0);
        h.step_after_shell[(size_t)s] = (int)(h.cellA.size() / (size_t)RT_THREADS);
    }
    // the kernel walks the steps three at a time and looks two steps ahead: pad to a multiple of
    // three steps and append four all-invalid steps so that every load stays inside the tables
    while ((h.cellA.size() / (size_t)RT_THREADS) % 3) for (int q = 0; q < RT_THREADS; ++q) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
    h.nsteps = (int)(h.cellA.size() / (size_t)RT_THREADS);
    if (h.step_after_shell.empty()) h.step_after_shell.push_back(0);
    h.step_after_shell.back() = h.nsteps;         // (the last shell's count includes the closing padding)
    for (int q = 0; q < 4 * RT_THREADS; ++q) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
}

// A quarter of a sector, as a unit of its own: for a handful of sources the call lasts as long as ONE workgroup, so a
// source is cut into more of them.  The own-face cells of a sector are split by their transverse offsets (U, V) of
// shell s into four wedges, 2U > s and 2V > s or not (every shell is partitioned exactly).  A wedge is not closed under
// the interpolation's dependencies -- a cell reads (U-1, V-1) ... (U, V) of shell s-1, so the cone that feeds a wedge
// widens towards the source -- hence the unit is the wedge's RATED cells plus everything they (transitively) read:
// found by marking backwards from the outermost shell through the corner links of the full sector's tables, then
// renumbering the shell-buffer slots of what is kept.  Evaluations double (each wedge re-derives the inner part of the
// sector), workgroups quadruple: the single-source trace gets about twice as fast.
struct WedgeEntry { uint4 a, b; };
using WedgeKeep = std::vector<std::vector<char>>;          // [shell index = s - 1][slot]

static std::vector<std::vector<WedgeEntry>> shells_of(const HostGeom &full, int RT_THREADS)
{
    std::vector<std::vector<WedgeEntry>> shells;
    std::vector<WedgeEntry> cur;
    const size_t nent = (size_t)full.nsteps * RT_THREADS;
    for (size_t st0 = 0; st0 < nent; st0 += RT_THREADS) {
        for (int q = 0; q < RT_THREADS; ++q)
            if (full.cellA[st0 + q].y & CELL_VALID) cur.push_back({full.cellA[st0 + q], full.cellB[st0 + q]});
        if (full.cellA[st0].y & CELL_LAST) { shells.push_back(cur); cur.clear(); }
    }
    return shells;
}

static bool entry_in_wedge(const uint4 &a, int wedge)           // (own-face cells: rated, or on the sphere and possibly rated later)
{
    if (!(a.y & (CELL_RATE | CELL_SPHERE))) return false;
    const int ca = a.x & 1023, cb = (a.x >> 10) & 1023, cc = (a.x >> 20) & 1023, face = a.x >> 30;
    const int s = std::max(ca, std::max(cb, cc));
    const int U = face == 0 ? cb : ca, V = face == 2 ? cb : cc;
    return ((2 * U > s ? 1 : 0) | (2 * V > s ? 2 : 0)) == wedge;
}

// backward marking: what the wedge's rated cells read, transitively
WedgeKeep wedge_keep(const HostGeom &full, int wedge, int RT_THREADS, uint32_t zero_slot_marker)
{
    const auto shells = shells_of(full, RT_THREADS);
    WedgeKeep keep(shells.size());
    for (size_t si = 0; si < shells.size(); ++si) {
        keep[si].assign(shells[si].size(), 0);
        for (size_t q = 0; q < shells[si].size(); ++q) keep[si][q] = entry_in_wedge(shells[si][q].a, wedge) ? 1 : 0;
    }
    for (size_t si = shells.size(); si-- > 1;)
        for (size_t q = 0; q < shells[si].size(); ++q) {
            if (!keep[si][q]) continue;
            const uint32_t c[4] = {shells[si][q].b.x, shells[si][q].b.y, shells[si][q].b.z, shells[si][q].b.w};
            for (uint32_t slot : c) if (slot != zero_slot_marker) keep[si - 1][slot] = 1;     // slot == rank in its shell
        }
    return keep;
}

HostGeom restrict_to_wedge(const HostGeom &full, int wedge, int RT_THREADS, uint32_t zero_slot_marker, const WedgeKeep &keep)
{
    using Entry = WedgeEntry;
    const auto shells = shells_of(full, RT_THREADS);
    auto in_wedge = [wedge](const uint4 &a) -> bool { return entry_in_wedge(a, wedge); };
    HostGeom h;
    h.S = full.S; h.inconsistent = full.inconsistent; h.on_sphere = full.on_sphere;
    const uint4 pad_a = {0u, 0u, 0u, 0u};
    const uint4 pad_b = {zero_slot_marker, zero_slot_marker, zero_slot_marker, zero_slot_marker};
    std::vector<uint32_t> prev_map, cur_map;
    for (size_t si = 0; si < shells.size(); ++si) {
        cur_map.assign(shells[si].size(), zero_slot_marker);
        uint32_t count = 0;
        for (size_t q = 0; q < shells[si].size(); ++q) {
            if (!keep[si][q]) continue;
            Entry e = shells[si][q];
            const uint32_t flags = e.a.y & ((7u << CELL_NEG_SHIFT) | (7u << CELL_ZERO_SHIFT));
            const bool mine = in_wedge(e.a);
            e.a.y = count | CELL_VALID | flags | ((mine && (e.a.y & CELL_RATE)) ? CELL_RATE : 0u) |
                    ((mine && (e.a.y & CELL_SPHERE)) ? CELL_SPHERE : 0u);
            if (si > 0) {       // corners of shell 1 point into shell 0 (the source cell, slot 0): unchanged
                auto remap = [&](uint32_t slot) -> uint32_t {
                    if (slot == zero_slot_marker) return slot;
                    if (prev_map[slot] == zero_slot_marker) h.inconsistent = true;            // a read corner was not kept
                    return prev_map[slot];
                };
                e.b.x = remap(e.b.x); e.b.y = remap(e.b.y); e.b.z = remap(e.b.z); e.b.w = remap(e.b.w);
            }
            h.cellA.push_back(e.a); h.cellB.push_back(e.b);
            cur_map[q] = count++;
        }
        if (count == 0) break;                        // the wedge holds nothing from here on (tiny radii)
        while (h.cellA.size() % RT_THREADS) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
        for (size_t q = h.cellA.size() - RT_THREADS; q < h.cellA.size(); ++q) h.cellA[q].y |= CELL_LAST;
        h.max_cells = std::max(h.max_cells, count);
        prev_map.swap(cur_map);
    }
    while ((h.cellA.size() / (size_t)RT_THREADS) % 3) for (int q = 0; q < RT_THREADS; ++q) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
    h.nsteps = (int)(h.cellA.size() / (size_t)RT_THREADS);
    for (int q = 0; q < 4 * RT_THREADS; ++q) { h.cellA.push_back(pad_a); h.cellB.push_back(pad_b); }
    return h;
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

// log2 table of log2_pos: interval centres c = 1 + (i + 1/2)/128, entries {1/c, log2 c}
int ensure_logtab(State &st)
{
    if (st.logtab_dev) return 0;
    std::vector<double2> lt(LOG_TABLE_SIZE);
    for (int i = 0; i < LOG_TABLE_SIZE; ++i) {
        const long double c = 1.0L + ((long double)i + 0.5L) / (long double)LOG_TABLE_SIZE;
#if ASORA_FREXP_LOG       // the mantissa comes as m in [0.5, 1): c/2 is its interval centre (rates_device.hpp)
        lt[i].x = (double)(2.0L / c);
        lt[i].y = (double)(std::log2(c) - 1.0L);
#else
        lt[i].x = (double)(1.0L / c);
        lt[i].y = (double)std::log2(c);
#endif
    }
    double2 *d = nullptr;
    ASORA_HIP_TRY(hipMalloc(&d, lt.size() * sizeof(double2)));
    ASORA_HIP_TRY(hipMemcpy(d, lt.data(), lt.size() * sizeof(double2), hipMemcpyHostToDevice));
    st.logtab_dev = d;
    return 0;
}

void release_geometry(State &st)
{
    for (void *q : st.geom_owned) (void)hipFree(q);
    st.geom_owned.clear();
    st.geom_bytes = 0;
    st.geom_sphere.clear();
    st.geom_valid = false;
}

// dr has changed: re-decide the RATE bit of the cells on the sphere (a handful: the lattice points with |d|^2 = R^2) and
// write the words in place, behind whatever the library's stream is still running with the old ones
__global__ void patch_words_kernel(const unsigned long long *__restrict__ pairs, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) *reinterpret_cast<uint32_t *>(pairs[2 * q]) = (uint32_t)pairs[2 * q + 1];
}

static int patch_sphere_cells(State &st, double R, double dr)
{
    const size_t n = st.geom_sphere.size();
    std::vector<unsigned long long> pairs(2 * n);
    for (size_t q = 0; q < n; ++q) {
        const State::SphereCell &c = st.geom_sphere[q];
        pairs[2 * q] = (unsigned long long)reinterpret_cast<uintptr_t>(c.dev_word);
        pairs[2 * q + 1] = c.word_without_rate | (inside_radius_reference(c.a, c.b, c.c, dr, R * R) ? CELL_RATE : 0u);
    }
    if (n > st.geom_patch_cap) {
        if (st.geom_patch_dev) (void)hipFree(st.geom_patch_dev);
        st.geom_patch_dev = nullptr; st.geom_patch_cap = 0;
        ASORA_HIP_TRY(hipMalloc(&st.geom_patch_dev, 2 * n * sizeof(unsigned long long)));
        st.geom_patch_cap = n;
    }
    // (a blocking copy from pageable memory: `pairs` may go out of scope right after; the kernel is stream-ordered)
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    for (int q = 0; q < 2; ++q) if (st.side[q]) ASORA_HIP_TRY(hipStreamSynchronize(st.side[q]));   // (traces of an earlier pipelined call)
    ASORA_HIP_TRY(hipMemcpy(st.geom_patch_dev, pairs.data(), 2 * n * sizeof(unsigned long long), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(patch_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.stream,
                       (const unsigned long long *)st.geom_patch_dev, (int)n);
    ASORA_HIP_TRY(hipGetLastError());
    // a pipelined call (asora_raytrace_begin) traces on the side streams, which only wait for what the main stream had
    // done when the call began: they must not read the patched words before the patch has run
    if (st.main_ready) {
        ASORA_HIP_TRY(hipEventRecord(st.main_ready, st.stream));
        for (int q = 0; q < 2; ++q) if (st.side[q]) ASORA_HIP_TRY(hipStreamWaitEvent(st.side[q], st.main_ready, 0));
    }
    return 0;
}

// Build (or reuse) the geometry tables for this (N, R, dr).  dr only enters through the
// classification of cells sitting exactly on the sphere (see inside_radius_reference).
int ensure_geometry(State &st, RtParams &p, int threads, int units, const SubboxGeometry *sbg, bool aligned)
{
    // aligned (units of one face: 6 or 12 per source): eight tables per unit, [class * units + unit], see build_unit_geometry
    if (aligned && !(units == 6 || units == 12)) return fail(11, "raytrace geometry: aligned tables for this kind of unit (internal error)");
    const int classes = aligned ? 8 : 1, tables = units * classes;
    const int N = p.N;
    // (the Fortran path has no octahedron bound and its own range instead of the ASORA window)
    const int q_max = sbg ? (1 << 28) : (int)std::ceil(1.73205080757 * std::min(p.R, 1.73205080757 * N / 2.0));   // raytracing.cu:14,101
    const int ext_pos = sbg ? sbg->ext_r : N / 2 - 1 + (N % 2);                                  // raytracing.cu:122
    const int ext_neg = sbg ? sbg->ext_l : N / 2;                                                // raytracing.cu:123
    const int boxsize = sbg ? sbg->boxsize : 0;
    // dr only matters for the cells that sit exactly on the sphere (a cosmological run changes dr every step): their RATE
    // bits are re-decided in place
    const bool on_host = st.opt[ASORA_OPT_GEOMETRY_ON_HOST] != 0;
    if (st.geom_valid && st.geom_on_host == on_host && st.geom_N == N && st.geom_R == p.R && st.geom_threads == threads && st.geom_units == units &&
        st.geom_aligned == aligned && st.geom_subbox == (sbg ? 1 : 0) && (!sbg || (st.geom_ext_r == ext_pos && st.geom_ext_l == ext_neg && st.geom_boxsize == boxsize))) {
        if (st.geom_dr != p.dr && !st.geom_sphere.empty()) {
            if (int rc = patch_sphere_cells(st, p.R, p.dr)) return rc;
        }
        st.geom_dr = p.dr;
        for (int o = 0; o < tables; ++o) p.geom[o] = st.geom_host[o];
        p.units = units;
        p.aligned = aligned ? 1 : 0;
        p.logtab = st.logtab_dev; p.S = st.geom_S; p.max_cells = st.geom_max_cells;
        return 0;
    }
    release_geometry(st);

    // Units of a source:
    //    8: unit = octant                              (bit ax of the octant index = negative side of axis ax)
    //   24: unit = sector*8 + octant                   (sector = face code 0:x 1:y 2:z)
    //    4: unit = q, two whole octants mirrored in x  (q = sign bits of (y,z))
    //   12: unit = sector*4 + q, two mirrored sectors  (z-sector mirrored in x: q = sign bits of (y,z);
    //                                                   y- and x-sector mirrored in z: q = sign bits of (x,y))
    //    1: the whole sphere                           (all three axes merged: no cell is evaluated twice, every row a full chord)
    //    2: unit = sign of y, a half sphere            (x and z merged: the rows of every face are full chords)
    //    3: unit = sector, all signs                   (a third of the LDS of the whole sphere)
    //    6: unit = sector*2 + sign of the dominant offset, both transverse axes merged
    // Units with the same sector and the same periodic window share one table; when the sphere does not reach
    // the window on any axis all of a sector's units are identical.
    const double R2hi_all = p.R * p.R * (1.0 + 1e-9) + 1e-9;
    const bool unclipped = std::isfinite(R2hi_all) && std::floor(std::sqrt(R2hi_all)) <= (double)std::min(ext_pos, ext_neg);
    //   96: unit = wedge*24 + sector*8 + octant       (a quarter of a sector and what it reads: restrict_to_wedge)
    UnitSpec spec[MAX_UNITS];
    int info[MAX_UNITS];
    for (int u = 0; u < units; ++u) {
        int neg[3] = {0, 0, 0};
        UnitSpec &us = spec[u];
        if (units == 96) {
            us.wedge = u / 24;
            us.face = (u % 24) >> 3;
            us.merge_mask = 0;
            for (int ax = 0; ax < 3; ++ax) neg[ax] = ((u & 7) >> ax) & 1;
        } else if (units == 4) {                      // two whole octants mirrored in x: q = sign bits of (y,z)
            us.face = -1;
            us.merge_mask = 1;
            neg[1] = u & 1; neg[2] = (u >> 1) & 1;
        } else if (units == 12) {
            us.face = u >> 2;
            us.merge_mask = us.face == 2 ? 1 : 4;
            const int q = u & 3;
            if (us.face == 2) { neg[1] = q & 1; neg[2] = (q >> 1) & 1; }
            else              { neg[0] = q & 1; neg[1] = (q >> 1) & 1; }
        } else if (units == 1) {
            us.face = -1;
            us.merge_mask = 7;
        } else if (units == 2) {
            us.face = -1;
            us.merge_mask = 5;
            neg[1] = u & 1;
        } else if (units == 3) {
            us.face = u;
            us.merge_mask = 7;
        } else if (units == 6) {                      // a sector with both signs of its two transverse axes: unit = sector*2 + sign of the dominant offset
            us.face = u >> 1;
            us.merge_mask = 7 & ~(1 << us.face);      // (face code = dominant axis: 0 x, 1 y, 2 z)
            neg[us.face] = u & 1;
        } else {
            us.face = units == 24 ? (u >> 3) : -1;
            us.merge_mask = 0;
            for (int ax = 0; ax < 3; ++ax) neg[ax] = ((u & 7) >> ax) & 1;
        }
        for (int ax = 0; ax < 3; ++ax) us.ext[ax] = neg[ax] ? ext_neg : ext_pos;
        us.ext_neg = ext_neg;
        // exactly one unit rates the source cell: the all-positive one (of the z-sector when there are sectors)
        const bool rates_source = !neg[0] && !neg[1] && !neg[2] && (us.face == -1 || us.face == 2) && us.wedge <= 0;
        info[u] = neg[0] | (neg[1] << 1) | (neg[2] << 2) | (us.merge_mask << 3) | (rates_source ? 64 : 0) | ((us.face + 1) << 8);
    }
    int owner[MAX_UNITS];
    for (int u = 0; u < units; ++u) {
        owner[u] = u;
        for (int u2 = 0; u2 < u; ++u2) {
            if (spec[u2].face != spec[u].face || spec[u2].merge_mask != spec[u].merge_mask || spec[u2].wedge != spec[u].wedge) continue;
            if (unclipped || (spec[u].ext[0] == spec[u2].ext[0] && spec[u].ext[1] == spec[u2].ext[1] &&
                              spec[u].ext[2] == spec[u2].ext[2])) {
                owner[u] = owner[u2];
                break;
            }
        }
    }
    OctGeomDev od[MAX_UNITS];
    int Smax = 0;
    uint32_t max_cells = 1;
    std::vector<int> steps_after[12];
    if (!on_host) {
        // ---- the default: the distinct tables are built on the device (geometry_device.hip) ----
        std::vector<GeomTableSpec> specs;
        std::vector<int> where((size_t)tables, -1);
        for (int cls = 0; cls < classes; ++cls)
            for (int u = 0; u < units; ++u) {
                if (owner[u] != u) continue;
                GeomTableSpec g;
                g.face = spec[u].face; g.merge_mask = spec[u].merge_mask; g.ext_neg = spec[u].ext_neg; g.wedge = spec[u].wedge;
                for (int ax = 0; ax < 3; ++ax) g.ext[ax] = spec[u].ext[ax];
                g.align_class = aligned ? cls : -1;
                where[(size_t)(cls * units + u)] = (int)specs.size();
                specs.push_back(g);
            }
        std::vector<OctGeomDev> built;
        std::vector<std::vector<int>> after;
        if (int rc = build_geometry_on_device(st, specs, p.R, p.dr, q_max, threads, boxsize, built, after, Smax, max_cells)) {
            release_geometry(st);
            return rc;
        }
        std::vector<int> at((size_t)specs.size(), -1);          // a place of every distinct table in od[]
        for (int v = tables - 1; v >= 0; --v) at[(size_t)where[(size_t)((v / units) * units + owner[v % units])]] = v;
        for (int v = 0; v < tables; ++v) {
            const int src = where[(size_t)((v / units) * units + owner[v % units])];
            od[v] = built[(size_t)src];
            od[v].info = info[v % units];
            if (od[v].inner) od[v].inner = (od[v].inner & ~255) | at[(size_t)(od[v].inner & 255)];        // (shared inner shells: which table holds them)
        }
        for (int u = 0; u < units && u < 12; ++u) steps_after[u] = after[(size_t)where[(size_t)owner[u]]];
    } else {
    std::vector<HostGeom> hg(tables);       // [class * units + unit]; the distinct ones are at [class * units + owner[unit]]
    const bool geom_timing = getenv("ASORA_GEOM_TIMING") != nullptr;
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    double t_sectors = t_begin;
    const uint32_t MARK = 0xffffffffu;
    {   // the tables of the distinct units are independent: one host thread each (a whole-box trace tabulates N^3
        // cells per unit set -- ~1 s on one core at 320^3)
        std::vector<std::thread> workers;
        const double R_all = p.R, dr_all = p.dr;
        if (units != 96) {
            for (int cls = 0; cls < classes; ++cls) {
                for (int u = 0; u < units; ++u) {
                    if (owner[u] != u) continue;
                    workers.emplace_back([&hg, &spec, u, cls, units, aligned, R_all, dr_all, q_max, threads, boxsize]() {
                        build_unit_geometry(hg[cls * units + u], spec[u], R_all, dr_all, q_max, MARK, threads, boxsize, aligned ? cls : -1);
                    });
                }
                for (auto &w : workers) w.join();
                workers.clear();
            }
        } else {
            // quarter sectors: every distinct sector ONCE (units 0..23 are the wedge-0 units, one per sector and octant),
            // then its four wedges and what they read, again side by side
            std::vector<HostGeom> sector(24);
            for (int v = 0; v < 24; ++v) {
                if (owner[v] != v) continue;
                workers.emplace_back([&sector, &spec, v, R_all, dr_all, q_max, threads]() {
                    UnitSpec whole = spec[v];
                    whole.wedge = -1;
                    build_unit_geometry(sector[v], whole, R_all, dr_all, q_max, MARK, threads);
                });
            }
            for (auto &w : workers) w.join();
            workers.clear();
            t_sectors = now_s();
            // what every wedge table keeps of its sector; the octant variants of one (sector, wedge) under a clipped window form a
            // family whose sectors hold the same entries up to shell m = the smallest extent: there the UNION of the members' needs
            // is kept, so that those shells come out identical in all members (geometry_device.hip shares their memory)
            std::vector<WedgeKeep> keep(units);
            for (int u = 0; u < units; ++u) {
                if (owner[u] != u) continue;
                workers.emplace_back([&keep, &sector, &spec, &owner, u, threads]() {
                    keep[u] = wedge_keep(sector[owner[u % 24]], spec[u].wedge, threads, MARK);
                });
            }
            for (auto &w : workers) w.join();
            workers.clear();
            if (!getenv("ASORA_GEOMETRY_NO_SHARING")) {
                std::vector<char> taken(units, 0);
                for (int u = 0; u < units; ++u) {
                    if (owner[u] != u || taken[u]) continue;
                    std::vector<int> fam;
                    for (int v = u; v < units; ++v)
                        if (owner[v] == v && !taken[v] && spec[v].face == spec[u].face && spec[v].wedge == spec[u].wedge) { fam.push_back(v); taken[v] = 1; }
                    if (fam.size() < 2) continue;
                    int m = 1 << 28;
                    for (int v : fam) m = std::min(m, std::min(spec[v].ext[0], std::min(spec[v].ext[1], spec[v].ext[2])));
                    bool ok = m >= 1;
                    for (int v : fam) ok = ok && (int)keep[v].size() >= m;
                    if (!ok) continue;
                    for (int si = 0; si < m; ++si) {
                        std::vector<char> &first = keep[fam[0]][(size_t)si];
                        for (size_t q = 1; q < fam.size(); ++q) for (size_t c = 0; c < first.size(); ++c) first[c] |= keep[fam[q]][(size_t)si][c];
                        for (size_t q = 1; q < fam.size(); ++q) keep[fam[q]][(size_t)si] = first;
                    }
                }
            }
            for (int u = 0; u < units; ++u) {
                if (owner[u] != u) continue;
                workers.emplace_back([&hg, &keep, &sector, &spec, &owner, u, threads]() {
                    hg[u] = restrict_to_wedge(sector[owner[u % 24]], spec[u].wedge, threads, MARK, keep[u]);
                });
            }
            for (auto &w : workers) w.join();
        }
    }
    for (int v = 0; v < tables; ++v) {
        if (owner[v % units] != v % units) continue;
        if (hg[v].inconsistent)
            return fail(11, "raytrace geometry: a cell of a unit reads a corner outside the unit (internal error)");
        // the kernel compiles the shell-1 factors into a unit's first three steps only
        for (size_t q = 3 * (size_t)threads; q < hg[v].cellA.size(); ++q) {
            const uint4 &ca = hg[v].cellA[q];
            if ((ca.y & CELL_VALID) && std::max({ca.x & 1023u, (ca.x >> 10) & 1023u, (ca.x >> 20) & 1023u}) == 1u)
                return fail(11, "raytrace geometry: a cell of shell 1 lies beyond the first three steps (internal error)");
        }
        Smax = std::max(Smax, hg[v].S);
        max_cells = std::max(max_cells, hg[v].max_cells);
    }
    const double t_built = now_s();
    // zero-slot marker -> max_cells (the slot that holds 0.0), then upload
    for (int v = 0; v < tables; ++v) {
        if (owner[v % units] != v % units) continue;
        HostGeom &h = hg[v];
        for (auto &nb : h.cellB) {
            if (nb.x == MARK) nb.x = max_cells;
            if (nb.y == MARK) nb.y = max_cells;
            if (nb.z == MARK) nb.z = max_cells;
            if (nb.w == MARK) nb.w = max_cells;
        }
        OctGeomDev d = {};
        d.nsteps = h.nsteps;
        if (int rc = upload(h.cellA, d.cellA, st.geom_owned)) return rc;
        if (int rc = upload(h.cellB, d.cellB, st.geom_owned)) return rc;
        st.geom_bytes += 2 * h.cellA.size() * sizeof(uint4);
        od[v] = d;
        if (h.on_sphere)         // where the on-sphere cells of this table live on the device (see patch_sphere_cells)
            for (size_t e = 0; e < h.cellA.size(); ++e)
                if (h.cellA[e].y & CELL_SPHERE) {
                    const uint32_t x = h.cellA[e].x;
                    st.geom_sphere.push_back({reinterpret_cast<uint32_t *>(const_cast<uint4 *>(d.cellA) + e) + 1,
                                              h.cellA[e].y & ~CELL_RATE, (int)(x & 1023), (int)((x >> 10) & 1023), (int)((x >> 20) & 1023)});
                }
    }
    for (int v = 0; v < tables; ++v) { od[v] = od[(v / units) * units + owner[v % units]]; od[v].info = info[v % units]; }
    if (geom_timing) {
        size_t entries = 0;
        for (int v = 0; v < tables; ++v) if (owner[v % units] == v % units) entries += hg[v].cellA.size();
        fprintf(stderr, "asora geometry: %d units, %.1f M entries (%.0f MB); sectors %.3f s, wedges/units %.3f s, markers + upload %.3f s\n",
                units, entries * 1e-6, entries * 32e-6, t_sectors - t_begin, t_built - t_sectors, now_s() - t_built);
    }

    for (int u = 0; u < units && u < 12; ++u) steps_after[u] = hg[owner[u]].step_after_shell;
    }   // host builder

    if (int rc = ensure_logtab(st)) return rc;
    const double2 *ltd = st.logtab_dev;

    for (int o = 0; o < tables; ++o) st.geom_host[o] = od[o];
    st.geom_units = units;
    st.geom_aligned = aligned;
    st.geom_N = N; st.geom_R = p.R; st.geom_dr = p.dr; st.geom_S = Smax; st.geom_max_cells = (int)max_cells;
    st.geom_threads = threads;
    st.geom_subbox = sbg ? 1 : 0; st.geom_ext_r = ext_pos; st.geom_ext_l = ext_neg; st.geom_boxsize = boxsize;
    for (int u = 0; u < units && u < 12; ++u) st.geom_step_after_shell[u] = steps_after[u];
    st.geom_on_host = on_host;
    st.geom_valid = true;
    for (int o = 0; o < tables; ++o) p.geom[o] = od[o];
    p.units = units;
    p.aligned = aligned ? 1 : 0;
    p.logtab = ltd; p.S = Smax; p.max_cells = (int)max_cells;
    return 0;
}

} // namespace asora

extern "C" size_t asora_debug_geometry_bytes(void) { return asora::state().geom_valid ? asora::state().geom_bytes : 0; }

extern "C" int asora_debug_geometry_table(int table, uint32_t *words, size_t capacity_entries, size_t *entries, int *nsteps, int *ntables,
                                          int *shells, int *max_cells, int *threads)
{
    using namespace asora;
    clear_error();
    State &st = state();
    if (!st.geom_valid) return fail(4, "debug_geometry_table: no geometry tables (no raytrace yet)");
    const int nt = st.geom_units * (st.geom_aligned ? 8 : 1);
    if (ntables) *ntables = nt;
    if (shells) *shells = st.geom_S;
    if (max_cells) *max_cells = st.geom_max_cells;
    if (threads) *threads = st.geom_threads;
    if (table < 0) return 0;
    if (table >= nt) return fail(3, "debug_geometry_table: no such table");
    const OctGeomDev &g = st.geom_host[table];
    const size_t n = ((size_t)g.nsteps + 4) * (size_t)st.geom_threads;
    if (entries) *entries = n;
    if (nsteps) *nsteps = g.nsteps;
    if (!words) return 0;
    if (capacity_entries < n) return fail(3, "debug_geometry_table: buffer too small");
    ASORA_HIP_TRY(hipStreamSynchronize(st.stream));
    std::vector<uint4> a(n), b(n);
    // (the steps a table reads through another one -- shared inner shells -- come from there)
    const size_t inner = std::min(n, (size_t)(g.inner >> 8) * (size_t)st.geom_threads);
    const OctGeomDev &first = st.geom_host[g.inner & 255];
    if (inner) {
        ASORA_HIP_TRY(hipMemcpy(a.data(), first.cellA, inner * sizeof(uint4), hipMemcpyDeviceToHost));
        ASORA_HIP_TRY(hipMemcpy(b.data(), first.cellB, inner * sizeof(uint4), hipMemcpyDeviceToHost));
    }
    ASORA_HIP_TRY(hipMemcpy(a.data() + inner, g.cellA + inner, (n - inner) * sizeof(uint4), hipMemcpyDeviceToHost));
    ASORA_HIP_TRY(hipMemcpy(b.data() + inner, g.cellB + inner, (n - inner) * sizeof(uint4), hipMemcpyDeviceToHost));
    for (size_t q = 0; q < n; ++q) {
        std::memcpy(words + 8 * q, &a[q], 16);
        std::memcpy(words + 8 * q + 4, &b[q], 16);
    }
    return 0;
}
