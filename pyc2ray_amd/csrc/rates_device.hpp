// Device functions shared by the raytracing kernels (raytrace.hip: ASORA path, subbox.hip: Fortran-path
// semantics): the table-based log2 and the photo_lookuptable of src/asora/rates.cu:70-83.
#pragma once
#include "asora_internal.hpp"

namespace asora {

constexpr double FOURPI = 12.566370614359172463991853874177;   // raytracing.cu:12
#ifndef ASORA_LOG_TABLE_BITS
#define ASORA_LOG_TABLE_BITS 8
#endif
constexpr int LOG_TABLE_BITS = ASORA_LOG_TABLE_BITS;
constexpr int LOG_TABLE_SIZE = 1 << LOG_TABLE_BITS;

// ---------------------------------------------------------------------------------------------
// Rates (src/asora/rates.cu)
// ---------------------------------------------------------------------------------------------

// log2 of a positive normal double: exponent + table (2^LOG_TABLE_BITS intervals of the mantissa: 1/c and
// log2 c at the interval centres, staged in LDS) + series in r = m/c - 1; it replaces log10 in the table
// lookup because two of them per cell dominated the instruction count.
//   7 bits (rounds 1-3a): |r| < 2^-8, six terms, truncation 3e-18;
//   8 bits (4 KiB of LDS): |r| < 2^-9, FOUR terms; the first one dropped is r^5/(5 ln 2) < 8.2e-15, against an ulp of
//   3.5e-15 ... 7e-15 of log2(tau) itself for tau outside [2^-16, 2^16] and of 2.3e-13 ... 3.6e-12 of the table index
//   k0 + k1 log2(tau) (k0 = 1 - minlogtau/dlogtau = 1668 ... 16668) that the logarithm is formed for: below the rounding
//   of the index, like libm's log10 in the reference.
// ASORA_FREXP_LOG = 1: mantissa and exponent through v_frexp_mant_f64 / v_frexp_exp_i32_f64 (x = m * 2^e, m in [0.5, 1)) instead of
// shifts and masks on the bit pattern (4 instead of 7 integer instructions per logarithm); the table then holds {2/c, log2(c) - 1}
// for the same interval centres c in [1, 2) (ensure_logtab), so that r = m * (2/c) - 1 and log2 x = e + (log2 c - 1) + log2(1 + r).
#ifndef ASORA_FREXP_LOG
#define ASORA_FREXP_LOG 1
#endif
__device__ __forceinline__ double log2_pos(double x, const double2 *__restrict__ logtab, int diag_linear_index = 0)
{
#if ASORA_FREXP_LOG
    const double m = __builtin_amdgcn_frexp_mant(x);
    const int e = __builtin_amdgcn_frexp_exp(x);
    int idx = (int)(__double_as_longlong(m) >> (52 - LOG_TABLE_BITS)) & (LOG_TABLE_SIZE - 1);
#else
    const long long bits = __double_as_longlong(x);
    const int e = (int)(bits >> 52) - 1023;
    int idx = (int)(bits >> (52 - LOG_TABLE_BITS)) & (LOG_TABLE_SIZE - 1);
    const double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
#endif
#ifdef ASORA_ENABLE_ABLATION
    if (diag_linear_index) idx = threadIdx.x & (LOG_TABLE_SIZE - 1);   // diagnostic: conflict-free table reads (wrong values)
#endif
    const double2 t = logtab[idx];                 // {1/c, log2 c}  (ASORA_FREXP_LOG: {2/c, log2 c - 1})
    const double r = fma(m, t.x, -1.0);
    // log2(1+r) = r/ln2 * (1 - r/2 + r^2/3 - r^3/4 + r^4/5 - r^5/6)
    const double C1 = 1.4426950408889634074, C2 = -0.72134752044448170368, C3 = 0.48089834696298780245,
                 C4 = -0.36067376022224085184, C5 = 0.28853900817779268147, C6 = -0.24044917348149390123;
#if ASORA_LOG_TABLE_BITS >= 8
    (void)C5; (void)C6;
    const double p = r * fma(r, fma(r, fma(r, C4, C3), C2), C1);
#else
    const double p = r * fma(r, fma(r, fma(r, fma(r, fma(r, C6, C5), C4), C3), C2), C1);
#endif
    return (double)e + (t.y + p);
}

// photo_lookuptable, rates.cu:70-83 (== photorates.f90:130-147), in two halves so that the two
// dependent table loads can be in flight while other work is done.  The reference forms
// 1 + (log10(tau) - minlogtau)/dlogtau; here that is one fused multiply-add on log2(tau) with
// k1 = log10(2)/dlogtau, k0 = 1 - minlogtau/dlogtau.  Indices are clamped to the last table
// element (the reference reads one past the end when NumTau == len(table), tau >= 10^maxlogtau).
// Device layout of the rate tables (ASORA_DENSE_TABLES):
//   1 (round 4): each table as it is, T[0 .. len-1] plus one more element T[len] = T[len-1]; a lookup is ONE 16-byte load of
//     {T[i], T[i+1]} from an 8-byte-aligned address and the difference is formed in the kernel (one more v_add_f64).  The entries a
//     wave's 64 lanes need then span half as many cache lines as with
//   0 (rounds 1-3): pairs {T[i], T[i+1] - T[i]} (last pair {T[last], 0}), 16 bytes per entry.
// The lookups are the loop's only divergent accesses (LABNOTES round 4: 14 % of the trace on the quiet benchmark medium, 25 % on a
// field with ionisation fronts).  Same bits either way: the host formed T[i+1] - T[i] with the same IEEE subtraction.
// The four tables of an allocation (thick, thin, heating thick, heating thin) follow each other at table_stride(len) entries.
#ifndef ASORA_DENSE_TABLES
#define ASORA_DENSE_TABLES 1
#endif
constexpr int TABLE_ENTRY_SHIFT = ASORA_DENSE_TABLES ? 3 : 4;          // log2 of the bytes per entry
__host__ __device__ constexpr int table_stride(int table_len) { return ASORA_DENSE_TABLES ? table_len + 1 : table_len; }
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));      // a 16-byte load that may start on any double

struct Lookup { double2 t; double2 h; double residual; };   // h: the heating table at the same index
template <bool HEAT = false, typename Params = RtParams>
__device__ __forceinline__ Lookup lookup_issue(const double2 *__restrict__ table, double tau, const Params &p,
                                               const double2 *__restrict__ logtab, int offset = 0)
{   // offset: entries to skip in front (the thin table follows the thick one: a per-lane offset instead of a per-lane pointer)
    // (upper clamp: log2_pos is only defined for finite arguments -- for tau = +inf the mantissa/exponent split gives -inf, i.e.
    //  the FIRST table entry, where the reference's log10(inf) -> min(NumTau, .) reads the last one; any finite value beyond
    //  10^maxlogtau is clamped to the last entry by numtau_f below)
#ifdef ASORA_ENABLE_ABLATION
    const double l2 = log2_pos(fmin(fmax(1.0e-20, tau), 1.0e300), logtab, p.ablate & 16);
#else
    const double l2 = log2_pos(fmin(fmax(1.0e-20, tau), 1.0e300), logtab);
#endif
    // numtau_f is clamped to table_len - 1 on the host (lut_index_limit): real_i >= table_len - 1 reads the last entry
    // (slope 0) whatever the residual, as the reference's i0 = i1 = NumTau does -- no integer clamp here
    const double real_i = fmin(p.numtau_f, fmax(0.0, fma(l2, p.lut_k1, p.lut_k0)));
    const int i0 = (int)real_i;
    Lookup L;
    L.residual = __builtin_amdgcn_fract(real_i);          // real_i - (double)i0 for real_i >= 0, one instruction
    unsigned i = (unsigned)(i0 + offset);
#ifdef ASORA_ENABLE_ABLATION
    if (p.ablate & 8) i = 15000u + (threadIdx.x & 3);   // diagnostic: perfectly coalesced lookups
#endif
    // a 32-bit byte offset from the (wave-uniform) table base: one shift, and the load takes base + offset by itself
    const char *base = reinterpret_cast<const char *>(table);
    auto load16 = [](const char *q) -> double2 {
        const double2_a8 v = *reinterpret_cast<const double2_a8 *>(q);
        return double2{v.x, v.y};
    };
    L.t = load16(base + (i << TABLE_ENTRY_SHIFT));
    if (HEAT) L.h = load16(base + ((i + 2u * (unsigned)table_stride(p.table_len)) << TABLE_ENTRY_SHIFT)); else L.h = L.t;
    return L;
}
#if ASORA_DENSE_TABLES
__device__ __forceinline__ double lookup_value(const Lookup &L) { return fma(L.residual, L.t.y - L.t.x, L.t.x); }
__device__ __forceinline__ double lookup_heat(const Lookup &L) { return fma(L.residual, L.h.y - L.h.x, L.h.x); }
#else
__device__ __forceinline__ double lookup_value(const Lookup &L) { return fma(L.residual, L.t.y, L.t.x); }
__device__ __forceinline__ double lookup_heat(const Lookup &L) { return fma(L.residual, L.h.y, L.h.x); }
#endif

// Host: table t (0 thick, 1 thin, 2 heating thick, 3 heating thin) of `len` entries into a buffer of 4 * len double2 (the
// allocation keeps that size in both layouts)
inline void pack_rate_table(double2 *buffer, int t, const double *src, int len)
{
#if ASORA_DENSE_TABLES
    double *d = reinterpret_cast<double *>(buffer) + (size_t)t * table_stride(len);
    for (int i = 0; i < len; ++i) d[i] = src[i];
    d[len] = src[len - 1];
#else
    for (int i = 0; i < len; ++i) {
        buffer[(size_t)t * len + i].x = src[i];
        buffer[(size_t)t * len + i].y = (i + 1 < len) ? src[i + 1] - src[i] : 0.0;
    }
#endif
}
// byte range of the tables [t0, t1) inside such a buffer
inline size_t rate_table_byte_offset(int t, int len) { return (size_t)t * table_stride(len) * ((size_t)1 << TABLE_ENTRY_SHIFT); }

// Products and sums that must NOT be contracted into a fused multiply-add, where the reference's result depends on
// each operation being rounded on its own (device code is compiled with -ffp-contract=fast, and HIP's __dmul_rn /
// __dadd_rn are plain operators that get contracted after inlining).  Instructions emitted under contract(off) carry
// no `contract` flag, so the backend leaves them alone wherever they are inlined.
__device__ __forceinline__ double mul_unfused(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double add_unfused(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}

// x / y for finite y != 0 of ordinary magnitude: hardware reciprocal, Newton on the reciprocal, one correction of the
// quotient -- 6 instructions instead of the 12 of the IEEE sequence (v_div_scale x 2, v_div_fmas, v_div_fixup guard against
// operands near the ends of the exponent range, which column densities, interpolation weights and cell volumes are not).
// ONE Newton step (ASORA_DIV_NEWTON_STEPS; two until the middle of round 3): v_rcp_f64 is good to 2^-24.4 (measured,
// tools/micro/div_accuracy.hip -> profiles/r03_div_accuracy.txt); a step squares that, and the correction q + r (x - y q)
// multiplies the quotient's error by the reciprocal's once more: 2^-73, far below half an ulp.  Over 6.7e7 random operand
// pairs both forms returned the correctly rounded (IEEE) quotient every time.
// y = 0 gives NaN: callers that can meet it handle it themselves (see pref in raytrace.hip).
#ifndef ASORA_DIV_NEWTON_STEPS
#define ASORA_DIV_NEWTON_STEPS 1
#endif
__device__ __forceinline__ double div_newton(double x, double y)
{
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
#if ASORA_DIV_NEWTON_STEPS >= 2
    r = fma(fma(-y, r, 1.0), r, r);
#endif
    const double q = x * r;
    return fma(fma(-y, q, x), r, q);
}

__device__ __forceinline__ int wrap_once(int x, int N)
{
    return x < 0 ? x + N : (x >= N ? x - N : x);
}

} // namespace asora
