// local_math.hpp -- per-element arithmetic of the ADMM local step, written for
// one GPU lane per element with everything in registers (no arrays indexed at
// run time, every loop over matrix entries unrolled at compile time).
//
// Built with -ffp-contract=off: the operation order below is the order the
// reference's Eigen/cppoptlib expressions evaluate in on x86-64 SSE2 (no FMA),
// and log() / exp() are glibc's algorithms (admm_log, admm_exp below), so every
// local step -- the Neo-Hookean and Fung proxes included -- is bit-identical to the reference.
//
// Reference (under /root/reference):
//   CORE = deps/admm-elastic-sca/src/system
//   OPT  = deps/admm-elastic-sca/deps/cppoptlib/include/cppoptlib
//   EIG  = deps/admm-elastic-sca/deps/Eigen3/Eigen/src
#pragma once

#include <float.h>
#include <math.h>

#include "log_glibc_data.hpp"

#if defined(__HIPCC__)
#define ADMM_HD __host__ __device__ __forceinline__
#else
#define ADMM_HD inline
#endif

#ifndef ADMM_REUSE_GRAD
#define ADMM_REUSE_GRAD 1   // reuse bitwise-identical gradient evaluations (see mt_linesearch)
#endif

namespace admm_dev {

// ---- wave timeline of the tet kernel (tools/probe/tet_timeline.py; build flag -DADMM_TET_TIMELINE, never on in the shipped library):
// the first lane of every wave stamps the 100 MHz real-time counter at its start and at its end, and the wave's largest per-lane
// count of line-search evaluations and L-BFGS iterations -- four words per one-wave block of the launch
#if defined(ADMM_TET_TIMELINE) && defined(__HIPCC__)
__device__ unsigned long long *g_tet_wave_t;
#endif
#if defined(ADMM_TET_TIMELINE) && defined(__HIP_DEVICE_COMPILE__)
#define ADMM_NFEV_COUNT 1
__device__ __forceinline__ unsigned long long wave_max_u(int v) {
    const unsigned long long act = __ballot(1);      // (lanes beyond the batch / the block's tets have left: their registers are not read)
    for (int o = 32; o; o >>= 1) { const int other = __shfl_xor(v, o); if (((act >> ((threadIdx.x & 63) ^ o)) & 1ull) && other > v) v = other; }
    return (unsigned long long)v;
}
#define ADMM_TET_STAMP(w, slot, val) do { if (admm_dev::g_tet_wave_t && (threadIdx.x & 63) == 0) admm_dev::g_tet_wave_t[4 * (size_t)(w) + (slot)] = (val); } while (0)
#define ADMM_TET_STAMP_MAX(w, slot, v) do { const unsigned long long m_ = admm_dev::wave_max_u(v); ADMM_TET_STAMP(w, slot, m_); } while (0)
#else
#define ADMM_NFEV_COUNT 0
#define ADMM_TET_STAMP(w, slot, val)
#define ADMM_TET_STAMP_MAX(w, slot, v)
#endif

// ---- phase attribution of the tet kernel (tools/tet_phase_profile.py; build flag -DADMM_TET_PROFILE, never on in the
// shipped library): s_memtime deltas accumulated by lane 0 of every wave, per-lane loop counts as (sum, 64 x wave maximum)
#if defined(ADMM_TET_PROFILE) && defined(__HIPCC__)
__device__ unsigned long long g_tet_prof[128];  // [0..31] phase ticks / loop counts, [32..63] histogram of line-search evaluations per tet, [64..95] of the wave maxima,
                                                // [96..127] how many times a WAVE executed each code region (ADMM_PROF_REGION; tools/audit/tet_inst_by_region.py)
__device__ float *g_tet_trace;                  // per tet of the current launch: [max |gradient| at the warm start, line-search evaluations] (tools/probe/ls_predict_gpu.py)
#endif
#if defined(ADMM_TET_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned long long prof_now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ void prof_time(int idx, unsigned long long &t) {
    const unsigned long long n = prof_now();
    if (threadIdx.x == 0) atomicAdd(&g_tet_prof[idx], n - t);
    t = prof_now();
}
__device__ __forceinline__ void prof_count(int idx, int v) {
    int sum = v, mx = v;
    for (int o = 32; o; o >>= 1) { sum += __shfl_xor(sum, o); const int other = __shfl_xor(mx, o); mx = other > mx ? other : mx; }
    if (threadIdx.x == 0) { atomicAdd(&g_tet_prof[idx], (unsigned long long)sum); atomicAdd(&g_tet_prof[idx + 1], (unsigned long long)(64 * mx)); }
}
#define ADMM_PROF_T0 unsigned long long prof_t = admm_dev::prof_now();
#define ADMM_PROF_TIME(i) admm_dev::prof_time(i, prof_t)
__device__ __forceinline__ void prof_hist(int v) {
    int mx = v;
    for (int o = 32; o; o >>= 1) { const int other = __shfl_xor(mx, o); mx = other > mx ? other : mx; }
    atomicAdd(&g_tet_prof[32 + (v < 31 ? v : 31)], 1ull);
    if (threadIdx.x == 0) atomicAdd(&g_tet_prof[64 + (mx < 31 ? mx : 31)], 1ull);
}
#define ADMM_PROF_COUNT(i, v) admm_dev::prof_count(i, v)
#define ADMM_PROF_HIST(v) admm_dev::prof_hist(v)
// one count per WAVE that reaches this point with any lane active (a wave issues a region's instructions once, whatever its lane count)
#define ADMM_PROF_REGION(i) do { const unsigned long long a_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)a_) - 1) atomicAdd(&admm_dev::g_tet_prof[96 + (i)], 1ull); } while (0)
#define ADMM_PROF_ON 1
#else
#define ADMM_PROF_T0
#define ADMM_PROF_TIME(i)
#define ADMM_PROF_COUNT(i, v)
#define ADMM_PROF_HIST(v)
#define ADMM_PROF_REGION(i)
#define ADMM_PROF_ON 0
#endif

// libstdc++ std::min / std::max (second argument wins only on strict compare;
// this fixes what happens with NaNs exactly like the reference build)
ADMM_HD double smin(double a, double b) { return (b < a) ? b : a; }
ADMM_HD double smax(double a, double b) { return (a < b) ? b : a; }
// x / |x| without the divide: IEEE division of a finite non-zero x by its own magnitude is exactly +-1
// (0/0, inf/inf and NaN give NaN, as the divide would)
ADMM_HD double unit_sign(double x) { return (x != 0.0 && fabs(x) <= DBL_MAX) ? copysign(1.0, x) : (x - x) / (x - x); }

constexpr double kFltMax = 3.40282346638528859811704183484516925e+38; // (double)FLT_MAX

// A 3x3 matrix as nine named registers, column-major names mRC.
struct Mat3 {
    double m00, m10, m20, m01, m11, m21, m02, m12, m22;
};

template <int R, int C> ADMM_HD double &at(Mat3 &m) {
    if (R == 0 && C == 0) return m.m00; if (R == 1 && C == 0) return m.m10; if (R == 2 && C == 0) return m.m20;
    if (R == 0 && C == 1) return m.m01; if (R == 1 && C == 1) return m.m11; if (R == 2 && C == 1) return m.m21;
    if (R == 0 && C == 2) return m.m02; if (R == 1 && C == 2) return m.m12; return m.m22;
}
template <int R, int C> ADMM_HD double at(const Mat3 &m) { return at<R, C>(const_cast<Mat3 &>(m)); }

ADMM_HD Mat3 identity3() { Mat3 m; m.m00 = 1; m.m10 = 0; m.m20 = 0; m.m01 = 0; m.m11 = 1; m.m21 = 0; m.m02 = 0; m.m12 = 0; m.m22 = 1; return m; }

// Matrix3d::determinant(), EIG/LU/Determinant.h:18-23,61-68
ADMM_HD double det3(const Mat3 &m) {
    double h0 = m.m00 * (m.m11 * m.m22 - m.m12 * m.m21);
    double h1 = m.m01 * (m.m10 * m.m22 - m.m12 * m.m20);
    double h2 = m.m02 * (m.m10 * m.m21 - m.m11 * m.m20);
    return h0 - h1 + h2;
}

// internal::apply_rotation_in_the_plane on one (x, y) pair, EIG/Jacobi/Jacobi.h:302-430
ADMM_HD void rot2(double &x, double &y, double c, double s) {
    double xi = x, yi = y;
    x = c * xi + s * yi;
    y = -s * xi + c * yi;
}
// rows P,Q of m (applyOnTheLeft)
template <int P, int Q> ADMM_HD void rot_rows(Mat3 &m, double c, double s) {
    if (c == 1.0 && s == 0.0) return;
    rot2(at<P, 0>(m), at<Q, 0>(m), c, s);
    rot2(at<P, 1>(m), at<Q, 1>(m), c, s);
    rot2(at<P, 2>(m), at<Q, 2>(m), c, s);
}
// columns P,Q of m (applyOnTheRight with the transposed rotation already folded into s)
template <int P, int Q> ADMM_HD void rot_cols(Mat3 &m, double c, double s) {
    if (c == 1.0 && s == 0.0) return;
    rot2(at<0, P>(m), at<0, Q>(m), c, s);
    rot2(at<1, P>(m), at<1, Q>(m), c, s);
    rot2(at<2, P>(m), at<2, Q>(m), c, s);
}

// ---- correctly rounded sqrt and reciprocal for arguments whose RANGE is known ------------------------------------------
// hipcc expands an fp64 sqrt into v_rsq_f64 + two coupled Newton steps (correctly rounded), wrapped in a rescaling of
// arguments below 2^-767 and a pass-through of 0 / +inf; an fp64 division into v_rcp_f64 + two Newton steps + one correction,
// wrapped in v_div_scale (x2), v_div_fmas and v_div_fixup for operands near the ends of the exponent range and for 0 / inf / nan.
// Where the argument provably lies in [1, 4) resp. [1, 2) the wrappers select nothing and the unscaled core produces the
// very same bits: the Jacobi rotation's sqrt(1 + (q/p)^2), sqrt(tau^2 + 1), sqrt(t^2 + 1) and 1 / sqrt(t^2 + 1)
// (EIG/Jacobi/Jacobi.h:80-110, EIG/Core/MathFunctions.h:284-302) drop 25 of their ~295 instructions per rotation -- the tet kernels
// sit on the VALU issue roof (profiles/r04/valu_roof.txt), so instructions are time.  A NaN argument comes out as NaN either way.
// Host builds (tests/host_math_shim.cpp) keep libm's sqrt and the plain division: both are correctly rounded, i.e. the same bits.
ADMM_HD double sqrt_core(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
#else
    return sqrt(x);
#endif
}
ADMM_HD double sqrt_in_1_4(double x) { return sqrt_core(x); }                                  // x in [1, 4) (or NaN)
ADMM_HD double sqrt_ge_1(double x) {                                                           // x in [1, +inf] (or NaN)
#if defined(__HIP_DEVICE_COMPILE__)
    const double g = sqrt_core(x);
    return x == __builtin_inf() ? x : g;
#else
    return sqrt(x);
#endif
}
ADMM_HD double rcp_in_1_2(double b) {                                                          // 1.0 / b for b in [1, 2) (or NaN)
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double r = __builtin_fma(-b, y, 1.0);      // (the quotient's first estimate 1.0 * y is y itself)
    return __builtin_fma(r, y, y);
#else
    return 1.0 / b;
#endif
}

// numext::hypot, EIG/Core/MathFunctions.h:284-302
ADMM_HD double eig_hypot(double x, double y) {
    double ax = fabs(x), ay = fabs(y);
    double p = smax(ax, ay);
    if (p == 0.0) return 0.0;
    double q = smin(ax, ay);
    double qp = q / p;
    return p * sqrt_in_1_4(1.0 + qp * qp);      // q <= p: the argument lies in [1, 2]
}

// One (p,q) step of the two-sided Jacobi sweep: threshold test,
// real_2x2_jacobi_svd (EIG/SVD/JacobiSVD.h:415-443) with
// JacobiRotation::makeJacobi (EIG/Jacobi/Jacobi.h:80-110), and the four
// rotation applications of JacobiSVD::compute (JacobiSVD.h:873-897).
template <int P, int Q> ADMM_HD bool jacobi_pq(Mat3 &W, Mat3 &U, Mat3 &V) {
    const double precision = 2.0 * DBL_EPSILON;
    const double considerAsZero = 2.0 * 4.9406564584124654e-324;
    double wpp = at<P, P>(W), wqq = at<Q, Q>(W), wpq = at<P, Q>(W), wqp = at<Q, P>(W);
    double threshold = smax(considerAsZero, precision * smax(fabs(wpp), fabs(wqq)));
    if (!(fabs(wpq) > threshold || fabs(wqp) > threshold)) return false;
    ADMM_PROF_REGION(2);      // a Jacobi rotation (any pair)
    double m00 = wpp, m01 = wpq, m10 = wqp, m11 = wqq;
    double c1, s1;
    double t = m00 + m11, d = m10 - m01;
    if (t == 0.0) { c1 = 0.0; s1 = d > 0.0 ? 1.0 : -1.0; }
    else {
        double t2d2 = eig_hypot(t, d);
        c1 = fabs(t) / t2d2;
        s1 = d / t2d2;
        if (t < 0.0) s1 = -s1;
    }
    if (!(c1 == 1.0 && s1 == 0.0)) { rot2(m00, m10, c1, s1); rot2(m01, m11, c1, s1); }
    double rc, rs;
    if (m01 == 0.0) { rc = 1.0; rs = 0.0; }
    else {
        double tau = (m00 - m11) / (2.0 * fabs(m01));
        double w = sqrt_ge_1(tau * tau + 1.0);
        double tt = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);      // |tau +- w| >= 1: |tt| <= 1
        double sign_t = tt > 0.0 ? 1.0 : -1.0;
        double n = rcp_in_1_2(sqrt_in_1_4(tt * tt + 1.0));                // the argument lies in [1, 2], its root in [1, 1.42]
        rs = -sign_t * unit_sign(m01) * fabs(tt) * n;
        rc = n;
    }
    double os = -rs;
    double lc = c1 * rc - s1 * os;
    double ls = c1 * os + s1 * rc;
    rot_rows<P, Q>(W, lc, ls);
    rot_cols<P, Q>(U, lc, ls);
    rot_cols<P, Q>(W, rc, -rs);
    rot_cols<P, Q>(V, rc, -rs);
    return true;
}

template <int A, int B> ADMM_HD void swap_cols(Mat3 &m) {
    double t;
    t = at<0, A>(m); at<0, A>(m) = at<0, B>(m); at<0, B>(m) = t;
    t = at<1, A>(m); at<1, A>(m) = at<1, B>(m); at<1, B>(m) = t;
    t = at<2, A>(m); at<2, A>(m) = at<2, B>(m); at<2, B>(m) = t;
}
template <int C> ADMM_HD void scale_col(Mat3 &m, double f) { at<0, C>(m) *= f; at<1, C>(m) *= f; at<2, C>(m) *= f; }

// Eigen::JacobiSVD<Matrix3d>(F, ComputeFullU|ComputeFullV), EIG/SVD/JacobiSVD.h:824-933
ADMM_HD void svd3(const Mat3 &F, Mat3 &U, double &s0, double &s1, double &s2, Mat3 &V) {
    double scale = fabs(F.m00);
    scale = smax(scale, fabs(F.m10)); scale = smax(scale, fabs(F.m20));
    scale = smax(scale, fabs(F.m01)); scale = smax(scale, fabs(F.m11)); scale = smax(scale, fabs(F.m21));
    scale = smax(scale, fabs(F.m02)); scale = smax(scale, fabs(F.m12)); scale = smax(scale, fabs(F.m22));
    if (scale == 0.0) scale = 1.0;
    Mat3 W;
    W.m00 = F.m00 / scale; W.m10 = F.m10 / scale; W.m20 = F.m20 / scale;
    W.m01 = F.m01 / scale; W.m11 = F.m11 / scale; W.m21 = F.m21 / scale;
    W.m02 = F.m02 / scale; W.m12 = F.m12 / scale; W.m22 = F.m22 / scale;
    U = identity3(); V = identity3();
    bool finished = false;
#if ADMM_PROF_ON
    int prof_sweeps = 0, prof_rot = 0;
#endif
    while (!finished) {
        ADMM_PROF_REGION(1);      // a Jacobi sweep
        bool a = jacobi_pq<1, 0>(W, U, V);
        bool b = jacobi_pq<2, 0>(W, U, V);
        bool c = jacobi_pq<2, 1>(W, U, V);
        finished = !(a || b || c);
#if ADMM_PROF_ON
        prof_sweeps++; prof_rot += (int)a + (int)b + (int)c;
#endif
    }
    ADMM_PROF_COUNT(8, prof_sweeps); ADMM_PROF_COUNT(10, prof_rot);
    double a0 = fabs(W.m00), a1 = fabs(W.m11), a2 = fabs(W.m22);
    if (a0 != 0.0) scale_col<0>(U, unit_sign(W.m00));
    if (a1 != 0.0) scale_col<1>(U, unit_sign(W.m11));
    if (a2 != 0.0) scale_col<2>(U, unit_sign(W.m22));
    s0 = a0; s1 = a1; s2 = a2;
    // descending sort, first maximum wins (EIG/SVD/JacobiSVD.h:910-926); a zero
    // maximum ends the loop
    {
        int pos = 0; double mx = s0;
        if (s1 > mx) { mx = s1; pos = 1; }
        if (s2 > mx) { mx = s2; pos = 2; }
        if (mx != 0.0) {
            if (pos == 1) { double t = s0; s0 = s1; s1 = t; swap_cols<0, 1>(U); swap_cols<0, 1>(V); }
            else if (pos == 2) { double t = s0; s0 = s2; s2 = t; swap_cols<0, 2>(U); swap_cols<0, 2>(V); }
            if (s2 > s1) { double t = s1; s1 = s2; s2 = t; swap_cols<1, 2>(U); swap_cols<1, 2>(V); }
        }
    }
    s0 *= scale; s1 *= scale; s2 *= scale;
}

// helper::oriented_svd, CORE/TetForce.cpp:80-102 (Vt returned as V with the
// sign fix applied to V's column 2, i.e. Vt's row 2)
ADMM_HD void oriented_svd(const Mat3 &F, double &s0, double &s1, double &s2, Mat3 &U, Mat3 &V) {
    svd3(F, U, s0, s1, s2, V);
    if (det3(U) < 0.0) { U.m02 = -U.m02; U.m12 = -U.m12; U.m22 = -U.m22; s2 *= -1.0; }
    // Vt.determinant() on the transposed matrix: same formula with indices swapped
    Mat3 Vt; Vt.m00 = V.m00; Vt.m01 = V.m10; Vt.m02 = V.m20; Vt.m10 = V.m01; Vt.m11 = V.m11; Vt.m12 = V.m21; Vt.m20 = V.m02; Vt.m21 = V.m12; Vt.m22 = V.m22;
    if (det3(Vt) < 0.0) { V.m02 = -V.m02; V.m12 = -V.m12; V.m22 = -V.m22; s2 *= -1.0; }
}

// U * diag(s) * V^T with Eigen's coefficient order ((p0+p1)+p2), p_k = (U(i,k)*s_k)*V(j,k)
ADMM_HD Mat3 recompose(const Mat3 &U, double s0, double s1, double s2, const Mat3 &V) {
    Mat3 r;
#define ADMM_RC(i, j) ((at<i, 0>(U) * s0) * at<j, 0>(V) + (at<i, 1>(U) * s1) * at<j, 1>(V)) + (at<i, 2>(U) * s2) * at<j, 2>(V)
    r.m00 = ADMM_RC(0, 0); r.m10 = ADMM_RC(1, 0); r.m20 = ADMM_RC(2, 0);
    r.m01 = ADMM_RC(0, 1); r.m11 = ADMM_RC(1, 1); r.m21 = ADMM_RC(2, 1);
    r.m02 = ADMM_RC(0, 2); r.m12 = ADMM_RC(1, 2); r.m22 = ADMM_RC(2, 2);
#undef ADMM_RC
    return r;
}

struct V3 { double a, b, c; };
ADMM_HD double dotd(const V3 &x, const V3 &y) { return (x.a * y.a + x.b * y.b) + x.c * y.c; } // dynamic-size order
ADMM_HD double absmax(const V3 &x) { double r = fabs(x.a); r = smax(r, fabs(x.b)); r = smax(r, fabs(x.c)); return r; }

// ---- log(): glibc's double-precision algorithm, restated ----------------------------------------------------------------
// NHProx calls libm's log() (CORE/TetForce.cpp:229-262).  OCML's log differs from it in the last bit on part of the
// arguments, which the reference's L-BFGS then amplifies; so the device evaluates glibc's own algorithm (glibc >= 2.28
// sysdeps/ieee754/dbl-64/e_log.c: z = x / 2^k in [0x1.6p-1, 0x1.6p0), 128-entry table of (1/c, log c), r = z/c - 1,
// degree-5 polynomial; a degree-11 polynomial in r = x - 1 for 1 - 2^-4 <= x < 1 + 0x1.09p-4) with glibc's constants
// (log_glibc_data.hpp) in the operation order of the build x86-64 hosts with FMA run (ifunc variant __log_fma: the
// fused operations below are the ones GCC contracted there, read off its object code; every other a*b+c stays two
// roundings).  tests/test_host_math.py::test_log: bit-identical with libm over 4M arguments incl. all special cases.
// About half the instructions of OCML's log.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const double g_log_tab[256] = ADMM_LOG_TAB;
#else
static const double g_log_tab[256] = ADMM_LOG_TAB;
#endif
ADMM_HD unsigned long long dbl_bits(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return u; }
ADMM_HD double bits_dbl(unsigned long long u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
// the table path on the bits of a positive, finite, NORMAL argument.  32-bit words on purpose: every constant involved has a zero low word,
// so the 64-bit subtractions, the shift by 52 and the int64 -> double conversion of glibc's source are one-word operations here (hipcc
// kept them two-word: 12 more instructions per call, and the kernels sit on the VALU issue roof)
ADMM_HD double admm_log_table(unsigned hi, unsigned lo) {
    const double A[5] = ADMM_LOG_A;
    const unsigned tmp_hi = hi - 0x3fe60000u;                        // tmp = ix - 0x3fe6000000000000
    const int i = (int)((tmp_hi >> 13) & 127u);                      // (tmp >> 45) & 127
    const int k = (int)tmp_hi >> 20;                                 // (int64) tmp >> 52
    const double z = bits_dbl(((unsigned long long)(hi - (tmp_hi & 0xfff00000u)) << 32) | lo);      // ix - (tmp & 0xfff0000000000000)
    const double invc = g_log_tab[2 * i], logc = g_log_tab[2 * i + 1];
    const double kd = (double)k;
    const double r = __builtin_fma(z, invc, -1.0);
    const double w = __builtin_fma(kd, ADMM_LOG_LN2HI, logc);
    const double hi_ = w + r;
    const double lo_ = __builtin_fma(kd, ADMM_LOG_LN2LO, (w - hi_) + r);
    const double r2 = r * r;
    const double q = __builtin_fma(__builtin_fma(r, A[4], A[3]), r2, __builtin_fma(r, A[2], A[1]));
    return __builtin_fma(r * r2, q, __builtin_fma(r2, A[0], lo_)) + hi_;
}
ADMM_HD double admm_log(double x) {
    const double B[11] = ADMM_LOG_B;
    const unsigned long long ix = dbl_bits(x);
    const unsigned hi = (unsigned)(ix >> 32), lo = (unsigned)ix;
    double res;
    if (hi - 0x3fee0000u < 0x00030900u) {                            // 1 - 0x1p-4 <= x < 1 + 0x1.09p-4  (ix - 0x3fee000000000000 < 0x0003090000000000)
        ADMM_PROF_REGION(11);
        const double r = x - 1.0;
        const double r2 = r * r, r3 = r * r2;
        const double p1 = __builtin_fma(r2, B[3], __builtin_fma(r, B[2], B[1]));
        const double p4 = __builtin_fma(r2, B[6], __builtin_fma(r, B[5], B[4]));
        double p7 = __builtin_fma(r2, B[9], __builtin_fma(r, B[8], B[7]));
        p7 = __builtin_fma(r3, B[10], p7);
        const double pol = __builtin_fma(__builtin_fma(p7, r3, p4), r3, p1);
        const double t = __builtin_fma(r, 0x1p27, r);              // r + w, w = r * 2^27
        const double rhi = __builtin_fma(-0x1p27, r, t);            // (r + w) - w
        const double rlo = r - rhi;
        const double rh2 = rhi * rhi;
        const double h = __builtin_fma(rh2, B[0], r);
        double l = __builtin_fma(rh2, B[0], r - h);
        l = __builtin_fma(B[0] * rlo, r + rhi, l);
        res = h + __builtin_fma(pol, r3, l);
        if (ix == 0x3ff0000000000000ull) res = 0.0;
    } else {
        ADMM_PROF_REGION(10);
        res = admm_log_table(hi, lo);
        const unsigned top = hi >> 16;
        if (top - 0x0010u >= 0x7ff0u - 0x0010u) {                    // zero, subnormal, negative, inf, nan: rare, and behind a branch a wave without such a lane skips
            if (ix * 2 == 0) res = -__builtin_inf();
            else if (ix == 0x7ff0000000000000ull) res = x;
            else if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) res = __builtin_nan("");
            else {                                                   // subnormal: renormalise, then the same path
                const unsigned long long is = dbl_bits(x * 0x1p52) - (52ull << 52);
                res = admm_log_table((unsigned)(is >> 32), (unsigned)is);
            }
        }
    }
    return res;
}

// ---- exp(): glibc's algorithm, same treatment (FungProx, CORE/TriangleForce.cpp:171-223, calls libm's exp) --------------
// glibc sysdeps/ieee754/dbl-64/e_exp.c: k/N = round(x N / ln 2), r = x - k ln2/N, 2^(k/N) from a 128-entry table, degree-5
// polynomial; results near the over/underflow thresholds go through its specialcase() scaling.  Operation order of the
// x86-64 FMA build (__exp_fma).  tests/test_host_math.py::test_exp: bit-identical with libm, all ranges.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const unsigned long long g_exp_tab[256] = ADMM_EXP_TAB;
#else
static const unsigned long long g_exp_tab[256] = ADMM_EXP_TAB;
#endif
ADMM_HD double admm_exp(double x) {
    const double C[4] = ADMM_EXP_C;
    const unsigned long long ix = dbl_bits(x);
    unsigned abstop = (unsigned)(ix >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {                                   // |x| < 2^-54 or |x| >= 512 or not finite
        if ((int)(abstop - 0x3c9u) < 0) return 1.0 + x;
        if (abstop > 0x408u) {                                       // |x| >= 1024, inf, nan
            if (ix == 0xfff0000000000000ull) return 0.0;
            if (abstop == 0x7ffu) return 1.0 + x;
            return (ix >> 63) ? 0.0 : __builtin_inf();               // (__math_uflow / __math_oflow)
        }
        abstop = 0;                                                  // 512 <= |x| < 1024: the result may need the careful scaling
    }
    double kd = __builtin_fma(x, ADMM_EXP_INVLN2N, ADMM_EXP_SHIFT);
    const unsigned long long ki = dbl_bits(kd);
    kd = kd - ADMM_EXP_SHIFT;
    const double r = __builtin_fma(kd, ADMM_EXP_NEGLN2LON, __builtin_fma(kd, ADMM_EXP_NEGLN2HIN, x));
    const int idx = 2 * (int)(ki & 127);
    const double tail = bits_dbl(g_exp_tab[idx]);
    unsigned long long sbits = g_exp_tab[idx + 1] + (ki << 45);
    const double r2 = r * r;
    const double lowp = __builtin_fma(__builtin_fma(r, C[1], C[0]), r2, r + tail);
    const double tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C[3], C[2]), lowp);
    if (abstop == 0) {                                               // e_exp.c specialcase()
        if ((ki & 0x80000000ull) == 0) {                             // k > 0: scale by 2^-1009 first, the product may overflow
            sbits -= 1009ull << 52;
            const double scale = bits_dbl(sbits);
            return 0x1p1009 * __builtin_fma(scale, tmp, scale);
        }
        sbits += 1022ull << 52;                                      // k < 0: the result may be subnormal
        const double scale = bits_dbl(sbits);
        const double st = tmp * scale;
        double y = scale + st;
        if (y < 1.0) {
            double lo = (scale - y) + st;
            const double hi = 1.0 + y;
            lo = ((1.0 - hi) + y) + lo;
            y = (lo + hi) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    const double scale = bits_dbl(sbits);
    return __builtin_fma(scale, tmp, scale);
}

// ---- prox objective: NHProx / StVKProx, CORE/TetForce.cpp:216-297 ----------
template <int TYPE> struct Prox {
    double mu, lambda, k;
    V3 s0;
#if ADMM_PROF_ON || ADMM_NFEV_COUNT
    mutable int prof_nfev = 0;
#endif

    ADMM_HD double value(const V3 &x) const {
        if (x.a < 0.0 || x.b < 0.0 || x.c < 0.0) return kFltMax;
        double da = x.a - s0.a, db = x.b - s0.b, dc = x.c - s0.c;
        if (TYPE == 0) {
            double Sig_det = (x.a * x.b * x.c);
            double I_1 = x.a * x.a + x.b * x.b + x.c * x.c;
            double I_3 = Sig_det * Sig_det;
            double log_I3 = admm_log(I_3);
            double t1 = 0.5 * mu * (I_1 - log_I3 - 3.0);
            double t2 = 0.125 * lambda * log_I3 * log_I3;
            double r = t1 + t2;
            double r2 = (k * 0.5) * ((da * da + db * db) + dc * dc);
            return (1.0 * r + r2);
        } else {
            double ea = 0.5 * (x.a * x.a - 1.0), eb = 0.5 * (x.b * x.b - 1.0), ec = 0.5 * (x.c * x.c - 1.0);
            double tr = (ea + eb) + ec;
            double st_tr2 = tr * tr;
            double dd = ea * ea + (eb * eb + ec * ec);
            double r = (mu * dd + (lambda * 0.5 * st_tr2));
            double r2 = (k * 0.5) * (da * da + (db * db + dc * dc));
            return (r + r2);
        }
    }
    ADMM_HD V3 gradient(const V3 &x) const {
        V3 g;
        if (TYPE == 0) {
            double detSigma = x.a * x.b * x.c;
            if (detSigma <= 0.0) { g.a = g.b = g.c = 1.0 * kFltMax; return g; }
            double ia = 1.0 / x.a, ib = 1.0 / x.b, ic = 1.0 / x.c;
            double ll = lambda * admm_log(detSigma);
            g.a = 1.0 * (mu * (x.a - ia) + ll * ia) + k * (x.a - s0.a);
            g.b = 1.0 * (mu * (x.b - ib) + ll * ib) + k * (x.b - s0.b);
            g.c = 1.0 * (mu * (x.c - ic) + ll * ic) + k * (x.c - s0.c);
        } else {
            double xx = (x.a * x.a + x.b * x.b) + x.c * x.c;
            double c2 = 0.5 * lambda * (xx - 3.0);
            g.a = mu * x.a * (x.a * x.a - 1.0) + c2 * x.a + k * (x.a - s0.a);
            g.b = mu * x.b * (x.b * x.b - 1.0) + c2 * x.b + k * (x.b - s0.b);
            g.c = mu * x.c * (x.c * x.c - 1.0) + c2 * x.c + k * (x.c - s0.c);
        }
        return g;
    }
};

// ---- MoreThuente::cstep, OPT/linesearch/morethuente.h:169-308 ---------------
// The reference's four cases (fp > fx | sgnd < 0 | |dp| < |dx| | else) each evaluate the same cubic-step expression tree on
// different operands.  The lanes of a wavefront sit in different cases (and would execute all four bodies one after the
// other: 24 divisions + 4 square roots per call), so the operands are selected first and the tree is evaluated ONCE
// (7 divisions + 1 square root).  Every lane still performs exactly its own case's operations on its own case's operands
// in the reference's order, so the results are bitwise those of the branching form (tests/test_host_math.py runs this
// code on the host against the reference's cstep through the projection fixtures).
//
// When ALL the lanes of a wavefront that reach the selection sit in the same case (31-43 % of the wave-level calls,
// profiles/r04/tet_phase_profile.txt) the selects buy nothing: mt_cstep_case<C> below is that one case written out -- the
// same operations on the same operands in the same order, minus the selects (and, in case 4 with no bracket anywhere in
// the wave, minus the whole cubic) -- and mt_cstep takes it through a wave-uniform branch.
#ifndef ADMM_CSTEP_UNIFORM
#define ADMM_CSTEP_UNIFORM 1
#endif
template <int C> ADMM_HD void mt_cstep_case(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp,
                                            double fp, double dp, bool &brackt, double stpmin, double stpmax, bool any_brackt) {
    double stpf;
    if (C == 1) {                   // fp > fx: the minimum is bracketed (:184-206)
        const double theta = 3. * (fx - fp) / (stp - stx) + dx + dp;
        const double s = smax(theta, smax(dx, dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        const double gp = gamma - dx;
        const double r = (gp + theta) / ((gp + gamma) + dp);
        const double stpc = stx + r * (stp - stx);
        const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.) * (stp - stx);
        if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2;
        brackt = true;
        sty = stp; fy = fp; dy = dp;
    } else if (C == 2) {            // derivatives of opposite sign: bracketed (:213-234)
        const double theta = 3. * (fx - fp) / (stp - stx) + dx + dp;
        const double s = smax(theta, smax(dx, dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        const double gp = gamma - dp;
        const double r = (gp + theta) / ((gp + gamma) + dx);
        const double stpc = stp + r * (stx - stp);
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = true;
        sty = stx; fy = fx; dy = dx;
        stx = stp; fx = fp; dx = dp;
    } else if (C == 3) {            // same sign, the derivative decreases in magnitude (:242-274)
        const double theta = 3. * (fx - fp) / (stp - stx) + dx + dp;
        const double s = smax(theta, smax(dx, dp));
        double gamma = s * sqrt(smax(0., (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        const double r = ((gamma - dp) + theta) / ((gamma + (dx - dp)) + gamma);
        double stpc = stp + r * (stx - stp);
        if (!((r < 0.0) & (gamma != 0.0))) stpc = (stp > stx) ? stpmax : stpmin;
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        const double ac = fabs(stp - stpc), aq = fabs(stp - stpq);
        stpf = (brackt ? (ac < aq) : (ac > aq)) ? stpc : stpq;
        stx = stp; fx = fp; dx = dp;
    } else {                        // same sign, the derivative does not decrease (:281-291)
        stpf = (stp > stx) ? stpmax : stpmin;
        if (any_brackt) {
            const double theta = 3. * (fp - fy) / (sty - stp) + dy + dp;
            const double s = smax(theta, smax(dy, dp));
            double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            const double gp = gamma - dp;
            const double r = (gp + theta) / ((gp + gamma) + dy);
            const double stpc = stp + r * (sty - stp);
            if (brackt) stpf = stpc;
        }
        stx = stp; fx = fp; dx = dp;
    }
    stpf = smin(stpmax, stpf);
    stpf = smax(stpmin, stpf);
    stp = stpf;
    if ((C == 1 || C == 3) && brackt) {
        if (sty > stx) stp = smin(stx + 0.66 * (sty - stx), stp);
        else stp = smax(stx + 0.66 * (sty - stx), stp);
    }
}

ADMM_HD void mt_cstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp,
                      double fp, double dp, bool &brackt, double stpmin, double stpmax, int &info) {
    info = 0;
    if ((brackt & ((stp <= smin(stx, sty)) | (stp >= smax(stx, sty)))) | (dx * (stp - stx) >= 0.0) | (stpmax < stpmin)) return;
    const double sgnd = dp * unit_sign(dx);
    const bool c1 = fp > fx;
    const bool c2 = !c1 & (sgnd < 0.0);
    const bool c3 = !c1 & !c2 & (fabs(dp) < fabs(dx));
    const bool c4 = !c1 & !c2 & !c3;
    info = c1 ? 1 : (c2 ? 2 : (c3 ? 3 : 4));
    const bool bound = c1 | c3;
#if ADMM_CSTEP_UNIFORM && defined(__HIP_DEVICE_COMPILE__) && !ADMM_PROF_ON
    {
        const unsigned long long act = __ballot(1);
        if (__ballot(c1) == act) { mt_cstep_case<1>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax, true); return; }
        if (__ballot(c2) == act) { mt_cstep_case<2>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax, true); return; }
        if (__ballot(c3) == act) { mt_cstep_case<3>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax, true); return; }
        if (__ballot(c4) == act) { mt_cstep_case<4>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax, __ballot(brackt) != 0ull); return; }
    }
#endif
#if ADMM_PROF_ON && defined(__HIP_DEVICE_COMPILE__)
    {   // how often do all lanes of a wave that reach the step selection sit in ONE of the four cases?  (regions 16..22)
        const unsigned long long act = __ballot(1), b1 = __ballot(c1), b2 = __ballot(c2), b3 = __ballot(c3), b4 = __ballot(c4);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {
            atomicAdd(&g_tet_prof[96 + 16], 1ull);
            if (b1 == act || b2 == act || b3 == act || b4 == act) atomicAdd(&g_tet_prof[96 + 17], 1ull);
            if (b1) atomicAdd(&g_tet_prof[96 + 18], 1ull);
            if (b2) atomicAdd(&g_tet_prof[96 + 19], 1ull);
            if (b3) atomicAdd(&g_tet_prof[96 + 20], 1ull);
            if (b4) atomicAdd(&g_tet_prof[96 + 21], 1ull);
            atomicAdd(&g_tet_prof[96 + 22], (unsigned long long)__popcll(act));
        }
    }
#endif
#ifdef ADMM_CSTEP_STATS     // tests/host_math_shim.cpp: which cases the fixtures reach
    ADMM_CSTEP_STATS[info + (brackt ? 4 : 0)]++;
#endif
    // theta = 3 (f_a - f_b) / (st_b - st_a) + d_a + dp        (:187, :216, :245, :283)
    const double num = c4 ? fp - fy : fx - fp;
    const double den = c4 ? sty - stp : stp - stx;
    const double d1 = c4 ? dy : dx;
    const double theta = 3. * num / den + d1 + dp;
    const double s = smax(theta, smax(d1, dp));
    double rad = (theta / s) * (theta / s) - (d1 / s) * (dp / s);
    if (c3) rad = smax(0., rad);
    double gamma = s * sqrt(rad);
    const bool flip = c1 ? (stp < stx) : (c4 ? (stp > sty) : (stp > stx));
    if (flip) gamma = -gamma;
    const double gp = gamma - (c1 ? dx : dp);
    const double p = gp + theta;
    const double q = c3 ? ((gamma + (dx - dp)) + gamma) : ((gp + gamma) + (c1 ? dp : (c2 ? dx : dy)));
    const double r = p / q;
    double stpc = (c1 ? stx : stp) + r * (c1 ? stp - stx : (c4 ? sty - stp : stx - stp));
    if (c3 & !((r < 0.0) & (gamma != 0.0))) stpc = (stp > stx) ? stpmax : stpmin;
    // quadratic / secant step: case 1 :196, cases 2 and 3 :224, :263
    const double qd = c1 ? ((fx - fp) / (stp - stx) + dx) : (dp - dx);
    const double ratio = (c1 ? dx : dp) / qd;
    const double stpq = c1 ? stx + (ratio / 2.) * (stp - stx) : stp + ratio * (stx - stp);
    double stpf;
    if (c1) {
        if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2;
        brackt = true;
    } else if (c2) {
        if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = true;
    } else if (c3) {
        if (brackt) { if (fabs(stp - stpc) < fabs(stp - stpq)) stpf = stpc; else stpf = stpq; }
        else { if (fabs(stp - stpc) > fabs(stp - stpq)) stpf = stpc; else stpf = stpq; }
    } else {
        if (brackt) stpf = stpc;
        else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
    // the interval update (:293-300) as selects on VALUES with one unconditional store per reference.  Written with branches
    // ("if (fp > fx) { fy = fp; ... } else { ...; fx = fp; ... }") hipcc sinks the two stores of fp into ONE store through a selected
    // POINTER, which keeps fx / fy / dx / dy of the caller in scratch memory: a store -> load round trip through the memory pipe on the
    // dependent chain of every line-search evaluation (15 scratch instructions per evaluation in the round-2 kernel).
    {
        const bool up = fp > fx, neg = sgnd < 0.0;
        const double nsty = up ? stp : (neg ? stx : sty), nfy = up ? fp : (neg ? fx : fy), ndy = up ? dp : (neg ? dx : dy);
        const double nstx = up ? stx : stp, nfx = up ? fx : fp, ndx = up ? dx : dp;
        sty = nsty; fy = nfy; dy = ndy; stx = nstx; fx = nfx; dx = ndx;
    }
    stpf = smin(stpmax, stpf);
    stpf = smax(stpmin, stpf);
    stp = stpf;
    if (brackt & bound) {
        if (sty > stx) stp = smin(stx + 0.66 * (sty - stx), stp);
        else stp = smax(stx + 0.66 * (sty - stx), stp);
    }
}

// ---- MoreThuente::linesearch + cvsrch, morethuente.h:25-167 ------------------
// x: base point, s: direction (= -q).  Returns the step length.
// g_at_x is the gradient at x (the reference re-evaluates it, morethuente.h:31-33:
// same input, same value -- reused here); g_out/evaluated return the gradient at
// the accepted point x + stp*s when the search evaluated it (the caller's next
// gradient(x0) is at bitwise the same point: x0 - rate*q == x + stp*(-q)).
template <class P> ADMM_HD double mt_linesearch(const P &prob, const V3 &x, const V3 &s, double alpha_init, const V3 &g_at_x, V3 &g_out, bool &evaluated) {
    double stp = alpha_init;
    evaluated = false;
    double f = prob.value(x);
    V3 g = ADMM_REUSE_GRAD ? g_at_x : prob.gradient(x);
    int info = 0, infoc = 1;
    const double xtol = 1e-15, ftol = 1e-4, gtol = 1e-2, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4;
    const int maxfev = 20;
    int nfev = 0;
    double dginit = dotd(g, s);
    if (dginit >= 0.0) return stp;
    bool brackt = false, stage1 = true;
    double finit = f, dgtest = ftol * dginit;
    double width = stpmax - stpmin, width1 = 2 * width;
    double stx = 0.0, fx = finit, dgx = dginit, sty = 0.0, fy = finit, dgy = dginit;
    double stmin = 0.0, stmax = 0.0;
#if ADMM_PROF_ON
    V3 prof_px = x;
#endif
    for (;;) {
        if (brackt) { stmin = smin(stx, sty); stmax = smax(stx, sty); }
        else { stmin = stx; stmax = stp + xtrapf * (stp - stx); }
        stp = smax(stp, stpmin);
        stp = smin(stp, stpmax);
        if ((brackt && ((stp <= stmin) | (stp >= stmax))) | (nfev >= maxfev - 1) | (infoc == 0) | (brackt & (stmax - stmin <= xtol * stmax))) stp = stx;
        V3 xn; xn.a = x.a + stp * s.a; xn.b = x.b + stp * s.b; xn.c = x.c + stp * s.c;
        ADMM_PROF_REGION(8);      // a line-search evaluation
#if ADMM_PROF_ON && defined(__HIP_DEVICE_COMPILE__)
        {   // how many evaluations happen at a point that was evaluated just before (the previous trial point or the base point)?
            const bool rep = (xn.a == prof_px.a && xn.b == prof_px.b && xn.c == prof_px.c) || (xn.a == x.a && xn.b == x.b && xn.c == x.c);
            const unsigned long long act = __ballot(1), reps = __ballot(rep);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {
                atomicAdd(&g_tet_prof[18], (unsigned long long)__popcll(act)); atomicAdd(&g_tet_prof[20], (unsigned long long)__popcll(reps));
                atomicAdd(&g_tet_prof[22], 1ull); if (reps == act) atomicAdd(&g_tet_prof[24], 1ull);
            }
            prof_px = xn;
        }
#endif
        f = prob.value(xn);
        g = prob.gradient(xn);
        g_out = g; evaluated = true;
        nfev++;
#if ADMM_PROF_ON || ADMM_NFEV_COUNT
        prob.prof_nfev++;
#endif
        double dg = dotd(g, s);
        double ftest1 = finit + stp * dgtest;
        if ((brackt & ((stp <= stmin) | (stp >= stmax))) | (infoc == 0)) info = 6;
        if ((stp == stpmax) & (f <= ftest1) & (dg <= dgtest)) info = 5;
        if ((stp == stpmin) & ((f > ftest1) | (dg >= dgtest))) info = 4;
        if (nfev >= maxfev) info = 3;
        if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
        if ((f <= ftest1) & (fabs(dg) <= gtol * (-dginit))) info = 1;
        if (info != 0) return stp;
        ADMM_PROF_REGION(9);      // ... that goes on to a step selection (mt_cstep)
        if (stage1 & (f <= ftest1) & (dg >= smin(ftol, gtol) * dginit)) stage1 = false;
        // modified function in stage 1 (:118-138): one cstep call on selected operands (see mt_cstep)
        // (fx, fy, dgx, dgy are modified IN PLACE around the call -- the reference's fxm = fx - stx * dgtest ... fx = fxm + stx * dgtest on
        // the same operands -- instead of going through four copies handed to mt_cstep by reference: hipcc kept those copies in
        // scratch memory, a store -> load round trip through the memory pipe on the dependent chain of every evaluation)
        const bool modified = stage1 & (f <= fx) & (f > ftest1);
        double cf = f, cdg = dg;
        if (modified) {
#ifdef ADMM_CSTEP_STATS
            ADMM_CSTEP_STATS[0]++;
#endif
            cf = f - stp * dgtest;
            fx = fx - stx * dgtest;
            fy = fy - sty * dgtest;
            cdg = dg - dgtest;
            dgx = dgx - dgtest;
            dgy = dgy - dgtest;
        }
        mt_cstep(stx, fx, dgx, sty, fy, dgy, stp, cf, cdg, brackt, stmin, stmax, infoc);
        if (modified) {
            fx = fx + stx * dgtest;
            fy = fy + sty * dgtest;
            dgx = dgx + dgtest;
            dgy = dgy + dgtest;
        }
        if (brackt) {
            if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
            width1 = width;
            width = fabs(sty - stx);
        }
    }
}

// ---- cppoptlib::lbfgssolver<double>::minimize, OPT/solver/lbfgssolver.h:43-144
// M = compile-time history capacity (>= min(maxIter,10)).  The history is a
// run-time indexed private array on purpose: it lives in scratch memory, not in
// VGPRs (it is only touched from the second outer iteration on, and keeping
// 2*3*M + 2*M doubles in registers would halve the kernel's occupancy).
template <int M, class P> ADMM_HD int lbfgs_minimize(const P &prob, V3 &x0, int maxIter, double gradTol, double &init_hess) {
    const int m_ = maxIter < 10 ? maxIter : 10;
    const double eps_g = gradTol, eps_x = 1e-8;
    double hs[M][3], hy[M][3], alpha[M], rho[M];
    V3 grad = prob.gradient(x0);
    double gamma_k = init_hess;
    double alpha_init = smin(1.0, 1.0 / absmax(grad));
    int globIter = 0;
    int maxiter = maxIter;
    double new_hess_guess = 1.0;
    for (int k = 0; k < maxiter; k++) {
        ADMM_PROF_REGION(6);      // an L-BFGS outer iteration
        V3 x_old = x0, grad_old = grad, q = grad;
        globIter++;
        const int iter = m_ < k ? m_ : k;
#pragma unroll 1
        for (int i = iter - 1; i >= 0; --i) {
            ADMM_PROF_REGION(7);      // a history pair in the two-loop recursion (either loop)
            V3 si, yi; si.a = hs[i][0]; si.b = hs[i][1]; si.c = hs[i][2]; yi.a = hy[i][0]; yi.b = hy[i][1]; yi.c = hy[i][2];
            const double r = 1.0 / dotd(si, yi);
            const double al = r * dotd(si, q);
            rho[i] = r; alpha[i] = al;
            q.a = q.a - al * yi.a; q.b = q.b - al * yi.b; q.c = q.c - al * yi.c;
        }
        q.a = gamma_k * q.a; q.b = gamma_k * q.b; q.c = gamma_k * q.c;
#pragma unroll 1
        for (int i = 0; i < iter; ++i) {
            ADMM_PROF_REGION(7);
            V3 si, yi; si.a = hs[i][0]; si.b = hs[i][1]; si.c = hs[i][2]; yi.a = hy[i][0]; yi.b = hy[i][1]; yi.c = hy[i][2];
            const double beta = rho[i] * dotd(q, yi);
            const double ab = alpha[i] - beta;
            q.a = q.a + ab * si.a; q.b = q.b + ab * si.b; q.c = q.c + ab * si.c;
        }
        double dir = dotd(q, grad);
        if (dir < 1e-4) {
            ADMM_PROF_REGION(14);
            q = grad;
            maxiter -= k;
            k = 0;
            alpha_init = smin(1.0, 1.0 / absmax(grad));
        }
        V3 mq; mq.a = -q.a; mq.b = -q.b; mq.c = -q.c;
        V3 g_new; bool have_g;
        const double rate = mt_linesearch(prob, x0, mq, alpha_init, grad, g_new, have_g);
        x0.a = x0.a - rate * q.a; x0.b = x0.b - rate * q.b; x0.c = x0.c - rate * q.c;
        V3 dxx; dxx.a = x_old.a - x0.a; dxx.b = x_old.b - x0.b; dxx.c = x_old.c - x0.c;
        if (dotd(dxx, dxx) < eps_x) break;
        grad = (ADMM_REUSE_GRAD && have_g) ? g_new : prob.gradient(x0);
        double gradNorm = absmax(grad);
        if (gradNorm < eps_g) { new_hess_guess = gamma_k; break; }
        V3 s_temp, y_temp;
        s_temp.a = x0.a - x_old.a; s_temp.b = x0.b - x_old.b; s_temp.c = x0.c - x_old.c;
        y_temp.a = grad.a - grad_old.a; y_temp.b = grad.b - grad_old.b; y_temp.c = grad.c - grad_old.c;
        if (k < m_) {
            hs[k][0] = s_temp.a; hs[k][1] = s_temp.b; hs[k][2] = s_temp.c;
            hy[k][0] = y_temp.a; hy[k][1] = y_temp.b; hy[k][2] = y_temp.c;
        } else {
#pragma unroll 1
            for (int i = 0; i < m_ - 1; ++i) {
                hs[i][0] = hs[i + 1][0]; hs[i][1] = hs[i + 1][1]; hs[i][2] = hs[i + 1][2];
                hy[i][0] = hy[i + 1][0]; hy[i][1] = hy[i + 1][1]; hy[i][2] = hy[i + 1][2];
            }
            hs[m_ - 1][0] = s_temp.a; hs[m_ - 1][1] = s_temp.b; hs[m_ - 1][2] = s_temp.c;
            hy[m_ - 1][0] = y_temp.a; hy[m_ - 1][1] = y_temp.b; hy[m_ - 1][2] = y_temp.c;
        }
        gamma_k = dotd(s_temp, y_temp) / dotd(y_temp, y_temp);
        alpha_init = 1.0;
    }
    init_hess = new_hess_guess;
    return globIter;
}

// ---- HyperElasticTet::project on F = Dx_i + u_i, CORE/TetForce.cpp:320-364 ----
// state: sa,sb,sc = last_prox_result, hess = solver->settings_.init_hess.
// Returns z (= U diag(sigma) V^T) and the L-BFGS iteration count.
struct NoMid { ADMM_HD void operator()() const {} };
// `mid` runs between the minimisation and the recomposition U diag(sigma) V^T: the tracking tet kernel requests its z_prev
// loads there, so that their latency hides under the recomposition instead of stalling the epilogue
template <int TYPE, int M, class Mid = NoMid>
ADMM_HD Mat3 project_hyper(const Mat3 &F, double mu, double lambda, int maxIter, double &sa, double &sb, double &sc, double &hess, int &n_iters, const Mid &mid = Mid()) {
    double s0, s1, s2; Mat3 U, V;
    ADMM_PROF_T0
    oriented_svd(F, s0, s1, s2, U, V);
    ADMM_PROF_TIME(1);
    Prox<TYPE> P;
    P.mu = mu; P.lambda = lambda; P.k = smin(mu, lambda);
    P.s0.a = s0; P.s0.b = s1; P.s0.c = s2;
    V3 x2; x2.a = sa; x2.b = sb; x2.c = sc;
    if (x2.c < 0.0) x2.c *= -1.0;
    else if (fabs(x2.a) < 1.e-3 && fabs(x2.b) < 1.e-3 && fabs(x2.c) < 1.e-3) { x2.a = 1.e-3; x2.b = 1.e-3; x2.c = 1.e-3; }
#if ADMM_PROF_ON
    const double prof_g0 = absmax(P.gradient(x2));
#endif
    n_iters = lbfgs_minimize<M>(P, x2, maxIter, 1e-8, hess);
#if ADMM_NFEV_COUNT
    n_iters += 256 * P.prof_nfev;      // (timeline build only: the lane's evaluation count rides out in the upper bits, project_tet_block splits it off)
#endif
    sa = x2.a; sb = x2.b; sc = x2.c;
    mid();
    ADMM_PROF_TIME(2);
#if ADMM_PROF_ON
    if (g_tet_trace) { float *t = g_tet_trace + 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); t[0] = (float)prof_g0; t[1] = (float)P.prof_nfev; }
    ADMM_PROF_COUNT(12, n_iters); ADMM_PROF_COUNT(14, P.prof_nfev); ADMM_PROF_HIST(P.prof_nfev);
    const Mat3 zr = recompose(U, x2.a, x2.b, x2.c, V);
    ADMM_PROF_TIME(3);
    return zr;
#else
    return recompose(U, x2.a, x2.b, x2.c, V);
#endif
}

// ---- LinearTetStrain::project (CORE/TetForce.cpp:127-153) and TetVolume::project (:173-210)
// d = Dx_i + u_i ; returns the projection p (before the weighted blend)
template <bool VOLUME> ADMM_HD Mat3 project_tet_p(const Mat3 &d, double limit_min, double limit_max) {
    double s0, s1, s2; Mat3 U, V;
    svd3(d, U, s0, s1, s2, V);
    double n0, n1, n2;
    if (!VOLUME) { n0 = 1.0; n1 = 1.0; n2 = 1.0; }
    else {
        double d0 = 0, d1 = 0, d2 = 0;
        n0 = s0; n1 = s1; n2 = s2;
        for (int it = 0; it < 4; ++it) {
            double detS = n0 * n1 * n2;
            double f = detS - smin(smax(detS, limit_min), limit_max);
            double g0 = n1 * n2, g1 = n0 * n2, g2 = n0 * n1;
            double gd = g0 * d0 + (g1 * d1 + g2 * d2);
            double gg = g0 * g0 + (g1 * g1 + g2 * g2);
            double sc = -((f - gd) / gg);
            d0 = sc * g0; d1 = sc * g1; d2 = sc * g2;
            n0 = s0 + d0; n1 = s1 + d1; n2 = s2 + d2;
        }
    }
    if (det3(d) < 0.0) n2 = -1.0;
    return recompose(U, n0, n1, n2, V);
}

// ---- Eigen::JacobiSVD<Matrix<double,3,2>>(F, ComputeFullU | ComputeFullV) ----------------------
// Default ColPivHouseholderQR preconditioner (EIG/SVD/JacobiSVD.h:153-194, EIG/QR/ColPivHouseholderQR.h:428-497,
// EIG/Householder/Householder.h:65-140), then the two-sided Jacobi on the 2x2 R factor with the
// steps of svd3 above.  F col-major 3x2; only the first two columns of U (what the triangle
// forces use) and V (2x2, col-major) are returned.  Follows oracle/admm_oracle.c orc_svd32.
ADMM_HD void svd32(const double F[6], double U2[6], double &s0, double &s1, double V[4]) {
    double scale = fabs(F[0]);
#pragma unroll
    for (int i = 1; i < 6; ++i) scale = smax(scale, fabs(F[i]));
    if (scale == 0.0) scale = 1.0;
    double a0 = F[0] / scale, a1 = F[1] / scale, a2 = F[2] / scale;   // column 0 of the QR work matrix
    double b0 = F[3] / scale, b1 = F[4] / scale, b2 = F[5] / scale;   // column 1
    const double n0 = a0 * a0 + (a1 * a1 + a2 * a2), n1 = b0 * b0 + (b1 * b1 + b2 * b2);
    const bool swapped = n1 > n0;                                      // column pivoting: larger column first
    if (swapped) { double t; t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; t = a2; a2 = b2; b2 = t; }
    // Householder 0 on (a0, a1, a2)
    double tau0, e01, e02, beta0;
    {
        const double tailSq = a1 * a1 + a2 * a2;
        if (tailSq == 0.0) { tau0 = 0.0; beta0 = a0; e01 = 0.0; e02 = 0.0; }
        else {
            beta0 = sqrt(a0 * a0 + tailSq);
            if (a0 >= 0.0) beta0 = -beta0;
            e01 = a1 / (a0 - beta0); e02 = a2 / (a0 - beta0);
            tau0 = (beta0 - a0) / beta0;
        }
        double tmp = e01 * b1 + e02 * b2;
        tmp += b0;
        b0 -= tau0 * tmp; b1 -= tau0 * e01 * tmp; b2 -= tau0 * e02 * tmp;
    }
    // Householder 1 on (b1, b2)
    double tau1, e12, beta1;
    {
        const double tailSq = b2 * b2;
        if (tailSq == 0.0) { tau1 = 0.0; beta1 = b1; e12 = 0.0; }
        else {
            beta1 = sqrt(b1 * b1 + tailSq);
            if (b1 >= 0.0) beta1 = -beta1;
            e12 = b2 / (b1 - beta1);
            tau1 = (beta1 - b1) / beta1;
        }
    }
    // work matrix = R (2x2 upper triangular): w00 w01 / 0 w11
    double w00 = beta0, w10 = 0.0, w01 = b0, w11 = beta1;
    // full Q = H0 H1 applied to the identity, right to left (HouseholderSequence::evalTo)
    double q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // col-major 3x3
    // k = 1: bottom-right 2x2 block, essential (e12)
#pragma unroll
    for (int c = 1; c < 3; ++c) {
        double *col = &q[1 + 3 * c];
        double tmp = e12 * col[1];
        tmp += col[0];
        col[0] -= tau1 * tmp;
        col[1] -= tau1 * e12 * tmp;
    }
    // k = 0: whole matrix, essential (e01, e02)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double *col = &q[3 * c];
        double tmp = e01 * col[1];
        tmp = tmp + e02 * col[2];
        tmp += col[0];
        col[0] -= tau0 * tmp;
        col[1] -= tau0 * e01 * tmp;
        col[2] -= tau0 * e02 * tmp;
    }
    // V = the column permutation
    double v00 = 1.0, v10 = 0.0, v01 = 0.0, v11 = 1.0;
    if (swapped) { v00 = 0.0; v10 = 1.0; v01 = 1.0; v11 = 0.0; }
    // two-sided Jacobi on W, pair (p, q) = (1, 0); rotations act on columns 0,1 of Q and V
    const double precision = 2.0 * DBL_EPSILON;
    const double considerAsZero = 2.0 * 4.9406564584124654e-324;
    bool finished = false;
    while (!finished) {
        finished = true;
        const double threshold = smax(considerAsZero, precision * smax(fabs(w11), fabs(w00)));
        if (fabs(w10) > threshold || fabs(w01) > threshold) {
            finished = false;
            // real_2x2_jacobi_svd on [[w11, w10], [w01, w00]] (rows/cols p=1, q=0)
            double m00 = w11, m01 = w10, m10 = w01, m11 = w00;
            double c1, s1r;
            const double t = m00 + m11, d = m10 - m01;
            if (t == 0.0) { c1 = 0.0; s1r = d > 0.0 ? 1.0 : -1.0; }
            else {
                const double t2d2 = eig_hypot(t, d);
                c1 = fabs(t) / t2d2;
                s1r = d / t2d2;
                if (t < 0.0) s1r = -s1r;
            }
            if (!(c1 == 1.0 && s1r == 0.0)) { rot2(m00, m10, c1, s1r); rot2(m01, m11, c1, s1r); }
            double rc, rs;
            if (m01 == 0.0) { rc = 1.0; rs = 0.0; }
            else {
                const double tau = (m00 - m11) / (2.0 * fabs(m01));
                const double w = sqrt_ge_1(tau * tau + 1.0);
                const double tt = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
                const double sign_t = tt > 0.0 ? 1.0 : -1.0;
                const double n = rcp_in_1_2(sqrt_in_1_4(tt * tt + 1.0));
                rs = -sign_t * unit_sign(m01) * fabs(tt) * n;
                rc = n;
            }
            const double os = -rs;
            const double lc = c1 * rc - s1r * os, ls = c1 * os + s1r * rc;
            // W.applyOnTheLeft(p, q, j_left): rows p=1 and q=0
            if (!(lc == 1.0 && ls == 0.0)) {
                rot2(w10, w00, lc, ls); rot2(w11, w01, lc, ls);
                // U.applyOnTheRight(p, q, j_left^T): columns 1 and 0 of Q
#pragma unroll
                for (int r = 0; r < 3; ++r) rot2(q[r + 3], q[r], lc, ls);
            }
            // W.applyOnTheRight(p, q, j_right), V likewise: columns 1 and 0
            if (!(rc == 1.0 && -rs == 0.0)) {
                rot2(w01, w00, rc, -rs); rot2(w11, w10, rc, -rs);
                rot2(v01, v00, rc, -rs); rot2(v11, v10, rc, -rs);
            }
        }
    }
    double sa = fabs(w00), sb = fabs(w11);
    if (sa != 0.0) { const double f = unit_sign(w00); q[0] *= f; q[1] *= f; q[2] *= f; }
    if (sb != 0.0) { const double f = unit_sign(w11); q[3] *= f; q[4] *= f; q[5] *= f; }
    if (sb > sa) {   // descending, first maximum wins; a zero maximum leaves everything in place
        double t = sa; sa = sb; sb = t;
#pragma unroll
        for (int r = 0; r < 3; ++r) { t = q[r]; q[r] = q[r + 3]; q[r + 3] = t; }
        t = v00; v00 = v01; v01 = t; t = v10; v10 = v11; v11 = t;
    }
    s0 = sa * scale; s1 = sb * scale;
#pragma unroll
    for (int i = 0; i < 6; ++i) U2[i] = q[i];
    V[0] = v00; V[1] = v10; V[2] = v01; V[3] = v11;
}

// U(:, :2) diag(s) V^T for the 3x2 case: ((U Diag) V^T), Diag's zero products add nothing
ADMM_HD void recompose32(const double U2[6], double s0, double s1, const double V[4], double out[6]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i + 3 * j] = (U2[i] * s0) * V[j] + (U2[i + 3] * s1) * V[j + 2];
}

// ---- TriArea::project's projection p (CORE/TriangleForce.cpp:251-283), d = Dx_i + u_i ----------
ADMM_HD void project_triarea_p(const double d[6], int iters, double lmin, double lmax, double p[6]) {
    double U2[6], V[4], sv0, sv1;
    svd32(d, U2, sv0, sv1, V);
    double S0 = sv0, S1 = sv1, d0 = 0.0, d1 = 0.0;
    for (int i = 0; i < iters; ++i) {
        const double v = S0 * S1;
        double c = (v < lmax ? v : lmax);
        c = (c > lmin ? c : lmin);
        const double f = v - c;
        const double g0 = S1, g1 = S0;
        const double q = -((f - (g0 * d0 + g1 * d1)) / (g0 * g0 + g1 * g1));
        d0 = q * g0; d1 = q * g1;
        S0 = sv0 + d0; S1 = sv1 + d1;
    }
    recompose32(U2, S0, S1, V, p);
}

// ---- FungProx (CORE/TriangleForce.cpp:120-169): two variables riding in V3 with c == 0 ----------
// (every reduction in the L-BFGS / line search is (a + b) + c, so the zero third component leaves
// the arithmetic of the 2-variable solver untouched)
struct FungProx {
#if ADMM_PROF_ON || ADMM_NFEV_COUNT
    mutable int prof_nfev = 0;
#endif
    double mu, k;
    V3 s0;
    ADMM_HD double value(const V3 &x) const {
        if (x.a <= 0.0 || x.b <= 0.0) return kFltMax;
        const double b = 1.0;
        const double s3 = 1.0 / (x.a * x.b);
        const double I_1 = x.a * x.a + x.b * x.b + s3 * s3;
        const double t1 = mu / (b * 2.0);
        const double t2 = admm_exp(b * (I_1 - 3.0)) - 1.0;
        const double r0 = isfinite(t2) ? (t1 * t2) : kFltMax;
        const double da = x.a - s0.a, db = x.b - s0.b;
        const double r2 = (k * 0.5) * (da * da + db * db);
        return (r0 + r2);
    }
    ADMM_HD V3 gradient(const V3 &x) const {
        V3 g; g.c = 0.0;
        const double minval = 1.17549435082228750797e-38; // FLT_MIN
        if (fabs(x.a) < minval || fabs(x.b) < minval) { g.a = g.b = 1.0 * kFltMax; return g; }
        const double b = 1.0;
        const double sig3 = 1.0 / (x.a * x.b);
        const double I_1 = (x.a * x.a + x.b * x.b + sig3 * sig3);
        const double t1 = 0.5 * mu * admm_exp(b * (I_1 - 3.0));
        const double t20 = k * (x.a - s0.a), t21 = k * (x.b - s0.b);
        g.a = t1 * (2.0 * x.a - 2.0 / (x.a * x.a * x.a * x.b * x.b)) + t20;
        g.b = t1 * (2.0 * x.b - 2.0 / (x.b * x.b * x.b * x.a * x.a)) + t21;
        return g;
    }
};

// ---- FungTriangle::project (CORE/TriangleForce.cpp:227-249): z = U diag(argmin prox) V^T ---------
ADMM_HD void project_fung(const double d[6], double mu, double &hess, int &n_iters, double z[6]) {
    double U2[6], V[4], sv0, sv1;
    svd32(d, U2, sv0, sv1, V);
    FungProx P; P.mu = mu; P.k = mu; P.s0.a = sv0; P.s0.b = sv1; P.s0.c = 0.0;
    V3 x2; x2.a = sv0; x2.b = sv1; x2.c = 0.0;
    n_iters = lbfgs_minimize<10>(P, x2, 10, 1e-6, hess);
    recompose32(U2, x2.a, x2.b, V, z);
}

} // namespace admm_dev
