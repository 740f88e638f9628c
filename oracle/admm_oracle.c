/*
 * admm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See admm_oracle.h.  Plain C99; compile with -ffp-contract=off so that the
 * operation order written here is the operation order executed (the reference
 * is built for x86-64 SSE2 without FMA).
 *
 * Reference paths (all under /root/reference):
 *   CORE = deps/admm-elastic-sca/src/system
 *   OPT  = deps/admm-elastic-sca/deps/cppoptlib/include/cppoptlib
 *   EIG  = deps/admm-elastic-sca/deps/Eigen3/Eigen/src
 *
 * Summation orders follow what Eigen 3.2.5 generates for the expression the
 * reference wrote: fixed-size 3-vectors reduce as a0+(a1+a2)
 * (EIG/Core/Redux.h redux_novec_unroller), dynamic-size vectors as (a0+a1)+a2
 * (Redux.h linear traversal), small products as ((p0+p1)+p2)
 * (EIG/Core/products/CoeffBasedProduct.h).
 */
#define _POSIX_C_SOURCE 199309L
#include "admm_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* std::min / std::max exactly as libstdc++ defines them (NaN behaviour) */
#define STD_MIN(a, b) (((b) < (a)) ? (b) : (a))
#define STD_MAX(a, b) (((a) < (b)) ? (b) : (a))
#define FLTMAX ((double)FLT_MAX)

/* fixed-size Vector3d reductions: a0 + (a1 + a2) */
static double dot3f(const double *a, const double *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
static double norm3f(const double *a) { return sqrt(dot3f(a, a)); }
static void cross3(const double *a, const double *b, double *c) {
    /* EIG/Geometry/OrthoMethods.h:35-39 */
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
/* dynamic-size reductions: ((a0 + a1) + a2) + ... */
static double dotn(const double *a, const double *b, int n) {
    double r = a[0] * b[0];
    for (int i = 1; i < n; ++i) r = r + a[i] * b[i];
    return r;
}
static double absmaxn(const double *a, int n) {
    double r = fabs(a[0]);
    for (int i = 1; i < n; ++i) { double c = fabs(a[i]); r = STD_MAX(r, c); }
    return r;
}
/* Matrix3d::determinant(), EIG/LU/Determinant.h:18-23,61-68 (col-major m) */
#define M3(m, r, c) ((m)[(r) + 3 * (c)])
static double det3(const double *m) {
    double h0 = M3(m, 0, 0) * (M3(m, 1, 1) * M3(m, 2, 2) - M3(m, 1, 2) * M3(m, 2, 1));
    double h1 = M3(m, 0, 1) * (M3(m, 1, 0) * M3(m, 2, 2) - M3(m, 1, 2) * M3(m, 2, 0));
    double h2 = M3(m, 0, 2) * (M3(m, 1, 0) * M3(m, 2, 1) - M3(m, 1, 1) * M3(m, 2, 0));
    return h0 - h1 + h2;
}
/* Matrix3d::inverse(), EIG/LU/Inverse.h:113-160 */
static double cof3(const double *m, int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M3(m, i1, j1) * M3(m, i2, j2) - M3(m, i1, j2) * M3(m, i2, j1);
}
static void inv3(const double *m, double *r) {
    double c0[3] = { cof3(m, 0, 0), cof3(m, 1, 0), cof3(m, 2, 0) };
    double det = c0[0] * M3(m, 0, 0) + (c0[1] * M3(m, 1, 0) + c0[2] * M3(m, 2, 0));
    double invdet = 1.0 / det;
    for (int j = 0; j < 3; ++j) M3(r, 0, j) = c0[j] * invdet;
    for (int j = 0; j < 3; ++j) M3(r, 1, j) = cof3(m, j, 1) * invdet;
    for (int j = 0; j < 3; ++j) M3(r, 2, j) = cof3(m, j, 2) * invdet;
}

/* ======================================================================= */
/* Eigen::JacobiSVD<Matrix3d>, EIG/SVD/JacobiSVD.h:824-933                  */
/* ======================================================================= */

/* internal::apply_rotation_in_the_plane, EIG/Jacobi/Jacobi.h:302-430 */
static void rot_apply(double *x, int incx, double *y, int incy, int n, double c, double s) {
    if (c == 1.0 && s == 0.0) return;
    for (int i = 0; i < n; ++i) {
        double xi = *x, yi = *y;
        *x = c * xi + s * yi;
        *y = -s * xi + c * yi;
        x += incx; y += incy;
    }
}
/* numext::hypot, EIG/Core/MathFunctions.h:284-302 */
static double eig_hypot(double x, double y) {
    double ax = fabs(x), ay = fabs(y);
    double p = STD_MAX(ax, ay);
    if (p == 0.0) return 0.0;
    double q = STD_MIN(ax, ay);
    double qp = q / p;
    return p * sqrt(1.0 + qp * qp);
}
/* JacobiRotation::makeJacobi(x,y,z), EIG/Jacobi/Jacobi.h:80-110 */
static void make_jacobi(double x, double y, double z, double *c, double *s) {
    if (y == 0.0) { *c = 1.0; *s = 0.0; return; }
    double tau = (x - z) / (2.0 * fabs(y));
    double w = sqrt(tau * tau + 1.0);
    double t;
    if (tau > 0.0) t = 1.0 / (tau + w); else t = 1.0 / (tau - w);
    double sign_t = t > 0.0 ? 1.0 : -1.0;
    double n = 1.0 / sqrt(t * t + 1.0);
    *s = -sign_t * (y / fabs(y)) * fabs(t) * n;
    *c = n;
}
/* internal::real_2x2_jacobi_svd, EIG/SVD/JacobiSVD.h:415-443 */
static void real_2x2_jacobi_svd(const double *W, int ld, int p, int q,
                                double *lc, double *ls, double *rc, double *rs) {
    double m00 = W[p + ld * p], m01 = W[p + ld * q], m10 = W[q + ld * p], m11 = W[q + ld * q];
    double c1, s1;
    double t = m00 + m11, d = m10 - m01;
    if (t == 0.0) { c1 = 0.0; s1 = d > 0.0 ? 1.0 : -1.0; }
    else {
        double t2d2 = eig_hypot(t, d);
        c1 = fabs(t) / t2d2;
        s1 = d / t2d2;
        if (t < 0.0) s1 = -s1;
    }
    /* m.applyOnTheLeft(0,1,rot1) */
    if (!(c1 == 1.0 && s1 == 0.0)) {
        double a = m00, b = m10;
        m00 = c1 * a + s1 * b; m10 = -s1 * a + c1 * b;
        a = m01; b = m11;
        m01 = c1 * a + s1 * b; m11 = -s1 * a + c1 * b;
    }
    make_jacobi(m00, m01, m11, rc, rs);
    /* *j_left = rot1 * j_right->transpose(), Jacobi.h:52-57,60 */
    double oc = *rc, os = -(*rs);
    *lc = c1 * oc - s1 * os;
    *ls = c1 * os + s1 * oc;
}

/* step 2-4 of JacobiSVD::compute on an n x n work matrix W (ld = n), with
 * U (mu x mu, ld mu; rotations touch cols < n) and V (n x n) pre-initialised. */
static void jacobi_sweeps(double *W, int n, double *U, int mu, double *V, double *S, double scale) {
    const double precision = 2.0 * DBL_EPSILON;
    const double considerAsZero = 2.0 * 4.9406564584124654e-324; /* 2*denorm_min */
    int finished = 0;
    while (!finished) {
        finished = 1;
        for (int p = 1; p < n; ++p) {
            for (int q = 0; q < p; ++q) {
                double app = fabs(W[p + n * p]), aqq = fabs(W[q + n * q]);
                double mx = STD_MAX(app, aqq);
                double pm = precision * mx;
                double threshold = STD_MAX(considerAsZero, pm);
                if (fabs(W[p + n * q]) > threshold || fabs(W[q + n * p]) > threshold) {
                    finished = 0;
                    double lc, ls, rc, rs;
                    real_2x2_jacobi_svd(W, n, p, q, &lc, &ls, &rc, &rs);
                    rot_apply(&W[p], n, &W[q], n, n, lc, ls);              /* W.applyOnTheLeft(p,q,j_left) */
                    rot_apply(&U[mu * p], 1, &U[mu * q], 1, mu, lc, ls);   /* U.applyOnTheRight(p,q,j_left^T) */
                    rot_apply(&W[n * p], 1, &W[n * q], 1, n, rc, -rs);     /* W.applyOnTheRight(p,q,j_right) */
                    rot_apply(&V[n * p], 1, &V[n * q], 1, n, rc, -rs);     /* V.applyOnTheRight(p,q,j_right) */
                }
            }
        }
    }
    /* step 3: make the diagonal positive */
    for (int i = 0; i < n; ++i) {
        double a = fabs(W[i + n * i]);
        S[i] = a;
        if (a != 0.0) { double f = W[i + n * i] / a; for (int r = 0; r < mu; ++r) U[r + mu * i] *= f; }
    }
    /* step 4: sort descending (first maximum wins, EIG/Core/Visitor.h) */
    for (int i = 0; i < n; ++i) {
        int pos = 0; double mx = S[i];
        for (int k = 1; k < n - i; ++k) if (S[i + k] > mx) { mx = S[i + k]; pos = k; }
        if (mx == 0.0) break;
        if (pos) {
            pos += i;
            double t = S[i]; S[i] = S[pos]; S[pos] = t;
            for (int r = 0; r < mu; ++r) { t = U[r + mu * i]; U[r + mu * i] = U[r + mu * pos]; U[r + mu * pos] = t; }
            for (int r = 0; r < n; ++r) { t = V[r + n * i]; V[r + n * i] = V[r + n * pos]; V[r + n * pos] = t; }
        }
    }
    for (int i = 0; i < n; ++i) S[i] *= scale;
}

void orc_svd3(const double F[9], double U[9], double S[3], double V[9]) {
    double scale = absmaxn(F, 9);
    if (scale == 0.0) scale = 1.0;
    double W[9];
    for (int i = 0; i < 9; ++i) { W[i] = F[i] / scale; U[i] = 0.0; V[i] = 0.0; }
    U[0] = U[4] = U[8] = 1.0; V[0] = V[4] = V[8] = 1.0;
    jacobi_sweeps(W, 3, U, 3, V, S, scale);
}

/* helper::oriented_svd, CORE/TetForce.cpp:80-102 */
void orc_oriented_svd(const double F[9], double S[3], double U[9], double Vt[9]) {
    double V[9];
    orc_svd3(F, U, S, V);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M3(Vt, r, c) = M3(V, c, r);
    if (det3(U) < 0.0) { for (int r = 0; r < 3; ++r) M3(U, r, 2) = -M3(U, r, 2); S[2] *= -1.0; }
    if (det3(Vt) < 0.0) { for (int c = 0; c < 3; ++c) M3(Vt, 2, c) = -M3(Vt, 2, c); S[2] *= -1.0; }
}

/*
 * Eigen::JacobiSVD<Matrix<double,3,2>> with the default
 * ColPivHouseholderQR preconditioner, EIG/SVD/JacobiSVD.h:153-194,
 * EIG/QR/ColPivHouseholderQR.h:428-497, EIG/Householder/Householder.h:65-140,
 * EIG/Householder/HouseholderSequence.h (evalTo, full Q).
 * U is 3x3, V is 2x2, all col-major.
 */
void orc_svd32(const double F[6], double U[9], double S[2], double V[4]) {
    double scale = absmaxn(F, 6);
    if (scale == 0.0) scale = 1.0;
    double qr[6];
    for (int i = 0; i < 6; ++i) qr[i] = F[i] / scale;
    double hc[2] = {0, 0};
    int trans[2] = {0, 1};
    double csn[2];
    for (int k = 0; k < 2; ++k) csn[k] = qr[3 * k] * qr[3 * k] + (qr[3 * k + 1] * qr[3 * k + 1] + qr[3 * k + 2] * qr[3 * k + 2]);
    for (int k = 0; k < 2; ++k) {
        int big = k;
        if (k == 0 && csn[1] > csn[0]) big = 1;
        /* recompute the squared norm of the selected column's tail (value unused further for 3x2) */
        trans[k] = big;
        if (big != k) {
            for (int r = 0; r < 3; ++r) { double t = qr[r + 3 * k]; qr[r + 3 * k] = qr[r + 3 * big]; qr[r + 3 * big] = t; }
            double t = csn[k]; csn[k] = csn[big]; csn[big] = t;
        }
        /* makeHouseholderInPlace on qr.col(k).tail(3-k) */
        int len = 3 - k;
        double *col = &qr[k + 3 * k];
        double tailSq;
        if (len == 3) tailSq = col[1] * col[1] + col[2] * col[2];
        else tailSq = col[1] * col[1];
        double c0 = col[0], beta, tau;
        if (tailSq == 0.0) { tau = 0.0; beta = c0; for (int i = 1; i < len; ++i) col[i] = 0.0; }
        else {
            beta = sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.0) beta = -beta;
            for (int i = 1; i < len; ++i) col[i] = col[i] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        col[0] = beta;
        /* apply to the remaining column (only k==0 has one) */
        if (k == 0) {
            double *c1 = &qr[3];
            double tmp = col[1] * c1[1] + col[2] * c1[2];
            tmp += c1[0];
            c1[0] -= tau * tmp;
            c1[1] -= tau * col[1] * tmp;
            c1[2] -= tau * col[2] * tmp;
            csn[1] -= c1[0] * c1[0];
        }
    }
    /* work matrix = upper triangle of R (2x2) */
    double W[4] = { qr[0], 0.0, qr[3], qr[4] };
    /* U = householderQ() as a full 3x3: Q = H0 H1, applied to identity right-to-left */
    for (int i = 0; i < 9; ++i) U[i] = 0.0;
    U[0] = U[4] = U[8] = 1.0;
    for (int k = 1; k >= 0; --k) {
        /* U.bottomRightCorner(3-k,3-k).applyHouseholderOnTheLeft(essential_k, hc[k]) */
        int len = 3 - k;
        const double *ess = &qr[k + 1 + 3 * k];
        for (int c = k; c < 3; ++c) {
            double *colp = &U[k + 3 * c];
            if (len == 1) { colp[0] *= 1.0 - hc[k]; continue; }
            double tmp = 0.0;
            for (int i = 1; i < len; ++i) tmp = (i == 1) ? ess[0] * colp[1] : tmp + ess[i - 1] * colp[i];
            tmp += colp[0];
            colp[0] -= hc[k] * tmp;
            for (int i = 1; i < len; ++i) colp[i] -= hc[k] * ess[i - 1] * tmp;
        }
    }
    /* V = column permutation */
    V[0] = V[3] = 1.0; V[1] = V[2] = 0.0;
    if (trans[0] == 1) { V[0] = 0.0; V[1] = 1.0; V[2] = 1.0; V[3] = 0.0; }
    jacobi_sweeps(W, 2, U, 3, V, S, scale);
}

/* ======================================================================= */
/* Prox problems: NHProx / StVKProx, CORE/TetForce.cpp:216-297              */
/* ======================================================================= */
typedef struct { int type; double mu, lambda, k; double s0[3]; int n_fev; } prox3;

/* NHProx::energyDensity, CORE/TetForce.cpp:216-225 */
static double nh_energy(const prox3 *p, const double *s) {
    double Sig_det = (s[0] * s[1] * s[2]);
    double I_1 = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    double I_3 = Sig_det * Sig_det;
    double log_I3 = log(I_3);
    double t1 = 0.5 * p->mu * (I_1 - log_I3 - 3.0);
    double t2 = 0.125 * p->lambda * log_I3 * log_I3;
    return t1 + t2;
}
/* StVKProx::energyDensity, CORE/TetForce.cpp:269-278 */
static double stvk_energy(const prox3 *p, const double *s) {
    double st[3];
    for (int i = 0; i < 3; ++i) st[i] = 0.5 * (s[i] * s[i] - 1.0);
    double tr = (st[0] + st[1]) + st[2];
    double st_tr2 = tr * tr;
    double dd = st[0] * st[0] + (st[1] * st[1] + st[2] * st[2]); /* trace of a fixed 3x3 */
    return (p->mu * dd + (p->lambda * 0.5 * st_tr2));
}
/*
 * FungProx (CORE/TriangleForce.cpp:120-169): a problem in TWO variables.  It runs through the
 * same 3-variable L-BFGS / line-search code with x[2] == s[2] == g[2] == 0: every reduction
 * there is the dynamic-size (a0 + a1) + a2, and adding a zero changes nothing, so the numbers
 * are those of the 2-variable solver.
 */
static double fung_value(const prox3 *p, const double *x) {
    if (x[0] <= 0.0 || x[1] <= 0.0) return FLTMAX;
    const double b = 1.0;
    double s3 = 1.0 / (x[0] * x[1]);
    double I_1 = x[0] * x[0] + x[1] * x[1] + s3 * s3;
    double t1 = p->mu / (b * 2.0);
    double t2 = exp(b * (I_1 - 3.0)) - 1.0;
    double r0;
    if (!isfinite(t2)) r0 = FLTMAX; else r0 = (t1 * t2);
    double d0 = x[0] - p->s0[0], d1 = x[1] - p->s0[1];
    double r2 = (p->k * 0.5) * (d0 * d0 + d1 * d1);
    return (r0 + r2);
}
static void fung_gradient(const prox3 *p, const double *x, double *g) {
    const double minval = (double)FLT_MIN, b = 1.0;
    g[2] = 0.0;
    if (fabs(x[0]) < minval || fabs(x[1]) < minval) { g[0] = g[1] = 1.0 * FLTMAX; return; }
    double sig3 = 1.0 / (x[0] * x[1]);
    double I_1 = (x[0] * x[0] + x[1] * x[1] + sig3 * sig3);
    double t1 = 0.5 * p->mu * exp(b * (I_1 - 3.0));
    double t20 = p->k * (x[0] - p->s0[0]), t21 = p->k * (x[1] - p->s0[1]);
    g[0] = t1 * (2.0 * x[0] - 2.0 / (x[0] * x[0] * x[0] * x[1] * x[1])) + t20;
    g[1] = t1 * (2.0 * x[1] - 2.0 / (x[1] * x[1] * x[1] * x[0] * x[0])) + t21;
}

/* NHProx::value :228-233 / StVKProx::value :281-287 */
static double prox_value(prox3 *p, const double *x) {
    p->n_fev++;
    if (p->type == 2) return fung_value(p, x);
    if (x[0] < 0.0 || x[1] < 0.0 || x[2] < 0.0) return FLTMAX;
    double d[3] = { x[0] - p->s0[0], x[1] - p->s0[1], x[2] - p->s0[2] };
    if (p->type == 0) {
        double r = nh_energy(p, x);
        double r2 = (p->k * 0.5) * ((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]); /* dynamic-size expr */
        return (1.0 * r + r2);
    } else {
        double r = stvk_energy(p, x);
        double r2 = (p->k * 0.5) * (d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])); /* fixed-size expr */
        return (r + r2);
    }
}
/* NHProx::gradient :235-243 / StVKProx::gradient :289-297 */
static void prox_gradient(prox3 *p, const double *x, double *g) {
    if (p->type == 2) { fung_gradient(p, x, g); return; }
    if (p->type == 0) {
        double detSigma = x[0] * x[1] * x[2];
        if (detSigma <= 0.0) { g[0] = g[1] = g[2] = 1.0 * FLTMAX; return; }
        double inv[3] = { 1.0 / x[0], 1.0 / x[1], 1.0 / x[2] };
        double ll = p->lambda * log(detSigma);
        for (int i = 0; i < 3; ++i)
            g[i] = 1.0 * (p->mu * (x[i] - inv[i]) + ll * inv[i]) + p->k * (x[i] - p->s0[i]);
    } else {
        double xx = (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2];
        double c2 = 0.5 * p->lambda * (xx - 3.0);
        for (int i = 0; i < 3; ++i) {
            double term1 = p->mu * x[i] * (x[i] * x[i] - 1.0);
            double term2 = c2 * x[i];
            g[i] = term1 + term2 + p->k * (x[i] - p->s0[i]);
        }
    }
}

/* ======================================================================= */
/* More-Thuente line search, OPT/linesearch/morethuente.h                   */
/* ======================================================================= */
#define NV 3

/* MoreThuente::cstep, morethuente.h:169-308 */
static int mt_cstep(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy, double *stp,
                    double fp, double dp, int *brackt, double stpmin, double stpmax, int *info) {
    *info = 0;
    int bound = 0;
    if ((*brackt & ((*stp <= STD_MIN(*stx, *sty)) | (*stp >= STD_MAX(*stx, *sty)))) | (*dx * (*stp - *stx) >= 0.0)
        | (stpmax < stpmin)) {
        return -1;
    }
    double sgnd = dp * (*dx / fabs(*dx));
    double stpf = 0, stpc = 0, stpq = 0;
    if (fp > *fx) {
        *info = 1; bound = 1;
        double theta = 3. * (*fx - fp) / (*stp - *stx) + *dx + dp;
        double s = STD_MAX(theta, STD_MAX(*dx, dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        double p = (gamma - *dx) + theta;
        double q = ((gamma - *dx) + gamma) + dp;
        double r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.) * (*stp - *stx);
        if (fabs(stpc - *stx) < fabs(stpq - *stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2;
        *brackt = 1;
    } else if (sgnd < 0.0) {
        *info = 2; bound = 0;
        double theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        double s = STD_MAX(theta, STD_MAX(*dx, dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        double p = (gamma - dp) + theta;
        double q = ((gamma - dp) + gamma) + *dx;
        double r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
        else stpf = stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) {
        *info = 3; bound = 1;
        double theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        double s = STD_MAX(theta, STD_MAX(*dx, dp));
        double a = (theta / s) * (theta / s) - (*dx / s) * (dp / s);
        double gamma = s * sqrt(STD_MAX(0., a));
        if (*stp > *stx) gamma = -gamma;
        double p = (gamma - dp) + theta;
        double q = (gamma + (*dx - dp)) + gamma;
        double r = p / q;
        if ((r < 0.0) & (gamma != 0.0)) stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) {
            if (fabs(*stp - stpc) < fabs(*stp - stpq)) stpf = stpc; else stpf = stpq;
        } else {
            if (fabs(*stp - stpc) > fabs(*stp - stpq)) stpf = stpc; else stpf = stpq;
        }
    } else {
        *info = 4; bound = 0;
        if (*brackt) {
            double theta = 3 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            double s = STD_MAX(theta, STD_MAX(*dy, dp));
            double gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            double p = (gamma - dp) + theta;
            double q = ((gamma - dp) + gamma) + *dy;
            double r = p / q;
            stpc = *stp + r * (*sty - *stp);
            stpf = stpc;
        } else if (*stp > *stx) stpf = stpmax;
        else stpf = stpmin;
    }
    if (fp > *fx) { *sty = *stp; *fy = fp; *dy = dp; }
    else {
        if (sgnd < 0.0) { *sty = *stx; *fy = *fx; *dy = *dx; }
        *stx = *stp; *fx = fp; *dx = dp;
    }
    stpf = STD_MIN(stpmax, stpf);
    stpf = STD_MAX(stpmin, stpf);
    *stp = stpf;
    if (*brackt & bound) {
        if (*sty > *stx) { double c = *stx + 0.66 * (*sty - *stx); *stp = STD_MIN(c, *stp); }
        else { double c = *stx + 0.66 * (*sty - *stx); *stp = STD_MAX(c, *stp); }
    }
    return 0;
}

/* MoreThuente::cvsrch, morethuente.h:43-167.  x (=wa) is the base point, s the
 * direction; returns through *stp. */
static int mt_cvsrch(prox3 *P, const double *wa, double f, double *g, double *stp, const double *s) {
    int info = 0, infoc = 1;
    const double xtol = 1e-15, ftol = 1e-4, gtol = 1e-2, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4;
    const int maxfev = 20;
    int nfev = 0;
    double dginit = dotn(g, s, NV);
    if (dginit >= 0.0) return -1;
    int brackt = 0, stage1 = 1;
    double finit = f, dgtest = ftol * dginit;
    double width = stpmax - stpmin, width1 = 2 * width;
    double stx = 0.0, fx = finit, dgx = dginit, sty = 0.0, fy = finit, dgy = dginit;
    double stmin = 0, stmax = 0;
    double x[NV];
    for (;;) {
        if (brackt) { stmin = STD_MIN(stx, sty); stmax = STD_MAX(stx, sty); }
        else { stmin = stx; stmax = *stp + xtrapf * (*stp - stx); }
        *stp = STD_MAX(*stp, stpmin);
        *stp = STD_MIN(*stp, stpmax);
        if ((brackt && ((*stp <= stmin) | (*stp >= stmax))) | (nfev >= maxfev - 1) | (infoc == 0)
            | (brackt & (stmax - stmin <= xtol * stmax))) {
            *stp = stx;
        }
        for (int i = 0; i < NV; ++i) x[i] = wa[i] + *stp * s[i];
        f = prox_value(P, x);
        prox_gradient(P, x, g);
        nfev++;
        double dg = dotn(g, s, NV);
        double ftest1 = finit + *stp * dgtest;
        if ((brackt & ((*stp <= stmin) | (*stp >= stmax))) | (infoc == 0)) info = 6;
        if ((*stp == stpmax) & (f <= ftest1) & (dg <= dgtest)) info = 5;
        if ((*stp == stpmin) & ((f > ftest1) | (dg >= dgtest))) info = 4;
        if (nfev >= maxfev) info = 3;
        if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
        if ((f <= ftest1) & (fabs(dg) <= gtol * (-dginit))) info = 1;
        if (info != 0) return -1;
        if (stage1 & (f <= ftest1) & (dg >= STD_MIN(ftol, gtol) * dginit)) stage1 = 0;
        if (stage1 & (f <= fx) & (f > ftest1)) {
            double fm = f - *stp * dgtest;
            double fxm = fx - stx * dgtest;
            double fym = fy - sty * dgtest;
            double dgm = dg - dgtest;
            double dgxm = dgx - dgtest;
            double dgym = dgy - dgtest;
            mt_cstep(&stx, &fxm, &dgxm, &sty, &fym, &dgym, stp, fm, dgm, &brackt, stmin, stmax, &infoc);
            fx = fxm + stx * dgtest;
            fy = fym + sty * dgtest;
            dgx = dgxm + dgtest;
            dgy = dgym + dgtest;
        } else {
            mt_cstep(&stx, &fx, &dgx, &sty, &fy, &dgy, stp, f, dg, &brackt, stmin, stmax, &infoc);
        }
        if (brackt) {
            if (fabs(sty - stx) >= 0.66 * width1) *stp = stx + 0.5 * (sty - stx);
            width1 = width;
            width = fabs(sty - stx);
        }
    }
    return 0;
}

/* MoreThuente::linesearch, morethuente.h:25-41 */
static double mt_linesearch(prox3 *P, const double *x, const double *dir, double alpha_init) {
    double ak = alpha_init;
    double fval = prox_value(P, x);
    double g[NV];
    prox_gradient(P, x, g);
    mt_cvsrch(P, x, fval, g, &ak, dir);
    return ak;
}

/* ======================================================================= */
/* cppoptlib::lbfgssolver<double>::minimize, OPT/solver/lbfgssolver.h:43-144 */
/* ======================================================================= */
#define LBFGS_MMAX 10
static int lbfgs_minimize(prox3 *P, double *x0, int maxIter, double gradTol, double *init_hess) {
    int m_ = STD_MIN(maxIter, 10);
    const double eps_g = gradTol, eps_x = 1e-8;
    double s[LBFGS_MMAX][NV], y[LBFGS_MMAX][NV], alpha[LBFGS_MMAX], rho[LBFGS_MMAX];
    memset(s, 0, sizeof s); memset(y, 0, sizeof y); memset(alpha, 0, sizeof alpha); memset(rho, 0, sizeof rho);
    double grad[NV], q[NV], grad_old[NV], x_old[NV];
    prox_gradient(P, x0, grad);
    double gamma_k = *init_hess;
    double gradNorm = 0;
    double gi = 1.0 / absmaxn(grad, NV);
    double alpha_init = STD_MIN(1.0, gi);
    int globIter = 0;
    int maxiter = maxIter;
    double new_hess_guess = 1.0;
    for (int k = 0; k < maxiter; k++) {
        for (int i = 0; i < NV; ++i) { x_old[i] = x0[i]; grad_old[i] = grad[i]; q[i] = grad[i]; }
        globIter++;
        int iter = STD_MIN(m_, k);
        for (int i = iter - 1; i >= 0; --i) {
            rho[i] = 1.0 / dotn(s[i], y[i], NV);
            alpha[i] = rho[i] * dotn(s[i], q, NV);
            for (int j = 0; j < NV; ++j) q[j] = q[j] - alpha[i] * y[i][j];
        }
        for (int j = 0; j < NV; ++j) q[j] = gamma_k * q[j];
        for (int i = 0; i < iter; ++i) {
            double beta = rho[i] * dotn(q, y[i], NV);
            for (int j = 0; j < NV; ++j) q[j] = q[j] + (alpha[i] - beta) * s[i][j];
        }
        double dir = dotn(q, grad, NV);
        if (dir < 1e-4) {
            for (int j = 0; j < NV; ++j) q[j] = grad[j];
            maxiter -= k;
            k = 0;
            gi = 1.0 / absmaxn(grad, NV);
            alpha_init = STD_MIN(1.0, gi);
        }
        double mq[NV];
        for (int j = 0; j < NV; ++j) mq[j] = -q[j];
        const double rate = mt_linesearch(P, x0, mq, alpha_init);
        for (int j = 0; j < NV; ++j) x0[j] = x0[j] - rate * q[j];
        double dxx[NV];
        for (int j = 0; j < NV; ++j) dxx[j] = x_old[j] - x0[j];
        if (dotn(dxx, dxx, NV) < eps_x) break;
        prox_gradient(P, x0, grad);
        gradNorm = absmaxn(grad, NV);
        if (gradNorm < eps_g) { new_hess_guess = gamma_k; break; }
        double s_temp[NV], y_temp[NV];
        for (int j = 0; j < NV; ++j) { s_temp[j] = x0[j] - x_old[j]; y_temp[j] = grad[j] - grad_old[j]; }
        if (k < m_) {
            for (int j = 0; j < NV; ++j) { s[k][j] = s_temp[j]; y[k][j] = y_temp[j]; }
        } else {
            for (int i = 0; i < m_ - 1; ++i) for (int j = 0; j < NV; ++j) { s[i][j] = s[i + 1][j]; y[i][j] = y[i + 1][j]; }
            for (int j = 0; j < NV; ++j) { s[m_ - 1][j] = s_temp[j]; y[m_ - 1][j] = y_temp[j]; }
        }
        gamma_k = dotn(s_temp, y_temp, NV) / dotn(y_temp, y_temp, NV);
        alpha_init = 1.0;
    }
    *init_hess = new_hess_guess;
    return globIter;
}

/* ======================================================================= */
/* Forces                                                                    */
/* ======================================================================= */
void orc_force_construct(orc_force *f, int kind, const int *idx, const double *params) {
    memset(f, 0, sizeof *f);
    f->kind = kind;
    for (int i = 0; i < ADMM_KIND_NODES[kind]; ++i) f->idx[i] = idx[i];
    for (int i = 0; i < ADMM_KIND_PARAMS[kind]; ++i) f->params[i] = params[i];
    f->active = 1;
    switch (kind) {
    case ADMM_KIND_ANCHOR: /* CORE/AnchorForce.hpp:57-60,88-92 */
        if (params[0] > 0.0) f->weight = params[0]; else f->weight = 1000.f;
        break;
    case ADMM_KIND_BEND: /* CORE/BendForce.hpp:32 */
        f->weight = sqrt(params[0]);
        break;
    case ADMM_KIND_COLLISION: /* CORE/CollisionForce.hpp:33 */
        f->weight = params[0];
        break;
    case ADMM_KIND_TET_NH: case ADMM_KIND_TET_STVK: /* CORE/TetForce.hpp:127-128; OPT/meta.h:33 */
        f->state[0] = f->state[1] = f->state[2] = 1.0; f->state[3] = 1.0;
        break;
    case ADMM_KIND_TRI_FUNG: /* solver settings init_hess, OPT/meta.h:33 */
        f->state[3] = 1.0;
        break;
    default: break;
    }
}

/* helper::init_tet_force, CORE/TetForce.cpp:28-57 */
static void init_tet_force(const int *idx, const double *x, double *volume, double *B) {
    const double *v0 = &x[3 * idx[0]], *v1 = &x[3 * idx[1]], *v2 = &x[3 * idx[2]], *v3 = &x[3 * idx[3]];
    double edges[9], inv[9];
    for (int r = 0; r < 3; ++r) { M3(edges, r, 0) = v1[r] - v0[r]; M3(edges, r, 1) = v2[r] - v0[r]; M3(edges, r, 2) = v3[r] - v0[r]; }
    inv3(edges, inv);
    /* B = D * Xg.inverse(), D = [-1 -1 -1; I3]: coefficient-based product ((p0+p1)+p2) */
    for (int j = 0; j < 3; ++j) {
        B[0 + 4 * j] = (-1.0 * M3(inv, 0, j) + -1.0 * M3(inv, 1, j)) + -1.0 * M3(inv, 2, j);
        B[1 + 4 * j] = (1.0 * M3(inv, 0, j) + 0.0 * M3(inv, 1, j)) + 0.0 * M3(inv, 2, j);
        B[2 + 4 * j] = (0.0 * M3(inv, 0, j) + 1.0 * M3(inv, 1, j)) + 0.0 * M3(inv, 2, j);
        B[3 + 4 * j] = (0.0 * M3(inv, 0, j) + 0.0 * M3(inv, 1, j)) + 1.0 * M3(inv, 2, j);
    }
    double a[3], b[3], c[3], cr[3];
    for (int r = 0; r < 3; ++r) { a[r] = v0[r] - v3[r]; b[r] = v1[r] - v3[r]; c[r] = v2[r] - v3[r]; }
    cross3(b, c, cr);
    *volume = fabs(dot3f(a, cr)) / 6.0;
}

/* LimitedTriangleStrain::initialize, CORE/TriangleForce.cpp:29-63 */
static void init_tri_force(const int *idx, const double *x, double *area, double *B) {
    const double *x1 = &x[3 * idx[0]], *x2 = &x[3 * idx[1]], *x3 = &x[3 * idx[2]];
    double e12[3], e13[3], n1[3], n2[3], t[3];
    for (int r = 0; r < 3; ++r) { e12[r] = x2[r] - x1[r]; e13[r] = x3[r] - x1[r]; }
    double l = norm3f(e12);
    for (int r = 0; r < 3; ++r) n1[r] = e12[r] / l;
    double d = dot3f(e13, n1);
    for (int r = 0; r < 3; ++r) t[r] = e13[r] - d * n1[r];
    l = norm3f(t);
    for (int r = 0; r < 3; ++r) n2[r] = t[r] / l;
    /* Xg = basis^T * edges (2x2), inner size 3: ((p0+p1)+p2) */
    double Xg[4];
    Xg[0] = (n1[0] * e12[0] + n1[1] * e12[1]) + n1[2] * e12[2];
    Xg[1] = (n2[0] * e12[0] + n2[1] * e12[1]) + n2[2] * e12[2];
    Xg[2] = (n1[0] * e13[0] + n1[1] * e13[1]) + n1[2] * e13[2];
    Xg[3] = (n2[0] * e13[0] + n2[1] * e13[1]) + n2[2] * e13[2];
    /* 2x2 inverse, EIG/LU/Inverse.h:44-53 */
    double det = Xg[0] * Xg[3] - Xg[2] * Xg[1];
    double invdet = 1.0 / det;
    double Xi[4];
    Xi[0] = Xg[3] * invdet; Xi[1] = -Xg[1] * invdet; Xi[2] = -Xg[2] * invdet; Xi[3] = Xg[0] * invdet;
    /* B = D * Xg^-1, D = [-1 -1; 1 0; 0 1], B is 3x2 col-major */
    for (int j = 0; j < 2; ++j) {
        B[0 + 3 * j] = -1.0 * Xi[0 + 2 * j] + -1.0 * Xi[1 + 2 * j];
        B[1 + 3 * j] = 1.0 * Xi[0 + 2 * j] + 0.0 * Xi[1 + 2 * j];
        B[2 + 3 * j] = 0.0 * Xi[0 + 2 * j] + 1.0 * Xi[1 + 2 * j];
    }
    *area = fabs(det / 2.0f);
}

/* BendForce::initialize, CORE/BendForce.cpp:26-56 (alpha only; jac/lambda are dead) */
static void init_bend(const int *idx, const double *x, double *alpha) {
    const double *x0 = &x[3 * idx[0]], *x1 = &x[3 * idx[1]], *x2 = &x[3 * idx[2]], *x3 = &x[3 * idx[3]];
    double xA[3], xB[3], xC[3] = {0, 0, 0}, xD[3];
    for (int r = 0; r < 3; ++r) { xA[r] = x0[r] - x2[r]; xB[r] = x1[r] - x2[r]; xD[r] = x3[r] - x2[r]; }
    double c[3];
    cross3(xA, xD, c); double area1 = 0.5 * norm3f(c);
    cross3(xD, xB, c); double area2 = 0.5 * norm3f(c);
    double hA = 2.0 * area1 / norm3f(xD);
    double hB = 2.0 * area2 / norm3f(xD);
    double a[3], b[3], nC[3], nD[3];
    for (int r = 0; r < 3; ++r) { a[r] = xC[r] - xB[r]; b[r] = xC[r] - xA[r]; }
    cross3(a, b, nC);
    for (int r = 0; r < 3; ++r) { a[r] = xD[r] - xA[r]; b[r] = xD[r] - xB[r]; }
    cross3(a, b, nD);
    alpha[0] = hB / (hA + hB);
    alpha[1] = hA / (hA + hB);
    alpha[2] = -norm3f(nD) / (norm3f(nC) + norm3f(nD));
    alpha[3] = -norm3f(nC) / (norm3f(nC) + norm3f(nD));
}

void orc_force_initialize(orc_force *f, const double *x) {
    switch (f->kind) {
    case ADMM_KIND_ANCHOR: /* StaticAnchor::initialize, CORE/AnchorForce.cpp:31-35 */
        if (!f->moving) for (int j = 0; j < 3; ++j) f->pos[j] = x[3 * f->idx[0] + j];
        break;
    case ADMM_KIND_SPRING: { /* Spring::initialize, CORE/Force.cpp:29-38 */
        double d[3];
        for (int j = 0; j < 3; ++j) d[j] = x[3 * f->idx[0] + j] - x[3 * f->idx[1] + j];
        f->measure = norm3f(d);
        f->weight = sqrt(f->params[0]);
        break;
    }
    case ADMM_KIND_TET_LINEAR: case ADMM_KIND_TET_VOLUME: /* CORE/TetForce.cpp:112-117,160-163 */
        init_tet_force(f->idx, x, &f->measure, f->B);
        f->weight = sqrtf(f->params[0]) * sqrtf(f->measure);
        break;
    case ADMM_KIND_TET_NH: case ADMM_KIND_TET_STVK: { /* HyperElasticTet::initialize, CORE/TetForce.cpp:303-310 */
        init_tet_force(f->idx, x, &f->measure, f->B);
        double stiff = STD_MIN(f->params[0], f->params[1]);
        f->weight = sqrtf(stiff) * sqrtf(f->measure);
        break;
    }
    case ADMM_KIND_TRI_STRAIN: case ADMM_KIND_TRI_AREA: /* TriArea inherits LimitedTriangleStrain::initialize */
        init_tri_force(f->idx, x, &f->measure, f->B);
        f->weight = sqrtf(f->params[0]) * sqrtf(f->measure);
        break;
    case ADMM_KIND_TRI_FUNG: /* FungTriangle::initialize, CORE/TriangleForce.cpp:171-210: double sqrt, k = mu */
        init_tri_force(f->idx, x, &f->measure, f->B);
        f->weight = sqrt(f->params[0]) * sqrt(f->measure);
        break;
    case ADMM_KIND_BEND:
        f->weight = sqrt(f->params[0]);
        init_bend(f->idx, x, f->alpha);
        break;
    }
}

/* U * diag(s) * Vt, coefficient order ((p0+p1)+p2) with p_k = (U(i,k)*s_k)*Vt(k,j) */
static void recompose3(const double *U, const double *s, const double *Vt, double *out) {
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i)
        M3(out, i, j) = ((M3(U, i, 0) * s[0]) * M3(Vt, 0, j) + (M3(U, i, 1) * s[1]) * M3(Vt, 1, j)) + (M3(U, i, 2) * s[2]) * M3(Vt, 2, j);
}

#ifdef ORC_LS_TRACE
/* probe only (tools/probe/ls_predict.py): per tet [max|g0|, f0, evaluations, outer iterations, x0 min, s0 min] of the last project() */
static double *orc_ls_trace_buf;
static size_t orc_ls_trace_stride;      /* doubles per ADMM iteration (0: every iteration overwrites the last) */
static int orc_ls_trace_it;
void orc_set_ls_trace(double *buf, size_t stride) { orc_ls_trace_buf = buf; orc_ls_trace_stride = stride; }
#endif
/* HyperElasticTet::project, CORE/TetForce.cpp:320-364 */
static void project_hyper(orc_force *f, const double *Dx, double *u, double *z) {
    double F[9];
    for (int i = 0; i < 9; ++i) F[i] = Dx[i] + u[i];
    double S0[3], U[9], Vt[9];
    orc_oriented_svd(F, S0, U, Vt);
    prox3 P;
    P.type = (f->kind == ADMM_KIND_TET_STVK) ? 1 : 0;
    P.mu = f->params[0]; P.lambda = f->params[1]; P.k = STD_MIN(f->params[0], f->params[1]);
    P.s0[0] = S0[0]; P.s0[1] = S0[1]; P.s0[2] = S0[2]; P.n_fev = 0;
    double x2[3] = { f->state[0], f->state[1], f->state[2] };
    if (x2[2] < 0.0) x2[2] *= -1.0;
    else if (fabs(x2[0]) < 1.e-3 && fabs(x2[1]) < 1.e-3 && fabs(x2[2]) < 1.e-3) { x2[0] = 1.e-3; x2[1] = 1.e-3; x2[2] = 1.e-3; }
#ifdef ORC_LS_TRACE
    double tr_g[3], tr_f = prox_value(&P, x2), tr_x = STD_MIN(x2[0], STD_MIN(x2[1], x2[2]));
    prox_gradient(&P, x2, tr_g);
    P.n_fev = 0;
#endif
    f->n_iters = lbfgs_minimize(&P, x2, (int)f->params[2], 1e-8, &f->state[3]);
    f->n_fev = P.n_fev;
#ifdef ORC_LS_TRACE
    if (orc_ls_trace_buf) {
        double *t = orc_ls_trace_buf + orc_ls_trace_stride * orc_ls_trace_it + 6 * (size_t)(f->global_idx / 9);
        t[0] = absmaxn(tr_g, 3); t[1] = tr_f; t[2] = P.n_fev; t[3] = f->n_iters; t[4] = tr_x; t[5] = STD_MIN(S0[0], STD_MIN(S0[1], S0[2]));
    }
#endif
    f->state[0] = x2[0]; f->state[1] = x2[1]; f->state[2] = x2[2];
    double zi[9];
    recompose3(U, x2, Vt, zi);
    for (int i = 0; i < 9; ++i) { u[i] = u[i] + (Dx[i] - zi[i]); z[i] = zi[i]; }
}

/* LinearTetStrain::project, CORE/TetForce.cpp:127-153; TetVolume::project :173-210 */
static void project_tet_blend(orc_force *f, const double *Dx, double *u, double *z) {
    double d[9];
    for (int i = 0; i < 9; ++i) d[i] = Dx[i] + u[i];
    double U[9], S[3], V[9], Vt[9];
    orc_svd3(d, U, S, V);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M3(Vt, r, c) = M3(V, c, r);
    double Sn[3];
    if (f->kind == ADMM_KIND_TET_LINEAR) {
        Sn[0] = 1.0; Sn[1] = 1.0; Sn[2] = 1.0;
    } else {
        double dd[3] = {0, 0, 0};
        Sn[0] = S[0]; Sn[1] = S[1]; Sn[2] = S[2];
        for (int it = 0; it < 4; ++it) {
            double detS = Sn[0] * Sn[1] * Sn[2];
            double cl = STD_MAX(detS, f->params[1]); cl = STD_MIN(cl, f->params[2]);
            double ff = detS - cl;
            double g[3] = { Sn[1] * Sn[2], Sn[0] * Sn[2], Sn[0] * Sn[1] };
            double sc = -((ff - dot3f(g, dd)) / dot3f(g, g));
            for (int j = 0; j < 3; ++j) dd[j] = sc * g[j];
            for (int j = 0; j < 3; ++j) Sn[j] = S[j] + dd[j];
        }
    }
    if (det3(d) < 0.0) Sn[2] = -1.0;
    double p[9];
    recompose3(U, Sn, Vt, p);
    double k = f->params[0] * f->measure;
    double w2 = f->weight * f->weight;
    for (int i = 0; i < 9; ++i) {
        double zi = (k * p[i] + w2 * d[i]) / (w2 + k);
        u[i] = u[i] + (Dx[i] - zi); z[i] = zi;
    }
}

/* LimitedTriangleStrain::project, CORE/TriangleForce.cpp:78-113 */
static void project_tri(orc_force *f, const double *Dx, double *u, double *z) {
    double d[6];
    for (int i = 0; i < 6; ++i) d[i] = Dx[i] + u[i];
    double U[9], S[2], V[4];
    orc_svd32(d, U, S, V);
    double T[6];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 3; ++i)
        T[i + 3 * j] = U[i] * V[j] + U[i + 3] * V[j + 2];
    double k = f->params[0] * f->measure;
    double w2 = f->weight * f->weight;
    double zi[6];
    for (int i = 0; i < 6; ++i) zi[i] = (k * T[i] + w2 * d[i]) / (w2 + k);
    if (f->params[3] != 0.0) {
        double lmin = f->params[1], lmax = f->params[2];
        double l0 = norm3f(zi), l1 = norm3f(zi + 3);
        if (l0 < lmin) { double sc = lmin / fmaxf(l0, 1e-6); for (int i = 0; i < 3; ++i) zi[i] *= sc; }
        if (l1 < lmin) { double sc = lmin / fmaxf(l1, 1e-6); for (int i = 3; i < 6; ++i) zi[i] *= sc; }
        if (l0 > lmax) { double sc = lmax / fmaxf(l0, 1e-6); for (int i = 0; i < 3; ++i) zi[i] *= sc; }
        if (l1 > lmax) { double sc = lmax / fmaxf(l1, 1e-6); for (int i = 3; i < 6; ++i) zi[i] *= sc; }
    }
    for (int i = 0; i < 6; ++i) { u[i] = u[i] + (Dx[i] - zi[i]); z[i] = zi[i]; }
}

/* U(:, :2) * diag(s) * V^T for a 3x2 F: ((U*Diag)*V^T), the zero products of Diag add nothing */
static void recompose32(const double *U, const double *s, const double *V, double *out) {
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 3; ++i)
        out[i + 3 * j] = (U[i] * s[0]) * V[j] + (U[i + 3] * s[1]) * V[j + 2];
}

/* TriArea::project, CORE/TriangleForce.cpp:251-295 */
static void project_triarea(orc_force *f, const double *Dx, double *u, double *z) {
    double d[6];
    for (int i = 0; i < 6; ++i) d[i] = Dx[i] + u[i];
    double U[9], sv[2], V[4];
    orc_svd32(d, U, sv, V);
    double S[2] = { sv[0], sv[1] }, dd[2] = { 0.0, 0.0 };
    const int iters = (int)f->params[1];
    const double lmin = f->params[2], lmax = f->params[3];
    for (int i = 0; i < iters; ++i) {
        double v = S[0] * S[1];
        double c = (v < lmax ? v : lmax);
        c = (c > lmin ? c : lmin);
        double fv = v - c;
        double g[2] = { S[1], S[0] };
        double q = -((fv - (g[0] * dd[0] + g[1] * dd[1])) / (g[0] * g[0] + g[1] * g[1]));
        dd[0] = q * g[0]; dd[1] = q * g[1];
        S[0] = sv[0] + dd[0]; S[1] = sv[1] + dd[1];
    }
    double p[6];
    recompose32(U, S, V, p);
    double k = f->params[0] * f->measure;
    double w2 = f->weight * f->weight;
    for (int i = 0; i < 6; ++i) {
        double zi = (k * p[i] + w2 * d[i]) / (w2 + k);
        u[i] = u[i] + (Dx[i] - zi); z[i] = zi;
    }
}

/* FungTriangle::project, CORE/TriangleForce.cpp:227-249 */
static void project_fung(orc_force *f, const double *Dx, double *u, double *z) {
    double d[6];
    for (int i = 0; i < 6; ++i) d[i] = Dx[i] + u[i];
    double U[9], sv[2], V[4];
    orc_svd32(d, U, sv, V);
    prox3 P;
    P.type = 2; P.mu = f->params[0]; P.lambda = 0.0; P.k = f->params[0];
    P.s0[0] = sv[0]; P.s0[1] = sv[1]; P.s0[2] = 0.0; P.n_fev = 0;
    double x2[3] = { sv[0], sv[1], 0.0 };
    f->n_iters = lbfgs_minimize(&P, x2, 10, 1e-6, &f->state[3]);
    f->n_fev = P.n_fev;
    double zi[6];
    recompose32(U, x2, V, zi);
    for (int i = 0; i < 6; ++i) { u[i] = u[i] + (Dx[i] - zi[i]); z[i] = zi[i]; }
}

/* BendForce::project + computeUsingProjection, CORE/BendForce.cpp:131-161 */
static void project_bend(orc_force *f, const double *Dx, double *u, double *z) {
    double d[9], p[9];
    for (int i = 0; i < 9; ++i) d[i] = Dx[i] + u[i];
    const double *a = f->alpha;
    double den = a[0] * a[0] + a[3] * a[3] + a[1] * a[1];
    for (int j = 0; j < 3; ++j) {
        double lam = 2.0 * (a[0] * d[j] + a[3] * d[3 + j] + a[1] * d[6 + j]) / den;
        p[j] = d[j] - 0.5 * a[0] * lam;
        p[3 + j] = d[3 + j] - 0.5 * a[3] * lam;
        p[6 + j] = d[6 + j] - 0.5 * a[1] * lam;
    }
    double st = f->params[0], w2 = f->weight * f->weight;
    double c = 1.0 / (w2 + st);
    for (int i = 0; i < 9; ++i) {
        double zi = c * (st * p[i] + w2 * d[i]);
        u[i] = u[i] + (Dx[i] - zi); z[i] = zi;
    }
}

/* Spring::project, CORE/Force.cpp:52-71 */
static void project_spring(orc_force *f, const double *Dx, double *u, double *z) {
    double d[3], dn[3];
    for (int i = 0; i < 3; ++i) d[i] = Dx[i] + u[i];
    double n = norm3f(d);
    for (int i = 0; i < 3; ++i) dn[i] = d[i] / n;
    if (n <= 0.0) dn[0] = dn[1] = dn[2] = 0.0;
    double st = f->params[0], w2 = f->weight * f->weight;
    double c = 1.0 / (w2 + st);
    for (int i = 0; i < 3; ++i) {
        double p = f->measure * dn[i];
        double zi = c * (st * p + w2 * d[i]);
        u[i] = u[i] + (Dx[i] - zi); z[i] = zi;
    }
}

/* StaticAnchor::project :46-55, MovingAnchor::project :71-89 (CORE/AnchorForce.cpp) */
static void project_anchor(orc_force *f, const double *Dx, double *u, double *z) {
    double zi[3];
    if (!f->moving || f->active) { for (int i = 0; i < 3; ++i) zi[i] = f->pos[i]; }
    else { for (int i = 0; i < 3; ++i) { zi[i] = Dx[i] + u[i]; f->pos[i] = Dx[i]; } }
    for (int i = 0; i < 3; ++i) { u[i] = u[i] + (Dx[i] - zi[i]); z[i] = zi[i]; }
}

/* CollisionForce::project + handleCollisions, CORE/CollisionForce.cpp:38-70, with the shapes of
 * deps/admm-elastic-sca/src/collision/CollisionFloor.hpp:51-58, CollisionSphere.hpp:50-66,
 * CollisionCylinder.hpp:48-66 */
static int g_n_shapes = 0; static int g_shape_type[ADMM_MAX_SHAPES]; static double g_shape_par[ADMM_MAX_SHAPES][4];
static void project_collision(orc_force *f, const double *Dx, double *u, double *z) {
    (void)f;
    double p[3];
    for (int i = 0; i < 3; ++i) p[i] = Dx[i] + u[i];
    for (int j = 0; j < g_n_shapes; ++j) {
        const double *sp = g_shape_par[j];
        if (g_shape_type[j] == ADMM_SHAPE_FLOOR) {
            double err = sp[1] - p[1];
            if (err > 0) p[1] = sp[1];
        } else if (g_shape_type[j] == ADMM_SHAPE_SPHERE) {
            double d[3] = { p[0] - sp[0], p[1] - sp[1], p[2] - sp[2] };
            double err = sp[3] - norm3f(d);
            if (err > 0) { double n = norm3f(d); for (int i = 0; i < 3; ++i) p[i] = sp[i] + sp[3] * (d[i] / n); }
        } else {
            double c[3] = { sp[0], sp[1], 0.0 };
            double d[3] = { p[0] - c[0], p[1] - c[1], 0.0 - c[2] };
            double err = sp[3] - norm3f(d);
            if (err > 0) { double n = norm3f(d); double add[3] = { 0.0, 0.0, p[2] }; for (int i = 0; i < 3; ++i) p[i] = (c[i] + sp[3] * (d[i] / n)) + add[i]; }
        }
    }
    for (int i = 0; i < 3; ++i) { u[i] = u[i] + (Dx[i] - p[i]); z[i] = p[i]; }
}
void orc_set_collision_shapes(orc_system *s, int n, const int *types, const double *params) {
    (void)s;
    g_n_shapes = n > ADMM_MAX_SHAPES ? ADMM_MAX_SHAPES : n;
    for (int j = 0; j < g_n_shapes; ++j) { g_shape_type[j] = types[j]; for (int q = 0; q < 4; ++q) g_shape_par[j][q] = params[4 * j + q]; }
}

void orc_force_project(orc_force *f, double dt, const double *Dx, double *u, double *z) {
    (void)dt;
    switch (f->kind) {
    case ADMM_KIND_COLLISION: project_collision(f, Dx, u, z); break;
    case ADMM_KIND_ANCHOR: project_anchor(f, Dx, u, z); break;
    case ADMM_KIND_SPRING: project_spring(f, Dx, u, z); break;
    case ADMM_KIND_TET_LINEAR: case ADMM_KIND_TET_VOLUME: project_tet_blend(f, Dx, u, z); break;
    case ADMM_KIND_TET_NH: case ADMM_KIND_TET_STVK: project_hyper(f, Dx, u, z); break;
    case ADMM_KIND_TRI_STRAIN: project_tri(f, Dx, u, z); break;
    case ADMM_KIND_TRI_AREA: project_triarea(f, Dx, u, z); break;
    case ADMM_KIND_TRI_FUNG: project_fung(f, Dx, u, z); break;
    case ADMM_KIND_BEND: project_bend(f, Dx, u, z); break;
    }
}

/* ======================================================================= */
/* System, CORE/System.cpp                                                   */
/* ======================================================================= */
typedef struct { int r, c; double v; } trip;

struct orc_system {
    double dt; int admm_iters; int ref_layout;
    int dof;
    double *x, *v, *m;
    orc_force *forces; int n_forces, cap_forces;
    double grav[8][3]; int n_grav;
    int ex_type[8]; int *ex_idx[8]; int ex_n[8];
    /* assembled */
    trip *T; long nT, capT;
    double *W; int nW, capW;
    int *Dp, *Dj; double *Dv;        /* CSR of D (rows x dof) */
    double *Dx, *u, *z, *b, *xc;
    /* LDL^T of A (dof x dof) */
    int *Lp, *Li, *Parent; double *Lx, *Dg;
    int initialized;
    /* residuals of the last step (the reference only describes them, CORE/System.cpp:64-65):
     * r = W (Dx - z), s = D^T W^T W (z - z_prev); norms per ADMM iteration; tol > 0 ends the loop early */
    double res_r[256], res_s[256]; int res_n; double tol_r, tol_s; int track_res;
};

orc_system *orc_create(void) {
    orc_system *s = (orc_system *)calloc(1, sizeof *s);
    s->dt = 0.04; s->admm_iters = 10; s->ref_layout = 0;
    return s;
}
void orc_destroy(orc_system *s) {
    if (!s) return;
    free(s->x); free(s->v); free(s->m); free(s->forces); free(s->T); free(s->W);
    free(s->Dp); free(s->Dj); free(s->Dv); free(s->Dx); free(s->u); free(s->z); free(s->b); free(s->xc);
    free(s->Lp); free(s->Li); free(s->Parent); free(s->Lx); free(s->Dg);
    free(s);
}
void orc_settings(orc_system *s, double dt, int it) { s->dt = dt; s->admm_iters = it; }
void orc_set_layout(orc_system *s, int r) { s->ref_layout = r; }

/* System::add_nodes, CORE/System.cpp:78-95 */
int orc_add_nodes(orc_system *s, int n3, const double *x, const double *m) {
    int old = s->dof, tot = old + n3;
    s->x = (double *)realloc(s->x, sizeof(double) * tot);
    s->v = (double *)realloc(s->v, sizeof(double) * tot);
    s->m = (double *)realloc(s->m, sizeof(double) * tot);
    for (int i = 0; i < n3; ++i) { s->x[old + i] = x[i]; s->v[old + i] = 0.0; s->m[old + i] = m[i]; }
    s->dof = tot;
    return tot / 3;
}
static orc_force *push_force(orc_system *s) {
    if (s->n_forces == s->cap_forces) {
        s->cap_forces = s->cap_forces ? 2 * s->cap_forces : 1024;
        s->forces = (orc_force *)realloc(s->forces, sizeof(orc_force) * s->cap_forces);
    }
    return &s->forces[s->n_forces++];
}
int orc_add_forces(orc_system *s, int kind, int n, const int *idx, const double *params) {
    if (kind < 0 || kind >= ADMM_KIND_COUNT) return -1;
    int nn = ADMM_KIND_NODES[kind], np = ADMM_KIND_PARAMS[kind];
    for (int e = 0; e < n; ++e) orc_force_construct(push_force(s), kind, idx + (size_t)e * nn, params + (size_t)e * np);
    return s->n_forces;
}
int orc_add_moving_anchor(orc_system *s, int idx, const double *pos, int active, double use_weight) {
    double p[2] = { use_weight, 1.0 };
    orc_force *f = push_force(s);
    orc_force_construct(f, ADMM_KIND_ANCHOR, &idx, p);
    f->moving = 1; f->active = active;
    for (int j = 0; j < 3; ++j) f->pos[j] = pos[j];
    return s->n_forces - 1;
}
void orc_set_control_point(orc_system *s, int fi, const double *pos, int active) {
    orc_force *f = &s->forces[fi];
    for (int j = 0; j < 3; ++j) f->pos[j] = pos[j];
    f->active = active;
}
void orc_add_gravity(orc_system *s, double gx, double gy, double gz) {
    if (s->n_grav < 8) { s->grav[s->n_grav][0] = gx; s->grav[s->n_grav][1] = gy; s->grav[s->n_grav][2] = gz; s->ex_type[s->n_grav] = ADMM_EXPLICIT_CONST; s->ex_idx[s->n_grav] = NULL; s->ex_n[s->n_grav] = 0; s->n_grav++; }
}
void orc_add_explicit(orc_system *s, int type, const double *dir, int n_idx, const int *idx) {
    if (s->n_grav >= 8) return;
    const int k = s->n_grav++;
    for (int j = 0; j < 3; ++j) s->grav[k][j] = dir[j];
    s->ex_type[k] = type; s->ex_n[k] = n_idx;
    const int cnt = (type == ADMM_EXPLICIT_WIND ? 3 : 1) * n_idx;
    s->ex_idx[k] = cnt ? (int *)malloc(sizeof(int) * cnt) : NULL;
    for (int i = 0; i < cnt; ++i) s->ex_idx[k][i] = idx[i];
}
void orc_set_explicit_dir(orc_system *s, int which, const double *dir) { for (int j = 0; j < 3; ++j) s->grav[which][j] = dir[j]; }

/* WindForce::project, CORE/ExplicitForce.cpp:42-98, serial triangle order */
static void wind_project(const orc_system *s, int k, double dt) {
    const double *dirv = s->grav[k];
    double *v = s->v; const double *x = s->x;
    for (int t = 0; t < s->ex_n[k]; ++t) {
        int id[3] = { s->ex_idx[k][3 * t] * 3, s->ex_idx[k][3 * t + 1] * 3, s->ex_idx[k][3 * t + 2] * 3 };
        double cv[3], vr[3];
        for (int j = 0; j < 3; ++j) { cv[j] = (v[id[0] + j] + v[id[1] + j] + v[id[2] + j]) / 3.0; vr[j] = cv[j] - dirv[j]; }
        double a[3], b[3], n[3];
        for (int j = 0; j < 3; ++j) { a[j] = x[id[1] + j] - x[id[0] + j]; b[j] = x[id[2] + j] - x[id[0] + j]; }
        cross3(a, b, n);
        double nn = norm3f(n);
        double normal[3] = { n[0] / nn, n[1] / nn, n[2] / nn };
        double area = 0.5 * norm3f(n);
        double v_n = dot3f(normal, vr);
        double c = -1000.0 * area * v_n * fabs(v_n);
        for (int j = 0; j < 3; ++j) {
            double force = c * normal[j];
            force *= 0.33; force *= dt;
            v[id[0] + j] += force; v[id[1] + j] += force; v[id[2] + j] += force;
        }
    }
}

static void push_trip(orc_system *s, int r, int c, double v) {
    if (s->nT == s->capT) { s->capT = s->capT ? 2 * s->capT : 4096; s->T = (trip *)realloc(s->T, sizeof(trip) * s->capT); }
    s->T[s->nT].r = r; s->T[s->nT].c = c; s->T[s->nT].v = v; s->nT++;
}
static void push_w(orc_system *s, double w) {
    if (s->nW == s->capW) { s->capW = s->capW ? 2 * s->capW : 4096; s->W = (double *)realloc(s->W, sizeof(double) * s->capW); }
    s->W[s->nW++] = w;
}

/* Force::get_selector for every kind; ref_layout reproduces the reference's
 * row bookkeeping including init_tet_Di's constraint_idx = triplets.size()
 * (CORE/TetForce.cpp:59-77 vs :312-318). */
static void get_selector(orc_system *s, orc_force *f) {
    f->global_idx = s->nW;
    const int g = f->global_idx;
    switch (f->kind) {
    case ADMM_KIND_ANCHOR: /* CORE/AnchorForce.cpp:37-44,61-68 */
        for (int i = 0; i < 3; ++i) { push_w(s, f->weight); push_trip(s, g + i, 3 * f->idx[0] + i, 1.0); }
        break;
    case ADMM_KIND_COLLISION: /* CORE/CollisionForce.cpp:27-35 (one element per node) */
        for (int i = 0; i < 3; ++i) { push_trip(s, g + i, 3 * f->idx[0] + i, 1.0); push_w(s, f->weight); }
        break;
    case ADMM_KIND_SPRING: /* CORE/Force.cpp:40-50 */
        for (int i = 0; i < 3; ++i) { push_trip(s, i + g, 3 * f->idx[0] + i, 1.0); push_trip(s, i + g, 3 * f->idx[1] + i, -1.0); }
        for (int i = 0; i < 3; ++i) push_w(s, f->weight);
        break;
    case ADMM_KIND_TET_LINEAR: case ADMM_KIND_TET_VOLUME: case ADMM_KIND_TET_NH: case ADMM_KIND_TET_STVK: {
        int cidx = s->ref_layout ? (int)s->nT : g;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) {
            double val = f->B[c + 4 * r]; /* Bt(r,c) = B(c,r) */
            for (int j = 0; j < 3; ++j) push_trip(s, 3 * r + j + cidx, 3 * f->idx[c] + j, val);
        }
        if (s->ref_layout) { for (long i = g; i < s->nT; ++i) push_w(s, f->weight); }
        else for (int i = 0; i < 9; ++i) push_w(s, f->weight);
        break;
    }
    case ADMM_KIND_TRI_STRAIN: case ADMM_KIND_TRI_AREA: case ADMM_KIND_TRI_FUNG: /* CORE/TriangleForce.cpp:66-75, 213-223 */
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            push_trip(s, i + g, 3 * f->idx[j] + i, f->B[j + 3 * 0]);
            push_trip(s, 3 + i + g, 3 * f->idx[j] + i, f->B[j + 3 * 1]);
        }
        for (int i = 0; i < 6; ++i) push_w(s, f->weight);
        break;
    case ADMM_KIND_BEND: { /* CORE/BendForce.cpp:58-118: rows (x0-x2, x3-x2, x1-x2) */
        for (int i = 0; i < 9; ++i) push_w(s, f->weight);
        int plus[3] = { f->idx[0], f->idx[3], f->idx[1] };
        for (int b = 0; b < 3; ++b) {
            for (int i = 0; i < 3; ++i) push_trip(s, 3 * b + i + g, 3 * plus[b] + i, 1.0);
            for (int i = 0; i < 3; ++i) push_trip(s, 3 * b + i + g, 3 * f->idx[2] + i, -1.0);
        }
        break;
    }
    }
}

/* ---- sparse LDL^T (up-looking; the algorithm of T. Davis' LDL package, which
 * Eigen::SimplicialLDLT restates: EIG/SparseCholesky/SimplicialCholesky_impl.h).
 * Natural ordering (the reference uses AMD; the ordering changes rounding only). */
typedef struct { int i, k; double v; } atrip;
static int atrip_cmp(const void *a, const void *b) {
    const atrip *x = (const atrip *)a, *y = (const atrip *)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    return 0;
}
static int factorize(orc_system *s) {
    const int n = s->dof, R = s->nW;
    const double dt = s->dt;
    /* upper-triangular triplets of A = M + dt^2 D^T W W D  (CORE/System.cpp:138) */
    long cnt = n;
    for (int r = 0; r < R; ++r) { long k = s->Dp[r + 1] - s->Dp[r]; cnt += k * (k + 1) / 2; }
    atrip *A = (atrip *)malloc(sizeof(atrip) * cnt);
    long na = 0;
    for (int i = 0; i < n; ++i) { A[na].i = i; A[na].k = i; A[na].v = s->m[i]; na++; }
    for (int r = 0; r < R; ++r) {
        double w = s->W[r];
        for (int p = s->Dp[r]; p < s->Dp[r + 1]; ++p) for (int q = s->Dp[r]; q < s->Dp[r + 1]; ++q) {
            int i = s->Dj[p], k = s->Dj[q];
            if (i > k) continue;
            A[na].i = i; A[na].k = k; A[na].v = (((dt * dt) * s->Dv[p]) * w) * w * s->Dv[q]; na++;
        }
    }
    qsort(A, na, sizeof(atrip), atrip_cmp);
    int *Ap = (int *)calloc(n + 1, sizeof(int)); int *Ai = (int *)malloc(sizeof(int) * na); double *Ax = (double *)malloc(sizeof(double) * na);
    long nz = 0;
    for (long t = 0; t < na;) {
        long e = t; double sum = 0.0;
        while (e < na && A[e].i == A[t].i && A[e].k == A[t].k) { sum += A[e].v; e++; }
        Ai[nz] = A[t].i; Ax[nz] = sum; Ap[A[t].k + 1]++; nz++;
        t = e;
    }
    free(A);
    for (int k = 0; k < n; ++k) Ap[k + 1] += Ap[k];
    /* symbolic */
    int *Parent = (int *)malloc(sizeof(int) * n), *Lnz = (int *)calloc(n, sizeof(int)), *Flag = (int *)malloc(sizeof(int) * n);
    for (int k = 0; k < n; ++k) {
        Parent[k] = -1; Flag[k] = k;
        for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
            int i = Ai[p];
            if (i < k) for (; Flag[i] != k; i = Parent[i]) { if (Parent[i] == -1) Parent[i] = k; Lnz[i]++; Flag[i] = k; }
        }
    }
    int *Lp = (int *)malloc(sizeof(int) * (n + 1));
    Lp[0] = 0;
    for (int k = 0; k < n; ++k) Lp[k + 1] = Lp[k] + Lnz[k];
    int *Li = (int *)malloc(sizeof(int) * (Lp[n] > 0 ? Lp[n] : 1)); double *Lx = (double *)malloc(sizeof(double) * (Lp[n] > 0 ? Lp[n] : 1));
    double *Dg = (double *)malloc(sizeof(double) * n), *Y = (double *)calloc(n, sizeof(double));
    int *Pattern = (int *)malloc(sizeof(int) * n);
    int ok = 1;
    for (int k = 0; k < n; ++k) {
        int top = n; Flag[k] = k; Lnz[k] = 0; Y[k] = 0.0;
        for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
            int i = Ai[p];
            if (i <= k) {
                Y[i] += Ax[p];
                int len;
                for (len = 0; Flag[i] != k; i = Parent[i]) { Pattern[len++] = i; Flag[i] = k; }
                while (len > 0) Pattern[--top] = Pattern[--len];
            }
        }
        Dg[k] = Y[k]; Y[k] = 0.0;
        for (; top < n; ++top) {
            int i = Pattern[top]; double yi = Y[i]; Y[i] = 0.0;
            int p2 = Lp[i] + Lnz[i], p;
            for (p = Lp[i]; p < p2; ++p) Y[Li[p]] -= Lx[p] * yi;
            double l_ki = yi / Dg[i];
            Dg[k] -= l_ki * yi;
            Li[p] = k; Lx[p] = l_ki; Lnz[i]++;
        }
        if (Dg[k] == 0.0) { ok = 0; break; }
    }
    free(Ap); free(Ai); free(Ax); free(Lnz); free(Flag); free(Y); free(Pattern);
    free(s->Lp); free(s->Li); free(s->Lx); free(s->Dg); free(s->Parent);
    s->Lp = Lp; s->Li = Li; s->Lx = Lx; s->Dg = Dg; s->Parent = Parent;
    return ok;
}
/* SimplicialCholeskyBase::_solve, EIG/SparseCholesky/SimplicialCholesky.h:153-177 */
static void ldl_solve(const orc_system *s, double *b) {
    const int n = s->dof;
    for (int j = 0; j < n; ++j) { double bj = b[j]; for (int p = s->Lp[j]; p < s->Lp[j + 1]; ++p) b[s->Li[p]] -= s->Lx[p] * bj; }
    for (int j = 0; j < n; ++j) b[j] /= s->Dg[j];
    for (int j = n - 1; j >= 0; --j) { double bj = b[j]; for (int p = s->Lp[j]; p < s->Lp[j + 1]; ++p) bj -= s->Lx[p] * b[s->Li[p]]; b[j] = bj; }
}

/* System::initialize, CORE/System.cpp:98-156 */
int orc_initialize(orc_system *s) {
    if (s->dt <= 0.0) s->dt = 0.04;
    if (s->dof < 3) return 0;
    for (int i = 0; i < s->dof; ++i) s->v[i] = 0.0;
    for (int i = 0; i < s->n_forces; ++i) orc_force_initialize(&s->forces[i], s->x);
    s->nT = 0; s->nW = 0;
    for (int i = 0; i < s->n_forces; ++i) get_selector(s, &s->forces[i]);
    const int R = s->nW;
    for (long t = 0; t < s->nT; ++t) if (s->T[t].r < 0 || s->T[t].r >= R || s->T[t].c < 0 || s->T[t].c >= s->dof) return 0;
    /* CSR of D, duplicates kept in push order within a row */
    free(s->Dp); free(s->Dj); free(s->Dv);
    s->Dp = (int *)calloc(R + 2, sizeof(int)); s->Dj = (int *)malloc(sizeof(int) * (s->nT ? s->nT : 1)); s->Dv = (double *)malloc(sizeof(double) * (s->nT ? s->nT : 1));
    for (long t = 0; t < s->nT; ++t) s->Dp[s->T[t].r + 2]++;
    for (int r = 0; r < R; ++r) s->Dp[r + 2] += s->Dp[r + 1];
    for (long t = 0; t < s->nT; ++t) { int p = s->Dp[s->T[t].r + 1]++; s->Dj[p] = s->T[t].c; s->Dv[p] = s->T[t].v; }
    free(s->Dx); free(s->u); free(s->z); free(s->b); free(s->xc);
    s->Dx = (double *)calloc(R + 1, sizeof(double)); s->u = (double *)calloc(R + 1, sizeof(double)); s->z = (double *)calloc(R + 1, sizeof(double));
    s->b = (double *)calloc(s->dof, sizeof(double)); s->xc = (double *)calloc(s->dof, sizeof(double));
    if (!factorize(s)) return 0;
    s->initialized = 1;
    return 1;
}

static void spmv_D(const orc_system *s, const double *x, double *y) {
    for (int r = 0; r < s->nW; ++r) {
        double acc = 0.0;
        for (int p = s->Dp[r]; p < s->Dp[r + 1]; ++p) acc += s->Dv[p] * x[s->Dj[p]];
        y[r] = acc;
    }
}

/* System::step, CORE/System.cpp:26-75 */
int orc_step(orc_system *s) {
    const int n = s->dof, R = s->nW; const double dt = s->dt;
    /* ExplicitForce::project, CORE/ExplicitForce.cpp:29-39 */
    for (int gI = 0; gI < s->n_grav; ++gI) {
        if (s->ex_type[gI] == ADMM_EXPLICIT_WIND) { wind_project(s, gI, dt); continue; }
        if (s->ex_n[gI] == 0) { for (int i = 0; i < n / 3; ++i) for (int j = 0; j < 3; ++j) s->v[3 * i + j] += (dt * s->grav[gI][j]); }
        else for (int q = 0; q < s->ex_n[gI]; ++q) for (int j = 0; j < 3; ++j) s->v[3 * s->ex_idx[gI][q] + j] += (dt * s->grav[gI][j]);
    }
    spmv_D(s, s->x, s->z);                                  /* curr_z = D*m_x          :43 */
    double *M_xbar = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) { double xb = s->x[i] + dt * s->v[i]; M_xbar[i] = s->m[i] * xb; s->xc[i] = xb; } /* :46-48 */
    double *zprev = NULL, *sv = NULL;
    if (s->track_res) { zprev = (double *)malloc(sizeof(double) * R); sv = (double *)malloc(sizeof(double) * n); }
    s->res_n = 0;
    for (int it = 0; it < s->admm_iters; ++it) {
#ifdef ORC_LS_TRACE
        orc_ls_trace_it = it;
#endif
        if (s->track_res) memcpy(zprev, s->z, sizeof(double) * R);
        spmv_D(s, s->xc, s->Dx);                            /* Dx = D*curr_x           :54 */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < s->n_forces; ++i) {             /* local step              :57-58 */
            orc_force *f = &s->forces[i];
            orc_force_project(f, dt, s->Dx + f->global_idx, s->u + f->global_idx, s->z + f->global_idx);
        }
        /* b = M_xbar + (dt^2 D^T W W)(z - u)               :61 ; column-major product = rows of D ascending */
        for (int i = 0; i < n; ++i) s->b[i] = 0.0;
        for (int r = 0; r < R; ++r) {
            double rhs = s->z[r] - s->u[r], w = s->W[r];
            for (int p = s->Dp[r]; p < s->Dp[r + 1]; ++p) s->b[s->Dj[p]] += ((((dt * dt) * s->Dv[p]) * w) * w) * rhs;
        }
        for (int i = 0; i < n; ++i) s->b[i] = M_xbar[i] + s->b[i];
        ldl_solve(s, s->b);                                 /* curr_x = solver.solve   :62 */
        for (int i = 0; i < n; ++i) s->xc[i] = s->b[i];
        if (s->track_res) {                                 /* :64-65 (comment in the reference) */
            double rr = 0.0, ss = 0.0;
            for (int i = 0; i < n; ++i) sv[i] = 0.0;
            for (int r = 0; r < R; ++r) {
                double w = s->W[r], pr = w * (s->Dx[r] - s->z[r]);
                rr += pr * pr;
                double q = w * w * (s->z[r] - zprev[r]);
                for (int p = s->Dp[r]; p < s->Dp[r + 1]; ++p) sv[s->Dj[p]] += s->Dv[p] * q;
            }
            for (int i = 0; i < n; ++i) ss += sv[i] * sv[i];
            if (s->res_n < 256) { s->res_r[s->res_n] = sqrt(rr); s->res_s[s->res_n] = sqrt(ss); s->res_n++; }
            if (s->tol_r > 0.0 && sqrt(rr) <= s->tol_r && sqrt(ss) <= s->tol_s) break;
        }
    }
    free(zprev); free(sv);
    for (int i = 0; i < n; ++i) { s->v[i] = (s->xc[i] - s->x[i]) * (1.0 / dt); s->x[i] = s->xc[i]; } /* :70-71 */
    free(M_xbar);
    return 1;
}

void orc_track_residuals(orc_system *s, int on, double tol_r, double tol_s) { s->track_res = on; s->tol_r = tol_r; s->tol_s = tol_s; }
int orc_get_residuals(orc_system *s, double *r, double *sd, int cap) {
    int n = s->res_n < cap ? s->res_n : cap;
    for (int i = 0; i < n; ++i) { r[i] = s->res_r[i]; sd[i] = s->res_s[i]; }
    return s->res_n;
}
int orc_dof(orc_system *s) { return s->dof; }
int orc_rows(orc_system *s) { return s->nW; }
int orc_n_forces(orc_system *s) { return s->n_forces; }
orc_force *orc_get_force(orc_system *s, int i) { return &s->forces[i]; }
double *orc_x(orc_system *s) { return s->x; }
double *orc_v(orc_system *s) { return s->v; }
double *orc_u(orc_system *s) { return s->u; }
double *orc_z(orc_system *s) { return s->z; }
double *orc_wdiag(orc_system *s) { return s->W; }
long orc_D_nnz(orc_system *s) { return s->nT; }
void orc_get_D(orc_system *s, int *rows, int *cols, double *vals) {
    for (long t = 0; t < s->nT; ++t) { rows[t] = s->T[t].r; cols[t] = s->T[t].c; vals[t] = s->T[t].v; }
}
/* x = solver.solve(b) on a caller-supplied right-hand side (System.cpp:62) */
void orc_solve(orc_system *s, const double *b, double *x) {
    for (int i = 0; i < s->dof; ++i) x[i] = b[i];
    ldl_solve(s, x);
}
long orc_L_nnz(orc_system *s) { return s->Lp ? s->Lp[s->dof] : 0; }

void orc_set_omp_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_omp_threads(void) { return omp_get_max_threads(); }

double orc_time_steps(orc_system *s, int frames) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int f = 0; f < frames; ++f) orc_step(s);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
