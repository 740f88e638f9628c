// TEST-ONLY: compiles the product's per-element device math
// (admm-elastic-sca_amd/csrc/local_math.hpp) for the HOST so that the CPU
// test-suite can compare it against the oracle without a GPU.  This library is
// never loaded by the product; the product path runs the same header inside
// the HIP kernels only.
static long g_cstep_stats[9];   // [0] modified-function calls, [case] / [4 + case]: cstep case reached without / with a bracket
#define ADMM_CSTEP_STATS g_cstep_stats
#include "../admm-elastic-sca_amd/csrc/local_math.hpp"
using namespace admm_dev;
static Mat3 ld(const double *f) { Mat3 m; m.m00=f[0]; m.m10=f[1]; m.m20=f[2]; m.m01=f[3]; m.m11=f[4]; m.m21=f[5]; m.m02=f[6]; m.m12=f[7]; m.m22=f[8]; return m; }
static void st(const Mat3 &m, double *f) { f[0]=m.m00; f[1]=m.m10; f[2]=m.m20; f[3]=m.m01; f[4]=m.m11; f[5]=m.m21; f[6]=m.m02; f[7]=m.m12; f[8]=m.m22; }
extern "C" {
void hm_cstep_stats(long *out) { for (int i = 0; i < 9; ++i) out[i] = g_cstep_stats[i]; }
void hm_libm_log(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = log(x[i]); }   // this host's libm: what the reference calls
void hm_libm_exp(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = exp(x[i]); }
void hm_exp(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = admm_exp(x[i]); }
void hm_log(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = admm_log(x[i]); }
void hm_svd3(const double *F, double *U, double *S, double *V) {
    Mat3 u, v; svd3(ld(F), u, S[0], S[1], S[2], v); st(u, U); st(v, V);
}
int hm_project_hyper(int type, int M, const double *F, double mu, double lambda, int maxIter, double *state, double *z) {
    int it = 0; Mat3 r;
    if (type == 0) { if (M == 5) r = project_hyper<0, 5>(ld(F), mu, lambda, maxIter, state[0], state[1], state[2], state[3], it); else r = project_hyper<0, 10>(ld(F), mu, lambda, maxIter, state[0], state[1], state[2], state[3], it); }
    else { if (M == 5) r = project_hyper<1, 5>(ld(F), mu, lambda, maxIter, state[0], state[1], state[2], state[3], it); else r = project_hyper<1, 10>(ld(F), mu, lambda, maxIter, state[0], state[1], state[2], state[3], it); }
    st(r, z); return it;
}
void hm_project_tet_p(int volume, const double *d, double lmin, double lmax, double *p) {
    Mat3 r = volume ? project_tet_p<true>(ld(d), lmin, lmax) : project_tet_p<false>(ld(d), lmin, lmax); st(r, p);
}
// the step selection two ways: the select form (what mixed waves run) and the one-case form (what case-uniform waves run).
// in: n x [stx fx dx sty fy dy stp fp dp stpmin stpmax brackt]; out_*: n x [stx fx dx sty fy dy stp brackt info]; any_brackt as a wave
// with (1) / without (0) another bracketing lane would pass it in case 4
void hm_cstep_forms(int n, const double *in, int others_bracket, double *out_select, double *out_case) {
    for (int i = 0; i < n; ++i) {
        const double *a = in + 12 * i;
        for (int form = 0; form < 2; ++form) {
            double stx = a[0], fx = a[1], dx = a[2], sty = a[3], fy = a[4], dy = a[5], stp = a[6];
            const double fp = a[7], dp = a[8], lo = a[9], hi = a[10];
            bool brackt = a[11] != 0.0; int info = 0;
            if (form == 0) mt_cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, lo, hi, info);
            else if (!((brackt & ((stp <= smin(stx, sty)) | (stp >= smax(stx, sty)))) | (dx * (stp - stx) >= 0.0) | (hi < lo))) {
                const double sgnd = dp * unit_sign(dx);
                const bool any = brackt || others_bracket;
                if (fp > fx) { info = 1; mt_cstep_case<1>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, lo, hi, any); }
                else if (sgnd < 0.0) { info = 2; mt_cstep_case<2>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, lo, hi, any); }
                else if (fabs(dp) < fabs(dx)) { info = 3; mt_cstep_case<3>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, lo, hi, any); }
                else { info = 4; mt_cstep_case<4>(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, lo, hi, any); }
            }
            double *o = (form ? out_case : out_select) + 9 * i;
            o[0] = stx; o[1] = fx; o[2] = dx; o[3] = sty; o[4] = fy; o[5] = dy; o[6] = stp; o[7] = brackt; o[8] = info;
        }
    }
}
void hm_svd32(const double *F, double *U2, double *S, double *V) { svd32(F, U2, S[0], S[1], V); }
void hm_project_triarea_p(const double *d, int iters, double lmin, double lmax, double *p) { project_triarea_p(d, iters, lmin, lmax, p); }
int hm_project_fung(const double *d, double mu, double *hess, double *z) { int it = 0; project_fung(d, mu, *hess, it, z); return it; }
}
