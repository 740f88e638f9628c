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
void hm_svd32(const double *F, double *U2, double *S, double *V) { svd32(F, U2, S[0], S[1], V); }
void hm_project_triarea_p(const double *d, int iters, double lmin, double lmax, double *p) { project_triarea_p(d, iters, lmin, lmax, p); }
int hm_project_fung(const double *d, double mu, double *hess, double *z) { int it = 0; project_fung(d, mu, *hess, it, z); return it; }
}
