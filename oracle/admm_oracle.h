/*
 * admm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's ADMM hot path
 * (mattoverby/admm-elastic-sca, deps/admm-elastic-sca/src/system/, the
 * vendored cppoptlib L-BFGS / More-Thuente and Eigen 3.2.5 JacobiSVD),
 * function by function, each citing the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (admm-elastic-sca_amd/) never does.
 *
 * Pinning: tests/test_oracle_vs_ref.py compares every function here against
 * the compiled reference (oracle/_ref/libadmm_ref.so, built by oracle/Makefile
 * from /root/reference) in this container, and tests/golden/ holds vectors
 * generated from that compiled reference (tests/golden/make_golden.py) plus
 * the reference's own two known answers (singletet 171.57142857142716,
 * singlenode -9.8/-29.4/-58.8/-98).
 */
#ifndef ADMM_ORACLE_H
#define ADMM_ORACLE_H

#include "../include/admm_kinds.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 3x3 / 3x2 SVD (Eigen JacobiSVD restatement); matrices col-major ---- */
void orc_svd3(const double F[9], double U[9], double S[3], double V[9]);
void orc_oriented_svd(const double F[9], double S[3], double U[9], double Vt[9]);
void orc_svd32(const double F[6], double U[9], double S[2], double V[4]);

/* ---- one force element ---- */
typedef struct orc_force {
    int kind;
    int idx[4];
    double params[4];     /* ADMM_KIND_PARAMS[kind] entries, see admm_kinds.h */
    /* computed by orc_force_initialize */
    double weight;
    double B[12];         /* tets: 4x3 col-major; tris: 3x2 col-major (first 6) */
    double measure;       /* tet volume / tri area / spring rest length */
    double alpha[4];      /* bend */
    double pos[3];        /* anchor target */
    int    active;        /* moving anchor: control point active */
    int    moving;        /* 1 = MovingAnchor semantics (pos is the control point) */
    /* warm-start state (HyperElasticTet): last_prox_result[3], init_hess */
    double state[4];
    int    n_iters;       /* L-BFGS outer iterations of the last project() */
    int    n_fev;         /* objective evaluations of the last project() (diagnostic) */
    /* set by get_selector */
    int global_idx;       /* first row in D/u/z */
} orc_force;

void orc_force_construct(orc_force *f, int kind, const int *idx, const double *params);
void orc_force_initialize(orc_force *f, const double *x);
/* project on the element's own rows: Dx, u, z point at the element's first row */
void orc_force_project(orc_force *f, double dt, const double *Dx, double *u, double *z);

/* ---- system ---- */
typedef struct orc_system orc_system;

orc_system *orc_create(void);
void orc_destroy(orc_system *s);
void orc_settings(orc_system *s, double dt, int admm_iters);
/* ref_layout = 1 reproduces the reference's row numbering bit-exactly (36 rows
 * per tet, TetForce.cpp:61 vs :313-317); 0 = compact rows (9 per tet). */
void orc_set_layout(orc_system *s, int ref_layout);
int  orc_add_nodes(orc_system *s, int n3, const double *x, const double *m);
int  orc_add_forces(orc_system *s, int kind, int n, const int *idx, const double *params);
int  orc_add_moving_anchor(orc_system *s, int idx, const double *pos, int active, double use_weight);
void orc_set_control_point(orc_system *s, int force_index, const double *pos, int active);
void orc_add_gravity(orc_system *s, double gx, double gy, double gz);
/* ExplicitForce with an index subset / WindForce over triangles (ExplicitForce.cpp:29-98).
 * Wind is evaluated in the reference's SERIAL order (triangle i sees the velocity updates of
 * triangles < i), i.e. what the reference computes with OMP_NUM_THREADS=1. */
void orc_add_explicit(orc_system *s, int type, const double *dir, int n_idx, const int *idx);
void orc_set_explicit_dir(orc_system *s, int which, const double *dir);
/* CollisionForce's shape table (CollisionForce.cpp:55-70) */
void orc_set_collision_shapes(orc_system *s, int n, const int *types, const double *params);
int  orc_initialize(orc_system *s);
int  orc_step(orc_system *s);
/* Residual norms per ADMM iteration of the last orc_step, as the comment at CORE/System.cpp:64-65 defines them:
 * r = W (Dx - z) with the Dx the local step used, s = D^T W^T W (z - z_prev).  tol_r > 0: the ADMM loop of a
 * step ends as soon as |r| <= tol_r and |s| <= tol_s.  (An extension: the reference never computes them.) */
void orc_track_residuals(orc_system *s, int on, double tol_r, double tol_s);
int  orc_get_residuals(orc_system *s, double *r, double *sdual, int cap);
int  orc_dof(orc_system *s);
int  orc_rows(orc_system *s);
int  orc_n_forces(orc_system *s);
orc_force *orc_get_force(orc_system *s, int i);
double *orc_x(orc_system *s);
double *orc_v(orc_system *s);
double *orc_u(orc_system *s);
double *orc_z(orc_system *s);
double *orc_wdiag(orc_system *s);
long orc_D_nnz(orc_system *s);
void orc_get_D(orc_system *s, int *rows, int *cols, double *vals); /* in push order */
void orc_solve(orc_system *s, const double *b, double *x);   /* x = A^-1 b with the LDL^T of orc_initialize (System.cpp:62) */
long orc_L_nnz(orc_system *s);
void orc_set_omp_threads(int n);   /* team size of the local-step loop (bench.py cpu_baseline tries a few) */
int orc_omp_threads(void);
double orc_time_steps(orc_system *s, int frames);

#ifdef __cplusplus
}
#endif
#endif
