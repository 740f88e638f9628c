// ref_shim.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A thin extern "C" wrapper around the *real* reference (admm::System and the
// admm::Force subclasses), compiled by oracle/Makefile from the sources where
// they lie under /root/reference into oracle/_ref/libadmm_ref.so.  Nothing of
// the reference is copied into this repository: this file only #includes the
// reference's public headers at build time and forwards calls.
//
// Used by
//   * tests/ (CPU, this container): to pin oracle/admm_oracle.c against the
//     reference and to generate the golden fixtures under tests/golden/
//     (tests/golden/make_golden.py);
//   * bench.py's cpu_baseline leg ("kind": "reference"): the prebuilt .so
//     travels to the GPU box; /root/reference itself does not.
//
// The wrapped calls are exactly the reference's own API
// (deps/admm-elastic-sca/src/system/System.hpp:29-76, Force.hpp:37-57).

#include "System.hpp"
#include "AnchorForce.hpp"
#include "TetForce.hpp"
#include "TriangleForce.hpp"
#include "BendForce.hpp"
#include "ExplicitForce.hpp"
#include "CollisionForce.hpp"
#include "CollisionFloor.hpp"
#include "CollisionSphere.hpp"
#include "CollisionCylinder.hpp"
#include <Eigen/SVD>
#include <chrono>
#include <cstring>
#include <omp.h>

#include "../include/admm_kinds.h"

using namespace admm;
using Eigen::VectorXd;

namespace {

// Derived-class accessor: reaches the protected ADMM state without touching
// the reference sources (System.hpp:78-101).
struct RefSystem : public System {
    Eigen::SparseMatrix<double> &D() { return m_D; }
    VectorXd &W() { return m_W_diag; }
    VectorXd &U() { return curr_u; }
    VectorXd &Z() { return curr_z; }
    Eigen::SimplicialLDLT<Eigen::SparseMatrix<double> > &LDLT() { return solver; }
    std::vector<std::shared_ptr<ControlPoint> > points; // for moving anchors
};

std::shared_ptr<Force> make_force(RefSystem *sys, int kind, const int *idx, const double *p) {
    switch (kind) {
    case ADMM_KIND_ANCHOR: {
        // p[0] = use_weight, p[1] = moving? (>=0: control point active flag) / <0 static
        return std::shared_ptr<Force>(new StaticAnchor(idx[0], p[0]));
    }
    case ADMM_KIND_SPRING: return std::shared_ptr<Force>(new Spring(idx[0], idx[1], p[0]));
    case ADMM_KIND_TET_LINEAR: return std::shared_ptr<Force>(new LinearTetStrain(idx[0], idx[1], idx[2], idx[3], p[0]));
    case ADMM_KIND_TET_VOLUME: return std::shared_ptr<Force>(new TetVolume(idx[0], idx[1], idx[2], idx[3], p[0], p[1], p[2]));
    case ADMM_KIND_TET_NH: return std::shared_ptr<Force>(new HyperElasticTet(idx[0], idx[1], idx[2], idx[3], p[0], p[1], (int)p[2], "nh"));
    case ADMM_KIND_TET_STVK: return std::shared_ptr<Force>(new HyperElasticTet(idx[0], idx[1], idx[2], idx[3], p[0], p[1], (int)p[2], "stvk"));
    case ADMM_KIND_TRI_STRAIN: return std::shared_ptr<Force>(new LimitedTriangleStrain(idx[0], idx[1], idx[2], p[0], p[1], p[2], p[3] != 0.0));
    case ADMM_KIND_BEND: return std::shared_ptr<Force>(new BendForce(idx[0], idx[1], idx[2], idx[3], p[0]));
    case ADMM_KIND_TRI_AREA: return std::shared_ptr<Force>(new TriArea(idx[0], idx[1], idx[2], p[0], (int)p[1], p[2], p[3]));
    case ADMM_KIND_TRI_FUNG: return std::shared_ptr<Force>(new FungTriangle(idx[0], idx[1], idx[2], p[0], p[1], p[2]));
    }
    return std::shared_ptr<Force>();
}

} // namespace

extern "C" {

void *ref_create() { return new RefSystem(); }
void ref_destroy(void *h) { delete (RefSystem *)h; }

void ref_settings(void *h, double dt, int iters, int verbose) {
    RefSystem *s = (RefSystem *)h;
    s->settings.timestep_s = dt; s->settings.admm_iters = iters; s->settings.verbose = verbose;
}

// n3 = number of doubles (3 per node); returns total node count (System.cpp:78-95)
int ref_add_nodes(void *h, int n3, const double *x, const double *m) {
    RefSystem *s = (RefSystem *)h;
    VectorXd xv = Eigen::Map<const VectorXd>(x, n3), mv = Eigen::Map<const VectorXd>(m, n3);
    return s->add_nodes(xv, mv);
}

// Adds n elements of one kind in order; idx is [n][nodes], params [n][ADMM_KIND_PARAMS[kind]].
int ref_add_forces(void *h, int kind, int n, const int *idx, const double *params) {
    RefSystem *s = (RefSystem *)h;
    const int nn = ADMM_KIND_NODES[kind], np = ADMM_KIND_PARAMS[kind];
    for (int e = 0; e < n; ++e) {
        std::shared_ptr<Force> f = make_force(s, kind, idx + (size_t)e * nn, params + (size_t)e * np);
        if (!f) return -1;
        s->forces.push_back(f);
    }
    return (int)s->forces.size();
}

// MovingAnchor (AnchorForce.hpp:86-104): returns the control-point handle index
int ref_add_moving_anchor(void *h, int idx, const double *pos, int active, double use_weight) {
    RefSystem *s = (RefSystem *)h;
    std::shared_ptr<ControlPoint> p(new ControlPoint(Eigen::Vector3d(pos[0], pos[1], pos[2])));
    p->active = active != 0;
    s->points.push_back(p);
    s->forces.push_back(std::shared_ptr<Force>(new MovingAnchor(idx, p, use_weight)));
    return (int)s->points.size() - 1;
}
void ref_set_control_point(void *h, int cp, const double *pos, int active) {
    RefSystem *s = (RefSystem *)h;
    s->points[cp]->pos = Eigen::Vector3d(pos[0], pos[1], pos[2]);
    s->points[cp]->active = active != 0;
}
void ref_get_control_point(void *h, int cp, double *pos) {
    RefSystem *s = (RefSystem *)h;
    for (int j = 0; j < 3; ++j) pos[j] = s->points[cp]->pos[j];
}

void ref_add_gravity(void *h, double gx, double gy, double gz) {
    RefSystem *s = (RefSystem *)h;
    s->explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new ExplicitForce(Eigen::Vector3d(gx, gy, gz))));
}
// WindForce over n_tris triangles (ExplicitForce.cpp:42-98)
void ref_add_wind(void *h, int n_tris, const int *tris, double dx, double dy, double dz) {
    RefSystem *s = (RefSystem *)h;
    std::vector<int> t(tris, tris + 3 * (size_t)n_tris);
    std::shared_ptr<WindForce> w(new WindForce(t));
    w->direction = Eigen::Vector3d(dx, dy, dz);
    s->explicit_forces.push_back(w);
}

// ExplicitForce on an index subset (ExplicitForce.hpp:56)
void ref_add_explicit_subset(void *h, int n, const int *idx, double gx, double gy, double gz) {
    RefSystem *s = (RefSystem *)h;
    std::vector<int> id(idx, idx + n);
    s->explicit_forces.push_back(std::shared_ptr<ExplicitForce>(new ExplicitForce(Eigen::Vector3d(gx, gy, gz), id)));
}
void ref_set_explicit_dir(void *h, int which, double gx, double gy, double gz) {
    ((RefSystem *)h)->explicit_forces[which]->direction = Eigen::Vector3d(gx, gy, gz);
}
// CollisionForce over all nodes with analytic shapes (CollisionForce.hpp:33; plinkopony.cpp:53-96)
int ref_add_collision(void *h, int n_shapes, const int *types, const double *params, double use_weight) {
    RefSystem *s = (RefSystem *)h;
    std::vector<std::shared_ptr<CollisionShape> > shapes;
    for (int j = 0; j < n_shapes; ++j) {
        const double *p = params + 4 * j;
        if (types[j] == ADMM_SHAPE_FLOOR) shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionFloor(Eigen::Vector3d(p[0], p[1], p[2]))));
        else if (types[j] == ADMM_SHAPE_SPHERE) shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionSphere(Eigen::Vector3d(p[0], p[1], p[2]), p[3])));
        else shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionCylinder(Eigen::Vector3d(p[0], p[1], p[2]), Eigen::Vector3d(1, 1, 1), p[3])));
    }
    s->forces.push_back(std::shared_ptr<Force>(new CollisionForce(shapes, use_weight)));
    return (int)s->forces.size();
}

int ref_initialize(void *h) { return ((RefSystem *)h)->initialize() ? 1 : 0; }
int ref_step(void *h) { return ((RefSystem *)h)->step() ? 1 : 0; }
void ref_recompute_weights(void *h) { ((RefSystem *)h)->recompute_weights(); }
void ref_set_force_weight(void *h, int i, double w) { ((RefSystem *)h)->forces[i]->weight = w; }

int ref_dof(void *h) { return (int)((RefSystem *)h)->m_x.size(); }
int ref_rows(void *h) { return (int)((RefSystem *)h)->U().size(); }
int ref_n_forces(void *h) { return (int)((RefSystem *)h)->forces.size(); }
double ref_elapsed(void *h) { return ((RefSystem *)h)->elapsed_s; }

void ref_get_x(void *h, double *x) { RefSystem *s = (RefSystem *)h; std::memcpy(x, s->m_x.data(), sizeof(double) * s->m_x.size()); }
void ref_set_x(void *h, const double *x) { RefSystem *s = (RefSystem *)h; std::memcpy(s->m_x.data(), x, sizeof(double) * s->m_x.size()); }
void ref_get_v(void *h, double *v) { RefSystem *s = (RefSystem *)h; std::memcpy(v, s->m_v.data(), sizeof(double) * s->m_v.size()); }
void ref_set_v(void *h, const double *v) { RefSystem *s = (RefSystem *)h; std::memcpy(s->m_v.data(), v, sizeof(double) * s->m_v.size()); }
void ref_get_masses(void *h, double *m) { RefSystem *s = (RefSystem *)h; std::memcpy(m, s->m_masses.data(), sizeof(double) * s->m_masses.size()); }
void ref_get_u(void *h, double *u) { RefSystem *s = (RefSystem *)h; std::memcpy(u, s->U().data(), sizeof(double) * s->U().size()); }
void ref_get_z(void *h, double *z) { RefSystem *s = (RefSystem *)h; std::memcpy(z, s->Z().data(), sizeof(double) * s->Z().size()); }
void ref_get_wdiag(void *h, double *w) { RefSystem *s = (RefSystem *)h; std::memcpy(w, s->W().data(), sizeof(double) * s->W().size()); }
int ref_force_global_idx(void *h, int i) { return ((RefSystem *)h)->forces[i]->global_idx; }
double ref_force_weight(void *h, int i) { return ((RefSystem *)h)->forces[i]->weight; }
long ref_D_nnz(void *h) { return (long)((RefSystem *)h)->D().nonZeros(); }
// D in triplet form (row, col, val), column-major traversal
void ref_get_D(void *h, int *rows, int *cols, double *vals) {
    RefSystem *s = (RefSystem *)h; long k = 0;
    for (int c = 0; c < s->D().outerSize(); ++c)
        for (Eigen::SparseMatrix<double>::InnerIterator it(s->D(), c); it; ++it, ++k) { rows[k] = it.row(); cols[k] = it.col(); vals[k] = it.value(); }
}
// nnz of the LDLT factor of the 3n x 3n system
long ref_L_nnz(void *h) { return (long)((RefSystem *)h)->LDLT().matrixL().nestedExpression().nonZeros(); }

// x = solver.solve(b): the reference's own SimplicialLDLT (AMD ordering, 3n x 3n) on a caller-supplied right-hand side
// (System.cpp:62; SimplicialCholesky.h:153-177) -- the direct pin of the global step (tests/golden/solve_*.npz)
void ref_solve(void *h, const double *b, double *x) {
    RefSystem *s = (RefSystem *)h;
    const int n = (int)s->m_x.size();
    VectorXd bv = Eigen::Map<const VectorXd>(b, n);
    VectorXd xv = s->LDLT().solve(bv);
    std::memcpy(x, xv.data(), sizeof(double) * n);
}

// HyperElasticTet warm-start state (TetForce.hpp:146; cppoptlib meta.h:33; isolver.h:20)
int ref_get_hyper_state(void *h, int i, double *state4) {
    if (FungTriangle *g = dynamic_cast<FungTriangle *>(((RefSystem *)h)->forces[i].get())) {
        state4[0] = state4[1] = state4[2] = 0.0; state4[3] = g->solver->settings_.init_hess; return g->solver->n_iters;
    }
    HyperElasticTet *f = dynamic_cast<HyperElasticTet *>(((RefSystem *)h)->forces[i].get());
    if (!f) return -1;
    for (int j = 0; j < 3; ++j) state4[j] = f->last_prox_result[j];
    state4[3] = f->solver->settings_.init_hess;
    return f->solver->n_iters;
}
int ref_set_hyper_state(void *h, int i, const double *state4) {
    HyperElasticTet *f = dynamic_cast<HyperElasticTet *>(((RefSystem *)h)->forces[i].get());
    if (!f) return -1;
    for (int j = 0; j < 3; ++j) f->last_prox_result[j] = state4[j];
    f->solver->settings_.init_hess = state4[3];
    return 0;
}

// ---------------------------------------------------------------------------
// Stand-alone single-element project(): builds ONE force of `kind` over its own
// small node set (rest positions x_rest[nodes][3]), runs initialize() +
// get_selector() (global_idx = 0) and then n_calls consecutive project() calls
// on caller-supplied Dx rows, carrying u and the warm-start state across calls
// exactly like System::step does across ADMM iterations (System.cpp:51-67).
//   Dx   : [n_calls][rows]    in
//   u    : [rows]             in/out (carried)
//   z_out: [n_calls][rows]    out
//   u_out: [n_calls][rows]    out
//   state: [4]                in/out (hyperelastic only)
//   n_iters_out: [n_calls]    out (hyperelastic only; L-BFGS outer iterations)
//   init_out: weight, then rest data: tets B(4x3 col-major,12) + volume;
//             tris B(3x2 col-major,6) + area; bend alpha(4); spring rest_length
// ---------------------------------------------------------------------------
int ref_project_single(int kind, const double *x_rest, const double *params, double dt,
                       int n_calls, const double *Dx, double *u, double *z_out, double *u_out,
                       double *state, int *n_iters_out, double *init_out) {
    const int nn = ADMM_KIND_NODES[kind], rows = ADMM_KIND_ROWS[kind];
    VectorXd x = Eigen::Map<const VectorXd>(x_rest, 3 * nn), v = VectorXd::Zero(3 * nn), m = VectorXd::Ones(3 * nn);
    int idx[4] = {0, 1, 2, 3};
    RefSystem dummy;
    std::shared_ptr<Force> f = make_force(&dummy, kind, idx, params);
    if (!f) return -1;
    f->initialize(x, v, m, dt);
    std::vector<Eigen::Triplet<double> > trip; std::vector<double> w;
    f->get_selector(x, trip, w);
    if (init_out) {
        int k = 0; init_out[k++] = f->weight;
        if (LinearTetStrain *t = dynamic_cast<LinearTetStrain *>(f.get())) { for (int i = 0; i < 12; ++i) init_out[k++] = t->B.data()[i]; init_out[k++] = t->volume; }
        else if (TetVolume *t = dynamic_cast<TetVolume *>(f.get())) { for (int i = 0; i < 12; ++i) init_out[k++] = t->B.data()[i]; init_out[k++] = t->rest_volume; }
        else if (HyperElasticTet *t = dynamic_cast<HyperElasticTet *>(f.get())) { for (int i = 0; i < 12; ++i) init_out[k++] = t->B.data()[i]; init_out[k++] = t->volume; }
        else if (LimitedTriangleStrain *t = dynamic_cast<LimitedTriangleStrain *>(f.get())) { for (int i = 0; i < 6; ++i) init_out[k++] = t->B.data()[i]; init_out[k++] = t->area; }
        else if (FungTriangle *t = dynamic_cast<FungTriangle *>(f.get())) { for (int i = 0; i < 6; ++i) init_out[k++] = t->B.data()[i]; init_out[k++] = t->area; }
        else if (BendForce *t = dynamic_cast<BendForce *>(f.get())) { for (int i = 0; i < 4; ++i) init_out[k++] = t->alpha[i]; }
        else if (Spring *t = dynamic_cast<Spring *>(f.get())) { init_out[k++] = t->rest_length; }
    }
    HyperElasticTet *he = dynamic_cast<HyperElasticTet *>(f.get());
    FungTriangle *fu = dynamic_cast<FungTriangle *>(f.get());
    if (he && state) { for (int j = 0; j < 3; ++j) he->last_prox_result[j] = state[j]; he->solver->settings_.init_hess = state[3]; }
    if (fu && state) fu->solver->settings_.init_hess = state[3];
    const int R = (int)w.size(); // 36 for tets (TetForce.cpp:313-317), else rows
    VectorXd Dxv = VectorXd::Zero(R), uv = VectorXd::Zero(R), zv = VectorXd::Zero(R);
    for (int r = 0; r < rows; ++r) uv[r] = u[r];
    for (int c = 0; c < n_calls; ++c) {
        for (int r = 0; r < rows; ++r) Dxv[r] = Dx[(size_t)c * rows + r];
        f->project(dt, Dxv, uv, zv);
        for (int r = 0; r < rows; ++r) { z_out[(size_t)c * rows + r] = zv[r]; u_out[(size_t)c * rows + r] = uv[r]; }
        if (he && n_iters_out) n_iters_out[c] = he->solver->n_iters;
        if (fu && n_iters_out) n_iters_out[c] = fu->solver->n_iters;
    }
    for (int r = 0; r < rows; ++r) u[r] = uv[r];
    if (he && state) { for (int j = 0; j < 3; ++j) state[j] = he->last_prox_result[j]; state[3] = he->solver->settings_.init_hess; }
    if (fu && state) state[3] = fu->solver->settings_.init_hess;
    return 0;
}

// Eigen::JacobiSVD<Matrix3d>(F, ComputeFullU|ComputeFullV) as used by
// helper::oriented_svd / LinearTetStrain::project (TetForce.cpp:83-86,136).
// F, U, V col-major 3x3; S[3].
void ref_svd3(const double *F, double *U, double *S, double *V) {
    Eigen::Matrix3d Fm = Eigen::Map<const Eigen::Matrix3d>(F);
    Eigen::JacobiSVD<Eigen::Matrix3d> svd(Fm, Eigen::ComputeFullU | Eigen::ComputeFullV);
    Eigen::Map<Eigen::Matrix3d> Um(U), Vm(V);
    Um = svd.matrixU(); Vm = svd.matrixV();
    for (int i = 0; i < 3; ++i) S[i] = svd.singularValues()[i];
}
// JacobiSVD<Matrix<double,3,2>> (TriangleForce.cpp:87): U 3x3, S[2], V 2x2, col-major
void ref_svd32(const double *F, double *U, double *S, double *V) {
    Eigen::Matrix<double, 3, 2> Fm = Eigen::Map<const Eigen::Matrix<double, 3, 2> >(F);
    Eigen::JacobiSVD<Eigen::Matrix<double, 3, 2> > svd(Fm, Eigen::ComputeFullU | Eigen::ComputeFullV);
    Eigen::Map<Eigen::Matrix3d> Um(U); Eigen::Map<Eigen::Matrix2d> Vm(V);
    Um = svd.matrixU(); Vm = svd.matrixV();
    for (int i = 0; i < 2; ++i) S[i] = svd.singularValues()[i];
}

// Wall-clock of `frames` calls of System::step() (System.cpp:26-75); seconds.
double ref_time_steps(void *h, int frames) {
    RefSystem *s = (RefSystem *)h;
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) s->step();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int ref_omp_threads() { return omp_get_max_threads(); }
// bench.py's cpu_baseline times the reference with a few team sizes and reports the best one (a team of every hardware
// thread of a 256-thread host is 3.6x slower than 16-64 threads on the 100k-tet sample)
void ref_set_omp_threads(int n) { if (n > 0) omp_set_num_threads(n); }

} // extern "C"
