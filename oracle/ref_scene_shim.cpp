// ref_scene_shim.cpp -- TEST INFRASTRUCTURE, not product code.
//
// extern "C" wrapper around the reference's *scene loading* layer (src/SimContext,
// src/ForceBuilder and the parts of deps/mclscene they use, all compiled from the
// sources where they lie under /root/reference by oracle/Makefile into
// oracle/_ref/libscene_ref.so; nothing is copied).  Used only in the build
// container to generate the scene-ingest golden fixtures
// (tests/golden/make_golden_scenes.py): what nodes, masses and forces the
// reference's own XML loader produces for its shipped sample scenes.
#include "SimContext.hpp"
#include "AnchorForce.hpp"
#include "BendForce.hpp"
#include "TetForce.hpp"
#include "TriangleForce.hpp"
#include <cstring>

#include "../include/admm_kinds.h"

using namespace admm;

extern "C" {

void *refscene_load(const char *xml) {
    SimContext *c = new SimContext();
    try { c->load(xml); }
    catch (std::exception &e) { fprintf(stderr, "refscene_load: %s\n", e.what()); delete c; return nullptr; }
    return c;
}
void refscene_destroy(void *h) { delete (SimContext *)h; }
int refscene_initialize(void *h) { try { ((SimContext *)h)->initialize(); } catch (std::exception &e) { fprintf(stderr, "%s\n", e.what()); return 0; } return 1; }
int refscene_step(void *h) { return ((SimContext *)h)->system->step() ? 1 : 0; }

int refscene_dof(void *h) { return (int)((SimContext *)h)->system->m_x.size(); }
int refscene_n_forces(void *h) { return (int)((SimContext *)h)->system->forces.size(); }
int refscene_n_explicit(void *h) { return (int)((SimContext *)h)->system->explicit_forces.size(); }
void refscene_get_x(void *h, double *x) { SimContext *c = (SimContext *)h; std::memcpy(x, c->system->m_x.data(), sizeof(double) * c->system->m_x.size()); }
void refscene_get_m(void *h, double *m) { SimContext *c = (SimContext *)h; std::memcpy(m, c->system->m_masses.data(), sizeof(double) * c->system->m_masses.size()); }
void refscene_settings(void *h, double *dt, int *iters) { SimContext *c = (SimContext *)h; *dt = c->system->settings.timestep_s; *iters = c->system->settings.admm_iters; }
void refscene_get_explicit(void *h, int i, double *dir) { SimContext *c = (SimContext *)h; for (int j = 0; j < 3; ++j) dir[j] = c->system->explicit_forces[i]->direction[j]; }

// kind (admm_kinds.h, -1 unknown), node ids (4), parameters (4) of force i
int refscene_get_force(void *h, int i, int *idx, double *p) {
    Force *f = ((SimContext *)h)->system->forces[i].get();
    for (int j = 0; j < 4; ++j) { idx[j] = 0; p[j] = 0.0; }
    if (HyperElasticTet *t = dynamic_cast<HyperElasticTet *>(f)) { for (int j = 0; j < 4; ++j) idx[j] = t->idx[j]; p[0] = t->mu; p[1] = t->lambda; p[2] = (double)t->solver->settings_.maxIter; return t->type == 1 ? ADMM_KIND_TET_STVK : ADMM_KIND_TET_NH; }
    if (LinearTetStrain *t = dynamic_cast<LinearTetStrain *>(f)) { for (int j = 0; j < 4; ++j) idx[j] = t->idx[j]; p[0] = t->stiffness; return ADMM_KIND_TET_LINEAR; }
    if (TetVolume *t = dynamic_cast<TetVolume *>(f)) { for (int j = 0; j < 4; ++j) idx[j] = t->idx[j]; p[0] = t->stiffness; p[1] = t->limit_min; p[2] = t->limit_max; return ADMM_KIND_TET_VOLUME; }
    if (dynamic_cast<TriArea *>(f) || dynamic_cast<FungTriangle *>(f)) return -1;
    if (LimitedTriangleStrain *t = dynamic_cast<LimitedTriangleStrain *>(f)) { idx[0] = t->id0; idx[1] = t->id1; idx[2] = t->id2; p[0] = t->stiffness; p[1] = t->limit_min; p[2] = t->limit_max; p[3] = t->strain_limiting ? 1.0 : 0.0; return ADMM_KIND_TRI_STRAIN; }
    if (BendForce *t = dynamic_cast<BendForce *>(f)) { for (int j = 0; j < 4; ++j) idx[j] = t->idx[j]; p[0] = t->stiffness; return ADMM_KIND_BEND; }
    if (Spring *t = dynamic_cast<Spring *>(f)) { idx[0] = t->idx0; idx[1] = t->idx1; p[0] = t->stiffness; return ADMM_KIND_SPRING; }
    if (StaticAnchor *t = dynamic_cast<StaticAnchor *>(f)) { idx[0] = t->idx; p[0] = t->weight; p[1] = 1.0; return ADMM_KIND_ANCHOR; }
    return -1;
}

} // extern "C"
