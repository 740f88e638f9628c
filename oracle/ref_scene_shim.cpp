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
#include "CollisionCylinder.hpp"
#include "CollisionFloor.hpp"
#include "CollisionSphere.hpp"
#include "CollisionForce.hpp"

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
    if (CollisionForce *t = dynamic_cast<CollisionForce *>(f)) { idx[0] = (int)t->collisionShapes.size(); p[0] = t->weight; return ADMM_KIND_COLLISION; }
    return -1;
}

// names of scene->object_params in the reference's own (unordered_map) iteration order, '\n'-joined;
// the sample programs and SimContext::initialize walk the objects in this order
int refscene_object_order(void *h, char *buf, int cap) {
    SimContext *c = (SimContext *)h; std::string s;
    for (auto it = c->scene->object_params.begin(); it != c->scene->object_params.end(); ++it) { s += it->first; s += "\n"; }
    if ((int)s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.c_str(), s.size() + 1); return (int)c->scene->object_params.size();
}
// surface/triangle faces of a named object as the loader left them (-1: no such object / no mesh)
int refscene_n_faces(void *h, const char *name) {
    SimContext *c = (SimContext *)h; if (!c->scene->objects_map.count(name)) return -1;
    std::shared_ptr<trimesh::TriMesh> m = c->scene->objects_map[name]->get_TriMesh(); return m ? (int)m->faces.size() : -1;
}
int refscene_n_vertices(void *h, const char *name) {
    SimContext *c = (SimContext *)h; if (!c->scene->objects_map.count(name)) return -1;
    std::shared_ptr<trimesh::TriMesh> m = c->scene->objects_map[name]->get_TriMesh(); return m ? (int)m->vertices.size() : -1;
}
void refscene_get_faces(void *h, const char *name, int *out) {
    SimContext *c = (SimContext *)h; std::shared_ptr<trimesh::TriMesh> m = c->scene->objects_map[name]->get_TriMesh();
    for (size_t f = 0; f < m->faces.size(); ++f) for (int j = 0; j < 3; ++j) out[3 * f + j] = m->faces[f][j];
}
// WindForce face list (node triples) of explicit force i, 0 if it is not a wind force
int refscene_wind_size(void *h, int i) { WindForce *w = dynamic_cast<WindForce *>(((SimContext *)h)->system->explicit_forces[i].get()); return w ? (int)w->tris.size() : 0; }
void refscene_get_wind(void *h, int i, int *out) { WindForce *w = dynamic_cast<WindForce *>(((SimContext *)h)->system->explicit_forces[i].get()); std::memcpy(out, w->tris.data(), sizeof(int) * w->tris.size()); }

// cylinders of CollisionForce i: (cx, cy, cz, radius) per shape, in list order
void refscene_get_cylinders(void *h, int i, double *out) {
    CollisionForce *t = dynamic_cast<CollisionForce *>(((SimContext *)h)->system->forces[i].get());
    for (size_t s = 0; s < t->collisionShapes.size(); ++s) {
        CollisionCylinder *c = dynamic_cast<CollisionCylinder *>(t->collisionShapes[s].get());
        for (int j = 0; j < 3; ++j) out[4 * s + j] = c->center[j];
        out[4 * s + 3] = c->radius;
    }
}

// ---- what the sample mains do between load() and initialize() (samples/*/*.cpp), so whole-scene
// ---- trajectories can be generated without the GL application
void refscene_set_x(void *h, const double *x) { SimContext *c = (SimContext *)h; std::memcpy(c->system->m_x.data(), x, sizeof(double) * c->system->m_x.size()); }
void refscene_add_static_anchor(void *h, int idx) { ((SimContext *)h)->system->forces.push_back(std::shared_ptr<Force>(new StaticAnchor(idx))); }
// samples/windyflag/windyflag.cpp:98-128: wind over the faces of every object that has a force
void refscene_add_wind(void *h, const double *dir) {
    SimContext *c = (SimContext *)h; std::vector<int> faces; int total = 0;
    for (auto it = c->scene->object_params.begin(); it != c->scene->object_params.end(); ++it) {
        bool has_force = false;
        for (size_t p = 0; p < it->second.size(); ++p) if (it->second[p].tag == "force") has_force = true;
        if (!has_force) continue;
        std::shared_ptr<trimesh::TriMesh> m = c->scene->objects_map[it->first]->get_TriMesh();
        for (size_t f = 0; f < m->faces.size(); ++f) for (int j = 0; j < 3; ++j) faces.push_back(m->faces[f][j] + total);
        total += (int)m->vertices.size();
    }
    std::shared_ptr<ExplicitForce> w(new WindForce(faces));
    w->direction = Eigen::Vector3d(dir[0], dir[1], dir[2]);
    c->system->explicit_forces.push_back(w);
}
// samples/plinkopony/plinkopony.cpp:53-96: one CollisionCylinder per object named c*, then one CollisionForce
int refscene_add_cylinder_collision(void *h) {
    SimContext *c = (SimContext *)h; std::vector<std::shared_ptr<CollisionShape> > shapes;
    for (auto it = c->scene->object_params.begin(); it != c->scene->object_params.end(); ++it) {
        if (it->first[0] != 'c') continue;
        double rad = 1.f; Eigen::Vector3d center(0, 0, 0), scale(1, 1, 1);
        for (size_t i = 0; i < it->second.size(); ++i) {
            if (it->second[i].tag == "scale_copy") { trimesh::vec v = it->second[i].as_vec3(); scale = Eigen::Vector3d(v[0], v[1], v[2]); }
            else if (it->second[i].tag == "translate_copy") { trimesh::vec v = it->second[i].as_vec3(); center = Eigen::Vector3d(v[0], v[1], v[2]); }
            else if (it->second[i].tag == "radius") rad = it->second[i].as_double();
        }
        shapes.push_back(std::shared_ptr<CollisionShape>(new CollisionCylinder(center, scale, rad)));
    }
    ((SimContext *)h)->system->forces.push_back(std::shared_ptr<Force>(new CollisionForce(shapes)));
    return (int)shapes.size();
}

} // extern "C"
