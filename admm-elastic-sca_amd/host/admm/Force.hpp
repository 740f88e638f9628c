// Force.hpp -- host-side mirror of the reference's force classes
// (deps/admm-elastic-sca/src/system/Force.hpp, TetForce.hpp, TriangleForce.hpp,
// BendForce.hpp, AnchorForce.hpp, ExplicitForce.hpp): same class names,
// constructor signatures and public data members, so scene code that builds
// forces and pushes them into System::forces compiles unchanged.
//
// What differs: a force here is a *description* (kind, node ids, parameters).
// Force::initialize / get_selector / project run inside libadmm_hip.so
// (admm_hip_finalize / admm_hip_step); there is no host project().  A
// user-defined subclass has no kernel: System::initialize() refuses it.
#pragma once
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "Vec.hpp"
#include "admm_kinds.h"

namespace admm {

class Force {
public:
    int global_idx;  // compact row of the force's first row in u/z (filled by System::initialize)
    double weight;   // computed by the library from the stiffness (filled by System::initialize)
    Force() : global_idx(0), weight(0.f) {}
    virtual ~Force() {}
    // -1 = no accelerated kernel (user subclass)
    virtual int kind() const { return -1; }
    // node ids, ADMM_KIND_PARAMS[kind] parameters (admm_kinds.h)
    virtual void describe(int *, double *) const {}
    virtual void set_eps(double) {}
};

class Spring : public Force {
public:
    Spring(int idx0_, int idx1_, double stiffness_) : idx0(idx0_), idx1(idx1_), stiffness(stiffness_), rest_length(0) {}
    int kind() const { return ADMM_KIND_SPRING; }
    void describe(int *idx, double *p) const { idx[0] = idx0; idx[1] = idx1; p[0] = stiffness; }
    int idx0, idx1;
    double stiffness, rest_length;
};

class LinearTetStrain : public Force {
public:
    LinearTetStrain(int i0, int i1, int i2, int i3, double stiffness_, double weight_scale_ = 1.f)
        : stiffness(stiffness_), volume(0.0), weight_scale(weight_scale_) { idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3; }
    int kind() const { return ADMM_KIND_TET_LINEAR; }
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = stiffness; }
    int idx[4];
    double stiffness, volume, weight_scale;
};

class TetVolume : public Force {
public:
    TetVolume(int i0, int i1, int i2, int i3, double stiffness_, double limit_min_, double limit_max_)
        : stiffness(stiffness_), rest_volume(0.0), limit_min(limit_min_), limit_max(limit_max_) { idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3; }
    int kind() const { return ADMM_KIND_TET_VOLUME; }
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = stiffness; p[1] = limit_min; p[2] = limit_max; }
    int idx[4];
    double stiffness, rest_volume, limit_min, limit_max;
};

// type "nh"/"0" -> Neo-Hookean, "stvk"/"1" -> St. Venant-Kirchhoff (TetForce.hpp:118-121)
class HyperElasticTet : public Force {
public:
    HyperElasticTet(int i0, int i1, int i2, int i3, double mu_, double lambda_, int max_iterations_, std::string type_)
        : mu(mu_), lambda(lambda_), volume(0.0), max_iterations(max_iterations_) {
        idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3;
        type = 0; if (type_ == "stvk" || type_ == "1") type = 1;
    }
    int kind() const { return type == 1 ? ADMM_KIND_TET_STVK : ADMM_KIND_TET_NH; }
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = mu; p[1] = lambda; p[2] = max_iterations; }
    int idx[4];
    int type;
    double mu, lambda, volume;
    int max_iterations;
};

class LimitedTriangleStrain : public Force {
public:
    LimitedTriangleStrain(int id0_, int id1_, int id2_, double stiffness_, double limit_min_, double limit_max_, bool strain_limiting_ = true)
        : id0(id0_), id1(id1_), id2(id2_), stiffness(stiffness_), limit_min(limit_min_), limit_max(limit_max_), area(0), strain_limiting(strain_limiting_) {}
    int kind() const { return ADMM_KIND_TRI_STRAIN; }
    void describe(int *id, double *p) const { id[0] = id0; id[1] = id1; id[2] = id2; p[0] = stiffness; p[1] = limit_min; p[2] = limit_max; p[3] = strain_limiting ? 1.0 : 0.0; }
    int id0, id1, id2;
    double stiffness, limit_min, limit_max, area;
    bool strain_limiting;
};

// TriangleForce.hpp:126-133: area-preserving triangle; inherits the strain triangle's rest data and weight
class TriArea : public LimitedTriangleStrain {
public:
    TriArea(int id0_, int id1_, int id2_, double stiffness_, int iters_, double limit_min_, double limit_max_)
        : LimitedTriangleStrain(id0_, id1_, id2_, stiffness_, limit_min_, limit_max_), iters(iters_) {}
    int kind() const { return ADMM_KIND_TRI_AREA; }
    void describe(int *id, double *p) const { id[0] = id0; id[1] = id1; id[2] = id2; p[0] = stiffness; p[1] = iters; p[2] = limit_min; p[3] = limit_max; }
    int iters;
};

// TriangleForce.hpp:106-124: Fung skin membrane, prox by L-BFGS (2 variables, maxIter 10, gradTol 1e-6)
class FungTriangle : public Force {
public:
    FungTriangle(int id0_, int id1_, int id2_, double mu_, double limit_min_, double limit_max_)
        : id0(id0_), id1(id1_), id2(id2_), mu(mu_), limit_min(limit_min_), limit_max(limit_max_), area(0) {}
    int kind() const { return ADMM_KIND_TRI_FUNG; }
    void describe(int *id, double *p) const { id[0] = id0; id[1] = id1; id[2] = id2; p[0] = mu; p[1] = limit_min; p[2] = limit_max; }
    int id0, id1, id2;
    double mu, limit_min, limit_max, area;
};

class BendForce : public Force {
public:
    BendForce(int i0, int i1, int i2, int i3, double stiffness_) : stiffness(stiffness_) { idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3; weight = std::sqrt(stiffness); }
    int kind() const { return ADMM_KIND_BEND; }
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = stiffness; }
    int idx[4];
    double stiffness;
};

class StaticAnchor : public Force {
public:
    StaticAnchor(int idx_, double use_weight_ = -1.0) : idx(idx_), use_weight(use_weight_) {
        if (use_weight_ > 0.0) weight = use_weight_; else weight = 1000.f;
    }
    int kind() const { return ADMM_KIND_ANCHOR; }
    void describe(int *id, double *p) const { id[0] = idx; p[0] = weight; p[1] = 1.0; }
    int idx;
    double use_weight;
    Vector3d pos;
};

class MovingAnchor;
class ControlPoint {
public:
    ControlPoint() : active(true), anchorForce(0) { pos.setZero(); }
    ControlPoint(Vector3d pos_) : pos(pos_), active(true), anchorForce(0) {}
    Vector3d pos;
    bool active;
    MovingAnchor *anchorForce;
};

class MovingAnchor : public Force {
public:
    MovingAnchor(int idx_, std::shared_ptr<ControlPoint> p_, double use_weight_ = -1.0) : idx(idx_), point(p_) {
        point->anchorForce = this;
        if (use_weight_ > 0.0) weight = use_weight_; else weight = 1000.f;
    }
    int kind() const { return ADMM_KIND_ANCHOR; }
    void describe(int *id, double *p) const { id[0] = idx; p[0] = weight; p[1] = point->active ? 1.0 : 0.0; }
    int idx;
    std::shared_ptr<ControlPoint> point;
};

namespace helper {
// AnchorForce.hpp:31-48
static inline Vector3d smooth_move(double total_elapsed_dt, double start_dt, double end_dt, Vector3d start, Vector3d end) {
    if (total_elapsed_dt < start_dt) return start;
    double tRatio = (total_elapsed_dt - start_dt) / (end_dt - start_dt);
    if (tRatio > 1.0) return end;
    Vector3d displacement = end - start;
    return (start + (3.0 * tRatio * tRatio - 2.0 * tRatio * tRatio * tRatio) * displacement);
}
static inline Vector3d linear_move(double total_elapsed_dt, double start_dt, double end_dt, Vector3d start, Vector3d end) {
    if (total_elapsed_dt < start_dt) return start;
    double tRatio = (total_elapsed_dt - start_dt) / (end_dt - start_dt);
    if (tRatio > 1.0) return end;
    Vector3d displacement = end - start;
    return (start + displacement);
}
} // namespace helper

// deps/admm-elastic-sca/src/collision/: analytic shapes tested in list order by CollisionForce
class CollisionShape {
public:
    CollisionShape(Vector3d shapeCenter) { center = shapeCenter; }
    virtual ~CollisionShape() {}
    virtual int shape_type() const = 0;                 // ADMM_SHAPE_*
    virtual double shape_radius() const { return 0.0; }
    Vector3d center;
};
class CollisionFloor : public CollisionShape {
public:
    CollisionFloor(Vector3d shapeCenter) : CollisionShape(shapeCenter), radius(0) {}
    int shape_type() const { return ADMM_SHAPE_FLOOR; }
    double radius;
};
class CollisionSphere : public CollisionShape {
public:
    CollisionSphere(Vector3d shapeCenter, double sphRadius) : CollisionShape(shapeCenter), radius(sphRadius) {}
    int shape_type() const { return ADMM_SHAPE_SPHERE; }
    double shape_radius() const { return radius; }
    double radius;
};
// axis parallel to z through (center.x, center.y); CollisionCylinder.hpp:48-50 drops center.z
class CollisionCylinder : public CollisionShape {
public:
    CollisionCylinder(Vector3d shapeCenter, Vector3d /*cylScale*/, double cylRadius) : CollisionShape(Vector3d(shapeCenter[0], shapeCenter[1], 0)), radius(cylRadius), length(0) {}
    int shape_type() const { return ADMM_SHAPE_CYLINDER; }
    double shape_radius() const { return radius; }
    double radius, length;
};

// One force over ALL nodes (CollisionForce.hpp:31-46): a batch with one element per node.
class CollisionForce : public Force {
public:
    CollisionForce(std::vector<std::shared_ptr<CollisionShape> > &collShapes, double use_weight = 32.0) : collisionShapes(collShapes), Di_rows(0), n_nodes(0) { weight = use_weight; }
    int kind() const { return ADMM_KIND_COLLISION; }
    std::vector<std::shared_ptr<CollisionShape> > collisionShapes;
    int Di_rows, n_nodes;
};

// ExplicitForce.hpp:51-59: constant acceleration on all nodes or on an index subset
class ExplicitForce {
public:
    ExplicitForce(std::vector<int> indices_ = std::vector<int>(0)) { indices = indices_; }
    ExplicitForce(Vector3d direction_, std::vector<int> indices_ = std::vector<int>(0)) { direction = direction_; indices = indices_; }
    virtual ~ExplicitForce() {}
    virtual int explicit_type() const { return ADMM_EXPLICIT_CONST; }
    virtual const std::vector<int> &index_list() const { return indices; }
    Vector3d direction;
    std::vector<int> indices;
};

// ExplicitForce.hpp:62-71: aerodynamic drag on a list of triangles; direction is host-mutable
class WindForce : public ExplicitForce {
public:
    WindForce(std::vector<int> &tris_) : tris(tris_) { this->direction = Vector3d(0, 0, 0); }
    int explicit_type() const { return ADMM_EXPLICIT_WIND; }
    const std::vector<int> &index_list() const { return tris; }
    std::vector<int> tris;
};

} // namespace admm
