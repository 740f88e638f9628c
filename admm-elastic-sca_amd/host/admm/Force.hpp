// Force.hpp -- host-side mirror of the reference's force classes
// (deps/admm-elastic-sca/src/system/Force.hpp, TetForce.hpp, TriangleForce.hpp,
// BendForce.hpp, AnchorForce.hpp, CollisionForce.hpp, ExplicitForce.hpp and
// src/collision/*.hpp): same class names, constructor signatures, virtuals and
// public data members, so scene code that builds forces and pushes them into
// System::forces compiles unchanged.  The reference's per-class headers
// (TetForce.hpp, AnchorForce.hpp ...) exist next to this file and forward here.
//
// The plugin surface is the reference's (Force.hpp:37-57):
//     subclass admm::Force, implement get_selector() + project(), push it into system->forces.
// Built-in classes are *descriptions* (kind, node ids, parameters): their
// initialize / get_selector / project run inside libadmm_hip.so, one HIP kernel
// per kind.  A user-written subclass is host code by nature: System collects its
// selector rows once, the device evaluates D_i x for them every ADMM iteration,
// project() runs on the host, and z - u goes back into the device-side
// right-hand side (admm_hip_add_generic_batch).  Same for a user-written
// CollisionShape (its CollisionForce then projects on the host) and a
// user-written ExplicitForce (applied on the host before the frame is uploaded).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <typeinfo>
#include <vector>

#include "Vec.hpp"
#include "admm_kinds.h"

namespace admm {

class Force {
public:
    int global_idx;  // position of this force's first row in u / z (built-ins: compact rows, filled by System::initialize;
                     // user forces: the weights.size() they saw in get_selector, i.e. their offset among the user rows)
    double weight;   // computed by the force from its stiffness (built-ins: filled by System::initialize)
    Force() : global_idx(0), weight(0.f) {}
    virtual ~Force() {}
    // Force.hpp:46-55
    virtual void initialize(const Eigen::VectorXd & /*x*/, const Eigen::VectorXd & /*v*/, const Eigen::VectorXd & /*masses*/, const double /*timestep*/) {}
    virtual void get_selector(const Eigen::VectorXd &x, std::vector<Eigen::Triplet<double> > &triplets, std::vector<double> &weights) = 0;
    virtual void project(double dt, const Eigen::VectorXd &Dx, Eigen::VectorXd &u, Eigen::VectorXd &z) const = 0;
    virtual void set_eps(double) {}
    // ---- mirror only ----
    // >= 0: a built-in kind with a HIP kernel; -1: user code, projected on the host
    virtual int kind() const { return -1; }
    // node ids, ADMM_KIND_PARAMS[kind] parameters (admm_kinds.h)
    virtual void describe(int *, double *) const {}
    // the class whose kernel kind() names; a subclass of a built-in that does not say so is refused by System::initialize
    virtual const std::type_info &device_type() const { return typeid(void); }
};

// A force whose Force::initialize / get_selector / project run inside libadmm_hip.so (no host arithmetic exists for it).
class DeviceForce : public Force {
public:
    void get_selector(const Eigen::VectorXd &, std::vector<Eigen::Triplet<double> > &, std::vector<double> &) { host_call("get_selector"); }
    void project(double, const Eigen::VectorXd &, Eigen::VectorXd &, Eigen::VectorXd &) const { host_call("project"); }
private:
    static void host_call(const char *what) {
        std::fprintf(stderr, "\n**Solver Error: %s() of a built-in force was called on the host; built-in forces run on the GPU inside System::initialize()/step()\n", what);
        std::abort();
    }
};
#define ADMM_DEVICE_FORCE(Class, Kind) int kind() const { return Kind; } const std::type_info &device_type() const { return typeid(Class); }

class Spring : public DeviceForce {
public:
    Spring(int idx0_, int idx1_, double stiffness_) : idx0(idx0_), idx1(idx1_), stiffness(stiffness_), rest_length(0) {}
    ADMM_DEVICE_FORCE(Spring, ADMM_KIND_SPRING)
    void describe(int *idx, double *p) const { idx[0] = idx0; idx[1] = idx1; p[0] = stiffness; }
    int idx0, idx1;
    double stiffness, rest_length;
};

class LinearTetStrain : public DeviceForce {
public:
    LinearTetStrain(int i0, int i1, int i2, int i3, double stiffness_, double weight_scale_ = 1.f)
        : stiffness(stiffness_), volume(0.0), weight_scale(weight_scale_) { idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3; }
    ADMM_DEVICE_FORCE(LinearTetStrain, ADMM_KIND_TET_LINEAR)
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = stiffness; }
    int idx[4];
    double stiffness, volume, weight_scale;
};

class TetVolume : public DeviceForce {
public:
    TetVolume(int i0, int i1, int i2, int i3, double stiffness_, double limit_min_, double limit_max_)
        : stiffness(stiffness_), rest_volume(0.0), limit_min(limit_min_), limit_max(limit_max_) { idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3; }
    ADMM_DEVICE_FORCE(TetVolume, ADMM_KIND_TET_VOLUME)
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = stiffness; p[1] = limit_min; p[2] = limit_max; }
    int idx[4];
    double stiffness, rest_volume, limit_min, limit_max;
};

// type "nh"/"0" -> Neo-Hookean, "stvk"/"1" -> St. Venant-Kirchhoff (TetForce.hpp:118-121)
class HyperElasticTet : public DeviceForce {
public:
    HyperElasticTet(int i0, int i1, int i2, int i3, double mu_, double lambda_, int max_iterations_, std::string type_)
        : mu(mu_), lambda(lambda_), volume(0.0), max_iterations(max_iterations_) {
        idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3;
        type = 0; if (type_ == "stvk" || type_ == "1") type = 1;
    }
    int kind() const { return type == 1 ? ADMM_KIND_TET_STVK : ADMM_KIND_TET_NH; }
    const std::type_info &device_type() const { return typeid(HyperElasticTet); }
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = mu; p[1] = lambda; p[2] = max_iterations; }
    int idx[4];
    int type;
    double mu, lambda, volume;
    int max_iterations;
};

class LimitedTriangleStrain : public DeviceForce {
public:
    LimitedTriangleStrain(int id0_, int id1_, int id2_, double stiffness_, double limit_min_, double limit_max_, bool strain_limiting_ = true)
        : id0(id0_), id1(id1_), id2(id2_), stiffness(stiffness_), limit_min(limit_min_), limit_max(limit_max_), area(0), strain_limiting(strain_limiting_) {}
    ADMM_DEVICE_FORCE(LimitedTriangleStrain, ADMM_KIND_TRI_STRAIN)
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
    ADMM_DEVICE_FORCE(TriArea, ADMM_KIND_TRI_AREA)
    void describe(int *id, double *p) const { id[0] = id0; id[1] = id1; id[2] = id2; p[0] = stiffness; p[1] = iters; p[2] = limit_min; p[3] = limit_max; }
    int iters;
};

// TriangleForce.hpp:106-124: Fung skin membrane, prox by L-BFGS (2 variables, maxIter 10, gradTol 1e-6)
class FungTriangle : public DeviceForce {
public:
    FungTriangle(int id0_, int id1_, int id2_, double mu_, double limit_min_, double limit_max_)
        : id0(id0_), id1(id1_), id2(id2_), mu(mu_), limit_min(limit_min_), limit_max(limit_max_), area(0) {}
    ADMM_DEVICE_FORCE(FungTriangle, ADMM_KIND_TRI_FUNG)
    void describe(int *id, double *p) const { id[0] = id0; id[1] = id1; id[2] = id2; p[0] = mu; p[1] = limit_min; p[2] = limit_max; }
    int id0, id1, id2;
    double mu, limit_min, limit_max, area;
};

class BendForce : public DeviceForce {
public:
    BendForce(int i0, int i1, int i2, int i3, double stiffness_) : stiffness(stiffness_) { idx[0] = i0; idx[1] = i1; idx[2] = i2; idx[3] = i3; weight = std::sqrt(stiffness); }
    ADMM_DEVICE_FORCE(BendForce, ADMM_KIND_BEND)
    void describe(int *id, double *p) const { for (int i = 0; i < 4; ++i) id[i] = idx[i]; p[0] = stiffness; }
    int idx[4];
    double stiffness;
};

class StaticAnchor : public DeviceForce {
public:
    StaticAnchor(int idx_, double use_weight_ = -1.0) : idx(idx_), use_weight(use_weight_) {
        if (use_weight_ > 0.0) weight = use_weight_; else weight = 1000.f;
    }
    ADMM_DEVICE_FORCE(StaticAnchor, ADMM_KIND_ANCHOR)
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

class MovingAnchor : public DeviceForce {
public:
    MovingAnchor(int idx_, std::shared_ptr<ControlPoint> p_, double use_weight_ = -1.0) : idx(idx_), point(p_) {
        point->anchorForce = this;
        if (use_weight_ > 0.0) weight = use_weight_; else weight = 1000.f;
    }
    ADMM_DEVICE_FORCE(MovingAnchor, ADMM_KIND_ANCHOR)
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

// deps/admm-elastic-sca/src/collision/: shapes tested in list order by CollisionForce.  The three analytic shapes have a
// device form (shape_type() >= 0); a user-written shape implements isColliding / projectOut like in the reference
// (CollisionShape.hpp:34-38) and makes its CollisionForce a host-projected force.
class CollisionShape {
public:
    CollisionShape(Vector3d shapeCenter) { center = shapeCenter; }
    virtual ~CollisionShape() {}
    virtual double isColliding(Vector3d pos) const = 0;            // > 0 inside
    virtual Vector3d projectOut(const Vector3d currPos) const = 0;
    virtual int shape_type() const { return -1; }                  // ADMM_SHAPE_*, -1 = user code
    virtual double shape_radius() const { return 0.0; }
    Vector3d center;
};
class CollisionFloor : public CollisionShape {
public:
    CollisionFloor(Vector3d shapeCenter) : CollisionShape(shapeCenter), radius(0) {}
    double isColliding(Vector3d pos) const { return center[1] - pos[1]; }
    Vector3d projectOut(const Vector3d currPos) const { return Vector3d(currPos[0], center[1], currPos[2]); }
    int shape_type() const { return typeid(*this) == typeid(CollisionFloor) ? ADMM_SHAPE_FLOOR : -1; }
    double radius;
};
class CollisionSphere : public CollisionShape {
public:
    CollisionSphere(Vector3d shapeCenter, double sphRadius) : CollisionShape(shapeCenter), radius(sphRadius) {}
    double isColliding(Vector3d pos) const { return radius - (pos - center).norm(); }
    Vector3d projectOut(const Vector3d currPos) const { Vector3d d = currPos - center; return center + radius * (d / d.norm()); }
    int shape_type() const { return typeid(*this) == typeid(CollisionSphere) ? ADMM_SHAPE_SPHERE : -1; }
    double shape_radius() const { return radius; }
    double radius;
};
// axis parallel to z through (center.x, center.y); CollisionCylinder.hpp:48-50 drops center.z
class CollisionCylinder : public CollisionShape {
public:
    CollisionCylinder(Vector3d shapeCenter, Vector3d /*cylScale*/, double cylRadius) : CollisionShape(Vector3d(shapeCenter[0], shapeCenter[1], 0)), radius(cylRadius), length(0) {}
    double isColliding(Vector3d pos) const { return radius - (Vector3d(pos[0], pos[1], 0) - center).norm(); }
    Vector3d projectOut(const Vector3d currPos) const {
        Vector3d flat(currPos[0], currPos[1], 0);
        Vector3d d = flat - center;
        return (center + radius * (d / d.norm())) + Vector3d(0, 0, currPos[2]);
    }
    int shape_type() const { return typeid(*this) == typeid(CollisionCylinder) ? ADMM_SHAPE_CYLINDER : -1; }
    double shape_radius() const { return radius; }
    double radius, length;
};

// One force over ALL nodes (CollisionForce.hpp:31-46).  With analytic shapes only: a device batch with one element per
// node.  With a user-written shape in the list the shapes' virtuals have to run on the host, so the force describes itself
// like any user force: identity rows (CollisionForce.cpp:29-36) and the projection loop of :38-70.
class CollisionForce : public Force {
public:
    CollisionForce(std::vector<std::shared_ptr<CollisionShape> > &collShapes, double use_weight = 32.0) : collisionShapes(collShapes), Di_rows(0), n_nodes(0) { weight = use_weight; }
    bool device_shapes() const { for (size_t j = 0; j < collisionShapes.size(); ++j) if (collisionShapes[j]->shape_type() < 0) return false; return true; }
    int kind() const { return (typeid(*this) == typeid(CollisionForce) && device_shapes()) ? ADMM_KIND_COLLISION : -1; }
    const std::type_info &device_type() const { return typeid(CollisionForce); }
    void initialize(const Eigen::VectorXd &x, const Eigen::VectorXd &, const Eigen::VectorXd &, const double) { n_nodes = (int)x.size() / 3; }
    void get_selector(const Eigen::VectorXd &x, std::vector<Eigen::Triplet<double> > &triplets, std::vector<double> &weights) {
        global_idx = (int)weights.size();
        Di_rows = (int)x.size();
        for (int i = 0; i < Di_rows; ++i) { triplets.push_back(Eigen::Triplet<double>(i + global_idx, i, 1.0)); weights.push_back(weight); }
    }
    void project(double, const Eigen::VectorXd &Dx, Eigen::VectorXd &u, Eigen::VectorXd &z) const {
        for (int i = 0; i < Di_rows; i += 3) {
            const int g = global_idx + i;
            Vector3d free_pos(Dx[g] + u[g], Dx[g + 1] + u[g + 1], Dx[g + 2] + u[g + 2]);
            Vector3d point = free_pos;
            for (size_t j = 0; j < collisionShapes.size(); ++j) if (collisionShapes[j]->isColliding(point) > 0) point = collisionShapes[j]->projectOut(point);
            for (int c = 0; c < 3; ++c) { u[g + c] += (Dx[g + c] - point[c]); z[g + c] = point[c]; }
        }
    }
    std::vector<std::shared_ptr<CollisionShape> > collisionShapes;
    int Di_rows, n_nodes;
};

// ExplicitForce.hpp:51-59: constant acceleration on all nodes or on an index subset.  ExplicitForce and WindForce themselves
// run on the device; a user-written subclass overrides project() and is applied on the host at the start of System::step().
class ExplicitForce {
public:
    ExplicitForce(std::vector<int> indices_ = std::vector<int>(0)) { indices = indices_; }
    ExplicitForce(Vector3d direction_, std::vector<int> indices_ = std::vector<int>(0)) { direction = direction_; indices = indices_; }
    virtual ~ExplicitForce() {}
    // ExplicitForce.cpp:29-39 (what a subclass inherits when it does not override)
    virtual void project(double dt, Eigen::VectorXd &x, Eigen::VectorXd &v, Eigen::VectorXd & /*m*/) const {
        if (indices.empty()) for (std::ptrdiff_t i = 0; i < (std::ptrdiff_t)x.size() / 3; ++i) { for (int c = 0; c < 3; ++c) v[3 * i + c] += dt * direction[c]; }
        else for (size_t q = 0; q < indices.size(); ++q) for (int c = 0; c < 3; ++c) v[3 * (std::ptrdiff_t)indices[q] + c] += dt * direction[c];
    }
    virtual int explicit_type() const { return ADMM_EXPLICIT_CONST; }
    virtual const std::vector<int> &index_list() const { return indices; }
    bool on_device() const { return typeid(*this) == device_class(); }
    virtual const std::type_info &device_class() const { return typeid(ExplicitForce); }
    Vector3d direction;
    std::vector<int> indices;
};

// ExplicitForce.hpp:62-71: aerodynamic drag on a list of triangles; direction is host-mutable
class WindForce : public ExplicitForce {
public:
    WindForce(std::vector<int> &tris_) : tris(tris_) { this->direction = Vector3d(0, 0, 0); }
    void project(double, Eigen::VectorXd &, Eigen::VectorXd &, Eigen::VectorXd &) const {
        std::fprintf(stderr, "\n**Solver Error: WindForce::project() runs on the GPU; a subclass must override project()\n");
        std::abort();
    }
    int explicit_type() const { return ADMM_EXPLICIT_WIND; }
    const std::vector<int> &index_list() const { return tris; }
    const std::type_info &device_class() const { return typeid(WindForce); }
    std::vector<int> tris;
};

} // namespace admm
