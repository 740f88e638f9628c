// System.hpp -- host-side mirror of admm::System
// (deps/admm-elastic-sca/src/system/System.hpp:29-76, System.cpp) over the C ABI
// of libadmm_hip.so.  Same public members and methods:
//   settings{timestep_s, verbose, admm_iters} (+ parse_args/help), elapsed_s,
//   m_x, m_v, m_masses, explicit_forces, forces, add_nodes, initialize, step,
//   recompute_weights, pre_step_callbacks.
// initialize() hands nodes and forces to the library (batches of consecutive
// forces of one kind, in forces[] order), step() runs one frame on the GPU and
// leaves m_x / m_v valid on return, like the reference.  There is no CPU path:
// without a GPU initialize() prints the library's error and returns false.
#pragma once
#include <cstdio>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "Force.hpp"
#include "admm_hip.h"

namespace admm {

class System {
public:
    System() : elapsed_s(0.0), device_id(0), initialized(false), gpu(nullptr) {}
    ~System() { if (gpu) admm_hip_destroy(gpu); }
    System(const System &) = delete;
    System &operator=(const System &) = delete;

    struct Settings {
        void parse_args(int argc, char **argv) {   // System.cpp:182-198
            for (int i = 1; i < argc - 1; ++i) {
                std::string arg(argv[i]);
                std::stringstream val(argv[i + 1]);
                if (arg == "-help") help();
                else if (arg == "-dt") val >> timestep_s;
                else if (arg == "-v") val >> verbose;
                else if (arg == "-it") val >> admm_iters;
            }
            if (argc > 0) { std::string arg(argv[argc - 1]); if (arg == "-help") help(); }
        }
        void help() {                               // System.cpp:200-208
            printf("\n==========================================\nArgs:\n\t-dt: time step (s)\n\t-v: verbosity (higher -> show more)\n\t-it: # admm iters\n==========================================\n");
        }
        double timestep_s;
        int verbose;
        int admm_iters;
        // extension (the reference only describes it, System.cpp:64-65): with residual_tol_primal > 0 a step's ADMM loop
        // ends once |W(Dx-z)| <= residual_tol_primal and |D^T W^T W (z-z_prev)| <= residual_tol_dual, tested every
        // residual_check_every iterations; admm_iters stays the upper bound.  0 (default) = the reference's fixed count.
        double residual_tol_primal, residual_tol_dual;
        int residual_check_every;
        Settings() : timestep_s(0.04), verbose(1), admm_iters(10), residual_tol_primal(0.0), residual_tol_dual(0.0), residual_check_every(1) {}
    } settings;

    double elapsed_s;
    VectorXd m_x, m_v, m_masses;   // xyz-interleaved, 3 per node
    std::vector<std::shared_ptr<ExplicitForce> > explicit_forces;
    std::vector<std::shared_ptr<Force> > forces;
    std::vector<std::function<void(System *)> > pre_step_callbacks;
    int device_id;                 // HIP device of this System (one process per GPU)

    // System.cpp:78-95
    int add_nodes(VectorXd x, VectorXd m) {
        const int old_n = (int)m_x.size(), add = (int)x.size();
        m_x.conservativeResize(old_n + add); m_v.conservativeResize(old_n + add); m_masses.conservativeResize(old_n + add);
        for (int i = 0; i < add; ++i) { m_x[old_n + i] = x[i]; m_v[old_n + i] = 0.0; m_masses[old_n + i] = m[i]; }
        return (old_n + add) / 3;
    }

    // System.cpp:98-156
    bool initialize() {
        const int dof = (int)m_x.size();
        if (settings.verbose > 0) std::cout << "Solver::initialize: " << std::endl;
        if (settings.timestep_s <= 0.0) {
            std::cerr << "\n**Solver Error: timestep set to " << settings.timestep_s << "s, changing to 0.04s." << std::endl;
            settings.timestep_s = 0.04;
        }
        if (!(m_masses.size() == m_x.size() && m_x.size() >= 3)) { std::cerr << "\n**Solver Error: Problem with node data!" << std::endl; return false; }
        if (m_v.size() < m_x.size()) m_v.resize(m_x.size());
        m_v.setZero();
        if (gpu) { admm_hip_destroy(gpu); gpu = nullptr; }
        if (admm_hip_create(&gpu, device_id) != ADMM_OK) { std::cerr << "\n**Solver Error: no usable HIP device " << device_id << " (this solver has no CPU path)" << std::endl; gpu = nullptr; return false; }
        if (!check(admm_hip_set_timestep(gpu, settings.timestep_s))) return false;
        if (!check(admm_hip_add_nodes(gpu, dof / 3, m_x.data(), m_masses.data(), nullptr))) return false;
        // consecutive forces of one kind (and one anchor flavour) -> one batch, order preserved
        batch_first.clear(); batch_count.clear(); batch_kind.clear(); batch_moving.clear();
        for (size_t i = 0; i < forces.size();) {
            const int kind = forces[i]->kind();
            if (kind < 0) { std::cerr << "\n**Solver Error: force " << i << " is a user-defined Force subclass; only the built-in kinds have GPU kernels" << std::endl; return false; }
            if (kind == ADMM_KIND_COLLISION) {           // one force over all nodes -> one element per node
                CollisionForce *cf = static_cast<CollisionForce *>(forces[i].get());
                const int nn_ = dof / 3;
                std::vector<int32_t> idx(nn_); std::vector<double> par(nn_, cf->weight);
                for (int q = 0; q < nn_; ++q) idx[q] = q;
                int b = -1;
                if (!check(admm_hip_add_batch(gpu, kind, nn_, idx.data(), par.data(), nullptr, &b))) return false;
                cf->n_nodes = nn_; cf->Di_rows = dof;
                if (!push_shapes(cf)) return false;
                batch_first.push_back((int)i); batch_count.push_back(1); batch_kind.push_back(kind); batch_moving.push_back(false);
                ++i;
                continue;
            }
            const bool moving = dynamic_cast<MovingAnchor *>(forces[i].get()) != nullptr;
            size_t j = i;
            std::vector<int32_t> idx; std::vector<double> par, tgt;
            const int nn = ADMM_KIND_NODES[kind], np = ADMM_KIND_PARAMS[kind];
            for (; j < forces.size() && forces[j]->kind() == kind && (dynamic_cast<MovingAnchor *>(forces[j].get()) != nullptr) == moving; ++j) {
                int id[4] = {0, 0, 0, 0}; double p[4] = {0, 0, 0, 0};
                forces[j]->describe(id, p);
                idx.insert(idx.end(), id, id + nn); par.insert(par.end(), p, p + np);
                if (moving) { const MovingAnchor *ma = static_cast<const MovingAnchor *>(forces[j].get()); for (int c = 0; c < 3; ++c) tgt.push_back(ma->point->pos[c]); }
            }
            int b = -1;
            if (!check(admm_hip_add_batch(gpu, kind, (int)(j - i), idx.data(), par.data(), moving ? tgt.data() : nullptr, &b))) return false;
            batch_first.push_back((int)i); batch_count.push_back((int)(j - i)); batch_kind.push_back(kind); batch_moving.push_back(moving);
            i = j;
        }
        for (size_t i = 0; i < explicit_forces.size(); ++i) {
            const ExplicitForce &ef = *explicit_forces[i];
            const double d[3] = {ef.direction[0], ef.direction[1], ef.direction[2]};
            const std::vector<int> &il = ef.index_list();
            const int type = ef.explicit_type();
            const int cnt = (int)il.size() / (type == ADMM_EXPLICIT_WIND ? 3 : 1);
            std::vector<int32_t> il32(il.begin(), il.end());
            if (!check(admm_hip_add_explicit(gpu, type, d, cnt, il32.empty() ? nullptr : il32.data(), nullptr))) return false;
        }
        if (!check(admm_hip_finalize(gpu))) return false;
        // write back what Force::initialize / get_selector compute in the reference
        for (size_t b = 0; b < batch_first.size(); ++b) {
            const int ne = batch_kind[b] == ADMM_KIND_COLLISION ? dof / 3 : batch_count[b];
            std::vector<double> w(ne), rest((size_t)ne * 12); std::vector<int32_t> g(ne);
            if (batch_kind[b] == ADMM_KIND_COLLISION) {
                if (!check(admm_hip_read_rest(gpu, (int)b, w.data(), rest.data(), g.data()))) return false;
                forces[batch_first[b]]->global_idx = g[0];
                continue;
            }
            if (!check(admm_hip_read_rest(gpu, (int)b, w.data(), rest.data(), g.data()))) return false;
            if (batch_kind[b] == ADMM_KIND_COLLISION) continue;   // weight stays use_weight; rows start at the batch's first element
            for (int e = 0; e < batch_count[b]; ++e) { Force *f = forces[batch_first[b] + e].get(); f->weight = w[e]; f->global_idx = g[e]; }
        }
        if (settings.verbose >= 1) std::cout << m_x.size() / 3 << " nodes, " << forces.size() << " forces" << std::endl;
        initialized = true;
        return true;
    }

    // System.cpp:26-75
    bool step() {
        if (!initialized) return false;
        for (size_t cb = 0; cb < pre_step_callbacks.size(); ++cb) pre_step_callbacks[cb](this);
        // host-mutable parameters: control points, explicit-force directions
        for (size_t b = 0; b < batch_first.size(); ++b) if (batch_moving[b]) {
            std::vector<double> tgt((size_t)batch_count[b] * 3); std::vector<int32_t> act(batch_count[b]);
            for (int e = 0; e < batch_count[b]; ++e) {
                const MovingAnchor *ma = static_cast<const MovingAnchor *>(forces[batch_first[b] + e].get());
                for (int c = 0; c < 3; ++c) tgt[3 * (size_t)e + c] = ma->point->pos[c];
                act[e] = ma->point->active ? 1 : 0;
            }
            if (!check(admm_hip_update_anchors(gpu, (int)b, tgt.data(), act.data()))) return false;
        }
        for (size_t i = 0; i < explicit_forces.size(); ++i) { const Vector3d &d = explicit_forces[i]->direction; if (!check(admm_hip_set_gravity(gpu, (int)i, d[0], d[1], d[2]))) return false; }
        for (size_t b = 0; b < batch_first.size(); ++b) if (batch_kind[b] == ADMM_KIND_COLLISION && !push_shapes(static_cast<CollisionForce *>(forces[batch_first[b]].get()))) return false;
        // m_x / m_v are public and may have been edited by the caller between steps
        if (!check(admm_hip_set_x(gpu, m_x.data())) || !check(admm_hip_set_v(gpu, m_v.data()))) return false;
        if (!check(admm_hip_set_tolerance(gpu, settings.residual_tol_primal, settings.residual_tol_dual, settings.residual_check_every < 1 ? 1 : settings.residual_check_every))) return false;
        if (!check(admm_hip_step(gpu, settings.admm_iters))) return false;
        if (!check(admm_hip_get_x(gpu, m_x.data())) || !check(admm_hip_get_v(gpu, m_v.data()))) return false;
        // released MovingAnchors follow their node: point->pos = Dx (AnchorForce.cpp:80-83)
        for (size_t b = 0; b < batch_first.size(); ++b) if (batch_moving[b]) {
            std::vector<double> tgt((size_t)batch_count[b] * 3);
            if (!check(admm_hip_read_local(gpu, (int)b, nullptr, nullptr, tgt.data(), nullptr))) return false;
            for (int e = 0; e < batch_count[b]; ++e) {
                MovingAnchor *ma = static_cast<MovingAnchor *>(forces[batch_first[b] + e].get());
                if (!ma->point->active) for (int c = 0; c < 3; ++c) ma->point->pos[c] = tgt[3 * (size_t)e + c];
            }
        }
        elapsed_s += settings.timestep_s;
        return true;
    }

    // System.cpp:159-179: Force::weight was edited by the caller
    void recompute_weights() {
        if (!initialized) return;
        for (size_t b = 0; b < batch_first.size(); ++b) {
            const int ne = batch_kind[b] == ADMM_KIND_COLLISION ? (int)m_x.size() / 3 : batch_count[b];
            std::vector<double> w(ne);
            for (int e = 0; e < ne; ++e) w[e] = forces[batch_first[b] + (batch_kind[b] == ADMM_KIND_COLLISION ? 0 : e)]->weight;
            if (!check(admm_hip_set_weights(gpu, (int)b, w.data()))) return;
        }
        check(admm_hip_recompute_weights(gpu));
    }

    admm_hip_ctx *context() { return gpu; }

protected:
    bool initialized;
    admm_hip_ctx *gpu;
    std::vector<int> batch_first, batch_count, batch_kind;
    std::vector<char> batch_moving;

    bool push_shapes(const CollisionForce *cf) {
        std::vector<int32_t> ty; std::vector<double> par;
        for (size_t q = 0; q < cf->collisionShapes.size(); ++q) {
            const CollisionShape &sh = *cf->collisionShapes[q];
            ty.push_back(sh.shape_type());
            par.push_back(sh.center[0]); par.push_back(sh.center[1]); par.push_back(sh.center[2]); par.push_back(sh.shape_radius());
        }
        return check(admm_hip_set_collision_shapes(gpu, (int)ty.size(), ty.data(), par.data()));
    }

    bool check(int rc) {
        if (rc == ADMM_OK) return true;
        std::cerr << "\n**Solver Error (admm_hip " << rc << "): " << (gpu ? admm_hip_last_error(gpu) : "no context") << std::endl;
        return false;
    }
};

} // namespace admm
