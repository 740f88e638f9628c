// System.hpp -- host-side mirror of admm::System
// (deps/admm-elastic-sca/src/system/System.hpp:29-76, System.cpp) over the C ABI
// of libadmm_hip.so.  Same public members and methods:
//   settings{timestep_s, verbose, admm_iters} (+ parse_args/help), elapsed_s,
//   m_x, m_v, m_masses, explicit_forces, forces, add_nodes, initialize, step,
//   recompute_weights, pre_step_callbacks.
// initialize() hands nodes and forces to the library (batches of consecutive
// forces of one kind, in forces[] order), step() runs one frame on the GPU and
// leaves m_x / m_v valid on return, like the reference.  There is no CPU path
// for the built-in forces: without a GPU initialize() prints the library's error
// and returns false.
//
// User-written plug-ins keep working (the reference's extension story,
// Force.hpp:37-57 and samples/singletet.cpp:100-102): a Force subclass is asked
// for its selector rows once (get_selector, System.cpp:121-124); every ADMM
// iteration the device evaluates D_i x for those rows, project() runs here on the
// host (System.cpp:57-58) and z - u returns to the device-side right-hand side.
// A user-written ExplicitForce::project runs here at the start of step().
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "Comm.hpp"
#include "Force.hpp"
#include "admm_hip.h"

namespace admm {

class System {
public:
    System() : elapsed_s(0.0), device_id(0), initialized(false), gpu(nullptr), user_rows(0), pinned_x(nullptr), pinned_v(nullptr), seen_x(nullptr), seen_v(nullptr), pinned_bytes(0), init_count(0) {}
    ~System() { release(); }
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

    // Multi-GPU (no reference counterpart: the reference's element loop, System.cpp:57-58, is one OpenMP team).  One process
    // per GPU, every process builds the SAME System (all nodes, all forces, same order) and sets its rank before initialize():
    // the elements shard across the ranks, the partial right-hand sides meet in one (distributed top: two) small all-reduce(s) per ADMM
    // iteration inside the library, and m_x / m_v are complete on every rank after every step().  Pre-step callbacks, control points and
    // recompute_weights() must run identically on all ranks; under subtree shards initialize() and recompute_weights() are COLLECTIVE calls
    // (every rank factors only its own elimination subtrees + its part of the top; the subtree roots' update matrices meet in one all-reduce).
    struct Shard {
        int rank, world;
        int mode;                        // ADMM_SHARD_SUBTREE (default: ranks own elimination subtrees, small exchange) or ADMM_SHARD_CONTIGUOUS
        bool factor_local;               // subtree shards: every rank factors and keeps its own share (default; false: the whole matrix on every rank, initialize() stays local)
        // RCCL (default transport): rank 0 publishes the communicator's 128-byte id in this file, the others wait for it
        // (Comm.hpp rccl_id_via_file; a path unique to the job, on a filesystem all ranks see)
        std::string rccl_id_file;
        double rendezvous_timeout_s, rendezvous_max_age_s;
        // any other transport instead: a hook that sums a DEVICE buffer (admm_hip_set_allreduce) or a HOST buffer
        // (admm_hip_set_host_allreduce, e.g. comm::ShmAllReduce::hook) in place across the ranks
        admm_hip_allreduce_fn allreduce; void *allreduce_user;
        admm_hip_host_allreduce_fn host_allreduce; void *host_allreduce_user;
        Shard() : rank(0), world(1), mode(ADMM_SHARD_SUBTREE), factor_local(true), rendezvous_timeout_s(600.0), rendezvous_max_age_s(600.0),
                  allreduce(nullptr), allreduce_user(nullptr), host_allreduce(nullptr), host_allreduce_user(nullptr) {}
        // what a launcher exports: torchrun / torch.distributed.run (RANK, WORLD_SIZE, LOCAL_RANK), Open MPI, Slurm;
        // ADMM_HIP_RCCL_ID_FILE names the rendezvous file.  Returns the local rank (the caller's device_id), or -1 if no launcher is seen.
        int from_env() {
            static const char *const rk[] = {"RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID"}, *const wd[] = {"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS"},
                              *const lr[] = {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"};
            for (int i = 0; i < 3; ++i) {
                const char *r = std::getenv(rk[i]), *w = std::getenv(wd[i]), *l = std::getenv(lr[i]);
                if (!r || !w) continue;
                rank = std::atoi(r); world = std::atoi(w);
                if (const char *f = std::getenv("ADMM_HIP_RCCL_ID_FILE")) rccl_id_file = f;
                else if (rccl_id_file.empty()) {
                    // per launch: the launcher's rendezvous port / job id tells two jobs on one node apart (and Comm.hpp's launch_nonce(), stored in the
                    // file, a crashed earlier launch's leftover).  Nothing job-unique in the environment: the name stays empty and initialize()
                    // fails with a message that asks for ADMM_HIP_RCCL_ID_FILE -- never a name two jobs could share.
                    const char *tag = std::getenv("MASTER_PORT"); if (!tag) tag = std::getenv("SLURM_JOB_ID"); if (!tag) tag = std::getenv("OMPI_MCA_ess_base_jobid");
                    if (!tag) tag = std::getenv("PMIX_NAMESPACE");
                    if (tag) rccl_id_file = std::string("/tmp/admm_hip_rccl_id.") + std::to_string((long)getuid()) + "." + tag;
                }
                return l ? std::atoi(l) : rank;
            }
            return -1;
        }
    } shard;

    // System.cpp:78-95
    int add_nodes(VectorXd x, VectorXd m) {
        const int old_n = (int)m_x.size(), add = (int)x.size();
        unpin_state();             // the vectors are about to be re-allocated: never leave a freed buffer page-locked
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
        release();                  // (also un-pins m_x / m_v before they may be re-allocated)
        // an UNCHANGED scene binary under a launcher: ADMM_HIP_RANKS_FROM_ENV=1 makes every process take its rank, world size
        // and GPU from the launcher's environment (mpirun -np 8 -x ADMM_HIP_RANKS_FROM_ENV=1 ./windyflag)
        if (shard.world == 1) { const char *auto_env = std::getenv("ADMM_HIP_RANKS_FROM_ENV"); if (auto_env && std::atoi(auto_env) != 0) { const int local = shard.from_env(); if (local >= 0) device_id = local; } }
        if (m_v.size() < m_x.size()) m_v.resize(m_x.size());
        m_v.setZero();
        if (admm_hip_create(&gpu, device_id) != ADMM_OK) { std::cerr << "\n**Solver Error: no usable HIP device " << device_id << " (the built-in forces have no CPU path)" << std::endl; gpu = nullptr; return false; }
        if (!check(admm_hip_set_timestep(gpu, settings.timestep_s))) return false;
        if (!check(admm_hip_keep_z(gpu, 0))) return false;       // curr_z is not part of the class's public surface: no need to store it every iteration
        if (!setup_shard()) return false;
        if (!check(admm_hip_add_nodes(gpu, dof / 3, m_x.data(), m_masses.data(), nullptr))) return false;
        // Force::initialize of the user-written forces (System.cpp:117-119); the built-in ones compute their rest data in the library
        for (size_t i = 0; i < forces.size(); ++i) {
            Force &f = *forces[i];
            if (f.kind() >= 0 && typeid(f) != f.device_type()) {
                std::cerr << "\n**Solver Error: force " << i << " derives from a built-in force class; a subclass must override kind() to return -1 and bring its own get_selector()/project()" << std::endl;
                return false;
            }
            if (f.kind() < 0) f.initialize(m_x, m_v, m_masses, settings.timestep_s);
        }
        // consecutive forces of one kind (and one anchor flavour) -> one batch, order preserved
        batch_first.clear(); batch_count.clear(); batch_kind.clear(); batch_moving.clear(); batch_urow0.clear();
        user_rows = 0;
        std::vector<double> user_weights;
        for (size_t i = 0; i < forces.size();) {
            const int kind = forces[i]->kind();
            if (kind < 0) {      // a run of user-written forces -> one generic batch
                std::vector<Eigen::Triplet<double> > trips;
                const size_t row0 = user_weights.size();
                std::vector<int32_t> erp(1, 0);
                size_t j = i;
                for (; j < forces.size() && forces[j]->kind() < 0; ++j) {
                    const size_t w0 = user_weights.size(), t0 = trips.size();
                    forces[j]->get_selector(m_x, trips, user_weights);
                    for (size_t t = t0; t < trips.size(); ++t) if (trips[t].row() < (long)w0 || trips[t].row() >= (long)user_weights.size() || trips[t].col() < 0 || trips[t].col() >= dof) {
                        std::cerr << "\n**Solver Error: force " << j << " pushed a selector triplet (row " << trips[t].row() << ", col " << trips[t].col() << ") outside the rows [" << w0 << ", " << user_weights.size() << ") it announced through weights" << std::endl;
                        return false;
                    }
                    erp.push_back((int32_t)(user_weights.size() - row0));
                }
                std::vector<int32_t> tr(trips.size()), tc(trips.size()); std::vector<double> tv(trips.size());
                for (size_t t = 0; t < trips.size(); ++t) { tr[t] = (int32_t)(trips[t].row() - (long)row0); tc[t] = (int32_t)trips[t].col(); tv[t] = trips[t].value(); }
                int b = -1;
                if (!check(admm_hip_add_generic_batch(gpu, (int)(j - i), erp.data(), (int64_t)trips.size(), tr.data(), tc.data(), tv.data(), user_weights.data() + row0, &b))) return false;
                batch_first.push_back((int)i); batch_count.push_back((int)(j - i)); batch_kind.push_back(ADMM_KIND_GENERIC); batch_moving.push_back(false); batch_urow0.push_back((long)row0);
                i = j;
                continue;
            }
            if (kind == ADMM_KIND_COLLISION) {           // one force over all nodes -> one element per node
                CollisionForce *cf = static_cast<CollisionForce *>(forces[i].get());
                const int nn_ = dof / 3;
                std::vector<int32_t> idx(nn_); std::vector<double> par(nn_, cf->weight);
                for (int q = 0; q < nn_; ++q) idx[q] = q;
                int b = -1;
                if (!check(admm_hip_add_batch(gpu, kind, nn_, idx.data(), par.data(), nullptr, &b))) return false;
                cf->n_nodes = nn_; cf->Di_rows = dof;
                if (!push_shapes(cf)) return false;
                batch_first.push_back((int)i); batch_count.push_back(1); batch_kind.push_back(kind); batch_moving.push_back(false); batch_urow0.push_back(-1);
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
            batch_first.push_back((int)i); batch_count.push_back((int)(j - i)); batch_kind.push_back(kind); batch_moving.push_back(moving); batch_urow0.push_back(-1);
            i = j;
        }
        user_rows = (long)user_weights.size();
        if (user_rows && !check(admm_hip_set_project_hook(gpu, &System::project_hook, this))) return false;
        // ExplicitForce / WindForce run on the device; a user-written subclass stays on the host (step())
        explicit_slot.assign(explicit_forces.size(), -1);
        for (size_t i = 0; i < explicit_forces.size(); ++i) {
            const ExplicitForce &ef = *explicit_forces[i];
            if (!ef.on_device()) continue;
            const double d[3] = {ef.direction[0], ef.direction[1], ef.direction[2]};
            const std::vector<int> &il = ef.index_list();
            const int type = ef.explicit_type();
            const int cnt = (int)il.size() / (type == ADMM_EXPLICIT_WIND ? 3 : 1);
            std::vector<int32_t> il32(il.begin(), il.end());
            if (!check(admm_hip_add_explicit(gpu, type, d, cnt, il32.empty() ? nullptr : il32.data(), &explicit_slot[i]))) return false;
        }
        if (!check(admm_hip_finalize(gpu))) return false;
        // write back what Force::initialize / get_selector compute in the reference
        user_local.clear();
        for (size_t b = 0; b < batch_first.size(); ++b) {
            if (batch_kind[b] == ADMM_KIND_GENERIC) {    // this rank's user forces (all of them on one GPU)
                int nl = 0;
                if (!check(admm_hip_local_elements(gpu, (int)b, nullptr, 0, &nl))) return false;
                std::vector<int32_t> ids(nl > 0 ? nl : 1);
                if (!check(admm_hip_local_elements(gpu, (int)b, ids.data(), nl, &nl))) return false;
                for (int q = 0; q < nl; ++q) user_local.push_back(batch_first[b] + ids[q]);
                continue;
            }
            const int ne = batch_kind[b] == ADMM_KIND_COLLISION ? dof / 3 : batch_count[b];
            std::vector<double> w(ne), rest((size_t)ne * 12); std::vector<int32_t> g(ne);
            if (!check(admm_hip_read_rest(gpu, (int)b, w.data(), rest.data(), g.data()))) return false;
            if (batch_kind[b] == ADMM_KIND_COLLISION) { forces[batch_first[b]]->global_idx = g[0]; continue; }   // weight stays use_weight
            for (int e = 0; e < batch_count[b]; ++e) { Force *f = forces[batch_first[b] + e].get(); f->weight = w[e]; f->global_idx = g[e]; }
        }
        user_Dx.resize(user_rows); user_u.resize(user_rows); user_z.resize(user_rows);
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
            std::vector<double> &tgt = anchor_tgt; std::vector<int32_t> &act = anchor_act;      // members: no allocation per frame
            tgt.resize((size_t)batch_count[b] * 3); act.resize(batch_count[b]);
            for (int e = 0; e < batch_count[b]; ++e) {
                const MovingAnchor *ma = static_cast<const MovingAnchor *>(forces[batch_first[b] + e].get());
                for (int c = 0; c < 3; ++c) tgt[3 * (size_t)e + c] = ma->point->pos[c];
                act[e] = ma->point->active ? 1 : 0;
            }
            if (!check(admm_hip_update_anchors(gpu, (int)b, tgt.data(), act.data()))) return false;
        }
        // user-written explicit forces act on the host copy before it is uploaded (System.cpp:37-39; they run before the
        // device-side ones, in their own relative order)
        for (size_t i = 0; i < explicit_forces.size(); ++i) {
            if (explicit_slot[i] < 0) { explicit_forces[i]->project(settings.timestep_s, m_x, m_v, m_masses); continue; }
            const Vector3d &d = explicit_forces[i]->direction;
            if (!check(admm_hip_set_gravity(gpu, explicit_slot[i], d[0], d[1], d[2]))) return false;
        }
        for (size_t b = 0; b < batch_first.size(); ++b) if (batch_kind[b] == ADMM_KIND_COLLISION && !push_shapes(static_cast<CollisionForce *>(forces[batch_first[b]].get()))) return false;
        // m_x / m_v are public and may have been edited by the caller between steps: they travel every frame, one DMA each
        // way per vector out of / into the vectors' own (page-locked) memory
        pin_state();
        if (!check(admm_hip_upload_state(gpu, m_x.data(), m_v.data()))) return false;
        if (!check(admm_hip_set_tolerance(gpu, settings.residual_tol_primal, settings.residual_tol_dual, settings.residual_check_every < 1 ? 1 : settings.residual_check_every))) return false;
        if (!check(admm_hip_step(gpu, settings.admm_iters))) return false;
        if (!check(admm_hip_download_state(gpu, m_x.data(), m_v.data()))) return false;
        // released MovingAnchors follow their node: point->pos = Dx (AnchorForce.cpp:80-83)
        for (size_t b = 0; b < batch_first.size(); ++b) if (batch_moving[b]) {
            bool any_released = false;
            for (int e = 0; e < batch_count[b] && !any_released; ++e) any_released = !static_cast<MovingAnchor *>(forces[batch_first[b] + e].get())->point->active;
            if (!any_released) continue;
            if (shard.world > 1) {
                // sharded: a rank holds the projection state of its own anchors only.  The owner's Dx of the last project() (the
                // reference's value, AnchorForce.cpp:80-83) + zeros elsewhere, summed over the ranks through the iterations' own
                // transport: the control points are the single-GPU run's on every rank
                std::vector<double> &all = anchor_tgt;
                all.assign((size_t)batch_count[b] * 3, 0.0);
                anchor_ids.resize((size_t)batch_count[b]);
                int n_local = 0;
                if (!check(admm_hip_local_elements(gpu, (int)b, anchor_ids.data(), batch_count[b], &n_local))) return false;
                anchor_loc.resize((size_t)(n_local > 0 ? n_local : 1) * 3);
                if (n_local > 0 && !check(admm_hip_read_local(gpu, (int)b, nullptr, nullptr, anchor_loc.data(), nullptr))) return false;
                for (int l = 0; l < n_local; ++l) for (int c = 0; c < 3; ++c) all[3 * (size_t)anchor_ids[l] + c] = anchor_loc[3 * (size_t)l + c];
                if (!check(admm_hip_allreduce_host(gpu, all.data(), (int64_t)all.size()))) return false;
                for (int e = 0; e < batch_count[b]; ++e) {
                    MovingAnchor *ma = static_cast<MovingAnchor *>(forces[batch_first[b] + e].get());
                    if (!ma->point->active) for (int c = 0; c < 3; ++c) ma->point->pos[c] = all[3 * (size_t)e + c];
                }
                continue;
            }
            std::vector<double> &tgt = anchor_tgt;
            tgt.resize((size_t)batch_count[b] * 3);
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
        // user forces announce their weights through get_selector again (System.cpp:165-168); same call order -> same global_idx
        std::vector<Eigen::Triplet<double> > trips; std::vector<double> user_weights;
        for (size_t i = 0; i < forces.size(); ++i) if (forces[i]->kind() < 0) forces[i]->get_selector(m_x, trips, user_weights);
        if ((long)user_weights.size() != user_rows) { std::cerr << "\n**Solver Error: user forces changed their row count in recompute_weights" << std::endl; return; }
        for (size_t b = 0; b < batch_first.size(); ++b) {
            if (batch_kind[b] == ADMM_KIND_GENERIC) {
                if (!check(admm_hip_set_weights(gpu, (int)b, user_weights.data() + batch_urow0[b]))) return;
                continue;
            }
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
    std::vector<long> batch_urow0;         // generic batches: first of their rows among the user rows (-1 otherwise)
    std::vector<int> explicit_slot;        // explicit_forces[i] -> index among the device-side explicit forces, -1 = host
    // user-written forces: their rows of Dx / u / z (the vectors Force::project receives) and this rank's forces
    long user_rows;
    std::vector<int> user_local;
    VectorXd user_Dx, user_u, user_z;
    void *pinned_x, *pinned_v, *seen_x, *seen_v; size_t pinned_bytes;
    std::vector<double> anchor_tgt, anchor_loc; std::vector<int32_t> anchor_act, anchor_ids;     // step(): this frame's control points
    int init_count;                        // initialize() calls so far (every rank counts alike: part of the rendezvous file's name)

    // world > 1: hand rank / world / mode to the library and install the all-reduce -- a caller's hook, or (default) an RCCL
    // communicator inside the library, its id bootstrapped through shard.rccl_id_file
    bool setup_shard() {
        ++init_count;
        if (shard.world <= 1) return true;
        if (shard.rank < 0 || shard.rank >= shard.world) { std::cerr << "\n**Solver Error: shard.rank " << shard.rank << " outside [0, " << shard.world << ")" << std::endl; return false; }
        if (!check(admm_hip_set_shard(gpu, shard.rank, shard.world)) || !check(admm_hip_set_shard_mode(gpu, shard.mode)) || !check(admm_hip_set_factor_local(gpu, shard.factor_local ? 1 : 0))) return false;
        if (shard.host_allreduce) return check(admm_hip_set_host_allreduce(gpu, shard.host_allreduce, shard.host_allreduce_user));
        if (shard.allreduce) return check(admm_hip_set_allreduce(gpu, shard.allreduce, shard.allreduce_user));
        unsigned char id[128];
        std::memset(id, 0, sizeof id);
        if (shard.rank == 0 && !check(admm_hip_rccl_unique_id(id))) return false;
        const std::string file = shard.rccl_id_file.empty() ? std::string() : shard.rccl_id_file + "." + std::to_string(init_count);
        std::string why;
        if (!comm::rccl_id_via_file(file, shard.rank, id, shard.rendezvous_timeout_s, shard.rendezvous_max_age_s, &why)) {
            std::cerr << "\n**Solver Error: multi-GPU rendezvous failed: " << why << " (set System::shard.rccl_id_file / ADMM_HIP_RCCL_ID_FILE to a path unique to this launch, or install an all-reduce hook)" << std::endl;
            return false;
        }
        const bool ok = check(admm_hip_rccl_init(gpu, id, shard.rank, shard.world));     // collective: returns once every rank has joined
        if (shard.rank == 0) std::remove(file.c_str());
        return ok;
    }

    // Force::project for every user force of this rank (System.cpp:57-58), on the rows the library hands over
    static int project_hook(void *self_, double dt, int64_t n_rows, const double *Dx, double *u, double *z) {
        System *self = static_cast<System *>(self_);
        if (n_rows != (int64_t)self->user_rows) return 1;
        const size_t bytes = sizeof(double) * (size_t)n_rows;
        std::memcpy(self->user_Dx.data(), Dx, bytes); std::memcpy(self->user_u.data(), u, bytes); std::memcpy(self->user_z.data(), z, bytes);
        const int nf = (int)self->user_local.size();
        // a team sized to the work (one thread per 512 forces, at most 16): this runs once per ADMM iteration between two GPU
        // phases, and the default team -- every hardware thread of the host, 128+ -- costs milliseconds to wake for microseconds of work
        int team = nf / 512; team = team > 16 ? 16 : (team < 1 ? 1 : team);
#pragma omp parallel for num_threads(team) if (team > 1) schedule(static)
        for (int i = 0; i < nf; ++i) self->forces[self->user_local[i]]->project(dt, self->user_Dx, self->user_u, self->user_z);
        std::memcpy(u, self->user_u.data(), bytes); std::memcpy(z, self->user_z.data(), bytes);
        return 0;
    }

    // page-lock m_x / m_v where they lie (again if the caller resized them)
    void pin_state() {
        const size_t bytes = sizeof(double) * (size_t)m_x.size();
        if (seen_x == (void *)m_x.data() && seen_v == (void *)m_v.data() && pinned_bytes == bytes) return;
        unpin_state();
        seen_x = m_x.data(); seen_v = m_v.data(); pinned_bytes = bytes;       // tried once per buffer: pageable memory still works, just slower
        if (admm_hip_pin_host(gpu, m_x.data(), bytes, 1) == ADMM_OK) pinned_x = m_x.data();
        if (admm_hip_pin_host(gpu, m_v.data(), bytes, 1) == ADMM_OK) pinned_v = m_v.data();
    }
    void unpin_state() {
        if (gpu && pinned_x) admm_hip_pin_host(gpu, pinned_x, pinned_bytes, 0);
        if (gpu && pinned_v) admm_hip_pin_host(gpu, pinned_v, pinned_bytes, 0);
        pinned_x = pinned_v = seen_x = seen_v = nullptr; pinned_bytes = 0;
    }
    void release() {
        unpin_state();
        if (gpu) { admm_hip_destroy(gpu); gpu = nullptr; }
        initialized = false;
    }

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
