/*
 * admm_hip.h -- C ABI of the MI355X-native ADMM elastic solver (libadmm_hip.so).
 *
 * This is the drop-in boundary for the hot path of mattoverby/admm-elastic-sca:
 * everything admm::System::initialize()/step() does between "forces and nodes
 * are known" and "m_x/m_v hold the new state" (reference
 * deps/admm-elastic-sca/src/system/System.cpp:26-75 and :98-179) runs behind
 * these entry points, on one GPU per context.  The reference itself has no
 * FFI: its plugin surface is the C++ classes admm::System / admm::Force.  The
 * host-side mirror of those classes (admm-elastic-sca_amd/host/admm/)
 * binds to exactly the functions declared here, and so does the Python
 * plumbing used by bench.py and tests/.  Plain C types only, caller-owned
 * host buffers, int error codes (0 = ok), no exceptions across the boundary.
 *
 * Every entry point cites the reference interface it replaces.
 */
#ifndef ADMM_HIP_H
#define ADMM_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "admm_kinds.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct admm_hip_ctx admm_hip_ctx;

enum admm_hip_err {
    ADMM_OK = 0,
    ADMM_ERR_ARG = 1,        /* bad argument / call order                       */
    ADMM_ERR_HIP = 2,        /* a HIP runtime call failed (no GPU, OOM, ...)    */
    ADMM_ERR_STATE = 3,      /* not finalized / already finalized               */
    ADMM_ERR_UNSUPPORTED = 4,/* input outside the accelerated path              */
    ADMM_ERR_FACTOR = 5,     /* matrix not positive definite                    */
    ADMM_ERR_COMM = 6        /* all-reduce hook failed                          */
};

/* ---- lifetime -------------------------------------------------------------
 * replaces: admm::System::System() / ~System()            (System.hpp:31)
 * device_id < 0: host-only context (assembly + factorization work, every
 * device call returns ADMM_ERR_HIP) -- used by the CPU test-suite.          */
int  admm_hip_create(admm_hip_ctx **out, int device_id);
void admm_hip_destroy(admm_hip_ctx *ctx);
const char *admm_hip_last_error(const admm_hip_ctx *ctx);
/* run on an existing hipStream_t (e.g. torch's current stream); NULL = own stream */
int  admm_hip_set_stream(admm_hip_ctx *ctx, void *hip_stream);

/* ---- settings -------------------------------------------------------------
 * replaces: System::settings.timestep_s / admm_iters      (System.hpp:36-44)
 * dt <= 0 is repaired to 0.04 like System::initialize     (System.cpp:103-107) */
int admm_hip_set_timestep(admm_hip_ctx *ctx, double dt);

/* ---- nodes ----------------------------------------------------------------
 * replaces: System::add_nodes(x, m)                       (System.cpp:78-95)
 * x, m: [3*n_nodes] xyz-interleaved like m_x / m_masses; velocities start at 0.
 * May be called several times before finalize; returns total node count in *total. */
int admm_hip_add_nodes(admm_hip_ctx *ctx, int n_nodes, const double *x, const double *m, int *total);

/* ---- forces ---------------------------------------------------------------
 * replaces: system->forces.push_back(new <Force>(...)) for n_elems elements of
 * one kind, in order                                      (System.hpp:52;
 * constructors: Force.hpp:65, TetForce.hpp:33,54,118, TriangleForce.hpp:32,
 * BendForce.hpp:31, AnchorForce.hpp:57,88).
 * idx    : [n_elems][ADMM_KIND_NODES[kind]] node ids
 * params : [n_elems][ADMM_KIND_PARAMS[kind]]  (layout in admm_kinds.h)
 * targets: ANCHOR only, [n_elems][3] control-point positions for MovingAnchor
 *          semantics, or NULL for StaticAnchor (target = x at finalize).
 * The order of add_batch calls and of elements inside a batch is the order of
 * system->forces; it defines global_idx exactly as Force::get_selector does.
 * Returns the batch id in *batch. */
int admm_hip_add_batch(admm_hip_ctx *ctx, int kind, int n_elems, const int32_t *idx,
                       const double *params, const double *targets, int *batch);

/* ---- user-defined forces ----------------------------------------------------
 * replaces: system->forces.push_back(new <user subclass of admm::Force>): the reference's extension
 * story is "subclass admm::Force, implement get_selector + project, push it into system->forces"
 * (Force.hpp:37-57; documented at samples/singletet.cpp:100-102; System.cpp:121-124 collects the
 * selector triplets and weights, System.cpp:57-58 calls project once per ADMM iteration).
 * A generic batch is a run of consecutive user forces ("elements"): element e owns the batch rows
 * [elem_row_ptr[e], elem_row_ptr[e+1]); its selector rows come as triplets (row relative to the batch,
 * col = 3*node + component like the reference's D, value), duplicates are summed; row_weight [n_rows]
 * is what get_selector pushed into `weights`.  The batch takes its place in the order of add_batch
 * calls.  The library needs  dt^2 D^T W^2 D = K (x) I3  for every element (true whenever a row touches
 * one coordinate and the x/y/z rows look alike, as for every force of the reference); anything else is
 * refused at finalize with ADMM_ERR_UNSUPPORTED.
 * Per ADMM iteration the device evaluates D_i x for the generic rows, the rows travel to the host and
 * the hook runs the user's project() there -- user code is host code --, then z - u returns to the
 * device and joins the right-hand side through the same per-node slots as every other force.
 * The hook mirrors Force::project(dt, Dx, u, z): Dx, u, z cover ALL generic rows of the context
 * (generic batches concatenated in add order: a force's offset is the `weights.size()` it saw in
 * get_selector when only user forces push weights); it must update u and z of this rank's elements
 * (admm_hip_local_elements) and leave the rest alone.  u starts at 0 and persists; z is D*m_x at the
 * start of every frame (System.cpp:43).  No HIP graph with generic batches (host code inside the iteration).   */
typedef int (*admm_hip_project_fn)(void *user, double dt, int64_t n_rows, const double *Dx, double *u, double *z);
int admm_hip_add_generic_batch(admm_hip_ctx *ctx, int n_elems, const int32_t *elem_row_ptr, int64_t n_triplets,
                               const int32_t *trip_row, const int32_t *trip_col, const double *trip_val,
                               const double *row_weight, int *batch);
int admm_hip_set_project_hook(admm_hip_ctx *ctx, admm_hip_project_fn fn, void *user);

/* replaces: system->explicit_forces.push_back(new ExplicitForce(dir))
 * (ExplicitForce.hpp:51-59, ExplicitForce.cpp:29-39): v += dt*dir on all nodes,
 * once per frame before the ADMM loop.                                       */
int admm_hip_add_gravity(admm_hip_ctx *ctx, double gx, double gy, double gz);
/* general form: replaces explicit_forces.push_back(new ExplicitForce(dir, indices)) and
 * new WindForce(tris) (ExplicitForce.hpp:51-71).  type ADMM_EXPLICIT_CONST: idx = node ids
 * (n_idx = 0: all nodes); ADMM_EXPLICIT_WIND: idx = [n_idx][3] triangle node ids, dir = wind
 * direction.  Explicit forces are applied in the order they were added.  The reference's
 * wind loop is an omp-parallel loop that reads velocities other threads are updating and
 * scatters under an omp critical, so its result depends on the thread schedule; this library
 * reproduces the loop run serially (OMP_NUM_THREADS=1): triangle i sees the increments of the
 * triangles before it, every node is incremented in triangle order (deterministic).        */
int admm_hip_add_explicit(admm_hip_ctx *ctx, int type, const double *dir, int n_idx, const int32_t *idx, int *which);
/* replaces: CollisionForce::collisionShapes (CollisionForce.hpp:39): the shape table used by
 * every ADMM_KIND_COLLISION batch, tested in order like CollisionForce::handleCollisions
 * (CollisionForce.cpp:55-70).  types [n], params [n][4] (admm_kinds.h).  May be updated between frames. */
int admm_hip_set_collision_shapes(admm_hip_ctx *ctx, int n_shapes, const int32_t *types, const double *params);

/* ---- multi-GPU ------------------------------------------------------------
 * Elements shard across ranks (see admm_hip_set_shard_mode); must be called before finalize.  The hook must sum `count`
 * doubles of a DEVICE buffer in place across the ranks (an all-reduce), ordered on `stream`; every rank makes the same
 * calls in the same order, buffers and counts differ between calls: per ADMM iteration once (contiguous shards: the
 * whole right-hand side; subtree shards: the top rows) or twice (distributed top: also the top's x), once per frame
 * (subtree shards: the full x), and inside admm_hip_finalize / admm_hip_recompute_weights under rank-local factorization
 * (admm_hip_set_factor_local).  No reference counterpart (the reference is single-process).   */
typedef int (*admm_hip_allreduce_fn)(void *user, void *dev_buf, int64_t count, void *hip_stream);
int admm_hip_set_shard(admm_hip_ctx *ctx, int rank, int world);
int admm_hip_set_allreduce(admm_hip_ctx *ctx, admm_hip_allreduce_fn fn, void *user);
/* Transports that only see HOST memory (MPI without GPU support, shared memory between the ranks of one node -- see
 * host/admm/Comm.hpp ShmAllReduce): the library stages the buffer through pinned host memory around fn, which must sum
 * host_buf[0..count) in place across the ranks.  Replaces any device hook; NULL removes it.                          */
typedef int (*admm_hip_host_allreduce_fn)(void *user, double *host_buf, int64_t count);
int admm_hip_set_host_allreduce(admm_hip_ctx *ctx, admm_hip_host_allreduce_fn fn, void *user);
/* RCCL inside the library (north_star: "host stays C++"): with a communicator installed the per-iteration exchange is
 * ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, <the context's stream>) issued by the step loop itself -- no
 * host hook between the kernels, and (ADMM_HIP_GRAPH_COMM=1) the whole multi-GPU iteration replays as one HIP graph.
 * librccl.so is bound at run time with dlopen (the copy already loaded in the process first, e.g. PyTorch's; env
 * ADMM_HIP_RCCL_LIB overrides); single-GPU users never load it.
 *   admm_hip_rccl_unique_id  rank 0: 128 bytes (ncclUniqueId) to hand to every rank by whatever means the host has
 *   admm_hip_rccl_init       every rank, collectively: ncclCommInitRank on the context's device; the library owns the communicator
 *   admm_hip_set_rccl_comm   use an ncclComm_t the caller already has (not owned; NULL = back to the hook)
 *   admm_hip_debug_allreduce one checked all-reduce of a caller-owned device buffer through whatever is installed
 *   admm_hip_rccl_async_error  ncclCommGetAsyncError of the installed communicator: *nccl_result = 0 (ncclSuccess) while the
 *                            communicator is healthy; a non-zero value also makes the call return ADMM_ERR_COMM with the RCCL
 *                            error text in admm_hip_last_error.  admm_hip_step polls it once per frame when a communicator is
 *                            installed, so a peer that died surfaces as a failed step instead of a silent hang in the next sync.
 *                            No communicator installed: *nccl_result = 0, ADMM_OK.                                             */
int admm_hip_rccl_unique_id(void *id128);
int admm_hip_rccl_init(admm_hip_ctx *ctx, const void *id128, int rank, int world);
int admm_hip_set_rccl_comm(admm_hip_ctx *ctx, void *nccl_comm);
int admm_hip_debug_allreduce(admm_hip_ctx *ctx, void *dev_buf, int64_t count);
int admm_hip_rccl_async_error(admm_hip_ctx *ctx, int *nccl_result);
/* A short HOST vector summed in place across the ranks through the same transport (no-op at world 1).  The class mirror uses
 * it for what the reference keeps per force object and a sharded run keeps on the owner rank only: a released MovingAnchor's
 * position, point->pos = Dx of the last project() (AnchorForce.cpp:80-83) -- owner's value, zeros elsewhere.              */
int admm_hip_allreduce_host(admm_hip_ctx *ctx, double *host_buf, int64_t count);
/* How the work is split across the ranks (before finalize; env ADMM_HIP_SHARD=contiguous|subtree overrides):
 *   ADMM_SHARD_CONTIGUOUS  every batch is cut into `world` contiguous element ranges; per ADMM iteration the whole right-hand
 *                          side (3 n doubles) is all-reduced and every rank runs the complete solve.
 *   ADMM_SHARD_SUBTREE     the elimination tree is cut below its top: every rank owns whole subtrees and the elements touching
 *                          them (an element's nodes lie in one subtree plus separators above it); per iteration ONE small
 *                          all-reduce carries the top separators' partial right-hand sides and the subtree roots' contributions,
 *                          only the top levels of the solve are replicated, and the full x is rebuilt once per frame.  With
 *                          2 / 4 / 8 / 16 ranks and >= 300k nodes (ADMM_HIP_DIST_TOP) the top is ONE root supernode whose product
 *                          with its explicit inverse is split by rows across the ranks: nothing of the solve is replicated, a
 *                          second small all-reduce per iteration gathers the top's x (admm_hip_info.dist_top).
 * admm_hip_local_elements: this rank's elements of a batch (ascending reference order) -- the order of read_local / write_local. */
enum { ADMM_SHARD_CONTIGUOUS = 0, ADMM_SHARD_SUBTREE = 1 };
int admm_hip_set_shard_mode(admm_hip_ctx *ctx, int mode);
int admm_hip_local_elements(admm_hip_ctx *ctx, int batch, int32_t *ids, int capacity, int *n_local);
/* test hook: the rank that owns each node's subtree (original node order), -1 = replicated top; 0 everywhere without subtree sharding */
int admm_hip_debug_node_owner(admm_hip_ctx *ctx, int32_t *owner);

/* ---- initialize -----------------------------------------------------------
 * replaces: System::initialize()                          (System.cpp:98-156)
 * Force::initialize + get_selector for every element (rest shape matrices,
 * weights, global_idx), assembly of A = M + dt^2 D^T W^2 D, its sparse
 * factorization (host) and the upload of every device-resident array.        */
int admm_hip_finalize(admm_hip_ctx *ctx);
/* Small systems (n_nodes <= ADMM_HIP_DENSE_MAX, default 2048; env, 0 disables): the factor is used once on the host to
 * form the dense A_s^-1 (n x n, <= 33 MB) and every solve becomes ONE kernel, x = A_s^-1 b, instead of ~2 launches per
 * elimination-tree level whose dependent latencies dominate at this size (the reference's shipped scenes have 777-1251 nodes). */

/* replaces: System::recompute_weights()                   (System.cpp:159-179)
 * after admm_hip_set_weights changed per-element weights: re-assemble, re-factor, re-upload. */
/* weights: [n_elems] (a generic batch: [n_rows], one per selector row as get_selector pushes them) */
int admm_hip_set_weights(admm_hip_ctx *ctx, int batch, const double *weights);
int admm_hip_recompute_weights(admm_hip_ctx *ctx);

/* ---- per-frame host-mutable parameters --------------------------------------
 * replaces: writes to ControlPoint::pos / ::active from callbacks
 * (AnchorForce.hpp:71-80; samples/poordillo/poordillo.cpp:196-248)           */
int admm_hip_update_anchors(admm_hip_ctx *ctx, int batch, const double *targets, const int32_t *active);
/* replaces: writes to ExplicitForce::direction (samples/windyflag/windyflag.cpp:141-152) */
int admm_hip_set_gravity(admm_hip_ctx *ctx, int which, double gx, double gy, double gz);

/* ---- step -----------------------------------------------------------------
 * replaces: System::step()                                (System.cpp:26-75)
 * one frame: explicit forces, x_bar, admm_iters x (local step, RHS, solve),
 * velocity update.  Asynchronous on the context's stream; admm_hip_sync or any
 * get_* call waits for it.  pre_step_callbacks stay on the host side
 * (host/admm/System.hpp runs them before calling this).                      */
int admm_hip_step(admm_hip_ctx *ctx, int admm_iters);
int admm_hip_sync(admm_hip_ctx *ctx);

/* ---- state access ---------------------------------------------------------
 * replaces: reads/writes of System::m_x, m_v              (System.hpp:47-49) */
int admm_hip_get_x(admm_hip_ctx *ctx, double *x);
int admm_hip_set_x(admm_hip_ctx *ctx, const double *x);
int admm_hip_get_v(admm_hip_ctx *ctx, double *v);
int admm_hip_set_v(admm_hip_ctx *ctx, const double *v);

/* The frame boundary of the class API: what host/admm/System.hpp does around admm_hip_step because m_x / m_v are
 * public members a caller may read or edit between steps (System.hpp:47-49; samples/singletet.cpp:44 writes m_x).
 * upload_state is asynchronous on the context's stream (x, v or both; NULL = leave the device copy), download_state
 * returns when both vectors have arrived.  admm_hip_pin_host page-locks (on = 1) or releases (on = 0) a caller buffer;
 * when both vectors travel and both are page-locked, ONE kernel each way addresses them directly over PCIe (linear on the
 * host side, the reordering to the factor's node order on the device side; ADMM_HIP_STATE_ZEROCOPY=0: one DMA per vector
 * + reordering kernels, also the path of pageable memory).  The caller must not touch x / v between upload_state and the
 * next synchronising call (admm_hip_download_state, admm_hip_sync): the kernel reads them asynchronously.  Systems of up to 12 288
 * nodes (ADMM_HIP_STATE_DIRECT), when both vectors travel: no DMA at all -- the vectors are copied by the host into / out of a
 * page-locked buffer of the context that the permutation kernels address directly (the DMAs' submission latency is most
 * of a small scene's frame boundary); upload_state has then read x and v when it returns.                          */
/* (a refused registration returns ADMM_ERR_HIP without touching last_error or HIP's sticky error: the caller may go on
 * with pageable memory)                                                                                               */
int admm_hip_pin_host(admm_hip_ctx *ctx, void *p, size_t bytes, int on);
int admm_hip_upload_state(admm_hip_ctx *ctx, const double *x, const double *v);
int admm_hip_download_state(admm_hip_ctx *ctx, double *x, double *v);

/* ---- parity / introspection ------------------------------------------------
 * u, z: [n_local_elems][rows] element-major (compact rows), state:
 * [n_local_elems][ADMM_KIND_STATE]; n_iters: L-BFGS outer iterations of the last
 * project (hyperelastic kinds).  Any pointer may be NULL.  Replaces the
 * protected System::curr_u / curr_z (System.hpp:98-99) and
 * HyperElasticTet::last_prox_result (TetForce.hpp:146).                      */
int admm_hip_read_local(admm_hip_ctx *ctx, int batch, double *u, double *z, double *state, int32_t *n_iters);
/* z of the tet batches is an output nobody reads back in a plain frame -- every project() overwrites it from Dx + u (the
 * reference's curr_z is a protected member, System.hpp:98-99).  admm_hip_keep_z(ctx, 0) stops admm_hip_step from storing
 * it (72 bytes per tet and ADMM iteration less; what host/admm/System.hpp and bench.py do); read_local's z is then the
 * value of the last call that kept it.  Default: kept.  The parity entry points below and residual tracking always keep it. */
int admm_hip_keep_z(admm_hip_ctx *ctx, int on);
int admm_hip_write_local(admm_hip_ctx *ctx, int batch, const double *u, const double *state);
/* rest data computed by Force::initialize: weight [n], rest [n][12] (tets: B 4x3
 * col-major; tris: B 3x2 in the first 6; bend: alpha[4]; spring: rest length),
 * global_idx [n] (compact row of the element's first row).                    */
int admm_hip_read_rest(admm_hip_ctx *ctx, int batch, double *weight, double *rest, int32_t *global_idx);

/* one local step on caller-supplied positions (no global step): runs the batch
 * kernels on x_cur = x and returns; used by the per-project parity tests.     */
int admm_hip_local_step_only(admm_hip_ctx *ctx, const double *x_cur);
/* one project() of every local element of `batch` on caller-supplied D_i x rows
 * (element-major [n_local][rows]) instead of the gather: replays the per-project
 * golden tuples captured from the reference (tests/golden/project_*.npz).       */
int admm_hip_local_step_dx(admm_hip_ctx *ctx, int batch, const double *dx);
/* solves A X = B for B = [n_nodes][3] on the device factor (parity tests).    */
int admm_hip_solve_only(admm_hip_ctx *ctx, const double *b, double *x);
/* host-side product with the assembled scalar matrix: y = A_s * x, x,y [n][3] */
int admm_hip_apply_A(admm_hip_ctx *ctx, const double *x, double *y);
/* CPU evaluation of the factor's two panel sweeps -- validation hook for the
 * CPU test-suite only; no product path calls it.                               */
int admm_hip_debug_panel_solve_host(admm_hip_ctx *ctx, const double *b, double *x);

/* the device's log() / exp() (glibc's algorithms restated, local_math.hpp admm_log / admm_exp) applied to n doubles:
 * op 0 = log, 1 = exp.  Parity tests compare them bit for bit with the host's libm.                                */
int admm_hip_debug_math(admm_hip_ctx *ctx, int op, int64_t n, const double *in, double *out);
/* Unit-test hooks for the kernels of the device factorization (csrc/factor_dev.hpp), host arrays in and out, column-major:
 *  gemm:      C (m x n, ldc) = beta C + alpha opA(A) opB(B) through gemm_f64_kernel; flags: 1 A transposed (stored k x m), 2 B transposed
 *             (stored n x k), 4 only tiles on / below the diagonal, 8 sum from the tile's first column, 16 from max(tile row, tile column);
 *             size_a / size_b / size_c: doubles in the arrays.
 *  potrf_inv: blk (w x w, ld, w <= 64; lower triangle read) <- its Cholesky factor, out (w x w, ld) <- the factor's inverse;
 *             returns ADMM_ERR_FACTOR when a pivot is not positive.                                                              */
int admm_hip_debug_gemm(admm_hip_ctx *ctx, int m, int n, int k, int lda, int ldb, int ldc, int flags, double alpha, double beta,
                        const double *A, int64_t size_a, const double *B, int64_t size_b, double *C, int64_t size_c);
int admm_hip_debug_potrf_inv(admm_hip_ctx *ctx, int w, int ld, double *blk, double *out);
/* launch mode of the last admm_hip_step (tests: "was the multi-GPU iteration really replayed as a graph?"):
 * *iter_graph = 1 when a per-iteration HIP graph exists, *frame_graph = ADMM iterations of the whole-frame graph (0: none),
 * *graph_launches = graph launches issued by the context so far.  Any pointer may be NULL.                               */
int admm_hip_debug_graph_state(admm_hip_ctx *ctx, int *iter_graph, int *frame_graph, int64_t *graph_launches);

typedef struct admm_hip_info {
    int64_t n_nodes, n_elems_total, n_elems_local, rows_compact;
    int64_t nnz_A;            /* scalar n x n system, lower triangle          */
    int64_t nnz_L;            /* entries of the supernodal factor panels read per triangular sweep */
    int64_t panel_bytes;      /* device bytes of the factor panels            */
    int64_t n_supernodes, n_levels, max_super_cols, max_super_rows;
    int64_t solve_contrib_rows; /* sum of below-diagonal block rows            */
    double  t_order_s, t_symbolic_s, t_numeric_s, t_upload_s; /* finalize phases */
    int32_t rank, world, device_id, host_threads;
    int32_t dense_solve;      /* 1: small system, solved as x = A_s^-1 b with the explicit inverse (see admm_hip_finalize) */
    int32_t device_factor;    /* 1: the numeric factorization ran on the GPU (csrc/factor_dev.hpp), 0: on the host */
    int64_t rhs_slots;        /* 24-byte slots the local kernels write and the RHS gather reads per ADMM iteration (this rank) */
    /* sharding: what THIS rank's sweeps stream and what it exchanges (one rank / contiguous shards: own = the whole factor, top = 0).
     * Entries are counted like nnz_L: k(k+1)/2 + r k per supernode.                                                          */
    int64_t sweep_entries_own;      /* supernodes of this rank's own subtrees (both sweeps)                                  */
    int64_t sweep_entries_top;      /* the replicated top of the tree (forward sweep: all of it, on every rank)              */
    int64_t sweep_entries_top_bwd;  /* the part of the top this rank's backward sweep covers (separators it reads + ancestors) */
    int64_t nodes_own, nodes_top;   /* nodes of the own subtrees / of the replicated top                                     */
    int64_t comm_doubles_iter;      /* doubles summed across the ranks per ADMM iteration (one collective; two with dist_top) */
    int64_t comm_doubles_frame;     /* additionally once per frame (subtree shards: the full x before the velocity update)   */
    /* rank-local factorization (subtree shards, admm_hip_set_factor_local): what THIS rank factors and keeps on its device   */
    int64_t factor_doubles_resident; /* doubles of factor panels (+ root inverses) resident on this rank's device; one rank / contiguous shards /
                                        factor_local = 0: the whole factor = panel_bytes / 8                                     */
    int64_t front_doubles;           /* doubles of frontal matrices this rank's numeric factorization held at once (0: factored on the host) */
    int64_t factor_exchange_doubles; /* doubles summed across the ranks ONCE per factorization (the subtree roots' update matrices) */
    int32_t factor_local;            /* 1: this rank factored only its own subtrees + the replicated top                          */
    int32_t dist_top;                /* 1: distributed top -- the top of the tree is ONE root supernode whose product with its explicit inverse is split
                                        by rows across the ranks (subtree shards of 2 / 4 / 8 / 16 ranks; ADMM_HIP_DIST_TOP=0: replicated top): two small
                                        collectives per ADMM iteration instead of one, no replicated sweep                                         */
} admm_hip_info;
int admm_hip_get_info(admm_hip_ctx *ctx, admm_hip_info *info);
/* Rank-local factorization (default on; ADMM_HIP_FACTOR_LOCAL=0 / 1 overrides).  Under subtree sharding with world > 1 a rank sweeps
 * only its own subtrees and the replicated top of the elimination tree, so that is all it assembles, factors and keeps on its device:
 * System::initialize's ONE solver.compute(A) (System.cpp:138-140) is split N ways instead of repeated N times.  The update matrices
 * of the subtree roots meet in ONE all-reduce through the installed transport (owner's values + zeros elsewhere), after which every
 * rank factors the top from the same bits.  That makes admm_hip_finalize and admm_hip_recompute_weights COLLECTIVE calls in this
 * mode: the transport must be installed before finalize and every rank must be inside the call at the same time.  With no transport
 * installed at finalize (or on = 0, or contiguous shards) every rank factors the whole matrix as before.  Before finalize only.  */
int admm_hip_set_factor_local(admm_hip_ctx *ctx, int on);

/* per-phase device timing of the last admm_hip_step (HIP events on the
 * context's stream; enabled with admm_hip_enable_timing).  ms per frame.      */
typedef struct admm_hip_timing {
    float prologue_ms, local_ms, rhs_ms, allreduce_ms, solve_fwd_ms, solve_bwd_ms, epilogue_ms, total_ms;
    int32_t iters;
} admm_hip_timing;
/* on = 1: events around the phases of every ADMM iteration (eager launches); on = k > 1: around every k-th iteration only
 * -- a HIP event is a barrier packet that costs ~5 us of launch overlap -- the other iterations run event-free (one graph
 * replay each where a graph exists) and the phase sums are scaled to the frame; total_ms is always the real span.      */
int admm_hip_enable_timing(admm_hip_ctx *ctx, int on);
int admm_hip_get_timing(admm_hip_ctx *ctx, admm_hip_timing *t);
/* the step BEFORE the last one (each timed step keeps its events until the step after the next is recorded): read it after the next
 * step has been queued and the GPU never waits for the host between frames.  ADMM_ERR_STATE if it was not timed / already read. */
int admm_hip_get_timing_previous(admm_hip_ctx *ctx, admm_hip_timing *t);

/* ---- residuals and convergence-based early exit ------------------------------------------
 * The reference only DESCRIBES these (comment at System.cpp:64-65, paper Eq. 22-23):
 *     r = W (Dx - z)            primal residual, with the Dx the local step used
 *     s = D^T W^T W (z - z_prev) dual residual
 * With tracking on, every ADMM iteration of admm_hip_step also computes |r|_2 and |s|_2 (extra
 * work inside the tet / anchor kernels plus one more gather: about +9 % per iteration at 1M tets; off by default = the reference's loop).
 * admm_hip_get_residuals copies the norms of the last step; *n_iters = ADMM iterations that step ran.
 * admm_hip_set_tolerance(eps_r, eps_s, check_every): with eps_r > 0 the ADMM loop of a step ends as
 * soon as |r| <= eps_r and |s| <= eps_s, tested every `check_every` iterations (each test is one
 * host-device round trip); admm_iters stays the upper bound.  Tracking is switched on implicitly. */
int admm_hip_enable_residuals(admm_hip_ctx *ctx, int on);
int admm_hip_get_residuals(admm_hip_ctx *ctx, double *r_norm, double *s_norm, int capacity, int *n_iters);
int admm_hip_set_tolerance(admm_hip_ctx *ctx, double eps_r, double eps_s, int check_every);

#ifdef __cplusplus
}
#endif
#endif
