// ctx.hpp -- the library's context (admm_hip_ctx) and the host-side records it is made of, shared by the library's translation units:
//   admm_hip.hip     device code (kernels_*.hpp, factor_dev.hpp) + device factorization, upload, launches, step loop, C ABI
//   host_setup.cpp   Force::initialize / get_selector data, assembly of A_s, ordering + symbolic / host numeric factorization
//   partition.cpp    subtree sharding: owners of supernodes and elements, XCD-aware item order
//   comm.cpp         RCCL (bound at run time) and the all-reduce hooks + their C ABI entry points
// See include/admm_hip.h for the contract and DESIGN.md for the design.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/admm_hip.h"
#include "factor.hpp"
#include "dev_types.hpp"

namespace admm_lib {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Batch {
    int kind = 0, n_total = 0, n_local = 0;
    std::vector<int32_t> local;    // this rank's elements (ascending reference order); contiguous range unless subtree sharding
    std::vector<int32_t> idx;      // [n_total][nodes] original ids
    std::vector<double> params;    // [n_total][P]
    std::vector<double> targets;   // anchors [n_total][3]
    bool moving = false;
    std::vector<int32_t> active;   // anchors [n_total]
    double *h_tg = nullptr; int32_t *h_ac = nullptr; hipEvent_t upd_ev = nullptr;   // anchors: pinned staging of this rank's targets / flags + "last update has left it"
    // finalize
    std::vector<double> weight, rest, measure;  // [n_total], [n_total][12], [n_total]
    std::vector<int32_t> global_idx;             // compact first row
    std::vector<int32_t> corner_perm;            // [n_total][nodes]: stored corner c holds original corner corner_perm[c]
    int64_t slot_base = 0;                       // first local force slot
    int max_iter = 0;                            // largest L-BFGS max_iterations (hyperelastic kinds)
    // device
    int *d_idx = nullptr, *d_dst = nullptr, *d_active = nullptr, *d_niters = nullptr;
    int *d_order = nullptr; unsigned int *d_cost = nullptr; int n_blocks_ordered = 0;      // tets: launch order by last frame's cost (see project_tet_kernel)
    double *d_rest = nullptr, *d_par = nullptr, *d_w2h2 = nullptr, *d_kblend = nullptr, *d_w2 = nullptr;
    double *d_u = nullptr, *d_z = nullptr, *d_state = nullptr, *d_targets = nullptr;
    double *d_dx_override = nullptr, *d_dx_buf = nullptr; // parity tests only
    double *d_u_prev = nullptr, *d_z_prev = nullptr, *d_G = nullptr;   // residual tracking only
    // tets: the corners' right-hand-side shares are summed per node inside every 64-tet block (LDS) before they go to the slots:
    // one slot per (block, node) instead of one per corner (project_tet_kernel's epilogue)
    bool prered = false;
    int tpb = 64;      // tets per one-wave block (admm_hip_ctx::tet_tpb): 64, or fewer in under-filled launches -- the lanes beyond stay idle
    unsigned int *d_pos4 = nullptr; int *d_bn_ptr = nullptr, *d_bn_dst = nullptr; unsigned short *d_bn_end = nullptr;
    double *d_res_partial = nullptr; bool res_fused = false;           // tets: residuals come out of the projection kernel itself (one |r|^2 partial per 64-tet block)
    std::vector<double> G;                // [12][n_local] selector block per element, corners in device order
    // ---- ADMM_KIND_GENERIC (user-defined forces): selector rows as CSR over the batch's rows
    std::vector<int64_t> g_elem_row;      // [n_total + 1] first batch row of every element
    std::vector<int64_t> g_rowptr;        // [rows + 1]
    std::vector<int32_t> g_col;           // entry columns (3 * node + component, original node ids), ascending inside a row
    std::vector<double> g_val, g_roww;    // entry values; weight per row
    std::vector<int64_t> g_elem_node;     // [n_total + 1]
    std::vector<int32_t> g_nodes;         // every element's nodes, ascending
    int64_t g_row0 = 0, g_rows = 0;       // position in the context-wide generic row space
    int g_lrows = 0, g_lslots = 0;        // this rank's rows / (element, node) slots
    std::vector<double> g_sval; std::vector<int32_t> g_srow_b;   // per slot entry: D value and batch row (coefficients are rebuilt on recompute_weights)
    int *d_g_lrow = nullptr, *d_g_rptr = nullptr, *d_g_col = nullptr, *d_g_sptr = nullptr, *d_g_srow = nullptr, *d_g_sdst = nullptr;
    double *d_g_val = nullptr, *d_g_scoef = nullptr, *d_g_scoef_res = nullptr;   // (_res: val * w^2, the dual residual's coefficients)
    int elem_nodes(int e, const int32_t **p) const {
        if (kind == ADMM_KIND_GENERIC) { *p = g_nodes.data() + g_elem_node[e]; return (int)(g_elem_node[e + 1] - g_elem_node[e]); }
        *p = idx.data() + (size_t)e * ADMM_KIND_NODES[kind]; return ADMM_KIND_NODES[kind];
    }
    int elem_rows(int e) const { return kind == ADMM_KIND_GENERIC ? (int)(g_elem_row[e + 1] - g_elem_row[e]) : ADMM_KIND_ROWS[kind]; }
};

struct Explicit {
    int type = 0; double dir[3] = {0, 0, 0};
    std::vector<int32_t> idx;            // CONST: node ids (empty = all); WIND: [n][3] triangle node ids
    int n = 0;                            // nodes / triangles
    int *d_idx = nullptr;                 // WIND: triangles sorted by dependency level (see wind_serial_kernel)
    int *d_level_ptr = nullptr; int n_levels = 0;
};

#ifndef ADMM_BWD_BIG_CW
#define ADMM_BWD_BIG_CW 1            // backward kernel: columns per wave on levels with supernodes wider than 64
#endif
struct LevelDev {
    int n_small = 0; admm_dev::SweepItem *d_small = nullptr;   // forward: wave items (levels below the split)
    int n_big = 0, big_nw = 16; admm_dev::SweepItem *d_big = nullptr;   // forward: block items; waves per tile (4 / 8 / 16 by the level's widest supernode)
    struct Root { int k, first; int64_t foff, inv_off; };
    std::vector<Root> roots;                                   // roots solved with their explicit inverse: gather + one row-wise product (no backward items)
    int n_bwd = 0, bwd_cw = 1, bwd_nw = 4; admm_dev::SweepItem *d_bwd = nullptr;   // backward: columns per wave, waves per block
    int level = 0; double mbytes = 0.0;                        // diagnostics: position in the tree, panel bytes of this level's supernodes
};

} // namespace admm_lib

using admm_lib::Batch; using admm_lib::Explicit; using admm_lib::LevelDev;
using admm_host::SymCSC; using admm_host::Factor;

struct admm_hip_ctx {
    int device_id = -1;
    bool own_stream = false;
    hipStream_t stream = nullptr;
    std::string err;
    double dt = 0.04;
    int rank = 0, world = 1;
    admm_hip_allreduce_fn allreduce = nullptr; void *allreduce_user = nullptr;
    void *rccl_comm = nullptr; bool rccl_owned = false;      // ncclComm_t: the all-reduce is ncclAllReduce on the context's stream (takes precedence over the hook)
    admm_hip_host_allreduce_fn host_allreduce = nullptr; void *host_allreduce_user = nullptr;   // transport that sums HOST buffers (admm_hip_set_host_allreduce)
    double *h_comm = nullptr; size_t h_comm_cap = 0;          // its pinned staging
    double *d_small = nullptr; size_t d_small_cap = 0;        // admm_hip_allreduce_host's device scratch
    bool finalized = false;
    int leaf_size = 0;                        // nested-dissection leaf size; 0 = by system size (host_factor)
    // host state
    int n_nodes = 0;
    std::vector<double> x, v, m3;
    std::vector<Batch> batches;
    admm_dev::Gravity grav{};             // fast path: only constant all-node forces
    std::vector<Explicit> explicits; bool explicit_simple = true;
    admm_dev::ShapeTable shapes{}; admm_dev::ShapeTable *d_shapes = nullptr;
    SymCSC A;
    Factor F;
    admm_hip_info info{};
    // device state (node arrays in factor order)
    double *d_x = nullptr, *d_v = nullptr, *d_m3 = nullptr, *d_mxbar = nullptr, *d_xcur = nullptr, *d_y = nullptr, *d_w = nullptr, *d_c = nullptr;
    double *d_fslot = nullptr; int64_t n_fslots = 0;
    int *d_perm = nullptr; double *d_stage = nullptr;          // frame boundary: factor position -> caller's node; [2][3n] staging in the caller's order
    int slot_stride = 0;                      // > 0: RHS slots rank-major (slot of a node's r-th incidence = r * slot_stride + node), 0: node-sorted
    int64_t *d_inc_ptr = nullptr;
    double *d_panels = nullptr; int *d_sn_first = nullptr, *d_sn_ncols = nullptr, *d_sn_nrows = nullptr, *d_rows = nullptr, *d_cg_slot = nullptr, *d_cg4 = nullptr;
    int64_t *d_sn_panel_off = nullptr, *d_sn_rows_off = nullptr, *d_sn_slot_off = nullptr, *d_sn_front_off = nullptr, *d_cg_ptr = nullptr;
    std::vector<LevelDev> levels;
    std::vector<void *> allocs;
    // subtree sharding (world > 1, ADMM_HIP_SHARD=subtree / admm_hip_set_shard_mode): every rank owns whole subtrees of the
    // elimination tree and the elements that touch them; only the top of the tree is replicated
    int shard_mode = 0;                       // 0: contiguous element ranges + replicated solve, 1: subtrees
    std::vector<int> sn_owner, node_owner;    // -1 = top (replicated); node_owner in factor order
    std::vector<LevelDev> levels_top;         // sweep items of the top supernodes (levels = this rank's own ones)
    // rank-local factorization (admm_hip_set_factor_local): under subtree sharding a rank assembles, factors and keeps only its own subtrees and
    // the replicated top; the subtree roots' update matrices meet in ONE all-reduce per factorization (dev_factorize.inc).  The device's panel
    // layout is then a compact one over those supernodes (dev_panel_off; -1 = not resident) -- Factor::panel_off stays the host layout.
    bool factor_local = true, factor_local_on = false;      // wanted / in effect (set by plan_device_panels at finalize)
    std::vector<int64_t> dev_panel_off, dev_root_inv_off; int64_t dev_panels_size = 0;
    // distributed top (host_setup.cpp host_factor): the top of the tree is ONE root supernode, its product with the explicit inverse split by rows across the ranks
    // ADMM_HIP_DIST_TOP = 0: the replicated top of rounds 2-5, 1: distributed whatever the size; default (-1): distributed from dist_top_min_nodes nodes on -- its
    // second collective per iteration costs more than it saves on small systems (1M-tet bar, 178.6k nodes, 8 ranks: the slowest rank's kernels 0.264 -> 0.253 ms for one
    // more all-reduce; 4M tets, 693k nodes: 1.113 -> 0.689 ms)
    int dist_top_wanted = -1, dist_top_min_nodes = 300000; bool dist_top = false;
    int root_sn = -1, root_k = 0, root_first = 0, root_r0 = 0, root_r1 = 0; int64_t root_foff = 0;      // the root, this rank's rows [r0, r1) of its inverse (at dev_root_inv_off[root_sn])
    bool tet_order = true; int tet_order_min_blocks = 3072;      // NH / StVK batches of more blocks than that start their costliest blocks first (ADMM_HIP_TET_ORDER=0: mesh order)
    int64_t frames = 0;
    int merge_small = 0;                          // dissection regions of at most that many nodes become four-way tree nodes (ADMM_HIP_MERGE_SMALL)
    bool fuse_anchor_tail = true;                 // an anchor batch right behind a tet batch goes out in the tet launch (ADMM_HIP_FUSE_ANCHORS=0: own launch)
    bool device_factor = true, device_numeric = false;      // numeric factorization on the GPU (ADMM_HIP_FACTOR=host: on the host); what this context does
    // the tet kernels' z is an output nobody reads back in a plain frame (admm_hip_read_local aside): admm_hip_keep_z(ctx, 0) -- what
    // the class mirror and the bench do -- stops storing it in admm_hip_step; the parity entry points (local_step_only / local_step_dx)
    // and residual tracking always store it.  ADMM_HIP_KEEP_Z=0 / 1 overrides.
    bool keep_z = true, keep_z_user = true;
    bool state_zero_copy = true;              // upload_state / download_state address the caller's page-locked vectors from ONE kernel each (any size; ADMM_HIP_STATE_ZEROCOPY=0: a DMA per vector + reordering kernels)
    int tet_tpb = 0;                          // ADMM_HIP_TPB: tets per one-wave block (4 / 8 / 16 / 32 / 64) for the NH / StVK batches; 0 = 64
    bool tet_prered = true;                   // ADMM_HIP_PRERED=0: one RHS slot per tet corner (the round-1/2 layout)
    int n_comm_top = 0, n_comm_slots = 0;
    int *d_comm_top = nullptr, *d_comm_slots = nullptr; unsigned char *d_comm_mine = nullptr, *d_base_mask = nullptr, *d_keep_mask = nullptr;
    double *d_comm_buf = nullptr;
    // small systems: explicit inverse of the scalar system (factor order), one kernel per solve
    bool root_inverse = true;                 // roots of the elimination tree: forward + backward as one product with (L L^T)^-1 (ADMM_HIP_ROOT_INVERSE=0: two sweeps)
    int dense_max = 2048; bool dense = false; std::vector<double> Ainv; double *d_ainv = nullptr;
    // one ADMM iteration (local kernels, RHS, all sweep launches) captured as a HIP graph: one launch per iteration
    // instead of 30-40; matters for the small shipped scenes, which are launch-bound.  Not used around timed iterations, with
    // residual tracking or user forces, or under sharding with a host hook (ncclAllReduce inside the library is capturable:
    // ADMM_HIP_GRAPH_COMM=1).  ADMM_HIP_GRAPH=0 disables.
    // Default (graph_forced = false): only for systems of < 100k nodes, where an iteration is ~20 short dependent kernels and the
    // host's launch work matters; at 1M tets the GPU is the limit and a replay is 0.5-2 % SLOWER than the same launches issued
    // eagerly (0.780 vs 0.766-0.775 ms per iteration, tools/graph_vs_eager.py).  ADMM_HIP_GRAPH=1 forces it, 0 disables it.
    bool graph_enabled = true, graph_forced = false; hipGraph_t iter_graph = nullptr; hipGraphExec_t iter_exec = nullptr;
    // the whole ADMM loop of a frame as ONE graph (one launch per frame instead of one per iteration: the ~5-9 us between two graph
    // launches are 5-15 % of an iteration on small and mid-size scenes); captured for the iteration count of the call, again when it changes
    bool frame_graph_on = true; hipGraph_t frame_graph = nullptr; hipGraphExec_t frame_exec = nullptr; int frame_iters = 0, last_step_iters = -1;
    bool local_multi = true;                      // the whole local step in ONE launch when the scene has several batches (project_multi_kernel; ADMM_HIP_LOCAL_MULTI=0: one launch per batch)
    // class API frame boundary of small systems: no DMA, the permutation kernels read / write this page-locked buffer ([x | v], caller's order)
    int state_direct_max_nodes = 12288; double *h_state = nullptr, *h_state_dev = nullptr; size_t h_state_cap = 0; int *d_iperm = nullptr; hipEvent_t state_in_ev = nullptr; bool state_in_pending = false;
    std::vector<std::pair<const char *, size_t> > pinned;      // host spans page-locked through admm_hip_pin_host (the zero-copy state kernels check them)
    std::vector<double> bounce;                                 // upload_state / download_state of a vector that is page-locked only in part (neither mappable nor DMA-able as one span)
    bool tree_search = true;                       // pick the elimination tree of mid-size systems by the sweeps' cost model (ADMM_HIP_TREE_SEARCH=0: the rule-based tree)
    bool top_bwd_needed_only = true;              // subtree sharding: the backward sweep over the replicated top skips the separators this rank never reads (ADMM_HIP_TOP_BWD_ALL=1: all of them)
    int root_fuse_k = 2048;                       // roots of at most that many columns: t is gathered inside the product kernel (ADMM_HIP_ROOT_FUSE_K; 0 = never)
    int fwd_small_k = 64, bwd_small_k = 64;       // levels whose widest supernode has at most this many columns: wave-per-tile forward kernel / 4 columns per wave backward
    // backward sweep, wide levels with more columns than the chip holds waves (8 x 4 x 256) but at most twice as many: two columns per wave
    // instead of a second round of workgroups for the few that did not fit (1M-tet bar: levels 3, 4, 6, 7 with 8.6-13.7 k columns:
    // backward 0.212 -> 0.203 ms; including the top levels with 4.5-5.5 k columns: 0.229).  ADMM_HIP_BWD_CW2_MIN / _MAX, MIN 0 = off
    int bwd_cw2_min_cols = 8192, bwd_cw2_max_cols = 16384;
    int bwd_nw = 8, bwd_small_nw = 4;                               // backward sweep, levels of wide supernodes: waves (= columns) per block sharing one staging (ADMM_HIP_BWD_NW = 4 / 8 / 16)
    int xcd_min_supernodes = 16;                  // levels with at least this many supernodes get the XCD-aware item order (0 = off; ADMM_HIP_XCD)
    int bwd_nw_min_cols = 4096, fwd_nw16_max_tiles = 512;
    int fwd_nw4_kmax = 200, fwd_nw8_kmax = 400;   // forward sweep: levels whose widest supernode has at most this many columns run 4 / 8 waves per tile (ADMM_HIP_FWD_NW4 / _NW8)
    int64_t graph_launches = 0;               // hipGraphLaunch calls so far (admm_hip_debug_graph_state)
    bool graphs_stale = false;                // a captured iteration holds the communicator it was captured with: set when that changes (comm.cpp), honoured by the next admm_hip_step
    bool graph_comm = false;                  // ADMM_HIP_GRAPH_COMM=1: also capture the multi-GPU iteration (ncclAllReduce inside the graph)
    // residual tracking / early exit (off by default)
    bool res_on = false, res_ready = false;
    double tol_r = 0.0, tol_s = 0.0; int check_every = 1;
    double *d_res = nullptr; int res_cap = 0, res_n = 0;      // [2 * res_cap]: r^2, s^2 per iteration
    double *d_res_slots = nullptr, *d_res_s = nullptr, *d_res_partial = nullptr; int res_partial_n = 0;

    // user-defined forces: host round trip per ADMM iteration (see admm_hip_add_generic_batch)
    admm_hip_project_fn project_hook = nullptr; void *project_user = nullptr;
    int64_t n_gen_rows = 0;
    double *d_gen_dx = nullptr, *d_gen_q = nullptr;                 // [n_gen_rows]
    double *h_gen_dx = nullptr, *h_gen_u = nullptr, *h_gen_z = nullptr, *h_gen_q = nullptr;   // pinned
    std::vector<double> h_gen_u_prev, h_gen_z_prev; double *d_gen_q2 = nullptr, *d_gen_r2 = nullptr;   // residual tracking of the user rows
    hipEvent_t gen_ev = nullptr;
    // timing: HIP events around the phases of every timing_stride-th ADMM iteration (1 = every iteration); an event is a
    // barrier packet that costs ~5 us of launch overlap, so the other iterations run event-free (as a graph replay when one exists)
    bool timing = false; int timing_stride = 1; int ev_timed = 0; int timing_frame = 0;
    std::vector<hipEvent_t> evpool;   // recorded in order during a step, read back lazily
    size_t ev_used = 0; int ev_iters = 0; bool ev_pending = false;
    // the step BEFORE the last one keeps its events (the two sets swap at the start of every timed step): a caller that reads them
    // through admm_hip_get_timing_previous after queueing the next step never makes the GPU wait for the host (bench.py)
    std::vector<hipEvent_t> evpool_prev; size_t ev_used_prev = 0; int ev_iters_prev = 0, ev_timed_prev = 0; bool ev_pending_prev = false;
    admm_hip_timing last_timing{};
};

namespace admm_lib {

int fail(admm_hip_ctx *c, int code, const char *fmt, ...);      // (comm.cpp)

// (a failed call also leaves HIP's sticky last-error set: cleared here, so that an unrelated hipGetLastError() check later -- in this context or
//  another one of the process -- does not report this failure a second time as its own)
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); return fail(ctx, ADMM_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)

template <class T> int dalloc(admm_hip_ctx *ctx, T **p, size_t n) {
    *p = nullptr;
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    ctx->allocs.push_back(q);
    *p = (T *)q;
    return ADMM_OK;
}
template <class T> int upload(admm_hip_ctx *ctx, T **p, const std::vector<T> &h) {
    int rc = dalloc(ctx, p, h.size());
    if (rc) return rc;
    if (!h.empty()) HIPCHK(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return ADMM_OK;
}
#define TRY(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// is local element `el` the last one of its launch block?
inline bool block_end(const Batch &b, int el) { return el % b.tpb == b.tpb - 1; }
// number of launch blocks of a batch (tets: `tpb` elements per block; everything else LOCAL_BLOCK)
inline int batch_blocks(const Batch &b) { return (b.n_local + b.tpb - 1) / b.tpb; }

// ---- what one translation unit offers the others ----
int do_allreduce(admm_hip_ctx *ctx, double *buf, int64_t count);                                   // comm.cpp
void comm_release(admm_hip_ctx *ctx);
int comm_poll(admm_hip_ctx *ctx, int *nccl_result);
int host_assemble(admm_hip_ctx *ctx, bool reuse_rest);                                             // host_setup.cpp
int host_factor(admm_hip_ctx *ctx, bool reuse_symbolic);
void element_G(int kind, const double *rest, double G[4][3], int &cols);
int idx_stride(int kind);
void partition_subtrees(admm_hip_ctx *ctx);                                                        // partition.cpp
void assign_elements(admm_hip_ctx *ctx);
void top_needs(const admm_hip_ctx *ctx, std::vector<std::vector<char> > &need, std::vector<int> &provider);
void shard_accounting(admm_hip_ctx *ctx);
void plan_device_panels(admm_hip_ctx *ctx);
void subtree_owners(const admm_host::Factor &F, int parts, std::vector<int> &owner, std::vector<double> &load, int &n_top, size_t &n_sub);
void xcd_order(std::vector<admm_dev::SweepItem> &items, int group, int min_supernodes);

} // namespace admm_lib
