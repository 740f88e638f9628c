// admm_hip.hip -- context, host orchestration and C ABI of libadmm_hip.so.
// See include/admm_hip.h for the contract and DESIGN.md for the design.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

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
#include "force_init.hpp"
#include "kernels_global.hpp"
#include "factor_dev.hpp"
#include "kernels_local.hpp"

extern "C" int omp_get_max_threads(void);

using namespace admm_host;
using admm_dev::BatchDev;
using admm_dev::FactorDev;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Batch {
    int kind = 0, n_total = 0, n_local = 0;
    std::vector<int32_t> local;    // this rank's elements (ascending reference order); contiguous range unless subtree sharding
    std::vector<int32_t> idx;      // [n_total][nodes] original ids
    std::vector<double> params;    // [n_total][P]
    std::vector<double> targets;   // anchors [n_total][3]
    bool moving = false;
    std::vector<int32_t> active;   // anchors [n_total]
    std::vector<int> grp_ptr, grp_blk;   // pipeline groups: element range / first 64-element block of every group in the (group-major) local order, [G + 1]
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

} // namespace

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
    // Concurrent subtree groups on ONE GPU (ADMM_HIP_GROUPS, not with subtree sharding): the elimination tree below a small top is
    // cut into `groups` sets of independent subtrees; group 0 runs on the context's stream (levels), the others on side streams
    // (levels_side), the top afterwards / before (levels_gtop).  One group's kernel fills the other's launch gaps and tails.
    bool tet_order = true; int tet_order_min_blocks = 3072;      // NH / StVK batches of more blocks than that start their costliest blocks first (ADMM_HIP_TET_ORDER=0: mesh order)
    int64_t frames = 0;
    int merge_small = 0;                          // dissection regions of at most that many nodes become four-way tree nodes (ADMM_HIP_MERGE_SMALL)
    bool fuse_anchor_tail = true;                 // an anchor batch right behind a tet batch goes out in the tet launch (ADMM_HIP_FUSE_ANCHORS=0: own launch)
    bool device_factor = true, device_numeric = false;      // numeric factorization on the GPU (ADMM_HIP_FACTOR=host: on the host); what this context does
    int groups = 1;
    // Pipelined groups on ONE GPU (ADMM_HIP_PIPE=G, world 1): the elements and the elimination subtrees below a small top are cut
    // into G independent groups (same partition as `groups`); group g's chain  bwd_g(k-1) -> local step_g(k) -> rhs_g(k) -> fwd_g(k)
    // runs on its own stream and only the top of the tree joins them, so one group's latency-bound sweeps run under another
    // group's VALU-bound local step.  The element arrays of every batch are group-major (Batch::grp_ptr); without the pipeline
    // (timed iterations, residual tracking) the same layout is launched group after group on one stream: bitwise the same result.
    // the tet kernels' z is an output nobody reads back in a plain frame (admm_hip_read_local aside): admm_hip_keep_z(ctx, 0) -- what
    // the class mirror and the bench do -- stops storing it in admm_hip_step; the parity entry points (local_step_only / local_step_dx)
    // and residual tracking always store it.  ADMM_HIP_KEEP_Z=0 / 1 overrides.
    bool keep_z = true, keep_z_user = true;
    bool state_zero_copy = true;              // upload_state / download_state address the caller's page-locked vectors from ONE kernel each (any size; ADMM_HIP_STATE_ZEROCOPY=0: a DMA per vector + reordering kernels)
    int tet_lds_pad = 0;                      // ADMM_HIP_TET_LDS_PAD (probes only): unused dynamic LDS per tet block, caps the waves per SIMD (160 KB per CU)
    int tet_tpb = 0;                          // ADMM_HIP_TPB: tets per one-wave block (4 / 8 / 16 / 32 / 64) for the NH / StVK batches; 0 = 64
    bool tet_prered = true;                   // ADMM_HIP_PRERED=0: one RHS slot per tet corner (the round-1/2 layout)
    int pipe = 0; bool pipe_chain = true, pipe_graph = true; int pipe_cu_mask = 0;
    std::vector<int> pipe_node_group;                                 // per node (factor order): group, -1 = top
    std::vector<std::vector<std::pair<int, int> > > pipe_nodes;      // [G + 1] node ranges (factor order) of every group's subtrees; last = the top
    std::vector<hipStream_t> pipe_local_streams;                      // optional CU-masked streams for the groups' local step (pipe_cu_mask)
    std::vector<hipEvent_t> pipe_ev_fwd, pipe_ev_tet, pipe_ev_sw; hipEvent_t pipe_ev_top = nullptr;
    hipGraphExec_t pipe_exec[3] = {nullptr, nullptr, nullptr}; hipGraph_t pipe_graph_h[3] = {nullptr, nullptr, nullptr};   // first / middle iteration, closing backward sweeps
    std::vector<int> grp_owner;               // per supernode: group, -1 = top
    std::vector<std::vector<LevelDev> > levels_side;
    std::vector<LevelDev> levels_gtop;
    std::vector<hipStream_t> side_streams; hipEvent_t ev_fork = nullptr; std::vector<hipEvent_t> ev_join;
    int n_comm_top = 0, n_comm_slots = 0;
    int *d_comm_top = nullptr, *d_comm_slots = nullptr; unsigned char *d_comm_mine = nullptr, *d_base_mask = nullptr, *d_keep_mask = nullptr;
    double *d_comm_buf = nullptr;
    // small systems: explicit inverse of the scalar system (factor order), one kernel per solve
    bool root_inverse = true;                 // roots of the elimination tree: forward + backward as one product with (L L^T)^-1 (ADMM_HIP_ROOT_INVERSE=0: two sweeps)
    int dense_max = 2048; bool dense = false; std::vector<double> Ainv; double *d_ainv = nullptr;
    // one ADMM iteration (local kernels, RHS, all sweep launches) captured as a HIP graph: one launch per iteration
    // instead of 30-40; matters for the small shipped scenes, which are launch-bound.  Not used with timing events,
    // residual tracking or sharding (the all-reduce hook runs host code inside the loop).  ADMM_HIP_GRAPH=0 disables.
    // Default (graph_forced = false): only for systems of < 100k nodes, where an iteration is ~20 short dependent kernels and the
    // host's launch work matters; at 1M tets the GPU is the limit and a replay is 0.5-2 % SLOWER than the same launches issued
    // eagerly (0.780 vs 0.766-0.775 ms per iteration, tools/graph_vs_eager.py).  ADMM_HIP_GRAPH=1 forces it, 0 disables it.
    bool graph_enabled = true, graph_forced = false; hipGraph_t iter_graph = nullptr; hipGraphExec_t iter_exec = nullptr;
    // the whole ADMM loop of a frame as ONE graph (one launch per frame instead of one per iteration: the ~5-9 us between two graph
    // launches are 5-15 % of an iteration on small and mid-size scenes); captured for the iteration count of the call, again when it changes
    bool frame_graph_on = true; hipGraph_t frame_graph = nullptr; hipGraphExec_t frame_exec = nullptr; int frame_iters = 0, last_step_iters = -1;
    // local step of scenes with several large batches (tets of two materials, cloth triangles, hinges ...): the batches are independent
    // (own elements, own slots), so every large one can get its own stream and the launches' tails overlap (ADMM_HIP_LOCAL_STREAMS=4; measured:
    // the cross-stream dependencies cost 10-25 us each, the single launch above does better), small batches follow on the context's stream
    bool local_multi = true;                      // the whole local step in ONE launch when the scene has several batches (project_multi_kernel; ADMM_HIP_LOCAL_MULTI=0: one launch per batch)
    int local_streams_max = 1, local_streams_min_elems = 16384; std::vector<hipStream_t> local_side; std::vector<hipEvent_t> local_join; hipEvent_t local_fork = nullptr;
    // class API frame boundary of small systems: no DMA, the permutation kernels read / write this page-locked buffer ([x | v], caller's order)
    int state_direct_max_nodes = 12288; double *h_state = nullptr, *h_state_dev = nullptr; size_t h_state_cap = 0; int *d_iperm = nullptr; hipEvent_t state_in_ev = nullptr; bool state_in_pending = false;
    bool tree_search = true;                       // pick the elimination tree of mid-size systems by the sweeps' cost model (ADMM_HIP_TREE_SEARCH=0: the rule-based tree)
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
    admm_hip_timing last_timing{};
};

namespace {

int fail(admm_hip_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    fprintf(stderr, "admm_hip: %s\n", buf);
    return code;
}

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, ADMM_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

// ---- RCCL, bound at run time (dlopen): the library has no link-time dependency on it, single-GPU users never load it ----
struct nccl_uid { char internal[128]; };
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(void **, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi *rccl_api(std::string *why) {
    static RcclApi api; static bool tried = false; static std::string err;
    if (!tried) {
        tried = true;
        // the copy already in the process first (PyTorch ships its own librccl.so): two RCCL instances must not share a job
        const char *names[] = {getenv("ADMM_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !api.handle; ++pass)
            for (const char *nm : names) { if (!nm || !*nm) continue; api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)); if (api.handle) break; }
        if (!api.handle) err = std::string("librccl.so not found (") + (dlerror() ? dlerror() : "no dlerror") + "); set ADMM_HIP_RCCL_LIB";
        else {
            api.GetUniqueId = (int (*)(nccl_uid *))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(void **, int, nccl_uid, int))dlsym(api.handle, "ncclCommInitRank");
            api.CommDestroy = (int (*)(void *))dlsym(api.handle, "ncclCommDestroy");
            api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(api.handle, "ncclAllReduce");
            api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) { err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce"; api.handle = nullptr; }
        }
    }
    if (!api.handle) { if (why) *why = err; return nullptr; }
    return &api;
}
// sum `count` doubles in place across the ranks, on the context's stream: RCCL directly when a communicator is installed
// (admm_hip_rccl_init / admm_hip_set_rccl_comm: no host code between the kernels, capturable), otherwise the caller's hook
int do_allreduce(admm_hip_ctx *ctx, double *buf, int64_t count) {
    if (ctx->rccl_comm) {
        RcclApi *R = rccl_api(nullptr);
        const int rc = R ? R->AllReduce(buf, buf, (size_t)count, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->rccl_comm, ctx->stream) : -1;
        if (rc != 0) return fail(ctx, ADMM_ERR_COMM, "ncclAllReduce failed: %s", (R && R->GetErrorString) ? R->GetErrorString(rc) : "RCCL not loaded");
        return ADMM_OK;
    }
    if (!ctx->allreduce) return fail(ctx, ADMM_ERR_COMM, "world size %d but neither an RCCL communicator nor an all-reduce hook is installed", ctx->world);
    if (ctx->allreduce(ctx->allreduce_user, buf, count, (void *)ctx->stream) != 0) return fail(ctx, ADMM_ERR_COMM, "all-reduce hook failed");
    return ADMM_OK;
}

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

// captured iterations carry device addresses and flags in their kernel arguments: whatever changes those drops the graphs
void drop_iteration_graphs(admm_hip_ctx *ctx) {
    if (ctx->iter_exec) { (void)hipGraphExecDestroy(ctx->iter_exec); ctx->iter_exec = nullptr; }
    if (ctx->iter_graph) { (void)hipGraphDestroy(ctx->iter_graph); ctx->iter_graph = nullptr; }
    if (ctx->frame_exec) { (void)hipGraphExecDestroy(ctx->frame_exec); ctx->frame_exec = nullptr; }
    if (ctx->frame_graph) { (void)hipGraphDestroy(ctx->frame_graph); ctx->frame_graph = nullptr; }
    ctx->frame_iters = 0;
}

void free_device(admm_hip_ctx *ctx) {
    drop_iteration_graphs(ctx);
    for (int q = 0; q < 3; ++q) {
        if (ctx->pipe_exec[q]) { (void)hipGraphExecDestroy(ctx->pipe_exec[q]); ctx->pipe_exec[q] = nullptr; }
        if (ctx->pipe_graph_h[q]) { (void)hipGraphDestroy(ctx->pipe_graph_h[q]); ctx->pipe_graph_h[q] = nullptr; }
    }
    for (void *p : ctx->allocs) (void)hipFree(p);
    ctx->allocs.clear();
    for (Batch &b : ctx->batches) {
        if (b.h_tg) (void)hipHostFree(b.h_tg);
        if (b.h_ac) (void)hipHostFree(b.h_ac);
        if (b.upd_ev) (void)hipEventDestroy(b.upd_ev);
        b.h_tg = nullptr; b.h_ac = nullptr; b.upd_ev = nullptr;
    }
    ctx->levels.clear(); ctx->levels_side.clear(); ctx->levels_gtop.clear();
}

// scalar "G" matrix of an element: nodes x cols, so that K_e = dt^2 w^2 G G^T
void element_G(int kind, const double *rest, double G[4][3], int &cols) {
    std::memset(G, 0, sizeof(double) * 12);
    switch (kind) {
    case ADMM_KIND_ANCHOR: case ADMM_KIND_COLLISION: cols = 1; G[0][0] = 1.0; break;
    case ADMM_KIND_SPRING: cols = 1; G[0][0] = 1.0; G[1][0] = -1.0; break;
    case ADMM_KIND_TET_LINEAR: case ADMM_KIND_TET_VOLUME: case ADMM_KIND_TET_NH: case ADMM_KIND_TET_STVK:
        cols = 3; for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) G[c][r] = rest[c + 4 * r]; break;
    case ADMM_KIND_TRI_STRAIN: case ADMM_KIND_TRI_AREA: case ADMM_KIND_TRI_FUNG: cols = 2; for (int c = 0; c < 3; ++c) for (int r = 0; r < 2; ++r) G[c][r] = rest[c + 3 * r]; break;
    case ADMM_KIND_BEND: cols = 3; G[0][0] = 1.0; G[2][0] = -1.0; G[3][1] = 1.0; G[2][1] = -1.0; G[1][2] = 1.0; G[2][2] = -1.0; break;
    default: cols = 0;
    }
}

int idx_stride(int kind) {
    switch (kind) { case ADMM_KIND_ANCHOR: case ADMM_KIND_COLLISION: return 1; case ADMM_KIND_SPRING: return 2; default: return 4; }
}

// A user-defined element's share of A_s.  The accelerated path factors the scalar system, so dt^2 D_e^T W^2 D_e must be
// K (x) I3: no coupling between coordinates and the same K for x, y and z.  Checked per element, refused loudly otherwise.
int assemble_generic(admm_hip_ctx *ctx, const Batch &b, std::vector<int> &ti, std::vector<int> &tj, std::vector<double> &tv) {
    const double dt = ctx->dt;
    struct Ent { int a, c, comp; double v; };      // (node a >= node c, coordinate) -> dt^2 sum_r D(r, a) w_r^2 D(r, c); sparse: an element may span all nodes
    std::vector<Ent> ent;
    for (int e = 0; e < b.n_total; ++e) {
        const int32_t *nodes; const int nn = b.elem_nodes(e, &nodes);
        if (nn && nodes[nn - 1] >= ctx->n_nodes) return fail(ctx, ADMM_ERR_ARG, "user-defined force %d references node %d (have %d nodes)", e, nodes[nn - 1], ctx->n_nodes);
        ent.clear();
        double cross = 0.0, kmax = 0.0;
        for (int64_t r = b.g_elem_row[e]; r < b.g_elem_row[e + 1]; ++r) {
            const double w = b.g_roww[r];
            for (int64_t p = b.g_rowptr[r]; p < b.g_rowptr[r + 1]; ++p) for (int64_t q = b.g_rowptr[r]; q < b.g_rowptr[r + 1]; ++q) {
                const int na = b.g_col[p] / 3, cp = b.g_col[p] % 3, nc = b.g_col[q] / 3, cq = b.g_col[q] % 3;
                const double t = (((dt * dt) * b.g_val[p]) * w) * w * b.g_val[q];
                if (cp != cq) { cross = std::max(cross, std::fabs(t)); continue; }
                if (na >= nc) ent.push_back({na, nc, cp, t});
            }
        }
        std::stable_sort(ent.begin(), ent.end(), [](const Ent &x, const Ent &y) { return x.a != y.a ? x.a < y.a : (x.c != y.c ? x.c < y.c : x.comp < y.comp); });
        // reduce runs of equal (a, c, comp), then compare the three coordinates of every (a, c)
        std::vector<Ent> red;
        for (const Ent &x : ent) { if (!red.empty() && red.back().a == x.a && red.back().c == x.c && red.back().comp == x.comp) red.back().v += x.v; else red.push_back(x); }
        for (const Ent &x : red) kmax = std::max(kmax, std::fabs(x.v));
        double dev = 0.0;
        for (size_t i = 0; i < red.size();) {
            size_t j = i; double k3[3] = {0.0, 0.0, 0.0};
            for (; j < red.size() && red[j].a == red[i].a && red[j].c == red[i].c; ++j) k3[red[j].comp] = red[j].v;
            dev = std::max(dev, std::max(std::fabs(k3[0] - k3[1]), std::fabs(k3[0] - k3[2])));
            ti.push_back(red[i].a); tj.push_back(red[i].c); tv.push_back(k3[0]);
            i = j;
        }
        if (cross > 1e-12 * kmax || dev > 1e-12 * kmax)
            return fail(ctx, ADMM_ERR_UNSUPPORTED, "user-defined force %d of a generic batch: D^T W^2 D is not of the form K (x) I3 (coordinate coupling %.3g, x/y/z mismatch %.3g of %.3g); "
                        "the accelerated path factors the scalar system", e, cross, dev, kmax);
    }
    return ADMM_OK;
}

// ---- host part of finalize: rest data, rows, A_s, ordering, factorization ----
int host_assemble(admm_hip_ctx *ctx, bool reuse_rest) {
    const int n = ctx->n_nodes;
    const double dt = ctx->dt;
    int64_t row = 0, ntot = 0;
    std::vector<int> ti, tj; std::vector<double> tv;
    for (int i = 0; i < n; ++i) {
        const double m = ctx->m3[3 * (size_t)i];
        if (ctx->m3[3 * (size_t)i + 1] != m || ctx->m3[3 * (size_t)i + 2] != m)
            return fail(ctx, ADMM_ERR_UNSUPPORTED, "node %d has different masses for x/y/z; the accelerated path factors the scalar system A_s (x) I3", i);
        ti.push_back(i); tj.push_back(i); tv.push_back(m);
    }
    for (Batch &b : ctx->batches) {
        if (b.kind == ADMM_KIND_GENERIC) {
            if (!reuse_rest) b.global_idx.assign(b.n_total, 0);
            for (int e = 0; e < b.n_total; ++e) { b.global_idx[e] = (int32_t)row; row += b.elem_rows(e); }
            TRY(assemble_generic(ctx, b, ti, tj, tv));
            ntot += b.n_total;
            continue;
        }
        const int nn = ADMM_KIND_NODES[b.kind], np = ADMM_KIND_PARAMS[b.kind], rows = ADMM_KIND_ROWS[b.kind];
        if (!reuse_rest) {
            b.weight.assign(b.n_total, 0.0); b.rest.assign((size_t)b.n_total * 12, 0.0); b.measure.assign(b.n_total, 0.0);
            b.global_idx.assign(b.n_total, 0);
        }
        for (int e = 0; e < b.n_total; ++e) {
            const int *id = b.idx.data() + (size_t)e * nn;
            for (int c = 0; c < nn; ++c) if (id[c] < 0 || id[c] >= n) return fail(ctx, ADMM_ERR_ARG, "batch element %d references node %d (have %d nodes)", e, id[c], n);
            if (!reuse_rest) {
                if (!force_initialize(b.kind, id, b.params.data() + (size_t)e * np, ctx->x.data(), &b.weight[e], &b.rest[(size_t)e * 12]))
                    return fail(ctx, ADMM_ERR_UNSUPPORTED, "force kind %d is not accelerated", b.kind);
                b.measure[e] = force_measure(b.kind, id, ctx->x.data());
                if (b.kind == ADMM_KIND_ANCHOR && !b.moving) for (int j = 0; j < 3; ++j) b.targets[3 * (size_t)e + j] = ctx->x[3 * (size_t)id[0] + j];
            }
            b.global_idx[e] = (int32_t)row; row += rows;
            double G[4][3]; int cols;
            element_G(b.kind, &b.rest[(size_t)e * 12], G, cols);
            const double w = b.weight[e];
            for (int a = 0; a < nn; ++a) for (int c = 0; c < nn; ++c) {
                if (id[a] < id[c]) continue; // lower triangle (i >= j); equal ids handled once per ordered pair below
                if (id[a] == id[c] && a < c) continue;
                double sacc = 0.0;
                for (int q = 0; q < cols; ++q) sacc += (((dt * dt) * G[a][q]) * w) * w * G[c][q];
                if (id[a] == id[c] && a != c) sacc *= 2.0; // both (a,c) and (c,a) land on the same diagonal entry
                ti.push_back(id[a]); tj.push_back(id[c]); tv.push_back(sacc);
            }
        }
        ntot += b.n_total;
    }
    build_symcsc(n, ti, tj, tv, ctx->A);
    ctx->info.n_nodes = n; ctx->info.n_elems_total = ntot; ctx->info.rows_compact = row;
    ctx->info.nnz_A = (int64_t)ctx->A.idx.size();
    return ADMM_OK;
}

int host_factor(admm_hip_ctx *ctx, bool reuse_symbolic) {
    // measured on the MI355X host (EPYC 9575F, 1M-tet bar): 8-16 threads 3.3 s, 32: 5.7 s, 128: 51 s --
    // the front pool and the small dense calls do not scale further, so cap the team.
    int threads = std::min(16, std::max(1, omp_get_max_threads()));
    if (const char *e = getenv("ADMM_HIP_THREADS")) if (atoi(e) > 0) threads = atoi(e);
    ctx->info.host_threads = threads;
    if (!reuse_symbolic) {
        std::vector<double> xyz(ctx->x);
        // Larger dissection leaves = fewer elimination-tree levels (each costs >= 7-10 us per sweep whatever its size) for a little
        // more fill.  Measured (tools/leaf_sweep.py, us per ADMM iteration, leaf 16 / 64 / 128 / 256): 18.8k nodes 266 / 251 / 235 / 229,
        // 44k nodes 334 / 318 / 321 / 322, 178.6k nodes 982 / 954 / 1026 / 1020.
        // Under subtree sharding what counts is a rank's share: 8 ranks of the 178.6k-node bar (22k nodes each) run 4 % faster with
        // leaves of 128 (per-rank kernel time 0.438 -> 0.421 ms, tools/fake_world.sh with ADMM_HIP_LEAF), 4 ranks are indifferent.
        const bool own_subtrees = ctx->world > 1 && ctx->shard_mode == ADMM_SHARD_SUBTREE;     // contiguous sharding replicates the whole solve: one GPU's choice
        const int64_t share = ctx->n_nodes / std::max(1, ctx->world);
        // (round 2, with this round's sweep kernels: per-rank forward + backward at 8 ranks, leaves 64 / 128 / 256 / 384 / 512:
        //  0.248 / 0.239 / 0.229 / 0.226 / 0.234 ms; at 4 ranks 64 / 128 / 256 / 384: 0.275 / 0.270 / 0.265 / 0.260; at 2 ranks 64 is best)
        // (round 3, tools/probe/tree_policy_ab.py, us per ADMM iteration, leaf 64 without four-way nodes -> leaf 256 with them: 26.9k nodes
        //  214 -> 181, 37.6k 253 -> 224, 47.5k 276 -> 254, 63.1k 311 -> 308; four-way nodes with leaves of 64: 63.1k 311 -> 301, 101.8k 443 -> 433)
        const int leaf = ctx->leaf_size > 0 ? ctx->leaf_size : (own_subtrees ? (share < 30000 ? 384 : (share < 60000 ? 256 : 64)) : (ctx->n_nodes < 55000 ? 256 : 64));
        // four-way tree nodes (a region's separator merged with its two half-separators) halve the level count again; worth 6-9 % on
        // mid-size scenes (10k / 18.8k nodes: 189 -> 173 / 228 -> 214 us per iteration), nothing at 178.6k nodes (tools/merge_sweep.py)
        // (round 3: with every region above the leaf size a four-way node -- threshold 100 instead of 1000 -- 3.7k nodes 111 -> 100, 10k 146 -> 125,
        //  37.6k 225 -> 210, 63.1k 303 -> 284, 101.8k 434 -> 419 us per iteration; at 178.6k any merging below the root costs 5 %: 673 -> 705-721)
        //  the mixed scene of BASELINE configs[4], 140.6k nodes: 570 -> 540)
        int merge_above = ctx->n_nodes < 160000 ? 100 : 0;
        if (own_subtrees) merge_above = ctx->n_nodes < 25000 ? 1000 : 0;      // (subtree sharding: not re-measured this round, the round-2 rule stands)
        if (const char *e = getenv("ADMM_HIP_MERGE")) merge_above = atoi(e);
        // large systems: only the top region merges (root = top separator + its two half-separators, solved as one dense product
        // with its explicit inverse): the two top levels of both sweeps -- ~20 us of latency each at 1M tets -- become one
        // HBM-rate product of k^2 doubles (3335^2 x 8 B = 89 MB at the 1M-tet bar)
        bool merge_root = merge_above == 0 && ctx->root_inverse;
        if (const char *e = getenv("ADMM_HIP_MERGE_ROOT")) merge_root = atoi(e) != 0;
        // Subtree sharding: a rank's own subtrees are mid-size systems (22k nodes each at 8 ranks of the 178.6k-node bar) whose levels are
        // latency-bound, the replicated top is not: regions of up to 4/3 of a rank's share become four-way nodes, the top keeps its tree.
        // Per-rank forward + backward (tools/fake_world_policy.sh, no-op all-reduce): 8 ranks 0.198 -> 0.173 ms (thresholds 20k / 30k / 40k /
        // 60k: 0.181 / 0.173 / 0.173 / 0.234), 4 ranks 0.247 -> 0.220 (30k: 0.225, 60k: 0.220), 2 ranks 0.299 -> 0.281 (60k / 120k alike).
        int merge_small = ctx->merge_small;
        if (own_subtrees && merge_small == 0 && share >= 4096) merge_small = (int)std::min<int64_t>(share * 4 / 3, 2000000000);      // (tiny shares: a merged node that moves to the top would be a large part of the system)
        if (const char *e = getenv("ADMM_HIP_MERGE_SMALL")) merge_small = atoi(e);
        // eight-way nodes (seven separators in one supernode) save one more level between 6k and 30k nodes: configs[2] (10k nodes) -4 %, 26.9k -1.7 %
        // (tools/probe/env_ab.py ADMM_HIP_MERGE_DEPTH 2 3, four alternations); 47.5k nodes +2 %, 63k and above +15 %: four-way there
        int merge_depth = (!own_subtrees && ctx->n_nodes >= 6000 && ctx->n_nodes < 30000) ? 3 : 2;
        if (const char *e = getenv("ADMM_HIP_MERGE_DEPTH")) merge_depth = atoi(e);
        int root_depth = 0;
        if (const char *e = getenv("ADMM_HIP_ROOT_DEPTH")) root_depth = atoi(e);
        analyze(ctx->A, xyz.data(), leaf, ctx->F, merge_above, merge_root, merge_small, merge_depth, root_depth);
        // Tree search (systems between the dense limit and 160k nodes on one GPU, no ordering knob set by hand): the thresholds above were
        // measured on bars; other shapes get the same trade-off from a cost model of the two sweeps fitted to 192 measured (scene, tree) pairs
        // (tools/probe/tree_model_data.py, NOTES section E): 11.9 us per level below the roots (both sweeps: launch + dependent chain), 0.48 us per MB
        // of panels (2 sweeps at ~4.2 TB/s), 0.18 us per MB of a root's explicit inverse (one product at ~5.7 TB/s); mean error 4-7 %, its pick
        // within 5 % of the best of 24 trees on every held-out scene.  Candidates: leaves of 64 / 128 / 256, four- or eight-way nodes, the root
        // spanning 4 bisection levels or not; ordering + symbolic analysis cost 2-60 ms each.  The rule-based tree stays unless the model
        // sees at least 3 % in another one.
        const bool by_hand = getenv("ADMM_HIP_LEAF") || getenv("ADMM_HIP_MERGE") || getenv("ADMM_HIP_MERGE_ROOT") || getenv("ADMM_HIP_MERGE_SMALL") || getenv("ADMM_HIP_MERGE_DEPTH") ||
                             getenv("ADMM_HIP_ROOT_DEPTH") || ctx->leaf_size > 0 || ctx->merge_small > 0;
        if (ctx->tree_search && !by_hand && !own_subtrees && ctx->world == 1 && ctx->n_nodes > ctx->dense_max && ctx->n_nodes < 160000) {
            auto model_us = [&](const Factor &T) {
                double us = 0.0;
                for (const std::vector<int> &L : T.levels) {
                    double mb = 0.0, inv_mb = 0.0; bool plain = false;
                    for (int sn : L) {
                        const Supernode &S = T.sn[sn];
                        if (S.parent < 0 && S.ncols > ROOT_INV_MIN_COLS && ctx->root_inverse) inv_mb += 8e-6 * (double)S.ncols * root_inv_ld(S.ncols);
                        else { plain = true; mb += 8e-6 * ((double)S.ncols * (S.ncols + 1) / 2 + (double)S.nrows * S.ncols); }
                    }
                    us += (plain ? 11.9 : 0.0) + 0.48 * mb + 0.176 * inv_mb;
                }
                return us;
            };
            const double base = model_us(ctx->F);
            double best = base; Factor bestF; bool found = false;
            const bool big = ctx->n_nodes >= 60000;
            int tried = 0;
            struct Cand { int leaf, merge, depth, root_depth; bool merge_root; };
            std::vector<Cand> cands;
            for (int lf : {64, 128, 256}) for (int dp : {2, 3}) for (int rd : {0, 4}) {
                if (big && (lf == 128 || dp == 3)) continue;                     // (each analysis costs 30-60 ms there; eight-way nodes never paid above 50k nodes)
                cands.push_back({lf, 100, dp, rd, false});
            }
            if (big) cands.push_back({64, 0, 2, 0, ctx->root_inverse});          // the binary tree with the merged root (what the largest systems use): irregular meshes fill in faster under four-way nodes
            for (const Cand &cd : cands) {
                if (cd.leaf == leaf && cd.depth == merge_depth && cd.root_depth == root_depth && cd.merge == merge_above && cd.merge_root == merge_root) continue;      // the rule-based tree itself
                Factor T;
                analyze(ctx->A, xyz.data(), cd.leaf, T, cd.merge, cd.merge_root, 0, cd.depth, cd.root_depth);
                ++tried;
                const double c = model_us(T);
                if (getenv("ADMM_HIP_VERBOSE")) fprintf(stderr, "admm_hip: tree search: leaf %3d, %s nodes, root depth %d: %zu levels, model %.1f us per solve\n", cd.leaf,
                                                        cd.merge ? (cd.depth == 3 ? "eight-way" : "four-way") : "binary", cd.root_depth, T.levels.size(), c);
                if (c < best) { best = c; bestF = std::move(T); found = true; }
            }
            if (getenv("ADMM_HIP_VERBOSE")) fprintf(stderr, "admm_hip: tree search: rule-based tree %.1f us, best of %d others %.1f us -> %s\n", base, tried, best, (found && best < 0.97 * base) ? "taken" : "rule-based tree kept");
            if (found && best < 0.97 * base) { const double t_o = ctx->F.t_order, t_s = ctx->F.t_symbolic; ctx->F = std::move(bestF); ctx->F.t_order += t_o; ctx->F.t_symbolic += t_s; }
        }
    }
    // numeric phase: on the device when there is one (device_factorize, from upload_factor / recompute_weights); the small-system
    // inverse and device-less contexts (CPU tests of the host factorization) factor here
    ctx->device_numeric = ctx->device_id >= 0 && ctx->device_factor && !(ctx->n_nodes > 0 && ctx->n_nodes <= ctx->dense_max);
    Factor &F = ctx->F;
    if (ctx->device_numeric) { plan_panels(F); F.panels.clear(); F.t_numeric = 0.0; }
    else {
        int err = factorize(ctx->A, ctx->F, threads);
        if (err) return fail(ctx, ADMM_ERR_FACTOR, "system matrix is not positive definite (supernode %d)", err - 1);
    }
    ctx->info.nnz_L = F.nnz_tri;
    ctx->info.panel_bytes = F.panels_size * 8;
    ctx->info.n_supernodes = (int64_t)F.sn.size();
    ctx->info.n_levels = (int64_t)F.levels.size();
    ctx->info.max_super_cols = F.max_cols; ctx->info.max_super_rows = F.max_rows;
    ctx->info.solve_contrib_rows = F.n_slots;
    if (getenv("ADMM_HIP_VERBOSE")) {
        for (size_t l = 0; l < F.levels.size(); ++l) {
            int64_t e = 0, rws = 0; int mk = 0, mf = 0, small = 0;
            for (int s : F.levels[l]) { const Supernode &S = F.sn[s]; e += (int64_t)S.ncols * (S.ncols + 1) / 2 + (int64_t)S.nrows * S.ncols; rws += S.ncols + S.nrows; mk = std::max(mk, S.ncols); mf = std::max(mf, S.ncols + S.nrows); small += S.ncols <= 64; }
            fprintf(stderr, "admm_hip: level %2zu: %6zu supernodes (%d with k<=64), max k %4d, max front %4d, front rows %8lld, entries %10lld (%.1f MB)\n", l, F.levels[l].size(), small, mk, mf, (long long)rws, (long long)e, e * 8e-6);
        }
    }
    ctx->info.t_order_s = F.t_order; ctx->info.t_symbolic_s = F.t_symbolic; ctx->info.t_numeric_s = F.t_numeric;
    // small system: form A_s^-1 in factor order with the factor itself, three unit vectors per solve
    const int n = F.n;
    ctx->dense = n > 0 && n <= ctx->dense_max;
    ctx->info.dense_solve = ctx->dense ? 1 : 0;
    ctx->Ainv.clear();
    if (ctx->dense) {
        const double t0 = now_s();
        ctx->Ainv.assign((size_t)n * n, 0.0);
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
        for (int j0 = 0; j0 < n; j0 += 3) {
            std::vector<double> b(3 * (size_t)n, 0.0), x(3 * (size_t)n);
            for (int c = 0; c < 3 && j0 + c < n; ++c) b[3 * (size_t)F.perm[j0 + c] + c] = 1.0;
            panel_solve_host(F, b.data(), x.data());
            for (int c = 0; c < 3 && j0 + c < n; ++c) for (int i = 0; i < n; ++i) ctx->Ainv[(size_t)i * n + j0 + c] = x[3 * (size_t)F.perm[i] + c];
        }
        // symmetrise (the two triangles differ by rounding): rows are what the kernel streams
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) { const double a = 0.5 * (ctx->Ainv[(size_t)i * n + j] + ctx->Ainv[(size_t)j * n + i]); ctx->Ainv[(size_t)i * n + j] = a; ctx->Ainv[(size_t)j * n + i] = a; }
        ctx->info.t_numeric_s += now_s() - t0;
    }
    return ADMM_OK;
}

// ---- numeric factorization on the device (factor_dev.hpp) -------------------------------------------------------------
// The symbolic structure (ctx->F: supernodes, rows, levels) is the host's; this fills ctx->d_panels.  Every call builds the task
// records against freshly allocated fronts (all fronts resident at once: sum f^2 doubles, 3.2 GB at the 1M-tet bar), runs the
// launches level by level on the context's stream and frees the fronts again.  Returns ADMM_ERR_NOMEM_DEVICE_FACTOR (> 0, internal)
// when the fronts do not fit: the caller then factors on the host as before.
constexpr int ADMM_DEVFACTOR_NOFIT = 1000;
int device_factorize(admm_hip_ctx *ctx) {
    using namespace admm_dev;
    const double t0 = now_s();
    Factor &F = ctx->F;
    const int ns = (int)F.sn.size();
    std::vector<int64_t> foff(ns);
    int64_t ftot = 0;
    for (int s = 0; s < ns; ++s) { const int64_t f = F.sn[s].ncols + F.sn[s].nrows; foff[s] = ftot; ftot += f * f; }
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    // the fronts plus this call's side buffers (A's values and scatter maps: ~20 bytes per entry; row maps; the task records)
    const double side_b = 20.0 * (double)ctx->A.val.size() + 4.0 * (double)F.rows.size() + 64.0 * 1024.0 * 1024.0;
    if ((double)ftot * 8.0 + side_b > 0.8 * (double)free_b) return ADMM_DEVFACTOR_NOFIT;
    // original entries: destination in the fronts, source in A
    SymCSC PA;
    permuted_lower(ctx->A, F, PA, true);
    const int64_t nnz = (int64_t)PA.idx.size();
    std::vector<int64_t> adst(nnz); std::vector<int> asrc(nnz);
    for (int s = 0; s < ns; ++s) {
        const Supernode &S = F.sn[s];
        const int k = S.ncols, f = k + S.nrows;
        const int *rows = F.rows.data() + S.rows_off;
        for (int j = 0; j < k; ++j) {
            const int col = S.first + j;
            for (int64_t p = PA.ptr[col]; p < PA.ptr[col + 1]; ++p) {
                const int row = PA.idx[p];
                const int loc = row < S.first + k ? row - S.first : k + (int)(std::lower_bound(rows, rows + S.nrows, row) - rows);
                adst[p] = foff[s] + loc + (int64_t)f * j; asrc[p] = (int)PA.val[p];
            }
        }
    }
    // a child's update rows in its parent's front
    std::vector<int> rel(std::max<size_t>(F.rows.size(), 1), 0);
    std::vector<std::vector<int> > kids(ns);
    for (int s = 0; s < ns; ++s) {
        const Supernode &S = F.sn[s];
        if (S.parent < 0) continue;
        kids[S.parent].push_back(s);
        const Supernode &Pn = F.sn[S.parent];
        const int *prow = F.rows.data() + Pn.rows_off, *rows = F.rows.data() + S.rows_off;
        for (int a = 0; a < S.nrows; ++a) {
            const int row = rows[a];
            rel[S.rows_off + a] = row < Pn.first + Pn.ncols ? row - Pn.first : Pn.ncols + (int)(std::lower_bound(prow, prow + Pn.nrows, row) - prow);
        }
    }
    // device buffers of this call
    double *d_fronts = nullptr, *d_aval = nullptr; int64_t *d_adst = nullptr; int *d_asrc = nullptr, *d_rel = nullptr, *d_fail = nullptr;
    GemmTask *d_gemm = nullptr; PotrfTask *d_potrf = nullptr; ExtendTask *d_ext = nullptr;
    auto cleanup = [&]() { for (void *p : {(void *)d_fronts, (void *)d_aval, (void *)d_adst, (void *)d_asrc, (void *)d_rel, (void *)d_fail, (void *)d_gemm, (void *)d_potrf, (void *)d_ext}) if (p) (void)hipFree(p); };
    if (hipMalloc(&d_fronts, sizeof(double) * std::max<int64_t>(ftot, 1)) != hipSuccess) { (void)hipGetLastError(); return ADMM_DEVFACTOR_NOFIT; }
    // an allocation that fails is "does not fit" (clean up, clear HIP's error, let the caller factor on the host); anything else is an error
#define DF_CHK(call) do { hipError_t e_ = (call); if (e_ == hipErrorOutOfMemory) { cleanup(); (void)hipGetLastError(); return ADMM_DEVFACTOR_NOFIT; } \
                          if (e_ != hipSuccess) { cleanup(); return fail(ctx, ADMM_ERR_HIP, "device factorization: %s: %s", #call, hipGetErrorString(e_)); } } while (0)
    hipStream_t st = ctx->stream;
    DF_CHK(hipMemsetAsync(d_fronts, 0, sizeof(double) * std::max<int64_t>(ftot, 1), st));
    DF_CHK(hipMemsetAsync(ctx->d_panels, 0, sizeof(double) * std::max<int64_t>(F.panels_size, 1), st));
    DF_CHK(hipMalloc(&d_aval, sizeof(double) * std::max<size_t>(ctx->A.val.size(), 1)));
    DF_CHK(hipMalloc(&d_adst, sizeof(int64_t) * std::max<int64_t>(nnz, 1)));
    DF_CHK(hipMalloc(&d_asrc, sizeof(int) * std::max<int64_t>(nnz, 1)));
    DF_CHK(hipMalloc(&d_rel, sizeof(int) * rel.size()));
    DF_CHK(hipMalloc(&d_fail, sizeof(int)));
    DF_CHK(hipMemcpyAsync(d_aval, ctx->A.val.data(), sizeof(double) * ctx->A.val.size(), hipMemcpyHostToDevice, st));
    DF_CHK(hipMemcpyAsync(d_adst, adst.data(), sizeof(int64_t) * nnz, hipMemcpyHostToDevice, st));
    DF_CHK(hipMemcpyAsync(d_asrc, asrc.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, st));
    DF_CHK(hipMemcpyAsync(d_rel, rel.data(), sizeof(int) * rel.size(), hipMemcpyHostToDevice, st));
    DF_CHK(hipMemsetAsync(d_fail, 0, sizeof(int), st));
    // task records and the launch sequence
    std::vector<GemmTask> gemm; std::vector<PotrfTask> potrf; std::vector<ExtendTask> ext;
    struct Launch { int kind, first, count, gx; };      // kind 0 extend-add, 1 potrf + block inverse, 2 gemm
    std::vector<Launch> launches;
    double *Pn = ctx->d_panels;
    auto front = [&](int s) { return d_fronts + foff[s]; };
    auto close_gemm = [&](size_t first) {
        if (gemm.size() == first) return;
        int gx = 1;
        for (size_t q = first; q < gemm.size(); ++q) gx = std::max(gx, ((gemm[q].m + 63) / 64) * ((gemm[q].n + 63) / 64));
        launches.push_back({2, (int)first, (int)(gemm.size() - first), gx});
    };
    auto add_gemm = [&](const double *A, int lda, const double *B, int ldb, double *C, int ldc, int m, int n, int k, int flags, double alpha, double beta) {
        if (m <= 0 || n <= 0 || k <= 0) return;
        GemmTask T{}; T.A = A; T.B = B; T.C = C; T.m = m; T.n = n; T.k = k; T.lda = lda; T.ldb = ldb; T.ldc = ldc; T.flags = flags; T.alpha = alpha; T.beta = beta;
        gemm.push_back(T);
    };
    for (size_t l = 0; l < F.levels.size(); ++l) {
        const std::vector<int> &lev = F.levels[l];
        // extend-add, one launch per child rank (two children of one front never in the same launch: fixed summation order)
        size_t max_kids = 0; for (int s : lev) max_kids = std::max(max_kids, kids[s].size());
        for (size_t q = 0; q < max_kids; ++q) {
            const size_t first = ext.size(); int gx = 1;
            for (int s : lev) {
                if (kids[s].size() <= q) continue;
                const int c = kids[s][q]; const Supernode &Cn = F.sn[c];
                if (Cn.nrows == 0) continue;
                const int fc = Cn.ncols + Cn.nrows;
                ExtendTask T{}; T.U = front(c) + Cn.ncols + (size_t)fc * Cn.ncols; T.P = front(s); T.rel = d_rel + Cn.rows_off; T.rc = Cn.nrows; T.fc = fc; T.fp = F.sn[s].ncols + F.sn[s].nrows;
                ext.push_back(T); gx = std::max(gx, (Cn.nrows + 7) / 8);
            }
            if (ext.size() > first) launches.push_back({0, (int)first, (int)(ext.size() - first), gx});
        }
        int kmax = 0; for (int s : lev) kmax = std::max(kmax, F.sn[s].ncols);
        for (int jb = 0; jb < kmax; jb += 64) {
            const size_t pfirst = potrf.size();
            for (int s : lev) {
                const Supernode &S = F.sn[s]; const int k = S.ncols, f = k + S.nrows;
                if (k <= jb) continue;
                PotrfTask T{}; T.w = std::min(64, k - jb); T.ld = f; T.ldo = f; T.id = s;
                T.blk = front(s) + jb + (size_t)f * jb; T.out = Pn + S.panel_off + jb + (size_t)f * jb;
                potrf.push_back(T);
            }
            launches.push_back({1, (int)pfirst, (int)(potrf.size() - pfirst), 1});
            size_t gfirst = gemm.size();
            for (int s : lev) {      // F[below, J] <- F[below, J] Dinv^T
                const Supernode &S = F.sn[s]; const int k = S.ncols, f = k + S.nrows;
                if (k <= jb) continue;
                const int w = std::min(64, k - jb);
                double *X = front(s) + (jb + w) + (size_t)f * jb;
                add_gemm(X, f, Pn + S.panel_off + jb + (size_t)f * jb, f, X, f, f - jb - w, w, w, GEMM_TRANS_B, 1.0, 0.0);
            }
            close_gemm(gfirst);
            gfirst = gemm.size();
            for (int s : lev) {      // F[below, below] -= F[below, J] F[below, J]^T (lower tiles)
                const Supernode &S = F.sn[s]; const int k = S.ncols, f = k + S.nrows;
                if (k <= jb) continue;
                const int w = std::min(64, k - jb);
                const double *X = front(s) + (jb + w) + (size_t)f * jb;
                add_gemm(X, f, X, f, front(s) + (jb + w) + (size_t)f * (jb + w), f, f - jb - w, f - jb - w, w, GEMM_TRANS_B | GEMM_LOWER_TILES, -1.0, 1.0);
            }
            close_gemm(gfirst);
        }
        // P[0:k, 0:k] = L11^-1 from the block inverses, by doubling: X[bottom, top] = -X[bottom, bottom] (L[bottom, top] X[top, top]);
        // the inner product lands (transposed) in the unused upper triangle of the front
        const int nbmax = (kmax + 63) / 64;
        for (int half = 1; half < nbmax; half *= 2) {
            for (int pass = 0; pass < 2; ++pass) {
                const size_t gfirst = gemm.size();
                for (int s : lev) {
                    const Supernode &S = F.sn[s]; const int k = S.ncols, f = k + S.nrows, nb = (k + 63) / 64;
                    for (int gb = 0; gb + half < nb; gb += 2 * half) {
                        const int g0 = 64 * gb, sz = 64 * half, h = std::min(k, 64 * (gb + 2 * half)) - (g0 + sz);
                        double *Tt = front(s) + g0 + (size_t)f * (g0 + sz);      // (sz x h): transpose of L[bottom, top] X[top, top]
                        double *Ps = Pn + S.panel_off;
                        if (pass == 0) add_gemm(Ps + g0 + (size_t)f * g0, f, front(s) + (g0 + sz) + (size_t)f * g0, f, Tt, f, sz, h, sz, GEMM_TRANS_A | GEMM_TRANS_B, 1.0, 0.0);
                        else add_gemm(Ps + (g0 + sz) + (size_t)f * (g0 + sz), f, Tt, f, Ps + (g0 + sz) + (size_t)f * g0, f, h, sz, h, GEMM_TRANS_B, -1.0, 0.0);
                    }
                }
                close_gemm(gfirst);
            }
        }
        {      // P[k:f, :] = L21 L11^-1
            const size_t gfirst = gemm.size();
            for (int s : lev) {
                const Supernode &S = F.sn[s]; const int k = S.ncols, f = k + S.nrows;
                add_gemm(front(s) + k, f, Pn + S.panel_off, f, Pn + S.panel_off + k, f, S.nrows, k, k, GEMM_K_FROM_COL, 1.0, 0.0);
            }
            close_gemm(gfirst);
        }
    }
    {      // roots: (L L^T)^-1 = L^-T L^-1, full symmetric
        const size_t gfirst = gemm.size();
        for (int s = 0; s < ns; ++s) {
            const Supernode &S = F.sn[s];
            if (S.root_inv_off < 0) continue;
            const int k = S.ncols;
            add_gemm(Pn + S.panel_off, k, Pn + S.panel_off, k, Pn + S.root_inv_off, root_inv_ld(k), k, k, k, GEMM_TRANS_A | GEMM_K_FROM_MAX, 1.0, 0.0);
        }
        close_gemm(gfirst);
    }
    DF_CHK(hipMalloc(&d_gemm, sizeof(GemmTask) * std::max<size_t>(gemm.size(), 1)));
    DF_CHK(hipMalloc(&d_potrf, sizeof(PotrfTask) * std::max<size_t>(potrf.size(), 1)));
    DF_CHK(hipMalloc(&d_ext, sizeof(ExtendTask) * std::max<size_t>(ext.size(), 1)));
    DF_CHK(hipMemcpyAsync(d_gemm, gemm.data(), sizeof(GemmTask) * gemm.size(), hipMemcpyHostToDevice, st));
    DF_CHK(hipMemcpyAsync(d_potrf, potrf.data(), sizeof(PotrfTask) * potrf.size(), hipMemcpyHostToDevice, st));
    DF_CHK(hipMemcpyAsync(d_ext, ext.data(), sizeof(ExtendTask) * ext.size(), hipMemcpyHostToDevice, st));
    DF_CHK(hipStreamSynchronize(st));
    const double t_setup = now_s() - t0;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, (const int64_t *)d_adst, (const int *)d_asrc, (const double *)d_aval, d_fronts);
    constexpr int YMAX = 32768;
    for (const Launch &Lc : launches) {
        for (int off = 0; off < Lc.count; off += YMAX) {
            const int cnt = std::min(YMAX, Lc.count - off);
            if (Lc.kind == 0) hipLaunchKernelGGL(extend_add_kernel, dim3(Lc.gx, cnt), dim3(256), 0, st, (const ExtendTask *)(d_ext + Lc.first + off));
            else if (Lc.kind == 1) hipLaunchKernelGGL(potrf_inv_kernel, dim3(cnt), dim3(256), 0, st, (const PotrfTask *)(d_potrf + Lc.first + off), d_fail);
            else hipLaunchKernelGGL(gemm_f64_kernel, dim3(Lc.gx, cnt), dim3(256), 0, st, (const GemmTask *)(d_gemm + Lc.first + off));
        }
    }
    DF_CHK(hipGetLastError());
    int failed = 0;
    DF_CHK(hipMemcpyAsync(&failed, d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
    DF_CHK(hipStreamSynchronize(st));
#undef DF_CHK
    cleanup();
    F.panels.clear(); F.panels.shrink_to_fit();
    ctx->info.t_numeric_s = now_s() - t0;
    ctx->info.device_factor = 1;
    if (getenv("ADMM_HIP_VERBOSE")) {
        double flop = 0.0;      // products as issued (triangular operands are multiplied as dense blocks from their first non-zero block on)
        for (const GemmTask &T : gemm) {
            const int tm = (T.m + 63) / 64, tn = (T.n + 63) / 64;
            for (int ti = 0; ti < tm; ++ti) for (int tj = 0; tj < tn; ++tj) {
                if ((T.flags & GEMM_LOWER_TILES) && tj > ti) continue;
                const int k0 = (T.flags & GEMM_K_FROM_COL) ? 64 * tj : ((T.flags & GEMM_K_FROM_MAX) ? 64 * std::max(ti, tj) : 0);
                flop += 2.0 * std::min(64, T.m - 64 * ti) * std::min(64, T.n - 64 * tj) * std::max(0, T.k - k0);
            }
        }
        const double t_gpu = ctx->info.t_numeric_s - t_setup;
        fprintf(stderr, "admm_hip: numeric factorization on the device: %.3f s = %.3f s host set-up (maps, task records, uploads) + %.3f s of kernels (%.1f GFLOP in products: %.1f TFLOP/s), "
                        "fronts %.2f GB, %zu launches, %zu gemm / %zu potrf / %zu extend-add tasks\n",
                ctx->info.t_numeric_s, t_setup, t_gpu, flop * 1e-9, flop / t_gpu * 1e-12, ftot * 8e-9, launches.size(), gemm.size(), potrf.size(), ext.size());
    }
    if (failed) return fail(ctx, ADMM_ERR_FACTOR, "system matrix is not positive definite (supernode %d)", failed - 1);
    return ADMM_OK;
}

// the factor's panels into ctx->d_panels (allocated): computed on the device, or on the host and copied
int panels_to_device(admm_hip_ctx *ctx) {
    if (ctx->device_numeric) {
        const int rc = device_factorize(ctx);
        if (rc != ADMM_DEVFACTOR_NOFIT) return rc;
        fprintf(stderr, "admm_hip: the fronts do not fit the device memory that is free, factoring on the host\n");
        const int err = factorize(ctx->A, ctx->F, std::max(1, (int)ctx->info.host_threads));
        if (err) return fail(ctx, ADMM_ERR_FACTOR, "system matrix is not positive definite (supernode %d)", err - 1);
        ctx->info.t_numeric_s = ctx->F.t_numeric;
    }
    ctx->info.device_factor = 0;
    HIPCHK(hipMemcpy(ctx->d_panels, ctx->F.panels.data(), ctx->F.panels.size() * sizeof(double), hipMemcpyHostToDevice));
    return ADMM_OK;
}

// ---- subtree sharding: who owns which supernode ------------------------------------------------------
// Split the heaviest open subtree at its root (the root joins the replicated top) until there are >= 4 open subtrees per
// rank, then give the subtrees to the ranks largest first (LPT).  Every vertex separator is a supernode, so an element
// whose nodes are not all in the top lies inside exactly ONE subtree plus its ancestors: it goes to that subtree's rank.
// `owner` <- part of every supernode (-1 = top) for `parts` parts; returns the loads through `load`, counts through n_top / n_sub
void subtree_owners(const Factor &F, int parts, std::vector<int> &owner, std::vector<double> &load, int &n_top, size_t &n_sub) {
    const int ns = (int)F.sn.size(), world = parts;
    owner.assign(ns, 0);
    std::vector<double> weight(ns, 0.0);
    std::vector<std::vector<int> > kids(ns);
    for (int s = 0; s < ns; ++s) {   // postorder: children come before parents
        weight[s] += (double)(F.sn[s].ncols + F.sn[s].nrows) * F.sn[s].ncols;
        if (F.sn[s].parent >= 0) { weight[F.sn[s].parent] += weight[s]; kids[F.sn[s].parent].push_back(s); }
    }
    std::vector<char> top(ns, 0);
    auto cmp = [&](int a, int b) { return weight[a] < weight[b] || (weight[a] == weight[b] && a > b); };
    std::vector<int> open, done;
    for (int s = 0; s < ns; ++s) if (F.sn[s].parent < 0) open.push_back(s);
    std::make_heap(open.begin(), open.end(), cmp);
    // LPT assignment of the current subtrees; returns max load / mean load
    load.assign(world, 0.0);
    std::vector<int> root_owner(ns, -2);
    auto assign = [&]() {
        std::vector<int> all(done); all.insert(all.end(), open.begin(), open.end());
        std::sort(all.begin(), all.end(), [&](int a, int b) { return weight[a] > weight[b] || (weight[a] == weight[b] && a < b); });
        std::fill(load.begin(), load.end(), 0.0); std::fill(root_owner.begin(), root_owner.end(), -2);
        double tot = 0.0;
        for (int s : all) { const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin()); root_owner[s] = r; load[r] += weight[s]; tot += weight[s]; }
        return tot > 0.0 ? *std::max_element(load.begin(), load.end()) * world / tot : 1.0;
    };
    // every split moves a separator into the replicated top: stop as soon as there is one subtree per rank and the loads
    // balance within 15 %, at the latest at four subtrees per rank
    while (!open.empty()) {
        const int have = (int)(open.size() + done.size());
        if (have >= 4 * world || (have >= world && assign() <= 1.15)) break;
        std::pop_heap(open.begin(), open.end(), cmp);
        const int s = open.back(); open.pop_back();
        if (kids[s].empty()) { done.push_back(s); continue; }
        top[s] = 1;
        for (int c : kids[s]) { open.push_back(c); std::push_heap(open.begin(), open.end(), cmp); }
    }
    assign();
    done.insert(done.end(), open.begin(), open.end());
    for (int s = ns - 1; s >= 0; --s) {   // parents before children
        if (top[s]) owner[s] = -1;
        else if (root_owner[s] != -2) owner[s] = root_owner[s];
        else owner[s] = owner[F.sn[s].parent];
    }
    n_top = 0; for (int s = 0; s < ns; ++s) n_top += top[s];
    n_sub = done.size();
}

void partition_subtrees(admm_hip_ctx *ctx) {
    const Factor &F = ctx->F;
    const int ns = (int)F.sn.size(), world = ctx->world;
    ctx->sn_owner.assign(ns, 0); ctx->node_owner.assign(F.n, 0);
    ctx->grp_owner.clear();
    if (ctx->pipe > 1 && (world > 1 || ctx->dense)) { ctx->pipe = 0; ctx->groups = 1; }      // the pipeline is a one-GPU mode of the panel sweeps
    if (!(ctx->shard_mode == 1 && world > 1) && ctx->groups > 1 && !ctx->dense) {      // concurrent groups on this GPU
        std::vector<double> load; int nt = 0; size_t nsub = 0;
        subtree_owners(F, ctx->groups, ctx->grp_owner, load, nt, nsub);
        if (getenv("ADMM_HIP_VERBOSE")) {
            fprintf(stderr, "admm_hip: %d concurrent subtree groups: %d top supernodes, %zu subtrees, load per group (1e6 entries):", ctx->groups, nt, nsub);
            for (double l : load) fprintf(stderr, " %.1f", l * 1e-6);
            fprintf(stderr, "\n");
        }
        ctx->pipe_node_group.clear(); ctx->pipe_nodes.clear();
        if (ctx->pipe > 1 && world == 1) {      // nodes of every group (supernodes are contiguous runs of the factor order; neighbours merge)
            ctx->pipe_node_group.assign(F.n, -1);
            ctx->pipe_nodes.assign(ctx->pipe + 1, {});
            std::vector<int> by_first(ns);
            std::iota(by_first.begin(), by_first.end(), 0);
            std::sort(by_first.begin(), by_first.end(), [&](int a, int b) { return F.sn[a].first < F.sn[b].first; });
            for (int s : by_first) {
                const int g = ctx->grp_owner[s], a = F.sn[s].first, e = a + F.sn[s].ncols;
                for (int j = a; j < e; ++j) ctx->pipe_node_group[j] = g;
                std::vector<std::pair<int, int> > &R = ctx->pipe_nodes[g < 0 ? ctx->pipe : g];
                if (!R.empty() && R.back().second == a) R.back().second = e; else R.push_back({a, e});
            }
        }
    }
    if (ctx->shard_mode != 1 || world <= 1) return;
    std::vector<double> load; int nt = 0; size_t nsub = 0;
    subtree_owners(F, world, ctx->sn_owner, load, nt, nsub);
    for (int s = 0; s < ns; ++s) for (int j = 0; j < F.sn[s].ncols; ++j) ctx->node_owner[F.sn[s].first + j] = ctx->sn_owner[s];
    if (getenv("ADMM_HIP_VERBOSE")) {
        fprintf(stderr, "admm_hip: subtree sharding: %d top supernodes, %zu subtrees, load per rank (1e6 entries):", nt, nsub);
        for (double l : load) fprintf(stderr, " %.1f", l * 1e-6);
        fprintf(stderr, "\n");
    }
}

// this rank's elements of every batch
void assign_elements(admm_hip_ctx *ctx) {
    const Factor &F = ctx->F;
    int64_t nloc = 0;
    for (Batch &b : ctx->batches) {
        b.local.clear();
        if (ctx->shard_mode == 1 && ctx->world > 1) {
            for (int e = 0; e < b.n_total; ++e) {
                int owner = -1;
                const int32_t *nd; const int nn = b.elem_nodes(e, &nd);
                for (int c = 0; c < nn && owner < 0; ++c) owner = ctx->node_owner[F.iperm[nd[c]]];
                if (owner < 0) owner = e % ctx->world;      // all nodes in the replicated top: any rank will do
                if (owner == ctx->rank) b.local.push_back(e);
            }
        } else {   // contiguous ranges (reference order preserved inside a rank)
            const int first = (int)((int64_t)b.n_total * ctx->rank / ctx->world), end = (int)((int64_t)b.n_total * (ctx->rank + 1) / ctx->world);
            for (int e = first; e < end; ++e) b.local.push_back(e);
        }
        b.grp_ptr.clear(); b.grp_blk.clear();
        if (ctx->pipe > 1 && ctx->world == 1 && !ctx->grp_owner.empty()) {
            // group-major: an element belongs to the group of its first node below the top (all its nodes below the top lie in ONE
            // subtree); elements entirely inside the top are dealt round-robin.  Reference order is kept inside a group.
            const int G = ctx->pipe;
            std::vector<int> grp(b.local.size());
            for (size_t el = 0; el < b.local.size(); ++el) {
                int g = -1;
                const int32_t *nd; const int nn = b.elem_nodes(b.local[el], &nd);
                for (int c = 0; c < nn && g < 0; ++c) g = ctx->pipe_node_group[F.iperm[nd[c]]];
                grp[el] = g < 0 ? (int)(b.local[el] % G) : g;
            }
            std::vector<int32_t> sorted; sorted.reserve(b.local.size());
            b.grp_ptr.assign(G + 1, 0); b.grp_blk.assign(G + 1, 0);
            for (int g = 0; g < G; ++g) {
                for (size_t el = 0; el < b.local.size(); ++el) if (grp[el] == g) sorted.push_back(b.local[el]);
                b.grp_ptr[g + 1] = (int)sorted.size();
                b.grp_blk[g + 1] = b.grp_blk[g] + (b.grp_ptr[g + 1] - b.grp_ptr[g] + admm_dev::LOCAL_BLOCK - 1) / admm_dev::LOCAL_BLOCK;
            }
            b.local.swap(sorted);
        }
        b.n_local = (int)b.local.size();
        nloc += b.n_local;
    }
    ctx->info.n_elems_local = nloc;
}

// XCD-aware order of a level's work items.  Workgroups are dealt round-robin to the 8 XCDs (workgroup i -> XCD i mod 8), each
// with its own L2: in plain order the tiles of ONE supernode land on all eight, and every L2 fetches that supernode's staged
// vector (y, contribution lists, the children's contributions / x of its rows) from HBM again.  Here every supernode of a level
// is given to one XCD (longest first onto the least loaded) and the list is interleaved so that its items get workgroup ids of
// that XCD; queues of unequal length are padded with empty items (k = r = 0: the kernels do nothing for them).  `group` = items
// per workgroup (the wave-per-tile forward kernel packs several).  Levels with few supernodes keep the plain order: there every
// XCD is needed for each of them.
void xcd_order(std::vector<admm_dev::SweepItem> &items, int group, int min_supernodes) {
    const int NX = 8;
    std::vector<std::pair<int, int> > runs;      // (first item, count) per supernode; a supernode's items are consecutive
    for (size_t i = 0; i < items.size();) { size_t j = i; while (j < items.size() && items[j].s == items[i].s) ++j; runs.push_back({(int)i, (int)(j - i)}); i = j; }
    if ((int)runs.size() < min_supernodes) return;
    std::vector<int> ord(runs.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return runs[a].second > runs[b].second; });
    std::vector<std::vector<admm_dev::SweepItem> > q(NX);
    for (int r : ord) {
        int best = 0;
        for (int x = 1; x < NX; ++x) if (q[x].size() < q[best].size()) best = x;
        q[best].insert(q[best].end(), items.begin() + runs[r].first, items.begin() + runs[r].first + runs[r].second);
    }
    size_t len = 0;
    for (int x = 0; x < NX; ++x) len = std::max(len, (q[x].size() + group - 1) / group * group);
    admm_dev::SweepItem none{};
    std::vector<admm_dev::SweepItem> out;
    out.reserve(len * NX);
    for (size_t g0 = 0; g0 < len; g0 += group) for (int x = 0; x < NX; ++x) for (int t = 0; t < group; ++t) out.push_back(g0 + t < q[x].size() ? q[x][g0 + t] : none);
    items.swap(out);
}

// ---- device upload ----------------------------------------------------------
template <class T> std::vector<T> permute_nodes(const std::vector<T> &h, const std::vector<int> &perm, int comps) {
    std::vector<T> o(h.size());
    for (size_t i = 0; i < perm.size(); ++i) for (int c = 0; c < comps; ++c) o[comps * i + c] = h[comps * (size_t)perm[i] + c];
    return o;
}

#ifdef ADMM_SWEEP_PROFILE
// tools/sweep_timeline.py only (variant build): every workgroup of every sweep launch owns a slot of one stamp buffer (the stamps
// of the LAST iteration stay); meta: (tag, level, workgroups, KB, list, first slot) per launch
unsigned long long *g_swp_base; size_t g_swp_wgs; std::vector<int> g_swp_meta;
#endif
int upload_factor(admm_hip_ctx *ctx) {
#ifdef ADMM_SWEEP_PROFILE
    g_swp_wgs = 0; g_swp_meta.clear();
#endif
    Factor &F = ctx->F;
    const int ns = (int)F.sn.size();
    std::vector<int> first(ns), ncols(ns), nrows(ns);
    std::vector<int64_t> poff(ns), roff(ns), soff(ns), foff(ns);
    for (int s = 0; s < ns; ++s) { first[s] = F.sn[s].first; ncols[s] = F.sn[s].ncols; nrows[s] = F.sn[s].nrows; poff[s] = F.sn[s].panel_off; roff[s] = F.sn[s].rows_off; soff[s] = F.sn[s].slot_off; foff[s] = F.sn[s].front_off; }
    TRY(dalloc(ctx, &ctx->d_panels, (size_t)std::max<int64_t>(F.panels_size, 1)));
    TRY(panels_to_device(ctx));
    TRY(upload(ctx, &ctx->d_sn_first, first)); TRY(upload(ctx, &ctx->d_sn_ncols, ncols)); TRY(upload(ctx, &ctx->d_sn_nrows, nrows));
    TRY(upload(ctx, &ctx->d_sn_panel_off, poff)); TRY(upload(ctx, &ctx->d_sn_rows_off, roff)); TRY(upload(ctx, &ctx->d_sn_slot_off, soff));
    TRY(upload(ctx, &ctx->d_rows, F.rows));
    TRY(upload(ctx, &ctx->d_sn_front_off, foff));
    TRY(upload(ctx, &ctx->d_cg_ptr, F.cg_ptr)); TRY(upload(ctx, &ctx->d_cg_slot, F.cg_slot));
    ctx->d_cg4 = nullptr;
    if (!F.cg4.empty()) TRY(upload(ctx, &ctx->d_cg4, F.cg4));
    TRY(dalloc(ctx, &ctx->d_c, 3 * (size_t)std::max<int64_t>(F.n_slots, 1)));
    ctx->d_ainv = nullptr;
    if (ctx->dense) TRY(upload(ctx, &ctx->d_ainv, ctx->Ainv));
    ctx->levels.assign(F.levels.size(), LevelDev());
    // Split level: the first level holding a supernode wider than FWD_SMALL_KMAX.  Below it every forward item is
    // a wave item and the backward kernel takes 4 columns per wave; from it upwards block items / ADMM_BWD_BIG_CW.
    // Per level, by its widest supernode: forward as wave items (one wave per 64-row tile, k <= fwd_small_k <= 64) or block
    // items (NW waves split a tile's columns); backward with 4 columns per wave (k <= bwd_small_k) or one.
    std::vector<int> level_kmax(F.levels.size(), 0);
    for (size_t l = 0; l < F.levels.size(); ++l) for (int s : F.levels[l]) level_kmax[l] = std::max(level_kmax[l], F.sn[s].ncols);
    const int fwd_small_k = std::min(ctx->fwd_small_k, admm_dev::FWD_SMALL_KMAX);
    const bool subtree = ctx->shard_mode == 1 && ctx->world > 1;
    ctx->levels_top.assign(subtree ? F.levels.size() : 0, LevelDev());
    const bool grouped = !subtree && !ctx->grp_owner.empty();
    if (grouped) {
        while ((int)ctx->side_streams.size() < ctx->groups - 1) {
            hipStream_t st; hipEvent_t e;
            HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); ctx->side_streams.push_back(st);
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_join.push_back(e);
        }
        if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    }
    ctx->levels_side.assign(grouped ? ctx->groups - 1 : 0, std::vector<LevelDev>(F.levels.size()));
    ctx->levels_gtop.assign(grouped ? F.levels.size() : 0, LevelDev());
    // passes: (which supernodes, into which list).  Plain: all -> levels.  Subtree sharding: this rank's -> levels, the replicated
    // top -> levels_top.  Concurrent groups: group 0 -> levels, group g -> levels_side[g - 1], the top -> levels_gtop.
    struct Pass { int want; std::vector<LevelDev> *into; };
    std::vector<Pass> passes;
    if (subtree) { passes.push_back({ctx->rank, &ctx->levels}); passes.push_back({-1, &ctx->levels_top}); }
    else if (grouped) { passes.push_back({0, &ctx->levels}); for (int g = 1; g < ctx->groups; ++g) passes.push_back({g, &ctx->levels_side[g - 1]}); passes.push_back({-1, &ctx->levels_gtop}); }
    else passes.push_back({0, &ctx->levels});
    const std::vector<int> *own = subtree ? &ctx->sn_owner : (grouped ? &ctx->grp_owner : nullptr);
    for (const Pass &ps : passes) {
        for (size_t l = 0; l < F.levels.size(); ++l) {
            LevelDev &L = (*ps.into)[l];
            std::vector<admm_dev::SweepItem> sm, bg, bw;
            L.level = (int)l; L.mbytes = 0.0;
            const bool fwd_small = level_kmax[l] <= fwd_small_k, bwd_small = level_kmax[l] <= ctx->bwd_small_k;
            L.bwd_cw = bwd_small ? 4 : ADMM_BWD_BIG_CW;   // columns per wave in the backward kernel
            // eight columns per block pay where a level has thousands of columns (one staging of the vector per 8 instead of 4
            // columns); on levels with few columns the larger number of blocks matters more (50k-tet bar: 4 waves 222 us / 8: 228)
            int level_cols = 0;
            for (int s : F.levels[l]) if (!own || (*own)[s] == ps.want) level_cols += F.sn[s].ncols;
            L.bwd_nw = bwd_small ? ctx->bwd_small_nw : (level_cols >= ctx->bwd_nw_min_cols ? ctx->bwd_nw : 4);
            if (!bwd_small && ctx->bwd_cw2_min_cols > 0 && level_cols > ctx->bwd_cw2_min_cols && level_cols <= ctx->bwd_cw2_max_cols) L.bwd_cw = 2;
            for (int s : F.levels[l]) {
                if (own && (*own)[s] != ps.want) continue;
                const Supernode &S = F.sn[s];
                admm_dev::SweepItem it{};
                it.s = s; it.k = S.ncols; it.r = S.nrows; it.first = S.first;
                it.panel_off = S.panel_off; it.front_off = S.front_off; it.slot_off = S.slot_off; it.rows_off = S.rows_off;
                const int f = S.ncols + S.nrows;
                const int tiles = (f + 63) / 64;
                L.mbytes += 8e-6 * ((double)f * S.ncols - 0.5 * (double)S.ncols * (S.ncols - 1));
                if (S.root_inv_off >= 0 && ctx->root_inverse) {      // a root: x = (L L^T)^-1 t in the forward sweep, nothing in the backward sweep
                    L.roots.push_back({S.ncols, S.first, S.front_off, S.root_inv_off});
                    continue;
                }
                for (int t = 0; t < tiles; ++t) { it.part = t; if (fwd_small) sm.push_back(it); else bg.push_back(it); }
                const int chunks = (S.ncols + L.bwd_nw * L.bwd_cw - 1) / (L.bwd_nw * L.bwd_cw);
                for (int c = 0; c < chunks; ++c) { it.part = c; bw.push_back(it); }
            }
            if (ctx->xcd_min_supernodes > 0) { xcd_order(sm, ADMM_FWD_SMALL_WAVES, ctx->xcd_min_supernodes); xcd_order(bg, 1, ctx->xcd_min_supernodes); xcd_order(bw, 1, ctx->xcd_min_supernodes); }
            L.n_small = (int)sm.size(); L.n_big = (int)bg.size(); L.n_bwd = (int)bw.size();
            {
                int kmax = 0;
                for (const admm_dev::SweepItem &q : bg) kmax = std::max(kmax, q.k);
                // few tiles (all resident at once even with 16 waves each): the more waves share a tile's columns the shorter its chain
                L.big_nw = (int)bg.size() <= ctx->fwd_nw16_max_tiles ? 16 : (kmax <= ctx->fwd_nw4_kmax ? 4 : (kmax <= ctx->fwd_nw8_kmax ? 8 : 16));
            }
#ifdef ADMM_SWEEP_PROFILE
            {
                const int list = ps.want < 0 ? 100 : (&ps - &passes[0]);
                auto reg = [&](std::vector<admm_dev::SweepItem> &v, int group, int tag) {
                    if (v.empty()) return;
                    const int n_wg = ((int)v.size() + group - 1) / group;
                    for (size_t i = 0; i < v.size(); ++i) v[i].pad2 = (v[i].k || v[i].r) ? (long long)(g_swp_wgs + i / group) : -1;
                    for (int q : {tag, (int)l, n_wg, (int)(L.mbytes * 1024), list, (int)g_swp_wgs}) g_swp_meta.push_back(q);
                    g_swp_wgs += n_wg;
                };
                reg(sm, ADMM_FWD_SMALL_WAVES, 0); reg(bg, 1, L.big_nw); reg(bw, 1, 100 + 10 * L.bwd_cw + (L.bwd_nw == 16 ? 6 : L.bwd_nw));
            }
#endif
            TRY(upload(ctx, &L.d_small, sm)); TRY(upload(ctx, &L.d_big, bg)); TRY(upload(ctx, &L.d_bwd, bw));
        }
    }
#ifdef ADMM_SWEEP_PROFILE
    {
        unsigned long long *p = nullptr;
        TRY(dalloc(ctx, (double **)&p, 4 * g_swp_wgs + 4));
        HIPCHK(hipMemset(p, 0, sizeof(unsigned long long) * (4 * g_swp_wgs + 4)));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(admm_dev::g_sweep_prof), &p, sizeof(p)));
        g_swp_base = p;
    }
#endif
    // subtree sharding: the exchange lists (see shard_pack_kernel) and the node masks
    ctx->n_comm_top = ctx->n_comm_slots = 0;
    ctx->d_comm_top = ctx->d_comm_slots = nullptr; ctx->d_comm_mine = ctx->d_base_mask = ctx->d_keep_mask = nullptr; ctx->d_comm_buf = nullptr;
    if (subtree) {
        std::vector<int> top_nodes, slots; std::vector<unsigned char> mine, base(F.n), keep(F.n);
        for (int i = 0; i < F.n; ++i) {
            const int o = ctx->node_owner[i];
            if (o < 0) top_nodes.push_back(i);
            base[i] = keep[i] = (o == ctx->rank || (o < 0 && ctx->rank == 0)) ? 1 : 0;
        }
        for (int s = 0; s < ns; ++s) {      // roots of the owned subtrees: their contribution rows feed the top
            const int par = F.sn[s].parent;
            if (ctx->sn_owner[s] < 0 || par < 0 || ctx->sn_owner[par] >= 0) continue;
            for (int q = 0; q < F.sn[s].nrows; ++q) { slots.push_back((int)(F.sn[s].slot_off + q)); mine.push_back(ctx->sn_owner[s] == ctx->rank ? 1 : 0); }
        }
        ctx->n_comm_top = (int)top_nodes.size(); ctx->n_comm_slots = (int)slots.size();
        TRY(upload(ctx, &ctx->d_comm_top, top_nodes)); TRY(upload(ctx, &ctx->d_comm_slots, slots)); TRY(upload(ctx, &ctx->d_comm_mine, mine));
        TRY(upload(ctx, &ctx->d_base_mask, base)); TRY(upload(ctx, &ctx->d_keep_mask, keep));
        TRY(dalloc(ctx, &ctx->d_comm_buf, 3 * (size_t)std::max(1, ctx->n_comm_top + ctx->n_comm_slots)));
    }
    return ADMM_OK;
}

// is local element `el` the last one of its 64-element launch block?  (blocks restart at every pipeline group's first element)
static bool block_end(const Batch &b, int el) {
    int base = 0;
    if (!b.grp_ptr.empty()) { size_t g = 0; while (g + 2 < b.grp_ptr.size() && el >= b.grp_ptr[g + 1]) ++g; base = b.grp_ptr[g]; if (el + 1 == b.grp_ptr[g + 1]) return true; }
    return (el - base) % b.tpb == b.tpb - 1;
}
// number of launch blocks of a batch (tets: `tpb` elements per block; everything else LOCAL_BLOCK)
static int batch_blocks(const Batch &b) { return b.grp_blk.empty() ? (b.n_local + b.tpb - 1) / b.tpb : b.grp_blk.back(); }

int upload_all(admm_hip_ctx *ctx) {
    const double t0 = now_s();
    const int n = ctx->n_nodes;
    const Factor &F = ctx->F;
    HIPCHK(hipSetDevice(ctx->device_id));
    free_device(ctx);
    ctx->res_ready = false;
    {
        std::vector<double> px = permute_nodes(ctx->x, F.perm, 3), pv = permute_nodes(ctx->v, F.perm, 3), pm = permute_nodes(ctx->m3, F.perm, 3);
        TRY(upload(ctx, &ctx->d_x, px)); TRY(upload(ctx, &ctx->d_v, pv)); TRY(upload(ctx, &ctx->d_m3, pm));
        TRY(upload(ctx, &ctx->d_xcur, px));
        TRY(dalloc(ctx, &ctx->d_mxbar, 3 * (size_t)n)); TRY(dalloc(ctx, &ctx->d_y, 3 * (size_t)n)); TRY(dalloc(ctx, &ctx->d_w, 3 * (size_t)n));
        TRY(upload(ctx, &ctx->d_perm, F.perm)); TRY(dalloc(ctx, &ctx->d_stage, 6 * (size_t)n));
        TRY(upload(ctx, &ctx->d_iperm, F.iperm));
    }
    TRY(upload_factor(ctx));
    // batches: shard, sort corners, SoA upload
    int64_t slot = 0, nloc = 0;
    // pass 1: local ranges, corner order, incidence counts per (factor-order) node
    std::vector<int64_t> inc_ptr(n + 1, 0);
    for (Batch &b : ctx->batches) {
        if (b.kind == ADMM_KIND_GENERIC) {      // one slot per (element, node)
            b.slot_base = slot;
            for (int el = 0; el < b.n_local; ++el) {
                const int32_t *nd; const int nn = b.elem_nodes(b.local[el], &nd);
                for (int c = 0; c < nn; ++c) inc_ptr[F.iperm[nd[c]] + 1]++;
                slot += nn;
            }
            nloc += b.n_local;
            continue;
        }
        const int nn = ADMM_KIND_NODES[b.kind];
        b.slot_base = slot;
        const bool is_tri = b.kind == ADMM_KIND_TRI_STRAIN || b.kind == ADMM_KIND_TRI_AREA || b.kind == ADMM_KIND_TRI_FUNG;
        const bool sort_corners = (b.kind >= ADMM_KIND_TET_LINEAR && b.kind <= ADMM_KIND_TET_STVK) || is_tri;
        b.max_iter = 0;
        if (b.kind == ADMM_KIND_TET_NH || b.kind == ADMM_KIND_TET_STVK)
            for (int e = 0; e < b.n_total; ++e) b.max_iter = std::max(b.max_iter, (int)b.params[(size_t)e * 3 + 2]);
        b.corner_perm.assign((size_t)b.n_total * nn, 0);
        b.prered = ctx->tet_prered && b.kind >= ADMM_KIND_TET_LINEAR && b.kind <= ADMM_KIND_TET_STVK;
        b.tpb = admm_dev::LOCAL_BLOCK;
        if ((b.kind == ADMM_KIND_TET_NH || b.kind == ADMM_KIND_TET_STVK) && b.grp_ptr.empty()) {
            // Fewer tets per one-wave block (the lanes beyond idle) shortens the union of paths a wave executes.  Measured (profiles/r04/underfilled.txt):
            // it pays only where the launch is under-filled AND ends in a long tail of a few pathological tets -- BASELINE configs[2] (50 700 StVK tets)
            // at frame 14: local step 136 -> 126 us (32 per block) -> 120 us (16); the same bar at frame 8: 78 -> 76 -> 92 us; Neo-Hookean bars of
            // 5 400 / 18 000 / 50 700 / 125 000 tets: 42 -> 47, 69 -> 72, 73 -> 75, 84 -> 98 us with 32 per block.  No size rule separates the cases
            // (the tail comes and goes with the deformation), so the default stays 64; ADMM_HIP_TPB sets it by hand.
            if (ctx->tet_tpb > 0) b.tpb = ctx->tet_tpb;
        }
        std::vector<int> blk_nodes;       // prered: the nodes of the current 64-tet block
        for (int el = 0; el < b.n_local; ++el) {
            const int e = b.local[el];
            const int *id = b.idx.data() + (size_t)e * nn;
            int ord[4] = {0, 1, 2, 3};
            if (sort_corners) std::stable_sort(ord, ord + nn, [&](int a, int c) { return id[a] < id[c]; });
            for (int c = 0; c < nn; ++c) {
                b.corner_perm[(size_t)e * nn + c] = ord[c];
                if (b.prered) blk_nodes.push_back(F.iperm[id[ord[c]]]); else inc_ptr[F.iperm[id[ord[c]]] + 1]++;
            }
            if (b.prered && (block_end(b, el) || el + 1 == b.n_local)) {      // one slot per distinct node of the block
                std::sort(blk_nodes.begin(), blk_nodes.end());
                blk_nodes.erase(std::unique(blk_nodes.begin(), blk_nodes.end()), blk_nodes.end());
                for (int pn : blk_nodes) inc_ptr[pn + 1]++;
                slot += (int64_t)blk_nodes.size();
                blk_nodes.clear();
            }
        }
        if (!b.prered) slot += (int64_t)b.n_local * nn;
        nloc += b.n_local;
    }
    int64_t maxdeg = 0;
    for (int i = 0; i < n; ++i) { maxdeg = std::max(maxdeg, inc_ptr[i + 1]); inc_ptr[i + 1] += inc_ptr[i]; }
    std::vector<int64_t> inc_pos(inc_ptr.begin(), inc_ptr.end() - 1);
    // RHS slot layout.  Node-sorted (a node's incidences contiguous) makes the gather's lanes -- one per (node, component) --
    // read 24-byte pieces 0.5 KB apart; rank-major (all nodes' r-th incidence contiguous) makes neighbouring lanes read
    // neighbouring words for every r, and neighbouring tets of a wave write neighbouring slots.  Same summation order per node
    // (r ascending = batch, element, corner), so the results are bitwise the same.  It pads every node to the largest incidence
    // count: used unless that more than doubles the array (meshes with a few very high-valence nodes).
    ctx->slot_stride = (maxdeg * n <= 2 * inc_ptr[n] + 1024 && getenv("ADMM_HIP_SLOTS_NODE_SORTED") == nullptr) ? n : 0;
    // pass 2: device arrays; every corner gets the next slot of its node (fixed order: batch, element, corner)
    for (Batch &b : ctx->batches) {
        if (b.kind == ADMM_KIND_GENERIC) {
            // this rank's rows (CSR in factor-order dofs) and, per (element, node) slot and component, the rows that feed it
            std::vector<int> lrow, rptr(1, 0), col, sptr(1, 0), srow, sdst;
            std::vector<double> val;
            b.g_sval.clear(); b.g_srow_b.clear();
            for (int el = 0; el < b.n_local; ++el) {
                const int e = b.local[el];
                const int32_t *nd; const int nn = b.elem_nodes(e, &nd);
                for (int64_t r = b.g_elem_row[e]; r < b.g_elem_row[e + 1]; ++r) {
                    lrow.push_back((int)(b.g_row0 + r));
                    for (int64_t p = b.g_rowptr[r]; p < b.g_rowptr[r + 1]; ++p) { col.push_back(3 * F.iperm[b.g_col[p] / 3] + b.g_col[p] % 3); val.push_back(b.g_val[p]); }
                    rptr.push_back((int)col.size());
                }
                // the element's entries by column (ascending row inside a column): one pass, an element may span all nodes
                std::vector<std::pair<int32_t, int64_t> > bycol;      // (column, entry)
                for (int64_t rr = b.g_elem_row[e]; rr < b.g_elem_row[e + 1]; ++rr) for (int64_t p = b.g_rowptr[rr]; p < b.g_rowptr[rr + 1]; ++p) bycol.push_back({b.g_col[p], p});
                std::stable_sort(bycol.begin(), bycol.end(), [](const std::pair<int32_t, int64_t> &x, const std::pair<int32_t, int64_t> &y) { return x.first < y.first; });
                std::vector<int64_t> entry_row(b.g_rowptr[b.g_elem_row[e + 1]] - b.g_rowptr[b.g_elem_row[e]]);
                for (int64_t rr = b.g_elem_row[e]; rr < b.g_elem_row[e + 1]; ++rr) for (int64_t p = b.g_rowptr[rr]; p < b.g_rowptr[rr + 1]; ++p) entry_row[p - b.g_rowptr[b.g_elem_row[e]]] = rr;
                size_t q = 0;
                for (int c = 0; c < nn; ++c) {
                    const int pn = F.iperm[nd[c]];
                    const int64_t r = inc_pos[pn]++;
                    sdst.push_back(ctx->slot_stride ? (int)((r - inc_ptr[pn]) * ctx->slot_stride + pn) : (int)r);
                    for (int comp = 0; comp < 3; ++comp) {
                        for (; q < bycol.size() && bycol[q].first == 3 * nd[c] + comp; ++q) {
                            const int64_t p = bycol[q].second, rr = entry_row[p - b.g_rowptr[b.g_elem_row[e]]];
                            srow.push_back((int)(b.g_row0 + rr)); b.g_srow_b.push_back((int32_t)rr); b.g_sval.push_back(b.g_val[p]);
                        }
                        sptr.push_back((int)srow.size());
                    }
                }
            }
            b.g_lrows = (int)lrow.size(); b.g_lslots = (int)sdst.size();
            std::vector<double> coef(b.g_sval.size());
            for (size_t i = 0; i < coef.size(); ++i) { const double w = b.g_roww[b.g_srow_b[i]]; coef[i] = b.g_sval[i] * ((ctx->dt * ctx->dt) * (w * w)); }
            TRY(upload(ctx, &b.d_g_lrow, lrow)); TRY(upload(ctx, &b.d_g_rptr, rptr)); TRY(upload(ctx, &b.d_g_col, col)); TRY(upload(ctx, &b.d_g_val, val));
            TRY(upload(ctx, &b.d_g_sptr, sptr)); TRY(upload(ctx, &b.d_g_srow, srow)); TRY(upload(ctx, &b.d_g_sdst, sdst)); TRY(upload(ctx, &b.d_g_scoef, coef));
            continue;
        }
        const int nn = ADMM_KIND_NODES[b.kind], np = ADMM_KIND_PARAMS[b.kind], rows = ADMM_KIND_ROWS[b.kind], ist = idx_stride(b.kind);
        const int nl = b.n_local;
        std::vector<int> idx((size_t)std::max(nl, 1) * ist, 0), dst((size_t)std::max(nl, 1) * ist, 0);
        b.G.assign((size_t)12 * std::max(nl, 1), 0.0);
        std::vector<double> rest((size_t)12 * std::max(nl, 1), 0.0), par((size_t)std::max(np, 1) * std::max(nl, 1), 0.0), w2h2(std::max(nl, 1)), kbl(std::max(nl, 1)), w2(std::max(nl, 1));
        // prered: per block, the corners sorted by (node, lane, corner) -> pos4 (where a corner's share goes in the block's LDS
        // staging, one byte per corner), and per distinct node an entry (slot, end of its run in the staging)
        std::vector<unsigned int> pos4(b.prered ? (size_t)std::max(nl, 1) : 0, 0u);
        std::vector<int> bn_ptr(1, 0), bn_dst; std::vector<unsigned short> bn_end;
        struct Corner { int pn; unsigned short lc; };      // lc = lane * 4 + corner
        std::vector<Corner> blk_c;
        int blk_first = 0;
        for (int el = 0; el < nl; ++el) {
            const int e = b.local[el];
            const int *id = b.idx.data() + (size_t)e * nn;
            const int *ord = b.corner_perm.data() + (size_t)e * nn;
            for (int c = 0; c < nn; ++c) {
                const int pn = F.iperm[id[ord[c]]];
                idx[(size_t)el * ist + c] = pn;
                if (b.prered) { blk_c.push_back({pn, (unsigned short)((el - blk_first) * 4 + c)}); continue; }
                const int64_t r = inc_pos[pn]++;
                dst[(size_t)el * ist + c] = ctx->slot_stride ? (int)((r - inc_ptr[pn]) * ctx->slot_stride + pn) : (int)r;
            }
            if (b.prered && (block_end(b, el) || el + 1 == nl)) {
                std::stable_sort(blk_c.begin(), blk_c.end(), [](const Corner &x, const Corner &y) { return x.pn < y.pn || (x.pn == y.pn && x.lc < y.lc); });
                for (size_t k = 0; k < blk_c.size(); ++k) {
                    const int lane = blk_c[k].lc >> 2, c = blk_c[k].lc & 3;
                    pos4[(size_t)blk_first + lane] |= (unsigned int)k << (8 * c);
                    if (k + 1 == blk_c.size() || blk_c[k + 1].pn != blk_c[k].pn) {
                        const int pn = blk_c[k].pn;
                        const int64_t r = inc_pos[pn]++;
                        bn_dst.push_back(ctx->slot_stride ? (int)((r - inc_ptr[pn]) * ctx->slot_stride + pn) : (int)r);
                        bn_end.push_back((unsigned short)(k + 1));
                    }
                }
                bn_ptr.push_back((int)bn_dst.size());
                blk_c.clear(); blk_first = el + 1;
            }
            const double *R = &b.rest[(size_t)e * 12];
            if (b.kind >= ADMM_KIND_TET_LINEAR && b.kind <= ADMM_KIND_TET_STVK) {
                for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) rest[(size_t)(c + 4 * r) * nl + el] = R[ord[c] + 4 * r];
            } else if (b.kind == ADMM_KIND_TRI_STRAIN || b.kind == ADMM_KIND_TRI_AREA || b.kind == ADMM_KIND_TRI_FUNG) {
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 2; ++r) rest[(size_t)(c + 3 * r) * nl + el] = R[ord[c] + 3 * r];
            } else {
                for (int i = 0; i < 12; ++i) rest[(size_t)i * nl + el] = R[i];
            }
            {   // selector block in device corner order (residual tracking)
                double Gm[4][3]; int cols;
                element_G(b.kind, R, Gm, cols);
                for (int c = 0; c < nn; ++c) for (int r = 0; r < 3; ++r) b.G[(size_t)(3 * c + r) * nl + el] = Gm[ord[c]][r];
            }
            for (int p = 0; p < np; ++p) par[(size_t)p * nl + el] = b.params[(size_t)e * np + p];
            const double w = b.weight[e];
            w2[el] = w * w;
            w2h2[el] = (ctx->dt * ctx->dt) * (w * w);
            kbl[el] = b.params[(size_t)e * np] * b.measure[e];
        }
        b.d_pos4 = nullptr; b.d_bn_ptr = nullptr; b.d_bn_dst = nullptr; b.d_bn_end = nullptr;
        if (b.prered) { TRY(upload(ctx, &b.d_pos4, pos4)); TRY(upload(ctx, &b.d_bn_ptr, bn_ptr)); TRY(upload(ctx, &b.d_bn_dst, bn_dst)); TRY(upload(ctx, &b.d_bn_end, bn_end)); }
        TRY(upload(ctx, &b.d_idx, idx)); TRY(upload(ctx, &b.d_dst, dst)); TRY(upload(ctx, &b.d_rest, rest)); TRY(upload(ctx, &b.d_par, par));
        TRY(upload(ctx, &b.d_w2h2, w2h2)); TRY(upload(ctx, &b.d_kblend, kbl)); TRY(upload(ctx, &b.d_w2, w2));
        TRY(dalloc(ctx, &b.d_u, (size_t)rows * std::max(nl, 1))); TRY(dalloc(ctx, &b.d_z, (size_t)rows * std::max(nl, 1)));
        HIPCHK(hipMemset(b.d_u, 0, sizeof(double) * (size_t)rows * std::max(nl, 1)));
        HIPCHK(hipMemset(b.d_z, 0, sizeof(double) * (size_t)rows * std::max(nl, 1)));
        std::vector<double> st((size_t)4 * std::max(nl, 1), 1.0);
        TRY(upload(ctx, &b.d_state, st));
        TRY(dalloc(ctx, &b.d_niters, (size_t)std::max(nl, 1)));
        HIPCHK(hipMemset(b.d_niters, 0, sizeof(int) * (size_t)std::max(nl, 1)));
        b.d_order = nullptr; b.d_cost = nullptr; b.n_blocks_ordered = 0;
        {
            // more blocks than the chip holds at once (2 waves x 4 SIMDs x 256 CUs): the launch order matters
            const int nblk = batch_blocks(b);
            if ((b.kind == ADMM_KIND_TET_NH || b.kind == ADMM_KIND_TET_STVK) && ctx->tet_order && nblk > ctx->tet_order_min_blocks) {
                std::vector<int> ident(nblk); std::iota(ident.begin(), ident.end(), 0);
                for (size_t g = 0; g + 1 < b.grp_blk.size(); ++g) std::iota(ident.begin() + b.grp_blk[g], ident.begin() + b.grp_blk[g + 1], 0);      // per group: ids relative to the group's first block
                TRY(upload(ctx, &b.d_order, ident));
                TRY(dalloc(ctx, &b.d_cost, (size_t)nblk));
                HIPCHK(hipMemset(b.d_cost, 0, sizeof(unsigned int) * (size_t)nblk));
                b.n_blocks_ordered = nblk;
            }
        }
        if (b.kind == ADMM_KIND_ANCHOR) {
            std::vector<double> tg((size_t)3 * std::max(nl, 1), 0.0); std::vector<int> ac(std::max(nl, 1), 1);
            for (int el = 0; el < nl; ++el) { for (int j = 0; j < 3; ++j) tg[3 * (size_t)el + j] = b.targets[3 * (size_t)b.local[el] + j]; ac[el] = b.active[b.local[el]]; }
            TRY(upload(ctx, &b.d_targets, tg)); TRY(upload(ctx, &b.d_active, ac));
        }
    }
    ctx->info.n_elems_local = nloc;
    ctx->info.rhs_slots = slot;
    if (ctx->slot_stride) slot = maxdeg * n;
    if (slot >= (int64_t)1 << 31) return fail(ctx, ADMM_ERR_UNSUPPORTED, "more than 2^31 RHS slots");
    ctx->n_fslots = slot;
    TRY(dalloc(ctx, &ctx->d_fslot, 3 * (size_t)std::max<int64_t>(slot, 1)));
    HIPCHK(hipMemset(ctx->d_fslot, 0, sizeof(double) * 3 * (size_t)std::max<int64_t>(slot, 1)));
    TRY(upload(ctx, &ctx->d_inc_ptr, inc_ptr));
    if (ctx->n_gen_rows) {      // user-defined forces: device and pinned host images of the generic row space
        const size_t bytes = sizeof(double) * (size_t)ctx->n_gen_rows;
        TRY(dalloc(ctx, &ctx->d_gen_dx, (size_t)ctx->n_gen_rows)); TRY(dalloc(ctx, &ctx->d_gen_q, (size_t)ctx->n_gen_rows));
        HIPCHK(hipMemset(ctx->d_gen_dx, 0, bytes)); HIPCHK(hipMemset(ctx->d_gen_q, 0, bytes));
        if (!ctx->h_gen_dx) {
            HIPCHK(hipHostMalloc((void **)&ctx->h_gen_dx, bytes)); HIPCHK(hipHostMalloc((void **)&ctx->h_gen_u, bytes));
            HIPCHK(hipHostMalloc((void **)&ctx->h_gen_z, bytes)); HIPCHK(hipHostMalloc((void **)&ctx->h_gen_q, bytes));
            HIPCHK(hipEventCreateWithFlags(&ctx->gen_ev, hipEventDisableTiming));
        }
        std::memset(ctx->h_gen_dx, 0, bytes); std::memset(ctx->h_gen_u, 0, bytes); std::memset(ctx->h_gen_z, 0, bytes); std::memset(ctx->h_gen_q, 0, bytes);
    }
    // collision shapes and the general explicit forces (index lists in factor order)
    TRY(dalloc(ctx, &ctx->d_shapes, 1));
    HIPCHK(hipMemcpy(ctx->d_shapes, &ctx->shapes, sizeof(admm_dev::ShapeTable), hipMemcpyHostToDevice));
    for (Explicit &E : ctx->explicits) {
        std::vector<int> pidx(E.idx.size());
        for (size_t i = 0; i < E.idx.size(); ++i) pidx[i] = F.iperm[E.idx[i]];
        E.d_idx = nullptr;
        if (E.type == ADMM_EXPLICIT_WIND && E.n) {
            // dependency levels of the serial loop: level(t) = 1 + max level of earlier triangles sharing a node
            std::vector<int> last(n, 0), lev(E.n);
            int nlev = 0;
            for (int t = 0; t < E.n; ++t) {
                const int *q = &pidx[3 * (size_t)t];
                const int l = 1 + std::max(last[q[0]], std::max(last[q[1]], last[q[2]]));
                lev[t] = l; last[q[0]] = last[q[1]] = last[q[2]] = l; nlev = std::max(nlev, l);
            }
            std::vector<int> lptr(nlev + 1, 0);
            for (int t = 0; t < E.n; ++t) lptr[lev[t]]++;            // level l (1-based) counted into lptr[l]
            for (int l = 0; l < nlev; ++l) lptr[l + 1] += lptr[l];   // lptr[l] = end of level l = start of level l+1
            std::vector<int> pos(lptr.begin(), lptr.end() - 1), sorted(pidx.size());
            for (int t = 0; t < E.n; ++t) { const int d = pos[lev[t] - 1]++; for (int c = 0; c < 3; ++c) sorted[3 * (size_t)d + c] = pidx[3 * (size_t)t + c]; }
            E.n_levels = nlev;
            TRY(upload(ctx, &E.d_idx, sorted)); TRY(upload(ctx, &E.d_level_ptr, lptr));
        } else if (!pidx.empty()) TRY(upload(ctx, &E.d_idx, pidx));
    }
    HIPCHK(hipDeviceSynchronize());
    ctx->info.t_upload_s = now_s() - t0;
    return ADMM_OK;
}

BatchDev batch_dev(const admm_hip_ctx *ctx, const Batch &b) {
    BatchDev d{};
    d.n = b.n_local; d.e0 = 0; d.e1 = b.n_local; d.keep_z = ctx->keep_z ? 1 : 0; d.idx = b.d_idx; d.rest = b.d_rest; d.par = b.d_par; d.w2h2 = b.d_w2h2; d.kblend = b.d_kblend; d.w2 = b.d_w2;
    d.u = b.d_u; d.z = b.d_z; d.state = b.d_state; d.n_iters = b.d_niters;
    d.fslot = ctx->d_fslot; d.dst = b.d_dst; d.targets = b.d_targets; d.active = b.d_active;
    d.dx_override = b.d_dx_override;
    d.order = b.d_order; d.cost = b.d_cost;
    d.res_slots = ctx->d_res_slots; d.res_partial = b.d_res_partial;
    d.pos4 = b.d_pos4; d.bn_ptr = b.d_bn_ptr; d.bn_dst = b.d_bn_dst; d.bn_end = b.d_bn_end;
    d.tpb = b.tpb;
    return d;
}

FactorDev factor_dev(const admm_hip_ctx *ctx) {
    FactorDev f{};
    f.panels = ctx->d_panels; f.sn_first = ctx->d_sn_first; f.sn_ncols = ctx->d_sn_ncols; f.sn_nrows = ctx->d_sn_nrows;
    f.sn_panel_off = ctx->d_sn_panel_off; f.sn_rows_off = ctx->d_sn_rows_off; f.sn_slot_off = ctx->d_sn_slot_off;
    f.rows = ctx->d_rows; f.sn_front_off = ctx->d_sn_front_off; f.cg_ptr = ctx->d_cg_ptr; f.cg_slot = ctx->d_cg_slot; f.cg4 = (const int4 *)ctx->d_cg4;
    return f;
}

int max_lbfgs_iters(const Batch &b) { return b.max_iter; }

// local step: every batch kernel on x_cur
#ifdef ADMM_TET_PROFILE
// tools/probe/ls_predict_gpu.py only (variant build): per-tet trace of the next `cap` launches of the tet kernel
float *g_trace_base; int g_trace_cap, g_trace_n, g_trace_count;
void tet_trace_next(hipStream_t st) {
    float *p = (g_trace_base && g_trace_count < g_trace_cap) ? g_trace_base + 2 * (size_t)g_trace_count * g_trace_n : nullptr;
    ++g_trace_count;
    hipMemcpyToSymbolAsync(HIP_SYMBOL(admm_dev::g_tet_trace), &p, sizeof(p), 0, hipMemcpyHostToDevice, st);
}
#endif
// `group` >= 0 (pipeline groups, Batch::grp_ptr): only that group's elements of every batch, on stream `st`; group < 0 with a
// group-major layout: group after group on one stream (the serial launch of the same layout)
// The scene's batches as segments of ONE project_multi_kernel launch (false: launch batch after batch -- a kind without a segment
// body, more than MULTI_MAX batches, pipeline groups, or a tet batch with its anchors right behind it, which already is one launch).
bool build_multi(admm_hip_ctx *ctx, admm_dev::MultiBatch &mb, int &blocks) {
    using namespace admm_dev;
    mb = MultiBatch{}; blocks = 0;
    // segments in the order of what a block costs, dearest first (the L-BFGS kinds, then the closed-form ones) whatever order the scene listed
    // its forces in (results do not depend on the order)
    auto dearness = [](int kind) {
        switch (kind) {
        case ADMM_KIND_TET_NH: return 0; case ADMM_KIND_TET_STVK: return 1; case ADMM_KIND_TRI_FUNG: return 2; case ADMM_KIND_TET_LINEAR: return 3; case ADMM_KIND_TET_VOLUME: return 4;
        case ADMM_KIND_BEND: return 5; case ADMM_KIND_TRI_STRAIN: return 6; case ADMM_KIND_TRI_AREA: return 7; case ADMM_KIND_SPRING: return 8; default: return 9;
        }
    };
    std::vector<const Batch *> seq;
    for (const Batch &b : ctx->batches) seq.push_back(&b);
    std::stable_sort(seq.begin(), seq.end(), [&](const Batch *a, const Batch *c) { return dearness(a->kind) < dearness(c->kind); });
    for (const Batch *bp : seq) {
        const Batch &b = *bp;
        if (b.n_local == 0 || b.kind == ADMM_KIND_GENERIC) continue;
        int code = -1;
        switch (b.kind) {
        case ADMM_KIND_TET_NH: code = max_lbfgs_iters(b) <= 5 ? MK_TET_NH : -1; break;
        case ADMM_KIND_TET_STVK: code = max_lbfgs_iters(b) <= 5 ? MK_TET_STVK : -1; break;
        case ADMM_KIND_TET_LINEAR: code = MK_TET_LINEAR; break;
        case ADMM_KIND_TET_VOLUME: code = MK_TET_VOLUME; break;
        case ADMM_KIND_ANCHOR: code = MK_ANCHOR; break;
        case ADMM_KIND_SPRING: code = MK_SPRING; break;
        case ADMM_KIND_BEND: code = MK_BEND; break;
        case ADMM_KIND_TRI_STRAIN: code = MK_TRI_STRAIN; break;
        case ADMM_KIND_TRI_AREA: code = MK_TRI_AREA; break;
        case ADMM_KIND_TRI_FUNG: code = MK_TRI_FUNG; break;
        case ADMM_KIND_COLLISION: code = MK_COLLISION; break;
        default: break;
        }
        if (code < 0 || !b.grp_ptr.empty() || mb.n == MULTI_MAX) return false;
        mb.b[mb.n] = batch_dev(ctx, b);
        const int epl = (code == MK_BEND || code == MK_TRI_STRAIN || code == MK_TRI_AREA) ? MULTI_EPL : 1;      // these segments' blocks cover 64 * MULTI_EPL elements
        blocks += (batch_blocks(b) + epl - 1) / epl;
        mb.code[mb.n] = code; mb.blk_end[mb.n] = blocks; ++mb.n;
    }
    if (mb.n < 2) return false;
    if (mb.n == 2 && mb.code[0] <= MK_TET_VOLUME && mb.code[1] == MK_ANCHOR && ctx->fuse_anchor_tail) {
        std::vector<int> live;      // ... in list order: the tail rides along only when the anchors come right behind their tets
        for (size_t bi = 0; bi < ctx->batches.size(); ++bi) if (ctx->batches[bi].n_local > 0 && ctx->batches[bi].kind != ADMM_KIND_GENERIC) live.push_back((int)bi);
        if (live.size() == 2 && live[1] == live[0] + 1 && ctx->batches[live[1]].kind == ADMM_KIND_ANCHOR) return false;
    }
    return true;
}

// the side streams of the concurrent batches (created outside stream capture: admm_hip_step / local_step_only call this first)
int ensure_local_streams(admm_hip_ctx *ctx) {
    int n_large = 0;
    if (ctx->local_streams_max > 1)
        for (const Batch &b : ctx->batches) if (b.kind != ADMM_KIND_GENERIC && b.grp_ptr.empty() && b.n_local >= ctx->local_streams_min_elems) ++n_large;
    if (n_large < 2) return ADMM_OK;
    while ((int)ctx->local_side.size() < std::min(n_large, ctx->local_streams_max) - 1) {
        hipStream_t q; hipEvent_t e;
        HIPCHK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking)); ctx->local_side.push_back(q);
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->local_join.push_back(e);
    }
    if (!ctx->local_fork) HIPCHK(hipEventCreateWithFlags(&ctx->local_fork, hipEventDisableTiming));
    return ADMM_OK;
}

int launch_local(admm_hip_ctx *ctx, int only_batch = -1, int group = -1, hipStream_t st = nullptr, bool track = false) {
    using namespace admm_dev;
    const bool plain = !st && only_batch < 0 && group < 0 && !track;
    if (!st) st = ctx->stream;
    // concurrent batches: the second, third ... LARGE batch of the scene each on a side stream (fork after what is on the context's stream, join
    // before the RHS gather); everything else, and everything in the special launch modes, on `st`
    int n_large = 0, side_used = 0;
    if (plain && ctx->local_streams_max > 1)
        for (const Batch &b : ctx->batches) if (b.kind != ADMM_KIND_GENERIC && b.grp_ptr.empty() && b.n_local >= ctx->local_streams_min_elems) ++n_large;
    const bool fan_out = n_large >= 2 && (int)ctx->local_side.size() >= std::min(n_large, ctx->local_streams_max) - 1 && ctx->local_fork;      // (streams: ensure_local_streams, outside any capture)
    if (fan_out) HIPCHK(hipEventRecord(ctx->local_fork, ctx->stream));
    hipStream_t const st_main = st;
    // the whole local step in one launch: every batch a segment of project_multi_kernel's grid
    if (plain && ctx->local_multi && !fan_out) {
        MultiBatch mb{}; int blocks = 0;
        if (build_multi(ctx, mb, blocks)) {
            hipLaunchKernelGGL(project_multi_kernel, dim3(blocks), dim3(LOCAL_BLOCK), 0, st, mb, (const double *)ctx->d_xcur, (const ShapeTable *)ctx->d_shapes);
            HIPCHK(hipGetLastError());
            return ADMM_OK;
        }
    }
    int large_seen = 0;
    bool skip_next = false;
    for (size_t bi = 0; bi < ctx->batches.size(); ++bi) {
        const Batch &b = ctx->batches[bi];
        if (only_batch >= 0 && (int)bi != only_batch) continue;
        if (skip_next) { skip_next = false; continue; }                  // (an anchor batch that went out with the tets before it)
        if (b.n_local == 0 || b.kind == ADMM_KIND_GENERIC) continue;     // user-defined forces: generic_begin / generic_finish
        const bool grouped = !b.grp_ptr.empty();
        st = st_main;
        if (fan_out && !grouped && b.n_local >= ctx->local_streams_min_elems) {
            const int lane = large_seen++ % std::min(n_large, ctx->local_streams_max);      // 0: the context's stream
            if (lane > 0) {
                st = ctx->local_side[lane - 1];
                if (lane > side_used) { HIPCHK(hipStreamWaitEvent(st, ctx->local_fork, 0)); side_used = lane; }
            }
        }
        const int g_first = grouped ? (group >= 0 ? group : 0) : 0, g_last = grouped ? (group >= 0 ? group + 1 : (int)b.grp_ptr.size() - 1) : 1;
      for (int g = g_first; g < g_last; ++g) {
        BatchDev d = batch_dev(ctx, b);
        if (grouped) {
            d.e0 = b.grp_ptr[g]; d.e1 = b.grp_ptr[g + 1];
            if (d.e1 == d.e0) continue;
            if (d.order) { d.order += b.grp_blk[g]; d.cost += b.grp_blk[g]; }
            if (d.res_partial) d.res_partial += b.grp_blk[g];
            if (d.bn_ptr) d.bn_ptr += b.grp_blk[g];
        }
        const bool trk = track && b.res_fused;
        dim3 grid((d.e1 - d.e0 + b.tpb - 1) / b.tpb), block(LOCAL_BLOCK);
        const double *x = ctx->d_xcur;
        // an anchor batch right behind a tet batch rides along in the tet launch (project_tet_kernel's tail)
        BatchDev tail{}; const int tail_block0 = (int)grid.x;
        const bool is_tet = b.kind == ADMM_KIND_TET_NH || b.kind == ADMM_KIND_TET_STVK || b.kind == ADMM_KIND_TET_LINEAR || b.kind == ADMM_KIND_TET_VOLUME;
        // (with residual tracking only if both batches track inside their kernels: the tail runs the tet launch's TRACK variant)
        if (is_tet && !grouped && only_batch < 0 && ctx->fuse_anchor_tail && bi + 1 < ctx->batches.size() && ctx->batches[bi + 1].kind == ADMM_KIND_ANCHOR && ctx->batches[bi + 1].n_local > 0 &&
            (!track || ctx->batches[bi + 1].res_fused == b.res_fused)) {
            tail = batch_dev(ctx, ctx->batches[bi + 1]);
            grid.x += (ctx->batches[bi + 1].n_local + LOCAL_BLOCK - 1) / LOCAL_BLOCK;
            skip_next = true;
        }
        switch (b.kind) {
        case ADMM_KIND_TET_NH:
#ifdef ADMM_TET_PROFILE
            tet_trace_next(st);
#endif
#define ADMM_TET(K, MM) do { if (trk) hipLaunchKernelGGL((project_tet_kernel<K, MM, true>), grid, block, ctx->tet_lds_pad, st, d, x, tail, tail_block0); \
                            else hipLaunchKernelGGL((project_tet_kernel<K, MM, false>), grid, block, ctx->tet_lds_pad, st, d, x, tail, tail_block0); } while (0)
            if (max_lbfgs_iters(b) <= 5) ADMM_TET(0, 5); else ADMM_TET(0, 10);
            break;
        case ADMM_KIND_TET_STVK:
            if (max_lbfgs_iters(b) <= 5) ADMM_TET(1, 5); else ADMM_TET(1, 10);
            break;
        case ADMM_KIND_TET_LINEAR: ADMM_TET(2, 1); break;
        case ADMM_KIND_TET_VOLUME: ADMM_TET(3, 1); break;
#undef ADMM_TET
        case ADMM_KIND_ANCHOR: if (trk) hipLaunchKernelGGL(project_anchor_kernel<true>, grid, block, 0, st, d, x); else hipLaunchKernelGGL(project_anchor_kernel<false>, grid, block, 0, st, d, x); break;
        case ADMM_KIND_SPRING: hipLaunchKernelGGL(project_spring_kernel, grid, block, 0, st, d, x); break;
        case ADMM_KIND_BEND: hipLaunchKernelGGL(project_bend_kernel, grid, block, 0, st, d, x); break;
        case ADMM_KIND_TRI_STRAIN: hipLaunchKernelGGL(project_tri_kernel<0>, grid, block, 0, st, d, x); break;
        case ADMM_KIND_TRI_AREA: hipLaunchKernelGGL(project_tri_kernel<1>, grid, block, 0, st, d, x); break;
        case ADMM_KIND_TRI_FUNG: hipLaunchKernelGGL(project_tri_kernel<2>, grid, block, 0, st, d, x); break;
        case ADMM_KIND_COLLISION: hipLaunchKernelGGL(project_collision_kernel, grid, block, 0, st, d, x, (const ShapeTable *)ctx->d_shapes); break;
        default: return fail(ctx, ADMM_ERR_UNSUPPORTED, "no kernel for kind %d", b.kind);
        }
      }
    }
    for (int q = 0; q < side_used; ++q) { HIPCHK(hipEventRecord(ctx->local_join[q], ctx->local_side[q])); HIPCHK(hipStreamWaitEvent(st_main, ctx->local_join[q], 0)); }
    HIPCHK(hipGetLastError());
    return ADMM_OK;
}

// user-defined forces, first half: D_i x of the generic rows -> pinned host memory (asynchronous; the built-in kernels are
// launched behind it and run while the host projects)
int generic_begin(admm_hip_ctx *ctx, const double *x) {
    if (!ctx->n_gen_rows) return ADMM_OK;
    for (const Batch &b : ctx->batches) if (b.kind == ADMM_KIND_GENERIC && b.g_lrows)
        hipLaunchKernelGGL(admm_dev::generic_dx_kernel, dim3((b.g_lrows + 255) / 256), dim3(256), 0, ctx->stream, b.g_lrows, (const int *)b.d_g_lrow, (const int *)b.d_g_rptr,
                           (const int *)b.d_g_col, (const double *)b.d_g_val, x, ctx->d_gen_dx);
    HIPCHK(hipMemcpyAsync(ctx->h_gen_dx, ctx->d_gen_dx, sizeof(double) * (size_t)ctx->n_gen_rows, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipEventRecord(ctx->gen_ev, ctx->stream));
    return ADMM_OK;
}
// second half: wait for the rows, run the caller's project() (Force::project for every user force, System.cpp:57-58),
// send z - u back and add the elements' shares of dt^2 D^T W^2 (z - u) to the per-node slots
int generic_finish(admm_hip_ctx *ctx) {
    if (!ctx->n_gen_rows) return ADMM_OK;
    HIPCHK(hipEventSynchronize(ctx->gen_ev));
    if (!ctx->project_hook) return fail(ctx, ADMM_ERR_STATE, "generic batches present but no project hook installed (admm_hip_set_project_hook)");
    if (ctx->project_hook(ctx->project_user, ctx->dt, ctx->n_gen_rows, ctx->h_gen_dx, ctx->h_gen_u, ctx->h_gen_z) != 0) return fail(ctx, ADMM_ERR_ARG, "project hook failed");
    for (int64_t r = 0; r < ctx->n_gen_rows; ++r) ctx->h_gen_q[r] = ctx->h_gen_z[r] - ctx->h_gen_u[r];
    HIPCHK(hipMemcpyAsync(ctx->d_gen_q, ctx->h_gen_q, sizeof(double) * (size_t)ctx->n_gen_rows, hipMemcpyHostToDevice, ctx->stream));
    for (const Batch &b : ctx->batches) if (b.kind == ADMM_KIND_GENERIC && b.g_lslots)
        hipLaunchKernelGGL(admm_dev::generic_rhs_kernel, dim3((3 * b.g_lslots + 255) / 256), dim3(256), 0, ctx->stream, 3 * b.g_lslots, (const int *)b.d_g_sptr, (const int *)b.d_g_srow,
                           (const double *)b.d_g_scoef, (const int *)b.d_g_sdst, (const double *)ctx->d_gen_q, ctx->d_fslot);
    HIPCHK(hipGetLastError());
    return ADMM_OK;
}

// group >= 0: only the nodes of that pipeline group's subtrees (group == ctx->pipe: the top), on stream `st`
int launch_rhs(admm_hip_ctx *ctx, int group = -1, hipStream_t st = nullptr) {
    if (!st) st = ctx->stream;
    auto range = [&](int a, int e) {
        const int n3 = 3 * (e - a);
        if (n3 > 0) hipLaunchKernelGGL(admm_dev::rhs_gather_kernel<false>, dim3((n3 + 255) / 256), dim3(256), 0, st, a, e, ctx->d_inc_ptr, ctx->slot_stride,
                                       ctx->d_fslot, ctx->d_mxbar, ctx->rank == 0 ? 1 : 0, (const unsigned char *)ctx->d_base_mask, ctx->d_y);
    };
    if (group < 0) range(0, ctx->n_nodes);
    else for (const std::pair<int, int> &r : ctx->pipe_nodes[group]) range(r.first, r.second);
    HIPCHK(hipGetLastError());
    return ADMM_OK;
}

// both triangular sweeps: d_y (rhs, destroyed) -> d_xcur
// part (pipeline groups): 0 = everything (default); otherwise ONE piece on stream `pst`: 1 = forward sweep of group `pg`'s subtrees,
// 2 = backward sweep of group pg, 3 = the top (forward, roots, backward)
int launch_solve(admm_hip_ctx *ctx, hipEvent_t mid, hipEvent_t ex0 = nullptr, hipEvent_t ex1 = nullptr, int part = 0, int pg = 0, hipStream_t pst = nullptr) {
    using namespace admm_dev;
    if (ctx->dense) {   // small system: one kernel, x = A_s^-1 b
        if (mid) HIPCHK(hipEventRecord(mid, ctx->stream));
        hipLaunchKernelGGL(dense_solve_kernel, dim3((ctx->n_nodes + 3) / 4), dim3(256), 0, ctx->stream, ctx->n_nodes, (const double *)ctx->d_ainv, (const double *)ctx->d_y, ctx->d_xcur);
        HIPCHK(hipGetLastError());
        return ADMM_OK;
    }
    const FactorDev F = factor_dev(ctx);
    auto forward = [&](const std::vector<LevelDev> &levels, hipStream_t st) {
        for (const LevelDev &L : levels) {
            if (L.n_small) {
                                if (F.cg4) hipLaunchKernelGGL((solve_fwd_small_kernel<true>), dim3((L.n_small + ADMM_FWD_SMALL_WAVES - 1) / ADMM_FWD_SMALL_WAVES), dim3(64 * ADMM_FWD_SMALL_WAVES), 0, st, L.n_small, L.d_small, F, ctx->d_y, ctx->d_w, ctx->d_c);
                else hipLaunchKernelGGL((solve_fwd_small_kernel<false>), dim3((L.n_small + ADMM_FWD_SMALL_WAVES - 1) / ADMM_FWD_SMALL_WAVES), dim3(64 * ADMM_FWD_SMALL_WAVES), 0, st, L.n_small, L.d_small, F, ctx->d_y, ctx->d_w, ctx->d_c);
            }
            if (L.n_big) {
#define ADMM_FWD_BIG(CG, NW) hipLaunchKernelGGL((solve_fwd_big_kernel<CG, NW>), dim3(L.n_big), dim3(64 * NW), 0, st, L.d_big, F, ctx->d_y, ctx->d_w, ctx->d_c)
                if (F.cg4) { if (L.big_nw == 4) ADMM_FWD_BIG(true, 4); else if (L.big_nw == 8) ADMM_FWD_BIG(true, 8); else ADMM_FWD_BIG(true, 16); }
                else { if (L.big_nw == 4) ADMM_FWD_BIG(false, 4); else if (L.big_nw == 8) ADMM_FWD_BIG(false, 8); else ADMM_FWD_BIG(false, 16); }
#undef ADMM_FWD_BIG
            }
            for (const LevelDev::Root &R : L.roots) {      // roots: both sweeps as one product with the explicit inverse, straight into x
                const dim3 pg((R.k + ROOT_ROWS - 1) / ROOT_ROWS), pb(64 * ROOT_ROWS);
                const double *Sinv = ctx->d_panels + R.inv_off;
                double *Xr = ctx->d_xcur + 3 * (size_t)R.first;
                if (R.k <= ctx->root_fuse_k) {      // small root: every block of the product gathers t itself (one launch less)
                    if (F.cg4) hipLaunchKernelGGL((root_product_kernel<true, true>), pg, pb, 0, st, R.k, root_inv_ld(R.k), Sinv, (const double *)ctx->d_y, Xr, R.first, R.foff, F, (const double *)ctx->d_c);
                    else hipLaunchKernelGGL((root_product_kernel<true, false>), pg, pb, 0, st, R.k, root_inv_ld(R.k), Sinv, (const double *)ctx->d_y, Xr, R.first, R.foff, F, (const double *)ctx->d_c);
                    continue;
                }
                double *T = ctx->d_w + 3 * (size_t)R.first;      // the root's own slice of W is free: it has no backward launch
                if (F.cg4) hipLaunchKernelGGL((root_gather_kernel<true>), dim3((R.k + 255) / 256), dim3(256), 0, st, R.k, R.first, R.foff, F, (const double *)ctx->d_y, (const double *)ctx->d_c, T);
                else hipLaunchKernelGGL((root_gather_kernel<false>), dim3((R.k + 255) / 256), dim3(256), 0, st, R.k, R.first, R.foff, F, (const double *)ctx->d_y, (const double *)ctx->d_c, T);
                hipLaunchKernelGGL((root_product_kernel<false, false>), pg, pb, 0, st, R.k, root_inv_ld(R.k), Sinv, (const double *)T, Xr, 0, (int64_t)0, F, (const double *)nullptr);
            }
        }
    };
    bool bad_pair = false;
    auto backward = [&](const std::vector<LevelDev> &levels, hipStream_t st) {
        for (int l = (int)levels.size() - 1; l >= 0; --l) {
            const LevelDev &L = levels[l];
            if (!L.n_bwd) continue;
            // the level's work items were cut for bwd_nw * bwd_cw columns per block (upload_factor): the kernel must be THAT pair
#define ADMM_BWD(CW, NWB) if (L.bwd_cw == CW && L.bwd_nw == NWB) { hipLaunchKernelGGL((solve_bwd_kernel<CW, NWB>), dim3(L.n_bwd), dim3(64 * NWB), 0, st, L.d_bwd, F, ctx->d_w, ctx->d_xcur); continue; }
            ADMM_BWD(4, 2) ADMM_BWD(4, 4) ADMM_BWD(4, 8) ADMM_BWD(4, 16)
            ADMM_BWD(2, 4) ADMM_BWD(2, 8) ADMM_BWD(2, 16)
            ADMM_BWD(1, 4) ADMM_BWD(1, 8) ADMM_BWD(1, 16)
#undef ADMM_BWD
            bad_pair = true;
        }
    };
    const int n_side = (int)ctx->levels_side.size();
    if (part) {
        const std::vector<LevelDev> &mine = pg == 0 ? ctx->levels : ctx->levels_side[pg - 1];
        if (part == 1) forward(mine, pst);
        else if (part == 2) backward(mine, pst);
        else { forward(ctx->levels_gtop, pst); backward(ctx->levels_gtop, pst); }
        HIPCHK(hipGetLastError());
        if (bad_pair) return fail(ctx, ADMM_ERR_STATE, "backward sweep: no kernel for a level's (columns per wave, waves per block) pair");
        return ADMM_OK;
    }
    if (ctx->pipe > 1) {      // pipeline layout launched serially: group after group on the context's stream, then the top
        forward(ctx->levels, ctx->stream);
        for (int g = 0; g < n_side; ++g) forward(ctx->levels_side[g], ctx->stream);
        forward(ctx->levels_gtop, ctx->stream);
        if (mid) HIPCHK(hipEventRecord(mid, ctx->stream));
        backward(ctx->levels_gtop, ctx->stream);
        backward(ctx->levels, ctx->stream);
        for (int g = 0; g < n_side; ++g) backward(ctx->levels_side[g], ctx->stream);
        HIPCHK(hipGetLastError());
        if (bad_pair) return fail(ctx, ADMM_ERR_STATE, "backward sweep: no kernel for a level's (columns per wave, waves per block) pair");
        return ADMM_OK;
    }
    // concurrent groups: the side streams start when the right-hand side is there and hand back before the top
    auto fork = [&]() -> int {
        if (!n_side) return ADMM_OK;
        HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int g = 0; g < n_side; ++g) HIPCHK(hipStreamWaitEvent(ctx->side_streams[g], ctx->ev_fork, 0));
        return ADMM_OK;
    };
    auto join = [&]() -> int {
        for (int g = 0; g < n_side; ++g) { HIPCHK(hipEventRecord(ctx->ev_join[g], ctx->side_streams[g])); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[g], 0)); }
        return ADMM_OK;
    };
    TRY(fork());
    for (int g = 0; g < n_side; ++g) forward(ctx->levels_side[g], ctx->side_streams[g]);
    forward(ctx->levels, ctx->stream);
    TRY(join());
    if (n_side) forward(ctx->levels_gtop, ctx->stream);
    if (!ctx->levels_top.empty()) {
        // subtree sharding: own subtrees are done; ONE small all-reduce carries the top nodes' partial right-hand sides and the
        // subtree roots' contributions to every rank, then everybody runs the (replicated) top of the tree
        const int n = ctx->n_comm_top + ctx->n_comm_slots;
        if (ex0) HIPCHK(hipEventRecord(ex0, ctx->stream));
        if (n > 0) {
            hipLaunchKernelGGL(shard_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_comm_top, (const int *)ctx->d_comm_top, ctx->n_comm_slots,
                               (const int *)ctx->d_comm_slots, (const unsigned char *)ctx->d_comm_mine, (const double *)ctx->d_y, (const double *)ctx->d_c, ctx->d_comm_buf);
            TRY(do_allreduce(ctx, ctx->d_comm_buf, 3 * (int64_t)n));
            hipLaunchKernelGGL(shard_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_comm_top, (const int *)ctx->d_comm_top, ctx->n_comm_slots,
                               (const int *)ctx->d_comm_slots, (const double *)ctx->d_comm_buf, ctx->d_y, ctx->d_c);
        }
        if (ex1) HIPCHK(hipEventRecord(ex1, ctx->stream));
        forward(ctx->levels_top, ctx->stream);
    }
    if (mid) HIPCHK(hipEventRecord(mid, ctx->stream));
    if (!ctx->levels_top.empty()) backward(ctx->levels_top, ctx->stream);
    if (n_side) backward(ctx->levels_gtop, ctx->stream);
    TRY(fork());
    for (int g = 0; g < n_side; ++g) backward(ctx->levels_side[g], ctx->side_streams[g]);
    backward(ctx->levels, ctx->stream);
    TRY(join());
    HIPCHK(hipGetLastError());
    if (bad_pair) return fail(ctx, ADMM_ERR_STATE, "backward sweep: no kernel for a level's (columns per wave, waves per block) pair");
    return ADMM_OK;
}

// ---- pipelined groups (ADMM_HIP_PIPE=G): one frame's ADMM loop ------------------------------------------------------------
// Streams: group 0 and the top on the context's stream M, group g > 0 on side stream g - 1.  Per ADMM iteration k
//   S_g : wait top(k-1) | bwd_g(k-1) | local step_g(k) [after local step_{g-1}(k) when chained] | rhs_g(k) | fwd_g(k) | done_g
//   M   : ... group 0's chain ... | wait done_g (g > 0) | rhs_top(k) | fwd_top, roots, bwd_top (k) | top(k)
// and after the last iteration every group's backward sweep.  Same kernels on the same data as the serial launch of this layout
// (launch_local / launch_rhs / launch_solve group after group): bitwise the same x.  With pipe_graph the three shapes of an
// iteration (first: no backward sweeps yet; middle; closing sweeps) are captured once as multi-stream graphs: one graph launch
// per iteration instead of ~25 launches + event operations per group.
static int pipe_piece(admm_hip_ctx *ctx, int shape) {
    const int G = ctx->pipe;
    auto S = [&](int g) { return g == 0 ? ctx->stream : ctx->side_streams[g - 1]; };
    auto L = [&](int g) { return ctx->pipe_local_streams.empty() ? S(g) : ctx->pipe_local_streams[g]; };
    HIPCHK(hipEventRecord(ctx->pipe_ev_top, ctx->stream));                       // what came before on M: prologue / the previous top
    for (int g = 1; g < G; ++g) HIPCHK(hipStreamWaitEvent(S(g), ctx->pipe_ev_top, 0));
    for (int g = 0; g < G; ++g) {
        if (shape >= 1) TRY(launch_solve(ctx, nullptr, nullptr, nullptr, 2, g, S(g)));      // bwd_g of the previous iteration
        if (shape == 2) continue;
        hipStream_t ls = L(g);
        if (ls != S(g)) { HIPCHK(hipEventRecord(ctx->pipe_ev_sw[g], S(g))); HIPCHK(hipStreamWaitEvent(ls, ctx->pipe_ev_sw[g], 0)); }
        if (ctx->pipe_chain && g > 0) HIPCHK(hipStreamWaitEvent(ls, ctx->pipe_ev_tet[g - 1], 0));
        TRY(launch_local(ctx, -1, g, ls));
        if (ctx->pipe_chain || ls != S(g)) HIPCHK(hipEventRecord(ctx->pipe_ev_tet[g], ls));
        if (ls != S(g)) HIPCHK(hipStreamWaitEvent(S(g), ctx->pipe_ev_tet[g], 0));
        TRY(launch_rhs(ctx, g, S(g)));
        TRY(launch_solve(ctx, nullptr, nullptr, nullptr, 1, g, S(g)));
    }
    for (int g = 1; g < G; ++g) { HIPCHK(hipEventRecord(ctx->pipe_ev_fwd[g], S(g))); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->pipe_ev_fwd[g], 0)); }
    if (shape == 2) return ADMM_OK;
    TRY(launch_rhs(ctx, G, ctx->stream));
    TRY(launch_solve(ctx, nullptr, nullptr, nullptr, 3, 0, ctx->stream));
    return ADMM_OK;
}
int pipe_frame(admm_hip_ctx *ctx, int admm_iters) {
    const int G = ctx->pipe;
    if ((int)ctx->pipe_ev_fwd.size() < G) {
        for (int g = 0; g < G; ++g) {
            hipEvent_t a, b, c;
            HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c, hipEventDisableTiming));
            ctx->pipe_ev_fwd.push_back(a); ctx->pipe_ev_tet.push_back(b); ctx->pipe_ev_sw.push_back(c);
        }
        HIPCHK(hipEventCreateWithFlags(&ctx->pipe_ev_top, hipEventDisableTiming));
        if (ctx->pipe_cu_mask > 0) {      // the local step on streams that leave `pipe_cu_mask` CUs of every XCD to the sweeps
            // CU mask bit i = CU i; CUs are numbered XCD-interleaved on this part (CU i -> XCD i mod 8): drop the highest-numbered ones
            const int ncu = 256, drop = std::min(ncu - 8, 8 * ctx->pipe_cu_mask);
            std::vector<uint32_t> mask(ncu / 32, 0xffffffffu);
            for (int i = ncu - drop; i < ncu; ++i) mask[i / 32] &= ~(1u << (i % 32));
            for (int g = 0; g < G; ++g) {
                hipStream_t st = nullptr;
                if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); ctx->pipe_local_streams.clear(); fprintf(stderr, "admm_hip: CU-masked streams unavailable\n"); break; }
                ctx->pipe_local_streams.push_back(st);
            }
        }
    }
    // (a CU-masked stream inside a stream capture crashes the HIP runtime of this image: masked runs launch eagerly)
    const bool want_graph = ctx->pipe_graph && ctx->graph_enabled && ctx->pipe_local_streams.empty();
    if (want_graph && !ctx->pipe_exec[0]) {
        for (int shape = 0; shape < 3; ++shape) {
            const hipError_t be = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
            const int rc = be == hipSuccess ? pipe_piece(ctx, shape) : ADMM_ERR_HIP;
            hipGraph_t g = nullptr;
            const hipError_t ce = be == hipSuccess ? hipStreamEndCapture(ctx->stream, &g) : be;
            if (rc || ce != hipSuccess || !g || hipGraphInstantiate(&ctx->pipe_exec[shape], g, nullptr, nullptr, 0) != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                for (int q = 0; q < 3; ++q) { if (ctx->pipe_exec[q]) (void)hipGraphExecDestroy(ctx->pipe_exec[q]); ctx->pipe_exec[q] = nullptr; if (ctx->pipe_graph_h[q]) (void)hipGraphDestroy(ctx->pipe_graph_h[q]); ctx->pipe_graph_h[q] = nullptr; }
                ctx->pipe_graph = false;
                fprintf(stderr, "admm_hip: pipeline graph capture unavailable (shape %d), launching eagerly\n", shape);
                break;
            }
            ctx->pipe_graph_h[shape] = g;
        }
    }
    const bool graph = want_graph && ctx->pipe_exec[0] && ctx->pipe_exec[1] && ctx->pipe_exec[2];
    for (int it = 0; it < admm_iters; ++it) {
        const int shape = it == 0 ? 0 : 1;
        if (graph) HIPCHK(hipGraphLaunch(ctx->pipe_exec[shape], ctx->stream)); else TRY(pipe_piece(ctx, shape));
    }
    if (graph) HIPCHK(hipGraphLaunch(ctx->pipe_exec[2], ctx->stream)); else TRY(pipe_piece(ctx, 2));
    return ADMM_OK;
}

// subtree sharding: after the solve a rank holds x only on its own subtrees and the top; rebuild the full vector
// (once per frame, and for the solve-only entry point)
int shard_sync_x(admm_hip_ctx *ctx) {
    if (ctx->levels_top.empty()) return ADMM_OK;
    const int n3 = 3 * ctx->n_nodes;
    hipLaunchKernelGGL(admm_dev::shard_mask_nodes_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const unsigned char *)ctx->d_keep_mask, ctx->d_xcur);
    TRY(do_allreduce(ctx, ctx->d_xcur, (int64_t)n3));
    HIPCHK(hipGetLastError());
    return ADMM_OK;
}


// ---- residual tracking (opt-in): buffers are created on first use -------------------------------------
int ensure_residual_buffers(admm_hip_ctx *ctx, int iters) {
    if (ctx->res_ready && iters <= ctx->res_cap) return ADMM_OK;
    HIPCHK(hipSetDevice(ctx->device_id));
    if (!ctx->res_ready) {
        int64_t slots = 0; int maxn = 0;
        for (Batch &b : ctx->batches) {
            if (b.kind == ADMM_KIND_GENERIC) {      // user rows: u, z live on the host; the dual residual goes through the same slots with w^2 coefficients
                std::vector<double> coef(b.g_sval.size());
                for (size_t i = 0; i < coef.size(); ++i) { const double w = b.g_roww[b.g_srow_b[i]]; coef[i] = b.g_sval[i] * (w * w); }
                TRY(upload(ctx, &b.d_g_scoef_res, coef));
                slots += b.g_lslots;
                continue;
            }
            const int rows = ADMM_KIND_ROWS[b.kind], nl = std::max(b.n_local, 1);
            // (ADMM_HIP_RES_UNFUSED=1: the separate passes, for comparison; not for pre-reduced tet batches, which have no per-corner slots)
            b.res_fused = ((b.kind >= ADMM_KIND_TET_LINEAR && b.kind <= ADMM_KIND_TET_STVK) || b.kind == ADMM_KIND_ANCHOR) && (b.prered || getenv("ADMM_HIP_RES_UNFUSED") == nullptr);
            if (b.res_fused) {      // the tet kernels produce their residuals themselves: no snapshots, one partial per 64-tet block
                const int nblk = std::max(batch_blocks(b), 1);
                TRY(dalloc(ctx, &b.d_res_partial, (size_t)nblk));
                HIPCHK(hipMemset(b.d_res_partial, 0, sizeof(double) * (size_t)nblk));
            } else { TRY(dalloc(ctx, &b.d_u_prev, (size_t)rows * nl)); TRY(dalloc(ctx, &b.d_z_prev, (size_t)rows * nl)); }
            TRY(upload(ctx, &b.d_G, b.G));
            slots += (int64_t)b.n_local * ADMM_KIND_NODES[b.kind]; maxn = std::max(maxn, b.n_local);
        }
        if (ctx->n_gen_rows) {
            ctx->h_gen_u_prev.assign((size_t)ctx->n_gen_rows, 0.0); ctx->h_gen_z_prev.assign((size_t)ctx->n_gen_rows, 0.0);
            TRY(dalloc(ctx, &ctx->d_gen_q2, (size_t)ctx->n_gen_rows)); TRY(dalloc(ctx, &ctx->d_gen_r2, 1));
        }
        slots = std::max<int64_t>(slots, ctx->n_fslots);      // same layout as the RHS slots
        TRY(dalloc(ctx, &ctx->d_res_slots, 3 * (size_t)std::max<int64_t>(slots, 1)));
        HIPCHK(hipMemset(ctx->d_res_slots, 0, sizeof(double) * 3 * (size_t)std::max<int64_t>(slots, 1)));
        TRY(dalloc(ctx, &ctx->d_res_s, 3 * (size_t)ctx->n_nodes));
        ctx->res_partial_n = std::max((maxn + admm_dev::RES_BLOCK - 1) / admm_dev::RES_BLOCK, (3 * ctx->n_nodes + admm_dev::RES_BLOCK - 1) / admm_dev::RES_BLOCK);
        TRY(dalloc(ctx, &ctx->d_res_partial, (size_t)std::max(ctx->res_partial_n, 1)));
        ctx->res_ready = true; ctx->res_cap = 0;
    }
    if (iters > ctx->res_cap) { ctx->res_cap = std::max(iters, 64); TRY(dalloc(ctx, &ctx->d_res, 2 * (size_t)ctx->res_cap)); }
    return ADMM_OK;
}
// before the local step: keep u and z of the previous iteration
int residual_snapshot(admm_hip_ctx *ctx, bool first_iteration) {
    if (ctx->n_gen_rows) {      // user rows (host): at a frame's first iteration h_gen_z already holds D * m_x (admm_hip_step)
        std::memcpy(ctx->h_gen_u_prev.data(), ctx->h_gen_u, sizeof(double) * (size_t)ctx->n_gen_rows);
        std::memcpy(ctx->h_gen_z_prev.data(), ctx->h_gen_z, sizeof(double) * (size_t)ctx->n_gen_rows);
    }
    for (Batch &b : ctx->batches) {
        if (!b.n_local || b.kind == ADMM_KIND_GENERIC) continue;
        const int rows = ADMM_KIND_ROWS[b.kind];
        const size_t bytes = sizeof(double) * (size_t)rows * b.n_local;
        if (b.res_fused) {     // z_prev is the kernel's own previous output; only the frame's warm start has to be put there
            if (first_iteration)
                hipLaunchKernelGGL(admm_dev::residual_dx_kernel, dim3((b.n_local + admm_dev::RES_BLOCK - 1) / admm_dev::RES_BLOCK), dim3(admm_dev::RES_BLOCK), 0, ctx->stream,
                                   b.n_local, ADMM_KIND_NODES[b.kind], rows / 3, idx_stride(b.kind), b.d_idx, b.d_G, ctx->d_x, b.d_z);
            continue;
        }
        const int64_t cnt = (int64_t)rows * b.n_local;      // one launch instead of two device-to-device copies (each a ~10 us operation whatever its size)
        hipLaunchKernelGGL(admm_dev::residual_snapshot_kernel, dim3((unsigned)((cnt + admm_dev::RES_BLOCK - 1) / admm_dev::RES_BLOCK)), dim3(admm_dev::RES_BLOCK), 0, ctx->stream,
                           cnt, (const double *)b.d_u, (const double *)b.d_z, b.d_u_prev, b.d_z_prev, first_iteration ? 0 : 1);
        if (first_iteration)   // the reference warm-starts curr_z = D * m_x before the loop (System.cpp:43)
            hipLaunchKernelGGL(admm_dev::residual_dx_kernel, dim3((b.n_local + admm_dev::RES_BLOCK - 1) / admm_dev::RES_BLOCK), dim3(admm_dev::RES_BLOCK), 0, ctx->stream,
                               b.n_local, ADMM_KIND_NODES[b.kind], rows / 3, idx_stride(b.kind), b.d_idx, b.d_G, ctx->d_x, b.d_z_prev);
        (void)bytes;
    }
    return ADMM_OK;
}
// after the local step: d_res[2 it] = |r|^2 (this rank's elements), d_res[2 it + 1] = |s|^2
int launch_residuals(admm_hip_ctx *ctx, int it) {
    using namespace admm_dev;
    const int n3 = 3 * ctx->n_nodes;
    double *r2 = ctx->d_res + 2 * (size_t)it, *s2 = r2 + 1;
    bool first = true;
    if (ctx->n_gen_rows) {      // user rows: |r|^2 of this rank's rows on the host, z - z_prev to the device for the dual residual
        double r2h = 0.0;
        for (const Batch &b : ctx->batches) if (b.kind == ADMM_KIND_GENERIC)
            for (int el = 0; el < b.n_local; ++el) for (int64_t r = b.g_elem_row[b.local[el]]; r < b.g_elem_row[b.local[el] + 1]; ++r) {
                const double d = ctx->h_gen_u[b.g_row0 + r] - ctx->h_gen_u_prev[b.g_row0 + r], w = b.g_roww[r];
                r2h += (w * w) * (d * d);
            }
        HIPCHK(hipStreamSynchronize(ctx->stream));      // h_gen_q may still be the source of generic_finish's asynchronous upload
        for (int64_t r = 0; r < ctx->n_gen_rows; ++r) ctx->h_gen_q[r] = ctx->h_gen_z[r] - ctx->h_gen_z_prev[r];
        HIPCHK(hipMemcpyAsync(ctx->d_gen_q2, ctx->h_gen_q, sizeof(double) * (size_t)ctx->n_gen_rows, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_gen_r2, &r2h, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));      // (r2h is a stack variable)
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(RES_BLOCK), 0, ctx->stream, 1, (const double *)ctx->d_gen_r2, r2, 0);
        first = false;
        for (const Batch &b : ctx->batches) if (b.kind == ADMM_KIND_GENERIC && b.g_lslots)
            hipLaunchKernelGGL(generic_rhs_kernel, dim3((3 * b.g_lslots + 255) / 256), dim3(256), 0, ctx->stream, 3 * b.g_lslots, (const int *)b.d_g_sptr, (const int *)b.d_g_srow,
                               (const double *)b.d_g_scoef_res, (const int *)b.d_g_sdst, (const double *)ctx->d_gen_q2, ctx->d_res_slots);
    }
    for (Batch &b : ctx->batches) {
        if (!b.n_local || b.kind == ADMM_KIND_GENERIC) continue;
        if (b.res_fused) {     // |r|^2 partials and the s slots were written by the projection kernel
            const int nblk = batch_blocks(b);
            hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(RES_BLOCK), 0, ctx->stream, nblk, (const double *)b.d_res_partial, r2, first ? 0 : 1);
            first = false;
            continue;
        }
        const int nb = (b.n_local + RES_BLOCK - 1) / RES_BLOCK, rows = ADMM_KIND_ROWS[b.kind];
        hipLaunchKernelGGL(residual_primal_kernel, dim3(nb), dim3(RES_BLOCK), 0, ctx->stream, b.n_local, rows, b.d_u, b.d_u_prev, b.d_w2, ctx->d_res_partial);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(RES_BLOCK), 0, ctx->stream, nb, ctx->d_res_partial, r2, first ? 0 : 1);
        hipLaunchKernelGGL(residual_dual_kernel, dim3(nb), dim3(RES_BLOCK), 0, ctx->stream, b.n_local, ADMM_KIND_NODES[b.kind], rows / 3, idx_stride(b.kind),
                           b.d_z, b.d_z_prev, b.d_w2, b.d_G, b.d_dst, ctx->d_res_slots);
        first = false;
    }
    if (first) HIPCHK(hipMemsetAsync(r2, 0, sizeof(double), ctx->stream));
    const int nb = (n3 + RES_BLOCK - 1) / RES_BLOCK;
    static_assert(RES_BLOCK == 256, "the gather's blocks are the norm's partials");
    if (ctx->world == 1) {      // one rank: the gather of s leaves its blocks' sums of squares behind, no norm pass
        hipLaunchKernelGGL(rhs_gather_kernel<true>, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, 0, ctx->n_nodes, ctx->d_inc_ptr, ctx->slot_stride, ctx->d_res_slots, ctx->d_mxbar, 0, (const unsigned char *)nullptr, ctx->d_res_s, ctx->d_res_partial);
    } else {                    // s is a sum over all ranks' elements (all-reduced before its norm); r^2 is additive
        hipLaunchKernelGGL(rhs_gather_kernel<false>, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, 0, ctx->n_nodes, ctx->d_inc_ptr, ctx->slot_stride, ctx->d_res_slots, ctx->d_mxbar, 0, (const unsigned char *)nullptr, ctx->d_res_s);
        if (ctx->world > 1) { TRY(do_allreduce(ctx, ctx->d_res_s, (int64_t)n3)); TRY(do_allreduce(ctx, r2, 1)); }
        hipLaunchKernelGGL(norm2_partial_kernel, dim3(nb), dim3(RES_BLOCK), 0, ctx->stream, n3, ctx->d_res_s, ctx->d_res_partial);
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(RES_BLOCK), 0, ctx->stream, nb, ctx->d_res_partial, s2, 0);
    HIPCHK(hipGetLastError());
    return ADMM_OK;
}

int require_device(admm_hip_ctx *ctx) {
    if (!ctx) return ADMM_ERR_ARG;
    if (ctx->device_id < 0) return fail(ctx, ADMM_ERR_HIP, "host-only context: no GPU path available (the product has no CPU fallback)");
    if (!ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "admm_hip_finalize has not been called");
    return ADMM_OK;
}

} // namespace

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int admm_hip_create(admm_hip_ctx **out, int device_id) {
    if (!out) return ADMM_ERR_ARG;
    *out = nullptr;
    admm_hip_ctx *ctx = new admm_hip_ctx();
    ctx->device_id = device_id;
    if (device_id >= 0) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= device_id) {
            fprintf(stderr, "admm_hip: no HIP device %d (%s, %d devices) -- refusing to run without a GPU\n", device_id, hipGetErrorString(e), count);
            delete ctx;
            return ADMM_ERR_HIP;
        }
        if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return ADMM_ERR_HIP; }
        // probe knob (tools/probe/cu_mask_probe.py): the context's own stream restricted to a CU set, ADMM_HIP_STREAM_CUMASK = 64 hex digits
        // (256 bits, most significant first) -- how the phases scale with the CUs they may use
        const char *cm = getenv("ADMM_HIP_STREAM_CUMASK");
        bool made = false;
        if (cm && std::strlen(cm) == 64) {
            uint32_t mask[8];
            for (int w = 0; w < 8; ++w) { char buf[9]; std::memcpy(buf, cm + 8 * (7 - w), 8); buf[8] = 0; mask[w] = (uint32_t)std::strtoul(buf, nullptr, 16); }
            made = hipExtStreamCreateWithCUMask(&ctx->stream, 8, mask) == hipSuccess;
            if (!made) { (void)hipGetLastError(); fprintf(stderr, "admm_hip: CU-masked stream unavailable\n"); }
        }
        if (!made && hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return ADMM_ERR_HIP; }
        ctx->own_stream = true;
    }
    ctx->info.device_id = device_id; ctx->info.world = 1;
    const char *ls = getenv("ADMM_HIP_LEAF");
    if (ls && atoi(ls) > 0) ctx->leaf_size = atoi(ls);
    if (const char *g = getenv("ADMM_HIP_GRAPH")) { ctx->graph_enabled = atoi(g) != 0; ctx->graph_forced = ctx->graph_enabled; }
#if defined(ADMM_TET_PROFILE) || defined(ADMM_TET_TIMELINE) || defined(ADMM_SWEEP_PROFILE)
    ctx->graph_enabled = false;      // diagnostic builds: eager launches only (their per-launch symbol updates are not capturable)
#endif
    if (const char *g = getenv("ADMM_HIP_GRAPH_COMM")) ctx->graph_comm = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_BWD_NW")) { const int v = atoi(g); if (v == 4 || v == 8 || v == 16) ctx->bwd_nw = v; }
    if (const char *g = getenv("ADMM_HIP_FWD_SMALL_K")) ctx->fwd_small_k = atoi(g);
    if (const char *g = getenv("ADMM_HIP_BWD_SMALL_K")) ctx->bwd_small_k = atoi(g);
    if (const char *g = getenv("ADMM_HIP_TREE_SEARCH")) ctx->tree_search = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_STATE_DIRECT")) ctx->state_direct_max_nodes = atoi(g);      // systems up to that many nodes: upload_state / download_state without DMA (0: never)
    if (const char *g = getenv("ADMM_HIP_LOCAL_MULTI")) ctx->local_multi = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_LOCAL_STREAMS")) ctx->local_streams_max = std::max(1, atoi(g));      // 1: every batch on the context's stream, one after the other
    if (const char *g = getenv("ADMM_HIP_LOCAL_STREAMS_MIN")) ctx->local_streams_min_elems = atoi(g);
    if (const char *g = getenv("ADMM_HIP_ROOT_FUSE_K")) ctx->root_fuse_k = std::min(atoi(g), (int)admm_dev::ROOT_KCHUNK);
    if (const char *g = getenv("ADMM_HIP_FRAME_GRAPH")) ctx->frame_graph_on = atoi(g) != 0;      // 0: one graph launch per ADMM iteration instead of one per frame
    if (const char *g = getenv("ADMM_HIP_BWD_SMALL_NW")) { const int v = atoi(g); if (v == 2 || v == 4 || v == 8 || v == 16) ctx->bwd_small_nw = v; }
    if (const char *g = getenv("ADMM_HIP_XCD")) ctx->xcd_min_supernodes = atoi(g);
    if (const char *g = getenv("ADMM_HIP_TET_ORDER")) ctx->tet_order = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_TET_ORDER_MIN")) ctx->tet_order_min_blocks = atoi(g);
    if (const char *g = getenv("ADMM_HIP_FUSE_ANCHORS")) ctx->fuse_anchor_tail = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_FACTOR")) ctx->device_factor = std::string(g) != "host";
    if (const char *g = getenv("ADMM_HIP_BWD_CW2_MIN")) ctx->bwd_cw2_min_cols = atoi(g);
    if (const char *g = getenv("ADMM_HIP_BWD_CW2_MAX")) ctx->bwd_cw2_max_cols = atoi(g);
    if (const char *g = getenv("ADMM_HIP_GROUPS")) { const int v = atoi(g); if (v >= 1 && v <= 8) ctx->groups = v; }
    if (const char *g = getenv("ADMM_HIP_PIPE")) { const int v = atoi(g); if (v >= 2 && v <= 8) { ctx->pipe = v; ctx->groups = v; } }
    if (const char *g = getenv("ADMM_HIP_PRERED")) ctx->tet_prered = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_STATE_ZEROCOPY")) ctx->state_zero_copy = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_TET_LDS_PAD")) ctx->tet_lds_pad = std::max(0, atoi(g));
    if (const char *g = getenv("ADMM_HIP_TPB")) { const int v = atoi(g); if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) ctx->tet_tpb = v; }
    if (const char *g = getenv("ADMM_HIP_KEEP_Z")) ctx->keep_z_user = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_PIPE_CHAIN")) ctx->pipe_chain = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_PIPE_GRAPH")) ctx->pipe_graph = atoi(g) != 0;
    if (const char *g = getenv("ADMM_HIP_PIPE_CUMASK")) ctx->pipe_cu_mask = atoi(g);
    if (const char *g = getenv("ADMM_HIP_BWD_NW_MIN_COLS")) ctx->bwd_nw_min_cols = atoi(g);
    if (const char *g = getenv("ADMM_HIP_FWD_NW16_TILES")) ctx->fwd_nw16_max_tiles = atoi(g);
    if (const char *g = getenv("ADMM_HIP_FWD_NW4")) ctx->fwd_nw4_kmax = atoi(g);
    if (const char *g = getenv("ADMM_HIP_FWD_NW8")) ctx->fwd_nw8_kmax = atoi(g);
    if (const char *g = getenv("ADMM_HIP_DENSE_MAX")) ctx->dense_max = atoi(g);
    if (const char *g = getenv("ADMM_HIP_ROOT_INVERSE")) ctx->root_inverse = atoi(g) != 0;
    *out = ctx;
    return ADMM_OK;
}

void admm_hip_destroy(admm_hip_ctx *ctx) {
    if (!ctx) return;
    if (ctx->device_id >= 0) {
        (void)hipSetDevice(ctx->device_id);
        (void)hipDeviceSynchronize();
        free_device(ctx);
        for (hipEvent_t e : ctx->evpool) (void)hipEventDestroy(e);
        if (ctx->rccl_comm && ctx->rccl_owned) { RcclApi *R = rccl_api(nullptr); if (R) (void)R->CommDestroy(ctx->rccl_comm); }
        for (double *h : {ctx->h_gen_dx, ctx->h_gen_u, ctx->h_gen_z, ctx->h_gen_q}) if (h) (void)hipHostFree(h);
        if (ctx->gen_ev) (void)hipEventDestroy(ctx->gen_ev);
        if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
        if (ctx->d_small) (void)hipFree(ctx->d_small);
        if (ctx->h_state) (void)hipHostFree(ctx->h_state);
        if (ctx->state_in_ev) (void)hipEventDestroy(ctx->state_in_ev);
        for (hipStream_t st : ctx->pipe_local_streams) (void)hipStreamDestroy(st);
        for (hipEvent_t e : ctx->pipe_ev_fwd) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->pipe_ev_tet) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->pipe_ev_sw) (void)hipEventDestroy(e);
        if (ctx->pipe_ev_top) (void)hipEventDestroy(ctx->pipe_ev_top);
        for (hipStream_t st : ctx->side_streams) (void)hipStreamDestroy(st);
        for (hipStream_t st : ctx->local_side) (void)hipStreamDestroy(st);
        for (hipEvent_t e : ctx->local_join) (void)hipEventDestroy(e);
        if (ctx->local_fork) (void)hipEventDestroy(ctx->local_fork);
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        for (hipEvent_t e : ctx->ev_join) (void)hipEventDestroy(e);
        if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char *admm_hip_last_error(const admm_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int admm_hip_set_stream(admm_hip_ctx *ctx, void *s) {
    if (!ctx || ctx->device_id < 0) return ADMM_ERR_ARG;
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    ctx->own_stream = false;
    ctx->stream = (hipStream_t)s;
    if (!s) { HIPCHK(hipStreamCreate(&ctx->stream)); ctx->own_stream = true; }
    return ADMM_OK;
}

int admm_hip_set_timestep(admm_hip_ctx *ctx, double dt) {
    if (!ctx) return ADMM_ERR_ARG;
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "timestep cannot change after finalize (System.hpp:40)");
    ctx->dt = dt;
    return ADMM_OK;
}

int admm_hip_add_nodes(admm_hip_ctx *ctx, int n_nodes, const double *x, const double *m, int *total) {
    if (!ctx || n_nodes < 0 || (n_nodes && (!x || !m))) return ADMM_ERR_ARG;
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "nodes cannot be added after finalize (System.hpp:60-62)");
    ctx->x.insert(ctx->x.end(), x, x + 3 * (size_t)n_nodes);
    ctx->m3.insert(ctx->m3.end(), m, m + 3 * (size_t)n_nodes);
    ctx->v.resize(ctx->x.size(), 0.0);
    ctx->n_nodes += n_nodes;
    if (total) *total = ctx->n_nodes;
    return ADMM_OK;
}

int admm_hip_add_batch(admm_hip_ctx *ctx, int kind, int n_elems, const int32_t *idx, const double *params, const double *targets, int *batch) {
    if (!ctx || n_elems < 0 || (n_elems && (!idx || !params))) return ADMM_ERR_ARG;
    if (kind < 0 || kind >= ADMM_KIND_COUNT) return fail(ctx, ADMM_ERR_UNSUPPORTED, "force kind %d has no accelerated kernel", kind);
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "forces cannot be added after finalize");
    Batch b; b.kind = kind; b.n_total = n_elems;
    b.idx.assign(idx, idx + (size_t)n_elems * ADMM_KIND_NODES[kind]);
    b.params.assign(params, params + (size_t)n_elems * ADMM_KIND_PARAMS[kind]);
    if (kind == ADMM_KIND_ANCHOR) {
        b.targets.assign((size_t)3 * n_elems, 0.0);
        b.active.assign(n_elems, 1);
        if (targets) {
            b.moving = true;
            std::copy(targets, targets + (size_t)3 * n_elems, b.targets.begin());
            for (int e = 0; e < n_elems; ++e) b.active[e] = params[2 * (size_t)e + 1] != 0.0;
        }
    }
    ctx->batches.push_back(std::move(b));
    if (batch) *batch = (int)ctx->batches.size() - 1;
    return ADMM_OK;
}

int admm_hip_add_generic_batch(admm_hip_ctx *ctx, int n_elems, const int32_t *elem_row_ptr, int64_t n_triplets, const int32_t *trip_row, const int32_t *trip_col,
                               const double *trip_val, const double *row_weight, int *batch) {
    if (!ctx || n_elems < 0 || n_triplets < 0 || !elem_row_ptr || (n_triplets && (!trip_row || !trip_col || !trip_val))) return ADMM_ERR_ARG;
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "forces cannot be added after finalize");
    Batch b; b.kind = ADMM_KIND_GENERIC; b.n_total = n_elems;
    b.g_elem_row.assign(elem_row_ptr, elem_row_ptr + n_elems + 1);
    if (b.g_elem_row[0] != 0) return fail(ctx, ADMM_ERR_ARG, "generic batch: elem_row_ptr[0] must be 0");
    for (int e = 0; e < n_elems; ++e) if (b.g_elem_row[e + 1] < b.g_elem_row[e]) return fail(ctx, ADMM_ERR_ARG, "generic batch: elem_row_ptr must not decrease");
    const int64_t rows = b.g_elem_row[n_elems];
    if (rows && !row_weight) return ADMM_ERR_ARG;
    b.g_rows = rows; b.g_row0 = ctx->n_gen_rows;
    b.g_roww.assign(row_weight, row_weight + rows);
    // triplets -> CSR: ascending (row, column), duplicates summed in the order given (Eigen's setFromTriplets sums them too)
    std::vector<int64_t> ord(n_triplets);
    std::iota(ord.begin(), ord.end(), (int64_t)0);
    for (int64_t t = 0; t < n_triplets; ++t) if (trip_row[t] < 0 || trip_row[t] >= rows || trip_col[t] < 0) return fail(ctx, ADMM_ERR_ARG, "generic batch: triplet %lld (row %d, col %d) out of range (%lld rows)", (long long)t, trip_row[t], trip_col[t], (long long)rows);
    std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t c) { return trip_row[a] != trip_row[c] ? trip_row[a] < trip_row[c] : trip_col[a] < trip_col[c]; });
    b.g_rowptr.assign(rows + 1, 0);
    for (int64_t q = 0; q < n_triplets; ++q) {
        const int64_t t = ord[q];
        if (!b.g_col.empty() && q > 0 && trip_row[ord[q - 1]] == trip_row[t] && b.g_col.back() == trip_col[t]) { b.g_val.back() += trip_val[t]; continue; }
        b.g_col.push_back(trip_col[t]); b.g_val.push_back(trip_val[t]); b.g_rowptr[trip_row[t] + 1]++;
    }
    for (int64_t r = 0; r < rows; ++r) b.g_rowptr[r + 1] += b.g_rowptr[r];
    b.g_elem_node.assign(1, 0);
    for (int e = 0; e < n_elems; ++e) {
        std::vector<int32_t> nd;
        for (int64_t p = b.g_rowptr[b.g_elem_row[e]]; p < b.g_rowptr[b.g_elem_row[e + 1]]; ++p) nd.push_back(b.g_col[p] / 3);
        std::sort(nd.begin(), nd.end()); nd.erase(std::unique(nd.begin(), nd.end()), nd.end());
        b.g_nodes.insert(b.g_nodes.end(), nd.begin(), nd.end());
        b.g_elem_node.push_back((int64_t)b.g_nodes.size());
    }
    ctx->n_gen_rows += rows;
    ctx->batches.push_back(std::move(b));
    if (batch) *batch = (int)ctx->batches.size() - 1;
    return ADMM_OK;
}
int admm_hip_set_project_hook(admm_hip_ctx *ctx, admm_hip_project_fn fn, void *user) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->project_hook = fn; ctx->project_user = user;
    return ADMM_OK;
}

int admm_hip_add_explicit(admm_hip_ctx *ctx, int type, const double *dir, int n_idx, const int32_t *idx, int *which) {
    if (!ctx || !dir || n_idx < 0 || (n_idx && !idx)) return ADMM_ERR_ARG;
    if (type != ADMM_EXPLICIT_CONST && type != ADMM_EXPLICIT_WIND) return fail(ctx, ADMM_ERR_UNSUPPORTED, "explicit force type %d", type);
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "explicit forces cannot be added after finalize");
    Explicit E; E.type = type; E.n = n_idx;
    for (int j = 0; j < 3; ++j) E.dir[j] = dir[j];
    E.idx.assign(idx, idx + (size_t)n_idx * (type == ADMM_EXPLICIT_WIND ? 3 : 1));
    for (int32_t v : E.idx) if (v < 0) return fail(ctx, ADMM_ERR_ARG, "negative node id in explicit force");
    const bool simple = type == ADMM_EXPLICIT_CONST && n_idx == 0;
    if (simple && ctx->explicit_simple && ctx->grav.n < admm_dev::MAX_GRAV) { double *g = ctx->grav.g[ctx->grav.n++]; g[0] = dir[0]; g[1] = dir[1]; g[2] = dir[2]; }
    else ctx->explicit_simple = false;
    ctx->explicits.push_back(std::move(E));
    if (which) *which = (int)ctx->explicits.size() - 1;
    return ADMM_OK;
}
int admm_hip_add_gravity(admm_hip_ctx *ctx, double gx, double gy, double gz) {
    const double d[3] = {gx, gy, gz};
    return admm_hip_add_explicit(ctx, ADMM_EXPLICIT_CONST, d, 0, nullptr, nullptr);
}
int admm_hip_set_gravity(admm_hip_ctx *ctx, int which, double gx, double gy, double gz) {
    if (!ctx || which < 0 || which >= (int)ctx->explicits.size()) return ADMM_ERR_ARG;
    double *d = ctx->explicits[which].dir;
    d[0] = gx; d[1] = gy; d[2] = gz;
    if (ctx->explicit_simple) { double *g = ctx->grav.g[which]; g[0] = gx; g[1] = gy; g[2] = gz; }
    return ADMM_OK;
}
int admm_hip_set_collision_shapes(admm_hip_ctx *ctx, int n_shapes, const int32_t *types, const double *params) {
    if (!ctx || n_shapes < 0 || (n_shapes && (!types || !params))) return ADMM_ERR_ARG;
    if (n_shapes > ADMM_MAX_SHAPES) return fail(ctx, ADMM_ERR_UNSUPPORTED, "at most %d collision shapes", ADMM_MAX_SHAPES);
    ctx->shapes.n = n_shapes;
    for (int j = 0; j < n_shapes; ++j) {
        if (types[j] < ADMM_SHAPE_FLOOR || types[j] > ADMM_SHAPE_CYLINDER) return fail(ctx, ADMM_ERR_UNSUPPORTED, "collision shape type %d", types[j]);
        ctx->shapes.type[j] = types[j];
        for (int q = 0; q < 4; ++q) ctx->shapes.par[j][q] = params[4 * (size_t)j + q];
    }
    if (ctx->finalized && ctx->device_id >= 0) {
        HIPCHK(hipSetDevice(ctx->device_id));
        HIPCHK(hipMemcpyAsync(ctx->d_shapes, &ctx->shapes, sizeof(admm_dev::ShapeTable), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return ADMM_OK;
}

int admm_hip_set_shard(admm_hip_ctx *ctx, int rank, int world) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return ADMM_ERR_ARG;
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "shard must be set before finalize");
    ctx->rank = rank; ctx->world = world;
    return ADMM_OK;
}
int admm_hip_set_shard_mode(admm_hip_ctx *ctx, int mode) {
    if (!ctx || (mode != ADMM_SHARD_CONTIGUOUS && mode != ADMM_SHARD_SUBTREE)) return ADMM_ERR_ARG;
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "shard mode must be set before finalize");
    ctx->shard_mode = mode;
    return ADMM_OK;
}
int admm_hip_local_elements(admm_hip_ctx *ctx, int batch, int32_t *ids, int capacity, int *n_local) {
    if (!ctx || !ctx->finalized || batch < 0 || batch >= (int)ctx->batches.size()) return ADMM_ERR_ARG;
    const Batch &b = ctx->batches[batch];
    if (n_local) *n_local = b.n_local;
    if (ids) { if (capacity < b.n_local) return ADMM_ERR_ARG; std::copy(b.local.begin(), b.local.end(), ids); }
    return ADMM_OK;
}
int admm_hip_debug_node_owner(admm_hip_ctx *ctx, int32_t *owner) {
    if (!ctx || !ctx->finalized || !owner) return ADMM_ERR_ARG;
    for (int i = 0; i < ctx->n_nodes; ++i) owner[i] = (ctx->shard_mode == ADMM_SHARD_SUBTREE && ctx->world > 1) ? ctx->node_owner[ctx->F.iperm[i]] : 0;
    return ADMM_OK;
}
#ifdef ADMM_TET_TIMELINE
// wave timeline of the NEXT tet launches (the buffer is overwritten by every launch: read it after the one of interest)
static unsigned long long *g_wave_t_buf; static size_t g_wave_t_n;
extern "C" int admm_hip_debug_tet_wave_times(long n_waves, unsigned long long *out) {
    if (!out) {      // arm
        hipFree(g_wave_t_buf); g_wave_t_buf = nullptr; g_wave_t_n = (size_t)n_waves;
        if (n_waves > 0 && hipMalloc(&g_wave_t_buf, 32 * g_wave_t_n) != hipSuccess) return ADMM_ERR_HIP;      // per wave: start, end, max evaluations, max iterations
        if (n_waves > 0) hipMemset(g_wave_t_buf, 0, 32 * g_wave_t_n);
        return hipMemcpyToSymbol(HIP_SYMBOL(admm_dev::g_tet_wave_t), &g_wave_t_buf, sizeof(g_wave_t_buf)) == hipSuccess ? ADMM_OK : ADMM_ERR_HIP;
    }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, g_wave_t_buf, 32 * g_wave_t_n, hipMemcpyDeviceToHost) != hipSuccess) return ADMM_ERR_HIP;
    return ADMM_OK;
}
#endif
#ifdef ADMM_SWEEP_PROFILE
// -> stamps[4 * workgroups], meta[6 * launches]; returns the number of launches (negative: error; call with NULL for the sizes)
extern "C" long admm_hip_debug_sweep_profile_read(unsigned long long *stamps, int *meta) {
    if (!stamps) return (long)g_swp_wgs;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(stamps, g_swp_base, sizeof(unsigned long long) * 4 * g_swp_wgs, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (size_t i = 0; i < g_swp_meta.size(); ++i) meta[i] = g_swp_meta[i];
    return (long)(g_swp_meta.size() / 6);
}
#endif
#ifdef ADMM_TET_PROFILE
// tools/probe/ls_predict_gpu.py only (variant build): per-tet trace of the next `cap` launches of the tet kernel (0: off)
extern "C" int admm_hip_debug_tet_trace(int cap, int n) {
    hipFree(g_trace_base); g_trace_base = nullptr; g_trace_cap = cap; g_trace_n = n; g_trace_count = 0;
    if (cap > 0 && hipMalloc(&g_trace_base, sizeof(float) * 2 * (size_t)cap * n) != hipSuccess) return ADMM_ERR_HIP;
    return ADMM_OK;
}
extern "C" int admm_hip_debug_tet_trace_read(float *out) {
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, g_trace_base, sizeof(float) * 2 * (size_t)g_trace_cap * g_trace_n, hipMemcpyDeviceToHost) != hipSuccess) return ADMM_ERR_HIP;
    g_trace_count = 0;
    return ADMM_OK;
}
// tools/tet_phase_profile.py only (variant build): read and clear the tet kernel's phase counters
extern "C" int admm_hip_debug_tet_profile(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(admm_dev::g_tet_prof), sizeof(unsigned long long) * 128) != hipSuccess) return ADMM_ERR_HIP;
    unsigned long long zero[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(admm_dev::g_tet_prof), zero, sizeof(zero)) != hipSuccess) return ADMM_ERR_HIP;
    return ADMM_OK;
}
#endif
int admm_hip_set_allreduce(admm_hip_ctx *ctx, admm_hip_allreduce_fn fn, void *user) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->allreduce = fn; ctx->allreduce_user = user;
    return ADMM_OK;
}

// transports that only see host memory (MPI without GPU support, shared memory between the ranks of a node): the buffer is
// staged through pinned host memory around the caller's function
static int host_allreduce_trampoline(void *self, void *dev_buf, int64_t count, void *hip_stream) {
    admm_hip_ctx *ctx = (admm_hip_ctx *)self;
    hipStream_t st = (hipStream_t)hip_stream;
    if (!ctx->host_allreduce) return 1;
    if ((size_t)count > ctx->h_comm_cap) {
        (void)hipStreamSynchronize(st);      // the previous call's host-to-device copy may still be reading the old staging buffer
        if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);      // (only the staging buffer: every stream / event of the context belongs to admm_hip_destroy)
        ctx->h_comm = nullptr; ctx->h_comm_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_comm, sizeof(double) * (size_t)count, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
        ctx->h_comm_cap = (size_t)count;
    }
    const size_t bytes = sizeof(double) * (size_t)count;
    if (hipMemcpyAsync(ctx->h_comm, dev_buf, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    if (ctx->host_allreduce(ctx->host_allreduce_user, ctx->h_comm, count) != 0) return 1;
    if (hipMemcpyAsync(dev_buf, ctx->h_comm, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    return 0;
}
int admm_hip_set_host_allreduce(admm_hip_ctx *ctx, admm_hip_host_allreduce_fn fn, void *user) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->host_allreduce = fn; ctx->host_allreduce_user = user;
    ctx->allreduce = fn ? host_allreduce_trampoline : nullptr; ctx->allreduce_user = fn ? ctx : nullptr;
    return ADMM_OK;
}

int admm_hip_rccl_unique_id(void *id128) {
    if (!id128) return ADMM_ERR_ARG;
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (!R) { fprintf(stderr, "admm_hip: %s\n", why.c_str()); return ADMM_ERR_COMM; }
    nccl_uid id;
    if (R->GetUniqueId(&id) != 0) return ADMM_ERR_COMM;
    std::memcpy(id128, &id, sizeof id);
    return ADMM_OK;
}
int admm_hip_rccl_init(admm_hip_ctx *ctx, const void *id128, int rank, int world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return ADMM_ERR_ARG;
    if (ctx->device_id < 0) return fail(ctx, ADMM_ERR_HIP, "host-only context: no RCCL communicator");
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (!R) return fail(ctx, ADMM_ERR_COMM, "%s", why.c_str());
    HIPCHK(hipSetDevice(ctx->device_id));       // the communicator binds to the calling thread's current device
    nccl_uid id; std::memcpy(&id, id128, sizeof id);
    void *comm = nullptr;
    const int rc = R->CommInitRank(&comm, world, id, rank);
    if (rc != 0 || !comm) return fail(ctx, ADMM_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device_id, R->GetErrorString ? R->GetErrorString(rc) : "?");
    if (ctx->rccl_comm && ctx->rccl_owned) (void)R->CommDestroy(ctx->rccl_comm);
    ctx->rccl_comm = comm; ctx->rccl_owned = true;
    return ADMM_OK;
}
int admm_hip_set_rccl_comm(admm_hip_ctx *ctx, void *nccl_comm) {
    if (!ctx) return ADMM_ERR_ARG;
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (nccl_comm && !R) return fail(ctx, ADMM_ERR_COMM, "%s", why.c_str());
    if (ctx->rccl_comm && ctx->rccl_owned && R) (void)R->CommDestroy(ctx->rccl_comm);
    ctx->rccl_comm = nccl_comm; ctx->rccl_owned = false;
    return ADMM_OK;
}
// parity / bring-up hook: sums `count` doubles of a caller-owned DEVICE buffer through the installed communicator or hook
int admm_hip_debug_allreduce(admm_hip_ctx *ctx, void *dev_buf, int64_t count) {
    if (!ctx || ctx->device_id < 0 || !dev_buf || count < 0) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    TRY(do_allreduce(ctx, (double *)dev_buf, count));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

// a small HOST vector summed across the ranks through the transport the iterations use (the class mirror: the released
// MovingAnchors' positions, owner's values + zeros elsewhere); world 1: nothing to do
int admm_hip_allreduce_host(admm_hip_ctx *ctx, double *host_buf, int64_t count) {
    if (!ctx || ctx->device_id < 0 || !host_buf || count < 0) return ADMM_ERR_ARG;
    if (ctx->world <= 1 || count == 0) return ADMM_OK;
    HIPCHK(hipSetDevice(ctx->device_id));
    if ((size_t)count > ctx->d_small_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->d_small) (void)hipFree(ctx->d_small);
        ctx->d_small = nullptr; ctx->d_small_cap = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_small, sizeof(double) * (size_t)count));
        ctx->d_small_cap = (size_t)count;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_small, host_buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    TRY(do_allreduce(ctx, ctx->d_small, count));
    HIPCHK(hipMemcpyAsync(host_buf, ctx->d_small, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

int admm_hip_finalize(admm_hip_ctx *ctx) {
    if (!ctx) return ADMM_ERR_ARG;
    if (ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "already finalized");
    if (ctx->dt <= 0.0) { fprintf(stderr, "\n**Solver Error: timestep set to %gs, changing to 0.04s.\n", ctx->dt); ctx->dt = 0.04; }
    if (ctx->n_nodes < 1 || ctx->m3.size() != ctx->x.size()) return fail(ctx, ADMM_ERR_ARG, "**Solver Error: Problem with node data!");
    std::fill(ctx->v.begin(), ctx->v.end(), 0.0); // System.cpp:113
    for (const Explicit &E : ctx->explicits) for (int32_t v : E.idx) if (v >= ctx->n_nodes) return fail(ctx, ADMM_ERR_ARG, "explicit force references node %d (have %d)", v, ctx->n_nodes);
    if (const char *e = getenv("ADMM_HIP_SHARD")) ctx->shard_mode = (std::string(e) == "subtree") ? ADMM_SHARD_SUBTREE : ADMM_SHARD_CONTIGUOUS;
    TRY(host_assemble(ctx, false));
    TRY(host_factor(ctx, false));
    ctx->info.rank = ctx->rank; ctx->info.world = ctx->world;
    if (ctx->dense) ctx->shard_mode = 0;          // small systems: one-kernel solve, nothing to shard
    partition_subtrees(ctx);
    assign_elements(ctx);
    if (ctx->device_id >= 0) TRY(upload_all(ctx));
    ctx->finalized = true;
    return ADMM_OK;
}

int admm_hip_set_weights(admm_hip_ctx *ctx, int batch, const double *weights) {
    if (!ctx || batch < 0 || batch >= (int)ctx->batches.size() || !weights) return ADMM_ERR_ARG;
    if (!ctx->finalized) return fail(ctx, ADMM_ERR_STATE, "set_weights before finalize");
    Batch &b = ctx->batches[batch];
    if (b.kind == ADMM_KIND_GENERIC) std::copy(weights, weights + b.g_rows, b.g_roww.begin());
    else std::copy(weights, weights + b.n_total, b.weight.begin());
    return ADMM_OK;
}

int admm_hip_recompute_weights(admm_hip_ctx *ctx) {
    if (!ctx || !ctx->finalized) return ADMM_ERR_STATE;
    TRY(host_assemble(ctx, true));
    TRY(host_factor(ctx, true));
    if (ctx->device_id >= 0) {
        HIPCHK(hipSetDevice(ctx->device_id));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        TRY(panels_to_device(ctx));
        if (ctx->dense && ctx->d_ainv) HIPCHK(hipMemcpy(ctx->d_ainv, ctx->Ainv.data(), ctx->Ainv.size() * sizeof(double), hipMemcpyHostToDevice));
        for (Batch &b : ctx->batches) {
            if (b.kind == ADMM_KIND_GENERIC) {
                std::vector<double> coef(b.g_sval.size());
                for (size_t i = 0; i < coef.size(); ++i) { const double w = b.g_roww[b.g_srow_b[i]]; coef[i] = b.g_sval[i] * ((ctx->dt * ctx->dt) * (w * w)); }
                if (!coef.empty()) HIPCHK(hipMemcpy(b.d_g_scoef, coef.data(), sizeof(double) * coef.size(), hipMemcpyHostToDevice));
                if (b.d_g_scoef_res && !coef.empty()) {
                    for (size_t i = 0; i < coef.size(); ++i) { const double w = b.g_roww[b.g_srow_b[i]]; coef[i] = b.g_sval[i] * (w * w); }
                    HIPCHK(hipMemcpy(b.d_g_scoef_res, coef.data(), sizeof(double) * coef.size(), hipMemcpyHostToDevice));
                }
                continue;
            }
            const int nl = b.n_local;
            std::vector<double> w2h2(std::max(nl, 1)), w2(std::max(nl, 1));
            for (int el = 0; el < nl; ++el) { const double w = b.weight[b.local[el]]; w2[el] = w * w; w2h2[el] = (ctx->dt * ctx->dt) * (w * w); }
            if (nl) { HIPCHK(hipMemcpy(b.d_w2h2, w2h2.data(), sizeof(double) * nl, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(b.d_w2, w2.data(), sizeof(double) * nl, hipMemcpyHostToDevice)); }
        }
    }
    return ADMM_OK;
}

int admm_hip_update_anchors(admm_hip_ctx *ctx, int batch, const double *targets, const int32_t *active) {
    if (!ctx || batch < 0 || batch >= (int)ctx->batches.size()) return ADMM_ERR_ARG;
    Batch &b = ctx->batches[batch];
    if (b.kind != ADMM_KIND_ANCHOR) return fail(ctx, ADMM_ERR_ARG, "batch %d is not an anchor batch", batch);
    if (targets) std::copy(targets, targets + (size_t)3 * b.n_total, b.targets.begin());
    if (active) std::copy(active, active + b.n_total, b.active.begin());
    if (ctx->finalized && ctx->device_id >= 0 && b.n_local) {
        // asynchronous on the context's stream (it is ordered before the next step's kernels): this rank's targets / flags go
        // through a pinned staging buffer owned by the batch; the only wait is for the PREVIOUS update to have left that buffer
        HIPCHK(hipSetDevice(ctx->device_id));
        if (!b.h_tg) {
            HIPCHK(hipHostMalloc((void **)&b.h_tg, sizeof(double) * 3 * (size_t)b.n_local, hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&b.h_ac, sizeof(int32_t) * (size_t)b.n_local, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&b.upd_ev, hipEventDisableTiming));
        } else HIPCHK(hipEventSynchronize(b.upd_ev));
        for (int el = 0; el < b.n_local; ++el) { for (int j = 0; j < 3; ++j) b.h_tg[3 * (size_t)el + j] = b.targets[3 * (size_t)b.local[el] + j]; b.h_ac[el] = b.active[b.local[el]]; }
        if (targets) HIPCHK(hipMemcpyAsync(b.d_targets, b.h_tg, sizeof(double) * 3 * b.n_local, hipMemcpyHostToDevice, ctx->stream));
        if (active) HIPCHK(hipMemcpyAsync(b.d_active, b.h_ac, sizeof(int) * b.n_local, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipEventRecord(b.upd_ev, ctx->stream));
    }
    return ADMM_OK;
}

// records the next pooled event on the stream (timing mode only)
static int mark(admm_hip_ctx *ctx, bool on = true) {
    if (!ctx->timing || !on) return ADMM_OK;
    if (ctx->ev_used == ctx->evpool.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ctx->evpool.push_back(e); }
    HIPCHK(hipEventRecord(ctx->evpool[ctx->ev_used++], ctx->stream));
    return ADMM_OK;
}

int admm_hip_step(admm_hip_ctx *ctx, int admm_iters) {
    TRY(require_device(ctx));
    using namespace admm_dev;
    HIPCHK(hipSetDevice(ctx->device_id));
    const int n3 = 3 * ctx->n_nodes;
    TRY(ensure_local_streams(ctx));
    // event layout (timing mode): E0 | prologue | E1 | per TIMED iteration: S local E rhs E allreduce E [exchange: E E] fwd E bwd E | Ea | epilogue | Eb
    ctx->ev_used = 0; ctx->ev_iters = admm_iters; ctx->ev_timed = 0; ctx->ev_pending = ctx->timing;
    TRY(mark(ctx));
    if (ctx->frames++ > 0)      // the blocks of the large tet batches by what they cost in the frame before
        for (const Batch &b : ctx->batches) if (b.n_blocks_ordered) {
            if (b.grp_blk.empty()) hipLaunchKernelGGL(order_by_cost_kernel, dim3(1), dim3(1024), 0, ctx->stream, b.n_blocks_ordered, b.d_cost, b.d_order);
            else for (size_t g = 0; g + 1 < b.grp_blk.size(); ++g) if (b.grp_blk[g + 1] > b.grp_blk[g])
                hipLaunchKernelGGL(order_by_cost_kernel, dim3(1), dim3(1024), 0, ctx->stream, b.grp_blk[g + 1] - b.grp_blk[g], b.d_cost + b.grp_blk[g], b.d_order + b.grp_blk[g]);
        }
    if (ctx->explicit_simple) {
        hipLaunchKernelGGL(prologue_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, ctx->dt, ctx->grav, ctx->d_x, ctx->d_v, ctx->d_m3, ctx->d_mxbar, ctx->d_xcur);
    } else {
        for (const Explicit &E : ctx->explicits) {      // in list order, like System.cpp:37-39
            if (E.type == ADMM_EXPLICIT_CONST) {
                const int cnt = E.idx.empty() ? ctx->n_nodes : E.n;
                if (cnt) hipLaunchKernelGGL(explicit_const_kernel, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, cnt, (const int *)E.d_idx, ctx->dt, E.dir[0], E.dir[1], E.dir[2], ctx->d_v);
            } else if (E.n) {
                hipLaunchKernelGGL(wind_serial_kernel, dim3(1), dim3(1024), 0, ctx->stream, E.n_levels, (const int *)E.d_level_ptr, (const int *)E.d_idx, ctx->dt, E.dir[0], E.dir[1], E.dir[2], ctx->d_x, ctx->d_v);
            }
        }
        hipLaunchKernelGGL(xbar_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, ctx->dt, ctx->d_x, ctx->d_v, ctx->d_m3, ctx->d_mxbar, ctx->d_xcur);
    }
    TRY(mark(ctx));
    const bool track = ctx->res_on || ctx->tol_r > 0.0;
    if (track) TRY(ensure_residual_buffers(ctx, admm_iters));
    ctx->keep_z = ctx->keep_z_user || track;
    if (ctx->n_gen_rows) {      // user-defined forces: curr_z = D * m_x before the loop (System.cpp:43)
        TRY(generic_begin(ctx, ctx->d_x));
        HIPCHK(hipEventSynchronize(ctx->gen_ev));
        std::memcpy(ctx->h_gen_z, ctx->h_gen_dx, sizeof(double) * (size_t)ctx->n_gen_rows);
    }
    // world > 1: the iteration contains an all-reduce.  A host hook cannot be captured; ncclAllReduce can (RCCL collectives are
    // stream-ordered device work), so with the communicator inside the library the multi-GPU iteration is one graph launch too.
    const bool comm_capturable = ctx->world == 1 || (ctx->rccl_comm != nullptr && ctx->graph_comm);
    const bool use_graph = ctx->graph_enabled && (ctx->graph_forced || ctx->n_nodes < 100000) && comm_capturable && !(ctx->timing && ctx->timing_stride <= 1) && !track && admm_iters > 0 && !ctx->n_gen_rows;
    if (use_graph && !ctx->iter_exec) {   // capture one iteration; every kernel argument is a fixed device address
        // a stream that cannot be captured (caller-supplied, already capturing ...) is not an error: launch eagerly instead
        const hipError_t be = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
        int rc = be == hipSuccess ? launch_local(ctx) : ADMM_ERR_HIP;
        if (!rc) rc = launch_rhs(ctx);
        if (!rc && ctx->world > 1 && ctx->levels_top.empty()) rc = do_allreduce(ctx, ctx->d_y, (int64_t)n3);   // contiguous sharding (subtree: inside launch_solve)
        if (!rc) rc = launch_solve(ctx, nullptr);
        hipGraph_t g = nullptr;
        const hipError_t ce = be == hipSuccess ? hipStreamEndCapture(ctx->stream, &g) : be;
        if (rc || ce != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); ctx->graph_enabled = false; fprintf(stderr, "admm_hip: graph capture unavailable, launching eagerly\n"); }
        else {
            ctx->iter_graph = g;
            if (hipGraphInstantiate(&ctx->iter_exec, g, nullptr, nullptr, 0) != hipSuccess) { ctx->iter_exec = nullptr; (void)hipGraphDestroy(g); ctx->iter_graph = nullptr; (void)hipGetLastError(); ctx->graph_enabled = false; }
        }
    }
    // the frame's whole loop as one graph (no timing events inside; iteration counts beyond 64 keep the per-iteration graph)
    // (captured for an iteration count only once two calls in a row have asked for it: a caller that changes the count from frame to frame would
    //  otherwise pay a graph instantiation per frame)
    const bool same_count = admm_iters == ctx->last_step_iters;
    ctx->last_step_iters = admm_iters;
    const bool use_frame_graph = use_graph && ctx->iter_exec && ctx->frame_graph_on && !ctx->timing && admm_iters >= 2 && admm_iters <= 64 && !(ctx->pipe > 1) &&
                                 (same_count || (ctx->frame_exec && ctx->frame_iters == admm_iters));
    if (use_frame_graph && (!ctx->frame_exec || ctx->frame_iters != admm_iters)) {
        if (ctx->frame_exec) { (void)hipGraphExecDestroy(ctx->frame_exec); ctx->frame_exec = nullptr; }
        if (ctx->frame_graph) { (void)hipGraphDestroy(ctx->frame_graph); ctx->frame_graph = nullptr; }
        const hipError_t be = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
        int rc = be == hipSuccess ? ADMM_OK : ADMM_ERR_HIP;
        for (int it = 0; it < admm_iters && !rc; ++it) {
            rc = launch_local(ctx);
            if (!rc) rc = launch_rhs(ctx);
            if (!rc && ctx->world > 1 && ctx->levels_top.empty()) rc = do_allreduce(ctx, ctx->d_y, (int64_t)n3);
            if (!rc) rc = launch_solve(ctx, nullptr);
        }
        hipGraph_t g = nullptr;
        const hipError_t ce = be == hipSuccess ? hipStreamEndCapture(ctx->stream, &g) : be;
        if (rc || ce != hipSuccess || !g || hipGraphInstantiate(&ctx->frame_exec, g, nullptr, nullptr, 0) != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError(); ctx->frame_exec = nullptr; ctx->frame_graph_on = false;      // the per-iteration graph stays
        } else { ctx->frame_graph = g; ctx->frame_iters = admm_iters; }
    }
    ctx->res_n = 0;
    int iters_done = 0;
    const int stride = std::max(1, ctx->timing_stride);
    // pipelined groups: the whole frame's ADMM loop on G streams (no timing events, residuals or user forces inside)
    if (ctx->pipe > 1 && !ctx->timing && !track && !ctx->n_gen_rows && admm_iters > 0) {
        TRY(pipe_frame(ctx, admm_iters));
        ctx->ev_iters = admm_iters;
        TRY(mark(ctx));
        hipLaunchKernelGGL(epilogue_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, ctx->dt, ctx->d_x, ctx->d_v, ctx->d_xcur);
        HIPCHK(hipGetLastError());
        TRY(mark(ctx));
        return ADMM_OK;
    }
    if (use_frame_graph && ctx->frame_exec && ctx->frame_iters == admm_iters) { HIPCHK(hipGraphLaunch(ctx->frame_exec, ctx->stream)); iters_done = admm_iters; }
    for (int it = iters_done; it < admm_iters; ++it) {
        // the sampled iterations rotate from frame to frame: an iteration's cost depends on its position in the frame (the first
        // ones after the prologue do the most line-search work), a fixed phase would bias the average
        const bool timed = ctx->timing && ((it + ctx->timing_frame) % stride == stride - 1);
        if (use_graph && ctx->iter_exec && !timed) { HIPCHK(hipGraphLaunch(ctx->iter_exec, ctx->stream)); iters_done = it + 1; continue; }
        if (timed) ++ctx->ev_timed;
        TRY(mark(ctx, timed));
        if (track) TRY(residual_snapshot(ctx, it == 0));
        TRY(generic_begin(ctx, ctx->d_xcur));
        TRY(launch_local(ctx, -1, -1, nullptr, track));
        TRY(generic_finish(ctx));
        TRY(mark(ctx, timed));
        if (track) { TRY(launch_residuals(ctx, it)); ctx->res_n = it + 1; }
        TRY(launch_rhs(ctx));
        TRY(mark(ctx, timed));
        if (ctx->world > 1 && ctx->levels_top.empty()) {     // contiguous sharding: the whole RHS is summed, the solve is replicated
            TRY(do_allreduce(ctx, ctx->d_y, (int64_t)n3));
        }
        TRY(mark(ctx, timed));
        // timing mode: one event between the sweeps; under subtree sharding two more around the exchange inside the forward
        // sweep (pack, all-reduce, unpack), so that allreduce_ms shows the communication and solve_fwd_ms only the sweeps
        hipEvent_t mid = nullptr, ex0 = nullptr, ex1 = nullptr;
        if (timed) {
            const int want = ctx->levels_top.empty() ? 1 : 3;
            while (ctx->ev_used + want > ctx->evpool.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ctx->evpool.push_back(e); }
            if (want == 3) { ex0 = ctx->evpool[ctx->ev_used++]; ex1 = ctx->evpool[ctx->ev_used++]; }
            mid = ctx->evpool[ctx->ev_used++];
        }
        TRY(launch_solve(ctx, mid, ex0, ex1));
        TRY(mark(ctx, timed));
        iters_done = it + 1;
        if (ctx->tol_r > 0.0 && (it + 1) % ctx->check_every == 0 && it + 1 < admm_iters) {   // convergence test: one round trip
            double rs[2];
            HIPCHK(hipMemcpyAsync(rs, ctx->d_res + 2 * (size_t)it, sizeof rs, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (std::sqrt(rs[0]) <= ctx->tol_r && std::sqrt(rs[1]) <= ctx->tol_s) break;
        }
    }
    ctx->ev_iters = iters_done;
    if (ctx->timing) ++ctx->timing_frame;
    TRY(mark(ctx));
    TRY(shard_sync_x(ctx));
    hipLaunchKernelGGL(epilogue_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, n3, ctx->dt, ctx->d_x, ctx->d_v, ctx->d_xcur);
    HIPCHK(hipGetLastError());
    TRY(mark(ctx));
    return ADMM_OK;
}

int admm_hip_sync(admm_hip_ctx *ctx) {
    TRY(require_device(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

static int get_nodes(admm_hip_ctx *ctx, const double *dsrc, double *out) {
    TRY(require_device(ctx));
    HIPCHK(hipSetDevice(ctx->device_id));
    std::vector<double> tmp(3 * (size_t)ctx->n_nodes);
    HIPCHK(hipMemcpyAsync(tmp.data(), dsrc, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const std::vector<int> &perm = ctx->F.perm;
    for (int i = 0; i < ctx->n_nodes; ++i) for (int c = 0; c < 3; ++c) out[3 * (size_t)perm[i] + c] = tmp[3 * (size_t)i + c];
    return ADMM_OK;
}
static int set_nodes(admm_hip_ctx *ctx, double *ddst, const double *in) {
    TRY(require_device(ctx));
    HIPCHK(hipSetDevice(ctx->device_id));
    std::vector<double> tmp(3 * (size_t)ctx->n_nodes);
    const std::vector<int> &perm = ctx->F.perm;
    for (int i = 0; i < ctx->n_nodes; ++i) for (int c = 0; c < 3; ++c) tmp[3 * (size_t)i + c] = in[3 * (size_t)perm[i] + c];
    HIPCHK(hipMemcpyAsync(ddst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

int admm_hip_get_x(admm_hip_ctx *ctx, double *x) {
    if (ctx && !ctx->finalized && x) { std::copy(ctx->x.begin(), ctx->x.end(), x); return ADMM_OK; }
    if (ctx && ctx->finalized && ctx->device_id < 0 && x) { std::copy(ctx->x.begin(), ctx->x.end(), x); return ADMM_OK; }
    return x ? get_nodes(ctx, ctx ? ctx->d_x : nullptr, x) : ADMM_ERR_ARG;
}
int admm_hip_set_x(admm_hip_ctx *ctx, const double *x) {
    if (!ctx || !x) return ADMM_ERR_ARG;
    std::copy(x, x + ctx->x.size(), ctx->x.begin());
    if (!ctx->finalized || ctx->device_id < 0) return ADMM_OK;
    return set_nodes(ctx, ctx->d_x, x);
}
int admm_hip_get_v(admm_hip_ctx *ctx, double *v) {
    if (ctx && (!ctx->finalized || ctx->device_id < 0) && v) { std::copy(ctx->v.begin(), ctx->v.end(), v); return ADMM_OK; }
    return v ? get_nodes(ctx, ctx ? ctx->d_v : nullptr, v) : ADMM_ERR_ARG;
}
int admm_hip_set_v(admm_hip_ctx *ctx, const double *v) {
    if (!ctx || !v) return ADMM_ERR_ARG;
    std::copy(v, v + ctx->v.size(), ctx->v.begin());
    if (!ctx->finalized || ctx->device_id < 0) return ADMM_OK;
    return set_nodes(ctx, ctx->d_v, v);
}

// ---- frame boundary of the class API ------------------------------------------------------------------------
// One DMA per vector straight from / into the caller's memory (pinned with admm_hip_pin_host: full PCIe rate, truly
// asynchronous), the reordering between the caller's node order and the factor order on the device.
int admm_hip_pin_host(admm_hip_ctx *ctx, void *p, size_t bytes, int on) {
    if (!ctx || !p) return ADMM_ERR_ARG;
    if (ctx->device_id < 0) return ADMM_OK;
    HIPCHK(hipSetDevice(ctx->device_id));
    // a refused registration is not an error of the solver (pageable memory works, only slower): report it, clear HIP's
    // sticky last-error so that no later hipGetLastError() check trips over it, leave last_error alone
    const hipError_t e = on ? hipHostRegister(p, bytes, hipHostRegisterDefault) : hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return ADMM_ERR_HIP; }
    return ADMM_OK;
}
// small systems: the page-locked [x | v] buffer the state kernels address directly (NULL: use the DMA path)
static double *state_buffer(admm_hip_ctx *ctx) {
    if (ctx->n_nodes > ctx->state_direct_max_nodes || !ctx->d_iperm) return nullptr;
    const size_t need = 6 * (size_t)ctx->n_nodes;
    if (ctx->h_state_cap < need) {
        if (ctx->h_state) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->h_state); ctx->h_state = nullptr; ctx->h_state_cap = 0; }
        void *dev = nullptr;
        if (hipHostMalloc((void **)&ctx->h_state, sizeof(double) * need, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dev, ctx->h_state, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->h_state) { (void)hipHostFree(ctx->h_state); ctx->h_state = nullptr; }
            ctx->state_direct_max_nodes = 0;      // no mapped host memory here: the DMA path from now on
            return nullptr;
        }
        ctx->h_state_dev = (double *)dev; ctx->h_state_cap = need;
        if (!ctx->state_in_ev && hipEventCreateWithFlags(&ctx->state_in_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->state_direct_max_nodes = 0; return nullptr; }
    }
    return ctx->h_state;
}

int admm_hip_upload_state(admm_hip_ctx *ctx, const double *x, const double *v) {
    TRY(require_device(ctx));
    HIPCHK(hipSetDevice(ctx->device_id));
    const int n3 = 3 * ctx->n_nodes;
    const size_t bytes = sizeof(double) * (size_t)n3;
    if (x && v) if (double *h = state_buffer(ctx)) {
        if (ctx->state_in_pending) { HIPCHK(hipEventSynchronize(ctx->state_in_ev)); ctx->state_in_pending = false; }      // the previous upload has left the buffer
        std::memcpy(h, x, bytes); std::memcpy(h + n3, v, bytes);
        hipLaunchKernelGGL(admm_dev::state_in_kernel, dim3((2 * n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_iperm, (const double *)ctx->h_state_dev, ctx->d_x, ctx->d_v);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->state_in_ev, ctx->stream)); ctx->state_in_pending = true;
        return ADMM_OK;
    }
    if (x && v && ctx->state_zero_copy && ctx->d_iperm) {      // the caller's page-locked vectors addressed by one kernel (no DMA, no staging)
        void *dx = nullptr, *dv = nullptr;
        if (hipHostGetDevicePointer(&dx, (void *)x, 0) == hipSuccess && hipHostGetDevicePointer(&dv, (void *)v, 0) == hipSuccess) {
            hipLaunchKernelGGL(admm_dev::state_in2_kernel, dim3(std::min((2 * n3 + 255) / 256, 4096)), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_iperm,
                               (const double *)dx, (const double *)dv, ctx->d_x, ctx->d_v);
            HIPCHK(hipGetLastError());
            return ADMM_OK;
        }
        (void)hipGetLastError();      // not page-locked (admm_hip_pin_host was not called on them): the DMA path below
    }
    if (x) {
        HIPCHK(hipMemcpyAsync(ctx->d_stage, x, bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(admm_dev::permute_in_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_perm, (const double *)ctx->d_stage, ctx->d_x);
    }
    if (v) {
        HIPCHK(hipMemcpyAsync(ctx->d_stage + n3, v, bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(admm_dev::permute_in_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_perm, (const double *)(ctx->d_stage + n3), ctx->d_v);
    }
    HIPCHK(hipGetLastError());
    return ADMM_OK;
}
int admm_hip_download_state(admm_hip_ctx *ctx, double *x, double *v) {
    TRY(require_device(ctx));
    HIPCHK(hipSetDevice(ctx->device_id));
    const int n3 = 3 * ctx->n_nodes;
    const size_t bytes = sizeof(double) * (size_t)n3;
    if (x && v) if (double *h = state_buffer(ctx)) {
        hipLaunchKernelGGL(admm_dev::state_out_kernel, dim3((2 * n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_iperm, (const double *)ctx->d_x, (const double *)ctx->d_v, ctx->h_state_dev);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));      // (also past any pending upload: the buffer is the host's again)
        ctx->state_in_pending = false;
        std::memcpy(x, h, bytes); std::memcpy(v, h + n3, bytes);
        return ADMM_OK;
    }
    if (x && v && ctx->state_zero_copy && ctx->d_iperm) {
        void *dx = nullptr, *dv = nullptr;
        if (hipHostGetDevicePointer(&dx, (void *)x, 0) == hipSuccess && hipHostGetDevicePointer(&dv, (void *)v, 0) == hipSuccess) {
            hipLaunchKernelGGL(admm_dev::state_out2_kernel, dim3(std::min((2 * n3 + 255) / 256, 4096)), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_iperm,
                               (const double *)ctx->d_x, (const double *)ctx->d_v, (double *)dx, (double *)dv);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ctx->stream));
            return ADMM_OK;
        }
        (void)hipGetLastError();
    }
    if (x) {
        hipLaunchKernelGGL(admm_dev::permute_out_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_perm, (const double *)ctx->d_x, ctx->d_stage);
        HIPCHK(hipMemcpyAsync(x, ctx->d_stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (v) {
        hipLaunchKernelGGL(admm_dev::permute_out_kernel, dim3((n3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const int *)ctx->d_perm, (const double *)ctx->d_v, ctx->d_stage + n3);
        HIPCHK(hipMemcpyAsync(v, ctx->d_stage + n3, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

// SoA [rows][n] device -> element-major [n][rows] host
static int read_soa(admm_hip_ctx *ctx, const double *d, int rows, int n, double *out) {
    if (!out || n == 0) return ADMM_OK;
    std::vector<double> tmp((size_t)rows * n);
    HIPCHK(hipMemcpy(tmp.data(), d, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int e = 0; e < n; ++e) for (int r = 0; r < rows; ++r) out[(size_t)e * rows + r] = tmp[(size_t)r * n + e];
    return ADMM_OK;
}

int admm_hip_read_local(admm_hip_ctx *ctx, int batch, double *u, double *z, double *state, int32_t *n_iters) {
    TRY(require_device(ctx));
    if (batch < 0 || batch >= (int)ctx->batches.size()) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const Batch &b = ctx->batches[batch];
    if (b.kind == ADMM_KIND_GENERIC) {      // u, z of this rank's user forces, element after element (they live on the host)
        size_t o = 0;
        for (int el = 0; el < b.n_local; ++el) for (int64_t r = b.g_elem_row[b.local[el]]; r < b.g_elem_row[b.local[el] + 1]; ++r, ++o) {
            if (u) u[o] = ctx->h_gen_u[b.g_row0 + r];
            if (z) z[o] = ctx->h_gen_z[b.g_row0 + r];
        }
        return ADMM_OK;
    }
    const int rows = ADMM_KIND_ROWS[b.kind];
    TRY(read_soa(ctx, b.d_u, rows, b.n_local, u));
    TRY(read_soa(ctx, b.d_z, rows, b.n_local, z));
    if (state && ADMM_KIND_STATE[b.kind]) TRY(read_soa(ctx, b.d_state, 4, b.n_local, state));
    if (state && b.kind == ADMM_KIND_ANCHOR && b.n_local) HIPCHK(hipMemcpy(state, b.d_targets, sizeof(double) * 3 * b.n_local, hipMemcpyDeviceToHost));
    if (n_iters && b.n_local) HIPCHK(hipMemcpy(n_iters, b.d_niters, sizeof(int) * b.n_local, hipMemcpyDeviceToHost));
    return ADMM_OK;
}

int admm_hip_write_local(admm_hip_ctx *ctx, int batch, const double *u, const double *state) {
    TRY(require_device(ctx));
    if (batch < 0 || batch >= (int)ctx->batches.size()) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const Batch &b = ctx->batches[batch];
    if (b.kind == ADMM_KIND_GENERIC) {
        size_t o = 0;
        if (u) for (int el = 0; el < b.n_local; ++el) for (int64_t r = b.g_elem_row[b.local[el]]; r < b.g_elem_row[b.local[el] + 1]; ++r, ++o) ctx->h_gen_u[b.g_row0 + r] = u[o];
        return ADMM_OK;
    }
    const int rows = ADMM_KIND_ROWS[b.kind], n = b.n_local;
    if (u && n) {
        std::vector<double> tmp((size_t)rows * n);
        for (int e = 0; e < n; ++e) for (int r = 0; r < rows; ++r) tmp[(size_t)r * n + e] = u[(size_t)e * rows + r];
        HIPCHK(hipMemcpy(b.d_u, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (state && n && ADMM_KIND_STATE[b.kind]) {
        std::vector<double> tmp((size_t)4 * n);
        for (int e = 0; e < n; ++e) for (int r = 0; r < 4; ++r) tmp[(size_t)r * n + e] = state[(size_t)e * 4 + r];
        HIPCHK(hipMemcpy(b.d_state, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return ADMM_OK;
}

int admm_hip_read_rest(admm_hip_ctx *ctx, int batch, double *weight, double *rest, int32_t *global_idx) {
    if (!ctx || !ctx->finalized || batch < 0 || batch >= (int)ctx->batches.size()) return ADMM_ERR_ARG;
    const Batch &b = ctx->batches[batch];
    if (b.kind == ADMM_KIND_GENERIC) { if (global_idx) std::copy(b.global_idx.begin(), b.global_idx.end(), global_idx); return ADMM_OK; }   // rest data and weights are the caller's
    if (weight) std::copy(b.weight.begin(), b.weight.end(), weight);
    if (rest) std::copy(b.rest.begin(), b.rest.end(), rest);
    if (global_idx) std::copy(b.global_idx.begin(), b.global_idx.end(), global_idx);
    return ADMM_OK;
}

int admm_hip_keep_z(admm_hip_ctx *ctx, int on) {
    if (!ctx) return ADMM_ERR_ARG;
    if (getenv("ADMM_HIP_KEEP_Z")) return ADMM_OK;      // the environment decides
    if (ctx->keep_z_user != (on != 0)) {      // captured iterations carry the flag in their kernel arguments: capture again
        drop_iteration_graphs(ctx);
        for (int q = 0; q < 3; ++q) {
            if (ctx->pipe_exec[q]) { (void)hipGraphExecDestroy(ctx->pipe_exec[q]); ctx->pipe_exec[q] = nullptr; }
            if (ctx->pipe_graph_h[q]) { (void)hipGraphDestroy(ctx->pipe_graph_h[q]); ctx->pipe_graph_h[q] = nullptr; }
        }
    }
    ctx->keep_z_user = on != 0;
    return ADMM_OK;
}

namespace { struct KeepZ { admm_hip_ctx *c; bool old; explicit KeepZ(admm_hip_ctx *c_) : c(c_), old(c_->keep_z) { c->keep_z = true; } ~KeepZ() { c->keep_z = old; } }; }

int admm_hip_local_step_only(admm_hip_ctx *ctx, const double *x_cur) {
    TRY(require_device(ctx));
    if (!x_cur) return ADMM_ERR_ARG;
    TRY(ensure_local_streams(ctx));
    KeepZ kz(ctx);
    TRY(set_nodes(ctx, ctx->d_xcur, x_cur));
    TRY(generic_begin(ctx, ctx->d_xcur));
    TRY(launch_local(ctx));
    TRY(generic_finish(ctx));
    TRY(launch_rhs(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

// Parity-test hook: one project() of every local element of `batch` on
// caller-supplied D_i x rows (element-major [n_local][rows]) instead of the
// gather from x -- replays the reference's recorded (Dx,u,state)->(u,z,state) tuples.
int admm_hip_local_step_dx(admm_hip_ctx *ctx, int batch, const double *dx) {
    TRY(require_device(ctx));
    if (batch < 0 || batch >= (int)ctx->batches.size() || !dx) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    Batch &b = ctx->batches[batch];
    if (b.kind == ADMM_KIND_GENERIC) return fail(ctx, ADMM_ERR_UNSUPPORTED, "local_step_dx: a generic batch's project() is the caller's own code");
    const int rows = ADMM_KIND_ROWS[b.kind], n = b.n_local;
    if (n == 0) return ADMM_OK;
    std::vector<double> tmp((size_t)rows * n);
    for (int e = 0; e < n; ++e) for (int r = 0; r < rows; ++r) tmp[(size_t)r * n + e] = dx[(size_t)e * rows + r];
    if (!b.d_dx_buf) TRY(dalloc(ctx, &b.d_dx_buf, tmp.size()));
    HIPCHK(hipMemcpy(b.d_dx_buf, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    b.d_dx_override = b.d_dx_buf;
    KeepZ kz(ctx);
    int rc = launch_local(ctx, batch);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    b.d_dx_override = nullptr;             // production launches never see the override
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, ADMM_ERR_HIP, "local_step_dx: %s", hipGetErrorString(e));
    return ADMM_OK;
}

int admm_hip_solve_only(admm_hip_ctx *ctx, const double *b, double *x) {
    TRY(require_device(ctx));
    if (!b || !x) return ADMM_ERR_ARG;
    TRY(set_nodes(ctx, ctx->d_y, b));
    if (!ctx->levels_top.empty())   // subtree sharding sums the top rows over the ranks: every rank was handed the whole b, keep it once
        hipLaunchKernelGGL(admm_dev::shard_mask_nodes_kernel, dim3((3 * ctx->n_nodes + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_nodes, (const unsigned char *)ctx->d_keep_mask, ctx->d_y);
    TRY(launch_solve(ctx, nullptr));
    TRY(shard_sync_x(ctx));
    return get_nodes(ctx, ctx->d_xcur, x);
}

int admm_hip_apply_A(admm_hip_ctx *ctx, const double *x, double *y) {
    if (!ctx || !ctx->finalized || !x || !y) return ADMM_ERR_ARG;
    sym_apply(ctx->A, x, y);
    return ADMM_OK;
}

// Validation hook for the CPU test-suite: evaluates the two panel sweeps of
// factor.hpp on the host.  NOT used by admm_hip_step or any product path.
int admm_hip_debug_panel_solve_host(admm_hip_ctx *ctx, const double *b, double *x) {
    if (!ctx || !ctx->finalized || !b || !x) return ADMM_ERR_ARG;
    if (ctx->F.panels.empty() && ctx->d_panels) {      // factored on the device: the host sweeps then check the DEVICE's factor
        ctx->F.panels.resize((size_t)ctx->F.panels_size);
        HIPCHK(hipSetDevice(ctx->device_id));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(ctx->F.panels.data(), ctx->d_panels, sizeof(double) * ctx->F.panels.size(), hipMemcpyDeviceToHost));
    }
    panel_solve_host(ctx->F, b, x);
    return ADMM_OK;
}

namespace admm_dev {
__global__ void debug_math_kernel(int op, int64_t n, const double *__restrict__ in, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = op == 0 ? admm_log(in[i]) : admm_exp(in[i]);
}
} // namespace admm_dev
int admm_hip_debug_gemm(admm_hip_ctx *ctx, int m, int n, int k, int lda, int ldb, int ldc, int flags, double alpha, double beta,
                        const double *A, int64_t size_a, const double *B, int64_t size_b, double *C, int64_t size_c) {
    TRY(require_device(ctx));
    if (!A || !B || !C || m < 1 || n < 1 || k < 1 || size_a < 1 || size_b < 1 || size_c < 1) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    double *dA = nullptr, *dB = nullptr, *dC = nullptr; admm_dev::GemmTask *dT = nullptr;
    auto done = [&](int rc) { for (void *p : {(void *)dA, (void *)dB, (void *)dC, (void *)dT}) if (p) (void)hipFree(p); return rc ? fail(ctx, rc, "debug_gemm failed") : ADMM_OK; };
    if (hipMalloc(&dA, 8 * size_a) != hipSuccess || hipMalloc(&dB, 8 * size_b) != hipSuccess || hipMalloc(&dC, 8 * size_c) != hipSuccess || hipMalloc(&dT, sizeof(admm_dev::GemmTask)) != hipSuccess) return done(ADMM_ERR_HIP);
    if (hipMemcpy(dA, A, 8 * size_a, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dB, B, 8 * size_b, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dC, C, 8 * size_c, hipMemcpyHostToDevice) != hipSuccess) return done(ADMM_ERR_HIP);
    admm_dev::GemmTask T{}; T.A = dA; T.B = dB; T.C = dC; T.m = m; T.n = n; T.k = k; T.lda = lda; T.ldb = ldb; T.ldc = ldc; T.flags = flags; T.alpha = alpha; T.beta = beta;
    if (hipMemcpy(dT, &T, sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return done(ADMM_ERR_HIP);
    hipLaunchKernelGGL(admm_dev::gemm_f64_kernel, dim3(((m + 63) / 64) * ((n + 63) / 64), 1), dim3(256), 0, ctx->stream, (const admm_dev::GemmTask *)dT);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(C, dC, 8 * size_c, hipMemcpyDeviceToHost) != hipSuccess) return done(ADMM_ERR_HIP);
    return done(ADMM_OK);
}
int admm_hip_debug_potrf_inv(admm_hip_ctx *ctx, int w, int ld, double *blk, double *out) {
    TRY(require_device(ctx));
    if (!blk || !out || w < 1 || w > 64 || ld < w) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    const size_t sz = 8 * (size_t)ld * w;
    double *dB = nullptr, *dO = nullptr; admm_dev::PotrfTask *dT = nullptr; int *dF = nullptr;
    auto done = [&](int rc) { for (void *p : {(void *)dB, (void *)dO, (void *)dT, (void *)dF}) if (p) (void)hipFree(p); return rc; };
    if (hipMalloc(&dB, sz) != hipSuccess || hipMalloc(&dO, sz) != hipSuccess || hipMalloc(&dT, sizeof(admm_dev::PotrfTask)) != hipSuccess || hipMalloc(&dF, sizeof(int)) != hipSuccess) return done(fail(ctx, ADMM_ERR_HIP, "debug_potrf_inv: hipMalloc"));
    admm_dev::PotrfTask T{}; T.blk = dB; T.out = dO; T.w = w; T.ld = ld; T.ldo = ld; T.id = 0;
    if (hipMemcpy(dB, blk, sz, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dO, 0, sz) != hipSuccess || hipMemset(dF, 0, sizeof(int)) != hipSuccess || hipMemcpy(dT, &T, sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return done(fail(ctx, ADMM_ERR_HIP, "debug_potrf_inv: copy"));
    hipLaunchKernelGGL(admm_dev::potrf_inv_kernel, dim3(1), dim3(256), 0, ctx->stream, (const admm_dev::PotrfTask *)dT, dF);
    int failed = 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(blk, dB, sz, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(out, dO, sz, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&failed, dF, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(ctx, ADMM_ERR_HIP, "debug_potrf_inv: run"));
    return done(failed ? fail(ctx, ADMM_ERR_FACTOR, "block is not positive definite") : ADMM_OK);
}
int admm_hip_debug_math(admm_hip_ctx *ctx, int op, int64_t n, const double *in, double *out) {
    TRY(require_device(ctx));
    if (!in || !out || n < 0 || op < 0 || op > 1) return ADMM_ERR_ARG;
    if (n == 0) return ADMM_OK;
    HIPCHK(hipSetDevice(ctx->device_id));
    double *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc(&d_in, sizeof(double) * n)); 
    if (hipMalloc(&d_out, sizeof(double) * n) != hipSuccess) { (void)hipFree(d_in); return fail(ctx, ADMM_ERR_HIP, "hipMalloc failed"); }
    int rc = ADMM_OK;
    if (hipMemcpy(d_in, in, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) rc = ADMM_ERR_HIP;
    if (!rc) {
        hipLaunchKernelGGL(admm_dev::debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, n, (const double *)d_in, d_out);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = ADMM_ERR_HIP;
    }
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc ? fail(ctx, rc, "debug_math failed") : ADMM_OK;
}

int admm_hip_get_info(admm_hip_ctx *ctx, admm_hip_info *info) {
    if (!ctx || !info) return ADMM_ERR_ARG;
    *info = ctx->info;
    return ADMM_OK;
}

int admm_hip_enable_residuals(admm_hip_ctx *ctx, int on) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->res_on = on != 0;
    return ADMM_OK;
}
int admm_hip_set_tolerance(admm_hip_ctx *ctx, double eps_r, double eps_s, int check_every) {
    if (!ctx || check_every < 1 || eps_r < 0.0 || eps_s < 0.0) return ADMM_ERR_ARG;
    ctx->tol_r = eps_r; ctx->tol_s = eps_s; ctx->check_every = check_every;
    return ADMM_OK;
}
int admm_hip_get_residuals(admm_hip_ctx *ctx, double *r_norm, double *s_norm, int capacity, int *n_iters) {
    TRY(require_device(ctx));
    if (capacity < 0 || (capacity && (!r_norm || !s_norm))) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int n = std::min(ctx->res_n, capacity);
    if (n > 0) {
        std::vector<double> h(2 * (size_t)n);
        HIPCHK(hipMemcpy(h.data(), ctx->d_res, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) { r_norm[i] = std::sqrt(h[2 * (size_t)i]); s_norm[i] = std::sqrt(h[2 * (size_t)i + 1]); }
    }
    if (n_iters) *n_iters = ctx->ev_iters;
    return ADMM_OK;
}
int admm_hip_enable_timing(admm_hip_ctx *ctx, int on) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->timing = on != 0;
    ctx->timing_stride = on > 1 ? on : 1;      // on = k > 1: events around every k-th ADMM iteration only
    ctx->timing_frame = 0;
    return ADMM_OK;
}
// Reads back the events of the last step recorded in timing mode (waits for it).
int admm_hip_get_timing(admm_hip_ctx *ctx, admm_hip_timing *t) {
    if (!ctx || !t) return ADMM_ERR_ARG;
    if (ctx->ev_pending && ctx->device_id >= 0) {
        HIPCHK(hipSetDevice(ctx->device_id));
        const std::vector<hipEvent_t> &E = ctx->evpool;
        const size_t per = ctx->levels_top.empty() ? 6 : 8;       // events per TIMED ADMM iteration (see admm_hip_step)
        const size_t need = 4 + per * (size_t)ctx->ev_timed;
        if (ctx->ev_used != need) return fail(ctx, ADMM_ERR_STATE, "timing events incomplete (%zu of %zu)", ctx->ev_used, need);
        HIPCHK(hipEventSynchronize(E[need - 1]));
        admm_hip_timing T{};
        float v;
        HIPCHK(hipEventElapsedTime(&v, E[0], E[1])); T.prologue_ms = v;
        for (int it = 0; it < ctx->ev_timed; ++it) {
            const size_t b = 2 + per * (size_t)it;
            HIPCHK(hipEventElapsedTime(&v, E[b], E[b + 1])); T.local_ms += v;
            HIPCHK(hipEventElapsedTime(&v, E[b + 1], E[b + 2])); T.rhs_ms += v;
            HIPCHK(hipEventElapsedTime(&v, E[b + 2], E[b + 3])); T.allreduce_ms += v;
            if (per == 8) {      // E[b+3] solve start | E[b+4] exchange start | E[b+5] exchange end | E[b+6] sweeps' midpoint | E[b+7] solve end
                HIPCHK(hipEventElapsedTime(&v, E[b + 3], E[b + 4])); T.solve_fwd_ms += v;
                HIPCHK(hipEventElapsedTime(&v, E[b + 4], E[b + 5])); T.allreduce_ms += v;
                HIPCHK(hipEventElapsedTime(&v, E[b + 5], E[b + 6])); T.solve_fwd_ms += v;
                HIPCHK(hipEventElapsedTime(&v, E[b + 6], E[b + 7])); T.solve_bwd_ms += v;
            } else {
                HIPCHK(hipEventElapsedTime(&v, E[b + 3], E[b + 4])); T.solve_fwd_ms += v;
                HIPCHK(hipEventElapsedTime(&v, E[b + 4], E[b + 5])); T.solve_bwd_ms += v;
            }
        }
        HIPCHK(hipEventElapsedTime(&v, E[need - 2], E[need - 1])); T.epilogue_ms = v;
        HIPCHK(hipEventElapsedTime(&v, E[0], E[need - 1])); T.total_ms = v;
        if (ctx->ev_timed > 0 && ctx->ev_timed != ctx->ev_iters) {      // sampled: phase sums scaled to the whole frame (total_ms is the real span)
            const float sc = (float)ctx->ev_iters / (float)ctx->ev_timed;
            T.local_ms *= sc; T.rhs_ms *= sc; T.allreduce_ms *= sc; T.solve_fwd_ms *= sc; T.solve_bwd_ms *= sc;
        }
        T.iters = ctx->ev_iters;
        ctx->last_timing = T;
        ctx->ev_pending = false;
    }
    *t = ctx->last_timing;
    return ADMM_OK;
}

} // extern "C"
