// host_setup.cpp -- host part of admm_hip_finalize / recompute_weights: per-force rest data and rows (Force::initialize, get_selector),
// the scalar system A_s, ordering + symbolic analysis, and the host numeric factorization (System.cpp:98-156, 159-179).
#include "ctx.hpp"
#include "force_init.hpp"

using namespace admm_host;
using namespace admm_lib;

namespace admm_lib {

extern "C" int omp_get_max_threads(void);

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
        // (round 6, profiles/r06/merge_small_sharded_*.txt: that holds while a rank's levels are latency-bound.  Where a rank streams more than ~100 MB per level of its subtrees the
        //  four-way nodes' extra fill costs more than the levels they save -- 4M-tet bar, 2 ranks: factor 7.9 -> 9.6 GB, slowest rank's kernels 1.74 -> 1.91 ms; 16M tets: 9.6 / 5.2 / 3.0
        //  against 10.6 / 5.6 / 3.2 ms at 2 / 4 / 8 ranks -- so the tree is built with this rule first and, if its levels turn out byte-bound, again with four-way nodes next to the
        //  leaves only (regions of <= 4000 nodes): see below.)
        bool merge_small_by_rule = false;
        if (own_subtrees && merge_small == 0 && share >= 4096) { merge_small = (int)std::min<int64_t>(share * 4 / 3, 2000000000); merge_small_by_rule = true; }      // (tiny shares: a merged node that moves to the top would be a large part of the system)
        if (const char *e = getenv("ADMM_HIP_MERGE_SMALL")) { merge_small = atoi(e); merge_small_by_rule = false; }
        // (a cap on the merged node's columns instead -- 3 x the region's separator -- was measured and does not separate the cases: profiles/r06/merge_small_sharded_cap.txt)
        // eight-way nodes (seven separators in one supernode) save one more level between 6k and 30k nodes: configs[2] (10k nodes) -4 %, 26.9k -1.7 %
        // (tools/probe/env_ab.py ADMM_HIP_MERGE_DEPTH 2 3, four alternations); 47.5k nodes +2 %, 63k and above +15 %: four-way there
        int merge_depth = (!own_subtrees && ctx->n_nodes >= 6000 && ctx->n_nodes < 30000) ? 3 : 2;
        if (const char *e = getenv("ADMM_HIP_MERGE_DEPTH")) merge_depth = atoi(e);
        int root_depth = 0;
        if (const char *e = getenv("ADMM_HIP_ROOT_DEPTH")) root_depth = atoi(e);
        // Distributed top (round 6; subtree shards of 2 / 4 / 8 / 16 ranks with rank-local factorization): the separators of the first log2(world)
        // bisection levels (at least two) form ONE root supernode whose children are the ranks' subtrees, one each (2 ranks: two each).  The root is solved as a single product with its
        // explicit inverse (what merge_root does on one GPU), and that product is split by rows across the ranks -- every rank streams 1 / world of
        // the inverse and the slices meet in a second small collective -- instead of every rank sweeping a replicated top of several levels: the
        // replicated top's bytes grow with the rank count (4M-tet bar: 0.85 GB at 4 ranks, 1.66 GB of a rank's 2.4 GB at 8), the split root's shrink.
        auto build_tree = [&]() {
        ctx->dist_top = false;
        if (own_subtrees && (ctx->dist_top_wanted == 1 || (ctx->dist_top_wanted < 0 && ctx->n_nodes >= ctx->dist_top_min_nodes)) && ctx->factor_local && ctx->root_inverse && ((ctx->device_id >= 0 && (ctx->rccl_comm || ctx->allreduce)) || getenv("ADMM_HIP_PLAN_AS_IF_DEVICE")) &&
            (ctx->world & (ctx->world - 1)) == 0 && ctx->world <= 16 && ctx->n_nodes > ctx->dense_max && !getenv("ADMM_HIP_ROOT_DEPTH") && !getenv("ADMM_HIP_MERGE_ROOT")) {
            int d = 0; while ((1 << d) < ctx->world) ++d;
            d = std::max(d, 2);      // (2 ranks: four subtrees, two per rank -- the merged-root tree of rounds 2-5; one level less inside a rank's subtrees than a two-way root)
            Factor T;
            analyze(ctx->A, xyz.data(), leaf, T, merge_above, false, merge_small, merge_depth, d, true);
            const int ns = (int)T.sn.size();
            int roots = 0, kids = 0; bool ok = ns > 0;
            for (int sn = 0; sn < ns; ++sn) { if (T.sn[sn].parent < 0) ++roots; else if (T.sn[sn].parent == ns - 1) ++kids; }
            ok = ok && roots == 1 && kids == (1 << d) && T.sn[ns - 1].ncols > 0;
            // (the split product pays its second collective only where a rank's slice is real work; the explicit inverse must fit beside the fronts)
            if (ok) { ctx->F = std::move(T); ctx->F.root_inv_min_cols = -1; ctx->dist_top = true; }
            else if (getenv("ADMM_HIP_VERBOSE")) fprintf(stderr, "admm_hip: distributed top: the dissection gives %d roots / %d subtrees (wanted %d for %d ranks) -- replicated top instead\n", roots, kids, 1 << d, ctx->world);
        }
        if (!ctx->dist_top)
        analyze(ctx->A, xyz.data(), leaf, ctx->F, merge_above, merge_root, merge_small, merge_depth, root_depth);
        };
        build_tree();
        if (merge_small_by_rule) {
            // what a rank streams per level of its own subtrees, on average (the root of a distributed top is streamed as its inverse, not as a panel)
            const Factor &T = ctx->F;
            double entries = (double)T.nnz_tri;
            if (ctx->dist_top) { const Supernode &R = T.sn.back(); entries -= (double)R.ncols * (R.ncols + 1) / 2; }
            const double mb_per_level = 8e-6 * entries / std::max(1, ctx->world) / std::max<size_t>(1, T.levels.size());
            if (mb_per_level > 100.0) {
                if (getenv("ADMM_HIP_VERBOSE")) fprintf(stderr, "admm_hip: a rank streams %.0f MB per level of its subtrees: byte-bound levels -- four-way nodes only next to the leaves (merge_small %d -> 4000)\n", mb_per_level, merge_small);
                const double t_o = T.t_order, t_s = T.t_symbolic;
                merge_small = 4000;
                build_tree();
                ctx->F.t_order += t_o; ctx->F.t_symbolic += t_s;
            }
        }
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

} // namespace admm_lib
