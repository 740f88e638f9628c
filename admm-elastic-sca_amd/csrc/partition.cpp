// partition.cpp -- subtree sharding (who owns which supernode / element) and the XCD-aware order of a level's work items.
#include "ctx.hpp"

using namespace admm_host;
using namespace admm_lib;

namespace admm_lib {

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
    // (ADMM_HIP_SUBTREES_PER_RANK = 2 / 3 / 4: at least that many subtrees per rank before the balance test may stop the splitting -- more,
    // smaller subtrees mix cheap and expensive regions of the mesh on every rank at the price of a larger replicated top; measured in
    // profiles/r04/subtrees_per_rank.txt, default 1)
    int min_per_rank = 1;
    if (const char *e = getenv("ADMM_HIP_SUBTREES_PER_RANK")) min_per_rank = std::min(4, std::max(1, atoi(e)));
    while (!open.empty()) {
        const int have = (int)(open.size() + done.size());
        if (have >= 4 * world || (have >= min_per_rank * world && assign() <= 1.15)) break;
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
    if (ctx->shard_mode != 1 || world <= 1) return;
    std::vector<double> load; int nt = 0; size_t nsub = 0;
    if (ctx->dist_top) {      // distributed top: the root is the top, its children (one per rank, host_factor checked) are the ranks' subtrees in tree order
        ctx->sn_owner.assign(ns, -1); load.assign(world, 0.0);
        int next = 0;
        std::vector<int> rank_of(ns, -1);
        int n_kids = 0;
        for (int s = 0; s < ns; ++s) if (F.sn[s].parent == ns - 1) ++n_kids;
        const int per_rank = std::max(1, n_kids / world);      // (a power of two of subtrees, host_factor checked: 2 ranks hold two each, neighbours in the dissection)
        for (int s = 0; s < ns; ++s) if (F.sn[s].parent == ns - 1) rank_of[s] = std::min(world - 1, next++ / per_rank);      // the root's children, in ascending supernode order
        for (int s = ns - 2; s >= 0; --s) {      // parents before children
            ctx->sn_owner[s] = F.sn[s].parent == ns - 1 ? rank_of[s] : ctx->sn_owner[F.sn[s].parent];
            load[ctx->sn_owner[s]] += (double)(F.sn[s].ncols + F.sn[s].nrows) * F.sn[s].ncols;
        }
        nt = 1; nsub = (size_t)n_kids;
    } else
    subtree_owners(F, world, ctx->sn_owner, load, nt, nsub);
    for (int s = 0; s < ns; ++s) for (int j = 0; j < F.sn[s].ncols; ++j) ctx->node_owner[F.sn[s].first + j] = ctx->sn_owner[s];
    if (getenv("ADMM_HIP_VERBOSE")) {
        fprintf(stderr, "admm_hip: subtree sharding: %d top supernodes, %zu subtrees, load per rank (1e6 entries):", nt, nsub);
        for (double l : load) fprintf(stderr, " %.1f", l * 1e-6);
        fprintf(stderr, "\n");
    }
}

// Subtree sharding: which top supernodes' x does a rank need?  The forward sweep over the top is complete on every rank (it sums
// towards the root), but the backward sweep only has to reach the separators a rank touches: those that appear among the front rows
// of its own subtrees, those its elements have a node in, and their ancestors.  Everybody evaluates the rule for EVERY rank (no
// communication): need[r][s] per rank and supernode; provider[s] = the lowest rank that computes top supernode s (its x goes into
// the per-frame assembly of the full vector from that rank).
void top_needs(const admm_hip_ctx *ctx, std::vector<std::vector<char> > &need, std::vector<int> &provider) {
    const Factor &F = ctx->F;
    const int ns = (int)F.sn.size();
    std::vector<int> sn_of(F.n, -1);
    for (int s2 = 0; s2 < ns; ++s2) for (int j = 0; j < F.sn[s2].ncols; ++j) sn_of[F.sn[s2].first + j] = s2;
    need.assign(ctx->world, std::vector<char>(ns, 0));
    provider.assign(ns, 0);
    if (ctx->top_bwd_needed_only) {
        for (int s2 = 0; s2 < ns; ++s2) {
            const int o = ctx->sn_owner[s2];
            if (o < 0) continue;
            for (int q = 0; q < F.sn[s2].nrows; ++q) { const int t = sn_of[F.rows[F.sn[s2].rows_off + q]]; if (ctx->sn_owner[t] < 0) need[o][t] = 1; }
        }
        for (const Batch &b : ctx->batches) for (int e = 0; e < b.n_total; ++e) {
            const int32_t *nd; const int nn = b.elem_nodes(e, &nd);
            int o = -1;
            for (int c = 0; c < nn && o < 0; ++c) o = ctx->node_owner[F.iperm[nd[c]]];
            if (o < 0) o = e % ctx->world;      // the rule of assign_elements
            for (int c = 0; c < nn; ++c) { const int i = F.iperm[nd[c]]; if (ctx->node_owner[i] < 0) need[o][sn_of[i]] = 1; }
        }
        for (int r = 0; r < ctx->world; ++r) for (int s2 = 0; s2 < ns; ++s2)      // postorder: a supernode's parent has a larger index
            if (need[r][s2] && F.sn[s2].parent >= 0) need[r][F.sn[s2].parent] = 1;
    } else for (int r = 0; r < ctx->world; ++r) std::fill(need[r].begin(), need[r].end(), 1);
    for (int s2 = 0; s2 < ns; ++s2) {
        if (ctx->sn_owner[s2] >= 0) continue;
        int p = -1;
        for (int r = 0; r < ctx->world && p < 0; ++r) if (need[r][s2]) p = r;
        if (p < 0) { p = 0; need[0][s2] = 1; }      // nobody touches it: rank 0 keeps it
        provider[s2] = p;
    }
}

// admm_hip_info's sharding figures: the factor entries this rank's sweeps stream (own supernodes, replicated top forward / backward),
// its nodes, and the doubles it all-reduces per ADMM iteration and per frame -- what bench.py's N > 1 roofline is priced on
void shard_accounting(admm_hip_ctx *ctx) {
    const Factor &F = ctx->F;
    const int ns = (int)F.sn.size();
    admm_hip_info &I = ctx->info;
    auto entries = [&](int s) { const Supernode &S = F.sn[s]; return (int64_t)S.ncols * (S.ncols + 1) / 2 + (int64_t)S.nrows * S.ncols; };
    I.sweep_entries_own = I.sweep_entries_top = I.sweep_entries_top_bwd = 0; I.nodes_own = I.nodes_top = 0;
    I.comm_doubles_iter = I.comm_doubles_frame = 0;
    const bool subtree = ctx->shard_mode == 1 && ctx->world > 1;
    if (!subtree) {
        for (int s = 0; s < ns; ++s) I.sweep_entries_own += entries(s);
        I.nodes_own = F.n;
        if (ctx->world > 1) I.comm_doubles_iter = 3 * (int64_t)F.n;      // contiguous shards: the whole right-hand side, every iteration
        return;
    }
    std::vector<std::vector<char> > need; std::vector<int> provider;
    top_needs(ctx, need, provider);
    int64_t slots = 0;
    for (int s = 0; s < ns; ++s) {
        const int o = ctx->sn_owner[s];
        if (o == ctx->rank) { I.sweep_entries_own += entries(s); I.nodes_own += F.sn[s].ncols; }
        else if (o < 0 && ctx->dist_top) { I.sweep_entries_top += (int64_t)(ctx->root_r1 - ctx->root_r0) * F.sn[s].ncols; I.nodes_top += F.sn[s].ncols; }      // this rank's rows of the root's inverse, once per iteration
        else if (o < 0) { I.sweep_entries_top += entries(s); I.nodes_top += F.sn[s].ncols; if (need[ctx->rank][s]) I.sweep_entries_top_bwd += entries(s); }
        const int par = F.sn[s].parent;
        if (o >= 0 && par >= 0 && ctx->sn_owner[par] < 0) slots += F.sn[s].nrows;      // roots of the owned subtrees: their contribution rows feed the top
    }
    I.comm_doubles_iter = 3 * (I.nodes_top + slots);      // [partial RHS on the top nodes | subtree roots' contribution rows]
    if (ctx->dist_top) I.comm_doubles_iter += 3 * I.nodes_top;      // + the second collective: every rank's rows of the top's x
    I.comm_doubles_frame = 3 * (int64_t)F.n;               // the full x, once per frame (shard_sync_x)
}

// The device's panel layout (after partition_subtrees): which supernodes' panels live on THIS rank's device and where.  One rank,
// contiguous shards, factor_local off or no transport installed yet: all of them, in the host layout.  Rank-local (subtree shards): the
// rank's own supernodes and the replicated top, packed in supernode order, the resident roots' explicit inverses behind them.
void plan_device_panels(admm_hip_ctx *ctx) {
    const Factor &F = ctx->F;
    const int ns = (int)F.sn.size();
    // (ADMM_HIP_PLAN_AS_IF_DEVICE: the CPU tests' host-only contexts plan the sharding as a context with a device and a transport would)
    ctx->factor_local_on = ctx->factor_local && ctx->shard_mode == 1 && ctx->world > 1 && ((ctx->device_id >= 0 && (ctx->rccl_comm || ctx->allreduce)) || getenv("ADMM_HIP_PLAN_AS_IF_DEVICE"));
    ctx->dev_panel_off.assign(ns, -1); ctx->dev_root_inv_off.assign(ns, -1);
    ctx->root_sn = -1;
    if (!ctx->factor_local_on) ctx->dist_top = false;      // (host_factor asked the same questions; a small system solved densely switched the sharding off since)
    if (!ctx->factor_local_on) {
        for (int s = 0; s < ns; ++s) { ctx->dev_panel_off[s] = F.sn[s].panel_off; ctx->dev_root_inv_off[s] = F.sn[s].root_inv_off; }
        ctx->dev_panels_size = F.panels_size;
    } else {
        int64_t size = 0;
        ctx->root_sn = -1;
        auto resident = [&](int s) { return ctx->sn_owner[s] == ctx->rank || ctx->sn_owner[s] < 0; };
        // distributed top: of the root this rank keeps rows [r0, r1) of the explicit inverse and nothing else (its L^-1 and the whole inverse are
        // temporaries of the factorization); slices start on multiples of 16 rows
        if (ctx->dist_top) {
            const int r = ns - 1, k = F.sn[r].ncols;
            auto cut = [&](int q) { return q >= ctx->world ? k : (int)(((int64_t)k * q / ctx->world) & ~(int64_t)15); };
            ctx->root_sn = r; ctx->root_k = k; ctx->root_first = F.sn[r].first; ctx->root_foff = F.sn[r].front_off;
            ctx->root_r0 = cut(ctx->rank); ctx->root_r1 = cut(ctx->rank + 1);
        }
        for (int s = 0; s < ns; ++s) if (resident(s) && s != ctx->root_sn) { ctx->dev_panel_off[s] = size; size += (int64_t)(F.sn[s].ncols + F.sn[s].nrows) * F.sn[s].ncols; }
        for (int s = 0; s < ns; ++s) {
            if (!resident(s) || F.sn[s].root_inv_off < 0) continue;
            const int64_t off = (size + 15) & ~(int64_t)15;
            ctx->dev_root_inv_off[s] = off;
            size = off + (int64_t)root_inv_ld(F.sn[s].ncols) * (s == ctx->root_sn ? ctx->root_r1 - ctx->root_r0 : F.sn[s].ncols);
        }
        ctx->dev_panels_size = size;
    }
    ctx->info.factor_local = ctx->factor_local_on ? 1 : 0;
    ctx->info.dist_top = ctx->dist_top ? 1 : 0;
    ctx->info.factor_doubles_resident = ctx->dev_panels_size;
    ctx->info.front_doubles = 0; ctx->info.factor_exchange_doubles = 0;
    if (getenv("ADMM_HIP_VERBOSE") && ctx->world > 1)
        fprintf(stderr, "admm_hip: rank %d: %s factorization, %.3f GB of the factor's %.3f GB resident\n", ctx->rank, ctx->factor_local_on ? "rank-local" : "whole (replicated)",
                ctx->dev_panels_size * 8e-9, F.panels_size * 8e-9);
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

} // namespace admm_lib
