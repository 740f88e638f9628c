// factor.cpp -- see factor.hpp.
#include "factor.hpp"
#include "dense.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <omp.h>

namespace admm_host {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void build_symcsc(int n, const std::vector<int> &ti, const std::vector<int> &tj, const std::vector<double> &tv, SymCSC &A) {
    A.n = n;
    const size_t nt = ti.size();
    std::vector<int64_t> cnt(n + 1, 0);
    for (size_t t = 0; t < nt; ++t) cnt[tj[t] + 1]++;
    for (int j = 0; j < n; ++j) cnt[j + 1] += cnt[j];
    std::vector<int> ri(nt);
    std::vector<double> rv(nt);
    {
        std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
        for (size_t t = 0; t < nt; ++t) { int64_t p = pos[tj[t]]++; ri[p] = ti[t]; rv[p] = tv[t]; }
    }
    A.ptr.assign(n + 1, 0);
    A.idx.clear(); A.val.clear();
    A.idx.reserve(nt / 4 + n); A.val.reserve(nt / 4 + n);
    std::vector<int> ord;
    for (int j = 0; j < n; ++j) {
        int64_t b = cnt[j], e = cnt[j + 1];
        ord.resize(e - b);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return ri[b + a] < ri[b + c]; });
        for (size_t q = 0; q < ord.size();) {
            int row = ri[b + ord[q]];
            double s = 0.0;
            while (q < ord.size() && ri[b + ord[q]] == row) { s += rv[b + ord[q]]; ++q; }
            A.idx.push_back(row); A.val.push_back(s);
        }
        A.ptr[j + 1] = (int64_t)A.idx.size();
    }
}

void sym_apply(const SymCSC &A, const double *x, double *y) {
    const int n = A.n;
    std::fill(y, y + 3 * (size_t)n, 0.0);
    for (int j = 0; j < n; ++j)
        for (int64_t p = A.ptr[j]; p < A.ptr[j + 1]; ++p) {
            int i = A.idx[p]; double v = A.val[p];
            for (int c = 0; c < 3; ++c) y[3 * (size_t)i + c] += v * x[3 * (size_t)j + c];
            if (i != j) for (int c = 0; c < 3; ++c) y[3 * (size_t)j + c] += v * x[3 * (size_t)i + c];
        }
}

// ---------------------------------------------------------------------------
// nested dissection
// ---------------------------------------------------------------------------
namespace {
struct ND {
    const std::vector<int64_t> *adjp; const std::vector<int> *adj; const double *xyz;
    int leaf;
    std::vector<int> tag;     // scratch region tag per node
    int next_tag = 1;
    std::vector<int> order;   // new -> old
    std::vector<Supernode> sn;

    int emit(const std::vector<int> &nodes) {
        Supernode s; s.first = (int)order.size(); s.ncols = (int)nodes.size();
        order.insert(order.end(), nodes.begin(), nodes.end());
        sn.push_back(s);
        return (int)sn.size() - 1;
    }

    // One bisection: median split of `nodes` along their longest axis; the smaller of the two boundary sets becomes the
    // separator.  L, R: the two sides without the separator; sep sorted along the axis.
    void bisect(std::vector<int> &nodes, std::vector<int> &L, std::vector<int> &R, std::vector<int> &sepc) {
        const int m = (int)nodes.size();
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int v : nodes) for (int c = 0; c < 3; ++c) { double q = xyz[3 * (size_t)v + c]; lo[c] = std::min(lo[c], q); hi[c] = std::max(hi[c], q); }
        int ax = 0; double ext = hi[0] - lo[0];
        for (int c = 1; c < 3; ++c) if (hi[c] - lo[c] > ext) { ext = hi[c] - lo[c]; ax = c; }
        const int half = m / 2;
        auto cmp = [&](int a, int b) { double qa = xyz[3 * (size_t)a + ax], qb = xyz[3 * (size_t)b + ax]; return qa < qb || (qa == qb && a < b); };
        std::nth_element(nodes.begin(), nodes.begin() + half, nodes.end(), cmp);
        const int tl = next_tag++, tr = next_tag++;
        for (int i = 0; i < half; ++i) tag[nodes[i]] = tl;
        for (int i = half; i < m; ++i) tag[nodes[i]] = tr;
        std::vector<int> bl, br;
        for (int i = 0; i < m; ++i) {
            int v = nodes[i]; const int other = (i < half) ? tr : tl;
            bool b = false;
            for (int64_t p = (*adjp)[v]; p < (*adjp)[v + 1]; ++p) if (tag[(*adj)[p]] == other) { b = true; break; }
            if (b) (i < half ? bl : br).push_back(v);
        }
        const bool sep_right = br.size() <= bl.size();
        std::vector<int> &sep = sep_right ? br : bl;
        if (sep.empty()) sep.push_back(nodes[m - 1]); // disconnected halves: any node roots the subtree
        const int ts = next_tag++;
        for (int v : sep) tag[v] = ts;
        L.clear(); R.clear();
        L.reserve(half); R.reserve(m - half);
        for (int i = 0; i < m; ++i) { int v = nodes[i]; if (tag[v] == tl) L.push_back(v); else if (tag[v] == tr) R.push_back(v); }
        sepc = sep;
        std::sort(sepc.begin(), sepc.end(), cmp);
        { std::vector<int>().swap(nodes); }
    }

    // Regions of more than `merge` nodes become FOUR-way tree nodes: the region's separator and the separators of its two
    // halves form one (dense) supernode with the four quarters as children -- half as many elimination-tree levels for the
    // zero blocks between the two half-separators.
    // merge_root: only the whole system's region does so -- its separator and the two half-separators become ONE root
    // supernode, which the GPU solves as a single dense product with its explicit inverse (solve_root kernels): the two
    // top levels of both sweeps, where every launch is latency, collapse into one product.
    int merge = 0;
    bool merge_root = false;
    int merge_small = 0;      // regions of at most that many nodes (and more than a leaf) become four-way nodes as well: one level less near the leaves
    int root_depth = 0;       // > 1: bisection levels the ROOT node spans (whatever the other nodes do)
    bool root_exact = false;  // root_depth >= 1 is binding: the root has exactly 2^root_depth children (subtree sharding's distributed top: one subtree per rank)
    int merge_depth = 2;      // bisection levels a merged node spans: 2 = four-way (three separators in one supernode), 3 = eight-way (seven)
    // the part `H` of a merged node: its separators down to `d` more bisections join `cols`, what is left below becomes children
    void gather(std::vector<int> &H, int d, int depth, std::vector<int> &kids, std::vector<int> &cols) {
        if (H.empty()) return;
        if ((int)H.size() <= leaf || d == 0) { kids.push_back(rec(H, depth)); return; }
        std::vector<int> a, b, sh;
        bisect(H, a, b, sh);
        gather(a, d - 1, depth + 1, kids, cols);
        gather(b, d - 1, depth + 1, kids, cols);
        cols.insert(cols.end(), sh.begin(), sh.end());
    }
    int rec(std::vector<int> &nodes, int depth = 0) {
        const int m = (int)nodes.size();
        if (m <= leaf) return emit(nodes);
        bool four = (merge > 0 && m > merge) || (merge_root && depth == 0) || (merge_small > 0 && m <= merge_small) || (root_depth > 1 && depth == 0);
        if (root_exact && depth == 0) four = root_depth > 1;      // the root spans EXACTLY root_depth bisection levels (1: a plain binary root), whatever the merge rules say
        std::vector<int> L, R, sep;
        bisect(nodes, L, R, sep);
        // the merged root's explicit inverse is k x k with k ~ 3 separators: only while that stays a modest stream (<= 134 MB)
        if (four && merge_root && depth == 0 && !(merge > 0 && m > merge) && 3 * sep.size() > 4096) four = false;
        std::vector<int> kids;
        std::vector<int> cols;
        if (four) {
            // the merged node's own root (merge_root alone) stays four-way: its explicit inverse is sized for three separators
            int d = ((merge > 0 && m > merge) || (merge_small > 0 && m <= merge_small)) ? merge_depth : 2;
            if (depth == 0 && root_depth > 1) d = root_depth;      // the root alone spans more bisection levels: it is solved as one dense product with its explicit inverse
            gather(L, d - 1, depth + 1, kids, cols);
            gather(R, d - 1, depth + 1, kids, cols);
        } else {
            if (!L.empty()) kids.push_back(rec(L, depth + 1));
            if (!R.empty()) kids.push_back(rec(R, depth + 1));
        }
        cols.insert(cols.end(), sep.begin(), sep.end());
        const int s = emit(cols);
        for (int c : kids) sn[c].parent = s;
        return s;
    }
};
} // namespace

int analyze(const SymCSC &A, const double *xyz, int leaf_size, Factor &F, int merge_above, bool merge_root, int merge_small, int merge_depth, int root_depth, bool root_exact) {
    const double t0 = now_s();
    const int n = A.n;
    F = Factor();
    F.n = n;
    // symmetric adjacency
    std::vector<int64_t> adjp(n + 1, 0);
    for (int j = 0; j < n; ++j) for (int64_t p = A.ptr[j]; p < A.ptr[j + 1]; ++p) { int i = A.idx[p]; if (i != j) { adjp[i + 1]++; adjp[j + 1]++; } }
    for (int j = 0; j < n; ++j) adjp[j + 1] += adjp[j];
    std::vector<int> adj(adjp[n]);
    {
        std::vector<int64_t> pos(adjp.begin(), adjp.end() - 1);
        for (int j = 0; j < n; ++j) for (int64_t p = A.ptr[j]; p < A.ptr[j + 1]; ++p) { int i = A.idx[p]; if (i != j) { adj[pos[i]++] = j; adj[pos[j]++] = i; } }
    }
    ND nd; nd.adjp = &adjp; nd.adj = &adj; nd.xyz = xyz; nd.leaf = std::max(1, leaf_size); nd.merge = merge_above; nd.merge_root = merge_root; nd.merge_small = merge_small; nd.merge_depth = std::max(2, merge_depth); nd.root_depth = root_depth; nd.root_exact = root_exact && root_depth >= 1;
    nd.tag.assign(n, 0); nd.order.reserve(n);
    std::vector<int> all(n);
    std::iota(all.begin(), all.end(), 0);
    if (n > 0) nd.rec(all);
    F.perm = nd.order;
    F.iperm.assign(n, -1);
    for (int i = 0; i < n; ++i) F.iperm[F.perm[i]] = i;
    F.sn = nd.sn;
    F.t_order = now_s() - t0;

    // ---- symbolic: below-row structure per supernode ------------------------
    const double t1 = now_s();
    const int ns = (int)F.sn.size();
    std::vector<std::vector<int>> children(ns);
    for (int s = 0; s < ns; ++s) if (F.sn[s].parent >= 0) children[F.sn[s].parent].push_back(s);
    std::vector<std::vector<int>> R(ns);
    std::vector<int> stamp(n, -1);
    for (int s = 0; s < ns; ++s) {
        Supernode &S = F.sn[s];
        const int last = S.first + S.ncols - 1;
        std::vector<int> &rs = R[s];
        for (int j = S.first; j <= last; ++j) {
            int v = F.perm[j];
            for (int64_t p = adjp[v]; p < adjp[v + 1]; ++p) { int ni = F.iperm[adj[p]]; if (ni > last && stamp[ni] != s) { stamp[ni] = s; rs.push_back(ni); } }
        }
        int lev = 0;
        for (int c : children[s]) {
            for (int ni : R[c]) if (ni > last && stamp[ni] != s) { stamp[ni] = s; rs.push_back(ni); }
            lev = std::max(lev, F.sn[c].level + 1);
        }
        std::sort(rs.begin(), rs.end());
        S.nrows = (int)rs.size();
        S.level = lev;
    }
    int64_t roff = 0, poff = 0, soff = 0, nnz = 0; int maxlev = 0;
    for (int s = 0; s < ns; ++s) {
        Supernode &S = F.sn[s];
        S.rows_off = roff; S.panel_off = poff; S.slot_off = soff;
        roff += S.nrows; soff += S.nrows;
        poff += (int64_t)(S.ncols + S.nrows) * S.ncols;
        nnz += (int64_t)S.ncols * (S.ncols + 1) / 2 + (int64_t)S.nrows * S.ncols;
        maxlev = std::max(maxlev, S.level);
        F.max_cols = std::max(F.max_cols, S.ncols); F.max_rows = std::max(F.max_rows, S.nrows);
    }
    F.rows.resize(roff);
    for (int s = 0; s < ns; ++s) std::copy(R[s].begin(), R[s].end(), F.rows.begin() + F.sn[s].rows_off);
    F.n_slots = soff; F.nnz_tri = nnz;
    F.levels.assign(maxlev + 1, std::vector<int>());
    for (int s = 0; s < ns; ++s) F.levels[F.sn[s].level].push_back(s);
    // multifrontal gather lists: each child's contribution row lands on one front row of its parent
    {
        int64_t foff = 0;
        for (int s = 0; s < ns; ++s) { F.sn[s].front_off = foff; foff += F.sn[s].ncols + F.sn[s].nrows; }
        F.cg_ptr.assign(foff + 1, 0);
        std::vector<int64_t> target(roff, -1);
        for (int c = 0; c < ns; ++c) {
            const Supernode &C = F.sn[c];
            if (C.parent < 0) continue;
            const Supernode &P = F.sn[C.parent];
            const int *prow = F.rows.data() + P.rows_off;
            const int plast = P.first + P.ncols - 1;
            for (int q = 0; q < C.nrows; ++q) {
                const int g = F.rows[C.rows_off + q];
                int i;
                if (g <= plast) i = g - P.first;
                else i = P.ncols + (int)(std::lower_bound(prow, prow + P.nrows, g) - prow);
                target[C.rows_off + q] = P.front_off + i;
                F.cg_ptr[P.front_off + i + 1]++;
            }
        }
        for (int64_t i = 0; i < foff; ++i) F.cg_ptr[i + 1] += F.cg_ptr[i];
        F.cg_slot.assign(F.cg_ptr[foff], 0);
        std::vector<int64_t> pos(F.cg_ptr.begin(), F.cg_ptr.end() - 1);
        for (int c = 0; c < ns; ++c) {               // ascending child index = fixed summation order
            const Supernode &C = F.sn[c];
            if (C.parent < 0) continue;
            for (int q = 0; q < C.nrows; ++q) F.cg_slot[pos[target[C.rows_off + q]]++] = (int)(C.slot_off + q);
        }
        bool four = true;
        for (int64_t i = 0; i < foff && four; ++i) four = (F.cg_ptr[i + 1] - F.cg_ptr[i]) <= 4;
        F.cg4.clear();
        if (four) {
            F.cg4.assign(4 * (size_t)foff, -1);
            for (int64_t i = 0; i < foff; ++i) for (int64_t g = F.cg_ptr[i]; g < F.cg_ptr[i + 1]; ++g) F.cg4[4 * i + (g - F.cg_ptr[i])] = F.cg_slot[g];
        }
    }
    F.panels.clear();
    F.t_symbolic = now_s() - t1;
    return 0;
}

// ---------------------------------------------------------------------------
// numeric multifrontal factorization
// ---------------------------------------------------------------------------
namespace {
// Recycles front buffers: the fronts of a multifrontal factorization are
// allocated and released thousands of times; handing them back to the OS each
// time costs more in page faults than the arithmetic.
struct FrontPool {
    struct Blk { double *p; size_t cap; };
    std::vector<Blk> free_;
    omp_lock_t lock;
    FrontPool() { omp_init_lock(&lock); }
    ~FrontPool() { for (auto &b : free_) std::free(b.p); omp_destroy_lock(&lock); }
    double *get(size_t n, size_t *cap) {
        n = std::max<size_t>(n, 1);
        double *p = nullptr; size_t c = 0;
        omp_set_lock(&lock);
        int best = -1;
        for (int i = 0; i < (int)free_.size(); ++i)
            if (free_[i].cap >= n && (best < 0 || free_[i].cap < free_[best].cap)) best = i;
        if (best >= 0 && free_[best].cap <= 4 * n + 4096) { p = free_[best].p; c = free_[best].cap; free_[best] = free_.back(); free_.pop_back(); }
        omp_unset_lock(&lock);
        if (!p) { c = n + n / 8; p = (double *)std::malloc(c * sizeof(double)); }
        std::memset(p, 0, n * sizeof(double));
        *cap = c;
        return p;
    }
    void put(double *p, size_t cap) { omp_set_lock(&lock); free_.push_back({p, cap}); omp_unset_lock(&lock); }
};

struct Numeric {
    FrontPool pool;
    std::vector<size_t> front_cap;
    const Factor *F; const SymCSC *PA; // permuted lower CSC
    std::vector<std::vector<int>> children;
    std::vector<double *> front;       // live fronts (f x f), freed by the parent
    std::vector<double> *panels;
    int fail = 0;
};

// process one supernode; loc is a thread-private n-sized scratch (-1 outside use)
static void do_front(Numeric &N, int s, std::vector<int> &loc, int threads) {
    const Factor &F = *N.F; const Supernode &S = F.sn[s];
    const int k = S.ncols, r = S.nrows, f = k + r;
    const int *rows = F.rows.data() + S.rows_off;
    size_t fcap = 0;
    double *Fm = N.pool.get((size_t)f * f, &fcap);
    for (int j = 0; j < k; ++j) loc[S.first + j] = j;
    for (int q = 0; q < r; ++q) loc[rows[q]] = k + q;
    // original entries of the supernode's columns
    for (int j = 0; j < k; ++j) {
        const int col = S.first + j;
        for (int64_t p = N.PA->ptr[col]; p < N.PA->ptr[col + 1]; ++p) Fm[loc[N.PA->idx[p]] + (size_t)f * j] += N.PA->val[p];
    }
    // extend-add the children's update matrices
    for (int c : N.children[s]) {
        const Supernode &C = F.sn[c];
        const int kc = C.ncols, rc = C.nrows, fc = kc + rc;
        const int *crow = F.rows.data() + C.rows_off;
        const double *U = N.front[c];
        if (U) {
            for (int b = 0; b < rc; ++b) {
                const int lb = loc[crow[b]];
                const double *ucol = U + (size_t)(kc + b) * fc + kc;
                double *dst = Fm + (size_t)f * lb;
                for (int a = b; a < rc; ++a) dst[loc[crow[a]]] += ucol[a];
            }
            N.pool.put(N.front[c], N.front_cap[c]); N.front[c] = nullptr;
        }
    }
    for (int j = 0; j < k; ++j) loc[S.first + j] = -1;
    for (int q = 0; q < r; ++q) loc[rows[q]] = -1;
    int err = partial_cholesky(f, k, Fm, f, threads);
    if (err) {
#pragma omp atomic write
        N.fail = s + 1;
    }
    // panel = [L11^-1 ; L21 L11^-1]
    double *P = N.panels->data() + S.panel_off;
    if (!err) {
        trtri_lower(k, Fm, f, P, f, threads);
        trmm_right_lower(r, k, Fm + k, f, P, f, P + k, f, threads);
    }
    if (r > 0 && S.parent >= 0) { N.front[s] = Fm; N.front_cap[s] = fcap; } else { N.pool.put(Fm, fcap); N.front[s] = nullptr; }
}

static void do_subtree(Numeric &N, int root, std::vector<int> &loc, std::vector<int> &stack) {
    // iterative postorder: indices are already postorder, a subtree is the
    // contiguous index range [first_desc, root]
    int lo = root;
    stack.clear(); stack.push_back(root);
    while (!stack.empty()) { int s = stack.back(); stack.pop_back(); lo = std::min(lo, s); for (int c : N.children[s]) stack.push_back(c); }
    for (int s = lo; s <= root; ++s) do_front(N, s, loc, 1);
}
} // namespace

void permuted_lower(const SymCSC &A, const Factor &F, SymCSC &PA, bool with_source) {
    const int n = A.n;
    PA = SymCSC(); PA.n = n;
    std::vector<int> ti, tj; std::vector<double> tv;
    ti.reserve(A.idx.size()); tj.reserve(A.idx.size()); tv.reserve(A.idx.size());
    for (int j = 0; j < n; ++j) for (int64_t p = A.ptr[j]; p < A.ptr[j + 1]; ++p) {
        int a = F.iperm[A.idx[p]], b = F.iperm[j];
        ti.push_back(std::max(a, b)); tj.push_back(std::min(a, b)); tv.push_back(with_source ? (double)p : A.val[p]);
    }
    build_symcsc(n, ti, tj, tv, PA);
}

void plan_panels(Factor &F) {
    int64_t size = 0;
    for (Supernode &S : F.sn) size += (int64_t)(S.ncols + S.nrows) * S.ncols;
    for (Supernode &S : F.sn) {
        S.root_inv_off = -1;
        if (S.parent >= 0 || S.nrows != 0 || S.ncols <= F.root_inv_min_cols) continue;
        const int64_t off = (size + 15) & ~(int64_t)15;
        S.root_inv_off = off;
        size = off + (int64_t)root_inv_ld(S.ncols) * S.ncols;
    }
    F.panels_size = size;
}

int factorize(const SymCSC &A, Factor &F, int threads) {
    const double t0 = now_s();
    const int n = A.n, ns = (int)F.sn.size();
    if (threads < 1) threads = 1;
    // permuted lower CSC
    SymCSC PA;
    permuted_lower(A, F, PA, false);
    int64_t ptot = 0;
    for (int s = 0; s < ns; ++s) ptot += (int64_t)(F.sn[s].ncols + F.sn[s].nrows) * F.sn[s].ncols;
    F.panels.assign(ptot, 0.0);

    Numeric N; N.F = &F; N.PA = &PA; N.panels = &F.panels;
    N.children.assign(ns, std::vector<int>());
    for (int s = 0; s < ns; ++s) if (F.sn[s].parent >= 0) N.children[F.sn[s].parent].push_back(s);
    N.front.assign(ns, nullptr);
    N.front_cap.assign(ns, 0);

    // split the tree: "top" supernodes (processed one by one with threaded dense
    // kernels) and independent subtrees below them (processed in parallel).
    std::vector<double> work(ns, 0.0);
    for (int s = 0; s < ns; ++s) {
        double k = F.sn[s].ncols, r = F.sn[s].nrows;
        work[s] += k * k * k / 3 + k * k * r + k * r * r + k * k * k / 6 + r * k * k;
        if (F.sn[s].parent >= 0) work[F.sn[s].parent] += work[s];
    }
    std::vector<char> is_top(ns, 0);
    std::vector<int> tasks;
    for (int s = 0; s < ns; ++s) if (F.sn[s].parent < 0) tasks.push_back(s);
    if (threads > 1) {
        const size_t want = (size_t)threads * 8;
        while (tasks.size() < want) {
            int bi = -1; double bw = -1;
            for (size_t t = 0; t < tasks.size(); ++t) if (!N.children[tasks[t]].empty() && work[tasks[t]] > bw) { bw = work[tasks[t]]; bi = (int)t; }
            if (bi < 0) break;
            int s = tasks[bi];
            is_top[s] = 1;
            tasks.erase(tasks.begin() + bi);
            for (int c : N.children[s]) tasks.push_back(c);
        }
    }
    std::sort(tasks.begin(), tasks.end(), [&](int a, int b) { return work[a] > work[b]; });
#pragma omp parallel num_threads(threads)
    {
        std::vector<int> loc(n, -1), stack;
#pragma omp for schedule(dynamic, 1)
        for (int t = 0; t < (int)tasks.size(); ++t) do_subtree(N, tasks[t], loc, stack);
    }
    {
        std::vector<int> loc(n, -1);
        for (int s = 0; s < ns; ++s) if (is_top[s]) do_front(N, s, loc, (F.sn[s].ncols + F.sn[s].nrows >= 384) ? threads : 1);
    }
    for (int s = 0; s < ns; ++s) if (N.front[s]) { N.pool.put(N.front[s], N.front_cap[s]); N.front[s] = nullptr; }
    // A root's two sweeps are y = L^-1 t followed at once by x = L^-T y: the GPU does them as ONE dense product with
    // (L L^T)^-1 = L^-T L^-1 (k x k, symmetric, stored full) -- one launch instead of two at the top of the tree, where every
    // launch is latency.  Appended to the panels (so uploads and re-factorizations carry it along).
    if (!N.fail) {
        for (int s = 0; s < ns; ++s) {
            Supernode &S = F.sn[s];
            S.root_inv_off = -1;
            if (S.parent >= 0 || S.nrows != 0 || S.ncols <= F.root_inv_min_cols) continue;
            const int k = S.ncols, ld = root_inv_ld(k);
            const size_t off = (F.panels.size() + 15) & ~(size_t)15;
            F.panels.resize(off + (size_t)ld * k, 0.0);
            const double *Li = F.panels.data() + S.panel_off;      // L^-1, lower triangular, ld = k
            double *Si = F.panels.data() + off;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
            for (int j = 0; j < k; ++j)
                for (int i = j; i < k; ++i) {                      // S^-1(i, j) = sum_{m >= i} L^-1(m, i) L^-1(m, j)
                    double acc = 0.0;
                    const double *ci = Li + (size_t)k * i, *cj = Li + (size_t)k * j;
                    for (int m = i; m < k; ++m) acc += ci[m] * cj[m];
                    Si[i + (size_t)ld * j] = acc; Si[j + (size_t)ld * i] = acc;
                }
            S.root_inv_off = (int64_t)off;
        }
    }
    F.panels_size = (int64_t)F.panels.size();
    F.t_numeric = now_s() - t0;
    return N.fail;
}

void panel_solve_host(const Factor &F, const double *b, double *x) {
    const int n = F.n, ns = (int)F.sn.size();
    std::vector<double> y(3 * (size_t)n), w(3 * (size_t)n), xs(3 * (size_t)n), C(3 * (size_t)std::max<int64_t>(F.n_slots, 1));
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) y[3 * (size_t)i + c] = b[3 * (size_t)F.perm[i] + c];
    for (int s = 0; s < ns; ++s) {
        const Supernode &S = F.sn[s];
        const int k = S.ncols, r = S.nrows, f = k + r;
        const double *P = F.panels.data() + S.panel_off;
        for (int j = 0; j < k; ++j) {
            const int col = S.first + j;
            for (int c = 0; c < 3; ++c) {
                double t = y[3 * (size_t)col + c];
                for (int64_t g = F.cg_ptr[S.front_off + j]; g < F.cg_ptr[S.front_off + j + 1]; ++g) t -= C[3 * (size_t)F.cg_slot[g] + c];
                y[3 * (size_t)col + c] = t;
            }
        }
        for (int i = 0; i < f; ++i) {
            double acc[3] = {0, 0, 0};
            const int jmax = i < k ? i : k - 1;
            for (int j = 0; j <= jmax; ++j) { double p = P[i + (size_t)f * j]; for (int c = 0; c < 3; ++c) acc[c] += p * y[3 * (size_t)(S.first + j) + c]; }
            if (i < k) for (int c = 0; c < 3; ++c) w[3 * (size_t)(S.first + i) + c] = acc[c];
            else {
                for (int64_t g = F.cg_ptr[S.front_off + i]; g < F.cg_ptr[S.front_off + i + 1]; ++g) for (int c = 0; c < 3; ++c) acc[c] += C[3 * (size_t)F.cg_slot[g] + c];
                for (int c = 0; c < 3; ++c) C[3 * (size_t)(S.slot_off + i - k) + c] = acc[c];
            }
        }
    }
    for (int s = ns - 1; s >= 0; --s) {
        const Supernode &S = F.sn[s];
        const int k = S.ncols, r = S.nrows, f = k + r;
        const double *P = F.panels.data() + S.panel_off;
        const int *rows = F.rows.data() + S.rows_off;
        for (int j = 0; j < k; ++j) {
            double acc[3] = {0, 0, 0};
            for (int i = j; i < k; ++i) { double p = P[i + (size_t)f * j]; for (int c = 0; c < 3; ++c) acc[c] += p * w[3 * (size_t)(S.first + i) + c]; }
            for (int q = 0; q < r; ++q) { double p = P[k + q + (size_t)f * j]; for (int c = 0; c < 3; ++c) acc[c] -= p * xs[3 * (size_t)rows[q] + c]; }
            for (int c = 0; c < 3; ++c) xs[3 * (size_t)(S.first + j) + c] = acc[c];
        }
    }
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) x[3 * (size_t)F.perm[i] + c] = xs[3 * (size_t)i + c];
}

} // namespace admm_host
