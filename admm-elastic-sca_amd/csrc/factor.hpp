// factor.hpp -- host side of the global step: nested-dissection ordering,
// supernodal symbolic analysis and multifrontal Cholesky of the scalar system
//     A_s = M_s + dt^2 * sum_e w_e^2 G_e G_e^T          (n_nodes x n_nodes)
// (the reference factors the 3n x 3n matrix A = A_s (x) I_3 with Eigen's
// SimplicialLDLT + AMD: deps/admm-elastic-sca/src/system/System.cpp:138-140),
// and the "panel" form of the factor the GPU triangular sweeps stream:
//
//   for every supernode s with columns C_s (k of them) and below-rows R_s (r):
//       P_s = [ L_ss^-1 ; L_rs L_ss^-1 ]        ((k+r) x k, column-major)
//   forward  (levels bottom-up):  t_s = b_s - (children's contributions landing on C_s)
//                                 [w_s ; c_s] = P_s t_s ;  c_s += children's contributions
//                                 landing on R_s (multifrontal pass-through)   (c_s -> slots)
//   backward (levels top-down):   x_s = P_s^T [w_s ; -x(R_s)]
//
// Both sweeps are plain dense panel x vector products with 3 right-hand sides
// (x,y,z of a node), no intra-supernode dependency, no atomics, fixed
// summation order -> bitwise reproducible.
#pragma once
#include <cstdint>
#include <vector>

namespace admm_host {

struct SymCSC {          // lower triangle (incl. diagonal), column-major, sorted rows
    int n = 0;
    std::vector<int64_t> ptr;
    std::vector<int> idx;
    std::vector<double> val;
};

struct Supernode {
    int first = 0, ncols = 0, nrows = 0, parent = -1, level = 0;
    int64_t rows_off = 0;   // into Factor::rows
    int64_t panel_off = 0;  // into Factor::panels (doubles), ld = ncols + nrows
    int64_t slot_off = 0;   // first contribution slot
    int64_t front_off = 0;  // index of the supernode's first front row (k + r rows) in the global front-row numbering
    int64_t root_inv_off = -1; // roots with more than ROOT_INV_MIN_COLS columns: offset in Factor::panels of (L_ss L_ss^T)^-1, k x k, full (symmetric), leading dimension root_inv_ld(k)
};
constexpr int ROOT_INV_MIN_COLS = 64;
// leading dimension of a root's explicit inverse: rows start on 128-byte lines (16-byte loads per lane in root_product_kernel)
inline int root_inv_ld(int k) { return (k + 15) & ~15; }

struct Factor {
    int n = 0;
    std::vector<int> perm, iperm;          // perm[new] = old, iperm[old] = new
    std::vector<Supernode> sn;             // postorder: children before parents
    std::vector<int> rows;                 // concatenated R_s (new indices, ascending)
    std::vector<double> panels;            // concatenated P_s, then the roots' explicit inverses (Supernode::root_inv_off)
    std::vector<std::vector<int>> levels;  // supernodes per level (level 0 = leaves)
    std::vector<int64_t> cg_ptr;           // per front row (front_off[s] + i): range into cg_slot
    std::vector<int> cg_slot;              // the CHILDREN's contribution slots that land on that front row (child order)
    std::vector<int> cg4;                  // [front rows][4]: the same lists when no row has more than 4 entries (-1 = none), else empty
    int64_t n_slots = 0;                   // sum of nrows
    int64_t panels_size = 0;               // doubles in `panels` (plan_panels: the P_s, then the roots' inverses), also when the numeric phase runs on the GPU and `panels` stays empty
    int64_t nnz_tri = 0;                   // sum k(k+1)/2 + r k  (entries read per sweep)
    int max_cols = 0, max_rows = 0;
    int root_inv_min_cols = ROOT_INV_MIN_COLS;   // roots with MORE columns than this get an explicit inverse (-1: every root -- the distributed top of subtree sharding needs it whatever the size)
    double t_order = 0, t_symbolic = 0, t_numeric = 0;
};

// Triplets of the lower triangle (i >= j), duplicates summed.
void build_symcsc(int n, const std::vector<int> &ti, const std::vector<int> &tj, const std::vector<double> &tv, SymCSC &A);

// Geometric nested dissection on the graph of A using node coordinates
// xyz[n][3]; fills perm/iperm, supernodes, rows, levels, gather lists.
// merge_above > 0: regions with more nodes than that become four-way tree nodes (their separator and the two half-separators
// in one supernode): half as many elimination-tree levels for a little more fill.
// merge_root: the top region alone does (one root supernode = top separator + the two half-separators)
// merge_small > 0: regions of at most that many nodes (above the leaf size) become four-way nodes too (fewer, fatter levels near the leaves)
// merge_depth: bisection levels a merged node spans (2: four-way nodes, 3: eight-way nodes with seven separators in one supernode);
// root_depth > 1: the same for the root node alone; root_exact: the root spans exactly root_depth >= 1 bisection levels (2^root_depth children)
int analyze(const SymCSC &A, const double *xyz, int leaf_size, Factor &F, int merge_above = 0, bool merge_root = false, int merge_small = 0, int merge_depth = 2, int root_depth = 0, bool root_exact = false);

// Layout of Factor::panels from the symbolic structure alone: Supernode::root_inv_off and Factor::panels_size.
void plan_panels(Factor &F);
// A permuted into factor order (lower CSC).  with_source: val[q] = index of the entry of A it came from (as a double) instead of its value.
void permuted_lower(const SymCSC &A, const Factor &F, SymCSC &PA, bool with_source);

// Multifrontal numeric factorization; fills F.panels.  Returns 0 or a
// non-zero code when A is not positive definite.
int factorize(const SymCSC &A, Factor &F, int threads);

// CPU evaluation of the two sweeps on the panel form (used by the CPU tests to
// validate the factor; the product's solves run on the GPU).  b, x: [n][3] in
// the ORIGINAL node order.
void panel_solve_host(const Factor &F, const double *b, double *x);

// y = A x for x,y [n][3]
void sym_apply(const SymCSC &A, const double *x, double *y);

} // namespace admm_host
