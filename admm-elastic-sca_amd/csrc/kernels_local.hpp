// kernels_local.hpp -- ADMM local step (Force::project for every element of a
// batch) as one hand-written HIP kernel per material, gfx950.
//
// One lane = one element.  Fused into each kernel:
//   * Dx_i = D_i x      (reference System.cpp:54 materialises a 36-rows-per-tet
//                        vector with a serial sparse product; here the lane
//                        gathers its nodes' x and multiplies by the rest
//                        matrix B in registers),
//   * project()         (SVD + prox / blend; local_math.hpp),
//   * u_i += Dx_i - z_i,
//   * the element's share of the global-step right-hand side,
//       f_c = dt^2 w^2 * B(c,:) (z_i - u_i)   per corner c
//     (reference System.cpp:61: solver_dt2_Dt_Wt_W * (curr_z - curr_u)),
//     written to per-corner "force slots" that rhs_gather_kernel sums per node.
//
// Data layout (HBM, per batch; n = elements of this rank's shard):
//   idx   int32 [n][4] (one 16-byte load per lane; corners sorted by original
//         node id so that Dx accumulates in the same order as Eigen's
//         column-major sparse product)
//   rest  f64 SoA [12][n]   (B for tets, see force_init.hpp)
//   par   f64 SoA [P][n]    (per-element constructor parameters)
//   w2h2  f64 [n]           dt^2 * weight^2
//   u, z  f64 SoA [rows][n]
//   state f64 SoA [4][n]    (sigma warm start, L-BFGS init_hess)
//   fslot f64 [slots][3]    the elements' shares of the right-hand side, summed per node by rhs_gather_kernel in fixed slot order.
//         Tets: ONE slot per (64-tet block, node) -- the block's 256 corner shares are summed per node in LDS first (pos4 / bn_ptr /
//         bn_end / bn_dst below); every other kind: one slot per corner (dst, int32 [n][stride]).  Rank-major
//         (the r-th slot of node i at r * n_nodes + i) unless a few high-valence nodes would more than double the array: node-sorted then.
#pragma once
#include <hip/hip_runtime.h>
#include "local_math.hpp"
#include "dev_types.hpp"
#include "../../include/admm_kinds.h"

namespace admm_dev {

// one wave per block: a finished wave's slot refills at once instead of waiting for the slowest of a block's four (round 1: tet kernel -5 %
// against 256-thread blocks, A/B'd twice in alternation; the kernel's current figures: DESIGN section 3)
// one wave per workgroup is built in: the tet kernels' block-level RHS pre-reduction keeps one byte per corner position (pos4),
// a 256-entry LDS staging per block, and s_waitcnt in place of a workgroup barrier; track_block_sum is a wave butterfly
static_assert(LOCAL_BLOCK == 64, "the local-step kernels assume one 64-lane wave per workgroup");
#ifndef ADMM_TET_WAVES
#define ADMM_TET_WAVES 2   // min waves per SIMD requested for the tet kernels (caps VGPRs at 512/ADMM_TET_WAVES)
#endif


struct BatchDev {
    int n;                 // local elements (= the SoA arrays' stride)
    int e0, e1;            // the launch covers elements [e0, e1) (the whole batch: 0, n)
    const int *idx;        // [n][4]
    const double *rest;    // SoA [12][n]
    const double *par;     // SoA [P][n]
    const double *w2h2;    // [n]
    const double *kblend;  // [n]  stiffness * measure (blended kinds)
    const double *w2;      // [n]  weight^2
    double *u, *z;         // SoA [rows][n]
    int keep_z;            // tet kernels: store z (see project_tet_kernel's epilogue)
    double *state;         // SoA [4][n]
    int *n_iters;          // [n]
    double *fslot;         // [total incidences][3], node-sorted (see rhs_gather_kernel)
    const int *dst;        // [n][stride]: slot of every corner in its node's incidence list
    double *targets;       // anchors: [n][3]
    const int *active;     // anchors: [n]
    const double *dx_override; // parity tests: SoA [rows][n] of D_i x to use instead of the gather (NULL in production)
    // tets with block-level pre-reduction of the RHS shares (NULL: one slot per corner through dst):
    const unsigned int *pos4;        // [n] four bytes per tet: where corner c's share goes in the block's staging (corners sorted by node)
    const int *bn_ptr;               // [blocks + 1] the block's node entries
    const int *bn_dst;               // [entries] slot of the (block, node) sum
    const unsigned short *bn_end;    // [entries] end of the node's run in the staging (its start = the previous entry's end, 0 for the first)
    double *res_slots;     // residual tracking fused into the tet kernels (TRACK): per-corner shares of s = D^T W^T W (z - z_prev), same slot layout as fslot
    double *res_partial;   // ... and per 64-tet block: sum of w^2 |u_new - u_old|^2 (the block's part of |r|^2)
    int tpb;               // tets: elements per one-wave block (64; fewer in under-filled launches, the lanes beyond idle)
    const int *order;      // tets, large batches: which 64-tet block workgroup i processes (costliest blocks of the last frame first), or NULL
    unsigned int *cost;    // [blocks]: real-time ticks every block took, summed over a frame (feeds `order`), or NULL
};

ADMM_HD Mat3 mat_add(const Mat3 &a, const Mat3 &b) {
    Mat3 r;
    r.m00 = a.m00 + b.m00; r.m10 = a.m10 + b.m10; r.m20 = a.m20 + b.m20;
    r.m01 = a.m01 + b.m01; r.m11 = a.m11 + b.m11; r.m21 = a.m21 + b.m21;
    r.m02 = a.m02 + b.m02; r.m12 = a.m12 + b.m12; r.m22 = a.m22 + b.m22;
    return r;
}

// The tet kernel's once-per-iteration streams (rest data, u, z; at level 2 also parameters and warm-start state) are loaded /
// stored non-temporally so that they do not push the RHS slots (read back by the gather right after) and the leaf-level
// panels out of the caches: rhs 44 -> 36 us, forward sweep 289 -> 278 us at 1M tets (A/B'd in alternation).  The RHS slots
// themselves are stored cached; non-temporal loads IN the gather double its time (they defeat its line reuse).
#ifndef ADMM_LOCAL_NT
#define ADMM_LOCAL_NT 2
#endif
__device__ __forceinline__ double ld_stream(const double *p) {
#if ADMM_LOCAL_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(double *p, double v) {
#if ADMM_LOCAL_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---------------------------------------------------------------------------
// Tets: KIND 0 = Neo-Hookean, 1 = StVK (HyperElasticTet, TetForce.cpp:320-364),
//       2 = LinearTetStrain (:127-153), 3 = TetVolume (:173-210)
// ---------------------------------------------------------------------------
// the lane's inputs: B (stored corner order), Dx = D_i x, u
__device__ __forceinline__ void tet_load(const BatchDev &b, const double *__restrict__ x, int e, int n, double (&B)[12], Mat3 &Dx, Mat3 &u) {
    const int4 id = reinterpret_cast<const int4 *>(b.idx)[e];
    const double *x0 = x + 3 * (size_t)id.x, *x1 = x + 3 * (size_t)id.y, *x2 = x + 3 * (size_t)id.z, *x3 = x + 3 * (size_t)id.w;
    const double p0x = x0[0], p0y = x0[1], p0z = x0[2];
    const double p1x = x1[0], p1y = x1[1], p1z = x1[2];
    const double p2x = x2[0], p2y = x2[1], p2z = x2[2];
    const double p3x = x3[0], p3y = x3[1], p3z = x3[2];
#pragma unroll
    for (int i = 0; i < 12; ++i) B[i] = ld_stream(&b.rest[(size_t)i * n + e]);
    // Dx(j, r) = sum_c B(c, r) * x_c[j], accumulated from 0 in stored corner order
#define ADMM_DX(r, px0, px1, px2, px3) (((0.0 + B[0 + 4 * r] * px0) + B[1 + 4 * r] * px1) + B[2 + 4 * r] * px2) + B[3 + 4 * r] * px3
    Dx.m00 = ADMM_DX(0, p0x, p1x, p2x, p3x); Dx.m10 = ADMM_DX(0, p0y, p1y, p2y, p3y); Dx.m20 = ADMM_DX(0, p0z, p1z, p2z, p3z);
    Dx.m01 = ADMM_DX(1, p0x, p1x, p2x, p3x); Dx.m11 = ADMM_DX(1, p0y, p1y, p2y, p3y); Dx.m21 = ADMM_DX(1, p0z, p1z, p2z, p3z);
    Dx.m02 = ADMM_DX(2, p0x, p1x, p2x, p3x); Dx.m12 = ADMM_DX(2, p0y, p1y, p2y, p3y); Dx.m22 = ADMM_DX(2, p0z, p1z, p2z, p3z);
#undef ADMM_DX
    if (b.dx_override) {
        const double *o = b.dx_override;
        Dx.m00 = o[(size_t)0 * n + e]; Dx.m10 = o[(size_t)1 * n + e]; Dx.m20 = o[(size_t)2 * n + e];
        Dx.m01 = o[(size_t)3 * n + e]; Dx.m11 = o[(size_t)4 * n + e]; Dx.m21 = o[(size_t)5 * n + e];
        Dx.m02 = o[(size_t)6 * n + e]; Dx.m12 = o[(size_t)7 * n + e]; Dx.m22 = o[(size_t)8 * n + e];
    }
    u.m00 = ld_stream(&b.u[(size_t)0 * n + e]); u.m10 = ld_stream(&b.u[(size_t)1 * n + e]); u.m20 = ld_stream(&b.u[(size_t)2 * n + e]);
    u.m01 = ld_stream(&b.u[(size_t)3 * n + e]); u.m11 = ld_stream(&b.u[(size_t)4 * n + e]); u.m21 = ld_stream(&b.u[(size_t)5 * n + e]);
    u.m02 = ld_stream(&b.u[(size_t)6 * n + e]); u.m12 = ld_stream(&b.u[(size_t)7 * n + e]); u.m22 = ld_stream(&b.u[(size_t)8 * n + e]);
}

// order[] <- the blocks by descending cost (256 buckets of cost / max; the order inside a bucket is whatever the atomics give: it
// only decides when a block starts, never what it computes); cost[] <- 0.  One workgroup.
__global__ __launch_bounds__(1024) void order_by_cost_kernel(int n_blocks, unsigned int *__restrict__ cost, int *__restrict__ order) {
    __shared__ unsigned int hist[256], start[256], mx;
    const int t = threadIdx.x;
    if (t < 256) hist[t] = 0;
    if (t == 0) mx = 1;
    __syncthreads();
    unsigned int m = 0;
    for (int i = t; i < n_blocks; i += 1024) m = max(m, cost[i]);
    atomicMax(&mx, m);
    __syncthreads();
    const unsigned long long top = mx;
    for (int i = t; i < n_blocks; i += 1024) atomicAdd(&hist[255 - (unsigned int)((unsigned long long)cost[i] * 255ull / top)], 1u);
    __syncthreads();
    if (t == 0) { unsigned int acc = 0; for (int q = 0; q < 256; ++q) { start[q] = acc; acc += hist[q]; } }
    __syncthreads();
    for (int i = t; i < n_blocks; i += 1024) {
        const unsigned int q = 255 - (unsigned int)((unsigned long long)cost[i] * 255ull / top);
        order[atomicAdd(&start[q], 1u)] = i;
    }
    __syncthreads();
    for (int i = t; i < n_blocks; i += 1024) cost[i] = 0;
}

// ---------------------------------------------------------------------------
// StaticAnchor / MovingAnchor, AnchorForce.cpp:46-55, 71-89
// ---------------------------------------------------------------------------
// Residual tracking fused into a projection kernel (TRACK): the wave's sum of its lanes' w^2 |u_new - u_old|^2 in a fixed butterfly
// order -> one partial per 64-element block.  Lanes beyond the batch have returned (they form a suffix of the wave), `act` = the
// lanes still here: a partner that is gone contributes 0.
__device__ __forceinline__ void track_block_sum(double a, double *partial_slot) {
    const unsigned long long act = __ballot(1);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(a, off, 64); const bool have = (act >> ((threadIdx.x & 63) ^ off)) & 1ull; a += have ? o : 0.0; }
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)act) - 1)) *partial_slot = a;
}

template <bool TRACK>
__device__ __forceinline__ void project_anchor_elem(const BatchDev &b, const double *__restrict__ x, int e, int blk) {
    const int n = b.n;
    if (e >= b.e1) return;
    const int id = b.idx[e];
    const int ds = b.dst[e];
    const double s = b.w2h2[e];
    const bool act = b.active[e] != 0;
    const double w2 = TRACK ? b.w2[e] : 0.0;
    // every load first, every store last (a load behind a store waits for it: coordinate by coordinate the element was three round trips long,
    // and as the tail of a tet launch these blocks are the last ones to start)
    double X[3], U[3], T[3], ZP[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        X[j] = x[3 * (size_t)id + j];
        if (b.dx_override) X[j] = b.dx_override[(size_t)j * n + e];
        U[j] = b.u[(size_t)j * n + e];
        T[j] = act ? b.targets[3 * (size_t)e + j] : 0.0;
        ZP[j] = TRACK ? b.z[(size_t)j * n + e] : 0.0;
    }
    double r2 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double dx = 0.0 + 1.0 * X[j];
        if (b.dx_override) dx = X[j];
        const double u = U[j];
        double zi;
        if (act) zi = T[j];
        else { zi = dx + u; b.targets[3 * (size_t)e + j] = dx; }
        const double un = u + (dx - zi);
        if (TRACK) {      // s: the one corner's share w^2 (z - z_prev); r: w^2 |u_new - u_old|^2
            const double du = un - u;
            r2 += du * du;
            b.res_slots[3 * (size_t)ds + j] = 0.0 + 1.0 * (w2 * (zi - ZP[j]));
        }
        b.u[(size_t)j * n + e] = un; b.z[(size_t)j * n + e] = zi;
        b.fslot[3 * (size_t)ds + j] = s * (zi - un);
    }
    if (TRACK) track_block_sum(w2 * r2, &b.res_partial[blk]);
}
template <bool TRACK>
__global__ __launch_bounds__(LOCAL_BLOCK) void project_anchor_kernel(BatchDev b, const double *__restrict__ x) {
    project_anchor_elem<TRACK>(b, x, b.e0 + blockIdx.x * LOCAL_BLOCK + threadIdx.x, (int)blockIdx.x);
}

// one 64-tet block of a tet batch; lb = the block's number in the launch's range of this batch (workgroup -> block through b.order)
template <int KIND, int M, bool TRACK>
__device__ __forceinline__ void project_tet_block(const BatchDev &b, const double *__restrict__ x, const int lb, double *stage) {      // stage: (TRACK ? 6 : 3) * 256 doubles of LDS
    // TRACK (admm_hip_enable_residuals / set_tolerance): the primal and dual residuals of the element (reference: described at
    // System.cpp:64-65, paper Eq. 22-23) come out of the same registers -- r = W (Dx - z) = W (u_new - u_old), and the corners'
    // shares of s = D^T W^T W (z - z_prev) with z_prev read back from this kernel's own previous output -- instead of two
    // snapshot copies and two more passes over u and z (+20 % per iteration before).
    // Launch order by cost: a block's time depends on its slowest line search (2-4 or 20 evaluations, spatially clustered); in mesh
    // order the expensive blocks of the 1M-tet bar come last and the launch ends with a 50 us tail of a few hundred waves.  The blocks
    // that took longest in the last frame start first (order_by_cost_kernel, once per frame); results do not depend on the order.
    const int blk = b.order ? b.order[lb] : lb;
    const unsigned long long t_begin = b.cost ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int e = b.e0 + blk * b.tpb + threadIdx.x;
    const int n = b.n;
    if ((int)threadIdx.x >= b.tpb || e >= b.e1) return;
    double B[12];
    Mat3 Dx, u, F, z;
    Mat3 zp;      // TRACK: z of the previous iteration
    ADMM_PROF_T0
    ADMM_PROF_REGION(0);      // a wave of the tet kernel
    ADMM_TET_STAMP(lb, 0, __builtin_amdgcn_s_memrealtime());
    tet_load(b, x, e, n, B, Dx, u);
    F = mat_add(Dx, u);
    ADMM_PROF_TIME(0);
    if (KIND <= 1) {
#if ADMM_LOCAL_NT >= 2
        const double mu = ld_stream(&b.par[(size_t)0 * n + e]), lambda = ld_stream(&b.par[(size_t)1 * n + e]);
        const int maxIter = (int)ld_stream(&b.par[(size_t)2 * n + e]);
        double sa = ld_stream(&b.state[(size_t)0 * n + e]), sb = ld_stream(&b.state[(size_t)1 * n + e]), sc = ld_stream(&b.state[(size_t)2 * n + e]), hs = ld_stream(&b.state[(size_t)3 * n + e]);
#else
        const double mu = b.par[(size_t)0 * n + e], lambda = b.par[(size_t)1 * n + e];
        const int maxIter = (int)b.par[(size_t)2 * n + e];
        double sa = b.state[(size_t)0 * n + e], sb = b.state[(size_t)1 * n + e], sc = b.state[(size_t)2 * n + e], hs = b.state[(size_t)3 * n + e];
#endif
        int it = 0;
        if (TRACK) {
            auto load_zprev = [&]() {
#define ADMM_ZP(mm, row) zp.mm = ld_stream(&b.z[(size_t)row * n + e]);
                ADMM_ZP(m00, 0) ADMM_ZP(m10, 1) ADMM_ZP(m20, 2) ADMM_ZP(m01, 3) ADMM_ZP(m11, 4) ADMM_ZP(m21, 5) ADMM_ZP(m02, 6) ADMM_ZP(m12, 7) ADMM_ZP(m22, 8)
#undef ADMM_ZP
            };
            z = project_hyper<KIND, M>(F, mu, lambda, maxIter, sa, sb, sc, hs, it, load_zprev);
        } else
        z = project_hyper<KIND, M>(F, mu, lambda, maxIter, sa, sb, sc, hs, it);
#if ADMM_LOCAL_NT >= 2
        st_stream(&b.state[(size_t)0 * n + e], sa); st_stream(&b.state[(size_t)1 * n + e], sb); st_stream(&b.state[(size_t)2 * n + e], sc); st_stream(&b.state[(size_t)3 * n + e], hs);
#else
        b.state[(size_t)0 * n + e] = sa; b.state[(size_t)1 * n + e] = sb; b.state[(size_t)2 * n + e] = sc; b.state[(size_t)3 * n + e] = hs;
#endif
#if ADMM_NFEV_COUNT
        ADMM_TET_STAMP_MAX(lb, 2, it >> 8); it &= 255; ADMM_TET_STAMP_MAX(lb, 3, it);
#endif
        if (TRACK || b.keep_z) b.n_iters[e] = it;      // (an introspection output like z: admm_hip_read_local's n_iters)
    } else {
        const double lmin = (KIND == 3) ? b.par[(size_t)1 * n + e] : 0.0, lmax = (KIND == 3) ? b.par[(size_t)2 * n + e] : 0.0;
        const Mat3 p = project_tet_p<KIND == 3>(F, lmin, lmax);
        const double k = b.kblend[e], w2 = b.w2[e];
        const double den = w2 + k;
        z.m00 = (k * p.m00 + w2 * F.m00) / den; z.m10 = (k * p.m10 + w2 * F.m10) / den; z.m20 = (k * p.m20 + w2 * F.m20) / den;
        z.m01 = (k * p.m01 + w2 * F.m01) / den; z.m11 = (k * p.m11 + w2 * F.m11) / den; z.m21 = (k * p.m21 + w2 * F.m21) / den;
        z.m02 = (k * p.m02 + w2 * F.m02) / den; z.m12 = (k * p.m12 + w2 * F.m12) / den; z.m22 = (k * p.m22 + w2 * F.m22) / den;
    }
    ADMM_PROF_TIME(4);      // the whole projection incl. parameter / state traffic (1 + 2 + 3 are inside it)
    // (Measured and dropped: re-deriving B, Dx, u from memory here instead of keeping them live
    // across the projection saves ~20 VGPRs but not enough for a third wave per SIMD: -2 %.)
    // u += Dx - z ; q = z - u
    Mat3 q;
    Mat3 dz; double r2 = 0.0;      // TRACK only
    if (TRACK) {
        if (KIND > 1) {
#define ADMM_ZP(mm, row) zp.mm = ld_stream(&b.z[(size_t)row * n + e]);
            ADMM_ZP(m00, 0) ADMM_ZP(m10, 1) ADMM_ZP(m20, 2) ADMM_ZP(m01, 3) ADMM_ZP(m11, 4) ADMM_ZP(m21, 5) ADMM_ZP(m02, 6) ADMM_ZP(m12, 7) ADMM_ZP(m22, 8)
#undef ADMM_ZP
        }
        dz.m00 = z.m00 - zp.m00; dz.m10 = z.m10 - zp.m10; dz.m20 = z.m20 - zp.m20; dz.m01 = z.m01 - zp.m01; dz.m11 = z.m11 - zp.m11; dz.m21 = z.m21 - zp.m21;
        dz.m02 = z.m02 - zp.m02; dz.m12 = z.m12 - zp.m12; dz.m22 = z.m22 - zp.m22;
    }
    // z is an output nobody reads back in a plain frame (every project() overwrites it from Dx + u): stored only when somebody
    // asked (keep_z: tracking, the parity entry points, admm_hip_keep_z) -- 72 B per tet and iteration less otherwise
    const bool keep_z = TRACK || b.keep_z;
#define ADMM_UZ(mm, row) { const double un = u.mm + (Dx.mm - z.mm); q.mm = z.mm - un; if (TRACK) { const double du = un - u.mm; r2 += du * du; } st_stream(&b.u[(size_t)row * n + e], un); if (keep_z) st_stream(&b.z[(size_t)row * n + e], z.mm); }
    ADMM_UZ(m00, 0) ADMM_UZ(m10, 1) ADMM_UZ(m20, 2) ADMM_UZ(m01, 3) ADMM_UZ(m11, 4) ADMM_UZ(m21, 5) ADMM_UZ(m02, 6) ADMM_UZ(m12, 7) ADMM_UZ(m22, 8)
#undef ADMM_UZ
    const double s = b.w2h2[e];
    // f_c[j] = s * sum_r B(c, r) q(j, r)
    double f[12];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f[3 * c + 0] = s * ((B[c] * q.m00 + B[c + 4] * q.m01) + B[c + 8] * q.m02);
        f[3 * c + 1] = s * ((B[c] * q.m10 + B[c + 4] * q.m11) + B[c + 8] * q.m12);
        f[3 * c + 2] = s * ((B[c] * q.m20 + B[c + 4] * q.m21) + B[c + 8] * q.m22);
    }
    double g[12];      // TRACK: corner c's share of s = D^T W^T W (z - z_prev): w^2 * sum_r B(c, r) dz(:, r) (residual_dual_kernel's arithmetic, G = B)
    if (TRACK) {
        const double w2 = b.w2[e];
        const double q0[3] = {w2 * dz.m00, w2 * dz.m10, w2 * dz.m20}, q1[3] = {w2 * dz.m01, w2 * dz.m11, w2 * dz.m21}, q2[3] = {w2 * dz.m02, w2 * dz.m12, w2 * dz.m22};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[3 * c + j] = ((0.0 + B[c] * q0[j]) + B[c + 4] * q1[j]) + B[c + 8] * q2[j];
        track_block_sum(w2 * r2, &b.res_partial[blk]);      // |r|^2: this block's sum in a fixed butterfly order
    }
    if (b.pos4) {
        // Block-level pre-reduction: the 64 tets of a block touch ~45 distinct nodes with their 256 corners.  Every corner's share
        // goes to its place in the block's LDS staging (corners sorted by node, then lane, then corner: pos4), then one lane per
        // distinct node sums the node's run front to back (fixed order: deterministic) and writes ONE 24-byte slot -- 1.1 KB of slot
        // traffic per block instead of 6 KB of scattered 8-byte stores, and a gather that reads ~6 slots per node instead of ~24.
        const unsigned int p4 = b.pos4[e];
        const unsigned long long act = __ballot(1);
        const int nact = __popcll(act), lane = threadIdx.x & 63;
        const int q0 = b.bn_ptr[blk], q1 = b.bn_ptr[blk + 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = (p4 >> (8 * c)) & 255;
            stage[p] = f[3 * c]; stage[256 + p] = f[3 * c + 1]; stage[512 + p] = f[3 * c + 2];
            if (TRACK) { stage[768 + p] = g[3 * c]; stage[1024 + p] = g[3 * c + 1]; stage[1280 + p] = g[3 * c + 2]; }      // the s shares ride along in the same pass
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave per block: its own LDS writes are done, no barrier needed
        for (int i = q0 + lane; i < q1; i += nact) {
            const int k1 = b.bn_end[i], k0 = (i == q0) ? 0 : (int)b.bn_end[i - 1];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
            for (int k = k0; k < k1; ++k) {
                a0 += stage[k]; a1 += stage[256 + k]; a2 += stage[512 + k];
                if (TRACK) { g0 += stage[768 + k]; g1 += stage[1024 + k]; g2 += stage[1280 + k]; }
            }
            const size_t d = 3 * (size_t)b.bn_dst[i];
            double *o = b.fslot + d;
            o[0] = a0; o[1] = a1; o[2] = a2;
            if (TRACK) { double *r = b.res_slots + d; r[0] = g0; r[1] = g1; r[2] = g2; }
        }
    } else {
        const int4 ds = reinterpret_cast<const int4 *>(b.dst)[e];
        double *o0 = b.fslot + 3 * (size_t)ds.x, *o1 = b.fslot + 3 * (size_t)ds.y, *o2 = b.fslot + 3 * (size_t)ds.z, *o3 = b.fslot + 3 * (size_t)ds.w;
        o0[0] = f[0]; o0[1] = f[1]; o0[2] = f[2];       // (read back by rhs_gather_kernel right away: cached stores)
        o1[0] = f[3]; o1[1] = f[4]; o1[2] = f[5];
        o2[0] = f[6]; o2[1] = f[7]; o2[2] = f[8];
        o3[0] = f[9]; o3[1] = f[10]; o3[2] = f[11];
        if (TRACK) {
            const int dsl[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) { double *o = b.res_slots + 3 * (size_t)dsl[c]; o[0] = g[3 * c]; o[1] = g[3 * c + 1]; o[2] = g[3 * c + 2]; }
        }
    }
    ADMM_PROF_TIME(5);
#if ADMM_PROF_ON
    if (threadIdx.x == 0) atomicAdd(&admm_dev::g_tet_prof[16], 1ull);
#endif
    ADMM_TET_STAMP(lb, 1, __builtin_amdgcn_s_memrealtime());
    if (b.cost && threadIdx.x == 0) b.cost[blk] += (unsigned int)(__builtin_amdgcn_s_memrealtime() - t_begin);
}

template <int KIND, int M, bool TRACK = false>
#if ADMM_TET_WAVES > 0
__global__ __launch_bounds__(LOCAL_BLOCK, ADMM_TET_WAVES)
#else
__global__ __launch_bounds__(LOCAL_BLOCK)
#endif
void project_tet_kernel(BatchDev b, const double *__restrict__ x, BatchDev tail, int tail_block0) {
    // the anchors that follow a tet batch ride along as the launch's last workgroups (tail.n = 0: none): one launch and one
    // kernel boundary less per ADMM iteration (anchor kernel 4.9 us + 1.5 us between the launches at the 1M-tet bar)
    if ((int)blockIdx.x >= tail_block0) { project_anchor_elem<TRACK>(tail, x, tail.e0 + ((int)blockIdx.x - tail_block0) * LOCAL_BLOCK + threadIdx.x, (int)blockIdx.x - tail_block0); return; }
    __shared__ double stage[(TRACK ? 6 : 3) * 256];
    project_tet_block<KIND, M, TRACK>(b, x, (int)blockIdx.x, stage);
}

// ---------------------------------------------------------------------------
// CollisionForce, CollisionForce.cpp:38-70: one element per node, D = I;
// the point Dx+u is pushed out of every analytic shape it penetrates, in list
// order (CollisionFloor.hpp:51-58, CollisionSphere.hpp:50-66, CollisionCylinder.hpp:48-66)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void project_collision_block(const BatchDev &b, const double *__restrict__ x, const ShapeTable *__restrict__ shapes, const int lb) {
    const int e = b.e0 + lb * LOCAL_BLOCK + threadIdx.x;
    const int n = b.n;
    if (e >= b.e1) return;
    const int id = b.idx[e];
    const int ds = b.dst[e];      // (with the other loads: behind the u / z stores it would wait for them)
    const double s = b.w2h2[e];
    double dx[3], u[3], p[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dx[j] = 0.0 + 1.0 * x[3 * (size_t)id + j];
        if (b.dx_override) dx[j] = b.dx_override[(size_t)j * n + e];
        u[j] = b.u[(size_t)j * n + e];
        p[j] = dx[j] + u[j];
    }
    const int ns = shapes->n;
    for (int q = 0; q < ns; ++q) {
        const double c0 = shapes->par[q][0], c1 = shapes->par[q][1], c2 = shapes->par[q][2], R = shapes->par[q][3];
        const int ty = shapes->type[q];
        if (ty == ADMM_SHAPE_FLOOR) {
            if (c1 - p[1] > 0) p[1] = c1;
        } else if (ty == ADMM_SHAPE_SPHERE) {
            const double d0 = p[0] - c0, d1 = p[1] - c1, d2 = p[2] - c2;
            const double nrm = sqrt(d0 * d0 + (d1 * d1 + d2 * d2));
            if (R - nrm > 0) { p[0] = c0 + R * (d0 / nrm); p[1] = c1 + R * (d1 / nrm); p[2] = c2 + R * (d2 / nrm); }
        } else {
            const double d0 = p[0] - c0, d1 = p[1] - c1, d2 = 0.0 - 0.0;
            const double nrm = sqrt(d0 * d0 + (d1 * d1 + d2 * d2));
            if (R - nrm > 0) { const double pz = p[2]; p[0] = (c0 + R * (d0 / nrm)) + 0.0; p[1] = (c1 + R * (d1 / nrm)) + 0.0; p[2] = (0.0 + R * (d2 / nrm)) + pz; }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double un = u[j] + (dx[j] - p[j]);
        b.u[(size_t)j * n + e] = un; b.z[(size_t)j * n + e] = p[j];
        b.fslot[3 * (size_t)ds + j] = s * (p[j] - un);
    }
}

// ---------------------------------------------------------------------------
// Spring, Force.cpp:52-71   (rows: x_a - x_b)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void project_spring_block(const BatchDev &b, const double *__restrict__ x, const int lb) {
    const int e = b.e0 + lb * LOCAL_BLOCK + threadIdx.x;
    const int n = b.n;
    if (e >= b.e1) return;
    const int ia = b.idx[2 * (size_t)e], ib = b.idx[2 * (size_t)e + 1];
    const int da = b.dst[2 * (size_t)e], db = b.dst[2 * (size_t)e + 1];      // (with the other loads: behind the u / z stores they would wait for them)
    const double st = b.par[e], w2 = b.w2[e], s = b.w2h2[e], rest_length = b.rest[e];
    double dx[3], u[3], d[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double xa = x[3 * (size_t)ia + j], xb = x[3 * (size_t)ib + j];
        // column-ascending accumulation of (+1) x_a + (-1) x_b
        dx[j] = (ia < ib) ? ((0.0 + 1.0 * xa) + -1.0 * xb) : ((0.0 + -1.0 * xb) + 1.0 * xa);
        if (b.dx_override) dx[j] = b.dx_override[(size_t)j * n + e];
        u[j] = b.u[(size_t)j * n + e];
        d[j] = dx[j] + u[j];
    }
    const double nrm = sqrt(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2]));
    const double c = 1.0 / (w2 + st);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double dn = d[j] / nrm;
        if (nrm <= 0.0) dn = 0.0;
        const double p = rest_length * dn;
        const double zi = c * (st * p + w2 * d[j]);
        const double un = u[j] + (dx[j] - zi);
        b.u[(size_t)j * n + e] = un; b.z[(size_t)j * n + e] = zi;
        const double f = s * (zi - un);
        b.fslot[3 * (size_t)da + j] = f;
        b.fslot[3 * (size_t)db + j] = -f;
    }
}

// ---------------------------------------------------------------------------
// BendForce, BendForce.cpp:131-161   rows (x0-x2, x3-x2, x1-x2)
// ---------------------------------------------------------------------------
// EPL elements per lane (block = 64 * EPL consecutive elements, a lane's elements 64 apart: every access stays coalesced); same arithmetic
// per element.  EPL = 2 was meant for the one-launch local step (MULTI_EPL below: measured, no effect); every launch uses 1.
template <int EPL = 1>
__device__ __forceinline__ void project_bend_block(const BatchDev &b, const double *__restrict__ x, const int lb) {
    const int e0 = b.e0 + lb * (LOCAL_BLOCK * EPL) + threadIdx.x;
    const int n = b.n;
    if (e0 >= b.e1) return;
    // every load first, every store last: the stores to u / z may alias the loads as far as the compiler knows, and a load behind a store waits
    // for it (one in-order memory counter) -- per coordinate in turn the kernel was three dependent round trips long (31 us for 149 k hinges)
    int4 id[EPL], ds[EPL]; double a0[EPL], a1[EPL], a3[EPL], st[EPL], w2[EPL], s[EPL], U[EPL][9], X2[EPL][3], XP[EPL][3][3], DXO[EPL][9]; bool ok[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        ok[t] = e0 + 64 * t < b.e1;
        const int e = ok[t] ? e0 + 64 * t : e0;      // (a lane without a second element re-reads its first: no store)
        id[t] = reinterpret_cast<const int4 *>(b.idx)[e];
        ds[t] = reinterpret_cast<const int4 *>(b.dst)[e];
        a0[t] = b.rest[(size_t)0 * n + e]; a1[t] = b.rest[(size_t)1 * n + e]; a3[t] = b.rest[(size_t)3 * n + e];
        st[t] = b.par[e]; w2[t] = b.w2[e]; s[t] = b.w2h2[e];
#pragma unroll
        for (int q = 0; q < 9; ++q) U[t][q] = b.u[(size_t)q * n + e];
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int plus[3] = {id[t].x, id[t].w, id[t].y};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            X2[t][j] = x[3 * (size_t)id[t].z + j];
#pragma unroll
            for (int r = 0; r < 3; ++r) XP[t][r][j] = x[3 * (size_t)plus[r] + j];
        }
        if (b.dx_override) {
            const int e = ok[t] ? e0 + 64 * t : e0;
#pragma unroll
            for (int q = 0; q < 9; ++q) DXO[t][q] = b.dx_override[(size_t)q * n + e];
        }
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        if (!ok[t]) continue;
        const int e = e0 + 64 * t;
        const int plus[3] = {id[t].x, id[t].w, id[t].y};
        const double den = a0[t] * a0[t] + a3[t] * a3[t] + a1[t] * a1[t];
        const double cc = 1.0 / (w2[t] + st[t]);
        double UN[9], ZI[9], F[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double x2 = X2[t][j];
            double dx[3], u[3], d[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double xp = XP[t][r][j];
                dx[r] = (plus[r] < id[t].z) ? ((0.0 + 1.0 * xp) + -1.0 * x2) : ((0.0 + -1.0 * x2) + 1.0 * xp);
                if (b.dx_override) dx[r] = DXO[t][3 * r + j];
                u[r] = U[t][3 * r + j];
                d[r] = dx[r] + u[r];
            }
            const double lam = 2.0 * (a0[t] * d[0] + a3[t] * d[1] + a1[t] * d[2]) / den;
            const double p0 = d[0] - 0.5 * a0[t] * lam, p1 = d[1] - 0.5 * a3[t] * lam, p2 = d[2] - 0.5 * a1[t] * lam;
            const double pr[3] = {p0, p1, p2};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double zi = cc * (st[t] * pr[r] + w2[t] * d[r]);
                const double un = u[r] + (dx[r] - zi);
                UN[3 * r + j] = un; ZI[3 * r + j] = zi;
                F[r][j] = s[t] * (zi - un);
            }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) { b.u[(size_t)q * n + e] = UN[q]; b.z[(size_t)q * n + e] = ZI[q]; }
        // corners in idx order: 0 -> row block 0, 1 -> row block 2, 2 -> minus all, 3 -> row block 1
        double *o0 = b.fslot + 3 * (size_t)ds[t].x, *o1 = b.fslot + 3 * (size_t)ds[t].y, *o2 = b.fslot + 3 * (size_t)ds[t].z, *o3 = b.fslot + 3 * (size_t)ds[t].w;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            o0[j] = F[0][j]; o1[j] = F[2][j]; o2[j] = -((F[0][j] + F[1][j]) + F[2][j]); o3[j] = F[1][j];
        }
    }
}

// ---------------------------------------------------------------------------
// Triangles, TriangleForce.cpp.  MODE 0: LimitedTriangleStrain (:78-113), T = U(:,0:2) V^T of the 3x2 SVD, blended and
// strain-limited; MODE 1: TriArea (:251-295); MODE 2: FungTriangle (:227-249).  All three run the bit-exact restatement of
// Eigen's JacobiSVD<3x2> (column-pivoted Householder QR + 2x2 Jacobi, local_math.hpp svd32): bit-identical with the reference.
// ---------------------------------------------------------------------------
template <int MODE, int EPL = 1>
__device__ __forceinline__ void project_tri_block(const BatchDev &b, const double *__restrict__ x, const int lb) {
    const int e0 = b.e0 + lb * (LOCAL_BLOCK * EPL) + threadIdx.x;
    const int n = b.n;
    if (e0 >= b.e1) return;
    // all loads of the lane's EPL elements first (see project_bend_block), then element after element: arithmetic, stores
    int4 id[EPL], ds[EPL]; bool ok[EPL];
    double kb[EPL], w2e[EPL], par1[EPL], par2[EPL], par3[EPL], s[EPL], B[EPL][6], XA[EPL][3], XB[EPL][3], XC[EPL][3], U[EPL][6], DXO[EPL][6];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        ok[t] = e0 + 64 * t < b.e1;
        const int e = ok[t] ? e0 + 64 * t : e0;
        id[t] = reinterpret_cast<const int4 *>(b.idx)[e];
        ds[t] = reinterpret_cast<const int4 *>(b.dst)[e];      // (with the other loads: behind the u / z stores it would wait for them)
        // (parameters with the first burst of loads, not after the SVD: one exposed round trip less in a wave whose life is memory latency)
        kb[t] = MODE != 2 ? b.kblend[e] : 0.0; w2e[t] = MODE != 2 ? b.w2[e] : 0.0;
        par1[t] = MODE != 2 ? b.par[(size_t)1 * n + e] : 0.0; par2[t] = MODE != 2 ? b.par[(size_t)2 * n + e] : 0.0; par3[t] = MODE != 2 ? b.par[(size_t)3 * n + e] : 0.0;
        s[t] = b.w2h2[e];
#pragma unroll
        for (int i = 0; i < 6; ++i) { B[t][i] = b.rest[(size_t)i * n + e]; U[t][i] = b.u[(size_t)i * n + e]; }
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { XA[t][j] = x[3 * (size_t)id[t].x + j]; XB[t][j] = x[3 * (size_t)id[t].y + j]; XC[t][j] = x[3 * (size_t)id[t].z + j]; }
        if (b.dx_override) {
            const int e = ok[t] ? e0 + 64 * t : e0;
#pragma unroll
            for (int i = 0; i < 6; ++i) DXO[t][i] = b.dx_override[(size_t)i * n + e];
        }
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        if (!ok[t]) continue;
        const int e = e0 + 64 * t;
        double dx[6], u[6], d[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double xa = XA[t][j], xb = XB[t][j], xc = XC[t][j];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                dx[3 * r + j] = ((0.0 + B[t][0 + 3 * r] * xa) + B[t][1 + 3 * r] * xb) + B[t][2 + 3 * r] * xc;
                if (b.dx_override) dx[3 * r + j] = DXO[t][3 * r + j];
                u[3 * r + j] = U[t][3 * r + j];
                d[3 * r + j] = dx[3 * r + j] + u[3 * r + j];
            }
        }
        double zi[6];
        if (MODE == 0) {
            // T = U(:, :2) V^T from the reference's own 3x2 Jacobi SVD (TriangleForce.cpp:83-92): bit-identical with it.
            // (The closed form F (F^T F)^-1/2 is 4x cheaper and agrees to 1e-12, but this kernel is bandwidth-bound anyway.)
            double U2[6], V[4], sv0, sv1, T[6];
            svd32(d, U2, sv0, sv1, V);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i + 3 * j] = U2[i] * V[j] + U2[i + 3] * V[j + 2];
            const double k = kb[t], w2 = w2e[t];
#pragma unroll
            for (int i = 0; i < 6; ++i) zi[i] = (k * T[i] + w2 * d[i]) / (w2 + k);
            if (par3[t] != 0.0) {
                const double lmin = par1[t], lmax = par2[t];
                const double l0 = sqrt(zi[0] * zi[0] + (zi[1] * zi[1] + zi[2] * zi[2]));
                const double l1 = sqrt(zi[3] * zi[3] + (zi[4] * zi[4] + zi[5] * zi[5]));
                const double m0 = (double)fmaxf((float)l0, (float)1e-6), m1 = (double)fmaxf((float)l1, (float)1e-6);
                if (l0 < lmin) { const double sc = lmin / m0; zi[0] *= sc; zi[1] *= sc; zi[2] *= sc; }
                if (l1 < lmin) { const double sc = lmin / m1; zi[3] *= sc; zi[4] *= sc; zi[5] *= sc; }
                if (l0 > lmax) { const double sc = lmax / m0; zi[0] *= sc; zi[1] *= sc; zi[2] *= sc; }
                if (l1 > lmax) { const double sc = lmax / m1; zi[3] *= sc; zi[4] *= sc; zi[5] *= sc; }
            }
        } else if (MODE == 1) {
            double p[6];
            project_triarea_p(d, (int)par1[t], par2[t], par3[t], p);
            const double k = kb[t], w2 = w2e[t];
#pragma unroll
            for (int i = 0; i < 6; ++i) zi[i] = (k * p[i] + w2 * d[i]) / (w2 + k);
        } else {
            double hs = b.state[(size_t)3 * n + e];
            int it = 0;
            project_fung(d, b.par[e], hs, it, zi);
            b.state[(size_t)3 * n + e] = hs;
            b.n_iters[e] = it;
        }
        double q[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double un = u[i] + (dx[i] - zi[i]);
            b.u[(size_t)i * n + e] = un; b.z[(size_t)i * n + e] = zi[i];
            q[i] = zi[i] - un;
        }
        const int dsl[3] = {ds[t].x, ds[t].y, ds[t].z};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) b.fslot[3 * (size_t)dsl[c] + j] = s[t] * (B[t][c] * q[j] + B[t][c + 3] * q[3 + j]);
    }
}

// ---------------------------------------------------------------------------
// the kernels: one batch per launch ...
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(LOCAL_BLOCK) void project_collision_kernel(BatchDev b, const double *__restrict__ x, const ShapeTable *__restrict__ shapes) { project_collision_block(b, x, shapes, (int)blockIdx.x); }
__global__ __launch_bounds__(LOCAL_BLOCK) void project_spring_kernel(BatchDev b, const double *__restrict__ x) { project_spring_block(b, x, (int)blockIdx.x); }
__global__ __launch_bounds__(LOCAL_BLOCK) void project_bend_kernel(BatchDev b, const double *__restrict__ x) { project_bend_block<1>(b, x, (int)blockIdx.x); }
template <int MODE>
__global__ __launch_bounds__(LOCAL_BLOCK) void project_tri_kernel(BatchDev b, const double *__restrict__ x) { project_tri_block<MODE, 1>(b, x, (int)blockIdx.x); }

// ... or the WHOLE local step of a scene with several batches in ONE launch (System.cpp:57-58 is one loop over all forces): the
// batches' blocks back to back in list order.  Launched one after the other, every batch ends with a tail of a few slow waves on a
// mostly idle chip (two tet materials, cloth triangles and hinges of BASELINE configs[4]: 108 + 61 + 31 + 16 + 5 us); in one launch
// the next batch's blocks fill the slots the tail leaves.  Side streams do the same but pay 10-25 us per cross-stream dependency.
// Same per-element arithmetic, own outputs per element: bitwise the same results.
constexpr int MULTI_MAX = 8;
// (two elements per lane for the hinge / triangle segments were built and A/B'd in round 4 -- profiles/r04/mixed_epl_ab.txt: no effect -- and removed)
enum { MK_TET_NH = 0, MK_TET_STVK, MK_TET_LINEAR, MK_TET_VOLUME, MK_ANCHOR, MK_SPRING, MK_BEND, MK_TRI_STRAIN, MK_TRI_AREA, MK_TRI_FUNG, MK_COLLISION };
// (Tried: the segments' blocks interleaved in proportion through a workgroup -> (segment, block) table, so that the memory-bound blocks of the cheap kinds
// share the SIMDs with the tet blocks all along the launch: local step of the mixed scene 0.178 -> 0.196 ms -- dearest first, back to back, is the better schedule.)
struct MultiBatch { int n; int code[MULTI_MAX]; int blk_end[MULTI_MAX]; BatchDev b[MULTI_MAX]; };
#if ADMM_TET_WAVES > 0
__global__ __launch_bounds__(LOCAL_BLOCK, ADMM_TET_WAVES)
#else
__global__ __launch_bounds__(LOCAL_BLOCK)
#endif
void project_multi_kernel(MultiBatch a, const double *__restrict__ x, const ShapeTable *__restrict__ shapes) {
    __shared__ double stage[3 * 256];      // the tet blocks' staging of the RHS shares (one buffer for all four tet kinds)
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i]) ++i;
    const int lb = (int)blockIdx.x - (i ? a.blk_end[i - 1] : 0);
    const BatchDev &b = a.b[i];
    switch (a.code[i]) {
    case MK_TET_NH: project_tet_block<0, 5, false>(b, x, lb, stage); break;
    case MK_TET_STVK: project_tet_block<1, 5, false>(b, x, lb, stage); break;
    case MK_TET_LINEAR: project_tet_block<2, 1, false>(b, x, lb, stage); break;
    case MK_TET_VOLUME: project_tet_block<3, 1, false>(b, x, lb, stage); break;
    case MK_ANCHOR: project_anchor_elem<false>(b, x, b.e0 + lb * LOCAL_BLOCK + threadIdx.x, lb); break;
    case MK_SPRING: project_spring_block(b, x, lb); break;
    case MK_BEND: project_bend_block<1>(b, x, lb); break;
    case MK_TRI_STRAIN: project_tri_block<0, 1>(b, x, lb); break;
    case MK_TRI_AREA: project_tri_block<1, 1>(b, x, lb); break;
    case MK_TRI_FUNG: project_tri_block<2, 1>(b, x, lb); break;
    case MK_COLLISION: project_collision_block(b, x, shapes, lb); break;
    default: break;
    }
}

// ---------------------------------------------------------------------------
// User-defined forces (admm_hip_add_generic_batch): the device's share is the two sparse products around the host hook.
// generic_dx_kernel: Dx = D x for this rank's generic rows, one lane per row, entries in ascending column order from 0.0 --
// the order of Eigen's column-major product at System.cpp:54.  generic_rhs_kernel: one lane per (node slot, component):
// sum of coef * (z - u) over the element's rows that touch this node's component, ascending rows (System.cpp:61).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void generic_dx_kernel(int n_rows, const int *__restrict__ lrow, const int *__restrict__ rptr, const int *__restrict__ col,
                                                         const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ dx) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    double acc = 0.0;
    for (int k = rptr[r]; k < rptr[r + 1]; ++k) acc += val[k] * x[col[k]];
    dx[lrow[r]] = acc;
}
__global__ __launch_bounds__(256) void generic_rhs_kernel(int n_slotcomps, const int *__restrict__ sptr, const int *__restrict__ srow, const double *__restrict__ scoef,
                                                          const int *__restrict__ sdst, const double *__restrict__ q, double *__restrict__ fslot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slotcomps) return;
    int k = sptr[i];
    const int e = sptr[i + 1];
    double acc = 0.0;
    if (k < e) { acc = scoef[k] * q[srow[k]]; for (++k; k < e; ++k) acc += scoef[k] * q[srow[k]]; }
    fslot[3 * (size_t)sdst[i / 3] + i % 3] = acc;
}

} // namespace admm_dev
