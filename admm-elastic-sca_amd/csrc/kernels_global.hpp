// kernels_global.hpp -- ADMM global step on the GPU (gfx950):
//   * frame prologue / epilogue        (System.cpp:37-48, 70-72)
//   * rhs_gather_kernel                b = M x_bar + dt^2 D^T W^2 (z - u)   (System.cpp:61)
//   * the two triangular sweeps of the pre-factored system (System.cpp:62,
//     Eigen SimplicialCholeskyBase::_solve) on the supernodal panel form built
//     by factor.cpp: per elimination-tree level one gather kernel and dense
//     panel x vector kernels with 3 right-hand sides (x,y,z of a node).
//
// All node-indexed vectors live in the factor's (nested-dissection) ordering,
// so no permutation pass exists on the device; element node ids are mapped
// once at upload.  Every reduction has a fixed order: results are bitwise
// reproducible from run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.hpp"

namespace admm_dev {


// v += dt*g (each explicit force in order, ExplicitForce.cpp:29-39);
// x_bar = x + dt v ; Mxbar = m x_bar ; x_cur = x_bar      (System.cpp:46-48)
__global__ void prologue_kernel(int ndof, double dt, Gravity gr, const double *__restrict__ x, double *__restrict__ v,
                                const double *__restrict__ m, double *__restrict__ mxbar, double *__restrict__ xcur) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ndof) return;
    const int c = i % 3;
    double vi = v[i];
    for (int k = 0; k < gr.n; ++k) vi += (dt * gr.g[k][c]);
    v[i] = vi;
    const double xb = x[i] + dt * vi;
    mxbar[i] = m[i] * xb;
    xcur[i] = xb;
}

// x_bar = x + dt v ; Mxbar = m x_bar ; x_cur = x_bar (no explicit force folded in)
__global__ void xbar_kernel(int ndof, double dt, const double *__restrict__ x, const double *__restrict__ v,
                            const double *__restrict__ m, double *__restrict__ mxbar, double *__restrict__ xcur) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ndof) return;
    const double xb = x[i] + dt * v[i];
    mxbar[i] = m[i] * xb;
    xcur[i] = xb;
}

// ExplicitForce::project, ExplicitForce.cpp:29-39: v += dt*dir on all nodes (idx == NULL) or a subset
__global__ void explicit_const_kernel(int n, const int *__restrict__ idx, double dt, double gx, double gy, double gz, double *__restrict__ v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int node = idx ? idx[i] : i;
    v[3 * (size_t)node] += (dt * gx); v[3 * (size_t)node + 1] += (dt * gy); v[3 * (size_t)node + 2] += (dt * gz);
}

// WindForce::project, ExplicitForce.cpp:42-98, with the result of the reference's loop run
// serially (OMP_NUM_THREADS=1): triangle i sees the velocities already incremented by the
// triangles before it, and every node receives its increments in triangle order.  The host
// sorts the triangles into dependency levels (a triangle's level is one more than the highest
// level among earlier triangles sharing a node with it), so the triangles of one level touch
// disjoint nodes and can run side by side; one workgroup walks the levels with a barrier in
// between.  Once per frame, O(#levels) barriers (a few hundred for a cloth sheet).
__global__ __launch_bounds__(1024) void wind_serial_kernel(int n_levels, const int *__restrict__ level_ptr, const int *__restrict__ tris,
                                                           double dt, double wx, double wy, double wz, const double *__restrict__ x, double *v) {
    const double dir[3] = {wx, wy, wz};
    for (int l = 0; l < n_levels; ++l) {
        for (int t = level_ptr[l] + (int)threadIdx.x; t < level_ptr[l + 1]; t += (int)blockDim.x) {
            const size_t i[3] = {3 * (size_t)tris[3 * t], 3 * (size_t)tris[3 * t + 1], 3 * (size_t)tris[3 * t + 2]};
            double vr[3], a[3], b[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double v0 = __hip_atomic_load(&v[i[0] + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double v1 = __hip_atomic_load(&v[i[1] + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double v2 = __hip_atomic_load(&v[i[2] + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vr[j] = (v0 + v1 + v2) / 3.0 - dir[j];
                a[j] = x[i[1] + j] - x[i[0] + j]; b[j] = x[i[2] + j] - x[i[0] + j];
            }
            const double n0 = a[1] * b[2] - a[2] * b[1], n1 = a[2] * b[0] - a[0] * b[2], n2 = a[0] * b[1] - a[1] * b[0];
            const double nn = sqrt(n0 * n0 + (n1 * n1 + n2 * n2));
            const double nm[3] = {n0 / nn, n1 / nn, n2 / nn};
            const double area = 0.5 * nn;
            const double v_n = nm[0] * vr[0] + (nm[1] * vr[1] + nm[2] * vr[2]);
            const double c = -1000.0 * area * v_n * fabs(v_n);
            double f[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { f[j] = c * nm[j]; f[j] *= 0.33; f[j] *= dt; }
            // corner by corner, re-reading: a triangle naming one node twice increments it twice
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double cur = __hip_atomic_load(&v[i[q] + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&v[i[q] + j], cur + f[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        __threadfence();
        __syncthreads();
    }
}

// m_v = (curr_x - m_x) * (1/dt) ; m_x = curr_x             (System.cpp:70-71)
__global__ void epilogue_kernel(int ndof, double dt, double *__restrict__ x, double *__restrict__ v, const double *__restrict__ xcur) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ndof) return;
    const double xc = xcur[i];
    v[i] = (xc - x[i]) * (1.0 / dt);
    x[i] = xc;
}

// Frame boundary of the class API (admm_hip_upload_state / admm_hip_download_state): the caller's node order <-> factor
// order, on the device, so that the host only ever sees one DMA per vector.  perm[i] = caller's node at factor position i.
__global__ void permute_in_kernel(int n_nodes, const int *__restrict__ perm, const double *__restrict__ src_caller, double *__restrict__ dst_factor) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_nodes) return;
    const int node = i / 3, c = i - 3 * node;
    dst_factor[i] = src_caller[3 * (size_t)perm[node] + c];
}
__global__ void permute_out_kernel(int n_nodes, const int *__restrict__ perm, const double *__restrict__ src_factor, double *__restrict__ dst_caller) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_nodes) return;
    const int node = i / 3, c = i - 3 * node;
    dst_caller[3 * (size_t)perm[node] + c] = src_factor[i];
}

// Small systems: the same frame boundary without any DMA -- the kernels read / write a page-locked host buffer directly (x then v, caller's
// order, LINEAR on the host side: coalesced PCIe bursts; the permutation is on the device side).  iperm[node] = factor position of the caller's node.
__global__ void state_in_kernel(int n_nodes, const int *__restrict__ iperm, const double *__restrict__ host_xv, double *__restrict__ x_factor, double *__restrict__ v_factor) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n3 = 3 * n_nodes;
    if (i >= 2 * n3) return;
    const int j = i < n3 ? i : i - n3, node = j / 3, c = j - 3 * node;
    (i < n3 ? x_factor : v_factor)[3 * (size_t)iperm[node] + c] = host_xv[i];
}
// The same for systems of any size straight on the CALLER's page-locked vectors (admm_hip_pin_host; ADMM_HIP_STATE_ZEROCOPY): no staging
// copy on either side, x and v in one launch, every lane one 8-byte word of the host vector (linear on the host side: coalesced PCIe bursts).
__global__ __launch_bounds__(256) void state_in2_kernel(int n_nodes, const int *__restrict__ iperm, const double *__restrict__ host_x, const double *__restrict__ host_v,
                                                        double *__restrict__ x_factor, double *__restrict__ v_factor) {
    const int n3 = 3 * n_nodes;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n3; i += gridDim.x * blockDim.x) {
        const int j = i < n3 ? i : i - n3, node = j / 3, c = j - 3 * node;
        (i < n3 ? x_factor : v_factor)[3 * (size_t)iperm[node] + c] = __builtin_nontemporal_load(&(i < n3 ? host_x : host_v)[j]);
    }
}
__global__ __launch_bounds__(256) void state_out2_kernel(int n_nodes, const int *__restrict__ iperm, const double *__restrict__ x_factor, const double *__restrict__ v_factor,
                                                         double *__restrict__ host_x, double *__restrict__ host_v) {
    const int n3 = 3 * n_nodes;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n3; i += gridDim.x * blockDim.x) {
        const int j = i < n3 ? i : i - n3, node = j / 3, c = j - 3 * node;
        __builtin_nontemporal_store((i < n3 ? x_factor : v_factor)[3 * (size_t)iperm[node] + c], &(i < n3 ? host_x : host_v)[j]);
    }
}
__global__ void state_out_kernel(int n_nodes, const int *__restrict__ iperm, const double *__restrict__ x_factor, const double *__restrict__ v_factor, double *__restrict__ host_xv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n3 = 3 * n_nodes;
    if (i >= 2 * n3) return;
    const int j = i < n3 ? i : i - n3, node = j / 3, c = j - 3 * node;
    host_xv[i] = (i < n3 ? x_factor : v_factor)[3 * (size_t)iperm[node] + c];
}

// One lane per dof: b = base + sum of the node's incident per-corner
// contributions.  The local kernels write every corner's 24 bytes straight to
// its slot (layouts: admm_hip.hip upload_all), summed here in fixed (batch, element, corner) order.
// base = M x_bar exactly once across ranks: on rank 0 (contiguous sharding: add_base), or where base_mask says this
// rank is responsible for the node (subtree sharding: the owner of the node's subtree; rank 0 for the replicated top).
#ifndef ADMM_RHS_UNROLL
#define ADMM_RHS_UNROLL 1
#endif
// The launch covers the nodes [node0, node1).
// NORM (residual tracking, one rank): the launch also leaves the sum of squares of its block's results in norm_partial[block]
// (fixed tree order), so that |s|^2 needs no pass of its own.
template <bool NORM = false>
__global__ __launch_bounds__(256) void rhs_gather_kernel(int node0, int node1, const int64_t *__restrict__ inc_ptr, int slot_stride,
                                  const double *__restrict__ fslot, const double *__restrict__ mxbar, int add_base,
                                  const unsigned char *__restrict__ base_mask, double *__restrict__ y, double *__restrict__ norm_partial = nullptr) {
    __shared__ double red[NORM ? 256 : 1];
    const int i = 3 * node0 + blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < 3 * node1;      // (no early return under NORM: every thread of the block takes part in the barriers below)
    if (!NORM && !ok) return;
    double out = 0.0;
    if (ok) {
        const int node = i / 3, c = i - 3 * node;
        double acc = 0.0;
        const int64_t p0 = inc_ptr[node], p1 = inc_ptr[node + 1];
        if (slot_stride) {      // rank-major slots: lane i reads word i of every rank's array
            const int deg = (int)(p1 - p0);
            const double *f = fslot + i;
            int r = 0;
#if ADMM_RHS_UNROLL
            for (; r + 8 <= deg; r += 8) {      // eight independent loads in flight; the sum keeps its order
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = f[3 * (size_t)(r + q) * slot_stride];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
#endif
            for (; r < deg; ++r) acc += f[3 * (size_t)r * slot_stride];
        } else {
            for (int64_t p = p0; p < p1; ++p) acc += fslot[3 * (size_t)p + c];
        }
        const bool base = base_mask ? (base_mask[node] != 0) : (add_base != 0);
        out = base ? (mxbar[i] + acc) : acc;
        y[i] = out;
    }
    if (NORM) {
        red[threadIdx.x] = out * out;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
        if (threadIdx.x == 0) norm_partial[blockIdx.x] = red[0];
    }
}

// ---- subtree sharding of the solve (one process per GPU): the exchange between a rank's own subtrees and the replicated
// ---- top of the elimination tree is ONE small all-reduce per ADMM iteration over this packed buffer:
// ----   [ y on the top nodes (partial sums of every rank) | the contribution slots of every subtree root (owner's values, 0 elsewhere) ]
__global__ void shard_pack_kernel(int n_top, const int *__restrict__ top_nodes, int n_slots, const int *__restrict__ slots, const unsigned char *__restrict__ slot_mine,
                                  const double *__restrict__ y, const double *__restrict__ C, double *__restrict__ buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_top) { const double *s = y + 3 * (size_t)top_nodes[i]; double *d = buf + 3 * (size_t)i; d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
    else if (i < n_top + n_slots) {
        const int j = i - n_top;
        double *d = buf + 3 * (size_t)i;
        if (slot_mine[j]) { const double *s = C + 3 * (size_t)slots[j]; d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
        else { d[0] = 0.0; d[1] = 0.0; d[2] = 0.0; }
    }
}
__global__ void shard_unpack_kernel(int n_top, const int *__restrict__ top_nodes, int n_slots, const int *__restrict__ slots,
                                    const double *__restrict__ buf, double *__restrict__ y, double *__restrict__ C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_top) { double *d = y + 3 * (size_t)top_nodes[i]; const double *s = buf + 3 * (size_t)i; d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
    else if (i < n_top + n_slots) { double *d = C + 3 * (size_t)slots[i - n_top]; const double *s = buf + 3 * (size_t)i; d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
}
// x is valid on a rank only where it solved: zero the rest before the once-per-frame all-reduce that rebuilds the full vector
__global__ void shard_mask_nodes_kernel(int n_nodes, const unsigned char *__restrict__ keep, double *__restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_nodes) return;
    if (!keep[i / 3]) x[i] = 0.0;
}

// ---------------------------------------------------------------------------
// Residuals (opt-in; the reference only describes them, System.cpp:64-65):
//   r = W (Dx - z)            -- equals W (u_new - u_old), since every kind updates u += Dx - z
//   s = D^T W^T W (z - z_prev)
// Kind-agnostic: the element's selector block is the scalar matrix G (nodes x cols, the same one the
// system matrix is assembled from), its rows are [cols][3].  All sums run in a fixed order.
// ---------------------------------------------------------------------------
constexpr int RES_BLOCK = 256;
// partial[blockIdx] = sum over this block's elements of w2 * |u - u_prev|^2  (fixed tree inside the block)
__global__ __launch_bounds__(RES_BLOCK) void residual_primal_kernel(int n, int rows, const double *__restrict__ u, const double *__restrict__ up,
                                                                      const double *__restrict__ w2, double *__restrict__ partial) {
    __shared__ double red[RES_BLOCK];
    const int e = blockIdx.x * RES_BLOCK + threadIdx.x;
    double a = 0.0;
    if (e < n) {
        for (int r = 0; r < rows; ++r) { const double d = u[(size_t)r * n + e] - up[(size_t)r * n + e]; a += d * d; }
        a *= w2[e];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = RES_BLOCK / 2; off >= 1; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// u, z -> u_prev, z_prev in one launch (kinds whose projection kernel does not track its residuals itself)
__global__ __launch_bounds__(RES_BLOCK) void residual_snapshot_kernel(int64_t n, const double *__restrict__ u, const double *__restrict__ z, double *__restrict__ up, double *__restrict__ zp, int copy_z) {
    const int64_t i = (int64_t)blockIdx.x * RES_BLOCK + threadIdx.x;
    if (i >= n) return;
    up[i] = u[i];
    if (copy_z) zp[i] = z[i];
}
// per corner c: slot[dst[c]] = sum_r G[c][r] * w2 * (z - z_prev)[3r .. 3r+2]
__global__ __launch_bounds__(RES_BLOCK) void residual_dual_kernel(int n, int nn, int cols, int ist, const double *__restrict__ z, const double *__restrict__ zp,
                                                                    const double *__restrict__ w2, const double *__restrict__ G, const int *__restrict__ dst,
                                                                    double *__restrict__ slots) {
    const int e = blockIdx.x * RES_BLOCK + threadIdx.x;
    if (e >= n) return;
    const double w = w2[e];
    double q[9];
    for (int i = 0; i < 3 * cols; ++i) q[i] = w * (z[(size_t)i * n + e] - zp[(size_t)i * n + e]);
    for (int c = 0; c < nn; ++c) {
        double o0 = 0.0, o1 = 0.0, o2 = 0.0;
        for (int r = 0; r < cols; ++r) { const double g = G[(size_t)(3 * c + r) * n + e]; o0 += g * q[3 * r]; o1 += g * q[3 * r + 1]; o2 += g * q[3 * r + 2]; }
        double *o = slots + 3 * (size_t)dst[(size_t)ist * e + c];
        o[0] = o0; o[1] = o1; o[2] = o2;
    }
}
// z_prev of a frame's first iteration is the reference's warm start curr_z = D * m_x (System.cpp:43)
__global__ __launch_bounds__(RES_BLOCK) void residual_dx_kernel(int n, int nn, int cols, int ist, const int *__restrict__ idx, const double *__restrict__ G,
                                                                  const double *__restrict__ x, double *__restrict__ out) {
    const int e = blockIdx.x * RES_BLOCK + threadIdx.x;
    if (e >= n) return;
    double acc[9];
    for (int i = 0; i < 3 * cols; ++i) acc[i] = 0.0;
    for (int c = 0; c < nn; ++c) {
        const double *xn = x + 3 * (size_t)idx[(size_t)ist * e + c];
        for (int r = 0; r < cols; ++r) { const double g = G[(size_t)(3 * c + r) * n + e]; acc[3 * r] += g * xn[0]; acc[3 * r + 1] += g * xn[1]; acc[3 * r + 2] += g * xn[2]; }
    }
    for (int i = 0; i < 3 * cols; ++i) out[(size_t)i * n + e] = acc[i];
}
__global__ __launch_bounds__(RES_BLOCK) void norm2_partial_kernel(int n, const double *__restrict__ v, double *__restrict__ partial) {
    __shared__ double red[RES_BLOCK];
    const int i = blockIdx.x * RES_BLOCK + threadIdx.x;
    const double a = (i < n) ? v[i] * v[i] : 0.0;
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = RES_BLOCK / 2; off >= 1; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] (+)= sqrt-less sum of partial[0..n) in index order (one block, fixed strided order then tree)
__global__ __launch_bounds__(RES_BLOCK) void sum_partials_kernel(int n, const double *__restrict__ partial, double *out, int accumulate) {
    __shared__ double red[RES_BLOCK];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += RES_BLOCK) a += partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = RES_BLOCK / 2; off >= 1; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + red[0] : red[0];
}

// ---------------------------------------------------------------------------
// triangular sweeps
// ---------------------------------------------------------------------------
struct FactorDev {
    const double *panels;
    const int *sn_first, *sn_ncols, *sn_nrows;
    const int64_t *sn_panel_off, *sn_rows_off, *sn_slot_off, *sn_front_off;
    const int *rows;
    const int64_t *cg_ptr;   // per front row: children's contribution slots landing on it
    const int *cg_slot;
    const int4 *cg4;         // same lists as fixed quadruples (-1 = none; trees with <= 4 contributions per front row), or NULL
};

// sum of the children's contributions that land on front row `fr` (fixed child order)
// ---- workgroup timeline of the sweep kernels (tools/sweep_timeline.py; variant build -DADMM_SWEEP_PROFILE, never on in the shipped
// library): every workgroup stamps the 100 MHz real-time counter at its start, after its first staging barrier and at its end
#if defined(ADMM_SWEEP_PROFILE) && defined(__HIPCC__)
__device__ unsigned long long *g_sweep_prof;
#define ADMM_SWEEP_T0 const unsigned long long swp_t0 = __builtin_amdgcn_s_memrealtime(); long long swp_slot = -1;
#define ADMM_SWEEP_SLOT(v) do { swp_slot = (v); if (threadIdx.x == 0 && admm_dev::g_sweep_prof && swp_slot >= 0) admm_dev::g_sweep_prof[4 * swp_slot] = swp_t0; } while (0)
#define ADMM_SWEEP_STAMP(slot) do { if (threadIdx.x == 0 && admm_dev::g_sweep_prof && swp_slot >= 0) admm_dev::g_sweep_prof[4 * swp_slot + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ADMM_SWEEP_T0
#define ADMM_SWEEP_SLOT(v)
#define ADMM_SWEEP_STAMP(slot)
#endif
template <bool CG2>
__device__ __forceinline__ void child_sum(const FactorDev &F, int64_t fr, const double *__restrict__ C, double &s0, double &s1, double &s2) {
    s0 = 0.0; s1 = 0.0; s2 = 0.0;
    if (CG2) {
        const int4 ab = F.cg4[fr];
        if (ab.x >= 0) { const double *c = C + 3 * (size_t)ab.x; s0 += c[0]; s1 += c[1]; s2 += c[2]; }
        if (ab.y >= 0) { const double *c = C + 3 * (size_t)ab.y; s0 += c[0]; s1 += c[1]; s2 += c[2]; }
        if (ab.z >= 0) { const double *c = C + 3 * (size_t)ab.z; s0 += c[0]; s1 += c[1]; s2 += c[2]; }
        if (ab.w >= 0) { const double *c = C + 3 * (size_t)ab.w; s0 += c[0]; s1 += c[1]; s2 += c[2]; }
    } else {
        for (int64_t g = F.cg_ptr[fr]; g < F.cg_ptr[fr + 1]; ++g) {
            const double *c = C + 3 * (size_t)F.cg_slot[g];
            s0 += c[0]; s1 += c[1]; s2 += c[2];
        }
    }
}

#ifndef ADMM_FWD_DEPTH
#define ADMM_FWD_DEPTH 8          // panel columns per load group in the big forward kernel
#endif
#ifndef ADMM_FWD_DEPTH16
#define ADMM_FWD_DEPTH16 4        // ... with 16 waves per tile (the top levels: few, long tiles)
#endif
#ifndef ADMM_FWD_SMALL_DEPTH
#define ADMM_FWD_SMALL_DEPTH 8    // same for the narrow-supernode (wave per tile) forward kernel
#endif
#ifndef ADMM_BWD_UNROLL
#define ADMM_BWD_UNROLL 4         // 64-row groups per load batch in the CW = 1 backward kernel
#endif
#ifndef ADMM_FWD_SMALL_WAVES
#define ADMM_FWD_SMALL_WAVES 4    // wave items per block in the narrow-supernode forward kernel
#endif
#ifndef ADMM_BWD_UNROLL2
#define ADMM_BWD_UNROLL2 1        // 64-row groups per load batch in the CW = 2 backward kernel (1 / 2 / 4: 0.198-0.199 / 0.198-0.201 / 0.202-0.203 ms backward at 1M tets)
#endif
#ifndef ADMM_BWD_PREFETCH
#define ADMM_BWD_PREFETCH 1       // CW > 1 backward kernel: first panel rows requested before the staging barrier
#endif
constexpr int FWD_SMALL_KMAX = 64;

// Forward sweep, supernodes with k <= 64: one wave = one (supernode, 64-row tile);
// lane = row of the panel; t_s = y_s - children's contributions, staged per wave in LDS.
// Everything that does not depend on t_s (the pass-through carry of the lane's own
// row, the first panel columns) is requested before the staging barrier so that the
// dependent index -> slot -> value chain overlaps with the panel stream.
template <bool CG2>
__global__ __launch_bounds__(64 * ADMM_FWD_SMALL_WAVES) void solve_fwd_small_kernel(int n_items, const SweepItem *__restrict__ items,
                                                              FactorDev F, const double *__restrict__ y, double *__restrict__ W, double *__restrict__ C) {
    __shared__ double ts[ADMM_FWD_SMALL_WAVES][FWD_SMALL_KMAX * 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * ADMM_FWD_SMALL_WAVES + wave;
    const bool live = item < n_items;
    ADMM_SWEEP_T0
    int tile = 0, k = 0, r = 0, first = 0;
    int64_t foff = 0, poff = 0, soff = 0;
    if (live) { const SweepItem it = items[item]; tile = it.part; k = it.k; r = it.r; first = it.first; foff = it.front_off; poff = it.panel_off; soff = it.slot_off; ADMM_SWEEP_SLOT(it.pad2); }
    const int f = k + r;
    const int i = tile * 64 + lane;
    const bool row_ok = live && i < f;
    const double *P = F.panels + poff + (row_ok ? i : 0);
    const int jend = row_ok ? ((i < k) ? i + 1 : k) : 0;
    // Same order as in the block kernel below: (1) both index quadruples (this lane's pass-through row, this lane's staged column) and
    // the column's right-hand side; (2) the contribution values, first and second slot together; the staged vector goes to LDS;
    // (3) only then the first group of panel columns.  The staged vector is private to the wave (ts[wave]): no workgroup barrier --
    // LDS operations of one wave execute in order -- so nothing here waits for the panel columns or for the other waves.
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    constexpr int DS = ADMM_FWD_SMALL_DEPTH;
    double cur[DS], nxt[DS];
    const bool pass = row_ok && i >= k, stage = live && lane < k;
    if (CG2) {
        int4 ab_c = F.cg4[foff + (pass ? i : 0)];                                 // (1)
        int4 ab_t = F.cg4[foff + (stage ? lane : 0)];
        const double *src = y + 3 * (size_t)(first + (stage ? lane : 0));
        const double y0 = src[0], y1 = src[1], y2 = src[2];
        if (!pass) ab_c = make_int4(-1, -1, -1, -1);
        if (!stage) ab_t = make_int4(-1, -1, -1, -1);
        const double *cxp = C + 3 * (size_t)max(ab_c.x, 0), *cyp = C + 3 * (size_t)max(ab_c.y, 0);      // (2)
        const double *txp = C + 3 * (size_t)max(ab_t.x, 0), *typ = C + 3 * (size_t)max(ab_t.y, 0);
        const double cx0 = cxp[0], cx1 = cxp[1], cx2 = cxp[2], cy0 = cyp[0], cy1 = cyp[1], cy2 = cyp[2];
        const double tx0 = txp[0], tx1 = txp[1], tx2 = txp[2], ty0 = typ[0], ty1 = typ[1], ty2 = typ[2];
        c0 += ab_c.x >= 0 ? cx0 : 0.0; c1 += ab_c.x >= 0 ? cx1 : 0.0; c2 += ab_c.x >= 0 ? cx2 : 0.0;
        c0 += ab_c.y >= 0 ? cy0 : 0.0; c1 += ab_c.y >= 0 ? cy1 : 0.0; c2 += ab_c.y >= 0 ? cy2 : 0.0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        s0 += ab_t.x >= 0 ? tx0 : 0.0; s1 += ab_t.x >= 0 ? tx1 : 0.0; s2 += ab_t.x >= 0 ? tx2 : 0.0;
        s0 += ab_t.y >= 0 ? ty0 : 0.0; s1 += ab_t.y >= 0 ? ty1 : 0.0; s2 += ab_t.y >= 0 ? ty2 : 0.0;
        // (third and fourth slot: only under four-way tree nodes)
        if (ab_c.z >= 0) { const double *c = C + 3 * (size_t)ab_c.z; c0 += c[0]; c1 += c[1]; c2 += c[2]; }
        if (ab_c.w >= 0) { const double *c = C + 3 * (size_t)ab_c.w; c0 += c[0]; c1 += c[1]; c2 += c[2]; }
        if (ab_t.z >= 0) { const double *c = C + 3 * (size_t)ab_t.z; s0 += c[0]; s1 += c[1]; s2 += c[2]; }
        if (ab_t.w >= 0) { const double *c = C + 3 * (size_t)ab_t.w; s0 += c[0]; s1 += c[1]; s2 += c[2]; }
        if (stage) { ts[wave][3 * lane] = y0 - s0; ts[wave][3 * lane + 1] = y1 - s1; ts[wave][3 * lane + 2] = y2 - s2; }
    } else {
        if (pass) child_sum<CG2>(F, foff + i, C, c0, c1, c2);   // pass-through of the children's rows beyond this supernode
        if (stage) {
            const double *src = y + 3 * (size_t)(first + lane);
            double s0, s1, s2;
            child_sum<CG2>(F, foff + lane, C, s0, s1, s2);
            ts[wave][3 * lane] = src[0] - s0; ts[wave][3 * lane + 1] = src[1] - s1; ts[wave][3 * lane + 2] = src[2] - s2;
        }
    }
    // (3) first group of panel columns; afterwards the next group is always requested before the current one is consumed.
    // Unconditional loads: a column beyond the row's range re-reads its last one and counts as zero.
    {
        const int jl = max(jend - 1, 0);
#pragma unroll
        for (int q = 0; q < DS; ++q) cur[q] = P[(size_t)f * min(q, jl)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own staged vector is in LDS (no workgroup barrier: ts[wave] is private)
#pragma unroll
    for (int q = 0; q < DS; ++q) cur[q] = (q < jend) ? cur[q] : 0.0;
    ADMM_SWEEP_STAMP(1);
    if (!row_ok) { ADMM_SWEEP_STAMP(2); return; }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int j = 0; j < jend; j += DS) {
#pragma unroll
        for (int q = 0; q < DS; ++q) nxt[q] = (j + DS + q < jend) ? P[(size_t)f * (j + DS + q)] : 0.0;
        const double *t = &ts[wave][3 * j];
#pragma unroll
        for (int q = 0; q < DS; ++q) if (j + q < jend) { a0 += cur[q] * t[3 * q]; a1 += cur[q] * t[3 * q + 1]; a2 += cur[q] * t[3 * q + 2]; }
#pragma unroll
        for (int q = 0; q < DS; ++q) cur[q] = nxt[q];
    }
    if (i < k) { double *dst = W + 3 * (size_t)(first + i); dst[0] = a0; dst[1] = a1; dst[2] = a2; }
    else {
        double *dst = C + 3 * (size_t)(soff + (i - k));
        dst[0] = a0 + c0; dst[1] = a1 + c1; dst[2] = a2 + c2;
    }
    ADMM_SWEEP_STAMP(2);
}

// Forward sweep, supernodes with k > 64: one block of NW waves = one
// (supernode, 64-row tile); the waves split the columns, partial sums are
// combined through LDS in wave order.  NW = 16 for the wide supernodes at the top of the tree; levels of narrower ones run
// with 4 or 8 waves per tile (chosen per level from its widest supernode, admm_hip.hip upload_factor): a CU holds the same
// number of waves either way, but four times as many tiles are resident at once and every wave has a full group of columns
// to stream instead of a handful -- such levels are bound by the per-tile prologue latency, not by bytes.
template <int NW> struct FwdBig { static constexpr int KCHUNK = NW == 16 ? 2048 : 512; };
template <bool CG2, int NW>
__global__ __launch_bounds__(64 * NW) void solve_fwd_big_kernel(const SweepItem *__restrict__ items,
                                                             FactorDev F, const double *__restrict__ y, double *__restrict__ W, double *__restrict__ C) {
    constexpr int FWD_BIG_KCHUNK = FwdBig<NW>::KCHUNK, NT = 64 * NW;
    __shared__ double ts[FWD_BIG_KCHUNK * 3];
    __shared__ double red[NW][64 * 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ADMM_SWEEP_T0
    const SweepItem it = items[blockIdx.x];
    ADMM_SWEEP_SLOT(it.pad2);
    const int tile = it.part, k = it.k, r = it.r, first = it.first;
    const int64_t foff = it.front_off;
    const int f = k + r;
    const int i = tile * 64 + lane;
    const bool row_ok = i < f;
    const double *P = F.panels + it.panel_off + (row_ok ? i : 0);
    // it.pad != 0: a root's explicit inverse (L L^T)^-1 -- full rows, and the caller passes x as W: forward and backward in one product
    const bool full = it.pad != 0;
    const int jend = (full || i >= k) ? k : i + 1;
    // columns beyond the tile's last row never contribute to a tile inside the triangle
    const int kneed = full ? k : min(k, (tile * 64 + 64 <= k) ? tile * 64 + 64 : k);
    // The path to the staging barrier in dependency order (vector loads complete in issue order): (1) the index quadruples of this
    // thread's carry row and of its first staged column + that column's right-hand side; (2) the contribution values they point to,
    // first and second slot together (an absent one reads slot 0 and counts as zero); the staged vector goes to LDS; (3) only then
    // the first group of panel columns (cold: the burst of all workgroups takes ~3 us to land) and a barrier that waits for LDS
    // alone, so the burst overlaps the barrier and the first LDS reads.  Before: carry chain, panel request + a barrier waiting
    // for it, staging chain, barrier -- three latencies in a row (probe stamps).
    const int kc_first = min(FWD_BIG_KCHUNK, kneed);
    const bool pre_ts = CG2 && (int)threadIdx.x < kc_first;          // this thread's first staged column
    const int c_ln = threadIdx.x / 3, c_c = threadIdx.x - 3 * c_ln, c_row = tile * 64 + c_ln;
    const bool pre_carry = threadIdx.x < 192 && c_row < f && c_row >= k;
    double carry = 0.0;
    if (CG2) {
        int4 ab_c = F.cg4[foff + (pre_carry ? c_row : 0)];                        // (1)
        int4 ab_t = F.cg4[foff + (pre_ts ? (int)threadIdx.x : 0)];
        const double *src = y + 3 * (size_t)(first + (pre_ts ? (int)threadIdx.x : 0));
        const double y0 = src[0], y1 = src[1], y2 = src[2];
        if (!pre_carry) ab_c = make_int4(-1, -1, -1, -1);
        if (!pre_ts) ab_t = make_int4(-1, -1, -1, -1);
        const double cvx = C[3 * (size_t)max(ab_c.x, 0) + c_c], cvy = C[3 * (size_t)max(ab_c.y, 0) + c_c];      // (2)
        const double *txp = C + 3 * (size_t)max(ab_t.x, 0), *typ = C + 3 * (size_t)max(ab_t.y, 0);
        const double tx0 = txp[0], tx1 = txp[1], tx2 = txp[2], ty0 = typ[0], ty1 = typ[1], ty2 = typ[2];
        carry += ab_c.x >= 0 ? cvx : 0.0; carry += ab_c.y >= 0 ? cvy : 0.0;
        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
        p0 += ab_t.x >= 0 ? tx0 : 0.0; p1 += ab_t.x >= 0 ? tx1 : 0.0; p2 += ab_t.x >= 0 ? tx2 : 0.0;
        p0 += ab_t.y >= 0 ? ty0 : 0.0; p1 += ab_t.y >= 0 ? ty1 : 0.0; p2 += ab_t.y >= 0 ? ty2 : 0.0;
        // (third and fourth slot: only under four-way tree nodes)
        if (ab_c.z >= 0) carry += C[3 * (size_t)ab_c.z + c_c];
        if (ab_c.w >= 0) carry += C[3 * (size_t)ab_c.w + c_c];
        if (ab_t.z >= 0) { const double *c = C + 3 * (size_t)ab_t.z; p0 += c[0]; p1 += c[1]; p2 += c[2]; }
        if (ab_t.w >= 0) { const double *c = C + 3 * (size_t)ab_t.w; p0 += c[0]; p1 += c[1]; p2 += c[2]; }
        if (pre_ts) { ts[3 * threadIdx.x] = y0 - p0; ts[3 * threadIdx.x + 1] = y1 - p1; ts[3 * threadIdx.x + 2] = y2 - p2; }
    } else if (pre_carry) {
        for (int64_t g = F.cg_ptr[foff + c_row]; g < F.cg_ptr[foff + c_row + 1]; ++g) carry += C[3 * (size_t)F.cg_slot[g] + c_c];
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int c0 = 0; c0 < kneed; c0 += FWD_BIG_KCHUNK) {
        const int kc = min(FWD_BIG_KCHUNK, kneed - c0);
        const int per = (kc + NW - 1) / NW;
        const int jb = c0 + wave * per;
        int je = min(jb + per, c0 + kc);
        je = row_ok ? min(je, jend) : jb;
        constexpr int D = NW == 16 ? ADMM_FWD_DEPTH16 : ADMM_FWD_DEPTH;
        double cur[D], nxt[D];
        if (c0 > 0) __syncthreads();
        // columns this thread did not stage ahead (k beyond one per thread; without the quadruple lists: all of them)
        for (int q = (c0 == 0 && CG2) ? (int)threadIdx.x + NT : (int)threadIdx.x; q < kc; q += NT) {
            const double *src = y + 3 * (size_t)(first + c0 + q);
            double s0, s1, s2;
            child_sum<CG2>(F, foff + c0 + q, C, s0, s1, s2);
            ts[3 * q] = src[0] - s0; ts[3 * q + 1] = src[1] - s1; ts[3 * q + 2] = src[2] - s2;
        }
        // (3) first group of this wave's panel columns; after the barrier the next group is always requested before the current one
        // is consumed (two groups in flight).  Unconditional loads: a column beyond the wave's range re-reads its last one, counts as 0.
        {
            // (a wave with no columns -- je == jb, possibly beyond k on a narrow supernode of a wide level -- re-reads column k - 1:
            // every speculative address stays inside this supernode's panel)
            const int jl = min(max(je - 1, 0), k - 1);
#pragma unroll
            for (int q = 0; q < D; ++q) cur[q] = P[(size_t)f * min(jb + q, jl)];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: the panel columns stay in flight
#pragma unroll
        for (int q = 0; q < D; ++q) cur[q] = (jb + q < je) ? cur[q] : 0.0;
        if (c0 == 0) ADMM_SWEEP_STAMP(1);
        for (int j = jb; j < je; j += D) {
#pragma unroll
            for (int q = 0; q < D; ++q) nxt[q] = (j + D + q < je) ? P[(size_t)f * (j + D + q)] : 0.0;
            const double *t = &ts[3 * (j - c0)];
#pragma unroll
            for (int q = 0; q < D; ++q) if (j + q < je) { a0 += cur[q] * t[3 * q]; a1 += cur[q] * t[3 * q + 1]; a2 += cur[q] * t[3 * q + 2]; }
#pragma unroll
            for (int q = 0; q < D; ++q) cur[q] = nxt[q];
        }
    }
    red[wave][3 * lane] = a0; red[wave][3 * lane + 1] = a1; red[wave][3 * lane + 2] = a2;
    __syncthreads();
    if (threadIdx.x < 192) {
        const int ln = threadIdx.x / 3, c = threadIdx.x - 3 * ln;
        const int row = tile * 64 + ln;
        if (row < f) {
            double acc = red[0][3 * ln + c];
#pragma unroll
            for (int w = 1; w < NW; ++w) acc += red[w][3 * ln + c];
            if (row < k) W[3 * (size_t)(first + row) + c] = acc;
            else C[3 * (size_t)(it.slot_off + (row - k)) + c] = acc + carry;
        }
    }
    ADMM_SWEEP_STAMP(2);
}

// Backward sweep: one 256-thread block = (supernode, 4*CW columns), each wave owns
// CW columns; lanes stride over the rows of the (contiguous) panel columns; the
// vector [w_s ; -x(R_s)] is staged in LDS in row chunks.  CW = 1 for the big
// supernodes (maximum number of blocks in flight), CW = 4 for levels of small
// supernodes (one staging of v per 16 columns instead of per 4).
// (Tried and dropped: walking whole bottom subtrees inside one workgroup with
// workgroup barriers between levels -- fewer launches but 5-20 % slower, the
// per-level launches expose more parallelism to hide the dependent index ->
// slot -> value loads.)
// Sum n per-lane values across the 64 lanes of a wave with a transposing butterfly: at every one of the six
// steps a lane keeps one half of its values and receives its partner's copies of that half, so the count
// halves while all lanes stay busy: 14 shuffles for 12 values (7 for 3) instead of 6 per value.  Fixed
// pairwise order.  On return, lanes with cnt >= 1 hold in v[0] the wave total of value index `base`.
template <int n, int off>
__device__ __forceinline__ void wave_sum_transpose(double *v, int lane, int &base, int &cnt) {
    constexpr int h = (n + 1) / 2;
    const bool up = (lane & off) != 0;
    double keep[h], recv[h];
#pragma unroll
    for (int t = 0; t < h; ++t) {
        const double lo = v[t], hi = (h + t < n) ? v[h + t] : 0.0;
        keep[t] = up ? hi : lo;
        recv[t] = __shfl_xor(up ? lo : hi, off, 64);
    }
#pragma unroll
    for (int t = 0; t < h; ++t) v[t] = keep[t] + recv[t];
    if (up) { base += h; cnt = max(cnt - h, 0); } else cnt = min(cnt, h);
    if constexpr (off > 1) wave_sum_transpose<h, off / 2>(v, lane, base, cnt);
}

#ifndef ADMM_BWD_RCHUNK
#define ADMM_BWD_RCHUNK 1024
#endif
constexpr int BWD_RCHUNK = ADMM_BWD_RCHUNK;
// NWB waves per block share one staging of the vector (4 is the original shape; 8 / 16 halve / quarter the staging work
// per column on levels of tall fronts).
template <int CW, int NWB = 4>
__global__ __launch_bounds__(64 * NWB) void solve_bwd_kernel(const SweepItem *__restrict__ items,
                                                        FactorDev F, const double *__restrict__ W, double *__restrict__ X) {
    __shared__ double vs[BWD_RCHUNK * 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ADMM_SWEEP_T0
    const SweepItem it = items[blockIdx.x];
    ADMM_SWEEP_SLOT(it.pad2);
    const int chunk = it.part, k = it.k, r = it.r, first = it.first;
    const int f = k + r;
    const int *rows = F.rows + it.rows_off;
    const int jc0 = chunk * (NWB * CW);         // first column of this block
    const int j0 = jc0 + wave * CW;             // this wave's first column
    const double *Pj = F.panels + it.panel_off + (size_t)f * min(j0, k - 1);
    double acc[CW][3];
#pragma unroll
    for (int c = 0; c < CW; ++c) { acc[c][0] = 0.0; acc[c][1] = 0.0; acc[c][2] = 0.0; }
    // the panel does not depend on the staged vector: its first rows are requested before the staging barrier
    double pf[CW];
    if (CW != 1 && ADMM_BWD_PREFETCH) {
#pragma unroll
        for (int c = 0; c < CW; ++c) { const int i = jc0 + lane; pf[c] = (j0 + c < k && i < f && i >= j0 + c) ? Pj[i + (size_t)f * c] : 0.0; }
    }
    // rows < jc0 are never needed by this block (lower triangular diagonal block)
    for (int r0 = jc0; r0 < f; r0 += BWD_RCHUNK) {
        const int rc = min(BWD_RCHUNK, f - r0);
        __syncthreads();
        for (int q = threadIdx.x; q < rc; q += 64 * NWB) {
            const int i = r0 + q;
            double v0, v1, v2;
            if (i < k) { const double *src = W + 3 * (size_t)(first + i); v0 = src[0]; v1 = src[1]; v2 = src[2]; }
            else { const double *src = X + 3 * (size_t)rows[i - k]; v0 = -src[0]; v1 = -src[1]; v2 = -src[2]; }
            vs[3 * q] = v0; vs[3 * q + 1] = v1; vs[3 * q + 2] = v2;
        }
        __syncthreads();
        if (r0 == jc0) ADMM_SWEEP_STAMP(1);
        if (j0 < k) {
            if (CW == 1) {
                int q = lane;
                constexpr int U = ADMM_BWD_UNROLL;
                for (; q + 64 * (U - 1) < rc; q += 64 * U) {
                    double p[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const int i = r0 + q + 64 * u; p[u] = (i >= j0) ? Pj[i] : 0.0; }
#pragma unroll
                    for (int u = 0; u < U; ++u) { const double *v = &vs[3 * (q + 64 * u)]; acc[0][0] += p[u] * v[0]; acc[0][1] += p[u] * v[1]; acc[0][2] += p[u] * v[2]; }
                }
                for (; q < rc; q += 64) {
                    const int i = r0 + q;
                    if (i >= j0) { const double p = Pj[i]; const double *v = &vs[3 * q]; acc[0][0] += p * v[0]; acc[0][1] += p * v[1]; acc[0][2] += p * v[2]; }
                }
            } else {
                int q = lane;
                if (ADMM_BWD_PREFETCH && r0 == jc0 && q < rc) {      // the rows requested before the staging barrier
                    const double *v = &vs[3 * q];
#pragma unroll
                    for (int c = 0; c < CW; ++c) { acc[c][0] += pf[c] * v[0]; acc[c][1] += pf[c] * v[1]; acc[c][2] += pf[c] * v[2]; }
                    q += 64;
                }
                constexpr int UC = CW == 2 ? ADMM_BWD_UNROLL2 : 1;
                for (; q + 64 * (UC - 1) < rc; q += 64 * UC) {
                    double p[UC][CW];
#pragma unroll
                    for (int u = 0; u < UC; ++u)
#pragma unroll
                        for (int c = 0; c < CW; ++c) { const int i = r0 + q + 64 * u; p[u][c] = (j0 + c < k && i >= j0 + c) ? Pj[i + (size_t)f * c] : 0.0; }
#pragma unroll
                    for (int u = 0; u < UC; ++u) {
                        const double *v = &vs[3 * (q + 64 * u)];
#pragma unroll
                        for (int c = 0; c < CW; ++c) { acc[c][0] += p[u][c] * v[0]; acc[c][1] += p[u][c] * v[1]; acc[c][2] += p[u][c] * v[2]; }
                    }
                }
                for (; q < rc; q += 64) {
                    const int i = r0 + q;
                    const double *v = &vs[3 * q];
#pragma unroll
                    for (int c = 0; c < CW; ++c) { const double p = (j0 + c < k && i >= j0 + c) ? Pj[i + (size_t)f * c] : 0.0; acc[c][0] += p * v[0]; acc[c][1] += p * v[1]; acc[c][2] += p * v[2]; }
                }
            }
        }
    }
    double v[3 * CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { v[3 * c] = acc[c][0]; v[3 * c + 1] = acc[c][1]; v[3 * c + 2] = acc[c][2]; }
    int base = 0, cnt = 3 * CW;
    wave_sum_transpose<3 * CW, 32>(v, lane, base, cnt);
    if (cnt >= 1) {
        const int c = base / 3, m = base - 3 * c;
        if (j0 + c < k) X[3 * (size_t)(first + j0 + c) + m] = v[0];
    }
    ADMM_SWEEP_STAMP(2);
}

// Roots of the elimination tree: x_s = (L_ss L_ss^T)^-1 t_s, both sweeps in one product.  t_s = y_s - (children's contributions)
// is gathered once (this kernel), then dense_solve_kernel -- one wave per ROW of the symmetric inverse, every load of a row
// independent of the others -- streams the inverse: 1089 columns in ~6 us where the tile kernel's 16 waves x 68 dependent
// columns needed 20 us, and a merged 3335-column root (ND::merge_root) at HBM rate.
template <bool CG2>
__global__ __launch_bounds__(256) void root_gather_kernel(int k, int first, int64_t foff, FactorDev F, const double *__restrict__ y, const double *__restrict__ C, double *__restrict__ T, double *__restrict__ Xzero = nullptr) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k) return;
    if (Xzero) { Xzero[3 * (size_t)j] = 0.0; Xzero[3 * (size_t)j + 1] = 0.0; Xzero[3 * (size_t)j + 2] = 0.0; }      // (distributed top: x of the root's rows other ranks compute -- zeros go into the exchange)
    double s0, s1, s2;
    child_sum<CG2>(F, foff + j, C, s0, s1, s2);
    const double *src = y + 3 * (size_t)(first + j);
    double *t = T + 3 * (size_t)j;
    t[0] = src[0] - s0; t[1] = src[1] - s1; t[2] = src[2] - s2;
}

// Small systems: x = A_s^-1 b with the explicit (symmetric) inverse, one wave per row, 4 rows per block; lanes stride
// over the columns (coalesced 512-B reads of the row, 1.5 KB of b), fixed-order lane sums, transposing butterfly.
__global__ __launch_bounds__(256) void dense_solve_kernel(int n, const double *__restrict__ Ainv, const double *__restrict__ b, double *__restrict__ X) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const double *row = Ainv + (size_t)i * n;
    double v[3] = {0.0, 0.0, 0.0};
    int j = lane;
    for (; j + 448 < n; j += 512) {      // eight independent 512-byte row pieces in flight per wave
        double a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = row[j + 64 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const double *bj = b + 3 * (size_t)(j + 64 * q); v[0] += a[q] * bj[0]; v[1] += a[q] * bj[1]; v[2] += a[q] * bj[2]; }
    }
    for (; j + 192 < n; j += 256) {
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = row[j + 64 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const double *bj = b + 3 * (size_t)(j + 64 * q); v[0] += a[q] * bj[0]; v[1] += a[q] * bj[1]; v[2] += a[q] * bj[2]; }
    }
    for (; j < n; j += 64) { const double a = row[j]; const double *bj = b + 3 * (size_t)j; v[0] += a * bj[0]; v[1] += a * bj[1]; v[2] += a * bj[2]; }
    int base = 0, cnt = 3;
    wave_sum_transpose<3, 32>(v, lane, base, cnt);
    if (cnt >= 1) X[3 * (size_t)i + base] = v[0];
}

// A root's product x = S^-1 t at HBM rate: one wave per row of the symmetric inverse (16 rows per block), t staged through
// LDS in chunks of 2048 columns (read once per block instead of once per row), eight independent 512-byte row pieces in
// flight per wave.  Fixed summation order: a lane's columns ascending, then the transposing butterfly.
#ifndef ADMM_ROOT_VEC2
#define ADMM_ROOT_VEC2 1
#endif
constexpr int ROOT_KCHUNK = 2048;
#ifndef ADMM_ROOT_ROWS
#define ADMM_ROOT_ROWS 16         // rows (= waves) per block of root_product_kernel: one staging of t per block (4 / 8 / 16 rows: 37 / 22 / 20 us at k = 3301)
#endif
constexpr int ROOT_ROWS = ADMM_ROOT_ROWS;
// ADMM_ROOT_NT: the root's explicit inverse is read ONCE per ADMM iteration (it serves both sweeps in one product): loaded
// non-temporally it does not take the Infinity Cache away from the top levels' panels, which the backward sweep re-reads next.
#ifndef ADMM_ROOT_NT
#define ADMM_ROOT_NT 1
#endif
// GATHER (roots of at most ROOT_KCHUNK columns): every block forms t = y - (children's contributions) itself instead of reading the T a
// root_gather_kernel launch wrote -- one launch less where a launch is 4-5 us of pure latency (mid-size systems); T then carries y.
template <bool GATHER, bool CG2>
// nrows: rows of the product this launch computes -- Sinv points at the first of them, X at its x (the whole root: nrows = k; subtree sharding's
// distributed top: this rank's slice of the rows, every row the same arithmetic whichever rank computes it).
__global__ __launch_bounds__(64 * ROOT_ROWS) void root_product_kernel(int k, int nrows, int ld, const double *__restrict__ Sinv, const double *__restrict__ T, double *__restrict__ X,
                                                                      int first, int64_t foff, FactorDev F, const double *__restrict__ C) {
    __shared__ double ts[ROOT_KCHUNK * 3 + 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROOT_ROWS + wave;
    const double *rp = Sinv + (size_t)min(row, nrows - 1) * ld;      // ld is a multiple of 16 doubles: every row starts on a 128-byte line
    double v[3] = {0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < k; c0 += ROOT_KCHUNK) {
        const int kc = min(ROOT_KCHUNK, k - c0);
        __syncthreads();
        if (GATHER) {
            for (int q = threadIdx.x; q < kc; q += 64 * ROOT_ROWS) {
                double s0, s1, s2;
                child_sum<CG2>(F, foff + c0 + q, C, s0, s1, s2);
                const double *src = T + 3 * (size_t)(first + c0 + q);
                ts[3 * q] = src[0] - s0; ts[3 * q + 1] = src[1] - s1; ts[3 * q + 2] = src[2] - s2;
            }
        } else
        for (int q = threadIdx.x; q < 3 * kc; q += 64 * ROOT_ROWS) ts[q] = T[3 * (size_t)c0 + q];
        if (kc & 1) { if (threadIdx.x < 3) ts[3 * kc + threadIdx.x] = 0.0; }      // the pair loads below may reach one column past an odd chunk (padding of the row: finite)
        __syncthreads();
#if ADMM_ROOT_VEC2
        // 16 bytes per lane and load: a wave covers 128 columns per instruction, eight instructions in flight
        const double2 *rp2 = reinterpret_cast<const double2 *>(rp + c0);
        const int kc2 = (kc + 1) >> 1;
        int j = lane;
        for (; j + 448 < kc2; j += 512) {
            double2 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ADMM_ROOT_NT) { typedef double d2_t __attribute__((ext_vector_type(2))); const d2_t t2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(&rp2[j + 64 * u])); a[u].x = t2.x; a[u].y = t2.y; }
                else a[u] = rp2[j + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double *t = &ts[6 * (j + 64 * u)];
                v[0] += a[u].x * t[0]; v[1] += a[u].x * t[1]; v[2] += a[u].x * t[2];
                v[0] += a[u].y * t[3]; v[1] += a[u].y * t[4]; v[2] += a[u].y * t[5];
            }
        }
        for (; j < kc2; j += 64) {
            const double2 a = rp2[j]; const double *t = &ts[6 * j];
            v[0] += a.x * t[0]; v[1] += a.x * t[1]; v[2] += a.x * t[2];
            v[0] += a.y * t[3]; v[1] += a.y * t[4]; v[2] += a.y * t[5];
        }
#else
        int j = lane;
        for (; j + 448 < kc; j += 512) {
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = ADMM_ROOT_NT ? __builtin_nontemporal_load(&rp[c0 + j + 64 * u]) : rp[c0 + j + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const double *t = &ts[3 * (j + 64 * u)]; v[0] += a[u] * t[0]; v[1] += a[u] * t[1]; v[2] += a[u] * t[2]; }
        }
        for (; j < kc; j += 64) { const double a = rp[c0 + j]; const double *t = &ts[3 * j]; v[0] += a * t[0]; v[1] += a * t[1]; v[2] += a * t[2]; }
#endif
    }
    int base = 0, cnt = 3;
    wave_sum_transpose<3, 32>(v, lane, base, cnt);
    if (cnt >= 1 && row < nrows) X[3 * (size_t)row + base] = v[0];
}

} // namespace admm_dev
