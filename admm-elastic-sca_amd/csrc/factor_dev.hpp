// factor_dev.hpp -- the NUMERIC multifrontal factorization on the GPU (kernels).
//
// The reference factors A = M + dt^2 D^T W^2 D once in System::initialize and again in recompute_weights
// (deps/admm-elastic-sca/src/system/System.cpp:138-140, :167-179: Eigen SimplicialLDLT on the host).  Ordering and symbolic analysis
// stay on the host (factor.cpp); the arithmetic -- assembly of the fronts, extend-add, the dense partial Cholesky of every front, the
// triangular inverses and the panel products the sweeps stream -- runs here, level by level of the elimination tree, every level
// as a handful of BATCHED launches over all its fronts:
//
//   front F_s (f x f, lower, column-major, all fronts resident at once):
//     F_s  = A(columns of s) (+) children's update matrices               assemble_kernel, extend_add_kernel
//     for every 64-column block J of the k pivot columns:
//         D = F[J,J] = L_D L_D^T ;  Dinv = L_D^-1  -> P[J,J]               potrf_inv_kernel   (one workgroup per front)
//         F[below J, J] <- F[below J, J] Dinv^T                            gemm_f64_kernel    (TRSM as a product with the block inverse)
//         F[below J, below J] -= F[below J, J] F[below J, J]^T  (lower)    gemm_f64_kernel
//     P[0:k, 0:k] = L_11^-1 by recursive doubling over the 64-blocks        gemm_f64_kernel x 2 per round
//     P[k:f, 0:k] = L_21 L_11^-1                                            gemm_f64_kernel
//   roots with an explicit inverse: S^-1 = L^-T L^-1                         gemm_f64_kernel
//
// gemm_f64_kernel is one fp64 MFMA kernel (v_mfma_f64_16x16x4_f64, 64 x 64 tile per 256-thread workgroup, operands staged through
// LDS in k-chunks of 16 with the next chunk's global loads in flight) driven by task records; a launch covers all tasks of one
// step of one level (blockIdx.y = task, blockIdx.x = tile; tiles beyond a task's extent leave at once).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace admm_dev {

enum { GEMM_TRANS_A = 1, GEMM_TRANS_B = 2, GEMM_LOWER_TILES = 4, GEMM_K_FROM_COL = 8, GEMM_K_FROM_MAX = 16 };

// C (m x n, ldc) = beta C + alpha opA(A) (m x k) opB(B) (k x n), column-major.
//   opA(A)(i, q) = TRANS_A ? A[q + lda i] : A[i + lda q];   opB(B)(q, j) = TRANS_B ? B[j + ldb q] : B[q + ldb j]
//   LOWER_TILES: only tiles with tile row >= tile column (C symmetric / lower);  K_FROM_COL: the sum starts at the tile's first column
//   (opB lower triangular);  K_FROM_MAX: at the larger of tile row / column start (A^T A of a lower triangular A)
struct GemmTask {
    const double *A, *B;
    double *C;
    int m, n, k, lda, ldb, ldc, flags, koff;      // koff: K_FROM_MAX counts the tile's column from here (C is a column slice [koff, koff + n) of the full product)
    double alpha, beta;
};

typedef double dbl4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_KT = 16;

__global__ __launch_bounds__(256) void gemm_f64_kernel(const GemmTask *__restrict__ tasks) {
    __shared__ double As[GEMM_KT][64 + 4];      // [q][i]
    __shared__ double Bs[GEMM_KT][64 + 4];      // [q][j]
    const GemmTask T = tasks[blockIdx.y];
    const int tm = (T.m + 63) >> 6, tn = (T.n + 63) >> 6;
    if ((int)blockIdx.x >= tm * tn) return;
    const int ti = blockIdx.x % tm, tj = blockIdx.x / tm;
    if ((T.flags & GEMM_LOWER_TILES) && tj > ti) return;
    const int i0 = ti << 6, j0 = tj << 6;
    int k0 = 0;
    if (T.flags & GEMM_K_FROM_COL) k0 = j0;
    if (T.flags & GEMM_K_FROM_MAX) k0 = max(i0, j0 + T.koff);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const bool ta = T.flags & GEMM_TRANS_A, tb = T.flags & GEMM_TRANS_B;
    // global -> register staging of one k-chunk (4 doubles of A, 4 of B per thread)
    double ra[4], rb[4];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int i, kk;
            if (!ta) { i = t & 63; kk = (t >> 6) + 4 * q; } else { kk = t & 15; i = (t >> 4) + 16 * q; }
            const int gi = i0 + i, gk = kc + kk;
            ra[q] = (gi < T.m && gk < T.k) ? (ta ? T.A[gk + (size_t)T.lda * gi] : T.A[gi + (size_t)T.lda * gk]) : 0.0;
            int j, kb;
            if (tb) { j = t & 63; kb = (t >> 6) + 4 * q; } else { kb = t & 15; j = (t >> 4) + 16 * q; }
            const int gj = j0 + j, gkb = kc + kb;
            rb[q] = (gj < T.n && gkb < T.k) ? (tb ? T.B[gj + (size_t)T.ldb * gkb] : T.B[gkb + (size_t)T.ldb * gj]) : 0.0;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int i, kk;
            if (!ta) { i = t & 63; kk = (t >> 6) + 4 * q; } else { kk = t & 15; i = (t >> 4) + 16 * q; }
            As[kk][i] = ra[q];
            int j, kb;
            if (tb) { j = t & 63; kb = (t >> 6) + 4 * q; } else { kb = t & 15; j = (t >> 4) + 16 * q; }
            Bs[kb][j] = rb[q];
        }
    };
    dbl4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (dbl4){0.0, 0.0, 0.0, 0.0};
    if (k0 < T.k) fetch(k0);
    for (int kc = k0; kc < T.k; kc += GEMM_KT) {
        __syncthreads();
        stash();
        __syncthreads();
        if (kc + GEMM_KT < T.k) fetch(kc + GEMM_KT);
        // D'[row = j][col = i]: the column of C (contiguous in memory) sits on the lane
#pragma unroll
        for (int s = 0; s < GEMM_KT / 4; ++s) {
            const int q = 4 * s + (lane >> 4);
            const double a0 = As[q][32 * wr + (lane & 15)], a1 = As[q][32 * wr + 16 + (lane & 15)];
            const double b0 = Bs[q][32 * wc + (lane & 15)], b1 = Bs[q][32 * wc + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
    }
    // acc[x][y] element g: C[i = i0 + 32 wr + 16 x + (lane & 15)][j = j0 + 32 wc + 16 y + (lane >> 4) + 4 g]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = i0 + 32 * wr + 16 * x + (lane & 15), j = j0 + 32 * wc + 16 * y + (lane >> 4) + 4 * g;
                if (i < T.m && j < T.n) {
                    double *c = T.C + i + (size_t)T.ldc * j;
                    const double v = T.alpha * acc[x][y][g];
                    *c = (T.beta != 0.0) ? (T.beta * *c + v) : v;
                }
            }
}

// Cholesky of one diagonal block D (w x w, w <= 64, lower triangle of the front at `blk`, leading dimension ld) in place and its
// inverse Dinv = L_D^-1 (lower, zeros above) to `out` (leading dimension ldo).  One workgroup per task.  *fail <- task index + 1
// when a pivot is not positive.
struct PotrfTask { double *blk; double *out; int w, ld, ldo, id; };

__global__ __launch_bounds__(256) void potrf_inv_kernel(const PotrfTask *__restrict__ tasks, int *__restrict__ fail) {
    __shared__ double L[64][65];
    __shared__ double X[64][65];
    __shared__ int bad;
    const PotrfTask T = tasks[blockIdx.x];
    const int w = T.w, t = threadIdx.x;
    if (t == 0) bad = 0;
    for (int e = t; e < 64 * 64; e += 256) { const int i = e & 63, j = e >> 6; L[i][j] = (i < w && j <= i) ? T.blk[i + (size_t)T.ld * j] : 0.0; X[i][j] = 0.0; }
    __syncthreads();
    // right-looking, one column per step; thread (row i = t & 63, column phase t >> 6): no index arithmetic in the update
    const int ri = t & 63, cp = t >> 6;
    for (int j = 0; j < w; ++j) {
        const double d = L[j][j];                       // final since the previous step's barrier
        const bool ok = d > 0.0;
        const double dj = ok ? sqrt(d) : 1.0;
        const double lij = (ri > j && ri < w) ? L[ri][j] / dj : 0.0;
        __syncthreads();                                // everybody has read column j and the pivot
        if (cp == 0) { if (ri == j) { L[j][j] = dj; if (!ok) bad = 1; } else if (ri > j && ri < w) L[ri][j] = lij; }
        __syncthreads();
        if (ri > j && ri < w) for (int c = j + 1 + cp; c <= ri; c += 4) L[ri][c] -= lij * L[c][j];
        __syncthreads();
    }
    // Dinv by doubling inside the block (rows / columns >= w: identity): the four 16 x 16 diagonal blocks by forward substitution
    // (one thread per column, <= 120 dependent steps instead of 2016 for the whole block), then X21 = -X22 (L21 X11) for the
    // 32 x 32 and the 64 x 64 level with all 256 threads on the products.  T: the inner product, in the (unused) upper triangle of L.
    for (int i = w + t; i < 64; i += 256) L[i][i] = 1.0;
    __syncthreads();
    if (t < 64) {
        const int b = 16 * (t >> 4), c = b + (t & 15);
        X[c][c] = 1.0 / L[c][c];
        for (int i = c + 1; i < b + 16; ++i) {
            double s = 0.0;
            for (int m = c; m < i; ++m) s += L[i][m] * X[m][c];
            X[i][c] = -s / L[i][i];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int sz = 16; sz < 64; sz *= 2) {
        const int pairs = 32 / sz, per = sz * sz;          // pairs of (top, bottom) blocks of size sz; entries of one off-diagonal block
        for (int e = t; e < pairs * per; e += 256) {       // T = L21 X11, stored transposed at L[top rows][bottom cols] (strictly upper)
            const int p = e / per, q = e - p * per, i = q % sz, j = q / sz, g0 = 2 * sz * p;
            double s = 0.0;
            for (int m = j; m < sz; ++m) s += L[g0 + sz + i][g0 + m] * X[g0 + m][g0 + j];
            L[g0 + j][g0 + sz + i] = s;
        }
        __syncthreads();
        for (int e = t; e < pairs * per; e += 256) {       // X21 = -X22 T
            const int p = e / per, q = e - p * per, i = q % sz, j = q / sz, g0 = 2 * sz * p;
            double s = 0.0;
            for (int m = 0; m <= i; ++m) s += X[g0 + sz + i][g0 + sz + m] * L[g0 + j][g0 + sz + m];
            X[g0 + sz + i][g0 + j] = -s;
        }
        __syncthreads();
    }
    for (int e = t; e < 64 * 64; e += 256) {
        const int i = e & 63, j = e >> 6;
        if (i < w && j < w) { if (j <= i) T.blk[i + (size_t)T.ld * j] = L[i][j]; T.out[i + (size_t)T.ldo * j] = (j <= i) ? X[i][j] : 0.0; }
    }
    if (t == 0 && bad) atomicCAS(fail, 0, T.id + 1);
}

// F[dst[q]] = val[src[q]]: the original entries of A into the (zeroed) fronts
__global__ void assemble_kernel(int64_t nnz, const int64_t *__restrict__ dst, const int *__restrict__ src, const double *__restrict__ val, double *__restrict__ fronts) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) fronts[dst[q]] = val[src[q]];
}

// extend-add of one child's update matrix U (rc x rc lower, at `U`, leading dimension fc) into its parent's front (leading dimension fp):
// P[rel[a] + fp rel[b]] += U[a + fc b], a >= b.  blockIdx.y = task, blockIdx.x = group of 8 columns.
// packed != 0: U is the lower triangle stored column after column without gaps (column b = rows b .. rc - 1) -- the form in which the
// update matrices of the subtree roots travel between the ranks (rank-local factorization, pack_lower_kernel).
struct ExtendTask { const double *U; double *P; const int *rel; int rc, fc, fp, packed; };

__device__ __forceinline__ size_t packed_col(int rc, int b) { return (size_t)b * rc - (size_t)b * (b - 1) / 2; }      // first entry of column b

__global__ __launch_bounds__(256) void extend_add_kernel(const ExtendTask *__restrict__ tasks) {
    const ExtendTask T = tasks[blockIdx.y];
    const int b0 = blockIdx.x * 8;
    if (b0 >= T.rc) return;
    for (int b = b0; b < min(b0 + 8, T.rc); ++b) {
        const size_t pc = (size_t)T.fp * T.rel[b];
        const double *Ucol = T.packed ? T.U + packed_col(T.rc, b) - b : T.U + (size_t)T.fc * b;
        for (int a = b + threadIdx.x; a < T.rc; a += 256) T.P[T.rel[a] + pc] += Ucol[a];
    }
}

// out (packed lower triangle, see above) = the update matrix U (rc x rc lower, leading dimension fc) of a factored front
struct PackTask { const double *U; double *out; int rc, fc; };

__global__ __launch_bounds__(256) void pack_lower_kernel(const PackTask *__restrict__ tasks) {
    const PackTask T = tasks[blockIdx.y];
    const int b0 = blockIdx.x * 8;
    if (b0 >= T.rc) return;
    for (int b = b0; b < min(b0 + 8, T.rc); ++b) {
        const double *Ucol = T.U + (size_t)T.fc * b;
        double *o = T.out + packed_col(T.rc, b) - b;
        for (int a = b + threadIdx.x; a < T.rc; a += 256) o[a] = Ucol[a];
    }
}

} // namespace admm_dev
