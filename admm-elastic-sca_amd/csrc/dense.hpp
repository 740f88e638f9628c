// dense.hpp -- small dense fp64 kernels for the host-side supernodal
// factorization (column-major, lower triangles).  Implemented in dense.cpp.
#pragma once
#include <cstddef>

namespace admm_host {

// C (m x n, ldc) -= A (m x k, lda) * B(n x k, ldb)^T ; `threads` > 1 lets the
// call split its column blocks over an OpenMP team.
void gemm_nt_sub(int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int threads);
// C (m x n) = A (m x k) * B (k x n)   (overwrites C)
void gemm_nn_set(int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int threads);

// Partial Cholesky of the leading k columns of the symmetric f x f front F
// (lower triangle stored, ld = ldf):  F11 = L11 L11^T, F21 <- F21 L11^-T,
// F22 <- F22 - L21 L21^T (lower).  Returns 0, or j+1 if pivot j is not positive.
int partial_cholesky(int f, int k, double *F, int ldf, int threads);

// X (k x k lower, ldx) = inverse of the lower-triangular L (k x k, ldl);
// the strict upper triangle of X is set to zero.
void trtri_lower(int k, const double *L, int ldl, double *X, int ldx, int threads);

// Z (r x k, ldz) = A (r x k, lda) * T (k x k lower triangular, ldt)
void trmm_right_lower(int r, int k, const double *A, int lda, const double *T, int ldt, double *Z, int ldz, int threads);

} // namespace admm_host
