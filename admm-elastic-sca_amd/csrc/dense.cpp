// dense.cpp -- host dense fp64 kernels behind the supernodal factorization.
// A packed 8x6 register-tile GEMM (GCC vector extensions; built with
// -mavx2 -mfma, falls back to SSE2 code generation elsewhere) plus the blocked
// partial Cholesky / triangular inverse / triangular product built on it.
// These run once per admm_hip_finalize(); the per-iteration work is on the GPU.
#include "dense.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace admm_host {

typedef double v4d __attribute__((vector_size(32), aligned(8)));

static const int MR = 8, NR = 6, KC = 256, MC = 192;

static inline v4d loadu(const double *p) { v4d v; std::memcpy(&v, p, 32); return v; }
static inline void storeu(double *p, v4d v) { std::memcpy(p, &v, 32); }

// acc[MR x NR] = sum_p a_p b_p^T over a packed A panel (kc x MR) and B panel (kc x NR)
static inline void micro_8x6(int kc, const double *__restrict Ap, const double *__restrict Bp, double *__restrict acc /*[NR][MR]*/) {
    v4d c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00;
    v4d c30 = c00, c31 = c00, c40 = c00, c41 = c00, c50 = c00, c51 = c00;
    for (int p = 0; p < kc; ++p) {
        v4d a0 = loadu(Ap), a1 = loadu(Ap + 4);
        v4d b;
        b = (v4d){Bp[0], Bp[0], Bp[0], Bp[0]}; c00 += a0 * b; c01 += a1 * b;
        b = (v4d){Bp[1], Bp[1], Bp[1], Bp[1]}; c10 += a0 * b; c11 += a1 * b;
        b = (v4d){Bp[2], Bp[2], Bp[2], Bp[2]}; c20 += a0 * b; c21 += a1 * b;
        b = (v4d){Bp[3], Bp[3], Bp[3], Bp[3]}; c30 += a0 * b; c31 += a1 * b;
        b = (v4d){Bp[4], Bp[4], Bp[4], Bp[4]}; c40 += a0 * b; c41 += a1 * b;
        b = (v4d){Bp[5], Bp[5], Bp[5], Bp[5]}; c50 += a0 * b; c51 += a1 * b;
        Ap += MR; Bp += NR;
    }
    storeu(acc + 0, c00); storeu(acc + 4, c01); storeu(acc + 8, c10); storeu(acc + 12, c11);
    storeu(acc + 16, c20); storeu(acc + 20, c21); storeu(acc + 24, c30); storeu(acc + 28, c31);
    storeu(acc + 32, c40); storeu(acc + 36, c41); storeu(acc + 40, c50); storeu(acc + 44, c51);
}

// pack rows [0,m) x cols [0,kc) of col-major A into MR-row panels
static void pack_A(int m, int kc, const double *A, int lda, double *out) {
    for (int i0 = 0; i0 < m; i0 += MR) {
        int mr = std::min(MR, m - i0);
        for (int p = 0; p < kc; ++p) {
            const double *src = A + i0 + (size_t)lda * p;
            for (int i = 0; i < mr; ++i) out[i] = src[i];
            for (int i = mr; i < MR; ++i) out[i] = 0.0;
            out += MR;
        }
    }
}
// B given as n x k (use b_p[j] = B[j,p]) -> NR-column panels
static void pack_Bt(int n, int kc, const double *B, int ldb, double *out) {
    for (int j0 = 0; j0 < n; j0 += NR) {
        int nr = std::min(NR, n - j0);
        for (int p = 0; p < kc; ++p) {
            const double *src = B + j0 + (size_t)ldb * p;
            for (int j = 0; j < nr; ++j) out[j] = src[j];
            for (int j = nr; j < NR; ++j) out[j] = 0.0;
            out += NR;
        }
    }
}
// B given as k x n (b_p[j] = B[p,j])
static void pack_Bn(int n, int kc, const double *B, int ldb, double *out) {
    for (int j0 = 0; j0 < n; j0 += NR) {
        int nr = std::min(NR, n - j0);
        for (int p = 0; p < kc; ++p) {
            for (int j = 0; j < nr; ++j) out[j] = B[p + (size_t)ldb * (j0 + j)];
            for (int j = nr; j < NR; ++j) out[j] = 0.0;
            out += NR;
        }
    }
}

enum Mode { SUB, SET, ADD };

// one (m x n) block, K already limited to kc <= KC: C op= A*B with packed B
static void block_kernel(int m, int n, int kc, const double *A, int lda, const double *Bpack, double *C, int ldc, Mode mode, double *Apack) {
    for (int i0 = 0; i0 < m; i0 += MC) {
        int mc = std::min(MC, m - i0);
        pack_A(mc, kc, A + i0, lda, Apack);
        for (int j0 = 0; j0 < n; j0 += NR) {
            int nr = std::min(NR, n - j0);
            const double *bp = Bpack + (size_t)(j0 / NR) * kc * NR;
            for (int ii = 0; ii < mc; ii += MR) {
                int mr = std::min(MR, mc - ii);
                double acc[MR * NR];
                micro_8x6(kc, Apack + (size_t)(ii / MR) * kc * MR, bp, acc);
                double *c = C + (i0 + ii) + (size_t)ldc * j0;
                if (mode == SUB) { for (int j = 0; j < nr; ++j) for (int i = 0; i < mr; ++i) c[i + (size_t)ldc * j] -= acc[i + MR * j]; }
                else if (mode == ADD) { for (int j = 0; j < nr; ++j) for (int i = 0; i < mr; ++i) c[i + (size_t)ldc * j] += acc[i + MR * j]; }
                else { for (int j = 0; j < nr; ++j) for (int i = 0; i < mr; ++i) c[i + (size_t)ldc * j] = acc[i + MR * j]; }
            }
        }
    }
}

static void gemm_driver(bool b_transposed, Mode mode, int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int threads) {
    if (m <= 0 || n <= 0) return;
    if (k <= 0) {
        if (mode == SET) for (int j = 0; j < n; ++j) std::memset(C + (size_t)ldc * j, 0, sizeof(double) * m);
        return;
    }
    const int NB = 96; // column block per task (multiple of NR)
    const int nblk = (n + NB - 1) / NB;
    auto run_block = [&](int jb) {
        static thread_local std::vector<double> Apack, Bpack;
        if (Apack.empty()) { Apack.resize((size_t)MC * KC + 64); Bpack.resize((size_t)(NB + NR) * KC + 64); }
        int j0 = jb * NB, nb = std::min(NB, n - j0);
        for (int p0 = 0; p0 < k; p0 += KC) {
            int kc = std::min(KC, k - p0);
            if (b_transposed) pack_Bt(nb, kc, B + j0 + (size_t)ldb * p0, ldb, Bpack.data());
            else pack_Bn(nb, kc, B + p0 + (size_t)ldb * j0, ldb, Bpack.data());
            Mode md = mode;
            if (mode == SET && p0 > 0) md = ADD;
            block_kernel(m, nb, kc, A + (size_t)lda * p0, lda, Bpack.data(), C + (size_t)ldc * j0, ldc, md, Apack.data());
        }
    };
    if (threads > 1 && nblk > 1) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int jb = 0; jb < nblk; ++jb) run_block(jb);
    } else {
        for (int jb = 0; jb < nblk; ++jb) run_block(jb);
    }
}

void gemm_nt_sub(int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int threads) {
    gemm_driver(true, SUB, m, n, k, A, lda, B, ldb, C, ldc, threads);
}
void gemm_nn_set(int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int threads) {
    gemm_driver(false, SET, m, n, k, A, lda, B, ldb, C, ldc, threads);
}

// unblocked Cholesky of an nb x nb diagonal block followed by the triangular
// solve of the rows below it (rows [nb, m) of the same column panel)
static int panel_factor(int m, int nb, double *P, int ld, int threads) {
    for (int c = 0; c < nb; ++c) {
        double d = P[c + (size_t)ld * c];
        if (!(d > 0.0)) return c + 1;
        d = std::sqrt(d);
        P[c + (size_t)ld * c] = d;
        const double inv = 1.0 / d;
        double *col = P + (size_t)ld * c;
        for (int i = c + 1; i < nb; ++i) col[i] *= inv;
        // update the remaining columns of the diagonal block
        for (int j = c + 1; j < nb; ++j) {
            const double l = col[j];
            double *cj = P + (size_t)ld * j;
            for (int i = j; i < nb; ++i) cj[i] -= col[i] * l;
        }
    }
    // rows below: X = A21 * L11^-T, column by column (contiguous axpys)
    const int rows = m - nb;
    if (rows > 0) {
        const int chunk = 512;
        const int nch = (rows + chunk - 1) / chunk;
        auto run_chunk = [&](int ch) {
            int r0 = nb + ch * chunk, r1 = std::min(m, r0 + chunk);
            for (int c = 0; c < nb; ++c) {
                double *col = P + (size_t)ld * c;
                for (int p = 0; p < c; ++p) {
                    const double l = P[c + (size_t)ld * p];
                    const double *cp = P + (size_t)ld * p;
                    for (int i = r0; i < r1; ++i) col[i] -= cp[i] * l;
                }
                const double inv = 1.0 / P[c + (size_t)ld * c];
                for (int i = r0; i < r1; ++i) col[i] *= inv;
            }
        };
        if (threads > 1 && nch > 1) {
#pragma omp parallel for num_threads(threads) schedule(static)
            for (int ch = 0; ch < nch; ++ch) run_chunk(ch);
        } else {
            for (int ch = 0; ch < nch; ++ch) run_chunk(ch);
        }
    }
    return 0;
}

int partial_cholesky(int f, int k, double *F, int ldf, int threads) {
    const int NBP = 96;
    for (int j0 = 0; j0 < k; j0 += NBP) {
        int jb = std::min(NBP, k - j0);
        int err = panel_factor(f - j0, jb, F + j0 + (size_t)ldf * j0, ldf, threads);
        if (err) return j0 + err;
        int t0 = j0 + jb, tm = f - t0;
        if (tm <= 0) continue;
        // trailing update, lower triangle by column blocks: C[J.., J] -= P[J..,:] P[J,:]^T
        const double *P = F + (size_t)ldf * j0;
        const int CB = 192;
        const int nblk = (tm + CB - 1) / CB;
        if (threads > 1 && nblk > 1) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
            for (int b = nblk - 1; b >= 0; --b) {
                int c0 = t0 + b * CB, cb = std::min(CB, f - c0);
                gemm_nt_sub(f - c0, cb, jb, P + c0, ldf, P + c0, ldf, F + c0 + (size_t)ldf * c0, ldf, 1);
            }
        } else {
            for (int b = 0; b < nblk; ++b) {
                int c0 = t0 + b * CB, cb = std::min(CB, f - c0);
                gemm_nt_sub(f - c0, cb, jb, P + c0, ldf, P + c0, ldf, F + c0 + (size_t)ldf * c0, ldf, 1);
            }
        }
    }
    return 0;
}

void trtri_lower(int k, const double *L, int ldl, double *X, int ldx, int threads) {
    auto run_col = [&](int j) {
        double *x = X + (size_t)ldx * j;
        for (int i = 0; i < j; ++i) x[i] = 0.0;
        x[j] = 1.0;
        for (int i = j + 1; i < k; ++i) x[i] = 0.0;
        for (int p = j; p < k; ++p) {
            const double xp = x[p] / L[p + (size_t)ldl * p];
            x[p] = xp;
            const double *lp = L + (size_t)ldl * p;
            for (int i = p + 1; i < k; ++i) x[i] -= lp[i] * xp;
        }
    };
    if (threads > 1 && k > 64) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
        for (int j = 0; j < k; ++j) run_col(j);
    } else {
        for (int j = 0; j < k; ++j) run_col(j);
    }
}

void trmm_right_lower(int r, int k, const double *A, int lda, const double *T, int ldt, double *Z, int ldz, int threads) {
    if (r <= 0) return;
    const int JB = 96;
    const int nblk = (k + JB - 1) / JB;
    auto run = [&](int b) {
        int j0 = b * JB, jb = std::min(JB, k - j0);
        gemm_nn_set(r, jb, k - j0, A + (size_t)lda * j0, lda, T + j0 + (size_t)ldt * j0, ldt, Z + (size_t)ldz * j0, ldz, 1);
    };
    if (threads > 1 && nblk > 1) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int b = 0; b < nblk; ++b) run(b);
    } else {
        for (int b = 0; b < nblk; ++b) run(b);
    }
}

} // namespace admm_host
