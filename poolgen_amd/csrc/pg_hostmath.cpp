// pg_hostmath.cpp -- small dense host-side linear algebra for the n x n (n = pools) problems
// that sit between the two GPU passes: symmetric eigen-decomposition of the kinship matrix
// (reference: `kinship.eig()`, gwas/ols.rs:296), an orthonormal basis of [1 | C], the
// pseudo-inverse used by gp::ols (base/helpers.rs:463-482) and the t-distribution series
// coefficients consumed by the device p-value code.  These are O(n^3) with n <= a few hundred;
// the O(n^2 p) work lives in the HIP kernels.
#include "pg_common.h"
#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

// The two inner loops of the Householder reduction, in a wide-vector and a baseline build chosen once at run time (the
// library is built for plain x86-64; the n x n step between the GPU passes is on the critical path of every m >= 1 step).
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#define PG_HOST_AVX2 __attribute__((target("avx2,fma")))
#else
#define PG_HOST_AVX2
#endif
// row j of the lower triangle: p_j += row[0..j) . u[0..j) + row[j] u_j;  p[0..j) += row[0..j) u_j
template <int DUMMY>
static inline void symv_row_body(const double *__restrict__ row, const double *__restrict__ u, double *__restrict__ p, int j) {
    const double uj = u[j];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= j; k += 4) {
        s0 += row[k] * u[k]; s1 += row[k + 1] * u[k + 1]; s2 += row[k + 2] * u[k + 2]; s3 += row[k + 3] * u[k + 3];
        p[k] += row[k] * uj; p[k + 1] += row[k + 1] * uj; p[k + 2] += row[k + 2] * uj; p[k + 3] += row[k + 3] * uj;
    }
    for (; k < j; ++k) { s0 += row[k] * u[k]; p[k] += row[k] * uj; }
    p[j] += ((s0 + s1) + (s2 + s3)) + row[j] * uj;
}
static void symv_row_base(const double *row, const double *u, double *p, int j) { symv_row_body<0>(row, u, p, j); }
PG_HOST_AVX2 static void symv_row_avx2(const double *row, const double *u, double *p, int j) { symv_row_body<1>(row, u, p, j); }
// row j of the rank-2 update of the lower triangle: row[0..j] -= uj q[0..j] + qj u[0..j]
template <int DUMMY>
static inline void rank2_row_body(double *__restrict__ row, const double *__restrict__ u, const double *__restrict__ q, double uj, double qj, int j) {
    for (int k = 0; k <= j; ++k) row[k] -= (uj * q[k] + qj * u[k]);
}
static void rank2_row_base(double *row, const double *u, const double *q, double uj, double qj, int j) { rank2_row_body<0>(row, u, q, uj, qj, j); }
PG_HOST_AVX2 static void rank2_row_avx2(double *row, const double *u, const double *q, double uj, double qj, int j) { rank2_row_body<1>(row, u, q, uj, qj, j); }
static bool host_has_avx2() {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return yes;
#else
    return false;
#endif
}
// row[t0..t1) -= a * col[t0..t1): the inner loop of the right-looking Cholesky below (element-wise: any vector width gives the same bits)
template <int DUMMY>
static inline void axpy_sub_body(double *__restrict__ row, const double *__restrict__ col, double a, int t0, int t1) {
    for (int t = t0; t < t1; ++t) row[t] -= a * col[t];
}
static void axpy_sub_base(double *row, const double *col, double a, int t0, int t1) { axpy_sub_body<0>(row, col, a, t0, t1); }
PG_HOST_AVX2 static void axpy_sub_avx2(double *row, const double *col, double a, int t0, int t1) { axpy_sub_body<1>(row, col, a, t0, t1); }
static inline void symv_row(const double *row, const double *u, double *p, int j) {
    if (host_has_avx2()) symv_row_avx2(row, u, p, j); else symv_row_base(row, u, p, j);
}
static inline void rank2_row(double *row, const double *u, const double *q, double uj, double qj, int j) {
    if (host_has_avx2()) rank2_row_avx2(row, u, q, uj, qj, j); else rank2_row_base(row, u, q, uj, qj, j);
}

// Householder reduction of a symmetric matrix to tridiagonal form.  On exit (want_q) `a` holds
// the orthogonal transformation, d the diagonal, e the sub-diagonal (e[0] = 0).
void tridiagonalise(std::vector<double> &a, int n, std::vector<double> &d, std::vector<double> &e,
                    bool want_q, std::vector<double> *hh = nullptr) {
    if (hh) hh->assign(n, 0.0);
    auto A = [&](int i, int j) -> double & { return a[(size_t)i * n + j]; };
    std::vector<double> gv;
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += std::fabs(A(i, k));
            if (scale == 0.0) {
                e[i] = A(i, l);
            } else {
                for (int k = 0; k <= l; ++k) {
                    A(i, k) /= scale;
                    h += A(i, k) * A(i, k);
                }
                double f = A(i, l);
                double g = (f >= 0.0) ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                A(i, l) = f - g;
                f = 0.0;
                if (want_q) {
                    for (int j = 0; j <= l; ++j) {
                        A(j, i) = A(i, j) / h;
                        g = 0.0;
                        for (int k = 0; k <= j; ++k) g += A(j, k) * A(i, k);
                        for (int k = j + 1; k <= l; ++k) g += A(k, j) * A(i, k);
                        e[j] = g / h;
                        f += e[j] * A(i, j);
                    }
                } else {
                    // p = A u over the leading block from its LOWER triangle, row by row: row j gives the dot product of its
                    // sub-diagonal part with u (into p_j) and, by symmetry, its multiples of u_j (into p_0 .. p_{j-1}) -- unit
                    // stride both ways (the column walk of the form above, A(k, j) for k > j, is what the solve spent its time in)
                    const double *u = &A(i, 0);
                    for (int j = 0; j <= l; ++j) e[j] = 0.0;
                    for (int j = 0; j <= l; ++j) symv_row(&A(j, 0), u, e.data(), j);
                    for (int j = 0; j <= l; ++j) {
                        e[j] /= h;
                        f += e[j] * u[j];
                    }
                }
                const double hh = f / (h + h);
                if (want_q) {
                    for (int j = 0; j <= l; ++j) {
                        f = A(i, j);
                        e[j] = g = e[j] - hh * f;
                        for (int k = 0; k <= j; ++k) A(j, k) -= (f * e[k] + g * A(i, k));
                    }
                } else {
                    const double *u = &A(i, 0);
                    for (int j = 0; j <= l; ++j) e[j] -= hh * u[j];
                    for (int j = 0; j <= l; ++j) rank2_row(&A(j, 0), u, e.data(), u[j], e[j], j);
                }
            }
        } else {
            e[i] = A(i, l);
        }
        d[i] = h;
        if (hh) (*hh)[i] = h; // P_i = I - u_i u_i^T / h_i with u_i = row i of `a`, columns 0..i-1
    }
    if (want_q) d[0] = 0.0;
    e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        if (want_q) {
            const int l = i - 1;
            if (d[i] != 0.0) {
                // g_j = sum_k A(i,k) A(k,j) for every j, then A(k,j) -= g_j A(k,i): the same sums in the same
                // order as the column-at-a-time form, but walking rows (unit stride) instead of columns
                gv.assign((size_t)l + 1, 0.0);
                for (int k = 0; k <= l; ++k) {
                    const double aik = A(i, k);
                    const double *row = &A(k, 0);
                    for (int j = 0; j <= l; ++j) gv[j] += aik * row[j];
                }
                for (int k = 0; k <= l; ++k) {
                    const double aki = A(k, i);
                    double *row = &A(k, 0);
                    for (int j = 0; j <= l; ++j) row[j] -= gv[j] * aki;
                }
            }
            d[i] = A(i, i);
            A(i, i) = 1.0;
            for (int j = 0; j <= l; ++j) A(j, i) = A(i, j) = 0.0;
        } else {
            d[i] = A(i, i);
        }
    }
}

// Implicit-shift QL iteration on a symmetric tridiagonal matrix.  z (n x n, row-major) is
// post-multiplied by the rotations when want_q.
int ql_implicit(std::vector<double> &d, std::vector<double> &e, int n, std::vector<double> &z,
                bool want_q) {
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    // The rotations mix COLUMNS i and i+1 of z.  Held transposed, those are two contiguous rows: unit-stride,
    // vectorisable updates instead of n cache lines touched for two doubles each (3 n^3 flops of the solve).
    std::vector<double> zt;
    if (want_q) {
        zt.resize((size_t)n * n);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) zt[(size_t)c * n + r] = z[(size_t)r * n + c];
    }
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return -1;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::sqrt(g * g + 1.0); // (no overflow to guard against: g is a ratio of kinship-sized numbers, and |g| > 1e150 would be an error upstream)
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    e[i + 1] = (r = std::sqrt(f * f + g * g)); // std::hypot costs more than the rest of the rotation
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    if (want_q) {
                        double *__restrict__ z0 = &zt[(size_t)i * n], *__restrict__ z1 = &zt[(size_t)(i + 1) * n];
                        for (int k = 0; k < n; ++k) {
                            const double f1 = z1[k], f0 = z0[k];
                            z1[k] = s * f0 + c * f1;
                            z0[k] = c * f0 - s * f1;
                        }
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    if (want_q)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) z[(size_t)r * n + c] = zt[(size_t)c * n + r];
    return 0;
}

} // namespace

int pg_sym_eig(const double *A, int n, double *evals, double *V, bool want_vectors) {
    std::vector<double> a(A, A + (size_t)n * n), d(n), e(n);
    if (n == 1) {
        evals[0] = A[0];
        if (want_vectors) V[0] = 1.0;
        return 0;
    }
    tridiagonalise(a, n, d, e, want_vectors);
    if (ql_implicit(d, e, n, a, want_vectors) != 0) return -1;
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] > d[y]; });
    for (int j = 0; j < n; ++j) evals[j] = d[order[j]];
    if (want_vectors) {
        for (int j = 0; j < n; ++j) {
            const int src = order[j];
            // fix the sign so that the result does not depend on rotation history: largest
            // |component| positive (the regression only depends on the span).
            double big = 0.0;
            for (int i = 0; i < n; ++i)
                if (std::fabs(a[(size_t)i * n + src]) > std::fabs(big)) big = a[(size_t)i * n + src];
            const double sg = big < 0.0 ? -1.0 : 1.0;
            for (int i = 0; i < n; ++i) V[(size_t)i * n + j] = sg * a[(size_t)i * n + src];
        }
    }
    return 0;
}

// All eigenvalues (descending) and only the m leading eigenvectors (columns of V, n x m row-major): values by
// QL on the tridiagonal form without accumulating the transformation (O(n^2) after the O(n^3) reduction), vectors
// by inverse iteration on the tridiagonal matrix and back-transformation through the Householder reflectors --
// what the kinship covariates need, at a third of the cost of the full decomposition.  The result is VERIFIED
// (residual and orthogonality against A itself); anything doubtful falls back to the full QL solve.
int pg_sym_eig_top(const double *A, int n, int m, double *evals, double *V) {
    if (m <= 0) return pg_sym_eig(A, n, evals, nullptr, false);
    auto full = [&]() {
        std::vector<double> Vf((size_t)n * n);
        if (pg_sym_eig(A, n, evals, Vf.data(), true) != 0) return -1;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) V[(size_t)i * m + j] = Vf[(size_t)i * n + j];
        return 0;
    };
    if (n < 8 || m > n / 2) return full();
    std::vector<double> a(A, A + (size_t)n * n), d(n), e(n), hh;
    tridiagonalise(a, n, d, e, false, &hh);
    std::vector<double> dq(d), eq(e), none;
    if (ql_implicit(dq, eq, n, none, false) != 0) return -1;
    std::sort(dq.begin(), dq.end(), std::greater<double>());
    for (int j = 0; j < n; ++j) evals[j] = dq[j];
    double tnorm = 0.0; // the tridiagonal matrix: diagonal d[0..n), sub-diagonal e[1..n) (e[0] = 0)
    for (int i = 0; i < n; ++i) tnorm = std::max(tnorm, std::fabs(d[i]) + std::fabs(e[i]) + (i + 1 < n ? std::fabs(e[i + 1]) : 0.0));
    if (!(tnorm > 0.0)) return full();
    const double eps = 2.220446049250313e-16;
    std::vector<double> Y((size_t)m * n), x(n), u0(n), u1(n), u2(n), rhs(n);
    std::vector<int> swapped(n);
    for (int c = 0; c < m; ++c) {
        double lam = evals[c];
        for (int b = 0; b < c; ++b) // nudge repeated eigenvalues apart so that the solves differ
            if (std::fabs(evals[b] - lam) <= 10.0 * eps * tnorm) lam -= 10.0 * eps * tnorm;
        // LU of (T - lam I) with partial pivoting: rows become (u0, u1, u2) = diagonal and two super-diagonals
        double pd = d[0] - lam, pe = n > 1 ? e[1] : 0.0; // current row: diagonal, super-diagonal
        std::vector<double> lmul(n, 0.0);
        for (int i = 0; i + 1 < n; ++i) {
            const double sub = e[i + 1], nd = d[i + 1] - lam, ne = (i + 2 < n) ? e[i + 2] : 0.0;
            if (std::fabs(sub) > std::fabs(pd)) { // swap row i and i + 1
                swapped[i] = 1;
                u0[i] = sub; u1[i] = nd; u2[i] = ne;
                const double f = pd / sub;
                lmul[i] = f;
                pd = pe - f * nd; pe = -f * ne;
            } else {
                swapped[i] = 0;
                if (pd == 0.0) pd = eps * tnorm;
                u0[i] = pd; u1[i] = pe; u2[i] = 0.0;
                const double f = sub / pd;
                lmul[i] = f;
                pd = nd - f * pe; pe = ne;
            }
        }
        if (pd == 0.0) pd = eps * tnorm;
        u0[n - 1] = pd; u1[n - 1] = 0.0; u2[n - 1] = 0.0;
        for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.01 * ((i * 7919 + c * 104729) % 97) / 97.0; // a vector with a component everywhere
        for (int it = 0; it < 4; ++it) {
            if (it > 0) { // forward substitution with the recorded row operations
                for (int i = 0; i + 1 < n; ++i) {
                    if (swapped[i]) { const double t = x[i]; x[i] = x[i + 1]; x[i + 1] = t - lmul[i] * x[i]; }
                    else x[i + 1] -= lmul[i] * x[i];
                }
            }
            for (int i = n - 1; i >= 0; --i) { // back substitution
                double sacc = x[i];
                if (i + 1 < n) sacc -= u1[i] * x[i + 1];
                if (i + 2 < n) sacc -= u2[i] * x[i + 2];
                x[i] = sacc / u0[i];
            }
            for (int b = 0; b < c; ++b) { // keep close eigenvalues' vectors orthogonal
                if (std::fabs(evals[b] - evals[c]) > 1e-3 * tnorm) continue;
                const double *yb = &Y[(size_t)b * n];
                double dot = 0.0;
                for (int i = 0; i < n; ++i) dot += yb[i] * x[i];
                for (int i = 0; i < n; ++i) x[i] -= dot * yb[i];
            }
            double nrm = 0.0;
            for (int i = 0; i < n; ++i) nrm += x[i] * x[i];
            nrm = std::sqrt(nrm);
            if (!(nrm > 0.0) || !std::isfinite(nrm)) return full();
            for (int i = 0; i < n; ++i) x[i] /= nrm;
        }
        std::copy(x.begin(), x.end(), Y.begin() + (size_t)c * n);
    }
    // back-transformation: T = P_1 ... P_{n-1} A P_{n-1} ... P_1, so A = Q T Q^T with Q = P_{n-1} ... P_2 P_1 and an
    // eigenvector of A is Q y: P_1 is applied first, P_{n-1} last
    for (int c = 0; c < m; ++c) {
        double *y = &Y[(size_t)c * n];
        for (int i = 1; i < n; ++i) {
            if (hh[i] == 0.0) continue;
            const double *ui = &a[(size_t)i * n];
            double dot = 0.0;
            for (int k2 = 0; k2 < i; ++k2) dot += ui[k2] * y[k2];
            dot /= hh[i];
            for (int k2 = 0; k2 < i; ++k2) y[k2] -= dot * ui[k2];
        }
    }
    // verification against A itself
    double worst = 0.0, a1 = std::fabs(evals[0]) > 0.0 ? std::fabs(evals[0]) : 1.0;
    for (int c = 0; c < m; ++c) {
        const double *y = &Y[(size_t)c * n];
        for (int i = 0; i < n; ++i) {
            double sacc = 0.0;
            const double *row = &A[(size_t)i * n];
            for (int j = 0; j < n; ++j) sacc += row[j] * y[j];
            worst = std::max(worst, std::fabs(sacc - evals[c] * y[i]) / a1);
        }
        for (int b = 0; b <= c; ++b) {
            const double *yb = &Y[(size_t)b * n];
            double dot = 0.0;
            for (int i = 0; i < n; ++i) dot += yb[i] * y[i];
            worst = std::max(worst, std::fabs(dot - (b == c ? 1.0 : 0.0)) * 1e-2); // orthogonality to 1e-10
        }
    }
    if (!(worst <= 1e-12 * n)) return full();
    for (int c = 0; c < m; ++c) {
        const double *y = &Y[(size_t)c * n];
        double big = 0.0; // same sign convention as pg_sym_eig: largest |component| positive
        for (int i = 0; i < n; ++i)
            if (std::fabs(y[i]) > std::fabs(big)) big = y[i];
        const double sg = big < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) V[(size_t)i * m + c] = sg * y[i];
    }
    return 0;
}

// Orthonormal basis of the columns of Z by twice-iterated classical Gram-Schmidt.  Columns
// that are numerically dependent on the previous ones (residual < 1e-10 of their norm) are
// dropped; returns the number of basis vectors written to Q (n x c row-major, first `rank`
// columns valid, the rest zero).
int pg_thin_qr(const double *Z, int n, int c, double *Q) {
    std::vector<double> v(n);
    int rank = 0;
    std::fill(Q, Q + (size_t)n * c, 0.0);
    for (int j = 0; j < c; ++j) {
        double norm0 = 0.0;
        for (int i = 0; i < n; ++i) {
            v[i] = Z[(size_t)i * c + j];
            norm0 += v[i] * v[i];
        }
        norm0 = std::sqrt(norm0);
        if (norm0 == 0.0) continue;
        for (int pass = 0; pass < 2; ++pass) {
            for (int q = 0; q < rank; ++q) {
                double dot = 0.0;
                for (int i = 0; i < n; ++i) dot += Q[(size_t)i * c + q] * v[i];
                for (int i = 0; i < n; ++i) v[i] -= dot * Q[(size_t)i * c + q];
            }
        }
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += v[i] * v[i];
        nrm = std::sqrt(nrm);
        if (!(nrm > 1e-10 * norm0)) continue;
        for (int i = 0; i < n; ++i) Q[(size_t)i * c + rank] = v[i] / nrm;
        ++rank;
    }
    return rank;
}

// Two-sided Student-t probability for INTEGER df as a finite trigonometric series
// (Abramowitz & Stegun 26.7.3 / 26.7.4).  With theta = atan(|t|/sqrt(df)), c2 = cos^2(theta):
//   df odd : A = 2/pi * (theta + sin*cos * sum_{j=0}^{(df-3)/2} coef[j] c2^j), coef[j] = prod (2i)/(2i+1)
//   df even: A = sin * sum_{j=0}^{(df-2)/2} coef[j] c2^j,                    coef[j] = prod (2i-1)/(2i)
// p = 1 - A.  The device evaluates the polynomial by Horner's rule from these coefficients.
std::vector<double> pg_tdist_coef(int df) {
    std::vector<double> c;
    if (df < 1) return c;
    if (df % 2 == 1) {
        const int terms = (df - 1) / 2;
        double v = 1.0;
        for (int j = 0; j < terms; ++j) {
            if (j > 0) v = v * (2.0 * j) / (2.0 * j + 1.0);
            c.push_back(v);
        }
    } else {
        const int terms = df / 2;
        double v = 1.0;
        for (int j = 0; j < terms; ++j) {
            if (j > 0) v = v * (2.0 * j - 1.0) / (2.0 * j);
            c.push_back(v);
        }
    }
    while (c.size() % 8) c.push_back(0.0); // the device evaluates the series in blocks of 8 coefficients (pg_stats_device.h)
    return c;
}

// pinv of a symmetric matrix with the reference's SVD tolerance eps * len(s) * max(s)
// (base/helpers.rs:463-482); for a symmetric matrix singular values are |eigenvalues|.
int pg_pinv_sym(const double *A, int n, double *out) {
    std::vector<double> ev(n), V((size_t)n * n);
    if (pg_sym_eig(A, n, ev.data(), V.data(), true) != 0) return -1;
    double smax = 0.0;
    for (int i = 0; i < n; ++i) smax = std::max(smax, std::fabs(ev[i]));
    const double tol = 2.220446049250313e-16 * (double)n * smax;
    std::fill(out, out + (size_t)n * n, 0.0);
    for (int e = 0; e < n; ++e) {
        if (std::fabs(ev[e]) > tol) {
            const double w = 1.0 / ev[e];
            for (int i = 0; i < n; ++i) {
                const double vi = V[(size_t)i * n + e] * w;
                for (int j = 0; j < n; ++j) out[(size_t)i * n + j] += vi * V[(size_t)j * n + e];
            }
        }
    }
    return 0;
}

// X = pinv(A) B for a symmetric A (n x n) and B (n x k).  A training subset's X X^T is symmetric positive
// definite unless pools are duplicated, and then pinv(A) = A^-1: a Cholesky factorisation (n^3/3 flops, inner
// loops over contiguous rows) answers in milliseconds where the eigen-decomposition behind pinv takes 0.1 s at
// n = 450 -- which was 98 % of the wall time of the ridge lambda path.  The truncation semantics of the
// reference's pinv (helpers.rs:463-482: singular values below eps * n * s_max are dropped) only matter for a
// (numerically) singular A: any pivot below 1e-10 of the largest diagonal entry sends the call to the
// eigen-based pg_pinv_sym instead.
int pg_pinv_solve_sym(const double *A, int n, const double *B, int k, double *X) {
    std::vector<double> L((size_t)n * n, 0.0), colj(n);
    double dmax = 0.0;
    for (int i = 0; i < n; ++i) dmax = std::max(dmax, std::fabs(A[(size_t)i * n + i]));
    bool spd = dmax > 0.0;
    // Right-looking (outer-product) form: column j is finished, then every later row i takes row_i[t] -= l_ij l_tj for j < t <= i --
    // an element-wise update of a contiguous row by a contiguous copy of the column, which the compiler vectorises (the row-by-row
    // dot-product form it replaces is a chain of dependent subtractions: 5 ms per 450 x 450 fold of config 4, and the device waits
    // for the first repetition's folds).  Every element receives the same subtractions in the same order (t ascending) as in the
    // dot-product form: the factor is the same to the last bit.
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) L[(size_t)i * n + j] = A[(size_t)i * n + j];
    const bool wide = host_has_avx2();
    for (int j = 0; j < n && spd; ++j) {
        const double s = L[(size_t)j * n + j];
        if (!(s > 1e-10 * dmax)) { spd = false; break; }
        const double d = std::sqrt(s);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            const double v = L[(size_t)i * n + j] / d;
            L[(size_t)i * n + j] = v;
            colj[i] = v;
        }
        for (int i = j + 1; i < n; ++i) {
            if (wide) axpy_sub_avx2(&L[(size_t)i * n], colj.data(), colj[i], j + 1, i + 1);
            else axpy_sub_base(&L[(size_t)i * n], colj.data(), colj[i], j + 1, i + 1);
        }
    }
    if (spd) {
        std::vector<double> y(n);
        for (int c = 0; c < k; ++c) {
            for (int i = 0; i < n; ++i) { // L y = b
                const double *Li = &L[(size_t)i * n];
                double s = B[(size_t)i * k + c];
                for (int t = 0; t < i; ++t) s -= Li[t] * y[t];
                y[i] = s / Li[i];
            }
            for (int i = n - 1; i >= 0; --i) { // L^T x = y
                double s = y[i];
                for (int t = i + 1; t < n; ++t) s -= L[(size_t)t * n + i] * y[t];
                y[i] = s / L[(size_t)i * n + i];
            }
            for (int i = 0; i < n; ++i) X[(size_t)i * k + c] = y[i];
        }
        return 0;
    }
    std::vector<double> Pi((size_t)n * n);
    if (pg_pinv_sym(A, n, Pi.data()) != 0) return -1;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += Pi[(size_t)i * n + j] * B[(size_t)j * k + c];
            X[(size_t)i * k + c] = s;
        }
    return 0;
}

// ---- exported host utilities (include/poolgen_hip.h, "Host-side pieces of the path") ----------
extern "C" int pg_host_sym_eig(const double *A, int n, double *evals, double *V) {
    if (!A || !evals || n < 1) return PG_ERR_INVALID;
    return pg_sym_eig(A, n, evals, V, V != nullptr) == 0 ? PG_OK : PG_ERR_INVALID;
}

extern "C" int pg_host_sym_eig_top(const double *A, int n, int m, double *evals, double *V) {
    if (!A || !evals || n < 1 || m < 0 || m > n || (m > 0 && !V)) return PG_ERR_INVALID;
    return pg_sym_eig_top(A, n, m, evals, V) == 0 ? PG_OK : PG_ERR_INVALID;
}

extern "C" int pg_host_n_eigenvecs(const double *ev, int n, double var_explained) {
    if (!ev || n < 1) return PG_ERR_INVALID;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum = sum + ev[i];
    std::vector<double> cum(n);
    for (int i = 0; i < n; ++i) cum[i] = ev[i] / sum;
    int m = n;
    for (int i = 1; i < n; ++i) {
        cum[i] = cum[i - 1] + cum[i];
        if ((cum[i - 1] >= var_explained) & (i - 1 < m)) m = i - 1;
    }
    return m;
}

extern "C" int pg_host_pinv_sym(const double *A, int n, double *out) {
    if (!A || !out || n < 1) return PG_ERR_INVALID;
    return pg_pinv_sym(A, n, out) == 0 ? PG_OK : PG_ERR_INVALID;
}

extern "C" double pg_host_t_two_sided_p(double t_abs, int df) {
    if (df < 1 || std::isnan(t_abs)) return NAN;
    if (std::isinf(t_abs)) return 0.0;
    const std::vector<double> coef = pg_tdist_coef(df);
    const double nu = (double)df;
    const double c2 = nu / (nu + t_abs * t_abs);
    const double s = std::sqrt(1.0 - c2);
    // the device's evaluation order (pg_stats_device.h): four interleaved Horner chains in c2^4
    const double x2 = c2 * c2, y = x2 * x2;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    for (int b = (int)coef.size() - 8; b >= 0; b -= 8) {
        p0 = std::fma(p0, y, coef[b + 4]); p1 = std::fma(p1, y, coef[b + 5]); p2 = std::fma(p2, y, coef[b + 6]); p3 = std::fma(p3, y, coef[b + 7]);
        p0 = std::fma(p0, y, coef[b]); p1 = std::fma(p1, y, coef[b + 1]); p2 = std::fma(p2, y, coef[b + 2]); p3 = std::fma(p3, y, coef[b + 3]);
    }
    const double poly = std::fma(std::fma(std::fma(p3, c2, p2), c2, p1), c2, p0);
    double A;
    if (df & 1) {
        const double c = std::sqrt(c2);
        A = 0.6366197723675814 * (std::atan2(s, c) + s * c * poly);
    } else {
        A = s * poly;
    }
    double p = 1.0 - A;
    p = p < 0.0 ? 0.0 : p;
    return p > 1.0 ? 1.0 : p;
}
