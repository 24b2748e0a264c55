/*
 * poolgen_exact.c -- the EXACT ARBITER of the floating-point rows of the hot path.  TEST INFRASTRUCTURE ONLY
 * (like poolgen_oracle.c: only tests/, smoke() and bench.py's cpu_baseline may load anything under oracle/).
 *
 * Why it exists: two fp64 implementations of an ill-conditioned fit cannot agree to 1e-10 with each other --
 * the reference's literal normal equations (gwas/ols.rs:58-118: X^T X -> LU inverse -> inv X^T y) lose
 * cond(X^T X) * eps digits, and [1 | v_1 ...] with v_1 the leading eigenvector of an UNCENTRED kinship is nearly
 * collinear (v_1 ~ 1/sqrt(n)).  So for the covariate fits (m >= 1) and the gp::ols family the literal oracle
 * is the noisier side and "GPU vs oracle" says nothing below ~1e-6.  Here the same mathematical objects
 * are evaluated in IEEE binary128 (__float128, 113-bit significand, eps = 1e-34) from the SAME fp64 inputs:
 *   - products of two doubles are exact in binary128, sums round at 1e-34: X^T X, X X^T, G G^T are exact to ~1e-31;
 *   - the small solves (Gauss with partial pivoting, Jacobi rotations) then lose at most cond * 1e-34.
 * The results, rounded once to fp64, are the reference POINT both fp64 implementations are measured against:
 * tests assert |GPU - exact| <= 1e-10 and report |oracle - exact| next to it.
 *
 * What is restated (file:line of the reference, /root/reference/src):
 *   exq_ols_covariate : one cell of ols_with_covariate (gwas/ols.rs:345-370) = the fit of gwas/ols.rs:58-160 with
 *                       X = [1 | C | g], last coefficient: b, v_b = ve * inv[last][last], ve = e'e / (n - P),
 *                       t = b / sqrt(v_b), p = 2 (1 - T_{n-1}(|t|)) with the special cases of :142-154.
 *   exq_kinship       : K = G G^T / p (gwas/ols.rs:291-295).
 *   exq_sym_eig       : eigen-decomposition of K (gwas/ols.rs:296) -- eigenvalues descending (the reference's stated intent).
 *   exq_gp_ols        : gp::ols, n < p branch (gp/ols.rs:47-72): b = X^T (X X^T)^-1 y over the training rows; the
 *                       reference's pinv (helpers.rs:463-482) equals the inverse whenever no singular value falls
 *                       under its tolerance, which is what this routine requires (it returns -2 otherwise).
 * The Student-t tail is the closed finite series for integer df (Abramowitz & Stegun 26.7.3 / 26.7.4) in binary128;
 * tests/test_exact_arbiter.py pins it against mpmath's regularised incomplete beta at 50 digits.
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __float128 q_t;
#define EXQ_EPS 2.220446049250313e-16 /* f64::EPSILON, the reference's threshold in gwas/ols.rs:143,147 */

static int exq_threads(int n_threads) {
    int nt = 1;
#ifdef _OPENMP
    nt = n_threads > 0 ? n_threads : (omp_get_max_threads() < 8 ? omp_get_max_threads() : 8);
#endif
    (void)n_threads;
    return nt;
}

/* two-sided Student-t tail probability P(|T_df| > t), t >= 0, integer df >= 1 (A&S 26.7.3 / 26.7.4) */
static q_t exq_t_two_sided(q_t t, int df) {
    const q_t nu = (q_t)df;
    const q_t theta = atanq(t / sqrtq(nu));
    const q_t c = cosq(theta), s = sinq(theta), c2 = c * c;
    q_t A;
    if (df == 1) {
        A = 2.0Q * theta / M_PIq;
    } else if (df & 1) {
        /* A = 2/pi { theta + sin(theta) [ cos + 2/3 cos^3 + ... + (2 4 ... (nu-3)) / (1 3 ... (nu-2)) cos^(nu-2) ] } */
        q_t term = c, sum = c;
        for (int j = 3; j <= df - 2; j += 2) {
            term = term * c2 * (q_t)(j - 1) / (q_t)j;
            sum += term;
        }
        A = 2.0Q / M_PIq * (theta + s * sum);
    } else {
        /* A = sin(theta) { 1 + 1/2 cos^2 + (1 3)/(2 4) cos^4 + ... + (1 3 ... (nu-3)) / (2 4 ... (nu-2)) cos^(nu-2) } */
        q_t term = 1.0Q, sum = 1.0Q;
        for (int j = 2; j <= df - 2; j += 2) {
            term = term * c2 * (q_t)(j - 1) / (q_t)j;
            sum += term;
        }
        A = s * sum;
    }
    return 1.0Q - A;
}

double exq_t_two_sided_p(double t_abs, int df) { return (double)exq_t_two_sided((q_t)t_abs, df); }

/* in-place Gauss-Jordan inverse with partial pivoting; returns -1 on an exactly zero pivot */
static int exq_inverse(q_t *a, int n, q_t *inv) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) inv[i * n + j] = (i == j) ? 1.0Q : 0.0Q;
    for (int c = 0; c < n; c++) {
        int piv = c;
        q_t best = fabsq(a[c * n + c]);
        for (int r = c + 1; r < n; r++)
            if (fabsq(a[r * n + c]) > best) { best = fabsq(a[r * n + c]); piv = r; }
        if (best == 0.0Q) return -1;
        if (piv != c)
            for (int j = 0; j < n; j++) {
                q_t t = a[c * n + j]; a[c * n + j] = a[piv * n + j]; a[piv * n + j] = t;
                t = inv[c * n + j]; inv[c * n + j] = inv[piv * n + j]; inv[piv * n + j] = t;
            }
        const q_t d = 1.0Q / a[c * n + c];
        for (int j = 0; j < n; j++) { a[c * n + j] *= d; inv[c * n + j] *= d; }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const q_t f = a[r * n + c];
            if (f == 0.0Q) continue;
            for (int j = 0; j < n; j++) { a[r * n + j] -= f * a[c * n + j]; inv[r * n + j] -= f * inv[c * n + j]; }
        }
    }
    return 0;
}

/* The cells (locus i, trait j) of ols_with_covariate (gwas/ols.rs:345-370) with given covariates C (n x m row-major,
 * may be NULL when m = 0).  G locus-major p x ld.  Outputs p x k row-major, NaN where X^T X is exactly singular.
 * tstat may be NULL. */
int exq_ols_covariate(const double *G, int64_t p, int n, int64_t ld, const double *Y, int k, const double *C, int m,
                      double *beta, double *var, double *tstat, double *pval, int n_threads) {
    const int P = m + 2, Z = m + 1;
    if (n <= P) return -1;
    const int nt = exq_threads(n_threads);
    /* the locus-independent part of X^T X and X^T y */
    q_t *ztz = (q_t *)calloc((size_t)Z * Z, sizeof(q_t));
    q_t *zty = (q_t *)calloc((size_t)Z * k, sizeof(q_t));
    q_t *yty = (q_t *)calloc((size_t)k, sizeof(q_t));
    for (int i = 0; i < n; i++) {
        for (int a = 0; a < Z; a++) {
            const q_t za = a == 0 ? 1.0Q : (q_t)C[(size_t)i * m + a - 1];
            for (int b = 0; b < Z; b++) {
                const q_t zb = b == 0 ? 1.0Q : (q_t)C[(size_t)i * m + b - 1];
                ztz[a * Z + b] += za * zb;
            }
            for (int j = 0; j < k; j++) zty[a * k + j] += za * (q_t)Y[(size_t)i * k + j];
        }
        for (int j = 0; j < k; j++) yty[j] += (q_t)Y[(size_t)i * k + j] * (q_t)Y[(size_t)i * k + j];
    }
#pragma omp parallel num_threads(nt)
    {
        q_t *xtx = (q_t *)malloc(sizeof(q_t) * P * P);
        q_t *inv = (q_t *)malloc(sizeof(q_t) * P * P);
        q_t *xty = (q_t *)malloc(sizeof(q_t) * P * k);
        q_t *b = (q_t *)malloc(sizeof(q_t) * P);
#pragma omp for schedule(static)
        for (int64_t l = 0; l < p; l++) {
            const double *g = G + l * ld;
            for (int a = 0; a < Z; a++)
                for (int c = 0; c < Z; c++) xtx[a * P + c] = ztz[a * Z + c];
            for (int a = 0; a < Z; a++) { xtx[a * P + Z] = 0.0Q; }
            q_t gg = 0.0Q;
            for (int j = 0; j < k; j++) xty[Z * k + j] = 0.0Q;
            for (int i = 0; i < n; i++) {
                const q_t gi = (q_t)g[i];
                xtx[0 * P + Z] += gi;
                for (int a = 1; a < Z; a++) xtx[a * P + Z] += (q_t)C[(size_t)i * m + a - 1] * gi;
                gg += gi * gi;
                for (int j = 0; j < k; j++) xty[Z * k + j] += gi * (q_t)Y[(size_t)i * k + j];
            }
            for (int a = 0; a < Z; a++) xtx[Z * P + a] = xtx[a * P + Z];
            xtx[Z * P + Z] = gg;
            for (int a = 0; a < Z; a++)
                for (int j = 0; j < k; j++) xty[a * k + j] = zty[a * k + j];
            const int bad = exq_inverse(xtx, P, inv) != 0;
            for (int j = 0; j < k; j++) {
                double bo = NAN, vo = NAN, to = NAN, po = NAN;
                if (!bad) {
                    /* b = inv X^T y; e'e = y'y - b' X^T y would cancel: form the residuals explicitly */
                    for (int a = 0; a < P; a++) {
                        q_t s = 0.0Q;
                        for (int c = 0; c < P; c++) s += inv[a * P + c] * xty[c * k + j];
                        b[a] = s;
                    }
                    q_t ee = 0.0Q;
                    for (int i = 0; i < n; i++) {
                        q_t e = (q_t)Y[(size_t)i * k + j] - b[0];
                        for (int a = 1; a < Z; a++) e -= (q_t)C[(size_t)i * m + a - 1] * b[a];
                        e -= (q_t)g[i] * b[Z];
                        ee += e * e;
                    }
                    const q_t ve = ee / ((q_t)n - (q_t)P);   /* gwas/ols.rs:102-103 */
                    const q_t vb = ve * inv[Z * P + Z];      /* :111-116 */
                    const q_t bl = b[Z];
                    q_t t = (fabsq(bl) <= EXQ_EPS) ? 0.0Q : bl / sqrtq(vb); /* :142-146 */
                    q_t pv;
                    if (fabsq(t) <= EXQ_EPS) pv = 1.0Q;          /* :147-149 */
                    else if (isnanq(t)) pv = 1.0Q;               /* :150-152 */
                    else pv = exq_t_two_sided(fabsq(t), n - 1);  /* :139, :153: df = n - 1 */
                    bo = (double)bl; vo = (double)vb; to = (double)t; po = (double)pv;
                }
                beta[l * k + j] = bo; var[l * k + j] = vo; pval[l * k + j] = po;
                if (tstat) tstat[l * k + j] = to;
            }
        }
        free(xtx); free(inv); free(xty); free(b);
    }
    free(ztz); free(zty); free(yty);
    return 0;
}

/* K = G G^T / p (gwas/ols.rs:291-295), accumulated in binary128, rounded once */
void exq_kinship(const double *G, int64_t p, int n, int64_t ld, double *K, int n_threads) {
    const int nt = exq_threads(n_threads);
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            q_t s = 0.0Q;
            for (int64_t l = 0; l < p; l++) s += (q_t)G[l * ld + i] * (q_t)G[l * ld + j];
            s = s / (q_t)p;
            K[(size_t)i * n + j] = (double)s;
            K[(size_t)j * n + i] = (double)s;
        }
}

/* cyclic Jacobi on a symmetric matrix held in binary128; eigenvalues descending, eigenvectors in the COLUMNS of V.
 * a_q: n x n (destroyed), v_q: n x n */
static void exq_jacobi(q_t *a, int n, q_t *v, q_t *ev) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0Q : 0.0Q;
    for (int sweep = 0; sweep < 60; sweep++) {
        q_t off = 0.0Q, diag = 0.0Q;
        for (int i = 0; i < n; i++) {
            diag += a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
        }
        if (off <= diag * 1e-62Q) break;
        for (int pq = 0; pq < n - 1; pq++)
            for (int qq = pq + 1; qq < n; qq++) {
                const q_t apq = a[pq * n + qq];
                if (apq == 0.0Q) continue;
                const q_t theta = (a[qq * n + qq] - a[pq * n + pq]) / (2.0Q * apq);
                const q_t t = (theta >= 0.0Q ? 1.0Q : -1.0Q) / (fabsq(theta) + sqrtq(theta * theta + 1.0Q));
                const q_t c = 1.0Q / sqrtq(t * t + 1.0Q), s = t * c;
                for (int r = 0; r < n; r++) {
                    const q_t arp = a[r * n + pq], arq = a[r * n + qq];
                    a[r * n + pq] = c * arp - s * arq;
                    a[r * n + qq] = s * arp + c * arq;
                }
                for (int r = 0; r < n; r++) {
                    const q_t apr = a[pq * n + r], aqr = a[qq * n + r];
                    a[pq * n + r] = c * apr - s * aqr;
                    a[qq * n + r] = s * apr + c * aqr;
                }
                for (int r = 0; r < n; r++) {
                    const q_t vrp = v[r * n + pq], vrq = v[r * n + qq];
                    v[r * n + pq] = c * vrp - s * vrq;
                    v[r * n + qq] = s * vrp + c * vrq;
                }
            }
    }
    for (int i = 0; i < n; i++) ev[i] = a[i * n + i];
    /* selection sort, descending, columns of v follow */
    for (int i = 0; i < n - 1; i++) {
        int best = i;
        for (int j = i + 1; j < n; j++)
            if (ev[j] > ev[best]) best = j;
        if (best != i) {
            q_t t = ev[i]; ev[i] = ev[best]; ev[best] = t;
            for (int r = 0; r < n; r++) { t = v[r * n + i]; v[r * n + i] = v[r * n + best]; v[r * n + best] = t; }
        }
    }
}

/* eigenvalues (descending) and eigenvectors (columns of V, n x n row-major) of a symmetric fp64 matrix, in binary128 */
int exq_sym_eig(const double *A, int n, double *evals, double *V) {
    q_t *a = (q_t *)malloc(sizeof(q_t) * n * n), *v = (q_t *)malloc(sizeof(q_t) * n * n), *ev = (q_t *)malloc(sizeof(q_t) * n);
    for (int i = 0; i < n * n; i++) a[i] = (q_t)A[i];
    exq_jacobi(a, n, v, ev);
    for (int i = 0; i < n; i++) evals[i] = (double)ev[i];
    if (V)
        for (int i = 0; i < n * n; i++) V[i] = (double)v[i];
    free(a); free(v); free(ev);
    return 0;
}

/* The kinship preamble of ols_with_covariate in binary128 END TO END (gwas/ols.rs:291-315): K from G, eigen-decomposition
 * of the UNROUNDED K, n_eigenvecs by the literal rule (:297-311) on descending eigenvalues (force_m >= 0 overrides), the m
 * leading eigenvectors rounded once to fp64 -> C_out (n x m row-major, room for n x n).  Returns m.  K_out may be NULL. */
int exq_kinship_covariates(const double *G, int64_t p, int n, int64_t ld, double var_explained, int force_m, double *K_out,
                           double *evals_out, double *C_out, int n_threads) {
    const int nt = exq_threads(n_threads);
    q_t *a = (q_t *)malloc(sizeof(q_t) * n * n), *v = (q_t *)malloc(sizeof(q_t) * n * n), *ev = (q_t *)malloc(sizeof(q_t) * n);
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            q_t s = 0.0Q;
            for (int64_t l = 0; l < p; l++) s += (q_t)G[l * ld + i] * (q_t)G[l * ld + j];
            s = s / (q_t)p;
            a[(size_t)i * n + j] = s;
            a[(size_t)j * n + i] = s;
        }
    if (K_out)
        for (int i = 0; i < n * n; i++) K_out[i] = (double)a[i];
    exq_jacobi(a, n, v, ev);
    int m = force_m;
    if (force_m < 0) {
        /* the rule as written (gwas/ols.rs:297-311), evaluated in binary128: a decision that hinges on the last bits of a
         * cumulative share is outside any tolerance; the tests use thresholds away from the shares */
        q_t sum = 0.0Q;
        for (int i = 0; i < n; i++) sum += ev[i];
        q_t *cum = (q_t *)malloc(sizeof(q_t) * n);
        for (int i = 0; i < n; i++) cum[i] = ev[i] / sum;
        m = n;
        for (int i = 1; i < n; i++) {
            cum[i] = cum[i - 1] + cum[i];
            if ((cum[i - 1] >= (q_t)var_explained) && (i - 1 < m)) m = i - 1;
        }
        free(cum);
    }
    if (evals_out)
        for (int i = 0; i < n; i++) evals_out[i] = (double)ev[i];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m && j < n; j++) C_out[(size_t)i * m + j] = (double)v[i * n + j];
    free(a); free(v); free(ev);
    return m;
}

/* gp::ols, n < p branch (gp/ols.rs:47-72), signature of orc_gp_ols: Xt locus-major P x ld with row 0 = the intercept,
 * beta P x k.  Returns 0, -1 (no intercept column, gp/ols.rs:26-31), -2 (X X^T numerically singular: the reference's pinv
 * would drop a direction, outside this routine), -3 (tall design). */
int exq_gp_ols(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k, const int64_t *row_idx, int n_rows,
               double *beta, int n_threads) {
    double s0 = 0.0;
    for (int i = 0; i < n; i++) s0 = s0 + Xt[i];
    if (s0 < (double)n) return -1;
    if ((int64_t)n >= P) return -3;
    const int nt = exq_threads(n_threads);
    const int r = n_rows;
    q_t *A = (q_t *)malloc(sizeof(q_t) * r * r), *inv = (q_t *)malloc(sizeof(q_t) * r * r);
    q_t *z = (q_t *)malloc(sizeof(q_t) * r * k);
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int a = 0; a < r; a++)
        for (int b = a; b < r; b++) {
            q_t s = 0.0Q;
            const int64_t ia = row_idx[a], ib = row_idx[b];
            for (int64_t c = 0; c < P; c++) s += (q_t)Xt[c * ld + ia] * (q_t)Xt[c * ld + ib];
            A[(size_t)a * r + b] = s;
            A[(size_t)b * r + a] = s;
        }
    q_t dmax = 0.0Q;
    for (int a = 0; a < r; a++)
        if (A[(size_t)a * r + a] > dmax) dmax = A[(size_t)a * r + a];
    /* Cholesky with a relative pivot floor: cond above ~1e20 is "singular" for the purposes of an fp64 comparison */
    int rc = 0;
    q_t *Lm = inv; /* reuse */
    memcpy(Lm, A, sizeof(q_t) * r * r);
    for (int c = 0; c < r && !rc; c++) {
        q_t d = Lm[(size_t)c * r + c];
        for (int t = 0; t < c; t++) d -= Lm[(size_t)c * r + t] * Lm[(size_t)c * r + t];
        if (!(d > dmax * 1e-20Q)) { rc = -2; break; }
        d = sqrtq(d);
        Lm[(size_t)c * r + c] = d;
        for (int rr = c + 1; rr < r; rr++) {
            q_t s = Lm[(size_t)rr * r + c];
            for (int t = 0; t < c; t++) s -= Lm[(size_t)rr * r + t] * Lm[(size_t)c * r + t];
            Lm[(size_t)rr * r + c] = s / d;
        }
    }
    if (!rc) {
        for (int j = 0; j < k; j++) {
            for (int a = 0; a < r; a++) { /* L w = y */
                q_t s = (q_t)Y[(size_t)row_idx[a] * k + j];
                for (int t = 0; t < a; t++) s -= Lm[(size_t)a * r + t] * z[(size_t)t * k + j];
                z[(size_t)a * k + j] = s / Lm[(size_t)a * r + a];
            }
            for (int a = r - 1; a >= 0; a--) { /* L^T z = w */
                q_t s = z[(size_t)a * k + j];
                for (int t = a + 1; t < r; t++) s -= Lm[(size_t)t * r + a] * z[(size_t)t * k + j];
                z[(size_t)a * k + j] = s / Lm[(size_t)a * r + a];
            }
        }
#pragma omp parallel for num_threads(nt) schedule(static)
        for (int64_t c = 0; c < P; c++)
            for (int j = 0; j < k; j++) {
                q_t s = 0.0Q;
                for (int a = 0; a < r; a++) s += (q_t)Xt[c * ld + row_idx[a]] * z[(size_t)a * k + j];
                beta[c * k + j] = (double)s;
            }
    }
    free(A); free(inv); free(z);
    return rc;
}

/* gp::ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199) in binary128, signature of orc_gp_proxy.  The inputs of the
 * arithmetic are what the reference holds in fp64: the centred columns x_c = x - mean (column means over the FIRST n_rows rows,
 * :124-129, formed and subtracted in fp64 as written; the last locus left out, :115).  From there on everything is binary128:
 * X_c X_c^T over the training rows, its leading eigenvector (cyclic Jacobi), and per locus the third coefficient of
 * y ~ [1 | PC1 | x_j] -- the exact solution where the 3 x 3 normal matrix has full rank, the minimum-norm solution
 * (LAPACK gelsd's) where x_j is constant over the training rows: with x_j = c 1 the solutions are
 * (a0 - c t, a1, t) for the fit y ~ a0 + a1 PC1, and the shortest has t = c a0 / (1 + c^2).  b: P x k. */
int exq_gp_proxy(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k, const int64_t *row_idx, int nr,
                 double *b, int n_threads) {
    (void)n;
    const int nt = exq_threads(n_threads);
    const int64_t pc = P - 1;
    double *xc = (double *)malloc(sizeof(double) * (size_t)nr * (pc > 0 ? pc : 1));
    for (int64_t j = 0; j < pc; j++) {
        double mean = 0.0;
        for (int i_ = 0; i_ < nr; i_++) mean += Xt[j * ld + i_];
        mean = mean / (double)nr;
        for (int a = 0; a < nr; a++) xc[(size_t)a * pc + j] = Xt[j * ld + row_idx[a]] - mean;
    }
    q_t *A = (q_t *)malloc(sizeof(q_t) * nr * nr), *V = (q_t *)malloc(sizeof(q_t) * nr * nr), *ev = (q_t *)malloc(sizeof(q_t) * nr);
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int a = 0; a < nr; a++)
        for (int c = a; c < nr; c++) {
            q_t s = 0.0Q;
            for (int64_t j = 0; j < pc; j++) s += (q_t)xc[(size_t)a * pc + j] * (q_t)xc[(size_t)c * pc + j];
            A[(size_t)a * nr + c] = s;
            A[(size_t)c * nr + a] = s;
        }
    exq_jacobi(A, nr, V, ev);
    q_t *e1 = (q_t *)malloc(sizeof(q_t) * nr);
    for (int a = 0; a < nr; a++) e1[a] = V[(size_t)a * nr + 0];
    /* y ~ a0 + a1 PC1 (for the constant-locus branch), and the trait means (:170-172) */
    q_t s11 = 0.0Q, s1e = 0.0Q, see = 0.0Q;
    for (int a = 0; a < nr; a++) { s11 += 1.0Q; s1e += e1[a]; see += e1[a] * e1[a]; }
    for (int j_ = 0; j_ < k; j_++) {
        q_t m = 0.0Q;
        for (int a = 0; a < nr; a++) m += (q_t)Y[row_idx[a] * k + j_];
        b[j_] = (double)(m / (q_t)nr);
    }
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t j = 1; j < P; j++) {
        int constant = 1;
        const double x0 = Xt[j * ld + row_idx[0]];
        for (int a = 1; a < nr; a++)
            if (Xt[j * ld + row_idx[a]] != x0) { constant = 0; break; }
        q_t M[9] = {0}, Mi[9];
        for (int a = 0; a < nr; a++) {
            const q_t xs[3] = {1.0Q, e1[a], (q_t)Xt[j * ld + row_idx[a]]};
            for (int u = 0; u < 3; u++)
                for (int v = 0; v < 3; v++) M[u * 3 + v] += xs[u] * xs[v];
        }
        for (int j_ = 0; j_ < k; j_++) {
            q_t r0 = 0.0Q, r1 = 0.0Q, r2 = 0.0Q;
            for (int a = 0; a < nr; a++) {
                const q_t y = (q_t)Y[row_idx[a] * k + j_];
                r0 += y; r1 += e1[a] * y; r2 += (q_t)Xt[j * ld + row_idx[a]] * y;
            }
            if (constant) {
                const q_t det = s11 * see - s1e * s1e;
                const q_t a0 = (see * r0 - s1e * r1) / det;
                const q_t c = (q_t)x0;
                b[j * k + j_] = (double)(c * a0 / (1.0Q + c * c));
            } else {
                q_t Mc[9];
                memcpy(Mc, M, sizeof Mc);
                if (exq_inverse(Mc, 3, Mi) != 0) { b[j * k + j_] = NAN; continue; }
                b[j * k + j_] = (double)(Mi[6] * r0 + Mi[7] * r1 + Mi[8] * r2);
            }
        }
    }
    free(xc); free(A); free(V); free(ev); free(e1);
    return 0;
}
