/*
 * poolgen_oracle.c -- CPU restatement of the poolgen per-locus regression hot path.
 * TEST INFRASTRUCTURE ONLY (see poolgen_oracle.h).  Plain C, sequential arithmetic in the
 * reference's operation order; compiled with -ffp-contract=off so that no FMA is formed.
 * Citations are file:line relative to /root/reference.
 */
#include "poolgen_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS DBL_EPSILON /* f64::EPSILON */

/* ======================================================================================
 * statrs 0.16.0 special functions (source not under /root/reference; restated from the
 * published crate: src/function/gamma.rs, src/function/beta.rs, src/distribution/
 * {students_t,chi_squared,gamma}.rs).  Pinned by correlation_test.rs:139 and chisq_test.rs:57.
 * ====================================================================================== */
static const double GAMMA_R = 10.900511;
static const double GAMMA_DK[11] = {
    2.48574089138753565546e-5, 1.05142378581721974210,  -3.45687097222016235469,
    4.51227709466894823700,    -2.98285225323576655721, 1.05639711577126713077,
    -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
    4.63399473359905636708e-6, -2.71994908488607703910e-9};
static const double LN_PI = 1.1447298858494001741434273513530587116472948129153;
static const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;

/* statrs::function::gamma::ln_gamma (Lanczos, g = 10.900511, n = 11) */
double orc_ln_gamma(double x) {
    if (x < 0.5) {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; i++) s += GAMMA_DK[i] / ((double)i - x);
        return LN_PI - log(sin(M_PI * x)) - log(s) - LN_2_SQRT_E_OVER_PI -
               (0.5 - x) * log((0.5 - x + GAMMA_R) / M_E);
    } else {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; i++) s += GAMMA_DK[i] / (x + (double)i - 1.0);
        return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + GAMMA_R) / M_E);
    }
}

/* approx ulps_eq!(x, 1.0) of the `approx` crate defaults (epsilon = f64::EPSILON, max_ulps 4) */
static int ulps_eq_one(double x) {
    if (fabs(x - 1.0) <= ORC_EPS) return 1;
    int64_t a, b;
    double one = 1.0;
    memcpy(&a, &x, 8);
    memcpy(&b, &one, 8);
    if ((a < 0) != (b < 0)) return 0;
    int64_t d = a > b ? a - b : b - a;
    return d <= 4;
}

/* statrs::function::beta::checked_beta_reg (modified Lentz continued fraction, <=140 iterations) */
double orc_beta_reg(double a, double b, double x) {
    if (!(a > 0.0) || !(b > 0.0) || !(x >= 0.0 && x <= 1.0)) return NAN;
    double bt;
    if (x == 0.0 || ulps_eq_one(x)) {
        bt = 0.0;
    } else {
        bt = exp(orc_ln_gamma(a + b) - orc_ln_gamma(a) - orc_ln_gamma(b) + a * log(x) +
                 b * log(1.0 - x));
    }
    int symm = x >= (a + 1.0) / (a + b + 2.0);
    const double eps = 1.1102230246251565e-16; /* prec::F64_PREC */
    const double fpmin = DBL_MIN / eps;
    if (symm) {
        double swap = a;
        x = 1.0 - x;
        a = b;
        b = swap;
    }
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0;
    double d = 1.0 - qab * x / qap;
    if (fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int mi = 1; mi < 141; mi++) {
        double m = (double)mi;
        double m2 = m * 2.0;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        h = h * d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) <= eps) break;
    }
    return symm ? 1.0 - bt * h / a : bt * h / a;
}

/* statrs StudentsT{0,1,nu}.cdf (students_t.rs): h = nu/(nu+x^2), ib = 0.5*I_h(nu/2, 1/2) */
double orc_students_t_cdf(double x, double freedom) {
    if (isinf(freedom)) return 0.5 * erfc(-x / M_SQRT2);
    double k = x;
    double h = freedom / (freedom + k * k);
    double ib = 0.5 * orc_beta_reg(freedom / 2.0, 0.5, h);
    return x <= 0.0 ? ib : 1.0 - ib;
}

static int almost_eq(double a, double b, double acc) { return fabs(a - b) < acc; }

/* statrs::function::gamma::checked_gamma_lr (Cephes igam/igamc port) */
double orc_gamma_lr(double a, double x) {
    if (isnan(a) || isnan(x)) return NAN;
    if (a <= 0.0 || isinf(a)) return NAN;
    if (x <= 0.0 || isinf(x)) return NAN;
    const double eps = 0.000000000000001;
    const double big = 4503599627370496.0;
    const double big_inv = 2.22044604925031308085e-16;
    if (almost_eq(a, 0.0, 1e-15)) return 1.0;
    if (almost_eq(x, 0.0, 1e-15)) return 0.0;
    double ax = a * log(x) - x - orc_ln_gamma(a);
    if (ax < -709.78271289338399) return a < x ? 1.0 : 0.0;
    if (x <= 1.0 || x <= a) {
        double r2 = a, c2 = 1.0, ans2 = 1.0;
        for (;;) {
            r2 += 1.0;
            c2 *= x / r2;
            ans2 += c2;
            if (c2 / ans2 <= eps) break;
        }
        return exp(ax) * ans2 / a;
    }
    double y = 1.0 - a;
    double z = x + y + 1.0;
    int c = 0;
    double p3 = 1.0, q3 = x, p2 = x + 1.0, q2 = z * x;
    double ans = p2 / q2;
    for (;;) {
        y += 1.0;
        z += 2.0;
        c += 1;
        double yc = y * (double)c;
        double p = p2 * z - p3 * yc;
        double q = q2 * z - q3 * yc;
        p3 = p2; p2 = p; q3 = q2; q2 = q;
        if (fabs(p) > big) { p3 *= big_inv; p2 *= big_inv; q3 *= big_inv; q2 *= big_inv; }
        if (q != 0.0) {
            double nextans = p / q;
            double error = fabs((ans - nextans) / nextans);
            ans = nextans;
            if (error <= eps) break;
        }
    }
    return 1.0 - exp(ax) * ans;
}

/* statrs ChiSquared{df}.cdf = Gamma{shape df/2, rate 1/2}.cdf(x) = gamma_lr(df/2, x/2) */
double orc_chisq_cdf(double x, double freedom) {
    if (x <= 0.0) return 0.0;
    if (isinf(x)) return 1.0;
    return orc_gamma_lr(freedom / 2.0, x * 0.5);
}

/* ======================================================================================
 * helpers.rs
 * ====================================================================================== */
/* sensible_round (helpers.rs:103-108): (x * 10^d).round() / 10^d, round = half away from zero */
double orc_sensible_round(double x, int n_digits) {
    char tmp[16];
    snprintf(tmp, sizeof tmp, "1e%d", n_digits);
    double factor = strtod(tmp, NULL);
    return round(x * factor) / factor;
}

/* Rust `impl Display for f64`: shortest round-trip digits, positional (never exponent) */
int orc_fmt_display(double x, char *buf, int cap) {
    if (isnan(x)) return snprintf(buf, cap, "NaN");
    if (isinf(x)) return snprintf(buf, cap, x > 0 ? "inf" : "-inf");
    if (x == 0.0) return snprintf(buf, cap, signbit(x) ? "-0" : "0");
    char e[40];
    int prec;
    for (prec = 1; prec <= 17; prec++) {
        snprintf(e, sizeof e, "%.*e", prec - 1, x);
        if (strtod(e, NULL) == x) break;
    }
    /* e = [-]d[.ddd]e[+-]XX */
    char digits[24];
    int nd = 0, neg = 0;
    const char *s = e;
    if (*s == '-') { neg = 1; s++; }
    while (*s && *s != 'e') {
        if (*s != '.') digits[nd++] = *s;
        s++;
    }
    int exp10 = atoi(s + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    char out[400];
    int o = 0;
    if (neg) out[o++] = '-';
    if (exp10 >= 0) {
        for (int i = 0; i <= exp10; i++) out[o++] = i < nd ? digits[i] : '0';
        if (nd > exp10 + 1) {
            out[o++] = '.';
            for (int i = exp10 + 1; i < nd; i++) out[o++] = digits[i];
        }
    } else {
        out[o++] = '0';
        out[o++] = '.';
        for (int i = 0; i < -exp10 - 1; i++) out[o++] = '0';
        for (int i = 0; i < nd; i++) out[o++] = digits[i];
    }
    out[o] = 0;
    return snprintf(buf, cap, "%s", out);
}

/* parse_f64_roundup_and_own (helpers.rs:111-117) */
int orc_parse_f64_roundup_and_own(double x, int n_digits, char *buf, int cap) {
    char s[400];
    int len = orc_fmt_display(x, s, sizeof s);
    if (len < n_digits) return snprintf(buf, cap, "%s", s);
    return orc_fmt_display(orc_sensible_round(x, n_digits), buf, cap);
}

/* ndarray 0.15 `sum()` on a contiguous slice: numeric_util::unrolled_fold (8 lanes) */
double orc_ndarray_sum(const double *x, int64_t len) {
    double acc = 0.0, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    while (len - i >= 8) {
        for (int j = 0; j < 8; j++) p[j] = p[j] + x[i + j];
        i += 8;
    }
    acc = acc + (p[0] + p[4]);
    acc = acc + (p[1] + p[5]);
    acc = acc + (p[2] + p[6]);
    acc = acc + (p[3] + p[7]);
    for (; i < len; i++) acc = acc + x[i];
    return acc;
}

/* ndarray 0.15 1-D dot: numeric_util::unrolled_dot */
static double unrolled_dot(const double *x, const double *y, int64_t len) {
    double sum = 0.0, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    while (len - i >= 8) {
        for (int j = 0; j < 8; j++) p[j] = p[j] + x[i + j] * y[i + j];
        i += 8;
    }
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (; i < len; i++) sum = sum + x[i] * y[i];
    return sum;
}

/* mean_array1_ignore_nan (helpers.rs:258-264) */
double orc_mean_ignore_nan(const double *x, int64_t len, int64_t stride) {
    double sum = 0.0, cnt = 0.0;
    for (int64_t i = 0; i < len; i++) {
        double a = x[i * stride];
        if (!isnan(a)) { sum = sum + a; cnt += 1.0; }
    }
    return sum / cnt;
}

/* ======================================================================================
 * dense linear algebra standing in for ndarray-linalg 0.16 / LAPACK
 * ====================================================================================== */
/* LU with partial pivoting (dgetf2 order: first max |a| in the column).  Returns 0, or k+1 for
 * an exactly zero pivot at step k (LAPACK info>0, which makes `.inv()` return Err). */
static int lu_factor(double *a, int n, int *piv, int *sign) {
    *sign = 1;
    int info = 0;
    for (int k = 0; k < n; k++) {
        int pi = k;
        double pm = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; i++) {
            double v = fabs(a[i * n + k]);
            if (v > pm) { pm = v; pi = i; }
        }
        piv[k] = pi;
        if (a[pi * n + k] != 0.0) {
            if (pi != k) {
                for (int j = 0; j < n; j++) {
                    double t = a[k * n + j]; a[k * n + j] = a[pi * n + j]; a[pi * n + j] = t;
                }
                *sign = -*sign;
            }
            double inv = 1.0 / a[k * n + k];
            for (int i = k + 1; i < n; i++) a[i * n + k] *= inv;
        } else if (info == 0) {
            info = k + 1;
        }
        for (int i = k + 1; i < n; i++) {
            double l = a[i * n + k];
            if (l != 0.0)
                for (int j = k + 1; j < n; j++) a[i * n + j] -= l * a[k * n + j];
        }
    }
    return info;
}

/* `.inv()` (dgetrf + dgetri): Err when a pivot is exactly zero (gwas/ols.rs:68-71, 77-80) */
int orc_lu_inverse(const double *a, int n, double *inv) {
    double *lu = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    int sign;
    memcpy(lu, a, sizeof(double) * n * n);
    int info = lu_factor(lu, n, piv, &sign);
    if (info != 0) { free(lu); free(piv); return -1; }
    /* solve A X = I column by column: P A = L U */
    double *col = (double *)malloc(sizeof(double) * n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) col[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; k++) {
            if (piv[k] != k) { double t = col[k]; col[k] = col[piv[k]]; col[piv[k]] = t; }
        }
        for (int i = 0; i < n; i++) {
            double s = col[i];
            for (int j = 0; j < i; j++) s -= lu[i * n + j] * col[j];
            col[i] = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = col[i];
            for (int j = i + 1; j < n; j++) s -= lu[i * n + j] * col[j];
            col[i] = s / lu[i * n + i];
        }
        for (int i = 0; i < n; i++) inv[i * n + c] = col[i];
    }
    free(col); free(lu); free(piv);
    return 0;
}

/* `.det()`: product of U's diagonal times the permutation sign; singular factorisation -> 0 */
double orc_lu_det(const double *a, int n) {
    double *lu = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    int sign;
    memcpy(lu, a, sizeof(double) * n * n);
    int info = lu_factor(lu, n, piv, &sign);
    double d = 0.0;
    if (info == 0) {
        d = (double)sign;
        for (int i = 0; i < n; i++) d *= lu[i * n + i];
    }
    free(lu); free(piv);
    return d;
}

/* Cyclic Jacobi for a symmetric matrix; eigenvalues DESCENDING, eigenvectors = columns of v.
 * The reference calls the general `.eig()` (dgeev) and assumes descending order
 * (gwas/ols.rs:296 comment); LAPACK's actual order is build-dependent, so the oracle (and the
 * product) implement the documented intent. */
int orc_sym_eig(const double *a_in, int n, double *evals, double *v) {
    double *a = (double *)malloc(sizeof(double) * n * n);
    memcpy(a, a_in, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
        }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++) {
            for (int q = p + 1; q < n; q++) {
                double apq = a[p * n + q];
                if (apq == 0.0) continue;
                double app = a[p * n + p], aqq = a[q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    /* sort descending (selection sort, swapping columns) */
    for (int i = 0; i < n; i++) evals[i] = a[i * n + i];
    for (int i = 0; i < n - 1; i++) {
        int mx = i;
        for (int j = i + 1; j < n; j++)
            if (evals[j] > evals[mx]) mx = j;
        if (mx != i) {
            double t = evals[i]; evals[i] = evals[mx]; evals[mx] = t;
            for (int k = 0; k < n; k++) {
                double u = v[k * n + i]; v[k * n + i] = v[k * n + mx]; v[k * n + mx] = u;
            }
        }
    }
    free(a);
    return 0;
}

/* pinv (helpers.rs:463-482) for a symmetric PSD-ish input (X X^T or X^T X): for a symmetric
 * matrix the SVD is U = V*sign, s = |lambda|; tolerance = eps * len(s) * max(s); singular values
 * <= tolerance are zeroed. */
int orc_pinv_sym(const double *a, int n, double *out) {
    double *ev = (double *)malloc(sizeof(double) * n);
    double *v = (double *)malloc(sizeof(double) * n * n);
    orc_sym_eig(a, n, ev, v);
    double smax = 0.0;
    for (int i = 0; i < n; i++)
        if (fabs(ev[i]) > smax) smax = fabs(ev[i]);
    double tol = ORC_EPS * (double)n * smax;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) out[i * n + j] = 0.0;
    for (int e = 0; e < n; e++) {
        if (fabs(ev[e]) > tol) {
            double w = 1.0 / ev[e];
            for (int i = 0; i < n; i++) {
                double vi = v[i * n + e] * w;
                for (int j = 0; j < n; j++) out[i * n + j] += vi * v[j * n + e];
            }
        }
    }
    free(ev); free(v);
    return 0;
}

/* ======================================================================================
 * base/sync.rs
 * ====================================================================================== */
static const char ALLELES[6] = {'A', 'T', 'C', 'G', 'N', 'D'}; /* sync.rs:134 reader order */

/* String::lparse -> LocusCounts (sync.rs:100-156) */
int orc_parse_sync_line(const char *line_in, char *chrom, int chrom_cap, uint64_t *pos,
                        uint64_t *counts, int max_pools) {
    size_t len = strlen(line_in);
    char *line = (char *)malloc(len + 1);
    memcpy(line, line_in, len + 1);
    if (len && line[len - 1] == '\n') { line[--len] = 0; if (len && line[len - 1] == '\r') line[--len] = 0; }
    if (len == 0) { free(line); return -2; }
    if (line[0] == '#') { free(line); return 0; }
    int field = 0, n = 0, rc = 0;
    char *p = line;
    while (p) {
        char *tab = strchr(p, '\t');
        if (tab) *tab = 0;
        if (field == 0) {
            snprintf(chrom, chrom_cap, "%s", p);
        } else if (field == 1) {
            char *end;
            if (*p == 0 || *p == '-' ) { rc = -3; break; }
            *pos = strtoull(p, &end, 10);
            if (*end != 0) { rc = -3; break; }
        } else if (field >= 3) {
            if (n >= max_pools) { rc = -4; break; }
            char *q = p;
            int j = 0;
            while (q && j < 6) {
                char *colon = strchr(q, ':');
                if (colon) *colon = 0;
                char *end;
                if (*q == 0) { rc = -5; break; }
                uint64_t v = strtoull(q, &end, 10);
                if (*end != 0) { rc = -5; break; }
                counts[n * 6 + j] = v;
                j++;
                q = colon ? colon + 1 : NULL;
            }
            if (rc) break;
            if (j < 6) { rc = -5; break; }
            n++;
        }
        field++;
        p = tab ? tab + 1 : NULL;
    }
    free(line);
    if (rc) return rc;
    return n;
}

/* LocusCounts::to_frequencies (sync.rs:166-192) */
void orc_to_frequencies(const uint64_t *counts, int n, int a, double *freq) {
    for (int i = 0; i < n; i++) {
        double rs = 0.0;
        for (int j = 0; j < a; j++) rs = rs + (double)counts[i * a + j];
        for (int j = 0; j < a; j++)
            freq[i * a + j] = (rs == 0.0) ? NAN : (double)counts[i * a + j] / rs;
    }
}

/* LocusCounts::filter (sync.rs:195-303) */
int orc_filter_locus(const uint64_t *counts, int n, const double *pool_sizes, const orc_filter *f,
                     int *allele_ids, uint64_t *out_counts) {
    int ids[6], a = 0;
    for (int j = 0; j < 6; j++)
        if (!(f->remove_ns && ALLELES[j] == 'N')) ids[a++] = j; /* :200-213 */
    /* minimum coverage (:217-229) */
    double min_cov = 0.0;
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < a; j++) s = s + (double)counts[i * 6 + ids[j]];
        if (i == 0 || s < min_cov) min_cov = s;
    }
    if (min_cov < (double)f->min_coverage_depth) return 0;
    /* frequencies over the remaining columns (:242-250) */
    double *fr = (double *)malloc(sizeof(double) * n * 6);
    for (int i = 0; i < n; i++) {
        double rs = 0.0;
        for (int j = 0; j < a; j++) rs = rs + (double)counts[i * 6 + ids[j]];
        for (int j = 0; j < a; j++)
            fr[i * 6 + j] = (rs == 0.0) ? NAN : (double)counts[i * 6 + ids[j]] / rs;
    }
    /* pool-size weighted allele frequency (:258-282); the weight is recomputed per term as
     * pool_sizes[i] / sum(pool_sizes) exactly as written */
    double total = 0.0;
    for (int i = 0; i < n; i++) total = total + pool_sizes[i];
    int keep[6], nk = 0;
    for (int j = 0; j < a; j++) {
        double q = 0.0;
        for (int i = 0; i < n; i++) {
            double v = fr[i * 6 + j];
            q += isnan(v) ? 0.0 : v * (pool_sizes[i] / total);
        }
        if ((q < f->min_allele_frequency) | (q > (1.00 - f->min_allele_frequency))) continue;
        keep[nk++] = j;
    }
    if (nk < 2) { free(fr); return 0; } /* :284-286 */
    int n_missing = 0;                  /* :288-299, first remaining allele */
    for (int i = 0; i < n; i++)
        if (isnan(fr[i * 6 + keep[0]])) n_missing++;
    free(fr);
    if (n_missing == n) return 0;
    if (((double)n_missing / (double)n) > f->max_missingness_rate) return 0;
    for (int j = 0; j < nk; j++) allele_ids[j] = ids[keep[j]];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nk; j++) out_counts[i * nk + j] = counts[i * 6 + ids[keep[j]]];
    return nk;
}

/* LocusFrequencies::sort_by_allele_freq (sync.rs:477-506): stable sort of the columns on the
 * NaN-ignoring column sums */
void orc_sort_by_allele_freq(double *freq, int n, int a, int *allele_ids, int decreasing) {
    double cs[6];
    int idx[6];
    for (int j = 0; j < a; j++) {
        double s = 0.0;
        for (int i = 0; i < n; i++)
            if (!isnan(freq[i * a + j])) s = s + freq[i * a + j];
        cs[j] = s;
        idx[j] = j;
    }
    for (int i = 1; i < a; i++) { /* stable insertion sort */
        int t = idx[i], j = i - 1;
        while (j >= 0 && (decreasing ? cs[idx[j]] < cs[t] : cs[idx[j]] > cs[t])) {
            idx[j + 1] = idx[j];
            j--;
        }
        idx[j + 1] = t;
    }
    double *tmp = (double *)malloc(sizeof(double) * n * a);
    int ids2[6];
    for (int j = 0; j < a; j++) {
        ids2[j] = allele_ids[idx[j]];
        for (int i = 0; i < n; i++) tmp[i * a + j] = freq[i * a + idx[j]];
    }
    memcpy(freq, tmp, sizeof(double) * n * a);
    memcpy(allele_ids, ids2, sizeof(int) * a);
    free(tmp);
}

/* ======================================================================================
 * gwas/ols.rs
 * ====================================================================================== */
/* estimate_effects / estimate_variances / estimate_significance (ols.rs:58-160) */
int orc_ols_fit(const double *X, const double *y, int n, int P, double *b, double *v_b, double *t,
                double *pval) {
    int rc = 0;
    double *e = (double *)malloc(sizeof(double) * n);
    if (n < P) {
        /* ols.rs:67-75: inv(X X^T); b = (X^T inv) y */
        double *xxt = (double *)malloc(sizeof(double) * n * n);
        double *inv = (double *)malloc(sizeof(double) * n * n);
        double *m = (double *)malloc(sizeof(double) * P * n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double s = 0.0;
                for (int c = 0; c < P; c++) s = s + X[i * P + c] * X[j * P + c];
                xxt[i * n + j] = s;
            }
        if (orc_lu_inverse(xxt, n, inv) != 0 || orc_lu_det(inv, n) == 0.0) {
            rc = -1;
        } else {
            for (int c = 0; c < P; c++)
                for (int j = 0; j < n; j++) {
                    double s = 0.0;
                    for (int i = 0; i < n; i++) s = s + X[i * P + c] * inv[i * n + j];
                    m[c * n + j] = s;
                }
            for (int c = 0; c < P; c++) b[c] = unrolled_dot(&m[c * n], y, n);
            for (int i = 0; i < n; i++) e[i] = y[i] - unrolled_dot(&X[i * P], b, P);
            double ve = unrolled_dot(e, e, n) / ((double)n - (double)P); /* ols.rs:103 */
            /* vcv = ve * X^T inv inv X (ols.rs:105-109); only the diagonal is used */
            double *m2 = (double *)malloc(sizeof(double) * P * n);
            for (int c = 0; c < P; c++)
                for (int j = 0; j < n; j++) {
                    double s = 0.0;
                    for (int i = 0; i < n; i++) s = s + m[c * n + i] * inv[i * n + j];
                    m2[c * n + j] = s;
                }
            for (int c = 0; c < P; c++) {
                double s = 0.0;
                for (int i = 0; i < n; i++) s = s + m2[c * n + i] * X[i * P + c];
                v_b[c] = ve * s;
            }
            free(m2);
        }
        free(xxt); free(inv); free(m);
    } else {
        /* ols.rs:77-84: inv(X^T X); b = (inv X^T) y */
        double *xtx = (double *)malloc(sizeof(double) * P * P);
        double *inv = (double *)malloc(sizeof(double) * P * P);
        double *m = (double *)malloc(sizeof(double) * P * n);
        for (int r = 0; r < P; r++)
            for (int c = 0; c < P; c++) {
                double s = 0.0;
                for (int i = 0; i < n; i++) s = s + X[i * P + r] * X[i * P + c];
                xtx[r * P + c] = s;
            }
        if (orc_lu_inverse(xtx, P, inv) != 0 || orc_lu_det(inv, P) == 0.0) {
            rc = -1;
        } else {
            for (int r = 0; r < P; r++)
                for (int i = 0; i < n; i++) {
                    double s = 0.0;
                    for (int c = 0; c < P; c++) s = s + inv[r * P + c] * X[i * P + c];
                    m[r * n + i] = s;
                }
            for (int r = 0; r < P; r++) b[r] = unrolled_dot(&m[r * n], y, n);
            for (int i = 0; i < n; i++) e[i] = y[i] - unrolled_dot(&X[i * P], b, P);
            double ve = unrolled_dot(e, e, n) / ((double)n - (double)P); /* ols.rs:102-103 */
            for (int r = 0; r < P; r++) v_b[r] = ve * inv[r * P + r];      /* ols.rs:111-116 */
        }
        free(xtx); free(inv); free(m);
    }
    free(e);
    if (rc) return rc;
    /* ols.rs:139-158: df = n - 1 */
    for (int i = 0; i < P; i++) {
        t[i] = (fabs(b[i]) <= ORC_EPS) ? 0.0 : b[i] / sqrt(v_b[i]);
        if (fabs(t[i]) <= ORC_EPS) pval[i] = 1.0;
        else if (isnan(t[i])) pval[i] = 1.0;
        else pval[i] = 2.00 * (1.00 - orc_students_t_cdf(fabs(t[i]), (double)n - 1.0));
    }
    return 0;
}

/* remove_missing (sync.rs:508-549): indices of pools whose phenotype row mean is not NaN */
static int pools_with_phenotypes(const double *Y, int n, int k, int *idx) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < k; j++) s = s + Y[i * k + j];
        if (!isnan(s / (double)k)) idx[m++] = i;
    }
    return m;
}

/* ols_iterate (ols.rs:201-276), numeric part */
int orc_ols_iterate_locus(const uint64_t *counts, int n, const double *Y, int k,
                          const double *pool_sizes, const orc_filter *f, orc_locus_hdr *hdr,
                          double *beta, double *pval) {
    hdr->n_alleles = 0;
    int *idx = (int *)malloc(sizeof(int) * n);
    int n2 = pools_with_phenotypes(Y, n, k, idx);
    if (n2 != n) {
        /* quirk: the reference shrinks the counts matrix but not FilterStats.pool_sizes, so the
         * assert at sync.rs:254-257 panics the worker; the restatement reports it as "dropped". */
        free(idx);
        return -1;
    }
    free(idx);
    int ids[6];
    uint64_t *fc = (uint64_t *)malloc(sizeof(uint64_t) * n * 6);
    int a = orc_filter_locus(counts, n, pool_sizes, f, ids, fc);
    if (a == 0) { free(fc); return 0; }
    double *fr = (double *)malloc(sizeof(double) * n * a);
    orc_to_frequencies(fc, n, a, fr);
    orc_sort_by_allele_freq(fr, n, a, ids, 1); /* ols.rs:222 */
    int P = a;                                 /* intercept + (a-1) alleles, ols.rs:227-246 */
    double *X = (double *)malloc(sizeof(double) * n * P);
    for (int i = 0; i < n; i++) {
        X[i * P] = 1.0;
        for (int j = 1; j < P; j++) X[i * P + j] = fr[i * a + j];
    }
    double *b = (double *)malloc(sizeof(double) * P * 4);
    double *y = (double *)malloc(sizeof(double) * n);
    int ok = 1;
    for (int j = 0; j < k && ok; j++) { /* ols(), ols.rs:163-199 */
        for (int i = 0; i < n; i++) y[i] = Y[i * k + j];
        if (orc_ols_fit(X, y, n, P, b, b + P, b + 2 * P, b + 3 * P) != 0) { ok = 0; break; }
        for (int i = 1; i < P; i++) {
            beta[(i - 1) * k + j] = b[i];
            pval[(i - 1) * k + j] = b[3 * P + i];
        }
    }
    if (ok) {
        hdr->n_alleles = P - 1;
        for (int i = 1; i < P; i++) {
            hdr->allele_ids[i - 1] = ids[i];
            double s = 0.0; /* x_matrix.column(i).mean() (ols.rs:266): sequential sum / n */
            for (int r = 0; r < n; r++) s = s + X[r * P + i];
            hdr->mean_freq[i - 1] = s / (double)n;
        }
    }
    free(fc); free(fr); free(X); free(b); free(y);
    return ok ? P - 1 : 0;
}

/* CSV fragment of ols_iterate (ols.rs:255-275) */
int orc_ols_iterate_csv(const char *chrom, uint64_t pos, const uint64_t *counts, int n,
                        const double *Y, int k, const double *pool_sizes, const orc_filter *f,
                        char *out, int cap) {
    orc_locus_hdr h;
    double *beta = (double *)malloc(sizeof(double) * 5 * k * 2);
    double *pv = beta + 5 * k;
    int na = orc_ols_iterate_locus(counts, n, Y, k, pool_sizes, f, &h, beta, pv);
    int o = 0;
    if (na > 0) {
        char s1[400], s2[400], s3[400];
        for (int i = 0; i < na; i++)
            for (int j = 0; j < k; j++) {
                orc_parse_f64_roundup_and_own(h.mean_freq[i], 8, s1, sizeof s1);
                orc_parse_f64_roundup_and_own(beta[i * k + j], 6, s2, sizeof s2);
                orc_parse_f64_roundup_and_own(pv[i * k + j], 12, s3, sizeof s3);
                o += snprintf(out + o, cap - o, "%s,%llu,%c,%s,Pheno_%d,%s,%s\n", chrom,
                              (unsigned long long)pos, ALLELES[h.allele_ids[i]], s1, j, s2, s3);
            }
    }
    free(beta);
    return na > 0 ? o : 0;
}

/* n_eigenvecs rule (ols.rs:297-311), literal, on eigenvalues in the given order */
int orc_n_eigenvecs_rule(const double *ev, int n, double threshold) {
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum = sum + ev[i];
    double *cum = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) cum[i] = ev[i] / sum;
    int m = n;
    for (int i = 1; i < n; i++) {
        cum[i] = cum[i - 1] + cum[i];
        if ((cum[i - 1] >= threshold) & (i - 1 < m)) m = i - 1;
    }
    free(cum);
    return m;
}

/* kinship = G G^T / p (ols.rs:291-295).  G locus-major (p x n, ld).
 * The reference hands this product to BLAS (ndarray `dot` -> MKL dgemm), so the restatement should not be a scalar loop when it
 * serves as the CPU baseline: loci are taken in blocks of 8 (one rank-8 update of the upper triangle per block: every K row is
 * loaded and stored once per 8 loci instead of once per locus), the inner loop over j is vectorised (omp simd; AVX2 / AVX-512
 * clones chosen at load time), threads own disjoint locus ranges and their partial sums are added in thread order.  Each entry is
 * still a plain sum of products over the loci -- only the association order differs from a sequential loop (1e-16 relative). */
#if defined(__GNUC__) && defined(__x86_64__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void kinship_block8(const double *G, int64_t l0, int64_t l1, int n, int64_t ld, double *kp) {
    int64_t l = l0;
    for (; l + 8 <= l1; l += 8) {
        const double *g0 = G + l * ld, *g1 = g0 + ld, *g2 = g1 + ld, *g3 = g2 + ld, *g4 = g3 + ld, *g5 = g4 + ld, *g6 = g5 + ld, *g7 = g6 + ld;
        for (int i = 0; i < n; i++) {
            const double a0 = g0[i], a1 = g1[i], a2 = g2[i], a3 = g3[i], a4 = g4[i], a5 = g5[i], a6 = g6[i], a7 = g7[i];
            double *row = kp + (size_t)i * n;
#pragma omp simd
            for (int j = i; j < n; j++)
                row[j] += ((a0 * g0[j] + a1 * g1[j]) + (a2 * g2[j] + a3 * g3[j])) + ((a4 * g4[j] + a5 * g5[j]) + (a6 * g6[j] + a7 * g7[j]));
        }
    }
    for (; l < l1; l++) {
        const double *g = G + l * ld;
        for (int i = 0; i < n; i++) {
            const double gi = g[i];
            double *row = kp + (size_t)i * n;
#pragma omp simd
            for (int j = i; j < n; j++) row[j] += gi * g[j];
        }
    }
}

void orc_kinship(const double *G, int64_t p, int n, int64_t ld, double *K, int n_threads) {
    int nt = 1;
#ifdef _OPENMP
    nt = n_threads > 0 ? n_threads : (omp_get_max_threads() < 8 ? omp_get_max_threads() : 8); /* a test oracle: never the whole box */
#endif
    (void)n_threads;
    double *part = (double *)calloc((size_t)nt * n * n, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        int tid = 0, nth = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nth = omp_get_num_threads();
#endif
        const int64_t l0 = p * tid / nth, l1 = p * (tid + 1) / nth;
        kinship_block8(G, l0, l1, n, ld, part + (size_t)tid * n * n);
    }
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            double s = 0.0;
            for (int t = 0; t < nt; t++) s += part[(size_t)t * n * n + (size_t)i * n + j];
            K[(size_t)i * n + j] = s / (double)p;
            K[(size_t)j * n + i] = s / (double)p;
        }
    free(part);
}

/* ols_with_covariate numeric core (ols.rs:291-370) */
int orc_ols_with_covariate(const double *G, int64_t p, int n, int64_t ld, const double *Y, int k,
                           double var_explained, int force_m, const double *covariate_in,
                           double *K_out, double *evals_out, double *cov_out, double *beta,
                           double *var, double *pval, int n_threads) {
    int m;
    double *C = NULL;
    if (covariate_in && force_m >= 0) {
        m = force_m;
        C = (double *)malloc(sizeof(double) * n * (m > 0 ? m : 1));
        memcpy(C, covariate_in, sizeof(double) * n * m);
    } else {
        double *K = (double *)malloc(sizeof(double) * n * n);
        double *ev = (double *)malloc(sizeof(double) * n);
        double *V = (double *)malloc(sizeof(double) * n * n);
        orc_kinship(G, p, n, ld, K, n_threads);
        orc_sym_eig(K, n, ev, V);
        m = force_m >= 0 ? force_m : orc_n_eigenvecs_rule(ev, n, var_explained);
        C = (double *)malloc(sizeof(double) * n * (m > 0 ? m : 1));
        for (int i = 0; i < n; i++)
            for (int j = 0; j < m; j++) C[i * m + j] = V[i * n + j]; /* ols.rs:312-315 */
        if (K_out) memcpy(K_out, K, sizeof(double) * n * n);
        if (evals_out) memcpy(evals_out, ev, sizeof(double) * n);
        free(K); free(ev); free(V);
    }
    if (cov_out) memcpy(cov_out, C, sizeof(double) * n * m);
    int P = m + 2;
    int nt = 1;
#ifdef _OPENMP
    nt = n_threads > 0 ? n_threads : (omp_get_max_threads() < 8 ? omp_get_max_threads() : 8); /* a test oracle: never the whole box */
#endif
#pragma omp parallel num_threads(nt)
    {
        double *X = (double *)malloc(sizeof(double) * n * P);
        double *y = (double *)malloc(sizeof(double) * n);
        double *r = (double *)malloc(sizeof(double) * P * 4);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < p; i++) {
            for (int j = 0; j < k; j++) { /* one cell (i, j), ols.rs:345-370 */
                for (int i_ = 0; i_ < n; i_++) {
                    X[i_ * P] = 1.0;
                    for (int j_ = 1; j_ < m + 1; j_++) X[i_ * P + j_] = C[i_ * m + j_ - 1];
                    X[i_ * P + m + 1] = G[i * ld + i_];
                    y[i_] = Y[i_ * k + j];
                }
                if (orc_ols_fit(X, y, n, P, r, r + P, r + 2 * P, r + 3 * P) == 0) {
                    beta[i * k + j] = r[m + 1];
                    var[i * k + j] = r[P + m + 1];
                    pval[i * k + j] = r[3 * P + m + 1];
                } else {
                    beta[i * k + j] = NAN; var[i * k + j] = NAN; pval[i * k + j] = NAN;
                }
            }
        }
        free(X); free(y); free(r);
    }
    free(C);
    return m;
}

/* ======================================================================================
 * gwas/mle.rs -- PARITY UNPINNED twice over: the reference has no test of it (`fn test_mle() {}`, mle.rs:470) and its numbers are
 * wherever argmin 0.8's Nelder-Mead simplex stands after <= 1000 iterations -- a crate whose source is not under /root/reference
 * (Cargo.toml:13 `argmin = "0.8.1"`, no lockfile).  What follows restates the PUBLISHED algorithm of that solver (Nelder & Mead 1965
 * as argmin 0.8 words it: alpha 1, gamma 2, rho 0.5, sigma 0.5, termination when the sample standard deviation of the vertex costs
 * drops below f64::EPSILON) around the reference's own cost function, start simplex and closing arithmetic, which ARE on disk.
 * ====================================================================================== */
/* bound_parameters_with_logit (base/helpers.rs:120-130) */
static double mle_bound(double x, double lo, double hi) { return lo + ((hi - lo) / (1.00 + exp(-x))); }

/* negative_likelihood_normal_distribution_sigma_and_beta (mle.rs:13-30), as written (note 1 / sigma2, not 1 / (2 sigma2)) */
static double mle_cost(const double *par, const double *X, const double *y, int n, int P) {
    const double sigma2 = mle_bound(par[0], ORC_EPS, 1e9);
    double ss = 0.0;
    for (int i = 0; i < n; i++) {
        double xb = 0.0;
        for (int c = 0; c < P; c++) xb = xb + X[i * P + c] * par[1 + c];
        const double e = y[i] - xb;
        ss = ss + e * e;
    }
    return ((double)n / 2.00) * log(2.00 * M_PI * sigma2) + (1.00 / sigma2) * ss;
}

/* Nelder-Mead on D parameters from the simplex of prepare_solver_neldermead(p = D, h = 1) (helpers.rs:132-146): D + 1 vertices,
 * all ones, 1.5 on the diagonal.  best: the parameters of the lowest vertex when the run ends.  Returns the iterations done. */
static int mle_nelder_mead(const double *X, const double *y, int n, int P, double *best) {
    const int D = P + 1, V = D + 1;
    double *sx = (double *)malloc(sizeof(double) * V * D), *cost = (double *)malloc(sizeof(double) * V);
    double *x0 = (double *)malloc(sizeof(double) * D * 4), *xr = x0 + D, *xe = xr + D, *tmp = xe + D;
    for (int i = 0; i < V; i++)
        for (int j = 0; j < D; j++) sx[i * D + j] = (i == j) ? 1.5 : 1.0;
    for (int i = 0; i < V; i++) cost[i] = mle_cost(&sx[i * D], X, y, n, P);
#define NM_SORT()                                                                                               \
    for (int a = 1; a < V; a++) { /* stable insertion sort by cost */                                           \
        const double ca = cost[a];                                                                              \
        memcpy(tmp, &sx[a * D], sizeof(double) * D);                                                            \
        int b = a - 1;                                                                                          \
        while (b >= 0 && cost[b] > ca) { cost[b + 1] = cost[b]; memcpy(&sx[(b + 1) * D], &sx[b * D], sizeof(double) * D); b--; } \
        cost[b + 1] = ca;                                                                                       \
        memcpy(&sx[(b + 1) * D], tmp, sizeof(double) * D);                                                      \
    }
    NM_SORT();
    int it = 0;
    for (; it < 1000; it++) { /* .configure(|state| state.max_iters(1_000)), mle.rs:98 */
        double mean = 0.0, sd = 0.0;
        for (int i = 0; i < V; i++) mean += cost[i];
        mean /= (double)V;
        for (int i = 0; i < V; i++) sd += (cost[i] - mean) * (cost[i] - mean);
        sd = sqrt(sd / ((double)V - 1.0));
        if (sd < ORC_EPS) break; /* sd_tolerance = EPSILON */
        for (int j = 0; j < D; j++) { /* centroid of all vertices but the worst */
            double c = sx[j];
            for (int i = 1; i < V - 1; i++) c += sx[i * D + j];
            x0[j] = c * (1.0 / ((double)V - 1.0));
        }
        const double *xw = &sx[(V - 1) * D];
        for (int j = 0; j < D; j++) xr[j] = x0[j] + (x0[j] - xw[j]) * 1.0;
        const double cr = mle_cost(xr, X, y, n, P);
        if (cr < cost[V - 2] && cr >= cost[0]) {
            memcpy(&sx[(V - 1) * D], xr, sizeof(double) * D); cost[V - 1] = cr;
        } else if (cr < cost[0]) {
            for (int j = 0; j < D; j++) xe[j] = x0[j] + (xr[j] - x0[j]) * 2.0;
            const double ce = mle_cost(xe, X, y, n, P);
            if (ce < cr) { memcpy(&sx[(V - 1) * D], xe, sizeof(double) * D); cost[V - 1] = ce; }
            else { memcpy(&sx[(V - 1) * D], xr, sizeof(double) * D); cost[V - 1] = cr; }
        } else {
            for (int j = 0; j < D; j++) xe[j] = x0[j] + (xw[j] - x0[j]) * 0.5;
            const double cc = mle_cost(xe, X, y, n, P);
            if (cc < cost[V - 1]) { memcpy(&sx[(V - 1) * D], xe, sizeof(double) * D); cost[V - 1] = cc; }
            else {
                for (int i = 1; i < V; i++) {
                    for (int j = 0; j < D; j++) sx[i * D + j] = sx[j] + (sx[i * D + j] - sx[j]) * 0.5;
                    cost[i] = mle_cost(&sx[i * D], X, y, n, P);
                }
            }
        }
        NM_SORT();
    }
#undef NM_SORT
    memcpy(best, sx, sizeof(double) * D);
    free(sx); free(cost); free(x0);
    return it;
}

/* estimate_effects / estimate_variances / estimate_significance of UnivariateMaximumLikelihoodEstimation (mle.rs:84-192), n >= p */
int orc_mle_fit(const double *X, const double *y, int n, int P, double *b, double *v_b, double *t, double *pval) {
    double *par = (double *)malloc(sizeof(double) * (P + 1));
    mle_nelder_mead(X, y, n, P, par);
    const double ve = mle_bound(par[0], ORC_EPS, 1e9); /* mle.rs:112 */
    for (int c = 0; c < P; c++) b[c] = par[1 + c];
    free(par);
    double *xtx = (double *)malloc(sizeof(double) * P * P), *inv = (double *)malloc(sizeof(double) * P * P);
    for (int r = 0; r < P; r++)
        for (int c = 0; c < P; c++) {
            double s = 0.0;
            for (int i = 0; i < n; i++) s = s + X[i * P + r] * X[i * P + c];
            xtx[r * P + c] = s;
        }
    int rc = 0;
    if (orc_lu_inverse(xtx, P, inv) != 0 || orc_lu_det(inv, P) == 0.0) rc = -1; /* mle.rs:140-147 */
    if (!rc)
        for (int i = 0; i < P; i++) {
            v_b[i] = ve * inv[i * P + i];
            t[i] = b[i] / v_b[i]; /* as written (mle.rs:175): the variance, not its square root */
            if (isinf(t[i])) pval[i] = 0.0;
            else if (isnan(t[i])) pval[i] = 1.0;
            else pval[i] = 2.00 * (1.00 - orc_students_t_cdf(fabs(t[i]), (double)n - 1.0));
        }
    free(xtx); free(inv);
    return rc;
}

/* mle_with_covariate numeric core (mle.rs:307-400): the kinship preamble of ols_with_covariate, then one Nelder-Mead fit per cell */
int orc_mle_with_covariate(const double *G, int64_t p, int n, int64_t ld, const double *Y, int k, double var_explained, int force_m,
                           const double *covariate_in, double *beta, double *var, double *pval, int n_threads) {
    int m;
    double *C = NULL;
    if (covariate_in && force_m >= 0) {
        m = force_m;
        C = (double *)malloc(sizeof(double) * n * (m > 0 ? m : 1));
        memcpy(C, covariate_in, sizeof(double) * n * m);
    } else {
        double *K = (double *)malloc(sizeof(double) * n * n), *ev = (double *)malloc(sizeof(double) * n), *V = (double *)malloc(sizeof(double) * n * n);
        orc_kinship(G, p, n, ld, K, n_threads);
        orc_sym_eig(K, n, ev, V);
        m = force_m >= 0 ? force_m : orc_n_eigenvecs_rule(ev, n, var_explained);
        C = (double *)malloc(sizeof(double) * n * (m > 0 ? m : 1));
        for (int i = 0; i < n; i++)
            for (int j = 0; j < m; j++) C[i * m + j] = V[i * n + j];
        free(K); free(ev); free(V);
    }
    const int P = m + 2;
    int nt = 1;
#ifdef _OPENMP
    nt = n_threads > 0 ? n_threads : (omp_get_max_threads() < 8 ? omp_get_max_threads() : 8);
#endif
#pragma omp parallel num_threads(nt)
    {
        double *X = (double *)malloc(sizeof(double) * n * P), *y = (double *)malloc(sizeof(double) * n), *r = (double *)malloc(sizeof(double) * P * 4);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < p; i++)
            for (int j = 0; j < k; j++) {
                for (int i_ = 0; i_ < n; i_++) {
                    X[i_ * P] = 1.0;
                    for (int j_ = 1; j_ < m + 1; j_++) X[i_ * P + j_] = C[i_ * m + j_ - 1];
                    X[i_ * P + m + 1] = G[i * ld + i_];
                    y[i_] = Y[i_ * k + j];
                }
                if (orc_mle_fit(X, y, n, P, r, r + P, r + 2 * P, r + 3 * P) == 0) {
                    beta[i * k + j] = r[m + 1]; var[i * k + j] = r[P + m + 1]; pval[i * k + j] = r[3 * P + m + 1];
                } else { beta[i * k + j] = NAN; var[i * k + j] = NAN; pval[i * k + j] = NAN; }
            }
        free(X); free(y); free(r);
    }
    free(C);
    return m;
}

/* ======================================================================================
 * gwas/correlation_test.rs
 * ====================================================================================== */
/* pearsons_correlation (correlation_test.rs:7-71), method = "sensible_corr" */
void orc_pearsons_correlation(const double *x, int64_t sx, const double *y, int64_t sy, int n,
                              double *r_out, double *p_out) {
    double *xf = (double *)malloc(sizeof(double) * (n > 0 ? n : 1) * 4);
    double *yf = xf + n, *xx = xf + 2 * n, *yy = xf + 3 * n;
    int m = 0;
    for (int i = 0; i < n; i++) {
        double a = x[i * sx], b = y[i * sy];
        if (!isnan(a) && !isnan(b)) { xf[m] = a; yf[m] = b; m++; }
    }
    double mu_x = orc_mean_ignore_nan(xf, m, 1), mu_y = orc_mean_ignore_nan(yf, m, 1);
    for (int i = 0; i < m; i++) {
        double dx = xf[i] - mu_x, dy = yf[i] - mu_y;
        xx[i] = dx * dx; yy[i] = dy * dy; xf[i] = dx * dy;
    }
    double numerator = orc_ndarray_sum(xf, m);
    double denominator = sqrt(orc_ndarray_sum(xx, m)) * sqrt(orc_ndarray_sum(yy, m));
    free(xf);
    double r = numerator / denominator;
    if (isnan(r)) { *r_out = NAN; *p_out = NAN; return; }
    double sden = (1.0 - r * r) / ((double)n - 2.0);
    if (sden <= 0.0) { *r_out = r; *p_out = ORC_EPS; return; }
    double t = r / sqrt(sden);
    double pv = (n > 2) ? 2.00 * (1.00 - orc_students_t_cdf(fabs(t), (double)n - 2.0)) : NAN;
    *r_out = orc_sensible_round(r, 7);
    *p_out = pv;
}

/* correlation operator (correlation_test.rs:73-129), numeric part */
int orc_correlation_locus(const uint64_t *counts, int n, const double *Y, int k,
                          const double *pool_sizes, const orc_filter *f, orc_locus_hdr *hdr,
                          double *corr, double *pval) {
    hdr->n_alleles = 0;
    int ids[6];
    uint64_t *fc = (uint64_t *)malloc(sizeof(uint64_t) * n * 6);
    int a = orc_filter_locus(counts, n, pool_sizes, f, ids, fc);
    if (a == 0) { free(fc); return 0; }
    double *fr = (double *)malloc(sizeof(double) * n * a);
    orc_to_frequencies(fc, n, a, fr);
    int p = a >= 2 ? a - 1 : a; /* drop the LAST column, unsorted (:95-98) */
    for (int i = 0; i < p; i++) {
        hdr->allele_ids[i] = ids[i];
        double s = 0.0;
        for (int r = 0; r < n; r++) s = s + fr[r * a + i];
        hdr->mean_freq[i] = s / (double)n;
        for (int j = 0; j < k; j++)
            orc_pearsons_correlation(fr + i, a, Y + j, k, n, &corr[i * k + j], &pval[i * k + j]);
    }
    hdr->n_alleles = p;
    free(fc); free(fr);
    return p;
}

int orc_correlation_csv(const char *chrom, uint64_t pos, const uint64_t *counts, int n,
                        const double *Y, int k, const double *pool_sizes, const orc_filter *f,
                        char *out, int cap) {
    orc_locus_hdr h;
    double *c = (double *)malloc(sizeof(double) * 5 * k * 2);
    double *pv = c + 5 * k;
    int na = orc_correlation_locus(counts, n, Y, k, pool_sizes, f, &h, c, pv);
    int o = 0;
    char s1[400], s2[400], s3[400];
    for (int i = 0; i < na; i++)
        for (int j = 0; j < k; j++) {
            orc_fmt_display(h.mean_freq[i], s1, sizeof s1);
            orc_parse_f64_roundup_and_own(c[i * k + j], 6, s2, sizeof s2);
            orc_fmt_display(pv[i * k + j], s3, sizeof s3);
            o += snprintf(out + o, cap - o, "%s,%llu,%c,%s,Pheno_%d,%s,%s\n", chrom,
                          (unsigned long long)pos, ALLELES[h.allele_ids[i]], s1, j, s2, s3);
        }
    free(c);
    return o;
}

/* ======================================================================================
 * tables/chisq_test.rs
 * ====================================================================================== */
int orc_chisq_locus(const uint64_t *counts, int n, const double *pool_sizes, const orc_filter *f,
                    int *allele_ids, double *chi2_out, double *pval_out) {
    uint64_t *fc = (uint64_t *)malloc(sizeof(uint64_t) * n * 6);
    int a = orc_filter_locus(counts, n, pool_sizes, f, allele_ids, fc);
    if (a == 0) { free(fc); return 0; }
    double *fr = (double *)malloc(sizeof(double) * n * a);
    orc_to_frequencies(fc, n, a, fr);
    double t = (double)(n * a);                 /* :17 */
    double total = orc_ndarray_sum(fr, (int64_t)n * a); /* matrix.sum(), :18 */
    double rs[1], *row_sums = (double *)malloc(sizeof(double) * n);
    (void)rs;
    double cs[6];
    for (int i = 0; i < n; i++) row_sums[i] = orc_ndarray_sum(fr + i * a, a); /* lane.sum() */
    for (int j = 0; j < a; j++) cs[j] = 0.0;
    for (int i = 0; i < n; i++) /* sum_axis(Axis(0)): res = res + row */
        for (int j = 0; j < a; j++) cs[j] = cs[j] + fr[i * a + j];
    double chi2 = 0.0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < a; j++) {
            double observed = fr[i * a + j];
            double expected = (row_sums[i] * cs[j]) / total;
            double d = observed - expected;
            chi2 += (d * d) / expected;
        }
    *chi2_out = chi2;
    *pval_out = 1.00 - orc_chisq_cdf(chi2, t - 1.0); /* :33-35 */
    free(fc); free(fr); free(row_sums);
    return a;
}

int orc_chisq_csv(const char *chrom, uint64_t pos, const uint64_t *counts, int n,
                  const double *pool_sizes, const orc_filter *f, char *out, int cap) {
    int ids[6];
    double chi2, pv;
    int a = orc_chisq_locus(counts, n, pool_sizes, f, ids, &chi2, &pv);
    if (a == 0) return 0;
    char al[8], s1[400], s2[400];
    for (int j = 0; j < a; j++) al[j] = ALLELES[ids[j]];
    al[a] = 0;
    orc_parse_f64_roundup_and_own(chi2, 6, s1, sizeof s1);
    orc_fmt_display(pv, s2, sizeof s2);
    return snprintf(out, cap, "%s,%llu,%s,%s,%s\n", chrom, (unsigned long long)pos, al, s1, s2);
}

/* ======================================================================================
 * gp: helpers.rs:151-255, gp/ols.rs:8-101, gp/penalise.rs:248-357
 * ====================================================================================== */
void orc_multiply_views_xx(const double *a, int a_ld, const double *b, int b_ld,
                           const int64_t *a_rows, int n_a_rows, const int64_t *a_cols,
                           const int64_t *b_rows, int n_inner, const int64_t *b_cols, int n_b_cols,
                           double *out) {
    for (int i = 0; i < n_a_rows; i++)
        for (int j = 0; j < n_b_cols; j++) {
            double x = 0.0;
            for (int k = 0; k < n_inner; k++)
                x += a[a_rows[i] * a_ld + a_cols[k]] * b[b_rows[k] * b_ld + b_cols[j]];
            out[i * n_b_cols + j] = x;
        }
}

void orc_multiply_views_xtx(const double *a, int a_ld, const double *b, int b_ld,
                            const int64_t *a_rows, int n_inner, const int64_t *a_cols, int n_a_cols,
                            const int64_t *b_rows, const int64_t *b_cols, int n_b_cols, double *out) {
    for (int i = 0; i < n_a_cols; i++)
        for (int j = 0; j < n_b_cols; j++) {
            double x = 0.0;
            for (int k = 0; k < n_inner; k++)
                x += a[a_rows[k] * a_ld + a_cols[i]] * b[b_rows[k] * b_ld + b_cols[j]];
            out[i * n_b_cols + j] = x;
        }
}

void orc_multiply_views_xxt(const double *a, int a_ld, const double *b, int b_ld,
                            const int64_t *a_rows, int n_a_rows, const int64_t *a_cols, int n_inner,
                            const int64_t *b_rows, int n_b_rows, const int64_t *b_cols, double *out) {
    for (int i = 0; i < n_a_rows; i++)
        for (int j = 0; j < n_b_rows; j++) {
            double x = 0.0;
            for (int k = 0; k < n_inner; k++)
                x += a[a_rows[i] * a_ld + a_cols[k]] * b[b_rows[j] * b_ld + b_cols[k]];
            out[i * n_b_rows + j] = x;
        }
}

/* gp::ols (gp/ols.rs:8-101).  Reference layout is X[n x P] row-major; here the transposed
 * (locus-major) storage Xt[P x n] is used, i.e. X[(i, j)] = Xt[j * ld + i]. */
int orc_gp_ols(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k,
               const int64_t *row_idx, int n_rows, double *beta, int n_threads) {
    double s0 = 0.0;
    for (int i = 0; i < n; i++) s0 = s0 + Xt[i];
    if (s0 < (double)n) return -1; /* gp/ols.rs:26-31 */
    int nt = 1;
#ifdef _OPENMP
    nt = n_threads > 0 ? n_threads : (omp_get_max_threads() < 8 ? omp_get_max_threads() : 8); /* a test oracle: never the whole box */
#endif
    (void)n_threads;
    int r = n_rows;
    if ((int64_t)n < P) {
        /* X X^T over the selected rows (multiply_views_xxt, helpers.rs:222-255) */
        double *xxt = (double *)calloc((size_t)r * r, sizeof(double));
#pragma omp parallel for num_threads(nt) schedule(static)
        for (int a = 0; a < r; a++)
            for (int b = 0; b < r; b++) {
                double x = 0.0;
                for (int64_t c = 0; c < P; c++) x += Xt[c * ld + row_idx[a]] * Xt[c * ld + row_idx[b]];
                xxt[a * r + b] = x;
            }
        double *pinv = (double *)malloc(sizeof(double) * r * r);
        orc_pinv_sym(xxt, r, pinv);
        /* (X^T pinv) y evaluated as the reference does: T = X^T pinv (P x r), b = T y */
#pragma omp parallel for num_threads(nt) schedule(static)
        for (int64_t c = 0; c < P; c++) {
            double *trow = (double *)malloc(sizeof(double) * r);
            for (int b = 0; b < r; b++) {
                double x = 0.0;
                for (int a = 0; a < r; a++) x += Xt[c * ld + row_idx[a]] * pinv[a * r + b];
                trow[b] = x;
            }
            for (int j = 0; j < k; j++) {
                double x = 0.0;
                for (int a = 0; a < r; a++) x += trow[a] * Y[row_idx[a] * k + j];
                beta[c * k + j] = x;
            }
            free(trow);
        }
        free(xxt); free(pinv);
    } else {
        int Pi = (int)P;
        double *xtx = (double *)calloc((size_t)Pi * Pi, sizeof(double));
        for (int a = 0; a < Pi; a++)
            for (int b = 0; b < Pi; b++) {
                double x = 0.0;
                for (int i = 0; i < r; i++) x += Xt[a * ld + row_idx[i]] * Xt[b * ld + row_idx[i]];
                xtx[a * Pi + b] = x;
            }
        double *pinv = (double *)malloc(sizeof(double) * Pi * Pi);
        orc_pinv_sym(xtx, Pi, pinv);
        double *T = (double *)malloc(sizeof(double) * Pi * r); /* pinv X^T (P x r) */
        for (int a = 0; a < Pi; a++)
            for (int i = 0; i < r; i++) {
                double x = 0.0;
                for (int b = 0; b < Pi; b++) x += pinv[a * Pi + b] * Xt[b * ld + row_idx[i]];
                T[a * r + i] = x;
            }
        for (int a = 0; a < Pi; a++)
            for (int j = 0; j < k; j++) {
                double x = 0.0;
                for (int i = 0; i < r; i++) x += T[a * r + i] * Y[row_idx[i] * k + j];
                beta[a * k + j] = x;
            }
        free(xtx); free(pinv); free(T);
    }
    return 0;
}

/* expand_and_contract (gp/penalise.rs:248-357) */
void orc_expand_and_contract(const double *b_in, const double *b_proxy, int64_t P, int k,
                             double alpha, double lambda, double *out) {
    memcpy(out, b_in, sizeof(double) * P * k);
    int64_t q = P - 1;
    double *normed = (double *)malloc(sizeof(double) * (q > 0 ? q : 1) * 2);
    double *nprox = normed + q;
    for (int j = 0; j < k; j++) {
        double intercept = out[j];
        for (int64_t i = 0; i < q; i++) {
            double v = out[(i + 1) * k + j], w = b_proxy[(i + 1) * k + j];
            normed[i] = ((1.00 - alpha) * (v * v) / 1.00) + (alpha * fabs(v));
            nprox[i] = ((1.00 - alpha) * (w * w) / 1.00) + (alpha * fabs(w));
        }
        double mx = nprox[0];
        for (int64_t i = 0; i < q; i++)
            if (nprox[i] > mx) mx = nprox[i];
        double sub_pen = 0.0, add_pen = 0.0, sub_dep = 0.0, add_dep = 0.0;
        for (int64_t i = 0; i < q; i++) {
            if (!(nprox[i] / mx < lambda)) continue;
            double *bv = &out[(i + 1) * k + j];
            if (*bv >= 0.0) {
                if ((*bv - normed[i]) < 0.0) { sub_pen += *bv; *bv = 0.0; }
                else { sub_pen += normed[i]; *bv -= normed[i]; }
            } else {
                if ((*bv + normed[i]) > 0.0) { add_pen += fabs(*bv); *bv = 0.0; }
                else { add_pen += normed[i]; *bv += normed[i]; }
            }
        }
        for (int64_t i = 0; i < q; i++) {
            if (!(nprox[i] / mx >= lambda)) continue;
            if (out[(i + 1) * k + j] >= 0.0) sub_dep += normed[i];
            else add_dep += normed[i];
        }
        if ((sub_pen > 0.0) & (sub_dep == 0.0)) { add_pen -= sub_pen; sub_pen = 0.0; }
        else if ((add_pen > 0.0) & (add_dep == 0.0)) { sub_pen -= add_pen; add_pen = 0.0; }
        for (int64_t i = 0; i < q; i++) {
            if (!(nprox[i] / mx >= lambda)) continue;
            double *bv = &out[(i + 1) * k + j];
            if (*bv >= 0.0) *bv += sub_pen * (normed[i] / sub_dep);
            else *bv -= add_pen * (normed[i] / add_dep);
        }
        out[j] = intercept;
    }
    free(normed);
}

/* ======================================================================================
 * gp/penalise.rs: error_index (:359-426) and the ridge-like lambda path (:133-140, :461-669)
 * ====================================================================================== */
/* error_index for trait j on the validation rows; Xt locus-major (P x n, ld), b P x k */
void orc_error_index(const double *Xt, int64_t P, int n, int64_t ld, const double *b, int k,
                     const double *Y, const int64_t *idx_val, int n_val, double *err_out) {
    (void)n;
    double *yt = (double *)malloc(sizeof(double) * n_val * 2);
    double *yp = yt + n_val;
    for (int j = 0; j < k; j++) {
        for (int i = 0; i < n_val; i++) {
            yt[i] = Y[idx_val[i] * k + j];
            double x = 0.0; /* multiply_views_xx: sequential over the columns (helpers.rs:176-180) */
            for (int64_t c = 0; c < P; c++) x += Xt[c * ld + idx_val[i]] * b[c * k + j];
            yp[i] = x;
        }
        double mn = yt[0], mx = yt[0];
        for (int i = 0; i < n_val; i++) { if (yt[i] < mn) mn = yt[i]; if (yt[i] > mx) mx = yt[i]; }
        double cor, pv;
        orc_pearsons_correlation(yt, 1, yp, 1, n_val, &cor, &pv);
        double mae = 0.0, mse = 0.0;
        for (int i = 0; i < n_val; i++) { double d = yt[i] - yp[i]; mae += fabs(d); }
        mae = mae / (mx - mn);
        for (int i = 0; i < n_val; i++) { double d = yt[i] - yp[i]; mse += d * d; }
        mse = mse / ((mx - mn) * (mx - mn));
        double rmse = sqrt(mse) / (mx - mn);
        err_out[j] = ((1.0 - fabs(cor)) + mae + mse + rmse) / 4.0; /* :409-418 */
    }
    free(yt);
}

/* gp::ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199), as written: the "kinship" is X_c X_c^T over the
 * training rows, X_c = columns 0..P-2 of x (the intercept is column 0; the LAST locus is left out, :115) centred by
 * the column means over the FIRST n_rows rows of x (not over row_idx, :124-129); PC1 = eigenvector 0 (:177; LAPACK
 * dgeev order read as "leading", as for gwas/ols.rs:296); b[0] = trait means (:170-172), b[j] = solution[2] of the
 * least-squares fit y ~ [1 | PC1 | x_j] (:193, gelsd: minimum-norm when rank deficient).  b: P x k. */
int orc_gp_proxy(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k, const int64_t *row_idx,
                 int nr, double *b, int n_threads) {
    (void)n;
    int nt = n_threads > 0 ? n_threads : 1;
    const int64_t pc = P - 1;
    double *xc = (double *)malloc(sizeof(double) * (size_t)nr * (pc > 0 ? pc : 1));
    for (int64_t j = 0; j < pc; j++) {
        double mean = 0.0;
        for (int i_ = 0; i_ < nr; i_++) mean += Xt[j * ld + i_];
        mean = mean / (double)nr;
        for (int a = 0; a < nr; a++) xc[(size_t)a * pc + j] = Xt[j * ld + row_idx[a]] - mean;
    }
    double *xxt = (double *)malloc(sizeof(double) * nr * nr);
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int a = 0; a < nr; a++)
        for (int c = 0; c < nr; c++) {
            double x = 0.0;
            for (int64_t j = 0; j < pc; j++) x += xc[(size_t)a * pc + j] * xc[(size_t)c * pc + j];
            xxt[a * nr + c] = x;
        }
    double *evals = (double *)malloc(sizeof(double) * nr);
    double *V = (double *)malloc(sizeof(double) * nr * nr);
    orc_sym_eig(xxt, nr, evals, V);
    double *ev = (double *)malloc(sizeof(double) * nr);
    for (int a = 0; a < nr; a++) ev[a] = V[a * nr + 0];
    for (int j_ = 0; j_ < k; j_++) {
        double m = 0.0;
        for (int a = 0; a < nr; a++) m += Y[row_idx[a] * k + j_];
        b[j_] = m / (double)nr;
    }
    /* orthonormal basis of [1 | PC1] */
    double *q1 = (double *)malloc(sizeof(double) * nr * 2);
    double *q2 = q1 + nr;
    {
        double s = 1.0 / sqrt((double)nr), d = 0.0, nn = 0.0;
        for (int a = 0; a < nr; a++) q1[a] = s;
        for (int a = 0; a < nr; a++) d += q1[a] * ev[a];
        for (int a = 0; a < nr; a++) { q2[a] = ev[a] - d * q1[a]; nn += q2[a] * q2[a]; }
        nn = sqrt(nn);
        for (int a = 0; a < nr; a++) q2[a] = nn > 0.0 ? q2[a] / nn : 0.0;
    }
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t j = 1; j < P; j++) {
        double *r = (double *)malloc(sizeof(double) * nr);
        double xx = 0.0, rr = 0.0;
        for (int a = 0; a < nr; a++) { r[a] = Xt[j * ld + row_idx[a]]; xx += r[a] * r[a]; }
        for (int pass = 0; pass < 2; pass++) {
            double d1 = 0.0, d2 = 0.0;
            for (int a = 0; a < nr; a++) { d1 += q1[a] * r[a]; }
            for (int a = 0; a < nr; a++) r[a] -= d1 * q1[a];
            for (int a = 0; a < nr; a++) { d2 += q2[a] * r[a]; }
            for (int a = 0; a < nr; a++) r[a] -= d2 * q2[a];
        }
        for (int a = 0; a < nr; a++) rr += r[a] * r[a];
        if (rr > 1e-28 * xx) { /* full rank: the locus coefficient is <r, y> / <r, r> */
            for (int j_ = 0; j_ < k; j_++) {
                double ry = 0.0;
                for (int a = 0; a < nr; a++) ry += r[a] * Y[row_idx[a] * k + j_];
                b[j * k + j_] = ry / rr;
            }
        } else { /* rank deficient: minimum-norm least squares = pinv(X^T X) X^T y */
            double xtx[9], pinv[9], xs[3];
            for (int u = 0; u < 9; u++) xtx[u] = 0.0;
            for (int a = 0; a < nr; a++) {
                xs[0] = 1.0; xs[1] = ev[a]; xs[2] = Xt[j * ld + row_idx[a]];
                for (int u = 0; u < 3; u++)
                    for (int v = 0; v < 3; v++) xtx[u * 3 + v] += xs[u] * xs[v];
            }
            orc_pinv_sym(xtx, 3, pinv);
            for (int j_ = 0; j_ < k; j_++) {
                double xty[3] = {0.0, 0.0, 0.0};
                for (int a = 0; a < nr; a++) {
                    double y = Y[row_idx[a] * k + j_];
                    xty[0] += y; xty[1] += ev[a] * y; xty[2] += Xt[j * ld + row_idx[a]] * y;
                }
                b[j * k + j_] = pinv[6] * xty[0] + pinv[7] * xty[1] + pinv[8] * xty[2];
            }
        }
        free(r);
    }
    free(xc); free(xxt); free(evals); free(V); free(ev); free(q1);
    return 0;
}

/* The fits inside the penalised path go through this pointer: NULL = orc_gp_ols (the literal restatement).  The exact
 * arbiter's tests install exq_gp_ols (oracle/poolgen_exact.c, same signature) so that everything DOWNSTREAM of the fits
 * (expand_and_contract, error_index, the arg-min / mode rules) is evaluated on binary128-accurate coefficients. */
typedef int (*orc_gp_ols_fn)(const double *, int64_t, int, int64_t, const double *, int, const int64_t *, int, double *, int);
static orc_gp_ols_fn orc_gp_ols_hook = NULL;
void orc_set_gp_ols_hook(void *fn) { orc_gp_ols_hook = (orc_gp_ols_fn)fn; }
/* the same for the proxy coefficients of the *_with_iterative_proxy_norms models (exq_gp_proxy has orc_gp_proxy's signature) */
static orc_gp_ols_fn orc_gp_proxy_hook = NULL;
void orc_set_gp_proxy_hook(void *fn) { orc_gp_proxy_hook = (orc_gp_ols_fn)fn; }
static int path_gp_ols(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k, const int64_t *row_idx,
                       int n_rows, double *beta, int n_threads) {
    return (orc_gp_ols_hook ? orc_gp_ols_hook : orc_gp_ols)(Xt, P, n, ld, Y, k, row_idx, n_rows, beta, n_threads);
}

/* penalised_lambda_path_with_k_fold_cross_validation (:461-669), every mode: alpha >= 0 one lambda path (a = 1),
 * alpha < 0 the grid of path values for alpha too (a = l, :479-498); iterative: the proxy coefficients above, fitted
 * on row_idx (:543, :656), pick the penalised set.  The reference draws the folds with an unseeded rand::thread_rng
 * (:452-453); here the fold of row_idx[i] in repetition rep is given: fold_of[rep * n_rows + i] in 0..nfolds-1
 * (nfolds itself: the left-over group, never validated).
 * perf (may be NULL): r x nfolds x A x L x k.  Returns L; alphas_out (may be NULL) / lambdas_out k, beta P x k. */
int orc_penalised_path_general(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k,
                               const int64_t *row_idx, int n_rows, const int32_t *fold_of, int r, int nfolds,
                               double alpha, int iterative, double lambda_step, double *beta, double *alphas_out,
                               double *lambdas_out, double *perf, int n_threads) {
    const int max_usize = (int)round(1.0 / lambda_step);
    const int L = max_usize + 1;
    const int A = alpha >= 0.0 ? 1 : L;
    double *path = (double *)malloc(sizeof(double) * L);
    for (int i = 0; i < L; i++) path[i] = (double)i / (double)max_usize;
    double *perf_l = perf ? perf : (double *)malloc(sizeof(double) * r * nfolds * A * L * k);
    double *b_hat = (double *)malloc(sizeof(double) * P * k * 3);
    double *b_new = b_hat + P * k;
    double *b_proxy = b_new + P * k;
    if (iterative) (orc_gp_proxy_hook ? orc_gp_proxy_hook : orc_gp_proxy)(Xt, P, n, ld, Y, k, row_idx, n_rows, b_proxy, n_threads);
    int64_t *itr = (int64_t *)malloc(sizeof(int64_t) * n_rows * 2);
    int64_t *iva = itr + n_rows;
    for (int rep = 0; rep < r; rep++)
        for (int fold = 0; fold < nfolds; fold++) {
            int nt = 0, nv = 0;
            for (int i = 0; i < n_rows; i++) {
                if (fold_of[rep * n_rows + i] == fold) iva[nv++] = row_idx[i];
                else itr[nt++] = row_idx[i];
            }
            path_gp_ols(Xt, P, n, ld, Y, k, itr, nt, b_hat, n_threads); /* :526 */
            for (int a = 0; a < A; a++)
                for (int li = 0; li < L; li++) {
                    orc_expand_and_contract(b_hat, iterative ? b_proxy : b_hat, P, k, alpha >= 0.0 ? alpha : path[a], path[li], b_new);
                    orc_error_index(Xt, P, n, ld, b_new, k, Y, iva, nv, &perf_l[(((rep * nfolds + fold) * A + a) * L + li) * k]);
                }
        }
    path_gp_ols(Xt, P, n, ld, Y, k, row_idx, n_rows, b_hat, n_threads); /* :573 */
    memcpy(beta, b_hat, sizeof(double) * P * k);
    int *acount = (int *)malloc(sizeof(int) * L * 2);
    int *lcount = acount + L;
    for (int j = 0; j < k; j++) {
        for (int c = 0; c < L; c++) { acount[c] = 0; lcount[c] = 0; }
        for (int rep = 0; rep < r; rep++) { /* per repetition: arg-min of the mean error across folds (:584-612) */
            double minv = 0.0;
            int arg = -1;
            for (int pass = 0; pass < 2 && arg < 0; pass++)
                for (int q = 0; q < A * L && arg < 0; q++) {
                    double s = 0.0;
                    for (int fold = 0; fold < nfolds; fold++) s += perf_l[(((rep * nfolds + fold) * A + q / L) * L + q % L) * k + j];
                    double m = s / (double)nfolds;
                    if (pass == 0) { if (q == 0 || m < minv) minv = m; }
                    else if (m == minv) arg = q;
                }
            if (arg >= 0) {
                double aval = alpha >= 0.0 ? alpha : path[arg / L], lval = path[arg % L];
                for (int c = 0; c < L; c++) { acount[c] += (aval == path[c]); lcount[c] += (lval == path[c]); }
            }
        }
        int abest = 0, lbest = 0, amx = 0, lmx = 0; /* mode, first maximum (:614-627) */
        for (int c = 0; c < L; c++) { if (acount[c] > amx) amx = acount[c]; if (lcount[c] > lmx) lmx = lcount[c]; }
        for (int c = 0; c < L; c++) if (acount[c] == amx) { abest = c; break; }
        for (int c = 0; c < L; c++) if (lcount[c] == lmx) { lbest = c; break; }
        /* an alpha >= 0 that is on the path grid is its own mode (0 and 1, the only values the reference passes) */
        double afinal = alpha >= 0.0 ? alpha : path[abest];
        if (alphas_out) alphas_out[j] = afinal;
        lambdas_out[j] = path[lbest];
        orc_expand_and_contract(b_hat, iterative ? b_proxy : b_hat, P, k, afinal, path[lbest], b_new);
        for (int64_t i = 0; i < P; i++) beta[i * k + j] = b_new[i * k + j];
    }
    free(acount); free(itr); free(b_hat); free(path);
    if (!perf) free(perf_l);
    return L;
}

/* alpha >= 0, iterative = false (penalise_lasso_like / penalise_ridge_like); perf r x nfolds x L x k */
int orc_penalised_lambda_path(const double *Xt, int64_t P, int n, int64_t ld, const double *Y, int k,
                              const int64_t *row_idx, int n_rows, const int32_t *fold_of, int r,
                              int nfolds, double alpha, double lambda_step, double *beta,
                              double *lambdas_out, double *perf, int n_threads) {
    return orc_penalised_path_general(Xt, P, n, ld, Y, k, row_idx, n_rows, fold_of, r, nfolds, alpha, 0, lambda_step, beta,
                                      NULL, lambdas_out, perf, n_threads);
}

/* ============================================================================================
 * popgen: define_sliding_windows (base/helpers.rs:294-403), fst (popgen/fst.rs:10-115, :158-200),
 * theta_pi (popgen/pi.rs:10-113).  Genotypes as for gp: Xt locus-major (P x ld), row 0 = intercept.
 * loci_idx: the L + 1 column starts of count_loci (sync.rs:73-97; the first is 1, the last is P).
 * cov: L x n (the reference stores the transpose, coverages[(pool, locus)], sync.rs:1129-1152).
 * ============================================================================================ */
/* chromosomes by id (only equality is used).  head/tail: room for l entries.  Returns the number of windows. */
int64_t orc_define_sliding_windows(const int32_t *chr, const uint64_t *pos, int64_t l, uint64_t window_size_bp,
                                   uint64_t window_slide_size_bp, uint64_t min_loci_per_window, int64_t *out_head,
                                   int64_t *out_tail) {
    if (l <= 0) return 0;
    int64_t *idx_head = (int64_t *)malloc(sizeof(int64_t) * (l + 1) * 2);
    int64_t *idx_tail = idx_head + (l + 1);
    uint64_t *cov = (uint64_t *)malloc(sizeof(uint64_t) * (l + 1));
    int64_t nw = 1;
    idx_head[0] = 0; idx_tail[0] = 0; cov[0] = 1;
    int marker_next_window_head = 0;
    int64_t idx_next_head = 0, i = 1;
    while (i < l) {
        const int32_t chr_head = chr[idx_head[nw - 1]];
        const uint64_t pos_head = pos[idx_head[nw - 1]];
        if ((chr[i] != chr_head) | (pos[i] > (pos_head + window_size_bp))) {
            i = marker_next_window_head ? idx_next_head : i; /* :331-335 */
            if (cov[nw - 1] >= min_loci_per_window) {
                idx_head[nw] = i; idx_tail[nw] = i; cov[nw] = 1; nw++;
            } else { /* ditch the ending window: its slot becomes the next window's start; its tail stays (:350-357) */
                idx_head[nw - 1] = i; cov[nw - 1] = 1;
            }
            marker_next_window_head = 0;
        } else {
            idx_tail[nw - 1] = i;
            cov[nw - 1] += 1;
            if ((marker_next_window_head == 0) & (pos[i] >= (pos_head + window_slide_size_bp))) {
                marker_next_window_head = 1;
                idx_next_head = i;
            }
        }
        i += 1;
    }
    /* remove redundant tails (:380-391) */
    int64_t no = 0;
    out_head[no] = idx_head[0]; out_tail[no] = idx_tail[0]; no++;
    for (int64_t w = 1; w < nw; w++)
        if (idx_tail[w] != out_tail[no - 1]) { out_head[no] = idx_head[w]; out_tail[no] = idx_tail[w]; no++; }
    free(idx_head); free(cov);
    return no;
}

/* the per-locus guard of fst (:66): |sum over pools of (sum over alleles) - n| <= eps, in ndarray's summation orders */
static int popgen_locus_sums_to_one(const double *Xt, int64_t ld, int n, int64_t c0, int64_t c1) {
    double *rs = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) rs[i] = 0.0;
    for (int i = 0; i < n; i++) { /* sum_axis(Axis(1)): row.sum() of a contiguous lane of < 8 alleles = left to right */
        double *tmp = (double *)malloc(sizeof(double) * (c1 - c0));
        for (int64_t c = c0; c < c1; c++) tmp[c - c0] = Xt[c * ld + i];
        rs[i] = orc_ndarray_sum(tmp, c1 - c0);
        free(tmp);
    }
    double tot = orc_ndarray_sum(rs, n);
    free(rs);
    return fabs(tot - (double)n) <= ORC_EPS;
}

/* fst_mean n x n; fst_win n_windows x (n*n) (may be NULL with n_windows 0).  Returns 0, or -1 when the reference's
 * assert (:66) fails at some locus. */
int orc_fst(const double *Xt, int64_t P, int n, int64_t ld, const int64_t *loci_idx, int64_t L, const double *cov,
            const int64_t *win_head, const int64_t *win_tail, int64_t n_windows, double *fst_mean, double *fst_win) {
    (void)P;
    double *f = (double *)malloc(sizeof(double) * (size_t)L * n * n);
    for (int64_t i = 0; i < L; i++) {
        const int64_t c0 = loci_idx[i], c1 = loci_idx[i + 1];
        if (!popgen_locus_sums_to_one(Xt, ld, n, c0, c1)) { free(f); return -1; }
        for (int j = 0; j < n; j++)
            for (int k = 0; k < n; k++) {
                const double nj = cov[i * n + j], nk = cov[i * n + k];
                double sj = 0.0, sk = 0.0, q2 = 0.0;
                for (int64_t c = c0; c < c1; c++) sj = sj + Xt[c * ld + j] * Xt[c * ld + j];
                for (int64_t c = c0; c < c1; c++) sk = sk + Xt[c * ld + k] * Xt[c * ld + k];
                for (int64_t c = c0; c < c1; c++) q2 = q2 + (Xt[c * ld + j] * Xt[c * ld + k]);
                const double q1_j = (sj * (nj / (nj - 1.00 + ORC_EPS))) + (1.00 - (nj / (nj - 1.00 + ORC_EPS)));
                const double q1_k = (sk * (nk / (nk - 1.00 + ORC_EPS))) + (1.00 - (nk / (nk - 1.00 + ORC_EPS)));
                const double fu = (0.5 * (q1_j + q1_k) - q2) / (1.00 - q2 + ORC_EPS);
                f[((size_t)i * n + j) * n + k] = fu < 0.0 ? 0.0 : (fu > 1.0 ? 1.0 : fu); /* NaN passes through */
            }
    }
    for (int j = 0; j < n * n; j++) { /* mean_axis(Axis(0)): left to right from zero */
        double s = 0.0;
        for (int64_t i = 0; i < L; i++) s = s + f[(size_t)i * n * n + j];
        fst_mean[j] = s / (double)L;
    }
    for (int64_t w = 0; w < n_windows; w++)
        for (int j = 0; j < n * n; j++) {
            double s = 0.0;
            for (int64_t i = win_head[w]; i <= win_tail[w]; i++) s = s + f[(size_t)i * n * n + j];
            fst_win[(size_t)w * n * n + j] = s / (double)(win_tail[w] + 1 - win_head[w]);
        }
    free(f);
    return 0;
}

/* pi_win n_windows x n, pi_mean n (mean across windows, pi.rs:133) */
int orc_theta_pi(const double *Xt, int64_t P, int n, int64_t ld, const int64_t *loci_idx, int64_t L, const double *cov,
                 const int64_t *win_head, const int64_t *win_tail, int64_t n_windows, double *pi_win, double *pi_mean) {
    (void)P;
    double *pi = (double *)malloc(sizeof(double) * (size_t)L * n);
    for (int64_t i = 0; i < L; i++)
        for (int j = 0; j < n; j++) {
            const double nj = cov[i * n + j];
            double sj = 0.0;
            for (int64_t c = loci_idx[i]; c < loci_idx[i + 1]; c++) sj = sj + Xt[c * ld + j] * Xt[c * ld + j];
            pi[(size_t)i * n + j] = fabs((sj * (nj / (nj - 1.00 + ORC_EPS))) - (nj / (nj - 1.00 + ORC_EPS)));
        }
    for (int64_t w = 0; w < n_windows; w++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int64_t i = win_head[w]; i <= win_tail[w]; i++) s = s + pi[(size_t)i * n + j];
            pi_win[(size_t)w * n + j] = s / (double)(win_tail[w] + 1 - win_head[w]);
        }
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int64_t w = 0; w < n_windows; w++) s = s + pi_win[(size_t)w * n + j];
        pi_mean[j] = s / (double)n_windows;
    }
    free(pi);
    return 0;
}

/* ============================================================================================
 * base/pileup.rs: String::lparse -> PileupLine (:11-155), PileupLine::filter (:239-337),
 * to_counts (:160-214), pileup_to_sync (:340-371).  Literal restatement, one line at a time.
 * Return: length of the sync line written to out (> 0), 0 = the reference returns None (locus
 * dropped; `filter` errors are swallowed by `_ => return None`, :343-346), < 0 = the reference
 * PANICS for this line (lparse error under .expect(), :425-431):
 *   -1 position, -2 reference allele, -3 coverage field, -4 coverage / codes / qualities mismatch
 *   (or a ragged pool triplet), -5 indel length is not a digit.
 * ============================================================================================ */
#define ORC_PILEUP_MAXF 4096
int orc_pileup_to_sync2(const char *line, int remove_ns, int keep_lowercase_reference, double max_base_error_rate,
                        uint64_t min_coverage_depth, double min_coverage_breadth, double min_allele_frequency,
                        const double *pool_sizes, int n_pool_sizes, char *out, int cap) {
    /* split("\t") */
    const char *fs[ORC_PILEUP_MAXF];
    int fl[ORC_PILEUP_MAXF], nf = 0;
    {
        const char *p = line;
        for (;;) {
            const char *t = strchr(p, '\t');
            if (nf >= ORC_PILEUP_MAXF) return -4;
            fs[nf] = p; fl[nf] = t ? (int)(t - p) : (int)strlen(p); nf++;
            if (!t) break;
            p = t + 1;
        }
    }
    if (nf < 3) return -4;
    /* position: parse::<u64> (digits, optional leading '+') */
    uint64_t position = 0;
    {
        int i = 0, nd = 0;
        if (fl[1] > 0 && fs[1][0] == '+') i = 1;
        for (; i < fl[1]; i++) { if (fs[1][i] < '0' || fs[1][i] > '9') return -1; position = position * 10u + (uint64_t)(fs[1][i] - '0'); nd++; }
        if (nd == 0) return -1;
    }
    if (fl[2] != 1 || (unsigned char)fs[2][0] >= 128) return -2; /* parse::<char>: exactly one character */
    const char ref = fs[2][0];
    int n = 0;
    for (int i = 3; i < nf; i += 3) n++;
    uint64_t *cov = (uint64_t *)calloc(n > 0 ? n : 1, sizeof(uint64_t));
    unsigned char **codes = (unsigned char **)calloc(n > 0 ? n : 1, sizeof(unsigned char *));
    unsigned char **quals = (unsigned char **)calloc(n > 0 ? n : 1, sizeof(unsigned char *));
    int *ncode = (int *)calloc(n > 0 ? n : 1, sizeof(int)), *nqual = (int *)calloc(n > 0 ? n : 1, sizeof(int));
    int rc = 0, ncodes_parsed = 0, nquals_parsed = 0;
    for (int i = 3, pool = 0; i < nf; i += 3, pool++) {
        int j = 0, nd = 0;
        uint64_t v = 0;
        if (fl[i] > 0 && fs[i][0] == '+') j = 1;
        for (; j < fl[i]; j++) { if (fs[i][j] < '0' || fs[i][j] > '9') { rc = -3; goto done; } v = v * 10u + (uint64_t)(fs[i][j] - '0'); nd++; }
        if (nd == 0) { rc = -3; goto done; }
        cov[pool] = v;
    }
    for (int i = 4; i < nf; i += 3) { /* read codes (:38-128) */
        const int pool = ((i - 1) / 3) - 1;
        ncodes_parsed++;
        if (cov[pool] > 0) {
            codes[pool] = (unsigned char *)malloc(fl[i] > 0 ? fl[i] : 1);
            int indel = 0;
            uint64_t count = 0, left = 4294967295ull;
            for (int j = 0; j < fl[i]; j++) {
                const unsigned char code = (unsigned char)fs[i][j];
                if (indel) {
                    if (count == 0 && left == 4294967295ull) {
                        if (code < '0' || code > '9') { rc = -5; goto done; }
                        count = (uint64_t)(code - '0');
                        continue;
                    }
                    if (count > 0 && left == 4294967295ull) {
                        if (code >= '0' && code <= '9') { count = count * 10u + (uint64_t)(code - '0'); continue; }
                        left = count - 1;
                        continue;
                    }
                    if (left > 0) { left -= 1; continue; }
                    indel = 0; count = 0; left = 4294967295ull;
                }
                if (code == 43 || code == 45) { indel = 1; count = 0; continue; }
                if (code == 94 || code == 36) {
                    if (code == 94) { indel = 1; count = 1; left = 1; }
                    continue;
                }
                unsigned char a;
                if (code == 44 || code == 46) a = (unsigned char)ref;
                else switch (code) {
                    case 65: case 97: a = 65; break;
                    case 84: case 116: a = 84; break;
                    case 67: case 99: a = 67; break;
                    case 71: case 103: a = 71; break;
                    case 42: a = 68; break;
                    default: a = 78;
                }
                codes[pool][ncode[pool]++] = a;
            }
        }
    }
    for (int i = 5; i < nf; i += 3) { /* qualities (:130-139) */
        const int pool = ((i - 1) / 3) - 1;
        nquals_parsed++;
        if (cov[pool] > 0) {
            quals[pool] = (unsigned char *)malloc(fl[i] > 0 ? fl[i] : 1);
            memcpy(quals[pool], fs[i], fl[i]);
            nqual[pool] = fl[i];
        }
    }
    if (ncodes_parsed != n || nquals_parsed != n) { rc = -4; goto done; } /* the sanity loop would index out of bounds */
    for (int i = 0; i < n; i++)
        if (cov[i] != (uint64_t)ncode[i] || cov[i] != (uint64_t)nqual[i]) { rc = -4; goto done; }
    /* ---- filter (:239-337): any Err here is None for pileup_to_sync -------------------------------- */
    if (n != n_pool_sizes) { rc = 0; goto done; }
    for (int i = 0; i < n; i++) {
        int j = 0;
        while (j < ncode[i]) {
            if (quals[i][j] < 33) { rc = 0; goto done; } /* "Phred score out of bounds." */
            const double q = pow(10.0, -((double)quals[i][j] - 33.0) / 10.0);
            if (q > max_base_error_rate) codes[i][j] = 78;
            if (remove_ns && codes[i][j] == 78) {
                memmove(codes[i] + j, codes[i] + j + 1, (size_t)(ncode[i] - j - 1));
                memmove(quals[i] + j, quals[i] + j + 1, (size_t)(ncode[i] - j - 1));
                ncode[i]--;
                cov[i] -= 1;
            } else j++;
        }
    }
    {
        const uint32_t min_breadth = (uint32_t)ceil(min_coverage_breadth * (double)n_pool_sizes);
        uint32_t covered = 0;
        for (int i = 0; i < n && covered < min_breadth; i++)
            if (cov[i] >= min_coverage_depth) covered++;
        if (covered != min_breadth) { rc = 0; goto done; }
    }
    if (keep_lowercase_reference) /* :280-299, after the N removal and the coverage test: a lower-case reference allele
                                     copied in for '.' / ',' becomes its base; whatever is not A/T/C/G in either case or '*'
                                     -- the 'D' (68) that lparse made of '*' included -- becomes N */
        for (int i = 0; i < n; i++)
            for (int j = 0; j < ncode[i]; j++)
                switch (codes[i][j]) {
                    case 65: case 97: codes[i][j] = 65; break;
                    case 84: case 116: codes[i][j] = 84; break;
                    case 67: case 99: codes[i][j] = 67; break;
                    case 71: case 103: codes[i][j] = 71; break;
                    case 42: codes[i][j] = 68; break;
                    default: codes[i][j] = 78;
                }
    {
        /* to_counts (A,T,C,G,D,N) and to_frequencies: count / row sum (0/0 = NaN) */
        uint64_t cnt[6];
        double *fr = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * 6);
        uint64_t *cm = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1) * 6);
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < 6; j++) cnt[j] = 0;
            for (int j = 0; j < ncode[i]; j++)
                switch (codes[i][j]) {
                    case 65: cnt[0]++; break; case 84: cnt[1]++; break; case 67: cnt[2]++; break;
                    case 71: cnt[3]++; break; case 68: cnt[4]++; break; default: cnt[5]++;
                }
            uint64_t rs = 0;
            for (int j = 0; j < 6; j++) { cm[i * 6 + j] = cnt[j]; rs += cnt[j]; }
            for (int j = 0; j < 6; j++) fr[i * 6 + j] = (double)cnt[j] / (double)rs;
        }
        int m = 6, j = 1; /* the loop as written (:311-331): a failing column is re-tested with a smaller m */
        while (j < m) {
            double q = 0.0;
            for (int i = 0; i < n; i++) q += fr[i * 6 + j] * pool_sizes[i];
            if ((q < min_allele_frequency) | (q > (1.00 - min_allele_frequency))) m -= 1;
            else j += 1;
        }
        if (m < 2) { free(fr); free(cm); rc = 0; goto done; }
        int len = snprintf(out, cap, "%.*s\t%llu\t%c", fl[0], fs[0], (unsigned long long)position, ref);
        for (int i = 0; i < n && len < cap; i++)
            len += snprintf(out + len, cap - len, "\t%llu:%llu:%llu:%llu:%llu:%llu", (unsigned long long)cm[i * 6],
                            (unsigned long long)cm[i * 6 + 1], (unsigned long long)cm[i * 6 + 2],
                            (unsigned long long)cm[i * 6 + 3], (unsigned long long)cm[i * 6 + 4],
                            (unsigned long long)cm[i * 6 + 5]);
        if (len < cap) len += snprintf(out + len, cap - len, "\n");
        rc = len;
        free(fr); free(cm);
    }
done:
    for (int i = 0; i < n; i++) { free(codes[i]); free(quals[i]); }
    free(cov); free(codes); free(quals); free(ncode); free(nqual);
    return rc;
}

/* keep_lowercase_reference = false (the CLI default, main.rs:60-62) */
int orc_pileup_to_sync(const char *line, int remove_ns, double max_base_error_rate, uint64_t min_coverage_depth,
                       double min_coverage_breadth, double min_allele_frequency, const double *pool_sizes,
                       int n_pool_sizes, char *out, int cap) {
    return orc_pileup_to_sync2(line, remove_ns, 0, max_base_error_rate, min_coverage_depth, min_coverage_breadth,
                               min_allele_frequency, pool_sizes, n_pool_sizes, out, cap);
}
