// pg_stats_device.h -- fp64 device functions for the p-values of the per-locus tests.
//
// The reference obtains p-values from statrs 0.16: StudentsT::cdf through the continued
// fraction of the regularised incomplete beta (gwas/ols.rs:139,153; correlation_test.rs:65-66)
// and ChiSquared::cdf through the regularised lower incomplete gamma (tables/chisq_test.rs:33-35).
// On the GPU the Student-t tail is evaluated with the closed finite series that exists for
// INTEGER degrees of freedom (Abramowitz & Stegun 26.7.3/26.7.4; df = n-1 or n-2 is always an
// integer here): branch-free, no divisions in the loop, identical trip count in every lane.
// It agrees with the statrs formulation to ~1e-15 absolute, inside the 1e-10 contract; the
// oracle keeps the statrs algorithm, so the two are independent implementations.
#pragma once
#include <hip/hip_runtime.h>

#define PG_EPS 2.220446049250313e-16

// Two-sided p-value P(|T_df| > |t|).  coef/ncoef from pg_tdist_coef(df) (host): ncoef is a multiple of 8 (zero padded).
// The series sum_j coef[j] h^j has ~df/2 terms (99 at 200 pools).  A plain Horner loop is ONE dependent chain of that many
// fp64 FMAs, each behind its own 8-byte scalar load: ~15 000 cycles per call, measured as 60 % of the sweep's arithmetic
// and the reason its loads and its arithmetic did not overlap (a wave sat in this loop with nothing in flight).  Here:
// four interleaved Horner chains in h^4 (all coefficients are positive: no cancellation, any order is good to an ulp or
// two), eight coefficients per step from ONE 64-byte scalar load.
__device__ __forceinline__ double pg_t_two_sided_p(double t_abs, int df,
                                                   const double *__restrict__ coef, int ncoef) {
    if (isinf(t_abs)) return 0.0;
    // Same input quantisation as statrs StudentsT::cdf: everything is a function of the ROUNDED
    // h = nu / (nu + t*t) and of 1 - h, so that for |t| << sqrt(nu) (where h rounds towards 1 and
    // the reference's p-value is quantised to 1 - c*sqrt(1-h)) both sides see the same argument.
    const double nu = (double)df;
    const double c2 = nu / (nu + t_abs * t_abs); // h = cos^2(theta)
    const double s = sqrt(1.0 - c2);             // sin(theta)
    const double x2 = c2 * c2, y4 = x2 * x2;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll 2
    for (int b = ncoef - 8; b >= 0; b -= 8) {
        const double *cb = coef + b; // wave-uniform: one s_load_dwordx16
        p0 = fma(p0, y4, cb[4]); p1 = fma(p1, y4, cb[5]); p2 = fma(p2, y4, cb[6]); p3 = fma(p3, y4, cb[7]);
        p0 = fma(p0, y4, cb[0]); p1 = fma(p1, y4, cb[1]); p2 = fma(p2, y4, cb[2]); p3 = fma(p3, y4, cb[3]);
    }
    const double poly = fma(fma(fma(p3, c2, p2), c2, p1), c2, p0);
    double A;
    if (df & 1) {
        const double c = sqrt(c2);
        const double theta = atan2(s, c);
        A = 0.6366197723675814 * (theta + s * c * poly); // 2/pi
    } else {
        A = s * poly;
    }
    double p = 1.0 - A;
    p = p < 0.0 ? 0.0 : p;
    return p > 1.0 ? 1.0 : p;
}

// Two tails at once: the same series for two arguments behind ONE stream of coefficient loads.  For a kernel that does nothing
// but close fits (k_sweep_finish) the call is bound by the latency of its ~13 dependent 64-byte scalar loads, not by its 100
// FMAs; two loci per lane halve that latency per locus.  Bit-identical to two calls of pg_t_two_sided_p.
__device__ __forceinline__ void pg_t_two_sided_p_x2(double ta, double tb, int df, const double *__restrict__ coef, int ncoef,
                                                    double &pa, double &pb) {
    const double nu = (double)df;
    const double ca = nu / (nu + ta * ta), cb2 = nu / (nu + tb * tb);
    const double sa = sqrt(1.0 - ca), sb = sqrt(1.0 - cb2);
    const double xa = ca * ca, ya = xa * xa, xb = cb2 * cb2, yb = xb * xb;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll 2
    for (int b = ncoef - 8; b >= 0; b -= 8) {
        const double *cb = coef + b; // wave-uniform: one s_load_dwordx16 for both arguments
        a0 = fma(a0, ya, cb[4]); a1 = fma(a1, ya, cb[5]); a2 = fma(a2, ya, cb[6]); a3 = fma(a3, ya, cb[7]);
        b0 = fma(b0, yb, cb[4]); b1 = fma(b1, yb, cb[5]); b2 = fma(b2, yb, cb[6]); b3 = fma(b3, yb, cb[7]);
        a0 = fma(a0, ya, cb[0]); a1 = fma(a1, ya, cb[1]); a2 = fma(a2, ya, cb[2]); a3 = fma(a3, ya, cb[3]);
        b0 = fma(b0, yb, cb[0]); b1 = fma(b1, yb, cb[1]); b2 = fma(b2, yb, cb[2]); b3 = fma(b3, yb, cb[3]);
    }
    const double polya = fma(fma(fma(a3, ca, a2), ca, a1), ca, a0), polyb = fma(fma(fma(b3, cb2, b2), cb2, b1), cb2, b0);
    double Aa, Ab;
    if (df & 1) {
        const double c1 = sqrt(ca), c2 = sqrt(cb2);
        Aa = 0.6366197723675814 * (atan2(sa, c1) + sa * c1 * polya); // 2/pi
        Ab = 0.6366197723675814 * (atan2(sb, c2) + sb * c2 * polyb);
    } else {
        Aa = sa * polya;
        Ab = sb * polyb;
    }
    pa = 1.0 - Aa; pa = pa < 0.0 ? 0.0 : pa; pa = pa > 1.0 ? 1.0 : pa;
    pb = 1.0 - Ab; pb = pb < 0.0 ? 0.0 : pb; pb = pb > 1.0 ? 1.0 : pb;
    if (isinf(ta)) pa = 0.0;
    if (isinf(tb)) pb = 0.0;
}

// ---- statrs ln_gamma (Lanczos g = 10.900511, 11 terms) --------------------------------------
__device__ __forceinline__ double pg_ln_gamma(double x) {
    const double dk[11] = {2.48574089138753565546e-5, 1.05142378581721974210,
                           -3.45687097222016235469,   4.51227709466894823700,
                           -2.98285225323576655721,   1.05639711577126713077,
                           -1.95428773191645869583e-1, 1.70970543404441224307e-2,
                           -5.71926117404305781283e-4, 4.63399473359905636708e-6,
                           -2.71994908488607703910e-9};
    const double R = 10.900511;
    const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
    const double LN_PI = 1.1447298858494001741434273513530587116472948129153;
    if (x < 0.5) {
        double s = dk[0];
        for (int i = 1; i < 11; ++i) s += dk[i] / ((double)i - x);
        return LN_PI - log(sin(3.141592653589793 * x)) - log(s) - LN_2_SQRT_E_OVER_PI -
               (0.5 - x) * log((0.5 - x + R) / 2.718281828459045);
    }
    double s = dk[0];
    for (int i = 1; i < 11; ++i) s += dk[i] / (x + (double)i - 1.0);
    return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + R) / 2.718281828459045);
}

// ---- statrs gamma_lr (Cephes igam / igamc) ----------------------------------------------------
// Regularised lower incomplete gamma P(a, x).  ln_gamma_a = pg_ln_gamma(a) is passed in because
// a = (n*alleles - 1)/2 takes at most a handful of values per launch.
__device__ __forceinline__ double pg_gamma_lr(double a, double x, double ln_gamma_a) {
    if (isnan(a) || isnan(x)) return NAN;
    if (!(a > 0.0) || isinf(a)) return NAN;
    if (!(x > 0.0) || isinf(x)) return NAN;
    const double eps = 0.000000000000001;
    const double big = 4503599627370496.0;
    const double big_inv = 2.22044604925031308085e-16;
    if (fabs(a) < 1e-15) return 1.0;
    if (fabs(x) < 1e-15) return 0.0;
    const double ax = a * log(x) - x - ln_gamma_a;
    if (ax < -709.78271289338399) return a < x ? 1.0 : 0.0;
    if (x <= 1.0 || x <= a) {
        double r2 = a, c2 = 1.0, ans2 = 1.0;
        for (int it = 0; it < 100000; ++it) {
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
    for (int it = 0; it < 100000; ++it) {
        y += 1.0;
        z += 2.0;
        c += 1;
        const double yc = y * (double)c;
        const double p = p2 * z - p3 * yc;
        const double q = q2 * z - q3 * yc;
        p3 = p2; p2 = p; q3 = q2; q2 = q;
        if (fabs(p) > big) { p3 *= big_inv; p2 *= big_inv; q3 *= big_inv; q2 *= big_inv; }
        if (q != 0.0) {
            const double nextans = p / q;
            const double error = fabs((ans - nextans) / nextans);
            ans = nextans;
            if (error <= eps) break;
        }
    }
    return 1.0 - exp(ax) * ans;
}

// 1 - ChiSquared(df).cdf(x)  (tables/chisq_test.rs:33-35)
__device__ __forceinline__ double pg_chisq_upper_p(double x, double df, double ln_gamma_half_df) {
    double cdf;
    if (isnan(x)) return NAN;
    if (x <= 0.0) cdf = 0.0;
    else if (isinf(x)) cdf = 1.0;
    else cdf = pg_gamma_lr(df / 2.0, x * 0.5, ln_gamma_half_df);
    return 1.00 - cdf;
}
