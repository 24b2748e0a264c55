// pg_mle.hip -- mle_iter_with_kinship == gwas::mle_with_covariate (gwas/mle.rs:307-463): the kinship preamble of
// ols_with_covariate (:317-343), then for every (column, trait) a maximum-likelihood fit of y ~ [1 | C | g] by Nelder-Mead
// (argmin 0.8, <= 1000 iterations) on (logit-bounded sigma^2, b) from the reference's start simplex, and its closing arithmetic
// (ve = sigma^2, v_b = ve * diag((X'X)^-1), t = b / v_b AS WRITTEN (:175: the variance, not its square root), p = 2 (1 - T_{n-1}(|t|))).
//
// PARITY UNPINNED, and it says so in the header, in DESIGN.md and in the tests: the reference has no test of this path
// (`fn test_mle() {}`, mle.rs:470) and what it prints is wherever the simplex of a crate stands whose source is not in the
// reference tree.  The solver below is the published Nelder-Mead as argmin 0.8 words it (alpha 1, gamma 2, rho 0.5, sigma 0.5,
// stop when the sample standard deviation of the vertex costs is below f64::EPSILON), restated once more, literally, in the
// oracle; the two are compared with each other and with the analytic optimum (the OLS coefficient) at the solver's own
// resolution (~1e-6), not at 1e-10.
//
// Device design.  The cost (mle.rs:13-30) depends on the data only through sufficient statistics,
//     sum (y - X b)^2 = y'y - 2 b'X'y + b'(X'X) b,
// so ONE streaming pass over G (the coefficient pass of gp::ols with Z = [1 | C | Y], which also returns g'g) leaves m + 2 + k
// numbers per column, and the 1000 simplex steps of a fit run on registers: thread = (column, trait), P = m + 2 <= 4 design
// columns (compile-time: the simplex lives in registers, vertices sorted by an unrolled insertion network).
#include "pg_common.h"
#include "pg_stats_device.h"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

constexpr int MLE_MAXP = 4;       // design columns of the register kernel (m <= 2)
constexpr int MLE_MAXP_LDS = 10;  // ... of the kernel whose simplex lives in LDS (m <= 8, the sweep's own limit on kinship covariates)

struct MleShared {          // what all columns share: Z = [1 | C]
    double ztz[(MLE_MAXP - 1) * (MLE_MAXP - 1)];
    double zty[(MLE_MAXP - 1) * 4]; // [a][trait]
    double yty[4];
    int n, m1, k, tdf, ntcoef;
};
struct MleSharedBig {       // the same for up to MLE_MAXP_LDS design columns (lives in device memory: indexed at run time)
    double ztz[(MLE_MAXP_LDS - 1) * (MLE_MAXP_LDS - 1)]; // row-major m1 x m1
    double zty[(MLE_MAXP_LDS - 1) * 4];                  // [a][trait]
    double yty[4];
    int n, m1, k, tdf, ntcoef;
};

__device__ __forceinline__ double mle_bound(double x) { // bound_parameters_with_logit(x, EPSILON, 1e9), helpers.rs:120-130
    return PG_EPS + ((1e9 - PG_EPS) / (1.00 + exp(-x)));
}

template <int P>
struct MleStats { double A[P][P], xy[P], yy; int n; };

template <int P>
__device__ __forceinline__ double mle_cost(const double (&par)[P + 1], const MleStats<P> &S) {
    const double sigma2 = mle_bound(par[0]);
    double q = 0.0, l = 0.0;
#pragma unroll
    for (int r = 0; r < P; ++r) {
        double ab = 0.0;
#pragma unroll
        for (int c = 0; c < P; ++c) ab = fma(S.A[r][c], par[1 + c], ab);
        q = fma(par[1 + r], ab, q);
        l = fma(par[1 + r], S.xy[r], l);
    }
    double ss = S.yy - 2.0 * l + q;
    ss = ss < 0.0 ? 0.0 : ss;
    return ((double)S.n / 2.00) * log(2.00 * 3.14159265358979323846 * sigma2) + (1.00 / sigma2) * ss;
}

template <int P>
__global__ __launch_bounds__(64) void k_mle_nm(const double *__restrict__ sums /* p x (m1 + k): Z'g, Y'g */, const double *__restrict__ gg,
                                               const double *__restrict__ tcoef, int64_t p, const MleShared H,
                                               double *__restrict__ beta, double *__restrict__ var, double *__restrict__ pval) {
    constexpr int D = P + 1, V = D + 1;
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= p * H.k) return;
    const int64_t l = cell / H.k;
    const int j = (int)(cell - l * H.k);
    const int ncol = H.m1 + H.k;
    MleStats<P> S;
    S.n = H.n;
    S.yy = H.yty[j];
#pragma unroll
    for (int r = 0; r < P - 1; ++r) {
#pragma unroll
        for (int c = 0; c < P - 1; ++c) S.A[r][c] = H.ztz[r * (P - 1) + c];
        const double zg = sums[l * ncol + r];
        S.A[r][P - 1] = zg;
        S.A[P - 1][r] = zg;
        S.xy[r] = H.zty[r * 4 + j];
    }
    S.A[P - 1][P - 1] = gg[l];
    S.xy[P - 1] = sums[l * ncol + H.m1 + j];

    // ---- Nelder-Mead from prepare_solver_neldermead(P + 1, 1.0) (helpers.rs:132-146): ones, 1.5 on the diagonal -------------
    double sx[V][D], cost[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
        for (int d = 0; d < D; ++d) sx[i][d] = (i == d) ? 1.5 : 1.0;
        cost[i] = mle_cost<P>(sx[i], S);
    }
    auto sort = [&]() { // stable insertion sort by cost, fully unrolled (vertices move with their costs)
#pragma unroll
        for (int a = 1; a < V; ++a) {
#pragma unroll
            for (int b = a; b >= 1; --b) {
                const bool sw = cost[b - 1] > cost[b];
                const double c0 = cost[b - 1], c1 = cost[b];
                cost[b - 1] = sw ? c1 : c0;
                cost[b] = sw ? c0 : c1;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const double v0 = sx[b - 1][d], v1 = sx[b][d];
                    sx[b - 1][d] = sw ? v1 : v0;
                    sx[b][d] = sw ? v0 : v1;
                }
            }
        }
    };
    sort();
    for (int it = 0; it < 1000; ++it) { // max_iters(1_000), mle.rs:98
        double mean = 0.0, sd = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) mean += cost[i];
        mean /= (double)V;
#pragma unroll
        for (int i = 0; i < V; ++i) sd = fma(cost[i] - mean, cost[i] - mean, sd);
        sd = sqrt(sd / ((double)V - 1.0));
        if (sd < PG_EPS) break;
        double x0[D], xr[D], xt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double c = sx[0][d];
#pragma unroll
            for (int i = 1; i < V - 1; ++i) c += sx[i][d];
            x0[d] = c * (1.0 / ((double)V - 1.0));
            xr[d] = x0[d] + (x0[d] - sx[V - 1][d]);
        }
        const double cr = mle_cost<P>(xr, S);
        if (cr < cost[V - 2] && cr >= cost[0]) { // reflection
#pragma unroll
            for (int d = 0; d < D; ++d) sx[V - 1][d] = xr[d];
            cost[V - 1] = cr;
        } else if (cr < cost[0]) { // expansion
#pragma unroll
            for (int d = 0; d < D; ++d) xt[d] = x0[d] + (xr[d] - x0[d]) * 2.0;
            const double ce = mle_cost<P>(xt, S);
            const bool e = ce < cr;
#pragma unroll
            for (int d = 0; d < D; ++d) sx[V - 1][d] = e ? xt[d] : xr[d];
            cost[V - 1] = e ? ce : cr;
        } else { // contraction, else shrink
#pragma unroll
            for (int d = 0; d < D; ++d) xt[d] = x0[d] + (sx[V - 1][d] - x0[d]) * 0.5;
            const double cc = mle_cost<P>(xt, S);
            if (cc < cost[V - 1]) {
#pragma unroll
                for (int d = 0; d < D; ++d) sx[V - 1][d] = xt[d];
                cost[V - 1] = cc;
            } else {
#pragma unroll
                for (int i = 1; i < V; ++i) {
#pragma unroll
                    for (int d = 0; d < D; ++d) sx[i][d] = sx[0][d] + (sx[i][d] - sx[0][d]) * 0.5;
                    cost[i] = mle_cost<P>(sx[i], S);
                }
            }
        }
        sort();
    }
    // ---- closing arithmetic (mle.rs:112-113, :120-150, :166-186) ------------------------------------------------------------
    const double ve = mle_bound(sx[0][0]);
    const double b = sx[0][P];
    // [(X'X)^-1]_(last,last) by elimination of the leading P - 1 columns (Schur complement); a zero pivot = the reference's
    // "Non-invertible x_matrix"
    double M[P][P];
#pragma unroll
    for (int r = 0; r < P; ++r)
#pragma unroll
        for (int c = 0; c < P; ++c) M[r][c] = S.A[r][c];
    bool singular = false;
#pragma unroll
    for (int q = 0; q < P - 1; ++q) {
        if (M[q][q] == 0.0) singular = true;
        const double inv = 1.0 / M[q][q];
#pragma unroll
        for (int r = q + 1; r < P; ++r) {
            const double f = M[r][q] * inv;
#pragma unroll
            for (int c = q + 1; c < P; ++c) M[r][c] = fma(-f, M[q][c], M[r][c]);
        }
    }
    singular = singular || !(M[P - 1][P - 1] > 1e-12 * S.A[P - 1][P - 1]);
    double bo = NAN, vo = NAN, po = NAN;
    if (!singular) {
        const double vb = ve / M[P - 1][P - 1];
        const double t = b / vb;
        bo = b; vo = vb;
        if (isinf(t)) po = 0.0;
        else if (isnan(t)) po = 1.0;
        else po = pg_t_two_sided_p(fabs(t), H.tdf, tcoef, H.ntcoef);
    }
    beta[cell] = bo; var[cell] = vo; pval[cell] = po;
}

// ---- the same solver for 5 .. 10 design columns (m = 3 .. 8): the simplex -- P + 2 vertices of P + 1 numbers -- no longer fits the
// register file, so the vertices and their costs live in LDS (lane-interleaved: conflict-free), and instead of moving vertices the
// sort permutes a rank -> vertex table held in one 64-bit register (4 bits per rank).  Every sum runs in rank order, i.e. the
// arithmetic is that of the register kernel / of the oracle's restatement on physically sorted vertices.
__device__ __forceinline__ int mle_nib(unsigned long long o, int r) { return (int)((o >> (4 * r)) & 15ull); }
__device__ __forceinline__ unsigned long long mle_swap(unsigned long long o, int r) { // ranks r - 1 and r trade places
    const unsigned long long a = (o >> (4 * (r - 1))) & 15ull, b = (o >> (4 * r)) & 15ull;
    o &= ~(0xffull << (4 * (r - 1)));
    return o | (b << (4 * (r - 1))) | (a << (4 * r));
}

template <int P>
__global__ __launch_bounds__(64) void k_mle_nm_lds(const double *__restrict__ sums, const double *__restrict__ gg,
                                                   const double *__restrict__ tcoef, int64_t p, const MleSharedBig *__restrict__ Hd,
                                                   double *__restrict__ beta, double *__restrict__ var, double *__restrict__ pval) {
    constexpr int D = P + 1, V = D + 1, Z1 = P - 1;
    extern __shared__ double mle_lds[];
    const int lane = threadIdx.x;
    auto SX = [&](int v, int d) -> double & { return mle_lds[(size_t)(v * D + d) * 64 + lane]; };
    auto CO = [&](int v) -> double & { return mle_lds[(size_t)(V * D + v) * 64 + lane]; };
    const int K = Hd->k, n = Hd->n;
    const int64_t cell0 = (int64_t)blockIdx.x * 64 + lane;
    const bool live = cell0 < p * K;
    const int64_t cell = live ? cell0 : p * K - 1; // (idle lanes shadow the last cell: no divergent exits around the LDS traffic)
    const int64_t l = cell / K;
    const int j = (int)(cell - l * K);
    const int ncol = Hd->m1 + K;
    double zg[Z1], xyz[Z1];
#pragma unroll
    for (int r = 0; r < Z1; ++r) { zg[r] = sums[l * ncol + r]; xyz[r] = Hd->zty[r * 4 + j]; }
    const double ggv = gg[l], xyg = sums[l * ncol + Hd->m1 + j], yy = Hd->yty[j];
    const double *__restrict__ ztz = Hd->ztz;

    auto cost_of = [&](const double (&par)[D]) { // mle_cost<P> with A = [[Z'Z, Z'g], [g'Z, g'g]], same order of operations
        const double sigma2 = mle_bound(par[0]);
        double q = 0.0, lsum = 0.0;
#pragma unroll
        for (int r = 0; r < Z1; ++r) {
            double ab = 0.0;
#pragma unroll
            for (int c = 0; c < Z1; ++c) ab = fma(ztz[r * Z1 + c], par[1 + c], ab);
            ab = fma(zg[r], par[P], ab);
            q = fma(par[1 + r], ab, q);
            lsum = fma(par[1 + r], xyz[r], lsum);
        }
        {
            double ab = 0.0;
#pragma unroll
            for (int c = 0; c < Z1; ++c) ab = fma(zg[c], par[1 + c], ab);
            ab = fma(ggv, par[P], ab);
            q = fma(par[P], ab, q);
            lsum = fma(par[P], xyg, lsum);
        }
        double ss = yy - 2.0 * lsum + q;
        ss = ss < 0.0 ? 0.0 : ss;
        return ((double)n / 2.00) * log(2.00 * 3.14159265358979323846 * sigma2) + (1.00 / sigma2) * ss;
    };
    auto load_vertex = [&](int v, double (&x)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = SX(v, d);
    };
    auto store_vertex = [&](int v, const double (&x)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) SX(v, d) = x[d];
    };
    // ---- start simplex (helpers.rs:132-146): ones, 1.5 on the diagonal --------------------------------------------------------
    unsigned long long ord = 0;
    for (int i = 0; i < V; ++i) {
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = (i == d) ? 1.5 : 1.0;
        store_vertex(i, x);
        CO(i) = cost_of(x);
        ord |= (unsigned long long)i << (4 * i);
    }
    // stable insertion sort of the ranks by cost; `from`: the ranks below it are already in order
    auto sort_ranks = [&](int from) {
        for (int a = from; a < V; ++a)
            for (int b = a; b >= 1; --b) {
                const bool sw = CO(mle_nib(ord, b - 1)) > CO(mle_nib(ord, b));
                ord = sw ? mle_swap(ord, b) : ord;
            }
    };
    sort_ranks(1);
    for (int it = 0; it < 1000; ++it) { // max_iters(1_000), mle.rs:98
        double mean = 0.0, sd = 0.0;
        for (int i = 0; i < V; ++i) mean += CO(mle_nib(ord, i));
        mean /= (double)V;
        for (int i = 0; i < V; ++i) { const double c = CO(mle_nib(ord, i)); sd = fma(c - mean, c - mean, sd); }
        sd = sqrt(sd / ((double)V - 1.0));
        if (__all(sd < PG_EPS)) break;          // (a lane that has converged keeps stepping with its wave: its simplex has collapsed
        const bool done = sd < PG_EPS;          //  to cost differences below EPSILON, but it must not move any more: guarded below)
        double x0[D], xr[D], xt[D], xw[D];
        const int vw = mle_nib(ord, V - 1), vb = mle_nib(ord, 0);
        load_vertex(vw, xw);
        {
            double x[D];
            load_vertex(vb, x0);
            for (int i = 1; i < V - 1; ++i) {
                load_vertex(mle_nib(ord, i), x);
#pragma unroll
                for (int d = 0; d < D; ++d) x0[d] += x[d];
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            x0[d] = x0[d] * (1.0 / ((double)V - 1.0));
            xr[d] = x0[d] + (x0[d] - xw[d]);
        }
        const double cr = cost_of(xr);
        const double c_best = CO(vb), c_second = CO(mle_nib(ord, V - 2)), c_worst = CO(vw);
        if (done) continue;
        bool shrink = false;
        if (cr < c_second && cr >= c_best) { // reflection
            store_vertex(vw, xr);
            CO(vw) = cr;
        } else if (cr < c_best) { // expansion
#pragma unroll
            for (int d = 0; d < D; ++d) xt[d] = x0[d] + (xr[d] - x0[d]) * 2.0;
            const double ce = cost_of(xt);
            if (ce < cr) { store_vertex(vw, xt); CO(vw) = ce; }
            else { store_vertex(vw, xr); CO(vw) = cr; }
        } else { // contraction, else shrink
#pragma unroll
            for (int d = 0; d < D; ++d) xt[d] = x0[d] + (xw[d] - x0[d]) * 0.5;
            const double cc = cost_of(xt);
            if (cc < c_worst) { store_vertex(vw, xt); CO(vw) = cc; }
            else shrink = true;
        }
        if (shrink) {
            double xb[D], x[D];
            load_vertex(vb, xb);
            for (int i = 1; i < V; ++i) {
                const int v = mle_nib(ord, i);
                load_vertex(v, x);
#pragma unroll
                for (int d = 0; d < D; ++d) x[d] = xb[d] + (x[d] - xb[d]) * 0.5;
                store_vertex(v, x);
                CO(v) = cost_of(x);
            }
            sort_ranks(1);
        } else {
            sort_ranks(V - 1); // only the replaced vertex is out of place
        }
    }
    // ---- closing arithmetic (mle.rs:112-113, :120-150, :166-186) ------------------------------------------------------------
    double xbest[D];
    load_vertex(mle_nib(ord, 0), xbest);
    const double ve = mle_bound(xbest[0]);
    const double b = xbest[P];
    // [(X'X)^-1]_(last,last) by elimination of the leading P - 1 columns; the matrix takes the simplex' place in LDS
    __builtin_amdgcn_wave_barrier();
    auto M = [&](int r, int c) -> double & { return mle_lds[(size_t)(r * P + c) * 64 + lane]; };
    for (int r = 0; r < Z1; ++r) {
        for (int c = 0; c < Z1; ++c) M(r, c) = ztz[r * Z1 + c];
        M(r, P - 1) = zg[r < Z1 ? r : 0];
        M(P - 1, r) = zg[r < Z1 ? r : 0];
    }
    M(P - 1, P - 1) = ggv;
    bool singular = false;
    for (int q = 0; q < P - 1; ++q) {
        const double piv = M(q, q);
        if (piv == 0.0) singular = true;
        const double inv = 1.0 / piv;
        for (int r = q + 1; r < P; ++r) {
            const double f = M(r, q) * inv;
            for (int c = q + 1; c < P; ++c) M(r, c) = fma(-f, M(q, c), M(r, c));
        }
    }
    const double schur = M(P - 1, P - 1);
    singular = singular || !(schur > 1e-12 * ggv);
    double bo = NAN, vo = NAN, po = NAN;
    if (!singular) {
        const double vb2 = ve / schur;
        const double t = b / vb2;
        bo = b; vo = vb2;
        if (isinf(t)) po = 0.0;
        else if (isnan(t)) po = 1.0;
        else po = pg_t_two_sided_p(fabs(t), Hd->tdf, tcoef, Hd->ntcoef);
    }
    if (live) { beta[cell] = bo; var[cell] = vo; pval[cell] = po; }
}

} // namespace

extern "C" int pg_mle_kinship_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y, int k,
                                  double var_explained, int force_m, int *m_out, double *K_out, double *beta_dev, double *var_dev,
                                  double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G_dev && Y && beta_dev && var_dev && pval_dev && p > 0 && n >= 3 && k >= 1 && k <= 4, "mle_kinship: bad arguments (1 <= k <= 4)");
    PG_CHECK(ctx, ld >= n && (ld % 2) == 0, "mle_kinship: ld (%lld) must be even and >= n (%d)", (long long)ld, n);
    for (int i = 0; i < n * k; ++i)
        PG_CHECK(ctx, !std::isnan(Y[i]), "mle_kinship: phenotype matrix contains NaN (the reference propagates it into every fit, mle.rs:345-360)");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    // ---- kinship, eigen rule (mle.rs:317-343 = gwas/ols.rs:291-315): the very code of ols_iter_with_kinship ----------------
    double *S = nullptr;
    PG_HIP(ctx, hipMalloc((void **)&S, sizeof(double) * n * n));
    std::vector<double> Kh((size_t)n * n), ev(n);
    int m = 0;
    int rc = pg_set_phenotypes(ctx, 0, nullptr, 0);
    if (!rc) rc = pg_kinship_partial_dev(ctx, G_dev, p, n, ld, S);
    if (!rc && (hipMemcpyAsync(Kh.data(), S, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = pg_fail(ctx, PG_ERR_HIP, "mle_kinship: D2H failed");
    (void)hipFree(S);
    if (rc) return rc;
    for (auto &x : Kh) x = x / (double)p;
    std::vector<double> V;
    if (force_m >= 0) m = force_m;
    else {
        if (pg_sym_eig(Kh.data(), n, ev.data(), nullptr, false) != 0) return pg_fail(ctx, PG_ERR_INVALID, "mle_kinship: eigen-decomposition did not converge");
        m = pg_host_n_eigenvecs(ev.data(), n, var_explained);
    }
    if (m_out) *m_out = m;
    if (K_out) std::memcpy(K_out, Kh.data(), sizeof(double) * n * n);
    if (m + 2 > MLE_MAXP_LDS)
        return pg_fail(ctx, PG_ERR_UNSUPPORTED, "mle_kinship: n_eigenvecs = %d; the simplex kernels carry at most %d design columns", m, MLE_MAXP_LDS);
    if (m + 2 >= n) return pg_fail(ctx, PG_ERR_UNSUPPORTED, "mle_kinship: no residual degrees of freedom");
    if (m > 0) {
        V.resize((size_t)n * m);
        if (pg_sym_eig_top(Kh.data(), n, m, ev.data(), V.data()) != 0) return pg_fail(ctx, PG_ERR_INVALID, "mle_kinship: eigen-decomposition did not converge");
    }
    // ---- one pass over G: Z'g, Y'g, g'g ------------------------------------------------------------------------------------------
    const int m1 = m + 1, ncol = m1 + k;
    std::vector<double> Z((size_t)n * ncol);
    MleShared H;
    MleSharedBig HB;
    std::memset(&H, 0, sizeof H);
    std::memset(&HB, 0, sizeof HB);
    // POOLGEN_MLE_LDS=1 sends the small designs through the LDS kernel too: same arithmetic in the same order, so the results must be
    // bit-identical to the register kernel's (tests/test_gpu_mle.py) -- the check that pins the LDS kernel's bookkeeping
    const bool big = m + 2 > MLE_MAXP || std::getenv("POOLGEN_MLE_LDS") != nullptr;
    for (int i = 0; i < n; ++i) {
        Z[(size_t)i * ncol] = 1.0;
        for (int a = 0; a < m; ++a) Z[(size_t)i * ncol + 1 + a] = V[(size_t)i * m + a];
        for (int j = 0; j < k; ++j) Z[(size_t)i * ncol + m1 + j] = Y[(size_t)i * k + j];
    }
    for (int a = 0; a < m1; ++a) {
        for (int b = 0; b < m1; ++b) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += Z[(size_t)i * ncol + a] * Z[(size_t)i * ncol + b];
            if (big) HB.ztz[a * m1 + b] = s; else H.ztz[a * m1 + b] = s;
        }
        for (int j = 0; j < k; ++j) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += Z[(size_t)i * ncol + a] * Y[(size_t)i * k + j];
            if (big) HB.zty[a * 4 + j] = s; else H.zty[a * 4 + j] = s;
        }
    }
    for (int j = 0; j < k; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += Y[(size_t)i * k + j] * Y[(size_t)i * k + j];
        H.yty[j] = s;
        HB.yty[j] = s;
    }
    H.n = n; H.m1 = m1; H.k = k;
    HB.n = n; HB.m1 = m1; HB.k = k;
    double *sums = nullptr, *gg = nullptr;
    if (hipMalloc((void **)&sums, sizeof(double) * (size_t)p * ncol) != hipSuccess || hipMalloc((void **)&gg, sizeof(double) * (size_t)p) != hipSuccess) {
        (void)hipFree(sums);
        return pg_fail(ctx, PG_ERR_HIP, "mle_kinship: out of device memory");
    }
    rc = pg_gp_beta_cols(ctx, G_dev, p, n, ld, Z.data(), ncol, sums, 0, gg);
    // ---- the fits -------------------------------------------------------------------------------------------------------------
    if (!rc) {
        const int df = n - 1;
        if (ctx->tcoef_df != df || !ctx->tcoef_dev) {
            std::vector<double> tc = pg_tdist_coef(df);
            if (ctx->tcoef_dev) (void)hipFree(ctx->tcoef_dev);
            ctx->tcoef_dev = nullptr;
            if (hipMalloc((void **)&ctx->tcoef_dev, sizeof(double) * (tc.size() + 1)) != hipSuccess ||
                (!tc.empty() && hipMemcpy(ctx->tcoef_dev, tc.data(), sizeof(double) * tc.size(), hipMemcpyHostToDevice) != hipSuccess))
                rc = pg_fail(ctx, PG_ERR_HIP, "mle_kinship: t-coefficient upload failed");
            ctx->tcoef_df = df;
            ctx->tcoef_len = (int)tc.size();
        }
        H.tdf = ctx->tcoef_df; H.ntcoef = ctx->tcoef_len;
        HB.tdf = ctx->tcoef_df; HB.ntcoef = ctx->tcoef_len;
    }
    if (!rc) {
        const int64_t cells = p * k;
        const unsigned grid = (unsigned)((cells + 63) / 64);
        MleSharedBig *HBd = nullptr;
        if (big) {
            if (hipMalloc((void **)&HBd, sizeof HB) != hipSuccess || hipMemcpyAsync(HBd, &HB, sizeof HB, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = pg_fail(ctx, PG_ERR_HIP, "mle_kinship: upload of the shared statistics failed");
        }
#define PG_MLE_LDS(PV)                                                                                                              \
    case PV: {                                                                                                                     \
        const size_t lds = sizeof(double) * 64 * ((size_t)(PV + 2) * (PV + 1) + (PV + 2));                                        \
        if (hipFuncSetAttribute((const void *)k_mle_nm_lds<PV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            rc = pg_fail(ctx, PG_ERR_HIP, "mle_kinship: LDS attribute");                                                           \
        else hipLaunchKernelGGL(k_mle_nm_lds<PV>, dim3(grid), dim3(64), lds, ctx->stream, sums, gg, ctx->tcoef_dev, p, HBd, beta_dev, var_dev, pval_dev); \
    } break;
        if (!rc && big) switch (m + 2) {
        PG_MLE_LDS(2) PG_MLE_LDS(3) PG_MLE_LDS(4) PG_MLE_LDS(5) PG_MLE_LDS(6) PG_MLE_LDS(7) PG_MLE_LDS(8) PG_MLE_LDS(9) PG_MLE_LDS(10)
        }
        else if (!rc) switch (m + 2) {
        case 2: hipLaunchKernelGGL(k_mle_nm<2>, dim3(grid), dim3(64), 0, ctx->stream, sums, gg, ctx->tcoef_dev, p, H, beta_dev, var_dev, pval_dev); break;
        case 3: hipLaunchKernelGGL(k_mle_nm<3>, dim3(grid), dim3(64), 0, ctx->stream, sums, gg, ctx->tcoef_dev, p, H, beta_dev, var_dev, pval_dev); break;
        default: hipLaunchKernelGGL(k_mle_nm<4>, dim3(grid), dim3(64), 0, ctx->stream, sums, gg, ctx->tcoef_dev, p, H, beta_dev, var_dev, pval_dev); break;
        }
#undef PG_MLE_LDS
        if (!rc && (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = pg_fail(ctx, PG_ERR_HIP, "mle_kinship: simplex kernel failed");
        if (HBd) (void)hipFree(HBd);
    }
    (void)hipFree(sums);
    (void)hipFree(gg);
    return rc;
}
