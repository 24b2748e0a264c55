// pg_gp.hip -- ridge-like penalised genomic prediction (gp/penalise.rs:133-159, :248-669).
//
// Reference flow (alpha = 0, iterative = false): for r repetitions x nfolds folds: b = gp::ols on the
// training pools; for every lambda of the path: b_lambda = expand_and_contract(b, b, alpha, lambda)
// (:248-357) and error_index on the validation pools (:359-426); per repetition the lambda with the
// smallest mean error over folds; the mode over repetitions; finally expand_and_contract of the
// all-rows fit.  The reference draws folds from an unseeded thread_rng (:452-453); here the fold of
// every training row in every repetition is an explicit argument, so results are reproducible.
//
// GPU decomposition per (repetition, fold):
//   1. pg_gp_ols_dev              b (1+p) x k        one streaming pass over G (k_gp_beta)
//   2. k_gp_norm_max              max of the penalty norm over the p slopes
//   3. k_gp_path_sums             for all L lambdas at once: the four redistribution masses of
//                                 expand_and_contract (one pass over b)
//   4. k_gp_blambda               B[l][i] = expand_and_contract(b)[l] for lambda_i   (p x L)
//   5. k_gp_predict (+ reduce)    yhat[pool][lambda] = b0 + sum_l G[l][pool] B[l][lambda]: the second
//                                 streaming pass over G, all lambdas at once
//   6. host: error_index on the validation pools (n_val x L numbers)
// All L lambdas share the two passes over G; the reference does 1 + L passes per fold.
#include "pg_common.h"
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <thread>
#include <vector>

namespace {

constexpr int GP_LMAX = 16; // lambdas per path (the reference uses 11: 0, 0.1, ..., 1)

struct PathParams {
    double alpha;
    double nmax;                  // max penalty norm (proxy = b itself, :262-283)
    double lambda[GP_LMAX];
    double sub_scale[GP_LMAX];    // subtracted_penalised / subtracted_depenalised (0 if nothing to expand)
    double add_scale[GP_LMAX];    // added_penalised / added_depenalised
    int L;
};

// The coefficients whose norms pick the penalised set (:262-283): the fit itself, or, for the *_with_iterative_proxy_norms
// models, the per-locus GWAS-like estimates of pg_gp_proxy_dev ((1+p) x k, row 0 unused).  b == nullptr: the fit itself.
struct Proxy {
    const double *b;
    int k, j;
};

__device__ __forceinline__ double gp_norm(double b, double alpha) { // :259-261
    return ((1.00 - alpha) * (b * b) / 1.00) + (alpha * fabs(b));
}

// max over the slopes (rows 1..p of column j) of the penalty norm; partial maxima per block
__global__ void k_gp_norm_max(const double *__restrict__ beta, int64_t p, int k, int j, int row0, double alpha,
                              double *__restrict__ part) {
    double m = 0.0;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < p; l += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, gp_norm(beta[(l + row0) * k + j], alpha));
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, sm[w]);
        part[blockIdx.x] = m;
    }
}

// all columns of a column-major matrix in one launch: blockIdx.y = column (its own values, or the proxy's column c % kx)
__global__ void k_gp_norm_max_cols(const double *__restrict__ cols, int64_t p, Proxy X, int kx, double alpha, const int *__restrict__ skip,
                                   double *__restrict__ part, int64_t part_stride) {
    const int c = blockIdx.y;
    if (skip[c]) return;
    const double *src = X.b ? X.b : cols + (size_t)c * p;
    const int k = X.b ? X.k : 1, j = X.b ? c % kx : 0, row0 = X.b ? 1 : 0;
    double m = 0.0;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < p; l += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, gp_norm(src[(l + row0) * k + j], alpha));
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, sm[w]);
        part[(size_t)c * part_stride + blockIdx.x] = m;
    }
}

// For every lambda_i: subtracted/added masses of the penalised set and the norm masses of the
// de-penalised set, split by the sign of b (:296-326).  part: [block][4][GP_LMAX].
// LP = the path length rounded up to even (a compile-time constant: the pass is bound by these 4 x LP conditional sums per element,
// 16 instead of 12 of them cost a third more)
template <int LP>
__device__ __forceinline__ void gp_path_sums_body(const double *__restrict__ beta, int64_t p, int k, int j, int row0,
                                                  const PathParams &P, Proxy X, double *__restrict__ part) {
    double sp[GP_LMAX], ap[GP_LMAX], sd[GP_LMAX], ad[GP_LMAX];
#pragma unroll
    for (int i = 0; i < GP_LMAX; ++i) { sp[i] = 0.0; ap[i] = 0.0; sd[i] = 0.0; ad[i] = 0.0; }
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < p; l += (int64_t)gridDim.x * blockDim.x) {
        const double b = beta[(l + row0) * k + j];
        const double nrm = gp_norm(b, P.alpha);
        const double nrx = X.b ? gp_norm(X.b[(l + 1) * X.k + X.j], P.alpha) : nrm;
        const double sc = nrx / P.nmax; // normed_proxy / normed_proxy_max (:282), a true division: max/max == 1
        const bool pos = b >= 0.0;
        const double pen_pos = pos ? (((b - nrm) < 0.0) ? b : nrm) : 0.0;       // :298-305
        const double pen_neg = pos ? 0.0 : (((b + nrm) > 0.0) ? fabs(b) : nrm); // :306-313
#pragma unroll
        for (int i = 0; i < LP; ++i) { // entries beyond P.L (lambda = 0) are never read: no guard, no branches
            const bool pen = sc < P.lambda[i];
            sp[i] += pen ? pen_pos : 0.0;
            ap[i] += pen ? pen_neg : 0.0;
            sd[i] += (!pen && pos) ? nrm : 0.0;
            ad[i] += (!pen && !pos) ? nrm : 0.0;
        }
    }
    __shared__ double sm[4][4 * GP_LMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < GP_LMAX; ++i) {
        double a = sp[i], b2 = ap[i], c = sd[i], d = ad[i];
        for (int off = 32; off >= 1; off >>= 1) {
            a += __shfl_xor(a, off); b2 += __shfl_xor(b2, off); c += __shfl_xor(c, off); d += __shfl_xor(d, off);
        }
        if (lane == 0) { sm[wave][i] = a; sm[wave][GP_LMAX + i] = b2; sm[wave][2 * GP_LMAX + i] = c; sm[wave][3 * GP_LMAX + i] = d; }
    }
    __syncthreads();
    if (threadIdx.x < 4 * GP_LMAX) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sm[w][threadIdx.x];
        part[(size_t)blockIdx.x * 4 * GP_LMAX + threadIdx.x] = s;
    }
}
template <int LP>
__global__ __launch_bounds__(256) void k_gp_path_sums(const double *__restrict__ beta, int64_t p, int k, int j, int row0, PathParams P,
                                                      Proxy X, double *__restrict__ part) {
    gp_path_sums_body<LP>(beta, p, k, j, row0, P, X, part);
}
// all columns of a column-major matrix in one launch: blockIdx.y = column, its norm maximum from nmax[] (device)
template <int LP>
__global__ __launch_bounds__(256) void k_gp_path_sums_cols(const double *__restrict__ cols, int64_t p, PathParams P, Proxy X, int kx,
                                                           const double *__restrict__ nmax, const int *__restrict__ skip,
                                                           double *__restrict__ part) {
    const int c = blockIdx.y;
    if (skip[c]) return;
    P.nmax = nmax[c];
    X.j = c % kx;
    gp_path_sums_body<LP>(cols + (size_t)c * p, p, 1, 0, 0, P, X, part + (size_t)c * gridDim.x * 4 * GP_LMAX);
}

// expand_and_contract of one coefficient for lambda_i (:296-352), given the global masses
__device__ __forceinline__ double gp_contract(double b, double bx, const PathParams &P, int i) {
    const double nrm = gp_norm(b, P.alpha);
    const double sc = gp_norm(bx, P.alpha) / P.nmax;
    if (sc < P.lambda[i]) { // penalised: contract by its own norm, not across zero
        if (b >= 0.0) return ((b - nrm) < 0.0) ? 0.0 : b - nrm;
        return ((b + nrm) > 0.0) ? 0.0 : b + nrm;
    }
    // de-penalised: receives its share of the contracted mass of its sign
    if (b >= 0.0) return b + P.sub_scale[i] * nrm;
    return b - P.add_scale[i] * nrm;
}

// B[l][i] for all lambdas (row stride GP_LMAX)
__global__ void k_gp_blambda(const double *__restrict__ beta, int64_t p, int k, int j, PathParams P, Proxy X,
                             double *__restrict__ B) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= p) return;
    const double b = beta[(l + 1) * k + j];
    const double bx = X.b ? X.b[(l + 1) * X.k + X.j] : b;
#pragma unroll
    for (int i = 0; i < GP_LMAX; ++i) B[l * GP_LMAX + i] = (i < P.L) ? gp_contract(b, bx, P, i) : 0.0;
}

// the all-rows fit's slopes, formed as columns of a batched coefficient pass (column-major), into the model's [1 + p] x k layout
__global__ void k_gp_cols_to_rows(const double *__restrict__ cols, int64_t p, int k, double *__restrict__ rows) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= p) return;
    for (int j = 0; j < k; ++j) rows[l * k + j] = cols[(size_t)j * p + l];
}

// single-lambda variant writing the penalised column back (final model, :653-662)
__global__ void k_gp_apply(double *__restrict__ beta, int64_t p, int k, int j, PathParams P, Proxy X, int i) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= p) return;
    const double b = beta[(l + 1) * k + j];
    beta[(l + 1) * k + j] = gp_contract(b, X.b ? X.b[(l + 1) * X.k + X.j] : b, P, i);
}

// yhat partials: thread = pool, block = slab of loci; B rows are wave-uniform (scalar loads)
__global__ __launch_bounds__(256) void k_gp_predict(const double *__restrict__ G, const double *__restrict__ B,
                                                    int64_t p, int n, int64_t ld, int64_t loci_per_block,
                                                    double *__restrict__ part) {
    const int pool = blockIdx.y * 256 + threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * loci_per_block;
    const int64_t l1 = min(p, l0 + loci_per_block);
    double acc[GP_LMAX];
#pragma unroll
    for (int i = 0; i < GP_LMAX; ++i) acc[i] = 0.0;
    const bool on = pool < n;
    const double *gp = G + (on ? pool : 0);
    int64_t l = l0;
    for (; l + 4 <= l1; l += 4) {
        double g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = gp[(l + u) * ld];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *bl = B + (l + u) * GP_LMAX;
#pragma unroll
            for (int i = 0; i < GP_LMAX; ++i) acc[i] = fma(g[u], bl[i], acc[i]);
        }
    }
    for (; l < l1; ++l) {
        const double g = gp[l * ld];
        const double *bl = B + l * GP_LMAX;
#pragma unroll
        for (int i = 0; i < GP_LMAX; ++i) acc[i] = fma(g, bl[i], acc[i]);
    }
    if (on) {
        double *o = part + ((size_t)blockIdx.x * n + pool) * GP_LMAX;
#pragma unroll
        for (int i = 0; i < GP_LMAX; ++i) o[i] = acc[i];
    }
}

// All folds of a repetition at once: bf (p x C) holds the slopes of every fold's training fit (column of pool i
// for this trait = colof[i], the fold that VALIDATES pool i -- every pool is validated by exactly one fold, so one
// pass over G serves all of them).  The contracted coefficient is formed on the fly from the fold's masses
// (the very arithmetic of gp_contract); partials and their reduction are laid out as in k_gp_predict.
struct FoldMasses { double nmax, sub_scale[GP_LMAX], add_scale[GP_LMAX]; };
// The contracted coefficient depends on (locus, fold, lambda) only, not on the pool: a block first forms them for a
// chunk of loci in LDS (chunk x folds x lambdas), then every thread (= pool) accumulates yhat from ITS fold's entries:
// per (locus, pool) one load of G, L LDS operands and L FMAs instead of the whole expand_and_contract arithmetic.
// LP = the path length rounded up to even, a compile-time constant: entries beyond L carry lambda = 0 and zero masses
// (harmless finite numbers nobody reads), so neither loop needs a guard -- guarded, every FMA became a branch with its
// own LDS round trip.
// Bound: the LDS return path -- L / 2 16-byte operand reads per (locus, pool): 96 bytes per element at 12 lambdas, 1.39 ms
// of LDS time per 2 M loci x 500 pools against 1.46 ms measured (0.685 of the HBM peak).  A matrix-core form (pools
// regrouped by fold into 16-pool tiles, two LDS reads per 4 loci x 16 pools x 16 lambdas) was built twice, parity green,
// and is slower: 0.55 with two 8-wave workgroups per CU and one chunk of rows ahead, 0.46 with one 16-wave workgroup and
// two chunks ahead -- the serial phases of a chunk (stage write, table, barrier, products) outlast its memory time
// (tools/experiments/).
template <int LP, bool GROUPED, bool ODD = false>
__global__ __launch_bounds__(512) void k_gp_predict_folds(const double *__restrict__ G, const double *__restrict__ bf,
                                                          int C, const int32_t *__restrict__ colof,
                                                          const FoldMasses *__restrict__ FM, PathParams P0, Proxy X,
                                                          int64_t p, int n, int64_t ld, int64_t loci_per_block,
                                                          int chunk, double *__restrict__ part, int groups) {
    extern __shared__ __attribute__((aligned(16))) double Bs[]; // [chunk][F][LS]
    constexpr int LS = LP + 2; // fold stride: 8 * LS bytes put the folds' 16-byte reads of one lambda pair on distinct banks
    // ODD: the path has LP - 1 values (the reference's 11): the last one is read alone (8 bytes) and the padding entry neither formed
    // nor read nor multiplied -- 88 instead of 96 bytes of LDS operands per (locus, pool) of a pass the LDS return path bounds
    constexpr int LN = ODD ? LP - 1 : LP;
    const int k = X.k, j = X.j;
    const int F = C / k;
    // 256 or 512 threads: one block spans up to 512 pools.  With fewer pools than that the block splits into `groups` of
    // blockDim / groups threads: group g takes the loci g, g + groups, ... of every chunk (a row of 100 pools would leave three
    // fifths of a 256-thread block without a pool, and the pass with a third of its loads in flight) and leaves its own partial.
    // (GROUPED is a template parameter: the run-time strides cost the plain form 50 registers and half its occupancy)
    const int gsz = GROUPED ? blockDim.x / groups : blockDim.x, grp = GROUPED ? threadIdx.x / gsz : 0;
    const int pool = GROUPED ? (int)threadIdx.x - grp * gsz : (int)(blockIdx.y * blockDim.x + threadIdx.x);
    const int64_t l0 = (int64_t)blockIdx.x * loci_per_block;
    const int64_t l1 = min(p, l0 + loci_per_block);
    double acc[LP];
#pragma unroll
    for (int i = 0; i < LP; ++i) acc[i] = 0.0;
    const bool inr = pool < n;
    const int c = inr ? colof[pool] : -1;
    const bool on = c >= 0;
    const int f = on ? c / k : 0;
    const double *gp = G + (inr ? pool : 0);
    // The table's inputs must not cost a memory round trip per chunk between the two barriers: the folds' masses sit in
    // LDS for the whole launch, and the first PF_NI items of a thread (all of them at the shipped shapes) get the next
    // chunk's coefficients while this chunk is accumulated.
    constexpr int PF_NI = 2, FMW = 2 * GP_LMAX + 1;
    double *masses = Bs + (size_t)chunk * F * LS; // [F][FMW]: { nmax, sub_scale[], add_scale[] } of this trait's columns
    for (int i = threadIdx.x; i < F * FMW; i += blockDim.x) {
        const int ff = i / FMW, w = i - ff * FMW;
        const FoldMasses *src = FM + (ff * k + j);
        masses[i] = w == 0 ? src->nmax : (w <= GP_LMAX ? src->sub_scale[w - 1] : src->add_scale[w - 1 - GP_LMAX]);
    }
    double bn[PF_NI], xn[PF_NI];
    auto fetch = [&](int64_t lc2) {
        const int m2 = (int)min((int64_t)chunk, l1 - lc2);
#pragma unroll
        for (int q = 0; q < PF_NI; ++q) {
            const int item = threadIdx.x + q * blockDim.x;
            bn[q] = 0.0; xn[q] = 0.0;
            if (item < m2 * F) {
                const int ff = item / m2, ll = item - ff * m2; // bf is column-major: consecutive threads, consecutive loci
                bn[q] = bf[(size_t)(ff * k + j) * p + lc2 + ll];
                if (X.b) xn[q] = X.b[(lc2 + ll + 1) * X.k + X.j];
            }
        }
    };
    if (l0 < l1) fetch(l0);
    for (int64_t lc = l0; lc < l1; lc += chunk) {
        const int m = (int)min((int64_t)chunk, l1 - lc);
        __syncthreads();
        auto table_row = [&](int item, double b, double bx) {
            const int ff = item / m, ll = item - ff * m;
            const double *fm = masses + ff * FMW;
            const double nrm = gp_norm(b, P0.alpha);
            const double sc = (X.b ? gp_norm(bx, P0.alpha) : nrm) / fm[0];
            const bool pos = b >= 0.0;
            const double pen = pos ? (((b - nrm) < 0.0) ? 0.0 : b - nrm) : (((b + nrm) > 0.0) ? 0.0 : b + nrm);
            double *o = Bs + (size_t)(ll * F + ff) * LS;
#pragma unroll
            for (int i = 0; i < LN; ++i) {
                const double dep = pos ? b + fm[1 + i] * nrm : b - fm[1 + GP_LMAX + i] * nrm;
                o[i] = (sc < P0.lambda[i]) ? pen : dep;
            }
        };
#pragma unroll
        for (int q = 0; q < PF_NI; ++q) {
            const int item = threadIdx.x + q * blockDim.x;
            if (item < m * F) table_row(item, bn[q], xn[q]);
        }
        for (int item = threadIdx.x + PF_NI * blockDim.x; item < m * F; item += blockDim.x) {
            const int ff = item / m, ll = item - ff * m;
            const int64_t l = lc + ll;
            table_row(item, bf[(size_t)(ff * k + j) * p + l], X.b ? X.b[(l + 1) * X.k + X.j] : 0.0);
        }
        __syncthreads();
        if (lc + chunk < l1) fetch(lc + chunk);
        if (on) {
            const double *bs = Bs + (size_t)f * LS;
            constexpr int U = 8; // loads of G in flight per thread
            // this group's loci of the chunk: grp, grp + groups, ...  (mg of them; ll counts them)
            // (uniform trip count: the loci every group has; the m % groups left-over loci go to the first groups below)
            const int mg = GROUPED ? m / groups : m;
            // GROUPED: uniform row base + one 32-bit per-thread offset (a per-thread 64-bit base costs the loop 50 registers)
            const double *gq = GROUPED ? G + lc * ld : gp + lc * ld;
            const uint32_t voff = GROUPED ? (uint32_t)(((int64_t)grp * ld + (inr ? pool : 0)) * 8) : 0u;
            const int64_t gstep = GROUPED ? (int64_t)groups * ld : ld;
            auto gload = [&](int i) -> double {
                return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(gq + i * gstep) + voff);
            };
            const double *bq = bs + (size_t)grp * F * LS;
            const int bstep = GROUPED ? groups * F * LS : F * LS;
            int ll = 0;
            for (; ll + U <= mg; ll += U) {
                double g[U];
#pragma unroll
                for (int u = 0; u < U; ++u) g[u] = gload(ll + u);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double *q = bq + (ll + u) * bstep;
#pragma unroll
                    for (int i = 0; i + 2 <= LN; i += 2) {
                        const double2 b2 = *reinterpret_cast<const double2 *>(q + i);
                        acc[i] = fma(g[u], b2.x, acc[i]);
                        acc[i + 1] = fma(g[u], b2.y, acc[i + 1]);
                    }
                    if constexpr (ODD) acc[LN - 1] = fma(g[u], q[LN - 1], acc[LN - 1]);
                }
            }
            for (; ll < mg; ++ll) {
                const double g = gload(ll);
                const double *q = bq + ll * bstep;
#pragma unroll
                for (int i = 0; i + 2 <= LN; i += 2) {
                    const double2 b2 = *reinterpret_cast<const double2 *>(q + i);
                    acc[i] = fma(g, b2.x, acc[i]);
                    acc[i + 1] = fma(g, b2.y, acc[i + 1]);
                }
                if constexpr (ODD) acc[LN - 1] = fma(g, q[LN - 1], acc[LN - 1]);
            }
            if (GROUPED && grp + groups * mg < m) { // one of the m % groups left-over loci
                const double g = gload(mg);
                const double *q = bq + mg * bstep;
#pragma unroll
                for (int i = 0; i + 2 <= LN; i += 2) {
                    const double2 b2 = *reinterpret_cast<const double2 *>(q + i);
                    acc[i] = fma(g, b2.x, acc[i]);
                    acc[i + 1] = fma(g, b2.y, acc[i + 1]);
                }
                if constexpr (ODD) acc[LN - 1] = fma(g, q[LN - 1], acc[LN - 1]);
            }
        }
    }
    if (inr) {
        double *o = part + (((size_t)blockIdx.x * (GROUPED ? groups : 1) + grp) * n + pool) * GP_LMAX;
#pragma unroll
        for (int i = 0; i < GP_LMAX; ++i) o[i] = i < LP ? acc[i] : 0.0;
    }
}

// 64 outputs x 8 groups of slabs per workgroup; the groups' sums are combined in group order (same result every run)
__global__ __launch_bounds__(512) void k_gp_predict_reduce(const double *__restrict__ part, int nblocks, int n,
                                                           double *__restrict__ out) {
    __shared__ double sm[8][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o; // pool * GP_LMAX + i
    double s = 0.0;
    if (idx < n * GP_LMAX)
        for (int b = g; b < nblocks; b += 8) s += part[(size_t)b * n * GP_LMAX + idx];
    sm[g][o] = s;
    __syncthreads();
    if (g == 0 && idx < n * GP_LMAX) {
        double t = sm[0][o];
        for (int q = 1; q < 8; ++q) t += sm[q][o];
        out[idx] = t;
    }
}

// yhat partials for a plain coefficient matrix: thread = pool, block = slab of loci, up to 8 traits
__global__ __launch_bounds__(256) void k_gp_predict_beta(const double *__restrict__ G, const double *__restrict__ beta,
                                                         int k, int64_t p, int n, int64_t ld, int64_t loci_per_block,
                                                         double *__restrict__ part) {
    const int pool = blockIdx.y * 256 + threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * loci_per_block;
    const int64_t l1 = min(p, l0 + loci_per_block);
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    const bool on = pool < n;
    const double *gp = G + (on ? pool : 0);
    for (int64_t l = l0; l < l1; ++l) {
        const double g = gp[l * ld];
        const double *bl = beta + (l + 1) * k; // row 0 of beta is the intercept
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < k) acc[j] = fma(g, bl[j], acc[j]);
    }
    if (on) {
        double *o = part + ((size_t)blockIdx.x * n + pool) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = acc[j];
    }
}
__global__ void k_gp_predict_beta_reduce(const double *__restrict__ part, int nblocks, int n, double *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x; // pool * 8 + j
    if (idx >= n * 8) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * n * 8 + idx];
    out[idx] = s;
}

// pearsons_correlation (gwas/correlation_test.rs:7-71) on two complete vectors, as error_index calls it
double host_pearson_r(const std::vector<double> &x, const std::vector<double> &y) {
    const int n = (int)x.size();
    double mx = 0, my = 0;
    for (int i = 0; i < n; ++i) { mx += x[i]; my += y[i]; }
    mx /= n; my /= n;
    double sxy = 0, sxx = 0, syy = 0;
    for (int i = 0; i < n; ++i) { const double dx = x[i] - mx, dy = y[i] - my; sxy += dx * dy; sxx += dx * dx; syy += dy * dy; }
    const double r = sxy / (std::sqrt(sxx) * std::sqrt(syy));
    if (std::isnan(r)) return NAN;
    const double sden = (1.0 - r * r) / ((double)n - 2.0);
    if (sden <= 0.0) return r;
    return std::round(r * 1e7) / 1e7; // sensible_round(r, 7)
}

struct RidgeWork {
    double *part = nullptr;   // block partials (max / path sums / predictions)
    double *B = nullptr;      // p x GP_LMAX
    double *yhat = nullptr;   // n x GP_LMAX
};

// steps 2-4 for trait j of `beta_dev`; returns the path parameters with the masses filled in
int ridge_path_params(pg_ctx *ctx, const double *beta_dev, int64_t p, int k, int j, double alpha,
                      const std::vector<double> &path, RidgeWork &W, PathParams &P, int row0 = 1, Proxy X = Proxy{nullptr, 0, 0}) {
    const int nb = 1024;
    std::vector<double> h((size_t)nb * 4 * GP_LMAX);
    if (X.b) hipLaunchKernelGGL(k_gp_norm_max, dim3(nb), dim3(256), 0, ctx->stream, X.b, p, X.k, X.j, 1, alpha, W.part);
    else hipLaunchKernelGGL(k_gp_norm_max, dim3(nb), dim3(256), 0, ctx->stream, beta_dev, p, k, j, row0, alpha, W.part);
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipMemcpyAsync(h.data(), W.part, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double mx = 0.0;
    for (int b = 0; b < nb; ++b) mx = std::max(mx, h[b]);
    std::memset(&P, 0, sizeof P);
    P.alpha = alpha;
    P.nmax = mx;
    P.L = (int)path.size();
    for (int i = 0; i < P.L; ++i) P.lambda[i] = path[i];
    switch ((P.L + 1) & ~1) {
#define PG_PATH_SUMS(LPV) case LPV: hipLaunchKernelGGL(k_gp_path_sums<LPV>, dim3(nb), dim3(256), 0, ctx->stream, beta_dev, p, k, j, row0, P, X, W.part); break;
        PG_PATH_SUMS(2) PG_PATH_SUMS(4) PG_PATH_SUMS(6) PG_PATH_SUMS(8) PG_PATH_SUMS(10) PG_PATH_SUMS(12) PG_PATH_SUMS(14)
        default: hipLaunchKernelGGL(k_gp_path_sums<GP_LMAX>, dim3(nb), dim3(256), 0, ctx->stream, beta_dev, p, k, j, row0, P, X, W.part); break;
#undef PG_PATH_SUMS
    }
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipMemcpyAsync(h.data(), W.part, sizeof(double) * nb * 4 * GP_LMAX, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < P.L; ++i) {
        double sp = 0, ap = 0, sd = 0, ad = 0;
        for (int b = 0; b < nb; ++b) {
            const double *q = &h[(size_t)b * 4 * GP_LMAX];
            sp += q[i]; ap += q[GP_LMAX + i]; sd += q[2 * GP_LMAX + i]; ad += q[3 * GP_LMAX + i];
        }
        // "absence of available slots" (:329-335)
        if ((sp > 0.0) & (sd == 0.0)) { ap -= sp; sp = 0.0; }
        else if ((ap > 0.0) & (ad == 0.0)) { sp -= ap; ap = 0.0; }
        // b += subtracted_penalised * (normed / subtracted_depenalised)  (:345-351); 0/0 never reaches a
        // coefficient because an empty de-penalised side has no coefficients to expand
        P.sub_scale[i] = (sd != 0.0) ? sp / sd : 0.0;
        P.add_scale[i] = (ad != 0.0) ? ap / ad : 0.0;
    }
    return PG_OK;
}


// per column: out[c][q] = max or sum over the blocks' partials, in block order (what the host loop of ridge_path_params does)
// one wave per output: lane t takes blocks t, t + 64, ...; the 64 lane sums are combined by a fixed butterfly
__global__ __launch_bounds__(256) void k_gp_reduce_parts(const double *__restrict__ part, int ncols, int nb, int64_t col_stride,
                                                         int elem_stride, int use, int is_max, double *__restrict__ out) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6); // c * use + q
    const int lane = threadIdx.x & 63;
    if (idx >= ncols * use) return;
    const int c = idx / use, q = idx - c * use;
    const double *src = part + (size_t)c * col_stride + q;
    double r = 0.0;
    for (int b = lane; b < nb; b += 64) r = is_max ? fmax(r, src[(size_t)b * elem_stride]) : r + src[(size_t)b * elem_stride];
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(r, off);
        r = is_max ? fmax(r, o) : r + o;
    }
    if (lane == 0) out[idx] = r;
}

// ridge_path_params for all columns of `cols_dev` (COLUMN-major, ncols x p: every column is one contiguous stream) at once: the columns' launches queue up behind
// each other and the host synchronises twice instead of 2 * ncols times.  skip[c] != 0: column not in use.
int ridge_path_params_cols(pg_ctx *ctx, const double *cols_dev, int64_t p, int ncols, int k, double alpha,
                           const std::vector<double> &path, RidgeWork &W, const double *proxy_dev, const std::vector<int> &skip,
                           std::vector<PathParams> &out) {
    // blocks per column: enough to fill the chip on a long column, few on a short one (every block's 64 partial sums are
    // cleared, written and reduced again: at p = 2e5 that overhead was most of the mass step)
    const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(32, p / 2048));
    const int width = 4 * GP_LMAX;
    double *red = W.part + (size_t)ncols * nb * width; // room reserved by the caller (sized for nb = 1024)
    std::vector<double> h((size_t)ncols * width);
    out.assign(ncols, PathParams{});
    PG_HIP(ctx, hipMemsetAsync(W.part, 0, sizeof(double) * (size_t)ncols * nb * width, ctx->stream)); // skipped columns reduce to 0
    // the maxima stay on the device for the path sums (and travel to the host with them, further down)
    double *nmax_dev = red + (size_t)ncols * width;
    int *skip_dev = reinterpret_cast<int *>(nmax_dev + ncols);
    PG_HIP(ctx, hipMemcpyAsync(skip_dev, skip.data(), sizeof(int) * ncols, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gp_norm_max_cols, dim3(nb, ncols), dim3(256), 0, ctx->stream, cols_dev, p, Proxy{proxy_dev, k, 0}, k, alpha, skip_dev,
                       W.part, (int64_t)nb * width); // every column in one launch (was one launch per column: 200 per config-4 run)
    hipLaunchKernelGGL(k_gp_reduce_parts, dim3((ncols + 3) / 4), dim3(256), 0, ctx->stream, W.part, ncols, nb, (int64_t)nb * width, 1,
                       1, 1, red);
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipMemcpyAsync(nmax_dev, red, sizeof(double) * ncols, hipMemcpyDeviceToDevice, ctx->stream));
    PathParams Pc;
    std::memset(&Pc, 0, sizeof Pc);
    Pc.alpha = alpha;
    Pc.L = (int)path.size();
    for (int i = 0; i < Pc.L; ++i) Pc.lambda[i] = path[i];
    switch ((Pc.L + 1) & ~1) {
#define PG_PATH_SUMS(LPV)                                                                                                   \
    case LPV:                                                                                                               \
        hipLaunchKernelGGL(k_gp_path_sums_cols<LPV>, dim3(nb, ncols), dim3(256), 0, ctx->stream, cols_dev, p, Pc, Proxy{proxy_dev, k, 0}, k, \
                           nmax_dev, skip_dev, W.part);                                                                     \
        break;
        PG_PATH_SUMS(2) PG_PATH_SUMS(4) PG_PATH_SUMS(6) PG_PATH_SUMS(8) PG_PATH_SUMS(10) PG_PATH_SUMS(12) PG_PATH_SUMS(14)
        default:
            hipLaunchKernelGGL(k_gp_path_sums_cols<GP_LMAX>, dim3(nb, ncols), dim3(256), 0, ctx->stream, cols_dev, p, Pc, Proxy{proxy_dev, k, 0}, k,
                               nmax_dev, skip_dev, W.part);
            break;
#undef PG_PATH_SUMS
    }
    hipLaunchKernelGGL(k_gp_reduce_parts, dim3((ncols * width + 3) / 4), dim3(256), 0, ctx->stream, W.part, ncols, nb,
                       (int64_t)nb * width, width, width, 0, red);
    PG_HIP(ctx, hipGetLastError());
    std::vector<double> hmax(ncols);
    PG_HIP(ctx, hipMemcpyAsync(h.data(), red, sizeof(double) * ncols * width, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(hmax.data(), nmax_dev, sizeof(double) * ncols, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `skip` (the caller's) has been consumed as well
    for (int c = 0; c < ncols; ++c) {
        PathParams &P = out[c];
        P = Pc;
        P.nmax = hmax[c];
        if (skip[c]) continue;
        const double *q = &h[(size_t)c * width];
        for (int i = 0; i < P.L; ++i) {
            double sp = q[i], ap = q[GP_LMAX + i], sd = q[2 * GP_LMAX + i], ad = q[3 * GP_LMAX + i];
            if ((sp > 0.0) & (sd == 0.0)) { ap -= sp; sp = 0.0; }      // "absence of available slots" (:329-335)
            else if ((ap > 0.0) & (ad == 0.0)) { sp -= ap; ap = 0.0; }
            P.sub_scale[i] = (sd != 0.0) ? sp / sd : 0.0;
            P.add_scale[i] = (ad != 0.0) ? ap / ad : 0.0;
        }
    }
    return PG_OK;
}

// The lambda path with k-fold cross-validation (:461-669) behind penalise_lasso_like / _ridge_like (alpha = 1 / 0, one
// path), penalise_glmnet (alpha < 0: the 2-D grid alpha x lambda over the same path values, :479-498) and the
// *_with_iterative_proxy_norms models (proxy != nullptr, :540-553, :655-657).
int penalised_path(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y, int k,
                   const int64_t *row_idx, int n_rows, const int32_t *fold_of, int n_reps, int n_folds, double alpha,
                   const double *proxy_dev, double lambda_step, double *beta_dev, double *alphas_out, double *lambdas_out,
                   double *perf_out, const double *xxt_host_or_null) {
    const int maxu = (int)std::llround(1.0 / lambda_step);
    const int L = maxu + 1;
    PG_CHECK(ctx, L <= GP_LMAX, "gp_ridge: at most %d lambdas on the path", GP_LMAX);
    std::vector<double> path(L);
    for (int i = 0; i < L; ++i) path[i] = (double)i / (double)maxu; // :470-476
    const int A = alpha >= 0.0 ? 1 : L;                              // :479-498
    auto alpha_at = [&](int a) { return alpha >= 0.0 ? alpha : path[a]; };
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_entry = std::chrono::steady_clock::now(); // (POOLGEN_GP_TIMING: where the wall goes outside the repetitions)

    // the full-data X X^T once; every training subset uses a principal sub-block
    std::vector<double> xxt((size_t)n * n);
    if (xxt_host_or_null) std::memcpy(xxt.data(), xxt_host_or_null, sizeof(double) * (size_t)n * n);
    else {
        if (ctx->S_n < n) {
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->S_dev) PG_HIP(ctx, hipFree(ctx->S_dev));
            ctx->S_dev = nullptr; ctx->S_n = 0;
            PG_HIP(ctx, hipMalloc((void **)&ctx->S_dev, sizeof(double) * n * n));
            ctx->S_n = n;
        }
        int rc = pg_gp_xxt_dev(ctx, G_dev, p, n, ld, ctx->S_dev);
        if (rc) return rc;
        PG_HIP(ctx, hipMemcpyAsync(xxt.data(), ctx->S_dev, sizeof(double) * n * n, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // scratch
    const int nblk = std::max(1, std::min<int>(ctx->cus * 4, (int)((p + 255) / 256)));
    const int64_t lpb = (p + nblk - 1) / nblk;
    const int nblk2 = (int)((p + lpb - 1) / lpb);
    RidgeWork W;
    const size_t part_doubles = std::max<size_t>((size_t)(n_folds * k + 1) * 1024 * 4 * GP_LMAX + (size_t)n_folds * k * 4 * GP_LMAX,
                                                 (size_t)nblk2 * n * GP_LMAX * 4); // (x 4: the prediction pass' locus groups at n <= 256)
    char *raw = nullptr;
    PG_HIP(ctx, hipMalloc((void **)&raw, sizeof(double) * (part_doubles + (size_t)p * GP_LMAX + (size_t)n * GP_LMAX)));
    W.part = reinterpret_cast<double *>(raw);
    W.B = W.part + part_doubles;
    W.yhat = W.B + (size_t)p * GP_LMAX;
    auto fail = [&](int rc) { (void)hipFree(raw); return rc; };

    // error indices (rep, fold, alpha, lambda, trait) as the reference's `performances` (:509)
    std::vector<double> perf((size_t)n_reps * n_folds * A * L * k, NAN), b0(k), yh((size_t)n * GP_LMAX);
    std::vector<int64_t> itr, iva;
    for (int i = 0; i < n_reps * n_rows; ++i)
        if (fold_of[i] < 0 || fold_of[i] > n_folds /* == n_folds: the left-over group of k_split (:444-448), never validated */) {
            ctx->err = "gp_ridge: fold id out of range"; return fail(PG_ERR_INVALID);
        }
    // error_index (:359-426) of trait j on the validation pools `iva`, for every lambda, from yhat (n x GP_LMAX)
    auto score = [&](int rep, int fold, int a, int j, double b0j, const std::vector<int64_t> &iva_) {
        const int nv = (int)iva_.size();
        std::vector<double> yt(nv), yp(nv);
        double mn = 0, mx = 0;
        for (int i = 0; i < nv; ++i) {
            yt[i] = Y[(size_t)iva_[i] * k + j];
            if (i == 0 || yt[i] < mn) mn = yt[i];
            if (i == 0 || yt[i] > mx) mx = yt[i];
        }
        for (int li = 0; li < L; ++li) {
            for (int i = 0; i < nv; ++i) yp[i] = b0j + yh[(size_t)iva_[i] * GP_LMAX + li];
            const double cor = host_pearson_r(yt, yp);
            double mae = 0, mse = 0;
            for (int i = 0; i < nv; ++i) { const double d = yt[i] - yp[i]; mae += std::fabs(d); mse += d * d; }
            mae /= (mx - mn);
            mse /= ((mx - mn) * (mx - mn));
            const double rmse = std::sqrt(mse) / (mx - mn);
            perf[((((size_t)rep * n_folds + fold) * A + a) * L + li) * k + j] = ((1.0 - std::fabs(cor)) + mae + mse + rmse) / 4.0;
        }
    };
    // Every pool is validated by exactly ONE fold of a repetition, so all folds share two passes over G: one that
    // forms the slopes of every fold's training fit (n_folds * k coefficient columns), one (per alpha) that predicts
    // every pool with the coefficients of the fold that holds it out.  (Fallback below: one pair of passes per fold.)
    const int C = n_folds * k;
    const bool fused = C <= PG_MAX_SWEEP_COLS && !std::getenv("POOLGEN_RIDGE_PER_FOLD");
    double *bf = nullptr;       // C x p (column-major) slopes of the folds' fits
    FoldMasses *fm_dev = nullptr;
    int32_t *colof_dev = nullptr;
    // The slopes of ALL repetitions' folds (and of the all-rows fit) are columns G Z of the same matrix: formed CP at a time,
    // whatever repetition they belong to, they take ceil((n_reps C + k) / CP) passes over G instead of n_reps + 1 (config 4:
    // 101 columns, 7 passes instead of 11).  Needs the n_reps C + k columns resident (config 4: 4 GB of the 288); otherwise,
    // or with POOLGEN_RIDGE_PER_REP=1, one pass per repetition as before.
    constexpr int CP = 16;      // columns per coefficient pass: what the sweep kernel's products mode carries at 500 pools
    const size_t ncols_all = (size_t)n_reps * C + k;
    bool batched = fused && n_reps > 1 && !std::getenv("POOLGEN_RIDGE_PER_REP");
    if (batched) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || sizeof(double) * (size_t)p * ncols_all > fr / 2) batched = false;
    }
    const auto t_alloc0 = std::chrono::steady_clock::now();
    if (fused) {
        if (hipMalloc((void **)&bf, sizeof(double) * (size_t)p * (batched ? ncols_all : (size_t)C)) != hipSuccess ||
            hipMalloc((void **)&fm_dev, sizeof(FoldMasses) * C) != hipSuccess ||
            hipMalloc((void **)&colof_dev, sizeof(int32_t) * n) != hipSuccess) {
            (void)hipFree(bf); (void)hipFree(fm_dev); (void)hipFree(colof_dev);
            return fail(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: out of device memory"));
        }
    }
    const double t_alloc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_alloc0).count();
    const double t_before = std::chrono::duration<double>(t_alloc0 - t_entry).count(); // X X^T, its copy to the host, the work buffers
    auto fail2 = [&](int rc) { (void)hipFree(bf); (void)hipFree(fm_dev); (void)hipFree(colof_dev); return fail(rc); };
    // POOLGEN_GP_TIMING=1: host-side phase times of the repetitions on stderr
    const bool timing = std::getenv("POOLGEN_GP_TIMING") != nullptr;
    double t_solve = 0, t_beta = 0, t_params = 0, t_predict = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // The folds' host solves (pinv(X X^T) y on every training block) need nothing from the GPU but X X^T: a background
    // thread works through the repetitions ahead of the device passes (one worker thread per fold inside).
    struct RepSolve {
        std::vector<std::vector<int64_t>> tr, va;
        std::vector<double> Z, b0c;
        bool bad = false;
    };
    std::vector<RepSolve> solves(fused ? n_reps : 0);
    std::vector<std::promise<void>> ready(fused ? n_reps + 1 : 0); // (the last one: the all-rows fit)
    std::vector<std::shared_future<void>> readyf;
    for (auto &pr : ready) readyf.push_back(pr.get_future().share());
    std::vector<double> Zall, b0all(k, 0.0); // the all-rows fit: pinv(X X^T) y scattered over the pools, its intercepts
    bool all_bad = false;
    std::thread solver;
    if (fused)
        solver = std::thread([&] {
            for (int rep = 0; rep < n_reps; ++rep) {
                RepSolve &R = solves[rep];
                R.tr.assign(n_folds, {}); R.va.assign(n_folds, {});
                for (int i = 0; i < n_rows; ++i) {
                    const int f = fold_of[(size_t)rep * n_rows + i];
                    for (int g = 0; g < n_folds; ++g) (g == f ? R.va[g] : R.tr[g]).push_back(row_idx[i]);
                }
                R.Z.assign((size_t)n * C, 0.0); R.b0c.assign(C, 0.0);
                std::vector<int> badf(n_folds, 0);
                std::vector<std::thread> th;
                for (int f = 0; f < n_folds; ++f) {
                    if (R.va[f].empty() || R.tr[f].empty()) continue; // an empty fold leaves NaN, as an empty slice would
                    th.emplace_back([&, f] {
                        const int r = (int)R.tr[f].size();
                        std::vector<double> V((size_t)r * k);
                        if (pg_gp_subset_solve(xxt.data(), n, Y, k, R.tr[f].data(), r, V.data()) != 0) { badf[f] = 1; return; }
                        for (int a2 = 0; a2 < r; ++a2)
                            for (int j = 0; j < k; ++j) {
                                R.Z[(size_t)R.tr[f][a2] * C + f * k + j] = V[(size_t)a2 * k + j];
                                R.b0c[f * k + j] += V[(size_t)a2 * k + j];
                            }
                    });
                }
                for (auto &x : th) x.join();
                for (int f = 0; f < n_folds; ++f) R.bad = R.bad || badf[f];
                ready[rep].set_value();
            }
            if (batched) { // the all-rows fit rides in the last batched pass (gp/ols.rs:47-72 on `row_idx`)
                std::vector<double> V((size_t)n_rows * k);
                Zall.assign((size_t)n * k, 0.0);
                if (pg_gp_subset_solve(xxt.data(), n, Y, k, row_idx, n_rows, V.data()) != 0) all_bad = true;
                else
                    for (int a2 = 0; a2 < n_rows; ++a2)
                        for (int j = 0; j < k; ++j) {
                            Zall[(size_t)row_idx[a2] * k + j] = V[(size_t)a2 * k + j];
                            b0all[j] += V[(size_t)a2 * k + j]; // intercept column of X is all ones
                        }
            }
            ready[n_reps].set_value();
        });
    struct Joiner { // every exit path waits for the solver before its captures go away
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{solver};
    // One prediction pass is kept OUT while the host goes on (the next repetition's coefficient passes, the next masses, the
    // scores of the pass before): its predictions land in yh_in[buf], the small host arrays its copies read stay alive in turn.
    struct Pending { bool on; int rep, a, j, buf; };
    Pending pend{false, 0, 0, 0, 0};
    std::vector<double> yh_in[2] = {std::vector<double>((size_t)n * GP_LMAX), std::vector<double>((size_t)n * GP_LMAX)};
    std::vector<int32_t> colof_h[2];
    std::vector<FoldMasses> fm_h[2];
    int fm_turn = 0;
    auto score_pending = [&](const Pending &q) {
        const RepSolve &R = solves[q.rep];
        yh.swap(yh_in[q.buf]); // (score reads yh)
        for (int f = 0; f < n_folds; ++f)
            if (!R.va[f].empty() && !R.tr[f].empty()) score(q.rep, f, q.a, q.j, R.b0c[f * k + q.j], R.va[f]);
        yh.swap(yh_in[q.buf]);
    };
    size_t formed = 0; // batched: columns [0, formed) of the global numbering (repetition-major, then the all-rows fit) are in bf
    auto form_cols = [&](size_t c0, size_t c1) -> int { // one pass over G for the columns [c0, c1)
        const int nc = (int)(c1 - c0);
        std::vector<double> Zb((size_t)n * nc, 0.0);
        double t0 = now();
        for (size_t c = c0; c < c1; ++c) {
            const bool fin = c >= (size_t)n_reps * C;
            const int rep = fin ? n_reps : (int)(c / C);
            readyf[rep].wait();
            if (fin ? all_bad : solves[rep].bad) return pg_fail(ctx, PG_ERR_INVALID, "gp_ridge: pinv failed");
            const double *src = fin ? Zall.data() : solves[rep].Z.data();
            const int stride = fin ? k : C, cc = fin ? (int)(c - (size_t)n_reps * C) : (int)(c % C);
            for (int i = 0; i < n; ++i) Zb[(size_t)i * nc + (c - c0)] = src[(size_t)i * stride + cc];
        }
        t_solve += now() - t0; t0 = now();
        const int rc = pg_gp_beta_cols(ctx, G_dev, p, n, ld, Zb.data(), nc, bf + c0 * (size_t)p, 1);
        t_beta += now() - t0;
        return rc;
    };
    for (int rep = 0; rep < n_reps && fused; ++rep) {
        double t0 = now();
        readyf[rep].wait();
        RepSolve &R = solves[rep];
        if (R.bad) return fail2(pg_fail(ctx, PG_ERR_INVALID, "gp_ridge: pinv failed"));
        const std::vector<std::vector<int64_t>> &tr = R.tr, &va = R.va;
        const std::vector<double> &Z = R.Z;
        t_solve += now() - t0; t0 = now();
        int rc = PG_OK;
        if (batched) {
            while (formed < (size_t)(rep + 1) * C && rc == PG_OK) {
                // the first pass takes repetition 0 alone when that costs no extra pass: it then waits for ONE repetition's host
                // solves (5 ms at config 4) instead of two before the device has anything to do
                const bool short_first = formed == 0 && (size_t)C < (size_t)CP &&
                                         1 + (ncols_all - C + CP - 1) / CP == (ncols_all + CP - 1) / CP;
                const size_t c1 = short_first ? (size_t)C : std::min(formed + (size_t)CP, ncols_all);
                rc = form_cols(formed, c1);
                formed = c1;
            }
        } else {
            rc = pg_gp_beta_cols(ctx, G_dev, p, n, ld, Z.data(), C, bf, 1); // :526 for every fold at once, column-major
            t_beta += now() - t0;
        }
        if (rc) return fail2(rc);
        const double *bfr = batched ? bf + (size_t)rep * C * (size_t)p : bf; // this repetition's C columns
        for (int a = 0; a < A; ++a) {
            t0 = now();
            // the redistribution masses of every (fold, trait) column.  Their launches queue up behind the prediction pass that is
            // still out (`pend`), so the latency of this chain of small kernels and of its synchronisation is the device's busy time
            std::vector<FoldMasses> &fm = fm_h[fm_turn ^= 1];
            fm.assign(C, FoldMasses{});
            PathParams P0;
            std::memset(&P0, 0, sizeof P0);
            std::vector<int> skip(C, 0);
            for (int f = 0; f < n_folds; ++f)
                for (int j = 0; j < k; ++j) skip[f * k + j] = va[f].empty() || tr[f].empty();
            std::vector<PathParams> PP;
            rc = ridge_path_params_cols(ctx, bfr, p, C, k, alpha_at(a), path, W, proxy_dev, skip, PP);
            if (rc) return fail2(rc);
            for (int c = 0; c < C; ++c) {
                if (skip[c]) { std::memset(&fm[c], 0, sizeof(FoldMasses)); fm[c].nmax = 1.0; continue; }
                fm[c].nmax = PP[c].nmax;
                for (int i = 0; i < GP_LMAX; ++i) { fm[c].sub_scale[i] = PP[c].sub_scale[i]; fm[c].add_scale[i] = PP[c].add_scale[i]; }
            }
            P0 = PP[0]; // alpha, lambda[], L are the same for every column
            P0.nmax = 0.0;
            if (hipMemcpyAsync(fm_dev, fm.data(), sizeof(FoldMasses) * C, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: H2D failed"));
            t_params += now() - t0; t0 = now();
            for (int j = 0; j < k; ++j) {
                // (the masses' synchronisation has seen the pass that was out: its predictions are on the host)
                const Pending prev = pend;
                const int turn = prev.on ? (prev.buf ^ 1) : 0;
                std::vector<int32_t> &colof = colof_h[turn];
                colof.assign(n, -1);
                for (int f = 0; f < n_folds; ++f)
                    if (!va[f].empty() && !tr[f].empty())
                        for (int64_t pool : va[f]) colof[pool] = f * k + j;
                if (prev.on && hipStreamSynchronize(ctx->stream) != hipSuccess) // (a no-op after the masses' own; the k > 1 traits of one alpha need it)
                    return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: prediction pass failed"));
                if (hipMemcpyAsync(colof_dev, colof.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: H2D failed"));
                const int LP = (P0.L + 1) & ~1;
                const size_t masses_b = sizeof(double) * n_folds * (2 * GP_LMAX + 1);
                const int chunk = std::max(4, std::min(64, (int)((49152 - masses_b) / (sizeof(double) * n_folds * (LP + 2)))));
                const int bthreads = n > 128 ? 512 : 256; // the coefficient stage is shared by all waves of a block
                const int groups = n <= 256 ? bthreads / (((n + 63) / 64) * 64) : 1; // locus groups inside a block (n <= 256: 2 .. 4)
                const dim3 grid(nblk2, groups > 1 ? 1 : (n + bthreads - 1) / bthreads);
                const size_t lds = sizeof(double) * chunk * n_folds * (LP + 2) + masses_b;
                const Proxy X{proxy_dev, k, j};
                pg_prof_begin(ctx, PG_K_GP_PREDICT);
#define PG_PREDICT_FOLDS2(LPV, OD)                                                                                         \
        if (groups > 1)                                                                                                    \
            hipLaunchKernelGGL((k_gp_predict_folds<LPV, true, OD>), grid, dim3(bthreads), lds, ctx->stream, G_dev, bfr, C, colof_dev, fm_dev, P0, X, \
                               p, n, ld, lpb, chunk, W.part, groups);                                                      \
        else                                                                                                               \
            hipLaunchKernelGGL((k_gp_predict_folds<LPV, false, OD>), grid, dim3(bthreads), lds, ctx->stream, G_dev, bfr, C, colof_dev, fm_dev, P0, X, \
                               p, n, ld, lpb, chunk, W.part, 1);
#define PG_PREDICT_FOLDS(LPV)                                                                                              \
    case LPV:                                                                                                              \
        if (P0.L & 1) { PG_PREDICT_FOLDS2(LPV, true) } else { PG_PREDICT_FOLDS2(LPV, false) }                              \
        break;
                switch (LP) {
                    PG_PREDICT_FOLDS(2) PG_PREDICT_FOLDS(4) PG_PREDICT_FOLDS(6) PG_PREDICT_FOLDS(8) PG_PREDICT_FOLDS(10)
                    PG_PREDICT_FOLDS(12) PG_PREDICT_FOLDS(14) PG_PREDICT_FOLDS(16)
                }
#undef PG_PREDICT_FOLDS
#undef PG_PREDICT_FOLDS2
                pg_prof_end(ctx);
                hipLaunchKernelGGL(k_gp_predict_reduce, dim3((n * GP_LMAX + 63) / 64), dim3(512), 0, ctx->stream, W.part, nblk2 * groups, n, W.yhat);
                if (hipGetLastError() != hipSuccess ||
                    hipMemcpyAsync(yh_in[turn].data(), W.yhat, sizeof(double) * n * GP_LMAX, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                    return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: prediction pass failed"));
                pend = Pending{true, rep, a, j, turn};
                // ... and while this pass runs, the host scores the one before it
                if (prev.on) score_pending(prev);
            }
            t_predict += now() - t0;
        }
    }
    if (pend.on) { // the last pass out
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: prediction pass failed"));
        score_pending(pend);
        pend.on = false;
    }
    if (batched) { // whatever is left of the columns (the all-rows fit at least, unless it rode in a repetition's pass)
        int rc = PG_OK;
        while (formed < ncols_all && rc == PG_OK) {
            const size_t c1 = std::min(formed + (size_t)CP, ncols_all);
            rc = form_cols(formed, c1);
            formed = c1;
        }
        if (rc) return fail2(rc);
        if (hipMemcpyAsync(beta_dev, b0all.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: H2D failed"));
        hipLaunchKernelGGL(k_gp_cols_to_rows, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, bf + (size_t)n_reps * C * (size_t)p, p, k, beta_dev + k);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) // (b0all is read by the copy)
            return fail2(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: the all-rows fit failed"));
    }
    const double t_free0 = now();
    (void)hipFree(bf); (void)hipFree(fm_dev); (void)hipFree(colof_dev);
    if (timing)
        std::fprintf(stderr, "gp path: before the repetitions %.1f ms; fold solves %.1f ms, coefficient passes %.1f ms, masses %.1f ms, prediction + scores %.1f ms; the columns' memory: hipMalloc %.1f ms, hipFree %.1f ms; since entry %.1f ms\n",
                     1e3 * t_before, 1e3 * t_solve, 1e3 * t_beta, 1e3 * t_params, 1e3 * t_predict, 1e3 * t_alloc, 1e3 * (now() - t_free0),
                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count());
    for (int rep = 0; rep < n_reps && !fused; ++rep)
        for (int fold = 0; fold < n_folds; ++fold) {
            itr.clear(); iva.clear();
            for (int i = 0; i < n_rows; ++i) {
                const int f = fold_of[(size_t)rep * n_rows + i];
                (f == fold ? iva : itr).push_back(row_idx[i]);
            }
            if (iva.empty() || itr.empty()) continue; // an empty fold leaves NaN, as an empty slice would
            int rc = pg_gp_ols_dev(ctx, G_dev, p, n, ld, Y, k, itr.data(), (int)itr.size(), xxt.data(), beta_dev); // :526
            if (rc) return fail(rc);
            if (hipMemcpyAsync(b0.data(), beta_dev, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                return fail(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: D2H failed"));
            for (int a = 0; a < A; ++a)
                for (int j = 0; j < k; ++j) {
                    PathParams P;
                    const Proxy X{proxy_dev, k, j};
                    rc = ridge_path_params(ctx, beta_dev, p, k, j, alpha_at(a), path, W, P, 1, X);
                    if (rc) return fail(rc);
                    hipLaunchKernelGGL(k_gp_blambda, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, beta_dev, p, k, j, P, X, W.B);
                    hipLaunchKernelGGL(k_gp_predict, dim3(nblk2, (n + 255) / 256), dim3(256), 0, ctx->stream, G_dev, W.B, p, n, ld, lpb, W.part);
                    hipLaunchKernelGGL(k_gp_predict_reduce, dim3((n * GP_LMAX + 63) / 64), dim3(512), 0, ctx->stream, W.part, nblk2, n, W.yhat);
                    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(yh.data(), W.yhat, sizeof(double) * n * GP_LMAX, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess)
                        return fail(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: prediction pass failed"));
                    score(rep, fold, a, j, b0[j], iva);
                }
        }
    // all-rows fit; per trait the mode over repetitions of the per-repetition arg-min over the (alpha, lambda) grid
    // (:573-627): alpha and lambda are counted separately, each against the path values
    int rc = batched ? PG_OK : pg_gp_ols_dev(ctx, G_dev, p, n, ld, Y, k, row_idx, n_rows, xxt.data(), beta_dev); // (batched: already in beta_dev)
    if (rc) return fail(rc);
    for (int j = 0; j < k; ++j) {
        std::vector<int> acount(L, 0), lcount(L, 0);
        for (int rep = 0; rep < n_reps; ++rep) {
            std::vector<double> mean((size_t)A * L);
            for (int a = 0; a < A; ++a)
                for (int li = 0; li < L; ++li) {
                    double sum = 0.0;
                    for (int fold = 0; fold < n_folds; ++fold) sum += perf[((((size_t)rep * n_folds + fold) * A + a) * L + li) * k + j];
                    mean[(size_t)a * L + li] = sum / (double)n_folds;
                }
            double mnv = mean[0];
            for (double x : mean) if (x < mnv) mnv = x;
            for (size_t q = 0; q < mean.size(); ++q)
                if (mean[q] == mnv) {
                    const double aval = alpha_at((int)(q / L)), lval = path[q % L];
                    for (int c = 0; c < L; ++c) { acount[c] += (aval == path[c]); lcount[c] += (lval == path[c]); }
                    break;
                }
        }
        int amax = 0, lmax = 0, abest = 0, lbest = 0;
        for (int c = 0; c < L; ++c) { amax = std::max(amax, acount[c]); lmax = std::max(lmax, lcount[c]); }
        for (int c = 0; c < L; ++c) if (acount[c] == amax) { abest = c; break; }
        for (int c = 0; c < L; ++c) if (lcount[c] == lmax) { lbest = c; break; }
        // a single alpha off the path grid counts nowhere in the reference (:605-608), which then reports and applies
        // path[0]; such an alpha never reaches it from its own callers (0, 1, grid), here it is kept as given
        const double afinal = alpha >= 0.0 ? alpha : path[abest];
        if (alphas_out) alphas_out[j] = afinal;
        lambdas_out[j] = path[lbest];
        PathParams P;
        const Proxy X{proxy_dev, k, j};
        rc = ridge_path_params(ctx, beta_dev, p, k, j, afinal, path, W, P, 1, X);
        if (rc) return fail(rc);
        hipLaunchKernelGGL(k_gp_apply, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, ctx->stream, beta_dev, p, k, j, P, X, lbest);
        if (hipGetLastError() != hipSuccess) return fail(pg_fail(ctx, PG_ERR_HIP, "gp_ridge: apply failed"));
    }
    if (perf_out) std::memcpy(perf_out, perf.data(), sizeof(double) * perf.size());
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(raw);
    if (std::getenv("POOLGEN_GP_TIMING"))
        std::fprintf(stderr, "gp path: whole call %.1f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count());
    return PG_OK;
}

} // namespace

extern "C" int pg_gp_ridge_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y,
                               int k, const int64_t *row_idx, int n_rows, const int32_t *fold_of, int n_reps,
                               int n_folds, double alpha, double lambda_step, double *beta_dev,
                               double *lambdas_out, double *perf_out) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G_dev && Y && row_idx && fold_of && beta_dev && lambdas_out, "gp_ridge: null pointer");
    PG_CHECK(ctx, p > 0 && n >= 3 && k >= 1 && k <= 8 && n_rows >= 3 && n_rows <= n && n_reps >= 1 && n_folds >= 2,
             "gp_ridge: bad shape");
    PG_CHECK(ctx, alpha >= 0.0 && alpha <= 1.0 && lambda_step > 0.0 && lambda_step <= 1.0, "gp_ridge: bad alpha / lambda step");
    return penalised_path(ctx, G_dev, p, n, ld, Y, k, row_idx, n_rows, fold_of, n_reps, n_folds, alpha, nullptr, lambda_step,
                          beta_dev, nullptr, lambdas_out, perf_out, nullptr);
}

extern "C" int pg_gp_penalised_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y,
                                   int k, const int64_t *row_idx, int n_rows, const int32_t *fold_of, int n_reps,
                                   int n_folds, double alpha, int iterative_proxy, double lambda_step, double *beta_dev,
                                   double *alphas_out, double *lambdas_out, double *perf_out, const double *XXt_host_or_null) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G_dev && Y && row_idx && fold_of && beta_dev && lambdas_out, "gp_penalised: null pointer");
    PG_CHECK(ctx, p > 0 && n >= 3 && k >= 1 && k <= 8 && n_rows >= 3 && n_rows <= n && n_reps >= 1 && n_folds >= 2,
             "gp_penalised: bad shape");
    PG_CHECK(ctx, alpha <= 1.0 && lambda_step > 0.0 && lambda_step <= 1.0, "gp_penalised: bad alpha / lambda step");
    double *proxy = nullptr;
    if (iterative_proxy) { // the same proxy serves every fold and the final fit (:543, :656: always on `row_idx`)
        PG_HIP(ctx, hipSetDevice(ctx->device));
        PG_HIP(ctx, hipMalloc((void **)&proxy, sizeof(double) * (size_t)(p + 1) * k));
        const int rc = pg_gp_proxy_dev(ctx, G_dev, p, n, ld, Y, k, row_idx, n_rows, XXt_host_or_null, proxy);
        if (rc) { (void)hipFree(proxy); return rc; }
    }
    const int rc = penalised_path(ctx, G_dev, p, n, ld, Y, k, row_idx, n_rows, fold_of, n_reps, n_folds, alpha, proxy, lambda_step,
                                  beta_dev, alphas_out, lambdas_out, perf_out, XXt_host_or_null);
    if (proxy) (void)hipFree(proxy);
    return rc;
}

namespace {
} // namespace

// yhat = X beta for every pool (the multiply_views_xx of gp/cv.rs:160-168, all rows at once)
extern "C" int pg_gp_predict_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *beta_dev,
                                 int k, double *yhat) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, G_dev && beta_dev && yhat && p > 0 && n >= 1 && k >= 1 && k <= 8, "gp_predict: bad arguments");
    PG_CHECK(ctx, ld >= n, "gp_predict: ld must be >= n");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int nblk = std::max(1, std::min<int>(ctx->cus * 4, (int)((p + 255) / 256)));
    const int64_t lpb = (p + nblk - 1) / nblk;
    const int nblk2 = (int)((p + lpb - 1) / lpb);
    const size_t need = sizeof(double) * ((size_t)nblk2 * n * 8 + (size_t)n * 8);
    int rc = pg_ws_reserve(ctx, need);
    if (rc) return rc;
    double *part = static_cast<double *>(ctx->ws);
    double *out = part + (size_t)nblk2 * n * 8;
    hipLaunchKernelGGL(k_gp_predict_beta, dim3(nblk2, (n + 255) / 256), dim3(256), 0, ctx->stream, G_dev, beta_dev, k, p, n, ld, lpb, part);
    hipLaunchKernelGGL(k_gp_predict_beta_reduce, dim3((n * 8 + 255) / 256), dim3(256), 0, ctx->stream, part, nblk2, n, out);
    PG_HIP(ctx, hipGetLastError());
    std::vector<double> h((size_t)n * 8), b0(k);
    PG_HIP(ctx, hipMemcpyAsync(h.data(), out, sizeof(double) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(b0.data(), beta_dev, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) yhat[(size_t)i * k + j] = b0[j] + h[(size_t)i * 8 + j];
    return PG_OK;
}
