/*
 * poolgen_hip.h -- C ABI of libpoolgen_hip.so: the MI355X (gfx950) implementation of poolgen's
 * per-locus regression hot path.  This is the drop-in boundary: plain pointers and sizes, no
 * C++/torch types.  Each entry point cites the reference interface (file:line relative to the
 * poolgen source tree) it replaces; INTEGRATION.md shows the Rust `extern "C"` binding a
 * poolgen maintainer would add.
 *
 * Conventions
 *  - All floating point is fp64.  Integer/index outputs are bit-exact w.r.t. the reference.
 *  - Genotype matrices are LOCUS-MAJOR: G[l * ld + i] = allele frequency of pool i at column
 *    (locus/allele) l, ld >= n, ld even (16-byte aligned rows).  The reference stores
 *    `intercept_and_allele_frequencies` pool-major n x (1+p) (base/structs_and_traits.rs:144);
 *    the intercept column is implicit here.
 *  - Phenotypes Y are n x k row-major on the HOST (the reference's Array2<f64>, :145).
 *  - `*_dev` pointers are device (HBM) addresses on the context's GPU; everything else is host.
 *  - Return value: 0 on success, negative pg_status on failure; pg_last_error() gives text.
 *    A single bad locus never fails a batch: it is reported per locus (None -> n_alleles = 0,
 *    regression failure -> NaN) exactly like the reference operators (gwas/ols.rs:210-253,
 *    :358-369).
 *  - One in-flight call per pg_ctx; contexts are independent (one per GPU / per process rank).
 *  - There is NO CPU fallback: without a usable gfx950 device every compute call fails.
 */
#ifndef POOLGEN_HIP_H
#define POOLGEN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pg_ctx pg_ctx;

enum pg_status {
    PG_OK = 0,
    PG_ERR_INVALID = -1,   /* bad shape / argument */
    PG_ERR_HIP = -2,       /* HIP runtime error (text in pg_last_error) */
    PG_ERR_NO_DEVICE = -3, /* no gfx950 device */
    PG_ERR_STATE = -4,     /* call order violated (e.g. sweep before covariates) */
    PG_ERR_UNSUPPORTED = -5
};

/* FilterStats subset used by the sync-derived operators (base/structs_and_traits.rs:68-78,
 * filled from the CLI flags at main.rs:202-211). */
typedef struct {
    int32_t remove_ns;            /* !--keep-ns */
    int32_t reserved;
    uint64_t min_coverage_depth;  /* --min-coverage-depth */
    double min_allele_frequency;  /* --min-allele-frequency */
    double max_missingness_rate;  /* --max-missingness-rate */
} pg_filter;

/* Per-locus output of the sync-derived batch operators (struct-of-arrays, slot-major, device or host; see below).
 * PG_MAX_OUT alleles (= rows = slots) per locus: 6 sync columns minus the one dropped. */
#define PG_MAX_OUT 5

/* ---------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------- */
/* device: HIP ordinal.  stream: a hipStream_t owned by the caller (e.g. torch's current
 * stream); NULL = the device's default (null) stream.  All kernels and copies of this context
 * are enqueued on that stream, so they are ordered with the caller's own work on it. */
int pg_create(pg_ctx **out, int device, void *stream);
void pg_destroy(pg_ctx *ctx);
const char *pg_last_error(const pg_ctx *ctx);
const char *pg_version(void);
int pg_synchronize(pg_ctx *ctx);
/* Per-kernel HIP-event timing (used by bench.py for the roofline record). */
enum pg_kernel_id { PG_K_KINSHIP = 0, PG_K_KINSHIP_REDUCE = 1, PG_K_SWEEP = 2, PG_K_OLS_ITER = 3,
                    PG_K_PEARSON = 4, PG_K_CHISQ = 5, PG_K_GP_XXT = 6, PG_K_GP_BETA = 7,
                    PG_K_SWEEP_FINISH = 8, PG_K_ALLREDUCE = 9, PG_K_GP_PREDICT = 10, PG_K_COUNT = 11 };
int pg_profile_enable(pg_ctx *ctx, int on);
int pg_profile_reset(pg_ctx *ctx);
/* Synchronises, then returns total milliseconds and launch count of one kernel id. */
int pg_profile_get(pg_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
/* Diagnostics of the last batch operator call (pg_ols_iter_batch[_dev], pg_pearson_batch[_dev], pg_chisq_batch[_dev],
 * pg_load_plan_dev) on this context: *loci = L of that call, *listed = the loci its streaming pass could not close in place
 * and handed to the second pass (three or more surviving alleles, or a speculated allele pair that did not hold; summed
 * over the launch groups of a multi-trait call).  Either pointer may be NULL. */
int pg_locus_op_stats(const pg_ctx *ctx, int64_t *loci, int64_t *listed);

/* ---------------------------------------------------------------------------------------
 * ols_iter_with_kinship   == gwas::ols_with_covariate (gwas/ols.rs:278-436, numeric core
 * :291-370; CLI main.rs:280-298).  Staged so that loci can be sharded over GPUs:
 *   1. pg_kinship_partial_dev : S_r = G_r^T-contraction over this rank's loci (unscaled)
 *   2. (multi-GPU) pg_allreduce_sum_dev : sum of S over the ranks (RCCL, see "Multi-GPU" below)
 *   3. pg_kinship_set        : K = S / p_total, eigen-decomposition, n_eigenvecs rule,
 *                               orthonormal basis of [1 | C], projected phenotypes
 *   4. pg_ols_sweep_dev      : per-column fit of y ~ [1 | C | g], last coefficient
 * pg_ols_kinship_dev runs 1,3,4 on one GPU.
 * ------------------------------------------------------------------------------------- */
/* Optional, before pg_kinship_partial_dev: announce the phenotypes (host, n x k row-major, no NaN,
 * k <= 4).  The kinship pass then also accumulates, from the SAME read of G, the three per-locus
 * sums an intercept-only fit needs.  If the n_eigenvecs rule later yields m = 0 (the outcome of the
 * default -x 0.75 on uncentred allele-frequency kinships, where lambda_1 carries ~98 % of the trace),
 * pg_ols_sweep_dev closes the fits from those sums instead of reading G a second time; for m > 0 the
 * sums are ignored and the regular sweep runs.  Results are the same either way (same formula,
 * different summation order).  pg_set_phenotypes(ctx, 0, NULL, 0) switches the behaviour off. */
int pg_set_phenotypes(pg_ctx *ctx, int n, const double *Y, int k);
/* S_dev: n x n fp64, row-major, full symmetric sum_l g_l g_l^T over this call's p columns. */
int pg_kinship_partial_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld,
                           double *S_dev);
/* S_dev: the (all-reduced) sum.  force_m < 0 => m by the cumulative-variance rule
 * (ols.rs:297-311) on eigenvalues sorted descending (the reference's stated intent, :296);
 * force_m >= 0 overrides it.  Y (host, n x k row-major) must have no NaN (remove_missing is the
 * caller's job, ols.rs:287).  Outputs (host, optional): m_out, K_out n x n, evals_out n. */
int pg_kinship_set(pg_ctx *ctx, const double *S_dev, int64_t p_total, int n, const double *Y,
                   int k, double var_explained, int force_m, int *m_out, double *K_out,
                   double *evals_out);
/* Same as pg_kinship_set but with caller-provided covariates C (host, n x m row-major) instead
 * of kinship eigenvectors; m = 0 gives plain y ~ [1 | g]. */
int pg_covariates_set(pg_ctx *ctx, int n, const double *C, int m, const double *Y, int k);
/* beta/var/pval: p x k row-major on the device; NaN where the reference's fit fails. */
int pg_ols_sweep_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld,
                     double *beta_dev, double *var_dev, double *pval_dev);
/* All three stages in one call.  K_out (host, n x n) is optional, and leaving it NULL (with force_m < 0) permits the LAZY route:
 * the n_eigenvecs rule (gwas/ols.rs:297-311) yields m = 0 as soon as lambda_1 / trace(K) >= var_explained, and
 * lambda_1 >= 1'K1 / n = sum_l (sum_i g_li)^2 / (n p) for any K, so when that bound clears var_explained by 1e-9 (an uncentred
 * kinship of allele frequencies: always) the intercept-only sweep alone -- ONE pass over G, no kinship matrix -- gives the
 * analysis' beta / var / pval (bit-identical to pg_kinship_partial_dev + pg_kinship_set + pg_ols_sweep_dev without fused sums);
 * otherwise the full route runs.  POOLGEN_NO_LAZY_KINSHIP=1 disables it. */
int pg_ols_kinship_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld,
                       const double *Y, int k, double var_explained, int force_m, int *m_out,
                       double *K_out, double *beta_dev, double *var_dev, double *pval_dev);
/* Host-buffer form (PCIe inclusive; what main.rs:285-291 hands over): G, beta, var, pval on the host.  G crosses the
 * bus ONCE, in slabs (default 256 MB, POOLGEN_HOST_SLAB_MB) on a copy stream while the partial kinship of the slab
 * that has landed runs; the matrix stays resident in HBM; after the n x n step the sweep runs slab by slab and the
 * results of slab s return while slab s + 1 is swept.  A pinned caller buffer (hipHostMalloc / hipHostRegister) is
 * used as is; pageable pages are pinned in place by the runtime.  The link sets the pace: ~0.3 s for the 16 GB of
 * 200 pools x 10 M loci against 10 ms of kernels. */
int pg_ols_kinship(pg_ctx *ctx, const double *G, int64_t p, int n, int64_t ld, const double *Y,
                   int k, double var_explained, int force_m, int *m_out, double *K_out,
                   double *beta, double *var, double *pval);

/* mle_iter_with_kinship == gwas::mle_with_covariate (gwas/mle.rs:307-463): the same kinship preamble (:317-343), then per
 * (column, trait) a Nelder-Mead maximum-likelihood fit of y ~ [1 | C | g] on (logit-bounded sigma^2, b) (mle.rs:13-30, :84-115;
 * argmin 0.8, start simplex of base/helpers.rs:132-146, <= 1000 iterations), ve = sigma^2, v_b = ve [(X'X)^-1]_last,
 * t = b / v_b as written (:175), p = 2 (1 - T_{n-1}(|t|)).  PARITY UNPINNED: the reference has no test of this path and the
 * solver's source is not in its tree; the published algorithm is restated (here and, literally, in the oracle) and the two
 * agree at the solver's resolution (~1e-6), not at 1e-10.  Up to 8 kinship covariates (10 design columns: what the eigen rule can
 * hand the sweep as well), k <= 4 traits.  With 3 and more covariates the 1000-iteration cap ends the simplex before it has
 * converged, in the reference as here (tests/test_gpu_mle.py).
 * beta/var/pval: p x k on the device. */
int pg_mle_kinship_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y, int k,
                       double var_explained, int force_m, int *m_out, double *K_out, double *beta_dev, double *var_dev,
                       double *pval_dev);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU: loci shard over the GPUs of a node in contiguous slabs, one pg_ctx (= one GPU, one host thread or
 * process) per rank.  The reference's parallel axis is the same one -- a worker per file chunk (base/sync.rs:913-939)
 * -- and the only quantity that needs every worker's loci is the kinship sum (gwas/ols.rs:291-295): ONE in-place
 * RCCL all-reduce(sum) of n x n doubles over xGMI (320 KB at n = 200: latency-bound).  Outputs stay sharded (rank
 * order = locus order).  ols_iter / pearson_corr / chisq_test need no exchange at all.
 *   bootstrap: one rank calls pg_comm_unique_id and hands the PG_COMM_ID_BYTES to the others by any means (threads
 *   of one process: shared memory; processes: a file, MPI, a torch.distributed store); then EVERY rank calls
 *   pg_comm_init_rank(ctx, id, nranks, rank) -- collective, like ncclCommInitRank.  RCCL itself is loaded on first use.
 * pg_allreduce_sum_dev is the identity on a context without communicator, so single-GPU callers may use the
 * sharded entry point unchanged.
 * ------------------------------------------------------------------------------------- */
#define PG_COMM_ID_BYTES 128
int pg_comm_unique_id(void *id_out);
int pg_comm_init_rank(pg_ctx *ctx, const void *id, int nranks, int rank);
int pg_comm_destroy(pg_ctx *ctx);
int pg_comm_size(const pg_ctx *ctx);
int pg_comm_rank(const pg_ctx *ctx);
int pg_comm_version(int *version_out); /* ncclGetVersion of the RCCL that was loaded */
/* In-place sum over the ranks, enqueued on the context's stream (ordered with the kernels around it). */
int pg_allreduce_sum_dev(pg_ctx *ctx, double *buf_dev, int64_t count);
/* One rank's share of ols_iter_with_kinship: partial kinship of its p_local columns -> all-reduce -> the n x n
 * step with p_total (every rank computes the same m, basis and projected phenotypes) -> sweep of its own slab.
 * beta/var/pval: p_local x k on this rank's device. */
int pg_ols_kinship_sharded_dev(pg_ctx *ctx, const double *G_dev, int64_t p_local, int64_t p_total, int n, int64_t ld,
                               const double *Y, int k, double var_explained, int force_m, int *m_out, double *K_out,
                               double *beta_dev, double *var_dev, double *pval_dev);

/* ---------------------------------------------------------------------------------------
 * Sync-derived per-locus operators.  The reference calls these once per text line through
 * ChunkyReadAnalyseWrite::read_analyse_write (base/structs_and_traits.rs:245-265,
 * base/sync.rs:864, :674); here one call handles a batch of L parsed loci.
 * counts: L x n x 6 uint32, locus-major, sync column order A,T,C,G,N,D (base/sync.rs:134); every count below 2^29
 *   (the streaming pass sums coverages as integers; a larger count fails the call with PG_ERR_INVALID).  The *_dev forms (and
 *   pg_load_plan_dev) require counts_dev to be 16-BYTE ALIGNED (PG_ERR_INVALID otherwise: the streaming pass reads 16-byte
 *   pieces; a hipMalloc'ed buffer always is, a view that starts at an odd locus of a batch with odd n is not -- copy it).
 *   Nothing behind the L * n * 24 bytes of the batch is ever interpreted, whatever it holds.
 *   Real counts carry reads of alleles the MAF filter drops, and the reference recomputes the frequencies on the FILTERED
 *   counts (gwas/ols.rs:210-230 -> base/sync.rs:166-192), so those reads change every denominator: the streaming pass closes
 *   such loci in place whenever exactly two alleles survive; pg_locus_op_stats says how many loci needed the second pass.
 * pool_sizes: host, n (normalised or not -- only ratios are used, sync.rs:266-268).
 * Outputs are struct-of-arrays, SLOT-MAJOR (a locus emits up to PG_MAX_OUT rows = slots):
 *   n_out[L]            int32  rows emitted per trait (0 = locus dropped = None)
 *   allele_ids[5*L]     int32  index into "ATCGND"; element (slot r, locus l) at r*L + l
 *   mean_freq[5*L]      double element (r, l) at r*L + l
 *   stat[5*L*k], pval[5*L*k] double  element (r, l, trait t) at (r*L + l)*k + t
 * Only the slots r < n_out[l] of a locus are specified (chisq_test: one row per locus whose allele list occupies the
 * allele_ids slots r < n_out[l]; its stat / pval are plain [L] arrays).  A biallelic locus emits one row and touches slot 0
 * only: the operators write 32 instead of 144 bytes per locus, and a caller that wants rows copies slot 0 and, where
 * n_out > 1, the further slots.
 * ------------------------------------------------------------------------------------- */
/* gwas::ols_iterate (gwas/ols.rs:201-276): stat = beta. */
int pg_ols_iter_batch_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n,
                          const double *pool_sizes, const pg_filter *filter, const double *Y, int k,
                          int32_t *n_out_dev, int32_t *allele_ids_dev, double *mean_freq_dev,
                          double *stat_dev, double *pval_dev);
/* gwas::correlation (gwas/correlation_test.rs:73-129): stat = Pearson r (rounded to 7 dp as the
 * reference's pearsons_correlation returns it, :70). */
int pg_pearson_batch_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n,
                         const double *pool_sizes, const pg_filter *filter, const double *Y, int k,
                         int32_t *n_out_dev, int32_t *allele_ids_dev, double *mean_freq_dev,
                         double *stat_dev, double *pval_dev);
/* tables::chisq (tables/chisq_test.rs:5-47): n_out = number of alleles kept (all are listed in
 * allele_ids), chi2[L], pval[L]. */
int pg_chisq_batch_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n,
                       const double *pool_sizes, const pg_filter *filter, int32_t *n_out_dev,
                       int32_t *allele_ids_dev, double *chi2_dev, double *pval_dev);
/* The loader, FileSyncPhen::load + into_genotypes_and_phenotypes (base/sync.rs:972-1180), on counts that
 * are already in HBM: per locus LocusCounts::filter (:195-303) -> to_frequencies over the surviving
 * alleles (:166-192) -> with keep_p_minus_1 sort by decreasing frequency and drop the first allele
 * (:1033-1037).  One column of G per surviving allele, loci in the caller's order (`order_dev`: a
 * permutation of 0..L-1, e.g. the (chromosome, position) sort of :1092-1101; NULL = input order).
 * Two calls because the column count is a result:
 *   pg_load_plan_dev  runs the filter for every locus and returns the number of columns p;
 *   pg_load_emit_dev  (directly afterwards, same ctx) writes G (p x ld, locus-major, pools kept =
 *                     pool_map[i] >= 0 at row position pool_map[i]; NULL = all pools), and for every
 *                     column the locus it came from and its allele (index into "ATCGND").
 * The coverage matrix the reference also fills (:1157-1167) is not produced (nothing on this path
 * reads it). */
int pg_load_plan_dev(pg_ctx *ctx, const uint32_t *counts_dev, int64_t L, int n, const double *pool_sizes,
                     const pg_filter *filter, int keep_p_minus_1, const int64_t *order_dev, int64_t *p_out);
int pg_load_emit_dev(pg_ctx *ctx, const int32_t *pool_map, int n_out, double *G_dev, int64_t ld,
                     int64_t *col_locus_dev, int32_t *col_allele_dev);
/* Counts stored as 16-bit integers (the host parser's compact form: half the bytes to pin and to copy) -> the 32-bit
 * L x n x 6 layout every operator reads.  Both buffers on the device, 16-byte aligned; n_values = L * n * 6. */
int pg_expand_counts_u16_dev(pg_ctx *ctx, const uint16_t *src_dev, int64_t n_values, uint32_t *dst_dev);
/* pg_load_emit_dev that also writes the coverages GenotypesAndPhenotypes carries (sync.rs:1129-1152): cov_dev is
 * p x ld like G; every column row of a locus holds, per pool, the depth summed over the locus' surviving alleles. */
int pg_load_emit_cov_dev(pg_ctx *ctx, const int32_t *pool_map, int n_out, double *G_dev, int64_t ld,
                         int64_t *col_locus_dev, int32_t *col_allele_dev, double *cov_dev);
/* Host-buffer forms of the three batch operators (H2D/D2H inside). */
int pg_ols_iter_batch(pg_ctx *ctx, const uint32_t *counts, int64_t L, int n,
                      const double *pool_sizes, const pg_filter *filter, const double *Y, int k,
                      int32_t *n_out, int32_t *allele_ids, double *mean_freq, double *stat,
                      double *pval);
int pg_pearson_batch(pg_ctx *ctx, const uint32_t *counts, int64_t L, int n,
                     const double *pool_sizes, const pg_filter *filter, const double *Y, int k,
                     int32_t *n_out, int32_t *allele_ids, double *mean_freq, double *stat,
                     double *pval);
int pg_chisq_batch(pg_ctx *ctx, const uint32_t *counts, int64_t L, int n, const double *pool_sizes,
                   const pg_filter *filter, int32_t *n_out, int32_t *allele_ids, double *chi2,
                   double *pval);

/* ---------------------------------------------------------------------------------------
 * Genomic prediction: gp::ols (gp/ols.rs:8-101) for the n < p case, b = X^T pinv(X X^T) y,
 * with X = [1 | G^T] (intercept implicit).  beta_dev: (1+p) x k row-major, row 0 = intercept.
 * row_idx: host, training rows (the reference's `row_idx: &Vec<usize>`).
 * pg_gp_xxt_dev exposes the full-data X X^T (n x n, incl. intercept) so that every training
 * subset's X X^T can be taken as a principal sub-block without another pass over G.
 * ------------------------------------------------------------------------------------- */
int pg_gp_xxt_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, double *XXt_dev);
int pg_gp_ols_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y,
                  int k, const int64_t *row_idx, int n_rows, const double *XXt_host_or_null,
                  double *beta_dev);

/* gp::penalise_ridge_like (gp/penalise.rs:133-159) = the lambda path with k-fold cross-validation
 * (:461-669) at alpha = 0, generalised to 0 <= alpha <= 1 (alpha = 1 is penalise_lasso_like, :101-130).
 * The reference draws its folds from an unseeded RNG (:452-453); here they are explicit:
 * fold_of[rep * n_rows + i] in 0..n_folds-1 is the fold of pool row_idx[i] in repetition rep; the value
 * n_folds marks the left-over group k_split makes (:444-448), which trains in every fold and is never validated.
 * lambda path = {0, step, 2 step, ..., 1} (step 0.1 in the reference).  Outputs: beta_dev (1+p) x k
 * (penalised coefficients of the all-rows fit, row 0 = intercept), lambdas_out[k] (host), optional
 * perf_out (host, n_reps x n_folds x L x k error indices, :359-426).
 * Device memory beside G: the slopes of every fold's fit of every repetition and of the all-rows fit stay resident while the
 * repetitions are scored -- (n_reps * n_folds * k + k) * p doubles (BASELINE configs[3]: 101 x 40 MB) -- so that they can be
 * formed 16 columns per pass over G whatever repetition they belong to; when that is more than half of the free device memory
 * (or with POOLGEN_RIDGE_PER_REP=1) one repetition's n_folds * k columns are resident at a time and every repetition costs a
 * pass of its own.  Same results, bit for bit. */
int pg_gp_ridge_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y,
                    int k, const int64_t *row_idx, int n_rows, const int32_t *fold_of, int n_reps,
                    int n_folds, double alpha, double lambda_step, double *beta_dev,
                    double *lambdas_out, double *perf_out);
/* The whole family behind penalised_lambda_path_with_k_fold_cross_validation (gp/penalise.rs:461-669):
 *   alpha in [0, 1], iterative_proxy = 0 : pg_gp_ridge_dev (penalise_lasso_like alpha = 1, penalise_ridge_like alpha = 0);
 *   alpha < 0                            : penalise_glmnet (:168-195): the grid alpha x lambda over the same path values
 *                                          (:479-498), alpha and lambda chosen by separate mode counts (:605-627);
 *   iterative_proxy != 0                 : the *_with_iterative_proxy_norms models (:197-246): the penalised set is
 *                                          picked by the norms of pg_gp_proxy_dev's coefficients (fitted once on
 *                                          row_idx, :543, :656), the amounts by the fit's own norms (:262-283).
 * alphas_out[k] (host, optional), lambdas_out[k]; perf_out (optional) n_reps x n_folds x A x L x k, A = 1 or L.
 * With iterative_proxy the covariate state of ctx (pg_covariates_set / pg_kinship_set) is overwritten.
 * XXt_host_or_null: the full-data X X^T of pg_gp_xxt_dev (host, n x n) when the caller already has it -- a harness that
 * fits many models on the same genotypes computes it once. */
int pg_gp_penalised_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y,
                        int k, const int64_t *row_idx, int n_rows, const int32_t *fold_of, int n_reps,
                        int n_folds, double alpha, int iterative_proxy, double lambda_step, double *beta_dev,
                        double *alphas_out, double *lambdas_out, double *perf_out, const double *XXt_host_or_null);
/* gp::ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199): proxy_dev (1+p) x k on the device, row 0 = the
 * trait means over the training pools, row 1+l = the locus coefficient of y ~ [1 | PC1 | g_l] on the training pools
 * (PC1: leading eigenvector of the reference's centred X X^T of those pools, :115-141, with its two indexing quirks:
 * columns = intercept and all loci but the last; means over the first n_rows pools).  Overwrites ctx's covariates. */
int pg_gp_proxy_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Y, int k,
                    const int64_t *row_idx, int n_rows, const double *XXt_host_or_null, double *proxy_dev);
/* yhat (n x k, host) = X beta for EVERY pool, X = [1 | G^T], beta (1+p) x k on the device: the prediction step of
 * the cross-validation harness (multiply_views_xx in gp/cv.rs:160-168); the caller reads the validation rows. */
int pg_gp_predict_dev(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *beta_dev,
                      int k, double *yhat);

/* ---------------------------------------------------------------------------------------
 * popgen on the loader's matrix: fst (popgen/fst.rs:10-115, :158-200) and
 * theta_pi / `heterozygosity` (popgen/pi.rs:10-113).  G and cov as written by pg_load_emit_cov_dev
 * (all alleles kept); locus_col (host, L + 1 entries, 0 .. p): the first column of every locus =
 * count_loci (sync.rs:73-97) minus the intercept; windows = inclusive locus index ranges from
 * pg_host_sliding_windows = define_sliding_windows (base/helpers.rs:294-403; chromosomes as ids,
 * only equality is used; head/tail need room for L entries; returns the number of windows).
 *   pg_pi_dev : pi_win (host, n_windows x n) per-window means, pi_mean (host, n) their mean (:133).
 *   pg_fst_dev: fst_mean (host, n x n) mean over all loci (:145), fst_win (host, n_windows x n*n);
 *               fails with PG_ERR_INVALID where the reference's assert does (a locus whose
 *               frequencies do not sum to one in every pool, :66).
 * ------------------------------------------------------------------------------------- */
int64_t pg_host_sliding_windows(const int32_t *chr_id, const uint64_t *pos, int64_t L, uint64_t window_size_bp,
                                uint64_t window_slide_size_bp, uint64_t min_loci_per_window, int64_t *head,
                                int64_t *tail);
int pg_pi_dev(pg_ctx *ctx, const double *G_dev, const double *cov_dev, int64_t p, int n, int64_t ld,
              const int64_t *locus_col, int64_t L, const int64_t *win_head, const int64_t *win_tail,
              int64_t n_windows, double *pi_win, double *pi_mean);
int pg_fst_dev(pg_ctx *ctx, const double *G_dev, const double *cov_dev, int64_t p, int n, int64_t ld,
               const int64_t *locus_col, int64_t L, const int64_t *win_head, const int64_t *win_tail,
               int64_t n_windows, double *fst_mean, double *fst_win);

/* ---------------------------------------------------------------------------------------
 * Host-side pieces of the path (O(n^3), n = pools): exported so that they can be validated
 * without a GPU and reused by a host integration.
 * ------------------------------------------------------------------------------------- */
/* Symmetric eigen-decomposition standing in for `kinship.eig()` (gwas/ols.rs:296): eigenvalues
 * DESCENDING, eigenvectors in the columns of V (n x n row-major; V may be NULL). */
int pg_host_sym_eig(const double *A, int n, double *evals, double *V);
/* All eigenvalues (descending) and the m leading eigenvectors only (V: n x m row-major) -- what pg_kinship_set
 * needs for the covariates of gwas/ols.rs:312-315: values-only QL + inverse iteration + back-transformation,
 * verified against A, full decomposition as the fallback. */
int pg_host_sym_eig_top(const double *A, int n, int m, double *evals, double *V);
/* n_eigenvecs from the cumulative-variance rule, literal (gwas/ols.rs:297-311). */
int pg_host_n_eigenvecs(const double *evals, int n, double var_explained);
/* Moore-Penrose pseudo-inverse of a symmetric matrix with the reference tolerance
 * eps * len(s) * max(s) (base/helpers.rs:463-482). */
int pg_host_pinv_sym(const double *A, int n, double *out);
/* Two-sided Student-t p-value for integer df, evaluated by the same finite series the device
 * code uses (stands in for 2 * (1 - StudentsT::cdf(|t|)), gwas/ols.rs:153). */
double pg_host_t_two_sided_p(double t_abs, int df);

#ifdef __cplusplus
}
#endif
#endif
