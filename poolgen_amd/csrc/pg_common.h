// pg_common.h -- internal declarations shared by the libpoolgen_hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/poolgen_hip.h"

#define PG_MAX_SWEEP_COLS 34 // Q columns (m+1) + traits handled by one sweep launch

struct pg_event_pair { hipEvent_t a, b; int kid; };

struct pg_ctx {
    int device = -1;
    int cus = 256; // compute units of the device (cached at pg_create)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // profiling
    bool prof = false;
    std::vector<pg_event_pair> ev_pending;
    std::vector<pg_event_pair> ev_free;
    double prof_ms[PG_K_COUNT] = {0};
    int64_t prof_n[PG_K_COUNT] = {0};
    // generic device workspace (grown on demand, reused between calls)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // regression state set by pg_kinship_set / pg_covariates_set
    int st_n = 0, st_m = -1, st_k = 0, st_cols = 0;
    double *W_dev = nullptr;     // n x st_cols row-major: [Q_0..Q_m | ytilde_0..ytilde_{k-1}]
    double *syy_dev = nullptr;   // k
    double *tcoef_dev = nullptr; // t-distribution series coefficients for df = n - 1
    int tcoef_df = 0, tcoef_len = 0;
    size_t W_cap = 0;
    double *S_dev = nullptr;     // n x n kinship sum of the single-GPU convenience path
    int S_n = 0;
    // speculative intercept-only sums produced by the kinship pass (see pg_set_phenotypes)
    std::vector<double> ph_Y;    // n x k row-major copy of the phenotypes announced up front
    int ph_n = 0, ph_k = 0;
    double *ph_ytil_dev = nullptr; // k x 256 centred phenotypes, zero padded
    double ph_syy[4] = {0, 0, 0, 0};
    double *spec_dev = nullptr;  // p x (2 + k): sum g', sum g'^2, sum g' ytil_t   (g' = g - g[0])
    size_t spec_cap = 0;
    const double *spec_G = nullptr;
    int64_t spec_p = 0, spec_ld = 0;
    int spec_n = 0, spec_k = 0;
    bool spec_valid = false;
    bool st_Y_matches_ph = false;
    // loader plan (pg_load_plan_dev -> pg_load_emit_dev); lives in ws, so any other ws user invalidates it
    bool load_valid = false;
    const uint32_t *load_counts = nullptr;
    const int64_t *load_order = nullptr;
    int64_t load_L = 0, load_total = 0, load_nunits = 0;
    int load_n = 0, load_kpm1 = 0, load_pshift = 0;
    size_t load_off_flags = 0, load_off_local = 0, load_off_blockoff = 0, load_off_poolmap = 0;
    int64_t lo_last_L = 0, lo_last_listed = 0; // pg_locus_op_stats
    bool rows_call[2] = {true, true};   // ... and what the current call's launch groups run
    bool rows_next[2] = {true, true}; // ols_iter, chisq_test: the next batch runs the order-free kernel (the last one looked error-bearing, or none has run)
    double *lz_dev = nullptr;    // per-wave (1'S1, trace S) partials of the lazy-kinship sweep
    bool lazy_taken = false;     // the last pg_ols_kinship_dev decided m = 0 without forming K
    std::vector<double> st_Y;    // phenotypes of the last m = 0 covariate state (lets an identical call skip the upload)
    // small pinned host staging
    void *pin = nullptr;
    size_t pin_bytes = 0;
    // RCCL communicator of the locus-sharded path (pg_comm.cpp); null = single GPU
    void *comm = nullptr;
    int comm_size = 1, comm_rank = 0;
};

int pg_fail(pg_ctx *ctx, int code, const char *fmt, ...);
#define PG_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return pg_fail(ctx, PG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                               \
    } while (0)
#define PG_CHECK(ctx, cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return pg_fail(ctx, PG_ERR_INVALID, __VA_ARGS__); \
    } while (0)

int pg_ws_reserve(pg_ctx *ctx, size_t bytes);
int pg_pin_reserve(pg_ctx *ctx, size_t bytes);
constexpr int PG_PROF_CONT = 0x100; // pg_prof_begin(kid | PG_PROF_CONT): more device time of an operation already counted
void pg_prof_begin(pg_ctx *ctx, int kid);
void pg_prof_end(pg_ctx *ctx);

// host math (pg_hostmath.cpp)
// symmetric eigen-decomposition: eigenvalues descending, eigenvectors in columns of V (row-major)
int pg_sym_eig(const double *A, int n, double *evals, double *V, bool want_vectors);
int pg_sym_eig_top(const double *A, int n, int m, double *evals, double *V); // all values, the m leading vectors (n x m)
// thin Householder QR of Z (n x c row-major) -> Q (n x c row-major), returns numerical rank
int pg_thin_qr(const double *Z, int n, int c, double *Q);
// t-distribution finite-series coefficients (Abramowitz & Stegun 26.7.3/26.7.4)
std::vector<double> pg_tdist_coef(int df);
// symmetric pseudo-inverse with the reference's tolerance (helpers.rs:463-482)
int pg_pinv_sym(const double *A, int n, double *out);
int pg_gp_subset_solve(const double *xxt, int n, const double *Y, int k, const int64_t *rows, int r, double *V);
int pg_gp_beta_cols(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const double *Z_host, int ncol,
                    double *out_dev, int colmajor = 0, double *ss_out_dev = nullptr); // out p x ncol, or ncol x p when colmajor; ss: g'g per row
int pg_pinv_solve_sym(const double *A, int n, const double *B, int k, double *X); // pinv(A) B, Cholesky when A is safely SPD

// launchers (defined in the .hip files)
int pg_launch_kinship(pg_ctx *ctx, const double *G, int64_t p, int n, int64_t ld, double *S,
                      bool add_intercept, int kid, bool allow_fuse = false);
int pg_launch_kinship_w8(pg_ctx *ctx, const double *G, int64_t p, int n, int64_t ld, double *S,
                         bool add_intercept, int kid, bool allow_fuse = false); // 8-wave workgroups (pg_kinship_w8.hip): <= 64 pools
