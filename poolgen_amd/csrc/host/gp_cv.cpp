// gp_cv.cpp -- see gp_cv.h
#include "gp_cv.h"
#include "host_util.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fcntl.h>
#include <stdexcept>
#include <unistd.h>

namespace pgh {

namespace {

void ok(pg_ctx *ctx, int rc, const char *what) {
    if (rc != PG_OK) throw std::runtime_error(std::string(what) + ": " + pg_last_error(ctx));
}
void hip_ok(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
FILE *create_new_file(const std::string &name) {
    const int fd = ::open(name.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0644);
    if (fd < 0) throw std::runtime_error("Unable to create file: " + name + " (it must not exist)");
    return ::fdopen(fd, "w");
}
std::string strip_ext(const std::string &f) {
    const auto p = f.rfind('.');
    return p == std::string::npos ? std::string() : f.substr(0, p);
}

// pearsons_correlation in "sensible_corr" mode on complete vectors (gwas/correlation_test.rs:7-71)
double pearson_sensible(const std::vector<double> &x, const std::vector<double> &y) {
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
    return std::round(r * 1e7) / 1e7;
}

// What every model function of the reference receives, `(x, y, row_idx)` (main.rs:402-411), in this build's terms: the
// matrix in HBM, the phenotypes, the one full-data X X^T, and the generator the inner folds are drawn from.
struct GpData {
    pg_ctx *ctx;
    const double *G_dev;
    int64_t p;
    int n;
    int64_t ld;
    const std::vector<double> *Y;
    int k;
    const std::vector<double> *xxt;
    SplitMix64 *rng;
};
// fn(&Array2<f64>, &Array2<f64>, &Vec<usize>) -> io::Result<(Array2<f64>, String)>: b_hat ((1 + p) x k) to the device buffer,
// the model's name returned
using GpModelFn = std::string (*)(const GpData &, const std::vector<int64_t> &row_idx, double *b_hat_dev);

std::string ols(const GpData &d, const std::vector<int64_t> &row_idx, double *b_hat_dev) { // gp/ols.rs:8-101
    ok(d.ctx, pg_gp_ols_dev(d.ctx, d.G_dev, d.p, d.n, d.ld, d.Y->data(), d.k, row_idx.data(), (int)row_idx.size(), d.xxt->data(), b_hat_dev), "ols");
    return "ols";
}

// penalised_lambda_path_with_k_fold_cross_validation(x, y, row_idx, alpha, iterative, 0.1, 10) (gp/penalise.rs:461-669) and the
// name the callers build from its alphas and lambdas (:118-129)
std::string penalised(const char *function_name, double alpha, int iterative, const GpData &d, const std::vector<int64_t> &rows,
                      double *b_hat_dev) {
    const int nr = (int)rows.size(), inner_reps = 10;
    int nf = 0;
    std::vector<int32_t> folds((size_t)inner_reps * nr);
    for (int rep = 0; rep < inner_reps; ++rep) { // 10 repetitions of k_split(row_idx, 10) (:509-512)
        std::vector<int64_t> perm = d.rng->permutation(nr), order(nr);
        for (int i = 0; i < nr; ++i) order[i] = rows[perm[i]]; // a shuffle of the row VALUES indexes the group list (:452-456)
        const std::vector<int32_t> g = k_split(nr, 10, order, nf);
        for (int i = 0; i < nr; ++i) folds[(size_t)rep * nr + i] = g[i]; // group nf: the left-over, never validated
    }
    std::vector<double> lam(d.k), al(d.k);
    ok(d.ctx, pg_gp_penalised_dev(d.ctx, d.G_dev, d.p, d.n, d.ld, d.Y->data(), d.k, rows.data(), nr, folds.data(), inner_reps, nf, alpha,
                                  iterative, 0.1, b_hat_dev, al.data(), lam.data(), nullptr, d.xxt->data()), function_name);
    std::string name = std::string(function_name) + "-alphas_";
    for (int j = 0; j < d.k; ++j) name += (j ? "_" : "") + rust_display(al[j]);
    name += "-lambdas_";
    for (int j = 0; j < d.k; ++j) name += (j ? "_" : "") + rust_display(lam[j]);
    return name;
}
std::string penalise_lasso_like(const GpData &d, const std::vector<int64_t> &r, double *b) { return penalised("penalise_lasso_like", 1.00, 0, d, r, b); }   // :101-130
std::string penalise_ridge_like(const GpData &d, const std::vector<int64_t> &r, double *b) { return penalised("penalise_ridge_like", 0.00, 0, d, r, b); }   // :133-159
std::string penalise_glmnet(const GpData &d, const std::vector<int64_t> &r, double *b) { return penalised("penalise_glmnet", -0.1, 0, d, r, b); }           // :162-188
std::string penalise_lasso_like_with_iterative_proxy_norms(const GpData &d, const std::vector<int64_t> &r, double *b) {                                     // :191-217
    return penalised("penalise_lasso_like_with_iterative_proxy_norms", 1.00, 1, d, r, b);
}
std::string penalise_ridge_like_with_iterative_proxy_norms(const GpData &d, const std::vector<int64_t> &r, double *b) {                                     // :220-246
    return penalised("penalise_ridge_like_with_iterative_proxy_norms", 1.00, 1, d, r, b); // alpha = 1 here too, as written (:225-227)
}

} // namespace

std::string gp_cross_validate(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const std::vector<double> &Y,
                              int m, const std::vector<std::string> &pool_names, const CvLabels &labels, const CvArgs &a) {
    // the `functions` vector of main.rs:402-411, in that order
    const GpModelFn functions[] = {ols,
                                   penalise_glmnet,
                                   penalise_lasso_like,
                                   penalise_ridge_like,
                                   penalise_lasso_like_with_iterative_proxy_norms,
                                   penalise_ridge_like_with_iterative_proxy_norms};
    const int nmod = 6;
    const int k = a.k_folds, r = a.n_reps;
    SplitMix64 rng(a.seed);
    double *beta_dev = nullptr;
    hip_ok(hipMalloc((void **)&beta_dev, sizeof(double) * (size_t)(p + 1) * m), "device memory for the coefficients");
    std::vector<double> xxt((size_t)n * n);
    { // the full-data X X^T once: every training subset of every fit uses a principal sub-block
        double *S = nullptr;
        hip_ok(hipMalloc((void **)&S, sizeof(double) * n * n), "device memory");
        ok(ctx, pg_gp_xxt_dev(ctx, G_dev, p, n, ld, S), "X X^T");
        hip_ok(hipMemcpy(xxt.data(), S, sizeof(double) * n * n, hipMemcpyDeviceToHost), "D2H");
        (void)hipFree(S);
    }
    const GpData data{ctx, G_dev, p, n, ld, &Y, m, &xxt, &rng};
    auto fit = [&](int mi, const std::vector<int64_t> &rows, std::string &name) { name = functions[mi](data, rows, beta_dev); };
    const size_t cells = (size_t)r * k * nmod * m;
    std::vector<double> cor(cells, NAN), mbe(cells, NAN), mae(cells, NAN), mse(cells, NAN), rmse(cells, NAN);
    std::vector<double> yvp((size_t)r * nmod * n * 2 * m, NAN); // predicted traits, then expected traits
    std::vector<std::string> names(nmod);
    std::vector<double> yhat((size_t)n * m);
    for (int rep = 0; rep < r; ++rep) {
        int kk = 0;
        const std::vector<int32_t> grp = k_split(n, k, rng.permutation(n), kk);
        for (int fold = 0; fold < kk && fold < k; ++fold) {
            std::vector<int64_t> val, tr;
            for (int i = 0; i < n; ++i) (grp[i] == fold ? val : tr).push_back(i);
            for (int mi = 0; mi < nmod; ++mi) {
                std::string name;
                fit(mi, tr, name);
                if (rep == 0 && fold == 0) names[mi] = name; // "for brevity" (cv.rs:171-173)
                ok(ctx, pg_gp_predict_dev(ctx, G_dev, p, n, ld, beta_dev, m, yhat.data()), "predict");
                for (int64_t pool : val)
                    for (int j = 0; j < m; ++j) {
                        yvp[(((size_t)rep * nmod + mi) * n + pool) * 2 * m + j] = yhat[(size_t)pool * m + j];
                        yvp[(((size_t)rep * nmod + mi) * n + pool) * 2 * m + m + j] = Y[(size_t)pool * m + j];
                    }
                for (int j = 0; j < m; ++j) { // performance (cv.rs:51-103): mae and mse are SUMS, as written
                    std::vector<double> yt, yp;
                    for (int64_t pool : val) { yt.push_back(Y[(size_t)pool * m + j]); yp.push_back(yhat[(size_t)pool * m + j]); }
                    double sd = 0, sa = 0, sq = 0;
                    for (size_t i = 0; i < yt.size(); ++i) { const double d = yt[i] - yp[i]; sd += d; sa += std::fabs(d); sq += d * d; }
                    const size_t c = (((size_t)rep * k + fold) * nmod + mi) * m + j;
                    cor[c] = pearson_sensible(yt, yp);
                    mbe[c] = sd / (double)yt.size();
                    mae[c] = sa;
                    mse[c] = sq;
                    rmse[c] = std::sqrt(sq);
                }
            }
        }
    }
    // ---- tabulate_predict_and_output (cv.rs:226-414) ----------------------------------------------------------
    std::string out = a.fname_output;
    if (out.empty()) {
        const double t = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        out = strip_ext(a.fname_input) + "-cross_validation-" + rust_display(t) + ".csv";
    }
    FILE *fo = create_new_file(out);
    fputs("#rep,fold,model,phenotype,pearsons_correlation,mean_bias_error,mean_absolute_error,mean_square_error,root_mean_square_error\n", fo);
    for (int rep = 0; rep < r; ++rep)
        for (int fold = 0; fold < k; ++fold)
            for (int mi = 0; mi < nmod; ++mi)
                for (int j = 0; j < m; ++j) {
                    const size_t c = (((size_t)rep * k + fold) * nmod + mi) * m + j;
                    const std::string line = std::to_string(rep) + "," + std::to_string(fold) + "," + names[mi] + "," + std::to_string(j) + "," +
                                             rust_display(cor[c]) + "," + rust_display(mbe[c]) + "," + rust_display(mae[c]) + "," +
                                             rust_display(mse[c]) + "," + rust_display(rmse[c]) + "\n";
                    fputs(line.c_str(), fo);
                }
    fclose(fo);
    const std::string base = strip_ext(out);
    fo = create_new_file(base + "-expected_and_predicted_phenotypes.csv");
    {
        std::string h = "#rep,model,pool";
        for (int j = 0; j < m; ++j) h += ",predicted_trait_" + std::to_string(j);
        for (int j = 0; j < m; ++j) h += ",expected_trait_" + std::to_string(j);
        fputs((h + "\n").c_str(), fo);
    }
    for (int rep = 0; rep < r; ++rep)
        for (int mi = 0; mi < nmod; ++mi)
            for (int pool = 0; pool < n; ++pool) {
                std::string line = std::to_string(rep) + "," + names[mi] + "," + pool_names[pool];
                for (int j = 0; j < 2 * m; ++j) line += "," + rust_display(yvp[(((size_t)rep * nmod + mi) * n + pool) * 2 * m + j]);
                fputs((line + "\n").c_str(), fo);
            }
    fclose(fo);
    // all-data predictors of every model (cv.rs:366-410)
    std::vector<int64_t> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    std::vector<double> b((size_t)(p + 1) * m);
    for (int mi = 0; mi < nmod; ++mi) {
        std::string name;
        fit(mi, all, name);
        ok(ctx, pg_synchronize(ctx), "fit");
        hip_ok(hipMemcpy(b.data(), beta_dev, sizeof(double) * b.size(), hipMemcpyDeviceToHost), "D2H coefficients");
        fo = create_new_file(base + "-genomic_predictors-" + name + ".csv");
        fputs("#chromosome,position,allele,phenotype,predictor\n", fo);
        std::string text;
        for (int64_t i = 0; i <= p; ++i)
            for (int j = 0; j < m; ++j) {
                text += labels.chromosome[i] + "," + std::to_string(labels.position[i]) + "," + labels.allele[i] + "," +
                        std::to_string(j) + "," + rust_display(b[(size_t)i * m + j]) + "\n";
                if (text.size() > (1u << 20)) { fwrite(text.data(), 1, text.size(), fo); text.clear(); }
            }
        fwrite(text.data(), 1, text.size(), fo);
        fclose(fo);
    }
    (void)hipFree(beta_dev);
    return out;
}

} // namespace pgh
