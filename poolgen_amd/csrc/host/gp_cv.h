// gp_cv.h -- genomic_prediction_cross_validation (gp/cv.rs:10-414) over the GPU fits of libpoolgen_hip: k-fold
// cross-validation with r replicates of ols / penalise_lasso_like / penalise_ridge_like, the performance table, the
// expected-vs-predicted table and the all-data predictors, in the reference's file formats.
#pragma once
#include "../../../include/poolgen_hip.h"
#include "host_util.h"
#include <cstdint>
#include <string>
#include <vector>

namespace pgh {

struct CvLabels { // per coefficient (intercept first), as the reference's GenotypesAndPhenotypes carries them
    std::vector<std::string> chromosome, allele;
    std::vector<uint64_t> position;
};

struct CvArgs {
    int k_folds = 10, n_reps = 3;
    uint64_t seed = 42;
    int n_threads = 1;
    std::string fname_input, fname_output;
};

// Runs the whole analysis on a resident G (p x ld, n pools) and writes the three kinds of output files; returns the name
// of the performance table.
std::string gp_cross_validate(pg_ctx *ctx, const double *G_dev, int64_t p, int n, int64_t ld, const std::vector<double> &Y,
                              int k_traits, const std::vector<std::string> &pool_names, const CvLabels &labels,
                              const CvArgs &args);

} // namespace pgh
