// operators.h -- the reference's per-locus operator interface on top of the C ABI, with the reference's names and
// argument meaning (base/structs_and_traits.rs:68-78, :104-136; the fn pointers handed to read_analyse_write,
// main.rs:245-271): one locus in, Option<String> out -- Some(CSV rows) or None when the locus is dropped; failures
// inside an operator are None, never an error (gwas/ols.rs:215-252).  One GPU call per locus: this is the interface a
// test written like the reference's own unit tests talks to; throughput comes from the batch entry points beneath it
// (pg_*_batch[_dev]), which the CLI drives.
#pragma once
#include "../../../include/poolgen_hip.h"
#include <cstdint>
#include <optional>
#include <string>
#include <vector>

namespace pgh {

struct FilterStats { // structs_and_traits.rs:68-78
    bool remove_ns = true;
    bool keep_lowercase_reference = false;
    double max_base_error_rate = 0.01;
    double min_coverage_breadth = 1.0;
    uint64_t min_coverage_depth = 1;
    double min_allele_frequency = 0.001;
    double max_missingness_rate = 0.0;
    std::vector<double> pool_sizes;
};

struct LocusCounts { // structs_and_traits.rs:104-110
    std::string chromosome;
    uint64_t position = 0;
    std::vector<std::string> alleles_vector; // column labels among A, T, C, G, N, D (sync.rs:134)
    std::vector<uint64_t> matrix;            // n x alleles_vector.size(), row-major
};

struct LocusCountsAndPhenotypes { // structs_and_traits.rs:131-136
    LocusCounts locus_counts;
    std::vector<double> phenotypes;          // n x k, row-major
    std::vector<std::string> pool_names;
};

// Formats the rows of one locus exactly as the reference's operators do; shared with the CLI's writer.
// mode 0: chisq (tables/chisq_test.rs:37-45), 1: correlation (gwas/correlation_test.rs:113-127), 2: ols_iterate
// (gwas/ols.rs:255-275).  n_out <= 0 appends nothing.  The pointers are those of the locus' slot 0 in the library's slot-major
// arrays (include/poolgen_hip.h), slot_stride = the L of the call that filled them (1 for a single locus).
void format_locus_rows(int mode, const std::string &chromosome, uint64_t position, int n_out, const int32_t *ids, const double *mean_freq,
                       const double *stat, const double *pval, int k, std::string &out, size_t slot_stride = 1);

class Operators {
public:
    explicit Operators(pg_ctx *ctx) : ctx_(ctx) {}
    std::optional<std::string> chisq(LocusCounts &locus, const FilterStats &f) const;                         // tables::chisq
    std::optional<std::string> correlation(LocusCountsAndPhenotypes &locus, const FilterStats &f) const;      // gwas::correlation
    std::optional<std::string> ols_iterate(LocusCountsAndPhenotypes &locus, const FilterStats &f) const;      // gwas::ols_iterate
private:
    std::optional<std::string> run(int mode, const LocusCounts &lc, const double *Y, int k, const FilterStats &f) const;
    pg_ctx *ctx_;
};

} // namespace pgh
