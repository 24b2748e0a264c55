// host_util.h -- host-side pieces of the poolgen CLI replacement: Rust-compatible number
// formatting (base/helpers.rs:103-117), phenotype parser (base/phen.rs:21-98), sync parser and
// locus filter (base/sync.rs:100-304, :477-506), all in the reference's operation order.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pgh {

// ---- formatting -------------------------------------------------------------------------------
std::string rust_display(double x);                      // Rust `{}` for f64
double sensible_round(double x, int n_digits);           // helpers.rs:103-108
std::string roundup_own(double x, int n_digits);         // helpers.rs:111-117

// ---- phenotypes ---------------------------------------------------------------------------------
struct Phen {
    std::vector<std::string> pool_names;
    std::vector<double> pool_sizes;    // normalised to sum 1 (phen.rs:83-84)
    std::vector<double> phen;          // n x k row-major, NaN = missing
    int n = 0, k = 0;
};
Phen parse_phen(const std::string &fname, const std::string &delim, int name_col, int size_col,
                const std::vector<int> &value_cols);

// ---- sync ---------------------------------------------------------------------------------------
struct SyncBatch {
    int n = 0;                          // pools
    std::vector<std::string> chrom;     // per locus
    std::vector<uint64_t> pos;
    std::vector<uint32_t> counts;       // L x n x 6, columns A,T,C,G,N,D (sync.rs:134)
    int64_t size() const { return (int64_t)pos.size(); }
};
// Parses the whole file with `n_threads` workers over byte ranges split at line starts
// (helpers.rs:74-91); loci come back in file order.  Comment lines are skipped (sync.rs:111-114).
SyncBatch parse_sync_file(const std::string &fname, int n_threads);

extern const char ALLELES[7];

} // namespace pgh
