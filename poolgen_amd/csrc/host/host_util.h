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
// Counts of a whole sync file, locus-major: L x n x 6 u32, columns A,T,C,G,N,D in the READER's order
// (sync.rs:134).  The buffer comes from `alloc` (default: malloc) so that the CLI can hand out pinned
// memory and copy it to the device without a staging pass.
struct SyncBatch {
    int n = 0;                              // pools
    int64_t L = 0;                          // loci
    std::vector<int32_t> chrom_id;          // per locus: index into chrom_names
    std::vector<std::string> chrom_names;   // distinct chromosome names in order of first appearance
    std::vector<uint64_t> pos;
    uint32_t *counts = nullptr;
    void (*release)(void *) = nullptr;
    int64_t size() const { return L; }
    const std::string &chrom(int64_t l) const { return chrom_names[chrom_id[l]]; }
    size_t counts_bytes() const { return sizeof(uint32_t) * (size_t)L * n * 6; }
    SyncBatch() = default;
    SyncBatch(const SyncBatch &) = delete;
    SyncBatch &operator=(const SyncBatch &) = delete;
    SyncBatch(SyncBatch &&o) noexcept { *this = std::move(o); }
    SyncBatch &operator=(SyncBatch &&o) noexcept;
    ~SyncBatch();
};
struct SyncAlloc {
    void *(*alloc)(size_t) = nullptr;   // nullptr = malloc / free
    void (*release)(void *) = nullptr;
};
// String::lparse for every line of the file (base/sync.rs:100-156): `n_threads` workers over byte
// ranges split at line starts (helpers.rs:74-91), loci in file order.  Lines starting with '#' and
// lines whose position is not an integer are skipped (both are ErrorKind::Other, which per_chunk
// answers with `continue`, sync.rs:111-128, :829-846); allele counts that are not integers are an error
// (`expect`, :141).  Only the first six ':'-separated counts of a pool are used, as in the reference.
SyncBatch parse_sync_file(const std::string &fname, int n_threads, SyncAlloc alloc = SyncAlloc());

extern const char ALLELES[7];

} // namespace pgh
