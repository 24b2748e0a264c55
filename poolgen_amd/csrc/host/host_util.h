// host_util.h -- host-side pieces of the poolgen CLI replacement: Rust-compatible number
// formatting (base/helpers.rs:103-117), phenotype parser (base/phen.rs:21-98), sync parser and
// locus filter (base/sync.rs:100-304, :477-506), all in the reference's operation order.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>

namespace pgh {

// ---- formatting -------------------------------------------------------------------------------
std::string rust_display(double x);                      // Rust `{}` for f64
void append_rust_display(std::string &out, double x);    // the same, appended in place (no temporary)
double sensible_round(double x, int n_digits);           // helpers.rs:103-108
std::string roundup_own(double x, int n_digits);         // helpers.rs:111-117
void append_roundup_own(std::string &out, double x, int n_digits); // the same, appended in place

// ---- numbers as Rust's `str::parse` reads them (main.rs flag values, base/phen.rs:60-78) ------------------------
// f64: [+-] digits [. digits] [e [+-] digits] | [+-] inf | infinity | nan (any case); no hex floats, no surrounding blanks,
// nothing after the number.  u64 / i64: [+] digits (i64 also -), in range, nothing else -- "2x", "-1" for an unsigned flag
// or "1e3" for an integer are errors there and here (strtod / stoi would have accepted them).
bool parse_f64_strict(const std::string &s, double &out);
bool parse_u64_strict(const std::string &s, uint64_t &out);
bool parse_i64_strict(const std::string &s, int64_t &out);

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
    uint32_t *counts = nullptr;             // L x n x 6, the device layout ...
    uint16_t *counts16 = nullptr;           // ... or, when asked for and every count fits, 16-bit (half the bytes to pin and to copy)
    std::function<void(void *)> release;   // empty = free()
    int64_t size() const { return L; }
    const std::string &chrom(int64_t l) const { return chrom_names[chrom_id[l]]; }
    size_t counts_bytes() const { return (counts16 ? sizeof(uint16_t) : sizeof(uint32_t)) * (size_t)L * n * 6; }
    const void *counts_raw() const { return counts16 ? (const void *)counts16 : (const void *)counts; }
    SyncBatch() = default;
    SyncBatch(const SyncBatch &) = delete;
    SyncBatch &operator=(const SyncBatch &) = delete;
    SyncBatch(SyncBatch &&o) noexcept { *this = std::move(o); }
    SyncBatch &operator=(SyncBatch &&o) noexcept;
    ~SyncBatch();
};
struct SyncAlloc {
    std::function<void *(size_t)> alloc;    // empty = malloc / free
    std::function<void(void *)> release;
};
// A read-only mapping of a file, and its cut points at line starts (helpers.rs:74-91) for a target piece size.
class MappedFile {
public:
    explicit MappedFile(const std::string &fname);
    ~MappedFile();
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    const char *data() const { return p_; }
    size_t size() const { return n_; }
    std::vector<size_t> cuts(size_t pieces) const; // pieces + 1 offsets (fewer if lines are long), first 0, last size()
private:
    const char *p_ = nullptr;
    size_t n_ = 0;
};
// String::lparse for every line of the file (base/sync.rs:100-156): `n_threads` workers over byte
// ranges split at line starts (helpers.rs:74-91), loci in file order.  Lines starting with '#' and
// lines whose position is not an integer are skipped (both are ErrorKind::Other, which per_chunk
// answers with `continue`, sync.rs:111-128, :829-846); allele counts that are not integers are an error
// (`expect`, :141).  Only the first six ':'-separated counts of a pool are used, as in the reference.
// compact16: store the counts as 16-bit integers when all of them fit (otherwise the batch comes back 32-bit as usual)
SyncBatch parse_sync_file(const std::string &fname, int n_threads, SyncAlloc alloc = SyncAlloc(), bool compact16 = false);
// the same for a byte range [b, e) that starts at a line start (a chunk of a file streamed in pieces); expect_n > 0
// fixes the number of pools (0 = take it from the first data line of the range)
SyncBatch parse_sync_buffer(const char *b, const char *e, int n_threads, int expect_n, SyncAlloc alloc = SyncAlloc(),
                            bool compact16 = false);

extern const char ALLELES[7];

// The reference draws its folds from rand::thread_rng (cv.rs:39-43, penalise.rs:452-453), i.e. unrepeatably; here a
// seeded generator takes that place so that a run can be reproduced and tested.
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    std::vector<int64_t> permutation(int64_t n) { // Fisher-Yates
        std::vector<int64_t> v(n);
        for (int64_t i = 0; i < n; ++i) v[i] = i;
        for (int64_t i = n - 1; i > 0; --i) { const int64_t j = (int64_t)(next() % (uint64_t)(i + 1)); std::swap(v[i], v[j]); }
        return v;
    }
};

// k_split (cv.rs:15-49 / penalise.rs:428-459): group sizes s = floor(n / k) with k lowered until s >= 10 (k = 2 when
// n < 20); groups 0..k-1 of s members and what is left in group k, which no fold ever validates; `order[i]` picks the
// group of position i out of that list.  Returns the group of every position, and k.
std::vector<int32_t> k_split(int64_t n, int k_requested, const std::vector<int64_t> &order, int &k_out);

} // namespace pgh
