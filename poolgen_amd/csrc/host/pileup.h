// pileup.h -- samtools mpileup text -> sync lines, the `pileup2sync` front end (base/pileup.rs:11-371, :373-545).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "host_util.h"

namespace pgh {

struct PileupFilter {            // the FilterStats fields PileupLine::filter reads (pileup.rs:239-337)
    bool remove_ns = true;                 // !--keep-ns
    bool keep_lowercase_reference = false; // --keep-lowercase-reference (pileup.rs:280-299)
    double max_base_error_rate = 0.01;
    uint64_t min_coverage_depth = 1;
    double min_coverage_breadth = 1.0;
    double min_allele_frequency = 0.001;
    std::vector<double> pool_sizes;        // normalised to sum 1 (phen.rs:83-84)
};

// One converter per run: the per-byte decisions (base code -> allele, phred -> "below the error rate?") are
// table look-ups.  convert() handles one line [b, e) without its newline: appends the sync line to `out` and
// returns true when the locus is kept, returns false when the reference returns None, throws
// std::runtime_error where the reference panics (String::lparse errors under .expect(), pileup.rs:425-431).
class PileupConverter {
public:
    struct Locus {                      // what a kept line decodes to
        const char *chrom = nullptr;    // into the line
        size_t chrom_len = 0;
        uint64_t pos = 0;
        unsigned char ref = 0;
        int n = 0;
        const uint64_t *counts = nullptr; // n x 6 in pileup_to_sync's column order A,T,C,G,D,N (thread-local storage)
    };
    explicit PileupConverter(const PileupFilter &f);
    bool decode(const char *b, const char *e, Locus &out) const;           // lparse + filter + to_counts
    bool convert(const char *b, const char *e, std::string &out) const;    // decode + the sync text line
    int pools() const { return (int)f_.pool_sizes.size(); }

private:
    PileupFilter f_;
    uint32_t min_breadth_;
    bool low_quality_[256];   // 10^(-(q-33)/10) > max_base_error_rate
    unsigned char base_[256]; // read code -> allele byte; 0 = "the reference allele"
    unsigned char column_[256]; // allele byte -> count column (A,T,C,G,D,N = 0..5), through the lower-case remap when asked for
    unsigned char special_[128]; // 1 for the read codes that are not a base: + - ^ $
};

// Whole file with `n_threads` workers over byte ranges split at line starts; loci in file order; header
// "#chr\tpos\tref\t<pool names>" (pileup.rs:519-521); refuses to overwrite (create_new, :511-517).  Returns the
// number of loci written.
int64_t pileup_to_sync_file(const std::string &fname, const std::vector<std::string> &pool_names,
                            const PileupFilter &f, const std::string &out_fname, int n_threads);

// The same conversion straight into a counts batch (what parse_sync_file would return for the sync file that
// pileup_to_sync_file writes, without the text round trip): column POSITIONS are kept, i.e. position 4 holds the
// deletion count and position 5 the N count, exactly what the sync reader finds there (and labels N and D,
// base/sync.rs:134 vs pileup.rs:184).
SyncBatch parse_pileup_file(const std::string &fname, int n_threads, const PileupFilter &f, const SyncAlloc &alloc);
SyncBatch parse_pileup_buffer(const char *b, const char *e, int n_threads, const PileupFilter &f, const SyncAlloc &alloc);

} // namespace pgh
