// operators.cpp -- see operators.h
#include "operators.h"
#include "host_util.h"
#include <stdexcept>

namespace pgh {

void format_locus_rows(int mode, const std::string &chromosome, uint64_t position, int n_out, const int32_t *ids, const double *mean_freq,
                       const double *stat, const double *pval, int k, std::string &line, size_t slot_stride) {
    if (n_out <= 0) return;
    const size_t S = slot_stride; // the library's arrays are slot-major: slot i of this locus is S elements (S * k for stat / pval) further on
    if (mode == 0) { // chisq_test.rs:37-45
        line += chromosome; line.push_back(','); line += std::to_string(position); line.push_back(',');
        for (int j = 0; j < n_out && j < PG_MAX_OUT; ++j) line.push_back(ALLELES[ids[j * S]]);
        line.push_back(',');
        append_roundup_own(line, stat[0], 6); line.push_back(',');
        append_rust_display(line, pval[0]); line.push_back('\n');
        return;
    }
    for (int i = 0; i < n_out; ++i)
        for (int j = 0; j < k; ++j) {
            const size_t e = (size_t)i * S * k + j;
            line += chromosome; line.push_back(','); line += std::to_string(position); line.push_back(',');
            line.push_back(ALLELES[ids[i * S]]); line.push_back(',');
            if (mode == 2) append_roundup_own(line, mean_freq[i * S], 8); // ols.rs:263-271
            else append_rust_display(line, mean_freq[i * S]);             // correlation_test.rs:117-124
            line += ",Pheno_"; line += std::to_string(j); line.push_back(',');
            append_roundup_own(line, stat[e], 6); line.push_back(',');
            if (mode == 2) append_roundup_own(line, pval[e], 12);
            else append_rust_display(line, pval[e]);
            line.push_back('\n');
        }
}

std::optional<std::string> Operators::run(int mode, const LocusCounts &lc, const double *Y, int k, const FilterStats &f) const {
    const int a = (int)lc.alleles_vector.size();
    if (a == 0 || lc.matrix.size() % (size_t)a != 0) return std::nullopt;
    const int n = (int)(lc.matrix.size() / (size_t)a);
    if (n != (int)f.pool_sizes.size()) return std::nullopt; // the filter's own check fails (sync.rs:254-257)
    // the labelled columns -> the six sync columns in the reader's order A,T,C,G,N,D (sync.rs:134); an absent allele is a zero column
    std::vector<uint32_t> counts((size_t)n * 6, 0);
    for (int j = 0; j < a; ++j) {
        int col = -1;
        for (int c = 0; c < 6; ++c)
            if (lc.alleles_vector[j].size() == 1 && lc.alleles_vector[j][0] == ALLELES[c]) col = c;
        if (col < 0) return std::nullopt;
        for (int i = 0; i < n; ++i) {
            const uint64_t v = lc.matrix[(size_t)i * a + j];
            if (v > 0xFFFFFFFFull) return std::nullopt;
            counts[(size_t)i * 6 + col] = (uint32_t)v;
        }
    }
    pg_filter flt{};
    flt.remove_ns = f.remove_ns ? 1 : 0;
    flt.min_coverage_depth = f.min_coverage_depth;
    flt.min_allele_frequency = f.min_allele_frequency;
    flt.max_missingness_rate = f.max_missingness_rate;
    int32_t n_out = 0, ids[PG_MAX_OUT] = {0};
    double mf[PG_MAX_OUT] = {0};
    std::vector<double> stat((size_t)PG_MAX_OUT * (k > 0 ? k : 1)), pv(stat.size());
    int rc;
    if (mode == 0) rc = pg_chisq_batch(ctx_, counts.data(), 1, n, f.pool_sizes.data(), &flt, &n_out, ids, stat.data(), pv.data());
    else if (mode == 1) rc = pg_pearson_batch(ctx_, counts.data(), 1, n, f.pool_sizes.data(), &flt, Y, k, &n_out, ids, mf, stat.data(), pv.data());
    else rc = pg_ols_iter_batch(ctx_, counts.data(), 1, n, f.pool_sizes.data(), &flt, Y, k, &n_out, ids, mf, stat.data(), pv.data());
    if (rc != PG_OK) throw std::runtime_error(pg_last_error(ctx_)); // a broken device, not a property of the locus
    if (n_out <= 0) return std::nullopt;
    std::string out;
    format_locus_rows(mode, lc.chromosome, lc.position, n_out, ids, mf, stat.data(), pv.data(), k, out);
    return out;
}

std::optional<std::string> Operators::chisq(LocusCounts &locus, const FilterStats &f) const { return run(0, locus, nullptr, 0, f); }

std::optional<std::string> Operators::correlation(LocusCountsAndPhenotypes &l, const FilterStats &f) const {
    const size_t n = l.locus_counts.alleles_vector.empty() ? 0 : l.locus_counts.matrix.size() / l.locus_counts.alleles_vector.size();
    if (n == 0 || l.phenotypes.size() % n != 0) return std::nullopt;
    return run(1, l.locus_counts, l.phenotypes.data(), (int)(l.phenotypes.size() / n), f);
}

std::optional<std::string> Operators::ols_iterate(LocusCountsAndPhenotypes &l, const FilterStats &f) const {
    const size_t n = l.locus_counts.alleles_vector.empty() ? 0 : l.locus_counts.matrix.size() / l.locus_counts.alleles_vector.size();
    if (n == 0 || l.phenotypes.size() % n != 0) return std::nullopt;
    return run(2, l.locus_counts, l.phenotypes.data(), (int)(l.phenotypes.size() / n), f);
}

} // namespace pgh
