// host_util.cpp -- see host_util.h.  Compiled with -ffp-contract=off: the filter arithmetic must
// round exactly like the reference (multiply, then add).
#include "host_util.h"
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace pgh {

const char ALLELES[7] = "ATCGND";

std::string rust_display(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed); // shortest round-trip, no exponent
    return std::string(buf, r.ptr);
}

double sensible_round(double x, int n_digits) {
    const std::string e = "1e" + std::to_string(n_digits);
    const double factor = std::strtod(e.c_str(), nullptr);
    return std::round(x * factor) / factor; // f64::round: half away from zero
}

std::string roundup_own(double x, int n_digits) {
    const std::string s = rust_display(x);
    if ((int)s.size() < n_digits) return s;
    return rust_display(sensible_round(x, n_digits));
}

static std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

static std::vector<std::string> split(const std::string &s, const std::string &delim) {
    std::vector<std::string> out;
    size_t start = 0;
    for (;;) {
        const size_t p = s.find(delim, start);
        if (p == std::string::npos) { out.push_back(s.substr(start)); break; }
        out.push_back(s.substr(start, p - start));
        start = p + delim.size();
    }
    return out;
}

Phen parse_phen(const std::string &fname, const std::string &delim, int name_col, int size_col,
                const std::vector<int> &value_cols) {
    std::ifstream in(fname);
    if (!in) throw std::runtime_error("Input phenotype file not found: " + fname);
    Phen ph;
    ph.k = (int)value_cols.size();
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) throw std::runtime_error("empty line in phenotype file " + fname);
        if (line[0] == '#') continue; // phen.rs:47-50
        std::vector<std::string> v = split(line, delim);
        for (auto &x : v) x = trim(x);
        const int need = std::max({name_col, size_col, ph.k ? *std::max_element(value_cols.begin(), value_cols.end()) : 0});
        if ((int)v.size() <= need) throw std::runtime_error("phenotype file: too few columns in line: " + line);
        ph.pool_names.push_back(v[name_col]);
        char *end = nullptr;
        const double sz = std::strtod(v[size_col].c_str(), &end);
        if (v[size_col].empty() || *end != 0)
            throw std::runtime_error("T_T Pool sizes column (column index: " + std::to_string(size_col) +
                                     ") is not a valid number. Line: " + line + ".");
        ph.pool_sizes.push_back(sz);
        for (int j = 0; j < ph.k; ++j) {
            const std::string &t = v[value_cols[j]];
            if (t == "" || t == "NA" || t == "NAN" || t == "NaN" || t == "na" || t == "nan") {
                ph.phen.push_back(NAN); // phen.rs:68-75
            } else {
                const double y = std::strtod(t.c_str(), &end);
                if (*end != 0)
                    throw std::runtime_error("T_T Error parsing the phenotype file. The trait values specified cannot be casted into float64.");
                ph.phen.push_back(y);
            }
        }
    }
    ph.n = (int)ph.pool_names.size();
    double total = 0.0;
    for (double s : ph.pool_sizes) total = total + s;
    for (double &s : ph.pool_sizes) s /= total; // phen.rs:83-84
    return ph;
}

// ---- sync parsing -----------------------------------------------------------------------------
static bool parse_u64(const char *b, const char *e, uint64_t &out) {
    if (b == e) return false;
    auto r = std::from_chars(b, e, out);
    return r.ec == std::errc() && r.ptr == e;
}

// one line [b, e) without the trailing newline; returns pools parsed, 0 for comments
static int parse_line(const char *b, const char *e, std::string &chrom, uint64_t &pos,
                      std::vector<uint32_t> &counts, int expect_n) {
    if (e > b && e[-1] == '\r') --e;
    if (b == e) throw std::runtime_error("empty line in sync file");
    if (*b == '#') return 0;
    int field = 0, n = 0;
    const char *p = b;
    while (p <= e) {
        const char *t = (const char *)std::memchr(p, '\t', e - p);
        const char *fe = t ? t : e;
        if (field == 0) chrom.assign(p, fe);
        else if (field == 1) {
            if (!parse_u64(p, fe, pos))
                throw std::runtime_error("Please check format of the file: position is not and integer.");
        } else if (field >= 3) {
            const char *q = p;
            for (int j = 0; j < 6; ++j) {
                const char *c = (const char *)std::memchr(q, ':', fe - q);
                const char *ce = c ? c : fe;
                uint64_t v;
                if (!parse_u64(q, ce, v) || v > 0xFFFFFFFFull)
                    throw std::runtime_error("Please check the input sync file as the allele counts are not valid integers.");
                counts.push_back((uint32_t)v);
                if (!c && j < 5)
                    throw std::runtime_error("Please check the input sync file as the allele counts are not valid integers.");
                q = c ? c + 1 : fe;
            }
            ++n;
        }
        ++field;
        if (!t) break;
        p = t + 1;
    }
    if (expect_n > 0 && n != expect_n) throw std::runtime_error("sync file: inconsistent number of pools");
    return n;
}

SyncBatch parse_sync_file(const std::string &fname, int n_threads) {
    std::ifstream in(fname, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("The input file: " + fname + " does not exist. Please make sure you are entering the correct filename and/or the correct path.");
    const size_t sz = (size_t)in.tellg();
    std::string buf(sz, '\0');
    in.seekg(0);
    in.read(&buf[0], sz);
    if (n_threads < 1) n_threads = 1;
    // byte ranges split at line starts (helpers.rs:74-91)
    std::vector<size_t> cuts{0};
    for (int t = 1; t < n_threads; ++t) {
        size_t c = sz / n_threads * t;
        if (c <= cuts.back()) continue;
        const void *nl = std::memchr(buf.data() + c, '\n', sz - c);
        c = nl ? (const char *)nl - buf.data() + 1 : sz;
        if (c > cuts.back() && c < sz) cuts.push_back(c);
    }
    cuts.push_back(sz);
    const int parts = (int)cuts.size() - 1;
    std::vector<SyncBatch> out(parts);
    std::vector<std::string> err(parts);
    auto work = [&](int t) {
        try {
            SyncBatch &sb = out[t];
            const char *p = buf.data() + cuts[t], *end = buf.data() + cuts[t + 1];
            std::string chrom;
            while (p < end) {
                const char *nl = (const char *)std::memchr(p, '\n', end - p);
                const char *le = nl ? nl : end;
                uint64_t pos = 0;
                const int n = parse_line(p, le, chrom, pos, sb.counts, sb.n);
                if (n > 0) {
                    sb.n = n;
                    sb.chrom.push_back(chrom);
                    sb.pos.push_back(pos);
                }
                p = nl ? nl + 1 : end;
            }
        } catch (const std::exception &e) { err[t] = e.what(); }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < parts; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    SyncBatch all;
    for (int t = 0; t < parts; ++t) {
        if (!err[t].empty()) throw std::runtime_error(err[t]);
        if (out[t].size() == 0) continue;
        if (all.n && out[t].n != all.n) throw std::runtime_error("sync file: inconsistent number of pools");
        all.n = out[t].n;
        all.chrom.insert(all.chrom.end(), out[t].chrom.begin(), out[t].chrom.end());
        all.pos.insert(all.pos.end(), out[t].pos.begin(), out[t].pos.end());
        all.counts.insert(all.counts.end(), out[t].counts.begin(), out[t].counts.end());
    }
    return all;
}

} // namespace pgh
