// host_util.cpp -- see host_util.h.  Compiled with -ffp-contract=off: the filter arithmetic must
// round exactly like the reference (multiply, then add).
#include "host_util.h"
#include <cctype>
#include <climits>
#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>
#include <utility>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pgh {

const char ALLELES[7] = "ATCGND";

std::string rust_display(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed); // shortest round-trip, no exponent
    return std::string(buf, r.ptr);
}

void append_rust_display(std::string &out, double x) {
    if (std::isnan(x)) { out += "NaN"; return; }
    if (std::isinf(x)) { out += x > 0 ? "inf" : "-inf"; return; }
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed);
    out.append(buf, r.ptr);
}

double sensible_round(double x, int n_digits) {
    // 10^n: exact in binary64 up to n = 22, so the table equals what parsing "1e<n>" gives (the reference parses the string)
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                   1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    double factor;
    if (n_digits >= 0 && n_digits <= 22) factor = P10[n_digits];
    else {
        const std::string e = "1e" + std::to_string(n_digits);
        factor = std::strtod(e.c_str(), nullptr);
    }
    return std::round(x * factor) / factor; // f64::round: half away from zero
}

std::string roundup_own(double x, int n_digits) {
    const std::string s = rust_display(x);
    if ((int)s.size() < n_digits) return s;
    return rust_display(sensible_round(x, n_digits));
}

void append_roundup_own(std::string &out, double x, int n_digits) {
    const size_t at = out.size();
    append_rust_display(out, x);
    if ((int)(out.size() - at) < n_digits) return; // shorter than n_digits characters: printed as it is (helpers.rs:112-115)
    out.resize(at);
    append_rust_display(out, sensible_round(x, n_digits));
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

bool parse_f64_strict(const std::string &s, double &out) {
    size_t i = 0;
    const size_t n = s.size();
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    auto lower_eq = [&](const char *w) {
        size_t k = 0;
        for (; w[k]; ++k)
            if (i + k >= n || std::tolower((unsigned char)s[i + k]) != w[k]) return false;
        return i + k == n;
    };
    if (lower_eq("inf") || lower_eq("infinity") || lower_eq("nan")) {
        out = std::strtod(s.c_str(), nullptr);
        return true;
    }
    size_t digits = 0;
    while (i < n && std::isdigit((unsigned char)s[i])) { ++i; ++digits; }
    if (i < n && s[i] == '.') {
        ++i;
        while (i < n && std::isdigit((unsigned char)s[i])) { ++i; ++digits; }
    }
    if (digits == 0) return false;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
        size_t ed = 0;
        while (i < n && std::isdigit((unsigned char)s[i])) { ++i; ++ed; }
        if (ed == 0) return false;
    }
    if (i != n) return false;
    out = std::strtod(s.c_str(), nullptr);
    return true;
}

bool parse_u64_strict(const std::string &s, uint64_t &out) {
    size_t i = (!s.empty() && s[0] == '+') ? 1 : 0;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (!std::isdigit((unsigned char)s[i])) return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

bool parse_i64_strict(const std::string &s, int64_t &out) {
    const bool neg = !s.empty() && s[0] == '-';
    uint64_t v = 0;
    if (!parse_u64_strict(neg ? s.substr(1) : s, v) || (neg && !s.empty() && s.size() > 1 && s[1] == '+')) return false;
    if (v > (uint64_t)INT64_MAX + (neg ? 1 : 0)) return false;
    out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return true;
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
        double sz = 0.0;
        if (!parse_f64_strict(v[size_col], sz))
            throw std::runtime_error("T_T Pool sizes column (column index: " + std::to_string(size_col) +
                                     ") is not a valid number. Line: " + line + ".");
        ph.pool_sizes.push_back(sz);
        for (int j = 0; j < ph.k; ++j) {
            const std::string &t = v[value_cols[j]];
            if (t == "" || t == "NA" || t == "NAN" || t == "NaN" || t == "na" || t == "nan") {
                ph.phen.push_back(NAN); // phen.rs:68-75
            } else {
                double y = 0.0;
                if (!parse_f64_strict(t, y))
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
SyncBatch &SyncBatch::operator=(SyncBatch &&o) noexcept {
    if (this != &o) {
        if (void *q = counts ? (void *)counts : (void *)counts16) { if (release) release(q); else std::free(q); }
        n = o.n; L = o.L;
        chrom_id = std::move(o.chrom_id); chrom_names = std::move(o.chrom_names); pos = std::move(o.pos);
        counts = o.counts; counts16 = o.counts16; release = std::move(o.release);
        o.counts = nullptr; o.counts16 = nullptr; o.L = 0;
    }
    return *this;
}
SyncBatch::~SyncBatch() {
    if (void *q = counts ? (void *)counts : (void *)counts16) { if (release) release(q); else std::free(q); }
}

MappedFile::MappedFile(const std::string &fname) {
    const std::string notfound = "The input file: " + fname + " does not exist. Please make sure you are entering the correct filename and/or the correct path.";
    const int fd = ::open(fname.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error(notfound);
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error(notfound); }
    n_ = (size_t)st.st_size;
    if (n_) {
        void *m = ::mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); throw std::runtime_error("cannot map " + fname); }
        p_ = (const char *)m;
    }
    ::close(fd);
}
MappedFile::~MappedFile() { if (p_ && n_) ::munmap(const_cast<char *>(p_), n_); }
std::vector<size_t> MappedFile::cuts(size_t pieces) const {
    if (pieces < 1) pieces = 1;
    std::vector<size_t> c{0};
    for (size_t t = 1; t < pieces; ++t) {
        size_t x = n_ / pieces * t;
        if (x <= c.back()) continue;
        const void *nl = std::memchr(p_ + x, '\n', n_ - x);
        x = nl ? (size_t)((const char *)nl - p_) + 1 : n_;
        if (x > c.back() && x < n_) c.push_back(x);
    }
    c.push_back(n_);
    return c;
}

namespace {

inline const char *line_end(const char *p, const char *end) {
    const void *nl = std::memchr(p, '\n', (size_t)(end - p));
    return nl ? (const char *)nl : end;
}

// A run of decimal digits -> value; false when there is no digit (Rust's `parse::<u64>` also accepts
// one leading '+', so does this).
inline bool digits(const char *&p, const char *e, uint64_t &v) {
    if (p < e && *p == '+') ++p;
    const char *b = p;
    uint64_t x = 0;
    while (p < e) {
        const unsigned c = (unsigned)(unsigned char)*p - (unsigned)'0';
        if (c > 9u) break;
        x = x * 10u + c;
        ++p;
    }
    v = x;
    return p != b && p - b <= 19;
}

struct ThreadOut {
    std::vector<std::string> names; // chromosome names met by this worker, in order
    int64_t lines = 0;              // data lines written
    bool overflow16 = false;        // a count above 65535 met while storing 16-bit counts
    std::string err;
};

// Parses the data lines of [b, e) into counts[(base + i) * n * 6 ...], chrom_local, pos.  n == 0: only
// count the candidate lines (non-empty, not starting with '#') and return that count.
template <typename CT>
int64_t parse_range(const char *b, const char *e, int n, CT *counts, int32_t *chrom_local, uint64_t *pos,
                    ThreadOut &out) {
    int64_t li = 0;
    const char *p = b;
    int last = -1;
    while (p < e) {
        const char *le = line_end(p, e);
        const char *next = le < e ? le + 1 : e;
        const char *q = le;
        if (q > p && q[-1] == '\r') --q;
        if (q == p) throw std::runtime_error("empty line in sync file"); // the reference indexes byte 0 of the line
        if (*p == '#') { p = next; continue; }
        if (n == 0) { ++li; p = next; continue; }
        // chromosome
        const char *t = (const char *)std::memchr(p, '\t', (size_t)(q - p));
        if (!t) throw std::runtime_error("sync file: a line has fewer than four tab-separated fields");
        const size_t clen = (size_t)(t - p);
        if (last < 0 || out.names[last].size() != clen || std::memcmp(out.names[last].data(), p, clen) != 0) {
            last = -1;
            for (size_t i = 0; i < out.names.size(); ++i)
                if (out.names[i].size() == clen && std::memcmp(out.names[i].data(), p, clen) == 0) { last = (int)i; break; }
            if (last < 0) { out.names.emplace_back(p, clen); last = (int)out.names.size() - 1; }
        }
        // position: not an integer -> the line is skipped (ErrorKind::Other, see the header)
        const char *c = t + 1;
        uint64_t pv = 0;
        const bool pos_ok = digits(c, q, pv) && c < q && *c == '\t';
        if (!pos_ok) {
            const char *t2 = (const char *)std::memchr(t + 1, '\t', (size_t)(q - t - 1));
            if (!t2) throw std::runtime_error("sync file: a line has fewer than four tab-separated fields");
            p = next;
            continue;
        }
        // reference allele: ignored
        const char *t3 = (const char *)std::memchr(c + 1, '\t', (size_t)(q - c - 1));
        if (!t3) throw std::runtime_error("sync file: a line has fewer than four tab-separated fields");
        c = t3 + 1;
        CT *dst = counts + (size_t)li * n * 6;
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < 6; ++j) {
                uint64_t v;
                if (!digits(c, q, v) || v > 0xFFFFFFFFull)
                    throw std::runtime_error("Please check the input sync file as the allele counts are not valid integers.");
                if (sizeof(CT) == 2 && v > 0xFFFFull) { out.overflow16 = true; throw std::runtime_error("count above 65535"); }
                dst[i * 6 + j] = (CT)v;
                if (j < 5) {
                    if (c >= q || *c != ':')
                        throw std::runtime_error("Please check the input sync file as the allele counts are not valid integers.");
                    ++c;
                }
            }
            // anything after the sixth count of a pool is ignored up to the next tab, as long as it is a
            // well-formed ':'-separated list of integers (the reference parses, then uses the first six)
            while (c < q && *c == ':') {
                ++c;
                uint64_t v;
                if (!digits(c, q, v))
                    throw std::runtime_error("Please check the input sync file as the allele counts are not valid integers.");
            }
            if (i + 1 < n) {
                if (c >= q || *c != '\t') throw std::runtime_error("sync file: inconsistent number of pools");
                ++c;
            }
        }
        if (c != q) {
            if (*c == '\t') throw std::runtime_error("sync file: inconsistent number of pools");
            throw std::runtime_error("Please check the input sync file as the allele counts are not valid integers.");
        }
        chrom_local[li] = last;
        pos[li] = pv;
        ++li;
        p = next;
    }
    return li;
}

} // namespace

SyncBatch parse_sync_file(const std::string &fname, int n_threads, SyncAlloc alloc, bool compact16) {
    const MappedFile mf(fname);
    return parse_sync_buffer(mf.data(), mf.data() + mf.size(), n_threads, 0, std::move(alloc), compact16);
}

SyncBatch parse_sync_buffer(const char *bb, const char *be, int n_threads, int expect_n, SyncAlloc alloc, bool compact16) {
    const bool timing = std::getenv("PGH_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_last = now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t = now();
        std::fprintf(stderr, "parse_sync_file: %-12s %.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
        t_last = t;
    };
    const char *buf = bb;
    const size_t sz = (size_t)(be - bb);
    const std::string fname = "the sync input";
    if (n_threads < 1) n_threads = 1;
    // byte ranges split at line starts (helpers.rs:74-91)
    std::vector<size_t> cuts{0};
    for (int t = 1; t < n_threads; ++t) {
        size_t c = sz / n_threads * t;
        if (c <= cuts.back()) continue;
        const void *nl = std::memchr(buf + c, '\n', sz - c);
        c = nl ? (size_t)((const char *)nl - buf) + 1 : sz;
        if (c > cuts.back() && c < sz) cuts.push_back(c);
    }
    cuts.push_back(sz);
    const int parts = (int)cuts.size() - 1;
    SyncBatch sb;
    // pools per line: from the first data line
    {
        const char *p = buf, *end = buf + sz;
        while (p < end) {
            const char *le = line_end(p, end);
            if (le > p && *p != '#') {
                int tabs = 0;
                for (const char *c = p; c < le; ++c) tabs += (*c == '\t');
                if (tabs < 3) throw std::runtime_error("sync file: a line has fewer than four tab-separated fields");
                sb.n = tabs - 2;
                if (expect_n > 0 && sb.n != expect_n) throw std::runtime_error("sync file: inconsistent number of pools");
                break;
            }
            if (le == p) throw std::runtime_error("empty line in sync file");
            p = le < end ? le + 1 : end;
        }
    }
    if (sb.n == 0) return sb;
    std::vector<ThreadOut> out(parts);
    std::vector<int64_t> cand(parts, 0), base(parts + 1, 0);
    auto run_all = [&](auto &&fn) -> bool {
        std::vector<std::thread> th;
        for (int t = 0; t < parts; ++t)
            th.emplace_back([&, t] {
                try { fn(t); } catch (const std::exception &e) { out[t].err = e.what(); }
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < parts; ++t)
            if (out[t].overflow16) return false; // asked for 16-bit counts and one does not fit: the caller starts over
        for (int t = 0; t < parts; ++t)
            if (!out[t].err.empty()) throw std::runtime_error(out[t].err);
        return true;
    };
    // pass 1: candidate lines per range, so that every worker can write straight into its slice
    run_all([&](int t) { cand[t] = parse_range<uint32_t>(buf + cuts[t], buf + cuts[t + 1], 0, nullptr, nullptr, nullptr, out[t]); });
    lap("count lines");
    for (int t = 0; t < parts; ++t) base[t + 1] = base[t] + cand[t];
    const int64_t Lcand = base[parts];
    sb.chrom_id.resize(Lcand);
    sb.pos.resize(Lcand);
    auto allocate = [&](size_t elem) -> void * {
        const size_t bytes = elem * (size_t)Lcand * sb.n * 6;
        void *q = alloc.alloc ? alloc.alloc(bytes ? bytes : 1) : std::malloc(bytes ? bytes : 1);
        if (alloc.alloc) sb.release = alloc.release ? alloc.release : [](void *) {};
        if (!q) throw std::runtime_error("out of memory for the allele counts of " + fname);
        return q;
    };
    bool done = false;
    if (compact16) {
        sb.counts16 = static_cast<uint16_t *>(allocate(sizeof(uint16_t)));
        lap("allocate");
        done = run_all([&](int t) {
            out[t].lines = parse_range<uint16_t>(buf + cuts[t], buf + cuts[t + 1], sb.n, sb.counts16 + (size_t)base[t] * sb.n * 6,
                                                 sb.chrom_id.data() + base[t], sb.pos.data() + base[t], out[t]);
        });
        if (!done) { // a count above 65535: 32-bit after all (the same allocator hands out a buffer of the right size)
            if (sb.release) sb.release(sb.counts16); else std::free(sb.counts16);
            sb.counts16 = nullptr;
            for (auto &o : out) o = ThreadOut();
        }
    }
    if (!done) {
        sb.counts = static_cast<uint32_t *>(allocate(sizeof(uint32_t)));
        lap("allocate");
        run_all([&](int t) {
            out[t].lines = parse_range<uint32_t>(buf + cuts[t], buf + cuts[t + 1], sb.n, sb.counts + (size_t)base[t] * sb.n * 6,
                                                 sb.chrom_id.data() + base[t], sb.pos.data() + base[t], out[t]);
        });
    }
    lap("parse");
    // chromosome names: worker-local ids -> global ids in order of first appearance
    std::vector<std::vector<int32_t>> remap(parts);
    for (int t = 0; t < parts; ++t)
        for (const std::string &nm : out[t].names) {
            int g = -1;
            for (size_t i = 0; i < sb.chrom_names.size(); ++i)
                if (sb.chrom_names[i] == nm) { g = (int)i; break; }
            if (g < 0) { sb.chrom_names.push_back(nm); g = (int)sb.chrom_names.size() - 1; }
            remap[t].push_back(g);
        }
    // lines skipped for a bad position leave holes at the end of a slice: close them (rare)
    int64_t w = 0;
    for (int t = 0; t < parts; ++t) {
        const int64_t r0 = base[t], cnt = out[t].lines;
        for (int64_t i = 0; i < cnt; ++i) sb.chrom_id[r0 + i] = remap[t][sb.chrom_id[r0 + i]];
        if (w != r0 && cnt > 0) {
            if (sb.counts16) std::memmove(sb.counts16 + (size_t)w * sb.n * 6, sb.counts16 + (size_t)r0 * sb.n * 6, sizeof(uint16_t) * (size_t)cnt * sb.n * 6);
            else std::memmove(sb.counts + (size_t)w * sb.n * 6, sb.counts + (size_t)r0 * sb.n * 6, sizeof(uint32_t) * (size_t)cnt * sb.n * 6);
            std::memmove(sb.chrom_id.data() + w, sb.chrom_id.data() + r0, sizeof(int32_t) * cnt);
            std::memmove(sb.pos.data() + w, sb.pos.data() + r0, sizeof(uint64_t) * cnt);
        }
        w += cnt;
    }
    lap("merge");
    sb.L = w;
    sb.chrom_id.resize(w);
    sb.pos.resize(w);
    return sb;
}

std::vector<int32_t> k_split(int64_t n, int k, const std::vector<int64_t> &order, int &k_out) {
    if ((k >= n) | (n <= 2))
        throw std::runtime_error("The number of splits, i.e. k, needs to be less than the number of pools, n, and n > 2. We are aiming for fold sizes of 10 or greater.");
    int64_t s = n / k;
    while (s < 10) {
        if (n < 20) { k = 2; s = n / k; break; }
        k -= 1;
        s = n / k;
    }
    std::vector<int32_t> g;
    for (int x = 0; x < k; ++x) g.insert(g.end(), (size_t)s, x);
    for (int64_t i = 0; i < n - s; ++i) g.push_back(k); // as written: more entries than ever used
    std::vector<int32_t> out(n);
    for (int64_t i = 0; i < n; ++i) {
        if (order[i] < 0 || (size_t)order[i] >= g.size()) throw std::runtime_error("k_split: index out of bounds"); // a panic there
        out[i] = g[order[i]];
    }
    k_out = k;
    return out;
}

} // namespace pgh
