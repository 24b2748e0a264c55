// pileup.cpp -- see pileup.h.  Compiled with -ffp-contract=off: the allele-frequency threshold is a sum of
// separately rounded products, as in the reference.
#include "pileup.h"
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pgh {

namespace {
[[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("Input pileup file error: ") + what); }

// decimal u64 as Rust's parse::<u64> takes it (digits, one optional leading '+')
inline bool take_u64(const char *b, const char *e, uint64_t &v) {
    if (b < e && *b == '+') ++b;
    if (b == e || e - b > 19) return false;
    uint64_t x = 0;
    for (; b < e; ++b) {
        const unsigned c = (unsigned)(unsigned char)*b - (unsigned)'0';
        if (c > 9u) return false;
        x = x * 10u + c;
    }
    v = x;
    return true;
}
inline const char *field_end(const char *p, const char *e) {
    const void *t = std::memchr(p, '\t', (size_t)(e - p));
    return t ? (const char *)t : e;
}
} // namespace

PileupConverter::PileupConverter(const PileupFilter &f) : f_(f) {
    min_breadth_ = (uint32_t)std::ceil(f.min_coverage_breadth * (double)f.pool_sizes.size()); // pileup.rs:267
    for (int q = 0; q < 256; ++q)
        low_quality_[q] = q >= 33 && std::pow(10.0, -((double)q - 33.0) / 10.0) > f.max_base_error_rate; // :253-256
    for (int c = 0; c < 256; ++c) base_[c] = 78; // anything else is an N (:110-121)
    base_[','] = 0; base_['.'] = 0;
    base_['A'] = base_['a'] = 65; base_['T'] = base_['t'] = 84; base_['C'] = base_['c'] = 67;
    base_['G'] = base_['g'] = 71; base_['*'] = 68;
    for (int c = 0; c < 128; ++c) special_[c] = (c == '+' || c == '-' || c == '^' || c == '$') ? 1 : 0;
    for (int c = 0; c < 256; ++c) column_[c] = 5; // to_counts (:181-188): A,T,C,G,D, everything else under N
    column_[65] = 0; column_[84] = 1; column_[67] = 2; column_[71] = 3; column_[68] = 4;
    if (f.keep_lowercase_reference) {
        // :280-299 remaps the codes AFTER the N removal and the coverage test and before to_counts: either case of
        // A/T/C/G is the base, '*' is D, everything else -- the 'D' lparse made of '*' included -- is N.  The only codes
        // that can still be anything but A/T/C/G/D/N are copies of the reference allele, so the remap folds into the
        // byte -> column table.
        for (int c = 0; c < 256; ++c) column_[c] = 5;
        column_['A'] = column_['a'] = 0; column_['T'] = column_['t'] = 1; column_['C'] = column_['c'] = 2;
        column_['G'] = column_['g'] = 3; column_['*'] = 4;
    }
}

bool PileupConverter::decode(const char *b, const char *e, Locus &L) const {
    // ---- String::lparse (pileup.rs:11-155): any defect of the line's format is fatal -------------------
    const char *t0 = field_end(b, e);
    if (t0 == e) bad("a line has fewer than three tab-separated fields");
    const char *t1 = field_end(t0 + 1, e);
    if (t1 == e) bad("a line has fewer than three tab-separated fields");
    uint64_t pos;
    if (!take_u64(t0 + 1, t1, pos)) bad("position is not a valid integer (i.e. u64).");
    const char *t2 = field_end(t1 + 1, e);
    if (t2 - (t1 + 1) != 1 || (unsigned char)t1[1] >= 128) bad("the reference allele is not a valid single character.");
    const unsigned char ref = (unsigned char)t1[1];

    thread_local std::vector<uint64_t> counts; // n x 6, A,T,C,G,D,N
    counts.clear();
    bool phred_out_of_bounds = false;
    int n = 0;
    const char *p = t2;
    while (p < e) { // p at the tab in front of a pool's coverage field
        const char *cb = p + 1;
        const char *ce = field_end(cb, e);
        uint64_t cov;
        if (!take_u64(cb, ce, cov)) bad("coverage field/s is/are not valid integer/s (i.e. u64).");
        if (ce == e) bad("the coverages, number of read alleles and read qualities do not match (ragged pool).");
        const char *rb = ce + 1, *re = field_end(rb, e);
        if (re == e) bad("the coverages, number of read alleles and read qualities do not match (ragged pool).");
        const char *qb = re + 1, *qe = field_end(qb, e);
        uint64_t cnt[6] = {0, 0, 0, 0, 0, 0};
        uint32_t lane_cnt[4][8] = {}; // four interleaved counter sets: consecutive reads mostly hit the SAME column
        if (cov > 0) {
            // read codes -> alleles: '+'/'-' <count> <count bases> and '^' <mapping quality> are skipped, '$' dropped.
            // Read j is paired with quality j on the fly (filter part 1, pileup.rs:247-265: a low-quality read
            // becomes an N, Ns are dropped when remove_ns); the length checks of lparse follow the loop.
            const size_t nq = (size_t)(qe - qb);
            size_t j = 0;
            const char *c = rb;
            while (c < re) {
                const unsigned char ch = (unsigned char)*c++;
                if (ch > 'z' || special_[ch] == 0) goto plain_base; // the common case first
                if (ch == '+' || ch == '-') {
                    if (c >= re) break; // the field ends inside the indel marker: nothing more to read
                    if ((unsigned)(unsigned char)*c - (unsigned)'0' > 9u)
                        bad("codes for insertions and deletion must be integers after '+' and '-'.");
                    uint64_t len = (uint64_t)(*c++ - '0');
                    if (len == 0) {
                        // "+0": the reference keeps waiting for a first non-zero digit
                        while (c < re && len == 0) {
                            if ((unsigned)(unsigned char)*c - (unsigned)'0' > 9u)
                                bad("codes for insertions and deletion must be integers after '+' and '-'.");
                            len = (uint64_t)(*c++ - '0');
                        }
                        if (len == 0) break;
                    }
                    while (c < re && (unsigned)(unsigned char)*c - (unsigned)'0' <= 9u) len = len * 10u + (uint64_t)(*c++ - '0');
                    // the first non-digit ends the number AND is the first of the `len` bases to drop
                    const uint64_t room = (uint64_t)(re - c);
                    c += len < room ? len : room;
                    continue;
                }
                if (ch == '^') { if (c < re) ++c; continue; }
                if (ch == '$') continue;
            plain_base: {
                unsigned char a = base_[ch];
                if (!a) a = ref;
                if (j < nq) {
                    const unsigned char q = (unsigned char)qb[j];
                    if (q < 33) phred_out_of_bounds = true; // Err("Phred score out of bounds.") -> None, below
                    if (low_quality_[q]) a = 78;
                    const unsigned col = (f_.remove_ns && a == 78) ? 6u : column_[a]; // slot 6 = dropped
                    lane_cnt[j & 3][col] += 1;
                }
                ++j;
            }
            }
            if (j != cov || nq != cov) bad("the coverages, number of read alleles and read qualities do not match.");
            for (int a = 0; a < 6; ++a) cnt[a] = (uint64_t)lane_cnt[0][a] + lane_cnt[1][a] + lane_cnt[2][a] + lane_cnt[3][a];
        }
        counts.insert(counts.end(), cnt, cnt + 6);
        ++n;
        p = qe;
    }
    // ---- PileupLine::filter (:239-337): every failure from here on is `None` for pileup_to_sync (:343-346) ---
    if (n != (int)f_.pool_sizes.size()) return false;
    if (phred_out_of_bounds) return false;
    uint32_t covered = 0;
    for (int i = 0; i < n && covered < min_breadth_; ++i) {
        uint64_t s = 0;
        for (int j = 0; j < 6; ++j) s += counts[(size_t)i * 6 + j];
        if (s >= f_.min_coverage_depth) ++covered;
    }
    if (covered != min_breadth_) return false;
    // minimum allele frequency, with the loop exactly as the reference wrote it (:311-331): a column that fails
    // is tested again against a smaller m, so only the FIRST tested columns ever decide
    {
        int m = 6, j = 1;
        while (j < m) {
            double q = 0.0;
            for (int i = 0; i < n; ++i) {
                uint64_t s = 0;
                for (int a = 0; a < 6; ++a) s += counts[(size_t)i * 6 + a];
                q += ((double)counts[(size_t)i * 6 + j] / (double)s) * f_.pool_sizes[i];
            }
            if ((q < f_.min_allele_frequency) | (q > (1.00 - f_.min_allele_frequency))) m -= 1;
            else j += 1;
        }
        if (m < 2) return false;
    }
    L.chrom = b; L.chrom_len = (size_t)(t0 - b); L.pos = pos; L.ref = ref; L.n = n; L.counts = counts.data();
    return true;
}

bool PileupConverter::convert(const char *b, const char *e, std::string &out) const {
    Locus L;
    if (!decode(b, e, L)) return false;
    // ---- pileup_to_sync (:348-371): chr, pos, ref, then A:T:C:G:D:N per pool -----------------------------
    out.append(L.chrom, L.chrom_len);
    out.push_back('\t');
    char num[24];
    auto put = [&](uint64_t v) { out.append(num, (size_t)(std::to_chars(num, num + sizeof num, v).ptr - num)); };
    put(L.pos);
    out.push_back('\t');
    out.push_back((char)L.ref);
    for (int i = 0; i < L.n; ++i)
        for (int j = 0; j < 6; ++j) {
            out.push_back(j == 0 ? '\t' : ':');
            put(L.counts[(size_t)i * 6 + j]);
        }
    out.push_back('\n');
    return true;
}

int64_t pileup_to_sync_file(const std::string &fname, const std::vector<std::string> &pool_names, const PileupFilter &f,
                            const std::string &out_fname, int n_threads) {
    const int fd = ::open(fname.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("The input file: " + fname + " does not exist. Please make sure you are entering the correct filename and/or the correct path.");
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat " + fname); }
    const size_t sz = (size_t)st.st_size;
    const char *buf = nullptr;
    if (sz) {
        void *m = ::mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); throw std::runtime_error("cannot map " + fname); }
        buf = (const char *)m;
    }
    ::close(fd);
    if (n_threads < 1) n_threads = 1;
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
    const PileupConverter conv(f);
    std::vector<std::string> text(parts), err(parts);
    std::vector<int64_t> kept(parts, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < parts; ++t)
        th.emplace_back([&, t] {
            try {
                const char *p = buf + cuts[t], *end = buf + cuts[t + 1];
                text[t].reserve((size_t)(end - p) / 3 + 4096); // sync text is a fraction of the pileup text
                while (p < end) {
                    const void *nl = std::memchr(p, '\n', (size_t)(end - p));
                    const char *le = nl ? (const char *)nl : end;
                    const char *q = le;
                    if (q > p && q[-1] == '\r') --q;
                    if (conv.convert(p, q, text[t])) ++kept[t];
                    p = nl ? le + 1 : end;
                }
            } catch (const std::exception &e) { err[t] = e.what(); }
        });
    for (auto &x : th) x.join();
    if (buf) ::munmap(const_cast<char *>(buf), sz);
    for (int t = 0; t < parts; ++t)
        if (!err[t].empty()) throw std::runtime_error(err[t]);
    const int ofd = ::open(out_fname.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0644);
    if (ofd < 0) throw std::runtime_error("Unable to create file: " + out_fname + " (it must not exist)");
    FILE *fo = ::fdopen(ofd, "w");
    std::string header = "#chr\tpos\tref";
    for (const std::string &nm : pool_names) { header.push_back('\t'); header += nm; }
    header.push_back('\n');
    std::fwrite(header.data(), 1, header.size(), fo);
    int64_t total = 0;
    for (int t = 0; t < parts; ++t) { std::fwrite(text[t].data(), 1, text[t].size(), fo); total += kept[t]; }
    std::fclose(fo);
    return total;
}

SyncBatch parse_pileup_file(const std::string &fname, int n_threads, const PileupFilter &f, const SyncAlloc &alloc) {
    const MappedFile mf(fname);
    return parse_pileup_buffer(mf.data(), mf.data() + mf.size(), n_threads, f, alloc);
}

SyncBatch parse_pileup_buffer(const char *bb, const char *be, int n_threads, const PileupFilter &f, const SyncAlloc &alloc) {
    const char *buf = bb;
    const size_t sz = (size_t)(be - bb);
    const std::string fname = "the pileup input";
    if (n_threads < 1) n_threads = 1;
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
    const PileupConverter conv(f);
    SyncBatch sb;
    sb.n = conv.pools();
    if (sb.n == 0 || sz == 0) return sb;
    // pass 1: lines per range (an upper bound of the loci kept), so that every worker owns a slice
    std::vector<int64_t> cand(parts, 0), base(parts + 1, 0), kept(parts, 0);
    std::vector<std::string> err(parts);
    std::vector<std::vector<std::string>> names(parts);
    auto run_all = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (int t = 0; t < parts; ++t)
            th.emplace_back([&, t] { try { fn(t); } catch (const std::exception &e) { err[t] = e.what(); } });
        for (auto &x : th) x.join();
        for (int t = 0; t < parts; ++t) if (!err[t].empty()) throw std::runtime_error(err[t]);
    };
    run_all([&](int t) {
        int64_t c = 0;
        for (const char *p = buf + cuts[t], *end = buf + cuts[t + 1]; p < end;) {
            const void *nl = std::memchr(p, '\n', (size_t)(end - p));
            ++c;
            p = nl ? (const char *)nl + 1 : end;
        }
        cand[t] = c;
    });
    for (int t = 0; t < parts; ++t) base[t + 1] = base[t] + cand[t];
    const int64_t Lcand = base[parts];
    const size_t bytes = sizeof(uint32_t) * (size_t)Lcand * sb.n * 6;
    sb.counts = static_cast<uint32_t *>(alloc.alloc ? alloc.alloc(bytes ? bytes : 1) : std::malloc(bytes ? bytes : 1));
    if (alloc.alloc) sb.release = alloc.release ? alloc.release : [](void *) {};
    if (!sb.counts) throw std::runtime_error("out of memory for the allele counts of " + fname);
    sb.chrom_id.resize(Lcand);
    sb.pos.resize(Lcand);
    run_all([&](int t) {
        int64_t li = base[t];
        int last = -1;
        std::vector<std::string> &nm = names[t];
        for (const char *p = buf + cuts[t], *end = buf + cuts[t + 1]; p < end;) {
            const void *nl = std::memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? (const char *)nl : end;
            const char *q = le;
            if (q > p && q[-1] == '\r') --q;
            PileupConverter::Locus L;
            if (conv.decode(p, q, L)) {
                if (last < 0 || nm[last].size() != L.chrom_len || std::memcmp(nm[last].data(), L.chrom, L.chrom_len) != 0) {
                    last = -1;
                    for (size_t i = 0; i < nm.size(); ++i)
                        if (nm[i].size() == L.chrom_len && std::memcmp(nm[i].data(), L.chrom, L.chrom_len) == 0) { last = (int)i; break; }
                    if (last < 0) { nm.emplace_back(L.chrom, L.chrom_len); last = (int)nm.size() - 1; }
                }
                uint32_t *dst = sb.counts + (size_t)li * sb.n * 6;
                for (int i = 0; i < sb.n * 6; ++i) {
                    if (L.counts[i] > 0xFFFFFFFFull) throw std::runtime_error("pileup: a count exceeds 32 bits");
                    dst[i] = (uint32_t)L.counts[i];
                }
                sb.chrom_id[li] = last;
                sb.pos[li] = L.pos;
                ++li;
            }
            p = nl ? le + 1 : end;
        }
        kept[t] = li - base[t];
    });
    std::vector<std::vector<int32_t>> remap(parts);
    for (int t = 0; t < parts; ++t)
        for (const std::string &s2 : names[t]) {
            int g = -1;
            for (size_t i = 0; i < sb.chrom_names.size(); ++i) if (sb.chrom_names[i] == s2) { g = (int)i; break; }
            if (g < 0) { sb.chrom_names.push_back(s2); g = (int)sb.chrom_names.size() - 1; }
            remap[t].push_back(g);
        }
    int64_t w = 0;
    for (int t = 0; t < parts; ++t) { // close the holes the dropped loci left at the end of every slice
        const int64_t r0 = base[t], cnt = kept[t];
        for (int64_t i = 0; i < cnt; ++i) sb.chrom_id[r0 + i] = remap[t][sb.chrom_id[r0 + i]];
        if (w != r0 && cnt > 0) {
            std::memmove(sb.counts + (size_t)w * sb.n * 6, sb.counts + (size_t)r0 * sb.n * 6, sizeof(uint32_t) * (size_t)cnt * sb.n * 6);
            std::memmove(sb.chrom_id.data() + w, sb.chrom_id.data() + r0, sizeof(int32_t) * cnt);
            std::memmove(sb.pos.data() + w, sb.pos.data() + r0, sizeof(uint64_t) * cnt);
        }
        w += cnt;
    }
    sb.L = w;
    sb.chrom_id.resize(w);
    sb.pos.resize(w);
    return sb;
}

} // namespace pgh
