// hostcheck.cpp -- GPU-free driver over the CLI's host logic, used by the CPU test-suite:
//   hostcheck fmt                      : stdin "x n_digits" per line -> "display(x) roundup_own(x,n)"
//   hostcheck phen <file> <delim> <name_col> <size_col> <c1,c2,..>
//   hostcheck parse <sync> <threads>   : "L n" then one line per locus: chrom pos counts[n*6]
#include "host_util.h"
#include "pileup.h"
#include "rank_gate.h"
#include <atomic>
#include <thread>
#include <fstream>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
using namespace pgh;

static std::vector<std::string> splitc(const std::string &s) {
    std::vector<std::string> v; std::stringstream ss(s); std::string t;
    while (std::getline(ss, t, ',')) v.push_back(t);
    return v;
}

int main(int argc, char **argv) {
    try {
        if (argc < 2) return 2;
        const std::string mode = argv[1];
        if (mode == "gate") { // hostcheck gate <ranks> <failing rank or -1> <rounds>: the rank threads' go / abort agreement
            const int R = std::atoi(argv[2]), bad = std::atoi(argv[3]), rounds = std::atoi(argv[4]);
            RankGate gate(R);
            std::atomic<int> entered{0}, own{0}, aborted{0}, finished{0};
            std::vector<std::thread> th;
            for (int r = 0; r < R; ++r)
                th.emplace_back([&, r] {
                    try {
                        for (int i = 0; i < rounds; ++i) {
                            gate.pass([&] { if (r == bad && i == rounds - 1) throw std::runtime_error("rank failed"); });
                            if (i == rounds - 1) ++entered; // the "collective": every rank or none
                        }
                        ++finished;
                    } catch (const RankAborted &) { ++aborted; } catch (const std::exception &) { ++own; }
                });
            for (auto &t : th) t.join();
            std::cout << "entered " << entered << " own " << own << " aborted " << aborted << " finished " << finished << "\n";
        } else if (mode == "num") { // hostcheck num <f64|u64|i64> <text>: the strict parsers of the flag values and the phenotype file
            const std::string kind = argv[2], text = argc > 3 ? argv[3] : "";
            double d; uint64_t u; int64_t i;
            if (kind == "f64") { if (parse_f64_strict(text, d)) std::cout << "ok " << rust_display(d) << "\n"; else std::cout << "reject\n"; }
            else if (kind == "u64") { if (parse_u64_strict(text, u)) std::cout << "ok " << u << "\n"; else std::cout << "reject\n"; }
            else { if (parse_i64_strict(text, i)) std::cout << "ok " << i << "\n"; else std::cout << "reject\n"; }
        } else if (mode == "fmt") {
            std::string line;
            while (std::getline(std::cin, line)) {
                char *end; const double x = std::strtod(line.c_str(), &end); const int nd = std::atoi(end);
                std::cout << rust_display(x) << " " << roundup_own(x, nd) << "\n";
                // the in-place forms the CSV writers use must print the same characters
                std::string a1 = "x", a2 = "y";
                append_rust_display(a1, x); append_roundup_own(a2, x, nd);
                if (a1 != "x" + rust_display(x) || a2 != "y" + roundup_own(x, nd)) throw std::runtime_error("append_* differ from the string forms for " + line);
            }
        } else if (mode == "phen") {
            std::vector<int> cols; for (auto &t : splitc(argv[6])) cols.push_back(std::stoi(t));
            const Phen ph = parse_phen(argv[2], argv[3], std::atoi(argv[4]), std::atoi(argv[5]), cols);
            for (int i = 0; i < ph.n; ++i) {
                std::cout << ph.pool_names[i] << " " << rust_display(ph.pool_sizes[i]);
                for (int j = 0; j < ph.k; ++j) std::cout << " " << rust_display(ph.phen[(size_t)i * ph.k + j]);
                std::cout << "\n";
            }
        } else if (mode == "parse") {
            // hostcheck parse <sync> <threads> [16]: "16" asks for 16-bit counts; the first line then ends with the width in use
            const bool want16 = argc > 4 && std::string(argv[4]) == "16";
            const SyncBatch sb = parse_sync_file(argv[2], std::atoi(argv[3]), SyncAlloc(), want16);
            std::cout << sb.size() << " " << sb.n;
            if (want16) std::cout << " " << (sb.counts16 ? 16 : 32);
            std::cout << "\n";
            for (int64_t l = 0; l < sb.size(); ++l) {
                std::cout << sb.chrom(l) << " " << sb.pos[l];
                for (int i = 0; i < sb.n * 6; ++i)
                    std::cout << " " << (sb.counts16 ? (uint32_t)sb.counts16[(size_t)l * sb.n * 6 + i] : sb.counts[(size_t)l * sb.n * 6 + i]);
                std::cout << "\n";
            }
        } else if (mode == "pileuplines") {
            // hostcheck pileuplines <file> <remove_ns> <max_err> <min_depth> <breadth> <maf> <pool sizes,..>
            // one answer per input line: "K <sync line>" kept, "D" dropped (None), "E" the reference would panic
            PileupFilter f;
            f.remove_ns = (std::atoi(argv[3]) & 1) != 0;
            f.keep_lowercase_reference = (std::atoi(argv[3]) & 2) != 0; // bit 1 of the first filter argument
            f.max_base_error_rate = std::strtod(argv[4], nullptr);
            f.min_coverage_depth = std::strtoull(argv[5], nullptr, 10);
            f.min_coverage_breadth = std::strtod(argv[6], nullptr);
            f.min_allele_frequency = std::strtod(argv[7], nullptr);
            for (auto &t : splitc(argv[8])) f.pool_sizes.push_back(std::strtod(t.c_str(), nullptr));
            const PileupConverter conv(f);
            std::ifstream in(argv[2]);
            std::string line, out;
            while (std::getline(in, line)) {
                if (!line.empty() && line.back() == '\r') line.pop_back();
                out.clear();
                try {
                    if (conv.convert(line.data(), line.data() + line.size(), out)) std::cout << "K " << out;
                    else std::cout << "D\n";
                } catch (const std::exception &) { std::cout << "E\n"; }
            }
        } else if (mode == "pileup2sync") { // hostcheck pileup2sync <file> <out> <threads> <pool names,..> <filter as above>
            PileupFilter f;
            f.remove_ns = std::atoi(argv[6]) != 0;
            f.max_base_error_rate = std::strtod(argv[7], nullptr);
            f.min_coverage_depth = std::strtoull(argv[8], nullptr, 10);
            f.min_coverage_breadth = std::strtod(argv[9], nullptr);
            f.min_allele_frequency = std::strtod(argv[10], nullptr);
            for (auto &t : splitc(argv[11])) f.pool_sizes.push_back(std::strtod(t.c_str(), nullptr));
            const auto t0 = std::chrono::steady_clock::now();
            const int64_t kept = pileup_to_sync_file(argv[2], splitc(argv[5]), f, argv[3], std::atoi(argv[4]));
            std::cout << kept << " loci in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s\n";
        } else if (mode == "ksplit") { // hostcheck ksplit <n> <k> <seed>: the seeded outer folds of the CV harness
            const int64_t n = std::atoll(argv[2]);
            SplitMix64 rng(std::strtoull(argv[4], nullptr, 10));
            int k = 0;
            const std::vector<int32_t> g = k_split(n, std::atoi(argv[3]), rng.permutation(n), k);
            std::printf("%d", k);
            for (int32_t x : g) std::printf(" %d", x);
            std::printf("\n");
        } else if (mode == "parsetime") { // throughput of parse_sync_file: hostcheck parsetime <sync> <threads>
            // a caller-owned buffer handed out again on the second run: the second time has no first-touch page faults in it
            static void *keep = nullptr; static size_t keep_cap = 0;
            SyncAlloc al;
            al.alloc = [](size_t bytes) -> void * { if (bytes > keep_cap) { std::free(keep); keep = std::malloc(bytes); keep_cap = bytes; } return keep; };
            al.release = [](void *) {};
            const bool c16 = argc > 4 && std::string(argv[4]) == "16";
            for (int rep = 0; rep < 2; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                const SyncBatch sb = parse_sync_file(argv[2], std::atoi(argv[3]), al, c16);
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::cout << sb.size() << " loci x " << sb.n << " pools in " << dt << " s" << (rep ? " (buffer warm)" : "") << "\n";
            }
        } else return 2;
        return 0;
    } catch (const std::exception &e) { std::cerr << "hostcheck: " << e.what() << "\n"; return 1; }
}
