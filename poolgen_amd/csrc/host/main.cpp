// main.cpp -- `poolgen` command line for the hot subcommands, flag-compatible with the
// reference CLI (src/main.rs:26-143): pileup2sync, chisq_test, pearson_corr, ols_iter, ols_iter_with_kinship
// (and gp_ols as a plain coefficient dump).  Parsing/formatting/ordering follow the reference
// (base/sync.rs:606-970, :972-1180; gwas/ols.rs:255-275, :372-433); all arithmetic on the loci
// is done by libpoolgen_hip.so through its C ABI.  Anything else the reference CLI offers is out
// of scope and reported as such.
#include "host_util.h"
#include "pileup.h"
#include "../../../include/poolgen_hip.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <climits>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <exception>
#include "gp_cv.h"
#include "operators.h"
#include "rank_gate.h"
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <time.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

using namespace pgh;

struct Args {
    std::string analysis, fname, output, phen_fname, phen_delim = ",";
    double max_base_error_rate = 0.01, min_coverage_breadth = 1.0, min_allele_frequency = 0.001,
           max_missingness_rate = 0.0, xxt = 0.75;
    uint64_t min_coverage_depth = 1;
    bool keep_ns = false, keep_p_minus_1 = false, generate_plots = false, sig_only = false, keep_lowercase_reference = false;
    int phen_name_col = 0, phen_pool_size_col = 1, n_threads = 1;
    long stream_chunk_mb = -1; // ols_iter_with_kinship: pieces of this size are parsed / copied / loaded in turn (-1 = automatic)
    std::vector<int> phen_value_col{2};
    int k_folds = 10, n_reps = 3; // genomic_prediction_cross_validation (main.rs:104-109)
    uint64_t seed = 42;           // ... and the seed of its folds (an extension: the reference's folds are unrepeatable)
    uint64_t window_size_bp = 100, window_slide_size_bp = 50, min_loci_per_window = 10; // fst / heterozygosity (main.rs:110-118)
    // multi-GPU (an extension: the reference's parallel axis is --n-threads, one worker per file chunk, sync.rs:913-939):
    // the input is cut into one contiguous byte range per GPU, each with its own parser threads.  0 = flag absent.
    int n_gpus = 0;
    std::vector<int> gpu_ids; // device ordinals of the ranks (default 0, 1, .., n_gpus - 1)
};

static double parse_valid_freq(const std::string &v, const std::string &flag) { // helpers.rs:93-100
    double x = 0.0;
    if (!parse_f64_strict(v, x)) throw std::runtime_error("`" + v + "` isn't a valid number (" + flag + ")");
    if (x < 0.0 || x > 1.0) throw std::runtime_error("Value must be between 0.0 and 1.0, got `" + rust_display(x) + "` (" + flag + ")");
    return x;
}

// integer flag values as clap's `parse::<usize / u64>()` reads them: digits only, nothing after them, no sign for the unsigned
static uint64_t flag_u64(const std::string &v, const std::string &flag) {
    uint64_t x = 0;
    if (!parse_u64_strict(v, x)) throw std::runtime_error("invalid value `" + v + "` for " + flag + ": a non-negative integer is expected");
    return x;
}
static int flag_int(const std::string &v, const std::string &flag, int64_t lo = 0) {
    int64_t x = 0;
    if (!parse_i64_strict(v, x) || x < lo || x > INT32_MAX)
        throw std::runtime_error("invalid value `" + v + "` for " + flag + ": an integer >= " + std::to_string(lo) + " is expected");
    return (int)x;
}

static const char *USAGE =
    "poolgen <analysis> -f <input> -p <phenotypes.csv> [flags]      (MI355X build of the per-locus regression path)\n"
    "analyses: pileup2sync, chisq_test, pearson_corr, ols_iter, ols_iter_with_kinship, mle_iter_with_kinship,\n"
    "          genomic_prediction_cross_validation, fst, heterozygosity\n"
    "  -f, --fname <file>                 *.sync, or *.pileup / *.mpileup (converted in memory: exactly what pileup2sync followed by the\n"
    "                                     analysis on its sync file gives -- including the reference's column quirk: pileup2sync\n"
    "                                     writes A:T:C:G:DEL:N, the sync reader labels the columns A,T,C,G,N,DEL, so on pileup-derived\n"
    "                                     counts the default N removal drops the DELETION column and --keep-ns keeps it)\n"
    "  -p, --phen-fname <file>            delimited file: pool name, pool size, trait value(s)\n"
    "  -o, --output <file>                must not exist; default: derived from the input name and the time\n"
    "      --phen-delim <,>  --phen-name-col <0>  --phen-pool-size-col <1>  --phen-value-col <2[,3..]>\n"
    "      --max-base-error-rate <0.01>  --min-coverage-depth <1>  --min-coverage-breadth <1.0>\n"
    "      --min-allele-frequency <0.001>  --max-missingness-rate <0.0>  --keep-ns  --keep-lowercase-reference\n"
    "      --keep-p-minus-1                drop the major allele of every locus when loading the matrix\n"
    "  -x, --xxt-eigen-variance-explained <0.75>   ols_iter_with_kinship: the n_eigenvecs rule's threshold\n"
    "      --k-folds <10>  --n-reps <3>  --seed <42>     genomic_prediction_cross_validation\n"
    "      --window-size-bp <100>  --window-slide-size-bp <50>  --min-loci-per-window <10>   fst, heterozygosity\n"
    "      --n-threads <1>                 parser / writer threads\n"
    "      --stream-chunk-mb <N>           size of the pieces the input is taken in (0: whole file, kinship path only)\n"
    "      --n-gpus <N>  [--gpu-ids a,b,..]  chisq_test, pearson_corr, ols_iter, ols_iter_with_kinship: one contiguous part of\n"
    "                                      the input per GPU (own parser threads: --n-threads is the total); the kinship sums are\n"
    "                                      all-reduced over the GPUs with RCCL; the kinship path then needs an input sorted by\n"
    "                                      (chromosome, position)\n"
    "environment: PGH_TIMING=1 prints the phases' wall-clock on stderr\n";

static Args parse_args(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; ++i)
        if (std::string(argv[i]) == "-h" || std::string(argv[i]) == "--help") { std::cout << USAGE; std::exit(0); }
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i], v;
        auto eq = k.find('=');
        bool has_v = false;
        if (k.rfind("--", 0) == 0 && eq != std::string::npos) { v = k.substr(eq + 1); k = k.substr(0, eq); has_v = true; }
        auto val = [&]() -> std::string {
            if (has_v) return v;
            if (i + 1 >= argc) throw std::runtime_error("missing value for " + k);
            return argv[++i];
        };
        if (k == "-f" || k == "--fname") a.fname = val();
        else if (k == "-o" || k == "--output") a.output = val();
        else if (k == "-p" || k == "--phen-fname") a.phen_fname = val();
        else if (k == "--phen-delim") a.phen_delim = val();
        else if (k == "--phen-name-col") a.phen_name_col = flag_int(val(), k);
        else if (k == "--phen-pool-size-col") a.phen_pool_size_col = flag_int(val(), k);
        else if (k == "--phen-value-col") {
            a.phen_value_col.clear();
            std::stringstream ss(val());
            std::string t;
            while (std::getline(ss, t, ',')) a.phen_value_col.push_back(flag_int(t, k));
        } else if (k == "--n-threads") a.n_threads = flag_int(val(), k, 1);
        else if (k == "--max-base-error-rate") a.max_base_error_rate = parse_valid_freq(val(), k);
        else if (k == "--min-coverage-breadth") a.min_coverage_breadth = parse_valid_freq(val(), k);
        else if (k == "--min-coverage-depth") a.min_coverage_depth = flag_u64(val(), k);
        else if (k == "--min-allele-frequency") a.min_allele_frequency = parse_valid_freq(val(), k);
        else if (k == "--max-missingness-rate") a.max_missingness_rate = parse_valid_freq(val(), k);
        else if (k == "-x" || k == "--xxt-eigen-variance-explained") a.xxt = parse_valid_freq(val(), k);
        else if (k == "--keep-ns") a.keep_ns = true;
        else if (k == "--keep-p-minus-1") a.keep_p_minus_1 = true;
        else if (k == "--generate-plots") a.generate_plots = true;
        else if (k == "--output-sig-snps-only") a.sig_only = true;
        else if (k == "--keep-lowercase-reference") a.keep_lowercase_reference = true; // pileup inputs only (pileup.rs:280-299)
        else if (k == "--stream-chunk-mb") a.stream_chunk_mb = flag_int(val(), k);
        else if (k == "--n-gpus") a.n_gpus = flag_int(val(), k, 1);
        else if (k == "--gpu-ids") {
            std::stringstream ss(val());
            std::string t;
            while (std::getline(ss, t, ',')) a.gpu_ids.push_back(flag_int(t, k));
        }
        else if (k == "--k-folds") a.k_folds = flag_int(val(), k, 1);
        else if (k == "--n-reps") a.n_reps = flag_int(val(), k, 1);
        else if (k == "--seed") a.seed = flag_u64(val(), k);
        else if (k == "--window-size-bp") a.window_size_bp = flag_u64(val(), k);
        else if (k == "--window-slide-size-bp") a.window_slide_size_bp = flag_u64(val(), k);
        else if (k == "--min-loci-per-window") a.min_loci_per_window = flag_u64(val(), k);
        else if (k.rfind("-", 0) == 0) throw std::runtime_error("unknown flag " + k);
        else pos.push_back(k);
    }
    if (pos.size() != 1) throw std::runtime_error("usage: poolgen <analysis> -f <sync> -p <phen.csv> [flags]   (--help lists them)");
    a.analysis = pos[0];
    if (a.fname.empty() || a.phen_fname.empty()) throw std::runtime_error("-f/--fname and -p/--phen-fname are required");
    if (!a.gpu_ids.empty()) {
        if (a.n_gpus == 0) a.n_gpus = (int)a.gpu_ids.size();
        if ((int)a.gpu_ids.size() != a.n_gpus) throw std::runtime_error("--gpu-ids must list exactly --n-gpus devices");
    }
    return a;
}

static std::string basename_no_ext(const std::string &f) { // sync.rs:889-902
    const auto p = f.rfind('.');
    return p == std::string::npos ? std::string() : f.substr(0, p);
}

static std::string unix_time_string() {
    const double t = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    return rust_display(t);
}

// OpenOptions::create_new: refuse to overwrite (sync.rs:906, :942-947; ols.rs:285, :402-407)
static FILE *create_new(const std::string &path) {
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0644);
    if (fd < 0) throw std::runtime_error("Unable to create file: " + path + " (it must not exist)");
    return fdopen(fd, "w");
}

struct Ctx {
    pg_ctx *c = nullptr;
    int device = 0;
    explicit Ctx(int dev = 0) : device(dev) {
        if (pg_create(&c, dev, nullptr) != PG_OK)
            throw std::runtime_error("GPU " + std::to_string(dev) + ": " + pg_last_error(nullptr));
    }
    Ctx(const Ctx &) = delete;
    Ctx &operator=(const Ctx &) = delete;
    ~Ctx() { pg_destroy(c); }
    void ok(int rc, const char *what) {
        if (rc != PG_OK) throw std::runtime_error(std::string(what) + ": " + pg_last_error(c));
    }
};

// pools with a missing phenotype are removed before the locus operators (remove_missing,
// sync.rs:508-549).  The reference forgets to shrink FilterStats.pool_sizes and panics in that
// case; here the pool sizes are subset consistently.
static std::vector<int> complete_pools(const Phen &ph) {
    std::vector<int> idx;
    for (int i = 0; i < ph.n; ++i) {
        double s = 0.0;
        for (int j = 0; j < ph.k; ++j) s += ph.phen[(size_t)i * ph.k + j];
        if (!std::isnan(s)) idx.push_back(i);
    }
    return idx;
}

// Formats items [0, count) with `fn(i, text)` on `n_threads` workers and writes them in item order: the file is byte-identical to
// a sequential loop.  Every worker owns ONE contiguous range per round (a round = what fits PGH_WRITE_ROUND_MB of text, default
// 1024: the 20 M rows of a 10 M-site kinship run are one round), formats it into its own buffer and -- once the sizes of the
// buffers in front of it are known -- copies it into the file itself with pwrite at its own offset; the file is extended once per
// round (ftruncate) so that the workers' writes never fight over its length.  (Round 3 re-spawned the workers twice per 64 k rows
// and took 0.6 - 1.2 s for those 20 M rows.)
template <typename F>
static void write_rows_parallel(FILE *fo, int64_t count, int n_threads, F fn) {
    if (n_threads < 1) n_threads = 1;
    if (count <= 0) return;
    size_t round_mb = 1024;
    if (const char *e = std::getenv("PGH_WRITE_ROUND_MB")) round_mb = (size_t)std::max(1L, std::atol(e));
    const int64_t per_round = std::max<int64_t>(n_threads, (int64_t)((round_mb << 20) / 64)); // ~64 bytes of text per item
    std::fflush(fo);
    const int fd = fileno(fo);
    off_t off = ftello(fo);
    for (int64_t base = 0; base < count; base += per_round) {
        const int64_t end = std::min(count, base + per_round);
        const int parts = (int)std::min<int64_t>(n_threads, end - base);
        const int64_t each = (end - base + parts - 1) / parts;
        std::vector<std::string> text(parts);
        std::vector<size_t> size(parts, 0);
        std::vector<int> failed(parts, 0);
        std::mutex mu;
        std::condition_variable cv;
        int formatted = 0;
        off_t round_off = off, round_end = off;
        bool sized = false;
        std::vector<std::thread> th;
        for (int t = 0; t < parts; ++t)
            th.emplace_back([&, t] {
                const int64_t lo = base + t * each, hi = std::min(end, lo + each);
                std::string &out = text[t];
                out.reserve((size_t)std::max<int64_t>(0, hi - lo) * 64);
                for (int64_t i = lo; i < hi; ++i) fn(i, out);
                off_t at;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    size[t] = out.size();
                    if (++formatted == parts) { // the last one to finish knows every size: extend the file once, release everybody
                        for (int q = 0; q < parts; ++q) round_end += (off_t)size[q];
                        if (::ftruncate(fd, round_end) != 0) failed[t] = 1;
                        sized = true;
                        cv.notify_all();
                    } else cv.wait(lk, [&] { return sized; });
                    at = round_off;
                    for (int q = 0; q < t; ++q) at += (off_t)size[q];
                }
                const char *q = out.data();
                size_t left = out.size();
                while (left > 0) {
                    const ssize_t w = ::pwrite(fd, q, left, at);
                    if (w <= 0) { failed[t] = 1; return; }
                    q += w; at += w; left -= (size_t)w;
                }
                std::string().swap(out);
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < parts; ++t)
            if (failed[t]) throw std::runtime_error("write failed (disk full?)");
        off = round_end;
    }
    fseeko(fo, off, SEEK_SET);
}

// PGH_TIMING=1: wall-clock of the CLI's phases on stderr
struct Lap {
    bool on = std::getenv("PGH_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "poolgen: %-24s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};

// The results are on disk and the file name is printed: end the process here.  Releasing tens of GB of device and pinned
// memory buffer by buffer (destructors, hipFree, the HIP runtime's shutdown) took 0.3 s of a 1.4 s run; the driver
// reclaims everything when the process ends.  PGH_CLEAN_EXIT=1 keeps the orderly teardown (leak checkers).
static void stamp(const char *what) { // PGH_TIMING=1: the wall clock itself, so that a wrapper can see what lies outside the laps
    if (!std::getenv("PGH_TIMING")) return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::fprintf(stderr, "poolgen: clock at %-15s %lld.%09ld\n", what, (long long)ts.tv_sec, ts.tv_nsec);
}
static int done_ok() {
    stamp("exit");
    std::cout.flush();
    std::cerr.flush();
    std::fflush(nullptr);
    if (!std::getenv("PGH_CLEAN_EXIT")) ::_exit(0);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// ols_iter_with_kinship on an input that is read in PIECES (config 5 of BASELINE.json: a file far larger than
// host memory): while the GPU takes piece c (H2D from pinned memory, loader, partial kinship), the worker
// threads already parse piece c + 1 into the other pinned buffer.  The frequency matrix stays resident in HBM
// piece by piece (288 GB hold 180 M columns of 200 pools); the host only keeps labels and results.
// The input must be sorted by (chromosome, position) -- what the whole-file path obtains by sorting
// (sync.rs:1092-1101) cannot be had across pieces -- and that is checked.
// ---------------------------------------------------------------------------------------------------------
struct UnsortedInput : std::runtime_error { using std::runtime_error::runtime_error; };

// The text of a piece is not looked at again once it is parsed: its pages leave the mapping here, on a thread of their own,
// instead of all at once when the process ends -- unmapping 27 GB of touched file pages (6.6 M page-table entries, one thread,
// inside exit) was 0.25 s per 5 GB of input AFTER the program's last line (profiles/r04_stream_*.log).  MADV_DONTNEED on a
// read-only private file mapping only drops the entries; the page cache keeps the file.  PGH_KEEP_MAPPED=1 leaves them.
// The helper threads belong to an object that lives SHORTER than the mapping (declare it after the MappedFile): its destructor
// joins them, so that no madvise is still on its way when the mapping goes (an exception, the fall-back to the whole-file path)
// and the address range may already belong to something else -- where MADV_DONTNEED would discard live data.
class TextDropper {
    std::mutex m_;
    std::vector<std::thread> th_;
    const bool keep_ = std::getenv("PGH_KEEP_MAPPED") != nullptr;
public:
    TextDropper() = default;
    TextDropper(const TextDropper &) = delete;
    TextDropper &operator=(const TextDropper &) = delete;
    void operator()(const char *b, const char *e) {
        if (keep_) return;
        const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
        const uintptr_t lo = ((uintptr_t)b + pg - 1) / pg * pg, hi = (uintptr_t)e / pg * pg;
        if (hi <= lo || hi - lo < ((uintptr_t)1 << 20)) return;
        std::lock_guard<std::mutex> g(m_);
        th_.emplace_back([lo, hi] { (void)::madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_DONTNEED); });
    }
    ~TextDropper() {
        for (auto &t : th_) if (t.joinable()) t.join();
    }
};

// The counts of a parsed batch -> the 32-bit device buffer the operators read.  A 16-bit batch (every count fits: the
// usual case) crosses the bus at half the size and is widened on the device; `stage16` is a reusable device scratch.
struct CountsUpload {
    uint16_t *stage16 = nullptr;
    size_t cap16 = 0;
    ~CountsUpload() { if (stage16) (void)hipFree(stage16); }
    void operator()(Ctx &gpu, const SyncBatch &sb, uint32_t *counts_dev) {
        auto hip_ok = [](hipError_t e, const char *what) {
            if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
        };
        if (!sb.counts16) {
            hip_ok(hipMemcpyAsync(counts_dev, sb.counts, sb.counts_bytes(), hipMemcpyHostToDevice, nullptr), "H2D counts");
            return;
        }
        if (sb.counts_bytes() > cap16) {
            if (stage16) hip_ok(hipFree(stage16), "free");
            stage16 = nullptr;
            cap16 = sb.counts_bytes() + sb.counts_bytes() / 8;
            hip_ok(hipMalloc((void **)&stage16, cap16), "device memory for the compact counts");
        }
        hip_ok(hipMemcpyAsync(stage16, sb.counts16, sb.counts_bytes(), hipMemcpyHostToDevice, nullptr), "H2D counts");
        gpu.ok(pg_expand_counts_u16_dev(gpu.c, stage16, (int64_t)sb.L * sb.n * 6, counts_dev), "expand counts");
    }
};

// ---- ranks: one GPU, one pg_ctx, one host thread (plus its share of the parser threads) each -------------------------
// The reference's parallel axis is a worker per contiguous byte range of the input (base/sync.rs:913-939); here a rank
// is such a worker with a GPU behind it.  Ranks are threads of this process; the kinship sums are all-reduced with RCCL
// inside libpoolgen_hip (pg_comm_init_rank / pg_allreduce_sum_dev).  PGH_COMM=host is a REHEARSAL mode for boxes with
// fewer GPUs than ranks (--gpu-ids 0,0: RCCL refuses two ranks on one device): the partial sums are then added on the
// host in rank order -- same control flow, same byte ranges, same files.
struct RankSetup {
    int n_ranks = 1;
    std::vector<int> devices;   // per rank
    std::vector<int> threads;   // parser / writer threads per rank
    bool rccl = false;          // --n-gpus given: a communicator is set up even for one rank
    bool host_comm = false;     // PGH_COMM=host
    unsigned char id[PG_COMM_ID_BYTES] = {0};
};

static RankSetup rank_setup(const Args &a, bool need_comm) {
    RankSetup r;
    r.n_ranks = a.n_gpus > 0 ? a.n_gpus : 1;
    for (int i = 0; i < r.n_ranks; ++i) r.devices.push_back(a.gpu_ids.empty() ? i : a.gpu_ids[i]);
    // every ordinal is checked BEFORE any rank thread exists: a rank that cannot open its device would otherwise leave the
    // others waiting for it inside ncclCommInitRank
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    for (int i = 0; i < r.n_ranks; ++i)
        if (r.devices[i] < 0 || r.devices[i] >= ndev)
            throw std::runtime_error("rank " + std::to_string(i) + " asks for GPU " + std::to_string(r.devices[i]) + " but " +
                                     std::to_string(ndev) + " GPU(s) are visible (--n-gpus / --gpu-ids)");
    const int total = std::max(a.n_threads, 1);
    for (int i = 0; i < r.n_ranks; ++i) r.threads.push_back(std::max(1, total / r.n_ranks + (i < total % r.n_ranks ? 1 : 0)));
    const char *e = std::getenv("PGH_COMM");
    r.host_comm = e && std::string(e) == "host";
    r.rccl = need_comm && a.n_gpus > 0 && !r.host_comm;
    if (r.rccl) {
        std::vector<int> d = r.devices;
        std::sort(d.begin(), d.end());
        if (std::adjacent_find(d.begin(), d.end()) != d.end())
            throw std::runtime_error("--gpu-ids lists a device twice: RCCL needs one GPU per rank (PGH_COMM=host rehearses the rank logic on fewer GPUs)");
        if (pg_comm_unique_id(r.id) != PG_OK) throw std::runtime_error(std::string("RCCL: ") + pg_last_error(nullptr));
    }
    return r;
}

// contiguous ranges of pieces per rank: rank r takes [first[r], first[r + 1])
static std::vector<int> deal_pieces(int npieces, int n_ranks) {
    std::vector<int> first(n_ranks + 1);
    for (int r = 0; r <= n_ranks; ++r) first[r] = (int)((int64_t)npieces * r / n_ranks);
    return first;
}

template <typename F>
static void run_ranks(int n_ranks, F fn) { // fn(rank) on one thread per rank; the first exception is rethrown
    std::vector<std::exception_ptr> err(n_ranks);
    std::vector<std::thread> th;
    for (int r = 1; r < n_ranks; ++r)
        th.emplace_back([&, r] { try { fn(r); } catch (...) { err[r] = std::current_exception(); } });
    try { fn(0); } catch (...) { err[0] = std::current_exception(); }
    for (auto &t : th) t.join();
    for (auto &e : err) // the rank that failed first-hand, not the ones that stopped because of it
        if (e) { try { std::rethrow_exception(e); } catch (const RankAborted &) {} catch (...) { throw; } }
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

struct KinRank {
    int rank = 0, device = 0, threads = 1, c0 = 0, c1 = 0;
    std::unique_ptr<Ctx> own;
    Ctx *gpu = nullptr;
    std::vector<double *> Gs;                         // the rank's pieces of the frequency matrix, resident in HBM
    std::vector<int64_t> ps;
    std::vector<std::string> chrom_names;             // the rank's dictionary
    std::vector<int32_t> lab_chr;                     // per column
    std::vector<uint64_t> lab_pos;
    std::vector<char> lab_al;
    std::vector<double> S_total;
    double *S_dev = nullptr;
    uint32_t *counts_dev = nullptr;
    struct Slot { void *p = nullptr; size_t cap = 0; } slot[2];
    std::string first_chrom, last_chrom;
    uint64_t first_pos = 0, last_pos = 0;
    bool have_last = false;
    int64_t p = 0, col0 = 0;
    int m = 0;
    double t_wait = 0, t_host = 0, t_gpu = 0;
    KinRank() = default;
    KinRank(const KinRank &) = delete;
    KinRank &operator=(const KinRank &) = delete;
    ~KinRank() { // an exception on the way (unsorted input, out of memory) must not leave the pieces behind: the caller may fall back to the whole-file path
        for (double *g : Gs) (void)hipFree(g);
        if (S_dev) (void)hipFree(S_dev);
        if (counts_dev) (void)hipFree(counts_dev);
        for (auto &sl : slot) if (sl.p) (void)hipHostFree(sl.p);
    }
};

static int run_kinship_streamed(const Args &a, const Phen &ph, Ctx &gpu0, Lap &lap, size_t chunk_bytes, bool is_pileup,
                                const PileupFilter &pf, const pg_filter &flt, const RankSetup &rs) {
    auto hip_ok = [](hipError_t e, const char *what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    };
    const MappedFile mf(a.fname);
    TextDropper drop_parsed_text; // (after mf: joined before the mapping goes)
    const int R = rs.n_ranks;
    const size_t want_pieces = std::max<size_t>((size_t)R, (mf.size() + chunk_bytes - 1) / chunk_bytes);
    const std::vector<size_t> cuts = mf.cuts(want_pieces);
    const int nchunks = (int)cuts.size() - 1;
    const int n = ph.n, k = ph.k;
    const std::vector<int> keep = complete_pools(ph); // remove_missing (ols.rs:287)
    if (keep.empty()) throw std::runtime_error("All pools have missing data. Please check the phenotype file.");
    const int n2 = (int)keep.size();
    const int64_t ld = n2 + (n2 & 1);
    std::vector<int32_t> pool_map(n, -1);
    for (int i = 0; i < n2; ++i) pool_map[keep[i]] = i;
    const std::vector<int> first = deal_pieces(nchunks, R);
    std::vector<KinRank> ranks(R);
    for (int r = 0; r < R; ++r) {
        ranks[r].rank = r; ranks[r].device = rs.devices[r]; ranks[r].threads = rs.threads[r];
        ranks[r].c0 = first[r]; ranks[r].c1 = first[r + 1];
    }
    auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    // every rank's context is created here, on the main thread: a GPU that cannot be opened ends the run before any rank
    // thread exists
    for (int r = 0; r < R; ++r) {
        KinRank &K = ranks[r];
        hip_ok(hipSetDevice(K.device), "hipSetDevice");
        if (r == 0 && gpu0.device == K.device) K.gpu = &gpu0;
        else { K.own.reset(new Ctx(K.device)); K.gpu = K.own.get(); }
    }
    RankGate gate(R);
    // ---- phase A, per rank: parse piece c + 1 || H2D + loader + partial kinship of piece c -----------------------
    run_ranks(R, [&](int r) {
        KinRank &K = ranks[r];
        Ctx &gpu = *K.gpu;
        gate.pass([&] {
            hip_ok(hipSetDevice(K.device), "hipSetDevice");
            K.S_total.assign((size_t)n2 * n2, 0.0);
            hip_ok(hipMalloc((void **)&K.S_dev, sizeof(double) * n2 * n2), "device memory");
        });
        if (rs.rccl) // collective over the rank threads: entered by all of them or by none, and its outcome agreed on
            gate.pass([&] { gpu.ok(pg_comm_init_rank(gpu.c, rs.id, R, r), "RCCL communicator"); });
        if (K.c0 >= K.c1) return;
        auto alloc_for = [&K](int i) { // two pinned buffers that grow on demand and are handed out in turn
            SyncAlloc al;
            al.alloc = [&K, i](size_t bytes) -> void * {
                if (bytes > K.slot[i].cap) {
                    if (K.slot[i].p) (void)hipHostFree(K.slot[i].p);
                    K.slot[i].p = nullptr; K.slot[i].cap = 0;
                    const size_t want = bytes + bytes / 8;
                    if (hipHostMalloc(&K.slot[i].p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
                    K.slot[i].cap = want;
                }
                return K.slot[i].p;
            };
            al.release = [](void *) {};
            return al;
        };
        auto parse_piece = [&](int c) {
            (void)hipSetDevice(K.device); // the pinned allocator runs on the parser's thread
            const char *b = mf.data() + cuts[c], *e = mf.data() + cuts[c + 1];
            SyncBatch parsed = is_pileup ? parse_pileup_buffer(b, e, K.threads, pf, alloc_for(c & 1))
                                         : parse_sync_buffer(b, e, K.threads, n, alloc_for(c & 1), true);
            drop_parsed_text(b, e);
            return parsed;
        };
        std::future<SyncBatch> next = std::async(std::launch::async, parse_piece, K.c0);
        std::vector<double> S_piece((size_t)n2 * n2);
        size_t counts_cap = 0;
        CountsUpload upload;
        for (int c = K.c0; c < K.c1; ++c) {
            double t0 = clk();
            SyncBatch sb = next.get();
            if (c + 1 < K.c1) next = std::async(std::launch::async, parse_piece, c + 1);
            K.t_wait += clk() - t0; t0 = clk();
            if (sb.L == 0) continue;
            if (sb.n != n) throw std::runtime_error("the number of pools in the input and in the phenotype file differ");
            for (int64_t l = 0; l < sb.L; ++l) { // sortedness, within and across the rank's pieces
                const std::string &ch = sb.chrom(l);
                if (K.have_last) {
                    const int cmp = K.last_chrom.compare(ch);
                    if (cmp > 0 || (cmp == 0 && K.last_pos > sb.pos[l]))
                        throw UnsortedInput("streamed ols_iter_with_kinship needs the input sorted by chromosome and position (line of " +
                                            ch + ":" + std::to_string(sb.pos[l]) + "); use --stream-chunk-mb 0 to load the whole file");
                    if (cmp != 0) K.last_chrom = ch;
                } else { K.last_chrom = ch; K.have_last = true; K.first_chrom = ch; K.first_pos = sb.pos[l]; }
                K.last_pos = sb.pos[l];
            }
            K.t_host += clk() - t0; t0 = clk();
            const size_t bytes32 = sizeof(uint32_t) * (size_t)sb.L * n * 6;
            if (bytes32 > counts_cap) {
                if (K.counts_dev) hip_ok(hipFree(K.counts_dev), "free");
                K.counts_dev = nullptr;
                counts_cap = bytes32 + bytes32 / 8;
                hip_ok(hipMalloc((void **)&K.counts_dev, counts_cap), "device memory for the counts");
            }
            upload(gpu, sb, K.counts_dev);
            int64_t pc = 0;
            gpu.ok(pg_load_plan_dev(gpu.c, K.counts_dev, sb.L, n, ph.pool_sizes.data(), &flt, a.keep_p_minus_1 ? 1 : 0, nullptr, &pc), "load");
            if (pc == 0) continue;
            double *G = nullptr;
            int64_t *col_locus_dev = nullptr;
            int32_t *col_allele_dev = nullptr;
            hip_ok(hipMalloc((void **)&G, sizeof(double) * (size_t)pc * ld), "device memory for the genotype matrix");
            K.Gs.push_back(G);
            K.ps.push_back(pc);
            hip_ok(hipMalloc((void **)&col_locus_dev, sizeof(int64_t) * pc), "device memory");
            hip_ok(hipMalloc((void **)&col_allele_dev, sizeof(int32_t) * pc), "device memory");
            gpu.ok(pg_load_emit_dev(gpu.c, pool_map.data(), n2, G, ld, col_locus_dev, col_allele_dev), "load");
            std::vector<int64_t> col_locus(pc);
            std::vector<int32_t> col_allele(pc);
            hip_ok(hipMemcpy(col_locus.data(), col_locus_dev, sizeof(int64_t) * pc, hipMemcpyDeviceToHost), "D2H labels");
            hip_ok(hipMemcpy(col_allele.data(), col_allele_dev, sizeof(int32_t) * pc, hipMemcpyDeviceToHost), "D2H labels");
            (void)hipFree(col_locus_dev); (void)hipFree(col_allele_dev);
            K.t_gpu += clk() - t0; t0 = clk();
            std::vector<int32_t> remap(sb.chrom_names.size());
            for (size_t i = 0; i < sb.chrom_names.size(); ++i) {
                int g = -1;
                for (size_t j = 0; j < K.chrom_names.size(); ++j) if (K.chrom_names[j] == sb.chrom_names[i]) { g = (int)j; break; }
                if (g < 0) { K.chrom_names.push_back(sb.chrom_names[i]); g = (int)K.chrom_names.size() - 1; }
                remap[i] = g;
            }
            for (int64_t q = 0; q < pc; ++q) {
                K.lab_chr.push_back(remap[sb.chrom_id[col_locus[q]]]);
                K.lab_pos.push_back(sb.pos[col_locus[q]]);
                K.lab_al.push_back(ALLELES[col_allele[q]]);
            }
            K.t_host += clk() - t0; t0 = clk();
            gpu.ok(pg_kinship_partial_dev(gpu.c, G, pc, n2, ld, K.S_dev), "kinship");
            hip_ok(hipMemcpy(S_piece.data(), K.S_dev, sizeof(double) * n2 * n2, hipMemcpyDeviceToHost), "D2H kinship");
            for (size_t i = 0; i < K.S_total.size(); ++i) K.S_total[i] += S_piece[i];
            K.p += pc;
            K.t_gpu += clk() - t0;
        }
        if (K.counts_dev) { (void)hipFree(K.counts_dev); K.counts_dev = nullptr; }
        for (auto &sl : K.slot) if (sl.p) { (void)hipHostFree(sl.p); sl.p = nullptr; sl.cap = 0; }
    });
    if (std::getenv("PGH_TIMING"))
        for (const KinRank &K : ranks)
            std::fprintf(stderr, "poolgen: rank %d (GPU %d, pieces %d..%d, %d parser threads): waited for the parser %.3f s, host bookkeeping %.3f s, copies + device %.3f s\n",
                         K.rank, K.device, K.c0, K.c1, K.threads, K.t_wait, K.t_host, K.t_gpu);
    lap("pieces: parse | H2D + loader + partial kinship");
    // ---- between the phases: the order across ranks, the global column offsets ------------------------------------
    int64_t p = 0;
    const KinRank *prev = nullptr;
    for (KinRank &K : ranks) {
        K.col0 = p;
        p += K.p;
        if (!K.have_last) continue;
        if (prev) {
            const int cmp = prev->last_chrom.compare(K.first_chrom);
            if (cmp > 0 || (cmp == 0 && prev->last_pos > K.first_pos))
                throw UnsortedInput("streamed ols_iter_with_kinship needs the input sorted by chromosome and position (line of " +
                                    K.first_chrom + ":" + std::to_string(K.first_pos) + "); use --stream-chunk-mb 0 to load the whole file");
        }
        prev = &K;
    }
    if (p <= 0) throw std::runtime_error("no loci passed the filters");
    std::vector<double> Y;
    for (int i : keep) for (int j = 0; j < k; ++j) Y.push_back(ph.phen[(size_t)i * k + j]);
    if (!a.output.empty()) { FILE *t = create_new(a.output); fclose(t); ::unlink(a.output.c_str()); } // ols.rs:285
    std::vector<double> S_host; // PGH_COMM=host, or no communicator at all: the ranks' sums added in rank order
    if (!rs.rccl) {
        S_host.assign((size_t)n2 * n2, 0.0);
        for (const KinRank &K : ranks)
            for (size_t i = 0; i < S_host.size(); ++i) S_host[i] += K.S_total[i];
    }
    // ---- phase B, per rank: all-reduce of the kinship sums, the n x n step (replicated), the rank's sweeps ---------
    std::vector<double> beta((size_t)p * k), pval((size_t)p * k);
    run_ranks(R, [&](int r) {
        KinRank &K = ranks[r];
        Ctx &gpu = *K.gpu;
        gate.pass([&] {
            hip_ok(hipSetDevice(K.device), "hipSetDevice");
            hip_ok(hipMemcpy(K.S_dev, rs.rccl ? K.S_total.data() : S_host.data(), sizeof(double) * n2 * n2, hipMemcpyHostToDevice), "H2D kinship");
        });
        if (rs.rccl) gpu.ok(pg_allreduce_sum_dev(gpu.c, K.S_dev, (int64_t)n2 * n2), "RCCL all-reduce of the kinship sums");
        gpu.ok(pg_kinship_set(gpu.c, K.S_dev, p, n2, Y.data(), k, a.xxt, -1, &K.m, nullptr, nullptr), "ols_iter_with_kinship");
        int64_t off = K.col0;
        for (size_t c = 0; c < K.Gs.size(); ++c) {
            double *out_dev = nullptr;
            const size_t cnt = (size_t)K.ps[c] * k;
            hip_ok(hipMalloc((void **)&out_dev, sizeof(double) * 3 * cnt), "device memory for the results");
            gpu.ok(pg_ols_sweep_dev(gpu.c, K.Gs[c], K.ps[c], n2, ld, out_dev, out_dev + cnt, out_dev + 2 * cnt), "ols_iter_with_kinship");
            gpu.ok(pg_synchronize(gpu.c), "ols_iter_with_kinship");
            hip_ok(hipMemcpy(beta.data() + (size_t)off * k, out_dev, sizeof(double) * cnt, hipMemcpyDeviceToHost), "D2H results");
            hip_ok(hipMemcpy(pval.data() + (size_t)off * k, out_dev + 2 * cnt, sizeof(double) * cnt, hipMemcpyDeviceToHost), "D2H results");
            (void)hipFree(out_dev);
            (void)hipFree(K.Gs[c]);
            K.Gs[c] = nullptr;
            off += K.ps[c];
        }
        K.Gs.clear();
    });
    const int m = ranks[0].m;
    for (const KinRank &K : ranks)
        if (K.m != m) throw std::runtime_error("internal error: the ranks disagree on n_eigenvecs");
    lap("all-reduce + eigen rule + fits + D2H");
    std::string out = a.output;
    if (out.empty()) // ols.rs:393-398
        out = basename_no_ext(a.fname) + "-ols_iterative_xxt_" + std::to_string(m + 1) + "_eigens-" + unix_time_string() + ".csv";
    // label of global column q: the rank whose range holds it
    std::vector<int64_t> starts;
    for (const KinRank &K : ranks) starts.push_back(K.col0);
    auto label = [&](int64_t q, std::string &text) {
        const int r = (int)(std::upper_bound(starts.begin(), starts.end(), q) - starts.begin()) - 1;
        const KinRank &K = ranks[r];
        const int64_t i = q - K.col0;
        text += K.chrom_names[K.lab_chr[i]]; text += ","; text += std::to_string(K.lab_pos[i]); text += ","; text.push_back(K.lab_al[i]);
    };
    FILE *fo = create_new(out);
    fputs("#chr,pos,alleles,phenotype,statistic,pvalue\n", fo); // ols.rs:409
    write_rows_parallel(fo, (int64_t)k * p, a.n_threads, [&](int64_t r, std::string &text) {
        const int64_t j = r / p, i = r - j * p; // rows are trait-major (ols.rs:411-433)
        // coefficient i carries label i of the (1+p)-long vectors whose entry 0 is "intercept" (ols.rs:421-425)
        if (i == 0) text += "intercept,0,intercept";
        else label(i - 1, text);
        text += ",Pheno_"; text += std::to_string(j); text.push_back(',');
        append_rust_display(text, beta[(size_t)i * k + j]); text.push_back(',');
        append_rust_display(text, pval[(size_t)i * k + j]); text.push_back('\n');
    });
    fclose(fo);
    lap("format + write CSV");
    std::cout << out << "\n";
    return done_ok();
}

// chisq_test / pearson_corr / ols_iter (main.rs:245-271): the per-locus operators know nothing beyond their own line, so
// the file is taken in pieces whatever its size -- the worker threads parse piece c + 1 into one of two pinned buffers
// (16-bit counts when they fit) while the GPU takes piece c and its rows are formatted and appended, in file order
// (sync.rs:927-946).  Nothing of the size of the input is ever allocated, pinned or copied in one go.
static int run_batch_streamed(const Args &a, const Phen &ph, Ctx &gpu0, Lap &lap, int mode, size_t chunk_bytes, bool is_pileup,
                              const PileupFilter &pf, const pg_filter &flt, const RankSetup &rs) {
    auto hip_ok = [](hipError_t e, const char *what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    };
    const MappedFile mf(a.fname);
    TextDropper drop_parsed_text; // (after mf: joined before the mapping goes)
    const int R = rs.n_ranks;
    const std::vector<size_t> cuts = mf.cuts(std::max<size_t>((size_t)R, (mf.size() + chunk_bytes - 1) / chunk_bytes));
    const int nchunks = (int)cuts.size() - 1;
    const std::vector<int> first = deal_pieces(nchunks, R);
    const int n = ph.n, k = ph.k;
    std::string out = a.output;
    if (out.empty()) out = basename_no_ext(a.fname) + "-" + unix_time_string() + "-" + a.analysis + ".csv"; // sync.rs:903
    { FILE *t = create_new(out); fclose(t); ::unlink(out.c_str()); } // probe, as the reference does before any work (sync.rs:906)
    // ols_iter drops the pools without phenotype first (ols.rs:206); pearson_corr keeps every pool (pairwise-complete inside)
    std::vector<int> keep(n);
    std::iota(keep.begin(), keep.end(), 0);
    std::vector<double> Y = ph.phen, ps = ph.pool_sizes;
    if (mode == 2) {
        keep = complete_pools(ph);
        if (keep.empty()) throw std::runtime_error("All pools have missing data. Please check the phenotype file.");
        if ((int)keep.size() != n) {
            std::cerr << "warning: " << n - keep.size() << " pools without phenotype removed (pool sizes subset accordingly)\n";
            Y.clear(); ps.clear();
            for (int i : keep) { ps.push_back(ph.pool_sizes[i]); for (int j = 0; j < k; ++j) Y.push_back(ph.phen[(size_t)i * k + j]); }
        }
    }
    const int n2 = (int)keep.size();
    const bool subset = n2 != n;
    const char *header = mode == 0 ? "#chr,pos,alleles,statistic,pvalue\n"                 // sync.rs:766
                                   : "#chr,pos,alleles,freq,phenotype,statistic,pvalue\n"; // sync.rs:950
    // One rank = one contiguous range of pieces, one GPU, its own pinned buffers and parser threads, and -- when there are
    // several -- its own part file, like the reference's one `.tmp` file per worker thread (sync.rs:794-870), concatenated in
    // rank order afterwards (sync.rs:951-968).  No exchange between the ranks: a locus needs nothing beyond its own line.
    std::vector<int64_t> totals(R, 0);
    std::vector<std::string> part(R);
    for (int r = 0; r < R; ++r) part[r] = R == 1 ? out : out + ".rank" + std::to_string(r) + ".tmp";
    auto cleanup_parts = [&] { if (R > 1) for (auto &f : part) ::unlink(f.c_str()); };
    try {
    run_ranks(R, [&](int r) {
        const int device = rs.devices[r], threads = rs.threads[r];
        hip_ok(hipSetDevice(device), "hipSetDevice");
        std::unique_ptr<Ctx> own;
        Ctx *gp = &gpu0;
        if (!(r == 0 && gpu0.device == device)) { own.reset(new Ctx(device)); gp = own.get(); }
        Ctx &gpu = *gp;
        struct Slot { void *p = nullptr; size_t cap = 0; } slot[2];
        auto alloc_for = [&](int i) {
            SyncAlloc al;
            al.alloc = [&slot, i](size_t bytes) -> void * {
                if (bytes > slot[i].cap) {
                    if (slot[i].p) (void)hipHostFree(slot[i].p);
                    slot[i].p = nullptr; slot[i].cap = 0;
                    const size_t want = bytes + bytes / 8;
                    if (hipHostMalloc(&slot[i].p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
                    slot[i].cap = want;
                }
                return slot[i].p;
            };
            al.release = [](void *) {};
            return al;
        };
        auto parse_piece = [&](int c) {
            (void)hipSetDevice(device);
            const char *b = mf.data() + cuts[c], *e = mf.data() + cuts[c + 1];
            SyncBatch parsed = is_pileup ? parse_pileup_buffer(b, e, threads, pf, alloc_for(c & 1))
                                         : parse_sync_buffer(b, e, threads, 0, alloc_for(c & 1), !subset);
            drop_parsed_text(b, e);
            return parsed;
        };
        const int c0 = first[r], c1 = first[r + 1];
        if (c0 >= c1) return;
        std::future<SyncBatch> next = std::async(std::launch::async, parse_piece, c0);
        FILE *fo = nullptr;
        uint32_t *counts_dev = nullptr;
        int32_t *n_out_dev = nullptr, *ids_dev = nullptr;
        double *mf_dev = nullptr, *stat_dev = nullptr, *pv_dev = nullptr;
        int64_t cap_loci = 0;
        CountsUpload upload;
        std::vector<uint32_t> counts2;
        std::vector<int32_t> n_out, ids;
        std::vector<double> mfq, stat, pv;
        const size_t per_stat = mode == 0 ? 1 : (size_t)PG_MAX_OUT * k;
        for (int c = c0; c < c1; ++c) {
            SyncBatch sb = next.get();
            if (c + 1 < c1) next = std::async(std::launch::async, parse_piece, c + 1);
            if (sb.L == 0) continue;
            if (sb.n != n) throw std::runtime_error("the number of pools in the sync file and in the phenotype file differ");
            const int64_t L = sb.L;
            totals[r] += L;
            if (L > cap_loci) {
                for (void *q : {(void *)counts_dev, (void *)n_out_dev, (void *)ids_dev, (void *)mf_dev, (void *)stat_dev, (void *)pv_dev})
                    if (q) (void)hipFree(q);
                cap_loci = L + L / 8;
                hip_ok(hipMalloc((void **)&counts_dev, sizeof(uint32_t) * (size_t)cap_loci * n * 6), "device memory for the counts");
                hip_ok(hipMalloc((void **)&n_out_dev, sizeof(int32_t) * cap_loci), "device memory");
                hip_ok(hipMalloc((void **)&ids_dev, sizeof(int32_t) * cap_loci * PG_MAX_OUT), "device memory");
                hip_ok(hipMalloc((void **)&mf_dev, sizeof(double) * cap_loci * PG_MAX_OUT), "device memory");
                hip_ok(hipMalloc((void **)&stat_dev, sizeof(double) * cap_loci * per_stat), "device memory");
                hip_ok(hipMalloc((void **)&pv_dev, sizeof(double) * cap_loci * per_stat), "device memory");
            }
            if (subset) { // rare: drop the pools without phenotype on the host (32-bit counts), then one copy
                counts2.resize((size_t)L * n2 * 6);
                for (int64_t l = 0; l < L; ++l)
                    for (int i = 0; i < n2; ++i) std::memcpy(&counts2[((size_t)l * n2 + i) * 6], &sb.counts[((size_t)l * n + keep[i]) * 6], 24);
                hip_ok(hipMemcpy(counts_dev, counts2.data(), sizeof(uint32_t) * counts2.size(), hipMemcpyHostToDevice), "H2D counts");
            } else
                upload(gpu, sb, counts_dev);
            if (mode == 0)
                gpu.ok(pg_chisq_batch_dev(gpu.c, counts_dev, L, n2, ps.data(), &flt, n_out_dev, ids_dev, stat_dev, pv_dev), "chisq_test");
            else if (mode == 1)
                gpu.ok(pg_pearson_batch_dev(gpu.c, counts_dev, L, n2, ps.data(), &flt, Y.data(), k, n_out_dev, ids_dev, mf_dev, stat_dev, pv_dev),
                       "pearson_corr");
            else
                gpu.ok(pg_ols_iter_batch_dev(gpu.c, counts_dev, L, n2, ps.data(), &flt, Y.data(), k, n_out_dev, ids_dev, mf_dev, stat_dev, pv_dev),
                       "ols_iter");
            n_out.resize(L); ids.resize((size_t)L * PG_MAX_OUT); mfq.resize((size_t)L * PG_MAX_OUT);
            stat.resize((size_t)L * per_stat); pv.resize((size_t)L * per_stat);
            hip_ok(hipMemcpy(n_out.data(), n_out_dev, sizeof(int32_t) * L, hipMemcpyDeviceToHost), "D2H results");
            // slot-major arrays: the slots any locus of the piece uses are a prefix of every array (one slot on biallelic data)
            int used = 0;
            for (int64_t l = 0; l < L; ++l) used = std::max(used, (int)n_out[l]);
            used = std::min(used, (int)PG_MAX_OUT);
            const size_t per_stat_used = mode == 0 ? 1 : (size_t)used * k;
            hip_ok(hipMemcpy(ids.data(), ids_dev, sizeof(int32_t) * L * used, hipMemcpyDeviceToHost), "D2H results");
            if (mode != 0) hip_ok(hipMemcpy(mfq.data(), mf_dev, sizeof(double) * L * used, hipMemcpyDeviceToHost), "D2H results");
            hip_ok(hipMemcpy(stat.data(), stat_dev, sizeof(double) * L * per_stat_used, hipMemcpyDeviceToHost), "D2H results");
            hip_ok(hipMemcpy(pv.data(), pv_dev, sizeof(double) * L * per_stat_used, hipMemcpyDeviceToHost), "D2H results");
            if (!fo) {
                fo = create_new(part[r]);
                if (R == 1) fputs(header, fo);
            }
            write_rows_parallel(fo, L, threads, [&](int64_t l, std::string &line) {
                // slot-major arrays: slot i of locus l sits i * L elements after its slot 0
                const size_t so = mode == 0 ? (size_t)l : (size_t)l * k;
                format_locus_rows(mode, sb.chrom(l), sb.pos[l], n_out[l], &ids[(size_t)l], &mfq[(size_t)l], &stat[so], &pv[so], k, line, (size_t)L);
            });
        }
        if (fo) fclose(fo);
        for (void *q : {(void *)counts_dev, (void *)n_out_dev, (void *)ids_dev, (void *)mf_dev, (void *)stat_dev, (void *)pv_dev})
            if (q) (void)hipFree(q);
        for (auto &sl : slot) if (sl.p) (void)hipHostFree(sl.p);
    });
    } catch (...) { cleanup_parts(); throw; }
    int64_t total = 0;
    for (int64_t t : totals) total += t;
    if (total == 0) { cleanup_parts(); throw std::runtime_error("no loci in " + a.fname); }
    if (R > 1) { // header + the ranks' parts in rank order = file order
        FILE *fo = create_new(out);
        fputs(header, fo);
        std::vector<char> buf(8 << 20);
        for (int r = 0; r < R; ++r) {
            if (totals[r] == 0) continue;
            FILE *fi = std::fopen(part[r].c_str(), "rb");
            if (!fi) { fclose(fo); cleanup_parts(); throw std::runtime_error("cannot reopen " + part[r]); }
            size_t got;
            while ((got = std::fread(buf.data(), 1, buf.size(), fi)) > 0)
                if (std::fwrite(buf.data(), 1, got, fo) != got) { fclose(fi); fclose(fo); cleanup_parts(); throw std::runtime_error("write failed (disk full?)"); }
            fclose(fi);
        }
        fclose(fo);
        cleanup_parts();
    }
    lap("pieces: parse | H2D + operator + D2H + format + write");
    std::cout << out << "\n"; // main.rs:507
    return done_ok();
}

// fst (popgen/fst.rs:10-261) and heterozygosity = pi (popgen/pi.rs:115-190) on the loaded matrix: loci and windows on
// the host (count_loci, define_sliding_windows), the per-locus arithmetic and the means on the GPU, the files as written
// by the reference.
static int run_popgen(const Args &a, bool is_fst, Ctx &gpu, const double *G_dev, const double *cov_dev, int64_t p, int n,
                      int64_t ld, const std::vector<std::string> &lab_chr, const std::vector<uint64_t> &lab_pos,
                      const std::vector<std::string> &pool_names, Lap &lap) {
    // count_loci (sync.rs:73-97) without the intercept entry: column starts, and each locus' coordinates
    std::vector<int64_t> locus_col;
    std::vector<int32_t> chr_id;
    std::vector<uint64_t> loc_pos;
    std::vector<std::string> loc_chr;
    for (int64_t c = 0; c < p; ++c)
        if (c == 0 || lab_chr[c] != lab_chr[c + 1] || lab_pos[c] != lab_pos[c + 1]) { // labels carry the intercept at [0]
            locus_col.push_back(c);
            if (!loc_chr.empty() && loc_chr.back() == lab_chr[c + 1]) chr_id.push_back(chr_id.back());
            else chr_id.push_back(chr_id.empty() ? 0 : chr_id.back() + 1);
            loc_chr.push_back(lab_chr[c + 1]);
            loc_pos.push_back(lab_pos[c + 1]);
        }
    locus_col.push_back(p);
    const int64_t L = (int64_t)loc_pos.size();
    std::vector<int64_t> wh(L), wt(L);
    const int64_t nw = pg_host_sliding_windows(chr_id.data(), loc_pos.data(), L, a.window_size_bp, a.window_slide_size_bp,
                                               a.min_loci_per_window, wh.data(), wt.data());
    wh.resize(nw); wt.resize(nw);
    const std::string win = std::to_string(a.window_size_bp);
    const std::string time = unix_time_string();
    if (!is_fst) {
        std::string out = a.output;
        if (out.empty()) out = basename_no_ext(a.fname) + "-pi-" + win + "_bp_windows-" + time + ".csv"; // pi.rs:135-159
        std::vector<double> pw((size_t)nw * n), pm(n);
        gpu.ok(pg_pi_dev(gpu.c, G_dev, cov_dev, p, n, ld, locus_col.data(), L, wh.data(), wt.data(), nw, pw.data(), pm.data()),
               "heterozygosity");
        lap("pi on the GPU");
        FILE *fo = create_new(out);
        std::string line = "Pool,Mean_across_windows";
        for (int64_t w = 0; w < nw; ++w)
            line += ",Window-" + loc_chr[wh[w]] + "_" + std::to_string(loc_pos[wh[w]]) + "_" + std::to_string(loc_pos[wt[w]]);
        fputs((line + "\n").c_str(), fo);
        for (int i = 0; i < n; ++i) {
            line = pool_names[i] + "," + rust_display(pm[i]);
            for (int64_t w = 0; w < nw; ++w) line += "," + roundup_own(pw[(size_t)w * n + i], 8);
            fputs((line + "\n").c_str(), fo);
        }
        fclose(fo);
        lap("write CSV");
        std::cout << out << "\n";
        return done_ok();
    }
    std::string out = a.output, out_win;
    if (out.empty()) { // fst.rs:93-131
        out = basename_no_ext(a.fname) + "-fst-averaged_across_genome-" + time + ".csv";
        out_win = basename_no_ext(a.fname) + "-fst-" + win + "_bp_windows-" + time + ".csv";
    } else
        out_win = basename_no_ext(out) + "-fst-" + win + "_bp_windows.csv";
    const size_t nn = (size_t)n * n;
    std::vector<double> mean(nn), fw((size_t)nw * nn);
    gpu.ok(pg_fst_dev(gpu.c, G_dev, cov_dev, p, n, ld, locus_col.data(), L, wh.data(), wt.data(), nw, mean.data(), fw.data()), "fst");
    lap("fst on the GPU");
    FILE *fo = create_new(out);
    std::string line;
    for (int i = 0; i < n; ++i) line += "," + pool_names[i];
    fputs((line + "\n").c_str(), fo);
    for (int i = 0; i < n; ++i) {
        line = pool_names[i];
        for (int j = 0; j < n; ++j) line += "," + roundup_own(mean[(size_t)i * n + j], 8);
        fputs((line + "\n").c_str(), fo);
    }
    fclose(fo);
    if (nw <= 0) // fst.rs:180, after the genome-wide file has been written
        throw std::runtime_error("There were no windows defined. Please check the sync file, the window size, slide size, and the minimum number of loci per window.");
    fo = create_new(out_win);
    line = "chr,pos_ini,pos_fin";
    for (int j = 0; j < n; ++j)
        for (int k2 = 0; k2 < n; ++k2) line += "," + pool_names[j] + "_vs_" + pool_names[k2];
    fputs((line + "\n").c_str(), fo);
    write_rows_parallel(fo, nw, a.n_threads, [&](int64_t w, std::string &text) {
        text += loc_chr[wh[w]] + "," + std::to_string(loc_pos[wh[w]]) + "," + std::to_string(loc_pos[wt[w]]);
        for (size_t q = 0; q < nn; ++q) text += "," + rust_display(fw[(size_t)w * nn + q]);
        text += "\n";
    });
    fclose(fo);
    lap("write CSV");
    std::cout << out << " and " << out_win << "\n"; // main.rs:441
    return done_ok();
}

// GenotypesAndPhenotypes (base/structs_and_traits.rs:139-148) with the matrix resident in HBM: the reference's dense
// n x (1 + p) `intercept_and_allele_frequencies` is here p x ld locus-major on the device, the column of ones implied; the
// three label vectors keep the leading "intercept" entry (sync.rs:1121-1126).
struct GenotypesAndPhenotypes {
    std::vector<std::string> chromosome, allele; // 1 + p entries
    std::vector<uint64_t> position;
    double *intercept_and_allele_frequencies = nullptr; // device
    int64_t p = 0, ld = 0;
    int n = 0, k = 0;
    std::vector<double> phenotypes;                     // n x k
    std::vector<std::string> pool_names;
    double *coverages = nullptr;                        // device, laid out like the matrix (pg_load_emit_cov_dev); optional
    GenotypesAndPhenotypes() = default;
    GenotypesAndPhenotypes(const GenotypesAndPhenotypes &) = delete;
    GenotypesAndPhenotypes &operator=(const GenotypesAndPhenotypes &) = delete;
    GenotypesAndPhenotypes(GenotypesAndPhenotypes &&o) noexcept { *this = std::move(o); }
    GenotypesAndPhenotypes &operator=(GenotypesAndPhenotypes &&o) noexcept {
        std::swap(chromosome, o.chromosome); std::swap(allele, o.allele); std::swap(position, o.position);
        std::swap(intercept_and_allele_frequencies, o.intercept_and_allele_frequencies); std::swap(p, o.p); std::swap(ld, o.ld);
        std::swap(n, o.n); std::swap(k, o.k); std::swap(phenotypes, o.phenotypes); std::swap(pool_names, o.pool_names);
        std::swap(coverages, o.coverages);
        return *this;
    }
    ~GenotypesAndPhenotypes() { (void)hipFree(intercept_and_allele_frequencies); (void)hipFree(coverages); }
};

// FileSyncPhen::into_genotypes_and_phenotypes (base/sync.rs:1106-1179) = load (:1044-1104: filter + frequencies per locus,
// sorted by (chromosome, position)) + the dense fill.  The host only sorts the locus order; filter, frequencies and the
// column layout run on the GPU from the parsed counts (pg_load_plan_dev / pg_load_emit[_cov]_dev), and the matrix never
// exists in host memory.  `remove_missing`: drop the pools without phenotype (what ols_with_covariate does first,
// gwas/ols.rs:287); the popgen tools keep every pool.
static GenotypesAndPhenotypes into_genotypes_and_phenotypes(Ctx &gpu, const SyncBatch &sb, const Phen &ph, const pg_filter &flt,
                                                            bool keep_p_minus_1, bool remove_missing, bool with_coverages, Lap &lap) {
    auto hip_ok = [](hipError_t e, const char *what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    };
    const int n = sb.n, k = ph.k;
    const int64_t L = sb.size();
    std::vector<int64_t> order(L);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
        const int c = sb.chrom(x).compare(sb.chrom(y));
        return c != 0 ? c < 0 : sb.pos[x] < sb.pos[y];
    });
    std::vector<int> keep(n);
    std::iota(keep.begin(), keep.end(), 0);
    if (remove_missing) keep = complete_pools(ph);
    if (keep.empty()) throw std::runtime_error("All pools have missing data. Please check the phenotype file.");
    GenotypesAndPhenotypes g;
    g.n = (int)keep.size();
    g.k = k;
    g.ld = g.n + (g.n & 1);
    std::vector<int32_t> pool_map(n, -1);
    for (int i = 0; i < g.n; ++i) pool_map[keep[i]] = i;
    uint32_t *counts_dev = nullptr;
    int64_t *order_dev = nullptr;
    hip_ok(hipMalloc((void **)&counts_dev, sizeof(uint32_t) * (size_t)L * n * 6), "device memory for the counts");
    hip_ok(hipMalloc((void **)&order_dev, sizeof(int64_t) * L), "device memory");
    CountsUpload upload;
    upload(gpu, sb, counts_dev);
    hip_ok(hipMemcpy(order_dev, order.data(), sizeof(int64_t) * L, hipMemcpyHostToDevice), "H2D order");
    gpu.ok(pg_load_plan_dev(gpu.c, counts_dev, L, n, ph.pool_sizes.data(), &flt, keep_p_minus_1 ? 1 : 0, order_dev, &g.p), "load");
    if (g.p <= 0) throw std::runtime_error("no loci passed the filters");
    int64_t *col_locus_dev = nullptr;
    int32_t *col_allele_dev = nullptr;
    hip_ok(hipMalloc((void **)&g.intercept_and_allele_frequencies, sizeof(double) * (size_t)g.p * g.ld), "device memory for the genotype matrix");
    hip_ok(hipMalloc((void **)&col_locus_dev, sizeof(int64_t) * g.p), "device memory");
    hip_ok(hipMalloc((void **)&col_allele_dev, sizeof(int32_t) * g.p), "device memory");
    if (with_coverages) {
        hip_ok(hipMalloc((void **)&g.coverages, sizeof(double) * (size_t)g.p * g.ld), "device memory for the coverages");
        gpu.ok(pg_load_emit_cov_dev(gpu.c, pool_map.data(), g.n, g.intercept_and_allele_frequencies, g.ld, col_locus_dev, col_allele_dev,
                                    g.coverages), "load");
    } else
        gpu.ok(pg_load_emit_dev(gpu.c, pool_map.data(), g.n, g.intercept_and_allele_frequencies, g.ld, col_locus_dev, col_allele_dev), "load");
    lap("sort + H2D + GPU loader");
    hip_ok(hipFree(counts_dev), "free");
    hip_ok(hipFree(order_dev), "free");
    std::vector<int64_t> col_locus(g.p);
    std::vector<int32_t> col_allele(g.p);
    hip_ok(hipMemcpy(col_locus.data(), col_locus_dev, sizeof(int64_t) * g.p, hipMemcpyDeviceToHost), "D2H labels");
    hip_ok(hipMemcpy(col_allele.data(), col_allele_dev, sizeof(int32_t) * g.p, hipMemcpyDeviceToHost), "D2H labels");
    (void)hipFree(col_locus_dev); (void)hipFree(col_allele_dev);
    g.chromosome.assign(1, "intercept"); g.allele.assign(1, "intercept"); g.position.assign(1, 0);
    g.chromosome.reserve(g.p + 1); g.allele.reserve(g.p + 1); g.position.reserve(g.p + 1);
    for (int64_t c = 0; c < g.p; ++c) {
        g.chromosome.push_back(sb.chrom(col_locus[c])); g.position.push_back(sb.pos[col_locus[c]]);
        g.allele.push_back(std::string(1, ALLELES[col_allele[c]]));
    }
    for (int i : keep) {
        g.pool_names.push_back(ph.pool_names[i]);
        for (int j = 0; j < k; ++j) g.phenotypes.push_back(ph.phen[(size_t)i * k + j]);
    }
    return g;
}

// gwas::ols_with_covariate (gwas/ols.rs:278-436): kinship, eigen rule, one fit per (column, trait), the CSV; returns the
// name of the file it wrote.
// ... and gwas::mle_with_covariate (gwas/mle.rs:307-463) when `mle`: same preamble, Nelder-Mead fits, same writer with its own file name
// (the reference also prints its index arrays, the covariates, y, g, beta and pval to stdout, mle.rs:364-368, :402-405: debug output,
// not reproduced).
static std::string ols_with_covariate(Ctx &gpu, GenotypesAndPhenotypes &g, double xxt_eigen_variance_explained,
                                      const std::string &fname_input, const std::string &fname_output, int n_threads, Lap &lap,
                                      bool mle = false) {
    auto hip_ok = [](hipError_t e, const char *what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    };
    const int64_t p = g.p;
    const int k = g.k;
    if (!fname_output.empty()) { FILE *t = create_new(fname_output); fclose(t); ::unlink(fname_output.c_str()); } // ols.rs:285
    std::vector<double> beta((size_t)p * k), pval((size_t)p * k);
    int m = 0;
    double *out_dev = nullptr;
    hip_ok(hipMalloc((void **)&out_dev, sizeof(double) * 3 * (size_t)p * k), "device memory for the results");
    if (mle)
        gpu.ok(pg_mle_kinship_dev(gpu.c, g.intercept_and_allele_frequencies, p, g.n, g.ld, g.phenotypes.data(), k, xxt_eigen_variance_explained, -1,
                                  &m, nullptr, out_dev, out_dev + (size_t)p * k, out_dev + 2 * (size_t)p * k), "mle_iter_with_kinship");
    else
    gpu.ok(pg_ols_kinship_dev(gpu.c, g.intercept_and_allele_frequencies, p, g.n, g.ld, g.phenotypes.data(), k, xxt_eigen_variance_explained, -1,
                              &m, nullptr, out_dev, out_dev + (size_t)p * k, out_dev + 2 * (size_t)p * k), "ols_iter_with_kinship");
    gpu.ok(pg_synchronize(gpu.c), "ols_iter_with_kinship");
    hip_ok(hipMemcpy(beta.data(), out_dev, sizeof(double) * (size_t)p * k, hipMemcpyDeviceToHost), "D2H results");
    hip_ok(hipMemcpy(pval.data(), out_dev + 2 * (size_t)p * k, sizeof(double) * (size_t)p * k, hipMemcpyDeviceToHost), "D2H results");
    (void)hipFree(out_dev);
    lap("kinship + fits + D2H");
    std::string out = fname_output;
    if (out.empty()) // ols.rs:393-398
        out = basename_no_ext(fname_input) + (mle ? "-mle_iterative_xxt_" : "-ols_iterative_xxt_") + std::to_string(m + 1) + "_eigens-" +
              unix_time_string() + ".csv"; // ols.rs:393-398, mle.rs:423-427
    FILE *fo = create_new(out);
    fputs("#chr,pos,alleles,phenotype,statistic,pvalue\n", fo); // ols.rs:409
    write_rows_parallel(fo, (int64_t)k * p, n_threads, [&](int64_t r, std::string &text) {
        const int64_t j = r / p, i = r - j * p; // rows are trait-major (ols.rs:411-433)
        // the reference labels coefficient i with entry i of the (1+p)-long label vectors, i.e.
        // shifted by the intercept entry (ols.rs:421-425) -- reproduced as is
        text += g.chromosome[i]; text.push_back(','); text += std::to_string(g.position[i]); text.push_back(','); text += g.allele[i];
        text += ",Pheno_"; text += std::to_string(j); text.push_back(',');
        append_rust_display(text, beta[(size_t)i * k + j]); text.push_back(',');
        append_rust_display(text, pval[(size_t)i * k + j]); text.push_back('\n');
    });
    fclose(fo);
    lap("format + write CSV");
    return out;
}

static int run(int argc, char **argv) {
    stamp("main");
    const Args a = parse_args(argc, argv);
    Lap lap;
    const std::map<std::string, int> known{{"chisq_test", 0}, {"pearson_corr", 1}, {"ols_iter", 2},
                                           {"ols_iter_with_kinship", 3}, {"pileup2sync", 4},
                                           {"genomic_prediction_cross_validation", 5}, {"fst", 6}, {"heterozygosity", 7},
                                           {"mle_iter_with_kinship", 8}};
    if (!known.count(a.analysis))
        throw std::runtime_error("Invalid analysis utility for this build: `" + a.analysis +
                                 "` (available: pileup2sync, chisq_test, pearson_corr, ols_iter, ols_iter_with_kinship, "
                                 "mle_iter_with_kinship, genomic_prediction_cross_validation, fst, heterozygosity)");
    if (a.generate_plots || a.sig_only)
        throw std::runtime_error("--generate-plots / --output-sig-snps-only call the reference's python scripts and are out of scope here");
    Phen ph = parse_phen(a.phen_fname, a.phen_delim, a.phen_name_col, a.phen_pool_size_col, a.phen_value_col);
    // the counts are parsed straight into pinned memory: the copy to the device needs no staging pass
    SyncAlloc pinned;
    pinned.alloc = [](size_t bytes) -> void * {
        void *p = nullptr;
        return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
    };
    pinned.release = [](void *p) { (void)hipHostFree(p); };
    if (a.analysis == "pileup2sync") { // main.rs:212-225; text to text, no GPU involved
        PileupFilter pf;
        pf.remove_ns = !a.keep_ns;
        pf.keep_lowercase_reference = a.keep_lowercase_reference;
        pf.max_base_error_rate = a.max_base_error_rate;
        pf.min_coverage_depth = a.min_coverage_depth;
        pf.min_coverage_breadth = a.min_coverage_breadth;
        pf.min_allele_frequency = a.min_allele_frequency;
        pf.pool_sizes = ph.pool_sizes;
        std::string out = a.output;
        if (out.empty()) out = basename_no_ext(a.fname) + "-" + unix_time_string() + ".sync"; // pileup.rs:478-494
        const int64_t kept = pileup_to_sync_file(a.fname, ph.pool_names, pf, out, a.n_threads);
        lap("pileup2sync");
        std::cerr << kept << " loci written\n";
        std::cout << out << "\n"; // main.rs:507
        return done_ok();
    }
    if (a.n_gpus > 0 && known.at(a.analysis) > 3)
        throw std::runtime_error("--n-gpus applies to chisq_test, pearson_corr, ols_iter and ols_iter_with_kinship; `" + a.analysis + "` runs on one GPU");
    const RankSetup ranks = rank_setup(a, a.analysis == "ols_iter_with_kinship");
    Ctx gpu(ranks.devices[0]); // first: the pinned allocator below needs a HIP context
    lap("start-up");
    // A pileup input (*.pileup / *.mpileup) is converted in memory -- the counts pileup2sync would write and the
    // sync reader would read back, without the text in between (an extension: the reference needs the sync file).
    auto ends_with = [](const std::string &x, const char *suf) {
        const size_t m = std::strlen(suf);
        return x.size() >= m && x.compare(x.size() - m, m, suf) == 0;
    };
    const bool is_pileup = ends_with(a.fname, ".pileup") || ends_with(a.fname, ".mpileup");
    PileupFilter pf;
    pf.remove_ns = !a.keep_ns;
    pf.keep_lowercase_reference = a.keep_lowercase_reference;
    pf.max_base_error_rate = a.max_base_error_rate;
    pf.min_coverage_depth = a.min_coverage_depth;
    pf.min_coverage_breadth = a.min_coverage_breadth;
    pf.min_allele_frequency = a.min_allele_frequency;
    pf.pool_sizes = ph.pool_sizes;
    pg_filter flt{};
    flt.remove_ns = a.keep_ns ? 0 : 1;
    flt.min_coverage_depth = a.min_coverage_depth;
    flt.min_allele_frequency = a.min_allele_frequency;
    flt.max_missingness_rate = a.max_missingness_rate;
    if (a.analysis == "ols_iter_with_kinship") {
        // inputs above 256 MiB are taken in pieces of 128 MiB of text, above 1 GiB in pieces of 256 MiB (parse of piece c + 1 overlaps the GPU work on
        // piece c, and the pinned buffers stay small); an unsorted input falls back to the whole-file path unless
        // the pieces were asked for explicitly
        struct stat st;
        const size_t fsize = ::stat(a.fname.c_str(), &st) == 0 ? (size_t)st.st_size : 0;
        const bool automatic = a.stream_chunk_mb < 0 && !std::getenv("PGH_STREAM_CHUNK_BYTES");
        long mb = a.stream_chunk_mb;
        if (mb < 0) mb = fsize > ((size_t)1 << 30) ? 256 : (fsize > ((size_t)256 << 20) ? 128 : 0); // no whole-file pinned buffer beyond 256 MiB
        size_t piece = (size_t)(mb > 0 ? mb : 0) << 20;
        if (const char *e = std::getenv("PGH_STREAM_CHUNK_BYTES")) piece = (size_t)std::strtoull(e, nullptr, 10); // tests: small pieces
        if (a.n_gpus > 0 && piece == 0) piece = std::max<size_t>(1, (fsize + ranks.n_ranks - 1) / ranks.n_ranks); // one piece per rank at least
        if (piece > 0 && (fsize > piece || a.n_gpus > 0)) {
            try {
                return run_kinship_streamed(a, ph, gpu, lap, piece, is_pileup, pf, flt, ranks);
            } catch (const UnsortedInput &e) {
                if (!automatic || a.n_gpus > 0) throw;
                std::cerr << "note: input is not sorted by (chromosome, position); loading the whole file instead\n";
            }
        }
    }
    if (known.at(a.analysis) <= 2) {
        size_t piece = (size_t)(a.stream_chunk_mb > 0 ? a.stream_chunk_mb : 128) << 20;
        if (const char *e = std::getenv("PGH_STREAM_CHUNK_BYTES")) piece = (size_t)std::strtoull(e, nullptr, 10); // tests: small pieces
        return run_batch_streamed(a, ph, gpu, lap, known.at(a.analysis), piece, is_pileup, pf, flt, ranks);
    }
    SyncBatch sb;
    if (is_pileup) {
        sb = parse_pileup_file(a.fname, a.n_threads, pf, pinned);
        lap("pileup -> counts");
    } else {
        // the analyses on the loaded matrix copy the counts themselves: 16-bit counts when they fit (the batch operators'
        // host-buffer entry points take the 32-bit layout)
        sb = parse_sync_file(a.fname, a.n_threads, pinned, known.at(a.analysis) >= 3);
        lap("parse sync");
    }
    if (sb.size() == 0) throw std::runtime_error("no loci in " + a.fname);
    if (sb.n != ph.n) throw std::runtime_error("the number of pools in the sync file and in the phenotype file differ");
    const int k = ph.k;
    const int mode = known.at(a.analysis);

    // ---------------- the analyses on the loaded matrix (main.rs:280-298, :397-455) ---------------------------
    const bool kpm1 = mode == 7 ? false : a.keep_p_minus_1; // heterozygosity: "we need all alleles in each locus" (main.rs:445)
    GenotypesAndPhenotypes genotypes_and_phenotypes =
        into_genotypes_and_phenotypes(gpu, sb, ph, flt, kpm1, /*remove_missing=*/mode < 6 || mode == 8, /*with_coverages=*/mode == 6 || mode == 7, lap);
    GenotypesAndPhenotypes &g = genotypes_and_phenotypes;
    if (mode == 6 || mode == 7) // fst / heterozygosity use every pool (main.rs:427-455)
        return run_popgen(a, mode == 6, gpu, g.intercept_and_allele_frequencies, g.coverages, g.p, g.n, g.ld, g.chromosome, g.position,
                          g.pool_names, lap);
    if (mode == 5) { // genomic_prediction_cross_validation (main.rs:397-426)
        CvLabels labels{g.chromosome, g.allele, g.position};
        CvArgs ca;
        ca.k_folds = a.k_folds; ca.n_reps = a.n_reps; ca.seed = a.seed; ca.n_threads = a.n_threads;
        ca.fname_input = a.fname; ca.fname_output = a.output;
        const std::string out = gp_cross_validate(gpu.c, g.intercept_and_allele_frequencies, g.p, g.n, g.ld, g.phenotypes, k, g.pool_names,
                                                  labels, ca);
        lap("cross-validation");
        std::cout << out << "\n";
        return done_ok();
    }
    const std::string out = ols_with_covariate(gpu, g, a.xxt, a.fname, a.output, a.n_threads, lap, /*mle=*/mode == 8);
    std::cout << out << "\n";
    return done_ok();
}

int main(int argc, char **argv) {
    try {
        return run(argc, argv);
    } catch (const std::exception &e) {
        std::cerr << "poolgen: " << e.what() << "\n";
        return 1;
    }
}
