// tools/parse_bench.cpp -- the sync parser alone, piece by piece as the streamed CLI drives it (no GPU work):
//   parse_bench <file.sync> <threads> <piece MB> <malloc|pinned> [populate] [drop] [pread]
// pinned = hipHostMalloc'd output slots (what the CLI parses into), malloc = plain memory; populate = MADV_POPULATE_READ on the
// piece before it is parsed; drop = MADV_DONTNEED behind it; pread = the piece is first copied out of the page cache into a reused
// buffer by the worker threads (pread of equal shares), then parsed from there.  Build: hipcc -O2 -std=c++17 -Ipoolgen_amd/csrc/host tools/parse_bench.cpp
// poolgen_amd/csrc/host/host_util.cpp -o /tmp/parse_bench -lpthread
#include "host_util.h"
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
using namespace pgh;
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
int main(int argc, char **argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: parse_bench file threads pieceMB malloc|pinned [populate] [drop]\n"); return 2; }
    const int thr = std::atoi(argv[2]);
    const size_t piece = (size_t)std::atol(argv[3]) << 20;
    const bool pinned = std::string(argv[4]) == "pinned";
    bool populate = false, drop = false, use_pread = false;
    for (int i = 5; i < argc; ++i) { populate |= std::string(argv[i]) == "populate"; drop |= std::string(argv[i]) == "drop"; use_pread |= std::string(argv[i]) == "pread"; }
    const int fd = ::open(argv[1], O_RDONLY);
    std::vector<char> pbuf;
    MappedFile mf(argv[1]);
    const std::vector<size_t> cuts = mf.cuts((mf.size() + piece - 1) / piece);
    struct Slot { void *p = nullptr; size_t cap = 0; } slot[2];
    auto alloc_for = [&](int i) {
        SyncAlloc al;
        al.alloc = [&slot, i, pinned](size_t bytes) -> void * {
            if (bytes > slot[i].cap) {
                const size_t want = bytes + bytes / 8;
                if (pinned) { if (hipHostMalloc(&slot[i].p, want, hipHostMallocDefault) != hipSuccess) return nullptr; }
                else { slot[i].p = std::malloc(want); std::memset(slot[i].p, 1, want); }
                slot[i].cap = want;
            }
            return slot[i].p;
        };
        al.release = [](void *) {};
        return al;
    };
    const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
    auto t0 = std::chrono::steady_clock::now();
    int64_t loci = 0;
    for (size_t c = 0; c + 1 < cuts.size(); ++c) {
        const char *b = mf.data() + cuts[c], *e = mf.data() + cuts[c + 1];
        const uintptr_t lo = ((uintptr_t)b + pg - 1) / pg * pg, hi = (uintptr_t)e / pg * pg;
        if (populate && hi > lo) (void)madvise((void *)lo, hi - lo, MADV_POPULATE_READ);
        if (use_pread) {
            const size_t len = (size_t)(e - b);
            if (pbuf.size() < len + 64) pbuf.resize(len + 64);
            std::vector<std::thread> th;
            for (int t = 0; t < thr; ++t)
                th.emplace_back([&, t] {
                    const size_t lo2 = len * t / thr, hi2 = len * (t + 1) / thr;
                    size_t done = 0;
                    while (lo2 + done < hi2) {
                        const ssize_t r = ::pread(fd, pbuf.data() + lo2 + done, hi2 - lo2 - done, (off_t)(cuts[c] + lo2 + done));
                        if (r <= 0) break;
                        done += (size_t)r;
                    }
                });
            for (auto &x : th) x.join();
            b = pbuf.data(); e = pbuf.data() + len;
        }
        SyncBatch sb = parse_sync_buffer(b, e, thr, 0, alloc_for((int)(c & 1)), true);
        loci += sb.L;
        if (drop && hi > lo) (void)madvise((void *)lo, hi - lo, MADV_DONTNEED);
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%s threads %d piece %zu MB %s%s%s: %lld loci, %.3f s, %.2f GB/s\n", argv[1], thr, piece >> 20, argv[4], populate ? " populate" : "",
                (std::string(drop ? " drop" : "") + (use_pread ? " pread" : "")).c_str(), (long long)loci, s, mf.size() / s / 1e9);
    return 0;
}
