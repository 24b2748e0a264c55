#include "host_util.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/resource.h>
using namespace pgh;
static double cpu() { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + 1e-6 * r.ru_utime.tv_usec + r.ru_stime.tv_sec + 1e-6 * r.ru_stime.tv_usec; }
int main(int argc, char **argv) {
    const int thr = argc > 2 ? atoi(argv[2]) : 8;
    static void *keep = nullptr; static size_t keepsz = 0;
    SyncAlloc al; al.alloc = [](size_t b) -> void * { if (b > keepsz) { keep = malloc(b); memset(keep, 1, b); keepsz = b; } return keep; }; al.release = [](void *) {};
    MappedFile mf(argv[1]);
    { volatile char x = 0; for (size_t i = 0; i < mf.size(); i += 4096) x += mf.data()[i]; }
    for (int rep = 0; rep < 6; ++rep) {
        auto t0 = std::chrono::steady_clock::now(); double c0 = cpu();
        SyncBatch sb = parse_sync_buffer(mf.data(), mf.data() + mf.size(), thr, 0, al, true);
        auto t1 = std::chrono::steady_clock::now(); double c1 = cpu();
        std::printf("L %lld  wall %.3f s  cpu %.3f s\n", (long long)sb.L, std::chrono::duration<double>(t1 - t0).count(), c1 - c0);
    }
}
