// mb_pattern.hip -- what HBM read rate does a given ACCESS PATTERN reach on MI355X, with the sweep's occupancy (8 waves per CU,
// 16 KB in flight per wave) and no arithmetic?  Prices the ceiling of k_ols_sweep's staging scheme against alternatives.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb_pattern.hip -o tools/mb_pattern      Run: tools/mb_pattern [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A wave tile = ROWS rows of `rowlen` bytes, rows `stride` bytes apart, read in chunks of PIECE bytes per row:
// per chunk ROWS * PIECE / 1024 load instructions (64 lanes x 16 B), all issued before the first is consumed.
// DEPTH = chunks in flight (1 = wait for a chunk before issuing the next, 2 = one chunk of prefetch).
template <int PIECE, int ROWS, int DEPTH>
__global__ __launch_bounds__(256) void k_pattern(const char *__restrict__ base, long long ntiles, long long rowlen, long long stride,
                                                 long long misalign, double *out) {
    extern __shared__ double lds[]; // only to pin the occupancy at 2 blocks per CU
    constexpr int LPR = PIECE / 16;       // lanes per row piece
    constexpr int RPI = 64 / LPR;         // rows per load instruction
    constexpr int NLD = ROWS / RPI;       // load instructions per chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane / LPR, piece = lane % LPR;
    const long long wstride = (long long)gridDim.x * 4;
    const int nch = (int)((rowlen + PIECE - 1) / PIECE);
    double s = 0.0;
    double2 v[DEPTH][NLD];
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += wstride) {
        const char *tb = base + t * ROWS * stride + misalign;
        for (int ch = 0; ch < nch + DEPTH - 1; ++ch) {
            if (ch < nch) {
#pragma unroll
                for (int r = 0; r < NLD; ++r) {
                    long long in_row = (long long)ch * PIECE + 16 * piece;
                    in_row = in_row < rowlen - 16 ? in_row : rowlen - 16; // lanes past the row's end re-read its last 16 bytes
                    long long off = (long long)(RPI * r + lr) * stride + in_row;
                    v[ch % DEPTH][r] = *reinterpret_cast<const double2 *>(tb + off);
                }
            }
            if (ch >= DEPTH - 1) {
#pragma unroll
                for (int r = 0; r < NLD; ++r) s += v[(ch - (DEPTH - 1)) % DEPTH][r].x + v[(ch - (DEPTH - 1)) % DEPTH][r].y;
            }
        }
    }
    if (s == 12345.678) out[0] = s + lds[0];
}

// k_locus_first's shape: ONE 128-byte line per row per stage (8 lanes per row, 8 rows per instruction, 8 instructions per
// stage of 64 rows), NST stages in flight
template <int NST>
__global__ __launch_bounds__(256) void k_lines(const char *__restrict__ base, long long ntiles, long long rowlen, long long stride, double *out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 3, piece = lane & 7;
    const long long wstride = (long long)gridDim.x * 4;
    const int nst = (int)((rowlen + 127) / 128);
    double s = 0.0;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += wstride) {
        const char *tb = base + t * 64 * stride;
        double2 a[8], b[8], c[8];
        auto issue = [&](double2 (&v)[8], int st) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                long long in_row = (long long)st * 128 + 16 * piece;
                in_row = in_row < rowlen - 16 ? in_row : rowlen - 16;
                v[r] = *reinterpret_cast<const double2 *>(tb + (long long)(8 * r + lr) * stride + in_row);
            }
        };
        auto use = [&](const double2 (&v)[8]) {
#pragma unroll
            for (int r = 0; r < 8; ++r) s += v[r].x + v[r].y;
        };
        if (NST == 1) { for (int st = 0; st < nst; ++st) { issue(a, st); use(a); } }
        else if (NST == 2) { issue(a, 0); for (int st = 0; st < nst; st += 2) { if (st + 1 < nst) issue(b, st + 1); use(a); if (st + 2 < nst) issue(a, st + 2); if (st + 1 < nst) use(b); } }
        else { issue(a, 0); if (nst > 1) issue(b, 1);
            for (int st = 0; st < nst; st += 3) {
                if (st + 2 < nst) issue(c, st + 2); use(a);
                if (st + 3 < nst) issue(a, st + 3); if (st + 1 < nst) use(b);
                if (st + 4 < nst) issue(b, st + 4); if (st + 2 < nst) use(c);
            } }
    }
    if (s == 12345.678) out[0] = s + lds[0];
}

// the same in-flight structure on a CONTIGUOUS slab: 16 consecutive 1 KB wave loads per step
__global__ __launch_bounds__(256) void k_linear16(const char *__restrict__ base, long long nsteps, double *out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wstride = (long long)gridDim.x * 4;
    double s = 0.0;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < nsteps; t += wstride) {
        double2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = *reinterpret_cast<const double2 *>(base + t * 16384 + r * 1024 + lane * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[r].x + v[r].y;
    }
    if (s == 12345.678) out[0] = s + lds[0];
}

template <typename K, typename... A>
static float timeit(K kern, int grid, size_t shmem, A... args) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, args...);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    char *buf; double *out;
    CK(hipMalloc(&buf, bytes + (1 << 20)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes + (1 << 20)));
    const size_t shmem = 69632; // 2 blocks of 4 waves per CU, as k_ols_sweep
    const int grid = cus * 8;
    auto rate = [&](float ms, double b) { return b / (ms * 1e-3) / 1e9; };
    printf("device %s, %d CUs; %.1f GiB buffer; 8 waves per CU\n", prop.gcnArchName, cus, gib);
    {
        float ms = timeit(k_linear16, grid, shmem, (const char *)buf, (long long)(bytes / 16384), out);
        printf("%-64s %8.3f ms %8.1f GB/s\n", "contiguous: 16 x 1 KB per wave step", ms, rate(ms, (double)bytes));
    }
    for (long long rl : {4800LL, 2400LL}) { // counts rows: 24 n bytes, n = 200 / 100
        const long long nt = (long long)(bytes / (64 * rl));
        const double useful = (double)nt * 64 * rl;
        float ms;
        ms = timeit(k_lines<1>, grid, shmem, (const char *)buf, nt, rl, rl, out);
        printf("counts rows %4lld B, one 128-B line per row per stage, 1 stage  in flight: %8.3f ms %8.1f GB/s\n", rl, ms, rate(ms, useful));
        ms = timeit(k_lines<2>, grid, shmem, (const char *)buf, nt, rl, rl, out);
        printf("counts rows %4lld B, one 128-B line per row per stage, 2 stages in flight: %8.3f ms %8.1f GB/s\n", rl, ms, rate(ms, useful));
        ms = timeit(k_lines<3>, grid, shmem, (const char *)buf, nt, rl, rl, out);
        printf("counts rows %4lld B, one 128-B line per row per stage, 3 stages in flight: %8.3f ms %8.1f GB/s\n", rl, ms, rate(ms, useful));
        ms = timeit(k_pattern<256, 64, 1>, grid, shmem, (const char *)buf, nt, rl, rl, 0LL, out);
        printf("counts rows %4lld B, 256-B pieces x 64 rows, 1 chunk in flight:            %8.3f ms %8.1f GB/s\n", rl, ms, rate(ms, useful));
    }
    struct Case { const char *name; long long rowlen, stride, mis; };
    const Case cases[] = {
        {"rows 1600 B (n = 200), per-row grid, rows 64 B off (v1)", 1600, 1600, 0},
        {"super-rows 3200 B (2 x n = 200), line aligned (v2)", 3200, 3200, 0},
        {"rows 1664 B (ld = 208), line aligned", 1664, 1664, 0},
        {"rows 2048 B, 256-B aligned", 2048, 2048, 0},
        {"super-rows 3200 B, pieces 256-B aligned? (base + 128)", 3200, 3200, 128},
        {"rows 800 B x 4 = 3200 B super-rows (n = 100)", 3200, 3200, 0},
    };
    for (const Case &c : cases) {
        const long long ntiles64 = (long long)(bytes / (64 * c.stride));
        const double useful = (double)ntiles64 * 64 * c.rowlen;
        float ms;
        ms = timeit(k_pattern<256, 64, 1>, grid, shmem, (const char *)buf, ntiles64, c.rowlen, c.stride, c.mis, out);
        printf("%-64s piece  256 B x 64 rows, depth 1: %8.3f ms %8.1f GB/s\n", c.name, ms, rate(ms, useful));
        ms = timeit(k_pattern<256, 64, 2>, grid, shmem, (const char *)buf, ntiles64, c.rowlen, c.stride, c.mis, out);
        printf("%-64s piece  256 B x 64 rows, depth 2: %8.3f ms %8.1f GB/s\n", c.name, ms, rate(ms, useful));
        ms = timeit(k_pattern<512, 32, 1>, grid, shmem, (const char *)buf, ntiles64 * 2, c.rowlen, c.stride, c.mis, out);
        printf("%-64s piece  512 B x 32 rows, depth 1: %8.3f ms %8.1f GB/s\n", c.name, ms, rate(ms, useful));
        ms = timeit(k_pattern<512, 64, 1>, grid, shmem, (const char *)buf, ntiles64, c.rowlen, c.stride, c.mis, out);
        printf("%-64s piece  512 B x 64 rows, depth 1: %8.3f ms %8.1f GB/s\n", c.name, ms, rate(ms, useful));
        ms = timeit(k_pattern<1024, 16, 1>, grid, shmem, (const char *)buf, ntiles64 * 4, c.rowlen, c.stride, c.mis, out);
        printf("%-64s piece 1024 B x 16 rows, depth 1: %8.3f ms %8.1f GB/s\n", c.name, ms, rate(ms, useful));
    }
    return 0;
}
