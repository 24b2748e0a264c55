// mb_corun.hip -- does a light streaming kernel (<= 64 VGPRs, 4 waves per CU, no LDS to speak of) run BESIDE the kinship pass
// (one 1024-thread workgroup per CU, 110 VGPRs per lane = 448 of a SIMD's 512, 53 KB of LDS), or only once it has left?
// Kinship through the library's C ABI on one stream; the side kernel (per-row sums of G, 16 bytes per lane) on another.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb_corun.hip -o tools/mb_corun -Lpoolgen_amd/csrc -lpoolgen_hip -Wl,-rpath,'$ORIGIN/../poolgen_amd/csrc'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/poolgen_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one wave = 4 rows of 1600 bytes per iteration: 16 lanes per row, 16 bytes per lane and load, the row's sum by DPP-free shuffles
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_side(const double *__restrict__ G, long long p, int ld, double *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nw = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + wave;
    const int q = lane >> 4, r = lane & 15;
    for (long long l0 = w * 4; l0 < p; l0 += nw * 4) {
        const long long l = l0 + q;
        double s = 0.0;
        if (l < p) {
            const double2 *row = reinterpret_cast<const double2 *>(G + l * ld);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int c = r + 16 * j;
                if (2 * c < ld) { const double2 v = row[c]; s += v.x + v.y; }
            }
        }
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
        if (r == 0 && l < p) out[l] = s;
    }
}

int main() {
    const int n = 200; const long long p = 10000000; const int ld = 200;
    double *G, *S, *out;
    CK(hipMalloc(&G, sizeof(double) * p * ld)); CK(hipMalloc(&S, sizeof(double) * n * n)); CK(hipMalloc(&out, sizeof(double) * p));
    {
        std::vector<double> h((size_t)1 << 20);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
        for (long long o = 0; o < p * ld; o += (long long)h.size()) {
            const long long c = std::min<long long>(h.size(), p * ld - o);
            CK(hipMemcpy(G + o, h.data(), sizeof(double) * c, hipMemcpyHostToDevice));
        }
    }
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    pg_ctx *ctx = nullptr;
    if (pg_create(&ctx, 0, sa) != 0) { printf("pg_create: %s\n", pg_last_error(nullptr)); return 1; }
    pg_set_phenotypes(ctx, 0, nullptr, 0);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto kin = [&] { return pg_kinship_partial_dev(ctx, G, p, n, ld, S); };
    for (int minw : {8, 4}) {
        auto side = [&](hipStream_t st) {
            if (minw == 8) hipLaunchKernelGGL(k_side<8>, dim3(256), dim3(256), 0, st, G, p, ld, out);
            else hipLaunchKernelGGL(k_side<4>, dim3(512), dim3(256), 0, st, G, p, ld, out);
        };
        for (int mode = 0; mode < 3; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                if (mode != 1) { if (kin()) { printf("kinship: %s\n", pg_last_error(ctx)); return 1; } }
                if (mode != 0) side(sb);
                CK(hipDeviceSynchronize());
                const double dt = (now() - t0) * 1e3;
                if (rep > 0 && dt < best) best = dt;
            }
            printf("side kernel build for %d waves/SIMD: %s %.3f ms\n", minw, mode == 0 ? "kinship alone      " : mode == 1 ? "side kernel alone  " : "both, two streams  ", best);
        }
    }
    return 0;
}
