// mb_msweep.hip -- prototype of the inner loop of an MFMA sweep: W'g for 16 loci x 16 columns per v_mfma_f64_16x16x4_f64 chain
// (A = 16 loci x 4 pools of G, B = 4 pools x 16 columns of W held in registers for the whole kernel), g'g on the vector ALU
// (one FMA per element), no LDS transposition.  What it prices: the HBM read rate of the A-operand access pattern (16 rows per
// load instruction, 16 B or 32 B per lane) with the MFMA and FMA work in place, against k_ols_sweep's coalesced + LDS staging.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb_msweep.hip -o tools/mb_msweep      Run: tools/mb_msweep [GiB]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: lane (i = l & 15, k = l >> 4) loads 16 B = pools 8 c + 2 k + {0, 1} of locus i          (64 B of a row per instruction)
// MODE 1: the same lane loads 32 B = pools 16 c + 4 k + {0 .. 3} in two instructions                   (128 B of a row per pair)
// NC = chunks per row (MODE 0: n / 8, MODE 1: ceil(n / 16)); all NC (x2) loads of the NEXT tile are issued as the registers of
// the current one are consumed, so a wave always has about one tile (16 rows) in flight.
template <int NC, int MODE, bool MATH>
__global__ __launch_bounds__(256, MODE ? 1 : 2) void k_msweep(const double *__restrict__ G, long long ntiles, long long ld, const double *__restrict__ Wt,
                                                   double *__restrict__ out) {
    constexpr int PER = MODE ? 4 : 2;               // pools per lane per chunk
    constexpr int NB = NC * PER;                    // MFMAs per tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, k = lane >> 4;
    double b[NB];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < PER; ++q) b[c * PER + q] = Wt[(long long)((4 * PER) * c + PER * k + q) * 16 + i];
    const long long wstride = (long long)gridDim.x * 4;
    long long t = (long long)blockIdx.x * 4 + wave;
    if (t >= ntiles) return;
    double2 v[NC * (MODE ? 2 : 1)];
    auto rowp = [&](long long tt) { return G + (tt * 16 + i) * ld + PER * k; };
    {
        const double *r = rowp(t);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (MODE == 0) v[c] = *reinterpret_cast<const double2 *>(r + 8 * c);
            else { v[2 * c] = *reinterpret_cast<const double2 *>(r + 16 * c); v[2 * c + 1] = *reinterpret_cast<const double2 *>(r + 16 * c + 2); }
        }
    }
    double chk = 0.0;
    for (; t < ntiles; t += wstride) {
        const long long tn = t + wstride < ntiles ? t + wstride : t;
        const double *r = rowp(tn);
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        double ss = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (MODE == 0) {
                const double2 x = v[c];
                v[c] = *reinterpret_cast<const double2 *>(r + 8 * c);
                if (MATH) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, b[2 * c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, b[2 * c + 1], acc, 0, 0, 0);
                    ss = fma(x.x, x.x, ss); ss = fma(x.y, x.y, ss);
                } else ss += x.x + x.y;
            } else {
                double2 x = v[2 * c], y = v[2 * c + 1];
                if (c == NC - 1 && 16 * c + 4 * k >= ld) { x.x = x.y = y.x = y.y = 0.0; } // pools past the row's end (the next row's)
                v[2 * c] = *reinterpret_cast<const double2 *>(r + 16 * c);
                v[2 * c + 1] = *reinterpret_cast<const double2 *>(r + 16 * c + 2);
                if (MATH) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, b[4 * c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, b[4 * c + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y.x, b[4 * c + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y.y, b[4 * c + 3], acc, 0, 0, 0);
                    ss = fma(x.x, x.x, ss); ss = fma(x.y, x.y, ss); ss = fma(y.x, y.x, ss); ss = fma(y.y, y.y, ss);
                } else ss += x.x + x.y + y.x + y.y;
            }
        }
        if (MATH) {
            const d4 z = {0.0, 0.0, 0.0, 0.0};
            const d4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(ss, 1.0, z, 0, 0, 0);
            if (out && t < 1) { // tile 0: the products, for the layout check on the host
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { out[((lane >> 4) + 4 * rr) * 17 + i] = acc[rr]; if (i == 0) out[((lane >> 4) + 4 * rr) * 17 + 16] = e[rr]; }
            }
            chk += acc[0] + acc[1] + acc[2] + acc[3] + e[0];
        } else chk += ss;
    }
    if (chk == 12345.678 && out) out[400] = chk;
}

template <typename K, typename... A>
static float timeit(K kern, int grid, A... args) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, args...);
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
    const double gib = argc > 1 ? atof(argv[1]) : 14.9;   // 200 x 10 M x 8 B = 14.9 GiB
    const int n = 200;
    const long long ld = n, p = (long long)(gib * (1ull << 30) / (8.0 * ld)) / 16 * 16;
    double *G, *Wt, *out;
    CK(hipMalloc(&G, (size_t)p * ld * 8 + 4096));
    CK(hipMalloc(&Wt, 208 * 16 * 8));
    CK(hipMalloc(&out, 512 * 8));
    std::vector<double> hG((size_t)16 * ld), hW(208 * 16, 0.0);
    for (size_t x = 0; x < hG.size(); ++x) hG[x] = std::sin(0.37 * (double)x) * 0.5 + 0.5;
    for (int r = 0; r < n; ++r) for (int j = 0; j < 16; ++j) hW[r * 16 + j] = std::cos(0.11 * r + j);
    CK(hipMemset(G, 0, (size_t)p * ld * 8 + 4096));
    CK(hipMemcpy(G, hG.data(), hG.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wt, hW.data(), hW.size() * 8, hipMemcpyHostToDevice));
    const long long ntiles = p / 16;
    const double bytes = (double)p * n * 8;
    printf("device %s, %d CUs; G %lld x %d fp64 (%.2f GB), 16-locus tiles, 8 waves per CU\n", prop.gcnArchName, cus, p, n, bytes / 1e9);
    auto check = [&](const char *what) {
        std::vector<double> ho(512);
        (void)hipMemcpy(ho.data(), out, 512 * 8, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int l = 0; l < 16; ++l) {
            for (int j = 0; j < 16; ++j) { double s = 0; for (int r = 0; r < n; ++r) s += hG[l * ld + r] * hW[r * 16 + j]; worst = fmax(worst, fabs(s - ho[l * 17 + j])); }
            double s = 0; for (int r = 0; r < n; ++r) s += hG[l * ld + r] * hG[l * ld + r];
            worst = fmax(worst, fabs(s - ho[l * 17 + 16]));
        }
        printf("   %s: tile 0 against the host, worst |diff| %.3g\n", what, worst);
    };
    for (int mult : {8, 16}) {
        const int grid = cus * mult;
        float ms;
        ms = timeit(k_msweep<25, 0, false>, grid, (const double *)G, ntiles, ld, (const double *)Wt, (double *)nullptr);
        printf("grid %4d  16 B / lane (64 B of a row per instruction), loads only : %8.3f ms %8.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = timeit(k_msweep<25, 0, true>, grid, (const double *)G, ntiles, ld, (const double *)Wt, out);
        printf("grid %4d  16 B / lane, 50 MFMA + 50 FMA per tile                    : %8.3f ms %8.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        if (mult == 8) check("16 B / lane");
        ms = timeit(k_msweep<13, 1, false>, grid, (const double *)G, ntiles, ld, (const double *)Wt, (double *)nullptr);
        printf("grid %4d  32 B / lane (128 B of a row per pair), loads only        : %8.3f ms %8.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = timeit(k_msweep<13, 1, true>, grid, (const double *)G, ntiles, ld, (const double *)Wt, out);
        printf("grid %4d  32 B / lane, 52 MFMA + 52 FMA per tile                    : %8.3f ms %8.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        if (mult == 8) check("32 B / lane");
    }
    {   // the count operators' row length (24 n bytes at n = 100 = 2400 B; here 296 doubles = 2368 B = 37 chunks), loads only
        const long long ld2 = 296, p2 = (long long)(gib * (1ull << 30) / (8.0 * ld2)) / 16 * 16;
        const double bytes2 = (double)p2 * ld2 * 8;
        for (int mult : {8, 16}) {
            float ms = timeit(k_msweep<37, 0, false>, cus * mult, (const double *)G, p2 / 16, ld2, (const double *)Wt, (double *)nullptr);
            printf("grid %4d  rows of %lld B, 16 B / lane (16 rows x 64 B per instruction), loads only : %8.3f ms %8.1f GB/s\n", cus * mult, ld2 * 8, ms,
                   bytes2 / ms / 1e6);
        }
    }
    return 0;
}
