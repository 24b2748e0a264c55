// mb_ksyrk.hip -- prototype of a REGISTER-DIRECT kinship pass: S = sum_l g_l g_l^T over a slab of loci with no LDS and no
// barriers.  A lane that loads G[locus k][pool 16 a + i] for (i = lane & 15, k = lane >> 4) holds the A operand of
// v_mfma_f64_16x16x4_f64 for pool block a AND the B operand for the same block (A[i][k] and B[k][j] sit in the same lane), so
// S_ab += frag_a * frag_b straight from the loaded registers.  The 13 * 14 / 2 = 91 tiles of a 200-pool matrix are dealt to the
// 4 waves of a workgroup (23 accumulators = 184 VGPRs each); every wave loads all 13 fragments of a 4-locus step itself (the
// four waves read the same lines: L1 / L2 hits), DEPTH steps ahead.
// What it prices: the MFMA utilisation such a loop reaches against k_kinship_syrk's LDS-staged form (0.72-0.83 of peak).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb_ksyrk.hip -o tools/mb_ksyrk      Run: tools/mb_ksyrk [loci]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NB = 13;             // pool blocks of 16 (n = 200: the last one holds 8 pools)
constexpr int NT = NB * (NB + 1) / 2;
constexpr int DEPTH = 4;

// tile t (0 .. 90) -> (a, b), a <= b, row by row
__host__ __device__ constexpr int tile_a(int t) { int a = 0; while (t >= NB - a) { t -= NB - a; ++a; } return a; }
__host__ __device__ constexpr int tile_b(int t) { int a = 0; while (t >= NB - a) { t -= NB - a; ++a; } return a + t; }

template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

template <int W, int IL>
__device__ __forceinline__ void run(const double *__restrict__ G, long long l_begin, long long l_end, int n, long long ld, double *__restrict__ part, int mode) {
    constexpr int MY = (NT - W + 3) / 4; // tiles W, W + 4, ...
    const int lane = threadIdx.x & 63, i = lane & 15, k = lane >> 4;
    d4 acc[MY];
#pragma unroll
    for (int t = 0; t < MY; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    const bool tail_ok = 16 * (NB - 1) + i < n;
    double f[DEPTH][NB];
    const long long steps = (l_end - l_begin + 3) / 4;
    auto issue = [&](double (&v)[NB], long long s) {
        long long l = l_begin + 4 * ((mode & 1) ? (s & 63) : s) + k;   // mode 1: the same 256 loci over and over (cache hits)
        l = l < l_end ? l : l_end - 1;                    // (past the slab: the last locus again; weighted 0 below)
        const double *row = G + l * ld + i;
#pragma unroll
        for (int a = 0; a < NB; ++a) v[a] = row[16 * a];
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(f[d], d);
    const long long turns = (steps + DEPTH - 1) / DEPTH;
    for (long long it = 0; it < turns; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long long s = it * DEPTH + d;
            double v[NB];
            const bool live = l_begin + 4 * s + k < l_end && s < steps;
#pragma unroll
            for (int a = 0; a < NB; ++a) v[a] = live ? f[d][a] : 0.0;
            v[NB - 1] = tail_ok ? v[NB - 1] : 0.0;
            if (IL != 2) issue(f[d], s + DEPTH); // (2: no loads inside the loop -- the matrix pipe alone, 23 accumulators, one wave per SIMD)
            static_for<MY>([&](auto tc) {
                constexpr int t = decltype(tc)::value * 4 + W, ta = tile_a(t), tb = tile_b(t);
                acc[decltype(tc)::value] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ta], v[tb], acc[decltype(tc)::value], 0, 0, 0);
            });
            if (IL == 1) { // interleave: one MFMA, then what fits in its 64-cycle shadow (a load, two vector ALU operations)
#pragma unroll
                for (int t = 0; t < MY; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
                }
            }
        }
    }
    // D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    static_for<MY>([&](auto tc) {
        constexpr int t = decltype(tc)::value * 4 + W;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(size_t)t * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[decltype(tc)::value][r];
    });
}

template <int IL>
__global__ __launch_bounds__(256, 1) void k_ksyrk(const double *__restrict__ G, long long p, int n, long long ld, double *__restrict__ part, int mode) {
    const long long per = (p + gridDim.x - 1) / gridDim.x;
    const long long l_begin = (long long)blockIdx.x * per, l_end = l_begin + per < p ? l_begin + per : p;
    if (l_begin >= l_end) return;
    double *mine = part + (size_t)blockIdx.x * NT * 256;
    switch (threadIdx.x >> 6) {
    case 0: run<0, IL>(G, l_begin, l_end, n, ld, mine, mode); break;
    case 1: run<1, IL>(G, l_begin, l_end, n, ld, mine, mode); break;
    case 2: run<2, IL>(G, l_begin, l_end, n, ld, mine, mode); break;
    default: run<3, IL>(G, l_begin, l_end, n, ld, mine, mode); break;
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long long p = argc > 1 ? atoll(argv[1]) : 10000000;
    const int n = 200;
    const long long ld = n;
    double *G, *part;
    CK(hipMalloc(&G, (size_t)p * ld * 8 + 4096));
    CK(hipMalloc(&part, (size_t)cus * NT * 256 * 8));
    std::vector<double> hG((size_t)p * ld > 4000000 ? 4000000 : (size_t)p * ld);
    for (size_t x = 0; x < hG.size(); ++x) hG[x] = std::sin(0.37 * (double)x) * 0.5 + 0.5;
    CK(hipMemset(G, 0, (size_t)p * ld * 8 + 4096));
    CK(hipMemcpy(G, hG.data(), hG.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 5; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 4) hipLaunchKernelGGL(k_ksyrk<2>, dim3(cus), dim3(256), 0, 0, (const double *)G, p, n, ld, part, mode);
        else if (mode & 2) hipLaunchKernelGGL(k_ksyrk<1>, dim3(cus), dim3(256), 0, 0, (const double *)G, p, n, ld, part, mode);
        else hipLaunchKernelGGL(k_ksyrk<0>, dim3(cus), dim3(256), 0, 0, (const double *)G, p, n, ld, part, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double exec_flops = (double)NT * 2048.0 * (double)((p + 3) / 4);
    printf("device %s, %d CUs; mode %d; %lld loci x %d pools: %.3f ms, %d tiles -> executed %.1f TFLOP/s (%.3f of 78.6), read %.1f GB/s\n", prop.gcnArchName, cus, mode, p, n,
           best, NT, exec_flops / best / 1e9, exec_flops / best / 1e9 / 78.6, (double)p * n * 8 / best / 1e6);
    }
    // check S[0..15][0..15] and a far tile against the host over the first 20000 loci (device partials summed)
    {
        const long long pc = 20000;
        hipLaunchKernelGGL(k_ksyrk<1>, dim3(cus), dim3(256), 0, 0, (const double *)G, pc, n, ld, part, 0);
        std::vector<double> hp((size_t)cus * NT * 256);
        CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
        const long long per = (pc + cus - 1) / cus;
        const int nblk = (int)((pc + per - 1) / per);
        double worst = 0.0;
        for (int t : {0, 5, 12, 50, 90}) {
            const int a = tile_a(t), b = tile_b(t);
            for (int ii = 0; ii < 16; ++ii) for (int jj = 0; jj < 16; ++jj) {
                const int pa = 16 * a + ii, pb = 16 * b + jj;
                double ref = 0.0;
                if (pa < n && pb < n) for (long long l = 0; l < pc; ++l) ref += hG[l * ld + pa] * hG[l * ld + pb];
                double got = 0.0;
                for (int bk = 0; bk < nblk; ++bk) got += hp[((size_t)bk * NT + t) * 256 + ii * 16 + jj];
                worst = fmax(worst, fabs(got - ref) / fmax(1.0, fabs(ref)));
            }
        }
        printf("   check over %lld loci, 5 tiles: worst relative difference %.3g\n", pc, worst);
    }
    return 0;
}
