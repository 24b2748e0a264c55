// mb_mfma4.hip -- issue rate of v_mfma_f64_4x4x4_4b_f64 against v_mfma_f64_16x16x4_f64 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC> __global__ void k16(double *out, int iters, double a0, double b0) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k4(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    double *out; (void)hipMalloc(&out, sizeof(double) * 1024 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps, blocks = prop.multiProcessorCount;
        float ms16 = 0, ms4 = 0;
        for (int rep = 0; rep < 2; ++rep) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms16, e0, e1); }
        for (int rep = 0; rep < 2; ++rep) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k4<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms4, e0, e1); }
        const double c16 = ms16 * 1e-3 * prop.clockRate * 1e3 / (iters * 4.0 * wps), c4 = ms4 * 1e-3 * prop.clockRate * 1e3 / (iters * 8.0 * wps);
        printf("%d waves/SIMD: 16x16x4 %.1f cycles per instruction per SIMD (%.1f TFLOP/s)   4x4x4_4b %.1f cycles (%.1f TFLOP/s)\n", wps, c16,
               (double)blocks * threads / 64 * iters * 4 * 2048.0 / (ms16 * 1e-3) / 1e12, c4, (double)blocks * threads / 64 * iters * 8 * 512.0 / (ms4 * 1e-3) / 1e12);
    }
    return 0;
}
