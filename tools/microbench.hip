// microbench.hip -- calibrates the two peaks the roofline fractions are priced against:
//   (1) v_mfma_f64_16x16x4_f64 issue rate (fp64 matrix peak; not listed in the local guide)
//   (2) HBM streaming read rate of a plain 16 B/lane reduction
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void k_mfma(double *out, int iters, double a0, double b0) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_read(const double2 *__restrict__ in, size_t n16, double *out) {
    double s = 0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        double2 v = in[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 1024 * 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        const int blocks = prop.multiProcessorCount;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double n_mfma = (double)blocks * (threads / 64) * iters * 4;
        double tflops = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
        double cyc = (ms * 1e-3) * (prop.clockRate * 1e3) / (iters * 4.0 * wps);
        printf("mfma_f64_16x16x4: %d waves/SIMD  %.3f ms  %.1f TFLOP/s  ~%.1f cycles per MFMA per SIMD (at nominal clock)\n",
               wps, ms, tflops, cyc);
    }
    size_t bytes = (size_t)8 << 30;
    double2 *buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_read, dim3(prop.multiProcessorCount * 8), dim3(256), 0, 0, buf, bytes / 16, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hbm read 8 GiB: %.3f ms  %.1f GB/s\n", ms, bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
