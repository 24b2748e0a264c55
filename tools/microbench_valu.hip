// fp64 VALU issue rate on gfx950: cycles per wave-instruction per SIMD for v_fma_f64 / v_add_f64 / v_mul_f64 /
// v_cvt_f64_u32 / v_rcp_f64, at 1, 2 and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mb_valu microbench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void k_valu(double *out, int iters, double a0, double b0) {
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = a0 + j + threadIdx.x;
    unsigned u = threadIdx.x + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 0) acc[j] = __builtin_fma(acc[j], b0, a0);
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(b0));
                if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(b0));
                if (OP == 3) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(acc[j]) : "v"(u));
                if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(acc[j]));
                if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(it) : );
                if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(acc[0]) : "v"(b0)); // dependent chain
            }
        }
    }
    double s = u;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char *name, double *out) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 4096;
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k_valu<OP><<<cus, 256 * wps>>>(out, 16, 1.0000001, 0.9999999);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_valu<OP><<<cus, 256 * wps>>>(out, iters, 1.0000001, 0.9999999);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 32 * wps; // wave-instructions each SIMD executes
        const double clk = prop.clockRate * 1e3;                  // Hz
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, wps, ms,
               ms * 1e-3 * clk / instr_per_simd, clk / 1e6);
    }
}

int main() {
    double *out;
    hipMalloc(&out, 64);
    run<0>("v_fma_f64", out);
    run<1>("v_add_f64", out);
    run<2>("v_mul_f64", out);
    run<3>("v_cvt_f64_u32", out);
    run<4>("v_rcp_f64", out);
    run<5>("v_cndmask_b32", out);
    run<6>("fma dep chain", out);
    return 0;
}
