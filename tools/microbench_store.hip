// HBM write rate on gfx950 for the record-store shape of k_locus_first: 8-byte-per-lane stores (512 B per wave
// instruction), 27 back to back per "unit", with a configurable amount of VALU work between units.
// hipcc --offload-arch=gfx950 -O3 -o tools/mb_store tools/microbench_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int W> // W = bytes per lane per store: 8 or 16
__global__ __launch_bounds__(256) void k_store(double *out, size_t units, int fields, int spin, double a0, int drip) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nw = (size_t)gridDim.x * 4;
    double acc = a0 + lane;
    for (size_t u = wave; u < units; u += nw) {
        if (!drip)
            for (int i = 0; i < spin; ++i) acc = __builtin_fma(acc, 0.9999999, 1e-9); // dependent chain: ~10 cycles each
        double *p = out + u * (size_t)27 * 64 * (W / 8);
        for (int f = 0; f < 27; ++f) {
            if (drip) // the same work, spread between the stores
                for (int i = 0; i < spin / 27; ++i) acc = __builtin_fma(acc, 0.9999999, 1e-9);
            if (f >= fields) continue;
            if (W == 8) p[(size_t)f * 64 + lane] = acc + f;
            else reinterpret_cast<double2 *>(p)[(size_t)f * 64 + lane] = make_double2(acc + f, acc);
        }
    }
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)2 << 30;
    double *out;
    (void)hipMalloc(&out, bytes);
    (void)hipMemset(out, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w : {8})
        for (int fields : {27, 0})
            for (int spin : {0, 2000, 2700, 20000})
                for (int drip : {0, 1})
                for (int bpc : {2}) {
                    const size_t units = bytes / ((size_t)27 * 64 * w);
                    float best = 1e9;
                    for (int rep = 0; rep < 3; ++rep) {
                        (void)hipEventRecord(e0);
                        if (w == 8) k_store<8><<<cus * bpc, 256>>>(out, units, fields, spin, 1.0, drip);
                        else k_store<16><<<cus * bpc, 256>>>(out, units, fields, spin, 1.0, drip);
                        (void)hipEventRecord(e1);
                        (void)hipEventSynchronize(e1);
                        float ms;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        best = ms < best ? ms : best;
                    }
                    printf("B/lane %2d fields/unit %2d spin %5d drip %d blocks/CU %d: %.3f ms  %.2f TB/s\n", w, fields, spin, drip, bpc, best,
                           bytes / (best * 1e-3) / 1e12);
                }
    return 0;
}
