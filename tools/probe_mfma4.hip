// probe_mfma4.hip -- lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4), found by one-hot operands:
// A = 1 in lane la only, B = 1 in lane lb only; D is non-zero in the lanes whose (block, i, j) take that product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(double *out) {
    const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    double d = 0.0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
    out[((size_t)la * 64 + lb) * 64 + lane] = d;
}
int main() {
    double *out;
    hipMalloc(&out, sizeof(double) * 64 * 64 * 64);
    hipLaunchKernelGGL(k_probe, dim3(64, 64), dim3(64), 0, 0, out);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for every A lane: which B lanes pair with it, and where the product lands
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            int cnt = 0, first = -1;
            for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) { if (first < 0) first = l; ++cnt; }
            if (cnt) printf(" B%d->D%d%s", lb, first, cnt > 1 ? "+" : "");
        }
        printf("\n");
    }
    return 0;
}
