// mb_overlap.hip -- does fp64 VALU work between the issue of a wave's loads and their use delay the loads?
// (k_ols_sweep measured: memory-only 2.55 ms, compute-only 0.97 ms, together 3.18 ms -- not max(), nearly the sum.)
// Arms, all with the sweep's geometry (super-rows of 3200 B, 64 rows x 256 B per chunk, 8 waves per CU):
//   regs  : 16 global_load_dwordx4 -> NF fp64 FMAs on other registers -> use the loaded values
//   glds  : 17 global_load_lds_dwordx4 (LDS-DMA, 272-B pitch image via the SOURCE address) -> the same FMAs -> one LDS read
//   valu  : the FMAs alone                      mem : NF = 0
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb_overlap.hip -o tools/mb_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE> // 0 regs, 1 glds, 2 valu only
__global__ __launch_bounds__(256) void k_overlap(const char *__restrict__ base, long long ntiles, long long stride, int nf, double *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *tile = lds + wave * 17408;
    const int lr = lane >> 4, piece = lane & 15;
    const long long wstride = (long long)gridDim.x * 4;
    const int nch = (int)((stride + 255) / 256);
    double s = 0.0;
    double a0 = 1.0 + lane * 1e-9, a1 = 0.5, a2 = 0.25, a3 = 0.125, a4 = 2.0, a5 = 3.0, a6 = 4.0, a7 = 5.0;
    const double m = 0.999999, c = 1e-7;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += wstride) {
        const char *tb = base + t * 64 * stride;
        for (int ch = 0; ch < nch; ++ch) {
            double2 v[16];
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    long long in_row = (long long)ch * 256 + 16 * piece;
                    in_row = in_row < stride - 16 ? in_row : stride - 16;
                    v[r] = *reinterpret_cast<const double2 *>(tb + (long long)(4 * r + lr) * stride + in_row);
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 17; ++r) { // LDS slot q = 64 r + lane of the 64 x 17 slot image: row q / 17, piece q % 17 (16 = pad)
                    const int q = 64 * r + lane;
                    const int row = q / 17, pc = q - row * 17;
                    long long in_row = (long long)ch * 256 + 16 * (pc < 16 ? pc : 15);
                    in_row = in_row < stride - 16 ? in_row : stride - 16;
                    const char *src = tb + (long long)row * stride + in_row;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                                     (__attribute__((address_space(3))) void *)(tile + r * 1024), 16, 0, 0);
                }
            }
            for (int i = 0; i < nf; i += 8) { // 8 independent chains
                a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
                a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
            }
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += v[r].x + v[r].y;
            } else if (MODE == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const double2 x = *reinterpret_cast<const double2 *>(tile + lane * 272);
                s += x.x + x.y;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    s += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678) out[0] = s;
}

template <typename K>
static float timeit(K kern, int grid, size_t shmem, const char *buf, long long ntiles, long long stride, int nf, double *out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, buf, ntiles, stride, nf, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return best;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const size_t bytes = (size_t)16 << 30;
    char *buf; double *out;
    CK(hipMalloc(&buf, bytes + (1 << 20)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes + (1 << 20)));
    const size_t shmem = 4 * 17408;
    const int grid = prop.multiProcessorCount * 8;
    const long long stride = 3200, ntiles = (long long)(bytes / (64 * stride));
    const double gb = (double)ntiles * 64 * stride / 1e9;
    printf("%d CUs, %.2f GB, 8 waves per CU, 16 KB per wave chunk\n", prop.multiProcessorCount, gb);
    for (int nf : {0, 64, 128, 256, 512, 1024}) {
        const float tr = timeit(k_overlap<0>, grid, shmem, buf, ntiles, stride, nf, out);
        const float tg = timeit(k_overlap<1>, grid, shmem, buf, ntiles, stride, nf, out);
        const float tv = timeit(k_overlap<2>, grid, shmem, buf, ntiles, stride, nf, out);
        printf("fp64 FMAs per chunk %5d: regs %7.3f ms (%6.0f GB/s)   glds %7.3f ms (%6.0f GB/s)   valu alone %7.3f ms\n", nf, tr,
               gb / tr * 1e3, tg, gb / tg * 1e3, tv);
    }
    return 0;
}
