// pg_popgen.hip -- fst (popgen/fst.rs:10-115, :158-200) and theta_pi / heterozygosity (popgen/pi.rs:10-113) on the
// resident locus-major genotype matrix G (p x ld, one row per allele column) of the loader, with the coverages the
// loader writes beside it (row of a locus' first column = the pools' depths over the surviving alleles,
// sync.rs:1142-1152).  A "locus" is a run of columns with the same (chromosome, position) (count_loci, sync.rs:73-97):
// locus_col[l] .. locus_col[l+1] are its rows of G.
//
//   k_pop_locus      thread = (locus, pool): sum of squared frequencies over the locus' alleles, the n/(n-1) factor,
//                    q1 (fst.rs:69-75) and pi (pi.rs:51-54): two L x n arrays, 1/a of the size of G
//   k_pop_check      thread = locus: the reference's guard that the frequencies of a locus sum to n (fst.rs:66), in
//                    ndarray's summation orders
//   k_range_mean_1d  thread = (window, pool): mean of pi over the window's loci, summed left to right as
//                    mean_axis does (pi.rs:84-95)
//   k_fst_ranges     block = (range of loci, 32 x 32 tile of pool pairs, upper triangle), thread = 4 x 4 pairs: per locus
//                    q2 = sum_a g_j g_k, the clamped ratio (fst.rs:76-91), summed left to right over the range;
//                    ranges are the windows (divide: per-window means, :178-199) or equal chunks of the genome
//                    (partial sums, then k_chunk_reduce -> the genome-wide mean, :145)
// All arithmetic is fp64 VALU; the matrix is read a few times (once per overlapping window + once for the genome-wide
// mean) through L2: these are elementwise n^2-per-locus reductions (2.0e4 pairs x ~30 flops per locus at n = 200), bound
// by the fp64 vector pipe, not by HBM and not GEMM-shaped (the ratio is taken per locus before averaging).
#include "pg_common.h"
#include <cmath>
#include <vector>

namespace {

constexpr double POP_EPS = 2.220446049250313e-16; // f64::EPSILON

__global__ __launch_bounds__(256) void k_pop_locus(const double *__restrict__ G, const double *__restrict__ cov,
                                                   const int64_t *__restrict__ locus_col, int64_t L, int n, int64_t ld,
                                                   double *__restrict__ Q1, double *__restrict__ PI) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t l = gid / n;
    if (l >= L) return;
    const int pool = (int)(gid - l * n);
    const int64_t c0 = locus_col[l], c1 = locus_col[l + 1];
    double s = 0.0;
    for (int64_t c = c0; c < c1; ++c) {
        const double g = G[c * ld + pool];
        s = s + g * g;
    }
    const double nj = cov[c0 * ld + pool];
    const double r = nj / (nj - 1.00 + POP_EPS);
    Q1[gid] = (s * r) + (1.00 - r);
    PI[gid] = fabs((s * r) - r);
}

// |sum_i (sum_a g) - n| <= eps with ndarray's orders: the lane of < 8 alleles left to right, the n row sums by
// unrolled_fold (8 partial sums, then (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7), then the tail left to right)
__global__ __launch_bounds__(256) void k_pop_check(const double *__restrict__ G, const int64_t *__restrict__ locus_col,
                                                   int64_t L, int n, int64_t ld, int *__restrict__ bad) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    const int64_t c0 = locus_col[l], c1 = locus_col[l + 1];
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto rowsum = [&](int i) {
        double s = 0.0;
        if (c1 - c0 >= 8) { // more alleles than the format has; kept for the arithmetic's sake
            double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int64_t c = c0;
            for (; c1 - c >= 8; c += 8)
                for (int u = 0; u < 8; ++u) q[u] = q[u] + G[(c + u) * ld + i];
            s = s + (q[0] + q[4]); s = s + (q[1] + q[5]); s = s + (q[2] + q[6]); s = s + (q[3] + q[7]);
            for (; c < c1; ++c) s = s + G[c * ld + i];
            return s;
        }
        for (int64_t c = c0; c < c1; ++c) s = s + G[c * ld + i];
        return s;
    };
    int i = 0;
    for (; n - i >= 8; i += 8)
        for (int u = 0; u < 8; ++u) p[u] = p[u] + rowsum(i + u);
    double acc = 0.0;
    acc = acc + (p[0] + p[4]); acc = acc + (p[1] + p[5]); acc = acc + (p[2] + p[6]); acc = acc + (p[3] + p[7]);
    for (; i < n; ++i) acc = acc + rowsum(i);
    if (!(fabs(acc - (double)n) <= POP_EPS)) atomicOr(bad, 1);
}

__global__ __launch_bounds__(256) void k_range_mean_1d(const double *__restrict__ V, const int64_t *__restrict__ head,
                                                       const int64_t *__restrict__ tail, int64_t nw, int n,
                                                       double *__restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t w = gid / n;
    if (w >= nw) return;
    const int pool = (int)(gid - w * n);
    double s = 0.0;
    for (int64_t l = head[w]; l <= tail[w]; ++l) s = s + V[l * n + pool];
    out[gid] = s / (double)(tail[w] + 1 - head[w]);
}

// a / b as hipcc's own fp64 division computes it for normal-range operands, minus the scaling and fix-up steps those
// operands do not need (1 - q2 + eps lies in [eps, 1 + eps]; the quotient is 0, NaN or of ordinary size): v_rcp_f64, two
// Newton steps, then quotient, fused residual, fused correction.  Same helper and same argument as pg_locus_ops.hip; the
// per-window table is compared bit for bit with the oracle's `/` (tests/test_gpu_popgen.py).
__device__ __forceinline__ double pop_div(double a, double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double r1 = fma(fma(-b, r0, 1.0), r0, r0);
    const double r = fma(fma(-b, r1, 1.0), r1, r1);
    const double q0 = a * r;
    return fma(fma(-b, q0, a), r, q0);
}

constexpr int FT = 8;       // threads per tile edge (one wave per block)
constexpr int FR = 4;       // pools per thread along each edge: a block covers a (FT*FR)^2 = 32 x 32 tile of pairs
constexpr int FTILE = FT * FR;

// grid.x = range, grid.y = upper-triangular tile id; out[range][n][n] (both triangles written).  Every thread keeps a
// 4 x 4 block of pairs: 8 frequencies and 8 q1 per allele row serve 16 ratios, and the 16 independent divisions
// overlap each other's latency.  Per pair the operations and their order are those of the reference (fst.rs:69-91).
__global__ __launch_bounds__(FT * FT) void k_fst_ranges(const double *__restrict__ G, const double *__restrict__ Q1,
                                                        const int64_t *__restrict__ locus_col,
                                                        const int64_t *__restrict__ head, const int64_t *__restrict__ tail,
                                                        int n, int64_t ld, int ntile, int divide, double *__restrict__ out) {
    // tile id -> (tj, tk) with tj <= tk
    int t = blockIdx.y, tj = 0;
    while (t >= ntile - tj) { t -= ntile - tj; ++tj; }
    const int tk = tj + t;
    const int j0 = tj * FTILE + (threadIdx.x / FT) * FR, k0 = tk * FTILE + (threadIdx.x % FT) * FR;
    int jj[FR], kk[FR];
#pragma unroll
    for (int u = 0; u < FR; ++u) { jj[u] = min(j0 + u, n - 1); kk[u] = min(k0 + u, n - 1); }
    const int64_t w = blockIdx.x;
    const int64_t l0 = head[w], l1 = tail[w];
    double s[FR][FR];
#pragma unroll
    for (int u = 0; u < FR; ++u)
#pragma unroll
        for (int v = 0; v < FR; ++v) s[u][v] = 0.0;
    int64_t c0 = locus_col[l0];
    for (int64_t l = l0; l <= l1; ++l) {
        const int64_t c1 = locus_col[l + 1];
        double q2[FR][FR];
#pragma unroll
        for (int u = 0; u < FR; ++u)
#pragma unroll
            for (int v = 0; v < FR; ++v) q2[u][v] = 0.0;
        for (int64_t c = c0; c < c1; ++c) {
            const double *row = G + c * ld;
            double gj[FR], gk[FR];
#pragma unroll
            for (int u = 0; u < FR; ++u) { gj[u] = row[jj[u]]; gk[u] = row[kk[u]]; }
#pragma unroll
            for (int u = 0; u < FR; ++u)
#pragma unroll
                for (int v = 0; v < FR; ++v) q2[u][v] = q2[u][v] + (gj[u] * gk[v]);
        }
        const double *qrow = Q1 + l * n;
        double q1j[FR], q1k[FR];
#pragma unroll
        for (int u = 0; u < FR; ++u) { q1j[u] = qrow[jj[u]]; q1k[u] = qrow[kk[u]]; }
#pragma unroll
        for (int u = 0; u < FR; ++u)
#pragma unroll
            for (int v = 0; v < FR; ++v) {
                const double fu = pop_div(0.5 * (q1j[u] + q1k[v]) - q2[u][v], 1.00 - q2[u][v] + POP_EPS);
                s[u][v] = s[u][v] + (fu < 0.0 ? 0.0 : (fu > 1.0 ? 1.0 : fu)); // NaN passes through, as in the reference
            }
        c0 = c1;
    }
    const double den = (double)(l1 + 1 - l0);
    double *o = out + (size_t)w * n * n;
#pragma unroll
    for (int u = 0; u < FR; ++u)
#pragma unroll
        for (int v = 0; v < FR; ++v) {
            const int j = j0 + u, k = k0 + v;
            if (j < n && k < n && j <= k) {
                const double r = divide ? s[u][v] / den : s[u][v];
                o[(size_t)j * n + k] = r;
                o[(size_t)k * n + j] = r; // every term is symmetric in (j, k): x*y, q1_j + q1_k
            }
        }
}

__global__ __launch_bounds__(256) void k_chunk_reduce(const double *__restrict__ part, int64_t nchunks, int64_t nn,
                                                      double denom, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double s = 0.0;
    for (int64_t c = 0; c < nchunks; ++c) s = s + part[c * nn + i];
    out[i] = s / denom;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

int check_shape(pg_ctx *ctx, const double *G_dev, const double *cov_dev, int64_t p, int n, int64_t ld,
                const int64_t *locus_col, int64_t L, const int64_t *wh, const int64_t *wt, int64_t nw, const char *who) {
    PG_CHECK(ctx, G_dev && cov_dev && locus_col && p > 0 && n >= 1 && ld >= n && L >= 1, "%s: bad arguments", who);
    PG_CHECK(ctx, locus_col[0] == 0 && locus_col[L] == p, "%s: locus_col must run from 0 to p", who);
    for (int64_t l = 0; l < L; ++l) PG_CHECK(ctx, locus_col[l] < locus_col[l + 1], "%s: empty locus %lld", who, (long long)l);
    PG_CHECK(ctx, nw >= 0 && (nw == 0 || (wh && wt)), "%s: windows missing", who);
    for (int64_t w = 0; w < nw; ++w)
        PG_CHECK(ctx, wh[w] >= 0 && wh[w] <= wt[w] && wt[w] < L, "%s: window %lld out of range", who, (long long)w);
    return PG_OK;
}

} // namespace

// define_sliding_windows (base/helpers.rs:294-403)
extern "C" int64_t pg_host_sliding_windows(const int32_t *chr, const uint64_t *pos, int64_t l, uint64_t window_size_bp,
                                           uint64_t window_slide_size_bp, uint64_t min_loci_per_window, int64_t *out_head,
                                           int64_t *out_tail) {
    if (l <= 0 || !chr || !pos || !out_head || !out_tail) return 0;
    std::vector<int64_t> head{0}, tail{0};
    std::vector<uint64_t> cnt{1};
    bool next_found = false;
    int64_t next_head = 0;
    for (int64_t i = 1; i < l; ++i) {
        const int64_t h = head.back();
        if (chr[i] != chr[h] || pos[i] > pos[h] + window_size_bp) {
            if (next_found) i = next_head;                 // the next window starts inside the ending one (:331-335)
            if (cnt.back() >= min_loci_per_window) { head.push_back(i); tail.push_back(i); cnt.push_back(1); }
            else { head.back() = i; cnt.back() = 1; }     // too few loci: the slot is reused, its tail stays (:350-357)
            next_found = false;
        } else {
            tail.back() = i;
            cnt.back() += 1;
            if (!next_found && pos[i] >= pos[h] + window_slide_size_bp) { next_found = true; next_head = i; }
        }
    }
    int64_t no = 0;
    for (size_t w = 0; w < head.size(); ++w)               // windows ending where the previous one ends are dropped (:380-391)
        if (w == 0 || tail[w] != out_tail[no - 1]) { out_head[no] = head[w]; out_tail[no] = tail[w]; ++no; }
    return no;
}

extern "C" int pg_pi_dev(pg_ctx *ctx, const double *G_dev, const double *cov_dev, int64_t p, int n, int64_t ld,
                         const int64_t *locus_col, int64_t L, const int64_t *win_head, const int64_t *win_tail,
                         int64_t n_windows, double *pi_win, double *pi_mean) {
    if (!ctx) return PG_ERR_INVALID;
    int rc = check_shape(ctx, G_dev, cov_dev, p, n, ld, locus_col, L, win_head, win_tail, n_windows, "pi");
    if (rc) return rc;
    PG_CHECK(ctx, n_windows >= 1 && pi_win && pi_mean, "pi: There were no windows defined."); // pi.rs:81
    PG_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf lc, q1, pi, wh, wt, out;
    PG_HIP(ctx, hipMalloc(&lc.p, sizeof(int64_t) * (L + 1)));
    PG_HIP(ctx, hipMalloc(&q1.p, sizeof(double) * (size_t)L * n));
    PG_HIP(ctx, hipMalloc(&pi.p, sizeof(double) * (size_t)L * n));
    PG_HIP(ctx, hipMalloc(&wh.p, sizeof(int64_t) * n_windows));
    PG_HIP(ctx, hipMalloc(&wt.p, sizeof(int64_t) * n_windows));
    PG_HIP(ctx, hipMalloc(&out.p, sizeof(double) * (size_t)n_windows * n));
    PG_HIP(ctx, hipMemcpyAsync(lc.p, locus_col, sizeof(int64_t) * (L + 1), hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(wh.p, win_head, sizeof(int64_t) * n_windows, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(wt.p, win_tail, sizeof(int64_t) * n_windows, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pop_locus, dim3((unsigned)(((size_t)L * n + 255) / 256)), dim3(256), 0, ctx->stream, G_dev, cov_dev,
                       lc.as<int64_t>(), L, n, ld, q1.as<double>(), pi.as<double>());
    hipLaunchKernelGGL(k_range_mean_1d, dim3((unsigned)(((size_t)n_windows * n + 255) / 256)), dim3(256), 0, ctx->stream,
                       pi.as<double>(), wh.as<int64_t>(), wt.as<int64_t>(), n_windows, n, out.as<double>());
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipMemcpyAsync(pi_win, out.p, sizeof(double) * (size_t)n_windows * n, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < n; ++j) { // mean_axis(Axis(0)) over the windows (pi.rs:133): left to right
        double s = 0.0;
        for (int64_t w = 0; w < n_windows; ++w) s = s + pi_win[(size_t)w * n + j];
        pi_mean[j] = s / (double)n_windows;
    }
    return PG_OK;
}

extern "C" int pg_fst_dev(pg_ctx *ctx, const double *G_dev, const double *cov_dev, int64_t p, int n, int64_t ld,
                          const int64_t *locus_col, int64_t L, const int64_t *win_head, const int64_t *win_tail,
                          int64_t n_windows, double *fst_mean, double *fst_win) {
    if (!ctx) return PG_ERR_INVALID;
    int rc = check_shape(ctx, G_dev, cov_dev, p, n, ld, locus_col, L, win_head, win_tail, n_windows, "fst");
    if (rc) return rc;
    PG_CHECK(ctx, fst_mean && (n_windows == 0 || fst_win), "fst: null output");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    // the genome-wide mean: equal chunks of loci -> partial sums -> one ordered reduction
    const int64_t chunk = std::max<int64_t>(64, (L + 2047) / 2048);
    const int64_t nchunks = (L + chunk - 1) / chunk;
    std::vector<int64_t> ch(nchunks), ct(nchunks);
    for (int64_t c = 0; c < nchunks; ++c) { ch[c] = c * chunk; ct[c] = std::min<int64_t>(L, (c + 1) * chunk) - 1; }
    DevBuf lc, q1, pi, wh, wt, chh, cht, part, mean, out, bad;
    PG_HIP(ctx, hipMalloc(&lc.p, sizeof(int64_t) * (L + 1)));
    PG_HIP(ctx, hipMalloc(&q1.p, sizeof(double) * (size_t)L * n));
    PG_HIP(ctx, hipMalloc(&pi.p, sizeof(double) * (size_t)L * n));
    PG_HIP(ctx, hipMalloc(&chh.p, sizeof(int64_t) * nchunks));
    PG_HIP(ctx, hipMalloc(&cht.p, sizeof(int64_t) * nchunks));
    PG_HIP(ctx, hipMalloc(&part.p, sizeof(double) * nchunks * nn));
    PG_HIP(ctx, hipMalloc(&mean.p, sizeof(double) * nn));
    PG_HIP(ctx, hipMalloc(&bad.p, sizeof(int)));
    PG_HIP(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(lc.p, locus_col, sizeof(int64_t) * (L + 1), hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(chh.p, ch.data(), sizeof(int64_t) * nchunks, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(cht.p, ct.data(), sizeof(int64_t) * nchunks, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pop_check, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream, G_dev, lc.as<int64_t>(), L, n,
                       ld, bad.as<int>());
    hipLaunchKernelGGL(k_pop_locus, dim3((unsigned)(((size_t)L * n + 255) / 256)), dim3(256), 0, ctx->stream, G_dev, cov_dev,
                       lc.as<int64_t>(), L, n, ld, q1.as<double>(), pi.as<double>());
    int hbad = 0;
    PG_HIP(ctx, hipMemcpyAsync(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hbad) // the reference's assert!((g.sum_axis(Axis(1)).sum() - n as f64).abs() <= f64::EPSILON) (fst.rs:66)
        return pg_fail(ctx, PG_ERR_INVALID, "fst: the allele frequencies of a locus do not sum up to one in every pool");
    const int ntile = (n + FTILE - 1) / FTILE;
    const int ntri = ntile * (ntile + 1) / 2;
    hipLaunchKernelGGL(k_fst_ranges, dim3((unsigned)nchunks, ntri), dim3(FT * FT), 0, ctx->stream, G_dev, q1.as<double>(),
                       lc.as<int64_t>(), chh.as<int64_t>(), cht.as<int64_t>(), n, ld, ntile, 0, part.as<double>());
    hipLaunchKernelGGL(k_chunk_reduce, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, part.as<double>(), nchunks,
                       (int64_t)nn, (double)L, mean.as<double>());
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipMemcpyAsync(fst_mean, mean.p, sizeof(double) * nn, hipMemcpyDeviceToHost, ctx->stream));
    if (n_windows > 0) {
        PG_HIP(ctx, hipMalloc(&wh.p, sizeof(int64_t) * n_windows));
        PG_HIP(ctx, hipMalloc(&wt.p, sizeof(int64_t) * n_windows));
        PG_HIP(ctx, hipMemcpyAsync(wh.p, win_head, sizeof(int64_t) * n_windows, hipMemcpyHostToDevice, ctx->stream));
        PG_HIP(ctx, hipMemcpyAsync(wt.p, win_tail, sizeof(int64_t) * n_windows, hipMemcpyHostToDevice, ctx->stream));
        // the per-window table in slabs of windows (it is n^2 doubles per window)
        const int64_t slab = std::max<int64_t>(1, std::min<int64_t>(n_windows, (int64_t)(((size_t)1 << 30) / (nn * sizeof(double)))));
        PG_HIP(ctx, hipMalloc(&out.p, sizeof(double) * slab * nn));
        for (int64_t w0 = 0; w0 < n_windows; w0 += slab) {
            const int64_t nwb = std::min<int64_t>(slab, n_windows - w0);
            hipLaunchKernelGGL(k_fst_ranges, dim3((unsigned)nwb, ntri), dim3(FT * FT), 0, ctx->stream, G_dev, q1.as<double>(),
                               lc.as<int64_t>(), wh.as<int64_t>() + w0, wt.as<int64_t>() + w0, n, ld, ntile, 1, out.as<double>());
            PG_HIP(ctx, hipGetLastError());
            PG_HIP(ctx, hipMemcpyAsync(fst_win + (size_t)w0 * nn, out.p, sizeof(double) * nwb * nn, hipMemcpyDeviceToHost, ctx->stream));
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PG_OK;
}
